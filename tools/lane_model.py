"""Development aid (CPU, numpy only): what fraction of the lane-evaluations the tile pass ISSUES land inside a
gaussian's alpha >= 1/255 box -- the kernels' own scheduling rules (csrc/gi2d_raster_core.h) replayed on a scene:

  forward   wave w of a tile owns pixel rows 4w..4w+3; it keeps two lists (entries whose box reaches columns 0..7 /
            8..15 of that strip), walks them two entries per trip, and needs max(|left|, |right|) / 2 trips
            (fwd_pixel_half_lists): issued = trips x 2 entries x 64 lanes, useful = box pixels inside the half strip.
  backward  items = (entry, aligned row pair), one trip per COLUMN of the box with the pair's two rows packed; items go
            to the lanes by length class of their gaussian (9..16, 5..8, 3..4, 1..2 columns), 256 per round, and a
            wave runs as long as its longest item (bwd_run_tile): issued = 64 lanes x 2 rows x longest item per wave,
            useful = box pixels.

usage: python tools/lane_model.py [N H W seed]     (the bench scene: 50000 512 768 3047; prints one JSON object that
       tools/make_profiles_rounds.py stores next to the VALU counters in profiles/traffic.json)"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import synth_cholesky  # noqa: E402


def project_cholesky(xyz, L, h, w):
    """foward2d.cu:12-69 in float64 (the model only needs boxes to a fraction of a pixel)."""
    cx, cy = 0.5 * w * (xyz[:, 0] + 1.0) - 0.5 + 0.5, 0.5 * h * (xyz[:, 1] + 1.0) - 0.5 + 0.5
    # ndc2pix(x, W) = 0.5 * W * (x + 1) - 0.5 + 0.5 in the reference's 2D projection (pixel centres at integers + 0.5 off)
    l11, l21, l22 = L[:, 0].astype(np.float64), L[:, 1].astype(np.float64), L[:, 2].astype(np.float64)
    sxx, sxy, syy = l11 * l11, l11 * l21, l21 * l21 + l22 * l22
    det = sxx * syy - sxy * sxy
    a, b, c = syy / det, -sxy / det, sxx / det
    mid = 0.5 * (sxx + syy)
    lam = mid + np.sqrt(np.maximum(0.1, mid * mid - det))
    radius = np.ceil(3.0 * np.sqrt(lam))
    return cx, cy, a, b, c, radius


def main():
    a = sys.argv[1:]
    n, h, w = (int(a[0]) if a else 50000), (int(a[1]) if len(a) > 1 else 512), (int(a[2]) if len(a) > 2 else 768)
    seed = int(a[3]) if len(a) > 3 else 3047
    xyz, L, col, op = synth_cholesky(n, h, w, seed)
    cx, cy, ca, cb, cc, rad = project_cholesky(xyz, L, h, w)
    opac = op.reshape(-1).astype(np.float64)
    tx_n, ty_n = (w + 15) // 16, (h + 15) // 16
    # alpha >= 1/255 box (gi2d_common.h::cull_extent)
    det = ca * cc - cb * cb
    tau2 = 2.0 * np.log(opac * 255.0) * 1.0002 + 1e-3
    hx = np.sqrt(tau2 / det * cc) * 1.0002 + 0.0625
    hy = np.sqrt(tau2 / det * ca) * 1.0002 + 0.0625
    x0, x1 = np.ceil(cx - hx), np.floor(cx + hx)
    y0, y1 = np.ceil(cy - hy), np.floor(cy + hy)
    # tile box of the 3-sigma radius (helpers.cuh:16-50: truncation, exclusive max)
    tmnx = np.clip(np.trunc((cx - rad) / 16.0), 0, tx_n).astype(int)
    tmxx = np.clip(np.trunc((cx + rad) / 16.0 + 1), 0, tx_n).astype(int)
    tmny = np.clip(np.trunc((cy - rad) / 16.0), 0, ty_n).astype(int)
    tmxy = np.clip(np.trunc((cy + rad) / 16.0 + 1), 0, ty_n).astype(int)
    tiles = [[] for _ in range(tx_n * ty_n)]
    for g in range(n):
        for ty in range(tmny[g], tmxy[g]):
            for tx in range(tmnx[g], tmxx[g]):
                tiles[ty * tx_n + tx].append(g)
    m = sum(len(t) for t in tiles)
    f_issued = f_useful = b_issued = b_useful = 0
    items_total = reaching = 0
    for t, ids in enumerate(tiles):
        ids = ids[:256]
        ty, tx = divmod(t, tx_n)
        last_row = min(15, h - 1 - 16 * ty)
        boxes = []
        for g in ids:
            r0, r1 = max(int(y0[g]) - 16 * ty, 0), min(int(y1[g]) - 16 * ty, last_row)
            c0, c1 = max(int(x0[g]) - 16 * tx, 0), min(int(x1[g]) - 16 * tx, 15)
            if r1 >= r0 and c1 >= c0:
                boxes.append((r0, r1, c0, c1))
        reaching += len(boxes)
        # forward
        for wv in range(4):
            nl = nr = 0
            for r0, r1, c0, c1 in boxes:
                lo, hi = max(r0, 4 * wv), min(r1, 4 * wv + 3)
                if hi < lo:
                    continue
                rows = hi - lo + 1
                if c0 <= 7:
                    nl += 1
                    f_useful += rows * (min(c1, 7) - c0 + 1)
                if c1 >= 8:
                    nr += 1
                    f_useful += rows * (c1 - max(c0, 8) + 1)
            f_issued += math.ceil(max(nl, nr) / 2) * 2 * 64
        # backward
        per_class = [[], [], [], []]
        for r0, r1, c0, c1 in boxes:
            nc = c1 - c0 + 1
            cls = 0 if nc >= 9 else 1 if nc >= 5 else 2 if nc >= 3 else 3
            for p in range(r0 >> 1, (r1 >> 1) + 1):
                rows = min(r1, 2 * p + 1) - max(r0, 2 * p) + 1
                per_class[cls].append((nc, rows))
        order = [it for cl in per_class for it in cl]
        items_total += len(order)
        for r in range(0, len(order), 256):
            rnd = order[r:r + 256]
            for w0 in range(0, len(rnd), 64):
                wave = rnd[w0:w0 + 64]
                b_issued += 64 * 2 * max(nc for nc, _ in wave)
                b_useful += sum(nc * rows for nc, rows in wave)
    out = {"scene": f"N={n} {w}x{h} seed {seed} (tests/helpers.py::synth_cholesky)", "num_intersects": m,
           "entries_reaching_their_tile": reaching, "backward_items_per_tile": items_total / len(tiles),
           "useful_lane_frac_fwd": f_useful / f_issued, "useful_lane_frac_bwd": b_useful / b_issued,
           # absolute counts (DESIGN.md "what a 16x16-tile design caps at"): pixel x gaussian evaluations inside a box,
           # and the lane-evaluations the kernels' scheduling issues for them (backward: two pixel rows per lane-trip)
           "useful_pair_evals_fwd": int(f_useful), "issued_lane_evals_fwd": int(f_issued),
           "useful_pair_evals_bwd": int(b_useful), "issued_lane_evals_bwd": int(b_issued),
           "fwd_wave_trips": f_issued // 128, "bwd_wave_trips": b_issued // 128,
           "lane_model": "tools/lane_model.py: the kernels' list / item scheduling replayed in numpy; useful = pixel "
                         "evaluations inside a gaussian's alpha >= 1/255 box, issued = 64 lanes per wave-trip"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
