#!/usr/bin/env python3
"""Rasterizer vectors produced by the REFERENCE'S OWN CPU rasterizer (run in the DEV container only; writes
refras_vectors.npz -- arrays only, no reference file travels).

The reference holds exactly one CPU statement of the per-pair arithmetic of its tile rasterizer:
gsplat/gsplat/_torch_impl.py:354-421 `rasterize_forward` (delta = centre - (j, i), sigma = 0.5 (a dx^2 + c dy^2) +
b dx dy, `sigma < 0` and `alpha < 1/255` skipped, alpha = min(0.999, opacity exp(-sigma))).  It is the ALPHA-BLENDING
rasterizer (vis = alpha T, T <- T (1 - alpha)), but for a tile list with ONE contributing gaussian T is 1 when that
gaussian is met and vis = alpha: with background 0 the call's image is exactly one gaussian's term of the sum
rasterizer (forward.cu:636-660), as long as opacity exp(-sigma) <= 0.999 so that neither clamp binds.  The sum
rasterizer's image is the sum of those terms; the function is torch, so autograd through the same call is the
reference's own derivative of that term = what backward.cu:1258-1300 must produce for the gaussian.

So, for every gaussian g of a seeded scene, this script
  * builds the lists with the reference's get_tile_bbox (:236), map_gaussian_to_intersects (:297), torch.sort,
    get_tile_bin_edges (:328) for the three-entry scene {g, s1, s2}: s1, s2 are two opacity-0 gaussians covering every
    tile.  They contribute nothing (alpha = 0 < 1/255 -> `continue` before T is touched) and exist only because the
    helpers need them: `rasterize_forward` reads the loop variable `idx` after the loop (unbound when the first
    pixel's tile is empty) and `get_tile_bin_edges` never closes a last tile that holds a single entry (:339-341);
  * calls `rasterize_forward` on it -- once on float32 tensors (the reference as it runs) and once on the same values
    as float64 tensors (the same code, rounding removed) -- and back-propagates <image, v_out> to xys / conics /
    colours / opacities of row 0;
  * sums the per-gaussian images in float64.

The scene: ragged 40x56 image (3 x 4 tiles, partial last row/column), 46 gaussians with opacity <= 0.999, two of
them with NON positive definite conics (the `sigma < 0` branch), centres inside and just outside the image, and no
pair closer than 1e-5 to the 1/255 cut-off or to sigma = 0 (asserted; the float32 and float64 runs land the same
pairs), with dozens of pairs within 1e-4 of it.

Stored next to the reference's outputs: the sum of absolute terms of every output (this repo's own dense formula --
a yardstick for the relative tolerance only, never an expected value).
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_ref_vectors import import_torch_impl  # noqa: E402

# REFRAS_LARGE=1: the second, larger scene (refras_vectors_large.npz): 72 x 104 pixels = 5 x 7 tiles with a partial
# last row and column, 160 gaussians (a quarter of an hour of the reference's per-pixel Python loop)
LARGE = os.environ.get("REFRAS_LARGE") == "1"
H, W = (72, 104) if LARGE else (40, 56)
N_REAL = 160 if LARGE else 46
SEED = int(os.environ.get("REFRAS_SEED", "31" if LARGE else "12"))
OUT_NAME = "refras_vectors_large.npz" if LARGE else "refras_vectors.npz"


def scene(seed):
    rng = np.random.default_rng(seed)
    n = N_REAL
    centre = (rng.random((n, 2)) * np.array([W + 4, H + 4]) - 2).astype(np.float32)
    a = rng.normal(size=(n, 2, 2)).astype(np.float32) * rng.uniform(0.7, 2.0, size=(n, 1, 1)).astype(np.float32)
    cov = (a @ a.transpose(0, 2, 1) + 0.5 * np.eye(2, dtype=np.float32)).astype(np.float32)
    colour = rng.uniform(0.05, 1.0, (n, 3)).astype(np.float32)
    opac = rng.uniform(0.15, 0.999, (n, 1)).astype(np.float32)
    opac[:3, 0] = [0.999, 0.9985, 0.2]
    v_out = (rng.normal(size=(H, W, 3)) / (H * W)).astype(np.float32)
    return centre, cov, colour, opac, v_out


def main():
    ti = import_torch_impl()
    tx, ty = (W + 15) // 16, (H + 15) // 16
    tb = (tx, ty, 1)
    centre, cov, colour, opac, v_out = scene(SEED)
    n = N_REAL
    conic, radius, _ = ti.compute_cov2d_bounds(torch.from_numpy(cov))
    conic, radius = conic.numpy().copy(), radius.numpy().copy()
    # two non positive definite conics (det < 0): sigma changes sign inside their box
    conic[5] = [0.30, 0.50, 0.20]
    radius[5] = 6.0
    centre[5] = [21.3, 17.6]
    conic[17] = [0.05, -0.35, 0.40]
    radius[17] = 7.0
    centre[17] = [40.4, 30.2]
    radii = radius.astype(np.int32)
    if LARGE:
        # 36 000 candidate pairs: no seed leaves them all 1e-5 away from the cut-offs, so the scene is repaired before
        # the long run -- a gaussian with a pair inside the band gets its opacity lowered by a fraction of a per cent
        # (which moves all its alphas), until none is left.  The look uses the dense restatement further down, on the
        # reference's own tile boxes; the reference's run below then asserts the same gaps on what it actually landed.
        tmin, tmax = ti.get_tile_bbox(torch.from_numpy(centre), torch.from_numpy(radii).float(), tb)
        jj, ii = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64), indexing="xy")
        txp, typ = (jj // 16)[None], (ii // 16)[None]
        mn, mx = tmin.numpy().astype(np.float64), tmax.numpy().astype(np.float64)
        in_box = (txp >= mn[:, 0, None, None]) & (txp < mx[:, 0, None, None]) & (typ >= mn[:, 1, None, None]) & \
                 (typ < mx[:, 1, None, None])
        dx = centre[:, 0].astype(np.float64)[:, None, None] - jj[None]
        dy = centre[:, 1].astype(np.float64)[:, None, None] - ii[None]
        c64 = conic.astype(np.float64)
        sg = 0.5 * (c64[:, 0, None, None] * dx * dx + c64[:, 2, None, None] * dy * dy) + c64[:, 1, None, None] * dx * dy
        gs_ = np.abs(sg)[in_box].min()
        assert gs_ > 1.2e-5, "a pixel sits on sigma = 0: set REFRAS_SEED to another seed"
        for it in range(400):
            al = opac.astype(np.float64)[:, 0, None, None] * np.exp(-sg)
            near_cut = (np.abs(al - 1.0 / 255.0) < 2e-5) & in_box & (sg >= 0)
            bad = np.nonzero(near_cut.any((1, 2)))[0]
            if len(bad) == 0:
                break
            opac[bad, 0] = (opac[bad, 0].astype(np.float64) * (1.0 - 0.002 - 0.001 * (it % 7))).astype(np.float32)
        else:
            raise AssertionError("could not move every pair off the alpha cut-off")
        print(f"seed {SEED}: scene repaired in {it} rounds; closest pair to sigma = 0 {gs_:.3g}", flush=True)
    sent_xy = np.array([[W / 2, H / 2]] * 2, np.float32)
    sent_conic = np.array([[0.01, 0.0, 0.01]] * 2, np.float32)
    sent_rad = np.array([400, 400], np.int32)
    vo = torch.from_numpy(v_out)

    def one_call(g, dtype):
        xy = torch.from_numpy(np.concatenate([centre[g:g + 1], sent_xy])).to(dtype).requires_grad_(True)
        co = torch.from_numpy(np.concatenate([conic[g:g + 1], sent_conic])).to(dtype).requires_grad_(True)
        cl = torch.from_numpy(np.concatenate([colour[g:g + 1], np.ones((2, 3), np.float32)])).to(dtype).requires_grad_(True)
        op = torch.from_numpy(np.concatenate([opac[g:g + 1], np.zeros((2, 1), np.float32)])).to(dtype).requires_grad_(True)
        rad = torch.from_numpy(np.concatenate([radii[g:g + 1], sent_rad]))
        with torch.no_grad():
            tmin, tmax = ti.get_tile_bbox(xy.detach().float(), rad.float(), tb)
            nth = ((tmax[:, 0] - tmin[:, 0]) * (tmax[:, 1] - tmin[:, 1])).to(torch.int32)
            cum = torch.cumsum(nth, 0).to(torch.int32)
            isect, gids = ti.map_gaussian_to_intersects(3, xy.detach().float(), torch.zeros(3), rad, cum, tb)
            srt, perm = torch.sort(isect, stable=True)
            gs = gids[perm]
            bins = ti.get_tile_bin_edges(int(cum[-1]), srt)
        img, final_T, _ = ti.rasterize_forward(tb, (16, 16, 1), (W, H, 1), gs, bins, xy, co, cl, op,
                                               torch.zeros(3, dtype=dtype))
        (img.to(torch.float64) * vo.double()).sum().backward()
        member = np.zeros(tx * ty, bool)
        for t in range(tx * ty):
            member[t] = bool((gs[int(bins[t, 0]):int(bins[t, 1])] == 0).any())
        return (img.detach().double().numpy(), final_T.detach().numpy(), member, int(nth[0]), xy.grad[0].double().numpy(),
                co.grad[0].double().numpy(), cl.grad[0].double().numpy(), op.grad[0].double().numpy())

    res = {}
    t0 = time.time()
    cache = os.environ.get("REFRAS_CACHE")  # scratch file with the two runs' raw results (development aid)
    if cache and os.path.exists(cache):
        import pickle
        res = pickle.load(open(cache, "rb"))
    for tag, dtype in (() if res else (("f64", torch.float64), ("f32", torch.float32))):
        img = np.zeros((H, W, 3))
        rows = {k: [] for k in ("v_xy", "v_conic", "v_rgb", "v_opacity", "member", "nth", "landed")}
        for g in range(n):
            im, fT, member, nth, vxy, vco, vcl, vop = one_call(g, dtype)
            img += im
            rows["v_xy"].append(vxy), rows["v_conic"].append(vco), rows["v_rgb"].append(vcl), rows["v_opacity"].append(vop)
            rows["member"].append(member), rows["nth"].append(nth)
            rows["landed"].append(fT != 1.0)  # T left 1 <=> the pair was skipped (alpha > 0 where it lands)
            print(f"{tag} gaussian {g}: {int((fT != 1.0).sum())} pairs landed, {time.time() - t0:.0f} s", flush=True)
        res[tag] = (img, {k: np.array(v) for k, v in rows.items()})
    if cache and not os.path.exists(cache):
        import pickle
        pickle.dump(res, open(cache, "wb"))
    img64, r64 = res["f64"]
    img32, r32 = res["f32"]
    assert np.array_equal(r64["landed"], r32["landed"]), "a pair lands in one precision only: pick another seed"
    assert np.array_equal(r64["member"], r32["member"])
    landed = r64["landed"]                      # [N, H, W]
    member = r64["member"].T.copy()             # [T, N] like ras_member of ref_vectors.npz

    # --- diagnostics and tolerance yardsticks (this repo's own dense formula in float64; NOT expected values)
    jj, ii = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64), indexing="xy")
    dx = centre[:, 0].astype(np.float64)[:, None, None] - jj[None]
    dy = centre[:, 1].astype(np.float64)[:, None, None] - ii[None]
    c64 = conic.astype(np.float64)
    sigma = 0.5 * (c64[:, 0, None, None] * dx * dx + c64[:, 2, None, None] * dy * dy) + c64[:, 1, None, None] * dx * dy
    alpha = opac.astype(np.float64)[:, 0, None, None] * np.exp(-sigma)
    tile_of = (ii.astype(int) // 16) * tx + (jj.astype(int) // 16)
    in_list = member.T[:, tile_of]               # [N, H, W]
    lands = in_list & (sigma >= 0) & (alpha >= 1.0 / 255.0)
    assert np.array_equal(lands, landed), "the dense restatement disagrees with the reference on which pairs land"
    assert alpha[lands].max() <= 0.999, "a clamp of the helper binds"
    gap_alpha = np.abs(alpha - 1.0 / 255.0)[in_list & (sigma >= 0)].min()
    gap_sigma = np.abs(sigma)[in_list].min()
    near = int(((np.abs(alpha - 1.0 / 255.0) < 1e-4) & in_list & (sigma >= 0)).sum())
    neg = int((in_list & (sigma < 0) & (alpha >= 1.0 / 255.0)).sum())
    print(f"closest pair to the alpha cut-off {gap_alpha:.3g}, to sigma = 0 {gap_sigma:.3g}; {near} pairs within 1e-4 "
          f"of the cut-off; {neg} pairs skipped by sigma < 0 alone; {int(lands.sum())} landing pairs")
    assert gap_alpha > 1e-5 and gap_sigma > 1e-5 and near >= 10 and neg >= 20
    w_ = np.where(lands, alpha, 0.0)
    va = np.einsum("hwc,nc->nhw", v_out.astype(np.float64), colour.astype(np.float64))
    wgt = w_ * np.abs(va)
    a_, b_, c_ = c64[:, 0, None, None], c64[:, 1, None, None], c64[:, 2, None, None]
    mag_xy = np.stack([(wgt * (np.abs(a_ * dx) + np.abs(b_ * dy))).sum((1, 2)),
                       (wgt * (np.abs(b_ * dx) + np.abs(c_ * dy))).sum((1, 2))], 1)
    mag_conic = np.stack([(0.5 * wgt * dx * dx).sum((1, 2)), (wgt * np.abs(dx * dy)).sum((1, 2)),
                          (0.5 * wgt * dy * dy).sum((1, 2))], 1)
    mag_rgb = np.einsum("nhw,hwc->nc", w_, np.abs(v_out.astype(np.float64)))
    mag_op = (np.where(lands, np.exp(-sigma), 0.0) * np.abs(va)).sum((1, 2))[:, None]
    abs_img = np.einsum("nhw,nc->hwc", w_, np.abs(colour.astype(np.float64)))
    # the two runs of the reference agree with each other far inside the bar the tests use
    for k, mag in (("v_xy", mag_xy), ("v_conic", mag_conic), ("v_rgb", mag_rgb), ("v_opacity", mag_op)):
        worst = float((np.abs(r64[k] - r32[k]) / (mag + 1e-30)).max())
        print(f"float32 run vs float64 run, {k}: worst |diff| / sum|terms| = {worst:.2e}")
        assert worst < 5e-6
    rel_img = np.abs(img64 - img32) / (abs_img + 1e-30)
    print(f"float32 run vs float64 run, out_img: worst |diff| / sum|terms| = {rel_img.max():.2e} at "
          f"{np.unravel_index(rel_img.argmax(), rel_img.shape)}")
    assert rel_img.max() < (2e-5 if LARGE else 5e-6)

    out = dict(refras_hw=np.array([H, W]), refras_xys=centre, refras_conics=conic.astype(np.float32),
               refras_radii=radii, refras_nth=r64["nth"].astype(np.int32), refras_colors=colour, refras_opacity=opac,
               refras_v_out=v_out, refras_member=member,
               refras_out_img=img64, refras_v_xy=r64["v_xy"], refras_v_conic=r64["v_conic"], refras_v_rgb=r64["v_rgb"],
               refras_v_opacity=r64["v_opacity"],
               refras32_out_img=img32, refras32_v_xy=r32["v_xy"], refras32_v_conic=r32["v_conic"],
               refras32_v_rgb=r32["v_rgb"], refras32_v_opacity=r32["v_opacity"],
               refras_landed=np.packbits(landed), refras_pairs_landing=np.array(int(lands.sum())),
               refras_abs_img=abs_img, refras_mag_xy=mag_xy, refras_mag_conic=mag_conic, refras_mag_rgb=mag_rgb,
               refras_mag_opacity=mag_op, refras_gap_alpha=np.array(gap_alpha), refras_gap_sigma=np.array(gap_sigma))
    np.savez_compressed(os.path.join(HERE, OUT_NAME), **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == "__main__":
    main()
