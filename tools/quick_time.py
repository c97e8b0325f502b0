"""Per-op timing of the native path at BASELINE sizes (development aid; bench.py is the contract)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import synth_cholesky, synth_gt  # noqa: E402
import gaussianimage_plus_amd.gsplat.cuda as C  # noqa: E402

dev = "cuda:0"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    h, w = 512, 768
    for n in [int(a) for a in (sys.argv[1:] or ["10000", "50000"])]:
        xyz, L, col, op = synth_cholesky(n, h, w, 3047)
        tb = ((w + 15) // 16, (h + 15) // 16, 1)
        T = tb[0] * tb[1]
        xyz_t, L_t, col_t, op_t = t(xyz), t(L), t(col), t(op)
        bg = torch.ones(3, device=dev)
        proj = lambda: C.project_gaussians_2d_forward(n, 3.0, xyz_t, L_t, h, w, tb, 0.01, 1.0, False)
        xys, depths, radii, conics, nth = proj()
        cum, total = C.cumsum_tiles_hit(nth)
        m = int(total.item())
        mp = lambda: C.map_gaussian_to_intersects(n, m, xys, depths, radii, cum, tb, 1.0, False)
        isect, gids = mp()
        srt_f = lambda: C.sort_intersects(isect, gids, T, want_inv_perm=True, want_bins=True, want_keys=False)
        srt = srt_f()
        gs, bins, inv = srt["gaussian_ids_sorted"], srt["tile_bins"], srt["inv_perm"]
        fwd = lambda: C.rasterize_sum_plus_forward(tb, (16, 16, 1), (w, h, 1), gs, bins, xys, conics, col_t, op_t, bg, False)
        out, fT, fidx = fwd()
        gt = t(synth_gt(h, w, 1))
        v_out = (2 * (out.clamp(0, 1) - gt) / (3 * h * w)).contiguous()
        bwd = lambda: C.rasterize_sum_plus_backward(h, w, 16, 16, gs, bins, xys, conics, col_t, op_t, bg, fT, fidx, v_out,
                                                    None, cum_tiles_hit=cum, inv_perm=inv)
        bwd_g = lambda: C.rasterize_sum_plus_backward(h, w, 16, 16, gs, bins, xys, conics, col_t, op_t, bg, fT, fidx, v_out, None)
        v_xy, v_conic, v_col, v_op = bwd()
        pb = lambda: C.project_gaussians_2d_backward(n, xyz_t, L_t, h, w, radii, conics, v_xy, None, v_conic)
        cnt = (bins[:, 1] - bins[:, 0])
        print(f"N={n} M={m} per-tile mean={cnt.float().mean():.1f} max={int(cnt.max())}")
        res = {}
        for name, fn in [("project_fwd", proj), ("cumsum", lambda: C.cumsum_tiles_hit(nth)), ("map", mp),
                         ("sort", srt_f), ("raster_fwd", fwd), ("raster_bwd(plan)", bwd),
                         ("raster_bwd(generic)", bwd_g), ("project_bwd", pb)]:
            res[name] = timeit(fn)
            print(f"  {name:22s} {res[name]:9.1f} us")
        B = 80 * m + 36 * h * w + 36 * n
        pair = res["raster_fwd"] + res["raster_bwd(plan)"]
        print(f"  fwd+bwd pair {pair:.1f} us -> {1e6 / pair:.0f} pairs/s; B={B / 1e6:.1f} MB -> {B / pair / 1e6:.3f} TB/s"
              f" ({B / pair / 1e6 / 8 * 100:.1f}% of 8 TB/s)")
        t0 = time.time()
        for _ in range(20):
            proj(); C.cumsum_tiles_hit(nth); mp(); srt_f(); fwd(); bwd(); pb()
        torch.cuda.synchronize()
        print(f"  full chain wall (incl. python): {(time.time() - t0) / 20 * 1e6:.1f} us/iter")


if __name__ == "__main__":
    main()
