"""Development aid: tile populations of a TRAINED scene (the adaptive per-image loop of launch.py) against those of the
bench's uniform synthetic scene, and the tile pass time at each -- what the tile pass of a real fit is bound by."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianimage_plus_amd import _lib  # noqa: E402
from gaussianimage_plus_amd.launch import synthetic_image  # noqa: E402
from gaussianimage_plus_amd.trainer import NativeFitter  # noqa: E402

dev = torch.device("cuda:0")
h, w = 512, 768
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
gt = synthetic_image(h, w, 0).to(dev)
fit = NativeFitter(gt, 5000, kind="covariance", lr=0.018, eps=1e-15, max_points=50000, track_best=True,
                   device_resident=True)


def stats(tag):
    torch.cuda.synchronize()
    gp, bp = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.call("gi2d_fast_workspace_views", fit.ws.data_ptr(), fit.ws.numel(), fit.cap, fit.tx, fit.ty,
              ctypes.byref(gp), ctypes.byref(bp))
    base = fit.ws.data_ptr()
    T = fit.tx * fit.ty
    bins = fit.ws[bp.value - base:bp.value - base + 8 * T].view(torch.int32).view(T, 2).cpu().numpy()
    pop = (bins[:, 1] - bins[:, 0]).astype(np.float64)
    t0 = time.perf_counter()
    fit.train(200)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200 * 1e6
    print(f"{tag}: n={fit.n} tiles={T} population mean {pop.mean():.1f} p50 {np.median(pop):.0f} p90 "
          f"{np.percentile(pop, 90):.0f} p99 {np.percentile(pop, 99):.0f} max {pop.max():.0f}; "
          f"{dt:.1f} us/iteration", flush=True)


gen = fit.fit_schedule(iters, prune_iter=100, grow_iter=iters // 10, chunk=iters // 5)
for _ in gen:
    if fit.iteration % (iters // 5) == 0:
        fit.sync_population()
        stats(f"iteration {fit.iteration}")
fit.sync_population()
stats("after the fit")
