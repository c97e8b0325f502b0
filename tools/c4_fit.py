"""A fit at BASELINE config 4's size (2040x1356) for kernel traces (development aid).
usage: c4_fit.py [iterations] [kind: covariance (adaptive 5 000 -> 50 000) | cholesky (50 000 fixed, Adan)]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianimage_plus_amd.launch import synthetic_image  # noqa: E402
from gaussianimage_plus_amd.trainer import NativeFitter  # noqa: E402

dev = torch.device("cuda:0")
h, w = 1356, 2040
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
kind = sys.argv[2] if len(sys.argv) > 2 else "covariance"
gt = synthetic_image(h, w, 3).to(dev)
if kind == "covariance":
    fit = NativeFitter(gt, 5000, kind="covariance", lr=0.018, eps=1e-15, max_points=50000, track_best=True,
                       device_resident=True)
else:
    fit = NativeFitter(gt, 50000, kind="cholesky", lr=1e-3, eps=1e-15, track_best=True, optimizer="adan")
torch.cuda.synchronize()
t0 = time.time()
if kind == "covariance":
    fit.fit(iters, prune_iter=100, grow_iter=iters // 10)
    fit.sync_population()
else:
    fit.train(iters)
torch.cuda.synchronize()
dt = time.time() - t0
cursor = int(fit.ws[:64].view(torch.int32)[8].item())
fit.check_status()
print(f"row pool: cursor {cursor} of {fit.tx * fit.ty * 256} rows")
print(f"2040x1356 {kind}: {iters} iterations in {dt:.2f} s = {dt / iters * 1e6:.1f} us per iteration; {fit.n} gaussians, "
      f"PSNR {fit.psnr():.2f}")
