"""train_quantize.py's schedule on Kodak pictures (development aid; run under rocprofv3 for the per-kernel breakdown).
usage: kodak_quant_fit.py [images] [iterations] [warmup_iter] [model: covariance | scale_rot] [0 = a stream per image / G batches]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gaussianimage_plus_amd import launch  # noqa: E402

a = sys.argv[1:]
count, iters = (int(a[0]) if a else 6), (int(a[1]) if len(a) > 1 else 12000)
warm = int(a[2]) if len(a) > 2 else 6000
model = a[3] if len(a) > 3 else "covariance"
groups = int(a[4]) if len(a) > 4 else 3
dev = torch.device("cuda:0")
names, pics = bench.load_kodak(count)
if model == "covariance":
    kw = dict(lr=0.018, kind="covariance", max_points=50000, prune_iter=100, grow_iter=max(warm // 6, 1), eps=1e-15)
    n0 = 5000
else:
    kw = dict(lr=1e-3, kind="scale_rot", eps=1e-15, optimizer="adam")  # training_setup(quantize=True) rebuilds Adam
    n0 = 30000
t0 = time.time()
rows = launch.fit_images_native([p.to(dev) for p in pics], n0, iters, seed=3047, eval_renders=1, quantize=True,
                                warmup_iter=warm, threaded=True,
                                batched=(False if groups == 0 else (True if groups == 1 else groups)), **kw)
torch.cuda.synchronize()
dt = time.time() - t0
print(f"[groups {groups}] {model}: {count} images x {iters} iterations ({warm} plain) in {dt:.2f} s = {count / dt:.3f} images/s; "
      f"{dt / iters / count * 1e6:.2f} us per image-iteration; mean PSNR {sum(r['psnr'] for r in rows) / count:.2f}, decoded "
      f"{sum(r['psnr_decoded'] for r in rows) / count:.2f}, bpp {sum(r['bpp'] for r in rows) / count:.3f}, "
      f"gaussians {sum(r['num_gaussians'] for r in rows) / count:.0f}")
