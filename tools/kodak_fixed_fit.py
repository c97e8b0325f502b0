"""Kodak pictures with a FIXED population (development aid): the Cholesky model with Adan (train.py's choice for it) or the
rotation-scale model, no prune / grow.  usage: kodak_fixed_fit.py [images] [iterations] [model] [points] [groups]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gaussianimage_plus_amd import launch  # noqa: E402

a = sys.argv[1:]
count, iters = (int(a[0]) if a else 24), (int(a[1]) if len(a) > 1 else 10000)
model, n = (a[2] if len(a) > 2 else "cholesky"), (int(a[3]) if len(a) > 3 else 50000)
groups = int(a[4]) if len(a) > 4 else 3
dev = torch.device("cuda:0")
names, pics = bench.load_kodak(count)
t0 = time.time()
rows = launch.fit_images_native([p.to(dev) for p in pics], n, iters, lr=1e-3, seed=3047, kind=model, eps=1e-15,
                                optimizer="adan", eval_renders=1, threaded=True,
                                batched=(False if groups == 0 else (True if groups == 1 else groups)))
torch.cuda.synchronize()
dt = time.time() - t0
print(f"[groups {groups}] {model} N={n}: {count} images x {iters} iterations in {dt:.2f} s = {count / dt:.3f} images/s; "
      f"{dt / iters / count * 1e6:.2f} us per image-iteration; mean PSNR {sum(r['psnr'] for r in rows) / count:.2f}")
