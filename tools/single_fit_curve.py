"""Development aid: wall time of ONE Kodak picture's adaptive fit (train.py's schedule) per 5 000 iterations, with the
host's enqueue time beside it -- is a lone fit (the three-image shard of an 8-GPU run is three of them) bound by the GPU
or by the host's launches?   usage: single_fit_curve.py [image] [iterations]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gaussianimage_plus_amd.trainer import NativeFitter  # noqa: E402

a = sys.argv[1:]
image, iters = (int(a[0]) if a else 0), (int(a[1]) if len(a) > 1 else 50000)
dev = torch.device("cuda:0")
gt = bench.load_kodak(image + 1)[1][image].to(dev)
fit = NativeFitter(gt, 5000, kind="covariance", lr=0.018, eps=1e-15, max_points=50000, track_best=True, device_resident=True)
torch.cuda.synchronize()
t_seg = time.perf_counter()
host = 0.0
last = 0
gen = fit.fit_schedule(iters, prune_iter=100, grow_iter=5000)
while True:
    h0 = time.perf_counter()
    try:
        local = next(gen)
    except StopIteration:
        break
    host += time.perf_counter() - h0
    if local // 5000 != last // 5000 or local == iters:
        torch.cuda.synchronize()
        now = time.perf_counter()
        n = local - last
        print(f"iterations {last:6d}..{local:6d}: {1e6 * (now - t_seg) / n:6.2f} us per iteration (host enqueue {1e6 * host / n:5.2f}), "
              f"{fit.sync_population()} gaussians", flush=True)
        t_seg, host, last = time.perf_counter(), 0.0, local
