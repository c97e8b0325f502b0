"""Row-pool usage of a Kodak fit (development aid): the bump allocator's cursor (csrc/gi2d_fast_internal.h: PrevBox)
against the pool's size, per image, after the launcher's adaptive schedule.  usage: pool_usage.py [images] [iterations]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gaussianimage_plus_amd.trainer import BatchFitter, NativeFitter  # noqa: E402

a = sys.argv[1:]
count, iters = (int(a[0]) if a else 6), (int(a[1]) if len(a) > 1 else 50000)
dev = torch.device("cuda:0")
names, pics = bench.load_kodak(count)
fits = [NativeFitter(p.to(dev), 5000, kind="covariance", lr=0.018, eps=1e-15, max_points=50000, track_best=True,
                     device_resident=True) for p in pics]
peak = [0] * count
gen = BatchFitter(fits).fit_schedule(iters, prune_iter=100, grow_iter=5000 if iters >= 20000 else max(iters // 10, 1),
                                     max_points=50000, chunk=1000)
for local in gen:
    for i, f in enumerate(fits):
        peak[i] = max(peak[i], int(f.ws[:64].view(torch.int32)[8].item()))
torch.cuda.synchronize()
for i, f in enumerate(fits):
    cap = f.tx * f.ty * 256
    f.check_status()
    print(f"{names[i]}: pool cursor peak {peak[i]} of {cap} rows ({100.0 * peak[i] / cap:.1f} %), {f.n} gaussians")
