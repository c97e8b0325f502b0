"""Development aid: how many gaussians of a Kodak fit lie on more than 32 tiles (the update kernel sums those with a
whole wave each, one after the other: gi2d_fast_internal.h::reduce_one) along the first iterations of the schedule.
usage: big_gaussian_stats.py [image index] [iterations ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gaussianimage_plus_amd.trainer import NativeFitter  # noqa: E402

a = sys.argv[1:]
idx = int(a[0]) if a else 0
stops = [int(x) for x in a[1:]] or [100, 1000, 3000, 5000]
names, pics = bench.load_kodak(idx + 1)
gt = pics[idx].to("cuda:0")
fit = NativeFitter(gt, 5000, kind="covariance", lr=0.018, eps=1e-15, track_best=True)
done = 0
for s in stops:
    fit.train(s - done)
    done = s
    fit.prune_non_definite()
    torch.cuda.synchronize()
    nth = fit.nth[:fit.n].cpu().numpy()
    big = nth > 32
    per_wave = np.add.reduceat(big.astype(np.int64), np.arange(0, len(nth), 64))
    print(f"{names[idx]} after {s:6d} iterations: n = {fit.n}, tiles per gaussian mean {nth.mean():.1f} p50 {np.median(nth):.0f} "
          f"p90 {np.percentile(nth, 90):.0f} max {nth.max()}; on > 32 tiles: {big.mean() * 100:.1f} %, per wave of 64 mean "
          f"{per_wave.mean():.1f} max {per_wave.max()}; > 64 tiles {np.mean(nth > 64) * 100:.1f} %, > 128 {np.mean(nth > 128) * 100:.1f} %")
