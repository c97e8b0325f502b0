"""The images/s leg of bench.py alone: Kodak images fitted as one batch (development aid; run under rocprofv3 for the
per-kernel breakdown).  usage: kodak_fit.py [images] [iterations] [0 = one stream per image / 1 = one batch / G = G batches on G streams]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gaussianimage_plus_amd import launch  # noqa: E402

a = sys.argv[1:]
count, iters, batched = (int(a[0]) if a else 24), (int(a[1]) if len(a) > 1 else 10000), (int(a[2]) if len(a) > 2 else 1)
dev = torch.device("cuda:0")
names, pics = bench.load_kodak(count)
grow = 5000 if iters >= 20000 else max(iters // 10, 1)
kw = dict(lr=0.018, seed=3047, kind="covariance", max_points=50000, prune_iter=100, grow_iter=grow, eps=1e-15,
          optimizer="adam", eval_renders=1)
t0 = time.time()
rows = launch.fit_images_native([p.to(dev) for p in pics], 5000, iters, batched=(True if batched == 1 else batched), threaded=True, **kw)
torch.cuda.synchronize()
dt = time.time() - t0
print(f"[mode {batched}] {count} images x {iters} iterations in {dt:.2f} s = {count / dt:.3f} images/s; "
      f"{dt / iters / count * 1e6:.2f} us per image-iteration; mean PSNR {sum(r['psnr'] for r in rows) / count:.2f}, "
      f"mean gaussians {sum(r['num_gaussians'] for r in rows) / count:.0f}")
