import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianimage_plus_amd.launch import synthetic_image
from gaussianimage_plus_amd.trainer import NativeFitter
dev = torch.device("cuda:0")
h, w = 1356, 2040
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
gt = synthetic_image(h, w, 3).to(dev)
fit = NativeFitter(gt, 5000, kind="covariance", lr=0.018, eps=1e-15, max_points=50000, track_best=True, device_resident=True)
torch.cuda.synchronize(); t0 = time.time()
fit.fit(iters, prune_iter=100, grow_iter=iters // 10)
fit.sync_population(); torch.cuda.synchronize()
dt = time.time() - t0
print(f"2040x1356: {iters} iterations in {dt:.2f} s = {dt / iters * 1e6:.1f} us per iteration; {fit.n} gaussians, PSNR {fit.psnr():.2f}")
