"""Development aid: quantisation-aware iteration rate (BASELINE config 5: N = 30 000, 768x512).
usage: quant_time.py [N] [iterations] [covariance | scale_rot]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gaussianimage_plus_amd.launch import synthetic_image
from gaussianimage_plus_amd.trainer import NativeFitter

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
kind = sys.argv[3] if len(sys.argv) > 3 else "covariance"
h, w = 512, 768
gt = synthetic_image(h, w, 1).to("cuda:0")
fit = NativeFitter(gt, n, kind=kind, lr=0.018 if kind == "covariance" else 1e-3, eps=1e-15, track_best=True)
fit.train(300)
fit.prune_non_definite()
torch.cuda.synchronize()
t0 = time.perf_counter(); fit.train(iters); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"N={fit.n} plain      : {dt / iters * 1e6:.1f} us/iter  {iters / dt:.0f} it/s  psnr {fit.last_step_psnr():.2f}")
fit.load_best()
fit.enable_quantize(*((12, 10, 6) if kind == "covariance" else (12, 6, 6)))
fit.train(50)
torch.cuda.synchronize()
t0 = time.perf_counter(); fit.train(iters); torch.cuda.synchronize(); dt = time.perf_counter() - t0
fit.check_status()
print(f"N={fit.n} quantised  : {dt / iters * 1e6:.1f} us/iter  {iters / dt:.0f} it/s  psnr {fit.last_step_psnr():.2f}")
print("best", fit.best())
