"""Development aid: full training iterations/s of the native fitter (gi2d_train_step)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gaussianimage_plus_amd.launch import synthetic_image
from gaussianimage_plus_amd.trainer import NativeFitter

dev = "cuda:0"
h, w = 512, 768
gt = synthetic_image(h, w, 7).to(dev)
for n in [int(a) for a in (sys.argv[1:] or ["2500", "10000", "50000"])]:
    fit = NativeFitter(gt, n, lr=1e-3 if n > 20000 else 5e-3)
    fit.train(50)
    torch.cuda.synchronize()
    p0 = fit.psnr()
    t0 = time.perf_counter()
    fit.train(2000)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fit.check_status()
    print(f"N={n}: {2000 / dt:.0f} train it/s ({dt / 2000 * 1e6:.1f} us/it), PSNR {p0:.2f} -> {fit.psnr():.2f}, M={int(fit.nth.sum())}")
