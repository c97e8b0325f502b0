"""Development aid: the INTEGRATION.md section 5 snippets, run end to end on a small image."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianimage_plus_amd.launch import synthetic_image
from gaussianimage_plus_amd.trainer import NativeFitter
gt = synthetic_image(128, 192, 3).cuda()
fit = NativeFitter(gt, num_points=1500, kind="covariance", lr=0.018, eps=1e-15, max_points=2500, track_best=True)
for _ in fit.fit_quantize_schedule(600, warmup_iter=400, bits=(12, 10, 6), prune_iter=100, grow_iter=100):
    pass
fit.check_status(); print("best", fit.load_best())
enc = fit.compress_wo_ec(); img = fit.decompress_wo_ec(enc); print(fit.analysis_wo_ec(enc, entropy_estimate=True))
import gaussianimage_plus_amd.quantize as q
sys.modules["quantize"] = q
from quantize import *
print(UniformQuantizer, HybirdQuant)

# INTEGRATION.md section 4 "Several images per launch"
from gaussianimage_plus_amd.trainer import BatchFitter
images_hwc_cuda = [synthetic_image(128, 192, 5).cuda(), synthetic_image(192, 128, 6).cuda(), synthetic_image(96, 96, 7).cuda()]
fits = [NativeFitter(g, 800, kind="covariance", lr=0.018, eps=1e-15, max_points=2000, track_best=True,
                     device_resident=True) for g in images_hwc_cuda]
BatchFitter(fits).fit(500, prune_iter=100, grow_iter=100)
psnrs = [f.load_best() for f in fits]
for f in fits:
    f.check_status()
print("batched fit: best PSNR per image", [round(p, 2) for p in psnrs], "gaussians", [f.n for f in fits])
assert all(p > 20 for p in psnrs)
