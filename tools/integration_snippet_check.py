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
