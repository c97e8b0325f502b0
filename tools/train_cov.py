"""Development aid: a short covariance-model training run (for rocprofv3 kernel stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gaussianimage_plus_amd.launch import synthetic_image
from gaussianimage_plus_amd.trainer import NativeFitter
gt = synthetic_image(512, 768, 100).cuda()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
f = NativeFitter(gt, n, kind="covariance", lr=0.018, eps=1e-15)
f.train(int(sys.argv[2]) if len(sys.argv) > 2 else 3000)
torch.cuda.synchronize()
nth = f.nth[:f.n]
print("M", int(nth.sum()), "max tiles", int(nth.max()), "gaussians on >16 tiles", int((nth > 16).sum()), ">32", int((nth > 32).sum()))
