"""Development aid: K images per GPU, round-robin from one host thread vs one host thread per image."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gaussianimage_plus_amd.launch import fit_images_native, synthetic_image

n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
dev = "cuda:0"
imgs = [synthetic_image(512, 768, 100 + i).to(dev) for i in range(8)]
kw = dict(lr=0.018, kind="covariance", max_points=n1, prune_iter=100, grow_iter=1000, eps=1e-15, eval_renders=1)
fit_images_native(imgs[:1], n0, 200, **kw)
for K in (1, 2, 4, 8):
    for threaded in (False, True):
        if K == 1 and threaded:
            continue
        torch.cuda.synchronize()
        t0 = time.time()
        for g0 in range(0, 8, K):
            rows = fit_images_native(imgs[g0:g0 + K], n0, iters, threaded=threaded, **kw)
        dt = time.time() - t0
        print(f"N {n0}->{n1}, {iters} its, K={K} threaded={threaded}: 8 images in {dt:.2f} s "
              f"({8 * iters / dt / 1e3:.1f} k image-iterations/s), psnr {rows[0]['psnr']:.2f}", flush=True)
