"""A TRAINED scene as the frozen input of cut-off / knock-out builds (development aid).
  trained_scene.py fit [image] [iterations] [path] [grow_iter] [total]   fit one Kodak picture with the adaptive covariance
                                                    schedule of launch.py (5 000 -> 50 000 gaussians) and save the activated
                                                    parameters + the picture; `total`: the schedule's length when the fit
                                                    stops early (10000 5000 50000 = the scene a fit works on for most of its
                                                    time: 6 000 large gaussians)
  trained_scene.py steps [steps] [path]             HotPath.step() on that frozen scene (what static_steps.py does on
                                                    the uniform synthetic one)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

dev = torch.device("cuda:0")
a = sys.argv[1:]
mode = a[0] if a else "fit"

if mode == "fit":
    import bench
    from gaussianimage_plus_amd.trainer import NativeFitter
    image, iters = int(a[1]) if len(a) > 1 else 0, int(a[2]) if len(a) > 2 else 20000
    path = a[3] if len(a) > 3 else "/tmp/trained_scene.pt"
    grow = int(a[4]) if len(a) > 4 else max(iters // 10, 1)
    total = int(a[5]) if len(a) > 5 else iters
    gt = bench.load_kodak(image + 1)[1][image].to(dev)
    fit = NativeFitter(gt, 5000, kind="covariance", lr=0.018, eps=1e-15, max_points=50000, track_best=True,
                       device_resident=True)
    fit.fit(iters, prune_iter=100, grow_iter=grow, total_iterations=total)
    fit.sync_population()
    torch.cuda.synchronize()
    torch.save({"means": fit.xyz.cpu(), "params": (fit.chol + fit.bound).cpu(), "colors": fit.feat.cpu(),
                "opac": fit.opacity.cpu(), "gt": gt.cpu()}, path)
    print(f"kodak image {image}: {fit.n} gaussians after {iters} iterations, PSNR {fit.psnr():.2f} -> {path}")
else:
    from gaussianimage_plus_amd.hotpath import HotPath
    steps = int(a[1]) if len(a) > 1 else 100
    s = torch.load(a[2] if len(a) > 2 else "/tmp/trained_scene.pt")
    n, (h, w) = s["means"].shape[0], s["gt"].shape[:2]
    hp = HotPath(n, h, w, device=dev, kind="covariance")
    hp.set_inputs(s["means"], s["params"], s["colors"], s["opac"])
    hp.set_target(s["gt"].to(dev))
    hp.forward()
    for _ in range(10):
        hp.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        hp.step()
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / steps * 1e6:.2f} us per step, N = {n}, M = {hp.num_intersects()}")
