import sys, time, torch
sys.path.insert(0, "/root/repo")
from gaussianimage_plus_amd.launch import synthetic_image
from gaussianimage_plus_amd.trainer import NativeFitter
dev = "cuda:0"
gt = synthetic_image(512, 768, 1).to(dev)
for n in (5000, 50000):
    fit = NativeFitter(gt, n, kind="covariance", lr=0.018, eps=1e-15, seed=1)
    fit.train(200); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fit.train(100)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"N={n}: host enqueue {1e6*(t1-t0)/2000:.2f} us per iteration, total {1e6*(t2-t0)/2000:.2f} us per iteration")
