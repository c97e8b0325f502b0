"""Development aid: cost of the pieces of the adaptive per-image loop (covariance model, N=2500 -> 5000)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gaussianimage_plus_amd.launch import synthetic_image
from gaussianimage_plus_amd.trainer import NativeFitter

gt = synthetic_image(512, 768, 100).cuda()
its = 20000


def run(name, make, go):
    fit = make()
    fit.train(200)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go(fit)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:40s} {dt / its * 1e6:7.1f} us/iter  n={fit.n}")


mk = lambda **kw: (lambda: NativeFitter(gt, 2500, kind="covariance", lr=0.018, eps=1e-15, **kw))
for rep in range(2):
    run("train only", mk(), lambda f: f.train(its))
    run("train + best snapshot", mk(max_points=5000, track_best=True), lambda f: f.train(its))
    run("fit: prune every 100, no growth", mk(max_points=5000, track_best=True),
        lambda f: f.fit(its, prune_iter=100, adaptive_add=False))
    run("fit: prune 100 + grow 5000", mk(max_points=5000, track_best=True),
        lambda f: f.fit(its, prune_iter=100, grow_iter=5000))
    run("train only", mk(), lambda f: f.train(its))

# where does a prune interval go?
fit = mk(max_points=5000, track_best=True)()
fit.train(200)
torch.cuda.synchronize()
ta = tb = tc = 0.0
T0 = time.perf_counter()
for _ in range(100):
    t0 = time.perf_counter()
    fit.train(100)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    fit.prune_non_definite()
    t3 = time.perf_counter()
    ta += t1 - t0; tb += t2 - t1; tc += t3 - t2
print(f"per interval of 100: enqueue {ta*10:.2f} ms, wait for GPU {tb*10:.2f} ms, prune check {tc*10:.2f} ms, total {(time.perf_counter()-T0)*10:.2f} ms")
