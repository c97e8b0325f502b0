"""Development aid: one call of k iterations vs k calls of one iteration (must be bitwise equal)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_trainer_gpu as T
for tb in (False, True):
    for k in (1, 2, 3, 5, 10, 40):
        a, _ = T._cov_fitter(800, 64, 96, track_best=tb)
        b, _ = T._cov_fitter(800, 64, 96, track_best=tb)
        for _ in range(k):
            a.train(1)
        b.train(k)
        out = []
        for nm in ("xyz", "chol", "feat", "xys", "conics"):
            x, y = getattr(a, nm)[:800], getattr(b, nm)[:800]
            d = (x - y).abs()
            out.append(f"{nm} {float(d.max()):.2e}/{int((d > 0).sum())}")
        print("track_best", tb, "k", k, *out)
