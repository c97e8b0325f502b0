"""Development aid ("run-to-run sensitivity of a Kodak fit"): train.py's schedule on single Kodak pictures
with a chosen build of the package, optionally with one initial value moved by a rounding-level amount, recording the
PSNR of the best model and the population after every prune check (so that two runs can be compared for the first
iteration at which a prune / grow decision differs).

usage: python tools/kodak_sensitivity.py <package root> <kodimNN,kodimNN,...> [perturbation 0..6] [iterations]
       <package root>: a directory holding gaussianimage_plus_amd/ with its built libgi2d_hip.so -- this repo ("."), or a
       checkout of an earlier round under build/old_r3, build/old_r4 (git archive <commit> | tar -x; make -C csrc)
       perturbation k: 0 none; 1..3 colour row k-1 starts at 2^-24 instead of 0 (an ulp of a colour of 0.5);
       4..6 covariance entry (k-4, 0) one ulp up
prints one JSON line per picture."""
import json
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
root = os.path.abspath(sys.argv[1])
sys.path.insert(0, root)
from gaussianimage_plus_amd.trainer import NativeFitter  # noqa: E402  (the build under test)

names = sys.argv[2].split(",")
perturb = int(sys.argv[3]) if len(sys.argv) > 3 else 0
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 50000
dev = torch.device("cuda:0")
z = np.load(os.path.join(HERE, "tests", "golden", "kodak24.npz"))
for name in names:
    gt = torch.from_numpy(z[name].astype(np.float32) / 255.0).to(dev)
    f = NativeFitter(gt, 5000, kind="covariance", lr=0.018, seed=3047, eps=1e-15, optimizer="adam", max_points=50000,
                     track_best=True, device_resident=True)
    if 1 <= perturb <= 3:
        f._feat[perturb - 1, 0] = 2.0 ** -24
    elif 4 <= perturb <= 6:
        v = f._chol[perturb - 4, 0].item()
        f._chol[perturb - 4, 0] = float(np.nextafter(np.float32(v), np.float32(np.inf)))
    trace = []
    grow = 5000 if iters >= 20000 else max(iters // 10, 1)
    for local in f.fit_schedule(iters, prune_iter=100, grow_iter=grow, adaptive_add=True, max_points=50000):
        trace.append(int(f.sync_population()))  # population after the events of iteration `local`
    f.check_status()
    final_n = int(f.n)
    f.load_best()
    img = f.render()
    mse = torch.nn.functional.mse_loss(img, f.gt).item()
    print(json.dumps({"root": os.path.relpath(root, HERE), "image": name, "perturbation": perturb, "iterations": iters,
                      "psnr": round(10 * math.log10(1.0 / max(mse, 1e-12)), 4), "best_n": int(f.n), "final_n": final_n,
                      "population_after_each_prune_check": trace}), flush=True)
