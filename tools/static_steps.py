"""HotPath.step() on frozen parameters, N steps: the driver of cut-off / knock-out builds (their gradients are garbage,
which a training loop would feed back into the scene).  usage: static_steps.py [steps] [N] [H] [W]   (development aid)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import synth_cholesky, synth_gt  # noqa: E402
from gaussianimage_plus_amd.hotpath import HotPath  # noqa: E402

a = sys.argv[1:]
steps = int(a[0]) if a else 100
n, h, w = (int(a[1]) if len(a) > 1 else 50000), (int(a[2]) if len(a) > 2 else 512), (int(a[3]) if len(a) > 3 else 768)
dev = torch.device("cuda:0")
hp = HotPath(n, h, w, device=dev)
hp.set_inputs(*synth_cholesky(n, h, w, 3047))
hp.set_target(torch.from_numpy(synth_gt(h, w, 1)).to(dev))
hp.forward()
for _ in range(10):
    hp.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    hp.step()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / steps * 1e6:.2f} us per step, M = {hp.num_intersects()}")
