"""Development aid: per-phase timeline of the single-pass tile kernel (build csrc with EXTRA=-DGI2D_FUSED_TRACE).
Prints, over all tiles of one launch, when each phase boundary is reached relative to the first workgroup's start."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import synth_cholesky, synth_gt  # noqa: E402
from gaussianimage_plus_amd import _lib  # noqa: E402
from gaussianimage_plus_amd.hotpath import HotPath  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 768)
dev = torch.device("cuda:0")
xyz, L, col, op = synth_cholesky(n, h, w, 3047)
hp = HotPath(n, h, w, device=dev)
hp.set_inputs(xyz, L, col, op)
hp.set_target(torch.from_numpy(synth_gt(h, w, 1)).to(dev))
lib = _lib.load()
lib.gi2d_debug_set_trace.argtypes = [ctypes.c_void_p]
trace = torch.zeros(hp.T, 16, dtype=torch.int64, device=dev)
assert lib.gi2d_debug_set_trace(trace.data_ptr()) == 0
for _ in range(20):
    hp.step()
torch.cuda.synchronize()
raw = trace.cpu().numpy()
order_cols = [0, 11, 12, 13, 2, 3, 5, 6, 7, 8, 9, 10]
names = ["start", "hdr+ids", "gathered", "barrier 1", "ranked+staged", "barrier 2", "fwd loop", "pixel out", "bwd items",
         "bwd lane0", "bwd handoff", "bwd done"]
t = raw.astype(np.float64)[:, order_cols] * 0.01  # 100 MHz ticks -> us
t -= t[:, 0].min()
print(f"N={n} M={hp.num_intersects()} tiles={hp.T}   (us since the first workgroup started)")
print(f"{'phase':14s} {'min':>7s} {'p50':>7s} {'p90':>7s} {'max':>7s}   {'dur p50':>8s} {'dur p90':>8s} {'dur max':>8s}")
for i, nm in enumerate(names):
    c = t[:, i]
    d = t[:, i] - t[:, i - 1] if i else np.zeros_like(c)
    print(f"{nm:14s} {c.min():7.2f} {np.percentile(c, 50):7.2f} {np.percentile(c, 90):7.2f} {c.max():7.2f}   "
          f"{np.percentile(d, 50):8.2f} {np.percentile(d, 90):8.2f} {d.max():8.2f}")
late = t[:, 0] > 3.0
print(f"workgroups starting later than 3 us: {int(late.sum())}")

