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
lib = _lib.load()
lib.gi2d_debug_set_trace.argtypes = [ctypes.c_void_p]
TRAIN = os.environ.get("TRACE_TRAIN", "0") == "1"  # the fitter's training iterations (tile order renewed every 16th step)
if TRAIN:
    import bench

    class _Fit:  # the two things the report below asks of `hp`
        def __init__(self):
            self.f = bench.make_fitter(torch.from_numpy(synth_gt(h, w, 1)).to(dev), xyz, L, col, n, h, w)
            self.T = self.f.tx * self.f.ty

        def num_intersects(self):
            return int(self.f.nth[:n].sum().item())
    hp = _Fit()
    trace = torch.zeros(hp.T, 16, dtype=torch.int64, device=dev)
    assert lib.gi2d_debug_set_trace(trace.data_ptr()) == 0
    hp.f.train(int(os.environ.get("TRACE_STEPS", "36")))
else:
    hp = HotPath(n, h, w, device=dev)
    hp.set_inputs(xyz, L, col, op)
    hp.set_target(torch.from_numpy(synth_gt(h, w, 1)).to(dev))
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


# placement: which workgroups shared a CU, and does the CU's total tile population explain who finishes last?
hw, xcc, pop = raw[:, 14], raw[:, 15] & 0xf, raw[:, 4].astype(np.float64)
cu_key = (xcc << 16) | (hw & 0xff00)  # XCC, SE/SH/CU fields of HW_ID
end = t[:, -1]
keys, inv = np.unique(cu_key, return_inverse=True)
per_cu_n = np.bincount(inv)
per_cu_pop = np.bincount(inv, weights=pop)
per_cu_end = np.array([end[inv == i].max() for i in range(len(keys))])
per_cu_mean_end = np.array([end[inv == i].mean() for i in range(len(keys))])
print(f"distinct CU keys: {len(keys)}, workgroups per CU min/max {per_cu_n.min()}/{per_cu_n.max()}")
print(f"tile population: mean {pop.mean():.1f} max {pop.max()}; per-CU sum mean {per_cu_pop.mean():.0f} min {per_cu_pop.min():.0f} max {per_cu_pop.max():.0f}")
print(f"CU finish time: min {per_cu_end.min():.1f} p50 {np.median(per_cu_end):.1f} max {per_cu_end.max():.1f}")
print(f"corr(CU population sum, CU finish time) = {np.corrcoef(per_cu_pop, per_cu_end)[0, 1]:.2f};  "
      f"corr(tile population, workgroup finish) = {np.corrcoef(pop, end)[0, 1]:.2f}")
order = np.argsort(per_cu_end)
print("slowest CUs: finish", np.round(per_cu_end[order[-6:]], 1), "population", per_cu_pop[order[-6:]], "n", per_cu_n[order[-6:]], "xcc", (keys[order[-6:]] >> 16))
print("fastest CUs: finish", np.round(per_cu_end[order[:6]], 1), "population", per_cu_pop[order[:6]], "n", per_cu_n[order[:6]], "xcc", (keys[order[:6]] >> 16))
for x in range(8):
    m = (keys >> 16) == x
    if m.any():
        print(f"  xcc {x}: CUs {int(m.sum())}, finish mean {per_cu_end[m].mean():.1f} max {per_cu_end[m].max():.1f}, start mean {t[:, 0][np.isin(inv, np.nonzero(m)[0])].mean():.2f}")
# which SIMD did wave 0 of each workgroup land on (HW_ID bits 5:4)?  The waves of a workgroup go to consecutive SIMDs,
# so a phase that loads its waves unevenly (backward items fill waves 0, 1, 2 in that order) loads the SIMDs unevenly
simd0 = (hw >> 4) & 3
print("wave 0 on SIMD 0/1/2/3:", np.bincount(simd0, minlength=4))
per_cu_simd = np.array([np.bincount(simd0[inv == i], minlength=4) for i in range(len(keys))])
print("per CU, workgroups whose wave 0 sits on the same SIMD: max", per_cu_simd.max(1).mean().round(2), "(mean over CUs)")
# workgroup index -> CU: is the dealing regular?  (trace rows are indexed by TILE; column 1 holds the workgroup's slot)
slot = raw[:, 1].astype(np.int64)
if len(np.unique(slot)) == hp.T:
    by_slot = np.empty(hp.T, np.int64)
    by_slot[slot] = np.arange(hp.T)
    cu_of_slot = inv[by_slot]
    x0 = np.arange(0, hp.T, 8)  # the workgroups of XCC (slot % 8 == 0)
    print("XCC of slots 0..15:", xcc[by_slot[:16]])
    seq = cu_of_slot[x0]
    print("CU index (within this trace's numbering) of slots 0, 8, 16, ...:", seq[:72])
    for period in (32, 64, 96, 128, 192):
        same = np.mean([cu_of_slot[s] == cu_of_slot[s + 8 * period] for s in range(hp.T - 8 * period)])
        print(f"  slots s and s + 8 x {period} on the same CU: {same:.3f}")
    same256 = np.mean([cu_of_slot[s] == cu_of_slot[s + 256] for s in range(hp.T - 256)])
    print(f"  slots s and s + 256 on the same CU: {same256:.3f}")
    np.save(os.path.join(ROOT, "gpurun_out", f"trace_by_slot_{int(TRAIN)}.npy"), raw[by_slot])  # every column, for offline looks
    np.save(os.path.join(ROOT, "gpurun_out", f"trace_tile_of_slot_{int(TRAIN)}.npy"), by_slot)
