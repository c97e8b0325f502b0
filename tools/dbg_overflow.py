import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import synth_cholesky
from gaussianimage_plus_amd.hotpath import HotPath
DEV = "cuda:0"
n, h, w = 1400, 32, 48
xyz, L, col, op = synth_cholesky(n, h, w, 4)
xyz = (xyz * 0.12).astype(np.float32)
fused = HotPath(n, h, w, device=DEV, mode="fused")
exact = HotPath(n, h, w, device=DEV, mode="exact")
v = torch.from_numpy(np.random.default_rng(2).normal(size=(h, w, 3)).astype(np.float32) * 1e-3).to(DEV)
for hp in (fused, exact):
    hp.set_inputs(xyz, L, col, op); hp.set_v_out(v)
fused.step()
print("status after fused step", fused.status.tolist())
try:
    fused.check_status()
except RuntimeError as e:
    print("raised", e)
fused.step_safe()
print("status after step_safe", fused.status.tolist(), "mode", fused.mode)
exact.step(); exact.check_status()
for nm in ("out_img", "v_params", "v_xy", "v_rgb", "xys", "radii", "nth"):
    a, b = getattr(fused, nm), getattr(exact, nm)
    print(nm, torch.equal(a, b), float((a.float() - b.float()).abs().max()))
print("M", fused.num_intersects(), exact.num_intersects(), "cap", fused.capacity, exact.capacity)
fused2 = HotPath(n, h, w, device=DEV, mode="exact")
fused2.set_inputs(xyz, L, col, op); fused2.set_v_out(v); fused2.step()
print("fresh exact vs exact", torch.equal(fused2.out_img, exact.out_img), torch.equal(fused2.v_params, exact.v_params))
