"""Development aid: time of the stand-alone project+fill launch (buckets are never consumed here, so only the timing
is meaningful)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import synth_cholesky
from gaussianimage_plus_amd.hotpath import HotPath
for n in (50000, 10000):
    hp = HotPath(n, 512, 768, device="cuda:0")
    hp.set_inputs(*synth_cholesky(n, 512, 768, 3047))
    st = hp._stream()
    for _ in range(20):
        hp._run(hp._f_bin, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        hp._run(hp._f_bin, st)
    e1.record()
    torch.cuda.synchronize()
    print(f"N={n}: project+fill {e0.elapsed_time(e1) / 300 * 1e3:.2f} us per launch (back to back)")
