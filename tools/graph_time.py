"""Development aid: eager vs hipGraph-replay step time of the fused HotPath."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import synth_cholesky, synth_gt
from gaussianimage_plus_amd.hotpath import HotPath

dev = "cuda:0"
for n in [int(a) for a in (sys.argv[1:] or ["10000", "50000"])]:
    h, w = 512, 768
    hp = HotPath(n, h, w, device=dev)
    hp.set_inputs(*synth_cholesky(n, h, w, 3047))
    out = hp.forward()
    gt = torch.from_numpy(synth_gt(h, w, 1)).to(dev)
    hp.set_target(gt)
    for _ in range(20):
        hp.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        hp.step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 500
    hp.capture_graph()
    for _ in range(20):
        hp.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        hp.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 500
    hp.check_status()
    print(f"N={n} M={hp.num_intersects()} eager {eager*1e6:.1f} us/step ({1/eager:.0f} it/s)  graph {graph*1e6:.1f} us/step ({1/graph:.0f} it/s)")
