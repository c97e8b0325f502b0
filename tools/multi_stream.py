"""Development aid: aggregate step rate of K independent images fitted concurrently on K HIP streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import synth_cholesky, synth_gt
from gaussianimage_plus_amd.hotpath import HotPath

dev = "cuda:0"
n, h, w = int(sys.argv[1]) if len(sys.argv) > 1 else 50000, 512, 768
for K in (1, 2, 3, 4, 6):
    hps, streams = [], []
    for i in range(K):
        hp = HotPath(n, h, w, device=dev)
        hp.set_inputs(*synth_cholesky(n, h, w, 3047 + i))
        out = hp.forward()
        gt = torch.from_numpy(synth_gt(h, w, 1 + i)).to(dev)
        hp.set_target(gt)
        hps.append(hp)
        streams.append(torch.cuda.Stream(device=dev))
    torch.cuda.synchronize()
    def run(iters):
        for _ in range(iters):
            for hp, st in zip(hps, streams):
                with torch.cuda.stream(st):
                    hp.step()
    run(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(300)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for hp in hps:
        hp.check_status()
    print(f"N={n} K={K}: {K * 300 / dt:.0f} steps/s aggregate ({dt / 300 * 1e6:.1f} us per round of {K})")
