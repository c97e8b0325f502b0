"""K images as S batched groups on S HIP streams, all driven by ONE host thread (calls of `chunk` iterations dealt to the
groups in turn): does one group's per-gaussian update kernel hide under another group's tile pass?
Usage: batch_groups.py [N] [H] [W] [kind] [K] [S ...]   (development aid)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianimage_plus_amd.launch import synthetic_image  # noqa: E402
from gaussianimage_plus_amd.trainer import BatchFitter, NativeFitter  # noqa: E402

dev = torch.device("cuda:0")


def run(n, h, w, kind, k, s, iters=400, chunk=8):
    fits = [NativeFitter(synthetic_image(h, w, 100 + i).to(dev), n, kind=kind, lr=1e-3, seed=3047 + i, track_best=True)
            for i in range(k)]
    groups = [BatchFitter(fits[g::s]) for g in range(s)]
    sts = [torch.cuda.Stream(device=dev) for _ in groups]
    torch.cuda.synchronize()

    def go(count):
        for _ in range(count // chunk):
            for b, st in zip(groups, sts):
                with torch.cuda.stream(st):
                    b.train(chunk)
        torch.cuda.synchronize()
    go(48)
    best = 1e9
    for _ in range(3):
        t0 = time.time()
        go(iters)
        best = min(best, time.time() - t0)
    print(f"K={k:2d} in {s} group(s) {kind} N={n} {w}x{h}: {best / iters / k * 1e6:7.2f} us per image-iteration, "
          f"{k * iters / best:9.0f} image-iterations/s", flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    n, h, w = int(a[0]) if a else 50000, int(a[1]) if len(a) > 1 else 512, int(a[2]) if len(a) > 2 else 768
    kind = a[3] if len(a) > 3 else "cholesky"
    k = int(a[4]) if len(a) > 4 else 24
    for s in [int(x) for x in a[5:]] or [1, 2, 3, 4]:
        run(n, h, w, kind, k, s)
