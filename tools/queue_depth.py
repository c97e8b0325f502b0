"""Development aid: does the depth of the launch queue change the iteration rate?  (train() in stretches of `call`
iterations, the host waiting every `sync_every` iterations for the work issued `lag` stretches earlier.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gaussianimage_plus_amd.launch import synthetic_image
from gaussianimage_plus_amd.trainer import NativeFitter

gt = synthetic_image(512, 768, 100).cuda()
its = 20000
for kind, n in (("covariance", 2500), ("cholesky", 50000)):
    for call, lag in ((256, None), (100, None), (100, 0), (100, 2), (32, 4), (256, 1)):
        f = NativeFitter(gt, n, kind=kind, lr=0.018 if kind == "covariance" else 1e-3, eps=1e-15)
        f.max_call = call
        f.train(200)
        torch.cuda.synchronize()
        evs = []
        t0 = time.perf_counter()
        done = 0
        while done < its:
            f.train(call)
            done += call
            if lag is not None:
                e = torch.cuda.Event()
                e.record()
                evs.append(e)
                if len(evs) > lag:
                    evs.pop(0).synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{kind} N={n} call={call} lag={lag}: {dt / done * 1e6:.1f} us/iter")
