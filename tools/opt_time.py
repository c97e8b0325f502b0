"""Training iteration time by optimizer (Adam / Adan, the reference's choice for the Cholesky and RS models) at the bench
size (development aid; run under rocprofv3 --kernel-trace for the update kernel alone)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianimage_plus_amd.launch import synthetic_image  # noqa: E402
from gaussianimage_plus_amd.trainer import NativeFitter  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 768)
for opt in ("adam", "adan"):
    fit = NativeFitter(synthetic_image(h, w, 100).to(dev), n, kind="cholesky", lr=1e-3, eps=1e-15, track_best=True,
                       optimizer=opt)
    fit.train(50)
    torch.cuda.synchronize()
    t0 = time.time()
    fit.train(400)
    torch.cuda.synchronize()
    print(opt, f"{(time.time() - t0) / 400 * 1e6:.1f} us per iteration")
    del fit
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    time.sleep(0.5)
