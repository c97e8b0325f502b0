"""Development aid: the drop-in autograd loop (launch.fit_image) eager against replayed from a captured HIP graph.
usage: dropin_graph_time.py [iterations]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianimage_plus_amd import launch  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda:0")
gt = launch.synthetic_image(512, 768, 3).to(dev)
launch.fit_image(gt, 50000, 300, eval_renders=1)
for g in (False, True, True, False):
    r = launch.fit_image(gt, 50000, iters, eval_renders=1, graph=g)
    print("graph" if g else "eager", f"{r['train_s'] / iters * 1e6:.1f} us per iteration, psnr {r['psnr']:.4f}", flush=True)
