#!/usr/bin/env python3
"""Dev-container only: drive the REFERENCE's Adan optimizer (/root/reference/optimizer.py) on CPU over a fixed gradient
sequence and commit inputs + outputs as a fixture (tests/golden/adan_reference.npz).  The fixture pins the plain-torch
statement of the update rule in tests/helpers_adan.py, which in turn is what the fused HIP update kernel is compared
with on the GPU.  Nothing of the reference travels: the fixture is data (parameters, gradients, results)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from optimizer import Adan  # noqa: E402  (the reference's own file, imported, not copied)


def run(foreach, steps=7, n=40, k=3, lr=1e-3, eps=1e-15, seed=11):
    g = torch.Generator().manual_seed(seed)
    p0 = torch.randn(n, k, generator=g)
    grads = torch.randn(steps, n, k, generator=g) * torch.logspace(-3, 0, steps).view(-1, 1, 1)
    p = torch.nn.Parameter(p0.clone())
    opt = Adan([{"params": [p], "lr": lr}], lr=0.0, eps=eps, foreach=foreach)
    traj = []
    for t in range(steps):
        p.grad = grads[t].clone()
        opt.step()
        opt.zero_grad(set_to_none=True)
        traj.append(p.detach().clone())
    st = opt.state[p]
    return p0, grads, torch.stack(traj), st["exp_avg"], st["exp_avg_sq"], st["exp_avg_diff"], st["neg_pre_grad"]


def main():
    a = run(True)
    b = run(False)
    for x, y in zip(a, b):  # the reference's multi-tensor and single-tensor forms agree
        assert torch.allclose(x, y, rtol=1e-6, atol=1e-9)
    p0, grads, traj, m, n2, d, npg = a
    np.savez(os.path.join(HERE, "adan_reference.npz"), p0=p0.numpy(), grads=grads.numpy(), traj=traj.numpy(),
             exp_avg=m.numpy(), exp_avg_sq=n2.numpy(), exp_avg_diff=d.numpy(), neg_pre_grad=npg.numpy(),
             lr=np.float64(1e-3), eps=np.float64(1e-15), betas=np.array([0.98, 0.92, 0.99]))
    print("wrote adan_reference.npz", traj.shape)


if __name__ == "__main__":
    main()
