"""CPU: tests/helpers_adan.py::AdanRef against outputs of the reference's own Adan (fixture generated in the dev
container by tests/golden/make_adan_golden.py from /root/reference/optimizer.py)."""
import os

import numpy as np
import torch

from helpers_adan import AdanRef

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "adan_reference.npz")


def test_adan_statement_matches_reference_fixture():
    z = np.load(GOLD)
    p = torch.nn.Parameter(torch.from_numpy(z["p0"]).clone())
    opt = AdanRef([p], lr=float(z["lr"]), betas=tuple(float(b) for b in z["betas"]), eps=float(z["eps"]))
    for t, g in enumerate(z["grads"]):
        p.grad = torch.from_numpy(g).clone()
        opt.step()
        want = torch.from_numpy(z["traj"][t])
        assert torch.allclose(p.detach(), want, rtol=2e-6, atol=1e-9), t
    s = opt.state[0]
    assert torch.allclose(s["m"], torch.from_numpy(z["exp_avg"]), rtol=2e-6, atol=1e-12)
    assert torch.allclose(s["n"], torch.from_numpy(z["exp_avg_sq"]), rtol=2e-6, atol=1e-12)
    assert torch.allclose(s["d"], torch.from_numpy(z["exp_avg_diff"]), rtol=2e-6, atol=1e-12)
    assert torch.equal(-s["prev"], torch.from_numpy(z["neg_pre_grad"]))
    # the steps are not tiny: the fixture would expose a wrong bias correction or a missing difference term
    assert float((torch.from_numpy(z["traj"][-1]) - torch.from_numpy(z["p0"])).abs().max()) > 3e-3
