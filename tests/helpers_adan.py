"""Plain-torch statement of the Adan update the fused update kernel implements (csrc/gi2d_train.hip::adan): one
tensor, weight decay 0, no gradient clipping.  Pinned to the reference's optimizer by tests/golden/adan_reference.npz
(tests/test_adan_cpu.py); the GPU tests compare the kernel with a training loop that uses this class."""
import math

import torch


class AdanRef:
    def __init__(self, params, lr, betas=(0.98, 0.92, 0.99), eps=1e-8):
        self.params, self.lr, self.betas, self.eps = list(params), lr, betas, eps
        self.step_count = 0
        self.state = [dict(m=torch.zeros_like(p), n=torch.zeros_like(p), d=torch.zeros_like(p), prev=None)
                      for p in self.params]

    @torch.no_grad()
    def step(self):
        self.step_count += 1
        b1, b2, b3 = self.betas
        t = self.step_count
        bc1, bc2, bc3s = 1.0 - b1 ** t, 1.0 - b2 ** t, math.sqrt(1.0 - b3 ** t)
        for p, s in zip(self.params, self.state):
            g = p.grad
            diff = torch.zeros_like(g) if (s["prev"] is None or t == 1) else g - s["prev"]
            s["m"].mul_(b1).add_(g, alpha=1 - b1)
            s["d"].mul_(b2).add_(diff, alpha=1 - b2)
            u = diff * b2 + g
            s["n"].mul_(b3).addcmul_(u, u, value=1 - b3)
            denom = s["n"].sqrt() / bc3s + self.eps
            p.addcdiv_(s["m"], denom, value=-self.lr / bc1)
            p.addcdiv_(s["d"], denom, value=-self.lr * b2 / bc2)
            s["prev"] = g.clone()

    def zero_grad(self):
        for p in self.params:
            p.grad = None
