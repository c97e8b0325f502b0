"""Development aid: which PyTorch operator every kernel of one drop-in training iteration (launch.fit_image's loop body) belongs
to -- torch.profiler on three eager iterations; prints the fill / copy kernels with the operator that launched them."""
import math
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianimage_plus_amd import launch  # noqa: E402
from gaussianimage_plus_amd.gsplat.project_gaussians_2d import project_gaussians_2d  # noqa: E402
from gaussianimage_plus_amd.gsplat.rasterize_sum_plus import rasterize_gaussians_plus  # noqa: E402

dev = torch.device("cuda:0")
h, w, n = 512, 768, 50000
gt = launch.synthetic_image(h, w, 3).to(dev)
tb = ((w + 15) // 16, (h + 15) // 16, 1)
g = torch.Generator(device="cpu").manual_seed(1)
xyz = torch.atanh(2 * (torch.rand(n, 2, generator=g) - 0.5)).to(dev).requires_grad_(True)
chol = torch.rand(n, 3, generator=g).to(dev).requires_grad_(True)
feat = torch.zeros(n, 3, device=dev, requires_grad=True)
opacity = torch.ones(n, 1, device=dev)
bound = torch.tensor([min(h * w / (9 * math.pi * n), 300), 0.0, min(h * w / (9 * math.pi * n), 300)], device=dev).view(1, 3)
bg = torch.ones(3, device=dev)
opt = torch.optim.Adam([xyz, chol, feat], lr=torch.tensor(1e-3, device=dev), capturable=True, fused=True)


def iteration():
    xys, depths, radii, conics, nth = project_gaussians_2d(torch.tanh(xyz), chol + bound, h, w, tb)
    img = rasterize_gaussians_plus(xys, depths, radii, conics, nth, feat, opacity, h, w, 16, 16, background=bg)
    loss = torch.nn.functional.mse_loss(torch.clamp(img, 0, 1), gt)
    loss.backward()
    opt.step()


for _ in range(5):
    iteration()
    opt.zero_grad(set_to_none=True)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    iteration()
    torch.cuda.synchronize()
ev = [e for e in prof.events()]
kern = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
cpu = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU]
print(len(kern), "kernels / copies in one iteration")
for k in kern:
    # innermost CPU op whose time range encloses the launch (correlation via time: the kernel is launched inside it)
    par = [c for c in cpu if c.time_range.start <= k.time_range.start and False]
    print(f"{k.name[:70]:70s} {k.time_range.elapsed_us():8.1f} us")
print(prof.key_averages(group_by_stack_n=0).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
