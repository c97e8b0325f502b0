"""Development aid: per-iteration divergence of the fused quantised iteration from the torch loop."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_quant_train_gpu as T

n, h, w = 3000, 96, 144
bits = (12, 10, 6)
for iters in (1, 2, 3, 5, 10):
    fit, gt = T._fitter(n, h, w, debug_grads=True)
    fit.train(20)
    fit.enable_quantize(*bits, debug_grads=True)
    lr = fit.current_lr()
    want = T._torch_quant_loop(fit, gt, iters, lr, bits)
    fit.train(iters)
    torch.cuda.synchronize()
    out = [iters]
    for got, ref in ((fit.xyz, want[0]), (fit.chol, want[1]), (fit.feat, want[2])):
        d = (got - ref).abs()
        out += [f"{d.mean().item():.2e}", f"{d.max().item():.2e}"]
    out.append((fit.qparams - want[3]).abs().max().item())
    print(*out)
    if iters == 1:
        print("qparams native", fit.qparams.tolist())
        print("qparams torch ", want[3].tolist())
        print("range", fit.qrange.tolist())
