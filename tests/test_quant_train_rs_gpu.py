"""GPU: the fused quantisation-aware iteration of the ROTATION-SCALE model (BASELINE config 5: gaussianimage_rs +
quantize-aware path) against the same iteration written the way models/gaussianimage_rs.py:443-485 writes it --
UniformQuantizer modules (positions 12 bit, raw scaling 6 bit, SIGNED 6-bit rotation on sigmoid * 2 pi, colours 6 bit;
:131-163) in front of project_gaussians_2d_scale_rot + rasterize_gaussians_sum, torch autograd, torch.optim.Adam for the
gaussians (eps 1e-15) and one Adam per quantiser optimizer.  The quantiser modules themselves are pinned to the
reference's classes by tests/test_quant_gpu.py."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BITS, ROT_BIT = (12, 6, 6), 6


def _fitter(n, h, w, seed=4, **kw):
    from gaussianimage_plus_amd.launch import synthetic_image
    from gaussianimage_plus_amd.trainer import NativeFitter
    gt = synthetic_image(h, w, 7).to(DEV)
    g = torch.Generator().manual_seed(seed)
    sigma = max(1.0, math.sqrt(h * w / n) * 0.6)  # footprints that cover the image a few times over
    init = {"xyz": torch.rand(n, 2, generator=g) * torch.tensor([float(w), float(h)]),
            "chol": torch.cat([torch.rand(n, 2, generator=g) * sigma + 0.5 * sigma, torch.randn(n, 1, generator=g)], 1),
            "feat": torch.rand(n, 3, generator=g) * 0.3}
    return NativeFitter(gt, n, kind="scale_rot", lr=0.005, eps=1e-15, seed=seed, init=init, **kw), gt


def _torch_rs_quant_loop(fit, gt, iters, lr, qlr=1e-3):
    import gaussianimage_plus_amd.gsplat as gs
    from gaussianimage_plus_amd.quantize import UniformQuantizer
    h, w = gt.shape[0], gt.shape[1]
    n = fit.n
    tb = ((w + 15) // 16, (h + 15) // 16, 1)
    xyz = fit.xyz.clone().requires_grad_(True)
    scaling = fit.chol[:, :2].clone().requires_grad_(True)
    rotation = fit.chol[:, 2:3].clone().requires_grad_(True)
    feat = fit.feat.clone().requires_grad_(True)
    opacity = torch.ones(n, 1, device=DEV)
    xyq = UniformQuantizer(signed=False, bits=BITS[0], learned=True, num_channels=2).to(DEV)
    sq = UniformQuantizer(signed=False, bits=BITS[1], learned=True, num_channels=2).to(DEV)
    rq = UniformQuantizer(signed=True, bits=ROT_BIT, learned=True, num_channels=1).to(DEV)
    fq = UniformQuantizer(signed=False, bits=BITS[2], learned=True, num_channels=3).to(DEV)
    opt = torch.optim.Adam([{"params": [xyz], "lr": lr}, {"params": [feat], "lr": lr}, {"params": [scaling], "lr": lr},
                            {"params": [rotation], "lr": lr}], lr=0.0, eps=1e-15)
    oq = [torch.optim.Adam(xyq.parameters(), lr=qlr),
          torch.optim.Adam(list(sq.parameters()) + list(rq.parameters()), lr=qlr, eps=1e-15),
          torch.optim.Adam(fq.parameters(), lr=qlr, eps=1e-15)]
    bg = torch.ones(3, device=DEV)
    qcat = lambda f: torch.cat([f(xyq.scale), f(xyq.beta), f(sq.scale), f(sq.beta), f(rq.scale), f(rq.beta),
                                f(fq.scale), f(fq.beta)])
    first, losses = None, []
    for it in range(iters):
        means, _, _, _ = xyq(xyz)
        sc, _, _, _ = sq(scaling)  # forward_quantize quantises the RAW scaling and uses it as it is (:451-453)
        rot, _, _, _ = rq(torch.sigmoid(rotation) * (2 * math.pi))
        col, _, _, _ = fq(feat)
        xys, depths, radii, conics, nth = gs.project_gaussians_2d_scale_rot(means, sc, rot, h, w, tb)
        sp = torch.zeros(n, 4, device=DEV)
        img, _, _ = gs.rasterize_gaussians_sum(xys, sp, depths, radii, conics, nth, col, opacity, h, w, 16, 16,
                                               background=bg)
        loss = torch.nn.functional.mse_loss(torch.clamp(img, 0, 1), gt)
        loss.backward()
        losses.append(float(loss.detach()))
        if it == 0:
            first = dict(g=torch.cat([xyz.grad, scaling.grad, rotation.grad, feat.grad], 1).clone(),
                         q=qcat(lambda p: p.grad).clone(), q0=qcat(lambda p: p.detach()).clone())
        opt.step()
        opt.zero_grad(set_to_none=True)
        for o in oq:
            o.step()
            o.zero_grad()
    params = torch.cat([xyz, scaling, rotation, feat], 1).detach()
    return params, qcat(lambda p: p.detach()), first, losses


def _native_params(fit):
    return torch.cat([fit.xyz, fit.chol, fit.feat], 1)


def _compare(fit, gt, n, iters=2, grad_tol=1e-5):
    lr = fit.current_lr()
    qp0 = fit.qparams.clone()
    want_p, want_q, first, losses = _torch_rs_quant_loop(fit, gt, iters, lr)
    # the data initialisation of the 16 quantiser values == what the modules derive on their first forward
    assert torch.equal(qp0, first["q0"]), (qp0, first["q0"])
    fit.train(1)
    fit.check_status()
    g_native, g_ref = fit.dbg_grads[:n].clone(), first["g"]
    scale = g_ref.abs().max(dim=0, keepdim=True).values + 1e-20
    rel = (g_native - g_ref).abs() / scale
    # a code that sits on a rounding boundary may flip with the last ulp of sigmoid / sin / cos: count, do not mask
    assert (rel > grad_tol).float().mean().item() < 2e-4, (rel.max().item(), (rel > grad_tol).sum().item())
    assert rel.mean().item() < 2e-6
    q_native, q_ref = fit.dbg_qgrads[:16], first["q"]
    qerr = ((q_native - q_ref).abs() / (q_ref.abs() + 1e-3 * q_ref.abs().max())).max().item()
    print(f"[trajectory] quantiser-value gradients: worst relative error {qerr:.3g}")
    assert qerr < 1e-5, (q_native, q_ref)  # measured: 8e-8 .. 2e-7 (sums closed in double on both sides)
    fit.train(iters - 1)
    fit.check_status()
    torch.cuda.synchronize()
    diff = (_native_params(fit) - want_p).abs()
    print(f"[trajectory] parameters after {iters} iterations: max diff {diff.max().item() / lr:.3g} lr, mean {diff.mean().item() / lr:.3g} lr, "
          f"share beyond 0.1 lr {(diff > 0.1 * lr).float().mean().item():.3g}")
    assert diff.max().item() < 0.1 * lr, diff.max().item()  # measured: 5e-5 lr (N = 3 000), 6e-3 lr (N = 30 000); no code flipped
    assert diff.mean().item() < 1e-5 * lr, diff.mean().item()  # measured: 1e-8 .. 7e-8 lr
    assert (fit.qparams - want_q).abs().max().item() < 2e-5, (fit.qparams, want_q)
    assert not torch.equal(fit.qparams, qp0)
    assert abs(fit.last_step_psnr() - 10 * math.log10(1.0 / losses[-1])) < 0.1


def test_rs_quantised_iteration_matches_torch_loop_small():
    n, h, w = 3000, 96, 144
    fit, gt = _fitter(n, h, w, debug_grads=True)
    fit.train(20)
    fit.enable_quantize(*BITS, rot_bit=ROT_BIT, debug_grads=True)
    _compare(fit, gt, n)


def test_rs_quantised_iteration_at_config5_size():
    """BASELINE config 5 as one workload: gaussianimage_rs + quantize-aware iteration, N = 30 000, 768x512."""
    n, h, w = 30000, 512, 768
    fit, gt = _fitter(n, h, w, debug_grads=True)
    fit.train(20)
    fit.enable_quantize(*BITS, rot_bit=ROT_BIT, debug_grads=True)
    _compare(fit, gt, n)
    # runs are bitwise repeatable (ordered sums, no float atomics) and a stretch equals the same steps one by one
    outs = []
    for split in (False, True):
        f2, _ = _fitter(n, h, w)
        f2.train(20)
        f2.enable_quantize(*BITS, rot_bit=ROT_BIT)
        if split:
            for _ in range(6):
                f2.train(1)
        else:
            f2.train(6)
        f2.check_status()
        outs.append((_native_params(f2).clone(), f2.qparams.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_rs_quantised_render_and_codec():
    """render() is forward_quantize; the codes decode to the trained render; sizes follow the bit depths."""
    import gaussianimage_plus_amd.gsplat as gs
    n, h, w = 2500, 80, 112
    fit, gt = _fitter(n, h, w, track_best=True)
    fit.train(150)
    fit.load_best()
    fit.enable_quantize(*BITS, rot_bit=ROT_BIT)
    img = fit.render()
    xyq, sq, rq, fq = fit.quantizers()
    with torch.no_grad():
        means, _, _, _ = xyq(fit.xyz)
        sc, _, _, _ = sq(fit.chol[:, :2].contiguous())
        rot, _, _, _ = rq((torch.sigmoid(fit.chol[:, 2:3]) * (2 * math.pi)).contiguous())
        col, _, _, _ = fq(fit.feat)
        xys, depths, radii, conics, nth = gs.project_gaussians_2d_scale_rot(means, sc, rot, h, w, (fit.tx, fit.ty, 1))
        ref, _, _ = gs.rasterize_gaussians_sum(xys, torch.zeros(n, 4, device=DEV), depths, radii, conics, nth, col,
                                               fit.opacity, h, w, 16, 16, background=torch.ones(3, device=DEV))
    assert (img - ref.clamp(0, 1)).abs().max().item() < 5e-5
    assert torch.equal(fit.qfeat[:n], col)  # LSQ values are bit-exact
    fit.train(200)
    fit.check_status()
    p_q = fit.load_best()
    enc = fit.compress_wo_ec()
    dec = fit.decompress_wo_ec(enc)
    p_dec = 10 * math.log10(1.0 / torch.nn.functional.mse_loss(dec, gt).item())
    assert abs(p_dec - p_q) < 0.5, (p_dec, p_q)
    for key, lo, hi in (("quant_means", 0, 4095), ("quant_scaling", 0, 63), ("quant_rotation", -32, 31),
                        ("feature_dc_index", 0, 63)):
        c = enc[key]
        assert c.shape[0] == fit.n and torch.equal(c, c.round()) and c.min() >= lo and c.max() <= hi, key
    a = fit.analysis_wo_ec(enc)
    bits = fit.n * (2 * 12 + 2 * 6 + 6 + 3 * 6) + 32 * 2 * (2 + 2 + 1 + 3)
    assert abs(a["bpp"] - bits / (h * w)) < 1e-12
    assert abs(a["cholesky_bpp"] - (a["scaling_bpp"] + a["rotation_bpp"])) < 1e-15
    sd = fit.state_dict()
    assert sd["rotation_quantizer.scale"].shape == (1,) and sd["scaling_quantizer.beta"].shape == (2,)


def test_rs_quantiser_learning_rate_zero_is_the_model_file_as_written():
    """models/gaussianimage_rs.py:473-485 steps only the gaussians' optimizer: Adam with lr 0 on the quantisers."""
    n, h, w = 1500, 64, 96
    fit, gt = _fitter(n, h, w)
    fit.train(10)
    fit.enable_quantize(*BITS, rot_bit=ROT_BIT, lr=0.0)
    q0 = fit.qparams.clone()
    before = _native_params(fit).clone()
    fit.train(5)
    fit.check_status()
    assert torch.equal(fit.qparams, q0) and not torch.equal(_native_params(fit), before)


def test_launcher_rs_quantised_schedule():
    from gaussianimage_plus_amd.launch import fit_images_native, synthetic_image
    gts = [synthetic_image(96, 144, 20 + i).to(DEV) for i in range(2)]
    rows = fit_images_native(gts, 1500, 500, lr=0.005, kind="scale_rot", eps=1e-15, optimizer="adam", quantize=True,
                             warmup_iter=300, bits=BITS, eval_renders=2)
    for r in rows:
        assert r["psnr"] > 18 and abs(r["psnr_decoded"] - r["psnr"]) < 0.6, r
        assert r["num_gaussians"] == 1500 and r["bpp"] > 0
