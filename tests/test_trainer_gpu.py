"""GPU: the fused training iteration (gi2d_train_step) against the same iteration written with the drop-in gsplat
surface + torch autograd + torch.optim.Adam -- i.e. against what models/gaussianimage_cholesky.py:302-317 /
models/gaussianimage_covariance.py:249-259 execute."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _torch_loop(kind, gt, init, bound, iters, lr, eps=1e-8, optimizer="adam"):
    import gaussianimage_plus_amd.gsplat as gs
    h, w = gt.shape[0], gt.shape[1]
    tb = ((w + 15) // 16, (h + 15) // 16, 1)
    xyz = init["xyz"].clone().to(DEV).requires_grad_(True)
    chol = init["chol"].clone().to(DEV).requires_grad_(True)
    feat = init["feat"].clone().to(DEV).requires_grad_(True)
    opacity = torch.ones(xyz.shape[0], 1, device=DEV)
    if optimizer == "adam":
        opt = torch.optim.Adam([xyz, chol, feat], lr=lr, eps=eps)
    else:
        from helpers_adan import AdanRef
        opt = AdanRef([xyz, chol, feat], lr=lr, eps=eps)
    bg = torch.ones(3, device=DEV)
    grads1, losses = None, []
    for it in range(iters):
        if kind == "cholesky":
            xys, depths, radii, conics, nth = gs.project_gaussians_2d(torch.tanh(xyz), chol + bound, h, w, tb)
        elif kind == "scale_rot":  # models/gaussianimage_rs.py:166-172
            scales = torch.abs(chol[:, :2] + bound[:2])
            rot = torch.sigmoid(chol[:, 2:3]) * 2 * math.pi
            xys, depths, radii, conics, nth = gs.project_gaussians_2d_scale_rot(xyz, scales, rot, h, w, tb)
        else:
            xys, depths, radii, conics, nth = gs.project_gaussians_2d_covariance(xyz, chol + bound, h, w, tb)
        img = gs.rasterize_gaussians_plus(xys, depths, radii, conics, nth, feat, opacity, h, w, 16, 16, background=bg)
        loss = torch.nn.functional.mse_loss(torch.clamp(img, 0, 1), gt)
        loss.backward()
        if it == 0:
            grads1 = torch.cat([xyz.grad, chol.grad, feat.grad], 1).clone()
        losses.append(float(loss.detach()))
        opt.step()
        if optimizer == "adam":
            opt.zero_grad(set_to_none=True)
        else:
            opt.zero_grad()
    return xyz.detach(), chol.detach(), feat.detach(), grads1, losses


@pytest.mark.parametrize("kind", ["cholesky", "covariance"])
def test_native_iteration_matches_torch_adam_loop(kind):
    from gaussianimage_plus_amd.launch import synthetic_image
    from gaussianimage_plus_amd.trainer import NativeFitter
    n, h, w, iters, lr = 3000, 96, 144, 12, 1e-2
    gt = synthetic_image(h, w, 5).to(DEV)
    g = torch.Generator().manual_seed(1)
    if kind == "cholesky":
        xyz = torch.atanh(2 * (torch.rand(n, 2, generator=g) - 0.5) * 0.98)
    else:
        xyz = torch.rand(n, 2, generator=g) * torch.tensor([float(w), float(h)])
    init = {"xyz": xyz, "chol": torch.rand(n, 3, generator=g) * torch.tensor([1.0, 0.3, 1.0]),
            "feat": torch.rand(n, 3, generator=g) * 0.3}
    low_pass = min(h * w / (9 * math.pi * n), 300)
    bound = torch.tensor([low_pass, 0.0, low_pass], device=DEV)
    fit = NativeFitter(gt, n, kind=kind, lr=lr, init=init, debug_grads=True)
    fit.train(1)
    fit.check_status()
    g_native = fit.dbg_grads.clone()
    want = _torch_loop(kind, gt, init, bound, iters, lr)
    # first-iteration gradients w.r.t. the raw parameters (through tanh / +bound / projection / rasterizer / L2)
    scale = want[3].abs().max(dim=0, keepdim=True).values + 1e-20
    err = ((g_native - want[3]).abs() / scale).max().item()
    print(f"first-step gradient error, relative to the column maximum: {err:.3e}")
    assert err < 1e-5, f"first-step gradient mismatch {err}"  # measured: 0 .. 4e-6 (sin / cos of the RS model)
    fit.train(iters - 1)
    torch.cuda.synchronize()
    for got, ref, nm in ((fit.xyz, want[0], "xyz"), (fit.chol, want[1], "chol"), (fit.feat, want[2], "feat")):
        d = (got - ref).abs().max().item()
        # 12 Adam steps of size lr: identical trajectories up to fp32 noise amplified by 1/sqrt(v)
        print(f"[trajectory] {kind} {nm}: max drift {d / (lr * iters):.3g}, mean {(got - ref).abs().mean().item() / (lr * iters):.3g} (units of lr * iters)")
        assert d < 2e-3 * lr * iters, f"{nm} drifted by {d}"                     # ten times the measured drift
        assert (got - ref).abs().mean().item() < 3e-6 * lr * iters, nm
    # the loss the native loop reports for its last render agrees with the torch loop's trajectory
    psnr_native = fit.last_step_psnr()
    psnr_torch = 10 * math.log10(1.0 / want[4][-1])
    assert abs(psnr_native - psnr_torch) < 0.05


def test_native_fit_improves_psnr_and_is_reproducible():
    from gaussianimage_plus_amd.launch import synthetic_image
    from gaussianimage_plus_amd.trainer import NativeFitter
    gt = synthetic_image(128, 192, 3).to(DEV)
    runs = []
    for _ in range(2):
        fit = NativeFitter(gt, 4000, kind="cholesky", lr=1e-2, seed=11)
        p0 = fit.psnr()
        fit.train(400)
        fit.check_status()
        runs.append((p0, fit.psnr(), fit.xyz.clone(), fit.feat.clone()))
    assert runs[0][1] > runs[0][0] + 8.0, runs[0][:2]   # fits the smooth target by a wide margin
    assert torch.equal(runs[0][2], runs[1][2]) and torch.equal(runs[0][3], runs[1][3])  # no atomics: bitwise repeatable
    assert abs(fit.current_lr() - 1e-2) < 1e-12


def test_launcher_fit_functions_agree():
    """launch.fit_image (autograd wrappers + torch Adam) and launch.fit_image_native reach the same quality."""
    from gaussianimage_plus_amd.launch import fit_image, fit_image_native, synthetic_image
    gt = synthetic_image(96, 128, 9).to(DEV)
    a = fit_image(gt, 1500, 150, lr=1e-2, seed=5, eval_renders=2)
    b = fit_image_native(gt, 1500, 150, lr=1e-2, seed=5, eval_renders=2)
    assert abs(a["psnr"] - b["psnr"]) < 0.3, (a["psnr"], b["psnr"])
    assert b["psnr"] > 20


def _cov_fitter(n, h, w, seed=2, **kw):
    from gaussianimage_plus_amd.launch import synthetic_image
    from gaussianimage_plus_amd.trainer import NativeFitter
    gt = synthetic_image(h, w, 7).to(DEV)
    return NativeFitter(gt, n, kind="covariance", lr=0.018, eps=1e-15, seed=seed, **kw), gt


def test_best_snapshot_on_device_equals_host_replay():
    """train.py:133-139: whenever a step's render beats the best PSNR so far, the state dict AFTER that step is the
    new best.  The device keeps that snapshot itself; replay the same run step by step on a second fitter and
    take the snapshots on the host."""
    n, h, w, steps = 800, 64, 96, 40
    fit, _ = _cov_fitter(n, h, w, track_best=True)
    ref, _ = _cov_fitter(n, h, w)
    fit.train(steps)
    best_sse, snap, snap_step = float("inf"), None, 0
    for it in range(1, steps + 1):
        ref.train(1)
        sse = float(ref.tile_sse.sum().item())  # not the device's summation order: compare with a margin below
        if sse < best_sse:
            best_sse, snap, snap_step = sse, (ref.xyz.clone(), ref.chol.clone(), ref.feat.clone()), it
    psnr, step, n_best = fit.best()
    assert n_best == n and step == snap_step
    assert torch.equal(fit.best_xyz[:n], snap[0]) and torch.equal(fit.best_chol[:n], snap[1])
    assert torch.equal(fit.best_feat[:n], snap[2])
    assert abs(psnr - 10 * math.log10(3.0 * h * w / best_sse)) < 1e-3
    # lr 0.018 overshoots now and then: the best step is not simply the last one in a longer run, and load_best
    # restores exactly the snapshot
    got = fit.load_best()
    assert abs(got - psnr) < 1e-9 and torch.equal(fit.xyz, snap[0])


def test_prune_non_definite_compacts_parameters_and_moments():
    h, w = 48, 64
    fit, _ = _cov_fitter(600, h, w, max_points=900)
    fit.train(3)
    torch.cuda.synchronize()
    first = fit.prune_non_definite()  # rand(3) + bound is not always definite: the initial draw loses some rows
    n = fit.n
    assert 0 < first < 300 and n == 600 - first
    bad = torch.tensor([5, 17, 18, n // 2, n - 1], device=DEV)
    fit._chol[bad] = torch.tensor([0.2, 5.0, 0.2], device=DEV) - fit._bound[bad]  # cov + bound = (0.2, 5, 0.2): indefinite
    before = [t[:n].clone() for t in fit._rows()]
    keep = torch.ones(n, dtype=torch.bool, device=DEV)
    keep[bad] = False
    assert fit.prune_non_definite() == 5 and fit.n == n - 5
    for t, b in zip(fit._rows(), before):
        assert torch.equal(t[:fit.n], b[keep])
    fit.train(2)  # the shorter model keeps training
    fit.check_status()
    assert fit.prune_non_definite() == 0 and fit.n == n - 5


def test_adaptive_fit_grows_to_the_cap_and_improves():
    from gaussianimage_plus_amd.trainer import select_new_points
    n0, cap, h, w = 400, 2600, 96, 144
    fit, gt = _cov_fitter(n0, h, w, max_points=cap, track_best=True)
    fit.train(20)
    # one growth step by hand: the same selection on a CPU copy of the render
    torch.cuda.synchronize()
    render = fit.out_img.clamp(0, 1).cpu()
    state = fit.rng.get_state()
    added = fit.add_sample_positions(20, 80, 20)
    fit.rng.set_state(state)
    want = select_new_points(render, gt.cpu(), 1000, torch.rand(1000, 3, generator=fit.rng))
    assert added == want["xyz"].shape[0] and fit.n == n0 + added
    assert torch.equal(fit.xyz[n0:].cpu(), want["xyz"]) and torch.equal(fit.chol[n0:].cpu(), want["cov2d"])
    assert float(fit.m_chol[n0:].abs().sum()) == 0.0 and float(fit.feat[n0:].abs().sum()) == 0.0
    low = min(h * w / (9 * math.pi * fit.n), 300)
    assert torch.allclose(fit.bound[n0:], torch.tensor([low, 0.0, low], device=DEV))
    assert torch.allclose(fit.bound[:n0], torch.tensor([min(h * w / (9 * math.pi * n0), 300), 0.0,
                                                        min(h * w / (9 * math.pi * n0), 300)], device=DEV))
    p_before = fit.psnr()
    # the scheduled loop: 80 iterations, growth every 20 (the last one, at 60, releases the whole budget)
    msgs = []
    fit.fit(80, prune_iter=10, grow_iter=20, log=msgs.append)
    fit.check_status()
    assert fit.n <= cap and fit.n > cap - 200, msgs  # all but the non-definite draws
    assert any("added" in m for m in msgs)
    assert fit.psnr() > p_before + 1.0
    psnr, step, n_best = fit.best()
    assert n0 < n_best <= cap  # the snapshot carries its own population (it may predate a later prune)
    assert fit.load_best() == psnr and abs(fit.psnr() - psnr) < 0.3


def test_native_adan_matches_autograd_loop_with_reference_adan():
    """train.py trains the Cholesky model with Adan (lr 1e-3, eps 1e-15; main(): opt_type "adan"): the fused update
    kernel against the drop-in wrappers + autograd + the Adan statement that the reference fixture pins."""
    from gaussianimage_plus_amd.launch import synthetic_image
    from gaussianimage_plus_amd.trainer import NativeFitter
    n, h, w, iters, lr = 3000, 96, 144, 12, 1e-3
    gt = synthetic_image(h, w, 5).to(DEV)
    g = torch.Generator().manual_seed(4)
    init = {"xyz": torch.atanh(2 * (torch.rand(n, 2, generator=g) - 0.5) * 0.98),
            "chol": torch.rand(n, 3, generator=g) * torch.tensor([1.0, 0.3, 1.0]),
            "feat": torch.rand(n, 3, generator=g) * 0.3}
    low_pass = min(h * w / (9 * math.pi * n), 300)
    bound = torch.tensor([low_pass, 0.0, low_pass], device=DEV)
    fit = NativeFitter(gt, n, kind="cholesky", lr=lr, eps=1e-15, init=init, optimizer="adan")
    fit.train(5)
    fit.train(iters - 5)   # two stretches: the previous gradient carries over between calls
    fit.check_status()
    want = _torch_loop("cholesky", gt, init, bound, iters, lr, eps=1e-15, optimizer="adan")
    for got, ref, nm in ((fit.xyz, want[0], "xyz"), (fit.chol, want[1], "chol"), (fit.feat, want[2], "feat")):
        d = (got - ref).abs()
        moved = (ref - init[nm].to(DEV)).abs().max().item()
        assert moved > 2 * lr, nm                      # Adan's first steps are ~lr each: the parameters did move
        print(f"[trajectory] {nm}: max drift {d.max().item() / (lr * iters):.3g}, mean {d.mean().item() / (lr * iters):.3g} (units of lr * iters)")
        assert d.max().item() < 2e-3 * lr * iters, f"{nm} drifted by {d.max().item()}"  # measured: <= 1.5e-4
        assert d.mean().item() < 3e-6 * lr * iters, nm                                      # measured: <= 2e-7
    # one call of 12 iterations == 12 calls of one iteration, bit for bit
    one = NativeFitter(gt, n, kind="cholesky", lr=lr, eps=1e-15, init=init, optimizer="adan")
    for _ in range(iters):
        one.train(1)
    assert torch.equal(one.xyz, fit.xyz) and torch.equal(one.chol, fit.chol) and torch.equal(one.feat, fit.feat)


def test_native_scale_rot_model_matches_autograd_loop():
    """The rotation-scale parameterisation (models/gaussianimage_rs.py: scale = |s + 0.5|, rot = sigmoid(r) 2 pi,
    pixel coordinates) on the native loop against the wrappers + autograd + torch Adam."""
    from gaussianimage_plus_amd.launch import synthetic_image
    from gaussianimage_plus_amd.trainer import NativeFitter
    n, h, w, iters, lr = 3000, 96, 144, 10, 5e-3
    gt = synthetic_image(h, w, 6).to(DEV)
    g = torch.Generator().manual_seed(9)
    init = {"xyz": torch.rand(n, 2, generator=g) * torch.tensor([float(w), float(h)]),
            "chol": torch.cat([torch.rand(n, 2, generator=g) * 3.0 - 1.5, torch.rand(n, 1, generator=g) * 4 - 2], 1),
            "feat": torch.rand(n, 3, generator=g) * 0.3}
    assert (init["chol"][:, :2] + 0.5 < 0).any()  # the |.| branch with a negative argument is exercised
    bound = torch.tensor([0.5, 0.5, 0.0], device=DEV)
    fit = NativeFitter(gt, n, kind="scale_rot", lr=lr, init=init, debug_grads=True)
    fit.train(1)
    fit.check_status()
    g_native = fit.dbg_grads.clone()
    want = _torch_loop("scale_rot", gt, init, bound, iters, lr)
    scale = want[3].abs().max(dim=0, keepdim=True).values + 1e-20
    err = ((g_native - want[3]).abs() / scale).max().item()
    print(f"first-step gradient error, relative to the column maximum: {err:.3e}")
    assert err < 1e-5, f"first-step gradient mismatch {err}"  # measured: 0 .. 4e-6 (sin / cos of the RS model)
    fit.train(iters - 1)
    for got, ref, nm in ((fit.xyz, want[0], "xyz"), (fit.chol, want[1], "chol"), (fit.feat, want[2], "feat")):
        d = (got - ref).abs()
        print(f"[trajectory] {nm}: max drift {d.max().item() / (lr * iters):.3g}, mean {d.mean().item() / (lr * iters):.3g} (units of lr * iters)")
        assert d.max().item() < 2e-3 * lr * iters, f"{nm} drifted by {d.max().item()}"  # measured: <= 1.5e-4
        assert d.mean().item() < 3e-6 * lr * iters, nm                                      # measured: <= 2e-7


def test_checkpoint_round_trip_in_the_reference_format(tmp_path):
    """state_dict keys / checkpoint fields are the reference's (train.py:62-77,173-175); a fitter restored from the
    file renders the same image bit for bit."""
    from gaussianimage_plus_amd.trainer import NativeFitter
    fit, gt = _cov_fitter(1500, 64, 96, max_points=2000, track_best=True)
    fit.fit(150, prune_iter=50, grow_iter=100)
    fit.load_best()
    img = fit.render().clone()
    sd = fit.state_dict()
    assert set(sd) == {"_xyz", "_cov2d", "_features_dc", "_opacity", "background", "bound"}
    assert sd["_xyz"].shape == (fit.n, 2) and sd["_cov2d"].shape == (fit.n, 3) and sd["_opacity"].shape == (fit.n, 1)
    path = str(tmp_path / "gaussian_model.pth.tar")
    fit.save_checkpoint(path, psnr=fit.psnr())
    ck = torch.load(path, map_location="cpu")
    assert set(ck) == {"gs", "num_gs", "psnr", "ms-ssim", "slv_bound"} and ck["num_gs"] == fit.n
    assert ck["slv_bound"].shape == (fit.n, 3)
    other = NativeFitter(gt, 100, kind="covariance", lr=0.018, eps=1e-15, max_points=2000)
    other.load_checkpoint(path)
    assert other.n == fit.n and torch.equal(other.render(), img)
    # quantiser values travel under the reference's module names
    fit.enable_quantize(12, 10, 6)
    fit.train(5)
    sdq = fit.state_dict()
    for k in ("xyz_quantizer.scale", "xyz_quantizer.beta", "cholesky_quantizer.cov_quantizer.scale",
              "cholesky_quantizer.cov_quantizer.beta", "features_dc_quantizer.scale", "features_dc_quantizer.beta"):
        assert k in sdq
    assert sdq["features_dc_quantizer.scale"].shape == (3,) and sdq["cholesky_quantizer.cov_quantizer.beta"].shape == (1,)
    chol = NativeFitter(gt, 300, kind="cholesky", lr=1e-3)
    assert "_cholesky" in chol.state_dict()
    rs = NativeFitter(gt, 300, kind="scale_rot", lr=1e-3)
    assert {"_scaling", "_rotation"} <= set(rs.state_dict())


def test_concurrent_images_threaded_equals_round_robin():
    """Several images per GPU: one host thread per image issues the same work as the single-threaded round-robin."""
    from gaussianimage_plus_amd.launch import fit_images_native, synthetic_image
    gts = [synthetic_image(96, 144, 40 + i).to(DEV) for i in range(3)]
    kw = dict(lr=0.018, kind="covariance", max_points=1800, prune_iter=50, grow_iter=100, eps=1e-15, eval_renders=1)
    a = fit_images_native(gts, 1200, 350, threaded=False, **kw)
    b = fit_images_native(gts, 1200, 350, threaded=True, **kw)
    for ra, rb in zip(a, b):
        assert ra["mse"] == rb["mse"] and ra["num_gaussians"] == rb["num_gaussians"]  # bitwise: same kernels, same order per image
        assert ra["psnr"] > 20


@pytest.mark.parametrize("optimizer", ["adam", "adan"])
def test_one_fused_update_step_equals_the_torch_optimizer_on_the_same_gradient(optimizer):
    """The optimizer arithmetic of the update kernel in isolation: take the gradient the kernel itself reports
    (dbg_grads) and the state it started from, apply torch.optim.Adam / the Adan statement pinned to the reference
    (tests/helpers_adan.py) to exactly those numbers, and compare the updated parameters and moments -- at step 1 and
    at a later step with non-zero moments.  Same operations in fp32 on both sides: 1e-6 relative."""
    from gaussianimage_plus_amd.launch import synthetic_image
    from gaussianimage_plus_amd.trainer import NativeFitter
    from helpers_adan import AdanRef
    n, h, w, lr = 4000, 96, 144, 1e-2
    gt = synthetic_image(h, w, 5).to(DEV)
    eps = 1e-15 if optimizer == "adan" else 1e-8
    g = torch.Generator().manual_seed(13)
    init = {"xyz": torch.rand(n, 2, generator=g) * torch.tensor([float(w), float(h)]),
            "chol": torch.rand(n, 3, generator=g) * torch.tensor([1.0, 0.3, 1.0]),
            "feat": torch.rand(n, 3, generator=g) * 0.3}  # non-zero colours: every parameter group has a gradient
    fit = NativeFitter(gt, n, kind="covariance", lr=lr, eps=eps, seed=13, debug_grads=True, optimizer=optimizer,
                       init=init)
    names = (("xyz", 0, 2), ("chol", 2, 5), ("feat", 5, 8))
    prev_grad = None
    for step in (1, 2, 3, 7):
        fit.train(step - fit.iteration - 1)                      # up to the step under test
        if step > 1:
            prev_grad = fit.dbg_grads[:n].clone()                # Adan's previous gradient = the last step's
        before = {nm: getattr(fit, nm).clone() for nm, _, _ in names}
        m0 = {nm: getattr(fit, "m_" + nm).clone() for nm, _, _ in names}
        v0 = {nm: getattr(fit, "v_" + nm).clone() for nm, _, _ in names}
        d0 = {nm: getattr(fit, "_d_" + nm)[:n].clone() for nm, _, _ in names} if optimizer == "adan" else None
        fit.train(1)
        torch.cuda.synchronize()
        grads = fit.dbg_grads[:n].clone()
        params = [before[nm].clone().requires_grad_(True) for nm, _, _ in names]
        for p, (nm, a, b) in zip(params, names):
            p.grad = grads[:, a:b].contiguous()
        if optimizer == "adam":
            opt = torch.optim.Adam(params, lr=lr, eps=eps)
            for p, (nm, _, _) in zip(params, names):
                opt.state[p] = {"step": torch.tensor(float(step - 1)), "exp_avg": m0[nm].clone(),
                                "exp_avg_sq": v0[nm].clone()}
            opt.step()
            got_m = {nm: opt.state[p]["exp_avg"] for p, (nm, _, _) in zip(params, names)}
            got_v = {nm: opt.state[p]["exp_avg_sq"] for p, (nm, _, _) in zip(params, names)}
        else:
            opt = AdanRef(params, lr, eps=eps)
            opt.step_count = step - 1
            for s, (nm, a, b) in zip(opt.state, names):
                s["m"], s["n"], s["d"] = m0[nm].clone(), v0[nm].clone(), d0[nm].clone()
                s["prev"] = None if prev_grad is None else prev_grad[:, a:b].contiguous()
            opt.step()
            got_m = {nm: s["m"] for s, (nm, _, _) in zip(opt.state, names)}
            got_v = {nm: s["n"] for s, (nm, _, _) in zip(opt.state, names)}
        for p, (nm, _, _) in zip(params, names):
            moved = (p.detach() - before[nm]).abs()
            assert moved.max().item() > 0.1 * lr, (step, nm)      # the step did something
            err = (getattr(fit, nm) - p.detach()).abs()
            # the update itself is reproduced to 1e-6 of its size, the parameter to the last three ulps of the largest
            # value its arithmetic passes through: Adan applies two terms one after the other (addcdiv_ twice), which
            # may be far larger than their sum
            mag = p.detach().abs() + before[nm].abs()
            if optimizer == "adan":
                s_ = opt.state[[q is p for q in params].index(True)]
                b1, b2, b3 = opt.betas
                denom = s_["n"].sqrt() / math.sqrt(1.0 - b3 ** step) + eps
                mag = mag + (lr / (1.0 - b1 ** step)) * (s_["m"] / denom).abs() \
                    + (lr * b2 / (1.0 - b2 ** step)) * (s_["d"] / denom).abs()
            tol = 1e-6 * moved + 4e-7 * mag + 1e-12
            worst = (err - tol).argmax()
            assert (err <= tol).all(), (step, nm, err.flatten()[worst].item(), tol.flatten()[worst].item(),
                                        moved.flatten()[worst].item(), p.detach().flatten()[worst].item())
            for got, ref, what in ((getattr(fit, "m_" + nm), got_m[nm], "m"), (getattr(fit, "v_" + nm), got_v[nm], "v")):
                assert torch.allclose(got, ref, rtol=1e-6, atol=1e-30), (step, nm, what)


def test_large_image_fit_switches_to_two_launches_and_equals_a_single_call():
    """A single image of more than one residency round of tiles (1040x1040: 4225) fitted by gi2d_train_steps: the first
    call on its workspace runs the tile pass as one launch, reports no tile above the small form's capacity, and later
    calls run it as two launches (include/gi2d.h: gi2d_batch_tile_pass_form, keyed by the workspace) -- the state equals
    the same iterations issued as ONE call (all general form) bit for bit.  One crowded tile does not change the form
    (two launches while at most one tile in sixteen is too full for the small one); a tenth of the tiles does."""
    import ctypes as C
    from gaussianimage_plus_amd import _lib
    from gaussianimage_plus_amd.launch import synthetic_image
    from gaussianimage_plus_amd.trainer import NativeFitter
    lib = _lib.load()
    h = w = 1040
    n = 6000

    def fitter():
        rng = np.random.default_rng(4)
        u = rng.uniform(-0.98, 0.98, (n, 2))
        L = np.stack([rng.uniform(0.8, 1.6, n), rng.uniform(-0.3, 0.3, n), rng.uniform(0.8, 1.6, n)], 1)
        init = {"xyz": torch.from_numpy(np.arctanh(u).astype(np.float32)), "chol": torch.from_numpy(L.astype(np.float32)),
                "feat": torch.from_numpy(rng.uniform(0, 0.2, (n, 3)).astype(np.float32)),
                "bound": torch.tensor([0.5, 0.0, 0.5])}
        return NativeFitter(synthetic_image(h, w, 9).to(DEV), n, kind="cholesky", lr=1e-3, seed=5, init=init)

    form = lambda f: int(lib.gi2d_batch_tile_pass_form(C.c_void_p(f.ws.data_ptr())))
    a, b = fitter(), fitter()
    assert form(a) == 0
    a.train(2)
    torch.cuda.synchronize()
    assert form(a) == 1
    a.train(3)  # two launches per tile pass
    b.train(5)  # one call on a fresh workspace: one launch per tile pass
    torch.cuda.synchronize()
    a.check_status(), b.check_status()
    for nm in ("_xyz", "_chol", "_feat", "m_xyz", "v_chol", "out_img", "tile_sse"):
        assert torch.equal(getattr(a, nm), getattr(b, nm)), nm
    # 600 gaussians into one tile: the second launch has work, the form stays
    rng = np.random.default_rng(77)
    u = np.stack([(rng.uniform(40, 48, 600) + 0.5) / (0.5 * w) - 1.0, (rng.uniform(40, 48, 600) + 0.5) / (0.5 * h) - 1.0], 1)
    a._xyz[:600] = torch.from_numpy(np.arctanh(u).astype(np.float32)).to(DEV)
    a.train(2)
    torch.cuda.synchronize()
    a.check_status()
    assert form(a) == 1 and int(a.status[3]) >= 512
    # 160 gaussians over the top-left 320 x 320 pixels (400 of the 4 225 tiles get more than 128 candidates): the next
    # call (still two launches) reports them, the one after runs as one launch
    v = np.stack([(rng.uniform(150, 170, 160) + 0.5) / (0.5 * w) - 1.0, (rng.uniform(150, 170, 160) + 0.5) / (0.5 * h) - 1.0], 1)
    a._xyz[600:760] = torch.from_numpy(np.arctanh(v).astype(np.float32)).to(DEV)
    a._chol[600:760] = torch.tensor([60.0, 0.0, 60.0], device=DEV)
    a.train(2)
    torch.cuda.synchronize()
    a.check_status()
    assert form(a) == 0
