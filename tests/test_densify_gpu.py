"""GPU: pruning and growth with the population count on the device (csrc/gi2d_densify.hip) against the independent
numpy statement of train.py:85-118 / models/gaussianimage_covariance.py:307-382 -- the one tests/test_densify_cpu.py
holds the host version to -- on a 768x512 image growing 5 000 -> 50 000 gaussians."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _numpy_growth(render, gt, n_live, max_points, budget_cap, rand3, w):
    """add_sample_positions + densification_postfix, plain numpy."""
    k = max(0, min(budget_cap, max_points - n_live))
    err = np.abs(np.clip(render, 0, 1) - gt).sum(axis=2, dtype=np.float32).reshape(-1)
    order = np.argsort(-err, kind="stable")[:k]               # torch.topk: largest first; ties: lower index first
    cov = rand3[:k] + np.array([0.5, 0.0, 0.5], np.float32)
    keep = (cov[:, 0] * cov[:, 2] - cov[:, 1] ** 2 > 0) & (cov[:, 0] > 0) & (cov[:, 2] > 0)
    xy = np.stack([order % w, order // w], 1).astype(np.float32)
    return xy[keep], cov[keep], int(keep.sum()), k


def _rows(fit, n):
    names = ["_xyz", "_chol", "_feat", "_opacity", "_m_xyz", "_v_xyz", "_m_chol", "_v_chol", "_m_feat", "_v_feat", "_bound"]
    return {nm: getattr(fit, nm)[:n].cpu().numpy().copy() for nm in names}


def test_growth_5000_to_50000_matches_numpy_statement():
    from gaussianimage_plus_amd.launch import synthetic_image
    from gaussianimage_plus_amd.trainer import NativeFitter
    h, w, n0, cap = 512, 768, 5000, 50000
    gt = synthetic_image(h, w, 31).to(DEV)
    fit = NativeFitter(gt, n0, kind="covariance", lr=0.018, eps=1e-15, seed=9, max_points=cap, track_best=True,
                       device_resident=True)
    gt_np = gt.cpu().numpy()
    iterations, grow_iter = 400, 40
    live = n0
    for step in range(1, 10):
        fit.train(grow_iter)
        fit.prune_non_definite()
        it = step * grow_iter
        live = fit.sync_population()          # test only: the loop itself never reads the count back
        before = _rows(fit, live)
        render = fit.out_img.cpu().numpy()
        state = fit.rng.get_state()
        budget_cap = cap if it == iterations - grow_iter else 1000
        assert fit.add_sample_positions(it, iterations, grow_iter) is None
        fit.rng.set_state(state)
        rand3 = torch.rand(min(budget_cap, cap), 3, generator=fit.rng).numpy()
        xy, cov, kept, k = _numpy_growth(render, gt_np, live, cap, budget_cap, rand3, w)
        assert fit.n == min(cap, live + budget_cap)           # the host's upper bound
        new_live = fit.sync_population()
        assert new_live == live + kept and 0 < kept <= k, (step, new_live, live, kept, k)
        after = _rows(fit, new_live)
        for nm in before:                                      # the old rows are untouched
            assert np.array_equal(after[nm][:live], before[nm]), (step, nm)
        assert np.array_equal(after["_xyz"][live:], xy) and np.array_equal(after["_chol"][live:], cov), step
        assert not after["_feat"][live:].any() and np.all(after["_opacity"][live:] == 1)
        for nm in ("_m_xyz", "_v_xyz", "_m_chol", "_v_chol", "_m_feat", "_v_feat"):
            assert not after[nm][live:].any(), nm
        low = np.float32(min(h * w / (9 * math.pi * new_live), 300))
        assert np.array_equal(after["_bound"][live:], np.tile(np.array([low, 0, low], np.float32), (kept, 1)))
        live = new_live
    assert 0.85 * cap < live <= cap                            # the last step released the whole remaining budget (a tenth of the draws is not positive definite)
    assert fit.dens_counts[1].item() == live - n0 + fit.dens_counts[0].item()
    fit.train(20)
    fit.check_status()
    assert fit.psnr() > 20


def test_prune_compacts_every_array_in_order():
    from gaussianimage_plus_amd.launch import synthetic_image
    from gaussianimage_plus_amd.trainer import NativeFitter
    h, w, n0 = 96, 144, 3000
    gt = synthetic_image(h, w, 5).to(DEV)
    for opt in ("adam", "adan"):
        fit = NativeFitter(gt, n0, kind="covariance", lr=0.018, eps=1e-15, seed=2, max_points=4000, optimizer=opt,
                           device_resident=True)
        fit.train(30)                                          # non-zero moments everywhere
        rng = np.random.default_rng(1)
        bad = np.sort(rng.choice(n0, 137, replace=False))
        cov = fit._chol[:n0].clone()
        cov[torch.from_numpy(bad).to(DEV)] = torch.tensor([0.2, 5.0, 0.3], device=DEV) - fit._bound[:n0][torch.from_numpy(bad).to(DEV)]
        cov[bad[0]] = -fit._bound[bad[0]]                      # covariance + bound == 0: det == 0
        fit._chol[:n0] = cov
        names = ["_xyz", "_chol", "_feat", "_opacity", "_m_xyz", "_v_xyz", "_m_chol", "_v_chol", "_m_feat", "_v_feat",
                 "_bound"] + (["_d_xyz", "_d_chol", "_d_feat", "_pg_xyz", "_pg_chol", "_pg_feat"] if opt == "adan" else [])
        before = {nm: getattr(fit, nm)[:n0].cpu().numpy().copy() for nm in names}
        full = before["_chol"] + before["_bound"]
        keep = (full[:, 0] * full[:, 2] - full[:, 1] ** 2 > 0) & (full[:, 0] > 0) & (full[:, 2] > 0)
        assert (~keep).sum() >= 137
        assert fit.prune_non_definite() is None
        assert fit.n == n0                                     # the host's bound does not move
        live = fit.sync_population()
        assert live == int(keep.sum()) and fit.dens_counts[0].item() == n0 - live
        for nm in names:
            assert np.array_equal(getattr(fit, nm)[:live].cpu().numpy(), before[nm][keep]), (opt, nm)
        # a second check finds nothing and moves nothing
        fit.prune_non_definite()
        assert fit.sync_population() == live
        fit.train(10)
        fit.check_status()
        assert np.isfinite(fit.psnr())


def test_device_resident_fit_equals_host_driven_fit():
    """The same schedule with the count on the device and with the host-side torch path: identical models."""
    from gaussianimage_plus_amd.launch import synthetic_image
    from gaussianimage_plus_amd.trainer import NativeFitter
    # budgets that the cap does not clip before the last growth step: the host path draws exactly `budget` rows of
    # uniform numbers per step, the device path `budget_cap` rows (it does not know the live count), so the two consume
    # the generator alike only while budget == budget_cap -- the numbers each USES are the same rows either way
    h, w, n0, cap = 96, 144, 400, 4000
    gt = synthetic_image(h, w, 7).to(DEV)
    fits = []
    for dr in (False, True):
        fit = NativeFitter(gt, n0, kind="covariance", lr=0.018, eps=1e-15, seed=4, max_points=cap, track_best=True,
                           device_resident=dr)
        fit.fit(80, prune_iter=10, grow_iter=20)
        fit.check_status()
        fits.append(fit)
    a, b = fits
    assert a.n == b.n and a.iteration == b.iteration
    for nm in ("xyz", "chol", "feat", "bound", "m_chol", "v_feat"):
        assert torch.equal(getattr(a, nm), getattr(b, nm)), nm
    assert a.best()[1:] == b.best()[1:]


# ------------------------------------------------------------------ against the reference's own prune / growth code
_FX_ROWS = (("_xyz", "xyz"), ("_chol", "cov2d"), ("_feat", "f_dc"), ("_m_xyz", "m_xyz"), ("_v_xyz", "v_xyz"),
            ("_m_chol", "m_cov2d"), ("_v_chol", "v_cov2d"), ("_m_feat", "m_f_dc"), ("_v_feat", "v_f_dc"),
            ("_bound", "bound"), ("_opacity", "opacity"))


@pytest.mark.parametrize("device_resident", [True, False])
def test_prune_and_growth_equal_the_reference_run(device_resident, monkeypatch):
    """tests/golden/densify_reference.npz holds the state of the REFERENCE's model + Adam before and after its own
    non_semi_definite_prune (models/gaussianimage_covariance.py:354-370) and three add_sample_positions /
    densification_postfix steps (train.py:85-118, :307-337) -- an ordinary one, one the cap clips, the last one that
    releases the whole budget.  gi2d_train_prune / gi2d_train_grow (and the host-driven torch path) start from the
    same rows, renders and uniform draws and must leave the same rows, bit for bit."""
    import os
    from gaussianimage_plus_amd.trainer import NativeFitter
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "densify_reference.npz"))
    h, w, n0, cap, iterations, grow_iter = (int(v) for v in fx["dims"])
    gt = torch.from_numpy(fx["gt"]).to(DEV)
    fit = NativeFitter(gt, n0, kind="covariance", lr=0.018, eps=1e-15, seed=1, max_points=cap, track_best=True,
                       device_resident=device_resident)
    assert fit.per_point_bound
    for attr, key in _FX_ROWS:
        getattr(fit, attr)[:n0] = torch.from_numpy(fx["p0_" + key]).to(DEV)

    def rows(n):
        return {key: getattr(fit, attr)[:n].cpu().numpy() for attr, key in _FX_ROWS}

    def same(tag, n):
        got = rows(n)
        for key in got:
            assert np.array_equal(got[key], fx[f"{tag}_{key}"]), (tag, key)

    # ---- prune
    pruned, n1 = (int(v) for v in fx["prune_counts"])
    res = fit.prune_non_definite()
    assert (res is None) if device_resident else (res == pruned)
    assert fit.sync_population() == n1
    same("p1", n1)
    # ---- growth: the render the reference was given, the uniform numbers it drew
    real_rand = torch.rand
    for tag in ("g1", "g2", "g3"):
        it, max_points, cur, k, new_n = (int(v) for v in fx[f"{tag}_args"])
        assert fit.sync_population() == cur
        fit.out_img.copy_(torch.from_numpy(fx[f"{tag}_render"]).to(DEV))
        draws = torch.from_numpy(fx[f"{tag}_rand3"])

        def fake_rand(rows_, cols, generator=None):  # the first k rows are what the reference drew (train.py:111)
            assert cols == 3 and rows_ >= k
            out = real_rand(rows_, 3, generator=generator)
            out[:k] = draws
            return out

        monkeypatch.setattr(torch, "rand", fake_rand)
        try:
            res = fit.add_sample_positions(it, iterations, grow_iter, max_points=max_points)
        finally:
            monkeypatch.setattr(torch, "rand", real_rand)
        assert (res is None) if device_resident else (res == new_n - cur)
        assert fit.sync_population() == new_n, tag
        same(tag, new_n)
    if device_resident:
        assert fit.dens_counts.tolist() == [pruned, int(fx["g3_args"][4]) - n1]
    fit.train(5)  # the fit goes on from the reference's state
    fit.check_status()
    assert np.isfinite(fit.psnr())
