"""CPU: the host-side pieces of densification / pruning (gaussianimage_plus_amd/trainer.py) against a second,
independently written statement of train.py:85-118 and models/gaussianimage_covariance.py:317-320,372-379."""
import numpy as np
import torch

from gaussianimage_plus_amd.trainer import growth_budget, positive_definite_mask, select_new_points


def test_positive_definite_mask_cases():
    cov = torch.tensor([[1.0, 0.0, 1.0],      # PD
                        [1.0, 1.0, 1.0],      # singular (det = 0) -> excluded
                        [1.0, 2.0, 1.0],      # indefinite
                        [-1.0, 0.0, -1.0],    # det > 0 but negative diagonal
                        [0.5, 0.49, 0.5],     # PD, barely
                        [float("nan"), 0.0, 1.0]])
    assert positive_definite_mask(cov).tolist() == [True, False, False, False, True, False]


def test_growth_budget_schedule():
    # 50 000 iterations, growth every 5 000 (train.py defaults): 1000 per step, the rest at iteration 45 000
    n, mx = 2500, 5000
    added = []
    for it in range(5000, 50000, 5000):
        k = growth_budget(it, 50000, 5000, n, mx)
        added.append(k)
        n += k
    assert added == [1000, 1000, 500, 0, 0, 0, 0, 0, 0] and n == mx
    # a large budget is only released at the last growth step
    assert growth_budget(5000, 50000, 5000, 5000, 50000) == 1000
    assert growth_budget(45000, 50000, 5000, 13000, 50000) == 37000
    assert growth_budget(45000, 50000, 5000, 50000, 50000) == 0


def test_select_new_points_matches_plain_numpy_statement():
    rng = np.random.default_rng(3)
    h, w, k = 37, 53, 200
    render = rng.random((h, w, 3)).astype(np.float32)
    gt = rng.random((h, w, 3)).astype(np.float32)
    rand3 = torch.from_numpy(rng.random((k, 3)).astype(np.float32))
    got = select_new_points(torch.from_numpy(render), torch.from_numpy(gt), k, rand3)
    # numpy: per-pixel sum of absolute errors, k largest, (x, y) = (index mod W, index div W)
    err = np.abs(render - gt).sum(axis=2, dtype=np.float32).reshape(-1)
    order = np.argsort(-err, kind="stable")[:k]
    cov = rand3.numpy() + np.array([0.5, 0.0, 0.5], np.float32)
    keep = (cov[:, 0] * cov[:, 2] - cov[:, 1] ** 2 > 0) & (cov[:, 0] > 0) & (cov[:, 2] > 0)
    assert 0 < keep.sum() < k  # the seeded draw contains non-definite covariances, which must be dropped
    want_xy = np.stack([order % w, order // w], 1).astype(np.float32)
    # top-k returns the same SET; errors are distinct here, so the descending order is the same too
    assert len(np.unique(err[order])) == k
    assert np.array_equal(got["xyz"].numpy(), want_xy[keep])
    assert np.array_equal(got["cov2d"].numpy(), cov[keep])
    assert got["feat"].shape == (int(keep.sum()), 3) and float(got["feat"].abs().sum()) == 0.0
    assert got["dropped"] == k - int(keep.sum())


def _reference_quantize_events(iterations, warmup_iter, prune_iter, grow_iter, n0, max_points):
    """train_quantize.py:124-175 as a plain loop over `iter`, recording what happens when (growth adds its whole
    budget: no non-PD draws in this statement)."""
    ev, n = [], n0
    for it in range(1, iterations):
        ev_it = []
        if it == warmup_iter:
            ev_it.append(("switch", n))
        if it % prune_iter == 0 and it < warmup_iter:
            ev_it.append(("prune", n))
        if it % grow_iter == 0 and it < warmup_iter:
            k = max(0, max_points - n) if it == iterations - grow_iter else max(0, min(1000, max_points - n))
            if k:
                n += k
                ev_it.append(("grow", k))
        ev += [(it,) + e for e in ev_it]
    ev.append((iterations - 1, "prune", n))
    return ev


class _ScheduleProbe:
    """The host logic of NativeFitter.fit_schedule / fit_quantize_schedule with the device calls replaced by a log."""
    kind, track_best, cap, device_resident = "covariance", True, 10 ** 9, False

    def __init__(self, n0):
        from gaussianimage_plus_amd.trainer import NativeFitter
        self.iteration, self.n, self.events, self.switched_at = 0, n0, [], None
        self.fit_schedule = NativeFitter.fit_schedule.__get__(self)
        for nm in ("_next_stop", "_schedule_events", "_schedule_end"):  # the pieces fit_schedule is made of
            setattr(self, nm, getattr(NativeFitter, nm).__get__(self))
        self.fit_quantize_schedule = NativeFitter.fit_quantize_schedule.__get__(self)

    def train(self, k):
        self.iteration += int(k)

    def prune_non_definite(self):
        self.events.append((self.iteration, "prune", self.n))
        return 0

    def add_sample_positions(self, iteration, iterations, grow_iter, max_points=None):
        k = growth_budget(iteration, iterations, grow_iter, self.n, max_points)
        if k:
            self.n += k
            self.events.append((self.iteration, "grow", k))
        return k

    def load_best(self):
        pass

    def sync_population(self):
        return self.n

    def enable_quantize(self, *bits):
        self.events.append((self.iteration + 1, "switch", self.n))


@__import__("pytest").mark.parametrize("iterations,warmup,prune,grow", [(10000, 6000, 1000, 5000), (50000, 6000, 100, 5000),
                                                                      (3000, 2000, 100, 500), (1200, 1000, 100, 1000)])
def test_quantize_schedule_follows_train_quantize(iterations, warmup, prune, grow):
    """Events of fit_quantize_schedule == train_quantize.py's loop, including the case iterations - grow_iter <
    warmup_iter, where the LAST growth step (whole remaining budget) falls inside the warm-up, and a growth / prune
    check at iteration warmup_iter - 1."""
    want = _reference_quantize_events(iterations, warmup, prune, grow, 2500, 20000)
    probe = _ScheduleProbe(2500)
    for _ in probe.fit_quantize_schedule(iterations, warmup, prune_iter=prune, grow_iter=grow, max_points=20000):
        pass
    got = probe.events
    # the port prunes the restored best model once more at the switch (its snapshot is taken before the prune)
    extra = [e for e in got if e[1] == "prune" and e[0] == warmup - 1 and (warmup - 1) % prune != 0]
    assert len(extra) == 1
    got = [e for e in got if e not in extra]
    assert sorted(got) == sorted(want)
    assert probe.iteration == iterations - 1


# ------------------------------------------------------------------ the reference's own prune / growth code (fixture)
def _densify_fixture():
    """tests/golden/densify_reference.npz: inputs and outputs of the REFERENCE's check_non_semi_definite /
    non_semi_definite_prune / add_sample_positions / densification_postfix run on CPU
    (tests/golden/make_densify_golden.py imports models/gaussianimage_covariance.py and train.py)."""
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "densify_reference.npz"))


def test_prune_statement_equals_the_reference_run():
    """models/gaussianimage_covariance.py:339-382: mask, compaction of parameters, both Adam moments and the bound."""
    fx = _densify_fixture()
    full = torch.from_numpy(fx["p0_cov2d"] + fx["p0_bound"])
    valid = positive_definite_mask(full).numpy()
    assert np.array_equal(valid, fx["prune_valid_mask"])
    pruned, n1 = (int(v) for v in fx["prune_counts"])
    assert pruned == int((~valid).sum()) and n1 == int(valid.sum())
    for nm in ("xyz", "cov2d", "f_dc"):
        for pre in ("", "m_", "v_"):
            assert np.array_equal(fx[f"p0_{pre}{nm}"][valid], fx[f"p1_{pre}{nm}"]), (pre, nm)
        assert fx[f"p0_step_{nm}"] == fx[f"p1_step_{nm}"]  # the step count survives the surgery
    assert np.array_equal(fx["p0_bound"][valid], fx["p1_bound"])
    assert fx["p1_opacity"].shape == (n1, 1) and np.all(fx["p1_opacity"] == 1)


@__import__("pytest").mark.parametrize("tag,prev", [("g1", "p1"), ("g2", "g1"), ("g3", "g2")])
def test_growth_statement_equals_the_reference_run(tag, prev):
    """train.py:85-118 + densification_postfix (:307-337): budget, top-k pixels, draws, PD filter, appended rows, zero
    moments, the new rows' bound from the NEW population size."""
    import math
    fx = _densify_fixture()
    h, w, _, _, iterations, grow_iter = (int(v) for v in fx["dims"])
    it, max_points, cur, k, new_n = (int(v) for v in fx[f"{tag}_args"])
    assert fx[f"{prev}_xyz"].shape[0] == cur
    assert growth_budget(it, iterations, grow_iter, cur, max_points) == k == fx[f"{tag}_rand3"].shape[0]
    got = select_new_points(torch.from_numpy(fx[f"{tag}_render"]), torch.from_numpy(fx["gt"]), k,
                            torch.from_numpy(fx[f"{tag}_rand3"]))
    kept = new_n - cur
    assert got["xyz"].shape[0] == kept and got["dropped"] == k - kept and 0 < kept < k
    assert np.array_equal(got["xyz"].numpy(), fx[f"{tag}_xyz"][cur:])
    assert np.array_equal(got["cov2d"].numpy(), fx[f"{tag}_cov2d"][cur:])
    assert not fx[f"{tag}_f_dc"][cur:].any()
    for nm in ("xyz", "cov2d", "f_dc"):  # the old rows and their moments stay, the new rows' moments are zero
        for pre in ("", "m_", "v_"):
            assert np.array_equal(fx[f"{tag}_{pre}{nm}"][:cur], fx[f"{prev}_{pre}{nm}"]), (pre, nm)
        assert not fx[f"{tag}_m_{nm}"][cur:].any() and not fx[f"{tag}_v_{nm}"][cur:].any()
    low = np.float32(min(h * w / (9 * math.pi * new_n), 300))
    assert np.array_equal(fx[f"{tag}_bound"][:cur], fx[f"{prev}_bound"])
    assert np.array_equal(fx[f"{tag}_bound"][cur:], np.tile(np.array([low, 0, low], np.float32), (kept, 1)))
