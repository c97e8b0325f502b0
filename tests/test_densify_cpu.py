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
