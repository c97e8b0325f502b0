"""Shared helpers for the parity tests: seeded synthetic inputs and the
tolerance rules.

Floating-point bar (BASELINE.json north_star): 1e-5 relative fp32.  "Relative" is taken against the
sum of absolute contributions of the quantity (pixel value or gradient component), which the oracle
reports, so that cancellation does not turn rounding noise into a failure.  Pairs whose alpha sits on
the 1/255 cut-off are evaluated with different exp implementations on host and device and may land
on either side (SURVEY.md section 7 "Threshold flips"); the oracle flags the pixels / gaussians such
pairs touch and they are excluded from the floating-point comparison (they are a ~1e-4 fraction).
"""
import math

import numpy as np

RTOL = 1e-5


def synth_cholesky(n, h, w, seed):
    rng = np.random.default_rng(seed)
    u = np.clip(2 * (rng.random((n, 2)) - 0.5), -0.999999, 0.999999)
    xyz = np.tanh(np.arctanh(u)).astype(np.float32)
    lp = min(h * w / (9 * math.pi * n), 300)
    L = (rng.random((n, 3)) + np.array([lp, 0, lp])).astype(np.float32)
    col = rng.random((n, 3)).astype(np.float32)
    op = np.ones((n, 1), np.float32)
    return xyz, L, col, op


def synth_gt(h, w, seed):
    """Seeded smooth image in [0,1] standing in for a Kodak picture."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w, 3), np.float32)
    for c in range(3):
        for _ in range(6):
            fx, fy = rng.uniform(0.5, 6, 2)
            ph = rng.uniform(0, 2 * np.pi, 2)
            img[..., c] += np.sin(2 * np.pi * fx * xx / w + ph[0]) * np.cos(2 * np.pi * fy * yy / h + ph[1])
    img = (img - img.min()) / (img.max() - img.min())
    return img.astype(np.float32)


def check_close(name, got, want, scale, mask=None, rtol=RTOL, atol=1e-9, max_bad_frac=0.0):
    got, want, scale = np.asarray(got, np.float64), np.asarray(want, np.float64), np.asarray(scale, np.float64)
    err = np.abs(got - want)
    tol = rtol * np.maximum(scale, np.abs(want)) + atol
    bad = err > tol
    if mask is not None:
        bad &= mask
    nbad = int(bad.sum())
    total = int(mask.sum()) if mask is not None else bad.size
    worst = float((err / tol)[bad].max()) if nbad else float((err / tol)[mask].max() if mask is not None else (err / tol).max())
    if rtol != RTOL or max_bad_frac > 0:
        # a comparison that asks for more room than the 1e-5 bar: say how much of it is used (pytest -s / -rP shows it)
        tol5 = RTOL * np.maximum(scale, np.abs(want)) + atol
        over = err > tol5
        if mask is not None:
            over &= mask
        where = np.argwhere(over)
        print(f"[tolerance] {name}: {int(over.sum())}/{total} elements beyond rtol=1e-5 "
              f"(worst err/tol(1e-5) = {float((err / tol5)[over].max()) if over.any() else float((err / tol5)[mask].max() if mask is not None else (err / tol5).max()):.3g}"
              f"{'; first at ' + str(where[:4].tolist()) if over.any() else ''}); asked: rtol={rtol}, "
              f"max_bad_frac={max_bad_frac}")
    assert nbad <= max_bad_frac * total, (
        f"{name}: {nbad}/{total} elements beyond rtol={rtol} (worst err/tol = {worst:.3g})")
    return worst


def rs_term_magnitudes(conics, v_conic, v_xy, scales, rot, nm):
    """Sum of the absolute terms each scale-rot gradient entry is made of (float64)."""
    conic, vc = np.asarray(conics, np.float64), np.asarray(v_conic, np.float64)
    X = np.abs(np.stack([np.stack([conic[:, 0], conic[:, 1]], 1), np.stack([conic[:, 1], conic[:, 2]], 1)], 1))
    G = np.abs(np.stack([np.stack([vc[:, 0], vc[:, 1]], 1), np.stack([vc[:, 1], vc[:, 2]], 1)], 1))
    S = X @ G @ X
    gm = np.stack([S[:, 0, 0], S[:, 0, 1] + S[:, 1, 0], S[:, 1, 1]], 1)
    if nm == "v_cov2d":
        return gm + 1e-30
    if nm == "v_mean2d":
        return np.abs(np.asarray(v_xy, np.float64)) + 1e-30
    s, r = np.asarray(scales, np.float64), np.asarray(rot, np.float64).reshape(-1)
    c, si = np.abs(np.cos(r)), np.abs(np.sin(r))
    sx, sy = np.abs(s[:, 0]), np.abs(s[:, 1])
    # |d Sigma / d sx| <= 2 sx (c^2, c s, s^2); |d Sigma / d sy| <= 2 sy (s^2, c s, c^2); |d Sigma / d theta| <= (sx^2 + sy^2) (2 c s, 1, 2 c s)
    if nm == "v_scale":
        return np.stack([2 * sx * (gm[:, 0] * c * c + 2 * gm[:, 1] * c * si + gm[:, 2] * si * si),
                         2 * sy * (gm[:, 0] * si * si + 2 * gm[:, 1] * c * si + gm[:, 2] * c * c)], 1) + 1e-30
    q = sx * sx + sy * sy
    return (q * (gm[:, 0] * 2 * c * si + 2 * gm[:, 1] + gm[:, 2] * 2 * c * si))[:, None] + 1e-30
