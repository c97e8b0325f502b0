"""The tile kernels skip every pixel outside a conservative `alpha >= 1/255` box around a gaussian
(csrc/gi2d_common.h::cull_extent: half extents sqrt(tau2 * c / det) * 1.0002 + 1/16 px with
tau2 = 2 ln(255 opacity) * 1.0002 + 1e-3).  This is the arithmetic of that function and of the kernels' pair evaluation
(gi2d_common.h::scale_conic / pair_sigma / pair_vis) restated in numpy float32, with the hardware's 1-ulp sqrt / rcp / log
pushed the WRONG way (extents shaved by 4 ulp, acceptance threshold lowered by 1e-4 relative): no pixel that could pass
the alpha test may lie outside the box, over shapes far beyond what a fit produces -- major axes 0.25 ... 150 px,
aspect ratios to 200, any orientation, opacities 0.004 ... 3, sub-pixel centre positions.  The GPU twin
(test_hip_parity.py::test_cull_box_never_drops_a_contributing_pair) holds the kernels themselves to the oracle."""
import numpy as np

F = np.float32
MARGIN = F(0.0625)  # GI2D_CULL_MARGIN


def _extents(a, b, c, opac):
    det = a * c - b * b
    tau2 = F(2.0) * np.log(opac * F(255.0)).astype(F) * F(1.0002) + F(1e-3)
    t = tau2 / det
    shave = F(1.0 - 5e-7)  # 4 ulp against the hardware's approximate sqrt / rcp / log
    ex = np.sqrt(t * c).astype(F) * shave * F(1.0002) + MARGIN
    ey = np.sqrt(t * a).astype(F) * shave * F(1.0002) + MARGIN
    return ex, ey


def _alpha(a, b, c, opac, dx, dy):
    l2e = F(1.4426950408889634)
    ha, hb, hc = F(0.5) * a * l2e, b * l2e, F(0.5) * c * l2e
    sig = dx * (ha * dx + hb * dy) + hc * dy * dy  # float32 throughout (numpy keeps the dtype)
    vis = np.exp2(-sig.astype(np.float64)).astype(F)
    return sig, opac * vis


def test_no_contributing_pixel_outside_the_box():
    rng = np.random.default_rng(5)
    n = 400
    major = np.exp(rng.uniform(np.log(0.25), np.log(150.0), n))
    minor = np.maximum(major / np.exp(rng.uniform(0.0, np.log(200.0), n)), 0.05)
    th = rng.uniform(0, np.pi, n)
    cs, sn = np.cos(th), np.sin(th)
    sxx = cs * cs * major ** 2 + sn * sn * minor ** 2
    sxy = cs * sn * (major ** 2 - minor ** 2)
    syy = sn * sn * major ** 2 + cs * cs * minor ** 2
    det = sxx * syy - sxy * sxy
    conic = np.stack([syy / det, -sxy / det, sxx / det], 1).astype(F)  # inverse covariance, as the projection leaves it
    opac = np.exp(rng.uniform(np.log(0.004), np.log(3.0), n)).astype(F)
    centre = rng.uniform(0, 64, (n, 2)).astype(F)
    worst = 0.0
    checked = 0
    for g in range(n):
        a, b, c = conic[g]
        if not (a > 0 and c > 0 and a * c - b * b > 0):
            continue  # the kernels evaluate every pixel of such a gaussian (no box)
        ex, ey = _extents(a, b, c, opac[g])
        gx, gy = centre[g]
        # integer pixels in a window three pixels wider than the box (the ellipse cannot reach further)
        x0, x1 = int(np.floor(gx - ex)) - 3, int(np.ceil(gx + ex)) + 3
        y0, y1 = int(np.floor(gy - ey)) - 3, int(np.ceil(gy + ey)) + 3
        xs = np.arange(x0, x1 + 1, dtype=F)
        ys = np.arange(y0, y1 + 1, dtype=F)
        dx = (gx - xs)[None, :].astype(F)
        dy = (gy - ys)[:, None].astype(F)
        sig, t = _alpha(a, b, c, opac[g], dx, dy)
        passes = (sig >= 0) & (np.minimum(F(1.0), t) >= F(1.0 / 255.0) * F(1.0 - 1e-4))
        outside = (np.abs(dx) > ex) | (np.abs(dy) > ey)
        bad = passes & outside
        checked += int(passes.sum())
        if bad.any():
            iy, ix = np.nonzero(bad)
            worst = max(worst, float(np.maximum(np.abs(dx[0, ix]) - ex, np.abs(dy[iy, 0]) - ey).max()))
        assert not bad.any(), (g, major[g], minor[g], float(opac[g]), worst)
    assert checked > 100000  # the sweep did look at contributing pixels
