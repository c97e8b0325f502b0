"""CPU: the oracle against tests/golden/ref_vectors.npz -- arrays produced by the reference's own Python helpers
(gsplat/gsplat/_torch_impl.py) and by float64 autograd of the formulas the kernels define (make_ref_vectors.py).
None of the expected values here comes from this repo's oracle."""
import os

import numpy as np
import pytest

from helpers import check_close


@pytest.fixture(scope="module")
def rv(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_vectors.npz"))


def cov3_of(cov22):
    return np.stack([cov22[:, 0, 0], cov22[:, 0, 1], cov22[:, 1, 1]], -1).astype(np.float32)


def test_compute_cov2d_bounds_equals_reference_arrays(oracle, rv):
    """_torch_impl.py:197 (gsplat/tests/test_cov2d_bounds.py's CPU side): conic and radius, element by element."""
    conic, radius = oracle.compute_cov2d_bounds(cov3_of(rv["cov2d_in"]), 3.0)
    v = rv["cov2d_valid"]
    assert v.all()
    np.testing.assert_allclose(conic, rv["cov2d_conic"], rtol=3e-7, atol=0)  # 1/det * x against x / det: one rounding
    assert np.array_equal(radius[:, 0], rv["cov2d_radius"])


def test_tile_bbox_and_map_equal_reference_arrays(oracle, rv):
    """_torch_impl.py:236,297: bit-exact keys (tile id << 32 | depth bits) and ids in gaussian order."""
    h, w = rv["bbox_hw"]
    tb = oracle.tile_bounds(int(h), int(w))
    # the oracle exposes the bbox through map: feed every survivor with the reference's own cumulative counts
    isect, gids = oracle.map_gaussian_to_intersects(len(rv["map_radii"]), int(rv["map_cum"][-1]), rv["map_xys"],
                                                    rv["map_depths"], rv["map_radii"], rv["map_cum"], tb, 1.0)
    assert np.array_equal(isect, rv["map_isect"]) and np.array_equal(gids, rv["map_gids"])
    # tile counts of ALL boxes (also the empty ones): area of the reference's (tile_min, tile_max)
    area = (rv["bbox_max"][:, 0] - rv["bbox_min"][:, 0]) * (rv["bbox_max"][:, 1] - rv["bbox_min"][:, 1])
    n = len(area)
    cum = np.cumsum(area).astype(np.int32)
    isect2, gids2 = oracle.map_gaussian_to_intersects(n, int(cum[-1]), rv["bbox_centre"], np.zeros(n, np.float32),
                                                      rv["bbox_radius"].astype(np.int32), cum, tb, 1.0)
    assert np.array_equal(np.bincount(gids2, minlength=n), area)
    first = np.concatenate([[0], cum[:-1]])
    has = area > 0
    tiles_first = (isect2[first[has]] >> 32).astype(np.int64)
    assert np.array_equal(tiles_first, rv["bbox_min"][has, 1].astype(np.int64) * tb[0] + rv["bbox_min"][has, 0])


def test_sort_and_bin_edges_equal_reference_arrays(oracle, rv):
    """torch.sort(stable) + gather, and _torch_impl.py:328 (gsplat/tests/test_get_tile_bin_edges.py's CPU side)."""
    so, go = oracle.sort_intersects(rv["map_isect"], rv["map_gids"])
    assert np.array_equal(so, rv["sort_isect"]) and np.array_equal(go, rv["sort_gids"])
    m = len(so)
    bins = oracle.get_tile_bin_edges(m, so)
    assert np.array_equal(bins, rv["bins"])


@pytest.mark.parametrize("tag", ["chol", "cov", "rs"])
def test_projection_forward_equals_reference_helpers(oracle, rv, tag):
    """Projection forward = parameter -> covariance, then the reference's compute_cov2d_bounds + get_tile_bbox."""
    h, w = (int(v) for v in rv["proj_hw"])
    tb = oracle.tile_bounds(h, w)
    n = len(rv[f"{tag}_radii"])
    if tag == "chol":
        p = oracle.project_gaussians_2d_forward(n, 3.0, rv["chol_means"], rv["chol_L"], h, w, tb, 0.01, 1.0)
    elif tag == "cov":
        p = oracle.project_gaussians_2d_covariance_forward(n, 3.0, rv["cov_means"], rv["cov_cov"], h, w, tb, 0.01, 1.0)
    else:
        p = oracle.project_gaussians_2d_scale_rot_forward(n, 3.0, rv["rs_means"], rv["rs_scales"], rv["rs_rot"], h, w, tb,
                                                          0.01, 1.0)
    xys, depths, radii, conics, nth = p
    # the kernels leave a gaussian with an empty tile box at radius > 0 / num_tiles_hit = 0, like the helpers
    bad = int((radii != rv[f"{tag}_radii"]).sum()) + int((nth != rv[f"{tag}_nth"]).sum())
    # the scale-rotation covariance goes through libm sin/cos here and torch's there: radii may sit on a ceil() edge
    assert bad <= (2 if tag == "rs" else 0), f"{tag}: {bad} radii / num_tiles_hit differ from the reference helpers"
    ok = (radii == rv[f"{tag}_radii"]) & (radii > 0)
    np.testing.assert_allclose(xys[ok], rv[f"{tag}_xys"][ok], rtol=1e-6, atol=1e-5)
    scale = np.abs(rv[f"{tag}_conics"][ok]).max(-1, keepdims=True)
    check_close(f"{tag} conics", conics[ok], rv[f"{tag}_conics"][ok], scale, rtol=1e-5 if tag == "rs" else 2e-6)


@pytest.mark.parametrize("tag", ["chol", "cov", "rs"])
def test_projection_backward_equals_autograd_with_the_documented_double_count(oracle, rv, tag):
    h, w = (int(v) for v in rv["proj_hw"])
    n = len(rv[f"{tag}_radii"])
    radii, conics = np.ones(n, np.int32), conic64(rv, tag).astype(np.float32)
    v_xy, v_conic = rv["proj_v_xy"], rv["proj_v_conic"]
    if tag == "chol":
        out = oracle.project_gaussians_2d_backward(n, rv["chol_means"], rv["chol_L"], h, w, radii, conics, v_xy, None,
                                                   v_conic)
        names = ["v_cov2d", "v_mean2d", "v_L"]
    elif tag == "cov":
        out = oracle.project_gaussians_2d_covariance_backward(n, rv["cov_means"], rv["cov_cov"], h, w, radii, conics,
                                                              v_xy, None, v_conic)
        names = ["v_cov2d", "v_mean2d", "v_cov"]
    else:
        out = oracle.project_gaussians_2d_scale_rot_backward(n, rv["rs_means"], rv["rs_scales"], rv["rs_rot"], h, w,
                                                             radii, conics, v_xy, None, v_conic)
        names = ["v_cov2d", "v_mean2d", "v_scale", "v_rot"]
    for got, nm in zip(out, names):
        want = rv[f"{tag}_{nm}"]
        key = f"{tag}_{nm}_mag"  # parameters: sum of the absolute terms; v_cov2d / v_mean2d: the row's largest entry
        scale = rv[key] if key in rv else np.abs(want).max(-1, keepdims=True) + 1e-30
        check_close(f"{tag} {nm}", got.reshape(want.shape), want, scale, rtol=1e-5)


def conic64(rv, tag):
    """Inverse covariance in float64 from the stored parameters (the backward kernels take the conic as an input)."""
    if tag == "chol":
        L = rv["chol_L"].astype(np.float64)
        c = np.stack([L[:, 0] ** 2, L[:, 0] * L[:, 1], L[:, 1] ** 2 + L[:, 2] ** 2], 1)
    elif tag == "cov":
        c = rv["cov_cov"].astype(np.float64)
    else:
        s, r = rv["rs_scales"].astype(np.float64), rv["rs_rot"].astype(np.float64)[:, 0]
        co, si = np.cos(r), np.sin(r)
        m00, m01, m10, m11 = co * s[:, 0], si * s[:, 1], -si * s[:, 0], co * s[:, 1]
        c = np.stack([m00 * m00 + m01 * m01, m00 * m10 + m01 * m11, m10 * m10 + m11 * m11], 1)
    det = c[:, 0] * c[:, 2] - c[:, 1] ** 2
    return np.stack([c[:, 2] / det, -c[:, 1] / det, c[:, 0] / det], 1)


def ras_lists(oracle, rv):
    h, w = (int(v) for v in rv["ras_hw"])
    tb = oracle.tile_bounds(h, w)
    n = len(rv["ras_radii"])
    m, cum = oracle.compute_cumulative_intersects(rv["ras_nth"])
    _, _, so, go, bins = oracle.bin_and_sort_gaussians(n, m, rv["ras_xys"], np.zeros(n, np.float32), rv["ras_radii"], cum,
                                                       tb, 1.0)
    return h, w, tb, n, go, bins


def test_rasterizer_lists_equal_reference_membership(oracle, rv):
    h, w, tb, n, go, bins = ras_lists(oracle, rv)
    member = np.zeros_like(rv["ras_member"])
    for t in range(tb[0] * tb[1]):
        member[t, go[bins[t, 0]:bins[t, 1]]] = True
    assert np.array_equal(member, rv["ras_member"])


def test_rasterizer_forward_and_backward_equal_float64_autograd(oracle, rv):
    """forward.cu:636-660 and backward.cu:1258-1300 against the float64 statement + autograd (opacity <= 1)."""
    h, w, tb, n, go, bins = ras_lists(oracle, rv)
    out, fT, fidx = oracle.rasterize_sum_forward(tb, (16, 16, 1), (w, h, 1), go, bins, rv["ras_xys"], rv["ras_conics"],
                                                 rv["ras_colors"], rv["ras_opacity"])
    okp = np.repeat(~rv["ras_pix_ambig"][..., None], 3, -1)
    assert okp.mean() > 0.99
    check_close("out_img", out, rv["ras_out_img"], rv["ras_abs_img"], mask=okp, rtol=1e-5)
    g = oracle.rasterize_sum_backward(h, w, 16, 16, go, bins, rv["ras_xys"], rv["ras_conics"], rv["ras_colors"],
                                      rv["ras_opacity"], None, fT, fidx, rv["ras_v_out"])
    okg = ~rv["ras_g_ambig"]
    assert okg.mean() > 0.9
    for got, nm in zip(g, ["v_xy", "v_conic", "v_rgb", "v_opacity"]):
        want, mag = rv[f"ras_{nm}"], rv[f"ras_mag_{nm[2:]}"]
        if nm == "v_conic":  # the kernels hand on HALF the gradient of the off-diagonal entry (backward.cu:952-955:
            got = got * np.array([1.0, 2.0, 1.0])  # 0.5 v_sigma dx dy); cov2d_to_conic_vjp doubles it back
        check_close(nm, got, want, mag, mask=np.repeat(okg[:, None], want.shape[1], 1), rtol=1e-5, atol=1e-12)


# ---- the reference's OWN CPU rasterizer (_torch_impl.py:354-421 `rasterize_forward`), one gaussian per call, summed;
# ---- gradients by autograd through the same calls (tests/golden/make_refras_vectors.py -> refras_vectors.npz)

@pytest.fixture(scope="module", params=["refras_vectors.npz", "refras_vectors_large.npz"])
def rr(golden_dir, request):
    """40x56 / 46 gaussians, and (round 4) 72x104 / 160 gaussians: tests/golden/make_refras_vectors.py [REFRAS_LARGE=1]"""
    return np.load(os.path.join(golden_dir, request.param))


def refras_lists(oracle, rr):
    h, w = (int(v) for v in rr["refras_hw"])
    tb = oracle.tile_bounds(h, w)
    n = len(rr["refras_radii"])
    m, cum = oracle.compute_cumulative_intersects(rr["refras_nth"])
    _, _, so, go, bins = oracle.bin_and_sort_gaussians(n, m, rr["refras_xys"], np.zeros(n, np.float32), rr["refras_radii"],
                                                       cum, tb, 1.0)
    return h, w, tb, n, go, bins


REFRAS_GRADS = ["v_xy", "v_conic", "v_rgb", "v_opacity"]


def check_refras_grads(tag, rr, grads, rtol=1e-5):
    """Held to BOTH runs of the reference function: float64 tensors (rounding removed) and float32 tensors (as it
    runs); no mask -- the scene has no pair within 1e-5 of a cut-off (refras_gap_alpha / refras_gap_sigma)."""
    worst = {}
    for got, nm in zip(grads, REFRAS_GRADS):
        mag = rr[f"refras_mag_{nm[2:]}"]
        got = np.asarray(got, np.float64).reshape(mag.shape)
        if nm == "v_conic":  # the kernels hand on HALF the gradient of the off-diagonal entry (backward.cu:952-955)
            got = got * np.array([1.0, 2.0, 1.0])
        for pre in ("refras", "refras32"):
            worst[f"{pre}_{nm}"] = check_close(f"{tag} {nm} vs {pre}", got, rr[f"{pre}_{nm}"], mag, rtol=rtol, atol=1e-12)
    return worst


def test_refras_scene_is_unambiguous_and_exercises_both_cutoffs(rr):
    assert float(rr["refras_gap_alpha"]) > 1e-5 and float(rr["refras_gap_sigma"]) > 1e-5
    # neither clamp binds (min(0.999, .) there, min(1, .) here): the generator asserts alpha <= 0.999 on every landing pair
    assert rr["refras_opacity"].max() <= np.float32(0.999)
    n = len(rr["refras_radii"])
    h, w = (int(v) for v in rr["refras_hw"])
    landed = np.unpackbits(rr["refras_landed"])[:n * h * w].reshape(n, h, w).astype(bool)
    assert int(landed.sum()) == int(rr["refras_pairs_landing"]) > 2000
    det = rr["refras_conics"][:, 0] * rr["refras_conics"][:, 2] - rr["refras_conics"][:, 1] ** 2
    assert (det < 0).sum() == 2 and landed[det < 0].any()  # non positive definite conics that still land pairs


def test_lists_equal_the_reference_helpers_membership(oracle, rr):
    h, w, tb, n, go, bins = refras_lists(oracle, rr)
    member = np.zeros_like(rr["refras_member"])
    for t in range(tb[0] * tb[1]):
        member[t, go[bins[t, 0]:bins[t, 1]]] = True
    assert np.array_equal(member, rr["refras_member"])


def test_oracle_rasterizer_equals_the_reference_cpu_rasterizer(oracle, rr):
    """oracle/gi2d_oracle.c forward (forward.cu:570-691) and backward (backward.cu:1168-1350) against the image and the
    gradients `_torch_impl.rasterize_forward` + autograd produced, at 1e-5 of the summed absolute terms, every element."""
    h, w, tb, n, go, bins = refras_lists(oracle, rr)
    out, fT, fidx = oracle.rasterize_sum_forward(tb, (16, 16, 1), (w, h, 1), go, bins, rr["refras_xys"],
                                                 rr["refras_conics"], rr["refras_colors"], rr["refras_opacity"])
    for pre in ("refras", "refras32"):
        check_close(f"out_img vs {pre}", out, rr[f"{pre}_out_img"], rr["refras_abs_img"], rtol=1e-5)
    g = oracle.rasterize_sum_backward(h, w, 16, 16, go, bins, rr["refras_xys"], rr["refras_conics"], rr["refras_colors"],
                                      rr["refras_opacity"], None, fT, fidx, rr["refras_v_out"])
    print(check_refras_grads("oracle", rr, g[:4]))
