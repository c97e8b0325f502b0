"""GPU: the HIP ops, through the `_C` op table, against tests/golden/ref_vectors.npz -- arrays produced by the
reference's own Python helpers and by float64 autograd of the kernels' formulas (tests/golden/make_ref_vectors.py).
No expected value here comes from this repo's oracle (tests/test_ref_vectors_cpu.py holds the oracle to the same file)."""
import os

import numpy as np
import pytest
import torch

from helpers import check_close
from test_ref_vectors_cpu import conic64, cov3_of

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def n(x):
    return x.detach().cpu().numpy()


@pytest.fixture(scope="module")
def C():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import gaussianimage_plus_amd.gsplat.cuda as _C
    from gaussianimage_plus_amd import _lib
    _lib.load()
    return _C


@pytest.fixture(scope="module")
def rv(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_vectors.npz"))


def test_compute_cov2d_bounds_equals_reference_arrays(C, rv):
    cov3 = cov3_of(rv["cov2d_in"])
    conic, radius = C.compute_cov2d_bounds(len(cov3), 3.0, t(cov3))
    np.testing.assert_allclose(n(conic), rv["cov2d_conic"], rtol=3e-7, atol=0)
    assert np.array_equal(n(radius)[:, 0], rv["cov2d_radius"])


def test_binning_equals_reference_arrays(C, rv):
    """map (_torch_impl.py:297), torch.sort(stable) + gather, bin edges (_torch_impl.py:328): bit-exact."""
    h, w = (int(v) for v in rv["bbox_hw"])
    tb = ((w + 15) // 16, (h + 15) // 16, 1)
    m = int(rv["map_cum"][-1])
    isect, gids = C.map_gaussian_to_intersects(len(rv["map_radii"]), m, t(rv["map_xys"]), t(rv["map_depths"]),
                                               t(rv["map_radii"]), t(rv["map_cum"]), tb, 1.0, False)
    assert np.array_equal(n(isect), rv["map_isect"]) and np.array_equal(n(gids), rv["map_gids"])
    srt = C.sort_intersects(isect, gids, tb[0] * tb[1])
    assert np.array_equal(n(srt["isect_ids_sorted"]), rv["sort_isect"])
    assert np.array_equal(n(srt["gaussian_ids_sorted"]), rv["sort_gids"])
    bins = C.get_tile_bin_edges(m, srt["isect_ids_sorted"])
    assert np.array_equal(n(bins), rv["bins"])
    # the fused sync-free binning (gi2d_bin_gaussians): same lists from centres + radii alone (depth == 0 there)
    gids_f, bins_f, status = C.bin_gaussians(t(rv["map_xys"]), t(rv["map_radii"]), tb, 1.0, m + 64)
    T = tb[0] * tb[1]
    go, tile_bins = n(gids_f), n(bins_f)
    assert n(status)[:2].tolist() == [m, 0]
    order = np.lexsort((rv["map_gids"], rv["map_isect"] >> 32))  # by tile, then ascending id
    assert np.array_equal(go[:m], rv["map_gids"][order])
    assert np.array_equal(tile_bins[:T], rv["bins"][:T])


@pytest.mark.parametrize("tag", ["chol", "cov", "rs"])
def test_projection_forward_equals_reference_helpers(C, rv, tag):
    h, w = (int(v) for v in rv["proj_hw"])
    tb = ((w + 15) // 16, (h + 15) // 16, 1)
    cnt = len(rv[f"{tag}_radii"])
    if tag == "chol":
        p = C.project_gaussians_2d_forward(cnt, 3.0, t(rv["chol_means"]), t(rv["chol_L"]), h, w, tb, 0.01, 1.0, False)
    elif tag == "cov":
        p = C.project_gaussians_2d_covariance_forward(cnt, 3.0, t(rv["cov_means"]), t(rv["cov_cov"]), h, w, tb, 0.01,
                                                      1.0, False)
    else:
        p = C.project_gaussians_2d_scale_rot_forward(cnt, 3.0, t(rv["rs_means"]), t(rv["rs_scales"]), t(rv["rs_rot"]),
                                                     h, w, tb, 0.01, 1.0, False)
    xys, depths, radii, conics, nth = (n(x) for x in p)
    bad = int((radii != rv[f"{tag}_radii"]).sum()) + int((nth != rv[f"{tag}_nth"]).sum())
    print(f"{tag}: {bad} of {2 * cnt} radii / num_tiles_hit differ from the reference helpers")
    # scale-rot goes through the device's sin/cos: a radius can sit on a ceil() edge; the others must be exact
    assert bad <= (2 if tag == "rs" else 0)
    ok = (radii == rv[f"{tag}_radii"]) & (radii > 0)
    np.testing.assert_allclose(xys[ok], rv[f"{tag}_xys"][ok], rtol=1e-6, atol=1e-5)
    scale = np.abs(rv[f"{tag}_conics"][ok]).max(-1, keepdims=True)
    check_close(f"{tag} conics", conics[ok], rv[f"{tag}_conics"][ok], scale, rtol=1e-5 if tag == "rs" else 2e-6)


@pytest.mark.parametrize("tag", ["chol", "cov", "rs"])
def test_projection_backward_equals_autograd_with_the_documented_double_count(C, rv, tag):
    h, w = (int(v) for v in rv["proj_hw"])
    cnt = len(rv[f"{tag}_radii"])
    radii, conics = t(np.ones(cnt, np.int32)), t(conic64(rv, tag).astype(np.float32))
    v_xy, v_conic = t(rv["proj_v_xy"]), t(rv["proj_v_conic"])
    if tag == "chol":
        out = C.project_gaussians_2d_backward(cnt, t(rv["chol_means"]), t(rv["chol_L"]), h, w, radii, conics, v_xy, None,
                                              v_conic)
        names = ["v_cov2d", "v_mean2d", "v_L"]
    elif tag == "cov":
        out = C.project_gaussians_2d_covariance_backward(cnt, t(rv["cov_means"]), t(rv["cov_cov"]), h, w, radii, conics,
                                                         v_xy, None, v_conic)
        names = ["v_cov2d", "v_mean2d", "v_cov"]
    else:
        out = C.project_gaussians_2d_scale_rot_backward(cnt, t(rv["rs_means"]), t(rv["rs_scales"]), t(rv["rs_rot"]), h, w,
                                                        radii, conics, v_xy, None, v_conic)
        names = ["v_cov2d", "v_mean2d", "v_scale", "v_rot"]
    for got, nm in zip(out, names):
        want = rv[f"{tag}_{nm}"]
        key = f"{tag}_{nm}_mag"
        scale = rv[key] if key in rv else np.abs(want).max(-1, keepdims=True) + 1e-30
        check_close(f"{tag} {nm}", n(got).reshape(want.shape), want, scale, rtol=1e-5)


def _lists(C, rv):
    h, w = (int(v) for v in rv["ras_hw"])
    tb = ((w + 15) // 16, (h + 15) // 16, 1)
    m = int(rv["ras_nth"].sum())
    gids, bins, status = C.bin_gaussians(t(rv["ras_xys"]), t(rv["ras_radii"]), tb, 1.0, m + 64)
    assert n(status)[:2].tolist() == [m, 0]
    # the list has exactly M entries for the ops that mirror the reference bindings (slots past M are scratch)
    return h, w, tb, gids[:m].contiguous(), bins


def test_rasterizer_forward_and_backward_equal_float64_autograd(C, rv):
    """forward.cu:636-660 / backward.cu:1258-1300 as float64 torch + autograd, against the plain ops AND the fused
    fast path (both forms of the product)."""
    h, w, tb, gids, bins = _lists(C, rv)
    member = np.zeros_like(rv["ras_member"])
    go, tbn = n(gids), n(bins)
    for tile in range(tb[0] * tb[1]):
        member[tile, go[tbn[tile, 0]:tbn[tile, 1]]] = True
    assert np.array_equal(member, rv["ras_member"])
    xys, conics, colors, opac = t(rv["ras_xys"]), t(rv["ras_conics"]), t(rv["ras_colors"]), t(rv["ras_opacity"])
    bg = torch.ones(3, device=DEV)
    out, fT, fidx = C.rasterize_sum_plus_forward(tb, (16, 16, 1), (w, h, 1), gids, bins, xys, conics, colors, opac, bg,
                                                 False)
    okp = np.repeat(~rv["ras_pix_ambig"][..., None], 3, -1)
    check_close("out_img", n(out), rv["ras_out_img"], rv["ras_abs_img"], mask=okp, rtol=1e-5)
    res = C.rasterize_sum_plus_backward(h, w, 16, 16, gids, bins, xys, conics, colors, opac, bg, fT, fidx,
                                        t(rv["ras_v_out"]), None)
    okg = ~rv["ras_g_ambig"]

    def check(tag, grads):
        for got, nm in zip(grads, ["v_xy", "v_conic", "v_rgb", "v_opacity"]):
            want, mag = rv[f"ras_{nm}"], rv[f"ras_mag_{nm[2:]}"]
            got = n(got).reshape(want.shape)
            if nm == "v_conic":  # the kernels pass on half the off-diagonal gradient (backward.cu:952-955)
                got = got * np.array([1.0, 2.0, 1.0])
            check_close(f"{tag} {nm}", got, want, mag, mask=np.repeat(okg[:, None], want.shape[1], 1), rtol=1e-5,
                        atol=1e-12)

    check("plain", res[:4])
    # fused fast path: bucket binning + fast forward / backward on the same gaussians
    ws = C.FastWorkspace(len(rv["ras_radii"]), tb, xys)
    radii = t(rv["ras_radii"])
    out_f = C.fast_forward(ws, xys, radii, conics, colors, opac, h, w, 1.0)
    assert n(ws.status)[1] == 0
    check_close("fast out_img", n(out_f), rv["ras_out_img"], rv["ras_abs_img"], mask=okp, rtol=1e-5)
    check("fast", C.fast_backward(ws, xys, radii, t(rv["ras_v_out"]), h, w, 1.0)[:4])


# ---- the reference's OWN CPU rasterizer (_torch_impl.py:354-421 `rasterize_forward`, one gaussian per call, summed;
# ---- gradients by autograd through the same calls): tests/golden/refras_vectors.npz

@pytest.fixture(scope="module", params=["refras_vectors.npz", "refras_vectors_large.npz"])
def rr(golden_dir, request):
    """40x56 / 46 gaussians, and (round 4) 72x104 / 160 gaussians: tests/golden/make_refras_vectors.py [REFRAS_LARGE=1]"""
    return np.load(os.path.join(golden_dir, request.param))


def test_hip_rasterizers_equal_the_reference_cpu_rasterizer(C, rr):
    """Every form of the product's tile rasterizer -- the ops that mirror the reference bindings
    (rasterize_sum_forward/backward, rasterize_sum_plus_forward/backward), the two-kernel fast path of the autograd
    wrappers and the single-pass forward+backward kernel that bench.py and the fitting loop run -- against the image and
    the gradients the reference's `rasterize_forward` + autograd produced: 1e-5 of the summed absolute terms, every
    element, no mask (the scene keeps 1e-5 clear of both cut-offs), against the float64-tensor run and the float32 one."""
    from gaussianimage_plus_amd import _lib
    from test_ref_vectors_cpu import check_refras_grads
    h, w = (int(v) for v in rr["refras_hw"])
    tb = ((w + 15) // 16, (h + 15) // 16, 1)
    cnt = len(rr["refras_radii"])
    m = int(rr["refras_nth"].sum())
    xys, conics, colors, opac = t(rr["refras_xys"]), t(rr["refras_conics"]), t(rr["refras_colors"]), t(rr["refras_opacity"])
    radii, v_out = t(rr["refras_radii"]), t(rr["refras_v_out"])
    gids, bins, status = C.bin_gaussians(xys, radii, tb, 1.0, m + 64)
    assert n(status)[:2].tolist() == [m, 0]
    gids = gids[:m].contiguous()
    member = np.zeros_like(rr["refras_member"])
    go, tbn = n(gids), n(bins)
    for tile in range(tb[0] * tb[1]):
        member[tile, go[tbn[tile, 0]:tbn[tile, 1]]] = True
    assert np.array_equal(member, rr["refras_member"])  # lists == the reference helpers' membership
    bg = torch.zeros(3, device=DEV)

    def check_img(tag, img):
        return {pre: check_close(f"{tag} out_img vs {pre}", n(img), rr[f"{pre}_out_img"], rr["refras_abs_img"], rtol=1e-5)
                for pre in ("refras", "refras32")}

    worst = {}
    # (1) the ops behind the reference's binding names
    out, fT, fidx = C.rasterize_sum_plus_forward(tb, (16, 16, 1), (w, h, 1), gids, bins, xys, conics, colors, opac, bg,
                                                 False)
    worst["plus img"] = check_img("plus", out)
    res = C.rasterize_sum_plus_backward(h, w, 16, 16, gids, bins, xys, conics, colors, opac, bg, fT, fidx, v_out, None)
    worst["plus"] = check_refras_grads("plus", rr, [n(g) for g in res[:4]])
    out_s = C.rasterize_sum_forward(tb, (16, 16, 1), (w, h, 1), gids, bins, xys, conics, colors, opac, bg, False)
    assert torch.equal(out_s[0], out)
    res_s = C.rasterize_sum_backward(h, w, 16, 16, gids, bins, xys, conics, colors, opac, bg, out_s[1], out_s[2], v_out,
                                     None)
    for a, b in zip(res_s[:4], res[:4]):
        assert torch.equal(a, b)
    # (2) the two-kernel fast path (what the autograd wrappers run)
    ws = C.FastWorkspace(cnt, tb, xys)
    out_f = C.fast_forward(ws, xys, radii, conics, colors, opac, h, w, 1.0)
    assert n(ws.status)[1] == 0
    worst["fast img"] = check_img("fast", out_f)
    worst["fast"] = check_refras_grads("fast", rr, [n(g) for g in C.fast_backward(ws, xys, radii, v_out, h, w, 1.0)[:4]])
    # (3) the single-pass kernel (fast_fwdbwd_kernel, the dominant kernel of bench.py) through the C ABI
    ws1 = C.FastWorkspace(cnt, tb, xys)
    out1 = torch.empty(h, w, 3, device=DEV)
    grads = [torch.empty(cnt, k, device=DEV) for k in (2, 3, 3, 1)]
    st = torch.cuda.current_stream().cuda_stream
    _lib.call("gi2d_fast_bin", cnt, xys.data_ptr(), radii.data_ptr(), conics.data_ptr(), colors.data_ptr(),
              opac.data_ptr(), tb[0], tb[1], 1.0, ws1.buf.data_ptr(), ws1.buf.numel(), ws1.status.data_ptr(), st)
    _lib.call("gi2d_fast_rasterize_forward_backward", cnt, tb[0], tb[1], w, h, None, v_out.data_ptr(), None, 0.0, None,
              ws1.buf.data_ptr(), ws1.buf.numel(), ws1.status.data_ptr(), out1.data_ptr(), st)
    _lib.call("gi2d_fast_rasterize_backward_reduce", cnt, tb[0], tb[1], ws1.buf.data_ptr(), ws1.buf.numel(),
              grads[0].data_ptr(), grads[1].data_ptr(), grads[2].data_ptr(), grads[3].data_ptr(), None, st)
    torch.cuda.synchronize()
    assert n(ws1.status)[1] == 0
    worst["single-pass img"] = check_img("single-pass", out1)
    worst["single-pass"] = check_refras_grads("single-pass", rr, [n(g) for g in grads])
    print(worst)
