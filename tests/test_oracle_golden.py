"""CPU-only: the oracle against the committed golden vectors, the reference cross-check record and
hand-derived known answers."""
import glob
import json
import os

import numpy as np
import pytest


def cases(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, "case_*.npz")))


def test_cholesky_backward_known_answer(oracle):
    """SURVEY fact 4, hand-derived (tests/golden/make_golden.py::cholesky_known_answer): L = (2, 1, 3), v_conic =
    (1, 0, 0) -> v_cov2d = (-100, 40, -4)/1296 and, with the off-diagonal counted twice as backward2d.cu:39-41 does,
    v_L = (-320, 152, -24)/1296 (the true gradient would be (-360, 72, -24)/1296).  The comparisons of the oracle's
    helpers with the reference's own Python code are in tests/test_ref_vectors_cpu.py, against stored arrays."""
    L = np.array([[2., 1., 3.]], np.float32)
    xy = np.zeros((1, 2), np.float32)
    tb = oracle.tile_bounds(64, 64)
    xys, depths, radii, conics, nth = oracle.project_gaussians_2d_forward(1, 3.0, xy, L, 64, 64, tb, 0.01, 1.0)
    np.testing.assert_allclose(conics[0], np.array([10., -2., 4.]) / 36., rtol=1e-6)
    v_cov2d, v_mean, v_L = oracle.project_gaussians_2d_backward(
        1, xy, L, 64, 64, radii, conics, np.array([[0.25, -0.5]], np.float32), None, np.array([[1., 0., 0.]], np.float32))
    np.testing.assert_allclose(v_L[0], np.array([-320., 152., -24.]) / 1296., rtol=1e-6)
    np.testing.assert_allclose(v_cov2d[0], np.array([-100., 40., -4.]) / 1296., rtol=1e-6)
    np.testing.assert_allclose(v_mean[0], [8.0, -16.0], rtol=1e-7)


def test_call_shape_record(golden_dir):
    """Argument kinds the reference's own autograd Functions hand to each `_C` op (recorded by driving them
    on CPU): our op table must accept exactly these positional lists."""
    rec = json.load(open(os.path.join(golden_dir, "call_shapes.json")))
    ops = rec["ops"]
    assert [k.split(":")[0] for k in ops["project_gaussians_2d_forward"]] == \
        ["int", "float", "tensor", "tensor", "int", "int", "tuple", "float", "float", "bool"]
    assert len(ops["project_gaussians_2d_scale_rot_forward"]) == 11
    assert len(ops["project_gaussians_2d_backward"]) == 10 and len(ops["project_gaussians_2d_scale_rot_backward"]) == 11
    assert len(ops["map_gaussian_to_intersects"]) == 9 and len(ops["get_tile_bin_edges"]) == 2
    assert len(ops["rasterize_sum_plus_forward"]) == 11 and len(ops["rasterize_sum_plus_backward"]) == 15
    assert len(rec["project_gaussians_2d.returns"]) == 5 and rec["scale_rot_grad_shapes"]["rot"][1] == 1
    import inspect
    import gaussianimage_plus_amd.gsplat.cuda as C
    for name, kinds in ops.items():
        # the ctypes table (the binding's fallback) ...
        sig = inspect.signature(C.CTYPES_TABLE.get(name, getattr(C, name)))
        positional = [p for p in sig.parameters.values() if p.default is inspect.Parameter.empty]
        assert len(positional) <= len(kinds) <= len(sig.parameters), name
        # ... and the compiled table (csrc/torch_ext), whose signature pybind11 writes into the docstring
        fn = getattr(C, name)
        if C.BINDING == "compiled" and name in C.CTYPES_TABLE:
            head = fn.__doc__.splitlines()[0]
            args = head[head.index("(") + 1:head.rindex(") ->")]
            params = [a for a in _split_top_level(args) if a.strip()]
            required = [a for a in params if "=" not in a]
            assert len(required) <= len(kinds) <= len(params), (name, head)


def _split_top_level(text):
    """Split a pybind11 signature's argument list at the commas that are not inside brackets."""
    out, depth, cur = [], 0, ""
    for ch in text:
        depth += ch in "[(" 
        depth -= ch in "])"
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    return out + [cur]


def test_oracle_reproduces_golden_vectors(oracle, golden_dir):
    for path in cases(golden_dir):
        g = np.load(path)
        n, h, w, kind = int(g["n"]), int(g["h"]), int(g["w"]), str(g["kind"])
        tb = oracle.tile_bounds(h, w)
        if kind == "cholesky":
            p = oracle.project_gaussians_2d_forward(n, 3.0, g["in_means"], g["in_L"], h, w, tb, 0.01, 1.0)
        elif kind == "covariance":
            p = oracle.project_gaussians_2d_covariance_forward(n, float(g["clip_coe"]), g["in_means"], g["in_L"], h, w, tb,
                                                               0.01, float(g["radius_clip"]))
        else:
            p = oracle.project_gaussians_2d_scale_rot_forward(n, 3.0, g["in_means"], g["in_scales"], g["in_rot"], h, w,
                                                              tb, 0.01, 1.0)
        for got, key in zip(p, ["xys", "depths", "radii", "conics", "num_tiles_hit"]):
            assert np.array_equal(got, g[key]), (path, key)
        m, cum = oracle.compute_cumulative_intersects(p[4])
        assert m == int(g["M"]) and np.array_equal(cum, g["cum_tiles_hit"])
        rclip = float(g["radius_clip"]) if "radius_clip" in g else 1.0
        isect, gids, so, go, bins = oracle.bin_and_sort_gaussians(n, m, p[0], p[1], p[2], cum, tb, rclip)
        for got, key in zip((isect, gids, so, go, bins), ["isect_ids", "gaussian_ids", "isect_sorted", "gids_sorted", "tile_bins"]):
            assert np.array_equal(got, g[key]), (path, key)
        out, fT, fidx = oracle.rasterize_sum_forward(tb, (16, 16, 1), (w, h, 1), go, bins, p[0], p[3], g["colors"], g["opacity"])
        assert np.array_equal(out, g["out_img"]) and np.array_equal(fidx, g["final_idx"]) and np.all(fT == 1)
        v = oracle.rasterize_sum_backward(h, w, 16, 16, go, bins, p[0], p[3], g["colors"], g["opacity"], None, fT, fidx, g["v_out"])
        for got, key in zip(v, ["v_xy", "v_conic", "v_rgb", "v_opacity"]):
            np.testing.assert_allclose(got, g[key], rtol=1e-6, atol=1e-12, err_msg=f"{path} {key}")


def test_golden_set_covers_the_edge_cases(golden_dir):
    info = {c["name"]: c for c in json.load(open(os.path.join(golden_dir, "cases.json")))}
    assert info["chol_crowded"]["max_per_tile"] > 256          # the 256-entry cap (forward.cu:553)
    assert info["chol_ragged"]["w"] % 16 and info["chol_ragged"]["h"] % 16  # ragged right / bottom tiles
    g = np.load(os.path.join(golden_dir, "case_chol_degenerate.npz"))
    assert g["radii"][0] == 0                                    # det == 0 -> culled (helpers.cuh:188)
    # a tiny covariance is NOT culled by the minor-axis test: max(0.1, b^2-det) makes v2 negative, its sqrt
    # is NaN and `NaN < radius_clip` is false (helpers.cuh:198-203, foward2d.cu:55) -- reference behaviour
    assert g["radii"][1] > 0 and g["num_tiles_hit"][1] > 0
    assert g["num_tiles_hit"][2] == g["tile_bins"].shape[0] or g["num_tiles_hit"][2] >= 9  # huge gaussian
    assert g["num_tiles_hit"][3] == 0 and g["radii"][3] > 0      # bbox outside the image
    assert g["v_rgb"][5].tolist() == [0, 0, 0]                   # opacity 0 never passes alpha >= 1/255


def test_cap_semantics_of_the_oracle(oracle, golden_dir):
    """Entries past the first 256 of a tile contribute nothing forward and get zero gradient."""
    g = np.load(os.path.join(golden_dir, "case_chol_crowded.npz"))
    bins, go = g["tile_bins"], g["gids_sorted"]
    lens = bins[:, 1] - bins[:, 0]
    assert lens.max() > 256
    within = set()
    for t in np.nonzero(lens)[0]:
        within.update(go[bins[t, 0]: min(bins[t, 1], bins[t, 0] + 256)].tolist())
    never_rasterized = sorted(set(go.tolist()) - within)
    assert len(never_rasterized) > 0 and np.all(g["v_rgb"][never_rasterized] == 0)
    assert np.any(g["v_rgb"][sorted(within)] != 0)
    assert g["final_idx"].max() <= (bins[:, 0] + 255).max()


def test_forward_is_linear_in_colour_and_additive(oracle):
    rng = np.random.default_rng(0)
    n, h, w = 300, 48, 64
    xyz = ((rng.random((n, 2)) - 0.5) * 1.9).astype(np.float32)
    L = (rng.random((n, 3)) + np.array([1.0, 0, 1.0])).astype(np.float32)
    col = rng.random((n, 3)).astype(np.float32)
    op = np.ones((n, 1), np.float32)
    a = oracle.render_cholesky(xyz, L, col, op, h, w)["ras"][0]
    b = oracle.render_cholesky(xyz, L, 2 * col, op, h, w)["ras"][0]
    np.testing.assert_allclose(b, 2 * a, rtol=1e-6, atol=1e-7)
    # splitting the gaussians in two sets and adding the images reproduces the full image
    half = n // 2
    i1 = oracle.render_cholesky(xyz[:half], L[:half], col[:half], op[:half], h, w)["ras"][0]
    i2 = oracle.render_cholesky(xyz[half:], L[half:], col[half:], op[half:], h, w)["ras"][0]
    np.testing.assert_allclose(i1 + i2, a, rtol=1e-5, atol=1e-6)


def test_empty_and_single_inputs(oracle):
    tb = oracle.tile_bounds(20, 20)
    m, cum = oracle.compute_cumulative_intersects(np.zeros(0, np.int32))
    assert m == 0 and cum.shape == (0,)
    bins = oracle.get_tile_bin_edges(0, np.zeros(0, np.int64), rows=4)
    assert bins.shape == (4, 2) and not bins.any()
    out, fT, fidx = oracle.rasterize_sum_forward(tb, (16, 16, 1), (20, 20, 1), np.zeros(0, np.int32), bins,
                                                 np.zeros((0, 2), np.float32), np.zeros((0, 3), np.float32),
                                                 np.zeros((0, 3), np.float32), np.zeros((0, 1), np.float32))
    assert not out.any() and np.all(fT == 1) and not fidx.any()
    # one gaussian exactly on a pixel centre: alpha = opacity there (sigma == 0 passes `sigma < 0` as false)
    xys, depths, radii, conics, nth = oracle.project_gaussians_2d_covariance_forward(
        1, 3.0, np.array([[8.0, 8.0]], np.float32), np.array([[4.0, 0.0, 4.0]], np.float32), 20, 20, tb, 0.01, 1.0)
    m, cum = oracle.compute_cumulative_intersects(nth)
    _, _, so, go, bins = oracle.bin_and_sort_gaussians(1, m, xys, depths, radii, cum, tb, 1.0)
    out, _, _ = oracle.rasterize_sum_forward(tb, (16, 16, 1), (20, 20, 1), go, bins, xys, conics,
                                             np.array([[0.5, 0.25, 1.0]], np.float32), np.array([[0.8]], np.float32))
    np.testing.assert_allclose(out[8, 8], [0.4, 0.2, 0.8], rtol=1e-6)
