#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ (run in the DEV container only).

What it does
  1. imports the reference's *Python* operator surface from /root/reference/gsplat (read-only;
     a 6-line `jaxtyping` stand-in is needed because that annotation-only package is absent) and
       a. cross-checks the oracle's helpers against the reference's own CPU code in
          gsplat/gsplat/_torch_impl.py (compute_cov2d_bounds :197, get_tile_bbox :236,
          map_gaussian_to_intersects :297, get_tile_bin_edges :328)        -> printed only; the
          reference's OUTPUT ARRAYS are committed by make_ref_vectors.py  -> ref_vectors.npz
       b. drives the reference's autograd Functions (_ProjectGaussians2d*, _RasterizeGaussiansSum)
          on CPU with `gsplat.cuda.<op>` pointed at the oracle, recording for every `_C` op the
          positional-argument kinds it receives and for every Function the arity / None pattern
          it returns                                                       -> call_shapes.json
  2. writes seeded input/output vectors of the oracle for small cases      -> case_*.npz
     (the reference has no CPU implementation and no fixture for the 2D projection or the sum
      rasterizer, so these vectors pin the oracle against regressions; what pins it against the
      reference is ref_vectors.npz).

Nothing from /root/reference is copied; only numbers and call shapes are stored.
"""
import json
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

REF = "/root/reference/gsplat"


def import_reference():
    jt = types.ModuleType("jaxtyping")

    class _Ann:
        def __class_getitem__(cls, item):
            return cls

    jt.Float = jt.Int = jt.Bool = _Ann
    sys.modules.setdefault("jaxtyping", jt)
    sys.path.insert(0, REF)
    import gsplat  # noqa
    import gsplat._torch_impl as ti
    import gsplat.cuda as rc
    return gsplat, ti, rc


# --------------------------------------------------------------------------------------------
def synth_cholesky(n, h, w, seed, slv=True):
    rng = np.random.default_rng(seed)
    xyz = np.tanh(np.arctanh(np.clip(2 * (rng.random((n, 2)) - 0.5), -0.999999, 0.999999))).astype(np.float32)
    lp = min(h * w / (9 * math.pi * n), 300) if slv else 0.5
    L = (rng.random((n, 3)) + np.array([lp, 0, lp])).astype(np.float32)
    col = rng.random((n, 3)).astype(np.float32)
    op = np.ones((n, 1), np.float32)
    return xyz, L, col, op


def crosscheck(ti):
    rep = {}
    rng = np.random.default_rng(7)
    # (a) compute_cov2d_bounds: PSD covariances with det > 1e-6 (torch impl clamps det, ours does not)
    A = rng.normal(size=(2000, 2, 2)).astype(np.float32) * 3
    cov = A @ A.transpose(0, 2, 1) + 0.05 * np.eye(2, dtype=np.float32)
    cov3 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 1, 1]], -1).astype(np.float32)
    conic_o, rad_o = O.compute_cov2d_bounds(cov3, 3.0)
    conic_t, rad_t, valid = ti.compute_cov2d_bounds(torch.from_numpy(cov))
    v = valid.numpy()
    rel = np.abs(conic_o[v] - conic_t.numpy()[v]) / (np.abs(conic_t.numpy()[v]) + 1e-12)
    rep["compute_cov2d_bounds"] = dict(n=int(v.sum()), conic_max_rel=float(rel.max()),
                                       radius_mismatch=int((rad_o[v, 0] != rad_t.numpy()[v]).sum()))
    # (b) get_tile_bbox via projection num_tiles_hit + map
    n, h, w = 400, 96, 150
    tb = O.tile_bounds(h, w)
    xyz, L, col, op = synth_cholesky(n, h, w, 11)
    xys, depths, radii, conics, nth = O.project_gaussians_2d_forward(n, 3.0, xyz, L, h, w, tb, 0.01, 1.0)
    tmin, tmax = ti.get_tile_bbox(torch.from_numpy(xys), torch.from_numpy(radii.astype(np.float32)), tb)
    area = ((tmax[:, 0] - tmin[:, 0]) * (tmax[:, 1] - tmin[:, 1])).numpy()
    keep = radii > 0
    rep["get_tile_bbox"] = dict(n=int(keep.sum()), num_tiles_hit_mismatch=int((area[keep] != nth[keep]).sum()))
    # (c) map_gaussian_to_intersects: torch impl `break`s at the first radii<=0, so feed survivors only
    xs, ds, rs, ns = xys[keep], depths[keep], radii[keep], nth[keep]
    m, cum = O.compute_cumulative_intersects(ns)
    isect_o, gid_o = O.map_gaussian_to_intersects(len(rs), m, xs, ds, rs, cum, tb, 1.0)
    isect_t, gid_t = ti.map_gaussian_to_intersects(len(rs), torch.from_numpy(xs), torch.from_numpy(ds),
                                                   torch.from_numpy(rs), torch.from_numpy(cum), tb)
    rep["map_gaussian_to_intersects"] = dict(m=int(m), isect_mismatch=int((isect_o != isect_t.numpy()).sum()),
                                             gid_mismatch=int((gid_o != gid_t.numpy()).sum()))
    # (d) get_tile_bin_edges: the torch loop `break`s on the last element before the boundary test, so
    # compare on a key list whose last tile has >= 2 entries (true here) -- all rows must agree.
    so, go = O.sort_intersects(isect_o, gid_o)
    srt, perm = torch.sort(torch.from_numpy(isect_o), stable=True)
    rep["sort_vs_torch_stable_sort"] = dict(key_mismatch=int((so != srt.numpy()).sum()),
                                            gid_mismatch=int((go != gid_o[perm.numpy()]).sum()))
    bins_o = O.get_tile_bin_edges(m, so)
    bins_t = ti.get_tile_bin_edges(m, torch.from_numpy(so)).numpy()
    last_two_same = bool((so[-1] >> 32) == (so[-2] >> 32))
    rep["get_tile_bin_edges"] = dict(m=int(m), last_tile_has_two=last_two_same,
                                     mismatch=int((bins_o != bins_t).sum()))
    # (e) SURVEY fact 4 known answer: the Cholesky backward double-counts the off-diagonal.
    rep["cholesky_bwd_known_answer"] = cholesky_known_answer()
    return rep


def cholesky_known_answer():
    """Hand-derivable sample: L=(2,1,3) -> Sigma=[[4,2],[2,10]], det=36, conic=(10/36,-2/36,4/36).
    With v_conic=(1,0,0): v_Sigma = -X G X with G=diag(1,0) -> -(x0 x0^T), x0 = first column of X
    = (10/36,-2/36): v_Sigma = -[[100,-20],[-20,4]]/1296, v_cov2d=(-100, +40, -4)/1296.
    Reference formula (backward2d.cu:39-41): vL = (2*l11*G11+2*G12*l21, 2*l11*G12+2*l21*G22, 2*l22*G22)
      = (2*2*(-100)+2*40*1, 2*2*40+2*1*(-4), 2*3*(-4))/1296 = (-320, 152, -24)/1296.
    True gradient (autograd) uses G12_true = 20/1296: (-360, 72, -24)/1296 -- differs, as SURVEY says."""
    L = np.array([[2., 1., 3.]], np.float32)
    xy = np.zeros((1, 2), np.float32)
    h = w = 64
    tb = O.tile_bounds(h, w)
    xys, depths, radii, conics, nth = O.project_gaussians_2d_forward(1, 3.0, xy, L, h, w, tb, 0.01, 1.0)
    v_conic = np.array([[1., 0., 0.]], np.float32)
    v_xy = np.array([[0.25, -0.5]], np.float32)
    v_cov2d, v_mean, v_L = O.project_gaussians_2d_backward(1, xy, L, h, w, radii, conics, v_xy, None, v_conic)
    exp_vL = np.array([-320., 152., -24.]) / 1296.
    exp_cov = np.array([-100., 40., -4.]) / 1296.
    Lt = torch.tensor(L[0], dtype=torch.float64, requires_grad=True)
    S = torch.stack([Lt[0] * Lt[0], Lt[0] * Lt[1], Lt[1] * Lt[1] + Lt[2] * Lt[2]])
    det = S[0] * S[2] - S[1] * S[1]
    (S[2] / det).backward()
    return dict(oracle_v_L=v_L[0].tolist(), expected_reference_v_L=exp_vL.tolist(),
                oracle_v_cov2d=v_cov2d[0].tolist(), expected_v_cov2d=exp_cov.tolist(),
                autograd_true_v_L=Lt.grad.tolist(), oracle_v_mean=v_mean[0].tolist(),
                expected_v_mean=[0.25 * 32, -0.5 * 32],
                max_abs_err_vs_reference_formula=float(np.abs(v_L[0] - exp_vL).max()))


# --------------------------------------------------------------------------------------------
def kind(a):
    if isinstance(a, torch.Tensor):
        return f"tensor:{str(a.dtype).replace('torch.', '')}:{list(a.shape)}"
    if isinstance(a, bool):
        return "bool"
    if isinstance(a, int):
        return "int"
    if isinstance(a, float):
        return "float"
    if isinstance(a, tuple):
        return "tuple:" + ",".join(type(x).__name__ for x in a)
    if a is None:
        return "None"
    return type(a).__name__


def call_shapes(gsplat, rc):
    """Run the reference wrappers on CPU with the oracle standing in for the CUDA ops."""
    rec = {}

    def t(x):
        return torch.from_numpy(np.ascontiguousarray(x))

    def n(x):
        return x.detach().cpu().numpy()

    def fake(name, fn):
        def f(*args):
            rec.setdefault("ops", {})[name] = [kind(a) for a in args]
            out = fn(*[n(a) if isinstance(a, torch.Tensor) else a for a in args])
            return tuple(t(o) for o in out) if isinstance(out, tuple) else t(out)
        setattr(rc, name, f)

    fake("project_gaussians_2d_forward", O.project_gaussians_2d_forward)
    fake("project_gaussians_2d_backward", O.project_gaussians_2d_backward)
    fake("project_gaussians_2d_covariance_forward", O.project_gaussians_2d_covariance_forward)
    fake("project_gaussians_2d_covariance_backward", O.project_gaussians_2d_covariance_backward)
    fake("project_gaussians_2d_scale_rot_forward", O.project_gaussians_2d_scale_rot_forward)
    fake("project_gaussians_2d_scale_rot_backward", O.project_gaussians_2d_scale_rot_backward)
    fake("map_gaussian_to_intersects", O.map_gaussian_to_intersects)
    fake("get_tile_bin_edges", lambda m, ids: O.get_tile_bin_edges(m, ids, rows=max(m, 4096)))
    fake("rasterize_sum_plus_forward", O.rasterize_sum_forward)
    fake("rasterize_sum_plus_backward", O.rasterize_sum_backward)

    h, w, npts = 70, 100, 300
    tb = O.tile_bounds(h, w)
    xyz, L, col, op = synth_cholesky(npts, h, w, 5)
    out = {}

    # covariance model path (the wired one): models/gaussianimage_covariance.py:194-208
    rng = np.random.default_rng(9)
    mean_px = (rng.random((npts, 2)) * np.array([w, h])).astype(np.float32)
    cov = (rng.random((npts, 3)) * np.array([1, 0.3, 1]) + np.array([6, 0, 6])).astype(np.float32)
    m_t = t(mean_px).requires_grad_(True)
    c_t = t(cov).requires_grad_(True)
    col_t = t(col).requires_grad_(True)
    op_t = t(op).requires_grad_(True)
    res = gsplat.project_gaussians_2d_covariance(m_t, c_t, h, w, tb, clip_coe=3.0, radius_clip=1.0)
    rec["project_gaussians_2d_covariance.returns"] = [kind(r) for r in res]
    xys, depths, radii, conics, nth = res
    img = gsplat.rasterize_gaussians_plus(xys, depths, radii, conics, nth, col_t, op_t, h, w, 16, 16,
                                          background=torch.ones(3), radius_clip=1.0)
    rec["rasterize_gaussians_plus.returns"] = kind(img)
    gt = torch.full_like(img, 0.5)
    loss = ((img - gt) ** 2).mean()
    loss.backward()
    rec["grads_present"] = dict(means=m_t.grad is not None, cov=c_t.grad is not None,
                                colors=col_t.grad is not None, opacity=op_t.grad is not None)
    out["cov_path"] = dict(mean_px=mean_px, cov=cov, col=col, op=op, img=n(img), g_mean=n(m_t.grad),
                           g_cov=n(c_t.grad), g_col=n(col_t.grad), g_op=n(op_t.grad))

    # cholesky + scale_rot projections (current 5-return form)
    x_t, L_t = t(xyz).requires_grad_(True), t(L).requires_grad_(True)
    res = gsplat.project_gaussians_2d(x_t, L_t, h, w, tb)
    rec["project_gaussians_2d.returns"] = [kind(r) for r in res]
    (res[0].sum() + (res[3] * torch.arange(3.)).sum()).backward()
    sc = (np.abs(rng.random((npts, 2)) + 0.5) * 3).astype(np.float32)
    rot = (rng.random((npts, 1)) * 6.28).astype(np.float32)
    s_t, r_t, m2 = t(sc).requires_grad_(True), t(rot).requires_grad_(True), t(mean_px).requires_grad_(True)
    res = gsplat.project_gaussians_2d_scale_rot(m2, s_t, r_t, h, w, tb)
    rec["project_gaussians_2d_scale_rot.returns"] = [kind(r) for r in res]
    (res[0].sum() + (res[3] * torch.arange(3.)).sum()).backward()
    rec["scale_rot_grad_shapes"] = dict(scale=list(s_t.grad.shape), rot=list(r_t.grad.shape))
    return rec, out


# --------------------------------------------------------------------------------------------
def golden_case(name, n, h, w, seed, kind_="cholesky", mutate=None):
    """Full-path vectors from the oracle."""
    tb = O.tile_bounds(h, w)
    rng = np.random.default_rng(seed + 1000)
    xyz, L, col, op = synth_cholesky(n, h, w, seed)
    d = dict(n=n, h=h, w=w, kind=kind_)
    if kind_ == "cholesky":
        if mutate:
            xyz, L, col, op = mutate(xyz, L, col, op)
        p = O.project_gaussians_2d_forward(n, 3.0, xyz, L, h, w, tb, 0.01, 1.0)
        d.update(in_means=xyz, in_L=L)
    elif kind_ == "covariance":
        mean_px = (rng.random((n, 2)) * np.array([w, h])).astype(np.float32)
        cov = (rng.random((n, 3)) * np.array([1, 0.5, 1]) + np.array([5, -0.25, 5])).astype(np.float32)
        p = O.project_gaussians_2d_covariance_forward(n, 2.5, mean_px, cov, h, w, tb, 0.01, 2.0)
        d.update(in_means=mean_px, in_L=cov, clip_coe=2.5, radius_clip=2.0)
    else:
        mean_px = (rng.random((n, 2)) * np.array([w, h])).astype(np.float32)
        sc = (np.abs(rng.random((n, 2)) + 0.5) * 2.5).astype(np.float32)
        rot = (1 / (1 + np.exp(-rng.random((n, 1)))) * 2 * math.pi).astype(np.float32)
        p = O.project_gaussians_2d_scale_rot_forward(n, 3.0, mean_px, sc, rot, h, w, tb, 0.01, 1.0)
        d.update(in_means=mean_px, in_scales=sc, in_rot=rot)
    xys, depths, radii, conics, nth = p
    m, cum = O.compute_cumulative_intersects(nth)
    rclip = d.get("radius_clip", 1.0)
    isect, gids, so, go, bins = O.bin_and_sort_gaussians(n, m, xys, depths, radii, cum, tb, rclip)
    out, fT, fidx, amb, absimg = O.rasterize_sum_forward(tb, (16, 16, 1), (w, h, 1), go, bins, xys, conics, col,
                                                         op, with_aux=True)
    gt = rng.random((h, w, 3)).astype(np.float32)
    v_out = (2 * (np.clip(out, 0, 1) - gt) / (3 * h * w)).astype(np.float32)
    v_xy, v_conic, v_rgb, v_op, gamb, abs9, vabs = O.rasterize_sum_backward(h, w, 16, 16, go, bins, xys, conics, col, op,
                                                                      None, fT, fidx, v_out, with_aux=True)
    if kind_ == "cholesky":
        pb = O.project_gaussians_2d_backward(n, d["in_means"], d["in_L"], h, w, radii, conics, v_xy, None, v_conic)
        d.update(v_cov2d=pb[0], v_mean2d=pb[1], v_L=pb[2])
    elif kind_ == "covariance":
        pb = O.project_gaussians_2d_covariance_backward(n, d["in_means"], d["in_L"], h, w, radii, conics, v_xy,
                                                        None, v_conic)
        d.update(v_cov2d=pb[0], v_mean2d=pb[1], v_L=pb[2])
    else:
        pb = O.project_gaussians_2d_scale_rot_backward(n, d["in_means"], d["in_scales"], d["in_rot"], h, w, radii,
                                                       conics, v_xy, None, v_conic)
        d.update(v_cov2d=pb[0], v_mean2d=pb[1], v_scale=pb[2], v_rot=pb[3])
    d.update(colors=col, opacity=op, xys=xys, depths=depths, radii=radii, conics=conics, num_tiles_hit=nth,
             cum_tiles_hit=cum, M=m, isect_ids=isect, gaussian_ids=gids, isect_sorted=so, gids_sorted=go,
             tile_bins=bins, out_img=out, final_Ts=fT, final_idx=fidx, pix_ambig=amb, pix_abs=absimg,
             v_out=v_out, v_xy=v_xy, v_conic=v_conic, v_rgb=v_rgb, v_opacity=v_op, g_ambig=gamb, g_abs9=abs9, v_abs_xy=vabs)
    np.savez_compressed(os.path.join(HERE, f"case_{name}.npz"), **d)
    return dict(name=name, n=n, h=h, w=w, M=int(m), max_per_tile=int((bins[:, 1] - bins[:, 0]).max()))


def main():
    gsplat, ti, rc = import_reference()
    # the comparison with the reference's own helpers is no longer stored as a report: make_ref_vectors.py commits
    # the reference's output arrays and tests/test_ref_vectors_*.py compare against those
    print(json.dumps(crosscheck(ti), indent=1))
    rec, out = call_shapes(gsplat, rc)
    with open(os.path.join(HERE, "call_shapes.json"), "w") as f:
        json.dump(rec, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "refwrap_cov_path.npz"), **out["cov_path"])
    print(json.dumps(rec, indent=1))

    def degenerate(xyz, L, col, op):
        L = L.copy()
        xyz = xyz.copy()
        L[0] = [0, 0, 0]            # det == 0 -> culled (helpers.cuh:188)
        L[1] = [0.05, 0, 0.05]      # minor radius < radius_clip -> culled (foward2d.cu:55)
        L[2] = [40, 5, 40]          # huge: covers every tile
        xyz[3] = [5.0, 5.0]         # far outside the image: bbox area 0
        xyz[4] = [-0.999, -0.999]   # corner
        op = op.copy()
        op[5] = 0.0                 # alpha always below 1/255
        op[6] = 3.0                 # alpha clamps to 1 near the centre
        return xyz, L, col, op

    def crowded(xyz, L, col, op):
        # > 256 gaussians in one tile: exercises the 256-entry cap (forward.cu:553)
        xyz = (xyz * 0.12).astype(np.float32)
        return xyz, L, col, op

    cases = [
        golden_case("chol_small", 192, 48, 64, 1),
        golden_case("chol_ragged", 500, 70, 100, 2),
        golden_case("chol_degenerate", 64, 40, 56, 3, mutate=degenerate),
        golden_case("chol_crowded", 700, 32, 48, 4, mutate=crowded),
        golden_case("cov_ragged", 400, 90, 75, 5, kind_="covariance"),
        golden_case("rs_small", 300, 64, 96, 6, kind_="scale_rot"),
    ]
    with open(os.path.join(HERE, "cases.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print(cases)


if __name__ == "__main__":
    main()
