#!/usr/bin/env python3
"""Vectors that do NOT come from this repo's oracle (run in the DEV container only; writes ref_vectors.npz).

The reference holds no fixture for the 2D path, but two things in it can produce numbers on a CPU:

  A. its own Python helpers, gsplat/gsplat/_torch_impl.py -- compute_cov2d_bounds (:197), get_tile_bbox (:236),
     map_gaussian_to_intersects (:297), get_tile_bin_edges (:328) -- the CPU side of the reference's own tests
     (gsplat/tests/test_cov2d_bounds.py:9-35, test_map_gaussians.py:9-73, test_get_tile_bin_edges.py:9-81).  They are
     imported and run here on seeded inputs; inputs AND outputs are stored, so tests compare against the reference's
     arrays, not against a record of a past comparison;

  B. the arithmetic the kernels define, differentiated by torch autograd in float64:
       * rasterizer: out[p] = sum_g colour_g * alpha_pg over the gaussians g of p's tile list,
         alpha = min(1, opacity * exp(-sigma)), sigma = 0.5 (a dx^2 + c dy^2) + b dx dy, dx = x_g - j, dy = y_g - i,
         pairs with sigma < 0 or alpha < 1/255 skipped (forward.cu:636-660).  The tile lists come from A.  With
         opacity <= 1 the min() never binds, so autograd of <out, v_out> is exactly what backward.cu:1258-1300 must
         produce (its v_sigma ignores the clamp, which only matters for opacity > 1): an independent derivation of the
         backward kernel's formulas;
       * projections: conic = inverse covariance (helpers.cuh:188-193) as a function of the covariance triple, its VJP
         by autograd, then the chain to the parameters: exact for the covariance model (backward2d.cu:194-196), and
         with the off-diagonal gradient counted twice for the Cholesky / scale-rotation models, which is what
         backward2d.cu:39-41,94-96 compute -- expected = J^T (G11, 2 G12, G22), J by autograd.

Nothing from /root/reference is copied into the repo; only arrays are stored.
"""
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/gsplat"


def import_torch_impl():
    jt = types.ModuleType("jaxtyping")  # annotation-only third-party package, absent here

    class _Ann:
        def __class_getitem__(cls, item):
            return cls

    jt.Float = jt.Int = jt.Bool = _Ann
    sys.modules.setdefault("jaxtyping", jt)
    sys.path.insert(0, REF)
    import gsplat._torch_impl as ti
    return ti


def helpers_part(ti, out):
    rng = np.random.default_rng(20260401)
    # --- compute_cov2d_bounds: the reference test's recipe (random 2x2, made symmetric PSD), plus spread-out scales
    a = rng.normal(size=(600, 2, 2)).astype(np.float32) * rng.uniform(0.3, 6.0, size=(600, 1, 1)).astype(np.float32)
    cov = (a @ a.transpose(0, 2, 1) + 0.05 * np.eye(2, dtype=np.float32)).astype(np.float32)
    conic, radius, valid = ti.compute_cov2d_bounds(torch.from_numpy(cov))
    out.update(cov2d_in=cov, cov2d_conic=conic.numpy(), cov2d_radius=radius.numpy(), cov2d_valid=valid.numpy())
    # --- get_tile_bbox: centres inside, on the border and outside of a ragged 150x96 image, integer-valued radii
    h, w = 96, 150
    tb = ((w + 15) // 16, (h + 15) // 16, 1)
    n = 500
    centre = (rng.random((n, 2)) * np.array([w + 60, h + 60]) - 30).astype(np.float32)
    centre[:8] = [[0, 0], [15.999, 16.0], [16.0, 15.999], [w, h], [-0.5, 3.0], [w - 0.01, 0.0], [31.5, 47.5], [1e6, -1e6]]
    rad = rng.integers(1, 40, n).astype(np.float32)
    # The python helper computes the exclusive maximum as int(c + r) + 1, the kernels as (int)(c + r + 1)
    # (helpers.cuh:26-29): identical except for -1 < c + r < 0 in tile units, where the helper reports one tile and
    # the kernel none.  The kernels are what this build reproduces, so the vectors stay where the two agree.
    lo = -rad[:, None] + 0.25
    centre[8:] = np.where((centre[8:] + rad[8:, None] < 0) & (centre[8:] + rad[8:, None] > -16), lo[8:], centre[8:])
    tmin, tmax = ti.get_tile_bbox(torch.from_numpy(centre), torch.from_numpy(rad), tb)
    out.update(bbox_hw=np.array([h, w]), bbox_centre=centre, bbox_radius=rad, bbox_min=tmin.numpy(), bbox_max=tmax.numpy())
    # --- map_gaussian_to_intersects + get_tile_bin_edges: survivors only (the python loop breaks at radii <= 0),
    # non-constant depths so the low key bits are exercised
    area = ((tmax[:, 0] - tmin[:, 0]) * (tmax[:, 1] - tmin[:, 1])).numpy().astype(np.int32)
    keep = area > 0
    xs, rs, ns = centre[keep], rad[keep].astype(np.int32), area[keep]
    ds = rng.uniform(0.1, 9.0, len(rs)).astype(np.float32)
    cum = np.cumsum(ns).astype(np.int32)
    isect, gids = ti.map_gaussian_to_intersects(len(rs), torch.from_numpy(xs), torch.from_numpy(ds),
                                                torch.from_numpy(rs), torch.from_numpy(cum), tb)
    srt, perm = torch.sort(isect, stable=True)
    assert int(srt[-1] >> 32) == int(srt[-2] >> 32), "pick a seed whose last tile holds two entries (see :339-341)"
    bins = ti.get_tile_bin_edges(int(cum[-1]), srt)
    out.update(map_xys=xs, map_depths=ds, map_radii=rs, map_cum=cum, map_isect=isect.numpy(), map_gids=gids.numpy(),
               sort_isect=srt.numpy(), sort_gids=gids[perm].numpy(), bins=bins.numpy())


def conic_of(cov3):
    det = cov3[:, 0] * cov3[:, 2] - cov3[:, 1] ** 2
    return torch.stack([cov3[:, 2] / det, -cov3[:, 1] / det, cov3[:, 0] / det], 1)


def projection_part(ti, out):
    """Forward through the reference's helpers (fp32), VJP by float64 autograd."""
    rng = np.random.default_rng(77)
    h, w = 64, 96
    tb = ((w + 15) // 16, (h + 15) // 16, 1)
    n = 300
    v_conic = rng.normal(size=(n, 3))
    v_xy = rng.normal(size=(n, 2))
    out.update(proj_hw=np.array([h, w]), proj_v_conic=v_conic.astype(np.float32), proj_v_xy=v_xy.astype(np.float32))

    def forward(tag, centre32, cov32):
        cov2 = torch.stack([torch.stack([cov32[:, 0], cov32[:, 1]], 1), torch.stack([cov32[:, 1], cov32[:, 2]], 1)], 1)
        conic, radius, _ = ti.compute_cov2d_bounds(cov2)
        tmin, tmax = ti.get_tile_bbox(centre32, radius, tb)
        nth = ((tmax[:, 0] - tmin[:, 0]) * (tmax[:, 1] - tmin[:, 1])).to(torch.int32)
        out.update({f"{tag}_xys": centre32.numpy(), f"{tag}_conics": conic.numpy(),
                    f"{tag}_radii": radius.to(torch.int32).numpy(), f"{tag}_nth": nth.numpy()})

    def vjp(tag, params64, cov_of, double_offdiag, names):
        ps = [p.clone().requires_grad_(True) for p in params64]
        cov = cov_of(*ps)
        covd = cov.detach().requires_grad_(True)
        # v_conic follows the kernels' convention: its middle entry is the gradient PER off-diagonal matrix entry (the
        # rasterizer backward emits 0.5 v_sigma dx dy, backward.cu:952-955, and cov2d_to_conic_vjp builds the symmetric
        # G = [[v0, v1], [v1, v2]], helpers.cuh:388-391), i.e. the loss is v0 X00 + v1 (X01 + X10) + v2 X11
        (conic_of(covd) * torch.from_numpy(v_conic * np.array([1.0, 2.0, 1.0]))).sum().backward()
        g = covd.grad.clone()  # (G11, G12 = vS01 + vS10, G22): the reference's v_cov2d (helpers.cuh:392-394)
        out[f"{tag}_v_cov2d"] = g.numpy().astype(np.float32)
        if double_offdiag:
            g[:, 1] *= 2  # backward2d.cu:39-41,94-96
        cov.backward(g, retain_graph=True)
        for p, nm in zip(ps, names):
            out[f"{tag}_{nm}"] = p.grad.numpy().astype(np.float32)
            p.grad = None
        # size of the terms each result is the sum of (|dSigma_k/dp * g_k| over k): the yardstick for a relative error
        # ... where g_k itself is a sum of products X G X (helpers.cuh:392-394) whose terms can cancel: |X| |G| |X|
        X = conic_of(covd).detach()
        Xm = torch.stack([torch.stack([X[:, 0], X[:, 1]], 1), torch.stack([X[:, 1], X[:, 2]], 1)], 1).abs()
        vc = torch.from_numpy(v_conic).abs()
        Gm = torch.stack([torch.stack([vc[:, 0], vc[:, 1]], 1), torch.stack([vc[:, 1], vc[:, 2]], 1)], 1)
        S = Xm @ Gm @ Xm
        g_mag = torch.stack([S[:, 0, 0], S[:, 0, 1] + S[:, 1, 0], S[:, 1, 1]], 1)
        out[f"{tag}_v_cov2d_mag"] = g_mag.numpy()
        if double_offdiag:
            g_mag[:, 1] *= 2
        mags = [torch.zeros_like(p) for p in ps]
        for k in range(3):
            onehot = torch.zeros_like(g)
            onehot[:, k] = g_mag[:, k]
            cov.backward(onehot, retain_graph=True)
            for p, m in zip(ps, mags):
                m += p.grad.abs()
                p.grad = None
        for m, nm in zip(mags, names):
            out[f"{tag}_{nm}_mag"] = m.detach().numpy()

    # Cholesky model: means in (-1, 1), centre = 0.5 W x + 0.5 W (foward2d.cu:41-42), Sigma = L L^T (:48)
    xy = np.clip(2 * (rng.random((n, 2)) - 0.5), -0.999, 0.999).astype(np.float32)
    L = (rng.random((n, 3)) + np.array([1.2, 0, 1.2])).astype(np.float32)
    out.update(chol_means=xy, chol_L=L)
    xt, Lt = torch.from_numpy(xy), torch.from_numpy(L)
    wh = torch.tensor([w, h], dtype=torch.float32)
    forward("chol", 0.5 * wh * xt + 0.5 * wh,
            torch.stack([Lt[:, 0] * Lt[:, 0], Lt[:, 0] * Lt[:, 1], Lt[:, 1] * Lt[:, 1] + Lt[:, 2] * Lt[:, 2]], 1))
    vjp("chol", [Lt.double()],
        lambda l: torch.stack([l[:, 0] ** 2, l[:, 0] * l[:, 1], l[:, 1] ** 2 + l[:, 2] ** 2], 1), True, ["v_L"])
    out["chol_v_mean2d"] = (v_xy * 0.5 * np.array([w, h])).astype(np.float32)  # backward2d.cu:48-49
    # covariance model: means in pixels, Sigma given (foward2d.cu:226,236); exact gradient (backward2d.cu:194-196)
    mean_px = (rng.random((n, 2)) * np.array([w, h])).astype(np.float32)
    cv = (rng.random((n, 3)) * np.array([1, 0.6, 1]) + np.array([3, -0.3, 3])).astype(np.float32)
    out.update(cov_means=mean_px, cov_cov=cv)
    forward("cov", torch.from_numpy(mean_px), torch.from_numpy(cv))
    vjp("cov", [torch.from_numpy(cv).double()], lambda c: c, False, ["v_cov"])
    out["cov_v_mean2d"] = v_xy.astype(np.float32)  # backward2d.cu:205-206
    # scale-rotation model: Sigma = M M^T, M = R S (foward2d.cu:158-164), with R[0][1] = -sin, R[1][0] = sin in glm's
    # [column][row] indexing (helpers.cuh:587-598), i.e. R = [[cos, sin], [-sin, cos]] as a row/column matrix
    sc = (np.abs(rng.random((n, 2)) + 0.5) * 2.0).astype(np.float32)
    rot = (rng.random((n, 1)) * 2 * math.pi).astype(np.float32)
    out.update(rs_means=mean_px, rs_scales=sc, rs_rot=rot)

    def rs_cov(s, r):
        c, si = torch.cos(r[:, 0]), torch.sin(r[:, 0])
        m00, m01, m10, m11 = c * s[:, 0], si * s[:, 1], -si * s[:, 0], c * s[:, 1]  # M = R diag(sx, sy)
        return torch.stack([m00 * m00 + m01 * m01, m00 * m10 + m01 * m11, m10 * m10 + m11 * m11], 1)

    forward("rs", torch.from_numpy(mean_px), rs_cov(torch.from_numpy(sc), torch.from_numpy(rot)).float())
    vjp("rs", [torch.from_numpy(sc).double(), torch.from_numpy(rot).double()], rs_cov, True, ["v_scale", "v_rot"])
    out["rs_v_mean2d"] = v_xy.astype(np.float32)


def rasterizer_part(ti, out):
    rng = np.random.default_rng(4242)
    h, w = 64, 90
    tx, ty = (w + 15) // 16, (h + 15) // 16
    tb = (tx, ty, 1)
    n = 260
    centre = (rng.random((n, 2)) * np.array([w + 8, h + 8]) - 4).astype(np.float32)
    a = rng.normal(size=(n, 2, 2)).astype(np.float32) * rng.uniform(0.8, 2.2, size=(n, 1, 1)).astype(np.float32)
    cov = (a @ a.transpose(0, 2, 1) + 0.6 * np.eye(2, dtype=np.float32)).astype(np.float32)
    conic, radius, _ = ti.compute_cov2d_bounds(torch.from_numpy(cov))
    radii = radius.to(torch.int32)
    tmin, tmax = ti.get_tile_bbox(torch.from_numpy(centre), radius, tb)
    member = np.zeros((ty * tx, n), bool)
    for g in range(n):
        for i in range(int(tmin[g, 1]), int(tmax[g, 1])):
            for j in range(int(tmin[g, 0]), int(tmax[g, 0])):
                member[i * tx + j, g] = True
    assert member.sum(1).max() <= 256, "stay below the 256-entries-per-tile cap (forward.cu:553)"
    nth = ((tmax[:, 0] - tmin[:, 0]) * (tmax[:, 1] - tmin[:, 1])).to(torch.int32).numpy()
    colour = rng.random((n, 3)).astype(np.float32)
    opac = rng.uniform(0.2, 1.0, (n, 1)).astype(np.float32)  # <= 1: the clamp never binds
    v_out = rng.normal(size=(h, w, 3)).astype(np.float32) / (h * w)

    xy_t = torch.from_numpy(centre).double().requires_grad_(True)
    co_t = conic.double().clone().requires_grad_(True)
    cl_t = torch.from_numpy(colour).double().requires_grad_(True)
    op_t = torch.from_numpy(opac).double().requires_grad_(True)
    jj, ii = torch.meshgrid(torch.arange(w, dtype=torch.float64), torch.arange(h, dtype=torch.float64), indexing="xy")
    dx = xy_t[:, 0][None, None, :] - jj[..., None]      # [H, W, N]
    dy = xy_t[:, 1][None, None, :] - ii[..., None]
    sigma = 0.5 * (co_t[:, 0] * dx * dx + co_t[:, 2] * dy * dy) + co_t[:, 1] * dx * dy
    alpha = op_t[:, 0] * torch.exp(-sigma)
    tile_of = (ii.long() // 16) * tx + (jj.long() // 16)
    in_list = torch.from_numpy(member)[tile_of]          # [H, W, N]
    lands = in_list & (sigma >= 0) & (alpha >= 1.0 / 255.0)
    assert float(alpha[lands].max()) <= 1.0
    contrib = torch.where(lands, alpha, torch.zeros_like(alpha))
    img = torch.einsum("hwn,nc->hwc", contrib, cl_t)
    (img * torch.from_numpy(v_out).double()).sum().backward()
    # pairs a float32 evaluation may put on the other side of a cut-off, and what they touch
    near = in_list & (((alpha - 1.0 / 255.0).abs() < 1e-6) | (sigma.abs() < 1e-6))
    absimg = torch.einsum("hwn,nc->hwc", contrib.detach().abs(), cl_t.detach().abs())
    out.update(ras_hw=np.array([h, w]), ras_xys=centre, ras_conics=conic.numpy(), ras_radii=radii.numpy(), ras_nth=nth,
               ras_colors=colour, ras_opacity=opac, ras_v_out=v_out, ras_member=member,
               ras_out_img=img.detach().numpy(), ras_abs_img=absimg.numpy(),
               ras_v_xy=xy_t.grad.numpy(), ras_v_conic=co_t.grad.numpy(), ras_v_rgb=cl_t.grad.numpy(),
               ras_v_opacity=op_t.grad.numpy(), ras_pix_ambig=near.any(2).numpy(), ras_g_ambig=near.any(0).any(0).numpy(),
               ras_pairs_landing=np.array(int(lands.sum())))
    # magnitude of every gradient component (sum of absolute per-pixel terms) for relative tolerances
    xy_a = torch.from_numpy(centre).double().requires_grad_(True)
    with torch.no_grad():
        va = torch.einsum("hwc,nc->hwn", torch.from_numpy(v_out).double(), cl_t.detach())  # v_alpha per pair
        wgt = contrib.detach() * va.abs()
        mag_xy = torch.stack([(wgt * (co_t[:, 0] * dx + co_t[:, 1] * dy).abs()).sum((0, 1)),
                              (wgt * (co_t[:, 1] * dx + co_t[:, 2] * dy).abs()).sum((0, 1))], 1)
        mag_conic = torch.stack([(0.5 * wgt * dx * dx).sum((0, 1)), (wgt * (dx * dy).abs()).sum((0, 1)),
                                 (0.5 * wgt * dy * dy).sum((0, 1))], 1)
        mag_rgb = torch.einsum("hwn,hwc->nc", contrib.detach(), torch.from_numpy(v_out).double().abs())
        mag_op = (torch.exp(-sigma) * lands * va.abs()).sum((0, 1))[:, None]
    out.update(ras_mag_xy=mag_xy.numpy(), ras_mag_conic=mag_conic.numpy(), ras_mag_rgb=mag_rgb.numpy(),
               ras_mag_opacity=mag_op.numpy())


def main():
    ti = import_torch_impl()
    out = {}
    helpers_part(ti, out)
    projection_part(ti, out)
    rasterizer_part(ti, out)
    np.savez_compressed(os.path.join(HERE, "ref_vectors.npz"), **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == "__main__":
    main()
