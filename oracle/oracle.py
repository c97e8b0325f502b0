"""numpy/ctypes front end of the CPU oracle (oracle/gi2d_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.

Parity status: pinned by arrays the reference's own Python layer produced in the dev container -- its helpers for
projection / binning (tests/golden/ref_vectors.npz) and its CPU rasterizer `_torch_impl.rasterize_forward` + autograd for
the tile rasterizer forward and backward (tests/golden/refras_vectors.npz) -- not by a fixture of the reference's tests
(it has none for this path); see the header of gi2d_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgi2d_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle shared library with gcc (idempotent)."""
    src = os.path.join(_HERE, "gi2d_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libgi2d_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.gi2d_oracle_cumsum.restype = C.c_int
        _lib.gi2d_oracle_num_threads.restype = C.c_int
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def num_threads() -> int:
    return int(lib().gi2d_oracle_num_threads())


def set_num_threads(t: int) -> None:
    lib().gi2d_oracle_set_num_threads(C.c_int(int(t)))


def tile_bounds(img_h: int, img_w: int, block: int = 16):
    return ((img_w + block - 1) // block, (img_h + block - 1) // block, 1)


# ----------------------------------------------------------------------------- projection
def compute_cov2d_bounds(cov2d, clip_coe=3.0):
    cov2d = _f(cov2d)
    n = cov2d.shape[0]
    conics = np.zeros((n, 3), np.float32)
    radii = np.zeros((n, 1), np.float32)
    lib().gi2d_oracle_compute_cov2d_bounds(C.c_int(n), C.c_float(clip_coe), _p(cov2d), _p(conics),
                                           _p(radii))
    return conics, radii


def _proj_out(n):
    return (np.zeros((n, 2), np.float32), np.zeros((n,), np.float32), np.zeros((n,), np.int32),
            np.zeros((n, 3), np.float32), np.zeros((n,), np.int32))


def project_gaussians_2d_forward(num_points, clip_coe, means2d, L, img_h, img_w, tb, clip_thresh=0.01,
                                 radius_clip=1.0, isprint=False):
    """Argument order of _C.project_gaussians_2d_forward (bindings.cu:1317-1381)."""
    means2d, L = _f(means2d), _f(L)
    xys, depths, radii, conics, nth = _proj_out(num_points)
    lib().gi2d_oracle_project_cholesky_fwd(
        C.c_int(num_points), C.c_float(clip_coe), _p(means2d), _p(L), C.c_int(img_h), C.c_int(img_w),
        C.c_int(tb[0]), C.c_int(tb[1]), C.c_float(radius_clip), _p(xys), _p(depths), _p(radii),
        _p(conics), _p(nth))
    return xys, depths, radii, conics, nth


def project_gaussians_2d_covariance_forward(num_points, clip_coe, means2d, cov, img_h, img_w, tb,
                                            clip_thresh=0.01, radius_clip=1.0, isprint=False):
    means2d, cov = _f(means2d), _f(cov)
    xys, depths, radii, conics, nth = _proj_out(num_points)
    lib().gi2d_oracle_project_covariance_fwd(
        C.c_int(num_points), C.c_float(clip_coe), _p(means2d), _p(cov), C.c_int(img_h), C.c_int(img_w),
        C.c_int(tb[0]), C.c_int(tb[1]), C.c_float(radius_clip), _p(xys), _p(depths), _p(radii),
        _p(conics), _p(nth))
    return xys, depths, radii, conics, nth


def project_gaussians_2d_scale_rot_forward(num_points, clip_coe, means2d, scales, rot, img_h, img_w, tb,
                                           clip_thresh=0.01, radius_clip=1.0, isprint=False):
    means2d, scales, rot = _f(means2d), _f(scales), _f(rot)
    xys, depths, radii, conics, nth = _proj_out(num_points)
    lib().gi2d_oracle_project_scale_rot_fwd(
        C.c_int(num_points), C.c_float(clip_coe), _p(means2d), _p(scales), _p(rot), C.c_int(img_h),
        C.c_int(img_w), C.c_int(tb[0]), C.c_int(tb[1]), C.c_float(radius_clip), _p(xys), _p(depths),
        _p(radii), _p(conics), _p(nth))
    return xys, depths, radii, conics, nth


def project_gaussians_2d_backward(num_points, means2d, L, img_h, img_w, radii, conics, v_xy, v_depth,
                                  v_conic):
    """-> (v_cov2d, v_mean2d, v_L)   (bindings.cu:1517-1564)"""
    L, radii, conics, v_xy, v_conic = _f(L), _i(radii), _f(conics), _f(v_xy), _f(v_conic)
    n = num_points
    v_cov2d, v_mean, v_L = np.zeros((n, 3), np.float32), np.zeros((n, 2), np.float32), np.zeros((n, 3), np.float32)
    lib().gi2d_oracle_project_cholesky_bwd(C.c_int(n), _p(L), C.c_int(img_h), C.c_int(img_w), _p(radii),
                                           _p(conics), _p(v_xy), _p(v_conic), _p(v_cov2d), _p(v_mean),
                                           _p(v_L))
    return v_cov2d, v_mean, v_L


def project_gaussians_2d_covariance_backward(num_points, means2d, cov, img_h, img_w, radii, conics, v_xy,
                                             v_depth, v_conic):
    radii, conics, v_xy, v_conic = _i(radii), _f(conics), _f(v_xy), _f(v_conic)
    n = num_points
    v_cov2d, v_mean, v_cov = np.zeros((n, 3), np.float32), np.zeros((n, 2), np.float32), np.zeros((n, 3), np.float32)
    lib().gi2d_oracle_project_covariance_bwd(C.c_int(n), _p(radii), _p(conics), _p(v_xy), _p(v_conic),
                                             _p(v_cov2d), _p(v_mean), _p(v_cov))
    return v_cov2d, v_mean, v_cov


def project_gaussians_2d_scale_rot_backward(num_points, means2d, scales, rot, img_h, img_w, radii, conics,
                                            v_xy, v_depth, v_conic):
    """-> (v_cov2d, v_mean2d, v_scale, v_rot[N,1])   (bindings.cu:1614-1668)"""
    scales, rot = _f(scales), _f(rot)
    radii, conics, v_xy, v_conic = _i(radii), _f(conics), _f(v_xy), _f(v_conic)
    n = num_points
    v_cov2d, v_mean = np.zeros((n, 3), np.float32), np.zeros((n, 2), np.float32)
    v_scale, v_rot = np.zeros((n, 2), np.float32), np.zeros((n, 1), np.float32)
    lib().gi2d_oracle_project_scale_rot_bwd(C.c_int(n), _p(scales), _p(rot), _p(radii), _p(conics),
                                            _p(v_xy), _p(v_conic), _p(v_cov2d), _p(v_mean), _p(v_scale),
                                            _p(v_rot))
    return v_cov2d, v_mean, v_scale, v_rot


# ----------------------------------------------------------------------------- binning
def compute_cumulative_intersects(num_tiles_hit):
    nth = _i(num_tiles_hit)
    cum = np.zeros_like(nth)
    m = lib().gi2d_oracle_cumsum(C.c_int(nth.shape[0]), _p(nth), _p(cum))
    return int(m), cum


def map_gaussian_to_intersects(num_points, num_intersects, xys, depths, radii, cum_tiles_hit, tb,
                               radius_clip=1.0, isprint=False):
    xys, depths, radii, cum = _f(xys), _f(depths), _i(radii), _i(cum_tiles_hit)
    isect = np.zeros((num_intersects,), np.int64)
    gids = np.zeros((num_intersects,), np.int32)
    lib().gi2d_oracle_map_gaussian_to_intersects(
        C.c_int(num_points), C.c_int(num_intersects), _p(xys), _p(depths), _p(radii), _p(cum),
        C.c_int(tb[0]), C.c_int(tb[1]), C.c_float(radius_clip), _p(isect), _p(gids))
    return isect, gids


def sort_intersects(isect_ids, gaussian_ids):
    isect = np.ascontiguousarray(isect_ids, np.int64)
    gids = _i(gaussian_ids)
    so, go = np.zeros_like(isect), np.zeros_like(gids)
    lib().gi2d_oracle_sort_intersects(C.c_int(isect.shape[0]), _p(isect), _p(gids), _p(so), _p(go))
    return so, go


def get_tile_bin_edges(num_intersects, isect_ids_sorted, rows=None):
    isect = np.ascontiguousarray(isect_ids_sorted, np.int64)
    rows = num_intersects if rows is None else rows
    bins = np.zeros((rows, 2), np.int32)
    lib().gi2d_oracle_get_tile_bin_edges(C.c_int(num_intersects), _p(isect), C.c_int(rows), _p(bins))
    return bins


def bin_and_sort_gaussians(num_points, num_intersects, xys, depths, radii, cum_tiles_hit, tb,
                           radius_clip=1.0, isprint=False):
    """gsplat/gsplat/utils.py:253-311; tile_bins gets max(M, T) rows."""
    isect, gids = map_gaussian_to_intersects(num_points, num_intersects, xys, depths, radii, cum_tiles_hit,
                                             tb, radius_clip)
    so, go = sort_intersects(isect, gids)
    bins = get_tile_bin_edges(num_intersects, so, rows=max(num_intersects, tb[0] * tb[1]))
    return isect, gids, so, go, bins


# ----------------------------------------------------------------------------- rasterizer
def rasterize_sum_forward(tb, block, img_size, gaussian_ids_sorted, tile_bins, xys, conics, colors,
                          opacities, background=None, isprint=False, with_aux=False):
    """Argument order of _C.rasterize_sum[_plus]_forward (bindings.cu:453-610).
    -> (out_img[H,W,3], final_Ts[H,W], final_idx[H,W]) (+ ambig[H,W], abs_img[H,W,3])"""
    img_w, img_h = img_size[0], img_size[1]
    gids, bins = _i(gaussian_ids_sorted), _i(tile_bins)
    xys, conics, colors, opac = _f(xys), _f(conics), _f(colors), _f(opacities)
    out = np.zeros((img_h, img_w, 3), np.float32)
    fT = np.zeros((img_h, img_w), np.float32)
    fidx = np.zeros((img_h, img_w), np.int32)
    amb = np.zeros((img_h, img_w), np.uint8) if with_aux else None
    absimg = np.zeros((img_h, img_w, 3), np.float32) if with_aux else None
    lib().gi2d_oracle_rasterize_forward_sum(
        C.c_int(tb[0]), C.c_int(tb[1]), C.c_int(img_w), C.c_int(img_h), _p(gids), _p(bins),
        C.c_int(bins.shape[0]), _p(xys), _p(conics), _p(colors), _p(opac), _p(fT), _p(fidx), _p(out),
        _p(amb), _p(absimg))
    if with_aux:
        return out, fT, fidx, amb, absimg
    return out, fT, fidx


def rasterize_sum_backward(img_h, img_w, block_h, block_w, gaussian_ids_sorted, tile_bins, xys, conics,
                           colors, opacities, background, final_Ts, final_idx, v_output,
                           v_output_alpha=None, with_aux=False, with_amb9=False):
    """Argument order of _C.rasterize_sum[_plus]_backward (bindings.cu:1166-1314).
    -> (v_xy, v_conic, v_colors, v_opacity[N,1]) (+ ambig[N], abs9[N,9], v_abs_xy[N,4]) (+ amb9[N,9]: what the pairs
    flagged as ambiguous add to each gaussian at most -- the bound for the gaussians the mask sets aside)"""
    assert block_h == 16 and block_w == 16
    gids, bins = _i(gaussian_ids_sorted), _i(tile_bins)
    xys, conics, colors, opac = _f(xys), _f(conics), _f(colors), _f(opacities)
    fidx, vout = _i(final_idx), _f(v_output)
    n = xys.shape[0]
    tb = tile_bounds(img_h, img_w)
    v_xy, v_conic = np.zeros((n, 2), np.float32), np.zeros((n, 3), np.float32)
    v_rgb, v_op = np.zeros((n, 3), np.float32), np.zeros((n, 1), np.float32)
    amb = np.zeros((n,), np.uint8) if with_aux else None
    abs9 = np.zeros((n, 9), np.float32) if with_aux else None
    vabs = np.zeros((n, 4), np.float32) if with_aux else None
    amb9 = np.zeros((n, 9), np.float32) if (with_aux and with_amb9) else None
    lib().gi2d_oracle_rasterize_backward_sum_ex(
        C.c_int(n), C.c_int(tb[0]), C.c_int(tb[1]), C.c_int(img_w), C.c_int(img_h), _p(gids), _p(bins),
        C.c_int(bins.shape[0]), _p(xys), _p(conics), _p(colors), _p(opac), _p(fidx), _p(vout), _p(v_xy),
        _p(v_conic), _p(v_rgb), _p(v_op), _p(amb), _p(abs9), _p(vabs), _p(amb9))
    if with_aux and with_amb9:
        return v_xy, v_conic, v_rgb, v_op, amb, abs9, vabs, amb9
    if with_aux:
        return v_xy, v_conic, v_rgb, v_op, amb, abs9, vabs
    return v_xy, v_conic, v_rgb, v_op


# ----------------------------------------------------------------------------- whole path
def render_cholesky(means2d, L, colors, opacities, img_h, img_w, radius_clip=1.0, with_aux=False):
    """project (Cholesky) -> bin -> rasterize; mirrors models/gaussianimage_cholesky.py:206-219
    (inputs are the already-activated tanh(xyz) and cholesky+bound)."""
    tb = tile_bounds(img_h, img_w)
    n = means2d.shape[0]
    xys, depths, radii, conics, nth = project_gaussians_2d_forward(n, 3.0, means2d, L, img_h, img_w, tb,
                                                                   0.01, radius_clip)
    m, cum = compute_cumulative_intersects(nth)
    _, _, so, go, bins = bin_and_sort_gaussians(n, m, xys, depths, radii, cum, tb, 1.0)
    ras = rasterize_sum_forward(tb, (16, 16, 1), (img_w, img_h, 1), go, bins, xys, conics, colors,
                                opacities, with_aux=with_aux)
    return dict(xys=xys, depths=depths, radii=radii, conics=conics, num_tiles_hit=nth, cum=cum, M=m,
                isect_sorted=so, gids_sorted=go, tile_bins=bins, ras=ras)
