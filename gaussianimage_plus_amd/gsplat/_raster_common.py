"""Core of the two additive-rasterizer autograd Functions (rasterize_sum.py / rasterize_sum_plus.py).

Reference orchestration (rasterize_sum_plus.py:98-172): cumsum -> `.item()` (host sync in the middle of
the forward) -> map -> torch.sort -> gather -> bin edges -> rasterize.  Here one forward is
    gi2d_bin_gaussians (count -> scan -> fill -> per-tile order, no host involvement)
    -> gi2d_rasterize_sum[_plus]_forward
on buffers sized by a grow-only capacity remembered per (device, N, H, W).  The true intersection count
is read back only AFTER the rasterizer has been enqueued (the GPU never waits for the host); if it
exceeded the capacity -- first call, or a >25 % jump -- the capacity is enlarged and the forward redone,
so results are always exact.  The backward needs no index from the forward: the per-gaussian sum
re-derives each gaussian's tiles from (xys, radii) and sums its per-tile partials in ascending tile
order -- no float atomics, bitwise reproducible."""
from __future__ import annotations

import os

import torch

from . import cuda as _C

BLOCK = 16
_capacity = {}  # (device index, N, H, W) -> intersection capacity
# GI2D_DEFER_COUNT_CHECK=1: check the count of call k at call k+1 instead of at the end of call k
# (fully asynchronous forward; an overflow then raises instead of being repaired in place).
_DEFER = os.environ.get("GI2D_DEFER_COUNT_CHECK", "0") == "1"
_pending = {}


def tile_bounds_of(img_height: int, img_width: int, block_h: int, block_w: int):
    return ((img_width + block_w - 1) // block_w, (img_height + block_h - 1) // block_h, 1)


def _check_pending(key):
    st = _pending.pop(key, None)
    if st is not None:
        m, overflow = st[0][:2].tolist()
        _capacity[key] = max(_capacity.get(key, 0), int(1.25 * m) + 1024)
        if overflow:
            raise RuntimeError(f"gsplat: intersection capacity {st[1]} overflowed (M={m}) in the previous "
                               "rasterize call; unset GI2D_DEFER_COUNT_CHECK for self-repairing behaviour")


def forward_impl(ctx, plus: bool, xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height,
                 img_width, BLOCK_H, BLOCK_W, background, radius_clip, isprint):
    num_points = xys.size(0)
    tile_bounds = tile_bounds_of(img_height, img_width, BLOCK_H, BLOCK_W)
    num_tiles = tile_bounds[0] * tile_bounds[1]
    block = (BLOCK_W, BLOCK_H, 1)
    img_size = (img_width, img_height, 1)
    if not plus and colors.shape[-1] != 3:  # rasterize_sum.py:170-171 would pick nd_rasterize_sum_forward
        raise NotImplementedError("N-channel rasterization is outside this build (RGB only)")
    radii = radii if radii.dtype == torch.int32 else radii.to(torch.int32)
    key = (xys.device.index, num_points, img_height, img_width)
    if _DEFER:
        _check_pending(key)
    capacity = _capacity.get(key) or max(4 * num_points, num_tiles, 1024)
    fwd = _C.rasterize_sum_plus_forward if plus else _C.rasterize_sum_forward
    while True:
        gaussian_ids_sorted, tile_bins, status = _C.bin_gaussians(xys, radii, tile_bounds, radius_clip, capacity)
        res = fwd(tile_bounds, block, img_size, gaussian_ids_sorted, tile_bins, xys, conics, colors, opacity,
                  background, isprint, num_intersects_dev=status)
        if _DEFER:
            _pending[key] = (status, capacity)
            num_intersects = -1
            break
        num_intersects, overflow = status[:2].tolist()  # after everything is enqueued
        want = int(1.25 * num_intersects) + 1024
        if not overflow:
            if capacity > 2 * want + num_tiles:  # shrink a grossly oversized first guess
                _capacity[key] = want
            else:
                _capacity.setdefault(key, capacity)
            break
        capacity = _capacity[key] = want
    out_img, final_Ts, final_idx = res[:3]
    cnt_gs_counts = None if plus else res[3]

    ctx.img_width, ctx.img_height = img_width, img_height
    ctx.BLOCK_H, ctx.BLOCK_W = BLOCK_H, BLOCK_W
    ctx.num_intersects = num_intersects
    ctx.radius_clip = float(radius_clip)
    ctx.save_for_backward(gaussian_ids_sorted, tile_bins, xys, radii, conics, colors, opacity, final_idx)
    return out_img, final_Ts, cnt_gs_counts


def backward_impl(ctx, plus: bool, v_out_img):
    gaussian_ids_sorted, tile_bins, xys, radii, conics, colors, opacity, final_idx = ctx.saved_tensors
    if ctx.num_intersects == 0:  # rasterize_sum_plus.py:198-202
        v_abs = None if plus else torch.zeros(xys.size(0), 4, device=xys.device)
        return (torch.zeros_like(xys), torch.zeros_like(conics), torch.zeros_like(colors),
                torch.zeros_like(opacity), v_abs)
    v_xy, v_conic, v_colors, v_opacity, v_abs = _C.rasterize_backward_fast(
        ctx.img_height, ctx.img_width, gaussian_ids_sorted, tile_bins, xys, radii, conics, colors, opacity,
        final_idx, v_out_img.contiguous(), ctx.radius_clip, with_abs=not plus)
    return v_xy, v_conic, v_colors, v_opacity.view_as(opacity), v_abs
