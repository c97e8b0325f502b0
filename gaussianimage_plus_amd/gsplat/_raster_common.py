"""Core of the two additive-rasterizer autograd Functions (rasterize_sum.py / rasterize_sum_plus.py).

One forward = scan -> map -> native stable sort (+ tile bins + inverse permutation) -> tile rasterizer;
the binning "plan" (cum_tiles_hit, inv_perm) is saved so the backward can sum per-gaussian partials in
a fixed order without float atomics.  The only host round trip is the 4-byte intersection count the
reference API also reads (utils.py:249)."""
from __future__ import annotations

import torch

from . import cuda as _C

BLOCK = 16


def tile_bounds_of(img_height: int, img_width: int, block_h: int, block_w: int):
    return ((img_width + block_w - 1) // block_w, (img_height + block_h - 1) // block_h, 1)


def forward_impl(ctx, plus: bool, xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height,
                 img_width, BLOCK_H, BLOCK_W, background, radius_clip, isprint):
    num_points = xys.size(0)
    tile_bounds = tile_bounds_of(img_height, img_width, BLOCK_H, BLOCK_W)
    block = (BLOCK_W, BLOCK_H, 1)
    img_size = (img_width, img_height, 1)
    nth = num_tiles_hit if num_tiles_hit.dtype == torch.int32 else num_tiles_hit.to(torch.int32)
    if num_points > 0:
        cum_tiles_hit, total = _C.cumsum_tiles_hit(nth.contiguous())
        num_intersects = int(total.item())
    else:
        cum_tiles_hit, num_intersects = nth, 0

    cnt_gs_counts = None
    if num_intersects < 1:  # rasterize_sum_plus.py:110-118 / rasterize_sum.py:130-139
        out_img = torch.ones(img_height, img_width, colors.shape[-1], device=xys.device) * background
        gaussian_ids_sorted = torch.zeros(0, dtype=torch.int32, device=xys.device)
        tile_bins = torch.zeros(0, 2, dtype=torch.int32, device=xys.device)
        final_Ts = torch.zeros(img_height, img_width, device=xys.device)
        final_idx = torch.zeros(img_height, img_width, dtype=torch.int32, device=xys.device)
        inv_perm = torch.zeros(0, dtype=torch.int32, device=xys.device)
        if not plus:
            cnt_gs_counts = torch.zeros(img_height, img_width, dtype=torch.int32, device=xys.device)
    else:
        isect_ids, gaussian_ids = _C.map_gaussian_to_intersects(
            num_points, num_intersects, xys, depths, radii, cum_tiles_hit, tile_bounds, radius_clip, isprint)
        srt = _C.sort_intersects(isect_ids, gaussian_ids, tile_bounds[0] * tile_bounds[1],
                                 want_inv_perm=True, want_bins=True, want_keys=False)
        gaussian_ids_sorted, tile_bins, inv_perm = srt["gaussian_ids_sorted"], srt["tile_bins"], srt["inv_perm"]
        if plus:
            out_img, final_Ts, final_idx = _C.rasterize_sum_plus_forward(
                tile_bounds, block, img_size, gaussian_ids_sorted, tile_bins, xys, conics, colors, opacity,
                background, isprint)
        else:
            if colors.shape[-1] != 3:  # rasterize_sum.py:170-171 would pick nd_rasterize_sum_forward
                raise NotImplementedError("N-channel rasterization is outside this build (RGB only)")
            out_img, final_Ts, final_idx, cnt_gs_counts = _C.rasterize_sum_forward(
                tile_bounds, block, img_size, gaussian_ids_sorted, tile_bins, xys, conics, colors, opacity,
                background, isprint)

    ctx.img_width, ctx.img_height = img_width, img_height
    ctx.BLOCK_H, ctx.BLOCK_W = BLOCK_H, BLOCK_W
    ctx.num_intersects = num_intersects
    ctx.save_for_backward(gaussian_ids_sorted, tile_bins, xys, conics, colors, opacity, background, final_Ts,
                          final_idx, cum_tiles_hit, inv_perm)
    return out_img, final_Ts, cnt_gs_counts


def backward_impl(ctx, plus: bool, v_out_img):
    (gaussian_ids_sorted, tile_bins, xys, conics, colors, opacity, background, final_Ts, final_idx,
     cum_tiles_hit, inv_perm) = ctx.saved_tensors
    if ctx.num_intersects < 1:
        v_abs = None if plus else torch.zeros(xys.size(0), 4, device=xys.device)
        return (torch.zeros_like(xys), torch.zeros_like(conics), torch.zeros_like(colors),
                torch.zeros_like(opacity), v_abs)
    fn = _C.rasterize_sum_plus_backward if plus else _C.rasterize_sum_backward
    res = fn(ctx.img_height, ctx.img_width, ctx.BLOCK_H, ctx.BLOCK_W, gaussian_ids_sorted, tile_bins, xys,
             conics, colors, opacity, background, final_Ts, final_idx, v_out_img.contiguous(), None,
             cum_tiles_hit=cum_tiles_hit, inv_perm=inv_perm)
    v_xy, v_conic, v_colors, v_opacity = res[:4]
    return v_xy, v_conic, v_colors, v_opacity.view_as(opacity), (None if plus else res[4])
