"""`rasterize_gaussians_sum` (reference: gsplat/gsplat/rasterize_sum.py) -- the call shape the
Cholesky / RS model files use (with the pass-through `screenspace_points`)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor
from torch.autograd import Function

from ._raster_common import backward_impl, forward_impl


def rasterize_gaussians_sum(xys: Tensor, screenspace_points: Tensor, depths: Tensor, radii: Tensor,
                            conics: Tensor, num_tiles_hit: Tensor, colors: Tensor, opacity: Tensor,
                            img_height: int, img_width: int, BLOCK_H: int = 16, BLOCK_W: int = 16,
                            background: Optional[Tensor] = None, return_alpha: Optional[bool] = False,
                            isprint: bool = False):
    """-> (out_img[H,W,3], cnt_gs_counts i32[H,W], screenspace_points), or (out_img, out_alpha) when
    return_alpha (rasterize_sum.py:232-236).  The gradient that flows back into `screenspace_points`
    is v_abs_xys[N,4] = per-gaussian sums of (v_x, v_y, |v_x|, |v_y|) (rasterize_sum.py:308,328)."""
    if xys.ndimension() != 2 or xys.size(1) != 2:
        raise ValueError("xys must have dimensions (N, 2)")
    if colors.ndimension() != 2:
        raise ValueError("colors must have dimensions (N, D)")
    if background is None:
        background = torch.ones(colors.shape[-1], dtype=torch.float32, device=colors.device)
    return _RasterizeGaussiansSum.apply(xys.contiguous(), screenspace_points.contiguous(), depths.contiguous(),
                                        radii.contiguous(), conics.contiguous(), num_tiles_hit.contiguous(),
                                        colors.contiguous(), opacity.contiguous(), img_height, img_width,
                                        BLOCK_H, BLOCK_W, background.contiguous(), return_alpha, isprint)


class _RasterizeGaussiansSum(Function):
    @staticmethod
    def forward(ctx, xys, screenspace_points, depths, radii, conics, num_tiles_hit, colors, opacity,
                img_height, img_width, BLOCK_H=16, BLOCK_W=16, background=None, return_alpha=False,
                isprint=False):
        out_img, final_Ts, cnt_gs_counts = forward_impl(
            ctx, False, xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height, img_width,
            BLOCK_H, BLOCK_W, background, 1.0, isprint)  # rasterize_sum.py:147-155: default radius_clip
        ctx.return_alpha = bool(return_alpha)
        if return_alpha:
            return out_img, 1 - final_Ts
        ctx.mark_non_differentiable(cnt_gs_counts)
        return out_img, cnt_gs_counts, screenspace_points.view_as(screenspace_points)

    @staticmethod
    def backward(ctx, v_out_img, *rest):
        v_xy, v_conic, v_colors, v_opacity, v_abs_xys = backward_impl(ctx, False, v_out_img)
        #      xys   screen     depths radii conics   nth   colors    opacity   H W BH BW bg alpha isprint
        return (v_xy, v_abs_xys, None, None, v_conic, None, v_colors, v_opacity, None, None, None, None, None,
                None, None)
