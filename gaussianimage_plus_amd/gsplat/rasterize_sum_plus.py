"""`rasterize_gaussians_plus` (reference: gsplat/gsplat/rasterize_sum_plus.py) -- the rasterizer
train.py actually drives through GaussianImage_Covariance."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor
from torch.autograd import Function

from ._raster_common import backward_impl, forward_impl


def rasterize_gaussians_plus(xys: Tensor, depths: Tensor, radii: Tensor, conics: Tensor, num_tiles_hit: Tensor,
                             colors: Tensor, opacity: Tensor, img_height: int, img_width: int,
                             BLOCK_H: int = 16, BLOCK_W: int = 16, background: Optional[Tensor] = None,
                             return_alpha: Optional[bool] = False, radius_clip: float = 1.0,
                             isprint: bool = False) -> Tensor:
    """Additive splat of N 2D gaussians into out_img[H,W,3]:
    out[p] = sum_g colors[g] * min(1, opacity[g] * exp(-sigma_g(p))) over the pairs with sigma >= 0 and
    alpha >= 1/255 (forward.cu:570-691).  Differentiable w.r.t. xys, conics, colors, opacity."""
    if colors.dtype == torch.uint8:
        colors = colors.float() / 255
    if background is not None:
        assert background.shape[0] == colors.shape[-1], (
            f"incorrect shape of background color tensor, expected shape {colors.shape[-1]}")
    else:
        background = torch.ones(colors.shape[-1], dtype=torch.float32, device=colors.device)
    if xys.ndimension() != 2 or xys.size(1) != 2:
        raise ValueError("xys must have dimensions (N, 2)")
    if colors.ndimension() != 2:
        raise ValueError("colors must have dimensions (N, D)")
    return _RasterizeGaussiansSum.apply(xys.contiguous(), depths.contiguous(), radii.contiguous(),
                                        conics.contiguous(), num_tiles_hit.contiguous(), colors.contiguous(),
                                        opacity.contiguous(), img_height, img_width, BLOCK_H, BLOCK_W,
                                        background.contiguous(), radius_clip, isprint)


class _RasterizeGaussiansSum(Function):
    @staticmethod
    def forward(ctx, xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height, img_width,
                BLOCK_H=16, BLOCK_W=16, background=None, radius_clip=1.0, isprint=False):
        out_img, _, _ = forward_impl(ctx, True, xys, depths, radii, conics, num_tiles_hit, colors, opacity,
                                     img_height, img_width, BLOCK_H, BLOCK_W, background, radius_clip, isprint)
        return out_img

    @staticmethod
    def backward(ctx, v_out_img, v_out_alpha=None):
        v_xy, v_conic, v_colors, v_opacity, _ = backward_impl(ctx, True, v_out_img)
        #      xys   depths radii conics   nth   colors    opacity    H     W    BH    BW    bg   clip  isprint
        return v_xy, None, None, v_conic, None, v_colors, v_opacity, None, None, None, None, None, None, None
