"""Direct-covariance 2D projection (reference: gsplat/gsplat/project_gaussians_2d_covariance.py)."""
from __future__ import annotations

from typing import Tuple

from torch import Tensor
from torch.autograd import Function

from . import cuda as _C
from ._project_common import grads


def project_gaussians_2d_covariance(means2d: Tensor, L_elements: Tensor, img_height: int, img_width: int,
                                    tile_bounds: Tuple[int, int, int], clip_thresh: float = 0.01,
                                    coords_norm: bool = False, isprint: bool = False, clip_coe: float = 3.0,
                                    radius_clip: float = 1.0):
    """-> (xys, depths, radii, conics, num_tiles_hit)

    means2d are PIXEL coordinates, L_elements = (var_x, cov_xy, var_y).  `coords_norm` is accepted and
    ignored exactly as in the reference (project_gaussians_2d_covariance.py:18,53-63)."""
    return _ProjectGaussians2d_covariance.apply(means2d.contiguous(), L_elements.contiguous(), img_height,
                                                img_width, tile_bounds, clip_thresh, clip_coe, radius_clip,
                                                isprint)


class _ProjectGaussians2d_covariance(Function):
    @staticmethod
    def forward(ctx, means2d, L_elements, img_height, img_width, tile_bounds, clip_thresh=0.01,
                clip_coe=3.0, radius_clip=1.0, isprint=False):
        # gradients of outputs nobody used (depths: always) arrive as None instead of freshly zero-filled tensors -- one
        # fill kernel per such output and backward, ~4 us each in a replayed graph
        ctx.set_materialize_grads(False)
        num_points = means2d.shape[-2]
        xys, depths, radii, conics, num_tiles_hit = _C.project_gaussians_2d_covariance_forward(
            num_points, clip_coe, means2d, L_elements, img_height, img_width, tile_bounds, clip_thresh,
            radius_clip, isprint)
        ctx.img_height, ctx.img_width, ctx.num_points = img_height, img_width, num_points
        ctx.save_for_backward(means2d, L_elements, radii, conics)
        ctx.mark_non_differentiable(radii, num_tiles_hit)
        return xys, depths, radii, conics, num_tiles_hit

    @staticmethod
    def backward(ctx, v_xys, v_depths, v_radii, v_conics, v_num_tiles_hit):
        means2d, L_elements, radii, conics = ctx.saved_tensors
        v_xys, v_conics = grads(ctx, v_xys, v_conics, means2d, conics)
        _, v_mean2d, v_L = _C.project_gaussians_2d_covariance_backward(
            ctx.num_points, means2d, L_elements, ctx.img_height, ctx.img_width, radii, conics, v_xys,
            v_depths, v_conics)
        return v_mean2d, v_L, None, None, None, None, None, None, None
