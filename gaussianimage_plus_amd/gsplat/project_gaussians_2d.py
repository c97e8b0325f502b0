"""Cholesky-parameterised 2D projection (reference: gsplat/gsplat/project_gaussians_2d.py)."""
from __future__ import annotations

from typing import Tuple

from torch import Tensor
from torch.autograd import Function

from . import cuda as _C
from ._project_common import grads, is_legacy_call


def project_gaussians_2d(*args, **kwargs):
    """project_gaussians_2d(means2d, L_elements, img_height, img_width, tile_bounds, clip_thresh=0.01,
    radius_clip=1.0, isprint=False) -> (xys, depths, radii, conics, num_tiles_hit)

    means2d are NDC coordinates in (-1, 1); L_elements = (l11, l21, l22), Sigma = L L^T; clip_coe is
    fixed at 3.0 (project_gaussians_2d.py:88).  Differentiable w.r.t. means2d and L_elements (with
    the reference's own Cholesky VJP, backward2d.cu:39-41).

    Legacy form, still used by models/gaussianimage_cholesky.py:208-209:
    project_gaussians_2d(xyz, screenspace_points[N,4], L_elements, H, W, tile_bounds, isprint=...)
    -> (xys, screenspace_points, depths, radii, conics, num_tiles_hit)."""
    if is_legacy_call(args):
        means2d, screenspace_points = args[0], args[1]
        out = _current(means2d, *args[2:], **kwargs)
        return (out[0], screenspace_points, *out[1:])
    return _current(*args, **kwargs)


def _current(means2d: Tensor, L_elements: Tensor, img_height: int, img_width: int,
             tile_bounds: Tuple[int, int, int], clip_thresh: float = 0.01, radius_clip: float = 1.0,
             isprint: bool = False):
    return _ProjectGaussians2d.apply(means2d.contiguous(), L_elements.contiguous(), img_height, img_width,
                                     tile_bounds, clip_thresh, radius_clip, isprint)


class _ProjectGaussians2d(Function):
    @staticmethod
    def forward(ctx, means2d, L_elements, img_height, img_width, tile_bounds, clip_thresh=0.01,
                radius_clip=2.0, isprint=False):
        # gradients of outputs nobody used (depths: always) arrive as None instead of freshly zero-filled tensors -- one
        # fill kernel per such output and backward, ~4 us each in a replayed graph
        ctx.set_materialize_grads(False)
        num_points = means2d.shape[-2]
        xys, depths, radii, conics, num_tiles_hit = _C.project_gaussians_2d_forward(
            num_points, 3.0, means2d, L_elements, img_height, img_width, tile_bounds, clip_thresh,
            radius_clip, isprint)
        ctx.img_height, ctx.img_width, ctx.num_points = img_height, img_width, num_points
        ctx.save_for_backward(means2d, L_elements, radii, conics)
        ctx.mark_non_differentiable(radii, num_tiles_hit)
        return xys, depths, radii, conics, num_tiles_hit

    @staticmethod
    def backward(ctx, v_xys, v_depths, v_radii, v_conics, v_num_tiles_hit):
        means2d, L_elements, radii, conics = ctx.saved_tensors
        v_xys, v_conics = grads(ctx, v_xys, v_conics, means2d, conics)
        _, v_mean2d, v_L = _C.project_gaussians_2d_backward(
            ctx.num_points, means2d, L_elements, ctx.img_height, ctx.img_width, radii, conics, v_xys,
            v_depths, v_conics)
        return v_mean2d, v_L, None, None, None, None, None, None
