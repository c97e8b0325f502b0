"""Scale + rotation 2D projection (reference: gsplat/gsplat/project_gaussians_2d_scale_rot.py)."""
from __future__ import annotations

from typing import Tuple

from torch import Tensor
from torch.autograd import Function

from . import cuda as _C
from ._project_common import grads, is_legacy_call


def project_gaussians_2d_scale_rot(*args, **kwargs):
    """project_gaussians_2d_scale_rot(means2d, scales2d, rotation, img_height, img_width, tile_bounds,
    clip_thresh=0.01, coords_norm=False, radius_clip=1.0, isprint=False)
    -> (xys, depths, radii, conics, num_tiles_hit)

    means2d in pixels; Sigma = R diag(s)^2 R^T with R = [[cos, sin], [-sin, cos]]; clip_coe fixed at 3.0.
    Legacy form (models/gaussianimage_rs.py:226-227, :463-465): the means are followed by a
    `screenspace_points` [N,4] tensor, which is passed through as the second result."""
    if is_legacy_call(args):
        means2d, screenspace_points = args[0], args[1]
        out = _current(means2d, *args[2:], **kwargs)
        return (out[0], screenspace_points, *out[1:])
    return _current(*args, **kwargs)


def _current(means2d: Tensor, scales2d: Tensor, rotation: Tensor, img_height: int, img_width: int,
             tile_bounds: Tuple[int, int, int], clip_thresh: float = 0.01, coords_norm: bool = False,
             radius_clip: float = 1.0, isprint: bool = False):
    return _ProjectGaussians2dScaleRot.apply(means2d.contiguous(), scales2d.contiguous(), rotation.contiguous(),
                                             img_height, img_width, tile_bounds, clip_thresh, radius_clip,
                                             isprint)


class _ProjectGaussians2dScaleRot(Function):
    @staticmethod
    def forward(ctx, means2d, scales2d, rotation, img_height, img_width, tile_bounds, clip_thresh=0.01,
                radius_clip=2.0, isprint=False):
        # gradients of outputs nobody used (depths: always) arrive as None instead of freshly zero-filled tensors -- one
        # fill kernel per such output and backward, ~4 us each in a replayed graph
        ctx.set_materialize_grads(False)
        num_points = means2d.shape[-2]
        if num_points < 1 or means2d.shape[-1] != 2:  # project_gaussians_2d_scale_rot.py:84-85
            raise ValueError(f"Invalid shape for means2d: {means2d.shape}")
        xys, depths, radii, conics, num_tiles_hit = _C.project_gaussians_2d_scale_rot_forward(
            num_points, 3.0, means2d, scales2d, rotation, img_height, img_width, tile_bounds, clip_thresh,
            radius_clip, isprint)
        ctx.img_height, ctx.img_width, ctx.num_points = img_height, img_width, num_points
        ctx.save_for_backward(means2d, scales2d, rotation, radii, conics)
        ctx.mark_non_differentiable(radii, num_tiles_hit)
        return xys, depths, radii, conics, num_tiles_hit

    @staticmethod
    def backward(ctx, v_xys, v_depths, v_radii, v_conics, v_num_tiles_hit):
        means2d, scales2d, rotation, radii, conics = ctx.saved_tensors
        v_xys, v_conics = grads(ctx, v_xys, v_conics, means2d, conics)
        _, v_mean2d, v_scale, v_rot = _C.project_gaussians_2d_scale_rot_backward(
            ctx.num_points, means2d, scales2d, rotation, ctx.img_height, ctx.img_width, radii, conics,
            v_xys, v_depths, v_conics)
        return v_mean2d, v_scale, v_rot.view_as(rotation), None, None, None, None, None, None
