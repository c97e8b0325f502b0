"""Shared plumbing of the three projection autograd Functions."""
from __future__ import annotations

import torch


def is_legacy_call(args) -> bool:
    """The Cholesky / RS model files still call the projection wrappers with an extra
    `screenspace_points` [N,4] tensor right after the means
    (models/gaussianimage_cholesky.py:208-209, models/gaussianimage_rs.py:226-227)."""
    return (len(args) >= 3 and isinstance(args[1], torch.Tensor) and args[1].dim() == 2
            and args[1].size(-1) == 4 and isinstance(args[2], torch.Tensor))


def grads(ctx, v_xys, v_conics, like_xy, like_conic):
    """autograd hands None for outputs that did not take part in the loss."""
    v_xys = torch.zeros_like(like_xy) if v_xys is None else v_xys.contiguous()
    v_conics = torch.zeros_like(like_conic) if v_conics is None else v_conics.contiguous()
    return v_xys, v_conics
