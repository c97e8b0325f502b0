"""Binning helpers of the operator surface (reference: gsplat/gsplat/utils.py).

Same public functions and return shapes; the work is done by the gfx950 kernels behind
`gsplat.cuda`.  `bin_and_sort_gaussians` does not call torch.sort: the (tile | depth) keys are
ordered by the native stable counting sort (csrc/gi2d_binning.hip).
"""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor

from . import cuda as _C


def map_gaussian_to_intersects(num_points: int, num_intersects: int, xys: Tensor, depths: Tensor,
                               radii: Tensor, cum_tiles_hit: Tensor, tile_bounds: Tuple[int, int, int],
                               radius_clip: float = 1.0, isprint: bool = False) -> Tuple[Tensor, Tensor]:
    """utils.py:12-57 -> (isect_ids i64[M], gaussian_ids i32[M]); not differentiable."""
    return _C.map_gaussian_to_intersects(num_points, num_intersects, xys.contiguous(), depths.contiguous(),
                                         radii.contiguous(), cum_tiles_hit.contiguous(), tile_bounds,
                                         radius_clip, isprint)


def get_tile_bin_edges(num_intersects: int, isect_ids_sorted: Tensor) -> Tensor:
    """utils.py:166-187 -> tile_bins i32[num_intersects, 2], rows indexed by tile id."""
    return _C.get_tile_bin_edges(num_intersects, isect_ids_sorted.contiguous())


def compute_cov2d_bounds(cov2d: Tensor, clip_coe: float = 3.0) -> Tuple[Tensor, Tensor]:
    """utils.py:190-209 -> (conics[N,3], radii[N,1])"""
    assert cov2d.shape[-1] == 3, (
        f"Expected input cov2d to be of shape (*batch, 3) (upper triangular values), but got {tuple(cov2d.shape)}")
    num_pts = cov2d.shape[0]
    assert num_pts > 0
    return _C.compute_cov2d_bounds(num_pts, clip_coe, cov2d.contiguous())


compute_cov2d_bounds_xy = compute_cov2d_bounds


def compute_cumulative_intersects(num_tiles_hit: Tensor) -> Tuple[int, Tensor]:
    """utils.py:231-250 -> (num_intersects: int, cum_tiles_hit i32[N]).

    The scan runs in one native launch; reading the total back is the host sync the reference API
    implies.  The rasterize wrappers use the same scan but keep the read-back to 4 bytes."""
    if num_tiles_hit.numel() == 0:
        return 0, torch.zeros_like(num_tiles_hit, dtype=torch.int32)
    nth = num_tiles_hit.contiguous()
    if nth.dtype != torch.int32:
        nth = nth.to(torch.int32)
    cum, total = _C.cumsum_tiles_hit(nth)
    return int(total.item()), cum


def bin_and_sort_gaussians(num_points: int, num_intersects: int, xys: Tensor, depths: Tensor, radii: Tensor,
                           cum_tiles_hit: Tensor, tile_bounds: Tuple[int, int, int], radius_clip: float = 1.0,
                           isprint: bool = False) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """utils.py:253-311 -> (isect_ids_unsorted, gaussian_ids_unsorted, isect_ids_sorted,
    gaussian_ids_sorted, tile_bins).

    tile_bins has max(num_intersects, #tiles) rows (the reference allocates num_intersects rows and
    indexes them by tile id, which overruns when there are fewer intersections than tiles)."""
    isect_ids, gaussian_ids = map_gaussian_to_intersects(num_points, num_intersects, xys, depths, radii,
                                                         cum_tiles_hit, tile_bounds, radius_clip, isprint)
    num_tiles = int(tile_bounds[0]) * int(tile_bounds[1])
    srt = _C.sort_intersects(isect_ids, gaussian_ids, num_tiles)
    tile_bins = _C.get_tile_bin_edges(num_intersects, srt["isect_ids_sorted"], rows=max(num_intersects, num_tiles))
    return isect_ids, gaussian_ids, srt["isect_ids_sorted"], srt["gaussian_ids_sorted"], tile_bins
