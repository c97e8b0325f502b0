"""Drop-in `gsplat` operator surface for GaussianImage++ on MI355X (gfx950).

Exposes the 2D-path names of the reference's gsplat/gsplat/__init__.py:3-52; the 3D / SH / N-channel
entries exist but raise NotImplementedError (outside this build).
"""
from typing import Any
import warnings

import torch

from .project_gaussians_2d import project_gaussians_2d, _ProjectGaussians2d
from .project_gaussians_2d_covariance import project_gaussians_2d_covariance, _ProjectGaussians2d_covariance
from .project_gaussians_2d_scale_rot import project_gaussians_2d_scale_rot, _ProjectGaussians2dScaleRot
from .rasterize_sum import rasterize_gaussians_sum
from .rasterize_sum_plus import rasterize_gaussians_plus
from .utils import (
    bin_and_sort_gaussians,
    compute_cov2d_bounds,
    compute_cumulative_intersects,
    get_tile_bin_edges,
    map_gaussian_to_intersects,
)
from .version import __version__


def _out_of_scope(name):
    def f(*args, **kwargs):
        raise NotImplementedError(f"gsplat.{name} (3D / SH path) is outside this build")
    f.__name__ = name
    return f


project_gaussians = _out_of_scope("project_gaussians")
rasterize_gaussians = _out_of_scope("rasterize_gaussians")
spherical_harmonics = _out_of_scope("spherical_harmonics")

__all__ = [
    "__version__",
    "project_gaussians",
    "project_gaussians_2d",
    "project_gaussians_2d_scale_rot",
    "project_gaussians_2d_covariance",
    "rasterize_gaussians",
    "rasterize_gaussians_sum",
    "rasterize_gaussians_plus",
    "spherical_harmonics",
    "bin_and_sort_gaussians",
    "compute_cumulative_intersects",
    "compute_cov2d_bounds",
    "get_tile_bin_edges",
    "map_gaussian_to_intersects",
    "ProjectGaussians2d",
    "ProjectGaussians2dScaleRot",
    "ProjectGaussians2d_covariance",
    "RasterizeGaussiansSum",
    "BinAndSortGaussians",
    "ComputeCumulativeIntersects",
    "ComputeCov2dBounds",
    "GetTileBinEdges",
    "MapGaussiansToIntersects",
]


def _deprecated_function(name, fn):
    """The reference keeps `Function.apply` style aliases that only warn and forward
    (gsplat/gsplat/__init__.py:57-228)."""

    class _Alias(torch.autograd.Function):
        @staticmethod
        def forward(ctx, *args, **kwargs):
            warnings.warn(f"{name} is deprecated, use {fn.__name__} instead", DeprecationWarning)
            return fn(*args, **kwargs)

        @staticmethod
        def backward(ctx: Any, *grad_outputs: Any) -> Any:
            raise NotImplementedError

    _Alias.__name__ = name
    return _Alias


MapGaussiansToIntersects = _deprecated_function("MapGaussiansToIntersects", map_gaussian_to_intersects)
ComputeCumulativeIntersects = _deprecated_function("ComputeCumulativeIntersects", compute_cumulative_intersects)
ComputeCov2dBounds = _deprecated_function("ComputeCov2dBounds", compute_cov2d_bounds)
GetTileBinEdges = _deprecated_function("GetTileBinEdges", get_tile_bin_edges)
BinAndSortGaussians = _deprecated_function("BinAndSortGaussians", bin_and_sort_gaussians)
ProjectGaussians2d = _deprecated_function("ProjectGaussians2d", project_gaussians_2d)
ProjectGaussians2dScaleRot = _deprecated_function("ProjectGaussians2dScaleRot", project_gaussians_2d_scale_rot)
ProjectGaussians2d_covariance = _deprecated_function("ProjectGaussians2d_covariance", project_gaussians_2d_covariance)
RasterizeGaussiansSum = _deprecated_function("RasterizeGaussiansSum", rasterize_gaussians_sum)
