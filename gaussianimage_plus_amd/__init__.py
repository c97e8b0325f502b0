"""gaussianimage_plus_amd -- MI355X-native 2D Gaussian rasterizer for GaussianImage++.

    gaussianimage_plus_amd.gsplat       drop-in `gsplat` operator surface (autograd Functions)
    gaussianimage_plus_amd.gsplat.cuda  native op table over the C ABI (include/gi2d.h)
    gaussianimage_plus_amd.csrc         hand-written gfx950 kernels + C ABI (libgi2d_hip.so)

`install_as_gsplat()` makes `import gsplat` resolve to this implementation so the reference's
model files and train.py run unmodified.
"""
import importlib
import sys

__version__ = "0.1.0"

_SUBMODULES = ("cuda", "utils", "version", "project_gaussians_2d", "project_gaussians_2d_covariance",
               "project_gaussians_2d_scale_rot", "rasterize_sum", "rasterize_sum_plus", "_raster_common")


def install_as_gsplat() -> None:
    """Register this package's operator surface under the module name `gsplat`."""
    pkg = importlib.import_module(__name__ + ".gsplat")
    sys.modules["gsplat"] = pkg
    for sub in _SUBMODULES:
        sys.modules["gsplat." + sub] = importlib.import_module(f"{__name__}.gsplat.{sub}")
