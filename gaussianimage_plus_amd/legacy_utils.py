"""The `utils`-level names the reference's Cholesky / rotation-scale model files expect.

`models/gaussianimage_cholesky.py:4` and `models/gaussianimage_rs.py:4` do `from utils import *` and then use
`loss_fn` (:305,339 / :260,476), `F` (:308 / :263) and `BasicPointCloud` (:172 / :561).  In the snapshot those names
live in `models/utils.py:2,60-80,174-177`, which the import does not reach (and which pulls in cv2, matplotlib, wandb
and pytorch_msssim at module level), so the two model files cannot be imported as they are.  This module supplies the
three names and `install_as_utils()` puts them where `from utils import *` finds them: into the reference's own
top-level `utils` module when it is importable (its image / logging helpers stay), otherwise into a fresh module of
that name.  Only the loss branches built from torch alone are provided ('L2' -- the default on every BASELINE config --
'L1' and 'Fusion3'); the SSIM / MS-SSIM fusions need the third-party `pytorch_msssim` and are served only when that
package is importable.
"""
from __future__ import annotations

import importlib
import sys
import types
from typing import NamedTuple

import numpy as np
import torch.nn.functional as F  # noqa: F401  (re-exported: the model files call F.mse_loss)

__all__ = ["loss_fn", "BasicPointCloud", "F", "install_as_utils"]


class BasicPointCloud(NamedTuple):
    """models/utils.py:174-177 (only named in the signature of create_from_pcd)."""
    points: np.ndarray
    colors: np.ndarray
    normals: np.ndarray


def _structural(kind: str):
    try:
        mod = importlib.import_module("pytorch_msssim")
    except ImportError as e:  # third-party metric, SURVEY section 2 item 12: not part of this build
        raise NotImplementedError(f"loss_fn: the {kind} terms need the third-party package pytorch_msssim") from e
    return getattr(mod, kind)


def loss_fn(pred, target, loss_type="L2", lambda_value=0.7):
    """models/utils.py:60-80: the target carries no gradient, both sides are compared in fp32; an unknown
    `loss_type` fails the way the reference's if-chain does (its result is never bound)."""
    target = target.detach().float()
    pred = pred.float()
    mse = lambda: F.mse_loss(pred, target)
    l1 = lambda: F.l1_loss(pred, target)
    one_minus = lambda kind, **kw: 1 - _structural(kind)(pred, target, data_range=1, size_average=True, **kw)
    if loss_type == "L2":
        return mse()
    if loss_type == "L1":
        return l1()
    if loss_type == "SSIM":
        return one_minus("ssim")
    if loss_type == "Fusion1":
        return lambda_value * mse() + (1 - lambda_value) * one_minus("ssim")
    if loss_type == "Fusion2":
        return lambda_value * l1() + (1 - lambda_value) * one_minus("ssim")
    if loss_type == "Fusion3":
        return lambda_value * mse() + (1 - lambda_value) * l1()
    if loss_type == "Fusion4":
        return lambda_value * l1() + (1 - lambda_value) * one_minus("ms_ssim")
    if loss_type == "Fusion_hinerv":
        return lambda_value * l1() + (1 - lambda_value) * one_minus("ms_ssim", win_size=5)
    raise UnboundLocalError(f"local variable 'loss' referenced before assignment (loss_type={loss_type!r})")


def install_as_utils() -> types.ModuleType:
    """Make `from utils import *` deliver loss_fn / BasicPointCloud / F.  Call it (like install_as_gsplat) before
    importing models.gaussianimage_cholesky or models.gaussianimage_rs."""
    mod = sys.modules.get("utils")
    if mod is None:
        try:
            mod = importlib.import_module("utils")  # the reference's top-level utils.py, when it is on the path
        except Exception:  # absent, or one of its own third-party imports (torchvision, constriction) is
            mod = types.ModuleType("utils")
            mod.__doc__ = "stand-in for the reference's utils module: only the names of legacy_utils"
            sys.modules["utils"] = mod
    for name in ("loss_fn", "BasicPointCloud", "F"):
        setattr(mod, name, globals()[name])
    if hasattr(mod, "__all__"):
        mod.__all__ = list(dict.fromkeys(list(mod.__all__) + ["loss_fn", "BasicPointCloud", "F"]))
    return mod
