"""CPU: the `utils`-level shim of SURVEY 8b -- the names models/gaussianimage_cholesky.py:4,172,305,308 and
models/gaussianimage_rs.py:4,260,263,561 take from `from utils import *`."""
import sys
import types

import numpy as np
import pytest
import torch


def test_star_import_delivers_the_three_names(monkeypatch):
    from gaussianimage_plus_amd import legacy_utils
    monkeypatch.setitem(sys.modules, "utils", types.ModuleType("utils"))  # whatever `utils` is already there
    sys.modules["utils"].image_path_to_tensor = lambda p: p              # ... keeps its own names
    legacy_utils.install_as_utils()
    ns = {}
    exec("from utils import *", ns)
    assert ns["loss_fn"] is legacy_utils.loss_fn and ns["BasicPointCloud"] is legacy_utils.BasicPointCloud
    assert ns["F"] is torch.nn.functional and "image_path_to_tensor" in ns
    pc = ns["BasicPointCloud"](points=np.zeros((2, 3)), colors=np.ones((2, 3)), normals=np.zeros((2, 3)))
    assert pc._fields == ("points", "colors", "normals") and pc.colors.sum() == 6


def test_loss_fn_l2_is_the_reference_expression():
    from gaussianimage_plus_amd.legacy_utils import loss_fn
    g = torch.Generator().manual_seed(5)
    pred = torch.rand(1, 3, 8, 12, generator=g, dtype=torch.float64, requires_grad=True)
    target = torch.rand(1, 3, 8, 12, generator=g, requires_grad=True)
    loss = loss_fn(pred, target, "L2", lambda_value=0.7)
    assert loss.dtype == torch.float32
    assert loss.item() == torch.nn.functional.mse_loss(pred.float(), target.detach().float()).item()
    loss.backward()
    assert target.grad is None  # target.detach()
    want = 2 * (pred.detach().float() - target.detach()) / pred.numel()
    assert torch.allclose(pred.grad.float(), want, rtol=1e-6, atol=1e-9)
    l1 = loss_fn(pred, target, "L1")
    f3 = loss_fn(pred, target, "Fusion3", lambda_value=0.25)
    assert abs(f3.item() - (0.25 * loss.item() + 0.75 * l1.item())) < 1e-7
    with pytest.raises(UnboundLocalError):
        loss_fn(pred, target, "no such loss")
    try:
        import pytorch_msssim  # noqa: F401
    except ImportError:
        with pytest.raises(NotImplementedError):
            loss_fn(pred, target, "Fusion1")


REFERENCE = "/root/reference"


@pytest.mark.skipif(not __import__("os").path.isdir(REFERENCE), reason="dev container only: the reference checkout")
def test_reference_cholesky_model_file_imports_against_the_drop_ins():
    """models/gaussianimage_cholesky.py, unmodified, imports once `gsplat`, `utils` and `quantize` resolve to this
    repo (its fourth import, `optimizer`, is the reference's own torch-only file).  Run in a child interpreter so the
    module table of the test session stays as it is."""
    import os
    import subprocess
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import gaussianimage_plus_amd as g; g.install_as_gsplat()\n"
        "from gaussianimage_plus_amd import legacy_utils, quantize\n"
        "sys.modules['quantize'] = quantize\n"
        "legacy_utils.install_as_utils()\n"
        "sys.path.append(%r)\n"
        "import importlib; m = importlib.import_module('models.gaussianimage_cholesky')\n"
        "assert m.loss_fn is legacy_utils.loss_fn and m.BasicPointCloud is legacy_utils.BasicPointCloud\n"
        "assert m.project_gaussians_2d.__module__.startswith('gaussianimage_plus_amd.gsplat')\n"
        "print('imported', m.GaussianImage_Cholesky.__name__)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), REFERENCE)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "imported GaussianImage_Cholesky" in out.stdout, out.stderr[-1500:]
