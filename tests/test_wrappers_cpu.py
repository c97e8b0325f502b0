"""CPU-only: host-side behaviour of the drop-in surface (import shim, legacy call detection, error types,
no silent CPU fallback)."""
import sys

import pytest
import torch


def test_import_shim_resolves_gsplat_to_this_package():
    for k in [k for k in sys.modules if k == "gsplat" or k.startswith("gsplat.")]:
        del sys.modules[k]
    import gsplat
    from gsplat.project_gaussians_2d import project_gaussians_2d
    from gsplat.project_gaussians_2d_covariance import project_gaussians_2d_covariance
    from gsplat.project_gaussians_2d_scale_rot import project_gaussians_2d_scale_rot
    from gsplat.rasterize_sum import rasterize_gaussians_sum
    from gsplat.rasterize_sum_plus import rasterize_gaussians_plus
    import gsplat.cuda as _C
    assert gsplat.__name__ == "gaussianimage_plus_amd.gsplat"
    for name in ["project_gaussians_2d", "project_gaussians_2d_scale_rot", "project_gaussians_2d_covariance",
                 "rasterize_gaussians_sum", "bin_and_sort_gaussians", "compute_cumulative_intersects",
                 "compute_cov2d_bounds", "get_tile_bin_edges", "map_gaussian_to_intersects"]:
        assert callable(getattr(gsplat, name)), name
    for op in ["project_gaussians_2d_forward", "project_gaussians_2d_backward", "project_gaussians_2d_covariance_forward",
               "project_gaussians_2d_covariance_backward", "project_gaussians_2d_scale_rot_forward",
               "project_gaussians_2d_scale_rot_backward", "compute_cov2d_bounds", "map_gaussian_to_intersects",
               "get_tile_bin_edges", "rasterize_sum_forward", "rasterize_sum_backward", "rasterize_sum_plus_forward",
               "rasterize_sum_plus_backward"]:
        assert callable(getattr(_C, op)), op
    with pytest.raises(NotImplementedError):
        _C.rasterize_forward()
    with pytest.raises(NotImplementedError):
        gsplat.project_gaussians(None)


def test_legacy_call_detection():
    from gaussianimage_plus_amd.gsplat._project_common import is_legacy_call
    xyz, screen, chol = torch.zeros(5, 2), torch.zeros(5, 4), torch.zeros(5, 3)
    assert is_legacy_call((xyz, screen, chol, 16, 16, (1, 1, 1)))       # models/gaussianimage_cholesky.py:208
    assert not is_legacy_call((xyz, chol, 16, 16, (1, 1, 1)))           # current 5-return form
    assert is_legacy_call((xyz, screen, torch.zeros(5, 2), torch.zeros(5, 1), 16, 16, (1, 1, 1)))  # RS model


def test_ops_refuse_cpu_tensors_instead_of_falling_back():
    import gaussianimage_plus_amd.gsplat.cuda as _C
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        _C.project_gaussians_2d_forward(4, 3.0, torch.zeros(4, 2), torch.zeros(4, 3), 16, 16, (1, 1, 1), 0.01, 1.0, False)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        _C.map_gaussian_to_intersects(1, 1, torch.zeros(1, 2), torch.zeros(1), torch.zeros(1, dtype=torch.int32),
                                      torch.zeros(1, dtype=torch.int32), (1, 1, 1), 1.0, False)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        _C.rasterize_sum_plus_forward((1, 1, 1), (16, 16, 1), (16, 16, 1), torch.zeros(0, dtype=torch.int32),
                                      torch.zeros(1, 2, dtype=torch.int32), torch.zeros(0, 2), torch.zeros(0, 3),
                                      torch.zeros(0, 3), torch.zeros(0, 1), torch.ones(3), False)


def test_wrapper_argument_errors_match_the_reference():
    import gaussianimage_plus_amd.gsplat as gs
    with pytest.raises(ValueError, match=r"\(N, 2\)"):  # rasterize_sum_plus.py:52-53
        gs.rasterize_gaussians_plus(torch.zeros(4, 3), None, None, None, None, torch.zeros(4, 3), None, 16, 16)
    with pytest.raises(ValueError, match=r"\(N, D\)"):  # rasterize_sum_plus.py:55-56
        gs.rasterize_gaussians_plus(torch.zeros(4, 2), None, None, None, None, torch.zeros(4), None, 16, 16)
    with pytest.raises(AssertionError):                 # rasterize_sum_plus.py:43-46
        gs.rasterize_gaussians_plus(torch.zeros(4, 2), None, None, None, None, torch.zeros(4, 3), None, 16, 16,
                                    background=torch.ones(4))
    with pytest.raises(AssertionError):                 # utils.py:203-205
        gs.compute_cov2d_bounds(torch.zeros(3, 2))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from gaussianimage_plus_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libgi2d_hip.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_hotpath_and_launcher_import_without_a_gpu():
    import gaussianimage_plus_amd.hotpath as hp
    import gaussianimage_plus_amd.launch as launch
    assert hasattr(hp, "HotPath") and callable(launch.fit_image)
    img = launch.synthetic_image(32, 48, 1)
    assert img.shape == (32, 48, 3) and 0.0 <= float(img.min()) and float(img.max()) <= 1.0
