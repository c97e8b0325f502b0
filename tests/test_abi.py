"""CPU-only: the C-ABI library loads and exports every symbol include/gi2d.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gi2d.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gi2d_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_twelve_reference_ops():
    syms = declared_symbols()
    for op in ["project_gaussians_2d_forward", "project_gaussians_2d_backward",
               "project_gaussians_2d_covariance_forward", "project_gaussians_2d_covariance_backward",
               "project_gaussians_2d_scale_rot_forward", "project_gaussians_2d_scale_rot_backward",
               "compute_cov2d_bounds", "map_gaussian_to_intersects", "get_tile_bin_edges",
               "rasterize_sum_forward", "rasterize_sum_backward", "rasterize_sum_plus_forward",
               "rasterize_sum_plus_backward"]:
        assert "gi2d_" + op in syms, op


def test_library_exports_every_declared_symbol():
    from gaussianimage_plus_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/gi2d.h but not exported"


def test_python_binding_covers_every_declared_symbol():
    from gaussianimage_plus_amd import _lib
    bound = set(_lib.SIGNATURES) | set(_lib.SIZE_FUNCS) | set(_lib.STRING_FUNCS)
    assert set(declared_symbols()) == bound
    assert _lib.version().startswith("gi2d")


def test_size_queries_need_no_gpu():
    from gaussianimage_plus_amd import _lib
    lib = _lib.load()
    assert lib.gi2d_sort_workspace_bytes(1000, 64) >= 4 * (1000 + 3 * 64)
    assert lib.gi2d_rasterize_backward_workspace_bytes(100, 1000) >= 1000 * 48
