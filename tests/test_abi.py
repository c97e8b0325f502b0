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


def test_quant_argument_checks_need_no_gpu():
    """Bad quantiser specs are rejected before anything is launched."""
    import ctypes as C
    from gaussianimage_plus_amd import _lib
    from gaussianimage_plus_amd.quantize import make_spec
    lib = _lib.load()
    assert lib.gi2d_quant_workspace_bytes(30000) >= (30000 // 256) * 16 * 4
    bad_kind = make_spec([7], [0], [255])
    bad_range = make_spec([0, 1], [0, 5], [255, 5])
    five = make_spec([0], [0], [255])
    five.channels = 5
    for spec in (bad_kind, bad_range, five):
        rc = lib.gi2d_quant_compress(C.byref(spec), 4, C.c_void_p(8), C.c_void_p(8), C.c_void_p(8), C.c_void_p(8), None)
        assert rc == -1, rc
        assert b"quant" in lib.gi2d_last_error_string()
    ok = make_spec([0, 1], [0, 0], [63, 1023])
    assert lib.gi2d_quant_forward(C.byref(ok), 10, C.c_void_p(8), C.c_void_p(8), None, None, None, 0, None) == -2
    assert lib.gi2d_quant_forward(C.byref(ok), 0, None, None, None, None, None, 0, None) == 0  # empty input: nothing to do


def test_product_library_carries_no_development_switch():
    """Cut-off / knock-out / trace builds (`make VARIANT=... EXTRA=-DGI2D_...`) live under build/variants/ and say so
    in gi2d_version(); the in-tree library must be a plain build, and the loader refuses anything else."""
    import subprocess
    import sys
    from gaussianimage_plus_amd import _lib
    assert "dev[" not in _lib.version(), _lib.version()
    variant = os.path.join(ROOT, "build", "variants")
    libs = [os.path.join(d, "libgi2d_hip.so") for d, _, f in os.walk(variant) if "libgi2d_hip.so" in f] \
        if os.path.isdir(variant) else []
    for path in libs[:1]:  # any development library in the tree: loading it without the override must fail
        ver = ctypes.CDLL(path).gi2d_version
        ver.restype = ctypes.c_char_p
        if b"dev[" not in ver():
            continue
        env = dict(os.environ, GI2D_LIB=path)
        env.pop("GI2D_ALLOW_DEV_BUILD", None)
        r = subprocess.run([sys.executable, "-c", "from gaussianimage_plus_amd import _lib; _lib.load()"], cwd=ROOT,
                           env=env, capture_output=True, text=True)
        assert r.returncode != 0 and "development build" in r.stderr
