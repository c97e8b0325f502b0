"""ctypes binding of the C ABI in include/gi2d.h (libgi2d_hip.so, built by csrc/Makefile).

There is deliberately no CPU fallback: if the HIP library is missing, loading fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

# torch bundles its own libamdhip64.so (soname libamdhip64.so.7).  It must be in the process BEFORE
# libgi2d_hip.so is dlopen'ed so that our NEEDED libamdhip64.so.7 binds to that same runtime; loading
# /opt/rocm's copy first would leave two HIP runtimes in one process (hipErrorNoDevice on the second).
import torch  # noqa: F401  (device memory + streams; see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# GI2D_LIB: a development variant built by `make VARIANT=...` (csrc/Makefile) instead of the product library -- for the
# measurement scripts under tools/ only; such a library is refused unless GI2D_ALLOW_DEV_BUILD=1 (see load()).
LIB_PATH = os.environ.get("GI2D_LIB") or os.path.join(_HERE, "libgi2d_hip.so")

_i, _u, _f, _p, _sz = C.c_int, C.c_uint, C.c_float, C.c_void_p, C.c_size_t

# name -> argtypes; mirrors include/gi2d.h one to one (tests/test_abi.py checks the symbol list)
SIGNATURES = {
    "gi2d_project_gaussians_2d_forward": [_i, _f, _p, _p, _u, _u, _i, _i, _f, _f, _p, _p, _p, _p, _p, _p],
    "gi2d_project_gaussians_2d_covariance_forward": [_i, _f, _p, _p, _u, _u, _i, _i, _f, _f, _p, _p, _p, _p, _p, _p],
    "gi2d_project_gaussians_2d_scale_rot_forward": [_i, _f, _p, _p, _p, _u, _u, _i, _i, _f, _f, _p, _p, _p, _p, _p, _p],
    "gi2d_project_gaussians_2d_backward": [_i, _p, _p, _u, _u, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "gi2d_project_gaussians_2d_covariance_backward": [_i, _p, _p, _u, _u, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "gi2d_project_gaussians_2d_scale_rot_backward": [_i, _p, _p, _p, _u, _u, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "gi2d_compute_cov2d_bounds": [_i, _f, _p, _p, _p, _p],
    "gi2d_cumsum_tiles_hit": [_i, _p, _p, _p, _p],
    "gi2d_map_gaussian_to_intersects": [_i, _i, _p, _p, _p, _p, _i, _i, _f, _p, _p, _p],
    "gi2d_sort_intersects": [_i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p],
    "gi2d_get_tile_bin_edges": [_i, _p, _i, _p, _p],
    "gi2d_rasterize_sum_forward": [_i, _i, _u, _u, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "gi2d_rasterize_sum_plus_forward": [_i, _i, _u, _u, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "gi2d_rasterize_sum_backward": [_i, _i, _u, _u, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p],
    "gi2d_bin_gaussians": [_i, _i, _p, _p, _i, _i, _f, _p, _p, _p, _p, _sz, _p],
    "gi2d_rasterize_backward_tiles": [_u, _u, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p],
    "gi2d_rasterize_backward_reduce": [_i, _p, _p, _i, _i, _f, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p],
    "gi2d_rasterize_sum_plus_backward": [_i, _i, _u, _u, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p],
}
SIGNATURES.update({
    "gi2d_fast_workspace_init": [_p, _sz, _i, _i, _i, _p],
    "gi2d_fast_workspace_views": [_p, _sz, _i, _i, _i, _p, _p],
    "gi2d_fast_bin": [_i, _p, _p, _p, _p, _p, _i, _i, _f, _p, _sz, _p, _p],
    "gi2d_fast_project_bin": [_i, _i, _f, _p, _p, _p, _p, _p, _u, _u, _i, _i, _f, _p, _p, _p, _p, _p, _p, _sz, _p, _p],
    "gi2d_fast_rasterize_forward": [_i, _i, _i, _u, _u, _p, _p, _sz, _p, _p, _p, _p, _p],
    "gi2d_fast_rasterize_forward_backward": [_i, _i, _i, _u, _u, _p, _p, _p, _f, _p, _p, _sz, _p, _p, _p],
    "gi2d_fast_rasterize_backward_tiles": [_i, _i, _i, _u, _u, _p, _p, _i, _p, _sz, _p],
    "gi2d_fast_rasterize_backward_reduce": [_i, _i, _i, _p, _sz, _p, _p, _p, _p, _p, _p],
    "gi2d_fast_reduce_project_backward": [_i, _i, _p, _p, _u, _u, _p, _p, _p, _i, _i, _f, _p, _sz, _p, _p, _p, _p, _p,
                                          _p, _p, _p, _p, _p],
    "gi2d_fast_reduce_project_backward_project_bin": [_i, _i, _f, _p, _p, _p, _p, _p, _u, _u, _p, _p, _p, _p, _p, _i, _i,
                                                      _f, _p, _sz, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "gi2d_fast_tile_capacity": [],
    "gi2d_timer_create": [_p], "gi2d_timer_destroy": [_p], "gi2d_timer_arm": [_p], "gi2d_timer_elapsed_us": [_p, _p],
    "gi2d_timer_arm_many": [_p, _i, _i],
    # struct gi2d_train_state* (gaussianimage_plus_amd/trainer.py::_TrainState)
    "gi2d_train_render": [_p, _p],
    "gi2d_train_step": [_p, _p, C.c_double, C.c_double, _f, _i, _p],
    "gi2d_train_steps": [_p, _p, C.c_double, C.c_double, _f, _i, _i, _p],
    # several images per launch: host array of struct gi2d_train_state* / of struct gi2d_fast_image
    "gi2d_train_steps_batched": [_i, _p, _p, _sz, _p, C.c_double, C.c_double, _f, _i, _i, _p],
    "gi2d_fast_rasterize_forward_backward_batched": [_i, _p, _p, _sz, _p],
    "gi2d_batch_tile_pass_form": [_p],
    "gi2d_train_prune": [_p, _p, _sz, _p, _p],
    "gi2d_train_grow": [_p, _i, _i, _p, _i, _p, _sz, _p, _p],
    # quantisers: struct gi2d_quant_spec* (gaussianimage_plus_amd/quantize.py::_QuantSpec)
    "gi2d_quant_init": [_p, _i, _p, _p, _p, _sz, _p],
    "gi2d_quant_forward": [_p, _i, _p, _p, _p, _p, _p, _sz, _p],
    "gi2d_quant_backward": [_p, _i, _p, _p, _p, _p, _p, _p, _sz, _p],
    "gi2d_quant_compress": [_p, _i, _p, _p, _p, _p, _p],
    "gi2d_quant_decompress": [_p, _i, _p, _p, _p, _p],
    "gi2d_quant_half": [_sz, _p, _p, _p],
})
SIZE_FUNCS = {
    "gi2d_fast_workspace_bytes": [_i, _i, _i],
    "gi2d_sort_workspace_bytes": [_i, _i],
    "gi2d_rasterize_backward_workspace_bytes": [_i, _i],
    "gi2d_bin_workspace_bytes": [_i, _i],
    "gi2d_quant_workspace_bytes": [_i],
    "gi2d_densify_scratch_bytes": [_p, _i],
    "gi2d_batch_bytes": [_i],
    "gi2d_train_inbox_bytes": [_i, _i],
}
STRING_FUNCS = ["gi2d_version", "gi2d_last_error_string"]

_lib = None


class Gi2dError(RuntimeError):
    """A C-ABI call returned a non-zero status (mirrors the RuntimeError TORCH_CHECK raises)."""


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the gfx950 library has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
            "gaussianimage_plus_amd/csrc`). There is no CPU fallback for this path.")
    lib = C.CDLL(LIB_PATH)
    for name in STRING_FUNCS:
        getattr(lib, name).restype = C.c_char_p
    ver = lib.gi2d_version().decode()  # first: a development build is named before any of its symbols is looked up
    if "dev[" in ver and os.environ.get("GI2D_ALLOW_DEV_BUILD") != "1":
        raise RuntimeError(
            f"{LIB_PATH} is a development build ({ver}): cut-off / knock-out switches give wrong results on purpose. "
            "Rebuild with `make -C gaussianimage_plus_amd/csrc` (no EXTRA), or set GI2D_ALLOW_DEV_BUILD=1 for a "
            "measurement script that knows what it loads.")
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    for name, args in SIZE_FUNCS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_size_t
    _lib = lib
    return lib


def call(name: str, *args) -> None:
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.gi2d_last_error_string().decode(errors="replace")
        raise Gi2dError(f"{name} failed (status {rc}): {msg}")


def version() -> str:
    return load().gi2d_version().decode()


class TilePassTimers:
    """`count` kernel timers (gi2d_timer_*) for the tile-pass launches the calling thread issues next: launch
    0, stride, 2 stride, ... carries one (gi2d_timer_arm_many), also inside gi2d_train_steps[_batched] calls.  us() waits
    for the kernels and returns their own execution times -- the figure a rocprofv3 kernel trace reports."""

    def __init__(self, count: int):
        self.handles = []
        for _ in range(int(count)):
            h = C.c_void_p()
            call("gi2d_timer_create", C.byref(h))
            self.handles.append(h)
        self.array = (C.c_void_p * len(self.handles))(*[h.value for h in self.handles])

    def arm(self, stride: int = 1) -> None:
        call("gi2d_timer_arm_many", self.array, len(self.handles), int(stride))

    def cancel(self) -> None:
        call("gi2d_timer_arm_many", None, 0, 1)

    def us(self):
        out = []
        for h in self.handles:
            v = C.c_float()
            call("gi2d_timer_elapsed_us", h, C.byref(v))
            out.append(v.value)
        return out

    def close(self) -> None:
        self.cancel()  # a plan still armed with these handles must not outlive them
        for h in self.handles:
            load().gi2d_timer_destroy(h)
        self.handles = []
