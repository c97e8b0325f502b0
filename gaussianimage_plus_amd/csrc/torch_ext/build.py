"""Builds the compiled op table (gi2d_torch_ext.cpp -> gaussianimage_plus_amd/_gi2d_torch.so) in-tree with g++: the
file holds no device code, only libtorch <-> C-ABI glue, and links libgi2d_hip.so through $ORIGIN.
    python gaussianimage_plus_amd/csrc/torch_ext/build.py     (called by __graft_entry__.build())"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(PKG, "_gi2d_torch.so")
SRC = os.path.join(HERE, "gi2d_torch_ext.cpp")


def up_to_date() -> bool:
    deps = [SRC, os.path.join(PKG, "..", "include", "gi2d.h"), os.path.join(PKG, "libgi2d_hip.so"), __file__]
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps if os.path.exists(d))


def build(force: bool = False) -> str:
    if up_to_date() and not force:
        return OUT
    import torch
    from torch.utils import cpp_extension as ce
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-w",
           "-DTORCH_EXTENSION_NAME=_gi2d_torch", "-DTORCH_API_INCLUDE_EXTENSION_H", "-D__HIP_PLATFORM_AMD__=1",
           "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           "-I" + os.path.join(PKG, "..", "include"), "-I" + sysconfig.get_paths()["include"],
           "-I" + os.path.join(rocm, "include")]
    cmd += ["-I" + p for p in ce.include_paths()]
    cmd += [SRC, "-o", OUT, "-L" + tlib, "-L" + PKG, "-l:libgi2d_hip.so", "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu",
            "-ltorch_hip", "-ltorch_python", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
