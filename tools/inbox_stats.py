"""Development aid: how the entered tiles of the bench scene's update kernel are served -- through the tiles' inboxes or
through the row headers' returning atomics (csrc/gi2d_fast_internal.h::Inbox).  Needs a library built with
-DGI2D_INBOX_STATS (make VARIANT=inbox_stats EXTRA=-DGI2D_INBOX_STATS) selected by GI2D_LIB / GI2D_ALLOW_DEV_BUILD=1:
    gpurun -- 'GI2D_LIB=$PWD/build/variants/inbox_stats/libgi2d_hip.so GI2D_ALLOW_DEV_BUILD=1 python tools/inbox_stats.py'
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from helpers import synth_cholesky, synth_gt  # noqa: E402
import bench  # noqa: E402
from gaussianimage_plus_amd import _lib  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
if len(sys.argv) > 2 and sys.argv[2] == "dense":  # the scene of tests/test_batched_gpu.py's BIG case: 39 gaussians per tile, lr 0.01
    from gaussianimage_plus_amd.launch import synthetic_image  # noqa: E402
    from gaussianimage_plus_amd.trainer import NativeFitter  # noqa: E402
    n, h, w = 15000, 256, 384
    fit = NativeFitter(synthetic_image(h, w, 60).cuda(), n, kind="cholesky", lr=0.01, seed=11)
elif len(sys.argv) > 2 and sys.argv[2] == "wild":  # tests/test_batched_gpu.py::_wild_fitter: every way into a tile
    import test_batched_gpu  # noqa: E402
    fit = test_batched_gpu._wild_fitter()
    n = fit.n
else:
    n, h, w = 50000, 512, 768
    xyz, L, col, op = synth_cholesky(n, h, w, 3047)
    gt = torch.from_numpy(synth_gt(h, w, 1)).cuda()
    fit = bench.make_fitter(gt, xyz, L, col, n, h, w)
fit.max_call = 1 << 30
fit.train(2 if n < 50000 else 50)
torch.cuda.synchronize()
import ctypes  # noqa: E402
gp, bp = ctypes.c_void_p(), ctypes.c_void_p()
_lib.call("gi2d_fast_workspace_views", fit.ws.data_ptr(), fit.ws.numel(), fit.cap, fit.tx, fit.ty, ctypes.byref(gp),
          ctypes.byref(bp))
off = gp.value - fit.ws.data_ptr()  # the tile rows: the counters sit in the padding of row 0's header
words = fit.ws[off:off + 64].view(torch.int32)
before = words[:16].clone()
fit.train(steps)
torch.cuda.synchronize()
d = (words[:16] - before).tolist()
print(_lib.load().gi2d_version().decode())
print(f"per step over {steps} steps:")
print(f"  lanes whose box changed        {d[13] / steps:9.1f}   (of {n})")
print(f"    ... with no rank to go by    {d[14] / steps:9.1f}")
print(f"  waves holding such a lane      {d[15] / steps:9.1f}   (of {(n + 63) // 64})")
print(f"  entered tiles via the inbox    {d[9] / steps:9.1f}")
print(f"  entered tiles via the header   {d[10] / steps:9.1f}   (not a neighbour {d[11] / steps:.1f}, neighbour without rank {d[12] / steps:.1f})")
print(f"  waves waiting for an atomic    {d[7] / steps:9.1f}")
print(f"  entrants taken in by the tiles {(d[5] + d[6]) / steps:9.1f}   (second and later ones of a lane's eight slots: {d[6] / steps:.1f})")
