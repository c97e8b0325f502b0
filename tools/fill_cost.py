"""Development aid: what the bucket fill costs inside the end-of-step kernel -- the same kernel with and without the
project+fill tail, each preceded by a cursor reset, under rocprofv3 (tools/fill_cost.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import synth_cholesky, synth_gt
from gaussianimage_plus_amd import _lib
from gaussianimage_plus_amd.hotpath import HotPath
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
hp = HotPath(n, 512, 768, device="cuda:0")
hp.set_inputs(*synth_cholesky(n, 512, 768, 3047))
hp.set_target(torch.from_numpy(synth_gt(512, 768, 1)).to("cuda:0"))
for _ in range(10):
    hp.step()
torch.cuda.synchronize()
st = hp._stream()
for it in range(200):
    _lib.call("gi2d_fast_workspace_init", hp.ws.data_ptr(), hp.ws.numel(), hp.n, hp.tx, hp.ty, st)
    hp._run(hp._f_red_next, st)     # reduce + project bwd + project + fill
    hp._run(hp._f_red, st)          # reduce + project bwd
    _lib.call("gi2d_fast_workspace_init", hp.ws.data_ptr(), hp.ws.numel(), hp.n, hp.tx, hp.ty, st)
    hp._run(hp._f_bin, st)          # project + fill alone
torch.cuda.synchronize()
print("done")
