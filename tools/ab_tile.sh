#!/bin/bash
# Development aid: tile pass / update kernel times (kernel trace) of the frozen scene, the training loop and a 24-image
# batch for the build in the tree.  usage: gpurun -- 'bash tools/ab_tile.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt1 /tmp/kt2
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt1 -o run -- python3 $R/tools/static_steps.py 200 > /dev/null 2>&1
echo "frozen scene (HotPath.step):"; python3 $R/tools/trace_by_grid.py /tmp/kt1 | grep "gi2d::" | head -3
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -o run -- python3 $R/tools/batch_time.py ${ARGS:-50000 512 768 cholesky 24} 2>&1 | grep "K=\|single"
echo "training (24 per launch, then single-image calls):"; python3 $R/tools/trace_by_grid.py /tmp/kt2 | grep "gi2d::" | head -4
