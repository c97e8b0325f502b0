#!/bin/bash
# Development aid: what the gaussians on more than 32 tiles cost the per-gaussian kernels of a TRAINED scene -- the
# end-of-step kernel with and without their (wave-strided) sums (-DGI2D_RU_KNOCK=64: wrong results, timing only).
cd $GRAFT_REPO_ROOT
source tools/variant.sh
python3 tools/trained_scene.py fit ${1:-0} ${2:-50000} /tmp/trained_scene.pt > /dev/null 2>&1
for X in "" "-DGI2D_RU_KNOCK=64"; do
  use_variant "$X"
  echo "== build '$X'"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/k1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k1 -o run -- python3 $GRAFT_REPO_ROOT/tools/trained_scene.py steps 100 2>&1 | grep "per step"; python3 $GRAFT_REPO_ROOT/tools/trace_by_grid.py /tmp/k1 | grep gi2d | head -2)
done
use_product
