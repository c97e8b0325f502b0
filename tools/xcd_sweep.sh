#!/bin/bash
# Development aid: batched training (tools/batch_time.py) for csrc build variants.
# usage: VARIANTS="'' '-DGI2D_NO_XCD_MAP'" ARGS="50000 512 768 cholesky 8 24" bash tools/xcd_sweep.sh
cd $GRAFT_REPO_ROOT
source tools/variant.sh
eval "set_variants=($VARIANTS)"
for v in "${set_variants[@]}"; do
  use_variant "$v"
  echo "variant: '$v'"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o run -- python3 $GRAFT_REPO_ROOT/tools/batch_time.py ${ARGS:-50000 512 768 cholesky 8 24} 2>&1 | grep "K=")
  python3 tools/trace_by_grid.py /tmp/kt | grep "gi2d::" | head -8
done
use_product
