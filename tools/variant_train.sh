#!/bin/bash
# Development aid: full training iteration time (bench.py --train-step) for csrc build variants.
# Usage: VARIANTS="'' '-DGI2D_NO_TILE_ORDER'" bash tools/variant_train.sh [bench args]
cd $GRAFT_REPO_ROOT
source tools/variant.sh
eval "set_variants=($VARIANTS)"
for v in "${set_variants[@]}"; do
  use_variant "$v"
  for rep in 1 2; do
    python3 bench.py --no-cpu-baseline --images 0 --train-step "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('variant $v: step us', round(d['ms_per_step']*1e3,2), ' train us/iter', round(d['train_step']['us_per_iter'],2))"
  done
done
use_product
