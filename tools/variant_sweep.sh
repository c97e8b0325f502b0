#!/bin/bash
# Development aid: tile-pass / end-of-step kernel time (rocprofv3 kernel trace) and step time of bench.py for csrc build
# variants.  Usage: VARIANTS="'' '-DGI2D_NO_TILE_ORDER' '-DGI2D_PG_BIG=128'" bash tools/variant_sweep.sh [bench args]
cd $GRAFT_REPO_ROOT
source tools/variant.sh
eval "set_variants=($VARIANTS)"
for v in "${set_variants[@]}"; do
  use_variant "$v"
  echo "variant: '$v'"
  bash tools/kernel_times.sh --images 0 "$@" 2>&1 | grep -E "fwdbwd|reduce_project|reduce_update"
  python3 bench.py --no-cpu-baseline --images 0 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   step us', round(d['ms_per_step']*1e3,2), ' tile pass (events) us', round(d['roofline']['avg_kernel_us'],2))"
done
use_product
