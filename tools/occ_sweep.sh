#!/bin/bash
# Development aid: residency of the single-pass tile kernel (workgroups per CU) against register spills, at 2040x1356
# (10880 tiles: several rounds of resident workgroups) and at 768x512 (1536 tiles: one round).
cd $GRAFT_REPO_ROOT
source tools/variant.sh
for v in "-DGI2D_FUSED_OCC=6" "-DGI2D_FUSED_OCC=7 -DGI2D_FWD_CHUNK=16" "-DGI2D_FUSED_OCC=8 -DGI2D_FWD_CHUNK=16 -DGI2D_BWD_PART_ROWS=64"; do
  use_variant "$v"
  echo "variant: $v"
  for CFG in "50000 1356 2040" "50000 512 768" "10000 512 768"; do
    set -- $CFG
    python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --images 0 --num-points $1 --height $2 --width $3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('   N=$1 $3x$2: %.1f us/step, tile pass %.1f us' % (d['ms_per_step']*1e3, r['avg_kernel_us']))"
  done
done
use_product
