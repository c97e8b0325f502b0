#!/bin/bash
# Development aid: build the rasterizer with different occupancy/unroll hints and time the bench.
cd $GRAFT_REPO_ROOT
for CFG in "6 1" "5 1" "5 2" "4 2"; do
  set -- $CFG
  rm -f gaussianimage_plus_amd/csrc/gi2d_raster.o
  make -s -C gaussianimage_plus_amd/csrc EXTRA="-DGI2D_BWD_OCC=$1 -DGI2D_BWD_UNROLL=$2" 2>&1 | grep -E "error"
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('OCC=$1 UNROLL=$2', 'iters/s', round(d['value']), 'fwd_us', round(d['rasterize_pair']['fwd_kernel_us'],2), 'bwd_us', round(d['rasterize_pair']['bwd_tile_kernel_us'],2))"
done
rm -f gaussianimage_plus_amd/csrc/gi2d_raster.o; make -s -C gaussianimage_plus_amd/csrc
