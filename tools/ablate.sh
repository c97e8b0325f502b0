#!/bin/bash
# Development aid: time the rasterizer kernels with their pixel loops removed (staging-only cost).
cd $GRAFT_REPO_ROOT
for CFG in "-DGI2D_ABLATE_BWD_COMPUTE -DGI2D_ABLATE_FWD_COMPUTE" ""; do
  rm -f gaussianimage_plus_amd/csrc/gi2d_raster.o
  make -s -C gaussianimage_plus_amd/csrc EXTRA="$CFG" 2>&1 | grep -E "error"
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('CFG=[$CFG]', 'iters/s', round(d['value']), 'fwd_us', round(d['rasterize_pair']['fwd_kernel_us'],2), 'bwd_us', round(d['rasterize_pair']['bwd_tile_kernel_us'],2))"
done
