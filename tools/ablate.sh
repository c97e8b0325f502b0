#!/bin/bash
# Development aid: time the fused backward tile kernel with successive phases enabled.
cd $GRAFT_REPO_ROOT
for CFG in "-DGI2D_ABLATE_BWD_LEVEL=0" "-DGI2D_ABLATE_BWD_LEVEL=1" "-DGI2D_ABLATE_BWD_LEVEL=2" "-DGI2D_ABLATE_BWD_LEVEL=3" "-DGI2D_ABLATE_BWD_LEVEL=3 -DGI2D_ABLATE_BWD_COMPUTE" "-DGI2D_ABLATE_BWD_LEVEL=4" "-DGI2D_ABLATE_BWD_COMPUTE" ""; do
  rm -f gaussianimage_plus_amd/csrc/gi2d_fast.o
  make -s -C gaussianimage_plus_amd/csrc EXTRA="$CFG" 2>&1 | grep -E "error"
  python bench.py --steps 400 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('CFG=[$CFG]', 'us/step', round(d['ms_per_step']*1e3,2), 'fwd_us', round(d['rasterize_pair']['fwd_kernel_us'],2), 'bwd_us', round(d['rasterize_pair']['bwd_tile_kernel_us'],2))"
done
rm -f gaussianimage_plus_amd/csrc/gi2d_fast.o; make -s -C gaussianimage_plus_amd/csrc
