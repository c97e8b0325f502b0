#!/bin/bash
# Development aid: what the tile-ordering workgroup of the end-of-step kernel costs / buys (GI2D_NO_TILE_ORDER).
cd $GRAFT_REPO_ROOT
for v in "" "-DGI2D_NO_TILE_ORDER"; do
  rm -f gaussianimage_plus_amd/csrc/gi2d_fast.o gaussianimage_plus_amd/csrc/gi2d_train.o
  make -s -C gaussianimage_plus_amd/csrc EXTRA="$v" 2>&1 | grep -E "error"
  echo "variant: $v"
  bash tools/kernel_times.sh --images 0 "$@" 2>&1 | grep -E "fwdbwd|reduce_project"
  python3 bench.py --no-cpu-baseline --images 0 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   step us', round(d['ms_per_step']*1e3,2))"
done
rm -f gaussianimage_plus_amd/csrc/gi2d_fast.o gaussianimage_plus_amd/csrc/gi2d_train.o; make -s -C gaussianimage_plus_amd/csrc
