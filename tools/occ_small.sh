#!/bin/bash
# Development aid: step rate at N=10k / 50k for build variants given as arguments.
cd $GRAFT_REPO_ROOT
for CFG in "" "$@"; do
  rm -f gaussianimage_plus_amd/csrc/gi2d_fast.o gaussianimage_plus_amd/csrc/gi2d_train.o
  make -s -C gaussianimage_plus_amd/csrc EXTRA="$CFG" 2>&1 | grep -E "error"
  for N in 10000 50000; do
  python bench.py --steps 600 --warmup 50 --no-cpu-baseline --train-step --num-points $N 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('CFG=[$CFG] N=$N', 'us/step', round(d['ms_per_step']*1e3,2), 'it/s', round(d['value']), 'tile pass us', round(d['rasterize_pair']['fwdbwd_kernel_us'],2), 'train it/s', round(d['train_step']['iters_per_s']))"
  done
done
rm -f gaussianimage_plus_amd/csrc/gi2d_fast.o gaussianimage_plus_amd/csrc/gi2d_train.o; make -s -C gaussianimage_plus_amd/csrc
