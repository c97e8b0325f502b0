#!/bin/bash
# Development aid: step time of bench.py (N=50k and N=10k) for whole-library build variants.
cd $GRAFT_REPO_ROOT
for CFG in "" "$@"; do
  make -s -C gaussianimage_plus_amd/csrc clean; make -s -C gaussianimage_plus_amd/csrc EXTRA="$CFG" 2>&1 | grep -E "error"
  for N in 50000 10000; do
    python bench.py --steps 600 --warmup 50 --no-cpu-baseline --train-step --num-points $N 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('CFG=[$CFG] N=$N: %.2f us/step, tile pass %.2f, train %.2f us/iter' % (d['ms_per_step']*1e3, d['roofline']['avg_kernel_us'], d['train_step']['us_per_iter']))"
  done
done
make -s -C gaussianimage_plus_amd/csrc clean; make -s -C gaussianimage_plus_amd/csrc
