#!/bin/bash
# Development aid: workgroup size of the one-lane-per-gaussian kernels above 32768 gaussians.
cd $GRAFT_REPO_ROOT
for B in 256 128 64; do
  rm -f gaussianimage_plus_amd/csrc/*.o
  make -s -C gaussianimage_plus_amd/csrc EXTRA="-DGI2D_PG_BIG=$B" 2>&1 | grep -E "error"
  for i in 1 2; do
    python bench.py --train-step --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('block $B: step', round(d['ms_per_step']*1e3,2), 'us, train', round(d['train_step']['us_per_iter'],2), 'us')"
  done
done
rm -f gaussianimage_plus_amd/csrc/*.o; make -s -C gaussianimage_plus_amd/csrc
