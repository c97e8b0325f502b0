#!/bin/bash
# Development aid: workgroup size of the per-gaussian kernels.
cd $GRAFT_REPO_ROOT
for B in 256 128 64; do
  rm -f gaussianimage_plus_amd/csrc/gi2d_fast.o
  make -s -C gaussianimage_plus_amd/csrc EXTRA="-DGI2D_PER_GAUSSIAN_BLOCK=$B" 2>&1 | grep -E "error"
  for N in 2500 50000; do echo "block $B N=$N"; bash tools/kernel_times.sh --num-points $N | grep "project"; done
done
rm -f gaussianimage_plus_amd/csrc/gi2d_fast.o; make -s -C gaussianimage_plus_amd/csrc
