#!/bin/bash
cd $GRAFT_REPO_ROOT
rm -f gaussianimage_plus_amd/csrc/gi2d_fast.o gaussianimage_plus_amd/csrc/gi2d_train.o
make -s -C gaussianimage_plus_amd/csrc EXTRA="-DGI2D_FUSED_TRACE $1" 2>&1 | grep -E "error"
python tools/trace_fused.py ${TRACE_N:-50000}
rm -f gaussianimage_plus_amd/csrc/gi2d_fast.o gaussianimage_plus_amd/csrc/gi2d_train.o; make -s -C gaussianimage_plus_amd/csrc
