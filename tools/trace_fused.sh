#!/bin/bash
cd $GRAFT_REPO_ROOT
source tools/variant.sh
use_variant "-DGI2D_FUSED_TRACE $1"
python tools/trace_fused.py ${TRACE_N:-50000}
use_product
