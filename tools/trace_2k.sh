#!/bin/bash
# Development aid: phase timeline of the tile pass at 2040x1356 (BASELINE config 4).
cd $GRAFT_REPO_ROOT
source tools/variant.sh
use_variant "-DGI2D_FUSED_TRACE $1"
python tools/trace_fused.py 50000 1356 2040 | head -16
use_product
