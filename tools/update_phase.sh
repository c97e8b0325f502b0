#!/bin/bash
# Development aid: where the single-image update kernel's time goes -- the kernel cut off after its two load rounds
# (GI2D_UPDATE_STOP=1), after projection backward + optimizer update (2), after the next iteration's projection (3), after the record (6),
# against the whole kernel, under rocprofv3's kernel trace on one box (the cut-off builds store nothing, so the scene
# stands still: only the update kernel's own time is meaningful).  DESIGN.md 3.5.
#   for v in 1 2 3 6; do make -C gaussianimage_plus_amd/csrc VARIANT=ustop$v EXTRA=-DGI2D_UPDATE_STOP=$v; done
#   gpurun -- 'bash tools/update_phase.sh'
cd ${GRAFT_REPO_ROOT:-.}
STEPS=${STEPS:-300} bash tools/ab_trace.sh product ustop1 ustop2 ustop3 ustop6
