#!/bin/bash
# Development aid: VALU wave-instructions of the single-pass tile kernel cut off after each phase (GI2D_STOP_AFTER =
# 1 head, 2 forward loop, 3 pixel out + gradient + item scan, 4 backward items placed, 5 backward item loop, 0 = whole
# kernel) on the frozen bench scene (tools/static_steps.py) -- ONE counter pass per cut (no timing run).
#   PREBUILT=1 gpurun -- 'bash tools/phase_pmc.sh [N H W]'     (build the variants first: tools/variant.sh)
cd $GRAFT_REPO_ROOT
source tools/variant.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/phase_pmc
rm -rf $OUT && mkdir -p $OUT
for v in ${PHASES:-1 2 3 4 5 0}; do
  if [ "$v" = 0 ]; then use_product; else use_variant "-DGI2D_STOP_AFTER=$v $XFLAGS"; fi
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc$v -o run -- python3 $GRAFT_REPO_ROOT/tools/static_steps.py 20 "$@" > /dev/null 2> $OUT/pmc$v.log)
  python3 - $v $OUT <<'PY'
import csv, glob, sys, collections
v, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(f"{out}/pmc{v}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fast_fwdbwd" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"stop_after={v}: " + "  ".join(f"{k}={sum(x)/len(x)/1e6:.3f}M" for k, x in sorted(acc.items())), flush=True)
PY
done
use_product
