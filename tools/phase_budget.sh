#!/bin/bash
# Development aid: VALU instruction count / busy cycles / time of the single-pass tile kernel cut off after each phase
# (GI2D_STOP_AFTER = 1 head, 2 forward loop, 3 pixel out + gradient, 4 backward items built, 5 backward item loop,
# 0 = whole kernel).
cd $GRAFT_REPO_ROOT
source tools/variant.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/phase_budget
rm -rf $OUT && mkdir -p $OUT
ARGS="$*"
for v in ${PHASES:-1 2 3 4 5 0}; do
  use_variant "-DGI2D_STOP_AFTER=$v"
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc$v -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --images 0 $ARGS > /dev/null 2> $OUT/pmc$v.log)
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st$v -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --images 0 $ARGS > /dev/null 2> $OUT/st$v.log)
  python3 - $v $OUT <<'PY'
import csv, glob, sys, collections
v, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(f"{out}/pmc{v}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fast_fwdbwd" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
t = None
for f in glob.glob(f"{out}/st{v}/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fast_fwdbwd" in r["Name"]:
            t = float(r["AverageNs"]) / 1e3
print(f"stop_after={v}: {t:.2f} us  " + "  ".join(f"{k}={sum(x)/len(x)/1e6:.3f}M" for k, x in sorted(acc.items())))
PY
done
use_product
