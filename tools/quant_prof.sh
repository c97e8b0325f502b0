#!/bin/bash
# Development aid: rocprofv3 kernel-trace averages of the quantisation-aware iteration (tools/quant_time.py).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/qprof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o q -- python3 $REPO/tools/quant_time.py "$@" > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gi2d::" in r["Name"]:
            print("  ", r["Name"].split("(")[0][-52:], r["Calls"], "avg", round(float(r["AverageNs"]) / 1e3, 2), "min", int(r["MinNs"]) / 1e3)
PY
