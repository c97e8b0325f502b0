#!/bin/bash
# Development aid: rocprofv3 kernel-trace averages of one bench.py run; arguments are passed to bench.py.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o run -- python3 $REPO/bench.py --no-cpu-baseline "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gi2d::" in r["Name"]:
            print("  ", r["Name"].split("(")[0][-44:], r["Calls"], "avg", round(float(r["AverageNs"]) / 1e3, 2), "min", int(r["MinNs"]) / 1e3)
PY
