#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fc
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fc -o run -- python3 $REPO/tools/fill_cost.py "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/fc/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gi2d::" in r["Name"] and int(r["Calls"]) >= 100:
            print("  ", r["Name"].split("(")[0][-48:], r["Calls"], "avg", round(float(r["AverageNs"]) / 1e3, 2), "min", int(r["MinNs"]) / 1e3)
PY
