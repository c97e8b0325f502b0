#!/bin/bash
# Development aid: time of the training update kernel with parts knocked out (GI2D_RU_KNOCK bits: 1 no best-model
# decision, 2 no gradient gather, 4 next iteration not prepared; wrong results, timing only).
cd $GRAFT_REPO_ROOT
source tools/variant.sh
for v in ${VARIANTS:-0 1 2 4 3 7}; do
  use_variant "-DGI2D_RU_KNOCK=$v"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o run -- python3 $GRAFT_REPO_ROOT/tools/batch_time.py ${ARGS:-50000 512 768 cholesky 1} > /dev/null 2>&1)
  echo "knock=$v"
  python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gi2d::" in r["Name"] and int(r["Calls"]) > 100:
            print("  ", r["Name"].split("(")[0][-52:], r["Calls"], "avg", round(float(r["AverageNs"]) / 1e3, 2), "min", int(r["MinNs"]) / 1e3)
PY
done
use_product
