#!/bin/bash
# Evidence for SURVEY 8f rank 3: the adaptive loop (prune every 100 iterations, growth every 1000) with the population
# count on the device issues no device->host copy inside the loop.  rocprofv3 memory-copy trace of one image.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/densify_trace
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$REPO:$PYTHONPATH
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT -o run -- python3 -m gaussianimage_plus_amd.launch --model covariance --synthetic 1 --num_points 5000 --max_num_points 50000 --iterations 10000 --grow_iter 1000 --prune_iter 100 > $OUT/launch.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(f"{out}/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    kinds = collections.Counter(r.get("Direction", r.get("Kind", "?")) for r in rows)
    print("memory copies by direction:", dict(kinds))
for f in glob.glob(f"{out}/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("prune_", "grow_", "train_reduce_update", "fast_fwdbwd", "fast_ws_init")):
            print("  ", r["Name"].split("(")[0][-48:], "calls", r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 2))
PY
grep -E "Average|image 0" $OUT/launch.log
