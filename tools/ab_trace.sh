#!/bin/bash
# Development aid: the headline loop's two kernels under rocprofv3's kernel trace, for several prebuilt libraries on ONE
# box ("product" = the library in the tree, any other name = build/variants/<name>/libgi2d_hip.so).
#   gpurun -- 'bash tools/ab_trace.sh base product'          (BATCH=24: tools/batched_bench_scene.py 24 instead)
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
A="--no-cpu-baseline --images 0 --no-batched --no-static --no-dropin --steps ${STEPS:-500}"
if [ -n "$C4" ]; then A="$A --height 1356 --width 2040"; fi   # C4=1: BASELINE config 4
mkdir -p gpurun_out/ab_trace
for name in "$@"; do
  if [ "$name" = product ]; then unset GI2D_LIB GI2D_ALLOW_DEV_BUILD; else export GI2D_LIB=$PWD/build/variants/$name/libgi2d_hip.so GI2D_ALLOW_DEV_BUILD=1; fi
  d=gpurun_out/ab_trace/$name; rm -rf $d
  if [ -n "$BATCH" ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -o run -- python3 tools/batched_bench_scene.py $BATCH > $d.out 2> $d.log
  else
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -o run -- python3 bench.py $A > $d.out 2> $d.log
  fi
  f=$(find $d -name "*kernel_stats.csv" 2>/dev/null | sort | sed -n 1p)
  echo "== $name"
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    print("  %-60s calls %6s  avg %7.2f us  min %7.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
  else echo "  (no kernel stats: see $d.log)"; fi
done
unset GI2D_LIB GI2D_ALLOW_DEV_BUILD
