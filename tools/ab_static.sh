#!/bin/bash
# Development aid: frozen scenes (no optimizer: a knock-out build's wrong gradients do not feed back) under the kernel
# trace, for several prebuilt libraries: the bench scene (tools/static_steps.py) and a Kodak picture after ITERS
# iterations of the schedule (tools/trained_scene.py), HotPath.step = fast_fwdbwd_kernel + fast_reduce_project_kernel.
#   gpurun -- 'ITERS=12000 bash tools/ab_static.sh product <variant> ...'
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
unset GI2D_LIB GI2D_ALLOW_DEV_BUILD
python3 tools/trained_scene.py fit 0 ${ITERS:-12000} /tmp/trained_scene.pt 5000 50000 2>&1 | tail -1
show() { f=$(find $1 -name "*kernel_stats.csv" | sort | sed -n 1p); python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:2]:
    print("    %-60s calls %6s  avg %7.2f us  min %7.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
}
for name in "$@"; do
  if [ "$name" = product ]; then unset GI2D_LIB GI2D_ALLOW_DEV_BUILD; else export GI2D_LIB=$PWD/build/variants/$name/libgi2d_hip.so GI2D_ALLOW_DEV_BUILD=1; fi
  echo "== $name"
  d=/tmp/abs_$name; rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d/bench -o run -- python3 tools/static_steps.py 300 > /dev/null 2>&1
  echo "  bench scene:"; show $d/bench
  rocprofv3 --kernel-trace --stats --output-format csv -d $d/kodak -o run -- python3 tools/trained_scene.py steps 300 > /dev/null 2>&1
  echo "  Kodak scene:"; show $d/kodak
done
