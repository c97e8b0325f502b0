#!/bin/bash
# Development aid: tile-pass kernel time (rocprofv3 kernel trace) and HBM counters for build variants.
# usage: bash tools/variants.sh "<EXTRA flags A>" "<EXTRA flags B>" ...   (the default build is always measured first)
cd $GRAFT_REPO_ROOT
REPO=$GRAFT_REPO_ROOT
for CFG in "" "$@"; do
  rm -f gaussianimage_plus_amd/csrc/gi2d_fast.o gaussianimage_plus_amd/csrc/gi2d_train.o
  make -s -C gaussianimage_plus_amd/csrc EXTRA="$CFG" 2>&1 | grep -E "error"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/var && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/var/s -o run -- python3 $REPO/bench.py --no-cpu-baseline --steps 160 > /tmp/var_bench.json 2>/dev/null; \
   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/var/f -o run -- python3 $REPO/bench.py --no-cpu-baseline --steps 16 --warmup 4 > /dev/null 2>&1; \
   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/var/w -o run -- python3 $REPO/bench.py --no-cpu-baseline --steps 16 --warmup 4 > /dev/null 2>&1)
  echo "CFG=[$CFG]"; python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('/tmp/var/s/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gi2d::' in r['Name']: print('  ', r['Name'].split('(')[0][-40:], r['Calls'], 'avg', round(float(r['AverageNs'])/1e3, 2), 'min', int(r['MinNs'])/1e3)
for d, c in (('f', 'FETCH_SIZE'), ('w', 'WRITE_SIZE')):
    acc = collections.defaultdict(list)
    for f in glob.glob(f'/tmp/var/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'gi2d::' in r['Kernel_Name']: acc[r['Kernel_Name'].split('(')[0][-40:]].append(float(r['Counter_Value']))
    print('  ', c, {k: round(sum(v)/len(v)) for k, v in acc.items()})
PY
done
rm -f gaussianimage_plus_amd/csrc/gi2d_fast.o gaussianimage_plus_amd/csrc/gi2d_train.o; make -s -C gaussianimage_plus_amd/csrc
