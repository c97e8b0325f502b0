#!/bin/bash
# Development aid: rocprofv3 kernel time of the fused forward with and without its pixel loop.
cd $GRAFT_REPO_ROOT
REPO=$GRAFT_REPO_ROOT
for CFG in "" "$@"; do
  rm -f gaussianimage_plus_amd/csrc/gi2d_fast.o
  make -s -C gaussianimage_plus_amd/csrc EXTRA="$CFG" 2>&1 | grep -E "error"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/abl && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl -o run -- python3 $REPO/bench.py --no-cpu-baseline --steps 100 > /dev/null 2>&1)
  echo "CFG=[$CFG]"; python3 - <<'PY'
import csv,glob
for f in glob.glob('/tmp/abl/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gi2d::fast' in r['Name']: print('  ', r['Name'].split('(')[0][-36:], r['Calls'], round(float(r['AverageNs'])/1e3,2), int(r['MinNs'])/1e3)
import collections
for f in glob.glob('/tmp/abl/**/*kernel_trace.csv', recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if 'fast_fwdbwd_kernel' in r['Kernel_Name']]
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
    a,b=d[20:120],d[170:]
    print('   fwdbwd hotpath avg', round(sum(a)/len(a),2), 'min', min(a), '| train avg', round(sum(b)/len(b),2), 'min', min(b))
PY
done
rm -f gaussianimage_plus_amd/csrc/gi2d_fast.o; make -s -C gaussianimage_plus_amd/csrc
