#!/bin/bash
# Step rate of bench.py at the BASELINE.json sizes (768x512: N = 2.5k, 10k, 30k, 50k; 2040x1356: N = 50k).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for CFG in "2500 512 768" "10000 512 768" "30000 512 768" "50000 512 768" "50000 1356 2040"; do
  set -- $CFG
  python bench.py --steps 400 --warmup 40 --no-cpu-baseline --images 0 --train-step --num-points $1 --height $2 --width $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('N=$1 $3x$2 M=%d: %.1f us/step (%d steps/s), tile pass %.1f us, %.0f GB/s algorithmic (frac %.3f), train %.1f us/iter' % (d['config']['num_intersects_rank0'], d['ms_per_step']*1e3, d['value'], r['avg_kernel_us'], r['achieved'], r['frac'], d['train_step']['us_per_iter']))"
done
