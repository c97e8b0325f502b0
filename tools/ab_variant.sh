#!/bin/bash
# Development aid: product library against one build variant ("$1" = EXTRA flags) on the three timings that matter --
# bench.py's headline loop, its `batched` block at K = 8 / 24, and the Kodak leg (24 images, three batches).
cd ${GRAFT_REPO_ROOT:-.}
. tools/variant.sh
A="--no-cpu-baseline --images 0 --no-batched --no-static --no-dropin"
run() {
  python bench.py $A 2>/dev/null | python -c "import json,sys; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  headline us/step', round(1e3*b['ms_per_step'],2), 'tile pass', round(b['roofline']['avg_kernel_us'],2))"
  python tools/batched_bench_scene.py 8 2>/dev/null | python -c "import json,sys; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['per_k'][0]; print('  K=8 tile pass per image', round(r['tile_pass_us_per_image'],2), 'iteration', round(r['us_per_image_iteration'],2))"
  python tools/batched_bench_scene.py 24 2>/dev/null | python -c "import json,sys; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['per_k'][0]; print('  K=24 tile pass per image', round(r['tile_pass_us_per_image'],2), 'iteration', round(r['us_per_image_iteration'],2))"
  python tools/kodak_fit.py 24 50000 3 2>&1 | tail -1 | cut -c1-110
}
echo "== product"; use_product; run
echo "== $1"; PREBUILT=${PREBUILT:-0} use_variant "$1"; run
use_product
