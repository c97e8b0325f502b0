#!/bin/bash
# Development aid: several prebuilt libraries side by side on the timings that matter -- bench.py's headline loop (tile
# pass by events, update kernel = the rest), its `batched` block at K = 24, and (KODAK=1) the Kodak leg.
#   gpurun -- 'bash tools/ab_multi.sh product base product base'    ("product" = the library in the tree, any other name
#   = build/variants/<name>/libgi2d_hip.so, built before the snapshot was sent)
cd ${GRAFT_REPO_ROOT:-.}
A="--no-cpu-baseline --images 0 --no-batched --no-static --no-dropin"
run() {
  python bench.py $A 2>/dev/null | python -c "import json,sys; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  headline us/step', round(1e3*b['ms_per_step'],2), 'tile pass avg', round(b['roofline']['avg_kernel_us'],2), 'min', round(b['roofline']['min_kernel_us'],2))"
  python tools/batched_bench_scene.py 24 2>/dev/null | python -c "import json,sys; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['per_k'][0]; print('  K=24 tile pass per image', round(r['tile_pass_us_per_image'],2), 'iteration', round(r['us_per_image_iteration'],2), r['tile_pass_form'][:12])"
  if [ "$KODAK" = 1 ]; then python tools/kodak_fit.py 24 50000 3 2>&1 | tail -1 | cut -c1-110; fi
}
for name in "$@"; do
  if [ "$name" = product ]; then unset GI2D_LIB GI2D_ALLOW_DEV_BUILD; else export GI2D_LIB=$PWD/build/variants/$name/libgi2d_hip.so GI2D_ALLOW_DEV_BUILD=1; fi
  echo "== $name"; run
done
unset GI2D_LIB GI2D_ALLOW_DEV_BUILD
