#!/bin/bash
# Development aid: the batched tile pass in its general form against the hinted two-phase form (GI2D_BATCH_TILE_PASS):
# parity tests in both, then bench.py's `batched` block and the Kodak leg under either setting.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/ab_batch_form; mkdir -p $O
for f in ${FORMS:-two-phase general auto}; do
  GI2D_BATCH_TILE_PASS=$f timeout -k 10 600 python -m pytest tests/test_batched_gpu.py -x -q > $O/pytest_$f.log 2>&1 || { tail -30 $O/pytest_$f.log; exit 1; }
  tail -1 $O/pytest_$f.log
done
for f in general auto; do
  GI2D_BATCH_TILE_PASS=$f timeout -k 10 300 python - > $O/batched_$f.log 2>&1 <<PY || { tail -20 $O/batched_$f.log; exit 1; }
import bench, torch, json
for r in bench.batched_rate(50000, 512, 768, torch.device("cuda:0"), ks=(8, 24))["per_k"]:
    print(json.dumps({k: r[k] for k in ("images_per_launch", "us_per_image_iteration", "tile_pass_us_per_image")}))
PY
  echo "== $f"; cat $O/batched_$f.log
  GI2D_BATCH_TILE_PASS=$f timeout -k 10 300 python tools/kodak_fit.py 24 50000 3 2>&1 | tail -1 | tee $O/kodak_$f.log
done
