#!/bin/bash
# Development aid: images/s of bench.py's per-image loop for several numbers of images fitted concurrently on one GPU.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for k in ${KS:-2 3 4 6 8}; do
  python bench.py --no-cpu-baseline --steps 50 --warmup 10 --images-per-gpu $k "$@" 2>/dev/null > /tmp/ipg.json
  python - <<PY
import json
d = json.loads(open("/tmp/ipg.json").read().strip().splitlines()[-1])["images_per_s"]
print("images per gpu $k:", round(d["value"], 3), "images/s, wall", round(d["wall_s"], 2), "s, psnr", round(d["avg_psnr"], 2))
PY
done
