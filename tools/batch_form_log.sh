#!/bin/bash
# Development aid: how many tiles of a Kodak batch sit above the small form's capacity, call by call (GI2D_BATCH_LOG build).
cd ${GRAFT_REPO_ROOT:-.}
. tools/variant.sh
O=gpurun_out/ab_batch_form; mkdir -p $O
PREBUILT=1 use_variant "-DGI2D_BATCH_LOG"
timeout -k 10 300 python tools/kodak_fit.py 24 50000 3 > $O/kodak_log.out 2> $O/kodak_log.err
tail -1 $O/kodak_log.out
python - <<PY
import re, collections
rows = [tuple(map(int, re.findall(r"tiles (\d+).*report: (-?\d+)", l)[0])) for l in open("$O/kodak_log.err") if "gi2d batch" in l]
n = len(rows)
print("calls", n)
for i in range(0, n, max(n // 40, 1)):
    t, b = rows[i]
    print(i, t, b, f"{b / t:.3f}")
PY
use_product
FORMS=auto bash tools/ab_batch_form.sh
