#!/bin/bash
# Development aid: where a Kodak fit's GPU time goes along the schedule -- tools/kodak_fit.py 24 50000 1 (ONE batch of 24 on
# one stream, so a launch's duration is its cost) under the kernel trace, tile pass (by kernel form) and update kernel
# per 5000 iterations.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kcc
rocprofv3 --kernel-trace --output-format csv -d /tmp/kcc -o run -- python3 $R/tools/kodak_fit.py 24 ${ITERS:-50000} 1 2>&1 | grep "mode"
python3 - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob("/tmp/kcc/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gi2d::" in n:
            rows.append((int(r["Start_Timestamp"]), n.split("(")[0].replace("void gi2d::", "").replace("gi2d::", ""),
                         int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
# iteration index = number of update kernels seen so far
it, per = 0, 5000
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))
for t, n, d in rows:
    b = it // per
    acc[n][b][0] += d
    acc[n][b][1] += 1
    if "reduce_update" in n:
        it += 1
nb = it // per + 1
tot = collections.defaultdict(float)
for n in sorted(acc):
    line = []
    for b in range(nb):
        s, c = acc[n][b]
        tot[b] += s
        line.append(f"{s / 1e3 / per:7.1f}")
    print(f"{n[:58]:58s}", " ".join(line))
print(f"{'ALL gi2d kernels, us per iteration (24 images)':58s}", " ".join(f"{tot[b] / 1e3 / per:7.1f}" for b in range(nb)))
PY
