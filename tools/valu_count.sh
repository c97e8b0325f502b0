#!/bin/bash
# Development aid: VALU wave-instructions and kernel time of the tile pass for the build in the tree -- 24 images per
# launch and single-image calls on moving gaussians (tools/batch_time.py), frozen scene (tools/static_steps.py).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/vc1 /tmp/vc2
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/vc1 -o run -- python3 $R/tools/batch_time.py ${ARGS:-50000 512 768 cholesky 24} > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/vc2 -o run -- python3 $R/tools/static_steps.py 30 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for d, tag in (("/tmp/vc1", "moving"), ("/tmp/vc2", "frozen")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            if "fwdbwd" in k or "reduce_update" in k:
                acc[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k, g), c in sorted(acc.items()):
        if len(c["SQ_INSTS_VALU"]) >= 20:
            print(tag, k, "grid", g, " ".join(f"{n}={sum(v)/len(v)/1e6:.3f}M" for n, v in sorted(c.items())))
PY
