#!/bin/bash
# Development aid: SQ counters of the batched tile pass (tools/batch_time.py under rocprofv3 --pmc).
# usage: gpurun -- 'ARGS="50000 512 768 cholesky 24" bash tools/pmc_batched.sh'
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_batched
rm -rf $OUT && mkdir -p $OUT
A=${ARGS:-50000 512 768 cholesky 24}
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq1 -o run -- python3 $GRAFT_REPO_ROOT/tools/batch_time.py $A > /dev/null 2> $OUT/sq1.log
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/sq2 -o run -- python3 $GRAFT_REPO_ROOT/tools/batch_time.py $A > /dev/null 2> $OUT/sq2.log
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("sq1", "sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-50:]
            if "gi2d" in k:
                acc[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k, g), d in sorted(acc.items()):
        if len(next(iter(d.values()))) < 50:
            continue
        print(k, "grid", g, "  ".join(f"{c}={sum(v)/len(v)/1e6:.3f}M" for c, v in sorted(d.items())))
PY
