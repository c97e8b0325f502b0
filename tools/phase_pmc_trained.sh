#!/bin/bash
# Development aid: tools/phase_pmc.sh on a TRAINED scene (tools/trained_scene.py: one Kodak picture fitted with the
# launcher's schedule, then frozen) -- VALU wave-instructions of the tile pass cut off after each phase.
#   PREBUILT=1 gpurun -- 'bash tools/phase_pmc_trained.sh [image] [iterations]'
cd $GRAFT_REPO_ROOT
source tools/variant.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/phase_pmc_trained
rm -rf $OUT && mkdir -p $OUT
use_product
python3 tools/trained_scene.py fit ${1:-0} ${2:-50000} /tmp/trained_scene.pt $FIT_ARGS || exit 1
for v in ${PHASES:-1 2 3 4 5 0}; do
  if [ "$v" = 0 ]; then use_product; else use_variant "-DGI2D_STOP_AFTER=$v $XFLAGS"; fi
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc$v -o run -- python3 $GRAFT_REPO_ROOT/tools/trained_scene.py steps 20 > /dev/null 2> $OUT/pmc$v.log)
  python3 - $v $OUT <<'PY'
import csv, glob, sys, collections
v, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(f"{out}/pmc{v}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fast_fwdbwd" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"stop_after={v}: " + "  ".join(f"{k}={sum(x)/len(x)/1e6:.3f}M" for k, x in sorted(acc.items())), flush=True)
PY
done
use_product
