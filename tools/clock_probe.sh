#!/bin/bash
# Development aid: the shader clock the tile pass actually runs at -- GRBM_GUI_ACTIVE (cycles the GPU was busy) of
# each dispatch against its duration in the kernel trace of the same run -- for 24 images per launch and for one image.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/cp1 /tmp/cp2
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/cp1 -o run -- python3 $R/tools/batch_time.py ${ARGS:-50000 512 768 cholesky 24} > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/cp2 -o run -- python3 $R/tools/static_steps.py 30 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for d, tag in (("/tmp/cp1", "batched/moving"), ("/tmp/cp2", "frozen")):
    dur = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            if "fwdbwd" in k:
                acc[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
                    acc[(k, r["Grid_Size"])]["ns"].append(dur[r["Dispatch_Id"]][0])
    for (k, g), c in sorted(acc.items()):
        if len(c["ns"]) >= 10:
            ns = sum(c["ns"]) / len(c["ns"])
            cyc = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
            print(tag, k, "grid", g, f"kernel {ns/1e3:.1f} us (under counters)  GRBM_GUI_ACTIVE {cyc/1e3:.1f} kcycles -> {cyc/ns:.2f} GHz  ",
                  " ".join(f"{n}={sum(v)/len(v)/1e6:.3f}M" for n, v in sorted(c.items()) if n not in ("ns",)))
PY
