#!/usr/bin/env python3
"""Turn gpurun_out/rp (tools/profile_round12.sh) into the tracked evidence under profiles/:
    <round>_bench_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the default bench.py run
    <round>_bench_under_rocprof.json  the JSON line bench.py printed in that run
    <round>_bench_plain.json          the JSON line of a plain run (no profiler) in the same gpurun call
    <round>_pmc_summary.txt           mean counter value per kernel, one line per kernel and --pmc pass
    traffic.json                      HBM bytes per launch per kernel (read by bench.py for roofline.traffic)
usage: python tools/make_profiles12.py round2"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "rp")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "round2"


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in glob.glob(os.path.join(SRC, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if "gi2d" not in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = dict(vgpr=r["VGPR_Count"], agpr=r["Accum_VGPR_Count"], sgpr=r["SGPR_Count"],
                           lds=r["LDS_Block_Size"], scratch=r["Scratch_Size"])
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}, meta


stats = glob.glob(os.path.join(SRC, "stats", "**", "*kernel_stats.csv"), recursive=True)
assert stats, "no kernel_stats.csv under gpurun_out/rp/stats"
shutil.copy(stats[0], os.path.join(DST, f"{tag}_bench_kernel_stats.csv"))
for name in ("bench_under_rocprof.json", "bench_plain.json"):
    lines = [l for l in open(os.path.join(SRC, name)) if l.startswith("{")]
    open(os.path.join(DST, f"{tag}_{name}"), "w").write(lines[-1])

lines = []
traffic = {}
fetch, meta = counters("fetch")
write, _ = counters("write")
for sub in ("fetch", "write", "sq1", "sq2"):
    vals, m = counters(sub)
    for k in sorted(vals):
        lines.append(f"{sub:5s} {k:45s} regs {m[k]} " + " ".join(f"{c}={round(v)}" for c, v in sorted(vals[k].items())))
for k in fetch:
    f_kib, w_kib = fetch[k]["FETCH_SIZE"], write.get(k, {}).get("WRITE_SIZE", 0.0)
    traffic[k] = {"FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib,
                  # MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE reports half the bytes of wide coalesced reads on
                  # gfx950 -> doubled (upper bound for the narrower reads in these kernels); WRITE_SIZE is exact
                  "hbm_bytes_per_launch": int((2 * f_kib + w_kib) * 1024)}
open(os.path.join(DST, f"{tag}_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
bench = json.loads(open(os.path.join(DST, f"{tag}_bench_under_rocprof.json")).read())
json.dump({"source": f"profiles/{tag}_pmc_summary.txt", "config": {k: bench["config"][k] for k in
           ("num_points", "height", "width", "num_intersects_rank0")}, "kernels": traffic},
          open(os.path.join(DST, "traffic.json"), "w"), indent=1)
print("\n".join(lines))
print(open(os.path.join(DST, f"{tag}_bench_kernel_stats.csv")).read()[:1500])
