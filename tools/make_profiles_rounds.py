#!/usr/bin/env python3
"""Turn gpurun_out/rp<N> (tools/profile_rounds.sh; N = the ROUND environment variable, default 5) into the tracked
evidence under profiles/:
    round3_<workload>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the workload (head, c4_2k, batched,
                                         c5_rs_quant, trained_fit)
    round3_bench_under_rocprof.json      the JSON line bench.py printed under the profiler (head workload)
    round3_bench_plain.json              the JSON line of the plain default run in the same gpurun call
    round3_batched_plain.txt             tools/batch_time.py K = 4 / 8 / 24 without the profiler
    round3_pmc_summary.txt               mean counter value per kernel, one line per workload, kernel and --pmc pass
    traffic.json                         HBM bytes per launch per kernel and workload (read by bench.py for
                                         roofline.traffic), the VALU counters of the same kernels (roofline_valu) and the
                                         lane model of the bench scene (tools/lane_model.py)
    round4_dropin_profile_after.txt      the drop-in autograd loop under cProfile
    round4_kodak_fit_50k_run.txt         the images/s leg alone (24 Kodak images x 50 000 iterations)
usage: python tools/make_profiles_rounds.py"""
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("ROUND", "6")
SRC = os.path.join(ROOT, "gpurun_out", "rp" + ROUND)
DST = os.path.join(ROOT, "profiles")
TAG = "round" + ROUND
NAMES = {"head": "bench", "c4": "c4_2k", "batched": "batched", "c5": "c5_rs_quant", "fit": "trained_fit",
         "frozen": "frozen_scene", "trained": "trained_scene", "batched4": "batched_k4", "batched8": "batched_k8"}


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def counters(work, sub):
    """mean per launch by (kernel, grid): the batched kernels run on several grid sizes in one program."""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in glob.glob(os.path.join(SRC, work, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if "gi2d" not in k:
                continue
            key = (k, int(r["Grid_Size"]))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[key] = dict(vgpr=r["VGPR_Count"], agpr=r["Accum_VGPR_Count"], sgpr=r["SGPR_Count"],
                             lds=r["LDS_Block_Size"], scratch=r["Scratch_Size"], launches=0)
    for key, d in acc.items():
        meta[key]["launches"] = max(len(v) for v in d.values())
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}, meta


lines, workloads = [], []
for work, name in NAMES.items():
    stats = glob.glob(os.path.join(SRC, work, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(DST, f"{TAG}_{name}_kernel_stats.csv"))
    fetch, _ = counters(work, "fetch")
    write, _ = counters(work, "write")
    for sub in ("fetch", "write", "sq1", "sq2"):
        vals, m = counters(work, sub)
        for k in sorted(vals):
            if m[k]["launches"] < 3:
                continue
            lines.append(f"{name:12s} {sub:5s} {k[0]:52s} grid {k[1]:9d} launches {m[k]['launches']:5d} regs "
                         f"v{m[k]['vgpr']} s{m[k]['sgpr']} lds {m[k]['lds']} scratch {m[k]['scratch']}  " +
                         " ".join(f"{c}={round(v)}" for c, v in sorted(vals[k].items())))
    kernels = {}
    sq1, sq1_meta = counters(work, "sq1")
    for k in sorted(set(fetch) | set(sq1)):
        entry = {"kernel": k[0], "grid": k[1]}
        if k in fetch and k in write:
            f_kib, w_kib = fetch[k]["FETCH_SIZE"], write[k]["WRITE_SIZE"]
            # MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE reports half the bytes of wide coalesced reads on gfx950
            # -> doubled (an upper bound for the narrower reads in these kernels); WRITE_SIZE is exact
            entry.update({"FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib,
                          "hbm_bytes_per_launch": int((2 * f_kib + w_kib) * 1024)})
        if k in sq1 and sq1_meta[k]["launches"] >= 3 and "SQ_INSTS_VALU" in sq1[k]:
            entry["valu"] = {c: sq1[k][c] for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES",
                                                    "SQ_WAIT_ANY") if c in sq1[k]}
        if len(entry) > 2:
            kernels[f"{k[0]} @grid {k[1]}"] = entry
    cfg = None
    out_file = os.path.join(SRC, f"{work}.out")
    if work in ("head", "c4") and os.path.exists(out_file):
        js = [l for l in open(out_file) if l.startswith("{")]
        if js:
            b = json.loads(js[-1])
            cfg = {k: b["config"][k] for k in ("num_points", "height", "width", "num_intersects_rank0")}
            if work == "head":
                open(os.path.join(DST, f"{TAG}_bench_under_rocprof.json"), "w").write(js[-1])
            else:
                open(os.path.join(DST, f"{TAG}_c4_2k_under_rocprof.json"), "w").write(js[-1])
    elif work in ("batched", "batched4", "batched8"):
        cfg = {"num_points": 50000, "height": 512, "width": 768, "images_per_launch": int(work[7:] or 24)}
    elif work == "frozen":
        cfg = {"num_points": 50000, "height": 512, "width": 768, "frozen": True}
    wl = {"workload": name, "config": cfg, "kernels": kernels}
    lane = os.path.join(SRC, "lane_model.json")
    if work == "head" and os.path.exists(lane):
        js = [l for l in open(lane) if l.startswith("{")]
        if js:
            wl["lane_model"] = json.loads(js[-1])
    if kernels:
        workloads.append(wl)

open(os.path.join(DST, f"{TAG}_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
plain = [l for l in open(os.path.join(SRC, "bench_plain.json")) if l.startswith("{")]
if plain:
    open(os.path.join(DST, f"{TAG}_bench_plain.json"), "w").write(plain[-1])
for src, dst in (("c4_plain.json", f"{TAG}_c4_2k_plain.json"), ("bench_driverlike.json", f"{TAG}_bench_steps20_plain.json")):
    if os.path.exists(os.path.join(SRC, src)):
        js = [l for l in open(os.path.join(SRC, src)) if l.startswith("{")]
        if js:
            open(os.path.join(DST, dst), "w").write(js[-1])
bp = os.path.join(SRC, "batched_plain.out")
if os.path.exists(bp):
    open(os.path.join(DST, f"{TAG}_batched_plain.txt"), "w").write(
        "".join(l for l in open(bp) if l.startswith("K=") or l.startswith("single")))
for src, dst in (("dropin_profile.txt", f"{TAG}_dropin_profile_after.txt"), ("kodak50k.out", f"{TAG}_kodak_fit_50k_run.txt"),
                 ("trained_fit.out", f"{TAG}_trained_scene_fit.txt"), ("kodak_shards.out", f"{TAG}_kodak_shards.txt")):
    if os.path.exists(os.path.join(SRC, src)):
        keep = [l for l in open(os.path.join(SRC, src)) if "amdgpu.ids" not in l]
        open(os.path.join(DST, dst), "w").write("".join(keep))
for work in ("c5", "fit"):
    f = os.path.join(SRC, f"{work}.out")
    if os.path.exists(f):
        keep = [l for l in open(f) if ("us/iter" in l or "images/s" in l or l.startswith("best"))]
        open(os.path.join(DST, f"{TAG}_{NAMES[work]}_run.txt"), "w").write("".join(keep))
json.dump({"source": f"profiles/{TAG}_pmc_summary.txt (tools/profile_rounds.sh, separate --pmc passes)",
           "workloads": workloads}, open(os.path.join(DST, "traffic.json"), "w"), indent=1)
print("\n".join(lines))
