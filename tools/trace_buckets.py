"""Kernel-trace post-processing (development aid): average duration of the tile pass and the update kernel per bucket of
consecutive launches -- how the cost of an adaptive fit moves as the population grows.
usage: trace_buckets.py <rocprofv3 output dir> [launches per bucket]"""
import csv
import glob
import sys

d, per = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5000
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "fast_fwdbwd" in n or "reduce_update" in n:
            rows.append((int(r["Start_Timestamp"]), "tile" if "fast_fwdbwd" in n else "update",
                         int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
for kind in ("tile", "update"):
    dur = [x[2] for x in rows if x[1] == kind]
    print(kind, " ".join(f"{sum(dur[i:i + per]) / max(len(dur[i:i + per]), 1) / 1e3:.0f}" for i in range(0, len(dur), per)),
          "us per launch, buckets of", per)
