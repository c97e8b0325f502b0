"""Average duration per (kernel, grid size) from a rocprofv3 --kernel-trace csv (development aid).
usage: trace_by_grid.py <dir with *kernel_trace.csv>"""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")[-60:]
        acc[(name, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]))].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{name:62s} grid {grid:9d} calls {len(v):6d} avg {sum(v) / len(v):9.2f} us  med {v2[len(v2) // 2]:9.2f}  "
          f"min {v2[0]:8.2f} max {v2[-1]:9.2f}")
