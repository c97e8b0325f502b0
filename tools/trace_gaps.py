"""Busy time and idle gaps of a rocprofv3 --kernel-trace csv, in windows of 1000 dispatches (development aid)."""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
rows.sort()
W = 1000
for i in range(0, len(rows) - W, W):
    win = rows[i:i + W]
    span = (win[-1][1] - win[0][0]) / 1e3
    busy = sum(e - s for s, e, _ in win) / 1e3
    gaps = sorted((win[j + 1][0] - win[j][1]) / 1e3 for j in range(W - 1))
    print(f"dispatch {i:7d}: span {span:10.1f} us busy {busy:10.1f} us ({100 * busy / span:5.1f} %)  gap med {gaps[W // 2]:7.2f} "
          f"p90 {gaps[int(W * 0.9)]:7.2f} max {gaps[-1]:9.2f}  first kernel {win[0][2]}")
