"""GPU busy time of a rocprofv3 --kernel-trace csv as the UNION of the dispatches' intervals (several streams overlap),
the idle time between them, and what the longest idle gaps sit between (development aid).  usage: busy_union.py <dir>"""
import collections
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[-48:]))
rows.sort()
t0, t1 = rows[0][0], max(e for _, e, _ in rows)
busy, cur_s, cur_e, cur_name = 0, rows[0][0], rows[0][1], rows[0][2]
gaps = []
for s, e, nm in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_name, nm, cur_e - t0))
        cur_s, cur_e, cur_name = s, e, nm
    elif e > cur_e:
        cur_e, cur_name = e, nm
busy += cur_e - cur_s
span = t1 - t0
print(f"dispatches {len(rows)}, span {span / 1e9:.3f} s, busy (union) {busy / 1e9:.3f} s = {100 * busy / span:.1f} %, "
      f"sum of durations {sum(e - s for s, e, _ in rows) / 1e9:.3f} s")
g = sorted(x[0] for x in gaps)
if g:
    tot = sum(g)
    print(f"idle gaps: {len(g)}, total {tot / 1e9:.3f} s; median {g[len(g) // 2] / 1e3:.1f} us, p90 {g[int(len(g) * .9)] / 1e3:.1f} us, "
          f"p99 {g[int(len(g) * .99)] / 1e3:.1f} us, max {g[-1] / 1e3:.1f} us")
    for lo, hi in ((0, 5e3), (5e3, 20e3), (20e3, 100e3), (100e3, 1e6), (1e6, 1e12)):
        sel = [x for x in g if lo <= x < hi]
        print(f"  gaps of {lo / 1e3:7.0f} .. {hi / 1e3:9.0f} us: {len(sel):7d}, {sum(sel) / 1e9:.3f} s")
    by = collections.Counter()
    for d, a, b, _ in gaps:
        by[(a, b)] += d
    print("idle time by (kernel before -> kernel after):")
    for (a, b), d in by.most_common(12):
        print(f"  {d / 1e9:.3f} s  {a}  ->  {b}")
tot = collections.Counter()
cnt = collections.Counter()
for s, e, nm in rows:
    tot[nm] += e - s
    cnt[nm] += 1
print("kernel time by name (sum of durations):")
for nm, d in tot.most_common(8):
    print(f"  {d / 1e9:.3f} s  {cnt[nm]:7d} x {d / cnt[nm] / 1e3:8.1f} us  {nm}")
