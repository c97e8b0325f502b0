"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
        for k in acc:
            if "gi2d" in k:
                print(d, k, "vgpr/agpr/sgpr/lds/scratch", meta[k], {c: round(sum(v) / len(v)) for c, v in acc[k].items()})
