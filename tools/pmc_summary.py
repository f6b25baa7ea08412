#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV: mean counter value per (kernel, workgroup size, mode arg)."""
import csv, glob, collections, sys
root = sys.argv[1]
files = glob.glob(root + "/**/*counter_collection.csv", recursive=True)
if not files:
    print("no counter_collection.csv under", root); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(files[0])):
    k = r["Kernel_Name"][:48] + "|wg" + r.get("Workgroup_Size", "?") + "|grid" + r.get("Grid_Size", "?")
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in agg:
    print(k)
    for c, v in agg[k].items():
        print("    %-22s n=%d mean=%.4g min=%.4g max=%.4g" % (c, len(v), sum(v) / len(v), min(v), max(v)))
