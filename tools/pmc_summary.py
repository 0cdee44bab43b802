#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, sum of each counter over dispatches."""
import csv, glob, sys, collections
pat = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_*/*/*_counter_collection.csv"
filt = sys.argv[2] if len(sys.argv) > 2 else "xck::"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for fn in sorted(glob.glob(pat)):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if filt not in k: continue
        k = k.split("(")[0].replace("void ", "")[:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in sorted(agg):
    print(k)
    for c, v in sorted(agg[k].items()):
        print("   %-24s %18.0f  (%d dispatches)" % (c, v, len(calls[(k, c)])))
