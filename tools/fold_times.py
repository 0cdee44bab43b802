#!/usr/bin/env python3
"""Kernel times of the last basefc fold and the last pileup fold in a rocprofv3 kernel trace of `bench.py --resident-only`.
usage: fold_times.py TRACE_DIR"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
j1 = [i for i, r in enumerate(rows) if 'k_join<unsigned long, 1>' in r['Kernel_Name']][-1]
j2 = [i for i, r in enumerate(rows) if 'k_join<unsigned long, 2>' in r['Kernel_Name']][-1]
for name, a, b in (("basefc fold", j1 + 1, j2 if j2 > j1 else len(rows)), ("pileup fold", j2 + 1, len(rows) if j2 > j1 else j1)):
    seg = [r for r in rows[a:b] if 'k_tile_meta' not in r['Kernel_Name'] and 'k_join' not in r['Kernel_Name']]
    tot = sum(dur(r) for r in seg)
    print("== %s: %d launches, %.1f us of kernels, %.1f us from first start to last end" % (name, len(seg), tot, (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3 if seg else 0))
    for r in seg:
        if dur(r) > 60: print("   %-58s %8.1f us" % (r['Kernel_Name'].replace('void ', '').replace('xck::', '')[:58], dur(r)))
