#!/usr/bin/env python3
"""Timeline of the LAST basefc fold in a rocprofv3 kernel trace (tools/join_time.py ... fc, or bench.py --resident-only): every launch
with its start offset, duration and the idle gap before it (no kernel of this process running), and the sums - where the stage's time
goes that no kernel accounts for (host syncs, launch gaps).  usage: fold_timeline.py TRACE_DIR [min_us_to_list]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
lim = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name'].replace('void ', '').replace('xck::', '').split('(')[0]
j1 = [i for i, r in enumerate(rows) if 'k_join<unsigned long, 1>' in r['Kernel_Name']][-1]
seg = rows[j1 + 1:]
stop = [i for i, r in enumerate(seg) if 'k_join' in r['Kernel_Name'] or 'k_tile_meta' in r['Kernel_Name']]
if stop: seg = seg[:stop[0]]
t0 = int(rows[j1]['End_Timestamp'])
busy_until, gaps, kern = t0, 0.0, 0.0
print("basefc fold after the last k_join<basefc>: %d launches" % len(seg))
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = max(0, s - busy_until) / 1e3
    gaps += gap; kern += (e - s) / 1e3
    if (e - s) / 1e3 >= lim or gap >= lim:
        print("  +%8.1f us  %-44s %8.1f us   idle before: %6.1f us" % ((s - t0) / 1e3, name(r)[:44], (e - s) / 1e3, gap))
    busy_until = max(busy_until, e)
print("first start to last end: %.1f us; sum of kernel durations %.1f us (overlapping streams count twice); idle (no kernel running) %.1f us" % ((busy_until - t0) / 1e3, kern, gaps))
