#!/usr/bin/env python3
"""inflate_timeline.py TRACE_DIR - what the device did during an end-to-end pass with the GPU share of the inflate on: per kernel
family (k_inflate, k_inflate_copy_out, joins, the rest) launches, mean / max duration, summed duration, union of the busy intervals,
and the mean number of k_inflate launches running at once.  Input: rocprofv3 --kernel-trace --output-format csv (kernel_trace.csv)."""
import csv, glob, sys
fn = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(fn) as f:
    for r in csv.DictReader(f):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
def fam(n):
    if "k_inflate_copy_out" in n: return "k_inflate_copy_out"
    if "k_inflate" in n: return "k_inflate"
    if "k_join" in n: return "k_join"
    if "k_tile_meta" in n: return "k_tile_meta"
    return "other"
t0 = min(r[1] for r in rows); t1 = max(r[2] for r in rows)
print("trace: %d launches over %.1f ms" % (len(rows), (t1 - t0) / 1e6))
for f_ in ("k_inflate", "k_inflate_copy_out", "k_join", "k_tile_meta", "other"):
    rs = sorted((a, b) for n, a, b in rows if fam(n) == f_)
    if not rs: continue
    tot = sum(b - a for a, b in rs); mx = max(b - a for a, b in rs)
    union = 0; cs, ce = rs[0]
    for a, b in rs[1:]:
        if a > ce: union += ce - cs; cs, ce = a, b
        else: ce = max(ce, b)
    union += ce - cs
    span = rs[-1][1] - rs[0][0]
    print("%-20s %6d launches  mean %8.3f ms  max %8.3f ms  sum %9.1f ms  busy (union) %9.1f ms of a span of %9.1f ms  mean concurrency while busy %.2f"
          % (f_, len(rs), tot / len(rs) / 1e6, mx / 1e6, tot / 1e6, union / 1e6, span / 1e6, tot / max(union, 1)))
