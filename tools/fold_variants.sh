#!/bin/bash
# Kernel times of the basefc partition fold under the knobs of csrc/fold_partition.h (one rocprofv3 kernel trace per variant of
# `bench.py --resident-only`).  usage: tools/fold_variants.sh OUTDIR "VAR=VAL ..." "VAR=VAL ..." ...
out=$1; shift
mkdir -p $out
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export TMPDIR=/tmp; cd "$REPO" || exit 1
i=0
for v in "$@"; do
  i=$((i+1))
  ( export $v; rocprofv3 --kernel-trace --stats --output-format csv -d $out/v$i -- python3 bench.py --resident-only --resident-passes 3 > $out/v$i.json 2> $out/v$i.err )
  echo "== variant $i: $v" | tee -a $out/summary.txt
  python3 - $out/v$i <<'PY' | tee -a $out/summary.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
sel = [r for r in rows if any(k in r['Kernel_Name'] for k in ('k_pf_', 'k_scan', 'fillBuffer'))]
idx = [i for i, r in enumerate(sel) if 'k_pf_rowhist' in r['Kernel_Name']][-1]
tot = 0
for r in sel[idx - 2:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    if d > 100: print("  %-44s %9.1f us" % (r['Kernel_Name'][:44], d))
print("  sum of fold kernels %.1f us" % tot)
PY
done
