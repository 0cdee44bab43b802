#!/bin/bash
# instruction counts (one SQ counter pass) and launch time of k_join<basefc> for the shipped library and every tools/scratch/libxck_*.so
# usage: tools/ablate_join.sh OUTDIR
out=$1; mkdir -p $out
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export TMPDIR=/tmp; cd "$REPO" || exit 1
for so in xcltk_amd/csrc/libxck.so tools/scratch/libxck_*.so; do
  [ -f "$so" ] || continue
  n=$(basename $so .so)
  XCK_LIB=$PWD/$so rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM --output-format csv -d $out/$n -- python3 tools/join_time.py 500000000 1 fc > $out/$n.log 2> $out/$n.err
  XCK_LIB=$PWD/$so python3 tools/join_time.py 500000000 3 fc 2>/dev/null | grep join > $out/$n.time
  python3 - $out/$n $n <<'PY'
import csv, glob, sys, collections
a = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_join' in r['Kernel_Name']: a[r['Counter_Name']] += float(r['Counter_Value'])
w = max(a.get('SQ_WAVES', 1), 1)
print("%-14s per wave: valu %.0f salu %.0f smem %.0f lds %.0f | wave life %.0f cycles | %s" % (sys.argv[2], a['SQ_INSTS_VALU'] / w, a['SQ_INSTS_SALU'] / w, a['SQ_INSTS_SMEM'] / w, a['SQ_INSTS_LDS'] / w,
      4 * a['SQ_WAVE_CYCLES'] / w, open(sys.argv[1] + '.time').read().strip()))
PY
  find $out/$n -name "*agent_info.csv" -delete
done | tee $out/summary.txt
