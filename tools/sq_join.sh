#!/bin/bash
# SQ counters of the join kernels only (HBM-resident configs[2] shape through tools/join_time.py, one pass per counter set).
# usage: tools/sq_join.sh OUTDIR [modes: fc,baf]   -> OUTDIR/sq_join_summary.txt
out=$1; modes=${2:-fc}; mkdir -p $out
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export TMPDIR=/tmp; cd "$REPO" || exit 1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/set$i -- python3 tools/join_time.py 500000000 1 $modes > $out/set$i.log 2> $out/set$i.err || exit 1
done
python3 - $out <<'PY' | tee $out/sq_join_summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + '/set*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_join' not in k: continue
        name = k.split('(')[0].replace('void xck::', '').replace('xck::', '')
        acc[name][r['Counter_Name']] += float(r['Counter_Value'])
for n in sorted(acc):
    a = acc[n]; wc = a.get('SQ_WAVE_CYCLES', 0) or 1; w = max(a.get('SQ_WAVES', 1), 1)
    print(n); print("   " + "  ".join("%s=%.4g" % (c.replace('SQ_', ''), v) for c, v in sorted(a.items())))
    print("   share of wave cycles: wait_any %.2f  wait_inst_any %.2f  active_inst_any %.2f | per wave: valu %.0f salu %.0f smem %.0f lds %.0f vmem_rd %.0f vmem_wr %.0f | wave life %.0f cycles" % (
        a.get('SQ_WAIT_ANY', 0) / wc, a.get('SQ_WAIT_INST_ANY', 0) / wc, a.get('SQ_ACTIVE_INST_ANY', 0) / wc, a.get('SQ_INSTS_VALU', 0) / w, a.get('SQ_INSTS_SALU', 0) / w,
        a.get('SQ_INSTS_SMEM', 0) / w, a.get('SQ_INSTS_LDS', 0) / w, a.get('SQ_INSTS_VMEM_RD', 0) / w, a.get('SQ_INSTS_VMEM_WR', 0) / w, 4 * wc / w))
PY
find $out -name "*agent_info.csv" -delete
