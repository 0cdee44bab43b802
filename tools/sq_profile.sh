#!/bin/bash
# SQ counters of the hand-written kernels on the HBM-resident form of configs[2] (bench.py --resident-only, one timed pass per counter
# set; 8 SQ counters per rocprofv3 pass).  usage: tools/sq_profile.sh OUTDIR   -> OUTDIR/sq_summary.txt (+ the raw CSVs)
out=$1; mkdir -p $out
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export TMPDIR=/tmp; cd "$REPO" || exit 1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/set$i -- python3 bench.py --resident-only --resident-passes 1 > $out/set$i.json 2> $out/set$i.err
done
python3 - $out <<'PY' | tee $out/sq_summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + '/set*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if not any(x in k for x in ('k_join', 'k_pf_', 'k_tile_meta', 'k_first', 'k_claim', 'k_tally', 'k_expand', 'k_hap')): continue
        name = k.split('(')[0].replace('void xck::', '').replace('xck::', '')
        acc[name][r['Counter_Name']] += float(r['Counter_Value'])
cols = ['SQ_WAVES', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_SMEM', 'SQ_INSTS_LDS',
        'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS', 'SQ_LDS_BANK_CONFLICT']
print("sums over all dispatches of the run (3 passes per engine: 2 sizing + 1 timed); *_CYCLES / WAIT / ACTIVE in quad-cycles")
for n in sorted(acc):
    a = acc[n]
    wc = a.get('SQ_WAVE_CYCLES', 0) or 1
    w = max(a.get('SQ_WAVES', 1), 1)
    print("%s" % n)
    print("   " + "  ".join("%s=%.4g" % (c.replace('SQ_', ''), a[c]) for c in cols if c in a))
    print("   share of wave cycles: wait_any %.2f  wait_inst_any %.2f  active_inst_any %.2f | per wave: valu %.0f salu %.0f lds %.0f vmem_rd %.0f vmem_wr %.0f" % (
        a.get('SQ_WAIT_ANY', 0) / wc, a.get('SQ_WAIT_INST_ANY', 0) / wc, a.get('SQ_ACTIVE_INST_ANY', 0) / wc,
        a.get('SQ_INSTS_VALU', 0) / w, a.get('SQ_INSTS_SALU', 0) / w, a.get('SQ_INSTS_LDS', 0) / w, a.get('SQ_INSTS_VMEM_RD', 0) / w, a.get('SQ_INSTS_VMEM_WR', 0) / w))
PY
