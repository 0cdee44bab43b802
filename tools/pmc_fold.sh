#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass as the MI355X guide prescribes) of this library's kernels (joins, both folds) on the
# HBM-resident form of configs[2].  usage: tools/pmc_fold.sh OUTDIR -> OUTDIR/pmc_fold_summary.txt
out=$1; mkdir -p $out
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export TMPDIR=/tmp; cd "$REPO" || exit 1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $out/$ctr -- python3 bench.py --resident-only --resident-passes 1 > $out/$ctr.json 2> $out/$ctr.err
done
python3 - $out <<'PY' | tee $out/pmc_fold_summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
last = collections.defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + '/' + ctr + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'xck::' not in k or r['Counter_Name'] != ctr: continue
            name = k.split('(')[0].replace('void xck::', '').replace('xck::', '') + ' grid=' + r['Grid_Size']
            d = int(r['Dispatch_Id'])
            if ctr not in last[name] or d > last[name][ctr][0]: last[name][ctr] = (d, float(r['Counter_Value']))
print("last dispatch of each (kernel, grid): HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)")
for name in sorted(last, key=lambda n: -(2 * last[n].get('FETCH_SIZE', (0, 0))[1] + last[n].get('WRITE_SIZE', (0, 0))[1])):
    f, w = last[name].get('FETCH_SIZE', (0, 0))[1], last[name].get('WRITE_SIZE', (0, 0))[1]
    if (2 * f + w) * 1024 < 5e7: continue
    print("  %-52s read %7.2f GB  write %7.2f GB" % (name[:52], 2 * f * 1024 / 1e9, w * 1024 / 1e9))
PY
