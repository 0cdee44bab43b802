#!/bin/bash
# run bench.py once per engine variant in tools/scratch (XCK_LIB selects the .so)
for so in tools/scratch/libxck_*.so; do
  XCK_LIB=$PWD/$so timeout -k 10 300 python bench.py --cpu-sample 0 --steps 3 --warmup 1 ${BENCH_ARGS} --serial 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-40s value=%.3g ms/step=%.2f %s hits=%s' % ('$so'.split('/')[-1], d['value'], d['ms_per_step'], r['stage_ms_per_step'], d['config']['hits']))"
done
