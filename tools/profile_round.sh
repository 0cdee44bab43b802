#!/bin/bash
# Round profile on the GPU box (one gpurun call): kernel trace of the HBM-resident sub-record (the launches the roofline is computed
# from) with the per-fold kernel table, kernel trace of the end-to-end pass, HBM counters of the join kernels and of the fold
# kernels (one --pmc pass per counter), SQ counters, phase stamps of k_pf_part<1>.
# usage: tools/profile_round.sh OUT_DIR      (OUT_DIR under gpurun_out/; bench.py must have generated /tmp/xck_bench before, or this does)
set -o pipefail
OUT=${1:-gpurun_out/prof_round}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$REPO
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO" || exit 1
echo "== kernel trace: bench.py --resident-only"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_resident" -- python3 bench.py --resident-only --resident-passes 5 > "$OUT/bench_resident_profiled.json" 2> "$OUT/trace_resident.err" || exit 1
python3 tools/fold_times.py "$OUT/trace_resident" > "$OUT/fold_kernel_times.txt" 2>&1; cat "$OUT/fold_kernel_times.txt"
echo "== unprofiled: bench.py --resident-only"
timeout -k 10 300 python3 bench.py --resident-only --resident-passes 5 > "$OUT/bench_resident.json" 2>/dev/null || exit 1
echo "== kernel trace: end-to-end pass"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_e2e" -- python3 bench.py --make-room --cpu-sample 0 --resident-passes 0 --sub-reads 0 > "$OUT/bench_e2e_profiled.json" 2> "$OUT/trace_e2e.err" || exit 1
for label in fc fc_filtered baf; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    echo "== pmc $label $ctr"
    timeout -k 10 400 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc/${label}_${ctr}" -- python3 tools/pmc_join.py $label > "$OUT/pmc_${label}_${ctr}.log" 2>&1 || exit 1
  done
done
python3 tools/pmc_traffic.py "$OUT/pmc" "$OUT/pmc_traffic.json" "see profiles/pmc_traffic.json" > /dev/null
echo "== pmc: fold kernels"; tools/pmc_fold.sh "$OUT/pmc_fold" > "$OUT/pmc_fold.log" 2>&1 || echo "pmc_fold.sh FAILED (see $OUT/pmc_fold.log)"; cat "$OUT/pmc_fold/pmc_fold_summary.txt" | head -30
echo "== SQ counters"; tools/sq_profile.sh "$OUT/sq" > "$OUT/sq.log" 2>&1 || echo "sq_profile.sh FAILED (see $OUT/sq.log)"; grep -A2 -E "^k_join|^k_pf_(bucket|hist<1>|part<1>)" "$OUT/sq/sq_summary.txt" | head -40
if [ -f tools/scratch/libxck_STAMPS.so ]; then echo "== stamps"; XCK_LIB=$REPO/tools/scratch/libxck_STAMPS.so python3 bench.py --resident-only --resident-passes 1 2>&1 | grep stamps | tail -1 | tee "$OUT/part1_stamps.txt"; fi
find "$OUT" -name "*_kernel_stats.csv" | head
# the raw traces are large: keep the stats and counter csvs only
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
du -sh "$OUT"
