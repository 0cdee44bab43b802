#!/bin/bash
# Round profile on the GPU box (one gpurun call): final bench line, kernel trace of the HBM-resident sub-record (the launches
# the roofline is computed from), kernel trace of the end-to-end pass, HBM counters of the join kernels (one --pmc pass per counter).
# usage: tools/profile_round.sh OUT_DIR      (OUT_DIR under gpurun_out/)
set -o pipefail
OUT=${1:-gpurun_out/prof_round}
REPO=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
echo "== bench (default command)"; timeout -k 10 600 python3 bench.py --verbose > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || exit 1
tail -3 "$OUT/bench_default.err"
echo "== kernel trace: bench.py --resident-only"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_resident" -- python3 bench.py --resident-only --resident-passes 5 > "$OUT/bench_resident_profiled.json" 2> "$OUT/trace_resident.err" || exit 1
echo "== unprofiled: bench.py --resident-only"
timeout -k 10 300 python3 bench.py --resident-only --resident-passes 5 > "$OUT/bench_resident.json" 2>/dev/null || exit 1
echo "== kernel trace: end-to-end pass"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_e2e" -- python3 bench.py --cpu-sample 0 --resident-passes 0 > "$OUT/bench_e2e_profiled.json" 2> "$OUT/trace_e2e.err" || exit 1
for label in fc fc_filtered baf; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    echo "== pmc $label $ctr"
    timeout -k 10 400 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc/${label}_${ctr}" -- python3 tools/pmc_join.py $label > "$OUT/pmc_${label}_${ctr}.log" 2>&1 || exit 1
  done
done
python3 tools/pmc_traffic.py "$OUT/pmc" "$OUT/pmc_traffic.json" "see profiles/pmc_traffic.json" > /dev/null
find "$OUT" -name "*_kernel_stats.csv" | head
# the raw traces are large: keep the stats and counter csvs only
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
du -sh "$OUT"
