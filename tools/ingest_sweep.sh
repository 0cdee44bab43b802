#!/bin/bash
# End-to-end A/B of decoder settings on the GPU box: every argument is one configuration - a quoted list of VAR=value pairs (or "base") -
# and runs `bench.py` at a reduced size under it (headline file + the two side files, oracle check included); one line per run.
#   usage: tools/ingest_sweep.sh OUTDIR READS "XCK_GPU_INFLATE=0" "XCK_GPU_INFLATE_DEPTH=6 XCK_GPU_INFLATE_FREE_CUS=16" base ...
# (the sweeps behind profiles/experiments/gpu_inflate/README.md were made this way)
out=$1; reads=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$REPO" || exit 1
mkdir -p "$out"
i=0
for cfg in "$@"; do
  i=$((i + 1)); tag=$(printf "%02d" $i)
  if [ "$cfg" = base ]; then envs=""; else envs="$cfg"; fi
  env $envs XCK_DEBUG_TIMING=1 timeout -k 10 500 python3 bench.py --make-room --reads "$reads" --cpu-sample 4000000 --resident-passes 0 --sub-reads 20000000 --well-sub-reads 0 > "$out/bench_$tag.json" 2> "$out/bench_$tag.err"
  python3 - "$out/bench_$tag.json" "$cfg" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-60s headline %.2f M reads/s | zlib6 %s | cellranger %s | device chunks %s | oracle %s" % (sys.argv[2], d["value"] / 1e6, [round(v / 1e6, 1) for v in d["end_to_end_zlib6"]["values_all_passes"]],
          [round(v / 1e6, 1) for v in d["cellranger_shape"]["values_all_passes"]], d["end_to_end"].get("gpu_inflate", {}).get("chunks_on_device"), d["cpu_baseline"]["gpu_rows_vs_oracle"][:2]))
except Exception as e:
    print("%-60s FAILED: %s" % (sys.argv[2], e))
PY
  grep -E "coordinator" "$out/bench_$tag.err" | grep -v "well/" | head -1 | cut -c60-520
done | tee "$out/summary.txt"
