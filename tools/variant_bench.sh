#!/bin/bash
# time the join / fold of every engine variant in tools/scratch (XCK_LIB selects the .so) on the configs[2] workload shape
# usage: tools/variant_bench.sh [reads] [reps] [modes]        e.g.  tools/variant_bench.sh 500000000 3 fc,baf
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$REPO" || exit 1
for so in xcltk_amd/csrc/libxck.so tools/scratch/libxck_*.so; do
  [ -f "$so" ] || continue
  echo "== $so"
  XCK_LIB=$PWD/$so timeout -k 10 300 python3 tools/join_time.py ${1:-500000000} ${2:-3} ${3:-fc,baf} 2>&1 | grep -E "join|rror|stamps" | tail -4
done
