#!/bin/bash
# VERDICT r03 item 8, scoped: what a GPU share of the BGZF inflate could add to the host decoder, per input shape.
# Generates three 8 M-record synthetic BAMs (the headline's fast-compressor blocks, zlib-6 blocks, Cell Ranger record shape + zlib 6),
# inflates every block on the GPU with the archived one-wave-per-block decoder (profiles/experiments/gpu_inflate: all blocks in one
# launch / 3000 per launch / 740 per launch = one decode chunk; every block compared with zlib) and times the host decoder alone
# (decode-only handle: inflate + record walk + parse) on the same files.   usage: tools/hybrid_probe.sh OUTDIR
out=$1; mkdir -p $out
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$REPO" || exit 1
W=/tmp/xck_hybrid; mkdir -p $W
E=profiles/experiments/gpu_inflate
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$E $E/gpu_inflate_bench.hip $E/inflate_dev.hip -lz -o $W/gpu_inflate_bench 2> $out/build.log || { echo "build failed"; exit 1; }
python3 tools/ingest_scaling.py $W/fast.bam $W/barcodes.tsv --gen 8000000 --level 0 --threads 24 --snps 1000 > $out/host_fast.log 2>&1
python3 tools/ingest_scaling.py $W/zlib6.bam $W/barcodes.tsv --gen 8000000 --level 6 --threads 24 --snps 1000 > $out/host_zlib6.log 2>&1
XCK_SYNTH_SHAPE=cellranger python3 tools/ingest_scaling.py $W/cr.bam $W/barcodes.tsv --gen 8000000 --level 6 --threads 24 --snps 1000 > $out/host_cr.log 2>&1
for f in fast zlib6 cr; do
  echo "== $f: $(ls -l $W/$f.bam | awk '{print $5}') bytes; host decoder (24 threads behind the box's CPU quota): $(grep decode_only $out/host_$f.log | tail -1)"
  for per in 0 3000 740; do
    if [ $per = 0 ]; then a=""; else a="$per"; fi
    echo "-- gpu inflate, blocks per launch: ${per/#0/all}"; timeout -k 10 120 $W/gpu_inflate_bench $W/$f.bam 3000000000 $a 2>&1 | grep -E "blocks,|rep 2|verified"
  done
done | tee $out/summary.txt
