#!/bin/bash
# csrc/inflate_dev.hip on three input shapes, every block compared with zlib (variant 10 = with phase clocks; XCK_AB_FLAGS: extra hipcc flags).   usage: tools/inflate_ab.sh OUTDIR [records]
out=$1; n=${2:-8000000}; mkdir -p $out
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$REPO" || exit 1
W=/tmp/xck_hybrid; mkdir -p $W
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $XCK_AB_FLAGS -Ixcltk_amd/csrc tools/gpu_inflate_bench.hip xcltk_amd/csrc/inflate_dev.hip -lz -o $W/gpu_inflate_bench 2> $out/build.log || { echo "build failed"; tail -5 $out/build.log; exit 1; }
python3 tools/ingest_scaling.py $W/fast.bam $W/barcodes.tsv --gen $n --level 0 --threads 24 --snps 1000 > $out/host_fast.log 2>&1
python3 tools/ingest_scaling.py $W/zlib6.bam $W/barcodes.tsv --gen $n --level 6 --threads 24 --snps 1000 > $out/host_zlib6.log 2>&1
XCK_SYNTH_SHAPE=cellranger python3 tools/ingest_scaling.py $W/cr.bam $W/barcodes.tsv --gen $n --level 6 --threads 24 --snps 1000 > $out/host_cr.log 2>&1
for f in fast zlib6 cr; do
  echo "== $f: $(ls -l $W/$f.bam | awk '{print $5}') bytes"
  for v in ${VARIANTS:-0 10}; do for per in 0 740; do
    if [ $per = 0 ]; then a=""; else a="$per"; fi
    echo "-- variant $v, blocks per launch: ${per/#0/all}"; INFLATE_VARIANT=$v timeout -k 10 120 $W/gpu_inflate_bench $W/$f.bam 3000000000 $a 2>&1 | grep -E "blocks,|rep 2|verified|differs|wave cycles" | tail -4
  done; done
done | tee $out/summary.txt
