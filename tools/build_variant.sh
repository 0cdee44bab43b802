#!/bin/bash
# build an engine variant next to the shipped library: tools/build_variant.sh NAME -DFLAG=... [-DFLAG=...]
#   -> tools/scratch/libxck_NAME.so   (select it with XCK_LIB=...; tools/variant_bench.sh times every variant found there)
set -e
name=$1; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
src=$REPO/xcltk_amd/csrc; out=$REPO/tools/scratch; mkdir -p $out/obj_$name
make -C $src -j4 bam.o api.o snptext.o inflate_dev.o > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-unused-value -I$REPO/include "$@" -c $src/engine.hip -o $out/obj_$name/engine.o 2> $out/obj_$name/build.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libxck_$name.so $out/obj_$name/engine.o $src/bam.o $src/api.o $src/snptext.o $src/inflate_dev.o -lz -lpthread
rm -rf $out/obj_$name
echo "built $out/libxck_$name.so ($*)"
