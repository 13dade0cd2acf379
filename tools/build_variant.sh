#!/bin/bash
# usage: tools/build_variant.sh NAME "-DFLAG ..." [source stem, default lopq_search]  -> columbiaimagesearch_amd/lib/libcis_NAME.so
# A build of the library with ONE source file compiled with extra flags (A/B experiments, tools/gpu_ab.sh).
set -e
cd "$(dirname "$0")/../columbiaimagesearch_amd/csrc"
NAME=$1; FLAGS=$2; SRC=${3:-lopq_search}
make -s all
OBJS=$(ls build/*.o | grep -v "build/variant_" | grep -v "build/$SRC.o")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-result \
  -I../../include $FLAGS -c $SRC.hip -o build/variant_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -no-hip-rt $OBJS build/variant_$NAME.o -o ../lib/libcis_$NAME.so
echo built libcis_$NAME.so
