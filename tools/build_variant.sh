#!/bin/bash
# usage: tools/build_variant.sh NAME "-DFLAG ..."  -> columbiaimagesearch_amd/lib/libcis_NAME.so
# A build of the library whose lopq_search.hip is compiled with extra flags (A/B experiments, tools/gpu_ab.sh).
set -e
cd "$(dirname "$0")/../columbiaimagesearch_amd/csrc"
NAME=$1; FLAGS=$2
make -s build/lopq_model.o build/cnn.o build/lopq_sort.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-result \
  -I../../include $FLAGS -c lopq_search.hip -o build/variant_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -no-hip-rt build/lopq_model.o build/cnn.o build/lopq_sort.o build/variant_$NAME.o -o ../lib/libcis_$NAME.so
echo built libcis_$NAME.so
