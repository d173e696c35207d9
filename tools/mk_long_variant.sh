#!/bin/bash
# (local) variant_<name>.so = the current library with lw_kernels_long.hip recompiled with extra flags:
#   tools/mk_long_variant.sh <name> "<flags>"      (other objects are taken from lewton_amd/_lib as built)
NAME=$1; FLAGS=$2
L=lewton_amd/_lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wall -Wno-unused-result -pthread \
  -fno-slp-vectorize $FLAGS -c lewton_amd/csrc/lw_kernels_long.hip -o $L/long_$NAME.o || exit 1
OBJS=$(ls $L/*.cpp.o $L/lw_kernels.hip.o $L/lw_kernels_entropy.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/variant_$NAME.so $OBJS $L/long_$NAME.o -pthread && echo built variant_$NAME "($FLAGS)"
