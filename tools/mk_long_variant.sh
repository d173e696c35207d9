#!/bin/bash
# (local) variant_<name>.so = the current library with lw_kernels_long.hip recompiled with extra flags:
#   tools/mk_long_variant.sh <name> "<flags>" [source]   (other objects are taken from lewton_amd/_lib as built;
#   [source]: another copy of lw_kernels_long.hip, e.g. a patched one outside the tree -- its includes still resolve to csrc/)
NAME=$1; FLAGS=$2; SRC=${3:-lewton_amd/csrc/lw_kernels_long.hip}
L=lewton_amd/_lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wall -Wno-unused-result -pthread \
  -fno-slp-vectorize -Ilewton_amd/csrc $FLAGS -c $SRC -o $L/long_$NAME.o || exit 1
OBJS=$(ls $L/*.cpp.o $L/lw_kernels.hip.o $L/lw_kernels_entropy.hip.o $L/lw_kernels_big.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/variant_$NAME.so $OBJS $L/long_$NAME.o -pthread && echo built variant_$NAME "($FLAGS)"
