#!/bin/bash
# gfx950 disassembly of one device translation unit of lewton_amd/_lib (for tools/device_func_diff.py)
#   usage: tools/device_dis.sh lw_kernels_long out.dis [objdir]
D=${3:-lewton_amd/_lib}
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin $D/$1.hip.o $T/x.bin &&
$B/clang-offload-bundler --unbundle --type=o --input=$T/x.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/x.co &&
$B/llvm-objdump -d $T/x.co > $2
rm -rf $T
