#!/bin/bash
# Identity of the device code in lewton_amd/_lib/*.hip.o: sha256 of the gfx950 disassembly of each translation unit
# (the code objects themselves also embed source paths).  Host-side changes must leave these unchanged; compare with
# profiles/rNN_device_code.sha256, the build the round's GPU tests, bench line and rocprof summaries were taken on.
#   usage: tools/device_code_id.sh [objdir]
D=${1:-lewton_amd/_lib}
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
for f in lw_kernels lw_kernels_long lw_kernels_big lw_kernels_entropy; do
  objcopy -O binary --only-section=.hip_fatbin $D/$f.hip.o $T/$f.bin &&
  $B/clang-offload-bundler --unbundle --type=o --input=$T/$f.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/$f.co &&
  echo "$($B/llvm-objdump -d $T/$f.co | tail -n +3 | sha256sum | cut -d' ' -f1)  $f.hip (gfx950 disassembly)"
done
rm -rf $T
