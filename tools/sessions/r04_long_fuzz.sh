#!/bin/bash
# (GPU box) longer differential campaigns on the round's final device code -> gpurun_out/r04_long_fuzz/
D=gpurun_out/r04_long_fuzz; mkdir -p $D
{ for S in surround51_bookless multichannel12 stereo surround51; do timeout 600 python tools/fuzz_gpu_entropy.py --packets 200000 --seed 5 --setup $S 2>&1 | tail -1 | cut -c1-200; done; } > $D/fuzz_gpu_entropy_long.txt 2>&1
timeout 900 python tools/fuzz_gpu_mixed.py --rounds 300 --seed 43 2>&1 | tail -1 > $D/fuzz_gpu_mixed_long.txt
cat $D/fuzz_gpu_entropy_long.txt $D/fuzz_gpu_mixed_long.txt
