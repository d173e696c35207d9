#!/bin/bash
# (GPU box) s_memtime timeline of k_long: builds the library with -DLW_STAMPS [+ $1], runs tools/stamps.py
D=gpurun_out/${TAG:-stamps}
mkdir -p $D
LW_EXTRA_FLAGS="-DLW_STAMPS $1" python lewton_amd/build.py --force > $D/build.log 2>&1
timeout 300 python tools/stamps.py 256 4096 > $D/stamps.txt 2>&1
tail -n 32 $D/stamps.txt
