#!/bin/bash
# usage (on the GPU box): tools/gpu_quick.sh <tag> [notest]  -- parity tests + bench with the shipped build, then the stamp profile
TAG=$1
D=gpurun_out/$TAG
mkdir -p $D
if [ "$2" != "notest" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $D/pytest.log 2>&1; echo "pytest rc=$?" >> $D/pytest.log
fi
timeout 300 python bench.py --no-cpu-baseline > $D/bench.json 2> $D/bench.err
LW_EXTRA_FLAGS=-DLW_STAMPS python lewton_amd/build.py --force > $D/build_stamps.log 2>&1
timeout 300 python tools/stamps.py 256 4096 > $D/stamps.txt 2>&1
