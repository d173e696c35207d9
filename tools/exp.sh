#!/bin/bash
# usage: tools/exp.sh <tag> "<extra hipcc flags>" [bench args]  -- builds a variant on the GPU box and benches it
TAG=$1; FLAGS=$2; shift; shift
mkdir -p gpurun_out/exp
LW_EXTRA_FLAGS="$FLAGS" python lewton_amd/build.py --force > gpurun_out/exp/$TAG.build.log 2>&1
python bench.py --no-cpu-baseline --steps 2000 --warmup 200 "$@" > gpurun_out/exp/$TAG.json 2> gpurun_out/exp/$TAG.err
