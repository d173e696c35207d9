#!/bin/bash
# (GPU box) k_short with split channel pairs: parity of the mixed tests, timing, kernel breakdown, PMC traffic
D=gpurun_out/r03c
mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_quoted_shapes.py tests/test_gpu_parity.py tests/test_gpu_ring.py tests/test_gpu_ogg.py -m gpu -x -q > $D/pytest.txt 2>&1
tail -5 $D/pytest.txt
for i in 1 2; do timeout 300 python tools/bench_configs.py --only 3 --steps 800 >> $D/other_configs.jsonl 2>> $D/other_configs.err; done
cut -c1-200 $D/other_configs.jsonl; tail -3 $D/other_configs.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$D/prof -o stats -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --only 3 --steps 400 --no-verify > $GRAFT_REPO_ROOT/$D/prof_out.txt 2> $GRAFT_REPO_ROOT/$D/rocprof.log
cd $GRAFT_REPO_ROOT
find $D/prof -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-160 | head -6
timeout 600 bash tools/pmc_mixed.sh r03c > $D/pmc_stdout.txt 2>&1
tail -60 $D/pmc_stdout.txt
