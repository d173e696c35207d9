#!/bin/bash
# (GPU box) round 3: parity with k_short / k_long<EDGE>, timing of the mixed configuration with a kernel breakdown
D=gpurun_out/r03b
mkdir -p $D
timeout 1200 python -m pytest tests -m gpu -x -q > $D/pytest.txt 2>&1
tail -25 $D/pytest.txt
timeout 300 python tools/bench_configs.py --only 3,4 > $D/other_configs.jsonl 2> $D/other_configs.err
cut -c1-330 $D/other_configs.jsonl; tail -3 $D/other_configs.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$D/prof -o stats -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --only 3 --steps 80 --no-verify > $GRAFT_REPO_ROOT/$D/prof_out.txt 2> $GRAFT_REPO_ROOT/$D/rocprof.log
cd $GRAFT_REPO_ROOT
find $D/prof -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200 | head -12
