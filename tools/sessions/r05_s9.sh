#!/bin/bash
# (GPU box) round 5, session 9: the whole GPU suite on the final build (after the final measurement set: interleaved-stereo stores of
# k_long10 / k_mix10 -- the only kernels whose code differs from the measured build, tools/device_func_diff.py -- and the single-mode route)
D=gpurun_out/r05_s9; mkdir -p $D
( time timeout 1800 python -m pytest tests -m gpu -q ) > $D/pytest.txt 2>&1
tail -12 $D/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.txt 2>&1; tail -2 $D/smoke.txt
