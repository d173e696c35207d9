#!/bin/bash
# (GPU box) config 3-5 kernel breakdown for library variants: tools/cfg_exp.sh "<flags>" ...
for F in "$@"; do
  LW_EXTRA_FLAGS="$F" python lewton_amd/build.py --force > /dev/null 2>&1
  echo "== flags [$F]"
  timeout 100 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "read_audio_packet or batch_many or window_mismatch" 2>&1 | tail -1
  timeout 150 tools/prof_cfg.sh | grep -v copyBuffer
  head -1 gpurun_out/cfg/prof/out.txt | cut -c1-130
done
