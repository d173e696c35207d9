#!/bin/bash
# (GPU box) round 5, last session: the ring's users after the drain fix, several rings on one GPU inside one process and in two
# processes (lw_decoder_set_shared_device), and a longer differential campaign on the final tree
D=gpurun_out/r05_final4; mkdir -p $D
( time timeout 400 python -m pytest tests/test_gpu_ring.py tests/test_gpu_shapes.py tests/test_gpu_ogg.py -m gpu -q ) > $D/pytest.txt 2>&1; tail -4 $D/pytest.txt
{
for sd in "" "--shared-device"; do
  echo "one process, 2 callers (a ring each) $sd: $(timeout 60 python tools/e2e.py --batches 300 --device-entropy --callers 2 $sd 2>&1 | tail -1 | cut -c1-110)"
done
for sd in "" "--shared-device"; do
  T=$(python3 -c "import time; print(time.time() + 25)")
  timeout 100 python tools/e2e.py --batches 400 --device-entropy --start-at $T $sd > $D/p1.txt 2>&1 &
  P1=$!
  timeout 100 python tools/e2e.py --batches 400 --device-entropy --start-at $T $sd > $D/p2.txt 2>&1 &
  P2=$!
  wait $P1 $P2
  echo "two processes, a ring each $sd: $(tail -1 $D/p1.txt | cut -c1-90) | $(tail -1 $D/p2.txt | cut -c1-90)"
done
} 2>&1 | tee $D/tenants.txt
timeout 400 python tools/fuzz_gpu_mixed.py --rounds 150 --seed 101 2>&1 | tail -1 | tee $D/fuzz.txt
timeout 400 python tools/fuzz_gpu_mixed.py --rounds 150 --seed 102 --mid 2>&1 | tail -1 | tee -a $D/fuzz.txt
timeout 300 python tools/fuzz_gpu_mixed.py --rounds 80 --seed 103 --big 2>&1 | tail -1 | tee -a $D/fuzz.txt
