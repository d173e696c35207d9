#!/bin/bash
# (GPU box) the round's measurement set on the current build: GPU parity suite, driver-style bench line (timed), rocprofv3 kernel
# trace + stats and PMC passes of the bench command (headline kernel), every other configuration with its oracle check (incl. the
# round's new ones: a stream shape behind k_prep, ONE stream x 4096 packets, the forced generic fallback, 16 384-packet mixed
# launches), rocprofv3 stats and PMC passes of k_prep + k_long and of the single-stream shapes (halo pre-pass), the differential
# campaigns (random SETUPS first), the single-stream container path (examples/perf), the end-to-end rates.
# Everything lands in gpurun_out/r06_final/; tools/sessions/r06_collect.sh copies the summaries into profiles/.
D=gpurun_out/r06_final
mkdir -p $D
bash tools/host_limits.sh > $D/gpu_box_host.txt 2>&1
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $D/pytest.txt 2>&1
tail -6 $D/pytest.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $D/bench.json 2> $D/bench.err
tail -4 $D/bench.err
timeout 400 bash tools/prof.sh r06_final --steps 20 --warmup 5 --no-end-to-end --no-other-configs > $D/prof_summary.txt 2>&1
timeout 600 bash tools/pmc.sh r06_final --no-end-to-end --no-other-configs > $D/pmc_stdout.txt 2>&1
timeout 900 python tools/bench_configs.py --only 3,4,5,6,7,10,11,12,13,14,15,16,17,18,19,20,21 --steps 600 > $D/other_configs.jsonl 2> $D/other_configs.err
timeout 300 python tools/bench_configs.py --only 9 --packets 2048 --force-generic --steps 100 >> $D/other_configs.jsonl 2>> $D/other_configs.err
timeout 400 python tools/bench_configs.py --only 3,11,12,14,15,16,20 --packets 16384 --steps 300 >> $D/other_configs.jsonl 2>> $D/other_configs.err
timeout 400 python tools/bench_configs.py --only 3,12,14 --packets 65536 --steps 100 >> $D/other_configs.jsonl 2>> $D/other_configs.err
for c in 16 17 18 19 20; do
  timeout 200 bash tools/prof_cfg.sh $c 200 r06_final/prof_cfg$c > $D/prof_cfg$c.txt 2>&1
  timeout 400 bash tools/pmc_cfg.sh $c r06_final > $D/pmc_cfg$c.txt 2>&1
done
timeout 900 python tools/fuzz_gpu_setups.py --setups 3000 --seed 20000 --procs 14 --quiet > $D/fuzz_gpu_setups.txt 2>&1
timeout 600 python tools/fuzz_gpu_setups.py --setups 2000 --seed 300000 --procs 14 --quiet --blocksizes 9:12,8:12 > $D/fuzz_gpu_setups_edge12.txt 2>&1
timeout 600 python tools/fuzz_gpu_mixed.py --rounds 100 --seed 81 > $D/fuzz_gpu_mixed_full.txt 2>&1; tail -1 $D/fuzz_gpu_mixed_full.txt > $D/fuzz_gpu_mixed.txt
timeout 600 python tools/fuzz_gpu_mixed.py --rounds 120 --seed 82 --mid > $D/fuzz_gpu_mid_full.txt 2>&1; tail -1 $D/fuzz_gpu_mid_full.txt > $D/fuzz_gpu_mid.txt
timeout 600 python tools/fuzz_gpu_mixed.py --rounds 80 --seed 83 --big > $D/fuzz_gpu_big_full.txt 2>&1; tail -1 $D/fuzz_gpu_big_full.txt > $D/fuzz_gpu_big.txt
{ for S in stereo surround51_bookless; do timeout 300 python tools/fuzz_gpu_entropy.py --packets 40000 --setup $S 2>&1 | tail -1 | cut -c1-200; done; } > $D/fuzz_gpu_entropy.txt 2>&1
# one stream through the container API (the reference's examples/perf.rs): 200 000 packets, packet by packet / look-ahead on both tiers
{ make -C examples > /dev/null 2>&1
  python tools/make_long_ogg.py /tmp/long.ogg 200000
  echo "packet by packet (first 20 000 packets of the file's length only: one synchronous round trip per packet):"; python tools/make_long_ogg.py /tmp/short.ogg 20000 > /dev/null; timeout 300 ./examples/perf /tmp/short.ogg 2>&1 | tail -2
  for K in 4096 16384; do
    echo "look-ahead K = $K, host entropy stage, 12 threads:"; timeout 300 ./examples/perf /tmp/long.ogg $K 12 2>&1 | tail -2
    echo "look-ahead K = $K, entropy stage on the device:"; timeout 300 ./examples/perf /tmp/long.ogg $K 2 dev 2>&1 | tail -2
    echo "packet by packet with lw_ogg_stream_set_read_ahead($K), host entropy stage, 12 threads:"; timeout 300 ./examples/perf /tmp/long.ogg $K 12 host ahead 2>&1 | tail -2
    echo "packet by packet with lw_ogg_stream_set_read_ahead($K), entropy stage on the device:"; timeout 300 ./examples/perf /tmp/long.ogg $K 2 dev ahead 2>&1 | tail -2
  done; } > $D/single_stream.txt 2>&1
{ echo "e2e_sharder: $(timeout 200 python tools/e2e_sharder.py 2>&1 | tail -1 | cut -c1-420)"
  echo "single ring, device entropy: $(timeout 200 python tools/e2e.py --batches 300 --device-entropy 2>&1 | tail -1 | cut -c1-300)"
  echo "single ring, device entropy, 16384-packet batches: $(timeout 200 python tools/e2e.py --batches 150 --packets 16384 --device-entropy 2>&1 | tail -1 | cut -c1-300)"
  echo "single ring, host entropy stage: $(timeout 200 python tools/e2e.py --batches 48 2>&1 | tail -1 | cut -c1-300)"; } > $D/end_to_end_sharder.txt 2>&1
python3 -c "
import json
d=json.loads([l for l in open('$D/bench.json') if l.startswith('{')][-1])
print('launch us', d['roofline']['launch_ms']*1e3, 'frac', d['roofline']['frac'], 'value M/s', d['value']/1e6, 'traffic', d['roofline']['traffic'])
e=d['end_to_end']; print('e2e', e['value'], 'dev', e.get('device_entropy',{}).get('value'), 'sharder', (e.get('sharder') or {}).get('value'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('synthesis_only',{}).get('value'))
for k,v in d['other_configs'].items(): print('  ', k[:70], v.get('us_per_launch'), v.get('frac'), v.get('kernels'), (v.get('parity') or v.get('error'))[:40])
"
tail -2 $D/prof_summary.txt
cut -c1-260 $D/other_configs.jsonl
grep -v '^setup' $D/fuzz_gpu_setups.txt | head -60
cat $D/single_stream.txt $D/end_to_end_sharder.txt $D/fuzz_gpu_mixed.txt $D/fuzz_gpu_mid.txt $D/fuzz_gpu_big.txt $D/fuzz_gpu_entropy.txt
for c in 16 17 18 19 20; do tail -6 $D/prof_cfg$c.txt; done
