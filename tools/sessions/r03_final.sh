#!/bin/bash
# (GPU box) the round's measurement set on the current build: GPU parity suite, driver-style bench line, rocprofv3 kernel trace
# + stats of the bench command, PMC passes (headline kernel; the two kernels of the mixed configuration), the other BASELINE
# configs with their oracle check, the device entropy stage (kernel duration, counters, differential runs), the end-to-end
# rates of ring and sharder, the host's CPU limits.  Everything lands in gpurun_out/r03_final/;
# tools/sessions/r03_collect.sh copies the summaries into profiles/.
D=gpurun_out/r03_final
mkdir -p $D
bash tools/host_limits.sh > $D/gpu_box_host.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q > $D/pytest.txt 2>&1
tail -4 $D/pytest.txt
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench.json 2> $D/bench.err
timeout 400 bash tools/prof.sh r03_final --steps 20 --warmup 5 --no-end-to-end > $D/prof_summary.txt 2>&1
timeout 600 bash tools/pmc.sh r03_final --no-end-to-end > $D/pmc_stdout.txt 2>&1
timeout 600 python tools/bench_configs.py --only 3,4,5,6,7,10,11,12 > $D/other_configs.jsonl 2> $D/other_configs.err
timeout 600 bash tools/pmc_mixed.sh r03_final > $D/pmc_mixed_stdout.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$D/prof_mixed -o stats -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --only 3 --steps 400 --no-verify > /dev/null 2> $GRAFT_REPO_ROOT/$D/rocprof_mixed.log
cd $GRAFT_REPO_ROOT
timeout 300 python tools/bench_configs.py --only 12 --packets 16384 --steps 200 >> $D/other_configs.jsonl 2>> $D/other_configs.err
# device entropy stage: kernel duration at two batch sizes, instruction counts per wave, differential runs against the host stage
for P in 4096 16384; do
  rm -rf /tmp/entprof
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/entprof -o s -- python $GRAFT_REPO_ROOT/tools/ent_bench.py --packets $P --reps 100 > /dev/null 2>&1)
  cp /tmp/entprof/s_kernel_stats.csv $D/entropy_kernel_stats_$P.csv 2>/dev/null
done
timeout 300 bash tools/ent_pmc.sh > $D/entropy_pmc.txt 2>&1
{ for S in stereo stereo_t1 mono_small surround51 spill_t1 spill_t2 real; do timeout 300 python tools/fuzz_gpu_entropy.py --packets 40000 --setup $S 2>&1 | tail -1 | cut -c1-200; done; } > $D/fuzz_gpu_entropy.txt 2>&1
timeout 900 python tools/fuzz_gpu_mixed.py --rounds 200 --seed 11 > $D/fuzz_gpu_mixed_full.txt 2>&1; tail -1 $D/fuzz_gpu_mixed_full.txt > $D/fuzz_gpu_mixed.txt
{ for a in "" "--copy-out" "--shards 3" "--host-entropy"; do echo "e2e_sharder $a: $(timeout 200 python tools/e2e_sharder.py $a 2>&1 | tail -1 | cut -c1-420)"; done
  echo "single ring, device entropy: $(timeout 200 python tools/e2e.py --batches 96 --device-entropy 2>&1 | tail -1 | cut -c1-300)"
  echo "single ring, device entropy, 16384-packet batches: $(timeout 200 python tools/e2e.py --batches 48 --packets 16384 --device-entropy 2>&1 | tail -1 | cut -c1-300)"
  echo "single ring, host entropy stage: $(timeout 200 python tools/e2e.py --batches 48 2>&1 | tail -1 | cut -c1-300)"; } > $D/end_to_end_sharder.txt 2>&1
python3 -c "
import json
d=json.loads(open('$D/bench.json').read().strip().splitlines()[-1])
print('launch us', d['roofline']['launch_ms']*1e3, 'frac', d['roofline']['frac'], 'value M/s', d['value']/1e6, 'traffic', d['roofline']['traffic'])
e=d['end_to_end']; print('e2e', e['value'], 'dev', e.get('device_entropy',{}).get('value'), 'sharder', (e.get('sharder') or {}).get('value'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('synthesis_only',{}).get('value'))
"
tail -2 $D/prof_summary.txt
cut -c1-230 $D/other_configs.jsonl
cat $D/end_to_end_sharder.txt
tail -25 $D/pmc_mixed_stdout.txt | head -12
grep k_entropy $D/entropy_kernel_stats_*.csv | cut -c1-260
cat $D/entropy_pmc.txt $D/fuzz_gpu_entropy.txt $D/fuzz_gpu_mixed.txt
