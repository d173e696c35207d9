#!/bin/bash
# (GPU box) the round's measurement set on the current build: driver-style bench line, rocprofv3 kernel trace + stats of the
# bench command, PMC passes, the other BASELINE configs, the end-to-end sweep, the stamp timeline of k_long, the host's
# CPU limits.  Everything lands in gpurun_out/r02_final/; tools/r02_collect.sh copies the summaries into profiles/.
D=gpurun_out/r02_final
mkdir -p $D
bash tools/host_limits.sh > $D/gpu_box_host.txt 2>&1
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench.json 2> $D/bench.err
timeout 400 bash tools/prof.sh r02_final --steps 20 --warmup 5 --no-end-to-end > $D/prof_summary.txt 2>&1
timeout 600 bash tools/pmc.sh r02_final --no-end-to-end > $D/pmc_stdout.txt 2>&1
timeout 400 python tools/bench_configs.py --only 3,4,5,10 > $D/other_configs.jsonl 2> $D/other_configs.err
timeout 400 bash tools/e2e_sweep.sh 500 > $D/end_to_end.txt 2>&1
for t in 16 20; do echo "tier B, threads $t: $(timeout 120 python tools/e2e.py --batches 500 --threads $t --device-vq 2>&1 | tail -1 | cut -c1-300)" >> $D/end_to_end.txt; done
make -C examples perf > $D/perf_build.log 2>&1
python tools/make_long_ogg.py /tmp/long.ogg 200000 > /dev/null 2>&1
{ echo "examples/perf on a 200 000-packet synthetic stereo file (one stream through the Ogg layer):";
  for cfg in "" "4096 12" "16384 12" "16384 16"; do echo "perf long.ogg $cfg: $(timeout 200 examples/perf /tmp/long.ogg $cfg 2>&1 | tail -1)"; done; } > $D/single_stream.txt 2>&1
timeout 300 bash tools/r02_stamps.sh > $D/stamps.txt 2>&1
python3 -c "
import json
d=json.loads(open('$D/bench.json').read().strip().splitlines()[-1])
print('launch us', d['roofline']['launch_ms']*1e3, 'frac', d['roofline']['frac'], 'value M/s', d['value']/1e6, 'traffic', d['roofline']['traffic'])
print('e2e', d['end_to_end']['value'], 'tier_b', d['end_to_end'].get('tier_b',{}).get('value'))
print('cpu', d['cpu_baseline']['value'])
"
cat $D/prof_summary.txt | tail -2
tail -5 $D/other_configs.jsonl | cut -c1-200
cat $D/end_to_end.txt
cat $D/single_stream.txt
