#!/bin/bash
# (local) copies the summaries of tools/r02_final.sh's run from gpurun_out/r02_final/ into profiles/ (tracked)
D=gpurun_out/r02_final
P=profiles
tail -n 1 $D/bench.json > $P/r02_bench.json
cp $D/prof/stats_kernel_stats.csv $P/r02_kernel_stats.csv
cp $D/kernel_duration_summary.json $P/r02_kernel_duration_summary.json
cp $D/pmc_summary.json $P/r02_pmc_summary.json
cp $D/other_configs.jsonl $P/r02_other_configs.jsonl
{ cat $D/end_to_end.txt; echo; cat $D/single_stream.txt; } > $P/r02_end_to_end.txt
cp $D/gpu_box_host.txt $P/r02_gpu_box_host.txt
cp $D/stamps.txt $P/r02_stamps.txt
tools/device_code_id.sh > $P/r02_device_code.sha256
# device entropy stage (tools/r02_tierc.sh)
T=gpurun_out/r02_tierc
cp $T/k_entropy_summary.json $P/r02_tierc_k_entropy_summary.json
cp $T/prof4096/s_kernel_stats.csv $P/r02_tierc_kernel_stats.csv
{ cat $T/end_to_end.txt; cat $T/ent_bench_4096.txt $T/ent_bench_16384.txt; } > $P/r02_tierc_end_to_end.txt
