#!/bin/bash
# (local) gpurun_out/r03_final -> profiles/r03_*
S=gpurun_out/r03_final; P=profiles
cp $S/bench.json $P/r03_bench.json
cp $S/prof/stats_kernel_stats.csv $P/r03_kernel_stats.csv
cp $S/kernel_duration_summary.json $P/r03_kernel_duration_summary.json
cp $S/pmc_summary.json $P/r03_pmc_summary.json
cp $S/other_configs.jsonl $P/r03_other_configs.jsonl
cp $S/pmc_mixed.json $P/r03_pmc_mixed.json
cp $S/prof_mixed/stats_kernel_stats.csv $P/r03_mixed_kernel_stats.csv
cp $S/end_to_end_sharder.txt $P/r03_end_to_end_sharder.txt
cp $S/gpu_box_host.txt $P/r03_gpu_box_host.txt
tail -3 $S/pytest.txt > $P/r03_gpu_pytest.txt
bash tools/device_code_id.sh > $P/r03_device_code.sha256
cat $P/r03_device_code.sha256
