#!/bin/bash
# (local) gpurun_out/r05_final -> profiles/r05_*
S=gpurun_out/r05_final; P=profiles
grep '^{' $S/bench.json | tail -1 > $P/r05_bench.json
cp $S/prof/stats_kernel_stats.csv $P/r05_kernel_stats.csv
cp $S/kernel_duration_summary.json $P/r05_kernel_duration_summary.json
cp $S/pmc_summary.json $P/r05_pmc_summary.json
cp $S/other_configs.jsonl $P/r05_other_configs.jsonl
cp $S/pmc_cfg_3.json $P/r05_pmc_mixed.json
cp $S/pmc_cfg_12.json $P/r05_pmc_k_long10.json
cp $S/pmc_cfg_11.json $P/r05_pmc_k_long12.json
cp $S/pmc_cfg_14.json $P/r05_pmc_k_mix10.json
for c in 12 11 14 15 3; do cp $S/prof_cfg$c/stats_kernel_stats.csv $P/r05_cfg${c}_kernel_stats.csv; done
cp $S/end_to_end_sharder.txt $P/r05_end_to_end_sharder.txt
cp $S/gpu_box_host.txt $P/r05_gpu_box_host.txt
tail -8 $S/pytest.txt > $P/r05_gpu_pytest.txt
cat $S/fuzz_gpu_mixed.txt $S/fuzz_gpu_mid.txt $S/fuzz_gpu_big.txt > $P/r05_fuzz_gpu_mixed.txt
cp $S/fuzz_gpu_entropy.txt $P/r05_fuzz_gpu_entropy.txt
bash tools/device_code_id.sh > $P/r05_device_code.sha256
cat $P/r05_device_code.sha256
