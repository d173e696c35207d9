#!/bin/bash
# (local) gpurun_out/r06_final -> profiles/r06_*
S=gpurun_out/r06_final; P=profiles
grep '^{' $S/bench.json | tail -1 > $P/r06_bench.json
cp $S/prof/stats_kernel_stats.csv $P/r06_kernel_stats.csv
cp $S/kernel_duration_summary.json $P/r06_kernel_duration_summary.json
cp $S/pmc_summary.json $P/r06_pmc_summary.json
cp $S/other_configs.jsonl $P/r06_other_configs.jsonl
cp $S/pmc_cfg_16.json $P/r06_pmc_k_long_pre.json
cp $S/pmc_cfg_19.json $P/r06_pmc_k_prep.json
cp $S/pmc_cfg_17.json $P/r06_pmc_single_stream_mixed.json
cp $S/pmc_cfg_18.json $P/r06_pmc_single_stream_long.json
cp $S/pmc_cfg_20.json $P/r06_pmc_k_long12_edge.json
for c in 16 17 18 19 20; do cp $S/prof_cfg$c/stats_kernel_stats.csv $P/r06_cfg${c}_kernel_stats.csv; done
cp $S/end_to_end_sharder.txt $P/r06_end_to_end_sharder.txt
cp $S/single_stream.txt $P/r06_single_stream.txt
cp $S/gpu_box_host.txt $P/r06_gpu_box_host.txt
tail -8 $S/pytest.txt > $P/r06_gpu_pytest.txt
grep -v '^setup' $S/fuzz_gpu_setups.txt > $P/r06_fuzz_gpu_setups.txt
grep -v '^setup' $S/fuzz_gpu_setups_edge12.txt > $P/r06_fuzz_gpu_setups_edge12.txt
cat $S/fuzz_gpu_mixed.txt $S/fuzz_gpu_mid.txt $S/fuzz_gpu_big.txt > $P/r06_fuzz_gpu_mixed.txt
cp $S/fuzz_gpu_entropy.txt $P/r06_fuzz_gpu_entropy.txt
bash tools/device_code_id.sh > $P/r06_device_code.sha256
cat $P/r06_device_code.sha256
