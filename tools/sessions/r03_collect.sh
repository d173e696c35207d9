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
python3 - <<'PY'
import csv, json
out = {}
for p in (4096, 16384):
    for r in csv.DictReader(open("gpurun_out/r03_final/entropy_kernel_stats_%d.csv" % p)):
        if "k_entropy" in r["Name"]:
            out["k_entropy_%d_packets" % p] = {"launches": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                                "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
pm = {}
for line in open("gpurun_out/r03_final/entropy_pmc.txt"):
    w = line.split()
    if len(w) == 4 and w[2] == "per" and w[3] == "wave":
        pm[w[0]] = float(w[1])
out["per_wave_4096_packets"] = pm
out["command"] = "rocprofv3 --kernel-trace --stats -- python tools/ent_bench.py --packets N --reps 100; tools/ent_pmc.sh"
json.dump(out, open("profiles/r03_k_entropy_summary.json", "w"), indent=1, sort_keys=True)
PY
cp $S/fuzz_gpu_entropy.txt $P/r03_fuzz_gpu_entropy.txt
cp $S/fuzz_gpu_mixed.txt $P/r03_fuzz_gpu_mixed.txt
bash tools/device_code_id.sh > $P/r03_device_code.sha256
cat $P/r03_device_code.sha256
