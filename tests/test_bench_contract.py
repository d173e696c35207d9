"""The bench line the driver consumes: the committed profiles/r0N_bench.json (output of bench.py on an MI355X) must carry
every field of the contract, with consistent numbers."""
import json
import os

from common import ROOT


import pytest


@pytest.mark.parametrize("name", ["r01_bench.json", "r02_bench.json", "r03_bench.json", "r04_bench.json", "r05_bench.json",
                                  "r05_bench_closing.json", "r05_bench_closing_box2.json", "r06_bench.json"])
def test_committed_bench_line_has_the_contract_fields(name):
    line = open(os.path.join(ROOT, "profiles", name)).readline()
    d = json.loads(line)
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"].split(" (")[0] == base["metric"].split(" (")[0] and d["unit"] == "packets/s"
    for k in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic bytes per launch / launch duration (12 420 B per stereo long packet)
    assert r["algorithmic_bytes_per_launch"] == 4096 * 12420
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9) < 1e-3 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] >= 0.9 * r["algorithmic_bytes_per_launch"]
    # value = packets of all ranks / wall time per step
    assert abs(d["value"] - 4096 * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "packets/s" and c["value"] > 0 and "sample" in c
    assert "bit-exact" in d["config"]["parity"]
    if name >= "r02":
        # round 2: the timed batch itself is verified, and the PCIe-inclusive rates ride along (never as `value`)
        assert "4096 packets" in d["config"]["parity"] or "256 streams" in d["config"]["parity"]
        e = d["end_to_end"]
        assert e["unit"] == "packets/s" and 0 < e["value"] < d["value"] and e["host_cpus_usable"] >= 1
        t = e.get("device_entropy") or e["tier_c"]   # (round 2 called it tier_c)
        assert "entropy stage on the device" in t["records"] and "k_entropy" in t["kernels"] and t["value"] > 0
    if name >= "r03":
        # round 3: the like-for-like CPU figure rides along, and the one-process multi-device path has its end-to-end leg
        assert c["synthesis_only"]["value"] > c["value"]
        sh = d["end_to_end"]["sharder"]
        assert sh["unit"] == "packets/s" and sh["shards"] >= 2 and sh["value"] > 0
    if name >= "r04":
        # round 4: every quoted shape under the same clock, each with the oracle check of its timed batch (never `value`)
        oc = d["other_configs"]
        assert len(oc) >= 6
        for label, e in oc.items():
            assert "bit-exact" in e["parity"], label
            assert e["us_per_launch"] > 0 and 0 < e["frac"] < 1
            assert abs(e["frac"] - e["algorithmic_bytes_per_launch"] / (e["us_per_launch"] * 1e-6) / 8e12) < 2e-3, label
        assert any("k_mix" in e["kernels"] for e in oc.values())
    if name >= "r05":
        # round 5: every entry is timed on a footprint the 256 MiB Infinity Cache cannot hold (>= 0.5 GiB of algorithmic bytes per
        # rotation), the two mixed block-size pairs of low-rate Vorbis are there, and the wave-pipeline kernels serve 1024 / 4096 points
        oc = d["other_configs"]
        assert len(oc) >= 8
        for label, e in oc.items():
            assert e["footprint_bytes"] >= 5e8 and e["footprint_bytes"] == e["batches_rotated"] * e["algorithmic_bytes_per_launch"], label
        ks = {e["kernels"] for e in oc.values()}
        assert {"k_long10", "k_long12", "k_mix10", "k_mix"} <= ks, ks
        if name < "r06":
            assert not any("generic" in k for k in ks)
        assert any(("r0%d_pmc_summary" % n) in d["roofline"]["traffic_unit"] for n in (4, 5, 6))
    if name >= "r06":
        # round 6: the fallback every specialised kernel is measured against is ON the line (forced: the only entry on the generic
        # kernels), SURVEY 8(d)'s config 3 as written (ONE stream) and its all-long counterpart, a stream shape behind the
        # canonicalising pre-pass, libvorbis' 5.1 coupling inside k_long's waves at >= 30 %, the mixed shapes at 16 384 packets
        oc = d["other_configs"]
        gen = [label for label, e in oc.items() if "k_imdct_generic" in e["kernels"]]
        assert gen == ["generic fallback (stereo 8/11 long blocks, forced)"] and oc[gen[0]]["frac"] < 0.08
        # a pair of block sizes without an edge form: only the transition blocks' overlap-add is left to a generic kernel
        td = [e for label, e in oc.items() if "1024/4096" in label]
        assert len(td) == 1 and td[0]["kernels"] == "k_long12,k_short,k_ola_generic" and td[0]["frac"] >= 0.14
        one = [e for label, e in oc.items() if "ONE stream" in label]
        assert len(one) == 2 and all(e["packets_per_launch"] == 4096 for e in one)
        assert any(e["kernels"] == "k_prep,k_long" for e in oc.values())
        lv = [e for label, e in oc.items() if "libvorbis' coupling" in label]
        assert len(lv) == 1 and lv[0]["kernels"] == "k_long" and lv[0]["frac"] >= 0.30
        # libvorbis' low-bitrate block sizes: the long blocks next to short ones in k_long12's EDGE form, none on the generic kernels
        lo = [e for label, e in oc.items() if "512/4096" in label]
        assert len(lo) == 1 and lo[0]["kernels"] == "k_long12,k_short" and lo[0]["frac"] >= 0.20
        assert sum(1 for e in oc.values() if e["packets_per_launch"] == 16384) == 3
    if name >= "r05_bench_closing":
        # the round's closing tree: the window state crossing HBM at the launch boundary rides along (never in `frac`), and two
        # logical shards on one GPU reach the rate of one ring (the copier thread of tenants' rings)
        assert d["roofline"]["state_bytes_per_launch"] == 256 * 2 * (2 * 1024 * 4)
        for label, e in d["other_configs"].items():
            want = (e["algorithmic_bytes_per_launch"] + e["state_bytes_per_launch"]) / (e["us_per_launch"] * 1e-6) / 8e12
            assert e["frac"] <= e["frac_incl_state"] < 1 and abs(e["frac_incl_state"] - want) < 2e-3, label
        e2e = d["end_to_end"]
        # (host-bound legs follow the box: the committed round-6 line was taken with a load average of 13-40 from the box's other
        # tenants, profiles/r06_gpu_box_host.txt -- 10.1 M there, 10.9-12.2 M on the round's quieter boxes)
        assert e2e["sharder"]["value"] >= 0.8 * e2e["device_entropy"]["value"] and e2e["sharder"]["value"] >= 10e6


def test_device_code_is_the_measured_build():
    """profiles/r06_device_code.sha256 identifies the kernels the round's GPU parity run, bench line and rocprof summaries were
    taken on (sha256 of the gfx950 disassembly, tools/device_code_id.sh).  Host-side work done without a GPU at hand must
    not change them: a kernel edit has to go through the GPU tests again and refresh the file together with the profiles."""
    import shutil
    import subprocess
    if not (shutil.which("objcopy") and os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump")):
        import pytest
        pytest.skip("binutils / ROCm LLVM tools not installed")
    import lewton_amd  # noqa: F401  (makes sure the library is built)
    out = subprocess.check_output([os.path.join(ROOT, "tools", "device_code_id.sh")], cwd=ROOT, text=True)
    want = open(os.path.join(ROOT, "profiles", "r06_device_code.sha256")).read()
    assert out.split() == want.split()
