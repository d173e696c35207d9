"""bench.py as the driver runs it at N > 1, on the one GPU a test box has (-m gpu): `--force-dist` initialises the RCCL process
group for a single rank, so the barrier, the MAX all-reduce of the ranks' spans (lewton_amd/shard.py: max_elapsed) and the
N > 1 shape of the JSON line all execute.  The 1/2/4/8-GPU runs themselves are the driver's (one process per GPU under
torch.distributed.run); this keeps the path they take from rotting between rounds."""
import json
import os
import subprocess
import sys

import pytest

from common import ROOT

pytestmark = pytest.mark.gpu


def _bench(*flags, env=None):
    e = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, env=e, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]   # ONE JSON line on rank 0
    return json.loads(lines[0])


def test_bench_line_through_the_distributed_path_on_one_gpu():
    d = _bench("--gpus", "1", "--force-dist", "--steps", "20", "--warmup", "8", "--no-end-to-end", "--no-other-configs",
               "--no-cpu-baseline")
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["unit"] == "packets/s" and d["scaling"] == "weak"
    assert d["cpu_baseline"] is None and d["end_to_end"] is None and d["other_configs"] is None
    # value = packets of all ranks / the MAX over the ranks' spans (one rank here), the span ending before the closing barrier
    assert abs(d["value"] - 4096 * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "bit-exact" in d["config"]["parity"] and d["config"]["kernels"] == "k_long"
    r = d["roofline"]
    assert r["algorithmic_bytes_per_launch"] == 4096 * 12420 and 0.05 < r["frac"] < 1.0
    assert r["launch_ms"] <= d["ms_per_step"] * 1.001   # device time of the K steps <= their wall-clock span


def test_two_ranks_on_one_gpu_run_the_n_gt_1_path():
    """The driver's N = 2 command line, with the two ranks sharing device 0 (--device-of-rank 0,0) and gloo carrying the barrier
    and the MAX all-reduce (LW_BENCH_BACKEND=gloo: RCCL cannot form a group of two ranks on ONE device).  Everything rank-dependent
    in bench.py executes: per-rank seeds, rank 1's own parity check, rank 0 alone printing, cpu_baseline / end_to_end /
    other_configs null at N > 1, value = packets of both ranks / the slower rank's span."""
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LW_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "8",
                          "--device-of-rank", "0,0"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]   # ONE JSON line, rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["scaling"] == "weak"
    assert d["cpu_baseline"] is None and d["end_to_end"] is None and d["other_configs"] is None
    ranks = d["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and ranks[0]["pool_seed"] != ranks[1]["pool_seed"] and ranks[0]["order_seed"] != ranks[1]["order_seed"]
    assert all("bit-exact" in r["parity"] for r in ranks), ranks
    slowest = max(r["elapsed_s"] for r in ranks)
    assert abs(d["ms_per_step"] - slowest / 20 * 1e3) < 1e-9 * max(1.0, d["ms_per_step"])
    assert abs(d["value"] - 2 * 4096 * 20 / slowest) < 1e-6 * d["value"]
