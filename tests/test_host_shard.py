"""The one-process multi-device path (lw_sharder_*, csrc/lw_shard.cpp) in the CPU suite: tests/san/shard_host.cpp links the
PRODUCT sources against stand-ins for the HIP runtime (tests/san/hip_standins.inc) and runs G logical shards with three calls
in flight, both collect forms in turn, against a one-shard sharder fed call by call -- statuses, sample counts and block
offsets of every packet (damaged ones included) must agree.  Built with ThreadSanitizer: the shards' worker threads, the
shared entropy pool and the callers' hand-overs race-free.  Sample VALUES of the same path: tests/test_gpu_shapes.py."""
import os
import struct
import subprocess

import numpy as np
import pytest

from common import ROOT, SETUPS, sg

CS = os.path.join(ROOT, "lewton_amd", "csrc")
SRC = [os.path.join(ROOT, "tests", "san", "shard_host.cpp")] + [
    os.path.join(CS, n) for n in ("lw_shard.cpp", "lw_ring.cpp", "lw_runtime.cpp", "lw_batch.cpp", "lw_packet.cpp", "lw_pool.cpp",
                                  "lw_dev_entropy.cpp", "lw_entropy.cpp", "lw_headers.cpp", "lw_fast.cpp")]
HIP_INC = "/opt/rocm/include"


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if not os.path.isdir(os.path.join(HIP_INC, "hip")):
        pytest.skip("HIP headers not installed")
    exe = str(tmp_path_factory.mktemp("hostshard") / "shard_host_tsan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__",
                           "-I" + HIP_INC] + SRC + ["-lpthread", "-o", exe])
    return exe


def _case(path, setup, pattern, count, seed):
    """a pool of consecutive packets of ONE stream, some of them damaged (cut short, bits flipped, a header packet)"""
    idp, _, stp = setup.headers()
    rng = np.random.default_rng(seed)
    pool = []
    for p in sg.make_stream(setup, pattern, count, seed=seed, p_floor_unused=0.1):
        p = bytearray(p)
        k = rng.integers(0, 12)
        if k == 0 and len(p) > 2:
            p = p[:int(rng.integers(1, len(p)))]
        elif k == 1:
            p[int(rng.integers(0, len(p)))] ^= 1 << int(rng.integers(0, 8))
        elif k == 2:
            p = bytearray(idp)
        pool.append(bytes(p))
    with open(path, "wb") as f:
        f.write(struct.pack("<I", 1))
        for b in (idp, stp):
            f.write(struct.pack("<I", len(b)) + bytes(b))
        f.write(struct.pack("<I", len(pool)))
        for p in pool:
            f.write(struct.pack("<I", len(p)) + p)


@pytest.mark.parametrize("name,pattern,shards,streams,per,calls,dev", [
    ("stereo", "LLSSLSL", 3, 10, 4, 9, 0), ("stereo", "L", 2, 7, 5, 7, 1), ("surround51", "LLSL", 4, 9, 3, 6, 0),
    ("mono_small", "LSSLL", 8, 20, 2, 5, 1)])
def test_pipelined_shards_agree_with_one_shard_call_by_call(harness, tmp_path, name, pattern, shards, streams, per, calls, dev):
    case = str(tmp_path / "case.bin")
    _case(case, SETUPS[name](), pattern, 90, seed=41)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([harness, case, str(shards), str(streams), str(per), str(calls), str(dev)], env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "sharder ok: %d shards, %d calls of %d packets" % (shards, calls, streams * per) in out.stdout, out.stdout


def test_a_failed_launch_restarts_one_shard_and_older_calls_say_so(harness, tmp_path):
    """a launch failure on one shard with older calls in flight on it -- the older calls' packets of that shard
    come back LW_ERR_DEVICE (not another call's data), the other shards' are intact, every collect consumes its call, and the
    shard decodes again afterwards (stand-in failure injection, ThreadSanitizer)"""
    case = str(tmp_path / "case.bin")
    _case(case, SETUPS["stereo"](), "L", 60, seed=7)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([harness, case, "3", "9", "4", "6", "0", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "sharder failure ok" in out.stdout, out.stdout
