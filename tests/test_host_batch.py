"""Host side of the batch path without a GPU (CPU suite): tools/micro/batch_host_bench.cpp links the PRODUCT sources
(lw_runtime.cpp with its worker pool, lw_entropy.cpp, lw_headers.cpp, lw_fast.cpp) against stand-ins for the HIP runtime and,
in check mode, compares the batch on 1 thread, on several threads and the one-packet host hook.  Built with
ThreadSanitizer: a data race in the pool or in the staging writes aborts the run."""
import os
import struct
import subprocess

import pytest

from common import ROOT, sg

SRC = [os.path.join(ROOT, "tools", "micro", "batch_host_bench.cpp")] + [
    os.path.join(ROOT, "lewton_amd", "csrc", n) for n in ("lw_runtime.cpp", "lw_batch.cpp", "lw_packet.cpp", "lw_pool.cpp", "lw_dev_entropy.cpp", "lw_entropy.cpp", "lw_headers.cpp", "lw_fast.cpp")]
HIP_INC = "/opt/rocm/include"


def _case(path, setup, pattern, count, **kw):
    idp, _, stp = setup.headers()
    pool = sg.make_stream(setup, pattern, count, **kw)
    with open(path, "wb") as f:
        f.write(struct.pack("<I", 1))
        for b in (idp, stp):
            f.write(struct.pack("<I", len(b)) + bytes(b))
        f.write(struct.pack("<I", len(pool)))
        for p in pool:
            f.write(struct.pack("<I", len(p)) + bytes(p))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if not os.path.isdir(os.path.join(HIP_INC, "hip")):
        pytest.skip("HIP headers not installed")
    exe = str(tmp_path_factory.mktemp("hostbatch") / "batch_host_bench_tsan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__",
                           "-I" + HIP_INC] + SRC + ["-lpthread", "-o", exe])
    return exe


@pytest.mark.parametrize("name,pattern", [("stereo", "L"), ("stereo", "LLSSLSL"), ("surround51", "LLSL"), ("mono_small", "LSSLL"),
                                          ("stereo_9_12", "LLLLLSLLL"), ("stereo_6_13", "L")])   # (the last two: k_big's slot planner)
def test_batch_entropy_threads_agree_under_tsan(harness, tmp_path, name, pattern):
    from common import SETUPS
    case = str(tmp_path / "case.bin")
    _case(case, SETUPS[name](), pattern, 96, seed=7, p_floor_unused=0.1)
    env = dict(os.environ, LW_HOST_BENCH_CHECK="1", TSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([harness, case, "384", "12", "2", "0", "2", "5", "8"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "check ok: 384 packets" in out.stdout
    if (name, pattern) == ("stereo", "L"):
        # lw_batch_algorithmic_bytes / lw_batch_state_bytes: 12 streams x 32 long packets, the first of each without a previous
        # window (no samples, audio.rs:1140-1152); nothing stored comes in, every stream leaves 2 x 1024 floats behind.  Floor
        # bytes: 2 + 2 per post used (21 posts here), an unused floor still costs its record
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("bytes ")][0].split()
        assert int(line[2]) == 12 * 2 * 1024 * 4
        assert int(line[1]) == 384 * (8192 + 132) + (384 - 12) * 4096


@pytest.mark.parametrize("name,pattern", [("stereo", "LLSSLSL"), ("surround51", "LSSLL"), ("mono_small", "SLLSL"),
                                          ("stereo_7_7", "LSL")])
def test_batch_statuses_and_sample_counts_equal_the_oracle(harness, tmp_path, name, pattern):
    """What lw_batch_entropy decides on the host -- AudioReadError codes, samples per packet (audio.rs:1140-1152), output
    offsets, and how an error leaves the stream's previous-window state (:1107-1111) -- against the oracle decoding the same
    streams packet by packet.  Packets are damaged on purpose (cut short, bits flipped, a header packet in between)."""
    import numpy as np
    from common import SETUPS, oracle_headers, po
    setup = SETUPS[name]()
    n_streams, per = 8, 24
    rng = np.random.default_rng(4)
    streams = []
    for k in range(n_streams):
        pk = [bytearray(p) for p in sg.make_stream(setup, pattern, per, seed=40 + k, p_floor_unused=0.1)]
        for i in range(per):
            r = rng.random()
            if r < 0.08:
                pk[i] = pk[i][: int(rng.integers(0, max(1, len(pk[i]))))]
            elif r < 0.16 and len(pk[i]):
                pk[i][int(rng.integers(0, len(pk[i])))] ^= 1 << int(rng.integers(0, 8))
            elif r < 0.21:
                pk[i] = bytearray(b"\x01vorbis")
        streams.append([bytes(p) for p in pk])
    idp, _, stp = setup.headers()
    case = str(tmp_path / "case.bin")
    with open(case, "wb") as f:
        f.write(struct.pack("<I", 1))
        for b in (idp, stp):
            f.write(struct.pack("<I", len(b)) + bytes(b))
        f.write(struct.pack("<I", n_streams * per))
        for st in streams:
            for p in st:
                f.write(struct.pack("<I", len(p)) + p)
    env = dict(os.environ, LW_HOST_BENCH_CHECK="1", LW_HOST_BENCH_DUMP="1", LW_HOST_BENCH_FILE_ORDER="1",
               TSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([harness, case, str(n_streams * per), str(n_streams), "1", "0", "3"], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    got = [tuple(int(v) for v in l.split()[1:]) for l in out.stdout.splitlines() if l.startswith("R ")]
    o_id, o_st = oracle_headers(setup)
    ch = o_id.audio_channels
    want, off = [], 0
    for st in streams:
        pwr = po.Pwr()
        for p in st:
            try:
                n = po.read_audio_packet(o_id, o_st, p, pwr, "i16").shape[1]
                want.append((0, n, off))
                off += n * ch
            except po.OracleError as e:
                want.append((e.code, 0, off))
    assert len(got) == len(want)
    assert got == want
    assert sum(1 for w in want if w[0]) >= 3          # the damage took (a cut packet usually still decodes: audio.rs:655-660)


@pytest.fixture(scope="module")
def harness_asan(tmp_path_factory):
    if not os.path.isdir(os.path.join(HIP_INC, "hip")):
        pytest.skip("HIP headers not installed")
    return build_harness_asan(str(tmp_path_factory.mktemp("hostbatch_asan") / "batch_host_bench_asan"))


def random_setup_case(harness, seed, case, blocksizes=None, n_streams=6, per=14):
    """one random setup (streamgen.random_setup, seed `seed`; `blocksizes` pins the block sizes) with damaged multi-stream batches
    through the harness `harness`: statuses, samples per packet and output offsets against the oracle.  Returns the number of
    packets the oracle rejected.  (also driven by tools/fuzz_host_setups.py over many seeds)"""
    import numpy as np
    from common import po
    rng = np.random.default_rng(seed)
    setup = sg.random_setup(rng, blocksizes=blocksizes)
    streams = [sg.random_stream(setup, rng, per, seed=10 * seed + k, p_damage=0.15) for k in range(n_streams)]
    idp, _, stp = setup.headers()
    with open(case, "wb") as f:
        f.write(struct.pack("<I", 1))
        for b in (idp, stp):
            f.write(struct.pack("<I", len(b)) + bytes(b))
        f.write(struct.pack("<I", n_streams * per))
        for st in streams:
            for p in st:
                f.write(struct.pack("<I", len(p)) + p)
    env = dict(os.environ, LW_HOST_BENCH_CHECK="1", LW_HOST_BENCH_DUMP="1", LW_HOST_BENCH_FILE_ORDER="1",
               ASAN_OPTIONS="detect_leaks=0")   # (the bench harness ends without tearing its decoder down)
    out = subprocess.run([harness, case, str(n_streams * per), str(n_streams), "1", "0", "2"], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, (seed, out.stdout[-2000:] + out.stderr[-4000:])
    got = [tuple(int(v) for v in l.split()[1:]) for l in out.stdout.splitlines() if l.startswith("R ")]
    o_id = po.Ident(idp)
    o_st = po.Setup(stp, o_id)
    ch = o_id.audio_channels
    want, off, n_bad = [], 0, 0
    for st in streams:
        pwr = po.Pwr()
        for p in st:
            try:
                n = po.read_audio_packet(o_id, o_st, p, pwr, "i16").shape[1]
                want.append((0, n, off))
                off += n * ch
            except po.OracleError as e:
                want.append((e.code, 0, off))
                n_bad += 1
    assert got == want, seed
    return n_bad


def build_harness_asan(exe):
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off",
                           "-D__HIP_PLATFORM_AMD__", "-I" + HIP_INC] + SRC + ["-lpthread", "-o", exe])
    return exe


def test_random_setups_plan_and_statuses_under_asan(harness_asan, tmp_path):
    """round 6: setup headers drawn at random (streamgen.random_setup) through the product's host side under ASan / UBSan -- header
    parser, the planner of the block classes (plan_units: native units, coupling steps inside the waves, the canonicalising
    pre-pass with the unit floor), the batch planner (work lists, halo items, slot lists, k_prep's packet list, the long-block
    kernels' edge forms) -- with the kernels as no-ops: statuses, samples per packet and output offsets of damaged multi-stream
    batches against the oracle.  Twelve draws from the generator's menu, three each pinned to 512/4096 and 256/1024 blocks."""
    n_bad = 0
    case = str(tmp_path / "case.bin")
    for seed in range(3000, 3012):
        n_bad += random_setup_case(harness_asan, seed, case)
    for seed in range(3100, 3103):
        n_bad += random_setup_case(harness_asan, seed, case, blocksizes=(9, 12))
        n_bad += random_setup_case(harness_asan, seed + 50, case, blocksizes=(8, 10))
    assert n_bad >= 3
