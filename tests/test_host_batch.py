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
    os.path.join(ROOT, "lewton_amd", "csrc", n) for n in ("lw_runtime.cpp", "lw_entropy.cpp", "lw_headers.cpp", "lw_fast.cpp")]
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


@pytest.mark.parametrize("name,pattern,symbols", [("stereo", "L", 0), ("stereo", "LLSSLSL", 0), ("stereo", "LSL", 1),
                                                  ("surround51", "LLSL", 0), ("mono_small", "LSSLL", 0)])
def test_batch_entropy_threads_agree_under_tsan(harness, tmp_path, name, pattern, symbols):
    from common import SETUPS
    case = str(tmp_path / "case.bin")
    _case(case, SETUPS[name](), pattern, 96, seed=7, p_floor_unused=0.1)
    env = dict(os.environ, LW_HOST_BENCH_CHECK="1", TSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([harness, case, "384", "12", "2", str(symbols), "2", "5", "8"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "check ok: 384 packets" in out.stdout
