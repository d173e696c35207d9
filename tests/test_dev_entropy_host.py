"""The device entropy stage's per-packet algorithm (lewton_amd/csrc/lw_dev_entropy.h -- the very function the HIP kernel runs
in every lane) compiled for the host and held against the host entropy stage (lw::entropy_decode) bit for bit: floor
records and residue vectors of intact, truncated and bit-flipped packets of every eligible stream shape, under ASan/UBSan.
The GPU side of the same comparison is tests/test_gpu_dev_entropy.py."""
import os
import struct
import subprocess

import pytest

from common import HOST_SETUPS, ROOT
from lewton_amd import streamgen as sg

CS = os.path.join(ROOT, "lewton_amd", "csrc")
SRC = [os.path.join(ROOT, "tests", "san", "dev_entropy_host.cpp")] + [
    os.path.join(CS, n) for n in ("lw_dev_entropy.cpp", "lw_entropy.cpp", "lw_headers.cpp")]


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("deventropy") / "dev_entropy_host")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-ffp-contract=off"] + SRC + ["-o", exe])
    return exe


def _case(path, setup, pattern, count, **kw):
    idp, _, stp = setup.headers()
    pk = sg.make_stream(setup, pattern, count, **kw)
    with open(path, "wb") as f:
        f.write(struct.pack("<I", 1))
        for b in (idp, stp):
            f.write(struct.pack("<I", len(b)) + bytes(b))
        f.write(struct.pack("<I", len(pk)))
        for p in pk:
            f.write(struct.pack("<I", len(p)) + bytes(p))


@pytest.mark.parametrize("name,pattern", [("stereo", "LLSSLSL"), ("stereo_t1", "LSL"), ("mono_small", "LSSLL"), ("stereo_9_12", "LSL"),
                                          ("stereo_7_7", "LSL"), ("stereo_single_entry", "LLS"), ("surround51", "LLSL"),
                                          ("stereo_spill_t1", "LSLL"), ("stereo_spill_t2", "LLSL"), ("surround51_bookless", "LLSL"), ("multichannel12", "LLSL")])
def test_device_algorithm_equals_host_stage(harness, tmp_path, name, pattern):
    case = str(tmp_path / "case.bin")
    _case(case, HOST_SETUPS[name](), pattern, 60, seed=11, p_floor_unused=0.15)
    out = subprocess.run([harness, case, "6", "3"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "device entropy algorithm == host entropy stage" in out.stdout, out.stdout


def test_ineligible_streams_say_why(harness, tmp_path):
    from common import FLOOR0_SETUPS, SETUPS
    for setup, word in ((FLOOR0_SETUPS["floor0"](), "floor type 0"), (FLOOR0_SETUPS["floor0_mixed"](), "floor type 0")):
        case = str(tmp_path / "case.bin")
        _case(case, setup, "LS", 4, seed=1)
        out = subprocess.run([harness, case, "0"], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and "not eligible" in out.stdout and word in out.stdout, out.stdout + out.stderr


def test_real_encoder_stream_is_eligible_and_identical(harness, tmp_path):
    """tests/golden/invalid_keypress.ogg (libvorbis-made: books with codes beyond the two table levels, walked through the
    tree on the device): its 26 audio packets, each also cut and bit-flipped 300 times"""
    from oracle import pyogg
    rd = pyogg.PacketReader(open(os.path.join(ROOT, "tests", "golden", "invalid_keypress.ogg"), "rb").read())
    pk = []
    while True:
        p = rd.read_packet()
        if p is None:
            break
        pk.append(bytes(p.data))
    case = str(tmp_path / "case.bin")
    with open(case, "wb") as f:
        f.write(struct.pack("<I", 1))
        for b in (pk[0], pk[2]):
            f.write(struct.pack("<I", len(b)) + b)
        f.write(struct.pack("<I", len(pk) - 3))
        for p in pk[3:]:
            f.write(struct.pack("<I", len(p)) + p)
    out = subprocess.run([harness, case, "300", "9"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "device entropy algorithm == host entropy stage" in out.stdout, out.stdout


def test_random_setups_device_algorithm_equals_host_stage(harness, tmp_path):
    """round 6: setup headers drawn at random (streamgen.random_setup): wherever the device entropy stage is eligible its
    per-packet algorithm must equal the host stage -- floors of 2..65 posts with any class structure, every residue shape, books
    of both lookup types, sparse / ordered / one-entry books -- incl. cut and bit-flipped packets; ineligible setups say why."""
    import numpy as np
    n_eligible = 0
    for seed in range(2000, 2040):
        rng = np.random.default_rng(seed)
        setup = sg.random_setup(rng)
        idp, _, stp = setup.headers()
        pk = sg.random_stream(setup, rng, 10, seed=seed)
        case = str(tmp_path / "case.bin")
        with open(case, "wb") as f:
            f.write(struct.pack("<I", 1))
            for b in (idp, stp):
                f.write(struct.pack("<I", len(b)) + bytes(b))
            f.write(struct.pack("<I", len(pk)))
            for p in pk:
                f.write(struct.pack("<I", len(p)) + bytes(p))
        out = subprocess.run([harness, case, "4", str(seed)], capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, (seed, out.stdout[-2000:] + out.stderr[-4000:])
        if "not eligible" in out.stdout:
            assert any(isinstance(fl, sg.Floor0) for fl in setup.floors) or "LDS" in out.stdout, (seed, out.stdout)
            continue
        assert "device entropy algorithm == host entropy stage" in out.stdout, (seed, out.stdout)
        n_eligible += 1
    assert n_eligible >= 25
