#!/usr/bin/env python3
"""Host side of the batch path (lw_batch_entropy: prologue pass, threaded entropy decode, planning pass) timed WITHOUT a
GPU: tools/micro/batch_host_bench.cpp links the product's host sources against stand-ins for the HIP runtime.
    python tools/batch_host_bench.py [--packets 4096] [--streams 256] [--reps 20] [--threads 1 2 4 8]"""
import argparse
import os
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lewton_amd import streamgen as sg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--packets", type=int, default=4096)
ap.add_argument("--streams", type=int, default=256)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--threads", type=int, nargs="*", default=[])
ap.add_argument("--cxx", default="/opt/rocm/lib/llvm/bin/clang++")
args = ap.parse_args()

setup = sg.stereo_setup(44100, 8, 11)
idp, _, stp = setup.headers()
pool = sg.make_stream(setup, "L", 512, seed=9)
with tempfile.TemporaryDirectory() as tmp:
    case = os.path.join(tmp, "case.bin")
    with open(case, "wb") as f:
        f.write(struct.pack("<I", 1))
        for b in (idp, stp):
            f.write(struct.pack("<I", len(b)) + bytes(b))
        f.write(struct.pack("<I", len(pool)))
        for p in pool:
            f.write(struct.pack("<I", len(p)) + bytes(p))
    exe = os.path.join(tmp, "batch_host_bench")
    src = [os.path.join(ROOT, "tools", "micro", "batch_host_bench.cpp")] + [
        os.path.join(ROOT, "lewton_amd", "csrc", n) for n in ("lw_runtime.cpp", "lw_batch.cpp", "lw_packet.cpp", "lw_pool.cpp", "lw_dev_entropy.cpp", "lw_entropy.cpp", "lw_headers.cpp", "lw_fast.cpp")]
    subprocess.check_call([args.cxx, "-std=c++17", "-O3", "-ffp-contract=off", "-fno-fast-math", "-D__HIP_PLATFORM_AMD__",
                           "-I/opt/rocm/include"] + src + ["-lpthread", "-o", exe])
    subprocess.check_call([exe, case, str(args.packets), str(args.streams), str(args.reps), "0"]
                          + [str(t) for t in args.threads])
