#!/usr/bin/env python3
"""Host entropy stage alone, one thread, no GPU: builds tools/micro/entropy_bench.cpp from the product sources and runs it
on the packets of the bench workload (512 stereo long-block packets, the pool bench.py and tools/e2e.py draw from).
    python tools/entropy_bench.py [--reps 30] [--runs 5]"""
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
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--runs", type=int, default=5, help="process launches (heap placement changes the result by ~10 %)")
ap.add_argument("--cxx", default="/opt/rocm/lib/llvm/bin/clang++", help="the compiler lewton_amd/build.py uses for host code")
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
    exe = os.path.join(tmp, "entropy_bench")
    src = [os.path.join(ROOT, "tools", "micro", "entropy_bench.cpp")] + [
        os.path.join(ROOT, "lewton_amd", "csrc", n) for n in ("lw_entropy.cpp", "lw_headers.cpp")]
    subprocess.check_call([args.cxx, "-std=c++17", "-O3", "-ffp-contract=off", "-fno-fast-math"] + src + ["-o", exe])
    for _ in range(args.runs):
        print(subprocess.check_output([exe, case, str(args.reps)], text=True).strip())
