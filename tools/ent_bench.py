#!/usr/bin/env python3
"""(GPU box) k_entropy alone, back to back: one 4096-packet batch of the bench workload in device-entropy mode, uploaded and
decoded `--reps` times on one stream (the GPU stays busy, clocks up).  Prints the wall time per repetition; run under
rocprofv3 --kernel-trace --stats for the kernel's own duration.
    python tools/ent_bench.py [--packets 4096] [--streams 256] [--reps 200]"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from lewton_amd import _native as N, audio, header, streamgen as sg  # noqa: E402
from lewton_amd.batch import Batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--packets", type=int, default=4096)
ap.add_argument("--streams", type=int, default=256)
ap.add_argument("--reps", type=int, default=200)
args = ap.parse_args()
setup = sg.stereo_setup(44100, 8, 11)
idp, _, stp = setup.headers()
ident = header.read_header_ident(idp)
st = header.read_header_setup(stp, 2, (8, 11))
dec = audio.decoder_for(ident, st, 0)
pool = sg.make_stream(setup, "L", 512, seed=9)
rng = np.random.default_rng(1)
per = args.packets // args.streams
pw = [audio.PreviousWindowRight() for _ in range(args.streams)]
b = Batch(dec, args.packets, "i16")
assert b.set_entropy_on_device(True)
b.entropy([(pool[int(i)], pw[k // per]) for k, i in enumerate(rng.integers(0, len(pool), args.packets))], n_threads=1)
for _ in range(5):
    b.upload()
    assert N.lw_batch_device_entropy(b._h, None) == 0
flat = b.synth_to_host()
t0 = time.perf_counter()
for _ in range(args.reps):
    b.upload()
    assert N.lw_batch_device_entropy(b._h, None) == 0
b.synth_to_host()
dt = time.perf_counter() - t0
print("upload + k_entropy: %.1f us per %d-packet batch (%d repetitions, one stream)" % (dt / args.reps * 1e6, args.packets, args.reps))
