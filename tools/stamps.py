#!/usr/bin/env python3
"""Per-wave s_memtime breakdown of k_long (needs a library built with LW_EXTRA_FLAGS=-DLW_STAMPS).
Usage (on the GPU box): LW_EXTRA_FLAGS=-DLW_STAMPS python lewton_amd/build.py --force && python tools/stamps.py [streams] [packets]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lewton_amd import _native as N  # noqa: E402
from lewton_amd import audio, header, streamgen as sg  # noqa: E402
from lewton_amd.batch import Batch  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
NSTAMP = 16
setup = sg.stereo_setup(44100, 8, 11)
idp, _, stp = setup.headers()
ident = header.read_header_ident(idp)
st = header.read_header_setup(stp, 2, (8, 11))
dec = audio.decoder_for(ident, st, 0)
pool = sg.make_stream(setup, "L", 256, seed=5)
rng = np.random.default_rng(3)
per = NP // S
batches, outs = [], []
for b in range(4):
    spw = [audio.PreviousWindowRight() for _ in range(S)]
    prime = Batch(dec, S, "i16")
    prime.entropy([(pool[int(rng.integers(0, 256))], pw) for pw in spw])
    prime.upload(None)
    prime.synth_to_host(None)
    prime.close()
    bt = Batch(dec, NP, "i16")
    order = rng.integers(0, 256, NP)
    bt.entropy([(pool[int(i)], spw[k // per]) for k, i in enumerate(order)])
    bt.upload(None)
    outs.append(torch.empty(bt.out_elems, dtype=torch.int16, device="cuda"))
    batches.append((bt, spw))
torch.cuda.synchronize()
nslots = 4096 * 16 * NSTAMP
buf = torch.zeros(nslots, dtype=torch.int64, device="cuda")
N.lib.lw_debug_set_stamp_buffer.argtypes = [C.c_void_p]
assert N.lib.lw_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr())) == 0
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(6):
    bt, _ = batches[rep % 4]
    buf.zero_()
    torch.cuda.synchronize()
    ev0.record()
    bt.synth(C.c_void_p(outs[rep % 4].data_ptr()), outs[rep % 4].numel(), None)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1)
    a = buf.cpu().numpy().reshape(-1, NSTAMP)
    a = a[a[:, 0] != 0]
    if rep < 2:
        continue
    print("rep %d: kernel(event) %.1f us, waves stamped %d" % (rep, ms * 1e3, len(a)))
    names = [(0, "entry"), (2, "image+sync"), (1, "floor-first: floor done"), (15, "residue loads issued"), (3, "residue landed"),
             (4, "spectrum done"), (8, "IMDCT done"), (9, "hand-over done"), (10, "finish issued"), (11, "all stores done")]
    # s_memtime counters are not synchronised across the chip: reference every wave to the earliest entry of its
    # own workgroup (all its waves run on one CU)
    full = buf.cpu().numpy().reshape(-1, 16, NSTAMP)   # [block][wave][stamp]
    ent = np.where(full[:, :, 0] != 0, full[:, :, 0], np.iinfo(np.int64).max).min(axis=1)
    absd = full - ent[:, None, None]
    ok = full[:, :, 0] != 0
    if rep == 5:
        cols = [i for i, _ in names if full[:, :, i].max() != 0]
        print("  per-wave medians (cycles since workgroup entry)")
        print("  wave " + " ".join("%8s" % ("s%d" % c) for c in cols) + " | landed->done")
        for w in range(16):
            if not ok[:, w].any():
                continue
            med = {}
            for c in cols:
                m = ok[:, w] & (full[:, w, c] != 0)
                med[c] = float(np.median(absd[:, w, c][m])) if m.any() else float("nan")
            print("  %4d " % w + " ".join("%8.0f" % med[c] for c in cols) + " | %8.0f" % (med.get(11, float("nan")) - med.get(3, float("nan"))))
        for i, nm in names:
            print("  s%-2d = %s" % (i, nm))
