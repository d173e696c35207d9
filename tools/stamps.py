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
NSTAMP = 64
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
    names = [(0, "entry"), (12, "kernargs in SGPRs"), (13, "unit loaded"), (14, "before issue_loads"), (15, "after issue_loads"), (1, "image loads issued"), (2, "image+sync"), (3, "r0 residue landed"), (4, "r0 floor+spec"),
             (5, "r0 stage B"), (6, "r0 stage C"), (7, "r0 stage D"), (8, "r0 stage E"), (9, "r0 handover done"), (10, "r0 phase2 issued"),
             (19, "r1 residue landed"), (20, "r1 floor+spec"), (21, "r1 stage B"), (22, "r1 stage C"), (23, "r1 stage D"),
             (24, "r1 stage E"), (25, "r1 handover"), (26, "r1 phase2 issued"), (59, "all stores done")]
    # s_memtime counters are not synchronised across the chip: reference every wave to the earliest entry of its
    # own workgroup (all its waves run on one CU)
    full = buf.cpu().numpy().reshape(-1, 16, NSTAMP)   # [block][wave][stamp]
    ent = np.where(full[:, :, 0] != 0, full[:, :, 0], np.iinfo(np.int64).max).min(axis=1)
    absd = full - ent[:, None, None]
    ok = full[:, :, 0] != 0
    for i, nm in names:
        if full[:, :, i].max() == 0:
            continue
        m = ok & (full[:, :, i] != 0)
        col = absd[:, :, i][m]
        early = absd[:, :8, i][m[:, :8]]
        lateh = absd[:, 8:, i][m[:, 8:]]
        print("  %2d %-20s since WG entry min/med/max %7d %7d %7d | waves0-7 med %7d max %7d | waves8-15 med %7d max %7d" % (
            i, nm, col.min(), np.median(col), col.max(), np.median(early) if len(early) else -1, early.max() if len(early) else -1,
            np.median(lateh) if len(lateh) else -1, lateh.max() if len(lateh) else -1))
    if rep == 5:
        # per-wave timeline (median over workgroups, cycles since WG entry)
        cols = [c for c in (2, 15, 3, 4, 8, 9, 10, 11) if full[:, :, c].max() != 0]
        print("  per-wave medians  " + " ".join("%7s" % ("s%d" % c) for c in cols) + " | phase lengths")
        for w in range(16):
            if not ok[:, w].any():
                continue
            med = [float(np.median(absd[:, w, c][ok[:, w] & (full[:, w, c] != 0)])) if (ok[:, w] & (full[:, w, c] != 0)).any() else float("nan") for c in cols]
            print("  wave %2d           " % w + " ".join("%7.0f" % m for m in med) + " | " + " ".join("%6.0f" % (med[i + 1] - med[i]) for i in range(len(med) - 1)))
    if rep == 5:
        e = full[:, 0, 0]
        print("entry stamps of wave 0, blocks 0..23:", [int(x) for x in e[:24]])
        print("end stamps  of wave 0, blocks 0..23:", [int(x) for x in full[:24, 0, 59]])
