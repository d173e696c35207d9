#!/usr/bin/env python3
"""End-to-end throughput of the decode path on one GPU: host entropy stage (C++, multi-threaded, writing into pinned
staging) -> hipMemcpyAsync H2D -> synthesis kernels -> D2H of the i16 PCM, with two batches in flight so that the entropy
decode of batch k+1 overlaps the GPU work of batch k (the north-star's staging ring, two slots deep).

Not the BASELINE metric (that one is kernel-resident, bench.py): this number is bounded by host cores and PCIe.
    python tools/e2e.py [--batches 24] [--threads 0]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lewton_amd import audio, header, streamgen as sg  # noqa: E402
from lewton_amd.batch import Batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, default=24)
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--packets", type=int, default=4096)
ap.add_argument("--streams", type=int, default=256)
ap.add_argument("--device-vq", action="store_true", help="Tier B: ship codeword symbols, inverse VQ in k_residue_vq")
ap.add_argument("--callers", type=int, default=1, choices=[1, 2],
                help="2: two host threads, each with its own batch object and its own 256 streams; the worker pool runs one "
                     "parallel region at a time, so one caller's sequential planning overlaps the other's threaded decode")
args = ap.parse_args()

setup = sg.stereo_setup(44100, 8, 11)
idp, _, stp = setup.headers()
ident = header.read_header_ident(idp)
st = header.read_header_setup(stp, 2, (8, 11))
dec = audio.decoder_for(ident, st, 0)
pool = sg.make_stream(setup, "L", 512, seed=9)
rng = np.random.default_rng(1)
S, NP = args.streams, args.packets
per = NP // S
payload = None
callers = []
for c in range(args.callers):
    pwrs = [audio.PreviousWindowRight() for _ in range(S)]      # every caller decodes its own streams
    work = []
    for b in range(args.batches):
        order = rng.integers(0, 512, NP)
        work.append([(pool[int(i)], pwrs[k // per]) for k, i in enumerate(order)])
    if payload is None:
        payload = sum(len(p) for p, _ in work[0])
    slots = []
    for i in range(2):
        stream = torch.cuda.Stream()
        bt = Batch(dec, NP, "i16")
        if args.device_vq:
            assert bt.set_residue_on_device(True)
        d_out = torch.empty(NP * 2 * 1024, dtype=torch.int16, device="cuda")
        h_out = torch.empty(NP * 2 * 1024, dtype=torch.int16).pin_memory()
        slots.append((stream, bt, d_out, h_out, torch.cuda.Event()))
    # the lw_packet arrays are built once: the timed loop measures the C ABI (lw_batch_entropy / upload / synth), not ctypes
    marshalled = [slots[k % len(slots)][1].marshal(w) for k, w in enumerate(work)]
    callers.append((slots, marshalled, pwrs))


def run_caller(c, n, t_ent):
    slots, marshalled, _ = callers[c]
    kernels_done = None                       # consecutive batches carry the streams' window state on the device
    for k in range(n):
        stream, bt, d_out, h_out, ev = slots[k % len(slots)]
        ev.synchronize()                      # the slot's previous batch has left the staging buffers
        t0 = time.perf_counter()
        bt.entropy_marshalled(marshalled[k % len(marshalled)], n_threads=args.threads)
        t_ent[c] += time.perf_counter() - t0
        sp = C.c_void_p(stream.cuda_stream)
        bt.upload(sp)
        if kernels_done is not None:
            stream.wait_event(kernels_done)   # kernels of batch k read the state batch k-1's kernels wrote (other HIP stream)
        bt.synth(C.c_void_p(d_out.data_ptr()), d_out.numel(), sp)
        kernels_done = torch.cuda.Event()
        kernels_done.record(stream)
        with torch.cuda.stream(stream):
            h_out[: bt.out_elems].copy_(d_out[: bt.out_elems], non_blocking=True)
            ev.record(stream)


def run(n):
    """n batches in total; with one caller two slots alternate (entropy of batch k+1 overlaps the GPU work of batch k), with
    two callers each thread does the same on its own two slots (ctypes releases the GIL inside the library calls)."""
    t_ent = [0.0] * args.callers
    if args.callers == 1:
        run_caller(0, n, t_ent)
    else:
        import threading
        ts = [threading.Thread(target=run_caller, args=(c, n // 2, t_ent)) for c in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    torch.cuda.synchronize()
    return max(t_ent) if args.callers == 1 else sum(t_ent) / 2


run(4)
t0 = time.perf_counter()
t_ent = run(args.batches)
dt = time.perf_counter() - t0
npk = (args.batches // args.callers) * args.callers * NP
h2d = None
try:
    from lewton_amd import _native as N
    h2d = "symbols" if args.device_vq else "f32 residues"
except Exception:
    pass
print("mode: %s; callers: %d; kernels: %s" % (h2d, args.callers, callers[0][0][0][1].last_kernels))
print("end-to-end: %d packets in %.3f s -> %.2f M packets/s (%.1f MB/s of Vorbis payload, %.2f GB/s of H2D records, "
      "%.2f GB/s of D2H PCM); host entropy stage alone %.2f M packets/s on %s threads" % (
          npk, dt, npk / dt / 1e6, payload * args.batches / dt / 1e6, npk * 8324 / dt / 1e9, npk * 4096 / dt / 1e9,
          npk / t_ent / 1e6, args.threads or os.cpu_count()))
