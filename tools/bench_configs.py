#!/usr/bin/env python3
"""Kernel-resident throughput of the other BASELINE.json configs (parity-test cases, not the bench metric):
  3: mixed short/long stream(s), block pattern L L S S S S S S S S L with overlap-add state carry
  4: 5.1-channel 48 kHz long blocks with channel coupling
  5: many independent stereo streams, ONE packet per stream per launch (state round trip through HBM every launch)
  (6-15: design probes and other block sizes, see lewton_amd/workloads.py)
Same method as bench.py: records resident in HBM, hipGraph replay of rotated batches, HIP events -- and, like bench.py,
the PCM the timed launches left for batch 0 is compared with the oracle, every packet (`parity` of each line; --no-verify
skips it).  The oracle is the checker only; nothing timed touches it.
    python tools/bench_configs.py [--steps 400] [--only 3,4]"""
import argparse
import dataclasses
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lewton_amd import audio, header, workloads as wl  # noqa: E402
from lewton_amd.batch import Batch  # noqa: E402

NB = None           # batches rotated: None = by footprint (rotation_batches below)
MALL_BYTES = 256 << 20   # MI355X Infinity Cache (MI355X_MICROARCH.md): a rotation that fits it is timed out of the cache, not HBM


def rotation_batches(alg_bytes):
    """Batches to rotate so that a timed rotation touches >= 0.5 GiB of algorithmic bytes (twice the 256 MiB Infinity Cache,
    SURVEY 7 "Measuring HBM, not Infinity Cache"), and at least 8 of them while that stays below 1.5 GiB."""
    need = -(-(2 * MALL_BYTES) // max(1, alg_bytes))
    return max(2, need, min(8, (3 * (1 << 29)) // max(1, alg_bytes)))


def measure(w, steps=400, nb=NB, verify=True, force_generic=False, distinct=None, settle_ms=40.0, mix=None, branches=1, long10=None):
    """Time workload `w` (lewton_amd.workloads.Workload): `nb` rotated batches resident in HBM (default: as many as
    rotation_batches() asks for -- a footprint of at least 0.5 GiB), one hipGraph of `nb` steps replayed, HIP events; then
    (verify) every packet of timed batch 0 against the oracle.  Returns the result line as a dict.
    Batch b holds the same packet sequences as batch 0 with the stream -> sequence assignment rotated by b (its own
    records, its own PCM buffer: what defeats the cache is the footprint, not the values)."""
    if distinct:
        w = dataclasses.replace(w, distinct=distinct)
    setup = w.setup()
    idp, _, stp = setup.headers()
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    dec = audio.decoder_for(ident, st, torch.cuda.current_device())
    NP = w.n_streams * w.per_stream
    batches, outs, material = [], [], []
    seqs0 = wl.stream_material(w, setup, batch=0)
    b = 0
    while nb is None or b < nb:
        pw = [audio.PreviousWindowRight() for _ in range(w.n_streams)]
        # every stream is primed with the packet that precedes its first timed one, so all timed packets yield samples
        r_ = b % len(seqs0)
        seqs = seqs0[r_:] + seqs0[:r_]
        prime_items, items = wl.items_of(w, seqs, pw)
        prime = Batch(dec, w.n_streams, "i16")
        prime.entropy(prime_items, n_threads=0)
        prime.upload(None)
        prime.synth_to_host(None)
        prime.close()
        bt = Batch(dec, NP, "i16")
        if mix is not None:
            bt.debug_set_mix(mix)   # lw_debug_batch_set_mix: 0 = mixed batches as two launches, -1 = as one where k_mix applies
        if force_generic:
            bt.set_force_generic(True)
        if long10 is not None:
            bt.debug_set_long10(long10)   # lw_debug_batch_set_long10: 1 = k_long10 / k_long12 without their EDGE form, 0 = k_short<32> / k_big<12>
        bt.entropy(items, n_threads=0)
        bt.upload(None)
        outs.append(torch.empty(max(1, bt.out_elems), dtype=torch.int16, device="cuda"))
        batches.append((bt, pw))
        material.append(seqs)
        if nb is None:
            nb = rotation_batches(bt.algorithmic_bytes)
        b += 1
    torch.cuda.synchronize()
    alg = batches[0][0].algorithmic_bytes
    state = batches[0][0].state_bytes
    stream = torch.cuda.current_stream()

    def step(k, sp):
        bt = batches[k % nb][0]
        bt.synth(C.c_void_p(outs[k % nb].data_ptr()), outs[k % nb].numel(), sp)

    for k in range(2 * nb):
        step(k, C.c_void_p(stream.cuda_stream))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    if branches > 1:
        # experiment (profiles/r06_graph_branches.txt): the rotated batches are independent of each other, so the graph may run them
        # as `branches` parallel chains -- the ramp of one sparse launch can then overlap the tail of another.  NOT for the one-launch
        # mixed kernels (k_mix / k_mix10: two such grids at once can starve each other): pass mix=0 with it.
        sides = [torch.cuda.Stream() for _ in range(branches)]
        with torch.cuda.graph(g):
            main = torch.cuda.current_stream()
            fork = torch.cuda.Event()
            fork.record(main)
            for sd in sides:
                sd.wait_event(fork)
            for k in range(nb):
                sd = sides[k % branches]
                with torch.cuda.stream(sd):
                    step(k, C.c_void_p(sd.cuda_stream))
            for sd in sides:
                join = torch.cuda.Event()
                join.record(sd)
                main.wait_event(join)
    else:
        with torch.cuda.graph(g):
            cs = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for k in range(nb):
                step(k, cs)
    import time
    t_settle = time.perf_counter()   # untimed: tens of milliseconds of load until the device clocks have settled (as bench.py does)
    while True:
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        if (time.perf_counter() - t_settle) * 1e3 >= settle_ms:
            break
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(2, -(-steps // nb))
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * nb)
    parity = "unchecked"
    if verify:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from common import verify_workload_batch
            bad = verify_workload_batch(w, setup, material[0], batches[0][0].results(), outs[0].cpu().numpy(), "i16")
            parity = ("timed batch 0: %d packets i16 bit-exact vs oracle" % NP) if bad == 0 else "MISMATCH: %d of %d packets" % (bad, NP)
        except Exception as e:  # the oracle is only a checker here
            parity = "unchecked: %r" % (e,)
    res = {"config": w.name, "packets_per_launch": NP, "streams": w.n_streams, "steps": reps * nb, "us_per_launch": round(us, 2),
           "M_packets_per_s": round(NP / us, 2), "algorithmic_bytes_per_launch": alg,
           "batches_rotated": nb, "footprint_bytes": nb * alg,
           "pct_of_8TBps": round(100 * alg / (us * 1e-6) / 8e12, 2),
           # informational: the streams' window state has to cross HBM at a launch boundary (stored right parts in, new ones out);
           # SURVEY 8(d)'s figure does not count it, so shapes with few packets per stream and launch look slower than the chip runs
           "state_bytes_per_launch": state, "pct_of_8TBps_incl_state": round(100 * (alg + state) / (us * 1e-6) / 8e12, 2),
           "kernels": batches[0][0].last_kernels, "parity": parity, "graph_branches": branches,
           "note": w.note}
    for bt, _ in batches:
        bt.close()
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--packets", type=int, default=4096)
    ap.add_argument("--only", default="", help="comma-separated config numbers (default: 3,4,5)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--force-generic", action="store_true")
    ap.add_argument("--nb", type=int, default=0, help="batches rotated (0 = by footprint: >= 0.5 GiB per rotation)")
    ap.add_argument("--branches", type=int, default=1, help="experiment: the rotated batches as this many parallel chains of the graph (use --mix 0 with mixed shapes)")
    ap.add_argument("--long10", type=int, default=None, help="lw_debug_batch_set_long10: 1 = without the EDGE form (long blocks next to short ones on the generic kernels)")
    ap.add_argument("--mix", type=int, default=None, help="lw_debug_batch_set_mix: 0 = mixed batches as two launches, -1 = as one where k_mix applies (default)")
    args = ap.parse_args()
    ONLY = set(args.only.split(",")) if args.only else {"3", "4", "5"}
    for w in wl.configs(args.packets):
        if w.key in ONLY:
            print(json.dumps(measure(w, args.steps, args.nb or None, not args.no_verify, args.force_generic, mix=args.mix, branches=args.branches, long10=args.long10)), flush=True)
