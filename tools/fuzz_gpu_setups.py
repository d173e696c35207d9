#!/usr/bin/env python3
"""(GPU box) Differential campaign over the space of stream SETUPS (round 6).  Every earlier campaign varied packets of ~20
hand-built setups; this one draws the setup header itself (streamgen.random_setup: channels 1-8, block sizes 6..13, 1-6 modes
with their own mappings, 1-3 submaps, any coupling list, floor 1 with 2..65 posts and any class / subclass structure, floor 0
mixed in, residue types 0/1/2 with begin / end / partition sizes of every kind, books of lookup types 1 and 2 incl. sequence_p,
sparse and ordered length lists, one-entry books) and, per setup, decodes random streams (runs of short and long blocks,
contradicting window flags, damaged packets, unused floors) cut into batches at random places

    through the product's DEFAULT path  (whatever kernels the planner picks),
    through its GENERIC kernels         (lw_batch_set_force_generic),
    where eligible through K_ENTROPY    (entropy stage on the device) + the default synthesis path,
    and through the ORACLE.

Every status, sample count, PCM sample (i16 / f32 bit patterns / interleaved i16, rotating) and final window state must be
identical.  Prints, per setup, the kernels that ran and the planner's census line (lw_debug_plan_census), and at the end the
census table: share of setups per kernel / per reason.

    python tools/fuzz_gpu_setups.py [--setups 2000] [--seed 0] [--packets 250] [--procs 12] [--quiet]"""
import argparse
import collections
import ctypes as C
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from common import f32_identical  # noqa: E402

FMTS = ["i16", "f32", "i16_interleaved"]
OFMT = {"i16": "i16", "f32": "f32", "i16_interleaved": "i16_itl"}
DISTINCT = 3          # distinct packet sequences generated per setup (stream s carries sequence s % DISTINCT)


def shape_of(seed, packets):
    """streams x length of the setup with this seed (deterministic, shared by the generator processes and the checker)"""
    r = np.random.default_rng(seed ^ 0x5EED)
    length = int(r.choice([10, 24, 40]))
    n_streams = max(1, min(64, packets // length))
    return n_streams, length


def generate(job):
    """(worker process, CPU only) the header packets and DISTINCT packet sequences of setup `seed`"""
    seed, packets, sizes = job
    from lewton_amd import streamgen as sg
    rng = np.random.default_rng(seed)
    setup = sg.random_setup(rng, blocksizes=sizes[seed % len(sizes)] if sizes else None)
    _n_streams, length = shape_of(seed, packets)
    seqs = [sg.random_stream(setup, rng, length, seed=1000 * seed + q, p_floor_unused=float(rng.choice([0.0, 0.05, 0.3])),
                             p_damage=0.04) for q in range(DISTINCT)]
    idp, _cmt, stp = setup.headers()
    return seed, setup.channels, idp, stp, seqs


def census_line(N, ident, st):
    buf = C.create_string_buffer(2048)
    N.lib.lw_debug_plan_census(ident._h, st._h, buf, 2048)
    return buf.value.decode()


def run_setup(seed, ch, idp, stp, seqs, packets, rng, mods, length=None):
    """returns (packets checked, kernels seen, census line, k_entropy eligible); raises SystemExit on the first difference.
    `length`: packets per sequence when the caller made `seqs` itself (default: shape_of(seed, packets))"""
    audio, header, Batch, po, N = mods
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    o_id = po.Ident(idp)
    o_st = po.Setup(stp, o_id)
    line = census_line(N, ident, st)
    dec = audio.Decoder(ident, st, 0)
    fmt = FMTS[seed % 3]
    if length is None:
        n_streams, length = shape_of(seed, packets)
    else:
        n_streams = max(1, min(64, packets // length))
    # the oracle once per distinct sequence
    want = []
    for q in range(DISTINCT):
        opw = po.Pwr()
        rows = []
        for p in seqs[q]:
            try:
                rows.append((0, np.asarray(po.read_audio_packet(o_id, o_st, p, opw, OFMT[fmt]))))
            except po.OracleError as e:
                rows.append((e.code, None))
        want.append((rows, opw.data(ch)))
    order = []
    if rng.random() < 0.5:
        chunk = int(rng.choice([1, 4, 16, length]))
        for c0 in range(0, length, chunk):
            for s in range(n_streams):
                order += [(s, t) for t in range(c0, min(length, c0 + chunk))]
    else:
        order = [(s, t) for t in range(length) for s in range(n_streams)]
    cuts = sorted(set([0, len(order)] + [int(x) for x in rng.integers(1, max(2, len(order)), int(rng.integers(0, 5)))]))
    cap = max(b - a for a, b in zip(cuts[:-1], cuts[1:]))
    paths = [("default", Batch(dec, cap, fmt)), ("generic", Batch(dec, cap, fmt))]
    paths[1][1].set_force_generic(True)
    be = Batch(dec, cap, fmt)
    if be.set_entropy_on_device(True):
        paths.append(("k_entropy", be))
    else:
        be.close()
    pws = {name: [audio.PreviousWindowRight() for _ in range(n_streams)] for name, _b in paths}
    kernels, checked = set(), 0
    try:
        for a, b in zip(cuts[:-1], cuts[1:]):
            items = order[a:b]
            got = {}
            for name, bt in paths:
                res = bt.entropy([(seqs[s % DISTINCT][t], pws[name][s]) for s, t in items], n_threads=2)
                bt.upload()
                got[name] = (res, bt.split(bt.synth_to_host(), ch))
                if name != "generic":
                    kernels.update(k for k in bt.last_kernels.split(",") if k)
            for i, (s, t) in enumerate(items):
                rc, w = want[s % DISTINCT][0][t]
                for name, _bt in paths:
                    res, pcm = got[name]
                    if res[i][0] != rc:
                        print("STATUS MISMATCH setup %d (%s path) stream %d packet %d: oracle %d product %d\n  %s" % (
                            seed, name, s, t, rc, res[i][0], line))
                        raise SystemExit(1)
                    if rc:
                        continue
                    g = pcm[i]
                    same = g.size == w.size and (f32_identical(g, w) if fmt == "f32" else np.array_equal(g.reshape(-1), w.reshape(-1)))
                    if not same:
                        bad = -1
                        if g.size == w.size:
                            bad = int(np.flatnonzero(g.reshape(-1) != w.reshape(-1))[0]) if fmt != "f32" else int(
                                np.flatnonzero(g.reshape(-1).view(np.uint32) != w.reshape(-1).view(np.uint32))[0])
                        print("PCM MISMATCH setup %d (%s path) fmt %s stream %d packet %d (first differing element %d of %d; kernels %s)\n  %s" % (
                            seed, name, fmt, s, t, bad, w.size, paths[0][1].last_kernels, line))
                        raise SystemExit(1)
                if rc == 0:
                    checked += 1
        for name, _bt in paths:
            for s in range(n_streams):
                o, g = want[s % DISTINCT][1], pws[name][s].data()
                same = (o is None) == (g is None) and (o is None or f32_identical(g, o))
                if not same:
                    print("STATE MISMATCH setup %d (%s path) stream %d\n  %s" % (seed, name, s, line))
                    raise SystemExit(1)
    finally:
        for name, bt in paths:
            pws[name] = None
            bt.close()
        dec.close()
    return checked, kernels, line, len(paths) == 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--setups", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=0, help="first setup seed")
    ap.add_argument("--packets", type=int, default=250, help="packets per setup (streams x length)")
    ap.add_argument("--procs", type=int, default=12, help="generator processes (CPU only)")
    ap.add_argument("--quiet", action="store_true", help="no line per setup")
    ap.add_argument("--blocksizes", default="", help="pin the block sizes: e.g. 9:12,8:12 (setup `seed` takes entry seed %% count); default: streamgen's menu")
    args = ap.parse_args()
    sizes = [tuple(int(v) for v in e.split(":")) for e in args.blocksizes.split(",") if e]
    jobs = [(s, args.packets, sizes) for s in range(args.seed, args.seed + args.setups)]
    ctx = mp.get_context("fork")
    pool = ctx.Pool(args.procs)                      # forked BEFORE this process touches HIP
    it = pool.imap(generate, jobs, chunksize=2)
    from lewton_amd import _native as N
    from lewton_amd import audio, header
    from lewton_amd.batch import Batch
    from oracle import pyoracle as po              # the checker
    mods = (audio, header, Batch, po, N)
    rng = np.random.default_rng(args.seed + 77)
    t0 = time.time()
    total = n_dev = 0
    by_field = {k: collections.Counter() for k in ("long", "short", "transitions", "entropy")}
    by_kernel = collections.Counter()
    for seed, ch, idp, stp, seqs in it:
        try:
            checked, kernels, line, dev = run_setup(seed, ch, idp, stp, seqs, args.packets, rng, mods)
        except RuntimeError as e:      # a library call failed: name the setup
            print("ERROR setup %d: %s" % (seed, e), flush=True)
            raise SystemExit(1)
        n_streams, length = shape_of(seed, args.packets)
        total += checked
        n_dev += dev
        for part in line.split(" | "):
            k, v = part.split("=", 1)
            by_field[k][v] += 1
        for k in kernels:
            by_kernel[k] += 1
        if not args.quiet:
            print("setup %5d ok: %d ch, %3d streams x %2d, %5d packets | ran %s | %s" % (
                seed, ch, n_streams, length, checked, ",".join(sorted(kernels)), line), flush=True)
    pool.close()
    n = args.setups
    print("fuzz_gpu_setups: %d random setups (seeds %d..%d), %d packets identical across default path, generic kernels%s and oracle; "
          "%.0f s" % (n, args.seed, args.seed + n - 1, total, ", k_entropy (%d setups eligible)" % n_dev, time.time() - t0))
    for field in ("long", "short", "transitions", "entropy"):
        print("census, %s blocks:" % field if field in ("long", "short") else "census, %s:" % field)
        for k, v in by_field[field].most_common():
            print("  %5.1f %%  %s" % (100.0 * v / n, k))
    print("kernels that ran on the default path (share of setups):")
    for k, v in by_kernel.most_common():
        print("  %5.1f %%  %s" % (100.0 * v / n, k))


if __name__ == "__main__":
    main()
