#!/usr/bin/env python3
"""(GPU box) Differential campaign of the short / long block kernels (k_short, k_long, k_long<EDGE>) at scale: random streams
of random block patterns -- runs of short blocks of every length, long blocks with every slope combination, including the
flag combinations an encoder never writes (a long block that announces a long neighbour next to a short one), damaged and
truncated packets, unused floors -- cut into batches at random places (window state through the state pool), streams
interleaved at random, through the product's default path AND through its generic kernels (lw_batch_set_force_generic) AND
through the oracle.  Every status, sample count, PCM sample and final state must be identical three ways.
    python tools/fuzz_gpu_mixed.py [--rounds 40] [--seed 1]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lewton_amd import audio, header, streamgen as sg, workloads as wl  # noqa: E402
from lewton_amd.batch import Batch  # noqa: E402
from oracle import pyoracle as po  # noqa: E402  (the checker)

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--big", action="store_true", help="streams with 4096 / 8192-point long blocks (k_big) instead of the 2048-point ones")
ap.add_argument("--mid", action="store_true", help="streams with 1024- / 512-point long blocks (k_long10 incl. its EDGE form, k_short<16 / 32>)")
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
SETUPS = {"stereo": lambda: sg.stereo_setup(44100, 8, 11), "stereo_t1": lambda: sg.stereo_setup(44100, 8, 11, residue_type=1),
          "surround51": lambda: sg.surround51_setup(48000, 8, 11), "mono": wl.mono, "uncoupled": wl.uncoupled_stereo}
if args.big:
    SETUPS = {"stereo_9_12": lambda: sg.stereo_setup(44100, 9, 12), "stereo_6_13_t1": lambda: sg.stereo_setup(44100, 6, 13, residue_type=1),
              "surround51_9_12": lambda: sg.surround51_setup(48000, 9, 12), "mono_7_12": lambda: sg.mono_setup(7, 12, 44100),
              "stereo_10_12": lambda: sg.stereo_setup(44100, 10, 12), "stereo_8_13": lambda: sg.stereo_setup(44100, 8, 13)}
if args.mid:
    def _s51():
        st = sg.surround51_setup(48000, 8, 10)
        st.floors[3].x_rest = [64, 16, 256, 128, 32, 384]   # (the generator's LFE floor repeats the implied end post at x = 512 for bs 10)
        return st

    def _unc():
        st = sg.stereo_setup(22050, 8, 10, residue_type=1)
        for m in st.mappings:
            m.coupling = []
        return st
    SETUPS = {"stereo_9_10": lambda: sg.stereo_setup(22050, 9, 10), "stereo_8_10_t1": lambda: sg.stereo_setup(22050, 8, 10, residue_type=1),
              "surround51_8_10": _s51, "mono_7_10": lambda: sg.mono_setup(7, 10, 16000), "uncoupled_8_10": _unc,
              "stereo_8_9": lambda: sg.stereo_setup(11025, 8, 9)}
FMTS = ["i16", "f32", "i16_interleaved"]
OFMT = {"i16": "i16", "f32": "f32", "i16_interleaved": "i16_itl"}
made = {}


def product(name):
    if name not in made:
        setup = SETUPS[name]()
        idp, _, stp = setup.headers()
        ident = header.read_header_ident(idp)
        st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
        o_id = po.Ident(idp)
        made[name] = (setup, audio.decoder_for(ident, st, 0), o_id, po.Setup(stp, o_id))
    return made[name]


def random_stream(setup, n, seed):
    """n packets of a random L/S sequence with CONSISTENT window flags most of the time, random flags sometimes"""
    pw = sg.PacketWriter(setup, seed, p_floor_unused=float(rng.choice([0.0, 0.05, 0.3])))
    short_mode = next(i for i, m in enumerate(setup.modes) if not m.blockflag)
    long_mode = next(i for i, m in enumerate(setup.modes) if m.blockflag)
    seq, k = [], 0
    while k < n:
        run = int(rng.choice([1, 1, 2, 3, 5, 8, 9, 17]))
        seq += [("S" if rng.random() < 0.5 else "L")] * run
        k += run
    seq = seq[:n]
    out = []
    for i, b in enumerate(seq):
        if b == "S":
            out.append(pw.packet(short_mode))
            continue
        pf = 1 if (i == 0 or seq[i - 1] == "L") else 0
        nf = 1 if (i + 1 >= n or seq[i + 1] == "L") else 0
        if rng.random() < 0.04:   # flags that contradict the neighbours (legal bitstream, odd windows: generic kernels)
            pf, nf = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        out.append(pw.packet(long_mode, pf, nf))
    for i in range(n):
        r = rng.random()
        if r < 0.02 and len(out[i]) > 2:
            out[i] = out[i][: int(rng.integers(1, len(out[i])))]
        elif r < 0.04:
            p = bytearray(out[i])
            p[int(rng.integers(0, len(p)))] ^= 1 << int(rng.integers(0, 8))
            out[i] = bytes(p)
    return out


total, kernels_seen = 0, set()
for rnd in range(args.rounds):
    name = list(SETUPS)[rnd % len(SETUPS)]
    fmt = FMTS[(rnd // len(SETUPS)) % 3]
    setup, dec, o_id, o_st = product(name)
    ch = setup.channels
    n_streams = int(rng.choice([1, 3, 17, 64]))
    length = int(rng.choice([12, 40, 90]))
    streams = [random_stream(setup, length, 1000 * rnd + s) for s in range(n_streams)]
    # submission order: stream-major chunks or round-robin, cut into batches at random places
    order = []
    if rng.random() < 0.5:
        chunk = int(rng.choice([1, 4, 16, length]))
        for c0 in range(0, length, chunk):
            for s in range(n_streams):
                order += [(s, t) for t in range(c0, min(length, c0 + chunk))]
    else:
        order = [(s, t) for t in range(length) for s in range(n_streams)]
    cuts = sorted(set([0, len(order)] + [int(x) for x in rng.integers(1, len(order), int(rng.integers(0, 6)))]))
    pw_a = [audio.PreviousWindowRight() for _ in range(n_streams)]
    pw_g = [audio.PreviousWindowRight() for _ in range(n_streams)]
    opw = [po.Pwr() for _ in range(n_streams)]
    cap = max(b - a for a, b in zip(cuts[:-1], cuts[1:]))
    ba, bg = Batch(dec, cap, fmt), Batch(dec, cap, fmt)
    bg.set_force_generic(True)
    for a, b in zip(cuts[:-1], cuts[1:]):
        items = order[a:b]
        ra = ba.entropy([(streams[s][t], pw_a[s]) for s, t in items], n_threads=2)
        ba.upload()
        fa = ba.split(ba.synth_to_host(), ch)
        rg = bg.entropy([(streams[s][t], pw_g[s]) for s, t in items], n_threads=2)
        bg.upload()
        fg = bg.split(bg.synth_to_host(), ch)
        kernels_seen.update(ba.last_kernels.split(","))
        for i, (s, t) in enumerate(items):
            try:
                want = np.asarray(po.read_audio_packet(o_id, o_st, streams[s][t], opw[s], OFMT[fmt]))
                rc = 0
            except po.OracleError as e:
                rc = e.code
            if ra[i][0] != rc or rg[i][0] != rc:
                print("STATUS MISMATCH round %d setup %s stream %d packet %d: oracle %d default %d generic %d" % (rnd, name, s, t, rc, ra[i][0], rg[i][0]))
                sys.exit(1)
            if rc:
                continue
            for which, got in (("default", fa[i]), ("generic", fg[i])):
                same = got.size == want.size and (np.array_equal(got.reshape(-1).view(np.uint32), want.reshape(-1).view(np.uint32))
                                                  if fmt == "f32" else np.array_equal(got.reshape(-1), want.reshape(-1)))
                if not same:
                    print("PCM MISMATCH (%s path) round %d setup %s fmt %s stream %d packet %d kernels %s" % (
                        which, rnd, name, fmt, s, t, ba.last_kernels))
                    sys.exit(1)
            total += 1
    for s in range(n_streams):
        o, a, g = opw[s].data(ch), pw_a[s].data(), pw_g[s].data()     # None: no stored right part (the stream's first packet failed ...)
        same = (o is None) == (a is None) == (g is None) if (o is None or a is None or g is None) else (
            np.array_equal(a.view(np.uint32), o.view(np.uint32)) and np.array_equal(g.view(np.uint32), o.view(np.uint32)))
        if not same:
            print("STATE MISMATCH round %d setup %s stream %d" % (rnd, name, s))
            sys.exit(1)
    ba.close()
    bg.close()
print("fuzz_gpu_mixed: %d rounds, %d packets identical three ways (default path, generic kernels, oracle); kernels seen: %s" % (
    args.rounds, total, ",".join(sorted(kernels_seen))))
