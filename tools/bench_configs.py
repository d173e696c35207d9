#!/usr/bin/env python3
"""Kernel-resident throughput of the other BASELINE.json configs (parity-test cases, not the bench metric):
  3: mixed short/long stream(s), block pattern L L S S S S S S S S L with overlap-add state carry
  4: 5.1-channel 48 kHz long blocks with channel coupling
  5: many independent stereo streams, ONE packet per stream per launch (state round trip through HBM every launch)
Same method as bench.py: records resident in HBM, hipGraph replay of rotated batches, HIP events.
    python tools/bench_configs.py [--steps 400]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lewton_amd import audio, header, streamgen as sg  # noqa: E402
from lewton_amd.batch import Batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--packets", type=int, default=4096)
ap.add_argument("--only", default="", help="comma-separated config numbers (default: 3,4,5)")
args = ap.parse_args()
ONLY = set(args.only.split(",")) if args.only else {"3", "4", "5"}
NB = 4


def run(name, setup, pattern, n_streams, per_stream, note):
    idp, _, stp = setup.headers()
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    dec = audio.decoder_for(ident, st, 0)
    NP = n_streams * per_stream
    rng = np.random.default_rng(7)
    batches, outs = [], []
    for b in range(NB):
        pw = [audio.PreviousWindowRight() for _ in range(n_streams)]
        # every stream is primed with the packet that precedes its first timed one, so all timed packets yield samples
        streams = [sg.make_stream(setup, pattern, per_stream + 1, seed=1000 * b + s) for s in range(min(n_streams, 64))]
        prime = Batch(dec, n_streams, "i16")
        prime.entropy([(streams[s % len(streams)][0], pw[s]) for s in range(n_streams)], n_threads=0)
        prime.upload(None)
        prime.synth_to_host(None)
        prime.close()
        bt = Batch(dec, NP, "i16")
        items = []
        for s in range(n_streams):
            for k in range(per_stream):
                items.append((streams[s % len(streams)][1 + k], pw[s]))
        bt.entropy(items, n_threads=0)
        bt.upload(None)
        outs.append(torch.empty(max(1, bt.out_elems), dtype=torch.int16, device="cuda"))
        batches.append((bt, pw))
    torch.cuda.synchronize()
    alg = batches[0][0].algorithmic_bytes
    stream = torch.cuda.current_stream()

    def step(k, sp):
        bt = batches[k % NB][0]
        bt.synth(C.c_void_p(outs[k % NB].data_ptr()), outs[k % NB].numel(), sp)

    for k in range(8):
        step(k, C.c_void_p(stream.cuda_stream))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cs = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for k in range(NB):
            step(k, cs)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps // NB):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (args.steps // NB * NB)
    res = {"config": name, "packets_per_launch": NP, "streams": n_streams, "us_per_launch": round(us, 2),
           "M_packets_per_s": round(NP / us, 2), "algorithmic_bytes_per_launch": alg,
           "pct_of_8TBps": round(100 * alg / (us * 1e-6) / 8e12, 2), "kernels": batches[0][0].last_kernels, "note": note}
    print(json.dumps(res))
    for bt, _ in batches:
        bt.close()

def uncoupled_stereo():
    """the bench stream without its coupling step: two single-channel units per packet instead of one coupled pair"""
    st = sg.stereo_setup(44100, 8, 11, residue_type=1)
    for m in st.mappings:
        m.coupling = []
    return st


def mono():
    st = sg.stereo_setup(44100, 8, 11, residue_type=1)
    st.channels = 1
    st.mappings = [sg.Mapping([], [0], [0], [0]), sg.Mapping([], [0], [1], [1])]
    return st


if "3" in ONLY:
    run("3 mixed short/long", sg.stereo_setup(44100, 8, 11), "LLSSSSSSSSL", 256, args.packets // 256,
        "256 streams x 16 consecutive packets of the pattern; state carried inside the launch")
if "4" in ONLY:
    run("4 5.1 @ 48 kHz long blocks", sg.surround51_setup(48000, 8, 11), "L", 256, args.packets // 256,
        "6 channels = 4 units per packet (2 coupled pairs + 2 single channels)")
if "5" in ONLY:
    run("5 independent streams, 1 packet per stream per launch", sg.stereo_setup(44100, 8, 11), "L", args.packets, 1,
        "state read from and written to the HBM state pool by every packet")
# design probes for k_long (not BASELINE configs): what single-channel waves cost
if "6" in ONLY:
    run("6 stereo long blocks WITHOUT coupling", uncoupled_stereo(), "L", 256, args.packets // 256,
        "two single-channel units per packet: 8 packets per round, 2 rounds per workgroup")
if "7" in ONLY:
    run("7 mono long blocks", mono(), "L", 256, args.packets // 256, "one single-channel unit per packet, 1 round")
if "8" in ONLY:
    run("8 mono long blocks, 2 x packets", mono(), "L", 256, 2 * args.packets // 256, "single-channel units, 2 rounds")
if "9" in ONLY:
    run("9 stereo long blocks, 2 x packets", sg.stereo_setup(44100, 8, 11), "L", 256, 2 * args.packets // 256,
        "coupled pairs, 2 rounds per workgroup")
if "10" in ONLY:
    # BASELINE configs[4] on ONE GPU (the share of an 8-GPU job is 1250 streams; here all 10 000): 4 consecutive packets of
    # every stream per launch = 40 000 packets per launch, state through the HBM state pool between launches
    run("5b 10 000 independent streams x 4 packets per launch", sg.stereo_setup(44100, 8, 11), "L", 10000, 4,
        "configs[4] stepping: 16 launches of this shape = 10 000 streams x 64 packets")
