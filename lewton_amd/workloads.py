"""The synthetic workloads BASELINE.json's `configs` name, as the bench tools lay them out (one definition shared by
tools/bench_configs.py, which times them, and tests/test_gpu_quoted_shapes.py, which checks every packet of the same
shapes against the oracle -- a number is only quoted for a shape that is also verified).

A workload = `n_streams` logical streams, each primed with ONE packet in an earlier launch (so that every timed packet
yields samples) and contributing `per_stream` consecutive packets to one dense launch, stream-major.

This module only builds inputs (streamgen); it contains no decoding logic."""
from dataclasses import dataclass
from typing import Callable, List

from . import streamgen as sg


def uncoupled_stereo():
    """the bench stream without its coupling step: one uncoupled channel pair per packet (k_long pairs up channels that
    are in no coupling step, lw_fast.cpp)"""
    st = sg.stereo_setup(44100, 8, 11, residue_type=1)
    for m in st.mappings:
        m.coupling = []
    return st


def mono():
    st = sg.stereo_setup(44100, 8, 11, residue_type=1)
    st.channels = 1
    st.mappings = [sg.Mapping([], [0], [0], [0]), sg.Mapping([], [0], [1], [1])]
    return st


def surround51_libvorbis_coupling(bs0: int = 8, bs1: int = 11):
    """5.1 @ 48 kHz with the four coupling steps libvorbis writes for six channels -- (0,2), (3,4), (0,1), (0,3): channel 0 takes
    part in three of them, channel 3 in two.  The disjoint prefix (0,2), (3,4) stays the units' own step, the two steps behind it
    (applied first) are evaluated inside k_long's waves (LwFastPlan::pre); other block sizes take the pre-pass k_prep"""
    st = sg.surround51_setup(48000, bs0, bs1)
    if bs1 == 10:
        st.floors[3].x_rest = [64, 16, 256, 128, 32, 384]   # (the generator's LFE floor repeats the implied end post at x = 512)
    for m in st.mappings:
        m.coupling = [(0, 2), (3, 4), (0, 1), (0, 3)]
    return st


def two_long_modes(bs0: int = 8, bs1: int = 11):
    """the bench stream with a second long mode that has its own mapping (another floor, no coupling): legal (header.rs:1060-1080),
    never written by an encoder -- the long blocks reach k_long behind the canonicalising pre-pass k_prep"""
    import numpy as np
    st = sg.stereo_setup(44100, bs0, bs1)
    st.floors.append(sg.random_floor1(np.random.default_rng(5), st.codebooks, bs1, posts=40))
    st.mappings.append(sg.Mapping([], [0, 0], [2], [1]))
    st.modes.append(sg.Mode(1, 2))
    return st


@dataclass
class Workload:
    key: str
    name: str
    setup: Callable[[], "sg.StreamSetup"]
    pattern: str
    n_streams: int
    per_stream: int
    note: str
    distinct: int = 64          # distinct packet sequences generated (stream s uses sequence s % distinct)


def configs(packets: int = 4096) -> List[Workload]:
    per = max(1, packets // 256)
    return [
        Workload("3", "3 mixed short/long", lambda: sg.stereo_setup(44100, 8, 11), "LLSSSSSSSSL", 256, per,
                 "256 streams x %d consecutive packets of the pattern; state carried inside the launch" % per),
        Workload("4", "4 5.1 @ 48 kHz long blocks", lambda: sg.surround51_setup(48000, 8, 11), "L", 256, per,
                 "6 channels = 3 units per packet (2 coupled pairs + 1 uncoupled pair)"),
        Workload("5", "5 independent streams, 1 packet per stream per launch", lambda: sg.stereo_setup(44100, 8, 11), "L",
                 packets, 1, "state read from and written to the HBM state pool by every packet"),
        # design probes for k_long (not BASELINE configs)
        Workload("6", "6 stereo long blocks WITHOUT coupling", uncoupled_stereo, "L", 256, per,
                 "one uncoupled channel pair per packet"),
        Workload("7", "7 mono long blocks", mono, "L", 256, per, "one single-channel unit per packet"),
        Workload("8", "8 mono long blocks, 2 x packets", mono, "L", 256, 2 * per, "single-channel units, 2 rounds"),
        Workload("9", "9 stereo long blocks, 2 x packets", lambda: sg.stereo_setup(44100, 8, 11), "L", 256, 2 * per,
                 "coupled pairs, 2 rounds per workgroup"),
        # BASELINE configs[4] on ONE GPU (the share of an 8-GPU job is 1250 streams; here all 10 000): 4 consecutive packets
        # of every stream per launch = 40 000 packets per launch, state through the HBM state pool between launches
        Workload("10", "5b 10 000 independent streams x 4 packets per launch", lambda: sg.stereo_setup(44100, 8, 11), "L",
                 10000, 4, "configs[4] stepping: 16 launches of this shape = 10 000 streams x 64 packets"),
        # other long block sizes (header.rs:236-247 allows 6..13)
        Workload("11", "stereo long blocks n = 4096", lambda: sg.stereo_setup(44100, 9, 12), "L", 256, per,
                 "blocksize_1 = 12"),
        Workload("12", "stereo long blocks n = 1024", lambda: sg.stereo_setup(44100, 8, 10), "L", 256, per,
                 "blocksize_1 = 10"),
        Workload("13", "stereo long blocks n = 8192", lambda: sg.stereo_setup(44100, 6, 13), "L", 256, per,
                 "blocksize_1 = 13"),
        # mixed short/long streams at the block sizes libvorbis writes at 16-22 kHz (audio.rs:1056-1073: the four window shapes of a
        # long block next to short ones)
        Workload("14", "mixed 512/1024", lambda: sg.stereo_setup(22050, 9, 10), "LLLSSSLLLL", 256, per,
                 "blocksize 9 / 10; long blocks with short slopes on either side of every run of short blocks"),
        Workload("15", "mixed 256/1024", lambda: sg.stereo_setup(22050, 8, 10), "LLLSSSLLLL", 256, per,
                 "blocksize 8 / 10"),
        # the block sizes libvorbis writes at 44.1 kHz for its lowest qualities (-1, 0: ~45-64 kbit/s): long blocks with short slopes in
        # k_long12's EDGE form
        Workload("20", "mixed 512/4096", lambda: sg.stereo_setup(44100, 9, 12), "LLLSSSLLLL", 256, per,
                 "blocksize 9 / 12; long blocks with short slopes on either side of every run of short blocks"),
        # a pair of block sizes without an edge form (no encoder writes it): the long blocks next to short ones keep their transform in
        # k_long12 and take the window / overlap-add from k_ola_generic (LW_RF_TDONLY)
        Workload("21", "mixed 1024/4096", lambda: sg.stereo_setup(44100, 10, 12), "LLLSSSLLLL", 256, per,
                 "blocksize 10 / 12; short blocks in k_short<32>, long blocks with short slopes: k_long12's time-domain block + k_ola_generic"),
        # round 6: a stream shape behind the canonicalising pre-pass; SURVEY 8(d) config 3 as written (ONE stream, state carried
        # through the whole launch: audio.rs:1082-1154, examples/perf.rs:35-44) and its all-long counterpart
        Workload("16", "5.1 @ 48 kHz long blocks, libvorbis' coupling steps (a channel in three steps)", surround51_libvorbis_coupling,
                 "L", 256, per, "the steps behind the disjoint prefix (0,2), (3,4) are evaluated inside k_long's waves"),
        Workload("19", "stereo long blocks of a stream with two long modes (own mappings)", two_long_modes, "L", 256, per,
                 "k_prep applies the coupling step and multiplies the channels' floors, k_long runs an uncoupled pair on the unit floor"),
        Workload("17", "3b ONE stream x %d consecutive packets, LLSSSSSSSSL" % packets, lambda: sg.stereo_setup(44100, 8, 11),
                 "LLSSSSSSSSL", 1, packets, "a single stream cut over the chip's workgroups: halo pre-pass at every chunk start", 1),
        Workload("18", "ONE stream x %d consecutive long packets" % packets, lambda: sg.stereo_setup(44100, 8, 11), "L", 1, packets,
                 "a single stream cut over the chip's workgroups: halo pre-pass at every chunk start", 1),
    ]


def by_key(key: str, packets: int = 4096) -> Workload:
    for w in configs(packets):
        if w.key == key:
            return w
    raise KeyError(key)


def stream_material(w: Workload, setup, batch: int = 0):
    """Packet sequences of one batch: `distinct` sequences of per_stream + 1 packets (element 0 primes the stream)."""
    n = min(w.n_streams, w.distinct)
    return [sg.make_stream(setup, w.pattern, w.per_stream + 1, seed=1000 * batch + s) for s in range(n)]


def items_of(w: Workload, seqs, pwrs):
    """(prime items, timed items): lists of (packet bytes, pwr) in submission order."""
    prime = [(seqs[s % len(seqs)][0], pwrs[s]) for s in range(w.n_streams)]
    timed = [(seqs[s % len(seqs)][1 + k], pwrs[s]) for s in range(w.n_streams) for k in range(w.per_stream)]
    return prime, timed
