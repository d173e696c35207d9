"""Shared helpers for the test-suite: stream fixtures and a numpy model of the floor record."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from lewton_amd import streamgen as sg  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

SETUPS = {
    "stereo": lambda: sg.stereo_setup(),
    "stereo_t1": lambda: sg.stereo_setup(residue_type=1),
    "surround51": lambda: sg.surround51_setup(),
    "mono_small": lambda: sg.mono_setup(),
    "stereo_9_12": lambda: sg.stereo_setup(bs0=9, bs1=12),
    "stereo_6_13": lambda: sg.stereo_setup(bs0=6, bs1=13),
    "stereo_7_7": lambda: sg.stereo_setup(bs0=7, bs1=7),
    # block sizes of k_short<L> in both roles: 512 / 1024 (what libvorbis writes at 16-22 kHz), 256 / 1024, 256 / 512, and
    # 1024-point SHORT blocks next to 4096-point long ones (generic)
    "stereo_9_10": lambda: sg.stereo_setup(22050, 9, 10),
    "stereo_8_10": lambda: sg.stereo_setup(22050, 8, 10, residue_type=1),
    "stereo_8_9": lambda: sg.stereo_setup(11025, 8, 9),
    "stereo_10_12": lambda: sg.stereo_setup(44100, 10, 12),
}
# host-stage tests only (no GPU test iterates these): a residue pass through a one-entry codebook
def spill_setup(residue_type):
    """stereo, partitions of 12 / 20 elements against books of 8, 4 and 2 dimensions: the last codeword of a partition
    reaches into the next one, and one that would reach past the end of the vector is read and dropped (audio.rs:600-612)"""
    st = sg.stereo_setup(44100, 8, 11, residue_type=residue_type)
    st.residues[0].partition_size = 12
    st.residues[1].partition_size = 20
    for rs, begin in zip(st.residues, (8, 4) if residue_type == 1 else (4, 8)):
        rs.begin = begin   # the last partition ends exactly at the end of the (clipped) vector: its last codeword is dropped
        rs.end = 1 << 20
    return st


def bookless_submap_setup():
    """5.1 whose FIRST submap's residues have no book in any class and pass: pass 0 still reads their class words
    (audio.rs:664-676), and the second submap's residue decodes from the bit position behind them"""
    st = sg.surround51_setup()
    for rs in st.residues[:2]:
        rs.books = [[-1] * 8 for _ in rs.books]
    return st


HOST_SETUPS = dict(SETUPS, stereo_single_entry=lambda: sg.stereo_setup(single_entry_book=True),
                   stereo_spill_t1=lambda: spill_setup(1), stereo_spill_t2=lambda: spill_setup(2),
                   surround51_bookless=bookless_submap_setup,
                   # more than 8 channels (7.1.4): the device entropy stage's channel map holds 16
                   multichannel12=lambda: sg.multichannel_setup(12))
# floor type 0: curve evaluated by the host stage, multiplied on the GPU
FLOOR0_SETUPS = {
    "floor0": lambda: sg.floor0_setup(),
    "floor0_mixed": lambda: sg.floor0_setup(mixed=True),
    "floor0_8_11": lambda: sg.floor0_setup(bs0=8, bs1=11, sample_rate=44100),
}


def f32_identical(a, b):
    """bit-identical f32 arrays, except that a NaN equals a NaN: sign and payload of a NaN an operation produces are not
    fixed by IEEE 754 (x86 makes 0xFFC00000, gfx950 0x7FC00000), so lewton itself differs between platforms there"""
    a, b = np.asarray(a, np.float32).reshape(-1), np.asarray(b, np.float32).reshape(-1)
    if a.size != b.size:
        return False
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def oracle_headers(setup):
    idp, cmt, stp = setup.headers()
    ident = po.Ident(idp)
    return ident, po.Setup(stp, ident)


def floor_x_sorted(setup, mode, channel):
    mp = setup.mappings[setup.modes[mode].mapping]
    fl = setup.floors[mp.submap_floor[mp.mux[channel]]]
    return sorted(fl.x_list)


def floor_from_record(rec, xs, n2, inv_db):
    """numpy model of the device floor rendering: record (u16 per ascending-x post) -> n2 floor values.
    Closed form of render_line; an unused floor gives zeros (audio.rs:1021-1024)."""
    if rec[0] == 0xFFFF:
        return np.zeros(n2, np.float32)
    F = len(xs)
    act = [(xs[i], int(rec[i] & 0xFF)) for i in range(F) if rec[i] & 0x8000]
    y = np.zeros(n2, np.int64)
    for (x0, y0), (x1, y1) in zip(act[:-1], act[1:]):
        if x0 >= n2:
            break
        k = np.arange(x0, min(x1, n2))
        dy, adx = y1 - y0, x1 - x0
        off = (abs(dy) * (k - x0)) // adx
        y[x0:min(x1, n2)] = y0 - off if dy < 0 else y0 + off
    lx, ly = act[-1]
    if lx < n2:
        y[lx:] = ly
    return inv_db[y].astype(np.float32)


def verify_workload_batch(w, setup, seqs, results, flat, fmt="i16"):
    """Every timed packet of a lewton_amd.workloads batch against the ORACLE: stream s = sequence s % len(seqs), primed
    with element 0, then per_stream consecutive packets (stream-major in `results` / `flat`).  Returns the number of
    packets that differ (0 = bit-exact); raises on a status or shape mismatch."""
    o_id, o_st = oracle_headers(setup)
    ch = setup.channels
    ofmt = {"i16": "i16", "f32": "f32", "i16_interleaved": "i16_itl"}[fmt]
    bad, k = 0, 0
    cache = {}
    for s in range(w.n_streams):
        q = s % len(seqs)
        if q not in cache:                      # streams that share a sequence share the expected output
            opw = po.Pwr()
            po.read_audio_packet(o_id, o_st, seqs[q][0], opw, ofmt)
            cache[q] = [np.asarray(po.read_audio_packet(o_id, o_st, p, opw, ofmt)).reshape(-1) for p in seqs[q][1:]]
        for want in cache[q]:
            status, m, off = results[k]
            assert status == 0 and m * ch == want.size, (s, k, status, m, want.size)
            got = np.asarray(flat[off:off + m * ch]).reshape(-1)
            if fmt == "f32":
                bad += not np.array_equal(got.view(np.uint32), want.view(np.uint32))
            else:
                bad += not np.array_equal(got, want)
            k += 1
    return bad
