"""The canonicalising pre-pass k_prep (round 6; LwPrepPlan in csrc/lw_fast.hpp): stream shapes the specialised kernels do not take
as they are -- a channel in several coupling steps (libvorbis' own 5.1 and 3-channel mappings), modes of one block size with
different mappings, a floor of 65 posts, three floor configurations in one mapping, floor 0 -- must leave the generic kernels for
k_prep + the specialised kernel of their block size and stay bit-exact against the oracle and against the generic kernels
(/root/reference/src/audio.rs:990-1002 any coupling list, :926-938 any mode, :1006-1039 floor x residue)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from common import ROOT, po, sg

sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_gpu_setups as fz  # noqa: E402


from lewton_amd.workloads import surround51_libvorbis_coupling, two_long_modes  # noqa: E402


def three_channels(bs0=8, bs1=11):
    """L / C / R: coupling steps (0,2), (0,1)"""
    st = sg.stereo_setup(44100, bs0, bs1, residue_type=1)
    st.channels = 3
    st.mappings = [sg.Mapping([(0, 2), (0, 1)], [0, 0, 0], [0], [0]), sg.Mapping([(0, 2), (0, 1)], [0, 0, 0], [1], [1])]
    return st


def floor_of_65_posts(bs0=8, bs1=11):
    st = sg.stereo_setup(44100, bs0, bs1)
    st.floors[1] = sg.random_floor1(np.random.default_rng(6), st.codebooks, bs1, posts=65)
    return st


def three_floor_configurations(bs0=8, bs1=11):
    """4 channels in three submaps, each with its own floor: more configurations than the kernels stage"""
    st = sg.stereo_setup(44100, bs0, bs1, residue_type=1)
    rng = np.random.default_rng(7)
    st.channels = 4
    st.floors += [sg.random_floor1(rng, st.codebooks, bs1, posts=12), sg.random_floor1(rng, st.codebooks, bs1, posts=31)]
    st.mappings = [sg.Mapping([(0, 1)], [0, 0, 0, 0], [0], [0]), sg.Mapping([(0, 1)], [0, 0, 1, 2], [1, 2, 3], [1, 1, 1])]
    return st


def floor_posts_beyond_the_block(bs0=8, bs1=11):
    """range bits 15 (header.rs:871-873): posts at x = 5000 ... 32768 whatever the block size.  A line from a post inside the block to
    one of those spans more bins than the kernels' two-FMA form of render_line is exact for (adx <= 4096): the random-setup campaign
    of round 6 met a line of 32 638 bins whose first bin came out one step off in k_long.  Such floors are evaluated by k_prep."""
    st = sg.stereo_setup(44100, bs0, bs1)
    f = st.floors[1]
    f.rangebits = 15
    f.x_rest = list(f.x_rest[:-4]) + [5000, 20000, 31880, 131]
    assert len(set(f.x_list)) == len(f.x_list)
    return st


CASES = {
    "surround51_libvorbis_coupling": (surround51_libvorbis_coupling, "a channel takes part in more than one coupling step"),
    "surround51_libvorbis_coupling_9_12": (lambda: surround51_libvorbis_coupling(9, 12), "a channel takes part"),
    "surround51_libvorbis_coupling_8_10": (lambda: surround51_libvorbis_coupling(8, 10), "a channel takes part"),
    "surround51_libvorbis_coupling_8_13": (lambda: surround51_libvorbis_coupling(8, 13), "a channel takes part"),
    "three_channels": (three_channels, "a channel takes part"),
    "two_long_modes": (two_long_modes, "long modes with different"),
    "floor_of_65_posts": (floor_of_65_posts, "more posts"),
    "floor_posts_beyond_the_block": (floor_posts_beyond_the_block, "posts beyond x = 4096"),
    "three_floor_configurations": (three_floor_configurations, "floor"),
    "floor0_long_blocks": (lambda: sg.floor0_setup(8, 11, 44100, mixed=True), "floor type 0"),
    # blocksize_0 = blocksize_1 with a flagged and an unflagged mode: one block shape, every mode planned as the long class; the two
    # modes have their own floors, so the class goes through k_prep
    "equal_sizes_8": (lambda: sg.stereo_setup(8000, 8, 8), "long modes with different floors"),
    "equal_sizes_9": (lambda: sg.stereo_setup(8000, 9, 9), "long modes with different floors"),
    "equal_sizes_10": (lambda: sg.stereo_setup(16000, 10, 10), "long modes with different floors"),
    "equal_sizes_11": (lambda: sg.stereo_setup(44100, 11, 11), "long modes with different floors"),
    "equal_sizes_12": (lambda: sg.stereo_setup(44100, 12, 12), "long modes with different floors"),
    "equal_sizes_13": (lambda: sg.stereo_setup(44100, 13, 13, residue_type=1), "long modes with different floors"),
}


def _census(setup):
    from lewton_amd import _native as N
    from lewton_amd import header
    idp, _cmt, stp = setup.headers()
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    buf = C.create_string_buffer(2048)
    N.lib.lw_debug_plan_census(ident._h, st._h, buf, 2048)
    return dict(x.split("=", 1) for x in buf.value.decode().split(" | "))


@pytest.mark.parametrize("name", sorted(CASES))
def test_plan_sends_the_shape_through_k_prep_not_the_generic_kernels(name):
    make, why = CASES[name]
    parts = _census(make())
    # (a coupling list that only needs a few more steps -- libvorbis' 5.1, three channels -- is evaluated inside k_long's waves)
    assert parts["long"].startswith("k_") and ("k_prep" in parts["long"] or "inside the waves" in parts["long"]) and why in parts["long"], parts
    if name in ("surround51_libvorbis_coupling", "three_channels"):
        assert parts["long"].startswith("k_long, coupling steps inside the waves"), parts


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("fmt_seed", [0, 1, 2])
def test_k_prep_shapes_three_ways(name, fmt_seed):
    from lewton_amd import _native as N
    from lewton_amd import audio, header
    from lewton_amd.batch import Batch
    setup = CASES[name][0]()
    rng = np.random.default_rng(fmt_seed)
    seqs = [sg.random_stream(setup, rng, 24, seed=100 * fmt_seed + q, p_floor_unused=0.1, p_damage=0.04) for q in range(fz.DISTINCT)]
    idp, _cmt, stp = setup.headers()
    # (the seed only picks the sample format and the stream count of run_setup: 3 k + fmt_seed -> format fmt_seed)
    checked, kernels, line, _dev = fz.run_setup(3 * 7 + fmt_seed, setup.channels, idp, stp, seqs, 24 * 10, rng,
                                                (audio, header, Batch, po, N), length=24)
    assert checked > 150 and ("k_prep" in kernels or "inside the waves" in line), (kernels, line)
    assert kernels & {"k_long", "k_long10", "k_long12", "k_big", "k_short", "k_mix", "k_mix10"}, kernels
    if name.startswith("equal_sizes"):
        assert "k_imdct_generic" not in kernels, kernels      # (no packet of such a stream is left to the generic kernels)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["surround51_libvorbis_coupling", "two_long_modes", "floor0_long_blocks", "equal_sizes_10"])
def test_k_prep_shapes_packet_by_packet_and_through_the_ogg_reader(name):
    """the same shapes through the drop-in single-packet call (audio::read_audio_packet: one-packet batches, state through the
    pool) and through OggStreamReader's look-ahead queue (rings of batches), against the oracle's reader"""
    from lewton_amd import audio, header
    from lewton_amd import inside_ogg as IO
    from lewton_amd import ogg
    from oracle import pyogg
    setup = CASES[name][0]()
    rng = np.random.default_rng(11)
    pk = sg.random_stream(setup, rng, 90, seed=5, p_floor_unused=0.05)
    idp, cmt, stp = setup.headers()
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    o_id = po.Ident(idp)
    o_st = po.Setup(stp, o_id)
    pwr, opw = audio.PreviousWindowRight(), po.Pwr()
    for i, p in enumerate(pk[:60]):   # (contradicting window flags make some of them AudioBadFormat, audio.rs:1107-1111: same code then)
        try:
            want, o_rc = po.read_audio_packet(o_id, o_st, p, opw, "i16"), 0
        except po.OracleError as e:
            o_rc = e.code
        try:
            got, rc = audio.read_audio_packet(ident, st, p, pwr), 0
        except audio.AudioReadError as e:
            rc = e.code
        assert rc == o_rc and pwr.is_empty() == opw.is_empty(), (i, rc, o_rc)
        if rc == 0:
            assert got.shape == want.shape and np.array_equal(got, want), i
    w = ogg.PageWriter(0x99)
    w.add_packet(idp, 0, flush=True)
    w.add_packet(cmt, 0)
    w.add_packet(stp, 0, flush=True)
    gp = 0
    for i, p in enumerate(pk):
        gp += po.get_decoded_sample_count(o_id, o_st, p) if i else 0
        w.add_packet(p, gp, flush=(i % 9 == 8), eos=(i == len(pk) - 1))
    data = w.bytes()
    s, o = IO.OggStreamReader(data), pyogg.OggStreamReader(data, "i16")
    n = 0
    while True:
        got = s.read_dec_packets(32, "i16", 2)
        if got is None:
            break
        for a in got:
            try:
                b = o.read_dec_packet()
            except pyogg.VorbisError as e:
                assert e.kind == "BadAudio" and isinstance(a, audio.AudioReadError) and a.code == e.inner, (n, a, e)
            else:
                assert b is not None and not isinstance(a, Exception) and np.array_equal(a, b), n
            n += 1
    assert n == len(pk) and o.read_dec_packet() is None
