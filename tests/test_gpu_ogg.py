"""Ogg layer, GPU part: lewton_amd.inside_ogg.OggStreamReader (C++ host layer + HIP decode path) against the
oracle's OggStreamReader (oracle/pyogg.py over oracle/lewton_oracle.c) -- i16 bit-exact, f32 within 1e-5 --
on a real Ogg/Vorbis file and on synthetic / chained / trimmed streams (inside_ogg.rs:114-313)."""
import os

import numpy as np
import pytest

from common import SETUPS, sg
from lewton_amd import inside_ogg as IO
from lewton_amd import ogg
from oracle import pyogg
from oracle import pyoracle as po
from test_ogg import GOLDEN, _vorbis_stream

pytestmark = pytest.mark.gpu

_OFMT = {"i16": "i16", "f32": "f32", "i16_interleaved": "i16_itl"}


def _same(a, b, fmt):
    assert a.shape == b.shape, (a.shape, b.shape)
    if fmt == "f32":
        assert np.max(np.abs(a - b), initial=0.0) <= 1e-5   # north_star tolerance for f32 output
    else:
        assert np.array_equal(a, b)


def _drain(data, fmt):
    s, o = IO.OggStreamReader(data), pyogg.OggStreamReader(data, _OFMT[fmt])
    n = 0
    while True:
        a, b = s.read_dec_packet_generic(fmt), o.read_dec_packet()
        assert (a is None) == (b is None), n
        if a is None:
            break
        _same(a, b, fmt)
        assert s.get_last_absgp() == o.get_last_absgp() and s.stream_serial() == o.stream_serial
        n += 1
    return n


@pytest.mark.parametrize("fmt", ["i16", "f32", "i16_interleaved"])
def test_real_file_matches_oracle(fmt):
    data = open(GOLDEN, "rb").read()
    assert _drain(data, fmt) == 26
    # plausibility figures of SURVEY 8(c): a clean fading-in tone, 4 short then long blocks
    s = IO.OggStreamReader(data)
    pk = []
    while True:
        p = s.read_dec_packet()
        if p is None:
            break
        pk.append(p)
    assert [p.shape[1] for p in pk] == [0, 128, 128, 128, 576] + [1024] * 21
    x = np.concatenate(pk, axis=1)
    assert list(x[0][x[0] != 0][:12]) == [1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 3]
    assert abs(np.abs(x).max() / 32768 - 0.6178) < 1e-4


@pytest.mark.parametrize("name,pattern,count,per_page,trim", [
    ("stereo", "LSSL", 24, 5, 0), ("stereo", "L", 20, 3, 700), ("surround51", "LLSSSL", 18, 4, 37),
    ("mono_small", "SL", 15, 1, 5), ("stereo", "S", 9, 2, 127)])
@pytest.mark.parametrize("fmt", ["i16", "f32", "i16_interleaved"])
def test_synthetic_stream_with_final_trim(name, pattern, count, per_page, trim, fmt):
    # the last packet is truncated so that the stream ends at the final granule position (inside_ogg.rs:219-227)
    _setup, pk, w = _vorbis_stream(name, pattern, count, per_page=per_page, trim=trim)
    assert _drain(w.bytes(), fmt) == count


def _chained():
    parts = []
    for k, (name, pattern, count, trim) in enumerate([("stereo", "LSL", 9, 11), ("surround51", "L", 6, 0),
                                                      ("mono_small", "SLL", 7, 3)]):
        parts.append(_vorbis_stream(name, pattern, count, seed=20 + k, serial=0x100 + k, per_page=2, trim=trim)[2].bytes())
    return b"".join(parts)


@pytest.mark.parametrize("fmt", ["i16", "f32"])
def test_chained_streams_reinitialise_the_context(fmt):
    data = _chained()
    s, o = IO.OggStreamReader(data), pyogg.OggStreamReader(data, _OFMT[fmt])
    serials, chans = [], []
    while True:
        a, b = s.read_dec_packet_generic(fmt), o.read_dec_packet()
        assert (a is None) == (b is None)
        if a is None:
            break
        _same(a, b, fmt)
        assert s.get_last_absgp() == o.get_last_absgp()
        if not serials or serials[-1] != s.stream_serial():
            serials.append(s.stream_serial())
            chans.append(s.ident_hdr.audio_channels)
    assert serials == [0x100, 0x101, 0x102] and chans == [2, 6, 1]


@pytest.mark.parametrize("dev_entropy", [False, True])
@pytest.mark.parametrize("max_packets", [1, 4, 64])
def test_look_ahead_queue_equals_packet_by_packet(max_packets, dev_entropy):
    for data in (open(GOLDEN, "rb").read(), _chained(), _vorbis_stream("stereo", "LLSL", 40, per_page=7, trim=300)[2].bytes()):
        one = IO.OggStreamReader(data)
        ref = []
        while True:
            p = one.read_dec_packet()
            if p is None:
                break
            ref.append(p)
        s = IO.OggStreamReader(data)
        if dev_entropy:                    # floors and residues of the look-ahead batches decoded by k_entropy
            s.set_entropy_on_device(True)
        got = []
        while True:
            r = s.read_dec_packets(max_packets)
            if r is None:
                break
            if not r:                      # chain boundary: cross it with the single-packet call
                p = s.read_dec_packet()
                if p is None:
                    break
                r = [p]
            assert len(r) <= max_packets
            got += r
        assert len(got) == len(ref)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b)
        assert s.get_last_absgp() == one.get_last_absgp()


@pytest.mark.parametrize("to_skip", [0, 1, 500, 1024, 5000, 9999, 10 ** 7])
def test_skip_samples_linear(to_skip):
    _setup, _pk, w = _vorbis_stream("stereo", "LSSLL", 30, per_page=4, trim=100)
    data = w.bytes()
    s, o = IO.OggStreamReader(data), pyogg.OggStreamReader(data)
    s.read_dec_packet(), o.read_dec_packet()
    (a, la), (b, lb) = s.skip_samples_linear(to_skip), o.skip_samples_linear(to_skip)
    assert la == lb and (a is None) == (b is None)
    if a is not None:
        assert np.array_equal(a, b)
    assert s.get_last_absgp() == o.get_last_absgp()
    a, b = s.read_dec_packet(), o.read_dec_packet()      # and decoding carries on identically
    assert (a is None) == (b is None)
    if a is not None:
        assert np.array_equal(a, b)


@pytest.mark.parametrize("goal", [0, 3000, 12345, 10 ** 9])
def test_seek_absgp_pg_then_decode(goal):
    _setup, _pk, w = _vorbis_stream("stereo", "LLSL", 40, per_page=3)
    data = w.bytes()
    s, o = IO.OggStreamReader(data), pyogg.OggStreamReader(data)
    for _ in range(5):
        s.read_dec_packet(), o.read_dec_packet()
    s.seek_absgp_pg(goal), o.seek_absgp_pg(goal)
    assert s.get_last_absgp() is None
    first = True
    while True:
        a, b = s.read_dec_packet(), o.read_dec_packet()
        assert (a is None) == (b is None)
        if a is None:
            break
        if first and goal > 0:
            assert a.shape[1] == 0        # the window state was reset: the first packet after a seek only primes it
        first = False
        assert np.array_equal(a, b)
        assert s.get_last_absgp() == o.get_last_absgp()
        if s.get_last_absgp() is not None and goal < 10 ** 9:
            pass
    # page granularity: the position reached is <= the target
    s.seek_absgp_pg(goal)
    s.read_dec_packet()
    while s.get_last_absgp() is None and s.read_dec_packet() is not None:
        pass


def test_bad_audio_packet_surfaces_as_bad_audio():
    setup, pk, w0 = _vorbis_stream("stereo", "L", 4)
    idp, cmt, stp = setup.headers()
    w = ogg.PageWriter(5)
    w.add_packet(idp, 0, flush=True)
    w.add_packet(cmt, 0)
    w.add_packet(stp, 0, flush=True)
    w.add_packet(pk[0], 0)
    w.add_packet(idp, 1024, flush=True, eos=True)        # a header packet where audio is expected
    s, o = IO.OggStreamReader(w.bytes()), pyogg.OggStreamReader(w.bytes())
    assert s.read_dec_packet().shape == (2, 0)
    o.read_dec_packet()
    with pytest.raises(IO.VorbisError) as e:
        s.read_dec_packet()
    with pytest.raises(pyogg.VorbisError) as eo:
        o.read_dec_packet()
    assert e.value.kind == eo.value.kind == "BadAudio" and e.value.code == eo.value.inner == po.AUDIO_IS_HEADER


@pytest.mark.parametrize("lookahead", [1, 7, 1024])
def test_ogg2wav_example_matches_oracle(tmp_path, lookahead):
    """examples/ogg2wav.c (C ABI only): the WAV payload is the oracle's interleaved decode of the same file."""
    import subprocess
    from common import ROOT
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples"), "ogg2wav"])
    setup, pk, w = _vorbis_stream("stereo", "LLSSLLL", 60, per_page=6, trim=333)
    src = tmp_path / "in.ogg"
    src.write_bytes(w.bytes())
    for path in (str(src), GOLDEN):
        dst = tmp_path / "out.wav"
        subprocess.check_call([os.path.join(ROOT, "examples", "ogg2wav"), path, str(dst), str(lookahead)])
        raw = dst.read_bytes()
        assert raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt " and raw[36:40] == b"data"
        n_bytes = int.from_bytes(raw[40:44], "little")
        got = np.frombuffer(raw[44:44 + n_bytes], np.int16)
        o = pyogg.OggStreamReader(open(path, "rb").read(), "i16_itl")
        want = []
        while True:
            p = o.read_dec_packet()
            if p is None:
                break
            want.append(p)
        want = np.concatenate(want)
        assert len(raw) == 44 + n_bytes and np.array_equal(got, want)


@pytest.mark.parametrize("dev_entropy", [False, True])
@pytest.mark.parametrize("k,singles,skip,goal", [(4, 2, 900, None), (9, 0, 1, 5000), (3, 4, 3000, 0), (64, 1, 128, 20000)])
def test_single_skip_seek_between_batched_calls_sample_values(k, singles, skip, goal, dev_entropy):
    """The look-ahead pipeline (three batches staged / in flight behind read_dec_packets) rolled back by read_dec_packet,
    skip_samples_linear and seek_absgp_pg (inside_ogg.rs:167-313): every sample delivered afterwards equals the oracle's
    OggStreamReader driven with the same calls packet by packet -- i.e. the PreviousWindowRight really is back at the state
    after the last DELIVERED packet, on the device too."""
    data = _vorbis_stream("stereo", "LLSLLLSSL", 80, per_page=4, trim=123)[2].bytes()
    s = IO.OggStreamReader(data)
    if dev_entropy:
        s.set_entropy_on_device(True)
    o = pyogg.OggStreamReader(data)

    def same(a, b):
        assert (a is None) == (b is None)
        if a is not None:
            assert a.shape == b.shape and np.array_equal(a, b)

    def batch():
        r = s.read_dec_packets(k)
        want = []
        for _ in range(k):
            d = o.read_dec_packet()
            if d is None:
                break
            want.append(d)
        if r is None:
            assert not want
            return False
        assert len(r) == len(want)
        for a, b in zip(r, want):
            same(a, b)
        assert s.get_last_absgp() == o.get_last_absgp()
        return True

    alive = batch()
    for _ in range(singles):
        if alive:
            a, b = s.read_dec_packet(), o.read_dec_packet()
            same(a, b)
            alive = a is not None
    alive = alive and batch()
    if alive and skip:
        (a, la), (b, lb) = s.skip_samples_linear(skip), o.skip_samples_linear(skip)
        same(a, b)
        assert la == lb
    alive = alive and batch()
    if alive and goal is not None:
        s.seek_absgp_pg(goal)
        o.seek_absgp_pg(goal)
        alive = batch()
    while alive:
        alive = batch()


# ---- lw_ogg_stream_set_read_ahead: read_dec_packet served from batches decoded ahead
def _damaged_stream(seed, count=90, p_bad=0.08):
    """a stereo stream with some packets damaged (truncated / a header bit / flipped bits): several of them fail with an
    AudioReadError, some of those after the previous window has been taken (audio.rs:1083, :1107-1111)"""
    setup, pk, _ = _vorbis_stream("stereo", "LLSLLLSSL", count, seed=seed)
    rng = np.random.default_rng(seed)
    idp, cmt, stp = setup.headers()
    o_id = po.Ident(idp)
    o_st = po.Setup(stp, o_id)
    w = ogg.PageWriter(0x77)
    w.add_packet(idp, 0, flush=True)
    w.add_packet(cmt, 0)
    w.add_packet(stp, 0, flush=True)
    gp = 0
    for i, p in enumerate(pk):
        if i and i + 1 < len(pk) and rng.random() < p_bad:
            kind = int(rng.integers(0, 3))
            if kind == 0:
                p = p[: max(1, len(p) // int(rng.integers(2, 9)))]
            elif kind == 1:
                p = bytes([p[0] | 1]) + p[1:]               # AudioIsHeader
            else:
                q = bytearray(p)
                for _ in range(3):
                    q[int(rng.integers(0, len(q)))] ^= 1 << int(rng.integers(0, 8))
                p = bytes(q)
        try:
            gp += po.get_decoded_sample_count(o_id, o_st, p) if i else 0
        except po.OracleError:
            pass
        w.add_packet(p, gp, flush=(i % 5 == 4), eos=(i == len(pk) - 1))
    return w.bytes()


def _one(s, o, fmt="i16"):
    """one read_dec_packet on both readers: the same samples or the same BadAudio code; returns False at the end of the stream"""
    ea = eb = a = b = None
    try:
        a = s.read_dec_packet_generic(fmt)
    except IO.VorbisError as e:
        ea = e
    try:
        b = o.read_dec_packet()
    except pyogg.VorbisError as e:
        eb = e
    assert (ea is None) == (eb is None), (ea, eb)
    if ea is not None:
        assert ea.kind == eb.kind == "BadAudio" and ea.code == eb.inner, (ea, eb)
    else:
        assert (a is None) == (b is None)
        if a is None:
            return False
        _same(a, b, fmt)
    assert s.get_last_absgp() == o.get_last_absgp() and s.stream_serial() == o.stream_serial
    return True


@pytest.mark.parametrize("dev_entropy", [False, True])
@pytest.mark.parametrize("k", [1, 5, 64, 1024])
def test_read_ahead_serves_the_packet_by_packet_sequence(k, dev_entropy):
    """every call's samples / error, granule position and serial, on a real file, a damaged stream, a trimmed one and a chained one"""
    for data in (open(GOLDEN, "rb").read(), _damaged_stream(3), _vorbis_stream("surround51", "LLSSSL", 30, per_page=4, trim=37)[2].bytes(),
                 _chained()):
        s, o = IO.OggStreamReader(data), pyogg.OggStreamReader(data)
        s.set_read_ahead(k, 2)
        if dev_entropy:
            s.set_entropy_on_device(True)
        n = 0
        while _one(s, o):
            n += 1
        assert n >= 20


@pytest.mark.parametrize("seed", range(8))
def test_read_ahead_under_a_random_call_script(seed):
    """read_dec_packet (three formats), read_dec_packets, skip_samples_linear, seek_absgp_pg, set_read_ahead and set_entropy_on_device in
    random order on a damaged stream: whatever lands in the middle of a served batch returns its packets and re-makes the
    PreviousWindowRight of the last packet handed out -- incl. the packets that failed behind it -- so every sample that follows
    equals the oracle reader's, which was driven packet by packet"""
    rng = np.random.default_rng(100 + seed)
    data = _damaged_stream(10 + seed, count=140, p_bad=0.12)
    s, o = IO.OggStreamReader(data), pyogg.OggStreamReader(data)
    s.set_read_ahead(int(rng.choice([2, 7, 33])), 2)
    alive, calls = True, 0
    while calls < 260:
        calls += 1
        r = rng.random()
        if not alive:          # the end of the stream: somewhere else, on
            r, alive = 0.88, True
        if r < 0.70:
            alive = _one(s, o)
        elif r < 0.78:
            kk = int(rng.integers(1, 9))
            got = s.read_dec_packets(kk)
            if got is None:
                assert o.read_dec_packet() is None
                alive = False
                continue
            for a in got:
                try:
                    b = o.read_dec_packet()
                except pyogg.VorbisError as e:
                    assert isinstance(a, Exception) and a.code == e.inner
                else:
                    assert not isinstance(a, Exception) and np.array_equal(a, b)
            assert s.get_last_absgp() == o.get_last_absgp()
        elif r < 0.86:
            n = int(rng.choice([0, 1, 700, 3000]))
            try:
                a, la = s.skip_samples_linear(n)
                ea = None
            except IO.VorbisError as e:
                ea = e
            try:
                b, lb = o.skip_samples_linear(n)
                eb = None
            except pyogg.VorbisError as e:
                eb = e
            assert (ea is None) == (eb is None), (ea, eb)
            if ea is None:
                assert la == lb and (a is None) == (b is None)
                if a is not None:
                    assert np.array_equal(a, b)
                assert s.get_last_absgp() == o.get_last_absgp()
        elif r < 0.90:
            goal = int(rng.integers(0, 120000))
            s.seek_absgp_pg(goal)
            o.seek_absgp_pg(goal)
        elif r < 0.95:
            s.set_read_ahead(int(rng.choice([0, 1, 4, 50])), 2)
        else:
            s.set_entropy_on_device(bool(rng.integers(0, 2)))
