"""Ogg layer, CPU part: the C++ demultiplexer of the library against the oracle's
restatement (oracle/pyogg.py, RFC 3533) on a real Ogg/Vorbis file and on synthetic physical streams built
with lewton_amd.ogg.PageWriter; header bootstrap of OggStreamReader (inside_ogg.rs:30-49)."""
import io
import os
import struct

import numpy as np
import pytest

from common import SETUPS, sg
from lewton_amd import inside_ogg as IO
from lewton_amd import ogg
from oracle import pyogg

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "invalid_keypress.ogg")


def _attrs(p):
    return (p.data, p.stream_serial(), p.absgp_page(), p.first_in_stream(), p.last_in_stream(), p.first_in_page(),
            p.last_in_page())


def _oattrs(p):
    return (p.data, p.serial, p.absgp_page, p.first_in_stream, p.last_in_stream, p.first_in_page, p.last_in_page)


def _all_packets(data, src=None):
    r = ogg.PacketReader(data if src is None else src)
    out = []
    while True:
        p = r.read_packet()
        if p is None:
            return out
        out.append(_attrs(p))


def _all_oracle(data):
    r = pyogg.PacketReader(data)
    out = []
    while True:
        p = r.read_packet()
        if p is None:
            return out
        out.append(_oattrs(p))


def test_crc_known_answers():
    # CRC-32 with polynomial 0x04c11db7, init 0, no reflection, no final xor ("123456789" -> ~0x765E7680, the
    # complement of the CRC-32/CKSUM check value, which shares these parameters apart from the final xor)
    assert ogg.crc32(b"123456789") == 0x89A1897F == pyogg.crc32_ogg(b"123456789")
    assert ogg.crc32(b"") == 0
    rng = np.random.default_rng(0)
    for n in (1, 27, 255, 4096):
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert ogg.crc32(b) == pyogg.crc32_ogg(b)
        assert ogg.crc32(b[n // 2:], ogg.crc32(b[: n // 2])) == ogg.crc32(b)  # incremental


def test_real_file_demux_matches_oracle():
    data = open(GOLDEN, "rb").read()
    got, want = _all_packets(data), _all_oracle(data)
    assert got == want and len(got) == 29
    assert [len(p[0]) for p in got[:3]] == [30, 45, 3832]           # ident, comment, setup
    assert got[0][3] and not got[1][3] and got[-1][4]                  # first_in_stream / last_in_stream
    assert got[-1][2] == 22050                                         # final granule position
    # the same through a path and through Read + Seek callbacks
    assert _all_packets(None, GOLDEN) == want
    assert _all_packets(None, io.BytesIO(data)) == want


def _rand_packets(rng, n, sizes=(0, 1, 17, 254, 255, 256, 509, 510, 511, 1000, 70000)):
    return [rng.integers(0, 256, int(rng.choice(sizes)), dtype=np.uint8).tobytes() for _ in range(n)]


@pytest.mark.parametrize("max_segments", [255, 17, 3, 1])
def test_synthetic_lacing_and_continuation(max_segments):
    rng = np.random.default_rng(max_segments)
    pk = _rand_packets(rng, 60)
    w = ogg.PageWriter(0xABCD, max_segments)
    gp = 0
    for i, p in enumerate(pk):
        gp += 100
        w.add_packet(p, gp, flush=(i % 7 == 6), eos=(i == len(pk) - 1))
    data = w.bytes()
    got, want = _all_packets(data), _all_oracle(data)
    assert got == want
    assert [g[0] for g in got] == pk                                   # round trip
    assert got[-1][4] and sum(g[4] for g in got) == 1
    # (the begin-of-stream attribute belongs to the packet that ENDS on the first page)
    assert got[0][3] == (len(pk[0]) < 255 * max_segments)
    # every packet carries the granule position of the page it ended on
    assert [g[2] for g in got if g[6]] == sorted(g[2] for g in got if g[6])


def test_multiplexed_and_chained_streams():
    rng = np.random.default_rng(5)
    a, b, c = ogg.PageWriter(1, 9), ogg.PageWriter(2, 5), ogg.PageWriter(3)
    for w, n in ((a, 40), (b, 25), (c, 10)):
        for i, p in enumerate(_rand_packets(rng, n, sizes=(0, 3, 200, 255, 600, 3000))):
            w.add_packet(p, 10 * (i + 1), flush=(i % 3 == 0), eos=(i == n - 1))
    data = ogg.interleave_pages(a, b) + c.bytes()                      # two multiplexed streams, then a chained link
    got, want = _all_packets(data), _all_oracle(data)
    assert got == want
    for serial, n in ((1, 40), (2, 25), (3, 10)):
        mine = [g for g in got if g[1] == serial]
        assert len(mine) == n and mine[0][3] and mine[-1][4]


def _expect_error(data, kind):
    with pytest.raises(pyogg.OggError) as eo:
        _all_oracle(data)
    assert eo.value.kind == kind
    with pytest.raises(ogg.OggReadError) as ep:
        _all_packets(data)
    assert ep.value.kind == kind


def test_container_errors():
    w = ogg.PageWriter(7, 4)
    for i in range(6):
        w.add_packet(bytes([i]) * 300, i + 1, flush=True, eos=(i == 5))
    good = w.bytes()
    assert len(_all_packets(good)) == 6
    first = len(w.pages[0])
    bad = bytearray(good)
    bad[first + 40] ^= 1                                               # body byte of the second page
    _expect_error(bytes(bad), "HashMismatch")
    bad = bytearray(good)
    bad[first] = ord("X")
    _expect_error(bytes(bad), "NoCapturePatternFound")
    bad = bytearray(good)
    bad[first + 4] = 1                                                 # stream structure version
    _expect_error(bytes(bad), "InvalidStreamStructVer")
    _expect_error(good[:-5], "ReadError")                              # truncated body
    _expect_error(good[: first + 10], "ReadError")                     # truncated header
    # read_packet_expected turns a clean end into an error; read_packet returns None
    r = ogg.PacketReader(good)
    for _ in range(6):
        r.read_packet_expected()
    assert r.read_packet() is None
    with pytest.raises(ogg.OggReadError) as e:
        r.read_packet_expected()
    assert e.value.kind == "ReadError"


def test_delete_unread_packets():
    w = ogg.PageWriter(9)
    for i in range(5):
        w.add_packet(b"p%d" % i, i, flush=(i == 2), eos=(i == 4))      # page 0: p0 p1 p2, page 1: p3 p4
    r = ogg.PacketReader(w.bytes())
    assert r.read_packet().data == b"p0"
    r.delete_unread_packets()
    assert r.read_packet().data == b"p3"


@pytest.mark.parametrize("big", [False, True])
def test_seek_absgp_matches_linear_oracle(big):
    # `big`: > 64 KiB per bisection interval, so the bisection loop (not only the final linear scan) runs
    rng = np.random.default_rng(11 + big)
    w = ogg.PageWriter(0x51, 40)
    n = 900 if big else 60
    gp = 0
    for i in range(n):
        gp += int(rng.integers(1, 2000))
        w.add_packet(rng.integers(0, 256, int(rng.integers(1, 5000 if big else 300)), dtype=np.uint8).tobytes(), gp,
                     flush=bool(rng.integers(0, 3) == 0), eos=(i == n - 1))
    data = w.bytes()
    assert (len(data) > 1 << 20) == big
    goals = [0, 1, gp // 3, gp // 2, gp - 1, gp, gp + 10] + [int(g) for g in rng.integers(0, gp, 12)]
    for goal in goals:
        for serial in (None, 0x51):
            r, o = ogg.PacketReader(data), pyogg.PacketReader(data)
            r.read_packet()
            o.read_packet()
            r.seek_absgp(serial, goal)
            o.seek_absgp(serial, goal)
            a, b = r.read_packet(), o.read_packet()
            assert (a is None) == (b is None), goal
            if a is not None:
                assert _attrs(a) == _oattrs(b), goal
    # a serial that does not occur: reading resumes at the start
    r = ogg.PacketReader(data)
    r.seek_absgp(0x99, gp // 2)
    assert r.read_packet().first_in_stream()


def _vorbis_stream(name="stereo", pattern="LSSL", count=12, seed=3, serial=0x77, per_page=5, trim=0, bos_junk=False):
    from oracle import pyoracle as po
    setup = SETUPS[name]()
    idp, cmt, stp = setup.headers()
    pk = sg.make_stream(setup, pattern, count, seed=seed)
    ident = po.Ident(idp)
    st = po.Setup(stp, ident)
    w = ogg.PageWriter(serial)
    w.add_packet(idp, 0, flush=True)
    w.add_packet(cmt, 0)
    w.add_packet(stp, 0, flush=True)
    gp = 0
    for i, p in enumerate(pk):
        cnt = po.get_decoded_sample_count(ident, st, p) if i else 0   # the first packet only primes the window
        gp += cnt
        last = i == len(pk) - 1
        w.add_packet(p, gp - (trim if last else 0), flush=(i % per_page == per_page - 1), eos=last)
    return setup, pk, w


def test_stream_reader_header_bootstrap_without_gpu():
    setup, pk, w = _vorbis_stream()
    other = ogg.PageWriter(0x1234)                                     # a foreign logical stream multiplexed in
    for i in range(4):
        other.add_packet(b"\x01foreign", i, flush=True, eos=(i == 3))
    data = ogg.interleave_pages(w, other)
    s = IO.OggStreamReader(data)
    assert s.stream_serial() == 0x77 and s.get_last_absgp() is None
    assert (s.ident_hdr.audio_channels, s.ident_hdr.audio_sample_rate) == (2, 44100)
    assert (s.ident_hdr.blocksize_0, s.ident_hdr.blocksize_1) == (8, 11)
    assert s.comment_hdr.vendor == "lewton_amd streamgen" and s.comment_hdr.comment_list == [("TITLE", "synthetic")]
    (ident, comment, _setup), serial = IO.read_headers(ogg.PacketReader(data))
    assert serial == 0x77 and ident.blocksize_1 == 11 and comment.vendor == "lewton_amd streamgen"
    # header errors surface as VorbisError::BadHeader, container errors as VorbisError::OggError
    with pytest.raises(IO.VorbisError) as e:
        IO.OggStreamReader(other.bytes())
    assert e.value.kind == "BadHeader" and e.value.inner.kind == "NotVorbisHeader"
    with pytest.raises(IO.VorbisError) as e:
        IO.OggStreamReader(w.pages[0])                                 # the physical stream ends after the ident header
    assert e.value.kind == "OggError" and e.value.inner.kind == "ReadError"
    real = IO.OggStreamReader(GOLDEN)
    assert real.comment_hdr.vendor.startswith("Xiph.Org libVorbis") and real.stream_serial() == 0x54C6F544


def test_mutated_containers_product_equals_oracle():
    """Random damage to a physical stream: the C++ demultiplexer and the oracle deliver the same packets and stop with
    the same error kind (nothing may crash or hang; sanitizer coverage of the packet layer is in test_fuzz_host.py)."""
    rng = np.random.default_rng(29)
    a, b = ogg.PageWriter(10, 7), ogg.PageWriter(11, 3)
    for w, n in ((a, 30), (b, 20)):
        for i, p in enumerate(_rand_packets(rng, n, sizes=(0, 5, 100, 255, 300, 900))):
            w.add_packet(p, 7 * (i + 1), flush=(i % 4 == 3), eos=(i == n - 1))
    good = ogg.interleave_pages(a, b)
    kinds = set()
    for trial in range(120):
        m = bytearray(good)
        k = trial % 4
        if k == 0:
            m = m[: int(rng.integers(1, len(m)))]
        elif k == 1:
            m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))
        elif k == 2:
            i = int(rng.integers(0, len(m)))
            del m[i:i + int(rng.integers(1, 40))]
        else:
            i = int(rng.integers(0, len(m)))
            m[i:i] = rng.integers(0, 256, int(rng.integers(1, 20)), dtype=np.uint8).tobytes()
        m = bytes(m)
        r, o = ogg.PacketReader(m), pyogg.PacketReader(m)
        n = 0
        while True:
            ea = eb = None
            pa = pb = None
            try:
                pa = r.read_packet()
            except ogg.OggReadError as e:
                ea = e.kind
            try:
                pb = o.read_packet()
            except pyogg.OggError as e:
                eb = e.kind
            assert ea == eb, (trial, n, ea, eb)
            if ea is not None:
                kinds.add(ea)
                break
            assert (pa is None) == (pb is None), (trial, n)
            if pa is None:
                break
            assert _attrs(pa) == _oattrs(pb), (trial, n)
            n += 1
    assert {"HashMismatch", "ReadError", "NoCapturePatternFound"} <= kinds


def test_seek_through_file_and_callback_sources(tmp_path):
    rng = np.random.default_rng(41)
    w = ogg.PageWriter(0x61, 30)
    gp = 0
    for i in range(400):
        gp += int(rng.integers(1, 3000))
        w.add_packet(rng.integers(0, 256, int(rng.integers(1, 2000)), dtype=np.uint8).tobytes(), gp,
                     flush=bool(rng.integers(0, 4) == 0), eos=(i == 399))
    data = w.bytes()
    path = tmp_path / "s.ogg"
    path.write_bytes(data)
    for goal in (0, gp // 4, gp // 2, gp - 1, gp + 5):
        want = pyogg.PacketReader(data)
        want.seek_absgp(0x61, goal)
        b = want.read_packet()
        for src in (str(path), io.BytesIO(data)):
            r = ogg.PacketReader(src)
            r.read_packet()
            r.seek_absgp(0x61, goal)
            a = r.read_packet()
            assert (a is None) == (b is None)
            if a is not None:
                assert _attrs(a) == _oattrs(b)


def test_decode_without_a_gpu_fails_loudly():
    """No CPU fallback: on a machine without a usable GPU every decoding call reports LW_ERR_DEVICE (the parity tests
    of the decode path carry the gpu marker)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    s = IO.OggStreamReader(GOLDEN)
    with pytest.raises(IO.VorbisError) as e:
        s.read_dec_packet()
    assert e.value.kind == "Library" and e.value.code == 33           # LW_ERR_DEVICE
    with pytest.raises(IO.VorbisError):
        s.read_dec_packets(8)
    from lewton_amd import capi
    setup = SETUPS["stereo"]()
    idp, cmt, stp = setup.headers()
    ed = capi.make_extradata(idp, cmt, stp)
    ctx = capi.lewton_context_from_extradata(ed, len(ed))
    assert capi.decode_packet(ctx, sg.make_stream(setup, "L", 1, seed=1)[0]) == (2, None)   # "no samples can be produced"
    capi.lewton_context_drop(ctx)
