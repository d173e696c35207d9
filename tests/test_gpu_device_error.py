"""A device error of a launched batch (round 6, ADVICE): the one-launch mixed kernels raise the batch's device error word when a
short block's wave never sees the raw edges of its long neighbours; lw_debug_break_mix makes every producer withhold them.
Through the staging ring the verdict is LATCHED in the slot -- every collect until the release says LW_ERR_DEVICE, never LW_OK over
the failed batch's samples -- and the Ogg stream reader's look-ahead rolls the stream back (packets re-queued, PreviousWindowRight
restored, ring drained), so that the next call decodes the very same packets again, bit-exact.  Also here: the latch that keeps the
two kinds of tenant stream out of one process (include/lewton_amd.h)."""
import ctypes as C

import numpy as np
import pytest

from common import SETUPS, oracle_headers, po, sg
from lewton_amd import _native as N

pytestmark = pytest.mark.gpu


def _product(setup):
    from lewton_amd import audio, header
    idp, _, stp = setup.headers()
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    return audio, ident, st


def test_ring_collect_keeps_saying_device_error_until_release():
    from lewton_amd.ring import Ring
    setup = SETUPS["stereo"]()
    audio, ident, st = _product(setup)
    o_id, o_st = oracle_headers(setup)
    dec = audio.Decoder(ident, st, 0)
    n_streams, per = 256, 16                         # BASELINE configs[2] at its bench shape: one k_mix launch
    seqs = [sg.make_stream(setup, "LLSSSSSSSSL", per + 1, seed=40 + q) for q in range(8)]
    pwrs = [audio.PreviousWindowRight() for _ in range(n_streams)]
    ring = Ring(dec, 2, n_streams * per, "i16")
    ring.submit(ring.marshal([(seqs[s % 8][0], pwrs[s]) for s in range(n_streams)]), n_threads=4)   # prime
    ring.collect()
    ring.release()
    saved = [(N.lw_pwr_len(p._h), p.data().copy()) for p in pwrs[:8]]
    items = [(seqs[s % 8][1 + t], pwrs[s]) for s in range(n_streams) for t in range(per)]
    m = ring.marshal(items)
    try:
        N.lw_debug_break_mix(2000)
        ring.submit(m, n_threads=4)
        assert "k_mix" in ring.last_kernels
        res = C.POINTER(N.PacketResult)()
        n, pcm, elems = C.c_size_t(0), C.c_void_p(0), C.c_size_t(0)
        for _again in range(3):                      # the verdict is the slot's until it is released
            assert N.lw_ring_collect(ring._h, C.byref(res), C.byref(n), C.byref(pcm), C.byref(elems)) == N.ERR_DEVICE
            assert n.value == len(items) and all(res[i].status == N.ERR_DEVICE for i in range(0, n.value, 97))
        ring.release()
    finally:
        N.lw_debug_break_mix(0)
    # the failed batch advanced the host halves of the window states: start the streams over, as the header says, then decode on
    for p in pwrs:
        p.reset()
    ring.submit(ring.marshal([(seqs[s % 8][0], pwrs[s]) for s in range(n_streams)]), n_threads=4)
    ring.collect()
    ring.release()
    for (ln, d), p in zip(saved, pwrs[:8]):
        assert N.lw_pwr_len(p._h) == ln and np.array_equal(p.data(), d)
    ring.submit(m, n_threads=4)
    out, flat = ring.collect()
    ring.release()
    want = {}
    for q in range(8):
        opw = po.Pwr()
        po.read_audio_packet(o_id, o_st, seqs[q][0], opw, "i16")
        want[q] = [np.asarray(po.read_audio_packet(o_id, o_st, p, opw, "i16")).reshape(-1) for p in seqs[q][1:]]
    for k, (status, ns, off) in enumerate(out):
        w = want[(k // per) % 8][k % per]
        assert status == 0 and np.array_equal(flat[off:off + 2 * ns], w), k
    ring.close()
    pwrs.clear()
    dec.close()


def test_ogg_look_ahead_rolls_back_after_a_device_error():
    from lewton_amd import inside_ogg as IO
    from lewton_amd import ogg
    from oracle import pyogg
    setup = SETUPS["stereo"]()
    idp, cmt, stp = setup.headers()
    pk = sg.make_stream(setup, "LLSSSSSSSSL", 600, seed=8)
    ident = po.Ident(idp)
    ost = po.Setup(stp, ident)
    w = ogg.PageWriter(0x51)
    w.add_packet(idp, 0, flush=True)
    w.add_packet(cmt, 0)
    w.add_packet(stp, 0, flush=True)
    gp = 0
    for i, p in enumerate(pk):
        gp += po.get_decoded_sample_count(ident, ost, p) if i else 0
        w.add_packet(p, gp, flush=(i % 7 == 6), eos=(i == len(pk) - 1))
    data = w.getvalue() if hasattr(w, "getvalue") else w.bytes()
    s, o = IO.OggStreamReader(data), pyogg.OggStreamReader(data, "i16")
    got = s.read_dec_packets(256, "i16", 2)          # a first batch, intact
    assert len(got) == 256
    for a in got:
        assert np.array_equal(a, o.read_dec_packet())
    N.lw_debug_break_mix(2000)
    try:
        with pytest.raises(IO.VorbisError) as e:     # the batch at the head of the look-ahead fails on the device ...
            s.read_dec_packets(256, "i16", 2)
        assert e.value.code == N.ERR_DEVICE
    finally:
        N.lw_debug_break_mix(0)
    n = 256
    while True:                                      # ... and nothing of it was lost or handed out: the same packets again, then the rest
        got = s.read_dec_packets(256, "i16", 2)
        if got is None:
            break
        for a in got:
            b = o.read_dec_packet()
            assert b is not None and np.array_equal(a, b), n
            n += 1
    assert n == len(pk) and o.read_dec_packet() is None


def test_cu_shares_are_refused_once_the_copier_stream_exists():
    """the sharder's logical shards (tenants without a CU share) make the device's copier stream; from then on this process gets no
    CU-masked streams on that device: lw_decoder_set_cu_share says LW_ERR_UNSUPPORTED, parts = 1 (the whole device) still works"""
    from lewton_amd import shard
    setup = SETUPS["stereo"]()
    audio, ident, st = _product(setup)
    sh = shard.Sharder(ident, st, [0, 0], max_packets_per_shard=64, samples="i16")
    dec = audio.Decoder(ident, st, 0)
    assert N.lw_decoder_set_cu_share(dec._h, 0, 2) == N.ERR_UNSUPPORTED
    assert N.lw_decoder_set_cu_share(dec._h, 0, 1) == 0 and N.lw_decoder_cu_count(dec._h) == N.lw_decoder_device_cu_count(dec._h)
    blocks, res = sh.decode([(s, p) for s in range(4) for p in sg.make_stream(setup, "L", 3, seed=s)], n_threads=2)
    assert all(r[0] == 0 for r in res)
    sh.close()
    dec.close()
