"""Entropy stage on the device (lw_*_set_entropy_on_device, k_entropy) against the oracle (-m gpu): the packets themselves go to
the GPU, one lane decodes floors and residues of one packet, the synthesis kernels run behind it.  i16 PCM bit-exact, the
pre-inverse-coupling residue tap equal to the host stage's, statuses of damaged packets equal to the oracle's.  The same
algorithm is held against the host entropy stage on the CPU in tests/test_dev_entropy_host.py."""
import numpy as np
import pytest

from common import FLOOR0_SETUPS, HOST_SETUPS, SETUPS, oracle_headers, po, sg

pytestmark = pytest.mark.gpu


def _product(setup):
    from lewton_amd import audio, header
    idp, _, stp = setup.headers()
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    return audio, ident, st


def _damage(pk, rng):
    out = []
    for p in pk:
        p = bytearray(p)
        r = rng.random()
        if r < 0.10 and len(p) > 1:
            p = p[: int(rng.integers(0, len(p)))]
        elif r < 0.20:
            for _ in range(int(rng.integers(1, 4))):
                p[int(rng.integers(0, len(p)))] ^= 1 << int(rng.integers(0, 8))
        out.append(bytes(p))
    return out


@pytest.mark.parametrize("name,pattern", [("stereo", "L"), ("stereo", "LLSSSSLLSL"), ("stereo_t1", "LSL"), ("mono_small", "LSSLL"),
                                          ("stereo_9_12", "LLS"), ("stereo_7_7", "LSL"), ("surround51", "LLSL"), ("stereo_6_13", "LSL"),
                                          ("stereo_spill_t1", "LSLL"), ("stereo_spill_t2", "LLSL"), ("stereo_single_entry", "LLS"),
                                          ("surround51_bookless", "LLSL"), ("multichannel12", "LSL")])
def test_ring_with_device_entropy_matches_oracle(name, pattern):
    from lewton_amd.ring import Ring
    setup = HOST_SETUPS[name]()
    audio, ident, st = _product(setup)
    o_id, o_st = oracle_headers(setup)
    dec = audio.decoder_for(ident, st)
    n_streams, per, n_batches = 40, 6, 5
    rng = np.random.default_rng(12)
    streams = [_damage(sg.make_stream(setup, pattern, per * n_batches, seed=900 + s, p_floor_unused=0.1), rng) for s in range(n_streams)]
    pwrs = [audio.PreviousWindowRight() for _ in range(n_streams)]
    opws = [po.Pwr() for _ in range(n_streams)]
    ring = Ring(dec, 3, n_streams * per, "i16")
    assert ring.set_entropy_on_device(True)
    ch = setup.channels
    pending = []

    def check(items, res, pcm):
        for (pkt, s), (status, m, off) in zip(items, res):
            try:
                want = po.read_audio_packet(o_id, o_st, pkt, opws[s], "i16")
                rc = 0
            except po.OracleError as e:
                rc = e.code
            assert status == rc
            if rc == 0:
                got = pcm[off:off + m * ch]
                assert got.size == want.size and np.array_equal(got, want.reshape(-1)), s

    for b in range(n_batches):
        items = [(streams[s][b * per + t], s) for s in range(n_streams) for t in range(per)]
        if ring.in_flight == ring.slots:
            res, pcm = ring.collect()
            ring.release()
            check(pending.pop(0), res, pcm)
        ring.submit(ring.marshal([(p, pwrs[s]) for p, s in items]), n_threads=2)
        assert "k_entropy" in ring.last_kernels
        pending.append(items)
    while pending:
        res, pcm = ring.collect()
        ring.release()
        check(pending.pop(0), res, pcm)
    for s in range(0, n_streams, 7):
        assert np.array_equal(pwrs[s].data().view(np.uint32), opws[s].data(ch).view(np.uint32))
    ring.close()


@pytest.mark.parametrize("name,pattern", [("stereo", "LLSL"), ("stereo_t1", "LSL"), ("mono_small", "SLL"), ("stereo_spill_t1", "LSL"),
                                          ("stereo_spill_t2", "LSL")])
def test_device_entropy_records_equal_host_stage(name, pattern):
    """the residue vectors k_entropy leaves in HBM (tap before inverse coupling) and the final samples, against the same
    batch decoded by the host entropy stage"""
    from lewton_amd import _native as N
    from lewton_amd.batch import Batch
    setup = HOST_SETUPS[name]()
    audio, ident, st = _product(setup)
    dec = audio.decoder_for(ident, st)
    pk = sg.make_stream(setup, pattern, 48, seed=31, p_floor_unused=0.1)
    ch = setup.channels
    mode_bits = max(0, (len(setup.modes) - 1).bit_length())

    def block_n(p):
        mode = (int.from_bytes(bytes(p[:2]) + b"\0\0", "little") >> 1) & ((1 << mode_bits) - 1)
        return 1 << (setup.bs1 if setup.modes[mode].blockflag else setup.bs0)

    outs = []
    for on in (False, True):
        pwr = audio.PreviousWindowRight()
        b = Batch(dec, len(pk), "f32")
        if on:
            assert b.set_entropy_on_device(True)
        b.entropy([(p, pwr) for p in pk], n_threads=2)
        b.upload()
        flat = b.synth_to_host()
        res = b.results()
        taps = [b.tap(i, N.TAP_RESIDUE_PRE_INVERSE, ch, block_n(pk[i])).copy() for i in range(len(pk)) if res[i][0] == 0]
        outs.append((res, flat.copy(), taps))
        b.close()
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][1].view(np.uint32), outs[1][1].view(np.uint32))
    assert len(outs[0][2]) == len(outs[1][2]) > 0
    for a, c in zip(outs[0][2], outs[1][2]):
        assert np.array_equal(a.view(np.uint32), c.view(np.uint32))


def test_ineligible_streams_stay_on_the_host_stage():
    from lewton_amd import _native as N
    from lewton_amd.ring import Ring
    import ctypes as C
    for setup, word in ((FLOOR0_SETUPS["floor0"](), b"floor type 0"), (FLOOR0_SETUPS["floor0_mixed"](), b"floor type 0")):
        audio, ident, st = _product(setup)
        dec = audio.decoder_for(ident, st)
        ring = Ring(dec, 2, 16, "i16")
        assert ring.set_entropy_on_device(True) is False
        why = C.c_char_p()
        assert N.lw_decoder_supports_device_entropy(dec._h, C.byref(why)) == 0 and word in why.value
        ring.close()


def test_sharder_with_device_entropy_matches_oracle():
    """lw_sharder_set_entropy_on_device: three logical shards, every shard's k_entropy and synthesis kernels on its own
    stream; mixed short / long streams with damaged packets, two calls with the state carried between them"""
    from lewton_amd import shard
    setup = SETUPS["stereo"]()
    audio, ident, st = _product(setup)
    o_id, o_st = oracle_headers(setup)
    n_streams, per = 50, 6
    sh = shard.Sharder(ident, st, [0, 0, 0], max_packets_per_shard=n_streams * per, samples="i16")
    assert sh.set_entropy_on_device(True)
    rng = np.random.default_rng(5)
    streams = [_damage(sg.make_stream(setup, "LLSSL", 2 * per, seed=300 + s, p_floor_unused=0.1), rng) for s in range(n_streams)]
    opws = [po.Pwr() for _ in range(n_streams)]
    for half in range(2):
        items = [(s, streams[s][half * per + t]) for t in range(per) for s in range(n_streams)]
        blocks, res = sh.decode(items, n_threads=2)
        for (s, pkt), b, r in zip(items, blocks, res):
            try:
                want = po.read_audio_packet(o_id, o_st, pkt, opws[s], "i16")
                rc = 0
            except po.OracleError as e:
                rc = e.code
            assert r[0] == rc, (half, s)
            if rc == 0:
                assert b.shape == want.shape and np.array_equal(b, want), (half, s)
    sh.close()


def test_real_file_through_the_device_entropy_stage():
    """tests/golden/invalid_keypress.ogg (a real encoder's setup header: 40-odd books, some with codes longer than the table
    levels): eligible, and every packet -- also cut and bit-flipped copies -- equals the oracle's samples"""
    import os
    from common import ROOT
    from lewton_amd import audio, header
    from lewton_amd.ring import Ring
    from oracle import pyogg
    rd = pyogg.PacketReader(open(os.path.join(ROOT, "tests", "golden", "invalid_keypress.ogg"), "rb").read())
    pk = []
    while True:
        p = rd.read_packet()
        if p is None:
            break
        pk.append(bytes(p.data))
    ident = header.read_header_ident(pk[0])
    st = header.read_header_setup(pk[2], ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    o_id = po.Ident(pk[0])
    o_st = po.Setup(pk[2], o_id)
    dec = audio.decoder_for(ident, st)
    rng = np.random.default_rng(3)
    n_streams = 12
    streams = [pk[3:]] + [_damage(pk[3:], rng) for _ in range(n_streams - 1)]
    ring = Ring(dec, 2, n_streams * len(pk), "i16")
    assert ring.set_entropy_on_device(True)
    pwrs = [audio.PreviousWindowRight() for _ in range(n_streams)]
    items = [(p, s) for s in range(n_streams) for p in streams[s]]
    ring.submit(ring.marshal([(p, pwrs[s]) for p, s in items]), n_threads=1)
    assert "k_entropy" in ring.last_kernels
    res, pcm = ring.collect()
    opws = [po.Pwr() for _ in range(n_streams)]
    ch = ident.audio_channels
    ok = 0
    for (pkt, s), (status, m, off) in zip(items, res):
        try:
            want = po.read_audio_packet(o_id, o_st, pkt, opws[s], "i16")
            rc = 0
        except po.OracleError as e:
            rc = e.code
        assert status == rc
        if rc == 0:
            ok += 1
            assert np.array_equal(pcm[off:off + m * ch], want.reshape(-1))
    assert ok > 200
    ring.release()
    ring.close()
