"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Bar: i16 PCM bit-exact; f32 taps within 1e-5 (north_star) -- in practice bit-identical.

Nothing here reads /root/reference; golden vectors come from tests/golden/reference_vectors.json."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from common import FLOOR0_SETUPS, SETUPS, oracle_headers, po, sg

pytestmark = pytest.mark.gpu

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))
F32_TOL = 1e-5  # BASELINE.json north_star: "f32 intermediates within 1e-5"


def _product(setup):
    from lewton_amd import audio, header
    idp, _, stp = setup.headers()
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    return audio, ident, st


def test_native_library_is_loaded():
    from lewton_amd import _native as N
    assert N.lw_device_count() >= 1, N.device_error()
    maps = open("/proc/self/maps").read()
    assert "liblewton_amd.so" in maps


@pytest.mark.parametrize("k,tol", [(1, 5e-5), (2, 5e-5), (3, 1e-3)])
def test_device_imdct_golden(k, tol):
    # src/imdct.rs:833 on the device; ARR_3 is the reference's (unused) n=2048 vector, inputs printed to 5 decimals
    from lewton_amd import _native as N
    audio, ident, st = _product(SETUPS["stereo"]())
    dec = audio.decoder_for(ident, st)
    x = np.array(G["imdct"]["IMDCT_INPUT_TEST_ARR_%d" % k], np.float32)
    want = np.array(G["imdct"]["IMDCT_OUTPUT_TEST_ARR_%d" % k], np.float32)
    out = np.zeros(len(want), np.float32)
    rc = N.lw_debug_imdct(dec._h, 1 if len(want) == 2048 else 0, x.ctypes.data_as(N.f32p), out.ctypes.data_as(N.f32p))
    assert rc == 0, N.device_error()
    assert int(np.sum(np.abs(out - want) >= np.float32(tol))) == 0
    # and bit-identical to the oracle's sequential transform
    assert np.array_equal(out.view(np.uint32), po.inverse_mdct(x, int(np.log2(len(want)))).view(np.uint32))


@pytest.mark.parametrize("name", sorted(SETUPS))
def test_device_imdct_random_all_sizes(name):
    from lewton_amd import _native as N
    setup = SETUPS[name]()
    audio, ident, st = _product(setup)
    dec = audio.decoder_for(ident, st)
    rng = np.random.default_rng(4)
    for flag, bs in ((0, setup.bs0), (1, setup.bs1)):
        n = 1 << bs
        x = (rng.standard_normal(n // 2) * 0.2).astype(np.float32)
        out = np.zeros(n, np.float32)
        assert N.lw_debug_imdct(dec._h, flag, x.ctypes.data_as(N.f32p), out.ctypes.data_as(N.f32p)) == 0
        assert np.array_equal(out.view(np.uint32), po.inverse_mdct(x, bs).view(np.uint32)), (name, bs)


PATTERNS = {"stereo": "LLSSSSSSSSL", "stereo_t1": "LSLLS", "surround51": "LLSSL", "mono_small": "LSSLLSL",
            "stereo_9_12": "LLSL", "stereo_6_13": "LSSL", "stereo_7_7": "LSLL",
            "stereo_9_10": "LLLSSSLLLL", "stereo_8_10": "LLLLSSSSSL", "stereo_8_9": "LLLSSL", "stereo_10_12": "LSSSSL",

            "floor0": "LLSSL", "floor0_mixed": "LSLLS", "floor0_8_11": "LLLSL"}
ALL_SETUPS = dict(SETUPS, **FLOOR0_SETUPS)   # floor-0 streams: curve from the host stage, multiply + IMDCT on the GPU


@pytest.mark.parametrize("name", sorted(ALL_SETUPS))
@pytest.mark.parametrize("fmt", ["i16", "f32", "i16_interleaved"])
def test_read_audio_packet_matches_oracle(name, fmt):
    """Drop-in call (audio.rs:919/1170), packet by packet with state carry, incl. unused floors and truncated packets."""
    setup = ALL_SETUPS[name]()
    audio, ident, st = _product(setup)
    o_id, o_st = oracle_headers(setup)
    pkts = sg.make_stream(setup, PATTERNS[name], 30, seed=21, p_floor_unused=0.1)
    rng = np.random.default_rng(8)
    pwr, o_pwr = audio.PreviousWindowRight(), po.Pwr()
    assert pwr.is_empty()
    ofmt = {"i16": "i16", "f32": "f32", "i16_interleaved": "i16_itl"}[fmt]
    for i, p in enumerate(pkts):
        if i % 7 == 6:
            p = p[: max(1, int(rng.integers(1, max(2, len(p)))))]
        try:
            want = po.read_audio_packet(o_id, o_st, p, o_pwr, ofmt)
            o_rc = 0
        except po.OracleError as e:
            o_rc = e.code
        try:
            got = audio.read_audio_packet_generic(ident, st, p, pwr, fmt)
            rc = 0
        except audio.AudioReadError as e:
            rc = e.code
        assert rc == o_rc, (i, rc, o_rc)
        assert pwr.is_empty() == o_pwr.is_empty()
        if rc:
            continue
        assert got.shape == want.shape, (i, got.shape, want.shape)
        if fmt == "f32":
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (i, np.abs(got - want).max())
        else:
            assert np.array_equal(got, want), i
        st_o = o_pwr.data(setup.channels)
        assert np.array_equal(pwr.data().view(np.uint32), st_o.view(np.uint32))


def test_window_mismatch_error_semantics():
    """audio.rs:1107-1111: a long stored right part meeting a short block -> AudioBadFormat and pwr left empty."""
    setup = SETUPS["stereo"]()
    audio, ident, st = _product(setup)
    o_id, o_st = oracle_headers(setup)
    pw = sg.PacketWriter(setup, 5)
    seq = [pw.packet(1, 1, 1), pw.packet(1, 1, 1), pw.packet(0), pw.packet(1, 0, 1), pw.packet(1, 1, 1)]
    pwr, o_pwr = audio.PreviousWindowRight(), po.Pwr()
    codes = []
    for p in seq:
        try:
            want = po.read_audio_packet(o_id, o_st, p, o_pwr, "i16")
            o_rc = 0
        except po.OracleError as e:
            o_rc = e.code
        try:
            got = audio.read_audio_packet(ident, st, p, pwr)
            rc = 0
        except audio.AudioReadError as e:
            rc = e.code
        codes.append(rc)
        assert rc == o_rc and pwr.is_empty() == o_pwr.is_empty()
        if rc == 0:
            assert np.array_equal(got, want)
    assert codes[2] == 2 and codes[3] == 0  # AudioBadFormat, then a fresh start with 0 samples


def test_pwr_clone_and_reset():
    setup = SETUPS["stereo"]()
    audio, ident, st = _product(setup)
    pkts = sg.make_stream(setup, "L", 6, seed=2)
    pwr = audio.PreviousWindowRight()
    for p in pkts[:3]:
        audio.read_audio_packet(ident, st, p, pwr)
    twin = pwr.clone()
    a = audio.read_audio_packet(ident, st, pkts[3], pwr)
    b = audio.read_audio_packet(ident, st, pkts[3], twin)
    assert np.array_equal(a, b) and a.shape[1] == 1024
    pwr.reset()  # inside_ogg.rs:307-313
    assert pwr.is_empty()
    assert audio.read_audio_packet(ident, st, pkts[4], pwr).shape[1] == 0
    assert audio.read_audio_packet(ident, st, pkts[5], pwr).shape[1] == 1024


@pytest.mark.parametrize("name", ["stereo", "surround51", "mono_small", "floor0", "floor0_mixed"])
def test_batch_many_streams_matches_oracle(name):
    """Batched decode, several interleaved streams, state carried across two batches, taps at the record_*! points."""
    from lewton_amd import _native as N
    from lewton_amd.batch import Batch
    setup = ALL_SETUPS[name]()
    audio, ident, st = _product(setup)
    o_id, o_st = oracle_headers(setup)
    dec = audio.decoder_for(ident, st)
    ch = setup.channels
    n_streams, per = 5, 12
    streams = [sg.make_stream(setup, PATTERNS[name], per, seed=100 + s, p_floor_unused=0.05) for s in range(n_streams)]
    pwrs = [audio.PreviousWindowRight() for _ in range(n_streams)]
    o_pwrs = [po.Pwr() for _ in range(n_streams)]
    batch = Batch(dec, n_streams * per, "i16")
    for half in range(2):
        items, want = [], []
        for t in range(half * per // 2, (half + 1) * per // 2):
            for s in range(n_streams):  # round-robin interleaving of the streams
                items.append((streams[s][t], pwrs[s]))
                want.append(po.read_audio_packet(o_id, o_st, streams[s][t], o_pwrs[s], "f32", taps=True))
        res = batch.entropy(items, n_threads=2)
        batch.upload()
        flat = batch.synth_to_host()
        got = batch.split(flat, ch)
        for i, ((w, taps), g, r) in enumerate(zip(want, got, res)):
            assert r[0] == 0 and r[1] == w.shape[1]
            wi = np.vectorize(po.lib().lwo_sample_i16, otypes=[np.int16])(w) if w.size else w.astype(np.int16)
            assert np.array_equal(g, wi), (half, i)
        # taps for a few packets
        for i in (0, len(items) // 2, len(items) - 1):
            n = want[i][1]["n"]
            for which, key in ((N.TAP_RESIDUE_PRE_INVERSE, "residue_pre_inverse"), (N.TAP_RESIDUE_POST_INVERSE, "residue_post_inverse"),
                               (N.TAP_PRE_MDCT, "pre_mdct"), (N.TAP_POST_MDCT, "post_mdct")):
                t = batch.tap(i, which, ch, n)
                ref = want[i][1][key]
                assert np.max(np.abs(t - ref)) <= F32_TOL
                assert np.array_equal(t.view(np.uint32), ref.view(np.uint32)), key
        for s in range(n_streams):
            assert np.array_equal(pwrs[s].data().view(np.uint32), o_pwrs[s].data(ch).view(np.uint32))
    assert any(k in batch.last_kernels for k in ("k_imdct_generic", "k_long", "k_short"))


def _decode_batch(setup, items_streams, fmt="i16", force_generic=False, batch=None, rounds=0):
    """items_streams: list of (packet, stream_index). Returns (per-packet outputs, batch, pwrs)."""
    from lewton_amd.batch import Batch
    audio, ident, st = _product(setup)
    dec = audio.decoder_for(ident, st)
    n_streams = 1 + max(s for _, s in items_streams)
    pwrs = [audio.PreviousWindowRight() for _ in range(n_streams)]
    b = batch or Batch(dec, len(items_streams), fmt)
    b.set_force_generic(force_generic)
    b.debug_set_rounds(rounds)
    b.entropy([(p, pwrs[s]) for p, s in items_streams], n_threads=2)
    b.upload()
    return b.split(b.synth_to_host(), setup.channels), b, pwrs


@pytest.mark.parametrize("name,pattern,count", [
    ("stereo", "L", 100), ("stereo", "LLLLLSLLLLLLLLLLLLLLLLLLLSSLLLLL", 90), ("surround51", "L", 40),
    ("surround51", "LLLLLLLSLLLLL", 40), ("stereo_t1", "LLLS", 50)])
@pytest.mark.parametrize("fmt", ["i16", "f32", "i16_interleaved"])
def test_long_block_kernel_matches_oracle(name, pattern, count, fmt):
    """The specialised n=2048 kernel: runs crossing workgroup boundaries (halo pre-pass), generic<->fast hand-over
    around short blocks, unused floors, all three sample formats."""
    setup = SETUPS[name]()
    o_id, o_st = oracle_headers(setup)
    pkts = sg.make_stream(setup, pattern, count, seed=31, p_floor_unused=0.08)
    got, b, pwrs = _decode_batch(setup, [(p, 0) for p in pkts], fmt)
    assert "k_long" in b.last_kernels
    o_pwr = po.Pwr()
    ofmt = {"i16": "i16", "f32": "f32", "i16_interleaved": "i16_itl"}[fmt]
    for i, p in enumerate(pkts):
        want = po.read_audio_packet(o_id, o_st, p, o_pwr, ofmt)
        g = got[i]
        assert g.shape == want.shape, (i, g.shape, want.shape)
        if fmt == "f32":
            assert np.array_equal(g.view(np.uint32), want.view(np.uint32)), (i, np.abs(g - want).max())
        else:
            assert np.array_equal(g, want), i
    assert np.array_equal(pwrs[0].data().view(np.uint32), o_pwr.data(setup.channels).view(np.uint32))


def test_long_block_kernel_equals_generic_kernels_many_streams():
    """Same batch through the specialised and the generic kernels: identical PCM; streams interleaved round-robin,
    state carried over three consecutive batches."""
    setup = SETUPS["stereo"]()
    n_streams, per = 37, 9
    streams = [sg.make_stream(setup, "L", 3 * per, seed=500 + s) for s in range(n_streams)]
    from lewton_amd.batch import Batch
    audio, ident, st = _product(setup)
    dec = audio.decoder_for(ident, st)
    pw_a = [audio.PreviousWindowRight() for _ in range(n_streams)]
    pw_b = [audio.PreviousWindowRight() for _ in range(n_streams)]
    ba, bb = Batch(dec, n_streams * per, "i16"), Batch(dec, n_streams * per, "i16")
    bb.set_force_generic(True)
    for r in range(3):
        items = [(streams[s][r * per + t], s) for t in range(per) for s in range(n_streams)]
        ba.entropy([(p, pw_a[s]) for p, s in items]); ba.upload(); fa = ba.synth_to_host()
        bb.entropy([(p, pw_b[s]) for p, s in items]); bb.upload(); fb = bb.synth_to_host()
        assert "k_long" in ba.last_kernels and "k_long" not in bb.last_kernels
        assert fa.shape == fb.shape and np.array_equal(fa, fb), r
    for s in range(n_streams):
        assert np.array_equal(pw_a[s].data().view(np.uint32), pw_b[s].data().view(np.uint32))


@pytest.mark.parametrize("rounds", [1, 2, 3, 5])
@pytest.mark.parametrize("name", ["stereo", "surround51"])
def test_long_block_kernel_rounds_and_handover(name, rounds):
    """The specialised kernel with a forced number of rounds per workgroup: right halves travel through LDS inside a
    round, across rounds, through the halo pre-pass at chunk boundaries and through the state pool; streams of
    different lengths so that chunks start and end in the middle of streams.  Bit-exact vs the oracle per stream."""
    setup = SETUPS[name]()
    lens = [1, 2, 5, 16, 17, 33, 40, 7]
    streams = [sg.make_stream(setup, "L", n + 1, seed=900 + i) for i, n in enumerate(lens)]
    items = [(p, s) for s, st in enumerate(streams) for p in st]
    got, b, _ = _decode_batch(setup, items, "i16", rounds=rounds)
    assert "k_long" in b.last_kernels
    o_id, o_st = oracle_headers(setup)
    k = 0
    for s, st in enumerate(streams):
        opw = po.Pwr()
        for p in st:
            want = po.read_audio_packet(o_id, o_st, p, opw, "i16")
            assert got[k].shape == want.shape and np.array_equal(got[k], want), (s, k)
            k += 1


def test_full_size_batch_properties():
    """BASELINE configs[1] size (4096 stereo long packets): checksum-of-checksums equality between the two kernel
    families, plus idempotence of re-launching an uploaded batch."""
    setup = SETUPS["stereo"]()
    pool = sg.make_stream(setup, "L", 64, seed=77)
    rng = np.random.default_rng(3)
    pkts = [pool[int(i)] for i in rng.integers(0, 64, 4096)]
    got_f, bf, _ = _decode_batch(setup, [(p, 0) for p in pkts], "i16")
    got_g, bg, _ = _decode_batch(setup, [(p, 0) for p in pkts], "i16", force_generic=True)
    fa = np.concatenate([g.reshape(-1) for g in got_f])
    ga = np.concatenate([g.reshape(-1) for g in got_g])
    assert fa.size == 4095 * 2 * 1024
    assert np.array_equal(fa, ga)
    again = bf.synth_to_host()
    assert np.array_equal(again, fa)
    assert bf.algorithmic_bytes == 4096 * (8192 + 132) + 4095 * 4096


@pytest.mark.parametrize("name,pattern", [("stereo", "LLLLLLSSLLLSLLLLLLLLLLLLLLLLLLLLLLLLLSSSSL"), ("surround51", "LLLSLLLLLLLLLLLLLLLLSSL")])
@pytest.mark.parametrize("fmt", ["i16", "f32"])
def test_long_mixed_single_stream_in_one_batch(name, pattern, fmt):
    """ONE stream, 700 packets in one batch: chunks of the specialised kernel joined by the halo pre-pass, LDS hand-over
    inside the chunks, long blocks with short slopes split between k_long (everything but the 128-sample overlap) and
    k_short (the short blocks and that overlap) -- against the oracle, packet by packet."""
    from lewton_amd.batch import Batch
    setup = ALL_SETUPS[name]()
    audio, ident, st = _product(setup)
    o_id, o_st = oracle_headers(setup)
    dec = audio.decoder_for(ident, st)
    ch = setup.channels
    pk = sg.make_stream(setup, pattern, 700, seed=77, p_floor_unused=0.02)
    pwr, o_pwr = audio.PreviousWindowRight(), po.Pwr()
    b = Batch(dec, len(pk), fmt)
    res = b.entropy([(p, pwr) for p in pk], n_threads=4)
    b.upload()
    got = b.split(b.synth_to_host(), ch)
    assert ("k_mix" in b.last_kernels or ("k_long" in b.last_kernels and "k_short" in b.last_kernels)) and "generic" not in b.last_kernels
    for i, p in enumerate(pk):
        want = po.read_audio_packet(o_id, o_st, p, o_pwr, fmt)
        assert res[i][0] == 0 and got[i].shape == want.shape, i
        if fmt == "f32":
            assert np.array_equal(got[i].view(np.uint32), want.view(np.uint32)), i
        else:
            assert np.array_equal(got[i], want), i
    assert np.array_equal(pwr.data().view(np.uint32), o_pwr.data(ch).view(np.uint32))


@pytest.mark.parametrize("name", sorted(ALL_SETUPS))
@pytest.mark.parametrize("fmt", ["i16", "f32", "i16_interleaved"])
def test_mixed_short_long_streams_both_batch_orders(name, fmt):
    """Runs of short blocks of every length up to 10 between long blocks, six streams interleaved round-robin (every
    packet's predecessor is far away in the batch) and stream-major (runs of consecutive packets), with unused floors and
    packets cut short, against the oracle -- and the product's default path against its generic kernels on the same batch."""
    from lewton_amd.batch import Batch
    setup = ALL_SETUPS[name]()
    audio, ident, st = _product(setup)
    o_id, o_st = oracle_headers(setup)
    dec = audio.decoder_for(ident, st)
    ch = setup.channels
    n_streams, per = 6, 26
    streams = [sg.make_stream(setup, "LSSSSSSSSSSLLSSL", per, seed=40 + s, p_floor_unused=0.06) for s in range(n_streams)]
    for s in range(n_streams):
        streams[s][7 + s] = streams[s][7 + s][: max(4, len(streams[s][7 + s]) // 2)]
    ofmt = {"i16": "i16", "f32": "f32", "i16_interleaved": "i16_itl"}[fmt]
    for order in ("stream_major", "round_robin"):
        if order == "stream_major":
            items = [(streams[s][t], s) for s in range(n_streams) for t in range(per)]
        else:
            items = [(streams[s][t], s) for t in range(per) for s in range(n_streams)]
        outs = {}
        for generic in (False, True):
            pwrs = [audio.PreviousWindowRight() for _ in range(n_streams)]
            b = Batch(dec, len(items), fmt)
            b.set_force_generic(generic)
            res = b.entropy([(p, pwrs[s]) for p, s in items], n_threads=2)
            b.upload()
            outs[generic] = (b.split(b.synth_to_host(), ch), res, b.last_kernels, pwrs)
        got, res, kernels, pwrs = outs[False]
        ref, _res2, kernels2, pwrs2 = outs[True]
        assert "k_long" not in kernels2 and "k_ola_generic" in kernels2, kernels2
        opws = [po.Pwr() for _ in range(n_streams)]
        for i, (p, s) in enumerate(items):
            try:
                want = po.read_audio_packet(o_id, o_st, p, opws[s], ofmt)
                rc = 0
            except po.OracleError as e:
                rc = e.code
            assert res[i][0] == rc, (order, i)
            if rc:
                continue
            for g in (got[i], ref[i]):
                assert g.size == want.size, (order, i)
                if fmt == "f32":
                    assert np.array_equal(g.reshape(-1).view(np.uint32), want.reshape(-1).view(np.uint32)), (order, i, kernels)
                else:
                    assert np.array_equal(g.reshape(-1), want.reshape(-1)), (order, i, kernels)
        for s in range(n_streams):
            assert np.array_equal(pwrs[s].data().view(np.uint32), opws[s].data(ch).view(np.uint32))
            assert np.array_equal(pwrs2[s].data().view(np.uint32), opws[s].data(ch).view(np.uint32))
