"""the stages the reference pins with no in-tree vector -- residue decode, inverse coupling, render_line,
window / overlap-add, whole packets -- rest on the C oracle's restatement.  This test pins the oracle against a SECOND,
independently written decoder (tests/independent_decoder.py: numpy float32, from SURVEY section 9 and the Vorbis I
specification only, sharing no code with oracle/ or the product): a real Ogg/Vorbis file and synthetic streams must come
out bit-identical at the four record_*! taps (src/lib.rs:56-94; audio.rs:988, 1004, 1041, 1054), at the f32 output and in
the window state, packet by packet."""
import os

import numpy as np
import pytest

import independent_decoder as ind
from common import SETUPS, po, sg

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "invalid_keypress.ogg")


def _compare(ident_pkt, setup_pkt, packets, expect_errors=False):
    o_id = po.Ident(ident_pkt)
    o_st = po.Setup(setup_pkt, o_id)
    opw = po.Pwr()
    dec = ind.Decoder(ind.Stream(ident_pkt, setup_pkt))
    n_ok = n_samples = 0
    for i, p in enumerate(packets):
        try:
            want, wt = po.read_audio_packet(o_id, o_st, p, opw, "f32", taps=True)
            o_err = None
        except po.OracleError as e:
            o_err = e.code
        try:
            got, gt = dec.decode(p)
            g_err = None
        except ind.EndOfPacket:
            g_err = po.AUDIO_END_OF_PACKET
        except ind.IsHeader:
            g_err = po.AUDIO_IS_HEADER
        except ind.BadFormat:
            g_err = po.AUDIO_BAD_FORMAT
        assert (o_err is None) == (g_err is None), (i, o_err, g_err)
        if o_err is not None:
            assert expect_errors and o_err == g_err
            continue
        assert gt["n"] == wt["n"], i
        for k in ("residue_pre_inverse", "residue_post_inverse", "pre_mdct", "post_mdct"):
            assert gt[k].shape == wt[k].shape, (i, k)
            assert np.array_equal(gt[k].view(np.uint32), wt[k].view(np.uint32)), (i, k, float(np.abs(gt[k] - wt[k]).max()))
        assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), i
        st = opw.data(dec.s.ch)
        assert (st is None) == (dec.prev is None)
        if st is not None:
            assert np.array_equal(st.view(np.uint32), dec.prev.view(np.uint32)), i
        n_ok += 1
        n_samples += got.shape[1]
    return n_ok, n_samples


def test_real_file_bit_identical_at_every_tap():
    pk = ind.demux_ogg(open(GOLDEN, "rb").read())
    assert pk[0][0][:7] == b"\x01vorbis" and pk[1][0][:7] == b"\x03vorbis" and pk[2][0][:7] == b"\x05vorbis"
    n_ok, n_samples = _compare(pk[0][0], pk[2][0], [p[0] for p in pk[3:]])
    assert n_ok == 26 and n_samples == 0 + 3 * 128 + 576 + 21 * 1024      # the plausibility figures of this file


@pytest.mark.parametrize("name,pattern,count", [("stereo", "LLSSLSLLLSSSSLL", 45), ("stereo_t1", "LSLLSSL", 30),
                                                 ("surround51", "LLSLSSLL", 24), ("stereo_9_12", "LSSLL", 12)])
def test_synthetic_streams_bit_identical_at_every_tap(name, pattern, count):
    setup = SETUPS[name]()
    idp, _cmt, stp = setup.headers()
    pk = sg.make_stream(setup, pattern, count, seed=91, p_floor_unused=0.08)
    rng = np.random.default_rng(3)
    for k in range(5, count, 7):                       # packets that end inside the floor / residue (audio.rs:655-660)
        pk[k] = pk[k][: max(2, int(rng.integers(2, len(pk[k]))))]
    n_ok, n_samples = _compare(idp, stp, pk, expect_errors=True)
    assert n_ok >= count - 8 and n_samples > 0


def test_independent_transform_matches_reference_vectors():
    """the numpy IMDCT against the reference's own vectors (imdct_test.rs; tests/golden/reference_vectors.json)"""
    import json
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))
    for k, tol in ((1, 5e-5), (2, 5e-5), (3, 1e-3)):
        x = np.array(G["imdct"]["IMDCT_INPUT_TEST_ARR_%d" % k], np.float32)
        want = np.array(G["imdct"]["IMDCT_OUTPUT_TEST_ARR_%d" % k], np.float32)
        n = len(want)
        got = ind.imdct(x[: n // 2], ind.Tables(n.bit_length() - 1))
        assert int(np.sum(np.abs(got - want) >= np.float32(tol))) == 0
