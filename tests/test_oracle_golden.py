"""Pin the CPU oracle against every known answer the reference's own tests hold for this path
(tests/golden/reference_vectors.json <- tests/golden/make_golden.py <- /root/reference/src)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as po

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))
L = po.lib()


def _mismatches(a, b, eps):
    # fuzzy_compare_array, src/imdct_test.rs:992-1005: mismatch iff |a-b| >= eps
    return int(np.sum(np.abs(np.asarray(a, np.float32) - np.asarray(b, np.float32)) >= np.float32(eps)))


@pytest.mark.parametrize("k,tol", [(1, G["imdct_tolerance"]), (2, G["imdct_tolerance"]), (3, 1e-3)])
def test_imdct_golden(k, tol):
    # src/imdct.rs:833 (`test_imdct`) uses ARR_1 at 5e-5 with 0 mismatches. ARR_2 is unused by the reference
    # but passes the same bar; ARR_3 (n=2048) has inputs printed to 5 decimals only -> 1e-3.
    x = np.array(G["imdct"]["IMDCT_INPUT_TEST_ARR_%d" % k], np.float32)
    want = np.array(G["imdct"]["IMDCT_OUTPUT_TEST_ARR_%d" % k], np.float32)
    bs = int(np.log2(len(want)))
    got = po.inverse_mdct(x, bs)
    assert _mismatches(got, want, tol) == 0


@pytest.mark.parametrize("k", [1, 2])
def test_imdct_slow_golden(k):
    # src/audio.rs:829 `test_imdct_slow`
    x = np.array(G["imdct"]["IMDCT_INPUT_TEST_ARR_%d" % k], np.float32)
    want = np.array(G["imdct"]["IMDCT_OUTPUT_TEST_ARR_%d" % k], np.float32)
    assert _mismatches(po.inverse_mdct_slow(x), want, G["imdct_tolerance"]) == 0


@pytest.mark.parametrize("bs", [6, 7, 8, 9, 10, 11, 12, 13])
def test_imdct_fast_vs_definition(bs):
    # definitional cross-check against an f64 DCT-IV based IMDCT (audio.rs:792-825 in double precision).
    # bs 6/7: lewton's literal stage structure runs extra butterfly stages (SURVEY 8c caveat) -> not compared.
    n = 1 << bs
    rng = np.random.default_rng(bs)
    x = (rng.standard_normal(n // 2) * 0.1).astype(np.float32)
    got = po.inverse_mdct(x, bs).astype(np.float64)
    m = n // 2
    i = np.arange(m)
    dct = np.cos(np.pi / (4 * m) * np.outer(2 * i + 1, 2 * i + 1)) @ x.astype(np.float64)
    n4, n34 = n // 4, n - n // 4
    want = np.empty(n)
    want[:n4] = dct[n4:n4 + n4]
    want[n4:n34] = -dct[::-1][: n34 - n4] if False else -dct[n34 - np.arange(n4, n34) - 1]
    want[n34:] = -dct[np.arange(n34, n) - n34]
    if bs >= 8:
        assert np.max(np.abs(got - want)) < 2e-5 * max(1.0, np.max(np.abs(want)))
    # output symmetries hold for every size
    n2 = n // 2
    assert np.array_equal(got[:n4], -got[n2 - 1:n4 - 1:-1])
    assert np.array_equal(got[n2:n2 + n4], got[n - 1:n2 + n4 - 1:-1])


def test_bitreverse_bs8():
    assert list(po.tables(8)[4]) == G["bitreverse_bs8"]


def test_tables_match_formulas():
    # header_cached.rs:43-99 evaluated independently in numpy float32 with glibc-equivalent f32 sin/cos.
    for bs in (6, 8, 11, 13):
        n = 1 << bs
        A, B, Ct, W, br = po.tables(bs)
        k = np.arange(n // 4, dtype=np.float32)
        p4 = np.float32(np.float32(4.0) * np.float32(np.pi)) / np.float32(n)
        # tolerance 1 ulp-ish: numpy's f32 cos may differ from glibc cosf in the last bit
        assert np.allclose(A[0::2], np.cos((k * p4).astype(np.float32)), atol=2e-7)
        assert np.allclose(A[1::2], -np.sin((k * p4).astype(np.float32)), atol=2e-7)
        assert np.all(W > 0) and np.all(np.diff(W) >= 0) and W[-1] <= 1.0
        # power complementarity of the Vorbis window: w[i]^2 + w[n/2-1-i]^2 == 1
        assert np.allclose(W.astype(np.float64) ** 2 + W[::-1].astype(np.float64) ** 2, 1.0, atol=1e-6)
        assert len(br) == n // 8 and len(set(br.tolist())) == n // 8 and int(br.max()) == n // 2 - 4


def test_render_point():
    for c in G["render_point"]:
        assert L.lwo_render_point(*c["args"]) == c["want"]


def test_neighbors():
    for c in G["neighbors"]:
        v = (C.c_uint32 * len(c["v"]))(*c["v"])
        idx, val = C.c_size_t(0), C.c_uint32(0)
        fn = L.lwo_low_neighbor if c["kind"] == "low" else L.lwo_high_neighbor
        rc = fn(v, c["x"], C.byref(idx), C.byref(val))
        if c.get("panics"):
            assert rc == -1
        else:
            assert rc == 0 and (idx.value, val.value) == (c["idx"], c["val"])


def test_ilog_lookup1_float32():
    for v, want in G["ilog"]:
        assert L.lwo_ilog(v) == want
    for e, d, want in G["lookup1_values"]:
        assert L.lwo_lookup1_values(e, d) == want
    for v, want in G["float32_unpack"]:
        assert L.lwo_float32_unpack(v) == want


def test_bitreader():
    for c in G["bitreader"]:
        data = bytes(c["data"])
        widths = (C.c_uint8 * len(c["reads"]))(*[w for w, _ in c["reads"]])
        vals = (C.c_uint64 * len(c["reads"]))()
        ok = L.lwo_bitread_seq(data, len(data), widths, len(c["reads"]), vals)
        assert ok == len(c["reads"])
        for (w, want), got in zip(c["reads"], vals):
            if want is not None:
                assert got == want, (c, got)
    # reads past the end fail without advancing (bitpacking.rs:110-113,137-140)
    widths = (C.c_uint8 * 3)(7, 2, 1)
    vals = (C.c_uint64 * 3)()
    assert L.lwo_bitread_seq(b"\xff", 1, widths, 3, vals) == 2 and vals[0] == 127 and vals[2] == 1


def _pack_codewords(cws):
    # codeword bits are emitted root-first; every bit goes into the stream LSb-first
    bits = []
    for path, ln in cws:
        bits += [(path >> (ln - 1 - i)) & 1 for i in range(ln)]
    out = bytearray((len(bits) + 7) // 8)
    for i, b in enumerate(bits):
        out[i >> 3] |= b << (i & 7)
    return bytes(out), len(bits)


def test_huffman():
    for h in G["huffman"]:
        lengths = (C.c_uint8 * len(h["lengths"]))(*h["lengths"])
        cws = h["codewords"]
        data, nbits = _pack_codewords([(p, l) for p, l, _ in cws])
        syms = (C.c_uint32 * (len(cws) + 8))()
        n = C.c_size_t(0)
        rc = L.lwo_huffman_check(lengths, len(h["lengths"]), data, len(data), syms, len(cws), C.byref(n))
        if h["valid"] is True:
            assert rc == 0, h
            assert [syms[i] for i in range(n.value)] == [v for _, _, v in cws]
        elif h["valid"] is False:
            assert rc != 0, h


def test_ident_header():
    pkt = bytes(G["ident_header"]["packet"])
    idh = po.Ident(pkt)
    for k, v in G["ident_header"]["fields"].items():
        assert getattr(idh, k) == v
    with pytest.raises(po.OracleError) as e:
        po.Ident(bytes(G["ident_header_bad_capture"]))
    assert e.value.code == po.HDR_NOT_VORBIS


def test_sample_conversion_and_couple():
    # samples.rs:92-103
    f = L.lwo_sample_i16
    assert f(0.0) == 0 and f(1.0) == 32767 and f(-1.0) == -32768 and f(-1.5) == -32768 and f(2.0) == 32767
    assert f(0.5) == 16384 and f(-0.5) == -16384
    assert f(np.float32(100.9 / 32768.0)) == 100 and f(np.float32(-100.9 / 32768.0)) == -100  # toward zero
    assert f(float("nan")) == 0
    # audio.rs:763-777
    nm, na = C.c_float(), C.c_float()
    for m, a, want in [(2.0, 1.0, (2.0, 1.0)), (2.0, -1.0, (1.0, 2.0)), (-2.0, 1.0, (-2.0, -1.0)),
                       (-2.0, -1.0, (-1.0, -2.0)), (0.0, 3.0, (0.0, 3.0)), (0.0, -3.0, (3.0, 0.0)), (1.0, 0.0, (1.0, 1.0))]:
        L.lwo_inverse_couple(m, a, C.byref(nm), C.byref(na))
        assert (nm.value, na.value) == want


def test_render_line_closed_form():
    # y(x0+t) = y0 + sgn(dy)*floor(t*|dy|/adx) reproduces the error-accumulating loop of audio.rs:503-524
    rng = np.random.default_rng(1)
    buf = (C.c_uint32 * 9000)()
    for _ in range(3000):
        x0 = int(rng.integers(0, 4000)); adx = int(rng.integers(1, 4097)); y0, y1 = (int(v) for v in rng.integers(0, 256, 2))
        n = L.lwo_render_line(x0, y0, x0 + adx, y1, buf)
        assert n == adx
        t = np.arange(adx)
        want = y0 + np.sign(y1 - y0) * ((t * abs(y1 - y0)) // adx)
        assert np.array_equal(np.frombuffer(buf, np.uint32, n).astype(np.int64), want)


def test_inverse_db_table():
    t = po.inverse_db_table()
    assert t[0] == np.float32(1.0649863e-07) and t[255] == 1.0 and np.all(np.diff(t) > 0)
    # spec section 10.1: the table is exp(ln(1e-7-ish) ...) geometric: ratio is constant to print precision
    r = t[1:] / t[:-1]
    assert np.allclose(r, r.mean(), rtol=2e-6)


@pytest.mark.parametrize("bs", [8, 11])
def test_tdac_reconstruction(bs):
    """Algebraic cross-check for what no in-tree vector of the reference pins (SURVEY 8c iii): a forward MDCT written here
    in float64 (standard phase i + 1/2 + n/4), the oracle's inverse transform and window table, and the overlap-add rule
    of audio.rs:1116-1118 reconstruct the input signal (time-domain aliasing cancellation) -- scale n/4 of the
    un-normalised transform pair, window power-complementary, halves aligned."""
    n = 1 << bs
    m = n // 2
    rng = np.random.default_rng(bs)
    blocks = 6
    x = rng.standard_normal((blocks + 1) * m)
    _A, _B, _C, W, _br = po.tables(bs)
    w_full = np.concatenate([W, W[::-1]]).astype(np.float64)
    i = np.arange(n)[:, None]
    k = np.arange(m)[None, :]
    basis = np.cos(np.pi / m * (i + 0.5 + m / 2.0) * (k + 0.5))          # [n][m]
    prev_right = None
    worst = 0.0
    for b in range(blocks):
        seg = x[b * m: b * m + n] * w_full
        X = seg @ basis                                                     # forward MDCT, m coefficients
        y = po.inverse_mdct((X * (2.0 / m)).astype(np.float32), bs).astype(np.float64)
        left, right = y[:m], y[m:]
        if prev_right is not None:
            out = left * W + prev_right * W[::-1]                           # out[i] = cur[i] w[i] + prev[i] w[m-1-i]
            worst = max(worst, float(np.max(np.abs(out - x[b * m:(b + 1) * m]))))
        prev_right = right
    assert worst < 5e-6 * np.max(np.abs(x)), worst
