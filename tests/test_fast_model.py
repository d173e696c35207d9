"""CPU tests of the specialised long-block kernel's design: the numpy lane model (tests/fast_model.py)
fed with the product's LDS table image must reproduce the oracle bit for bit."""
import ctypes as C

import numpy as np
import pytest

import fast_model as fm
from common import SETUPS, floor_from_record, floor_x_sorted, oracle_headers, po, sg
from lewton_amd import _native as N
from lewton_amd import audio, header

NAMES = ["apair", "tw_s2", "tw_l0", "tw_l1", "tw_l2", "tw_l3", "tw_l4", "a2", "c4", "b_lo", "b_hi", "win", "inv_db",
         "xsf", "sid16", "total"]


def _image(setup):
    idp, _, stp = setup.headers()
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    offs = (C.c_uint32 * 16)()
    size = N.lw_debug_fast_image(ident._h, st._h, None, 0, offs)
    if size == 0:
        return None, None, ident, st
    buf = (C.c_uint8 * size)()
    N.lw_debug_fast_image(ident._h, st._h, buf, size, offs)
    return bytes(buf), dict(zip(NAMES, list(offs))), ident, st


def test_fast_image_eligibility():
    assert _image(SETUPS["stereo"]())[0] is not None
    assert _image(SETUPS["surround51"]())[0] is not None
    assert _image(SETUPS["stereo_9_12"]())[0] is None  # blocksize_1 != 11 -> generic kernels
    blob, offs, _, _ = _image(SETUPS["stereo"]())
    assert offs["total"] == len(blob) and len(blob) < 28 * 1024
    assert all(offs[k] % 16 == 0 for k in NAMES)


def test_lane_model_imdct_bit_exact():
    blob, offs, _, _ = _image(SETUPS["stereo"]())
    img = fm.Image(blob, offs)
    rng = np.random.default_rng(0)
    for trial in range(4):
        x = (rng.standard_normal(1024) * (0.3 if trial else 1e-20)).astype(np.float32)
        if trial == 3:
            x[rng.integers(0, 1024, 900)] = 0.0
        got = fm.imdct_wave(x, img)
        want = po.inverse_mdct(x, 11)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_lds_slot_functions_are_bijections_and_conflict_free():
    p = np.arange(512)
    for f in (fm.slot_t2, fm.slot_t3, fm.slot_t4):
        assert sorted(f(p).tolist()) == list(range(512))
    lam = np.arange(64)
    # reads of one ds_read_b64 instruction are serviced in two 32-lane halves over 32 bank pairs
    for half in (lam[:32], lam[32:]):
        for y in range(8):  # T2 read (layout C)
            assert len(set((fm.slot_t2(64 * (half >> 3) + 8 * y + (half & 7)) % 32).tolist())) == 32
        for z in range(8):  # T3 read (layout D)
            assert len(set((fm.slot_t3(8 * half + z) % 32).tolist())) == 32
        for c in range(2):  # T4 gather
            q2 = 2 * fm.rev_bits(2 * half + c, 7)
            for pp in (q2, q2 + 256, 255 - q2, 511 - q2):
                assert len(set((fm.slot_t4(pp) % 32).tolist())) == 32


def test_lane_model_floor_matches_oracle():
    setup = SETUPS["stereo"]()
    blob, offs, ident, st = _image(setup)
    img = fm.Image(blob, offs)
    o_id, o_st = oracle_headers(setup)
    pkts = sg.make_stream(setup, "L", 12, seed=4, p_floor_unused=0.0)
    pwr = po.Pwr()
    for p in pkts:
        _, taps = po.read_audio_packet(o_id, o_st, p, pwr, "f32", taps=True)
        rec = audio.entropy_decode_host(ident, st, p)
        for c in range(2):
            xs = floor_x_sorted(setup, rec["mode"], c)
            a = fm.floor_lane_model(rec["floor"][c], xs, img.inv_db)
            b = floor_from_record(rec["floor"][c], xs, 1024, img.inv_db)
            assert np.array_equal(a, b)
            spec = a * taps["residue_post_inverse"][c]
            assert np.array_equal(spec.view(np.uint32), taps["pre_mdct"][c].view(np.uint32))


def test_floor_division_by_reciprocal_is_exact():
    """trunc((t*dy +- 0.5) * fl(1/adx)) == trunc(t*dy/adx) for EVERY segment the long-block kernel can meet: every
    dy in -255..255, every adx in 1..1024 (and some larger ones) with every offset 0 <= t < min(adx, 1024) inside it --
    exhaustively, and on a reciprocal perturbed by one ulp either way as well (the device uses v_rcp_f32, 1 ulp)."""
    dy = np.arange(-255, 256, dtype=np.int64)[:, None]
    half = np.where(dy >= 0, np.float32(0.5), np.float32(-0.5)).astype(np.float32)
    for adx in list(range(1, 1025)) + [1100, 2048, 4096, 8191, 32768]:
        t = np.arange(0, min(adx, 1024), dtype=np.int64)[None, :]
        z = (t * dy).astype(np.float32) + half                      # exact: |t * dy| < 2^18
        want = np.sign(dy) * ((t * np.abs(dy)) // adx)
        r0 = np.float32(1.0) / np.float32(adx)
        for rinv in (r0, np.nextafter(r0, np.float32(2)), np.nextafter(r0, np.float32(0))):
            q = np.trunc(z * rinv).astype(np.int64)
            assert np.array_equal(q, want), (adx, rinv)


def test_floor_by_two_fmas_and_a_mask_is_exact():
    """k_long's floor_bin: y = ((bits(fma(fma(k, dy, c0), fl(1/adx), 2^21 + 1 + y_base)) & 0x7fc) >> 2) - 1 equals render_line's
    y0 + trunc((k - x0) dy / adx) for EVERY segment the long-block kernel can meet: every dy in -255..255, every adx in
    1..1024 (and some larger ones) with every offset inside it, at both ends of the y range the segment can sit in, and with
    the reciprocal one ulp off either way (v_rcp_f32).  The f64 product z * rinv is exact (<= 22 + 24 bits) and the f64 sum is
    rounded far below the quarter the f32 result is rounded to, so f32(f64 expression) is the fused multiply-add's result
    except on exact quarter ties, which do not touch the integer part (see the kernel's comment)."""
    dy = np.arange(-255, 256, dtype=np.int64)[:, None]
    y0s = (np.where(dy >= 0, 0, -dy), np.where(dy >= 0, 255 - dy, 255))   # the lowest / highest y0 with y0, y0 + dy in 0..255
    for adx in list(range(1, 1025)) + [1100, 2048, 4096]:
        t = np.arange(0, min(adx, 1024), dtype=np.int64)[None, :]
        off = np.sign(dy) * ((t * np.abs(dy)) // adx)
        # (k - x0) = t: c0 carries - x0 dy (resp. - x1 dy), so the inner sum is t dy + 1/2 - adx/8 resp. (adx - t)|dy| + 7 adx/8 - 1/2
        z = (np.where(dy >= 0, t * dy + 0.5, (adx - t) * np.abs(dy) + adx - 0.5) - adx / 8.0).astype(np.float64)
        assert np.array_equal(z.astype(np.float32).astype(np.float64), z)
        r0 = np.float32(1.0) / np.float32(adx)
        for y0 in y0s:
            w = (np.where(dy >= 0, y0, y0 + dy) + 2097153.0).astype(np.float64)
            for rinv in (r0, np.nextafter(r0, np.float32(2)), np.nextafter(r0, np.float32(0))):
                tt = (z * np.float64(rinv) + w).astype(np.float32)
                got = ((tt.view(np.uint32) & 0x7FC) >> 2).astype(np.int64) - 1
                assert np.array_equal(got, y0 + off), (adx, rinv)


def test_floor_by_two_fmas_is_not_exact_for_lines_beyond_4096_bins():
    """why floors with posts beyond x = 4096 (range bits 13 .. 15) are evaluated by k_prep, whose form falls back to the integer
    division there (lw_fast.cpp: plan_units): the line of 32 638 bins from (130, 198) to (32768, 22) that round 6's random-setup
    campaign met (setup 607025) comes out one step off in its first bin -- (adx - t) |dy| + 7 adx / 8 - 1 / 2 needs 25 bits there"""
    f = np.float32
    x0, y0, adx, dy = 130, 198, 32638, -176
    k = np.arange(x0, 1024, dtype=np.int64)
    want = y0 - (176 * (k - x0)) // adx
    c0 = f(f(f(0.875) * f(adx)) - f(0.5)) - f(f(x0 + adx) * f(dy))
    inner = (k.astype(np.float64) * float(dy) + float(c0)).astype(np.float32)
    tt = (inner.astype(np.float64) * float(f(1.0) / f(adx)) + float(f(y0 + dy) + f(2097153.0))).astype(np.float32)
    got = ((tt.view(np.uint32) & 0x7FC) >> 2).astype(np.int64) - 1
    assert list(np.flatnonzero(got != want)) == [0] and got[0] == want[0] + 1


def test_packed_op_model_bit_exact():
    """The v_pk_mul/add formulation (op_sel / neg modifiers emulated in numpy) == oracle, incl. window/overlap-add."""
    blob, offs, _, _ = _image(SETUPS["stereo"]())
    img = fm.Image(blob, offs)
    rng = np.random.default_rng(1)
    W = po.tables(11)[3]
    for trial in range(3):
        x = (rng.standard_normal(1024) * 0.3).astype(np.float32)
        xprev = (rng.standard_normal(1024) * 0.3).astype(np.float32)
        prev = po.inverse_mdct(xprev, 11)[1024:]
        got, ola = fm.imdct_wave_pk(x, img, prev_pb=prev[:512], window=W)
        want = po.inverse_mdct(x, 11)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        i = np.arange(1024)
        ref = (want[:1024] * W[i]).astype(np.float32) + (prev * W[1023 - i]).astype(np.float32)
        assert np.array_equal(ola.view(np.uint32), ref.astype(np.float32).view(np.uint32))
