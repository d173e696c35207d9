"""CPU tests of the short-block kernel's design (k_short): the numpy lane model (tests/short_model.py) fed with the product's
LDS table image must reproduce the oracle bit for bit -- transform, overlap-add and floor curve."""
import ctypes as C

import numpy as np
import pytest

import short_model as sm
from common import SETUPS, floor_from_record, floor_x_sorted, oracle_headers, po, sg
from lewton_amd import _native as N
from lewton_amd import audio, header


def _image(setup):
    idp, _, stp = setup.headers()
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    n_units = C.c_size_t(16)
    units = (C.c_uint8 * 128)()
    size = N.lw_debug_short_image(ident._h, st._h, None, 0, units, C.byref(n_units))
    if size == 0:
        return None, None, ident, st
    buf = (C.c_uint8 * size)()
    N.lw_debug_short_image(ident._h, st._h, buf, size, None, None)
    return bytes(buf), np.frombuffer(bytes(units), np.int8)[: 8 * n_units.value].reshape(-1, 8), ident, st


def test_short_image_eligibility_and_units():
    blob, units, _, _ = _image(SETUPS["stereo"]())
    assert blob is not None and len(blob) == sm.LWS["total"]
    assert units.tolist() == [[0, 1, 1, 0, 0, units[0][5], units[0][5], 0]]       # one coupled pair, one floor
    blob51, units51, _, _ = _image(SETUPS["surround51"]())
    assert blob51 is not None and len(units51) == 3                                # two coupled pairs + one uncoupled pair
    assert _image(SETUPS["stereo_9_12"]())[0] is None                             # other block sizes -> generic kernels
    assert _image(SETUPS["mono_small"]())[0] is None


def test_lane_model_short_imdct_bit_exact():
    blob, _, _, _ = _image(SETUPS["stereo"]())
    img = sm.Image(blob)
    rng = np.random.default_rng(0)
    for trial in range(4):
        x = (rng.standard_normal((8, 128)) * (0.3 if trial else 1e-20)).astype(np.float32)
        if trial == 3:
            x[rng.integers(0, 8, 700), rng.integers(0, 128, 700)] = 0.0
        got = sm.imdct_wave(x, img)
        for g in range(8):
            want = po.inverse_mdct(x[g], 8)
            assert np.array_equal(got[g].view(np.uint32), want.view(np.uint32)), (trial, g)


def test_lane_model_short_overlap_add_bit_exact():
    """audio.rs:1116-1118 with the short window: out[i] = cur[i] * w[i] + prev_right[i] * w[127 - i]"""
    setup = SETUPS["stereo"]()
    blob, _, ident, _ = _image(setup)
    img = sm.Image(blob)
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((8, 128)) * 0.3).astype(np.float32)
    prev = (rng.standard_normal((8, 128)) * 0.3).astype(np.float32)
    prev_pb = np.stack([po.inverse_mdct(prev[g], 8)[128:192] for g in range(8)])     # pb(0..63) of each predecessor
    blocks, ola, pb = sm.imdct_wave(x, img, prev_pb)
    w = po.window(8) if hasattr(po, "window") else None
    for g in range(8):
        cur = po.inverse_mdct(x[g], 8)
        pr = po.inverse_mdct(prev[g], 8)[128:]
        assert np.array_equal(pb[g].view(np.uint32), cur[128:192].view(np.uint32))
        if w is None:   # window from the image: (s[q], s[127 - q]) pairs
            w = np.zeros(128, np.float32)
            for c in range(2):
                for l in range(8):
                    mp = 2 * l + c
                    for k, q in enumerate((63 - 2 * mp, 62 - 2 * mp, 1 + 2 * mp, 2 * mp)):
                        w[q] = img.win[c][l][2 * k]
                        w[127 - q] = img.win[c][l][2 * k + 1]
        want = (cur[:128] * w).astype(np.float32) + (pr * w[::-1]).astype(np.float32)
        assert np.array_equal(ola[g].view(np.uint32), want.astype(np.float32).view(np.uint32)), g


@pytest.mark.parametrize("name", ["stereo", "surround51"])
def test_short_floor_curve_model_equals_render_line(name):
    setup = SETUPS[name]()
    blob, units, hid, hst = _image(setup)
    img = sm.Image(blob)
    short_mode = next(i for i, m in enumerate(setup.modes) if not m.blockflag)
    pk = sg.make_stream(setup, "S", 12, seed=5, p_floor_unused=0.1)
    n = 0
    for p in pk:
        d = audio.entropy_decode_host(hid, hst, p)
        for u in units:
            for c, slot in ((int(u[0]), int(u[3])), (int(u[1]), int(u[4]))):
                if c < 0:
                    continue
                xs = floor_x_sorted(setup, short_mode, c)
                assert [float(v) for v in xs] == img.xsf[slot][: len(xs)].tolist()
                rec = d["floor"][c]
                got = sm.floor_group_model(rec, xs, img.inv_db, img.sid16[slot])
                want = floor_from_record(rec, xs, 128, img.inv_db)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (name, c)
                n += 1
    assert n >= 24
