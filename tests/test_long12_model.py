"""CPU tests of k_long12's design (one wave per 4096-point channel, 16 pairs per lane; lw_long12.inc): the numpy lane model
(tests/long12_model.py) fed with the product's LDS image (LwL12Layout, built by lw_fast.cpp) must reproduce the oracle bit for
bit -- transform and overlap-add -- and its gather must be free of bank conflicts."""
import ctypes as C

import numpy as np
import pytest

import long12_model as lm
from common import po, sg
from lewton_amd import _native as N
from lewton_amd import header


def _image(setup):
    idp, _, stp = setup.headers()
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    n_units = C.c_size_t(16)
    units = (C.c_uint8 * 128)()
    lanes = C.c_uint32(0)
    size = N.lw_debug_short_image(ident._h, st._h, 1, None, 0, units, C.byref(n_units), C.byref(lanes))
    buf = (C.c_uint8 * max(1, size))()
    N.lw_debug_short_image(ident._h, st._h, 1, buf, size, None, None, None)
    return bytes(buf)[:size], lanes.value


def test_image_is_built_for_blocksize_12_only():
    blob, lanes = _image(sg.stereo_setup(44100, 9, 12))
    assert lanes == 128 and len(blob) == lm.LAYOUT["total"]
    blob13, lanes13 = _image(sg.stereo_setup(44100, 6, 13))
    assert lanes13 == 256 and len(blob13) == 0           # 8192 points: k_big (no LDS image)


def test_lane_model_imdct_and_overlap_add_bit_exact():
    blob, _ = _image(sg.stereo_setup(44100, 9, 12))
    img = lm.Image(blob)
    A, B, Ct, W, _ = po.tables(12)
    assert np.array_equal(img.inv_db, po.inverse_db_table())
    rng = np.random.default_rng(0)
    for trial in range(4):
        x = (rng.standard_normal(lm.N2) * (0.3 if trial else 1e-20)).astype(np.float32)
        if trial == 3:
            x[rng.integers(0, lm.N2, 1500)] = 0.0
        prev = (rng.standard_normal(lm.N2) * 0.3).astype(np.float32)
        prev_td = po.inverse_mdct(prev, 12)
        td, ola, pb = lm.imdct_wave(x, img, A, prev_td[lm.N2:lm.N2 + lm.N4].copy())
        want = po.inverse_mdct(x, 12)
        assert np.array_equal(td.view(np.uint32), want.view(np.uint32)), trial
        i = np.arange(lm.N2)
        want_ola = (want[:lm.N2] * W[i]).astype(np.float32) + (prev_td[lm.N2:] * W[lm.N2 - 1 - i]).astype(np.float32)   # audio.rs:1116-1118
        assert np.array_equal(ola.view(np.uint32), want_ola.astype(np.float32).view(np.uint32)), trial
        assert np.array_equal(pb, want[lm.N2:lm.N2 + lm.N4])


def test_gather_is_free_of_bank_conflicts():
    assert sorted(lm.gather_slot(p) for p in range(lm.P)) == list(range(lm.P))
    assert lm.gather_bank_cycles() == (64, 32)


@pytest.mark.parametrize("E", [64, 128])
@pytest.mark.parametrize("edge_l,edge_r", [(True, False), (False, True), (True, True)])
def test_edge_form_geometry(E, edge_l, edge_r):
    """k_long12<EDGE>: what a wave stores for a long block with short slopes (E = blocksize_0 / 4) against the block's time-domain
    samples -- the un-windowed samples between the slopes (audio.rs:1119), the raw edges k_short overlaps, the short right part
    that becomes the stream's state (audio.rs:1056-1073 for the bounds)."""
    rng = np.random.default_rng(E + 2 * edge_l + edge_r)
    x = (rng.standard_normal(lm.N2) * 0.3).astype(np.float32)
    td = po.inverse_mdct(x, 12)
    pa, pb = td[:lm.N4].copy(), td[lm.N2:lm.N2 + lm.N4].copy()
    out, left, right, state = lm.edge_form(pa, pb, E, edge_l, edge_r)
    ls = 1024 - E if edge_l else 0                  # window bounds of the block
    rs = 3072 - E if edge_r else 2048
    want = {}
    if edge_l:                                      # past the short left slope: positions 1024 + E .. 2047
        want.update({p - ls: td[p] for p in range(1024 + E, 2048)})
        assert np.array_equal(left, td[1024 - E:1024])
    if edge_r:                                      # up to the short right slope: positions 2048 .. 3071 - E
        want.update({p - ls: td[p] for p in range(2048, rs)})
        assert np.array_equal(right, td[3072 - E:3072])
        assert np.array_equal(state, td[3072 - E:3072 + E])
    assert set(out) == set(want)
    assert all(np.float32(out[k]).view(np.uint32) == np.float32(want[k]).view(np.uint32) for k in want)
    assert max(want) < rs - ls and min(want) >= (2 * E if edge_l else 2048)
