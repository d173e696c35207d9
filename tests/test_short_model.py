"""CPU tests of the block kernel's design (k_short<L>, L = 8 / 16 / 32 lanes per block = 256 / 512 / 1024 points): the numpy lane
model (tests/short_model.py) fed with the product's LDS table image must reproduce the oracle bit for bit -- transform,
overlap-add and floor curve -- for both block classes (short blocks; long blocks of streams k_long does not cover)."""
import ctypes as C

import numpy as np
import pytest

import short_model as sm
from common import SETUPS, floor_from_record, floor_x_sorted, oracle_headers, po, sg
from lewton_amd import _native as N
from lewton_amd import audio, header


def _image(setup, blockflag=0):
    idp, _, stp = setup.headers()
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    n_units = C.c_size_t(16)
    units = (C.c_uint8 * 128)()
    lanes = C.c_uint32(0)
    size = N.lw_debug_short_image(ident._h, st._h, blockflag, None, 0, units, C.byref(n_units), C.byref(lanes))
    if size == 0:
        return None, None, 0, ident, st
    buf = (C.c_uint8 * size)()
    N.lw_debug_short_image(ident._h, st._h, blockflag, buf, size, None, None, None)
    return bytes(buf), np.frombuffer(bytes(units), np.int8)[: 8 * n_units.value].reshape(-1, 8), lanes.value, ident, st


# (setup, blockflag) -> lanes per block
CASES = {"short256": (lambda: sg.stereo_setup(44100, 8, 11), 0, 8), "short512": (lambda: sg.stereo_setup(44100, 9, 12), 0, 16),
         "long512": (lambda: sg.stereo_setup(22050, 8, 9), 1, 16), "long1024": (lambda: sg.stereo_setup(22050, 9, 10), 1, 32),
         "short1024": (lambda: sg.stereo_setup(44100, 10, 12), 0, 32)}


def test_image_eligibility_and_units():
    blob, units, lanes, _, _ = _image(SETUPS["stereo"]())
    assert blob is not None and lanes == 8 and len(blob) == sm.layout(8)["total"]
    assert units.tolist() == [[0, 1, 1, 0, 0, units[0][5], units[0][5], 0]]       # one coupled pair, one floor
    blob51, units51, _, _, _ = _image(SETUPS["surround51"]())
    assert blob51 is not None and len(units51) == 3                                # two coupled pairs + one uncoupled pair
    assert _image(SETUPS["stereo"](), 1)[0] is None                                # its long blocks are k_long's
    assert _image(SETUPS["mono_small"]())[0] is None                               # 64-point blocks -> generic kernels
    assert _image(SETUPS["stereo_9_12"](), 0)[2] == 16
    b12, _, l12, _, _ = _image(SETUPS["stereo_9_12"](), 1)
    assert l12 == 128 and len(b12) == 30208            # 4096: k_long12's image (LwL12Layout; tests/test_long12_model.py)
    assert _image(SETUPS["stereo_6_13"](), 1)[0] is None or len(_image(SETUPS["stereo_6_13"](), 1)[0]) == 0   # 8192: k_big (no LDS image)
    assert _image(SETUPS["stereo_7_7"](), 1)[0] is None
    for name, (mk, flag, L) in CASES.items():
        blob, _, lanes, _, _ = _image(mk(), flag)
        assert blob is not None and lanes == L and len(blob) == sm.layout(L)["total"], name


@pytest.mark.parametrize("case", sorted(CASES))
def test_lane_model_imdct_bit_exact(case):
    mk, flag, L = CASES[case]
    blob, _, _, _, _ = _image(mk(), flag)
    img = sm.Image(blob, L)
    S, n2, bs = 64 // L, 16 * L, {8: 8, 16: 9, 32: 10}[L]
    rng = np.random.default_rng(0)
    for trial in range(4):
        x = (rng.standard_normal((S, n2)) * (0.3 if trial else 1e-20)).astype(np.float32)
        if trial == 3:
            x[rng.integers(0, S, 700), rng.integers(0, n2, 700)] = 0.0
        got = sm.imdct_wave(x, img)
        for g in range(S):
            want = po.inverse_mdct(x[g], bs)
            assert np.array_equal(got[g].view(np.uint32), want.view(np.uint32)), (trial, g)


@pytest.mark.parametrize("case", sorted(CASES))
def test_lane_model_overlap_add_bit_exact(case):
    """audio.rs:1116-1118 with the block's own window: out[i] = cur[i] * w[i] + prev_right[i] * w[n/2 - 1 - i]"""
    mk, flag, L = CASES[case]
    blob, _, _, _, _ = _image(mk(), flag)
    img = sm.Image(blob, L)
    S, n2, n4, bs = 64 // L, 16 * L, 8 * L, {8: 8, 16: 9, 32: 10}[L]
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((S, n2)) * 0.3).astype(np.float32)
    prev = (rng.standard_normal((S, n2)) * 0.3).astype(np.float32)
    prev_pb = np.stack([po.inverse_mdct(prev[g], bs)[n2:n2 + n4] for g in range(S)])     # pb(0 .. n/4) of each predecessor
    blocks, ola, pb = sm.imdct_wave(x, img, prev_pb)
    w = np.zeros(n2, np.float32)      # window from the image: (s[q], s[n/2 - 1 - q]) pairs
    for c in range(2):
        for l in range(L):
            mp = 2 * l + c
            for k, q in enumerate((n4 - 1 - 2 * mp, n4 - 2 - 2 * mp, 1 + 2 * mp, 2 * mp)):
                w[q] = img.win[c][l][2 * k]
                w[n2 - 1 - q] = img.win[c][l][2 * k + 1]
    for g in range(S):
        cur = po.inverse_mdct(x[g], bs)
        pr = po.inverse_mdct(prev[g], bs)[n2:]
        assert np.array_equal(pb[g].view(np.uint32), cur[n2:n2 + n4].view(np.uint32))
        want = (cur[:n2] * w).astype(np.float32) + (pr * w[::-1]).astype(np.float32)
        assert np.array_equal(ola[g].view(np.uint32), want.astype(np.float32).view(np.uint32)), g


@pytest.mark.parametrize("case", ["short256", "short512", "long1024", "surround51"])
def test_floor_curve_model_equals_render_line(case):
    mk, flag, L = CASES.get(case, (lambda: sg.surround51_setup(48000, 8, 11), 0, 8))
    setup = mk()
    blob, units, _, hid, hst = _image(setup, flag)
    img = sm.Image(blob, L)
    mode = next(i for i, m in enumerate(setup.modes) if bool(m.blockflag) == bool(flag))
    pk = sg.make_stream(setup, "L" if flag else "S", 12, seed=5, p_floor_unused=0.1)
    n = 0
    for p in pk:
        d = audio.entropy_decode_host(hid, hst, p)
        for u in units:
            for c, slot in ((int(u[0]), int(u[3])), (int(u[1]), int(u[4]))):
                if c < 0:
                    continue
                xs = floor_x_sorted(setup, mode, c)
                assert [float(v) for v in xs] == img.xsf[slot][: len(xs)].tolist()
                rec = d["floor"][c]
                got = sm.floor_group_model(rec, xs, img.inv_db, img.sid16[slot], L)
                want = floor_from_record(rec, xs, 16 * L, img.inv_db)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (case, c)
                n += 1
    assert n >= 24


@pytest.mark.parametrize("L", [8, 16, 32])
def test_gather_slots_are_free_of_bank_conflicts(L):
    """the bit-reverse gather of the block kernels (k_short<L>, and k_long10 with L = 32 per half-wave): in the plain order a wave's
    writes are 8-way and its reads 2- to 4-way bank-conflicted; blk_slot's order costs the conflict-free minimum on both sides"""
    plain = sm.gather_bank_cycles(L, lambda g, p: 8 * L * g + p)
    assert plain[0] == 256 and plain[1] >= 32, plain
    assert sm.gather_bank_cycles(L) == (32, 16)
    # linear over GF(2) (the kernel forms every address as a lane term xor a compile-time constant)
    rng = np.random.default_rng(3)
    for _ in range(200):
        g, p, q = int(rng.integers(0, 64 // L)), int(rng.integers(0, 8 * L)), int(rng.integers(0, 8 * L))
        assert sm.blk_slot(L, g, p ^ q) == sm.blk_slot(L, g, p) ^ sm.blk_slot(L, 0, q)
