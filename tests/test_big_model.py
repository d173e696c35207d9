"""CPU tests of the big-block kernel's design (k_big<BS>, 4096 / 8192-point blocks): the numpy thread model (tests/big_model.py)
must reproduce the oracle bit for bit -- transform, overlap-add, floor curve."""
import numpy as np
import pytest

import big_model as bm
from common import po


@pytest.mark.parametrize("bs", [12, 13])
def test_thread_model_imdct_and_overlap_add_bit_exact(bs):
    n = 1 << bs
    n2, n4 = n // 2, n // 4
    tabs = po.tables(bs)
    W = tabs[3]
    rng = np.random.default_rng(bs)
    for trial in range(3):
        x = (rng.standard_normal(n2) * (0.3 if trial else 1e-20)).astype(np.float32)
        if trial == 2:
            x[rng.integers(0, n2, n2 // 2)] = 0.0
        prev = (rng.standard_normal(n2) * 0.3).astype(np.float32)
        prev_td = po.inverse_mdct(prev, bs)
        td, ola, pb = bm.block(x, bs, tabs, prev_td[n2:n2 + n4][::-1].copy())   # pb(p) = right half at q = n/4 - 1 - p
        want = po.inverse_mdct(x, bs)
        assert np.array_equal(td.view(np.uint32), want.view(np.uint32)), trial
        i = np.arange(n2)
        want_ola = (want[:n2] * W[i]) + (prev_td[n2:] * W[n2 - 1 - i])          # audio.rs:1116-1118
        assert np.array_equal(ola.view(np.uint32), want_ola.astype(np.float32).view(np.uint32)), trial
        assert np.array_equal(pb, want[n2:n2 + n4][::-1])


def _render(xs, ys, n2):
    """audio.rs:503-548: render_line between consecutive active posts, flat to n/2 behind the last"""
    out = np.zeros(n2, np.int64)
    lx, ly = int(xs[0]), int(ys[0])
    for x, y in zip(xs[1:], ys[1:]):
        x, y = int(x), int(y)
        dy, adx = y - ly, x - lx
        ady, base = abs(dy), dy // adx if dy >= 0 else -((-dy) // adx)
        sy = base - 1 if dy < 0 else base + 1
        ady -= abs(base) * adx
        yy, err = ly, 0
        if lx < n2:
            out[lx] = yy
        for xx in range(lx + 1, min(x, n2)):
            err += ady
            if err >= adx:
                err -= adx
                yy += sy
            else:
                yy += base
            out[xx] = yy
        lx, ly = x, y
    out[lx:] = ly
    return out


@pytest.mark.parametrize("n2", [2048, 4096])
def test_floor_segment_table_equals_render_line(n2):
    rng = np.random.default_rng(n2)
    for trial in range(60):
        Fp = int(rng.integers(2, 66))
        inner = np.sort(rng.choice(np.arange(1, n2), Fp - 2, replace=False)) if Fp > 2 else np.zeros(0, np.int64)
        if trial % 4 == 0 and Fp > 10:   # runs of adjacent posts
            inner = np.sort(np.unique(np.concatenate([inner[: Fp // 2], np.arange(100, 100 + Fp)])))[: Fp - 2]
        xs = np.concatenate([[0], inner, [n2]]).astype(np.int64)
        Fp = len(xs)
        ys = rng.integers(0, 256, Fp)
        active = rng.random(Fp) < (0.2, 0.6, 1.0)[trial % 3]
        active[0] = active[1 if Fp == 2 else Fp - 1] = True     # posts 0 and 1 (x = 0 and x = n/2 here) are always used (audio.rs:374-386)
        ax, ay = xs[active], ys[active]
        want = _render(ax, ay, n2)
        got = bm.floor_bins(xs, bm.floor_entries(xs, ys, active), n2)
        assert np.array_equal(got, want), trial


def test_floor_by_two_fmas_and_a_mask_is_exact_up_to_4096_bins():
    """k_big's floor_bin (k_long's, tests/test_fast_model.py, for segments up to 4096 bins long): y = ((bits(fma(fma(k, dy, c0),
    fl(1/adx), 2^21 + 1 + y_base)) & 0x7fc) >> 2) - 1 equals render_line's y0 + trunc((k - x0) dy / adx) for every adx in
    1025..4096 (1..1024: the other test) with EVERY offset inside it, at both ends of the y range, with the reciprocal one ulp
    off either way (v_rcp_f32) -- for |dy| in a set that holds the extremes, the smallest values and a different random dozen
    per adx (every dy for every 64th adx).  The f64 product z * rinv is exact (<= 24 + 24 bits) and the f64 sum is rounded far
    below the quarter the f32 result is rounded to, so f32(f64 expression) is the fused multiply-add's result except on exact
    quarter ties, which do not touch the integer part (see the kernel's comment)."""
    rng = np.random.default_rng(4096)
    fixed = np.array([1, 2, 3, 5, 127, 128, 129, 253, 254, 255])
    for adx in range(1025, 4097):
        mags = np.arange(1, 256) if adx % 64 == 0 else np.unique(np.concatenate([fixed, rng.integers(1, 256, 12)]))
        dy = np.concatenate([-mags[::-1], [0], mags]).astype(np.int64)[:, None]
        y0s = (np.where(dy >= 0, 0, -dy), np.where(dy >= 0, 255 - dy, 255))   # the lowest / highest y0 with y0, y0 + dy in 0..255
        t = np.arange(0, adx, dtype=np.int64)[None, :]
        off = np.sign(dy) * ((t * np.abs(dy)) // adx)
        z = (np.where(dy >= 0, t * dy + 0.5, (adx - t) * np.abs(dy) + adx - 0.5) - adx / 8.0).astype(np.float64)
        assert np.array_equal(z.astype(np.float32).astype(np.float64), z)
        r0 = np.float32(1.0) / np.float32(adx)
        for y0 in y0s:
            w = (np.where(dy >= 0, y0, y0 + dy) + 2097153.0).astype(np.float64)
            for rinv in (r0, np.nextafter(r0, np.float32(2)), np.nextafter(r0, np.float32(0))):
                tt = (z * np.float64(rinv) + w).astype(np.float32)
                got = ((tt.view(np.uint32) & 0x7FC) >> 2).astype(np.int64) - 1
                assert np.array_equal(got, y0 + off), (adx, rinv)


def test_wave_level_index_maps_of_the_next_design_bit_exact():
    """one wave per 4096-point channel, 16 pairs per lane, four stages before the first LDS round trip (not built
    yet; the maps are pinned here so that the kernel can be written against them)"""
    rng = np.random.default_rng(5)
    for trial in range(2):
        x = (rng.standard_normal(2048) * 0.3).astype(np.float32)
        got = bm.block_wave(x, 12)
        want = po.inverse_mdct(x, 12)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), trial
