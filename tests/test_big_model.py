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
def test_floor_group_equals_render_line(n2):
    rng = np.random.default_rng(n2)
    for trial in range(40):
        K = int(rng.integers(2, 66))
        inner = np.sort(rng.choice(np.arange(1, n2), K - 2, replace=False)) if K > 2 else np.zeros(0, np.int64)
        if trial % 4 == 0 and K > 10:   # runs of adjacent posts
            inner = np.sort(np.unique(np.concatenate([inner[: K // 2], np.arange(100, 100 + K)])))[: K - 2]
        xs = np.concatenate([[0], inner, [n2]]).astype(np.int64)
        K = len(xs)
        ys = rng.integers(0, 256, K)
        want = _render(xs, ys, n2)
        got = np.concatenate([bm.floor_group(xs, ys, K, k0) for k0 in range(0, n2, 4)])
        assert np.array_equal(got, want), trial
