"""lewton's C API (include/lewton.h = src/capi.rs:13-147) served by liblewton_amd.so."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from common import ROOT, SETUPS, oracle_headers, sg
from lewton_amd import capi
from oracle import pyoracle as po


def test_header_and_exports_match_the_reference_surface():
    hdr = open(os.path.join(ROOT, "include", "lewton.h")).read()
    declared = set(re.findall(r"\b(lewton_[a-z0-9_]+)\s*\(", hdr))
    # the #[no_mangle] functions of capi.rs:78,95,103,125,141,146 plus lewton_samples_f32 (:132, see lewton.h)
    assert declared == {"lewton_context_from_extradata", "lewton_context_reset", "lewton_decode_packet",
                        "lewton_samples_count", "lewton_samples_f32", "lewton_samples_drop", "lewton_context_drop"}
    assert declared == set(capi.CAPI_SYMBOLS)
    assert "LEWTON_LEWTON_H" in hdr                      # cbindgen.toml include_guard


def test_context_from_extradata_parsing():
    setup = SETUPS["stereo"]()
    idp, cmt, stp = setup.headers()
    long_comment = setup.comment_packet(comments=(b"A=" + b"x" * 700,))   # two lacing bytes for the comment length
    for c in (cmt, long_comment):
        ctx = capi.lewton_context_from_extradata(capi.make_extradata(idp, c, stp), len(capi.make_extradata(idp, c, stp)))
        assert ctx
        capi.lewton_context_drop(ctx)
    good = capi.make_extradata(idp, cmt, stp)
    assert not capi.lewton_context_from_extradata(None, 0)             # NULL data
    assert not capi.lewton_context_from_extradata(b"", 0)              # empty
    assert not capi.lewton_context_from_extradata(b"\x01" + good[1:], len(good))   # must start with 2
    assert not capi.lewton_context_from_extradata(b"\x02\xff", 2)      # lacing runs off the end
    assert not capi.lewton_context_from_extradata(good[:40], 40)       # headers shorter than announced (reference: panic)
    assert not capi.lewton_context_from_extradata(good[:-20], len(good) - 20)      # truncated setup header
    bad = bytearray(good)
    bad[3 + 1] ^= 0xFF                                                  # 'v' of the ident header's "vorbis"
    assert not capi.lewton_context_from_extradata(bytes(bad), len(bad))
    # NULL arguments of lewton_decode_packet return 1 (capi.rs:106-108)
    ctx = capi.lewton_context_from_extradata(good, len(good))
    out = C.c_void_p()
    assert capi.lewton_decode_packet(None, b"x", 1, C.byref(out)) == 1
    assert capi.lewton_decode_packet(ctx, None, 0, C.byref(out)) == 1
    assert capi.lewton_decode_packet(ctx, b"x", 1, None) == 1
    capi.lewton_context_drop(ctx)


@pytest.mark.gpu
@pytest.mark.parametrize("name,pattern", [("stereo", "LSSL"), ("surround51", "LLS"), ("mono_small", "SL")])
def test_decode_matches_oracle(name, pattern):
    setup = SETUPS[name]()
    idp, cmt, stp = setup.headers()
    ident, st = oracle_headers(setup)
    pk = sg.make_stream(setup, pattern, 14, seed=4)
    ed = capi.make_extradata(idp, cmt, stp)
    ctx = capi.lewton_context_from_extradata(ed, len(ed))
    pwr = po.Pwr()
    for i, p in enumerate(pk):
        if i == 9:                                        # seek: both sides forget the previous window
            capi.lewton_context_reset(ctx)
            pwr = po.Pwr()
        rc, ch = capi.decode_packet(ctx, p)
        want = po.read_audio_packet(ident, st, p, pwr, "f32")
        assert rc == 0 and len(ch) == ident.audio_channels
        for c in range(ident.audio_channels):
            assert ch[c].shape == want[c].shape
            assert np.max(np.abs(ch[c] - want[c]), initial=0.0) <= 1e-5
        if i in (0, 9):
            assert all(len(c) == 0 for c in ch)         # lewton_samples_count == 0 right after a reset
    # an undecodable packet -> 2 (capi.rs:113-117); a header packet is AudioIsHeader in the reference
    assert capi.decode_packet(ctx, idp) == (2, None)
    out = C.c_void_p()
    capi.lewton_decode_packet(ctx, pk[0], len(pk[0]), C.byref(out))
    assert not capi.lewton_samples_f32(out, ident.audio_channels)       # no such channel -> NULL
    capi.lewton_samples_drop(out)
    capi.lewton_context_drop(ctx)
