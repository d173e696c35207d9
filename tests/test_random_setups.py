"""CPU side of the random-setup campaign (round 6): setup headers drawn from the whole space /root/reference/src/header.rs
accepts (streamgen.random_setup) through
  * both header parsers (product, oracle) -- a generated setup is legal, neither may reject it;
  * the product's host entropy stage against the oracle's taps, packet by packet, incl. damaged packets (bit cursor, residue
    vectors before the inverse coupling, floor records x residue = the pre-IMDCT spectrum, floor-0 curves);
  * the independent second decoder (tests/independent_decoder.py) against the oracle at all four taps, the samples and the window
    state -- the two restatements share no code;
  * the planner's census line (lw_debug_plan_census), which names a kernel for every block class.
The GPU half is tests/test_gpu_random_setups.py / tools/fuzz_gpu_setups.py."""
import ctypes as C

import numpy as np
import pytest

import independent_decoder as ind
from common import floor_from_record, po, sg
from lewton_amd import _native as N
from lewton_amd import audio, header
from test_oracle_independent import _compare


def _headers(setup):
    idp, _cmt, stp = setup.headers()
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    o_id = po.Ident(idp)
    return ident, st, o_id, po.Setup(stp, o_id)


def check_host_stage(setup, packets):
    """product host entropy stage == oracle on every packet (each against a fresh window state: the taps do not depend on it)"""
    ident, st, o_id, o_st = _headers(setup)
    inv_db = po.inverse_db_table()
    L = po.lib()
    L.lwo_debug_bits_consumed.restype = C.c_size_t
    n_ok = 0
    for i, p in enumerate(packets):
        c1 = c2 = None
        try:
            c1 = audio.get_decoded_sample_count(ident, st, p)
        except audio.AudioReadError as e:
            c1 = -e.code
        try:
            c2 = po.get_decoded_sample_count(o_id, o_st, p)
        except po.OracleError as e:
            c2 = -e.code
        assert c1 == c2, (i, c1, c2)
        try:
            _out, taps = po.read_audio_packet(o_id, o_st, p, po.Pwr(), "f32", taps=True)
            o_rc = 0
        except po.OracleError as e:
            o_rc = e.code
        try:
            rec = audio.entropy_decode_host(ident, st, p)
            rc = 0
        except audio.AudioReadError as e:
            rc = e.code
        assert rc == o_rc, (i, rc, o_rc)
        if rc:
            continue
        assert rec["bits"] == L.lwo_debug_bits_consumed(), i
        n = taps["n"]
        assert np.array_equal(rec["residue"].view(np.uint32), taps["residue_pre_inverse"].view(np.uint32)), i
        mp = setup.mappings[setup.modes[rec["mode"]].mapping]
        for c in range(setup.channels):
            fl = setup.floors[mp.submap_floor[mp.mux[c]]]
            kind = int(rec["floor"][c, 0])
            if kind == 0xFFFF:
                assert not taps["pre_mdct"][c].any(), (i, c)
            elif isinstance(fl, sg.Floor0):
                assert kind == 0xFFFE
                with np.errstate(all="ignore"):
                    got = (rec["floor_curve"][c] * taps["residue_post_inverse"][c][: n // 2]).astype(np.float32)
                assert np.array_equal(got.view(np.uint32), taps["pre_mdct"][c][: n // 2].view(np.uint32)), (i, c)
            else:
                spec = floor_from_record(rec["floor"][c], sorted(fl.x_list), n // 2, inv_db) * taps["residue_post_inverse"][c]
                assert np.array_equal(spec.view(np.uint32), taps["pre_mdct"][c].view(np.uint32)), (i, c)
        n_ok += 1
    return n_ok


@pytest.mark.parametrize("first", [0, 25, 50])
def test_random_setups_host_stage_equals_oracle(first):
    n_ok = 0
    for seed in range(first, first + 25):
        rng = np.random.default_rng(seed)
        setup = sg.random_setup(rng)
        n_ok += check_host_stage(setup, sg.random_stream(setup, rng, 10, seed=seed, p_damage=0.1))
    assert n_ok > 200


@pytest.mark.parametrize("first", [1000, 1020])
def test_random_setups_independent_decoder_equals_oracle(first):
    n_ok = 0
    for seed in range(first, first + 20):
        rng = np.random.default_rng(seed)
        setup = sg.random_setup(rng, allow_floor0=False)     # (the independent decoder covers floor 1 only)
        idp, _cmt, stp = setup.headers()
        pk = sg.random_stream(setup, rng, 8, seed=seed, p_damage=0.1)
        ok, _samples = _compare(idp, stp, pk, expect_errors=True)
        n_ok += ok
    assert n_ok > 120


def test_small_block_transform_is_lewtons_not_the_definitions():
    """64- / 128-point blocks: stages l = 0 (and 1 for 128 points) run although the definition has ld - 6 = 0 / 1 of them
    (imdct.rs:445-452, :95).  Oracle (literal loops) and independent decoder (data-parallel form) must agree."""
    rng = np.random.default_rng(5)
    for bs in (6, 7, 8):
        x = rng.standard_normal((1 << bs) // 2).astype(np.float32)
        got = ind.imdct(x, ind.Tables(bs))
        want = po.inverse_mdct(x, bs)
        assert np.array_equal(got.view(np.uint32), np.asarray(want, np.float32).view(np.uint32)), bs


def test_plan_census_names_a_kernel_for_every_block_class():
    seen = set()
    for seed in range(60):
        setup = sg.random_setup(np.random.default_rng(seed))
        ident, st, _o_id, _o_st = _headers(setup)
        buf = C.create_string_buffer(2048)
        assert N.lib.lw_debug_plan_census(ident._h, st._h, buf, 2048) < 2048
        parts = dict(x.split("=", 1) for x in buf.value.decode().split(" | "))
        assert set(parts) == {"long", "short", "transitions", "entropy"}
        any_long = any(m.blockflag for m in setup.modes)
        any_short = any(not m.blockflag for m in setup.modes)
        if setup.bs0 == setup.bs1 and any_long:
            any_short = False          # (equal block sizes: every mode is planned with the long blocks, lw_unified_classes)
        assert (parts["long"] == "none") == (not any_long) and (parts["short"] == "none") == (not any_short)
        assert parts["entropy"] == "device" or parts["entropy"].startswith("host (")
        seen.add(parts["long"].split(" ")[0])
    assert {"k_long", "k_long10", "k_long12"} <= seen
