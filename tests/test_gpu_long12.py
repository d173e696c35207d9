"""k_long12 -- k_long's design for blocksize_1 = 12 (4096-point long blocks, one wave per channel; lw_long12.inc) -- against the
ORACLE, and against the workgroup-pipeline kernel k_big<12> on the very same batches (-m gpu).

Covered: dense launches (the bench shape: 256 streams x 16 packets = two rounds per workgroup), streams cut across rounds and
workgroups (hand-over through LDS inside a round and from round to round -- the published right half shares the wave's gather
area --, through the halo pre-pass at chunk starts, through the state pool between launches), 5.1 (two coupled pairs and two
uncoupled channels = six waves per packet), mono, stereo without coupling, residue types 1 and 2, unused floors, truncated
packets, the three sample formats, and mixed short/long streams (short blocks in k_short; long blocks next to 256- / 512-point
short ones in k_long12's EDGE form, next to 1024-point ones in the generic kernels: time-domain blocks exchanged both ways)."""
import numpy as np
import pytest

from common import sg
from test_gpu_long10 import _compare, _decode, _decoder, _oracle

pytestmark = pytest.mark.gpu


def _uncoupled_9_12():
    st = sg.stereo_setup(44100, 9, 12, residue_type=1)
    for m in st.mappings:
        m.coupling = []
    return st


L12_SETUPS = {
    "stereo_9_12": lambda: sg.stereo_setup(44100, 9, 12),
    "stereo_8_12": lambda: sg.stereo_setup(44100, 8, 12),
    "stereo_10_12_t1": lambda: sg.stereo_setup(44100, 10, 12, residue_type=1),
    "surround51_9_12": lambda: sg.surround51_setup(48000, 9, 12),
    "mono_7_12": lambda: sg.mono_setup(7, 12, 44100),
    "uncoupled_9_12": _uncoupled_9_12,
}


@pytest.mark.parametrize("name", sorted(L12_SETUPS))
@pytest.mark.parametrize("fmt", ["i16", "f32", "i16_interleaved"])
def test_long12_all_long_streams_vs_oracle_and_workgroup_kernel(name, fmt):
    setup = L12_SETUPS[name]()
    audio, dec = _decoder(setup)
    n_streams = 21
    streams = [sg.make_stream(setup, "L", 19 + (s % 5), seed=5100 + s, p_floor_unused=0.08) for s in range(n_streams)]
    streams[3][7] = streams[3][7][: len(streams[3][7]) // 2]      # a truncated packet (end of packet inside the residue)
    want, wstates = _oracle(setup, streams, fmt)
    cuts = [0, 1, 2, 9, 24]             # one packet per stream per launch twice (state pool), then runs
    got, seen, states = _decode(dec, audio, streams, cuts, fmt)
    assert "k_long12" in seen and "k_big" not in seen, seen
    _compare(got, want, fmt, name)
    for s in range(n_streams):
        assert np.array_equal(states[s].view(np.uint32), wstates[s].view(np.uint32)), s
    got2, seen2, _ = _decode(dec, audio, streams, cuts, fmt, l10=0)
    assert "k_big" in seen2 and "k_long12" not in seen2, seen2
    _compare(got2, want, fmt, name + " (k_big)")


@pytest.mark.parametrize("rounds", [1, 2, 3, 5])
def test_long12_forced_rounds_hand_over_paths(rounds):
    setup = L12_SETUPS["stereo_9_12"]()
    audio, dec = _decoder(setup)
    streams = [sg.make_stream(setup, "L", 30 + 3 * (s % 4), seed=5300 + s) for s in range(9)]
    want, wstates = _oracle(setup, streams, "i16")
    got, seen, states = _decode(dec, audio, streams, [0, 5, 42], "i16", rounds=rounds)
    assert "k_long12" in seen, seen
    if rounds < 5:
        assert "k_long12<halo>" in seen, seen
    _compare(got, want, "i16", "rounds=%d" % rounds)
    for s in range(len(streams)):
        assert np.array_equal(states[s].view(np.uint32), wstates[s].view(np.uint32)), s


@pytest.mark.parametrize("name", ["stereo_9_12", "stereo_8_12", "stereo_10_12_t1", "surround51_9_12", "uncoupled_9_12"])
@pytest.mark.parametrize("fmt", ["i16", "f32", "i16_interleaved"])
def test_long12_mixed_short_long_streams(name, fmt):
    """short blocks of 256 / 512 points: long blocks with a short slope stay in k_long12 (its EDGE form: the raw edges go to
    k_short<8 / 16> through the edge buffer) -- no packet of such a stream on the generic kernels; 1024-point short blocks
    (k_short<32> has no edge form): those long blocks keep floor, coupling and transform in k_long12, which writes their whole
    time-domain block, and take window / overlap-add / state from k_ola_generic (LW_RF_TDONLY); no k_imdct_generic either way"""
    setup = L12_SETUPS[name]()
    audio, dec = _decoder(setup)
    pats = ["LLLSSSLLLL", "LLSLLLSSLLLLL", "LSSSSSSLLL", "LLLLLLLSL", "SLSLLSSL"]
    streams = [sg.make_stream(setup, pats[s % 5], 26 + s % 3, seed=5500 + s, p_floor_unused=0.05) for s in range(9)]
    streams[2][11] = streams[2][11][: len(streams[2][11]) // 3]      # a truncated packet
    want, wstates = _oracle(setup, streams, fmt)
    got, seen, states = _decode(dec, audio, streams, [0, 1, 2, 3, 4, 5, 6, 17, 29], fmt)
    assert "k_long12" in seen and "k_short" in seen, seen
    edge = setup.bs0 in (8, 9)
    assert any("generic" in k for k in seen) == (not edge), seen
    assert "k_imdct_generic" not in seen and "k_decouple" not in seen, seen
    _compare(got, want, fmt, name)
    for s in range(len(streams)):
        assert np.array_equal(states[s].view(np.uint32), wstates[s].view(np.uint32)), s
    if edge:    # the same batches with the edge form off (lw_debug_batch_set_long10(1)): the generic kernels' route, same bytes
        got2, seen2, states2 = _decode(dec, audio, streams, [0, 1, 2, 3, 4, 5, 6, 17, 29], fmt, l10=1)
        assert any("generic" in k for k in seen2), seen2
        _compare(got2, want, fmt, name + " (edge form off)")


@pytest.mark.parametrize("rounds", [1, 2, 4])
def test_long12_edge_form_across_rounds_and_launches(rounds):
    """forced launch shapes: chunk starts right behind / in front of short runs, one packet per stream per launch (every raw edge and
    short right part crosses the state pool), long runs (through LDS)"""
    setup = L12_SETUPS["stereo_9_12"]()
    audio, dec = _decoder(setup)
    pats = ["LSLSLLSSL", "LLLLSSSSLLLLLLLL", "SSLLLLLLLS", "LLSLLLLLLLLLLLSLL"]
    streams = [sg.make_stream(setup, pats[s % 4], 31 + s % 4, seed=5700 + s, p_floor_unused=0.05) for s in range(11)]
    want, wstates = _oracle(setup, streams, "i16")
    got, seen, states = _decode(dec, audio, streams, [0, 1, 2, 3, 4, 5, 6, 7, 8, 19, 40], "i16", rounds=rounds)
    assert "k_long12" in seen and "k_short" in seen and not any("generic" in k for k in seen), seen
    _compare(got, want, "i16", "rounds=%d" % rounds)
    for s in range(len(streams)):
        assert np.array_equal(states[s].view(np.uint32), wstates[s].view(np.uint32)), s


def test_long12_dense_bench_shapes():
    from test_gpu_quoted_shapes import _run_dense
    for packets in (4096, 8192):
        bad, kernels, n = _run_dense("11", "i16", packets=packets)
        assert n == packets and bad == 0 and kernels == "k_long12", (bad, kernels)


@pytest.mark.parametrize("bs", [(7, 10), (6, 10), (10, 12), (7, 12), (11, 12)])
@pytest.mark.parametrize("fmt", ["i16", "f32", "i16_interleaved"])
def test_short_slopes_without_an_edge_form_keep_the_transform_in_the_wave_kernel(bs, fmt):
    """long blocks of k_long10 / k_long12 next to short blocks that have no edge form (64- / 128-point and 2048-point short blocks run on
    the generic kernels, 1024-point ones on k_short<32>): floor, inverse coupling and transform stay in the wave kernel, which writes
    the block's whole time-domain samples; k_ola_generic does window / overlap-add / state (LW_RF_TDONLY, as next to k_long).
    Against the oracle, and against the route with that switched off (lw_debug_batch_set_long10(1): k_decouple + k_imdct_generic)."""
    setup = sg.stereo_setup(44100, *bs)
    audio, dec = _decoder(setup)
    pats = ["LLLSSSLLLL", "LLSLLLSSLLLLL", "LSSSSSSLLL", "LLLLLLLSL", "SLSLLSSL"]
    streams = [sg.make_stream(setup, pats[s % 5], 24 + s % 3, seed=5900 + s, p_floor_unused=0.05) for s in range(9)]
    streams[4][10] = streams[4][10][: len(streams[4][10]) // 3]
    want, wstates = _oracle(setup, streams, fmt)
    cuts = [0, 1, 2, 3, 4, 5, 6, 15, 27]
    got, seen, states = _decode(dec, audio, streams, cuts, fmt)
    wave = "k_long10" if bs[1] == 10 else "k_long12"
    assert wave in seen and "k_ola_generic" in seen, seen
    _compare(got, want, fmt, "%s/%s" % bs)
    for s in range(len(streams)):
        assert np.array_equal(states[s].view(np.uint32), wstates[s].view(np.uint32)), s
    got2, seen2, _ = _decode(dec, audio, streams, cuts, fmt, l10=1)
    assert "k_imdct_generic" in seen2, seen2
    _compare(got2, want, fmt, "%s/%s (generic route)" % bs)
    if bs[0] == 10:     # k_short<32> takes the short blocks: nothing of the stream's transforms is left to the generic kernels
        assert "k_imdct_generic" not in seen and "k_decouple" not in seen, seen
