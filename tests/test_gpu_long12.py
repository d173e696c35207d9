"""k_long12 -- k_long's design for blocksize_1 = 12 (4096-point long blocks, one wave per channel; lw_long12.inc) -- against the
ORACLE, and against the workgroup-pipeline kernel k_big<12> on the very same batches (-m gpu).

Covered: dense launches (the bench shape: 256 streams x 16 packets = two rounds per workgroup), streams cut across rounds and
workgroups (hand-over through LDS inside a round and from round to round -- the published right half shares the wave's gather
area --, through the halo pre-pass at chunk starts, through the state pool between launches), 5.1 (two coupled pairs and two
uncoupled channels = six waves per packet), mono, stereo without coupling, residue types 1 and 2, unused floors, truncated
packets, the three sample formats, and mixed short/long streams (1024- / 512-point short blocks in k_short, long blocks next to
short ones in the generic kernels: time-domain blocks exchanged both ways)."""
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


@pytest.mark.parametrize("name", ["stereo_9_12", "stereo_10_12_t1", "surround51_9_12"])
@pytest.mark.parametrize("fmt", ["i16", "f32"])
def test_long12_mixed_short_long_streams(name, fmt):
    setup = L12_SETUPS[name]()
    audio, dec = _decoder(setup)
    pats = ["LLLSSSLLLL", "LLSLLLSSLLLLL", "LSSSSSSLLL", "LLLLLLLSL", "SLSLLSSL"]
    streams = [sg.make_stream(setup, pats[s % 5], 26 + s % 3, seed=5500 + s, p_floor_unused=0.05) for s in range(9)]
    want, wstates = _oracle(setup, streams, fmt)
    got, seen, states = _decode(dec, audio, streams, [0, 1, 2, 3, 4, 5, 6, 17, 29], fmt)
    assert "k_long12" in seen and "k_short" in seen and any("generic" in k for k in seen), seen
    _compare(got, want, fmt, name)
    for s in range(len(streams)):
        assert np.array_equal(states[s].view(np.uint32), wstates[s].view(np.uint32)), s


def test_long12_dense_bench_shapes():
    from test_gpu_quoted_shapes import _run_dense
    for packets in (4096, 8192):
        bad, kernels, n = _run_dense("11", "i16", packets=packets)
        assert n == packets and bad == 0 and kernels == "k_long12", (bad, kernels)
