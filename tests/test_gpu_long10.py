"""k_long10 -- k_long's design for blocksize_1 = 10 (1024-point long blocks, lw_long10.inc) -- against the ORACLE, and against the
block kernel k_short<32> on the very same batches (-m gpu).

Covered: dense launches (256 streams x 16 packets, the bench shape; 48 packets per stream = three rounds per workgroup), streams
cut across rounds and workgroups (hand-over through LDS inside a round, from round to round, and through the halo pre-pass at
chunk starts), one packet per stream per launch (every right part through the state pool), 5.1 (two coupled pairs + one
uncoupled pair = three units per packet), mono (single-channel units: half a wave idle), stereo without coupling, residue
types 1 and 2, unused floors, truncated packets, the three sample formats, and mixed short/long streams whose transition blocks
go through the generic kernels (time-domain blocks exchanged both ways)."""
import numpy as np
import pytest

from common import oracle_headers, po, sg

pytestmark = pytest.mark.gpu


def _uncoupled_8_10():
    st = sg.stereo_setup(22050, 8, 10, residue_type=1)
    for m in st.mappings:
        m.coupling = []
    return st


def _surround51_8_10():
    st = sg.surround51_setup(48000, 8, 10)
    st.floors[3].x_rest = [64, 16, 256, 128, 32, 384]   # (the generator's LFE floor has a post at x = 512 = the implied end post for bs 10)
    return st


L10_SETUPS = {
    "stereo_9_10": lambda: sg.stereo_setup(22050, 9, 10),
    "stereo_8_10_t1": lambda: sg.stereo_setup(22050, 8, 10, residue_type=1),
    "surround51_8_10": _surround51_8_10,
    "mono_7_10": lambda: sg.mono_setup(7, 10, 16000),
    "uncoupled_8_10": _uncoupled_8_10,
}


def _decoder(setup):
    from lewton_amd import audio, header
    idp, _, stp = setup.headers()
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    return audio, audio.decoder_for(ident, st)


def _decode(dec, audio, streams, cuts, fmt, l10=-1, rounds=0, mix=-1):
    """streams[s] = packets; cuts = batch boundaries in packets-per-stream ([0, 3, 10, ...]): every batch holds packets cuts[k] ..
    cuts[k+1] of every stream, stream-major.  Returns (per packet arrays, kernels seen, final window states)."""
    from lewton_amd.batch import Batch
    ch = dec.ident.audio_channels
    pwrs = [audio.PreviousWindowRight() for _ in streams]
    out = [[None] * len(s) for s in streams]
    seen = set()
    cap = max(b - a for a, b in zip(cuts[:-1], cuts[1:])) * len(streams)
    bt = Batch(dec, cap, fmt)
    bt.debug_set_long10(l10)
    bt.debug_set_mix(mix)
    if rounds:
        bt.debug_set_rounds(rounds)
    for a, b in zip(cuts[:-1], cuts[1:]):
        items = [(s, t) for s in range(len(streams)) for t in range(a, min(b, len(streams[s])))]
        res = bt.entropy([(streams[s][t], pwrs[s]) for s, t in items], n_threads=2)
        bt.upload()
        got = bt.split(bt.synth_to_host(), ch)
        seen.update(bt.last_kernels.split(","))
        for (s, t), r, g in zip(items, res, got):
            out[s][t] = (r[0], None if g is None else g.copy())
    states = [p.data().copy() for p in pwrs]
    bt.close()
    return out, seen, states


def _oracle(setup, streams, fmt):
    o_id, o_st = oracle_headers(setup)
    ofmt = {"i16": "i16", "f32": "f32", "i16_interleaved": "i16_itl"}[fmt]
    out, states = [], []
    for pk in streams:
        opw = po.Pwr()
        row = []
        for p in pk:
            try:
                row.append((0, np.asarray(po.read_audio_packet(o_id, o_st, p, opw, ofmt))))
            except po.OracleError as e:
                row.append((e.code, None))
        out.append(row)
        states.append(opw.data(setup.channels))
    return out, states


def _same(a, b, fmt):
    if a is None or b is None:
        return a is None and b is None
    a, b = np.asarray(a).reshape(-1), np.asarray(b).reshape(-1)
    if a.size != b.size:
        return False
    return np.array_equal(a.view(np.uint32), b.view(np.uint32)) if fmt == "f32" else np.array_equal(a, b)


def _compare(got, want, fmt, tag):
    for s, (gr, wr) in enumerate(zip(got, want)):
        for t, ((gs, g), (ws, w)) in enumerate(zip(gr, wr)):
            assert gs == ws, (tag, s, t, gs, ws)
            assert _same(g, w, fmt), (tag, s, t)


@pytest.mark.parametrize("name", sorted(L10_SETUPS))
@pytest.mark.parametrize("fmt", ["i16", "f32", "i16_interleaved"])
def test_long10_all_long_streams_vs_oracle_and_block_kernel(name, fmt):
    setup = L10_SETUPS[name]()
    audio, dec = _decoder(setup)
    n_streams = 37                      # not a multiple of anything: chunks cut streams, the last workgroup is partly idle
    streams = [sg.make_stream(setup, "L", 23 + (s % 5), seed=4100 + s, p_floor_unused=0.08) for s in range(n_streams)]
    streams[3][7] = streams[3][7][: len(streams[3][7]) // 2]      # a truncated packet (end of packet inside the residue)
    want, wstates = _oracle(setup, streams, fmt)
    cuts = [0, 1, 2, 9, 28]             # one packet per stream per launch twice (state pool), then runs
    got, seen, states = _decode(dec, audio, streams, cuts, fmt)
    assert "k_long10" in seen and not any(k.startswith("k_short") for k in seen), seen
    _compare(got, want, fmt, name)
    for s in range(n_streams):
        assert np.array_equal(states[s].view(np.uint32), wstates[s].view(np.uint32)), s
    # the block kernel on the same batches: identical bytes
    got2, seen2, states2 = _decode(dec, audio, streams, cuts, fmt, l10=0)
    assert "k_short" in seen2 and "k_long10" not in seen2, seen2
    _compare(got2, want, fmt, name + " (k_short<32>)")


@pytest.mark.parametrize("rounds", [1, 2, 3, 5])
def test_long10_forced_rounds_hand_over_paths(rounds):
    """the same streams with 1 / 2 / 3 / 5 rounds per workgroup: right halves through LDS inside a round and from the last wave of a
    round to the first of the next, through the halo pre-pass where a chunk starts inside a stream, through the state pool between
    launches"""
    setup = L10_SETUPS["stereo_9_10"]()
    audio, dec = _decoder(setup)
    streams = [sg.make_stream(setup, "L", 40 + 3 * (s % 4), seed=4300 + s) for s in range(11)]
    want, wstates = _oracle(setup, streams, "i16")
    got, seen, states = _decode(dec, audio, streams, [0, 5, 52], "i16", rounds=rounds)
    assert "k_long10" in seen, seen
    if rounds < 5:
        assert "k_long10<halo>" in seen, seen
    _compare(got, want, "i16", "rounds=%d" % rounds)
    for s in range(len(streams)):
        assert np.array_equal(states[s].view(np.uint32), wstates[s].view(np.uint32)), s


@pytest.mark.parametrize("name", ["stereo_9_10", "stereo_8_10_t1", "surround51_8_10", "uncoupled_8_10"])
@pytest.mark.parametrize("fmt", ["i16", "f32", "i16_interleaved"])
@pytest.mark.parametrize("mode", [-1, 1])
def test_long10_mixed_short_long_streams(name, fmt, mode):
    """mixed short/long streams (512/1024 and 256/1024 points), mode -1: the long blocks in k_long10 -- those with a short slope in
    its EDGE form, which leaves the raw edges for k_short -- and the short blocks in k_short<8 / 16>, no generic kernel, no time-domain
    block through HBM; mode 1: the long blocks next to short ones through the generic kernels instead (right halves cross between
    all three paths through time-domain blocks and the state pool).  Batch cuts put every window shape at a launch boundary (the
    short state written by one path, read by another)."""
    setup = L10_SETUPS[name]()
    audio, dec = _decoder(setup)
    pats = ["LLLSSSLLLL", "LLSLLLSSLLLLL", "LSSSSSSLLL", "LLLLLLLSL", "SLSLLSSL"]
    streams = [sg.make_stream(setup, pats[s % 5], 30 + s % 3, seed=4500 + s, p_floor_unused=0.05) for s in range(13)]
    streams[5][9] = streams[5][9][: len(streams[5][9]) // 3]
    want, wstates = _oracle(setup, streams, fmt)
    cuts = [0, 1, 2, 3, 4, 5, 6, 7, 8, 17, 33]
    got, seen, states = _decode(dec, audio, streams, cuts, fmt, l10=mode)
    assert seen & {"k_long10", "k_mix10"} and seen & {"k_short", "k_mix10"}, seen   # (k_mix10: both roles in one launch where the chip holds it)
    assert any("generic" in k for k in seen) == (mode == 1), seen
    _compare(got, want, fmt, "%s mode %d" % (name, mode))
    for s in range(len(streams)):
        assert np.array_equal(states[s].view(np.uint32), wstates[s].view(np.uint32)), s


def test_long10_dense_bench_shapes():
    """the shapes tools/bench_configs.py times for blocksize_1 = 10: 256 streams x 16 (one round) and x 48 (three rounds)"""
    from test_gpu_quoted_shapes import _run_dense
    for packets in (4096, 12288):
        bad, kernels, n = _run_dense("12", "i16", packets=packets)
        assert n == packets and bad == 0 and kernels == "k_long10", (bad, kernels)


@pytest.mark.parametrize("name", ["stereo_9_10", "stereo_8_10_t1", "surround51_8_10"])
@pytest.mark.parametrize("fmt", ["i16", "f32", "i16_interleaved"])
def test_mix10_one_launch_equals_two_launches(name, fmt):
    """k_mix10 (the long and the short blocks of a mixed blocksize_1 = 10 batch in ONE launch, the short blocks' waves waiting for the
    long blocks' raw edges through flags in HBM) against the same batches as two launches (k_long10<EDGE>, k_short<8 / 16>) and
    against the oracle: identical samples and states, also across re-planned batches of the same Batch object"""
    setup = L10_SETUPS[name]()
    audio, dec = _decoder(setup)
    pats = ["LLLSSSLLLL", "LLSLLLSSLLLLL", "LSSSSSSLLL", "LLLLLLLSL", "SLSLLSSL"]
    streams = [sg.make_stream(setup, pats[s % 5], 40, seed=4700 + s, p_floor_unused=0.05) for s in range(24)]
    want, wstates = _oracle(setup, streams, fmt)
    cuts = [0, 1, 14, 27, 40]
    got1, seen1, st1 = _decode(dec, audio, streams, cuts, fmt)
    got2, seen2, st2 = _decode(dec, audio, streams, cuts, fmt, mix=0)
    assert "k_mix10" in seen1 and "k_mix10" not in seen2 and {"k_long10", "k_short"} <= seen2, (seen1, seen2)
    _compare(got1, want, fmt, name + " (k_mix10)")
    _compare(got2, want, fmt, name + " (two launches)")
    for s in range(len(streams)):
        assert np.array_equal(st1[s].view(np.uint32), wstates[s].view(np.uint32)) and np.array_equal(st2[s].view(np.uint32), wstates[s].view(np.uint32)), s


def test_mix10_bench_shapes_and_error_exit():
    """the two mixed lines of the bench (256 streams x 16 packets of LLLSSSLLLL at 512/1024 and 256/1024 points) are ONE launch each;
    and a k_mix10 launch whose producers never signal ends in LW_ERR_DEVICE for the batch, not in samples"""
    from lewton_amd import _native as N
    from lewton_amd.batch import Batch
    from test_gpu_quoted_shapes import _run_dense
    for key in ("14", "15"):
        bad, kernels, n = _run_dense(key, "i16", packets=4096)
        assert n == 4096 and bad == 0 and kernels == "k_mix10", (key, bad, kernels)
    setup = L10_SETUPS["stereo_9_10"]()
    audio, dec = _decoder(setup)
    streams = [sg.make_stream(setup, "LLLSSSLLLL", 20, seed=4900 + s) for s in range(16)]
    pwrs = [audio.PreviousWindowRight() for _ in streams]
    bt = Batch(dec, 16 * 20, "i16")
    bt.entropy([(streams[s][t], pwrs[s]) for s in range(16) for t in range(20)], n_threads=2)
    bt.upload()
    bt.debug_break_mix(2000)
    with pytest.raises(RuntimeError) as ei:
        bt.synth_to_host()
    assert "lw_batch_synth_to_host: %d" % N.ERR_DEVICE in str(ei.value) and "k_mix10" in bt.last_kernels
    assert all(r[0] == N.ERR_DEVICE and r[1] == 0 for r in bt.results())
    bt.close()


@pytest.mark.parametrize("fmt", ["i16", "i16_interleaved"])
def test_long10_single_mode_stream_with_equal_block_sizes(fmt):
    """blocksize_0 = blocksize_1 = 10 with ONE mode (no block flag): what libvorbis writes at 16 / 22 kHz, lowest quality.  Every block is
    a 1024-point block with two full slopes: k_long10 on the "short" class's units and image (the block kernel k_short<32> when
    switched off)"""
    setup = sg.stereo_setup(22050, 10, 10)
    setup.modes = [sg.Mode(0, 0)]
    audio, dec = _decoder(setup)
    streams = []
    for s in range(19):
        pw = sg.PacketWriter(setup, 4950 + s, p_floor_unused=0.05)
        streams.append([pw.packet(0) for _ in range(21 + s % 4)])
    want, wstates = _oracle(setup, streams, fmt)
    for l10, kernel in ((-1, "k_long10"), (0, "k_short")):
        got, seen, states = _decode(dec, audio, streams, [0, 1, 2, 11, 25], fmt, l10=l10)
        assert kernel in seen and not any("generic" in k for k in seen), (l10, seen)
        _compare(got, want, fmt, "single mode, l10=%d" % l10)
        for s in range(len(streams)):
            assert np.array_equal(states[s].view(np.uint32), wstates[s].view(np.uint32)), s
