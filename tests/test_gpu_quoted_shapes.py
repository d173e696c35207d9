"""Every shape a number is quoted for (profiles/*_other_configs.jsonl, bench `end_to_end`) checked packet by
packet against the ORACLE at that very shape (-m gpu):

* BASELINE configs[2] as tools/bench_configs.py lays it out: 256 streams x 16 consecutive packets of `LLSSSSSSSSL`
  in ONE launch (mixed short/long windows, state carried inside the launch);
* BASELINE configs[3] dense: 256 streams x 16 packets of 5.1 @ 48 kHz (two coupled pairs + one uncoupled pair per packet);
* the stereo stream with its coupling list removed (uncoupled-pair units of k_long), mono (single-channel units);
* other long block sizes at the quoted shape;
* one 4096-packet batch through the staging ring per record tier (host entropy stage / k_entropy), PCM compared.

The workload definitions are the ones tools/bench_configs.py times (lewton_amd/workloads.py)."""
import numpy as np
import pytest

from common import po, sg, verify_workload_batch  # noqa: F401

pytestmark = pytest.mark.gpu


def _decoder(setup):
    from lewton_amd import audio, header
    idp, _, stp = setup.headers()
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    return audio, audio.decoder_for(ident, st)


def _run_dense(key, fmt="i16", packets=4096, l10=-1):
    from lewton_amd import workloads as wl
    from lewton_amd.batch import Batch
    w = wl.by_key(key, packets)
    setup = w.setup()
    audio, dec = _decoder(setup)
    seqs = wl.stream_material(w, setup, batch=3)
    pwrs = [audio.PreviousWindowRight() for _ in range(w.n_streams)]
    prime_items, items = wl.items_of(w, seqs, pwrs)
    prime = Batch(dec, w.n_streams, fmt)
    prime.entropy(prime_items, n_threads=4)
    prime.upload()
    prime.synth_to_host()
    prime.close()
    bt = Batch(dec, len(items), fmt)
    bt.debug_set_long10(l10)
    res = bt.entropy(items, n_threads=4)
    bt.upload()
    flat = bt.synth_to_host()
    kernels = bt.last_kernels
    bad = verify_workload_batch(w, setup, seqs, res, flat, fmt)
    bt.close()
    return bad, kernels, len(items)


@pytest.mark.parametrize("fmt", ["i16", "f32", "i16_interleaved"])
def test_configs2_mixed_short_long_bench_shape(fmt):
    bad, kernels, n = _run_dense("3", fmt)
    assert n == 4096 and bad == 0, (bad, kernels)
    assert kernels == "k_mix"               # ONE launch (the chip holds the whole grid), no generic kernel, no time-domain block through HBM


@pytest.mark.parametrize("fmt", ["i16", "f32"])
def test_configs3_surround51_dense_bench_shape(fmt):
    bad, kernels, n = _run_dense("4", fmt)
    assert n == 4096 and bad == 0, (bad, kernels)
    assert kernels == "k_long"


@pytest.mark.parametrize("key", ["6", "7"])
def test_uncoupled_pair_and_single_channel_units_bench_shape(key):
    bad, kernels, n = _run_dense(key)
    assert n == 4096 and bad == 0, (bad, kernels)
    assert kernels == "k_long"


@pytest.mark.parametrize("key", ["11", "12", "13"])
def test_other_long_block_sizes_bench_shape(key):
    bad, kernels, n = _run_dense(key, packets=1024)
    assert n == 1024 and bad == 0, (bad, kernels)
    assert kernels == {"12": "k_long10", "11": "k_long12", "13": "k_big"}[key]


@pytest.mark.parametrize("key,fmt", [("11", "i16"), ("11", "f32"), ("13", "i16_interleaved")])
def test_long_blocks_of_4096_and_8192_points_dense_bench_shape(key, fmt):
    """256 streams x 16 long blocks in one launch through k_big: runs of consecutive blocks per workgroup (right parts stay in
    the threads' registers), a recomputed predecessor where a run starts inside a stream (4096 points: the default path is
    k_long12, tests/test_gpu_long12.py; k_big<12> stays as the independent second implementation)"""
    bad, kernels, n = _run_dense(key, fmt, packets=4096, l10=0)
    assert n == 4096 and bad == 0, (bad, kernels)
    assert kernels == "k_big"


@pytest.mark.parametrize("tier,pattern", [("host", "L"), ("device", "L"), ("host", "LLSSSSSSSSL"), ("device", "LLSSSSSSSSL")])
def test_ring_4096_packet_batches_pcm_vs_oracle(tier, pattern):
    """the end_to_end object of the bench line is a rate of THESE bytes: three 4096-packet batches through a 3-slot ring,
    the same 256 streams in every batch, every packet's PCM against the oracle"""
    from lewton_amd import workloads as wl
    from lewton_amd.ring import Ring
    w = wl.Workload("r", "ring", lambda: sg.stereo_setup(44100, 8, 11), pattern, 256, 16, "", distinct=32)
    setup = w.setup()
    audio, dec = _decoder(setup)
    n_batches = 3
    seqs = [sg.make_stream(setup, pattern, n_batches * w.per_stream + 1, seed=900 + s) for s in range(w.distinct)]
    pwrs = [audio.PreviousWindowRight() for _ in range(w.n_streams)]
    ring = Ring(dec, 3, w.n_streams * w.per_stream, "i16")
    if tier == "device":
        assert ring.set_entropy_on_device(True)
    ring.submit(ring.marshal([(seqs[s % len(seqs)][0], pwrs[s]) for s in range(w.n_streams)]), n_threads=4)
    ring.collect()
    ring.release()
    for b in range(n_batches):
        items = [(seqs[s % len(seqs)][1 + b * w.per_stream + k], pwrs[s]) for s in range(w.n_streams) for k in range(w.per_stream)]
        ring.submit(ring.marshal(items), n_threads=4)
    if tier == "device":
        assert "k_entropy" in ring.last_kernels
    from common import oracle_headers
    o_id, o_st = oracle_headers(setup)
    expect = []
    for q in range(len(seqs)):
        opw = po.Pwr()
        po.read_audio_packet(o_id, o_st, seqs[q][0], opw, "i16")
        expect.append([np.asarray(po.read_audio_packet(o_id, o_st, p, opw, "i16")).reshape(-1) for p in seqs[q][1:]])
    for b in range(n_batches):
        res, pcm = ring.collect()
        bad, k = 0, 0
        for s in range(w.n_streams):
            for t in range(w.per_stream):
                want = expect[s % len(seqs)][b * w.per_stream + t]
                status, m, off = res[k]
                assert status == 0 and 2 * m == want.size, (b, s, t, status, m)
                bad += not np.array_equal(pcm[off:off + 2 * m], want)
                k += 1
        ring.release()
        assert bad == 0, (tier, pattern, b, bad)
    ring.close()


def _mono_8_11():
    from lewton_amd import workloads as wl
    return wl.mono()


def _uncoupled_8_11():
    from lewton_amd import workloads as wl
    return wl.uncoupled_stereo()


MIXED_SETUPS = {"stereo": lambda: sg.stereo_setup(44100, 8, 11), "surround51": lambda: sg.surround51_setup(48000, 8, 11),
                "mono": _mono_8_11, "uncoupled": _uncoupled_8_11}


BLK_SETUPS = {"stereo_9_10": lambda: sg.stereo_setup(22050, 9, 10), "stereo_8_10": lambda: sg.stereo_setup(22050, 8, 10, residue_type=1),
              "stereo_8_9": lambda: sg.stereo_setup(11025, 8, 9),
              "stereo_9_11": lambda: sg.stereo_setup(44100, 9, 11), "stereo_10_12": lambda: sg.stereo_setup(44100, 10, 12),
              # 4096 / 8192-point long blocks: k_big<12 / 13> (two long slopes), the generic kernels next to short blocks
              "stereo_9_12": lambda: sg.stereo_setup(44100, 9, 12), "stereo_6_13": lambda: sg.stereo_setup(44100, 6, 13),
              "stereo_8_13_t1": lambda: sg.stereo_setup(44100, 8, 13, residue_type=1),
              "surround51_9_12": lambda: sg.surround51_setup(48000, 9, 12), "mono_7_12": lambda: sg.mono_setup(7, 12, 44100)}
BIG = {"stereo_10_12", "stereo_9_12", "stereo_6_13", "stereo_8_13_t1", "surround51_9_12", "mono_7_12"}


@pytest.mark.parametrize("name", sorted(BLK_SETUPS))
@pytest.mark.parametrize("fmt", ["i16", "f32", "i16_interleaved"])
def test_block_kernel_streams_state_round_trip_and_runs(name, fmt):
    """Streams whose blocks run through k_short<L> in either role (512 / 1024-point long blocks with two long slopes, 256 /
    512 / 1024-point short blocks; the long blocks next to short ones through the generic kernels or k_long<TD>): first one
    packet per stream per launch (every right part through the state pool), then the rest of every stream in ONE launch
    (runs inside a wave, recomputed predecessors at wave boundaries, time-domain blocks exchanged with the generic kernels)."""
    from lewton_amd.batch import Batch
    from common import oracle_headers
    setup = BLK_SETUPS[name]()
    audio, dec = _decoder(setup)
    o_id, o_st = oracle_headers(setup)
    ch = setup.channels
    n_streams, steps, tail = 10, 9, 40
    streams = [sg.make_stream(setup, "LLLLSSSLLSLLLLLLLSSSSSSSSSL"[s % 7:], steps + tail, seed=170 + s, p_floor_unused=0.06)
               for s in range(n_streams)]
    pwrs = [audio.PreviousWindowRight() for _ in range(n_streams)]
    opws = [po.Pwr() for _ in range(n_streams)]
    ofmt = {"i16": "i16", "f32": "f32", "i16_interleaved": "i16_itl"}[fmt]
    seen = set()

    def check(bt, items):
        res = bt.entropy([(streams[s][t], pwrs[s]) for s, t in items], n_threads=2)
        bt.upload()
        got = bt.split(bt.synth_to_host(), ch)
        seen.update(bt.last_kernels.split(","))
        for i, (s, t) in enumerate(items):
            want = np.asarray(po.read_audio_packet(o_id, o_st, streams[s][t], opws[s], ofmt))
            assert res[i][0] == 0 and got[i].size == want.size, (s, t)
            if fmt == "f32":
                assert np.array_equal(got[i].reshape(-1).view(np.uint32), want.reshape(-1).view(np.uint32)), (s, t, bt.last_kernels)
            else:
                assert np.array_equal(got[i].reshape(-1), want.reshape(-1)), (s, t, bt.last_kernels)

    b1 = Batch(dec, n_streams, fmt)
    for t in range(steps):
        check(b1, [(s, t) for s in range(n_streams)])
    b2 = Batch(dec, n_streams * tail, fmt)
    check(b2, [(s, steps + t) for s in range(n_streams) for t in range(tail)])
    if name in BIG:
        assert ("k_big" if "13" in name else "k_long12") in seen, seen   # 8192 points: the workgroup pipeline; 4096: one wave per channel
    else:
        assert seen & {"k_short", "k_mix10"}, seen   # (k_mix10: k_short's work in the last waves of k_long10's launch)
    if name in ("stereo_9_10", "stereo_8_10"):
        assert seen & {"k_long10", "k_mix10"}, seen  # their long blocks (lw_long10.inc)
    for s in range(n_streams):
        assert np.array_equal(pwrs[s].data().view(np.uint32), opws[s].data(ch).view(np.uint32)), s
    b1.close()
    b2.close()


def test_long_blocks_of_1024_points_dense_bench_shape_block_kernel():
    """blocksize_1 = 10 (what libvorbis writes at 16-22 kHz): 256 streams x 16 long blocks in one launch through the block kernel
    k_short<32> (a block's predecessor recomputed in the slot in front: two slots per wave) -- the default path for this shape is
    k_long10 (tests/test_gpu_long10.py); the block kernel stays as the independent second implementation"""
    bad, kernels, n = _run_dense("12", "i16", packets=4096, l10=0)
    assert n == 4096 and bad == 0, (bad, kernels)
    assert kernels == "k_short"


def test_long_blocks_of_1024_points_three_passes_per_wave():
    """a launch with enough blocks that a wave of k_short<32> works through three passes of two slots (one recomputed predecessor
    per five blocks, right parts handed from pass to pass through the double-buffered LDS area): 256 streams x 48 blocks"""
    bad, kernels, n = _run_dense("12", "i16", packets=12288, l10=0)
    assert n == 12288 and bad == 0, (bad, kernels)
    assert kernels == "k_short"


@pytest.mark.parametrize("name", sorted(MIXED_SETUPS))
@pytest.mark.parametrize("fmt", ["i16", "f32", "i16_interleaved"])
def test_mixed_streams_one_packet_per_launch(name, fmt):
    """Every window shape with its previous right part coming from the STATE POOL (one packet per stream per launch): short
    after long and long after short across launches (the 128-sample state written by k_long<EDGE> / k_short, read by a
    k_short block or by the slot that only carries a stored right part to a long block's left edge), single-channel and
    uncoupled-pair units, all three sample formats; every packet and the final states against the oracle."""
    from lewton_amd.batch import Batch
    from common import oracle_headers
    setup = MIXED_SETUPS[name]()
    audio, dec = _decoder(setup)
    o_id, o_st = oracle_headers(setup)
    ch = setup.channels
    n_streams, steps = 12, 14
    streams = [sg.make_stream(setup, "LSLLSSLSSSL"[s % 5:] + "LS", steps, seed=70 + s, p_floor_unused=0.08) for s in range(n_streams)]
    pwrs = [audio.PreviousWindowRight() for _ in range(n_streams)]
    opws = [po.Pwr() for _ in range(n_streams)]
    ofmt = {"i16": "i16", "f32": "f32", "i16_interleaved": "i16_itl"}[fmt]
    bt = Batch(dec, n_streams, fmt)
    seen = set()
    for t in range(steps):
        res = bt.entropy([(streams[s][t], pwrs[s]) for s in range(n_streams)], n_threads=2)
        bt.upload()
        got = bt.split(bt.synth_to_host(), ch)
        seen.update(bt.last_kernels.split(","))
        for s in range(n_streams):
            want = np.asarray(po.read_audio_packet(o_id, o_st, streams[s][t], opws[s], ofmt))
            assert res[s][0] == 0 and got[s].size == want.size, (t, s)
            if fmt == "f32":
                assert np.array_equal(got[s].reshape(-1).view(np.uint32), want.reshape(-1).view(np.uint32)), (t, s, bt.last_kernels)
            else:
                assert np.array_equal(got[s].reshape(-1), want.reshape(-1)), (t, s, bt.last_kernels)
    assert ("k_mix" in seen or ("k_short" in seen and "k_long" in seen)) and not any("generic" in k for k in seen), seen
    for s in range(n_streams):
        assert np.array_equal(pwrs[s].data().view(np.uint32), opws[s].data(ch).view(np.uint32)), s
    bt.close()


def test_many_short_blocks_one_wave_per_channel_pair():
    """more than 1024 waves' worth of short blocks in one launch: k_short keeps a channel pair in ONE wave then (smaller
    launches split it over two); runs of 37 consecutive short blocks per stream cross wave boundaries (recomputed
    predecessor in the first slot)"""
    from lewton_amd.batch import Batch
    from common import oracle_headers
    setup = sg.stereo_setup(44100, 8, 11)
    audio, dec = _decoder(setup)
    o_id, o_st = oracle_headers(setup)
    n_streams, per, distinct = 256, 37, 16
    seqs = [sg.make_stream(setup, "S", per + 1, seed=500 + q, p_floor_unused=0.05) for q in range(distinct)]
    pwrs = [audio.PreviousWindowRight() for _ in range(n_streams)]
    prime = Batch(dec, n_streams, "i16")
    prime.entropy([(seqs[s % distinct][0], pwrs[s]) for s in range(n_streams)], n_threads=4)
    prime.upload()
    prime.synth_to_host()
    prime.close()
    bt = Batch(dec, n_streams * per, "i16")
    res = bt.entropy([(seqs[s % distinct][1 + k], pwrs[s]) for s in range(n_streams) for k in range(per)], n_threads=4)
    bt.upload()
    flat = bt.synth_to_host()
    assert bt.last_kernels == "k_short"
    expect = []
    for q in range(distinct):
        opw = po.Pwr()
        po.read_audio_packet(o_id, o_st, seqs[q][0], opw, "i16")
        expect.append(([np.asarray(po.read_audio_packet(o_id, o_st, p, opw, "i16")).reshape(-1) for p in seqs[q][1:]], opw))
    bad, k = 0, 0
    for s in range(n_streams):
        for t in range(per):
            want = expect[s % distinct][0][t]
            status, m, off = res[k]
            assert status == 0 and 2 * m == want.size
            bad += not np.array_equal(flat[off:off + 2 * m], want)
            k += 1
    assert bad == 0
    for s in range(0, n_streams, 17):
        assert np.array_equal(pwrs[s].data().view(np.uint32), expect[s % distinct][1].data(2).view(np.uint32))
    bt.close()


@pytest.mark.parametrize("fmt", ["i16", "f32", "i16_interleaved"])
@pytest.mark.parametrize("name", ["stereo", "surround51"])
def test_mixed_batch_as_one_launch_equals_two_launches(name, fmt):
    """k_mix (long and short blocks of a mixed batch in ONE launch, the short blocks' waves waiting for the long blocks' raw edges
    through flags in HBM) against the same batch as two launches (k_long<EDGE>, k_short): identical bytes, identical state,
    also when the same uploaded batch is launched again (the flags are back to zero) and after a re-plan of the same Batch."""
    from lewton_amd.batch import Batch
    from common import SETUPS
    setup = SETUPS[name]()
    audio, dec = _decoder(setup)
    n_streams, per = 24, 22
    streams = [sg.make_stream(setup, "LLSSSSSSSSLLSSLSL", 2 * per, seed=700 + s, p_floor_unused=0.05) for s in range(n_streams)]
    outs = {}
    for mode in (-1, 0):
        pwrs = [audio.PreviousWindowRight() for _ in range(n_streams)]
        bt = Batch(dec, n_streams * per, fmt)
        bt.debug_set_mix(mode)
        flats = []
        for half in range(2):                      # the second half re-plans the same Batch object (other window shapes, same buffers)
            bt.entropy([(streams[s][half * per + t], pwrs[s]) for s in range(n_streams) for t in range(per)], n_threads=2)
            bt.upload()
            flats.append(bt.synth_to_host().copy())
            assert ("k_mix" in bt.last_kernels) == (mode == -1), bt.last_kernels
            again = bt.synth_to_host()             # launching an uploaded batch again is idempotent (state parity, flags cleared)
            assert np.array_equal(again.view(np.uint8), flats[-1].view(np.uint8))
        outs[mode] = (flats, [p.data().copy() for p in pwrs])
    for a, b in zip(outs[-1][0], outs[0][0]):
        assert a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))
    for a, b in zip(outs[-1][1], outs[0][1]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_mix_launch_without_producers_is_an_error_not_samples():
    """k_mix's short blocks' waves wait for the raw edges of their long neighbours through flags in HBM.  A wave whose flags never come
    (here: the test hook makes every long block's wave skip its flags; in the field: a grid that is not resident at once) must end in a
    STATUS -- LW_ERR_DEVICE for the whole batch, every packet result marked, audio.rs:27-41 -- not in PCM computed from whatever the
    edge buffer held.  Afterwards the very same Batch object decodes correctly again (the error word and the flags are cleared)."""
    from lewton_amd import _native as N
    from lewton_amd.batch import Batch
    setup = sg.stereo_setup()
    audio, dec = _decoder(setup)
    n_streams, per = 24, 22
    streams = [sg.make_stream(setup, "LLSSSSSSSSLLSSLSL", per, seed=900 + s) for s in range(n_streams)]
    o_id = po.Ident(setup.headers()[0])
    o_st = po.Setup(setup.headers()[2], o_id)
    bt = Batch(dec, n_streams * per, "i16")
    pwrs = [audio.PreviousWindowRight() for _ in range(n_streams)]
    bt.entropy([(streams[s][t], pwrs[s]) for s in range(n_streams) for t in range(per)], n_threads=2)
    bt.upload()
    bt.debug_break_mix(2000)                 # ~2000 polls of a few hundred ns: milliseconds, not the default second
    with pytest.raises(RuntimeError) as ei:
        bt.synth_to_host()
    assert "lw_batch_synth_to_host: %d" % N.ERR_DEVICE in str(ei.value) and "k_mix" in bt.last_kernels
    assert all(r[0] == N.ERR_DEVICE and r[1] == 0 for r in bt.results())
    assert bt.device_status() == 0           # the word was consumed by the failing call
    # normal operation again, same Batch: fresh window states (those of the failed batch are undefined by contract)
    bt.debug_break_mix(0)
    pwrs = [audio.PreviousWindowRight() for _ in range(n_streams)]
    res = bt.entropy([(streams[s][t], pwrs[s]) for s in range(n_streams) for t in range(per)], n_threads=2)
    bt.upload()
    flat = bt.synth_to_host()
    assert "k_mix" in bt.last_kernels
    k = 0
    for s in range(n_streams):
        opw = po.Pwr()
        for t in range(per):
            want = np.asarray(po.read_audio_packet(o_id, o_st, streams[s][t], opw, "i16")).reshape(-1)
            status, m, off = res[k]
            assert status == 0 and m * 2 == want.size and np.array_equal(flat[off:off + m * 2], want), (s, t)
            k += 1
    bt.close()


@pytest.mark.parametrize("key,kernel,packets", [("18", "k_long", 1536), ("17", "k_mix", 4096), ("12", "k_long10", 1536), ("11", "k_long12", 1536)])
def test_one_stream_in_one_launch_halo_items_equal_the_pre_pass(key, kernel, packets):
    """SURVEY 8(d) config 3 as written -- ONE stream, thousands of consecutive packets in one launch -- cuts the stream over the
    chip's workgroups; every chunk that starts inside the stream needs its predecessor's right half (audio.rs:1082-1154).  Round 6
    recomputes that predecessor INSIDE the launch (an item of its own in front of the chunk: no samples, right half through LDS)
    instead of in a pre-pass launch.  Both forms against the oracle and against each other, on the bench's single-stream shapes and
    on one stream of 1024- / 4096-point long blocks."""
    import dataclasses
    from lewton_amd import audio, header
    from lewton_amd import workloads as wl
    from lewton_amd.batch import Batch
    # (sizes at which the planner does use the extra items: they must not cost every workgroup another round, nor be every third item)
    w = wl.by_key(key, packets)
    if w.n_streams != 1:
        w = dataclasses.replace(w, n_streams=1, per_stream=packets, distinct=1)
    setup = w.setup()
    idp, _, stp = setup.headers()
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    dec = audio.Decoder(ident, st, 0)
    seqs = wl.stream_material(w, setup)
    flats = {}
    for mode in (-1, 0):
        pw = [audio.PreviousWindowRight()]
        prime_items, items = wl.items_of(w, seqs, pw)
        prime = Batch(dec, 1, "i16")
        prime.entropy(prime_items, n_threads=1)
        prime.upload()
        prime.synth_to_host()
        prime.close()
        bt = Batch(dec, len(items), "i16")
        bt.debug_set_halo(mode)
        res = bt.entropy(items, n_threads=4)
        bt.upload()
        flat = bt.synth_to_host()
        assert kernel in bt.last_kernels and ("<halo>" in bt.last_kernels) == (mode == 0), (mode, bt.last_kernels)
        assert verify_workload_batch(w, setup, seqs, res, flat, "i16") == 0, mode
        flats[mode] = flat.copy()
        state = pw[0].data().copy()
        flats[(mode, "state")] = state
        bt.close()
        pw.clear()
    assert np.array_equal(flats[-1], flats[0]) and np.array_equal(flats[(-1, "state")].view(np.uint32), flats[(0, "state")].view(np.uint32))
    dec.close()
