"""GPU parity at the shapes the bench line and BASELINE.json quote (-m gpu), every packet against the ORACLE (not
against the generic kernels):

* configs[1] as bench.py lays it out: 256 primed streams x 16 consecutive long packets = one dense 4096-packet launch;
* configs[4] stepping shape: thousands of independent streams, ONE packet per stream per launch, the window state making
  a round trip through the HBM state pool between launches (28 804 B/packet);
* configs[4] sharded: 10 000 streams x 4 packets, `stream_id mod 8` onto 8 logical shards (all on device 0 here: one
  lw_decoder per shard, as one process per GPU would hold), reassembled in stream order;
* two decoders alive in one process (same device twice, and every other device the box has), decoding interleaved.

The oracle decodes ~30 k packets/s on one core, so even the 40 000-packet case is seconds of CPU."""
import os

import numpy as np
import pytest

from common import SETUPS, oracle_headers, po, sg

pytestmark = pytest.mark.gpu


def _product(setup):
    from lewton_amd import audio, header
    idp, _, stp = setup.headers()
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    return audio, ident, st


OFMT = {"i16": "i16", "f32": "f32", "i16_interleaved": "i16_itl"}


def _same(got, want, fmt):
    if fmt == "f32":
        return np.array_equal(np.asarray(got).view(np.uint32).reshape(-1), np.asarray(want).view(np.uint32).reshape(-1))
    return np.array_equal(np.asarray(got).reshape(-1), np.asarray(want).reshape(-1))


@pytest.mark.parametrize("fmt", ["i16", "f32", "i16_interleaved"])
def test_dense_bench_batch_every_packet_vs_oracle(fmt):
    """bench.py's timed shape: 256 streams, each primed with one packet in an earlier launch, then 16 consecutive long
    packets per stream in ONE dense launch of 4096 (one workgroup per stream, LDS hand-over along the stream)."""
    from lewton_amd.batch import Batch
    setup = SETUPS["stereo"]()
    audio, ident, st = _product(setup)
    o_id, o_st = oracle_headers(setup)
    dec = audio.decoder_for(ident, st)
    S, per = 256, 16
    pool = sg.make_stream(setup, "L", 192, seed=4242, p_floor_unused=0.03)
    rng = np.random.default_rng(12)
    pidx = rng.integers(0, len(pool), S)
    order = rng.integers(0, len(pool), S * per)
    pwrs = [audio.PreviousWindowRight() for _ in range(S)]
    prime = Batch(dec, S, fmt)
    prime.entropy([(pool[int(i)], pw) for i, pw in zip(pidx, pwrs)], n_threads=4)
    prime.upload()
    assert prime.synth_to_host().size == 0                   # a first packet yields no samples (audio.rs:1140-1152)
    bt = Batch(dec, S * per, fmt)
    res = bt.entropy([(pool[int(i)], pwrs[k // per]) for k, i in enumerate(order)], n_threads=4)
    bt.upload()
    flat = bt.synth_to_host()
    assert bt.last_kernels == "k_long"
    assert all(r[0] == 0 and r[1] == 1024 for r in res) and flat.size == S * per * 2 * 1024
    assert bt.algorithmic_bytes == S * per * (12420 + (4096 if fmt == "f32" else 0))
    # every stream's stored right part comes in from the state pool and the new one goes out: 2 channels x 1024 floats each way;
    # the priming batch had nothing to read
    assert bt.state_bytes == S * 2 * (2 * 1024 * 4) and prime.state_bytes == S * (2 * 1024 * 4)
    bad = 0
    for s in range(S):
        opw = po.Pwr()
        po.read_audio_packet(o_id, o_st, pool[int(pidx[s])], opw, OFMT[fmt])
        for k in range(s * per, (s + 1) * per):
            want = po.read_audio_packet(o_id, o_st, pool[int(order[k])], opw, OFMT[fmt])
            bad += not _same(flat[k * 2048:(k + 1) * 2048], want, fmt)
        if s % 37 == 0:                                      # the state the launch left in the pool = the oracle's pwr
            assert np.array_equal(pwrs[s].data().view(np.uint32), opw.data(2).view(np.uint32)), s
    assert bad == 0


@pytest.mark.parametrize("name,n_streams", [("stereo", 4096), ("surround51", 700)])
def test_one_packet_per_stream_per_launch_state_round_trip(name, n_streams):
    """Independent-stream stepping (BASELINE configs[4] per launch): launch t decodes packet t of EVERY stream, so each
    packet's previous right half comes from the state pool in HBM and its own goes back there."""
    from lewton_amd.batch import Batch
    setup = SETUPS[name]()
    audio, ident, st = _product(setup)
    o_id, o_st = oracle_headers(setup)
    dec = audio.decoder_for(ident, st)
    ch = setup.channels
    steps = 4
    pool = sg.make_stream(setup, "L", 96, seed=99, p_floor_unused=0.05)
    rng = np.random.default_rng(5)
    pick = rng.integers(0, len(pool), (steps, n_streams))
    pwrs = [audio.PreviousWindowRight() for _ in range(n_streams)]
    opws = [po.Pwr() for _ in range(n_streams)]
    bt = Batch(dec, n_streams, "i16")
    for t in range(steps):
        res = bt.entropy([(pool[int(pick[t, s])], pwrs[s]) for s in range(n_streams)], n_threads=4)
        bt.upload()
        got = bt.split(bt.synth_to_host(), ch)
        assert "k_long" in bt.last_kernels
        for s in range(n_streams):
            want = po.read_audio_packet(o_id, o_st, pool[int(pick[t, s])], opws[s], "i16")
            assert res[s][0] == 0 and got[s].shape == want.shape and np.array_equal(got[s], want), (t, s)
    for s in range(0, n_streams, 97):
        assert np.array_equal(pwrs[s].data().view(np.uint32), opws[s].data(ch).view(np.uint32))


def test_ten_thousand_streams_sharded_mod_8():
    """BASELINE configs[4] at reduced length: 10 000 independent stereo streams x 4 packets, sharded `stream_id mod 8`
    (lewton_amd/shard.py = the rule `lw_sharder` applies) onto 8 logical shards, each with its own decoder, state pool
    and batch as a rank of an 8-GPU job would have; every packet of every stream against the oracle."""
    from lewton_amd import shard
    from lewton_amd.batch import Batch
    setup = SETUPS["stereo"]()
    audio, ident, st = _product(setup)
    o_id, o_st = oracle_headers(setup)
    G, n_streams, per = 8, 10000, 4
    pool = sg.make_stream(setup, "L", 128, seed=7)
    rng = np.random.default_rng(21)
    pick = rng.integers(0, len(pool), (n_streams, per))
    decs = [audio.Decoder(ident, st, 0) for _ in range(G)]   # 8 separate lw_decoder objects (device 0 eight times)
    total_ok = 0
    for g in range(G):
        mine = shard.shard_streams(n_streams, G, g)
        assert all(shard.owner_of(s, G) == g for s in mine[:5])
        pwrs = {s: audio.PreviousWindowRight() for s in mine}
        bt = Batch(decs[g], len(mine) * per, "i16")
        # stream-major inside the shard: the 4 packets of a stream are consecutive (LDS hand-over inside the launch)
        items = [(pool[int(pick[s, t])], pwrs[s]) for s in mine for t in range(per)]
        res = bt.entropy(items, n_threads=4)
        bt.upload()
        got = bt.split(bt.synth_to_host(), 2)
        k = 0
        for s in mine:
            opw = po.Pwr()
            for t in range(per):
                want = po.read_audio_packet(o_id, o_st, pool[int(pick[s, t])], opw, "i16")
                assert res[k][0] == 0 and got[k].shape == want.shape and np.array_equal(got[k], want), (g, s, t)
                k += 1
                total_ok += 1
        bt.close()
    assert total_ok == n_streams * per


def test_two_decoders_in_one_process_interleaved():
    """Two (and, with more GPUs, more) lw_decoder objects alive at once, decoding alternately: kernel attributes are per
    device (152 KB dynamic LDS opt-in of k_long), state pools and streams must not leak between decoders."""
    from lewton_amd import _native as N
    from lewton_amd.batch import Batch
    setup = SETUPS["stereo"]()
    audio, ident, st = _product(setup)
    o_id, o_st = oracle_headers(setup)
    devices = [0, 0] + list(range(1, N.lw_device_count()))
    decs = [audio.Decoder(ident, st, d) for d in devices]
    streams = [sg.make_stream(setup, "LLLLSLLL", 24, seed=60 + i) for i in range(len(decs))]
    pwrs = [audio.PreviousWindowRight() for _ in decs]
    opws = [po.Pwr() for _ in decs]
    bts = [Batch(d, 8, "i16") for d in decs]
    for r in range(3):
        for i in range(len(decs)):                           # alternate between the decoders inside every round
            pk = streams[i][r * 8:(r + 1) * 8]
            bts[i].entropy([(p, pwrs[i]) for p in pk], n_threads=1)
            bts[i].upload()
            got = bts[i].split(bts[i].synth_to_host(), 2)
            for p, g in zip(pk, got):
                want = po.read_audio_packet(o_id, o_st, p, opws[i], "i16")
                assert g.shape == want.shape and np.array_equal(g, want), (r, i)
    # packet by packet through the drop-in call as well, alternating decoders
    for i in range(len(decs)):
        p = sg.make_stream(setup, "L", 3, seed=80 + i)
        for q in p:
            a = audio.read_audio_packet_on(decs[i], q, pwrs[i])
            w = po.read_audio_packet(o_id, o_st, q, opws[i], "i16")
            assert a.shape == w.shape and np.array_equal(a, w)


def test_sharder_ten_thousand_streams_one_process():
    """The same 10 000 streams through the C-level sharder (lw_sharder_*): eight logical shards, one worker thread, decoder,
    batch and HIP stream each (every visible device in turn, so on a one-GPU box all on device 0), two calls of two packets
    per stream with the state carried on each shard between the calls; every packet against the oracle."""
    from lewton_amd import _native as N
    from lewton_amd import shard
    setup = SETUPS["stereo"]()
    audio, ident, st = _product(setup)
    o_id, o_st = oracle_headers(setup)
    G, n_streams = 8, 10000
    ndev = max(1, N.lw_device_count())
    sh = shard.Sharder(ident, st, [g % ndev for g in range(G)], max_packets_per_shard=2 * (n_streams // G + 1), samples="i16")
    assert sh.shards == G and sh.shard_of(12345) == 12345 % G
    pool = sg.make_stream(setup, "L", 128, seed=17)
    rng = np.random.default_rng(4)
    pick = rng.integers(0, len(pool), (n_streams, 4))
    opws = [po.Pwr() for _ in range(n_streams)]
    for half in range(2):
        # round-robin over the streams inside a call: the sharder sorts out who owns what, stream order is kept
        items = [(s, pool[int(pick[s, 2 * half + t])]) for t in range(2) for s in range(n_streams)]
        blocks, res = sh.decode(items, n_threads=8)
        for (s, pkt), b, r in zip(items, blocks, res):
            want = po.read_audio_packet(o_id, o_st, pkt, opws[s], "i16")
            assert r[0] == 0 and b.shape == want.shape and np.array_equal(b, want), (half, s)
    sh.close()


def test_two_tenants_on_one_gpu():
    """Two logical shards on one device are tenants of it: their rings hand the PCM copies to the device's copier thread (a copy
    queued behind its kernels would block the copy engine for the other ring's ready copies) and run their launches' kernels in
    launch order.  Both shards busy at once, three calls in flight, entropy stage on the device; every packet against the
    oracle (tests/tenants_worker.py).  The variant with CU shares on top (lw_decoder_set_cu_share: 16 of the 32 CUs of every XCD
    each, CU-masked HIP streams, launches planned for 128 CUs -- a dense 4096-packet batch = two rounds per workgroup with the LDS
    hand-over where the whole device runs one) is the OTHER of the two modes a process can be in (include/lewton_amd.h: the
    library keeps CU-masked streams and the copier's own stream out of one process), so it runs as a process of its own; that
    process also checks the share of a lone decoder, CUs [32 j / k, 32 (j + 1) / k) of each XCD.  In THIS process the copier's
    stream exists by now, and a CU share is refused."""
    import subprocess
    import sys
    from lewton_amd import _native as N
    import tenants_worker
    assert tenants_worker.run(False) == 512 * 16 * 5
    setup = SETUPS["stereo"]()
    audio, ident, st = _product(setup)
    dec = audio.Decoder(ident, st, 0)
    assert N.lw_decoder_set_cu_share(dec._h, 0, 2) == N.ERR_UNSUPPORTED and N.lw_decoder_set_cu_share(dec._h, 0, 1) == 0
    dec.close()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tenants_worker.py")
    out = subprocess.run([sys.executable, worker, "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "TENANTS_OK 40960 packets" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


@pytest.mark.parametrize("tier", ["host", "device"])
def test_sharder_pipelined_calls_in_flight(tier):
    """lw_sharder_submit / lw_sharder_collect: three logical shards, each on its own staging ring, up to three calls in
    flight (the host stage of call k+1 overlaps the GPU work of call k on every shard), the same streams carried from call
    to call, mixed block sizes; every packet of every call against the oracle, and the states at the end."""
    from lewton_amd import _native as N
    from lewton_amd import shard
    setup = SETUPS["stereo"]()
    audio, ident, st = _product(setup)
    o_id, o_st = oracle_headers(setup)
    G, n_streams, per, n_calls = 3, 30, 6, 7
    ndev = max(1, N.lw_device_count())
    sh = shard.Sharder(ident, st, [g % ndev for g in range(G)], max_packets_per_shard=per * (n_streams // G + 1), samples="i16")
    if tier == "device":
        assert sh.set_entropy_on_device(True)
    streams = [sg.make_stream(setup, "LLSSSLLSL", per * n_calls, seed=300 + s, p_floor_unused=0.04) for s in range(n_streams)]
    opws = [po.Pwr() for _ in range(n_streams)]
    calls = [[(s, streams[s][c * per + t]) for s in range(n_streams) for t in range(per)] for c in range(n_calls)]
    pending = []

    def check(items, flat, res):
        for (s, pkt), (status, m, off) in zip(items, res):
            want = po.read_audio_packet(o_id, o_st, pkt, opws[s], "i16")
            assert status == 0 and m == want.shape[1]
            assert np.array_equal(flat[off:off + 2 * m], want.reshape(-1)), s

    def check_pinned(items):   # the zero-copy form: per-shard views of the pinned PCM, offsets relative to the owner's block
        views, res = sh.collect_pinned()
        for (s, pkt), (status, m, off) in zip(items, res):
            want = po.read_audio_packet(o_id, o_st, pkt, opws[s], "i16")
            assert status == 0 and m == want.shape[1]
            assert np.array_equal(views[sh.shard_of(s)][off:off + 2 * m], want.reshape(-1)), s
        sh.release()

    taken = 0

    def take():
        nonlocal taken
        if taken % 2:
            check_pinned(pending.pop(0))
        else:
            flat, res = sh.collect()
            check(pending.pop(0), flat, res)
        taken += 1

    for items in calls:
        if sh.in_flight == 3:
            take()
        sh.submit(sh.marshal(items), n_threads=2)
        pending.append(items)
    assert sh.in_flight == 3
    while pending:
        take()
    assert sh.in_flight == 0
    # the synchronous form still works on the drained pipeline
    tail = [(s, sg.make_stream(setup, "L", 1, seed=900 + s)[0]) for s in range(n_streams)]
    blocks, res = sh.decode(tail, n_threads=2)
    for (s, pkt), b in zip(tail, blocks):
        want = po.read_audio_packet(o_id, o_st, pkt, opws[s], "i16")
        assert b.shape == want.shape and np.array_equal(b, want)
    sh.close()
