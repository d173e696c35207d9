"""N > 1 path on CPU: two gloo processes shard streams `stream_id mod 2`, each runs the product's host entropy stage
on its own streams; together they must cover every packet exactly once and reproduce the single-process result."""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from lewton_amd import shard  # noqa: E402

N_STREAMS, PER_STREAM = 7, 6


def _digest_streams(stream_ids):
    """sha256 of the GPU-stage records (floor posts + residues) of every packet of the given streams."""
    from lewton_amd import audio, header, streamgen as sg
    setup = sg.stereo_setup()
    idp, _, stp = setup.headers()
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, 2, (8, 11))
    out = {}
    for s in stream_ids:
        h = hashlib.sha256()
        for p in sg.make_stream(setup, "LLSSL", PER_STREAM, seed=4000 + s):
            rec = audio.entropy_decode_host(ident, st, p)
            h.update(rec["floor"].tobytes())
            h.update(rec["residue"].tobytes())
        out[s] = h.hexdigest()
    return out


def _worker(rank, world, port, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    mine = shard.shard_streams(N_STREAMS, world, rank)
    digests = _digest_streams(mine)
    gathered = [None] * world
    dist.all_gather_object(gathered, digests)
    elapsed = shard.max_elapsed(1.0 + rank, dist)          # slowest rank wins
    value = shard.job_throughput(len(mine) * PER_STREAM, 1.0 + rank, dist)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        q.put((gathered, elapsed, value))


def test_shard_partition_properties():
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            ids = shard.shard_streams(10000, world, r)
            assert all(shard.owner_of(i, world) == r for i in ids)
            seen += ids
        assert sorted(seen) == list(range(10000))          # every stream exactly once (BASELINE configs[4])
        sizes = [len(shard.shard_streams(10000, world, r)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.shard_streams(4, 2, 2)


def test_two_process_gloo_sharding_matches_single_process():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, elapsed, value = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    merged = {}
    for d in gathered:
        assert not (set(d) & set(merged))                  # no stream decoded twice
        merged.update(d)
    assert merged == _digest_streams(range(N_STREAMS))     # same records as one process decoding everything
    assert elapsed == 2.0                                   # MAX over ranks
    assert np.isclose(value, (4 * PER_STREAM) * 2 / 2.0)    # rank 0 owns 4 streams: units_per_rank * world / max time
