"""End-to-end throughput of the decode path through the library's staging ring (lw_ring_*): host entropy stage (C++,
multi-threaded, writing into pinned staging) -> hipMemcpyAsync H2D -> synthesis kernels -> D2H of the PCM into pinned
host memory, with the entropy decode of batch N+1 overlapping the GPU work of batch N.

Not the BASELINE metric (that one is kernel-resident, bench.py `value`): this number is bounded by host cores and PCIe.
Used by tools/e2e.py and by bench.py's `end_to_end` object."""
import os
import threading
import time

import numpy as np

from . import audio
from . import _native as _N
from .ring import Ring


def measure(dec, pool, n_batches=32, packets=4096, streams=256, threads=0, slots=3, callers=1, seed=1,
            samples="i16", warm=8, device_entropy=False):
    """Returns a dict: packets/s, H2D / D2H GB/s, host entropy stage alone, kernels used.
    Every caller thread owns a ring and `streams` independent streams, each contributing packets/streams consecutive
    packets per batch (the bench workload, BASELINE configs[1])."""
    rng = np.random.default_rng(seed)
    per = packets // streams
    rings, work, payload = [], [], 0
    for c in range(callers):
        ring = Ring(dec, slots, packets, samples)
        if device_entropy and not ring.set_entropy_on_device(True):
            raise RuntimeError("stream not eligible for the device entropy stage")
        pwrs = [audio.PreviousWindowRight() for _ in range(streams)]
        batches = []
        for b in range(min(n_batches, 8)):       # 8 distinct batches, reused round-robin
            order = rng.integers(0, len(pool), packets)
            items = [(pool[int(i)], pwrs[k // per]) for k, i in enumerate(order)]
            if not payload:
                payload = sum(len(p) for p, _ in items)
            batches.append(ring.marshal(items))
        rings.append(ring)
        work.append((batches, pwrs))

    t_host = [0.0] * callers

    def run(c, n):
        ring, (batches, _) = rings[c], work[c]
        for k in range(n):
            if ring.in_flight == ring.slots:       # ring full: take the oldest batch out
                ring.collect_nocopy()
                ring.release()
            t0 = time.perf_counter()
            ring.stage(batches[k % len(batches)], threads)
            t_host[c] += time.perf_counter() - t0
            ring.launch()
        while ring.in_flight:
            ring.collect_nocopy()
            ring.release()

    def run_all(n):
        for c in range(callers):
            t_host[c] = 0.0
        if callers == 1:
            run(0, n)
        else:
            ts = [threading.Thread(target=run, args=(c, n)) for c in range(callers)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()

    run_all(warm)
    t0 = time.perf_counter()
    run_all(n_batches)
    dt = time.perf_counter() - t0
    npk = n_batches * packets * callers
    ch, half = dec.ident.audio_channels, (1 << dec.ident.blocksize_1) // 2
    rec_bytes = ch * half * 4 + 132 + 32   # f32 residues + floor records + packet record
    if device_entropy:
        rec_bytes = payload / packets + 8 + 16 + 32                     # the packet, its padding and descriptor, the packet record
    esz = 4 if samples == "f32" else 2
    out = {
        "value": npk / dt, "unit": "packets/s", "packets": npk, "seconds": dt,
        "records": "raw packets, entropy stage on the device (k_entropy)" if device_entropy else "f32 residue vectors (host entropy stage)",
        "h2d_GBps": npk * rec_bytes / dt / 1e9,
        "d2h_GBps": npk * ch * half * esz / dt / 1e9,
        "vorbis_payload_MBps": payload / packets * npk / dt / 1e6,
        "host_threads": threads or _N.lw_default_host_threads(), "callers": callers, "ring_slots": slots,
        "host_entropy_stage_alone": npk / max(t_host) if callers == 1 else npk / (sum(t_host) / callers),
        "kernels": rings[0].last_kernels,
        "path": "lw_ring_stage (host entropy decode into pinned staging) -> lw_ring_launch (hipMemcpyAsync H2D, kernels, "
                "hipMemcpyAsync D2H into pinned memory) -> lw_ring_collect, %d slots" % slots,
    }
    for r in rings:
        r.close()
    return out


def measure_sharder(ident, setup, pool, devices, n_calls=32, packets_per_shard=4096, streams_per_shard=256, threads=0,
                    samples="i16", device_entropy=False, seed=1, warm=8, copy_out=False):
    """End-to-end rate of the one-process multi-device path (lw_sharder_*): packets of len(devices) x streams_per_shard
    streams per call, stream_id mod G onto the shards, every shard on its own staging ring; up to three calls in flight.
    Returns a dict like measure()."""
    from .shard import Sharder
    G = len(devices)
    sh = Sharder(ident, setup, list(devices), packets_per_shard, samples)
    if device_entropy and not sh.set_entropy_on_device(True):
        raise RuntimeError("stream not eligible for the device entropy stage")
    rng = np.random.default_rng(seed)
    n_streams = G * streams_per_shard
    per = packets_per_shard // streams_per_shard
    calls = []
    for b in range(min(n_calls, 6)):
        order = rng.integers(0, len(pool), n_streams * per)
        # stream-major: the packets of a stream are consecutive (LDS hand-over inside the launch); the sharder sorts out owners
        calls.append(sh.marshal([(k // per, pool[int(i)]) for k, i in enumerate(order)]))
    out = None

    def take():   # the oldest call out: copied into one buffer, or left in the shards' pinned buffers (like Ring.collect_nocopy)
        nonlocal out
        if copy_out:
            out, _ = sh.collect(out, want_results=False)
        else:
            sh.collect_pinned(want_results=False)
            sh.release()

    def run(n):
        for k in range(n):
            if sh.in_flight == 3:
                take()
            sh.submit(calls[k % len(calls)], threads)
        while sh.in_flight:
            take()

    run(warm)
    t0 = time.perf_counter()
    run(n_calls)
    dt = time.perf_counter() - t0
    npk = n_calls * n_streams * per
    res = {"value": npk / dt, "unit": "packets/s", "packets": npk, "seconds": dt, "shards": G, "devices": list(devices),
           "per_shard": npk / dt / G, "packets_per_call": n_streams * per, "host_threads_per_shard": threads or
           max(1, _N.lw_default_host_threads() // G),
           "records": "raw packets, entropy stage on each shard's device (k_entropy)" if device_entropy else
                      "f32 residue vectors (host entropy stage)",
           "path": "lw_sharder_submit (per shard: lw_ring_stage + lw_ring_launch on the shard's thread and device) -> " +
                   ("lw_sharder_collect (pinned PCM -> caller's buffer)" if copy_out else
                    "lw_sharder_collect_pinned / lw_sharder_release (PCM left in the shards' pinned buffers)") + ", 3 calls in flight"}
    sh.close()
    return res
