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
            samples="i16", warm=4, device_entropy=False):
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
