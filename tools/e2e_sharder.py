#!/usr/bin/env python3
"""End-to-end rate of the one-process multi-device path (lw_sharder_* on staging rings; lewton_amd/e2e.py measure_sharder).
On a box with one GPU the shards are logical (the same device several times): what is measured then is that the sharder's
pipeline costs nothing against a single ring.      python tools/e2e_sharder.py [--shards 2] [--calls 40] [--copy-out]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lewton_amd import _native as N, e2e, header, streamgen as sg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shards", type=int, default=0, help="logical shards (default: one per visible GPU, 2 on a one-GPU box)")
ap.add_argument("--calls", type=int, default=40)
ap.add_argument("--packets-per-shard", type=int, default=4096)
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--host-entropy", action="store_true", help="host entropy stage instead of k_entropy")
ap.add_argument("--copy-out", action="store_true", help="lw_sharder_collect (copy into one buffer) instead of collect_pinned")
args = ap.parse_args()
setup = sg.stereo_setup(44100, 8, 11)
idp, _, stp = setup.headers()
ident = header.read_header_ident(idp)
st = header.read_header_setup(stp, 2, (8, 11))
pool = sg.make_stream(setup, "L", 512, seed=9)
ndev = max(1, N.lw_device_count())
G = args.shards or (ndev if ndev > 1 else 2)
r = e2e.measure_sharder(ident, st, pool, [g % ndev for g in range(G)], n_calls=args.calls, packets_per_shard=args.packets_per_shard,
                        streams_per_shard=256, threads=args.threads, device_entropy=not args.host_entropy, copy_out=args.copy_out)
print(json.dumps(r))
