#!/usr/bin/env python3
"""End-to-end throughput through the library's staging ring (see lewton_amd/e2e.py).
    python tools/e2e.py [--batches 48] [--threads 0] [--slots 3] [--callers 2]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lewton_amd import audio, e2e, header, streamgen as sg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, default=48)
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--packets", type=int, default=4096)
ap.add_argument("--streams", type=int, default=256)
ap.add_argument("--slots", type=int, default=3)
ap.add_argument("--callers", type=int, default=1, help="host threads, each with its own ring and its own streams")
ap.add_argument("--device-entropy", action="store_true", help="ship the packets, entropy stage in k_entropy")
ap.add_argument("--shared-device", action="store_true",
                help="lw_decoder_set_shared_device: several rings on this GPU (--callers > 1, or other processes): copies by the copier thread")
ap.add_argument("--start-at", type=float, default=0.0, help="unix time to start the timed part at (two processes side by side)")
args = ap.parse_args()

setup = sg.stereo_setup(44100, 8, 11)
idp, _, stp = setup.headers()
ident = header.read_header_ident(idp)
st = header.read_header_setup(stp, 2, (8, 11))
dec = audio.decoder_for(ident, st, 0)
pool = sg.make_stream(setup, "L", 512, seed=9)
if args.shared_device:
    dec.set_shared_device(True)
if args.start_at:
    import time
    e2e.measure(dec, pool, 16, args.packets, args.streams, args.threads, args.slots, args.callers, device_entropy=args.device_entropy)
    time.sleep(max(0.0, args.start_at - time.time()))
r = e2e.measure(dec, pool, args.batches, args.packets, args.streams, args.threads, args.slots, args.callers,
                device_entropy=args.device_entropy)
print(json.dumps(r))
print("end-to-end: %d packets in %.3f s -> %.2f M packets/s (%s; H2D %s GB/s, D2H %.2f GB/s); host entropy stage alone "
      "%.2f M packets/s on %s threads, %d caller(s), %d slots" % (
          r["packets"], r["seconds"], r["value"] / 1e6, r["records"],
          ("%.2f" % r["h2d_GBps"]) if r["h2d_GBps"] else "n/a", r["d2h_GBps"], r["host_entropy_stage_alone"] / 1e6,
          r["host_threads"], r["callers"], r["ring_slots"]))
