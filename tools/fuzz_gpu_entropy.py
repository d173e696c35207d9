#!/usr/bin/env python3
"""(GPU box) Differential campaign of the device entropy stage against the host entropy stage at scale: the same batches of
intact, truncated and bit-flipped packets go through the staging ring twice -- host stage, then k_entropy -- and every
packet's status and every PCM sample must be identical.  (Both sides are product code; the oracle comparison of either
path is tests/test_gpu_*.py.)
    python tools/fuzz_gpu_entropy.py [--packets 200000] [--seed 1] [--setup stereo|stereo_t1|mono_small|surround51|surround51_bookless|multichannel12|spill_t1|spill_t2|real]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lewton_amd import audio, header, streamgen as sg  # noqa: E402
from lewton_amd.ring import Ring  # noqa: E402
from common import bookless_submap_setup, spill_setup  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--packets", type=int, default=200000)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--setup", default="stereo")
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
if args.setup == "real":
    from oracle import pyogg
    rd = pyogg.PacketReader(open(os.path.join(ROOT, "tests", "golden", "invalid_keypress.ogg"), "rb").read())
    pk = []
    while True:
        p = rd.read_packet()
        if p is None:
            break
        pk.append(bytes(p.data))
    idp, stp, base = pk[0], pk[2], pk[3:]
    pattern_len = len(base)
else:
    setups = {"stereo": sg.stereo_setup, "stereo_t1": lambda: sg.stereo_setup(residue_type=1), "mono_small": sg.mono_setup,
              "surround51": sg.surround51_setup, "spill_t1": lambda: spill_setup(1), "spill_t2": lambda: spill_setup(2), "surround51_bookless": bookless_submap_setup, "multichannel12": lambda: sg.multichannel_setup(12)}
    setup = setups[args.setup]()
    idp, _, stp = setup.headers()
    base = sg.make_stream(setup, "LLSLSSLL", 400, seed=args.seed, p_floor_unused=0.1)
    pattern_len = len(base)
ident = header.read_header_ident(idp)
st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
dec = audio.decoder_for(ident, st, 0)


def damage(p):
    p = bytearray(p)
    r = rng.random()
    if r < 0.25 and len(p) > 1:
        p = p[: int(rng.integers(0, len(p)))]
    elif r < 0.55 and len(p):
        for _ in range(int(rng.integers(1, 5))):
            p[int(rng.integers(0, len(p)))] ^= 1 << int(rng.integers(0, 8))
    return bytes(p)


per_batch, n_streams = 4096, 128
per = per_batch // n_streams
done = bad = ok_packets = 0
rings = [Ring(dec, 2, per_batch, "i16") for _ in range(2)]
assert rings[1].set_entropy_on_device(True), "stream not eligible"
pwrs = [[audio.PreviousWindowRight() for _ in range(n_streams)] for _ in range(2)]
while done < args.packets:
    start = int(rng.integers(0, pattern_len))
    items = [damage(base[(start + s * 7 + t) % pattern_len]) for s in range(n_streams) for t in range(per)]
    out = []
    for k, ring in enumerate(rings):
        ring.submit(ring.marshal([(p, pwrs[k][i // per]) for i, p in enumerate(items)]), n_threads=4)
        res, pcm = ring.collect()
        out.append((list(res), pcm.copy()))
        ring.release()
    if out[0][0] != out[1][0] or not np.array_equal(out[0][1], out[1][1]):
        bad += 1
        print("DIFFERENCE in batch starting at packet", done)
    ok_packets += sum(1 for r in out[0][0] if r[0] == 0)
    done += per_batch
print("%s: %d packets (%d decoded, the rest rejected alike) through host stage and k_entropy: %s" % (
    args.setup, done, ok_packets, "IDENTICAL" if bad == 0 else "%d batches differ" % bad))
sys.exit(1 if bad else 0)
