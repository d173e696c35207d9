#!/usr/bin/env python3
"""Writes a long synthetic Ogg/Vorbis file (44.1 kHz stereo, mostly long blocks) for examples/perf:
    python tools/make_long_ogg.py out.ogg [packets]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lewton_amd import header, audio, ogg, streamgen as sg  # noqa: E402

out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
setup = sg.stereo_setup(44100, 8, 11)
idp, cmt, stp = setup.headers()
ident = header.read_header_ident(idp)
st = header.read_header_setup(stp, 2, (8, 11))
pool = sg.make_stream(setup, "LLLLLLLLLLLLLLLLLLLSSSSL", 480, seed=11)   # a repeating, self-consistent block pattern
w = ogg.PageWriter(0x4C57)
w.add_packet(idp, 0, flush=True)
w.add_packet(cmt, 0)
w.add_packet(stp, 0, flush=True)
gp = 0
for i in range(n):
    p = pool[i % len(pool)]
    if i:
        gp += audio.get_decoded_sample_count(ident, st, p)
    w.add_packet(p, gp, flush=(i % 24 == 23), eos=(i == n - 1))
open(out, "wb").write(w.bytes())
print("%s: %d packets, %.1f s of audio, %d bytes" % (out, n, gp / 44100.0, len(w.bytes())))
