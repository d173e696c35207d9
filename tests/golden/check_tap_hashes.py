#!/usr/bin/env python3
"""Compares the TAPHASH lines integration/lewton_tap_hashes/tap_hashes.rs prints (lewton itself) with tests/golden/tap_hashes.json
(the oracle).  Reads stdin or the file given; exit code 0 = every hash and count equal = parity of the un-pinned stages
(residue, inverse coupling, floor x residue, IMDCT, overlap-add, i16 conversion) is pinned to the reference."""
import json
import os
import sys

want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tap_hashes.json")))["files"]
text = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
seen, bad = 0, 0
for line in text.splitlines():
    if "TAPHASH " not in line:
        continue
    f = line[line.index("TAPHASH "):].split()
    name, key = f[1], f[2]
    if key == "audio_packets":
        ok = int(f[3]) == want[name]["audio_packets"]
    else:
        ok = f[3] == want[name][key]["sha256"] and int(f[4]) == want[name][key]["values"]
    seen += 1
    bad += not ok
    print(("ok      " if ok else "MISMATCH"), name, key)
expected = sum(6 for _ in want)
if seen != expected:
    print("expected %d TAPHASH lines, got %d" % (expected, seen))
    sys.exit(2)
sys.exit(1 if bad else 0)
