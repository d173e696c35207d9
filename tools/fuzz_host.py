#!/usr/bin/env python3
"""Longer differential / sanitizer fuzz campaigns of the HOST code than the CPU suite can afford (no GPU needed):

  entropy   mutated audio packets of every test setup through lw_entropy_decode_host vs the oracle: error code, residue
            vectors bit for bit, bit cursor
  stream    mutated Ogg files (page CRCs repaired so that the damage reaches the decoder) through the stream layer built
            over the HIP stand-ins with ASan + UBSan (tests/san/ogg_stream_host.cpp): packet-by-packet trace vs the oracle's
            OggStreamReader, and look-ahead / skip / seek runs that must not trip a sanitizer

    python tools/fuzz_host.py [--seed 1] [--entropy-packets 40] [--stream-cases 200]
The CPU suite runs the same checks at a fixed small scale (tests/test_fuzz_host.py, tests/test_host_ogg.py)."""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_fuzz_host import ALL, _mutate, oracle_headers, po, sg  # noqa: E402
from test_host_ogg import SRC, HIP_INC, _files, _oracle_trace  # noqa: E402
from lewton_amd import audio, header, ogg  # noqa: E402
from oracle import pyogg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--entropy-packets", type=int, default=40, help="packets per setup (7 mutations each)")
ap.add_argument("--stream-cases", type=int, default=200)
args = ap.parse_args()
rng = np.random.default_rng(args.seed)

# ---- entropy stage ---------------------------------------------------------------------------------------------
tot = ok = 0
for name in sorted(ALL):
    setup = ALL[name]()
    idp, _c, stp = setup.headers()
    o_id, o_st = oracle_headers(setup)
    hid = header.read_header_ident(idp)
    hst = header.read_header_setup(stp, hid.audio_channels, (hid.blocksize_0, hid.blocksize_1))
    for p in sg.make_stream(setup, "LSLLS", args.entropy_packets, seed=args.seed, p_floor_unused=0.1):
        for kind in (0, 1, 1, 2, 3, 0, 2):
            m = _mutate(rng, p, kind)
            try:
                _o, taps = po.read_audio_packet(o_id, o_st, m, po.Pwr(), "f32", taps=True)
                want, bits_o = 0, po.lib().lwo_debug_bits_consumed()
            except po.OracleError as e:
                want = e.code
            try:
                got = audio.entropy_decode_host(hid, hst, m)
                rc = 0
            except audio.AudioReadError as e:
                rc = e.code
            tot += 1
            assert rc == want, (name, kind, rc, want, m.hex())
            if rc == 0:
                ok += 1
                assert np.array_equal(got["residue"].view(np.uint32), taps["residue_pre_inverse"].view(np.uint32)), (name, m.hex())
                assert got["bits"] == bits_o, (name, got["bits"], bits_o, m.hex())
print("entropy stage: %d mutated packets, %d decodable, all equal to the oracle" % (tot, ok))


# ---- stream layer ----------------------------------------------------------------------------------------------
def fix_crcs(d):
    d, i = bytearray(d), 0
    while True:
        i = bytes(d).find(b"OggS", i)
        if i < 0 or i + 27 > len(d):
            break
        nseg = d[i + 26]
        if i + 27 + nseg > len(d):
            break
        size = 27 + nseg + sum(d[i + 27:i + 27 + nseg])
        if i + size > len(d):
            break
        d[i + 22:i + 26] = b"\0\0\0\0"
        d[i + 22:i + 26] = ogg.crc32(bytes(d[i:i + size])).to_bytes(4, "little")
        i += size
    return bytes(d)


with tempfile.TemporaryDirectory() as tmp:
    exe, path = os.path.join(tmp, "ogg_stream_host"), os.path.join(tmp, "in.ogg")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I" + HIP_INC] + SRC + ["-lpthread", "-o", exe])
    files = _files()
    names = sorted(files)
    agree = opened = comment_rejects = 0
    for t in range(args.stream_cases):
        d = bytearray(files[names[t % len(names)]])
        kind = int(rng.integers(0, 5))
        for _ in range(int(rng.integers(1, 5))):
            a = int(rng.integers(len(d) // 3 if t % 2 else 0, len(d)))
            if kind == 0:
                d[a] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                d[a] = int(rng.integers(0, 256))
            elif kind == 2:
                d[a:a + int(rng.integers(1, 8))] = b""
            elif kind == 3:
                d[a:a] = bytes(rng.integers(0, 256, int(rng.integers(1, 8)), dtype=np.uint8))
            else:
                d = d[:max(28, a)]
        data = fix_crcs(d) if rng.random() < 0.8 else bytes(d)
        with open(path, "wb") as f:
            f.write(data)
        for mode in (["seq"], ["ahead", "7"], ["skip", str(int(rng.integers(0, 30000)))], ["seek", str(int(rng.integers(0, 40000)))]):
            r = subprocess.run([exe, path] + mode, capture_output=True, text=True, timeout=300)
            if r.returncode == 0 and mode[0] in ("seq", "skip", "seek"):
                # the same calls with the packet-by-packet call served from batches decoded ahead (lw_ogg_stream_set_read_ahead):
                # the same trace, whatever the damage
                ra = subprocess.run([exe, path] + mode, capture_output=True, text=True, timeout=300,
                                    env=dict(os.environ, LW_OSH_READ_AHEAD=str(int(rng.choice([1, 3, 40])))))
                if ra.returncode == 0 and ra.stdout != r.stdout:
                    keep = os.path.join(ROOT, "gpurun_out", "fuzz_readahead_%d_%s.ogg" % (t, mode[0]))
                    os.makedirs(os.path.dirname(keep), exist_ok=True)
                    open(keep, "wb").write(data)
                    raise SystemExit("read-ahead trace differs in mode %s, input kept as %s" % (mode, keep))
                if ra.returncode != 0:
                    r = ra
            if r.returncode != 0:
                keep = os.path.join(ROOT, "gpurun_out", "fuzz_crash_%d_%s.ogg" % (t, mode[0]))
                os.makedirs(os.path.dirname(keep), exist_ok=True)
                open(keep, "wb").write(data)
                raise SystemExit("sanitizer / crash in mode %s, input kept as %s\n%s" % (mode, keep, r.stderr[-3000:]))
            if mode == ["seq"]:
                got = [l.split() for l in r.stdout.splitlines()]
                try:
                    try:
                        want = _oracle_trace(pyogg.OggStreamReader(data))
                    except pyogg.VorbisError:
                        assert got and got[0][0] == "E", (t, got[:2])
                        continue
                    if got and got[0][0] == "E":
                        # the oracle does not parse comment headers (oracle/pyogg.py read_headers); the product and the
                        # reference do (inside_ogg.rs:38-41): a damaged comment packet is a header error there
                        try:
                            header.read_header_comment(pyogg.OggStreamReader(data).comment_packet)
                        except header.HeaderReadError as e:
                            assert int(got[0][1]) == e.code, (t, got[0], e.code)
                            comment_rejects += 1
                            continue
                    opened += 1
                    if want[-1] == ["EOF"]:
                        assert got == want, (t, want[-2:], got[-2:])
                        agree += 1
                    else:
                        e, k = want[-1][1], len(want) - 1
                        assert got[:k] == want[:k] and got[k][0] == "E", (t, want[-2:], got[k - 1:k + 1])
                        assert e.kind != "BadAudio" or got[k][1] == str(e.inner), (t, e, got[k])
                except AssertionError:
                    keep = os.path.join(ROOT, "gpurun_out", "fuzz_diff_%d.ogg" % t)
                    os.makedirs(os.path.dirname(keep), exist_ok=True)
                    open(keep, "wb").write(data)
                    print("trace differs from the oracle's; input kept as", keep)
                    raise
    print("stream layer: %d mutated files x 4 modes (three of them also with the read-ahead on: identical traces) without a sanitizer report; %d opened, %d read to a clean end, all traces equal "
          "to the oracle (%d damaged comment headers rejected by the product's parser only: the oracle has none)" % (
              args.stream_cases, opened, agree, comment_rejects))
