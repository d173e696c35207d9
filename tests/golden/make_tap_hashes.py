#!/usr/bin/env python3
"""Whole-packet parity fixture for whoever has cargo (the image has neither rustc nor a network):

* writes two small synthetic Ogg/Vorbis files next to the real-file fixture (deterministic: lewton_amd.streamgen, fixed seeds);
* decodes the three files with the ORACLE (oracle/lewton_oracle.c through oracle/pyogg.py) and records, per file, the
  SHA-256 of the interleaved i16 PCM (`OggStreamReader::read_dec_packet_itl` semantics incl. the final-granule truncation,
  inside_ogg.rs:183-231) and of the four debug taps of lewton (`record_*!`, /root/reference/src/lib.rs:56-94; call sites
  audio.rs:988, :1004, :1041, :1054) as raw f32 bit patterns, every audio packet in decode order, channel by channel;
* writes tests/golden/tap_hashes.json.

The Rust side (integration/lewton_tap_hashes/: a patch that turns the commented-out macro bodies into recorders, and an
integration test that prints the same hashes) is committed as source; `python tests/golden/check_tap_hashes.py <its output>`
compares.  One `cargo test` then pins the stages the reference holds no vectors for: residue decode / accumulate,
inverse coupling, floor curve x residue, IMDCT, window / overlap-add, i16 conversion.

    python tests/golden/make_tap_hashes.py          # regenerate files + hashes (needs the built oracle, no GPU)"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from lewton_amd import streamgen as sg  # noqa: E402
from oracle import pyogg, pyoracle as po  # noqa: E402

TAPS = ("residue_pre_inverse", "residue_post_inverse", "pre_mdct", "post_mdct")


def synth_file(setup, pattern, count, seed, serial):
    pk = sg.make_stream(setup, pattern, count, seed=seed, p_floor_unused=0.05)
    o_id = po.Ident(setup.headers()[0])
    o_st = po.Setup(setup.headers()[2], o_id)
    pwr = po.Pwr()
    counts = [po.read_audio_packet(o_id, o_st, p, pwr, "i16").shape[1] for p in pk]   # granule positions of the pages
    return sg.ogg_stream(setup, pk, counts, serial=serial, packets_per_page=5, final_trim=37)


def hashes_of(data):
    # PCM as OggStreamReader::read_dec_packet_itl delivers it
    rd = pyogg.OggStreamReader(data, fmt="i16_itl")
    pcm = hashlib.sha256()
    n_samples = 0
    n_packets = 0
    while True:
        dec = rd.read_dec_packet()
        if dec is None:
            break
        pcm.update(np.asarray(dec, "<i2").tobytes())
        n_samples += len(dec)
        n_packets += 1
    # taps: every audio packet of the (single) logical stream in decode order
    pr = pyogg.PacketReader(data)
    (ident, _cmt, setup), _serial = pyogg.read_headers(pr)
    pwr = po.Pwr()
    th = {k: hashlib.sha256() for k in TAPS}
    tn = {k: 0 for k in TAPS}
    while True:
        pck = pr.read_packet()
        if pck is None:
            break
        _out, taps = po.read_audio_packet(ident, setup, pck.data, pwr, "f32", taps=True)
        for k in TAPS:
            a = np.ascontiguousarray(taps[k], "<f4")
            th[k].update(a.tobytes())
            tn[k] += a.size
    out = {"audio_packets": n_packets, "pcm_i16_interleaved": {"sha256": pcm.hexdigest(), "values": n_samples}}
    for k in TAPS:
        out[k] = {"sha256": th[k].hexdigest(), "values": tn[k]}
    return out


def main():
    files = {
        "invalid_keypress.ogg": None,   # real encoder output (MathJax, Apache-2.0; see NOTICE.md)
        "synth_stereo_mixed.ogg": lambda: synth_file(sg.stereo_setup(44100, 8, 11), "LLSSSSSSSSLLSLLLSSL", 60, 2026, 0x4C57),
        "synth_surround51.ogg": lambda: synth_file(sg.surround51_setup(48000, 8, 11), "LLSSLLLSL", 24, 2027, 0x4C58),
    }
    res = {}
    for name, make in files.items():
        path = os.path.join(HERE, name)
        if make is not None:
            blob = make()
            if not os.path.exists(path) or open(path, "rb").read() != blob:
                open(path, "wb").write(blob)
        data = open(path, "rb").read()
        res[name] = dict(hashes_of(data), bytes=len(data), file_sha256=hashlib.sha256(data).hexdigest())
    doc = {"_about": "SHA-256 of the oracle's output for the files beside this one: interleaved i16 PCM "
                     "(read_dec_packet_itl) and lewton's four record_*! taps as f32 bit patterns (little endian), all audio "
                     "packets in decode order, channel by channel.  tests/golden/make_tap_hashes.py writes it, "
                     "integration/lewton_tap_hashes prints the same from lewton itself.",
           "files": res}
    json.dump(doc, open(os.path.join(HERE, "tap_hashes.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(doc, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
