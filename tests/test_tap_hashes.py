"""tests/golden/tap_hashes.json (the whole-packet fixture handed to the first person with cargo):
the committed hashes are what the oracle produces today (CPU), and the HIP path produces the same PCM (GPU)."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

from common import ROOT

GOLD = os.path.join(ROOT, "tests", "golden")
WANT = json.load(open(os.path.join(GOLD, "tap_hashes.json")))["files"]


def test_committed_hashes_are_the_oracles():
    sys.path.insert(0, GOLD)
    import make_tap_hashes as mk
    for name, w in WANT.items():
        data = open(os.path.join(GOLD, name), "rb").read()
        assert hashlib.sha256(data).hexdigest() == w["file_sha256"], name
        got = mk.hashes_of(data)
        for k in ("audio_packets", "pcm_i16_interleaved") + mk.TAPS:
            assert got[k] == w[k], (name, k)


def test_rust_side_is_committed_and_checker_accepts_the_oracle():
    """the checker script against lines in the format integration/lewton_tap_hashes/tap_hashes.rs prints"""
    import subprocess
    d = os.path.join(ROOT, "integration", "lewton_tap_hashes")
    assert os.path.getsize(os.path.join(d, "lewton_taps.patch")) > 500 and os.path.getsize(os.path.join(d, "tap_hashes.rs")) > 2000
    lines = []
    for name, w in WANT.items():
        lines.append("TAPHASH %s audio_packets %d" % (name, w["audio_packets"]))
        for k in ("pcm_i16_interleaved", "residue_pre_inverse", "residue_post_inverse", "pre_mdct", "post_mdct"):
            lines.append("test output noise TAPHASH %s %s %s %d" % (name, k, w[k]["sha256"], w[k]["values"]))
    chk = os.path.join(GOLD, "check_tap_hashes.py")
    ok = subprocess.run([sys.executable, chk], input="\n".join(lines), text=True, capture_output=True)
    assert ok.returncode == 0, ok.stdout
    bad = subprocess.run([sys.executable, chk], input="\n".join(lines).replace("a", "b", 1), text=True, capture_output=True)
    assert bad.returncode != 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(WANT))
def test_hip_path_pcm_hash(name):
    from lewton_amd import inside_ogg as IO
    data = open(os.path.join(GOLD, name), "rb").read()
    s = IO.OggStreamReader(data)
    h, n, k = hashlib.sha256(), 0, 0
    while True:
        p = s.read_dec_packet_itl()
        if p is None:
            break
        h.update(np.asarray(p, "<i2").tobytes())
        n += len(p)
        k += 1
    assert k == WANT[name]["audio_packets"] and n == WANT[name]["pcm_i16_interleaved"]["values"]
    assert h.hexdigest() == WANT[name]["pcm_i16_interleaved"]["sha256"]
