"""GPU parity over the space of stream SETUPS (round 6; /root/reference/src/header.rs:771-918, 922-981, 985-1058, 1060-1154): a
fixed-seed slice of tools/fuzz_gpu_setups.py's campaign.  Every setup header is drawn at random (streamgen.random_setup:
channels, block sizes, modes / mappings / submaps, coupling lists, floor-1 post lists of 2..65 posts and class structures,
floor 0, residue shapes, codebooks); per setup random streams go through the product's default path, its generic kernels,
k_entropy where eligible, and the oracle -- statuses, sample counts, PCM and final window state identical."""
import os
import sys

import numpy as np
import pytest

from common import ROOT, po

sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_gpu_setups as fz  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first", [0, 10, 20, 30, 40])
def test_random_setups_three_ways(first):
    from lewton_amd import _native as N
    from lewton_amd import audio, header
    from lewton_amd.batch import Batch
    mods = (audio, header, Batch, po, N)
    rng = np.random.default_rng(first + 77)
    total, kernels = 0, set()
    for seed in range(first, first + 10):
        _seed, ch, idp, stp, seqs = fz.generate((seed, 120, None))
        checked, ks, _line, _dev = fz.run_setup(seed, ch, idp, stp, seqs, 120, rng, mods)
        total += checked
        kernels |= ks
    assert total > 600 and kernels


def test_random_setups_at_512_4096_and_256_4096():
    """the same with the block sizes pinned to 9 / 12 and 8 / 12 (streamgen's menu draws them for one setup in fourteen): random
    setups whose long blocks with short slopes run through k_long12's EDGE form"""
    from lewton_amd import _native as N
    from lewton_amd import audio, header
    from lewton_amd.batch import Batch
    mods = (audio, header, Batch, po, N)
    rng = np.random.default_rng(4242)
    total, kernels = 0, set()
    for seed in range(700, 712):
        _seed, ch, idp, stp, seqs = fz.generate((seed, 120, [(9, 12), (8, 12)]))
        checked, ks, _line, _dev = fz.run_setup(seed, ch, idp, stp, seqs, 120, rng, mods)
        total += checked
        kernels |= ks
    assert total > 600 and "k_long12" in kernels, kernels
