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


@pytest.mark.parametrize("first", [900, 910])
def test_random_setups_through_the_ogg_reader_with_read_ahead(first):
    """random setups wrapped in Ogg pages: OggStreamReader.read_dec_packet served from batches decoded ahead (both entropy tiers where
    the stream is eligible) against the oracle's reader, call for call -- samples or the BadAudio code, granule position"""
    from lewton_amd import inside_ogg as IO
    from lewton_amd import ogg
    from lewton_amd import streamgen as sg
    from oracle import pyogg
    total = 0
    for seed in range(first, first + 10):
        rng = np.random.default_rng(seed)
        setup = sg.random_setup(rng)
        pk = sg.random_stream(setup, rng, 60, seed=seed, p_floor_unused=0.05, p_damage=0.05)
        idp, cmt, stp = setup.headers()
        o_id = po.Ident(idp)
        o_st = po.Setup(stp, o_id)
        w = ogg.PageWriter(seed)
        w.add_packet(idp, 0, flush=True)
        w.add_packet(cmt, 0)
        w.add_packet(stp, 0, flush=True)
        gp = 0
        for i, p in enumerate(pk):
            try:
                gp += po.get_decoded_sample_count(o_id, o_st, p) if i else 0
            except po.OracleError:
                pass
            w.add_packet(p, gp, flush=(i % 7 == 6), eos=(i == len(pk) - 1))
        data = w.bytes()
        for k, dev in ((5, False), (64, True)):
            s, o = IO.OggStreamReader(data), pyogg.OggStreamReader(data)
            s.set_read_ahead(k, 2)
            s.set_entropy_on_device(dev)
            while True:
                ea = eb = a = b = None
                try:
                    a = s.read_dec_packet()
                except IO.VorbisError as e:
                    ea = e
                try:
                    b = o.read_dec_packet()
                except pyogg.VorbisError as e:
                    eb = e
                assert (ea is None) == (eb is None), (seed, ea, eb)
                if ea is not None:
                    assert ea.kind == eb.kind == "BadAudio" and ea.code == eb.inner, (seed, ea, eb)
                else:
                    assert (a is None) == (b is None), seed
                    if a is None:
                        break
                    assert a.shape == b.shape and np.array_equal(a, b), seed
                    total += 1
                assert s.get_last_absgp() == o.get_last_absgp(), seed
    assert total > 500
