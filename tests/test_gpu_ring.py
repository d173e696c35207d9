"""The staging ring (lw_ring_*) against the oracle (-m gpu): batches in flight while the next one is entropy-decoded, the
same streams carried from batch to batch (state ordered across the slots' HIP streams), stage / launch split over two
threads, drain + state roll-back."""
import threading

import numpy as np
import pytest

from common import SETUPS, oracle_headers, po, sg

pytestmark = pytest.mark.gpu


def _product(setup):
    from lewton_amd import audio, header
    idp, _, stp = setup.headers()
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    return audio, ident, st


def _check(setup, o_id, o_st, opws, items, res, pcm, fmt="i16"):
    ch = setup.channels
    for (pkt, s), (status, m, off) in zip(items, res):
        try:
            want = po.read_audio_packet(o_id, o_st, pkt, opws[s], fmt)
            rc = 0
        except po.OracleError as e:
            rc = e.code
        assert status == rc
        if rc == 0:
            got = pcm[off:off + m * ch]
            assert got.size == want.size and np.array_equal(got, want.reshape(-1)), s


@pytest.mark.parametrize("name,pattern,slots", [("stereo", "L", 3), ("stereo", "LLSSSSLL", 2), ("surround51", "LLLS", 4)])
def test_ring_batches_in_flight_match_oracle(name, pattern, slots):
    from lewton_amd.ring import Ring
    setup = SETUPS[name]()
    audio, ident, st = _product(setup)
    o_id, o_st = oracle_headers(setup)
    dec = audio.decoder_for(ident, st)
    n_streams, per, n_batches = 24, 8, 7
    streams = [sg.make_stream(setup, pattern, per * n_batches, seed=700 + s, p_floor_unused=0.05) for s in range(n_streams)]
    pwrs = [audio.PreviousWindowRight() for _ in range(n_streams)]
    opws = [po.Pwr() for _ in range(n_streams)]
    ring = Ring(dec, slots, n_streams * per, "i16")
    batches = []
    for b in range(n_batches):     # stream-major inside a batch, the same streams in every batch
        batches.append([(streams[s][b * per + t], s) for s in range(n_streams) for t in range(per)])
    pending = []
    for b, items in enumerate(batches):
        if ring.in_flight == ring.slots:
            res, pcm = ring.collect()
            ring.release()
            _check(setup, o_id, o_st, opws, pending.pop(0), res, pcm)
        ring.submit(ring.marshal([(p, pwrs[s]) for p, s in items]), n_threads=3)
        pending.append(items)
    assert ring.in_flight == min(slots, n_batches)
    while pending:
        res, pcm = ring.collect()
        ring.release()
        _check(setup, o_id, o_st, opws, pending.pop(0), res, pcm)
    for s in range(0, n_streams, 5):
        assert np.array_equal(pwrs[s].data().view(np.uint32), opws[s].data(setup.channels).view(np.uint32))
    ring.close()


def test_ring_stage_and_launch_on_two_threads():
    """the Ogg reader's arrangement: one thread stages (host entropy decode), another launches / collects"""
    from lewton_amd.ring import Ring
    setup = SETUPS["stereo"]()
    audio, ident, st = _product(setup)
    o_id, o_st = oracle_headers(setup)
    dec = audio.decoder_for(ident, st)
    stream = sg.make_stream(setup, "LLLLLSSLLL", 600, seed=5)
    pwr, opw = audio.PreviousWindowRight(), po.Pwr()
    ring = Ring(dec, 3, 64, "i16")
    chunks = [stream[i:i + 50] for i in range(0, len(stream), 50)]
    staged = threading.Semaphore(0)
    free = threading.Semaphore(3)
    errors = []

    def producer():
        try:
            for c in chunks:
                free.acquire()
                ring.stage(ring.marshal([(p, pwr) for p in c]), n_threads=2)
                staged.release()
        except Exception as e:  # pragma: no cover
            errors.append(e)
            staged.release()

    t = threading.Thread(target=producer)
    t.start()
    got = []
    for c in chunks:
        staged.acquire()
        assert not errors, errors
        ring.launch()
        res, pcm = ring.collect()
        ring.release()
        free.release()
        got.append((c, res, pcm))
    t.join()
    for c, res, pcm in got:
        _check(setup, o_id, o_st, [opw], [(p, 0) for p in c], res, pcm)
    ring.close()


def test_ring_drain_and_state_rollback():
    """a staged-and-launched batch that is dropped: restoring the saved host state puts the stream back where it was (the
    launch wrote the other parity buffer), and decoding the same packets again gives the oracle's samples"""
    import ctypes as C
    from lewton_amd import _native as N
    from lewton_amd.ring import Ring
    setup = SETUPS["stereo"]()
    audio, ident, st = _product(setup)
    o_id, o_st = oracle_headers(setup)
    dec = audio.decoder_for(ident, st)
    stream = sg.make_stream(setup, "L", 40, seed=11)
    pwr, opw = audio.PreviousWindowRight(), po.Pwr()
    ring = Ring(dec, 2, 32, "i16")
    first = [(p, pwr) for p in stream[:10]]
    ring.submit(ring.marshal(first))
    res, pcm = ring.collect()
    ring.release()
    _check(setup, o_id, o_st, [opw], [(p, 0) for p in stream[:10]], res, pcm)
    saved = N.PwrState()
    N.lw_pwr_get_state(pwr._h, C.byref(saved))
    ring.submit(ring.marshal([(p, pwr) for p in stream[10:25]]))      # in flight ... and dropped
    ring.drain()
    assert ring.in_flight == 0
    N.lw_pwr_set_state(pwr._h, C.byref(saved))
    ring.submit(ring.marshal([(p, pwr) for p in stream[10:40]]))
    res, pcm = ring.collect()
    ring.release()
    _check(setup, o_id, o_st, [opw], [(p, 0) for p in stream[10:40]], res, pcm)
    ring.close()
