"""Two logical shards on one GPU through the sharder, three calls in flight, entropy stage on the device, every packet against
the oracle (tests/test_gpu_shapes.py::test_two_tenants_on_one_gpu).  As a script: `python tests/tenants_worker.py <share_cus>`
prints TENANTS_OK -- the variant with CU-masked streams runs in a process of its own (see the test)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from common import SETUPS, oracle_headers, po, sg  # noqa: E402


def run(share):
    from lewton_amd import _native as N
    from lewton_amd import audio, header, shard
    setup = SETUPS["stereo"]()
    idp, _, stp = setup.headers()
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    o_id, o_st = oracle_headers(setup)
    dec = audio.Decoder(ident, st, 0)
    total = N.lw_decoder_cu_count(dec._h)
    if share:   # the share of a lone decoder: CUs [32 j / k, 32 (j + 1) / k) of each of the eight XCDs
        assert total == 256 and [dec.set_cu_share(j, 3) for j in range(3)] == [88, 88, 80]
        assert dec.set_cu_share(0, 1) == total and dec.set_cu_share(31, 32) == 8
        assert N.lw_decoder_set_cu_share(dec._h, 0, 33) == N.ERR_UNSUPPORTED and N.lw_decoder_set_cu_share(dec._h, 2, 2) != 0
    dec.close()
    S, per, n_calls = 512, 16, 5
    streams = [sg.make_stream(setup, "L", n_calls * per, seed=900 + s) for s in range(16)]
    sh = shard.Sharder(ident, st, [0, 0], max_packets_per_shard=S // 2 * per, samples="i16", share_cus=share)
    assert [sh.shard_cus(0), sh.shard_cus(1)] == ([total // 2] * 2 if share else [total] * 2)
    assert sh.set_entropy_on_device(True)
    opws = [po.Pwr() for _ in range(S)]
    calls = [[(s, streams[s % 16][c * per + t]) for s in range(S) for t in range(per)] for c in range(n_calls)]
    done = 0

    def take():
        nonlocal done
        views, res = sh.collect_pinned()
        for (s, pkt), (status, m, off) in zip(calls[done], res):
            want = po.read_audio_packet(o_id, o_st, pkt, opws[s], "i16")
            assert status == 0 and m == want.shape[1], (share, done, s)
            assert np.array_equal(views[sh.shard_of(s)][off:off + 2 * m], want.reshape(-1)), (share, done, s)
        sh.release()
        done += 1

    for c in range(n_calls):
        if sh.in_flight == 3:
            take()
        sh.submit(sh.marshal(calls[c]), 8)
    while sh.in_flight:
        take()
    sh.close()
    return S * per * n_calls


if __name__ == "__main__":
    n = run(bool(int(sys.argv[1])))
    print("TENANTS_OK %d packets" % n, flush=True)
