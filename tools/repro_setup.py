#!/usr/bin/env python3
"""(GPU box) one random setup of tools/fuzz_gpu_setups.py under many batch cuts, with a detailed report of the first difference:
    python tools/repro_setup.py SEED [TRIALS]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_gpu_setups as fz  # noqa: E402


def main():
    seed = int(sys.argv[1])
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    from lewton_amd import _native as N
    from lewton_amd import audio, header
    from lewton_amd.batch import Batch
    from oracle import pyoracle as po
    _s, ch, idp, stp, seqs = fz.generate((seed, 250, None))
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    o_id = po.Ident(idp)
    o_st = po.Setup(stp, o_id)
    n_streams, length = fz.shape_of(seed, 250)
    fmt = fz.FMTS[seed % 3]
    print("setup", seed, "ch", ch, "fmt", fmt, "streams", n_streams, "x", length, fz.census_line(N, ident, st))
    want = []
    for q in range(fz.DISTINCT):
        opw = po.Pwr()
        rows = []
        for p in seqs[q]:
            try:
                rows.append((0, np.asarray(po.read_audio_packet(o_id, o_st, p, opw, fz.OFMT[fmt]))))
            except po.OracleError as e:
                rows.append((e.code, None))
        want.append(rows)
    dec = audio.Decoder(ident, st, 0)
    n_bad = 0
    for trial in range(trials):
        rng = np.random.default_rng(1000 + trial)
        if rng.random() < 0.5:
            chunk = int(rng.choice([1, 4, 16, length]))
            order = []
            for c0 in range(0, length, chunk):
                for s in range(n_streams):
                    order += [(s, t) for t in range(c0, min(length, c0 + chunk))]
        else:
            order = [(s, t) for t in range(length) for s in range(n_streams)]
        cuts = sorted(set([0, len(order)] + [int(x) for x in rng.integers(1, max(2, len(order)), int(rng.integers(0, 5)))]))
        cap = max(b - a for a, b in zip(cuts[:-1], cuts[1:]))
        bt = Batch(dec, cap, fmt)
        pws = [audio.PreviousWindowRight() for _ in range(n_streams)]
        bad = None
        for a, b in zip(cuts[:-1], cuts[1:]):
            items = order[a:b]
            res = bt.entropy([(seqs[s % fz.DISTINCT][t], pws[s]) for s, t in items], n_threads=2)
            bt.upload()
            pcm = bt.split(bt.synth_to_host(), ch)
            for i, (s, t) in enumerate(items):
                rc, w = want[s % fz.DISTINCT][t]
                if rc or res[i][0]:
                    continue
                g = pcm[i].reshape(-1)
                w = w.reshape(-1)
                if g.size != w.size or not (fz.f32_identical(g, w) if fmt == "f32" else np.array_equal(g, w)):
                    idx = np.flatnonzero(g != w) if g.size == w.size else np.array([], int)
                    m = w.size // ch
                    if fmt == "i16_interleaved":
                        chans, samp = idx % ch, idx // ch
                    else:
                        chans, samp = idx // max(1, m), idx % max(1, m)
                    pk = seqs[s % fz.DISTINCT][t]
                    bad = (trial, cuts, (a, b), i, s, t, m, len(idx), sorted(set(chans.tolist())), int(samp.min()) if len(idx) else -1,
                           int(samp.max()) if len(idx) else -1, bt.last_kernels, pk[:2].hex(),
                           [(int(g[j]), int(w[j])) for j in idx[:6]])
                    break
            if bad:
                break
        bt.close()
        if bad:
            n_bad += 1
            print("MISMATCH trial %d cuts %s batch %s item %d stream %d packet %d: m=%d, %d elements differ, channels %s, samples %d..%d, kernels %s, "
                  "packet head %s, (got, want) %s" % bad)
    print("done: %d of %d trials differ" % (n_bad, trials))


if __name__ == "__main__":
    main()
