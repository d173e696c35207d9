#!/usr/bin/env python3
"""(CPU only) the random-setup checks of tests/test_random_setups.py over many seeds, on several processes:
  host    the product's host entropy stage against the oracle's taps (statuses, floors, residue vectors, bit cursor)
  indep   tests/independent_decoder.py against the oracle at all four taps and on the samples (two restatements sharing no code)
    python tools/fuzz_cpu_setups.py --host 2000 --indep 600 [--seed 20000] [--procs 7]"""
import argparse
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def work(job):
    kind, seed = job
    import numpy as np
    import test_random_setups as t
    from lewton_amd import streamgen as sg
    rng = np.random.default_rng(seed)
    try:
        if kind == "host":
            setup = sg.random_setup(rng)
            return kind, seed, t.check_host_stage(setup, sg.random_stream(setup, rng, 10, seed=seed, p_damage=0.1)), None
        setup = sg.random_setup(rng, allow_floor0=False)
        idp, _cmt, stp = setup.headers()
        ok, _s = t._compare(idp, stp, sg.random_stream(setup, rng, 8, seed=seed, p_damage=0.1), expect_errors=True)
        return kind, seed, ok, None
    except Exception as e:   # noqa: BLE001 (report the seed)
        return kind, seed, 0, repr(e)[:1500]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--host", type=int, default=1000)
    ap.add_argument("--indep", type=int, default=300)
    ap.add_argument("--seed", type=int, default=20000)
    ap.add_argument("--procs", type=int, default=7)
    a = ap.parse_args()
    jobs = [("host", s) for s in range(a.seed, a.seed + a.host)] + [("indep", s) for s in range(a.seed + 10 ** 6, a.seed + 10 ** 6 + a.indep)]
    n = {"host": 0, "indep": 0}
    bad = 0
    with mp.get_context("fork").Pool(a.procs) as pool:
        for kind, seed, ok, err in pool.imap_unordered(work, jobs, chunksize=4):
            n[kind] += ok
            if err:
                bad += 1
                print("FAILED %s seed %d: %s" % (kind, seed, err), flush=True)
    print("fuzz_cpu_setups: %d random setups through the host entropy stage (%d packets decoded, all equal to the oracle's taps), %d through "
          "the independent decoder (%d packets equal to the oracle at all four taps): %s" % (
              a.host, n["host"], a.indep, n["indep"], "no difference" if not bad else "%d FAILED" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
