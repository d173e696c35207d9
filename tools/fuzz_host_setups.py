#!/usr/bin/env python3
"""(CPU only) random setup headers through the product's host side under ASan / UBSan, many seeds: the loop of
tests/test_host_batch.py::test_random_setups_plan_and_statuses_under_asan (header parser, class planners, batch planner, slot and
item lists with the kernels as no-ops; statuses / sample counts / output offsets of damaged multi-stream batches against the
oracle) over `--setups` seeds on `--procs` processes.
    python tools/fuzz_host_setups.py --setups 600 [--seed 50000] [--blocksizes 9:12,8:12,8:10,9:10,8:11]"""
import argparse
import multiprocessing as mp
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def work(job):
    import test_host_batch as thb
    exe, seed, sizes, tmp = job
    bs = sizes[seed % len(sizes)] if sizes else None
    try:
        return seed, thb.random_setup_case(exe, seed, os.path.join(tmp, "case_%d.bin" % os.getpid()), blocksizes=bs), None
    except AssertionError as e:
        return seed, 0, repr(e)[:3000]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--setups", type=int, default=300)
    ap.add_argument("--seed", type=int, default=50000)
    ap.add_argument("--procs", type=int, default=6)
    ap.add_argument("--blocksizes", default="", help="pin the block sizes, e.g. 9:12,8:12 (setup `seed` takes entry seed %% count)")
    args = ap.parse_args()
    import test_host_batch as thb
    sizes = [tuple(int(v) for v in e.split(":")) for e in args.blocksizes.split(",") if e]
    with tempfile.TemporaryDirectory() as tmp:
        exe = thb.build_harness_asan(os.path.join(tmp, "batch_host_bench_asan"))
        jobs = [(exe, s, sizes, tmp) for s in range(args.seed, args.seed + args.setups)]
        bad = rejected = 0
        with mp.get_context("fork").Pool(args.procs) as pool:
            for seed, n_bad, err in pool.imap_unordered(work, jobs, chunksize=2):
                rejected += n_bad
                if err:
                    bad += 1
                    print("FAILED setup %d: %s" % (seed, err), flush=True)
    print("fuzz_host_setups: %d random setups (seeds %d..%d%s) through the host side under ASan / UBSan, %d packets each, %d rejected "
          "alike by the oracle: %s" % (args.setups, args.seed, args.seed + args.setups - 1,
                                       ", block sizes " + args.blocksizes if sizes else "", 6 * 14, rejected,
                                       "statuses, sample counts and offsets IDENTICAL, no sanitizer report" if not bad else "%d FAILED" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
