"""CPU baseline worker (test infrastructure, part of bench.py's cpu_baseline leg): decodes a synthetic 44.1 kHz stereo
long-block stream with the C oracle for ~T seconds on ONE thread and prints `packets seconds`.  bench.py starts one
worker per host core to report the all-cores figure next to the single-thread one ("one stream per core").
    python -m oracle.cpu_bench <seed> <seconds>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    seed, seconds = int(sys.argv[1]), float(sys.argv[2])
    from lewton_amd import streamgen as sg   # the stream GENERATOR only (no decoding logic, no GPU library)
    from oracle import pyoracle as po
    setup = sg.stereo_setup(44100, 8, 11)
    idp, _, stp = setup.headers()
    pool = sg.make_stream(setup, "L", 256, seed=seed)
    o_id = po.Ident(idp)
    o_st = po.Setup(stp, o_id)
    _, _, s1 = po.decode_stream_i16(o_id, o_st, pool, keep=False)
    reps = max(1, int(seconds / max(s1, 1e-3)))
    _, _, secs = po.decode_stream_i16(o_id, o_st, pool * reps, keep=False)
    print(len(pool) * reps, secs)


if __name__ == "__main__":
    main()
