import sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np
from lewton_amd import audio, header, streamgen as sg
from lewton_amd.ring import Ring
P = int(sys.argv[1]); NB = int(sys.argv[2]); SL = int(sys.argv[3]) if len(sys.argv) > 3 else 3
setup = sg.stereo_setup(44100, 8, 11)
idp, _, stp = setup.headers()
ident = header.read_header_ident(idp); st = header.read_header_setup(stp, 2, (8, 11))
dec = audio.decoder_for(ident, st, 0)
pool = sg.make_stream(setup, "L", 512, seed=9)
ring = Ring(dec, SL, P, "i16"); assert ring.set_entropy_on_device(True)
streams = 256; per = P // streams
rng = np.random.default_rng(1)
pwrs = [audio.PreviousWindowRight() for _ in range(streams)]
batches = [ring.marshal([(pool[int(i)], pwrs[k // per]) for k, i in enumerate(rng.integers(0, len(pool), P))]) for _ in range(8)]
ts = []; tstage = []; tlaunch = []; tcol = []
def run(n):
    for k in range(n):
        if ring.in_flight == ring.slots:
            a = time.perf_counter(); ring.collect_nocopy(); ring.release(); tcol.append(time.perf_counter() - a)
            ts.append(time.perf_counter())
        a = time.perf_counter(); ring.stage(batches[k % 8], 0); b = time.perf_counter(); ring.launch(); c = time.perf_counter()
        tstage.append(b - a); tlaunch.append(c - b)
    while ring.in_flight:
        ring.collect_nocopy(); ring.release(); ts.append(time.perf_counter())
run(4); ts.clear(); tstage.clear(); tlaunch.clear(); tcol.clear()
t0 = time.perf_counter(); run(NB); dt = time.perf_counter() - t0
d = np.diff(np.array(ts)) * 1e6
print("slots=%d P=%d NB=%d: %.2f M packets/s; period us: median %.0f p10 %.0f p90 %.0f max %.0f; first 8: %s" % (SL, P, NB, NB * P / dt / 1e6, np.median(d), np.percentile(d, 10), np.percentile(d, 90), d.max(), np.round(d[:8])))
print("  stage us median %.0f max %.0f; launch us median %.0f max %.0f; collect wait median %.0f" % (np.median(tstage) * 1e6, max(tstage) * 1e6, np.median(tlaunch) * 1e6, max(tlaunch) * 1e6, np.median(tcol) * 1e6))
big = np.argsort(d)[-6:]
print("  largest periods at batch", sorted(big.tolist()), np.round(d[sorted(big.tolist())]))

print("  periods 100..160:", " ".join("%d" % x for x in d[100:160]))
print("  collect waits 100..160:", " ".join("%d" % (x * 1e6) for x in tcol[100:160]))
