"""probe: per-call timing of the sharder on logical shards of one GPU.  usage: sharder_probe.py [shards] [calls] [packets per shard] [share CUs: 0 (default) | 1] [ring policy: -1 (default: by tenancy) | bit 0 kernels in launch order + bit 1 copies by the device's copier]"""
import sys, os, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # as bench.py: every slot stream of the rings on a hardware queue of its own
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from lewton_amd import header, streamgen as sg
from lewton_amd.shard import Sharder
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 200
P = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
SHARE = (int(sys.argv[4]) if len(sys.argv) > 4 else 0) != 0
POLICY = int(sys.argv[5]) if len(sys.argv) > 5 else -1
from lewton_amd import _native as _N
_N.lw_debug_ring_policy(POLICY)
setup = sg.stereo_setup(44100, 8, 11)
idp, _, stp = setup.headers()
ident = header.read_header_ident(idp); st = header.read_header_setup(stp, 2, (8, 11))
pool = sg.make_stream(setup, "L", 512, seed=9)
sh = Sharder(ident, st, [0] * G, P, "i16", share_cus=SHARE)
assert sh.set_entropy_on_device(True)
rng = np.random.default_rng(1)
S = 256; per = P // S; n_streams = G * S
calls = [sh.marshal([(k // per, pool[int(i)]) for k, i in enumerate(rng.integers(0, len(pool), n_streams * per))]) for _ in range(6)]
ts, tsub, tcol = [], [], []
def take():
    a = time.perf_counter(); sh.collect_pinned(want_results=False); sh.release(); tcol.append(time.perf_counter() - a); ts.append(time.perf_counter())
def run(n):
    for k in range(n):
        if sh.in_flight == 3:
            take()
        a = time.perf_counter(); sh.submit(calls[k % 6], 0); tsub.append(time.perf_counter() - a)
    while sh.in_flight:
        take()
run(8); ts.clear(); tsub.clear(); tcol.clear()
t0 = time.perf_counter(); run(NC); dt = time.perf_counter() - t0
d = np.diff(np.array(ts)) * 1e6
print("shards=%d (%s CUs each, policy %d) packets/shard=%d: %.2f M packets/s; call period us median %.0f p10 %.0f p90 %.0f max %.0f" % (G, "/".join(str(sh.shard_cus(g)) for g in range(G)), POLICY, P, NC * n_streams * per / dt / 1e6, np.median(d), np.percentile(d, 10), np.percentile(d, 90), d.max()))
print("  submit us median %.0f max %.0f; collect+release us median %.0f p90 %.0f" % (np.median(tsub) * 1e6, max(tsub) * 1e6, np.median(tcol) * 1e6, np.percentile(tcol, 90) * 1e6))
print("  periods 50..90:", " ".join("%d" % x for x in d[50:90]))
print("  submit  50..90:", " ".join("%d" % (x * 1e6) for x in tsub[50:90]))
print("  collect 50..90:", " ".join("%d" % (x * 1e6) for x in tcol[50:90]))
sh.close()
