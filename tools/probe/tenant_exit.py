import sys, os, time, faulthandler
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
faulthandler.dump_traceback_later(12, exit=True)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from lewton_amd import header, streamgen as sg, _native as N
from lewton_amd.shard import Sharder
SHARE, POL = int(sys.argv[1]), int(sys.argv[2])
N.lw_debug_ring_policy(POL)
setup = sg.stereo_setup(44100, 8, 11)
idp, _, stp = setup.headers()
ident = header.read_header_ident(idp); st = header.read_header_setup(stp, 2, (8, 11))
pool = sg.make_stream(setup, "L", 64, seed=9)
sh = Sharder(ident, st, [0, 0], 4096, "i16", share_cus=bool(SHARE))
assert sh.set_entropy_on_device(True)
rng = np.random.default_rng(1)
call = sh.marshal([(k // 16, pool[int(i)]) for k, i in enumerate(rng.integers(0, len(pool), 8192))])
for k in range(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
    if sh.in_flight == 3:
        sh.collect_pinned(want_results=False); sh.release()
    sh.submit(call, 0)
while sh.in_flight:
    sh.collect_pinned(want_results=False); sh.release()
print("done, closing", file=sys.stderr, flush=True)
sh.close()
print("closed", file=sys.stderr, flush=True)
