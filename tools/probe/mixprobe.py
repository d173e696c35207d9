import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import bench_configs as bc
from lewton_amd import workloads as wl
from lewton_amd import batch as B
mode = int(sys.argv[1])
orig = B.Batch.__init__
def init(self, *a, **k):
    orig(self, *a, **k); self.debug_set_mix(mode)
B.Batch.__init__ = init
bc.Batch = B.Batch
r = bc.measure(wl.by_key("3"), steps=400, nb=4, verify=True)
print("mix_mode", mode, r["us_per_launch"], r["kernels"], r["parity"][:45])
