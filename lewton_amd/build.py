"""Builds the in-tree HIP library lewton_amd/_lib/liblewton_amd.so for gfx950.

hipcc cross-compiles without a GPU.  Every translation unit is compiled with -ffp-contract=off:
bit-exact parity with the reference needs individually rounded f32 multiplies and adds.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "_lib")
LIB = os.path.join(LIBDIR, "liblewton_amd.so")
SOURCES = ["lw_headers.cpp", "lw_entropy.cpp", "lw_dev_entropy.cpp", "lw_pool.cpp", "lw_runtime.cpp", "lw_batch.cpp", "lw_packet.cpp", "lw_fast.cpp", "lw_ring.cpp", "lw_shard.cpp", "lw_ogg.cpp", "lw_capi.cpp", "lw_kernels.hip", "lw_kernels_long.hip", "lw_kernels_big.hip", "lw_kernels_entropy.hip"]
ARCH = os.environ.get("LW_OFFLOAD_ARCH", "gfx950")  # e.g. gfx950:xnack- for an experiment
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-Wall",
         "-Wno-unused-result", "-pthread"]


# The long-block kernel writes its packed arithmetic itself; the SLP vectoriser only adds register moves around the
# scalar floor evaluation (-0.3 us per launch on MI355X).
PER_FILE_FLAGS = {"lw_kernels_long.hip": ["-fno-slp-vectorize"]}


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            if os.path.getmtime(os.path.join(root, f)) > t:
                return True
    return False


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(LIBDIR, s + ".o")
        objs.append(obj)
        cmd = [hipcc()] + FLAGS + PER_FILE_FLAGS.get(s, []) + os.environ.get("LW_EXTRA_FLAGS", "").split() + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on " + s)
        if verbose and out:
            print(out.decode())
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-pthread"]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
