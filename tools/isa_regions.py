#!/usr/bin/env python3
"""Static instruction counts per marked region of k_long<0,false,false> (stereo-only build with -DLW_MARKS).

Caveat: the marks are `asm volatile` statements and the build is not the shipped one -- it spills, and it carries ~60 16-bit
byte-shuffle operations per item (v_bitop3_b16 / v_lshlrev_b16 / v_or_b32_sdwa) that the production kernel does not have
(4 in total there; count them in the shipped code object with llvm-objdump before chasing them).  Use the region table for
the packed-arithmetic / LDS split, not for plain-VALU overhead."""
import collections, re, subprocess, sys, os, glob
os.makedirs("/tmp/kl", exist_ok=True)
os.chdir("/tmp/kl")
flags = sys.argv[1:]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
                "-DLW_EXP_STEREO_ONLY", "-DLW_MARKS", "-save-temps", "-c", "/root/repo/lewton_amd/csrc/lw_kernels_long.hip", "-o", "/tmp/kl/m.o"] + flags,
               check=True, stderr=subprocess.DEVNULL)
src = open(glob.glob("lw_kernels_long-hip-amdgcn*.s")[0]).read()
body = src[src.index("_Z6k_longILi0ELb0ELb0EEv"):]
body = body[:body.index("s_endpgm")]
region, order, cnt = "prologue", ["prologue"], collections.defaultdict(collections.Counter)
for line in body.splitlines():
    m = re.search(r"; LWMARK (\S+)", line)
    if m:
        region = m.group(1)
        if region not in order:
            order.append(region)
        continue
    m = re.match(r"\s+([a-z_0-9]+)", line)
    if not m or line.strip().startswith(";") or line.strip().startswith("."):
        continue
    op = m.group(1)
    if op.startswith("v_pk_"): k = "pk"
    elif op.startswith("v_mov") : k = "mov"
    elif op.startswith("v_"): k = "valu"
    elif op.startswith("ds_"): k = "lds"
    elif op == "s_nop": k = "nop"
    elif op == "s_waitcnt": k = "wait"
    elif op.startswith("s_"): k = "salu"
    elif op.startswith("global_") or op.startswith("scratch_") or op.startswith("buffer_"): k = "vmem"
    else: k = "other"
    cnt[region][k] += 1
tot = collections.Counter()
print("%-12s %5s %5s %5s %5s %5s %5s %5s %5s" % ("region", "pk", "valu", "mov", "lds", "nop", "wait", "salu", "vmem"))
for r in order:
    c = cnt[r]; tot.update(c)
    print("%-12s %5d %5d %5d %5d %5d %5d %5d %5d" % (r, c["pk"], c["valu"], c["mov"], c["lds"], c["nop"], c["wait"], c["salu"], c["vmem"]))
print("%-12s %5d %5d %5d %5d %5d %5d %5d %5d" % ("TOTAL", tot["pk"], tot["valu"], tot["mov"], tot["lds"], tot["nop"], tot["wait"], tot["salu"], tot["vmem"]))
