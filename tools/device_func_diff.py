#!/usr/bin/env python3
"""Per-kernel comparison of two gfx950 disassemblies (llvm-objdump -d of the unbundled code object): which kernels' instruction
streams differ.  Addresses and encodings are dropped; only mnemonics and operands count.
    tools/device_func_diff.py old.dis new.dis"""
import re
import sys


def funcs(path):
    out, cur = {}, None
    for ln in open(path, errors="replace"):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None or not ln.startswith("\t"):
            continue
        ins = ln.split("//")[0].strip()
        ins = re.sub(r"<[^>]+>", "<L>", ins)            # symbolic branch targets
        if ins and not ins.startswith("s_nop") and not ins.startswith("s_code_end"):
            cur.append(ins)
    return out


a, b = funcs(sys.argv[1]), funcs(sys.argv[2])
same = changed = 0
for k in sorted(set(a) | set(b)):
    if k not in a:
        print("NEW     %6d  %s" % (len(b[k]), k))
    elif k not in b:
        print("GONE    %6d  %s" % (len(a[k]), k))
    elif a[k] != b[k]:
        changed += 1
        print("CHANGED %6d -> %6d  %s" % (len(a[k]), len(b[k]), k))
    else:
        same += 1
print("%d kernels identical, %d changed" % (same, changed))
