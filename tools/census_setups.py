#!/usr/bin/env python3
"""(no GPU needed) Census of the planner over random setups: which kernels the long / short blocks of `n` draws of
streamgen.random_setup are routed to and why (lw_debug_plan_census).   python tools/census_setups.py [n] [first seed]"""
import collections
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lewton_amd import _native as N, header, streamgen as sg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fields = {k: collections.Counter() for k in ("long", "short", "transitions", "entropy")}
for seed in range(first, first + n):
    st = sg.random_setup(np.random.default_rng(seed))
    idp, _cmt, stp = st.headers()
    ident = header.read_header_ident(idp)
    s = header.read_header_setup(stp, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    buf = C.create_string_buffer(2048)
    N.lib.lw_debug_plan_census(ident._h, s._h, buf, 2048)
    for part in buf.value.decode().split(" | "):
        k, v = part.split("=", 1)
        fields[k][v] += 1
for k in ("long", "short", "transitions", "entropy"):
    c = fields[k]
    print("== %s (%d setups, seeds %d..%d)" % (k, n, first, first + n - 1))
    groups = collections.Counter()
    for v, cnt in c.items():
        groups["none" if v == "none" else "generic" if v.startswith("generic") else "behind k_prep" if "k_prep" in v else "as it is"] += cnt
    if k in ("long", "short"):
        print("   " + ", ".join("%s %.1f %%" % (g, 100.0 * cnt / n) for g, cnt in groups.most_common()))
    for v, cnt in c.most_common():
        print("  %5.1f %%  %s" % (100.0 * cnt / n, v))
