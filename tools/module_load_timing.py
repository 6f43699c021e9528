#!/usr/bin/env python3
"""What the first call into each translation unit of librabbitkssd.so costs (code object load + first allocations): a tiny
collection, so that the work itself is nothing.  python3 tools/module_load_timing.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rabbitkssd_amd import capi, synth  # noqa: E402

t = [time.time()]
ctx = capi.Context(0)
t.append(time.time())
names, h, off = synth.clade_sketches(64, 100, 24, seed=1)
sk = ctx.sketches_from_host(h, off)          # rk_sketch.o: classification kernels
t.append(time.time())
idx = ctx.index_build(sk, 24)                # rk_index.o
t.append(time.time())
hits, _ = ctx.dist_rows(idx, None, 1, 0, 20, 0.05)   # rk_dist.o (near kernel + fallback)
t.append(time.time())
hits2, _ = ctx.dist_rows(idx, sk, 0, 0, 20, 0.05)     # rk_distq.o
t.append(time.time())
idx2 = ctx.index_build(sk, 24)
t.append(time.time())
hits, _ = ctx.dist_rows(idx2, None, 1, 0, 20, 0.05)
t.append(time.time())
lab = ["context", "sketches_from_host (rk_sketch.o)", "index_build (rk_index.o)", "dist_rows self (rk_dist.o)", "dist_rows queries (rk_distq.o)",
       "index_build again", "dist_rows self again"]
for a, b, c in zip(lab, t[:-1], t[1:]):
    print("%-40s %8.2f ms" % (a, (c - b) * 1e3))
