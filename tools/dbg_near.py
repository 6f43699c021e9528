import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as ok
from rabbitkssd_amd import capi, synth
names, h, off = synth.clade_sketches(20, 150, 24, seed=1)
postings, counts = ok.index_build32(h, off, 24)
sizes = np.diff(off).astype(np.uint32)
want, _ = ok.index_dist32(counts, 24, postings, sizes, h, off, 1, 0, 20, 0.05, threads=2)
for env in ({}, {"RK_DIST_PAIR": "2"}, {"RK_DIST_NEAR": "0"}):
    for k, v in env.items():
        os.environ[k] = v
    c = capi.Context(0)
    for k in env:
        del os.environ[k]
    idx = c.index_build(c.sketches_from_host(h, off), 24)
    mine, _ = c.dist_rows(idx, None, 1, 0, 20, 0.05)
    print(env, c.dist_kernel_name(idx, None, 1, 0, 20, 0.05), len(mine), len(want), idx.order[:20])
    ws = set(zip(want["row"].tolist(), want["col"].tolist()))
    ms = set(zip(mine["row"].tolist(), mine["col"].tolist()))
    print("  missing", sorted(ws - ms)[:30])
    c.close()
