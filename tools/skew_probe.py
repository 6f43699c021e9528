#!/usr/bin/env python3
"""Developer probe: rk_index_build over a collection whose hashes crowd the low end of the hash space (h -> 2^bits (h / 2^bits)^p:
the fullest buckets of the bucket sort hold several times the mean) -- real sketches are pieces of k-mers, not uniform values.
    python3 tools/skew_probe.py [n_genomes] [power x 10; 0 / 1: quarters of the hash space filled 7 : 5 : 3 : 1, on one / two levels]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from rabbitkssd_amd import capi, synth  # noqa: E402


def main(n=10000, p10=15):
    bits = 28
    names, h, off = synth.clade_sketches(n, 1220, bits)
    p = p10 / 10.0
    x = h.astype(np.float64) / (1 << bits)
    if p10 in (0, 1):   # the canonical k-mer's leading base: quarters of the hash space filled 7 : 5 : 3 : 1 (0), and once more inside every quarter (1)
        cum = np.array([0.0, 7.0, 12.0, 15.0, 16.0]) / 16.0

        def quarters(u):   # u in [0, 1) uniform -> position in [0, 1) with the quarters filled 7 : 5 : 3 : 1
            q = np.minimum(3, np.searchsorted(cum, u, side="right") - 1)
            return (q + (u - cum[q]) / (cum[q + 1] - cum[q])) / 4.0
        y = quarters(x)
        if p10 == 1:
            y = (np.floor(y * 4.0) + quarters((y * 4.0) % 1.0)) / 4.0
        x, p = y, 1.0
    hs = np.minimum((1 << bits) - 1, np.floor(x ** p * (1 << bits))).astype(np.uint32)
    # per genome: sorted already (monotone map); drop the repeats the map creates
    gid = np.repeat(np.arange(n, dtype=np.int64), np.diff(off).astype(np.int64))
    key = (gid << bits) | hs
    keep = np.concatenate(([True], key[1:] != key[:-1]))
    hs, gid = hs[keep], gid[keep]
    off2 = np.zeros(n + 1, dtype=np.uint64)
    off2[1:] = np.cumsum(np.bincount(gid, minlength=n))
    top = np.bincount(hs >> (bits - 13), minlength=1 << 13)
    print("skew %.1f: %d hashes, buckets of 2^13: mean %.0f, fullest %d, over 4096: %d" % (p, len(hs), top.mean(), top.max(), int((top > 4096).sum())), flush=True)
    ctx = capi.Context(0)
    sk = ctx.sketches_from_host(hs, off2)
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.time()
        idx = ctx.index_build(sk, bits)
        print("index build %d: %.3f ms (fast=%d products=%d)" % (rep, (time.time() - t0) * 1e3, idx.built_fast, idx.products), flush=True)
    hits, _ = ctx.dist_rows(idx, None, 1, 0, 20, 0.05)
    print("hits", len(hits))


if __name__ == "__main__":
    main(*[int(x) for x in sys.argv[1:]])
