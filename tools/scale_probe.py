#!/usr/bin/env python3
"""Developer probe of the scale leg: a Zipf-species collection generated on the device, rk_index_build, one sparse self join.
    python3 tools/scale_probe.py [n_genomes] [max_species]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from rabbitkssd_amd import capi, synth  # noqa: E402


def main(n=100000, max_species=10000):
    ctx = capi.Context(0)
    t0 = time.time()
    h, off, sp = synth.scale_collection_torch(n, max_species=max_species)
    torch.cuda.synchronize()
    print("generated %d genomes, %d hashes, %d species (largest %d) in %.1f s" % (n, len(h), int(sp.max()) + 1, int(torch.bincount(sp).max()), time.time() - t0), flush=True)
    off_u = off.to(torch.int64)
    sk = ctx.sketches_from_dev(h.data_ptr(), off_u.data_ptr(), n)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        idx = ctx.index_build(sk, 28)
        print("index build %d: %.2f ms (H=%d U=%d fast=%d products=%d) pool %s" % (rep, (time.time() - t0) * 1e3, idx.total, idx.distinct, idx.built_fast, idx.products,
                                                                               [x >> 20 for x in ctx.pool_stats()[:2]]), flush=True)
        if rep == 0:
            del idx
    cap = 1 << 26
    hits = torch.empty(cap * capi.HIT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    for i in range(3):
        torch.cuda.synchronize()
        t0 = time.time()
        ctx.dist_rows_dev(idx, 1, 0, 20, 0.05, hits.data_ptr(), cap, cnt.data_ptr() + 8 * i, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        print("join %d: %.3f ms, %d hits, kernel %s, products %d, tile stats %s" % (i, (time.time() - t0) * 1e3, int(cnt[i].item()), ctx.dist_kernel_name(idx, None, 1, 0, 20, 0.05),
                                                                          idx.products, idx.tile_stats()), flush=True)


if __name__ == "__main__":
    main(*[int(x) for x in sys.argv[1:]])
