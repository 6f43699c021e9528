#!/usr/bin/env python3
"""Developer probe: ONE shard of an S-shard build of the scale collection -- shard build, join build from the records all shards
would send it, join.   python3 tools/shard_probe.py [n_genomes] [S]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from rabbitkssd_amd import capi, shard, synth  # noqa: E402


def main(n=500000, S=8):
    os.environ.setdefault("RK_POOL_LIMIT_MB", "196608")
    ctx = capi.Context(0)
    h, off, sp = synth.scale_collection_torch(n)
    torch.cuda.synchronize()   # rk_sketches_from_dev copies on the context's stream: the arrays must be complete
    sk = ctx.sketches_from_dev(h.data_ptr(), off.data_ptr(), n)
    sends, counts, part0 = [], [], None
    for r in range(S):
        for rep in range(2 if r == 0 else 1):
            torch.cuda.synchronize()
            t0 = time.time()
            part = ctx.index_build_shard(sk, 28, r, S)
            torch.cuda.synchronize()
            print("shard %d build %.2f ms (postings %d)" % (r, (time.time() - t0) * 1e3, part.total), flush=True)
        cnt = part.shard_records(S)
        buf = torch.empty(max(1, sum(cnt) * 12), dtype=torch.uint8, device="cuda")
        part.shard_pack(buf.data_ptr())
        torch.cuda.synchronize()
        sends.append(buf); counts.append(cnt)
        if part0 is None:
            part0 = part
    d = 0
    recv = torch.cat([sends[r][12 * sum(counts[r][:d]): 12 * sum(counts[r][:d + 1])] for r in range(S)])
    n_recv = sum(counts[r][d] for r in range(S))
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.time()
        join = ctx.index_join_shard(part0, recv.data_ptr(), n_recv)
        torch.cuda.synchronize()
        print("join build %.2f ms (%d records, tiles %s)" % ((time.time() - t0) * 1e3, n_recv, join.tile_stats()), flush=True)


if __name__ == "__main__":
    main(*[int(x) for x in sys.argv[1:]])
