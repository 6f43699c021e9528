"""world_size-2 test of the N>1 path on CPU (gloo): interleaved row sharding, one broadcast of
the reference index, concatenation of per-rank hits.  The per-rank compute is done by the
oracle here (no GPU in this tier); on the GPU box bench.py drives the same plumbing over
RCCL with the HIP kernels."""
import os
import socket

import numpy as np
import pytest

from oracle import oracle as ok
from rabbitkssd_amd import shard, synth

WORLD = 2
BITS = 20


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, outdir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        names, h, off = synth.clade_sketches(300, 80, BITS, seed=21)
        n = len(names)
        blob = None
        if rank == 0:  # rank 0 owns the index; peers receive it in one broadcast
            postings, counts = ok.index_build32(h, off, BITS)
            sizes = np.diff(off).astype(np.uint32)
            payload = np.concatenate([np.array([len(postings), len(sizes)], dtype=np.uint64).view(np.uint8),
                                      postings.view(np.uint8), counts.view(np.uint8), sizes.view(np.uint8)])
            blob = torch.from_numpy(payload.copy())
        blob = shard.broadcast_blob(blob, 0, torch.device("cpu"), dist)
        raw = blob.numpy()
        n_post, n_sizes = (int(x) for x in raw[:16].view(np.uint64))
        p0 = 16
        postings = raw[p0:p0 + 4 * n_post].view(np.uint32)
        counts = raw[p0 + 4 * n_post:p0 + 4 * n_post + 4 * (1 << BITS)].view(np.uint32)
        sizes = raw[p0 + 4 * n_post + 4 * (1 << BITS):].view(np.uint32)
        assert len(sizes) == n_sizes == n
        first, step, rows = shard.rank_rows(n, rank, WORLD)
        hits, _ = ok.index_dist32(counts, BITS, postings, sizes, h, off, 1, 0, 20, 0.1)
        mine = hits[(hits["row"] // shard.ROW_BLOCK) % step == first]          # this rank's rows only
        assert set(np.unique(mine["row"])).issubset(set(rows.tolist()))
        merged = shard.gather_hits(mine, dist, 0)
        pairs = torch.tensor([shard.rank_pairs(n, rank, WORLD)], dtype=torch.int64)
        dist.all_reduce(pairs)
        if rank == 0:
            assert int(pairs.item()) == n * (n - 1) // 2     # the shards tile the triangle exactly
            assert merged.tobytes() == hits.tobytes()        # union of shards == unsharded result
            open(os.path.join(outdir, "ok"), "w").write("%d" % len(merged))
    finally:
        dist.destroy_process_group()


def test_two_rank_row_sharding_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    assert int(open(tmp_path / "ok").read()) > 0


def test_partition_arithmetic():
    for n in (1, 2, 7, 100, 10000):
        for world in (1, 2, 4, 8):
            rows = np.concatenate([shard.rank_rows(n, r, world)[2] for r in range(world)])
            assert sorted(rows.tolist()) == list(range(n))
            assert sum(shard.rank_pairs(n, r, world) for r in range(world)) == n * (n - 1) // 2
    assert shard.weak_scaling_genomes(10000, 1) == 10000
    assert shard.weak_scaling_genomes(10000, 4) == 20000
    # dealing blocks of 16 rows round-robin keeps the triangle balanced: max/min pairs per rank
    # within 3 % at 10k x 8
    p = [shard.rank_pairs(10000, r, 8) for r in range(8)]
    assert max(p) / min(p) < 1.03
