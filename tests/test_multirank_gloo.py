"""world_size-2 tests of the N>1 path (gloo): block-cyclic row sharding, one broadcast of the reference index,
concatenation of per-rank hits.  CPU tier: the per-rank compute is the oracle's (partition arithmetic and the
broadcast helper).  GPU tier (-m gpu): the same two ranks compute with the HIP kernels on the box's one card, and two
library contexts in one process exchange the index through rk_index_broadcast (what `rabbit_kssd --gpus N` does)."""
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


def test_canonical_skew_keeps_sketches_sets_and_fills_the_quarters_7_5_3_1():
    # the synthetic collections are uniform; synth.canonical_skew spreads their hashes as a sketcher's are (bench.py: alldist_variants)
    names, h, off = synth.clade_sketches(400, 300, 28, seed=5)
    for levels in (1, 2):
        hs, off2 = synth.canonical_skew(h, off, 28, levels=levels)
        assert len(off2) == len(off) and int(off2[-1]) == len(hs) <= len(h) and len(hs) > 0.99 * len(h)
        for g in range(len(names)):
            a = hs[int(off2[g]):int(off2[g + 1])]
            assert np.all(a[1:] > a[:-1])
        share = np.bincount(hs >> 26, minlength=4) / len(hs)
        assert np.allclose(share, np.array([7, 5, 3, 1]) / 16.0, atol=0.01)
    share16 = np.bincount(hs >> 24, minlength=16) / len(hs)
    assert abs(share16[0] - 49 / 256) < 0.01 and abs(share16[15] - 1 / 256) < 0.003


def test_partition_arithmetic():
    for n in (1, 2, 7, 100, 10000):
        for world in (1, 2, 4, 8):
            rows = np.concatenate([shard.rank_rows(n, r, world)[2] for r in range(world)])
            assert sorted(rows.tolist()) == list(range(n))
            assert sum(shard.rank_pairs(n, r, world) for r in range(world)) == n * (n - 1) // 2
    assert shard.weak_scaling_genomes(10000, 1) == 10000
    assert shard.weak_scaling_genomes(10000, 4) == 20000
    # dealing blocks of 32 rows round-robin keeps the triangle balanced: max/min pairs per rank within 5 % at 10k x 8 (the
    # kernels of the sparse self join cost the same per row whatever the number of columns behind it: what has to be even
    # is the number of rows, and it is to one block)
    p = [shard.rank_pairs(10000, r, 8) for r in range(8)]
    assert max(p) / min(p) < 1.05


# --------------------------------------------------------------------------- the same plumbing with the HIP path (GPU)
def _worker_gpu(rank, port, outdir):
    """two ranks share cuda:0 (the GPU box has one card): rank 0 builds the index with the HIP kernels, packs it into
    one blob, the blob travels in ONE broadcast, rank 1 unpacks it, both compute their block-cyclic rows with
    rk_dist_rows and the hits are concatenated; the union must equal the oracle's unsharded result"""
    import torch
    import torch.distributed as dist
    from rabbitkssd_amd import capi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        torch.cuda.set_device(0)
        ctx = capi.Context(0)
        names, h, off = synth.clade_sketches(700, 90, BITS, seed=23)
        blob = None
        if rank == 0:
            index = ctx.index_build(ctx.sketches_from_host(h, off), BITS)
            dev_blob = torch.empty(index.blob_bytes, dtype=torch.uint8, device="cuda")
            index.pack_dev(dev_blob.data_ptr(), dev_blob.numel())
            blob = dev_blob.cpu()
        blob = shard.broadcast_blob(blob, 0, torch.device("cpu"), dist)
        if rank != 0:
            dev_blob = blob.cuda()
            index = ctx.index_unpack_dev(dev_blob.data_ptr(), dev_blob.numel())
        mine, _ = ctx.dist_rows(index, None, 1, 0, 20, 0.1, row_first=rank, row_step=WORLD, row_block=shard.ROW_BLOCK)
        assert np.all(index.shard_of(mine, WORLD, shard.ROW_BLOCK) == rank)   # blocks of the index's internal genome order
        merged = shard.gather_hits(mine, dist, 0)
        if rank == 0:
            postings, counts = ok.index_build32(h, off, BITS)
            want, _ = ok.index_dist32(counts, BITS, postings, np.diff(off).astype(np.uint32), h, off, 1, 0, 20, 0.1)
            assert len(want) > 1000 and len(merged) == len(want)
            for f in ("row", "col", "common", "size0", "size1"):
                assert np.array_equal(merged[f], want[f]), f
            assert np.array_equal(merged["dist"], want["dist"])
            open(os.path.join(outdir, "ok"), "w").write("%d" % len(merged))
        del index
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_row_sharding_hip_kernels_one_gpu(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker_gpu, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    assert int(open(tmp_path / "ok").read()) > 1000


@pytest.mark.gpu
def test_two_contexts_in_one_process_index_broadcast(tmp_path):
    """what `rabbit_kssd --gpus N` does: one context per GPU in one process (here both on device 0), the index
    replicated with rk_index_broadcast, every context computing its rows; dist: contiguous query blocks"""
    from rabbitkssd_amd import capi
    a, b = capi.Context(0), capi.Context(0)
    names, h, off = synth.clade_sketches(500, 60, BITS, seed=29)
    ia = a.index_build(a.sketches_from_host(h, off), BITS)
    ib = b.index_broadcast_from(ia)
    assert (ib.total, ib.distinct, ib.genomes, ib.sum_sq) == (ia.total, ia.distinct, ia.genomes, ia.sum_sq)
    full, _ = a.dist_rows(ia, None, 1, 0, 20, 0.2)
    pa, _ = a.dist_rows(ia, None, 1, 0, 20, 0.2, row_first=0, row_step=2, row_block=16)
    pb, _ = b.dist_rows(ib, None, 1, 0, 20, 0.2, row_first=1, row_step=2, row_block=16)
    merged = np.concatenate([pa, pb])
    merged = merged[np.lexsort((merged["col"], merged["row"]))]
    assert len(full) > 500 and merged.tobytes() == full.tobytes()
    # ref-vs-query on the copy: queries 250..499 as a block of their own
    q_off = (off[250:] - off[250]).astype(np.uint64)
    qs = b.sketches_from_host(h[int(off[250]):], q_off)
    got, _ = b.dist_rows(ib, qs, 0, 1, 20, 0.3)
    postings, counts = ok.index_build32(h, off, BITS)
    want, _ = ok.index_dist32(counts, BITS, postings, np.diff(off).astype(np.uint32), h[int(off[250]):], q_off, 0, 1, 20, 0.3)
    assert len(want) > 250 and got["row"].tolist() == want["row"].tolist() and got["col"].tolist() == want["col"].tolist()
    assert np.array_equal(got["common"], want["common"])


# --------------------------------------------------------------------------- sharded build: the exchange of tile records
def _worker_exchange(rank, port, outdir):
    """CPU tier: the all-to-all of 12-byte records between two gloo ranks (rank r sends 100 (d + 1) + 7 r records to rank d,
    each stamped with its source, destination and number)"""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        counts = [100 * (d + 1) + 7 * rank for d in range(WORLD)]
        recs = np.zeros((sum(counts), 3), dtype=np.uint32)
        at = 0
        for d, c in enumerate(counts):
            recs[at:at + c, 0], recs[at:at + c, 1], recs[at:at + c, 2] = rank, d, np.arange(c)
            at += c
        send = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy())
        recv, n = shard.exchange_records(send, counts, dist, torch.device("cpu"))
        got = recv.numpy()[: n * shard.REC_BYTES].view(np.uint32).reshape(-1, 3)
        want_n = sum(100 * (rank + 1) + 7 * r for r in range(WORLD))
        assert n == want_n and np.all(got[:, 1] == rank)
        at = 0
        for r in range(WORLD):   # contiguous by source rank, in the order they were packed
            c = 100 * (rank + 1) + 7 * r
            assert np.all(got[at:at + c, 0] == r) and np.array_equal(got[at:at + c, 2], np.arange(c))
            at += c
        open(os.path.join(outdir, "ok%d" % rank), "w").write("%d" % n)
    finally:
        dist.destroy_process_group()


def test_two_rank_record_exchange_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker_exchange, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    assert int(open(tmp_path / "ok0").read()) == 207 and int(open(tmp_path / "ok1").read()) == 407


def _worker_sharded_gpu(rank, port, outdir):
    """two ranks share cuda:0: every rank holds the sketches, builds the posting lists of ITS half of the hash space, the tile
    records travel in one all-to-all (through the host under gloo), every rank sorts what arrived and joins ITS rows; the union
    of the hits is the oracle's unsharded result -- no index is replicated"""
    import torch
    import torch.distributed as dist
    from rabbitkssd_amd import capi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        torch.cuda.set_device(0)
        ctx = capi.Context(0)
        names, h, off = synth.clade_sketches(1500, 120, BITS, strains_per_clade=40, seed=53)
        sk = ctx.sketches_from_host(h, off)
        join, part, secs, n_sent, n_recv = shard.sharded_join_index(ctx, sk, BITS, dist, torch.device("cuda", 0))
        assert join.products == 6 and n_sent > 0 and n_recv > 0
        mine, _ = ctx.dist_rows(join, None, 1, 0, 20, 0.1)
        assert np.all(part.shard_of(mine, WORLD, shard.ROW_BLOCK) == rank)
        merged = shard.gather_hits(mine, dist, 0)
        if rank == 0:
            postings, counts = ok.index_build32(h, off, BITS)
            want, _ = ok.index_dist32(counts, BITS, postings, np.diff(off).astype(np.uint32), h, off, 1, 0, 20, 0.1)
            assert len(want) > 1000 and len(merged) == len(want)
            for f in ("row", "col", "common", "size0", "size1"):
                assert np.array_equal(merged[f], want[f]), f
            assert np.array_equal(merged["dist"], want["dist"])
            open(os.path.join(outdir, "ok"), "w").write("%d" % len(merged))
        del join, part
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_sharded_build_and_join_one_gpu(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker_sharded_gpu, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    assert int(open(tmp_path / "ok").read()) > 1000
