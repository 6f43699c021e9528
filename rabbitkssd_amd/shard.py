"""Multi-GPU plumbing of the distance path: query rows are independent, so the N x N (or
Q x R) matrix is sharded by rows, block-cyclically: blocks of ROW_BLOCK consecutive rows are
dealt round-robin to the ranks (block b -> rank b mod world).  Neighbouring genomes of a
sorted collection are close relatives that share posting lists, so a block keeps them on one
GPU (and lets the kernel walk them in pairs), while the round-robin still balances the
triangle.  The reference index is sent to every rank with ONE broadcast and per-rank hits are
only concatenated -- there is no reduction.  Backend-agnostic: "nccl" (= RCCL over xGMI) on
GPUs, "gloo" in the CPU tests."""
import math

import numpy as np


ROW_BLOCK = 32  # rk_dist_opts.row_block used by the multi-GPU callers (whole 32-genome blocks: the tile kernel counts a tile once)


def rank_rows(n_rows, rank, world, row_block=ROW_BLOCK):
    """row_first/row_step of this rank (what rk_dist_opts takes, together with row_block)
    and the row indices it owns."""
    rows = np.arange(n_rows, dtype=np.int64)
    return rank, world, rows[(rows // row_block) % world == rank]


def rank_pairs(n_genomes, rank, world, row_block=ROW_BLOCK):
    """all-vs-all pairs (j > i) whose row i belongs to this rank."""
    rows = rank_rows(n_genomes, rank, world, row_block)[2]
    return int(((n_genomes - 1) - rows).sum())


def weak_scaling_genomes(base_genomes, world):
    """dataset size that keeps the pairs per GPU at the 1-GPU level: N(N-1)/2 ~ world * P1."""
    return int(round(base_genomes * math.sqrt(world)))


def broadcast_blob(blob, src, device, dist):
    """Broadcast a 1-D uint8 torch tensor from `src` (other ranks pass None): size first,
    then the payload in one collective.  Returns the tensor on every rank.  A collective that
    fails or times out (the process group's timeout: see init_timeout) is reported with the rank
    and the byte count instead of a bare backend error."""
    import torch
    rank = dist.get_rank()
    # (the byte count of the message is fixed on the host BEFORE anything can fail: the handler must not touch the device)
    n_known = "%d bytes" % blob.numel() if rank == src else "size not yet received"
    size = torch.tensor([blob.numel() if rank == src else 0], dtype=torch.int64, device=device)
    try:
        dist.broadcast(size, src)
        n = int(size.item())
        n_known = "%d bytes" % n
        if rank != src:
            blob = torch.empty(n, dtype=torch.uint8, device=device)
        dist.broadcast(blob, src)
        if device.type == "cuda":
            torch.cuda.synchronize(device)   # RCCL errors surface at the synchronisation
    except Exception as e:  # noqa: BLE001 -- every backend raises its own type
        raise RuntimeError("rank %d/%d: index broadcast from rank %d (%s) failed: %s"
                           % (rank, dist.get_world_size(), src, n_known, e)) from e
    return blob


def init_timeout():
    """timeout for torch.distributed.init_process_group: a rank that never arrives (a dead GPU, a missing xGMI link) must end
    the job with an error after two minutes, not hang it"""
    import datetime
    return datetime.timedelta(seconds=120)


def gather_hits(local_hits, dist, dst=0):
    """Concatenate per-rank structured hit arrays on `dst`, sorted by (row, col)."""
    world = dist.get_world_size()
    parts = [None] * world
    dist.all_gather_object(parts, local_hits.tobytes())
    if dist.get_rank() != dst:
        return None
    merged = np.concatenate([np.frombuffer(p, dtype=local_hits.dtype) for p in parts])
    return merged[np.lexsort((merged["col"], merged["row"]))]


# ---- sharded build (round 5): hash ranges build, rows join, ONE all-to-all of 12-byte tile records in between ------------
REC_BYTES = 12


def exchange_records(send, send_counts, dist, device):
    """all-to-all of tile records: `send` = this rank's records contiguous by destination (uint8 tensor, REC_BYTES each),
    send_counts[d] = records for rank d.  Returns (recv uint8 tensor on `device`, records received).  RCCL: one
    all_to_all_single on device memory; gloo (CPU rehearsal): the same collective on host copies, or -- where the backend has no
    all-to-all -- one broadcast per source rank."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = torch.tensor(send_counts, dtype=torch.int64)
    table = [torch.zeros(world, dtype=torch.int64) for _ in range(world)]
    if dist.get_backend() == "nccl":
        table = [t.to(device) for t in table]
        dist.all_gather(table, counts.to(device))
        table = [t.cpu() for t in table]
    else:
        dist.all_gather(table, counts)
    recv_counts = [int(table[r][rank]) for r in range(world)]
    n_recv = sum(recv_counts)
    in_split = [int(c) * REC_BYTES for c in send_counts]
    out_split = [c * REC_BYTES for c in recv_counts]
    if dist.get_backend() == "nccl":
        recv = torch.empty(max(1, n_recv * REC_BYTES), dtype=torch.uint8, device=device)
        dist.all_to_all_single(recv[: n_recv * REC_BYTES], send[: sum(in_split)], out_split, in_split)
        return recv, n_recv
    host = send[: sum(in_split)].cpu()
    recv = torch.empty(n_recv * REC_BYTES, dtype=torch.uint8)
    try:
        dist.all_to_all_single(recv, host, out_split, in_split)
    except Exception:  # noqa: BLE001 -- a backend without all-to-all: every rank's buffer travels whole, each takes its slice
        at = 0
        for r in range(world):
            n_r = int(table[r].sum()) * REC_BYTES
            buf = host if r == rank else torch.empty(n_r, dtype=torch.uint8)
            dist.broadcast(buf, r)
            lo = int(table[r][:rank].sum()) * REC_BYTES
            recv[at:at + out_split[r]] = buf[lo:lo + out_split[r]]
            at += out_split[r]
    out = torch.empty(max(1, n_recv * REC_BYTES), dtype=torch.uint8, device=device)
    out[: n_recv * REC_BYTES] = recv.to(device)
    return out, n_recv


def sharded_join_index(ctx, sketches, hash_bits, dist, device, stream=0):
    """this rank's join-only index of a sharded all-vs-all: build the lists of its hash range, exchange the tile records, sort what
    arrives (rk_index_build_shard -> rk_index_shard_records / _pack -> all-to-all -> rk_index_join_shard).  Returns
    (join index, part index, seconds spent [build, exchange, join build], records sent, records received)."""
    import time
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    t0 = time.perf_counter()
    part = ctx.index_build_shard(sketches, hash_bits, rank, world)
    counts = part.shard_records(world)
    send = torch.empty(max(1, sum(counts) * REC_BYTES), dtype=torch.uint8, device=device)
    part.shard_pack(send.data_ptr(), stream)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    t1 = time.perf_counter()
    recv, n_recv = exchange_records(send, counts, dist, device)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    t2 = time.perf_counter()
    join = ctx.index_join_shard(part, recv.data_ptr(), n_recv)
    t3 = time.perf_counter()
    return join, part, (t1 - t0, t2 - t1, t3 - t2), sum(counts), n_recv
