"""Multi-GPU plumbing of the distance path: query rows are independent, so the N x N (or
Q x R) matrix is sharded by interleaved rows (row r -> rank r mod world), the reference
index is sent to every rank with ONE broadcast and per-rank hits are only concatenated --
there is no reduction.  Backend-agnostic: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the
CPU tests."""
import math

import numpy as np


def rank_rows(n_rows, rank, world):
    """row_first/row_step of this rank (what rk_dist_opts takes) and the row indices."""
    return rank, world, np.arange(rank, n_rows, world, dtype=np.int64)


def rank_pairs(n_genomes, rank, world):
    """all-vs-all pairs (j > i) whose row i belongs to this rank."""
    rows = np.arange(rank, n_genomes, world, dtype=np.int64)
    return int(((n_genomes - 1) - rows).sum())


def weak_scaling_genomes(base_genomes, world):
    """dataset size that keeps the pairs per GPU at the 1-GPU level: N(N-1)/2 ~ world * P1."""
    return int(round(base_genomes * math.sqrt(world)))


def broadcast_blob(blob, src, device, dist):
    """Broadcast a 1-D uint8 torch tensor from `src` (other ranks pass None): size first,
    then the payload in one collective.  Returns the tensor on every rank."""
    import torch
    rank = dist.get_rank()
    size = torch.tensor([blob.numel() if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(size, src)
    if rank != src:
        blob = torch.empty(int(size.item()), dtype=torch.uint8, device=device)
    dist.broadcast(blob, src)
    return blob


def gather_hits(local_hits, dist, dst=0):
    """Concatenate per-rank structured hit arrays on `dst`, sorted by (row, col)."""
    world = dist.get_world_size()
    parts = [None] * world
    dist.all_gather_object(parts, local_hits.tobytes())
    if dist.get_rank() != dst:
        return None
    merged = np.concatenate([np.frombuffer(p, dtype=local_hits.dtype) for p in parts])
    return merged[np.lexsort((merged["col"], merged["row"]))]
