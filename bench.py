#!/usr/bin/env python3
"""bench.py -- headline benchmark: genome-pairs/s of `alldist` on 10,000 synthetic bacterial
sketches (L3K10, ~1,220 hashes each, clades of 10 strains), distance kernel only, inputs
resident in HBM (BASELINE.json metric; config "10,000 synthetic 5 Mb bacteria, L3K10,
alldist from precomputed .sketch/.dict").

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (intersection counting through the inverted index +
Jaccard->Mash epilogue + hit compaction, ONE kernel launch per rank) over this rank's query rows.
Multi-GPU: blocks of 32 consecutive query rows are dealt round-robin to the ranks (block-cyclic),
the reference index is built on rank 0 and sent to every peer with ONE RCCL broadcast (outside the
timed region); there is no data-path collective and no reduction.  Scaling is STRONG by default --
the metric is "10k bacteria at 1/2/4/8 GPUs", so the dataset stays 10,000 genomes at every N
(--scaling weak grows it to round(10000*sqrt(N)) genomes instead, constant pairs per GPU).

Prints ONE JSON line on rank 0 (contract in the task statement).  Besides the headline fields:
  roofline      distance kernel: achieved / frac = HBM bytes really moved per second (counters of profiles/pmc_traffic.json,
                else the kernel's own stream) over 8 TB/s; contract_* = SURVEY 8d's byte model (kept for continuity, exceeds 1);
                issue_frac = share of the kernel's duration the SIMDs issue vector instructions
  build_plus_dist   second headline: rk_index_build + the distance kernel from resident sketches, with the build's own block
  alldist_order     the same collection listed in shuffled / completion-jitter order: time, variant, compact share, same pairs
  scaling_rehearsal per-shard kernel ms of 1/2/4/8 row shards played on one GPU and the efficiency they predict
  cpu_baseline  the reference's own dist.cpp (oracle/_ref/ref_driver) timed on the host cores, N=1 only
  config3       BASELINE configs[3]: alldist over 50,000 sketches, strong scaling (where 8 GPUs have work)
  dist_rq       BASELINE configs[4] shape: 100,000 reference sketches x 1,000 3-Gb-genome queries (24-bit hashes),
                one fused kernel; queries shard contiguously over the ranks; reference index_dist beside it at N=1
  sketch        sketch k-mers/s (N=1): pass rate, the scan kernel's own roofline block, CPU port beside it
  setup         index build (cold / steady), PCIe-inclusive API path, `rabbit_kssd alldist` wall vs the reference's
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
HASH_BITS = 28          # L3K10: 4*(half_k - drlevel)
HASHES_PER_GENOME = 1220
KMER = 20
MAX_DIST = 0.05
TOOL = os.path.join(ROOT, "rabbitkssd_amd", "rabbit_kssd")
REF = os.path.join(ROOT, "oracle", "_ref", "ref_driver")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--genomes", type=int, default=10000)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--replicate", choices=["sketches", "blob"], default="sketches",
                    help="N > 1: how every rank gets the index -- broadcast of the CSR sketches + a build per rank (default), or "
                         "broadcast of rank 0's packed index")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to "
                    "rehearse the N>1 plumbing with all ranks on ONE GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--no-sketch", action="store_true")
    ap.add_argument("--no-sketch-big", action="store_true", help="skip the one-1-Gb-genome leg (K10 S7 L4)")
    ap.add_argument("--no-config3", action="store_true")
    ap.add_argument("--no-dist-rq", action="store_true")
    ap.add_argument("--no-orders", action="store_true", help="skip the shuffled / jitter order legs")
    ap.add_argument("--no-rehearsal", action="store_true", help="skip the 1/2/4/8 row-shard rehearsal")
    ap.add_argument("--no-variants", action="store_true", help="skip the wide-species / tiny-sketch legs")
    ap.add_argument("--sketch-genomes", type=int, default=1000, help="BASELINE configs[1]: 1,000 x 5 Mb")
    ap.add_argument("--sketch-length", type=int, default=5_000_000)
    ap.add_argument("--config3-genomes", type=int, default=50000)
    ap.add_argument("--bare-sketch", action="store_true",
                    help="also time the reference from a bare .sketch: its own transSketches (src/sketch.cpp:894-1021, ~80 s at 10,000 "
                         "genomes) + alldist; without it the figure of BASELINE.md (8-vCPU container) stands in, labelled as such")
    ap.add_argument("--no-scale", action="store_true", help="skip the large-collection leg (scale)")
    ap.add_argument("--scale-genomes", type=int, default=500000)
    return ap.parse_args()


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


class Env:
    """process-wide handles: rank layout, torch, the library context"""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        from rabbitkssd_amd import capi, shard
        self.torch, self.dist, self.capi, self.args = torch, dist, capi, args
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            if self.rank == 0:
                print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, self.world),
                      file=sys.stderr)
            sys.exit(2)
        if not torch.cuda.is_available():
            sys.exit("bench.py needs a GPU: the engine has no CPU fallback")
        if args.same_device:
            self.local_rank = 0
        torch.cuda.set_device(self.local_rank)
        if self.world > 1:
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, timeout=shard.init_timeout(),
                                        device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(args.backend, rank=self.rank, world_size=self.world, timeout=shard.init_timeout())
        # (the scale leg keeps ~60 GB of build temporaries in the context's pool: with the default 32 GiB limit on idle blocks every
        # build would go back to hipMalloc / hipFree -- hundreds of ms of driver time that are not the build's)
        os.environ.setdefault("RK_POOL_LIMIT_MB", "196608")
        self.ctx = capi.Context(self.local_rank)
        # the device's stream-read rate, measured in this very run, stated next to the nominal 8 TB/s (a 2 GiB buffer:
        # eight times the 256 MB Infinity Cache, 16 bytes per lane, HIP events around five launches)
        self.peak_measured = self.ctx.stream_read_gbs(2048, 5) if self.rank == 0 else None
        self.ctx.trim()
        self.dev = torch.device("cuda", self.local_rank)
        # a stream of our own: torch.cuda.Event measures the stream it is recorded on
        self.stream = torch.cuda.Stream(device=self.dev)

    def fence(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.int64, device=self.dev)
        self.dist.all_reduce(t)
        return int(t.item())

    def share_index(self, index, hashes=None, off=None, hash_bits=HASH_BITS):
        """every rank gets the index of rank 0's collection; returns (index, bytes moved per rank, seconds, mode).
        mode "sketches" (default): ONE broadcast of the CSR sketches (4 B per hash), then every rank runs rk_index_build --
        a third of the bytes of the index blob, and the builds run at the same time (10,000 genomes: 49 MB + 0.5 ms against a
        147 MB blob; 50,000: 244 MB + 2.6 ms against 733 MB).  mode "blob" (--replicate blob): rank 0 packs its index,
        one broadcast, every peer unpacks.  The clock runs barrier to barrier, from "rank 0 holds sketches and index in
        HBM" to "every rank holds the index"."""
        from rabbitkssd_amd import shard
        if self.world == 1:
            return index, index.blob_bytes, 0.0, "none"
        torch = self.torch
        mode = self.args.replicate if hashes is not None or self.rank != 0 else "blob"
        if mode == "sketches":
            dh = doff = None
            if self.rank == 0:
                dh = torch.from_numpy(np.ascontiguousarray(hashes).view(np.uint8)).to(self.dev)
                doff = torch.from_numpy(np.ascontiguousarray(off, dtype=np.uint64).view(np.uint8)).to(self.dev)
            torch.cuda.synchronize()
            self.dist.barrier()
            t0 = time.time()
            dh = shard.broadcast_blob(dh, 0, self.dev, self.dist)
            doff = shard.broadcast_blob(doff, 0, self.dev, self.dist)
            if self.rank != 0:
                sk = self.ctx.sketches_from_dev(dh.data_ptr(), doff.data_ptr(), doff.numel() // 8 - 1)
                index = self.ctx.index_build(sk, hash_bits)
                del sk
            torch.cuda.synchronize()
            self.dist.barrier()
            return index, dh.numel() + doff.numel(), time.time() - t0, mode
        blob, nbytes = None, 0
        if self.rank == 0:
            nbytes = index.blob_bytes
            blob = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        torch.cuda.synchronize()
        self.dist.barrier()
        t0 = time.time()
        if self.rank == 0:
            index.pack_dev(blob.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
        blob = shard.broadcast_blob(blob, 0, self.dev, self.dist)
        if self.rank != 0:
            index = self.ctx.index_unpack_dev(blob.data_ptr(), blob.numel(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        self.dist.barrier()
        return index, blob.numel(), time.time() - t0, mode

    def gather_hits_ms(self, hits_tensor, n_hits, itemsize):
        """this rank's first n_hits device hit records to rank 0 (what a user of N GPUs waits for after the kernels): counts
        first, then one gather of equal-sized padded buffers; returns milliseconds, barrier to barrier"""
        if self.world == 1:
            return 0.0
        torch = self.torch
        self.fence()
        t0 = time.time()
        on = self.dev if self.args.backend == "nccl" else torch.device("cpu")   # (gloo gathers host tensors: rehearsals only)
        cnt = torch.tensor([n_hits], dtype=torch.int64, device=on)
        all_cnt = [torch.zeros_like(cnt) for _ in range(self.world)]
        self.dist.all_gather(all_cnt, cnt)
        most = int(max(int(c.item()) for c in all_cnt))
        mine = hits_tensor[: most * itemsize].to(on)
        got = [torch.empty_like(mine) for _ in range(self.world)] if self.rank == 0 else None
        self.dist.gather(mine, got, dst=0)
        self.fence()
        return (time.time() - t0) * 1e3

    def share_sketches(self, hashes, off):
        """rank 0's host sketches to every rank (queries of the dist path are scattered by the host)"""
        if self.world == 1:
            return hashes, off
        box = [hashes, off] if self.rank == 0 else [None, None]
        self.dist.broadcast_object_list(box, src=0)
        return box[0], box[1]


def sharded_block(env, hashes, off, hash_bits, steps=20):
    """N > 1 (round 5): the all-vs-all sharded twice -- every rank builds the posting lists of ITS hash range
    (rk_index_build_shard), the 12-byte tile records change hands in ONE all-to-all (RCCL over xGMI), every rank sorts what arrived
    (rk_index_join_shard) and joins ITS rows; only the CSR sketches are replicated (one broadcast), no index is.  Times are barrier
    to barrier, max over ranks; rank 0 returns the report.  A failure (the collective has never run across real devices in the
    builder's rehearsals) is reported in the JSON, not raised."""
    from rabbitkssd_amd import capi, shard
    torch, ctx, rank, world = env.torch, env.ctx, env.rank, env.world
    if world == 1 or world & (world - 1):
        return None
    out = {}
    try:
        dh = doff = None
        if rank == 0:
            dh = torch.from_numpy(np.ascontiguousarray(hashes).view(np.uint8)).to(env.dev)
            doff = torch.from_numpy(np.ascontiguousarray(off, dtype=np.uint64).view(np.uint8)).to(env.dev)
        env.fence()
        t0 = time.perf_counter()
        dh = shard.broadcast_blob(dh, 0, env.dev, env.dist)
        doff = shard.broadcast_blob(doff, 0, env.dev, env.dist)
        sk = ctx.sketches_from_dev(dh.data_ptr(), doff.data_ptr(), doff.numel() // 8 - 1)
        env.fence()
        t_repl = time.perf_counter() - t0
        # twice: the first run fills the context's pool (hipMalloc) and loads RCCL's all-to-all; the second is reported
        for rep in range(2):
            if rep:
                del join, part
            env.fence()
            t0 = time.perf_counter()
            join, part, secs, n_sent, n_recv = shard.sharded_join_index(ctx, sk, hash_bits, env.dist, env.dev, env.stream.cuda_stream)
            env.fence()
            t_all = time.perf_counter() - t0
        hits_cap = 1 << 22
        hits = torch.empty(hits_cap * capi.HIT_DTYPE.itemsize, dtype=torch.uint8, device=env.dev)
        counters = torch.zeros(counter_slots(steps, 2), dtype=torch.int64, device=env.dev)

        def launch(i):
            ctx.dist_rows_dev(join, 1, 0, KMER, MAX_DIST, hits.data_ptr(), hits_cap, counters.data_ptr() + 8 * i, stream=env.stream.cuda_stream)
        elapsed, kernel_ms, _ = timed_steps(env, launch, steps, 2)
        my_hits = int(counters[2 + steps - 1].item())
        tot = env.sum_over_ranks(my_hits)
        gather_ms = env.max_over_ranks(env.gather_hits_ms(hits, min(my_hits, hits_cap), capi.HIT_DTYPE.itemsize))
        out = {"replicate_sketches_ms": env.max_over_ranks(t_repl) * 1e3, "shard_build_ms": env.max_over_ranks(secs[0]) * 1e3,
               "exchange_ms": env.max_over_ranks(secs[1]) * 1e3, "join_build_ms": env.max_over_ranks(secs[2]) * 1e3,
               "build_exchange_join_ms": env.max_over_ranks(t_all) * 1e3, "step_ms": elapsed / steps * 1e3, "kernel_ms_rank0": kernel_ms,
               "gather_ms": gather_ms, "hits": int(tot), "records_sent_rank0": int(n_sent), "records_received_rank0": int(n_recv),
               "note": "every rank builds the lists of its hash range, ONE all-to-all of 12-byte tile records, every rank sorts what arrived and "
                       "joins its rows: e2e_ms = replicate_sketches + build_exchange_join + step + gather (barrier to barrier, max over ranks)"}
        out["e2e_ms"] = out["replicate_sketches_ms"] + out["build_exchange_join_ms"] + out["step_ms"] + out["gather_ms"]
        del join, part, sk
    except Exception as e:  # noqa: BLE001
        out = {"error": "%s: %s" % (type(e).__name__, e)}
    return out if rank == 0 else None


def spread_steps(steps):
    return min(steps, 50)


def counter_slots(steps, warmup):
    """one zeroed u64 hit counter per launch of timed_steps: warm-up, timed stretch, per-step spread stretch"""
    return warmup + steps + spread_steps(steps) + 1


def timed_steps(env, launch, steps, warmup):
    """W untimed + K timed launches on env.stream, barrier + synchronize on both sides; returns
    (wall seconds max over ranks, mean kernel ms by HIP events on the launch stream, per-step ms [min, median, max]).
    The timed region holds nothing but the K launches between two events; the per-step spread comes from a second,
    untimed stretch of up to 50 launches with an event after each (an event costs ~1-2 us on the stream)."""
    torch = env.torch
    with torch.cuda.stream(env.stream):
        for i in range(warmup):
            launch(i)
        env.fence()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(env.stream)
        for i in range(steps):
            launch(warmup + i)
        ev1.record(env.stream)
        env.fence()
        elapsed = time.perf_counter() - t0
        n = spread_steps(steps)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        evs[0].record(env.stream)
        for i in range(n):
            launch(warmup + steps + i)   # (slots of their own: a launch ADDS its hits to its counter slot)
            evs[i + 1].record(env.stream)
        env.fence()
    per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n))
    spread = [per[0], per[len(per) // 2], per[-1]]
    return env.max_over_ranks(elapsed), ev0.elapsed_time(ev1) / steps, spread


def canonical_pairs(hits_tensor, n, capi, order=None):
    """sorted (row, col, common) of the first n device hit records; with `order`, genome i of the run is genome order[i]
    of the collection as generated: the pairs are mapped back so that runs over differently ordered input can be compared"""
    raw = hits_tensor[: n * capi.HIT_DTYPE.itemsize].cpu().numpy().tobytes()
    h = np.frombuffer(raw, dtype=capi.HIT_DTYPE)
    r, c = h["row"].astype(np.int64), h["col"].astype(np.int64)
    if order is not None:
        r, c = order[r], order[c]
    lo, hi = np.minimum(r, c), np.maximum(r, c)
    key = np.lexsort((h["common"], hi, lo))
    return np.stack([lo[key], hi[key], h["common"][key].astype(np.int64)])


def threshold_legs(env, index, n_pairs, steps=20):
    """the same resident index under looser thresholds: the tile kernel is output-sensitive -- it only starts the tiles that can
    hold a reportable pair -- so its pairs/s is an EFFECTIVE rate that falls as -D admits more of the matrix (-D 1.0: alldist's
    default, every pair that shares a hash)"""
    from rabbitkssd_amd import capi, shard
    torch, ctx = env.torch, env.ctx
    out = {}
    hits_cap = 1 << 22
    hits = torch.empty(hits_cap * capi.HIT_DTYPE.itemsize, dtype=torch.uint8, device=env.dev)
    for D in (MAX_DIST, 0.3, 1.0):
        counters = torch.zeros(counter_slots(steps, 2), dtype=torch.int64, device=env.dev)

        def launch(i, D=D, counters=counters):
            ctx.dist_rows_dev(index, 1, 0, KMER, D, hits.data_ptr(), hits_cap, counters.data_ptr() + 8 * i,
                              stream=env.stream.cuda_stream, row_block=shard.ROW_BLOCK)
        _, ms, _ = timed_steps(env, launch, steps, 2)
        ts = index.tile_stats(1, 0, KMER, D)
        out["D%g" % D] = {"max_dist": D, "kernel": ctx.dist_kernel_name(index, None, 1, 0, KMER, D), "kernel_ms": ms,
                          "hits": int(counters[2 + steps - 1].item()), "effective_pairs_per_s": n_pairs / (ms * 1e-3),
                          "tiles_started": ts[1], "tiles_with_records": ts[0],
                          "cells_formed_share": min(1.0, ts[1] * 1024.0 / n_pairs) if n_pairs else None}
    return out


def alldist_block(env, n_genomes, steps, warmup, keep=None, order_mode="sorted", build_reps=0, strains=10, tiny=0, skew=0):
    """alldist over n_genomes synthetic sketches, this rank's block-cyclic row shard; rank 0 returns the report.
    order_mode: the order the collection is listed in ("sorted" as generated, "shuffled", "jitter": synth.genome_order);
    strains: genomes per clade (> 10: a species tree, synth.strain_rates); tiny: extra 40-hash sketches; skew: the hashes fill the
    hash space as a canonical k-mer's leading bases do (synth.canonical_skew, levels = skew)"""
    from rabbitkssd_amd import capi, shard, synth
    torch, ctx, rank, world = env.torch, env.ctx, env.rank, env.world
    n_pairs = n_genomes * (n_genomes - 1) // 2
    t_cold = t_steady = 0.0
    index = None
    names = hashes = off = order = None
    sk = None
    if rank == 0:
        names, hashes, off = synth.clade_sketches(n_genomes, HASHES_PER_GENOME, HASH_BITS, kmer_size=KMER, strains_per_clade=strains,
                                                  tiny=tiny)
        n_genomes = len(names)
        n_pairs = n_genomes * (n_genomes - 1) // 2
        if skew:
            hashes, off = synth.canonical_skew(hashes, off, HASH_BITS, levels=skew)
        if order_mode != "sorted":
            order = synth.genome_order(n_genomes, order_mode)
            names, hashes, off = synth.permute_genomes(names, hashes, off, order)
        sk = ctx.sketches_from_host(hashes, off)
        t0 = time.time()
        index = ctx.index_build(sk, HASH_BITS)      # first call: the context's pool is cold (hipMalloc)
        t_cold = time.time() - t0
        del index
        ts = []
        for _ in range(5):
            t0 = time.time()
            index = ctx.index_build(sk, HASH_BITS)  # steady state: no allocation, one read-back
            ts.append(time.time() - t0)
            if _ < 4:
                del index
        t_steady = sorted(ts)[len(ts) // 2]
    fast = bool(index.built_fast) if rank == 0 else False
    stats = index.self_stats if rank == 0 else (0, 0, 0)
    index, nbytes, t_bcast, repl_mode = env.share_index(index, hashes, off)
    H, T = index.total, index.sum_sq
    hits_cap = 1 << 20
    hits = torch.empty(hits_cap * capi.HIT_DTYPE.itemsize, dtype=torch.uint8, device=env.dev)
    counters = torch.zeros(counter_slots(steps, warmup), dtype=torch.int64, device=env.dev)

    def launch(i):
        ctx.dist_rows_dev(index, 1, 0, KMER, MAX_DIST, hits.data_ptr(), hits_cap, counters.data_ptr() + 8 * i,
                          row_first=rank, row_step=world, stream=env.stream.cuda_stream, row_block=shard.ROW_BLOCK)

    # the first call over a fresh index (whatever the kernel builds lazily -- fallback list, tile records -- is in it)
    first = torch.zeros(1, dtype=torch.int64, device=env.dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.dist_rows_dev(index, 1, 0, KMER, MAX_DIST, hits.data_ptr(), hits_cap, first.data_ptr(), row_first=rank, row_step=world,
                      stream=env.stream.cuda_stream, row_block=shard.ROW_BLOCK)
    env.stream.synchronize()
    t_first = time.perf_counter() - t0
    products = index.products   # 1 slice records, 2 tile records, 4 the tile records came with the build (rk_index_products)
    elapsed, kernel_ms, spread = timed_steps(env, launch, steps, warmup)
    # cold steps: other options in between make the library forget what it learned about this (index, options) pair -- here
    # that rk_near_kernel's fallback list is empty --, so every launch of this stretch is a first one (two option sets taking turns)
    n_cold = min(steps, 20)
    cold_cnt = torch.zeros(2 * n_cold, dtype=torch.int64, device=env.dev)
    with torch.cuda.stream(env.stream):
        env.fence()
        evc0, evc1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        evc0.record(env.stream)
        for i in range(2 * n_cold):
            ctx.dist_rows_dev(index, 1, 0, KMER, MAX_DIST * (1.0 if i % 2 == 0 else 1.0 - 1e-9), hits.data_ptr(), hits_cap,
                              cold_cnt.data_ptr() + 8 * i, row_first=rank, row_step=world, stream=env.stream.cuda_stream,
                              row_block=shard.ROW_BLOCK)
        evc1.record(env.stream)
        env.fence()
    kernel_ms_cold = evc0.elapsed_time(evc1) / (2 * n_cold)
    per_launch = counters[:warmup + steps + spread_steps(steps)].cpu().numpy()
    my_hits = int(per_launch[warmup + steps - 1])   # the last timed launch
    if not (per_launch == my_hits).all():
        sys.exit("bench.py: launches of the same shard reported different hit counts: %s" % sorted(set(per_launch.tolist())))
    tot_hits = env.sum_over_ranks(my_hits)
    gather_ms = env.max_over_ranks(env.gather_hits_ms(hits, min(my_hits, hits_cap), capi.HIT_DTYPE.itemsize))
    kernel = ctx.dist_kernel_name(index, None, 1, 0, KMER, MAX_DIST, row_first=rank, row_step=world, row_block=shard.ROW_BLOCK)
    pairs = None
    if rank == 0 and world == 1 and my_hits <= hits_cap:
        pairs = canonical_pairs(hits, my_hits, capi, order)   # the last launch's records (every launch writes the same set)
    # resident sketches -> hits in HBM: index build + distance kernel, one call each, steady state
    t_bd = None
    if rank == 0 and world == 1 and build_reps:
        ts = []
        cnt2 = torch.zeros(build_reps, dtype=torch.int64, device=env.dev)
        for i in range(build_reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            idx2 = ctx.index_build(sk, HASH_BITS)
            ctx.dist_rows_dev(idx2, 1, 0, KMER, MAX_DIST, hits.data_ptr(), hits_cap, cnt2.data_ptr() + 8 * i,
                              stream=env.stream.cuda_stream, row_block=shard.ROW_BLOCK)
            env.stream.synchronize()
            ts.append(time.perf_counter() - t0)
            del idx2
        t_bd = sorted(ts)[len(ts) // 2]
    if keep is not None and rank == 0:
        keep.update(names=names, hashes=hashes, off=off, index=index, sk=sk)
    if rank != 0:
        return None
    # SURVEY.md 8d's byte model of one launch on this rank: 12 B per query hash (hash + two index offsets) + 4 B per posting
    # streamed (T = sum c_h^2) + 4 B per count cell produced; rank 0 holds ~1/world of each term
    b_contract = (12.0 * H + 4.0 * T) / world + 4.0 * shard.rank_pairs(n_genomes, 0, world)
    # what THIS kernel streams by construction: the 8-byte slice records it walks + the 40-byte hit records it writes
    # (counts stay in LDS; compact records carry their posting list)
    b_stream = 8.0 * stats[2] / world + 40.0 * my_hits
    tile_records = None
    tile_stats = None
    if kernel.startswith("rk_tile_kernel"):
        # the tile kernel's stream: its 8-byte tile records (one per posting list and pair of 32-genome blocks) + 40 B per hit
        tile_records = int(index.self_stats[3])
        b_stream = 8.0 * tile_records + 40.0 * my_hits
        ts = index.tile_stats(1, 0, KMER, MAX_DIST)
        tile_stats = {"tiles_with_records": ts[0], "tiles_started": ts[1], "tile_records": ts[2],
                      "cells_formed_share": min(1.0, ts[1] * 1024.0 / n_pairs)}
    return {
        "tile_records": tile_records, "tile_stats": tile_stats, "index_products": products,
        "value": n_pairs * steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "genomes": n_genomes, "pairs": n_pairs,
        "hashes": int(H), "postings_streamed_T": int(T), "hits": int(tot_hits), "steps": steps, "warmup": warmup,
        "kernel": kernel, "kernel_ms": kernel_ms, "kernel_ms_min_median_max": spread,
        "contract_bytes_per_launch": b_contract, "stream_bytes_per_launch": b_stream,
        "index_build_cold_ms": t_cold * 1e3, "index_build_ms": t_steady * 1e3, "index_built_fast": fast,
        "index_blob_bytes": int(index.blob_bytes), "replicate_mode": repl_mode, "replicate_bytes": int(nbytes),
        "replicate_ms": t_bcast * 1e3, "gather_ms": gather_ms,
        "e2e_ms": t_bcast * 1e3 + elapsed / steps * 1e3 + gather_ms,
        "slice_records": stats[0], "compact_share": (stats[1] / stats[0]) if stats[0] else None, "records_walked": stats[2],
        "build_plus_dist_ms": t_bd * 1e3 if t_bd else None, "pairs_canonical": pairs, "order": order_mode,
        "first_call_ms": t_first * 1e3, "kernel_ms_cold": kernel_ms_cold,
    }


def shard_rehearsal(env, index, n_genomes, steps=30):
    """one GPU plays every row shard of a 2-, 4- and 8-GPU run in turn (same kernel, same block-cyclic rows): per-shard
    kernel ms and the strong-scaling efficiency they predict, t(1) / (S x slowest shard of S) -- launch skew and the
    broadcast excluded; lets the driver's multi-GPU numbers be cross-checked"""
    from rabbitkssd_amd import capi, shard
    torch, ctx = env.torch, env.ctx
    hits_cap = 1 << 20
    hits = torch.empty(hits_cap * capi.HIT_DTYPE.itemsize, dtype=torch.uint8, device=env.dev)
    out = {}
    t1 = None
    for S in (1, 2, 4, 8):
        per = []
        for r in range(S):
            counters = torch.zeros(counter_slots(steps, 2), dtype=torch.int64, device=env.dev)

            def launch(i, r=r, S=S, counters=counters):
                ctx.dist_rows_dev(index, 1, 0, KMER, MAX_DIST, hits.data_ptr(), hits_cap, counters.data_ptr() + 8 * i,
                                  row_first=r, row_step=S, stream=env.stream.cuda_stream, row_block=shard.ROW_BLOCK)
            _, ms, _ = timed_steps(env, launch, steps, 2)
            per.append(ms)
        if S == 1:
            t1 = per[0]
        out[str(S)] = {"shard_ms": per, "slowest_ms": max(per), "predicted_efficiency": t1 / (S * max(per))}
    # the floor of any shard: ONE block of 32 rows (16 units, 16 waves) -- a kernel launch plus one unit's chain of dependent loads
    blocks = (n_genomes + shard.ROW_BLOCK - 1) // shard.ROW_BLOCK
    counters = torch.zeros(counter_slots(steps, 2), dtype=torch.int64, device=env.dev)

    def launch_one(i):
        ctx.dist_rows_dev(index, 1, 0, KMER, MAX_DIST, hits.data_ptr(), hits_cap, counters.data_ptr() + 8 * i,
                          row_first=0, row_step=blocks, stream=env.stream.cuda_stream, row_block=shard.ROW_BLOCK)
    _, ms, _ = timed_steps(env, launch_one, steps, 2)
    out["floor"] = {"rows": shard.ROW_BLOCK, "ms": ms,
                    "note": "one block of rows: a kernel launch + one unit's chain of dependent loads; no shard can be faster"}
    return out


XGMI_LINK_GBS = 153.0   # MI355X_MICROARCH.md: seven point-to-point links per GPU, ~153 GB/s each


def scale_block(env, n_genomes):
    """A collection where eight GPUs and HBM matter (one rank; round 5): `n_genomes` sketches generated on the device -- species
    sizes Zipf-distributed up to 10,000 strains, sketch sizes log-uniform in 200..3,000 (synth.scale_collection_torch) --, -D 0.05.
    One GPU: index build (the bucket sort in several passes over the hash space, species-wide lists through k_bucket_heavy),
    one self join, bytes resident.  Then the SHARDED run played on this one GPU for 2 / 4 / 8 shards: every shard builds the lists
    of its hash range, the tile records change hands (here: slices of the send buffers), every shard sorts what arrived and joins
    its rows.  Checked: the shards' hits add up to the single GPU's (count and a checksum over row, column and count), and the
    hits of sampled genomes equal a brute-force Jaccard over their species on the host."""
    from rabbitkssd_amd import capi, shard, synth
    torch, ctx = env.torch, env.ctx
    t0 = time.perf_counter()
    h, off, species = synth.scale_collection_torch(n_genomes, HASH_BITS, KMER, device=env.dev)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    H = int(h.numel())
    sk = ctx.sketches_from_dev(h.data_ptr(), off.data_ptr(), n_genomes)
    ts = []
    index = None
    for _ in range(3):   # (the first build fills the context's pool: hipMalloc; the median of the other two is reported)
        del index
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        index = ctx.index_build(sk, HASH_BITS)
        ts.append(time.perf_counter() - t0)
    build_ms = min(ts[1:]) * 1e3
    pool = ctx.pool_stats()
    hits_cap = 1 << 25
    hits = torch.empty(hits_cap * capi.HIT_DTYPE.itemsize, dtype=torch.uint8, device=env.dev)
    steps = 5
    counters = torch.zeros(counter_slots(steps, 1), dtype=torch.int64, device=env.dev)

    def launch(i):
        ctx.dist_rows_dev(index, 1, 0, KMER, MAX_DIST, hits.data_ptr(), hits_cap, counters.data_ptr() + 8 * i, stream=env.stream.cuda_stream)
    _, join_ms, _ = timed_steps(env, launch, steps, 1)
    n_hits = int(counters[steps].item())
    if n_hits > hits_cap:
        sys.exit("bench.py: scale leg: %d hits do not fit the buffer" % n_hits)

    def digest(buf, n):   # order-independent checksum of (row, col, common) on the device
        rec = buf[: n * capi.HIT_DTYPE.itemsize].view(torch.int32).view(-1, capi.HIT_DTYPE.itemsize // 4)
        r, c, m = rec[:, 0].to(torch.int64), rec[:, 1].to(torch.int64), rec[:, 2].to(torch.int64)
        return int(((r * 1000003 + c) * 31 + m).sum().item() & ((1 << 62) - 1)), rec

    want_digest, rec = digest(hits, n_hits)
    # ---- sampled genomes against a brute-force join over their species (host)
    sp_sizes = torch.bincount(species)
    order = torch.argsort(sp_sizes, descending=True)
    picks = [int(order[0]), int(order[len(order) // 200]), int(order[len(order) // 20]), int(order[len(order) // 3])]
    off_h = off.cpu().numpy()
    sampled, sample_ok = 0, True
    t_min = np.exp(-KMER * MAX_DIST)
    for sp in picks:
        members = torch.nonzero(species == sp).flatten().cpu().numpy()
        lo, hi = int(members[0]), int(members[-1]) + 1           # a species is a contiguous range of genomes
        sh = h[int(off_h[lo]):int(off_h[hi])].cpu().numpy().view(np.uint32)
        gid = np.repeat(np.arange(lo, hi), np.diff(off_h[lo:hi + 1]).astype(np.int64))
        sizes = np.diff(off_h[lo:hi + 1]).astype(np.int64)
        for g in members[:: max(1, len(members) // 3)][:3]:
            a = sh[int(off_h[g] - off_h[lo]):int(off_h[g + 1] - off_h[lo])]
            common = np.bincount(gid[np.isin(sh, a)] - lo, minlength=hi - lo)
            j = common / np.maximum(1, sizes[g - lo] + sizes - common)
            with np.errstate(divide="ignore"):
                d = np.where(j >= 1.0, 0.0, np.where(j <= 0.0, 1.0, -np.log(2.0 * j / (1.0 + j)) / KMER))
            mates = set((np.nonzero((d < MAX_DIST) & (np.arange(lo, hi) != g))[0] + lo).tolist())
            sel = rec[(rec[:, 0] == int(g)) | (rec[:, 1] == int(g))]
            got = set((sel[:, 0] + sel[:, 1] - int(g)).cpu().numpy().tolist())
            cm = {int(x[0] + x[1] - int(g)): int(x[2]) for x in sel.cpu().numpy()}
            sample_ok = sample_ok and got == mates and all(cm[x] == int(common[x - lo]) for x in mates)
            sampled += 1
    del rec
    ts1 = index.tile_stats(1, 0, KMER, MAX_DIST)
    single = {"index_build_ms": build_ms, "index_products": index.products, "index_built_fast": bool(index.built_fast), "join_ms": join_ms,
              "hits": n_hits, "kernel": ctx.dist_kernel_name(index, None, 1, 0, KMER, MAX_DIST), "e2e_ms": build_ms + join_ms,
              "pool_bytes": int(pool[0]), "tile_records": ts1[2], "tiles_with_records": ts1[0], "tiles_started": ts1[1],
              "pairs_per_s": n_genomes * (n_genomes - 1) / 2 / (join_ms * 1e-3),
              "hbm": hbm_block(8.0 * ts1[2] * (ts1[1] / max(1, ts1[0])) + 40.0 * n_hits, join_ms * 1e-3, env.peak_measured,
                               "the join's stream by construction: 8 B per record of the started tiles (estimated from their share of the tiles) + 40 B per hit")}
    del index
    # ---- the sharded run, played shard by shard on this GPU
    rehearsal = {}
    for S in (2, 4, 8):
        sends, counts, t_build, part0 = [], [], [], None
        for r in range(S):
            for rep in range(2):   # (the second run: the pool holds the blocks of this shard size)
                if rep:
                    del part, buf
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                part = ctx.index_build_shard(sk, HASH_BITS, r, S)
                cnt = part.shard_records(S)
                buf = torch.empty(max(1, sum(cnt) * shard.REC_BYTES), dtype=torch.uint8, device=env.dev)
                part.shard_pack(buf.data_ptr(), env.stream.cuda_stream)
                env.stream.synchronize()
                t_one = time.perf_counter() - t0
            t_build.append(t_one)
            sends.append(buf)
            counts.append(cnt)
            if part0 is None:
                part0 = part
            else:
                del part
        t_join, t_step, got_hits, got_digest, sent_bytes = [], [], 0, 0, []
        for d in range(S):
            chunks = [sends[r][shard.REC_BYTES * sum(counts[r][:d]): shard.REC_BYTES * sum(counts[r][:d + 1])] for r in range(S)]
            recv = torch.cat(chunks)
            n_recv = sum(counts[r][d] for r in range(S))
            for rep in range(2):
                if rep:
                    del join
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                join = ctx.index_join_shard(part0, recv.data_ptr(), n_recv)
                t_one = time.perf_counter() - t0
            t_join.append(t_one)
            cnts = torch.zeros(counter_slots(steps, 1), dtype=torch.int64, device=env.dev)

            def launch_d(i, join=join, cnts=cnts):
                ctx.dist_rows_dev(join, 1, 0, KMER, MAX_DIST, hits.data_ptr(), hits_cap, cnts.data_ptr() + 8 * i, stream=env.stream.cuda_stream)
            _, ms, _ = timed_steps(env, launch_d, steps, 1)
            t_step.append(ms)
            nh = int(cnts[steps].item())
            got_hits += nh
            got_digest = (got_digest + digest(hits, nh)[0]) & ((1 << 62) - 1)
            sent_bytes.append(shard.REC_BYTES * sum(counts[d][x] for x in range(S) if x != d))
            del join, recv
        # the exchange on real links: every rank sends (S - 1) / S of its records, one peer per link (up to seven)
        t_xchg = max(sent_bytes) / (XGMI_LINK_GBS * 1e9 * min(S - 1, 7)) * 1e3
        # (in front of it: ONE broadcast of the CSR sketches, 4 B per hash, ~100 GB/s out of a ring over 153 GB/s links)
        t_repl = 4.0 * H / 100e9 * 1e3
        e2e = max(t_build) * 1e3 + t_xchg + max(a * 1e3 + b for a, b in zip(t_join, t_step))
        rehearsal[str(S)] = {"shard_build_ms": [t * 1e3 for t in t_build], "join_build_ms": [t * 1e3 for t in t_join], "step_ms": t_step,
                             "records_sent_bytes_per_rank": sent_bytes, "exchange_ms_estimated": t_xchg,
                             "e2e_ms_predicted": e2e, "e2e_speedup_vs_one_gpu": single["e2e_ms"] / e2e,
                             "replicate_sketches_ms_estimated": t_repl, "e2e_incl_replicate_speedup_vs_one_gpu": single["e2e_ms"] / (e2e + t_repl),
                             "e2e_efficiency_predicted": single["e2e_ms"] / e2e / S,
                             "step_efficiency_predicted": join_ms / (S * max(t_step)),
                             "same_hits_as_one_gpu": bool(got_hits == n_hits and got_digest == want_digest)}
        if not rehearsal[str(S)]["same_hits_as_one_gpu"]:
            sys.exit("bench.py: scale leg: %d shards report %d hits, one GPU %d" % (S, got_hits, n_hits))
        del sends, part0
    del sk
    ctx.trim()
    return {"workload": "alldist over %d sketches generated on the device: species sizes Zipf(2.0) up to 10,000 strains, sketch sizes log-uniform "
                        "200..3,000 (one per species), species trees as in the 10,000-genome variants, 28-bit hashes, -D %g" % (n_genomes, MAX_DIST),
            "genomes": n_genomes, "hashes": H, "species": int(len(sp_sizes)), "largest_species": int(sp_sizes.max()), "generate_s": t_gen,
            "one_gpu": single, "sharded_rehearsal": rehearsal,
            "sampled_genomes_vs_brute_force": {"genomes": sampled, "same_pairs_and_counts": bool(sample_ok)},
            "note": "sharded_rehearsal: ONE GPU plays every shard in turn (build of its hash range incl. the records' packing, join build from "
                    "the records it would receive, its rows' join); e2e_ms_predicted = slowest shard build + the all-to-all of 12-byte tile "
                    "records on %d GB/s links (estimated: no second device here) + slowest (join build + step); the broadcast of the sketches "
                    "that precedes it is in multi_gpu.replicate_ms of an N > 1 run" % int(XGMI_LINK_GBS)}


def dist_rq_block(env, n_ref=100000, n_query=1000, steps=20, warmup=3, keep=None):
    """BASELINE configs[4] shape on one node: references 100,000 x 76 hashes, queries 1,000 x 45,776 hashes, 24-bit
    hashes (K10 S7 L4), -D 0.05.  Queries shard contiguously over the ranks; the index is broadcast once."""
    from rabbitkssd_amd import capi, synth
    torch, ctx, rank, world = env.torch, env.ctx, env.rank, env.world
    bits, kmer = 24, 20
    index = None
    qh = qoff = rh = roff = None
    if rank == 0:
        _, rh, roff = synth.clade_sketches(n_ref, 76, bits, seed=31)
        _, qh, qoff = synth.clade_sketches(n_query, 45776, bits, seed=32)
        index = ctx.index_build(ctx.sketches_from_host(rh, roff), bits)
    index, _, t_bcast, _ = env.share_index(index, rh, roff, bits)
    qh, qoff = env.share_sketches(qh, qoff)
    qs = ctx.sketches_from_host(qh, qoff)
    hits_cap = 1 << 20
    hits = torch.empty(hits_cap * capi.HIT_DTYPE.itemsize, dtype=torch.uint8, device=env.dev)
    counters = torch.zeros(counter_slots(steps, warmup), dtype=torch.int64, device=env.dev)
    per_rank = (n_query + world - 1) // world  # contiguous query blocks

    def launch(i):
        ctx.dist_rows_dev(index, 0, 0, kmer, MAX_DIST, hits.data_ptr(), hits_cap, counters.data_ptr() + 8 * i,
                          row_first=rank, row_step=world, row_block=per_rank, stream=env.stream.cuda_stream, queries=qs)

    elapsed, kernel_ms, spread = timed_steps(env, launch, steps, warmup)
    tot_hits = env.sum_over_ranks(int(counters[warmup + steps - 1].item()))
    # one GPU plays every contiguous query shard of a 2-, 4-, 8-GPU run in turn (src/dist.cpp:560: the rows -- queries -- are independent)
    rq_rehearsal = None
    if world == 1 and not env.args.no_rehearsal:
        rq_rehearsal = {}
        for S in (2, 4, 8):
            per = []
            blk = (n_query + S - 1) // S
            for r in range(S):
                c2 = torch.zeros(counter_slots(8, 1), dtype=torch.int64, device=env.dev)

                def launch_s(i, r=r, S=S, blk=blk, c2=c2):
                    ctx.dist_rows_dev(index, 0, 0, kmer, MAX_DIST, hits.data_ptr(), hits_cap, c2.data_ptr() + 8 * i, row_first=r, row_step=S,
                                      row_block=blk, stream=env.stream.cuda_stream, queries=qs)
                _, ms, _ = timed_steps(env, launch_s, 8, 1)
                per.append(ms)
            rq_rehearsal[str(S)] = {"shard_ms": per, "slowest_ms": max(per), "predicted_efficiency": kernel_ms / (S * max(per))}
    if rank != 0:
        return None
    if keep is not None:
        keep.update(rh=rh, roff=roff, qh=qh, qoff=qoff)
    # postings streamed: every query hash that is indexed walks its list once
    uh, cnt = np.unique(rh, return_counts=True)
    pos = np.searchsorted(uh, qh)
    pos[pos >= len(uh)] = 0
    T_all = int(cnt[pos][uh[pos] == qh].sum())
    my_q = min(per_rank, n_query)
    b_alg = (12.0 * len(qh) + 4.0 * T_all) * my_q / n_query + 4.0 * my_q * n_ref
    achieved = b_alg / (kernel_ms * 1e-3) / 1e9
    n_pairs = n_query * n_ref
    return {
        "workload": "dist: %d reference sketches x 76 hashes vs %d queries x 45,776 hashes, 24-bit hashes (K10 S7 L4), "
                    "-D %g; one fused kernel per rank, queries in %d contiguous block(s)" % (n_ref, n_query, MAX_DIST, world),
        "value": n_pairs * steps / elapsed, "unit": "genome-pairs/s", "ms_per_step": elapsed / steps * 1e3,
        "pairs": n_pairs, "query_hashes": int(len(qh)), "postings_streamed_T": T_all, "hits": int(tot_hits),
        "steps": steps, "warmup": warmup, "replicate_ms": t_bcast * 1e3, "scaling_rehearsal": rq_rehearsal,
        "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                     "contract_achieved": achieved, "contract_frac": achieved / HBM_PEAK_GBS,
                     "kernel": ctx.dist_kernel_name(index, qs, 0, 0, kmer, MAX_DIST), "kernel_ms": kernel_ms,
                     "kernel_ms_min_median_max": spread,
                     "contract_bytes_per_launch": b_alg,
                     "stream_bytes_per_launch": 4.0 * len(qh) * my_q / n_query,
                     "note": "achieved / frac: HBM bytes the counters saw (profiles/pmc_traffic_rq.json) per kernel time; "
                             "contract_*: SURVEY 8d's byte model (12 B/query hash + 4 B/posting + 4 B/count cell), which bills "
                             "the 4 B x %d count cells that never leave LDS; stream_bytes: the query hashes, the kernel's "
                             "only compulsory stream" % (my_q * n_ref)},
    }


def sketch_block(env, n_genomes, length, steps=5, pmc_file=None, cpu=True):
    """secondary metric: sketch k-mers/s, sequence bytes resident in HBM (BASELINE configs[1]: 1,000 x 5 Mb)"""
    from rabbitkssd_amd import capi, synth
    torch, ctx = env.torch, env.ctx
    params = capi.params_init(10, 6, 3)
    table = synth.shuf_table(10, 6, 3)  # the product's own `rabbit_kssd shuffle`
    flt = ctx.filter(params, table)
    stride = (length + 1023) // 1024 * 1024
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
    packed = torch.zeros(n_genomes * stride, dtype=torch.uint8, device="cuda")
    view = packed.view(n_genomes, stride)
    for i in range(n_genomes):
        view[i, :length] = lut[torch.randint(0, 4, (length,), generator=g, device="cuda")]
    gbeg = np.arange(n_genomes, dtype=np.uint64) * stride
    gend = gbeg + np.uint64(length)
    stream = env.stream.cuda_stream
    torch.cuda.synchronize()
    ctx.set_timing(True)
    sk = ctx.sketch_packed_dev(flt, packed.data_ptr(), packed.numel(), gbeg, gend, stream)  # warm-up
    windows = sk.windows
    kms = []
    t0 = time.time()
    for _ in range(steps):
        sk = ctx.sketch_packed_dev(flt, packed.data_ptr(), packed.numel(), gbeg, gend, stream)
        kms.append(ctx.last_ms(0))
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    ctx.set_timing(False)
    kernel_ms = sum(kms) / len(kms)
    # L3K10 variant of the scan kernel: the two-stage scan (default), or rk_sketch_kernel with its 64 KiB / 144 KiB LDS image
    sk_kernel = {"0": "rk_sketch_kernel<20, 8, true, 0>", "1": "rk_sketch_kernel<20, 8, false, 1>"}.get(
        os.environ.get("RK_SKETCH_IMG", "2"), "rk_scan2_kernel<20, 8>")
    b_alg = windows * 1.001  # 1 B per k-mer window + 4 B per emitted hash
    achieved = b_alg / (kernel_ms * 1e-3) / 1e9
    out = {"kmers_per_s": windows / dt, "genomes": n_genomes, "genome_length": length, "kmers": int(windows),
           "hashes": int(sk.total), "ms_per_pass": dt * 1e3,
           "pass": "scan kernel + per-genome LDS dedup + CSR placement, one upload and one read-back per batch; synthetic "
                   "uniform ACGT resident in HBM; 1.001 B/k-mer -> %.1f GB/s for the whole pass" % (b_alg / dt / 1e9),
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": sk_kernel,
                        "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": b_alg,
                        "limiter": "vector issue and LDS together: 146 vector instructions and 16 random LDS reads (8 cycles each with bank "
                                   "conflicts) per wave and 1,024 bases, 8 waves per SIMD; the waves are alive 75 % of the kernel's time "
                                   "(uneven pace of the XCDs in the second half of a pass; profiles/r03_pmc_summary.csv, DESIGN.md 4.1)"}}
    pmc = load_pmc(sk_kernel, pmc_file) if pmc_file else None
    roof = out["roofline"]
    roof["peak_measured"] = env.peak_measured
    roof["algorithmic"] = hbm_block(b_alg, kernel_ms * 1e-3, env.peak_measured, "SURVEY 8d: 1.001 B per k-mer window", working_set=b_alg)
    if pmc and pmc.get("hbm_bytes_per_launch"):
        roof["traffic"] = pmc["hbm_bytes_per_launch"]
        roof["hbm"] = hbm_block(pmc["hbm_bytes_per_launch"], kernel_ms * 1e-3, env.peak_measured,
                                "HBM counters (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, profiles/)", working_set=b_alg)
        roof["hbm_frac"] = roof["hbm"]["frac"]
        if pmc.get("SQ_ACTIVE_INST_VALU"):
            roof["issue_frac"] = pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / N_SIMD / ENGINE_CLOCK_HZ / (kernel_ms * 1e-3)
            # what binds the scan kernel is vector issue (+ its LDS probes): `bound` / `frac` say that, the byte model stays beside it
            roof.update({"bound": "valu_issue", "achieved": roof["issue_frac"] * N_SIMD * ENGINE_CLOCK_HZ, "peak": N_SIMD * ENGINE_CLOCK_HZ,
                         "unit": "SIMD issue cycles/s", "frac": roof["issue_frac"], "algorithmic_frac": achieved / HBM_PEAK_GBS})
    if cpu and not env.args.no_cpu_baseline:
        out["cpu_baseline"] = sketch_cpu_reference(packed, stride, length, n_genomes) or \
            sketch_cpu_baseline(packed, stride, length, n_genomes, table)
    del packed, view
    torch.cuda.empty_cache()
    return out


REF_SKETCH = os.path.join(ROOT, "oracle", "_ref", "ref_sketch_driver")


def sketch_big_block(env, length=1_000_000_000, sample=200_000_000):
    """BASELINE configs[4]'s query side: ONE 1 Gb genome at K10 S7 L4 (24-bit hashes), resident in HBM -- the genome is cut
    into chunks over the whole chip, its ~15,000 hashes are deduplicated per genome.  Beside it the reference's own
    sketchFastaFile arithmetic (src/sketch.cpp:487-530, small-file loop, one thread: a single file is one task there) on the
    first `sample` bases of the same genome."""
    from rabbitkssd_amd import capi, synth
    torch, ctx = env.torch, env.ctx
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as tmp:
        shuf = os.path.join(tmp, "L4K10.shuf")
        if subprocess.run([TOOL, "shuffle", "-k", "10", "-s", "7", "-l", "4", "-o", shuf], stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL).returncode != 0:
            return None
        table = np.fromfile(shuf, dtype=np.int32)[4:].copy()
        flt = ctx.filter(capi.params_init(10, 7, 4), table)
        g = torch.Generator(device="cuda")
        g.manual_seed(4321)
        lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
        pad = (length + 1023) // 1024 * 1024
        packed = torch.zeros(pad, dtype=torch.uint8, device="cuda")
        packed[:length] = lut[torch.randint(0, 4, (length,), generator=g, device="cuda")]
        gbeg, gend = np.array([0], dtype=np.uint64), np.array([length], dtype=np.uint64)
        stream = env.stream.cuda_stream
        torch.cuda.synchronize()
        ctx.set_timing(True)
        sk = ctx.sketch_packed_dev(flt, packed.data_ptr(), packed.numel(), gbeg, gend, stream)   # warm-up
        kms, t0 = [], time.time()
        for _ in range(3):
            sk = ctx.sketch_packed_dev(flt, packed.data_ptr(), packed.numel(), gbeg, gend, stream)
            kms.append(ctx.last_ms(0))
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 3
        ctx.set_timing(False)
        kernel_ms = sum(kms) / len(kms)
        b_alg = sk.windows * 1.001
        out = {"workload": "one %d-base genome, K10 S7 L4 (24-bit hashes), resident in HBM" % length, "kmers": int(sk.windows),
               "hashes": int(sk.total), "ms_per_pass": dt * 1e3, "kmers_per_s": sk.windows / dt,
               "roofline": {"bound": "hbm", "achieved": b_alg / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": b_alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": "rk_scan2_kernel (K10 S7 variant)",
                            "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": b_alg}}
        if not env.args.no_cpu_baseline and os.path.exists(REF_SKETCH):
            n = min(sample, length)
            fa = os.path.join(tmp, "big.fa")
            open(fa, "wb").write(synth.fasta_text("big", packed[:n].cpu().numpy()))
            open(os.path.join(tmp, "l"), "w").write(fa + "\n")
            t0 = time.time()
            p = subprocess.run([REF_SKETCH, "sketch", shuf, os.path.join(tmp, "l"), os.path.join(tmp, "o"), "1", "1"],
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            t_ref = time.time() - t0
            if p.returncode == 0:
                out["cpu_baseline"] = {"value": (n - 19) / t_ref, "unit": "k-mers/s", "cores": 1, "kind": "reference",
                                       "sample": "the first %d bases of the genome as one FASTA file, the reference's sketchFastaFile -t 1 (a "
                                                 "single file is one task of its small-file loop; its big-file branch needs RabbitFX), process "
                                                 "wall %.2f s incl. reading the 1 GiB .shuf" % (n, t_ref), "wall_s": t_ref}
        del packed
        torch.cuda.empty_cache()
        return out


def sketch_cpu_reference(packed, stride, length, n_genomes):
    """the reference's own sketchFastaFile (src/sketch.cpp:455-566, small-file path; oracle/_ref/ref_sketch_driver) on FASTA
    files of the same synthetic genomes, -t = host cores (capped at the file count: equal sizes keep every file below
    totalSize/numThreads, i.e. off the RabbitFX big-file branch), and -- on the very same files -- `rabbit_kssd sketch`"""
    from rabbitkssd_amd import synth
    if not os.path.exists(REF_SKETCH) or not os.path.exists(TOOL):
        return None
    from concurrent.futures import ThreadPoolExecutor
    cores = host_cores()
    sample = min(n_genomes, 1000 if cores >= 32 else 128)   # (all of configs[1] where the host has the cores for it)
    threads = max(1, min(cores, sample))
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as tmp:
        names = [os.path.join(tmp, "g%04d.fna" % g) for g in range(sample)]
        rows = packed.view(n_genomes, stride)

        def write(g):
            open(names[g], "wb").write(synth.fasta_text("g%04d" % g, rows[g, :length].cpu().numpy()))
        with ThreadPoolExecutor(max_workers=min(16, cores)) as ex:
            list(ex.map(write, range(sample)))
        open(os.path.join(tmp, "g.list"), "w").write("\n".join(names) + "\n")
        shuf = os.path.join(tmp, "L3K10.shuf")
        if subprocess.run([TOOL, "shuffle", "-k", "10", "-s", "6", "-l", "3", "-o", shuf], stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL).returncode != 0:
            return None
        t0 = time.time()
        p = subprocess.run([REF_SKETCH, "sketch", shuf, os.path.join(tmp, "g.list"), os.path.join(tmp, "ref_out"), str(threads), "1"],
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t_ref = time.time() - t0
        if p.returncode != 0:
            return None
        walls = []
        for _ in range(3):
            t0 = time.time()
            q = subprocess.run([TOOL, "sketch", "-i", os.path.join(tmp, "g.list"), "-L", shuf, "-o", os.path.join(tmp, "gpu_out"), "-q",
                                "-t", str(min(cores, 16))], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            walls.append(time.time() - t0)
            if q.returncode != 0:
                walls = []
                break
        kmers = sample * (length - 19)
        res = {"value": kmers / t_ref, "unit": "k-mers/s", "cores": threads, "kind": "reference",
               "sample": "%d synthetic 5 Mb genomes as FASTA files (page cache), the reference's sketchFastaFile -t %d, process wall %.3f s "
                         "(reads the 64 MiB .shuf + %d MB of FASTA, sketches, writes the .sketch)" % (sample, threads, t_ref, sample * length // 1000000),
               "wall_s": t_ref}
        if walls:
            med = sorted(walls)[len(walls) // 2]
            res["tool_wall_s"] = med
            res["tool_wall_runs_s"] = walls
            res["tool_vs_reference_wall"] = t_ref / med
            res["tool_note"] = "`rabbit_kssd sketch` on the same list and .shuf (process start, HIP init, parse, upload, kernels, .sketch), median of 3"
        return res


def sketch_cpu_baseline(packed, stride, length, n_genomes, table):
    """the oracle's restatement of src/sketch.cpp:487-530 under the reference's own parallelism: OpenMP
    `schedule(dynamic)` over genomes (:455-457), selected .shuf entries in a cache-resident membership structure
    (the role of `shuffled_map`, :338-345); a bounded sample of the same synthetic genomes, already in memory"""
    from oracle import oracle as ok
    cores = host_cores()
    sample = min(n_genomes, max(2 * cores, 16))
    param = ok.init_param(10, 6, 3)
    seq = packed.view(n_genomes, stride)[:sample, :length].contiguous().cpu().numpy().reshape(-1)
    goff = np.arange(sample + 1, dtype=np.uint64) * np.uint64(length)
    threads = min(cores, sample)
    ok.sketch_genomes_mt(param, table, seq[:200000], np.array([0, 200000], dtype=np.uint64), 1)  # load the library
    t0 = time.time()
    sizes = ok.sketch_genomes_mt(param, table, seq, goff, threads)
    dt = time.time() - t0
    kmers = sample * (length - 19)
    return {"value": kmers / dt, "unit": "k-mers/s", "cores": threads, "kind": "port",
            "sample": "%d of the %d synthetic 5 Mb genomes (in memory: no file parsing), OpenMP dynamic over genomes on %d "
                      "threads, %.3f s incl. building the selection bitmap; oracle/kssd_oracle.c restatement of the "
                      "reference loop, %d hashes" % (sample, n_genomes, threads, dt, int(sizes.sum()))}


def load_pmc(kernel, fname="pmc_traffic.json"):
    """counter record of profiles/ (HBM bytes per launch, vector-issue cycles), only if recorded for exactly this kernel variant"""
    path = os.path.join(ROOT, "profiles", fname)
    if not os.path.exists(path):
        return None
    try:
        d = json.load(open(path))
    except Exception:
        return None
    if d.get("kernel") != kernel:
        print("bench.py: %s was recorded for %r, the launched kernel is %r: traffic not reported (re-run "
              "tools/measure_round.sh)" % (fname, d.get("kernel"), kernel), file=sys.stderr)
        return None
    return d


ENGINE_CLOCK_HZ = 2.4e9   # MI355X peak engine clock (MI355X_MICROARCH.md)
N_SIMD = 1024             # 256 CUs x 4 SIMDs


INFINITY_CACHE_BYTES = 256 << 20   # MI355X_MICROARCH.md: 256 MB memory-side cache in front of HBM
ISSUE_PEAK = N_SIMD * ENGINE_CLOCK_HZ   # SIMD cycles per second: a SIMD issues one vector instruction per 4 cycles of a wave


def hbm_block(bytes_per_launch, secs, peak_measured, source, working_set=None):
    """the HBM side of a roofline block: GB/s moved against the nominal and the MEASURED peak"""
    gbs = bytes_per_launch / secs / 1e9
    blk = {"achieved": gbs, "peak": HBM_PEAK_GBS, "peak_measured": peak_measured, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
           "frac_of_measured": (gbs / peak_measured) if peak_measured else None, "bytes_per_launch": bytes_per_launch, "from": source}
    if working_set is not None:
        blk["working_set_bytes"] = working_set
        if working_set < INFINITY_CACHE_BYTES:
            blk["note"] = ("the launch's working set (%.0f MB) is re-read every step and fits the 256 MB Infinity Cache: the counters "
                           "count fabric requests, not DRAM accesses" % (working_set / 1e6))
    return blk


def apply_pmc(roof, pmc, peak_measured=None, issue_bound=False, working_set=None):
    """fills traffic (HBM bytes the counters saw), the `hbm` side block and issue_frac (cycles the SIMDs spent issuing vector
    instructions: SQ_ACTIVE_INST_VALU x 4 cycles / 1,024 SIMDs, over the kernel's duration) of a roofline block.  issue_bound:
    the kernel is bound by vector issue (tile / near-window / scan kernels): `bound`, `achieved`, `peak`, `frac` then say THAT --
    busy SIMD cycles per second against 1,024 SIMDs x 2.4 GHz -- and the HBM fraction stays beside it as hbm_frac."""
    if not pmc or not pmc.get("hbm_bytes_per_launch"):
        return
    secs = roof["kernel_ms"] * 1e-3
    roof["traffic"] = pmc["hbm_bytes_per_launch"]
    roof["hbm"] = hbm_block(pmc["hbm_bytes_per_launch"], secs, peak_measured,
                            "HBM counters (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, profiles/)", working_set)
    roof["hbm_frac"] = roof["hbm"]["frac"]
    if pmc.get("SQ_ACTIVE_INST_VALU"):
        roof["issue_frac"] = pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / ISSUE_PEAK / secs
    if issue_bound and roof.get("issue_frac"):
        roof.update({"bound": "valu_issue", "achieved": roof["issue_frac"] * ISSUE_PEAK, "peak": ISSUE_PEAK, "unit": "SIMD issue cycles/s",
                     "frac": roof["issue_frac"],
                     "achieved_from": "SQ_ACTIVE_INST_VALU x 4 cycles (profiles/) over the kernel's duration measured in this run"})
    else:
        roof.update({"bound": "hbm", "achieved": roof["hbm"]["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": roof["hbm"]["frac"],
                     "achieved_from": roof["hbm"]["from"]})
    roof["peak_measured"] = peak_measured


def cli_stamps(stderr_text):
    """the tool's own clock (RK_TIMING=1: `[timing]  12.345 ms  what`) as a table of phase durations"""
    marks = [(float(m.group(1)), m.group(2).strip()) for m in re.finditer(r"\[timing\]\s+([0-9.]+) ms\s+(.*)", stderr_text)]
    out, prev = {}, 0.0
    for t, what in marks:
        out[what] = t - prev
        prev = t
    return out


def alldist_cpu_and_cli(keep, n_pairs, bare_sketch=False):
    """the reference's index_tridist on the same sketches on this box's host cores, and -- on the very same files --
    the product's `rabbit_kssd alldist` command line (wall clock incl. process + HIP start-up)"""
    from oracle import oracle as ok
    from rabbitkssd_amd import synth
    cores = host_cores()
    names, hashes, off = keep["names"], keep["hashes"], keep["off"]
    n = len(off) - 1
    res, cli = None, None
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as tmp:
        sk = os.path.join(tmp, "bench.sketch")
        synth.write_sketch_file(sk, 10, 6, 3, names, hashes, off)
        if os.path.exists(REF):
            t0 = time.time()
            postings, counts = ok.index_build32(hashes, off, HASH_BITS)
            ok.write_index32(sk + ".dict", sk + ".index", postings, counts, HASH_BITS)
            del counts, postings
            t_files = time.time() - t0
            t0 = time.time()
            p = subprocess.run([REF, "alldist", tmp, sk, "ref.out", str(MAX_DIST), "0", str(cores)],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            wall = time.time() - t0
            err = p.stderr.decode(errors="replace")
            m_load = re.search(r"time of read index and offset sketch file is: ([0-9.eE+-]+)", err)
            m_dist = re.search(r"time of multiple threads distance computing and save the subFile is: ([0-9.eE+-]+)", err)
            if p.returncode == 0 and m_dist:
                t_dist = float(m_dist.group(1))
                lines = sum(1 for _ in open(os.path.join(tmp, "ref.out"))) - 1
                res = {"value": n_pairs / t_dist, "unit": "genome-pairs/s", "cores": cores, "kind": "reference",
                       "sample": "full workload: %d sketches, %d pairs, -D %g, -t %d; reference index_tridist (src/dist.cpp) "
                                 "distance-loop phase %.3f s; index load + prefix-sum phase %.3f s; process wall %.3f s; %d "
                                 "hits (.dict/.index written beforehand by the oracle in %.1f s: the reference's own "
                                 "transSketches needs 79 s)" % (n, n_pairs, MAX_DIST, cores, t_dist,
                                                                float(m_load.group(1)) if m_load else float("nan"), wall,
                                                                lines, t_files),
                       "wall_s": wall, "wall_pairs_per_s": n_pairs / wall, "hits": lines}
        # the product's command line on the same .sketch (the .dict/.index pair exists, so nothing is rewritten)
        if os.path.exists(TOOL):
            walls, stamps = [], []
            for _ in range(5):
                t0 = time.time()
                p = subprocess.run([TOOL, "alldist", "-i", sk, "-D", str(MAX_DIST), "-o", "gpu.out", "-t", str(min(cores, 16))],
                                   cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, RK_TIMING="1"))
                walls.append(time.time() - t0)
                stamps.append(cli_stamps(p.stderr.decode(errors="replace")))
                if p.returncode != 0:
                    walls = []
                    break
            if walls:
                lines = sum(1 for _ in open(os.path.join(tmp, "gpu.out"))) - 1
                mid = sorted(range(len(walls)), key=lambda i: walls[i])[len(walls) // 2]
                med = walls[mid]
                cli = {"cli_wall_ms": med * 1e3, "cli_wall_runs_ms": [w * 1e3 for w in walls], "cli_hits": lines,
                       "cli_breakdown_ms": stamps[mid],
                       "cli_breakdown_note": "the tool's own clock (RK_TIMING=1) in the median run, each phase since the one before: "
                                             "`context ready` is hipInit + context (the .sketch is read meanwhile), `index built` upload + "
                                             "rk_index_build incl. its code objects, `distances on the host` rk_dist_rows (the tile kernel, "
                                             "hit download, host-side ordering and libm), `text written` the output file; the wall clock "
                                             "adds process start and exit",
                       "cli_note": "`rabbit_kssd alldist -i bench.sketch -D %g` end to end (process start, HIP init, read .sketch, "
                                   "index build, distances, text output), median of 5" % MAX_DIST}
                if res:
                    cli["cli_vs_reference_wall"] = res["wall_s"] / med
                    cli["cli_same_hits_as_reference"] = lines == res["hits"]
                    # ... and against the reference started from the same bare .sketch: it first writes .dict/.index itself
                    # (transSketches, src/subCommand.cpp:165-169 -> src/sketch.cpp:894-1021)
                    t_trans, src = 79.0, "BASELINE.md [probe]: 79 s in the 8-vCPU build container (not timed in this run: --bare-sketch)"
                    if bare_sketch and os.path.exists(REF_SKETCH):
                        t0 = time.time()
                        p = subprocess.run([REF_SKETCH, "resave", sk, os.path.join(tmp, "bare.sketch")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                        if p.returncode == 0:
                            t_trans, src = time.time() - t0, "timed in this run: the reference's readSketches + saveSketches + transSketches (oracle/_ref/ref_sketch_driver resave)"
                    cli["cli_vs_reference_wall_bare_sketch"] = {"ratio": (t_trans + res["wall_s"]) / med, "reference_transSketches_s": t_trans,
                                                                "reference_alldist_wall_s": res["wall_s"], "source": src}
    if res is None:
        # port: the C restatement (same dense index, per-thread counter row, OpenMP dynamic rows)
        t0 = time.time()
        postings, counts = ok.index_build32(hashes, off, HASH_BITS)
        t_build = time.time() - t0
        sizes = np.diff(off).astype(np.uint32)
        t0 = time.time()
        hits, _ = ok.index_dist32(counts, HASH_BITS, postings, sizes, hashes, off, 1, 0, KMER, MAX_DIST, threads=cores)
        t = time.time() - t0
        res = {"value": n_pairs / t, "unit": "genome-pairs/s", "cores": cores, "kind": "port",
               "sample": "full workload: %d sketches, %d pairs, -D %g, %d threads; oracle port incl. 2^28 prefix sum %.3f s "
                         "(index build %.3f s not counted); %d hits" % (n, n_pairs, MAX_DIST, cores, t, t_build, len(hits)),
               "hits": int(len(hits))}
    return res, cli


def reference_alldist_pairs(names, hashes, off):
    """the reference's index_tridist (oracle/_ref/ref_driver) on these sketches: (distance-loop seconds, process wall seconds,
    sorted (lo, hi, common) of every reported pair) -- or None when the reference build is not there"""
    from oracle import oracle as ok
    from rabbitkssd_amd import synth
    if not os.path.exists(REF):
        return None
    cores = host_cores()
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as tmp:
        sk = os.path.join(tmp, "v.sketch")
        synth.write_sketch_file(sk, 10, 6, 3, names, hashes, off)
        postings, counts = ok.index_build32(hashes, off, HASH_BITS)
        ok.write_index32(sk + ".dict", sk + ".index", postings, counts, HASH_BITS)
        del counts, postings
        t0 = time.time()
        p = subprocess.run([REF, "alldist", tmp, sk, "ref.out", str(MAX_DIST), "0", str(cores)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        wall = time.time() - t0
        m = re.search(r"time of multiple threads distance computing and save the subFile is: ([0-9.eE+-]+)", p.stderr.decode(errors="replace"))
        if p.returncode != 0 or not m:
            return None
        where = {n: i for i, n in enumerate(names)}
        lo, hi, cm = [], [], []
        with open(os.path.join(tmp, "ref.out")) as f:
            next(f)
            for line in f:
                a, b, c = line.split("\t")[:3]
                i, j = where[a], where[b]
                lo.append(min(i, j))
                hi.append(max(i, j))
                cm.append(int(c.split("|")[0]))
        lo, hi, cm = np.array(lo, dtype=np.int64), np.array(hi, dtype=np.int64), np.array(cm, dtype=np.int64)
        key = np.lexsort((cm, hi, lo))
        return float(m.group(1)), wall, np.stack([lo[key], hi[key], cm[key]])


def alldist_variant(env, head, n_genomes, steps, strains, tiny, what, skew=0):
    """the headline workload on a collection that does NOT look like the clade-of-ten generator: species of 100 / 1,000
    strains, a 40-hash sketch among the bacteria.  Kernel time, pairs per second against the headline's, and whether the
    reported pairs and counts are the reference's."""
    k = {}
    b = alldist_block(env, n_genomes, steps, 3, k, strains=strains, tiny=tiny, build_reps=5, skew=skew)
    if env.rank != 0:
        return None
    res = {"workload": what, "genomes": b["genomes"], "pairs": b["pairs"], "hits": b["hits"], "kernel": b["kernel"],
           "kernel_ms": b["kernel_ms"], "kernel_ms_min_median_max": b["kernel_ms_min_median_max"],
           "pairs_per_s": b["pairs"] / (b["kernel_ms"] * 1e-3),
           "pairs_per_s_vs_headline": (b["pairs"] / (b["kernel_ms"] * 1e-3)) / (head["pairs"] / (head["kernel_ms"] * 1e-3)),
           "index_build_ms": b["index_build_ms"], "first_call_ms": b.get("first_call_ms"), "build_plus_dist_ms": b.get("build_plus_dist_ms"),
           "index_products": b.get("index_products")}
    stats = k["index"].self_stats
    if b["kernel"].startswith("rk_tile_kernel"):
        # the tile kernel's stream: the 8-byte tile records (one per posting list and pair of 32-genome blocks) + 40 B per hit
        stream = 8.0 * stats[3] + 40.0 * b["hits"]
        secs = b["kernel_ms"] * 1e-3
        roof = {"bound": "hbm", "achieved": stream / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": stream / secs / 1e9 / HBM_PEAK_GBS,
                "traffic": None, "achieved_from": "the kernel's own stream: 8 B per tile record + 40 B per hit written",
                "stream_bytes_per_launch": stream, "tile_records": int(stats[3]), "kernel": b["kernel"], "kernel_ms": b["kernel_ms"],
                "issue_frac": None,
                "limited_by": "vector issue (bit-sliced adds: ~4 vector instructions per tile record) and the length of a tile's chain of "
                              "records per wave; HBM traffic is a few per cent of the roof by construction (a record stands for up to "
                              "1,024 cell increments)"}
        apply_pmc(roof, load_pmc(b["kernel"], "pmc_traffic_tile_clade%d.json" % strains) if strains > 10 else None, env.peak_measured,
                  issue_bound=True, working_set=stream)
        roof["effective"] = True
        roof["tile_stats"] = b.get("tile_stats")
        res["roofline"] = roof
    if not env.args.no_cpu_baseline and env.world == 1:
        ref = reference_alldist_pairs(k["names"], k["hashes"], k["off"])
        if ref:
            res["reference"] = {"distance_loop_s": ref[0], "wall_s": ref[1], "hits": int(ref[2].shape[1]), "cores": host_cores()}
            res["same_hits_as_reference"] = bool(b["pairs_canonical"] is not None and np.array_equal(b["pairs_canonical"], ref[2]))
            if not res["same_hits_as_reference"]:
                print(json.dumps(res), flush=True)
                sys.exit("bench.py: %s: the GPU's pairs differ from the reference's" % what)
    k.clear()
    return res


def use64_block(env, n_genomes=10000, steps=10):
    """a K12 L3 collection (36-bit hashes: the 64-bit layout, half_k - drlevel > 8): index build and alldist kernel"""
    from rabbitkssd_amd import capi, synth
    torch, ctx = env.torch, env.ctx
    names, hashes, off = synth.clade_sketches(n_genomes, HASHES_PER_GENOME, 36, kmer_size=24, seed=20261003, wide=True)
    sk = ctx.sketches_from_host64(hashes, off)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        index = ctx.index_build(sk, 36)
        ts.append(time.perf_counter() - t0)
    hits_cap = 1 << 20
    hits = torch.empty(hits_cap * capi.HIT_DTYPE.itemsize, dtype=torch.uint8, device=env.dev)
    counters = torch.zeros(counter_slots(steps, 2), dtype=torch.int64, device=env.dev)

    def launch(i):
        ctx.dist_rows_dev(index, 1, 0, 24, MAX_DIST, hits.data_ptr(), hits_cap, counters.data_ptr() + 8 * i, stream=env.stream.cuda_stream)
    _, kernel_ms, _ = timed_steps(env, launch, steps, 2)
    n_pairs = n_genomes * (n_genomes - 1) // 2
    return {"workload": "alldist over %d synthetic sketches with 36-bit hashes (K12 S6 L3: the 64-bit hash layout), -D %g" % (n_genomes, MAX_DIST),
            "index_build_ms": sorted(ts[1:])[1] * 1e3, "index_built_fast": bool(index.built_fast), "kernel": ctx.dist_kernel_name(index, None, 1, 0, 24, MAX_DIST),
            "kernel_ms": kernel_ms, "pairs_per_s": n_pairs / (kernel_ms * 1e-3), "hits": int(counters[2 + steps - 1].item())}


def dist_rq_cpu_baseline(keep, n_pairs):
    """the reference's index_dist (src/dist.cpp:429-776) on a bounded sample of the same queries"""
    from oracle import oracle as ok
    from rabbitkssd_amd import synth
    if not os.path.exists(REF):
        return None
    cores = host_cores()
    rh, roff, qh, qoff = keep["rh"], keep["roff"], keep["qh"], keep["qoff"]
    n_ref, n_query = len(roff) - 1, len(qoff) - 1
    sample = n_query if cores >= 32 else min(n_query, 100)
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as tmp:
        rs, qs = os.path.join(tmp, "ref.sketch"), os.path.join(tmp, "qry.sketch")
        synth.write_sketch_file(rs, 10, 7, 4, ["r%d" % i for i in range(n_ref)], rh, roff)
        synth.write_sketch_file(qs, 10, 7, 4, ["q%d" % i for i in range(sample)], qh[:int(qoff[sample])], qoff[:sample + 1])
        postings, counts = ok.index_build32(rh, roff, 24)
        ok.write_index32(rs + ".dict", rs + ".index", postings, counts, 24)
        t0 = time.time()
        p = subprocess.run([REF, "dist", tmp, rs, qs, "rq.out", str(MAX_DIST), "1", "0", "0", str(cores)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        wall = time.time() - t0
        m = re.search(r"time of multiple threads distance computing and save the subFile is: ([0-9.eE+-]+)",
                      p.stderr.decode(errors="replace"))
        if p.returncode != 0 or not m:
            return None
        t_dist = float(m.group(1))
        return {"value": sample * n_ref / t_dist, "unit": "genome-pairs/s", "cores": cores, "kind": "reference",
                "sample": "%d of the %d queries against all %d references, -D %g, -t %d: reference index_dist (src/dist.cpp) "
                          "distance-loop phase %.3f s, process wall %.3f s" % (sample, n_query, n_ref, MAX_DIST, cores, t_dist, wall),
                "wall_pairs_per_s": sample * n_ref / wall}


def multi_gpu_block(block, world):
    """what a user of N GPUs waits for: replication of the index, the slowest rank's step, the gather of the hits"""
    if world == 1:
        return None
    return {"replicate_mode": block["replicate_mode"], "replicate_bytes_per_rank": block["replicate_bytes"],
            "replicate_ms": block["replicate_ms"], "step_ms": block["ms_per_step"], "gather_ms": block["gather_ms"],
            "e2e_ms": block["e2e_ms"],
            "note": "e2e = replicate + slowest rank's step + gather of the hit records on rank 0, each barrier to barrier; `value` "
                    "times the steps alone (the index is resident, BASELINE configs[2]).  replicate: one RCCL broadcast of the CSR "
                    "sketches + rk_index_build on every rank (--replicate blob: rank 0's packed index instead)"}


def dist_roofline(block, pmc_file=None, peak_measured=None):
    """roofline block of a self-join launch.  What binds these kernels is vector issue, not HBM: with a counter record of this
    very kernel variant in profiles/ the block says `bound: valu_issue` and `frac` = the share of the launch the SIMDs spent
    issuing vector instructions; the HBM side (bytes the counters saw per second, against the nominal AND the measured peak)
    stays beside it as `hbm` / `hbm_frac`.  Without a record: the kernel's own stream against HBM.  `contract_*`: SURVEY 8d's
    byte model, which bills count cells that are never formed and postings that are never streamed (it exceeds 1: continuity)."""
    secs = block["kernel_ms"] * 1e-3
    stream = block["stream_bytes_per_launch"]
    tile = block["kernel"].startswith("rk_tile_kernel")
    roof = {"bound": "hbm", "achieved": stream / secs / 1e9, "peak": HBM_PEAK_GBS, "peak_measured": peak_measured, "unit": "GB/s",
            "frac": stream / secs / 1e9 / HBM_PEAK_GBS, "traffic": None,
            "achieved_from": "the kernel's own stream: 8 B per %s record + 40 B per hit written (no counter record for this variant)" % ("tile" if tile else "slice"),
            "stream_bytes_per_launch": stream,
            "contract_bytes_per_launch": block["contract_bytes_per_launch"],
            "contract_achieved": block["contract_bytes_per_launch"] / secs / 1e9,
            "contract_frac": block["contract_bytes_per_launch"] / secs / 1e9 / HBM_PEAK_GBS,
            "issue_frac": None,
            "kernel": block["kernel"], "kernel_ms": block["kernel_ms"],
            "kernel_ms_min_median_max": block["kernel_ms_min_median_max"],
            "kernel_ms_cold": block["kernel_ms_cold"]}
    if tile:
        roof["tile_records"] = block.get("tile_records")
        roof["effective"] = True
        roof["effective_note"] = ("pairs/s is an EFFECTIVE rate: the kernel starts only the tiles that can hold a reportable pair under "
                                  "-D (tile_stats: tiles_started of tiles_with_records; cells_formed_share of the pair matrix) and never "
                                  "streams T postings -- same hits as the reference, which forms every cell (src/dist.cpp:194-255); "
                                  "threshold_legs shows the rate under -D 0.3 and -D 1.0")
        roof["tile_stats"] = block.get("tile_stats")
        roof["cold_note"] = ("the tile records come with rk_index_build (collections of 4,000 genomes and more): the FIRST self join over an "
                             "index runs on this kernel, which keeps no per-launch state -- kernel_ms, kernel_ms_cold and first_call_ms "
                             "(a fresh index, host clock around one call) are the same launch")
        roof["limited_by"] = ("vector issue (bit-sliced adds: ~2.6 vector instructions per tile record) and, at 10,000 genomes, the length "
                              "of a tile's chain of records per wave (the launch is one round of ~560 workgroups); HBM traffic is a few "
                              "per cent of the roof by construction (a record stands for up to 1,024 cell increments), DESIGN.md 4.3c")
        if pmc_file:
            pmc_file = {"pmc_traffic.json": "pmc_traffic_tile_clade10.json", "pmc_traffic_50k.json": "pmc_traffic_tile_50k.json"}.get(pmc_file, pmc_file)
    else:
        roof["cold_note"] = ("kernel_ms: launches that repeat one (index, options) pair (after the first completed one the empty fallback "
                             "launch of rk_near_kernel is skipped); kernel_ms_cold: every launch a first one (two option sets taking turns)")
        roof["limited_by"] = ("the dependent memory round trips of a unit (row bounds -> slice records -> sizes and ids of the "
                              "reportable cells) with one short unit per wave, and vector issue; not HBM bandwidth, DESIGN.md 4.3")
    if pmc_file:
        apply_pmc(roof, load_pmc(block["kernel"], pmc_file), peak_measured, issue_bound=True, working_set=stream)
    return roof


def build_roofline(block, peak_measured=None):
    """rk_index_build as its own roofline block: sketches in HBM -> index in HBM.  Algorithmic bytes: the hashes read once
    (4 B), the postings written (4 B), the distinct hashes + posting offsets (8 B each), and the join structure the build
    emits: one 8-byte slice record per (genome, hash) with later sharers, or -- from 4,000 genomes on -- 16 bytes per tile
    record (AoS + the split copy) + 64 bytes of directory per tile."""
    ts = block.get("tile_stats") or {}
    join_bytes = 16.0 * ts.get("tile_records", 0) + 64.0 * ts.get("tiles_with_records", 0) if (block.get("index_products", 0) & 4) else 8.0 * block["slice_records"]
    b = 8.0 * block["hashes"] + 8.0 * block.get("distinct", 0) + join_bytes
    secs = block["index_build_ms"] * 1e-3
    return {"bound": "hbm", "achieved": b / secs / 1e9, "peak": HBM_PEAK_GBS, "peak_measured": peak_measured, "unit": "GB/s",
            "frac": b / secs / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_build": b, "build_ms": block["index_build_ms"],
            "products": "tile records (rk_index_tiles.inc)" if (block.get("index_products", 0) & 4) else "slice records",
            "path": "bucket sort, second level in LDS (rk_index_fast.inc)" if block["index_built_fast"] else "device-wide radix sort",
            "limited_by": "a chain of ~20 dependent kernels none of which is bandwidth-bound: every workgroup of the in-LDS bucket sort is a "
                          "chain of four memory round trips and eleven barriers (130 us at 10,000 genomes, eight rounds of workgroups per CU), "
                          "the two-pass partition 120 us, the tile sort 100 us (profiles/r05_index_build_kernels.txt)"}


def main():
    args = parse_args()
    env = Env(args)
    from rabbitkssd_amd import shard
    rank, world, ctx = env.rank, env.world, env.ctx

    n_genomes = args.genomes if args.scaling == "strong" else shard.weak_scaling_genomes(args.genomes, world)
    keep = {}
    head = alldist_block(env, n_genomes, args.steps, args.warmup, keep, build_reps=(15 if world == 1 else 0))

    # the same collection listed in other orders (random permutation of the ids; completion-order jitter): the index
    # renumbers the genomes internally, so kernel variant, time and hits must not depend on the order
    orders = {}
    rehearsal = None
    if world == 1 and not args.no_orders:
        for mode in ("shuffled", "jitter"):
            o = alldist_block(env, n_genomes, max(20, args.steps // 4), 5, None, order_mode=mode, build_reps=8)
            orders[mode] = {
                "ms_per_step": o["ms_per_step"], "kernel": o["kernel"], "kernel_ms": o["kernel_ms"], "hits": o["hits"],
                "compact_share": o["compact_share"], "index_build_ms": o["index_build_ms"], "build_plus_dist_ms": o["build_plus_dist_ms"],
                "vs_sorted": o["kernel_ms"] / head["kernel_ms"],
                "same_pairs_and_counts_as_sorted": bool(o["pairs_canonical"] is not None and head["pairs_canonical"] is not None and
                                                        np.array_equal(o["pairs_canonical"], head["pairs_canonical"]))}
    # collections that do not look like the generator of the headline: wide species, a tiny sketch
    variants = None
    if world == 1 and not args.no_variants:
        vs = max(10, args.steps // 2)
        variants = {
            "clade100": alldist_variant(env, head, n_genomes, vs, 100, 0, "alldist over %d sketches in species of 100 strains (10 sub-lineages "
                                        "of 10; every pair of a species within -D %g)" % (n_genomes, MAX_DIST)),
            "clade1000": alldist_variant(env, head, n_genomes, vs, 1000, 0, "alldist over %d sketches in species of 1,000 strains (10 lineages "
                                         "of 100: pairs across lineages share ~28 %% of their hashes and are NOT within -D %g)" % (n_genomes, MAX_DIST)),
            "tiny": alldist_variant(env, head, n_genomes, vs, 10, 1, "the headline collection plus one 40-hash sketch (a plasmid)"),
            "canonical_skew": alldist_variant(env, head, n_genomes, vs, 10, 0, "the headline collection with its hashes spread as a real sketcher's are: "
                                              "the quarters of the hash space filled 7 : 5 : 3 : 1 (the leading base of a canonical k-mer; the bucket "
                                              "sort's buckets are equal ranges of the hash space, DESIGN.md 8)", skew=1),
            "canonical_skew2": alldist_variant(env, head, n_genomes, vs, 10, 0, "the same with the 7 : 5 : 3 : 1 once more inside every quarter (fullest "
                                               "bucket 3.06x the mean: beyond the 2.67x the LDS sort holds -- k_bucket_heavy)", skew=2),
        }
    sharded_head = sharded_block(env, keep.get("hashes"), keep.get("off"), HASH_BITS) if world > 1 else None
    legs = None
    if world == 1 and not args.no_variants:
        legs = threshold_legs(env, keep["index"], head["pairs"])
    if world == 1 and not args.no_rehearsal:
        rehearsal = {"10000": shard_rehearsal(env, keep["index"], n_genomes)}
    distinct = int(keep["index"].distinct) if rank == 0 else 0

    t_host_inclusive, n_host_hits = 0.0, 0
    if rank == 0:
        # PCIe-inclusive reference point (never `value`): host sketches -> upload -> index build -> distance kernel ->
        # hits back on the host, steady state (second pass)
        for _ in range(2):
            t0 = time.time()
            sk2 = ctx.sketches_from_host(keep["hashes"], keep["off"])
            idx2 = ctx.index_build(sk2, HASH_BITS)
            host_hits, _ = ctx.dist_rows(idx2, None, 1, 0, KMER, MAX_DIST)
            t_host_inclusive = time.time() - t0
            n_host_hits = len(host_hits)
            del sk2, idx2
    keep.pop("index", None)
    keep.pop("sk", None)

    config3 = None
    if not args.no_config3:
        k3 = {}
        config3 = alldist_block(env, args.config3_genomes, max(10, args.steps // 10), 3, k3, build_reps=(5 if world == 1 else 0))
        sharded3 = sharded_block(env, k3.get("hashes"), k3.get("off"), HASH_BITS, steps=10) if world > 1 else None
        if rank == 0 and sharded3 is not None:
            config3["sharded"] = sharded3
        if rank == 0:
            config3["distinct"] = int(k3["index"].distinct)
            if world == 1 and not args.no_rehearsal:
                rehearsal[str(args.config3_genomes)] = shard_rehearsal(env, k3["index"], args.config3_genomes, steps=10)
        k3.clear()
    rq_keep = {}
    rq = None
    if not args.no_dist_rq:
        rq = dist_rq_block(env, keep=rq_keep)

    if rank != 0:
        if world > 1:
            env.dist.destroy_process_group()
        return

    head["distinct"] = distinct
    n_pairs = head["pairs"]
    u16 = head["kernel"].startswith("rk_dist_kernel<true")
    near = head["kernel"].startswith("rk_near_kernel")
    tile = head["kernel"].startswith("rk_tile_kernel")
    out = {
        "metric": "genome-pairs/sec alldist (10k bacteria, L3K10)",
        "hbm_peak": {"nominal_gbs": HBM_PEAK_GBS, "measured_gbs": env.peak_measured,
                     "how": "k_calib_read: a streaming read of a 2 GiB buffer (8 x the Infinity Cache), 16 B per lane, HIP events around 5 "
                            "launches, at the start of this run (SURVEY 8d's measured denominator)"},
        "value": head["value"],
        "unit": "genome-pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"],
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": ("bit-sliced counters of 32 x 32 tiles in registers, 16 planes (exact), u32 counts after extraction" if tile else
                  "u32 window counts in registers, accumulated bit-sliced per lane (exact)" if near else
                  "u16 intersection counters in LDS (exact: a count is bounded by the sketch size, < 65536)" if u16 else
                  "u32 intersection counters in LDS") + " / f64 jaccard + distance",
        "data": "synthetic",
        "config": {"workload": "alldist over %d synthetic bacterial sketches (sketch-level clade generator, seed 20261003, "
                               "L3K10: 28-bit hashes, ~%d per genome, clades of 10), -D %g, distance kernel only from an "
                               "HBM-resident index (BASELINE configs[2]: precomputed .sketch/.dict)" % (n_genomes, HASHES_PER_GENOME, MAX_DIST),
                   "genomes": n_genomes, "pairs": n_pairs, "hashes": head["hashes"],
                   "postings_streamed_T": head["postings_streamed_T"], "hits": head["hits"], "max_dist": MAX_DIST,
                   "slice_records": head["slice_records"], "compact_share": head["compact_share"],
                   "records_walked_per_launch": head["records_walked"],
                   "sharding": "query rows in blocks of %d dealt round-robin to %d rank(s); sketches broadcast once (RCCL on GPUs), index built per rank" % (shard.ROW_BLOCK, world),
                   "first_call_ms": head["first_call_ms"], "tile_records": head.get("tile_records"),
                   "build_plus_dist_ms": head["build_plus_dist_ms"],
                   "build_plus_dist_pairs_per_s": (n_pairs / (head["build_plus_dist_ms"] * 1e-3)) if head["build_plus_dist_ms"] else None,
                   "index_build_ms": head["index_build_ms"], "index_products": head.get("index_products"),
                   "effective": bool(tile), "tile_stats": head.get("tile_stats"), "threshold_legs": legs,
                   "step": ("one rk_dist_rows_dev call over the resident index: rk_tile_kernel counts 32 x 32 tiles of the pair matrix from "
                            "one 8-byte record per posting list and pair of 32-genome blocks and evaluates their cells.  The tile records "
                            "are a product of rk_index_build (collections of 4,000 genomes and more): the FIRST join over an index "
                            "(first_call_ms, the command-line tool, the reference-side binding, build_plus_dist_ms = build + one join "
                            "from resident sketches) runs this very kernel; the choice of kernel follows the index's size and shape, "
                            "never how often it was joined.  `value` is an effective rate (roofline.effective_note)") if tile else
                           ("one rk_dist_rows_dev call: rk_near_kernel counts and evaluates every pair of the launch; its exact "
                            "fallback pass (rows whose far cells could be reportable) is launched until a completed launch with "
                            "the same options has shown that list to be empty -- here after the warm-up -- and skipped from then on "
                            "(RK_DIST_FB_SKIP=0 launches it always: +3 us per step)")},
        "roofline": dist_roofline(head, "pmc_traffic.json" if world == 1 else None, env.peak_measured),
        # second headline: what an alldist costs when the index is NOT there yet -- sketches resident in HBM -> hits in HBM
        "build_plus_dist": {
            "ms": head["build_plus_dist_ms"], "pairs_per_s": (n_pairs / (head["build_plus_dist_ms"] * 1e-3)) if head["build_plus_dist_ms"] else None,
            "note": "rk_index_build + rk_dist_rows_dev from device-resident sketches, median of 15, one synchronisation (the "
                    "build's 32-byte read-back); `value` above times the distance kernel alone, as BASELINE configs[2] words it",
            "index_build": build_roofline(head, env.peak_measured)},
        "multi_gpu": multi_gpu_block(head, world),
        "multi_gpu_sharded": sharded_head,
        "alldist_order": orders or None,
        "alldist_variants": variants,
        "scaling_rehearsal": rehearsal,
        "setup": {"index_build_ms": head["index_build_ms"], "index_build_cold_ms": head["index_build_cold_ms"],
                  "index_built_fast": head["index_built_fast"],
                  "index_blob_bytes": head["index_blob_bytes"],
                  "host_inclusive_ms": t_host_inclusive * 1e3,
                  "host_inclusive_note": "host sketches -> H2D -> rk_index_build -> rk_dist_rows -> %d hits on the host "
                                         "(PCIe-inclusive, whole dataset, one pass, steady state; not `value`)" % n_host_hits},
    }
    if config3:
        out["config3"] = {
            "workload": "alldist over %d synthetic bacterial sketches (BASELINE configs[3]), -D %g, strong scaling: the same "
                        "dataset at every N" % (config3["genomes"], MAX_DIST),
            "value": config3["value"], "unit": "genome-pairs/s", "ms_per_step": config3["ms_per_step"], "scaling": "strong",
            "pairs": config3["pairs"], "hits": config3["hits"], "steps": config3["steps"], "warmup": config3["warmup"],
            "index_build_ms": config3["index_build_ms"], "index_built_fast": config3["index_built_fast"],
            "build_plus_dist_ms": config3["build_plus_dist_ms"], "compact_share": config3["compact_share"],
            "index_blob_bytes": config3["index_blob_bytes"], "multi_gpu": multi_gpu_block(config3, world), "multi_gpu_sharded": config3.get("sharded"),
            "roofline": dist_roofline(config3, "pmc_traffic_50k.json" if world == 1 else None, env.peak_measured),
            "index_build": build_roofline(config3, env.peak_measured)}
    if rq:
        if world == 1:
            apply_pmc(rq["roofline"], load_pmc(rq["roofline"]["kernel"], "pmc_traffic_rq.json"), env.peak_measured,
                      working_set=rq["roofline"]["stream_bytes_per_launch"])
            if rq["roofline"]["achieved"] is None:   # no counter record for this variant: the compulsory stream
                secs = rq["roofline"]["kernel_ms"] * 1e-3
                rq["roofline"]["achieved"] = rq["roofline"]["stream_bytes_per_launch"] / secs / 1e9
                rq["roofline"]["frac"] = rq["roofline"]["achieved"] / HBM_PEAK_GBS
                rq["roofline"]["achieved_from"] = "the query hashes (4 B each), the kernel's compulsory stream"
            if not args.no_cpu_baseline:
                cb = dist_rq_cpu_baseline(rq_keep, rq["pairs"])
                if cb:
                    rq["cpu_baseline"] = cb
        out["dist_rq"] = rq
    if world == 1 and not args.no_variants:
        out["use64"] = use64_block(env)
    if world == 1 and not args.no_scale:
        out["scale"] = scale_block(env, args.scale_genomes)
    if world == 1 and not args.no_sketch:
        out["sketch"] = sketch_block(env, args.sketch_genomes, args.sketch_length,
                                     pmc_file="pmc_traffic_sketch%d.json" % args.sketch_genomes)
        if args.sketch_genomes != 128:   # round 1-3's batch, kept as a second figure (kernel and pass only)
            out["sketch_128"] = sketch_block(env, 128, args.sketch_length, pmc_file="pmc_traffic_sketch.json", cpu=False)
        if not args.no_sketch_big:
            out["sketch_big"] = sketch_big_block(env)
    if world == 1 and not args.no_cpu_baseline:
        cb, cli = alldist_cpu_and_cli(keep, n_pairs, args.bare_sketch)
        out["cpu_baseline"] = cb
        if cb.get("hits") is not None and cb["hits"] != head["hits"]:
            print(json.dumps(out), flush=True)
            sys.exit("bench.py: the GPU reports %d hits, the CPU baseline %d" % (head["hits"], cb["hits"]))
        if cb.get("wall_pairs_per_s"):
            cb["gpu_kernel_vs_cpu_distance_loop"] = out["value"] / cb["value"]
            cb["gpu_api_vs_cpu_wall"] = (n_pairs / t_host_inclusive) / cb["wall_pairs_per_s"]
        if cli:
            out["setup"].update(cli)
    print(json.dumps(out), flush=True)
    if world > 1:
        env.dist.destroy_process_group()


if __name__ == "__main__":
    main()
