#!/usr/bin/env python3
"""bench.py -- headline benchmark: genome-pairs/s of `alldist` on 10,000 synthetic bacterial
sketches (L3K10, ~1,220 hashes each, clades of 10 strains), distance kernel only, inputs
resident in HBM (BASELINE.json metric; config "10,000 synthetic 5 Mb bacteria, L3K10,
alldist from precomputed .sketch/.dict").

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (intersection counting through the inverted index +
Jaccard->Mash epilogue + hit compaction, ONE kernel launch per rank) over this rank's
query rows.  Multi-GPU: query rows are interleaved over the ranks (row r -> rank r mod N),
the reference index is built on rank 0 and sent to every peer with ONE RCCL broadcast
(outside the timed region); there is no data-path collective and no reduction.  Scaling
is weak: the dataset has round(10000*sqrt(N)) genomes so that the number of pairs per GPU
stays that of the 1-GPU workload (--scaling strong keeps 10,000 genomes instead).

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the
distance kernel and, at N=1, `cpu_baseline` (the reference's own dist.cpp when
oracle/_ref/ref_driver was built, else the oracle port) timed on the host cores.
"""
import argparse
import ctypes
import json
import math
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


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--genomes", type=int, default=10000)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to "
                    "rehearse the N>1 plumbing with all ranks on ONE GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--no-sketch", action="store_true")
    ap.add_argument("--sketch-genomes", type=int, default=128)
    ap.add_argument("--sketch-length", type=int, default=5_000_000)
    return ap.parse_args()


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(names, hashes, off, n_pairs):
    """The reference CPU alldist on the same sketches, on this box's host cores."""
    from oracle import oracle as ok
    cores = host_cores()
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    n = len(off) - 1
    t0 = time.time()
    postings, counts = ok.index_build32(hashes, off, HASH_BITS)
    t_build = time.time() - t0
    if os.path.exists(ref):
        with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as tmp:
            sk = os.path.join(tmp, "bench.sketch")
            ok.save_sketches32(sk, 10, 6, 3, names, hashes, off)
            ok.write_index32(sk + ".dict", sk + ".index", postings, counts, HASH_BITS)
            del counts
            t0 = time.time()
            p = subprocess.run([ref, "alldist", tmp, sk, "bench.out", str(MAX_DIST), "0", str(cores)],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            wall = time.time() - t0
            err = p.stderr.decode(errors="replace")
            m_load = re.search(r"time of read index and offset sketch file is: ([0-9.eE+-]+)", err)
            m_dist = re.search(r"time of multiple threads distance computing and save the subFile is: ([0-9.eE+-]+)", err)
            if p.returncode == 0 and m_dist:
                t_dist = float(m_dist.group(1))
                lines = sum(1 for _ in open(os.path.join(tmp, "bench.out"))) - 1
                return {"value": n_pairs / t_dist, "unit": "genome-pairs/s", "cores": cores,
                        "kind": "reference",
                        "sample": "full workload: %d sketches, %d pairs, -D %g, -t %d; reference "
                                  "index_tridist (src/dist.cpp) distance-loop phase %.3f s; index load+"
                                  "prefix-sum phase %.3f s; process wall %.3f s; %d hits"
                                  % (n, n_pairs, MAX_DIST, cores, t_dist,
                                     float(m_load.group(1)) if m_load else float("nan"), wall, lines),
                        "wall_pairs_per_s": n_pairs / wall}
    # port: the C restatement (same dense index, per-thread counter row, OpenMP dynamic rows)
    sizes = np.diff(off).astype(np.uint32)
    t0 = time.time()
    hits, _ = ok.index_dist32(counts, HASH_BITS, postings, sizes, hashes, off, 1, 0, KMER, MAX_DIST,
                              threads=cores)
    t = time.time() - t0
    return {"value": n_pairs / t, "unit": "genome-pairs/s", "cores": cores, "kind": "port",
            "sample": "full workload: %d sketches, %d pairs, -D %g, %d threads; oracle port incl. "
                      "2^28 prefix sum %.3f s (index build %.3f s not counted); %d hits"
                      % (n, n_pairs, MAX_DIST, cores, t, t_build, len(hits))}


def sketch_leg(ctx, capi, torch, n_genomes, length, steps=3):
    """secondary metric: sketch k-mers/s, sequence bytes resident in HBM."""
    from rabbitkssd_amd import synth
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
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    sk = ctx.sketch_packed_dev(flt, packed.data_ptr(), packed.numel(), gbeg, gend, stream)  # warm-up
    windows = sk.windows
    t0 = time.time()
    for _ in range(steps):
        sk = ctx.sketch_packed_dev(flt, packed.data_ptr(), packed.numel(), gbeg, gend, stream)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    return {"kmers_per_s": windows / dt, "genomes": n_genomes, "genome_length": length,
            "kmers": int(windows), "hashes": int(sk.total), "ms_per_pass": dt * 1e3,
            "note": "scan kernel + device sort/unique dedup, synthetic uniform ACGT resident in HBM; "
                    "1.001 B/k-mer -> %.1f GB/s" % (windows * 1.001 / dt / 1e9)}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from rabbitkssd_amd import capi, shard, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world),
                  file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the engine has no CPU fallback")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    ctx = capi.Context(local_rank)
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream().cuda_stream

    n_genomes = args.genomes if args.scaling == "strong" else shard.weak_scaling_genomes(args.genomes, world)
    n_pairs = n_genomes * (n_genomes - 1) // 2

    # ---- setup (untimed): rank 0 generates the sketches and builds the index on its GPU; the
    # index travels to the peers as one blob in one RCCL broadcast
    names = hashes = off = None
    if rank == 0:
        names, hashes, off = synth.clade_sketches(n_genomes, HASHES_PER_GENOME, HASH_BITS, kmer_size=KMER)
        sk = ctx.sketches_from_host(hashes, off)
        t0 = time.time()
        index = ctx.index_build(sk, HASH_BITS)
        torch.cuda.synchronize()
        t_index_build = time.time() - t0
        nbytes = index.blob_bytes
        # PCIe-inclusive reference point (never `value`): host sketches -> upload -> index build ->
        # distance kernel -> hits back on the host
        t0 = time.time()
        sk2 = ctx.sketches_from_host(hashes, off)
        idx2 = ctx.index_build(sk2, HASH_BITS)
        host_hits, _ = ctx.dist_rows(idx2, None, 1, 0, KMER, MAX_DIST)
        t_host_inclusive = time.time() - t0
        del sk2, idx2
    t_bcast = 0.0
    if world > 1:
        blob = None
        if rank == 0:
            blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            index.pack_dev(blob.data_ptr(), nbytes, stream)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.time()
        blob = shard.broadcast_blob(blob, 0, dev, dist)   # ONE RCCL broadcast of the whole index
        torch.cuda.synchronize()
        t_bcast = time.time() - t0
        nbytes = blob.numel()
        if rank != 0:
            index = ctx.index_unpack_dev(blob.data_ptr(), nbytes, stream)
        del blob
    H, T = index.total, index.sum_sq

    hits_cap = 1 << 20
    hits = torch.empty(hits_cap * capi.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    counters = torch.zeros(args.steps + args.warmup + 1, dtype=torch.int64, device=dev)

    def step(i):
        ctx.dist_rows_dev(index, 1, 0, KMER, MAX_DIST, hits.data_ptr(), hits_cap,
                          counters.data_ptr() + 8 * i, row_first=rank, row_step=world, stream=stream,
                          row_block=shard.ROW_BLOCK)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step(args.warmup + i)
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    n_hits_rank = int(counters[args.warmup].item())
    tot_hits = torch.tensor([n_hits_rank], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(tot_hits)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- rank 0: report
    # algorithmic bytes of one launch on this rank: 12 B per query hash (hash +
    # two index offsets) + 4 B per posting streamed (T = sum c_h^2) + 4 B per count cell
    # produced.  Rank 0 holds 1/world of the rows (interleaved -> ~1/world of each term).
    b_alg = (12.0 * H + 4.0 * T) / world + 4.0 * shard.rank_pairs(n_genomes, 0, world)
    achieved = b_alg / (kernel_ms * 1e-3) / 1e9
    out = {
        "metric": "genome-pairs/sec alldist (10k bacteria, L3K10)",
        "value": n_pairs * args.steps / elapsed,
        "unit": "genome-pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u32 counts / f64 distance",
        "data": "synthetic",
        "config": {"workload": "alldist over %d synthetic bacterial sketches (sketch-level clade generator, "
                               "seed 20261003, L3K10: 28-bit hashes, ~%d per genome, clades of 10), -D %g, "
                               "distance kernel only from an HBM-resident index" % (n_genomes, HASHES_PER_GENOME, MAX_DIST),
                   "genomes": n_genomes, "pairs": n_pairs, "hashes": int(H), "postings_streamed_T": int(T),
                   "hits": int(tot_hits.item()), "max_dist": MAX_DIST,
                   "sharding": "query rows in blocks of 16 dealt round-robin to %d rank(s); index broadcast once (RCCL on GPUs)" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "rk_dist_kernel", "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_launch": b_alg},
        "setup": {"index_build_ms": t_index_build * 1e3, "index_blob_bytes": int(nbytes),
                  "rccl_broadcast_ms": t_bcast * 1e3,
                  "host_inclusive_ms": t_host_inclusive * 1e3,
                  "host_inclusive_note": "host sketches -> H2D -> rk_index_build -> rk_dist_rows -> %d hits on the "
                                         "host (PCIe-inclusive, whole dataset, one pass; not `value`)" % len(host_hits)},
    }
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc) and world == 1:
        try:
            out["roofline"]["traffic"] = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except Exception:
            pass
    if world == 1 and not args.no_sketch:
        out["sketch"] = sketch_leg(ctx, capi, torch, args.sketch_genomes, args.sketch_length)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(names, hashes, off, n_pairs)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
