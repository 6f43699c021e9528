#!/usr/bin/env python3
"""Tiny drivers for rocprofv3 runs (kernel-trace or --pmc): one leg only, few launches.
    python3 tools/prof_driver.py sketch [n_genomes] [length]
    python3 tools/prof_driver.py dist [n_genomes] [steps]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from rabbitkssd_amd import capi, synth  # noqa: E402


def sketch(n_genomes=128, length=5_000_000, steps=3):
    ctx = capi.Context(0)
    flt = ctx.filter(capi.params_init(10, 6, 3), synth.shuf_table(10, 6, 3))
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
    torch.cuda.synchronize()
    ctx.set_timing(True)
    for _ in range(steps):
        t0 = time.time()
        sk = ctx.sketch_packed_dev(flt, packed.data_ptr(), packed.numel(), gbeg, gend, 0)
        torch.cuda.synchronize()
        print("sketch pass %.3f ms (scan kernel %.4f ms), %d windows, %d hashes" % ((time.time() - t0) * 1e3, ctx.last_ms(0), sk.windows, sk.total))


def dist(n_genomes=10000, steps=5, row_step=1, row_block=0, order=0, clade=10, tiny=0, metric=0):
    """order: 0 as generated (clade members adjacent), 1 random permutation of the genome ids, 2 completion-order jitter;
    clade: strains per clade (> 10: the species tree of synth.strain_rates); tiny: extra 40-hash sketches"""
    ctx = capi.Context(0)
    names, hashes, off = synth.clade_sketches(n_genomes, 1220, 28, strains_per_clade=clade, tiny=tiny)
    n_genomes = len(names)
    if order:
        names, hashes, off = synth.permute_genomes(names, hashes, off, synth.genome_order(n_genomes, ["sorted", "shuffled", "jitter"][order]))
    index = ctx.index_build(ctx.sketches_from_host(hashes, off), 28)
    hits = torch.empty((1 << 23) * capi.HIT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    counters = torch.zeros(steps + 1, dtype=torch.int64, device="cuda")
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        def launch(i):
            ctx.dist_rows_dev(index, 1, metric, 20, 0.05, hits.data_ptr(), 1 << 23, counters.data_ptr() + 8 * i,
                              row_first=0, row_step=row_step, row_block=row_block, stream=stream.cuda_stream)
        launch(steps)  # warm-up: the first join over the index (near-window kernel) ...
        launch(steps)  # ... and the second (a resident index moves to the tile kernel: its records are built here)
        torch.cuda.synchronize()
        counters.zero_()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.time()
        ev0.record(stream)
        for i in range(steps):
            launch(i)
        ev1.record(stream)
        t_host = time.time() - t0   # time the host needed to enqueue everything
        torch.cuda.synchronize()
    print("dist %.4f ms/step (events; host enqueue %.3f ms/step), hits %d (row_step %d, row_block %d, order %d, clade %d, tiny %d, metric %d) %s" % (
        ev0.elapsed_time(ev1) / steps, t_host * 1e3 / steps, int(counters[0].item()), row_step, row_block, order, clade, tiny, metric,
        ctx.dist_kernel_name(index, None, 1, metric, 20, 0.05, row_first=0, row_step=row_step, row_block=row_block)))


def dist_rq(n_ref=100000, n_query=1000, steps=3):
    """configs[4] shape: ref-vs-query, 24-bit hashes (K10 S7 L4), queries of 45,776 hashes"""
    ctx = capi.Context(0)
    rn, rh, roff = synth.clade_sketches(n_ref, 76, 24, seed=31)
    qn, qh, qoff = synth.clade_sketches(n_query, 45776, 24, seed=32)
    t0 = time.time()
    index = ctx.index_build(ctx.sketches_from_host(rh, roff), 24)
    qs = ctx.sketches_from_host(qh, qoff)
    torch.cuda.synchronize()
    print("index build + uploads %.1f ms (H=%d, U=%d)" % ((time.time() - t0) * 1e3, index.total, index.distinct))
    for _ in range(steps):
        t0 = time.time()
        hits, _ = ctx.dist_rows(index, qs, 0, 0, 20, 0.05)
        dt = time.time() - t0
        print("dist ref-vs-query %.3f ms, %d x %d = %.3g pairs -> %.3g pairs/s, %d hits"
              % (dt * 1e3, n_query, n_ref, n_query * n_ref, n_query * n_ref / dt, len(hits)))


def dist_rq_dev(n_ref=100000, n_query=1000, steps=5, m_ref=76, m_query=45776, bits=24):
    """configs[4] shape by default, kernel only (rk_dist_rows_dev with explicit queries), HIP-event time per launch"""
    ctx = capi.Context(0)
    rn, rh, roff = synth.clade_sketches(n_ref, m_ref, bits, seed=31)
    qn, qh, qoff = synth.clade_sketches(n_query, m_query, bits, seed=32 if m_query != m_ref or n_query != n_ref else 31)
    index = ctx.index_build(ctx.sketches_from_host(rh, roff), bits)
    qs = ctx.sketches_from_host(qh, qoff)
    hits = torch.empty((1 << 20) * capi.HIT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    counters = torch.zeros(steps + 1, dtype=torch.int64, device="cuda")
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        def launch(i):
            ctx.dist_rows_dev(index, 0, 0, 20, 0.05, hits.data_ptr(), 1 << 20, counters.data_ptr() + 8 * i,
                              stream=stream.cuda_stream, queries=qs)
        launch(steps)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        for i in range(steps):
            launch(i)
        ev1.record(stream)
        torch.cuda.synchronize()
    print("dist_rq kernel %.3f ms/launch (events), %d x %d, hits %d" % (ev0.elapsed_time(ev1) / steps, n_query, n_ref,
                                                                         int(counters[0].item())))


def index_only(n_genomes=10000, reps=5, clade=10):
    """rk_index_build alone (profiler runs, developer ablations that leave the index unusable)"""
    ctx = capi.Context(0)
    names, hashes, off = synth.clade_sketches(n_genomes, 1220, 28, strains_per_clade=clade)
    sk = ctx.sketches_from_host(hashes, off)
    for r in range(reps):
        torch.cuda.synchronize()
        t0 = time.time()
        idx = ctx.index_build(sk, 28)
        dt = time.time() - t0
        print("index build %d: %.3f ms (H=%d U=%d fast=%d products=%d)" % (r, dt * 1e3, idx.total, idx.distinct, idx.built_fast, idx.products))
        del idx


def index(n_genomes=10000, reps=5):
    """rk_index_build wall time, first call (cold pool) and steady state"""
    ctx = capi.Context(0)
    names, hashes, off = synth.clade_sketches(n_genomes, 1220, 28)
    sk = ctx.sketches_from_host(hashes, off)
    for r in range(reps):
        torch.cuda.synchronize()
        t0 = time.time()
        idx = ctx.index_build(sk, 28)
        dt = time.time() - t0
        print("index build %d: %.3f ms (H=%d U=%d fast=%d)" % (r, dt * 1e3, idx.total, idx.distinct, idx.built_fast))
        del idx
    for r in range(3):
        t0 = time.time()
        sk2 = ctx.sketches_from_host(hashes, off)
        t1 = time.time()
        idx = ctx.index_build(sk2, 28)
        t2 = time.time()
        h, _ = ctx.dist_rows(idx, None, 1, 0, 20, 0.05)
        t3 = time.time()
        print("host-inclusive %d: upload %.3f + build %.3f + dist_rows %.3f = %.3f ms (%d hits)" % (
            r, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3, len(h)))
        del idx, sk2


def calib(mbytes=1024, width=16):
    """a streaming read of `mbytes` MiB at `width` bytes per lane (k_calib_read): FETCH_SIZE of this launch against the
    known byte count calibrates the counter for that access width (run under rocprofv3 --pmc FETCH_SIZE)"""
    import ctypes as C
    ctx = capi.Context(0)
    L = capi.lib()
    L.rk_debug_calib_read.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
    for _ in range(2):
        rc = L.rk_debug_calib_read(ctx._h, C.c_uint64(mbytes << 20), int(width))
        assert rc == 0
    print("calib: read %d bytes at %d B/lane" % (mbytes << 20, width))


if __name__ == "__main__":
    which = sys.argv[1]
    args = [int(x) for x in sys.argv[2:]]
    {"sketch": sketch, "dist": dist, "dist_rq": dist_rq, "dist_rq_dev": dist_rq_dev, "index": index, "index_only": index_only, "calib": calib}[which](*args)
