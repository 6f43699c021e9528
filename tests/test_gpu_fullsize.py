"""Full-size checks at BASELINE.json's configurations (GPU): exact agreement with the oracle where
it finishes in seconds on the box's host cores, size-independent properties otherwise
(45 hits per complete 10-strain clade at -D 0.05, union of row shards == whole, blob round trip)."""
import os

import numpy as np
import pytest

from oracle import oracle as ok
from rabbitkssd_amd import capi, synth

pytestmark = pytest.mark.gpu
CORES = max(1, len(os.sched_getaffinity(0)))


@pytest.fixture(scope="module")
def ctx():
    return capi.Context(0)


def check_hits(mine, want):
    assert len(mine) == len(want)
    for f in ("row", "col", "common", "size0", "size1"):
        assert np.array_equal(mine[f], want[f]), f
    assert np.array_equal(mine["jorc"], want["jorc"])
    assert np.max(np.abs(mine["dist"] - want["dist"]), initial=0.0) <= 1e-12


def test_config2_alldist_10k_exact(ctx):
    """configs[2]: 10,000 sketches, L3K10, alldist -D 0.05 -- every hit identical to the oracle."""
    import torch
    names, h, off = synth.clade_sketches(10000, 1220, 28)
    idx = ctx.index_build(ctx.sketches_from_host(h, off), 28)
    assert (idx.total, idx.distinct, idx.sum_sq) == (12199994, 3153593, 90120058)
    mine, _ = ctx.dist_rows(idx, None, 1, 0, 20, 0.05)
    postings, counts = ok.index_build32(h, off, 28)
    assert np.array_equal(idx.export(want_counts=False)[0], postings)
    want, _ = ok.index_dist32(counts, 28, postings, np.diff(off).astype(np.uint32), h, off, 1, 0, 20, 0.05,
                              threads=CORES)
    assert len(want) == 45000          # 45 pairs per 10-strain clade
    check_hits(mine, want)
    # containment metric and a looser threshold on the same index
    want, _ = ok.index_dist32(counts, 28, postings, np.diff(off).astype(np.uint32), h, off, 1, 1, 20, 0.2,
                              threads=CORES)
    mine, _ = ctx.dist_rows(idx, None, 1, 1, 20, 0.2)
    check_hits(mine, want)
    # the single-blob form used for the RCCL broadcast reproduces the index
    blob = torch.empty(idx.blob_bytes, dtype=torch.uint8, device="cuda")
    idx.pack_dev(blob.data_ptr(), blob.numel())
    idx2 = ctx.index_unpack_dev(blob.clone().data_ptr(), blob.numel())
    assert (idx2.total, idx2.distinct, idx2.sum_sq, idx2.genomes) == (idx.total, idx.distinct, idx.sum_sq, 10000)
    parts = [ctx.dist_rows(idx2, None, 1, 0, 20, 0.05, row_first=r, row_step=8)[0] for r in range(8)]
    merged = np.concatenate(parts)
    merged = merged[np.lexsort((merged["col"], merged["row"]))]
    full, _ = ctx.dist_rows(idx, None, 1, 0, 20, 0.05)
    assert merged.tobytes() == full.tobytes()
    # device-resident asynchronous entry point used by bench.py
    hits = torch.empty((1 << 17) * capi.HIT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    ctx.dist_rows_dev(idx, 1, 0, 20, 0.05, hits.data_ptr(), 1 << 17, cnt.data_ptr(),
                      stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    n = int(cnt.item())
    dev = np.frombuffer(hits.cpu().numpy().tobytes()[: n * capi.HIT_DTYPE.itemsize], dtype=capi.HIT_DTYPE)
    dev = dev[np.lexsort((dev["col"], dev["row"]))]
    assert dev.tobytes() == full.tobytes()


def test_config3_alldist_50k_properties_and_exact(ctx):
    """configs[3] (one rank's view): 50,000 sketches -> 1.25e9 pairs, 225,000 hits."""
    names, h, off = synth.clade_sketches(50000, 1220, 28)
    idx = ctx.index_build(ctx.sketches_from_host(h, off), 28)
    mine, _ = ctx.dist_rows(idx, None, 1, 0, 20, 0.05)
    assert len(mine) == 45 * 5000
    assert np.all(mine["row"] // 10 == mine["col"] // 10) and np.all(mine["col"] > mine["row"])
    sizes = np.diff(off).astype(np.int64)
    assert np.array_equal(mine["size0"], sizes[mine["row"]]) and np.array_equal(mine["size1"], sizes[mine["col"]])
    # exact check of the counts against set intersections on a sample of hits
    rng = np.random.default_rng(0)
    for t in rng.integers(0, len(mine), size=200):
        i, j = int(mine["row"][t]), int(mine["col"][t])
        a = h[int(off[i]):int(off[i + 1])]
        b = h[int(off[j]):int(off[j + 1])]
        assert mine["common"][t] == len(np.intersect1d(a, b, assume_unique=True))
        jac, d = ok.distance(mine["common"][t], len(a), len(b), 0, 20)
        assert mine["jorc"][t] == jac and abs(mine["dist"][t] - d) <= 1e-12
    # rows of one rank out of 8 (what a GPU of configs[3] computes) are a subset with the same records
    part, _ = ctx.dist_rows(idx, None, 1, 0, 20, 0.05, row_first=3, row_step=8)
    sel = mine[mine["row"] % 8 == 3]
    assert part.tobytes() == sel.tobytes()
    # the block-cyclic shard the multi-GPU callers use (blocks of 16 rows)
    part, _ = ctx.dist_rows(idx, None, 1, 0, 20, 0.05, row_first=5, row_step=8, row_block=16)
    sel = mine[(mine["row"] // 16) % 8 == 5]
    assert part.tobytes() == sel.tobytes()


def test_dense_output_ordered_on_the_device(ctx):
    """-D 1.5 reports every pair: 2,000 genomes -> 1,999,000 hits, more than the 2^20 above which the
    result is ordered on the device; order and content equal the oracle's"""
    names, h, off = synth.clade_sketches(2000, 200, 24, seed=5)
    idx = ctx.index_build(ctx.sketches_from_host(h, off), 24)
    postings, counts = ok.index_build32(h, off, 24)
    want, _ = ok.index_dist32(counts, 24, postings, np.diff(off).astype(np.uint32), h, off, 1, 0, 20, 1.5, threads=CORES)
    mine, _ = ctx.dist_rows(idx, None, 1, 0, 20, 1.5)
    assert len(mine) == 2000 * 1999 // 2
    check_hits(mine, want)


def test_self_join_with_tiled_columns_90k(ctx):
    """more genomes than one LDS row holds (> ~78,000 columns): the self join runs tile by tile with
    range-checked postings; exact vs the oracle"""
    names, h, off = synth.clade_sketches(90000, 12, 22, seed=77)
    idx = ctx.index_build(ctx.sketches_from_host(h, off), 22)
    postings, counts = ok.index_build32(h, off, 22)
    sizes = np.diff(off).astype(np.uint32)
    want, _ = ok.index_dist32(counts, 22, postings, sizes, h, off, 1, 0, 20, 0.05, threads=CORES)
    assert len(want) > 100000
    mine, _ = ctx.dist_rows(idx, None, 1, 0, 20, 0.05)
    check_hits(mine, want)
    part, _ = ctx.dist_rows(idx, None, 1, 0, 20, 0.05, row_first=1, row_step=3, row_block=16)
    sel = mine[(mine["row"] // 16) % 3 == 1]
    assert part.tobytes() == sel.tobytes()


def test_config4_ref_vs_query_100k(ctx):
    """configs[4] shape at reduced query count: 100,000 reference sketches (76 hashes, 24-bit, L4K10S7)
    against queries of 45,776 hashes (3 Gb genomes): tiled LDS row, resolved ranges, exact vs oracle."""
    rn, rh, roff = synth.clade_sketches(100000, 76, 24, seed=31)
    qn, qh, qoff = synth.clade_sketches(24, 45776, 24, seed=32)
    # plant reference sketches inside the queries (a mammal-sized query "contains" some bacteria)
    parts = []
    for q in range(24):
        own = qh[int(qoff[q]):int(qoff[q + 1])]
        planted = [rh[int(roff[r]):int(roff[r + 1])] for r in range(q * 10, q * 10 + 6)]
        parts.append(np.unique(np.concatenate([own] + planted)))
    qoff = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
    qh = np.concatenate(parts)
    idx = ctx.index_build(ctx.sketches_from_host(rh, roff), 24)
    qs = ctx.sketches_from_host(qh, qoff)
    postings, counts = ok.index_build32(rh, roff, 24)
    sizes = np.diff(roff).astype(np.uint32)
    for metric, D in ((1, 0.05), (0, 0.5)):
        want, _ = ok.index_dist32(counts, 24, postings, sizes, qh, qoff, 0, metric, 20, D, threads=CORES)
        mine, _ = ctx.dist_rows(idx, qs, 0, metric, 20, D)
        assert len(want) >= 24 * 6
        check_hits(mine, want)


def test_config0_fasta_to_alldist_64x5mb(ctx):
    """configs[0] on the GPU: 64 synthetic 5 Mb genomes, L3K10, sketch -> index -> alldist -D 0.05."""
    k, s, l = 10, 6, 3
    param, table = ok.init_param(k, s, l), ok.shuffle_table(k, s, l)
    flt = ctx.filter(capi.params_init(k, s, l), table)
    n, length = 64, 5_000_000
    genomes = synth.clade_genome_set(n, length)
    seq = np.concatenate([b for _, b in genomes])
    rec_off = np.arange(n + 1, dtype=np.uint64) * length
    sk = ctx.sketch_batch(flt, seq, rec_off, np.arange(n + 1, dtype=np.uint64))
    gh, goff = sk.download()
    assert sk.windows == n * (length - 2 * k + 1)
    for g in (0, 1, 9, 10, 37, 63):   # sampled genomes: identical hash sets
        want = ok.sketch_records(param, table, genomes[g][1], np.array([0, length], dtype=np.uint64))
        assert np.array_equal(gh[int(goff[g]):int(goff[g + 1])].astype(np.uint64), want), g
    sizes = np.diff(goff)
    assert 1100 < sizes.min() and sizes.max() < 1350      # ~1,233 hashes per 5 Mb genome at L3K10
    idx = ctx.index_build(sk, 28)
    mine, _ = ctx.dist_rows(idx, None, 1, 0, 2 * k, 0.05)
    postings, counts = ok.index_build32(gh, goff, 28)
    want, _ = ok.index_dist32(counts, 28, postings, sizes.astype(np.uint32), gh, goff, 1, 0, 2 * k, 0.05,
                              threads=CORES)
    assert len(want) == 6 * 45 + 6      # six complete clades + the 4-strain remainder
    check_hits(mine, want)
