"""GPU parity tests: the HIP path, called through the C ABI, against the oracle and the
golden fixtures.  Integer results (hash sets, postings, intersection counts, sizes) must be
bit-exact; FP64 jaccard/containment exact; Mash/AafD distance bit-identical through the synchronous API (host libm has
the last word), within 1e-12 (north_star) where the hits stay on the device."""
import json
import os

import numpy as np
import torch
import pytest

from conftest import GOLDEN
from oracle import oracle as ok
from rabbitkssd_amd import capi, synth

pytestmark = pytest.mark.gpu
DIST_TOL = 0.0  # rk_dist_rows recomputes the distances of its hits with the host libm: bit-identical to the oracle (the
                # device-resident API rk_dist_rows_dev stays within 1e-12, north_star's tolerance: tests/test_gpu_fullsize.py)


@pytest.fixture(scope="module")
def ctx():
    return capi.Context(0)


def assert_hits_equal(mine, want):
    assert len(mine) == len(want)
    for f in ("row", "col", "common", "size0", "size1"):
        assert np.array_equal(mine[f], want[f]), f
    assert np.array_equal(mine["jorc"], want["jorc"])          # one IEEE division
    assert np.max(np.abs(mine["dist"] - want["dist"]), initial=0.0) <= DIST_TOL


def load_dist_case():
    d = os.path.join(GOLDEN, "dist")
    man = json.load(open(os.path.join(d, "manifest.json")))
    _, rnames, rh, roff = ok.read_sketches32(os.path.join(d, "ref.sketch"))
    _, qnames, qh, qoff = ok.read_sketches32(os.path.join(d, "qry.sketch"))
    return d, man, (rnames, rh, roff), (qnames, qh, qoff)


def test_index_build_matches_dict_and_index_files(ctx):
    _, man, (rnames, rh, roff), _ = load_dist_case()
    postings, counts = ok.index_build32(rh, roff, man["hash_bits"])
    sk = ctx.sketches_from_host(rh, roff)
    idx = ctx.index_build(sk, man["hash_bits"])
    p2, c2 = idx.export()
    assert np.array_equal(p2, postings)      # .dict payload, src/sketch.cpp:991-1001
    assert np.array_equal(c2, counts)        # .index payload, src/sketch.cpp:1008-1011
    assert idx.total == len(rh) and idx.genomes == len(rnames)
    assert idx.distinct == int((counts > 0).sum())
    assert idx.sum_sq == int((counts.astype(np.int64) ** 2).sum())


def test_golden_alldist_and_dist_text(ctx):
    d, man, (rnames, rh, roff), (qnames, qh, qoff) = load_dist_case()
    kmer = 2 * man["half_k"]
    rsk = ctx.sketches_from_host(rh, roff)
    qsk = ctx.sketches_from_host(qh, qoff)
    built = ctx.index_build(rsk, man["hash_bits"])
    postings, counts = ok.index_build32(rh, roff, man["hash_bits"])
    imported = ctx.index_import(postings, counts, man["hash_bits"], np.diff(roff))
    for case in man["cases"]:
        want = open(os.path.join(d, case["file"])).read().split("\n")[:-1]
        if case["cmd"] == "alldist":
            variants = [(built, None), (built, rsk), (imported, rsk)]
            for idx, q in variants:
                hits, _ = ctx.dist_rows(idx, q, 1, case["metric"], kmer, case["max_dist"])
                mine = sorted(capi.format_hit(rnames[h["col"]], rnames[h["row"]], h).rstrip("\n")
                              for h in hits)
                assert mine == want, case["file"]
        else:
            for idx in (built, imported):
                hits, _ = ctx.dist_rows(idx, qsk, 0, case["metric"], kmer, case["max_dist"])
                if case["max_neighbor"]:
                    hits = capi.topn_rows(hits, case["max_neighbor"])
                mine = [capi.format_hit(qnames[h["row"]], rnames[h["col"]], h).rstrip("\n") for h in hits]
                assert mine == want, case["file"]


def test_dense_counts_bit_exact(ctx):
    _, man, (rnames, rh, roff), (qnames, qh, qoff) = load_dist_case()
    postings, counts = ok.index_build32(rh, roff, man["hash_bits"])
    rsizes = np.diff(roff).astype(np.uint32)
    idx = ctx.index_build(ctx.sketches_from_host(rh, roff), man["hash_bits"])
    for (h, off, tri) in ((qh, qoff, 0), (rh, roff, 1)):
        want_hits, want = ok.index_dist32(counts, man["hash_bits"], postings, rsizes, h, off, tri, 0, 16,
                                          0.2, want_dense=True)
        hits, dense = ctx.dist_rows(idx, ctx.sketches_from_host(h, off), tri, 0, 16, 0.2, want_dense=True)
        assert np.array_equal(dense, want)
        assert_hits_equal(hits, want_hits)


@pytest.mark.parametrize("n,m,bits,seed", [(300, 150, 24, 1), (1000, 60, 20, 2), (64, 1200, 28, 3)])
def test_random_alldist_vs_oracle(ctx, n, m, bits, seed):
    names, h, off = synth.clade_sketches(n, m, bits, seed=seed)
    postings, counts = ok.index_build32(h, off, bits)
    sizes = np.diff(off).astype(np.uint32)
    idx = ctx.index_build(ctx.sketches_from_host(h, off), bits)
    for metric, D in ((0, 0.05), (1, 0.1), (0, 1.0)):
        want, _ = ok.index_dist32(counts, bits, postings, sizes, h, off, 1, metric, 20, D, threads=4)
        mine, _ = ctx.dist_rows(idx, None, 1, metric, 20, D)
        assert_hits_equal(mine, want)
    # dense mode (threshold admits distance 1.0): every j>i pair is reported
    want, _ = ok.index_dist32(counts, bits, postings, sizes, h, off, 1, 0, 20, 1.5, threads=4)
    mine, _ = ctx.dist_rows(idx, None, 1, 0, 20, 1.5)
    assert len(mine) == n * (n - 1) // 2
    assert_hits_equal(mine, want)


@pytest.mark.parametrize("n,m,bits", [(600, 200, 12), (800, 300, 10)])
def test_crowded_hash_space_long_lists_and_full_rows(ctx, n, m, bits):
    # a tiny hash space: every hash sits in dozens to hundreds of genomes (posting lists far
    # beyond the 8 a quad gathers and the 16 more the early loads cover: the whole-wave
    # streaming path) and every pair shares hashes (hundreds of non-zero cells per row: more
    # than the LDS cell list holds, the column-walk path)
    rng = np.random.default_rng(n + bits)
    parts = [np.unique(rng.integers(0, 1 << bits, size=m, dtype=np.uint64).astype(np.uint32)) for _ in range(n)]
    off = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
    h = np.concatenate(parts)
    postings, counts = ok.index_build32(h, off, bits)
    assert counts.max() > 8 + 16 and (bits > 10 or counts.max() > 8 + 16 + 128)
    sizes = np.diff(off).astype(np.uint32)
    idx = ctx.index_build(ctx.sketches_from_host(h, off), bits)
    qs = ctx.sketches_from_host(h, off)
    want_hits, want = ok.index_dist32(counts, bits, postings, sizes, h, off, 1, 0, 20, 0.2, threads=4, want_dense=True)
    assert (np.triu(want, 1) > 0).sum(axis=1).max() > 256
    hits, dense = ctx.dist_rows(idx, qs, 1, 0, 20, 0.2, want_dense=True)
    assert np.array_equal(dense, want)
    assert_hits_equal(hits, want_hits)
    for metric, D in ((0, 0.12), (1, 0.08), (0, 0.3)):
        want_hits, _ = ok.index_dist32(counts, bits, postings, sizes, h, off, 1, metric, 20, D, threads=4)
        assert len(want_hits) > 0 or metric == 1
        mine, _ = ctx.dist_rows(idx, None, 1, metric, 20, D)          # self join: unfiltered slices
        assert_hits_equal(mine, want_hits)
        mine, _ = ctx.dist_rows(idx, qs, 1, metric, 20, D)            # explicit queries: filtered postings
        assert_hits_equal(mine, want_hits)
    # ref-vs-query over the same crowded index
    want_hits, _ = ok.index_dist32(counts, bits, postings, sizes, h[: int(off[50])], off[:51], 0, 0, 20, 0.12, threads=4)
    mine, _ = ctx.dist_rows(idx, ctx.sketches_from_host(h[: int(off[50])], off[:51]), 0, 0, 20, 0.12)
    assert_hits_equal(mine, want_hits)


def test_query_path_long_lists_keep_the_wave_queue_bounded(ctx):
    # explicit queries of ~1,800 hashes against 3,000 references in a 13-bit hash space: every query hash is indexed and
    # its list holds ~100 genomes spread over all ids (no compact record), so every walk re-queues all 64 lists it
    # popped while the look-ups of the same wave keep appending 64 more: the per-wave queue must drain before it
    # appends (round-2 advice: `if (qn >= 64) walk()` let it run past its 128 slots into the next wave's queue)
    rng = np.random.default_rng(77)
    bits = 13
    parts = [np.unique(rng.integers(0, 1 << bits, size=300, dtype=np.uint64).astype(np.uint32)) for _ in range(3000)]
    roff = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
    rh = np.concatenate(parts)
    qparts = [np.unique(rng.integers(0, 1 << bits, size=2000, dtype=np.uint64).astype(np.uint32)) for _ in range(24)]
    qoff = np.concatenate([[0], np.cumsum([len(p) for p in qparts])]).astype(np.uint64)
    qh = np.concatenate(qparts)
    postings, counts = ok.index_build32(rh, roff, bits)
    assert counts.min() > 64
    sizes = np.diff(roff).astype(np.uint32)
    idx = ctx.index_build(ctx.sketches_from_host(rh, roff), bits)
    want_hits, want = ok.index_dist32(counts, bits, postings, sizes, qh, qoff, 0, 0, 20, 0.14, threads=4, want_dense=True)
    hits, dense = ctx.dist_rows(idx, ctx.sketches_from_host(qh, qoff), 0, 0, 20, 0.14, want_dense=True)
    assert np.array_equal(dense, want)
    assert_hits_equal(hits, want_hits)
    want_hits, _ = ok.index_dist32(counts, bits, postings, sizes, qh, qoff, 0, 1, 20, 0.07, threads=4)
    assert len(want_hits) > 1000
    hits, _ = ctx.dist_rows(idx, ctx.sketches_from_host(qh, qoff), 0, 1, 20, 0.07)   # sparse report, clean rows
    assert_hits_equal(hits, want_hits)


@pytest.mark.parametrize("n,block,world", [(500, 2, 4), (501, 16, 3), (37, 16, 8), (1000, 6, 5)])
def test_block_cyclic_row_sharding_union_equals_full(ctx, n, block, world):
    # blocks of `block` rows dealt round-robin to `world` shards; even blocks run the pair kernel
    names, h, off = synth.clade_sketches(n, 100, 22, seed=19)
    idx = ctx.index_build(ctx.sketches_from_host(h, off), 22)
    for D in (0.08, 1.5):  # sparse and dense epilogue
        full, _ = ctx.dist_rows(idx, None, 1, 0, 20, D)
        parts = [ctx.dist_rows(idx, None, 1, 0, 20, D, row_first=r, row_step=world, row_block=block)[0] for r in range(world)]
        for r, p in enumerate(parts):
            assert np.all(idx.shard_of(p, world, block) == r)   # blocks of the index's internal genome order
        merged = np.concatenate(parts)
        merged = merged[np.lexsort((merged["col"], merged["row"]))]
        assert merged.tobytes() == full.tobytes()


def test_default_threshold_of_alldist_stays_sparse():
    # `alldist` without -D compares `dist < 1.0` (src/main.cpp:46, src/dist.cpp:232): pairs that share nothing are NOT
    # reported.  The synchronous API hands the kernel a threshold a few ulps wider; that must not turn the call into the
    # dense report (every pair a 40-byte record, sorted and downloaded, then dropped again on the host)
    c = capi.Context(0)
    n = 3000
    names, h, off = synth.clade_sketches(n, 100, 22, seed=41)
    postings, counts = ok.index_build32(h, off, 22)
    sizes = np.diff(off).astype(np.uint32)
    idx = c.index_build(c.sketches_from_host(h, off), 22)
    want, _ = ok.index_dist32(counts, 22, postings, sizes, h, off, 1, 0, 20, 1.0, threads=4)
    assert 0 < len(want) < n * (n - 1) // 2 // 10
    before = c.pool_stats()[0]
    mine, _ = c.dist_rows(idx, None, 1, 0, 20, 1.0)
    assert_hits_equal(mine, want)
    assert c.pool_stats()[0] - before < 40 * n * (n - 1) // 2 // 4   # no O(n^2) hit buffer
    mine, _ = c.dist_rows(idx, c.sketches_from_host(h, off), 1, 0, 20, 1.0)   # explicit queries, same rule
    assert_hits_equal(mine, want)
    # `dist` compares `<=` (src/dist.cpp:624): there the default reports every pair
    q = c.sketches_from_host(h[: int(off[40])], off[:41])
    want, _ = ok.index_dist32(counts, 22, postings, sizes, h[: int(off[40])], off[:41], 0, 0, 20, 1.0, threads=4)
    assert len(want) == 40 * n
    mine, _ = c.dist_rows(idx, q, 0, 0, 20, 1.0)
    assert_hits_equal(mine, want)
    c.close()


@pytest.mark.parametrize("mode", ["shuffled", "jitter", "sorted"])
def test_genome_order_does_not_matter(ctx, mode):
    # The same collection listed in another order (a random permutation of the genome ids; OpenMP completion order, i.e.
    # permuted inside windows, src/sketch.cpp:558-568).  rk_index_build renumbers the genomes internally so that relatives
    # are neighbours again; hits, dense rows and the exported .dict postings must be the oracle's for the order GIVEN.
    n = 1500
    names, h, off = synth.clade_sketches(n, 220, 24, seed=61)
    order = synth.genome_order(n, mode, seed=3, window=64)
    names, h, off = synth.permute_genomes(names, h, off, order)
    postings, counts = ok.index_build32(h, off, 24)
    sizes = np.diff(off).astype(np.uint32)
    sk = ctx.sketches_from_host(h, off)
    idx = ctx.index_build(sk, 24)
    assert np.array_equal(idx.export(want_counts=False)[0], postings)
    # clades (10 consecutive genomes of the generator) end up as runs of consecutive internal ids
    internal = order[idx.order.astype(np.int64)] // 10          # clade of every internal position
    assert (np.diff(internal) != 0).sum() < 1.2 * (n // 10)
    for metric, D in ((0, 0.05), (1, 0.2), (0, 1.5)):
        want, _ = ok.index_dist32(counts, 24, postings, sizes, h, off, 1, metric, 20, D, threads=4)
        assert len(want) > 0
        mine, _ = ctx.dist_rows(idx, None, 1, metric, 20, D)
        assert_hits_equal(mine, want)
    # explicit queries against the renumbered index: triangle on the caller's ids, dense rows in the caller's columns
    want, dense_want = ok.index_dist32(counts, 24, postings, sizes, h, off, 1, 0, 20, 0.08, threads=4, want_dense=True)
    mine, dense = ctx.dist_rows(idx, sk, 1, 0, 20, 0.08, want_dense=True)
    assert np.array_equal(dense, dense_want)
    assert_hits_equal(mine, want)
    q = ctx.sketches_from_host(h[: int(off[70])], off[:71])
    for D in (0.08, 1.0):
        want, _ = ok.index_dist32(counts, 24, postings, sizes, h[: int(off[70])], off[:71], 0, 0, 20, D, threads=4)
        mine, _ = ctx.dist_rows(idx, q, 0, 0, 20, D)
        assert_hits_equal(mine, want)
    # row shards partition the pairs; a shard is a set of blocks of the internal order
    full, _ = ctx.dist_rows(idx, None, 1, 0, 20, 0.05)
    parts = [ctx.dist_rows(idx, None, 1, 0, 20, 0.05, row_first=r, row_step=3, row_block=16)[0] for r in range(3)]
    for r, p in enumerate(parts):
        assert np.all(idx.shard_of(p, 3, 16) == r)
    merged = np.concatenate(parts)
    assert merged[np.lexsort((merged["col"], merged["row"]))].tobytes() == full.tobytes()
    # the single-blob form (multi-GPU broadcast) carries the order
    blob = torch.empty(idx.blob_bytes, dtype=torch.uint8, device="cuda")
    idx.pack_dev(blob.data_ptr(), blob.numel())
    idx2 = ctx.index_unpack_dev(blob.data_ptr(), blob.numel())
    assert np.array_equal(idx2.order, idx.order)
    assert ctx.dist_rows(idx2, None, 1, 0, 20, 0.05)[0].tobytes() == full.tobytes()


@pytest.mark.parametrize("n,m,bits", [(3000, 700, 28), (20000, 90, 26), (150, 60, 24)])
def test_index_partition_in_one_pass_and_in_two_gives_the_same_index(ctx, monkeypatch, n, m, bits):
    # the bucket partition of the fast build: two coalescing passes (default; needs >= 128 buckets and six spare key bits)
    # or one scattering pass (RK_INDEX_PART2=0), with and without the XCD-contiguous workgroup mapping: the exported
    # .dict/.index payloads must be the oracle's in every combination, and so must the hits
    names, h, off = synth.clade_sketches(n, m, bits, seed=300 + n)
    postings, counts = ok.index_build32(h, off, bits)
    hits = None
    for part2, xcd in (("1", "1"), ("0", "1"), ("1", "0"), ("0", "0")):
        monkeypatch.setenv("RK_INDEX_PART2", part2)
        monkeypatch.setenv("RK_INDEX_XCD", xcd)
        idx = ctx.index_build(ctx.sketches_from_host(h, off), bits)
        assert idx.built_fast
        p2, c2 = idx.export()
        assert np.array_equal(p2, postings) and np.array_equal(c2, counts)
        got = ctx.dist_rows(idx, None, 1, 0, 20, 0.05)[0].tobytes()
        assert hits is None or got == hits
        hits = got
        del idx


def test_index_without_renumbering_gives_the_same_results(monkeypatch):
    names, h, off = synth.clade_sketches(900, 150, 24, seed=62)
    names, h, off = synth.permute_genomes(names, h, off, synth.genome_order(900, "shuffled", seed=4))
    monkeypatch.setenv("RK_INDEX_RELABEL", "0")
    c0 = capi.Context(0)
    monkeypatch.delenv("RK_INDEX_RELABEL")
    c1 = capi.Context(0)
    i0, i1 = c0.index_build(c0.sketches_from_host(h, off), 24), c1.index_build(c1.sketches_from_host(h, off), 24)
    assert np.array_equal(i0.order, np.arange(900)) and not np.array_equal(i1.order, np.arange(900))
    for D in (0.05, 0.3):
        assert c0.dist_rows(i0, None, 1, 0, 20, D)[0].tobytes() == c1.dist_rows(i1, None, 1, 0, 20, D)[0].tobytes()
    c0.close()
    c1.close()


@pytest.mark.parametrize("strains,n,m,D,metric", [(10, 2000, 400, 0.05, 0), (70, 1500, 300, 0.05, 0), (25, 1201, 500, 0.08, 1),
                                                  (10, 900, 300, 0.02, 0)])
def test_near_window_self_join_and_its_fallback(monkeypatch, strains, n, m, D, metric):
    # rk_near_kernel keeps no counter rows: a unit's members inside the 32 columns behind its first row are counted in
    # registers, and a unit whose members BEYOND the window could add up to a reportable pair goes to the fallback list
    # (rk_dist_kernel in list mode).  Clades of 10: nothing falls back.  Clades of 25 / 70: the relatives of a row reach
    # beyond the window -- most units fall back.  Results must be the oracle's either way, also for row shards, odd
    # collection sizes, the containment metric and single rows (RK_DIST_PAIR=2).
    names, h, off = synth.clade_sketches(n, m, 26, strains_per_clade=strains, seed=100 + strains)
    order = synth.genome_order(n, "shuffled", seed=strains)
    names, h, off = synth.permute_genomes(names, h, off, order)
    postings, counts = ok.index_build32(h, off, 26)
    sizes = np.diff(off).astype(np.uint32)
    want, _ = ok.index_dist32(counts, 26, postings, sizes, h, off, 1, metric, 20, D, threads=4)
    assert len(want) > n
    monkeypatch.setenv("RK_DIST_TILES", "0")    # (a collection with wide clusters would take the tile kernel)
    c = capi.Context(0)
    monkeypatch.setenv("RK_DIST_PAIR", "2")
    c1 = capi.Context(0)
    monkeypatch.delenv("RK_DIST_PAIR")
    monkeypatch.delenv("RK_DIST_TILES")
    for cc, pair in ((c, "true"), (c1, "false")):
        idx = cc.index_build(cc.sketches_from_host(h, off), 26)
        assert cc.dist_kernel_name(idx, None, 1, metric, 20, D).startswith("rk_near_kernel<%s, " % pair)
        assert_hits_equal(cc.dist_rows(idx, None, 1, metric, 20, D)[0], want)
        assert_hits_equal(cc.dist_rows(idx, None, 1, metric, 20, D)[0], want)      # again: the fallback list was reset
        # and again: once a completed launch with these options has shown the list to be empty, the (empty) fallback launch is
        # skipped -- and never when it was not empty; other options in between forget what was known
        assert_hits_equal(cc.dist_rows(idx, None, 1, metric, 20, D)[0], want)
        other, _ = ok.index_dist32(counts, 26, postings, sizes, h, off, 1, metric, 20, D * 0.5, threads=4)
        assert_hits_equal(cc.dist_rows(idx, None, 1, metric, 20, D * 0.5)[0], other)
        assert_hits_equal(cc.dist_rows(idx, None, 1, metric, 20, D * 0.5)[0], other)
        assert_hits_equal(cc.dist_rows(idx, None, 1, metric, 20, D)[0], want)
        parts = [cc.dist_rows(idx, None, 1, metric, 20, D, row_first=r, row_step=3, row_block=16)[0] for r in range(3)]
        for r, p in enumerate(parts):
            assert np.all(idx.shard_of(p, 3, 16) == r)
        merged = np.concatenate(parts)
        assert_hits_equal(merged[np.lexsort((merged["col"], merged["row"]))], want)
    c.close()
    c1.close()


@pytest.mark.parametrize("strains,n,m,D,metric,tiny", [(10, 2000, 400, 0.05, 0, 0), (70, 1500, 300, 0.05, 0, 3), (100, 2100, 300, 0.05, 1, 2),
                                                       (300, 1801, 200, 0.05, 0, 0), (1000, 3000, 150, 0.03, 0, 5), (10, 777, 60, 0.2, 0, 4)])
def test_tile_self_join(monkeypatch, strains, n, m, D, metric, tiny):
    # rk_tile_kernel counts 32 x 32 tiles of the pair matrix from (row mask, column mask) records, one per posting list and
    # pair of blocks it touches; tiles with fewer records than a reportable cell needs are skipped, every other tile is
    # counted exactly: species of any width, tiny sketches, loose thresholds, both metrics, row shards -- no fallback.
    names, h, off = synth.clade_sketches(n, m, 26, strains_per_clade=strains, seed=200 + strains, tiny=tiny)
    n = len(names)
    order = synth.genome_order(n, "shuffled", seed=strains)
    names, h, off = synth.permute_genomes(names, h, off, order)
    postings, counts = ok.index_build32(h, off, 26)
    sizes = np.diff(off).astype(np.uint32)
    want, _ = ok.index_dist32(counts, 26, postings, sizes, h, off, 1, metric, 20, D, threads=4)
    assert len(want) > n
    monkeypatch.setenv("RK_DIST_TILES", "1")
    c = capi.Context(0)
    monkeypatch.delenv("RK_DIST_TILES")
    auto = capi.Context(0)
    idx = c.index_build(c.sketches_from_host(h, off), 26)
    assert c.dist_kernel_name(idx, None, 1, metric, 20, D).startswith("rk_tile_kernel")   # (asking changes nothing: the records are built by the first launch)
    assert_hits_equal(c.dist_rows(idx, None, 1, metric, 20, D)[0], want)
    assert c.dist_kernel_name(idx, None, 1, metric, 20, D).startswith("rk_tile_kernel<")
    other, _ = ok.index_dist32(counts, 26, postings, sizes, h, off, 1, 1 - metric, 20, D * 0.5, threads=4)
    for srow in ("0", "1"):   # both variants of the kernel (masks through LDS / row masks as 64-bit scalars), whatever the launch would pick
        monkeypatch.setenv("RK_TILE_SROW", srow)
        assert c.dist_kernel_name(idx, None, 1, metric, 20, D).endswith(", true>" if srow == "1" else ", false>")
        assert_hits_equal(c.dist_rows(idx, None, 1, metric, 20, D)[0], want)
        assert_hits_equal(c.dist_rows(idx, None, 1, 1 - metric, 20, D * 0.5)[0], other)
        for step, block in ((3, 16), (2, 64), (5, 1)):
            parts = [c.dist_rows(idx, None, 1, metric, 20, D, row_first=r, row_step=step, row_block=block)[0] for r in range(step)]
            for r, p in enumerate(parts):
                assert np.all(idx.shard_of(p, step, block) == r)
            merged = np.concatenate(parts)
            assert_hits_equal(merged[np.lexsort((merged["col"], merged["row"]))], want)
    monkeypatch.delenv("RK_TILE_SROW")
    # a dense report (-D 1.0 under `dist` semantics is not a self join; alldist with D > 1) stays with the counter rows
    assert c.dist_kernel_name(idx, None, 1, metric, 20, 1.5).startswith("rk_dist_kernel")
    # the default: the tile kernel for collections with clusters wider than the window of rk_near_kernel
    idx2 = auto.index_build(auto.sketches_from_host(h, off), 26)
    name = auto.dist_kernel_name(idx2, None, 1, metric, 20, D)
    assert strains < 70 or name.startswith("rk_tile_kernel")
    if strains == 10 and D < 0.1:
        assert name.startswith("rk_near_kernel")
    assert_hits_equal(auto.dist_rows(idx2, None, 1, metric, 20, D)[0], want)
    del idx, idx2
    c.close()
    auto.close()


@pytest.mark.parametrize("threads,from_build", [("256", "0"), ("512", "1"), ("1024", "0"), ("256", "1")])
def test_tile_kernel_heavy_tiles_move_their_planes_mid_tile(monkeypatch, threads, from_build):
    # 96 genomes (three blocks of 32) sharing up to 30,000 hashes: tiles of 30,000+ records, more than eleven bit planes
    # hold for a lane of a four-wave workgroup (2 half-waves x 4 waves x 1,984) -- the planes move to the LDS counts in the
    # middle of the tile (before the end-of-tile merge of the waves' planes adds the rest) -- and sketch sizes that leave
    # steps of 1..8 groups of eight records behind (every arm of the adder tree: carries of weight 64, 32, 16 and 8)
    rng = np.random.default_rng(5)
    core = np.unique(rng.integers(0, 1 << 26, size=30000, dtype=np.uint64).astype(np.uint32))
    parts = []
    for g in range(96):
        keep = core[rng.random(len(core)) < (0.97 if g < 40 else 0.6)]
        own = rng.integers(0, 1 << 26, size=37 * (g % 9), dtype=np.uint64).astype(np.uint32)
        parts.append(np.unique(np.concatenate([keep, own])))
    off = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
    h = np.concatenate(parts)
    postings, counts = ok.index_build32(h, off, 26)
    sizes = np.diff(off).astype(np.uint32)
    monkeypatch.setenv("RK_DIST_TILES", "1")
    monkeypatch.setenv("RK_TILE_THREADS", threads)
    monkeypatch.setenv("RK_INDEX_TILES", from_build)   # the records from the build's bucket emission / from the postings, on first use
    c = capi.Context(0)
    idx = c.index_build(c.sketches_from_host(h, off), 26)
    assert idx.products == (6 if from_build == "1" else 1)
    for metric, D in ((0, 0.2), (1, 0.02)):
        want, _ = ok.index_dist32(counts, 26, postings, sizes, h, off, 1, metric, 20, D, threads=8)
        assert len(want) > 500 and want["common"].max() > 25000
        for srow in ("false", "true"):   # both variants of the kernel: masks through LDS / row masks as 64-bit scalars
            monkeypatch.setenv("RK_TILE_SROW", "1" if srow == "true" else "0")
            assert_hits_equal(c.dist_rows(idx, None, 1, metric, 20, D)[0], want)
            assert c.dist_kernel_name(idx, None, 1, metric, 20, D) == "rk_tile_kernel<%su, %s>" % (threads, srow)
    del idx
    c.close()


def test_self_join_kernel_follows_size_and_shape_not_call_history(monkeypatch):
    # (round 5) which kernel a self join takes is decided by the index's size / shape and the options, never by how often the
    # index was joined before: below RK_DIST_TILES_MIN_GENOMES the build emits slice records and every join runs on
    # rk_near_kernel; from there on the build emits TILE records and the first join already runs on rk_tile_kernel -- in a
    # single-shot context (the command-line tool) too.  rk_dist_kernel_name changes nothing.  Same hits every time.
    names, h, off = synth.clade_sketches(2000, 600, 26, seed=91)
    postings, counts = ok.index_build32(h, off, 26)
    sizes = np.diff(off).astype(np.uint32)
    want, _ = ok.index_dist32(counts, 26, postings, sizes, h, off, 1, 0, 20, 0.05, threads=4)
    c = capi.Context(0)   # (default switches: 2,000 genomes are below the size from which tile records pay, 4,000)
    idx = c.index_build(c.sketches_from_host(h, off), 26)
    assert idx.products == 1
    for _ in range(3):
        assert c.dist_kernel_name(idx, None, 1, 0, 20, 0.05).startswith("rk_near_kernel")
        assert_hits_equal(c.dist_rows(idx, None, 1, 0, 20, 0.05)[0], want)
    assert idx.products == 1
    del idx
    c.close()
    monkeypatch.setenv("RK_DIST_TILES_MIN_GENOMES", "1000")
    for single_shot in (False, True):
        c = capi.Context(0)
        if single_shot:
            c.set_single_shot(True)
        idx = c.index_build(c.sketches_from_host(h, off), 26)
        assert idx.products == 2 | 4 and idx.self_stats[3] > 0
        p2, c2 = idx.export()
        assert np.array_equal(p2, postings) and np.array_equal(c2, counts)
        seen = []
        for _ in range(3):
            seen.append(c.dist_kernel_name(idx, None, 1, 0, 20, 0.05).split("<")[0])
            assert_hits_equal(c.dist_rows(idx, None, 1, 0, 20, 0.05)[0], want)
        assert seen == ["rk_tile_kernel"] * 3, seen
        # a small row shard: the tile kernel as well while the index has no slice records ...
        assert c.dist_kernel_name(idx, None, 1, 0, 20, 0.05, row_first=1, row_step=4, row_block=32).startswith("rk_tile_kernel<")
        parts = [c.dist_rows(idx, None, 1, 0, 20, 0.05, row_first=r, row_step=4, row_block=32)[0] for r in range(4)]
        merged = np.concatenate(parts)
        assert_hits_equal(merged[np.lexsort((merged["col"], merged["row"]))], want)
        # ... a dense report makes them on first use (every pair, counter rows), and small shards then prefer the near-window kernel
        dense_want, _ = ok.index_dist32(counts, 26, postings, sizes, h, off, 1, 0, 20, 1.5, threads=4)
        assert_hits_equal(c.dist_rows(idx, None, 1, 0, 20, 1.5)[0], dense_want)
        assert idx.products == 1 | 2 | 4
        assert c.dist_kernel_name(idx, None, 1, 0, 20, 0.05, row_first=1, row_step=4, row_block=32).startswith("rk_near_kernel")
        assert c.dist_kernel_name(idx, None, 1, 0, 20, 0.05).startswith("rk_tile_kernel<")
        parts = [c.dist_rows(idx, None, 1, 0, 20, 0.05, row_first=r, row_step=4, row_block=32)[0] for r in range(4)]
        merged = np.concatenate(parts)
        assert_hits_equal(merged[np.lexsort((merged["col"], merged["row"]))], want)
        del idx
        c.close()


@pytest.mark.parametrize("strains,n,m,tiny", [(10, 3000, 300, 0), (100, 3000, 200, 0), (1000, 3000, 120, 0), (10, 2500, 300, 2), (36, 1296, 500, 0)])
def test_tile_records_from_the_build_equal_the_lazily_built_ones(monkeypatch, strains, n, m, tiny):
    # RK_INDEX_TILES=1: the bucket emission writes tile records (rk_index_tiles.inc); RK_INDEX_TILES=0 + RK_DIST_TILES=1: slice
    # records, and the tile records derived from the postings on the first join (rk_tiles.hip).  Same postings, same number of
    # tile records, and for tight, loose and default thresholds and both metrics the oracle's hits from both.
    names, h, off = synth.clade_sketches(n, m, 26, strains_per_clade=strains, seed=500 + strains, tiny=tiny)
    postings, counts = ok.index_build32(h, off, 26)
    sizes = np.diff(off).astype(np.uint32)
    monkeypatch.setenv("RK_DIST_TILES", "1")
    monkeypatch.setenv("RK_INDEX_TILES", "1")
    a = capi.Context(0)
    monkeypatch.setenv("RK_INDEX_TILES", "0")
    b = capi.Context(0)
    ia = a.index_build(a.sketches_from_host(h, off), 26)
    ib = b.index_build(b.sketches_from_host(h, off), 26)
    assert ia.products == 6 and ib.products == 1
    pa, ca = ia.export()
    assert np.array_equal(pa, postings) and np.array_equal(ca, counts)
    assert np.array_equal(ia.order, ib.order)
    for metric, D in ((0, 0.05), (1, 0.05), (0, 0.3), (0, 1.0)):
        want, _ = ok.index_dist32(counts, 26, postings, sizes, h, off, 1, metric, 20, D, threads=8)
        assert a.dist_kernel_name(ia, None, 1, metric, 20, D).startswith("rk_tile_kernel<")
        assert_hits_equal(a.dist_rows(ia, None, 1, metric, 20, D)[0], want)
        assert_hits_equal(b.dist_rows(ib, None, 1, metric, 20, D)[0], want)
    assert ib.products == 3 and ia.self_stats[3] == ib.self_stats[3]
    del ia, ib
    a.close()
    b.close()


@pytest.mark.parametrize("strains,n,m,W", [(10, 3000, 300, 2), (10, 3000, 300, 8), (100, 2600, 200, 4), (36, 1296, 500, 8)])
def test_sharded_build_and_join_equal_the_whole(monkeypatch, strains, n, m, W):
    # (round 5) the all-vs-all sharded twice: every shard builds the posting lists of ONE range of the hash space
    # (rk_index_build_shard) and groups their tile records by the shard that owns the row block; the records are exchanged (here:
    # by slicing the send buffers of one process); every shard sorts what arrived (rk_index_join_shard) and joins ITS rows.
    # The shards' postings, in range order, are the whole .dict; the union of their hits is the oracle's result; a hit belongs to
    # the shard of its row block.
    import torch
    names, h, off = synth.clade_sketches(n, m, 26, strains_per_clade=strains, seed=700 + strains)
    order = synth.genome_order(len(names), "shuffled", seed=W)
    names, h, off = synth.permute_genomes(names, h, off, order)
    postings, counts = ok.index_build32(h, off, 26)
    sizes = np.diff(off).astype(np.uint32)
    c = capi.Context(0)
    sk = c.sketches_from_host(h, off)
    parts = [c.index_build_shard(sk, 26, r, W) for r in range(W)]
    assert sum(p.total for p in parts) == len(h)
    assert np.array_equal(np.concatenate([p.export(want_counts=False)[0][: p.total] for p in parts]), postings)
    for p in parts[1:]:
        assert np.array_equal(p.order, parts[0].order)          # every shard computes the same internal genome order
    with pytest.raises(Exception, match="rk_index_join_shard"):
        c.dist_rows(parts[0], None, 1, 0, 20, 0.05)
    sent = [p.shard_records(W) for p in parts]
    bufs = []
    for p, cnt in zip(parts, sent):
        b = torch.empty(max(1, sum(cnt) * 12), dtype=torch.uint8, device="cuda")
        p.shard_pack(b.data_ptr())
        bufs.append(b)
    torch.cuda.synchronize()
    joins = []
    for d in range(W):
        chunks = [bufs[r][12 * sum(sent[r][:d]): 12 * sum(sent[r][:d + 1])] for r in range(W)]
        recv = torch.cat(chunks) if sum(len(x) for x in chunks) else torch.empty(1, dtype=torch.uint8, device="cuda")
        joins.append((c.index_join_shard(parts[d], recv.data_ptr(), sum(sent[r][d] for r in range(W))), recv))
    whole = c.index_build(sk, 26)
    assert sum(j.self_stats[3] for j, _ in joins) == (whole.self_stats[3] if whole.products & 2 else sum(sum(x) for x in sent))
    for metric, D in ((0, 0.05), (1, 0.05), (0, 0.3), (0, 1.0)):
        want, _ = ok.index_dist32(counts, 26, postings, sizes, h, off, 1, metric, 20, D, threads=8)
        got = []
        for d, (j, _) in enumerate(joins):
            hits, _ = c.dist_rows(j, None, 1, metric, 20, D)
            assert np.all(parts[0].shard_of(hits, W, 32) == d)
            got.append(hits)
        merged = np.concatenate(got)
        assert_hits_equal(merged[np.lexsort((merged["col"], merged["row"]))], want)
    del joins, parts, whole
    c.close()


@pytest.mark.parametrize("pass_bits", ["1", "3"])
def test_index_build_in_several_passes_over_the_hash_space(monkeypatch, pass_bits):
    # a collection with more postings than 2^15 buckets hold is built range by range of the hash space, the postings of a pass
    # behind those of the pass before (RK_INDEX_PASS_BITS forces it on a small one): same .dict / .index, same hits
    names, h, off = synth.clade_sketches(3000, 300, 26, seed=41, tiny=2)
    postings, counts = ok.index_build32(h, off, 26)
    sizes = np.diff(off).astype(np.uint32)
    monkeypatch.setenv("RK_INDEX_PASS_BITS", pass_bits)
    monkeypatch.setenv("RK_INDEX_TILES", "1")
    c = capi.Context(0)
    idx = c.index_build(c.sketches_from_host(h, off), 26)
    assert idx.products == 6 and idx.built_fast and idx.total == len(h)
    p2, c2 = idx.export()
    assert np.array_equal(p2, postings) and np.array_equal(c2, counts)
    for metric, D in ((0, 0.05), (1, 0.1), (0, 1.0)):
        want, _ = ok.index_dist32(counts, 26, postings, sizes, h, off, 1, metric, 20, D, threads=4)
        assert_hits_equal(c.dist_rows(idx, None, 1, metric, 20, D)[0], want)
    q = c.sketches_from_host(h[: int(off[64])], off[:65])   # explicit queries read the same postings
    want, _ = ok.index_dist32(counts, 26, postings, sizes, h[: int(off[64])], off[:65], 0, 0, 20, 0.1, threads=4)
    assert_hits_equal(c.dist_rows(idx, q, 0, 0, 20, 0.1)[0], want)
    del idx, q
    c.close()


def test_hashes_that_crowd_one_end_of_the_hash_space(monkeypatch):
    # a real sketch's hashes are pieces of canonical k-mers: the quarter of the hash space that starts with A is seven times as full
    # as the one that starts with T (synth.canonical_skew; twice over here: the fullest buckets hold 3x the mean and leave the LDS
    # sort for k_bucket_heavy although no list is long) -- same .dict / .index and hits as the oracle, tile records from the build
    names, h, off = synth.clade_sketches(6000, 500, 26, seed=41)
    h, off = synth.canonical_skew(h, off, 26, levels=2)
    postings, counts = ok.index_build32(h, off, 26)
    sizes = np.diff(off).astype(np.uint32)
    top = np.bincount(h >> (26 - 11), minlength=1 << 11)
    assert top.max() > 4096 and counts.max() < 200          # (buckets beyond the LDS sort, and not because of a long list)
    monkeypatch.setenv("RK_INDEX_TILES", "1")
    c = capi.Context(0)
    idx = c.index_build(c.sketches_from_host(h, off), 26)
    assert idx.built_fast and idx.products == 6
    p2, c2 = idx.export()
    assert np.array_equal(p2, postings) and np.array_equal(c2, counts)
    for metric, D in ((0, 0.05), (1, 0.05), (0, 1.0)):
        want, _ = ok.index_dist32(counts, 26, postings, sizes, h, off, 1, metric, 20, D, threads=8)
        assert_hits_equal(c.dist_rows(idx, None, 1, metric, 20, D)[0], want)
    del idx
    c.close()


def test_a_range_with_more_keys_than_estimated_is_built_again(monkeypatch):
    # the ranges of a real hash space are not equally full (the hashes are pieces of k-mers: base composition shows in their top
    # bits); a range pass -- of a one-GPU build or a shard -- whose keys exceed the estimated buffers has counted what it needs and
    # is repeated with that, instead of refusing the collection (a shard) or taking the device-wide sort (one GPU)
    names, h, off = synth.clade_sketches(3000, 400, 26, strains_per_clade=36, seed=77)
    postings, counts = ok.index_build32(h, off, 26)
    sizes = np.diff(off).astype(np.uint32)
    want, _ = ok.index_dist32(counts, 26, postings, sizes, h, off, 1, 0, 20, 0.05, threads=4)
    monkeypatch.setenv("RK_INDEX_KEYS_CAP_PCT", "60")   # (buffers for 60 % of a range's expected keys)
    monkeypatch.setenv("RK_INDEX_PASS_BITS", "2")
    c = capi.Context(0)
    sk = c.sketches_from_host(h, off)
    idx = c.index_build(sk, 26)
    assert idx.built_fast and idx.products == 6 and idx.total == len(h)
    p2, c2 = idx.export()
    assert np.array_equal(p2, postings) and np.array_equal(c2, counts)
    assert_hits_equal(c.dist_rows(idx, None, 1, 0, 20, 0.05)[0], want)
    monkeypatch.delenv("RK_INDEX_PASS_BITS")
    parts = [c.index_build_shard(sk, 26, r, 4) for r in range(4)]
    assert sum(p.total for p in parts) == len(h)
    assert np.array_equal(np.concatenate([p.export(want_counts=False)[0][: p.total] for p in parts]), postings)
    del idx, parts
    c.close()


@pytest.mark.parametrize("bits,n,m,strains", [(26, 6000, 150, 3000), (20, 5000, 250, 2500), (24, 4500, 400, 1500)])
def test_buckets_beyond_the_lds_sort(monkeypatch, bits, n, m, strains):
    # a hash shared by thousands of genomes (a species of 1,500 - 3,000 strains) is a posting list that overflows a bucket of the
    # in-LDS sort on its own: k_bucket_heavy builds such buckets slab by slab, single heavy lists through a bitmap over the genomes
    # (26 / 24 bits: the sub-buckets are split by their remaining hash bits first; 20 bits: a sub-bucket IS one hash value).
    # Until round 5 one such bucket sent the whole build to the device-wide sort (RK_INDEX_NO_HEAVY=1 still does).
    names, h, off = synth.clade_sketches(n, m, bits, strains_per_clade=strains, seed=900 + strains)
    postings, counts = ok.index_build32(h, off, bits)
    sizes = np.diff(off).astype(np.uint32)
    assert counts.max() > 1000
    monkeypatch.setenv("RK_INDEX_TILES", "1")
    c = capi.Context(0)
    monkeypatch.setenv("RK_INDEX_NO_HEAVY", "1")
    g = capi.Context(0)
    sk = c.sketches_from_host(h, off)
    idx = c.index_build(sk, bits)
    old = g.index_build(g.sketches_from_host(h, off), bits)
    assert idx.built_fast and idx.products == 6 and not old.built_fast
    p2, c2 = idx.export()
    assert np.array_equal(p2, postings) and np.array_equal(c2, counts)
    assert np.array_equal(idx.order, old.order)
    for metric, D in ((0, 0.05), (1, 0.02), (0, 1.0)):
        want, _ = ok.index_dist32(counts, bits, postings, sizes, h, off, 1, metric, 20, D, threads=8)
        assert len(want) > 10000
        assert_hits_equal(c.dist_rows(idx, None, 1, metric, 20, D)[0], want)
    assert_hits_equal(g.dist_rows(old, None, 1, 0, 20, 0.05)[0], ok.index_dist32(counts, bits, postings, sizes, h, off, 1, 0, 20, 0.05, threads=8)[0])
    # the same buckets in a sharded build
    parts = [c.index_build_shard(sk, bits, r, 4) for r in range(4)]
    assert np.array_equal(np.concatenate([p.export(want_counts=False)[0][: p.total] for p in parts]), postings)
    assert sum(sum(p.shard_records(4)) for p in parts) == idx.self_stats[3]
    del idx, old, parts
    c.close()
    g.close()


def test_tile_records_that_do_not_fit_fall_back_to_slice_records(monkeypatch):
    # the build's unsorted tile records have a fixed capacity (H / 2 + 64 K); a collection whose lists scatter over many blocks
    # overflows it: the index is then built with slice records after all (RK_TILE_REC_CAP forces it), same results; and the
    # lazy builder has a budget too (RK_TILE_BUDGET): beyond it the self join stays with the row kernels
    names, h, off = synth.clade_sketches(2000, 300, 26, seed=17)
    postings, counts = ok.index_build32(h, off, 26)
    sizes = np.diff(off).astype(np.uint32)
    want, _ = ok.index_dist32(counts, 26, postings, sizes, h, off, 1, 0, 20, 0.05, threads=4)
    monkeypatch.setenv("RK_INDEX_TILES", "1")
    monkeypatch.setenv("RK_TILE_REC_CAP", "4096")
    c = capi.Context(0)
    idx = c.index_build(c.sketches_from_host(h, off), 26)
    assert idx.products == 1 and idx.built_fast
    p2, c2 = idx.export()
    assert np.array_equal(p2, postings) and np.array_equal(c2, counts)
    assert_hits_equal(c.dist_rows(idx, None, 1, 0, 20, 0.05)[0], want)
    del idx
    c.close()
    monkeypatch.delenv("RK_TILE_REC_CAP")
    monkeypatch.setenv("RK_INDEX_TILES", "0")
    monkeypatch.setenv("RK_DIST_TILES", "1")
    monkeypatch.setenv("RK_TILE_BUDGET", "1000")
    c = capi.Context(0)
    idx = c.index_build(c.sketches_from_host(h, off), 26)
    assert_hits_equal(c.dist_rows(idx, None, 1, 0, 20, 0.05)[0], want)
    assert idx.products == 1 and not c.dist_kernel_name(idx, None, 1, 0, 20, 0.05).startswith("rk_tile_kernel")
    del idx
    c.close()


def test_self_join_moves_to_the_tile_kernel_after_a_fallback(monkeypatch):
    # default switches: clades of 36 -- only the first three rows of a clade have relatives beyond the near-window kernel's 32
    # columns, so few slice records are wide ones (8 %: the index does not count as one with wide clusters) -- yet those rows
    # fall back.  The first launches run on rk_near_kernel with its exact fallback; once a completed launch has shown the
    # fallback list to be non-empty, later launches over this index take the tile kernel.  Same hits every time.
    names, h, off = synth.clade_sketches(1296, 500, 26, strains_per_clade=36, seed=136)
    order = np.arange(len(names))   # (the species tree of synth for clades > 10: 36 strains = 4 sub-lineages, all within -D 0.08)
    names, h, off = synth.permute_genomes(names, h, off, order)
    postings, counts = ok.index_build32(h, off, 26)
    sizes = np.diff(off).astype(np.uint32)
    want, _ = ok.index_dist32(counts, 26, postings, sizes, h, off, 1, 0, 20, 0.08, threads=4)
    c = capi.Context(0)
    idx = c.index_build(c.sketches_from_host(h, off), 26)
    seen = []
    for _ in range(4):
        seen.append(c.dist_kernel_name(idx, None, 1, 0, 20, 0.08).split("<")[0])
        assert_hits_equal(c.dist_rows(idx, None, 1, 0, 20, 0.08)[0], want)
    assert seen[0] == "rk_near_kernel" and seen[-1] == "rk_tile_kernel", seen
    assert sorted(seen, key=lambda k: k == "rk_tile_kernel") == seen     # once on the tile kernel, it stays there
    del idx
    c.close()


def test_index_without_slice_records(monkeypatch):
    # an index of 2^31 postings and more is built without slice records (their posting offsets would collide with the tag bit
    # of the compact form); RK_INDEX_NO_SELF=1 builds a small one that way: .dict / .index content, explicit queries and the
    # sparse self join (tile kernel: it reads the posting lists themselves) are the oracle's, a dense self join is refused
    names, h, off = synth.clade_sketches(1500, 200, 26, strains_per_clade=40, seed=77, tiny=2)
    postings, counts = ok.index_build32(h, off, 26)
    sizes = np.diff(off).astype(np.uint32)
    monkeypatch.setenv("RK_INDEX_NO_SELF", "1")
    c = capi.Context(0)
    monkeypatch.delenv("RK_INDEX_NO_SELF")
    idx = c.index_build(c.sketches_from_host(h, off), 26)
    p2, c2 = idx.export()
    assert np.array_equal(p2, postings) and np.array_equal(c2, counts)
    assert idx.self_stats[0] == 0
    for metric, D in ((0, 0.05), (1, 0.08)):
        want, _ = ok.index_dist32(counts, 26, postings, sizes, h, off, 1, metric, 20, D, threads=4)
        assert c.dist_kernel_name(idx, None, 1, metric, 20, D).startswith("rk_tile_kernel")
        assert_hits_equal(c.dist_rows(idx, None, 1, metric, 20, D)[0], want)
        assert c.dist_kernel_name(idx, None, 1, metric, 20, D).startswith("rk_tile_kernel<")
    with pytest.raises(Exception, match="no slice records"):
        c.dist_rows(idx, None, 1, 0, 20, 1.5)
    q = c.sketches_from_host(h[: int(off[100])], off[:101])
    want, _ = ok.index_dist32(counts, 26, postings, sizes, h[: int(off[100])], off[:101], 0, 0, 20, 0.1, threads=4)
    assert_hits_equal(c.dist_rows(idx, q, 0, 0, 20, 0.1)[0], want)
    del idx, q
    c.close()


def test_pair_kernel_equals_single_row_kernel(ctx, monkeypatch):
    # the same self join through the pair kernel (default) and the single-row kernel (RK_DIST_PAIR=2; the developer
    # switches are read when a context is created)
    names, h, off = synth.clade_sketches(777, 300, 24, seed=23)
    monkeypatch.setenv("RK_DIST_NEAR", "0")     # the kernels with full counter rows (the near-window kernel's fallback)
    ctx0 = capi.Context(0)
    monkeypatch.setenv("RK_DIST_PAIR", "2")
    ctx1 = capi.Context(0)
    monkeypatch.delenv("RK_DIST_PAIR")
    monkeypatch.delenv("RK_DIST_NEAR")
    ctx = ctx0
    idx = ctx.index_build(ctx.sketches_from_host(h, off), 24)
    idx1 = ctx1.index_build(ctx1.sketches_from_host(h, off), 24)
    assert ", 2, " in ctx.dist_kernel_name(idx, None, 1, 0, 20, 0.05) and ", 1, " in ctx1.dist_kernel_name(idx1, None, 1, 0, 20, 0.05)
    postings, counts = ok.index_build32(h, off, 24)
    sizes = np.diff(off).astype(np.uint32)
    for metric, D in ((0, 0.05), (1, 0.02), (0, 0.5)):
        want, _ = ok.index_dist32(counts, 24, postings, sizes, h, off, 1, metric, 20, D, threads=4)
        pair, _ = ctx.dist_rows(idx, None, 1, metric, 20, D)
        single, _ = ctx1.dist_rows(idx1, None, 1, metric, 20, D)
        assert_hits_equal(pair, want)
        assert_hits_equal(single, want)
    del idx1
    ctx1.close()


@pytest.mark.parametrize("n,lds_kb,world", [(12000, 0, 1), (8000, 24, 1), (12000, 0, 3), (9000, 20, 2)])
def test_self_join_in_bands(monkeypatch, n, lds_kb, world):
    """Row i only counts in the columns behind it: the rows are cut into bands whose LDS rows start at the band's first
    row (tiled -> single rows -> row pairs as the rows get shorter).  Developer switches bring the band boundaries
    down to test sizes; the result must equal the oracle's and the one-launch path's, also per shard."""
    names, h, off = synth.clade_sketches(n, 40, 24, seed=37)
    monkeypatch.setenv("RK_DIST_NEAR", "0")     # the band planner belongs to the kernel with full counter rows
    monkeypatch.setenv("RK_DIST_BAND_MIN_ROWS", "256")
    if lds_kb:
        monkeypatch.setenv("RK_DIST_LDS_KB", str(lds_kb))
    banded = capi.Context(0)
    monkeypatch.setenv("RK_DIST_BANDS", "0")
    plain = capi.Context(0)
    for v in ("RK_DIST_BAND_MIN_ROWS", "RK_DIST_BANDS", "RK_DIST_NEAR"):
        monkeypatch.delenv(v)
    ib = banded.index_build(banded.sketches_from_host(h, off), 24)
    ip = plain.index_build(plain.sketches_from_host(h, off), 24)
    name_b, name_p = banded.dist_kernel_name(ib, None, 1, 0, 20, 0.05), plain.dist_kernel_name(ip, None, 1, 0, 20, 0.05)
    assert "bands]" in name_b and "bands]" not in name_p, (name_b, name_p)
    postings, counts = ok.index_build32(h, off, 24)
    sizes = np.diff(off).astype(np.uint32)
    for metric, D in ((0, 0.05), (1, 0.2)):
        want, _ = ok.index_dist32(counts, 24, postings, sizes, h, off, 1, metric, 20, D, threads=8)
        assert len(want) > n
        if world == 1:
            assert_hits_equal(banded.dist_rows(ib, None, 1, metric, 20, D)[0], want)
            assert_hits_equal(plain.dist_rows(ip, None, 1, metric, 20, D)[0], want)
        else:
            parts = [banded.dist_rows(ib, None, 1, metric, 20, D, row_first=r, row_step=world, row_block=16)[0] for r in range(world)]
            for r, p in enumerate(parts):
                assert np.all(ib.shard_of(p, world, 16) == r)
            merged = np.concatenate(parts)
            assert_hits_equal(merged[np.lexsort((merged["col"], merged["row"]))], want)
    del ib, ip
    banded.close()
    plain.close()


def test_self_join_of_big_sketches_uses_32bit_counters_in_pairs(ctx):
    # sketches of >= 65536 hashes (3 Gb genomes): the self join runs the pair kernel with u32 LDS counters
    names, h, off = synth.clade_sketches(41, 70000, 26, seed=31)
    assert np.diff(off).max() >= 65536
    idx = ctx.index_build(ctx.sketches_from_host(h, off), 26)
    postings, counts = ok.index_build32(h, off, 26)
    sizes = np.diff(off).astype(np.uint32)
    for metric, D in ((0, 0.05), (1, 0.3), (0, 1.5)):
        want, _ = ok.index_dist32(counts, 26, postings, sizes, h, off, 1, metric, 20, D, threads=4)
        mine, _ = ctx.dist_rows(idx, None, 1, metric, 20, D)
        assert len(want) > 0
        assert_hits_equal(mine, want)


def test_row_sharding_union_equals_full(ctx):
    names, h, off = synth.clade_sketches(500, 100, 22, seed=9)
    idx = ctx.index_build(ctx.sketches_from_host(h, off), 22)
    full, _ = ctx.dist_rows(idx, None, 1, 0, 20, 0.08)
    parts = [ctx.dist_rows(idx, None, 1, 0, 20, 0.08, row_first=r, row_step=4)[0] for r in range(4)]
    for r, p in enumerate(parts):
        assert np.all(idx.shard_of(p, 4) == r)
    merged = np.concatenate(parts)
    merged = merged[np.lexsort((merged["col"], merged["row"]))]
    assert merged.tobytes() == full.tobytes()


def test_ref_vs_query_random_and_tiled_reference(ctx):
    # 45,000 references: the LDS counter row is tiled (2 tiles of <= 40,960 columns)
    rn, rh, roff = synth.clade_sketches(45000, 24, 24, seed=4)
    qn, qh, qoff = synth.clade_sketches(40, 300, 24, seed=5)
    qh[: 24] = rh[: 24]  # make query 0 contain reference 0
    # keep every per-genome set sorted/unique after the overwrite
    parts = [np.unique(qh[int(qoff[i]):int(qoff[i + 1])]) for i in range(40)]
    qoff = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
    qh = np.concatenate(parts)
    postings, counts = ok.index_build32(rh, roff, 24)
    sizes = np.diff(roff).astype(np.uint32)
    idx = ctx.index_build(ctx.sketches_from_host(rh, roff), 24)
    qs = ctx.sketches_from_host(qh, qoff)
    for metric, D in ((0, 0.3), (1, 0.2)):
        want, _ = ok.index_dist32(counts, 24, postings, sizes, qh, qoff, 0, metric, 20, D, threads=4)
        mine, _ = ctx.dist_rows(idx, qs, 0, metric, 20, D)
        assert len(want) > 0
        assert_hits_equal(mine, want)


def test_sliced_membership_pass_equals_the_fused_kernel(monkeypatch):
    # RK_DISTQ_SLICED=1: the present hashes of every query are found by k_member_sliced (the rank bitmap in 48 KiB slices through
    # LDS, a cursor per query) and counted by the kernel's pre-resolved variant; big queries (several slices and slice ranges,
    # planted references), tiny and empty ones, a tiled reference row, both metrics, the dense counter matrix
    rn, rh, roff = synth.clade_sketches(45000, 24, 24, seed=4)
    rng = np.random.default_rng(21)
    parts = []
    for q in range(37):
        size = [40000, 9000, 300, 1, 0][q % 5]
        p = rng.integers(0, 1 << 24, size=size, dtype=np.uint64).astype(np.uint32)
        if q % 3 == 0 and size:
            refs = rng.choice(45000, size=4, replace=False)
            p = np.concatenate([p] + [rh[int(roff[r]):int(roff[r + 1])] for r in refs])
        parts.append(np.unique(p))
    qoff = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
    qh = np.concatenate(parts).astype(np.uint32)
    postings, counts = ok.index_build32(rh, roff, 24)
    sizes = np.diff(roff).astype(np.uint32)
    monkeypatch.setenv("RK_DISTQ_SLICED", "1")
    c = capi.Context(0)
    idx = c.index_build(c.sketches_from_host(rh, roff), 24)
    qs = c.sketches_from_host(qh, qoff)
    assert c.dist_kernel_name(idx, qs, 0, 1, 20, 0.2).startswith("rk_distq_kernel<8, 3,")   # (references of 24 hashes: 8-bit counters)
    for metric, D in ((1, 0.2), (0, 0.4)):
        want, _ = ok.index_dist32(counts, 24, postings, sizes, qh, qoff, 0, metric, 20, D, threads=8)
        assert len(want) >= 12 * 4
        for _ in range(2):   # (the records' scratch is kept with the queries: a second pass reuses it)
            assert_hits_equal(c.dist_rows(idx, qs, 0, metric, 20, D)[0], want)
    few = c.sketches_from_host(qh[: int(qoff[4])], qoff[:5])
    want, wdense = ok.index_dist32(counts, 24, postings, sizes, qh[: int(qoff[4])], qoff[:5], 0, 1, 20, 0.2, want_dense=True, threads=8)
    mine, dense = c.dist_rows(idx, few, 0, 1, 20, 0.2, want_dense=True)
    assert np.array_equal(dense, wdense)
    assert_hits_equal(mine, want)
    del idx
    c.close()


def test_large_query_sketch_uses_32bit_counters(ctx):
    # a query with >= 65536 hashes (3 Gb genome scale) forces the u32 LDS counter row;
    # the same data through the u16 path (small queries) must agree on the shared rows
    bits = 22
    rn, rh, roff = synth.clade_sketches(2000, 500, bits, seed=12)
    rng = np.random.default_rng(13)
    big = np.unique(np.concatenate([rng.integers(0, 1 << bits, size=90000, dtype=np.uint64).astype(np.uint32),
                                    rh[: int(roff[40])]]))
    small = rh[int(roff[7]):int(roff[8])]
    qh = np.concatenate([big, small])
    qoff = np.array([0, len(big), len(big) + len(small)], dtype=np.uint64)
    assert len(big) >= 65536
    postings, counts = ok.index_build32(rh, roff, bits)
    sizes = np.diff(roff).astype(np.uint32)
    idx = ctx.index_build(ctx.sketches_from_host(rh, roff), bits)
    want, wdense = ok.index_dist32(counts, bits, postings, sizes, qh, qoff, 0, 1, 20, 0.2, want_dense=True)
    mine, dense = ctx.dist_rows(idx, ctx.sketches_from_host(qh, qoff), 0, 1, 20, 0.2, want_dense=True)
    assert np.array_equal(dense, wdense)
    assert wdense.max() >= 500
    assert_hits_equal(mine, want)
    # small query alone -> u16 path
    q2 = ctx.sketches_from_host(small, np.array([0, len(small)], dtype=np.uint64))
    mine2, dense2 = ctx.dist_rows(idx, q2, 0, 1, 20, 0.2, want_dense=True)
    assert np.array_equal(dense2[0], wdense[1])


def test_edge_cases_empty_inputs(ctx):
    off = np.array([0, 0, 3, 3], dtype=np.uint64)
    h = np.array([1, 5, 9], dtype=np.uint32)
    sk = ctx.sketches_from_host(h, off)
    idx = ctx.index_build(sk, 12)
    hits, dense = ctx.dist_rows(idx, sk, 1, 0, 20, 2.0, want_dense=True)
    assert dense.tolist() == [[0, 0, 0], [0, 3, 0], [0, 0, 0]]
    assert len(hits) == 3 and np.all(hits["dist"] == 1.0) and np.all(hits["jorc"] == 0.0)
    empty = ctx.sketches_from_host(np.zeros(0, dtype=np.uint32), np.zeros(1, dtype=np.uint64))
    idx0 = ctx.index_build(empty, 12)
    hits, _ = ctx.dist_rows(idx0, None, 1, 0, 20, 0.5)
    assert len(hits) == 0 and idx0.total == 0 and idx0.distinct == 0
    # query hash outside the reference hash space is ignored
    q = ctx.sketches_from_host(np.array([1, 5, 9, 1 << 20], dtype=np.uint32), np.array([0, 4], dtype=np.uint64))
    hits, dense = ctx.dist_rows(idx, q, 0, 0, 20, 0.9, want_dense=True)
    assert dense.tolist() == [[0, 3, 0]] and len(hits) == 1 and hits[0]["common"] == 3


def test_device_sketches_with_a_broken_offset_table_are_refused(ctx):
    """rk_sketches_from_dev: offsets that are no CSR table (what a caller gets who hands over arrays another stream is still
    writing) are an argument error -- the kernels never walk them"""
    h = torch.arange(64, dtype=torch.int32, device="cuda")
    for bad in ([0, 40, 20, 64], [8, 16, 32, 64]):
        off = torch.tensor(bad, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        with pytest.raises(capi.RkError):
            ctx.sketches_from_dev(h.data_ptr(), off.data_ptr(), 3)
    off = torch.tensor([0, 20, 40, 64], dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    sk = ctx.sketches_from_dev(h.data_ptr(), off.data_ptr(), 3)
    assert sk.count == 3 and sk.total == 64


# ------------------------------------------------------------------ sketching
def sketch_case(ctx, k, s, l, genomes):
    """genomes: list of (seq uint8, rec_off) -> GPU CSR vs oracle sets"""
    param = ok.init_param(k, s, l)
    table = ok.shuffle_table(k, s, l)
    flt = ctx.filter(capi.params_init(k, s, l), table)
    seqs, rec_off, genome_rec = [], [0], [0]
    for seq, off in genomes:
        seqs.append(seq)
        for r in range(len(off) - 1):
            rec_off.append(rec_off[-1] + int(off[r + 1] - off[r]))
        genome_rec.append(len(rec_off) - 1)
    seq = np.concatenate(seqs) if seqs else np.zeros(0, dtype=np.uint8)
    sk = ctx.sketch_batch(flt, seq, np.array(rec_off, dtype=np.uint64), np.array(genome_rec, dtype=np.uint64))
    gh, goff = sk.download()
    assert sk.count == len(genomes)
    windows = 0
    for g, (gs, off) in enumerate(genomes):
        want = ok.sketch_records(param, table, gs, off)
        mine = gh[int(goff[g]):int(goff[g + 1])]
        assert np.array_equal(mine.astype(np.uint64), want), "genome %d" % g
        windows += ok.count_windows(param, gs, off)
    assert sk.windows == windows
    return sk


def test_sketch_golden_fixtures(ctx):
    d = os.path.join(GOLDEN, "sketch")
    exp = json.load(open(os.path.join(d, "expected.json")))
    genomes, want = [], []
    for fn, e in sorted(exp["files"].items()):
        genomes.append(ok.read_fasta(os.path.join(d, fn)))
        want.append(e["hashes"])
    sk = sketch_case(ctx, exp["half_k"], exp["half_subk"], exp["drlevel"], genomes)
    gh, goff = sk.download()
    for g, w in enumerate(want):
        assert gh[int(goff[g]):int(goff[g + 1])].tolist() == w


@pytest.mark.parametrize("k,s,l", [(10, 6, 3), (8, 5, 2), (9, 5, 2), (6, 4, 1), (10, 7, 4), (12, 6, 4)])
def test_sketch_parameter_sets_vs_oracle(ctx, k, s, l):
    if s - l < 3:
        with pytest.raises(capi.RkError):
            capi.params_init(k, s, l)
        return
    rng = np.random.default_rng(k * 100 + s * 10 + l)
    genomes = []
    for g in range(5):
        n = [200000, 70001, 1024, 2048 + 19, 5][g]
        b = synth.clade_genome(g // 2, g % 2, n).copy()
        if n > 5000:
            for p in rng.integers(0, n - 50, size=6):
                b[p:p + int(rng.integers(1, 40))] = ord("N")
            b[100:200] = np.frombuffer(bytes(b[100:200]).lower(), dtype=np.uint8)
        cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n, size=3)]))
        genomes.append((b, np.array(cuts, dtype=np.uint64)))
    genomes.append((np.zeros(0, dtype=np.uint8), np.array([0], dtype=np.uint64)))      # no records
    genomes.append((np.frombuffer(b"ACGT", dtype=np.uint8), np.array([0, 0, 4], dtype=np.uint64)))
    sketch_case(ctx, k, s, l, genomes)


def test_sketch_low_complexity_overflow_retry(ctx):
    # a homopolymer / tandem repeat emits the same hash for every window if it passes
    # the filter: exercises the candidate-overflow retry path and the dedup
    k, s, l = 8, 5, 2
    table = ok.shuffle_table(k, s, l)
    param = ok.init_param(k, s, l)
    # find a 2-base repeat unit whose dim_id passes the filter
    best = None
    for unit in (b"AC", b"AG", b"AT", b"CA", b"CG", b"GA", b"TA", b"AA", b"CC"):
        seq = np.frombuffer(unit * 400, dtype=np.uint8)
        if len(ok.sketch_records(param, table, seq, np.array([0, len(seq)], dtype=np.uint64))):
            best = unit
            break
    genomes = [(np.frombuffer((best or b"AC") * 300000, dtype=np.uint8), np.array([0, 600000], dtype=np.uint64)),
               (synth.clade_genome(1, 0, 50000), np.array([0, 50000], dtype=np.uint64))]
    sketch_case(ctx, k, s, l, genomes)


def test_end_to_end_sketch_index_alldist(ctx):
    """config[0]-shaped plumbing at reduced genome length: FASTA -> sketch -> index -> alldist."""
    k, s, l = 10, 6, 3
    param, table = ok.init_param(k, s, l), ok.shuffle_table(k, s, l)
    flt = ctx.filter(capi.params_init(k, s, l), table)
    gs = synth.clade_genome_set(24, 300000)
    seq = np.concatenate([b for _, b in gs])
    rec_off = np.arange(25, dtype=np.uint64) * 300000
    sk = ctx.sketch_batch(flt, seq, rec_off, np.arange(25, dtype=np.uint64))
    gh, goff = sk.download()
    want_parts = [ok.sketch_records(param, table, b, np.array([0, len(b)], dtype=np.uint64)) for _, b in gs]
    want_h = np.concatenate(want_parts).astype(np.uint32)
    want_off = np.concatenate([[0], np.cumsum([len(p) for p in want_parts])]).astype(np.uint64)
    assert np.array_equal(gh, want_h) and np.array_equal(goff, want_off)
    idx = ctx.index_build(sk, 28)
    postings, counts = ok.index_build32(want_h, want_off, 28)
    p2, _ = idx.export(want_counts=False)
    assert np.array_equal(p2, postings)
    want, _ = ok.index_dist32(counts, 28, postings, np.diff(want_off).astype(np.uint32), want_h, want_off,
                              1, 0, 20, 0.05)
    mine, _ = ctx.dist_rows(idx, None, 1, 0, 20, 0.05)
    assert len(want) >= 45 * 2  # two full clades of 10 strains
    assert_hits_equal(mine, want)


# ------------------------------------------------------------------ FASTQ (f1)
def make_fastq(seed, n_reads=400, read_len=150, genome_len=20000):
    """reads sampled (with repeats -> occurrence counts > 1) from one genome, random qualities"""
    rng = np.random.default_rng(seed)
    g = synth.clade_genome(seed, 0, genome_len)
    out = []
    for r in range(n_reads):
        p = int(rng.integers(0, genome_len - read_len))
        seq = g[p:p + read_len].copy()
        if r % 17 == 0:
            seq[int(rng.integers(0, read_len))] = ord("N")
        q = rng.integers(33, 74, size=read_len).astype(np.uint8)
        out.append(b"@r%d extra\n" % r + seq.tobytes() + b"\n+\n" + q.tobytes() + b"\n")
    return b"".join(out)


@pytest.mark.parametrize("least_qual,least_num", [(0, 1), (40, 1), (0, 2), (45, 3), (127, 1)])
def test_fastq_quality_gate_and_occurrence_count(ctx, least_qual, least_num):
    k, s, l = 8, 5, 2
    param, table = ok.init_param(k, s, l), ok.shuffle_table(k, s, l)
    flt = ctx.filter(capi.params_init(k, s, l), table)
    files = [make_fastq(seed) for seed in (1, 2, 3)]
    seqs, quals, rec_off, genome_rec, wants = [], [], [0], [0], []
    for data in files:
        sq, ql, off = ok.parse_fastq_bytes(data)
        wants.append(ok.sketch_records_fastq(param, table, sq, ql, off, least_qual, least_num))
        seqs.append(sq)
        quals.append(ql)
        rec_off.extend((rec_off[-1] + off[1:]).tolist())
        genome_rec.append(len(rec_off) - 1)
    sk = ctx.sketch_batch_fastq(flt, np.concatenate(seqs), np.concatenate(quals), np.array(rec_off, dtype=np.uint64),
                                np.array(genome_rec, dtype=np.uint64), least_qual, least_num)
    gh, goff = sk.download()
    for g, want in enumerate(wants):
        assert np.array_equal(gh[int(goff[g]):int(goff[g + 1])].astype(np.uint64), want), (g, least_qual, least_num)
    if least_qual == 0 and least_num == 1:
        assert sum(len(w) for w in wants) > 100
    if least_qual == 127:
        assert sk.total == 0


def test_steady_state_calls_do_not_touch_the_driver_allocator():
    """the context's pool: after a warm-up pass, upload -> index build -> self join -> ref-vs-query -> frees repeat
    without a single hipMalloc / hipFree (rk_ctx_pool_stats), and rk_ctx_trim gives the cache back"""
    c = capi.Context(0)
    names, h, off = synth.clade_sketches(3000, 300, 26, seed=3)
    q_off = (off[:101] - off[0]).astype(np.uint64)

    def one_pass():
        sk = c.sketches_from_host(h, off)
        idx = c.index_build(sk, 26)
        a, _ = c.dist_rows(idx, None, 1, 0, 20, 0.05)
        qs = c.sketches_from_host(h[:int(off[100])], q_off)
        b, _ = c.dist_rows(idx, qs, 0, 0, 20, 0.05)
        for o in (qs, idx, sk):
            o.close()
        return len(a), len(b)

    first = one_pass()
    one_pass()
    before = c.pool_stats()
    for _ in range(3):
        assert one_pass() == first
    after = c.pool_stats()
    assert after[2] == before[2] and after[3] == before[3], (before, after)
    assert after[1] == after[0] > 0          # everything is back in the cache
    c.trim()
    assert c.pool_stats()[:2] == (0, 0)
    c.close()
