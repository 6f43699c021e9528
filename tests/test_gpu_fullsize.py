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
    assert np.array_equal(mine["dist"], want["dist"])   # rk_dist_rows: the host libm has the last word


def test_config2_alldist_10k_exact(ctx):
    """configs[2]: 10,000 sketches, L3K10, alldist -D 0.05 -- every hit identical to the oracle."""
    import torch
    names, h, off = synth.clade_sketches(10000, 1220, 28)
    idx = ctx.index_build(ctx.sketches_from_host(h, off), 28)
    assert (idx.total, idx.distinct, idx.sum_sq) == (12199994, 3153593, 90120058)
    # (round 5) the tile records come with the build: the FIRST join over the index runs on the headline's kernel, and it is this
    # launch -- metric 0, -D 0.05, 10,000 genomes -- that is compared with the oracle below
    assert idx.products == 6 and ctx.dist_kernel_name(idx, None, 1, 0, 20, 0.05).startswith("rk_tile_kernel<512u")
    mine, _ = ctx.dist_rows(idx, None, 1, 0, 20, 0.05)
    postings, counts = ok.index_build32(h, off, 28)
    assert np.array_equal(idx.export(want_counts=False)[0], postings)
    want, _ = ok.index_dist32(counts, 28, postings, np.diff(off).astype(np.uint32), h, off, 1, 0, 20, 0.05,
                              threads=CORES)
    assert len(want) == 45000          # 45 pairs per 10-strain clade
    check_hits(mine, want)
    want_j = want
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
    assert ctx.dist_kernel_name(idx, None, 1, 0, 20, 0.05).startswith("rk_tile_kernel<512u")
    check_hits(full, want_j)
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
    # (the hits of the device-resident API carry the device's own `log`: north_star's 1e-12; the synchronous API's are
    # recomputed with the host libm and equal the oracle's bit for bit, checked above)
    assert len(dev) == len(full)
    for f in ("row", "col", "common", "size0", "size1", "jorc"):
        assert np.array_equal(dev[f], full[f]), f
    assert np.max(np.abs(dev["dist"] - full["dist"])) <= 1e-12


def dev_hits(ctx, idx, triangle, metric, kmer, D, queries=None, cap=1 << 19, **shard):
    """hits of the asynchronous all-in-HBM entry point (what bench.py times), read back and ordered by (row, col)"""
    import torch
    buf = torch.empty(cap * capi.HIT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    ctx.dist_rows_dev(idx, triangle, metric, kmer, D, buf.data_ptr(), cap, cnt.data_ptr(),
                      stream=torch.cuda.current_stream().cuda_stream, queries=queries, **shard)
    torch.cuda.synchronize()
    n = int(cnt.item())
    assert n <= cap
    dev = np.frombuffer(buf.cpu().numpy().tobytes()[: n * capi.HIT_DTYPE.itemsize], dtype=capi.HIT_DTYPE)
    return dev[np.lexsort((dev["col"], dev["row"]))]


def check_dev_hits(dev, want):
    # (the device-resident API keeps the device's own `log`: north_star's 1e-12; everything else is exact)
    assert len(dev) == len(want)
    for f in ("row", "col", "common", "size0", "size1", "jorc"):
        assert np.array_equal(dev[f], want[f]), f
    assert np.max(np.abs(dev["dist"] - want["dist"]), initial=0.0) <= 1e-12


def test_config3_alldist_50k_exact(ctx):
    """configs[3] (one rank's view): 50,000 sketches -> 1.25e9 pairs, 225,000 hits -- every hit identical to the oracle's
    (all host cores), through the synchronous API and through rk_dist_rows_dev for the banded whole launch and for one
    1/8 block-cyclic shard (what a GPU of the 8-GPU run computes)."""
    names, h, off = synth.clade_sketches(50000, 1220, 28)
    idx = ctx.index_build(ctx.sketches_from_host(h, off), 28)
    assert idx.built_fast and idx.products == 6 and ctx.dist_kernel_name(idx, None, 1, 0, 20, 0.05).startswith("rk_tile_kernel<")
    postings, counts = ok.index_build32(h, off, 28)
    want, _ = ok.index_dist32(counts, 28, postings, np.diff(off).astype(np.uint32), h, off, 1, 0, 20, 0.05, threads=CORES)
    del counts
    assert len(want) == 45 * 5000
    mine, _ = ctx.dist_rows(idx, None, 1, 0, 20, 0.05)
    check_hits(mine, want)
    assert np.array_equal(idx.export(want_counts=False)[0], postings)
    check_dev_hits(dev_hits(ctx, idx, 1, 0, 20, 0.05), want)
    # rows of one rank out of 8 are a subset with the same records
    part, _ = ctx.dist_rows(idx, None, 1, 0, 20, 0.05, row_first=3, row_step=8)
    assert part.tobytes() == mine[idx.shard_of(mine, 8) == 3].tobytes()
    # the block-cyclic shard the multi-GPU callers use (blocks of 16 rows)
    part, _ = ctx.dist_rows(idx, None, 1, 0, 20, 0.05, row_first=5, row_step=8, row_block=16)
    sel = idx.shard_of(mine, 8, 16) == 5
    assert part.tobytes() == mine[sel].tobytes()
    check_dev_hits(dev_hits(ctx, idx, 1, 0, 20, 0.05, row_first=5, row_step=8, row_block=16), want[sel])
    # the same collection in shuffled order: renumbered internally, same pairs (mapped back), same kernel variants
    order = synth.genome_order(50000, "shuffled", seed=11)
    _, h2, off2 = synth.permute_genomes(names, h, off, order)
    idx2 = ctx.index_build(ctx.sketches_from_host(h2, off2), 28)
    # (a fresh index and one that has been joined several times take the same kernel: the choice follows size and shape)
    assert ctx.dist_kernel_name(idx2, None, 1, 0, 20, 0.05) == ctx.dist_kernel_name(idx, None, 1, 0, 20, 0.05)
    assert ctx.dist_kernel_name(idx, None, 1, 0, 20, 0.05).startswith("rk_tile_kernel<")
    m2, _ = ctx.dist_rows(idx2, None, 1, 0, 20, 0.05)
    a, b = order[m2["row"].astype(np.int64)], order[m2["col"].astype(np.int64)]
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    key = np.lexsort((hi, lo))
    assert np.array_equal(lo[key], want["row"]) and np.array_equal(hi[key], want["col"])
    assert np.array_equal(m2["common"][key], want["common"]) and np.array_equal(m2["dist"][key], want["dist"])


def test_self_join_kernels_agree_at_full_size(monkeypatch):
    """size-independent property: on 30,000 sketches with species of 10 and of 100 strains mixed, a tiny sketch among them,
    the near-window kernel (with its fallback), the tile kernel in both variants (masks through LDS, row masks as 64-bit
    scalars) and row shards of either report the same pairs with the same counts and distances -- for thresholds and metrics
    the oracle would need minutes for at this size (the 10,000- and 50,000-genome tests above pin the oracle itself)."""
    n1, h1, o1 = synth.clade_sketches(20000, 1220, 28, seed=901)
    n2, h2, o2 = synth.clade_sketches(10000, 1220, 28, strains_per_clade=100, seed=902, tiny=1)
    h = np.concatenate([h1, h2])
    off = np.concatenate([o1, o2[1:] + o1[-1]]).astype(np.uint64)
    monkeypatch.setenv("RK_DIST_TILES", "0")
    monkeypatch.setenv("RK_INDEX_TILES", "0")   # (the build with slice records, as below 4,000 genomes)
    near = capi.Context(0)
    monkeypatch.delenv("RK_INDEX_TILES")
    monkeypatch.setenv("RK_DIST_TILES", "1")
    tiles = capi.Context(0)
    monkeypatch.delenv("RK_DIST_TILES")
    idx_n = near.index_build(near.sketches_from_host(h, off), 28)
    idx_t = tiles.index_build(tiles.sketches_from_host(h, off), 28)
    assert idx_n.products == 1 and idx_t.products == 6
    for metric, D in ((0, 0.05), (0, 0.12), (1, 0.03)):
        assert not near.dist_kernel_name(idx_n, None, 1, metric, 20, D).startswith("rk_tile_kernel")
        want = dev_hits(near, idx_n, 1, metric, 20, D, cap=1 << 22)
        assert len(want) > 500000
        for srow in ("0", "1"):
            monkeypatch.setenv("RK_TILE_SROW", srow)
            assert tiles.dist_kernel_name(idx_t, None, 1, metric, 20, D).startswith("rk_tile_kernel<")
            check_dev_hits(dev_hits(tiles, idx_t, 1, metric, 20, D, cap=1 << 22), want)
        monkeypatch.delenv("RK_TILE_SROW")
        parts = [dev_hits(tiles, idx_t, 1, metric, 20, D, cap=1 << 22, row_first=r, row_step=4, row_block=32) for r in range(4)]
        merged = np.concatenate(parts)
        check_dev_hits(merged[np.lexsort((merged["col"], merged["row"]))], want)
    del idx_n, idx_t
    near.close()
    tiles.close()


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
    sel = mine[idx.shard_of(mine, 3, 16) == 1]
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
        check_dev_hits(dev_hits(ctx, idx, 0, metric, 20, D, queries=qs), want)   # the fused query kernel, hits left in HBM


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


# ------------------------------------------------------------------ configs[1]: 1,000 x 5 Mb FASTA -> sketch -> alldist
def _make_clade(args):
    """worker: writes the 10 strains of one clade as FASTA files; returns [(path, oracle hash set or None)]"""
    clade, out_dir, length, oracle_strains = args
    anc = np.random.default_rng(1000 + clade).integers(0, 4, size=length, dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    param = table = None
    res = []
    for s in range(10):
        codes = anc
        if s:
            rng = np.random.default_rng(5000 + 10 * clade + s)
            mut = rng.random(length) < 0.002 * s
            shift = rng.integers(1, 4, size=length, dtype=np.uint8)
            codes = np.where(mut, (anc + shift) & 3, anc).astype(np.uint8)
        bases = acgt[codes]
        path = os.path.join(out_dir, "c%03d_s%d.fna" % (clade, s))
        with open(path, "wb") as f:
            f.write(synth.fasta_text("c%d_s%d" % (clade, s), bases))
        want = None
        if s in oracle_strains:
            if param is None:
                param, table = ok.init_param(10, 6, 3), ok.shuffle_table(10, 6, 3)
            want = ok.sketch_records(param, table, bases, np.array([0, length], dtype=np.uint64))
        res.append((path, want))
    return res


def test_config1_cli_1000x5mb_sketch_then_alldist(tmp_path):
    """configs[1] at full size through the host tool: 1,000 synthetic 5 Mb genomes (100 clades of 10 strains, 5 GB of
    FASTA written by a process pool) -> `rabbit_kssd sketch` -> `rabbit_kssd alldist -D 0.05`.  Every 20th genome's
    hash set equals the oracle's, all sizes are those of a 5 Mb genome at L3K10, exactly 45 hits per clade, and the
    `common` of sampled hits is the set intersection of the two sketches."""
    import multiprocessing as mp
    import shutil
    import subprocess
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rabbitkssd_amd", "rabbit_kssd")
    n_clades, length = 100, 5_000_000
    free = shutil.disk_usage(str(tmp_path)).free
    batch_clades = n_clades if free > 14 * (1 << 30) else 20   # all 5 GB at once if the disk allows, else 1 GB batches
    assert free > 3 * (1 << 30), "needs 3 GB of scratch space"
    shuf = tmp_path / "L3K10.shuf"
    subprocess.run([tool, "shuffle", "-k", "10", "-s", "6", "-l", "3", "-o", str(shuf)], check=True, capture_output=True)
    wants, parts = {}, []
    with mp.get_context("fork").Pool(min(16, CORES)) as pool:
        for b0 in range(0, n_clades, batch_clades):
            d = tmp_path / ("b%d" % b0)
            d.mkdir()
            # genome index g = 10 * clade + strain; every 20th genome = strain 0 of the even clades
            jobs = [(c, str(d), length, {0} if c % 2 == 0 else set()) for c in range(b0, min(n_clades, b0 + batch_clades))]
            files = []
            for res in pool.map(_make_clade, jobs):
                for path, want in res:
                    files.append(path)
                    if want is not None:
                        wants[path] = want
            lst = tmp_path / ("b%d.list" % b0)
            lst.write_text("".join(f + "\n" for f in files))
            out = tmp_path / ("b%d.sketch" % b0)
            p = subprocess.run([tool, "sketch", "-q", "-i", str(lst), "-L", str(shuf), "-o", str(out), "-t", str(min(16, CORES))],
                               capture_output=True)
            assert p.returncode == 0, p.stderr.decode()[-2000:]
            parts.append(str(out))
            shutil.rmtree(d)
    merged = parts[0]
    if len(parts) > 1:
        (tmp_path / "parts.list").write_text("".join(p + "\n" for p in parts))
        merged = str(tmp_path / "all.sketch")
        p = subprocess.run([tool, "merge", "-i", str(tmp_path / "parts.list"), "-o", merged], capture_output=True)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
    info, names, h, off = ok.read_sketches32(merged)
    assert (info.half_k, info.half_subk, info.drlevel, info.genomeNumber) == (10, 6, 3, 10 * n_clades)
    sizes = np.diff(off).astype(np.int64)
    # 5,000,000 / 4096 = 1,221 expected hashes, sd ~35: the extremes of 1,000 genomes lie ~3.5 sd out
    assert 1050 < sizes.min() and sizes.max() < 1400 and 1200 < sizes.mean() < 1245, (sizes.min(), sizes.max(), sizes.mean())
    assert len(wants) == 10 * n_clades // 20
    for g, name in enumerate(names):
        if name in wants:
            assert g % 20 == 0
            assert np.array_equal(h[int(off[g]):int(off[g + 1])].astype(np.uint64), wants[name]), name
    p = subprocess.run([tool, "alldist", "-i", merged, "-D", "0.05", "-o", "c1.out"], cwd=str(tmp_path), capture_output=True)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = (tmp_path / "c1.out").read_text().split("\n")
    assert lines[0] == " genome0\tgenome1\tcommon|size0|size1\tjaccard\tmashD"
    body = [x for x in lines[1:] if x]
    assert len(body) == 45 * n_clades
    index_of = {n: i for i, n in enumerate(names)}
    rng = np.random.default_rng(1)
    for t in rng.integers(0, len(body), size=300):
        a, b, triple, jac, dist = body[t].split("\t")
        j, i = index_of[a], index_of[b]            # alldist prints (sketch[j], sketch[i]) with j > i, sizes (size_i, size_j)
        assert j > i and i // 10 == j // 10
        common, s0, s1 = (int(x) for x in triple.split("|"))
        si, sj = h[int(off[i]):int(off[i + 1])], h[int(off[j]):int(off[j + 1])]
        assert (s0, s1) == (len(si), len(sj))
        assert common == len(np.intersect1d(si, sj, assume_unique=True))
        jj, dd = ok.distance(common, s0, s1, 0, 20)
        assert jac == "%f" % jj and dist == "%f" % dd


# ------------------------------------------------------------------ configs[4]: all 1,000 queries x 100,000 references
def test_config4_ref_vs_query_all_1000_queries(ctx):
    """configs[4] with every query: 1,000 queries of ~45,776 hashes (3 Gb genomes at L4K10) against 100,000 reference
    sketches of 76 hashes, 24-bit hashes.  50 sampled queries get reference sketches planted into them and are compared
    hit for hit with the oracle (containment -D 0.05 and jaccard -D 0.5); the other 950 share only chance hashes with any
    reference, so the whole result must be exactly the oracle's result on the sample."""
    rn, rh, roff = synth.clade_sketches(100000, 76, 24, seed=31)
    qn, qh, qoff = synth.clade_sketches(1000, 45776, 24, seed=32)
    rng = np.random.default_rng(4)
    sample = np.sort(rng.choice(1000, size=50, replace=False))
    parts = [qh[int(qoff[q]):int(qoff[q + 1])] for q in range(1000)]
    for q in sample:
        refs = rng.choice(100000, size=5, replace=False)
        planted = [rh[int(roff[r]):int(roff[r + 1])] for r in refs]
        parts[q] = np.unique(np.concatenate([parts[q]] + planted))
    qoff = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
    qh = np.concatenate(parts)
    idx = ctx.index_build(ctx.sketches_from_host(rh, roff), 24)
    qs = ctx.sketches_from_host(qh, qoff)
    assert ctx.dist_kernel_name(idx, qs, 0, 1, 20, 0.05) == "rk_distq_kernel<8, 0, true>"   # one 100 KB tile of 8-bit counters, the look-up pipelined over three batches
    postings, counts = ok.index_build32(rh, roff, 24)
    sizes = np.diff(roff).astype(np.uint32)
    s_off = np.concatenate([[0], np.cumsum([len(parts[q]) for q in sample])]).astype(np.uint64)
    s_h = np.concatenate([parts[q] for q in sample])
    for metric, D in ((1, 0.05), (0, 0.5)):
        want, _ = ok.index_dist32(counts, 24, postings, sizes, s_h, s_off, 0, metric, 20, D, threads=CORES)
        want = want.copy()
        want["row"] = sample[want["row"]]        # sample-local query numbers -> global
        mine, _ = ctx.dist_rows(idx, qs, 0, metric, 20, D)
        if metric == 1:
            assert len(want) >= 50 * 5
            check_hits(mine, want)               # nothing outside the sample may be reported
        else:
            # jaccard of a 76-hash reference inside a 45,776-hash query is at most 0.0017: distance 0.28 > 0.05 but < 0.5
            sel = mine[np.isin(mine["row"], sample)]
            check_hits(sel, want)
            rest = mine[~np.isin(mine["row"], sample)]
            for t in rng.integers(0, max(1, len(rest)), size=min(200, len(rest))):
                q, r = int(rest["row"][t]), int(rest["col"][t])
                a = rh[int(roff[r]):int(roff[r + 1])]
                assert rest["common"][t] == len(np.intersect1d(parts[q], a, assume_unique=True))
                assert (rest["size0"][t], rest["size1"][t]) == (len(a), len(parts[q]))
    # dense counter rows of three whole queries (one planted, two not) against brute-force set intersections
    three = [int(sample[0]), 1, 998]
    t_off = np.concatenate([[0], np.cumsum([len(parts[q]) for q in three])]).astype(np.uint64)
    t_qs = ctx.sketches_from_host(np.concatenate([parts[q] for q in three]), t_off)
    _, dense = ctx.dist_rows(idx, t_qs, 0, 0, 20, 0.05, want_dense=True)
    _, wdense = ok.index_dist32(counts, 24, postings, sizes, np.concatenate([parts[q] for q in three]), t_off, 0, 0, 20, 0.05,
                                want_dense=True)
    assert np.array_equal(dense, wdense)


def test_one_gigabase_genome_k10s7l4(ctx, tmp_path):
    """configs[4]'s queries are mammalian genomes at K10 S7 L4 (24-bit hashes, 1 / 65,536 of the windows).  One 1.1 Gb
    genome -- more distinct hashes (~16,800) than the per-genome LDS sort holds (16,384): the device-wide sort, the
    run-length pass and the select of rk_sketch.hip take over -- as ONE record, and 400 Mb of it as 61 records with N runs, through
    rk_sketch_packed_dev (resident in HBM) and through `rabbit_kssd sketch` (one big plain FASTA file, streamed in pieces):
    every hash set equals the oracle's restatement of src/sketch.cpp:487-530 on the same bytes."""
    import subprocess
    import torch
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rabbitkssd_amd", "rabbit_kssd")
    k, s, l = 10, 7, 4
    param, table = ok.init_param(k, s, l), ok.shuffle_table(k, s, l)
    n = 1_100_000_000
    rng = np.random.default_rng(20261004)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n, dtype=np.uint8)]
    one = np.array([0, n], dtype=np.uint64)
    want_one = ok.sketch_records(param, table, bases, one)
    assert len(want_one) > 16384
    # the first 400 Mb as 61 records of uneven length, with an N run inside every fifth
    nm = 400_000_000
    cuts = np.unique(np.concatenate([[0, nm], rng.integers(1, nm, 60)])).astype(np.uint64)
    multi = bases[:nm].copy()
    for i in range(0, len(cuts) - 1, 5):
        a = int(cuts[i]) + 1000
        multi[a:a + 50 + 7 * i] = ord("N")
    want_multi = ok.sketch_records(param, table, multi, cuts)
    assert len(want_multi) > 5000

    # ---- resident in HBM: rk_sketch_packed_dev (records of a genome separated by one 0x00, start at a multiple of 1,024)
    flt = ctx.filter(capi.params_init(k, s, l), table)
    pad = (n + 1023) // 1024 * 1024
    packed = torch.zeros(pad, dtype=torch.uint8, device="cuda")
    packed[:n] = torch.from_numpy(bases).cuda()
    sk = ctx.sketch_packed_dev(flt, packed.data_ptr(), packed.numel(), np.array([0], dtype=np.uint64), np.array([n], dtype=np.uint64))
    gh, goff = sk.download()
    assert np.array_equal(gh.astype(np.uint64), want_one)
    del sk
    # the multi-record genome through the record API (rk_sketch_batch packs it: one separator per record boundary)
    sk = ctx.sketch_batch(flt, multi, cuts, np.array([0, len(cuts) - 1], dtype=np.uint64))
    gh, goff = sk.download()
    assert np.array_equal(gh.astype(np.uint64), want_multi)
    del sk, packed
    torch.cuda.empty_cache()

    # ---- the command line: each genome as one big FASTA file next to a tiny one (size rule of src/sketch.cpp:366-374:
    # the big one takes the streamed path), 80-column lines / 61 records with lines of 70
    shuf = tmp_path / "L4K10.shuf"
    assert subprocess.run([tool, "shuffle", "-k", str(k), "-s", str(s), "-l", str(l), "-o", str(shuf)], stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL).returncode == 0
    (tmp_path / "one.fa").write_bytes(synth.fasta_text("chr_one", bases))
    with open(tmp_path / "multi.fa", "wb") as f:
        for i in range(len(cuts) - 1):
            f.write(synth.fasta_text("scaffold_%d extra words" % i, multi[int(cuts[i]):int(cuts[i + 1])], width=70))
    (tmp_path / "small.fa").write_bytes(b">s\nACGTACGTACGTACGTACGTAAAACCCCGGGGTTTTACGATCGATCGAT\n")
    for name, want in (("one", want_one), ("multi", want_multi)):
        lst = tmp_path / (name + ".list")
        lst.write_text("%s\n%s\n" % (tmp_path / "small.fa", tmp_path / (name + ".fa")))
        p = subprocess.run([tool, "sketch", "-i", str(lst), "-L", str(shuf), "-o", str(tmp_path / name), "-t", "8", "-q"], cwd=tmp_path,
                           env=dict(os.environ, RK_TIMING="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        assert b"big file:" in p.stderr                               # the streamed path ran
        _, names, h, off = ok.read_sketches32(str(tmp_path / (name + ".sketch")))
        assert np.array_equal(h[int(off[1]):int(off[2])].astype(np.uint64), want), name


def test_threshold_within_one_ulp_device_log_vs_host_libm(ctx):
    """The reference keeps a pair when mashD < maxDist (src/dist.cpp:232), mashD from glibc's log.  rk_dist_rows recomputes every
    reported distance with the host libm and decides there: bit for bit the reference, also for a threshold ON a pair's distance.
    rk_dist_rows_dev keeps the device's own log (north_star: 1e-12): a pair whose distance lies within an ulp or two of -D may
    land on either side there -- this test pins how far: thresholds four ulps away from any pair's distance agree with the host,
    and the device's distance itself is within 2 ulps of the host's."""
    import math
    rng = np.random.default_rng(12)
    m = 400
    base = np.sort(rng.choice(1 << 26, size=m, replace=False)).astype(np.uint32)
    parts = [base]
    for c in (397, 380, 351, 300, 260, 201):           # genomes sharing c of base's m hashes
        own = rng.choice(1 << 26, size=m - c, replace=False).astype(np.uint32)
        parts.append(np.unique(np.concatenate([base[:c], own])))
    off = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
    h = np.concatenate(parts)
    idx = ctx.index_build(ctx.sketches_from_host(h, off), 26)
    postings, counts = ok.index_build32(h, off, 26)
    sizes = np.diff(off).astype(np.uint32)
    allp, _ = ok.index_dist32(counts, 26, postings, sizes, h, off, 1, 0, 20, 0.9999)
    dists = sorted(set(float(x) for x in allp["dist"] if 0.0 < x < 0.5))
    assert len(dists) >= 6
    ulp_off = 0
    for d0 in dists[:6]:
        ulp = math.ulp(d0)
        for k, D in ((-4, d0 - 4 * ulp), (-1, math.nextafter(d0, 0.0)), (0, d0), (1, math.nextafter(d0, 1.0)), (4, d0 + 4 * ulp)):
            want, _ = ok.index_dist32(counts, 26, postings, sizes, h, off, 1, 0, 20, D)
            check_hits(ctx.dist_rows(idx, None, 1, 0, 20, D)[0], want)          # the synchronous API: the reference's decision, always
            dev = dev_hits(ctx, idx, 1, 0, 20, D)
            if abs(k) >= 4:
                assert len(dev) == len(want) and np.array_equal(dev["row"], want["row"]) and np.array_equal(dev["col"], want["col"]), (d0, k)
            else:
                ulp_off += int(len(dev) != len(want))                            # (allowed: the device's log is its own)
        dev = dev_hits(ctx, idx, 1, 0, 20, 0.9999)
        mine = dev[np.isclose(dev["dist"], d0, rtol=0, atol=8 * ulp)]
        assert len(mine) >= 1 and np.all(np.abs(mine["dist"] - d0) <= 2 * ulp)
    print("thresholds within one ulp of a pair's distance where the device-resident API decided differently from the host: %d of 18" % ulp_off)
