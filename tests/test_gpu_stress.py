"""Seeded sweep of the self join and the query path over many small shapes and plans (GPU): collection sizes around the
workgroup / pair / band / tile boundaries, hash spaces from crowded to sparse, both metrics, sparse and dense
thresholds, row shards -- every result against the oracle.  The developer switches shrink the planner's idea of the
LDS so that tiles, bands, single rows and row pairs all occur at these sizes."""
import numpy as np
import pytest

from oracle import oracle as ok
from rabbitkssd_amd import capi, synth

pytestmark = pytest.mark.gpu


def _check(mine, want):
    assert len(mine) == len(want)
    for f in ("row", "col", "common", "size0", "size1"):
        assert np.array_equal(mine[f], want[f]), f
    assert np.array_equal(mine["jorc"], want["jorc"])
    assert np.array_equal(mine["dist"], want["dist"])   # the host libm has the last word in rk_dist_rows


PLANS = [
    {},                                                              # as shipped
    # (RK_DIST_NEAR=0: the kernel with full counter rows, which is also the near-window kernel's fallback)
    {"RK_DIST_NEAR": "0", "RK_DIST_BAND_MIN_ROWS": "64"},                                 # bands wherever the variant changes
    {"RK_DIST_NEAR": "0", "RK_DIST_BAND_MIN_ROWS": "64", "RK_DIST_LDS_KB": "16"},         # tiled first band, then single rows
    {"RK_DIST_NEAR": "0", "RK_DIST_BAND_MIN_ROWS": "32", "RK_DIST_LDS_KB": "20", "RK_DIST_CAND_CAP": "16"},   # cell lists overflow
    {"RK_DIST_NEAR": "0", "RK_DIST_BANDS": "0", "RK_DIST_PAIR": "2"},                     # one launch, no pairs
    {"RK_DIST_NEAR": "0", "RK_DIST_THREADS": "1024", "RK_DIST_CAND_CAP": "8"},            # batched evaluation with tiny lists
    {"RK_DIST_NEAR_MIN": "1", "RK_DIST_LDS_KB": "16"},               # near-window kernel (also for tiny sketches), its fallback with tiled rows
    {"RK_DIST_NEAR_MIN": "1", "RK_DIST_PAIR": "2", "RK_DIST_CAND_CAP": "8"},   # near-window kernel on single rows
]


@pytest.mark.parametrize("plan", range(len(PLANS)))
def test_self_join_shapes(monkeypatch, plan):
    for k, v in PLANS[plan].items():
        monkeypatch.setenv(k, v)
    ctx = capi.Context(0)
    rng = np.random.default_rng(100 + plan)
    shapes = [(1, 5, 16), (2, 9, 16), (3, 40, 18), (17, 30, 14), (64, 64, 20), (129, 25, 12), (500, 33, 20), (1023, 12, 16),
              (2049, 18, 22), (4100, 10, 18), (7001, 8, 24)]
    for n, m, bits in shapes:
        names, h, off = synth.clade_sketches(n, m, bits, seed=int(rng.integers(1 << 30)))
        idx = ctx.index_build(ctx.sketches_from_host(h, off), bits)
        postings, counts = ok.index_build32(h, off, bits)
        sizes = np.diff(off).astype(np.uint32)
        for metric, D in ((0, 0.05), (1, 0.15), (0, 1.5)):
            want, _ = ok.index_dist32(counts, bits, postings, sizes, h, off, 1, metric, 20, D, threads=8)
            _check(ctx.dist_rows(idx, None, 1, metric, 20, D)[0], want)
        if n >= 64:  # three uneven row shards
            want, _ = ok.index_dist32(counts, bits, postings, sizes, h, off, 1, 0, 20, 0.08, threads=8)
            parts = [ctx.dist_rows(idx, None, 1, 0, 20, 0.08, row_first=r, row_step=3, row_block=6)[0] for r in range(3)]
            merged = np.concatenate(parts)
            _check(merged[np.lexsort((merged["col"], merged["row"]))], want)
        del idx
    ctx.close()


def test_query_path_shapes():
    ctx = capi.Context(0)
    rng = np.random.default_rng(7)
    for n_ref, m_ref, n_q, m_q, bits in [(1, 4, 1, 4, 12), (10, 20, 3, 500, 16), (300, 76, 40, 3000, 20), (2000, 30, 25, 9000, 24),
                                         (5000, 12, 7, 70000, 22), (900, 40, 900, 40, 18)]:
        rn, rh, roff = synth.clade_sketches(n_ref, m_ref, bits, seed=int(rng.integers(1 << 30)))
        qn, qh, qoff = synth.clade_sketches(n_q, m_q, bits, seed=int(rng.integers(1 << 30)))
        idx = ctx.index_build(ctx.sketches_from_host(rh, roff), bits)
        qs = ctx.sketches_from_host(qh, qoff)
        postings, counts = ok.index_build32(rh, roff, bits)
        sizes = np.diff(roff).astype(np.uint32)
        for metric, D in ((0, 0.2), (1, 0.3), (0, 1.5)):
            want, _ = ok.index_dist32(counts, bits, postings, sizes, qh, qoff, 0, metric, 20, D, threads=8)
            _check(ctx.dist_rows(idx, qs, 0, metric, 20, D)[0], want)
        del idx, qs
    ctx.close()


@pytest.mark.parametrize("img", [2, 1, 0])
def test_sketch_shapes(monkeypatch, img):
    """random genomes (IUPAC codes, lower case, N runs, empty and tiny records, many records) under several parameter
    sets, with the two-stage scan kernel (default; K10S6 and K8S5 have a compile-time variant, the rest falls back), with
    rk_sketch_kernel's 64 KiB LDS image (RK_SKETCH_IMG=1) and with its 144 KiB one (RK_SKETCH_IMG=0): hash sets == oracle"""
    from test_gpu_parity import sketch_case
    monkeypatch.setenv("RK_SKETCH_IMG", str(img))
    ctx = capi.Context(0)
    rng = np.random.default_rng(900 + img)
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTacgtNnRYKMSWBDHV-", dtype=np.uint8)
    for k, s, l in [(10, 6, 3), (8, 5, 2), (10, 7, 4), (7, 4, 1), (11, 6, 2), (12, 6, 3), (16, 6, 3)]:
        genomes = []
        for g in range(9):
            n = int([0, 1, 2 * k - 1, 2 * k, 2 * k + 1, 1023, 1024 + 2 * k, 33333, 150001][g])
            b = alphabet[rng.integers(0, 16 if g % 2 else len(alphabet), size=n)].copy()
            if n > 1000:  # a repeat, so that the dedup has something to do
                b[n // 2:n // 2 + 300] = b[100:400]
            cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, size=int(rng.integers(0, 6)))]))
            genomes.append((b, np.array(cuts, dtype=np.uint64)))
        sketch_case(ctx, k, s, l, genomes)
    ctx.close()


def test_sketch_packed_offsets_beyond_2_and_4_gib():
    """rk_sketch_packed_dev on a 4.6 GB packed buffer: genomes whose byte offsets need bit 31 and bit 32 (the chunk
    table carries 64-bit offsets; a sign-extended low word once sent a wave 16 EB away) -- hash sets == oracle"""
    import torch
    k, s, l = 10, 6, 3
    param, table = ok.init_param(k, s, l), ok.shuffle_table(k, s, l)
    ctx = capi.Context(0)
    flt = ctx.filter(capi.params_init(k, s, l), table)
    rng = np.random.default_rng(4321)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    total = 4_600_000_000 // 1024 * 1024
    packed = torch.full((total,), ord("N"), dtype=torch.uint8, device="cuda")
    mib = 1 << 20  # 2.5 MB each: two genomes straddle the 2 GiB and the 4 GiB offset, two start just beyond them
    places = [0, (1 << 31) - mib, (1 << 31) + 8 * mib + 1024, (1 << 32) - mib, (1 << 32) + 8 * mib + 5 * 1024, total - 3 * mib]
    genomes, gbeg, gend = [], [], []
    for i, at in enumerate(places):
        n = 2_500_000 + 1000 * i + 7
        g = lut[rng.integers(0, 4, n)]
        packed[at:at + n] = torch.from_numpy(g).cuda()
        genomes.append(g)
        gbeg.append(at)
        gend.append(at + n)
    torch.cuda.synchronize()
    sk = ctx.sketch_packed_dev(flt, packed.data_ptr(), packed.numel(), np.array(gbeg, dtype=np.uint64), np.array(gend, dtype=np.uint64))
    gh, goff = sk.download()
    for i, g in enumerate(genomes):
        want = ok.sketch_records(param, table, g, np.array([0, len(g)], dtype=np.uint64))
        assert np.array_equal(gh[int(goff[i]):int(goff[i + 1])].astype(np.uint64), want), "genome %d" % i
    del sk, packed
    ctx.close()


@pytest.mark.parametrize("k,s,l", [(10, 6, 3), (10, 7, 4), (8, 5, 2)])
def test_sketch_sequences_dense_in_selected_windows(k, s, l):
    """genomes built FROM selected k-mers (every 2k-th window passes both bitmaps of the scan kernel, 50x the density of a
    random genome), poly-A and short tandem repeats: the queues of the scan kernel overflow in every block and its
    in-place path runs; hash sets and window counts == oracle"""
    from test_gpu_parity import sketch_case
    param, table = ok.init_param(k, s, l), ok.shuffle_table(k, s, l)
    sel = np.nonzero((table >= param.dim_start) & (table < param.dim_end))[0]
    rng = np.random.default_rng(77 + k + s)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)          # BaseMap codes (src/common.h:27-37)
    inner, outer = 2 * s, k - s
    n_kmers = 40_000
    d = sel[rng.integers(0, len(sel), n_kmers)].astype(np.uint64)
    shifts = (2 * (inner - 1 - np.arange(inner))).astype(np.uint64)   # first inner base in the high bits
    inner_codes = ((d[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.uint8)
    kmers = np.concatenate([rng.integers(0, 4, (n_kmers, outer), dtype=np.uint8), inner_codes,
                            rng.integers(0, 4, (n_kmers, outer), dtype=np.uint8)], axis=1)
    dense = lut[kmers.reshape(-1)]
    genomes = [(dense, np.array([0, len(dense)], dtype=np.uint64)),
               (np.full(300_000, ord("A"), dtype=np.uint8), np.array([0, 300_000], dtype=np.uint64)),
               (np.tile(np.frombuffer(b"ACGTTGCAAC", dtype=np.uint8), 40_000), np.array([0, 150_000, 400_000], dtype=np.uint64)),
               (np.concatenate([dense[:100_000], np.full(50, ord("N"), dtype=np.uint8), dense[100_000:300_000]]),
                np.array([0, 300_050], dtype=np.uint64))]
    want = ok.sketch_records(param, table, dense, genomes[0][1])
    assert len(want) > 200, "the construction must hit selected windows (%d hashes)" % len(want)
    ctx = capi.Context(0)
    sketch_case(ctx, k, s, l, genomes)
    ctx.close()
