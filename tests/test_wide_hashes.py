"""64-bit hash layout (use64: half_k - drlevel > 8).
tests/golden/dist64 was produced by the REAL reference's use64 branch (index_tridist/index_dist
reading a .dict/.index pair whose posting blocks are in scrambled, hash-map-like order)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle as ok
from rabbitkssd_amd import capi, synth

D64 = os.path.join(GOLDEN, "dist64")


def load():
    man = json.load(open(os.path.join(D64, "manifest.json")))
    _, rnames, rh, roff = ok.read_sketches64(os.path.join(D64, "ref64.sketch"))
    _, qnames, qh, qoff = ok.read_sketches64(os.path.join(D64, "qry64.sketch"))
    return man, (rnames, rh, roff), (qnames, qh, qoff)


def test_oracle64_matches_reference_text():
    man, (rnames, rh, roff), (qnames, qh, qoff) = load()
    uhash, ucount, postings = ok.index_build64(rh, roff)
    rsizes = np.diff(roff).astype(np.uint32)
    kmer = 2 * man["half_k"]
    for case in man["cases"]:
        want = open(os.path.join(D64, case["file"])).read().split("\n")[:-1]
        if case["cmd"] == "alldist":
            hits, _ = ok.index_dist64(uhash, ucount, postings, rsizes, rh, roff, 1, case["metric"], kmer,
                                      case["max_dist"], threads=2)
            assert sorted(x.rstrip("\n") for x in ok.alldist_text(rnames, hits)) == want, case["file"]
        else:
            hits, _ = ok.index_dist64(uhash, ucount, postings, rsizes, qh, qoff, 0, case["metric"], kmer,
                                      case["max_dist"])
            if case["max_neighbor"]:
                hits = np.concatenate([ok.topn_row(hits[hits["row"] == q], case["max_neighbor"])
                                       for q in range(len(qnames))])
            assert [x.rstrip("\n") for x in ok.dist_text(qnames, rnames, hits)] == want, case["file"]


def test_oracle64_index_files_roundtrip(tmp_path):
    _, (rnames, rh, roff), _ = load()
    uhash, ucount, postings = ok.index_build64(rh, roff)
    dp, ip = str(tmp_path / "x.dict"), str(tmp_path / "x.index")
    ok.write_index64(dp, ip, postings, uhash, ucount)
    assert os.path.getsize(ip) == 8 + 12 * len(uhash) and os.path.getsize(dp) == 4 * len(rh)   # src/sketch.cpp:961-963
    p2, h2, c2 = ok.read_index64(dp, ip)
    assert np.array_equal(p2, postings) and np.array_equal(h2, uhash) and np.array_equal(c2, ucount)


@pytest.fixture(scope="module")
def ctx():
    return capi.Context(0)


@pytest.mark.gpu
def test_gpu_wide_index_and_distances_match_reference(ctx):
    man, (rnames, rh, roff), (qnames, qh, qoff) = load()
    bits, kmer = man["hash_bits"], 2 * man["half_k"]
    rsk = ctx.sketches_from_host64(rh, roff)
    qsk = ctx.sketches_from_host64(qh, qoff)
    assert rsk.is64
    built = ctx.index_build(rsk, bits)
    uhash, ucount, postings = ok.index_build64(rh, roff)
    p2, h2, c2 = built.export64()
    assert np.array_equal(p2, postings) and np.array_equal(h2, uhash) and np.array_equal(c2, ucount)
    # import from a file-like scrambled block order
    rng = np.random.default_rng(3)
    perm = rng.permutation(len(uhash))
    starts = np.concatenate([[0], np.cumsum(ucount.astype(np.int64))])
    scr = np.concatenate([postings[int(starts[i]):int(starts[i + 1])] for i in perm])
    imported = ctx.index_import64(scr, uhash[perm], ucount[perm], bits, np.diff(roff))
    assert np.array_equal(imported.export64()[0], postings)
    for case in man["cases"]:
        want = open(os.path.join(D64, case["file"])).read().split("\n")[:-1]
        if case["cmd"] == "alldist":
            for idx, q in ((built, None), (imported, rsk)):
                hits, _ = ctx.dist_rows(idx, q, 1, case["metric"], kmer, case["max_dist"])
                mine = sorted(capi.format_hit(rnames[h["col"]], rnames[h["row"]], h).rstrip("\n") for h in hits)
                assert mine == want, case["file"]
        else:
            for idx in (built, imported):
                hits, _ = ctx.dist_rows(idx, qsk, 0, case["metric"], kmer, case["max_dist"])
                if case["max_neighbor"]:
                    hits = capi.topn_rows(hits, case["max_neighbor"])
                mine = [capi.format_hit(qnames[h["row"]], rnames[h["col"]], h).rstrip("\n") for h in hits]
                assert mine == want, case["file"]
    # dense counter rows
    want_hits, want = ok.index_dist64(uhash, ucount, postings, np.diff(roff).astype(np.uint32), qh, qoff, 0, 0, kmer,
                                      0.4, want_dense=True)
    hits, dense = ctx.dist_rows(built, qsk, 0, 0, kmer, 0.4, want_dense=True)
    assert np.array_equal(dense, want) and len(hits) == len(want_hits)
    # width mismatch is an error, not a silent wrong answer
    q32 = ctx.sketches_from_host(np.array([1, 2, 3], dtype=np.uint32), np.array([0, 3], dtype=np.uint64))
    with pytest.raises(capi.RkError):
        ctx.dist_rows(built, q32, 0, 0, kmer, 0.4)


@pytest.mark.gpu
@pytest.mark.parametrize("k,s,l", [(12, 6, 3), (11, 6, 2), (16, 7, 4)])
def test_gpu_wide_sketch_vs_oracle(ctx, k, s, l):
    """K12L3 -> 36-bit, K11L2 -> 36-bit (1/256 sampling), K16 S7 L4 -> 48-bit hashes"""
    param, table = ok.init_param(k, s, l), ok.shuffle_table(k, s, l)
    flt = ctx.filter(capi.params_init(k, s, l), table)
    genomes = synth.clade_genome_set(5, 120000)
    seq = np.concatenate([b for _, b in genomes])
    seq[1000:1040] = ord("N")
    rec_off = np.arange(6, dtype=np.uint64) * 120000
    sk = ctx.sketch_batch(flt, seq, rec_off, np.arange(6, dtype=np.uint64))
    assert sk.is64
    gh, goff = sk.download()
    assert gh.dtype == np.uint64
    total = 0
    for g in range(5):
        want = ok.sketch_records(param, table, seq[g * 120000:(g + 1) * 120000], np.array([0, 120000], dtype=np.uint64))
        assert np.array_equal(gh[int(goff[g]):int(goff[g + 1])], want), g
        total += len(want)
    assert total > 0 and int(gh.max()) >= (1 << 32)
    # end to end on the wide path: index + alldist
    idx = ctx.index_build(sk, 4 * (k - l))
    uhash, ucount, postings = ok.index_build64(gh, goff)
    want, _ = ok.index_dist64(uhash, ucount, postings, np.diff(goff).astype(np.uint32), gh, goff, 1, 0, 2 * k, 0.2)
    mine, _ = ctx.dist_rows(idx, None, 1, 0, 2 * k, 0.2)
    assert len(mine) == len(want) and np.array_equal(mine["common"], want["common"])
    assert np.array_equal(mine["dist"], want["dist"])


@pytest.mark.gpu
def test_gpu_wide_index_fast_build_equals_general_build(monkeypatch):
    """36-bit hashes (K12 L3) take the bucket-sort build like 32-bit ones (the kernels that read the sketches are templated
    on the hash type): postings, distinct hashes, list lengths and the self join equal the general build's and the oracle's"""
    names, h, off = synth.clade_sketches(3000, 400, 36, kmer_size=24, seed=5, wide=True)
    order = synth.genome_order(len(names), "shuffled", seed=3)
    names, h, off = synth.permute_genomes(names, h, off, order)
    assert h.dtype == np.uint64 and int(h.max()) >= (1 << 32)
    fast = capi.Context(0)
    monkeypatch.setenv("RK_INDEX_FAST", "0")
    slow = capi.Context(0)
    monkeypatch.delenv("RK_INDEX_FAST")
    i_fast = fast.index_build(fast.sketches_from_host64(h, off), 36)
    i_slow = slow.index_build(slow.sketches_from_host64(h, off), 36)
    assert i_fast.built_fast and not i_slow.built_fast
    uhash, ucount, postings = ok.index_build64(h, off)
    for idx in (i_fast, i_slow):
        p2, h2, c2 = idx.export64()
        assert np.array_equal(p2, postings) and np.array_equal(h2, uhash) and np.array_equal(c2, ucount)
    want, _ = ok.index_dist64(uhash, ucount, postings, np.diff(off).astype(np.uint32), h, off, 1, 0, 24, 0.05)
    assert len(want) > 3000
    for c, idx in ((fast, i_fast), (slow, i_slow)):
        mine, _ = c.dist_rows(idx, None, 1, 0, 24, 0.05)
        assert len(mine) == len(want) and np.array_equal(mine["row"], want["row"]) and np.array_equal(mine["col"], want["col"])
        assert np.array_equal(mine["common"], want["common"]) and np.array_equal(mine["dist"], want["dist"])
    del i_fast, i_slow
    fast.close()
    slow.close()


@pytest.mark.gpu
def test_gpu_wide_index_in_passes_and_in_shards(monkeypatch):
    """36-bit hashes through the range passes of the bucket sort (k_range_filter<u64>: its hashes are read twice, not kept in
    registers; 64-bit sort keys in the bucket emission) and through the sharded build + join: the same postings, distinct hashes
    and hits as the oracle"""
    import torch
    names, h, off = synth.clade_sketches(3000, 300, 36, kmer_size=24, seed=15, wide=True)
    order = synth.genome_order(len(names), "shuffled", seed=4)
    names, h, off = synth.permute_genomes(names, h, off, order)
    uhash, ucount, postings = ok.index_build64(h, off)
    sizes = np.diff(off).astype(np.uint32)
    want, _ = ok.index_dist64(uhash, ucount, postings, sizes, h, off, 1, 0, 24, 0.05)
    assert len(want) > 3000
    monkeypatch.setenv("RK_INDEX_PASS_BITS", "2")   # (read by the build itself)
    c = capi.Context(0)
    sk = c.sketches_from_host64(h, off)
    idx = c.index_build(sk, 36)
    monkeypatch.delenv("RK_INDEX_PASS_BITS")
    assert idx.built_fast and idx.products == 6
    p2, h2, c2 = idx.export64()
    assert np.array_equal(p2, postings) and np.array_equal(h2, uhash) and np.array_equal(c2, ucount)

    def same(mine):
        assert len(mine) == len(want) and np.array_equal(mine["row"], want["row"]) and np.array_equal(mine["col"], want["col"])
        assert np.array_equal(mine["common"], want["common"]) and np.array_equal(mine["dist"], want["dist"])
    same(c.dist_rows(idx, None, 1, 0, 24, 0.05)[0])
    del idx
    c.close()
    c = capi.Context(0)
    sk = c.sketches_from_host64(h, off)
    W = 4
    parts = [c.index_build_shard(sk, 36, r, W) for r in range(W)]
    assert sum(p.total for p in parts) == len(h)
    sent = [p.shard_records(W) for p in parts]
    bufs = []
    for p, cnt in zip(parts, sent):
        b = torch.empty(max(1, sum(cnt) * 12), dtype=torch.uint8, device="cuda")
        p.shard_pack(b.data_ptr())
        bufs.append(b)
    torch.cuda.synchronize()
    got = []
    for d in range(W):
        recv = torch.cat([bufs[r][12 * sum(sent[r][:d]): 12 * sum(sent[r][:d + 1])] for r in range(W)] + [torch.empty(1, dtype=torch.uint8, device="cuda")])
        j = c.index_join_shard(parts[d], recv.data_ptr(), sum(sent[r][d] for r in range(W)))
        got.append(c.dist_rows(j, None, 1, 0, 24, 0.05)[0])
        del j
    merged = np.concatenate(got)
    same(merged[np.lexsort((merged["col"], merged["row"]))])
    del parts
    c.close()


# ------------------------------------------------------------------ host tool on the 64-bit layout
def _tool(args, cwd=None, check=True):
    import subprocess
    from conftest import ROOT
    p = subprocess.run([os.path.join(ROOT, "rabbitkssd_amd", "rabbit_kssd")] + [str(a) for a in args], cwd=cwd,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if check and p.returncode != 0:
        raise AssertionError("rabbit_kssd %s failed:\n%s" % (args, p.stderr.decode()))
    return p


def test_tool_set_algebra_and_info_on_wide_sketches(tmp_path):
    _, (rnames, rh, roff), (qnames, qh, qoff) = load()
    ref, qry = os.path.join(D64, "ref64.sketch"), os.path.join(D64, "qry64.sketch")
    _tool(["union", "-i", ref, "-o", tmp_path / "u.sketch"])
    info, names, h, off = ok.read_sketches64(str(tmp_path / "u.sketch"))
    assert info.genomeNumber == 1 and np.array_equal(h, np.unique(rh))
    _tool(["sub", "--rs", ref, "--qs", qry, "-o", tmp_path / "s.sketch"])
    info, names, h, off = ok.read_sketches64(str(tmp_path / "s.sketch"))
    refset = set(rh.tolist())
    for i in range(len(qnames)):
        assert h[int(off[i]):int(off[i + 1])].tolist() == [x for x in qh[int(qoff[i]):int(qoff[i + 1])].tolist()
                                                            if x not in refset]
    _tool(["info", "-i", qry, "-o", tmp_path / "i.txt", "-F"])
    lines = (tmp_path / "i.txt").read_text().split("\n")
    assert lines[1] == "%s\t%d" % (qnames[0], qoff[1] - qoff[0]) and lines[2].split("\t")[0] == str(qh[0])
    assert _tool(["convert", "--reverse", "-i", qry, "-o", tmp_path / "k"], check=False).returncode == 1


@pytest.mark.gpu
def test_tool_wide_alldist_dist_and_sketch(tmp_path):
    man, (rnames, rh, roff), (qnames, qh, qoff) = load()
    ref = tmp_path / "ref64.sketch"
    ref.write_bytes(open(os.path.join(D64, "ref64.sketch"), "rb").read())
    for case in man["cases"]:
        want = open(os.path.join(D64, case["file"])).read().split("\n")[:-1]
        if case["cmd"] == "alldist":
            _tool(["alldist", "-i", ref, "-D", case["max_dist"], "-M", case["metric"], "-o", "o.txt"], cwd=tmp_path)
            got = (tmp_path / "o.txt").read_text().split("\n")
            assert sorted(x for x in got[1:] if x) == want, case["file"]
        else:
            args = ["dist", "-r", ref, "-q", os.path.join(D64, "qry64.sketch"), "-D", case["max_dist"], "-M",
                    case["metric"], "-o", "o.txt"]
            if case["max_neighbor"]:
                args += ["-N", case["max_neighbor"]]
            _tool(args, cwd=tmp_path)
            assert [x for x in (tmp_path / "o.txt").read_text().split("\n")[1:] if x] == want, case["file"]
    # the tool wrote the sparse .dict/.index pair the reference's use64 reader expects
    postings, uhash, ucount = ok.read_index64(str(ref) + ".dict", str(ref) + ".index")
    wu, wc, wp = ok.index_build64(rh, roff)
    assert np.array_equal(postings, wp) and np.array_equal(uhash, wu) and np.array_equal(ucount, wc)
    # sketch with a K12 shuffle file -> 64-bit .sketch
    k, s, l = 12, 6, 3
    shuf = tmp_path / "k12.shuf"
    _tool(["shuffle", "-k", k, "-s", s, "-l", l, "-o", shuf])
    files = []
    for i, (name, bases) in enumerate(synth.clade_genome_set(3, 150000)):
        p = tmp_path / (name + ".fa")
        p.write_bytes(synth.fasta_text(name, bases))
        files.append(str(p))
    lst = tmp_path / "w.list"
    lst.write_text("\n".join(files) + "\n")
    _tool(["sketch", "-i", lst, "-L", shuf, "-o", tmp_path / "w", "-q"], cwd=tmp_path)
    info, names, h, off = ok.read_sketches64(str(tmp_path / "w.sketch"))
    param, table = ok.init_param(k, s, l), ok.shuffle_table(k, s, l)
    for i, f in enumerate(files):
        seq, o = ok.read_fasta(f)
        assert np.array_equal(h[int(off[i]):int(off[i + 1])], ok.sketch_records(param, table, seq, o))
