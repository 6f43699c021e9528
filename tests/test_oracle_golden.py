"""The CPU restatement (oracle/) against the committed golden fixtures.

params.txt, shuf.json and dist/*.ref.txt were produced by the REAL reference objects
(/root/reference/src/{common,shuffle,dist}.cpp via oracle/_ref/ref_driver); see
tests/golden/make_golden.py.  sketch/expected.json comes from the restatement; the reference-produced sketch fixtures are
tests/golden/sketch_ref (tests/test_sketch_ref_golden.py)."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle as ok
from conftest import GOLDEN


def test_params_match_reference_initParameter():
    # src/common.cpp:35-78
    for line in open(os.path.join(GOLDEN, "params.txt")):
        f = line.split()
        k, s, l = int(f[0]), int(f[1]), int(f[2])
        p = ok.init_param(k, s, l)
        mine = "%d %d %d %d %d %d %d %d %x %x %x %x" % (
            p.half_k, p.half_subk, p.drlevel, p.rev_add_move, p.half_outctx_len, p.dim_start,
            p.dim_end, p.kmer_size, p.domask, p.tupmask, p.undomask0, p.undomask1)
        assert mine == line.strip()


def test_param_rejects_small_dim():
    # src/common.cpp:37 -- L4 with -s 6 (as init_shuffle.sh generates it) is rejected
    with pytest.raises(ValueError):
        ok.init_param(10, 6, 4)


def test_shuffle_table_matches_reference_md5():
    # src/shuffle.cpp:25-104 (glibc rand())
    for case in json.load(open(os.path.join(GOLDEN, "shuf.json"))):
        t = ok.shuffle_table(case["k"], case["s"], case["l"])
        hdr = np.array(case["header"], dtype="<i4").tobytes()
        assert t[:8].tolist() == case["first8"]
        assert hashlib.md5(hdr + t.astype("<i4").tobytes()).hexdigest() == case["md5"]
        assert int((t < (1 << (4 * (case["s"] - case["l"])))).sum()) == case["n_below_dim_end"]
        assert np.array_equal(np.sort(t), np.arange(len(t)))  # a permutation


def _load_dist_case():
    d = os.path.join(GOLDEN, "dist")
    man = json.load(open(os.path.join(d, "manifest.json")))
    _, rnames, rh, roff = ok.read_sketches32(os.path.join(d, "ref.sketch"))
    _, qnames, qh, qoff = ok.read_sketches32(os.path.join(d, "qry.sketch"))
    postings, counts = ok.index_build32(rh, roff, man["hash_bits"])
    return d, man, (rnames, rh, roff), (qnames, qh, qoff), postings, counts


def test_alldist_and_dist_text_match_reference():
    d, man, (rnames, rh, roff), (qnames, qh, qoff), postings, counts = _load_dist_case()
    rsizes = np.diff(roff).astype(np.uint32)
    kmer = 2 * man["half_k"]
    for case in man["cases"]:
        want = open(os.path.join(d, case["file"])).read().split("\n")[:-1]
        assert len(want) == case["lines"]
        if case["cmd"] == "alldist":
            hits, _ = ok.index_dist32(counts, man["hash_bits"], postings, rsizes, rh, roff, 1,
                                      case["metric"], kmer, case["max_dist"], threads=2)
            mine = sorted(x.rstrip("\n") for x in ok.alldist_text(rnames, hits))
            assert mine == want, case["file"]
        else:
            hits, _ = ok.index_dist32(counts, man["hash_bits"], postings, rsizes, qh, qoff, 0,
                                      case["metric"], kmer, case["max_dist"])
            if case["max_neighbor"]:
                hits = np.concatenate([ok.topn_row(hits[hits["row"] == q], case["max_neighbor"])
                                       for q in range(len(qnames))])
            mine = [x.rstrip("\n") for x in ok.dist_text(qnames, rnames, hits)]
            assert mine == want, case["file"]


def test_dense_counts_equal_set_intersection():
    _, man, (rnames, rh, roff), (qnames, qh, qoff), postings, counts = _load_dist_case()
    rsizes = np.diff(roff).astype(np.uint32)
    _, dense = ok.index_dist32(counts, man["hash_bits"], postings, rsizes, qh, qoff, 0, 0, 16, 1.0,
                               want_dense=True)
    for q in range(len(qnames)):
        sq = set(qh[int(qoff[q]):int(qoff[q + 1])].tolist())
        for r in range(len(rnames)):
            sr = set(rh[int(roff[r]):int(roff[r + 1])].tolist())
            assert dense[q, r] == len(sq & sr)


def test_index_files_roundtrip(tmp_path):
    _, man, (rnames, rh, roff), _, postings, counts = _load_dist_case()
    dp, ip = str(tmp_path / "x.dict"), str(tmp_path / "x.index")
    ok.write_index32(dp, ip, postings, counts, man["hash_bits"])
    assert os.path.getsize(ip) == 16 + 4 * (1 << man["hash_bits"])  # src/sketch.cpp:1008-1011
    assert os.path.getsize(dp) == 4 * len(rh)
    p2, c2 = ok.read_index32(dp, ip)
    assert np.array_equal(p2, postings) and np.array_equal(c2, counts)


def _fnv(data):
    h = 1469598103934665603
    for c in bytes(data):
        h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def test_base_code_matches_reference_basemap():
    # tests/golden/basemap.txt: the BaseMap[128] table of src/common.h printed by oracle/_ref/ref_driver basemap
    want = [int(x) for x in open(os.path.join(GOLDEN, "basemap.txt")).read().split()]
    assert len(want) == 128 and sorted(set(want)) == [-1, 0, 1, 2, 3]
    L = ok.lib()
    assert [L.ok_base_code(c) for c in range(128)] == want
    assert all(L.ok_base_code(c) == -1 for c in range(128, 256))


def test_record_reader_restatement_matches_the_real_kseq():
    """S0 pinned: tests/golden/kseq/expected.tsv holds what the reference's own kseq.h returned (via
    oracle/_ref/ref_driver kseq) for 29 inputs; the C restatement must agree on records, bases,
    sequence bytes, quality bytes and boundaries"""
    import gzip
    d = os.path.join(GOLDEN, "kseq")
    rows = [l.split("\t") for l in open(os.path.join(d, "expected.tsv")).read().split("\n") if l and not l.startswith("#")]
    assert len(rows) == 29
    for w in rows:
        path = os.path.join(d, w[0])
        seq, off = ok.read_fasta(path)                       # gzopen path, plain or compressed
        assert len(off) - 1 == int(w[1]) and len(seq) == int(w[2]), w[0]
        assert _fnv(seq) == w[3], w[0]
        assert [int(x) for x in off[1:]] == [int(x) for x in w[5:]], w[0]
        raw = open(path, "rb").read()
        if w[0].endswith(".gz"):
            import zlib
            dec, raw2 = b"", raw
            while raw2:                                      # every member, like gzread
                z = zlib.decompressobj(31)
                dec += z.decompress(raw2)
                raw2 = z.unused_data
            raw = dec
        s2, q2, o2 = ok.parse_fastq_bytes(raw)
        assert _fnv(s2) == w[3] and _fnv(q2) == w[4], w[0]
        assert [int(x) for x in o2[1:]] == [int(x) for x in w[5:]], w[0]


def test_sketch_fixture_hash_sets():
    """C restatement == expected.json (cross-checked by numpy; the same files, produced by the reference itself, are in
    tests/golden/sketch_ref)."""
    d = os.path.join(GOLDEN, "sketch")
    exp = json.load(open(os.path.join(d, "expected.json")))
    param = ok.init_param(exp["half_k"], exp["half_subk"], exp["drlevel"])
    table = ok.shuffle_table(exp["half_k"], exp["half_subk"], exp["drlevel"])
    for fn, e in exp["files"].items():
        seq, off = ok.read_fasta(os.path.join(d, fn))
        assert len(off) - 1 == e["n_records"] and int(off[-1]) == e["n_bases"]
        assert ok.count_windows(param, seq, off) == e["n_windows"]
        h = ok.sketch_records(param, table, seq, off)
        assert h.tolist() == e["hashes"], fn
    # strains of one clade share most hashes, CRLF stripped, lowercase accepted
    a = set(exp["files"]["c0_s0.fa"]["hashes"])
    b = set(exp["files"]["c0_s1.fa"]["hashes"])
    assert len(a & b) > 0.8 * len(a)


def test_sketch_file_roundtrip(tmp_path):
    names = ["a/b.fna", "c.fna", "empty.fna"]
    hashes = np.array([5, 9, 1 << 27, 3, 4], dtype=np.uint32)
    off = np.array([0, 3, 5, 5], dtype=np.uint64)
    p = str(tmp_path / "t.sketch")
    ok.save_sketches32(p, 10, 6, 3, names, hashes, off)
    info, n2, h2, o2 = ok.read_sketches32(p)
    assert (info.id, info.half_k, info.half_subk, info.drlevel, info.genomeNumber) == (2659, 10, 6, 3, 3)
    assert n2 == names and np.array_equal(h2, hashes) and np.array_equal(o2, off)
    assert os.path.getsize(p) == 20 + 8 * 3 + sum(map(len, names)) + 4 * 5


def test_distance_special_cases():
    # src/dist.cpp:221-231
    assert ok.distance(0, 10, 10, 0, 20) == (0.0, 1.0)
    assert ok.distance(10, 10, 10, 0, 20) == (1.0, 0.0)
    assert ok.distance(0, 0, 10, 0, 20) == (0.0, 1.0)
    j, d = ok.distance(5, 10, 10, 0, 20)
    assert j == 5 / 15 and d == -1.0 / 20 * np.log(2 * j / (1 + j))
    c, a = ok.distance(5, 10, 20, 1, 20)
    assert c == 0.5 and a == -1.0 / 20 * np.log(0.5)
