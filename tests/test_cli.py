"""The C++ host tool rabbitkssd_amd/rabbit_kssd (reference command line above the C ABI).
CPU tests cover the host-only parts (formats, record reader, shuffle); GPU tests run the
sketch/alldist/dist subcommands end to end against the golden fixtures."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from oracle import oracle as ok

TOOL = os.path.join(ROOT, "rabbitkssd_amd", "rabbit_kssd")


def run(args, cwd=None, check=True):
    p = subprocess.run([TOOL] + [str(a) for a in args], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if check and p.returncode != 0:
        raise AssertionError("rabbit_kssd %s failed:\n%s" % (args, p.stderr.decode()))
    return p


def test_tool_is_built():
    assert os.path.exists(TOOL), "python -m rabbitkssd_amd.build"


def test_shuffle_matches_reference_md5(tmp_path):
    for case in json.load(open(os.path.join(GOLDEN, "shuf.json"))):
        out = tmp_path / "x.shuf"
        run(["shuffle", "-k", case["k"], "-s", case["s"], "-l", case["l"], "-o", out])
        assert hashlib.md5(out.read_bytes()).hexdigest() == case["md5"]
    assert run(["shuffle", "-k", 5, "-s", 6, "-l", 3, "-o", tmp_path / "bad"], check=False).returncode == 1


def test_record_reader_matches_oracle_kseq_restatement():
    d = os.path.join(GOLDEN, "sketch")
    files = sorted(f for f in os.listdir(d) if f.endswith((".fa", ".fq")))
    out = run(["_parse"] + [os.path.join(d, f) for f in files]).stdout.decode().strip().split("\n")
    for line, f in zip(out, files):
        cols = line.split("\t")
        seq, off = ok.read_fasta(os.path.join(d, f))
        h = 1469598103934665603
        for c in seq.tobytes():
            h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        assert int(cols[1]) == len(off) - 1 and int(cols[2]) == len(seq), f
        assert cols[3] == "%016x" % h, f
        assert [int(x) for x in cols[5:]] == [int(x) for x in off[1:]], f


def test_record_reader_edge_cases(tmp_path):
    cases = {
        "empty.fa": b"",
        "header_only.fa": b">x",
        "header_nl.fa": b">x\n",
        "one_base.fa": b">x\nA",
        "cr_only.fa": b">x\nA\r\nC\r\n",
        "blank_lines.fa": b"\n\n>a desc\n\nACGT\n\nAC\n>b\n>c\nTT",
        "leading_junk.fa": b"junk\n>a\nACGT\n",
        "fq.fq": b"@r\nACGT\n+\nIIII\n@r2\nAC\n+r2\nII\n",
        "gt_in_seq.fa": b">a\nAC\n>\nGG\n",
    }
    for name, data in cases.items():
        p = tmp_path / name
        p.write_bytes(data)
        cols = run(["_parse", p]).stdout.decode().strip().split("\t")
        seq, off = ok.parse_fasta_bytes(data)
        assert int(cols[1]) == len(off) - 1, name
        assert int(cols[2]) == len(seq), name
        assert [int(x) for x in cols[5:]] == [int(x) for x in off[1:]], name


def test_on_disk_structs_match_the_reference_headers():
    # tests/golden/layout.txt: sizeof / offsetof of sketchInfo_t, dim_shuffle_stat_t, co_dstat_t taken from the
    # reference's own headers (oracle/_ref/ref_driver layout)
    want = open(os.path.join(GOLDEN, "layout.txt")).read()
    assert run(["_layout"]).stdout.decode() == want
    assert want.splitlines()[0].split()[:2] == ["sketchInfo_t", "20"]


def test_record_reader_matches_the_real_kseq(tmp_path):
    """tests/golden/kseq/expected.tsv was written by the reference's own kseq.h (src/kseq.h, looped like
    src/sketch.cpp:462-479) through oracle/_ref/ref_driver kseq: records, bases, sequence bytes, quality
    bytes and record boundaries of 29 well-formed and malformed inputs, plain and gzip'd"""
    d = os.path.join(GOLDEN, "kseq")
    want = [l.split("\t") for l in open(os.path.join(d, "expected.tsv")).read().split("\n") if l and not l.startswith("#")]
    assert len(want) == 29
    out = run(["_parse"] + [os.path.join(d, w[0]) for w in want]).stdout.decode().strip().split("\n")
    for line, w in zip(out, want):
        cols = line.split("\t")
        assert os.path.basename(cols[0]) == w[0]
        assert cols[1:] == w[1:], w[0]


def test_info_and_merge(tmp_path):
    src = os.path.join(GOLDEN, "dist", "qry.sketch")
    info, names, h, off = ok.read_sketches32(src)
    run(["info", "-i", src, "-o", tmp_path / "i.txt", "-F"])
    lines = (tmp_path / "i.txt").read_text().split("\n")
    assert lines[0] == "the number of sketches are: %d" % len(names)   # src/subCommand.cpp:93
    assert lines[1] == "%s\t%d" % (names[0], off[1] - off[0])
    assert lines[2].split("\t")[:3] == [str(x) for x in h[:3]]
    lst = tmp_path / "m.list"
    lst.write_text(src + "\n" + src + "\n")
    run(["merge", "-i", lst, "-o", tmp_path / "m.sketch"])
    info2, names2, h2, off2 = ok.read_sketches32(str(tmp_path / "m.sketch"))
    assert names2 == names + names and np.array_equal(h2, np.concatenate([h, h]))
    assert info2.genomeNumber == 2 * len(names) and info2.id == info.id


def test_union_and_sub_set_algebra(tmp_path):
    # src/subCommand.cpp:307-794: union = one sketch with the ascending union of all hashes,
    # sub = query sketches minus every hash present in any reference sketch (order kept)
    d = os.path.join(GOLDEN, "dist")
    _, rnames, rh, roff = ok.read_sketches32(os.path.join(d, "ref.sketch"))
    _, qnames, qh, qoff = ok.read_sketches32(os.path.join(d, "qry.sketch"))
    run(["union", "-i", os.path.join(d, "ref.sketch"), "-o", tmp_path / "u.sketch"])
    info, names, h, off = ok.read_sketches32(str(tmp_path / "u.sketch"))
    assert info.genomeNumber == 1 and names == [os.path.join(d, "ref.sketch") + " merged sketches"]
    assert np.array_equal(h, np.unique(rh))
    run(["sub", "--rs", os.path.join(d, "ref.sketch"), "--qs", os.path.join(d, "qry.sketch"), "-o", tmp_path / "s.sketch"])
    info, names, h, off = ok.read_sketches32(str(tmp_path / "s.sketch"))
    assert names == qnames
    refset = set(rh.tolist())
    for i in range(len(qnames)):
        want = [x for x in qh[int(qoff[i]):int(qoff[i + 1])].tolist() if x not in refset]
        assert h[int(off[i]):int(off[i + 1])].tolist() == want


@pytest.mark.parametrize("tag", ["32", "64"])
def test_info_union_sub_merge_equal_the_reference_outputs(tmp_path, tag):
    # tests/golden/f4: what the reference's own command_info / command_union / command_sub / command_merge
    # (src/subCommand.cpp:70-147, :307-543, :545-794, :796-892, compiled unmodified: tests/golden/make_f4_golden.py) wrote
    # for the committed 32- and 64-bit sketch fixtures.  The paths are relative to tests/golden, like the generator's:
    # the reference stores the path strings in its outputs.
    man = json.load(open(os.path.join(GOLDEN, "f4", "manifest.json")))[tag]
    ref, qry, lst = man["ref"], man["qry"], man["list"]
    want = lambda name: open(os.path.join(GOLDEN, "f4", name), "rb").read()
    run(["info", "-i", qry, "-o", tmp_path / "i.txt"], cwd=GOLDEN)
    assert (tmp_path / "i.txt").read_bytes() == want("info%s.txt" % tag)
    run(["info", "-i", qry, "-o", tmp_path / "d.txt", "-F"], cwd=GOLDEN)
    assert (tmp_path / "d.txt").read_bytes() == want("info%s_detail.txt" % tag)
    run(["union", "-i", ref, "-o", tmp_path / "u.sketch"], cwd=GOLDEN)
    assert (tmp_path / "u.sketch").read_bytes() == want("union%s.sketch" % tag)
    run(["sub", "--rs", ref, "--qs", qry, "-o", tmp_path / "s.sketch"], cwd=GOLDEN)
    assert (tmp_path / "s.sketch").read_bytes() == want("sub%s.sketch" % tag)
    run(["merge", "-i", lst, "-o", tmp_path / "m.sketch"], cwd=GOLDEN)
    assert (tmp_path / "m.sketch").read_bytes() == want("merge%s.sketch" % tag)


def test_convert_to_kssd_layout_and_back(tmp_path):
    # SURVEY Appendix A.5 / src/sketch.cpp:1288-1365
    src = os.path.join(GOLDEN, "dist", "qry.sketch")
    info, names, h, off = ok.read_sketches32(src)
    kd = tmp_path / "kssd"
    run(["convert", "--reverse", "-i", src, "-o", kd])
    stat = (kd / "cofiles.stat").read_bytes()
    n = len(names)
    assert len(stat) == 32 + 4 * n + 256 * n
    shuf_id, = np.frombuffer(stat[0:4], "<u4")
    kmerlen, dim_rd_len, comp_num, infile_num = np.frombuffer(stat[8:24], "<i4")
    all_ctx, = np.frombuffer(stat[24:32], "<u8")
    assert (shuf_id, stat[4], kmerlen, dim_rd_len, comp_num, infile_num, all_ctx) == (
        info.id, 0, 2 * info.half_k, 2 * info.drlevel, 1, n, len(h))
    assert np.array_equal(np.frombuffer(stat[32:32 + 4 * n], "<u4"), np.diff(off))
    assert stat[32 + 4 * n:32 + 4 * n + 256].rstrip(b"\0").decode() == names[0]
    assert np.array_equal(np.frombuffer((kd / "combco.index.0").read_bytes(), "<u8"), off)
    assert np.array_equal(np.frombuffer((kd / "combco.0").read_bytes(), "<u4"), h)
    # and back (query mode: no index, so no GPU needed); half_subk comes back as 6 (:1197)
    run(["convert", "-q", "-i", kd, "-o", tmp_path / "back"])
    info2, names2, h2, off2 = ok.read_sketches32(str(tmp_path / "back.sketch"))
    assert names2 == names and np.array_equal(h2, h) and np.array_equal(off2, off)
    assert (info2.half_k, info2.half_subk, info2.drlevel) == (info.half_k, 6, info.drlevel)


# --------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_cli_sketch_alldist_dist_end_to_end(tmp_path):
    d = os.path.join(GOLDEN, "sketch")
    exp = json.load(open(os.path.join(d, "expected.json")))
    k, s, l = exp["half_k"], exp["half_subk"], exp["drlevel"]
    shuf = tmp_path / "t.shuf"
    run(["shuffle", "-k", k, "-s", s, "-l", l, "-o", shuf])
    # FASTA files only: like the reference (isFastaList, src/sketch.cpp:68-80) the tool rejects
    # a list that mixes FASTA and FASTQ
    files = sorted(f for f in exp["files"] if f.endswith(".fa"))
    lst = tmp_path / "g.list"
    lst.write_text("".join(os.path.join(d, f) + "\n" for f in files))
    run(["sketch", "-i", lst, "-L", shuf, "-o", tmp_path / "g"], cwd=tmp_path)
    info, names, h, off = ok.read_sketches32(str(tmp_path / "g.sketch"))
    assert (info.half_k, info.half_subk, info.drlevel, info.genomeNumber) == (k, s, l, len(files))
    assert names == [os.path.join(d, f) for f in files]
    for i, f in enumerate(files):
        assert h[int(off[i]):int(off[i + 1])].tolist() == exp["files"][f]["hashes"], f
    # .dict/.index written by the tool == transSketches layout
    bits = 4 * (k - l)
    postings, counts = ok.index_build32(h, off, bits)
    p2, c2 = ok.read_index32(str(tmp_path / "g.sketch.dict"), str(tmp_path / "g.sketch.index"))
    assert np.array_equal(p2, postings) and np.array_equal(c2, counts)
    assert os.path.getsize(tmp_path / "g.sketch.index") == 16 + 4 * (1 << bits)
    # alldist from the sketch file and straight from the list give the same text as the oracle
    want_hits, _ = ok.index_dist32(counts, bits, postings, np.diff(off).astype(np.uint32), h, off, 1, 0, 2 * k, 0.2)
    want = sorted(ok.alldist_text(names, want_hits))
    run(["alldist", "-i", tmp_path / "g.sketch", "-D", 0.2, "-o", "a.out"], cwd=tmp_path)
    got = (tmp_path / "a.out").read_text().split("\n")
    assert got[0] == " genome0\tgenome1\tcommon|size0|size1\tjaccard\tmashD"
    assert sorted(x + "\n" for x in got[1:] if x) == want
    run(["alldist", "-i", lst, "-L", shuf, "-D", 0.2, "-o", "b.out"], cwd=tmp_path)
    assert sorted(x + "\n" for x in (tmp_path / "b.out").read_text().split("\n")[1:] if x) == want


@pytest.mark.gpu
def test_cli_sketch_streaming_pipeline_many_batches_and_a_big_file(tmp_path):
    """the tool's sketch pipeline beyond one batch: 36 genomes of 5 Mb (several page-locked staging batches,
    parser threads and the GPU thread overlapping) plus one 70 Mb genome of 7 records, which is read and parsed
    by all threads and staged in ordinary memory; every hash set must equal the oracle's"""
    from rabbitkssd_amd import synth
    k, s, l = 10, 6, 3
    shuf = tmp_path / "L3K10.shuf"
    run(["shuffle", "-k", k, "-s", s, "-l", l, "-o", shuf])
    param, table = ok.init_param(k, s, l), ok.shuffle_table(k, s, l)
    files, seqs = [], []
    for name, bases in synth.clade_genome_set(36, 5_000_000):
        p = tmp_path / (name.replace("/", "_") + ".fa")
        p.write_bytes(synth.fasta_text(name, bases))
        files.append(str(p))
        seqs.append([bases])
    big = [synth.clade_genome(900 + r, 0, 10_000_000) for r in range(7)]      # 7 records of 10 Mb
    p = tmp_path / "big.fa"
    p.write_bytes(b"".join(synth.fasta_text("chr%d" % r, b, width=60) for r, b in enumerate(big)))
    assert p.stat().st_size > (64 << 20)
    files.insert(17, str(p))
    seqs.insert(17, big)
    lst = tmp_path / "g.list"
    lst.write_text("".join(f + "\n" for f in files))
    run(["sketch", "-i", lst, "-L", shuf, "-o", tmp_path / "g", "-t", 8], cwd=tmp_path)
    info, names, h, off = ok.read_sketches32(str(tmp_path / "g.sketch"))
    assert names == files and info.genomeNumber == len(files)
    for g in (0, 5, 16, 17, 18, 36):
        recs = seqs[g]
        seq = np.concatenate(recs)
        rec_off = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.uint64)
        want = ok.sketch_records(param, table, seq, rec_off)
        assert np.array_equal(h[int(off[g]):int(off[g + 1])].astype(np.uint64), want), g


@pytest.mark.gpu
@pytest.mark.parametrize("piece_kb", [0, 193, 1021])
def test_cli_big_fasta_streamed_in_pieces_equals_the_whole_file_path(tmp_path, piece_kb):
    """One big plain FASTA file is streamed: cut into pieces at line starts, parsed piece by piece into page-locked
    buffers while earlier pieces are uploaded; a piece starts with the last k-1 bases in front of it, so that every
    window is seen exactly once.  Tiny pieces (RK_BIG_PIECE_KB) put piece boundaries everywhere: next to headers, inside
    records shorter than k, around empty lines and N runs, in lines of every width.  The hash set must be the oracle's
    and the whole-file path's (RK_BIG_WHOLE=1)."""
    k, s, l = 8, 5, 2
    shuf = tmp_path / "L2K8.shuf"
    run(["shuffle", "-k", k, "-s", s, "-l", l, "-o", shuf])
    param, table = ok.init_param(k, s, l), ok.shuffle_table(k, s, l)
    rng = np.random.default_rng(2026 + piece_kb)
    lut = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)
    recs, text = [], []
    total = 0
    while total < (66 << 20):
        n = int(rng.choice([1, 7, 15, 16, 17, 40, 5000, 200_000, 3_000_000]))
        codes = rng.integers(0, 8, n, dtype=np.uint8)
        if n > 1000 and rng.random() < 0.5:
            a = int(rng.integers(0, n - 100))
            codes[a:a + int(rng.integers(1, 90))] = 8               # an N run
        bases = lut[codes]
        recs.append(bases)
        width = int(rng.choice([60, 61, 80, 200, 100000]))
        body = bases.tobytes()
        lines = [body[i:i + width] for i in range(0, n, width)]
        if rng.random() < 0.2:
            lines.insert(len(lines) // 2, b"")                       # an empty line inside the record
        text.append(b">r%d some comment\n" % len(recs) + b"\n".join(lines) + b"\n")
        total += n
    big = tmp_path / "big.fa"
    big.write_bytes(b"".join(text))
    small = tmp_path / "small.fa"
    small.write_bytes(b">s\nACGTACGTACGTACGTACGTAAAACCCCGGGGTTTT\n")
    lst = tmp_path / "g.list"
    lst.write_text("%s\n%s\n" % (small, big))
    env = dict(os.environ)
    if piece_kb:
        env["RK_BIG_PIECE_KB"] = str(piece_kb)
    p = subprocess.run([TOOL, "sketch", "-i", str(lst), "-L", str(shuf), "-o", str(tmp_path / "st"), "-t", "6", "-q"], cwd=tmp_path,
                       env=dict(env, RK_TIMING="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    assert b"big file:" in p.stderr                                   # the streamed path ran
    _, names, h, off = ok.read_sketches32(str(tmp_path / "st.sketch"))
    seq = np.concatenate(recs)
    rec_off = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.uint64)
    want = ok.sketch_records(param, table, seq, rec_off)
    assert np.array_equal(h[int(off[1]):int(off[2])].astype(np.uint64), want)
    if piece_kb == 0:
        p = subprocess.run([TOOL, "sketch", "-i", str(lst), "-L", str(shuf), "-o", str(tmp_path / "wh"), "-t", "6", "-q"], cwd=tmp_path,
                           env=dict(os.environ, RK_BIG_WHOLE="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()
        assert (tmp_path / "wh.sketch").read_bytes() == (tmp_path / "st.sketch").read_bytes()


@pytest.mark.gpu
def test_cli_sketch_gzip_inputs_incl_multi_member(tmp_path):
    """.gz genomes in the streaming pipeline: a single-member file (its trailer gives the slot size) and a
    three-member file whose trailer understates the size -- the slot overflows and the genome takes the
    whole-file path, spliced back in list order"""
    import gzip
    from rabbitkssd_amd import synth
    k, s, l = 10, 6, 3
    shuf = tmp_path / "L3K10.shuf"
    run(["shuffle", "-k", k, "-s", s, "-l", l, "-o", shuf])
    param, table = ok.init_param(k, s, l), ok.shuffle_table(k, s, l)
    genomes = synth.clade_genome_set(5, 1_500_000)
    files = []
    for i, (name, bases) in enumerate(genomes):
        text = synth.fasta_text(name, bases)
        if i == 1:
            p = tmp_path / (name + ".fa.gz")
            p.write_bytes(gzip.compress(text, mtime=0))
        elif i == 3:  # three members; the last one is the smallest
            p = tmp_path / (name + ".fa.gz")
            cut1, cut2 = len(text) // 2, len(text) - 1000
            p.write_bytes(gzip.compress(text[:cut1], mtime=0) + gzip.compress(text[cut1:cut2], mtime=0) +
                          gzip.compress(text[cut2:], mtime=0))
        else:
            p = tmp_path / (name + ".fa")
            p.write_bytes(text)
        files.append(str(p))
    lst = tmp_path / "g.list"
    lst.write_text("".join(f + "\n" for f in files))
    run(["sketch", "-i", lst, "-L", shuf, "-o", tmp_path / "g", "-t", 4], cwd=tmp_path)
    info, names, h, off = ok.read_sketches32(str(tmp_path / "g.sketch"))
    assert names == files
    for g, (name, bases) in enumerate(genomes):
        want = ok.sketch_records(param, table, bases, np.array([0, len(bases)], dtype=np.uint64))
        assert np.array_equal(h[int(off[g]):int(off[g + 1])].astype(np.uint64), want), g


@pytest.mark.gpu
def test_cli_big_gzip_and_fastq_files_are_streamed(tmp_path):
    """One big gzip'ed FASTA file (three members, 210 MB of text) and one big FASTQ file (plain and gzip'ed) are read by a
    reader thread piece by piece -- inflate, parse into page-locked buffers, upload overlap -- instead of being inflated and
    parsed whole (the role of the RabbitFX producer, src/sketch.cpp:380-450 / :658-737).  The device buffer is forced to
    start small so that it has to grow; small pieces (RK_BIG_PIECE_KB) put boundaries everywhere.  Hash sets == oracle,
    also with the quality gate and -n 2 (occurrences counted over the whole file)."""
    import gzip
    k, s, l = 8, 5, 2
    shuf = tmp_path / "L2K8.shuf"
    run(["shuffle", "-k", k, "-s", s, "-l", l, "-o", shuf])
    param, table = ok.init_param(k, s, l), ok.shuffle_table(k, s, l)
    rng = np.random.default_rng(99)
    lut = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)
    # ---- FASTA: records of every size, 60 / 80 / 200-column lines, N runs, 210 MB of text in three gzip members
    recs, text = [], []
    total = 0
    while total < (205 << 20):
        n = int(rng.choice([1, 15, 16, 17, 300, 5000, 1_000_000, 9_000_000]))
        codes = rng.integers(0, 8, n, dtype=np.uint8)
        if n > 1000 and rng.random() < 0.5:
            a = int(rng.integers(0, n - 100))
            codes[a:a + int(rng.integers(1, 90))] = 8
        bases = lut[codes]
        recs.append(bases)
        width = int(rng.choice([60, 80, 200]))
        body = bases.tobytes()
        text.append(b">r%d c\n" % len(recs) + b"\n".join(body[i:i + width] for i in range(0, n, width)) + b"\n")
        total += n
    blob = b"".join(text)
    c1, c2 = len(blob) // 3, len(blob) - 777
    big = tmp_path / "big.fa.gz"
    big.write_bytes(gzip.compress(blob[:c1], 1, mtime=0) + gzip.compress(blob[c1:c2], 1, mtime=0) + gzip.compress(blob[c2:], 1, mtime=0))
    small = tmp_path / "small.fa"
    small.write_bytes(b">s\nACGTACGTACGTACGTACGTAAAACCCCGGGGTTTT\n")
    lst = tmp_path / "g.list"
    lst.write_text("%s\n%s\n" % (small, big))
    seq = np.concatenate(recs)
    rec_off = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.uint64)
    want = ok.sketch_records(param, table, seq, rec_off)
    for env in ({"RK_BIG_DEV_MB": "64"}, {"RK_BIG_PIECE_KB": "997", "RK_BIG_DEV_MB": "16"}):
        p = subprocess.run([TOOL, "sketch", "-i", str(lst), "-L", str(shuf), "-o", str(tmp_path / "gz"), "-t", "6", "-q"], cwd=tmp_path,
                           env=dict(os.environ, RK_TIMING="1", **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        assert b"big file (sequential reader):" in p.stderr
        _, names, h, off = ok.read_sketches32(str(tmp_path / "gz.sketch"))
        assert np.array_equal(h[int(off[1]):int(off[2])].astype(np.uint64), want), env
    # ---- FASTQ: 150-base reads with qualities, some reads with N; plain and gzip'ed; -Q 36 -n 2 and the defaults
    n_reads = 500_000
    codes = rng.integers(0, 4, (n_reads, 150), dtype=np.uint8)
    codes[rng.random((n_reads, 150)) < 0.001] = 8
    codes[n_reads // 2:] = codes[: n_reads - n_reads // 2]            # every read twice: -n 2 keeps what -n 1 keeps
    reads = lut[codes]
    quals = rng.integers(33, 75, (n_reads, 150), dtype=np.uint8)
    quals[::7, 0] = ord("@")                                           # quality lines that start like a header line
    lines = []
    for i in range(n_reads):
        lines.append(b"@read%d/1\n" % i + reads[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n")
    fq_text = b"".join(lines)
    assert len(fq_text) > (150 << 20)
    fq = tmp_path / "reads.fq"
    fq.write_bytes(fq_text)
    fqz = tmp_path / "readsz.fq.gz"
    fqz.write_bytes(gzip.compress(fq_text, 1, mtime=0))
    smallq = tmp_path / "small.fq"
    smallq.write_bytes(b"@q\nACGTACGTACGTACGTACGTAAAACCCCGGGGTTTT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n")
    off_q = (np.arange(n_reads + 1, dtype=np.uint64) * 150)
    for least_qual, least_num in ((0, 1), (36, 2)):
        want = ok.sketch_records_fastq(param, table, reads.reshape(-1), quals.reshape(-1), off_q, least_qual, least_num)
        assert len(want) > 1000
        for f, env in ((fq, {"RK_BIG_PIECE_KB": "2053"}), (fqz, {"RK_BIG_DEV_MB": "32"})):
            lst.write_text("%s\n%s\n" % (smallq, f))
            p = subprocess.run([TOOL, "sketch", "-i", str(lst), "-L", str(shuf), "-o", str(tmp_path / "fq"), "-t", "6", "-q", "-Q", str(least_qual),
                                "-n", str(least_num)], cwd=tmp_path, env=dict(os.environ, RK_TIMING="1", **env), stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE)
            assert p.returncode == 0, p.stderr.decode()[-2000:]
            assert b"big file (sequential reader):" in p.stderr
            _, names, h, off = ok.read_sketches32(str(tmp_path / "fq.sketch"))
            assert np.array_equal(h[int(off[1]):int(off[2])].astype(np.uint64), want), (str(f), least_qual, least_num)


@pytest.mark.gpu
def test_cli_dist_matches_reference_text(tmp_path):
    d = os.path.join(GOLDEN, "dist")
    man = json.load(open(os.path.join(d, "manifest.json")))
    ref = tmp_path / "ref.sketch"
    ref.write_bytes(open(os.path.join(d, "ref.sketch"), "rb").read())
    for case in man["cases"]:
        want = open(os.path.join(d, case["file"])).read().split("\n")[:-1]
        if case["cmd"] == "alldist":
            run(["alldist", "-i", ref, "-D", case["max_dist"], "-M", case["metric"], "-o", "o.txt"], cwd=tmp_path)
            got = (tmp_path / "o.txt").read_text().split("\n")
            assert sorted(x for x in got[1:] if x) == want, case["file"]
        else:
            args = ["dist", "-r", ref, "-q", os.path.join(d, "qry.sketch"), "-D", case["max_dist"], "-M",
                    case["metric"], "-o", "o.txt"]
            if case["max_neighbor"]:
                args += ["-N", case["max_neighbor"]]
            run(args, cwd=tmp_path)
            got = (tmp_path / "o.txt").read_text().split("\n")
            assert [x for x in got[1:] if x] == want, case["file"]   # same order as the reference at -t 1
    assert os.path.exists(str(ref) + ".dict") and os.path.exists(str(ref) + ".index")


@pytest.mark.gpu
def test_cli_fastq_list_with_quality_and_count_options(tmp_path):
    """sketch -Q/-n on a FASTQ list (sketchFastqFile, src/sketch.cpp:596-890) == oracle restatement"""
    from test_gpu_parity import make_fastq
    k, s, l = 8, 5, 2
    shuf = tmp_path / "t.shuf"
    run(["shuffle", "-k", k, "-s", s, "-l", l, "-o", shuf])
    param, table = ok.init_param(k, s, l), ok.shuffle_table(k, s, l)
    paths = []
    for seed in (5, 6):
        p = tmp_path / ("reads%d.fq" % seed)
        p.write_bytes(make_fastq(seed))
        paths.append(str(p))
    lst = tmp_path / "fq.list"
    lst.write_text("\n".join(paths) + "\n")
    run(["sketch", "-i", lst, "-L", shuf, "-o", tmp_path / "fq", "-Q", 45, "-n", 2, "-q"], cwd=tmp_path)
    info, names, h, off = ok.read_sketches32(str(tmp_path / "fq.sketch"))
    assert names == paths and not os.path.exists(tmp_path / "fq.sketch.dict")   # -q: no index
    for i, p in enumerate(paths):
        sq, ql, o = ok.parse_fastq_bytes(open(p, "rb").read())
        want = ok.sketch_records_fastq(param, table, sq, ql, o, 45, 2)
        assert h[int(off[i]):int(off[i + 1])].astype(np.uint64).tolist() == want.tolist()
    # a list mixing FASTA and FASTQ is rejected like the reference does
    mixed = tmp_path / "mixed.list"
    mixed.write_text(paths[0] + "\n" + os.path.join(GOLDEN, "sketch", "c0_s0.fa") + "\n")
    assert run(["sketch", "-i", mixed, "-L", shuf, "-o", tmp_path / "m"], check=False).returncode == 1


# --------------------------------------------------------------------------- D6: sub-files and their index (CPU)
def _fake_hits(n_genomes, rng):
    """a (row, col)-sorted structured hit array with ~60 % of the pairs, some rows without any hit"""
    from rabbitkssd_amd import capi
    rows, cols = np.triu_indices(n_genomes, 1)
    keep = rng.random(len(rows)) < 0.6
    keep &= ~np.isin(rows, [3, 17, n_genomes - 2])
    rows, cols = rows[keep], cols[keep]
    h = np.zeros(len(rows), dtype=capi.HIT_DTYPE)
    h["row"], h["col"] = rows, cols
    h["common"] = rng.integers(1, 100, size=len(rows))
    h["size0"], h["size1"] = 100 + h["row"], 100 + h["col"]
    h["jorc"] = h["common"] / (h["size0"] + h["size1"] - h["common"])
    h["dist"] = rng.random(len(rows))
    return h


@pytest.mark.parametrize("parts,threads", [(1, 1), (1, 5), (3, 6), (4, 2)])
def test_distance_text_single_file_and_subfile_layout(tmp_path, parts, threads):
    """the writer alone (`rabbit_kssd _format`, no GPU): below the merge limit one file with the reference's header and
    lines (src/dist.cpp:286-310); above it -- limit injected through RK_DIST_MAX_MERGE_BYTES -- the per-worker sub-files
    stay under <out>.dir/ and <out>.index lists every row exactly once with its sub-file (:311-335)"""
    rng = np.random.default_rng(parts * 10 + threads)
    n = 70
    names = ["dir/g%02d.fna" % i for i in range(n)]
    hits = _fake_hits(n, rng)
    (tmp_path / "names.txt").write_text("".join(x + "\n" for x in names))
    (tmp_path / "hits.bin").write_bytes(hits.tobytes())
    want = sorted(ok.alldist_text(names, hits))
    run(["_format", "alldist", "names.txt", "hits.bin", "one.out", parts, threads], cwd=tmp_path)
    got = (tmp_path / "one.out").read_text().split("\n")
    assert got[0] == " genome0\tgenome1\tcommon|size0|size1\tjaccard\tmashD"
    body = [x + "\n" for x in got[1:] if x]
    assert sorted(body) == want and not (tmp_path / "one.out.dir").exists()
    if parts == 1:
        assert body == ok.alldist_text(names, hits)   # one worker: rows ascending
    env = dict(os.environ, RK_DIST_MAX_MERGE_BYTES="20000")
    p = subprocess.run([TOOL, "_format", "alldist", "names.txt", "hits.bin", "big.out", str(parts), str(threads)], cwd=tmp_path,
                       env=env, capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    assert not (tmp_path / "big.out").exists()
    idx = (tmp_path / "big.out.index").read_text().split("\n")
    assert idx[0] == "genomeName\tdistFileName" and idx[-1] == ""
    pairs = [x.split("\t") for x in idx[1:-1]]
    assert sorted(a for a, _ in pairs) == sorted(names)          # every row once, with or without hits
    files = []
    for _, f in pairs:
        if f not in files:
            files.append(f)
    assert all(f.startswith("big.out.dir/big.out.") for f in files)
    assert sorted(os.listdir(tmp_path / "big.out.dir")) == sorted(os.path.basename(f) for f in
                                                                 ["big.out.dir/big.out.%d" % t for t in range(len(os.listdir(tmp_path / "big.out.dir")))])
    where = dict(pairs)
    cat = []
    for t in range(len(os.listdir(tmp_path / "big.out.dir"))):
        f = "big.out.dir/big.out.%d" % t
        for line in (tmp_path / f).read_text().split("\n"):
            if line:
                cat.append(line + "\n")
                assert where[line.split("\t")[1]] == f            # alldist: the second name is the row genome
    assert cat == body                                            # sub-files in order == the merged file


@pytest.mark.gpu
def test_cli_multi_gpu_row_shards_reproduce_the_reference_text(tmp_path):
    """`--gpus N --same-device`: N contexts and host threads in one process (all on the one card of the GPU box), every
    context building its own index from the host's sketches (2 GPUs) or the first one's replicated with rk_index_broadcast
    (RK_MULTI_BROADCAST=1, 3 GPUs), block-cyclic rows (alldist) / contiguous query blocks (dist); the text must be
    the real reference's, also when the output is kept as sub-files"""
    d = os.path.join(GOLDEN, "dist")
    man = json.load(open(os.path.join(d, "manifest.json")))
    for f in ("ref.sketch", "qry.sketch"):
        (tmp_path / f).write_bytes(open(os.path.join(d, f), "rb").read())
    for case in man["cases"]:
        want = open(os.path.join(d, case["file"])).read().split("\n")[:-1]
        for gpus in (2, 3):
            os.environ["RK_MULTI_BROADCAST"] = "1" if gpus == 3 else "0"
            if case["cmd"] == "alldist":
                run(["alldist", "-i", "ref.sketch", "-D", case["max_dist"], "-M", case["metric"], "-o", "o.txt", "--gpus", gpus,
                     "--same-device"], cwd=tmp_path)
                lines = (tmp_path / "o.txt").read_text().split("\n")[:-1]
                assert sorted(lines[1:]) == want, (case["file"], gpus)
            else:
                args = ["dist", "-r", "ref.sketch", "-q", "qry.sketch", "-D", case["max_dist"], "-M", case["metric"], "-o", "o.txt",
                        "--gpus", gpus, "--same-device"]
                if case["max_neighbor"]:
                    args += ["-N", case["max_neighbor"]]
                run(args, cwd=tmp_path)
                lines = (tmp_path / "o.txt").read_text().split("\n")[:-1]
                assert lines[1:] == want, (case["file"], gpus)     # queries in order, -N heap order per query
    os.environ.pop("RK_MULTI_BROADCAST", None)
    # (round 5) with two GPUs and the .dict / .index pair in place a sparse alldist takes the SHARDED flow -- every GPU builds the
    # lists of its hash range, the tile records change hands, every GPU joins its rows --, RK_MULTI_REPLICATE=1 the old one: same text
    case = [c for c in man["cases"] if c["cmd"] == "alldist" and c["max_dist"] < 1.0 and c["metric"] == 0][0]
    want = open(os.path.join(d, case["file"])).read().split("\n")[:-1]
    for repl in ("0", "1"):
        p = subprocess.run([TOOL, "alldist", "-i", "ref.sketch", "-D", str(case["max_dist"]), "-o", "s.txt", "--gpus", "2", "--same-device", "-t", "4"],
                           cwd=tmp_path, env=dict(os.environ, RK_TIMING="1", RK_MULTI_REPLICATE=repl), capture_output=True)
        assert p.returncode == 0, p.stderr.decode()
        assert (b"tile records exchanged" in p.stderr) == (repl == "0") and b"refused" not in p.stderr
        assert sorted((tmp_path / "s.txt").read_text().split("\n")[1:-1]) == want
    # the same through the sub-file layout (src/dist.cpp:311-335)
    case = [c for c in man["cases"] if c["cmd"] == "alldist" and c["max_dist"] == 1.5 and c["metric"] == 0][0]
    want = open(os.path.join(d, case["file"])).read().split("\n")[:-1]
    env = dict(os.environ, RK_DIST_MAX_MERGE_BYTES="4096")
    p = subprocess.run([TOOL, "alldist", "-i", "ref.sketch", "-D", "1.5", "-o", "big.txt", "--gpus", "2", "--same-device", "-t", "4"],
                       cwd=tmp_path, env=env, capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    idx = (tmp_path / "big.txt.index").read_text().split("\n")
    assert idx[0] == "genomeName\tdistFileName" and len(idx) - 2 == 45      # every reference genome once
    cat = []
    for f in sorted(set(x.split("\t")[1] for x in idx[1:-1])):
        cat += [x for x in (tmp_path / f).read_text().split("\n") if x]
    assert sorted(cat) == want and not (tmp_path / "big.txt").exists()
