"""Sketch-side fixtures produced by the REFERENCE'S OWN code (tests/golden/make_sketch_golden.py):
tests/golden/sketch_ref/expected.json holds what sketchFastaFile (src/sketch.cpp:455-566) and sketchFastqFile
(:741-866) returned for the committed inputs, the md5 of the .sketch that the real readSketches/saveSketches
round-trips and of the .dict/.index the real transSketches (:970-1017) wrote; tests/golden/kssd holds a Kssd
directory written and read back by the real converters (:1179-1365).

CPU part: the oracle restatement and the host tool's format code against them.
GPU part: the HIP sketcher through the C ABI and through `rabbit_kssd sketch`."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from oracle import oracle as ok

D = os.path.join(GOLDEN, "sketch_ref")
IN = os.path.join(D, "inputs")
KSSD = os.path.join(GOLDEN, "kssd")
TOOL = os.path.join(ROOT, "rabbitkssd_amd", "rabbit_kssd")


def expected():
    e = json.load(open(os.path.join(D, "expected.json")))
    assert e["pinned"] is True
    return e["cases"]


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def tool(args, cwd=None):
    p = subprocess.run([TOOL] + [str(a) for a in args], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    return p


def read_fastq(path):
    return ok.parse_fastq_bytes(open(path, "rb").read())


# ------------------------------------------------------------------------------- CPU: oracle
def test_restatement_equals_reference_fasta_hash_sets():
    n = 0
    for case in expected():
        if case["kind"] != "fasta":
            continue
        k, s, l = case["half_k"], case["half_subk"], case["drlevel"]
        param, table = ok.init_param(k, s, l), ok.shuffle_table(k, s, l)
        for fn, want in case["files"].items():
            seq, off = ok.read_fasta(os.path.join(IN, fn))
            assert [int(x) for x in ok.sketch_records(param, table, seq, off)] == want, (fn, k, s, l)
            n += len(want)
    assert n > 3000


def test_restatement_equals_reference_fastq_hash_sets():
    gates = set()
    for case in expected():
        if case["kind"] != "fastq":
            continue
        k, s, l = case["half_k"], case["half_subk"], case["drlevel"]
        param, table = ok.init_param(k, s, l), ok.shuffle_table(k, s, l)
        q, n = case["least_qual"], case["least_num"]
        gates.add((q, n))
        for fn, want in case["files"].items():
            sq, ql, off = read_fastq(os.path.join(IN, fn))
            assert [int(x) for x in ok.sketch_records_fastq(param, table, sq, ql, off, q, n)] == want, (fn, q, n)
        if (q, n) == (127, 1):
            assert all(len(w) == 0 for w in case["files"].values())
    assert len(gates) == 5


def pinned_fasta_case():
    case = [c for c in expected() if "files_pin" in c][0]
    names = sorted(case["files"])
    parts = [np.array(case["files"][n], dtype=np.uint32) for n in names]
    off = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
    return case, names, np.concatenate(parts), off


def test_restated_sketch_dict_index_files_equal_reference_bytes(tmp_path):
    case, names, hh, off = pinned_fasta_case()
    k, s, l = case["half_k"], case["half_subk"], case["drlevel"]
    sk = str(tmp_path / "x.sketch")
    ok.save_sketches32(sk, k, s, l, names, hh, off)
    assert md5(sk) == case["files_pin"]["sketch_md5"]          # the real readSketches -> saveSketches reproduces it
    bits = 4 * (k - l)
    postings, counts = ok.index_build32(hh, off, bits)
    ok.write_index32(sk + ".dict", sk + ".index", postings, counts, bits)
    assert md5(sk + ".dict") == case["files_pin"]["dict_md5"]    # the real transSketches wrote the same bytes
    assert md5(sk + ".index") == case["files_pin"]["index_md5"]
    assert os.path.getsize(sk + ".index") == case["files_pin"]["index_bytes"]


# ------------------------------------------------------------------------------- CPU: host tool formats
def test_tool_sketch_reader_writer_roundtrip_reference_bytes(tmp_path):
    """read_sketches + save_sketches of the tool (via `merge` of a one-entry list) reproduce the file the
    real saveSketches wrote"""
    (tmp_path / "l").write_text(os.path.join(KSSD, "in.sketch") + "\n")
    tool(["merge", "-i", tmp_path / "l", "-o", tmp_path / "m.sketch"])
    assert (tmp_path / "m.sketch").read_bytes() == open(os.path.join(KSSD, "in.sketch"), "rb").read()
    case = pinned_fasta_case()[0]
    assert md5(os.path.join(KSSD, "in.sketch")) == case["files_pin"]["sketch_md5"]


def test_tool_kssd_convert_equals_reference_converters(tmp_path):
    # .sketch -> Kssd directory (convert_from_RabbitKSSDSketch_to_KssdSketch, src/sketch.cpp:1288-1365)
    tool(["convert", "--reverse", "-i", os.path.join(KSSD, "in.sketch"), "-o", tmp_path / "kd"])
    def normalised(f, raw):
        # cofiles.stat as the reference writes it holds uninitialised memory: the padding after `bool koc`
        # (bytes 5..7, src/sketch.h:38-47) and whatever follows the NUL in each malloc'ed 256-byte name slot
        # (src/sketch.cpp:1319-1320).  Everything else must match byte for byte.
        if f != "cofiles.stat":
            return bytes(raw)
        b = bytearray(raw)
        b[5:8] = b"\0\0\0"
        n = int(np.frombuffer(bytes(b[20:24]), "<i4")[0])
        assert len(b) == 32 + 4 * n + 256 * n
        for i in range(n):
            at = 32 + 4 * n + 256 * i
            end = bytes(b[at:at + 256]).index(b"\0")
            b[at + end:at + 256] = b"\0" * (256 - end)
        return bytes(b)
    for f in ("cofiles.stat", "combco.0", "combco.index.0"):
        assert normalised(f, (tmp_path / "kd" / f).read_bytes()) == \
            normalised(f, open(os.path.join(KSSD, "kssd_dir", f), "rb").read()), f
    # Kssd directory -> .sketch (convertSketch, :1179-1285); -q: no index, no GPU
    tool(["convert", "-q", "-i", os.path.join(KSSD, "kssd_dir"), "-o", tmp_path / "back"])
    assert (tmp_path / "back.sketch").read_bytes() == open(os.path.join(KSSD, "back.sketch"), "rb").read()


# ------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def ctx():
    from rabbitkssd_amd import capi
    return capi.Context(0)


@pytest.mark.gpu
def test_hip_sketch_equals_reference_fasta_hash_sets(ctx):
    from rabbitkssd_amd import capi
    for case in expected():
        if case["kind"] != "fasta":
            continue
        k, s, l = case["half_k"], case["half_subk"], case["drlevel"]
        flt = ctx.filter(capi.params_init(k, s, l), ok.shuffle_table(k, s, l))
        names = sorted(case["files"])
        seqs, rec_off, genome_rec = [], [0], [0]
        for fn in names:
            seq, off = ok.read_fasta(os.path.join(IN, fn))
            seqs.append(seq)
            rec_off.extend((rec_off[-1] + off[1:]).tolist())
            genome_rec.append(len(rec_off) - 1)
        sk = ctx.sketch_batch(flt, np.concatenate(seqs), np.array(rec_off, dtype=np.uint64),
                              np.array(genome_rec, dtype=np.uint64))
        gh, goff = sk.download()
        assert sk.is64 == (k - l > 8)
        for g, fn in enumerate(names):
            assert [int(x) for x in gh[int(goff[g]):int(goff[g + 1])]] == case["files"][fn], (fn, k, s, l)


@pytest.mark.gpu
def test_hip_sketch_equals_reference_fastq_hash_sets(ctx):
    from rabbitkssd_amd import capi
    for case in expected():
        if case["kind"] != "fastq":
            continue
        k, s, l = case["half_k"], case["half_subk"], case["drlevel"]
        flt = ctx.filter(capi.params_init(k, s, l), ok.shuffle_table(k, s, l))
        names = sorted(case["files"])
        seqs, quals, rec_off, genome_rec = [], [], [0], [0]
        for fn in names:
            sq, ql, off = read_fastq(os.path.join(IN, fn))
            seqs.append(sq)
            quals.append(ql)
            rec_off.extend((rec_off[-1] + off[1:]).tolist())
            genome_rec.append(len(rec_off) - 1)
        sk = ctx.sketch_batch_fastq(flt, np.concatenate(seqs), np.concatenate(quals), np.array(rec_off, dtype=np.uint64),
                                    np.array(genome_rec, dtype=np.uint64), case["least_qual"], case["least_num"])
        gh, goff = sk.download()
        for g, fn in enumerate(names):
            assert [int(x) for x in gh[int(goff[g]):int(goff[g + 1])]] == case["files"][fn], (fn, case["least_qual"])


@pytest.mark.gpu
def test_cli_sketch_files_equal_reference_bytes(tmp_path):
    """`rabbit_kssd sketch` on the committed inputs (plain + two-member gzip), run from the inputs directory with a
    relative list like the fixture generator: the .sketch is byte-identical to what the real saveSketches writes for
    sorted sets in list order, .dict/.index byte-identical to the real transSketches' files"""
    case, names, hh, off = pinned_fasta_case()
    k, s, l = case["half_k"], case["half_subk"], case["drlevel"]
    shuf = tmp_path / "t.shuf"
    tool(["shuffle", "-k", k, "-s", s, "-l", l, "-o", shuf])
    lst = tmp_path / "fa.list"
    lst.write_text("".join(n + "\n" for n in names))
    tool(["sketch", "-i", lst, "-L", shuf, "-o", tmp_path / "out"], cwd=IN)
    assert md5(tmp_path / "out.sketch") == case["files_pin"]["sketch_md5"]
    assert md5(tmp_path / "out.sketch.dict") == case["files_pin"]["dict_md5"]
    assert md5(tmp_path / "out.sketch.index") == case["files_pin"]["index_md5"]
    # FASTQ list with -Q / -n
    for fq in (c for c in expected() if c["kind"] == "fastq"):
        fnames = sorted(fq["files"])
        (tmp_path / "fq.list").write_text("".join(n + "\n" for n in fnames))
        tool(["sketch", "-q", "-i", tmp_path / "fq.list", "-L", shuf, "-o", tmp_path / "fq", "-Q", fq["least_qual"],
              "-n", fq["least_num"]], cwd=IN)
        _, n2, h2, o2 = ok.read_sketches32(str(tmp_path / "fq.sketch"))
        assert n2 == fnames
        for g, fn in enumerate(fnames):
            assert [int(x) for x in h2[int(o2[g]):int(o2[g + 1])]] == fq["files"][fn], (fn, fq["least_qual"], fq["least_num"])
