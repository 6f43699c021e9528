"""The reference-side binding (integration/gpu_backend.cpp: the bodies of index_tridist / index_dist
replaced by calls into the C ABI, compiled against the REFERENCE's own headers into
oracle/_ref/ref_driver_gpu by `make -C oracle ref_gpu`) must write the text the real CPU functions
wrote -- the golden fixtures under tests/golden/dist{,64}."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver_gpu")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def run(args, cwd):
    p = subprocess.run([DRIVER] + [str(a) for a in args], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]


@pytest.mark.parametrize("sub,ref,qry", [("dist", "ref.sketch", "qry.sketch"), ("dist64", "ref64.sketch", "qry64.sketch")])
def test_binding_reproduces_the_reference_text(tmp_path, sub, ref, qry):
    if not os.path.exists(DRIVER):
        pytest.skip("oracle/_ref/ref_driver_gpu not built (needs /root/reference at build time)")
    d = os.path.join(GOLDEN, sub)
    man = json.load(open(os.path.join(d, "manifest.json")))
    for case in man["cases"]:
        want = open(os.path.join(d, case["file"])).read().split("\n")[:-1]
        if case["cmd"] == "alldist":
            run(["alldist", tmp_path, os.path.join(d, ref), "out.txt", case["max_dist"], case["metric"], 1], tmp_path)
            lines = (tmp_path / "out.txt").read_text().split("\n")[:-1]
            assert lines[0] == " genome0\tgenome1\tcommon|size0|size1\tjaccard\tmashD"
            assert sorted(lines[1:]) == want, case["file"]   # the fixture stores alldist lines sorted
        else:
            n = case["max_neighbor"]
            run(["dist", tmp_path, os.path.join(d, ref), os.path.join(d, qry), "out.txt", case["max_dist"], n,
                 1 if n else 0, case["metric"], 1], tmp_path)
            lines = (tmp_path / "out.txt").read_text().split("\n")[:-1]
            assert lines[1:] == want, case["file"]           # dist: the reference's own order, -N included


# ---------------------------------------------------------------------------------------------------------------------
# the sketch half: integration/gpu_sketch_backend.cpp (sketchFastaFile / transSketches of src/sketch.h:62,66 over the C
# ABI; kseq.h, saveSketches and readSketches stay the reference's own) in oracle/_ref/ref_sketch_driver_gpu
SKETCH_DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_sketch_driver_gpu")
TOOL = os.path.join(ROOT, "rabbitkssd_amd", "rabbit_kssd")


def test_sketch_binding_reproduces_the_reference_hash_sets_and_index_files(tmp_path):
    import hashlib
    if not os.path.exists(SKETCH_DRIVER):
        pytest.skip("oracle/_ref/ref_sketch_driver_gpu not built (needs /root/reference at build time)")
    d = os.path.join(GOLDEN, "sketch_ref")
    inputs = os.path.join(d, "inputs")
    cases = [c for c in json.load(open(os.path.join(d, "expected.json")))["cases"] if c["kind"] == "fasta"]
    assert len(cases) == 5
    for case in cases:
        k, s, l = case["half_k"], case["half_subk"], case["drlevel"]
        tag = "k%ds%dl%d" % (k, s, l)
        shuf = tmp_path / (tag + ".shuf")
        assert subprocess.run([TOOL, "shuffle", "-k", str(k), "-s", str(s), "-l", str(l), "-o", str(shuf)],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode == 0
        names = sorted(case["files"])
        lst = tmp_path / (tag + ".list")
        lst.write_text("".join(n + "\n" for n in names))
        out = tmp_path / (tag + ".sketch")
        p = subprocess.run([SKETCH_DRIVER, "sketch", str(shuf), str(lst), str(out), "1", "0"], cwd=inputs,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        # what the reference's own readSketches makes of the file the binding's saveSketches call wrote
        p = subprocess.run([SKETCH_DRIVER, "dump", str(out)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        rows = p.stdout.decode().split("\n")
        assert rows[0] == "info %d %d %d %d %d" % ((k << 8) + (s << 4) + l, k, s, l, len(names))
        got = {}
        for row in rows[1:]:
            if row:
                f = row.split("\t")
                assert int(f[1]) == len(f) - 2
                got[f[0]] = [int(x) for x in f[2:]]
        assert sorted(got) == names
        for n in names:
            assert got[n] == case["files"][n], (n, tag)     # == the REAL sketchFastaFile's hash set
        if "files_pin" in case:   # .sketch as the real saveSketches writes it, .dict/.index == the REAL transSketches' bytes
            md5 = lambda path: hashlib.md5(open(path, "rb").read()).hexdigest()
            assert md5(out) == case["files_pin"]["sketch_md5"]
            assert md5(str(out) + ".dict") == case["files_pin"]["dict_md5"]
            assert md5(str(out) + ".index") == case["files_pin"]["index_md5"]
