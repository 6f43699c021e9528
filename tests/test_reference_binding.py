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
