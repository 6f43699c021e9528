#!/usr/bin/env python3
"""Generates tests/golden/f4/ (run in the build container only): the outputs of the REFERENCE'S OWN `info`, `union`, `sub`
and `merge` sub-commands on the committed sketch fixtures (tests/golden/dist: 32-bit hashes, tests/golden/dist64: 64-bit).

oracle/_ref/ref_cmd_driver (make -C oracle ref_cmd) is src/subCommand.cpp compiled unmodified (it needs no RabbitFX
header) on top of the RabbitFX-free object of sketch.cpp and the unmodified dist.cpp / common.cpp / shuffle.cpp; the
driver only calls command_info (src/subCommand.cpp:70-147), command_union (:307-543), command_sub (:545-794) and
command_merge (:796-892).  Paths are given relative to tests/golden, as the test gives them to `rabbit_kssd`, because
the reference stores the path strings in its outputs."""
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DRV = os.path.join(HERE, "..", "..", "oracle", "_ref", "ref_cmd_driver")
OUT = os.path.join(HERE, "f4")

CASES = {"32": ("dist/ref.sketch", "dist/qry.sketch"), "64": ("dist64/ref64.sketch", "dist64/qry64.sketch")}


def run(*args):
    subprocess.run([DRV] + [str(a) for a in args], check=True, cwd=HERE, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def main():
    if not os.path.exists(DRV):
        sys.exit("build the reference driver first: make -C oracle ref_cmd")
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    manifest = {}
    for tag, (ref, qry) in CASES.items():
        lst = "f4/merge%s.list" % tag
        with open(os.path.join(HERE, lst), "w") as f:
            f.write(qry + "\n" + ref + "\n" + qry + "\n")
        run("info", qry, 0, "f4/info%s.txt" % tag)
        run("info", qry, 1, "f4/info%s_detail.txt" % tag)
        run("union", ref, "f4/union%s.sketch" % tag, 4)
        run("sub", ref, qry, "f4/sub%s.sketch" % tag, 2)   # one producer + ONE consumer: with more, the sketches come out in completion order
        run("merge", lst, "f4/merge%s.sketch" % tag, 1)
        manifest[tag] = {"ref": ref, "qry": qry, "list": lst}
    json.dump(manifest, open(os.path.join(OUT, "manifest.json"), "w"), indent=1)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
