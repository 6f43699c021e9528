#!/usr/bin/env python3
"""Generates tests/golden/reflayout/ (run in the build container only): a `.sketch` pair exactly as the REFERENCE writes
them -- hashes in `unordered_set` iteration order (src/sketch.cpp:537-553, the sort is commented out at :553), genomes in
OpenMP completion order (:558-568) -- and the distance texts the reference computes from them.

  ref.sketch / qry.sketch   written by the reference's own sketchFastaFile + saveSketches (oracle/_ref/ref_sketch_driver:
                            the RabbitFX-free lines of sketch.cpp compiled unmodified) from synthetic clade genomes that
                            are generated on the fly and NOT committed
  alldist_*.ref.txt         the reference's index_tridist on ref.sketch (its own transSketches wrote .dict/.index)
  dist_*.ref.txt            the reference's index_dist, ref.sketch x qry.sketch
This is the drop-in input of `alldist` / `dist`: the GPU tests push it through rk_sketches_from_host -> rk_index_build ->
both distance paths and through the command line."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as ok  # noqa: E402
from rabbitkssd_amd import synth  # noqa: E402

SK = os.path.join(ROOT, "oracle", "_ref", "ref_sketch_driver")
REF = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
OUT = os.path.join(HERE, "reflayout")
K, S, L = 8, 5, 2
LENGTH = 400000


def run(exe, *args, cwd=None):
    subprocess.run([exe] + [str(a) for a in args], check=True, cwd=cwd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def write_genomes(tmp, sub, members):
    os.makedirs(os.path.join(tmp, sub), exist_ok=True)
    names = []
    for c, s in members:
        name = "%s/c%d_s%d.fna" % (sub, c, s)
        open(os.path.join(tmp, name), "wb").write(synth.fasta_text("c%d_s%d" % (c, s), synth.clade_genome(c, s, LENGTH)))
        names.append(name)
    return names


def main():
    for exe in (SK, REF):
        if not os.path.exists(exe):
            sys.exit("build the reference drivers first: make -C oracle ref ref_sketch")
    os.makedirs(OUT, exist_ok=True)
    for f in os.listdir(OUT):
        os.remove(os.path.join(OUT, f))
    with tempfile.TemporaryDirectory() as tmp:
        run(REF, "shuffle", K, S, L, os.path.join(tmp, "L2K8.shuf"))
        rnames = write_genomes(tmp, "g", [(c, s) for c in (40, 41, 42) for s in range(7)])
        qnames = write_genomes(tmp, "q", [(40, 8), (41, 9), (42, 7), (50, 0), (51, 3), (40, 9)])
        open(os.path.join(tmp, "ref.list"), "w").write("\n".join(rnames) + "\n")
        open(os.path.join(tmp, "qry.list"), "w").write("\n".join(qnames) + "\n")
        # 4 threads over 21 equal files / 3 over 6: every file <= totalSize/numThreads (the small-file path, :366-374)
        run(SK, "sketch", "L2K8.shuf", "ref.list", "ref.sketch", 4, 0, cwd=tmp)
        run(SK, "sketch", "L2K8.shuf", "qry.list", "qry.sketch", 3, 1, cwd=tmp)
        if not os.path.exists(os.path.join(tmp, "ref.sketch.dict")):   # the reference's own transSketches
            run(SK, "resave", "ref.sketch", "ref2.sketch", cwd=tmp)
            assert open(os.path.join(tmp, "ref2.sketch"), "rb").read() == open(os.path.join(tmp, "ref.sketch"), "rb").read()
            os.rename(os.path.join(tmp, "ref2.sketch.dict"), os.path.join(tmp, "ref.sketch.dict"))
            os.rename(os.path.join(tmp, "ref2.sketch.index"), os.path.join(tmp, "ref.sketch.index"))
        info, names, h, off = ok.read_sketches32(os.path.join(tmp, "ref.sketch"))
        unsorted = sum(bool(np.any(np.diff(h[int(off[g]):int(off[g + 1])].astype(np.int64)) < 0)) for g in range(len(names)))
        assert unsorted == len(names), "expected the reference to write every sketch unsorted"
        assert sorted(names) == sorted(rnames)
        cases = []
        wref, wqry = os.path.join(tmp, "ref.sketch"), os.path.join(tmp, "qry.sketch")
        for metric in (0, 1):
            for D in (0.05, 0.3, 1.0):
                name = "alldist_M%d_D%g" % (metric, D)
                run(REF, "alldist", tmp, wref, name + ".out", D, metric, 3)
                lines = open(os.path.join(tmp, name + ".out")).read().split("\n")
                body = sorted(x for x in lines[1:] if x)     # thread by thread in the file: compared as a sorted set of lines
                open(os.path.join(OUT, name + ".ref.txt"), "w").write("\n".join(body) + "\n")
                cases.append({"file": name + ".ref.txt", "cmd": "alldist", "metric": metric, "max_dist": D, "lines": len(body)})
            for D, N in ((0.1, 0), (1.0, 0), (1.0, 3)):
                name = "dist_M%d_D%g_N%d" % (metric, D, N)
                run(REF, "dist", tmp, wref, wqry, name + ".out", D, N, 1 if N else 0, metric, 1)
                body = [x for x in open(os.path.join(tmp, name + ".out")).read().split("\n")[1:] if x]
                open(os.path.join(OUT, name + ".ref.txt"), "w").write("\n".join(body) + "\n")
                cases.append({"file": name + ".ref.txt", "cmd": "dist", "metric": metric, "max_dist": D, "max_neighbor": N,
                              "lines": len(body)})
        for f in ("ref.sketch", "qry.sketch"):
            open(os.path.join(OUT, f), "wb").write(open(os.path.join(tmp, f), "rb").read())
        json.dump({"half_k": K, "half_subk": S, "drlevel": L, "hash_bits": 4 * (K - L), "genome_order": names,
                   "source": "sketches written by the reference's sketchFastaFile/saveSketches (oracle/_ref/ref_sketch_driver), texts by "
                             "its index_tridist/index_dist (oracle/_ref/ref_driver)", "cases": cases},
                  open(os.path.join(OUT, "manifest.json"), "w"), indent=1)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
