#!/usr/bin/env python3
"""Generates the committed fixtures in tests/golden/ (run in the build container only).

Inputs are synthetic (rabbitkssd_amd.synth, fixed seeds).  Expected outputs come from
  * the REAL reference objects (oracle/_ref/ref_driver = /root/reference/src/{dist,common,
    shuffle}.cpp compiled unmodified): params.txt, shuf.json, dist/*.ref.txt
  * the C restatement (oracle/) for sketch/expected.json (marked "pinned": false: this script does not run
    the reference's sketch code).  The same input files are part of tests/golden/sketch_ref, whose expected sets
    come from the reference's own sketchFastaFile (tests/golden/make_sketch_golden.py) and are identical.
The script refuses to write a dist fixture when the restatement and the real reference
disagree, so a committed fixture is always one both agree on.
"""
import hashlib
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

REF = os.path.join(ROOT, "oracle", "_ref", "ref_driver")

PARAM_SETS = [(10, 6, 3), (10, 7, 4), (8, 5, 2), (8, 6, 3), (12, 6, 3), (7, 6, 3), (6, 4, 1),
              (6, 6, 3), (16, 7, 4), (9, 5, 2)]
SHUF_SETS = [(10, 6, 3), (8, 5, 2), (6, 4, 1)]


def ref(*args):
    return subprocess.run([REF] + [str(a) for a in args], check=True, stdout=subprocess.PIPE,
                          stderr=subprocess.DEVNULL).stdout.decode()


def gen_params():
    lines = []
    for k, s, l in PARAM_SETS:
        lines.append(ref("param", k, s, l).strip())
        p = ok.init_param(k, s, l)
        mine = "%d %d %d %d %d %d %d %d %x %x %x %x" % (
            p.half_k, p.half_subk, p.drlevel, p.rev_add_move, p.half_outctx_len, p.dim_start,
            p.dim_end, p.kmer_size, p.domask, p.tupmask, p.undomask0, p.undomask1)
        assert mine == lines[-1], (mine, lines[-1])
    open(os.path.join(HERE, "params.txt"), "w").write("\n".join(lines) + "\n")


def gen_shuf(tmp):
    out = []
    for k, s, l in SHUF_SETS:
        path = os.path.join(tmp, "L%dK%dS%d.shuf" % (l, k, s))
        ref("shuffle", k, s, l, path)
        raw = open(path, "rb").read()
        hdr = np.frombuffer(raw[:16], dtype="<i4").tolist()
        tab = np.frombuffer(raw[16:], dtype="<i4")
        mine = ok.shuffle_table(k, s, l)
        assert np.array_equal(mine, tab), "restated shuffle differs from reference"
        out.append({"k": k, "s": s, "l": l, "header": hdr, "first8": tab[:8].tolist(),
                    "size": len(raw), "md5": hashlib.md5(raw).hexdigest(),
                    "n_below_dim_end": int((tab < (1 << (4 * (s - l)))).sum())})
    json.dump(out, open(os.path.join(HERE, "shuf.json"), "w"), indent=1)


def dist_case_sketches():
    """40 clade genomes + edge cases, 24-bit hash space (k=8,s=5,l=2)."""
    names, hashes, off = synth.clade_sketches(40, 200, 24, kmer_size=16, seed=7)
    parts = [hashes[int(off[i]):int(off[i + 1])] for i in range(40)]
    parts.append(np.zeros(0, dtype=np.uint32)); names.append("edge/empty.fna")
    parts.append(parts[3].copy()); names.append("edge/dup_of_3.fna")
    parts.append(parts[3][:50].copy()); names.append("edge/subset_of_3.fna")
    parts.append(np.array([0, 1, (1 << 24) - 1], dtype=np.uint32)); names.append("edge/extremes.fna")
    parts.append(np.array([(1 << 24) - 1], dtype=np.uint32)); names.append("edge/single.fna")
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(p) for p in parts])
    return names, np.concatenate(parts), off


def query_case_sketches(ref_parts_names):
    rnames, rh, roff = ref_parts_names
    rng = np.random.default_rng(11)
    names, parts = [], []
    for qi in range(12):
        src = rh[int(roff[qi * 3]):int(roff[qi * 3 + 1])]
        keep = src[rng.random(len(src)) < 0.8]
        extra = rng.integers(0, 1 << 24, size=40, dtype=np.uint64).astype(np.uint32)
        parts.append(np.unique(np.concatenate([keep, extra])))
        names.append("qry/q%02d.fna" % qi)
    parts.append(np.zeros(0, dtype=np.uint32)); names.append("qry/empty.fna")
    parts.append(rh[int(roff[3]):int(roff[4])].copy()); names.append("qry/same_as_ref3.fna")
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(p) for p in parts])
    return names, np.concatenate(parts), off


def sorted_body(text):
    lines = text.split("\n")
    assert lines[0].startswith(" genome0\t"), lines[0]
    return sorted(x for x in lines[1:] if x)


def gen_dist(tmp):
    d = os.path.join(HERE, "dist")
    os.makedirs(d, exist_ok=True)
    K, S, L, BITS = 8, 5, 2, 24
    rnames, rh, roff = dist_case_sketches()
    qnames, qh, qoff = query_case_sketches((rnames, rh, roff))
    rpath = os.path.join(d, "ref.sketch")
    qpath = os.path.join(d, "qry.sketch")
    ok.save_sketches32(rpath, K, S, L, rnames, rh, roff)
    ok.save_sketches32(qpath, K, S, L, qnames, qh, qoff)
    # .dict/.index from the restatement, consumed by the REAL index_tridist/index_dist
    wref = os.path.join(tmp, "ref.sketch")
    ok.save_sketches32(wref, K, S, L, rnames, rh, roff)
    postings, counts = ok.index_build32(rh, roff, BITS)
    ok.write_index32(wref + ".dict", wref + ".index", postings, counts, BITS)
    rsizes = np.diff(roff).astype(np.uint32)
    manifest = []
    for metric in (0, 1):
        for D in (0.05, 0.3, 1.0, 1.5):
            name = "alldist_M%d_D%g" % (metric, D)
            ref("alldist", tmp, wref, name + ".out", D, metric, 3)
            got = sorted_body(open(os.path.join(tmp, name + ".out")).read())
            hits, _ = ok.index_dist32(counts, BITS, postings, rsizes, rh, roff, 1, metric, 2 * K, D)
            mine = sorted(x.rstrip("\n") for x in ok.alldist_text(rnames, hits))
            assert mine == got, "alldist %s: restatement != reference" % name
            open(os.path.join(d, name + ".ref.txt"), "w").write("\n".join(got) + "\n")
            manifest.append({"file": name + ".ref.txt", "cmd": "alldist", "metric": metric,
                             "max_dist": D, "lines": len(got)})
    # independent brute force (dead code tri_dist, src/dist.cpp:345) must agree with the index path
    ref("tridist", tmp, wref, "tri.out", 0.3, 2)
    tri = sorted(x[1:] for x in open(os.path.join(tmp, "tri.out")).read().split("\n")[1:] if x)
    idx = sorted(open(os.path.join(d, "alldist_M0_D0.3.ref.txt")).read().split("\n")[:-1])
    assert tri == idx, "reference tri_dist != reference index_tridist on oracle-written index"
    for metric in (0, 1):
        for D in (0.1, 1.0):
            for N in (0, 1, 3, 100):
                name = "dist_M%d_D%g_N%d" % (metric, D, N)
                ref("dist", tmp, wref, qpath, name + ".out", D, N, 1 if N else 0, metric, 1)
                text = open(os.path.join(tmp, name + ".out")).read()
                body = [x for x in text.split("\n")[1:] if x]
                hits, _ = ok.index_dist32(counts, BITS, postings, rsizes, qh, qoff, 0, metric,
                                          2 * K, D)
                if N:
                    sel = []
                    for q in range(len(qnames)):
                        sel.append(ok.topn_row(hits[hits["row"] == q], N))
                    hits = np.concatenate(sel)
                mine = [x.rstrip("\n") for x in ok.dist_text(qnames, rnames, hits)]
                assert mine == body, "dist %s: restatement != reference" % name
                open(os.path.join(d, name + ".ref.txt"), "w").write("\n".join(body) + "\n")
                manifest.append({"file": name + ".ref.txt", "cmd": "dist", "metric": metric,
                                 "max_dist": D, "max_neighbor": N, "lines": len(body)})
    json.dump({"half_k": K, "half_subk": S, "drlevel": L, "hash_bits": BITS,
               "source": "real reference index_tridist/index_dist via oracle/_ref/ref_driver",
               "pinned": True, "cases": manifest}, open(os.path.join(d, "manifest.json"), "w"),
              indent=1)


def gen_dist64(tmp):
    """use64 layout (half_k - drlevel > 8): K12 S6 L3 -> 36-bit hashes; real index_tridist/index_dist."""
    d = os.path.join(HERE, "dist64")
    os.makedirs(d, exist_ok=True)
    K, S, L, BITS = 12, 6, 3, 36
    rng = np.random.default_rng(64)
    rparts, rnames = [], []
    for c in range(3):
        anc = np.unique(rng.integers(0, 1 << BITS, size=260, dtype=np.uint64))
        for st in range(8):
            keep = anc[rng.random(len(anc)) < 0.97 - 0.02 * st]
            extra = rng.integers(0, 1 << BITS, size=12, dtype=np.uint64)
            rparts.append(np.unique(np.concatenate([keep, extra])))
            rnames.append("syn64/c%d_s%d.fna" % (c, st))
    rparts.append(np.zeros(0, dtype=np.uint64)); rnames.append("edge/empty.fna")
    rparts.append(rparts[2].copy()); rnames.append("edge/dup_of_2.fna")
    rparts.append(np.array([0, 1, (1 << BITS) - 1], dtype=np.uint64)); rnames.append("edge/extremes.fna")
    roff = np.concatenate([[0], np.cumsum([len(p) for p in rparts])]).astype(np.uint64)
    rh = np.concatenate(rparts)
    qparts, qnames = [], []
    for qi in range(6):
        src = rparts[qi * 4]
        keep = src[rng.random(len(src)) < 0.85]
        qparts.append(np.unique(np.concatenate([keep, rng.integers(0, 1 << BITS, size=25, dtype=np.uint64)])))
        qnames.append("qry64/q%d.fna" % qi)
    qparts.append(np.zeros(0, dtype=np.uint64)); qnames.append("qry64/empty.fna")
    qoff = np.concatenate([[0], np.cumsum([len(p) for p in qparts])]).astype(np.uint64)
    qh = np.concatenate(qparts)
    ok.save_sketches64(os.path.join(d, "ref64.sketch"), K, S, L, rnames, rh, roff)
    ok.save_sketches64(os.path.join(d, "qry64.sketch"), K, S, L, qnames, qh, qoff)
    wref = os.path.join(tmp, "ref64.sketch")
    ok.save_sketches64(wref, K, S, L, rnames, rh, roff)
    uhash, ucount, postings = ok.index_build64(rh, roff)
    # posting blocks in a scrambled (hash-map-like) order: any order is legal for the reader
    perm = rng.permutation(len(uhash))
    starts = np.concatenate([[0], np.cumsum(ucount.astype(np.int64))]).astype(np.int64)
    scr_post = np.concatenate([postings[int(starts[i]):int(starts[i + 1])] for i in perm]) if len(perm) else postings
    ok.write_index64(wref + ".dict", wref + ".index", scr_post, uhash[perm], ucount[perm])
    rsizes = np.diff(roff).astype(np.uint32)
    manifest = []
    for metric, D in ((0, 0.05), (0, 1.0), (1, 0.2), (0, 1.5)):
        name = "alldist64_M%d_D%g" % (metric, D)
        ref("alldist", tmp, wref, name + ".out", D, metric, 2)
        got = sorted_body(open(os.path.join(tmp, name + ".out")).read())
        hits, _ = ok.index_dist64(uhash, ucount, postings, rsizes, rh, roff, 1, metric, 2 * K, D)
        mine = sorted(x.rstrip("\n") for x in ok.alldist_text(rnames, hits))
        assert mine == got, "alldist64 %s: restatement != reference" % name
        open(os.path.join(d, name + ".ref.txt"), "w").write("\n".join(got) + "\n")
        manifest.append({"file": name + ".ref.txt", "cmd": "alldist", "metric": metric, "max_dist": D, "lines": len(got)})
    for metric, D, N in ((0, 0.3, 0), (1, 1.0, 2), (0, 1.0, 0)):
        name = "dist64_M%d_D%g_N%d" % (metric, D, N)
        ref("dist", tmp, wref, os.path.join(d, "qry64.sketch"), name + ".out", D, N, 1 if N else 0, metric, 1)
        body = [x for x in open(os.path.join(tmp, name + ".out")).read().split("\n")[1:] if x]
        hits, _ = ok.index_dist64(uhash, ucount, postings, rsizes, qh, qoff, 0, metric, 2 * K, D)
        if N:
            hits = np.concatenate([ok.topn_row(hits[hits["row"] == q], N) for q in range(len(qnames))])
        mine = [x.rstrip("\n") for x in ok.dist_text(qnames, rnames, hits)]
        assert mine == body, "dist64 %s: restatement != reference" % name
        open(os.path.join(d, name + ".ref.txt"), "w").write("\n".join(body) + "\n")
        manifest.append({"file": name + ".ref.txt", "cmd": "dist", "metric": metric, "max_dist": D,
                         "max_neighbor": N, "lines": len(body)})
    json.dump({"half_k": K, "half_subk": S, "drlevel": L, "hash_bits": BITS, "pinned": True,
               "source": "real reference index_tridist/index_dist (use64 branch) via oracle/_ref/ref_driver",
               "cases": manifest}, open(os.path.join(d, "manifest.json"), "w"), indent=1)


def numpy_sketch(param, table, seq):
    """Independent (vectorised) restatement of src/sketch.cpp:491-530 for ONE record."""
    lut = np.full(256, -1, dtype=np.int64)
    for i, ch in enumerate("ACGT"):
        lut[ord(ch)] = i
        lut[ord(ch.lower())] = i
    code = lut[seq]
    n, k = len(seq), int(param.kmer_size)
    if n < k:
        return np.zeros(0, dtype=np.uint64)
    bad = (code < 0).astype(np.int64)
    cs = np.concatenate([[0], np.cumsum(bad)])
    valid = (cs[k:] - cs[:-k]) == 0  # window [i, i+k) has no invalid base
    c = np.where(code < 0, 0, code).astype(np.uint64)
    fwd = np.zeros(n - k + 1, dtype=np.uint64)
    rev = np.zeros(n - k + 1, dtype=np.uint64)
    for t in range(k):
        fwd = (fwd << np.uint64(2)) | c[t:n - k + 1 + t]
        rev = rev | ((c[t:n - k + 1 + t] ^ np.uint64(3)) << np.uint64(2 * t))
    uni = np.minimum(fwd, rev)[valid]
    out = int(param.half_outctx_len)
    dim = ((uni & np.uint64(param.domask)) >> np.uint64(2 * out)).astype(np.int64)
    v = table[dim]
    keep = (v < param.dim_end) & (v >= param.dim_start)
    uni, v = uni[keep], v[keep].astype(np.uint64)
    sh = np.uint64(2 * k - 4 * out)
    dr = (((uni & np.uint64(param.undomask0)) | ((uni & np.uint64(param.undomask1)) << sh))
          >> np.uint64(4 * param.drlevel)) | v
    return np.unique(dr)


def gen_sketch():
    d = os.path.join(HERE, "sketch")
    os.makedirs(d, exist_ok=True)
    K, S, L = 8, 5, 2
    param = ok.init_param(K, S, L)
    table = ok.shuffle_table(K, S, L)
    files = {}
    g = synth.clade_genome_set(4, 30000)
    for name, bases in g:
        files[name + ".fa"] = synth.fasta_text(name, bases, 70)
    b = synth.clade_genome(7, 0, 24000).copy()
    b[5000:5040] = ord("N")          # N run resets the window (src/sketch.cpp:502-504)
    b[9000] = ord("R")               # IUPAC code is invalid
    b[12000:13000] = np.frombuffer(bytes(b[12000:13000]).lower(), dtype=np.uint8)
    rec1, rec2, rec3 = b[:8000], b[8000:8010], b[8010:]
    txt = (synth.fasta_text("multi r1 with comment", rec1, 60) + b"\n\n" +
           synth.fasta_text("short_record_below_k", rec2, 60) + b">empty_record\n" +
           synth.fasta_text("r3", rec3, 61))
    files["multi_record.fa"] = txt
    files["crlf.fa"] = synth.fasta_text("crlf", synth.clade_genome(8, 1, 9000), 50).replace(b"\n", b"\r\n")
    files["no_trailing_newline.fa"] = synth.fasta_text("x", synth.clade_genome(9, 2, 5003), 80).rstrip(b"\n")
    files["fastq_like.fq"] = b"@r1\n" + synth.clade_genome(3, 0, 300).tobytes() + b"\n+\n" + b"I" * 300 + b"\n"
    expected = {}
    for fn, data in files.items():
        open(os.path.join(d, fn), "wb").write(data)
        seq, off = ok.parse_fasta_bytes(data)
        h = ok.sketch_records(param, table, seq, off)
        indep = [numpy_sketch(param, table, seq[int(off[r]):int(off[r + 1])])
                 for r in range(len(off) - 1)]
        indep = np.unique(np.concatenate(indep)) if indep else np.zeros(0, dtype=np.uint64)
        assert np.array_equal(h, indep), "C restatement != numpy restatement for " + fn
        expected[fn] = {"n_records": len(off) - 1, "n_bases": int(off[-1]),
                        "n_windows": ok.count_windows(param, seq, off),
                        "hashes": [int(x) for x in h]}
    json.dump({"half_k": K, "half_subk": S, "drlevel": L, "pinned": False,
               "source": "oracle/kssd_oracle.c cross-checked by an independent numpy restatement; "
                         "the reference's sketch.cpp is unbuildable here (RabbitFX absent)",
               "files": expected}, open(os.path.join(d, "expected.json"), "w"))


def main():
    if not os.path.exists(REF):
        sys.exit("build oracle/_ref first: make -C oracle ref")
    with tempfile.TemporaryDirectory() as tmp:
        gen_params()
        gen_shuf(tmp)
        gen_dist(tmp)
        gen_dist64(tmp)
    gen_sketch()
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
