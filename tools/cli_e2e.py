#!/usr/bin/env python3
"""End-to-end timing of the rabbit_kssd host tool on synthetic FASTA files (GPU box).
    python3 tools/cli_e2e.py [n_genomes] [genome_length] [threads]
Writes the genomes under $TMPDIR (default /tmp), runs shuffle + sketch + alldist, prints wall times."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "rabbitkssd_amd", "rabbit_kssd")


def main(n=200, length=5_000_000, threads=16):
    tmp = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rk_e2e")
    os.makedirs(tmp, exist_ok=True)
    rng = np.random.default_rng(7)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    t0 = time.time()
    base = lut[rng.integers(0, 4, length)]
    paths = []
    for g in range(n):
        seq = base.copy()
        pos = rng.integers(0, length, length // 200)          # 0.5 % substitutions per strain
        seq[pos] = lut[rng.integers(0, 4, len(pos))]
        lines = seq.reshape(-1, 80) if length % 80 == 0 else None
        p = os.path.join(tmp, "g%04d.fa" % g)
        with open(p, "wb") as f:
            f.write(b">g%d synthetic\n" % g)
            if lines is not None:
                f.write(np.concatenate([lines, np.full((lines.shape[0], 1), 10, np.uint8)], axis=1).tobytes())
            else:
                f.write(seq.tobytes() + b"\n")
        paths.append(p)
    lst = os.path.join(tmp, "list.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    total = sum(os.path.getsize(p) for p in paths)
    print("wrote %d genomes, %.2f GB in %.1f s" % (n, total / 1e9, time.time() - t0), flush=True)

    def run(args):
        t = time.time()
        r = subprocess.run([TOOL] + args, cwd=tmp, capture_output=True, text=True)
        dt = time.time() - t
        if r.returncode:
            print(r.stdout[-2000:], r.stderr[-2000:])
            raise SystemExit("rabbit_kssd %s failed" % args[0])
        return dt, r.stderr

    dt, _ = run(["shuffle", "-k", "10", "-s", "6", "-l", "3", "-o", "L3K10.shuf"])
    print("shuffle %.2f s" % dt)
    for rep in range(2):  # second pass: files in the page cache
        dt, err = run(["sketch", "-L", "L3K10.shuf", "-i", lst, "-o", "out", "-t", str(threads)])
        print("sketch pass %d: %.2f s wall -> %.2f GB/s of FASTA, %.2f genomes/s" % (rep, dt, total / dt / 1e9, n / dt))
        print("   " + " | ".join(l for l in err.splitlines() if "time" in l or "timing" in l))
    dt, err = run(["alldist", "-i", "out.sketch", "-o", "out.dist", "-D", "0.05", "-t", str(threads)])
    print("alldist %.2f s wall, %d output lines" % (dt, sum(1 for _ in open(os.path.join(tmp, "out.dist")))))
    print("   " + " | ".join(l for l in err.splitlines() if "time" in l or "timing" in l))


def write_sketch_file(path, names, hashes, off, half_k=10, half_subk=6, drlevel=3):
    """.sketch layout: info, name lengths, hash counts, then name+hashes per genome"""
    n = len(names)
    with open(path, "wb") as f:
        np.array([(half_k << 8) + (half_subk << 4) + drlevel, half_k, half_subk, drlevel, n], dtype=np.int32).tofile(f)
        np.array([len(x) for x in names], dtype=np.int32).tofile(f)
        np.diff(off).astype(np.int32).tofile(f)
        for i, name in enumerate(names):
            f.write(name.encode())
            hashes[int(off[i]):int(off[i + 1])].astype(np.uint32).tofile(f)


def alldist(n=10000, threads=16, max_dist="0.05"):
    """configs[2]: alldist from a precomputed .sketch (and its .dict/.index after the first run)"""
    sys.path.insert(0, ROOT)
    from rabbitkssd_amd import synth
    tmp = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rk_e2e")
    os.makedirs(tmp, exist_ok=True)
    names, hashes, off = synth.clade_sketches(n, 1220, 28)
    sk = os.path.join(tmp, "syn%d.sketch" % n)
    write_sketch_file(sk, names, hashes, off)
    for f in (sk + ".dict", sk + ".index"):
        if os.path.exists(f):
            os.remove(f)
    for rep in range(3):  # run 0 builds and writes .dict/.index, runs 1-2 load them
        t = time.time()
        r = subprocess.run([TOOL, "alldist", "-i", sk, "-o", os.path.join(tmp, "syn.dist"), "-D", str(max_dist), "-t", str(threads)],
                           cwd=tmp, capture_output=True, text=True)
        dt = time.time() - t
        if r.returncode:
            print(r.stderr[-2000:])
            raise SystemExit("alldist failed")
        lines = sum(1 for _ in open(os.path.join(tmp, "syn.dist")))
        print("alldist run %d: %.2f s wall, %d pairs reported, %.3g genome-pairs/s end to end" % (rep, dt, lines, n * (n - 1) / 2 / dt))
        print("   " + " | ".join(l.strip("= ") for l in r.stderr.splitlines() if "time" in l or "timing" in l))


def big(length=1_000_000_000, n=2, threads=16):
    """S7: a few large genomes (one FASTA record of `length` bases each, 60-column lines)"""
    tmp = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rk_e2e")
    os.makedirs(tmp, exist_ok=True)
    rng = np.random.default_rng(11)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    block = 60_000_000
    paths = []
    t0 = time.time()
    for g in range(n):
        p = os.path.join(tmp, "big%d.fa" % g)
        with open(p, "wb") as f:
            f.write(b">chr1 big genome %d\n" % g)
            done = 0
            while done < length:
                m = min(block, length - done)
                m -= m % 60
                if m == 0:
                    break
                seq = lut[rng.integers(0, 4, m, dtype=np.uint8)].reshape(-1, 60)
                f.write(np.concatenate([seq, np.full((seq.shape[0], 1), 10, np.uint8)], axis=1).tobytes())
                done += m
        paths.append(p)
    lst = os.path.join(tmp, "big.list")
    open(lst, "w").write("\n".join(paths) + "\n")
    total = sum(os.path.getsize(p) for p in paths)
    print("wrote %d genomes, %.2f GB in %.1f s" % (n, total / 1e9, time.time() - t0), flush=True)
    subprocess.run([TOOL, "shuffle", "-k", "10", "-s", "6", "-l", "3", "-o", "L3K10.shuf"], cwd=tmp, capture_output=True)
    for rep in range(2):
        t = time.time()
        r = subprocess.run([TOOL, "sketch", "-L", "L3K10.shuf", "-i", lst, "-o", "bigout", "-t", str(threads)], cwd=tmp, env=dict(os.environ, RK_TIMING="1"),
                           capture_output=True, text=True)
        dt = time.time() - t
        if r.returncode:
            print(r.stderr[-2000:])
            raise SystemExit("sketch failed")
        print("big sketch pass %d: %.2f s wall -> %.2f GB/s" % (rep, dt, total / dt / 1e9))
        print("   " + " | ".join(l.strip("= ") for l in r.stderr.splitlines() if "time" in l or "timing" in l))
    for p in paths:
        os.remove(p)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        big(*[int(x) for x in sys.argv[2:]])
    elif len(sys.argv) > 1 and sys.argv[1] == "alldist":
        alldist(int(sys.argv[2]) if len(sys.argv) > 2 else 10000, int(sys.argv[3]) if len(sys.argv) > 3 else 16,
                sys.argv[4] if len(sys.argv) > 4 else "0.05")
    else:
        main(*[int(x) for x in sys.argv[1:]])
