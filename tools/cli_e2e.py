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
        print("   " + " | ".join(l for l in err.splitlines() if "time" in l))
    dt, err = run(["alldist", "-i", "out.sketch", "-o", "out.dist", "-d", "0.05", "-t", str(threads)])
    print("alldist %.2f s wall, %d output lines" % (dt, sum(1 for _ in open(os.path.join(tmp, "out.dist")))))
    print("   " + " | ".join(l for l in err.splitlines() if "time" in l))


if __name__ == "__main__":
    main(*[int(x) for x in sys.argv[1:]])
