#!/usr/bin/env python3
"""`rabbit_kssd alldist` on the bench's 10,000 synthetic sketches with RK_TIMING=1: wall time and the tool's own stamps (GPU box).
    python3 tools/cli_alldist_timing.py [runs]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rabbitkssd_amd import synth

def main(runs=4):
    names, hashes, off = synth.clade_sketches(10000, 1220, 28, kmer_size=20)
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as tmp:
        sk = os.path.join(tmp, "bench.sketch")
        synth.write_sketch_file(sk, 10, 6, 3, names, hashes, off)
        tool = os.path.join(ROOT, "rabbitkssd_amd", "rabbit_kssd")
        for r in range(runs):
            t = time.time()
            p = subprocess.run([tool, "alldist", "-i", sk, "-o", os.path.join(tmp, "out.dist"), "-D", "0.05", "-t", "16"],
                               capture_output=True, text=True, env=dict(os.environ, RK_TIMING="1"))
            dt = time.time() - t
            print("run %d: %.1f ms wall, rc %d" % (r, dt * 1e3, p.returncode))
            for l in p.stderr.splitlines():
                if "rk" in l or "time" in l:
                    print("    " + l)

if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
