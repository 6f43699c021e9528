#!/usr/bin/env python3
"""`rabbit_kssd alldist` on the bench's 10,000 sketches under a few runtime environment settings: wall + the tool's stamps."""
import os, subprocess, sys, tempfile, time, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rabbitkssd_amd import synth

def main(runs=9):
    names, hashes, off = synth.clade_sketches(10000, 1220, 28, kmer_size=20)
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as tmp:
        sk = os.path.join(tmp, "bench.sketch")
        synth.write_sketch_file(sk, 10, 6, 3, names, hashes, off)
        tool = os.path.join(ROOT, "rabbitkssd_amd", "rabbit_kssd")
        # first run writes .dict/.index
        subprocess.run([tool, "alldist", "-i", sk, "-o", "o.dist", "-D", "0.05", "-t", "16"], cwd=tmp, capture_output=True)
        settings = [{}, {"HSA_ENABLE_INTERRUPT": "0"}, {"HSA_ENABLE_SDMA": "1"}, {}, {"HSA_ENABLE_INTERRUPT": "0"}]

        for env in settings:
            walls, stamps = [], []
            for r in range(runs):
                t = time.time()
                p = subprocess.run([tool, "alldist", "-i", sk, "-o", "o.dist", "-D", "0.05", "-t", "16"], cwd=tmp, capture_output=True, text=True,
                                   env=dict(os.environ, RK_TIMING="1", **env))
                walls.append((time.time() - t) * 1e3)
                marks = [(float(m.group(1)), m.group(2).strip()) for m in re.finditer(r"\[timing\]\s+([0-9.]+) ms\s+(.*)", p.stderr)]
                stamps.append(marks)
            i = sorted(range(runs), key=lambda k: walls[k])[runs // 2]
            prev, parts = 0.0, []
            for t, w in stamps[i]:
                parts.append("%s %.1f" % (w, t - prev)); prev = t
            print("%-70s wall median %.1f ms (%s) | %s" % (env, walls[i], " ".join("%.0f" % w for w in walls), "; ".join(parts)), flush=True)

if __name__ == "__main__":
    main()
