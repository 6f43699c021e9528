#!/usr/bin/env python3
"""Developer tool: where do the waves of rk_dist_kernel spend their cycles?

    python3 tools/dist_phase_profile.py build          # cross-compiles a -DRK_DIST_PROFILE copy of the library
    python3 tools/dist_phase_profile.py run [n] [steps]  # on the GPU box: phase table

The instrumented copy lives under rabbitkssd_amd/build/prof/ and is never loaded by the product."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF_DIR = os.path.join(ROOT, "rabbitkssd_amd", "build", "prof")
PROF_LIB = os.path.join(PROF_DIR, "librabbitkssd.so")
PHASES = ["loop head (cursor, issue next slices)", "barrier: scatters done", "epilogue: scan row",
          "barrier: scan done", "epilogue: evaluate cells", "barrier: epilogue done", "zero row + barrier",
          "gather issue (+wait slices)", "gather wait", "bump (LDS atomics)", "long lists", "final flush"]


def build():
    from rabbitkssd_amd import build as b
    os.makedirs(PROF_DIR, exist_ok=True)
    objs = []
    for s in sorted(f for f in os.listdir(b.CSRC) if f.endswith(".hip")):
        obj = os.path.join(PROF_DIR, s[:-4] + ".o")
        subprocess.check_call([b.HIPCC] + b.FLAGS + ["-DRK_DIST_PROFILE", "-c", os.path.join(b.CSRC, s), "-o", obj])
        objs.append(obj)
    subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PROF_LIB] + objs)
    print(PROF_LIB)


def run(n=10000, steps=20):
    import torch
    from rabbitkssd_amd import capi, synth
    capi.LIB_PATH = PROF_LIB
    ctx = capi.Context(0)
    names, hashes, off = synth.clade_sketches(n, 1220, 28)
    index = ctx.index_build(ctx.sketches_from_host(hashes, off), 28)
    hits = torch.empty((1 << 20) * capi.HIT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    counters = torch.zeros(steps, dtype=torch.int64, device="cuda")
    L = capi.lib()
    buf = (C.c_ulonglong * 16)()
    ctx.dist_rows_dev(index, 1, 0, 20, 0.05, hits.data_ptr(), 1 << 20, counters.data_ptr())
    torch.cuda.synchronize()
    L.rk_debug_dist_prof(buf, 1)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for i in range(steps):
        ctx.dist_rows_dev(index, 1, 0, 20, 0.05, hits.data_ptr(), 1 << 20, counters.data_ptr() + 8 * i)
    ev1.record()
    torch.cuda.synchronize()
    import numpy as np
    raw = np.zeros((65536, 16), dtype=np.uint64)
    L.rk_debug_dist_prof_raw(raw.ctypes.data_as(C.c_void_p), C.c_ulonglong(65536))
    L.rk_debug_dist_prof(buf, 1)
    live = raw[raw[:, 13] > 0]
    t0, t1 = live[:, 14].astype(np.int64), live[:, 15].astype(np.int64)   # last launch, 100 MHz wall clock
    base = t0.min()
    span = (t1.max() - base)
    print("last launch: %d waves, span %.1f us; wave lifetime us: mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (
        len(live), span / 100.0, (t1 - t0).mean() / 100.0, *[np.percentile(t1 - t0, q) / 100.0 for q in (10, 50, 90)],
        (t1 - t0).max() / 100.0))
    nw = len(live)
    print("  by blockIdx decile: start us | lifetime us | share of wave ticks per phase 0..11 (accumulated over all launches)")
    for d in range(10):
        part = live[d * nw // 10:(d + 1) * nw // 10]
        ph = part[:, :12].sum(axis=0) / max(1, part[:, 12].sum())
        print("  %d: %6.1f | %5.1f | %s" % (d, (part[:, 14].astype(np.int64) - base).mean() / 100.0,
                                          (part[:, 15].astype(np.int64) - part[:, 14].astype(np.int64)).mean() / 100.0,
                                          " ".join("%4.1f" % (100 * x) for x in ph)))
    edges = np.linspace(0, span, 17)
    for a, b in zip(edges[:-1], edges[1:]):
        mid = base + (a + b) / 2
        print("  t=%6.1f us resident waves %5d   started so far %5d" % ((a + b) / 200.0, int(((t0 <= mid) & (t1 > mid)).sum()),
                                                                    int((t0 <= mid).sum())))
    ms = ev0.elapsed_time(ev1) / steps
    total, waves = buf[12] / steps, buf[13] / steps
    print("instrumented kernel %.3f ms/launch, %d waves, mean wave lifetime %.0f ticks" % (ms, waves, total / waves))
    for i, name in enumerate(PHASES):
        print("%-40s %6.2f %%" % (name, 100.0 * buf[i] / steps / total))
    print("%-40s %6.2f %%" % ("(unaccounted)", 100.0 * (1 - sum(buf[:12]) / steps / total)))
    print("wave-ticks per launch %.3e  (x / ticks-per-second / kernel time = mean resident waves)" % total)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(*[int(x) for x in sys.argv[2:]])
