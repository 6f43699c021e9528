#!/usr/bin/env python3
"""Developer aid: per-wave timeline of the two-stage scan kernel (RK_SCAN2_TRACE=file written by the library).
    python3 tools/trace_scan.py gpurun_out/scan_trace.bin"""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
t0, t1, ch, bl = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2], t[:, 3]
base = t0[t0 > 0].min()
s, e = (t0 - base) / 100.0, (t1 - base) / 100.0   # 100 MHz -> us
print("waves %d, kernel span %.1f us" % (len(t), e.max()))
print("start us: min %.1f p50 %.1f p90 %.1f max %.1f" % (s.min(), np.median(s), np.percentile(s, 90), s.max()))
print("end   us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (e.min(), np.percentile(e, 10), np.median(e), np.percentile(e, 90), e.max()))
print("alive us: mean %.1f (%.0f %% of the span)" % ((e - s).mean(), 100 * (e - s).mean() / e.max()))
for k in np.unique(ch):
    m = ch == k
    print("chunks=%d: %d waves, blocks mean %.1f, alive mean %.1f us, end mean %.1f us" % (k, m.sum(), bl[m].mean(), (e - s)[m].mean(), e[m].mean()))
wg = np.arange(len(t)) // 16
for lo in (0, 256):
    m = (wg >= lo) & (wg < lo + 256)
    print("workgroups %d..%d: start mean %.1f, end mean %.1f" % (lo, lo + 255, s[m].mean(), e[m].mean()))
xcd = wg % 8
print("per XCD end mean:", " ".join("%.0f" % e[xcd == x].mean() for x in range(8)))
dr, fl, fr, nd = t[:, 4].astype(np.int64) / 100.0, t[:, 5].astype(np.int64) / 100.0, (t[:, 6].astype(np.int64) - base) / 100.0, t[:, 7].astype(np.int64)
print("time in drains of the second queue: mean %.1f us per wave (%.1f drains, %.2f us each); in the run-end flushes (incl. their drains): mean %.1f us" % (dr.mean(), nd.mean(), dr.sum() / max(1, nd.sum()), fl.mean()))
print("end of the chunk loop (before the last drain): p10 %.1f p50 %.1f p90 %.1f us; last drain: mean %.1f us, p90 %.1f, max %.1f" % (tuple(np.percentile(fr, [10, 50, 90])) + ((e - fr).mean(), np.percentile(e - fr, 90), (e - fr).max())))
for x in range(8):
    m = xcd == x
    print("xcd %d: end p50 %.0f, loop end p50 %.0f, drains %.1f us, flushes %.1f us, blocks %.1f" % (x, np.median(e[m]), np.median(fr[m]), dr[m].mean(), fl[m].mean(), bl[m].mean()))
