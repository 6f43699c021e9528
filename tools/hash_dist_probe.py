#!/usr/bin/env python3
"""Developer probe: how evenly the sketcher's hashes (K10 S6 L3, 200 random 5 Mb genomes) fill the hash space -- the canonical k-mer is the
smaller of a k-mer and its reverse complement, so its leading bases (the hash's top bits) favour A over T 7 : 1.
    python3 tools/hash_dist_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rabbitkssd_amd import capi, synth
ctx = capi.Context(0)
flt = ctx.filter(capi.params_init(10, 6, 3), synth.shuf_table(10, 6, 3))
n, length = 1000, 5_000_000
stride = (length + 1023) // 1024 * 1024
g = torch.Generator(device="cuda"); g.manual_seed(7)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
packed = torch.zeros(n * stride, dtype=torch.uint8, device="cuda")
view = packed.view(n, stride)
for i in range(n):
    view[i, :length] = lut[torch.randint(0, 4, (length,), generator=g, device="cuda")]
gbeg = np.arange(n, dtype=np.uint64) * stride
gend = gbeg + np.uint64(length)
sk = ctx.sketch_packed_dev(flt, packed.data_ptr(), packed.numel(), gbeg, gend, 0)
h, off = sk.download()
print("hashes", len(h), "max", int(h.max()), "bits", int(h.max()).bit_length())
for tb in (2, 4, 6, 8, 10, 11, 12):
    c = np.bincount(h >> (28 - tb), minlength=1 << tb)
    print("top %2d bits: mean %.0f min %d max %d  max/mean %.2f" % (tb, c.mean(), c.min(), c.max(), c.max() / c.mean()))
c = np.bincount(h >> (28 - 13), minlength=1 << 13).astype(np.float64)
srt = np.sort(c)[::-1]
print("top 13 bits: mean %.0f; buckets above 2.67x the mean: %d of 8192 (%.1f %% of the hashes); above 2x: %d; fullest %.2fx" % (
    c.mean(), int((c > 2.67 * c.mean()).sum()), 100.0 * c[c > 2.67 * c.mean()].sum() / c.sum(), int((c > 2 * c.mean()).sum()), srt[0] / c.mean()))
