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
n, length = 200, 5_000_000
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
for tb in (2, 4, 8, 13):
    c = np.bincount(h >> (28 - tb), minlength=1 << tb)
    print("top %2d bits: mean %.0f min %d max %d  max/mean %.2f" % (tb, c.mean(), c.min(), c.max(), c.max() / c.mean()))
