"""CPU-only checks of the C ABI: the library loads, exports every symbol include/rabbitkssd.h
declares, and its host-side helpers (no compute) match the oracle / golden fixtures."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from oracle import oracle as ok
from rabbitkssd_amd import capi


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rabbitkssd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rk_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = capi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "librabbitkssd.so does not export " + name
    assert declared == set(capi.EXPORTS)
    assert b"gfx950" in L.rk_version()


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert capi.lib().rk_device_count() == 0
    with pytest.raises(capi.RkError) as e:
        capi.Context(0)
    assert e.value.code == -2  # RK_ERR_NO_DEVICE: there is no CPU fallback


def test_params_match_reference_golden():
    for line in open(os.path.join(GOLDEN, "params.txt")):
        f = line.split()
        k, s, l = int(f[0]), int(f[1]), int(f[2])
        if k > 16:
            continue
        p = capi.params_init(k, s, l)
        mine = "%d %d %d %d %d %d %d %d %x %x %x %x" % (
            p.half_k, p.half_subk, p.drlevel, p.rev_add_move, p.half_outctx_len, p.dim_start,
            p.dim_end, p.kmer_size, p.domask, p.tupmask, p.undomask0, p.undomask1)
        assert mine == line.strip()
    with pytest.raises(capi.RkError):
        capi.params_init(10, 6, 4)  # src/common.cpp:37
    with pytest.raises(capi.RkError):
        capi.params_init(5, 6, 3)   # src/shuffle.cpp:26
    with pytest.raises(capi.RkError):
        capi.params_init(10, 8, 3)  # src/shuffle.cpp:30
    assert capi.hash_bits(capi.params_init(10, 6, 3)) == 28


def test_pack_layout():
    seq = np.frombuffer(b"ACGTACGTAAAACCCCGG", dtype=np.uint8)
    rec_off = np.array([0, 8, 8, 16, 18], dtype=np.uint64)      # 4 records, one empty
    genome_rec = np.array([0, 2, 2, 4], dtype=np.uint64)        # genome 1 has no records
    packed, gbeg, gend = capi.pack_genomes(seq, rec_off, genome_rec)
    assert len(packed) % 1024 == 0 and all(int(b) % 1024 == 0 for b in gbeg)
    assert bytes(packed[int(gbeg[0]):int(gend[0])]) == b"ACGTACGT\x00"
    assert gbeg[1] == gend[1]
    assert bytes(packed[int(gbeg[2]):int(gend[2])]) == b"AAAACCCC\x00GG"
    assert not packed[int(gend[2]):].any()


def test_topn_matches_oracle_heap_order():
    rng = np.random.default_rng(5)
    hits = np.zeros(400, dtype=capi.HIT_DTYPE)
    hits["row"] = np.repeat(np.arange(8), 50)
    hits["col"] = np.tile(np.arange(50), 8)
    hits["dist"] = rng.integers(0, 6, size=400) / 5.0   # many ties
    hits["common"] = rng.integers(0, 100, size=400)
    for n in (1, 2, 3, 7, 50, 60):
        mine = capi.topn_rows(hits, n)
        want = np.concatenate([ok.topn_row(hits[hits["row"] == r].astype(ok.HIT_DTYPE), n)
                               for r in range(8)])
        assert mine.tobytes() == want.astype(capi.HIT_DTYPE).tobytes(), n


def test_format_hit_matches_oracle():
    h = np.zeros(1, dtype=capi.HIT_DTYPE)[0]
    h["common"], h["size0"], h["size1"], h["jorc"], h["dist"] = 7, 1200, 1234, 0.123456789, 0.05
    assert capi.format_hit("a/b.fna", "c.fna", h) == ok.format_hit("a/b.fna", "c.fna", 7, 1200, 1234,
                                                                  0.123456789, 0.05)
