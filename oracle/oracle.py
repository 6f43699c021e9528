"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module (the checker); the product path (rabbitkssd_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Param(C.Structure):
    _fields_ = [("half_k", C.c_int32), ("half_subk", C.c_int32), ("drlevel", C.c_int32),
                ("rev_add_move", C.c_int32), ("half_outctx_len", C.c_int32),
                ("dim_start", C.c_int32), ("dim_end", C.c_int32), ("kmer_size", C.c_uint32),
                ("domask", C.c_uint64), ("tupmask", C.c_uint64), ("undomask0", C.c_uint64),
                ("undomask1", C.c_uint64)]


class SketchInfo(C.Structure):
    _fields_ = [("id", C.c_int32), ("half_k", C.c_int32), ("half_subk", C.c_int32),
                ("drlevel", C.c_int32), ("genomeNumber", C.c_int32)]


HIT_DTYPE = np.dtype([("row", "<u4"), ("col", "<u4"), ("common", "<i4"), ("size0", "<i4"),
                      ("size1", "<i4"), ("pad", "<i4"), ("jorc", "<f8"), ("dist", "<f8")])


def build(force=False):
    so = os.path.join(_DIR, "liboracle.so")
    srcs = [os.path.join(_DIR, f) for f in ("kssd_oracle.c", "kssd_oracle64.c", "kssd_oracle.h")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.ok_sketch_records.restype = C.c_int64
        L.ok_sketch_records_fastq.restype = C.c_int64
        L.ok_count_windows.restype = C.c_uint64
        L.ok_index_dist32.restype = C.c_int64
        L.ok_index_dist64.restype = C.c_int64
        L.ok_topn_row.restype = C.c_uint32
        L.ok_free.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def init_param(half_k, half_subk, drlevel):
    p = Param()
    rc = lib().ok_init_param(half_k, half_subk, drlevel, C.byref(p))
    if rc:
        raise ValueError("half_subk - drlevel must be >= 3")
    return p


def shuffle_table(half_k, half_subk, drlevel):
    t = np.empty(1 << (4 * half_subk), dtype=np.int32)
    rc = lib().ok_shuffle_table(half_k, half_subk, drlevel, _p(t, C.c_int32))
    if rc < 0:
        raise ValueError("bad shuffle arguments")
    return t


def write_shuf(path, half_k, half_subk, drlevel):
    rc = lib().ok_write_shuf(path.encode(), half_k, half_subk, drlevel)
    if rc:
        raise OSError("ok_write_shuf failed: %d" % rc)


def read_shuf(path):
    hdr = (C.c_int32 * 4)()
    tab = C.POINTER(C.c_int32)()
    rc = lib().ok_read_shuf(path.encode(), hdr, C.byref(tab))
    if rc:
        raise OSError("ok_read_shuf failed: %d" % rc)
    n = 1 << (4 * hdr[2])
    out = np.ctypeslib.as_array(tab, shape=(n,)).copy()
    lib().ok_free(tab)
    return list(hdr), out


def parse_fasta_bytes(data):
    buf = np.frombuffer(data, dtype=np.uint8)
    seq = C.POINTER(C.c_uint8)()
    off = C.POINTER(C.c_uint64)()
    n = C.c_uint64()
    rc = lib().ok_parse_fasta_mem(_p(buf, C.c_uint8), C.c_uint64(len(buf)), C.byref(seq),
                                  C.byref(off), C.byref(n))
    if rc:
        raise OSError("ok_parse_fasta_mem failed: %d" % rc)
    o = np.ctypeslib.as_array(off, shape=(n.value + 1,)).copy()
    s = np.ctypeslib.as_array(seq, shape=(max(int(o[-1]), 1),))[: int(o[-1])].copy()
    lib().ok_free(seq)
    lib().ok_free(off)
    return s, o


def parse_fastq_bytes(data):
    """(seq, qual, rec_off) of a FASTQ/FASTA text; qual has one character per base."""
    buf = np.frombuffer(data, dtype=np.uint8)
    seq = C.POINTER(C.c_uint8)()
    qual = C.POINTER(C.c_uint8)()
    off = C.POINTER(C.c_uint64)()
    n = C.c_uint64()
    rc = lib().ok_parse_fastq_mem(_p(buf, C.c_uint8), C.c_uint64(len(buf)), C.byref(seq), C.byref(qual),
                                  C.byref(off), C.byref(n))
    if rc:
        raise OSError("ok_parse_fastq_mem failed: %d" % rc)
    o = np.ctypeslib.as_array(off, shape=(n.value + 1,)).copy()
    m = int(o[-1])
    s = np.ctypeslib.as_array(seq, shape=(max(m, 1),))[:m].copy()
    q = np.ctypeslib.as_array(qual, shape=(max(m, 1),))[:m].copy()
    for ptr in (seq, qual, off):
        lib().ok_free(ptr)
    return s, q, o


def sketch_records_fastq(param, shuffled_dim, seq, qual, rec_off, least_qual=0, least_num=1):
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    qual = np.ascontiguousarray(qual, dtype=np.uint8)
    rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
    tab = np.ascontiguousarray(shuffled_dim, dtype=np.int32)
    out = C.POINTER(C.c_uint64)()
    n = lib().ok_sketch_records_fastq(C.byref(param), _p(tab, C.c_int32), _p(seq, C.c_uint8),
                                      _p(qual, C.c_uint8), int(least_qual), int(least_num),
                                      _p(rec_off, C.c_uint64), C.c_uint64(len(rec_off) - 1), C.byref(out))
    if n < 0:
        raise MemoryError("ok_sketch_records_fastq failed")
    h = np.ctypeslib.as_array(out, shape=(max(n, 1),))[:n].copy()
    lib().ok_free(out)
    return h


def read_fasta(path):
    seq = C.POINTER(C.c_uint8)()
    off = C.POINTER(C.c_uint64)()
    n = C.c_uint64()
    rc = lib().ok_read_fasta(path.encode(), C.byref(seq), C.byref(off), C.byref(n))
    if rc:
        raise OSError("ok_read_fasta(%s) failed: %d" % (path, rc))
    o = np.ctypeslib.as_array(off, shape=(n.value + 1,)).copy()
    s = np.ctypeslib.as_array(seq, shape=(max(int(o[-1]), 1),))[: int(o[-1])].copy()
    lib().ok_free(seq)
    lib().ok_free(off)
    return s, o


def sketch_records(param, shuffled_dim, seq, rec_off):
    """sorted unique dr_tuples (np.uint64) of one genome given as records."""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
    tab = np.ascontiguousarray(shuffled_dim, dtype=np.int32)
    out = C.POINTER(C.c_uint64)()
    n = lib().ok_sketch_records(C.byref(param), _p(tab, C.c_int32), _p(seq, C.c_uint8),
                                _p(rec_off, C.c_uint64), C.c_uint64(len(rec_off) - 1),
                                C.byref(out))
    if n < 0:
        raise MemoryError("ok_sketch_records failed")
    h = np.ctypeslib.as_array(out, shape=(max(n, 1),))[:n].copy()
    lib().ok_free(out)
    return h


def sketch_genomes_mt(param, shuffled_dim, seq, goff, threads):
    """sketch sizes of the genomes seq[goff[g]:goff[g+1]] (one record each), `threads` OpenMP threads over genomes
    like the reference's small-file loop (src/sketch.cpp:455-457): the timed CPU baseline of bench.py's sketch leg"""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    goff = np.ascontiguousarray(goff, dtype=np.uint64)
    tab = np.ascontiguousarray(shuffled_dim, dtype=np.int32)
    sizes = np.zeros(len(goff) - 1, dtype=np.uint64)
    rc = lib().ok_sketch_genomes_mt(C.byref(param), _p(tab, C.c_int32), _p(seq, C.c_uint8), _p(goff, C.c_uint64),
                                    C.c_uint64(len(goff) - 1), int(threads), _p(sizes, C.c_uint64))
    if rc:
        raise MemoryError("ok_sketch_genomes_mt failed")
    return sizes


def count_windows(param, seq, rec_off):
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
    return int(lib().ok_count_windows(C.byref(param), _p(seq, C.c_uint8), _p(rec_off, C.c_uint64),
                                      C.c_uint64(len(rec_off) - 1)))


def save_sketches32(path, half_k, half_subk, drlevel, names, hashes, off):
    info = SketchInfo(0, half_k, half_subk, drlevel, len(names))
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    hashes = np.ascontiguousarray(hashes, dtype=np.uint32)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    rc = lib().ok_save_sketches32(path.encode(), C.byref(info), arr, _p(hashes, C.c_uint32),
                                  _p(off, C.c_uint64))
    if rc:
        raise OSError("ok_save_sketches32 failed: %d" % rc)


def read_sketches32(path):
    info = SketchInfo()
    blob = C.c_void_p()
    h = C.POINTER(C.c_uint32)()
    off = C.POINTER(C.c_uint64)()
    rc = lib().ok_read_sketches32(path.encode(), C.byref(info), C.byref(blob), C.byref(h),
                                  C.byref(off))
    if rc:
        raise OSError("ok_read_sketches32 failed: %d" % rc)
    n = info.genomeNumber
    o = np.ctypeslib.as_array(off, shape=(n + 1,)).copy()
    hashes = np.ctypeslib.as_array(h, shape=(max(int(o[-1]), 1),))[: int(o[-1])].copy()
    names = []
    addr = blob.value
    for _ in range(n):
        s = C.string_at(addr)
        names.append(s.decode())
        addr += len(s) + 1
    for ptr in (blob, h, off):
        lib().ok_free(ptr)
    return info, names, hashes, o


def index_build32(hashes, off, hash_bits):
    hashes = np.ascontiguousarray(hashes, dtype=np.uint32)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    n = len(off) - 1
    post = C.POINTER(C.c_uint32)()
    cnt = C.POINTER(C.c_uint32)()
    tot = C.c_uint64()
    rc = lib().ok_index_build32(_p(hashes, C.c_uint32), _p(off, C.c_uint64), C.c_uint32(n),
                                hash_bits, C.byref(post), C.byref(cnt), C.byref(tot))
    if rc:
        raise MemoryError("ok_index_build32 failed: %d" % rc)
    postings = np.ctypeslib.as_array(post, shape=(max(tot.value, 1),))[: tot.value].copy()
    counts = np.ctypeslib.as_array(cnt, shape=(1 << hash_bits,)).copy()
    lib().ok_free(post)
    lib().ok_free(cnt)
    return postings, counts


def write_index32(dict_path, index_path, postings, counts, hash_bits):
    postings = np.ascontiguousarray(postings, dtype=np.uint32)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    rc = lib().ok_write_index32(dict_path.encode(), index_path.encode(), _p(postings, C.c_uint32),
                                _p(counts, C.c_uint32), hash_bits, C.c_uint64(len(postings)))
    if rc:
        raise OSError("ok_write_index32 failed: %d" % rc)


def read_index32(dict_path, index_path):
    post = C.POINTER(C.c_uint32)()
    cnt = C.POINTER(C.c_uint32)()
    hs = C.c_uint64()
    tot = C.c_uint64()
    rc = lib().ok_read_index32(dict_path.encode(), index_path.encode(), C.byref(post),
                               C.byref(cnt), C.byref(hs), C.byref(tot))
    if rc:
        raise OSError("ok_read_index32 failed: %d" % rc)
    postings = np.ctypeslib.as_array(post, shape=(max(tot.value, 1),))[: tot.value].copy()
    counts = np.ctypeslib.as_array(cnt, shape=(hs.value,)).copy()
    lib().ok_free(post)
    lib().ok_free(cnt)
    return postings, counts


def distance(common, size0, size1, metric, kmer_size):
    j = C.c_double()
    d = C.c_double()
    lib().ok_distance(int(common), int(size0), int(size1), int(metric), int(kmer_size),
                      C.byref(j), C.byref(d))
    return j.value, d.value


def index_dist32(counts, hash_bits, postings, ref_sizes, q_hashes, q_off, triangle, metric,
                 kmer_size, max_dist, threads=1, want_dense=False):
    """returns (hits structured array, dense int32 [Q,R] or None)."""
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    postings = np.ascontiguousarray(postings, dtype=np.uint32)
    ref_sizes = np.ascontiguousarray(ref_sizes, dtype=np.uint32)
    q_hashes = np.ascontiguousarray(q_hashes, dtype=np.uint32)
    q_off = np.ascontiguousarray(q_off, dtype=np.uint64)
    nq, nr = len(q_off) - 1, len(ref_sizes)
    dense = np.zeros((nq, nr), dtype=np.int32) if want_dense else None
    hits = C.c_void_p()
    n = lib().ok_index_dist32(_p(counts, C.c_uint32), hash_bits, _p(postings, C.c_uint32),
                              _p(ref_sizes, C.c_uint32), C.c_uint32(nr), _p(q_hashes, C.c_uint32),
                              _p(q_off, C.c_uint64), C.c_uint32(nq), int(triangle), int(metric),
                              int(kmer_size), C.c_double(max_dist), int(threads),
                              _p(dense, C.c_int32) if want_dense else None, C.byref(hits))
    if n < 0:
        raise MemoryError("ok_index_dist32 failed: %d" % n)
    buf = C.string_at(hits.value, n * HIT_DTYPE.itemsize) if n else b""
    lib().ok_free(hits)
    return np.frombuffer(buf, dtype=HIT_DTYPE).copy(), dense


def topn_row(row_hits, max_neighbor):
    row_hits = np.ascontiguousarray(row_hits, dtype=HIT_DTYPE)
    out = np.zeros(len(row_hits) + 1, dtype=HIT_DTYPE)
    k = lib().ok_topn_row(row_hits.ctypes.data_as(C.c_void_p), C.c_uint32(len(row_hits)),
                          C.c_uint64(max_neighbor), out.ctypes.data_as(C.c_void_p))
    return out[:k]


def format_hit(name_a, name_b, common, size0, size1, jorc, dist):
    buf = C.create_string_buffer(len(name_a) + len(name_b) + 128)
    lib().ok_format_hit(buf, C.c_size_t(len(buf)), name_a.encode(), name_b.encode(), int(common),
                        int(size0), int(size1), C.c_double(jorc), C.c_double(dist))
    return buf.value.decode()


def alldist_text(names, hits):
    """lines as src/dist.cpp:233 writes them: name[j] \\t name[i] ..."""
    return [format_hit(names[h["col"]], names[h["row"]], h["common"], h["size0"], h["size1"],
                       h["jorc"], h["dist"]) for h in hits]


def dist_text(qnames, rnames, hits):
    """lines as src/dist.cpp:642 writes them: query \\t ref ..."""
    return [format_hit(qnames[h["row"]], rnames[h["col"]], h["common"], h["size0"], h["size1"],
                       h["jorc"], h["dist"]) for h in hits]


# ---- 64-bit hash layout (use64) ---------------------------------------------------------------
def save_sketches64(path, half_k, half_subk, drlevel, names, hashes, off):
    info = SketchInfo(0, half_k, half_subk, drlevel, len(names))
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    rc = lib().ok_save_sketches64(path.encode(), C.byref(info), arr, _p(hashes, C.c_uint64), _p(off, C.c_uint64))
    if rc:
        raise OSError("ok_save_sketches64 failed: %d" % rc)


def read_sketches64(path):
    info = SketchInfo()
    blob = C.c_void_p()
    h = C.POINTER(C.c_uint64)()
    off = C.POINTER(C.c_uint64)()
    rc = lib().ok_read_sketches64(path.encode(), C.byref(info), C.byref(blob), C.byref(h), C.byref(off))
    if rc:
        raise OSError("ok_read_sketches64 failed: %d" % rc)
    n = info.genomeNumber
    o = np.ctypeslib.as_array(off, shape=(n + 1,)).copy()
    hashes = np.ctypeslib.as_array(h, shape=(max(int(o[-1]), 1),))[: int(o[-1])].copy()
    names = []
    addr = blob.value
    for _ in range(n):
        s = C.string_at(addr)
        names.append(s.decode())
        addr += len(s) + 1
    for ptr in (blob, h, off):
        lib().ok_free(ptr)
    return info, names, hashes, o


def index_build64(hashes, off):
    """(uhash ascending u64[U], ucount u32[U], postings u32[H])"""
    hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    uh = C.POINTER(C.c_uint64)()
    uc = C.POINTER(C.c_uint32)()
    post = C.POINTER(C.c_uint32)()
    nh = C.c_uint64()
    tot = C.c_uint64()
    rc = lib().ok_index_build64(_p(hashes, C.c_uint64), _p(off, C.c_uint64), C.c_uint32(len(off) - 1),
                                C.byref(uh), C.byref(uc), C.byref(post), C.byref(nh), C.byref(tot))
    if rc:
        raise MemoryError("ok_index_build64 failed: %d" % rc)
    uhash = np.ctypeslib.as_array(uh, shape=(max(nh.value, 1),))[: nh.value].copy()
    ucount = np.ctypeslib.as_array(uc, shape=(max(nh.value, 1),))[: nh.value].copy()
    postings = np.ctypeslib.as_array(post, shape=(max(tot.value, 1),))[: tot.value].copy()
    for ptr in (uh, uc, post):
        lib().ok_free(ptr)
    return uhash, ucount, postings


def write_index64(dict_path, index_path, postings, uhash, ucount):
    postings = np.ascontiguousarray(postings, dtype=np.uint32)
    uhash = np.ascontiguousarray(uhash, dtype=np.uint64)
    ucount = np.ascontiguousarray(ucount, dtype=np.uint32)
    rc = lib().ok_write_index64(dict_path.encode(), index_path.encode(), _p(postings, C.c_uint32),
                                _p(uhash, C.c_uint64), _p(ucount, C.c_uint32), C.c_uint64(len(uhash)),
                                C.c_uint64(len(postings)))
    if rc:
        raise OSError("ok_write_index64 failed: %d" % rc)


def read_index64(dict_path, index_path):
    post = C.POINTER(C.c_uint32)()
    uh = C.POINTER(C.c_uint64)()
    uc = C.POINTER(C.c_uint32)()
    nh = C.c_uint64()
    tot = C.c_uint64()
    rc = lib().ok_read_index64(dict_path.encode(), index_path.encode(), C.byref(post), C.byref(uh), C.byref(uc),
                               C.byref(nh), C.byref(tot))
    if rc:
        raise OSError("ok_read_index64 failed: %d" % rc)
    postings = np.ctypeslib.as_array(post, shape=(max(tot.value, 1),))[: tot.value].copy()
    uhash = np.ctypeslib.as_array(uh, shape=(max(nh.value, 1),))[: nh.value].copy()
    ucount = np.ctypeslib.as_array(uc, shape=(max(nh.value, 1),))[: nh.value].copy()
    for ptr in (post, uh, uc):
        lib().ok_free(ptr)
    return postings, uhash, ucount


def index_dist64(uhash, ucount, postings, ref_sizes, q_hashes, q_off, triangle, metric, kmer_size, max_dist,
                 threads=1, want_dense=False):
    uhash = np.ascontiguousarray(uhash, dtype=np.uint64)
    ucount = np.ascontiguousarray(ucount, dtype=np.uint32)
    postings = np.ascontiguousarray(postings, dtype=np.uint32)
    ref_sizes = np.ascontiguousarray(ref_sizes, dtype=np.uint32)
    q_hashes = np.ascontiguousarray(q_hashes, dtype=np.uint64)
    q_off = np.ascontiguousarray(q_off, dtype=np.uint64)
    nq, nr = len(q_off) - 1, len(ref_sizes)
    dense = np.zeros((nq, nr), dtype=np.int32) if want_dense else None
    hits = C.c_void_p()
    n = lib().ok_index_dist64(_p(uhash, C.c_uint64), _p(ucount, C.c_uint32), C.c_uint64(len(uhash)),
                              _p(postings, C.c_uint32), _p(ref_sizes, C.c_uint32), C.c_uint32(nr),
                              _p(q_hashes, C.c_uint64), _p(q_off, C.c_uint64), C.c_uint32(nq), int(triangle),
                              int(metric), int(kmer_size), C.c_double(max_dist), int(threads),
                              _p(dense, C.c_int32) if want_dense else None, C.byref(hits))
    if n < 0:
        raise MemoryError("ok_index_dist64 failed: %d" % n)
    buf = C.string_at(hits.value, n * HIT_DTYPE.itemsize) if n else b""
    lib().ok_free(hits)
    return np.frombuffer(buf, dtype=HIT_DTYPE).copy(), dense
