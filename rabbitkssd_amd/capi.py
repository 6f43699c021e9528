"""ctypes binding of librabbitkssd.so (include/rabbitkssd.h), used by tests and bench.py.

This is plumbing only: every call goes through the C ABI into the HIP kernels.  There is
no Python or CPU fallback -- a missing library or a missing GPU raises."""
import ctypes as C
import weakref
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librabbitkssd.so")
_LIB = None

HIT_DTYPE = np.dtype([("row", "<u4"), ("col", "<u4"), ("common", "<i4"), ("size0", "<i4"),
                      ("size1", "<i4"), ("pad", "<i4"), ("jorc", "<f8"), ("dist", "<f8")])

EXPORTS = [
    "rk_device_count", "rk_ctx_create", "rk_ctx_destroy", "rk_ctx_trim", "rk_ctx_pool_stats", "rk_ctx_set_timing", "rk_ctx_set_single_shot", "rk_ctx_last_ms", "rk_dist_kernel_name", "rk_last_error", "rk_version",
    "rk_free_host", "rk_pinned_alloc", "rk_pinned_free", "rk_dev_alloc", "rk_dev_free", "rk_stream_create",
    "rk_stream_destroy", "rk_stream_sync", "rk_upload_async", "rk_dev_copy_async", "rk_params_init", "rk_hash_bits", "rk_filter_create", "rk_filter_free",
    "rk_sketch_batch", "rk_sketch_batch_ex", "rk_sketch_packed_dev", "rk_sketch_packed_dev_ex", "rk_pack_layout", "rk_pack_genomes",
    "rk_sketches_from_host", "rk_sketches_from_host64", "rk_sketches_download64", "rk_sketches_is64", "rk_sketches_from_dev", "rk_sketches_count", "rk_sketches_total", "rk_sketches_windows",
    "rk_sketches_download", "rk_sketches_hashes_dev", "rk_sketches_off_dev", "rk_sketches_free",
    "rk_index_build", "rk_index_import", "rk_index_export", "rk_index_export_lists", "rk_index_import64", "rk_index_export64", "rk_index_total",
    "rk_index_distinct", "rk_index_genomes", "rk_index_order", "rk_index_hash_bits", "rk_index_built_fast", "rk_index_products", "rk_index_sum_sq", "rk_index_self_stats", "rk_index_tile_stats", "rk_index_build_shard", "rk_index_shard_records", "rk_index_shard_pack", "rk_index_join_shard", "rk_index_shard_exchange",
    "rk_index_free", "rk_index_blob_bytes", "rk_index_pack_dev", "rk_index_unpack_dev", "rk_index_broadcast", "rk_dist_rows", "rk_dist_rows_dev", "rk_topn_rows", "rk_format_hit",
]


class Params(C.Structure):
    _fields_ = [("half_k", C.c_int32), ("half_subk", C.c_int32), ("drlevel", C.c_int32),
                ("rev_add_move", C.c_int32), ("half_outctx_len", C.c_int32),
                ("dim_start", C.c_int32), ("dim_end", C.c_int32), ("kmer_size", C.c_uint32),
                ("domask", C.c_uint64), ("tupmask", C.c_uint64), ("undomask0", C.c_uint64),
                ("undomask1", C.c_uint64)]


class DistOpts(C.Structure):
    _fields_ = [("triangle", C.c_int32), ("metric", C.c_int32), ("kmer_size", C.c_int32),
                ("row_block", C.c_int32), ("max_dist", C.c_double), ("row_first", C.c_uint32),
                ("row_step", C.c_uint32)]


class RkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rabbitkssd error %d: %s" % (code, msg))
        self.code = code


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("librabbitkssd.so is not built (python -m rabbitkssd_amd.build); "
                              "there is no CPU fallback")
        # When the process also uses PyTorch-ROCm (tests and bench.py do, for device tensors and
        # torch.distributed), torch must load ITS HIP runtime first: both libraries then share
        # one libamdhip64.  The other order leaves torch without devices.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.rk_last_error.restype = C.c_char_p
        L.rk_version.restype = C.c_char_p
        L.rk_free_host.argtypes = [C.c_void_p]
        L.rk_ctx_destroy.argtypes = [C.c_void_p]
        L.rk_ctx_trim.argtypes = [C.c_void_p]
        L.rk_ctx_pool_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.rk_ctx_set_timing.argtypes = [C.c_void_p, C.c_int]
        L.rk_ctx_set_single_shot.argtypes = [C.c_void_p, C.c_int]
        L.rk_ctx_last_ms.argtypes = [C.c_void_p, C.c_int]
        L.rk_ctx_last_ms.restype = C.c_double
        for f in ("rk_filter_free", "rk_sketches_free", "rk_index_free"):
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("rk_sketches_total", "rk_sketches_windows", "rk_index_total", "rk_index_distinct",
                  "rk_index_sum_sq", "rk_index_blob_bytes"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [C.c_void_p]
        L.rk_sketches_is64.argtypes = [C.c_void_p]
        for f in ("rk_sketches_count", "rk_index_genomes"):
            getattr(L, f).restype = C.c_uint32
            getattr(L, f).argtypes = [C.c_void_p]
        L.rk_index_hash_bits.argtypes = [C.c_void_p]
        L.rk_index_built_fast.argtypes = [C.c_void_p]
        L.rk_index_products.argtypes = [C.c_void_p]
        for f in ("rk_sketches_hashes_dev", "rk_sketches_off_dev"):
            getattr(L, f).restype = C.c_void_p
            getattr(L, f).argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def params_init(half_k, half_subk, drlevel):
    p = Params()
    rc = lib().rk_params_init(half_k, half_subk, drlevel, C.byref(p))
    if rc:
        raise RkError(rc, "rk_params_init(%d,%d,%d)" % (half_k, half_subk, drlevel))
    return p


def hash_bits(p):
    return lib().rk_hash_bits(C.byref(p))


class Context:
    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().rk_ctx_create(int(device), C.byref(self._h))
        if rc:
            raise RkError(rc, "rk_ctx_create(device=%d) failed -- a GPU is required" % device)
        self.device = device
        self._objects = weakref.WeakSet()  # library objects must be freed before their context

    def close(self):
        if self._h:
            for o in list(self._objects):
                o.close()
            lib().rk_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def set_timing(self, on=True):
        lib().rk_ctx_set_timing(self._h, 1 if on else 0)

    def stream_read_gbs(self, mbytes=2048, reps=5):
        """GB/s of a streaming read over `mbytes` MiB of HBM (k_calib_read, 16 B per lane, HIP events): the measured denominator
        next to the nominal HBM peak"""
        L = lib()
        L.rk_debug_stream_read_gbs.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_double)]
        out = C.c_double()
        self.check(L.rk_debug_stream_read_gbs(self._h, C.c_uint64(int(mbytes) << 20), int(reps), C.byref(out)))
        return float(out.value)

    def set_single_shot(self, on=True):
        """a process that makes one pass (rk_ctx_set_single_shot): host-side ordering of small hit sets, no switch to the tile kernel"""
        lib().rk_ctx_set_single_shot(self._h, 1 if on else 0)

    def last_ms(self, which=0):
        """duration of the dominant kernel of the last pass (0: sketch kernel), HIP events on its stream"""
        return float(lib().rk_ctx_last_ms(self._h, int(which)))

    def dist_kernel_name(self, index, queries, triangle, metric, kmer_size, max_dist, row_first=0, row_step=1, row_block=0):
        opts = DistOpts(int(triangle), int(metric), int(kmer_size), int(row_block), float(max_dist),
                        int(row_first), int(row_step))
        buf = C.create_string_buffer(128)
        self.check(lib().rk_dist_kernel_name(self._h, index._h, queries._h if queries is not None else None, C.byref(opts),
                                             buf, C.c_size_t(128)))
        return buf.value.decode()

    def pool_stats(self):
        """(bytes from the driver, idle bytes in the cache, hipMalloc calls, hipFree calls) of the context's allocator"""
        out = (C.c_uint64 * 4)()
        lib().rk_ctx_pool_stats(self._h, out)
        return tuple(int(x) for x in out)

    def trim(self):
        """returns the device memory cached by the context's allocator to the driver"""
        lib().rk_ctx_trim(self._h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc:
            raise RkError(rc, lib().rk_last_error(self._h).decode())

    # ---- sketching
    def filter(self, params, shuffled_dim):
        tab = np.ascontiguousarray(shuffled_dim, dtype=np.int32)
        assert len(tab) == 1 << (4 * params.half_subk)
        h = C.c_void_p()
        self.check(lib().rk_filter_create(self._h, C.byref(params), _ptr(tab), C.byref(h)))
        return Filter(self, h, params)

    def sketch_batch(self, flt, seq, rec_off, genome_rec):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        genome_rec = np.ascontiguousarray(genome_rec, dtype=np.uint64)
        h = C.c_void_p()
        self.check(lib().rk_sketch_batch(self._h, flt._h, _ptr(seq), _ptr(rec_off),
                                         C.c_uint64(len(rec_off) - 1), _ptr(genome_rec),
                                         C.c_uint32(len(genome_rec) - 1), C.byref(h)))
        return Sketches(self, h)

    def sketch_batch_fastq(self, flt, seq, qual, rec_off, genome_rec, least_qual=0, min_count=1):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        genome_rec = np.ascontiguousarray(genome_rec, dtype=np.uint64)
        h = C.c_void_p()
        self.check(lib().rk_sketch_batch_ex(self._h, flt._h, _ptr(seq), _ptr(qual), int(least_qual),
                                            C.c_uint32(min_count), _ptr(rec_off), C.c_uint64(len(rec_off) - 1),
                                            _ptr(genome_rec), C.c_uint32(len(genome_rec) - 1), C.byref(h)))
        return Sketches(self, h)

    def sketch_packed_dev(self, flt, packed_dev_ptr, packed_bytes, gbeg, gend, stream=0):
        gbeg = np.ascontiguousarray(gbeg, dtype=np.uint64)
        gend = np.ascontiguousarray(gend, dtype=np.uint64)
        h = C.c_void_p()
        self.check(lib().rk_sketch_packed_dev(self._h, flt._h, C.c_void_p(packed_dev_ptr),
                                              C.c_uint64(packed_bytes), _ptr(gbeg), _ptr(gend),
                                              C.c_uint32(len(gbeg)), C.c_void_p(stream), C.byref(h)))
        return Sketches(self, h)

    def sketches_from_host(self, hashes, off):
        hashes = np.ascontiguousarray(hashes, dtype=np.uint32)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        h = C.c_void_p()
        self.check(lib().rk_sketches_from_host(self._h, _ptr(hashes), _ptr(off),
                                               C.c_uint32(len(off) - 1), C.byref(h)))
        return Sketches(self, h)

    def sketches_from_host64(self, hashes, off):
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        h = C.c_void_p()
        self.check(lib().rk_sketches_from_host64(self._h, _ptr(hashes), _ptr(off), C.c_uint32(len(off) - 1), C.byref(h)))
        return Sketches(self, h)

    def index_import64(self, postings, hashes, counts, hash_bits_, ref_sizes):
        postings = np.ascontiguousarray(postings, dtype=np.uint32)
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        ref_sizes = np.ascontiguousarray(ref_sizes, dtype=np.uint32)
        h = C.c_void_p()
        self.check(lib().rk_index_import64(self._h, _ptr(postings), C.c_uint64(len(postings)), _ptr(hashes),
                                           _ptr(counts), C.c_uint64(len(hashes)), int(hash_bits_),
                                           _ptr(ref_sizes), C.c_uint32(len(ref_sizes)), C.byref(h)))
        return Index(self, h)

    def sketches_from_dev(self, hashes_dev_ptr, off_dev_ptr, n_genomes):
        h = C.c_void_p()
        self.check(lib().rk_sketches_from_dev(self._h, C.c_void_p(hashes_dev_ptr), C.c_void_p(off_dev_ptr),
                                              C.c_uint32(n_genomes), C.byref(h)))
        return Sketches(self, h)

    def index_broadcast_from(self, index, n=1):
        """replicates `index` (of another context) onto this context n times is not meaningful; this helper copies it
        once onto this context (rk_index_broadcast with one destination)"""
        ctxs = (C.c_void_p * 1)(self._h)
        outs = (C.c_void_p * 1)()
        index.ctx.check(lib().rk_index_broadcast(index._h, ctxs, C.c_uint32(1), outs))
        return Index(self, C.c_void_p(outs[0]))

    def index_unpack_dev(self, blob_dev_ptr, blob_bytes, stream=0):
        h = C.c_void_p()
        self.check(lib().rk_index_unpack_dev(self._h, C.c_void_p(blob_dev_ptr), C.c_uint64(blob_bytes),
                                             C.c_void_p(stream), C.byref(h)))
        return Index(self, h)

    # ---- index
    def index_build(self, sketches, hash_bits_):
        h = C.c_void_p()
        self.check(lib().rk_index_build(self._h, sketches._h, int(hash_bits_), C.byref(h)))
        return Index(self, h)

    def index_build_shard(self, sketches, hash_bits_, shard, n_shards):
        """the lists of hash range `shard` of `n_shards` + their tile records grouped by destination shard (rk_index_build_shard)"""
        h = C.c_void_p()
        L = lib()
        L.rk_index_build_shard.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        self.check(L.rk_index_build_shard(self._h, sketches._h, int(hash_bits_), int(shard), int(n_shards), C.byref(h)))
        return Index(self, h)

    def index_join_shard(self, part, recv_dev_ptr, n_records):
        """a join-only index over this shard's rows from the tile records that arrived (rk_index_join_shard)"""
        h = C.c_void_p()
        L = lib()
        L.rk_index_join_shard.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
        self.check(L.rk_index_join_shard(self._h, part._h, C.c_void_p(recv_dev_ptr), C.c_uint64(n_records), C.byref(h)))
        return Index(self, h)

    def index_import(self, postings, counts, hash_bits_, ref_sizes):
        postings = np.ascontiguousarray(postings, dtype=np.uint32)
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        ref_sizes = np.ascontiguousarray(ref_sizes, dtype=np.uint32)
        assert len(counts) == 1 << hash_bits_
        h = C.c_void_p()
        self.check(lib().rk_index_import(self._h, _ptr(postings), C.c_uint64(len(postings)),
                                         _ptr(counts), int(hash_bits_), _ptr(ref_sizes),
                                         C.c_uint32(len(ref_sizes)), C.byref(h)))
        return Index(self, h)

    # ---- distances
    def dist_rows(self, index, queries, triangle, metric, kmer_size, max_dist, row_first=0,
                  row_step=1, want_dense=False, row_block=0):
        opts = DistOpts(int(triangle), int(metric), int(kmer_size), int(row_block), float(max_dist),
                        int(row_first), int(row_step))
        hits = C.c_void_p()
        n = C.c_uint64()
        dense = None
        if want_dense:
            nq = queries.count if queries is not None else index.genomes
            dense = np.zeros((nq, index.genomes), dtype=np.int32)
        self.check(lib().rk_dist_rows(self._h, index._h, queries._h if queries is not None else None,
                                      C.byref(opts), C.byref(hits), C.byref(n), _ptr(dense)))
        buf = C.string_at(hits.value, n.value * HIT_DTYPE.itemsize) if n.value else b""
        lib().rk_free_host(hits)
        return np.frombuffer(buf, dtype=HIT_DTYPE).copy(), dense

    def dist_rows_dev(self, index, triangle, metric, kmer_size, max_dist, hits_dev_ptr, hits_cap,
                      n_hits_dev_ptr, row_first=0, row_step=1, stream=0, row_block=0, queries=None):
        opts = DistOpts(int(triangle), int(metric), int(kmer_size), int(row_block), float(max_dist),
                        int(row_first), int(row_step))
        self.check(lib().rk_dist_rows_dev(self._h, index._h, queries._h if queries is not None else None, C.byref(opts),
                                          C.c_void_p(hits_dev_ptr), C.c_uint64(hits_cap),
                                          C.c_void_p(n_hits_dev_ptr), C.c_void_p(stream)))


class _Obj:
    _free = None

    def __init__(self, ctx, h):
        self.ctx, self._h = ctx, h
        ctx._objects.add(self)

    def close(self):
        if self._h:
            if self.ctx._h:  # a destroyed context has already released the device memory
                getattr(lib(), self._free)(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Filter(_Obj):
    _free = "rk_filter_free"

    def __init__(self, ctx, h, params):
        super().__init__(ctx, h)
        self.params = params


class Sketches(_Obj):
    _free = "rk_sketches_free"

    @property
    def count(self):
        return lib().rk_sketches_count(self._h)

    @property
    def total(self):
        return lib().rk_sketches_total(self._h)

    @property
    def windows(self):
        return lib().rk_sketches_windows(self._h)

    @property
    def is64(self):
        return bool(lib().rk_sketches_is64(self._h))

    def download(self):
        """(hashes, off); hashes are uint64 for the 64-bit layout, uint32 otherwise"""
        off = np.zeros(self.count + 1, dtype=np.uint64)
        if self.is64:
            hashes = np.zeros(self.total, dtype=np.uint64)
            self.ctx.check(lib().rk_sketches_download64(self._h, _ptr(hashes), _ptr(off)))
        else:
            hashes = np.zeros(self.total, dtype=np.uint32)
            self.ctx.check(lib().rk_sketches_download(self._h, _ptr(hashes), _ptr(off)))
        return hashes, off


class Index(_Obj):
    _free = "rk_index_free"

    @property
    def total(self):
        return lib().rk_index_total(self._h)

    @property
    def distinct(self):
        return lib().rk_index_distinct(self._h)

    @property
    def genomes(self):
        return lib().rk_index_genomes(self._h)

    @property
    def hash_bits(self):
        return lib().rk_index_hash_bits(self._h)

    @property
    def built_fast(self):
        return bool(lib().rk_index_built_fast(self._h))

    @property
    def products(self):
        """bit mask: 1 slice records, 2 tile records, 4 the tile records came with rk_index_build (rk_index_products)"""
        return int(lib().rk_index_products(self._h))

    @property
    def sum_sq(self):
        return lib().rk_index_sum_sq(self._h)

    @property
    def blob_bytes(self):
        return lib().rk_index_blob_bytes(self._h)

    @property
    def self_stats(self):
        """(slice records of the self join, of which compact, records the row-pair kernel walks, tile records of the tile
        kernel -- 0 until a self join has built them)"""
        out = (C.c_uint64 * 4)()
        self.ctx.check(lib().rk_index_self_stats(self._h, out))
        return int(out[0]), int(out[1]), int(out[2]), int(out[3])

    def shard_records(self, n_shards):
        """tile records this shard holds for every destination shard (rk_index_shard_records)"""
        out = (C.c_uint64 * int(n_shards))()
        L = lib()
        L.rk_index_shard_records.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        self.ctx.check(L.rk_index_shard_records(self._h, out))
        return [int(x) for x in out]

    def shard_pack(self, send_dev_ptr, stream=0):
        L = lib()
        L.rk_index_shard_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self.ctx.check(L.rk_index_shard_pack(self._h, C.c_void_p(send_dev_ptr), C.c_void_p(stream)))

    def tile_stats(self, triangle=1, metric=0, kmer_size=20, max_dist=0.05):
        """(tiles with records, tiles a launch with these options starts, tile records, record slots) -- rk_index_tile_stats"""
        opts = DistOpts(int(triangle), int(metric), int(kmer_size), 0, float(max_dist), 0, 1)
        out = (C.c_uint64 * 4)()
        L = lib()
        L.rk_index_tile_stats.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
        self.ctx.check(L.rk_index_tile_stats(self._h, C.byref(opts), out))
        return int(out[0]), int(out[1]), int(out[2]), int(out[3])

    @property
    def order(self):
        """orig[i] = caller's index of internal genome i (rk_index_order)"""
        orig = np.zeros(self.genomes, dtype=np.uint32)
        self.ctx.check(lib().rk_index_order(self._h, _ptr(orig)))
        return orig

    def shard_of(self, hits, row_step, row_block=1):
        """which row shard (rk_dist_opts.row_first) computes each hit of a self join: rows are dealt in blocks of row_block
        consecutive genomes of the INTERNAL order, and a pair belongs to the member that comes first in that order"""
        inv = np.empty(self.genomes, dtype=np.int64)
        inv[self.order] = np.arange(self.genomes)
        irow = np.minimum(inv[hits["row"]], inv[hits["col"]])
        return (irow // max(1, row_block)) % max(1, row_step)

    def pack_dev(self, blob_dev_ptr, blob_cap, stream=0):
        self.ctx.check(lib().rk_index_pack_dev(self._h, C.c_void_p(blob_dev_ptr), C.c_uint64(blob_cap),
                                               C.c_void_p(stream)))

    def export64(self):
        """(postings u32[H], hashes u64[U] ascending, counts u32[U]) -- the sparse .dict/.index content"""
        postings = np.zeros(self.total, dtype=np.uint32)
        hashes = np.zeros(self.distinct, dtype=np.uint64)
        counts = np.zeros(self.distinct, dtype=np.uint32)
        self.ctx.check(lib().rk_index_export64(self._h, _ptr(postings), _ptr(hashes), _ptr(counts)))
        return postings, hashes, counts

    def export(self, want_counts=True):
        postings = np.zeros(self.total, dtype=np.uint32)
        counts = np.zeros(1 << self.hash_bits, dtype=np.uint32) if want_counts else None
        self.ctx.check(lib().rk_index_export(self._h, _ptr(postings), _ptr(counts)))
        return postings, counts


def topn_rows(hits, max_neighbor):
    hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE).copy()
    n = C.c_uint64(len(hits))
    rc = lib().rk_topn_rows(_ptr(hits), C.byref(n), C.c_uint64(max_neighbor))
    if rc:
        raise RkError(rc, "rk_topn_rows")
    return hits[: n.value]


def format_hit(name_a, name_b, hit):
    rec = np.zeros(1, dtype=HIT_DTYPE)
    rec[0] = hit
    buf = C.create_string_buffer(len(name_a) + len(name_b) + 128)
    lib().rk_format_hit(buf, C.c_size_t(len(buf)), name_a.encode(), name_b.encode(), _ptr(rec))
    return buf.value.decode()


def pack_genomes(seq, rec_off, genome_rec):
    """host helper: (packed uint8 array, gbeg, gend) in the layout rk_sketch_packed_dev reads."""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
    genome_rec = np.ascontiguousarray(genome_rec, dtype=np.uint64)
    n = len(genome_rec) - 1
    gbeg = np.zeros(n, dtype=np.uint64)
    gend = np.zeros(n, dtype=np.uint64)
    nbytes = C.c_uint64()
    rc = lib().rk_pack_layout(_ptr(rec_off), C.c_uint64(len(rec_off) - 1), _ptr(genome_rec),
                              C.c_uint32(n), _ptr(gbeg), _ptr(gend), C.byref(nbytes))
    if rc:
        raise RkError(rc, "rk_pack_layout")
    packed = np.zeros(nbytes.value, dtype=np.uint8)
    rc = lib().rk_pack_genomes(_ptr(seq), _ptr(rec_off), C.c_uint64(len(rec_off) - 1),
                               _ptr(genome_rec), C.c_uint32(n), _ptr(gbeg), _ptr(packed), nbytes)
    if rc:
        raise RkError(rc, "rk_pack_genomes")
    return packed, gbeg, gend
