// rk_internal.h -- shared declarations of librabbitkssd.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "rabbitkssd.h"

struct rk_ctx {
    int device = 0;
    int num_cu = 0;
    size_t max_lds = 0;  // bytes of LDS one workgroup may use
    std::string err;
};

int rk_fail(rk_ctx *ctx, int code, const char *fmt, ...);

#define RK_HIP(ctx, call)                                                                   \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess)                                                              \
            return rk_fail((ctx), RK_ERR_HIP, "%s failed: %s (%s:%d)", #call,               \
                           hipGetErrorString(e__), __FILE__, __LINE__);                     \
    } while (0)

// owning device pointer; freed on scope exit unless release()d
template <class T> struct DevBuf {
    T *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { reset(); }
    hipError_t alloc(size_t n) {
        reset();
        return hipMalloc(reinterpret_cast<void **>(&p), (n ? n : 1) * sizeof(T));
    }
    void reset() {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    T *release() {
        T *q = p;
        p = nullptr;
        return q;
    }
    operator T *() const { return p; }
};

struct rk_filter {
    rk_ctx *ctx = nullptr;
    rk_params params{};
    int32_t *d_table = nullptr;   // the full .shuf table, int32[16^half_subk]
    uint32_t *d_bitmap = nullptr; // 64 KiB LDS filter image (bitmap [+ exact key/value table])
    int bitmap_bits = 0;
    bool exact = false;           // image holds the exact table: survivors never leave the CU
    uint32_t n_keys = 0;          // entries with value in [dim_start, dim_end)
};

struct rk_sketches {
    rk_ctx *ctx = nullptr;
    uint32_t n = 0;
    uint64_t total = 0;
    uint64_t windows = 0;
    bool wide = false;            // 64-bit hash layout (use64: half_k - drlevel > 8)
    uint32_t *d_hashes = nullptr; // u32[total] when !wide
    uint64_t *d_hashes64 = nullptr; // u64[total] when wide
    uint64_t *d_off = nullptr;
    std::vector<uint64_t> h_off;  // host mirror of d_off (n+1)
};

struct rk_index {
    rk_ctx *ctx = nullptr;
    uint32_t n_ref = 0;
    uint64_t H = 0;         // postings
    uint64_t U = 0;         // distinct hashes
    int hash_bits = 0;
    int dir_bits = 0, dir_shift = 0;
    uint64_t sum_sq = 0;
    uint64_t max_src_size = 0;       // largest source sketch (built index only)
    uint32_t *d_postings = nullptr;  // u32[H]   (.dict order)
    bool wide = false;               // 64-bit hashes: d_uhash64 instead of d_uhash
    uint32_t *d_uhash = nullptr;     // u32[U]   sorted distinct hashes
    uint64_t *d_uhash64 = nullptr;   // u64[U]   (wide)
    uint32_t *d_upos = nullptr;      // u32[U+1] posting offsets
    uint32_t *d_dir = nullptr;       // u32[2^dir_bits+1] prefix directory into d_uhash
    uint32_t *d_sizes = nullptr;     // u32[n_ref] sketch sizes
    uint2 *d_selfrange = nullptr;    // uint2[n_self]: per source element with a non-empty slice,
                                     // the postings of LATER genomes (rows back to back)
    uint64_t *d_self_off = nullptr;  // u64[n_ref+1] offsets of each genome's slices in d_selfrange
    uint64_t *d_self_split = nullptr; // u64[n_ref]: a row's slices from here on are "covered" by its pair partner
                                      // (genome 2p+1 shares the hash with 2p): the pair kernel skips them
    uint64_t n_self = 0;
    uint64_t *d_src_off = nullptr;   // u64[n_ref+1] offsets of the source sketches (built only)
};

// kernels/launchers implemented in the .hip files
// drops the empty slices of `ranges` (rows delimited by off_dev[n_rows+1]); outputs are hipMalloc'd
int rk_compact_ranges(rk_ctx *ctx, const uint2 *ranges_dev, uint64_t n, const uint64_t *off_dev, uint32_t n_rows,
                      uint2 **out_ranges_dev, uint64_t **out_off_dev, uint64_t *n_out, hipStream_t stream);
// q_hashes_dev: u32[] or u64[] matching idx->wide
int rk_resolve_ranges(rk_ctx *ctx, const rk_index *idx, const void *q_hashes_dev, uint64_t n,
                      uint2 *ranges_dev, hipStream_t stream);
