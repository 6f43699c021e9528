// rk_prims.hip -- the device-wide primitives (rocprim: radix sorts, scans, unique, run-length encode, select) behind plain
// functions.  A translation unit of its own on purpose: every rocprim algorithm instantiates dozens of kernels (one per
// tuning configuration), and the HIP runtime loads the whole code object of a translation unit when its first kernel is
// launched -- 3.3 ms per MB of object file, measured (tools/module_load_timing.py): with the primitives inline, the first
// rk_sketches_from_host cost 33 ms and the first rk_index_build 30 ms for code that the usual call never runs (the
// bucket-sort index build, sorted sketches and the LDS dedup use none of it).  The hit-record sort lives in
// rk_prims_hits.hip for the same reason.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "rk_internal.h"

#define RK_PRIM(call)                                                                                         \
    do {                                                                                                      \
        hipError_t e__ = (call);                                                                              \
        if (e__ != hipSuccess) return rk_fail(ctx, RK_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e__)); \
    } while (0)

// two-phase rocprim call: size query, temporary from the context's pool, the call itself (enqueued on `st`; the temporary
// goes back to the pool behind an event on `st`: rk_pool_free_after)
#define RK_TWO_PHASE(EXPR)                                      \
    do {                                                        \
        size_t tb = 0;                                          \
        void *tmp_p = nullptr;                                  \
        RK_PRIM(EXPR);                                          \
        DevBuf<char> tmp(ctx);                                  \
        if (tmp.alloc(tb) != hipSuccess) return rk_fail(ctx, RK_ERR_NOMEM, "cannot allocate %zu bytes of sort/scan scratch", tb); \
        tmp_p = tmp.p;                                          \
        RK_PRIM(EXPR);                                          \
        rk_pool_free_after(ctx, tmp.release(), st);             \
    } while (0)

int rk_prim_sort_keys_u64(rk_ctx *ctx, const unsigned long long *in, unsigned long long *out, uint64_t n, unsigned begin_bit, unsigned end_bit,
                          hipStream_t st, void **tmp_keep)
{
    if (tmp_keep) *tmp_keep = nullptr;
    if (!n) return RK_OK;
    size_t tb = 0;
    RK_PRIM(rocprim::radix_sort_keys(nullptr, tb, in, out, (size_t)n, begin_bit, end_bit, st));
    DevBuf<char> tmp(ctx);
    if (tmp.alloc(tb) != hipSuccess) return rk_fail(ctx, RK_ERR_NOMEM, "cannot allocate %zu bytes of sort scratch", tb);
    RK_PRIM(rocprim::radix_sort_keys(tmp.p, tb, in, out, (size_t)n, begin_bit, end_bit, st));
    if (tmp_keep) *tmp_keep = tmp.release();   // (the caller frees it once `st` is done: a stream other than the context's)
    else rk_pool_free_after(ctx, tmp.release(), st);
    return RK_OK;
}

int rk_prim_sort_pairs_u32_u32(rk_ctx *ctx, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, uint64_t n, unsigned end_bit,
                               hipStream_t st)
{
    if (!n) return RK_OK;
    RK_TWO_PHASE(rocprim::radix_sort_pairs(tmp_p, tb, kin, kout, vin, vout, (size_t)n, 0, end_bit, st));
    return RK_OK;
}

int rk_prim_sort_pairs_u64_u32(rk_ctx *ctx, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout, uint64_t n, unsigned end_bit,
                               hipStream_t st)
{
    if (!n) return RK_OK;
    RK_TWO_PHASE(rocprim::radix_sort_pairs(tmp_p, tb, kin, kout, vin, vout, (size_t)n, 0, end_bit, st));
    return RK_OK;
}
