// rk_prims_scan.hip -- more of the device-wide primitives (see rk_prims.hip: one translation unit per group, so that a call loads
// only the code object it needs).  rk_prims.hip -- the device-wide primitives (rocprim: radix sorts, scans, unique, run-length encode, select) behind plain
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

int rk_prim_inclusive_scan_u32(rk_ctx *ctx, const uint32_t *in, uint32_t *out, uint64_t n, hipStream_t st)
{
    if (!n) return RK_OK;
    RK_TWO_PHASE(rocprim::inclusive_scan(tmp_p, tb, in, out, (size_t)n, rocprim::plus<uint32_t>(), st));
    return RK_OK;
}

int rk_prim_exclusive_scan_u32(rk_ctx *ctx, const uint32_t *in, uint32_t *out, uint64_t n, hipStream_t st)
{
    if (!n) return RK_OK;
    RK_TWO_PHASE(rocprim::exclusive_scan(tmp_p, tb, in, out, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
    return RK_OK;
}

int rk_prim_unique_u64(rk_ctx *ctx, const unsigned long long *in, unsigned long long *out, unsigned long long *n_out_dev, uint64_t n, hipStream_t st)
{
    RK_TWO_PHASE(rocprim::unique(tmp_p, tb, in, out, n_out_dev, (size_t)n, rocprim::equal_to<unsigned long long>(), st));
    return RK_OK;
}

int rk_prim_rle_u64(rk_ctx *ctx, const unsigned long long *in, uint64_t n, unsigned long long *runs, unsigned int *counts, unsigned long long *n_runs_dev,
                    hipStream_t st)
{
    RK_TWO_PHASE(rocprim::run_length_encode(tmp_p, tb, in, (size_t)n, runs, counts, n_runs_dev, st));
    return RK_OK;
}

int rk_prim_select_u64(rk_ctx *ctx, const unsigned long long *in, const unsigned char *flags, unsigned long long *out, unsigned long long *n_out_dev,
                       uint64_t n, hipStream_t st)
{
    RK_TWO_PHASE(rocprim::select(tmp_p, tb, in, flags, out, n_out_dev, (size_t)n, st));
    return RK_OK;
}
