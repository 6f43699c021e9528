// rk_prims_hits.hip -- ordering a big result by (row, col) on the device: a radix sort of (key, 40-byte hit record) pairs.
// Its own translation unit (see rk_prims.hip): a command-line run that reports 45,000 pairs orders them on the host in
// less time than loading this code object takes.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "rk_internal.h"

int rk_prim_sort_hits(rk_ctx *ctx, const unsigned long long *keys, unsigned long long *keys_out, const rk_hit *hits, rk_hit *hits_out, uint64_t n,
                      unsigned end_bit, hipStream_t st)
{
    if (!n) return RK_OK;
    size_t tb = 0;
    if (rocprim::radix_sort_pairs(nullptr, tb, keys, keys_out, hits, hits_out, (size_t)n, 0, end_bit, st) != hipSuccess) return RK_ERR_HIP;
    DevBuf<char> tmp(ctx);
    if (tmp.alloc(tb) != hipSuccess) return RK_ERR_NOMEM;
    if (rocprim::radix_sort_pairs(tmp.p, tb, keys, keys_out, hits, hits_out, (size_t)n, 0, end_bit, st) != hipSuccess) return RK_ERR_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return RK_ERR_HIP;   // (the temporary returns to the pool)
    return RK_OK;
}
