// rk_dist_common.h -- device helpers shared by the distance kernels (rk_dist.hip: the all-vs-all self join,
// rk_distq.hip: explicit queries).  Not part of the public ABI.
#pragma once
#include "rk_internal.h"

namespace {

// D3/D4: src/dist.cpp:218-231 and :238-250, FP64, same operation order.
// noinline: one copy of the FP64 divide + log sequence (~600 instructions) instead of one per call
// site keeps the kernel inside the instruction cache; the pair is returned in registers
struct JorcDist {
    double jorc, dist;
};
__device__ __noinline__ JorcDist rk_distance(int common, int size0, int size1, int metric, int kmer_size)
{
    JorcDist r;
    if (!metric) {
        const int denom = size0 + size1 - common;
        double j = (size0 == 0 || size1 == 0) ? 0.0 : (double)common / (double)denom;
        double d;
        if (j == 1.0) d = 0.0;
        else if (j == 0.0) d = 1.0;
        else d = (-1.0 / (double)kmer_size) * log((2 * j) / (1.0 + j));
        r.jorc = j;
        r.dist = d;
    } else {
        const int denom = size0 < size1 ? size0 : size1;
        double c = (size0 == 0 || size1 == 0) ? 0.0 : (double)common / (double)denom;
        double d;
        if (c == 1.0) d = 0.0;
        else if (c == 0.0) d = 1.0;
        else d = (-1.0 / (double)kmer_size) * log(c);
        r.jorc = c;
        r.dist = d;
    }
    return r;
}

// value held by lane J of the same quad (DPP quad_perm:[J,J,J,J]: full-rate VALU, no LDS)
template <int J> __device__ inline uint32_t quad_bcast(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, J * 0x55, 0xF, 0xF, true);
}

// two consecutive postings; dword-aligned only (a slice starts anywhere)
struct __attribute__((packed, aligned(4))) PostingPair {
    uint32_t x, y;
};

}  // namespace
