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
__host__ __device__ __noinline__ JorcDist rk_distance(int common, int size0, int size1, int metric, int kmer_size)
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

// bitwise OR over the 64 lanes of the wave (uniform result): four row_shr steps leave the OR of a 16-lane row in its last
// lane, row_bcast:15 / row_bcast:31 carry it on to lane 63
__device__ inline uint32_t wave_or(uint32_t x)
{
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);  // row_shr:1
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);  // row_shr:2
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);  // row_shr:4
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);  // row_shr:8
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, true);  // row_bcast:15 -> rows 1, 3
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, true);  // row_bcast:31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

// unsigned minimum over the 64 lanes of the wave (uniform result); same data movement as wave_or (lanes without a
// source keep ~0, the neutral element)
#define RK_MIN_STEP(CTRL, ROWS)                                                                          \
    do {                                                                                                 \
        const uint32_t o__ = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, CTRL, ROWS, 0xF, false); \
        x = o__ < x ? o__ : x;                                                                           \
    } while (0)
__device__ inline uint32_t wave_min(uint32_t x)
{
    RK_MIN_STEP(0x111, 0xF);  // row_shr:1
    RK_MIN_STEP(0x112, 0xF);  // row_shr:2
    RK_MIN_STEP(0x114, 0xF);  // row_shr:4
    RK_MIN_STEP(0x118, 0xF);  // row_shr:8
    RK_MIN_STEP(0x142, 0xA);  // row_bcast:15 -> rows 1, 3
    RK_MIN_STEP(0x143, 0xC);  // row_bcast:31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
#undef RK_MIN_STEP

// Wave-level column counts.  Every lane holds the bitmask of one list (bit p = column p of a 32-column window) for the
// unit's first row (ma) and its partner (mb).  For every occupied column one ballot counts the lanes that name it and
// the count travels from the scalar unit straight into lane p (v_writelane_b32 with the lane as an inline constant: no
// compare + select); lane p ends up with the counts of column p.  Recursion instead of a loop: the lane select must be a
// compile-time constant.
template <bool PAIR, int P>
__device__ inline void count_columns(uint32_t occupied, uint32_t ma, uint32_t mb, uint32_t &ca, uint32_t &cb)
{
    if constexpr (P < 32) {
        if ((occupied >> P) == 0) return;  // uniform: nothing from here on
        if ((occupied >> P) & 1u) {        // uniform: some list names this column
            const int na = __popcll(__ballot(ma & (1u << P)));
            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(ca) : "s"(na), "n"(P));
            if (PAIR) {
                const int nb = __popcll(__ballot(mb & (1u << P)));
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(cb) : "s"(nb), "n"(P));
            }
        }
        count_columns<PAIR, P + 1>(occupied, ma, mb, ca, cb);
    }
}

// two consecutive postings; dword-aligned only (a slice starts anywhere)
struct __attribute__((packed, aligned(4))) PostingPair {
    uint32_t x, y;
};

}  // namespace
