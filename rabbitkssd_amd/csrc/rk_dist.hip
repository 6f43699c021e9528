// rk_dist.hip -- intersection counting through the inverted index + distance epilogue.
// Replaces the row loops of index_tridist (src/dist.cpp:174-258) and index_dist
// (src/dist.cpp:560-692).
//
// One workgroup per (run of consecutive query rows, reference tile).  The counter row of the
// reference (`int intersectionArr[tid][numRef]`, src/dist.cpp:167) lives in LDS, two 16-bit
// counters per word when no count can overflow.  Per row: the posting ranges of the row's
// hashes are staged in LDS (the next row's are prefetched into registers meanwhile), the
// posting lists are gathered from HBM/L2 by 8-lane groups with 10 independent gathers in
// flight per lane and scattered into the LDS row with ds_add_u32.  The epilogue scans the row
// 16 B per lane, compacts the non-zero cells into an LDS list and evaluates the
// Jaccard/Mash (or containment/AafD) formula in FP64 one cell per lane; reported pairs are
// staged in LDS and flushed with one device-scope atomic per workgroup.
// Integer/index work: bound by VALU issue, LDS atomics and gather latency -- no MFMA.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "rk_internal.h"

namespace {

constexpr int kDistThreads = 256;
constexpr int kGroup = 8;                      // lanes per posting list
constexpr uint32_t kRowsPerXcdChunk = 16;      // consecutive rows kept on one XCD (their L2 shares a clade's postings)
constexpr uint32_t kStageHitsDefault = 24;     // reported pairs staged in LDS per workgroup
constexpr uint32_t kCandCapDefault = 128;      // non-zero cells of one row compacted in LDS

struct DistArgs {
    const uint2 *ranges;        // posting slices [x,y) of the query hashes, rows back to back
    const uint64_t *range_off;  // u64[n_query+1] offsets of each row's slices in `ranges`
    const uint64_t *size_off;   // u64[n_query+1] offsets whose differences are the sketch sizes
    const uint32_t *postings;
    const uint32_t *ref_sizes;
    uint32_t n_query, n_ref;
    uint32_t row_first, row_step, n_rows;
    uint32_t tile_cols, cnt_words, rows_per_wg, runs_per_chunk;
    uint32_t cand_cap, stage_hits;  // LDS carve-up (entries)
    int triangle, metric, kmer_size, dense_mode;
    double max_dist;
    double min_jorc;            // conservative lower bound on jaccard/containment of a reportable pair
    rk_hit *hits;
    unsigned long long cap;
    unsigned long long *n_hits;
    int32_t *common_dense;      // optional [n_query, n_ref]
};

// D3/D4: src/dist.cpp:218-231 and :238-250, FP64, same operation order.
// noinline: one copy of the FP64 divide + log sequence (~600 instructions) instead of one per call
// site keeps the kernel inside the instruction cache
__device__ __noinline__ void rk_distance(int common, int size0, int size1, int metric, int kmer_size,
                                         double &jorc, double &dist)
{
    if (!metric) {
        const int denom = size0 + size1 - common;
        double j = (size0 == 0 || size1 == 0) ? 0.0 : (double)common / (double)denom;
        double d;
        if (j == 1.0) d = 0.0;
        else if (j == 0.0) d = 1.0;
        else d = (-1.0 / (double)kmer_size) * log((2 * j) / (1.0 + j));
        jorc = j;
        dist = d;
    } else {
        const int denom = size0 < size1 ? size0 : size1;
        double c = (size0 == 0 || size1 == 0) ? 0.0 : (double)common / (double)denom;
        double d;
        if (c == 1.0) d = 0.0;
        else if (c == 0.0) d = 1.0;
        else d = (-1.0 / (double)kmer_size) * log(c);
        jorc = c;
        dist = d;
    }
}

// value held by lane (lane & 0x18) | J of the same half-wave: broadcast inside 8-lane groups
// (ds_swizzle bit mode: and_mask 0x18, or_mask J; LDS crossbar only, no memory, no index VALU)
template <int J> __device__ inline uint32_t group_bcast(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x18 | (J << 5));
}

// U16: two 16-bit counters per LDS word.  Valid when no count can reach 65536, i.e. the
// largest query sketch has < 65536 hashes (a count never exceeds |S_q|); halves the LDS row
// and doubles the resident rows per CU.
// FILTER=false: single reference tile and slices that already exclude ids <= row (the
// self-join slices of rk_index_build), so every posting lands in the row unchecked.
//
// One workgroup handles rows_per_wg consecutive row slots (strains of one clade share their
// posting lists: the second row finds them in this CU's L1/L2) and stages the reported
// pairs in LDS, so the contended device-scope atomic on the hit counter is paid once per
// workgroup instead of once per reporting wave.
template <bool U16, bool FILTER>
__global__ __launch_bounds__(kDistThreads) void rk_dist_kernel(DistArgs a)
{
    // one dynamic LDS region (16-byte aligned base):
    // counter row | non-zero cell list | staged hits | scalars
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *cnt = lds;
    uint2 *cand = reinterpret_cast<uint2 *>(lds + a.cnt_words);  // (col, common) of the current row
    rk_hit *stage = reinterpret_cast<rk_hit *>(cand + a.cand_cap);
    unsigned long long &s_base = *reinterpret_cast<unsigned long long *>(stage + a.stage_hits);
    uint32_t &s_total = *(reinterpret_cast<uint32_t *>(stage + a.stage_hits) + 2);
    uint32_t &s_cursor = *(reinterpret_cast<uint32_t *>(stage + a.stage_hits) + 3);
    const uint32_t kCandCap = a.cand_cap, kStageHits = a.stage_hits;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;

    // blockIdx.x -> run of row slots.  Workgroups are dealt round-robin over the 8 XCDs, so
    // blocks b, b+8, ... share an L2: consecutive runs of one XCD are adjacent rows, while
    // heavy (early) rows stay spread over all XCDs.
    const uint32_t xcd = blockIdx.x & 7, s8 = blockIdx.x >> 3;
    const uint32_t rpc = a.runs_per_chunk;
    const uint32_t run = ((s8 / rpc) * 8 + xcd) * rpc + s8 % rpc;
    const uint32_t slot0 = run * a.rows_per_wg;
    if (slot0 >= a.n_rows) return;
    const uint32_t col0 = blockIdx.y * a.tile_cols;
    const uint32_t col1 = min(a.n_ref, col0 + a.tile_cols);
    const uint32_t ncol = col1 - col0;
    if (tid == 0) s_cursor = 0;

    const uint32_t sub = lane % kGroup;
    const bool tri_filter = a.triangle && !a.common_dense;
    constexpr uint32_t kPerWord = U16 ? 2 : 1;
    const uint4 *c4 = reinterpret_cast<const uint4 *>(cnt);
    const uint32_t slot_end = min(a.n_rows, slot0 + a.rows_per_wg);

    auto row_of = [&](uint32_t slot) { return a.row_first + slot * a.row_step; };
    auto skipped = [&](uint32_t row) { return a.triangle && col1 <= row + 1 && !a.common_dense; };

    // ---- the run's work as one stream of batches -----------------------------------------
    // A batch = up to 256 posting slices of one row, one slice per thread, loaded straight
    // from HBM into a register pair (no LDS staging, no barrier).  Every row has at least one
    // (possibly empty) batch so that it flows through the pipeline and gets its epilogue.
    struct Cursor {
        uint32_t slot;      // row slot of the batch, slot_end = exhausted
        uint32_t b, nb;     // batch index / batches of that row
        uint64_t e0, e1;    // the row's slice range in `ranges`
    };
    auto open_row = [&](Cursor &c) {  // c.slot points at a candidate row: skip tiles below the diagonal
        while (c.slot < slot_end && skipped(row_of(c.slot))) c.slot++;
        c.b = 0;
        if (c.slot < slot_end) {
            c.e0 = a.range_off[row_of(c.slot)];
            c.e1 = a.range_off[row_of(c.slot) + 1];
            c.nb = max(1u, (uint32_t)((c.e1 - c.e0 + kDistThreads - 1) / kDistThreads));
        }
    };
    auto advance = [&](Cursor &c) {
        if (++c.b >= c.nb) { c.slot++; open_row(c); }
    };
    auto load_ranges = [&](const Cursor &c) -> uint2 {
        if (c.slot >= slot_end) return make_uint2(0, 0);
        const uint64_t e = c.e0 + (uint64_t)c.b * kDistThreads + tid;
        return e < c.e1 ? a.ranges[e] : make_uint2(0, 0);
    };

    struct Gathered {  // head postings of the 8 slices a lane's group walks in one batch
        uint32_t id[kGroup];
        bool ok[kGroup];
    };
    // a wave walks its 64 slices in 8 steps; in step j the 8-lane group g handles the slice
    // held by lane 8g+j (ds_swizzle broadcast inside the group), 8 gathers in flight per lane
    auto gather = [&](const uint2 rg, Gathered &g) {
        auto step = [&](int j, uint32_t rx, uint32_t ry) {
            const uint32_t k = rx + sub;
            g.ok[j] = k < ry;
            g.id[j] = a.postings[g.ok[j] ? k : 0];  // unconditional load, index 0 is always mapped
        };
        step(0, group_bcast<0>(rg.x), group_bcast<0>(rg.y));
        step(1, group_bcast<1>(rg.x), group_bcast<1>(rg.y));
        step(2, group_bcast<2>(rg.x), group_bcast<2>(rg.y));
        step(3, group_bcast<3>(rg.x), group_bcast<3>(rg.y));
        step(4, group_bcast<4>(rg.x), group_bcast<4>(rg.y));
        step(5, group_bcast<5>(rg.x), group_bcast<5>(rg.y));
        step(6, group_bcast<6>(rg.x), group_bcast<6>(rg.y));
        step(7, group_bcast<7>(rg.x), group_bcast<7>(rg.y));
        static_assert(kGroup == 8, "the step list above is written for 8-lane groups");
    };

    // ---- epilogue of one row (src/dist.cpp:207-255 / :600-682) ---------------------------
    auto epilogue = [&](uint32_t row) {
        if (a.common_dense) {
            int32_t *dst = a.common_dense + (size_t)row * a.n_ref + col0;
            for (uint32_t i = tid; i < ncol; i += kDistThreads)
                dst[i] = (int32_t)(U16 ? (cnt[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu : cnt[i]);
        }
        const int qsize = (int)(a.size_off[row + 1] - a.size_off[row]);
        const uint32_t jbeg = a.triangle ? max(col0, row + 1) : col0;  // :207 / :600

        // evaluates one (row, j) cell; returns true when it is reported
        auto evaluate = [&](uint32_t j, int common, rk_hit &hrec) -> bool {
            const int rs = (int)a.ref_sizes[j];
            const int size0 = a.triangle ? qsize : rs;  // :215-216 / :607-608
            const int size1 = a.triangle ? rs : qsize;
            // cheap exact-safe reject before the FP64 divide + log: the distance is monotone in
            // jaccard/containment and min_jorc sits strictly below the value at the threshold
            const int denom = a.metric ? min(size0, size1) : size0 + size1 - common;
            if ((double)common < a.min_jorc * (double)denom) return false;
            double jorc, dist;
            rk_distance(common, size0, size1, a.metric, a.kmer_size, jorc, dist);
            hrec.row = row;
            hrec.col = j;
            hrec.common = common;
            hrec.size0 = size0;
            hrec.size1 = size1;
            hrec.pad_ = 0;
            hrec.jorc = jorc;
            hrec.dist = dist;
            return a.triangle ? (dist < a.max_dist) : (dist <= a.max_dist);  // :232 / :624
        };
        auto stage_hit = [&](const rk_hit &hrec) {
            const uint32_t sl = atomicAdd(&s_cursor, 1u);
            if (sl < kStageHits) stage[sl] = hrec;
            else {  // staging full: rare, pay the device-scope atomic per hit
                const unsigned long long at = atomicAdd(a.n_hits, 1ULL);
                if (at < a.cap) a.hits[at] = hrec;
            }
        };

        if (!a.dense_mode) {
            // Sparse mode: the threshold excludes distance 1.0 (== common 0), so only cells
            // that share a hash can be reported.  Scan the LDS row 16 B per lane skipping
            // all-zero quads, compact the non-zero cells into an LDS list, then evaluate the
            // list one cell per lane (FP64 divide + log run in parallel, not serialised on
            // the lane that happened to own a clade's adjacent columns).
            const uint32_t q_first = ((jbeg - col0) / kPerWord) / 4;
            const uint32_t q_end = ((ncol + kPerWord - 1) / kPerWord + 3) / 4;
            for (uint32_t q = q_first + tid; q < q_end; q += kDistThreads) {
                const uint4 v = c4[q];
                if ((v.x | v.y | v.z | v.w) == 0) continue;
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int wi = 0; wi < 4; wi++) {
                    if (w[wi] == 0) continue;
#pragma unroll
                    for (uint32_t h = 0; h < kPerWord; h++) {
                        const uint32_t j = col0 + (q * 4 + wi) * kPerWord + h;
                        const uint32_t common = U16 ? (w[wi] >> (16 * h)) & 0xFFFFu : w[wi];
                        if (common == 0 || j < jbeg || j >= col1) continue;
                        const uint32_t sl = atomicAdd(&s_total, 1u);
                        if (sl < kCandCap) cand[sl] = make_uint2(j, common);
                        else {  // list full: evaluate in place (slow, only for very dense rows)
                            rk_hit hrec;
                            if (evaluate(j, (int)common, hrec)) stage_hit(hrec);
                        }
                    }
                }
            }
            __syncthreads();
            const uint32_t n_cand = min(s_total, kCandCap);
            for (uint32_t i = tid; i < n_cand; i += kDistThreads) {
                const uint2 cj = cand[i];
                rk_hit hrec;
                if (evaluate(cj.x, (int)cj.y, hrec)) stage_hit(hrec);
            }
        } else {
            // Dense mode: every cell of [jbeg, col1) can be reported.  One column per lane;
            // pass 0 counts this row's reports, one atomic reserves their slots, pass 1
            // re-evaluates and writes them.
            for (int pass_no = 0; pass_no < 2; pass_no++) {
                uint32_t mine = 0;
                for (uint32_t j = jbeg + tid; j < col1; j += kDistThreads) {
                    const uint32_t c = j - col0;
                    const int common = (int)(U16 ? (cnt[c >> 1] >> ((c & 1) * 16)) & 0xFFFFu : cnt[c]);
                    rk_hit hrec;
                    if (!evaluate(j, common, hrec)) continue;
                    if (pass_no == 0) { mine++; continue; }
                    const unsigned long long at = s_base + atomicAdd(&s_total, 1u);
                    if (at < a.cap) a.hits[at] = hrec;
                }
                if (pass_no == 0) {
                    if (mine) atomicAdd(&s_total, mine);
                    __syncthreads();
                    const uint32_t total = s_total;
                    __syncthreads();
                    if (total == 0) break;  // uniform
                    if (tid == 0) { s_base = atomicAdd(a.n_hits, (unsigned long long)total); s_total = 0; }
                    __syncthreads();
                }
            }
        }
    };

    // ---- main loop: the slices of the NEXT batch (same or next row) are always in flight while
    // the current batch is gathered and scattered.  (A deeper pipeline that also kept the
    // posting gathers one batch ahead measured slower: more live registers, same issue load.)
    Cursor cur_c{slot0, 0, 0, 0, 0};
    open_row(cur_c);
    if (cur_c.slot >= slot_end) return;  // nothing to do in this tile (uniform)
    Cursor nxt_c = cur_c;
    uint2 pre = load_ranges(cur_c);
    uint32_t cur_slot = 0xFFFFFFFFu;     // row whose counters are live in LDS

    while (cur_c.slot < slot_end) {
        const uint2 rg = pre;
        advance(nxt_c);
        pre = load_ranges(nxt_c);                            // slices of the next batch

        if (cur_c.slot != cur_slot) {                        // first batch of a new row
            if (cur_slot != 0xFFFFFFFFu) {
                __syncthreads();                             // all scatters of the previous row done
                epilogue(row_of(cur_slot));
            }
            __syncthreads();                                 // epilogue done with the LDS row / lists
            uint4 *z4 = reinterpret_cast<uint4 *>(cnt);      // memset row (src/dist.cpp:179)
            for (uint32_t i = tid; i < a.cnt_words / 4; i += kDistThreads) z4[i] = make_uint4(0, 0, 0, 0);
            if (tid == 0) s_total = 0;
            __syncthreads();
            cur_slot = cur_c.slot;
        }
        const uint32_t row = row_of(cur_slot);
        const uint32_t lo_id = tri_filter ? row + 1 : 0;     // ids below are not needed (j > i)
        auto bump = [&](uint32_t id, bool valid) {           // scatter, src/dist.cpp:199-202
            const uint32_t c = id - col0;
            if (valid && (!FILTER || (c < ncol && id >= lo_id))) {
                if (U16) atomicAdd(&cnt[c >> 1], (c & 1) ? 0x10000u : 1u);
                else atomicAdd(&cnt[c], 1u);
            }
        };
        Gathered g;
        gather(rg, g);
#pragma unroll
        for (int j = 0; j < kGroup; j++) bump(g.id[j], g.ok[j]);
        // lists longer than the group (1.7 % at 10,000 genomes): the whole wave streams the rest
        unsigned long long longs = __ballot(rg.y - rg.x > (uint32_t)kGroup);
        while (longs) {
            const int L = __ffsll((long long)longs) - 1;
            longs &= longs - 1;
            const uint32_t sx = __builtin_amdgcn_readlane(rg.x, L), sy = __builtin_amdgcn_readlane(rg.y, L);
            for (uint32_t k = sx + kGroup + lane; k < sy; k += 64) bump(a.postings[k], true);
        }
        advance(cur_c);
    }
    __syncthreads();
    epilogue(row_of(cur_slot));

    // flush the staged hits of this workgroup: one device-scope atomic, coalesced 8-byte stores
    __syncthreads();
    const uint32_t n_st = min(s_cursor, kStageHits);
    if (n_st == 0) return;
    if (tid == 0) s_base = atomicAdd(a.n_hits, (unsigned long long)n_st);
    __syncthreads();
    const unsigned long long at0 = s_base;
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(stage);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(a.hits);
    constexpr uint32_t kW = sizeof(rk_hit) / 8;
    for (uint32_t i = tid; i < n_st * kW; i += kDistThreads)
        if (at0 + i / kW < a.cap) dst[at0 * kW + i] = src[i];
}

inline uint32_t envu_chunk()
{
    const char *v = getenv("RK_DIST_XCD_ROWS");
    return v && atoi(v) > 0 ? (uint32_t)atoi(v) : kRowsPerXcdChunk;
}

struct Plan {
    uint32_t n_rows, tile_cols, n_tiles, cnt_words;
    uint32_t cand_cap, stage_hits, rows_per_wg;
    size_t lds_bytes;
    int dense_mode;
    bool u16;
};

int make_plan(rk_ctx *ctx, const rk_index *idx, uint32_t n_query, uint64_t max_query_size,
              const rk_dist_opts *o, Plan *p)
{
    if (o->kmer_size <= 0) return rk_fail(ctx, RK_ERR_ARG, "kmer_size must be positive");
    if (o->metric != 0 && o->metric != 1) return rk_fail(ctx, RK_ERR_ARG, "metric must be 0 or 1");
    const uint32_t step = o->row_step ? o->row_step : 1;
    p->n_rows = o->row_first < n_query ? (n_query - o->row_first + step - 1) / step : 0;
    // counter row in LDS; tile the reference range when it does not fit.  40 KiB rows let
    // four workgroups share a CU, which hides the posting-gather latency.
    p->u16 = max_query_size < 65536;
    auto envu = [](const char *name, uint32_t dflt) {
        const char *v = getenv(name);
        return v && atoi(v) > 0 ? (uint32_t)atoi(v) : dflt;
    };
    p->cand_cap = (envu("RK_DIST_CAND_CAP", kCandCapDefault) + 1) & ~1u;
    p->stage_hits = envu("RK_DIST_STAGE_HITS", kStageHitsDefault);
    p->rows_per_wg = envu("RK_DIST_ROWS", 2);
    const size_t fixed = (size_t)p->cand_cap * sizeof(uint2) + p->stage_hits * sizeof(rk_hit) + 64;
    const size_t lds_cap = (ctx->max_lds > 160 * 1024 ? 160 * 1024 : ctx->max_lds) - fixed;
    const uint32_t max_cols = (uint32_t)(p->u16 ? lds_cap / 2 : lds_cap / 4) & ~63u;
    uint32_t tile = idx->n_ref ? idx->n_ref : 1;
    if (tile > max_cols) {
        const uint32_t nt = (idx->n_ref + max_cols - 1) / max_cols;
        tile = ((idx->n_ref + nt - 1) / nt + 63) & ~63u;
    }
    p->tile_cols = tile;
    p->n_tiles = idx->n_ref ? (idx->n_ref + tile - 1) / tile : 1;
    p->cnt_words = ((p->u16 ? (tile + 1) / 2 : tile) + 3) & ~3u;  // whole 16-byte quads
    p->lds_bytes = (size_t)p->cnt_words * 4 + fixed;
    // does a pair with distance exactly 1.0 (common == 0) pass the threshold?
    p->dense_mode = o->triangle ? (1.0 < o->max_dist) : (1.0 <= o->max_dist);
    return RK_OK;
}

int launch_dist(rk_ctx *ctx, const rk_index *idx, const uint2 *ranges, const uint64_t *range_off,
                const uint64_t *size_off, uint32_t n_query, const rk_dist_opts *o, const Plan &p, rk_hit *hits_dev,
                uint64_t cap, unsigned long long *n_hits_dev, int32_t *dense_dev, hipStream_t stream)
{
    if (!p.n_rows || !idx->n_ref) return RK_OK;
    DistArgs a;
    a.ranges = ranges;
    a.range_off = range_off;
    a.size_off = size_off;
    a.postings = idx->d_postings;
    a.ref_sizes = idx->d_sizes;
    a.n_query = n_query;
    a.n_ref = idx->n_ref;
    a.row_first = o->row_first;
    a.row_step = o->row_step ? o->row_step : 1;
    a.n_rows = p.n_rows;
    a.tile_cols = p.tile_cols;
    a.cnt_words = p.cnt_words;
    a.triangle = o->triangle;
    a.metric = o->metric;
    a.kmer_size = o->kmer_size;
    a.dense_mode = p.dense_mode;
    a.max_dist = o->max_dist;
    // distance < D  <=>  jaccard > t/(2-t), t = exp(-k D)   (containment: c > t); 1e-6 relative slack
    // keeps the reject conservative, the exact formula still decides.  Disabled in dense mode.
    a.min_jorc = 0.0;
    if (!p.dense_mode && o->max_dist > 0.0) {
        const double t = exp(-(double)o->kmer_size * o->max_dist);
        a.min_jorc = (o->metric ? t : t / (2.0 - t)) * (1.0 - 1e-6);
    }
    a.hits = hits_dev;
    a.cap = cap;
    a.n_hits = n_hits_dev;
    a.common_dense = dense_dev;
    a.rows_per_wg = p.rows_per_wg;
    a.runs_per_chunk = std::max<uint32_t>(1, envu_chunk() / p.rows_per_wg);
    a.cand_cap = p.cand_cap;
    a.stage_hits = p.stage_hits;
    // postings need no range check when there is one tile and the ranges are the index's own
    // "later genomes" slices
    const bool filter = !(p.n_tiles == 1 && ranges == idx->d_selfrange && o->triangle && !dense_dev);
    void (*kern)(DistArgs);
    if (filter) kern = p.u16 ? rk_dist_kernel<true, true> : rk_dist_kernel<false, true>;
    else kern = p.u16 ? rk_dist_kernel<true, false> : rk_dist_kernel<false, false>;
    if (p.lds_bytes > 48 * 1024)
        RK_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)p.lds_bytes));
    const uint32_t runs = (p.n_rows + a.rows_per_wg - 1) / a.rows_per_wg;
    const uint32_t per = 8 * a.runs_per_chunk;  // grid padded to whole XCD chunks
    const uint32_t gx = (runs + per - 1) / per * per;
    hipLaunchKernelGGL(kern, dim3(gx, p.n_tiles), dim3(kDistThreads), p.lds_bytes, stream, a);
    RK_HIP(ctx, hipGetLastError());
    return RK_OK;
}

}  // namespace

extern "C" {

int rk_dist_rows_dev(rk_ctx *ctx, const rk_index *idx, const rk_sketches *queries,
                     const rk_dist_opts *opts, rk_hit *hits_dev, uint64_t hits_cap,
                     uint64_t *n_hits_dev, void *stream)
{
    if (!ctx || !idx || !opts || !n_hits_dev || (!hits_dev && hits_cap)) return RK_ERR_ARG;
    if (queries)
        return rk_fail(ctx, RK_ERR_UNSUPPORTED,
                       "rk_dist_rows_dev with explicit queries needs a workspace; use rk_dist_rows");
    if (!opts->triangle || !idx->d_selfrange)
        return rk_fail(ctx, RK_ERR_ARG, "queries == NULL needs triangle=1 and an index built by rk_index_build");
    Plan p;
    int rc = make_plan(ctx, idx, idx->n_ref, idx->max_src_size, opts, &p);
    if (rc) return rc;
    return launch_dist(ctx, idx, idx->d_selfrange, idx->d_self_off, idx->d_src_off, idx->n_ref, opts, p, hits_dev,
                       hits_cap, (unsigned long long *)n_hits_dev, nullptr, (hipStream_t)stream);
}

int rk_dist_rows(rk_ctx *ctx, const rk_index *idx, const rk_sketches *queries,
                 const rk_dist_opts *opts, rk_hit **hits_out, uint64_t *n_hits, int32_t *common_dense)
{
    if (!ctx || !idx || !opts || !hits_out || !n_hits) return RK_ERR_ARG;
    *hits_out = nullptr;
    *n_hits = 0;
    RK_HIP(ctx, hipSetDevice(ctx->device));
    const bool self = (queries == nullptr);
    if (self && (!opts->triangle || !idx->d_selfrange))
        return rk_fail(ctx, RK_ERR_ARG, "queries == NULL needs triangle=1 and an index built by rk_index_build");
    const uint32_t n_query = self ? idx->n_ref : queries->n;
    if (opts->triangle && n_query != idx->n_ref)
        return rk_fail(ctx, RK_ERR_ARG, "triangle mode needs the indexed sketches as queries (%u vs %u)",
                       n_query, idx->n_ref);
    Plan p;
    uint64_t max_q = idx->max_src_size;
    if (queries) {
        max_q = 0;
        for (uint32_t g = 0; g < queries->n; g++)
            max_q = std::max<uint64_t>(max_q, queries->h_off[g + 1] - queries->h_off[g]);
    }
    int rc = make_plan(ctx, idx, n_query, max_q, opts, &p);
    if (rc) return rc;

    // posting ranges of every query hash: precomputed for the self join, resolved through
    // the prefix directory otherwise (or when full counter rows are requested)
    const uint2 *ranges = idx->d_selfrange;
    const uint64_t *range_off = idx->d_self_off;
    const uint64_t *size_off = idx->d_src_off;
    DevBuf<uint2> resolved, kept;
    DevBuf<uint64_t> kept_off;
    if (!self || common_dense) {
        const rk_sketches *qs = queries;
        DevBuf<uint32_t> dummy;
        const void *qh = qs ? (qs->wide ? (const void *)qs->d_hashes64 : (const void *)qs->d_hashes) : nullptr;
        uint64_t qn = qs ? qs->total : idx->H;
        if (!qs)
            return rk_fail(ctx, RK_ERR_ARG, "common_dense needs explicit query sketches");
        if (qs->wide != idx->wide)
            return rk_fail(ctx, RK_ERR_ARG, "query sketches and index use different hash widths");
        RK_HIP(ctx, resolved.alloc(qn));
        rc = rk_resolve_ranges(ctx, idx, qh, qn, resolved.p, 0);
        if (rc) return rc;
        // most query hashes of an unrelated genome are absent from the index: drop their empty
        // slices so the kernel only walks real posting lists (size_off keeps the sketch sizes)
        uint64_t n_kept = 0;
        rc = rk_compact_ranges(ctx, resolved.p, qn, qs->d_off, qs->n, &kept.p, &kept_off.p, &n_kept, 0);
        if (rc) return rc;
        resolved.reset();
        ranges = kept.p;
        range_off = kept_off.p;
        size_off = qs->d_off;
    }

    DevBuf<int32_t> dense;
    if (common_dense) {
        RK_HIP(ctx, dense.alloc((size_t)n_query * idx->n_ref));
        RK_HIP(ctx, hipMemset(dense.p, 0, (size_t)n_query * idx->n_ref * 4));
    }
    DevBuf<unsigned long long> counter;
    RK_HIP(ctx, counter.alloc(1));

    // sparse mode: optimistic capacity, exact retry on overflow.  dense mode: every
    // selected (row, col) cell is a hit, so the count is known up front.
    uint64_t cap;
    if (p.dense_mode) {
        if (opts->triangle) {
            cap = 0;
            const uint32_t step = opts->row_step ? opts->row_step : 1;
            for (uint32_t r = opts->row_first; r < n_query; r += step) cap += idx->n_ref - 1 - r;
        } else cap = (uint64_t)p.n_rows * idx->n_ref;
    } else cap = std::max<uint64_t>(1 << 16, (uint64_t)p.n_rows * 64);
    std::vector<rk_hit> host;
    for (int attempt = 0; attempt < 2; attempt++) {
        DevBuf<rk_hit> hits;
        if (hits.alloc(cap) != hipSuccess)
            return rk_fail(ctx, RK_ERR_NOMEM, "cannot allocate %llu hit records on the device",
                           (unsigned long long)cap);
        RK_HIP(ctx, hipMemset(counter.p, 0, 8));
        rc = launch_dist(ctx, idx, ranges, range_off, size_off, n_query, opts, p, hits.p, cap, counter.p,
                         common_dense ? dense.p : nullptr, 0);
        if (rc) return rc;
        unsigned long long n = 0;
        RK_HIP(ctx, hipMemcpy(&n, counter.p, 8, hipMemcpyDeviceToHost));
        if (n > cap) {  // overflow: rerun with the exact count
            cap = n;
            continue;
        }
        rk_hit *out = (rk_hit *)malloc((n ? n : 1) * sizeof(rk_hit));
        if (!out) return rk_fail(ctx, RK_ERR_NOMEM, "host allocation of %llu hits failed", n);
        if (n) {
            hipError_t e = hipMemcpy(out, hits.p, n * sizeof(rk_hit), hipMemcpyDeviceToHost);
            if (e != hipSuccess) {
                free(out);
                return rk_fail(ctx, RK_ERR_HIP, "hit download failed: %s", hipGetErrorString(e));
            }
        }
        std::sort(out, out + n, [](const rk_hit &x, const rk_hit &y) {
            return x.row != y.row ? x.row < y.row : x.col < y.col;
        });
        *hits_out = out;
        *n_hits = n;
        if (common_dense)
            RK_HIP(ctx, hipMemcpy(common_dense, dense.p, (size_t)n_query * idx->n_ref * 4,
                                  hipMemcpyDeviceToHost));
        return RK_OK;
    }
    return rk_fail(ctx, RK_ERR_CAPACITY, "hit buffer overflow persisted after resize");
}

}  // extern "C"
