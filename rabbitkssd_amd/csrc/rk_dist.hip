// rk_dist.hip -- intersection counting through the inverted index + distance epilogue.
// Replaces the row loops of index_tridist (src/dist.cpp:174-258) and index_dist
// (src/dist.cpp:560-692).
//
// One workgroup per (run of consecutive query rows, reference tile).  The
// counter row of the reference (`int intersectionArr[tid][numRef]`, src/dist.cpp:167) lives in
// LDS, two 16-bit counters per word when no count can overflow.  Per row: the posting slices
// of the row's hashes stream from HBM into registers one per thread (the next batch is always
// in flight), the posting lists are gathered from HBM/L2 by 8-lane groups with 8-10
// independent gathers in flight per lane and scattered into the LDS row with ds_add_u32.  The
// epilogue scans the row 16 B per lane, compacts the non-zero cells into an LDS list and
// evaluates the Jaccard/Mash (or containment/AafD) formula in FP64 one cell per lane; reported
// pairs are staged in LDS and flushed with one device-scope atomic per workgroup.
// Integer/index work: bound by VALU issue, LDS atomics and gather latency -- no MFMA.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "rk_internal.h"

namespace {

constexpr int kGroup = 8;                      // postings per slice walked by a quad (4 lanes x 2 postings)
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

#ifdef RK_DIST_PROFILE
// developer build only (-DRK_DIST_PROFILE, tools/dist_phase_profile.py): per-phase wave cycles
constexpr int kProfWaves = 1 << 16;
__device__ unsigned long long g_prof[kProfWaves][16];  // one row per wave, summed on the host
#define PROF_MARK(i)                                                    \
    do {                                                                \
        const long long t_now = clock64();                              \
        prof_acc[i] += t_now - prof_t;                                  \
        prof_t = t_now;                                                 \
    } while (0)
#define PROF_FENCE() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define PROF_FLUSH()                                                                         \
    do {                                                                                     \
        const uint32_t wv = (blockIdx.x * (kDistThreads / 64) + tid / 64) % kProfWaves;      \
        if (lane == 0) {                                                                     \
            for (int i = 0; i < 12; i++) g_prof[wv][i] += (unsigned long long)prof_acc[i];   \
            g_prof[wv][12] += (unsigned long long)(clock64() - prof_t0);                     \
            g_prof[wv][13] += 1ULL;                                                          \
            g_prof[wv][14] = (unsigned long long)prof_w0;                                    \
            g_prof[wv][15] = (unsigned long long)wall_clock64();                             \
        }                                                                                    \
    } while (0)
#else
#define PROF_MARK(i)
#define PROF_FENCE()
#define PROF_FLUSH()
#endif

// U16: two 16-bit counters per LDS word.  Valid when no count can reach 65536, i.e. the
// largest query sketch has < 65536 hashes (a count never exceeds |S_q|); halves the LDS row
// and doubles the resident rows per CU.
// FILTER=false: single reference tile and slices that already exclude ids <= row (the
// self-join slices of rk_index_build), so every posting lands in the row unchecked.
//
// One workgroup handles rows_per_wg consecutive row slots (strains of one clade share their
// posting lists: the second row finds them in this CU's L1/L2) and stages the reported pairs in LDS, so the contended device-scope atomic on
// the hit counter is paid once per workgroup instead of once per reporting wave.
// Per row: zero the LDS row | barrier | gather + scatter all slices | barrier | scan the row
// into the cell list | barrier | evaluate the cells.  Three barriers; the evaluation of row r
// overlaps the zeroing and scattering of row r+1 of faster waves (cell counters alternate).
//
// THREADS: a counter row of N columns occupies 2N (U16) or 4N bytes of the CU's 160 KiB, which
// caps the resident workgroups; bigger rows get bigger workgroups so that the CU keeps 16-32
// waves to hide the gather latency (measured at 28,284 / 50,000 columns: 512 / 1024 threads are
// 1.25x / 1.9x faster than 256).
template <bool U16, bool FILTER, int THREADS>
__global__ __launch_bounds__(THREADS) void rk_dist_kernel(DistArgs a)
{
    constexpr uint32_t kDistThreads = THREADS;
#ifdef RK_DIST_PROFILE
    long long prof_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long prof_t = clock64();
    const long long prof_t0 = prof_t;
    const long long prof_w0 = wall_clock64();
#endif
    // one dynamic LDS region (16-byte aligned base):
    // counter row | non-zero cell list | staged hits | scalars
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *cnt = lds;
    uint2 *cand = reinterpret_cast<uint2 *>(lds + a.cnt_words);  // (col, common) of the current row
    rk_hit *stage = reinterpret_cast<rk_hit *>(cand + a.cand_cap);
    uint32_t *scal = reinterpret_cast<uint32_t *>(stage + a.stage_hits);
    unsigned long long &s_base = *reinterpret_cast<unsigned long long *>(scal);
    uint32_t *s_cells = scal + 2;   // [2] cells of the row being scanned (alternating per row)
    uint32_t &s_cursor = scal[4];   // staged hits
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

    const uint32_t sub = lane & 3;  // lane of the quad
    const bool tri_filter = a.triangle && !a.common_dense;
    constexpr uint32_t kPerWord = U16 ? 2 : 1;
    const uint4 *c4 = reinterpret_cast<const uint4 *>(cnt);
    const uint32_t slot_end = min(a.n_rows, slot0 + a.rows_per_wg);

    auto row_of = [&](uint32_t slot) { return a.row_first + slot * a.row_step; };
    auto skipped = [&](uint32_t row) { return a.triangle && col1 <= row + 1 && !a.common_dense; };
    // first slot >= s whose row has work in this tile
    auto next_live = [&](uint32_t s, uint32_t &row) {
        while (s < slot_end) {
            row = row_of(s);
            if (!skipped(row)) break;
            s++;
        }
        return s;
    };
    auto load_slice = [&](uint64_t e, uint64_t e1) -> uint2 {
        return e < e1 ? a.ranges[e] : make_uint2(0, 0);
    };

    struct Gathered {  // head postings of the 4 slices a lane's quad walks in one batch
        PostingPair id[4];
        bool ok0[4], ok1[4];
    };
    // A wave walks its 64 slices in 4 steps; in step j quad q handles the slice held by lane
    // 4q+j: each lane fetches two postings (8 B), the quad the first 8 of the list.  The L1
    // looks up tags one quad per cycle, so a slice costs one lookup (two when it straddles a
    // line) and no two lookups of a step wait on the same in-flight line.
    auto gather = [&](const uint2 rg, Gathered &g) {
        auto step = [&](int j, uint32_t rx, uint32_t ry) {
            const uint32_t k = rx + 2 * sub;
            g.ok0[j] = k < ry;
            g.ok1[j] = k + 1 < ry;
            // unconditional load, postings[0..1] are always mapped
            g.id[j] = *reinterpret_cast<const PostingPair *>(a.postings + (g.ok0[j] ? k : 0));
        };
        step(0, quad_bcast<0>(rg.x), quad_bcast<0>(rg.y));
        step(1, quad_bcast<1>(rg.x), quad_bcast<1>(rg.y));
        step(2, quad_bcast<2>(rg.x), quad_bcast<2>(rg.y));
        step(3, quad_bcast<3>(rg.x), quad_bcast<3>(rg.y));
        static_assert(kGroup == 8, "a quad covers 4 lanes x 2 postings");
    };

    // ---- epilogue of one row (src/dist.cpp:207-255 / :600-682) ---------------------------
    auto epilogue = [&](uint32_t row, uint32_t &s_total) {
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
            const JorcDist jd = rk_distance(common, size0, size1, a.metric, a.kmer_size);
            hrec.row = row;
            hrec.col = j;
            hrec.common = common;
            hrec.size0 = size0;
            hrec.size1 = size1;
            hrec.pad_ = 0;
            hrec.jorc = jd.jorc;
            hrec.dist = jd.dist;
            return a.triangle ? (jd.dist < a.max_dist) : (jd.dist <= a.max_dist);  // :232 / :624
        };
        auto stage_hit = [&](const rk_hit &hrec) {
            const uint32_t sl = atomicAdd(&s_cursor, 1u);
            if (sl < kStageHits) stage[sl] = hrec;
            else {  // staging full: rare, pay the device-scope atomic per hit
                const unsigned long long at = atomicAdd(a.n_hits, 1ULL);
                if (at < a.cap) a.hits[at] = hrec;
            }
        };
        auto cell = [&](uint32_t c) -> uint32_t {
            return U16 ? (cnt[c >> 1] >> ((c & 1) * 16)) & 0xFFFFu : cnt[c];
        };

        if (!a.dense_mode) {
            // Sparse mode: the threshold excludes distance 1.0 (== common 0), so only cells
            // that share a hash can be reported.  Scan the LDS row 16 B per lane skipping
            // all-zero quads and compact the non-zero cells into an LDS list (one LDS atomic
            // per lane that found any), then evaluate the list one cell per lane: the FP64
            // divide + log run in parallel, not serialised on the lane that happened to own a
            // clade's adjacent columns.
            const uint32_t q_first = ((jbeg - col0) / kPerWord) / 4;
            const uint32_t q_end = ((ncol + kPerWord - 1) / kPerWord + 3) / 4;
            for (uint32_t q = q_first + tid; q < q_end; q += kDistThreads) {
                const uint4 v = c4[q];
                if ((v.x | v.y | v.z | v.w) == 0) continue;
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                uint32_t n = 0;
#pragma unroll
                for (int wi = 0; wi < 4; wi++) {
                    if (U16) n += ((w[wi] & 0xFFFFu) != 0) + ((w[wi] >> 16) != 0);
                    else n += w[wi] != 0;
                }
                uint32_t at = atomicAdd(&s_total, n);
                const uint32_t jq = col0 + q * 4 * kPerWord;
#pragma unroll
                for (int wi = 0; wi < 4; wi++) {
#pragma unroll
                    for (uint32_t h = 0; h < kPerWord; h++) {
                        const uint32_t common = U16 ? (w[wi] >> (16 * h)) & 0xFFFFu : w[wi];
                        if (common) {
                            if (at < kCandCap) cand[at] = make_uint2(jq + wi * kPerWord + h, common);
                            at++;
                        }
                    }
                }
            }
            PROF_MARK(2);
            __syncthreads();
            PROF_MARK(3);
            const uint32_t n_cells = s_total;
            if (n_cells <= kCandCap) {
                for (uint32_t i = tid; i < n_cells; i += kDistThreads) {
                    const uint2 cj = cand[i];
                    rk_hit hrec;
                    if (cj.x >= jbeg && cj.x < col1 && evaluate(cj.x, (int)cj.y, hrec)) stage_hit(hrec);
                }
            } else {
                // more sharing columns than the list holds: walk the row, one column per lane
                for (uint32_t j = jbeg + tid; j < col1; j += kDistThreads) {
                    const uint32_t common = cell(j - col0);
                    rk_hit hrec;
                    if (common && evaluate(j, (int)common, hrec)) stage_hit(hrec);
                }
                __syncthreads();  // the row is zeroed next
            }
            PROF_MARK(4);
        } else {
            // Dense mode: every cell of [jbeg, col1) can be reported.  One column per lane;
            // pass 0 counts this row's reports, one atomic reserves their slots, pass 1
            // re-evaluates and writes them.
            for (int pass_no = 0; pass_no < 2; pass_no++) {
                uint32_t mine = 0;
                for (uint32_t j = jbeg + tid; j < col1; j += kDistThreads) {
                    const int common = (int)cell(j - col0);
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
            __syncthreads();  // the row is zeroed next
        }
    };

    // ---- rows of this run ---------------------------------------------------------------------
    uint32_t row = 0;
    uint32_t slot = next_live(slot0, row);
    if (slot >= slot_end) return;  // nothing to do in this tile (uniform)
    uint64_t e0 = a.range_off[row], e1 = a.range_off[row + 1];
    uint2 pre = load_slice(e0 + tid, e1);  // the slices of the next batch are always in flight
    uint32_t parity = 0;

    while (slot < slot_end) {
        // look ahead: the next row's slice range is fetched while this row is processed
        uint32_t nrow = 0;
        const uint32_t nslot = next_live(slot + 1, nrow);
        uint64_t ne0 = 0, ne1 = 0;
        if (nslot < slot_end) { ne0 = a.range_off[nrow]; ne1 = a.range_off[nrow + 1]; }

        uint4 *z4 = reinterpret_cast<uint4 *>(cnt);  // memset row (src/dist.cpp:179)
        for (uint32_t i = tid; i < a.cnt_words / 4; i += kDistThreads) z4[i] = make_uint4(0, 0, 0, 0);
        if (tid == 0) s_cells[parity] = 0;
        __syncthreads();
        PROF_MARK(6);

        const uint32_t lo_id = tri_filter ? row + 1 : 0;  // ids below are not needed (j > i)
        auto bump = [&](uint32_t id, bool valid) {        // scatter, src/dist.cpp:199-202
            const uint32_t c = id - col0;
            if (valid && (!FILTER || (c < ncol && id >= lo_id))) {
                if (U16) atomicAdd(&cnt[c >> 1], (c & 1) ? 0x10000u : 1u);
                else atomicAdd(&cnt[c], 1u);
            }
        };
        auto stream_list = [&](uint32_t sx, uint32_t sy) {  // whole wave, 2 x 64 postings in flight
            for (uint32_t k = sx + lane; k < sy; k += 128) {
                const bool ok1 = k + 64 < sy;
                const uint32_t i0 = a.postings[k], i1 = a.postings[ok1 ? k + 64 : 0];
                bump(i0, true);
                bump(i1, ok1);
            }
        };

        // A batch = up to 256 slices of the row, one per thread, loaded straight from HBM into
        // a register pair (no LDS staging, no barrier).
        const uint32_t nb = max(1u, (uint32_t)((e1 - e0 + kDistThreads - 1) / kDistThreads));
        for (uint32_t b = 0; b < nb; b++) {
            const uint2 rg = pre;
            if (b + 1 < nb) pre = load_slice(e0 + (uint64_t)(b + 1) * kDistThreads + tid, e1);
            else if (nslot < slot_end) pre = load_slice(ne0 + tid, ne1);
            else pre = make_uint2(0, 0);
            PROF_MARK(0);

            Gathered g;
            gather(rg, g);
            // Lists longer than 8 (2.5 % of the slices at 10,000 genomes, 1.6 per wave and
            // batch): postings 8..23 of up to four of them are requested right away, one list
            // per 16-lane row, so their latency overlaps the quad gathers instead of adding a
            // dependent round trip each; whatever is longer still is streamed by the whole wave.
            unsigned long long longs = __ballot(rg.y - rg.x > (uint32_t)kGroup);
            uint32_t lx[4] = {0, 0, 0, 0}, ly[4] = {0, 0, 0, 0};  // uniform
            uint32_t qx = 0, qy = 0;                              // this lane's row
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (longs) {  // uniform
                    const int L = __ffsll((long long)longs) - 1;
                    longs &= longs - 1;
                    lx[t] = __builtin_amdgcn_readlane(rg.x, L) + kGroup;
                    ly[t] = __builtin_amdgcn_readlane(rg.y, L);
                    if ((lane >> 4) == (uint32_t)t) { qx = lx[t]; qy = ly[t]; }
                }
            }
            const uint32_t kq = qx + (lane & 15);
            const bool lokq = kq < qy;
            const uint32_t lidq = a.postings[lokq ? kq : 0];
            PROF_MARK(7);
            PROF_FENCE();
            PROF_MARK(8);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                bump(g.id[j].x, g.ok0[j]);
                bump(g.id[j].y, g.ok1[j]);
            }
            PROF_FENCE();
            PROF_MARK(9);
            bump(lidq, lokq);
#pragma unroll
            for (int t = 0; t < 4; t++)
                if (lx[t] + 16 < ly[t]) stream_list(lx[t] + 16, ly[t]);  // uniform
            while (longs) {
                const int L = __ffsll((long long)longs) - 1;
                longs &= longs - 1;
                stream_list(__builtin_amdgcn_readlane(rg.x, L) + kGroup, __builtin_amdgcn_readlane(rg.y, L));
            }
            PROF_FENCE();
            PROF_MARK(10);
        }
        __syncthreads();  // all scatters of the row done
        PROF_MARK(1);
        epilogue(row, s_cells[parity]);
        parity ^= 1;
        slot = nslot;
        row = nrow;
        e0 = ne0;
        e1 = ne1;
    }

    // flush the staged hits of this workgroup: one device-scope atomic, coalesced 8-byte stores
    __syncthreads();
    const uint32_t n_st = min(s_cursor, kStageHits);
    PROF_MARK(11);
    PROF_FLUSH();
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
    uint32_t cand_cap, stage_hits, rows_per_wg, threads;
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
    // workgroup size by row size: 7 x 256 threads fit up to ~22 KiB rows; bigger rows leave room for
    // fewer workgroups, which then need more waves each
    p->threads = p->lds_bytes <= 24 * 1024 ? 256 : (p->lds_bytes <= 64 * 1024 ? 512 : 1024);
    const uint32_t forced = envu("RK_DIST_THREADS", 0);
    if (forced == 256 || forced == 512 || forced == 1024) p->threads = forced;
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
    void (*kern)(DistArgs) = nullptr;
#define RK_PICK(T)                                                                              \
    (filter ? (p.u16 ? rk_dist_kernel<true, true, T> : rk_dist_kernel<false, true, T>)          \
            : (p.u16 ? rk_dist_kernel<true, false, T> : rk_dist_kernel<false, false, T>))
    if (p.threads == 256) kern = RK_PICK(256);
    else if (p.threads == 512) kern = RK_PICK(512);
    else kern = RK_PICK(1024);
#undef RK_PICK
    if (p.lds_bytes > 48 * 1024)
        RK_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)p.lds_bytes));
    const uint32_t runs = (p.n_rows + a.rows_per_wg - 1) / a.rows_per_wg;
    const uint32_t per = 8 * a.runs_per_chunk;  // grid padded to whole XCD chunks
    const uint32_t gx = (runs + per - 1) / per * per;
    hipLaunchKernelGGL(kern, dim3(gx, p.n_tiles), dim3(p.threads), p.lds_bytes, stream, a);
    RK_HIP(ctx, hipGetLastError());
    return RK_OK;
}

}  // namespace

#ifdef RK_DIST_PROFILE
extern "C" int rk_debug_dist_prof_raw(unsigned long long *out, unsigned long long n_waves)
{
    if (n_waves > (unsigned long long)kProfWaves) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), n_waves * 16 * 8) == hipSuccess ? 0 : -1;
}
extern "C" int rk_debug_dist_prof(unsigned long long *out16, int reset)
{
    std::vector<unsigned long long> all((size_t)kProfWaves * 16);
    if (hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(g_prof), sizeof(g_prof)) != hipSuccess) return -1;
    for (int i = 0; i < 16; i++) out16[i] = 0;
    for (size_t w = 0; w < (size_t)kProfWaves; w++)
        for (int i = 0; i < 16; i++) out16[i] += all[w * 16 + i];
    if (reset) {
        std::fill(all.begin(), all.end(), 0ULL);
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), all.data(), sizeof(g_prof)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

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
