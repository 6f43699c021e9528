// rk_dist.hip -- intersection counting through the inverted index + distance epilogue.
// Replaces the row loops of index_tridist (src/dist.cpp:174-258) and index_dist
// (src/dist.cpp:560-692).
//
// One workgroup per (query row, reference tile).  The counter row of the reference
// (`int intersectionArr[tid][numRef]`, src/dist.cpp:167) lives in LDS; the postings of the
// row's hashes are streamed from HBM by 8-lane groups (a posting list averages a few
// entries, so 8 lanes x 4 B covers most lists in one 32-byte request) and scattered into
// the LDS row with ds_add_u32.  The epilogue re-reads the LDS row, evaluates the
// Jaccard/Mash (or containment/AafD) formula in FP64 only where a pair can pass the
// threshold, and appends hit records with one wave-aggregated atomic per wave.
// Integer/index work: HBM/LDS-bound, no MFMA.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "rk_internal.h"

namespace {

constexpr int kDistThreads = 256;
constexpr int kRangeChunk = 512;   // posting ranges staged in LDS per pass (4 KiB)
constexpr int kUnroll = 4;         // independent posting gathers in flight per lane
constexpr int kRowsPerXcdChunk = 16;

struct DistArgs {
    const uint2 *ranges;      // per query element: [x,y) slice of postings
    const uint64_t *q_off;    // u64[n_query+1]
    const uint32_t *postings;
    const uint32_t *ref_sizes;
    uint32_t n_query, n_ref;
    uint32_t row_first, row_step, n_rows;
    uint32_t tile_cols, cnt_words;
    int triangle, metric, kmer_size, dense_mode;
    double max_dist;
    rk_hit *hits;
    unsigned long long cap;
    unsigned long long *n_hits;
    int32_t *common_dense;    // optional [n_query, n_ref]
};

// D3/D4: src/dist.cpp:218-231 and :238-250, FP64, same operation order.
__device__ inline void rk_distance(int common, int size0, int size1, int metric, int kmer_size,
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

// U16: two 16-bit counters per LDS word.  Valid when no count can reach 65536, i.e. the
// largest query sketch has < 65536 hashes (a count never exceeds |S_q|); halves the LDS row
// and doubles the resident rows per CU.
template <bool U16>
__global__ __launch_bounds__(kDistThreads) void rk_dist_kernel(DistArgs a)
{
    extern __shared__ uint32_t lds[];
    uint32_t *cnt = lds;
    uint2 *srange = reinterpret_cast<uint2 *>(lds + a.cnt_words);
    const uint32_t tid = threadIdx.x;

    // blockIdx.x -> row slot.  Workgroups are dealt round-robin over the 8 XCDs, so slots
    // b, b+8, ... share an L2: give each XCD runs of 16 consecutive rows (strains of one
    // clade share their posting lists) while keeping heavy (early) rows spread over all XCDs.
    const uint32_t xcd = blockIdx.x & 7, s8 = blockIdx.x >> 3;
    const uint32_t slot = ((s8 / kRowsPerXcdChunk) * 8 + xcd) * kRowsPerXcdChunk + s8 % kRowsPerXcdChunk;
    if (slot >= a.n_rows) return;
    const uint32_t row = a.row_first + slot * a.row_step;
    const uint32_t col0 = blockIdx.y * a.tile_cols;
    const uint32_t col1 = min(a.n_ref, col0 + a.tile_cols);
    const uint32_t ncol = col1 - col0;
    // tile entirely at or below the diagonal: nothing to report (uniform exit)
    if (a.triangle && col1 <= row + 1 && !a.common_dense) return;

    for (uint32_t i = tid; i < a.cnt_words; i += kDistThreads) cnt[i] = 0;  // memset row, :179

    const uint64_t e0 = a.q_off[row], e1 = a.q_off[row + 1];
    const uint32_t grp = tid >> 3, sub = tid & 7;
    const bool tri_filter = a.triangle && !a.common_dense;
    const uint32_t lo_id = tri_filter ? row + 1 : 0;  // ids below are not needed (j > i)

    auto bump = [&](uint32_t id) {
        const uint32_t c = id - col0;
        if (c < ncol && id >= lo_id) {
            if (U16) atomicAdd(&cnt[c >> 1], (c & 1) ? 0x10000u : 1u);
            else atomicAdd(&cnt[c], 1u);
        }
    };

    for (uint64_t cb = e0; cb < e1; cb += kRangeChunk) {  // :194-203
        const uint32_t n = (uint32_t)min<uint64_t>(kRangeChunk, e1 - cb);
        __syncthreads();  // previous chunk fully consumed (and the row zeroed)
        for (uint32_t i = tid; i < n; i += kDistThreads) srange[i] = a.ranges[cb + i];
        __syncthreads();
        for (uint32_t base = 0; base < n; base += (kDistThreads / 8) * kUnroll) {
            uint2 rg[kUnroll];
            uint32_t id[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                const uint32_t i = base + grp + (kDistThreads / 8) * u;
                rg[u] = i < n ? srange[i] : make_uint2(0, 0);
            }
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {  // first 8 postings of 4 lists: 4 gathers in flight
                const uint32_t k = rg[u].x + sub;
                id[u] = k < rg[u].y ? a.postings[k] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int u = 0; u < kUnroll; u++)
                if (id[u] != 0xFFFFFFFFu) bump(id[u]);
#pragma unroll
            for (int u = 0; u < kUnroll; u++)  // tails of lists longer than 8
                for (uint32_t k = rg[u].x + sub + 8; k < rg[u].y; k += 8) bump(a.postings[k]);
        }
    }
    __syncthreads();

    auto count_at = [&](uint32_t c) -> uint32_t {
        return U16 ? (cnt[c >> 1] >> ((c & 1) * 16)) & 0xFFFFu : cnt[c];
    };

    if (a.common_dense) {
        int32_t *dst = a.common_dense + (size_t)row * a.n_ref + col0;
        for (uint32_t i = tid; i < ncol; i += kDistThreads) dst[i] = (int32_t)count_at(i);
    }

    const int qsize = (int)(e1 - e0);
    const uint32_t jbeg = a.triangle ? max(col0, row + 1) : col0;  // :207 / :600
    const uint32_t lane = tid & 63;
    for (uint32_t base = jbeg - (jbeg % kDistThreads); base < col1; base += kDistThreads) {
        const uint32_t j = base + tid;
        bool pass = false;
        int common = 0, size0 = 0, size1 = 0;
        double jorc = 0.0, dist = 1.0;
        if (j >= jbeg && j < col1) {
            common = (int)count_at(j - col0);
            // common == 0 gives dist == 1.0 exactly in both metrics; in sparse mode the
            // threshold excludes 1.0, so the pair cannot be reported
            if (a.dense_mode || common) {
                const int rs = (int)a.ref_sizes[j];
                size0 = a.triangle ? qsize : rs;  // :215-216 / :607-608
                size1 = a.triangle ? rs : qsize;
                rk_distance(common, size0, size1, a.metric, a.kmer_size, jorc, dist);
                pass = a.triangle ? (dist < a.max_dist) : (dist <= a.max_dist);  // :232 / :624
            }
        }
        const unsigned long long m = __ballot(pass);
        if (m) {
            unsigned long long slot_h = 0;
            const int leader = __ffsll((long long)m) - 1;
            if ((int)lane == leader) slot_h = atomicAdd(a.n_hits, (unsigned long long)__popcll(m));
            slot_h = __shfl(slot_h, leader);
            if (pass) {
                slot_h += __popcll(m & ((1ULL << lane) - 1));
                if (slot_h < a.cap) {
                    rk_hit h;
                    h.row = row;
                    h.col = j;
                    h.common = common;
                    h.size0 = size0;
                    h.size1 = size1;
                    h.pad_ = 0;
                    h.jorc = jorc;
                    h.dist = dist;
                    a.hits[slot_h] = h;
                }
            }
        }
    }
}

struct Plan {
    uint32_t n_rows, tile_cols, n_tiles, cnt_words;
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
    const size_t lds_cap = (ctx->max_lds > 160 * 1024 ? 160 * 1024 : ctx->max_lds) - kRangeChunk * sizeof(uint2);
    const uint32_t max_cols = (uint32_t)(p->u16 ? lds_cap / 2 : lds_cap / 4) & ~63u;
    uint32_t tile = idx->n_ref ? idx->n_ref : 1;
    if (tile > max_cols) {
        const uint32_t nt = (idx->n_ref + max_cols - 1) / max_cols;
        tile = ((idx->n_ref + nt - 1) / nt + 63) & ~63u;
    }
    p->tile_cols = tile;
    p->n_tiles = idx->n_ref ? (idx->n_ref + tile - 1) / tile : 1;
    p->cnt_words = ((p->u16 ? (tile + 1) / 2 : tile) + 1) & ~1u;  // keeps srange 8-byte aligned
    p->lds_bytes = (size_t)p->cnt_words * 4 + kRangeChunk * sizeof(uint2);
    // does a pair with distance exactly 1.0 (common == 0) pass the threshold?
    p->dense_mode = o->triangle ? (1.0 < o->max_dist) : (1.0 <= o->max_dist);
    return RK_OK;
}

int launch_dist(rk_ctx *ctx, const rk_index *idx, const uint2 *ranges, const uint64_t *q_off,
                uint32_t n_query, const rk_dist_opts *o, const Plan &p, rk_hit *hits_dev,
                uint64_t cap, unsigned long long *n_hits_dev, int32_t *dense_dev, hipStream_t stream)
{
    if (!p.n_rows || !idx->n_ref) return RK_OK;
    DistArgs a;
    a.ranges = ranges;
    a.q_off = q_off;
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
    a.hits = hits_dev;
    a.cap = cap;
    a.n_hits = n_hits_dev;
    a.common_dense = dense_dev;
    void (*kern)(DistArgs) = p.u16 ? rk_dist_kernel<true> : rk_dist_kernel<false>;
    if (p.lds_bytes > 48 * 1024)
        RK_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)p.lds_bytes));
    const uint32_t per = 8 * kRowsPerXcdChunk;  // grid padded to whole XCD chunks
    const uint32_t gx = (p.n_rows + per - 1) / per * per;
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
    return launch_dist(ctx, idx, idx->d_selfrange, idx->d_src_off, idx->n_ref, opts, p, hits_dev,
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
    const uint64_t *q_off = idx->d_src_off;
    DevBuf<uint2> resolved;
    if (!self || common_dense) {
        const rk_sketches *qs = queries;
        DevBuf<uint32_t> dummy;
        const uint32_t *qh = qs ? qs->d_hashes : nullptr;
        uint64_t qn = qs ? qs->total : idx->H;
        if (!qs)
            return rk_fail(ctx, RK_ERR_ARG, "common_dense needs explicit query sketches");
        RK_HIP(ctx, resolved.alloc(qn));
        rc = rk_resolve_ranges(ctx, idx, qh, qn, resolved.p, 0);
        if (rc) return rc;
        ranges = resolved.p;
        q_off = qs->d_off;
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
        rc = launch_dist(ctx, idx, ranges, q_off, n_query, opts, p, hits.p, cap, counter.p,
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
