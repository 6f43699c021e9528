// rk_index.hip -- device-side inverted index (replaces transSketches, src/sketch.cpp:970-1017,
// and the .index load + prefix sum of src/dist.cpp:86-129).
//
// HBM layout of an index over N genomes, H postings, U distinct hashes:
//   postings  u32[H]    genome ids ordered (hash asc, genome asc)      == the .dict payload
//   uhash     u32[U]    sorted distinct hashes                         (compact, not 2^bits)
//   upos      u32[U+1]  posting offsets of each distinct hash
//   dir       u32[2^d+1] prefix directory: uhash index of the first hash >= b << shift
//   sizes     u32[N]    sketch sizes
//   selfrange uint2[H]  for source element e (genome g, hash h): the slice of h's posting
//                       list holding genomes > g   (all-vs-all triangle needs no lookup)
// The dense 2^bits count array of the .index file is only materialised by
// rk_index_export / consumed by rk_index_import.
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstdlib>

#include "rk_internal.h"

namespace {

constexpr int kThreads = 256;
inline unsigned blocks_for(uint64_t n) { return (unsigned)((n + kThreads - 1) / kThreads); }

__global__ void k_iota(uint32_t *v, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}

__global__ void k_sizes(const uint64_t *off, uint32_t n, uint32_t *sizes)
{
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) sizes[g] = (uint32_t)(off[g + 1] - off[g]);
}

template <class K> __global__ void k_head_flags(const K *keys, uint64_t n, uint32_t *flags)
{
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) flags[k] = (k == 0 || keys[k] != keys[k - 1]) ? 1u : 0u;
}

// gidx = inclusive scan of head flags (1-based group number)
template <class K>
__global__ void k_scatter_heads(const K *keys, const uint32_t *gidx, uint64_t n, K *uhash, uint32_t *upos,
                                uint64_t U)
{
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t g = gidx[k] - 1;
    if (k == 0 || gidx[k - 1] != gidx[k]) {
        uhash[g] = keys[k];
        upos[g] = (uint32_t)k;
    }
    if (k == n - 1) upos[U] = (uint32_t)n;
}

// genome of source element e: largest g with off[g] <= e
__device__ inline uint32_t genome_of(const uint64_t *off, uint32_t n, uint64_t e)
{
    uint32_t lo = 0, hi = n;  // invariant off[lo] <= e < off[hi]
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}

// covered[e] (self join, row pairs): genome 2p+1 shares this hash with genome 2p, i.e. its
// predecessor in the posting list is its pair partner.  Walking the partner's slice then serves
// both rows (the slice of 2p starts with 2p+1 and continues with exactly the slice of 2p+1), so the
// pair kernel skips the covered slices of the odd row.
__global__ void k_postings_selfrange(const uint32_t *sorted_e, const uint32_t *gidx,
                                     const uint32_t *upos, const uint64_t *off, uint32_t n_genomes,
                                     uint64_t n, uint32_t *postings, uint2 *selfrange, uint8_t *covered)
{
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t e = sorted_e[k];
    const uint32_t me = genome_of(off, n_genomes, e);
    postings[k] = me;
    const uint32_t g = gidx[k] - 1;
    // the sort is stable and source elements are genome-major, so positions k+1..end of
    // the group hold strictly later genomes (a sketch is a set: no repeated hash inside it)
    selfrange[e] = make_uint2((uint32_t)k + 1, upos[g + 1]);
    covered[e] = (me & 1u) && k > upos[g] && genome_of(off, n_genomes, sorted_e[k - 1]) == me - 1;
}

// ---- self-join slices: drop the empty ones, put a row's covered slices behind the others ------
__global__ void k_self_flags2(const uint2 *self, const uint8_t *covered, uint64_t n, uint32_t *f_open, uint32_t *f_cov)
{
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const bool nonempty = self[e].y > self[e].x;
    f_open[e] = nonempty && !covered[e];
    f_cov[e] = nonempty && covered[e];
}

// r_open / r_cov: exclusive scans of the two flag arrays, with one extra element holding the totals
__global__ void k_self_place(const uint2 *self, const uint8_t *covered, const uint32_t *r_open, const uint32_t *r_cov,
                             const uint64_t *off, uint32_t n_genomes, uint64_t n, uint2 *out)
{
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n || self[e].y <= self[e].x) return;
    const uint32_t g = genome_of(off, n_genomes, e);
    const uint64_t e0 = off[g], e1 = off[g + 1];
    const uint64_t row0 = (uint64_t)r_open[e0] + r_cov[e0];          // slices of earlier rows
    const uint64_t n_open = r_open[e1] - r_open[e0];
    const uint64_t at = covered[e] ? row0 + n_open + (r_cov[e] - r_cov[e0]) : row0 + (r_open[e] - r_open[e0]);
    out[at] = self[e];
}

__global__ void k_self_off2(const uint64_t *off, const uint32_t *r_open, const uint32_t *r_cov, uint32_t n_genomes,
                            uint64_t *self_off, uint64_t *self_split)
{
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n_genomes) return;
    const uint64_t e0 = off[g];
    const uint64_t row0 = (uint64_t)r_open[e0] + r_cov[e0];
    self_off[g] = row0;
    if (g < n_genomes) self_split[g] = row0 + (r_open[off[g + 1]] - r_open[e0]);
}

// ---- drop the empty "later genomes" slices (26 % of the elements at 10,000 genomes) ---------
__global__ void k_self_flags(const uint2 *self, uint64_t n, uint32_t *flags)
{
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) flags[e] = self[e].y > self[e].x ? 1u : 0u;
}

// rank = exclusive scan of flags
__global__ void k_self_compact(const uint2 *self, const uint32_t *flags, const uint32_t *rank, uint64_t n,
                               uint2 *out)
{
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n && flags[e]) out[rank[e]] = self[e];
}

__global__ void k_self_off(const uint64_t *off, const uint32_t *rank, uint32_t n_genomes, uint64_t H,
                           uint64_t n_self, uint64_t *self_off)
{
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n_genomes) return;
    const uint64_t e = off[g];
    self_off[g] = e < H ? rank[e] : n_self;
}

template <class K>
__global__ void k_dir(const K *uhash, uint64_t U, int shift, uint32_t n_buckets, uint32_t *dir)
{
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_buckets) return;
    if (b == n_buckets) { dir[b] = (uint32_t)U; return; }
    const uint64_t key = (uint64_t)b << shift;
    uint64_t lo = 0, hi = U;  // first index with uhash >= key
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if ((uint64_t)uhash[mid] < key) lo = mid + 1; else hi = mid;
    }
    dir[b] = (uint32_t)lo;
}

__global__ void k_sum_sq(const uint32_t *upos, uint64_t U, unsigned long long *acc)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0;
    if (i < U) {
        unsigned long long c = upos[i + 1] - upos[i];
        v = c * c;
    }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    __shared__ unsigned long long part[kThreads / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < kThreads / 64; w++) t += part[w];
        if (t) atomicAdd(acc, t);
    }
}

// ---- import/export of the dense .index array --------------------------------------
__global__ void k_nonzero_flags(const uint32_t *counts, uint64_t n, uint32_t *flags)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = counts[i] ? 1u : 0u;
}

// rank = exclusive scan of flags, cpos = exclusive scan of counts
__global__ void k_compact_dense(const uint32_t *counts, const uint32_t *rank, const uint32_t *cpos,
                                uint64_t n, uint32_t *uhash, uint32_t *upos)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && counts[i]) {
        uhash[rank[i]] = (uint32_t)i;
        upos[rank[i]] = cpos[i];
    }
}

__global__ void k_scatter_counts(const uint32_t *uhash, const uint32_t *upos, uint64_t U,
                                 uint32_t *counts)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < U) counts[uhash[i]] = upos[i + 1] - upos[i];
}

template <class K>
__global__ void k_resolve(const K *q, uint64_t n, const K *uhash, const uint32_t *upos, const uint32_t *dir,
                          int dir_shift, int hash_bits, uint2 *ranges)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const K h = q[i];
    uint2 r = make_uint2(0, 0);
    if (hash_bits >= (int)(8 * sizeof(K)) || (h >> hash_bits) == 0) {
        const uint32_t b = (uint32_t)(h >> dir_shift);
        uint32_t lo = dir[b], hi = dir[b + 1];
        while (lo < hi) {
            uint32_t mid = (lo + hi) >> 1;
            if (uhash[mid] < h) lo = mid + 1; else hi = mid;
        }
        if (lo < dir[b + 1] && uhash[lo] == h) r = make_uint2(upos[lo], upos[lo + 1]);
    }
    ranges[i] = r;
}

int finish_index(rk_ctx *ctx, rk_index *idx)
{
    // prefix directory + sum of squares; idx->d_uhash/d_upos/U/H/hash_bits are set
    int dbits = 10;
    while (dbits < 24 && (1ULL << dbits) < 2 * idx->U) dbits++;
    if (dbits > idx->hash_bits) dbits = idx->hash_bits;
    idx->dir_bits = dbits;
    idx->dir_shift = idx->hash_bits - dbits;
    const uint32_t nb = 1u << dbits;
    DevBuf<uint32_t> dir;
    RK_HIP(ctx, dir.alloc((size_t)nb + 1));
    if (idx->wide)
        hipLaunchKernelGGL(k_dir<uint64_t>, dim3(blocks_for((uint64_t)nb + 1)), dim3(kThreads), 0, 0,
                           idx->d_uhash64, idx->U, idx->dir_shift, nb, dir.p);
    else
        hipLaunchKernelGGL(k_dir<uint32_t>, dim3(blocks_for((uint64_t)nb + 1)), dim3(kThreads), 0, 0,
                           idx->d_uhash, idx->U, idx->dir_shift, nb, dir.p);
    DevBuf<unsigned long long> acc;
    RK_HIP(ctx, acc.alloc(1));
    RK_HIP(ctx, hipMemset(acc.p, 0, 8));
    if (idx->U)
        hipLaunchKernelGGL(k_sum_sq, dim3(blocks_for(idx->U)), dim3(kThreads), 0, 0, idx->d_upos,
                           idx->U, acc.p);
    unsigned long long ss = 0;
    RK_HIP(ctx, hipMemcpy(&ss, acc.p, 8, hipMemcpyDeviceToHost));
    RK_HIP(ctx, hipGetLastError());
    idx->sum_sq = ss;
    idx->d_dir = dir.release();
    return RK_OK;
}

}  // namespace

int rk_compact_ranges(rk_ctx *ctx, const uint2 *ranges_dev, uint64_t n, const uint64_t *off_dev, uint32_t n_rows,
                      uint2 **out_ranges_dev, uint64_t **out_off_dev, uint64_t *n_out, hipStream_t stream)
{
    *out_ranges_dev = nullptr;
    *out_off_dev = nullptr;
    *n_out = 0;
    DevBuf<uint64_t> new_off;
    RK_HIP(ctx, new_off.alloc((size_t)n_rows + 1));
    DevBuf<uint2> compact;
    if (n) {
        DevBuf<uint32_t> flags, rank;
        RK_HIP(ctx, flags.alloc(n));
        RK_HIP(ctx, rank.alloc(n));
        hipLaunchKernelGGL(k_self_flags, dim3(blocks_for(n)), dim3(kThreads), 0, stream, ranges_dev, n, flags.p);
        size_t tb = 0;
        RK_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, flags.p, rank.p, 0u, n, rocprim::plus<uint32_t>(), stream));
        DevBuf<char> tmp;
        RK_HIP(ctx, tmp.alloc(tb));
        RK_HIP(ctx, rocprim::exclusive_scan(tmp.p, tb, flags.p, rank.p, 0u, n, rocprim::plus<uint32_t>(), stream));
        uint32_t last_rank = 0, last_flag = 0;
        RK_HIP(ctx, hipMemcpyAsync(&last_rank, rank.p + (n - 1), 4, hipMemcpyDeviceToHost, stream));
        RK_HIP(ctx, hipMemcpyAsync(&last_flag, flags.p + (n - 1), 4, hipMemcpyDeviceToHost, stream));
        RK_HIP(ctx, hipStreamSynchronize(stream));
        const uint64_t m = (uint64_t)last_rank + last_flag;
        RK_HIP(ctx, compact.alloc(m));
        hipLaunchKernelGGL(k_self_compact, dim3(blocks_for(n)), dim3(kThreads), 0, stream, ranges_dev, flags.p, rank.p,
                           n, compact.p);
        hipLaunchKernelGGL(k_self_off, dim3(blocks_for((uint64_t)n_rows + 1)), dim3(kThreads), 0, stream, off_dev,
                           rank.p, n_rows, n, m, new_off.p);
        RK_HIP(ctx, hipGetLastError());
        RK_HIP(ctx, hipStreamSynchronize(stream));  // flags/rank die with this scope
        *n_out = m;
    } else {
        RK_HIP(ctx, compact.alloc(0));
        RK_HIP(ctx, hipMemsetAsync(new_off.p, 0, ((size_t)n_rows + 1) * 8, stream));
    }
    *out_ranges_dev = compact.release();
    *out_off_dev = new_off.release();
    return RK_OK;
}

// self-join variant of rk_compact_ranges: also reorders every row (uncovered slices first) and
// returns the split points; see k_postings_selfrange
static int compact_self(rk_ctx *ctx, const uint2 *ranges_dev, const uint8_t *covered_dev, uint64_t n,
                        const uint64_t *off_dev, uint32_t n_rows, uint2 **out_ranges_dev, uint64_t **out_off_dev,
                        uint64_t **out_split_dev, uint64_t *n_out)
{
    *out_ranges_dev = nullptr;
    *out_off_dev = nullptr;
    *out_split_dev = nullptr;
    *n_out = 0;
    DevBuf<uint64_t> new_off, split;
    RK_HIP(ctx, new_off.alloc((size_t)n_rows + 1));
    RK_HIP(ctx, split.alloc((size_t)n_rows + 1));
    DevBuf<uint2> compact;
    DevBuf<uint32_t> f_open, f_cov, r_open, r_cov;
    RK_HIP(ctx, f_open.alloc(n + 1));
    RK_HIP(ctx, f_cov.alloc(n + 1));
    RK_HIP(ctx, r_open.alloc(n + 1));
    RK_HIP(ctx, r_cov.alloc(n + 1));
    RK_HIP(ctx, hipMemset(f_open.p + n, 0, 4));
    RK_HIP(ctx, hipMemset(f_cov.p + n, 0, 4));
    if (n) hipLaunchKernelGGL(k_self_flags2, dim3(blocks_for(n)), dim3(kThreads), 0, 0, ranges_dev, covered_dev, n, f_open.p, f_cov.p);
    size_t tb = 0;
    RK_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, f_open.p, r_open.p, 0u, n + 1, rocprim::plus<uint32_t>()));
    DevBuf<char> tmp;
    RK_HIP(ctx, tmp.alloc(tb));
    RK_HIP(ctx, rocprim::exclusive_scan(tmp.p, tb, f_open.p, r_open.p, 0u, n + 1, rocprim::plus<uint32_t>()));
    RK_HIP(ctx, rocprim::exclusive_scan(tmp.p, tb, f_cov.p, r_cov.p, 0u, n + 1, rocprim::plus<uint32_t>()));
    uint32_t t_open = 0, t_cov = 0;
    RK_HIP(ctx, hipMemcpy(&t_open, r_open.p + n, 4, hipMemcpyDeviceToHost));
    RK_HIP(ctx, hipMemcpy(&t_cov, r_cov.p + n, 4, hipMemcpyDeviceToHost));
    const uint64_t m = (uint64_t)t_open + t_cov;
    RK_HIP(ctx, compact.alloc(m));
    if (n) hipLaunchKernelGGL(k_self_place, dim3(blocks_for(n)), dim3(kThreads), 0, 0, ranges_dev, covered_dev, r_open.p,
                              r_cov.p, off_dev, n_rows, n, compact.p);
    hipLaunchKernelGGL(k_self_off2, dim3(blocks_for((uint64_t)n_rows + 1)), dim3(kThreads), 0, 0, off_dev, r_open.p,
                       r_cov.p, n_rows, new_off.p, split.p);
    RK_HIP(ctx, hipGetLastError());
    RK_HIP(ctx, hipDeviceSynchronize());  // the temporaries die with this scope
    *n_out = m;
    *out_ranges_dev = compact.release();
    *out_off_dev = new_off.release();
    *out_split_dev = split.release();
    return RK_OK;
}

int rk_resolve_ranges(rk_ctx *ctx, const rk_index *idx, const void *q_hashes_dev, uint64_t n,
                      uint2 *ranges_dev, hipStream_t stream)
{
    if (!n) return RK_OK;
    if (idx->wide)
        hipLaunchKernelGGL(k_resolve<uint64_t>, dim3(blocks_for(n)), dim3(kThreads), 0, stream,
                           (const uint64_t *)q_hashes_dev, n, idx->d_uhash64, idx->d_upos, idx->d_dir,
                           idx->dir_shift, idx->hash_bits, ranges_dev);
    else
        hipLaunchKernelGGL(k_resolve<uint32_t>, dim3(blocks_for(n)), dim3(kThreads), 0, stream,
                           (const uint32_t *)q_hashes_dev, n, idx->d_uhash, idx->d_upos, idx->d_dir,
                           idx->dir_shift, idx->hash_bits, ranges_dev);
    RK_HIP(ctx, hipGetLastError());
    return RK_OK;
}

extern "C" {

void rk_index_free(rk_index *idx)
{
    if (!idx) return;
    (void)hipFree(idx->d_postings);
    (void)hipFree(idx->d_uhash);
    (void)hipFree(idx->d_uhash64);
    (void)hipFree(idx->d_upos);
    (void)hipFree(idx->d_dir);
    (void)hipFree(idx->d_sizes);
    (void)hipFree(idx->d_selfrange);
    (void)hipFree(idx->d_self_off);
    (void)hipFree(idx->d_self_split);
    (void)hipFree(idx->d_src_off);
    delete idx;
}

uint64_t rk_index_total(const rk_index *idx) { return idx ? idx->H : 0; }
uint64_t rk_index_distinct(const rk_index *idx) { return idx ? idx->U : 0; }
uint32_t rk_index_genomes(const rk_index *idx) { return idx ? idx->n_ref : 0; }
int rk_index_hash_bits(const rk_index *idx) { return idx ? idx->hash_bits : 0; }
uint64_t rk_index_sum_sq(const rk_index *idx) { return idx ? idx->sum_sq : 0; }

int rk_index_build(rk_ctx *ctx, const rk_sketches *s, int hash_bits, rk_index **out)
{
    if (!ctx || !s || !out) return RK_ERR_ARG;
    *out = nullptr;
    if (hash_bits < 1) return rk_fail(ctx, RK_ERR_ARG, "hash_bits must be positive");
    if (hash_bits > 64) return rk_fail(ctx, RK_ERR_ARG, "hash_bits=%d", hash_bits);
    if ((hash_bits > 32) != s->wide)
        return rk_fail(ctx, RK_ERR_ARG, "hash_bits=%d does not match the sketches' %s-bit layout", hash_bits,
                       s->wide ? "64" : "32");
    const uint64_t H = s->total;
    if (H >= 0xFFFFFFFFULL) return rk_fail(ctx, RK_ERR_UNSUPPORTED, "more than 2^32-1 postings");
    RK_HIP(ctx, hipSetDevice(ctx->device));
    rk_index *idx = new (std::nothrow) rk_index;
    if (!idx) return RK_ERR_NOMEM;
    idx->ctx = ctx;
    idx->n_ref = s->n;
    idx->H = H;
    idx->hash_bits = hash_bits;
    idx->wide = s->wide;
    for (uint32_t g = 0; g < s->n; g++)
        idx->max_src_size = std::max<uint64_t>(idx->max_src_size, s->h_off[g + 1] - s->h_off[g]);
    struct Guard { rk_index *p; ~Guard() { if (p) rk_index_free(p); } } guard{idx};

    RK_HIP(ctx, hipMalloc((void **)&idx->d_sizes, ((size_t)s->n + 1) * 4));
    RK_HIP(ctx, hipMalloc((void **)&idx->d_src_off, ((size_t)s->n + 1) * 8));
    RK_HIP(ctx, hipMemcpy(idx->d_src_off, s->d_off, ((size_t)s->n + 1) * 8, hipMemcpyDeviceToDevice));
    if (s->n)
        hipLaunchKernelGGL(k_sizes, dim3(blocks_for(s->n)), dim3(kThreads), 0, 0, s->d_off, s->n,
                           idx->d_sizes);
    RK_HIP(ctx, hipMalloc((void **)&idx->d_postings, (H + 4) * 4));
    RK_HIP(ctx, hipMalloc((void **)&idx->d_selfrange, (H + 1) * sizeof(uint2)));

    DevBuf<uint32_t> iota, keys_sorted, sorted_e, flags;
    DevBuf<uint64_t> keys_sorted64;
    DevBuf<uint8_t> covered;
    RK_HIP(ctx, covered.alloc(H + 1));
    RK_HIP(ctx, iota.alloc(H));
    RK_HIP(ctx, keys_sorted.alloc(H));
    if (idx->wide) RK_HIP(ctx, keys_sorted64.alloc(H));
    RK_HIP(ctx, sorted_e.alloc(H));
    RK_HIP(ctx, flags.alloc(H));
    if (H) {
        hipLaunchKernelGGL(k_iota, dim3(blocks_for(H)), dim3(kThreads), 0, 0, iota.p, H);
        // stable LSD radix sort by hash; values = source element index (genome-major), so
        // equal hashes stay in ascending genome order == hashMapId[hash].push_back(i) for
        // i ascending (src/sketch.cpp:979-985)
        size_t tmp_bytes = 0;
        DevBuf<char> tmp;
        if (idx->wide) {
            RK_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, s->d_hashes64, keys_sorted64.p, iota.p,
                                                  sorted_e.p, H, 0, (unsigned)hash_bits));
            RK_HIP(ctx, tmp.alloc(tmp_bytes));
            RK_HIP(ctx, rocprim::radix_sort_pairs(tmp.p, tmp_bytes, s->d_hashes64, keys_sorted64.p, iota.p,
                                                  sorted_e.p, H, 0, (unsigned)hash_bits));
            hipLaunchKernelGGL(k_head_flags<uint64_t>, dim3(blocks_for(H)), dim3(kThreads), 0, 0, keys_sorted64.p,
                               H, flags.p);
        } else {
            RK_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, s->d_hashes, keys_sorted.p, iota.p,
                                                  sorted_e.p, H, 0, (unsigned)hash_bits));
            RK_HIP(ctx, tmp.alloc(tmp_bytes));
            RK_HIP(ctx, rocprim::radix_sort_pairs(tmp.p, tmp_bytes, s->d_hashes, keys_sorted.p, iota.p,
                                                  sorted_e.p, H, 0, (unsigned)hash_bits));
            hipLaunchKernelGGL(k_head_flags<uint32_t>, dim3(blocks_for(H)), dim3(kThreads), 0, 0, keys_sorted.p, H,
                               flags.p);
        }
        size_t tmp2 = 0;
        RK_HIP(ctx, rocprim::inclusive_scan(nullptr, tmp2, flags.p, iota.p, H, rocprim::plus<uint32_t>()));
        if (tmp2 > tmp_bytes) { RK_HIP(ctx, tmp.alloc(tmp2)); }
        RK_HIP(ctx, rocprim::inclusive_scan(tmp.p, tmp2, flags.p, iota.p, H, rocprim::plus<uint32_t>()));
        uint32_t U32 = 0;
        RK_HIP(ctx, hipMemcpy(&U32, iota.p + (H - 1), 4, hipMemcpyDeviceToHost));
        idx->U = U32;
    }
    uint32_t *gidx = iota.p;  // 1-based group number of each sorted position
    if (idx->wide) RK_HIP(ctx, hipMalloc((void **)&idx->d_uhash64, (idx->U + 1) * 8));
    else RK_HIP(ctx, hipMalloc((void **)&idx->d_uhash, (idx->U + 1) * 4));
    RK_HIP(ctx, hipMalloc((void **)&idx->d_upos, (idx->U + 2) * 4));
    if (H) {
        if (idx->wide)
            hipLaunchKernelGGL(k_scatter_heads<uint64_t>, dim3(blocks_for(H)), dim3(kThreads), 0, 0,
                               keys_sorted64.p, gidx, H, idx->d_uhash64, idx->d_upos, idx->U);
        else
            hipLaunchKernelGGL(k_scatter_heads<uint32_t>, dim3(blocks_for(H)), dim3(kThreads), 0, 0,
                               keys_sorted.p, gidx, H, idx->d_uhash, idx->d_upos, idx->U);
        hipLaunchKernelGGL(k_postings_selfrange, dim3(blocks_for(H)), dim3(kThreads), 0, 0, sorted_e.p,
                           gidx, idx->d_upos, s->d_off, s->n, H, idx->d_postings, idx->d_selfrange, covered.p);
    } else {
        RK_HIP(ctx, hipMemset(idx->d_upos, 0, 8));
    }
    RK_HIP(ctx, hipGetLastError());
    // compact away the empty slices (26 % of the elements at 10,000 genomes); covered slices go last in their row
    {
        uint2 *compact = nullptr;
        int rcc = compact_self(ctx, idx->d_selfrange, covered.p, H, s->d_off, s->n, &compact, &idx->d_self_off,
                               &idx->d_self_split, &idx->n_self);
        if (rcc) return rcc;
        (void)hipFree(idx->d_selfrange);
        idx->d_selfrange = compact;
    }
    int rc = finish_index(ctx, idx);
    if (rc) return rc;
    RK_HIP(ctx, hipDeviceSynchronize());
    guard.p = nullptr;
    *out = idx;
    return RK_OK;
}

int rk_index_import(rk_ctx *ctx, const uint32_t *postings, uint64_t total, const uint32_t *counts,
                    int hash_bits, const uint32_t *ref_sizes, uint32_t n_ref, rk_index **out)
{
    if (!ctx || !out || !counts || (!postings && total) || (!ref_sizes && n_ref)) return RK_ERR_ARG;
    *out = nullptr;
    if (hash_bits < 1 || hash_bits > 32)
        return rk_fail(ctx, RK_ERR_UNSUPPORTED, "hash_bits=%d outside the 32-bit layout", hash_bits);
    if (total >= 0xFFFFFFFFULL) return rk_fail(ctx, RK_ERR_UNSUPPORTED, "more than 2^32-1 postings");
    RK_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t hs = 1ULL << hash_bits;
    rk_index *idx = new (std::nothrow) rk_index;
    if (!idx) return RK_ERR_NOMEM;
    idx->ctx = ctx;
    idx->n_ref = n_ref;
    idx->H = total;
    idx->hash_bits = hash_bits;
    struct Guard { rk_index *p; ~Guard() { if (p) rk_index_free(p); } } guard{idx};

    RK_HIP(ctx, hipMalloc((void **)&idx->d_sizes, ((size_t)n_ref + 1) * 4));
    RK_HIP(ctx, hipMemcpy(idx->d_sizes, ref_sizes, (size_t)n_ref * 4, hipMemcpyHostToDevice));
    RK_HIP(ctx, hipMalloc((void **)&idx->d_postings, (total + 4) * 4));
    RK_HIP(ctx, hipMemcpy(idx->d_postings, postings, total * 4, hipMemcpyHostToDevice));

    DevBuf<uint32_t> d_counts, flags, rank, cpos;
    RK_HIP(ctx, d_counts.alloc(hs));
    RK_HIP(ctx, flags.alloc(hs));
    RK_HIP(ctx, rank.alloc(hs));
    RK_HIP(ctx, cpos.alloc(hs));
    RK_HIP(ctx, hipMemcpy(d_counts.p, counts, hs * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_nonzero_flags, dim3(blocks_for(hs)), dim3(kThreads), 0, 0, d_counts.p, hs,
                       flags.p);
    size_t tb = 0;
    RK_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, flags.p, rank.p, 0u, hs, rocprim::plus<uint32_t>()));
    DevBuf<char> tmp;
    RK_HIP(ctx, tmp.alloc(tb));
    RK_HIP(ctx, rocprim::exclusive_scan(tmp.p, tb, flags.p, rank.p, 0u, hs, rocprim::plus<uint32_t>()));
    RK_HIP(ctx, rocprim::exclusive_scan(tmp.p, tb, d_counts.p, cpos.p, 0u, hs, rocprim::plus<uint32_t>()));
    uint32_t last_rank = 0, last_flag = 0, last_cpos = 0, last_cnt = 0;
    RK_HIP(ctx, hipMemcpy(&last_rank, rank.p + (hs - 1), 4, hipMemcpyDeviceToHost));
    RK_HIP(ctx, hipMemcpy(&last_flag, flags.p + (hs - 1), 4, hipMemcpyDeviceToHost));
    RK_HIP(ctx, hipMemcpy(&last_cpos, cpos.p + (hs - 1), 4, hipMemcpyDeviceToHost));
    RK_HIP(ctx, hipMemcpy(&last_cnt, d_counts.p + (hs - 1), 4, hipMemcpyDeviceToHost));
    idx->U = (uint64_t)last_rank + last_flag;
    if ((uint64_t)last_cpos + last_cnt != total)  // src/dist.cpp:107-110
        return rk_fail(ctx, RK_ERR_ARG, "mismatched total hash number: index says %llu, dict has %llu",
                       (unsigned long long)last_cpos + last_cnt, (unsigned long long)total);
    RK_HIP(ctx, hipMalloc((void **)&idx->d_uhash, (idx->U + 1) * 4));
    RK_HIP(ctx, hipMalloc((void **)&idx->d_upos, (idx->U + 2) * 4));
    hipLaunchKernelGGL(k_compact_dense, dim3(blocks_for(hs)), dim3(kThreads), 0, 0, d_counts.p,
                       rank.p, cpos.p, hs, idx->d_uhash, idx->d_upos);
    const uint32_t tot32 = (uint32_t)total;
    RK_HIP(ctx, hipMemcpy(idx->d_upos + idx->U, &tot32, 4, hipMemcpyHostToDevice));
    RK_HIP(ctx, hipGetLastError());
    int rc = finish_index(ctx, idx);
    if (rc) return rc;
    RK_HIP(ctx, hipDeviceSynchronize());
    guard.p = nullptr;
    *out = idx;
    return RK_OK;
}

int rk_index_export(const rk_index *idx, uint32_t *postings, uint32_t *counts)
{
    if (!idx) return RK_ERR_ARG;
    rk_ctx *ctx = idx->ctx;
    if (idx->wide && counts)
        return rk_fail(ctx, RK_ERR_ARG, "64-bit index: use rk_index_export64 (sparse .index layout)");
    RK_HIP(ctx, hipSetDevice(ctx->device));
    if (postings && idx->H)
        RK_HIP(ctx, hipMemcpy(postings, idx->d_postings, idx->H * 4, hipMemcpyDeviceToHost));
    if (counts) {
        const uint64_t hs = 1ULL << idx->hash_bits;
        DevBuf<uint32_t> d_counts;
        RK_HIP(ctx, d_counts.alloc(hs));
        RK_HIP(ctx, hipMemset(d_counts.p, 0, hs * 4));
        if (idx->U)
            hipLaunchKernelGGL(k_scatter_counts, dim3(blocks_for(idx->U)), dim3(kThreads), 0, 0,
                               idx->d_uhash, idx->d_upos, idx->U, d_counts.p);
        RK_HIP(ctx, hipGetLastError());
        RK_HIP(ctx, hipMemcpy(counts, d_counts.p, hs * 4, hipMemcpyDeviceToHost));
    }
    return RK_OK;
}

}  // extern "C"

// ---- single-blob form for the RCCL broadcast ------------------------------------------
namespace {
struct BlobHeader {
    uint64_t magic, bytes;
    uint64_t H, U, sum_sq, max_src_size, n_self;
    uint32_t n_ref, has_self;
    int32_t hash_bits, dir_bits, dir_shift, wide;
    uint64_t off_postings, off_uhash, off_upos, off_dir, off_sizes, off_self, off_selfoff, off_src, off_split;
};
constexpr uint64_t kBlobMagic = 0x32584449444b5352ULL;  // "RSKDIDX2"
inline uint64_t al256(uint64_t x) { return (x + 255) & ~255ULL; }

void blob_layout(const rk_index *idx, BlobHeader *h)
{
    memset(h, 0, sizeof(*h));
    h->magic = kBlobMagic;
    h->H = idx->H;
    h->U = idx->U;
    h->sum_sq = idx->sum_sq;
    h->max_src_size = idx->max_src_size;
    h->n_self = idx->n_self;
    h->n_ref = idx->n_ref;
    h->has_self = idx->d_selfrange ? 1 : 0;
    h->hash_bits = idx->hash_bits;
    h->dir_bits = idx->dir_bits;
    h->dir_shift = idx->dir_shift;
    h->wide = idx->wide ? 1 : 0;
    uint64_t p = al256(sizeof(BlobHeader));
    h->off_postings = p; p = al256(p + (idx->H + 1) * 4);
    h->off_uhash = p;    p = al256(p + (idx->U + 1) * (idx->wide ? 8 : 4));
    h->off_upos = p;     p = al256(p + (idx->U + 2) * 4);
    h->off_dir = p;      p = al256(p + ((1ULL << idx->dir_bits) + 1) * 4);
    h->off_sizes = p;    p = al256(p + ((uint64_t)idx->n_ref + 1) * 4);
    if (h->has_self) {
        h->off_self = p;    p = al256(p + (idx->n_self + 1) * sizeof(uint2));
        h->off_selfoff = p; p = al256(p + ((uint64_t)idx->n_ref + 1) * 8);
        h->off_src = p;     p = al256(p + ((uint64_t)idx->n_ref + 1) * 8);
        h->off_split = p;   p = al256(p + ((uint64_t)idx->n_ref + 1) * 8);
    }
    h->bytes = p;
}
}  // namespace

extern "C" {

uint64_t rk_index_blob_bytes(const rk_index *idx)
{
    if (!idx) return 0;
    BlobHeader h;
    blob_layout(idx, &h);
    return h.bytes;
}

int rk_index_pack_dev(const rk_index *idx, void *blob_dev, uint64_t blob_cap, void *stream_v)
{
    if (!idx || !blob_dev) return RK_ERR_ARG;
    rk_ctx *ctx = idx->ctx;
    hipStream_t st = (hipStream_t)stream_v;
    BlobHeader h;
    blob_layout(idx, &h);
    if (blob_cap < h.bytes) return rk_fail(ctx, RK_ERR_CAPACITY, "blob needs %llu bytes", (unsigned long long)h.bytes);
    char *b = (char *)blob_dev;
    RK_HIP(ctx, hipSetDevice(ctx->device));
    RK_HIP(ctx, hipMemcpyAsync(b, &h, sizeof(h), hipMemcpyHostToDevice, st));
    RK_HIP(ctx, hipMemcpyAsync(b + h.off_postings, idx->d_postings, idx->H * 4, hipMemcpyDeviceToDevice, st));
    if (idx->wide) RK_HIP(ctx, hipMemcpyAsync(b + h.off_uhash, idx->d_uhash64, idx->U * 8, hipMemcpyDeviceToDevice, st));
    else RK_HIP(ctx, hipMemcpyAsync(b + h.off_uhash, idx->d_uhash, idx->U * 4, hipMemcpyDeviceToDevice, st));
    RK_HIP(ctx, hipMemcpyAsync(b + h.off_upos, idx->d_upos, (idx->U + 1) * 4, hipMemcpyDeviceToDevice, st));
    RK_HIP(ctx, hipMemcpyAsync(b + h.off_dir, idx->d_dir, ((1ULL << idx->dir_bits) + 1) * 4, hipMemcpyDeviceToDevice, st));
    RK_HIP(ctx, hipMemcpyAsync(b + h.off_sizes, idx->d_sizes, (uint64_t)idx->n_ref * 4, hipMemcpyDeviceToDevice, st));
    if (h.has_self) {
        RK_HIP(ctx, hipMemcpyAsync(b + h.off_self, idx->d_selfrange, idx->n_self * sizeof(uint2), hipMemcpyDeviceToDevice, st));
        RK_HIP(ctx, hipMemcpyAsync(b + h.off_selfoff, idx->d_self_off, ((uint64_t)idx->n_ref + 1) * 8, hipMemcpyDeviceToDevice, st));
        RK_HIP(ctx, hipMemcpyAsync(b + h.off_src, idx->d_src_off, ((uint64_t)idx->n_ref + 1) * 8, hipMemcpyDeviceToDevice, st));
        RK_HIP(ctx, hipMemcpyAsync(b + h.off_split, idx->d_self_split, (uint64_t)idx->n_ref * 8, hipMemcpyDeviceToDevice, st));
    }
    RK_HIP(ctx, hipStreamSynchronize(st));
    return RK_OK;
}

int rk_index_unpack_dev(rk_ctx *ctx, const void *blob_dev, uint64_t blob_bytes, void *stream_v, rk_index **out)
{
    if (!ctx || !blob_dev || !out || blob_bytes < sizeof(BlobHeader)) return RK_ERR_ARG;
    *out = nullptr;
    hipStream_t st = (hipStream_t)stream_v;
    RK_HIP(ctx, hipSetDevice(ctx->device));
    BlobHeader h;
    RK_HIP(ctx, hipMemcpyAsync(&h, blob_dev, sizeof(h), hipMemcpyDeviceToHost, st));
    RK_HIP(ctx, hipStreamSynchronize(st));
    if (h.magic != kBlobMagic || h.bytes > blob_bytes) return rk_fail(ctx, RK_ERR_ARG, "not an index blob");
    rk_index *idx = new (std::nothrow) rk_index;
    if (!idx) return RK_ERR_NOMEM;
    struct Guard { rk_index *p; ~Guard() { if (p) rk_index_free(p); } } guard{idx};
    idx->ctx = ctx;
    idx->n_ref = h.n_ref;
    idx->H = h.H;
    idx->U = h.U;
    idx->sum_sq = h.sum_sq;
    idx->max_src_size = h.max_src_size;
    idx->n_self = h.n_self;
    idx->hash_bits = h.hash_bits;
    idx->dir_bits = h.dir_bits;
    idx->dir_shift = h.dir_shift;
    idx->wide = h.wide != 0;
    BlobHeader chk;
    blob_layout(idx, &chk);  // offsets must be the ones this library would produce
    idx->d_selfrange = nullptr;
    const char *b = (const char *)blob_dev;
    const uint64_t nb = (1ULL << idx->dir_bits) + 1;
    RK_HIP(ctx, hipMalloc((void **)&idx->d_postings, (idx->H + 4) * 4));
    if (idx->wide) RK_HIP(ctx, hipMalloc((void **)&idx->d_uhash64, (idx->U + 1) * 8));
    else RK_HIP(ctx, hipMalloc((void **)&idx->d_uhash, (idx->U + 1) * 4));
    RK_HIP(ctx, hipMalloc((void **)&idx->d_upos, (idx->U + 2) * 4));
    RK_HIP(ctx, hipMalloc((void **)&idx->d_dir, nb * 4));
    RK_HIP(ctx, hipMalloc((void **)&idx->d_sizes, ((size_t)idx->n_ref + 1) * 4));
    RK_HIP(ctx, hipMemcpyAsync(idx->d_postings, b + h.off_postings, idx->H * 4, hipMemcpyDeviceToDevice, st));
    if (idx->wide) RK_HIP(ctx, hipMemcpyAsync(idx->d_uhash64, b + h.off_uhash, idx->U * 8, hipMemcpyDeviceToDevice, st));
    else RK_HIP(ctx, hipMemcpyAsync(idx->d_uhash, b + h.off_uhash, idx->U * 4, hipMemcpyDeviceToDevice, st));
    RK_HIP(ctx, hipMemcpyAsync(idx->d_upos, b + h.off_upos, (idx->U + 1) * 4, hipMemcpyDeviceToDevice, st));
    RK_HIP(ctx, hipMemcpyAsync(idx->d_dir, b + h.off_dir, nb * 4, hipMemcpyDeviceToDevice, st));
    RK_HIP(ctx, hipMemcpyAsync(idx->d_sizes, b + h.off_sizes, (uint64_t)idx->n_ref * 4, hipMemcpyDeviceToDevice, st));
    if (h.has_self) {
        RK_HIP(ctx, hipMalloc((void **)&idx->d_selfrange, (idx->n_self + 1) * sizeof(uint2)));
        RK_HIP(ctx, hipMalloc((void **)&idx->d_self_off, ((size_t)idx->n_ref + 1) * 8));
        RK_HIP(ctx, hipMalloc((void **)&idx->d_src_off, ((size_t)idx->n_ref + 1) * 8));
        RK_HIP(ctx, hipMemcpyAsync(idx->d_selfrange, b + h.off_self, idx->n_self * sizeof(uint2), hipMemcpyDeviceToDevice, st));
        RK_HIP(ctx, hipMemcpyAsync(idx->d_self_off, b + h.off_selfoff, ((uint64_t)idx->n_ref + 1) * 8, hipMemcpyDeviceToDevice, st));
        RK_HIP(ctx, hipMemcpyAsync(idx->d_src_off, b + h.off_src, ((uint64_t)idx->n_ref + 1) * 8, hipMemcpyDeviceToDevice, st));
        RK_HIP(ctx, hipMalloc((void **)&idx->d_self_split, ((size_t)idx->n_ref + 1) * 8));
        RK_HIP(ctx, hipMemcpyAsync(idx->d_self_split, b + h.off_split, (uint64_t)idx->n_ref * 8, hipMemcpyDeviceToDevice, st));
    }
    RK_HIP(ctx, hipStreamSynchronize(st));
    guard.p = nullptr;
    *out = idx;
    return RK_OK;
}

}  // extern "C"

// ---- 64-bit hash layout: sparse .index = {u64 n; u64 hash[n]; u32 count[n]} ------------------
// (src/sketch.cpp:961-963, read back at src/dist.cpp:36-82; any block order is legal)
namespace {
__global__ void k_counts_from_upos(const uint32_t *upos, uint64_t U, uint32_t *counts)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < U) counts[i] = upos[i + 1] - upos[i];
}
}  // namespace

extern "C" {

int rk_index_export64(const rk_index *idx, uint32_t *postings, uint64_t *hashes, uint32_t *counts)
{
    if (!idx) return RK_ERR_ARG;
    rk_ctx *ctx = idx->ctx;
    if (!idx->wide) return rk_fail(ctx, RK_ERR_ARG, "32-bit index: use rk_index_export (dense .index layout)");
    RK_HIP(ctx, hipSetDevice(ctx->device));
    if (postings && idx->H) RK_HIP(ctx, hipMemcpy(postings, idx->d_postings, idx->H * 4, hipMemcpyDeviceToHost));
    if (hashes && idx->U) RK_HIP(ctx, hipMemcpy(hashes, idx->d_uhash64, idx->U * 8, hipMemcpyDeviceToHost));
    if (counts && idx->U) {
        DevBuf<uint32_t> c;
        RK_HIP(ctx, c.alloc(idx->U));
        hipLaunchKernelGGL(k_counts_from_upos, dim3(blocks_for(idx->U)), dim3(kThreads), 0, 0, idx->d_upos, idx->U, c.p);
        RK_HIP(ctx, hipGetLastError());
        RK_HIP(ctx, hipMemcpy(counts, c.p, idx->U * 4, hipMemcpyDeviceToHost));
    }
    return RK_OK;
}

int rk_index_import64(rk_ctx *ctx, const uint32_t *postings, uint64_t total, const uint64_t *hashes,
                      const uint32_t *counts, uint64_t n_hash, int hash_bits, const uint32_t *ref_sizes,
                      uint32_t n_ref, rk_index **out)
{
    if (!ctx || !out || (!postings && total) || ((!hashes || !counts) && n_hash) || (!ref_sizes && n_ref))
        return RK_ERR_ARG;
    *out = nullptr;
    if (hash_bits <= 32 || hash_bits > 64) return rk_fail(ctx, RK_ERR_ARG, "hash_bits=%d is not a 64-bit layout", hash_bits);
    if (total >= 0xFFFFFFFFULL || n_hash >= 0xFFFFFFFFULL) return rk_fail(ctx, RK_ERR_UNSUPPORTED, "more than 2^32-1 postings");
    // the file may list the posting blocks in any order (the reference writes hash-map order):
    // bring them into ascending hash order on the host, then upload
    std::vector<uint64_t> start(n_hash + 1, 0);
    for (uint64_t i = 0; i < n_hash; i++) start[i + 1] = start[i] + counts[i];
    if (start[n_hash] != total)
        return rk_fail(ctx, RK_ERR_ARG, "mismatched total hash number: index says %llu, dict has %llu",
                       (unsigned long long)start[n_hash], (unsigned long long)total);
    std::vector<uint32_t> order(n_hash);
    for (uint64_t i = 0; i < n_hash; i++) order[i] = (uint32_t)i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hashes[a] < hashes[b]; });
    std::vector<uint64_t> uh(n_hash + 1);
    std::vector<uint32_t> up(n_hash + 2, 0), post(total + 1);
    uint64_t w = 0;
    for (uint64_t r = 0; r < n_hash; r++) {
        const uint32_t i = order[r];
        if (r && hashes[i] == uh[r - 1]) return rk_fail(ctx, RK_ERR_ARG, "duplicate hash in the index file");
        uh[r] = hashes[i];
        up[r] = (uint32_t)w;
        memcpy(post.data() + w, postings + start[i], (size_t)counts[i] * 4);
        w += counts[i];
    }
    up[n_hash] = (uint32_t)w;
    RK_HIP(ctx, hipSetDevice(ctx->device));
    rk_index *idx = new (std::nothrow) rk_index;
    if (!idx) return RK_ERR_NOMEM;
    struct Guard { rk_index *p; ~Guard() { if (p) rk_index_free(p); } } guard{idx};
    idx->ctx = ctx;
    idx->wide = true;
    idx->n_ref = n_ref;
    idx->H = total;
    idx->U = n_hash;
    idx->hash_bits = hash_bits;
    RK_HIP(ctx, hipMalloc((void **)&idx->d_sizes, ((size_t)n_ref + 1) * 4));
    RK_HIP(ctx, hipMalloc((void **)&idx->d_postings, (total + 4) * 4));
    RK_HIP(ctx, hipMalloc((void **)&idx->d_uhash64, (n_hash + 1) * 8));
    RK_HIP(ctx, hipMalloc((void **)&idx->d_upos, (n_hash + 2) * 4));
    RK_HIP(ctx, hipMemcpy(idx->d_sizes, ref_sizes, (size_t)n_ref * 4, hipMemcpyHostToDevice));
    RK_HIP(ctx, hipMemcpy(idx->d_postings, post.data(), total * 4, hipMemcpyHostToDevice));
    RK_HIP(ctx, hipMemcpy(idx->d_uhash64, uh.data(), n_hash * 8, hipMemcpyHostToDevice));
    RK_HIP(ctx, hipMemcpy(idx->d_upos, up.data(), (n_hash + 1) * 4, hipMemcpyHostToDevice));
    int rc = finish_index(ctx, idx);
    if (rc) return rc;
    RK_HIP(ctx, hipDeviceSynchronize());
    guard.p = nullptr;
    *out = idx;
    return RK_OK;
}

}  // extern "C"
