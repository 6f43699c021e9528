// rk_tiles.hip -- the tile records of the self join over 32 x 32 tiles (rk_dist_tile.inc), built once per index.
// A translation unit of its own: the scans, sorts and the run-length encoding come from rocprim, whose kernels make a
// code object of several MB -- the HIP runtime loads a code object when the first kernel of its translation unit is
// launched, and a self join that runs on the near-window kernel (or a query) must not pay for that.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include <rocprim/rocprim.hpp>

#include "rk_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// Tile records of an index, built on the first sparse self join (rk_tiles_build) from the postings and the list offsets:
//   k_list_starts    one byte per posting position: a list starts here
//   k_run_flags      posting p starts a "word run" (its list, or its block of 32 ids, differs from p - 1's)
//   (scan)           -> the run's number
//   k_run_emit       per run: block id, bitmask of the members, number of the list's first run, records it will write
//   (scan)           -> where its records go
//   k_contrib_emit   run j of a list pairs with every earlier run i of the same list (tile (block_i, block_j)) and, when it
//                    holds two members or more, with itself
//   radix sort by tile, run-length encode -> directory
// Element-parallel throughout: a hash shared by every genome is 10,000 postings, 313 runs, 49 k records -- no thread walks it.

// one byte per posting position: a list starts here (plain stores to distinct addresses: the first version set bits of a
// bitmap with atomicOr, 3 M atomics on 380 k words = 112 us)
__global__ void k_list_starts(const uint32_t *upos, uint64_t U, uint8_t *starts)
{
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const uint32_t p = upos[u];
    if (upos[u + 1] > p) starts[p] = 1;
}

// flags[p] = posting p starts a run (its list, or its block of 32 ids, differs from p - 1's); bm = the list starts as a
// bitmap (one ballot per 64 positions).  256 threads per workgroup, flags has H + 1 entries (the last one 0: the exclusive
// scan then ends with the number of runs).
__global__ void k_run_flags(const uint32_t *postings, const uint8_t *starts, uint64_t H, uint32_t *flags, uint32_t *bm)
{
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = p < H;
    const bool ls = in && starts[p];
    const bool run = in && (ls || (postings[p] >> 5) != (postings[p - (p ? 1 : 0)] >> 5));
    if (p <= H) flags[p] = run ? 1u : 0u;
    const unsigned long long m = __ballot(ls);
    if ((threadIdx.x & 31) == 0 && (p >> 5) <= (H >> 5)) bm[p >> 5] = (uint32_t)(m >> (threadIdx.x & 32));
}

// rank = exclusive scan of the flags (H + 1 entries): posting q starts a run iff rank[q + 1] != rank[q].  first_run: the
// list's first run (its list start's rank)
__global__ void k_run_emit(const uint32_t *postings, const uint32_t *bm, const uint32_t *rank, uint64_t H, uint32_t *run_block,
                           uint32_t *run_mask, uint32_t *run_first, uint32_t *run_records)
{
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H) return;
    const uint32_t j = rank[p];
    if (rank[p + 1] == j) return;   // not a run start
    const uint32_t blk = postings[p] >> 5;
    uint32_t mask = 1u << (postings[p] & 31);
    for (uint64_t q = p + 1; q < H && rank[q + 1] == rank[q]; q++) mask |= 1u << (postings[q] & 31);   // at most 31 more members
    // the list's start: the nearest set bit of the bitmap at or before p
    uint64_t wd = p >> 5;
    uint32_t bits = bm[wd] & (0xFFFFFFFFu >> (31 - (p & 31)));
    while (!bits) bits = bm[--wd];
    const uint64_t ls = (wd << 5) + (31 - __builtin_clz(bits));
    const uint32_t first = rank[ls];
    run_block[j] = blk;
    run_mask[j] = mask;
    run_first[j] = first;
    run_records[j] = (j - first) + (__popc(mask) >= 2 ? 1u : 0u);
}

__global__ void k_contrib_emit(const uint32_t *run_block, const uint32_t *run_mask, const uint32_t *run_first, const unsigned long long *run_at,
                               uint64_t n_runs, int kbits, unsigned long long *keys, uint2 *vals)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_runs) return;
    const uint32_t wj = run_block[j], mj = run_mask[j];
    unsigned long long at = run_at[j];
    for (uint32_t i = run_first[j]; i < j; i++) {
        keys[at] = ((unsigned long long)run_block[i] << kbits) | wj;
        vals[at] = make_uint2(run_mask[i], mj);
        at++;
    }
    if (__popc(mj) >= 2) {
        keys[at] = ((unsigned long long)wj << kbits) | wj;
        vals[at] = make_uint2(mj, mj);
    }
}

__global__ void k_blk_min(const uint32_t *sizes, uint32_t n_ref, uint32_t n_blocks, uint32_t *blk_min)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    uint32_t m = 0xFFFFFFFFu;
    for (uint32_t g = b * 32; g < min(n_ref, b * 32 + 32); g++) {
        const uint32_t s = sizes[g];
        if (s && s < m) m = s;
    }
    blk_min[b] = m;
}

// directory keys from the sort keys: b << kbits | w  ->  b << 32 | w
__global__ void k_tile_key_expand(const unsigned long long *packed, unsigned long long n, int kbits, unsigned long long *out)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = ((packed[i] >> kbits) << 32) | (packed[i] & ((1ULL << kbits) - 1ULL));
}

__global__ void k_widen_u32(const uint32_t *in, uint64_t n, unsigned long long *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// Sort keys of the tile directory: records per smallest sketch of the tile (per largest of the two block minima for
// jaccard, per smallest for containment), as a float, complemented so that an ascending sort lists the densest tiles first
__global__ void k_tile_ratio_keys(const unsigned long long *tile_key, const unsigned long long *tile_start, const uint32_t *blk_min,
                                  unsigned long long n_tiles, uint32_t *key_j, uint32_t *key_c, uint32_t *ids)
{
    const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const unsigned long long key = tile_key[t];
    const uint32_t mb = blk_min[(uint32_t)(key >> 32)], mw = blk_min[(uint32_t)key];
    const float n = (float)(tile_start[t + 1] - tile_start[t]);
    key_j[t] = ~__float_as_uint(n / (float)max(mb, mw));
    key_c[t] = ~__float_as_uint(n / (float)min(mb, mw));
    ids[t] = (uint32_t)t;
}

// table[k] = number of tiles whose ratio is at least 2^(-k/8) (less a margin for the float rounding): the host picks the
// launch size of a threshold from it.  keys: the complemented ratios, ascending.
__global__ void k_tile_prefix_table(const uint32_t *keys, unsigned long long n_tiles, unsigned long long *table)
{
    const int k = threadIdx.x;
    if (k >= kTileTable) return;
    const float theta = exp2f(-(float)k / 8.0f) * (1.0f - 1e-4f);
    const uint32_t limit = ~__float_as_uint(theta);   // ratio >= theta  <=>  key <= limit
    unsigned long long lo = 0, hi = n_tiles;
    while (lo < hi) {
        const unsigned long long mid = (lo + hi) >> 1;
        if (keys[mid] <= limit) lo = mid + 1; else hi = mid;
    }
    table[k] = lo;
}

// the directory in launch order (tiles by density, descending, for one metric), everything a workgroup of rk_tile_kernel needs
// to know in ONE 32-byte entry -- before: order -> key -> start, start + 1 -> two block minima, a chain of four round trips at
// the start of every workgroup.  Entry: {b, w, first record (2 words)} {records, lower bound of any cell's denominator, first slot
// of its row masks in the padded array (2 words)}
__global__ void k_tile_dir(const uint32_t *order, const unsigned long long *tile_key, const unsigned long long *tile_start,
                           const unsigned long long *slot_start, const uint32_t *blk_min, unsigned long long n_tiles, int metric, uint4 *out)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tiles) return;
    const uint32_t t = order[i];
    const unsigned long long key = tile_key[t], s0 = tile_start[t], n = tile_start[t + 1] - s0, d0 = slot_start[t];   // (d0: even, in the row-mask array)
    const uint32_t b = (uint32_t)(key >> 32), w = (uint32_t)key;
    const uint32_t mb = blk_min[b], mw = blk_min[w];
    out[2 * i] = make_uint4(b, w, (uint32_t)s0, (uint32_t)(s0 >> 32));
    out[2 * i + 1] = make_uint4((uint32_t)min(n, 0xFFFFFFFFULL), metric ? min(mb, mw) : max(mb, mw), (uint32_t)d0, (uint32_t)(d0 >> 32));
}

__global__ void k_tile_max(const unsigned long long *tile_start, unsigned long long n_tiles, unsigned long long *out)
{
    const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = t < n_tiles ? tile_start[t + 1] - tile_start[t] : 0ULL;
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned long long)__shfl_xor((long long)v, o));
    if ((threadIdx.x & 63) == 0 && v) atomicMax(out, v);
}
// padded[t] = the tile's record count rounded up to even (scanned in place into the tiles' first slots)
__global__ void k_tile_padded_counts(const unsigned long long *tile_start, unsigned long long n_tiles, unsigned long long *padded)
{
    const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_tiles) padded[t] = (tile_start[t + 1] - tile_start[t] + 1ULL) & ~1ULL;
}
// one wave per tile: its records (sorted AoS) go to the split arrays from the tile's even first slot on
__global__ void k_tile_split(const uint2 *recs, const unsigned long long *tile_start, const unsigned long long *slot_start, unsigned long long n_tiles,
                             uint32_t *rows, uint32_t *cols)
{
    const unsigned long long t = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= n_tiles) return;
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long s0 = tile_start[t], n = tile_start[t + 1] - s0, d0 = slot_start[t];
    for (unsigned long long i = lane; i < n; i += 64) {
        const uint2 r = recs[s0 + i];
        rows[d0 + i] = r.x;
        cols[d0 + i] = r.y;
    }
}

#define RK_TILE_TRY(call) do { int rc__ = (call); if (rc__) return rc__; } while (0)

// exclusive scan of n u64 values in place + their total (one read-back: the build is lazy and once per index)
int tile_scan_u64(rk_ctx *ctx, unsigned long long *v, uint64_t n, unsigned long long *total, hipStream_t st)
{
    *total = 0;
    if (!n) return RK_OK;
    unsigned long long last_in = 0, last_out = 0;
    RK_TILE_TRY(rk_read_back(ctx, &last_in, v + (n - 1), 8, st));
    size_t tb = 0;
    RK_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, v, v, 0ULL, n, rocprim::plus<unsigned long long>(), st));
    DevBuf<char> tmp(ctx);
    RK_HIP(ctx, tmp.alloc(tb));
    RK_HIP(ctx, rocprim::exclusive_scan(tmp.p, tb, v, v, 0ULL, n, rocprim::plus<unsigned long long>(), st));
    RK_TILE_TRY(rk_read_back(ctx, &last_out, v + (n - 1), 8, st));
    *total = last_in + last_out;
    return RK_OK;
}

}  // namespace

static int tiles_build_body(rk_ctx *ctx, rk_index *idx, hipStream_t st);
int rk_tiles_build(rk_ctx *ctx, rk_index *idx, hipStream_t st)
{
    const int rc = tiles_build_body(ctx, idx, st);
    if (rc) {   // (kernels of a failing build may still write into temporaries that went back to the pool)
        (void)hipStreamSynchronize(st);
        (void)hipGetLastError();
    }
    return rc;
}
static int tiles_build_body(rk_ctx *ctx, rk_index *idx, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(idx->lazy_mu);
    if (idx->tiles_ready || idx->tiles_unusable) return RK_OK;
    const auto t_begin = std::chrono::steady_clock::now();
    const uint64_t H = idx->H, U = idx->U;
    const uint32_t n_blocks = (idx->n_ref + 31) / 32;
    const unsigned tpb = 256;
    auto nb = [&](uint64_t n) { return dim3((unsigned)((n + tpb - 1) / tpb)); };
    DevBuf<uint32_t> blk_min(ctx);
    RK_HIP(ctx, blk_min.alloc(n_blocks));
    hipLaunchKernelGGL(k_blk_min, nb(n_blocks), dim3(tpb), 0, st, idx->d_sizes, idx->n_ref, n_blocks, blk_min.p);
    DevBuf<unsigned long long> keys_out(ctx), tile_key(ctx), tile_start(ctx);
    DevBuf<uint2> vals_out(ctx);
    DevBuf<uint32_t> order_j(ctx), order_c(ctx);
    DevBuf<uint4> dir_j(ctx), dir_c(ctx);
    DevBuf<uint32_t> t_rows(ctx), t_cols(ctx);
    unsigned long long n_slots = 0;
    unsigned long long n_c = 0, n_t = 0;
    if (H && U) {
        DevBuf<uint32_t> bm(ctx), rank(ctx);
        DevBuf<uint8_t> starts(ctx);
        RK_HIP(ctx, bm.alloc(H / 32 + 2));
        RK_HIP(ctx, rank.alloc(H + 1));
        RK_HIP(ctx, starts.alloc(H + 1));
        RK_HIP(ctx, hipMemsetAsync(starts.p, 0, H + 1, st));
        hipLaunchKernelGGL(k_list_starts, nb(U), dim3(tpb), 0, st, idx->d_upos, U, starts.p);
        hipLaunchKernelGGL(k_run_flags, nb(H + 1), dim3(tpb), 0, st, idx->d_postings, starts.p, H, rank.p, bm.p);
        RK_HIP(ctx, hipGetLastError());
        uint32_t n_runs32 = 0;
        {
            size_t tb = 0;
            RK_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, rank.p, rank.p, 0u, H + 1, rocprim::plus<uint32_t>(), st));
            DevBuf<char> tmp(ctx);
            RK_HIP(ctx, tmp.alloc(tb));
            RK_HIP(ctx, rocprim::exclusive_scan(tmp.p, tb, rank.p, rank.p, 0u, H + 1, rocprim::plus<uint32_t>(), st));
            RK_TILE_TRY(rk_read_back(ctx, &n_runs32, rank.p + H, 4, st));
        }
        const uint64_t n_runs = n_runs32;
        DevBuf<uint32_t> run_block(ctx), run_mask(ctx), run_first(ctx), run_records(ctx);
        DevBuf<unsigned long long> run_at(ctx);
        RK_HIP(ctx, run_block.alloc(n_runs));
        RK_HIP(ctx, run_mask.alloc(n_runs));
        RK_HIP(ctx, run_first.alloc(n_runs));
        RK_HIP(ctx, run_records.alloc(n_runs));
        RK_HIP(ctx, run_at.alloc(n_runs));
        hipLaunchKernelGGL(k_run_emit, nb(H), dim3(tpb), 0, st, idx->d_postings, bm.p, rank.p, H, run_block.p, run_mask.p, run_first.p,
                           run_records.p);
        hipLaunchKernelGGL(k_widen_u32, nb(n_runs), dim3(tpb), 0, st, run_records.p, n_runs, run_at.p);
        RK_HIP(ctx, hipGetLastError());
        RK_TILE_TRY(tile_scan_u64(ctx, run_at.p, n_runs, &n_c, st));
        // A list that touches B blocks of 32 ids writes B (B + 1) / 2 records: a few hundred hashes shared by thousands of scattered
        // genomes are hundreds of millions of records (32 bytes each here, and the sort's scratch).  Beyond a multiple of the
        // postings the tile kernel is the wrong tool: the index is marked, and the self join stays with the row kernels.
        {
            const char *env = getenv("RK_TILE_BUDGET");
            const unsigned long long budget = env ? strtoull(env, nullptr, 10) : 4ULL * H + (1ULL << 22);
            if (n_c > budget) {
                idx->tiles_unusable = true;
                if (ctx->sw_dist_debug) fprintf(stderr, "[rk] tiles: %llu records for %llu postings exceed the budget of %llu: not built\n", n_c, (unsigned long long)H, budget);
                RK_HIP(ctx, hipStreamSynchronize(st));
                return RK_OK;
            }
        }
        if (n_c) {
            DevBuf<unsigned long long> keys(ctx);
            DevBuf<uint2> vals(ctx);
            if (keys.alloc(n_c) != hipSuccess || vals.alloc(n_c) != hipSuccess || keys_out.alloc(n_c) != hipSuccess || vals_out.alloc(n_c + 256) != hipSuccess)
                return rk_fail(ctx, RK_ERR_NOMEM, "cannot allocate %llu tile records", n_c);
            int kbits = 1;
            while (kbits < 32 && (1u << kbits) < n_blocks) kbits++;
            hipLaunchKernelGGL(k_contrib_emit, nb(n_runs), dim3(tpb), 0, st, run_block.p, run_mask.p, run_first.p, run_at.p, n_runs, kbits, keys.p, vals.p);
            RK_HIP(ctx, hipGetLastError());
            int bbits = 1;   // (the key is b << bbits | w: no more radix passes than the collection's size asks for)
            while (bbits < 32 && (1u << bbits) < n_blocks) bbits++;
            size_t tb = 0;
            RK_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tb, keys.p, keys_out.p, vals.p, vals_out.p, (size_t)n_c, 0, 2 * bbits, st));
            DevBuf<char> tmp(ctx);
            RK_HIP(ctx, tmp.alloc(tb));
            RK_HIP(ctx, rocprim::radix_sort_pairs(tmp.p, tb, keys.p, keys_out.p, vals.p, vals_out.p, (size_t)n_c, 0, 2 * bbits, st));
            // directory: distinct keys and their run lengths (reusing `keys` for the distinct keys)
            DevBuf<unsigned long long> n_runs_out(ctx);
            DevBuf<uint32_t> counts(ctx);
            RK_HIP(ctx, n_runs_out.alloc(1));
            RK_HIP(ctx, counts.alloc(n_c));
            size_t tb2 = 0;
            RK_HIP(ctx, rocprim::run_length_encode(nullptr, tb2, keys_out.p, (size_t)n_c, keys.p, counts.p, n_runs_out.p, st));
            DevBuf<char> tmp2(ctx);
            RK_HIP(ctx, tmp2.alloc(tb2));
            RK_HIP(ctx, rocprim::run_length_encode(tmp2.p, tb2, keys_out.p, (size_t)n_c, keys.p, counts.p, n_runs_out.p, st));
            RK_TILE_TRY(rk_read_back(ctx, &n_t, n_runs_out.p, 8, st));
            RK_HIP(ctx, tile_key.alloc(n_t));
            RK_HIP(ctx, tile_start.alloc(n_t + 1));
            hipLaunchKernelGGL(k_tile_key_expand, nb(n_t), dim3(tpb), 0, st, keys.p, n_t, bbits, tile_key.p);
            hipLaunchKernelGGL(k_widen_u32, nb(n_t), dim3(tpb), 0, st, counts.p, n_t, tile_start.p);
            RK_HIP(ctx, hipGetLastError());
            unsigned long long tot = 0;
            RK_TILE_TRY(tile_scan_u64(ctx, tile_start.p, n_t, &tot, st));
            RK_HIP(ctx, hipMemcpyAsync(tile_start.p + n_t, &n_c, 8, hipMemcpyHostToDevice, st));
            RK_HIP(ctx, hipStreamSynchronize(st));
            if (tot != n_c) return rk_fail(ctx, RK_ERR_HIP, "tile directory does not add up (%llu vs %llu records)", tot, n_c);
            if (n_t >= 0xFFFFFFFFULL) return rk_fail(ctx, RK_ERR_UNSUPPORTED, "more than 2^32-1 tiles");
            // the directory ordered by density, once per metric, and the launch sizes of 256 thresholds
            DevBuf<uint32_t> key_j(ctx), key_c(ctx), ids(ctx), key_out(ctx);
            DevBuf<unsigned long long> table(ctx);
            RK_HIP(ctx, key_j.alloc(n_t));
            RK_HIP(ctx, key_c.alloc(n_t));
            RK_HIP(ctx, ids.alloc(n_t));
            RK_HIP(ctx, key_out.alloc(n_t));
            RK_HIP(ctx, order_j.alloc(n_t));
            RK_HIP(ctx, order_c.alloc(n_t));
            RK_HIP(ctx, table.alloc(2 * kTileTable + 1));
            RK_HIP(ctx, hipMemsetAsync(table.p + 2 * kTileTable, 0, 8, st));
            hipLaunchKernelGGL(k_tile_max, nb(n_t), dim3(tpb), 0, st, tile_start.p, n_t, table.p + 2 * kTileTable);
            hipLaunchKernelGGL(k_tile_ratio_keys, nb(n_t), dim3(tpb), 0, st, tile_key.p, tile_start.p, blk_min.p, n_t, key_j.p, key_c.p, ids.p);
            RK_HIP(ctx, hipGetLastError());
            size_t tb3 = 0;
            RK_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tb3, key_j.p, key_out.p, ids.p, order_j.p, (size_t)n_t, 0, 32, st));
            DevBuf<char> tmp3(ctx);
            RK_HIP(ctx, tmp3.alloc(tb3));
            RK_HIP(ctx, rocprim::radix_sort_pairs(tmp3.p, tb3, key_j.p, key_out.p, ids.p, order_j.p, (size_t)n_t, 0, 32, st));
            hipLaunchKernelGGL(k_tile_prefix_table, dim3(1), dim3(kTileTable), 0, st, key_out.p, n_t, table.p);
            RK_HIP(ctx, hipStreamSynchronize(st));   // (key_out is reused)
            RK_HIP(ctx, rocprim::radix_sort_pairs(tmp3.p, tb3, key_c.p, key_out.p, ids.p, order_c.p, (size_t)n_t, 0, 32, st));
            hipLaunchKernelGGL(k_tile_prefix_table, dim3(1), dim3(kTileTable), 0, st, key_out.p, n_t, table.p + kTileTable);
            // a split copy of the records (row masks / column masks), every tile from an even slot on: the SROW variant of
            // rk_tile_kernel reads two neighbouring row masks as one 64-bit scalar (the lane mask of a v_cndmask) and streams the
            // column masks alone through LDS
            DevBuf<unsigned long long> slot_start(ctx);
            RK_HIP(ctx, slot_start.alloc(n_t + 1));
            hipLaunchKernelGGL(k_tile_padded_counts, nb(n_t), dim3(tpb), 0, st, tile_start.p, n_t, slot_start.p);
            RK_HIP(ctx, hipGetLastError());
            RK_TILE_TRY(tile_scan_u64(ctx, slot_start.p, n_t, &n_slots, st));
            RK_HIP(ctx, t_rows.alloc(n_slots + 256));
            RK_HIP(ctx, t_cols.alloc(n_slots + 256));
            RK_HIP(ctx, hipMemsetAsync(t_rows.p, 0, (n_slots + 256) * 4, st));
            RK_HIP(ctx, hipMemsetAsync(t_cols.p, 0, (n_slots + 256) * 4, st));
            hipLaunchKernelGGL(k_tile_split, dim3((unsigned)((n_t + 3) / 4)), dim3(256), 0, st, vals_out.p, tile_start.p, slot_start.p, n_t, t_rows.p, t_cols.p);
            RK_HIP(ctx, dir_j.alloc(2 * n_t));
            RK_HIP(ctx, dir_c.alloc(2 * n_t));
            hipLaunchKernelGGL(k_tile_dir, nb(n_t), dim3(tpb), 0, st, order_j.p, tile_key.p, tile_start.p, slot_start.p, blk_min.p, n_t, 0, dir_j.p);
            hipLaunchKernelGGL(k_tile_dir, nb(n_t), dim3(tpb), 0, st, order_c.p, tile_key.p, tile_start.p, slot_start.p, blk_min.p, n_t, 1, dir_c.p);
            RK_HIP(ctx, hipGetLastError());
            RK_HIP(ctx, hipStreamSynchronize(st));   // (slot_start leaves scope)
            RK_HIP(ctx, hipGetLastError());
            RK_HIP(ctx, hipMemcpyAsync(idx->tile_prefix, table.p, sizeof(idx->tile_prefix), hipMemcpyDeviceToHost, st));
            RK_HIP(ctx, hipMemcpyAsync(&idx->tile_max_records, table.p + 2 * kTileTable, 8, hipMemcpyDeviceToHost, st));
            RK_HIP(ctx, hipStreamSynchronize(st));
        }
    }
    RK_HIP(ctx, hipStreamSynchronize(st));
    idx->d_tile_contrib = vals_out.release();
    idx->d_tile_rows = t_rows.release();
    idx->d_tile_cols = t_cols.release();
    idx->n_tile_slots = n_slots;
    idx->d_tile_key = tile_key.release();
    idx->d_tile_start = tile_start.release();
    idx->d_blk_min = blk_min.release();
    idx->d_tile_dir[0] = dir_j.release();
    idx->d_tile_dir[1] = dir_c.release();
    idx->d_tile_order[0] = order_j.release();
    idx->d_tile_order[1] = order_c.release();
    idx->n_tiles = n_t;
    idx->n_tile_records = n_c;
    idx->tiles_ready = true;
    if (ctx->sw_dist_debug)
        fprintf(stderr, "[rk] tiles: %llu tiles, %llu records, biggest tile %llu (%u genomes), built in %.3f ms\n", n_t, n_c, (unsigned long long)idx->tile_max_records, idx->n_ref,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    return RK_OK;
}

