// rk_index.hip -- device-side inverted index (replaces transSketches, src/sketch.cpp:970-1017,
// and the .index load + prefix sum of src/dist.cpp:86-129).
//
// HBM layout of an index over N genomes, H postings, U distinct hashes:
//   postings  u32[H]    genome ids ordered (hash asc, genome asc)      == the .dict payload
//   uhash     u32[U]    sorted distinct hashes                         (compact, not 2^bits)
//   upos      u32[U+1]  posting offsets of each distinct hash
//   sizes     u32[N]    sketch sizes
//   selfrange uint2[n_self]  for every source element (genome g, hash h) with later sharers: the slice of h's posting
//                       list holding genomes > g (the all-vs-all triangle needs no lookup), either as the posting range
//                       (x, y) or, when the genomes lie within 32 ids, COMPACT as (bit 31 | first genome, bitmask of
//                       the ids first .. first+31); rows back to back, inside a row the slices "covered" by the
//                       pair partner last (self_split)
// Derived on demand and cached: dir (prefix directory into uhash, for 64-bit / > 2^30 hash spaces), the rank bitmap
// of the query path (rk_distq.hip), sum of squared list lengths.  The dense 2^bits count array of the .index file
// is only materialised by rk_index_export / consumed by rk_index_import.
//
// Build = ONE library primitive (rocprim radix sort of (hash, source element)) + 7 small kernels on the context's
// stream, every buffer from the context's pool, ONE 32-byte read-back at the end: no hipMalloc, no hipFree and
// no intermediate synchronisation in steady state.
#include <cstring>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <thread>

#include "rk_internal.h"

#define RK_TRY(call) do { int rc__ = (call); if (rc__) return rc__; } while (0)

namespace {

constexpr int kThreads = 256;
inline unsigned blocks_for(uint64_t n) { return (unsigned)((n + kThreads - 1) / kThreads); }

struct BuildResult {  // written by the kernels, read back once
    unsigned long long U, n_self, dups, flags;
    unsigned long long flagged;   // slice records that are near-flagged posting ranges (fast path only): related lists wider than a compact record
};

// Single-workgroup exclusive scan: store(i, sum of load(j), j < i) for i < n; returns the grand total.  The array is taken as
// rows of 64 (lane = column: every load and store of a wave is one coalesced access -- round 4; a thread owning a contiguous
// stretch made each access 64 separate lines), a wave owns R consecutive rows of every stretch of nw * R rows and holds them
// in registers: all its loads are in flight at once, so a stretch (16,384 entries with 16 waves) costs ONE memory round trip,
// a block-wide exchange of the waves' totals and R wave scans with a running carry.
template <class Acc, class L, class W> __device__ inline Acc block_scan_sweeps(uint32_t n, L load, W store)
{
    __shared__ Acc scan_part[1024 / 64];
    constexpr uint32_t R = 16;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const uint32_t rows = (n + 63) / 64;
    Acc carry = 0;
    for (uint32_t base = 0; base < rows; base += nw * R) {
        const uint32_t r0 = base + wave * R;
        Acc v[R], sum = 0;
#pragma unroll
        for (uint32_t u = 0; u < R; u++) {
            const uint32_t i = (r0 + u) * 64 + lane;
            v[u] = i < n ? load(i) : Acc(0);
        }
#pragma unroll
        for (uint32_t u = 0; u < R; u++) sum += v[u];
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (lane == 0) scan_part[wave] = sum;
        __syncthreads();
        Acc run = carry, all = 0;
        for (uint32_t w = 0; w < nw; w++) {
            if (w < wave) run += scan_part[w];
            all += scan_part[w];
        }
#pragma unroll
        for (uint32_t u = 0; u < R; u++) {
            const uint32_t i = (r0 + u) * 64 + lane;
            if ((r0 + u) * 64 < n) {   // (wave-uniform)
                Acc incl = v[u];
                for (int o = 1; o < 64; o <<= 1) {
                    const Acc t = __shfl_up(incl, o);
                    if ((int)lane >= o) incl += t;
                }
                if (i < n) store(i, run + incl - v[u]);
                run += __shfl(incl, 63);
            }
        }
        carry += all;
        __syncthreads();
    }
    return carry;
}

__global__ void k_sizes(const uint64_t *off, uint32_t n, uint32_t *sizes, uint64_t *off_copy)
{
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) sizes[g] = (uint32_t)(off[g + 1] - off[g]);
    if (g <= n) off_copy[g] = off[g];
}

// one wave per genome: gid[e] = genome of source element e, iota[e] = e (the sort's value array)
__global__ void k_fill_gid(const uint64_t *off, uint32_t n_genomes, uint32_t *gid, uint32_t *iota)
{
    const uint32_t g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g >= n_genomes) return;
    const uint64_t e1 = off[g + 1];
    for (uint64_t e = off[g] + (threadIdx.x & 63); e < e1; e += 64) {
        gid[e] = g;
        iota[e] = (uint32_t)e;
    }
}

template <class K> __global__ void k_head_flags(const K *keys, uint64_t n, uint32_t *flags)
{
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) flags[k] = (k == 0 || keys[k] != keys[k - 1]) ? 1u : 0u;
}

// gidx = inclusive scan of head flags (1-based group number)
template <class K>
__global__ void k_scatter_heads(const K *keys, const uint32_t *gidx, uint64_t n, K *uhash, uint32_t *upos, BuildResult *res)
{
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t g = gidx[k] - 1;
    if (k == 0 || gidx[k - 1] != gidx[k]) {
        uhash[g] = keys[k];
        upos[g] = (uint32_t)k;
    }
    if (k == n - 1) {
        upos[g + 1] = (uint32_t)n;
        res->U = (unsigned long long)g + 1;
    }
}

// class of a source element's "later genomes" slice, carried by the slice itself so that no separate (byte-granular,
// randomly scattered) class array is needed: (x, y) with x < y = open, x == y = empty, stored reversed (y, x) = covered
enum : uint8_t { kEmpty = 0, kOpen = 1, kCovered = 2 };
__device__ inline uint8_t slice_class(uint2 r) { return r.x == r.y ? kEmpty : (r.x < r.y ? kOpen : kCovered); }
__device__ inline uint2 slice_range(uint2 r) { return r.x <= r.y ? r : make_uint2(r.y, r.x); }

// postings[k] = genome of the k-th sorted element; selfrange of its source element = (k+1, end of the group): the
// sort is stable and source elements are genome-major, so the rest of the group holds the later genomes.
// covered (self join, row pairs): genome 2p+1 shares this hash with genome 2p, i.e. its predecessor in the posting
// list is its pair partner.  Walking the partner's slice then serves both rows (the slice of 2p starts with 2p+1 and
// continues with exactly the slice of 2p+1), so the pair kernel skips the covered slices of the odd row.
// CHECK_DUPS (sketches not known to be sets): a genome that repeats a hash sits next to itself in the group.
template <bool CHECK_DUPS, bool NO_SELF = false>
__global__ void k_postings_selfrange(const uint32_t *sorted_e, const uint32_t *gidx, const uint32_t *upos,
                                     const uint32_t *gid, uint64_t n, uint32_t *postings, uint2 *self_raw,
                                     BuildResult *res)
{
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t e = sorted_e[k];
    const uint32_t me = gid[e];
    postings[k] = me;
    if (NO_SELF) return;   // (an index without slice records: set sketches, nothing to check)
    const uint32_t g = gidx[k] - 1;
    const uint32_t end = upos[g + 1];
    const bool has_prev = k > upos[g];
    uint32_t prev = 0xFFFFFFFFu;
    if (has_prev && (CHECK_DUPS || (me & 1u))) prev = gid[sorted_e[k - 1]];
    if (CHECK_DUPS && has_prev && prev == me) res->dups = 1;
    const bool covered = (me & 1u) && has_prev && prev == me - 1 && (uint32_t)k + 1 < end;
    self_raw[e] = covered ? make_uint2(end, (uint32_t)k + 1) : make_uint2((uint32_t)k + 1, end);
}

// one wave per genome: number of open / covered slices of its row
__global__ void k_row_counts(const uint64_t *off, uint32_t n_genomes, const uint2 *self_raw, uint32_t *n_open, uint32_t *n_cov)
{
    const uint32_t g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g >= n_genomes) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t e1 = off[g + 1];
    uint32_t no = 0, nc = 0;
    for (uint64_t e = off[g] + lane; e < e1; e += 64) {
        const uint8_t c = slice_class(self_raw[e]);
        no += c == kOpen;
        nc += c == kCovered;
    }
    for (int o = 32; o > 0; o >>= 1) {
        no += __shfl_down(no, o);
        nc += __shfl_down(nc, o);
    }
    if (lane == 0) {
        n_open[g] = no;
        n_cov[g] = nc;
    }
}

// single workgroup: self_off = exclusive scan of the row lengths, self_split = where a row's covered slices start
__global__ void k_row_scan(const uint32_t *n_open, const uint32_t *n_cov, uint32_t n_genomes, uint64_t *self_off,
                           uint64_t *self_split, BuildResult *res)
{
    const unsigned long long all = block_scan_sweeps<unsigned long long>(
        n_genomes, [&](uint32_t g) { return (unsigned long long)n_open[g] + n_cov[g]; },
        [&](uint32_t g, unsigned long long before) {
            self_off[g] = before;
            self_split[g] = before + n_open[g];
        });
    if (threadIdx.x == 0) {
        self_off[n_genomes] = all;
        self_split[n_genomes] = all;
        res->n_self = all;
    }
}

// A slice whose genomes all lie within 32 ids -- the normal case in a collection ordered by similarity, where a hash is
// shared by a handful of neighbouring strains -- is stored COMPACT: (bit 31 | first genome, bitmask of the genomes
// first+0 .. first+31; bit 0 is always set).  The distance kernel then needs no posting load for it at all: the
// 8-byte slice record IS the posting list.  Other slices stay posting ranges (x, y), x < 2^31.
// A slice that stays a posting range carries the NEAR flag in bit 31 of y: its first genome lies within 32 ids of the
// element's own genome g (only then can it have members inside the window of rk_near_kernel).
__device__ inline uint2 compact_slice(uint2 r, const uint32_t *postings, uint32_t g, bool compact)
{
    const uint32_t first = postings[r.x], last = postings[r.y - 1];
    if (!compact || last - first > 31u) return make_uint2(r.x, r.y | (first - g <= 32u ? 0x80000000u : 0u));
    uint32_t mask = 1;
    for (uint32_t k = r.x + 1; k < r.y; k++) mask |= 1u << (postings[k] - first);
    return make_uint2(0x80000000u | first, mask);
}

// one wave per genome: open slices to the front of the row, covered ones behind them, empty ones dropped; the order
// inside each class is the source order (ballot ranks)
// compact: the sketches are sets.  A genome that repeats a hash sits several times in the list and must be counted as
// often (src/dist.cpp:199-202): a bitmask cannot say that, so the slices of a multiset index stay posting ranges.
__global__ void k_row_place(const uint64_t *off, uint32_t n_genomes, const uint2 *self_raw,
                            const uint64_t *self_off, const uint64_t *self_split, const uint32_t *postings, bool compact, uint2 *out)
{
    const uint32_t g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g >= n_genomes) return;
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    uint64_t at_open = self_off[g], at_cov = self_split[g];
    const uint64_t e0 = off[g], e1 = off[g + 1];
    for (uint64_t base = e0; base < e1; base += 64) {
        const uint64_t e = base + lane;
        const uint2 raw = e < e1 ? self_raw[e] : make_uint2(0, 0);
        const uint8_t c = slice_class(raw);
        uint2 r = slice_range(raw);
        if (c != kEmpty) r = compact_slice(r, postings, g, compact);
        const unsigned long long mo = __ballot(c == kOpen), mc = __ballot(c == kCovered);
        if (c == kOpen) out[at_open + __popcll(mo & lt)] = r;
        if (c == kCovered) out[at_cov + __popcll(mc & lt)] = r;
        at_open += __popcll(mo);
        at_cov += __popcll(mc);
    }
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

// one wave per genome: compact records of its row, records the pair kernel walks (all of an even row, the uncovered of an odd one)
__global__ void k_self_stats(const uint2 *selfrange, const uint64_t *self_off, const uint64_t *self_split, uint32_t n_genomes,
                             unsigned long long *acc)
{
    const uint32_t g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g >= n_genomes) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t e0 = self_off[g], e1 = self_off[g + 1];
    unsigned long long cpt = 0;
    for (uint64_t e = e0 + lane; e < e1; e += 64) cpt += selfrange[e].x >> 31;
    for (int o = 32; o > 0; o >>= 1) cpt += __shfl_down(cpt, o);
    if (lane == 0) {
        if (cpt) atomicAdd(&acc[0], cpt);
        atomicAdd(&acc[1], (unsigned long long)(((g & 1u) ? self_split[g] : e1) - e0));
    }
}

// a posting list that names a genome twice (or out of order): the imported index does not come from set sketches
__global__ void k_check_lists(const uint32_t *postings, const uint32_t *upos, uint64_t U, uint32_t *bad)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= U) return;
    for (uint32_t k = upos[i] + 1; k < upos[i + 1]; k++)
        if (postings[k] <= postings[k - 1]) { *bad = 1; return; }
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


// ---- internal genome order ------------------------------------------------------------------------------------
// Compact slices, list records and row pairs pay off when the genomes that share a hash have neighbouring ids.  A
// collection arrives in whatever order its files were listed (or completed, src/sketch.cpp:558-568), so the build
// renumbers it: genomes are clustered by their kMinK smallest hashes (a bottom-k MinHash of the sketch: two genomes
// with Jaccard similarity j share each of them with probability ~j), every genome attaches to the smallest-id genome
// that shares at least two of them, clusters become runs of consecutive internal ids (ordered by their smallest
// member, members by their original id -- a collection that already lists relatives together keeps its order).
// The clustering only shapes the data layout: every count stays exact whatever it decides.
constexpr uint32_t kMinK = 16;
constexpr unsigned long long kEmptySlot = ~0ULL;
__device__ inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
template <class K> __device__ inline uint32_t fold_hash(K h)
{
    if constexpr (sizeof(K) == 8) return (uint32_t)(h ^ (h >> 32));
    else return (uint32_t)h;
}

// table slot = (hash << 32 | smallest genome that lists the hash among its kMinK smallest); open addressing.
// (round 5) A workgroup holds 64 neighbouring genomes -- relatives, in a collection listed species by species -- and first settles
// "the smallest genome per hash" among ITS 1,024 elements in an LDS table; only the winners go to the device-wide table.  Its loads
// and atomics are device-scope round trips (~30 us per million, and whatever streams beside them waits: 0.55 ms at 500,000 genomes,
// paid by every shard of a sharded build); among relatives an LDS table leaves a sixteenth of them.
constexpr uint32_t kInsertThreads = 1024, kInsertLocal = 2048;
__device__ inline void minhash_insert_global(unsigned long long *table, uint32_t mask, unsigned long long mine)
{
    const uint32_t h = (uint32_t)(mine >> 32);
    uint32_t slot = mix32(h) & mask;
    for (uint32_t probe = 0; probe <= mask; probe++) {
        unsigned long long cur = __hip_atomic_load(&table[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == kEmptySlot) {
            cur = atomicCAS(&table[slot], kEmptySlot, mine);
            if (cur == kEmptySlot) return;
        }
        if ((uint32_t)(cur >> 32) == h) {
            // (the slot only ever decreases: a genome that finds a smaller one there has nothing to add)
            if (cur > mine) atomicMin(&table[slot], mine);
            return;
        }
        slot = (slot + 1) & mask;
    }
}
template <class K>
__global__ __launch_bounds__(kInsertThreads) void k_minhash_insert(const K *hashes, const uint64_t *off, uint32_t n_genomes, unsigned long long *table, uint32_t mask)
{
    __shared__ unsigned long long loc[kInsertLocal];   // (at most half full: 1,024 elements)
    for (uint32_t s = threadIdx.x; s < kInsertLocal; s += kInsertThreads) loc[s] = kEmptySlot;
    __syncthreads();
    const uint64_t t = (uint64_t)blockIdx.x * kInsertThreads + threadIdx.x;
    const uint32_t g = (uint32_t)(t / kMinK), i = (uint32_t)(t % kMinK);
    if (g < n_genomes) {
        const uint64_t e = off[g] + i;
        if (e < off[g + 1]) {
            const uint32_t h = fold_hash(hashes[e]);
            const unsigned long long mine = ((unsigned long long)h << 32) | g;
            uint32_t slot = (mix32(h) >> 11) & (kInsertLocal - 1);   // (other bits than the device-wide table's)
            for (uint32_t probe = 0; probe < kInsertLocal; probe++) {
                unsigned long long cur = loc[slot];
                if (cur == kEmptySlot) {
                    cur = atomicCAS(&loc[slot], kEmptySlot, mine);
                    if (cur == kEmptySlot) break;
                }
                if ((uint32_t)(cur >> 32) == h) {
                    if (cur > mine) atomicMin(&loc[slot], mine);
                    break;
                }
                slot = (slot + 1) & (kInsertLocal - 1);
            }
        }
    }
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < kInsertLocal; s += kInsertThreads) {
        const unsigned long long v = loc[s];
        if (v != kEmptySlot) minhash_insert_global(table, mask, v);
    }
}

// parent[g] = the smallest genome below g that shares at least two of g's kMinK smallest hashes (g itself if none)
template <class K>
__global__ void k_minhash_vote(const K *hashes, const uint64_t *off, uint32_t n_genomes, const unsigned long long *table,
                               uint32_t mask, uint32_t *parent)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_genomes) return;
    const uint64_t e0 = off[g];
    const uint32_t n = (uint32_t)min((uint64_t)kMinK, off[g + 1] - e0);
    uint32_t r[kMinK];
#pragma unroll
    for (uint32_t i = 0; i < kMinK; i++) {
        r[i] = g;
        if (i < n) {
            const uint32_t h = fold_hash(hashes[e0 + i]);
            uint32_t slot = mix32(h) & mask;
            for (uint32_t probe = 0; probe <= mask; probe++) {
                const unsigned long long cur = table[slot];
                if (cur == kEmptySlot) break;
                if ((uint32_t)(cur >> 32) == h) { r[i] = (uint32_t)cur; break; }
                slot = (slot + 1) & mask;
            }
        }
    }
    uint32_t best = g;
#pragma unroll
    for (uint32_t i = 0; i < kMinK; i++) {
        uint32_t votes = 0;
#pragma unroll
        for (uint32_t j = 0; j < kMinK; j++) votes += r[j] == r[i];
        if (votes >= 2 && r[i] < best) best = r[i];
    }
    parent[g] = best;
}

// key[g] = (root of g's cluster, g): parents only ever point to smaller ids, so the walk ends at a fixed point
__global__ void k_cluster_keys(const uint32_t *parent, uint32_t n_genomes, int id_bits, unsigned long long *keys)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_genomes) return;
    uint32_t p = parent[g];
    for (uint32_t hop = 0; hop < n_genomes; hop++) {
        const uint32_t q = parent[p];
        if (q == p) break;
        p = q;
    }
    keys[g] = ((unsigned long long)p << id_bits) | g;
}

// sorted keys -> orig[internal id], sizes in internal order
__global__ void k_order_from_keys(const unsigned long long *keys, uint32_t n_genomes, int id_bits, const uint64_t *off,
                                  uint32_t *orig, uint32_t *sizes)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_genomes) return;
    const uint32_t g = (uint32_t)(keys[i] & ((1ULL << id_bits) - 1ULL));
    orig[i] = g;
    sizes[i] = (uint32_t)(off[g + 1] - off[g]);
}

// single workgroup: off_new = exclusive scan of the sizes in internal order
__global__ void k_offsets_scan(const uint32_t *sizes, uint32_t n_genomes, uint64_t *off_new)
{
    const unsigned long long all = block_scan_sweeps<unsigned long long>(
        n_genomes, [&](uint32_t g) { return (unsigned long long)sizes[g]; },
        [&](uint32_t g, unsigned long long before) { off_new[g] = before; });
    if (threadIdx.x == 0) off_new[n_genomes] = all;
}

// Small collections (<= 32,768 genomes): the (root, genome) keys are ordered by COUNTING -- the keys are distinct, so a
// key's position is the number of smaller keys; every workgroup compares 256 keys with a stretch of 512 others (scalar
// loads: the stretch is the same for all lanes) and adds its partial counts.  10,000 keys: 800 workgroups, a few
// microseconds -- a device-wide sort of so few keys costs 70 us of launches.
constexpr uint32_t kRankThreads = 256, kRankPer = 4, kRankQ = kRankThreads * kRankPer, kRankStretch = 256, kRankMaxN = 32768;
// (root, genome) packed into 32 bits: 2 * id_bits <= 30 for up to 32,768 genomes
// (also zeroes the rank counters k_rank_keys adds to: a fill of its own is one more 6 us dispatch in the chain)
__global__ void k_cluster_keys32(const uint32_t *parent, uint32_t n_genomes, int id_bits, uint32_t *keys, uint32_t *rank)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_genomes) return;
    rank[g] = 0;
    uint32_t p = parent[g];
    for (uint32_t hop = 0; hop < n_genomes; hop++) {
        const uint32_t q = parent[p];
        if (q == p) break;
        p = q;
    }
    keys[g] = (p << id_bits) | g;
}
// every thread ranks kRankPer keys against a stretch of 256 others staged in LDS (16-byte broadcast reads: four
// others per read, sixteen comparisons per read)
__global__ __launch_bounds__(kRankThreads) void k_rank_keys(const uint32_t *__restrict__ keys, uint32_t n, uint32_t *rank)
{
    __shared__ __attribute__((aligned(16))) uint32_t others[kRankStretch];
    const uint32_t j0 = blockIdx.y * kRankStretch;
    for (uint32_t j = threadIdx.x; j < kRankStretch; j += kRankThreads) others[j] = j0 + j < n ? keys[j0 + j] : 0xFFFFFFFFu;
    uint32_t mine[kRankPer], below[kRankPer];
#pragma unroll
    for (uint32_t i = 0; i < kRankPer; i++) {
        const uint32_t q = blockIdx.x * kRankQ + i * kRankThreads + threadIdx.x;
        mine[i] = q < n ? keys[q] : 0u;
        below[i] = 0;
    }
    __syncthreads();
    const uint4 *o4 = reinterpret_cast<const uint4 *>(others);
#pragma unroll 4
    for (uint32_t j = 0; j < kRankStretch / 4; j++) {
        const uint4 v = o4[j];
#pragma unroll
        for (uint32_t i = 0; i < kRankPer; i++) below[i] += (v.x < mine[i]) + (v.y < mine[i]) + (v.z < mine[i]) + (v.w < mine[i]);
    }
#pragma unroll
    for (uint32_t i = 0; i < kRankPer; i++) {
        const uint32_t q = blockIdx.x * kRankQ + i * kRankThreads + threadIdx.x;
        if (q < n && below[i]) atomicAdd(&rank[q], below[i]);
    }
}
__global__ void k_order_from_rank(const uint32_t *rank, uint32_t n_genomes, const uint64_t *off, uint32_t *orig, uint32_t *sizes)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_genomes) return;
    const uint32_t i = rank[g];   // (the g-th key belongs to genome g: k_cluster_keys32)
    orig[i] = g;
    sizes[i] = (uint32_t)(off[g + 1] - off[g]);
}

// one wave per internal genome: its hashes move to their place in the internal-order CSR
template <class K>
__global__ void k_gather_sketches(const K *hashes, const uint64_t *off, const uint32_t *orig, const uint64_t *off_new,
                                  uint32_t n_genomes, K *out)
{
    const uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n_genomes) return;
    const uint64_t src = off[orig[i]], dst = off_new[i], n = off_new[i + 1] - dst;
    for (uint64_t k = threadIdx.x & 63; k < n; k += 64) out[dst + k] = hashes[src + k];
}

// inv[orig[i]] = i
// tab[caller's genome] = (internal id, start of its sketch in the internal-order element space): what k_bucket_emit needs of
// a key's genome, in one gather (the fast build takes H < 2^30)
__global__ void k_emit_table(const uint32_t *inv, const uint64_t *off_new, uint32_t n, uint2 *tab)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) tab[g] = make_uint2(inv[g], (uint32_t)off_new[inv[g]]);
}

__global__ void k_invert_order(const uint32_t *orig, uint32_t n, uint32_t *inv)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) inv[orig[i]] = i;
}

__global__ void k_iota(uint32_t n, uint32_t *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}


// .dict order of a renumbered index: within every list the genomes ascend by the CALLER's ids (src/sketch.cpp:979-985
// pushes genome i onto hashMapId[hash] for i ascending).  key = (list number, original id), one radix sort.
__global__ void k_export_keys(const uint32_t *postings, const uint32_t *upos, uint64_t U, uint64_t H, const uint32_t *orig,
                              unsigned long long *keys)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= H) return;
    uint64_t lo = 0, hi = U;  // largest u with upos[u] <= k
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (upos[mid] <= k) lo = mid; else hi = mid;
    }
    keys[k] = ((unsigned long long)lo << 32) | orig[postings[k]];
}
__global__ void k_low_halves(const unsigned long long *keys, uint64_t n, uint32_t *out)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = (uint32_t)keys[k];
}

// the postings as the .dict file lists them, in a pool buffer the caller frees (null: the index's own array already is)
int postings_in_caller_ids(rk_ctx *ctx, const rk_index *idx, hipStream_t st, uint32_t **out)
{
    *out = nullptr;
    if (!idx->relabeled || !idx->H) return RK_OK;
    DevBuf<unsigned long long> keys(ctx), sorted(ctx);
    DevBuf<uint32_t> res(ctx);
    DevBuf<char> tmp(ctx);
    RK_HIP(ctx, keys.alloc(idx->H));
    RK_HIP(ctx, sorted.alloc(idx->H));
    RK_HIP(ctx, res.alloc(idx->H));
    int ubits = 1;
    while (ubits < 32 && (1ULL << ubits) < idx->U) ubits++;
    hipLaunchKernelGGL(k_export_keys, dim3(blocks_for(idx->H)), dim3(kThreads), 0, st, idx->d_postings, idx->d_upos, idx->U, idx->H,
                       idx->d_orig, keys.p);
    RK_TRY(rk_prim_sort_keys_u64(ctx, keys.p, sorted.p, idx->H, 0, (unsigned)(32 + ubits), st));
    hipLaunchKernelGGL(k_low_halves, dim3(blocks_for(idx->H)), dim3(kThreads), 0, st, sorted.p, idx->H, res.p);
    RK_HIP(ctx, hipGetLastError());
    RK_HIP(ctx, hipStreamSynchronize(st));  // the temporaries return to the pool
    *out = res.release();
    return RK_OK;
}

#include "rk_index_fast.inc"
#include "rk_index_tiles.inc"

template <class T> int pool_array(rk_ctx *ctx, T **out, size_t n)
{
    *out = static_cast<T *>(rk_pool_alloc(ctx, (n ? n : 1) * sizeof(T)));
    return *out ? RK_OK : rk_fail(ctx, RK_ERR_NOMEM, "cannot allocate %llu bytes on the device", (unsigned long long)(n * sizeof(T)));
}

void set_dir_shape(rk_index *idx)
{
    int dbits = 10;
    while (dbits < 24 && (1ULL << dbits) < 2 * idx->U) dbits++;
    if (dbits > idx->hash_bits) dbits = idx->hash_bits;
    idx->dir_bits = dbits;
    idx->dir_shift = idx->hash_bits - dbits;
}

// imported index: are the posting lists strictly ascending (no genome twice)?
int classify_lists(rk_ctx *ctx, rk_index *idx, hipStream_t st)
{
    DevBuf<uint32_t> bad(ctx);
    RK_HIP(ctx, bad.alloc(1));
    RK_HIP(ctx, hipMemsetAsync(bad.p, 0, 4, st));
    if (idx->U)
        hipLaunchKernelGGL(k_check_lists, dim3(blocks_for(idx->U)), dim3(kThreads), 0, st, idx->d_postings, idx->d_upos, idx->U, bad.p);
    RK_HIP(ctx, hipGetLastError());
    uint32_t b = 0;
    RK_TRY(rk_read_back(ctx, &b, bad.p, 4, st));
    idx->ref_sets = b == 0;
    return RK_OK;
}

}  // namespace

// prefix directory into the sorted distinct hashes: built on first use (64-bit hashes, hash spaces above 2^30)
// ---- slice records on first use (an index built with tile records has none) ------------------------------------------
// one thread per posting list: the slice record of every member, by posting position, as the bucket emission writes them
// (class carried by the record: slice_class2).  A list longer than a compact record walks once; the compact ones are short.
__global__ void k_slices_from_lists(const uint32_t *postings, const uint32_t *upos, uint64_t U, uint2 *raw, uint32_t *pos)
{
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const uint32_t s = upos[u], e = upos[u + 1];
    const uint32_t last = e > s ? postings[e - 1] : 0u;
    uint32_t prev = 0xFFFFFFFFu;
    for (uint32_t k = s; k < e; k++) {
        const uint32_t g = postings[k];
        uint2 rec = make_uint2(0, 0);
        if (k + 1 < e) {
            const uint32_t first = postings[k + 1];
            const bool covered = (g & 1u) && prev == g - 1;
            if (last - first <= 31u) {
                uint32_t mask = 0;
                for (uint32_t m = k + 1; m < e; m++) mask |= 1u << (postings[m] - first);
                rec = make_uint2(0x80000000u | first, covered ? mask & ~1u : mask);
            } else {
                rec = make_uint2((k + 1) | (covered ? 0x40000000u : 0u), e | (first - g <= 32u ? 0x80000000u : 0u));
            }
        }
        raw[k] = rec;
        pos[k] = k;
        prev = g;
    }
}
__global__ void k_gather_slices(const uint2 *raw, const uint32_t *sorted_pos, uint64_t H, uint2 *out)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < H) out[e] = raw[sorted_pos[e]];
}

int rk_index_ensure_slices(rk_ctx *ctx, rk_index *idx, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(idx->lazy_mu);
    if (idx->d_selfrange) return RK_OK;
    // (the raw record keeps its class in bit 30 of a posting offset: as in the fast build, H < 2^30 -- every index that was built
    // with tile records is)
    if (idx->slices_refused || idx->H >= (1ULL << 30) || !idx->ref_sets || !idx->d_src_off || idx->wide != (idx->d_uhash64 != nullptr))
        return rk_fail(ctx, RK_ERR_UNSUPPORTED, "this index has no slice records (2^31 postings or more, or RK_INDEX_NO_SELF): only sparse self "
                                                "joins (a threshold below distance 1.0) run on it");
    const auto t_begin = std::chrono::steady_clock::now();
    const uint64_t H = idx->H, U = idx->U;
    const uint32_t N = idx->n_ref;
    DevBuf<uint2> raw(ctx), self_raw(ctx), out(ctx);
    DevBuf<uint32_t> pos(ctx), keys_sorted(ctx), pos_sorted(ctx), n_open(ctx), n_cov(ctx);
    DevBuf<uint64_t> self_off(ctx), self_split(ctx);
    DevBuf<BuildResult> res(ctx);
    RK_HIP(ctx, raw.alloc(H));
    RK_HIP(ctx, self_raw.alloc(H));
    RK_HIP(ctx, out.alloc(H + 1));
    RK_HIP(ctx, pos.alloc(H));
    RK_HIP(ctx, keys_sorted.alloc(H));
    RK_HIP(ctx, pos_sorted.alloc(H));
    RK_HIP(ctx, n_open.alloc((size_t)N + 1));
    RK_HIP(ctx, n_cov.alloc((size_t)N + 1));
    RK_HIP(ctx, self_off.alloc((size_t)N + 1));
    RK_HIP(ctx, self_split.alloc((size_t)N + 1));
    RK_HIP(ctx, res.alloc(1));
    RK_HIP(ctx, hipMemsetAsync(res.p, 0, sizeof(BuildResult), st));
    BuildResult r{0, 0, 0, 0, 0};
    if (H) {
        hipLaunchKernelGGL(k_slices_from_lists, dim3(blocks_for(U)), dim3(kThreads), 0, st, idx->d_postings, idx->d_upos, U, raw.p, pos.p);
        RK_HIP(ctx, hipGetLastError());
        // a genome's elements in ascending hash order = its row (the sketches are sets, sorted): a STABLE sort of the posting
        // positions by genome
        int gbits = 1;
        while ((1ULL << gbits) < N) gbits++;
        RK_TRY(rk_prim_sort_pairs_u32_u32(ctx, idx->d_postings, keys_sorted.p, pos.p, pos_sorted.p, H, (unsigned)gbits, st));
        hipLaunchKernelGGL(k_gather_slices, dim3(blocks_for(H)), dim3(kThreads), 0, st, raw.p, pos_sorted.p, H, self_raw.p);
        const unsigned wave_blocks = (N + 3) / 4;
        hipLaunchKernelGGL(k_row_counts2, dim3(wave_blocks), dim3(kThreads), 0, st, idx->d_src_off, N, self_raw.p, n_open.p, n_cov.p);
        hipLaunchKernelGGL(k_row_scan, dim3(1), dim3(1024), 0, st, n_open.p, n_cov.p, N, self_off.p, self_split.p, res.p);
        hipLaunchKernelGGL(k_row_place2, dim3(wave_blocks), dim3(kThreads), 0, st, idx->d_src_off, N, self_raw.p, self_off.p, self_split.p, out.p);
        RK_HIP(ctx, hipGetLastError());
        RK_TRY(rk_read_back(ctx, &r, res.p, sizeof(r), st));
    } else {
        RK_HIP(ctx, hipMemsetAsync(self_off.p, 0, ((size_t)N + 1) * 8, st));
        RK_HIP(ctx, hipMemsetAsync(self_split.p, 0, ((size_t)N + 1) * 8, st));
        RK_HIP(ctx, hipStreamSynchronize(st));
    }
    idx->n_self = r.n_self;
    idx->d_self_off = self_off.release();
    idx->d_self_split = self_split.release();
    idx->d_selfrange = out.release();   // (last: the other threads test this pointer)
    if (ctx->sw_dist_debug)
        fprintf(stderr, "[rk] slice records built on first use: %llu records in %.3f ms\n", (unsigned long long)r.n_self,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    return RK_OK;
}

int rk_index_ensure_dir(rk_ctx *ctx, rk_index *idx, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(idx->lazy_mu);
    if (idx->d_dir) return RK_OK;
    const uint32_t nb = 1u << idx->dir_bits;
    DevBuf<uint32_t> dir(ctx);
    RK_HIP(ctx, dir.alloc((size_t)nb + 1));
    if (idx->wide)
        hipLaunchKernelGGL(k_dir<uint64_t>, dim3(blocks_for((uint64_t)nb + 1)), dim3(kThreads), 0, st,
                           idx->d_uhash64, idx->U, idx->dir_shift, nb, dir.p);
    else
        hipLaunchKernelGGL(k_dir<uint32_t>, dim3(blocks_for((uint64_t)nb + 1)), dim3(kThreads), 0, st,
                           idx->d_uhash, idx->U, idx->dir_shift, nb, dir.p);
    RK_HIP(ctx, hipGetLastError());
    RK_HIP(ctx, hipStreamSynchronize(st));  // once per index: a later call may come on another stream
    idx->d_dir = dir.release();
    return RK_OK;
}

extern "C" {

void rk_index_free(rk_index *idx)
{
    if (!idx) return;
    rk_ctx *ctx = idx->ctx;
    rk_pool_free(ctx, idx->d_postings);
    rk_pool_free(ctx, idx->d_uhash);
    rk_pool_free(ctx, idx->d_uhash64);
    rk_pool_free(ctx, idx->d_upos);
    rk_pool_free(ctx, idx->d_dir);
    rk_pool_free(ctx, idx->d_rankbm);
    rk_pool_free(ctx, idx->d_rankbase);
    rk_pool_free(ctx, idx->d_urec);
    rk_pool_free(ctx, idx->d_sizes);
    rk_pool_free(ctx, idx->d_selfrange);
    rk_pool_free(ctx, idx->d_shard_rec);
    rk_pool_free(ctx, idx->d_self_off);
    rk_pool_free(ctx, idx->d_self_split);
    rk_pool_free(ctx, idx->d_src_off);
    rk_pool_free(ctx, idx->d_orig);
    rk_pool_free(ctx, idx->d_fb);
    rk_pool_free(ctx, idx->d_tile_contrib);
    rk_pool_free(ctx, idx->d_tile_rows);
    rk_pool_free(ctx, idx->d_tile_cols);
    rk_pool_free(ctx, idx->d_tile_key);
    rk_pool_free(ctx, idx->d_tile_start);
    rk_pool_free(ctx, idx->d_blk_min);
    rk_pool_free(ctx, idx->d_tile_dir[0]);
    rk_pool_free(ctx, idx->d_tile_dir[1]);
    rk_pool_free(ctx, idx->d_tile_order[0]);
    rk_pool_free(ctx, idx->d_tile_order[1]);
    if (idx->h_fb_seen) (void)hipHostFree(idx->h_fb_seen);
    if (idx->fb_event) (void)hipEventDestroy((hipEvent_t)idx->fb_event);
    delete idx;
}

int rk_index_order(const rk_index *idx, uint32_t *orig_out)
{
    if (!idx || (!orig_out && idx->n_ref)) return RK_ERR_ARG;
    rk_ctx *ctx = idx->ctx;
    if (!idx->relabeled || !idx->d_orig) {
        for (uint32_t i = 0; i < idx->n_ref; i++) orig_out[i] = i;
        return RK_OK;
    }
    RK_HIP(ctx, hipSetDevice(ctx->device));
    RK_HIP(ctx, hipMemcpyAsync(orig_out, idx->d_orig, (size_t)idx->n_ref * 4, hipMemcpyDeviceToHost, ctx->stream));
    RK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RK_OK;
}

uint64_t rk_index_total(const rk_index *idx) { return idx ? idx->H : 0; }
uint64_t rk_index_distinct(const rk_index *idx) { return idx ? idx->U : 0; }
uint32_t rk_index_genomes(const rk_index *idx) { return idx ? idx->n_ref : 0; }
int rk_index_hash_bits(const rk_index *idx) { return idx ? idx->hash_bits : 0; }
int rk_index_built_fast(const rk_index *idx) { return idx && idx->built_fast ? 1 : 0; }
int rk_index_products(const rk_index *idx)
{
    if (!idx) return 0;
    return (idx->d_selfrange ? 1 : 0) | (idx->tiles_ready ? 2 : 0) | (idx->tiles_ready && idx->tiles_from_build ? 4 : 0);
}

uint64_t rk_index_sum_sq(const rk_index *cidx)
{
    if (!cidx) return 0;
    rk_index *idx = const_cast<rk_index *>(cidx);  // cached on first use: only the roofline report asks for it
    std::lock_guard<std::mutex> lk(idx->lazy_mu);
    if (idx->sum_sq_known) return idx->sum_sq;
    rk_ctx *ctx = idx->ctx;
    if (hipSetDevice(ctx->device) != hipSuccess) return 0;
    DevBuf<unsigned long long> acc(ctx);
    if (acc.alloc(1) != hipSuccess || hipMemsetAsync(acc.p, 0, 8, ctx->stream) != hipSuccess) return 0;
    if (idx->U)
        hipLaunchKernelGGL(k_sum_sq, dim3(blocks_for(idx->U)), dim3(kThreads), 0, ctx->stream, idx->d_upos, idx->U, acc.p);
    unsigned long long ss = 0;
    if (rk_read_back(ctx, &ss, acc.p, 8, ctx->stream) != RK_OK) return 0;
    idx->sum_sq = ss;
    idx->sum_sq_known = true;
    return ss;
}

int rk_index_self_stats(const rk_index *cidx, uint64_t out[4])
{
    if (!cidx || !out) return RK_ERR_ARG;
    rk_index *idx = const_cast<rk_index *>(cidx);
    std::lock_guard<std::mutex> lk(idx->lazy_mu);
    rk_ctx *ctx = idx->ctx;
    if (!idx->self_stats_known && idx->d_selfrange && idx->n_ref) {
        RK_HIP(ctx, hipSetDevice(ctx->device));
        DevBuf<unsigned long long> acc(ctx);
        RK_HIP(ctx, acc.alloc(2));
        RK_HIP(ctx, hipMemsetAsync(acc.p, 0, 16, ctx->stream));
        hipLaunchKernelGGL(k_self_stats, dim3((idx->n_ref + 3) / 4), dim3(kThreads), 0, ctx->stream, idx->d_selfrange, idx->d_self_off,
                           idx->d_self_split, idx->n_ref, acc.p);
        RK_HIP(ctx, hipGetLastError());
        unsigned long long v[2] = {0, 0};
        RK_TRY(rk_read_back(ctx, v, acc.p, 16, ctx->stream));
        idx->self_stats[0] = idx->n_self;
        idx->self_stats[1] = v[0];
        idx->self_stats[2] = v[1];
        idx->self_stats_known = true;
    }
    idx->self_stats[3] = idx->tiles_ready ? idx->n_tile_records : 0;
    for (int i = 0; i < 4; i++) out[i] = idx->self_stats[i];
    return RK_OK;
}

static int index_build_impl(rk_ctx *ctx, const rk_sketches *s, int hash_bits, uint32_t shard_id, uint32_t n_shards, rk_index **out);

// (a failing build may leave kernels in flight on both of the context's streams that still write into temporaries its DevBufs
// have just returned to the pool: nothing may be handed out again before they are done)
static int settle_streams(rk_ctx *ctx, int rc)
{
    if (rc && ctx) {
        if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        (void)hipGetLastError();
    }
    return rc;
}

int rk_index_build(rk_ctx *ctx, const rk_sketches *s, int hash_bits, rk_index **out)
{
    return settle_streams(ctx, index_build_impl(ctx, s, hash_bits, 0, 1, out));
}

int rk_index_build_shard(rk_ctx *ctx, const rk_sketches *s, int hash_bits, uint32_t shard, uint32_t n_shards, rk_index **out)
{
    if (!ctx) return RK_ERR_ARG;
    if (!n_shards || n_shards > kRecRegions || (n_shards & (n_shards - 1)) || shard >= n_shards)
        return rk_fail(ctx, RK_ERR_ARG, "rk_index_build_shard: %u shards (a power of two up to %u), shard %u", n_shards, kRecRegions, shard);
    return settle_streams(ctx, index_build_impl(ctx, s, hash_bits, shard, n_shards, out));
}

static int index_build_impl(rk_ctx *ctx, const rk_sketches *s, int hash_bits, uint32_t shard_id, uint32_t n_shards, rk_index **out)
{
    if (!ctx || !s || !out) return RK_ERR_ARG;
    *out = nullptr;
    if (hash_bits < 1) return rk_fail(ctx, RK_ERR_ARG, "hash_bits must be positive");
    if (hash_bits > 64) return rk_fail(ctx, RK_ERR_ARG, "hash_bits=%d", hash_bits);
    if ((hash_bits > 32) != s->wide)
        return rk_fail(ctx, RK_ERR_ARG, "hash_bits=%d does not match the sketches' %s-bit layout", hash_bits,
                       s->wide ? "64" : "32");
    const uint64_t H = s->total;
    const uint32_t N = s->n;
    // Bit 31 of a slice record tags its compact form, so posting offsets inside slice records stay below 2^31.  An index of
    // 2^31-1 .. 2^32-2 postings (all of GenBank's bacteria at ~1,200 hashes each) is built WITHOUT slice records: its
    // postings, list offsets and distinct hashes are complete (.dict / .index export, sparse self joins through the tile
    // kernel, which reads the posting lists themselves); what needs slice records -- a dense report, sketches that repeat
    // a hash -- is refused for it.  RK_INDEX_NO_SELF=1 builds any index that way (tests).
    if (H >= 0xFFFFFFFFULL || N >= 0x7FFFFFFFu) return rk_fail(ctx, RK_ERR_UNSUPPORTED, "more than 2^32-2 postings or 2^31-1 genomes");
    const bool no_self = H >= 0x7FFFFFFFULL || ctx->sw_index_no_self;
    if (no_self && !s->is_set)
        return rk_fail(ctx, RK_ERR_UNSUPPORTED, "an index of more than 2^31-1 postings needs set sketches (no hash twice in a genome)");
    RK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    rk_index *idx = new (std::nothrow) rk_index;
    if (!idx) return RK_ERR_NOMEM;
    idx->ctx = ctx;
    idx->n_ref = N;
    idx->H = H;
    idx->hash_bits = hash_bits;
    idx->wide = s->wide;
    idx->max_src_size = idx->max_ref_size = s->max_size;
    for (uint32_t g = 0; g < s->n; g++) {
        const uint64_t sz = s->h_off[g + 1] - s->h_off[g];
        if (sz && (!idx->min_ref_size || sz < idx->min_ref_size)) idx->min_ref_size = sz;
    }
    struct Guard { rk_index *p; ~Guard() { if (p) rk_index_free(p); } } guard{idx};

    // distinct hashes: at most H, at most the hash space
    const uint64_t Ucap = hash_bits < 40 ? std::min<uint64_t>(H, 1ULL << hash_bits) : H;
    RK_TRY(pool_array(ctx, &idx->d_sizes, (size_t)N + 1));
    RK_TRY(pool_array(ctx, &idx->d_src_off, (size_t)N + 1));
    RK_TRY(pool_array(ctx, &idx->d_postings, H + 8));   // (padded: the kernels read up to eight postings from any list start)
    if (idx->wide) RK_TRY(pool_array(ctx, &idx->d_uhash64, Ucap + 1));
    else RK_TRY(pool_array(ctx, &idx->d_uhash, Ucap + 1));
    RK_TRY(pool_array(ctx, &idx->d_upos, Ucap + 2));
    auto alloc_slices = [&]() -> int {   // (not for an index whose build emits tile records: rk_index_ensure_slices, on first use)
        if (no_self || idx->d_selfrange) return RK_OK;
        RK_TRY(pool_array(ctx, &idx->d_selfrange, H + 1));
        RK_TRY(pool_array(ctx, &idx->d_self_off, (size_t)N + 1));
        RK_TRY(pool_array(ctx, &idx->d_self_split, (size_t)N + 1));
        return RK_OK;
    };
    RK_TRY(pool_array(ctx, &idx->d_orig, (size_t)N + 1));
    // ---- which build: the bucket sort (rk_index_fast.inc) when the key fields fit, and then with TILE records as its product
    // (rk_index_tiles.inc) from RK_DIST_TILES_MIN_GENOMES genomes on -- the self join runs on rk_tile_kernel from its first launch --,
    // with slice records (rk_near_kernel, rk_dist_kernel) below
    // (round 5) The hash space is covered in RANGES (its top bits): one range per shard of a multi-GPU build
    // (rk_index_build_shard: this call builds the lists of ITS range only), and inside a shard as many passes as it takes to keep
    // a pass's keys within 2^15 buckets of ~1,536 -- a collection of any size takes the bucket sort, pass after pass on one
    // stream, the postings of a pass behind those of the pass before.
    int shard_bits = 0, pass_bits = 0;
    while ((1u << shard_bits) < n_shards) shard_bits++;
    const uint64_t H_shard = H / n_shards + (n_shards > 1 ? H / (8ULL * n_shards) + 4096 : 0);   // (estimate: the hashes are spread evenly)
    // (a pass may fill its 2^15 buckets to ~2,600 keys on average: the LDS sort holds 4,096, and every extra pass reads all hashes again)
    while (pass_bits < 8 && (H_shard >> pass_bits) > (2600ULL << kMaxBucketBits)) pass_bits++;
    if (getenv("RK_INDEX_PASS_BITS")) pass_bits = std::max(0, std::min(7, atoi(getenv("RK_INDEX_PASS_BITS"))));   // (tests: several passes over a small collection)
    const int range_bits = shard_bits + pass_bits;
    const uint32_t n_pass = 1u << pass_bits;
    const int eff_bits = hash_bits - range_bits;   // hash bits inside a range
    const uint64_t H_pass = range_bits ? (H_shard >> pass_bits) : H;
    int B = 1, gb = 1, rb = 1;
    // (at most kMaxBucketBits: a bigger collection gets fuller buckets, up to the LDS capacity -- beyond it the kernels raise the overflow flag)
    const uint64_t bucket_target = getenv("RK_INDEX_BUCKET_TARGET") ? std::max(64, atoi(getenv("RK_INDEX_BUCKET_TARGET"))) : kBucketTarget;
    while (B < eff_bits && B < kMaxBucketBits && ((H_pass + bucket_target - 1) / bucket_target) > (1ULL << B)) B++;
    while ((1ULL << gb) < N) gb++;
    while ((1ULL << rb) < s->max_size) rb++;
    const int low_bits = eff_bits - B;
    // (64-bit hashes -- use64, e.g. K12 L3: 36 bits -- take the same path as long as the key fields fit: the kernels that read
    // the sketches are templated on the hash type, the bucket sort itself only ever sees the low bits)
    const bool fast_common = ctx->sw_index_fast && H && s->is_set && eff_bits >= 1 && B <= kMaxBucketBits && low_bits >= 0 && low_bits <= 31 && gb <= 31 && rb <= 31;
    const bool slices_ok = fast_common && !range_bits && !no_self && H < (1ULL << 30) && low_bits + gb + rb <= 63;   // (slice records: one pass, offsets below 2^30)
    const uint32_t n_blocks = (N + 31) / 32;
    const bool tiles_ok = fast_common && N >= 2 && n_blocks <= kTileMaxBlocks && low_bits + gb <= 63 && ctx->sw_index_tiles != 0;
    bool tiles_mode = tiles_ok && (range_bits || no_self || ctx->sw_index_tiles == 1 || N >= (uint32_t)ctx->sw_dist_tiles_min_genomes);
    const bool fast_ok = tiles_mode || slices_ok;
    if (n_shards > 1 && !tiles_mode)
        return rk_fail(ctx, RK_ERR_UNSUPPORTED, "rk_index_build_shard needs set sketches of 2 .. %u genomes whose key fields fit the bucket sort "
                                                "(hash bits %d, %u shards)", kTileMaxBlocks * 32, hash_bits, n_shards);
    if (tiles_mode) RK_TRY(pool_array(ctx, &idx->d_blk_min, (size_t)n_blocks));
    const unsigned wave_blocks = (N + 3) / 4;  // 4 waves (genomes) per 256-thread workgroup

    // ---- internal genome order: relatives next to each other (see k_minhash_insert) ---------------------------------
    // d_orig, d_sizes and d_src_off (the CSR offsets in internal order); the rest of the build, and every kernel that
    // uses the index, works in that order
    // (round 4) It runs on a stream of its own: the partition of the hashes below does not need it -- it walks the sketches
    // in the CALLER's order and k_bucket_emit translates the genome ids through `inv` --, so the two overlap (≈ 70 us of
    // small dependent kernels next to ≈ 160 us of partition); `join()` makes ctx->stream wait for it where its results are
    // first used.  The temporaries live until the function returns (the pool's reuse is ordered on ctx->stream only).
    DevBuf<unsigned long long> rl_table(ctx), rl_keys(ctx), rl_keys_sorted(ctx);
    DevBuf<uint32_t> rl_parent(ctx), rl_keys32(ctx), rl_rank(ctx), rl_inv(ctx);
    DevBuf<uint2> rl_tab(ctx);
    DevBuf<char> rl_tmp(ctx);   // the sort's scratch (kept until this function returns: the sort runs on the second stream)
    const uint32_t *inv = nullptr;   // caller's genome index -> internal id (null: identity)
    bool forked = false, joined = true;
    auto join = [&]() -> int {
        if (!joined) {
            RK_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
            joined = true;
        }
        return RK_OK;
    };
    const bool relabel = ctx->sw_index_relabel && s->is_set && N > 1 && H;
    const bool two_streams = relabel && !getenv("RK_INDEX_ONE_STREAM");
    if (two_streams) {
        if (!ctx->stream2) {
            {   // (its kernels are small and many: at the highest priority they are not queued behind the partition's workgroups)
                int prio_lo = 0, prio_hi = 0;
                const bool hi = !getenv("RK_INDEX_STREAM2_PRIO") || atoi(getenv("RK_INDEX_STREAM2_PRIO")) != 0;
                if (hi && hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) == hipSuccess && prio_hi != prio_lo)
                    RK_HIP(ctx, hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, prio_hi));
                else
                    RK_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
            }
            RK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
            RK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
            RK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_inv, hipEventDisableTiming));
        }
        RK_HIP(ctx, hipEventRecord(ctx->ev_fork, st));          // (everything enqueued on ctx->stream so far comes first)
    }
    // The launches of the renumbering are ENQUEUED after the partition's (the fast path calls this once its own first
    // kernels are in the queue): the host needs ~5 us per launch, and with the renumbering's ten launches in front the
    // partition started 70 us late.
    bool inv_recorded = false;   // ctx->ev_inv was recorded behind the kernel that completes `inv`
    bool renumbering_enqueued = false, want_tab = false;   // want_tab: the fast path is taken (H < 2^30) and wants k_emit_table's table
    auto enqueue_renumbering = [&]() -> int {
      if (renumbering_enqueued) return RK_OK;
      renumbering_enqueued = true;
      if (relabel) {
        hipStream_t s2 = st;
        if (two_streams) {
            s2 = ctx->stream2;
            RK_HIP(ctx, hipStreamWaitEvent(s2, ctx->ev_fork, 0));
            forked = true;
        }
        int id_bits = 1;
        while ((1ULL << id_bits) < N) id_bits++;
        uint32_t slots = 1024;
        // (at most half full, usually far less -- relatives share their smallest hashes.  Twice this size was 268 MB at 500,000 genomes:
        // beyond the last-level cache, every probe a DRAM row of its own, and the kernels streaming beside it slowed to a third)
        const unsigned long long want_slots = (getenv("RK_INDEX_TABLE_X") ? strtoull(getenv("RK_INDEX_TABLE_X"), nullptr, 10) : 2ULL) * N * kMinK;
        while (slots < want_slots && slots < (1u << 30)) slots <<= 1;
        RK_HIP(ctx, rl_table.alloc(slots));
        RK_HIP(ctx, rl_parent.alloc(N));
        RK_HIP(ctx, hipMemsetAsync(rl_table.p, 0xFF, (size_t)slots * 8, s2));
        const unsigned nb_ins = (unsigned)(((uint64_t)N * kMinK + kInsertThreads - 1) / kInsertThreads), nb_n = blocks_for(N);
        if (idx->wide) {
            hipLaunchKernelGGL(k_minhash_insert<uint64_t>, dim3(nb_ins), dim3(kInsertThreads), 0, s2, s->d_hashes64, s->d_off, N, rl_table.p, slots - 1);
            hipLaunchKernelGGL(k_minhash_vote<uint64_t>, dim3(nb_n), dim3(kThreads), 0, s2, s->d_hashes64, s->d_off, N, rl_table.p, slots - 1, rl_parent.p);
        } else {
            hipLaunchKernelGGL(k_minhash_insert<uint32_t>, dim3(nb_ins), dim3(kInsertThreads), 0, s2, s->d_hashes, s->d_off, N, rl_table.p, slots - 1);
            hipLaunchKernelGGL(k_minhash_vote<uint32_t>, dim3(nb_n), dim3(kThreads), 0, s2, s->d_hashes, s->d_off, N, rl_table.p, slots - 1, rl_parent.p);
        }
        if (N <= kRankMaxN) {
            RK_HIP(ctx, rl_keys32.alloc(N));
            RK_HIP(ctx, rl_rank.alloc(N));
            hipLaunchKernelGGL(k_cluster_keys32, dim3(nb_n), dim3(kThreads), 0, s2, rl_parent.p, N, id_bits, rl_keys32.p, rl_rank.p);
            hipLaunchKernelGGL(k_rank_keys, dim3((N + kRankQ - 1) / kRankQ, (N + kRankStretch - 1) / kRankStretch), dim3(kRankThreads), 0, s2,
                               rl_keys32.p, N, rl_rank.p);
            if (forked) {   // (a genome's rank among the keys IS its internal id: all the bucket emission of tile records needs)
                RK_HIP(ctx, hipEventRecord(ctx->ev_inv, s2));
                inv_recorded = true;
            }
            hipLaunchKernelGGL(k_order_from_rank, dim3(nb_n), dim3(kThreads), 0, s2, rl_rank.p, N, s->d_off, idx->d_orig, idx->d_sizes);
            hipLaunchKernelGGL(k_offsets_scan, dim3(1), dim3(1024), 0, s2, idx->d_sizes, N, idx->d_src_off);
            inv = rl_rank.p;   // (a genome's rank among the keys IS its internal id)
        } else {
            RK_HIP(ctx, rl_keys.alloc(N));
            RK_HIP(ctx, rl_keys_sorted.alloc(N));
            RK_HIP(ctx, rl_inv.alloc(N));
            hipLaunchKernelGGL(k_cluster_keys, dim3(nb_n), dim3(kThreads), 0, s2, rl_parent.p, N, id_bits, rl_keys.p);
            void *scratch = nullptr;
            RK_TRY(rk_prim_sort_keys_u64(ctx, rl_keys.p, rl_keys_sorted.p, N, 0, (unsigned)(2 * id_bits), s2, &scratch));
            rl_tmp.p = static_cast<char *>(scratch);
            hipLaunchKernelGGL(k_order_from_keys, dim3(nb_n), dim3(kThreads), 0, s2, rl_keys_sorted.p, N, id_bits, s->d_off, idx->d_orig, idx->d_sizes);
            hipLaunchKernelGGL(k_offsets_scan, dim3(1), dim3(1024), 0, s2, idx->d_sizes, N, idx->d_src_off);
            hipLaunchKernelGGL(k_invert_order, dim3(nb_n), dim3(kThreads), 0, s2, idx->d_orig, N, rl_inv.p);
            inv = rl_inv.p;
        }
        if (want_tab) {
            RK_HIP(ctx, rl_tab.alloc(N));
            hipLaunchKernelGGL(k_emit_table, dim3(nb_n), dim3(kThreads), 0, s2, inv, idx->d_src_off, N, rl_tab.p);
        }
        if (idx->d_blk_min) hipLaunchKernelGGL(k_blk_min_sizes, dim3(blocks_for(n_blocks)), dim3(kThreads), 0, s2, idx->d_sizes, N, n_blocks, idx->d_blk_min);
        idx->relabeled = true;
        RK_HIP(ctx, hipGetLastError());
        if (forked) {
            RK_HIP(ctx, hipEventRecord(ctx->ev_join, s2));
            joined = false;
        }
      } else {
        hipLaunchKernelGGL(k_iota, dim3(blocks_for((uint64_t)N + 1)), dim3(kThreads), 0, st, N, idx->d_orig);
        hipLaunchKernelGGL(k_sizes, dim3(blocks_for((uint64_t)N + 1)), dim3(kThreads), 0, st, s->d_off, N, idx->d_sizes, idx->d_src_off);
        if (idx->d_blk_min) hipLaunchKernelGGL(k_blk_min_sizes, dim3(blocks_for(n_blocks)), dim3(kThreads), 0, st, idx->d_sizes, N, n_blocks, idx->d_blk_min);
      }
      return RK_OK;
    };

    DevBuf<BuildResult> res(ctx);
    RK_HIP(ctx, res.alloc(1));
    BuildResult r{0, 0, 0, 0, 0};
    bool built = false;

    // ---- fast path: two-level bucket sort, second level and all emission in LDS (rk_index_fast.inc) -----------------
    if (ctx->sw_dist_debug && !fast_ok)
        fprintf(stderr, "[rk] index build: general path (H %llu, wide %d, sets %d, B %d, low bits %d, genome bits %d, position bits %d)\n",
                (unsigned long long)H, (int)idx->wide, (int)s->is_set, B, low_bits, gb, rb);
    TileResult tr;
    memset(&tr, 0, sizeof tr);
    bool fast_refused = false;
    DevBuf<uint2> t_contrib(ctx);
    DevBuf<uint32_t> t_rows(ctx), t_cols(ctx);
    DevBuf<uint4> t_dir_j(ctx), t_dir_c(ctx);
    DevBuf<uint3> t_rec(ctx);   // a shard's tile records, grouped by destination shard (they leave for the exchange unsorted)
    uint32_t t_region_cap = 0;
    uint64_t rec_cap_retry = 0;   // tile records the first attempt asked for, had they fit
    uint64_t keys_cap_retry = 0;  // keys of the fullest range pass, had they fit (the ranges of a real hash space are not equally full)
    for (int attempt = 0; fast_ok && !built && attempt < 3; attempt++) {
        // (the tile records of an attempt did not fit their buffer -- wide species, lists scattered over many blocks --: the
        // attempt has counted what it needs, and the next one gets exactly that, within a budget of 6 records per posting; beyond
        // it the index is built with slice records after all, where those can be had)
        if (attempt >= 1 && (fast_refused || !tr.overflow)) break;
        if (attempt == 1) {
            unsigned long long worst = 0;
            for (uint32_t q = 0; q < kRecRegions; q++) worst = std::max<unsigned long long>(worst, tr.rec_count[q]);
            rec_cap_retry = (worst + worst / 16 + 1024) * kRecRegions;
            if (rec_cap_retry > 6 * (H / n_shards) + (1u << 22) || ctx->sw_tile_rec_cap) rec_cap_retry = 0;   // (RK_TILE_REC_CAP: a test forces the fallback)
        }
        if (attempt == 2 || (attempt == 1 && !rec_cap_retry)) {
            if (!slices_ok) break;
            if (!tiles_mode) break;
            tiles_mode = false;
            rk_pool_free(ctx, idx->d_blk_min);
            idx->d_blk_min = nullptr;
        }
        if (!tiles_mode) RK_TRY(alloc_slices());
        const uint32_t passes = tiles_mode ? n_pass : 1;
        FastArgs fa;
        fa.hashes = idx->wide ? (const void *)s->d_hashes64 : (const void *)s->d_hashes;
        fa.off = s->d_off;
        fa.orig = nullptr;           // the partition walks the sketches in the caller's order (see the renumbering above)
        fa.off_new = s->d_off;
        fa.n_genomes = N;
        fa.H = H;
        fa.hash_bits = hash_bits;
        fa.low_bits = low_bits;
        fa.gb = gb;
        fa.rb = tiles_mode ? 0 : rb;   // (tile records: nobody needs an element's position inside its sketch)
        fa.xcd_map = getenv("RK_INDEX_XCD") ? atoi(getenv("RK_INDEX_XCD")) : 1;
        fa.nb = 1u << B;
        fa.n_chunks = (uint32_t)((H + kPartChunk - 1) / kPartChunk);
        fa.range_bits = tiles_mode ? range_bits : 0;
        fa.range_id = 0;
        // what a pass may hold: exactly H without ranges; with ranges an estimate + slack (a pass that exceeds it raises the overflow
        // flag in k_part_starts and the kernels behind it stand still)
        // (RK_INDEX_KEYS_CAP_PCT: tests make the estimate too small)
        const uint64_t keys_pct = getenv("RK_INDEX_KEYS_CAP_PCT") ? std::max(1, atoi(getenv("RK_INDEX_KEYS_CAP_PCT"))) : 125;
        const uint64_t keys_cap = fa.range_bits ? std::min<uint64_t>(H, keys_cap_retry ? keys_cap_retry : H_pass * keys_pct / 100 + (keys_pct >= 100 ? (1u << 20) : 0)) : H;
        fa.keys_cap = keys_cap;
        fa.filtered = nullptr;
        fa.n_filtered = nullptr;
        DevBuf<uint32_t> chunk_first(ctx), matrix(ctx), total(ctx), bstart(ctx), ucount(ctx), ubase(ctx), tmp_uhash(ctx), tmp_upos(ctx), n_open(ctx), n_cov(ctx);
        DevBuf<unsigned long long> keys(ctx), tmp_uhash64(ctx), zeroed(ctx);
        DevBuf<uint2> self_raw(ctx);
        // tile records: unsorted (64 regions), binned by row block, the directory's proto entries
        DevBuf<uint32_t> bins(ctx), tb(ctx), big_list(ctx);
        DevBuf<uint3> brec(ctx);
        DevBuf<uint4> proto(ctx);
        DevBuf<unsigned long long> level_start(ctx);
        const bool wide = idx->wide;
        const size_t nb1 = (size_t)fa.nb + 1;
        // the two-pass partition needs six spare key bits and >= 128 buckets
        const bool part2 = (getenv("RK_INDEX_PART2") ? atoi(getenv("RK_INDEX_PART2")) != 0 : true) && B >= 7 && low_bits + gb + fa.rb <= 64 - (int)kFineBits;
        // a range pass partitions what k_range_filter kept of the hashes (RK_INDEX_FILTER=0: every kernel of the pass walks them all)
        const bool use_filter = fa.range_bits && part2 && eff_bits + gb <= 64 && (getenv("RK_INDEX_FILTER") ? atoi(getenv("RK_INDEX_FILTER")) != 0 : true);
        const uint32_t part_chunks = use_filter ? (uint32_t)((keys_cap + kPartChunk - 1) / kPartChunk) : fa.n_chunks;   // rows of the count matrix
        RK_HIP(ctx, chunk_first.alloc((size_t)fa.n_chunks + 1));
        RK_HIP(ctx, matrix.alloc((size_t)part_chunks * fa.nb));
        RK_HIP(ctx, total.alloc(fa.nb));
        RK_HIP(ctx, bstart.alloc(nb1 * passes));   // (per pass: the list heads of a pass are placed while the next one partitions)
        RK_HIP(ctx, ucount.alloc(nb1 * passes));
        RK_HIP(ctx, ubase.alloc(nb1 * passes));
        if (wide) RK_HIP(ctx, tmp_uhash64.alloc(H));
        else RK_HIP(ctx, tmp_uhash.alloc(H));
        RK_HIP(ctx, tmp_upos.alloc(H));
        RK_HIP(ctx, keys.alloc(keys_cap));
        uint64_t rec_cap = 0, tile_cap = 0, slot_cap = 0;
        uint32_t region_cap = 0;
        const bool sort_here = tiles_mode && n_shards == 1;   // (a shard's records leave for the exchange: rk_index_join_shard sorts what arrives)
        if (tiles_mode) {
            // related lists write ~0.15-0.3 records per posting; chance collisions of a crowded hash space add H x lambda / 2
            // (lambda = postings per hash value: 500,000 genomes in 28 bits share every value twice over)
            const double lambda = hash_bits < 48 ? (double)H / (double)(1ULL << hash_bits) : 0.0;
            rec_cap = ctx->sw_tile_rec_cap ? ctx->sw_tile_rec_cap : rec_cap_retry ? rec_cap_retry : (uint64_t)((double)(H / n_shards) * (0.5 + 0.6 * lambda)) + 65536;
            rec_cap = std::min<uint64_t>(rec_cap, 0x7FFF0000ULL);
            region_cap = (uint32_t)((rec_cap + kRecRegions - 1) / kRecRegions);
            rec_cap = (uint64_t)region_cap * kRecRegions;
            RK_HIP(ctx, t_rec.alloc(rec_cap));
            t_region_cap = region_cap;
        }
        if (sort_here) {
            tile_cap = std::min<uint64_t>(rec_cap, (uint64_t)n_blocks * (n_blocks + 1) / 2);
            slot_cap = rec_cap + tile_cap;   // (every tile from an even slot on)
            RK_HIP(ctx, brec.alloc(rec_cap));
            RK_HIP(ctx, tb.alloc(3 * (size_t)n_blocks));
            RK_HIP(ctx, bins.alloc((size_t)n_blocks + 1));   // bin starts (the counts and cursors are in `zeroed`)
            RK_HIP(ctx, proto.alloc(2 * tile_cap));
            RK_HIP(ctx, level_start.alloc(2 * (kTileTable + 1)));
            RK_HIP(ctx, t_contrib.alloc(slot_cap + 256));
            if (rowsort_stage(n_blocks)) {   // (the split copy serves the scalar-row variant of the tile kernel: short launches over small collections)
                RK_HIP(ctx, t_rows.alloc(slot_cap + 256));
                RK_HIP(ctx, t_cols.alloc(slot_cap + 256));
            }
            RK_HIP(ctx, t_dir_j.alloc(2 * tile_cap));
            RK_HIP(ctx, t_dir_c.alloc(2 * tile_cap));
        }
        if (!tiles_mode) {
            RK_HIP(ctx, self_raw.alloc(H));
            RK_HIP(ctx, n_open.alloc((size_t)N + 1));
            RK_HIP(ctx, n_cov.alloc((size_t)N + 1));
        }
        // everything the kernels expect zeroed, in one buffer and one fill (each fill is ~5 us on the stream): the result
        // records, where the postings of each pass start, the cursors of the two-pass partition (per pass) and of the tile sort
        const bool small_wgs = (fa.nb >> kFineBits) <= 128;   // several workgroups per chunk: they share its stretch through counters
        const size_t w_cursor = part2 ? (fa.nb + 1) / 2 : 0, w_taken = part2 && small_wgs ? ((size_t)fa.n_chunks * (fa.nb >> kFineBits) + 1) / 2 : 0;
        const size_t z_res = 0, z_tres = z_res + (sizeof(BuildResult) + 7) / 8, z_pass = z_tres + (sizeof(TileResult) + 7) / 8,
                     z_big = z_pass + passes + 1, z_filt = z_big + (passes + 1) / 2, z_hq = z_filt + passes, z_cursor = z_hq + (passes + 1) / 2, z_taken = z_cursor + w_cursor * passes, z_tcur = z_taken + w_taken * passes,
                     z_bins = z_tcur + (sort_here ? (sizeof(TileCursors) + 7) / 8 : 0),
                     z_end = z_bins + (sort_here ? (size_t)n_blocks + 1 : 0);   // bin counts (u32[n_blocks + 1]) + bin cursors (u32[n_blocks])
        RK_HIP(ctx, zeroed.alloc(z_end));   // (zeroed by k_chunk_first, the first launch)
        BuildResult *const fres = reinterpret_cast<BuildResult *>(zeroed.p + z_res);
        TileResult *const tres = reinterpret_cast<TileResult *>(zeroed.p + z_tres);
        unsigned long long *const pass_base = zeroed.p + z_pass;   // [passes + 1]: postings before pass p; the last one = all of them
        const size_t part_lds = (size_t)fa.nb * 4 + 2 * kStageGenomes * 8;   // bucket counters + the chunk's genome bounds
        if (part_lds > 48 * 1024) {
            RK_HIP(ctx, hipFuncSetAttribute((const void *)k_part_hist<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)part_lds));
            RK_HIP(ctx, hipFuncSetAttribute((const void *)k_part_scatter<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)part_lds));
            RK_HIP(ctx, hipFuncSetAttribute((const void *)k_part_hist<uint64_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)part_lds));
            RK_HIP(ctx, hipFuncSetAttribute((const void *)k_part_scatter<uint64_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)part_lds));
        }
        hipLaunchKernelGGL(k_chunk_first, dim3(blocks_for(std::max<uint64_t>((uint64_t)fa.n_chunks + 1, z_end))), dim3(kThreads), 0, st, s->d_off, N, fa.n_chunks,
                           chunk_first.p, zeroed.p, (uint32_t)z_end);
        DevBuf<unsigned long long> mid(ctx);
        if (part2) RK_HIP(ctx, mid.alloc(keys_cap));
        const int emit_t = getenv("RK_INDEX_EMIT_T") ? atoi(getenv("RK_INDEX_EMIT_T")) : 512;
        const bool narrow = low_bits + gb <= 32;  // (hash_low, genome) fits 32 bits
        // buckets beyond the LDS sort (a hash shared by thousands of genomes) go to k_bucket_heavy: tile records, 32-bit sort keys
        const bool big_ok = tiles_mode && narrow && ctx->sw_index_heavy;
        if (big_ok) RK_HIP(ctx, big_list.alloc((size_t)fa.nb * passes));
        const size_t heavy_lds = heavy_lds_bytes(n_blocks);
        if (big_ok) RK_HIP(ctx, hipFuncSetAttribute((const void *)k_bucket_heavy<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)heavy_lds));
        for (uint32_t pass = 0; pass < passes; pass++) {
            fa.range_id = (shard_id << pass_bits) | pass;
            uint32_t *const bstart_p = bstart.p + nb1 * pass, *const ucount_p = ucount.p + nb1 * pass, *const ubase_p = ubase.p + nb1 * pass;
            uint32_t *const fine_cursor = reinterpret_cast<uint32_t *>(zeroed.p + z_cursor + w_cursor * pass);
            uint32_t *const seg_taken = reinterpret_cast<uint32_t *>(zeroed.p + z_taken + w_taken * pass);
            FastArgs pa = fa;   // what the partition kernels of this pass see
            if (use_filter) {
                unsigned long long *const n_filt = zeroed.p + z_filt + pass;
                // (the filtered elements lie in `keys`: the coarse pass reads them and writes `mid`, the fine pass writes `keys` again)
                if (wide) hipLaunchKernelGGL(k_range_filter<uint64_t>, dim3(fa.n_chunks * kFilterSplit), dim3(kFilterThreads), 0, st, fa, chunk_first.p, keys.p, n_filt, fres);
                else hipLaunchKernelGGL(k_range_filter<uint32_t>, dim3(fa.n_chunks * kFilterSplit), dim3(kFilterThreads), 0, st, fa, chunk_first.p, keys.p, n_filt, fres);
                // (tried: starting the renumbering BEHIND the filter -- then k_part_hist takes 0.84 ms instead of 0.16 beside
                // k_minhash_insert: whatever runs beside that kernel pays its 0.55 ms)
                pa.filtered = keys.p;
                pa.n_filtered = n_filt;
                pa.hash_bits = eff_bits;
                pa.range_bits = 0;
                pa.range_id = 0;
            }
            if (wide) hipLaunchKernelGGL(k_part_hist<uint64_t>, dim3(part_chunks), dim3(kPartThreads), part_lds, st, pa, chunk_first.p, matrix.p, fres);
            else hipLaunchKernelGGL(k_part_hist<uint32_t>, dim3(part_chunks), dim3(kPartThreads), part_lds, st, pa, chunk_first.p, matrix.p, fres);
            hipLaunchKernelGGL(k_part_colscan, dim3((fa.nb + 63) / 64), dim3(1024), 0, st, matrix.p, part_chunks, fa.nb, total.p);
            uint32_t *const n_big_p = reinterpret_cast<uint32_t *>(zeroed.p + z_big) + pass;
            hipLaunchKernelGGL(k_part_starts, dim3(1), dim3(1024), 0, st, total.p, fa.nb, bstart_p, fres, pass_base + pass, (unsigned long long)keys_cap,
                               big_ok ? big_list.p + (size_t)fa.nb * pass : nullptr, n_big_p);
            // the partition itself: two coalescing passes (rk_index_fast.inc), or one scattering pass
            if (part2) {
#define RK_COARSE(TT, HT, GRID)                                                                                                              \
    do {                                                                                                                                     \
        RK_HIP(ctx, hipFuncSetAttribute((const void *)k_part_coarse<TT, HT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)part2_lds(TT))); \
        hipLaunchKernelGGL((k_part_coarse<TT, HT>), dim3(GRID), dim3(TT), part2_lds(TT), st, pa, chunk_first.p, matrix.p, bstart_p, seg_taken, mid.p, fres); \
    } while (0)
                if (small_wgs) { if (wide) RK_COARSE(256, uint64_t, part_chunks * 4); else RK_COARSE(256, uint32_t, part_chunks * 4); }
                else { if (wide) RK_COARSE(1024, uint64_t, part_chunks); else RK_COARSE(1024, uint32_t, part_chunks); }
#undef RK_COARSE
                hipLaunchKernelGGL(k_part_fine, dim3(fa.nb >> kFineBits, 16), dim3(kPartThreads), 0, st, pa, bstart_p, mid.p, keys.p, fine_cursor, fres);
            } else if (wide) {
                hipLaunchKernelGGL(k_part_scatter<uint64_t>, dim3(fa.n_chunks), dim3(kPartThreads), part_lds, st, fa, chunk_first.p, matrix.p, bstart_p, keys.p, fres);
            } else {
                hipLaunchKernelGGL(k_part_scatter<uint32_t>, dim3(fa.n_chunks), dim3(kPartThreads), part_lds, st, fa, chunk_first.p, matrix.p, bstart_p, keys.p, fres);
            }
            if (pass == 0) {
                want_tab = !tiles_mode;
                RK_TRY(enqueue_renumbering());   // (behind the partition's launches in the host's queue, beside them on the device)
                // the internal order is needed from here on: the translation table alone for tile records (sizes and offsets in internal
                // order are still on their way on the second stream: joined in front of the row sort), everything for slice records
                if (tiles_mode && inv_recorded && !joined) RK_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_inv, 0));
                else RK_TRY(join());
                if (!tiles_mode && inv && !rl_tab.p) {   // (second attempt: the renumbering ran for tile records, without the table)
                    RK_HIP(ctx, rl_tab.alloc(N));
                    hipLaunchKernelGGL(k_emit_table, dim3(blocks_for(N)), dim3(kThreads), 0, st, inv, idx->d_src_off, N, rl_tab.p);
                }
            }
            if (tiles_mode) {
                TileEmitArgs ea;
                ea.keys = keys.p;
                ea.bstart = bstart_p;
                ea.inv = inv;
                ea.low_bits = low_bits;
                ea.gb = gb;
                ea.rb = 0;
                ea.nb = fa.nb;
                ea.postings = idx->d_postings;
                ea.tmp_uhash = tmp_uhash.p;
                ea.tmp_uhash64 = wide ? tmp_uhash64.p : nullptr;
                ea.tmp_upos = tmp_upos.p;
                ea.ucount = ucount_p;
                ea.rec = t_rec.p;
                ea.region_cap = region_cap;
                ea.pass_base = pass_base + pass;
                ea.hash_base = fa.range_bits ? ((unsigned long long)fa.range_id << eff_bits) : 0ULL;
                ea.n_dest = n_shards;
                ea.stop = fa.range_bits ? fres : nullptr;
                ea.big_ok = big_ok ? 1 : 0;
                ea.big_list = big_ok ? big_list.p + (size_t)fa.nb * pass : nullptr;
                ea.n_big = n_big_p;
                ea.stop_rw = fres;
                ea.tres = tres;
                ea.xcd_map = fa.xcd_map;
                ea.debug = getenv("RK_INDEX_DEBUG") ? atoi(getenv("RK_INDEX_DEBUG")) : 0;
                if (ea.debug) {  // developer ablations leave stages out: whatever they do not write must still be harmless downstream
                    RK_HIP(ctx, hipMemsetAsync(ucount_p, 0, (size_t)fa.nb * 4, st));
                    RK_HIP(ctx, hipMemsetAsync(tmp_upos.p, 0, H * 4, st));
                }
#define RK_EMIT(TT) do { if (narrow) hipLaunchKernelGGL((k_bucket_emit_tiles<TT, uint32_t>), dim3(fa.nb), dim3(TT), 0, st, ea); \
                         else hipLaunchKernelGGL((k_bucket_emit_tiles<TT, unsigned long long>), dim3(fa.nb), dim3(TT), 0, st, ea); } while (0)
                if (emit_t == 256) RK_EMIT(256);
                else if (emit_t == 1024) RK_EMIT(1024);
                else RK_EMIT(512);
#undef RK_EMIT
                if (big_ok) {
                    HeavyArgs ha;
                    ha.e = ea;
                    ha.big_list = big_list.p + (size_t)fa.nb * pass;
                    ha.n_big = n_big_p;
                    ha.next = reinterpret_cast<uint32_t *>(zeroed.p + z_hq) + pass;
                    ha.n_blocks = n_blocks;
                    hipLaunchKernelGGL(k_bucket_heavy<1024>, dim3((unsigned)std::max(1, ctx->num_cu)), dim3(1024), heavy_lds, st, ha);
                }
            } else {
                EmitArgs ea;
                ea.keys = keys.p;
                ea.bstart = bstart_p;
                ea.off_new = idx->d_src_off;
                ea.inv = inv;
                ea.tab = inv ? rl_tab.p : nullptr;
                ea.low_bits = low_bits;
                ea.gb = gb;
                ea.rb = rb;
                ea.nb = fa.nb;
                ea.postings = idx->d_postings;
                ea.tmp_uhash = tmp_uhash.p;
                ea.tmp_uhash64 = wide ? tmp_uhash64.p : nullptr;
                ea.tmp_upos = tmp_upos.p;
                ea.ucount = ucount_p;
                ea.self_raw = self_raw.p;
                ea.res = fres;
                ea.xcd_map = fa.xcd_map;
                ea.debug = getenv("RK_INDEX_DEBUG") ? atoi(getenv("RK_INDEX_DEBUG")) : 0;
                if (ea.debug) {  // developer ablations leave stages out: whatever they do not write must still be harmless downstream
                    RK_HIP(ctx, hipMemsetAsync(ucount_p, 0, (size_t)fa.nb * 4, st));
                    RK_HIP(ctx, hipMemsetAsync(self_raw.p, 0, H * sizeof(uint2), st));
                    RK_HIP(ctx, hipMemsetAsync(tmp_upos.p, 0, H * 4, st));
                }
#define RK_EMIT(TT) do { if (narrow) hipLaunchKernelGGL((k_bucket_emit<TT, uint32_t>), dim3(fa.nb), dim3(TT), 0, st, ea); \
                         else hipLaunchKernelGGL((k_bucket_emit<TT, unsigned long long>), dim3(fa.nb), dim3(TT), 0, st, ea); } while (0)
                if (emit_t == 256) RK_EMIT(256);
                else if (emit_t == 1024) RK_EMIT(1024);
                else RK_EMIT(512);
#undef RK_EMIT
            }
            // the list heads (scan + placement) hang on the emission alone: they go to the second stream (one pass) and run beside
            // the rows / the tile sort; with several passes they stay in line (the next pass's partition is the bigger job)
            hipStream_t sh = st;
            const bool heads_aside = forked && passes == 1;
            if (heads_aside) {
                RK_HIP(ctx, hipEventRecord(ctx->ev_fork, st));
                RK_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
                sh = ctx->stream2;
            }
            hipLaunchKernelGGL(k_heads_scan, dim3(1), dim3(1024), 0, sh, ucount_p, fa.nb, ubase_p, fres);
            if (wide)
                hipLaunchKernelGGL(k_heads_place<unsigned long long>, dim3(fa.nb), dim3(kThreads), 0, sh, tmp_uhash64.p, tmp_upos.p, bstart_p, ucount_p, ubase_p,
                                   fa.nb, pass_base + pass, (unsigned long long *)idx->d_uhash64, idx->d_upos);
            else
                hipLaunchKernelGGL(k_heads_place<uint32_t>, dim3(fa.nb), dim3(kThreads), 0, sh, tmp_uhash.p, tmp_upos.p, bstart_p, ucount_p, ubase_p, fa.nb,
                                   pass_base + pass, idx->d_uhash, idx->d_upos);
            if (heads_aside) {
                RK_HIP(ctx, hipEventRecord(ctx->ev_join, sh));
                joined = false;
            }
        }
        TileSortArgs ta;
        memset(&ta, 0, sizeof ta);
        if (sort_here) {
            ta.rec = t_rec.p;
            ta.cur = reinterpret_cast<TileCursors *>(zeroed.p + z_tcur);
            ta.region_cap = region_cap;
            ta.n_blocks = n_blocks;
            ta.bin_count = reinterpret_cast<uint32_t *>(zeroed.p + z_bins);
            ta.bin_cursor = ta.bin_count + n_blocks + 1;
            ta.bin_start = bins.p;
            ta.brec = brec.p;
            ta.blk_min = idx->d_blk_min;
            ta.contrib = t_contrib.p;
            ta.rows = t_rows.p;
            ta.cols = t_cols.p;
            ta.tb_base = tb.p;
            ta.tb_cnt = tb.p + n_blocks;
            ta.order = tb.p + 2 * (size_t)n_blocks;
            ta.proto = proto.p;
            ta.level_start = level_start.p;
            ta.dir[0] = t_dir_j.p;
            ta.dir[1] = t_dir_c.p;
            ta.tres = tres;
            RK_TRY(launch_tile_sort(ctx, ta, st, [&]() { return join(); }));
        } else if (!tiles_mode) {
            hipLaunchKernelGGL(k_row_counts2, dim3(wave_blocks), dim3(kThreads), 0, st, idx->d_src_off, N, self_raw.p, n_open.p, n_cov.p);
            hipLaunchKernelGGL(k_row_scan, dim3(1), dim3(1024), 0, st, n_open.p, n_cov.p, N, idx->d_self_off, idx->d_self_split, fres);
            hipLaunchKernelGGL(k_row_place2, dim3(wave_blocks), dim3(kThreads), 0, st, idx->d_src_off, N, self_raw.p, idx->d_self_off,
                               idx->d_self_split, idx->d_selfrange);
        }
        RK_HIP(ctx, hipGetLastError());
        RK_TRY(join());
        unsigned long long n_postings = 0;
        {   // the one synchronisation of the build: both result records (and the postings of all passes) in one read-back
            struct { BuildResult r; TileResult t; unsigned long long pass_base[257]; } both;
            static_assert(sizeof(BuildResult) % 8 == 0 && offsetof(decltype(both), pass_base) == sizeof(BuildResult) + sizeof(TileResult), "the records lie back to back");
            RK_TRY(rk_read_back(ctx, &both, fres, sizeof(BuildResult) + sizeof(TileResult) + ((size_t)passes + 1) * 8, st));
            r = both.r;
            tr = both.t;
            n_postings = both.pass_base[passes];
        }
        if (ctx->sw_dist_debug && big_ok) {   // (developer output: what k_bucket_heavy had to take)
            std::vector<uint32_t> nbig(passes), bs((size_t)nb1 * passes);
            RK_TRY(rk_read_back(ctx, nbig.data(), zeroed.p + z_big, (size_t)passes * 4, st));
            RK_TRY(rk_read_back(ctx, bs.data(), bstart.p, bs.size() * 4, st));
            for (uint32_t pass = 0; pass < passes; pass++) {
                std::vector<uint32_t> bl(nbig[pass]);
                if (nbig[pass]) RK_TRY(rk_read_back(ctx, bl.data(), big_list.p + (size_t)fa.nb * pass, bl.size() * 4, st));
                unsigned long long keys_in = 0, biggest = 0;
                for (uint32_t b : bl) {
                    const unsigned long long n = bs[nb1 * pass + b + 1] - bs[nb1 * pass + b];
                    keys_in += n;
                    biggest = std::max(biggest, n);
                }
                fprintf(stderr, "[rk] index build pass %u: %u of %u buckets for k_bucket_heavy, %llu keys, biggest %llu\n", pass, nbig[pass], fa.nb, keys_in, biggest);
            }
        }
        if (ctx->sw_dist_debug)
            fprintf(stderr, "[rk] index build: fast path flags %llu (B %d, low bits %d, genome bits %d, position bits %d; shard %u of %u, %u pass(es), %llu postings)%s\n",
                    r.flags, B, low_bits, gb, rb, shard_id, n_shards, passes, n_postings, tiles_mode ? (tr.overflow ? ", tile records overflowed" : ", tile records") : "");
        if ((r.flags & kFastOverflow) && use_filter && !keys_cap_retry) {
            // a range holds more keys than estimated: every pass's filter has counted what it needs -- once more with exactly that
            std::vector<unsigned long long> asked(passes);
            RK_TRY(rk_read_back(ctx, asked.data(), zeroed.p + z_filt, (size_t)passes * 8, st));
            const unsigned long long need = *std::max_element(asked.begin(), asked.end());
            if (need > keys_cap && need <= H) {
                keys_cap_retry = need + need / 64 + 65536;
                if (ctx->sw_dist_debug) fprintf(stderr, "[rk] index build: a range pass holds %llu keys (buffers for %llu): again\n", need, (unsigned long long)keys_cap);
                memset(&tr, 0, sizeof tr);
                attempt--;
                continue;
            }
        }
        if (r.flags == 0 && !(tiles_mode && tr.overflow)) built = true;
        else if (r.flags) {   // a bucket beyond the LDS sort, a pass beyond its key buffer, or a hash outside the hash space: the general path decides
            fast_refused = true;
            r = BuildResult{0, 0, 0, 0, 0};
        }
        if (built && range_bits) idx->H = n_postings;   // (a shard: the postings of ITS hash range; all passes of one shard: == H)
        if (built && sort_here) {
            idx->d_tile_contrib = t_contrib.release();
            idx->d_tile_rows = t_rows.release();
            idx->d_tile_cols = t_cols.release();
            idx->d_tile_dir[0] = t_dir_j.release();
            idx->d_tile_dir[1] = t_dir_c.release();
            idx->n_tile_slots = tr.n_slots;
            idx->n_tiles = tr.n_tiles;
            idx->n_tile_records = tr.n_records;
            idx->tile_max_records = tr.max_records;
            for (int m = 0; m < 2; m++)
                for (int k = 0; k < kTileTable; k++) idx->tile_prefix[m][k] = tr.level_count[m][k];
            idx->tiles_ready = true;
            idx->tiles_from_build = true;
            t_rec.reset();
            if (ctx->sw_dist_debug)
                fprintf(stderr, "[rk] tiles from the build: %llu tiles, %llu records in %llu slots (capacity %llu), biggest tile %llu\n", tr.n_tiles, tr.n_records,
                        tr.n_slots, (unsigned long long)rec_cap, tr.max_records);
        }
        if (built && tiles_mode && n_shards > 1) {   // the shard's records wait for the exchange (rk_index_shard_records / _pack)
            idx->d_shard_rec = t_rec.release();
            idx->shard_region_cap = t_region_cap;
            idx->n_shards = n_shards;
            idx->shard_id = shard_id;
            for (uint32_t q = 0; q < kRecRegions; q++) idx->shard_rec_count[q] = tr.rec_count[q];
        }
    }
    if (!built && n_shards > 1)
        return rk_fail(ctx, RK_ERR_UNSUPPORTED, "rk_index_build_shard: the bucket sort refused this collection (flags %s: a bucket or a pass beyond its "
                                                "buffer, a hash outside the hash space, or tile records beyond their capacity)", tr.overflow ? "tile overflow" : "bucket / key overflow");
    if (!built) {   // (no tile records after all)
        t_contrib.reset();
        t_rows.reset();
        t_cols.reset();
        t_dir_j.reset();
        t_dir_c.reset();
        if (idx->d_blk_min) { rk_pool_free(ctx, idx->d_blk_min); idx->d_blk_min = nullptr; }
        RK_TRY(alloc_slices());
    }

    RK_TRY(enqueue_renumbering());
    RK_TRY(join());
    if (!built && H) {
        RK_HIP(ctx, hipMemsetAsync(res.p, 0, sizeof(BuildResult), st));
        // ---- general path: device-wide stable radix sort of (hash, source element) -------------------------------------
        const uint32_t *src_hashes = s->d_hashes;
        const uint64_t *src_hashes64 = s->d_hashes64;
        const uint64_t *src_off = s->d_off;
        DevBuf<uint32_t> perm_hashes(ctx);
        DevBuf<uint64_t> perm_hashes64(ctx);
        if (idx->relabeled) {  // the sketches in internal order
            if (idx->wide) {
                RK_HIP(ctx, perm_hashes64.alloc(H));
                hipLaunchKernelGGL(k_gather_sketches<uint64_t>, dim3(wave_blocks), dim3(kThreads), 0, st, s->d_hashes64, s->d_off, idx->d_orig,
                                   idx->d_src_off, N, perm_hashes64.p);
                src_hashes64 = perm_hashes64.p;
            } else {
                RK_HIP(ctx, perm_hashes.alloc(H));
                hipLaunchKernelGGL(k_gather_sketches<uint32_t>, dim3(wave_blocks), dim3(kThreads), 0, st, s->d_hashes, s->d_off, idx->d_orig,
                                   idx->d_src_off, N, perm_hashes.p);
                src_hashes = perm_hashes.p;
            }
            src_off = idx->d_src_off;
        }
        DevBuf<uint32_t> iota(ctx), keys_sorted(ctx), sorted_e(ctx), flags(ctx), gid(ctx), n_open(ctx), n_cov(ctx);
        DevBuf<uint64_t> keys_sorted64(ctx);
        DevBuf<uint2> self_raw(ctx);
        DevBuf<char> tmp(ctx);
        RK_HIP(ctx, iota.alloc(H));
        RK_HIP(ctx, sorted_e.alloc(H));
        RK_HIP(ctx, flags.alloc(H));
        RK_HIP(ctx, gid.alloc(H));
        RK_HIP(ctx, self_raw.alloc(no_self ? 1 : H));
        RK_HIP(ctx, n_open.alloc((size_t)N + 1));
        RK_HIP(ctx, n_cov.alloc((size_t)N + 1));
        if (idx->wide) RK_HIP(ctx, keys_sorted64.alloc(H));
        else RK_HIP(ctx, keys_sorted.alloc(H));
        hipLaunchKernelGGL(k_fill_gid, dim3(wave_blocks), dim3(kThreads), 0, st, src_off, N, gid.p, iota.p);
        // stable LSD radix sort by hash; values = source element index (genome-major), so equal hashes stay in
        // ascending genome order == hashMapId[hash].push_back(i) for i ascending (src/sketch.cpp:979-985)
        if (idx->wide) {
            RK_TRY(rk_prim_sort_pairs_u64_u32(ctx, src_hashes64, keys_sorted64.p, iota.p, sorted_e.p, H, (unsigned)hash_bits, st));
            hipLaunchKernelGGL(k_head_flags<uint64_t>, dim3(blocks_for(H)), dim3(kThreads), 0, st, keys_sorted64.p, H, flags.p);
        } else {
            RK_TRY(rk_prim_sort_pairs_u32_u32(ctx, src_hashes, keys_sorted.p, iota.p, sorted_e.p, H, (unsigned)hash_bits, st));
            hipLaunchKernelGGL(k_head_flags<uint32_t>, dim3(blocks_for(H)), dim3(kThreads), 0, st, keys_sorted.p, H, flags.p);
        }
        RK_TRY(rk_prim_inclusive_scan_u32(ctx, flags.p, iota.p, H, st));
        uint32_t *gidx = iota.p;  // 1-based group number of each sorted position
        if (idx->wide)
            hipLaunchKernelGGL(k_scatter_heads<uint64_t>, dim3(blocks_for(H)), dim3(kThreads), 0, st, keys_sorted64.p, gidx, H,
                               idx->d_uhash64, idx->d_upos, res.p);
        else
            hipLaunchKernelGGL(k_scatter_heads<uint32_t>, dim3(blocks_for(H)), dim3(kThreads), 0, st, keys_sorted.p, gidx, H,
                               idx->d_uhash, idx->d_upos, res.p);
        if (no_self)
            hipLaunchKernelGGL((k_postings_selfrange<false, true>), dim3(blocks_for(H)), dim3(kThreads), 0, st, sorted_e.p, gidx,
                               idx->d_upos, gid.p, H, idx->d_postings, self_raw.p, res.p);
        else if (s->is_set)
            hipLaunchKernelGGL(k_postings_selfrange<false>, dim3(blocks_for(H)), dim3(kThreads), 0, st, sorted_e.p, gidx,
                               idx->d_upos, gid.p, H, idx->d_postings, self_raw.p, res.p);
        else
            hipLaunchKernelGGL(k_postings_selfrange<true>, dim3(blocks_for(H)), dim3(kThreads), 0, st, sorted_e.p, gidx,
                               idx->d_upos, gid.p, H, idx->d_postings, self_raw.p, res.p);
        if (!no_self) {
            // drop the empty slices (26 % of the elements at 10,000 genomes), covered slices last in their row
            hipLaunchKernelGGL(k_row_counts, dim3(wave_blocks), dim3(kThreads), 0, st, src_off, N, self_raw.p, n_open.p, n_cov.p);
            hipLaunchKernelGGL(k_row_scan, dim3(1), dim3(1024), 0, st, n_open.p, n_cov.p, N, idx->d_self_off, idx->d_self_split, res.p);
            hipLaunchKernelGGL(k_row_place, dim3(wave_blocks), dim3(kThreads), 0, st, src_off, N, self_raw.p,
                               idx->d_self_off, idx->d_self_split, idx->d_postings, s->is_set, idx->d_selfrange);
        }
        RK_HIP(ctx, hipGetLastError());
        RK_TRY(rk_read_back(ctx, &r, res.p, sizeof(r), st));  // the one synchronisation of the build
    } else if (!H) {
        RK_HIP(ctx, hipMemsetAsync(idx->d_upos, 0, 8, st));
        if (!no_self) {
            RK_HIP(ctx, hipMemsetAsync(idx->d_self_off, 0, ((size_t)N + 1) * 8, st));
            RK_HIP(ctx, hipMemsetAsync(idx->d_self_split, 0, ((size_t)N + 1) * 8, st));
        }
        RK_HIP(ctx, hipStreamSynchronize(st));
    }
    idx->U = r.U;
    idx->n_self = r.n_self;
    idx->slices_refused = no_self;
    idx->ref_sets = s->is_set || r.dups == 0;
    idx->built_fast = built;
    if (built) {   // (the same rule as rk_dist.hip self_uses_tiles, which counts the records itself for an index built the general way)
        idx->spread = r.flagged * 8 > r.n_self;
        idx->spread_known = 1;
    }
    set_dir_shape(idx);
    guard.p = nullptr;
    *out = idx;
    return RK_OK;
}

// ---- the exchange of a sharded build -----------------------------------------------------------------------------
namespace {
struct PackArgs {
    unsigned long long dst[kRecRegions];   // where region q's records go in the send buffer
    unsigned long long cnt[kRecRegions];
};
// one workgroup row per region: its valid records to their place in the send buffer (contiguous by destination shard)
__global__ void k_shard_pack(const uint3 *rec, uint32_t region_cap, PackArgs pa, uint3 *out)
{
    const uint32_t q = blockIdx.y;
    const unsigned long long n = pa.cnt[q];
    const uint3 *src = rec + (size_t)q * region_cap;
    uint3 *dst = out + pa.dst[q];
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) dst[i] = src[i];
}
// the received records as kRecRegions equal virtual regions for the tile sort: region q holds records [q * cap, min(n, (q + 1) * cap))
__global__ void k_virtual_regions(unsigned long long n, uint32_t cap, unsigned long long *rec_count)
{
    const uint32_t q = threadIdx.x;
    if (q >= kRecRegions) return;
    const unsigned long long lo = (unsigned long long)q * cap;
    rec_count[q] = n > lo ? min((unsigned long long)cap, n - lo) : 0ULL;
}
}  // namespace

int rk_index_shard_records(const rk_index *idx, uint64_t *counts_out)
{
    if (!idx || !counts_out) return RK_ERR_ARG;
    if (!idx->d_shard_rec || !idx->n_shards) return rk_fail(idx->ctx, RK_ERR_ARG, "not a shard of a sharded build (rk_index_build_shard)");
    const uint32_t sub = kRecRegions / idx->n_shards;
    for (uint32_t d = 0; d < idx->n_shards; d++) {
        counts_out[d] = 0;
        for (uint32_t q = d * sub; q < (d + 1) * sub; q++) counts_out[d] += idx->shard_rec_count[q];
    }
    return RK_OK;
}

int rk_index_shard_pack(const rk_index *idx, void *send_dev, void *stream_v)
{
    if (!idx || !send_dev) return RK_ERR_ARG;
    rk_ctx *ctx = idx->ctx;
    if (!idx->d_shard_rec || !idx->n_shards) return rk_fail(ctx, RK_ERR_ARG, "not a shard of a sharded build (rk_index_build_shard)");
    RK_HIP(ctx, hipSetDevice(ctx->device));
    PackArgs pa;
    unsigned long long at = 0;
    for (uint32_t q = 0; q < kRecRegions; q++) {   // (regions are numbered destination-major)
        pa.dst[q] = at;
        pa.cnt[q] = idx->shard_rec_count[q];
        at += pa.cnt[q];
    }
    if (at) hipLaunchKernelGGL(k_shard_pack, dim3(64, kRecRegions), dim3(256), 0, (hipStream_t)stream_v, idx->d_shard_rec, idx->shard_region_cap, pa, (uint3 *)send_dev);
    RK_HIP(ctx, hipGetLastError());
    return RK_OK;
}

// One process, one context per GPU (what `rabbit_kssd alldist --gpus N` does): the all-to-all of the shards' tile records by
// direct copies -- GPU d pulls chunk d of every shard's packed records over its own links (hipMemcpyPeerAsync; on one device a
// plain copy), no communicator.  recv_dev[d] (library-owned: rk_dev_free on parts[d]'s context) holds what shard d joins.
int rk_index_shard_exchange(rk_index *const *parts, uint32_t n, void **recv_dev, uint64_t *n_recv)
{
    if (!parts || !n || !recv_dev || !n_recv || n > kRecRegions) return RK_ERR_ARG;
    for (uint32_t r = 0; r < n; r++) {
        recv_dev[r] = nullptr;
        n_recv[r] = 0;
        if (!parts[r] || !parts[r]->d_shard_rec || parts[r]->n_shards != n || parts[r]->shard_id != r)
            return rk_fail(parts[0] ? parts[0]->ctx : nullptr, RK_ERR_ARG, "rk_index_shard_exchange: parts[r] must be shard r of %u (rk_index_build_shard)", n);
    }
    std::vector<std::vector<uint64_t>> cnt(n, std::vector<uint64_t>(n, 0));
    std::vector<uint3 *> send(n, nullptr);
    auto cleanup = [&](bool also_recv) {
        for (uint32_t r = 0; r < n; r++) {
            rk_ctx *c = parts[r]->ctx;
            if (hipSetDevice(c->device) == hipSuccess) (void)hipStreamSynchronize(c->stream);
            if (send[r]) rk_pool_free(c, send[r]);
            if (also_recv && recv_dev[r]) { (void)hipFree(recv_dev[r]); recv_dev[r] = nullptr; }   // (plain hipMalloc: the caller frees with rk_dev_free)
        }
        (void)hipGetLastError();
    };
    for (uint32_t r = 0; r < n; r++) {   // every shard packs its records, contiguous by destination
        rk_ctx *c = parts[r]->ctx;
        int rc = rk_index_shard_records(parts[r], cnt[r].data());
        if (rc) { cleanup(true); return rc; }
        uint64_t tot = 0;
        for (uint32_t d = 0; d < n; d++) tot += cnt[r][d];
        if (hipSetDevice(c->device) != hipSuccess) { cleanup(true); return rk_fail(c, RK_ERR_HIP, "cannot select device %d", c->device); }
        send[r] = static_cast<uint3 *>(rk_pool_alloc(c, std::max<uint64_t>(1, tot) * sizeof(uint3)));
        if (!send[r]) { cleanup(true); return rk_fail(c, RK_ERR_NOMEM, "cannot allocate the send buffer of %llu tile records", (unsigned long long)tot); }
        rc = rk_index_shard_pack(parts[r], send[r], c->stream);
        if (rc) { cleanup(true); return rc; }
    }
    for (uint32_t r = 0; r < n; r++) {
        rk_ctx *c = parts[r]->ctx;
        if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { cleanup(true); return rk_fail(c, RK_ERR_HIP, "packing the tile records of shard %u failed", r); }
    }
    for (uint32_t d = 0; d < n; d++) {   // every destination pulls its chunks
        rk_ctx *c = parts[d]->ctx;
        uint64_t tot = 0;
        for (uint32_t r = 0; r < n; r++) tot += cnt[r][d];
        n_recv[d] = tot;
        if (hipSetDevice(c->device) != hipSuccess) { cleanup(true); return rk_fail(c, RK_ERR_HIP, "cannot select device %d", c->device); }
        if (hipMalloc(&recv_dev[d], std::max<uint64_t>(1, tot) * sizeof(uint3)) != hipSuccess) {
            recv_dev[d] = nullptr;
            (void)hipGetLastError();
            cleanup(true);
            return rk_fail(c, RK_ERR_NOMEM, "cannot allocate the receive buffer of %llu tile records", (unsigned long long)tot);
        }
        uint64_t at = 0;
        for (uint32_t r = 0; r < n; r++) {
            const uint64_t c_rd = cnt[r][d];
            if (!c_rd) continue;
            uint64_t from = 0;
            for (uint32_t q = 0; q < d; q++) from += cnt[r][q];
            rk_ctx *sc = parts[r]->ctx;
            uint3 *dst = static_cast<uint3 *>(recv_dev[d]) + at;
            hipError_t e;
            if (sc->device == c->device) e = hipMemcpyAsync(dst, send[r] + from, c_rd * sizeof(uint3), hipMemcpyDeviceToDevice, c->stream);
            else {
                int can = 0;
                (void)hipDeviceCanAccessPeer(&can, c->device, sc->device);
                if (can && hipDeviceEnablePeerAccess(sc->device, 0) != hipSuccess) (void)hipGetLastError();   // (already enabled)
                e = hipMemcpyPeerAsync(dst, c->device, send[r] + from, sc->device, c_rd * sizeof(uint3), c->stream);
            }
            if (e != hipSuccess) { cleanup(true); return rk_fail(c, RK_ERR_HIP, "copy of tile records from device %d to device %d failed: %s", sc->device, c->device, hipGetErrorString(e)); }
            at += c_rd;
        }
    }
    for (uint32_t d = 0; d < n; d++) {   // bounded wait: a peer copy that never completes must end in an error, not in a hang
        rk_ctx *c = parts[d]->ctx;
        hipError_t qe = hipSetDevice(c->device);
        const auto t_start = std::chrono::steady_clock::now();
        while (qe == hipSuccess) {
            qe = hipStreamQuery(c->stream);
            if (qe != hipErrorNotReady) break;
            if (std::chrono::steady_clock::now() - t_start > std::chrono::seconds(120)) break;
            std::this_thread::sleep_for(std::chrono::microseconds(50));
            qe = hipSuccess;
        }
        (void)hipGetLastError();
        if (qe != hipSuccess) {
            // (a copy that is stuck: its buffers are leaked rather than waited for)
            return rk_fail(c, RK_ERR_HIP, "the exchange of tile records did not complete on device %d: %s", c->device, hipGetErrorString(qe));
        }
    }
    for (uint32_t r = 0; r < n; r++) { rk_pool_free(parts[r]->ctx, send[r]); send[r] = nullptr; }
    return RK_OK;
}

int rk_index_join_shard(rk_ctx *ctx, const rk_index *part, const void *recv_dev, uint64_t n_records, rk_index **out)
{
    if (!ctx || !part || !out || (!recv_dev && n_records)) return RK_ERR_ARG;
    *out = nullptr;
    if (!part->d_src_off || !part->d_sizes || !part->n_ref || part->ctx != ctx)
        return rk_fail(ctx, RK_ERR_ARG, "rk_index_join_shard needs an index built on this context (rk_index_build_shard) for the genomes' sizes and order");
    if (n_records >= 0x7FFF0000ULL) return rk_fail(ctx, RK_ERR_UNSUPPORTED, "more than 2^31 tile records for one shard");
    RK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint32_t N = part->n_ref, n_blocks = (N + 31) / 32;
    if (n_blocks > kTileMaxBlocks) return rk_fail(ctx, RK_ERR_UNSUPPORTED, "more than %u genomes", kTileMaxBlocks * 32);
    rk_index *idx = new (std::nothrow) rk_index;
    if (!idx) return RK_ERR_NOMEM;
    struct Guard { rk_index *p; ~Guard() { if (p) rk_index_free(p); } } guard{idx};
    idx->ctx = ctx;
    idx->n_ref = N;
    idx->hash_bits = part->hash_bits;
    idx->wide = part->wide;
    idx->ref_sets = true;
    idx->max_src_size = part->max_src_size;
    idx->max_ref_size = part->max_ref_size;
    idx->min_ref_size = part->min_ref_size;
    idx->slices_refused = true;   // (a join-only index: tile records of this shard's rows, no postings)
    RK_TRY(pool_array(ctx, &idx->d_sizes, (size_t)N + 1));
    RK_TRY(pool_array(ctx, &idx->d_src_off, (size_t)N + 1));
    RK_TRY(pool_array(ctx, &idx->d_blk_min, (size_t)n_blocks));
    RK_HIP(ctx, hipMemcpyAsync(idx->d_sizes, part->d_sizes, (size_t)N * 4, hipMemcpyDeviceToDevice, st));
    RK_HIP(ctx, hipMemcpyAsync(idx->d_src_off, part->d_src_off, ((size_t)N + 1) * 8, hipMemcpyDeviceToDevice, st));
    if (part->relabeled && part->d_orig) {
        RK_TRY(pool_array(ctx, &idx->d_orig, (size_t)N + 1));
        RK_HIP(ctx, hipMemcpyAsync(idx->d_orig, part->d_orig, (size_t)N * 4, hipMemcpyDeviceToDevice, st));
        idx->relabeled = true;
    }
    hipLaunchKernelGGL(k_blk_min_sizes, dim3(blocks_for(n_blocks)), dim3(kThreads), 0, st, idx->d_sizes, N, n_blocks, idx->d_blk_min);
    const uint32_t region_cap = (uint32_t)std::max<uint64_t>(1, (n_records + kRecRegions - 1) / kRecRegions);
    const uint64_t tile_cap = std::max<uint64_t>(1, std::min<uint64_t>(n_records, (uint64_t)n_blocks * (n_blocks + 1) / 2)), slot_cap = n_records + tile_cap;
    DevBuf<uint2> t_contrib(ctx);
    DevBuf<uint32_t> t_rows(ctx), t_cols(ctx), bins(ctx), tb(ctx);
    DevBuf<uint4> t_dir_j(ctx), t_dir_c(ctx), proto(ctx);
    DevBuf<uint3> brec(ctx);
    DevBuf<unsigned long long> level_start(ctx), zeroed(ctx);
    RK_HIP(ctx, brec.alloc(std::max<uint64_t>(1, n_records)));
    RK_HIP(ctx, tb.alloc(3 * (size_t)n_blocks));
    RK_HIP(ctx, bins.alloc((size_t)n_blocks + 1));
    RK_HIP(ctx, proto.alloc(2 * tile_cap));
    RK_HIP(ctx, level_start.alloc(2 * (kTileTable + 1)));
    RK_HIP(ctx, t_contrib.alloc(slot_cap + 256));
    if (rowsort_stage(n_blocks)) {
        RK_HIP(ctx, t_rows.alloc(slot_cap + 256));
        RK_HIP(ctx, t_cols.alloc(slot_cap + 256));
    }
    RK_HIP(ctx, t_dir_j.alloc(2 * tile_cap));
    RK_HIP(ctx, t_dir_c.alloc(2 * tile_cap));
    const size_t z_tres = 0, z_tcur = z_tres + (sizeof(TileResult) + 7) / 8, z_bins = z_tcur + (sizeof(TileCursors) + 7) / 8, z_end = z_bins + (size_t)n_blocks + 1;
    RK_HIP(ctx, zeroed.alloc(z_end));
    RK_HIP(ctx, hipMemsetAsync(zeroed.p, 0, z_end * 8, st));
    TileResult *const tres = reinterpret_cast<TileResult *>(zeroed.p + z_tres);
    hipLaunchKernelGGL(k_virtual_regions, dim3(1), dim3(64), 0, st, (unsigned long long)n_records, region_cap, &tres->rec_count[0]);
    TileSortArgs ta;
    memset(&ta, 0, sizeof ta);
    ta.rec = (const uint3 *)recv_dev;
    ta.cur = reinterpret_cast<TileCursors *>(zeroed.p + z_tcur);
    ta.region_cap = region_cap;
    ta.n_blocks = n_blocks;
    ta.bin_count = reinterpret_cast<uint32_t *>(zeroed.p + z_bins);
    ta.bin_cursor = ta.bin_count + n_blocks + 1;
    ta.bin_start = bins.p;
    ta.brec = brec.p;
    ta.blk_min = idx->d_blk_min;
    ta.contrib = t_contrib.p;
    ta.rows = t_rows.p;
    ta.cols = t_cols.p;
    ta.tb_base = tb.p;
    ta.tb_cnt = tb.p + n_blocks;
    ta.order = tb.p + 2 * (size_t)n_blocks;
    ta.proto = proto.p;
    ta.level_start = level_start.p;
    ta.dir[0] = t_dir_j.p;
    ta.dir[1] = t_dir_c.p;
    ta.tres = tres;
    RK_TRY(launch_tile_sort(ctx, ta, st, []() { return RK_OK; }));
    TileResult tr;
    RK_TRY(rk_read_back(ctx, &tr, tres, sizeof(tr), st));
    if (ctx->sw_dist_debug) {   // (developer output: how the records spread over the row blocks)
        std::vector<uint32_t> bc(n_blocks);
        RK_TRY(rk_read_back(ctx, bc.data(), ta.bin_count, (size_t)n_blocks * 4, st));
        std::sort(bc.begin(), bc.end(), std::greater<uint32_t>());
        unsigned long long top = 0, all = 0;
        uint32_t non_empty = 0;
        for (uint32_t i = 0; i < n_blocks; i++) {
            all += bc[i];
            if (i < 64) top += bc[i];
            non_empty += bc[i] != 0;
        }
        fprintf(stderr, "[rk] join shard: %llu records in %u of %u row blocks; fullest %u, 64th %u, the 64 fullest hold %llu\n", all, non_empty, n_blocks, bc[0],
                bc[std::min<uint32_t>(63, n_blocks - 1)], top);
    }
    idx->d_tile_contrib = t_contrib.release();
    idx->d_tile_rows = t_rows.release();
    idx->d_tile_cols = t_cols.release();
    idx->d_tile_dir[0] = t_dir_j.release();
    idx->d_tile_dir[1] = t_dir_c.release();
    idx->n_tile_slots = tr.n_slots;
    idx->n_tiles = tr.n_tiles;
    idx->n_tile_records = tr.n_records;
    idx->tile_max_records = tr.max_records;
    for (int m = 0; m < 2; m++)
        for (int k = 0; k < kTileTable; k++) idx->tile_prefix[m][k] = tr.level_count[m][k];
    idx->tiles_ready = true;
    idx->tiles_from_build = true;
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
    hipStream_t st = ctx->stream;
    const uint64_t hs = 1ULL << hash_bits;
    rk_index *idx = new (std::nothrow) rk_index;
    if (!idx) return RK_ERR_NOMEM;
    idx->ctx = ctx;
    idx->n_ref = n_ref;
    idx->H = total;
    idx->hash_bits = hash_bits;
    for (uint32_t g = 0; g < n_ref; g++) {
        idx->max_ref_size = std::max<uint64_t>(idx->max_ref_size, ref_sizes[g]);
        if (ref_sizes[g] && (!idx->min_ref_size || ref_sizes[g] < idx->min_ref_size)) idx->min_ref_size = ref_sizes[g];
    }
    struct Guard { rk_index *p; ~Guard() { if (p) rk_index_free(p); } } guard{idx};

    RK_TRY(pool_array(ctx, &idx->d_sizes, (size_t)n_ref + 1));
    RK_TRY(pool_array(ctx, &idx->d_postings, total + 8));
    RK_HIP(ctx, hipMemcpyAsync(idx->d_sizes, ref_sizes, (size_t)n_ref * 4, hipMemcpyHostToDevice, st));
    RK_HIP(ctx, hipMemcpyAsync(idx->d_postings, postings, total * 4, hipMemcpyHostToDevice, st));

    DevBuf<uint32_t> d_counts(ctx), flags(ctx), rank(ctx), cpos(ctx);
    RK_HIP(ctx, d_counts.alloc(hs));
    RK_HIP(ctx, flags.alloc(hs));
    RK_HIP(ctx, rank.alloc(hs));
    RK_HIP(ctx, cpos.alloc(hs));
    RK_HIP(ctx, hipMemcpyAsync(d_counts.p, counts, hs * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_nonzero_flags, dim3(blocks_for(hs)), dim3(kThreads), 0, st, d_counts.p, hs, flags.p);
    RK_TRY(rk_prim_exclusive_scan_u32(ctx, flags.p, rank.p, hs, st));
    RK_TRY(rk_prim_exclusive_scan_u32(ctx, d_counts.p, cpos.p, hs, st));
    uint32_t last[4] = {0, 0, 0, 0};  // rank, flag, cpos, count of the last hash value
    DevBuf<uint32_t> last_dev(ctx);
    RK_HIP(ctx, last_dev.alloc(4));
    RK_HIP(ctx, hipMemcpyAsync(last_dev.p + 0, rank.p + (hs - 1), 4, hipMemcpyDeviceToDevice, st));
    RK_HIP(ctx, hipMemcpyAsync(last_dev.p + 1, flags.p + (hs - 1), 4, hipMemcpyDeviceToDevice, st));
    RK_HIP(ctx, hipMemcpyAsync(last_dev.p + 2, cpos.p + (hs - 1), 4, hipMemcpyDeviceToDevice, st));
    RK_HIP(ctx, hipMemcpyAsync(last_dev.p + 3, d_counts.p + (hs - 1), 4, hipMemcpyDeviceToDevice, st));
    RK_TRY(rk_read_back(ctx, last, last_dev.p, 16, st));
    idx->U = (uint64_t)last[0] + last[1];
    if ((uint64_t)last[2] + last[3] != total)  // src/dist.cpp:107-110
        return rk_fail(ctx, RK_ERR_ARG, "mismatched total hash number: index says %llu, dict has %llu",
                       (unsigned long long)last[2] + last[3], (unsigned long long)total);
    RK_TRY(pool_array(ctx, &idx->d_uhash, idx->U + 1));
    RK_TRY(pool_array(ctx, &idx->d_upos, idx->U + 2));
    hipLaunchKernelGGL(k_compact_dense, dim3(blocks_for(hs)), dim3(kThreads), 0, st, d_counts.p, rank.p, cpos.p, hs,
                       idx->d_uhash, idx->d_upos);
    const uint32_t tot32 = (uint32_t)total;
    memcpy(ctx->pinned, &tot32, 4);
    RK_HIP(ctx, hipMemcpyAsync(idx->d_upos + idx->U, ctx->pinned, 4, hipMemcpyHostToDevice, st));
    RK_HIP(ctx, hipGetLastError());
    set_dir_shape(idx);
    RK_TRY(classify_lists(ctx, idx, st));  // synchronises
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
    hipStream_t st = ctx->stream;
    uint32_t *mapped = nullptr;
    if (postings && idx->H) {
        RK_TRY(postings_in_caller_ids(ctx, idx, st, &mapped));
        struct Free { rk_ctx *c; uint32_t *p; ~Free() { rk_pool_free(c, p); } } guard{ctx, mapped};
        RK_HIP(ctx, hipMemcpyAsync(postings, mapped ? mapped : idx->d_postings, idx->H * 4, hipMemcpyDeviceToHost, st));
        RK_HIP(ctx, hipStreamSynchronize(st));
    }
    if (counts) {
        const uint64_t hs = 1ULL << idx->hash_bits;
        DevBuf<uint32_t> d_counts(ctx);
        RK_HIP(ctx, d_counts.alloc(hs));
        RK_HIP(ctx, hipMemsetAsync(d_counts.p, 0, hs * 4, st));
        if (idx->U)
            hipLaunchKernelGGL(k_scatter_counts, dim3(blocks_for(idx->U)), dim3(kThreads), 0, st,
                               idx->d_uhash, idx->d_upos, idx->U, d_counts.p);
        RK_HIP(ctx, hipGetLastError());
        RK_HIP(ctx, hipMemcpyAsync(counts, d_counts.p, hs * 4, hipMemcpyDeviceToHost, st));
        RK_HIP(ctx, hipStreamSynchronize(st));
    }
    RK_HIP(ctx, hipStreamSynchronize(st));
    return RK_OK;
}

}  // extern "C"

// ---- single-blob form for the RCCL broadcast ------------------------------------------
namespace {
// a received blob is only trusted after its payload was checked against its own header: posting ids, posting offsets, the
// genome order and every slice record must point inside the index (a corrupt record would index LDS out of range)
__global__ void k_validate_blob(const uint32_t *postings, uint64_t H, const uint32_t *upos, uint64_t U, const uint32_t *orig,
                                const uint2 *selfrange, uint64_t n_self, const uint64_t *self_off, const uint64_t *self_split,
                                uint32_t n_ref, const uint64_t *src_off, uint32_t *bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = true;
    if (i < H) ok = ok && postings[i] < n_ref;
    if (src_off && i < n_ref) ok = ok && src_off[i] <= src_off[i + 1] && src_off[i + 1] <= H && (i != 0 || src_off[0] == 0);
    if (i < U) ok = ok && upos[i] < upos[i + 1] && upos[i + 1] <= H;
    if (orig && i < n_ref) ok = ok && orig[i] < n_ref;
    if (selfrange) {
        if (i < n_self) {
            const uint2 r = selfrange[i];
            if (r.x >> 31) ok = ok && r.y != 0 && (uint64_t)(r.x & 0x7FFFFFFFu) + (31 - __clz((int)r.y)) < n_ref;
            else ok = ok && r.x < (r.y & 0x7FFFFFFFu) && (r.y & 0x7FFFFFFFu) <= H;
        }
        if (i < n_ref) ok = ok && self_off[i] <= self_split[i] && self_split[i] <= self_off[i + 1] && self_off[i + 1] <= n_self;
    }
    if (!ok) *bad = 1;
}
}  // namespace

namespace {
struct BlobHeader {
    uint64_t magic, bytes;
    uint64_t H, U, max_src_size, max_ref_size, min_ref_size, n_self;
    uint32_t n_ref, has_self, ref_sets, relabeled;
    int32_t hash_bits, wide, has_src;   // has_src: the CSR offsets of the source sketches travel (an index built here: it can self join)
    uint64_t off_postings, off_uhash, off_upos, off_sizes, off_self, off_selfoff, off_src, off_split, off_orig;
};
constexpr uint64_t kBlobMagic = 0x37584449444b5352ULL;  // "RSKDIDX7"
inline uint64_t al256(uint64_t x) { return (x + 255) & ~255ULL; }

// the layout this library produces for an index of these dimensions (derived arrays -- prefix directory, rank
// bitmap -- are not shipped: every rank rebuilds them on first use)
void blob_layout(BlobHeader *h)
{
    const bool wide = h->wide != 0;
    uint64_t p = al256(sizeof(BlobHeader));
    h->off_postings = p; p = al256(p + (h->H + 1) * 4);
    h->off_uhash = p;    p = al256(p + (h->U + 1) * (wide ? 8 : 4));
    h->off_upos = p;     p = al256(p + (h->U + 2) * 4);
    h->off_sizes = p;    p = al256(p + ((uint64_t)h->n_ref + 1) * 4);
    h->off_self = h->off_selfoff = h->off_src = h->off_split = h->off_orig = 0;
    if (h->relabeled) { h->off_orig = p; p = al256(p + ((uint64_t)h->n_ref + 1) * 4); }
    if (h->has_src) { h->off_src = p; p = al256(p + ((uint64_t)h->n_ref + 1) * 8); }
    if (h->has_self) {
        h->off_self = p;    p = al256(p + (h->n_self + 1) * sizeof(uint2));
        h->off_selfoff = p; p = al256(p + ((uint64_t)h->n_ref + 1) * 8);
        h->off_split = p;   p = al256(p + ((uint64_t)h->n_ref + 1) * 8);
    }
    h->bytes = p;
}

void blob_header(const rk_index *idx, BlobHeader *h)
{
    memset(h, 0, sizeof(*h));
    h->magic = kBlobMagic;
    h->H = idx->H;
    h->U = idx->U;
    h->max_src_size = idx->max_src_size;
    h->max_ref_size = idx->max_ref_size;
    h->min_ref_size = idx->min_ref_size;
    h->n_self = idx->n_self;
    h->n_ref = idx->n_ref;
    h->has_self = idx->d_selfrange ? 1 : 0;
    h->has_src = idx->d_src_off ? 1 : 0;
    h->ref_sets = idx->ref_sets ? 1 : 0;
    h->relabeled = idx->relabeled && idx->d_orig ? 1 : 0;
    h->hash_bits = idx->hash_bits;
    h->wide = idx->wide ? 1 : 0;
    blob_layout(h);
}
}  // namespace

extern "C" {

uint64_t rk_index_blob_bytes(const rk_index *idx)
{
    if (!idx) return 0;
    BlobHeader h;
    blob_header(idx, &h);
    return h.bytes;
}

int rk_index_pack_dev(const rk_index *idx, void *blob_dev, uint64_t blob_cap, void *stream_v)
{
    if (!idx || !blob_dev) return RK_ERR_ARG;
    rk_ctx *ctx = idx->ctx;
    hipStream_t st = (hipStream_t)stream_v;
    BlobHeader h;
    blob_header(idx, &h);
    if (blob_cap < h.bytes) return rk_fail(ctx, RK_ERR_CAPACITY, "blob needs %llu bytes", (unsigned long long)h.bytes);
    char *b = (char *)blob_dev;
    RK_HIP(ctx, hipSetDevice(ctx->device));
    RK_HIP(ctx, hipMemcpyAsync(b, &h, sizeof(h), hipMemcpyHostToDevice, st));
    RK_HIP(ctx, hipMemcpyAsync(b + h.off_postings, idx->d_postings, idx->H * 4, hipMemcpyDeviceToDevice, st));
    if (idx->wide) RK_HIP(ctx, hipMemcpyAsync(b + h.off_uhash, idx->d_uhash64, idx->U * 8, hipMemcpyDeviceToDevice, st));
    else RK_HIP(ctx, hipMemcpyAsync(b + h.off_uhash, idx->d_uhash, idx->U * 4, hipMemcpyDeviceToDevice, st));
    RK_HIP(ctx, hipMemcpyAsync(b + h.off_upos, idx->d_upos, (idx->U + 1) * 4, hipMemcpyDeviceToDevice, st));
    RK_HIP(ctx, hipMemcpyAsync(b + h.off_sizes, idx->d_sizes, (uint64_t)idx->n_ref * 4, hipMemcpyDeviceToDevice, st));
    if (h.relabeled) RK_HIP(ctx, hipMemcpyAsync(b + h.off_orig, idx->d_orig, (uint64_t)idx->n_ref * 4, hipMemcpyDeviceToDevice, st));
    if (h.has_self) {
        RK_HIP(ctx, hipMemcpyAsync(b + h.off_self, idx->d_selfrange, idx->n_self * sizeof(uint2), hipMemcpyDeviceToDevice, st));
        RK_HIP(ctx, hipMemcpyAsync(b + h.off_selfoff, idx->d_self_off, ((uint64_t)idx->n_ref + 1) * 8, hipMemcpyDeviceToDevice, st));
        RK_HIP(ctx, hipMemcpyAsync(b + h.off_split, idx->d_self_split, ((uint64_t)idx->n_ref + 1) * 8, hipMemcpyDeviceToDevice, st));
    }
    if (h.has_src) RK_HIP(ctx, hipMemcpyAsync(b + h.off_src, idx->d_src_off, ((uint64_t)idx->n_ref + 1) * 8, hipMemcpyDeviceToDevice, st));
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
    // the offsets must be exactly the ones this library computes for the stated dimensions, and fit the buffer:
    // a truncated or foreign blob is rejected before any copy is sized from it
    BlobHeader chk = h;
    blob_layout(&chk);
    if (h.magic != kBlobMagic || h.H >= 0xFFFFFFFFULL || h.U > h.H || h.hash_bits < 1 || h.hash_bits > 64 ||
        (h.wide != 0) != (h.hash_bits > 32) || memcmp(&chk, &h, sizeof(h)) != 0 || h.bytes > blob_bytes)
        return rk_fail(ctx, RK_ERR_ARG, "not an index blob of this library (or truncated: %llu of %llu bytes)",
                       (unsigned long long)blob_bytes, (unsigned long long)h.bytes);
    rk_index *idx = new (std::nothrow) rk_index;
    if (!idx) return RK_ERR_NOMEM;
    struct Guard { rk_index *p; ~Guard() { if (p) rk_index_free(p); } } guard{idx};
    idx->ctx = ctx;
    idx->n_ref = h.n_ref;
    idx->H = h.H;
    idx->U = h.U;
    idx->max_src_size = h.max_src_size;
    idx->max_ref_size = h.max_ref_size;
    idx->min_ref_size = h.min_ref_size;
    idx->n_self = h.n_self;
    idx->ref_sets = h.ref_sets != 0;
    idx->hash_bits = h.hash_bits;
    idx->wide = h.wide != 0;
    set_dir_shape(idx);
    const char *b = (const char *)blob_dev;
    RK_TRY(pool_array(ctx, &idx->d_postings, idx->H + 8));
    if (idx->wide) RK_TRY(pool_array(ctx, &idx->d_uhash64, idx->U + 1));
    else RK_TRY(pool_array(ctx, &idx->d_uhash, idx->U + 1));
    RK_TRY(pool_array(ctx, &idx->d_upos, idx->U + 2));
    RK_TRY(pool_array(ctx, &idx->d_sizes, (size_t)idx->n_ref + 1));
    RK_HIP(ctx, hipMemcpyAsync(idx->d_postings, b + h.off_postings, idx->H * 4, hipMemcpyDeviceToDevice, st));
    if (idx->wide) RK_HIP(ctx, hipMemcpyAsync(idx->d_uhash64, b + h.off_uhash, idx->U * 8, hipMemcpyDeviceToDevice, st));
    else RK_HIP(ctx, hipMemcpyAsync(idx->d_uhash, b + h.off_uhash, idx->U * 4, hipMemcpyDeviceToDevice, st));
    RK_HIP(ctx, hipMemcpyAsync(idx->d_upos, b + h.off_upos, (idx->U + 1) * 4, hipMemcpyDeviceToDevice, st));
    RK_HIP(ctx, hipMemcpyAsync(idx->d_sizes, b + h.off_sizes, (uint64_t)idx->n_ref * 4, hipMemcpyDeviceToDevice, st));
    if (h.relabeled) {
        RK_TRY(pool_array(ctx, &idx->d_orig, (size_t)idx->n_ref + 1));
        RK_HIP(ctx, hipMemcpyAsync(idx->d_orig, b + h.off_orig, (uint64_t)idx->n_ref * 4, hipMemcpyDeviceToDevice, st));
        idx->relabeled = true;
    }
    if (h.has_self) {
        RK_TRY(pool_array(ctx, &idx->d_selfrange, idx->n_self + 1));
        RK_TRY(pool_array(ctx, &idx->d_self_off, (size_t)idx->n_ref + 1));
        RK_TRY(pool_array(ctx, &idx->d_self_split, (size_t)idx->n_ref + 1));
        RK_HIP(ctx, hipMemcpyAsync(idx->d_selfrange, b + h.off_self, idx->n_self * sizeof(uint2), hipMemcpyDeviceToDevice, st));
        RK_HIP(ctx, hipMemcpyAsync(idx->d_self_off, b + h.off_selfoff, ((uint64_t)idx->n_ref + 1) * 8, hipMemcpyDeviceToDevice, st));
        RK_HIP(ctx, hipMemcpyAsync(idx->d_self_split, b + h.off_split, ((uint64_t)idx->n_ref + 1) * 8, hipMemcpyDeviceToDevice, st));
    }
    if (h.has_src) {   // (an index without slice records -- built with tile records -- still self joins: its tile records are rebuilt on first use)
        RK_TRY(pool_array(ctx, &idx->d_src_off, (size_t)idx->n_ref + 1));
        RK_HIP(ctx, hipMemcpyAsync(idx->d_src_off, b + h.off_src, ((uint64_t)idx->n_ref + 1) * 8, hipMemcpyDeviceToDevice, st));
    }
    {
        DevBuf<uint32_t> bad(ctx);
        RK_HIP(ctx, bad.alloc(1));
        RK_HIP(ctx, hipMemsetAsync(bad.p, 0, 4, st));
        const uint64_t span = std::max<uint64_t>(std::max<uint64_t>(idx->H, idx->n_self), (uint64_t)idx->n_ref + 1);
        if (span)
            hipLaunchKernelGGL(k_validate_blob, dim3(blocks_for(span)), dim3(kThreads), 0, st, idx->d_postings, idx->H, idx->d_upos, idx->U,
                               idx->relabeled ? idx->d_orig : nullptr, idx->d_selfrange, idx->n_self, idx->d_self_off, idx->d_self_split,
                               idx->n_ref, idx->d_src_off, bad.p);
        RK_HIP(ctx, hipGetLastError());
        uint32_t b = 0;
        RK_TRY(rk_read_back(ctx, &b, bad.p, 4, st));  // synchronises: the copies above are complete
        if (b) return rk_fail(ctx, RK_ERR_ARG, "index blob payload is inconsistent with its header (corrupt or foreign blob)");
    }
    guard.p = nullptr;
    *out = idx;
    return RK_OK;
}

}  // extern "C"

// ---- one-to-all replication inside one process (the host tool's --gpus N) -----------------------------------
static constexpr int kPeerCopyTimeoutS = 120;
extern "C" int rk_index_broadcast(const rk_index *src, rk_ctx *const *dst, uint32_t n_dst, rk_index **out)
{
    if (!src || (n_dst && (!dst || !out))) return RK_ERR_ARG;
    rk_ctx *sctx = src->ctx;
    for (uint32_t i = 0; i < n_dst; i++) out[i] = nullptr;
    if (!n_dst) return RK_OK;
    const uint64_t bytes = rk_index_blob_bytes(src);
    RK_HIP(sctx, hipSetDevice(sctx->device));
    DevBuf<char> blob(sctx);
    if (blob.alloc(bytes) != hipSuccess) return rk_fail(sctx, RK_ERR_NOMEM, "cannot allocate the %llu-byte index blob", (unsigned long long)bytes);
    int rc = rk_index_pack_dev(src, blob.p, bytes, sctx->stream);  // synchronises the source stream
    if (rc) return rc;
    // every peer pulls the blob over its own xGMI link at the same time (the links are point to point, so a
    // one-to-all of direct copies moves the blob once per link, like a pipelined ring, without a communicator)
    std::vector<char *> peer(n_dst, nullptr);
    for (uint32_t i = 0; i < n_dst; i++)
        if (!dst[i]) return RK_ERR_ARG;
    // error exit: copies already enqueued into the first n peer blobs may still be running -- wait for them before the
    // blocks return to their pools (a later allocation could receive a block that is still being written), and leave
    // the caller's device current
    auto abandon = [&](uint32_t n) {
        for (uint32_t j = 0; j < n; j++) {
            if (!peer[j]) continue;
            if (hipSetDevice(dst[j]->device) == hipSuccess) (void)hipStreamSynchronize(dst[j]->stream);
            rk_pool_free(dst[j], peer[j]);
        }
        (void)hipGetLastError();
        (void)hipSetDevice(sctx->device);
    };
    for (uint32_t i = 0; i < n_dst; i++) {
        rk_ctx *d = dst[i];
        if (hipSetDevice(d->device) != hipSuccess) {
            abandon(i);
            return rk_fail(d, RK_ERR_HIP, "cannot select device %d", d->device);
        }
        peer[i] = static_cast<char *>(rk_pool_alloc(d, bytes));
        if (!peer[i]) {
            abandon(i);
            return rk_fail(d, RK_ERR_NOMEM, "cannot allocate the %llu-byte index blob on device %d", (unsigned long long)bytes, d->device);
        }
        hipError_t e;
        if (d->device == sctx->device) {
            e = hipMemcpyAsync(peer[i], blob.p, bytes, hipMemcpyDeviceToDevice, d->stream);
        } else {
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, d->device, sctx->device);
            if (can) {
                hipError_t pe = hipDeviceEnablePeerAccess(sctx->device, 0);
                if (pe != hipSuccess) (void)hipGetLastError();  // already enabled
            }
            e = hipMemcpyPeerAsync(peer[i], d->device, blob.p, sctx->device, bytes, d->stream);
        }
        if (e != hipSuccess) {
            abandon(i + 1);
            return rk_fail(d, RK_ERR_HIP, "index copy to device %d failed: %s", d->device, hipGetErrorString(e));
        }
    }
    bool stuck = false;  // a copy that never completed: its blocks are leaked rather than waited for
    for (uint32_t i = 0; i < n_dst && !rc; i++) {
        rk_ctx *d = dst[i];
        // bounded wait: a peer copy that never completes (a link that is down) must end in an error, not in a hang
        hipError_t qe = hipSetDevice(d->device);
        const auto t_start = std::chrono::steady_clock::now();
        while (qe == hipSuccess) {
            qe = hipStreamQuery(d->stream);
            if (qe != hipErrorNotReady) break;
            if (std::chrono::steady_clock::now() - t_start > std::chrono::seconds(kPeerCopyTimeoutS)) break;
            std::this_thread::sleep_for(std::chrono::microseconds(50));
            qe = hipSuccess;
        }
        (void)hipGetLastError();  // (hipStreamQuery's hipErrorNotReady is sticky for the next hipGetLastError check)
        if (qe == hipErrorNotReady) stuck = true;
        if (qe == hipErrorNotReady)
            rc = rk_fail(d, RK_ERR_HIP, "index copy of %llu bytes from device %d to device %d did not complete within %d s", (unsigned long long)bytes,
                         sctx->device, d->device, kPeerCopyTimeoutS);
        else if (qe != hipSuccess)
            rc = rk_fail(d, RK_ERR_HIP, "index copy to device %d failed: %s", d->device, hipGetErrorString(qe));
        else
            rc = rk_index_unpack_dev(d, peer[i], bytes, d->stream, &out[i]);
    }
    if (rc) {
        // the caller asks the SOURCE context for the message: carry the failing peer's text over
        for (uint32_t i = 0; i < n_dst; i++)
            if (dst[i] != sctx && !dst[i]->err.empty()) sctx->err = "device " + std::to_string(dst[i]->device) + ": " + dst[i]->err;
        if (!stuck) abandon(n_dst);  // waits for the copies of the peers behind the failing one
        else (void)hipSetDevice(sctx->device);
        for (uint32_t i = 0; i < n_dst; i++) {
            rk_index_free(out[i]);
            out[i] = nullptr;
        }
        return rc;
    }
    for (uint32_t i = 0; i < n_dst; i++) rk_pool_free(dst[i], peer[i]);
    (void)hipSetDevice(sctx->device);
    return rc;
}

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

int rk_index_export_lists(const rk_index *idx, uint32_t *postings, uint32_t *hashes, uint32_t *counts)
{
    if (!idx) return RK_ERR_ARG;
    rk_ctx *ctx = idx->ctx;
    if (idx->wide) return rk_fail(ctx, RK_ERR_ARG, "64-bit index: use rk_index_export64");
    RK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (postings && idx->H) {
        uint32_t *mapped = nullptr;
        RK_TRY(postings_in_caller_ids(ctx, idx, st, &mapped));
        struct Free { rk_ctx *c; uint32_t *p; ~Free() { rk_pool_free(c, p); } } guard{ctx, mapped};
        RK_HIP(ctx, hipMemcpyAsync(postings, mapped ? mapped : idx->d_postings, idx->H * 4, hipMemcpyDeviceToHost, st));
        RK_HIP(ctx, hipStreamSynchronize(st));
    }
    if (hashes && idx->U) RK_HIP(ctx, hipMemcpyAsync(hashes, idx->d_uhash, idx->U * 4, hipMemcpyDeviceToHost, st));
    DevBuf<uint32_t> c(ctx);
    if (counts && idx->U) {
        RK_HIP(ctx, c.alloc(idx->U));
        hipLaunchKernelGGL(k_counts_from_upos, dim3(blocks_for(idx->U)), dim3(kThreads), 0, st, idx->d_upos, idx->U, c.p);
        RK_HIP(ctx, hipGetLastError());
        RK_HIP(ctx, hipMemcpyAsync(counts, c.p, idx->U * 4, hipMemcpyDeviceToHost, st));
    }
    RK_HIP(ctx, hipStreamSynchronize(st));
    return RK_OK;
}

int rk_index_export64(const rk_index *idx, uint32_t *postings, uint64_t *hashes, uint32_t *counts)
{
    if (!idx) return RK_ERR_ARG;
    rk_ctx *ctx = idx->ctx;
    if (!idx->wide) return rk_fail(ctx, RK_ERR_ARG, "32-bit index: use rk_index_export (dense .index layout)");
    RK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (postings && idx->H) {
        uint32_t *mapped = nullptr;
        RK_TRY(postings_in_caller_ids(ctx, idx, st, &mapped));
        struct Free { rk_ctx *c; uint32_t *p; ~Free() { rk_pool_free(c, p); } } guard{ctx, mapped};
        RK_HIP(ctx, hipMemcpyAsync(postings, mapped ? mapped : idx->d_postings, idx->H * 4, hipMemcpyDeviceToHost, st));
        RK_HIP(ctx, hipStreamSynchronize(st));
    }
    if (hashes && idx->U) RK_HIP(ctx, hipMemcpyAsync(hashes, idx->d_uhash64, idx->U * 8, hipMemcpyDeviceToHost, st));
    DevBuf<uint32_t> c(ctx);
    if (counts && idx->U) {
        RK_HIP(ctx, c.alloc(idx->U));
        hipLaunchKernelGGL(k_counts_from_upos, dim3(blocks_for(idx->U)), dim3(kThreads), 0, st, idx->d_upos, idx->U, c.p);
        RK_HIP(ctx, hipGetLastError());
        RK_HIP(ctx, hipMemcpyAsync(counts, c.p, idx->U * 4, hipMemcpyDeviceToHost, st));
    }
    RK_HIP(ctx, hipStreamSynchronize(st));
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
    for (uint32_t g = 0; g < n_ref; g++) {
        idx->max_ref_size = std::max<uint64_t>(idx->max_ref_size, ref_sizes[g]);
        if (ref_sizes[g] && (!idx->min_ref_size || ref_sizes[g] < idx->min_ref_size)) idx->min_ref_size = ref_sizes[g];
    }
    hipStream_t st = ctx->stream;
    RK_TRY(pool_array(ctx, &idx->d_sizes, (size_t)n_ref + 1));
    RK_TRY(pool_array(ctx, &idx->d_postings, total + 8));
    RK_TRY(pool_array(ctx, &idx->d_uhash64, n_hash + 1));
    RK_TRY(pool_array(ctx, &idx->d_upos, n_hash + 2));
    RK_HIP(ctx, hipMemcpyAsync(idx->d_sizes, ref_sizes, (size_t)n_ref * 4, hipMemcpyHostToDevice, st));
    RK_HIP(ctx, hipMemcpyAsync(idx->d_postings, post.data(), total * 4, hipMemcpyHostToDevice, st));
    RK_HIP(ctx, hipMemcpyAsync(idx->d_uhash64, uh.data(), n_hash * 8, hipMemcpyHostToDevice, st));
    RK_HIP(ctx, hipMemcpyAsync(idx->d_upos, up.data(), (n_hash + 1) * 4, hipMemcpyHostToDevice, st));
    set_dir_shape(idx);
    RK_TRY(classify_lists(ctx, idx, st));  // synchronises: the host vectors above may go
    guard.p = nullptr;
    *out = idx;
    return RK_OK;
}

}  // extern "C"
