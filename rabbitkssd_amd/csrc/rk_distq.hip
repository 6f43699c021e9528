// rk_distq.hip -- ref-vs-query distances in ONE pass (replaces the row loop of index_dist,
// src/dist.cpp:560-692: per query `memset(row)`, for every query hash the index probe :565-570 and the
// posting walk :571-588, then the epilogue :600-682).
//
// A workgroup owns one (query, reference tile) unit at a time; the reference's per-thread counter row
// (`intersectionArr[tid][numRef]`, src/dist.cpp:541) is an LDS row of 8-, 16- or 32-bit counters -- as narrow as
// the bound min(|largest query|, |largest reference|) on an intersection count allows, so that 100,000 reference
// columns of 76-hash bacterial sketches are ONE 100 KB tile.
//
// Per unit:  for every query hash: LOOK UP its list record, keep the present ones, COUNT / WALK them into the row |
//            barrier | scan the row into the cell list, clearing it | barrier | evaluate, stage hits.
//            (The row is zeroed once per workgroup; a unit that reports every cell or copies the dense row out zeroes
//            it per unit behind two more barriers.)
//
// Look-up (fused; round 1 ran it as a separate pass that wrote 8 B per query hash to HBM and read them back
// through flags/scan/compact passes): the index's distinct hashes as a RANK BITMAP over the hash space, one 8-byte
// entry per 48 consecutive hash values: 48 presence bits + a 16-bit rank relative to a u32 base every 64 entries
// (2.8 MB for the 24-bit hashes of L4K10, 45 MB for the 28 bits of L3K10).  One 8-byte load (+ the base: one cache
// line for the 64 sorted hashes of a wave) answers "is h indexed" and "which distinct hash is it"; a query's hashes
// are sorted, so consecutive lanes read consecutive entries.  Only present hashes (~11 % for an unrelated mammal
// against 100,000 bacteria) go on to load their LIST RECORD (8 B): the posting range, or -- for a list that spans
// fewer than 32 genome ids, the neighbouring members of a clade -- the list itself as (bit 31 | first genome, bitmask).
// Hash spaces above 2^30 and 64-bit hashes use the prefix directory + binary search instead of the bitmap.
//
// Compaction, counting and walk: present records are appended to a per-wave LDS queue (ballot + popcount, no atomics);
// whenever 64 are queued the wave pops them.  Compact lists are counted by the WAVE when they fall into one 32-column
// window (a query with relatives among the references names the same columns in most of its lists): one ballot per
// occupied column, one conflict-free add per column; the others are scattered lane by lane.  Posting ranges are walked
// in 4 steps: in step j quad q serves the range held by lane 4q+j, every lane fetching two postings with one 8-byte
// load, i.e. the first 8 postings of 16 lists per step; lists longer than 8 re-enter the queue as (x+8, y).
// Integer/index work: bound by the L1's pending misses (one cache line per probe) -- no MFMA.
#include <algorithm>
#include <cmath>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "rk_dist_common.h"

namespace {

constexpr uint32_t kQueueCap = 128;     // ranges per wave: < 64 left over (the append drains first) + <= 64 appended
constexpr uint32_t kMaxThreads = 1024;
constexpr uint32_t kLookups = 4;        // query hashes per lane and iteration (independent loads in flight)

enum { kLookRank = 0, kLookDir32 = 1, kLookDir64 = 2, kLookPre = 3 };   // kLookPre: the present hashes were resolved by k_member_sliced
constexpr uint32_t kSegStride = 8;   // slice ranges per query (kLookPre): room in the segment tables

struct DistQArgs {
    const void *q_hashes;        // u32[] or u64[] (kLookDir64)
    const uint64_t *q_off;       // u64[n_query+1]
    const uint2 *rankbm;         // kLookRank: entry of 48 hash values
    const uint32_t *rankbase;    // kLookRank: rank at the start of every 64th entry
    const uint2 *urec;           // per distinct hash: posting range or compact list (rk_internal.h d_urec); null: ranges from upos
    const void *uhash;           // kLookDir*: sorted distinct hashes
    const uint32_t *dir;
    uint32_t min_fit;            // compact lists of a popped batch that must share the 32-column window before the wave counts columns (else all are scattered member by member)
    const uint32_t *rec, *seg_start, *seg_cnt;   // kLookPre: ranks of the present hashes, per query and slice range where (relative to q_off) and how many
    const uint32_t *upos;        // u32[U+1] posting offsets of the distinct hashes
    const uint32_t *postings;
    const uint32_t *ref_sizes;
    const uint32_t *orig;        // internal reference id -> the caller's (null: identity)
    int32_t hash_bits, dir_shift;
    uint32_t n_query, n_ref;
    uint32_t row_first, row_step, row_block, n_units;   // block-cyclic row shard (rk_dist_opts)
    uint32_t tile_cols, n_tiles, cnt_words;
    uint32_t cand_cap, stage_hits;
    int32_t triangle, metric, kmer_size, dense_mode;
    double max_dist, min_jorc;
    uint32_t min_ref_size;       // smallest non-empty reference sketch
    rk_hit *hits;
    unsigned long long cap;
    unsigned long long *n_hits;
    int32_t *common_dense;
};

// PIPE: the look-up's three memory levels belong to three batches in flight (below); for counter rows of 64 KiB and more --
// one or two workgroups per CU whatever the registers --: the 33 registers it costs take a resident workgroup from small rows
template <int CBITS, int LOOK, bool PIPE>
__global__ __launch_bounds__(kMaxThreads) void rk_distq_kernel(DistQArgs a)
{
    typedef typename std::conditional<LOOK == kLookDir64, uint64_t, uint32_t>::type K;
    constexpr uint32_t kPerWord = 32 / CBITS;
    constexpr uint32_t kCellMask = CBITS == 32 ? 0xFFFFFFFFu : ((1u << (CBITS & 31)) - 1u);
    const uint32_t nthreads = blockDim.x, nwaves = nthreads >> 6;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 3;

    // dynamic LDS: counter row | per-wave range queues | non-zero cell list | staged hits | scalars
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *cnt = lds;
    uint2 *queue = reinterpret_cast<uint2 *>(lds + a.cnt_words) + wave * kQueueCap;
    uint2 *cand = reinterpret_cast<uint2 *>(lds + a.cnt_words) + nwaves * kQueueCap;
    rk_hit *stage = reinterpret_cast<rk_hit *>(cand + a.cand_cap);
    uint32_t *scal = reinterpret_cast<uint32_t *>(stage + a.stage_hits);
    unsigned long long &s_base = *reinterpret_cast<unsigned long long *>(scal);
    uint32_t &s_total = scal[2];
    uint32_t &s_cursor = scal[3];
    uint32_t *s_cells = scal + 4;  // [2] cell-list counters of consecutive units, taking turns
    if (tid == 0) s_cursor = 0;
    // Sparse report without dense counter output: the row is zeroed ONCE.  The scan covers every cell a unit can
    // touch and clears the non-zero quads it meets, so the row is clean again when the next unit scatters: no zero pass
    // and two barriers less per unit (what matters for many small queries; a 45,776-hash query does not notice).
    const bool keep_clean = !a.dense_mode && !a.common_dense;
    if (keep_clean) {
        uint4 *z = reinterpret_cast<uint4 *>(cnt);
        for (uint32_t i = tid; i < a.cnt_words / 4; i += nthreads) z[i] = make_uint4(0, 0, 0, 0);
        if (tid == 0) s_cells[0] = s_cells[1] = 0;
        __syncthreads();
    }
    uint32_t par = 0;  // uniform: which of the two counters this unit's scan uses

    const K *qh = reinterpret_cast<const K *>(a.q_hashes);
    const K *uhash = reinterpret_cast<const K *>(a.uhash);
    const unsigned long long lt_mask = lane ? (~0ULL >> (64 - lane)) : 0ULL;

    auto bump_cell = [&](uint32_t c) {  // src/dist.cpp:578 `intersectionArr[tid][curIndex]++`
        if (CBITS == 32) atomicAdd(&cnt[c], 1u);
        else atomicAdd(&cnt[c / kPerWord], 1u << ((c % kPerWord) * CBITS));
    };
    auto cell = [&](uint32_t c) -> uint32_t {
        return CBITS == 32 ? cnt[c] : (cnt[c / kPerWord] >> ((c % kPerWord) * CBITS)) & kCellMask;
    };

    // units: (row slot, tile).  The tiles of a row are consecutive queue positions of one XCD (workgroups b, b+8, ...
    // share an L2: they repeat the same look-ups), rows are dealt round-robin over the XCDs.
    const uint32_t total = a.n_units * a.n_tiles;
    const uint32_t per_round = 8 * a.n_tiles;
    const uint32_t total_padded = (total + per_round - 1) / per_round * per_round;
    for (uint32_t it = blockIdx.x; it < total_padded; it += gridDim.x) {
        const uint32_t x = it & 7, p = it >> 3;
        const uint32_t u = ((p / a.n_tiles) * 8 + x) * a.n_tiles + p % a.n_tiles;
        if (u >= total) continue;
        const uint32_t slot = u / a.n_tiles, tile = u % a.n_tiles;
        // block-cyclic rows: this shard owns blocks row_first, row_first + row_step, ... of row_block rows
        const uint32_t blk = slot / a.row_block;
        const uint64_t row64 = ((uint64_t)a.row_first + (uint64_t)blk * a.row_step) * a.row_block + slot % a.row_block;
        if (row64 >= a.n_query) continue;
        const uint32_t row = (uint32_t)row64;
        const uint32_t col0 = tile * a.tile_cols;
        const uint32_t col1 = min(a.n_ref, col0 + a.tile_cols);
        const uint32_t ncol = col1 - col0;
        // (a renumbered index: the columns are internal ids, `j > i` is decided on the caller's ids where a cell is evaluated)
        const bool tri_filter = a.triangle && !a.common_dense && !a.orig;
        if (tri_filter && col1 <= row + 1) continue;       // nothing right of the diagonal in this tile
        const uint32_t lo_id = tri_filter ? row + 1 : 0;   // src/dist.cpp:207: j > i

        uint4 *z4 = reinterpret_cast<uint4 *>(cnt);        // memset(row), src/dist.cpp:563
        if (!keep_clean) {
            for (uint32_t i = tid; i < a.cnt_words / 4; i += nthreads) z4[i] = make_uint4(0, 0, 0, 0);
            if (tid == 0) s_total = 0;
            __syncthreads();
        }

        auto bump = [&](uint32_t id, bool valid) {
            const uint32_t c = id - col0;
            if (valid && c < ncol && id >= lo_id) bump_cell(c);
        };
        auto bump_n = [&](uint32_t id, uint32_t n) {  // a wave-level count (<= 64; the counter width bounds the total)
            const uint32_t c = id - col0;
            if (n && c < ncol && id >= lo_id) {
                if (CBITS == 32) atomicAdd(&cnt[c], n);
                else atomicAdd(&cnt[c / kPerWord], n << ((c % kPerWord) * CBITS));
            }
        };
        uint32_t qn = 0;  // wave-uniform: ranges in this wave's queue
        // pops up to 64 ranges and walks the first 8 postings of each; longer lists re-enter the queue
        auto walk = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const uint32_t take = min(qn, 64u);
            qn -= take;
            uint2 ent = make_uint2(0, 0);
            if (lane < take) ent = queue[qn + lane];
            // Compact records carry their posting list (the neighbouring ids of a clade): no posting load.  A query that
            // has relatives among the references meets the same few columns in most of its lists: as in the self join the
            // WAVE counts them -- every list is shifted onto the 32 columns from the smallest first genome on, one ballot
            // per occupied column, one conflict-free add per column -- and only lists that reach beyond that window (the
            // clades of an unrelated query) are scattered lane by lane.
            const bool cpt = a.urec && (ent.x >> 31) != 0;
            const uint2 rg = cpt ? make_uint2(0, 0) : ent;
            if (__ballot(cpt)) {  // uniform
                const uint32_t first = ent.x & 0x7FFFFFFFu;
                const uint32_t base = wave_min(cpt ? first : 0xFFFFFFFFu);
                const uint32_t rel = first - base;
                bool fits = cpt && rel < 32u && (rel == 0 || (ent.y >> (32u - rel)) == 0);
                // (a query without relatives among the references: its 64 lists lie in 64 clades, one or two share the window of
                // the smallest -- the column count would cost more than the members it saves from the loop below)
                if (__popcll(__ballot(fits)) < (int)a.min_fit) fits = false;   // (uniform)
                const uint32_t mw = fits ? ent.y << rel : 0u;
                uint32_t ca = 0, cb = 0;
                if (__ballot(fits)) {   // (uniform)
                    count_columns<false, 0>(wave_or(mw), mw, 0u, ca, cb);
                    bump_n(base + lane, ca);
                }
                uint32_t m = cpt && !fits ? ent.y : 0u;
                while (__ballot(m != 0)) {
                    const bool v = m != 0;
                    bump(first + (v ? (uint32_t)__ffs((int)m) - 1u : 0u), v);
                    m &= m - 1u;
                }
            }
            if (!__ballot(rg.y > rg.x)) {  // uniform: no posting range among them
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                return;
            }
            PostingPair id[4];
            bool ok0[4], ok1[4];
            auto step = [&](int j, uint32_t rx, uint32_t ry) {
                const uint32_t k = rx + 2 * sub;
                ok0[j] = k < ry;
                ok1[j] = k + 1 < ry;
                id[j] = *reinterpret_cast<const PostingPair *>(a.postings + (ok0[j] ? k : 0));  // postings[0..1] are mapped
            };
            step(0, quad_bcast<0>(rg.x), quad_bcast<0>(rg.y));
            step(1, quad_bcast<1>(rg.x), quad_bcast<1>(rg.y));
            step(2, quad_bcast<2>(rg.x), quad_bcast<2>(rg.y));
            step(3, quad_bcast<3>(rg.x), quad_bcast<3>(rg.y));
            const bool more = rg.y - rg.x > 8u;
            const unsigned long long m = __ballot(more);
            if (m) {  // uniform
                if (more) queue[qn + __popcll(m & lt_mask)] = make_uint2(rg.x + 8, rg.y);
                qn += __popcll(m);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                bump(id[j].x, ok0[j]);
                bump(id[j].y, ok1[j]);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        };

        const uint64_t qb = a.q_off[row], qe = a.q_off[row + 1];
        if (LOOK == kLookPre) {
            // the query's present hashes were found by k_member_sliced: their ranks lie in up to kSegStride segments of
            // a.rec; a flat index over them (the segment of an index by a compare chain over the eight prefix sums)
            uint32_t pre[kSegStride + 1], rel[kSegStride];
            pre[0] = 0;
#pragma unroll
            for (uint32_t y = 0; y < kSegStride; y++) {
                pre[y + 1] = pre[y] + a.seg_cnt[(size_t)row * kSegStride + y];
                rel[y] = a.seg_start[(size_t)row * kSegStride + y] - pre[y];   // record address = rel[segment] + flat index
            }
            const uint32_t total = pre[kSegStride];
            const uint32_t *recs = a.rec + qb;
            for (uint32_t i0 = 0; i0 < total; i0 += kLookups * nthreads) {
                uint32_t pos[kLookups];
                bool present[kLookups];
#pragma unroll
                for (uint32_t i = 0; i < kLookups; i++) {
                    const uint32_t ix = i0 + i * nthreads + tid;
                    present[i] = ix < total;
                    uint32_t off = rel[0];
#pragma unroll
                    for (uint32_t y = 1; y < kSegStride; y++) off = ix >= pre[y] ? rel[y] : off;
                    pos[i] = present[i] ? recs[off + ix] : 0u;
                }
                PostingPair r[kLookups];
#pragma unroll
                for (uint32_t i = 0; i < kLookups; i++)
                    if (present[i]) {
                        const uint2 t = a.urec[pos[i]];
                        r[i].x = t.x;
                        r[i].y = t.y;
                    }
#pragma unroll
                for (uint32_t i = 0; i < kLookups; i++) {
                    const unsigned long long m = __ballot(present[i]);
                    if (m) {  // uniform
                        if (present[i]) queue[qn + __popcll(m & lt_mask)] = make_uint2(r[i].x, r[i].y);
                        qn += __popcll(m);
                        while (qn >= 64) walk();
                    }
                }
            }
        } else if (PIPE) {
            // (round 4) The look-up is a chain of three dependent memory accesses per hash -- the hash, the bitmap entry it
            // names, the list record of a present one -- and a workgroup holds ONE 100 KB counter row: four waves per SIMD,
            // which waited 70 % of their time with the chain issued and awaited batch by batch (SQ_WAIT_ANY).  Now the three
            // levels belong to three different batches: while the records of batch n-1 are queued and walked, the list records
            // of batch n, the bitmap entries of batch n+1 and the hashes of batch n+2 are in flight.
            const uint64_t step = (uint64_t)kLookups * nthreads;
            K h0[kLookups], h1[kLookups];             // h0: the batch resolved next (its bitmap entries are in flight); h1: the one after (its hashes are)
            uint2 w0[kLookups];
            uint32_t base0[kLookups];
            PostingPair r_old[kLookups];              // list records requested a turn ago
            bool p_old[kLookups];
            auto load_h = [&](uint64_t e0, K (&h)[kLookups]) {
#pragma unroll
                for (uint32_t i = 0; i < kLookups; i++) {
                    const uint64_t e = e0 + (uint64_t)i * nthreads + tid;
                    h[i] = e < qe ? qh[e] : (K)~(K)0;   // (all ones: outside every hash space of at most 30 bits -- "not in bounds")
                }
            };
            auto in_bounds = [&](K h) { return (h >> a.hash_bits) == 0; };   // (kLookRank: hash_bits <= 30)
            auto entry_of = [&](K h) { return in_bounds(h) ? (uint32_t)(((uint64_t)(uint32_t)h * 0xAAAAAAABull) >> 37) : 0u; };   // h / 48
            auto request_entries = [&]() {
#pragma unroll
                for (uint32_t i = 0; i < kLookups; i++) {
                    const uint32_t ent = entry_of(h0[i]);
                    w0[i] = a.rankbm[ent];
                    base0[i] = a.rankbase[ent >> 6];
                }
            };
#pragma unroll
            for (uint32_t i = 0; i < kLookups; i++) p_old[i] = false;
            load_h(qb, h0);
            request_entries();
            load_h(qb + step, h1);
            for (uint64_t e0 = qb; e0 < qe + step; e0 += step) {   // (one more turn than batches: the last batch's records are queued in it)
                // A: resolve this turn's batch (its entries were requested a turn ago) and request the list records of the present hashes
                PostingPair r_new[kLookups];
                bool p_new[kLookups];
#pragma unroll
                for (uint32_t i = 0; i < kLookups; i++) {
                    const uint32_t ent = entry_of(h0[i]);
                    const uint32_t b = (uint32_t)h0[i] - 48u * ent;
                    const uint64_t bits = (uint64_t)w0[i].x | ((uint64_t)(w0[i].y & 0xFFFFu) << 32);
                    p_new[i] = e0 < qe && in_bounds(h0[i]) && ((bits >> b) & 1u);
                    if (p_new[i]) {
                        const uint2 t = a.urec[base0[i] + (w0[i].y >> 16) + (uint32_t)__popcll(bits & ((1ULL << b) - 1ULL))];
                        r_new[i].x = t.x;
                        r_new[i].y = t.y;
                    }
                }
                // B: the next batch's hashes arrived during the last turn: request its entries; and the hashes of the batch after it
#pragma unroll
                for (uint32_t i = 0; i < kLookups; i++) h0[i] = h1[i];
                request_entries();
                load_h(e0 + 2 * step, h1);
                // C: queue and walk what the previous turn requested
#pragma unroll
                for (uint32_t i = 0; i < kLookups; i++) {
                    const unsigned long long m = __ballot(p_old[i]);
                    if (m) {  // uniform
                        if (p_old[i]) queue[qn + __popcll(m & lt_mask)] = make_uint2(r_old[i].x, r_old[i].y);
                        qn += __popcll(m);
                        while (qn >= 64) walk();
                    }
                }
#pragma unroll
                for (uint32_t i = 0; i < kLookups; i++) {
                    r_old[i] = r_new[i];
                    p_old[i] = p_new[i];
                }
            }
        } else
        for (uint64_t e0 = qb; e0 < qe; e0 += (uint64_t)kLookups * nthreads) {
            K h[kLookups];
            bool inb[kLookups];
#pragma unroll
            for (uint32_t i = 0; i < kLookups; i++) {
                const uint64_t e = e0 + (uint64_t)i * nthreads + tid;
                inb[i] = e < qe;
                h[i] = inb[i] ? qh[e] : (K)0;
                // a query hash outside the reference's hash space cannot be indexed
                if (a.hash_bits < (int)(8 * sizeof(K))) inb[i] = inb[i] && (h[i] >> a.hash_bits) == 0;
            }
            uint32_t pos[kLookups];
            bool present[kLookups];
            if (LOOK == kLookRank) {
                // one 8-byte entry answers "is h indexed, and which distinct hash is it" for 48 consecutive hash values
                // (48 presence bits + a 16-bit rank relative to a base every 64 entries: the bases of a wave's 64 sorted
                // hashes share a cache line).  48 values per entry instead of 32: a third fewer lines per query.
                uint2 w[kLookups];
                uint32_t base[kLookups], ent[kLookups];
#pragma unroll
                for (uint32_t i = 0; i < kLookups; i++) {
                    ent[i] = inb[i] ? (uint32_t)(((uint64_t)(uint32_t)h[i] * 0xAAAAAAABull) >> 37) : 0u;  // h / 48
                    w[i] = a.rankbm[ent[i]];
                    base[i] = a.rankbase[ent[i] >> 6];
                }
#pragma unroll
                for (uint32_t i = 0; i < kLookups; i++) {
                    const uint32_t b = (uint32_t)h[i] - 48u * ent[i];
                    const uint64_t bits = (uint64_t)w[i].x | ((uint64_t)(w[i].y & 0xFFFFu) << 32);
                    present[i] = inb[i] && ((bits >> b) & 1u);
                    pos[i] = base[i] + (w[i].y >> 16) + (uint32_t)__popcll(bits & ((1ULL << b) - 1ULL));
                }
            } else {
                // prefix directory + binary search in the sorted distinct hashes (k_resolve of round 1, inlined)
                uint32_t lo[kLookups], hi[kLookups];
#pragma unroll
                for (uint32_t i = 0; i < kLookups; i++) {
                    const uint32_t b = inb[i] ? (uint32_t)(h[i] >> a.dir_shift) : 0u;
                    lo[i] = a.dir[b];
                    hi[i] = inb[i] ? a.dir[b + 1] : lo[i];
                }
#pragma unroll
                for (uint32_t i = 0; i < kLookups; i++) {
                    const uint32_t end = hi[i];
                    while (lo[i] < hi[i]) {
                        const uint32_t mid = (lo[i] + hi[i]) >> 1;
                        if (uhash[mid] < h[i]) lo[i] = mid + 1; else hi[i] = mid;
                    }
                    present[i] = inb[i] && lo[i] < end && uhash[lo[i]] == h[i];
                    pos[i] = lo[i];
                }
            }
            // posting ranges of the present hashes only (masked lanes issue no request); all of a batch are requested
            // before the first is queued
            PostingPair r[kLookups];
#pragma unroll
            for (uint32_t i = 0; i < kLookups; i++)
                if (present[i]) {
                    if (a.urec) {  // uniform
                        const uint2 t = a.urec[pos[i]];
                        r[i].x = t.x;
                        r[i].y = t.y;
                    } else {
                        r[i] = *reinterpret_cast<const PostingPair *>(a.upos + pos[i]);
                    }
                }
#pragma unroll
            for (uint32_t i = 0; i < kLookups; i++) {
                const unsigned long long m = __ballot(present[i]);
                if (m) {  // uniform
                    if (present[i]) queue[qn + __popcll(m & lt_mask)] = make_uint2(r[i].x, r[i].y);
                    qn += __popcll(m);
                    // a walk re-queues every list with more than 8 postings left, so one walk need not shrink the
                    // queue: drain until fewer than 64 wait, then the next 64 fit
                    while (qn >= 64) walk();
                }
            }
        }
        while (qn) walk();
        __syncthreads();  // all scatters of the unit done
        // (clean rows: the counter of the NEXT unit's list; every thread read it for the unit before this one ahead of the barrier)
        if (keep_clean && tid == 0) s_cells[par ^ 1] = 0;
        uint32_t &s_ncells = keep_clean ? s_cells[par] : s_total;

        // ---- epilogue (src/dist.cpp:600-682; :207-255 in triangle mode) -------------------------------------
        if (a.common_dense) {
            int32_t *dst = a.common_dense + (size_t)row * a.n_ref;
            for (uint32_t i = tid; i < ncol; i += nthreads) dst[a.orig ? a.orig[col0 + i] : col0 + i] = (int32_t)cell(i);
        }
        const int qsize = (int)(qe - qb);
        const uint32_t jbeg = a.triangle && !a.orig ? max(col0, row + 1) : col0;  // :207 / :600
        // Row-level reject, applied while the row is scanned: a reportable cell needs common >= min_jorc * denominator
        // (the exact-safe pre-filter below) and the denominator is at least the query's size (jaccard: |q| + |r| - common
        // with |r| >= common) or min(|q|, smallest non-empty reference) (containment).  A 45,776-hash query shares 1-2
        // chance hashes with ~15,000 of 100,000 unrelated references: they never enter the cell list, and their
        // reference sizes are never fetched.
        const uint32_t lb = a.metric ? min((uint32_t)qsize, a.min_ref_size) : (uint32_t)qsize;
        const uint32_t min_common = max(1u, (uint32_t)floor(a.min_jorc * (double)lb));
        auto evaluate = [&](uint32_t j, int common, rk_hit &hrec) -> bool {
            const int rs = (int)a.ref_sizes[j];
            const int size0 = a.triangle ? qsize : rs;  // :215-216 / :607-608
            const int size1 = a.triangle ? rs : qsize;
            // exact-safe reject before the FP64 divide + log (the distance is monotone in jaccard/containment
            // and min_jorc sits strictly below the value at the threshold)
            const int denom = a.metric ? min(size0, size1) : size0 + size1 - common;
            if ((double)common < a.min_jorc * (double)denom) return false;
            const uint32_t out_col = a.orig ? a.orig[j] : j;
            if (a.orig && a.triangle && out_col <= row) return false;  // :207, on the caller's ids
            const JorcDist jd = rk_distance(common, size0, size1, a.metric, a.kmer_size);
            hrec.row = row;
            hrec.col = out_col;
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
            if (sl < a.stage_hits) stage[sl] = hrec;
            else {  // staging full: pay the device-scope atomic per hit
                const unsigned long long at = atomicAdd(a.n_hits, 1ULL);
                if (at < a.cap) a.hits[at] = hrec;
            }
        };
        if (!a.dense_mode) {
            // the threshold excludes distance 1.0 (== common 0): scan the row 16 B per lane skipping all-zero
            // quads, compact the non-zero cells into an LDS list, evaluate the list one cell per lane
            const uint4 *c4 = reinterpret_cast<const uint4 *>(cnt);
            const uint32_t q_first = ((jbeg - col0) / kPerWord) / 4;
            const uint32_t q_end = ((ncol + kPerWord - 1) / kPerWord + 3) / 4;
            for (uint32_t q = q_first + tid; q < q_end; q += nthreads) {
                const uint4 v = c4[q];
                if ((v.x | v.y | v.z | v.w) == 0) continue;
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                uint32_t n = 0;
#pragma unroll
                for (int wi = 0; wi < 4; wi++)
#pragma unroll
                    for (uint32_t s = 0; s < kPerWord; s++) n += ((w[wi] >> (s * CBITS % 32)) & kCellMask) >= min_common;
                if (!n) {
                    if (keep_clean) z4[q] = make_uint4(0, 0, 0, 0);
                    continue;
                }
                uint32_t at = atomicAdd(&s_ncells, n);
                // a quad whose cells do not all fit the list stays in the row (and is not cleared): the walk below finds it
                const bool fits = at + n <= a.cand_cap;
                if (keep_clean && fits) z4[q] = make_uint4(0, 0, 0, 0);
                const uint32_t cq = q * 4 * kPerWord;
#pragma unroll
                for (int wi = 0; wi < 4; wi++)
#pragma unroll
                    for (uint32_t s = 0; s < kPerWord; s++) {
                        const uint32_t common = (w[wi] >> (s * CBITS % 32)) & kCellMask;
                        if (common >= min_common) {
                            if (at < a.cand_cap) cand[at] = make_uint2(!keep_clean || fits ? cq + wi * kPerWord + s : 0xFFFFFFFFu, common);
                            at++;
                        }
                    }
            }
            __syncthreads();
            const uint32_t n_cells = s_ncells;
            if (n_cells <= a.cand_cap || keep_clean) {
                for (uint32_t i = tid; i < min(n_cells, a.cand_cap); i += nthreads) {
                    const uint2 cj = cand[i];
                    const uint32_t j = col0 + cj.x;
                    rk_hit hrec;
                    if (cj.x != 0xFFFFFFFFu && j >= jbeg && j < col1 && evaluate(j, (int)cj.y, hrec)) stage_hit(hrec);
                }
            }
            if (n_cells > a.cand_cap) {  // more sharing columns than the list holds: walk (what is left of) the row, one cell per lane
                for (uint32_t c = (jbeg - col0) + tid; c < ncol; c += nthreads) {
                    const uint32_t common = cell(c);
                    rk_hit hrec;
                    if (common >= min_common && evaluate(col0 + c, (int)common, hrec)) stage_hit(hrec);
                }
                if (keep_clean) {  // the leftovers
                    __syncthreads();
                    for (uint32_t i = tid; i < a.cnt_words / 4; i += nthreads) z4[i] = make_uint4(0, 0, 0, 0);
                    __syncthreads();
                }
            }
        } else {
            // every cell of [jbeg, col1) can be reported: pass 0 counts the unit's reports, one atomic reserves
            // their slots, pass 1 re-evaluates and writes them
            for (int pass_no = 0; pass_no < 2; pass_no++) {
                uint32_t mine = 0;
                for (uint32_t j = jbeg + tid; j < col1; j += nthreads) {
                    rk_hit hrec;
                    if (!evaluate(j, (int)cell(j - col0), hrec)) continue;
                    if (pass_no == 0) { mine++; continue; }
                    const unsigned long long at = s_base + atomicAdd(&s_total, 1u);
                    if (at < a.cap) a.hits[at] = hrec;
                }
                if (pass_no == 0) {
                    if (mine) atomicAdd(&s_total, mine);
                    __syncthreads();
                    const uint32_t tot = s_total;
                    __syncthreads();
                    if (tot == 0) break;  // uniform
                    if (tid == 0) { s_base = atomicAdd(a.n_hits, (unsigned long long)tot); s_total = 0; }
                    __syncthreads();
                }
            }
        }
        if (!keep_clean) __syncthreads();  // the row is zeroed next
        par ^= 1;
    }

    // flush the staged hits of this workgroup: one device-scope atomic, coalesced 8-byte stores
    __syncthreads();
    const uint32_t n_st = min(s_cursor, a.stage_hits);
    if (n_st == 0) return;
    if (tid == 0) s_base = atomicAdd(a.n_hits, (unsigned long long)n_st);
    __syncthreads();
    const unsigned long long at0 = s_base;
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(stage);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(a.hits);
    constexpr uint32_t kW = sizeof(rk_hit) / 8;
    for (uint32_t i = tid; i < n_st * kW; i += nthreads)
        if (at0 + i / kW < a.cap) dst[at0 * kW + i] = src[i];
}

// ---- sliced membership (round 4) ------------------------------------------------------------------------------
// What bounds the fused look-up of a big query is the NUMBER of L1 misses: its sorted hashes lie ~60 bytes apart in the rank
// bitmap, so every look-up of a wave is a cache line of its own and the L1's miss queue is full half of the kernel's time
// (TCP_PENDING_STALL, 27 M requests to L2 for 45.8 M look-ups).  Here the bitmap comes to the hashes instead: a workgroup of
// 16 waves -- one query each -- walks a range of 48 KiB slices of the bitmap; a slice is loaded into LDS once (coalesced),
// every wave reads on through its sorted query (a cursor: no search after the first slice) and answers from LDS; the rank
// of every present hash is appended to the query's record list (ballot + prefix: the records of a query and range are one
// compact, sorted segment).  The counting kernel (LOOK == kLookPre) then reads 4 bytes per PRESENT hash.
constexpr uint32_t kSliceEntries = 6144;   // bitmap entries (of 48 hash values) per slice: 48 KiB, 96 rank bases
constexpr uint32_t kMemberWaves = 16;

struct MemberArgs {
    const uint32_t *q_hashes;
    const uint64_t *q_off;
    uint32_t n_query;
    const uint2 *rankbm;
    const uint32_t *rankbase;
    uint64_t n_entries;
    uint32_t n_slices, slices_per_range;
    uint32_t *rec, *seg_start, *seg_cnt;
    int debug;   // developer ablations (RK_MEMBER_DEBUG): 1 no look-ups, 2 no slice loads, 4 no search
};

// number of elements of the sorted array h[0 .. n) below `target`, found by the whole wave: 64 probes per round
__device__ inline uint64_t wave_lower_bound(const uint32_t *h, uint64_t n, uint64_t target, uint32_t lane)
{
    uint64_t lo = 0, hi = n;   // the answer lies in [lo, hi]
    while (hi - lo > 64) {
        const uint64_t span = hi - lo;
        const uint64_t p = lo + (uint64_t)(lane + 1) * span / 65;   // lo < p < hi, ascending with the lane
        const unsigned long long below = __ballot((uint64_t)h[p] < target);   // (a prefix of the lanes: the array ascends)
        const uint32_t c = (uint32_t)__popcll(below);
        const uint64_t new_lo = c ? lo + (uint64_t)c * span / 65 + 1 : lo;        // probe c - 1 is below: the answer is behind it
        const uint64_t new_hi = c < 64 ? lo + (uint64_t)(c + 1) * span / 65 : hi; // probe c is not: the answer is at most its index
        lo = new_lo;
        hi = new_hi;
    }
    const uint64_t ix = lo + lane;
    const unsigned long long below = __ballot(ix < hi && (uint64_t)h[ix] < target);
    return lo + (uint64_t)__popcll(below);
}

// (two workgroups per CU: 64 registers -- with 90 a launch of 315 workgroups ran in two rounds, 127 -> 205 us)
__global__ __launch_bounds__(kMemberWaves * 64, 8) void k_member_sliced(MemberArgs a)
{
    __shared__ __attribute__((aligned(16))) uint2 bm[kSliceEntries];
    __shared__ uint32_t base[kSliceEntries / 64];
    // a wave's records gather here and leave in bursts: on this hardware stores and loads retire in one order, so a store per
    // vector of hashes made the wait for the NEXT vector a wait for that store's acknowledgement (measured: 127 -> 205 us with
    // the hashes prefetched and the stores left in the loop)
    constexpr uint32_t kOutBuf = 192, kOutFlush = 128;
    __shared__ uint32_t obuf[kMemberWaves][kOutBuf];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t q = blockIdx.x * kMemberWaves + wave, y = blockIdx.y;
    const bool live = q < a.n_query;
    const uint32_t s0 = y * a.slices_per_range, s1 = min(a.n_slices, s0 + a.slices_per_range);
    const uint64_t qb = live ? a.q_off[q] : 0;
    const uint32_t qn = live ? (uint32_t)(a.q_off[q + 1] - qb) : 0u;   // (the host takes this path for fewer than 2^32 query hashes in all)
    const uint32_t *qh = a.q_hashes + qb;
    const unsigned long long lt_mask = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    uint32_t cursor = live && s0 < s1 && !(a.debug & 4) ? (uint32_t)wave_lower_bound(qh, qn, (uint64_t)s0 * kSliceEntries * 48, lane) : 0u;
    const uint32_t start = cursor;
    uint32_t *out = a.rec + qb + start;   // (a range's records start where its hashes do: there are no more of them than hashes)
    uint32_t n_out = 0, n_buf = 0;
    auto flush_out = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = lane; i < n_buf; i += 64) out[n_out + i] = obuf[wave][i];
        __builtin_amdgcn_wave_barrier();
        n_out += n_buf;
        n_buf = 0;
    };
    // the query is read in aligned vectors of 64 hashes, four of them in flight: a vector stays when a slice ends inside it
    // (its remaining lanes belong to the next slice), so the stream never waits for a slice boundary
    constexpr int kAhead = 4;
    uint32_t v = cursor & ~63u;
    uint32_t hv[kAhead];
    auto load_vec = [&](uint32_t at) { return at + lane < qn ? qh[at + lane] : 0xFFFFFFFFu; };   // (qn + 256 does not wrap: fewer than 2^32 - 2^20 hashes)
#pragma unroll
    for (int j = 0; j < kAhead; j++) hv[j] = load_vec(v + 64u * j);
    for (uint32_t s = s0; s < s1; s++) {
        const uint64_t e_first = (uint64_t)s * kSliceEntries;
        const uint32_t n_e = (uint32_t)min((uint64_t)kSliceEntries, a.n_entries - e_first);
        __syncthreads();   // (every wave is done with the slice before)
        // (the next slice prefetched into registers while this one is used: 12 registers that did not fit the 64 of two
        // workgroups per CU -- spilled, the loop below ran ten times slower)
        if (!(a.debug & 2))
            for (uint32_t i = tid; i < n_e; i += kMemberWaves * 64) bm[i] = a.rankbm[e_first + i];
        for (uint32_t i = tid; i < (n_e + 63) / 64; i += kMemberWaves * 64) base[i] = a.rankbase[(e_first >> 6) + i];
        __syncthreads();
        const uint64_t end_value = (e_first + n_e) * 48;   // first hash value behind the slice
        while (v < qn && !(a.debug & 1)) {   // (uniform)
            const uint32_t ix = v + lane;
            const uint32_t h = hv[0];
            const bool valid = ix >= cursor && ix < qn;
            const bool in = valid && (uint64_t)h < end_value;   // (the query ascends: the lanes of a slice are a stretch of the vector)
            const uint32_t ent = (uint32_t)(((uint64_t)h * 0xAAAAAAABull) >> 37);   // h / 48
            const uint32_t loc = in ? ent - (uint32_t)e_first : 0u;
            const uint2 w = bm[loc];
            const uint32_t b = h - 48u * ent;
            const uint64_t bits = (uint64_t)w.x | ((uint64_t)(w.y & 0xFFFFu) << 32);
            const bool present = in && ((bits >> b) & 1u);
            const unsigned long long m = __ballot(present);
            if (present) obuf[wave][n_buf + (uint32_t)__popcll(m & lt_mask)] = base[loc >> 6] + (w.y >> 16) + (uint32_t)__popcll(bits & ((1ULL << b) - 1ULL));
            n_buf += (uint32_t)__popcll(m);
            if (n_buf > kOutFlush) flush_out();   // (uniform; at most 64 more fit behind kOutFlush)
            cursor += (uint32_t)__popcll(__ballot(in));
            if (__ballot(valid && !in)) break;   // the rest of this vector lies behind the slice
            v += 64;
#pragma unroll
            for (int j = 0; j + 1 < kAhead; j++) hv[j] = hv[j + 1];
            hv[kAhead - 1] = load_vec(v + 64u * (kAhead - 1));
        }
    }
    flush_out();
    if (live && lane == 0) {
        a.seg_start[(size_t)q * kSegStride + y] = (uint32_t)start;
        a.seg_cnt[(size_t)q * kSegStride + y] = n_out;
    }
}

// ---- rank bitmap over the hash space --------------------------------------------------------------------
// one thread per entry of 48 hash values: uhash is sorted, so a binary search finds the entry's first distinct hash
// (its rank) and the few that follow set its presence bits -- no temporary, no atomics
constexpr uint32_t kRankSpan = 48, kRankBlock = 64;  // values per entry, entries per rank base
__device__ inline uint64_t first_at_least(const uint32_t *uhash, uint64_t U, uint64_t key)
{
    uint64_t lo = 0, hi = U;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((uint64_t)uhash[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__global__ void k_rank_fill(const uint32_t *uhash, uint64_t U, uint64_t n_entries, uint2 *out, uint32_t *base)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) return;
    const uint64_t key = e * kRankSpan;
    const uint64_t at = first_at_least(uhash, U, key);
    const uint64_t at_block = first_at_least(uhash, U, (e / kRankBlock) * kRankBlock * kRankSpan);
    uint64_t bits = 0;
    for (uint64_t i = at; i < U && (uint64_t)uhash[i] < key + kRankSpan; i++) bits |= 1ULL << ((uint64_t)uhash[i] - key);
    out[e] = make_uint2((uint32_t)bits, (uint32_t)(bits >> 32) | ((uint32_t)(at - at_block) << 16));  // < 64 * 48 distinct below
    if (e % kRankBlock == 0) base[e / kRankBlock] = (uint32_t)at;
}

// per distinct hash: its posting range, or the list itself when it spans fewer than 32 genome ids
__global__ void k_urec(const uint32_t *upos, const uint32_t *postings, uint64_t U, uint2 *out)
{
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const uint32_t x = upos[u], y = upos[u + 1];
    uint2 r = make_uint2(x, y);
    if (y > x) {
        const uint32_t first = postings[x];
        if (postings[y - 1] - first < 32u) {  // (a list is sorted by genome)
            uint32_t mask = 0;
            for (uint32_t k = x; k < y; k++) mask |= 1u << (postings[k] - first);
            r = make_uint2(0x80000000u | first, mask);
        }
    }
    out[u] = r;
}

int ensure_urec(rk_ctx *ctx, rk_index *idx, hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(idx->lazy_mu);
    // bit 31 tags a compact record: posting offsets and genome ids must stay below it; a genome that sits twice in a list
    // (sketches with repeats) cannot be a bit
    if (idx->d_urec || !idx->U || !idx->ref_sets || idx->H >= (1ULL << 31) || idx->n_ref >= (1u << 31)) return RK_OK;
    DevBuf<uint2> out(ctx);
    if (out.alloc(idx->U) != hipSuccess) return rk_fail(ctx, RK_ERR_NOMEM, "cannot allocate %llu list records", (unsigned long long)idx->U);
    hipLaunchKernelGGL(k_urec, dim3((unsigned)((idx->U + 255) / 256)), dim3(256), 0, stream, idx->d_upos, idx->d_postings, idx->U, out.p);
    RK_HIP(ctx, hipGetLastError());
    RK_HIP(ctx, hipStreamSynchronize(stream));  // once per index: a later call may come on another stream
    idx->d_urec = out.release();
    return RK_OK;
}

constexpr int kRankMaxBits = 30;  // 2^30 / 48 entries x 8 B = 171 MiB; above that: directory + binary search

int ensure_rankbm(rk_ctx *ctx, rk_index *idx, hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(idx->lazy_mu);
    if (idx->d_rankbm || idx->wide || idx->hash_bits > kRankMaxBits) return RK_OK;
    const uint64_t n_entries = ((1ULL << idx->hash_bits) + kRankSpan - 1) / kRankSpan;
    DevBuf<uint2> out(ctx);
    DevBuf<uint32_t> base(ctx);
    if (out.alloc(n_entries) != hipSuccess || base.alloc(n_entries / kRankBlock + 1) != hipSuccess)
        return rk_fail(ctx, RK_ERR_NOMEM, "cannot allocate the %llu-entry rank bitmap", (unsigned long long)n_entries);
    hipLaunchKernelGGL(k_rank_fill, dim3((unsigned)((n_entries + 255) / 256)), dim3(256), 0, stream, idx->d_uhash, idx->U,
                       n_entries, out.p, base.p);
    RK_HIP(ctx, hipGetLastError());
    RK_HIP(ctx, hipStreamSynchronize(stream));  // once per index: a later call may come on another stream
    idx->d_rankbm = out.release();
    idx->d_rankbase = base.release();
    return RK_OK;
}

struct PlanQ {
    int cbits, look;
    uint32_t tile_cols, n_tiles, cnt_words, threads, cand_cap, stage_hits, n_units, grid;
    uint32_t row_first, row_step, row_block;
    size_t lds_bytes;
};

constexpr uint32_t kCandCap = 256, kStageHits = 96;
size_t distq_fixed_bytes(uint32_t threads)
{
    return (size_t)(threads / 64) * kQueueCap * sizeof(uint2) + (size_t)kCandCap * sizeof(uint2) + (size_t)kStageHits * sizeof(rk_hit) + 64;
}
// one tile if the counter row fits next to the queues of a 1024-thread workgroup, else equal tiles
void plan_tiles(const rk_ctx *ctx, const rk_index *idx, int cbits, uint32_t *tile_cols, uint32_t *n_tiles, uint32_t *cnt_words)
{
    const size_t lds_max = std::min<size_t>(ctx->max_lds, 160 * 1024);
    const size_t row_cap = lds_max - distq_fixed_bytes(kMaxThreads);
    const uint32_t max_cols = (uint32_t)(row_cap * 8 / cbits) & ~127u;
    uint32_t tile = idx->n_ref;
    if (tile > max_cols) {
        const uint32_t nt = (idx->n_ref + max_cols - 1) / max_cols;
        tile = ((idx->n_ref + nt - 1) / nt + 127) & ~127u;
    }
    *tile_cols = tile;
    *n_tiles = (idx->n_ref + tile - 1) / tile;
    *cnt_words = (uint32_t)((((uint64_t)tile * cbits + 31) / 32 + 3) & ~3ULL);  // whole 16-byte quads
}
// the pipelined look-up (PIPE) where the counter row alone limits a CU to one or two workgroups
bool distq_pipe(const rk_ctx *ctx, const rk_index *idx, int cbits, int look)
{
    if (look != kLookRank || (getenv("RK_DISTQ_PIPE") && !atoi(getenv("RK_DISTQ_PIPE")))) return false;
    uint32_t tile = 0, nt = 0, words = 0;
    plan_tiles(ctx, idx, cbits, &tile, &nt, &words);
    return (size_t)words * 4 >= 64 * 1024;
}

// the sliced membership pass + the counting kernel on its records (sorted queries against an index with a rank bitmap).
// NOT the default: built and measured in round 4 on the shape it was costed for (BASELINE configs[4], 1,000 queries of 45,776
// hashes against 100,000 references) -- k_member_sliced 125 us + the counting kernel 118 us = 0.244 ms against 0.215 ms of the
// fused kernel with the pipelined look-up.  The slices do what they were meant to (3.6 M requests from L1 to L2 instead of
// 27 M), but the two kernels issue 77.8 M vector instructions where the fused one issues 60.7 M (61 per 64 look-ups for the
// cursor, the slice bounds and the record list alone), and the counting half -- the walks, 34 M -- is as vector-bound as
// before.  RK_DISTQ_SLICED=1 takes this path whenever it is possible (tests/test_gpu_parity.py runs it against the oracle).
bool distq_sliced(const rk_ctx *ctx, const rk_index *idx, const rk_sketches *qs, int cbits, int look)
{
    (void)ctx; (void)idx; (void)cbits;
    const char *e = getenv("RK_DISTQ_SLICED");
    if (look != kLookRank || !qs->is_set || qs->wide || !qs->n || qs->total >= 0xFFF00000ULL) return false;
    return e && *e && atoi(e) == 1;
}

typedef void (*distq_kernel_t)(DistQArgs);
distq_kernel_t pick_kernel(int cbits, int look, bool pipe)
{
#define RK_Q(C)                                                                                                  \
    (look == kLookRank ? (pipe ? rk_distq_kernel<C, kLookRank, true> : rk_distq_kernel<C, kLookRank, false>)     \
                       : (look == kLookDir32 ? rk_distq_kernel<C, kLookDir32, false> : rk_distq_kernel<C, kLookDir64, false>))
    if (look == kLookPre) return cbits == 8 ? rk_distq_kernel<8, kLookPre, false> : (cbits == 16 ? rk_distq_kernel<16, kLookPre, false> : rk_distq_kernel<32, kLookPre, false>);
    return cbits == 8 ? RK_Q(8) : (cbits == 16 ? RK_Q(16) : RK_Q(32));
#undef RK_Q
}

}  // namespace

static int counter_bits(const rk_index *idx, const rk_sketches *qs)
{
    // an intersection count never exceeds the smaller sketch when both sides are sets; a query with repeated
    // hashes still cannot exceed its own length as long as the references are sets
    uint64_t bound = ~0ULL;
    if (idx->ref_sets) bound = qs->is_set ? std::min<uint64_t>(qs->max_size, idx->max_ref_size) : qs->max_size;
    return bound < 256 ? 8 : (bound < 65536 ? 16 : 32);
}

int rk_distq_kernel_name(rk_ctx *ctx, const rk_index *idx, const rk_sketches *qs, char *buf, size_t cap)
{
    const int look = idx->wide ? kLookDir64 : (idx->hash_bits <= kRankMaxBits ? kLookRank : kLookDir32);
    const int cbits = counter_bits(idx, qs);
    if (distq_sliced(ctx, idx, qs, cbits, look)) snprintf(buf, cap, "rk_distq_kernel<%d, %d, false>", cbits, (int)kLookPre);
    else snprintf(buf, cap, "rk_distq_kernel<%d, %d, %s>", cbits, look, distq_pipe(ctx, idx, cbits, look) ? "true" : "false");
    return RK_OK;
}

int rk_distq_launch(rk_ctx *ctx, const rk_index *idx, const rk_sketches *qs, const rk_dist_opts *o, bool dense_mode,
                    rk_hit *hits_dev, uint64_t cap, unsigned long long *n_hits_dev, int32_t *dense_dev, hipStream_t stream)
{
    if (o->kmer_size <= 0) return rk_fail(ctx, RK_ERR_ARG, "kmer_size must be positive");
    if (o->row_block < 0) return rk_fail(ctx, RK_ERR_ARG, "row_block must be >= 0");
    if (qs->wide != idx->wide) return rk_fail(ctx, RK_ERR_ARG, "query sketches and index use different hash widths");
    if (!qs->n || !idx->n_ref) return RK_OK;
    if (idx->H >= 0x7FFFFFFFULL)   // (bit 31 of a list record tags its compact form: posting offsets stay below it)
        return rk_fail(ctx, RK_ERR_UNSUPPORTED, "explicit queries against an index of 2^31-1 postings or more");
    PlanQ p;
    p.cbits = counter_bits(idx, qs);
    int rc = ensure_rankbm(ctx, const_cast<rk_index *>(idx), stream);  // lazily built, cached in the index
    if (rc) return rc;
    rc = ensure_urec(ctx, const_cast<rk_index *>(idx), stream);
    if (rc) return rc;
    p.look = idx->wide ? kLookDir64 : (idx->d_rankbm ? kLookRank : kLookDir32);
    if (p.look != kLookRank) {  // prefix directory, lazily built like the rank bitmap
        rc = rk_index_ensure_dir(ctx, const_cast<rk_index *>(idx), stream);
        if (rc) return rc;
    }
    p.cand_cap = kCandCap;
    p.stage_hits = kStageHits;
    const size_t lds_max = std::min<size_t>(ctx->max_lds, 160 * 1024);
    auto fixed_bytes = [&](uint32_t threads) { return distq_fixed_bytes(threads); };
    plan_tiles(ctx, idx, p.cbits, &p.tile_cols, &p.n_tiles, &p.cnt_words);
    // workgroup size follows the LDS footprint (it caps the resident workgroups): 7 x 256, 3 x 512, 2 x 768, 1 x 1024
    // workgroup size: the one that keeps most waves resident per CU (registers admit ~5 waves per SIMD, LDS comes in
    // 1,280-byte granules -- the runtime's occupancy answer, clamped by that rule); ties go to the bigger workgroup.
    // Small counter rows therefore run as many 256-thread workgroups (10,000 16-bit columns: 5 per CU instead of 2 x 512).
    const size_t row_bytes = (size_t)p.cnt_words * 4;
    const bool sliced = distq_sliced(ctx, idx, qs, p.cbits, p.look);
    distq_kernel_t kern_for_plan = sliced ? pick_kernel(p.cbits, kLookPre, false) : pick_kernel(p.cbits, p.look, distq_pipe(ctx, idx, p.cbits, p.look));
    auto resident_wgs = [&](uint32_t threads) -> int {
        const size_t lds = row_bytes + fixed_bytes(threads);
        if (lds > lds_max) return 0;
        if (lds > 48 * 1024 &&
            hipFuncSetAttribute((const void *)kern_for_plan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        const int api = rk_occupancy(ctx, (const void *)kern_for_plan, (int)threads, lds);
        const size_t granule = 1280;
        return std::min<int>(api, (int)((160 * 1024) / ((lds + granule - 1) / granule * granule)));
    };
    p.threads = 1024;
    int best_waves = -1;
    for (uint32_t t : {256u, 512u, 768u, 1024u}) {
        const int waves = resident_wgs(t) * (int)(t / 64);
        if (waves >= best_waves && waves > 0) { best_waves = waves; p.threads = t; }
    }
    p.lds_bytes = row_bytes + fixed_bytes(p.threads);
    p.row_step = o->row_step ? o->row_step : 1;
    p.row_first = o->row_first;
    p.row_block = o->row_block > 0 ? (uint32_t)o->row_block : 1;
    const uint64_t n_blocks = ((uint64_t)qs->n + p.row_block - 1) / p.row_block;
    const uint64_t my_blocks = p.row_first < n_blocks ? (n_blocks - p.row_first + p.row_step - 1) / p.row_step : 0;
    p.n_units = (uint32_t)std::min<uint64_t>(my_blocks * p.row_block, 0xFFFFFFF0u / p.n_tiles);
    if (!p.n_units) return RK_OK;

    DistQArgs a;
    a.q_hashes = qs->wide ? (const void *)qs->d_hashes64 : (const void *)qs->d_hashes;
    a.q_off = qs->d_off;
    a.rankbm = idx->d_rankbm;
    a.rankbase = idx->d_rankbase;
    a.urec = idx->d_urec;
    a.uhash = idx->wide ? (const void *)idx->d_uhash64 : (const void *)idx->d_uhash;
    a.dir = idx->d_dir;
    a.upos = idx->d_upos;
    a.postings = idx->d_postings;
    a.ref_sizes = idx->d_sizes;
    a.orig = idx->relabeled ? idx->d_orig : nullptr;
    a.hash_bits = idx->hash_bits;
    a.dir_shift = idx->dir_shift;
    a.n_query = qs->n;
    a.n_ref = idx->n_ref;
    a.row_first = p.row_first;
    a.row_step = p.row_step;
    a.row_block = p.row_block;
    a.n_units = p.n_units;
    a.tile_cols = p.tile_cols;
    a.n_tiles = p.n_tiles;
    a.cnt_words = p.cnt_words;
    a.cand_cap = p.cand_cap;
    a.stage_hits = p.stage_hits;
    a.triangle = o->triangle;
    a.metric = o->metric != 0;  // the reference treats any non-zero isContainment as containment
    a.kmer_size = o->kmer_size;
    a.dense_mode = dense_mode ? 1 : 0;
    a.max_dist = o->max_dist;
    a.min_jorc = 0.0;
    a.min_ref_size = (uint32_t)std::min<uint64_t>(idx->min_ref_size, 0xFFFFFFFFu);
    if (!a.dense_mode && o->max_dist > 0.0) {
        // distance < D  <=>  jaccard > t/(2-t), t = exp(-k D)  (containment: c > t); 1e-6 relative slack keeps the
        // reject conservative, the exact formula still decides
        const double t = exp(-(double)o->kmer_size * o->max_dist);
        a.min_jorc = (a.metric ? t : t / (2.0 - t)) * (1.0 - 1e-6);
    }
    a.hits = hits_dev;
    a.cap = cap;
    a.n_hits = n_hits_dev;
    a.common_dense = dense_dev;

    distq_kernel_t kern = kern_for_plan;
    if (p.lds_bytes > 48 * 1024)
        RK_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes));
    // persistent grid: as many workgroups as the chip holds, in whole rounds of 8 XCDs x n_tiles
    const int per_cu = resident_wgs(p.threads);
    const uint32_t per_round = 8 * p.n_tiles;
    const uint32_t total = p.n_units * p.n_tiles;
    const uint32_t total_padded = (total + per_round - 1) / per_round * per_round;
    uint32_t resident = (uint32_t)std::max(1, per_cu) * (uint32_t)std::max(1, ctx->num_cu);
    resident = std::max(per_round, resident / per_round * per_round);
    p.grid = std::min(total_padded, resident);
    a.min_fit = getenv("RK_DISTQ_MIN_FIT") ? (uint32_t)atoi(getenv("RK_DISTQ_MIN_FIT")) : 8u;
    a.rec = a.seg_start = a.seg_cnt = nullptr;
    if (sliced) {
        rk_sketches *q = const_cast<rk_sketches *>(qs);
        {
            std::lock_guard<std::mutex> lk(q->lazy_mu);
            if (!q->d_member_rec) {
                DevBuf<uint32_t> rec(ctx), seg(ctx);
                if (rec.alloc(qs->total + 1) != hipSuccess || seg.alloc((size_t)2 * kSegStride * qs->n) != hipSuccess)
                    return rk_fail(ctx, RK_ERR_NOMEM, "cannot allocate the membership records of %llu query hashes", (unsigned long long)qs->total);
                q->d_member_rec = rec.release();
                q->d_member_seg = seg.release();
            }
        }
        MemberArgs m;
        m.q_hashes = qs->d_hashes;
        m.q_off = qs->d_off;
        m.n_query = qs->n;
        m.rankbm = idx->d_rankbm;
        m.rankbase = idx->d_rankbase;
        m.n_entries = ((1ULL << idx->hash_bits) + kRankSpan - 1) / kRankSpan;
        m.n_slices = (uint32_t)((m.n_entries + kSliceEntries - 1) / kSliceEntries);
        // slice ranges per batch of 16 queries: enough workgroups for the chip, at most kSegStride segments per query
        const uint32_t batches = (qs->n + kMemberWaves - 1) / kMemberWaves;
        uint32_t ranges = 1;
        while (ranges < kSegStride && ranges < m.n_slices && (uint64_t)batches * (ranges + 1) <= 2ULL * (uint64_t)ctx->num_cu) ranges++;   // (two workgroups fit a CU: one round)
        m.slices_per_range = (m.n_slices + ranges - 1) / ranges;
        ranges = (m.n_slices + m.slices_per_range - 1) / m.slices_per_range;
        m.debug = getenv("RK_MEMBER_DEBUG") ? atoi(getenv("RK_MEMBER_DEBUG")) : 0;
        m.rec = q->d_member_rec;
        m.seg_start = q->d_member_seg;
        m.seg_cnt = q->d_member_seg + (size_t)kSegStride * qs->n;
        RK_HIP(ctx, hipMemsetAsync(q->d_member_seg, 0, (size_t)2 * kSegStride * qs->n * 4, stream));   // (ranges a launch does not use count zero)
        hipLaunchKernelGGL(k_member_sliced, dim3(batches, ranges), dim3(kMemberWaves * 64), 0, stream, m);
        RK_HIP(ctx, hipGetLastError());
        a.rec = m.rec;
        a.seg_start = m.seg_start;
        a.seg_cnt = m.seg_cnt;
    }
    hipLaunchKernelGGL(kern, dim3(p.grid), dim3(p.threads), p.lds_bytes, stream, a);
    RK_HIP(ctx, hipGetLastError());
    return RK_OK;
}
