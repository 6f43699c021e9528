// rk_internal.h -- shared declarations of librabbitkssd.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "rabbitkssd.h"

struct rk_ctx {
    int device = 0;
    int num_cu = 0;
    size_t max_lds = 0;  // bytes of LDS one workgroup may use
    std::string err;
    // the context's own stream: every synchronous API call enqueues here (never on the null stream, whose implicit
    // synchronisation would serialise the caller's other streams) and synchronises it before returning
    hipStream_t stream = nullptr;
    // a second stream + two events (created on first use): rk_index_build computes the internal genome order there while
    // the partition of the hashes runs on `stream`
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_inv = nullptr;   // ev_inv: the genome translation table of the renumbering is ready (the rest of it -- sizes, offsets -- is not needed by the bucket emission of tile records)
    void *pinned = nullptr;  // page-locked scratch for small read-backs / uploads (grow-only, see rk_pinned_scratch)
    size_t pinned_bytes = 0;
    // caching device allocator: hipMalloc / hipFree cost 50-300 us each (hipFree also synchronises the device) and the
    // index build alone needs ~25 buffers, so freed blocks are kept and handed out again; steady-state calls
    // allocate nothing.  Blocks return to the driver in rk_ctx_trim / rk_ctx_destroy.
    std::mutex mu;
    std::multimap<size_t, void *> free_blocks;
    std::unordered_map<void *, size_t> live;  // every block handed out or cached -> its size
    size_t pool_bytes = 0;           // everything obtained from the driver
    size_t cached_bytes = 0;         // of which idle in free_blocks
    uint64_t driver_allocs = 0, driver_frees = 0;  // hipMalloc / hipFree calls made by the pool
    // blocks whose last user is still running on some stream (rk_pool_free_after: the scratch of a rocprim call): they return to
    // free_blocks once their event has passed -- looked at whenever the pool is asked for memory
    std::vector<std::pair<void *, hipEvent_t>> deferred;
    std::vector<hipEvent_t> spare_events;
    size_t cache_limit = 0;          // idle bytes above this go straight back to the driver (RK_POOL_LIMIT_MB, default 32 GiB)
    std::map<std::tuple<const void *, int, size_t>, int> occupancy;  // hipOccupancy... costs 10-70 us per query
    // optional HIP-event timing of the dominant kernel of a pass (rk_ctx_set_timing): [0] sketch kernel
    bool single_shot = false;  // rk_ctx_set_single_shot: a command-line run -- avoid device-side paths whose one-time setup (a code
                               // object load) costs more than they save on one call
    bool timing = false;
    hipEvent_t ev[2] = {nullptr, nullptr};
    double last_ms[4] = {0, 0, 0, 0};
    // developer switches (environment), read once at context creation
    uint32_t sw_dist_threads = 0, sw_dist_rows = 0, sw_dist_pair = 1, sw_dist_pair_minwg = 3, sw_dist_persist = 1;
    uint32_t sw_dist_cand_cap = 0, sw_dist_stage_hits = 0, sw_dist_xcd_rows = 0;
    int sw_dist_bands = 1;  // RK_DIST_BANDS=0: the self join in one launch, every row as wide as the whole collection
    int sw_dist_band_min_rows = 3072;  // RK_DIST_BAND_MIN_ROWS: no band with fewer rows of the shard than this (tests: small collections in several bands)
    int sw_dist_near = 1;   // RK_DIST_NEAR=0: the self join always with full counter rows (rk_dist_kernel)
    int sw_dist_near_uw = 0;    // RK_DIST_NEAR_UW=1|2|4: waves that share a unit of the near-window kernel (default: by the launch's size)
    int sw_dist_fb_skip = 1;   // RK_DIST_FB_SKIP=0: always launch the fallback pass of the near-window self join
    int sw_dist_tiles = 2;      // RK_DIST_TILES: 0 never the tile kernel (rk_dist_tile.inc), 1 for every sparse self join over sets, 2 by the index's size and shape (rk_dist.hip self_uses_tiles)
    // RK_DIST_TILES_MIN_GENOMES: from this many genomes on rk_index_build emits tile records (not slice records) and the self join
    // runs on the tile kernel from its first launch; _MIN_SHARD_ROWS: row shards smaller than this prefer the near-window kernel
    // when the index HAS slice records (a tile costs the same whatever the shard)
    int sw_dist_tiles_min_genomes = 4000, sw_dist_tiles_min_shard_rows = 12000;
    int sw_dist_near_min = 16;  // RK_DIST_NEAR_MIN: the near-window kernel is used when a reportable pair of the smallest sketch needs at least this count
    int sw_dist_debug = 0;  // RK_DIST_DEBUG=1: the bands of every self join on stderr
    int sw_dist_lds_kb = 0;  // RK_DIST_LDS_KB: plan as if a CU had this much LDS (tests: tiled bands at small sizes)
    // RK_SKETCH_IMG: 2 = the two-stage scan (rk_sketch_scan2.inc; parameter sets without a compile-time variant fall
    // back to 1), 1 = rk_sketch_kernel with the 64 KiB LDS image, 0 = with the 144 KiB image and the exact table
    int sw_sketch_img = 2;
    int sw_index_fast = 1;     // RK_INDEX_FAST=0: always the general (device-wide radix sort) build
    int sw_index_relabel = 1;  // RK_INDEX_RELABEL=0: keep the caller's genome order inside the index
    int sw_index_no_self = 0;  // RK_INDEX_NO_SELF=1: build every index without slice records (as one of 2^31 postings and more is)
    int sw_index_heavy = 1;    // RK_INDEX_NO_HEAVY=1: a bucket beyond the LDS sort refuses the bucket-sort build (as until round 5) instead of going to k_bucket_heavy
    int sw_index_tiles = 2;    // RK_INDEX_TILES: 0 the fast build always emits slice records, 1 tile records whenever it can, 2 from RK_DIST_TILES_MIN_GENOMES genomes on
    unsigned long long sw_tile_rec_cap = 0;   // RK_TILE_REC_CAP: capacity of the build's unsorted tile records (default H / 2 + 64 K; tests force the overflow)
};
constexpr size_t kPinnedBytes = 1 << 16;

int rk_fail(rk_ctx *ctx, int code, const char *fmt, ...);
void *rk_pool_alloc(rk_ctx *ctx, size_t bytes);  // nullptr when the device is out of memory
void rk_pool_free(rk_ctx *ctx, void *p);
// the block goes back to the pool once everything enqueued on `st` so far has run (an event): for temporaries of work that is still
// in flight when the host function returns -- the pool itself is not stream-aware, and another host thread of the same context may
// ask for memory in the meantime
void rk_pool_free_after(rk_ctx *ctx, void *p, hipStream_t st);
// device -> host copy of a few bytes through the pinned scratch (a pageable hipMemcpy costs 20-30 us more);
// synchronises `stream`
int rk_read_back(rk_ctx *ctx, void *dst, const void *src_dev, size_t bytes, hipStream_t stream);
int rk_occupancy(rk_ctx *ctx, const void *kernel, int threads, size_t lds_bytes);
// the context's page-locked scratch, grown to at least `bytes` (contents are not preserved); nullptr on failure
void *rk_pinned_scratch(rk_ctx *ctx, size_t bytes);

#define RK_HIP(ctx, call)                                                                   \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess)                                                              \
            return rk_fail((ctx), RK_ERR_HIP, "%s failed: %s (%s:%d)", #call,               \
                           hipGetErrorString(e__), __FILE__, __LINE__);                     \
    } while (0)

// owning device pointer from the context's pool; returned to it on scope exit unless release()d.  Work that uses
// the buffer must be enqueued on ctx->stream (reuse is stream-ordered) or be complete before the scope ends.
template <class T> struct DevBuf {
    rk_ctx *ctx;
    T *p = nullptr;
    explicit DevBuf(rk_ctx *c) : ctx(c) {}
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { reset(); }
    hipError_t alloc(size_t n) {
        reset();
        p = static_cast<T *>(rk_pool_alloc(ctx, (n ? n : 1) * sizeof(T)));
        return p ? hipSuccess : hipErrorOutOfMemory;
    }
    void reset() {
        if (p) rk_pool_free(ctx, p);
        p = nullptr;
    }
    T *release() {
        T *q = p;
        p = nullptr;
        return q;
    }
    operator T *() const { return p; }
};

struct rk_filter {
    rk_ctx *ctx = nullptr;
    rk_params params{};
    int32_t *d_table = nullptr;   // the full .shuf table, int32[16^half_subk]
    uint32_t *d_bitmap = nullptr; // 64 KiB LDS filter image (bitmap [+ exact key/value table])
    int bitmap_bits = 0;
    bool exact = false;           // image holds the exact table: survivors never leave the CU
    int img = 0;                  // LDS image variant (rk_sketch.hip Img<>)
    uint32_t n_keys = 0;          // entries with value in [dim_start, dim_end)
};

struct rk_sketches {
    rk_ctx *ctx = nullptr;
    uint32_t n = 0;
    uint64_t total = 0;
    uint64_t windows = 0;
    bool wide = false;            // 64-bit hash layout (use64: half_k - drlevel > 8)
    uint32_t *d_hashes = nullptr; // u32[total] when !wide
    uint64_t *d_hashes64 = nullptr; // u64[total] when wide
    uint64_t *d_off = nullptr;
    std::vector<uint64_t> h_off;  // host mirror of d_off (n+1)
    bool is_set = false;          // every genome's hashes are strictly ascending (sorted, no repeats)
    uint64_t max_size = 0;        // largest sketch
    // scratch of the sliced membership pass (rk_distq.hip), kept with the QUERIES it belongs to: per query hash room for one
    // 4-byte record (the rank of a present hash), per query and slice range the place and number of its records.  One
    // query join per sketches object at a time may use it (the header says so).
    std::mutex lazy_mu;
    uint32_t *d_member_rec = nullptr;   // u32[total]
    uint32_t *d_member_seg = nullptr;   // u32[2 * 8 * n]: starts, then counts
};

struct rk_index {
    rk_ctx *ctx = nullptr;
    uint32_t n_ref = 0;
    uint64_t H = 0;         // postings
    uint64_t U = 0;         // distinct hashes
    int hash_bits = 0;
    int dir_bits = 0, dir_shift = 0;
    uint64_t sum_sq = 0;             // sum of squared list lengths, computed on first request
    bool sum_sq_known = false;
    uint64_t self_stats[4] = {0, 0, 0, 0};  // rk_index_self_stats, computed on first request
    bool self_stats_known = false;
    uint64_t max_src_size = 0;       // largest source sketch (built index only)
    uint64_t max_ref_size = 0;       // largest reference sketch (every index)
    uint64_t min_ref_size = 0;       // smallest NON-EMPTY reference sketch (0: all empty): lower bound of a containment denominator
    bool ref_sets = false;           // no genome appears twice in a posting list (the sketches are sets): an
                                     // intersection count is then bounded by the smaller sketch
    uint2 *d_rankbm = nullptr;       // lazily built by the query path (rk_distq.hip): per 48 consecutive hash values
                                     // {presence bits 0..31, presence bits 32..47 | 16-bit rank << 16}; the rank (number
                                     // of distinct indexed hashes below) is relative to d_rankbase[entry / 64]
    uint32_t *d_rankbase = nullptr;  // u32[entries / 64 + 1]
    uint2 *d_urec = nullptr;         // lazily built by the query path: per distinct hash its posting range (upos[u], upos[u+1]) or, when
                                     // the list spans < 32 genome ids, the list itself: (bit 31 | first genome, bitmask of first..first+31)
    uint32_t *d_postings = nullptr;  // u32[H]   (.dict order)
    bool wide = false;               // 64-bit hashes: d_uhash64 instead of d_uhash
    uint32_t *d_uhash = nullptr;     // u32[U]   sorted distinct hashes
    uint64_t *d_uhash64 = nullptr;   // u64[U]   (wide)
    uint32_t *d_upos = nullptr;      // u32[U+1] posting offsets
    uint32_t *d_dir = nullptr;       // u32[2^dir_bits+1] prefix directory into d_uhash
    uint32_t *d_sizes = nullptr;     // u32[n_ref] sketch sizes
    uint2 *d_selfrange = nullptr;    // uint2[n_self]: per source element with a non-empty slice,
                                     // the postings of LATER genomes (rows back to back)
    uint64_t *d_self_off = nullptr;  // u64[n_ref+1] offsets of each genome's slices in d_selfrange
    uint64_t *d_self_split = nullptr; // u64[n_ref]: a row's slices from here on are "covered" by its pair partner
                                      // (genome 2p+1 shares the hash with 2p): the pair kernel skips them
    uint64_t n_self = 0;
    uint64_t *d_src_off = nullptr;   // u64[n_ref+1] offsets of the source sketches (built only), in INTERNAL genome order
    // Internal genome order.  rk_index_build renumbers the genomes so that relatives (genomes that share several of their
    // smallest hashes) get neighbouring ids, whatever order the collection arrived in: compact slices / list records
    // (all members within 32 ids) and row pairs depend on that.  Every device array above is in internal ids; d_orig
    // maps an internal id back to the caller's genome index (null: identity -- imported indexes), and the kernels apply
    // it where a hit record or a dense counter row leaves the device.
    uint32_t *d_orig = nullptr;      // u32[n_ref]
    bool relabeled = false;          // d_orig may differ from the identity
    bool built_fast = false;         // built by the bucket-sort path (rk_index_fast.inc)
    uint32_t *d_fb = nullptr;        // fallback list of the near-window self join (rk_dist_near.inc): [0] count, [1] done, [4..] rows;
                                     // allocated on first use; one self join per index in flight at a time
    uint32_t *h_fb_seen = nullptr;   // page-locked host word: rows the last fallback launch found in the list (a hint for the next
                                     // launch's grid: an empty list is the rule, and an empty launch should be a small one)
    // The fallback list of a self join is a function of the index and the options alone (the kernels are deterministic): once
    // a launch with these options has COMPLETED (fb_event) with an empty list, later launches with the same options skip the
    // empty fallback launch (3 us of a 15 us row shard).  fb_state: 0 nothing known for fb_key, 1 a launch is in flight,
    // 2 known empty, 3 known non-empty.  Guarded by lazy_mu.
    unsigned char fb_key[40] = {0};
    int fb_state = 0;
    void *fb_event = nullptr;        // hipEvent_t
    // Tile records of the self join over 32 x 32 tiles (rk_dist_tile.inc), built on first use: per tile (block b of rows,
    // block w >= b of columns) the (row mask, column mask) pairs of the posting lists that touch both blocks, sorted by tile
    uint2 *d_tile_contrib = nullptr;            // uint2[n_tile_records + pad]: (row mask, column mask), sorted by tile
    uint32_t *d_tile_rows = nullptr, *d_tile_cols = nullptr;   // u32[n_tile_slots + 256] each: a split copy of the records in which every tile starts at an EVEN slot (an odd tile is padded with one empty record): the SROW variant of the kernel (short launches) reads the row masks of two neighbouring records as one 64-bit scalar
    uint64_t n_tile_slots = 0, tile_max_records = 0;   // (the biggest tile)
    unsigned long long *d_tile_key = nullptr;   // u64[n_tiles]: b << 32 | w, ascending
    unsigned long long *d_tile_start = nullptr; // u64[n_tiles + 1]
    uint32_t *d_blk_min = nullptr;              // u32[ceil(n_ref / 32)]: smallest non-empty sketch of the block
    uint4 *d_tile_dir[2] = {nullptr, nullptr};       // the directory in launch order, one 32-byte entry per tile (rk_tiles.hip k_tile_dir): all a workgroup needs in one access
    uint32_t *d_tile_order[2] = {nullptr, nullptr};  // tile numbers by records per smallest sketch, descending: [0] jaccard, [1] containment
    unsigned long long tile_prefix[2][256] = {};     // (kTileTable entries each)     // [metric][k]: tiles with at least 2^(-k/8) records per smallest sketch
    uint64_t n_tiles = 0, n_tile_records = 0;
    bool tiles_ready = false;
    bool tiles_from_build = false;   // the tile records came with rk_index_build (rk_index_tiles.inc), not from the lazy rk_tiles_build
    // a SHARD of a multi-GPU build (rk_index_build_shard): the lists of one range of the hash space; its tile records, grouped by
    // the shard that owns their row block, wait here for the exchange (rk_index_shard_records / rk_index_shard_pack)
    uint3 *d_shard_rec = nullptr;
    uint32_t shard_region_cap = 0, n_shards = 0, shard_id = 0;
    unsigned long long shard_rec_count[64] = {};
    bool slices_refused = false;     // built without slice records on purpose (2^31 postings and more, RK_INDEX_NO_SELF): none on first use either
    bool tiles_unusable = false;     // rk_tiles_build found more records than its budget: the self join stays with the row kernels
    int spread_known = 0;            // 1: `spread` below is valid (rk_dist.hip self_uses_tiles)
    bool spread = false;             // many related lists span more than the 32-column window of rk_near_kernel
    std::mutex lazy_mu;              // serialises the lazy builders (prefix directory, rank bitmap, list records, sum of
                                     // squares): two host threads may query one index
};

// device-wide primitives behind plain functions (rk_prims.hip, rk_prims_hits.hip: translation units of their own, so that
// the code objects of the callers stay small -- a code object is loaded when the first kernel of its translation unit runs).
// All enqueue on `st`; temporaries come from the context's pool and return to it when the wrapper returns.
int rk_prim_sort_keys_u64(rk_ctx *ctx, const unsigned long long *in, unsigned long long *out, uint64_t n, unsigned begin_bit, unsigned end_bit,
                          hipStream_t st, void **tmp_keep = nullptr);   // tmp_keep: the scratch block is handed to the caller (rk_pool_free)
int rk_prim_sort_pairs_u32_u32(rk_ctx *ctx, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, uint64_t n, unsigned end_bit,
                               hipStream_t st);
int rk_prim_sort_pairs_u64_u32(rk_ctx *ctx, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout, uint64_t n, unsigned end_bit,
                               hipStream_t st);
int rk_prim_inclusive_scan_u32(rk_ctx *ctx, const uint32_t *in, uint32_t *out, uint64_t n, hipStream_t st);
int rk_prim_exclusive_scan_u32(rk_ctx *ctx, const uint32_t *in, uint32_t *out, uint64_t n, hipStream_t st);
int rk_prim_unique_u64(rk_ctx *ctx, const unsigned long long *in, unsigned long long *out, unsigned long long *n_out_dev, uint64_t n, hipStream_t st);
int rk_prim_rle_u64(rk_ctx *ctx, const unsigned long long *in, uint64_t n, unsigned long long *runs, unsigned int *counts, unsigned long long *n_runs_dev,
                    hipStream_t st);
int rk_prim_select_u64(rk_ctx *ctx, const unsigned long long *in, const unsigned char *flags, unsigned long long *out, unsigned long long *n_out_dev,
                       uint64_t n, hipStream_t st);
int rk_prim_segmented_sort_u32(rk_ctx *ctx, const uint32_t *in, uint32_t *out, unsigned n, unsigned segments, const uint64_t *off, hipStream_t st);
int rk_prim_segmented_sort_u64(rk_ctx *ctx, const uint64_t *in, uint64_t *out, unsigned n, unsigned segments, const uint64_t *off, hipStream_t st);
int rk_prim_sort_hits(rk_ctx *ctx, const unsigned long long *keys, unsigned long long *keys_out, const rk_hit *hits, rk_hit *hits_out, uint64_t n,
                      unsigned end_bit, hipStream_t st);   // synchronises `st`
// tile records of the self join over 32 x 32 tiles, built on first use and cached in the index (rk_tiles.hip)
constexpr int kTileTable = 256;   // entries of rk_index::tile_prefix per metric
int rk_tiles_build(rk_ctx *ctx, rk_index *idx, hipStream_t stream);
// slice records of an index that was built without them (tile records instead), on first use: rk_near_kernel on small row
// shards, dense reports (rk_index.hip)
int rk_index_ensure_slices(rk_ctx *ctx, rk_index *idx, hipStream_t stream);
// prefix directory into the sorted distinct hashes, built on first use (rk_index.hip)
int rk_index_ensure_dir(rk_ctx *ctx, rk_index *idx, hipStream_t stream);
// explicit queries (index_dist): lookup + counting + epilogue in one kernel (rk_distq.hip).  Enqueues on `stream`,
// allocates nothing once the index's rank bitmap exists.  dense_dev: optional int32[n_query * n_ref].
// dense_mode: every cell is reportable (rk_dense_mode of the caller's EXACT options).
int rk_distq_launch(rk_ctx *ctx, const rk_index *idx, const rk_sketches *queries, const rk_dist_opts *opts, bool dense_mode,
                    rk_hit *hits_dev, uint64_t cap, unsigned long long *n_hits_dev, int32_t *dense_dev,
                    hipStream_t stream);
// the kernel variant rk_distq_launch would pick, as a profiler prints it
int rk_distq_kernel_name(rk_ctx *ctx, const rk_index *idx, const rk_sketches *queries, char *buf, size_t cap);
// true when every posting is reportable regardless of its count (the threshold admits distance 1.0)
inline bool rk_dense_mode(const rk_dist_opts *o) { return o->triangle ? (1.0 < o->max_dist) : (1.0 <= o->max_dist); }
// sets s->is_set / s->max_size from the device arrays (one small kernel + a 4-byte read-back)
int rk_sketches_classify(rk_ctx *ctx, rk_sketches *s);
