/*
 * rabbitkssd.h -- C ABI of the MI355X-native sketch + distance engine (librabbitkssd.so).
 *
 * Drop-in boundary for RabbitKSSD's hot path.  The reference has no FFI of its own; the
 * seams this ABI replaces are four C++ free functions and the arithmetic inside them
 * (citations are file:line into the reference tree):
 *
 *   rk_params_init         <- initParameter             src/common.cpp:35-78
 *   rk_filter_create       <- shuffled_map construction  src/sketch.cpp:336-345
 *   rk_sketch_batch        <- per-genome loop of sketchFastaFile
 *                                                        src/sketch.cpp:455-566 (and :173-238)
 *   rk_sketch_batch_ex     <- per-file loop of sketchFastqFile   src/sketch.cpp:741-866
 *   rk_index_build         <- transSketches              src/sketch.cpp:970-1017 (32-bit), :904-969 (64-bit)
 *   rk_index_import/export <- .dict/.index load/store    src/dist.cpp:86-129, src/sketch.cpp:991-1011
 *   rk_dist_rows           <- row loop of index_tridist  src/dist.cpp:174-258
 *                             row loop of index_dist     src/dist.cpp:560-692 (without -N)
 *   rk_topn_rows           <- -N max-heap of index_dist  src/dist.cpp:599,625-640,683-689
 *
 * Conventions
 *   - plain C types only; every call returns 0 on success or a negative rk_status and
 *     never calls exit(); rk_last_error(ctx) returns a message for the last failure on
 *     that context.
 *   - pointers are HOST pointers unless the parameter name ends in _dev.
 *   - objects (rk_filter, rk_sketches, rk_index) are library-owned, device-resident and
 *     freed with their *_free function; buffers the caller receives from the library are
 *     released with rk_free_host.  Nothing crosses allocators.
 *   - calls without a `stream` argument are synchronous at return.  *_dev calls are
 *     asynchronous on the given HIP stream (void* == hipStream_t, NULL = default stream).
 *   - one context per (process, GPU); one host thread per context.
 *   - the engine has NO CPU fallback: creating a context without a usable GPU fails
 *     with RK_ERR_NO_DEVICE.
 */
#ifndef RABBITKSSD_H
#define RABBITKSSD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rk_status {
    RK_OK = 0,
    RK_ERR_ARG = -1,        /* invalid argument (incl. the reference's parameter checks) */
    RK_ERR_NO_DEVICE = -2,  /* no HIP device / device ordinal out of range              */
    RK_ERR_HIP = -3,        /* a HIP runtime call failed                                */
    RK_ERR_NOMEM = -4,      /* host or device allocation failed                         */
    RK_ERR_CAPACITY = -5,   /* caller-provided output buffer too small                  */
    RK_ERR_UNSUPPORTED = -6 /* outside this build's limits (e.g. >= 2^32 postings)      */
} rk_status;

typedef struct rk_ctx rk_ctx;
typedef struct rk_filter rk_filter;
typedef struct rk_sketches rk_sketches;
typedef struct rk_index rk_index;

/* kssd_parameter_t (src/common.h:8-25), the fields the hot path reads. */
typedef struct rk_params {
    int32_t half_k, half_subk, drlevel;
    int32_t rev_add_move, half_outctx_len;
    int32_t dim_start, dim_end;
    uint32_t kmer_size;
    uint64_t domask, tupmask, undomask0, undomask1;
} rk_params;

/* One reported pair.  alldist (triangle=1): row=i, col=j>i, size0=|S_i|, size1=|S_j|
 * (src/dist.cpp:207-217); dist (triangle=0): row=query, col=ref, size0=|ref|,
 * size1=|query| (src/dist.cpp:600-609). */
typedef struct rk_hit {
    uint32_t row, col;
    int32_t common, size0, size1, pad_;
    double jorc; /* jaccard (metric 0) or containment (metric 1) */
    double dist; /* mashD   (metric 0) or AafD        (metric 1) */
} rk_hit;

/* ---- context ------------------------------------------------------------------- */
int rk_device_count(void);
int rk_ctx_create(int device, rk_ctx **out);
/* Objects (rk_filter, rk_sketches, rk_index) must be freed before the context they were created on. */
void rk_ctx_destroy(rk_ctx *ctx);
/* Device memory released by the library's objects and temporaries is cached in the context and handed out
 * again (steady-state calls allocate nothing: hipMalloc/hipFree cost 50-300 us each and hipFree synchronises the
 * device); rk_ctx_trim returns the cached blocks to the driver. */
void rk_ctx_trim(rk_ctx *ctx);
/* out[0] bytes obtained from the driver, out[1] of which idle in the cache, out[2] hipMalloc calls, out[3] hipFree
 * calls made by the context's allocator so far (a steady-state call sequence leaves out[2] and out[3] unchanged). */
void rk_ctx_pool_stats(rk_ctx *ctx, uint64_t out[4]);
/* Measurement: with timing on, a pass brackets its dominant kernel with HIP events on the stream it is launched on;
 * rk_ctx_last_ms(ctx, RK_MS_SKETCH_KERNEL) then returns that kernel's duration in milliseconds for the last
 * rk_sketch_* call.  (The distance entry points launch one kernel per call or band: bracket rk_dist_rows_dev yourself.) */
#define RK_MS_SKETCH_KERNEL 0
void rk_ctx_set_timing(rk_ctx *ctx, int on);
/* A process that makes ONE pass (a command-line tool) says so: the library then keeps work on the host where the device path
 * would first have to load a code object that costs more than it saves on a single call (today: ordering up to 2^18 hit
 * records by (row, col) in rk_dist_rows: 8 ms to load the sort against 3 ms of std::sort for 45,000 records), and a self join
 * never pays for structures that only pay off on later joins.  A context that is NOT single-shot treats an index as resident:
 * from the second unsharded sparse self join over one index on, the tile kernel runs (its records are built by that second
 * call, 1-2 ms once; 0.030 against 0.032 ms per join at 10,000 genomes, 0.084 against 0.115 at 50,000).  Results are the same
 * either way. */
void rk_ctx_set_single_shot(rk_ctx *ctx, int on);
double rk_ctx_last_ms(const rk_ctx *ctx, int which);
const char *rk_last_error(const rk_ctx *ctx);
const char *rk_version(void);
void rk_free_host(void *p);

/* ---- memory / stream helpers ----------------------------------------------------------
 * What a host compiled without the HIP headers (the reference's g++ build) needs to keep the
 * device fed the way src/sketch.cpp's producer/consumer threads keep the CPU cores fed
 * (src/sketch.cpp:318-460): page-locked staging buffers the parser threads fill, device
 * buffers, a stream, and an upload that returns immediately so parsing of the next batch
 * overlaps it.  A stream is an opaque handle; NULL is the default stream. */
int rk_pinned_alloc(rk_ctx *ctx, uint64_t bytes, void **out);
void rk_pinned_free(void *p);
int rk_dev_alloc(rk_ctx *ctx, uint64_t bytes, void **out);
void rk_dev_free(void *p);
int rk_stream_create(rk_ctx *ctx, void **out);
void rk_stream_destroy(void *stream);
int rk_stream_sync(rk_ctx *ctx, void *stream);
int rk_upload_async(rk_ctx *ctx, void *dst_dev, const void *src_host, uint64_t bytes, void *stream);
/* device-to-device copy on `stream` (a host that streams a file of unknown inflated size grows its device buffer) */
int rk_dev_copy_async(rk_ctx *ctx, void *dst_dev, const void *src_dev, uint64_t bytes, void *stream);

/* ---- parameters (host arithmetic only) -------------------------------------------- */
/* RK_ERR_ARG when half_subk - drlevel < 3 (src/common.cpp:37), half_k < half_subk or
 * half_subk >= 8 (src/shuffle.cpp:26,30), half_k > 16. */
int rk_params_init(int half_k, int half_subk, int drlevel, rk_params *out);
/* number of hash bits = 4*(half_k-drlevel); >32 means the 64-bit layout (use64) */
int rk_hash_bits(const rk_params *p);

/* ---- sketching ---------------------------------------------------------------- */
/* Uploads the .shuf table (int32[16^half_subk], src/shuffle.cpp:8-23) and builds the
 * on-chip pre-filter for entries < dim_end. */
int rk_filter_create(rk_ctx *ctx, const rk_params *p, const int32_t *shuffled_dim,
                     rk_filter **out);
void rk_filter_free(rk_filter *f);

/* Sketches n_genomes genomes.  seq holds the sequence bytes of all records back to
 * back (newlines already removed, kseq semantics); rec_off[n_rec+1] delimits records
 * (windows never span records, src/sketch.cpp:487-488); genome_rec[n_genomes+1] gives
 * each genome's record range.  Result: per-genome SORTED UNIQUE 32-bit hashes. */
int rk_sketch_batch(rk_ctx *ctx, const rk_filter *f, const uint8_t *seq, const uint64_t *rec_off,
                    uint64_t n_rec, const uint64_t *genome_rec, uint32_t n_genomes,
                    rk_sketches **out);

/* Same, inputs already packed and resident in HBM: packed_dev holds each genome at
 * gbeg[g] (a multiple of 1024) .. gend[g]; records of a genome are separated by one
 * 0x00 byte and the gap up to the next multiple of 1024 is zero-filled (see
 * rk_pack_layout / rk_pack_genomes).  gbeg/gend are host arrays.  All device work is
 * enqueued on `stream`; the call synchronises that stream before returning (it needs the
 * candidate and hash counts on the host). */
int rk_sketch_packed_dev(rk_ctx *ctx, const rk_filter *f, const uint8_t *packed_dev,
                         uint64_t packed_bytes, const uint64_t *gbeg, const uint64_t *gend,
                         uint32_t n_genomes, void *stream, rk_sketches **out);
/* FASTQ variant of both (sketchFastqFile, src/sketch.cpp:596-890): qual (optional) holds one quality
 * character per base of seq, a base with quality < least_qual is invalid (src/sketch.cpp:785);
 * a hash is kept only if it occurred at least min_count times (leastNumKmer, :828-845).
 * rk_sketch_batch == rk_sketch_batch_ex(qual = NULL, least_qual = 0, min_count = 1). */
int rk_sketch_batch_ex(rk_ctx *ctx, const rk_filter *f, const uint8_t *seq, const uint8_t *qual,
                       int least_qual, uint32_t min_count, const uint64_t *rec_off, uint64_t n_rec,
                       const uint64_t *genome_rec, uint32_t n_genomes, rk_sketches **out);
int rk_sketch_packed_dev_ex(rk_ctx *ctx, const rk_filter *f, const uint8_t *packed_dev,
                            uint64_t packed_bytes, const uint64_t *gbeg, const uint64_t *gend,
                            uint32_t n_genomes, uint32_t min_count, void *stream, rk_sketches **out);
/* host helpers for the packed layout: sizes first, then fill a caller buffer */
int rk_pack_layout(const uint64_t *rec_off, uint64_t n_rec, const uint64_t *genome_rec,
                   uint32_t n_genomes, uint64_t *gbeg, uint64_t *gend, uint64_t *packed_bytes);
int rk_pack_genomes(const uint8_t *seq, const uint64_t *rec_off, uint64_t n_rec,
                    const uint64_t *genome_rec, uint32_t n_genomes, const uint64_t *gbeg,
                    uint8_t *packed, uint64_t packed_bytes);

/* device-resident CSR of sketches.  A genome's hashes may come in any order (the reference writes `unordered_set`
 * iteration order, src/sketch.cpp:537-553): the library sorts them on the device, rk_sketches_download returns them
 * ascending.  A sketch that repeats a hash keeps its repeats (they count, src/dist.cpp:199-202). */
int rk_sketches_from_host(rk_ctx *ctx, const uint32_t *hashes, const uint64_t *off,
                          uint32_t n_genomes, rk_sketches **out);
/* same from device-resident arrays (copied device-to-device into a library-owned object).  The copies run on the
 * context's own stream: the arrays must be COMPLETE when the call is made (synchronise the stream that produced them
 * first).  RK_ERR_ARG when off_dev is not a CSR offset table (off[0] = 0, non-decreasing). */
int rk_sketches_from_dev(rk_ctx *ctx, const uint32_t *hashes_dev, const uint64_t *off_dev,
                         uint32_t n_genomes, rk_sketches **out);
/* 64-bit hash layout (use64: half_k - drlevel > 8, src/sketch.cpp:336): the sketch kernel produces it
 * by itself when rk_hash_bits() > 32; these move such sketches across the boundary. */
int rk_sketches_from_host64(rk_ctx *ctx, const uint64_t *hashes, const uint64_t *off,
                            uint32_t n_genomes, rk_sketches **out);
int rk_sketches_download64(const rk_sketches *s, uint64_t *hashes, uint64_t *off);
int rk_sketches_is64(const rk_sketches *s);
uint32_t rk_sketches_count(const rk_sketches *s);
uint64_t rk_sketches_total(const rk_sketches *s);
/* number of k-mer windows seen by the last sketch call that produced s (0 if imported) */
uint64_t rk_sketches_windows(const rk_sketches *s);
/* copies off[n+1] and hashes[total] to caller buffers (either may be NULL) */
int rk_sketches_download(const rk_sketches *s, uint32_t *hashes, uint64_t *off);
const uint32_t *rk_sketches_hashes_dev(const rk_sketches *s);
const uint64_t *rk_sketches_off_dev(const rk_sketches *s);
void rk_sketches_free(rk_sketches *s);

/* ---- inverted index ------------------------------------------------------------- */
/* Builds the reference index from device-resident sketches: postings ordered
 * (hash asc, genome asc) exactly like the .dict file, a compact CSR over the distinct
 * hashes, and the per-hash "later genomes" ranges used by the all-vs-all triangle. */
int rk_index_build(rk_ctx *ctx, const rk_sketches *s, int hash_bits, rk_index **out);
/* From the on-disk pair: postings = .dict payload (u32[total]), counts = the dense
 * u32[2^hash_bits] array of the .index file, ref_sizes = sketch sizes of the
 * reference genomes (from the .sketch).  Replaces the load + prefix sum of
 * src/dist.cpp:86-129. */
int rk_index_import(rk_ctx *ctx, const uint32_t *postings, uint64_t total, const uint32_t *counts,
                    int hash_bits, const uint32_t *ref_sizes, uint32_t n_ref, rk_index **out);
/* To the on-disk pair: postings[total] and (optional) dense counts[2^hash_bits]. */
int rk_index_export(const rk_index *idx, uint32_t *postings, uint32_t *counts);
/* The same content without the dense array: the distinct hashes (ascending, rk_index_distinct() of them) and the length of
 * each one's posting list -- 8 bytes per distinct hash instead of 4 * 2^hash_bits; a host that writes the dense .index
 * file (src/sketch.cpp:1008-1011) scatters the counts into the zero-filled file itself.  32-bit hash layout only. */
int rk_index_export_lists(const rk_index *idx, uint32_t *postings, uint32_t *hashes, uint32_t *counts);
/* 64-bit hash layout: the sparse .index variant {u64 n; u64 hash[n]; u32 count[n]} with the .dict
 * posting blocks in the same order (src/sketch.cpp:942-963, read at src/dist.cpp:36-82).  Export
 * lists the hashes ascending; import accepts any block order (the reference writes hash-map order). */
int rk_index_export64(const rk_index *idx, uint32_t *postings, uint64_t *hashes, uint32_t *counts);
int rk_index_import64(rk_ctx *ctx, const uint32_t *postings, uint64_t total, const uint64_t *hashes,
                      const uint32_t *counts, uint64_t n_hash, int hash_bits, const uint32_t *ref_sizes,
                      uint32_t n_ref, rk_index **out);
/* Internal genome order.  rk_index_build renumbers the genomes so that relatives (genomes that share several of their
 * smallest hashes) become neighbours, whatever order the collection was listed in -- the reference's OpenMP loop leaves
 * them in completion order (src/sketch.cpp:558-568); the device layout (compact posting slices, row pairs) relies on
 * neighbours being relatives.  Nothing of this is visible in results: hit records, dense counter rows and exported
 * postings carry the caller's genome indices.  It only decides WHICH pairs a row shard (rk_dist_opts.row_first/row_step/
 * row_block) computes: shards partition the rows of the internal order.  orig_out[i] = caller's index of internal genome i. */
int rk_index_order(const rk_index *idx, uint32_t *orig_out);
uint64_t rk_index_total(const rk_index *idx);    /* H = number of postings        */
uint64_t rk_index_distinct(const rk_index *idx); /* U = number of distinct hashes */
uint32_t rk_index_genomes(const rk_index *idx);
int rk_index_hash_bits(const rk_index *idx);
/* 1 when rk_index_build took its bucket-sort path (set sketches with 32-bit hashes whose buckets fit the LDS sort),
 * 0 for the general path (device-wide radix sort) or an imported index: lets a harness say which build it timed. */
int rk_index_built_fast(const rk_index *idx);
/* Which structures of the all-vs-all join the index carries right now, as a bit mask: 1 = slice records (one per (genome,
 * hash) element: rk_near_kernel, rk_dist_kernel), 2 = tile records (one per posting list and pair of 32-genome blocks it
 * touches: rk_tile_kernel), 4 = the tile records came with rk_index_build itself (collections of RK_DIST_TILES_MIN_GENOMES =
 * 4,000 genomes and more: the FIRST self join already runs on the tile kernel; slice records are then made on first use --
 * a dense report, RK_DIST_TILES=0).  Which kernel a self join takes depends on this, the index's size and the options --
 * never on how often the index was joined before.  (The reference has one loop for every collection, src/dist.cpp:174-258.) */
int rk_index_products(const rk_index *idx);
/* sum over all reference hashes h of c_h^2 = postings streamed by a full alldist
 * (the T of the roofline formula) */
uint64_t rk_index_sum_sq(const rk_index *idx);
/* The all-vs-all join reads one 8-byte slice record per (genome, hash) pair with later sharers.  out[0] = records,
 * out[1] = of which in compact form (first genome + bitmask: the record IS the posting list, no posting is gathered),
 * out[2] = records the row-pair kernel walks (the rest are covered by the pair partner), out[3] = the 8-byte tile records of
 * the tile kernel (one per posting list and pair of 32-genome blocks it touches; 0 while the index has none: rk_index_products).
 * Computed on first request (one small kernel) and cached; 0s for an imported index. */
int rk_index_self_stats(const rk_index *idx, uint64_t out[4]);
/* Multi-GPU, sharded build (round 5).  The all-vs-all join shards twice: the BUILD by hash range (a posting list lives in one
 * range: its tile records are found without looking at the other ranges), the JOIN by rows (a tile of the pair matrix belongs
 * to the shard of its row block; blocks of 32 genomes of the internal order dealt round-robin, as rk_dist_opts.row_block = 32
 * does).  Between the two lies ONE all-to-all of 12-byte tile records -- no index is replicated, nothing is reduced:
 *   every shard:  rk_index_build_shard(sketches, hash_bits, shard, n_shards)   the lists of hash range `shard` (n_shards a power of
 *                 two <= 64; every shard holds the same CSR sketches -- one broadcast -- and computes the same internal order)
 *                 rk_index_shard_records(part, counts[n_shards])    tile records it holds for every destination shard
 *                 rk_index_shard_pack(part, send_dev, stream)       ... contiguous by destination (12 bytes each)
 *   (the caller's all-to-all: RCCL / torch.distributed.all_to_all_single, or peer copies)
 *                 rk_index_join_shard(ctx, part, recv_dev, n, &join)  sorts what arrived by tile: a join-only index over THIS
 *                 shard's rows; rk_dist_rows(_dev)(join, NULL, opts with row_step = 1) reports the pairs whose row block it
 *                 owns.  The union over the shards is the whole result (the reference's rows are independent, src/dist.cpp:174).
 * A collection too big for one pass of the bucket sort (more than ~5 * 10^7 postings) is built the same way on ONE device:
 * rk_index_build covers the hash space range by range, the postings of a range behind those of the range before. */
int rk_index_build_shard(rk_ctx *ctx, const rk_sketches *s, int hash_bits, uint32_t shard, uint32_t n_shards, rk_index **out);
int rk_index_shard_records(const rk_index *part, uint64_t *counts_out);
int rk_index_shard_pack(const rk_index *part, void *send_dev, void *stream);
int rk_index_join_shard(rk_ctx *ctx, const rk_index *part, const void *recv_dev, uint64_t n_records, rk_index **out);
/* One process, one context per GPU (`rabbit_kssd alldist --gpus N`): the exchange without a communicator -- GPU d pulls chunk d
 * of every shard's packed records over its own links (hipMemcpyPeerAsync, bounded wait).  parts[r] = shard r of n, each on its own
 * context; recv_dev[d] (free with rk_dev_free) and n_recv[d] are what rk_index_join_shard takes on GPU d.
 * Called from one host thread; the shard builds before it and the joins behind it may run one thread per GPU. */
int rk_index_shard_exchange(rk_index *const *parts, uint32_t n, void **recv_dev, uint64_t *n_recv);
/* Multi-GPU: the whole index as ONE contiguous device blob, so that the owner can hand it
 * to an RCCL broadcast (one collective, no reduction: query rows are independent) and every
 * peer rebuilds an identical rk_index from the received bytes.  pack/unpack only enqueue
 * device-to-device copies on `stream` and synchronise it before returning. */
uint64_t rk_index_blob_bytes(const rk_index *idx);
int rk_index_pack_dev(const rk_index *idx, void *blob_dev, uint64_t blob_cap, void *stream);
int rk_index_unpack_dev(rk_ctx *ctx, const void *blob_dev, uint64_t blob_bytes, void *stream,
                        rk_index **out);
/* The same inside ONE process that drives several GPUs (one context and one host thread per GPU): replicates the
 * index of src's context onto each of the n_dst contexts, all peers copying at the same time over their own xGMI link
 * (hipMemcpyPeerAsync; contexts on the source's device get a device-to-device copy).  out[i] belongs to dst[i]. */
int rk_index_broadcast(const rk_index *src, rk_ctx *const *dst, uint32_t n_dst, rk_index **out);
void rk_index_free(rk_index *idx);

/* ---- distances ------------------------------------------------------------------- */
typedef struct rk_dist_opts {
    int32_t triangle;   /* 1: alldist (queries ARE the indexed sketches, cols j>i, keep
                              dist <  max_dist, src/dist.cpp:207,232)
                           0: dist (all cols, keep dist <= max_dist, src/dist.cpp:600,624) */
    int32_t metric;     /* 0 jaccard->mashD, 1 containment->AafD (-M)                  */
    int32_t kmer_size;  /* 2*half_k                                                    */
    int32_t row_block;  /* rows are dealt to the shards in blocks of row_block consecutive rows
                           (block-cyclic); 0,1 = single rows.  An even row_block lets the
                           all-vs-all kernel walk neighbouring rows in pairs, a multiple of 32
                           lets the tile kernel count every tile on one shard only (32 is a
                           good value for multi-GPU runs)                                   */
    double max_dist;    /* -D                                                          */
    uint32_t row_first; /* this shard owns blocks row_first, row_first+row_step, ...:  */
    uint32_t row_step;  /*   row sharding across GPUs; 0,1 = all rows.  queries == NULL (self join): rows of the
                             index's internal genome order (rk_index_order); a reported pair belongs to the shard of
                             its member that comes first in that order                                            */
} rk_dist_opts;

/* The tile kernel is output-sensitive: of the 32 x 32 tiles of the pair matrix that hold a record at all, a launch gives a
 * workgroup only to those that can hold a reportable pair under `opts` (records per smallest sketch of the tile against the
 * threshold; the test is exact, the counted tiles are counted exactly).  out[0] = tiles with records, out[1] = tiles a launch
 * with these options starts (opts == NULL: 0), out[2] = tile records, out[3] = record slots (every tile from an even slot).
 * All 0 while the index has no tile records.  (The reference forms every cell of every row, src/dist.cpp:194-255.) */
int rk_index_tile_stats(const rk_index *idx, const rk_dist_opts *opts, uint64_t out[4]);

/* Counts |S_q n S_r| through the inverted index and applies the reference's epilogue.
 * queries == NULL is only valid with triangle=1 (the indexed sketches are the queries).
 * hits_out is library-allocated (rk_free_host), sorted by (row, col).  The jaccard/containment and
 * distance of every returned pair are recomputed on the host with the C library's log -- the
 * expression of src/dist.cpp:218-231 / :239-252 -- and the threshold is applied to that value (the
 * device reports with a threshold a few ulps wider): values and hit set are the reference's bit for bit.
 * common_dense (optional, host, n_query*n_ref int32, row-major) receives the full
 * counter rows of the selected rows (other rows untouched) -- used by parity tests. */
int rk_dist_rows(rk_ctx *ctx, const rk_index *idx, const rk_sketches *queries,
                 const rk_dist_opts *opts, rk_hit **hits_out, uint64_t *n_hits,
                 int32_t *common_dense);

/* Asynchronous all-in-HBM variant: hits are appended (unordered) to hits_dev
 * (capacity hits_cap records); *n_hits_dev (uint64, zeroed by the caller) counts every
 * hit, including those beyond the capacity, so an overflow is detectable.  Distances here are the
 * device's own FP64 evaluation (its log may differ from the C library's in the last bit: <= 1e-12).
 * Concurrency: calls with explicit queries and self joins that run on the tile kernel (collections with wide clusters or
 * tiny sketches, RK_DIST_TILES=1) keep no per-launch state in the index and may overlap freely on different streams.  A
 * self join that runs on the near-window kernel keeps its fallback list IN the index: at most one such self join per index
 * may be in flight at a time (serialise them on one stream, or use one index object per stream).  (With RK_DISTQ_SLICED=1 --
 * a measured, slower variant of the query path, off by default -- the scratch of the membership pass is kept with the QUERY
 * sketches: one call per query-sketches object at a time.)
 * Limits: fewer than 2^31-1 genomes and 2^32-1 postings per index.  From 2^31-1 postings on (all of GenBank's bacteria at
 * ~1,200 hashes each) rk_index_build leaves out the slice records of the row kernels (their posting offsets would collide
 * with the tag bit of the compact form): export, explicit queries below 2^31 postings and SPARSE self joins (the tile
 * kernel reads the posting lists themselves) work on such an index; a dense self join (a threshold that admits distance
 * 1.0) and sketches that repeat a hash return RK_ERR_UNSUPPORTED. */
int rk_dist_rows_dev(rk_ctx *ctx, const rk_index *idx, const rk_sketches *queries,
                     const rk_dist_opts *opts, rk_hit *hits_dev, uint64_t hits_cap,
                     uint64_t *n_hits_dev, void *stream);

/* Name (as a profiler prints it, e.g. "rk_dist_kernel<true, 2, 512>") of the kernel rk_dist_rows(_dev) launches for
 * these arguments: lets a harness check that a stored counter profile belongs to the variant it is timing.  A big self
 * join runs as several launches (bands of rows, each with LDS rows as wide as the columns behind its first row): the
 * name is then that of the first band followed by " [N bands]". */
int rk_dist_kernel_name(rk_ctx *ctx, const rk_index *idx, const rk_sketches *queries, const rk_dist_opts *opts,
                        char *buf, size_t cap);

/* -N: keeps, per query row, the max_neighbor nearest hits with the reference's heap
 * order (emitted largest distance first).  hits must be sorted by (row, col); the
 * result is written in place and its length returned through n_hits. */
int rk_topn_rows(rk_hit *hits, uint64_t *n_hits, uint64_t max_neighbor);

/* one output line, "%s\t%s\t%d|%d|%d\t%f\t%f\n" (src/dist.cpp:233 / :642) */
int rk_format_hit(char *buf, size_t cap, const char *name_a, const char *name_b,
                  const rk_hit *hit);

#ifdef __cplusplus
}
#endif
#endif /* RABBITKSSD_H */
