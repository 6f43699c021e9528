// rk_core.hip -- context, parameters and host-side helpers of the C ABI.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "rk_internal.h"

int rk_fail(rk_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

// ---- caching device allocator ----------------------------------------------------------------------
static size_t pool_round(size_t bytes)
{
    if (bytes <= (1u << 16)) return (bytes + 511) & ~(size_t)511;
    if (bytes <= (1u << 24)) return (bytes + 65535) & ~(size_t)65535;
    return (bytes + ((1u << 21) - 1)) & ~(size_t)((1u << 21) - 1);
}

// (ctx->mu held) a block back into the cache, or to the driver when the cache is full
static void pool_put_locked(rk_ctx *ctx, void *p)
{
    auto it = ctx->live.find(p);
    if (it == ctx->live.end()) {  // not from this pool
        (void)hipFree(p);
        return;
    }
    if (ctx->cached_bytes + it->second > ctx->cache_limit) {  // the cache is full: back to the driver
        ctx->pool_bytes -= it->second;
        ctx->live.erase(it);
        ctx->driver_frees++;
        (void)hipFree(p);
        return;
    }
    ctx->cached_bytes += it->second;
    ctx->free_blocks.emplace(it->second, p);
}

// (ctx->mu held) deferred blocks whose event has passed return to the cache; wait: all of them (the stream is synchronised first)
static void pool_collect_locked(rk_ctx *ctx, bool wait)
{
    size_t kept = 0;
    for (auto &d : ctx->deferred) {
        const hipError_t e = wait ? hipEventSynchronize(d.second) : hipEventQuery(d.second);
        if (e == hipSuccess || (wait && e != hipErrorNotReady)) {
            pool_put_locked(ctx, d.first);
            ctx->spare_events.push_back(d.second);
        } else {
            (void)hipGetLastError();   // (hipErrorNotReady is not an error here)
            ctx->deferred[kept++] = d;
        }
    }
    ctx->deferred.resize(kept);
}

void rk_pool_free_after(rk_ctx *ctx, void *p, hipStream_t st)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(ctx->mu);
    hipEvent_t ev = nullptr;
    if (!ctx->spare_events.empty()) {
        ev = ctx->spare_events.back();
        ctx->spare_events.pop_back();
    } else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
        ev = nullptr;
    }
    if (!ev || hipEventRecord(ev, st) != hipSuccess) {   // no event to be had: wait for the stream instead
        (void)hipGetLastError();
        if (ev) ctx->spare_events.push_back(ev);
        (void)hipStreamSynchronize(st);
        pool_put_locked(ctx, p);
        return;
    }
    ctx->deferred.emplace_back(p, ev);
}

void *rk_pool_alloc(rk_ctx *ctx, size_t bytes)
{
    const size_t want = pool_round(bytes ? bytes : 1);
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (!ctx->deferred.empty()) pool_collect_locked(ctx, false);
        auto it = ctx->free_blocks.lower_bound(want);
        // best fit, but never hand a block far bigger than the request (it would be missing when its own size is asked for)
        if (it != ctx->free_blocks.end() && (it->first <= 2 * want || it->first <= (1u << 16))) {
            void *p = it->second;
            ctx->cached_bytes -= it->first;
            ctx->free_blocks.erase(it);
            return p;
        }
    }
    void *p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) {
        (void)hipGetLastError();
        rk_ctx_trim(ctx);  // give cached blocks back and try once more
        if (hipMalloc(&p, want) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->live[p] = want;
    ctx->pool_bytes += want;
    ctx->driver_allocs++;
    return p;
}

void rk_pool_free(rk_ctx *ctx, void *p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(ctx->mu);
    pool_put_locked(ctx, p);
}

void *rk_pinned_scratch(rk_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->pinned_bytes) return ctx->pinned;
    size_t want = std::max<size_t>(kPinnedBytes, ctx->pinned_bytes);
    while (want < bytes) want *= 2;
    void *p = nullptr;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    ctx->pinned = p;
    ctx->pinned_bytes = want;
    return p;
}

int rk_read_back(rk_ctx *ctx, void *dst, const void *src_dev, size_t bytes, hipStream_t stream)
{
    if (bytes <= ctx->pinned_bytes && ctx->pinned) {
        RK_HIP(ctx, hipMemcpyAsync(ctx->pinned, src_dev, bytes, hipMemcpyDeviceToHost, stream));
        RK_HIP(ctx, hipStreamSynchronize(stream));
        memcpy(dst, ctx->pinned, bytes);
    } else {
        RK_HIP(ctx, hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, stream));
        RK_HIP(ctx, hipStreamSynchronize(stream));
    }
    return RK_OK;
}

int rk_occupancy(rk_ctx *ctx, const void *kernel, int threads, size_t lds_bytes)
{
    const auto key = std::make_tuple(kernel, threads, lds_bytes);
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        auto it = ctx->occupancy.find(key);
        if (it != ctx->occupancy.end()) return it->second;
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds_bytes) != hipSuccess) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->occupancy[key] = per_cu;
    return per_cu;
}

// ---- counter calibration (developer): a streaming read of a known byte count at 4, 8 or 16 bytes per lane, so that
// FETCH_SIZE can be calibrated on the access widths the kernels use (MI355X_MICROARCH.md: only the 16 B/lane
// factor is documented).  tools/prof_driver.py calib
template <class V> __global__ void k_calib_read(const V *p, size_t n, unsigned long long *sink)
{
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const V v = p[i];
        const unsigned char *b = reinterpret_cast<const unsigned char *>(&v);
        acc += b[0] + b[sizeof(V) - 1];
    }
    if (acc == 0x1234567ULL) *sink = acc;  // (keeps the loads alive)
}
extern "C" int rk_debug_calib_read(rk_ctx *ctx, uint64_t bytes, int width)
{
    if (!ctx || (width != 4 && width != 8 && width != 16)) return RK_ERR_ARG;
    RK_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf<char> buf(ctx);
    DevBuf<unsigned long long> sink(ctx);
    RK_HIP(ctx, buf.alloc(bytes));
    RK_HIP(ctx, sink.alloc(1));
    RK_HIP(ctx, hipMemsetAsync(buf.p, 1, bytes, ctx->stream));
    const unsigned grid = (unsigned)ctx->num_cu * 8;
    if (width == 4) hipLaunchKernelGGL(k_calib_read<uint32_t>, dim3(grid), dim3(256), 0, ctx->stream, (const uint32_t *)buf.p, bytes / 4, sink.p);
    else if (width == 8) hipLaunchKernelGGL(k_calib_read<uint2>, dim3(grid), dim3(256), 0, ctx->stream, (const uint2 *)buf.p, bytes / 8, sink.p);
    else hipLaunchKernelGGL(k_calib_read<uint4>, dim3(grid), dim3(256), 0, ctx->stream, (const uint4 *)buf.p, bytes / 16, sink.p);
    RK_HIP(ctx, hipGetLastError());
    RK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RK_OK;
}

// the device's streaming-read rate over `bytes` of HBM (bigger than the 256 MB Infinity Cache), HIP events around `reps`
// launches: the measured denominator SURVEY 8d asks for beside the nominal 8 TB/s (bench.py: roofline.peak_measured)
extern "C" int rk_debug_stream_read_gbs(rk_ctx *ctx, uint64_t bytes, int reps, double *gbs)
{
    if (!ctx || !gbs || reps < 1 || bytes < (1u << 20)) return RK_ERR_ARG;
    *gbs = 0.0;
    RK_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf<char> buf(ctx);
    DevBuf<unsigned long long> sink(ctx);
    RK_HIP(ctx, buf.alloc(bytes));
    RK_HIP(ctx, sink.alloc(1));
    RK_HIP(ctx, hipMemsetAsync(buf.p, 1, bytes, ctx->stream));
    const unsigned grid = (unsigned)ctx->num_cu * 8;
    hipEvent_t e0, e1;
    RK_HIP(ctx, hipEventCreate(&e0));
    RK_HIP(ctx, hipEventCreate(&e1));
    hipLaunchKernelGGL(k_calib_read<uint4>, dim3(grid), dim3(256), 0, ctx->stream, (const uint4 *)buf.p, bytes / 16, sink.p);   // warm-up
    (void)hipEventRecord(e0, ctx->stream);
    for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL(k_calib_read<uint4>, dim3(grid), dim3(256), 0, ctx->stream, (const uint4 *)buf.p, bytes / 16, sink.p);
    (void)hipEventRecord(e1, ctx->stream);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    RK_HIP(ctx, e);
    if (ms > 0.f) *gbs = (double)bytes * reps / (ms * 1e-3) / 1e9;
    return RK_OK;
}

extern "C" {

const char *rk_version(void) { return "rabbitkssd-amd 0.1.0 (gfx950)"; }

int rk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static uint32_t env_u32(const char *name, uint32_t dflt)
{
    const char *v = getenv(name);
    return v && atoi(v) > 0 ? (uint32_t)atoi(v) : dflt;
}

int rk_ctx_create(int device, rk_ctx **out)
{
    if (!out) return RK_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RK_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return RK_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return RK_ERR_HIP;
    rk_ctx *ctx = new (std::nothrow) rk_ctx;
    if (!ctx) return RK_ERR_NOMEM;
    ctx->device = device;
    // single attributes instead of hipGetDeviceProperties (which fills ~100 fields, some of them slow to query)
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || v <= 0) {
        delete ctx;
        return RK_ERR_HIP;
    }
    ctx->num_cu = v;
    ctx->max_lds = 64 * 1024;  // default limit; larger via opt-in attribute
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && v > 0)
        ctx->max_lds = (size_t)v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeSharedMemPerBlockOptin, device) == hipSuccess &&
        (size_t)v > ctx->max_lds)
        ctx->max_lds = (size_t)v;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess || !rk_pinned_scratch(ctx, kPinnedBytes)) {
        rk_ctx_destroy(ctx);
        return RK_ERR_HIP;
    }
    ctx->cache_limit = (size_t)env_u32("RK_POOL_LIMIT_MB", 32768) << 20;
    ctx->sw_dist_threads = env_u32("RK_DIST_THREADS", 0);
    ctx->sw_dist_rows = env_u32("RK_DIST_ROWS", 0);
    ctx->sw_dist_pair = env_u32("RK_DIST_PAIR", 1);
    ctx->sw_dist_pair_minwg = env_u32("RK_DIST_PAIR_MINWG", 3);
    ctx->sw_dist_persist = env_u32("RK_DIST_PERSIST", 1);
    ctx->sw_dist_cand_cap = env_u32("RK_DIST_CAND_CAP", 0);
    ctx->sw_dist_stage_hits = env_u32("RK_DIST_STAGE_HITS", 0);
    ctx->sw_dist_xcd_rows = env_u32("RK_DIST_XCD_ROWS", 0);
    ctx->sw_dist_bands = getenv("RK_DIST_BANDS") ? atoi(getenv("RK_DIST_BANDS")) != 0 : 1;
    if (getenv("RK_DIST_BAND_MIN_ROWS")) ctx->sw_dist_band_min_rows = std::max(1, atoi(getenv("RK_DIST_BAND_MIN_ROWS")));
    ctx->sw_dist_near = getenv("RK_DIST_NEAR") ? atoi(getenv("RK_DIST_NEAR")) != 0 : 1;
    if (getenv("RK_DIST_TILES")) ctx->sw_dist_tiles = atoi(getenv("RK_DIST_TILES"));
    if (getenv("RK_DIST_TILES_MIN_GENOMES")) ctx->sw_dist_tiles_min_genomes = atoi(getenv("RK_DIST_TILES_MIN_GENOMES"));
    if (getenv("RK_DIST_TILES_MIN_SHARD_ROWS")) ctx->sw_dist_tiles_min_shard_rows = atoi(getenv("RK_DIST_TILES_MIN_SHARD_ROWS"));
    if (getenv("RK_DIST_NEAR_UW")) ctx->sw_dist_near_uw = atoi(getenv("RK_DIST_NEAR_UW"));
    if (getenv("RK_DIST_NEAR_MIN")) ctx->sw_dist_near_min = std::max(1, atoi(getenv("RK_DIST_NEAR_MIN")));
    if (getenv("RK_DIST_FB_SKIP")) ctx->sw_dist_fb_skip = atoi(getenv("RK_DIST_FB_SKIP")) != 0;
    ctx->sw_dist_debug = getenv("RK_DIST_DEBUG") ? atoi(getenv("RK_DIST_DEBUG")) : 0;
    if (getenv("RK_DIST_LDS_KB")) ctx->sw_dist_lds_kb = std::max(0, atoi(getenv("RK_DIST_LDS_KB")));
    ctx->sw_sketch_img = getenv("RK_SKETCH_IMG") ? std::min(2, std::max(0, atoi(getenv("RK_SKETCH_IMG")))) : 2;
    ctx->sw_index_fast = getenv("RK_INDEX_FAST") ? atoi(getenv("RK_INDEX_FAST")) != 0 : 1;
    ctx->sw_index_relabel = getenv("RK_INDEX_RELABEL") ? atoi(getenv("RK_INDEX_RELABEL")) != 0 : 1;
    ctx->sw_index_no_self = getenv("RK_INDEX_NO_SELF") ? atoi(getenv("RK_INDEX_NO_SELF")) != 0 : 0;
    if (getenv("RK_INDEX_TILES")) ctx->sw_index_tiles = atoi(getenv("RK_INDEX_TILES"));
    if (getenv("RK_INDEX_NO_HEAVY")) ctx->sw_index_heavy = atoi(getenv("RK_INDEX_NO_HEAVY")) == 0;
    if (getenv("RK_TILE_REC_CAP")) ctx->sw_tile_rec_cap = strtoull(getenv("RK_TILE_REC_CAP"), nullptr, 10);
    *out = ctx;
    return RK_OK;
}

void rk_ctx_set_single_shot(rk_ctx *ctx, int on)
{
    if (ctx) ctx->single_shot = on != 0;
}

void rk_ctx_set_timing(rk_ctx *ctx, int on)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (on && !ctx->ev[0] && (hipEventCreate(&ctx->ev[0]) != hipSuccess || hipEventCreate(&ctx->ev[1]) != hipSuccess)) return;
    ctx->timing = on != 0;
}

void rk_ctx_pool_stats(rk_ctx *ctx, uint64_t out[4])
{
    if (!ctx || !out) return;
    std::lock_guard<std::mutex> lk(ctx->mu);
    out[0] = ctx->pool_bytes;
    out[1] = ctx->cached_bytes;
    out[2] = ctx->driver_allocs;
    out[3] = ctx->driver_frees;
}

double rk_ctx_last_ms(const rk_ctx *ctx, int which) { return ctx && which >= 0 && which < 4 ? ctx->last_ms[which] : 0.0; }

void rk_ctx_trim(rk_ctx *ctx)
{
    if (!ctx) return;
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    if (!ctx->deferred.empty()) pool_collect_locked(ctx, true);
    for (auto &b : ctx->free_blocks) {
        ctx->live.erase(b.second);
        ctx->pool_bytes -= b.first;
        ctx->driver_frees++;
        (void)hipFree(b.second);
    }
    ctx->free_blocks.clear();
    ctx->cached_bytes = 0;
}

void rk_ctx_destroy(rk_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
        if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
        if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
        if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
        if (ctx->ev_inv) (void)hipEventDestroy(ctx->ev_inv);
    }
    rk_ctx_trim(ctx);
    for (hipEvent_t e : ctx->spare_events) (void)hipEventDestroy(e);
    // blocks still handed out belong to objects the caller has not freed: they are released with the context
    for (auto &b : ctx->live) (void)hipFree(b.first);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    for (hipEvent_t e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    delete ctx;
}

const char *rk_last_error(const rk_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }

void rk_free_host(void *p) { free(p); }

// ---- memory / stream helpers: what a host without the HIP headers needs to keep the device
// fed (pinned staging, device buffers, an upload that overlaps the parser threads) -----------
int rk_pinned_alloc(rk_ctx *ctx, uint64_t bytes, void **out)
{
    if (!ctx || !out) return RK_ERR_ARG;
    *out = nullptr;
    RK_HIP(ctx, hipSetDevice(ctx->device));
    if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        *out = nullptr;
        return rk_fail(ctx, RK_ERR_NOMEM, "cannot pin %llu bytes of host memory", (unsigned long long)bytes);
    }
    return RK_OK;
}
void rk_pinned_free(void *p) { if (p) (void)hipHostFree(p); }

int rk_dev_alloc(rk_ctx *ctx, uint64_t bytes, void **out)
{
    if (!ctx || !out) return RK_ERR_ARG;
    *out = nullptr;
    RK_HIP(ctx, hipSetDevice(ctx->device));
    if (hipMalloc(out, bytes ? bytes : 1) != hipSuccess) {
        *out = nullptr;
        return rk_fail(ctx, RK_ERR_NOMEM, "cannot allocate %llu bytes on the device", (unsigned long long)bytes);
    }
    return RK_OK;
}
void rk_dev_free(void *p) { if (p) (void)hipFree(p); }

int rk_stream_create(rk_ctx *ctx, void **out)
{
    if (!ctx || !out) return RK_ERR_ARG;
    *out = nullptr;
    RK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = nullptr;
    RK_HIP(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *out = (void *)st;
    return RK_OK;
}
void rk_stream_destroy(void *stream) { if (stream) (void)hipStreamDestroy((hipStream_t)stream); }

int rk_stream_sync(rk_ctx *ctx, void *stream)
{
    if (!ctx) return RK_ERR_ARG;
    RK_HIP(ctx, hipSetDevice(ctx->device));
    RK_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    return RK_OK;
}

int rk_upload_async(rk_ctx *ctx, void *dst_dev, const void *src_host, uint64_t bytes, void *stream)
{
    if (!ctx || (bytes && (!dst_dev || !src_host))) return RK_ERR_ARG;
    RK_HIP(ctx, hipSetDevice(ctx->device));
    if (bytes) RK_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return RK_OK;
}

int rk_dev_copy_async(rk_ctx *ctx, void *dst_dev, const void *src_dev, uint64_t bytes, void *stream)
{
    if (!ctx || (bytes && (!dst_dev || !src_dev))) return RK_ERR_ARG;
    RK_HIP(ctx, hipSetDevice(ctx->device));
    if (bytes) RK_HIP(ctx, hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return RK_OK;
}

// src/common.cpp:35-78 (initParameter); argument checks of src/shuffle.cpp:26,30 folded in.
int rk_params_init(int half_k, int half_subk, int drlevel, rk_params *p)
{
    if (!p) return RK_ERR_ARG;
    if (half_subk - drlevel < 3) return RK_ERR_ARG;  // src/common.cpp:37
    if (half_k < half_subk || half_subk >= 8 || half_subk < 1) return RK_ERR_ARG;
    if (half_k > 16 || drlevel < 0) return RK_ERR_ARG;
    memset(p, 0, sizeof(*p));
    const int out = half_k - half_subk;
    p->half_k = half_k;
    p->half_subk = half_subk;
    p->drlevel = drlevel;
    p->half_outctx_len = out;
    p->rev_add_move = 4 * half_k - 2;
    p->kmer_size = 2u * (uint32_t)half_k;
    p->dim_start = 0;
    p->dim_end = 1 << (4 * (half_subk - drlevel));
    const uint64_t tupmask = ~0ULL >> (64 - 4 * half_k);
    const uint64_t domask = (tupmask >> (4 * out)) << (2 * out);
    const uint64_t undomask = (tupmask ^ domask) & tupmask;
    const uint64_t undomask1 = undomask & (tupmask >> ((half_k + half_subk) * 2));
    p->tupmask = tupmask;
    p->domask = domask;
    p->undomask1 = undomask1;
    p->undomask0 = undomask ^ undomask1;
    return RK_OK;
}

int rk_hash_bits(const rk_params *p) { return p ? 4 * (p->half_k - p->drlevel) : 0; }

// ---- packed layout for the sketch kernel --------------------------------------------
// genome g occupies [gbeg[g], gend[g]) of the packed buffer, gbeg a multiple of 1024;
// its records are separated by one 0x00 byte; the gap to the next genome is zero-filled.
int rk_pack_layout(const uint64_t *rec_off, uint64_t n_rec, const uint64_t *genome_rec,
                   uint32_t n_genomes, uint64_t *gbeg, uint64_t *gend, uint64_t *packed_bytes)
{
    if (!rec_off || !genome_rec || !gbeg || !gend || !packed_bytes) return RK_ERR_ARG;
    uint64_t pos = 0;
    for (uint32_t g = 0; g < n_genomes; g++) {
        const uint64_t r0 = genome_rec[g], r1 = genome_rec[g + 1];
        if (r1 < r0 || r1 > n_rec) return RK_ERR_ARG;
        uint64_t len = rec_off[r1] - rec_off[r0];
        if (r1 > r0) len += (r1 - r0 - 1);  // separators
        gbeg[g] = pos;
        gend[g] = pos + len;
        pos = (pos + len + 1023) & ~1023ULL;
        if (len == 0) pos += 0;
    }
    *packed_bytes = pos ? pos : 1024;
    return RK_OK;
}

int rk_pack_genomes(const uint8_t *seq, const uint64_t *rec_off, uint64_t n_rec,
                    const uint64_t *genome_rec, uint32_t n_genomes, const uint64_t *gbeg,
                    uint8_t *packed, uint64_t packed_bytes)
{
    if (!seq || !rec_off || !genome_rec || !gbeg || !packed) return RK_ERR_ARG;
    memset(packed, 0, packed_bytes);
    for (uint32_t g = 0; g < n_genomes; g++) {
        uint64_t pos = gbeg[g];
        for (uint64_t r = genome_rec[g]; r < genome_rec[g + 1]; r++) {
            if (r >= n_rec) return RK_ERR_ARG;
            const uint64_t len = rec_off[r + 1] - rec_off[r];
            if (pos + len > packed_bytes) return RK_ERR_CAPACITY;
            memcpy(packed + pos, seq + rec_off[r], len);
            pos += len + 1;  // leaves one 0x00 separator
        }
    }
    return RK_OK;
}

// ---- -N nearest neighbours -----------------------------------------------------------
// std::priority_queue<DistInfo, vector<DistInfo>, cmpDistInfo> (src/dist.h:19-32) as used
// at src/dist.cpp:599,625-640,683-689.  The heap is kept with the same libstdc++
// primitives the reference gets from <queue>, so ties are broken identically.
struct HitLess {
    bool operator()(const rk_hit &a, const rk_hit &b) const { return a.dist < b.dist; }
};

int rk_topn_rows(rk_hit *hits, uint64_t *n_hits, uint64_t max_neighbor)
{
    if (!hits || !n_hits) return RK_ERR_ARG;
    const uint64_t n = *n_hits;
    std::vector<rk_hit> heap;
    uint64_t w = 0, i = 0;
    HitLess less;
    while (i < n) {
        const uint32_t row = hits[i].row;
        heap.clear();
        uint64_t j = i;
        for (; j < n && hits[j].row == row; j++) {
            if (heap.size() < max_neighbor) {
                heap.push_back(hits[j]);
                std::push_heap(heap.begin(), heap.end(), less);
            } else if (!heap.empty() && hits[j].dist < heap.front().dist) {
                heap.push_back(hits[j]);
                std::push_heap(heap.begin(), heap.end(), less);
                std::pop_heap(heap.begin(), heap.end(), less);
                heap.pop_back();
            }
        }
        while (!heap.empty()) {  // top() first: largest distance first
            hits[w++] = heap.front();
            std::pop_heap(heap.begin(), heap.end(), less);
            heap.pop_back();
        }
        i = j;
    }
    *n_hits = w;
    return RK_OK;
}

int rk_format_hit(char *buf, size_t cap, const char *name_a, const char *name_b, const rk_hit *h)
{
    if (!buf || !name_a || !name_b || !h) return RK_ERR_ARG;
    return snprintf(buf, cap, "%s\t%s\t%d|%d|%d\t%f\t%f\n", name_a, name_b, h->common, h->size0,
                    h->size1, h->jorc, h->dist);
}

}  // extern "C"
