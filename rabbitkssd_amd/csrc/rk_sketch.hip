// rk_sketch.hip -- shuffled-k-mer dimension-reduction sketcher.
// Replaces the per-genome loop of sketchFastaFile (src/sketch.cpp:455-566) and the
// identical arithmetic of consumer_fasta_task (src/sketch.cpp:173-238).
//
// Data layout (HBM): the "packed" sequence buffer holds every genome at a 1024-byte
// aligned offset; records inside a genome are separated by one 0x00 byte (an invalid
// base resets the window exactly like a record boundary does, src/sketch.cpp:487-488,
// 502-504) and the tail up to the next 1 KiB boundary is zero.
//
// Kernel: one wavefront walks a chunk of consecutive 1 KiB blocks.  Per block every lane
// loads 16 contiguous bases (one coalesced global_load_dwordx4 per lane, 1 KiB per wave),
// converts them into a 32-bit word of 2-bit codes (shift + mask + v_dot4_u32_u8 per dword)
// plus a 16-bit validity mask, and fetches the words of the two preceding lanes (DPP).  All
// 16 k-mer windows that END inside the lane are then cut out of that 96-bit string with
// funnel shifts: no serial rolling dependency, 16 independent windows per lane.  The inner
// 2*subk bases of the window are tested against bitmaps of the (symmetrised) selected .shuf
// entries held in LDS; the ~0.2 % survivors wait in a per-wave LDS queue and are resolved 64 at a time: both strands,
// the canonical k-mer (dim_id, src/sketch.cpp:508-509), confirmation against the .shuf table (L2 / Infinity Cache; the
// big LDS image variant holds an exact table), and the dr_tuple goes to the genome's candidate region.
// Per-genome dedup (the reference's unordered_set): one workgroup per genome sorts its region in LDS
// (k_dedup), a scan and a placement kernel build the CSR; one upload and one read-back per batch.
#include <cstring>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "rk_internal.h"

#define RK_TRY(call) do { int rc__ = (call); if (rc__) return rc__; } while (0)
template <class T> static int pool_array(rk_ctx *ctx, T **out, size_t n)
{
    *out = static_cast<T *>(rk_pool_alloc(ctx, (n ? n : 1) * sizeof(T)));
    return *out ? RK_OK : rk_fail(ctx, RK_ERR_NOMEM, "cannot allocate %llu bytes on the device", (unsigned long long)(n * sizeof(T)));
}

namespace {

constexpr int kSketchThreads = 1024;           // 16 waves share one LDS filter image
constexpr int kWavesPerBlock = kSketchThreads / 64;
// LDS filter image (one workgroup per CU):
//   The inner 2*half_subk bases of a k-mer sit symmetrically inside it, so the dim_id of
//   the canonical k-mer (src/sketch.cpp:508-509) is either d_f, the inner bases of the
//   FORWARD window, or revcomp(d_f).  The hot loop therefore only cuts d_f out of the
//   forward words and tests it against bitmaps of the SYMMETRISED selection
//   {d : shuf[d] selected or shuf[revcomp(d)] selected}; reverse strand, canonical compare
//   and dr_tuple are computed for the survivors only.
//   bitmap A: 2^19 bits on the low 19 bits of d_f, bitmap B: 2^18 bits on its high 18 bits;
//   a window survives if both bits are set.  With 4096 selected entries (8192 after
//   symmetrising) a non-member passes with probability 2^-6 * 2^-5 = 0.05 %, the same
//   order as the true-positive rate, so the divergent survivor loop handles ~1 window per
//   1024-window wave step.
//   EXACT mode (<= 4096 selected entries, the usual half_subk - drlevel == 3) adds an
//   8192-slot open-addressing key table (32 KiB) + u16 values (16 KiB): survivors are
//   confirmed without leaving the CU.  Otherwise they are confirmed against the .shuf
//   table in HBM/L2.
constexpr int kExactSlotsLog2 = 13;
constexpr uint32_t kExactSlots = 1u << kExactSlotsLog2;
constexpr uint32_t kEmptyKey = 0xFFFFFFFFu;
// Two LDS images:
//   IMG 0 "one workgroup per CU": bitmap A 2^19 bits + bitmap B 2^18 bits + the exact table = 144 KiB.
//   IMG 1 "two workgroups per CU" (default): bitmap A 2^18 + bitmap B 2^18 bits = 64 KiB, no exact table (survivors are
//         confirmed against the .shuf table in L2 / Infinity Cache): twice the waves per SIMD, one block per loop
//         iteration (60 VGPRs, no spills), ~4x the false positives of the pre-filter (0.2 % of the windows instead of
//         0.05 %).  Measured 0.360 ms against 0.392 ms for IMG 0 on 128 x 5 Mb; A 2^19 + B 2^16 measured 0.365 ms.
template <int IMG> struct Img {
    static constexpr int kBitsA = IMG ? 18 : 19, kBitsB = 18;
    static constexpr uint32_t kStageCap = 96;  // slots of a wave's window queue (drained from 64 on)
    static constexpr uint32_t kWordsA = (1u << kBitsA) / 32, kWordsB = (1u << kBitsB) / 32;
    static constexpr bool kHasExact = IMG == 0;
    static constexpr size_t kFilterLdsBytes = (kWordsA + kWordsB) * 4 + (kHasExact ? kExactSlots * 4 + kExactSlots * 2 : 0);
    static constexpr int kWavesPerEu = IMG ? 8 : 4;
};
// per-wave queue in LDS of the windows that passed both bitmaps: resolved 64 at a time, one global atomic and one
// store burst per round
template <int IMG> constexpr size_t stage_lds_bytes() { return kWavesPerBlock * (Img<IMG>::kStageCap * 8 + 8); }
static_assert(Img<0>::kFilterLdsBytes + stage_lds_bytes<0>() <= 160 * 1024, "LDS image must fit one CU");
static_assert(2 * (Img<1>::kFilterLdsBytes + stage_lds_bytes<1>()) <= 160 * 1024, "two small images must fit one CU");

// per-genome row of the pass table (one small upload per pass): where the genome lies in the packed buffer, which
// chunks (runs of `chunk_blocks` consecutive 1 KiB blocks, one wave each) are its own, and its candidate region
struct GenomeRow {
    uint64_t beg;          // byte offset in the packed buffer (multiple of 1024)
    uint64_t reg_off;      // first slot of the candidate region
    uint32_t nblk;         // 1 KiB blocks
    uint32_t first_chunk;  // global index of its first chunk; row[n_genomes].first_chunk = number of chunks
    uint32_t reg_cap;      // slots in the candidate region
    uint32_t is_big;       // region beyond the LDS sort capacity: device-wide sort
};
static_assert(sizeof(GenomeRow) == 32, "GenomeRow is uploaded as raw bytes");

struct SketchArgs {
    const uint8_t *packed;
    const GenomeRow *rows;
    uint32_t n_genomes, n_chunks, chunk_blocks;
    const uint32_t *filter_image;  // kFilterLdsBytes, copied to LDS by every workgroup
    const int32_t *table;          // int32[16^half_subk]
    uint64_t tupmask, undomask0, undomask1;
    int32_t kmer, out2, dim_bits, hi_shift, dim_start, dim_end, dr_shift, und1_shift;
    // emitted dr_tuples go to the candidate region of their genome: cand[reg_off[g] .. + reg_cap[g]); gcount[g]
    // counts every emitted key, also those beyond the capacity (overflow is detected, the host retries)
    unsigned long long *cand;
    uint32_t *gcount;
    unsigned long long *n_windows;
    // chunks are handed out in order within `n_groups` equal ranges, one ticket counter each (128 B apart, zeroed
    // per pass); workgroup b works in range b % n_groups
    uint32_t *tickets;
    uint32_t n_groups;
    const struct ChunkDesc *chunks;  // one per chunk, written by k_chunk_table
    uint32_t prio_turns;             // RK_SCAN2_PRIO: 2 (default) issue priority by progress within the workgroup, 1 the workgroups of a CU alternate, 0 none
    unsigned long long *trace;       // developer aid (RK_SCAN2_TRACE=file): eight words per wave, see tools/trace_scan.py
};
// where a chunk lies: the waves read this instead of searching the genome rows (seven dependent loads per chunk)
struct ChunkDesc {
    uint64_t beg;       // byte offset of its first block in the packed buffer
    uint32_t nb_first;  // its blocks; bit 31: the first chunk of its genome
    uint32_t gid;
};
static_assert(sizeof(ChunkDesc) == 16, "read as one uint4");
constexpr uint32_t kTicketGroups = 32, kTicketStride = 32;  // counters, dwords between them

// low 32 bits of the 96-bit string w2:w1:w0 shifted right by sh (0 <= sh < 96); with a compile-time sh this is one
// v_alignbit_b32
__device__ inline uint32_t ext96_lo(uint32_t w2, uint32_t w1, uint32_t w0, int sh)
{
    if (sh < 32) return __builtin_amdgcn_alignbit(w1, w0, sh);
    if (sh < 64) return __builtin_amdgcn_alignbit(w2, w1, sh - 32);
    return w2 >> (sh - 64);
}

// low 64 bits of the same for a shift only known at run time, branch-free (5 selects + 2 funnel shifts)
__device__ inline uint64_t ext96v(uint32_t w2, uint32_t w1, uint32_t w0, uint32_t sh)
{
    const uint32_t q = sh >> 5;
    const uint32_t a0 = q == 0 ? w0 : (q == 1 ? w1 : w2);
    const uint32_t a1 = q == 0 ? w1 : (q == 1 ? w2 : 0u);
    const uint32_t a2 = q == 0 ? w2 : 0u;
    const uint32_t lo = __builtin_amdgcn_alignbit(a1, a0, sh & 31u), hi = __builtin_amdgcn_alignbit(a2, a1, sh & 31u);
    return ((uint64_t)hi << 32) | lo;
}

// 16 ASCII bases -> G: 2-bit codes, base i of the lane at bits 2i (the orientation of the
// reverse strand register `rvs_tuple`, src/sketch.cpp:499, before complementing);
// V: bit i set when base i is one of ACGTacgt (BaseMap, src/common.h:27-37).
// The hot loop works in the code the ASCII table gives for free, (c >> 1) & 3: A=0 C=1 T=2 G=3
// ("scan code"; the reference's BaseMap has G=2 T=3).  The filter bitmaps are built for scan-coded
// indices on the host, and the few windows that survive them are converted before the
// reference arithmetic (to_base_code): the conversion costs 3 VALU per word on 0.05 % of the
// windows instead of 3 VALU per input dword on all of them.
__device__ inline void pack16(const uint4 w, uint32_t &G, uint32_t &V)
{
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
    uint32_t y[4];
    G = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t u = (ws[d] >> 1) & 0x03030303u;  // scan code in bits 1:0 of every byte
        // v_dot4_u32_u8 with weights 1,4,16,64 squeezes the four 2-bit codes into one byte
        const uint32_t four = __builtin_amdgcn_udot4(u, 0x40100401u, 0u, false);
        G = d ? (four << (8 * d)) | G : four;
        // valid <=> the (case-folded) byte equals the letter its code stands for: one v_perm_b32
        // looks up "actg"[code] for all four bytes
        y[d] = (ws[d] | 0x20202020u) ^ __builtin_amdgcn_perm(0x67746361u, 0x67746361u, u);
    }
    V = 0xFFFFu;
    if ((y[0] | y[1] | y[2] | y[3]) != 0) {  // some byte is not a base (N, separator, padding): rare
        V = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t nz = ((y[d] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y[d];  // bit 7 <=> byte != 0, exact
            const uint32_t f = ((~nz) >> 7) & 0x01010101u;                    // 1 per valid byte
            V |= ((f | (f >> 7) | (f >> 14) | (f >> 21)) & 0xFu) << (4 * d);
        }
    }
}

// scan code <-> BaseMap code of every 2-bit group (swaps 2 and 3; an involution)
__host__ __device__ inline uint32_t to_base_code(uint32_t g) { return g ^ ((g >> 1) & 0x55555555u); }

// value of lane-1 (DPP wave_shr:1); lane 0 receives lane0_val
__device__ inline uint32_t wave_shr1(uint32_t v, uint32_t lane0_val)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0_val, (int)v, 0x138, 0xF, 0xF, false);
}

// The range of chunks workgroup blockIdx.x draws from.  Workgroups 8q .. 8q+7 (one per XCD, in dispatch order) share
// group q % n_groups, and with 512 workgroups on 32 groups so do both workgroups of a CU: neither the XCDs nor the two
// workgroups of a CU advance at the same pace (per-wave trace, RK_SCAN2_TRACE), and a counter shared across them lets
// the faster ones take more chunks.  gridDim.x is a multiple of 8 * n_groups, or n_groups is 1.
struct ChunkRange {
    uint32_t first, next, hi;  // this wave's first chunk; ticket t stands for chunk next + t; end of the range
    uint32_t *counter;
};
__device__ inline ChunkRange chunk_range(const SketchArgs &a, uint32_t wave)
{
    const uint32_t oct = blockIdx.x >> 3, g = oct % a.n_groups;
    const uint32_t wgs = a.n_groups == 1 ? gridDim.x : gridDim.x / a.n_groups;                    // workgroups of a group
    const uint32_t idx = a.n_groups == 1 ? blockIdx.x : oct / a.n_groups * 8 + (blockIdx.x & 7);  // mine among them
    const uint32_t lo = (uint32_t)((uint64_t)a.n_chunks * g / a.n_groups), hi = (uint32_t)((uint64_t)a.n_chunks * (g + 1) / a.n_groups);
    ChunkRange r;
    r.first = min(hi, lo + idx * kWavesPerBlock + wave);
    r.next = min(hi, lo + wgs * kWavesPerBlock);
    r.hi = hi;
    r.counter = a.tickets + (size_t)g * kTicketStride;
    return r;
}

// KS/OUT2: compile-time kmer_size and 2*half_outctx_len (0/-1: taken from the arguments)
template <int KS, int OUT2, bool EXACT, int IMG>
__global__ __launch_bounds__(kSketchThreads, Img<IMG>::kWavesPerEu) void rk_sketch_kernel(SketchArgs a)
{
    static_assert(!EXACT || Img<IMG>::kHasExact, "the exact table lives in the big image only");
    constexpr int kBitsA = Img<IMG>::kBitsA;
    constexpr uint32_t kWordsA = Img<IMG>::kWordsA, kWordsB = Img<IMG>::kWordsB;
    constexpr size_t kFilterLdsBytes = Img<IMG>::kFilterLdsBytes;
    constexpr uint32_t kStageCap = Img<IMG>::kStageCap;
    constexpr size_t kSketchLdsBytes = kFilterLdsBytes + stage_lds_bytes<IMG>();
    // static allocation: the compiler knows every LDS address and folds the table bases into the
    // offset field of ds_read (with a dynamic array each probe pays an extra v_add of the base)
    __shared__ __attribute__((aligned(16))) uint32_t lds[kSketchLdsBytes / 4];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.filter_image);
        uint4 *dst = reinterpret_cast<uint4 *>(lds);
        for (uint32_t i = threadIdx.x; i < kFilterLdsBytes / 16; i += kSketchThreads) dst[i] = src[i];
    }
    const uint32_t *bmA = lds;
    const uint32_t *bmB = lds + kWordsA;
    const uint32_t *keys = lds + kWordsA + kWordsB;
    const uint16_t *vals = reinterpret_cast<const uint16_t *>(keys + kExactSlots);
    unsigned long long *stage = reinterpret_cast<unsigned long long *>(lds + kFilterLdsBytes / 4) +
                                (threadIdx.x >> 6) * (kStageCap + 1);
    uint32_t *stage_n = reinterpret_cast<uint32_t *>(stage + kStageCap);
    if ((threadIdx.x & 63) == 0) *stage_n = 0;
    __syncthreads();

    const int k = KS ? KS : a.kmer;
    const int out2 = KS ? OUT2 : a.out2;
    // 4 * half_subk inner bits; a compile-time constant in the specialised kernels (k - 2 * outer context bases)
    const int dim_bits = KS ? 2 * KS - 2 * OUT2 : a.dim_bits;
    const int hi_shift = KS ? (2 * KS - 2 * OUT2 > Img<IMG>::kBitsB ? 2 * KS - 2 * OUT2 - Img<IMG>::kBitsB : 0) : a.hi_shift;
    const uint32_t dim_mask = (1u << dim_bits) - 1;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long windows = 0;
    uint32_t staged = 0;  // wave-uniform: windows in this wave's queue

    // A window that passed both bitmaps (0.2 % of all) is not resolved where it is found -- that would run the
    // reference arithmetic below with 1-2 of 64 lanes active, several times per block --: its 2k bits go to the
    // wave's queue in LDS, and once 64 are together every lane confirms one.
    // W: the window in the orientation of G (oldest base in the low bits, scan code).
    auto confirm = [&](uint64_t W, unsigned long long &key) -> bool {
        const uint64_t Bc = W ^ ((W >> 1) & 0x5555555555555555ULL);       // BaseMap codes (to_base_code)
        const uint64_t rvs = ~Bc & a.tupmask;                             // :499
        uint64_t r = __brevll(Bc);                                        // forward strand: base order reversed
        r = ((r >> 1) & 0x5555555555555555ULL) | ((r & 0x5555555555555555ULL) << 1);
        const uint64_t tuple = r >> (64 - 2 * k);                         // :498
        const uint64_t uni = tuple < rvs ? tuple : rvs;                   // :508
        const uint32_t dim = (uint32_t)(uni >> out2) & dim_mask;          // :509
        int32_t v = -1;
        if (EXACT) {
            uint32_t slot = (dim * 0x9E3779B1u) >> (32 - kExactSlotsLog2);
            for (;;) {
                const uint32_t kk = keys[slot];
                if (kk == dim) { v = (int32_t)vals[slot] + a.dim_start; break; }
                if (kk == kEmptyKey) break;
                slot = (slot + 1) & (kExactSlots - 1);
            }
        } else {
            v = a.table[dim];
        }
        if (v < a.dim_start || v >= a.dim_end) return false;              // :341,:516
        const uint64_t pf = (uint64_t)(v - a.dim_start);                  // :519-521
        key = (((uni & a.undomask0) | ((uni & a.undomask1) << a.und1_shift)) >> a.dr_shift) | pf;  // :524
        return true;
    };
    // wave-level drain of the queue (all 64 lanes call it together): 64 windows per round, one device-scope atomic
    // on the genome's counter and one burst of stores per round
    auto drain = [&](uint32_t gid, unsigned long long *region, uint32_t region_cap) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t n = min(*(volatile uint32_t *)stage_n, kStageCap);
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            unsigned long long key = 0;
            bool ok = false;
            if (i0 + lane < n) ok = confirm(stage[i0 + lane], key);
            const unsigned long long m = __ballot(ok);
            if (m) {
                uint32_t basep = 0;
                if (lane == 0) basep = atomicAdd(a.gcount + gid, (uint32_t)__popcll(m));
                basep = __shfl(basep, 0) + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (ok && basep < region_cap) region[basep] = key;
            }
        }
        if (n) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (lane == 0) *(volatile uint32_t *)stage_n = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    };

    // Chunks are handed out in order, not by striding (16,512 chunks on 8,192 striding waves ran for three rounds, not
    // 2.02).  One counter for the whole grid would serialise 10^4..10^5 same-address atomics (13 ns each, measured):
    // the chunks are cut into n_groups equal ranges with a counter each; a wave's first chunk is its index in its group.
    const ChunkRange cr = chunk_range(a, wave);
    for (uint32_t c = cr.first; c < cr.hi;) {
        uint32_t ticket = 0;
        if (lane == 0) ticket = atomicAdd(cr.counter, 1u);  // consumed at the end of this chunk
        const ChunkDesc cd = a.chunks[c];
        const uint32_t gid = cd.gid;
        const GenomeRow row = a.rows[gid];
        const uint32_t b0 = (c - row.first_chunk) * a.chunk_blocks;
        const uint32_t nb = cd.nb_first & 0x7FFFFFFFu;
        const uint32_t nbf = cd.nb_first & 0x80000000u;  // bit 31: first chunk of its genome
        const uint8_t *base = a.packed + row.beg + (size_t)b0 * 1024;
        unsigned long long *region = a.cand + row.reg_off;
        const uint32_t region_cap = row.reg_cap;

        // words of the two 16-base groups before the chunk (lanes "-2" and "-1")
        uint32_t cG2 = 0, cG1 = 0, cV2 = 0, cV1 = 0;
        if (!(nbf >> 31)) {
            uint4 w = make_uint4(0, 0, 0, 0);
            if (lane < 2) w = *reinterpret_cast<const uint4 *>(base - 32 + 16 * lane);
            uint32_t G, V;
            pack16(w, G, V);
            cG2 = __builtin_amdgcn_readlane(G, 0); cG1 = __builtin_amdgcn_readlane(G, 1);
            cV2 = __builtin_amdgcn_readlane(V, 0); cV1 = __builtin_amdgcn_readlane(V, 1);
        }

        // Two blocks per iteration: both are packed, exchanged and probed against bitmap A in one
        // straight-line stretch (32 independent LDS reads per lane in flight), which is what hides
        // the LDS latency with only four waves per SIMD.  Two more blocks are prefetched; the
        // index is clamped instead of branching so the compiler keeps the loads in flight across
        // the loop body.  A chunk with an odd number of blocks processes its last block twice and
        // discards the second result.
        auto load_block = [&](uint32_t blk) {
            return *reinterpret_cast<const uint4 *>(base + (size_t)min(blk, nb - 1) * 1024 + 16 * lane);
        };
        struct Blk {
            uint32_t G, G1, G2, bad, maybe;
        };
        // windows of a block that contain an invalid base (bit j: the window ending at base j)
        auto bad_windows = [&](uint32_t V, uint32_t V1, uint32_t V2) -> uint32_t {
            uint32_t bad = 0;
            if ((V & V1 & V2) != 0xFFFFu) {
                uint64_t inv = (uint64_t)(~V2 & 0xFFFFu) | ((uint64_t)(~V1 & 0xFFFFu) << 16) |
                               ((uint64_t)(~V & 0xFFFFu) << 32);
                int covered = 1;
                while (covered * 2 <= k) { inv |= inv << covered; covered *= 2; }
                if (k > covered) inv |= inv << (k - covered);
                bad = (uint32_t)(inv >> 32) & 0xFFFFu;
            }
            return bad;
        };
        const uint32_t off_mask = (dim_mask & ((1u << kBitsA) - 1)) >> 5 << 2;  // dims may have < 19 bits
        // level 1 (all 16 windows, branch-free).  x = inner bases of the window ending at base j in
        // the G orientation (= complement of the reverse strand's inner bases).  The selection
        // bitmaps are symmetric under reverse complement and stored pre-complemented, so x indexes
        // them directly: bitmap A on the low bits.
        auto probe_a = [&](const Blk &o) -> uint32_t {
            uint32_t acc = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                // bit x[18:0] of bitmap A: the word's byte offset (x >> 5) * 4 is cut straight out of
                // the base string (one funnel shift + mask), the bit position is x[4:0] (a variable shift reads
                // only the low 5 bits of its operand), and the bit enters the result from the top through a
                // funnel shift, which needs no masking: 5 VALU per window
                const int sh = 2 * (33 + j - k) + out2;
                const uint32_t x = ext96_lo(o.G, o.G1, o.G2, sh);
                const uint32_t off = ext96_lo(o.G, o.G1, o.G2, sh + 3) & off_mask;
                const uint32_t word = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(bmA) + off);
                acc = __builtin_amdgcn_alignbit(word >> (x & 31u), acc, 1);
            }
            return acc >> 16;  // window j at bit j
        };
        // level 2 (1.6-3 % of the windows): bitmap B on the high bits; the ~0.2 % that pass both are queued.
        // Returns the windows this lane queued.
        auto survivors = [&](const Blk &o) -> uint32_t {
            uint32_t queued = 0;
            uint32_t maybe = o.maybe;
            const uint32_t G = o.G, G1 = o.G1, G2 = o.G2;
            while (maybe) {
                const int j = __ffs((int)maybe) - 1;
                maybe &= maybe - 1;
                const int sh = 2 * (33 + j - k) + out2;  // 2..94
                uint32_t x;
                if (KS && 2 * (33 - KS) + OUT2 >= 32) {  // every window's inner bases start in G1 or G: one 64-bit shift
                    x = (uint32_t)(((((uint64_t)G) << 32) | G1) >> (sh - 32));
                } else {
                    const uint32_t wa = sh < 32 ? G2 : (sh < 64 ? G1 : G);
                    const uint32_t wb = sh < 32 ? G1 : (sh < 64 ? G : 0u);
                    x = __builtin_amdgcn_alignbit(wb, wa, sh & 31);
                }
                const uint32_t ib = (x & dim_mask) >> hi_shift;
                if (!((bmB[ib >> 5] >> (ib & 31)) & 1u)) continue;
                const uint64_t W = ext96v(G, G1, G2, 2 * (33 + j - k)) & a.tupmask;
                const uint32_t sl = atomicAdd(stage_n, 1u);
                queued++;
                if (sl < kStageCap) stage[sl] = W;
                else {  // queue full inside one block (low-complexity sequence): resolve in place
                    unsigned long long key;
                    if (confirm(W, key)) {
                        const uint32_t slot = atomicAdd(a.gcount + gid, 1u);
                        if (slot < region_cap) region[slot] = key;
                    }
                }
            }
            return queued;
        };

        uint4 c0 = load_block(0), c1 = load_block(1);
        if (IMG == 1) {
            // small image, 8 waves per SIMD: one block per iteration keeps the kernel within 64 VGPRs; the other
            // waves of the SIMD supply the independent work
            for (uint32_t b = 0; b < nb; b++) {
                const uint4 n0 = load_block(b + 2);
                Blk A;
                uint32_t VA;
                pack16(c0, A.G, VA);
                A.G1 = wave_shr1(A.G, cG1); A.G2 = wave_shr1(A.G1, cG2);
                const uint32_t VA1 = wave_shr1(VA, cV1), VA2 = wave_shr1(VA1, cV2);
                cG2 = __builtin_amdgcn_readlane(A.G, 62); cG1 = __builtin_amdgcn_readlane(A.G, 63);
                cV2 = __builtin_amdgcn_readlane(VA, 62); cV1 = __builtin_amdgcn_readlane(VA, 63);
                A.bad = bad_windows(VA, VA1, VA2);
                windows += __popc(~A.bad & 0xFFFFu);
                A.maybe = probe_a(A) & ~A.bad;
                const uint32_t queued = survivors(A);
                for (uint32_t lvl = 1; ; lvl++) {
                    const unsigned long long m = __ballot(queued >= lvl);
                    if (!m) break;
                    staged += __popcll(m);
                }
                c0 = c1;
                c1 = n0;
                if (staged >= 64) { drain(gid, region, region_cap); staged = 0; }
            }
        } else
        for (uint32_t b = 0; b < nb; b += 2) {
            const uint4 n0 = load_block(b + 2), n1 = load_block(b + 3);
            const bool has1 = b + 1 < nb;  // uniform

            Blk A, B;
            uint32_t VA, VB;
            pack16(c0, A.G, VA);
            pack16(c1, B.G, VB);
            A.G1 = wave_shr1(A.G, cG1); A.G2 = wave_shr1(A.G1, cG2);
            const uint32_t VA1 = wave_shr1(VA, cV1), VA2 = wave_shr1(VA1, cV2);
            cG2 = __builtin_amdgcn_readlane(A.G, 62); cG1 = __builtin_amdgcn_readlane(A.G, 63);
            cV2 = __builtin_amdgcn_readlane(VA, 62); cV1 = __builtin_amdgcn_readlane(VA, 63);
            B.G1 = wave_shr1(B.G, cG1); B.G2 = wave_shr1(B.G1, cG2);
            const uint32_t VB1 = wave_shr1(VB, cV1), VB2 = wave_shr1(VB1, cV2);
            cG2 = __builtin_amdgcn_readlane(B.G, 62); cG1 = __builtin_amdgcn_readlane(B.G, 63);
            cV2 = __builtin_amdgcn_readlane(VB, 62); cV1 = __builtin_amdgcn_readlane(VB, 63);
            A.bad = bad_windows(VA, VA1, VA2);
            B.bad = has1 ? bad_windows(VB, VB1, VB2) : 0xFFFFu;
            windows += __popc(~A.bad & 0xFFFFu) + __popc(~B.bad & 0xFFFFu);

            const uint32_t ma = probe_a(A), mb = probe_a(B);
            A.maybe = ma & ~A.bad;
            B.maybe = mb & ~B.bad;

            uint32_t queued = survivors(A);
            queued += survivors(B);
            for (uint32_t lvl = 1; ; lvl++) {  // wave-uniform count of staged keys, no LDS round trip
                const unsigned long long m = __ballot(queued >= lvl);
                if (!m) break;
                staged += __popcll(m);
            }
            c0 = n0;
            c1 = n1;
            if (staged >= 64) { drain(gid, region, region_cap); staged = 0; }
        }
        drain(gid, region, region_cap);
        staged = 0;
        c = cr.next + __builtin_amdgcn_readfirstlane(ticket);
    }
    for (int o = 32; o > 0; o >>= 1) windows += __shfl_down(windows, o);
    if (lane == 0 && windows) atomicAdd(a.n_windows, windows);
}

#include "rk_sketch_scan2.inc"

// ---- per-genome dedup (the reference's unordered_set, src/sketch.cpp:470,526-529,537-550) -------------------
// One workgroup per genome: its candidate region (~1,250 dr_tuples for a 5 Mb genome at L3) is sorted in LDS
// (bitonic), equal neighbours collapse (FASTQ: a hash survives only with >= min_count occurrences, :828-845), and
// the sorted distinct hashes are written back coalesced together with their number.  Replaces the ~20 launches
// of a device-wide radix sort + unique + split that doubled the device time of a batch in round 1.
// 16 waves per genome: a step of the sort is ~35 dependent instructions per wave, and one wave per SIMD issues them
// one every few cycles (256 threads: 41 us for 128 genomes of ~1,250 candidates; 1,024: see DESIGN.md 4.1)
constexpr uint32_t kDedupThreads = 1024;
constexpr uint32_t kDedupMaxBytes = 64 * 1024;  // LDS sort capacity: 16,384 32-bit or 8,192 64-bit keys

enum : uint32_t { kFlagOverflow = 1 };

struct SketchTail {  // device-side results of one sketch pass, read back in one copy
    unsigned long long windows, flags;
};

template <class K>
__global__ __launch_bounds__(kDedupThreads) void k_dedup(const unsigned long long *cand, const GenomeRow *rows,
                                                        const uint32_t *gcount, uint32_t min_count, K *sorted_out,
                                                        uint32_t *usize, SketchTail *tail)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dedup_lds[];
    K *a = reinterpret_cast<K *>(dedup_lds);
    __shared__ uint32_t wave_tot[kDedupThreads / 64];
    const uint32_t g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const GenomeRow row = rows[g];
    if (row.is_big) return;  // sorted by the device-wide path
    const uint32_t n = gcount[g];
    if (n > row.reg_cap) {  // the region overflowed: the host reruns the pass with the exact capacities
        if (tid == 0) {
            usize[g] = 0;
            atomicOr(&tail->flags, (unsigned long long)kFlagOverflow);
        }
        return;
    }
    uint32_t P = 1;
    while (P < n) P <<= 1;
    const unsigned long long *src = cand + row.reg_off;
    // padding keys are all ones: they sort behind every real key (or tie with it), so the first n are the real ones
    for (uint32_t i = tid; i < P; i += kDedupThreads) a[i] = i < n ? (K)src[i] : (K)~(K)0;
    __syncthreads();
    // Pair t compares elements (i, i + j).  A wave owns pairs 64w .. 64w+63 (+ multiples of the workgroup size): for
    // j <= 64 these are exactly the elements 128w .. 128w+127, in every such step -- the wave only ever meets its own
    // earlier writes, and LDS operations of one wave complete in order.  Only the steps that reach across 128-element
    // blocks (j >= 128), and the step before one, end with a workgroup barrier: 10 of the 66 steps for 2,048 keys.
    static_assert(kDedupThreads % 64 == 0, "whole waves");
    for (uint32_t k = 2; k <= P; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < P / 2; t += kDedupThreads) {
                const uint32_t i = 2 * t - (t & (j - 1));  // partner pairs (i, i + j)
                const bool up = (i & k) == 0;
                const K x = a[i], y = a[i + j];
                if ((x > y) == up) {
                    a[i] = y;
                    a[i + j] = x;
                }
            }
            if (j >= 128 || (j == 1 && k >= 128)) __syncthreads();
            else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    __syncthreads();
    // run heads that are kept; every thread owns a contiguous stretch so that the output stays sorted
    const uint32_t per = (n + kDedupThreads - 1) / kDedupThreads;
    const uint32_t i0 = min(n, tid * per), i1 = min(n, i0 + per);
    auto kept = [&](uint32_t i) -> bool {
        if (i && a[i] == a[i - 1]) return false;
        if (min_count <= 1) return true;
        uint32_t len = 1;
        while (len < min_count && i + len < n && a[i + len] == a[i]) len++;
        return len >= min_count;
    };
    uint32_t mine = 0;
    for (uint32_t i = i0; i < i1; i++) mine += kept(i);
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if ((int)lane >= o) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t at = incl - mine;
    for (uint32_t w = 0; w < wave; w++) at += wave_tot[w];
    K *dst = sorted_out + row.reg_off;
    for (uint32_t i = i0; i < i1; i++)
        if (kept(i)) dst[at++] = a[i];
    if (tid == kDedupThreads - 1) usize[g] = at;
}

// single workgroup: off = exclusive scan of the per-genome sizes
__global__ void k_size_scan(const uint32_t *usize, uint32_t n_genomes, uint64_t *off, uint64_t *off_copy)
{
    __shared__ unsigned long long part[1024 / 64];
    __shared__ unsigned long long carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_genomes; base += blockDim.x) {
        const uint32_t g = base + tid;
        const unsigned long long len = g < n_genomes ? usize[g] : 0ULL;
        unsigned long long incl = len;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = __shfl_up(incl, o);
            if ((int)lane >= o) incl += t;
        }
        if (lane == 63) part[wave] = incl;
        __syncthreads();
        unsigned long long before = carry;
        for (uint32_t w = 0; w < wave; w++) before += part[w];
        if (g < n_genomes) off[g] = off_copy[g] = before + incl - len;
        __syncthreads();
        if (tid == 0) {
            unsigned long long t = carry;
            for (uint32_t w = 0; w < nw; w++) t += part[w];
            carry = t;
        }
        __syncthreads();
    }
    if (tid == 0) off[n_genomes] = off_copy[n_genomes] = carry;
}

// one workgroup per genome: its sorted distinct hashes move to their place in the CSR
template <class K>
__global__ void k_csr_place(const K *sorted_out, const GenomeRow *rows, const uint32_t *usize, const uint64_t *off, K *hashes)
{
    const uint32_t g = blockIdx.x;
    const K *src = sorted_out + rows[g].reg_off;
    K *dst = hashes + off[g];
    for (uint32_t i = threadIdx.x; i < usize[g]; i += blockDim.x) dst[i] = src[i];
}

// big genomes (candidate region beyond the LDS sort capacity): device-wide sort + unique, then narrow the keys
__global__ void k_count_flags(const unsigned int *counts, uint64_t n, uint32_t min_count, unsigned char *flags)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = counts[i] >= min_count ? 1 : 0;
}

template <class K> __global__ void k_narrow_keys(const unsigned long long *ukeys, uint64_t n, K *out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (K)ukeys[i];
}

// every genome's hashes strictly ascending?  A descent may only sit where a new genome starts: the rare candidates
// look their position up in the offsets
template <class K> __global__ void k_check_sets(const K *hashes, uint64_t total, const uint64_t *off, uint32_t n_genomes, uint32_t *bad)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (e >= total || hashes[e] > hashes[e - 1]) return;
    uint32_t lo = 0, hi = n_genomes;  // largest g with off[g] <= e
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= e) lo = mid; else hi = mid;
    }
    if (off[lo] != e) *bad = 1;
}

inline unsigned blocks_for(uint64_t n, int t = 256) { return (unsigned)((n + t - 1) / t); }

// the chunk table of a pass: chunk c belongs to the largest g with first_chunk[g] <= c
__global__ void k_chunk_table(const GenomeRow *rows, uint32_t n_genomes, uint32_t n_chunks, uint32_t chunk_blocks, ChunkDesc *out)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    uint32_t g = 0;
    for (uint32_t hi = n_genomes; hi - g > 1;) {
        const uint32_t mid = (g + hi) >> 1;
        if (rows[mid].first_chunk <= c) g = mid; else hi = mid;
    }
    const uint32_t b0 = (c - rows[g].first_chunk) * chunk_blocks;
    ChunkDesc d;
    d.beg = rows[g].beg + (uint64_t)b0 * 1024;
    d.nb_first = min(chunk_blocks, rows[g].nblk - b0) | (b0 == 0 ? 0x80000000u : 0u);
    d.gid = g;
    out[c] = d;
}

typedef void (*sketch_kernel_t)(SketchArgs);
sketch_kernel_t pick_kernel(int kmer, int out2, bool exact, int img)
{
    if (img == 2) {  // the two-stage scan (rk_sketch_scan2.inc): compile-time parameter sets only
        if (kmer == 20 && out2 == 8) return rk_scan2_kernel<20, 8>;  // K10 S6
        if (kmer == 20 && out2 == 6) return rk_scan2_kernel<20, 6>;  // K10 S7
        if (kmer == 16 && out2 == 6) return rk_scan2_kernel<16, 6>;  // K8 S5
        return nullptr;
    }
#define RK_SK(K, O) (img ? rk_sketch_kernel<K, O, false, 1> : (exact ? rk_sketch_kernel<K, O, true, 0> : rk_sketch_kernel<K, O, false, 0>))
    if (kmer == 20 && out2 == 8) return RK_SK(20, 8);   // K10 S6
    if (kmer == 20 && out2 == 6) return RK_SK(20, 6);   // K10 S7
    if (kmer == 16 && out2 == 6) return RK_SK(16, 6);   // K8 S5
    return RK_SK(0, 0);
#undef RK_SK
}

}  // namespace
extern "C" {

int rk_filter_create(rk_ctx *ctx, const rk_params *p, const int32_t *shuffled_dim, rk_filter **out)
{
    if (!ctx || !p || !shuffled_dim || !out) return RK_ERR_ARG;
    *out = nullptr;
    if (p->half_subk < 1 || p->half_subk >= 8 || p->half_k > 16 || p->half_k < p->half_subk)
        return rk_fail(ctx, RK_ERR_ARG, "bad kssd parameters");
    RK_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n = 1ULL << (4 * p->half_subk);
    // selected entries (value in [dim_start, dim_end)), src/sketch.cpp:338-345
    std::vector<uint32_t> sel_key;
    std::vector<uint32_t> sel_val;
    for (uint64_t t = 0; t < n; t++) {
        const int32_t v = shuffled_dim[t];
        if (v >= p->dim_start && v < p->dim_end) {
            sel_key.push_back((uint32_t)t);
            sel_val.push_back((uint32_t)(v - p->dim_start));
        }
    }
    int img = ctx->sw_sketch_img;
    if (img == 2 && !pick_kernel(2 * p->half_k, 2 * p->half_outctx_len, false, 2)) img = 1;  // no compile-time variant
    const int kBitsA = img == 2 ? 18 : (img ? Img<1>::kBitsA : Img<0>::kBitsA);
    const int kBitsB = img == 2 ? Scan2::kBitsB : (img ? Img<1>::kBitsB : Img<0>::kBitsB);
    const uint32_t kWordsA = (1u << kBitsA) / 32;
    const size_t kFilterLdsBytes = img == 2 ? Scan2::kFilterLdsBytes : (img ? Img<1>::kFilterLdsBytes : Img<0>::kFilterLdsBytes);
    const bool exact = img == 0 && sel_key.size() <= kExactSlots / 2 && (p->dim_end - p->dim_start) <= 65536;
    std::vector<uint32_t> image(kFilterLdsBytes / 4, 0);
    const int dim_bits = 4 * p->half_subk;
    const int hi_shift = std::max(0, dim_bits - kBitsB);
    auto revcomp_dim = [&](uint32_t d) {  // reverse complement of the 2*half_subk inner bases
        uint32_t r = 0;
        for (int i = 0; i < dim_bits / 2; i++) r |= ((~(d >> (2 * i))) & 3u) << (dim_bits - 2 - 2 * i);
        return r;
    };
    const uint32_t dmask = (uint32_t)((1ULL << dim_bits) - 1);
    auto set_bits = [&](uint32_t dd) {
        // the kernel indexes with the COMPLEMENT of the reverse strand's inner bases.  The
        // shuffle is a random permutation, so both bit fields of a selected index are uniform
        // ... and in scan code (pack16), hence to_base_code(), which maps either code to the other
        const uint32_t d = to_base_code(~dd & dmask) & dmask;
        const uint32_t ia = d & ((1u << kBitsA) - 1), ib = d >> hi_shift;
        if (img == 2) image[Scan2::word_a(d)] |= 1u << Scan2::bit_a(d);
        else image[ia >> 5] |= 1u << (ia & 31);
        image[kWordsA + (ib >> 5)] |= 1u << (ib & 31);
    };
    for (uint32_t key : sel_key) {
        set_bits(key);
        set_bits(revcomp_dim(key));
    }
    if (exact) {
        uint32_t *keys = image.data() + kWordsA + (1u << kBitsB) / 32;
        uint16_t *vals = reinterpret_cast<uint16_t *>(keys + kExactSlots);
        for (uint32_t i = 0; i < kExactSlots; i++) keys[i] = kEmptyKey;
        for (size_t i = 0; i < sel_key.size(); i++) {
            uint32_t slot = (sel_key[i] * 0x9E3779B1u) >> (32 - kExactSlotsLog2);
            while (keys[slot] != kEmptyKey) slot = (slot + 1) & (kExactSlots - 1);
            keys[slot] = sel_key[i];
            vals[slot] = (uint16_t)sel_val[i];
        }
    }
    DevBuf<int32_t> table(ctx);
    DevBuf<uint32_t> d_image(ctx);
    RK_HIP(ctx, table.alloc(n));
    RK_HIP(ctx, d_image.alloc(kFilterLdsBytes / 4));
    RK_HIP(ctx, hipMemcpyAsync(table.p, shuffled_dim, n * 4, hipMemcpyHostToDevice, ctx->stream));
    RK_HIP(ctx, hipMemcpyAsync(d_image.p, image.data(), kFilterLdsBytes, hipMemcpyHostToDevice, ctx->stream));
    RK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rk_filter *f = new (std::nothrow) rk_filter;
    if (!f) return RK_ERR_NOMEM;
    f->ctx = ctx;
    f->params = *p;
    f->d_table = table.release();
    f->d_bitmap = d_image.release();
    f->bitmap_bits = kBitsA;
    f->img = img;
    f->exact = exact;
    f->n_keys = (uint32_t)sel_key.size();
    *out = f;
    return RK_OK;
}

void rk_filter_free(rk_filter *f)
{
    if (!f) return;
    rk_pool_free(f->ctx, f->d_table);
    rk_pool_free(f->ctx, f->d_bitmap);
    delete f;
}

void rk_sketches_free(rk_sketches *s)
{
    if (!s) return;
    rk_pool_free(s->ctx, s->d_hashes);
    rk_pool_free(s->ctx, s->d_hashes64);
    rk_pool_free(s->ctx, s->d_off);
    rk_pool_free(s->ctx, s->d_member_rec);
    rk_pool_free(s->ctx, s->d_member_seg);
    delete s;
}

uint32_t rk_sketches_count(const rk_sketches *s) { return s ? s->n : 0; }
uint64_t rk_sketches_total(const rk_sketches *s) { return s ? s->total : 0; }
uint64_t rk_sketches_windows(const rk_sketches *s) { return s ? s->windows : 0; }
const uint32_t *rk_sketches_hashes_dev(const rk_sketches *s) { return s ? s->d_hashes : nullptr; }
const uint64_t *rk_sketches_off_dev(const rk_sketches *s) { return s ? s->d_off : nullptr; }

int rk_sketches_from_host(rk_ctx *ctx, const uint32_t *hashes, const uint64_t *off, uint32_t n,
                          rk_sketches **out)
{
    if (!ctx || !off || !out || (!hashes && off[n])) return RK_ERR_ARG;
    *out = nullptr;
    for (uint32_t g = 0; g < n; g++)
        if (off[g + 1] < off[g]) return rk_fail(ctx, RK_ERR_ARG, "offsets must be non-decreasing");
    if (off[0] != 0) return rk_fail(ctx, RK_ERR_ARG, "off[0] must be 0");
    RK_HIP(ctx, hipSetDevice(ctx->device));
    rk_sketches *s = new (std::nothrow) rk_sketches;
    if (!s) return RK_ERR_NOMEM;
    s->ctx = ctx;
    s->n = n;
    s->total = off[n];
    s->h_off.assign(off, off + n + 1);
    struct Guard { rk_sketches *p; ~Guard() { if (p) rk_sketches_free(p); } } guard{s};
    RK_TRY(pool_array(ctx, &s->d_hashes, s->total + 1));
    RK_TRY(pool_array(ctx, &s->d_off, (size_t)n + 1));
    if (s->total) RK_HIP(ctx, hipMemcpyAsync(s->d_hashes, hashes, s->total * 4, hipMemcpyHostToDevice, ctx->stream));
    RK_HIP(ctx, hipMemcpyAsync(s->d_off, off, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    RK_TRY(rk_sketches_classify(ctx, s));  // synchronises the stream
    guard.p = nullptr;
    *out = s;
    return RK_OK;
}

int rk_sketches_from_dev(rk_ctx *ctx, const uint32_t *hashes_dev, const uint64_t *off_dev, uint32_t n,
                         rk_sketches **out)
{
    if (!ctx || !off_dev || !out) return RK_ERR_ARG;
    *out = nullptr;
    RK_HIP(ctx, hipSetDevice(ctx->device));
    rk_sketches *s = new (std::nothrow) rk_sketches;
    if (!s) return RK_ERR_NOMEM;
    s->ctx = ctx;
    s->n = n;
    struct Guard { rk_sketches *p; ~Guard() { if (p) rk_sketches_free(p); } } guard{s};
    s->h_off.resize((size_t)n + 1);
    RK_HIP(ctx, hipMemcpyAsync(s->h_off.data(), off_dev, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    RK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // (the offsets steer every kernel that walks the sketches: a table that is not a CSR table -- e.g. one whose producer on
    // another stream had not finished when this call was made -- is refused here, not met as a memory fault later)
    if (s->h_off[0] != 0) return rk_fail(ctx, RK_ERR_ARG, "rk_sketches_from_dev: off[0] must be 0");
    for (uint32_t g = 0; g < n; g++)
        if (s->h_off[g + 1] < s->h_off[g])
            return rk_fail(ctx, RK_ERR_ARG, "rk_sketches_from_dev: offsets must be non-decreasing (genome %u; were the arrays complete when the call was made?)", g);
    s->total = s->h_off[n];
    if (!hashes_dev && s->total) return rk_fail(ctx, RK_ERR_ARG, "hashes_dev is NULL");
    RK_TRY(pool_array(ctx, &s->d_hashes, s->total + 1));
    RK_TRY(pool_array(ctx, &s->d_off, (size_t)n + 1));
    if (s->total) RK_HIP(ctx, hipMemcpyAsync(s->d_hashes, hashes_dev, s->total * 4, hipMemcpyDeviceToDevice, ctx->stream));
    RK_HIP(ctx, hipMemcpyAsync(s->d_off, off_dev, ((size_t)n + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
    RK_TRY(rk_sketches_classify(ctx, s));
    guard.p = nullptr;
    *out = s;
    return RK_OK;
}

int rk_sketches_is64(const rk_sketches *s) { return s && s->wide ? 1 : 0; }

int rk_sketches_from_host64(rk_ctx *ctx, const uint64_t *hashes, const uint64_t *off, uint32_t n, rk_sketches **out)
{
    if (!ctx || !off || !out || (!hashes && off[n])) return RK_ERR_ARG;
    *out = nullptr;
    for (uint32_t g = 0; g < n; g++)
        if (off[g + 1] < off[g]) return rk_fail(ctx, RK_ERR_ARG, "offsets must be non-decreasing");
    if (off[0] != 0) return rk_fail(ctx, RK_ERR_ARG, "off[0] must be 0");
    RK_HIP(ctx, hipSetDevice(ctx->device));
    rk_sketches *s = new (std::nothrow) rk_sketches;
    if (!s) return RK_ERR_NOMEM;
    s->ctx = ctx;
    s->n = n;
    s->wide = true;
    s->total = off[n];
    s->h_off.assign(off, off + n + 1);
    struct Guard { rk_sketches *p; ~Guard() { if (p) rk_sketches_free(p); } } guard{s};
    RK_TRY(pool_array(ctx, &s->d_hashes64, s->total + 1));
    RK_TRY(pool_array(ctx, &s->d_off, (size_t)n + 1));
    if (s->total) RK_HIP(ctx, hipMemcpyAsync(s->d_hashes64, hashes, s->total * 8, hipMemcpyHostToDevice, ctx->stream));
    RK_HIP(ctx, hipMemcpyAsync(s->d_off, off, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    RK_TRY(rk_sketches_classify(ctx, s));
    guard.p = nullptr;
    *out = s;
    return RK_OK;
}

int rk_sketches_download64(const rk_sketches *s, uint64_t *hashes, uint64_t *off)
{
    if (!s) return RK_ERR_ARG;
    rk_ctx *ctx = s->ctx;
    if (!s->wide) return rk_fail(ctx, RK_ERR_ARG, "32-bit sketches: use rk_sketches_download");
    RK_HIP(ctx, hipSetDevice(ctx->device));
    if (off) memcpy(off, s->h_off.data(), ((size_t)s->n + 1) * 8);
    if (hashes && s->total) {
        RK_HIP(ctx, hipMemcpyAsync(hashes, s->d_hashes64, s->total * 8, hipMemcpyDeviceToHost, ctx->stream));
        RK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return RK_OK;
}

int rk_sketches_download(const rk_sketches *s, uint32_t *hashes, uint64_t *off)
{
    if (!s) return RK_ERR_ARG;
    rk_ctx *ctx = s->ctx;
    if (s->wide && hashes) return rk_fail(ctx, RK_ERR_ARG, "64-bit sketches: use rk_sketches_download64");
    RK_HIP(ctx, hipSetDevice(ctx->device));
    if (off) memcpy(off, s->h_off.data(), ((size_t)s->n + 1) * 8);
    if (hashes && s->total) {
        RK_HIP(ctx, hipMemcpyAsync(hashes, s->d_hashes, s->total * 4, hipMemcpyDeviceToHost, ctx->stream));
        RK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return RK_OK;
}

int rk_sketch_packed_dev(rk_ctx *ctx, const rk_filter *f, const uint8_t *packed_dev,
                         uint64_t packed_bytes, const uint64_t *gbeg, const uint64_t *gend,
                         uint32_t n_genomes, void *stream_v, rk_sketches **out)
{
    return rk_sketch_packed_dev_ex(ctx, f, packed_dev, packed_bytes, gbeg, gend, n_genomes, 1, stream_v, out);
}

}  // extern "C"

// one big genome's candidate region: device-wide radix sort + unique (or run-length select), narrowed into its
// slot of sorted_out; returns the number of distinct hashes kept
template <class K>
static int dedup_big(rk_ctx *ctx, unsigned long long *region, uint32_t n, int hash_bits, uint32_t min_count, K *dst,
                     uint32_t *n_out, hipStream_t stream)
{
    *n_out = 0;
    if (!n) return RK_OK;
    DevBuf<unsigned long long> sorted(ctx), uniq(ctx), d_n(ctx);
    RK_HIP(ctx, sorted.alloc(n));
    RK_HIP(ctx, uniq.alloc(n));
    RK_HIP(ctx, d_n.alloc(1));
    RK_TRY(rk_prim_sort_keys_u64(ctx, region, sorted.p, n, 0, (unsigned)hash_bits, stream));
    unsigned long long nu = 0;
    if (min_count <= 1) {  // set semantics (FASTA, src/sketch.cpp:526-529)
        RK_TRY(rk_prim_unique_u64(ctx, sorted.p, uniq.p, d_n.p, n, stream));
        RK_TRY(rk_read_back(ctx, &nu, d_n.p, 8, stream));
    } else {  // FASTQ: keep a hash only if it occurred >= min_count times (src/sketch.cpp:828-845)
        DevBuf<unsigned long long> runs(ctx);
        DevBuf<unsigned int> counts(ctx);
        DevBuf<unsigned char> flags(ctx);
        RK_HIP(ctx, runs.alloc(n));
        RK_HIP(ctx, counts.alloc(n));
        RK_HIP(ctx, flags.alloc(n));
        RK_TRY(rk_prim_rle_u64(ctx, sorted.p, n, runs.p, counts.p, d_n.p, stream));
        unsigned long long n_runs = 0;
        RK_TRY(rk_read_back(ctx, &n_runs, d_n.p, 8, stream));
        if (n_runs) {
            hipLaunchKernelGGL(k_count_flags, dim3(blocks_for(n_runs)), dim3(256), 0, stream, counts.p, n_runs, min_count, flags.p);
            RK_TRY(rk_prim_select_u64(ctx, runs.p, flags.p, uniq.p, d_n.p, n_runs, stream));
            RK_TRY(rk_read_back(ctx, &nu, d_n.p, 8, stream));
        }
    }
    if (nu) hipLaunchKernelGGL(k_narrow_keys<K>, dim3(blocks_for(nu)), dim3(256), 0, stream, uniq.p, nu, dst);
    RK_HIP(ctx, hipGetLastError());
    RK_HIP(ctx, hipStreamSynchronize(stream));  // the temporaries above return to the pool
    *n_out = (uint32_t)nu;
    return RK_OK;
}

extern "C" {

int rk_sketch_packed_dev_ex(rk_ctx *ctx, const rk_filter *f, const uint8_t *packed_dev,
                            uint64_t packed_bytes, const uint64_t *gbeg, const uint64_t *gend,
                            uint32_t n_genomes, uint32_t min_count, void *stream_v, rk_sketches **out)
{
    if (!ctx || !f || !out || (!packed_dev && packed_bytes) || ((!gbeg || !gend) && n_genomes))
        return RK_ERR_ARG;
    *out = nullptr;
    hipStream_t stream = (hipStream_t)stream_v;
    RK_HIP(ctx, hipSetDevice(ctx->device));
    const rk_params &P = f->params;
    const int hash_bits = rk_hash_bits(&P);
    const bool wide = hash_bits > 32;  // use64 layout
    const size_t key_bytes = wide ? 8 : 4;

    // ---- the pass table: one row per genome (one wave walks one chunk of `cb` consecutive 1 KiB blocks)
    uint64_t total_blocks = 0;
    for (uint32_t g = 0; g < n_genomes; g++) {
        if (gend[g] < gbeg[g] || (gbeg[g] & 1023) || gend[g] > packed_bytes)
            return rk_fail(ctx, RK_ERR_ARG, "genome %u: bad packed range", g);
        if (((gend[g] + 1023) & ~1023ULL) > packed_bytes && gend[g] > gbeg[g])
            return rk_fail(ctx, RK_ERR_ARG, "packed buffer must be padded to a multiple of 1024 bytes");
        if (((gend[g] - gbeg[g] + 1023) >> 10) > 0x7FFFFFFFULL) return rk_fail(ctx, RK_ERR_UNSUPPORTED, "genome %u is too long", g);
        total_blocks += (gend[g] - gbeg[g] + 1023) >> 10;
    }
    // Chunk length: (nearly) the smallest for which there are at most two chunks per wave (2 workgroups of 16 waves per CU).
    // The waves run at the same speed, so the kernel lasts ceil(chunks / waves) chunks whatever the order they are
    // handed out in: 2.4 chunks per wave cost three rounds (measured: 32 blocks 0.242 ms, 38 blocks 0.211 ms on
    // 128 x 5 Mb); and a chunk has a fixed cost (its genome's row, the 32 bases before it, the last partial rounds
    // of its queues), so more and shorter chunks are slower too (20 blocks: 0.245 ms).
    // (a genome adds at most one partial chunk: total / len + n_genomes bounds the count from above)
    const uint64_t wave_slots = (uint64_t)ctx->num_cu * 2 * kWavesPerBlock, limit = 2 * wave_slots;
    uint64_t cb = n_genomes < limit ? (total_blocks + (limit - n_genomes) - 1) / (limit - n_genomes) : (total_blocks + limit - 1) / limit;
    cb = std::min<uint64_t>(1u << 20, std::max<uint64_t>(16, cb));
    if (getenv("RK_SKETCH_CB")) cb = std::min<uint64_t>(1u << 20, std::max<uint64_t>(1, strtoull(getenv("RK_SKETCH_CB"), nullptr, 10)));
    std::vector<GenomeRow> rows((size_t)n_genomes + 1);
    uint64_t n_chunks64 = 0;
    for (uint32_t g = 0; g < n_genomes; g++) {
        GenomeRow &r = rows[g];
        r.beg = gbeg[g];
        r.nblk = (uint32_t)((gend[g] - gbeg[g] + 1023) >> 10);
        r.first_chunk = (uint32_t)n_chunks64;
        n_chunks64 += (r.nblk + cb - 1) / cb;
        // candidate region: expected survivors = windows / 16^drlevel; x2 + slack, exact retry on overflow
        const uint64_t want = 2 * ((gend[g] - gbeg[g]) >> (4 * P.drlevel)) + 192;
        if (want > 0xFFFFFFF0ULL || n_chunks64 > 0xFFFFFFF0ULL)
            return rk_fail(ctx, RK_ERR_UNSUPPORTED, "genome %u is too long for one sketch pass", g);
        r.reg_cap = (uint32_t)want;
    }
    const uint32_t n_chunks = (uint32_t)n_chunks64;
    rows[n_genomes] = GenomeRow{0, 0, 0, n_chunks, 0, 0};

    // device side of the pass: table in, results out -- one upload and one read-back through the pinned scratch
    const size_t rows_bytes = rows.size() * sizeof(GenomeRow);
    const size_t res_bytes = sizeof(SketchTail) + ((size_t)n_genomes + 1) * 8 + (size_t)n_genomes * 4;  // tail | off | gcount
    char *pinned = static_cast<char *>(rk_pinned_scratch(ctx, std::max(rows_bytes, res_bytes)));
    if (!pinned) return rk_fail(ctx, RK_ERR_NOMEM, "cannot pin %llu bytes", (unsigned long long)std::max(rows_bytes, res_bytes));
    DevBuf<GenomeRow> d_rows(ctx);
    DevBuf<ChunkDesc> d_chunks(ctx);
    DevBuf<char> d_res(ctx);  // tail | off | gcount (read back) | per-genome sketch sizes: cleared with ONE fill per pass
    const size_t tickets_off = (res_bytes + (size_t)n_genomes * 4 + 127) & ~(size_t)127;
    const size_t work_bytes = tickets_off + kTicketGroups * kTicketStride * 4;  // ... | the scan kernel's ticket counters
    RK_HIP(ctx, d_rows.alloc(rows.size()));
    RK_HIP(ctx, d_res.alloc(work_bytes));
    SketchTail *d_tail = reinterpret_cast<SketchTail *>(d_res.p);
    uint64_t *d_off_copy = reinterpret_cast<uint64_t *>(d_res.p + sizeof(SketchTail));
    uint32_t *d_gcount = reinterpret_cast<uint32_t *>(d_res.p + sizeof(SketchTail) + ((size_t)n_genomes + 1) * 8);
    struct { uint32_t *p; } d_usize{d_gcount + n_genomes};

    rk_sketches *s = new (std::nothrow) rk_sketches;
    if (!s) return RK_ERR_NOMEM;
    s->ctx = ctx;
    s->n = n_genomes;
    s->wide = wide;
    s->is_set = true;  // sorted, distinct
    struct Guard { rk_sketches *p; ~Guard() { if (p) rk_sketches_free(p); } } guard{s};
    RK_TRY(pool_array(ctx, &s->d_off, (size_t)n_genomes + 1));
    s->h_off.assign((size_t)n_genomes + 1, 0);

    sketch_kernel_t kern = pick_kernel((int)P.kmer_size, 2 * P.half_outctx_len, f->exact, f->img);
    DevBuf<unsigned long long> cand(ctx);
    DevBuf<char> sorted_out(ctx);
    std::vector<uint32_t> gcount(n_genomes, 0);
    SketchTail tail{0, 0};
    for (int attempt = 0; attempt < 2; attempt++) {
        uint64_t total_cap = 0;
        size_t small_lds = key_bytes;
        bool any_big = false;
        for (uint32_t g = 0; g < n_genomes; g++) {
            rows[g].reg_off = total_cap;
            total_cap += rows[g].reg_cap;
            size_t p2 = 1;
            while (p2 < rows[g].reg_cap) p2 <<= 1;
            rows[g].is_big = p2 * key_bytes > kDedupMaxBytes;
            if (rows[g].is_big) any_big = true;
            else small_lds = std::max(small_lds, p2 * key_bytes);
        }
        if (cand.alloc(total_cap) != hipSuccess || sorted_out.alloc(total_cap * key_bytes) != hipSuccess)
            return rk_fail(ctx, RK_ERR_NOMEM, "cannot allocate %llu candidate slots", (unsigned long long)total_cap);
        // the CSR is allocated for the capacity (distinct hashes <= candidates), so that its placement needs no
        // count on the host: the whole pass runs without an intermediate synchronisation
        if (wide) { rk_pool_free(ctx, s->d_hashes64); s->d_hashes64 = nullptr; RK_TRY(pool_array(ctx, &s->d_hashes64, total_cap + 1)); }
        else { rk_pool_free(ctx, s->d_hashes); s->d_hashes = nullptr; RK_TRY(pool_array(ctx, &s->d_hashes, total_cap + 1)); }
        memcpy(pinned, rows.data(), rows_bytes);
        RK_HIP(ctx, hipMemcpyAsync(d_rows.p, pinned, rows_bytes, hipMemcpyHostToDevice, stream));
        RK_HIP(ctx, hipMemsetAsync(d_res.p, 0, work_bytes, stream));
        if (n_chunks) {
            SketchArgs a;
            a.packed = packed_dev;
            a.rows = d_rows.p;
            a.n_genomes = n_genomes;
            a.n_chunks = n_chunks;
            a.chunk_blocks = (uint32_t)cb;
            a.filter_image = f->d_bitmap;
            a.table = f->d_table;
            a.tupmask = P.tupmask;
            a.dim_bits = 4 * P.half_subk;
            a.hi_shift = std::max(0, 4 * P.half_subk - (f->img ? Img<1>::kBitsB : Img<0>::kBitsB));  // unused by the two-stage scan
            a.undomask0 = P.undomask0;
            a.undomask1 = P.undomask1;
            a.kmer = (int32_t)P.kmer_size;
            a.out2 = 2 * P.half_outctx_len;
            a.dim_start = P.dim_start;
            a.dim_end = P.dim_end;
            a.dr_shift = 4 * P.drlevel;
            a.und1_shift = (int32_t)P.kmer_size * 2 - P.half_outctx_len * 4;
            a.cand = cand.p;
            a.gcount = d_gcount;
            a.n_windows = &d_tail->windows;
            const uint32_t want = (n_chunks + kWavesPerBlock - 1) / kWavesPerBlock;
            // persistent: as many workgroups as the LDS image admits per CU (two; one with the 144 KiB image), the filter
            // image staged once per workgroup; chunks by table and ticket (chunk_range)
            uint32_t grid = std::min<uint32_t>(want, (uint32_t)ctx->num_cu * (f->img ? 2u : 1u));
            RK_HIP(ctx, d_chunks.alloc(n_chunks));
            hipLaunchKernelGGL(k_chunk_table, dim3(blocks_for(n_chunks)), dim3(256), 0, stream, d_rows.p, n_genomes, n_chunks, (uint32_t)cb, d_chunks.p);
            a.chunks = d_chunks.p;
            a.n_groups = std::max(1u, std::min(grid / 16, kTicketGroups));
            if (a.n_groups > 1) grid -= grid % (8 * a.n_groups);
            a.tickets = reinterpret_cast<uint32_t *>(d_res.p + tickets_off);
            a.prio_turns = getenv("RK_SCAN2_PRIO") ? (uint32_t)atoi(getenv("RK_SCAN2_PRIO")) : 2u;
            a.trace = nullptr;
            DevBuf<unsigned long long> d_trace(ctx);
            const char *trace_path = getenv("RK_SCAN2_TRACE");
            if (trace_path && f->img == 2) {
                RK_HIP(ctx, d_trace.alloc((size_t)grid * kWavesPerBlock * 8));
                RK_HIP(ctx, hipMemsetAsync(d_trace.p, 0, (size_t)grid * kWavesPerBlock * 64, stream));
                a.trace = d_trace.p;
            }
            if (!kern) return rk_fail(ctx, RK_ERR_UNSUPPORTED, "no scan kernel for this parameter set");
            if (ctx->timing) RK_HIP(ctx, hipEventRecord(ctx->ev[0], stream));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kSketchThreads), 0, stream, a);
            if (ctx->timing) RK_HIP(ctx, hipEventRecord(ctx->ev[1], stream));
            RK_HIP(ctx, hipGetLastError());
            if (a.trace) {
                std::vector<unsigned long long> h((size_t)grid * kWavesPerBlock * 8);
                RK_HIP(ctx, hipMemcpyAsync(h.data(), a.trace, h.size() * 8, hipMemcpyDeviceToHost, stream));
                RK_HIP(ctx, hipStreamSynchronize(stream));
                if (FILE *fp = fopen(trace_path, "wb")) { fwrite(h.data(), 8, h.size(), fp); fclose(fp); }
            }
        }
        if (n_genomes) {
            // ---- per-genome dedup in LDS, sizes -> offsets, CSR placement: three launches, no host round trip
            if (wide) {
                if (small_lds > 48 * 1024)
                    RK_HIP(ctx, hipFuncSetAttribute((const void *)k_dedup<uint64_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds));
                hipLaunchKernelGGL(k_dedup<uint64_t>, dim3(n_genomes), dim3(kDedupThreads), small_lds, stream, cand.p, d_rows.p,
                                   d_gcount, min_count, (uint64_t *)sorted_out.p, d_usize.p, d_tail);
            } else {
                if (small_lds > 48 * 1024)
                    RK_HIP(ctx, hipFuncSetAttribute((const void *)k_dedup<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds));
                hipLaunchKernelGGL(k_dedup<uint32_t>, dim3(n_genomes), dim3(kDedupThreads), small_lds, stream, cand.p, d_rows.p,
                                   d_gcount, min_count, (uint32_t *)sorted_out.p, d_usize.p, d_tail);
            }
            RK_HIP(ctx, hipGetLastError());
        }
        bool big_overflow = false;
        if (any_big) {  // 3 Gb genomes: their regions exceed the LDS sort; needs their candidate counts on the host
            RK_HIP(ctx, hipMemcpyAsync(gcount.data(), d_gcount, (size_t)n_genomes * 4, hipMemcpyDeviceToHost, stream));
            RK_HIP(ctx, hipStreamSynchronize(stream));
            for (uint32_t g = 0; g < n_genomes && !big_overflow; g++) big_overflow = rows[g].is_big && gcount[g] > rows[g].reg_cap;
            for (uint32_t g = 0; g < n_genomes && !big_overflow; g++) {
                if (!rows[g].is_big) continue;
                uint32_t nu = 0;
                if (wide)
                    RK_TRY(dedup_big<uint64_t>(ctx, cand.p + rows[g].reg_off, gcount[g], hash_bits, min_count,
                                               (uint64_t *)sorted_out.p + rows[g].reg_off, &nu, stream));
                else
                    RK_TRY(dedup_big<uint32_t>(ctx, cand.p + rows[g].reg_off, gcount[g], hash_bits, min_count,
                                               (uint32_t *)sorted_out.p + rows[g].reg_off, &nu, stream));
                memcpy(pinned, &nu, 4);
                RK_HIP(ctx, hipMemcpyAsync(d_usize.p + g, pinned, 4, hipMemcpyHostToDevice, stream));
                RK_HIP(ctx, hipStreamSynchronize(stream));  // the pinned word is reused
            }
        }
        hipLaunchKernelGGL(k_size_scan, dim3(1), dim3(1024), 0, stream, d_usize.p, n_genomes, s->d_off, d_off_copy);
        if (n_genomes) {
            if (wide)
                hipLaunchKernelGGL(k_csr_place<uint64_t>, dim3(n_genomes), dim3(256), 0, stream, (const uint64_t *)sorted_out.p,
                                   d_rows.p, d_usize.p, s->d_off, s->d_hashes64);
            else
                hipLaunchKernelGGL(k_csr_place<uint32_t>, dim3(n_genomes), dim3(256), 0, stream, (const uint32_t *)sorted_out.p,
                                   d_rows.p, d_usize.p, s->d_off, s->d_hashes);
        }
        RK_HIP(ctx, hipGetLastError());
        // ---- the one read-back of the common case: window count, offsets, candidate counts (for the retry)
        RK_HIP(ctx, hipMemcpyAsync(pinned, d_res.p, res_bytes, hipMemcpyDeviceToHost, stream));
        RK_HIP(ctx, hipStreamSynchronize(stream));
        memcpy(&tail, pinned, sizeof(tail));
        memcpy(s->h_off.data(), pinned + sizeof(SketchTail), ((size_t)n_genomes + 1) * 8);
        if (n_genomes) memcpy(gcount.data(), pinned + sizeof(SketchTail) + ((size_t)n_genomes + 1) * 8, (size_t)n_genomes * 4);
        bool overflow = big_overflow || (tail.flags & kFlagOverflow) != 0;
        for (uint32_t g = 0; g < n_genomes; g++)
            if (gcount[g] > rows[g].reg_cap) {
                overflow = true;
                rows[g].reg_cap = gcount[g];  // exact on the second pass
            }
        if (!overflow) break;
        if (attempt == 1) return rk_fail(ctx, RK_ERR_CAPACITY, "candidate overflow persisted");
    }
    if (ctx->timing && n_chunks) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]) == hipSuccess) ctx->last_ms[RK_MS_SKETCH_KERNEL] = ms;
    }
    s->windows = tail.windows;
    s->total = s->h_off[n_genomes];
    for (uint32_t g = 0; g < n_genomes; g++) s->max_size = std::max<uint64_t>(s->max_size, s->h_off[g + 1] - s->h_off[g]);
    guard.p = nullptr;
    *out = s;
    return RK_OK;
}

int rk_sketch_batch(rk_ctx *ctx, const rk_filter *f, const uint8_t *seq, const uint64_t *rec_off,
                    uint64_t n_rec, const uint64_t *genome_rec, uint32_t n_genomes, rk_sketches **out)
{
    return rk_sketch_batch_ex(ctx, f, seq, nullptr, 0, 1, rec_off, n_rec, genome_rec, n_genomes, out);
}

int rk_sketch_batch_ex(rk_ctx *ctx, const rk_filter *f, const uint8_t *seq, const uint8_t *qual, int least_qual,
                       uint32_t min_count, const uint64_t *rec_off, uint64_t n_rec, const uint64_t *genome_rec,
                       uint32_t n_genomes, rk_sketches **out)
{
    if (!ctx || !f || !rec_off || !genome_rec || !out || (!seq && rec_off[n_rec])) return RK_ERR_ARG;
    *out = nullptr;
    std::vector<uint64_t> gbeg((size_t)n_genomes + 1), gend((size_t)n_genomes + 1);
    uint64_t bytes = 0;
    int rc = rk_pack_layout(rec_off, n_rec, genome_rec, n_genomes, gbeg.data(), gend.data(), &bytes);
    if (rc) return rk_fail(ctx, rc, "bad record/genome offsets");
    RK_HIP(ctx, hipSetDevice(ctx->device));
    uint8_t *h_packed = nullptr;
    RK_HIP(ctx, hipHostMalloc((void **)&h_packed, bytes, hipHostMallocDefault));
    rc = rk_pack_genomes(seq, rec_off, n_rec, genome_rec, n_genomes, gbeg.data(), h_packed, bytes);
    if (!rc && qual) {
        // FASTQ quality gate (src/sketch.cpp:785): a base whose quality character is below
        // least_qual is not a base.  Applied while packing: the byte becomes 0x00, which the
        // kernel treats exactly like any other invalid base.
        for (uint32_t g = 0; g < n_genomes; g++) {
            uint64_t pos = gbeg[g];
            for (uint64_t r = genome_rec[g]; r < genome_rec[g + 1]; r++) {
                const uint64_t len = rec_off[r + 1] - rec_off[r];
                const uint8_t *q = qual + rec_off[r];
                for (uint64_t i = 0; i < len; i++)
                    if ((int)(char)q[i] < least_qual) h_packed[pos + i] = 0;
                pos += len + 1;
            }
        }
    }
    DevBuf<uint8_t> d_packed(ctx);
    if (!rc && d_packed.alloc(bytes) != hipSuccess) rc = rk_fail(ctx, RK_ERR_NOMEM, "device alloc of %llu bytes failed", (unsigned long long)bytes);
    if (!rc && (hipMemcpyAsync(d_packed.p, h_packed, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess))
        rc = rk_fail(ctx, RK_ERR_HIP, "sequence upload failed");
    (void)hipHostFree(h_packed);
    if (rc) return rc;
    return rk_sketch_packed_dev_ex(ctx, f, d_packed.p, bytes, gbeg.data(), gend.data(), n_genomes,
                                   min_count ? min_count : 1, ctx->stream, out);
}

}  // extern "C"

static int check_sets(rk_ctx *ctx, rk_sketches *s, bool *ascending)
{
    DevBuf<uint32_t> bad(ctx);
    RK_HIP(ctx, bad.alloc(1));
    RK_HIP(ctx, hipMemsetAsync(bad.p, 0, 4, ctx->stream));
    if (s->n && s->total > 1) {
        const unsigned blocks = blocks_for(s->total - 1);
        if (s->wide)
            hipLaunchKernelGGL(k_check_sets<uint64_t>, dim3(blocks), dim3(256), 0, ctx->stream, s->d_hashes64, s->total, s->d_off, s->n, bad.p);
        else
            hipLaunchKernelGGL(k_check_sets<uint32_t>, dim3(blocks), dim3(256), 0, ctx->stream, s->d_hashes, s->total, s->d_off, s->n, bad.p);
        RK_HIP(ctx, hipGetLastError());
    }
    uint32_t b = 0;
    RK_TRY(rk_read_back(ctx, &b, bad.p, 4, ctx->stream));
    *ascending = b == 0;
    return RK_OK;
}

// Sketches that crossed the boundary from the host (or from caller-owned device arrays).  A `.sketch` written by the
// reference lists every genome's hashes in `unordered_set` iteration order (src/sketch.cpp:537-553: the sort is commented
// out): such sketches are SORTED here, on the device, genome by genome (one segmented radix sort; every consumer treats a
// sketch as a set, and rk_sketches_download documents the ascending order) -- they then are sets in ascending order like the
// sketcher's own output: an intersection count is bounded by the smaller sketch (narrow LDS counters, rk_distq.hip), the
// index build takes its fast path and rows pair up.  Only a sketch that repeats a hash (a malformed or foreign file;
// the reference would count the repeats, src/dist.cpp:199-202) stays a multiset: is_set = false, 32-bit counters, general
// index build.  Synchronises ctx->stream.
int rk_sketches_classify(rk_ctx *ctx, rk_sketches *s)
{
    s->max_size = 0;
    for (uint32_t g = 0; g < s->n; g++) s->max_size = std::max<uint64_t>(s->max_size, s->h_off[g + 1] - s->h_off[g]);
    bool ascending = true;
    RK_TRY(check_sets(ctx, s, &ascending));
    if (!ascending && s->total < 0xFFFFFFFFULL) {
        const unsigned int n = (unsigned int)s->total;
        if (s->wide) {
            DevBuf<uint64_t> sorted(ctx);
            RK_HIP(ctx, sorted.alloc(s->total + 1));
            RK_TRY(rk_prim_segmented_sort_u64(ctx, s->d_hashes64, sorted.p, n, s->n, s->d_off, ctx->stream));
            RK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            rk_pool_free(ctx, s->d_hashes64);
            s->d_hashes64 = sorted.release();
        } else {
            DevBuf<uint32_t> sorted(ctx);
            RK_HIP(ctx, sorted.alloc(s->total + 1));
            RK_TRY(rk_prim_segmented_sort_u32(ctx, s->d_hashes, sorted.p, n, s->n, s->d_off, ctx->stream));
            RK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            rk_pool_free(ctx, s->d_hashes);
            s->d_hashes = sorted.release();
        }
        RK_TRY(check_sets(ctx, s, &ascending));  // still not strictly ascending: some genome repeats a hash
    }
    s->is_set = ascending;
    return RK_OK;
}
