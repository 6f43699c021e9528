// rabbit_kssd.cpp -- host tool with RabbitKSSD's command line (src/main.cpp:34-145) for the
// hot-path subcommands, built on the C ABI of librabbitkssd.so (include/rabbitkssd.h).
//
//   shuffle  -k -s -l -o                     (host only, src/shuffle.cpp:25-104)
//   sketch   -i list -o out [-L shuf] [-q] [-Q q -n c]   (GPU: rk_sketch_batch_ex [+ rk_index_build];
//                                             FASTA or FASTQ lists, plain or .gz)
//   alldist  -i sketch|list -o out [-D -M -L]            (GPU: rk_index_build, rk_dist_rows)
//   dist     -r ref -q qry -o out [-D -M -N -L]          (GPU: rk_dist_rows [+ rk_topn_rows])
//   info     -i sketch -o out [-F]           (host only, src/subCommand.cpp:70-147)
//   merge    -i list -o out                  (host only, src/subCommand.cpp:796-892)
//   union / sub                              (host only, src/subCommand.cpp:307-794)
//   convert [--reverse]                      (host formats, src/sketch.cpp:1179-1365; index on GPU)
// Errors follow the reference: a message on stderr and exit(1).  There is no CPU fallback:
// without a GPU the GPU subcommands fail at rk_ctx_create.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <deque>
#include <cstdarg>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include "formats.hpp"
#include "rabbitkssd.h"

using namespace rkhost;
using std::cerr;
using std::endl;
using std::string;
using std::vector;

static double get_sec()
{
    struct timeval tv;
    gettimeofday(&tv, nullptr);
    return (double)tv.tv_sec + (double)tv.tv_usec / 1e6;
}

[[noreturn]] static void die(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    fprintf(stderr, "ERROR: ");
    vfprintf(stderr, fmt, ap);
    fprintf(stderr, "\n");
    va_end(ap);
    exit(1);
}

static const double g_t_start = get_sec();
static void stamp(const char *what)
{
    static const bool on = getenv("RK_TIMING") != nullptr;
    if (on) fprintf(stderr, "[timing] %8.3f ms  %s\n", (get_sec() - g_t_start) * 1e3, what);
}

struct Gpu {
    rk_ctx *ctx = nullptr;
    explicit Gpu(int device)
    {
        stamp("before rk_ctx_create");
        int rc = rk_ctx_create(device, &ctx);
        if (rc) die("no usable GPU (rk_ctx_create(%d) = %d): this build has no CPU path", device, rc);
        rk_ctx_set_single_shot(ctx, 1);   // one pass per process: nothing amortises a one-time device-side setup
        stamp("context ready");
    }
    // The process ends right after its last GPU call (main leaves through _exit): returning ~600 MB of pooled device
    // memory block by block and tearing the HIP runtime down costs 60-100 ms that buy nothing.
    ~Gpu() {}
    void check(int rc, const char *what) const
    {
        if (rc) die("%s failed (%d): %s", what, rc, rk_last_error(ctx));
    }
};

// ---- one big plain FASTA file, streamed (the role of src/sketch.cpp:380-450: RabbitFX chunks of a big file go to the
// consumers while the producer is still reading) --------------------------------------------------------------------
// A 3 Gb genome used to be read completely, parsed in two passes into ordinary memory and only then uploaded
// (0.71 s + 0.09 s).  Now the file is mapped and cut into pieces at line starts; parser threads turn piece after piece
// into the packed layout -- ONE pass, straight into a small ring of page-locked buffers -- and the calling thread
// uploads every finished piece to its own slot of the device buffer while the others are still being parsed.  A piece
// starts with the last k-1 bases in front of it (unless a header line lies between), so that every window is seen
// exactly once: by the piece it ENDS in.  Slots are sized for the piece's file bytes (newlines become padding: zero
// bytes, which the kernel treats like any invalid base), so no piece has to know how much the pieces before it shrank.
// Returns false -- nothing done -- for inputs this path does not take (FASTQ, '\r' line ends, no header at the start):
// the caller falls back to the whole-file path.
static bool sketch_big_fasta_streamed(Gpu &gpu, const rk_filter *flt, const string &path, int kmer, int threads, uint32_t min_count,
                                      void *stream, rk_sketches **sk_out, bool timing)
{
    const double t0 = get_sec();
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) || st.st_size < 64) { close(fd); return false; }
    const size_t sz = (size_t)st.st_size;
    const uint8_t *b = (const uint8_t *)mmap(nullptr, sz, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (b == MAP_FAILED) return false;
    (void)madvise((void *)b, sz, MADV_SEQUENTIAL);
    struct Unmap { const uint8_t *p; size_t n; ~Unmap() { munmap((void *)p, n); } } unmap{b, sz};
    if (b[0] != '>') return false;   // (kseq skips to the first header: leave odd files to the serial reader)

    size_t piece = (size_t)(getenv("RK_BIG_PIECE_MB") ? std::max(1, atoi(getenv("RK_BIG_PIECE_MB"))) : 32) << 20;
    if (getenv("RK_BIG_PIECE_KB")) piece = (size_t)std::max(1, atoi(getenv("RK_BIG_PIECE_KB"))) << 10;   // tests
    vector<size_t> cut{0};
    while (cut.back() < sz) {
        size_t p = std::min(sz, cut.back() + piece);
        if (p < sz) {
            const void *e = memchr(b + p, '\n', sz - p);
            p = e ? (size_t)((const uint8_t *)e - b) + 1 : sz;
        }
        cut.push_back(p);
    }
    const size_t n_pieces = cut.size() - 1;
    vector<uint64_t> slot_off(n_pieces + 1, 0);
    uint64_t max_slot = 0;
    for (size_t i = 0; i < n_pieces; i++) {
        const uint64_t cap = ((cut[i + 1] - cut[i]) + (uint64_t)kmer + 2 + 1023) & ~1023ULL;   // prefix + separator + the piece
        slot_off[i + 1] = slot_off[i] + cap;
        max_slot = std::max(max_slot, cap);
    }
    const uint64_t dev_bytes = slot_off[n_pieces];
    const int n_ring = (int)std::min<size_t>(n_pieces, (size_t)std::max(2, std::min(threads, 8)));
    vector<uint8_t *> ring((size_t)n_ring, nullptr);
    void *d_buf = nullptr;
    gpu.check(rk_dev_alloc(gpu.ctx, dev_bytes, &d_buf), "rk_dev_alloc");
    for (int j = 0; j < n_ring; j++) gpu.check(rk_pinned_alloc(gpu.ctx, max_slot, (void **)&ring[(size_t)j]), "rk_pinned_alloc");
    if (timing) fprintf(stderr, "[timing] big file: %zu piece(s), %d x %.1f MB page-locked + %.1f MB device: %.3f s\n", n_pieces, n_ring,
                        max_slot / 1e6, dev_bytes / 1e6, get_sec() - t0);

    std::mutex mu;
    std::condition_variable cv;
    std::deque<int> free_slots;
    std::deque<std::pair<size_t, int>> ready;   // (piece, ring slot)
    for (int j = 0; j < n_ring; j++) free_slots.push_back(j);
    std::atomic<size_t> next_piece{0};
    std::atomic<bool> bad{false};

    // the last k-1 bases in front of position `pos` (a line start), unless a header line lies between: walked backwards
    // line by line.  Returns the number of bases written to the END of tmp[0 .. k-1).
    auto bases_before = [&](size_t pos, uint8_t *tmp, size_t need) -> size_t {
        size_t got = 0;
        size_t le = pos;   // one past the newline that ends the line being looked at
        while (got < need && le > 0) {
            const size_t nl = le - 1;   // b[nl] == '\n'
            const void *q = nl ? memrchr(b, '\n', nl) : nullptr;
            const size_t ls = q ? (size_t)((const uint8_t *)q - b) + 1 : 0;
            if (b[ls] == '>') return got;   // a record starts here: nothing in front of it belongs to the windows of this piece
            if (b[ls] == '+' || b[ls] == '@') { bad = true; return 0; }
            for (size_t c = nl; c > ls && got < need; c--) tmp[need - 1 - got++] = b[c - 1];
            le = ls;
        }
        return got;
    };
    auto parse_piece = [&](size_t i, uint8_t *dst) {
        const uint64_t cap = slot_off[i + 1] - slot_off[i];
        uint8_t *w = dst;
        if (i > 0) {
            uint8_t tmp[64];
            const size_t need = (size_t)std::min(63, kmer - 1);
            const size_t got = bases_before(cut[i], tmp, need);
            memcpy(w, tmp + (need - got), got);
            w += got;
        }
        size_t pos = cut[i];
        const size_t end = cut[i + 1];
        while (pos < end) {
            const void *q = memchr(b + pos, '\n', end - pos);
            const size_t e = q ? (size_t)((const uint8_t *)q - b) : end;
            const uint8_t c = b[pos];
            if (c == '>') {
                if (pos != 0) *w++ = 0;   // record separator (a window never spans records, src/sketch.cpp:487-488)
            } else if (c == '+' || c == '@') {
                bad = true;
                return;
            } else if (e > pos) {
                if (b[e - 1] == '\r') { bad = true; return; }
                memcpy(w, b + pos, e - pos);
                w += e - pos;
            }
            pos = e + 1;
        }
        memset(w, 0, (size_t)(dst + cap - w));   // padding: invalid bases
    };

    vector<std::thread> pool;
    const int nt = (int)std::min<size_t>(n_pieces, (size_t)std::max(1, threads));
    for (int t = 0; t < nt; t++)
        pool.emplace_back([&]() {
            for (;;) {
                const size_t i = next_piece.fetch_add(1);
                if (i >= n_pieces) return;
                int j;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !free_slots.empty(); });
                    j = free_slots.front();
                    free_slots.pop_front();
                }
                if (!bad) parse_piece(i, ring[(size_t)j]);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    ready.emplace_back(i, j);
                }
                cv.notify_all();
            }
        });
    // this thread: every finished piece to its slot of the device buffer, while the others are parsed
    for (size_t done = 0; done < n_pieces; done++) {
        std::pair<size_t, int> pj;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !ready.empty(); });
            pj = ready.front();
            ready.pop_front();
        }
        if (!bad) {
            gpu.check(rk_upload_async(gpu.ctx, (uint8_t *)d_buf + slot_off[pj.first], ring[(size_t)pj.second],
                                      slot_off[pj.first + 1] - slot_off[pj.first], stream), "rk_upload_async");
            gpu.check(rk_stream_sync(gpu.ctx, stream), "rk_stream_sync");
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            free_slots.push_back(pj.second);
        }
        cv.notify_all();
    }
    for (auto &th : pool) th.join();
    const double t1 = get_sec();
    bool ok = !bad;
    if (ok) {
        const uint64_t gbeg = 0, gend = dev_bytes;
        gpu.check(rk_sketch_packed_dev_ex(gpu.ctx, flt, (const uint8_t *)d_buf, dev_bytes, &gbeg, &gend, 1, min_count, stream, sk_out),
                  "rk_sketch_packed_dev_ex");
    }
    if (timing) fprintf(stderr, "[timing] big file: parse + upload of %.1f MB (overlapped, %d thread(s)): %.3f s, sketch kernels: %.3f s\n",
                        sz / 1e6, nt, t1 - t0, get_sec() - t1);
    for (int j = 0; j < n_ring; j++) rk_pinned_free(ring[(size_t)j]);
    rk_dev_free(d_buf);
    return ok;
}


// ---- one big gzip'ed file, or one big FASTQ file, streamed (the role of src/sketch.cpp:380-450 and :658-737: the
// RabbitFX producer hands chunks of FA / FQ_SE records to the consumers while it is still reading) -------------------
// Such a file used to be inflated and parsed completely before its upload.  Now a reader thread (gzread: plain, gzip'ed
// and multi-member files alike) fills text pieces of ~32 MB, each cut where the parser of a piece needs no context: a
// FASTQ piece ends in front of a record (an '@' line whose next line but one starts with '+'; RecordReader::parse_packed
// -- kseq semantics, quality gate -- parses it like a file of its own), a FASTA piece at a line start and carries the
// last k-1 bases of the text in front of it unless a header line lies between (as sketch_big_fasta_streamed).  Parser
// threads turn pieces into the packed layout in a ring of page-locked buffers, the calling thread uploads them to
// consecutive slots of a device buffer that doubles when the inflated size outgrows it, and ONE sketch call over the
// whole buffer follows (so -n, the minimum number of occurrences, counts over the whole file).
// Returns false -- nothing done -- for inputs this path does not take ('\r' line ends in FASTA, multi-line FASTQ records
// longer than a piece, a file that starts without a header): the caller falls back to the whole-file path.
static bool sketch_big_sequential(Gpu &gpu, const rk_filter *flt, const string &path, bool fastq, int least_qual, int kmer, int threads,
                                  uint32_t min_count, void *stream, rk_sketches **sk_out, bool timing)
{
    const double t0 = get_sec();
    gzFile fp = gzopen(path.c_str(), "r");
    if (!fp) return false;
    gzbuffer(fp, 1 << 20);
    size_t piece = (size_t)(getenv("RK_BIG_PIECE_MB") ? std::max(1, atoi(getenv("RK_BIG_PIECE_MB"))) : 32) << 20;
    if (getenv("RK_BIG_PIECE_KB")) piece = (size_t)std::max(1, atoi(getenv("RK_BIG_PIECE_KB"))) << 10;   // tests
    const size_t need = (size_t)std::min(63, kmer - 1);
    struct Piece {
        RawBuf text;
        size_t len = 0;
        uint8_t prefix[64];
        size_t n_prefix = 0;
        bool first = false;
        uint64_t dev_off = 0, cap = 0;
    };
    const int n_ring = std::max(2, std::min(threads, 4));
    const uint64_t slot_cap = ((uint64_t)piece * 2 + (uint64_t)kmer + 2 + 1024 + 1023) & ~1023ULL;   // (a piece holds at most 2 x `piece` text bytes)
    vector<uint8_t *> ring((size_t)n_ring, nullptr);
    for (int j = 0; j < n_ring; j++) gpu.check(rk_pinned_alloc(gpu.ctx, slot_cap, (void **)&ring[(size_t)j]), "rk_pinned_alloc");
    struct stat st;
    uint64_t dev_cap = 256ull << 20;
    if (stat(path.c_str(), &st) == 0) dev_cap = std::max<uint64_t>(dev_cap, (uint64_t)st.st_size * (ends_with(path, ".gz") ? 4 : 1) + (64ull << 20));
    if (getenv("RK_BIG_DEV_MB")) dev_cap = (uint64_t)std::max(1, atoi(getenv("RK_BIG_DEV_MB"))) << 20;   // tests: force the growth
    dev_cap = (dev_cap + 1023) & ~1023ULL;
    void *d_buf = nullptr;
    gpu.check(rk_dev_alloc(gpu.ctx, dev_cap, &d_buf), "rk_dev_alloc");

    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::unique_ptr<Piece>> todo;              // read, not yet parsed
    std::deque<std::pair<std::unique_ptr<Piece>, int>> ready;   // parsed into ring slot
    std::deque<int> free_slots;
    for (int j = 0; j < n_ring; j++) free_slots.push_back(j);
    bool eof = false;
    std::atomic<bool> bad{false};
    size_t n_pieces_read = 0, n_parsed = 0;
    uint64_t text_bytes = 0;

    // where a text may be cut so that what follows parses on its own: returns the length of the part to keep
    auto cut_point = [&](const uint8_t *b, size_t n) -> size_t {
        if (fastq) {
            // the last '@' line start whose next line but one starts with '+' (a quality line may start with '@' too: the
            // line after ITS next is a sequence line, never '+')
            size_t pos = n;
            for (int tries = 0; tries < 4096 && pos > 0; tries++) {
                const void *q = memrchr(b, '\n', pos - 1);
                const size_t ls = q ? (size_t)((const uint8_t *)q - b) + 1 : 0;
                if (b[ls] == '@') {
                    const void *e1 = memchr(b + ls, '\n', n - ls);
                    const void *e2 = e1 ? memchr((const uint8_t *)e1 + 1, '\n', n - ((const uint8_t *)e1 + 1 - b)) : nullptr;
                    if (e2 && (size_t)((const uint8_t *)e2 + 1 - b) < n && ((const uint8_t *)e2)[1] == '+') {
                        // ... and, so that a quality line that starts with '@' inside a MULTI-line record (two lines further on: another
                        // quality line starting with '+': both are valid Phred characters) is not taken for a record start: the
                        // candidate's sequence line and its quality line have the same length (a four-line record; a multi-line file
                        // offers no such candidate and takes the whole-file path)
                        const uint8_t *p3 = (const uint8_t *)e2 + 1;
                        const void *e3 = memchr(p3, '\n', n - (size_t)(p3 - b));
                        const uint8_t *p4 = e3 ? (const uint8_t *)e3 + 1 : nullptr;
                        const void *e4 = p4 ? memchr(p4, '\n', n - (size_t)(p4 - b)) : nullptr;
                        if (e4) {
                            auto len = [](const uint8_t *a, const void *e) { size_t l = (size_t)((const uint8_t *)e - a); return l && a[l - 1] == '\r' ? l - 1 : l; };
                            if (len((const uint8_t *)e1 + 1, e2) == len(p4, e4)) return ls;
                        }
                    }
                }
                if (ls == 0) break;
                pos = ls;   // (the newline before this line start is at ls - 1)
                if (pos == 0) break;
            }
            return 0;
        }
        const void *q = memrchr(b, '\n', n);
        return q ? (size_t)((const uint8_t *)q - b) + 1 : 0;
    };
    // FASTA: the last k-1 bases of a text that ends at a line start, unless a header line lies between
    auto tail_bases = [&](const uint8_t *b, size_t n, const uint8_t *older, size_t n_older, uint8_t *out) -> size_t {
        uint8_t tmp[64];
        size_t got = 0;
        auto walk = [&](const uint8_t *t, size_t len) -> bool {   // false: stop (a header line, or enough)
            size_t le = len;
            while (got < need && le > 0) {
                const size_t nl = le - 1;
                const void *q = nl ? memrchr(t, '\n', nl) : nullptr;
                const size_t ls = q ? (size_t)((const uint8_t *)q - t) + 1 : 0;
                if (t[ls] == '>') return false;
                for (size_t c = nl; c > ls && got < need; c--) tmp[need - 1 - got++] = t[c - 1];
                le = ls;
            }
            return got < need;
        };
        if (walk(b, n) && older)   // (`older`: the bases in front of this text, no line structure)
            for (size_t c = n_older; c > 0 && got < need; c--) tmp[need - 1 - got++] = older[c - 1];
        memcpy(out, tmp + (need - got), got);
        return got;
    };

    std::thread reader([&]() {
        std::unique_ptr<Piece> cur(new Piece);
        cur->text.resize(2 * piece + (1u << 20));
        cur->first = true;
        uint8_t carry_prefix[64];
        size_t n_carry_prefix = 0;
        for (;;) {
            // fill the current piece up to `piece` bytes beyond what it already holds
            bool end = false;
            while (cur->len < piece) {
                const int r = gzread(fp, cur->text.data() + cur->len, (unsigned)std::min<size_t>(piece - cur->len, 1u << 30));
                if (r <= 0) { end = true; break; }
                cur->len += (size_t)r;
            }
            if (cur->first && cur->len && cur->text.data()[0] != (fastq ? '@' : '>')) { bad = true; end = true; cur->len = 0; }
            std::unique_ptr<Piece> next(new Piece);
            if (!end) {
                const size_t keep = cut_point(cur->text.data(), cur->len);
                if (keep == 0) { bad = true; end = true; cur->len = 0; }   // no cut point inside a piece: not a file for this path
                else {
                    next->text.resize(2 * piece + (1u << 20));
                    next->len = cur->len - keep;
                    memcpy(next->text.data(), cur->text.data() + keep, next->len);
                    cur->len = keep;
                }
            }
            if (!fastq) {
                memcpy(cur->prefix, carry_prefix, n_carry_prefix);
                cur->n_prefix = cur->first ? 0 : n_carry_prefix;
                // (a piece with fewer than k-1 bases of its own hands the bases in front of it on)
                if (cur->len) n_carry_prefix = tail_bases(cur->text.data(), cur->len, cur->prefix, cur->n_prefix, carry_prefix);
            }
            if (cur->len) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return todo.size() < (size_t)n_ring || bad; });   // (bounded: the reader runs at most n_ring pieces ahead)
                text_bytes += cur->len;
                n_pieces_read++;
                todo.push_back(std::move(cur));
                lk.unlock();
                cv.notify_all();
            }
            if (end) break;
            cur = std::move(next);
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            eof = true;
        }
        cv.notify_all();
    });

    auto parse_piece = [&](Piece &pc, uint8_t *dst) -> uint64_t {   // returns the bytes written (padded to 1 KiB by the caller)
        uint8_t *w = dst;
        const uint8_t *b = pc.text.data();
        if (fastq) {
            const RecordReader::Packed pk = RecordReader::parse_packed(b, pc.len, w, slot_cap - 1024, least_qual);
            if (pk.overflow) { bad = true; return 0; }
            w += pk.bytes;
            *w++ = 0;   // the next piece starts a new record
            return (uint64_t)(w - dst);
        }
        memcpy(w, pc.prefix, pc.n_prefix);
        w += pc.n_prefix;
        size_t pos = 0;
        while (pos < pc.len) {
            const void *q = memchr(b + pos, '\n', pc.len - pos);
            const size_t e = q ? (size_t)((const uint8_t *)q - b) : pc.len;
            const uint8_t c = b[pos];
            if (c == '>') {
                if (!(pc.first && pos == 0)) *w++ = 0;   // record separator (a window never spans records, src/sketch.cpp:487-488)
            } else if (c == '+' || c == '@') {
                bad = true;
                return 0;
            } else if (e > pos) {
                if (b[e - 1] == '\r') { bad = true; return 0; }
                memcpy(w, b + pos, e - pos);
                w += e - pos;
            }
            pos = e + 1;
        }
        return (uint64_t)(w - dst);
    };
    vector<std::thread> pool;
    const int nt = std::max(1, std::min(threads, n_ring));
    for (int t = 0; t < nt; t++)
        pool.emplace_back([&]() {
            for (;;) {
                std::unique_ptr<Piece> pc;
                int j;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return (!todo.empty() && !free_slots.empty()) || (eof && todo.empty()); });
                    if (todo.empty()) return;
                    pc = std::move(todo.front());
                    todo.pop_front();
                    j = free_slots.front();
                    free_slots.pop_front();
                }
                cv.notify_all();
                uint64_t bytes = bad ? 0 : parse_piece(*pc, ring[(size_t)j]);
                const uint64_t cap = (bytes + 1023) & ~1023ULL;
                memset(ring[(size_t)j] + bytes, 0, (size_t)(cap - bytes));   // padding: invalid bases
                pc->cap = cap;
                pc->text.resize(0);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    ready.emplace_back(std::move(pc), j);
                }
                cv.notify_all();
            }
        });
    // this thread: every parsed piece to the next free stretch of the device buffer (the order of the pieces in the buffer
    // does not matter to a FASTQ file -- every record is a record --, but it does to FASTA: a piece carries the bases in
    // front of it, so pieces may go in any order as long as each is followed by padding, which they are)
    uint64_t dev_used = 0;
    for (;;) {
        std::pair<std::unique_ptr<Piece>, int> pj;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !ready.empty() || (eof && todo.empty() && n_parsed == n_pieces_read); });
            if (ready.empty()) break;
            pj = std::move(ready.front());
            ready.pop_front();
            n_parsed++;
        }
        if (!bad && pj.first->cap) {
            if (dev_used + pj.first->cap > dev_cap) {   // the inflated file outgrew the buffer: twice the size, copy, go on
                const uint64_t bigger = std::max(dev_cap * 2, dev_used + pj.first->cap);
                void *d_new = nullptr;
                gpu.check(rk_dev_alloc(gpu.ctx, bigger, &d_new), "rk_dev_alloc");
                gpu.check(rk_dev_copy_async(gpu.ctx, d_new, d_buf, dev_used, stream), "rk_dev_copy_async");
                gpu.check(rk_stream_sync(gpu.ctx, stream), "rk_stream_sync");
                rk_dev_free(d_buf);
                d_buf = d_new;
                dev_cap = bigger;
            }
            gpu.check(rk_upload_async(gpu.ctx, (uint8_t *)d_buf + dev_used, ring[(size_t)pj.second], pj.first->cap, stream), "rk_upload_async");
            gpu.check(rk_stream_sync(gpu.ctx, stream), "rk_stream_sync");
            dev_used += pj.first->cap;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            free_slots.push_back(pj.second);
        }
        cv.notify_all();
    }
    reader.join();
    for (auto &th : pool) th.join();
    gzclose(fp);
    const double t1 = get_sec();
    bool ok = !bad;
    if (ok) {
        const uint64_t gbeg = 0, gend = dev_used;
        gpu.check(rk_sketch_packed_dev_ex(gpu.ctx, flt, (const uint8_t *)d_buf, dev_used, &gbeg, &gend, 1, min_count, stream, sk_out),
                  "rk_sketch_packed_dev_ex");
    }
    if (timing) fprintf(stderr, "[timing] big file (sequential reader): %zu piece(s), %.1f MB of text, %.1f MB packed, read + parse + upload "
                        "(overlapped, %d parser thread(s)): %.3f s, sketch kernels: %.3f s\n", n_pieces_read, text_bytes / 1e6, dev_used / 1e6, nt,
                        t1 - t0, get_sec() - t1);
    for (int j = 0; j < n_ring; j++) rk_pinned_free(ring[(size_t)j]);
    rk_dev_free(d_buf);
    return ok;
}


// HIP runtime start-up (hipInit: 150-190 ms on an MI355X box) is the largest single item of a short alldist/dist run:
// it starts on a thread of its own while the main thread reads the .sketch files
struct AsyncGpu {
    std::unique_ptr<Gpu> g;
    std::thread th;
    explicit AsyncGpu(int device) : th([this, device] { g.reset(new Gpu(device)); }) {}
    Gpu &get()
    {
        if (th.joinable()) th.join();
        return *g;
    }
    ~AsyncGpu() { if (th.joinable()) th.join(); }
};

// ---- tiny option parser -----------------------------------------------------------------
struct Args {
    std::map<string, string> kv;
    std::map<string, bool> seen;
    bool has(const string &k) const { return seen.count(k) != 0; }
    string str(const string &k, const string &d) const { auto it = kv.find(k); return it == kv.end() ? d : it->second; }
    int num(const string &k, int d) const { auto it = kv.find(k); return it == kv.end() ? d : atoi(it->second.c_str()); }
    double real(const string &k, double d) const { auto it = kv.find(k); return it == kv.end() ? d : atof(it->second.c_str()); }
};

// spec: {"-k","--halfk"} -> canonical "k"; flags take no value
static Args parse_args(int argc, char **argv, int first, const std::map<string, string> &alias,
                       const std::vector<string> &flags)
{
    Args a;
    for (int i = first; i < argc; i++) {
        string t = argv[i], val;
        bool has_val = false;
        auto eq = t.find('=');
        if (t.rfind("--", 0) == 0 && eq != string::npos) { val = t.substr(eq + 1); t = t.substr(0, eq); has_val = true; }
        auto it = alias.find(t);
        if (it == alias.end()) die("unknown option %s", argv[i]);
        const string key = it->second;
        a.seen[key] = true;
        if (std::find(flags.begin(), flags.end(), key) != flags.end()) { a.kv[key] = "1"; continue; }
        if (!has_val) {
            if (i + 1 >= argc) die("option %s needs a value", argv[i]);
            val = argv[++i];
        }
        a.kv[key] = val;
    }
    return a;
}

// ---- sketching --------------------------------------------------------------------------
// Replaces sketchFastaFile (src/sketch.cpp:318-593): genomes keep list order and hashes are
// sorted (the reference's order is unordered_set / OpenMP completion order; every consumer
// treats them as sets).
struct FastqOpts {
    bool fastq = false;   // the list holds FASTQ files (sketchFastqFile, src/sketch.cpp:596-890)
    int least_qual = 0;   // -Q
    int least_num = 1;    // -n
};

// '>' -> FASTA list, '@' -> FASTQ list, anything else / mixed -> error (src/sketch.cpp:52-94,
// src/subCommand.cpp:54-62).  *.gz names are accepted by suffix like isFastaGZList/isFastqGZList.
static bool list_is_fastq(const vector<string> &files)
{
    int n_fa = 0, n_fq = 0;
    for (const string &f : files) {
        int c;
        if (ends_with(f, ".gz")) c = (ends_with(f, ".fq.gz") || ends_with(f, ".fastq.gz")) ? '@' : '>';
        else c = first_byte(f);
        if (c == '>') n_fa++;
        else if (c == '@') n_fq++;
        else die("the input file list for sketching must be list of fasta and fastq file in normal format or gz format");
    }
    if (n_fa && n_fq) die("the input file list for sketching must be list of fasta and fastq file in normal format or gz format");
    return n_fq > 0;
}

static void sketch_list(Gpu &gpu, const string &list, bool is_query, const Shuf &shuf, int threads,
                        SketchSet &out, const string &out_path_in, const FastqOpts &fq = FastqOpts())
{
    const double t0 = get_sec();
    rk_params P;
    if (rk_params_init(shuf.k, shuf.subk, shuf.drlevel, &P))
        die("the half_subk - drlevel should at least 3 (half_subk=%d drlevel=%d), half_k >= half_subk, half_subk < 8",
            shuf.subk, shuf.drlevel);  // src/common.cpp:37
    rk_filter *flt = nullptr;
    gpu.check(rk_filter_create(gpu.ctx, &P, shuf.table.data(), &flt), "rk_filter_create");
    if (getenv("RK_TIMING")) fprintf(stderr, "[timing] rk_filter_create: %.3f s\n", get_sec() - t0);

    const vector<string> files = read_list(list);
    cerr << "the total fileNumber is: " << files.size() << endl;
    out = SketchSet();
    out.info.half_k = shuf.k;
    out.info.half_subk = shuf.subk;
    out.info.drlevel = shuf.drlevel;
    out.names = files;
    out.off.assign(1, 0);

    // ---- streaming pipeline (the role of src/sketch.cpp:318-460's producer/consumer threads) ----
    // Files are grouped into batches that fit one page-locked staging buffer.  The parser
    // threads write every genome straight into its 1 KiB-aligned slot of the staging buffer
    // in the packed layout of rk_sketch_packed_dev (no intermediate copy); a GPU thread
    // uploads the batch, runs the sketch kernels and downloads the hashes while the parser
    // threads already fill the other staging buffer.
    struct Slot {
        uint64_t off = 0, cap = 0;  // position / capacity in the staging buffer
        uint64_t len = 0;           // packed bytes written
        bool overflow = false;      // the size estimate was too small (multi-member .gz): slow path
    };
    struct Batch {
        size_t first = 0, last = 0;  // files [first, last)
        uint64_t bytes = 0;          // staging bytes used
        bool pageable = false;       // one oversized file: staged in ordinary memory
        bool streamed = false;       // ... or, a plain FASTA file, streamed piece by piece (sketch_big_fasta_streamed)
        vector<Slot> slots;
    };
    // upper bound of a file's packed size: a plain file cannot expand; a .gz member stores its
    // length mod 2^32 in the trailer
    auto packed_bound = [](const string &f) -> uint64_t {
        struct stat st;
        if (stat(f.c_str(), &st)) die("cannot open the genome file: %s", f.c_str());
        uint64_t sz = (uint64_t)st.st_size;
        FILE *fp = fopen(f.c_str(), "rb");
        if (!fp) die("cannot open the genome file: %s", f.c_str());
        unsigned char m[2] = {0, 0}, tail[4];
        if (fread(m, 1, 2, fp) == 2 && m[0] == 0x1f && m[1] == 0x8b && sz >= 18 && !fseek(fp, -4, SEEK_END) &&
            fread(tail, 1, 4, fp) == 4) {
            const uint64_t isize = (uint64_t)tail[0] | ((uint64_t)tail[1] << 8) | ((uint64_t)tail[2] << 16) |
                                   ((uint64_t)tail[3] << 24);
            sz = std::max<uint64_t>(isize, sz);
        }
        fclose(fp);
        return sz;
    };
    vector<uint64_t> bound(files.size());
    uint64_t total_bound = 0, max_bound = 0;
    for (size_t i = 0; i < files.size(); i++) {
        bound[i] = ((packed_bound(files[i]) + 1 + 1023) & ~1023ULL);
        total_bound += bound[i];
        max_bound = std::max(max_bound, bound[i]);
    }
    // staging buffers: two, an eighth of the input each (uploads and kernels overlap the parsing of the next batch), between
    // 64 and 256 MiB -- page-locking costs 0.2 s per GB and releasing 0.13 s per GB (measured: 2 x 1 GiB for 1,000 x 5 Mb
    // were 0.42 + 0.27 s of a 1.4 s run whose parsing, uploads and kernels take 0.2 s).  A file bigger than the staging
    // buffer gets a batch of its own: a plain FASTA file is streamed in pieces through a ring of small page-locked buffers
    // (sketch_big_fasta_streamed), anything else is parsed into ordinary memory (page-locking and releasing 3 GB costs 0.9 s,
    // the slower upload 0.2 s).
    uint64_t stage_bytes = std::min<uint64_t>(256ull << 20, std::max<uint64_t>(64ull << 20, total_bound / 8));
    if (getenv("RK_STAGE_MB")) stage_bytes = std::max<uint64_t>(1, (uint64_t)atoll(getenv("RK_STAGE_MB"))) << 20;
    stage_bytes = (stage_bytes + 1023) & ~1023ULL;
    const uint64_t dev_bytes = std::max(stage_bytes, max_bound);
    vector<Batch> batches;
    for (size_t i = 0; i < files.size();) {
        Batch bt;
        bt.first = i;
        uint64_t pos = 0;
        if (bound[i] > stage_bytes) {  // oversized: alone, pageable staging
            Slot sl;
            sl.cap = bound[i];
            bt.slots.push_back(sl);
            bt.last = i + 1;
            bt.bytes = bound[i];
            bt.pageable = true;
            batches.push_back(std::move(bt));
            i++;
            continue;
        }
        while (i < files.size() && pos + bound[i] <= stage_bytes) {
            Slot sl;
            sl.off = pos;
            sl.cap = bound[i];
            bt.slots.push_back(sl);
            pos += bound[i];
            i++;
        }
        bt.last = i;
        bt.bytes = pos;
        batches.push_back(std::move(bt));
    }

    const bool timing = getenv("RK_TIMING") != nullptr;
    const double t_alloc = get_sec();
    const int n_buf = batches.size() > 1 ? 2 : 1;
    uint8_t *stage[2] = {nullptr, nullptr};
    void *dev[2] = {nullptr, nullptr};
    void *stream = nullptr;
    gpu.check(rk_stream_create(gpu.ctx, &stream), "rk_stream_create");
    bool need_pinned = false;
    for (const Batch &bt : batches) need_pinned |= !bt.pageable;
    for (int k = 0; k < n_buf; k++) {
        if (need_pinned) gpu.check(rk_pinned_alloc(gpu.ctx, stage_bytes, (void **)&stage[k]), "rk_pinned_alloc");
        gpu.check(rk_dev_alloc(gpu.ctx, dev_bytes, &dev[k]), "rk_dev_alloc");
    }

    if (timing) fprintf(stderr, "[timing] %d staging buffer(s) of %.1f MB pinned + device: %.3f s\n", n_buf, stage_bytes / 1e6, get_sec() - t_alloc);
    RawBuf big_stage;  // pageable staging of an oversized file
    // hand-over of filled staging buffers to the GPU thread
    std::mutex mu;
    std::condition_variable cv;
    size_t filled = 0, consumed = 0;  // batches parsed / batches whose staging buffer is free again
    uint64_t total_windows = 0;
    string gpu_error;

    auto slow_path = [&](size_t file_idx, rk_sketches **sk) {  // whole file through rk_sketch_batch_ex
        vector<uint8_t> seq, qual;
        vector<uint64_t> rec_off, genome_rec{0};
        if (!RecordReader::read_file(files[file_idx], seq, rec_off, fq.fastq ? &qual : nullptr))
            die("cannot open the genome file: %s", files[file_idx].c_str());
        if (rec_off.empty()) rec_off.push_back(0);
        genome_rec.push_back(rec_off.size() - 1);
        gpu.check(rk_sketch_batch_ex(gpu.ctx, flt, seq.data(), fq.fastq ? qual.data() : nullptr, fq.least_qual,
                                     (uint32_t)std::max(1, fq.least_num), rec_off.data(), rec_off.size() - 1,
                                     genome_rec.data(), 1, sk), "rk_sketch_batch_ex");
    };
    auto append_sketches = [&](rk_sketches *sk, uint32_t n) {
        vector<uint64_t> off((size_t)n + 1);
        const uint64_t b0 = out.total();
        if (out.wide()) {
            vector<uint64_t> h(rk_sketches_total(sk));
            gpu.check(rk_sketches_download64(sk, h.data(), off.data()), "rk_sketches_download64");
            out.hashes64.insert(out.hashes64.end(), h.begin(), h.end());
        } else {
            vector<uint32_t> h(rk_sketches_total(sk));
            gpu.check(rk_sketches_download(sk, h.data(), off.data()), "rk_sketches_download");
            out.hashes.insert(out.hashes.end(), h.begin(), h.end());
        }
        total_windows += rk_sketches_windows(sk);
        for (size_t i = 1; i <= n; i++) out.off.push_back(b0 + off[i]);
    };

    std::thread gpu_thread([&]() {
        for (size_t k = 0; k < batches.size(); k++) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return filled > k; });
            }
            Batch &bt = batches[k];
            const int bi = (int)(k % (size_t)n_buf);
            const uint32_t nb = (uint32_t)(bt.last - bt.first);
            vector<uint64_t> gbeg(nb), gend(nb);
            bool any_overflow = false;
            for (uint32_t i = 0; i < nb; i++) {
                gbeg[i] = bt.slots[i].off;
                gend[i] = bt.slots[i].off + (bt.slots[i].overflow ? 0 : bt.slots[i].len);
                any_overflow |= bt.slots[i].overflow;
            }
            rk_sketches *sk = nullptr;
            const double t_gpu = get_sec();
            if (bt.streamed) {
                const bool plain_fasta = !fq.fastq && !ends_with(files[bt.first], ".gz");
                const bool done = plain_fasta
                    ? sketch_big_fasta_streamed(gpu, flt, files[bt.first], 2 * shuf.k, threads, (uint32_t)std::max(1, fq.least_num), stream, &sk, timing)
                    : sketch_big_sequential(gpu, flt, files[bt.first], fq.fastq, fq.fastq ? fq.least_qual : 0, 2 * shuf.k, threads,
                                            (uint32_t)std::max(1, fq.least_num), stream, &sk, timing);
                if (!done) slow_path(bt.first, &sk);   // not a file these paths take ('\r' line ends, odd records): the serial reader
                {
                    std::lock_guard<std::mutex> lk(mu);
                    consumed = k + 1;
                }
                cv.notify_all();
                append_sketches(sk, 1);
                rk_sketches_free(sk);
                cerr << "finshed sketching: " << bt.last << " genomes" << endl;
                continue;
            }
            gpu.check(rk_upload_async(gpu.ctx, dev[bi], bt.pageable ? big_stage.data() : stage[bi], bt.bytes, stream),
                      "rk_upload_async");
            gpu.check(rk_sketch_packed_dev_ex(gpu.ctx, flt, (const uint8_t *)dev[bi], bt.bytes, gbeg.data(), gend.data(),
                                              nb, (uint32_t)std::max(1, fq.least_num), stream, &sk),
                      "rk_sketch_packed_dev_ex");
            if (timing) fprintf(stderr, "[timing] batch %zu: upload + sketch of %.1f MB: %.3f s\n", k, bt.bytes / 1e6, get_sec() - t_gpu);
            {   // the call above synchronised the stream: the staging buffer can be refilled
                std::lock_guard<std::mutex> lk(mu);
                consumed = k + 1;
            }
            cv.notify_all();
            if (!any_overflow) {
                append_sketches(sk, nb);
            } else {  // rare: splice the slow-path genomes in, keeping list order
                vector<uint64_t> off((size_t)nb + 1);
                vector<uint32_t> h32;
                vector<uint64_t> h64;
                if (out.wide()) { h64.resize(rk_sketches_total(sk)); gpu.check(rk_sketches_download64(sk, h64.data(), off.data()), "rk_sketches_download64"); }
                else { h32.resize(rk_sketches_total(sk)); gpu.check(rk_sketches_download(sk, h32.data(), off.data()), "rk_sketches_download"); }
                total_windows += rk_sketches_windows(sk);
                for (uint32_t i = 0; i < nb; i++) {
                    if (bt.slots[i].overflow) {
                        rk_sketches *one = nullptr;
                        slow_path(bt.first + i, &one);
                        append_sketches(one, 1);
                        rk_sketches_free(one);
                    } else {
                        if (out.wide()) out.hashes64.insert(out.hashes64.end(), h64.begin() + off[i], h64.begin() + off[i + 1]);
                        else out.hashes.insert(out.hashes.end(), h32.begin() + off[i], h32.begin() + off[i + 1]);
                        out.off.push_back(out.off.back() + (off[i + 1] - off[i]));
                    }
                }
            }
            rk_sketches_free(sk);
            cerr << "finshed sketching: " << bt.last << " genomes" << endl;
        }
    });

    for (size_t k = 0; k < batches.size(); k++) {
        {   // wait until the GPU thread has released this staging buffer
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return k < (size_t)n_buf || consumed + (size_t)n_buf > k; });
        }
        Batch &bt = batches[k];
        if (bt.pageable && !getenv("RK_BIG_WHOLE")) {
            // a big file: the GPU thread streams it piece by piece (read / inflate, parse, upload and kernels overlap)
            bt.streamed = true;
            {
                std::lock_guard<std::mutex> lk(mu);
                filled = k + 1;
            }
            cv.notify_all();
            continue;
        }
        if (bt.pageable) {  // its own buffer; the previous oversized batch must be uploaded first
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return consumed == k; });
            big_stage.resize(bt.bytes);
        }
        uint8_t *base = bt.pageable ? big_stage.data() : stage[k % (size_t)n_buf];
        const size_t nb = bt.last - bt.first;
        const double t_parse = get_sec();
        std::atomic<size_t> next_file{0};
        std::atomic<int> failed{-1};
        const int nt = std::max(1, std::min<int>(threads, (int)nb));
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; t++)
            pool.emplace_back([&]() {
                RawBuf buf;  // one allocation per thread, reused for all its files
                for (;;) {
                    const size_t i = next_file.fetch_add(1);
                    if (i >= nb) break;
                    Slot &sl = bt.slots[i];
                    size_t n = 0;
                    // a big plain file in a batch with idle parser threads: read and parse it in parallel
                    const int spare = std::max(1, threads / nt);
                    const bool big = spare > 1 && sl.cap >= (64u << 20) && !ends_with(files[bt.first + i], ".gz");
                    if (!(big ? RecordReader::slurp_parallel(files[bt.first + i], buf, n, spare)
                              : RecordReader::slurp(files[bt.first + i], buf, n))) { failed = (int)i; continue; }
                    RecordReader::Packed pk;
                    if (!(big && !fq.fastq &&
                          RecordReader::parse_packed_parallel(buf.data(), n, base + sl.off, sl.cap, spare, pk)))
                        pk = RecordReader::parse_packed(buf.data(), n, base + sl.off, sl.cap, fq.fastq ? fq.least_qual : 0);
                    sl.overflow = pk.overflow;
                    sl.len = pk.overflow ? 0 : pk.bytes;
                    // zero-fill up to the next multiple of 1024 (the kernel reads whole 1 KiB lines)
                    const uint64_t end = sl.len, pad_end = std::min<uint64_t>(sl.cap, (end + 1024) & ~1023ULL);
                    memset(base + sl.off + end, 0, pad_end - end);
                }
            });
        for (auto &th : pool) th.join();
        if (timing) fprintf(stderr, "[timing] batch %zu: read + parse of %zu file(s) on %d thread(s): %.3f s\n", k, nb, nt, get_sec() - t_parse);
        if (failed >= 0) die("cannot open the genome file: %s", files[bt.first + (size_t)failed].c_str());
        {
            std::lock_guard<std::mutex> lk(mu);
            filled = k + 1;
        }
        cv.notify_all();
    }
    gpu_thread.join();
    // the page-locked buffers go back to the system while the sketches are saved and the index is built (unpinning is
    // kernel work that needs nothing from this thread)
    const double t_free = get_sec();
    for (int k = 0; k < n_buf; k++) rk_dev_free(dev[k]);
    rk_stream_destroy(stream);
    // (detached: the GPU subcommands leave through _exit once their output is on disk)
    std::thread([st0 = stage[0], st1 = stage[1], timing, t_free] {
        rk_pinned_free(st0);
        rk_pinned_free(st1);
        if (timing) fprintf(stderr, "[timing] staging buffers released (in the background): %.3f s\n", get_sec() - t_free);
    }).detach();
    rk_filter_free(flt);

    string out_path = out_path_in;
    if (!is_sketch_file(out_path)) out_path += ".sketch";  // src/sketch.cpp:570-572
    string err;
    if (!save_sketches(out_path, out, err)) die("%s", err.c_str());
    cerr << "save the sketches into: " << out_path << endl;
    cerr << "===================time of sketching " << files.size() << " genomes (" << total_windows
         << " k-mers) is: " << get_sec() - t0 << endl;
    (void)is_query;
}

static rk_sketches *upload(Gpu &gpu, const SketchSet &s)
{
    rk_sketches *sk = nullptr;
    if (s.wide())
        gpu.check(rk_sketches_from_host64(gpu.ctx, s.hashes64.data(), s.off.data(), (uint32_t)s.size(), &sk),
                  "rk_sketches_from_host64");
    else
        gpu.check(rk_sketches_from_host(gpu.ctx, s.hashes.data(), s.off.data(), (uint32_t)s.size(), &sk),
                  "rk_sketches_from_host");
    return sk;
}

// builds the index on the GPU and, when asked, writes <sketch>.dict/.index (transSketches,
// src/sketch.cpp:894-1021)
static rk_index *build_index(Gpu &gpu, const SketchSet &s, const string &sketch_path, bool write_files)
{
    const double t0 = get_sec();
    const int bits = 4 * (s.info.half_k - s.info.drlevel);
    rk_sketches *sk = upload(gpu, s);
    rk_index *idx = nullptr;
    gpu.check(rk_index_build(gpu.ctx, sk, bits, &idx), "rk_index_build");
    rk_sketches_free(sk);
    if (write_files) {
        string err;
        vector<uint32_t> postings(rk_index_total(idx));
        if (s.wide()) {  // sparse layout, src/sketch.cpp:942-963
            vector<uint64_t> hashes(rk_index_distinct(idx));
            vector<uint32_t> counts(rk_index_distinct(idx));
            gpu.check(rk_index_export64(idx, postings.data(), hashes.data(), counts.data()), "rk_index_export64");
            if (!write_index64(sketch_path + ".dict", sketch_path + ".index", postings, hashes, counts, err)) die("%s", err.c_str());
        } else if (getenv("RK_INDEX_WRITE_LISTS")) {   // dense layout from the sparse form, scattered on the host (measured slower:
                                                          // 0.52-0.62 s against 0.28 s for 1,000 genomes at 28 hash bits)
            vector<uint32_t> hashes(rk_index_distinct(idx)), counts(rk_index_distinct(idx));
            gpu.check(rk_index_export_lists(idx, postings.data(), hashes.data(), counts.data()), "rk_index_export_lists");
            if (!write_index_lists(sketch_path + ".dict", sketch_path + ".index", postings.data(), postings.size(), hashes.data(),
                                   counts.data(), hashes.size(), (uint64_t)1 << bits, err,
                                   (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()))))
                die("%s", err.c_str());
        } else {         // dense layout, src/sketch.cpp:991-1011
            RawBuf counts;  // 4 * 2^bits bytes, every one of them written by the export: no zero-fill
            counts.resize((size_t)4 << bits);
            gpu.check(rk_index_export(idx, postings.data(), (uint32_t *)counts.data()), "rk_index_export");
            if (!write_index(sketch_path + ".dict", sketch_path + ".index", postings.data(), postings.size(),
                             (const uint32_t *)counts.data(), (uint64_t)1 << bits, err,
                             (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()))))
                die("%s", err.c_str());
        }
    }
    cerr << "===============the time of transSketches is: " << get_sec() - t0 << endl;
    return idx;
}

// ---- distance text (src/dist.cpp:233,256,263-336 / :642,690,696-770) --------------------------------------------
// The reference's worker threads each append their rows to a sub-file `<out>.dir/<out>.<tid>` and note
// "rowName\tsubFile" in `<out>.index.<tid>`; afterwards the sub-files are concatenated into <out> behind the header
// line when they total at most 4 GiB, otherwise they stay and the notes become `<out>.index` (:276-336).  Here a
// worker is a (GPU, formatter thread) pair: every GPU's hits (sorted by row) are cut at row boundaries into pieces,
// each piece is formatted once by one thread, and the pieces are either written into <out> with concurrent pwrites
// or kept as the sub-files.  Every row is listed exactly once in the index, hits or not, like the reference's.
struct HitPart {              // what one GPU computed
    const rk_hit *hits = nullptr;
    uint64_t n = 0;
    vector<uint32_t> rows;    // the rows it owns, ascending
};

static uint64_t max_merge_bytes()
{
    const char *v = getenv("RK_DIST_MAX_MERGE_BYTES");  // tests force the split on small outputs
    return v && atoll(v) > 0 ? (uint64_t)atoll(v) : 1ULL << 32;
}

static void write_hits(const string &out, const vector<HitPart> &parts, bool alldist, const vector<string> &rows,
                       const vector<string> &cols, int threads = 1)
{
    const double t0 = get_sec();
    const uint64_t max_size = max_merge_bytes();
    auto line = [&](const rk_hit &h, char *buf, size_t cap) {
        const string &a = alldist ? cols[h.col] : rows[h.row];
        const string &b = alldist ? rows[h.row] : cols[h.col];
        if (a.size() + b.size() + 128 > cap) die("genome name too long");
        return rk_format_hit(buf, cap, a.c_str(), b.c_str(), &h);
    };
    struct Piece {
        const rk_hit *hits;
        uint64_t n;
        const uint32_t *row_begin, *row_end;  // rows listed under this piece in the index
        string text;
    };
    vector<Piece> pieces;
    const int per_part = std::max(1, std::max(1, threads) / (int)std::max<size_t>(1, parts.size()));
    for (const HitPart &pt : parts) {
        // cut the part into up to per_part pieces of ~equal hit counts at row boundaries
        // (pieces of ~4,096 hits: the 45,000 lines of a 10,000-genome alldist were ONE piece, formatted by one thread in 10 ms)
        uint64_t want = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)per_part * 4, (pt.n + 4095) / 4096));
        if (max_size != (1ULL << 32)) want = (uint64_t)per_part;  // sub-file layout under test: one piece per worker
        uint64_t i0 = 0;
        size_t r0 = 0;
        for (uint64_t p = 0; p < want; p++) {
            uint64_t i1 = p + 1 == want ? pt.n : std::max(i0, pt.n / want * (p + 1));
            while (i1 < pt.n && i1 > 0 && pt.hits[i1].row == pt.hits[i1 - 1].row) i1++;  // whole rows
            size_t r1 = pt.rows.size();
            if (p + 1 < want && i1 < pt.n) r1 = std::lower_bound(pt.rows.begin(), pt.rows.end(), pt.hits[i1].row) - pt.rows.begin();
            if (p + 1 == want) i1 = pt.n;
            pieces.push_back(Piece{pt.hits + i0, i1 - i0, pt.rows.data() + r0, pt.rows.data() + r1, string()});
            i0 = i1;
            r0 = r1;
        }
    }
    const uint64_t n_pieces = pieces.size();
    {
        std::atomic<uint64_t> next{0};
        auto work = [&]() {
            vector<char> lb(1 << 16);
            for (;;) {
                const uint64_t p = next.fetch_add(1);
                if (p >= n_pieces) break;
                string &t = pieces[p].text;
                t.reserve((size_t)pieces[p].n * 72);
                for (uint64_t i = 0; i < pieces[p].n; i++) t.append(lb.data(), (size_t)line(pieces[p].hits[i], lb.data(), lb.size()));
            }
        };
        vector<std::thread> pool;
        for (int t = 1; t < std::max(1, threads) && (uint64_t)t < n_pieces; t++) pool.emplace_back(work);
        work();
        for (auto &th : pool) th.join();
    }
    uint64_t total = 0;
    for (const Piece &pc : pieces) total += pc.text.size();
    std::atomic<int> failed{0};
    if (total <= max_size) {  // isMerge, src/dist.cpp:286-310
        cerr << "-----save the output distance file: " << out << endl;
        const char *head = " genome0\tgenome1\tcommon|size0|size1\tjaccard\tmashD\n";
        const int fd = open(out.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) die("cannot write %s", out.c_str());
        bool good = pwrite(fd, head, strlen(head), 0) == (ssize_t)strlen(head);
        vector<uint64_t> at(n_pieces);
        uint64_t pos = strlen(head);
        for (uint64_t p = 0; p < n_pieces; p++) { at[p] = pos; pos += pieces[p].text.size(); }
        std::atomic<uint64_t> next{0};
        auto put = [&]() {
            for (;;) {
                const uint64_t p = next.fetch_add(1);
                if (p >= n_pieces) break;
                const string &t = pieces[p].text;
                uint64_t done = 0;
                while (done < t.size()) {
                    const ssize_t r = pwrite(fd, t.data() + done, t.size() - done, (off_t)(at[p] + done));
                    if (r <= 0) { failed = 1; break; }
                    done += (uint64_t)r;
                }
            }
        };
        vector<std::thread> pool;
        for (int t = 1; t < std::min(8, std::max(1, threads)) && (uint64_t)t < n_pieces; t++) pool.emplace_back(put);
        put();
        for (auto &th : pool) th.join();
        if (close(fd) || failed || !good) die("write error on %s", out.c_str());
    } else {  // src/dist.cpp:311-335: the sub-files stay, <out>.index maps every row to its sub-file
        const string dir = out + ".dir";
        if (mkdir(dir.c_str(), 0777) && errno != EEXIST) die("cannot create %s", dir.c_str());
        cerr << "-----the output distance file is too big to merge into one single file, saving the result into directory: "
             << dir << endl;
        const string index_path = out + ".index";
        cerr << "-----save the index between genomes and distance sub-files into: " << index_path << endl;
        auto sub_name = [&](uint64_t p) { return dir + '/' + out + '.' + std::to_string(p); };  // :154 / :543
        std::atomic<uint64_t> next{0};
        auto put = [&]() {
            for (;;) {
                const uint64_t p = next.fetch_add(1);
                if (p >= n_pieces) break;
                FILE *fp = fopen(sub_name(p).c_str(), "w");
                if (!fp || fwrite(pieces[p].text.data(), 1, pieces[p].text.size(), fp) != pieces[p].text.size()) failed = 1;
                if (fp && fclose(fp)) failed = 1;
            }
        };
        vector<std::thread> pool;
        for (int t = 1; t < std::min(8, std::max(1, threads)) && (uint64_t)t < n_pieces; t++) pool.emplace_back(put);
        put();
        for (auto &th : pool) th.join();
        FILE *fidx = fopen(index_path.c_str(), "w");
        if (!fidx) die("cannot write %s", index_path.c_str());
        fprintf(fidx, "genomeName\tdistFileName\n");
        for (uint64_t p = 0; p < n_pieces; p++) {
            const string name = sub_name(p);
            for (const uint32_t *r = pieces[p].row_begin; r != pieces[p].row_end; ++r) fprintf(fidx, "%s\t%s\n", rows[*r].c_str(), name.c_str());
        }
        if (fclose(fidx) || failed) die("write error under %s", dir.c_str());
    }
    cerr << "===================time of merge the subFiles into final files is: " << get_sec() - t0 << endl;
}

static void load_or_sketch(Gpu &gpu, const string &input, bool is_query, const Args &a, int threads, SketchSet &s,
                           string &sketch_path)
{
    string err;
    if (is_sketch_file(input)) {
        sketch_path = input;
        if (!read_sketches(input, s, err)) die("readSketches(), %s", err.c_str());
        return;
    }
    // list of FASTA files (src/subCommand.cpp:174-181): sketch into <list>.sketch
    const vector<string> files = read_list(input);
    if (files.empty()) die("cannot open the inputFile or empty list: %s", input.c_str());
    FastqOpts fq;
    fq.fastq = list_is_fastq(files);
    fq.least_qual = a.num("Q", 0);
    fq.least_num = a.num("n", 1);
    Shuf shuf;
    const string shuf_file = a.str("L", "shuf_file/L3K10.shuf");
    cerr << "---read the shuffle file: " << shuf_file << endl;
    if (!read_shuf(shuf_file, shuf, err)) die("read_shuffle_dim(), %s", err.c_str());
    sketch_path = input + ".sketch";
    sketch_list(gpu, input, is_query, shuf, threads, s, sketch_path, fq);
}

static int cmd_shuffle(const Args &a)
{
    const string out = a.str("o", "result.out");
    cerr << "-----generate the shuffle file: " << out << endl;
    string err;
    if (!write_shuf(out, a.num("k", 10), a.num("s", 6), a.num("l", 3), err)) die("write_shuffle_dim_file(), %s", err.c_str());
    return 0;
}

static int cmd_sketch(const Args &a)
{
    if (!a.has("i") || !a.has("o")) die("sketch needs -i and -o");
    const string in = a.str("i", ""), out = a.str("o", "");
    const bool is_query = a.has("q");
    const int threads = a.num("t", (int)std::thread::hardware_concurrency());
    Gpu gpu(a.num("device", 0));
    string err;
    if (is_sketch_file(in)) {  // src/main.cpp:189-214: copy, and (re)build the index unless -q
        SketchSet s;
        if (!read_sketches(in, s, err)) die("readSketches(), %s", err.c_str());
        if (!save_sketches(out, s, err)) die("%s", err.c_str());
        if (!is_query) rk_index_free(build_index(gpu, s, out, true));
        return 0;
    }
    Shuf shuf;
    const string shuf_file = a.str("L", "shuf_file/L3K10.shuf");
    cerr << "---read the shuffle file: " << shuf_file << endl;
    if (!read_shuf(shuf_file, shuf, err)) die("read_shuffle_dim(), %s", err.c_str());
    FastqOpts fq;
    fq.fastq = list_is_fastq(read_list(in));
    fq.least_qual = a.num("Q", 0);
    fq.least_num = a.num("n", 1);
    SketchSet s;
    string out_path = out;
    sketch_list(gpu, in, is_query, shuf, threads, s, out_path, fq);
    if (!is_sketch_file(out_path)) out_path += ".sketch";
    if (!is_query) rk_index_free(build_index(gpu, s, out_path, true));
    return 0;
}

// ---- several GPUs in one process (rows are independent: `#pragma omp parallel for` over rows, src/dist.cpp:174,560)
// One context and one host thread per GPU.  Every GPU builds its own copy of the index from the host's sketches
// (build_everywhere; RK_MULTI_BROADCAST=1: built once, replicated with rk_index_broadcast); alldist deals blocks of 32 rows round-robin,
// dist hands every GPU a contiguous block of queries; per-GPU hits go to the writer as they are (no reduction).
struct GpuSet {
    vector<std::unique_ptr<AsyncGpu>> gpus;
    GpuSet(int first_device, int n, bool same_device)
    {
        if (n < 1) die("--gpus must be at least 1");
        for (int g = 0; g < n; g++) gpus.emplace_back(new AsyncGpu(same_device ? first_device : first_device + g));
    }
    size_t size() const { return gpus.size(); }
    Gpu &operator[](size_t g) { return gpus[g]->get(); }
};

static const int kRowBlock = 32;   // rows are dealt to the GPUs in blocks of 32: a multiple of the tile kernel's block, an even number for the row pairs

// The index on every GPU of the set.  Default: every GPU uploads the sketches over its own PCIe link and builds its own
// index, all at the same time (the build is deterministic: same internal genome order, same postings everywhere) -- 49 MB
// of sketches and 0.5 ms of build per GPU at 10,000 genomes, against a 147 MB blob pulled from the first GPU once IT is done.
// RK_MULTI_BROADCAST=1: build on the first GPU, replicate with rk_index_broadcast (peer copies over xGMI).
static void build_everywhere(GpuSet &set, const SketchSet &s, const string &path, bool write_files, vector<rk_index *> &idx)
{
    const size_t G = set.size();
    if (G > 1 && getenv("RK_MULTI_BROADCAST") && atoi(getenv("RK_MULTI_BROADCAST"))) {
        idx[0] = build_index(set[0], s, path, write_files);
        vector<rk_ctx *> peers;
        for (size_t g = 1; g < G; g++) peers.push_back(set[g].ctx);
        set[0].check(rk_index_broadcast(idx[0], peers.data(), (uint32_t)peers.size(), idx.data() + 1), "rk_index_broadcast");
        return;
    }
    vector<std::thread> pool;
    for (size_t g = 1; g < G; g++) pool.emplace_back([&, g] { idx[g] = build_index(set[g], s, path, false); });
    idx[0] = build_index(set[0], s, path, write_files);
    for (auto &th : pool) th.join();
}

static int cmd_alldist(const Args &a)
{
    if (!a.has("i")) die("alldist needs -i");
    const double max_dist = a.real("D", 1.0);
    if (max_dist < 0.0) die("command_alldist(), maxDist must be > 0\nUse -D to set the maxDist");
    const string out = a.str("o", "result.out");
    const int metric = a.num("M", 0);
    const int threads = a.num("t", (int)std::thread::hardware_concurrency());
    GpuSet set(a.num("device", 0), a.num("gpus", 1), a.has("same-device"));
    const double t0 = get_sec();
    SketchSet s;
    string sketch_path;
    if (is_sketch_file(a.str("i", ""))) {  // read while the runtime starts
        string err;
        sketch_path = a.str("i", "");
        if (!read_sketches(sketch_path, s, err)) die("readSketches(), %s", err.c_str());
    }
    Gpu &gpu = set[0];
    if (sketch_path.empty()) load_or_sketch(gpu, a.str("i", ""), false, a, threads, s, sketch_path);
    stamp("sketches read");
    // the .dict/.index pair is (re)written only if missing (src/subCommand.cpp:165-169); the
    // device index is always rebuilt from the sketches: faster than reading the 2^bits array
    const bool missing = !exist_file(sketch_path + ".index") || !exist_file(sketch_path + ".dict");
    const size_t G = set.size();
    vector<rk_index *> idx(G, nullptr);
    // Several GPUs (round 5): the all-vs-all shards twice -- every GPU builds the posting lists of ITS range of the hash space, the
    // tile records change hands once (GPU d pulls what belongs to its rows over its own links), every GPU sorts what arrived and
    // joins its rows.  No index is replicated.  Taken for a power-of-two number of GPUs when the .dict / .index pair exists (writing
    // them needs the whole index on one GPU) and the collection takes the bucket sort with tile records (set sketches, 2 and more
    // genomes); anything else -- and RK_MULTI_REPLICATE=1 -- builds the whole index on every GPU as before.
    // (a dense report -- -D above 1.0: every pair -- needs counter rows over slice records, which a join-only index has not)
    bool sharded = G > 1 && (G & (G - 1)) == 0 && !missing && !(1.0 < max_dist) && !(getenv("RK_MULTI_REPLICATE") && atoi(getenv("RK_MULTI_REPLICATE")));
    if (sharded) {
        const int bits = 4 * (s.info.half_k - s.info.drlevel);
        vector<rk_index *> part(G, nullptr);
        vector<int> rcs(G, 0);
        auto build_part = [&](size_t g) {
            rk_sketches *sk = upload(set[g], s);
            rcs[g] = rk_index_build_shard(set[g].ctx, sk, bits, (uint32_t)g, (uint32_t)G, &part[g]);
            rk_sketches_free(sk);
        };
        {
            vector<std::thread> pool;
            for (size_t g = 1; g < G; g++) pool.emplace_back(build_part, g);
            build_part(0);
            for (auto &th : pool) th.join();
        }
        bool ok = true;
        for (size_t g = 0; g < G; g++) ok = ok && rcs[g] == 0;
        vector<void *> recv(G, nullptr);
        vector<uint64_t> n_recv(G, 0);
        if (ok) ok = rk_index_shard_exchange(part.data(), (uint32_t)G, recv.data(), n_recv.data()) == 0;
        if (ok) stamp("shards built, tile records exchanged");
        if (ok) {
            auto join_part = [&](size_t g) { rcs[g] = rk_index_join_shard(set[g].ctx, part[g], recv[g], n_recv[g], &idx[g]); };
            vector<std::thread> pool;
            for (size_t g = 1; g < G; g++) pool.emplace_back(join_part, g);
            join_part(0);
            for (auto &th : pool) th.join();
            for (size_t g = 0; g < G; g++) ok = ok && rcs[g] == 0;
        }
        for (size_t g = 0; g < G; g++) {
            rk_dev_free(recv[g]);
            rk_index_free(part[g]);
            if (!ok && idx[g]) { rk_index_free(idx[g]); idx[g] = nullptr; }
        }
        if (!ok && getenv("RK_TIMING")) fprintf(stderr, "[timing] sharded build refused (%s): the whole index on every GPU\n", rk_last_error(set[0].ctx));
        sharded = ok;
    }
    if (!sharded) build_everywhere(set, s, sketch_path, missing, idx);
    stamp("index built");
    // (the reference's phase line, src/dist.cpp:132-135: there the phase loads .dict/.index, here it builds the index on the device)
    cerr << "===================time of read index and offset sketch file is: " << get_sec() - t0 << endl;
    const double t1 = get_sec();
    cerr << "=====total: " << s.size() << endl;
    vector<rk_hit *> hits(G, nullptr);
    vector<uint64_t> n_hits(G, 0);
    auto rows_of = [&](size_t g) {
        rk_dist_opts o{};
        o.triangle = 1;
        o.metric = metric;
        o.kmer_size = 2 * s.info.half_k;
        o.max_dist = max_dist;
        if (G > 1 && !sharded) {   // (a join-only index of the sharded flow holds this GPU's rows and nothing else)
            o.row_first = (uint32_t)g;
            o.row_step = (uint32_t)G;
            o.row_block = kRowBlock;
        }
        set[g].check(rk_dist_rows(set[g].ctx, idx[g], nullptr, &o, &hits[g], &n_hits[g], nullptr), "rk_dist_rows");
    };
    {
        vector<std::thread> pool;
        for (size_t g = 1; g < G; g++) pool.emplace_back(rows_of, g);
        rows_of(0);
        for (auto &th : pool) th.join();
    }
    stamp("distances on the host");
    cerr << "===================time of multiple threads distance computing and save the subFile is: " << get_sec() - t1 << endl;
    vector<HitPart> parts(G);
    for (size_t g = 0; g < G; g++) {
        parts[g].hits = hits[g];
        parts[g].n = n_hits[g];
        for (uint32_t r = 0; r < (uint32_t)s.size(); r++)
            if (G == 1 || (r / kRowBlock) % G == g) parts[g].rows.push_back(r);
    }
    write_hits(out, parts, true, s.names, s.names, threads);
    stamp("text written");
    // The output is on disk and the process is about to end: handing 50 MB of sketches, the hit records and the index back block
    // by block costs 4-5 ms that buy nothing (the tool's last stamp).  Straight out, like the GPU subcommands' `leave` in main.
    stamp("done");
    fflush(stdout);
    fflush(stderr);
    _exit(0);
}

static int cmd_dist(const Args &a)
{
    if (!a.has("r") || !a.has("q")) die("dist needs -r and -q");
    const double max_dist = a.real("D", 1.0);
    if (max_dist < 0.0) die("command_dist(), maxDist must be > 0\nUse -D to set the maxDist");
    const int max_neighbor = a.num("N", 1);
    if (max_neighbor < 0) die("command_dist(), maxNeighbor must be > 0\nUse -N to set the maxNeighbor");
    const bool is_neighbor = a.has("N");
    const string out = a.str("o", "result.out");
    const int metric = a.num("M", 0);
    const int threads = a.num("t", (int)std::thread::hardware_concurrency());
    GpuSet set(a.num("device", 0), a.num("gpus", 1), a.has("same-device"));
    const double t_cmd0 = get_sec();
    SketchSet ref, qry;
    string ref_path, qry_path;
    {   // .sketch inputs are read while the runtime starts
        string err;
        if (is_sketch_file(a.str("r", ""))) {
            ref_path = a.str("r", "");
            if (!read_sketches(ref_path, ref, err)) die("readSketches(), %s", err.c_str());
        }
        if (is_sketch_file(a.str("q", ""))) {
            qry_path = a.str("q", "");
            if (!read_sketches(qry_path, qry, err)) die("readSketches(), %s", err.c_str());
        }
    }
    Gpu &gpu = set[0];
    if (ref_path.empty()) load_or_sketch(gpu, a.str("r", ""), false, a, threads, ref, ref_path);
    cerr << "the ref_sketch size is: " << ref.size() << endl;
    if (qry_path.empty()) load_or_sketch(gpu, a.str("q", ""), true, a, threads, qry, qry_path);
    if (qry.info.id != ref.info.id)  // src/subCommand.cpp:297-301
        die("command_dist(), the sketch infos between reference and query files are not match\n"
            "try to use the same shuffle file to generate sketches of the reference and query datasets");
    const bool missing = !exist_file(ref_path + ".index") || !exist_file(ref_path + ".dict");
    const size_t G = set.size();
    vector<rk_index *> idx(G, nullptr);
    build_everywhere(set, ref, ref_path, missing, idx);
    const double t1 = get_sec();
    cerr << "===================time of read index and offset sketch file is: " << t1 - t_cmd0 << endl;   // src/dist.cpp:482-485
    cerr << "=====total: " << qry.size() << endl;
    // contiguous query blocks: GPU g gets queries [q0[g], q0[g+1]) as a sketch set of its own (the host scatters the
    // queries) and reports rows relative to it
    const uint32_t Q = (uint32_t)qry.size();
    vector<uint32_t> q0(G + 1, Q);
    for (size_t g = 0; g <= G; g++) q0[g] = (uint32_t)((uint64_t)Q * g / G);
    vector<rk_hit *> hits(G, nullptr);
    vector<uint64_t> n_hits(G, 0);
    auto rows_of = [&](size_t g) {
        Gpu &dev = set[g];
        const uint32_t nq = q0[g + 1] - q0[g];
        vector<uint64_t> off(nq + 1);
        for (uint32_t i = 0; i <= nq; i++) off[i] = qry.off[q0[g] + i] - qry.off[q0[g]];
        rk_sketches *qs = nullptr;
        if (qry.wide())
            dev.check(rk_sketches_from_host64(dev.ctx, qry.hashes64.data() + qry.off[q0[g]], off.data(), nq, &qs), "rk_sketches_from_host64");
        else
            dev.check(rk_sketches_from_host(dev.ctx, qry.hashes.data() + qry.off[q0[g]], off.data(), nq, &qs), "rk_sketches_from_host");
        rk_dist_opts o{};
        o.triangle = 0;
        o.metric = metric;
        o.kmer_size = 2 * ref.info.half_k;
        o.max_dist = max_dist;
        dev.check(rk_dist_rows(dev.ctx, idx[g], qs, &o, &hits[g], &n_hits[g], nullptr), "rk_dist_rows");
        for (uint64_t i = 0; i < n_hits[g]; i++) hits[g][i].row += q0[g];
        if (is_neighbor) rk_topn_rows(hits[g], &n_hits[g], (uint64_t)max_neighbor);
        rk_sketches_free(qs);
    };
    {
        vector<std::thread> pool;
        for (size_t g = 1; g < G; g++) pool.emplace_back(rows_of, g);
        rows_of(0);
        for (auto &th : pool) th.join();
    }
    cerr << "===================time of multiple threads distance computing and save the subFile is: " << get_sec() - t1 << endl;
    vector<HitPart> parts(G);
    for (size_t g = 0; g < G; g++) {
        parts[g].hits = hits[g];
        parts[g].n = n_hits[g];
        for (uint32_t r = q0[g]; r < q0[g + 1]; r++) parts[g].rows.push_back(r);
    }
    write_hits(out, parts, false, qry.names, ref.names, threads);
    for (size_t g = 0; g < G; g++) {
        rk_free_host(hits[g]);
        rk_index_free(idx[g]);
    }
    return 0;
}

// test helper: the text writer alone (no GPU): `_format alldist|dist names.txt hits.bin out pieces threads`;
// hits.bin = rk_hit records sorted by (row, col), dealt to `pieces` parts in blocks of 32 rows like --gpus does
static int cmd_format(int argc, char **argv)
{
    if (argc != 8) die("_format alldist|dist names.txt hits.bin out parts threads");
    const bool alldist = string(argv[2]) == "alldist";
    const vector<string> names = read_list(argv[3]);
    FILE *fp = fopen(argv[4], "rb");
    if (!fp) die("cannot open %s", argv[4]);
    vector<rk_hit> all;
    rk_hit h;
    while (fread(&h, sizeof(h), 1, fp) == 1) all.push_back(h);
    fclose(fp);
    const size_t G = (size_t)std::max(1, atoi(argv[6]));
    vector<vector<rk_hit>> per(G);
    for (const rk_hit &x : all) per[G == 1 ? 0 : (x.row / kRowBlock) % G].push_back(x);
    vector<HitPart> parts(G);
    for (size_t g = 0; g < G; g++) {
        parts[g].hits = per[g].data();
        parts[g].n = per[g].size();
        for (uint32_t r = 0; r < (uint32_t)names.size(); r++)
            if (G == 1 || (r / kRowBlock) % G == g) parts[g].rows.push_back(r);
    }
    write_hits(argv[5], parts, alldist, names, names, atoi(argv[7]));
    return 0;
}

static int cmd_info(const Args &a)
{
    if (!a.has("i")) die("info needs -i");
    SketchSet s;
    string err;
    if (!read_sketches(a.str("i", ""), s, err)) die("command_info(), %s", err.c_str());
    cerr << "the number of genome is: " << s.size() << endl;
    FILE *fp = fopen(a.str("o", "result.out").c_str(), "w+");
    if (!fp) die("cannot write %s", a.str("o", "result.out").c_str());
    fprintf(fp, "the number of sketches are: %d\n", (int)s.size());  // src/subCommand.cpp:93
    for (size_t i = 0; i < s.size(); i++) {
        const uint64_t cnt = s.off[i + 1] - s.off[i];
        fprintf(fp, "%s\t%d\n", s.names[i].c_str(), (int)cnt);
        if (a.has("F")) {
            for (uint64_t j = 0; j < cnt; j++) {
                if (s.wide()) fprintf(fp, "%lu\t", (unsigned long)s.hashes64[s.off[i] + j]);
                else fprintf(fp, "%u\t", s.hashes[s.off[i] + j]);
                if (j % 10 == 9) fprintf(fp, "\n");
            }
            fprintf(fp, "\n");
        }
    }
    fclose(fp);
    return 0;
}

static int cmd_merge(const Args &a)
{
    if (!a.has("i") || !a.has("o")) die("merge needs -i and -o");
    SketchSet all;
    string err;
    bool first = true;
    for (const string &f : read_list(a.str("i", ""))) {
        if (!is_sketch_file(f)) die("command_merge(), the file: %s is not a sketch file in the list file: %s", f.c_str(), a.str("i", "").c_str());
        SketchSet s;
        if (!read_sketches(f, s, err)) die("command_merge(), %s", err.c_str());
        if (first) { all.info = s.info; first = false; }
        else if (s.info.id != all.info.id) die("command_merge(), mismatched sketch parameters in %s", f.c_str());
        const uint64_t b0 = all.total();
        all.names.insert(all.names.end(), s.names.begin(), s.names.end());
        all.hashes.insert(all.hashes.end(), s.hashes.begin(), s.hashes.end());
        all.hashes64.insert(all.hashes64.end(), s.hashes64.begin(), s.hashes64.end());
        for (size_t i = 1; i < s.off.size(); i++) all.off.push_back(b0 + s.off[i]);
    }
    string out = a.str("o", "");
    if (!is_sketch_file(out)) out += ".sketch";
    if (!save_sketches(out, all, err)) die("%s", err.c_str());
    return 0;
}

// ---- Kssd <-> RabbitKSSD conversion (src/sketch.cpp:1179-1365, src/subCommand.cpp:13-47) ------
struct CoDstat {  // co_dstat_t, src/sketch.h:38-47 (32 bytes, natural x86-64 layout)
    uint32_t shuf_id;
    uint8_t koc, pad_[3];
    int32_t kmerlen, dim_rd_len, comp_num, infile_num;
    uint64_t all_ctx_ct;
};
static_assert(sizeof(CoDstat) == 32, "co_dstat_t is 32 bytes");
static const int kPathLen = 256;  // PATHLEN, src/sketch.cpp:25

static int cmd_convert(const Args &a)
{
    if (!a.has("i") || !a.has("o")) die("convert needs -i and -o");
    const string in = a.str("i", ""), out_arg = a.str("o", "");
    string err;
    if (a.has("reverse")) {  // RabbitKSSD .sketch -> Kssd directory (:1288-1365)
        if (!is_sketch_file(in)) die("command_convert(), need input RabbitKSSD sketch file: %s", in.c_str());
        SketchSet s;
        if (!read_sketches(in, s, err)) die("readSketches(), %s", err.c_str());
        if (s.wide()) die("the Kssd sketch format holds 32-bit hashes only (half_k - drlevel <= 8)");
        if (mkdir(out_arg.c_str(), 0777) && errno != EEXIST) die("cannot create %s", out_arg.c_str());
        FILE *fs = fopen((out_arg + "/combco.0").c_str(), "w");
        if (!fs) die("cannot open: %s/combco.0", out_arg.c_str());
        fwrite(s.hashes.data(), 4, s.hashes.size(), fs);
        fclose(fs);
        FILE *fi = fopen((out_arg + "/combco.index.0").c_str(), "w+");
        if (!fi) die("cannot open: %s/combco.index.0", out_arg.c_str());
        fwrite(s.off.data(), 8, s.off.size(), fi);  // size_t prefix[infile_num+1]
        fclose(fi);
        CoDstat st{};
        st.shuf_id = (uint32_t)s.info.id;
        st.koc = 0;
        st.kmerlen = s.info.half_k * 2;
        st.dim_rd_len = s.info.drlevel * 2;
        st.comp_num = 1;
        st.infile_num = s.info.genomeNumber;
        st.all_ctx_ct = s.hashes.size();
        FILE *ft = fopen((out_arg + "/cofiles.stat").c_str(), "w+");
        if (!ft) die("cannot open: %s/cofiles.stat", out_arg.c_str());
        fwrite(&st, sizeof(st), 1, ft);
        for (size_t i = 0; i < s.size(); i++) {
            const uint32_t c = (uint32_t)(s.off[i + 1] - s.off[i]);
            fwrite(&c, 4, 1, ft);
        }
        for (size_t i = 0; i < s.size(); i++) {
            if (s.names[i].size() >= (size_t)kPathLen) die("genome name longer than %d bytes: %s", kPathLen - 1, s.names[i].c_str());
            char name[kPathLen] = {0};
            memcpy(name, s.names[i].data(), s.names[i].size());
            fwrite(name, 1, kPathLen, ft);
        }
        fclose(ft);
        return 0;
    }
    // Kssd directory -> .sketch (+ .dict/.index unless -q) (:1179-1285)
    FILE *ft = fopen((in + "/cofiles.stat").c_str(), "r");
    if (!ft) die("cannot open: %s/cofiles.stat", in.c_str());
    CoDstat st;
    if (fread(&st, sizeof(st), 1, ft) != 1 || st.infile_num < 0) die("convertSketch(), mismatched read cur_stat");
    SketchSet s;
    s.info.half_k = st.kmerlen / 2;
    s.info.half_subk = 6;  // hard-wired by the reference, :1197
    s.info.drlevel = st.dim_rd_len / 2;
    if (s.info.half_k - s.info.drlevel > 8) die("64-bit hash sketches are not supported by this build");
    const size_t n = (size_t)st.infile_num;
    vector<uint32_t> ctx_ct(n);
    if (fread(ctx_ct.data(), 4, n, ft) != n) die("convertSketch(), mismatched read tmp_ctx_ct");
    s.names.resize(n);
    for (size_t i = 0; i < n; i++) {
        char name[kPathLen + 1] = {0};
        if (fread(name, 1, kPathLen, ft) == 0) cerr << "Warning: convertSketch(), the read path length is zero " << endl;
        s.names[i] = name;
    }
    fclose(ft);
    FILE *fi = fopen((in + "/combco.index.0").c_str(), "rb");
    if (!fi) die("convertSketch(), cannot open: %s/combco.index.0", in.c_str());
    s.off.assign(n + 1, 0);
    if (fread(s.off.data(), 8, n + 1, fi) != n + 1) die("convertSketch(), mismatched read cbdcoindex");
    fclose(fi);
    FILE *fs = fopen((in + "/combco.0").c_str(), "rb");
    if (!fs) die("cannot open: %s/combco.0", in.c_str());
    s.hashes.resize(s.off[n]);
    if (st.all_ctx_ct != s.off[n] || fread(s.hashes.data(), 4, s.hashes.size(), fs) != s.hashes.size())
        die("the total hash number is not match to the state info, exit");
    fclose(fs);
    for (size_t i = 0; i < n; i++)
        if (s.off[i + 1] < s.off[i]) die("convertSketch(), corrupt combco.index.0");
    string out = out_arg;
    if (!is_sketch_file(out)) out += ".sketch";
    if (!save_sketches(out, s, err)) die("%s", err.c_str());
    if (!a.has("q")) {
        Gpu gpu(a.num("device", 0));
        rk_index_free(build_index(gpu, s, out, true));
    }
    return 0;
}

// ---- set algebra on sketch files (src/subCommand.cpp:307-794); host only ------------------
static int cmd_union(const Args &a)
{
    if (!a.has("i") || !a.has("o")) die("union needs -i and -o");
    const string in = a.str("i", "");
    if (!is_sketch_file(in)) die("command_union, %s is not sketch file, need input sketch file", in.c_str());
    SketchSet s;
    string err;
    if (!read_sketches(in, s, err)) die("command_union(), %s", err.c_str());
    cerr << "the total genome number in sketch file is: " << s.size() << endl;
    SketchSet u;
    u.info = s.info;
    // ascending hash order == the reference's bitmap walk (:493-520)
    if (s.wide()) {
        u.hashes64 = s.hashes64;
        std::sort(u.hashes64.begin(), u.hashes64.end());
        u.hashes64.erase(std::unique(u.hashes64.begin(), u.hashes64.end()), u.hashes64.end());
    } else {
        u.hashes = s.hashes;
        std::sort(u.hashes.begin(), u.hashes.end());
        u.hashes.erase(std::unique(u.hashes.begin(), u.hashes.end()), u.hashes.end());
    }
    u.names = {in + " merged sketches"};  // :372
    u.off = {0, s.wide() ? u.hashes64.size() : u.hashes.size()};
    if (!save_sketches(a.str("o", ""), u, err)) die("%s", err.c_str());
    return 0;
}

static int cmd_sub(const Args &a)
{
    if (!a.has("rs") || !a.has("qs") || !a.has("o")) die("sub needs --rs, --qs and -o");
    const string rs = a.str("rs", ""), qs = a.str("qs", "");
    if (!is_sketch_file(rs)) die("command_sub(), %s is not sketch file, need input sketch file", rs.c_str());
    if (!is_sketch_file(qs)) die("command_sub(): %s is not sketch file, need input sketch file", qs.c_str());
    SketchSet ref, qry;
    string err;
    if (!read_sketches(rs, ref, err)) die("command_sub(), %s", err.c_str());
    if (!read_sketches(qs, qry, err)) die("command_sub(), %s", err.c_str());
    if (qry.info.id != ref.info.id)
        die("command_sub(): the sketch infos between subtraction reference and query sketches are not same");
    SketchSet out;
    out.info = qry.info;
    out.names = qry.names;
    out.off.assign(1, 0);
    if (qry.wide()) {  // the reference's 2^bits bitmap (:561-572) is replaced by a sorted set
        vector<uint64_t> rs64 = ref.hashes64;
        std::sort(rs64.begin(), rs64.end());
        for (size_t i = 0; i < qry.size(); i++) {
            for (uint64_t e = qry.off[i]; e < qry.off[i + 1]; e++)
                if (!std::binary_search(rs64.begin(), rs64.end(), qry.hashes64[e])) out.hashes64.push_back(qry.hashes64[e]);
            out.off.push_back(out.hashes64.size());
        }
    } else {
        // one bit per hash value, MSB first (:575-583); sized by the sketches' hash space (2^28 values = 32 MiB of bits
        // at L3K10), not by the 2^32 of the type
        const int bits = std::min(32, std::max(1, 4 * (qry.info.half_k - qry.info.drlevel)));
        uint32_t top = 0;
        for (uint32_t h : ref.hashes) top = std::max(top, h);
        for (uint32_t h : qry.hashes) top = std::max(top, h);
        const uint64_t space = std::max<uint64_t>((uint64_t)1 << bits, (uint64_t)top + 1);  // a foreign file may exceed its declared space
        vector<uint64_t> dict((size_t)((space + 63) / 64), 0);
        for (uint32_t h : ref.hashes) dict[h / 64] |= 0x8000000000000000ULL >> (h % 64);
        for (size_t i = 0; i < qry.size(); i++) {
            for (uint64_t e = qry.off[i]; e < qry.off[i + 1]; e++) {
                const uint32_t h = qry.hashes[e];
                if (!(dict[h / 64] & (0x8000000000000000ULL >> (h % 64)))) out.hashes.push_back(h);
            }
            out.off.push_back(out.hashes.size());
        }
    }
    if (!save_sketches(a.str("o", ""), out, err)) die("%s", err.c_str());
    return 0;
}

// test helper: record reader parity (prints records, bases, FNV-1a hashes of the sequence and of the
// quality bytes, record end offsets -- the format of oracle/_ref/ref_driver kseq)
static int cmd_parse(int argc, char **argv)
{
    for (int i = 2; i < argc; i++) {
        vector<uint8_t> seq, qual;
        vector<uint64_t> off;
        if (!RecordReader::read_file(argv[i], seq, off, &qual)) die("cannot open %s", argv[i]);
        if (off.empty()) off.push_back(0);
        qual.resize(seq.size(), '~');
        {   // the packed sink must produce the same records, separated by 0x00
            vector<uint8_t> buf, packed;
            size_t n = 0;
            RecordReader::slurp(argv[i], buf, n);
            packed.assign(n + 2, 0xEE);
            const RecordReader::Packed pk = RecordReader::parse_packed(buf.data(), n, packed.data(), n + 1);
            vector<uint8_t> want;
            for (size_t r = 1; r < off.size(); r++) {
                want.insert(want.end(), seq.begin() + off[r - 1], seq.begin() + off[r]);
                if (r + 1 < off.size()) want.push_back(0);
            }
            if (pk.overflow || pk.n_rec != off.size() - 1 || pk.bytes != want.size() ||
                memcmp(want.data(), packed.data(), want.size()) != 0)
                die("packed parse differs from the vector parse for %s", argv[i]);
            // the parallel parser either declines or agrees
            for (int threads = 2; threads <= 5; threads += 3) {
                RecordReader::Packed pp;
                std::fill(packed.begin(), packed.end(), 0xEE);
                if (RecordReader::parse_packed_parallel(buf.data(), n, packed.data(), n + 1, threads, pp, 0) &&
                    (pp.overflow || pp.n_rec != pk.n_rec || pp.bytes != pk.bytes ||
                     memcmp(want.data(), packed.data(), want.size()) != 0))
                    die("parallel packed parse differs from the serial parse for %s (%d threads)", argv[i], threads);
            }
        }
        uint64_t h = 1469598103934665603ULL, hq = 1469598103934665603ULL;
        for (uint8_t c : seq) { h ^= c; h *= 1099511628211ULL; }
        for (uint8_t c : qual) { hq ^= c; hq *= 1099511628211ULL; }
        printf("%s\t%zu\t%zu\t%016llx\t%016llx", argv[i], off.size() - 1, seq.size(), (unsigned long long)h,
               (unsigned long long)hq);
        for (size_t r = 1; r < off.size(); r++) printf("\t%llu", (unsigned long long)off[r]);
        printf("\n");
    }
    return 0;
}

static int usage()
{
    cerr << "rabbit_kssd (MI355X build, " << rk_version() << ")\n"
            "subcommands: shuffle sketch alldist dist union sub convert merge info\n"
            "  shuffle -k K -s S -l L -o out.shuf\n"
            "  sketch  -i genomes.list -o out[.sketch] [-L file.shuf] [-t T] [-q] [--device N]\n"
            "  alldist -i in.sketch|genomes.list -o out [-D maxDist] [-M 0|1] [-L file.shuf] [--device N] [--gpus G]\n"
            "  dist    -r ref.sketch|list -q qry.sketch|list -o out [-D maxDist] [-N n] [-M 0|1] [--device N] [--gpus G]\n"
            "  info    -i in.sketch -o out [-F]\n"
            "  merge   -i sketches.list -o out.sketch\n"
            "  union   -i in.sketch -o out.sketch\n"
            "  sub     --rs ref.sketch --qs qry.sketch -o out.sketch\n"
            "  convert -i kssd_dir -o out[.sketch] [-q]   |   convert --reverse -i in.sketch -o kssd_dir\n";
    return 1;
}

int main(int argc, char **argv)
{
    stamp("main");
    if (argc < 2) return usage();
    const string sub = argv[1];
    const std::map<string, string> alias = {
        {"-k", "k"}, {"--halfk", "k"}, {"-s", "s"}, {"--subk", "s"}, {"-l", "l"}, {"--reduction", "l"},
        {"-o", "o"}, {"--output", "o"}, {"-i", "i"}, {"--input", "i"}, {"-L", "L"}, {"-t", "t"}, {"--threads", "t"},
        {"-n", "n"}, {"--leastNumKmer", "n"}, {"-Q", "Q"}, {"--leastQuality", "Q"}, {"-D", "D"}, {"--maxDist", "D"},
        {"-M", "M"}, {"--metric", "M"}, {"-N", "N"}, {"--neighborN_max", "N"}, {"-r", "r"}, {"--reference", "r"},
        {"-F", "F"}, {"--Fined", "F"}, {"--device", "device"}, {"--query", "q"}, {"-q", "q"},
        {"--reverse", "reverse"}, {"--rs", "rs"}, {"--qs", "qs"}, {"--gpus", "gpus"}, {"--same-device", "same-device"}};
    if (sub == "_parse") return cmd_parse(argc, argv);
    if (sub == "_format") return cmd_format(argc, argv);
    if (sub == "_layout") {  // test helper: the on-disk structs of this tool, in the format of `ref_driver layout`
        printf("sketchInfo_t %zu %zu %zu %zu %zu %zu\n", sizeof(SketchInfo), offsetof(SketchInfo, id), offsetof(SketchInfo, half_k),
               offsetof(SketchInfo, half_subk), offsetof(SketchInfo, drlevel), offsetof(SketchInfo, genomeNumber));
        printf("dim_shuffle_stat_t %zu %zu %zu %zu %zu\n", sizeof(ShufHeader), offsetof(ShufHeader, id), offsetof(ShufHeader, k),
               offsetof(ShufHeader, subk), offsetof(ShufHeader, drlevel));
        printf("co_dstat_t %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(CoDstat), offsetof(CoDstat, shuf_id), offsetof(CoDstat, koc),
               offsetof(CoDstat, kmerlen), offsetof(CoDstat, dim_rd_len), offsetof(CoDstat, comp_num), offsetof(CoDstat, infile_num),
               offsetof(CoDstat, all_ctx_ct));
        return 0;
    }
    if (sub == "shuffle") { cerr << "-----run the subcommand: shuffle" << endl; return cmd_shuffle(parse_args(argc, argv, 2, alias, {})); }
    // A distance run over .sketch files moves ~50 MB to the device and a few MB back, once: the runtime's DMA engines cost more to
    // set up (their queues: ~10 ms before the first upload, ~10 ms before the first read-back -- `index built` 34 -> 16 ms,
    // `distances` 14.6 -> 3.7 ms of the stamps of RK_TIMING) than blit kernels need to copy it.  Sketching from FASTA lists keeps
    // them: there gigabytes of uploads run beside the scan kernel.  (Set HSA_ENABLE_SDMA yourself to overrule.)
    if (sub == "alldist" || sub == "dist") {
        bool from_sketches = true;
        for (int i = 2; i + 1 < argc; i++) {
            const string f = argv[i];
            if (f == "-i" || f == "--input" || f == "-r" || f == "--reference" || f == "-q" || f == "--query") from_sketches = from_sketches && is_sketch_file(argv[i + 1]);
        }
        if (from_sketches) setenv("HSA_ENABLE_SDMA", "0", 0);
    }
    // the GPU subcommands leave through _exit once their output is on disk (see Gpu::~Gpu)
    auto leave = [](int rc) -> int {
        stamp("done");
        fflush(stdout);
        fflush(stderr);
        _exit(rc);
        return rc;
    };
    if (sub == "sketch") { cerr << "-----run the subcommand: sketch" << endl; return leave(cmd_sketch(parse_args(argc, argv, 2, alias, {"q"}))); }
    if (sub == "alldist") { cerr << "-----run the subcommand: alldist" << endl; return leave(cmd_alldist(parse_args(argc, argv, 2, alias, {"same-device"}))); }
    if (sub == "dist") { cerr << "-----run the subcommand: dist" << endl; return leave(cmd_dist(parse_args(argc, argv, 2, alias, {"same-device"}))); }
    if (sub == "info") { cerr << "-----run the subcommand: info" << endl; return cmd_info(parse_args(argc, argv, 2, alias, {"F"})); }
    if (sub == "merge") { cerr << "-----run the subcommand: merge" << endl; return cmd_merge(parse_args(argc, argv, 2, alias, {})); }
    if (sub == "convert") { cerr << "-----run the subcommand: convert" << endl; return cmd_convert(parse_args(argc, argv, 2, alias, {"q", "reverse"})); }
    if (sub == "union") { cerr << "-----run the subcommand: union" << endl; return cmd_union(parse_args(argc, argv, 2, alias, {})); }
    if (sub == "sub") { cerr << "-----run the subcommand: sub" << endl; return cmd_sub(parse_args(argc, argv, 2, alias, {})); }
    return usage();
}
