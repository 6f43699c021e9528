// formats.hpp -- host-side file formats and record reader of the rabbit_kssd tool
// (drop-in surface of RabbitKSSD: .shuf, .sketch, .sketch.dict, .sketch.index, FASTA/Q input).
// Written from the format description; citations are file:line into
// the reference tree.  Host code only: no arithmetic of the hot path lives here.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

namespace rkhost {

struct SketchInfo {  // sketchInfo_t, src/sketch.h:27-34 (20 bytes on disk)
    int32_t id = 0, half_k = 0, half_subk = 0, drlevel = 0, genomeNumber = 0;
};

struct SketchSet {  // in-memory form of a .sketch file
    SketchInfo info;
    std::vector<std::string> names;
    std::vector<uint32_t> hashes;    // CSR values, 32-bit layout (half_k - drlevel <= 8)
    std::vector<uint64_t> hashes64;  // CSR values, 64-bit layout (use64, src/sketch.cpp:336)
    std::vector<uint64_t> off{0};    // CSR offsets, names.size()+1
    size_t size() const { return names.size(); }
    bool wide() const { return info.half_k - info.drlevel > 8; }
    uint64_t total() const { return off.empty() ? 0 : off.back(); }
};

inline bool ends_with(const std::string &s, const std::string &suf)
{
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

// src/sketch.cpp:163-169: a file is a sketch iff the text after the last '.' is "sketch"
inline bool is_sketch_file(const std::string &f)
{
    auto p = f.find_last_of('.');
    return p != std::string::npos && f.substr(p + 1) == "sketch";
}

inline bool exist_file(const std::string &f)
{
    if (FILE *fp = fopen(f.c_str(), "r")) { fclose(fp); return true; }
    return false;
}

// ---- .shuf (src/shuffle.cpp:8-23, :25-104) -------------------------------------------
struct ShufHeader {  // dim_shuffle_stat_t, src/shuffle.h:11-17 (16 bytes on disk)
    int32_t id = 0, k = 0, subk = 0, drlevel = 0;
};
struct Shuf {
    int32_t id = 0, k = 0, subk = 0, drlevel = 0;
    std::vector<int32_t> table;
};

inline bool read_shuf(const std::string &path, Shuf &out, std::string &err)
{
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) { err = "cannot read shuffle file: " + path; return false; }
    ShufHeader hdr;
    if (fread(&hdr, sizeof(hdr), 1, fp) != 1 || hdr.subk < 1 || hdr.subk >= 8) {
        fclose(fp);
        err = "error read dim_shuffle header: " + path;
        return false;
    }
    out.id = hdr.id; out.k = hdr.k; out.subk = hdr.subk; out.drlevel = hdr.drlevel;
    const size_t n = (size_t)1 << (4 * out.subk);
    out.table.resize(n);
    const size_t r = fread(out.table.data(), 4, n, fp);
    fclose(fp);
    if (r != n) { err = "error read shuffled_dim: " + path; return false; }
    return true;
}

// Fisher-Yates driven by glibc srand()/rand(), seeds 23 then id (src/shuffle.cpp:50-54,76-104)
inline bool write_shuf(const std::string &path, int k, int subk, int drlevel, std::string &err)
{
    if (k < subk) { err = "half_k should be larger than sub_k"; return false; }       // :26
    if (subk >= 8) { err = "subk should be smaller than 8"; return false; }           // :30
    const int n = 1 << (4 * subk);
    if (n > RAND_MAX) { err = "shuffling array length must be less than RAND_MAX"; return false; }
    if ((1 << (4 * (subk - drlevel))) < 256)
        fprintf(stderr, "Warning: dimension after reduction is smaller than the suggested minimal\n");  // :36-38
    std::vector<int32_t> t(n);
    for (int i = 0; i < n; i++) t[i] = i;
    const int id = (k << 8) + (subk << 4) + drlevel;
    for (unsigned seed : {23u, (unsigned)id}) {
        srand(seed);
        for (int i = n - 1; i > 0; i--) {
            const int j = rand() % (i + 1);
            std::swap(t[i], t[j]);
        }
    }
    FILE *fp = fopen(path.c_str(), "wb");
    if (!fp) { err = "error open shuffle file " + path; return false; }
    ShufHeader hdr;
    hdr.id = id; hdr.k = k; hdr.subk = subk; hdr.drlevel = drlevel;
    fwrite(&hdr, sizeof(hdr), 1, fp);
    fwrite(t.data(), 4, (size_t)n, fp);
    fclose(fp);
    return true;
}

// ---- .sketch (src/sketch.cpp:1024-1154) -------------------------------------------------
inline bool save_sketches(const std::string &path, SketchSet &s, std::string &err)
{
    s.info.genomeNumber = (int32_t)s.size();
    s.info.id = (s.info.half_k << 8) + (s.info.half_subk << 4) + s.info.drlevel;  // :1029
    FILE *fp = fopen(path.c_str(), "wb");
    if (!fp) { err = "cannot write " + path; return false; }
    fwrite(&s.info, sizeof(SketchInfo), 1, fp);
    std::vector<int32_t> len(s.size()), cnt(s.size());
    for (size_t i = 0; i < s.size(); i++) {
        len[i] = (int32_t)s.names[i].size();
        cnt[i] = (int32_t)(s.off[i + 1] - s.off[i]);
    }
    fwrite(len.data(), 4, s.size(), fp);
    fwrite(cnt.data(), 4, s.size(), fp);
    for (size_t i = 0; i < s.size(); i++) {
        fwrite(s.names[i].data(), 1, s.names[i].size(), fp);
        if (s.wide()) fwrite(s.hashes64.data() + s.off[i], 8, (size_t)cnt[i], fp);  // :1055-1058
        else fwrite(s.hashes.data() + s.off[i], 4, (size_t)cnt[i], fp);
    }
    if (fclose(fp)) { err = "write error on " + path; return false; }
    return true;
}

inline bool read_sketches(const std::string &path, SketchSet &s, std::string &err)
{
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) { err = "cannot open the file: " + path; return false; }
    s = SketchSet();
    if (fread(&s.info, sizeof(SketchInfo), 1, fp) != 1 || s.info.genomeNumber < 0) {
        fclose(fp); err = "mismatched read sketch_info: " + path; return false;
    }
    const bool wide = s.wide();
    const size_t n = (size_t)s.info.genomeNumber;
    std::vector<int32_t> len(n), cnt(n);
    if (fread(len.data(), 4, n, fp) != n || fread(cnt.data(), 4, n, fp) != n) {
        fclose(fp); err = "mismatched read genome_name_size, hash_set_size: " + path; return false;
    }
    uint64_t tot = 0;
    for (size_t i = 0; i < n; i++) {
        if (len[i] < 0 || cnt[i] < 0) { fclose(fp); err = "corrupt sketch file: " + path; return false; }
        tot += (uint64_t)cnt[i];
    }
    s.names.resize(n);
    if (wide) s.hashes64.resize(tot);
    else s.hashes.resize(tot);
    s.off.assign(n + 1, 0);
    for (size_t i = 0; i < n; i++) {
        s.names[i].resize((size_t)len[i]);
        if (len[i] && fread(&s.names[i][0], 1, (size_t)len[i], fp) != (size_t)len[i]) {
            fclose(fp); err = "the read nameLength is not equal to the saved nameLength: " + path; return false;
        }
        const size_t got = !cnt[i] ? 0 : wide ? fread(s.hashes64.data() + s.off[i], 8, (size_t)cnt[i], fp)
                                              : fread(s.hashes.data() + s.off[i], 4, (size_t)cnt[i], fp);
        if (got != (size_t)cnt[i]) {
            fclose(fp); err = "the read hashNumber is not equal to the saved hashNumber: " + path; return false;
        }
        s.off[i + 1] = s.off[i] + (uint64_t)cnt[i];
    }
    fclose(fp);
    return true;
}

// ---- .dict / .index, 32-bit layout (src/sketch.cpp:991-1011, src/dist.cpp:86-129) -------
// `bytes` at file offset `at`, cut into pieces written concurrently (the dense .index is 1 GiB at
// 28 hash bits: the page-cache copies scale with threads)
inline bool pwrite_parallel(int fd, const void *data, uint64_t bytes, uint64_t at, int threads)
{
    const uint64_t T = (uint64_t)std::max(1, std::min(threads, (int)(bytes >> 24) + 1));
    std::vector<int> ok(T, 1);
    std::vector<std::thread> pool;
    for (uint64_t i = 0; i < T; i++)
        pool.emplace_back([&, i]() {
            uint64_t pos = bytes / T * i;
            const uint64_t end = i + 1 == T ? bytes : bytes / T * (i + 1);
            while (pos < end) {
                const ssize_t r = pwrite(fd, (const char *)data + pos, (size_t)std::min<uint64_t>(end - pos, 1u << 30), (off_t)(at + pos));
                if (r <= 0) { ok[i] = 0; break; }
                pos += (uint64_t)r;
            }
        });
    for (auto &th : pool) th.join();
    for (uint64_t i = 0; i < T; i++) if (!ok[i]) return false;
    return true;
}

inline bool write_index(const std::string &dict, const std::string &index, const uint32_t *postings, uint64_t total,
                        const uint32_t *counts, uint64_t hash_size, std::string &err, int threads = 8)
{
    FILE *fd = fopen(dict.c_str(), "wb");
    if (!fd) { err = "cannot write " + dict; return false; }
    fwrite(postings, 4, total, fd);
    if (fclose(fd)) { err = "write error on " + dict; return false; }
    // no O_TRUNC: an existing .index of the same parameters has the same size, and overwriting its
    // pages in place is twice as fast as freeing 1 GiB of page cache first
    const int fi = open(index.c_str(), O_WRONLY | O_CREAT, 0644);
    if (fi < 0) { err = "cannot write " + index; return false; }
    const uint64_t head[2] = {hash_size, total};
    bool good = pwrite(fi, head, 16, 0) == 16 && pwrite_parallel(fi, counts, hash_size * 4, 16, threads) &&
                ftruncate(fi, (off_t)(16 + hash_size * 4)) == 0;
    if (close(fi)) good = false;
    if (!good) { err = "write error on " + index; return false; }
    return true;
}
// The dense .index from the index's own sparse form (distinct hashes ascending + list lengths): no 4 * 2^bits byte array
// crosses PCIe (1 GiB at 28 hash bits for 3 M distinct hashes).  Every thread owns a stretch of the hash space: it
// scatters that stretch's counts into an anonymous zero mapping and writes the stretch to its place in the file.
// (Scattering into a mapping of the file itself measured slower: 0.36-0.59 s against 0.28 s for the dense export.)
inline bool write_index_lists(const std::string &dict, const std::string &index, const uint32_t *postings, uint64_t total,
                              const uint32_t *hashes, const uint32_t *counts, uint64_t n_lists, uint64_t hash_size, std::string &err,
                              int threads = 8)
{
    FILE *fd = fopen(dict.c_str(), "wb");
    if (!fd) { err = "cannot write " + dict; return false; }
    fwrite(postings, 4, total, fd);
    if (fclose(fd)) { err = "write error on " + dict; return false; }
    // no O_TRUNC: an existing .index of the same parameters has the same size, and overwriting its pages in place is
    // twice as fast as freeing 1 GiB of page cache first
    const int fi = open(index.c_str(), O_WRONLY | O_CREAT, 0644);
    if (fi < 0) { err = "cannot write " + index; return false; }
    void *m = mmap(nullptr, hash_size * 4, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) { close(fi); err = "out of memory writing " + index; return false; }
    uint32_t *dense = (uint32_t *)m;
    const uint64_t T = (uint64_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)threads, (hash_size >> 22) + 1));
    std::vector<int> ok(T, 1);
    std::vector<std::thread> pool;
    for (uint64_t t = 0; t < T; t++)
        pool.emplace_back([&, t]() {
            const uint64_t lo = hash_size * t / T, hi = hash_size * (t + 1) / T;
            const uint32_t *a = std::lower_bound(hashes, hashes + n_lists, lo, [](uint32_t h, uint64_t v) { return (uint64_t)h < v; });
            const uint32_t *b = std::lower_bound(hashes, hashes + n_lists, hi, [](uint32_t h, uint64_t v) { return (uint64_t)h < v; });
            for (const uint32_t *p = a; p < b; p++) dense[*p] = counts[p - hashes];
            uint64_t pos = lo * 4;
            while (pos < hi * 4) {
                const ssize_t r = pwrite(fi, (const char *)dense + pos, (size_t)std::min<uint64_t>(hi * 4 - pos, 1u << 30), (off_t)(16 + pos));
                if (r <= 0) { ok[t] = 0; break; }
                pos += (uint64_t)r;
            }
        });
    for (auto &th : pool) th.join();
    const uint64_t head[2] = {hash_size, total};
    bool good = pwrite(fi, head, 16, 0) == 16 && ftruncate(fi, (off_t)(16 + hash_size * 4)) == 0;
    for (uint64_t t = 0; t < T; t++) good = good && ok[t];
    munmap(m, hash_size * 4);
    if (close(fi)) good = false;
    if (!good) { err = "write error on " + index; return false; }
    return true;
}

inline bool write_index(const std::string &dict, const std::string &index, const std::vector<uint32_t> &postings,
                        const std::vector<uint32_t> &counts, std::string &err)
{
    return write_index(dict, index, postings.data(), postings.size(), counts.data(), counts.size(), err);
}

// 64-bit layout: .index = {u64 n; u64 hash[n]; u32 count[n]}, .dict = posting blocks in that order
// (src/sketch.cpp:942-963)
inline bool write_index64(const std::string &dict, const std::string &index, const std::vector<uint32_t> &postings,
                          const std::vector<uint64_t> &hashes, const std::vector<uint32_t> &counts, std::string &err)
{
    FILE *fd = fopen(dict.c_str(), "wb");
    if (!fd) { err = "cannot write " + dict; return false; }
    fwrite(postings.data(), 4, postings.size(), fd);
    if (fclose(fd)) { err = "write error on " + dict; return false; }
    FILE *fi = fopen(index.c_str(), "wb");
    if (!fi) { err = "cannot write " + index; return false; }
    const uint64_t n = hashes.size();
    fwrite(&n, 8, 1, fi);
    fwrite(hashes.data(), 8, hashes.size(), fi);
    fwrite(counts.data(), 4, counts.size(), fi);
    if (fclose(fi)) { err = "write error on " + index; return false; }
    return true;
}

inline bool read_index(const std::string &dict, const std::string &index, std::vector<uint32_t> &postings,
                       std::vector<uint32_t> &counts, std::string &err)
{
    FILE *fi = fopen(index.c_str(), "rb");
    if (!fi) { err = "cannot open the index sketch file: " + index; return false; }
    uint64_t hash_size = 0, total = 0;
    if (fread(&hash_size, 8, 1, fi) != 1 || fread(&total, 8, 1, fi) != 1 || hash_size > (1ULL << 32)) {
        fclose(fi); err = "error read hash_size, total_index: " + index; return false;
    }
    counts.resize(hash_size);
    if (fread(counts.data(), 4, hash_size, fi) != hash_size) { fclose(fi); err = "error read sketch_size_arr: " + index; return false; }
    fclose(fi);
    FILE *fd = fopen(dict.c_str(), "rb");
    if (!fd) { err = "cannot open the dictionary sketch file: " + dict; return false; }
    postings.resize(total);
    if (fread(postings.data(), 4, total, fd) != total) { fclose(fd); err = "error read index_arr: " + dict; return false; }
    fclose(fd);
    return true;
}

// ---- FASTA/FASTQ record reader with kseq_read semantics (src/kseq.h:176-215) -------------
// One control flow, two sinks:
//  * vectors: sequence bytes of every record appended to `seq` (newlines dropped, a trailing
//    '\r' of a line dropped once the record holds more than one byte), rec_off gets one more
//    entry per record, optional quality characters;
//  * packed: the same bytes written straight into a caller buffer in the layout
//    rk_sketch_packed_dev expects (records separated by one 0x00 byte), which is how the
//    parser threads of the sketch pipeline fill the page-locked staging buffer without an
//    intermediate copy.  The FASTQ quality gate (src/sketch.cpp:785) is applied in place.
// growable byte buffer that does not zero-fill (a 3 GB std::vector costs a 3 GB memset)
struct RawBuf {
    uint8_t *p = nullptr;
    size_t cap = 0;
    RawBuf() = default;
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    ~RawBuf() { free(p); }
    uint8_t *data() { return p; }
    size_t size() const { return cap; }
    void resize(size_t n)
    {
        if (n <= cap) return;
        uint8_t *q = (uint8_t *)realloc(p, n);
        if (!q) { fprintf(stderr, "ERROR: out of memory reading a file (%zu bytes)\n", n); exit(1); }
        p = q;
        cap = n;
    }
};

class RecordReader {
  public:
    // whole file into buf (plain or gzip'd, src/sketch.cpp:462 reads both through gzopen);
    // buf is caller-owned so that a parser thread reuses one allocation for all its files
    template <class Buf> static bool slurp(const std::string &path, Buf &buf, size_t &n)
    {
        n = 0;
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) return false;
        unsigned char magic[2] = {0, 0};
        const size_t got = fread(magic, 1, 2, f);
        if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {  // gzip
            fclose(f);
            gzFile fp = gzopen(path.c_str(), "r");
            if (!fp) return false;
            gzbuffer(fp, 1 << 20);
            for (;;) {
                if (buf.size() - n < (1u << 20)) buf.resize(buf.size() < (1u << 22) ? (1u << 22) : buf.size() * 2);
                const int r = gzread(fp, buf.data() + n, (unsigned)std::min<size_t>(buf.size() - n, 1u << 30));
                if (r <= 0) break;
                n += (size_t)r;
            }
            gzclose(fp);
            return true;
        }
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (sz > 0) {
            if (buf.size() < (size_t)sz) buf.resize((size_t)sz);
            n = fread(buf.data(), 1, (size_t)sz, f);
        }
        fclose(f);
        return true;
    }

    // qual (optional): one quality character per base; '~' for records without qualities
    static bool read_file(const std::string &path, std::vector<uint8_t> &seq, std::vector<uint64_t> &rec_off,
                          std::vector<uint8_t> *qual = nullptr)
    {
        std::vector<uint8_t> buf;
        size_t n = 0;
        if (!slurp(path, buf, n)) return false;
        parse(buf.data(), n, seq, rec_off, qual);
        return true;
    }

    static void parse(const uint8_t *b, size_t n, std::vector<uint8_t> &seq, std::vector<uint64_t> &rec_off,
                      std::vector<uint8_t> *qual = nullptr)
    {
        if (rec_off.empty()) rec_off.push_back(seq.size());
        VecSink sink{seq, rec_off, qual};
        parse_core(b, n, sink);
    }

    struct Packed {
        uint64_t bytes = 0;   // bytes written: records and their separators, without the last separator
        uint64_t n_rec = 0;
        bool overflow = false;  // dst was too small: nothing usable was written
    };
    // least_qual: bases of FASTQ records whose quality character is below it become 0x00
    // (pass a value <= 0 to keep everything).  dst needs at most n bytes for any input of n bytes.
    static Packed parse_packed(const uint8_t *b, size_t n, uint8_t *dst, size_t cap, int least_qual = 0)
    {
        PackedSink sink{dst, cap, least_qual};
        parse_core(b, n, sink);
        Packed r;
        r.n_rec = sink.n_rec;
        r.overflow = sink.overflow;
        r.bytes = sink.n_rec ? sink.w - 1 : 0;
        return r;
    }

    // Big plain FASTA files (one 3 Gb genome is a single list entry): the buffer is cut into
    // `threads` pieces at line starts -- kseq looks for '>' only at the start of a line
    // (src/kseq.h:192-196), so a piece needs no context -- each piece counts its packed bytes, a
    // prefix sum places it, and the pieces are copied concurrently.  Returns false, leaving dst
    // undefined, for inputs that need the serial parser: FASTQ ('@' / '+' lines), '\r' line ends,
    // small buffers.
    static bool parse_packed_parallel(const uint8_t *b, size_t n, uint8_t *dst, size_t cap, int threads, Packed &out,
                                      size_t min_bytes = 32u << 20)
    {
        if (threads < 2 || n < min_bytes) return false;
        size_t start = 0;
        while (start < n && b[start] != '>' && b[start] != '@') start++;  // :180-184
        if (start >= n || b[start] == '@') return false;
        const size_t T = (size_t)threads;
        std::vector<size_t> cut(T + 1);
        cut[0] = start;
        cut[T] = n;
        for (size_t i = 1; i < T; i++) {
            size_t p = start + (n - start) / T * i;
            if (p < cut[i - 1]) p = cut[i - 1];
            const uint8_t *e = line_end(b + p, b + n);
            cut[i] = e < b + n ? (size_t)(e - b) + 1 : n;
        }
        struct Piece { uint64_t bytes = 0, headers = 0; bool bad = false; };
        std::vector<Piece> pc(T);
        // what a piece contributes: every header line starts a record (a separator, except for the
        // file's first), every other non-empty line its characters
        auto walk = [&](size_t i, uint8_t *to) {
            Piece &q = pc[i];
            size_t pos = cut[i];
            const size_t end = cut[i + 1];
            uint8_t *w = to;
            uint64_t bytes = 0, headers = 0;
            while (pos < end) {
                const uint8_t c = b[pos];
                const size_t e = (size_t)(line_end(b + pos, b + end) - b);
                if (c == '>') {
                    if (pos != start) { if (w) *w++ = 0; bytes++; }
                    headers++;
                } else if (c == '+' || c == '@') {
                    q.bad = true;
                    return;
                } else if (e > pos) {
                    if (b[e - 1] == '\r') { q.bad = true; return; }
                    if (w) { memcpy(w, b + pos, e - pos); w += e - pos; }
                    bytes += e - pos;
                }
                pos = e + 1;
            }
            q.bytes = bytes;
            q.headers = headers;
        };
        auto run = [&](uint8_t *const *targets) {
            std::vector<std::thread> pool;
            for (size_t i = 0; i < T; i++) pool.emplace_back([&, i]() { walk(i, targets ? targets[i] : nullptr); });
            for (auto &th : pool) th.join();
        };
        run(nullptr);
        uint64_t total = 0, headers = 0;
        std::vector<uint8_t *> targets(T);
        for (size_t i = 0; i < T; i++) {
            if (pc[i].bad) return false;
            targets[i] = dst + total;
            total += pc[i].bytes;
            headers += pc[i].headers;
        }
        if (total > cap) { out = Packed(); out.overflow = true; return true; }
        run(targets.data());
        out = Packed();
        out.bytes = total;
        out.n_rec = headers;
        return true;
    }

    // whole plain file with `threads` concurrent preads (page-cache copies scale with threads)
    template <class Buf> static bool slurp_parallel(const std::string &path, Buf &buf, size_t &n, int threads)
    {
        n = 0;
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) || st.st_size <= 0) { close(fd); return st.st_size == 0; }
        const size_t sz = (size_t)st.st_size;
        if (buf.size() < sz) buf.resize(sz);
        const size_t T = (size_t)std::max(1, threads);
        std::vector<std::thread> pool;
        std::vector<int> ok(T, 1);
        for (size_t i = 0; i < T; i++)
            pool.emplace_back([&, i]() {
                size_t pos = sz / T * i;
                const size_t end = i + 1 == T ? sz : sz / T * (i + 1);
                while (pos < end) {
                    const ssize_t r = pread(fd, buf.data() + pos, end - pos, (off_t)pos);
                    if (r <= 0) { ok[i] = 0; break; }
                    pos += (size_t)r;
                }
            });
        for (auto &th : pool) th.join();
        close(fd);
        for (size_t i = 0; i < T; i++) if (!ok[i]) return false;
        n = sz;
        return true;
    }

  private:
    struct VecSink {
        std::vector<uint8_t> &seq;
        std::vector<uint64_t> &rec_off;
        std::vector<uint8_t> *qual;
        size_t size() const { return seq.size(); }
        void push(uint8_t c) { seq.push_back(c); }
        void append(const uint8_t *x, const uint8_t *y) { seq.insert(seq.end(), x, y); }
        uint8_t back() const { return seq.back(); }
        void pop() { seq.pop_back(); }
        void truncate(size_t start) { seq.resize(start); }
        void end_fasta() {
            if (qual) qual->resize(seq.size(), '~');
            rec_off.push_back(seq.size());
        }
        void end_fastq(size_t start, const std::vector<uint8_t> &q) {
            if (qual) { qual->resize(start, '~'); qual->insert(qual->end(), q.begin(), q.end()); }
            rec_off.push_back(seq.size());
        }
    };
    struct PackedSink {
        uint8_t *dst;
        size_t cap;
        int least_qual;
        size_t w = 0;
        uint64_t n_rec = 0;
        bool overflow = false;
        size_t size() const { return w; }
        void push(uint8_t c) { if (w < cap) dst[w] = c; else overflow = true; w++; }
        void append(const uint8_t *x, const uint8_t *y) {
            const size_t len = (size_t)(y - x);
            if (w + len <= cap) memcpy(dst + w, x, len); else overflow = true;
            w += len;
        }
        uint8_t back() const { return w <= cap ? dst[w - 1] : 0; }
        void pop() { w--; }
        void truncate(size_t start) { w = start; }
        void separator() { if (w < cap) dst[w] = 0; else overflow = true; w++; n_rec++; }
        void end_fasta() { separator(); }
        void end_fastq(size_t start, const std::vector<uint8_t> &q) {
            if (least_qual > 0 && w <= cap)
                for (size_t i = 0; i < q.size(); i++)
                    if ((int)(char)q[i] < least_qual) dst[start + i] = 0;
            separator();
        }
    };

    static const uint8_t *line_end(const uint8_t *p, const uint8_t *end)
    {
        const void *e = memchr(p, '\n', (size_t)(end - p));
        return e ? (const uint8_t *)e : end;
    }

    template <class Sink> static void parse_core(const uint8_t *b, size_t n, Sink &out)
    {
        size_t pos = 0;
        int last_char = 0;
        std::vector<uint8_t> q;
        for (;;) {
            if (last_char == 0) {  // jump to the next header line, :180-184
                while (pos < n && b[pos] != '>' && b[pos] != '@') pos++;
                if (pos >= n) return;
                last_char = b[pos++];
            }
            if (pos >= n) return;  // name read fails at EOF, :186
            int c = 0;
            while (pos < n && !is_space(b[pos])) pos++;  // name
            if (pos < n) c = b[pos++];
            if (c != '\n') {  // comment up to end of line, :187
                pos = (size_t)(line_end(b + pos, b + n) - b);
                if (pos < n) pos++;
            }
            const size_t start = out.size();
            int stop = -1;
            while (pos < n) {  // :192-196
                const int ch = b[pos++];
                if (ch == '>' || ch == '+' || ch == '@') { stop = ch; break; }
                if (ch == '\n') continue;
                out.push((uint8_t)ch);
                if (pos >= n) break;  // ks_getuntil2 at EOF returns before the '\r' rule
                const size_t e = (size_t)(line_end(b + pos, b + n) - b);
                out.append(b + pos, b + e);
                pos = e < n ? e + 1 : n;
                if (out.size() - start > 1 && out.back() == '\r') out.pop();  // :140
            }
            if (stop == '>' || stop == '@') last_char = stop;  // :197
            if (stop != '+') {  // FASTA record, :204
                out.end_fasta();
                if (stop < 0 && pos >= n) return;  // EOF: the next call finds no name and ends the file
                continue;
            }
            // FASTQ: skip the rest of the '+' line, then read quality lines, :209-214
            pos = (size_t)(line_end(b + pos, b + n) - b);
            if (pos >= n) { out.truncate(start); return; }  // -2: no quality string
            pos++;
            const size_t want = out.size() - start;
            q.clear();
            while (pos < n && q.size() < want) {  // quality lines appended like sequence lines, :211
                const size_t e = (size_t)(line_end(b + pos, b + n) - b);
                q.insert(q.end(), b + pos, b + e);
                pos = e < n ? e + 1 : n;
                if (q.size() > 1 && q.back() == '\r') q.pop_back();
            }
            last_char = 0;
            if (q.size() != want) { out.truncate(start); return; }  // -2: quality string of a different length
            out.end_fastq(start, q);
        }
    }

    static bool is_space(int c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
};

// first byte of a file (list-type sniffing, src/sketch.cpp:52-94)
inline int first_byte(const std::string &path)
{
    std::ifstream ifs(path);
    std::string line;
    std::getline(ifs, line);
    return line.empty() ? -1 : (unsigned char)line[0];
}

inline std::vector<std::string> read_list(const std::string &path)
{
    std::vector<std::string> v;
    std::ifstream ifs(path);
    std::string line;
    while (std::getline(ifs, line)) v.push_back(line);
    return v;
}

}  // namespace rkhost
