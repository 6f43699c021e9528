// integration/gpu_backend.cpp -- the reference-side binding of librabbitkssd.so, as a maintainer of
// RabbitKSSD would add it to src/ (see INTEGRATION.md).  It is written against the REFERENCE's own
// headers (sketch.h, dist.h: vector<sketch_t>, sketchInfo_t, the signatures of src/dist.h:35-37) and
// replaces the bodies of index_tridist / index_dist with calls into the C ABI of include/rabbitkssd.h.
//
// It is not part of the product (the product's host tool is rabbitkssd_amd/host/rabbit_kssd.cpp, which
// needs no reference header).  It is compiled only by `make -C oracle ref_gpu`, against the headers where
// they lie under /root/reference/src, into oracle/_ref/ref_driver_gpu -- the reference's data structures
// and text conventions on top of the GPU library -- and tests/test_reference_binding.py checks on an
// MI355X that this binary reproduces the golden text the REAL index_tridist / index_dist wrote.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "dist.h"        // reference: index_tridist / index_dist signatures, includes sketch.h
#include "rabbitkssd.h"  // this repository: the C ABI

using std::string;
using std::vector;

static rk_ctx *g_ctx = nullptr;
static void gpu_init()
{
    if (!g_ctx && rk_ctx_create(0, &g_ctx) != RK_OK) {
        fprintf(stderr, "ERROR: no MI355X available\n");
        exit(1);
    }
}
static void gpu_check(int rc, const char *what)
{
    if (rc) {  // the reference's convention: message on stderr, exit(1)
        fprintf(stderr, "ERROR: %s: %s\n", what, rk_last_error(g_ctx));
        exit(1);
    }
}

// vector<sketch_t> -> CSR on the device: the only marshalling the ABI needs.  use64 selects the member
// that holds the hashes exactly like the reference does (src/sketch.cpp:336, src/dist.cpp:181,566).
static rk_sketches *upload(const vector<sketch_t> &s, bool use64)
{
    vector<uint64_t> off(1, 0);
    rk_sketches *sk = nullptr;
    if (use64) {
        vector<uint64_t> h;
        for (const sketch_t &x : s) {
            h.insert(h.end(), x.hashSet64.begin(), x.hashSet64.end());
            off.push_back(h.size());
        }
        gpu_check(rk_sketches_from_host64(g_ctx, h.data(), off.data(), (uint32_t)s.size(), &sk), "rk_sketches_from_host64");
    } else {
        vector<uint32_t> h;
        for (const sketch_t &x : s) {
            h.insert(h.end(), x.hashSet.begin(), x.hashSet.end());
            off.push_back(h.size());
        }
        gpu_check(rk_sketches_from_host(g_ctx, h.data(), off.data(), (uint32_t)s.size(), &sk), "rk_sketches_from_host");
    }
    return sk;
}

static void write_text(const string &path, const rk_hit *hits, uint64_t n, const vector<sketch_t> &rows,
                       const vector<sketch_t> &cols, bool alldist)
{
    FILE *fp = fopen(path.c_str(), "w");
    if (!fp) { fprintf(stderr, "ERROR: cannot write %s\n", path.c_str()); exit(1); }
    fprintf(fp, " genome0\tgenome1\tcommon|size0|size1\tjaccard\tmashD\n");  // src/dist.cpp:291,:722
    char line[8192];
    for (uint64_t i = 0; i < n; i++) {
        // alldist prints (col, row), dist prints (query row, reference col): src/dist.cpp:233 / :642
        const string &a = alldist ? cols[hits[i].col].fileName : rows[hits[i].row].fileName;
        const string &b = alldist ? rows[hits[i].row].fileName : cols[hits[i].col].fileName;
        rk_format_hit(line, sizeof line, a.c_str(), b.c_str(), &hits[i]);
        fputs(line, fp);
    }
    fclose(fp);
}

// same signature as src/dist.h:35 (refSketchOut and numThreads are unused: no .dict/.index round trip,
// no host threads)
void index_tridist_gpu(vector<sketch_t> &sketches, sketchInfo_t &info, string refSketchOut, string outputFile,
                       int kmer_size, double maxDist, int isContainment, int numThreads)
{
    (void)refSketchOut;
    (void)numThreads;
    gpu_init();
    const bool use64 = info.half_k - info.drlevel > 8;
    rk_sketches *sk = upload(sketches, use64);
    rk_index *idx = nullptr;
    gpu_check(rk_index_build(g_ctx, sk, 4 * (info.half_k - info.drlevel), &idx), "rk_index_build");
    rk_dist_opts o = {};
    o.triangle = 1;
    o.metric = isContainment;
    o.kmer_size = kmer_size;
    o.max_dist = maxDist;
    rk_hit *hits = nullptr;
    uint64_t n = 0;
    gpu_check(rk_dist_rows(g_ctx, idx, nullptr, &o, &hits, &n, nullptr), "rk_dist_rows");
    write_text(outputFile, hits, n, sketches, sketches, true);
    rk_free_host(hits);
    rk_index_free(idx);
    rk_sketches_free(sk);
}

// same signature as src/dist.h:37
void index_dist_gpu(vector<sketch_t> &ref_sketches, sketchInfo_t &ref_info, string refSketchOut,
                    vector<sketch_t> &query_sketches, string outputFile, int kmer_size, double maxDist,
                    uint64_t maxNeighbor, bool isNeighbor, int isContainment, int numThreads)
{
    (void)refSketchOut;
    (void)numThreads;
    gpu_init();
    const bool use64 = ref_info.half_k - ref_info.drlevel > 8;
    rk_sketches *rs = upload(ref_sketches, use64), *qs = upload(query_sketches, use64);
    rk_index *idx = nullptr;
    gpu_check(rk_index_build(g_ctx, rs, 4 * (ref_info.half_k - ref_info.drlevel), &idx), "rk_index_build");
    rk_dist_opts o = {};
    o.triangle = 0;
    o.metric = isContainment;
    o.kmer_size = kmer_size;
    o.max_dist = maxDist;
    rk_hit *hits = nullptr;
    uint64_t n = 0;
    gpu_check(rk_dist_rows(g_ctx, idx, qs, &o, &hits, &n, nullptr), "rk_dist_rows");
    if (isNeighbor) gpu_check(rk_topn_rows(hits, &n, maxNeighbor), "rk_topn_rows");  // -N, src/dist.cpp:599,625-640
    write_text(outputFile, hits, n, query_sketches, ref_sketches, false);
    rk_free_host(hits);
    rk_index_free(idx);
    rk_sketches_free(qs);
    rk_sketches_free(rs);
}
