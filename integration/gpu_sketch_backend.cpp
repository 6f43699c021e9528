// integration/gpu_sketch_backend.cpp -- the reference-side binding of librabbitkssd.so for the SKETCH half of the
// boundary, as a maintainer of RabbitKSSD would add it to src/ (see INTEGRATION.md): sketchFastaFile and transSketches
// (src/sketch.h:62,66; call sites src/subCommand.cpp:55,177,243, src/main.cpp:203, src/sketch.cpp:583) with their bodies
// replaced by calls into the C ABI of include/rabbitkssd.h.  Written against the REFERENCE's own headers (sketch.h:
// sketch_t, sketchInfo_t; common.h: kssd_parameter_t; kseq.h: the record reader of src/sketch.cpp:462-485).
//
// Not part of the product (the product's host tool is rabbitkssd_amd/host/rabbit_kssd.cpp).  Compiled only by
// `make -C oracle ref_sketch_gpu`, against the headers where they lie under /root/reference/src, into
// oracle/_ref/ref_sketch_driver_gpu: the reference's file reader, data structures, saveSketches and on-disk
// conventions on top of the GPU library.  tests/test_reference_binding.py checks on an MI355X that this binary
// reproduces the hash sets the REAL sketchFastaFile computed (tests/golden/sketch_ref/expected.json) and the
// .dict / .index bytes the REAL transSketches wrote.
#include <err.h>
#include <errno.h>
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "common.h"      // reference: kssd_parameter_t
#include "kseq.h"        // reference: the FASTA/FASTQ record reader
#include "sketch.h"      // reference: sketch_t, sketchInfo_t, saveSketches, isSketchFile
#include "rabbitkssd.h"  // this repository: the C ABI

KSEQ_INIT(gzFile, gzread);

using std::string;
using std::vector;

static rk_ctx *g_sctx = nullptr;
static void sgpu_init()
{
    if (!g_sctx && rk_ctx_create(0, &g_sctx) != RK_OK) {
        fprintf(stderr, "ERROR: no MI355X available\n");
        exit(1);
    }
}
static void sgpu_check(int rc, const char *what)
{
    if (rc) {  // the reference's convention: message on stderr, exit(1)
        fprintf(stderr, "ERROR: %s: %s\n", what, rk_last_error(g_sctx));
        exit(1);
    }
}

void transSketches_gpu(vector<sketch_t> &sketches, sketchInfo_t &info, string dictFile, string indexFile, int numThreads);

// same signature as src/sketch.h:62.  The reference's small-file loop (src/sketch.cpp:455-566) with the k-mer arithmetic
// (:487-530), the unordered_set and the OpenMP loop replaced by ONE rk_sketch_batch call over all files; the big-file
// branch (:380-450) needs no twin here -- the library cuts genomes of any size into chunks by itself.
bool sketchFastaFile_gpu(string inputFile, bool isQuery, int numThreads, kssd_parameter_t parameter, vector<sketch_t> &sketches,
                         sketchInfo_t &info, string outputFile)
{
    (void)numThreads;
    sgpu_init();
    const int half_k = parameter.half_k, half_subk = parameter.half_subk, drlevel = parameter.drlevel;
    const bool use64 = half_k - drlevel > 8;   // src/sketch.cpp:336
    rk_params prm;
    sgpu_check(rk_params_init(half_k, half_subk, drlevel, &prm), "rk_params_init");
    rk_filter *flt = nullptr;
    sgpu_check(rk_filter_create(g_sctx, &prm, parameter.shuffled_dim, &flt), "rk_filter_create");

    // the list of genome files, one sketch per file (src/sketch.cpp:347-364)
    std::ifstream fs(inputFile);
    if (!fs) err(errno, "cannot open the list file: %s\n", inputFile.c_str());
    vector<string> files;
    string line;
    while (getline(fs, line)) files.push_back(line);

    // kseq, exactly as src/sketch.cpp:462-479: every record's sequence, records of a file back to back
    vector<uint8_t> seq;
    vector<uint64_t> rec_off(1, 0), genome_rec(1, 0);
    for (const string &name : files) {
        gzFile fp = gzopen(name.c_str(), "r");
        if (fp == NULL) err(errno, "cannot open the genome file: %s\n", name.c_str());
        kseq_t *ks = kseq_init(fp);
        while (1) {
            const int length = kseq_read(ks);
            if (length < 0) break;
            seq.insert(seq.end(), (const uint8_t *)ks->seq.s, (const uint8_t *)ks->seq.s + length);
            rec_off.push_back(seq.size());
        }
        genome_rec.push_back(rec_off.size() - 1);
        kseq_destroy(ks);
        gzclose(fp);
    }
    rk_sketches *sk = nullptr;
    sgpu_check(rk_sketch_batch(g_sctx, flt, seq.data(), rec_off.data(), rec_off.size() - 1, genome_rec.data(),
                               (uint32_t)files.size(), &sk), "rk_sketch_batch");
    // device CSR -> vector<sketch_t> (hashSet / hashSet64 as the reference selects them, src/sketch.cpp:533-551)
    const uint32_t n = rk_sketches_count(sk);
    vector<uint64_t> off(n + 1);
    sketches.clear();
    if (use64) {
        vector<uint64_t> h(rk_sketches_total(sk));
        sgpu_check(rk_sketches_download64(sk, h.data(), off.data()), "rk_sketches_download64");
        for (uint32_t g = 0; g < n; g++) {
            sketch_t s;
            s.fileName = files[g];
            s.id = (int)g;
            s.hashSet64.assign(h.begin() + off[g], h.begin() + off[g + 1]);
            sketches.push_back(s);
        }
    } else {
        vector<uint32_t> h(rk_sketches_total(sk));
        sgpu_check(rk_sketches_download(sk, h.data(), off.data()), "rk_sketches_download");
        for (uint32_t g = 0; g < n; g++) {
            sketch_t s;
            s.fileName = files[g];
            s.id = (int)g;
            s.hashSet.assign(h.begin() + off[g], h.begin() + off[g + 1]);
            sketches.push_back(s);
        }
    }
    rk_sketches_free(sk);
    rk_filter_free(flt);

    if (!isSketchFile(outputFile)) outputFile = outputFile + ".sketch";   // src/sketch.cpp:570-572
    info.half_k = half_k;
    info.half_subk = half_subk;
    info.drlevel = drlevel;
    info.id = (half_k << 8) + (half_subk << 4) + drlevel;
    info.genomeNumber = (int)sketches.size();
    saveSketches(sketches, info, outputFile);   // the reference's own writer
    if (!isQuery) transSketches_gpu(sketches, info, outputFile + ".dict", outputFile + ".index", numThreads);
    return true;
}

// same signature as src/sketch.h:66: the inverted index of src/sketch.cpp:894-1021 built by rk_index_build, written in
// the reference's .dict / .index layouts (32-bit :991-1011, 64-bit :942-963)
void transSketches_gpu(vector<sketch_t> &sketches, sketchInfo_t &info, string dictFile, string indexFile, int numThreads)
{
    (void)numThreads;
    sgpu_init();
    const bool use64 = info.half_k - info.drlevel > 8;
    const int bits = 4 * (info.half_k - info.drlevel);
    vector<uint64_t> off(1, 0);
    rk_sketches *sk = nullptr;
    if (use64) {
        vector<uint64_t> h;
        for (const sketch_t &x : sketches) {
            h.insert(h.end(), x.hashSet64.begin(), x.hashSet64.end());
            off.push_back(h.size());
        }
        sgpu_check(rk_sketches_from_host64(g_sctx, h.data(), off.data(), (uint32_t)sketches.size(), &sk), "rk_sketches_from_host64");
    } else {
        vector<uint32_t> h;
        for (const sketch_t &x : sketches) {
            h.insert(h.end(), x.hashSet.begin(), x.hashSet.end());
            off.push_back(h.size());
        }
        sgpu_check(rk_sketches_from_host(g_sctx, h.data(), off.data(), (uint32_t)sketches.size(), &sk), "rk_sketches_from_host");
    }
    rk_index *idx = nullptr;
    sgpu_check(rk_index_build(g_sctx, sk, bits, &idx), "rk_index_build");
    const uint64_t total = rk_index_total(idx);
    vector<uint32_t> postings(total ? total : 1);
    FILE *fd = fopen(dictFile.c_str(), "wb"), *fi = fopen(indexFile.c_str(), "wb");
    if (!fd || !fi) { fprintf(stderr, "ERROR: cannot write %s / %s\n", dictFile.c_str(), indexFile.c_str()); exit(1); }
    if (use64) {
        const uint64_t n_hash = rk_index_distinct(idx);
        vector<uint64_t> hashes(n_hash ? n_hash : 1);
        vector<uint32_t> counts(n_hash ? n_hash : 1);
        sgpu_check(rk_index_export64(idx, postings.data(), hashes.data(), counts.data()), "rk_index_export64");
        fwrite(&n_hash, sizeof(uint64_t), 1, fi);
        fwrite(hashes.data(), sizeof(uint64_t), n_hash, fi);
        fwrite(counts.data(), sizeof(uint32_t), n_hash, fi);
    } else {
        const uint64_t hashSize = 1ULL << bits;
        vector<uint32_t> counts(hashSize);
        sgpu_check(rk_index_export(idx, postings.data(), counts.data()), "rk_index_export");
        fwrite(&hashSize, sizeof(uint64_t), 1, fi);
        fwrite(&total, sizeof(uint64_t), 1, fi);
        fwrite(counts.data(), sizeof(uint32_t), hashSize, fi);
    }
    fwrite(postings.data(), sizeof(uint32_t), total, fd);
    fclose(fd);
    fclose(fi);
    rk_index_free(idx);
    rk_sketches_free(sk);
}
