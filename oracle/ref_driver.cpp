// ref_driver.cpp -- TEST INFRASTRUCTURE.  A thin command-line driver around the REAL
// reference objects that compile from their own sources (common.cpp, shuffle.cpp,
// dist.cpp under /root/reference/src; see oracle/Makefile target `ref`).  It contains no
// reference code: it only calls the reference's public functions
//   initParameter        (src/common.h:48)
//   write_shuffle_dim_file (src/shuffle.h:28)
//   index_tridist / index_dist / tri_dist (src/dist.h:35-38)
// so that the C restatement in kssd_oracle.c can be validated against them and golden
// fixtures can be generated (tests/golden/make_golden.py).
//
// sketch.cpp (sketchFastaFile/transSketches/readSketches) needs the un-vendored RabbitFX
// submodule and is NOT built; .sketch/.dict/.index inputs are produced by the oracle.
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>
#include <vector>

#include <zlib.h>

#include "common.h"
#include "dist.h"
#include "kseq.h"     // the reference's record reader, instantiated exactly like src/sketch.cpp:17
#include "shuffle.h"
#include "sketch.h"

KSEQ_INIT(gzFile, gzread)

#include "kssd_oracle.h"

static void load_sketches(const char *path, std::vector<sketch_t> &out, sketchInfo_t &info)
{
    ok_sketch_info_t oi;
    char *names = nullptr;
    uint32_t *hashes = nullptr;
    uint64_t *hashes64 = nullptr;
    uint64_t *off = nullptr;
    int rc = ok_read_sketches32(path, &oi, &names, &hashes, &off);
    if (rc == -5) rc = ok_read_sketches64(path, &oi, &names, &hashes64, &off);  // use64 layout
    if (rc) {
        fprintf(stderr, "ref_driver: cannot read %s (%d)\n", path, rc);
        exit(2);
    }
    info.id = oi.id;
    info.half_k = oi.half_k;
    info.half_subk = oi.half_subk;
    info.drlevel = oi.drlevel;
    info.genomeNumber = oi.genomeNumber;
    const char *p = names;
    for (int i = 0; i < oi.genomeNumber; i++) {
        sketch_t s;
        s.fileName = p;
        p += strlen(p) + 1;
        s.id = i;
        if (hashes64) s.hashSet64.assign(hashes64 + off[i], hashes64 + off[i + 1]);
        else s.hashSet.assign(hashes + off[i], hashes + off[i + 1]);
        out.push_back(s);
    }
    free(names);
    free(hashes);
    free(hashes64);
    free(off);
}

static int usage()
{
    fprintf(stderr,
            "usage:\n"
            "  ref_driver param K S L\n"
            "  ref_driver shuffle K S L out.shuf\n"
            "  ref_driver kseq FILE...        (records as the reference's kseq_read loop sees them)\n"
            "  ref_driver basemap             (the BaseMap table of src/common.h)\n"
            "  ref_driver layout              (sizeof/offsetof of the structs written to disk)\n"
            "  ref_driver alldist WORKDIR in.sketch OUT maxDist isContainment threads\n"
            "  ref_driver tridist WORKDIR in.sketch OUT maxDist threads\n"
            "  ref_driver dist WORKDIR ref.sketch qry.sketch OUT maxDist maxNeighbor isNeighbor "
            "isContainment threads\n"
            "(sketch paths absolute; OUT is a bare file name created inside WORKDIR;\n"
            " <in.sketch>.dict/.index must already exist)\n");
    return 2;
}

#ifdef RK_GPU_BINDING
// oracle/_ref/ref_driver_gpu: the same driver over integration/gpu_backend.cpp (the reference-side
// binding of librabbitkssd.so) instead of the reference's CPU loops
void index_tridist_gpu(vector<sketch_t> &, sketchInfo_t &, string, string, int, double, int, int);
void index_dist_gpu(vector<sketch_t> &, sketchInfo_t &, string, vector<sketch_t> &, string, int, double, uint64_t, bool,
                    int, int);
#define index_tridist index_tridist_gpu
#define index_dist index_dist_gpu
#endif

int main(int argc, char **argv)
{
    if (argc < 2) return usage();
    std::string cmd = argv[1];
    if (cmd == "param" && argc == 5) {
        int k = atoi(argv[2]), s = atoi(argv[3]), l = atoi(argv[4]);
        kssd_parameter_t p = initParameter(k, s, l, nullptr);
        printf("%d %d %d %d %d %d %d %u %lx %lx %lx %lx\n", p.half_k, p.half_subk, p.drlevel,
               p.rev_add_move, p.half_outctx_len, p.dim_start, p.dim_end, p.kmer_size,
               (unsigned long)p.domask, (unsigned long)p.tupmask, (unsigned long)p.undomask0,
               (unsigned long)p.undomask1);
        return 0;
    }
    if (cmd == "shuffle" && argc == 6) {
        dim_shuffle_stat_t st;
        st.k = atoi(argv[2]);
        st.subk = atoi(argv[3]);
        st.drlevel = atoi(argv[4]);
        st.id = 0;
        write_shuffle_dim_file(&st, argv[5]);
        return 0;
    }
    if (cmd == "layout" && argc == 2) {  // the structs the reference fwrite()s: name size offsets...
        printf("sketchInfo_t %zu %zu %zu %zu %zu %zu\n", sizeof(sketchInfo_t), offsetof(sketchInfo_t, id),
               offsetof(sketchInfo_t, half_k), offsetof(sketchInfo_t, half_subk), offsetof(sketchInfo_t, drlevel),
               offsetof(sketchInfo_t, genomeNumber));
        printf("dim_shuffle_stat_t %zu %zu %zu %zu %zu\n", sizeof(dim_shuffle_stat_t), offsetof(dim_shuffle_stat_t, id),
               offsetof(dim_shuffle_stat_t, k), offsetof(dim_shuffle_stat_t, subk), offsetof(dim_shuffle_stat_t, drlevel));
        printf("co_dstat_t %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(co_dstat_t), offsetof(co_dstat_t, shuf_id),
               offsetof(co_dstat_t, koc), offsetof(co_dstat_t, kmerlen), offsetof(co_dstat_t, dim_rd_len),
               offsetof(co_dstat_t, comp_num), offsetof(co_dstat_t, infile_num), offsetof(co_dstat_t, all_ctx_ct));
        return 0;
    }
    if (cmd == "basemap" && argc == 2) {  // the reference's base coding table, src/common.h:27-37
        for (int c = 0; c < 128; c++) printf("%d%c", BaseMap[c], c == 127 ? '\n' : ' ');
        return 0;
    }
    if (cmd == "kseq" && argc >= 3) {
        // the loop of src/sketch.cpp:462-479 (and :753-776 for FASTQ): gzopen, kseq_init, kseq_read until < 0.
        // One line per file: name, records, bases, FNV-1a of all sequence bytes, FNV-1a of all quality
        // bytes ('~' for records without qualities), then the end offset of every record.
        for (int i = 2; i < argc; i++) {
            gzFile fp = gzopen(argv[i], "r");
            if (!fp) { fprintf(stderr, "ref_driver: cannot open %s\n", argv[i]); return 2; }
            kseq_t *ks = kseq_init(fp);
            unsigned long long h = 1469598103934665603ULL, hq = 1469598103934665603ULL, total = 0;
            std::vector<unsigned long long> ends;
            for (;;) {
                const int length = kseq_read(ks);
                if (length < 0) break;
                for (int t = 0; t < length; t++) {
                    h ^= (unsigned char)ks->seq.s[t];
                    h *= 1099511628211ULL;
                    const unsigned char q = ks->qual.l ? (unsigned char)ks->qual.s[t] : (unsigned char)'~';
                    hq ^= q;
                    hq *= 1099511628211ULL;
                }
                total += (unsigned long long)length;
                ends.push_back(total);
            }
            kseq_destroy(ks);
            gzclose(fp);
            printf("%s\t%zu\t%llu\t%016llx\t%016llx", argv[i], ends.size(), total, h, hq);
            for (unsigned long long e : ends) printf("\t%llu", e);
            printf("\n");
        }
        return 0;
    }
    if (cmd == "alldist" && argc == 8) {
        if (chdir(argv[2])) return 3;
        std::vector<sketch_t> sk;
        sketchInfo_t info;
        load_sketches(argv[3], sk, info);
        index_tridist(sk, info, argv[3], argv[4], 2 * info.half_k, atof(argv[5]), atoi(argv[6]),
                      atoi(argv[7]));
        return 0;
    }
#ifndef RK_GPU_BINDING
    if (cmd == "tridist" && argc == 7) {
        if (chdir(argv[2])) return 3;
        std::vector<sketch_t> sk;
        sketchInfo_t info;
        load_sketches(argv[3], sk, info);
        tri_dist(sk, argv[4], 2 * info.half_k, atof(argv[5]), atoi(argv[6]));
        return 0;
    }
#endif
    if (cmd == "dist" && argc == 11) {
        if (chdir(argv[2])) return 3;
        std::vector<sketch_t> rs, qs;
        sketchInfo_t ri, qi;
        load_sketches(argv[3], rs, ri);
        load_sketches(argv[4], qs, qi);
        index_dist(rs, ri, argv[3], qs, argv[5], 2 * ri.half_k, atof(argv[6]),
                   strtoull(argv[7], nullptr, 10), atoi(argv[8]) != 0, atoi(argv[9]),
                   atoi(argv[10]));
        return 0;
    }
    return usage();
}
