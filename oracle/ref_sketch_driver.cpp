// ref_sketch_driver.cpp -- TEST INFRASTRUCTURE.  Command-line driver around the reference's own
// sketch functions, linked against oracle/_ref/ref_sketch.o (see oracle/Makefile target `ref_sketch`:
// the lines of /root/reference/src/sketch.cpp that do not need the absent RabbitFX submodule, compiled
// unmodified from where they lie).  This file contains no reference code; it only calls
//   read_shuffle_dim / initParameter             (src/shuffle.h, src/common.h:48)
//   sketchFastaFile / sketchFastqFile            (src/sketch.h:62-63; small-file path :455-566 / :741-866)
//   saveSketches / readSketches / transSketches  (src/sketch.h:64-66)
//   convertSketch / convert_from_RabbitKSSDSketch_to_KssdSketch (src/sketch.h:69-70)
// so that the C restatement (kssd_oracle.c) and the HIP path can be checked against what the reference
// itself computes (tests/golden/make_sketch_golden.py).
//
// Every input file of a sketch run must be <= totalSize/numThreads bytes (src/sketch.cpp:366-374), else the
// reference would enter its big-file branch, which is not compiled here; the driver enforces that.
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "common.h"
#include "shuffle.h"
#include "sketch.h"

#ifdef RK_GPU_BINDING
// oracle/_ref/ref_sketch_driver_gpu: the same driver over integration/gpu_sketch_backend.cpp (the reference-side binding
// of librabbitkssd.so for sketchFastaFile / transSketches) instead of the reference's CPU loops; saveSketches and
// readSketches stay the reference's own
bool sketchFastaFile_gpu(std::string, bool, int, kssd_parameter_t, std::vector<sketch_t> &, sketchInfo_t &, std::string);
void transSketches_gpu(std::vector<sketch_t> &, sketchInfo_t &, std::string, std::string, int);
#define sketchFastaFile sketchFastaFile_gpu
#define transSketches transSketches_gpu
#endif

static int usage()
{
    fprintf(stderr,
            "usage:\n"
            "  ref_sketch_driver sketch  file.shuf list out[.sketch] threads isQuery\n"
            "  ref_sketch_driver sketchfq file.shuf list out[.sketch] threads isQuery leastQual leastNumKmer\n"
            "  ref_sketch_driver dump in.sketch       (readSketches -> text: info, then name, count, sorted hashes)\n"
            "  ref_sketch_driver resave in.sketch out.sketch   (readSketches -> saveSketches + transSketches)\n"
            "  ref_sketch_driver tokssd in.sketch outdir       (readSketches -> Kssd directory)\n"
            "  ref_sketch_driver fromkssd indir out.sketch     (Kssd directory -> saveSketches)\n");
    return 2;
}

// the reference's own split (src/sketch.cpp:352-374): refuse inputs that would take the big-file branch
static void check_small_files(const char *list, int threads)
{
    std::ifstream fs(list);
    std::string name;
    std::vector<uint64_t> sizes;
    uint64_t total = 0;
    while (std::getline(fs, name)) {
        struct stat st;
        if (stat(name.c_str(), &st)) { fprintf(stderr, "ref_sketch_driver: cannot stat %s\n", name.c_str()); exit(2); }
        sizes.push_back((uint64_t)st.st_size);
        total += (uint64_t)st.st_size;
    }
    const uint64_t limit = total / (uint64_t)threads;
    for (uint64_t s : sizes)
        if (s > limit) {
            fprintf(stderr, "ref_sketch_driver: a %llu-byte file exceeds totalSize/numThreads = %llu: the big-file "
                            "branch (RabbitFX) is not part of this build\n", (unsigned long long)s, (unsigned long long)limit);
            exit(2);
        }
}

static void dump(std::vector<sketch_t> &sk, const sketchInfo_t &info)
{
    printf("info %d %d %d %d %d\n", info.id, info.half_k, info.half_subk, info.drlevel, info.genomeNumber);
    const bool use64 = info.half_k - info.drlevel > 8;
    for (sketch_t &s : sk) {
        if (use64) {
            std::sort(s.hashSet64.begin(), s.hashSet64.end());
            printf("%s\t%zu", s.fileName.c_str(), s.hashSet64.size());
            for (uint64_t h : s.hashSet64) printf("\t%llu", (unsigned long long)h);
        } else {
            std::sort(s.hashSet.begin(), s.hashSet.end());
            printf("%s\t%zu", s.fileName.c_str(), s.hashSet.size());
            for (uint32_t h : s.hashSet) printf("\t%u", h);
        }
        printf("\n");
    }
}

int main(int argc, char **argv)
{
    if (argc < 2) return usage();
    const std::string cmd = argv[1];
    if ((cmd == "sketch" && argc == 7) || (cmd == "sketchfq" && argc == 9)) {
        dim_shuffle_t *sh = read_shuffle_dim(argv[2]);
        kssd_parameter_t p = initParameter(sh->dim_shuffle_stat.k, sh->dim_shuffle_stat.subk, sh->dim_shuffle_stat.drlevel,
                                           sh->shuffled_dim);
        const int threads = atoi(argv[5]);
#ifndef RK_GPU_BINDING
        check_small_files(argv[3], threads);
#endif
        std::vector<sketch_t> sk;
        sketchInfo_t info;
        bool ok;
        if (cmd == "sketch") ok = sketchFastaFile(argv[3], atoi(argv[6]) != 0, threads, p, sk, info, argv[4]);
        else ok = sketchFastqFile(argv[3], atoi(argv[6]) != 0, threads, p, atoi(argv[7]), atoi(argv[8]), sk, info, argv[4]);
        return ok ? 0 : 1;
    }
    if (cmd == "dump" && argc == 3) {
        std::vector<sketch_t> sk;
        sketchInfo_t info;
        readSketches(sk, info, argv[2]);
        dump(sk, info);
        return 0;
    }
    if (cmd == "resave" && argc == 4) {
        std::vector<sketch_t> sk;
        sketchInfo_t info;
        readSketches(sk, info, argv[2]);
        saveSketches(sk, info, argv[3]);
        transSketches(sk, info, std::string(argv[3]) + ".dict", std::string(argv[3]) + ".index", 2);
        return 0;
    }
    if (cmd == "tokssd" && argc == 4) {
        std::vector<sketch_t> sk;
        sketchInfo_t info;
        readSketches(sk, info, argv[2]);
        convert_from_RabbitKSSDSketch_to_KssdSketch(sk, info, argv[3], 1);
        return 0;
    }
    if (cmd == "fromkssd" && argc == 4) {
        std::vector<sketch_t> sk;
        sketchInfo_t info;
        convertSketch(sk, info, argv[2], 1);
        saveSketches(sk, info, argv[3]);
        return 0;
    }
    return usage();
}
