/*
 * kssd_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of RabbitKSSD's hot path (sketch + index + distance),
 * written from the reference sources as a specification; every function cites the
 * reference file:line it follows.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library, and only as the checker.
 *
 * Pinning status (see oracle/README.md and DESIGN.md):
 *   - parameters/masks (S1), .shuf generation (S2), the whole distance path
 *     (D1-D6: counts, Jaccard/Mash, containment/AafD, thresholds, top-N, text)
 *     are pinned against the REAL reference objects (common.cpp, shuffle.cpp,
 *     dist.cpp compiled unmodified into oracle/_ref/ref_driver) and against the
 *     committed fixtures in tests/golden/ produced by that binary.
 *   - the .dict/.index layout (I1) is pinned indirectly: the real index_tridist /
 *     index_dist consume files written by ok_index_build32 and must reproduce the
 *     real brute-force tri_dist (dist.cpp:345) results.
 *   - the FASTA/FASTQ record reader (S0) is pinned against the reference's own kseq.h
 *     (header-only, instantiated and looped like src/sketch.cpp:17,462-479 in
 *     oracle/_ref/ref_driver kseq): 29 well-formed and malformed inputs under
 *     tests/golden/kseq/.
 *   - the sketch arithmetic (S3-S5) and .sketch I/O (S8) live in sketch.cpp, which needs
 *     the un-vendored RabbitFX submodule and is therefore unbuildable here: PARITY
 *     UNPINNED by execution; restated line-by-line and cross-checked by an independent
 *     numpy restatement.
 */
#ifndef KSSD_ORACLE_H
#define KSSD_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/common.h:8-25 (kssd_parameter_t), the fields the hot path reads. */
typedef struct ok_param {
    int32_t half_k, half_subk, drlevel;
    int32_t rev_add_move, half_outctx_len;
    int32_t dim_start, dim_end;
    uint32_t kmer_size;
    uint64_t domask, tupmask, undomask0, undomask1;
} ok_param_t;

/* src/sketch.h:27-34 (sketchInfo_t), 20 bytes on disk. */
typedef struct ok_sketch_info {
    int32_t id, half_k, half_subk, drlevel, genomeNumber;
} ok_sketch_info_t;

/* one reported pair (src/dist.cpp:207-255 alldist, :600-682 dist). */
typedef struct ok_hit {
    uint32_t row;   /* query index i                          */
    uint32_t col;   /* reference index j                      */
    int32_t common; /* |S_i n S_j|                            */
    int32_t size0;  /* alldist: size of row i; dist: ref size */
    int32_t size1;  /* alldist: size of col j; dist: qry size */
    int32_t pad_;
    double jorc;    /* jaccard (metric 0) or containment (1)  */
    double dist;    /* mashD (metric 0) or AafD (1)           */
} ok_hit_t;

/* S1: src/common.cpp:35-78.  Returns 0, or -1 when half_subk - drlevel < 3 (:37). */
int ok_init_param(int half_k, int half_subk, int drlevel, ok_param_t *p);

/* S2: src/shuffle.cpp:25-104.  Fills table[16^half_subk] exactly like
 * write_shuffle_dim_file (glibc rand(): srand(23) pass then srand(id) pass).
 * Returns id=(k<<8)+(subk<<4)+l, or -1 on the reference's argument errors. */
int ok_shuffle_table(int half_k, int half_subk, int drlevel, int32_t *table);
int ok_write_shuf(const char *path, int half_k, int half_subk, int drlevel);
/* src/shuffle.cpp:8-23.  *table is malloc'd; header returned in hdr[4]. */
int ok_read_shuf(const char *path, int32_t hdr[4], int32_t **table);

/* S0: kseq_read semantics (src/kseq.h:176-215) as used at src/sketch.cpp:462-485.
 * Reads a plain or gzip'd FASTA/FASTQ file; returns the sequence bytes of all
 * records concatenated (*seq, malloc'd), with rec_off[n_rec+1] (malloc'd). */
int ok_read_fasta(const char *path, uint8_t **seq, uint64_t **rec_off, uint64_t *n_rec);
/* same parser over a memory buffer holding the (already decompressed) file text */
int ok_parse_fasta_mem(const uint8_t *buf, uint64_t n, uint8_t **seq, uint64_t **rec_off,
                       uint64_t *n_rec);

/* FASTQ: also the quality characters, one per base (records without qualities get '~') */
int ok_parse_fastq_mem(const uint8_t *buf, uint64_t n, uint8_t **seq, uint8_t **qual,
                       uint64_t **rec_off, uint64_t *n_rec);
/* f1: src/sketch.cpp:781-845 (sketchFastqFile): base valid iff ACGT and quality >= least_qual;
 * hash kept iff it occurred >= least_num times.  Output sorted. */
int ok_sketch_genomes_mt(const ok_param_t *p, const int32_t *shuffled_dim, const uint8_t *seq, const uint64_t *goff,
                         uint64_t n_genomes, int threads, uint64_t *sizes_out);
int64_t ok_sketch_records_fastq(const ok_param_t *p, const int32_t *shuffled_dim, const uint8_t *seq,
                                const uint8_t *qual, int least_qual, int least_num,
                                const uint64_t *rec_off, uint64_t n_rec, uint64_t **hashes_out);

/* BaseMap of src/common.h:27-37: 0..3 for ACGT/acgt, -1 otherwise */
int ok_base_code(int ch);

/* S3-S5: src/sketch.cpp:487-550.  Windows never span records.  Output: sorted
 * unique dr_tuples (u64; the 32-bit path narrows them).  Returns count or <0. */
int64_t ok_sketch_records(const ok_param_t *p, const int32_t *shuffled_dim, const uint8_t *seq,
                          const uint64_t *rec_off, uint64_t n_rec, uint64_t **hashes_out);
/* number of k-mer windows (positions where base > kmer_size) in the records */
uint64_t ok_count_windows(const ok_param_t *p, const uint8_t *seq, const uint64_t *rec_off,
                          uint64_t n_rec);

/* S8: src/sketch.cpp:1024-1154, 32-bit hash layout (half_k - drlevel <= 8). */
int ok_save_sketches32(const char *path, const ok_sketch_info_t *info, const char *const *names,
                       const uint32_t *hashes, const uint64_t *off);
/* all outputs malloc'd; names is one malloc'd blob of NUL-terminated strings */
int ok_read_sketches32(const char *path, ok_sketch_info_t *info, char **names_blob,
                       uint32_t **hashes, uint64_t **off);

/* I1: src/sketch.cpp:970-1017.  postings ordered (hash asc, genome asc);
 * counts is the dense 2^hash_bits array of the .index file.  Both malloc'd. */
int ok_index_build32(const uint32_t *hashes, const uint64_t *off, uint32_t n_genomes,
                     int hash_bits, uint32_t **postings, uint32_t **counts, uint64_t *total);
int ok_write_index32(const char *dict_path, const char *index_path, const uint32_t *postings,
                     const uint32_t *counts, int hash_bits, uint64_t total);
int ok_read_index32(const char *dict_path, const char *index_path, uint32_t **postings,
                    uint32_t **counts, uint64_t *hash_size, uint64_t *total);

/* D3/D4 scalar epilogue: src/dist.cpp:218-231 (metric 0), :238-250 (metric 1). */
void ok_distance(int common, int size0, int size1, int metric, int kmer_size, double *jorc,
                 double *dist);

/* D1-D4: src/dist.cpp:86-129,174-258 (triangle=1: alldist, cols j>i, '<' maxDist)
 *        src/dist.cpp:490-522,560-692 (triangle=0: dist, all cols, '<=' maxDist).
 * counts/postings/ref_sizes describe the reference index; q_* the query CSR
 * (for alldist pass the same sketches).  If common_dense != NULL it receives
 * the full n_query x n_ref counter rows.  hits_out malloc'd, row-major order,
 * ascending col inside a row.  threads>1 uses OpenMP schedule(dynamic) like the
 * reference.  Returns number of hits or <0. */
int64_t ok_index_dist32(const uint32_t *counts, int hash_bits, const uint32_t *postings,
                        const uint32_t *ref_sizes, uint32_t n_ref, const uint32_t *q_hashes,
                        const uint64_t *q_off, uint32_t n_query, int triangle, int metric,
                        int kmer_size, double max_dist, int threads, int32_t *common_dense,
                        ok_hit_t **hits_out);

/* D5: src/dist.cpp:599,625-640,683-689.  Applies the reference's max-heap
 * (std::priority_queue over cmpDistInfo, src/dist.h:19-32) to the hits of ONE
 * query row given in ascending col order; writes the survivors in the order the
 * reference emits them (largest distance first).  Returns how many. */
uint32_t ok_topn_row(const ok_hit_t *row_hits, uint32_t n, uint64_t max_neighbor, ok_hit_t *out);

/* D6: one output line, src/dist.cpp:233 / :642 (std::to_string(double) == "%f"). */
int ok_format_hit(char *buf, size_t cap, const char *name_a, const char *name_b, int common,
                  int size0, int size1, double jorc, double dist);

/* ---- 64-bit hash layout (use64 = half_k - drlevel > 8), kssd_oracle64.c ---------------- */
int ok_save_sketches64(const char *path, const ok_sketch_info_t *info, const char *const *names,
                       const uint64_t *hashes, const uint64_t *off);
int ok_read_sketches64(const char *path, ok_sketch_info_t *info, char **names_blob, uint64_t **hashes,
                       uint64_t **off);
/* I2: src/sketch.cpp:904-969; uhash ascending, postings grouped by hash, genome asc inside */
int ok_index_build64(const uint64_t *hashes, const uint64_t *off, uint32_t n_genomes, uint64_t **uhash,
                     uint32_t **ucount, uint32_t **postings, uint64_t *n_hash, uint64_t *total);
int ok_write_index64(const char *dict_path, const char *index_path, const uint32_t *postings,
                     const uint64_t *uhash, const uint32_t *ucount, uint64_t n_hash, uint64_t total);
int ok_read_index64(const char *dict_path, const char *index_path, uint32_t **postings, uint64_t **uhash,
                    uint32_t **ucount, uint64_t *n_hash, uint64_t *total);
/* src/dist.cpp:181-191 / :566-576 + the common epilogue */
int64_t ok_index_dist64(const uint64_t *uhash, const uint32_t *ucount, uint64_t n_hash,
                        const uint32_t *postings, const uint32_t *ref_sizes, uint32_t n_ref,
                        const uint64_t *q_hashes, const uint64_t *q_off, uint32_t n_query, int triangle,
                        int metric, int kmer_size, double max_dist, int threads, int32_t *common_dense,
                        ok_hit_t **hits_out);

void ok_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
