/*
 * kssd_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see kssd_oracle.h).
 * Plain-C restatement of RabbitKSSD's sketch/index/distance path.  Citations are
 * file:line into the reference tree (/root/reference, not present on the GPU box).
 */
#define _GNU_SOURCE
#include "kssd_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void ok_free(void *p) { free(p); }

/* ------------------------------------------------------------------ S1 */
/* src/common.cpp:35-78 */
int ok_init_param(int half_k, int half_subk, int drlevel, ok_param_t *p)
{
    if (half_subk - drlevel < 3) return -1; /* :37 */
    memset(p, 0, sizeof(*p));
    p->half_k = half_k;
    p->half_subk = half_subk;
    p->drlevel = drlevel;
    int out = half_k - half_subk; /* :44 */
    p->half_outctx_len = out;
    p->rev_add_move = 4 * half_k - 2; /* :46 */
    p->kmer_size = 2u * (unsigned)half_k; /* :47 */
    p->dim_start = 0;
    p->dim_end = 1 << 4 * (half_subk - drlevel); /* :49 */
    int comp_bittl = 64 - 4 * half_k; /* :60 */
    uint64_t tupmask = 0xffffffffffffffffULL >> comp_bittl; /* :63 */
    uint64_t domask = (tupmask >> (4 * out)) << (2 * out); /* :64 */
    uint64_t undomask = (tupmask ^ domask) & tupmask; /* :65 */
    uint64_t undomask1 = undomask & (tupmask >> ((half_k + half_subk) * 2)); /* :66 */
    uint64_t undomask0 = undomask ^ undomask1; /* :67 */
    p->tupmask = tupmask;
    p->domask = domask;
    p->undomask0 = undomask0;
    p->undomask1 = undomask1;
    return 0;
}

/* ------------------------------------------------------------------ S2 */
/* src/shuffle.cpp:87-104: Fisher-Yates driven by glibc srand()/rand(). */
static void fisher_yates(int32_t *arr, int length, uint64_t seed)
{
    srand((unsigned)seed);
    for (int i = length - 1; i > 0; i--) {
        int j = rand() % (i + 1);
        int32_t tmp = arr[i];
        arr[i] = arr[j];
        arr[j] = tmp;
    }
}

/* src/shuffle.cpp:25-61 (write_shuffle_dim_file) + :76-85 (shuffleN). */
int ok_shuffle_table(int half_k, int half_subk, int drlevel, int32_t *table)
{
    if (half_k < half_subk) return -1; /* :26 */
    if (half_subk >= 8) return -1;     /* :30 */
    int n = 1 << 4 * half_subk;
    if (n > RAND_MAX) return -1; /* :89 */
    int id = (half_k << 8) + (half_subk << 4) + drlevel; /* :50 */
    for (int i = 0; i < n; i++) table[i] = i; /* shuffleN(n, 0) :80-82 */
    fisher_yates(table, n, 23);               /* :84 */
    fisher_yates(table, n, (uint64_t)id);     /* :54 */
    return id;
}

int ok_write_shuf(const char *path, int half_k, int half_subk, int drlevel)
{
    int n = 1 << 4 * half_subk;
    int32_t *t = (int32_t *)malloc((size_t)n * 4);
    if (!t) return -2;
    int id = ok_shuffle_table(half_k, half_subk, drlevel, t);
    if (id < 0) { free(t); return -1; }
    FILE *fp = fopen(path, "wb");
    if (!fp) { free(t); return -3; }
    int32_t hdr[4] = {id, half_k, half_subk, drlevel}; /* dim_shuffle_stat_t, shuffle.h:11-17 */
    size_t w = fwrite(hdr, sizeof(hdr), 1, fp);
    w += fwrite(t, 4, (size_t)n, fp);
    fclose(fp);
    free(t);
    return w == (size_t)n + 1 ? 0 : -4;
}

/* src/shuffle.cpp:8-23 */
int ok_read_shuf(const char *path, int32_t hdr[4], int32_t **table)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return -1;
    if (fread(hdr, 16, 1, fp) != 1) { fclose(fp); return -2; }
    if (hdr[2] < 0 || hdr[2] >= 8) { fclose(fp); return -2; }
    size_t n = (size_t)1 << 4 * hdr[2];
    int32_t *t = (int32_t *)malloc(n * 4);
    if (!t) { fclose(fp); return -3; }
    size_t r = fread(t, 4, n, fp);
    fclose(fp);
    if (r != n) { free(t); return -2; }
    *table = t;
    return 0;
}

/* ------------------------------------------------------------------ S0 */
/* A growable byte string (kstring_t). */
typedef struct { uint8_t *s; uint64_t l, m; } kstr_t;
static int kstr_reserve(kstr_t *k, uint64_t need)
{
    if (need <= k->m) return 0;
    uint64_t m = k->m ? k->m : 256;
    while (m < need) m <<= 1;
    uint8_t *ns = (uint8_t *)realloc(k->s, m);
    if (!ns) return -1;
    k->s = ns;
    k->m = m;
    return 0;
}

typedef struct { const uint8_t *buf; uint64_t n, pos; } mstream_t;
static int ms_getc(mstream_t *ms) { return ms->pos < ms->n ? (int)ms->buf[ms->pos++] : -1; }
static int is_space(int c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

/* ks_getuntil2 (src/kseq.h:94-143) over a whole-file buffer.
 * mode 0 = KS_SEP_SPACE, 2 = KS_SEP_LINE.  dst may be NULL (discard). */
static int64_t ms_getuntil(mstream_t *ms, int mode, kstr_t *dst, int *dret, int append)
{
    if (dret) *dret = 0;
    if (dst && !append) dst->l = 0;
    if (ms->pos >= ms->n) return -1; /* !gotany && eof, :136 */
    uint64_t i = ms->pos;
    if (mode == 2) while (i < ms->n && ms->buf[i] != '\n') i++;
    else while (i < ms->n && !is_space(ms->buf[i])) i++;
    if (dst) {
        if (kstr_reserve(dst, dst->l + (i - ms->pos) + 1)) return -3;
        memcpy(dst->s + dst->l, ms->buf + ms->pos, i - ms->pos);
        dst->l += i - ms->pos;
    }
    if (i < ms->n) {
        if (dret) *dret = ms->buf[i];
        ms->pos = i + 1;
    } else ms->pos = ms->n;
    if (dst && mode == 2 && dst->l > 1 && dst->s[dst->l - 1] == '\r') --dst->l; /* :140 */
    return dst ? (int64_t)dst->l : 0;
}

/* kseq_read, src/kseq.h:176-215.  Returns seq length, -1 EOF, -2 bad FASTQ. */
static int64_t ms_kseq_read(mstream_t *ms, int *last_char, kstr_t *seq, kstr_t *qual)
{
    int c;
    if (*last_char == 0) { /* :180-184 */
        while ((c = ms_getc(ms)) != -1 && c != '>' && c != '@') {}
        if (c == -1) return -1;
        *last_char = c;
    }
    seq->l = 0;
    qual->l = 0;
    kstr_t name = {0, 0, 0};
    int64_t r = ms_getuntil(ms, 0, &name, &c, 0); /* :186 */
    free(name.s);
    if (r < 0) return -1;
    if (c != '\n') ms_getuntil(ms, 2, NULL, NULL, 0); /* comment, :187 */
    if (kstr_reserve(seq, 256)) return -3;
    while ((c = ms_getc(ms)) != -1 && c != '>' && c != '+' && c != '@') { /* :192 */
        if (c == '\n') continue;
        if (kstr_reserve(seq, seq->l + 2)) return -3;
        seq->s[seq->l++] = (uint8_t)c;
        if (ms_getuntil(ms, 2, seq, NULL, 1) == -3) return -3; /* :195 */
    }
    if (c == '>' || c == '@') *last_char = c; /* :197 */
    if (c != '+') return (int64_t)seq->l;       /* FASTA, :204 */
    while ((c = ms_getc(ms)) != -1 && c != '\n') {} /* :209 */
    if (c == -1) return -2;
    while (ms_getuntil(ms, 2, qual, NULL, 1) >= 0 && qual->l < seq->l) {} /* :211 */
    *last_char = 0;
    if (seq->l != qual->l) return -2;
    return (int64_t)seq->l;
}

static int parse_mem(const uint8_t *buf, uint64_t n, uint8_t **seq_out, uint8_t **qual_out,
                     uint64_t **rec_off_out, uint64_t *n_rec_out);

int ok_parse_fasta_mem(const uint8_t *buf, uint64_t n, uint8_t **seq_out, uint64_t **rec_off_out,
                       uint64_t *n_rec_out)
{
    return parse_mem(buf, n, seq_out, NULL, rec_off_out, n_rec_out);
}

/* same, also returning the per-base quality characters (FASTQ, src/sketch.cpp:775-776);
 * records without a quality string (FASTA) get '~' so they pass every gate */
int ok_parse_fastq_mem(const uint8_t *buf, uint64_t n, uint8_t **seq_out, uint8_t **qual_out,
                       uint64_t **rec_off_out, uint64_t *n_rec_out)
{
    return parse_mem(buf, n, seq_out, qual_out, rec_off_out, n_rec_out);
}

static int parse_mem(const uint8_t *buf, uint64_t n, uint8_t **seq_out, uint8_t **qual_out,
                     uint64_t **rec_off_out, uint64_t *n_rec_out)
{
    mstream_t ms = {buf, n, 0};
    int last_char = 0;
    kstr_t seq = {0, 0, 0}, qual = {0, 0, 0}, all = {0, 0, 0}, allq = {0, 0, 0};
    uint64_t cap = 16, nrec = 0;
    uint64_t *off = (uint64_t *)malloc((cap + 1) * 8);
    if (!off) return -3;
    off[0] = 0;
    for (;;) {
        int64_t len = ms_kseq_read(&ms, &last_char, &seq, &qual);
        if (len < 0) break; /* src/sketch.cpp:475-478 */
        if (nrec == cap) {
            cap *= 2;
            uint64_t *no = (uint64_t *)realloc(off, (cap + 1) * 8);
            if (!no) { free(off); free(seq.s); free(qual.s); free(all.s); return -3; }
            off = no;
        }
        if (kstr_reserve(&all, all.l + (uint64_t)len + 1)) { free(off); return -3; }
        memcpy(all.s + all.l, seq.s, (size_t)len);
        all.l += (uint64_t)len;
        if (qual_out) {
            if (kstr_reserve(&allq, allq.l + (uint64_t)len + 1)) { free(off); return -3; }
            if (qual.l == (uint64_t)len) memcpy(allq.s + allq.l, qual.s, (size_t)len);
            else memset(allq.s + allq.l, '~', (size_t)len);
            allq.l += (uint64_t)len;
        }
        off[++nrec] = all.l;
    }
    free(seq.s);
    free(qual.s);
    if (!all.s) all.s = (uint8_t *)calloc(1, 1);
    if (qual_out) {
        if (!allq.s) allq.s = (uint8_t *)calloc(1, 1);
        *qual_out = allq.s;
    }
    *seq_out = all.s;
    *rec_off_out = off;
    *n_rec_out = nrec;
    return 0;
}

int ok_read_fasta(const char *path, uint8_t **seq, uint64_t **rec_off, uint64_t *n_rec)
{
    gzFile fp = gzopen(path, "r"); /* src/sketch.cpp:462 (gzopen reads plain text too) */
    if (!fp) return -1;
    uint64_t cap = 1 << 20, n = 0;
    uint8_t *buf = (uint8_t *)malloc(cap);
    if (!buf) { gzclose(fp); return -3; }
    for (;;) {
        if (cap - n < (1 << 16)) {
            cap *= 2;
            uint8_t *nb = (uint8_t *)realloc(buf, cap);
            if (!nb) { free(buf); gzclose(fp); return -3; }
            buf = nb;
        }
        int r = gzread(fp, buf + n, (unsigned)(cap - n > (1u << 30) ? (1u << 30) : cap - n));
        if (r <= 0) break;
        n += (uint64_t)r;
    }
    gzclose(fp);
    int rc = ok_parse_fasta_mem(buf, n, seq, rec_off, n_rec);
    free(buf);
    return rc;
}

/* --------------------------------------------------------------- S3-S5 */
/* src/common.h:27-37: ACGT/acgt -> 0..3, everything else -1.  Bytes >= 0x80
 * index the reference table out of bounds; treated as -1. */
static int base_code(uint8_t ch)
{
    switch (ch) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}

/* BaseMap, src/common.h:27-37, for the test that pins it against the reference's table */
int ok_base_code(int ch) { return ch >= 0 && ch < 256 ? base_code((uint8_t)ch) : -1; }

static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

/* src/sketch.cpp:487-530 (identical copies :198-231).  emit==NULL only counts. */
static uint64_t scan_records_q(const ok_param_t *p, const int32_t *shuffled_dim, const uint8_t *seq,
                               const uint8_t *qual, int least_qual, const uint64_t *rec_off,
                               uint64_t n_rec, kstr_t *emit, uint64_t *n_windows, int *err);

static uint64_t scan_records(const ok_param_t *p, const int32_t *shuffled_dim, const uint8_t *seq,
                             const uint64_t *rec_off, uint64_t n_rec, kstr_t *emit,
                             uint64_t *n_windows, int *err)
{
    return scan_records_q(p, shuffled_dim, seq, NULL, 0, rec_off, n_rec, emit, n_windows, err);
}

/* membership bits of the selected .shuf entries: the role of the reference's `shuffled_map` (src/sketch.cpp:338-345,
 * a 4096-entry hash map that stays in cache), so that the timed CPU port does not pay a DRAM access into the 64 MiB
 * table for every window.  Results are unchanged: a set bit still goes through the table. */
static const uint64_t *g_sel_bits = NULL;

/* qual != NULL adds the FASTQ gate `quality[i] >= leastQual` of src/sketch.cpp:785 */
static uint64_t scan_records_q(const ok_param_t *p, const int32_t *shuffled_dim, const uint8_t *seq,
                               const uint8_t *qual, int least_qual, const uint64_t *rec_off,
                               uint64_t n_rec, kstr_t *emit, uint64_t *n_windows, int *err)
{
    const int rev_add_move = p->rev_add_move, out = p->half_outctx_len;
    const int kmer_size = (int)p->kmer_size, drlevel = p->drlevel;
    const uint64_t tupmask = p->tupmask, domask = p->domask;
    const uint64_t undomask0 = p->undomask0, undomask1 = p->undomask1;
    uint64_t n_emit = 0, windows = 0;
    for (uint64_t r = 0; r < n_rec; r++) {
        uint64_t tuple = 0, rvs_tuple = 0; /* :487 */
        int base = 1;                      /* :488 */
        for (uint64_t i = rec_off[r]; i < rec_off[r + 1]; i++) {
            int basenum = base_code(seq[i]); /* :494-495 */
            if (basenum != -1 && (!qual || (int)(char)qual[i] >= least_qual)) {
                tuple = ((tuple << 2) | (uint64_t)basenum) & tupmask;                        /* :498 */
                rvs_tuple = (rvs_tuple >> 2) + (((uint64_t)basenum ^ 3ULL) << rev_add_move); /* :499 */
                base++;
            } else base = 1; /* :503 */
            if (base > kmer_size) { /* :506 */
                windows++;
                uint64_t uni = tuple < rvs_tuple ? tuple : rvs_tuple; /* :508 */
                int dim_id = (int)((uni & domask) >> (out * 2));      /* :509 */
                if (!shuffled_dim) continue;
                if (g_sel_bits && !((g_sel_bits[(uint32_t)dim_id >> 6] >> (dim_id & 63)) & 1ULL)) continue;
                int32_t v = shuffled_dim[dim_id];
                if (!(v < p->dim_end && v >= p->dim_start)) continue; /* :341,:516 */
                uint64_t pfilter = (uint64_t)(v - p->dim_start);     /* :519-521 */
                uint64_t dr = (((uni & undomask0) |
                                ((uni & undomask1) << (kmer_size * 2 - out * 4))) >>
                               (drlevel * 4)) | pfilter; /* :524 */
                if (emit) {
                    if (kstr_reserve(emit, (n_emit + 1) * 8)) { *err = -3; return 0; }
                    ((uint64_t *)emit->s)[n_emit] = dr;
                }
                n_emit++;
            }
        }
    }
    if (n_windows) *n_windows = windows;
    return n_emit;
}

int64_t ok_sketch_records(const ok_param_t *p, const int32_t *shuffled_dim, const uint8_t *seq,
                          const uint64_t *rec_off, uint64_t n_rec, uint64_t **hashes_out)
{
    kstr_t emit = {0, 0, 0};
    int err = 0;
    uint64_t n = scan_records(p, shuffled_dim, seq, rec_off, n_rec, &emit, NULL, &err);
    if (err) { free(emit.s); return err; }
    uint64_t *h = (uint64_t *)emit.s;
    /* the reference's unordered_set (:470,:526-529) == sort + unique */
    if (n) qsort(h, n, 8, cmp_u64);
    uint64_t u = 0;
    for (uint64_t i = 0; i < n; i++)
        if (u == 0 || h[i] != h[u - 1]) h[u++] = h[i];
    if (!h) h = (uint64_t *)calloc(1, 8);
    *hashes_out = h;
    return (int64_t)u;
}

/* The reference's small-file loop (`#pragma omp parallel for ... schedule(dynamic)` over files, src/sketch.cpp:455-457)
 * over genomes already in memory: genome g = seq[goff[g] .. goff[g+1]), one record each.  sizes_out[g] = sketch size.
 * Used as the timed CPU baseline of the sketch leg (bench.py). */
int ok_sketch_genomes_mt(const ok_param_t *p, const int32_t *shuffled_dim, const uint8_t *seq, const uint64_t *goff,
                         uint64_t n_genomes, int threads, uint64_t *sizes_out)
{
    const uint64_t n_dim = 1ULL << (4 * p->half_subk);
    uint64_t *bits = (uint64_t *)calloc((n_dim + 63) / 64, 8);
    if (!bits) return -3;
    for (uint64_t t = 0; t < n_dim; t++)
        if (shuffled_dim[t] < p->dim_end && shuffled_dim[t] >= p->dim_start) bits[t >> 6] |= 1ULL << (t & 63);
    g_sel_bits = bits;
    int failed = 0;
#pragma omp parallel for num_threads(threads) schedule(dynamic)
    for (uint64_t g = 0; g < n_genomes; g++) {
        uint64_t *h = NULL;
        const uint64_t off[2] = {goff[g], goff[g + 1]};
        int64_t n = ok_sketch_records(p, shuffled_dim, seq, off, 1, &h);
        if (n < 0) failed = 1;
        sizes_out[g] = n < 0 ? 0 : (uint64_t)n;
        free(h);
    }
    g_sel_bits = NULL;
    free(bits);
    return failed ? -3 : 0;
}

/* FASTQ variant, src/sketch.cpp:781-845: quality gate + per-hash occurrence count; a hash is
 * kept when it occurred at least least_num times (hashValueMap[dr_tuple]++ ... >= leastNumKmer). */
int64_t ok_sketch_records_fastq(const ok_param_t *p, const int32_t *shuffled_dim, const uint8_t *seq,
                                const uint8_t *qual, int least_qual, int least_num,
                                const uint64_t *rec_off, uint64_t n_rec, uint64_t **hashes_out)
{
    kstr_t emit = {0, 0, 0};
    int err = 0;
    uint64_t n = scan_records_q(p, shuffled_dim, seq, qual, least_qual, rec_off, n_rec, &emit, NULL, &err);
    if (err) { free(emit.s); return err; }
    uint64_t *h = (uint64_t *)emit.s;
    if (n) qsort(h, n, 8, cmp_u64);
    uint64_t u = 0;
    for (uint64_t i = 0; i < n;) {
        uint64_t j = i;
        while (j < n && h[j] == h[i]) j++;
        if ((int64_t)(j - i) >= (int64_t)least_num) h[u++] = h[i];
        i = j;
    }
    if (!h) h = (uint64_t *)calloc(1, 8);
    *hashes_out = h;
    return (int64_t)u;
}

uint64_t ok_count_windows(const ok_param_t *p, const uint8_t *seq, const uint64_t *rec_off,
                          uint64_t n_rec)
{
    uint64_t w = 0;
    int err = 0;
    scan_records(p, NULL, seq, rec_off, n_rec, NULL, &w, &err);
    return w;
}

/* ------------------------------------------------------------------ S8 */
/* src/sketch.cpp:1024-1068 */
int ok_save_sketches32(const char *path, const ok_sketch_info_t *info_in, const char *const *names,
                       const uint32_t *hashes, const uint64_t *off)
{
    ok_sketch_info_t info = *info_in;
    int n = info.genomeNumber;
    info.id = (info.half_k << 8) + (info.half_subk << 4) + info.drlevel; /* :1029 */
    FILE *fp = fopen(path, "wb");
    if (!fp) return -1;
    fwrite(&info, sizeof(info), 1, fp);
    int32_t *len = (int32_t *)malloc(((size_t)n + 1) * 4), *cnt = (int32_t *)malloc(((size_t)n + 1) * 4);
    for (int i = 0; i < n; i++) {
        len[i] = (int32_t)strlen(names[i]);
        cnt[i] = (int32_t)(off[i + 1] - off[i]);
    }
    fwrite(len, 4, (size_t)n, fp); /* :1050 */
    fwrite(cnt, 4, (size_t)n, fp); /* :1051 */
    for (int i = 0; i < n; i++) {
        fwrite(names[i], 1, (size_t)len[i], fp);      /* :1054 */
        fwrite(hashes + off[i], 4, (size_t)cnt[i], fp); /* :1061 */
    }
    free(len);
    free(cnt);
    return fclose(fp) == 0 ? 0 : -4;
}

/* src/sketch.cpp:1070-1154 */
int ok_read_sketches32(const char *path, ok_sketch_info_t *info, char **names_blob,
                       uint32_t **hashes_out, uint64_t **off_out)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return -1;
    if (fread(info, sizeof(*info), 1, fp) != 1) { fclose(fp); return -2; }
    if (info->half_k - info->drlevel > 8) { fclose(fp); return -5; } /* 64-bit layout */
    size_t n = (size_t)info->genomeNumber;
    int32_t *len = (int32_t *)malloc((n + 1) * 4), *cnt = (int32_t *)malloc((n + 1) * 4);
    if (fread(len, 4, n, fp) != n || fread(cnt, 4, n, fp) != n) { fclose(fp); return -2; }
    uint64_t tot_name = 0, tot_hash = 0;
    for (size_t i = 0; i < n; i++) { tot_name += (uint64_t)len[i] + 1; tot_hash += (uint64_t)cnt[i]; }
    char *blob = (char *)malloc(tot_name + 1);
    uint32_t *h = (uint32_t *)malloc((tot_hash + 1) * 4);
    uint64_t *off = (uint64_t *)malloc((n + 1) * 8);
    uint64_t np = 0;
    off[0] = 0;
    int rc = 0;
    for (size_t i = 0; i < n; i++) {
        if (fread(blob + np, 1, (size_t)len[i], fp) != (size_t)len[i]) { rc = -2; break; }
        np += (uint64_t)len[i];
        blob[np++] = 0;
        if (fread(h + off[i], 4, (size_t)cnt[i], fp) != (size_t)cnt[i]) { rc = -2; break; }
        off[i + 1] = off[i] + (uint64_t)cnt[i];
    }
    fclose(fp);
    free(len);
    free(cnt);
    if (rc) { free(blob); free(h); free(off); return rc; }
    *names_blob = blob;
    *hashes_out = h;
    *off_out = off;
    return 0;
}

/* ------------------------------------------------------------------ I1 */
/* src/sketch.cpp:970-1017.  hashMapId[hash].push_back(i) for i ascending (:979-985)
 * then concatenation in ascending hash (:993-1000) == a stable counting sort. */
int ok_index_build32(const uint32_t *hashes, const uint64_t *off, uint32_t n_genomes,
                     int hash_bits, uint32_t **postings_out, uint32_t **counts_out,
                     uint64_t *total_out)
{
    uint64_t hash_size = 1ULL << hash_bits; /* :971 */
    uint64_t H = off[n_genomes];
    uint32_t *counts = (uint32_t *)calloc(hash_size, 4);
    uint32_t *postings = (uint32_t *)malloc((H + 1) * 4);
    uint64_t *cursor = (uint64_t *)malloc(hash_size * 8);
    if (!counts || !postings || !cursor) { free(counts); free(postings); free(cursor); return -3; }
    for (uint64_t e = 0; e < H; e++) {
        if ((uint64_t)hashes[e] >= hash_size) { free(counts); free(postings); free(cursor); return -2; }
        counts[hashes[e]]++;
    }
    uint64_t acc = 0;
    for (uint64_t h = 0; h < hash_size; h++) { cursor[h] = acc; acc += counts[h]; }
    for (uint32_t g = 0; g < n_genomes; g++)
        for (uint64_t e = off[g]; e < off[g + 1]; e++) postings[cursor[hashes[e]]++] = g;
    free(cursor);
    *postings_out = postings;
    *counts_out = counts;
    *total_out = H; /* totalIndex, :997 */
    return 0;
}

/* .dict = bare u32 postings (:991-1001); .index = {size_t hashSize; u64 totalIndex;
 * u32 count[hashSize]} (:1008-1011). */
int ok_write_index32(const char *dict_path, const char *index_path, const uint32_t *postings,
                     const uint32_t *counts, int hash_bits, uint64_t total)
{
    uint64_t hash_size = 1ULL << hash_bits;
    FILE *fd = fopen(dict_path, "wb");
    if (!fd) return -1;
    size_t w = fwrite(postings, 4, (size_t)total, fd);
    if (fclose(fd) || w != total) return -4;
    FILE *fi = fopen(index_path, "wb");
    if (!fi) return -1;
    fwrite(&hash_size, 8, 1, fi);
    fwrite(&total, 8, 1, fi);
    w = fwrite(counts, 4, (size_t)hash_size, fi);
    if (fclose(fi) || w != hash_size) return -4;
    return 0;
}

/* src/dist.cpp:86-129 (the file reads; the prefix sum lives in ok_index_dist32) */
int ok_read_index32(const char *dict_path, const char *index_path, uint32_t **postings_out,
                    uint32_t **counts_out, uint64_t *hash_size_out, uint64_t *total_out)
{
    FILE *fi = fopen(index_path, "rb");
    if (!fi) return -1;
    uint64_t hash_size, total;
    if (fread(&hash_size, 8, 1, fi) != 1 || fread(&total, 8, 1, fi) != 1) { fclose(fi); return -2; }
    uint32_t *counts = (uint32_t *)malloc((hash_size + 1) * 4);
    if (!counts) { fclose(fi); return -3; }
    if (fread(counts, 4, (size_t)hash_size, fi) != hash_size) { fclose(fi); free(counts); return -2; }
    fclose(fi);
    FILE *fd = fopen(dict_path, "rb");
    if (!fd) { free(counts); return -1; }
    uint32_t *postings = (uint32_t *)malloc((total + 1) * 4);
    if (!postings) { fclose(fd); free(counts); return -3; }
    if (fread(postings, 4, (size_t)total, fd) != total) { fclose(fd); free(counts); free(postings); return -2; }
    fclose(fd);
    *postings_out = postings;
    *counts_out = counts;
    *hash_size_out = hash_size;
    *total_out = total;
    return 0;
}

/* --------------------------------------------------------------- D3/D4 */
/* src/dist.cpp:218-231 (Jaccard -> Mash) and :238-250 (containment -> AafD). */
void ok_distance(int common, int size0, int size1, int metric, int kmer_size, double *jorc,
                 double *dist)
{
    if (!metric) {
        int denom = size0 + size1 - common; /* :219 */
        double jaccard;
        if (size0 == 0 || size1 == 0) jaccard = 0.0; /* :221 */
        else jaccard = (double)common / denom;       /* :224 */
        double mashD;
        if (jaccard == 1.0) mashD = 0.0;
        else if (jaccard == 0.0) mashD = 1.0;
        else mashD = (double)-1.0 / kmer_size * log((2 * jaccard) / (1.0 + jaccard)); /* :231 */
        *jorc = jaccard;
        *dist = mashD;
    } else {
        int denom = size0 < size1 ? size0 : size1; /* :238 */
        double containment;
        if (size0 == 0 || size1 == 0) containment = 0.0;
        else containment = (double)common / denom;
        double AafD;
        if (containment == 1.0) AafD = 0.0;
        else if (containment == 0.0) AafD = 1.0;
        else AafD = (double)-1.0 / kmer_size * log(containment); /* :250 */
        *jorc = containment;
        *dist = AafD;
    }
}

/* --------------------------------------------------------------- D1-D4 */
typedef struct { ok_hit_t *v; uint64_t n, cap; } hitvec_t;
static int hv_push(hitvec_t *hv, const ok_hit_t *h)
{
    if (hv->n == hv->cap) {
        uint64_t nc = hv->cap ? hv->cap * 2 : 1024;
        ok_hit_t *nv = (ok_hit_t *)realloc(hv->v, nc * sizeof(ok_hit_t));
        if (!nv) return -1;
        hv->v = nv;
        hv->cap = nc;
    }
    hv->v[hv->n++] = *h;
    return 0;
}

int64_t ok_index_dist32(const uint32_t *counts, int hash_bits, const uint32_t *postings,
                        const uint32_t *ref_sizes, uint32_t n_ref, const uint32_t *q_hashes,
                        const uint64_t *q_off, uint32_t n_query, int triangle, int metric,
                        int kmer_size, double max_dist, int threads, int32_t *common_dense,
                        ok_hit_t **hits_out)
{
    uint64_t hash_size = 1ULL << hash_bits;
    /* D1: inclusive prefix sum into size_t offset[hashSize], src/dist.cpp:100-106 */
    uint64_t *offset = (uint64_t *)malloc(hash_size * 8);
    if (!offset) return -3;
    for (uint64_t i = 0; i < hash_size; i++) {
        offset[i] = counts[i];
        if (i > 0) offset[i] += offset[i - 1];
    }
    if (threads < 1) threads = 1;
    hitvec_t *row_hits = (hitvec_t *)calloc(n_query ? n_query : 1, sizeof(hitvec_t));
    int fail = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
    {
        int32_t *row = (int32_t *)malloc(((size_t)n_ref + 1) * 4); /* intersectionArr[tid], :167 */
#ifdef _OPENMP
#pragma omp for schedule(dynamic) /* :174 / :560 */
#endif
        for (int64_t i = 0; i < (int64_t)n_query; i++) {
            if (!row) { fail = 1; continue; }
            memset(row, 0, (size_t)n_ref * 4); /* :179 */
            for (uint64_t e = q_off[i]; e < q_off[i + 1]; e++) { /* :194-203 */
                uint32_t hash = q_hashes[e];
                if ((uint64_t)hash >= hash_size) continue; /* reference would read out of bounds */
                if (counts[hash] == 0) continue;       /* :196 */
                uint64_t start = hash > 0 ? offset[hash - 1] : 0; /* :197 */
                uint64_t end = offset[hash];                      /* :198 */
                for (uint64_t k = start; k < end; k++) row[postings[k]]++; /* :199-202 */
            }
            if (common_dense) memcpy(common_dense + (size_t)i * n_ref, row, (size_t)n_ref * 4);
            int qsize = (int)(q_off[i + 1] - q_off[i]);
            for (uint32_t j = triangle ? (uint32_t)i + 1 : 0; j < n_ref; j++) { /* :207 / :600 */
                ok_hit_t h;
                h.row = (uint32_t)i;
                h.col = j;
                h.common = row[j];
                h.pad_ = 0;
                if (triangle) { h.size0 = qsize; h.size1 = (int)ref_sizes[j]; }      /* :215-216 */
                else          { h.size0 = (int)ref_sizes[j]; h.size1 = qsize; }      /* :607-608 */
                ok_distance(h.common, h.size0, h.size1, metric, kmer_size, &h.jorc, &h.dist);
                int keep = triangle ? (h.dist < max_dist) : (h.dist <= max_dist); /* :232 / :624 */
                if (keep && hv_push(&row_hits[i], &h)) fail = 1;
            }
        }
        free(row);
    }
    free(offset);
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_query; i++) total += row_hits[i].n;
    ok_hit_t *out = (ok_hit_t *)malloc((total + 1) * sizeof(ok_hit_t));
    if (!out) fail = 1;
    uint64_t p = 0;
    for (uint32_t i = 0; i < n_query; i++) {
        if (!fail && row_hits[i].n) memcpy(out + p, row_hits[i].v, row_hits[i].n * sizeof(ok_hit_t));
        p += row_hits[i].n;
        free(row_hits[i].v);
    }
    free(row_hits);
    if (fail) { free(out); return -3; }
    *hits_out = out;
    return (int64_t)total;
}

/* ------------------------------------------------------------------ D5 */
/* std::priority_queue<DistInfo, vector, cmpDistInfo> (src/dist.h:19-32) restated
 * with libstdc++'s push_heap/pop_heap sift order so ties fall the same way. */
static int heap_less(const ok_hit_t *a, const ok_hit_t *b) { return a->dist < b->dist; }

static void heap_push_at(ok_hit_t *first, int64_t hole, int64_t top, ok_hit_t value)
{
    int64_t parent = (hole - 1) / 2;
    while (hole > top && heap_less(&first[parent], &value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

static void heap_adjust(ok_hit_t *first, int64_t hole, int64_t len, ok_hit_t value)
{
    const int64_t top = hole;
    int64_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (heap_less(&first[child], &first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    heap_push_at(first, hole, top, value);
}

static void heap_pop(ok_hit_t *first, uint64_t *n)
{
    if (*n > 1) {
        ok_hit_t value = first[*n - 1];
        first[*n - 1] = first[0];
        heap_adjust(first, 0, (int64_t)*n - 1, value);
    }
    (*n)--;
}

uint32_t ok_topn_row(const ok_hit_t *row_hits, uint32_t n, uint64_t max_neighbor, ok_hit_t *out)
{
    ok_hit_t *heap = (ok_hit_t *)malloc(((size_t)n + 2) * sizeof(ok_hit_t));
    uint64_t hn = 0;
    for (uint32_t t = 0; t < n; t++) {
        if (hn < max_neighbor) { /* src/dist.cpp:633-635 */
            heap[hn] = row_hits[t];
            hn++;
            heap_push_at(heap, (int64_t)hn - 1, 0, heap[hn - 1]);
        } else if (hn > 0 && row_hits[t].dist < heap[0].dist) { /* :636-639 */
            heap[hn] = row_hits[t];
            hn++;
            heap_push_at(heap, (int64_t)hn - 1, 0, heap[hn - 1]);
            heap_pop(heap, &hn);
        }
    }
    uint32_t k = 0;
    while (hn) { /* :684-688, top() first == largest distance first */
        out[k++] = heap[0];
        heap_pop(heap, &hn);
    }
    free(heap);
    return k;
}

/* ------------------------------------------------------------------ D6 */
int ok_format_hit(char *buf, size_t cap, const char *name_a, const char *name_b, int common,
                  int size0, int size1, double jorc, double dist)
{
    return snprintf(buf, cap, "%s\t%s\t%d|%d|%d\t%f\t%f\n", name_a, name_b, common, size0, size1,
                    jorc, dist);
}
