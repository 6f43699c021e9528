/*
 * kssd_oracle64.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see kssd_oracle.h).
 * The 64-bit hash layout (use64 = half_k - drlevel > 8, src/sketch.cpp:336): .sketch with u64
 * hashes, the sparse .index variant and the map-based counting of the distance path.
 */
#define _GNU_SOURCE
#include "kssd_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* src/sketch.cpp:1024-1068, use64 branch (:1055-1058) */
int ok_save_sketches64(const char *path, const ok_sketch_info_t *info_in, const char *const *names,
                       const uint64_t *hashes, const uint64_t *off)
{
    ok_sketch_info_t info = *info_in;
    int n = info.genomeNumber;
    info.id = (info.half_k << 8) + (info.half_subk << 4) + info.drlevel;
    FILE *fp = fopen(path, "wb");
    if (!fp) return -1;
    fwrite(&info, sizeof(info), 1, fp);
    int32_t *len = (int32_t *)malloc(((size_t)n + 1) * 4), *cnt = (int32_t *)malloc(((size_t)n + 1) * 4);
    for (int i = 0; i < n; i++) {
        len[i] = (int32_t)strlen(names[i]);
        cnt[i] = (int32_t)(off[i + 1] - off[i]);
    }
    fwrite(len, 4, (size_t)n, fp);
    fwrite(cnt, 4, (size_t)n, fp);
    for (int i = 0; i < n; i++) {
        fwrite(names[i], 1, (size_t)len[i], fp);
        fwrite(hashes + off[i], 8, (size_t)cnt[i], fp);
    }
    free(len);
    free(cnt);
    return fclose(fp) == 0 ? 0 : -4;
}

/* src/sketch.cpp:1070-1154, use64 branch (:1128-1136) */
int ok_read_sketches64(const char *path, ok_sketch_info_t *info, char **names_blob, uint64_t **hashes_out,
                       uint64_t **off_out)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return -1;
    if (fread(info, sizeof(*info), 1, fp) != 1) { fclose(fp); return -2; }
    if (info->half_k - info->drlevel <= 8) { fclose(fp); return -5; } /* 32-bit layout */
    size_t n = (size_t)info->genomeNumber;
    int32_t *len = (int32_t *)malloc((n + 1) * 4), *cnt = (int32_t *)malloc((n + 1) * 4);
    if (fread(len, 4, n, fp) != n || fread(cnt, 4, n, fp) != n) { fclose(fp); return -2; }
    uint64_t tot_name = 0, tot_hash = 0;
    for (size_t i = 0; i < n; i++) { tot_name += (uint64_t)len[i] + 1; tot_hash += (uint64_t)cnt[i]; }
    char *blob = (char *)malloc(tot_name + 1);
    uint64_t *h = (uint64_t *)malloc((tot_hash + 1) * 8);
    uint64_t *off = (uint64_t *)malloc((n + 1) * 8);
    uint64_t np = 0;
    int rc = 0;
    off[0] = 0;
    for (size_t i = 0; i < n; i++) {
        if (fread(blob + np, 1, (size_t)len[i], fp) != (size_t)len[i]) { rc = -2; break; }
        np += (uint64_t)len[i];
        blob[np++] = 0;
        if (fread(h + off[i], 8, (size_t)cnt[i], fp) != (size_t)cnt[i]) { rc = -2; break; }
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

typedef struct { uint64_t h; uint32_t g; } pair64_t;
static int cmp_pair(const void *a, const void *b)
{
    const pair64_t *x = (const pair64_t *)a, *y = (const pair64_t *)b;
    if (x->h != y->h) return x->h < y->h ? -1 : 1;
    return x->g < y->g ? -1 : x->g > y->g;
}

/* I2: src/sketch.cpp:904-969.  hash_map_arr[cur_hash].push_back(i) for i ascending; the reference
 * writes the postings in hash-map iteration order, which no reader depends on (the .index file
 * carries the hash of every posting block).  The restatement uses ascending hash order. */
int ok_index_build64(const uint64_t *hashes, const uint64_t *off, uint32_t n_genomes, uint64_t **uhash_out,
                     uint32_t **ucount_out, uint32_t **postings_out, uint64_t *n_hash_out, uint64_t *total_out)
{
    uint64_t H = off[n_genomes];
    pair64_t *p = (pair64_t *)malloc((H + 1) * sizeof(pair64_t));
    if (!p) return -3;
    for (uint32_t g = 0; g < n_genomes; g++)
        for (uint64_t e = off[g]; e < off[g + 1]; e++) { p[e].h = hashes[e]; p[e].g = g; }
    if (H) qsort(p, H, sizeof(pair64_t), cmp_pair);
    uint64_t U = 0;
    for (uint64_t i = 0; i < H; i++)
        if (i == 0 || p[i].h != p[i - 1].h) U++;
    uint64_t *uh = (uint64_t *)malloc((U + 1) * 8);
    uint32_t *uc = (uint32_t *)calloc(U + 1, 4), *post = (uint32_t *)malloc((H + 1) * 4);
    uint64_t u = 0;
    for (uint64_t i = 0; i < H; i++) {
        if (i == 0 || p[i].h != p[i - 1].h) uh[u++] = p[i].h;
        uc[u - 1]++;
        post[i] = p[i].g;
    }
    free(p);
    *uhash_out = uh;
    *ucount_out = uc;
    *postings_out = post;
    *n_hash_out = U;
    *total_out = H;
    return 0;
}

/* .index = {size_t hash_number; u64 hash[n]; u32 count[n]} (src/sketch.cpp:961-963),
 * .dict = posting blocks in the same order (:942-948). */
int ok_write_index64(const char *dict_path, const char *index_path, const uint32_t *postings, const uint64_t *uhash,
                     const uint32_t *ucount, uint64_t n_hash, uint64_t total)
{
    FILE *fd = fopen(dict_path, "wb");
    if (!fd) return -1;
    size_t w = fwrite(postings, 4, (size_t)total, fd);
    if (fclose(fd) || w != total) return -4;
    FILE *fi = fopen(index_path, "wb");
    if (!fi) return -1;
    fwrite(&n_hash, 8, 1, fi);
    fwrite(uhash, 8, (size_t)n_hash, fi);
    w = fwrite(ucount, 4, (size_t)n_hash, fi);
    if (fclose(fi) || w != n_hash) return -4;
    return 0;
}

/* src/dist.cpp:36-82 */
int ok_read_index64(const char *dict_path, const char *index_path, uint32_t **postings_out, uint64_t **uhash_out,
                    uint32_t **ucount_out, uint64_t *n_hash_out, uint64_t *total_out)
{
    FILE *fi = fopen(index_path, "rb");
    if (!fi) return -1;
    uint64_t n = 0;
    if (fread(&n, 8, 1, fi) != 1) { fclose(fi); return -2; }
    uint64_t *uh = (uint64_t *)malloc((n + 1) * 8);
    uint32_t *uc = (uint32_t *)malloc((n + 1) * 4);
    if (fread(uh, 8, (size_t)n, fi) != n || fread(uc, 4, (size_t)n, fi) != n) { fclose(fi); free(uh); free(uc); return -2; }
    fclose(fi);
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; i++) total += uc[i];
    uint32_t *post = (uint32_t *)malloc((total + 1) * 4);
    FILE *fd = fopen(dict_path, "rb");
    if (!fd) { free(uh); free(uc); free(post); return -1; }
    if (fread(post, 4, (size_t)total, fd) != total) { fclose(fd); free(uh); free(uc); free(post); return -2; }
    fclose(fd);
    *postings_out = post;
    *uhash_out = uh;
    *ucount_out = uc;
    *n_hash_out = n;
    *total_out = total;
    return 0;
}

/* D2 for use64: src/dist.cpp:181-191 / :566-576 (hash_map_arr lookup), epilogue as the 32-bit path.
 * uhash must be ascending (binary search stands in for the reference's hash map). */
int64_t ok_index_dist64(const uint64_t *uhash, const uint32_t *ucount, uint64_t n_hash, const uint32_t *postings,
                        const uint32_t *ref_sizes, uint32_t n_ref, const uint64_t *q_hashes, const uint64_t *q_off,
                        uint32_t n_query, int triangle, int metric, int kmer_size, double max_dist, int threads,
                        int32_t *common_dense, ok_hit_t **hits_out)
{
    uint64_t *upos = (uint64_t *)malloc((n_hash + 1) * 8);
    if (!upos) return -3;
    upos[0] = 0;
    for (uint64_t i = 0; i < n_hash; i++) upos[i + 1] = upos[i] + ucount[i];
    if (threads < 1) threads = 1;
    typedef struct { ok_hit_t *v; uint64_t n, cap; } hv_t;
    hv_t *rows = (hv_t *)calloc(n_query ? n_query : 1, sizeof(hv_t));
    int fail = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
    {
        int32_t *row = (int32_t *)malloc(((size_t)n_ref + 1) * 4);
#ifdef _OPENMP
#pragma omp for schedule(dynamic)
#endif
        for (int64_t i = 0; i < (int64_t)n_query; i++) {
            if (!row) { fail = 1; continue; }
            memset(row, 0, (size_t)n_ref * 4);
            for (uint64_t e = q_off[i]; e < q_off[i + 1]; e++) {
                const uint64_t h = q_hashes[e];
                uint64_t lo = 0, hi = n_hash;
                while (lo < hi) {
                    uint64_t mid = (lo + hi) >> 1;
                    if (uhash[mid] < h) lo = mid + 1; else hi = mid;
                }
                if (lo >= n_hash || uhash[lo] != h) continue; /* hash_map_arr.count(hash64) == 0, :184 */
                for (uint64_t k = upos[lo]; k < upos[lo + 1]; k++) row[postings[k]]++; /* :186-189 */
            }
            if (common_dense) memcpy(common_dense + (size_t)i * n_ref, row, (size_t)n_ref * 4);
            int qsize = (int)(q_off[i + 1] - q_off[i]);
            for (uint32_t j = triangle ? (uint32_t)i + 1 : 0; j < n_ref; j++) {
                ok_hit_t h;
                h.row = (uint32_t)i;
                h.col = j;
                h.common = row[j];
                h.pad_ = 0;
                if (triangle) { h.size0 = qsize; h.size1 = (int)ref_sizes[j]; }
                else          { h.size0 = (int)ref_sizes[j]; h.size1 = qsize; }
                ok_distance(h.common, h.size0, h.size1, metric, kmer_size, &h.jorc, &h.dist);
                int keep = triangle ? (h.dist < max_dist) : (h.dist <= max_dist);
                if (!keep) continue;
                hv_t *hv = &rows[i];
                if (hv->n == hv->cap) {
                    uint64_t nc = hv->cap ? hv->cap * 2 : 256;
                    ok_hit_t *nv = (ok_hit_t *)realloc(hv->v, nc * sizeof(ok_hit_t));
                    if (!nv) { fail = 1; continue; }
                    hv->v = nv;
                    hv->cap = nc;
                }
                hv->v[hv->n++] = h;
            }
        }
        free(row);
    }
    free(upos);
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_query; i++) total += rows[i].n;
    ok_hit_t *out = (ok_hit_t *)malloc((total + 1) * sizeof(ok_hit_t));
    uint64_t p = 0;
    for (uint32_t i = 0; i < n_query; i++) {
        if (out && rows[i].n) memcpy(out + p, rows[i].v, rows[i].n * sizeof(ok_hit_t));
        p += rows[i].n;
        free(rows[i].v);
    }
    free(rows);
    if (fail || !out) { free(out); return -3; }
    *hits_out = out;
    return (int64_t)total;
}
