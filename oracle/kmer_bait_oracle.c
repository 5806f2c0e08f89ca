/* TEST INFRASTRUCTURE ONLY -- see kmer_bait_oracle.h for the header comment.
 * PARITY UNPINNED BY THE REFERENCE (no k-mer filter exists in MitoFlex);
 * pinned instead against oracle/kmer_bait_ref.py by tests/test_oracle_kmer.py.
 *
 * Deliberately the plain algorithm: one read at a time, rolling forward and
 * reverse-complement words, min(), hash, linear probe, count.  No screening,
 * no SIMD.  Threads only split the read range.
 */
#define _GNU_SOURCE
#include "kmer_bait_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

typedef unsigned __int128 u128;

/* ---------------------------------------------------------------- alphabet */
/* B1: A/a=0 C/c=1 G/g=2 T/t=3, everything else invalid (4). */
static inline int base_code(unsigned char c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

/* -------------------------------------------------------------------- hash */
/* Table hash (B3): two 32-bit multiplies per 64-bit word, folded, xor-shift finish. */
static inline uint32_t fold32(uint64_t x)
{
    const uint32_t a = (uint32_t)x * 0x9E3779B1u, b = (uint32_t)(x >> 32) * 0x85EBCA77u;
    uint32_t h = a ^ ((b << 15) | (b >> 17));
    return h ^ (h >> 15);
}

uint64_t mfo_hash64(uint64_t lo, uint64_t hi, int kw)
{
    if (kw == 1) return fold32(lo);
    uint32_t h = fold32(lo) ^ (fold32(hi) * 0xC2B2AE3Du);
    return h ^ (h >> 16);
}

/* -------------------------------------------------------------- file input */
static int has_gz_ext(const char *path)
{   /* reference rule: Path::extension() == "gz" (filter_bin/src/helper.rs:22) */
    const char *slash = strrchr(path, '/');
    const char *name = slash ? slash + 1 : path;
    const char *dot = strrchr(name, '.');
    return dot && dot != name && strcmp(dot, ".gz") == 0;
}

static int slurp(const char *path, char **buf_out, size_t *len_out)
{
    size_t cap = 1 << 20, len = 0;
    char *buf = (char *)malloc(cap);
    if (!buf) return -1;
    if (has_gz_ext(path)) {
        gzFile g = gzopen(path, "rb");
        if (!g) { free(buf); return -2; }
        for (;;) {
            if (cap - len < (1 << 19)) { cap *= 2; buf = (char *)realloc(buf, cap); if (!buf) return -1; }
            int n = gzread(g, buf + len, (unsigned)(cap - len > (1u << 30) ? (1u << 30) : cap - len));
            if (n < 0) { gzclose(g); free(buf); return -3; }
            if (n == 0) break;
            len += (size_t)n;
        }
        gzclose(g);
    } else {
        FILE *f = fopen(path, "rb");
        if (!f) { free(buf); return -2; }
        for (;;) {
            if (cap - len < (1 << 19)) { cap *= 2; buf = (char *)realloc(buf, cap); if (!buf) return -1; }
            size_t n = fread(buf + len, 1, cap - len, f);
            if (n == 0) break;
            len += n;
        }
        fclose(f);
    }
    *buf_out = buf; *len_out = len;
    return 0;
}

/* ----------------------------------------------------------------- packing */
int mfo_pack_seqs(const char *concat, const uint64_t *so, uint64_t n, mfo_reads *out)
{
    memset(out, 0, sizeof *out);
    uint64_t total = so[n] - so[0];
    uint64_t n_words = (total + 15) / 16;
    out->words = (uint32_t *)calloc(n_words + 8, sizeof(uint32_t));
    out->offsets = (uint64_t *)malloc((n + 1) * sizeof(uint64_t));
    uint64_t ncap = 1024; out->npos = (uint64_t *)malloc(ncap * sizeof(uint64_t));
    if (!out->words || !out->offsets || !out->npos) return -1;
    out->n_words = n_words; out->n_reads = n;
    uint64_t g = 0;
    for (uint64_t r = 0; r < n; r++) {
        out->offsets[r] = g;
        for (uint64_t i = so[r]; i < so[r + 1]; i++, g++) {
            int c = base_code((unsigned char)concat[i]);
            if (c == 4) {
                if (out->n_npos == ncap) { ncap *= 2; out->npos = (uint64_t *)realloc(out->npos, ncap * sizeof(uint64_t)); }
                out->npos[out->n_npos++] = g;
            } else {
                out->words[g >> 4] |= (uint32_t)c << (2 * (g & 15));
            }
        }
    }
    out->offsets[n] = g;
    return 0;
}

/* strict 4-line FASTQ -> arrays of line starts/lengths */
typedef struct { const char *h, *s, *q; uint32_t hl, sl, ql; } fq_rec;

static uint64_t parse_fastq(const char *buf, size_t len, fq_rec **recs_out)
{
    uint64_t cap = 1024, n = 0;
    fq_rec *recs = (fq_rec *)malloc(cap * sizeof *recs);
    const char *p = buf, *end = buf + len;
    const char *ls[4]; uint32_t ll[4]; int li = 0;
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;          /* last line may lack LF */
        size_t L = (size_t)(le - p);
        if (L && p[L - 1] == '\r') L--;          /* lines() strips \r\n */
        ls[li] = p; ll[li] = (uint32_t)L; li++;
        if (li == 4) {
            if (n == cap) { cap *= 2; recs = (fq_rec *)realloc(recs, cap * sizeof *recs); }
            recs[n].h = ls[0]; recs[n].hl = ll[0];
            recs[n].s = ls[1]; recs[n].sl = ll[1];
            recs[n].q = ls[3]; recs[n].ql = ll[3];
            n++; li = 0;
        }
        if (!nl) break;
        p = nl + 1;
    }
    *recs_out = recs;
    return n;
}

static int pack_recs(const fq_rec *recs, uint64_t n, mfo_reads *out)
{
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; i++) total += recs[i].sl;
    char *concat = (char *)malloc(total + 1);
    uint64_t *so = (uint64_t *)malloc((n + 1) * sizeof(uint64_t));
    if (!concat || !so) return -1;
    uint64_t g = 0;
    for (uint64_t i = 0; i < n; i++) { so[i] = g; memcpy(concat + g, recs[i].s, recs[i].sl); g += recs[i].sl; }
    so[n] = g;
    int rc = mfo_pack_seqs(concat, so, n, out);
    free(concat); free(so);
    return rc;
}

int mfo_pack_fastq(const char *path, mfo_reads *out)
{
    char *buf; size_t len;
    int rc = slurp(path, &buf, &len);
    if (rc) return rc;
    fq_rec *recs; uint64_t n = parse_fastq(buf, len, &recs);
    rc = pack_recs(recs, n, out);
    free(recs); free(buf);
    return rc;
}

void mfo_reads_free(mfo_reads *r)
{
    free(r->words); free(r->offsets); free(r->npos);
    memset(r, 0, sizeof *r);
}

/* ------------------------------------------------------------------- table */
static int cmp_u128(const void *a, const void *b)
{
    u128 x = *(const u128 *)a, y = *(const u128 *)b;
    return x < y ? -1 : x > y;
}

static inline u128 kmask(int k) { return k == 64 ? ~(u128)0 : (((u128)1 << (2 * k)) - 1); }

int mfo_table_build(const char *text, size_t len, int k, mfo_table *out)
{
    memset(out, 0, sizeof *out);
    if (k < 11 || k > 63) return -10;
    /* B5: records, windows never span records, invalid letters break windows */
    uint64_t cap = 1024, nk = 0, n_windows = 0;
    u128 *keys = (u128 *)malloc(cap * sizeof *keys);
    u128 fwd = 0, rc = 0; const u128 M = kmask(k);
    uint64_t run = 0, reclen = 0;
    size_t i = 0;
    int at_line_start = 1, in_header = 0;
    for (; i <= len; i++) {
        int eof = (i == len);
        unsigned char c = eof ? '\n' : (unsigned char)text[i];
        if (in_header) { if (c == '\n') { in_header = 0; at_line_start = 1; } continue; }
        if (at_line_start && c == '>' && !eof) {
            /* close previous record */
            if (reclen >= (uint64_t)k) n_windows += reclen - k + 1;
            reclen = 0; run = 0; fwd = rc = 0; in_header = 1; at_line_start = 0;
            continue;
        }
        if (c == '\n') { at_line_start = 1; continue; }
        at_line_start = 0;
        if (c == '\r' || c == ' ' || c == '\t' || c == '\v' || c == '\f') continue;
        reclen++;
        int b = base_code(c);
        if (b == 4) { run = 0; fwd = rc = 0; continue; }
        fwd = (fwd >> 2) | ((u128)b << (2 * (k - 1)));
        rc = ((rc << 2) | (u128)(3 - b)) & M;
        if (++run >= (uint64_t)k) {
            u128 can = fwd < rc ? fwd : rc;
            if (nk == cap) { cap *= 2; keys = (u128 *)realloc(keys, cap * sizeof *keys); }
            keys[nk++] = can;
        }
    }
    if (reclen >= (uint64_t)k) n_windows += reclen - k + 1;

    qsort(keys, nk, sizeof *keys, cmp_u128);
    uint64_t nu = 0;
    for (uint64_t j = 0; j < nk; j++) if (j == 0 || keys[j] != keys[j - 1]) keys[nu++] = keys[j];

    uint64_t slots = 1024;
    while (slots < 2 * n_windows) slots <<= 1;
    int kw = k > 32 ? 2 : 1;
    uint64_t *tab = (uint64_t *)malloc(slots * kw * sizeof(uint64_t));
    if (!tab) return -1;
    memset(tab, 0xFF, slots * kw * sizeof(uint64_t));
    for (uint64_t j = 0; j < nu; j++) {           /* ascending order, plain linear probing */
        uint64_t lo = (uint64_t)keys[j], hi = (uint64_t)(keys[j] >> 64);
        uint64_t s = mfo_hash64(lo, hi, kw) & (slots - 1);
        for (;;) {
            int empty = tab[s * kw] == ~0ULL && (kw == 1 || tab[s * kw + 1] == ~0ULL);
            if (empty) { tab[s * kw] = lo; if (kw == 2) tab[s * kw + 1] = hi; break; }
            s = (s + 1) & (slots - 1);
        }
    }
    free(keys);
    out->k = k; out->kw = kw; out->slots = slots; out->n_keys = nu; out->keys = tab;
    return 0;
}

int mfo_table_build_file(const char *path, int k, mfo_table *out)
{
    char *buf; size_t len;
    int rc = slurp(path, &buf, &len);
    if (rc) return rc;
    rc = mfo_table_build(buf, len, k, out);
    free(buf);
    return rc;
}

int mfo_table_contains(const mfo_table *t, uint64_t lo, uint64_t hi)
{
    const int kw = t->kw;
    uint64_t s = mfo_hash64(lo, hi, kw) & (t->slots - 1);
    for (;;) {
        uint64_t a = t->keys[s * kw], b = kw == 2 ? t->keys[s * kw + 1] : 0;
        if (a == ~0ULL && (kw == 1 || b == ~0ULL)) return 0;
        if (a == lo && (kw == 1 || b == hi)) return 1;
        s = (s + 1) & (t->slots - 1);
    }
}

void mfo_table_free(mfo_table *t) { free(t->keys); memset(t, 0, sizeof *t); }

/* ------------------------------------------------------------------ filter */
static inline int get_base(const mfo_reads *r, uint64_t g)
{
    return (int)((r->words[g >> 4] >> (2 * (g & 15))) & 3u);
}

/* first index in npos with npos[i] >= g */
static uint64_t npos_lower_bound(const mfo_reads *r, uint64_t g)
{
    uint64_t lo = 0, hi = r->n_npos;
    while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (r->npos[mid] < g) lo = mid + 1; else hi = mid; }
    return lo;
}

static uint32_t read_hits(const mfo_table *t, const mfo_reads *r, uint64_t idx)
{
    const int k = t->k; const u128 M = kmask(k);
    uint64_t b0 = r->offsets[idx], b1 = r->offsets[idx + 1];
    uint64_t ni = npos_lower_bound(r, b0);
    u128 fwd = 0, rc = 0; uint64_t run = 0; uint32_t hits = 0;
    for (uint64_t g = b0; g < b1; g++) {
        if (ni < r->n_npos && r->npos[ni] == g) { ni++; run = 0; fwd = rc = 0; continue; }
        int b = get_base(r, g);
        fwd = (fwd >> 2) | ((u128)b << (2 * (k - 1)));
        rc = ((rc << 2) | (u128)(3 - b)) & M;
        if (++run >= (uint64_t)k) {
            u128 can = fwd < rc ? fwd : rc;
            hits += (uint32_t)mfo_table_contains(t, (uint64_t)can, (uint64_t)(can >> 64));
        }
    }
    return hits;
}

/* same loop with 64-bit words for k <= 32 (the common case; identical results) */
static uint32_t read_hits64(const mfo_table *t, const mfo_reads *r, uint64_t idx)
{
    const int k = t->k; const uint64_t M = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
    uint64_t b0 = r->offsets[idx], b1 = r->offsets[idx + 1];
    uint64_t ni = npos_lower_bound(r, b0);
    uint64_t fwd = 0, rc = 0, run = 0; uint32_t hits = 0;
    for (uint64_t g = b0; g < b1; g++) {
        if (ni < r->n_npos && r->npos[ni] == g) { ni++; run = 0; fwd = rc = 0; continue; }
        uint64_t b = (uint64_t)get_base(r, g);
        fwd = (fwd >> 2) | (b << (2 * (k - 1)));
        rc = ((rc << 2) | (3 - b)) & M;
        if (++run >= (uint64_t)k) hits += (uint32_t)mfo_table_contains(t, fwd < rc ? fwd : rc, 0);
    }
    return hits;
}

typedef struct {
    const mfo_table *t; const mfo_reads *r; uint64_t first, lo, hi; uint32_t thr;
    uint32_t *bits; uint32_t *hits;
    const int8_t *plut;          /* non-NULL: protein-space baiting with this codon -> residue table */
} job_t;

static uint32_t pread_hits(const mfo_table *t, const mfo_reads *r, uint64_t idx, const int8_t *lut);

static void *filter_job(void *arg)
{
    job_t *j = (job_t *)arg;
    for (uint64_t i = j->lo; i < j->hi; i++) {
        uint32_t h = j->plut ? pread_hits(j->t, j->r, j->first + i, j->plut)
                   : j->t->kw == 1 ? read_hits64(j->t, j->r, j->first + i) : read_hits(j->t, j->r, j->first + i);
        if (j->hits) j->hits[i] = h;
        if (j->bits && h >= j->thr) __atomic_fetch_or(&j->bits[i >> 5], 1u << (i & 31), __ATOMIC_RELAXED);
    }
    return NULL;
}

static int filter_impl(const mfo_table *t, const mfo_reads *r, uint64_t first, uint64_t count,
                       uint32_t thr, uint32_t *bits, uint32_t *hits, int n_threads, const int8_t *plut)
{
    if (thr < 1) return -11;
    if (first + count > r->n_reads) return -12;
    if (bits) memset(bits, 0, ((count + 31) / 32) * sizeof(uint32_t));
    if (n_threads < 1) n_threads = 1;
    if ((uint64_t)n_threads > count) n_threads = count ? (int)count : 1;
    pthread_t *th = (pthread_t *)malloc(n_threads * sizeof *th);
    job_t *jobs = (job_t *)malloc(n_threads * sizeof *jobs);
    for (int i = 0; i < n_threads; i++) {
        jobs[i] = (job_t){ t, r, first, count * i / n_threads, count * (i + 1) / n_threads, thr, bits, hits, plut };
        if (n_threads == 1) filter_job(&jobs[i]);
        else pthread_create(&th[i], NULL, filter_job, &jobs[i]);
    }
    if (n_threads > 1) for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
    free(th); free(jobs);
    return 0;
}

int mfo_filter(const mfo_table *t, const mfo_reads *r, uint64_t first, uint64_t count,
               uint32_t thr, uint32_t *bits, uint32_t *hits, int n_threads)
{
    return filter_impl(t, r, first, count, thr, bits, hits, n_threads, NULL);
}

/* -------------------------------------------------------------- whole files */
static int write_survivors(const char *path, const fq_rec *recs, uint64_t n, const uint8_t *keep)
{
    int gz = has_gz_ext(path);
    gzFile g = NULL; FILE *f = NULL;
    if (gz) { g = gzopen(path, "wb6"); if (!g) return -2; }
    else { f = fopen(path, "wb"); if (!f) return -2; }
#define PUT(ptr, n_) do { if (gz) gzwrite(g, ptr, (unsigned)(n_)); else fwrite(ptr, 1, n_, f); } while (0)
    for (uint64_t i = 0; i < n; i++) {
        if (!keep[i]) continue;
        PUT(recs[i].h, recs[i].hl); PUT("\n", 1);
        PUT(recs[i].s, recs[i].sl); PUT("\n+\n", 3);
        PUT(recs[i].q, recs[i].ql); PUT("\n", 1);
    }
#undef PUT
    if (gz) gzclose(g); else fclose(f);
    return 0;
}

static int files_impl(mfo_table t, const int8_t *plut, uint32_t thr, int pair_mode,
                      const char *fq1, const char *fq2, const char *out1, const char *out2,
                      uint64_t *kept, uint64_t *total, int n_threads)
{
    int rc;
    char *b1 = NULL, *b2 = NULL; size_t l1 = 0, l2 = 0;
    fq_rec *r1 = NULL, *r2 = NULL; uint64_t n1 = 0, n2 = 0;
    if ((rc = slurp(fq1, &b1, &l1))) return rc;
    n1 = parse_fastq(b1, l1, &r1);
    if (fq2) { if ((rc = slurp(fq2, &b2, &l2))) return rc; n2 = parse_fastq(b2, l2, &r2); }
    /* PE: records are zipped, the shorter file bounds the pair count
     * (filter_bin/src/main.rs:214 `fq1.lines().zip(fq2.lines())`) */
    uint64_t n = fq2 ? (n1 < n2 ? n1 : n2) : n1;
    uint8_t *keep = (uint8_t *)calloc(n ? n : 1, 1);
    uint32_t *h1 = (uint32_t *)malloc((n ? n : 1) * 4), *h2 = (uint32_t *)malloc((n ? n : 1) * 4);
    mfo_reads p1, p2;
    pack_recs(r1, n, &p1);
    filter_impl(&t, &p1, 0, n, thr, NULL, h1, n_threads, plut);
    if (fq2) { pack_recs(r2, n, &p2); filter_impl(&t, &p2, 0, n, thr, NULL, h2, n_threads, plut); }
    uint64_t kc = 0;
    for (uint64_t i = 0; i < n; i++) {
        int a = h1[i] >= thr, b = fq2 ? h2[i] >= thr : 0;
        keep[i] = fq2 ? (pair_mode == 1 ? (a && b) : (a || b)) : a;
        kc += keep[i];
    }
    rc = write_survivors(out1, r1, n, keep);
    if (!rc && fq2) rc = write_survivors(out2, r2, n, keep);
    if (kept) *kept = kc;
    if (total) *total = n;
    mfo_reads_free(&p1); if (fq2) mfo_reads_free(&p2);
    free(keep); free(h1); free(h2); free(r1); free(r2); free(b1); free(b2);
    mfo_table_free(&t);
    return rc;
}

int mfo_filter_fastq_files(const char *bait, int k, uint32_t thr, int pair_mode,
                           const char *fq1, const char *fq2, const char *out1, const char *out2,
                           uint64_t *kept, uint64_t *total, int n_threads)
{
    mfo_table t; int rc = mfo_table_build_file(bait, k, &t);
    if (rc) return rc;
    return files_impl(t, NULL, thr, pair_mode, fq1, fq2, out1, out2, kept, total, n_threads);
}

/* ---------------------------------------------------- protein-space baiting
 * Spec P of oracle/prot_bait_ref.py (PARITY UNPINNED BY THE REFERENCE, which
 * only ever hands profile/MT_database to tblastn: annotation/annotation_tookit.py:55-97).  Written
 * the long way round on purpose: the reverse strand is built explicitly and translated like the
 * forward one. */
static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

static const char *gcode_string(int code)
{
    switch (code) {   /* NCBI transl_table, codons ordered TTT TTC TTA TTG TCT ... GGG */
    case 1: case 11: return "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    case 2:  return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSS**VVVVAAAADDEEGGGG";
    case 3:  return "FFLLSSSSYY**CCWWTTTTPPPPHHQQRRRRIIMMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    case 4:  return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    case 5:  return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSSSVVVVAAAADDEEGGGG";
    case 9:  return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG";
    case 13: return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSGGVVVVAAAADDEEGGGG";
    case 14: return "FFLLSSSSYYY*CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG";
    case 21: return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNNKSSSSVVVVAAAADDEEGGGG";
    default: return NULL;
    }
}

static int aa_code(unsigned char c)
{
    static const char *AA = "ACDEFGHIKLMNPQRSTVWY";
    if (c >= 'a' && c <= 'z') c = (unsigned char)(c - 32);
    const char *p = c ? strchr(AA, c) : NULL;
    return p ? (int)(p - AA) : -1;
}

/* lut[b1*16 + b2*4 + b3] with A=0 C=1 G=2 T=3: residue code, or -1 for a stop */
static int codon_lut(int gcode, int8_t lut[64])
{
    const char *tab = gcode_string(gcode);
    if (!tab) return -13;
    static const int tcag_of[4] = {2, 1, 3, 0};          /* A C G T -> index in "TCAG" */
    for (int b1 = 0; b1 < 4; b1++) for (int b2 = 0; b2 < 4; b2++) for (int b3 = 0; b3 < 4; b3++)
        lut[b1 * 16 + b2 * 4 + b3] = (int8_t)aa_code((unsigned char)tab[16 * tcag_of[b1] + 4 * tcag_of[b2] + tcag_of[b3]]);
    return 0;
}

int mfo_ptable_build(const char *text, size_t len, int kp, mfo_table *out)
{
    memset(out, 0, sizeof *out);
    if (kp < 4 || kp > 12) return -10;
    uint64_t cap = 1024, nk = 0, n_windows = 0, run = 0, reclen = 0, key = 0;
    uint64_t *keys = (uint64_t *)malloc(cap * sizeof *keys);
    int at_line_start = 1, in_header = 0;
    for (size_t i = 0; i <= len; i++) {
        int eof = (i == len);
        unsigned char c = eof ? '\n' : (unsigned char)text[i];
        if (in_header) { if (c == '\n') { in_header = 0; at_line_start = 1; } continue; }
        if (at_line_start && c == '>' && !eof) {
            if (reclen >= (uint64_t)kp) n_windows += reclen - kp + 1;
            reclen = 0; run = 0; key = 0; in_header = 1; at_line_start = 0;
            continue;
        }
        if (c == '\n') { at_line_start = 1; continue; }
        at_line_start = 0;
        if (c == '\r' || c == ' ' || c == '\t' || c == '\v' || c == '\f') continue;
        reclen++;
        int a = aa_code(c);
        if (a < 0) { run = 0; key = 0; continue; }
        key = (key >> 5) | ((uint64_t)a << (5 * (kp - 1)));
        if (++run >= (uint64_t)kp) {
            if (nk == cap) { cap *= 2; keys = (uint64_t *)realloc(keys, cap * sizeof *keys); }
            keys[nk++] = key;
        }
    }
    if (reclen >= (uint64_t)kp) n_windows += reclen - kp + 1;
    qsort(keys, nk, sizeof *keys, cmp_u64);
    uint64_t nu = 0;
    for (uint64_t j = 0; j < nk; j++) if (j == 0 || keys[j] != keys[j - 1]) keys[nu++] = keys[j];
    uint64_t slots = 1024;
    while (slots < 2 * n_windows) slots <<= 1;
    uint64_t *tab = (uint64_t *)malloc(slots * sizeof(uint64_t));
    if (!tab) return -1;
    memset(tab, 0xFF, slots * sizeof(uint64_t));
    for (uint64_t j = 0; j < nu; j++) {
        uint64_t s = mfo_hash64(keys[j], 0, 1) & (slots - 1);
        while (tab[s] != ~0ULL) s = (s + 1) & (slots - 1);
        tab[s] = keys[j];
    }
    free(keys);
    out->k = kp; out->kw = 1; out->slots = slots; out->n_keys = nu; out->keys = tab;
    return 0;
}

/* hits of one strand given as base codes (0..3, 4 = invalid) */
static uint32_t strand_hits(const mfo_table *t, const uint8_t *b, uint64_t L, const int8_t *lut)
{
    const int kp = t->k; uint32_t hits = 0;
    for (uint64_t off = 0; off < 3; off++) {
        uint64_t run = 0, key = 0;
        for (uint64_t p = off; p + 3 <= L; p += 3) {
            int a = (b[p] > 3 || b[p + 1] > 3 || b[p + 2] > 3) ? -1 : lut[b[p] * 16 + b[p + 1] * 4 + b[p + 2]];
            if (a < 0) { run = 0; key = 0; continue; }
            key = (key >> 5) | ((uint64_t)a << (5 * (kp - 1)));
            if (++run >= (uint64_t)kp) hits += (uint32_t)mfo_table_contains(t, key, 0);
        }
    }
    return hits;
}

static uint32_t pread_hits(const mfo_table *t, const mfo_reads *r, uint64_t idx, const int8_t *lut)
{
    uint64_t b0 = r->offsets[idx], b1 = r->offsets[idx + 1], L = b1 - b0;
    uint8_t stack_f[512], stack_r[512];
    uint8_t *f = L <= 512 ? stack_f : (uint8_t *)malloc(L), *rv = L <= 512 ? stack_r : (uint8_t *)malloc(L);
    uint64_t ni = npos_lower_bound(r, b0);
    for (uint64_t g = b0; g < b1; g++) {
        if (ni < r->n_npos && r->npos[ni] == g) { ni++; f[g - b0] = 4; }
        else f[g - b0] = (uint8_t)get_base(r, g);
    }
    for (uint64_t i = 0; i < L; i++) rv[i] = f[L - 1 - i] > 3 ? 4 : (uint8_t)(3 - f[L - 1 - i]);
    uint32_t h = strand_hits(t, f, L, lut) + strand_hits(t, rv, L, lut);
    if (L > 512) { free(f); free(rv); }
    return h;
}

int mfo_pfilter(const mfo_table *t, const mfo_reads *r, uint64_t first, uint64_t count, uint32_t thr,
                int genetic_code, uint32_t *bits, uint32_t *hits, int n_threads)
{
    int8_t lut[64];
    int rc = codon_lut(genetic_code, lut);
    if (rc) return rc;
    return filter_impl(t, r, first, count, thr, bits, hits, n_threads, lut);
}

int mfo_pfilter_fastq_files(const char *bait, int kp, int genetic_code, uint32_t thr, int pair_mode,
                            const char *fq1, const char *fq2, const char *out1, const char *out2,
                            uint64_t *kept, uint64_t *total, int n_threads)
{
    int8_t lut[64];
    int rc = codon_lut(genetic_code, lut);
    if (rc) return rc;
    char *buf; size_t len;
    if ((rc = slurp(bait, &buf, &len))) return rc;
    mfo_table t; rc = mfo_ptable_build(buf, len, kp, &t);
    free(buf);
    if (rc) return rc;
    return files_impl(t, lut, thr, pair_mode, fq1, fq2, out1, out2, kept, total, n_threads);
}
