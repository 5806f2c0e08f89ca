/* TEST INFRASTRUCTURE ONLY -- CPU oracle ("Oracle B") for the k-mer bait filter.
 *
 * PARITY UNPINNED BY THE REFERENCE.  MitoFlex has no k-mer read filter
 * (SURVEY.md section 0): nothing under /root/reference computes what this file
 * computes, so there are no reference golden vectors for it.  The semantics
 * are the ones written down in oracle/kmer_bait_ref.py (rows B1-B5 of
 * SURVEY.md 8a) and this C restatement is pinned against that string-level
 * specification by tests/test_oracle_kmer.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product path (mitoflex_amd/) never links or loads it.
 *
 * Conventions taken from the reference where it has any:
 *   FASTQ = strict 4-line records, line 3 ignored, partial tail dropped, CR
 *   stripped          (filter/filter_bin/src/main.rs:287-321)
 *   gzip chosen by the ".gz" extension   (filter/filter_bin/src/helper.rs:22)
 */
#ifndef MF_KMER_BAIT_ORACLE_H
#define MF_KMER_BAIT_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Packed read set (B1).  Dense little-endian 2-bit stream: base i lives in
 * words[i>>4] bits [2*(i&15), 2*(i&15)+1]; invalid bases are stored as 0 and
 * their global base indices listed (ascending) in npos.  offsets has
 * n_reads+1 entries (base offsets).  words is padded with >= 8 zero words. */
typedef struct {
    uint32_t *words;
    uint64_t  n_words;   /* words that hold bases (padding not counted) */
    uint64_t *offsets;
    uint64_t  n_reads;
    uint64_t *npos;
    uint64_t  n_npos;
} mfo_reads;

/* Bait table (B3/B5).  kw = 1 for k<=32, 2 for 33<=k<=63.  keys holds
 * slots*kw u64 (lo word first); empty slot = all ones.  Layout = distinct
 * canonical keys inserted in ascending order with plain linear probing from
 * mf_hash(key) & (slots-1); slots = pow2 >= max(1024, 2*n_windows) where
 * n_windows = sum over records of max(0, len-k+1). */
typedef struct {
    int       k;
    int       kw;
    uint64_t  slots;
    uint64_t  n_keys;
    uint64_t *keys;
} mfo_table;

int  mfo_pack_seqs(const char *concat, const uint64_t *seq_offsets, uint64_t n, mfo_reads *out);
int  mfo_pack_fastq(const char *path, mfo_reads *out);
void mfo_reads_free(mfo_reads *r);

int  mfo_table_build(const char *fasta_text, size_t len, int k, mfo_table *out);
int  mfo_table_build_file(const char *fasta_path, int k, mfo_table *out);
int  mfo_table_contains(const mfo_table *t, uint64_t lo, uint64_t hi);
void mfo_table_free(mfo_table *t);

/* hits_out (optional, n_reads u32) gets the full hit count of every read;
 * bits_out (optional, ceil(n_reads/32) u32) bit r = hits(r) >= threshold.
 * Reads [first, first+count) only; bit/hit index is relative to `first`. */
int  mfo_filter(const mfo_table *t, const mfo_reads *r, uint64_t first, uint64_t count,
                uint32_t threshold, uint32_t *bits_out, uint32_t *hits_out, int n_threads);

/* Whole-file convenience used by the end-to-end tests: filter fq1 (and fq2 if
 * non-NULL, pair mode 0 = either / 1 = both) against bait FASTA, write
 * survivors (header/seq/+/qual) to out1/out2, return kept count in *kept. */
int  mfo_filter_fastq_files(const char *bait_fasta, int k, uint32_t threshold, int pair_mode,
                            const char *fq1, const char *fq2, const char *out1, const char *out2,
                            uint64_t *kept, uint64_t *total, int n_threads);

uint64_t mfo_hash64(uint64_t lo, uint64_t hi, int kw);

/* ---- protein-space baiting (Spec P, oracle/prot_bait_ref.py).  The table
 * holds peptide k-mers (5 bits per residue, first residue least significant; k = kp in 4..12,
 * kw = 1) in the same layout; reads are translated in six frames with NCBI genetic code
 * `genetic_code` (1, 2, 3, 4, 5, 9, 11, 13, 14, 21). */
int  mfo_ptable_build(const char *protein_fasta_text, size_t len, int kp, mfo_table *out);
int  mfo_pfilter(const mfo_table *t, const mfo_reads *r, uint64_t first, uint64_t count, uint32_t threshold,
                 int genetic_code, uint32_t *bits_out, uint32_t *hits_out, int n_threads);
int  mfo_pfilter_fastq_files(const char *protein_fasta, int kp, int genetic_code, uint32_t threshold, int pair_mode,
                             const char *fq1, const char *fq2, const char *out1, const char *out2,
                             uint64_t *kept, uint64_t *total, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
