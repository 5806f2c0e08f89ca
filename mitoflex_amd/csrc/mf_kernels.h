// Launcher declarations for mf_kernels.hip (internal to libmitofilter_hip).
#pragma once
#include <hip/hip_runtime.h>
#include "mf_common.h"
#include "mf_kernels_cfg.h"

namespace mf {

// packed bait on the device: same 2-bit layout as reads; runlen[p] = number of
// consecutive valid bases starting at p inside its record, capped at 255
struct BaitView {
    const uint32_t *words;
    uint64_t        total;
    const uint8_t  *runlen;
};

// optional kernel-attached timing: the events receive the dispatch's own begin and end timestamps
struct KernelTiming { hipEvent_t start, stop; };

// screen = screen_kernel (records stage-1 positives) + mark_kernel (finishes them, sets candidate bits)
uint64_t screen_grid_for(const ReadsView &R, int n_cu, int stride);
uint64_t screen_rec_cap_for(const ReadsView &R, int n_cu, int stride);          // 16-byte records per screen workgroup
// clear (optional): a result bitmap of clear_vec4 uint4 that the screen zeroes on the side (for the pass after this one)
hipError_t launch_screen(const ReadsView &R, const KmerSetView &S, void *recs, uint32_t *rec_counts, int n_cu, hipStream_t st,
                         const KernelTiming *tm = nullptr, uint32_t *clear = nullptr, uint64_t clear_vec4 = 0);
// threshold 1, no hit counts: two launches of finish_kernel (runs, then the rest) settle every stage-1 record and set the pass
// bits (bits must be clean).  partials: 3 * EXACT_MAX_GRID tally pairs (phase 0, phase 1, the exact kernel behind them).
// done (optional): an event that completes with the last finish kernel
// cand (optional): a clean candidate bitmap -- phase 1 then marks the reads that hold a bait s-mer outside any run instead of counting their windows itself, and
// the caller runs launch_exact(cand, thr = 1, merge = true) behind it (third tally region)
hipError_t launch_finish(const ReadsView &R, const KmerSetView &S, const void *recs, const uint32_t *rec_counts, uint32_t *bits,
                         unsigned long long *partials, int n_cu, hipStream_t st, const KernelTiming *tm = nullptr, hipEvent_t done = nullptr, uint32_t *cand = nullptr);
hipError_t launch_mark(const ReadsView &R, const KmerSetView &S, const void *recs, const uint32_t *rec_counts, uint32_t *cand, int n_cu,
                       hipStream_t st, const KernelTiming *tm = nullptr);
hipError_t launch_exact(const ReadsView &R, const KmerSetView &S, uint32_t *cand, uint32_t thr, bool count_all,
                        uint32_t *out_bits, uint32_t *hits_out, unsigned long long *counters, int n_cu, hipStream_t st,
                        const KernelTiming *tm = nullptr, bool coresident = false, hipEvent_t done = nullptr, bool merge = false);
hipError_t launch_build_kbloom(const uint64_t *keys, uint64_t slots, int kw, uint32_t *kbloom, uint32_t kb_log2w, hipStream_t st);
hipError_t launch_build_table(const BaitView &B, int k, int kw, uint64_t *keys, uint64_t slots, uint32_t *postab_scratch,
                              hipStream_t st);
// front2 / front3 (optional): the bait-sized fronts of the large-bait screen, 1 << log2b blocks of 128 bits each, zeroed
hipError_t launch_build_screen(const BaitView &B, int s, uint32_t *bloom, uint32_t log2w, uint32_t log2w2, uint32_t *stab,
                               uint32_t stab_slots, uint32_t *has_ones, uint32_t *front2, uint32_t f2_log2b, uint32_t *front3, uint32_t f3_log2b,
                               uint32_t *pre, uint32_t pre_log2w, bool canon, hipStream_t st);          // pre (optional): mode 4's one-bit LDS table, zeroed
hipError_t launch_count_keys(const uint64_t *keys, uint64_t slots, int kw, const uint32_t *stab, uint64_t stab_slots,
                             unsigned long long *out2, hipStream_t st);
// FASTQ quality filter: a record's (cut) sequence and quality strings as offsets into the uploaded text
struct QualRec { uint32_t s_off, s_len, q_off, q_len; };
hipError_t launch_qualscan(const uint8_t *text, const QualRec *recs, uint32_t n, uint32_t quality, uint32_t *n_count, uint32_t *bad_count,
                           uint64_t *hashes /* nullptr: no dedup hashes */, hipStream_t st);
// the quality filter's de-duplication set on the device (HashSet<u64> semantics, first occurrence by file index wins): keys and
// first have `slots` (a power of two) entries -- keys zeroed, first all ones -- zero_idx is all ones, n_keys counts the entries.
// dup[i] = 1 for a live record whose hash an earlier live record (of this or an earlier call; base = file index of record 0) carried
hipError_t launch_dedup(const uint64_t *hashes, const uint8_t *alive, uint32_t n, uint64_t base, unsigned long long *keys,
                        unsigned long long *first, uint64_t slots, unsigned long long *zero_idx, unsigned long long *n_keys, uint8_t *dup,
                        hipStream_t st);
hipError_t launch_dedup_rehash(const unsigned long long *old_keys, const unsigned long long *old_first, uint64_t old_slots,
                               unsigned long long *keys, unsigned long long *first, uint64_t slots, hipStream_t st);
// protein-space baiting (mf_protein.hip): peptide k-mer table builder and the six-frame filter kernel
hipError_t launch_build_ptable(const uint8_t *aa, const uint8_t *runlen, uint64_t total, int kp, uint64_t *keys, uint64_t slots,
                               hipStream_t st);
hipError_t launch_build_pbits(const uint64_t *keys, uint64_t slots, uint32_t *kbloom, uint32_t kb_log2w, hipStream_t st);
hipError_t launch_pfilter(const ReadsView &R, const KmerSetView &S, uint32_t thr, bool count_all, uint32_t *out_bits,
                          uint32_t *hits_out, unsigned long long *counters, int n_cu, hipStream_t st, const KernelTiming *tm = nullptr);
// npos_blk: n_blk = (total_bases >> NPOS_BLK_SHIFT) + 3 entries
hipError_t launch_build_npos_blk(const uint64_t *npos, uint64_t n_npos, uint64_t n_blk, uint32_t *blk, hipStream_t st);
// off_blk: n_blk = (total_bases >> OFF_BLK_SHIFT) + 2 entries (ragged read sets)
hipError_t launch_build_off_blk(const uint64_t *offsets, uint64_t n_reads, uint64_t n_blk, uint64_t *blk, hipStream_t st);
hipError_t launch_mark_has_n(const ReadsView &R, uint32_t *has_n, hipStream_t st);

} // namespace mf
