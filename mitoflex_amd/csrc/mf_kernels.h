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
uint64_t screen_grid_for(const ReadsView &R, int n_cu);
uint64_t screen_rec_cap_for(const ReadsView &R, int n_cu);          // 16-byte records per screen workgroup
hipError_t launch_screen(const ReadsView &R, const KmerSetView &S, void *recs, uint32_t *rec_counts, int n_cu, hipStream_t st,
                         const KernelTiming *tm = nullptr);
hipError_t launch_mark(const ReadsView &R, const KmerSetView &S, const void *recs, const uint32_t *rec_counts, uint32_t *cand, int n_cu,
                       hipStream_t st, const KernelTiming *tm = nullptr);
// fused pass (screen + mark + exact in one launch): geometry of a read set and the buffers one pass works on
struct FusedGeom {
    uint32_t n_stream;          // streaming waves per workgroup (MF_STREAM_WAVES, default 14)
    uint64_t chunk_vec;         // uint4 per chunk (n_stream x 64 x SCREEN_U)
    uint64_t n_chunks, grid;
    uint64_t ovf_cap;           // overflow records per streaming wave
    uint64_t words_needed;      // device words the kernel may read (zero past the data)
    bool ok;                    // false: the set is too large for the record format (fall back to the split path)
};
struct FusedBuffers {
    uint32_t *cand, *bits, *cand_other, *bits_other;
    uint64_t bitmap_vec4;
    unsigned long long *ovf;
    unsigned long long *dfr; uint32_t dfr_cap;     // parked sixteen-window items: [grid][16][dfr_cap]
    uint32_t *hits_out;
    unsigned long long *partials;
    uint32_t flags;                 // debugging: bit 0 drops the stage-1 records (stream-only timing)
    unsigned long long *dbg;        // debugging: per-wave timestamps and counts, or nullptr
};
FusedGeom fused_geom_for(uint64_t n_words, int n_cu);
hipError_t launch_fused(const ReadsView &R, const KmerSetView &S, const FusedGeom &G, const FusedBuffers &B, uint32_t thr, bool count_all,
                        hipStream_t st, const KernelTiming *tm = nullptr);
hipError_t launch_exact(const ReadsView &R, const KmerSetView &S, uint32_t *cand, uint32_t thr, bool count_all,
                        uint32_t *out_bits, uint32_t *hits_out, unsigned long long *counters, int n_cu, hipStream_t st,
                        const KernelTiming *tm = nullptr);
hipError_t launch_build_kbloom(const uint64_t *keys, uint64_t slots, int kw, uint32_t *kbloom, uint32_t kb_log2w, hipStream_t st);
hipError_t launch_build_table(const BaitView &B, int k, int kw, uint64_t *keys, uint64_t slots, uint32_t *postab_scratch,
                              hipStream_t st);
hipError_t launch_build_screen(const BaitView &B, int s, uint32_t *bloom, uint32_t log2w, uint32_t log2w2, uint32_t *stab,
                               uint32_t stab_slots, uint32_t *has_ones, hipStream_t st);
hipError_t launch_count_keys(const uint64_t *keys, uint64_t slots, int kw, const uint32_t *stab, uint64_t stab_slots,
                             unsigned long long *out2, hipStream_t st);
// FASTQ quality filter: a record's (cut) sequence and quality strings as offsets into the uploaded text
struct QualRec { uint32_t s_off, s_len, q_off, q_len; };
hipError_t launch_qualscan(const uint8_t *text, const QualRec *recs, uint32_t n, uint32_t quality, uint32_t *n_count, uint32_t *bad_count,
                           uint64_t *hashes /* nullptr: no dedup hashes */, hipStream_t st);
// protein-space baiting (mf_protein.hip): peptide k-mer table builder and the six-frame filter kernel
hipError_t launch_build_ptable(const uint8_t *aa, const uint8_t *runlen, uint64_t total, int kp, uint64_t *keys, uint64_t slots,
                               hipStream_t st);
hipError_t launch_build_pbits(const uint64_t *keys, uint64_t slots, uint32_t *kbloom, uint32_t kb_log2w, hipStream_t st);
hipError_t launch_pfilter(const ReadsView &R, const KmerSetView &S, uint32_t thr, bool count_all, uint32_t *out_bits,
                          uint32_t *hits_out, unsigned long long *counters, int n_cu, hipStream_t st, const KernelTiming *tm = nullptr);
// npos_blk: n_blk = (total_bases >> NPOS_BLK_SHIFT) + 3 entries
hipError_t launch_build_npos_blk(const uint64_t *npos, uint64_t n_npos, uint64_t n_blk, uint32_t *blk, hipStream_t st);
hipError_t launch_mark_has_n(const ReadsView &R, uint32_t *has_n, hipStream_t st);

} // namespace mf
