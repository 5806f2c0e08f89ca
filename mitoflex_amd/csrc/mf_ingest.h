// FASTQ text that is already in device memory -> line index -> 2-bit packed read set -> survivors' text (mf_ingest.hip).
// The device-side counterpart of the host readers and packer (mf_pipeline.cpp, mf_host.cpp pack_records): same conventions --
// the reference's (filter/filter_bin/src/main.rs:287-321): strict 4-line records, line 3 ignored, a CR in front of the LF
// stripped, a partial record at the very end dropped; Spec B's alphabet: A/a C/c G/g T/t, anything else invalid
// (stored as 0 and listed).  Used by the device ingest path of mf_filter_fastq_files (mf_devingest.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mf {

void ingest_preload();          // loads this file's code object now instead of at its first launch

constexpr uint32_t INGEST_TILE = 4096;          // bytes of text per workgroup of the line kernels

// exclusive prefix sums, u32 -> u64: out[0 .. n] (n + 1 values, out[n] = total).  scratch: (n / 4096 + 2) u64
hipError_t launch_scan_u32(const uint32_t *in, uint64_t n, uint64_t *out, uint64_t *scratch, hipStream_t st);

// newlines per tile of INGEST_TILE bytes of text[0 .. n); text must be readable up to n + 16
hipError_t launch_count_newlines(const uint8_t *text, uint64_t n, uint32_t *tile_cnt, hipStream_t st);
// line_start[k] = offset of the first byte of line k (line 0 starts at 0; line k + 1 starts behind the k-th newline);
// tile_base = exclusive scan of tile_cnt
hipError_t launch_line_starts(const uint8_t *text, uint64_t n, const uint64_t *tile_base, uint64_t *line_start, hipStream_t st);
// sequence length of records [0, n_rec): line 4r+1 without its LF and without a CR in front of it; minmax[0] = min, [1] = max
// (preset to ~0, 0).  line_start must have 4 * n_rec + 1 entries.
hipError_t launch_seq_lens(const uint8_t *text, const uint64_t *line_start, uint64_t n_rec, uint32_t *seq_len, uint32_t *minmax, hipStream_t st);
// 2-bit pack of a batch of records, APPENDED to a read set: the batch's bases take the places base .. base + total_bases - 1 of
// the stream (words[g >> 4] holds base g; the word the batch shares with its predecessor is completed, the words behind must
// not hold anything yet).  offsets: the batch's own n_rec + 1 base offsets, starting at 0 (nullptr when every read of the batch
// has uniform_len bases).  MODE 0 (npos == nullptr): writes words and, per workgroup of 256 words, the number of invalid bases
// to inv_cnt (pack_blocks() entries).  MODE 1: inv_base = exclusive scan of inv_cnt; writes the stream positions of the invalid
// bases, ascending, to npos[0 ..].
uint64_t pack_blocks(uint64_t total_bases, uint64_t base);
hipError_t launch_pack(const uint8_t *text, const uint64_t *line_start, const uint64_t *offsets, uint32_t uniform_len, uint64_t n_rec,
                       uint64_t total_bases, uint64_t base, uint32_t *words, uint32_t *inv_cnt, const uint64_t *inv_base, uint64_t *npos, hipStream_t st);
hipError_t launch_bytes_from_host(void *dst, const void *pinned_src, uint64_t n, hipStream_t st);      // dst[0..n) = pinned host memory, read by a kernel (not the copy engine: that one carries the uploads)
hipError_t launch_bytes_to_host(void *pinned_dst, const void *src, uint64_t n, hipStream_t st);        // pinned host memory [0..n) = src, written by a kernel; read it after waiting for the stream
hipError_t launch_add_base(uint64_t *dst, const uint64_t *src, uint64_t n, uint64_t base, hipStream_t st);       // dst[i] = src[i] + base
// pass bits of one batch (bit i = record i of the batch) into the file-wide bitmap at record index rec_base
hipError_t launch_store_bits(const uint32_t *batch_bits, uint64_t n_rec, uint32_t *file_bits, uint64_t rec_base, hipStream_t st);
// output bytes of records [0, n_rec) of a batch whose first record has file index rec_base: header + seq + "+" + qual with LF
// line ends when kept (bits_a[r] | bits_b[r], or & when both; bits_b may be nullptr), else 0
hipError_t launch_out_lens(const uint8_t *text, const uint64_t *line_start, uint64_t n_rec, uint64_t rec_base, const uint32_t *bits_a,
                           const uint32_t *bits_b, int both, uint32_t *out_len, hipStream_t st);
// the same two steps over a list of records: out_len[i] = output bytes of record sel[i]; the records go to out[out_off[i] ..]
hipError_t launch_sel_lens(const uint8_t *text, const uint64_t *line_start, const uint32_t *sel, uint64_t n_sel, uint32_t *out_len, hipStream_t st);
hipError_t launch_sel_gather(const uint8_t *text, const uint64_t *line_start, const uint32_t *sel, uint64_t n_sel, const uint64_t *out_off, uint8_t *out, hipStream_t st);
// copies the kept records to out[out_off[r] ..] (out_off = exclusive scan of out_len)
hipError_t launch_gather(const uint8_t *text, const uint64_t *line_start, uint64_t n_rec, const uint32_t *out_len, const uint64_t *out_off,
                         uint8_t *out, hipStream_t st);

// ---- the FASTQ quality filter (the reference's filter_v2, filter/filter_bin/src/main.rs:188-323) over records of text on the device
// per-record flags of launch_qual_scan
constexpr uint32_t QF_HIGH = 1;       // a byte >= 0x80 in the header, sequence or quality line: the host checks those lines for UTF-8 (the reference unwraps them)
constexpr uint32_t QF_SHORT = 2;      // the sequence or the quality string is shorter than the cut's start: the reference panics
constexpr uint32_t QF_NFAIL = 4;      // more than `ns` 'N' in the cut sequence
constexpr uint32_t QF_LONG = 8;       // a record of 4 GiB or more: not handled here
// One pass over the records' bytes.  start / cap: the cut [start, start + cap) of sequence and quality string (cap = ~0: to the end).
// bad[r] = bytes <= quality of the cut quality string; cut_sl / cut_ql = lengths of the cut strings; olen[r] = bytes of the record as
// it is written (header LF sequence LF '+' LF quality LF); *first_flag (preset to ~0) = the first record with QF_HIGH, QF_SHORT or QF_LONG.
// text must be readable 16 bytes past the last record.
hipError_t launch_qual_scan(const uint8_t *text, const uint64_t *line_start, uint64_t n_rec, uint64_t start, uint64_t cap, uint32_t quality, uint64_t ns,
                            uint32_t *bad, uint8_t *flags, uint32_t *cut_sl, uint32_t *cut_ql, uint32_t *olen, uint32_t *first_flag, hipStream_t st);
// SipHash-1-3 of the cut sequences followed by 0xff (Rust's `str::hash` into DefaultHasher); reads whole dwords around the sequence
hipError_t launch_qual_hash(const uint8_t *text, const uint64_t *line_start, uint64_t n_rec, uint64_t start, const uint32_t *cut_sl, uint64_t *hashes, hipStream_t st);
// alive[i] = 0 when record i fails the N or the quality test (of either mate with pe; *2 = the other mate's scan results); all 1 with trunc
hipError_t launch_qual_decide(uint64_t n, bool pe, bool trunc, float limit, const uint32_t *bad1, const uint8_t *fl1, const uint32_t *sl1, const uint32_t *ql1,
                              const uint32_t *bad2, const uint8_t *fl2, uint8_t *alive, hipStream_t st);
// keep[i] = alive[i] && !dup[i] (dup, keep may be null); out_len[i] = keep ? olen[i] : 0; *kept += number kept (may be null)
hipError_t launch_qual_keep(uint64_t n, const uint8_t *alive, const uint8_t *dup, const uint32_t *olen, uint8_t *keep, uint32_t *out_len, unsigned long long *kept,
                            hipStream_t st);
// the kept records (out_len != 0) as the reference writes them, at out[out_off[r] ..]
hipError_t launch_qual_gather(const uint8_t *text, const uint64_t *line_start, uint64_t n_rec, uint64_t start, const uint32_t *cut_sl, const uint32_t *cut_ql,
                              const uint32_t *out_len, const uint64_t *out_off, uint8_t *out, hipStream_t st);

} // namespace mf
