// Chunked FASTQ -> pack -> GPU -> survivors pipeline behind mf_filter_fastq_files.
// Stages run on their own threads and overlap: inflate/read + parse (one reader per mate),
// 2-bit pack, H2D + kernels (one worker per device, batches of whole pairs dealt round robin,
// no collective), ordered write-out (one writer per output file).
#pragma once
#include "mf_host.h"
#include <functional>
#include <string>
#include <vector>

namespace mf {

// filter one packed mate batch on `device`; fills bits (ceil(n/32) u32).  Returns 0 or an MF_E_* code.
using BatchFilterFn = std::function<int(int device, const PackedHost &, uint64_t n, std::vector<uint32_t> &bits, std::string &err)>;

struct PipelineStats { uint64_t kept = 0, total = 0, batches = 0; };

int run_fastq_pipeline(const char *fq1, const char *fq2, const char *out1, const char *out2, bool pair_both,
                       int n_devices, int pack_threads, uint64_t batch_reads, const BatchFilterFn &filter,
                       PipelineStats &stats, std::string &err);

// ---------------------------------------------------------------- FASTQ quality filter (filter_v2)
struct QualParams {
    uint64_t start = 0, end = 0, ns = 10, trim = 0;
    uint32_t quality = 55; float limit = 0.2f;
    bool dedup = false, trunc = false;
};
bool utf8_valid(const char *p, size_t n);          // what Rust's `lines()` accepts (the reference unwraps header, sequence and quality line: main.rs:214-216, 287-289)
struct QualSpan { uint32_t s_off, s_len, q_off, q_len; };     // same layout as the kernels' QualRec
// counts of n records whose strings are offsets into text; with want_hashes the SipHash-1-3 values of the sequences are computed
// too and KEPT BY THE CALLEE for the dedup call that follows for the same records
using QualScanFn = std::function<int(const char *text, size_t len, const QualSpan *recs, uint32_t n, uint32_t quality,
                                     uint32_t *n_count, uint32_t *bad_count, bool want_hashes, std::string &err)>;
// de-duplication (main.rs:244-250) of the n records hashed by the last scan call, in file order over all calls: alive[i] = the record
// reached the dedup test; dup[i] = 1 when an earlier live record (of this or an earlier call) carried the same hash value
using QualDedupFn = std::function<int(const uint8_t *alive, uint32_t n, uint8_t *dup, std::string &err)>;
struct QualStats { uint64_t kept = 0, total = 0; bool panicked = false; };
// fq1 == nullptr: standard input; out2 == nullptr with fq2 set: standard output (reference: helper.rs:14-52)
int run_qualfilter_pipeline(const char *fq1, const char *fq2, const char *out1, const char *out2, const QualParams &p,
                            int threads, uint64_t batch_reads, const QualScanFn &scan, const QualDedupFn &dedup, QualStats &stats,
                            std::string &err);

} // namespace mf
