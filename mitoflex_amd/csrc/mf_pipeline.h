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

} // namespace mf
