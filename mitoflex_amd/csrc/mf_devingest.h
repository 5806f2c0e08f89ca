// The device ingest path of mf_filter_fastq_files (mf_devingest.cpp): the file's bytes -- compressed, if it is a .gz -- go up to
// the GPU as they are, and inflate, line indexing, 2-bit packing, the filter and the copy of the survivors all run there
// (configs[4] of BASELINE.json; the reference's conventions: .gz by extension and 4-line records,
// filter/filter_bin/src/helper.rs:14-31, main.rs:287-321).  The host maps the files, feeds the copy engine and writes the
// survivors; it decodes deflate data only across the rare places the device decoder cannot link (a block start no chunk found,
// files of stored blocks only).
#pragma once
#include "../../include/mitofilter.h"
#include <stdint.h>
#include <string>

namespace mf {

constexpr int MF_DEVINGEST_DECLINED = 1;      // not an input this path takes (a pipe, BGZF, an empty file ...): use the host pipeline

// what a call of the device path did (mf_last_ingest_stats of the C ABI)
struct IngestStats {
    uint64_t input_bytes = 0, text_bytes = 0, records = 0;     // bytes of the input files as they lie; text they hold; FASTQ records cut from it (both mates)
    double seconds = 0, decode_busy_seconds = 0;               // wall time of the call; time during which at least one inflate kernel was running (0: no .gz input)
    uint64_t pool_bytes_peak = 0, device_bytes_peak = 0;       // this call's buffers / everything in use on the device, at most, on any one device
    uint64_t chunks = 0, chunks_linked = 0, gaps = 0, gap_bytes = 0;
    int n_devices = 0, consumers = 0;
};

// fq2 / out2 null: single end.  devices: the (logical) devices the slabs of the input are dealt to, round robin.
// Returns MF_OK, MF_DEVINGEST_DECLINED (no survivor has been written: the caller takes the host pipeline) or an MF_E_* code with err set.
int run_device_ingest(mf_kmerset *ks, const char *fq1, const char *fq2, const char *out1, const char *out2, uint32_t threshold,
                      bool pair_both, const int *devices, int n_devices, uint64_t *kept, uint64_t *total, std::string &err, IngestStats *stats = nullptr);

// The same path with the reference's FASTQ quality filter (filter_v2: filter/filter_bin/src/main.rs:188-323) as its job instead of the
// bait filter: counting, hashing, the de-duplication set, the decisions and the formatting of the kept records run on `device` over
// the text where the decoder left it; the host checks the rare line that is not ASCII, spends the -t budget and writes.  Declines
// (MF_DEVINGEST_DECLINED, nothing touched) standard input, pipes, BGZF, empty files and .gz outputs.
struct QualParams;
int run_device_qualfilter(const char *fq1, const char *fq2, const char *out1, const char *out2, const QualParams &P, int device, uint64_t *kept,
                          uint64_t *total, bool *panicked, std::string &err, IngestStats *stats = nullptr);

// What a file-level call's set-up needs of (logical) device `device` is started in the background: the maker thread of the device's
// streams, two pinned staging buffers.  Returns at once; calling it again is free.  (mf_set_option("expect_files", "1") makes the library
// call it when a device's first context is made.)
void ingest_prefetch(int device);
// The process makes one file-level call and ends (the CLIs): what costs the process's exit more than it saves the call is left out -- the
// CU-masked stream set below 8 GB of compressed input.  (mf_set_option("short_lived", "1"))
void ingest_short_lived(bool yes);

} // namespace mf
