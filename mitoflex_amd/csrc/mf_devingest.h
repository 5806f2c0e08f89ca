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

// fq2 / out2 null: single end.  devices: the (logical) devices the slabs of the input are dealt to, round robin.
// Returns MF_OK, MF_DEVINGEST_DECLINED (no survivor has been written: the caller takes the host pipeline) or an MF_E_* code with err set.
int run_device_ingest(mf_kmerset *ks, const char *fq1, const char *fq2, const char *out1, const char *out2, uint32_t threshold,
                      bool pair_both, const int *devices, int n_devices, uint64_t *kept, uint64_t *total, std::string &err);

} // namespace mf
