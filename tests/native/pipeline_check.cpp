// The file pipeline (readers, pools, pack, ordered writers) with a stand-in filter on the CPU, so that it can run
// under sanitizers and without a GPU: a read passes iff its first base is A/a.
//   pipeline_check FQ1 FQ2|- OUT1 OUT2|- BATCH_READS PACK_THREADS [both|either [N_DEVICES]]   -> prints "kept total"
// With N_DEVICES > 1 the stand-in filter sleeps a pseudo-random time that depends on the device and the batch, so that
// batches come back out of order and the dealing / reordering logic of run_fastq_pipeline is exercised.
#include "mf_pipeline.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <thread>
int main(int argc, char **argv)
{
    if (argc < 7) return 2;
    const char *fq2 = strcmp(argv[2], "-") ? argv[2] : nullptr, *out2 = strcmp(argv[4], "-") ? argv[4] : nullptr;
    const int n_devices = argc > 8 ? atoi(argv[8]) : 1;
    std::atomic<unsigned> calls{0};
    mf::BatchFilterFn fn = [n_devices, &calls](int device, const mf::PackedHost &P, uint64_t n, std::vector<uint32_t> &bits, std::string &) -> int {
        if (n_devices > 1) {
            const unsigned c = calls.fetch_add(1);
            std::this_thread::sleep_for(std::chrono::microseconds(((c * 2654435761u) >> 20) % 3000 + (device % 3) * 500));
        }
        bits.assign((n + 31) / 32 + 1, 0);
        size_t ni = 0;
        for (uint64_t i = 0; i < n; i++) {
            const uint64_t g = P.offsets[i];
            while (ni < P.npos.size() && P.npos[ni] < g) ni++;
            const bool invalid = ni < P.npos.size() && P.npos[ni] == g;
            if (P.offsets[i + 1] > g && !invalid && ((P.words[g >> 4] >> (2 * (g & 15))) & 3u) == 0) bits[i >> 5] |= 1u << (i & 31);
        }
        return 0;
    };
    mf::PipelineStats st; std::string err;
    const int rc = mf::run_fastq_pipeline(argv[1], fq2, argv[3], out2, argc > 7 && !strcmp(argv[7], "both"), n_devices, atoi(argv[6]),
                                          strtoull(argv[5], nullptr, 10), fn, st, err);
    if (rc) { printf("error %d: %s\n", rc, err.c_str()); return 0; }
    printf("%llu %llu\n", (unsigned long long)st.kept, (unsigned long long)st.total);
    return 0;
}
