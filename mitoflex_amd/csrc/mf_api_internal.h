// Internals of mf_api.cpp that the device ingest path (mf_devingest.cpp) shares: device contexts, the read-set handle and
// the filter call.  Not part of the C ABI.
#pragma once
#include "../../include/mitofilter.h"
#include "mf_common.h"
#include "mf_kernels.h"
#include <hip/hip_runtime.h>
#include <string>

// stream2: finish kernels of pipelined passes (stream4: those of every other pass when the finish kernels are what a pass waits for); stream3: every other screen
struct DevCtx { int device = -1; hipStream_t stream = nullptr, stream2 = nullptr, stream3 = nullptr, stream4 = nullptr; int n_cu = 0; };

int fail(int code, const char *fmt, ...);               // sets the thread's error message, returns code
const std::string &mf_thread_error();
int phys(int device);                                   // logical -> physical device (MF_FAKE_DEVICES)
int get_ctx(int device, DevCtx **out, int lane = 0);

// The device ingest path keeps device buffers of earlier calls in a pool per device (mf_devingest.cpp).  Memory that sits there idle must
// never make another allocation of this library fail: release_cached_device_memory() gives the current device's idle pool buffers (and, with
// all = true, every cache of every device: pool, consumers' scratch and read sets, pinned staging) back to the runtime; dev_malloc() is
// hipMalloc that calls it and tries once more when the device is full.  Returns the bytes released.
namespace mf { size_t release_cached_device_memory(bool all); }
inline hipError_t dev_malloc(void **p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        if (mf::release_cached_device_memory(false)) e = hipMalloc(p, bytes);
        if (e != hipSuccess) (void)hipGetLastError();
    }
    return e;
}
template <class T> inline hipError_t dev_malloc(T **p, size_t bytes) { return dev_malloc(reinterpret_cast<void **>(p), bytes); }

// grow a device buffer to at least `bytes` (with some slack when it is being re-used)
template <class T> inline hipError_t dev_reserve(T *&p, size_t &cap, size_t bytes, bool slack)
{
    if (bytes <= cap && p) return hipSuccess;
    if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
    const size_t want = slack ? bytes + bytes / 4 + 4096 : (bytes ? bytes : 16);
    hipError_t e = dev_malloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
}


// Buffer sets a pipelined pass rotates through.  (Round 2 measured nothing from a third; since the finish kernels of two-word keys run on two streams and
// outlast a screen, it is worth 4 % of a pass for them: profiles/r06/n_three_sets.txt -- one-word keys keep to two, mf_api.cpp enqueue_pass.  A fourth adds nothing.)
#ifndef MF_NSETS
#define MF_NSETS 3
#endif
constexpr int NSETS = MF_NSETS;
struct mf_reads {
    int device = 0, lane = 0;     // lane: which of the device's contexts (streams) this read set works on
    mf::ReadsView v{};
    uint32_t *d_words = nullptr; uint64_t *d_offsets = nullptr, *d_npos = nullptr;
    uint32_t *d_has_n = nullptr, *d_hits = nullptr, *d_npos_blk = nullptr; uint64_t *d_off_blk = nullptr;
    // Threshold-1 passes (screen_kernel + finish_kernel) are pipelined: the finish kernel of pass i runs on a second stream
    // under the screen kernel of pass i + 1, the way consecutive batches of a file do.  What a pass writes therefore
    // exists NSETS times and rotates: record lists, result bitmap, tally buffer.  `cur` holds the latest result.
    uint32_t *d_cand[NSETS] = {}, *d_bits[NSETS] = {};
    void *d_recs[NSETS] = {}; uint32_t *d_rec_counts[NSETS] = {};     // stage-1 positive records (screen -> finish / mark)
    unsigned long long *d_counters[NSETS] = {};                       // 2 * EXACT_MAX_GRID tally pairs each, in pinned HOST memory: the kernels store
                                                                      // their pair there directly and a call ends without a device-to-host copy
    hipEvent_t ev_screen[NSETS] = {}, ev_finish[NSETS] = {};          // ordering between the two streams
    hipEvent_t ev_call[2] = {};                                       // begin / end of a call's passes
    bool cand_clean[NSETS] = {}, sample_pass = false;     // sample_pass: the latest pass was a screen + finish one
    // Bait-rich input (more than a few per cent of the reads are bait reads -- what the `bim` loop enriches towards) is better
    // served by the candidate-bitmap pass: one thread per stage-1 record means several records per bait read, and the screen
    // writes them all.  The choice follows the work the last call of this read set (the last batch of this device) saw.
    bool prefer_split = false;
    bool finish_two = false;          // bait-rich input (from the last call's tallies): the finish kernels of consecutive passes go to two streams
    bool split_serial = false;      // ... and with very many candidates (> 5 % of the reads) its kernels do not fit beside the next screen: one stream
    int cur = 0;
    int flip = 0;               // parity of the pipelined passes enqueued so far (which of two streams a pass's screen / finish kernels take)
    unsigned long long *tally_override = nullptr;       // set per pass by filter_common when every pass's tally is wanted
    size_t bitmap_bytes = 0;
    // capacities (bytes), so that a handle can be refilled batch after batch without touching the allocator
    size_t cap_words = 0, cap_offsets = 0, cap_npos = 0, cap_bitmap = 0, cap_recs = 0, cap_rec_counts = 0, cap_hits = 0, cap_npos_blk = 0, cap_off_blk = 0;
};


void reads_release(mf_reads *r);
// Device buffers of a read set that the caller fills on the device: packed words (padded, the tail behind n_words zeroed),
// offsets (n_reads + 1, unless uniform_len), room for npos_cap invalid positions.
int reads_reserve(mf_reads *r, bool reuse, uint64_t n_words, uint64_t n_reads, uint32_t uniform_len, uint64_t npos_cap, DevCtx *ctx);
// ... and what follows once words / offsets / npos are in place (index over the invalid positions, bitmaps, record lists).
// Ends synchronised.
int reads_finish(mf_reads *r, bool reuse, uint64_t n_words, uint64_t n_reads, uint64_t total_bases, uint32_t uniform_len, uint64_t n_npos, DevCtx *ctx);
// one or more passes of the filter over a resident read set; out_bits / hits_out may be null (the result stays in r->d_bits[r->cur])
int filter_common(const mf_kmerset *ks, const mf_reads *reads, uint32_t thr, int mode, uint32_t *out_bits, uint32_t *hits_out, int steps,
                  mf_filter_stats_t *stats, uint64_t *pass_per_step = nullptr);
