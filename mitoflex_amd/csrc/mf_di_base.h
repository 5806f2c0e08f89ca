// Device ingest path, part 0: what every part includes -- headers, the knob table (every environment variable of the path), small helpers.
#pragma once
#include "mf_devingest.h"
#include "mf_api_internal.h"
#include "mf_gzdev.h"
#include "mf_host.h"
#include "mf_ingest.h"
#include "mf_pinflate.h"
#include "mf_pipeline.h"
#include "mf_qualsink.h"
#include "mf_coldtrace.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <fcntl.h>
#include <map>
#include <memory>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <errno.h>
#include <unistd.h>
#include <vector>

namespace mf {
namespace {

// ---- every environment variable of this path, in ONE table.  The values are read when the library is loaded and again at the start of every
// file-level call (run_ingest: the tests change them between calls of one process); the code asks the table, never the environment.
// All optional; what stands in for an unset one is written where it is used.  (DESIGN.md section 11 describes them for a user.)
#ifdef MF_TEST_HOOKS          // (libmitofilter_hip_hooks.so, what the failed-allocation test loads; the shipped library has no such knob)
#define MF_DI_TEST_KNOBS(X) X(DEVPOOL_FAIL_AT)          /* the n-th new pool allocation of the process fails as if the device were full */
#else
#define MF_DI_TEST_KNOBS(X)
#endif
#define MF_DI_KNOBS(X) \
    X(DEVINGEST_TRACE)              /* the ingest timeline on stderr */ \
    X(PIPE_TIMING)                  /* stage busy times of a call on stderr */ \
    X(KEEP_BUFFERS)                 /* "0": nothing is kept between calls (device pool, pinned staging, consumers' scratch) */ \
    X(DEVPOOL_GB)                   /* device memory a process keeps between calls, per device (8) */ \
    X(INGEST_BUDGET_GB)             /* device memory a call may hold in all (8 x its compressed bytes, 3..24) */ \
    X(INGEST_TEXT_BUFS)             /* text buffers a mate may hold */ \
    X(INGEST_CONSUMERS)             /* consumer threads */ \
    X(INGEST_CARRY_ROOM)            /* bytes of room for a record that straddles two pieces */ \
    X(INGEST_SLAB_BYTES)            /* text bytes per slab of a plain file */ \
    X(UPLOAD_THREADS)               /* threads that read a file into pinned staging (8) */ \
    X(UPLOAD_STAGED)                /* never register a file's mapping: always stage */ \
    X(UPLOAD_REGISTER_MAX_MB)       /* largest file whose mapping is registered (512) */ \
    X(GZDEV_CHUNK_BYTES)            /* compressed bytes per speculative chunk */ \
    X(GZDEV_SLAB_CHUNKS)            /* chunks per slab */ \
    X(GZDEV_SLABS_IN_FLIGHT)        /* slabs whose decode kernels may be in flight per device */ \
    X(GZDEV_EXPAND)                 /* symbols of room per compressed byte (fixed, instead of what the file has shown) */ \
    X(GZDEV_RING_BYTES)             /* the ring of compressed bytes */ \
    X(GZDEV_MARGIN)                 /* read-ahead margin of the ring */ \
    X(GZDEV_TEXT_PIECE)             /* most text bytes of a piece */ \
    X(GZDEV_RETRY_BYTES)            /* symbol room a slab may be decoded again with */ \
    X(GZDEV_UPLOAD_BUFS)            /* the uploader's staging buffers (2) */ \
    X(GZDEV_UPLOAD_PIECE_MB)        /* ... and their size (32) */ \
    X(GZDEV_LARGE_MB)               /* compressed megabytes from which a call takes the CU-masked stream set */ \
    X(GZDEV_NO_CUMASK)              /* never mask */ \
    X(GZDEV_RESERVED_CUS)           /* CUs the masked decode streams leave free (32) */ \
    X(GZDEV_DEC_STREAMS)            /* decode streams a call may use */ \
    X(GZDEV_RESOLVE_STREAM)         /* "1" / "0": bodies and CRC on a second post stream always / never */ \
    X(GZDEV_DESTROY_STREAMS_AT_EXIT)/* destroy the path's streams at exit (as under a profiler) */ \
    X(QUAL_DEVICE_GZ_OUT)           /* quality filter: .gz outputs through this path too */ \
    X(QUAL_OUT_CHUNK)               /* quality filter: bytes per pinned output chunk */ \
    X(QUAL_OUT_CHUNKS)              /* ... and how many */ \
    X(DEDUP_LOG2_SLOTS)             /* quality filter: initial slots of the de-duplication set */ \
    MF_DI_TEST_KNOBS(X)
enum KnobId {
#define X(n) KN_##n,
    MF_DI_KNOBS(X)
#undef X
    KN_COUNT
};
class Knobs {
public:
    Knobs() { refresh(); }
    void refresh()
    {
        static const char *const names[KN_COUNT] = {
#define X(n) "MF_" #n,
            MF_DI_KNOBS(X)
#undef X
        };
        for (int i = 0; i < KN_COUNT; i++) {
            const char *s = getenv(names[i]);
            v_[i].u.store(s && *s ? strtoull(s, nullptr, 10) : 0, std::memory_order_relaxed);
            v_[i].flags.store((uint8_t)((s ? 1 : 0) | (s && *s ? 2 : 0) | (s && s[0] == '0' ? 4 : 0) | (s && s[0] == '1' ? 8 : 0)), std::memory_order_relaxed);
        }
    }
    uint64_t u64(KnobId id, uint64_t dflt) const { return (v_[id].flags.load(std::memory_order_relaxed) & 2) ? v_[id].u.load(std::memory_order_relaxed) : dflt; }      // set and not empty
    bool is_set(KnobId id) const { return v_[id].flags.load(std::memory_order_relaxed) & 1; }
    bool starts_0(KnobId id) const { return v_[id].flags.load(std::memory_order_relaxed) & 4; }
    bool starts_1(KnobId id) const { return v_[id].flags.load(std::memory_order_relaxed) & 8; }
private:
    struct V { std::atomic<uint64_t> u{0}; std::atomic<uint8_t> flags{0}; };          // (atomics: a call's threads read while another call of the process refreshes)
    V v_[KN_COUNT];
};
Knobs g_knobs;

#define DCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err = std::string(#call) + " failed: " + hipGetErrorString(e_); return e_ == hipErrorOutOfMemory ? MF_E_NOMEM : MF_E_HIP; } } while (0)

// Pinned host memory that KERNELS read or write (launch_bytes_from_host / _to_host, the survivors' list): coherent (fine-grained), so that
// nothing of it sits in the device's L2 from one kernel to the next while the host rewrites it.  hipHostMallocDefault is coherent by itself;
// hipHostMallocPortable alone is not (it follows HIP_HOST_COHERENT, 0 by default).  Staging buffers only the copy engine reads stay as they were.
constexpr unsigned PINNED_FOR_KERNELS = hipHostMallocPortable | hipHostMallocCoherent;
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define TRACE(...) do { if (g_knobs.is_set(KN_DEVINGEST_TRACE)) { const double t_ = now_s(); fprintf(stderr, "[devingest %.3f] ", t_ - (double)((long)t_ / 1000 * 1000)); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)
// a bounded wait on a condition variable (polling loops).  Against the system clock on purpose: that is pthread_cond_timedwait, which
// ThreadSanitizer knows; wait_for() is pthread_cond_clockwait, which the libtsan of this toolchain does not intercept (it then believes the
// mutex still held and reports a double lock at the next wait).  A clock step only stretches or cuts one nap of a few hundred microseconds.
void nap(std::condition_variable &cv, std::unique_lock<std::mutex> &lk, unsigned us) { cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(us)); }
size_t pow2_ceil(size_t v) { size_t p = 1; while (p < v) p <<= 1; return p; }

} // namespace
} // namespace mf
