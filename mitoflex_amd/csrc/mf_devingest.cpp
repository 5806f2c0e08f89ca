// Device ingest path of mf_filter_fastq_files (see mf_devingest.h).  Streaming: what a file holds in device memory at any time is
// bounded, whatever its size.
//
// Per mate file one PRODUCER thread turns the file into pieces of text in device buffers, in order:
//   * a .gz is decoded in slabs of a few hundred speculative chunks (mf_gzdev.h).  Its compressed bytes pass through a RING per
//     device (a power of two of bytes; byte b of the file lives at ring[b % R]) that an uploader thread fills a piece at a time and
//     that is recycled as slabs are linked; the decode kernels of the slabs ahead run on the device's decode streams, each followed
//     by the copy of its chunks' descriptors to the host; which chunks are accepted is decided on the host from those (gz_link_walk);
//     the windows, marker resolution and CRC are kernels on the mate's post stream, and the piece is handed over WITH AN EVENT --
//     the producer waits for a slab's descriptors and for a text buffer, never for a kernel of its own;
//     symbol room per chunk follows the expansion the file has shown so far, slabs in flight the call's memory budget;
//     every slab's text goes to a buffer of its own (the 32 KiB window and the carry in front of it);
//   * a plain FASTQ file IS the text: it is read straight into such buffers (three staging buffers, the device's copy stream).
//   With n devices the slabs are dealt to them round robin: the link step of slab k needs the state the link step of slab k - 1
//   left (a few scalars on the host, the last 32 KiB of text through pinned memory), nothing else crosses devices.
// The streams all of this runs on are made once per process and device by a maker thread, in the order a cold call needs them
// (StreamSets below): the reference calls this path a process at a time.
// A few CONSUMER threads take the pieces.  Per piece, on the device that holds it: cut the text into records where it lies
// (mf_ingest.h; one piece of a mate at a time, in order: what is behind the last complete record of a piece, the carry, goes to the
// front of the next piece's buffer), then -- several pieces side by side, each consumer on its own streams -- the piece's job:
//   * the bait filter (mf_filter_fastq_files): 2-bit pack into the consumer's refillable read set, ONE filter pass over the piece's
//     reads; the pass bits come back to the host (a bit per read), where the pair rule is applied across the two mates' file-wide
//     bitmaps -- the pieces of the two mates do not cover the same records, the mate that is behind in records is advanced --; a piece
//     whose records the other mate has covered gets its list of kept records, its survivors are gathered on the device and copied out
//     to a writer thread per output file;
//   * the quality filter (mf_qualfilter_files, the reference's filter_v2): one pass over the records' bytes (counts, flags, cut
//     lengths, SipHash), decisions a piece of mate 1 at a time in file order (the other mate's counts through per-record arrays on the
//     host, the de-duplication set on the device, the -t budget on the host), the kept records formatted on the device and sent down
//     through pinned chunks to a writer thread per output file -- nearly every record is kept, so this job writes as much as it reads.
// A piece's buffers go back to the pool when its part of the output is on its way.  A producer blocks when its mate holds
// MF_INGEST_TEXT_BUFS text buffers.
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

#define DCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err = std::string(#call) + " failed: " + hipGetErrorString(e_); return e_ == hipErrorOutOfMemory ? MF_E_NOMEM : MF_E_HIP; } } while (0)

// Pinned host memory that KERNELS read or write (launch_bytes_from_host / _to_host, the survivors' list): coherent (fine-grained), so that
// nothing of it sits in the device's L2 from one kernel to the next while the host rewrites it.  hipHostMallocDefault is coherent by itself;
// hipHostMallocPortable alone is not (it follows HIP_HOST_COHERENT, 0 by default).  Staging buffers only the copy engine reads stay as they were.
constexpr unsigned PINNED_FOR_KERNELS = hipHostMallocPortable | hipHostMallocCoherent;
uint64_t env_u64(const char *name, uint64_t dflt) { const char *v = getenv(name); return v && *v ? strtoull(v, nullptr, 10) : dflt; }
static const bool g_trace = getenv("MF_DEVINGEST_TRACE") != nullptr;
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define TRACE(...) do { if (g_trace) { const double t_ = now_s(); fprintf(stderr, "[devingest %.3f] ", t_ - (double)((long)t_ / 1000 * 1000)); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)
// a bounded wait on a condition variable (polling loops).  Against the system clock on purpose: that is pthread_cond_timedwait, which
// ThreadSanitizer knows; wait_for() is pthread_cond_clockwait, which the libtsan of this toolchain does not intercept (it then believes the
// mutex still held and reports a double lock at the next wait).  A clock step only stretches or cuts one nap of a few hundred microseconds.
void nap(std::condition_variable &cv, std::unique_lock<std::mutex> &lk, unsigned us) { cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(us)); }
size_t pow2_ceil(size_t v) { size_t p = 1; while (p < v) p <<= 1; return p; }

// Device memory of this path comes from a pool per device that outlives the call.  Two reasons.  hipFree waits for the whole
// device to go idle -- with decode kernels in flight on other streams that is tens of milliseconds a call -- so nothing is freed
// while a file is being processed: buffers go back to the pool and are handed out again (a slab's symbol and text buffers have
// the size of the slab before).  And allocating (and later releasing) gigabytes costs a large fraction of a second, which a caller
// that filters file after file (the bim loop) would pay every time: a call's buffers are kept for the next one, up to
// MF_DEVPOOL_GB (see run_ingest for the default; MF_KEEP_BUFFERS=0: nothing is kept).  Memory that idles here is given back whenever another
// allocation of the library finds the device full (release_cached_device_memory, mf_api_internal.h) and by mf_release_cached() of the C ABI.
// get() wants the caller's current device to be `dev`.
class DevPool {
public:
    static size_t round_up(size_t bytes)
    {
        size_t unit = (size_t)1 << 20;
        while (unit * 16 < bytes && unit < ((size_t)256 << 20)) unit <<= 1;         // 1 MiB steps for small blocks, up to 256 MiB steps
        return (bytes + unit - 1) / unit * unit;
    }
    hipError_t get(int dev, void **p, size_t bytes, size_t *got)
    {
        const size_t want = round_up(bytes ? bytes : 1);
        {
            std::lock_guard<std::mutex> lk(mu_);
            PerDev &D = dev_[dev];
            auto it = D.free_.lower_bound(want);
            if (it != D.free_.end() && it->first <= want + want / 2 + ((size_t)64 << 20)) {
                *p = it->second; *got = it->first; D.held -= it->first; D.free_.erase(it);
                account(dev, (long long)*got);
                return hipSuccess;
            }
        }
        const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        // (test hook: MF_DEVPOOL_FAIL_AT=n makes the n-th new allocation of the process fail as if the device were full -- the call must then
        // hand the input to the host pipeline, tests/test_gpu_devingest.py::test_a_failed_allocation_hands_the_call_to_the_host_pipeline)
        static const uint64_t fail_at = env_u64("MF_DEVPOOL_FAIL_AT", 0);
        static std::atomic<uint64_t> n_new{0};
        if (fail_at && ++n_new >= fail_at) { *p = nullptr; *got = 0; return hipErrorOutOfMemory; }
        hipError_t e = hipMalloc(p, want);
        { std::lock_guard<std::mutex> lk(mu_); t_malloc_ += std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0; n_malloc_++; }
        if (e != hipSuccess) {                      // make room: release what the pool holds for this device and try once more
            (void)hipGetLastError();
            trim_dev(dev, 0);
            e = hipMalloc(p, want);
            if (e != hipSuccess) (void)hipGetLastError();
        }
        *got = want;
        if (e == hipSuccess) { std::lock_guard<std::mutex> lk(mu_); account(dev, (long long)want); }
        return e;
    }
    void put(int dev, void *p, size_t bytes)
    {
        if (!p) return;
        std::lock_guard<std::mutex> lk(mu_);
        PerDev &D = dev_[dev];
        D.free_.emplace(bytes, p); D.held += bytes;
        account(dev, -(long long)bytes);
    }
    size_t held(int dev) { std::lock_guard<std::mutex> lk(mu_); return dev_[dev].held; }          // bytes waiting for the next call
    size_t release(int dev) { const size_t h = held(dev); trim_dev(dev, 0); return h; }          // the idle buffers of one device back to the runtime; returns their bytes
    size_t release_all() { size_t h = 0; std::vector<int> devs; { std::lock_guard<std::mutex> lk(mu_); for (auto &kv : dev_) { devs.push_back(kv.first); h += kv.second.held; } } for (int d : devs) trim_dev(d, 0); return h; }
    // high-water mark of the bytes in use (handed out and not yet returned) on any one device since reset_peak()
    size_t peak() { std::lock_guard<std::mutex> lk(mu_); size_t m = 0; for (auto &kv : dev_) m = std::max(m, kv.second.peak); return m; }
    void reset_peak() { std::lock_guard<std::mutex> lk(mu_); for (auto &kv : dev_) kv.second.peak = kv.second.used; t_malloc_ = 0; n_malloc_ = 0; }
    void malloc_time(double &t, uint64_t &n) { std::lock_guard<std::mutex> lk(mu_); t = t_malloc_; n = n_malloc_; }          // seconds inside hipMalloc (summed over the threads) and calls since reset_peak()
    void trim(size_t keep_per_dev)                  // (only when no kernel of this path is in flight)
    {
        std::vector<int> devs;
        { std::lock_guard<std::mutex> lk(mu_); for (auto &kv : dev_) devs.push_back(kv.first); }
        for (int d : devs) trim_dev(d, keep_per_dev);
    }
private:
    struct PerDev { std::multimap<size_t, void *> free_; size_t held = 0, used = 0, peak = 0; };
    void account(int dev, long long delta) { PerDev &D = dev_[dev]; D.used = (size_t)((long long)D.used + delta); if (D.used > D.peak) D.peak = D.used; }     // (mu_ held)
    void trim_dev(int dev, size_t keep)
    {
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> lk(mu_);
            PerDev &D = dev_[dev];
            while (D.held > keep && !D.free_.empty()) { auto it = D.free_.begin(); drop.push_back(it->second); D.held -= it->first; D.free_.erase(it); }
        }
        if (drop.empty()) return;
        int cur = -1; (void)hipGetDevice(&cur);
        if (cur != dev) (void)hipSetDevice(dev);
        for (void *q : drop) (void)hipFree(q);
        if (cur != dev && cur >= 0) (void)hipSetDevice(cur);
    }
    std::mutex mu_; std::map<int, PerDev> dev_; double t_malloc_ = 0; uint64_t n_malloc_ = 0;
};
DevPool g_pool;

// a device buffer from the pool; `dev` is the PHYSICAL device
template <class T> struct DevBuf {
    T *p = nullptr; size_t cap = 0;                      // cap in elements
    size_t bytes_ = 0; int dev_ = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete; DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), cap(o.cap), bytes_(o.bytes_), dev_(o.dev_) { o.p = nullptr; o.cap = 0; o.bytes_ = 0; }
    ~DevBuf() { g_pool.put(dev_, p, bytes_); }
    void release() { g_pool.put(dev_, p, bytes_); p = nullptr; cap = 0; bytes_ = 0; }
    hipError_t need(int dev, size_t n, bool slack = true)          // contents are NOT kept
    {
        if (n <= cap && p && dev == dev_) return hipSuccess;
        release();
        dev_ = dev;
        const size_t want = slack ? n + n / 2 + 1024 : (n ? n : 1);
        void *q = nullptr; size_t got = 0;
        hipError_t e = g_pool.get(dev, &q, want * sizeof(T), &got);
        if (e == hipSuccess) { p = (T *)q; bytes_ = got; cap = got / sizeof(T); }
        return e;
    }
};

struct Mapped {
    const uint8_t *p = nullptr; size_t n = 0; int fd = -1;          // (the descriptor stays open: the uploader reads through it)
    ~Mapped() { if (p) munmap(const_cast<uint8_t *>(p), n); if (fd >= 0) ::close(fd); }
    // regular = false: not a file this path takes (a pipe, a device ...) -- it has NOT been opened (opening a FIFO blocks until a
    // writer appears, and closing it again may break that writer's pipe before the host pipeline opens it)
    bool open(const char *path, bool &regular)
    {
        regular = false;
        struct stat st;
        if (stat(path, &st) != 0) return false;
        if (!S_ISREG(st.st_mode)) return true;
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); fd = -1; return true; }
        regular = true;
        n = (size_t)st.st_size;
        if (n) {
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { n = 0; return false; }
            p = (const uint8_t *)m;
            madvise(m, n, MADV_SEQUENTIAL);
        }
        return true;
    }
};

// ---- page cache -> pinned memory on a few threads, with pread: reading a mapping instead takes a fault per 64 KiB and does
// 3 GB/s a thread (with four of those the whole path once ran at the 11 GB/s of that copy, whatever the decoder did); the
// mapping stays for what the host looks at (headers, trailers, gaps)
// pinned staging buffers are kept from call to call (allocating and releasing two 32 MiB pinned buffers costs several milliseconds,
// which is most of what a call on a small file spends outside its pipeline); MF_KEEP_BUFFERS=0 releases them with the call
class PinnedCache {
public:
    hipError_t get(uint8_t **p, size_t bytes)
    {
        { std::lock_guard<std::mutex> lk(mu_); auto it = free_.lower_bound(bytes); if (it != free_.end() && it->first <= std::max<size_t>(bytes * 2 + 4096, ((size_t)32 << 20) + 4096)) { *p = it->second; size_[*p] = it->first; free_.erase(it); return hipSuccess; } }
        void *q = nullptr;
        hipError_t e = hipHostMalloc(&q, bytes, hipHostMallocPortable);
        if (e == hipSuccess) { *p = (uint8_t *)q; std::lock_guard<std::mutex> lk(mu_); size_[*p] = bytes; }
        return e;
    }
    void put(uint8_t *p)
    {
        if (!p) return;
        const char *kb = getenv("MF_KEEP_BUFFERS");
        std::unique_lock<std::mutex> lk(mu_);
        const size_t n = size_[p];
        if ((kb && kb[0] == '0') || free_.size() >= 8) { size_.erase(p); lk.unlock(); (void)hipHostFree(p); return; }
        free_.emplace(n, p);
    }
    void prefill(int n, size_t bytes)          // n buffers of `bytes` into the cache (a thread of its own does this while a cold call maps its files)
    {
        for (int i = 0; i < n; i++) {
            { std::lock_guard<std::mutex> lk(mu_); if (free_.size() >= 4) return; }
            void *q = nullptr;
            if (hipHostMalloc(&q, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return; }
            std::lock_guard<std::mutex> lk(mu_); size_[(uint8_t *)q] = bytes; free_.emplace(bytes, (uint8_t *)q);
        }
    }
    size_t idle() { std::lock_guard<std::mutex> lk(mu_); return free_.size(); }          // staging buffers waiting for the next call
    void clear() { std::vector<uint8_t *> v; { std::lock_guard<std::mutex> lk(mu_); for (auto &kv : free_) { v.push_back(kv.second); size_.erase(kv.second); } free_.clear(); } for (uint8_t *q : v) (void)hipHostFree(q); }
private:
    std::mutex mu_; std::multimap<size_t, uint8_t *> free_; std::map<uint8_t *, size_t> size_;
};
PinnedCache g_pinned;
std::atomic<bool> g_short_lived{false};          // the process makes one file-level call and ends (a CLI): mf_set_option("short_lived", "1")

// (the threads are the stager's own and live as long as it does: starting eight threads per 32 MiB piece was a tenth of the time of a read)
struct Stager {
    std::vector<uint8_t *> buf; size_t piece = 0; int fd = -1; int nthr = 8;
    ~Stager()
    {
        { std::lock_guard<std::mutex> lk(mu_); quit_ = true; gen_++; }
        cv_.notify_all();
        for (auto &t : pool_) t.join();
        for (auto &b : buf) g_pinned.put(b);
    }
    hipError_t init(size_t piece_bytes, int fd_, int n_buf = 2)
    {
        piece = piece_bytes; fd = fd_;
        nthr = (int)std::min<uint64_t>(16, std::max<uint64_t>(1, env_u64("MF_UPLOAD_THREADS", 8)));
        buf.assign((size_t)n_buf, nullptr);
        for (auto &b : buf) { hipError_t e = g_pinned.get(&b, piece + 256); if (e != hipSuccess) return e; }
        if (piece >= ((size_t)1 << 20)) for (int t = 1; t < nthr; t++) pool_.emplace_back([this, t] { work(t); });
        return hipSuccess;
    }
    bool read(int b, size_t off, size_t len)          // false: the file could not be read (truncated under us, an I/O error)
    {
        const int nt = (len < ((size_t)1 << 20) || pool_.empty()) ? 1 : nthr;
        ok_ = true;
        if (nt > 1) {
            { std::lock_guard<std::mutex> lk(mu_); dst_ = buf[(size_t)b]; off_ = off; len_ = len; nt_ = nt; left_ = nt - 1; gen_++; }
            cv_.notify_all();
        } else { dst_ = buf[(size_t)b]; off_ = off; len_ = len; nt_ = 1; }
        part(0);
        if (nt > 1) { std::unique_lock<std::mutex> lk(mu_); done_.wait(lk, [&] { return left_ == 0; }); }
        return ok_;
    }
private:
    void part(int t)
    {
        size_t a = len_ * (size_t)t / (size_t)nt_; const size_t e = len_ * (size_t)(t + 1) / (size_t)nt_;
        while (a < e) {
            const ssize_t got = pread(fd, dst_ + a, e - a, (off_t)(off_ + a));
            if (got < 0 && errno == EINTR) continue;
            if (got <= 0) { ok_ = false; return; }
            a += (size_t)got;
        }
    }
    void work(int t)
    {
        uint64_t seen = 0;
        for (;;) {
            { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [&] { return gen_ != seen; }); seen = gen_; if (quit_) return; }
            if (t < nt_) part(t);
            { std::lock_guard<std::mutex> lk(mu_); if (t < nt_ && --left_ == 0) done_.notify_all(); }
        }
    }
    std::vector<std::thread> pool_; std::mutex mu_; std::condition_variable cv_, done_; uint64_t gen_ = 0; bool quit_ = false;
    uint8_t *dst_ = nullptr; size_t off_ = 0, len_ = 0; int nt_ = 1, left_ = 0; std::atomic<bool> ok_{true};
};

// ---- A file of up to 512 MiB goes to the device FROM WHERE THE PAGE CACHE HOLDS IT (round 5): its read-only mapping is registered with the
// runtime (hipHostRegister, read-only) and the copy engine reads the pages themselves -- no staging buffers to pin (0.2 ms per MiB, which a
// cold call of a small file pays in full), no host thread touching a byte.  The page tables are filled first (madvise POPULATE_READ, a
// thread per 64 MiB: the pages are in the page cache, nothing is read) -- registering pages the process has not touched faults them in one
// by one, 2-8 GB/s.  Larger files go through pinned staging buffers: what registering costs there is host work per page -- page tables
// 0.7, hipHostRegister 0.4, hipHostUnregister 1.3 and munmap of the filled mapping 0.7 ms per 100 MiB -- and even with all of it on
// threads of its own, ahead of and behind the copies, configs[4] took 0.30 s against 0.236 s staged, its plain text 0.284 s (0.07 s of it the
// munmap) against 0.22-0.30 s (profiles/r05/g_upload_registered_vs_staged.txt).  Where a mapping cannot be registered at all (a file
// system whose pages cannot be pinned) ensure() says no and the caller stages as well.
// This is the COLD call's way.  Registering is host work with every call (2.4 ms per 100 MiB, and the munmap), staging buffers are pinned once and
// kept: a warm call of a 0.16 GB plain file took 7.1 ms staged and 16.1 ms registered (profiles/r05/e_masks_ab.txt, g_masks_ab_after.txt).  So
// a file is registered only while the process holds no idle staging buffers; a process that lives on (not "short_lived") pins a set behind its
// first call (StreamSets::stage_later), and the calls after that stage.
class PinnedMap {
public:
    PinnedMap(const uint8_t *p, size_t n) : p_(p), n_(n)
    {
        static const bool off = getenv("MF_UPLOAD_STAGED") != nullptr;
        usable_ = !off && p && n && n <= (size_t)env_u64("MF_UPLOAD_REGISTER_MAX_MB", 512) << 20 && g_pinned.idle() == 0;
    }
    ~PinnedMap()
    {
        if (registered_) {
            int cur = -1; (void)hipGetDevice(&cur);
            for (auto &e : ev_) { (void)hipSetDevice(e.first); (void)hipEventSynchronize(e.second); }          // (the copies that read the mapping have run)
            (void)hipHostUnregister(const_cast<uint8_t *>(p_));
            if (cur >= 0) (void)hipSetDevice(cur);
        }
        for (auto &e : ev_) { (void)hipSetDevice(e.first); (void)hipEventDestroy(e.second); }
    }
    // the file's bytes can be given to hipMemcpyAsync as they lie in the mapping
    bool ensure()
    {
        if (!usable_ || registered_) return usable_;
        constexpr size_t PART = (size_t)64 << 20;
        if (n_ > PART / 2) {
            std::vector<std::thread> th;
            for (size_t q = PART; q < n_; q += PART) th.emplace_back([this, q] { fill(q, std::min(n_, q + PART)); });
            fill(0, std::min(n_, PART));
            for (auto &x : th) x.join();
        }
        const size_t len = (n_ + 4095) & ~(size_t)4095;          // (the mapping runs to the end of the file's last page)
        if (hipHostRegister(const_cast<uint8_t *>(p_), len, hipHostRegisterPortable | hipHostRegisterReadOnly) != hipSuccess) { (void)hipGetLastError(); usable_ = false; }
        else registered_ = true;
        return usable_;
    }
    // a copy that reads the mapping has been issued on stream st of device dev: the registration stays until it has run
    bool after_copy(int dev, hipStream_t st)
    {
        hipEvent_t ev = nullptr;
        for (auto &e : ev_) if (e.first == dev) ev = e.second;
        if (!ev) { if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return false; ev_.emplace_back(dev, ev); }
        return hipEventRecord(ev, st) == hipSuccess;
    }
private:
    void fill(size_t a, size_t b) const
    {
#ifndef MADV_POPULATE_READ
        constexpr int MADV_POPULATE_READ = 22;          // Linux 5.14
#endif
        static std::atomic<bool> have{true};
        if (have.load(std::memory_order_relaxed) && madvise(const_cast<uint8_t *>(p_) + a, b - a, MADV_POPULATE_READ) == 0) return;
        have.store(false, std::memory_order_relaxed);
        unsigned acc = 0;
        for (size_t q = a; q < b; q += 4096) acc += *(const volatile uint8_t *)(p_ + q);
        (void)acc;
    }
    const uint8_t *p_; size_t n_; bool usable_ = false, registered_ = false; std::vector<std::pair<int, hipEvent_t>> ev_;
};

// ---- the streams of this path, per device.  What a stream costs to make (profiles/r05/a_stream_probe.log): a CU-masked one is a
// hardware queue of its own, 16 ms, always; a plain one 16-30 ms while the process has fewer than four queues, 2-3 ms afterwards
// (it then shares one); the runtime makes them one after the other whoever asks, without holding up launches on the streams that
// exist.  The reference calls this path a process at a time (utility/helper.py:78-86), so a call starts cold more often than
// not: the streams are made ONCE per process and device by a maker thread, in the order a cold call needs them, while the call
// maps its files, pins its staging buffers and reads the first bytes -- whoever needs a stream that is not there yet waits for it.
//   decode streams (dec[]): CU-masked, so that decode wavefronts leave a few CUs alone (below) and a decode kernel of 10-30 ms never
//     sits in front of a short kernel in a shared queue; shared by the mates of a call; the first is made first, the rest
//     behind everything a small file needs;
//   copy: the uploads of every mate (they share the link to the device anyway);
//   post[]: per mate, everything behind a slab's decode kernel -- link, marker resolution, CRC, in that order, so one stream;
//     plain ones, and (made late, for inputs large enough to keep the chip full of decode wavefronts for a long time) ones
//     masked to the CUs the decode streams leave free: profiles/r04/g_configs4_link_stream_ab.txt.
// Never destroyed (destroying a CU-masked stream right after use was seen to hang inside the runtime, ROCm 7.2) -- except under a
// profiler, at exit.
constexpr uint32_t GZ_NSTREAM = 4, GZ_NPOST = 2;          // (four decode streams do what ten did, profiles/r05/g_dec_streams_ab.txt: six hardware queues fewer to make, to hold and to tear down at exit)
struct DeviceStreams {
    int device = -1;
    hipStream_t dec[GZ_NSTREAM] = {}, copy = nullptr, post[GZ_NPOST] = {}, post_masked[GZ_NPOST] = {}, post_b[GZ_NPOST] = {};
    std::atomic<uint32_t> n_dec{0};
    std::mutex mu; std::condition_variable cv; int made = 0; bool failed = false, post_busy[GZ_NPOST] = {false, false};
    std::thread maker; std::atomic<bool> stop{false};
    uint32_t words = 0; int n_cu = 0; std::vector<uint32_t> mask, mask_rest; bool masked = false;
    // the order of making: what a cold call on a small file waits for comes first
    enum What { DEC0, COPY, POST0, POST1, POSTM0, POSTB0, DEC1, DEC2, DEC3, POSTM1, POSTB1, N_WHAT };          // (a large input's first link step waits for POSTM0: in front of the further decode streams)
    bool make_masked(hipStream_t *q, const std::vector<uint32_t> &m) const
    {
        if (masked && hipExtStreamCreateWithCUMask(q, words, m.data()) == hipSuccess) return true;
        (void)hipGetLastError();
        return hipStreamCreateWithFlags(q, hipStreamNonBlocking) == hipSuccess;
    }
    // The maker goes as far as somebody has asked for (want): a small file's call asks for the first decode stream, the copy stream and the
    // post streams and nothing else -- every further decode stream is asked for by the launch that could have used it (which takes an
    // existing one meanwhile), the masked post streams by a large input.  The runtime makes streams one after the other, whoever asks: a
    // maker that ran through all of them (sixteen then) at once held up the consumers' own streams for a tenth of a second (profiles/r05/c_cold_calls_factory.log).
    int want = POST1 + 1;
    void ask(int upto) { { std::lock_guard<std::mutex> lk(mu); if (upto > want) want = upto; } cv.notify_all(); }
    void run()
    {
        if (hipSetDevice(device) != hipSuccess) { fail_(); return; }
        for (int w = 0; w < N_WHAT && !stop; w++) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || want > w; }); if (stop) return; }
            bool ok = true;
            switch (w) {
            case DEC0: ok = make_masked(&dec[0], mask); if (ok) n_dec = 1; break;
            case COPY: ok = hipStreamCreateWithFlags(&copy, hipStreamNonBlocking) == hipSuccess; break;
            case POST0: case POST1: ok = hipStreamCreateWithFlags(&post[w - POST0], hipStreamNonBlocking) == hipSuccess; break;
            case DEC1: case DEC2: case DEC3: ok = make_masked(&dec[1 + w - DEC1], mask); if (ok) n_dec = 2 + (uint32_t)(w - DEC1); break;
            case POSTM0: case POSTM1: ok = make_masked(&post_masked[w == POSTM0 ? 0 : 1], mask_rest); break;
            case POSTB0: case POSTB1: ok = hipStreamCreateWithFlags(&post_b[w == POSTB0 ? 0 : 1], hipStreamNonBlocking) == hipSuccess; break;
            }
            if (!ok) { fail_(); return; }
            { std::lock_guard<std::mutex> lk(mu); made = w + 1; }
            cv.notify_all();
            if (w == DEC0) cold_mark("streams: first decode stream made");
            if (w == POST1) cold_mark("streams: copy and post streams made");
        }
    }
    void fail_() { { std::lock_guard<std::mutex> lk(mu); failed = true; } cv.notify_all(); }
    bool wait_for(What w) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return failed || made > (int)w; }); return made > (int)w; }
    // limit: decode streams this call may use.  A CU-masked stream is a hardware queue with its own save area on the device -- 12 of them hold
    // 2 GB (configs[4]: 15.8 GB in use against 12.7 GB of buffers; on plain streams the difference is 0.9 GB) --, so an input below a gigabyte,
    // whose memory is to follow its size, gets three.
    hipStream_t pick_dec(uint32_t seq, uint32_t limit = GZ_NSTREAM)
    {
        if (!wait_for(DEC0)) return nullptr;
        const uint32_t n = std::min<uint32_t>(n_dec.load(), std::max<uint32_t>(1, limit));
        if (seq >= n && n < std::min<uint32_t>(GZ_NSTREAM, limit)) ask(DEC1 + (int)n);          // (one more for the next launch)
        return dec[seq % std::max<uint32_t>(1, n)];
    }
    hipStream_t copy_stream() { return wait_for(COPY) ? copy : nullptr; }
    // a post stream for one mate of one call (given back with give_post); want_masked: a large input
    hipStream_t take_post(bool want_masked, int *slot)
    {
        int k = -1;
        { std::lock_guard<std::mutex> lk(mu); for (int i = 0; i < (int)GZ_NPOST; i++) if (!post_busy[i]) { post_busy[i] = true; k = i; break; } }
        *slot = k;
        if (!wait_for(k == 1 ? POST1 : POST0)) {          // (a single-end call does not wait for the second post stream)
            if (k >= 0) { std::lock_guard<std::mutex> lk(mu); post_busy[k] = false; }          // the maker has failed: the slot is not taken
            *slot = -1;
            return nullptr;
        }
        if (k < 0) { hipStream_t q = nullptr; return hipStreamCreateWithFlags(&q, hipStreamNonBlocking) == hipSuccess ? q : nullptr; }      // (more than two mates at a time on one device: concurrent calls)
        if (want_masked && masked) { ask((k == 0 ? POSTM0 : POSTM1) + 1); if (wait_for(k == 0 ? POSTM0 : POSTM1)) return post_masked[k]; }
        return post[k];
    }
    // the second stream of a mate's post work (marker resolution of the chunks' bodies and the CRC, behind the link step they belong to): a plain one
    hipStream_t take_post_b(int slot)
    {
        if (slot < 0 || slot >= (int)GZ_NPOST) return nullptr;
        ask((slot == 0 ? POSTB0 : POSTB1) + 1);
        return wait_for(slot == 0 ? POSTB0 : POSTB1) ? post_b[slot] : nullptr;
    }
    void give_post(int slot, hipStream_t q)
    {
        if (slot >= 0) { std::lock_guard<std::mutex> lk(mu); post_busy[slot] = false; }
        else if (q) { (void)hipStreamSynchronize(q); (void)hipStreamDestroy(q); }
    }
};
class StreamSets {
public:
    // the streams of physical device `device` (the maker is started on first use and runs on by itself)
    // Two sets per device: one of plain streams -- what every call on a file of less than a gigabyte uses -- and one whose decode streams are
    // CU-masked, for large inputs.  A CU-masked stream is a hardware queue of its own: 16 ms to make, and the process's EXIT waits for the
    // kernel driver to tear each of them down -- 0.2-0.25 s of a process that lived for 0.4 (profiles/r05/d_exit_probe.log: a quality-filter
    // call on a 2 M-pair .gz pair, caller saw 0.61 / 0.69 s with masks, 0.39 / 0.41 s without).  A process per call is the reference's
    // boundary, so the masks are worth their price only where the chip is full of decode wavefronts for long.
    DeviceStreams *get(int device, bool want_masks, std::string &err)
    {
        std::lock_guard<std::mutex> lk(mu_);
        const int key = device * 2 + (want_masks ? 1 : 0);
        auto it = dev_.find(key);
        if (it != dev_.end()) return it->second;
        std::unique_ptr<DeviceStreams> d(new DeviceStreams());
        d->device = device;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) { err = "hipGetDeviceProperties failed"; return nullptr; }
        const int n_cu = prop.multiProcessorCount, words = (n_cu + 31) / 32;
        // Whatever else has to run while decode wavefronts fill the chip -- the link step, marker resolution, CRC, the consumers'
        // kernels, all short and all on some host thread's critical path -- needs CUs of its own.  Four per XCD (mask bit b is a CU of XCD b mod 8).
        int reserve = (int)env_u64("MF_GZDEV_RESERVED_CUS", 32);
        reserve = std::max(8, std::min(n_cu / 2, reserve)) & ~7;
        d->mask.assign((size_t)words, 0);
        for (int b = 0; b < n_cu - reserve; b++) d->mask[b / 32] |= 1u << (b % 32);
        d->mask_rest.resize(d->mask.size());
        for (size_t i = 0; i < d->mask.size(); i++) d->mask_rest[i] = ~d->mask[i];
        if (n_cu % 32) d->mask_rest.back() &= (1u << (n_cu % 32)) - 1;
        d->words = (uint32_t)words; d->n_cu = n_cu;
        d->masked = want_masks && n_cu >= 64 && !getenv("MF_GZDEV_NO_CUMASK");
        DeviceStreams *dp = d.release();
        dp->maker = std::thread([dp] { dp->run(); });
        dev_[key] = dp;
        return dp;
    }
    // Under rocprofv3 a process that still owns CU-masked streams when it exits dies in the profiler's finaliser (SIGSEGV below
    // __cxa_finalize, after the profile has been written; without a profiler the exit is clean).  So when a profiler is loaded
    // the streams are destroyed here, at exit, after a device synchronisation -- not otherwise: destroying such a stream was seen to
    // hang now and then, and an exit that hangs is worse than one a profiler complains about.
    // the code objects of the decoder and of the line kernels, loaded on a thread of their own, once per process
    void prefill_pinned(int device)
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (prefill_started_) return;
        prefill_started_ = true;
        prefill_ = std::thread([device] {
            if (hipSetDevice(device) != hipSuccess) return;
            gz_preload(); ingest_preload(); cold_mark("prefetch: code objects of the decoder and the line kernels loaded");          // (no staging buffers: the uploads read the page cache's pages)
        });
    }
    // staging buffers for the calls to come, pinned on a thread of their own behind a process's first call (not for a process that makes one call and ends)
    void stage_later()
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (stage_started_ || g_short_lived.load() || g_pinned.idle()) return;
        stage_started_ = true;
        stage_ = std::thread([] { g_pinned.prefill(4, (size_t)32 << 20); });
    }
    void forget_staging() { std::lock_guard<std::mutex> lk(mu_); if (stage_.joinable()) stage_.join(); stage_started_ = false; }          // (the cache has been emptied on request: the next call is a cold one again)
    ~StreamSets()
    {
        if (prefill_.joinable()) prefill_.join();
        if (stage_.joinable()) stage_.join();
        for (auto &kv : dev_) { kv.second->stop = true; kv.second->cv.notify_all(); if (kv.second->maker.joinable()) kv.second->maker.join(); }
        const char *pre = getenv("LD_PRELOAD");
        const bool profiled = (pre && strstr(pre, "rocprof")) || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") || getenv("ROCP_TOOL_LIBRARIES") || getenv("MF_GZDEV_DESTROY_STREAMS_AT_EXIT");
        if (!profiled) return;
        for (auto &kv : dev_) {
            if (hipSetDevice(kv.second->device) != hipSuccess) continue;
            (void)hipDeviceSynchronize();
            DeviceStreams &D = *kv.second;
            for (auto &q : D.dec) if (q) (void)hipStreamDestroy(q);
            for (auto &q : D.post) if (q) (void)hipStreamDestroy(q);
            for (auto &q : D.post_masked) if (q) (void)hipStreamDestroy(q);
            for (auto &q : D.post_b) if (q) (void)hipStreamDestroy(q);
            if (D.copy) (void)hipStreamDestroy(D.copy);
        }
    }
private:
    std::mutex mu_; std::map<int, DeviceStreams *> dev_; std::thread prefill_, stage_; bool prefill_started_ = false, stage_started_ = false;
};
StreamSets g_streams;

// ---- how many text buffers a mate may hold at a time (the producer waits for one to come back)
struct Slots {
    std::mutex mu; std::condition_variable cv; int free_ = 0; std::atomic<bool> *stop = nullptr;
    bool take() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return free_ > 0 || (stop && *stop); }); if (free_ <= 0) return false; free_--; return true; }
    void give() { { std::lock_guard<std::mutex> lk(mu); free_++; } cv.notify_all(); }
    void wake() { cv.notify_all(); }
    bool none_free() { std::lock_guard<std::mutex> lk(mu); return free_ <= 0; }
};

// ---- the text of one piece of an input file on one device.  In front of the text: `pad` readable bytes -- the 32 KiB deflate
// window of the piece's first chunk (written by the link step), and room for the carry: the head of the record that the piece
// before left unfinished is copied there, so that a record is always contiguous.
constexpr size_t TEXT_FRONT = 32768 + 256;     // a damaged stream may point a full window back from its first byte
struct TextBuf {
    int dev = 0, ldev = 0;                      // physical / logical device
    uint8_t *raw = nullptr; size_t raw_bytes = 0;
    uint8_t *p = nullptr; size_t pad = 0, cap = 0;      // p = raw + pad; cap text bytes fit behind p (and 64 more are readable)
    Slots *slots = nullptr;
    hipEvent_t ready = nullptr; bool ready_recorded = false;      // recorded by the producer behind the last kernel that writes the text: a consumer's stream waits for it
    hipEvent_t ready_event() { if (!ready) { (void)hipSetDevice(dev); if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess) ready = nullptr; } ready_recorded = ready != nullptr; return ready; }
    ~TextBuf() { if (ready) { (void)hipSetDevice(dev); if (ready_recorded) (void)hipEventSynchronize(ready); (void)hipEventDestroy(ready); } g_pool.put(dev, raw, raw_bytes); if (slots) slots->give(); }      // (a piece that is dropped unread: whatever still writes it finishes first)
    static hipError_t make(std::unique_ptr<TextBuf> &out, int dev, int ldev, size_t pad, size_t text_bytes, Slots *slots)
    {
        std::unique_ptr<TextBuf> b(new TextBuf());
        b->dev = dev; b->ldev = ldev;
        pad = (pad + 255) & ~(size_t)255;
        hipError_t e = g_pool.get(dev, (void **)&b->raw, pad + text_bytes + 64, &b->raw_bytes);
        if (e != hipSuccess) { b->raw = nullptr; b->raw_bytes = 0; if (slots) slots->give(); return e; }
        b->slots = slots;
        b->pad = pad; b->p = b->raw + pad; b->cap = b->raw_bytes - pad - 64;
        out = std::move(b);
        return hipSuccess;
    }
};

// a range of an input's text that has become available, in order
struct TextPiece { std::unique_ptr<TextBuf> buf; uint64_t T0 = 0, len = 0; bool last = false; double grow = 1.0; };      // grow: how much larger than this one the file's pieces become (the first slabs of a .gz are short)

// ---- a .gz file's bytes -> the rings of the devices that decode it, in order, a piece at a time.  The slab layout says which
// devices want which bytes; the producer moves the low-water mark (everything in front of it has been linked) and the uploader
// keeps within a ring's length of it.  Decode streams wait for the event of the piece that completes the range they read.
// The thread pins its staging buffers itself and takes the device's copy stream when it is made: a call's set-up does not wait for either.
class GzUploader {
public:
    struct Lane { int dev = 0; uint8_t *ring = nullptr; DeviceStreams *ds = nullptr; hipStream_t st = nullptr; };
    ~GzUploader()
    {
        stop_ = true; cv_.notify_all();
        if (th_.joinable()) th_.join();
        for (size_t l = 0; l < lanes_.size(); l++) {
            (void)hipSetDevice(lanes_[l].dev);
            // run() leaves early when it is stopped (the longer mate of a pair, a failed call) or fails: copies it has queued on the
            // device's shared copy stream may still be on their way into the ring and out of the staging buffers, and both go back to
            // their pools right after this destructor -- nothing of this uploader may be in flight then
            if (lanes_[l].st) (void)hipStreamSynchronize(lanes_[l].st);
            for (auto &e : ev_[l]) if (e) (void)hipEventDestroy(e);
            for (int b = 0; b < UP_BUFS_MAX; b++) if (free_ev_[l][b]) (void)hipEventDestroy(free_ev_[l][b]);
        }
    }
    // piece_lanes[i]: bit l set = lane l wants piece i
    void start(const uint8_t *map, int fd, size_t n, size_t ring_bytes, size_t piece, std::vector<Lane> lanes, std::vector<uint64_t> piece_lanes)
    {
        map_ = map; fd_ = fd; n_ = n; ring_ = ring_bytes; piece_ = piece; lanes_ = std::move(lanes); want_ = std::move(piece_lanes);
        ev_.assign(lanes_.size(), std::vector<hipEvent_t>(want_.size(), nullptr));
        free_ev_.assign(lanes_.size(), std::array<hipEvent_t, UP_BUFS_MAX>{});
        for (auto &u : stage_used_) u = 0;
        n_bufs_ = (int)std::max<uint64_t>(2, std::min<uint64_t>(UP_BUFS_MAX, env_u64("MF_GZDEV_UPLOAD_BUFS", 2)));
        low_ = 0;
        th_ = std::thread([this] { run(); });
    }
    void set_low_water(uint64_t byte) { { std::lock_guard<std::mutex> lk(mu_); if (byte > low_) low_ = byte; } cv_.notify_all(); }
    // the copy of bytes [0, upto) has been issued (so wait_for would not block the host)
    bool issued(size_t upto)
    {
        if (n_ == 0 || upto == 0) return true;
        if (upto > n_) upto = n_;
        std::lock_guard<std::mutex> lk(mu_);
        return failed_ || enqueued_ > (upto - 1) / piece_;
    }
    // make `st` (a stream of lane l's device) wait until the bytes [.., upto) that lane l wants are in its ring.  false: the uploader failed (failure(): why)
    bool wait_for(size_t l, hipStream_t st, size_t upto)
    {
        if (n_ == 0 || upto == 0) return true;
        if (upto > n_) upto = n_;
        size_t j = (upto - 1) / piece_;
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return enqueued_ > j || failed_; });
        if (failed_) return false;
        while (!((want_[j] >> l) & 1)) { if (!j) return true; j--; }       // (the copy stream is in order: the last piece of this lane at or in front of j)
        return hipStreamWaitEvent(st, ev_[l][j], 0) == hipSuccess;
    }
    int failure() { std::lock_guard<std::mutex> lk(mu_); return fail_rc_; }
private:
    void run()
    {
        if (!lanes_.empty() && hipSetDevice(lanes_[0].dev) != hipSuccess) { fail_(MF_E_HIP); return; }
        PinnedMap reg(map_, n_);          // a file of up to 512 MiB: the copy engine reads the page cache's own pages; otherwise staging buffers
        bool staged = false;
        const size_t np = want_.size();
        for (size_t i = 0; i < np && !stop_; i++) {
            const size_t off = i * piece_, len = std::min(piece_, n_ - off);
            if (!want_[i]) { { std::lock_guard<std::mutex> lk(mu_); enqueued_ = i + 1; } cv_.notify_all(); continue; }
            const double t_a = now_s();
            {   // not more than a ring's length ahead of what has been linked
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || off + len + 256 <= low_ + ring_; });
                if (stop_) return;
            }
            const double t_b = now_s();
            const int b = (int)(i % (size_t)n_bufs_);
            const bool direct = reg.ensure();
            const uint8_t *src = map_ + off;
            double t_c = now_s();
            if (!direct) {
                if (!staged) { const hipError_t e = stage_.init(piece_, fd_, n_bufs_); if (e != hipSuccess) { fail_(e == hipErrorOutOfMemory ? MF_E_NOMEM : MF_E_HIP); return; } staged = true; }
                for (size_t l = 0; l < lanes_.size(); l++)          // the copies that read this staging buffer are done
                    if ((stage_used_[b] >> l) & 1) { if (hipSetDevice(lanes_[l].dev) != hipSuccess || hipEventSynchronize(free_ev_[l][b]) != hipSuccess) { fail_(MF_E_HIP); return; } }
                stage_used_[b] = 0;
                t_c = now_s();
                if (!stage_.read(b, off, len)) { fail_(MF_E_IO); return; }
                src = stage_.buf[(size_t)b];
            }
            t_ring_ += t_b - t_a; t_copy_wait_ += t_c - t_b; t_read_ += now_s() - t_c;
            if (i == 0) cold_mark(direct ? "uploader: the file's mapping registered" : "uploader: first piece of the file read into pinned memory");
            const size_t total = len;
            for (size_t l = 0; l < lanes_.size(); l++) {
                if (!((want_[i] >> l) & 1)) continue;
                Lane &L = lanes_[l];
                if (hipSetDevice(L.dev) != hipSuccess) { fail_(MF_E_HIP); return; }
                if (!L.st && !(L.st = L.ds->copy_stream())) { fail_(MF_E_HIP); return; }
                if (!ev_[l][i] && hipEventCreateWithFlags(&ev_[l][i], hipEventDisableTiming) != hipSuccess) { fail_(MF_E_HIP); return; }
                if (!free_ev_[l][b] && hipEventCreateWithFlags(&free_ev_[l][b], hipEventDisableTiming) != hipSuccess) { fail_(MF_E_HIP); return; }
                // (a piece never straddles the end of the ring -- the ring is a multiple of the piece)
                const size_t r0 = ring_mask_off(off), first = std::min(total, ring_ - r0);
                if (hipMemcpyAsync(L.ring + r0, src, first, hipMemcpyHostToDevice, L.st) != hipSuccess) { fail_(MF_E_HIP); return; }
                if (first < total && hipMemcpyAsync(L.ring, src + first, total - first, hipMemcpyHostToDevice, L.st) != hipSuccess) { fail_(MF_E_HIP); return; }
                if (off + len == n_) {          // readable and zero behind the last byte (256 bytes: they may straddle the end of the ring)
                    const size_t z0 = ring_mask_off(off + len), zf = std::min<size_t>(256, ring_ - z0);
                    if (hipMemsetAsync(L.ring + z0, 0, zf, L.st) != hipSuccess || (zf < 256 && hipMemsetAsync(L.ring, 0, 256 - zf, L.st) != hipSuccess)) { fail_(MF_E_HIP); return; }
                }
                if (hipEventRecord(ev_[l][i], L.st) != hipSuccess) { fail_(MF_E_HIP); return; }
                if (direct) { if (!reg.after_copy(L.dev, L.st)) { fail_(MF_E_HIP); return; } }
                else { if (hipEventRecord(free_ev_[l][b], L.st) != hipSuccess) { fail_(MF_E_HIP); return; } stage_used_[b] |= (uint64_t)1 << l; }
            }
            { std::lock_guard<std::mutex> lk(mu_); enqueued_ = i + 1; }
            cv_.notify_all();
            if (i == 0) cold_mark("uploader: first copy to the device issued");
        }
        for (auto &L : lanes_) { if (L.st && hipSetDevice(L.dev) == hipSuccess) (void)hipStreamSynchronize(L.st); }          // (before the windows are unregistered)
    }
    size_t ring_mask_off(size_t off) const { return off & (ring_ - 1); }
    void fail_(int rc) { { std::lock_guard<std::mutex> lk(mu_); failed_ = true; fail_rc_ = rc; } cv_.notify_all(); }
public:
    double t_ring_ = 0, t_copy_wait_ = 0, t_read_ = 0;          // the uploader thread's time: waiting for room in the ring, for the copy out of a staging buffer, reading the file
private:
    const uint8_t *map_ = nullptr; size_t n_ = 0, ring_ = 0, piece_ = 0; int fd_ = -1;
    std::vector<Lane> lanes_; std::vector<uint64_t> want_;
    static constexpr int UP_BUFS_MAX = 4;
    Stager stage_; uint64_t stage_used_[UP_BUFS_MAX] = {}; int n_bufs_ = 2;
    std::vector<std::vector<hipEvent_t>> ev_; std::vector<std::array<hipEvent_t, UP_BUFS_MAX>> free_ev_;
    std::thread th_; std::mutex mu_; std::condition_variable cv_; size_t enqueued_ = 0; uint64_t low_ = 0; bool failed_ = false; int fail_rc_ = MF_OK; std::atomic<bool> stop_{false};
};

// ---- one gzip file decoded on the device(s) (runs on the mate's producer thread, on its own streams)
class GzStream {
public:
    ~GzStream()
    {
        TRACE("~GzStream");
        up_.reset();                                  // (the uploader's copies go to the rings below)
        for (auto &L : lanes_) {                      // nothing of this decoder may be in flight when its buffers go back to the pool
            if (!L.ds) continue;
            (void)hipSetDevice(L.dev);
            for (uint32_t i = 0, n = L.ds->n_dec.load(); i < n; i++) (void)hipStreamSynchronize(L.ds->dec[i]);      // (the maker thread may still be writing the handles behind n)
            if (L.post) (void)hipStreamSynchronize(L.post);
            if (L.post_b && L.post_b != L.post) (void)hipStreamSynchronize(L.post_b);
            if (L.ev_base) (void)hipEventDestroy(L.ev_base);
            if (L.ev_a) (void)hipEventDestroy(L.ev_a);
            for (auto &C : L.crc) if (C.ev) (void)hipEventDestroy(C.ev);
            for (auto &e : L.ev_list) if (e) (void)hipEventDestroy(e);
        }
        reap(true);
        for (auto &S : slabs_) drop_events(*S);
        slabs_.clear(); cur_buf_.reset();
        for (auto &L : lanes_) {
            L.ring.release(); L.d_chunks.release(); L.d_window.release(); for (auto &C : L.crc) C.d.release(); L.d_acc.release(); L.d_acc_off.release(); L.d_link.release();
            (void)hipSetDevice(L.dev);
            for (auto &C : L.crc) if (C.h) (void)hipHostFree(C.h);
            if (L.h_list) (void)hipHostFree(L.h_list);
            if (L.ds) L.ds->give_post(L.post_slot, L.post);
        }
        if (h_win_) (void)hipHostFree(h_win_);
        if (h_chunks_) (void)hipHostFree(h_chunks_);
        TRACE("~GzStream done");
    }
    // data: the mapped file (what the host looks at: headers, trailers, gaps); devices: the logical devices that decode it
    // nslab: slabs whose decode kernels may be in flight per device (enough wavefronts to fill the chip: twelve for one file, seven each for two mates)
    // large: an input that keeps the chip full of decode wavefronts for a long time (its link streams are the CU-masked ones)
    // budget: device bytes this mate may hold in all -- ring, symbol buffers, code lists and its text_bufs text buffers; what is in flight follows
    // from it (0: no bound)
    int open(const uint8_t *data, size_t size, int fd, const std::vector<int> &devices, const std::string &path, Slots *slots, size_t carry_room,
             uint32_t nslab, bool large, uint64_t budget, uint32_t text_bufs, bool small_chunks, std::atomic<bool> *stop, std::string &err)
    {
        uint32_t NSLAB = std::max<uint32_t>(1, (uint32_t)env_u64("MF_GZDEV_SLABS_IN_FLIGHT", nslab));
        data_ = data; size_ = size; path_ = path; slots_ = slots; pad_ = TEXT_FRONT + carry_room; stop_ = stop;
        const uint32_t nl = (uint32_t)devices.size();
        const bool big = budget == 0 || budget >= ((uint64_t)2 << 30);          // (a mate's share of a call that plans for several gigabytes)
        dec_limit_ = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(GZ_NSTREAM, env_u64("MF_GZDEV_DEC_STREAMS", big ? GZ_NSTREAM : 3)));
        // chunks: large enough that a slab's fixed costs stay small, small enough that a file keeps the chip busy.  Measured over 0.1 / 0.3 / 1 / 3 GB of
        // .gz a mate x {64, 96, 128, 192, 256} KiB (profiles/r05/f_chunk_size_probe.txt, tools/chunk_size_probe.sh): 64 KiB is the fastest up to
        // 0.3 GB, 96 KiB at 1 GB (SE 0.087 s against 0.101 with 64 KiB and 0.123 with 256; PE 0.163 against 0.207), 192 KiB at 3 GB -- the file's
        // size / 10 923, between 64 and 192 KiB; the quality filter is fastest with 64 KiB at every size (0.44 s against 0.70-0.81 at 1 GB).
        size_t dflt = small_chunks ? (size_t)64 << 10 : size / 10923;
        dflt = std::min<size_t>(std::max<size_t>(dflt, (size_t)64 << 10), (size_t)192 << 10) & ~(size_t)4095;
        chunk_ = (size_t)env_u64("MF_GZDEV_CHUNK_BYTES", dflt);
        if (chunk_ < 1024) chunk_ = 1024;
        cps_ = (uint32_t)env_u64("MF_GZDEV_SLAB_CHUNKS", std::max<uint64_t>(256, ((uint64_t)128 << 20) / chunk_));
        if (cps_ < 1) cps_ = 1;
        // slabs in flight are counted in slabs of 512 chunks (the 256 KiB chunks of a large file): what fills the chip is chunks, and a file
        // of a gigabyte, with its smaller chunks and more of them to a slab, would hold twice the symbol room for nothing
        if (!getenv("MF_GZDEV_SLABS_IN_FLIGHT") && cps_ > 512) NSLAB = std::max<uint32_t>(2, (uint32_t)(((uint64_t)NSLAB * 512 + cps_ - 1) / cps_));
        // symbols of room per compressed byte: a first guess (FASTQ compresses three- to fivefold: 4.5, and 64 Ki symbols for the block behind the
        // range), then what the file has shown plus a quarter; a slab that overflows is decoded again with four times the room
        expand_ = getenv("MF_GZDEV_EXPAND") ? (double)env_u64("MF_GZDEV_EXPAND", 4) : 4.5;
        expand_fixed_ = getenv("MF_GZDEV_EXPAND") != nullptr;
        // The device memory of the path follows the INPUT.  What a chunk in flight holds: its symbol room (16-bit symbols, 4.5 : 1 and 64 Ki of
        // slack at first, then what the file has shown), 256 KiB of code lists, its bytes in the ring (twice: the ring is a power of two), and its share of
        // the text buffers (a slab's text each, 4.5 bytes per compressed byte, text_bufs of them over the slabs in flight).  The chunks in flight
        // are what the budget pays for -- in slabs small enough that four of them are in flight, so that upload, decode, link and the
        // consumers still overlap.  (Round 4 held 28 GB for a 0.6 GB pair: twelve slabs of a 5 GB file's size whatever the file.)
        if (budget && !getenv("MF_GZDEV_SLABS_IN_FLIGHT") && !getenv("MF_GZDEV_SLAB_CHUNKS")) {
            const uint64_t per_chunk = (uint64_t)sym_cap_first() * 2 + ((uint64_t)256 << 10) + (uint64_t)chunk_ * 2 + (uint64_t)chunk_ * 45 / 10 * std::max<uint32_t>(text_bufs, 1) / 4;
            const uint64_t fit = std::max<uint64_t>(64, budget / per_chunk);                     // chunks in flight the budget allows
            if ((uint64_t)NSLAB * cps_ > fit) {
                cps_ = (uint32_t)std::max<uint64_t>(16, std::min<uint64_t>(cps_, fit / 4));
                NSLAB = (uint32_t)std::max<uint64_t>(2, fit / cps_);
            }
        }
        text_piece_max_ = env_u64("MF_GZDEV_TEXT_PIECE", (uint64_t)1 << 30);
        size_t pos = 0;
        if (!member_header(pos, err)) return MF_E_FORMAT;
        base_byte_ = pos;
        n_chunks_ = (uint32_t)((size_ - base_byte_ + chunk_ - 1) / chunk_);
        if (n_chunks_ == 0) n_chunks_ = 1;
        if (cps_ > n_chunks_) cps_ = n_chunks_;
        // the ring: room for the slabs in flight, the bytes a slab's last chunk reads behind its range, and the uploader's pieces
        margin_ = (size_t)env_u64("MF_GZDEV_MARGIN", (size_t)8 << 20);
        const size_t slab_bytes = (size_t)cps_ * chunk_;
        size_t ring = pow2_ceil(std::max<size_t>(size_ + 512, 4096));
        {
            uint64_t want = env_u64("MF_GZDEV_RING_BYTES", 0);
            if (!want) want = (uint64_t)NSLAB * nl * slab_bytes + margin_ + 3 * std::min<uint64_t>((uint64_t)32 << 20, std::max<uint64_t>(slab_bytes, (uint64_t)4 << 20));      // (the slabs in flight, the read-ahead, three pieces of the uploader)
            want = pow2_ceil(std::max<uint64_t>(want, 4096));
            if (want < ring) ring = (size_t)want;
        }
        piece_ = std::min<size_t>(pow2_ceil((size_t)env_u64("MF_GZDEV_UPLOAD_PIECE_MB", 32)) << 20, std::max<size_t>(ring / 8, 512));
        for (;;) {          // slabs in flight: what the ring holds beside the margin and three pieces of the uploader
            const size_t fixed = margin_ + 3 * piece_ + 512;
            if (ring > fixed + slab_bytes) { max_inflight_ = (uint32_t)std::min<size_t>((size_t)NSLAB * nl, (ring - fixed) / slab_bytes); break; }
            ring <<= 1;
        }
        ring_ = ring;
        // the slabs: short ones first (the consumer gets text, and the decoder its estimate of the expansion, early), dealt round robin
        {
            uint32_t lo = 0, n = std::max<uint32_t>(std::min<uint32_t>(cps_, 16), cps_ / 8), s = 0;
            while (lo < n_chunks_) {
                const uint32_t hi = std::min(n_chunks_, lo + n);
                plan_.push_back(SlabPlan{lo, hi, s % nl});
                lo = hi; s++;
                n = std::min(cps_, n * 2);
            }
        }
        const double ts0 = now_s();
        DCHK(hipHostMalloc((void **)&h_chunks_, (size_t)n_chunks_ * sizeof(GzChunk) + 64, PINNED_FOR_KERNELS));
        memset(h_chunks_, 0, (size_t)n_chunks_ * sizeof(GzChunk));
        DCHK(hipHostMalloc((void **)&h_win_, GZ_WINDOW, PINNED_FOR_KERNELS));
        memset(h_win_, 0, GZ_WINDOW);
        link_ = GzLinkState(); link_.cur_bit = (uint64_t)base_byte_ * 8;
        lanes_.resize(nl);
        std::vector<GzUploader::Lane> ul(nl);
        for (uint32_t l = 0; l < nl; l++) {
            Lane &L = lanes_[l];
            L.ldev = devices[l]; L.dev = phys(devices[l]);
            DCHK(hipSetDevice(L.dev));
            const double tl0 = now_s();
            L.ds = g_streams.get(L.dev, large, err);          // (starts the maker thread if this is the set's first use; nothing here waits for a stream)
            if (!L.ds) return MF_E_HIP;
            L.want_masked_post = large && big;
            if (n_chunks_ > 4 * cps_) L.ds->ask(DeviceStreams::N_WHAT);          // a file of many slabs: every stream of the set, now -- they are made while the first slabs decode
            t_open_streams_ += now_s() - tl0;
            DCHK(L.ring.need(L.dev, ring_ + 4096, false));
            DCHK(L.d_chunks.need(L.dev, n_chunks_, false)); DCHK(L.d_window.need(L.dev, GZ_WINDOW, false));
            DCHK(L.d_acc.need(L.dev, (size_t)LIST_SLOTS * (cps_ + 1), false)); DCHK(L.d_acc_off.need(L.dev, (size_t)LIST_SLOTS * (cps_ + 1), false)); DCHK(L.d_link.need(L.dev, gz_link_scratch_bytes(cps_), false));
            DCHK(hipHostMalloc((void **)&L.h_list, (size_t)LIST_SLOTS * (cps_ + 1) * 12, PINNED_FOR_KERNELS));
            L.ev_list.assign(LIST_SLOTS, nullptr);
            ul[l].dev = L.dev; ul[l].ring = L.ring.p; ul[l].ds = L.ds;
        }
        win_dev_ = -1; win_on_host_ = true;
        // which lanes want which pieces of the file
        const size_t np = (size_ + piece_ - 1) / piece_;
        std::vector<uint64_t> want(np, 0);
        for (const SlabPlan &P : plan_) {
            const size_t a = P.lo ? base_byte_ + (size_t)P.lo * chunk_ : 0, b = std::min(size_, base_byte_ + (size_t)P.hi * chunk_ + margin_);
            for (size_t i = a / piece_; i <= (b - 1) / piece_ && i < np; i++) want[i] |= (uint64_t)1 << P.lane;
        }
        const double tu0 = now_s();
        up_.reset(new GzUploader());
        up_->start(data_, fd, size_, ring_, piece_, ul, want);
        t_open_upload_ = now_s() - tu0; t_open_ = now_s() - ts0;
        in_member_ = true;
        TRACE("gz open: %u chunks of %zu B, %zu slabs (<= %u chunks), ring %zu MiB, pieces of %zu KiB, %u slabs in flight, %u lanes", n_chunks_, chunk_, plan_.size(), cps_,
              ring_ >> 20, piece_ >> 10, max_inflight_, nl);
        return MF_OK;
    }
    // The next piece of text (possibly nothing: out.buf is null).  Nothing here waits for the link, resolve or CRC kernels of a piece: the
    // piece is handed over with an event (TextBuf::ready) that the consumer's stream waits for; what the producer does wait for is the
    // decode kernel of the front slab (it needs the chunks' descriptors) and a text buffer.
    int next(TextPiece &out, std::string &err)
    {
        struct Timed { double &acc, t0; ~Timed() { acc += now_s() - t0; } } timed{t_next_, now_s()};
        out = TextPiece();
        { const double t = now_s(); reap(false); t_reap_ += now_s() - t; }
        if (done_ || (slabs_.empty() && next_plan_ >= plan_.size())) return MF_OK;
        // decode runs ahead of the text: the front slab (waiting for its bytes if need be) and as many of the following ones as the
        // ring has room and uploaded bytes for
        const double tla = now_s();
        int rc = launch_ahead(err);
        t_launch_ += now_s() - tla;
        if (rc) return rc;
        Slab &S = *slabs_.front();
        Lane &L = lanes_[S.lane];
        DCHK(hipSetDevice(L.dev));
        rc = lane_post(L, err); if (rc) return rc;
        hipStream_t sp = L.post;
        if (!S.read_back) {
            TRACE("slab %u..%u on lane %u: waiting for decode", S.lo, S.hi, S.lane);
            const double tw0 = now_s();
            DCHK(hipEventSynchronize(S.ev));          // (the descriptors came down on the slab's own decode stream, behind its kernel)
            t_wait_decode_ += now_s() - tw0;
            if (!first_decoded_) { first_decoded_ = true; cold_mark("producer: first slab decoded"); }
            for (;;) {
                bool overflow = false;
                for (uint32_t c = S.lo; c < S.hi; c++) if (h_chunks_[c].status == GZ_OVERFLOW) overflow = true;
                if (!overflow) break;
                // text that expands more than the symbol buffers allow for (a run of identical reads, say): this slab again, with four
                // times the room -- the first half of it only, when that would be a very large buffer
                if (S.cap > chunk_ * 2048) { err = "gzip data in " + path_ + " expands more than a thousandfold: not decoded on the device"; return MF_E_FORMAT; }
                const size_t budget = (size_t)env_u64("MF_GZDEV_RETRY_BYTES", (size_t)4 << 30);
                while (S.hi - S.lo > 1 && (size_t)(S.hi - S.lo) * S.cap * 4 * 2 > budget) {
                    const uint32_t mid = S.lo + (S.hi - S.lo) / 2;
                    std::unique_ptr<Slab> B(new Slab());
                    B->lo = mid; B->hi = S.hi; B->lane = S.lane; B->cap = 0;      // (decoded when it is the front slab: its bytes are in the ring)
                    S.hi = mid;
                    slabs_.insert(slabs_.begin() + 1, std::move(B));
                    n_splits_++;
                }
                S.cap *= 4;
                TRACE("slab %u..%u overflowed: again with %zu symbols per chunk", S.lo, S.hi, S.cap);
                DCHK(S.sym.need(L.dev, (size_t)(S.hi - S.lo) * S.cap, false));
                hipStream_t sd = L.ds->pick_dec(0);
                DCHK(launch_gz_decode(L.ring.p, ring_, size_, S.limit, base_byte_, chunk_, S.lo, S.hi - S.lo, 0, (uint64_t)base_byte_ * 8, S.sym.p, S.cap, L.d_chunks.p, S.lst.p, sd));
                DCHK(hipMemcpyAsync(h_chunks_ + S.lo, L.d_chunks.p + S.lo, (S.hi - S.lo) * sizeof(GzChunk), hipMemcpyDeviceToHost, sd));
                DCHK(hipStreamSynchronize(sd));
            }
            uint32_t mx = 0;
            for (uint32_t c = S.lo; c < S.hi; c++) { const GzChunk &ch = h_chunks_[c]; if (ch.status == GZ_AT_BOUNDARY || ch.status == GZ_MEMBER_END) mx = std::max(mx, ch.n_sym); }
            max_sym_seen_ = std::max(max_sym_seen_, mx);
            {   // when its decode kernel ran, on the lane's clock (for the busy time of the decoder)
                float a_ms = 0, b_ms = 0;
                if (L.ev_base && hipEventElapsedTime(&a_ms, L.ev_base, S.ev0) == hipSuccess && hipEventElapsedTime(&b_ms, L.ev_base, S.ev1) == hipSuccess) L.spans.emplace_back((double)a_ms, (double)b_ms);
                else (void)hipGetLastError();
            }
            S.read_back = true; S.cur = S.lo;
        }
        // the chunks of this piece: as many of the slab's as make a text buffer of reasonable size
        const uint32_t a = S.cur; uint32_t b = a; uint64_t sum = 0;
        while (b < S.hi) {
            const GzChunk &ch = h_chunks_[b];
            const uint64_t n = (ch.status == GZ_AT_BOUNDARY || ch.status == GZ_MEMBER_END) ? ch.n_sym : 0;
            if (b > a && sum + n > text_piece_max_) break;
            sum += n; b++;
        }
        const bool last_piece = b == n_chunks_;
        const uint64_t T0 = link_.total;
        const double tl0 = now_s();
        TRACE("piece: chunks %u..%u, %llu symbols, text from %llu", a, b, (unsigned long long)sum, (unsigned long long)T0);
        rc = new_text(L, T0, sum + ((size_t)1 << 20), err);
        if (rc) return rc;
        t_newtext_ += now_s() - tl0;
        // link; the host steps in where the walk stops
        for (;;) {
            if (!done_ && in_member_) {
                // which chunks are accepted: a walk over the descriptors, here; the windows: kernels, on the post stream
                const uint32_t wlen_before = link_.wlen;
                gz_link_walk(h_chunks_, b, link_, acc_, acc_off_);
                TRACE("link: %zu chunks accepted, stop %u next %u cur_bit %llu total %llu linked %u", acc_.size(), link_.stop, link_.next, (unsigned long long)link_.cur_bit, (unsigned long long)link_.total, link_.linked);
                if (!acc_.empty()) {
                    uint32_t mx = 0;
                    for (uint32_t c : acc_) mx = std::max(mx, h_chunks_[c].n_sym);
                    rc = window_to(S.lane, err); if (rc) return rc;
                    uint32_t slot = 0;
                    rc = lists_up(L, slot, err); if (rc) return rc;
                    const uint32_t *da = L.d_acc.p + (size_t)slot * (cps_ + 1); const uint64_t *dao = L.d_acc_off.p + (size_t)slot * (cps_ + 1);
                    // the tails and the window on the post stream -- the next link step waits for nothing else --, the bodies behind them on post_b
                    DCHK(launch_gz_link(da, dao, (uint32_t)acc_.size(), mx, L.d_chunks.p, S.lo, S.sym.p, S.cap, L.d_window.p, wlen_before, L.d_link.p, cur_buf_->p, T0, acc_off_[0], sp));
                    rc = b_behind_a(L, err); if (rc) return rc;
                    DCHK(launch_gz_resolve(da, dao, (uint32_t)acc_.size(), mx, L.d_chunks.p, S.lo, S.sym.p, S.cap, cur_buf_->p, T0, L.post_b));
                    DCHK(hipEventRecord(L.ev_list[slot], L.post_b));
                    win_dev_ = (int)S.lane; win_on_host_ = false;
                    if (lanes_.size() > 1) { rc = window_down(err); if (rc) return rc; }          // (the next slab is linked on another device)
                }
            }
            if (done_) break;
            uint64_t to_bit = 0;
            // (behind a member's end the walk goes on with the rest of the piece's chunks, from the next member's first block)
            if (link_.stop == GZ_STOP_MEMBER_END) { rc = member_end(S, T0, err); if (rc) return rc; if (done_) break; continue; }
            if (link_.stop == GZ_STOP_GAP) to_bit = h_chunks_[link_.next].start_bit;
            else if (link_.stop == GZ_STOP_NONE) {
                if (!last_piece) break;
                to_bit = (uint64_t)size_ * 8;             // behind the last chunk: the host decodes to the end of the member
            }
            // ---- decode across the gap on the host, with the window behind the accepted data
            rc = window_down(err); if (rc) return rc;
            std::vector<uint8_t> bytes; uint64_t end_bit = 0; bool mend = false; std::string why;
            if (!inflate_gap(data_, size_, link_.cur_bit, to_bit, h_win_, link_.wlen, bytes, end_bit, mend, why)) {
                err = "gzip read error in " + path_ + ": " + why; return MF_E_FORMAT;
            }
            gap_bytes_ += bytes.size(); n_gaps_++;
            TRACE("gap: %zu bytes, ends at bit %llu (wanted %llu), member end %d", bytes.size(), (unsigned long long)end_bit, (unsigned long long)to_bit, (int)mend);
            rc = grow_text(L, T0, link_.total + bytes.size() + sum + ((size_t)1 << 20), err);
            if (rc) return rc;
            if (!bytes.empty()) { DCHK(hipMemcpyAsync(cur_buf_->p + (link_.total - T0), bytes.data(), bytes.size(), hipMemcpyHostToDevice, sp)); DCHK(hipStreamSynchronize(sp)); }
            // the window behind the gap (on the host now: it goes up again before the next link)
            if (bytes.size() >= GZ_WINDOW) { memcpy(h_win_, bytes.data() + bytes.size() - GZ_WINDOW, GZ_WINDOW); link_.wlen = GZ_WINDOW; }
            else {
                const size_t keep = std::min<size_t>(link_.wlen, GZ_WINDOW - bytes.size());
                memmove(h_win_ + GZ_WINDOW - keep - bytes.size(), h_win_ + GZ_WINDOW - keep, keep);
                memcpy(h_win_ + GZ_WINDOW - bytes.size(), bytes.data(), bytes.size());
                link_.wlen = (uint32_t)(keep + bytes.size());
            }
            win_dev_ = -1; win_on_host_ = true;
            link_.cur_bit = end_bit; link_.total += bytes.size();
            if (link_.stop == GZ_STOP_NONE && last_piece && !mend && bytes.empty()) { err = "gzip read error in " + path_ + ": truncated deflate stream"; return MF_E_FORMAT; }
            link_.stop = mend ? GZ_STOP_MEMBER_END : GZ_STOP_NONE;
            if (mend) { rc = member_end(S, T0, err); if (rc) return rc; if (done_) break; }
        }
        t_link_ += now_s() - tl0;
        // the rest of the member's CRC over this piece: launched here, taken in when it has come down (or at the member's end)
        const double tc0 = now_s();
        rc = b_behind_a(L, err); if (rc) return rc;          // (the bytes of a gap, the zeros in front of the text: whatever post has been given for this piece)
        if (link_.total > crc_done_) { rc = crc_launch(L, crc_done_, link_.total, T0, L.post_b, err); if (rc) return rc; }
        t_crc_ += now_s() - tc0;
        S.cur = b;
        const bool slab_done = S.cur == S.hi || done_;
        if (last_piece && !done_) { err = "gzip read error in " + path_ + ": unexpected end of file"; return MF_E_FORMAT; }   // the data ran out inside a member
        // the piece is text once everything queued on the post streams up to here has run: it is handed over now, with the event that says so
        DCHK(hipEventRecord(cur_buf_->ready_event(), L.post_b));
        out.buf = std::move(cur_buf_); out.T0 = T0; out.len = link_.total - T0; out.last = done_;
        out.grow = b > a ? std::max(1.0, (double)cps_ / (double)(b - a)) : 1.0;
        if (slab_done) {
            // its symbols are being resolved: the slab is kept until the post stream has passed this point
            Retired R; R.slab = std::move(slabs_.front()); slabs_.pop_front();
            DCHK(hipEventCreateWithFlags(&R.done, hipEventDisableTiming)); DCHK(hipEventRecord(R.done, L.post_b));
            R.dev = L.dev;
            retired_.push_back(std::move(R));
            // what is in front of the next slab has been linked: the ring may take new bytes there
            const uint32_t lo_next = !slabs_.empty() ? slabs_.front()->lo : (next_plan_ < plan_.size() ? plan_[next_plan_].lo : n_chunks_);
            up_->set_low_water(base_byte_ + (uint64_t)lo_next * chunk_);
        }
        return MF_OK;
    }
    bool finished() const { return done_ || (slabs_.empty() && next_plan_ >= plan_.size()); }
    uint64_t text_bytes() const { return link_.total; }
    double launch_seconds() const { return t_launch_; }
    void open_parts(double &all, double &streams, double &uploader) const { all = t_open_; streams = t_open_streams_; uploader = t_open_upload_; }
    void link_parts(double &newtext, double &post_wait) const { newtext = t_newtext_; post_wait = t_post_wait_; }
    double slot_seconds() const { return t_slot_; }
    // where the producer thread's time went: waiting for decode kernels, the link step (incl. the wait for a text buffer); the uploader's
    void other_times(double &reap, double &crc, double &all) const { reap = t_reap_; crc = t_crc_; all = t_next_; }
    void producer_times(double &wait_decode, double &link, double &up_ring, double &up_copy, double &up_read) const
    { wait_decode = t_wait_decode_; link = t_link_; up_ring = up_ ? up_->t_ring_ : 0; up_copy = up_ ? up_->t_copy_wait_ : 0; up_read = up_ ? up_->t_read_ : 0; }
    // seconds during which at least one decode kernel of this stream was running on a device, summed over the devices
    double decode_busy_seconds() const
    {
        double sum = 0;
        for (const Lane &L : lanes_) {
            std::vector<std::pair<double, double>> v = L.spans;
            std::sort(v.begin(), v.end());
            double a = 0, b = -1;
            for (auto &x : v) { if (x.first > b) { if (b > a) sum += b - a; a = x.first; b = x.second; } else if (x.second > b) b = x.second; }
            if (b > a) sum += b - a;
        }
        return sum / 1e3;
    }
    uint64_t gap_bytes() const { return gap_bytes_; }
    uint64_t gaps() const { return n_gaps_; }
    uint64_t chunks_linked() const { return link_.linked; }
    uint32_t chunks() const { return n_chunks_; }
    size_t chunk_bytes() const { return chunk_; }
    size_t ring_bytes() const { return ring_; }
    uint32_t splits() const { return n_splits_; }
private:
    static constexpr uint32_t LIST_SLOTS = 4;          // pinned staging for the accepted-chunk lists on their way up: a few link steps may be queued
    // A piece's CRC launch leaves its results in one of a few slots, taken in -- in text order -- when they have come down: the producer
    // does not wait for the post stream piece by piece (it did, for the launch before: every piece then cost the producer the whole of the
    // previous piece's link, resolve and CRC kernels, 4-5 ms a slab of configs[4], and the decode launches behind it came that much later).
    static constexpr uint32_t CRC_SLOTS = 4;
    struct CrcSlot { DevBuf<uint32_t> d; uint32_t *h = nullptr; size_t h_cap = 0; hipEvent_t ev = nullptr; uint64_t n = 0; bool out = false; };      // h: pinned
    struct Lane {
        int dev = 0, ldev = 0; DeviceStreams *ds = nullptr; hipStream_t post = nullptr, post_b = nullptr; int post_slot = -1; bool want_masked_post = false;          // post_b: see lane_post
        hipEvent_t ev_a = nullptr;          // on post, behind a link step: post_b's kernels of the same chunks wait for it
        hipEvent_t ev_base = nullptr; std::vector<std::pair<double, double>> spans;
        DevBuf<uint8_t> ring, d_window, d_link; DevBuf<GzChunk> d_chunks; DevBuf<uint32_t> d_acc; DevBuf<uint64_t> d_acc_off;
        CrcSlot crc[CRC_SLOTS]; uint32_t crc_seq = 0;
        uint8_t *h_list = nullptr; std::vector<hipEvent_t> ev_list; uint32_t list_seq = 0;      // pinned: LIST_SLOTS x {offsets, chunk numbers}
    };
    struct SlabPlan { uint32_t lo, hi, lane; };
    struct Slab {
        uint32_t lo = 0, hi = 0, lane = 0, cur = 0; DevBuf<uint16_t> sym; DevBuf<uint32_t> lst; size_t cap = 0, limit = 0;
        hipEvent_t ev = nullptr, ev0 = nullptr, ev1 = nullptr;     // lst: the lane-parallel kernel's code lists; ev0 / ev1: in front of and behind the slab's decode kernel; ev: behind the descriptors' copy to the host     // cap: symbols of room per chunk; limit: bytes of the file on the device when it was launched
        bool launched = false, read_back = false;
    };
    struct Retired { std::unique_ptr<Slab> slab; hipEvent_t done = nullptr; int dev = 0; };
    void drop_events(Slab &S) { (void)hipSetDevice(lanes_[S.lane].dev); for (hipEvent_t *e : {&S.ev, &S.ev0, &S.ev1}) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; } }
    // slabs whose symbols the post stream is done with give their buffers back (all: wait for them)
    void reap(bool all)
    {
        while (!retired_.empty()) {
            Retired &R = retired_.front();
            (void)hipSetDevice(R.dev);
            if (all) (void)hipEventSynchronize(R.done);
            else if (hipEventQuery(R.done) != hipSuccess) { (void)hipGetLastError(); break; }
            (void)hipEventDestroy(R.done);
            drop_events(*R.slab);
            retired_.pop_front();
        }
    }
    // symbols of room per chunk before the file has shown its expansion: 4.5 : 1 and 64 Ki of slack for the block a chunk decodes past its range (the
    // first slabs are short: one that overflows -- text that expands more -- is decoded again with four times the room, cheaply, and the rule below takes over)
    size_t sym_cap_first() const { return (size_t)((double)chunk_ * expand_) + (expand_fixed_ ? 262144 : 65536); }
    size_t sym_cap_now() const
    {
        if (expand_fixed_ || !max_sym_seen_) return sym_cap_first();
        // what the largest chunk so far needed, and a quarter; a chunk reads one block past its range (and up to a chunk's worth of
        // stored blocks), which the maximum has seen as well
        return (size_t)max_sym_seen_ + max_sym_seen_ / 4 + 65536;
    }
    // the lane's post stream, taken from the device's set when the lane first links (the set's maker may still be at it)
    int lane_post(Lane &L, std::string &err)
    {
        if (L.post) return MF_OK;
        const double t0 = now_s();
        L.post = L.ds->take_post(L.want_masked_post, &L.post_slot);
        t_post_wait_ += now_s() - t0;
        if (!L.post) { err = "hipStreamCreate failed"; return MF_E_HIP; }
        // A link step waits for the one before through the window, and for nothing else; the bodies of its chunks and the CRC of its text are
        // three quarters of a piece's post work (2.0 + 1.0 of 3.6 ms a slab of configs[4], profiles/r05/devingest_kernel_stats.txt) and nothing
        // of the next piece waits for them: on an input of many slabs they go to a stream of their own, behind the link step (b_behind_a).
        // (one stream for it all paced the whole pipeline at the sum: profiles/r05/g_configs4_timing_crc_ring.txt)
        const char *two = getenv("MF_GZDEV_RESOLVE_STREAM");
        L.post_b = (two ? two[0] == '1' : (L.want_masked_post || n_chunks_ > 4 * cps_)) ? L.ds->take_post_b(L.post_slot) : nullptr;
        if (!L.post_b) L.post_b = L.post;
        DCHK(hipEventCreate(&L.ev_base)); DCHK(hipEventRecord(L.ev_base, L.post));
        return MF_OK;
    }
    int launch_ahead(std::string &err)
    {
        // slabs that were split off an overflowing one wait at the front without a launch
        while (slabs_.size() < max_inflight_ && next_plan_ < plan_.size()) {
            std::unique_ptr<Slab> S(new Slab());
            const SlabPlan &P = plan_[next_plan_++];
            S->lo = P.lo; S->hi = P.hi; S->lane = P.lane;
            slabs_.push_back(std::move(S));
        }
        for (size_t i = 0; i < slabs_.size(); i++) {
            Slab &S = *slabs_[i];
            if (S.launched) continue;
            Lane &L = lanes_[S.lane];
            // the chunks read past their own range up to the end of a block, and the reader's ring a little further
            const size_t upto = std::min(size_, base_byte_ + (size_t)S.hi * chunk_ + margin_);
            if (i > 0 && !up_->issued(upto)) break;
            // (decode kernels on one stream run one after the other, each waiting for the last straggler of the one before: while the
            // device's decode streams are still being made -- a process's first large file -- no more than two slabs are queued per stream)
            if (i >= 2 * (size_t)std::max<uint32_t>(1, std::min(L.ds->n_dec.load(), dec_limit_))) break;
            DCHK(hipSetDevice(L.dev));
            if (!S.cap) S.cap = sym_cap_now();
            DCHK(S.sym.need(L.dev, (size_t)(S.hi - S.lo) * S.cap, false));
            if (!gz_decode_serial()) DCHK(S.lst.need(L.dev, gz_decode_scratch_bytes(S.hi - S.lo) / 4, false));
            if (!S.ev) DCHK(hipEventCreateWithFlags(&S.ev, hipEventDisableTiming));
            if (!S.ev0) DCHK(hipEventCreate(&S.ev0));
            if (!S.ev1) DCHK(hipEventCreate(&S.ev1));
            hipStream_t st = L.ds->pick_dec(launch_seq_++, dec_limit_);
            if (!st) { err = "hipStreamCreate failed"; return MF_E_HIP; }
            if (!up_->wait_for(S.lane, st, upto)) {
                const int why = up_->failure();
                err = why == MF_E_NOMEM ? "hipHostMalloc failed: no pinned memory for the staging buffers of " + path_ : "upload of " + path_ + " failed";
                return why ? why : MF_E_IO;
            }
            S.limit = upto;
            DCHK(hipEventRecord(S.ev0, st));
            DCHK(launch_gz_decode(L.ring.p, ring_, size_, S.limit, base_byte_, chunk_, S.lo, S.hi - S.lo, 0, (uint64_t)base_byte_ * 8, S.sym.p, S.cap, L.d_chunks.p, S.lst.p, st));
            DCHK(hipEventRecord(S.ev1, st));
            DCHK(launch_bytes_to_host(h_chunks_ + S.lo, L.d_chunks.p + S.lo, (S.hi - S.lo) * sizeof(GzChunk), st));
            DCHK(hipEventRecord(S.ev, st));
            if (!first_launched_) { first_launched_ = true; cold_mark("producer: first decode kernel launched"); }
            S.launched = true;
        }
        return MF_OK;
    }
    // the accepted chunks of this link step -> one of the lane's device lists (through a slot of pinned staging: a few steps may be queued; a slot
    // is free again when the resolve kernel that read it has run)
    int lists_up(Lane &L, uint32_t &slot, std::string &err)
    {
        slot = L.list_seq++ % LIST_SLOTS;
        const uint32_t n = (uint32_t)acc_.size();
        if (L.ev_list[slot]) DCHK(hipEventSynchronize(L.ev_list[slot])); else DCHK(hipEventCreateWithFlags(&L.ev_list[slot], hipEventDisableTiming));
        uint8_t *h = L.h_list + (size_t)slot * (cps_ + 1) * 12;
        memcpy(h, acc_off_.data(), (size_t)n * 8); memcpy(h + (size_t)(cps_ + 1) * 8, acc_.data(), (size_t)n * 4);
        DCHK(launch_bytes_from_host(L.d_acc_off.p + (size_t)slot * (cps_ + 1), h, (size_t)n * 8, L.post));          // (not the copy engine: mf_ingest.h)
        DCHK(launch_bytes_from_host(L.d_acc.p + (size_t)slot * (cps_ + 1), h + (size_t)(cps_ + 1) * 8, (size_t)n * 4, L.post));
        return MF_OK;
    }
    // what post has been given up to here, post_b runs behind
    int b_behind_a(Lane &L, std::string &err)
    {
        if (L.post_b == L.post) return MF_OK;
        if (!L.ev_a) DCHK(hipEventCreateWithFlags(&L.ev_a, hipEventDisableTiming));
        DCHK(hipEventRecord(L.ev_a, L.post));
        DCHK(hipStreamWaitEvent(L.post_b, L.ev_a, 0));
        return MF_OK;
    }
    // the window is on lane l's device (it travels through the host between lanes, and after the host has decoded across a gap)
    int window_to(uint32_t l, std::string &err)
    {
        if (win_dev_ == (int)l) return MF_OK;
        int rc = window_down(err); if (rc) return rc;
        Lane &L = lanes_[l];
        DCHK(hipSetDevice(L.dev));
        DCHK(launch_bytes_from_host(L.d_window.p, h_win_, GZ_WINDOW, L.post));
        DCHK(hipStreamSynchronize(L.post));          // (h_win_ is the host's to change again)
        win_dev_ = (int)l;
        return MF_OK;
    }
    // ... and on the host
    int window_down(std::string &err)
    {
        if (win_on_host_) return MF_OK;
        Lane &W = lanes_[(size_t)win_dev_];
        int cur = -1; (void)hipGetDevice(&cur);
        DCHK(hipSetDevice(W.dev));
        DCHK(hipMemcpyAsync(h_win_, W.d_window.p, GZ_WINDOW, hipMemcpyDeviceToHost, W.post));
        DCHK(hipStreamSynchronize(W.post));
        if (cur >= 0 && cur != W.dev) DCHK(hipSetDevice(cur));
        win_on_host_ = true;
        return MF_OK;
    }
    // a fresh buffer for the piece that begins at text offset T0
    int new_text(Lane &L, uint64_t T0, size_t text_bytes, std::string &err)
    {
        (void)T0;
        const double ts = now_s();
        if (!slots_->take()) { err = "stopped"; return MF_E_IO; }
        t_slot_ += now_s() - ts;
        DCHK(TextBuf::make(cur_buf_, L.dev, L.ldev, pad_, text_bytes, slots_));
        // (a damaged stream may point a full window back from the first byte of the text: zeros there, ahead of the link step on its stream)
        DCHK(hipMemsetAsync(cur_buf_->p - TEXT_FRONT, 0, TEXT_FRONT, L.post));
        return MF_OK;
    }
    // ... holds at least `need_abs - T0` bytes (what is in it moves along)
    int grow_text(Lane &L, uint64_t T0, uint64_t need_abs, std::string &err)
    {
        if (need_abs - T0 <= cur_buf_->cap) return MF_OK;
        std::unique_ptr<TextBuf> nb;
        DCHK(TextBuf::make(nb, L.dev, L.ldev, pad_, (size_t)((need_abs - T0) + (need_abs - T0) / 2), nullptr));
        const uint64_t have = link_.total - T0;
        if (L.post_b != L.post) DCHK(hipStreamSynchronize(L.post_b));          // (bodies on their way into the old buffer)
        DCHK(hipMemcpyAsync(nb->raw, cur_buf_->raw, cur_buf_->pad + have, hipMemcpyDeviceToDevice, L.post)); DCHK(hipStreamSynchronize(L.post));
        nb->slots = cur_buf_->slots; cur_buf_->slots = nullptr;          // (the slot moves to the new buffer)
        cur_buf_ = std::move(nb);
        return MF_OK;
    }
    // gzip header at byte pos -> pos = first byte of deflate data
    bool member_header(size_t &pos, std::string &err)
    {
        const uint8_t *d = data_;
        if (size_ - pos < 18 || d[pos] != 0x1f || d[pos + 1] != 0x8b) { err = "gzip read error in " + path_ + ": not in gzip format"; return false; }
        if (d[pos + 2] != 8) { err = "gzip read error in " + path_ + ": unknown compression method"; return false; }
        const unsigned flg = d[pos + 3];
        size_t p = pos + 10;
        if (flg & 4) { if (p + 2 > size_) goto trunc; { const size_t xlen = d[p] | ((size_t)d[p + 1] << 8); p += 2 + xlen; } if (p > size_) goto trunc; }
        for (unsigned bit = 8; bit <= 16; bit <<= 1)
            if (flg & bit) { const void *z = p < size_ ? memchr(d + p, 0, size_ - p) : nullptr; if (!z) goto trunc; p = (size_t)((const uint8_t *)z - d) + 1; }
        if (flg & 2) p += 2;
        if (p + 8 > size_) goto trunc;
        pos = p;
        return true;
    trunc:
        err = "gzip read error in " + path_ + ": truncated gzip header";
        return false;
    }
    // the accepted data ends behind the final block of a member: check the trailer, look for another member
    int member_end(Slab &S, uint64_t T0, std::string &err)
    {
        Lane &L = lanes_[S.lane];
        hipStream_t sp = L.post;
        const size_t pos = (size_t)((link_.cur_bit + 7) >> 3);
        if (pos + 8 > size_) { err = "gzip read error in " + path_ + ": truncated gzip trailer"; return MF_E_FORMAT; }
        uint32_t want_crc, want_len; memcpy(&want_crc, data_ + pos, 4); memcpy(&want_len, data_ + pos + 4, 4);
        // CRC of the member's text up to here (everything of it is queued on the post stream: link, resolve, the bytes of a gap)
        { const int rc = b_behind_a(L, err); if (rc) return rc; }
        if (link_.total > crc_done_) { const int rc = crc_launch(L, crc_done_, link_.total, T0, L.post_b, err); if (rc) return rc; }
        { const int rc = crc_take(L, nullptr, true, err); if (rc) return rc; }
        DCHK(hipStreamSynchronize(sp));
        if (L.post_b != sp) DCHK(hipStreamSynchronize(L.post_b));
        TRACE("member end: crc %08x want %08x", crc_, want_crc);
        if (crc_ != want_crc) { err = "gzip read error in " + path_ + ": incorrect data check"; return MF_E_FORMAT; }
        if ((uint32_t)(link_.total - member_T0_) != want_len) { err = "gzip read error in " + path_ + ": incorrect length check"; return MF_E_FORMAT; }
        crc_ = 0; member_T0_ = link_.total;
        size_t p = pos + 8;
        if (p >= size_ || size_ - p < 2 || data_[p] != 0x1f || data_[p + 1] != 0x8b) { done_ = true; in_member_ = false; return MF_OK; }   // trailing bytes that are no member: ignored
        if (!member_header(p, err)) return MF_E_FORMAT;
        // (a new member begins with an empty window: whatever holds the old one is out of date)
        link_.cur_bit = (uint64_t)p * 8; link_.wlen = 0; link_.stop = GZ_STOP_NONE;
        memset(h_win_, 0, GZ_WINDOW); win_dev_ = -1; win_on_host_ = true;
        return MF_OK;
    }
    // running CRC of the member over the text [from, to) of the current piece: the kernel and the copy of its piece CRCs (crc_launch),
    // the combination on the host (crc_take)
    int crc_launch(Lane &L, uint64_t from, uint64_t to, uint64_t T0, hipStream_t st, std::string &err)
    {
        CrcSlot &C = L.crc[L.crc_seq++ % CRC_SLOTS];
        // the slot's last launch (four pieces ago on this lane) is taken in first if it has not been, and whatever else has come down
        { const int rc = crc_take(L, &C, false, err); if (rc) return rc; }
        const uint64_t n = to - from;
        const size_t np = (size_t)((n + GZ_CRC_PIECE - 1) / GZ_CRC_PIECE);
        DCHK(C.d.need(L.dev, np));
        if (np > C.h_cap) { if (C.h) (void)hipHostFree(C.h); C.h = nullptr; C.h_cap = 0; DCHK(hipHostMalloc((void **)&C.h, (np + np / 2 + 64) * 4, hipHostMallocDefault)); C.h_cap = np + np / 2 + 64; }
        if (!C.ev) DCHK(hipEventCreateWithFlags(&C.ev, hipEventDisableTiming));
        DCHK(launch_gz_crc(cur_buf_->p + (from - T0), n, C.d.p, st));
        DCHK(launch_bytes_to_host(C.h, C.d.p, np * 4, st));
        DCHK(hipEventRecord(C.ev, st));
        C.n = n; C.out = true; crc_done_ = to;
        crc_q_.emplace_back((uint32_t)(&L - lanes_.data()), (uint32_t)(&C - L.crc));
        return MF_OK;
    }
    // CRC launches taken in, oldest first (a member's CRC is combined in text order, whichever lane a piece was on): all of them (waiting), or
    // up to and including slot `until` if that is still out (waiting), and then those that have come down already.  `cur`: the lane whose device is current.
    int crc_take(Lane &cur, const CrcSlot *until, bool all, std::string &err)
    {
        int dev = cur.dev;
        while (!crc_q_.empty()) {
            Lane &O = lanes_[crc_q_.front().first]; CrcSlot &C = O.crc[crc_q_.front().second];
            if (O.dev != dev) { DCHK(hipSetDevice(O.dev)); dev = O.dev; }
            if (all || (until && until->out)) DCHK(hipEventSynchronize(C.ev));
            else if (hipEventQuery(C.ev) != hipSuccess) { (void)hipGetLastError(); break; }
            crc_ = gz_crc_combine(crc_, gz_crc_finish(C.h, C.n), C.n);
            C.n = 0; C.out = false;
            crc_q_.pop_front();
        }
        if (dev != cur.dev) DCHK(hipSetDevice(cur.dev));
        return MF_OK;
    }

    const uint8_t *data_ = nullptr; size_t size_ = 0; std::string path_; Slots *slots_ = nullptr; size_t pad_ = TEXT_FRONT; std::atomic<bool> *stop_ = nullptr;
    size_t chunk_ = 0, base_byte_ = 0, margin_ = 0, ring_ = 0, piece_ = 0; double expand_ = 6; bool expand_fixed_ = false; uint32_t max_sym_seen_ = 0;
    uint64_t text_piece_max_ = 0;
    uint32_t cps_ = 0, n_chunks_ = 0, max_inflight_ = 1, launch_seq_ = 0, n_splits_ = 0, dec_limit_ = GZ_NSTREAM;
    std::vector<Lane> lanes_; std::vector<SlabPlan> plan_; size_t next_plan_ = 0;
    std::deque<std::unique_ptr<Slab>> slabs_;          // launched or waiting, in stream order; front = being linked
    std::deque<Retired> retired_;                      // linked, their symbols on their way to becoming text
    std::unique_ptr<GzUploader> up_;
    GzLinkState link_; uint8_t *h_win_ = nullptr; int win_dev_ = -1; bool win_on_host_ = true;      // h_win_: pinned, the window when the host has it; win_dev_: the lane whose d_window is current (-1: none)
    std::vector<uint32_t> acc_; std::vector<uint64_t> acc_off_;
    GzChunk *h_chunks_ = nullptr;                      // pinned: every chunk's descriptor, copied down behind its slab's decode kernel
    std::unique_ptr<TextBuf> cur_buf_;
    double t_open_ = 0, t_open_streams_ = 0, t_open_upload_ = 0;
    bool in_member_ = false, done_ = false, first_launched_ = false, first_decoded_ = false;
    uint32_t crc_ = 0; uint64_t crc_done_ = 0, member_T0_ = 0, gap_bytes_ = 0, n_gaps_ = 0; std::deque<std::pair<uint32_t, uint32_t>> crc_q_;      // crc_q_: (lane, slot) of the CRC launches not taken in yet, in text order
    double t_wait_decode_ = 0, t_link_ = 0, t_launch_ = 0, t_newtext_ = 0, t_post_wait_ = 0, t_slot_ = 0, t_reap_ = 0, t_crc_ = 0, t_next_ = 0;
};

// ---- survivors on their way to the output file (one writer thread per mate; pieces arrive in order)
class Writer {
public:
    bool open(const char *path) { ok_ = of_.open(path); if (ok_) th_ = std::thread([this] { run(); }); return ok_; }
    void push(std::vector<char> &&b) { { std::lock_guard<std::mutex> lk(mu_); q_.push_back(std::move(b)); } cv_.notify_one(); }
    bool close()
    {
        if (th_.joinable()) { { std::lock_guard<std::mutex> lk(mu_); fin_ = true; } cv_.notify_one(); th_.join(); }
        return of_.close() && ok_;
    }
    ~Writer() { if (th_.joinable()) { { std::lock_guard<std::mutex> lk(mu_); fin_ = true; } cv_.notify_one(); th_.join(); } }
private:
    void run()
    {
        for (;;) {
            std::vector<char> b;
            { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [&] { return fin_ || !q_.empty(); }); if (q_.empty()) return; b = std::move(q_.front()); q_.pop_front(); }
            if (ok_ && !b.empty() && !of_.write(b.data(), b.size())) ok_ = false;
        }
    }
    OutFile of_; bool ok_ = false, fin_ = false;
    std::thread th_; std::mutex mu_; std::condition_variable cv_; std::deque<std::vector<char>> q_;
};

// bits [r0, r0 + n) of a bitmap -> out (bit 0 = bit r0); out has (n + 31) / 32 words
void extract_bits(const std::vector<uint32_t> &v, uint64_t r0, uint64_t n, uint32_t *out)
{
    const uint64_t nw = (n + 31) / 32, w0 = r0 >> 5; const uint32_t sh = (uint32_t)(r0 & 31);
    for (uint64_t j = 0; j < nw; j++) {
        const uint32_t a = w0 + j < v.size() ? v[w0 + j] : 0, b = w0 + j + 1 < v.size() ? v[w0 + j + 1] : 0;
        out[j] = sh ? (a >> sh) | (b << (32 - sh)) : a;
    }
    if (n & 31) out[nw - 1] &= (1u << (n & 31)) - 1;
}
// the other way: n bits of src (bit 0 first) become bits [r0, r0 + n) of v
void append_bits(std::vector<uint32_t> &v, uint64_t r0, uint64_t n, const uint32_t *src)
{
    if (!n) return;
    { const size_t need = (size_t)((r0 + n + 31) / 32 + 1); if (v.size() < need) v.resize(need, 0); }      // (pieces finish out of order: never shrink)
    const uint64_t nw = (n + 31) / 32, w0 = r0 >> 5; const uint32_t sh = (uint32_t)(r0 & 31);
    for (uint64_t j = 0; j < nw; j++) {
        uint32_t x = src[j];
        if (j == nw - 1 && (n & 31)) x &= (1u << (n & 31)) - 1;
        v[w0 + j] |= x << sh;
        if (sh) v[w0 + j + 1] |= x >> (32 - sh);
    }
}

// records of one piece of text, cut where they lie; waits (with its text) until the other mate's pass bits cover it
struct Batch {
    std::unique_ptr<TextBuf> buf;
    const uint8_t *text = nullptr;       // the first record's header (the piece's text less the carry in front of it)
    DevBuf<uint64_t> line_start;         // offsets from `text`
    uint64_t n_rec = 0, rec_base = 0, n_text = 0, n_lines = 0;
    int ldev = 0;
    bool filtered = false;               // its pass bits are in the mate's bitmap
    // the quality filter's job (QualState below): what one pass over the records found, kept with the batch until its turn to be decided
    DevBuf<uint32_t> q_bad, q_sl, q_ql, q_olen; DevBuf<uint8_t> q_fl; DevBuf<uint64_t> q_hash;
    uint64_t q_done = 0;                                  // (Ingest::mu) records of it that have been decided
};
// records [r0, r0 + n) of a batch, decided: their text goes to bytes [out_at, out_at + bytes) of the mate's output file
struct QPart { uint64_t r0 = 0, n = 0, out_at = 0, bytes = 0; };

// what a consumer thread keeps per device: scratch buffers and the refillable read set (its own context of the device: own streams)
struct DevScratch {
    int ldev = 0, dev = 0, lane = 0; DevCtx *ctx = nullptr;
    mf_reads *reads = nullptr;
    DevBuf<uint32_t> tile_cnt, seq_len, inv_cnt, out_len, minmax, mask; DevBuf<uint64_t> tile_base, scan_tmp, inv_base, out_off, offsets_tmp; DevBuf<uint8_t> d_out;
    // small results the host waits for (counts that size the next buffers), in pinned memory: a copy to pageable memory is a
    // synchronisation of its own.  [0] newlines [1] last byte [2] used [3] bases [4] min/max length [5] invalid bases [6] output bytes;
    // [7], [8]: values on their way TO the device (the virtual end of an unterminated last line, the start value of min/max)
    uint64_t *h_small = nullptr;
    uint32_t *h_bits = nullptr; size_t h_bits_cap = 0;       // pinned: the pass bits of a piece on their way to the host, the keep mask on its way back
    uint8_t *h_out = nullptr; size_t h_out_cap = 0;          // pinned: survivors on their way to the writer
    // the quality filter's job: the other mate's scan results and keep flags on their way up, keep flags on their way down (pinned), per-record scratch
    uint8_t *h_stage = nullptr; size_t h_stage_cap = 0;
    DevBuf<uint32_t> q_bad2; DevBuf<uint8_t> q_fl2, q_alive, q_dup, q_keep;
    hipError_t stage(size_t bytes)
    {
        if (bytes <= h_stage_cap) return hipSuccess;
        if (h_stage) (void)hipHostFree(h_stage);
        h_stage = nullptr; h_stage_cap = 0;
        hipError_t e = hipHostMalloc((void **)&h_stage, bytes + bytes / 2 + 65536, hipHostMallocDefault);
        if (e == hipSuccess) h_stage_cap = bytes + bytes / 2 + 65536;
        return e;
    }
    ~DevScratch()
    {
        reads_release(reads); (void)hipSetDevice(dev);
        if (h_small) (void)hipHostFree(h_small); if (h_bits) (void)hipHostFree(h_bits); if (h_out) (void)hipHostFree(h_out); if (h_stage) (void)hipHostFree(h_stage);
    }
};

// A consumer's scratch -- small pinned buffers, device buffers, the refillable read set with everything the filter hangs on it -- is kept
// from call to call per (device, consumer): making it anew costs a call a few milliseconds at the start and a hipFree per buffer of the
// read set (each waits for the device to go idle) at the end.  MF_KEEP_BUFFERS=0 releases it with the call.
class ScratchCache {
public:
    std::unique_ptr<DevScratch> take(int ldev, int lane)
    {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = kept_.find(std::make_pair(ldev, lane));
        if (it == kept_.end()) return nullptr;
        std::unique_ptr<DevScratch> p = std::move(it->second);
        kept_.erase(it);
        return p;
    }
    void give(std::unique_ptr<DevScratch> p)
    {
        const char *kb = getenv("MF_KEEP_BUFFERS");
        if (!p || (kb && kb[0] == '0')) return;
        std::lock_guard<std::mutex> lk(mu_);
        kept_[std::make_pair(p->ldev, p->lane)] = std::move(p);
    }
    void clear() { std::map<std::pair<int, int>, std::unique_ptr<DevScratch>> gone; { std::lock_guard<std::mutex> lk(mu_); gone.swap(kept_); } }      // (their device buffers go back to the pool: clear the pool after this)
private:
    std::mutex mu_; std::map<std::pair<int, int>, std::unique_ptr<DevScratch>> kept_;
};
ScratchCache &g_scratch = *new ScratchCache();          // (never destroyed: its entries would call into HIP while the process is being torn down)

// ---- the quality filter's job on this path (the reference's filter_v2: filter/filter_bin/src/main.rs:188-323)
struct QualState {
    QualParams P; bool pe = false; uint64_t cap = ~0ull;          // cap: the longest a cut string gets (end - start), ~0: no end
    SegArray<uint32_t> bad2; SegArray<uint8_t> fl2, keep;         // mate 2's scan results, mate 1's decisions: per record of the file, on the host
    uint64_t panic_rec[2] = {~0ull, ~0ull};                       // (Ingest::mu) the first record at which the reference would panic, per mate, among the scanned pieces
    // decisions are taken a piece of mate 1 at a time, in order (Ingest::emit_mu):
    uint64_t budget = 0, kept = 0, out_pos[2] = {0, 0};
    uint64_t decided = 0; bool decided_final = false, panicked = false;      // (Ingest::mu) records [0, decided) have their keep flags; final: no more will be decided
    int in_flight = 0;                                            // (Ingest::mu) pieces being gathered and written
    // the de-duplication set (keys, smallest file index per key; mf_kernels.hip): on the device for the whole file
    DevBuf<unsigned long long> dd_keys, dd_first, dd_small; uint64_t dd_slots = 0, dd_n = 0;
    OutChunks chunks; QSink sink[2];                              // (the pool first: the sinks' threads give their last chunks back to it)
    double t_scan = 0, t_decide = 0, t_gather = 0, t_chunk = 0;   // (Ingest::mu) summed over the consumers; t_chunk: waiting for a free chunk = for the writers
};

struct Mate {
    std::string path; Mapped map; bool gz = false;
    std::unique_ptr<GzStream> gzs; Slots slots;
    // producer: text pieces in order
    std::thread prod; std::mutex mu; std::condition_variable cv; std::deque<TextPiece> ready; int prod_rc = MF_OK; std::string prod_err; bool prod_done = false;
    std::atomic<bool> stop{false};
    // consumers (under Ingest::mu): pieces are taken in order; their line index is cut in that order too (the carry links them),
    // packing and filtering of several pieces run side by side
    uint64_t taken = 0, a_turn = 0;      // pieces handed to a consumer; the piece whose line index may be cut now
    bool eof = false;                    // the last piece has been taken
    uint64_t rec_indexed = 0;            // records of the pieces indexed so far (the next piece's first record)
    uint64_t rec_filtered = 0;           // ... of the leading pieces whose pass bits are in `bits`
    uint8_t *h_carry = nullptr; size_t h_carry_cap = 0, carry = 0;      // pinned: the head of the record the last piece left unfinished
    std::deque<std::shared_ptr<Batch>> batches;          // indexed, in order; leave when written
    std::vector<uint32_t> bits;                          // pass bits of the whole file so far, one per record
    Writer out;
    ~Mate()
    {
        TRACE("~Mate");
        stop = true; slots.wake();
        batches.clear(); ready.clear();                   // (text buffers give their slots back: a producer waiting for one wakes up)
        if (prod.joinable()) prod.join();
        ready.clear();
        gzs.reset();
        if (h_carry) (void)hipHostFree(h_carry);
    }
};

struct Ingest {
    mf_kmerset *ks = nullptr; uint32_t threshold = 1; bool pair_both = false; std::vector<int> devices;
    QualState *qual = nullptr;          // set: the job is the quality filter (one device), not the bait filter
    Mate m[2]; int nm = 1;
    uint64_t kept = 0, total = 0;
    std::atomic<bool> first_indexed_{false}, first_filtered_{false};
    std::atomic<bool> wrote_any{false};          // a byte of the output has been handed to a writer: the call can no longer be given to the host pipeline
    size_t mem_used_max = 0;           // device memory in use (everything on the device, this path's buffers and the rest), the largest seen after a piece
    size_t carry_room = (size_t)1 << 20;
    bool timing = false; double t_wait = 0, t_index = 0, t_pack = 0, t_filter = 0, t_emit = 0;      // summed over the consumer threads
    // consumers
    struct Worker {
        int id = 0; std::map<int, std::unique_ptr<DevScratch>> scratch; std::thread th;
        ~Worker() { for (auto &kv : scratch) g_scratch.give(std::move(kv.second)); }
    };
    std::vector<std::unique_ptr<Worker>> workers;
    std::mutex mu; std::condition_variable cv;          // the state the consumers share (turns, record counts, batches, bitmaps, timing sums)
    std::mutex emit_mu;                                 // one consumer at a time writes survivors (batches leave in order)
    bool failed = false; int fail_rc = MF_OK; std::string fail_err;

    std::mutex mu_all; std::condition_variable cv_all;          // any producer has something new
    // The producers use this object's mutexes and condition variables to their last line, and on a failed run they are still running when
    // the call unwinds: they are stopped and joined before any member goes (the members' own order would destroy cv_all, declared behind
    // m[], before ~Mate joins its producer -- a notify on a destroyed condition variable; found under ThreadSanitizer by tests/native/ingest_check.cpp)
    ~Ingest()
    {
        for (auto &M : m) { M.stop = true; M.slots.wake(); }
        for (auto &M : m) {
            { std::lock_guard<std::mutex> lk(mu); M.batches.clear(); }
            { std::lock_guard<std::mutex> lk(M.mu); M.ready.clear(); }          // (text buffers give their slots back: a producer waiting for one wakes up and sees stop)
            if (M.prod.joinable()) M.prod.join();
        }
    }
    double t_begin = 0, t_first_piece = 0, t_last_piece = 0, t_consumed = 0;      // when the first / last piece of text was handed over, when the last consumer was done (seconds into the call)
    void publish(Mate &M, TextPiece &&t)
    {
        { std::lock_guard<std::mutex> lk(M.mu); M.ready.push_back(std::move(t)); }
        { std::lock_guard<std::mutex> lk(mu_all); const double now = now_s() - t_begin; if (t_first_piece == 0) { t_first_piece = now; cold_mark("device ingest: first piece of text handed over"); } t_last_piece = now; }          // (two producers)
        M.cv.notify_all(); cv_all.notify_all();
    }

    void producer(Mate &M)
    {
        std::string err; int rc = MF_OK;
        if (M.gz) {
            for (;;) {
                TextPiece t;
                rc = M.gzs->next(t, err);
                if (rc || M.stop) break;
                if (!t.buf) { if (M.gzs->finished()) break; continue; }
                const bool last = t.last;
                publish(M, std::move(t));
                if (last) break;
            }
        } else rc = plain_producer(M, err);
        if (M.stop && rc) { rc = MF_OK; err.clear(); }                 // (told to stop: not a failure of its own)
        TRACE("producer done rc %d", rc);
        { std::lock_guard<std::mutex> lk(M.mu); M.prod_rc = rc; M.prod_err = err; M.prod_done = true; }
        M.cv.notify_all(); cv_all.notify_all();
    }

    // a plain file is its own text: slabs of it go straight into text buffers, dealt to the devices round robin.  A file of up to 512 MiB is
    // read by the copy engine where the page cache holds it (PinnedMap); a larger one goes through three pinned staging buffers, the stager's
    // own threads reading the next while the copies of the two before are in flight.  Either way a slab is handed over the moment its copies
    // have been ISSUED -- the consumer's stream waits for them (TextBuf::ready), the producer does not.
    int plain_producer(Mate &M, std::string &err)
    {
        const uint64_t slab = std::max<uint64_t>(env_u64("MF_INGEST_SLAB_BYTES", (uint64_t)256 << 20), 64);
        const size_t piece = (size_t)std::min<uint64_t>((uint64_t)32 << 20, std::max<uint64_t>(slab, 4096));
        constexpr int NBUF = 3;
        DCHK(hipSetDevice(phys(devices[0])));
        PinnedMap reg(M.map.p, M.map.n);
        Stager stg; bool staged = false;
        struct PerDev { DeviceStreams *ds = nullptr; hipStream_t st = nullptr; hipEvent_t ev[NBUF] = {}; };
        std::vector<PerDev> pd(devices.size());
        // (declared behind `reg`: runs first -- the copies have run when the windows are unregistered)
        struct Cleanup { std::vector<PerDev> &pd; const std::vector<int> &devs; ~Cleanup() { for (size_t i = 0; i < pd.size(); i++) { (void)hipSetDevice(phys(devs[i])); if (pd[i].st) (void)hipStreamSynchronize(pd[i].st); for (auto &e : pd[i].ev) if (e) (void)hipEventDestroy(e); } } } cleanup{pd, devices};
        uint64_t n_piece = 0; int used_by[NBUF]; for (auto &u : used_by) u = -1;
        uint64_t s = 0; double t_slot = 0; const double t_begin = now_s();
        // (slabs grow from 32 MiB at the front of the file -- the consumers start on the first after 0.6 ms of copying, not 4.5 -- and shrink
        // again towards its end: what is left when the last copy has run is one consumer's work on a small piece)
        const uint64_t small_slab = std::min<uint64_t>(slab, (uint64_t)32 << 20);
        for (uint64_t T0 = 0; T0 < M.map.n && !M.stop; s++) {
            const uint64_t left = M.map.n - T0;
            uint64_t want = std::min<uint64_t>(slab, small_slab << std::min<uint64_t>(s, 8));
            if (left < 3 * want) want = std::max<uint64_t>(small_slab, left / 3);
            if (left < want + small_slab / 2) want = left;
            const uint64_t T1 = T0 + want;
            const size_t li = (size_t)(s % devices.size());
            const int ldev = devices[li], dev = phys(ldev);
            DCHK(hipSetDevice(dev));
            PerDev &P = pd[li];
            if (!P.st) {
                P.ds = g_streams.get(dev, false, err); if (!P.ds) return MF_E_HIP;
                P.st = P.ds->copy_stream(); if (!P.st) { err = "hipStreamCreate failed"; return MF_E_HIP; }
                for (auto &e : P.ev) DCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            }
            const double ts = now_s();
            if (!M.slots.take()) break;
            t_slot += now_s() - ts;
            TextPiece t;
            DCHK(TextBuf::make(t.buf, dev, ldev, TEXT_FRONT + carry_room, (size_t)(T1 - T0), &M.slots));
            if (reg.ensure()) {
                DCHK(hipMemcpyAsync(t.buf->p, M.map.p + T0, (size_t)(T1 - T0), hipMemcpyHostToDevice, P.st));
                if (!reg.after_copy(dev, P.st)) { err = "hipEventRecord failed"; return MF_E_HIP; }
            } else {
                if (!staged) { DCHK(stg.init(piece, M.map.fd, NBUF)); staged = true; }
                for (uint64_t off = T0; off < T1; off += piece, n_piece++) {
                    const int b = (int)(n_piece % NBUF);
                    if (used_by[b] >= 0) { const size_t lj = (size_t)used_by[b]; DCHK(hipSetDevice(phys(devices[lj]))); DCHK(hipEventSynchronize(pd[lj].ev[b])); DCHK(hipSetDevice(dev)); }
                    const size_t len = (size_t)std::min<uint64_t>(piece, T1 - off);
                    if (!stg.read(b, (size_t)off, len)) { err = "read error on " + M.path; return MF_E_IO; }
                    DCHK(hipMemcpyAsync(t.buf->p + (off - T0), stg.buf[(size_t)b], len, hipMemcpyHostToDevice, P.st));
                    DCHK(hipEventRecord(P.ev[b], P.st));
                    used_by[b] = (int)li;
                }
            }
            DCHK(hipEventRecord(t.buf->ready_event(), P.st));
            t.T0 = T0; t.len = T1 - T0; t.last = T1 == M.map.n;
            publish(M, std::move(t));
            T0 = T1;
        }
        TRACE("plain producer: %llu slabs in %.4f s, of which waiting for the consumers to hand a text buffer back %.4f s", (unsigned long long)s, now_s() - t_begin, t_slot);
        return MF_OK;
    }

    DevScratch *scratch_for(Worker &W, int ldev, std::string &err)
    {
        auto it = W.scratch.find(ldev);
        if (it != W.scratch.end()) { if (hipSetDevice(it->second->dev) != hipSuccess) { err = "hipSetDevice failed"; return nullptr; } return it->second.get(); }
        if (std::unique_ptr<DevScratch> kept = g_scratch.take(ldev, W.id)) {          // (a consumer's buffers and read set of an earlier call)
            if (hipSetDevice(kept->dev) != hipSuccess) { err = "hipSetDevice failed"; return nullptr; }
            DevScratch *p = kept.get();
            W.scratch[ldev] = std::move(kept);
            return p;
        }
        std::unique_ptr<DevScratch> S(new DevScratch());
        S->ldev = ldev; S->dev = phys(ldev); S->lane = W.id;
        if (get_ctx(ldev, &S->ctx, W.id)) { err = mf_thread_error(); return nullptr; }
        if (hipHostMalloc((void **)&S->h_small, 128, hipHostMallocDefault) != hipSuccess) { err = "hipHostMalloc failed"; return nullptr; }
        memset(S->h_small, 0, 128);
        DevScratch *p = S.get();
        W.scratch[ldev] = std::move(S);
        return p;
    }

    // ---- step A of a piece (one piece of a mate at a time, in order): the carry goes in front of its text, lines are counted and
    // indexed, the records counted; what is behind the last complete record is the next piece's carry.
    int index_piece(Worker &W, Mate &M, TextPiece &P, std::shared_ptr<Batch> &Bout, std::string &err)
    {
        const double t0 = now_s();
        DevScratch *Sp = scratch_for(W, P.buf->ldev, err);
        if (!Sp) return MF_E_HIP;
        DevScratch &S = *Sp;
        const int dev = S.dev;
        hipStream_t sp = S.ctx->stream;
        if (P.buf->ready_recorded) DCHK(hipStreamWaitEvent(sp, P.buf->ready, 0));          // (the piece's link, resolve and CRC kernels may still be running)
        const bool first_piece = !first_indexed_.exchange(true);
        if (first_piece) cold_mark("consumer: first piece taken");
        // the carry in front of the piece's text.  It fits the room in front of the buffer -- or the piece moves to a buffer that
        // holds both (records longer than the room: tests, mostly)
        if (M.carry > P.buf->pad) {
            std::unique_ptr<TextBuf> nb;
            DCHK(TextBuf::make(nb, dev, P.buf->ldev, TEXT_FRONT + M.carry, (size_t)P.len, nullptr));
            DCHK(hipMemcpyAsync(nb->p, P.buf->p, P.len, hipMemcpyDeviceToDevice, sp));
            DCHK(hipStreamSynchronize(sp));
            nb->slots = P.buf->slots; P.buf->slots = nullptr;
            P.buf = std::move(nb);
        }
        if (M.carry) DCHK(launch_bytes_from_host(P.buf->p - M.carry, M.h_carry, M.carry, sp));          // (not the copy engine: mf_ingest.h)
        std::shared_ptr<Batch> B(new Batch());
        B->ldev = S.ldev;
        B->text = P.buf->p - M.carry;
        const uint8_t *text = B->text;
        const uint64_t n = M.carry + P.len;
        B->n_text = n;
        const uint64_t tiles = (n + INGEST_TILE - 1) / INGEST_TILE;
        volatile uint64_t *hs = S.h_small;
        uint64_t n_lines = 0, used = 0;
        hs[2] = 0;
        if (n) {
            DCHK(S.tile_cnt.need(dev, tiles)); DCHK(S.tile_base.need(dev, tiles + 1)); DCHK(S.scan_tmp.need(dev, tiles / 4096 + 4));
            DCHK(launch_count_newlines(text, n, S.tile_cnt.p, sp));
            DCHK(launch_scan_u32(S.tile_cnt.p, tiles, S.tile_base.p, S.scan_tmp.p, sp));
            hs[1] = 0;
            DCHK(launch_bytes_to_host(S.h_small + 0, S.tile_base.p + tiles, 8, sp));
            DCHK(launch_bytes_to_host(S.h_small + 1, text + n - 1, 1, sp));
            DCHK(hipStreamSynchronize(sp));
            if (first_piece) cold_mark("consumer: first piece is text, its newlines counted");
            const uint64_t newlines = hs[0]; const uint8_t last_byte = (uint8_t)hs[1];
            const bool open_line = P.last && last_byte != '\n';      // lines() yields an unterminated last line
            n_lines = newlines + (open_line ? 1 : 0);
            DCHK(B->line_start.need(dev, n_lines + 2, false));
            DCHK(launch_line_starts(text, n, S.tile_base.p, B->line_start.p, sp));
            if (open_line) { S.h_small[7] = n + 1; DCHK(launch_bytes_from_host(B->line_start.p + n_lines, S.h_small + 7, 8, sp)); }
            B->n_rec = n_lines / 4; B->n_lines = n_lines;
            DCHK(launch_bytes_to_host(S.h_small + 2, B->line_start.p + 4 * B->n_rec, 8, sp));
            DCHK(hipStreamSynchronize(sp));
        }
        used = hs[2];
        if (used > n) used = n;                                       // (the virtual line end of an unterminated last line)
        const size_t carry = P.last ? 0 : (size_t)(n - used);         // a partial record at the very end is dropped
        if (carry) {
            if (carry > M.h_carry_cap) {
                uint8_t *q = nullptr;
                DCHK(hipHostMalloc((void **)&q, carry + carry / 2 + 4096, PINNED_FOR_KERNELS));
                if (M.h_carry) (void)hipHostFree(M.h_carry);
                M.h_carry = q; M.h_carry_cap = carry + carry / 2 + 4096;
            }
            DCHK(launch_bytes_to_host(M.h_carry, text + used, carry, sp));
            DCHK(hipStreamSynchronize(sp));
        }
        M.carry = carry;
        B->buf = std::move(P.buf);
        Bout = std::move(B);
        if (first_piece) cold_mark("consumer: first piece indexed");
        if (timing) { std::lock_guard<std::mutex> lk(mu); t_index += now_s() - t0; }
        return MF_OK;
    }

    // ---- step B (several pieces side by side, each on its consumer's own streams): records -> the consumer's read set -> one
    // filter pass; the pass bits come back in S.h_bits
    int filter_piece(Worker &W, Batch &Bt, double grow, std::string &err)
    {
        const double t1 = now_s();
        DevScratch *Sp = scratch_for(W, Bt.ldev, err);
        if (!Sp) return MF_E_HIP;
        DevScratch &S = *Sp;
        const int dev = S.dev;
        hipStream_t sp = S.ctx->stream;
        volatile uint64_t *hs = S.h_small;
        const uint8_t *text = Bt.text;
        const uint64_t n_rec = Bt.n_rec;
        const bool first_set = !S.reads;
        if (!S.reads) { S.reads = new (std::nothrow) mf_reads(); if (!S.reads) { err = "out of memory"; return MF_E_NOMEM; } S.reads->device = S.ldev; S.reads->lane = S.lane; }
        mf_reads *R = S.reads;
        // sequence lengths, the piece's own base offsets
        DCHK(S.seq_len.need(dev, n_rec)); DCHK(S.minmax.need(dev, 2)); DCHK(S.offsets_tmp.need(dev, n_rec + 1)); DCHK(S.scan_tmp.need(dev, n_rec / 4096 + 4));
        S.h_small[8] = (uint64_t)0xFFFFFFFFull;                   // {~0u, 0u}
        DCHK(launch_bytes_from_host(S.minmax.p, S.h_small + 8, 8, sp));
        DCHK(launch_seq_lens(text, Bt.line_start.p, n_rec, S.seq_len.p, S.minmax.p, sp));
        DCHK(launch_scan_u32(S.seq_len.p, n_rec, S.offsets_tmp.p, S.scan_tmp.p, sp));
        DCHK(launch_bytes_to_host(S.h_small + 3, S.offsets_tmp.p + n_rec, 8, sp));
        DCHK(launch_bytes_to_host(S.h_small + 4, S.minmax.p, 8, sp));
        DCHK(hipStreamSynchronize(sp));
        const uint64_t nb = hs[3]; const uint32_t mm[2] = {(uint32_t)hs[4], (uint32_t)(hs[4] >> 32)};
        const uint32_t uniform = (mm[0] == mm[1] && mm[0] > 0) ? mm[0] : 0;
        const uint64_t n_words = (nb + 15) / 16;
        // invalid bases are rare (N calls): room for one in 64 bases, more when a piece proves to need it
        const uint64_t pb = pack_blocks(nb, 0);
        uint64_t npos_cap = std::max<uint64_t>(nb / 64 + 1024, S.reads->cap_npos / 8);
        int rc = MF_OK;
        if (first_set && grow > 1.0) {
            // The read set is refilled piece after piece, and growing it means hipFree -- which waits for every kernel on the
            // device, the decoder's included.  The first slabs of a .gz are short ones: give the set the size of a full slab's now.
            const double g = std::min(grow, 64.0) * 1.2;
            const uint64_t nw = (uint64_t)((double)n_words * g), nr = (uint64_t)((double)n_rec * g);
            rc = reads_reserve(R, true, nw, nr, 0, (uint64_t)((double)npos_cap * g), S.ctx);
            if (!rc) rc = reads_finish(R, true, nw, nr, nw * 16, 0, 0, S.ctx);        // (no invalid positions: nothing of the empty set is read)
            if (rc) { err = mf_thread_error(); return rc; }
            const size_t bw = (size_t)(nr / 32 + 1024);
            if (bw > S.h_bits_cap) { if (S.h_bits) (void)hipHostFree(S.h_bits); S.h_bits = nullptr; S.h_bits_cap = 0; DCHK(hipHostMalloc((void **)&S.h_bits, bw * 4, hipHostMallocDefault)); S.h_bits_cap = bw; }
        }
        rc = reads_reserve(R, true, n_words, n_rec, uniform, npos_cap, S.ctx);
        if (rc) { err = mf_thread_error(); return rc; }
        if (!uniform) DCHK(hipMemcpyAsync(R->d_offsets, S.offsets_tmp.p, (n_rec + 1) * 8, hipMemcpyDeviceToDevice, sp));
        uint64_t inv = 0;
        if (pb) {
            DCHK(S.inv_cnt.need(dev, pb)); DCHK(S.inv_base.need(dev, pb + 1)); DCHK(S.scan_tmp.need(dev, pb / 4096 + 4));
            DCHK(launch_pack(text, Bt.line_start.p, uniform ? nullptr : S.offsets_tmp.p, uniform, n_rec, nb, 0, R->d_words, S.inv_cnt.p, nullptr, nullptr, sp));
            DCHK(launch_scan_u32(S.inv_cnt.p, pb, S.inv_base.p, S.scan_tmp.p, sp));
            DCHK(launch_bytes_to_host(S.h_small + 5, S.inv_base.p + pb, 8, sp));
            DCHK(hipStreamSynchronize(sp));
            inv = hs[5];
            if (inv) {
                if (inv > npos_cap) { rc = reads_reserve(R, true, n_words, n_rec, uniform, inv, S.ctx); if (rc) { err = mf_thread_error(); return rc; } }     // (words and offsets stay where they are: only the list grows)
                DCHK(launch_pack(text, Bt.line_start.p, uniform ? nullptr : S.offsets_tmp.p, uniform, n_rec, nb, 0, R->d_words, S.inv_cnt.p, S.inv_base.p, R->d_npos, sp));
            }
        }
        const double t2 = now_s();
        rc = reads_finish(R, true, n_words, n_rec, nb, uniform, inv, S.ctx);
        if (rc) { err = mf_thread_error(); return rc; }
        const size_t bw = (size_t)((n_rec + 31) / 32);
        if (bw + 2 > S.h_bits_cap) { if (S.h_bits) (void)hipHostFree(S.h_bits); S.h_bits = nullptr; S.h_bits_cap = 0; DCHK(hipHostMalloc((void **)&S.h_bits, (bw + bw / 2 + 1024) * 4, hipHostMallocDefault)); S.h_bits_cap = bw + bw / 2 + 1024; }
        rc = filter_common(ks, R, threshold, MF_MODE_SCREENED, S.h_bits, nullptr, 1, nullptr);
        if (rc) { err = mf_thread_error(); return rc; }
        if (!first_filtered_.exchange(true)) cold_mark("consumer: first piece packed and filtered");
        {
            size_t f = 0, t = 0; const bool got = hipMemGetInfo(&f, &t) == hipSuccess;
            std::lock_guard<std::mutex> lk(mu);
            t_pack += t2 - t1; t_filter += now_s() - t2;
            if (got) mem_used_max = std::max(mem_used_max, t - f);
        }
        return MF_OK;
    }

    // survivors of the first n_emit records of batch B -> the mate's writer (emit_mu held)
    int emit(Worker &W, Mate &M, int mi, Batch &B, uint64_t n_emit, std::string &err)
    {
        if (n_emit > B.n_rec) n_emit = B.n_rec;
        if (!n_emit) return MF_OK;
        DevScratch *Sp = scratch_for(W, B.ldev, err);
        if (!Sp) return MF_E_HIP;
        DevScratch &S = *Sp;
        const int dev = S.dev;
        hipStream_t sp = S.ctx->stream;
        // the pair rule, on the host: this mate's bits and the other's over the batch's records
        const size_t bw = (size_t)((n_emit + 31) / 32);
        if (bw + 2 > S.h_bits_cap) { if (S.h_bits) (void)hipHostFree(S.h_bits); S.h_bits = nullptr; S.h_bits_cap = 0; DCHK(hipHostMalloc((void **)&S.h_bits, (bw + bw / 2 + 1024) * 4, hipHostMallocDefault)); S.h_bits_cap = bw + bw / 2 + 1024; }
        uint64_t keep_n = 0;
        {
            std::lock_guard<std::mutex> lk(mu);                      // (the bitmaps grow under other consumers' hands)
            extract_bits(M.bits, B.rec_base, n_emit, S.h_bits);
            if (nm == 2) {
                std::vector<uint32_t> other(bw);
                extract_bits(m[1 - mi].bits, B.rec_base, n_emit, other.data());
                for (size_t j = 0; j < bw; j++) S.h_bits[j] = pair_both ? (S.h_bits[j] & other[j]) : (S.h_bits[j] | other[j]);
            }
        }
        for (size_t j = 0; j < bw; j++) keep_n += (uint64_t)__builtin_popcount(S.h_bits[j]);
        if (mi == 0) kept += keep_n;
        if (!keep_n) return MF_OK;
        // the survivors are few: their record numbers go up as a list (in place of the mask they were read from), and the kernels
        // that measure and copy them run over the list
        {
            std::vector<uint32_t> idx; idx.reserve((size_t)keep_n);
            for (size_t j = 0; j < bw; j++) for (uint32_t wv = S.h_bits[j]; wv; wv &= wv - 1) idx.push_back((uint32_t)(j * 32 + (uint32_t)__builtin_ctz(wv)));
            if (keep_n > S.h_bits_cap) { (void)hipHostFree(S.h_bits); S.h_bits = nullptr; S.h_bits_cap = 0; DCHK(hipHostMalloc((void **)&S.h_bits, (keep_n + keep_n / 2 + 1024) * 4, hipHostMallocDefault)); S.h_bits_cap = keep_n + keep_n / 2 + 1024; }
            memcpy(S.h_bits, idx.data(), keep_n * 4);
        }
        DCHK(S.out_len.need(dev, keep_n)); DCHK(S.out_off.need(dev, keep_n + 1)); DCHK(S.scan_tmp.need(dev, keep_n / 4096 + 4));
        // The kernels read the list where it lies, in pinned host memory (a few thousand numbers a piece).  As a copy to the device it went
        // through the engine that carries the uploads, BEHIND them: with a plain pair's twelve 256 MiB slabs queued that was 40-50 ms a time
        // during which no text buffer came back and the link to the device ran dry (profiles/r05/g_pe_plain_trace_before.txt).
        const uint32_t *list = S.h_bits;
        DCHK(launch_sel_lens(B.text, B.line_start.p, list, keep_n, S.out_len.p, sp));
        DCHK(launch_scan_u32(S.out_len.p, keep_n, S.out_off.p, S.scan_tmp.p, sp));
        DCHK(launch_bytes_to_host(S.h_small + 6, S.out_off.p + keep_n, 8, sp));
        DCHK(hipStreamSynchronize(sp));
        const uint64_t bytes = ((volatile uint64_t *)S.h_small)[6];
        if (bytes) {
            DCHK(S.d_out.need(dev, bytes));
            DCHK(launch_sel_gather(B.text, B.line_start.p, list, keep_n, S.out_off.p, S.d_out.p, sp));
            if (bytes > S.h_out_cap) { if (S.h_out) (void)hipHostFree(S.h_out); S.h_out = nullptr; S.h_out_cap = 0; DCHK(hipHostMalloc((void **)&S.h_out, bytes + bytes / 2 + 65536, hipHostMallocDefault)); S.h_out_cap = bytes + bytes / 2 + 65536; }
            DCHK(hipMemcpyAsync(S.h_out, S.d_out.p, bytes, hipMemcpyDeviceToHost, sp));
            DCHK(hipStreamSynchronize(sp));
            wrote_any = true;
            M.out.push(std::vector<char>(S.h_out, S.h_out + bytes));
        }
        return MF_OK;
    }

    // write what can be written: the leading batches of either mate that are filtered and whose records the other mate's pass bits
    // cover (all that are left, cut at `total`, once `fin`).  One consumer at a time.
    int drain(Worker &W, bool fin, std::string &err)
    {
        std::lock_guard<std::mutex> elk(emit_mu);
        const double te = now_s();
        for (int i = 0; i < nm; i++) {
            Mate &M = m[i];
            for (;;) {
                std::shared_ptr<Batch> B; uint64_t covered = 0;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (M.batches.empty() || !M.batches.front()->filtered) break;
                    covered = fin ? total : (nm == 2 ? std::min(m[0].rec_filtered, m[1].rec_filtered) : M.rec_filtered);
                    // Pairs end with the shorter file: once the other mate has been filtered to its end, nothing of this mate at or behind
                    // that record will ever be written -- such a batch must not wait for the end of the call with its text buffer in hand
                    // (with the longer mate's buffers all held that way its producer never got another one: found by tests/native/ingest_check.cpp)
                    uint64_t end = M.batches.front()->rec_base + M.batches.front()->n_rec;
                    if (!fin && nm == 2 && scans_done(m[1 - i])) end = std::min(end, m[1 - i].rec_indexed);
                    if (!fin && end > covered) break;
                    B = std::move(M.batches.front()); M.batches.pop_front();
                }
                if (B->rec_base < covered) { const int rc = emit(W, M, i, *B, covered - B->rec_base, err); if (rc) return rc; }       // (pairs end with the shorter file)
            }
        }
        if (timing) { std::lock_guard<std::mutex> lk(mu); t_emit += now_s() - te; }
        return MF_OK;
    }

    // the next piece for a consumer: of the mate that is behind in records, if it has one ready (a mate whose text is not there yet
    // does not hold up the other).  false: nothing more will come (or the run has failed)
    // again (quality filter): set when nothing is ready yet but more may come -- the caller has other work to look after
    bool take_piece(int &mi, TextPiece &P, uint64_t &seq, std::string &err, int &rc, bool *again = nullptr)
    {
        const double tw = now_s();
        for (int round = 0;; round++) {
            {
                std::lock_guard<std::mutex> lk(mu);
                if (failed) return false;
                int order[2] = {0, 1};
                if (nm == 2 && m[1].rec_indexed < m[0].rec_indexed) { order[0] = 1; order[1] = 0; }
                bool any_open = false;
                for (int k = 0; k < nm; k++) {
                    Mate &M = m[order[k]];
                    if (M.eof) continue;
                    if (qual && qual->decided_final && M.rec_indexed >= qual->decided) {      // nothing behind the last decided record is wanted
                        M.eof = true; M.stop = true; M.slots.wake();
                        continue;
                    }
                    if (!qual && nm == 2 && scans_done(m[1 - order[k]]) && M.a_turn == M.taken && M.rec_indexed >= m[1 - order[k]].rec_indexed) {
                        M.eof = true; M.stop = true; M.slots.wake();          // the other mate has ended in front of this one's next record: pairs end with the shorter file
                        continue;
                    }
                    std::unique_lock<std::mutex> plk(M.mu);
                    if (!M.ready.empty()) {
                        P = std::move(M.ready.front()); M.ready.pop_front();
                        mi = order[k]; seq = M.taken++;
                        if (P.last) M.eof = true;
                        if (timing) t_wait += now_s() - tw;
                        return true;
                    }
                    if (M.prod_done) {
                        if (M.prod_rc) { rc = M.prod_rc; err = M.prod_err; return false; }
                        M.eof = true;                              // (an input without text: an empty file cannot get here, but a .gz of nothing can)
                    } else any_open = true;
                }
                if (!any_open) { if (timing) t_wait += now_s() - tw; return false; }
                if (again && round) { *again = true; if (timing) t_wait += now_s() - tw; return false; }
            }
            std::unique_lock<std::mutex> lk(mu_all);
            nap(cv_all, lk, 300);
        }
    }


    // ================================================================ the quality filter's job
    // A piece goes through: line index (in turn per mate, as above) -> SCAN (side by side: one pass over the records' bytes) ->
    // DECIDE (one piece at a time, mate 1's in file order: the tests, the de-duplication set, the -t budget; mate 2's pieces
    // pick up the keep flags of their records) -> GATHER + WRITE (side by side again: the pieces' places in the output files are
    // known from the decisions).  The two mates' pieces do not cover the same records, so what one mate's step needs of the other
    // travels through per-record arrays on the host (mate 2's counts, mate 1's keep flags).

    void update_scanned(Mate &M)          // (mu held) records of the leading scanned pieces
    {
        uint64_t upto = M.rec_filtered;
        for (auto &q : M.batches) { if (q->rec_base < upto) continue; if (q->rec_base != upto || !q->filtered) break; upto = q->rec_base + q->n_rec; }
        M.rec_filtered = upto;
    }
    bool scans_done(const Mate &M) const { return M.eof && M.a_turn == M.taken && M.rec_filtered == M.rec_indexed; }      // (mu held) every piece that will ever come is scanned

    int q_scan(Worker &W, int mi, Batch &B, std::string &err)
    {
        const double t0 = now_s();
        QualState &Q = *qual;
        DevScratch *Sp = scratch_for(W, B.ldev, err);
        if (!Sp) return MF_E_HIP;
        DevScratch &S = *Sp;
        const int dev = S.dev;
        hipStream_t sp = S.ctx->stream;
        volatile uint64_t *hs = S.h_small;
        const uint64_t n = B.n_rec;
        if (n >= 0xFFFFFFF0ull) { err = "a piece of text with 2^32 records"; return MF_E_ARG; }
        DCHK(B.q_bad.need(dev, n, false)); DCHK(B.q_sl.need(dev, n, false)); DCHK(B.q_ql.need(dev, n, false)); DCHK(B.q_olen.need(dev, n, false)); DCHK(B.q_fl.need(dev, n, false));
        DCHK(S.minmax.need(dev, 2));
        S.h_small[8] = ~0ull;
        DCHK(launch_bytes_from_host(S.minmax.p, S.h_small + 8, 8, sp));
        DCHK(launch_qual_scan(B.text, B.line_start.p, n, Q.P.start, Q.cap, Q.P.quality, Q.P.ns, B.q_bad.p, B.q_fl.p, B.q_sl.p, B.q_ql.p, B.q_olen.p, S.minmax.p, sp));
        if (mi == 0 && Q.P.dedup && !Q.P.trunc) { DCHK(B.q_hash.need(dev, n, false)); DCHK(launch_qual_hash(B.text, B.line_start.p, n, Q.P.start, B.q_sl.p, B.q_hash.p, sp)); }
        DCHK(launch_bytes_to_host(S.h_small + 4, S.minmax.p, 4, sp));
        uint32_t *h_bad = nullptr; uint8_t *h_fl = nullptr;
        if (mi == 1) {
            DCHK(S.stage(n * 5 + 16));
            h_bad = (uint32_t *)S.h_stage; h_fl = S.h_stage + n * 4;
            DCHK(hipMemcpyAsync(h_bad, B.q_bad.p, n * 4, hipMemcpyDeviceToHost, sp));
            DCHK(hipMemcpyAsync(h_fl, B.q_fl.p, n, hipMemcpyDeviceToHost, sp));
        }
        DCHK(hipStreamSynchronize(sp));
        const uint32_t first_flag = (uint32_t)hs[4];
        if (mi == 1 && (!Q.bad2.put(B.rec_base, n, h_bad) || !Q.fl2.put(B.rec_base, n, h_fl))) { err = "out of memory"; return MF_E_NOMEM; }
        uint64_t panic_at = ~0ull;
        if (first_flag != ~0u) {
            // rare: a byte that is not ASCII in a line the reference unwraps, or a string shorter than the cut's start.  The flagged
            // records are looked at on the host, in order, until one makes the reference panic (a header in UTF-8 does not).
            std::vector<uint8_t> fl(n), text(B.n_text + 1); std::vector<uint64_t> ls(4 * n + 1);
            // (on the consumer's own stream: a copy on the null stream would wait for every decode kernel in flight on the blocking CU-masked streams)
            DCHK(hipMemcpyAsync(fl.data(), B.q_fl.p, n, hipMemcpyDeviceToHost, sp));
            DCHK(hipMemcpyAsync(ls.data(), B.line_start.p, (4 * n + 1) * 8, hipMemcpyDeviceToHost, sp));
            DCHK(hipMemcpyAsync(text.data(), B.text, B.n_text, hipMemcpyDeviceToHost, sp));
            DCHK(hipStreamSynchronize(sp));
            auto line = [&](uint64_t k, const char *&p, size_t &len) {
                const uint64_t a = ls[k], b = std::min<uint64_t>(ls[k + 1], B.n_text + 1);
                len = (size_t)(b - a - 1); p = (const char *)text.data() + a;
                if (len && p[len - 1] == '\r') len--;
            };
            for (uint64_t r = first_flag; r < n && panic_at == ~0ull; r++) {
                if (!(fl[r] & (QF_HIGH | QF_SHORT | QF_LONG))) continue;
                if (fl[r] & QF_LONG) { err = "a FASTQ record of 4 GiB or more"; return MF_E_ARG; }
                if (fl[r] & QF_SHORT) { panic_at = r; break; }
                for (int k : {0, 1, 3}) { const char *p; size_t len; line(4 * r + k, p, len); if (!utf8_valid(p, len)) { panic_at = r; break; } }
            }
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            Mate &M = m[mi];
            if (panic_at != ~0ull) Q.panic_rec[mi] = std::min(Q.panic_rec[mi], B.rec_base + panic_at);
            B.filtered = true;
            update_scanned(M);
            Q.t_scan += now_s() - t0;
            size_t f = 0, t = 0;
            if (hipMemGetInfo(&f, &t) == hipSuccess) mem_used_max = std::max(mem_used_max, t - f);
        }
        return MF_OK;
    }

    int q_dedup_room(DevScratch &S, uint64_t n, std::string &err)          // the set holds at most half its slots after n more keys
    {
        QualState &Q = *qual;
        const int dev = S.dev; hipStream_t sp = S.ctx->stream;
        if (!Q.dd_slots) {
            // (as many slots as four times the records the input is likely to hold, at most 2^24 to begin with: the set doubles as it fills)
            uint64_t text_est = 0;
            for (int i = 0; i < nm; i++) text_est += m[i].gz ? m[i].map.n * 4 : m[i].map.n;
            uint64_t lg_est = 16; while (lg_est < 24 && ((uint64_t)1 << lg_est) < text_est / (uint64_t)nm / 300 * 4) lg_est++;
            uint64_t lg = env_u64("MF_DEDUP_LOG2_SLOTS", lg_est);
            lg = std::min<uint64_t>(std::max<uint64_t>(lg, 4), 34);
            Q.dd_slots = (uint64_t)1 << lg;
            DCHK(Q.dd_keys.need(dev, Q.dd_slots, false)); DCHK(Q.dd_first.need(dev, Q.dd_slots, false));
            DCHK(hipMemsetAsync(Q.dd_keys.p, 0, Q.dd_slots * 8, sp)); DCHK(hipMemsetAsync(Q.dd_first.p, 0xFF, Q.dd_slots * 8, sp));
        }
        while (2 * (Q.dd_n + n) > Q.dd_slots) {
            DevBuf<unsigned long long> k2, f2;
            DCHK(k2.need(dev, Q.dd_slots * 2, false)); DCHK(f2.need(dev, Q.dd_slots * 2, false));
            DCHK(hipMemsetAsync(k2.p, 0, Q.dd_slots * 16, sp)); DCHK(hipMemsetAsync(f2.p, 0xFF, Q.dd_slots * 16, sp));
            DCHK(launch_dedup_rehash(Q.dd_keys.p, Q.dd_first.p, Q.dd_slots, k2.p, f2.p, Q.dd_slots * 2, sp));
            DCHK(hipStreamSynchronize(sp));
            std::swap(Q.dd_keys.p, k2.p); std::swap(Q.dd_keys.cap, k2.cap); std::swap(Q.dd_keys.bytes_, k2.bytes_);
            std::swap(Q.dd_first.p, f2.p); std::swap(Q.dd_first.cap, f2.cap); std::swap(Q.dd_first.bytes_, f2.bytes_);
            Q.dd_slots *= 2;
        }
        return MF_OK;
    }

    // records [r0, r0 + n) of mate 1's piece B (emit_mu held): the tests, the de-duplication, the budget; where their output goes.
    // *stopped: the -t budget ran out among them (part.n: the records in front of the one that overflowed it)
    int q_decide(Worker &W, Batch &B, uint64_t r0, uint64_t n, QPart &part, bool *stopped, std::string &err)
    {
        const uint64_t g0 = B.rec_base + r0;          // file index of the first
        const double t0 = now_s();
        QualState &Q = *qual;
        DevScratch *Sp = scratch_for(W, B.ldev, err);
        if (!Sp) return MF_E_HIP;
        DevScratch &S = *Sp;
        const int dev = S.dev;
        hipStream_t sp = S.ctx->stream;
        volatile uint64_t *hs = S.h_small;
        uint64_t bytes = 0, kept_here = 0;
        if (n) {
            DCHK(S.q_alive.need(dev, n)); DCHK(S.q_keep.need(dev, n)); DCHK(S.out_len.need(dev, n)); DCHK(S.out_off.need(dev, n + 1)); DCHK(S.scan_tmp.need(dev, n / 4096 + 4));
            DCHK(S.stage(n * 5 + 16));
            if (Q.pe) {
                DCHK(S.q_bad2.need(dev, n)); DCHK(S.q_fl2.need(dev, n));
                Q.bad2.get(g0, n, (uint32_t *)S.h_stage); Q.fl2.get(g0, n, S.h_stage + n * 4);
                DCHK(launch_bytes_from_host(S.q_bad2.p, S.h_stage, n * 4, sp));
                DCHK(launch_bytes_from_host(S.q_fl2.p, S.h_stage + n * 4, n, sp));
            }
            DCHK(launch_qual_decide(n, Q.pe, Q.P.trunc, Q.P.limit, B.q_bad.p + r0, B.q_fl.p + r0, B.q_sl.p + r0, B.q_ql.p + r0, S.q_bad2.p, S.q_fl2.p, S.q_alive.p, sp));
            const bool dd = Q.P.dedup && !Q.P.trunc;
            if (!Q.dd_small.p) {          // [0] file index of the hash value 0, [1] keys in the set, [2] kept records of a piece
                DCHK(Q.dd_small.need(dev, 4, false));
                DCHK(hipMemsetAsync(Q.dd_small.p, 0xFF, 8, sp)); DCHK(hipMemsetAsync(Q.dd_small.p + 1, 0, 24, sp));
            }
            if (dd) { const int rc = q_dedup_room(S, n, err); if (rc) return rc; }
            if (dd) {
                DCHK(S.q_dup.need(dev, n));
                DCHK(launch_dedup(B.q_hash.p + r0, S.q_alive.p, (uint32_t)n, g0, Q.dd_keys.p, Q.dd_first.p, Q.dd_slots, Q.dd_small.p, Q.dd_small.p + 1, S.q_dup.p, sp));
            }
            DCHK(hipMemsetAsync(Q.dd_small.p + 2, 0, 8, sp));
            DCHK(launch_qual_keep(n, S.q_alive.p, dd ? S.q_dup.p : nullptr, B.q_olen.p + r0, S.q_keep.p, S.out_len.p, Q.dd_small.p + 2, sp));
            uint8_t *h_keep = S.h_stage;
            if (Q.P.trim) {
                // the budget is sequential (main.rs:254-259, 311-316): the first kept record that overflows it ends the run
                uint32_t *h_sl = (uint32_t *)(S.h_stage + ((n + 15) & ~(uint64_t)15));          // (n * 5 + 16 bytes are there)
                DCHK(hipMemcpyAsync(h_keep, S.q_keep.p, n, hipMemcpyDeviceToHost, sp));
                DCHK(hipMemcpyAsync(h_sl, B.q_sl.p + r0, n * 4, hipMemcpyDeviceToHost, sp));
                DCHK(hipStreamSynchronize(sp));
                uint64_t i = 0;
                for (; i < n; i++) {
                    if (!h_keep[i]) continue;
                    Q.budget += h_sl[i];
                    if (Q.budget > Q.P.trim) { *stopped = true; break; }
                    kept_here++;
                }
                n = i;
            }
            if (n) {
                DCHK(launch_scan_u32(S.out_len.p, n, S.out_off.p, S.scan_tmp.p, sp));
                DCHK(launch_bytes_to_host(S.h_small + 6, S.out_off.p + n, 8, sp));
                DCHK(launch_bytes_to_host(S.h_small + 3, Q.dd_small.p + 1, 16, sp));       // keys of the set, kept of the piece
                if (Q.pe && !Q.P.trim) DCHK(hipMemcpyAsync(h_keep, S.q_keep.p, n, hipMemcpyDeviceToHost, sp));
                DCHK(hipStreamSynchronize(sp));
                bytes = hs[6]; if (dd) Q.dd_n = hs[3];
                if (!Q.P.trim) kept_here = hs[4];
                if (Q.pe && !Q.keep.put(g0, n, h_keep)) { err = "out of memory"; return MF_E_NOMEM; }
            }
        }
        part.r0 = r0; part.n = n; part.bytes = bytes; part.out_at = Q.out_pos[0]; Q.out_pos[0] += bytes;
        Q.kept += kept_here;
        { std::lock_guard<std::mutex> lk(mu); Q.t_decide += now_s() - t0; }
        return MF_OK;
    }

    // mate 2's piece B, its first n records (emit_mu held): the keep flags mate 1's decisions left for them
    int q_keep2(Worker &W, Batch &B, uint64_t n, QPart &part, std::string &err)
    {
        const double t0 = now_s();
        QualState &Q = *qual;
        DevScratch *Sp = scratch_for(W, B.ldev, err);
        if (!Sp) return MF_E_HIP;
        DevScratch &S = *Sp;
        const int dev = S.dev;
        hipStream_t sp = S.ctx->stream;
        uint64_t bytes = 0;
        if (n) {
            DCHK(S.q_keep.need(dev, n)); DCHK(S.out_len.need(dev, n)); DCHK(S.out_off.need(dev, n + 1)); DCHK(S.scan_tmp.need(dev, n / 4096 + 4));
            DCHK(S.stage(n + 16));
            Q.keep.get(B.rec_base, n, S.h_stage);
            DCHK(launch_bytes_from_host(S.q_keep.p, S.h_stage, n, sp));
            DCHK(launch_qual_keep(n, S.q_keep.p, nullptr, B.q_olen.p, nullptr, S.out_len.p, nullptr, sp));
            DCHK(launch_scan_u32(S.out_len.p, n, S.out_off.p, S.scan_tmp.p, sp));
            DCHK(launch_bytes_to_host(S.h_small + 6, S.out_off.p + n, 8, sp));
            DCHK(hipStreamSynchronize(sp));
            bytes = ((volatile uint64_t *)S.h_small)[6];
        }
        part.r0 = 0; part.n = n; part.bytes = bytes; part.out_at = Q.out_pos[1]; Q.out_pos[1] += bytes;
        { std::lock_guard<std::mutex> lk(mu); Q.t_decide += now_s() - t0; }
        return MF_OK;
    }

    // the kept records of a decided part -> its place in the output file (any number of parts at a time; S.out_len / S.out_off are
    // still those of the part: the consumer that decided it is the one that gathers it, and does nothing in between).  The text is
    // done with once the records are gathered: the last part's consumer lets go of the piece (B) there, and its buffer goes back
    // to the producer while the output is on its way down and out.
    int q_emit(Worker &W, int mi, std::shared_ptr<Batch> &B, const QPart &part, std::string &err)
    {
        QualState &Q = *qual;
        if (!part.bytes) { B.reset(); return MF_OK; }
        const double t0 = now_s();
        DevScratch *Sp = scratch_for(W, B->ldev, err);
        if (!Sp) return MF_E_HIP;
        DevScratch &S = *Sp;
        const int dev = S.dev;
        hipStream_t sp = S.ctx->stream;
        const uint64_t bytes = part.bytes;
        DCHK(S.d_out.need(dev, bytes));
        DCHK(launch_qual_gather(B->text, B->line_start.p + 4 * part.r0, part.n, Q.P.start, B->q_sl.p + part.r0, B->q_ql.p + part.r0, S.out_len.p, S.out_off.p, S.d_out.p, sp));
        DCHK(hipStreamSynchronize(sp));
        B.reset();
        const size_t chunk = Q.chunks.chunk();
        double tw = 0;
        { const double w0 = now_s(); if (!Q.sink[mi].wait_turn(part.out_at)) { err = "abandoned"; return MF_E_IO; } tw += now_s() - w0; }      // (standard output, a pipe, a .gz: the parts' chunks are taken in file order -- tests/native/qualsink_check.cpp hangs without it)      // (standard output, a pipe, a .gz: the parts' chunks are taken in file order)
        for (uint64_t off = 0; off < bytes; off += chunk) {
            const uint64_t len = std::min<uint64_t>(chunk, bytes - off);
            const double w0 = now_s();
            bool no_mem = false;
            uint8_t *p = Q.chunks.take(&no_mem);
            tw += now_s() - w0;
            if (!p) { if (no_mem) { err = "hipHostMalloc failed: no pinned memory for the output's chunks"; return MF_E_NOMEM; } err = "abandoned"; return MF_E_IO; }
            hipError_t c = hipMemcpyAsync(p, S.d_out.p + off, len, hipMemcpyDeviceToHost, sp);
            if (c == hipSuccess) c = hipStreamSynchronize(sp);
            if (c != hipSuccess) { Q.chunks.give(p); err = std::string("copy of the output failed: ") + hipGetErrorString(c); return MF_E_HIP; }
            wrote_any = true;                   // (before the first byte reaches the sink: a later failure must not hand the call to the host pipeline, which would write them again)
            Q.sink[mi].push(part.out_at + off, p, (size_t)len);
            if (!Q.sink[mi].ok()) { err = std::string("write error on ") + out_name(mi); return MF_E_IO; }
        }
        { std::lock_guard<std::mutex> lk(mu); Q.t_gather += now_s() - t0 - tw; Q.t_chunk += tw; }
        return MF_OK;
    }
    std::string out_path_[2];
    const char *out_name(int mi) const { return out_path_[mi].empty() ? "<stdout>" : out_path_[mi].c_str(); }

    // decide and write what can be decided and written.  true: did something
    bool q_progress(Worker &W, std::string &err, int &rc)
    {
        QualState &Q = *qual;
        bool did = false;
        for (;;) {
            std::shared_ptr<Batch> B; int mi = -1; uint64_t r0 = 0, n = 0; bool final = false, by_panic = false, stopped = false, whole = false;
            QPart part;
            {
                std::unique_lock<std::mutex> elk(emit_mu, std::try_to_lock);          // (somebody else is deciding: there is other work)
                if (!elk.owns_lock()) break;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (failed) break;
                    Mate &A = m[0];
                    while (Q.decided_final && !A.batches.empty() && A.batches.front()->filtered) { A.batches.pop_front(); did = true; }      // (nothing of them is wanted)
                    if (!Q.decided_final && !A.batches.empty() && A.batches.front()->filtered) {
                        Batch &F = *A.batches.front();
                        const uint64_t end = F.rec_base + F.n_rec, cur = F.rec_base + F.q_done;
                        uint64_t limit = std::min(end, Q.panic_rec[0]), upto = limit;          // limit: what of the piece will ever be decided
                        bool ready = true;
                        if (Q.pe) {
                            const bool other_done = scans_done(m[1]);
                            limit = std::min(limit, Q.panic_rec[1]);
                            if (other_done) limit = std::min(limit, m[1].rec_indexed);          // (pairs end with the shorter file)
                            upto = other_done ? limit : std::min(limit, m[1].rec_filtered);     // ... and what can be now: the records mate 2's scanned pieces cover
                            // A part of the piece is decided only when waiting for the rest cannot end: the other mate holds all its text buffers
                            // (its pieces wait for THESE decisions before they are written and their buffers come back)
#ifdef MF_TEST_WITHOUT_PARTIAL_DECISIONS         // (tests/test_ingest_orchestration.py: the check must hang without this rule, as the path did before it had it)
                            ready = upto == limit;
#else
                            ready = upto == limit || (upto > cur && m[1].slots.none_free());
#endif
                        }
                        if (ready) {
                            whole = upto == limit;
                            final = whole && limit < end;
                            by_panic = final && std::min(Q.panic_rec[0], Q.panic_rec[1]) == limit;
                            r0 = F.q_done; n = upto > cur ? upto - cur : 0;
                            B = A.batches.front(); mi = 0;
                            if (whole) A.batches.pop_front();
                        }
                    }
                    if (mi < 0 && !Q.decided_final && A.eof && A.a_turn == A.taken && A.batches.empty()) { Q.decided_final = true; did = true; }      // mate 1 has been decided to its end
                    if (mi < 0 && Q.pe && !m[1].batches.empty() && m[1].batches.front()->filtered) {
                        Batch &F = *m[1].batches.front();
                        const uint64_t end = F.rec_base + F.n_rec;
                        if (Q.decided >= end || Q.decided_final) {
                            const uint64_t upto = std::min(end, Q.decided);
                            n = upto > F.rec_base ? upto - F.rec_base : 0;
                            B = m[1].batches.front(); m[1].batches.pop_front(); mi = 1;
                        }
                    }
                    if (mi >= 0) Q.in_flight++;
                }
                if (mi < 0) break;
                rc = mi == 0 ? q_decide(W, *B, r0, n, part, &stopped, err) : q_keep2(W, *B, n, part, err);
                if (!rc && mi == 0) {
                    std::lock_guard<std::mutex> lk(mu);
                    B->q_done = r0 + part.n;
                    Q.decided = B->rec_base + B->q_done;
                    if (stopped && !whole) m[0].batches.pop_front();                            // (B is the front: nobody else decides)
                    if (final || stopped) { Q.decided_final = true; if (by_panic && !stopped) Q.panicked = true; }
                }
            }
            cv.notify_all(); cv_all.notify_all();
            if (!rc) rc = q_emit(W, mi, B, part, err);     // (lets go of B as soon as the records are gathered: the text buffer goes back, a producer may be waiting for one)
            B.reset();
            { std::lock_guard<std::mutex> lk(mu); Q.in_flight--; }
            cv.notify_all(); cv_all.notify_all();
            did = true;
            if (rc) return true;
        }
        return did;
    }

    void q_abandon() { qual->chunks.abort(); for (auto &sk : qual->sink) sk.abort(); }
    bool q_all_done()          // (takes mu)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (failed) return true;
        for (int i = 0; i < nm; i++) { Mate &M = m[i]; if (!M.eof || M.a_turn != M.taken || !M.batches.empty()) return false; }
        return qual->in_flight == 0;
    }

    void consume_q(Worker &W)
    {
        std::string err;
        for (;;) {
            int rc = MF_OK;
            bool did = q_progress(W, err, rc);
            if (rc) { fail_with(rc, err); q_abandon(); return; }
            int mi = 0; TextPiece P; uint64_t seq = 0; bool again = false;
            if (take_piece(mi, P, seq, err, rc, &again)) {
                Mate &M = m[mi];
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return failed || M.a_turn == seq; });
                    if (failed) return;
                }
                std::shared_ptr<Batch> B;
                rc = index_piece(W, M, P, B, err);
                Batch *Bp = nullptr;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (!rc) {
                        B->rec_base = M.rec_indexed; M.rec_indexed += B->n_rec;
                        if (B->n_rec) { Bp = B.get(); M.batches.push_back(std::move(B)); }
                    }
                    M.a_turn = seq + 1;
                }
                cv.notify_all();
                if (rc) { fail_with(rc, err); q_abandon(); return; }
                B.reset();
                if (Bp) { rc = q_scan(W, mi, *Bp, err); if (rc) { fail_with(rc, err); q_abandon(); return; } }
                continue;
            }
            if (rc) { fail_with(rc, err); q_abandon(); return; }
            if (q_all_done()) return;
            if (!did && !again) { std::unique_lock<std::mutex> lk(mu_all); nap(cv_all, lk, 200); }
        }
    }

    void fail_with(int rc, const std::string &err)
    {
        { std::lock_guard<std::mutex> lk(mu); if (!failed) { failed = true; fail_rc = rc; fail_err = err; } }
        cv.notify_all(); cv_all.notify_all();
    }

    void consume(Worker &W)
    {
        std::string err;
        for (;;) {
            int mi = 0, rc = MF_OK; TextPiece P; uint64_t seq = 0;
            if (!take_piece(mi, P, seq, err, rc)) { if (rc) fail_with(rc, err); TRACE("consumer %d: nothing more to take (rc %d)", W.id, rc); return; }
            Mate &M = m[mi];
            {   // the line index of a mate's pieces is cut in order
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return failed || M.a_turn == seq; });
                if (failed) return;
            }
            const double grow = P.grow;
            TRACE("consumer %d: piece %llu of mate %d (%llu bytes of text%s)", W.id, (unsigned long long)seq, mi + 1, (unsigned long long)P.len, P.last ? ", the last" : "");
            std::shared_ptr<Batch> B;
            rc = index_piece(W, M, P, B, err);
            Batch *Bp = nullptr;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!rc) {
                    B->rec_base = M.rec_indexed; M.rec_indexed += B->n_rec;
                    if (B->n_rec) { Bp = B.get(); M.batches.push_back(std::move(B)); }       // (a piece without a complete record has nothing to write: its buffer goes back now)
                }
                M.a_turn = seq + 1;
            }
            cv.notify_all();
            if (rc) { fail_with(rc, err); return; }
            B.reset();
            if (Bp) {
                rc = filter_piece(W, *Bp, grow, err);
                if (rc) { fail_with(rc, err); return; }
                TRACE("consumer %d: piece %llu of mate %d filtered: records %llu .. %llu", W.id, (unsigned long long)seq, mi + 1, (unsigned long long)Bp->rec_base, (unsigned long long)(Bp->rec_base + Bp->n_rec));
                DevScratch &S = *W.scratch[Bp->ldev];
                std::lock_guard<std::mutex> lk(mu);
                append_bits(M.bits, Bp->rec_base, Bp->n_rec, S.h_bits);
                Bp->filtered = true;
                // records of the leading filtered pieces (a piece without records is not in the list and holds nobody up)
                uint64_t upto = M.rec_filtered;
                for (auto &q : M.batches) { if (q->rec_base < upto) continue; if (q->rec_base != upto || !q->filtered) break; upto = q->rec_base + q->n_rec; }
                M.rec_filtered = upto;
            }
            rc = drain(W, false, err);
            if (rc) { fail_with(rc, err); return; }
        }
    }

    int run(std::string &err)
    {
        for (int i = 0; i < nm; i++) m[i].prod = std::thread([this, i] { producer(m[i]); });
        // consumers: three (the quality filter's, which spend their time writing: six) -- fewer for an input so small that a second
        // consumer's set-up (a stream, a read set) would take longer than the first one needs for the whole of it
        uint64_t text_est = 0;
        for (int i = 0; i < nm; i++) text_est += m[i].gz ? m[i].map.n * 4 : m[i].map.n;
        const uint64_t by_size = 1 + text_est / ((uint64_t)192 << 20);
        const int nw = (int)std::max<uint64_t>(1, std::min<uint64_t>(16, env_u64("MF_INGEST_CONSUMERS", std::min<uint64_t>(qual ? 6 : 3, by_size))));
        for (int w = 0; w < nw; w++) { workers.emplace_back(new Worker()); workers.back()->id = w; }
        for (auto &W : workers) { Worker *wp = W.get(); wp->th = std::thread([this, wp] { if (qual) consume_q(*wp); else consume(*wp); }); }
        for (auto &W : workers) W->th.join();
        t_consumed = now_s() - t_begin;
        if (failed) { err = fail_err; return fail_rc; }
        if (qual) {          // (a producer that was told to stop early -- the budget spent, a panic, the shorter mate's end -- has not failed)
            for (int i = 0; i < nm; i++) { m[i].stop = true; m[i].slots.wake(); }
            for (int i = 0; i < nm; i++) { Mate &M = m[i]; if (M.prod.joinable()) M.prod.join(); }
            total = qual->decided; kept = qual->kept;
            return MF_OK;
        }
        for (int i = 0; i < nm; i++) { Mate &M = m[i]; if (M.prod.joinable()) M.prod.join(); if (M.prod_rc) { err = M.prod_err; return M.prod_rc; } }
        total = nm == 2 ? std::min(m[0].rec_indexed, m[1].rec_indexed) : m[0].rec_indexed;
        return drain(*workers[0], true, err);
    }
};

bool alloc_failure(int rc) { return rc == MF_E_NOMEM; }

} // namespace

// what the two jobs of this path share: is it an input for the path, set-up, the run, what the caller learns about it
static int run_ingest(Ingest &I, const char *fq1, const char *fq2, const char *out1, const char *out2, std::string &err, IngestStats *stats)
{
    // (declared first: runs after everything of this call is gone.  What a process keeps between calls: up to MF_DEVPOOL_GB per device, default 8
    // -- enough for a caller that filters file after file of up to a gigabyte or two never to ask the runtime twice; a call on a file of several
    // gigabytes holds up to 20 GB and gives the rest back: hipMalloc of gigabytes takes a millisecond on this runtime (profiles/r05/a_cold_calls_before.log),
    // and a process that sits on 24 GB between calls, as round 4's did, is a poor neighbour on a shared GPU.  mf_release_cached() gives back all of it.)
    struct EndOfCall {
        bool timing = false; double t0 = 0;
        ~EndOfCall()
        {
            const char *kb = getenv("MF_KEEP_BUFFERS");
            g_pool.trim(kb && kb[0] == '0' ? 0 : (size_t)env_u64("MF_DEVPOOL_GB", 8) << 30);
            if (timing) fprintf(stderr, "[mf device ingest] streams, threads and buffers of the call put away in %.3f s\n", now_s() - t0);
        }
    } end_of_call;
    I.nm = fq2 ? 2 : 1;
    if (I.devices.empty() || I.devices.size() > 64) { err = "bad device list"; return MF_E_ARG; }
    I.timing = getenv("MF_PIPE_TIMING") != nullptr;
    I.carry_room = (size_t)env_u64("MF_INGEST_CARRY_ROOM", (size_t)1 << 20);
    const char *in_path[2] = {fq1, fq2}, *out_path[2] = {out1, out2};
    for (int i = 0; i < I.nm; i++) I.out_path_[i] = out_path[i] ? out_path[i] : "";
    // ---- is this an input for the device path?
    for (int i = 0; i < I.nm; i++) {
        Mate &M = I.m[i];
        if (!in_path[i]) return MF_DEVINGEST_DECLINED;          // (standard input)
        M.path = in_path[i]; M.gz = has_gz_ext(in_path[i]);
        bool regular = false;
        if (!M.map.open(in_path[i], regular)) { err = std::string("Cannot open file ") + in_path[i]; return MF_E_IO; }
        if (!regular || M.map.n == 0) return MF_DEVINGEST_DECLINED;
        if (M.gz) {
            const uint8_t *d = M.map.p;
            if (M.map.n < 18 || d[0] != 0x1f || d[1] != 0x8b) return MF_DEVINGEST_DECLINED;      // (gzread hands such a file through; so does the host reader)
            if ((d[3] & 4) && M.map.n >= 18 && d[12] == 'B' && d[13] == 'C') return MF_DEVINGEST_DECLINED;   // BGZF: the host reader decodes its members side by side
        }
    }
    cold_mark("device ingest: inputs mapped");
    for (int d : I.devices) ingest_prefetch(d);          // (the streams' maker and the staging buffers: started now if nobody has yet)
    for (int d : I.devices) { DevCtx *c = nullptr; const int rc = get_ctx(d, &c, 0); if (rc) { err = mf_thread_error(); return rc; } }
    cold_mark("device ingest: device contexts ready");
    const double t_begin = now_s();
    I.t_begin = t_begin;
    g_pool.reset_peak();
    if (g_trace) { size_t f = 0, t = 0; (void)hipMemGetInfo(&f, &t); TRACE("device memory in use as the call starts %.3f GB, of which idle buffers of earlier calls %.3f GB", (double)(t - f) / 1e9, (double)g_pool.held(phys(I.devices[0])) / 1e9); }
    // text buffers a mate may hold: the consumers each hold one, the decoder one, the rest wait for the other mate or for a consumer (the quality
    // filter's pieces wait longer: their text is written out)
    // (a call that plans for less than 8 GB keeps two fewer in flight: a text buffer is a slab's text, 4.5 times its compressed bytes)
    uint64_t in_bytes = 0;
    for (int i = 0; i < I.nm; i++) in_bytes += I.m[i].map.n;
    const bool small_call = 8 * in_bytes < ((uint64_t)8 << 30) && !getenv("MF_INGEST_BUDGET_GB");
    const int text_bufs = (int)std::max<uint64_t>(2, env_u64("MF_INGEST_TEXT_BUFS", (I.qual ? 8 : 6) - (small_call ? 2 : 0))) + (int)I.devices.size() - 1;
    int rc = MF_OK;
    // an input that keeps the chip full of decode wavefronts for a long time gets the CU-masked set of streams (16 ms apiece to make and a
    // quarter of a second of the process's exit: a small file must not pay for them)
    uint64_t gz_bytes = 0;
    for (int i = 0; i < I.nm; i++) if (I.m[i].gz) gz_bytes += I.m[i].map.n;
    // Which set of streams: the CU-masked set is faster the moment decode kernels fill the chip for more than a few slabs -- a paired 2 x 1.2 GB input
    // 0.149 s against 0.263 s on plain streams, configs[4] 0.225 against 0.368 (calls of a warm process, profiles/r05/e_masks_ab.txt) -- and costs a
    // quarter of a second of the process's exit.  A library user's process lives on: masked from 256 MB of compressed input.  A process that
    // makes one call and ends (the CLIs say so: mf_set_option("short_lived", "1")) pays the exit with every call: masked only where the
    // difference is larger than that, from 8 GB.  MF_GZDEV_LARGE_MB overrides either.
    const bool large = gz_bytes >= (env_u64("MF_GZDEV_LARGE_MB", g_short_lived.load() ? 8192 : 256) << 20);
    // Device memory follows the input: everything in use on the device stays within 8 bytes per compressed byte of the call, at least 3 GB,
    // at most 24 (MF_INGEST_BUDGET_GB sets it).  Of that, 1.2 GB are not this path's (the runtime's own 0.83 GB as a process starts, the
    // streams' queues, the bait tables); and for every byte the mates plan for their rings, symbol rooms, code lists and text buffers
    // (GzStream::open) the call holds 0.5-0.7 more -- the consumers' read sets and line indexes, which grow with the text pieces, and
    // buffers of one size idle in the pool while another size is asked for (profiles/r05/g_mem_probe.txt: planned 1.61 GB -> 2.35 GB of
    // buffers, 3.49-3.70 GB in use; planned 5.8 -> 7.7, 9.4-10.0 in use).  The mates share what is left.
    const uint64_t budget = getenv("MF_INGEST_BUDGET_GB") ? env_u64("MF_INGEST_BUDGET_GB", 24) << 30
                                                         : std::min<uint64_t>((uint64_t)24 << 30, std::max<uint64_t>((uint64_t)3 << 30, 8 * gz_bytes));
    const uint64_t not_ours = (uint64_t)1200 << 20;
    const uint64_t gz_budget = (budget > 2 * not_ours ? (budget - not_ours) * 10 / 17 : budget / 4) / (uint64_t)I.nm;
    {   // (the two mates' decoders side by side)
        int rcs[2] = {MF_OK, MF_OK}; std::string errs[2]; std::thread th[2];
        for (int i = 0; i < I.nm; i++) {
            Mate &M = I.m[i];
            M.slots.free_ = text_bufs; M.slots.stop = &M.stop;
            if (!M.gz) continue;
            M.gzs.reset(new GzStream());
            auto open = [&I, &M, &rcs, &errs, i, large, gz_budget, text_bufs] { rcs[i] = M.gzs->open(M.map.p, M.map.n, M.map.fd, I.devices, M.path, &M.slots, I.carry_room, I.nm == 2 ? 7 : 12, large, gz_budget, (uint32_t)text_bufs, I.qual != nullptr, &M.stop, errs[i]); };
            if (i == 0 && I.nm == 2 && I.m[1].gz == false) open(); else if (i == 0 && I.nm == 2) th[0] = std::thread(open); else open();
        }
        for (auto &t : th) if (t.joinable()) t.join();
        for (int i = 0; i < I.nm && !rc; i++) if (rcs[i]) { rc = rcs[i]; err = errs[i]; }
    }
    if (alloc_failure(rc)) { TRACE("declined: %s", err.c_str()); return MF_DEVINGEST_DECLINED; }      // (nothing has been touched: the host pipeline streams the file)
    if (rc) return rc;
    for (int i = 0; i < I.nm; i++)
        if (!(I.qual ? I.qual->sink[i].open(out_path[i], &I.qual->chunks) : I.m[i].out.open(out_path[i]))) { err = std::string("Cannot open file ") + (out_path[i] ? out_path[i] : "<stdout>"); return MF_E_IO; }
    const double t_setup = now_s() - t_begin;
    cold_mark("device ingest: decoders and outputs open");
    rc = I.run(err);
    cold_mark("device ingest: consumers done");
    TRACE("run returned %d", rc);
    for (int i = 0; i < I.nm; i++) { I.m[i].stop = true; I.m[i].slots.wake(); }
    bool wrote = true;
    for (int i = 0; i < I.nm; i++) wrote = (I.qual ? I.qual->sink[i].close() : I.m[i].out.close()) && wrote;
    // out of device memory before a byte of the survivors was written: the host pipeline takes the file (it truncates the outputs again)
    if (alloc_failure(rc) && !I.wrote_any) { TRACE("declined after a failed allocation: %s", err.c_str()); return MF_DEVINGEST_DECLINED; }
    if (rc) return rc;
    if (!wrote) { err = std::string("write error on ") + (out_path[0] ? out_path[0] : "<stdout>"); return MF_E_IO; }
    if (stats) {
        *stats = IngestStats();
        stats->seconds = now_s() - t_begin; stats->n_devices = (int)I.devices.size(); stats->consumers = (int)I.workers.size();
        stats->pool_bytes_peak = g_pool.peak(); stats->device_bytes_peak = I.mem_used_max;
        for (int i = 0; i < I.nm; i++) {
            Mate &M = I.m[i];
            stats->input_bytes += M.map.n; stats->records += M.rec_indexed;
            if (M.gzs) {
                stats->text_bytes += M.gzs->text_bytes(); stats->decode_busy_seconds += M.gzs->decode_busy_seconds();
                stats->chunks += M.gzs->chunks(); stats->chunks_linked += M.gzs->chunks_linked(); stats->gaps += M.gzs->gaps(); stats->gap_bytes += M.gzs->gap_bytes();
            } else stats->text_bytes += M.map.n;
        }
    }
    if (rc == MF_OK) g_streams.stage_later();
    if (I.timing) {
        if (I.qual)
            fprintf(stderr, "[mf device ingest] quality filter: wall %.3f s | set-up %.3f | consumers (summed over %zu): waiting for text %.3f, line index %.3f, scan %.3f, decisions %.3f, gather + copy down %.3f, waiting for the writers %.3f; writers busy %.3f %.3f | %llu + %llu bytes written | buffers of this call at most %.2f GB, device memory in use at most %.2f GB",
                    now_s() - t_begin, t_setup, I.workers.size(), I.t_wait, I.t_index, I.qual->t_scan, I.qual->t_decide, I.qual->t_gather, I.qual->t_chunk, I.qual->sink[0].busy(), I.qual->sink[1].busy(),
                    (unsigned long long)I.qual->out_pos[0], (unsigned long long)I.qual->out_pos[1], (double)g_pool.peak() / 1e9, (double)I.mem_used_max / 1e9);
        else
            fprintf(stderr, "[mf device ingest] wall %.3f s | set-up %.3f | waiting for text (upload, inflate, link, CRC on the producer threads) %.3f | line index %.3f | pack %.3f | filter %.3f | survivors %.3f | buffers of this call at most %.2f GB on a device, device memory in use at most %.2f GB, %zu device(s)",
                    now_s() - t_begin, t_setup, I.t_wait, I.t_index, I.t_pack, I.t_filter, I.t_emit, (double)g_pool.peak() / 1e9, (double)I.mem_used_max / 1e9, I.devices.size());
        { double tm; uint64_t nm; g_pool.malloc_time(tm, nm); fprintf(stderr, " | %llu new device allocations took %.3f s (summed over the threads that asked); idle buffers now %.2f GB", (unsigned long long)nm, tm, (double)g_pool.held(phys(I.devices[0])) / 1e9); }
        fprintf(stderr, " | first text after %.3f s, last after %.3f, consumers done after %.3f", I.t_first_piece, I.t_last_piece, I.t_consumed);
        for (int i = 0; i < I.nm; i++)
            if (I.m[i].gzs) { double a, b, c, d, e; I.m[i].gzs->producer_times(a, b, c, d, e); fprintf(stderr, " | mate %d producer: launching (incl. waiting for the upload) %.3f, waiting for decode %.3f, link %.3f; uploader: ring full %.3f, copy wait %.3f, file read %.3f", i + 1, I.m[i].gzs->launch_seconds(), a, b, c, d, e);
                              { double r, k, n; I.m[i].gzs->other_times(r, k, n); fprintf(stderr, "; giving back the buffers of linked slabs %.3f, CRC launch and results %.3f, all of the producer's steps %.3f", r, k, n); }
                              { double oa, os, ou; I.m[i].gzs->open_parts(oa, os, ou); fprintf(stderr, "; set-up %.3f (streams %.3f, uploader's buffers and thread %.3f)", oa, os, ou); }
                              double x, z; I.m[i].gzs->link_parts(x, z); fprintf(stderr, " (of the link time: text buffer %.3f of which waiting for the consumers to hand one back %.3f; waiting for the post stream to be made %.3f)", x, I.m[i].gzs->slot_seconds(), z); }
        for (int i = 0; i < I.nm; i++)
            if (I.m[i].gzs) fprintf(stderr, " | mate %d: inflate kernels busy %.3f s (%.1f GB/s of text), %llu of %u chunks of %zu KiB linked, %llu gaps bridged on the host, %llu bytes decoded there, ring %zu MiB, %u slab splits", i + 1, I.m[i].gzs->decode_busy_seconds(), I.m[i].gzs->decode_busy_seconds() > 0 ? (double)I.m[i].gzs->text_bytes() / I.m[i].gzs->decode_busy_seconds() / 1e9 : 0.0, (unsigned long long)I.m[i].gzs->chunks_linked(),
                                    I.m[i].gzs->chunks(), I.m[i].gzs->chunk_bytes() >> 10, (unsigned long long)I.m[i].gzs->gaps(), (unsigned long long)I.m[i].gzs->gap_bytes(), I.m[i].gzs->ring_bytes() >> 20, I.m[i].gzs->splits());
        fprintf(stderr, "\n");
        end_of_call.timing = true; end_of_call.t0 = now_s();
    }
    return MF_OK;
}

// file-level calls running in this process right now: what they hold in the caches (consumers' scratch and read sets, pinned staging) is in
// use, so a release of everything (mf_release_cached, an allocation elsewhere that found the device full) leaves those alone meanwhile
static std::atomic<int> g_calls_running{0};
struct CallRunning { CallRunning() { g_calls_running++; } ~CallRunning() { g_calls_running--; } };

int run_device_ingest(mf_kmerset *ks, const char *fq1, const char *fq2, const char *out1, const char *out2, uint32_t threshold,
                      bool pair_both, const int *devices, int n_devices, uint64_t *kept, uint64_t *total, std::string &err, IngestStats *stats)
{
    CallRunning running;
    Ingest I;
    I.ks = ks; I.threshold = threshold; I.pair_both = pair_both;
    for (int i = 0; i < n_devices; i++) I.devices.push_back(devices[i]);
    const int rc = run_ingest(I, fq1, fq2, out1, out2, err, stats);
    if (rc) return rc;
    if (kept) *kept = I.kept;
    if (total) *total = I.total;
    return MF_OK;
}

int run_device_qualfilter(const char *fq1, const char *fq2, const char *out1, const char *out2, const QualParams &P, int device, uint64_t *kept,
                          uint64_t *total, bool *panicked, std::string &err, IngestStats *stats)
{
    if (out1 && has_gz_ext(out1) && env_u64("MF_QUAL_DEVICE_GZ_OUT", 0) == 0) return MF_DEVINGEST_DECLINED;      // compressing the output is the host pipeline's (many threads)
    if (out2 && has_gz_ext(out2) && env_u64("MF_QUAL_DEVICE_GZ_OUT", 0) == 0) return MF_DEVINGEST_DECLINED;
    QualState Q;                      // (before the Ingest: its batches hold buffers the state does not own, but the set's go back to the pool last)
    Q.P = P; Q.pe = fq2 != nullptr; Q.cap = P.end ? P.end - P.start : ~0ull;
    Q.chunks.init((size_t)std::max<uint64_t>(env_u64("MF_QUAL_OUT_CHUNK", (uint64_t)4 << 20), 4096), (int)std::max<uint64_t>(2, env_u64("MF_QUAL_OUT_CHUNKS", 24)),
                  [](size_t n) -> void * { void *q = nullptr; return hipHostMalloc(&q, n, hipHostMallocPortable) == hipSuccess ? q : nullptr; }, [](void *q) { (void)hipHostFree(q); });
    CallRunning running;
    Ingest I;
    I.qual = &Q;
    I.devices.push_back(device);
    const int rc = run_ingest(I, fq1, fq2, out1, out2, err, stats);
    if (rc) return rc;
    if (kept) *kept = I.kept;
    if (total) *total = I.total;
    if (panicked) *panicked = Q.panicked;
    return MF_OK;
}

void ingest_short_lived(bool yes) { g_short_lived = yes; }

void ingest_prefetch(int device)
{
    const int dev = phys(device);
    int cur = -1; (void)hipGetDevice(&cur);
    if (hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); return; }
    std::string err;
    (void)g_streams.get(dev, false, err);
    g_streams.prefill_pinned(dev);
    if (cur >= 0 && cur != dev) (void)hipSetDevice(cur);
}

size_t release_cached_device_memory(bool all)
{
    // (the pool's free lists hold idle buffers only -- giving those back is always safe, if slow beside a running call: hipFree waits for the
    // device --; the scratch, read-set and pinned caches are taken apart only when no file-level call of this process is running)
    if (all) { if (g_calls_running.load() == 0) { g_scratch.clear(); g_streams.forget_staging(); g_pinned.clear(); } return g_pool.release_all(); }
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return g_pool.release(dev);
}

} // namespace mf
