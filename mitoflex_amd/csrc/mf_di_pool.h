// Device ingest path, part 1 of 8 (mf_devingest.cpp includes them in order; one translation unit): where its memory comes from --
// the device buffer pool, the mapped input file, pinned staging buffers and the threads that fill them, registered mappings.
#pragma once
#include "mf_di_base.h"

namespace mf {
namespace {

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
        // Nothing idle fits.  A call's buffers and what idles beside them stay within the call's limit (run_ingest: what the call may hold on the
        // device, which follows its input): a process that filters inputs of very different sizes in turn would otherwise carry the last call's
        // 8 GB of idle buffers beside this call's own (14.8 GB in use for a 0.6 GB pair behind a 5 GB file) -- the idle ones go first, largest first.
        {
            size_t keep = ~(size_t)0;
            {
                std::lock_guard<std::mutex> lk(mu_);
                PerDev &D = dev_[dev];
                if (D.limit && D.used + D.held + want > D.limit) { const size_t over = D.used + D.held + want - D.limit; keep = D.held > over ? D.held - over : 0; }
            }
            if (keep != ~(size_t)0) trim_dev(dev, keep, true);
        }
        const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
#ifdef MF_TEST_HOOKS
        // (test hook, libmitofilter_hip_hooks.so only: MF_DEVPOOL_FAIL_AT=n makes the n-th new allocation of the process fail as if the device were full -- the
        // call must then hand the input to the host pipeline, tests/test_gpu_devingest.py::test_a_failed_allocation_hands_the_call_to_the_host_pipeline)
        const uint64_t fail_at = g_knobs.u64(KN_DEVPOOL_FAIL_AT, 0);
        static std::atomic<uint64_t> n_new{0};
        if (fail_at && ++n_new >= fail_at) { *p = nullptr; *got = 0; return hipErrorOutOfMemory; }
#endif
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
    void set_limit(int dev, size_t bytes) { std::lock_guard<std::mutex> lk(mu_); dev_[dev].limit = bytes; }      // in use + idle on `dev` while a call runs (0: no limit)
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
    struct PerDev { std::multimap<size_t, void *> free_; size_t held = 0, used = 0, peak = 0, limit = 0; };
    void account(int dev, long long delta) { PerDev &D = dev_[dev]; D.used = (size_t)((long long)D.used + delta); if (D.used > D.peak) D.peak = D.used; }     // (mu_ held)
    void trim_dev(int dev, size_t keep, bool largest_first = false)
    {
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> lk(mu_);
            PerDev &D = dev_[dev];
            while (D.held > keep && !D.free_.empty()) { auto it = largest_first ? std::prev(D.free_.end()) : D.free_.begin(); drop.push_back(it->second); D.held -= it->first; D.free_.erase(it); }
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
        std::unique_lock<std::mutex> lk(mu_);
        const size_t n = size_[p];
        if (g_knobs.starts_0(KN_KEEP_BUFFERS) || free_.size() >= 8) { size_.erase(p); lk.unlock(); (void)hipHostFree(p); return; }
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
        nthr = (int)std::min<uint64_t>(16, std::max<uint64_t>(1, g_knobs.u64(KN_UPLOAD_THREADS, 8)));
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
        const bool off = g_knobs.is_set(KN_UPLOAD_STAGED);
        usable_ = !off && p && n && n <= (size_t)g_knobs.u64(KN_UPLOAD_REGISTER_MAX_MB, 512) << 20 && g_pinned.idle() == 0;
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

} // namespace
} // namespace mf
