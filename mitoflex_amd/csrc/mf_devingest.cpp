// Device ingest path of mf_filter_fastq_files (see mf_devingest.h).
//
// An uploader thread reads the file into pinned staging and copies it up; one producer thread per mate turns it into text in a
// contiguous arena -- a plain FASTQ file IS the text; a .gz is decoded in slabs of a few hundred speculative chunks (mf_gzdev.h):
// the decode kernels of up to twelve slabs ahead run on their own streams while slab k is linked and slab k - 1 is resolved and
// CRC-checked --; the calling thread is the consumer: it cuts the text into records where it lies (mf_ingest.h; what is behind
// the last complete record of a piece, the carry, is the front of the next piece's text), packs them behind what the whole-file
// read set already holds, runs ONE filter pass over the file when both mates are in, and copies the survivors out to a writer
// thread per output file.  Pass bits are a file-wide bitmap per mate, so the pair rule does not care that the pieces of the two
// mates do not cover the same records; the mate that is behind in records is advanced.
#include "mf_devingest.h"
#include "mf_api_internal.h"
#include "mf_gzdev.h"
#include "mf_host.h"
#include "mf_ingest.h"
#include "mf_pinflate.h"

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

#define DCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err = std::string(#call) + " failed: " + hipGetErrorString(e_); return MF_E_HIP; } } while (0)

uint64_t env_u64(const char *name, uint64_t dflt) { const char *v = getenv(name); return v && *v ? strtoull(v, nullptr, 10) : dflt; }
static const bool g_trace = getenv("MF_DEVINGEST_TRACE") != nullptr;
#define TRACE(...) do { if (g_trace) { fprintf(stderr, "[devingest %.3f] ", now_s() - (double)(long)now_s() + ((long)now_s() % 1000)); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)
double now_s();
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Device memory of this path comes from a pool that outlives the call.  Two reasons.  hipFree waits for the whole device to go
// idle -- with decode kernels in flight on other streams that is tens of milliseconds a call -- so nothing is freed while a
// file is being processed: outgrown buffers go back to the pool.  And allocating (and later releasing) the tens of gigabytes a
// large file takes costs more than a second, which a caller that filters file after file (the bim loop) would pay every
// time: a call's buffers are kept for the next one, up to MF_DEVPOOL_GB (default 96; MF_KEEP_BUFFERS=0: nothing is kept).
class DevPool {
public:
    static size_t round_up(size_t bytes)
    {
        size_t unit = (size_t)1 << 20;
        while (unit * 16 < bytes && unit < ((size_t)256 << 20)) unit <<= 1;         // 1 MiB steps for small blocks, up to 256 MiB steps
        return (bytes + unit - 1) / unit * unit;
    }
    hipError_t get(void **p, size_t bytes, size_t *got)
    {
        const size_t want = round_up(bytes ? bytes : 1);
        {
            std::lock_guard<std::mutex> lk(mu_);
            auto it = free_.lower_bound(want);
            if (it != free_.end() && it->first <= want + want / 2 + ((size_t)64 << 20)) { *p = it->second; *got = it->first; held_ -= it->first; free_.erase(it); return hipSuccess; }
        }
        hipError_t e = hipMalloc(p, want);
        if (e != hipSuccess) {                      // make room: release what the pool holds and try once more
            (void)hipGetLastError();
            trim(0);
            e = hipMalloc(p, want);
        }
        *got = want;
        return e;
    }
    void put(void *p, size_t bytes) { if (!p) return; std::lock_guard<std::mutex> lk(mu_); free_.emplace(bytes, p); held_ += bytes; }
    size_t held() { std::lock_guard<std::mutex> lk(mu_); return held_; }          // bytes waiting for the next call
    void trim(size_t keep)                          // (only when no kernel of this path is in flight)
    {
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> lk(mu_);
            while (held_ > keep && !free_.empty()) { auto it = free_.begin(); drop.push_back(it->second); held_ -= it->first; free_.erase(it); }
        }
        for (void *q : drop) (void)hipFree(q);
    }
private:
    std::mutex mu_; std::multimap<size_t, void *> free_; size_t held_ = 0;
};
DevPool g_pool;
// (buffers that an mf_reads owns are hipMalloc'ed; outgrown ones are parked here and freed when the call is over)
struct Trash {
    std::mutex mu; std::vector<void *> v;
    void add(void *p) { if (p) { std::lock_guard<std::mutex> lk(mu); v.push_back(p); } }
    void empty() { std::lock_guard<std::mutex> lk(mu); for (void *p : v) (void)hipFree(p); v.clear(); }
};
Trash g_trash;

template <class T> struct DevBuf {
    T *p = nullptr; size_t cap = 0;                      // cap in elements
    size_t bytes_ = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete; DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { g_pool.put(p, bytes_); }
    hipError_t need(size_t n, bool slack = true)          // contents are NOT kept
    {
        if (n <= cap && p) return hipSuccess;
        g_pool.put(p, bytes_); p = nullptr; cap = 0; bytes_ = 0;
        const size_t want = slack ? n + n / 2 + 1024 : (n ? n : 1);
        void *q = nullptr; size_t got = 0;
        hipError_t e = g_pool.get(&q, want * sizeof(T), &got);
        if (e == hipSuccess) { p = (T *)q; bytes_ = got; cap = got / sizeof(T); }
        return e;
    }
};

struct Mapped {
    const uint8_t *p = nullptr; size_t n = 0; int fd = -1;          // (the descriptor stays open: the uploader reads through it)
    ~Mapped() { if (p) munmap(const_cast<uint8_t *>(p), n); if (fd >= 0) ::close(fd); }
    bool open(const char *path, bool &regular)
    {
        regular = false;
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) return true;
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

// ---- the file's bytes -> device memory, in order, on a copy stream of its own.  Consumers make their stream wait for the
// event of the piece that completes the range they read.
class Uploader {
public:
    static constexpr size_t PIECE = (size_t)32 << 20;
    ~Uploader()
    {
        TRACE("~Uploader");
        stop_ = true;
        if (th_.joinable()) th_.join();
        TRACE("uploader joined");
        for (auto &e : ev_) if (e) (void)hipEventDestroy(e);
        for (auto &e : free_ev_) if (e) (void)hipEventDestroy(e);
        for (auto &s : stage_) if (s) (void)hipHostFree(s);
        if (st_) (void)hipStreamDestroy(st_);
    }
    int start(const uint8_t *src, int fd, size_t n, uint8_t *dst, int device, std::string &err)
    {
        src_ = src; fd_ = fd; n_ = n; dst_ = dst; device_ = device;
        const size_t np = (n + PIECE - 1) / PIECE;
        ev_.assign(np, nullptr);
        for (auto &e : ev_) DCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        DCHK(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            DCHK(hipHostMalloc((void **)&stage_[i], PIECE, hipHostMallocDefault));
            DCHK(hipEventCreateWithFlags(&free_ev_[i], hipEventDisableTiming));
        }
        th_ = std::thread([this] { run(); });
        return MF_OK;
    }
    // the copy of bytes [0, upto) has been issued (so wait_for would not block the host)
    bool issued(size_t upto)
    {
        if (n_ == 0 || upto == 0) return true;
        if (upto > n_) upto = n_;
        std::lock_guard<std::mutex> lk(mu_);
        return failed_ || enqueued_ > (upto - 1) / PIECE;
    }
    // make `st` wait until bytes [0, upto) are on the device (upto is clamped to the file).  false: the uploader failed
    bool wait_for(hipStream_t st, size_t upto)
    {
        if (n_ == 0 || upto == 0) return true;
        if (upto > n_) upto = n_;
        const size_t j = (upto - 1) / PIECE;
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return enqueued_ > j || failed_; });
        if (failed_) return false;
        return hipStreamWaitEvent(st, ev_[j], 0) == hipSuccess;
    }
private:
    void run()
    {
        if (hipSetDevice(device_) != hipSuccess) { fail_(); return; }
        const size_t np = ev_.size();
        const int nthr = (int)std::min<uint64_t>(16, std::max<uint64_t>(1, env_u64("MF_UPLOAD_THREADS", 8)));
        for (size_t i = 0; i < np && !stop_; i++) {
            const int b = (int)(i & 1);
            if (i >= 2 && hipEventSynchronize(free_ev_[b]) != hipSuccess) { fail_(); return; }       // the copy that read this staging buffer is done
            const size_t off = i * PIECE, len = std::min(PIECE, n_ - off);
            {   // page cache -> pinned memory on a few threads, with pread: reading the mapping instead takes a fault per 64 KiB and
                // does 3 GB/s a thread, and with four of those the whole path ran at the 11 GB/s of this copy (a 4.9 GB .gz in 0.44 s
                // whatever the decoder did); the mapping stays for what the host looks at (headers, trailers, gaps)
                auto part = [&](int t) {
                    size_t a = len * t / nthr; const size_t e = len * (t + 1) / nthr;
                    while (a < e) {
                        const ssize_t got = fd_ >= 0 ? pread(fd_, stage_[b] + a, e - a, (off_t)(off + a)) : -1;
                        if (got <= 0) { if (got < 0 && errno == EINTR) continue; memcpy(stage_[b] + a, src_ + off + a, e - a); break; }     // (a file that cannot be read this way: through the mapping)
                        a += (size_t)got;
                    }
                };
                std::vector<std::thread> th;
                for (int t = 1; t < nthr; t++) th.emplace_back(part, t);
                part(0);
                for (auto &x : th) x.join();
            }
            if (hipMemcpyAsync(dst_ + off, stage_[b], len, hipMemcpyHostToDevice, st_) != hipSuccess || hipEventRecord(ev_[i], st_) != hipSuccess ||
                hipEventRecord(free_ev_[b], st_) != hipSuccess) { fail_(); return; }
            { std::lock_guard<std::mutex> lk(mu_); enqueued_ = i + 1; }
            cv_.notify_all();
        }
        (void)hipStreamSynchronize(st_);
    }
    void fail_() { { std::lock_guard<std::mutex> lk(mu_); failed_ = true; } cv_.notify_all(); }
    const uint8_t *src_ = nullptr; int fd_ = -1; size_t n_ = 0; uint8_t *dst_ = nullptr; int device_ = 0;
    hipStream_t st_ = nullptr; uint8_t *stage_[2] = {nullptr, nullptr}; hipEvent_t free_ev_[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> ev_;
    std::thread th_; std::mutex mu_; std::condition_variable cv_; size_t enqueued_ = 0; bool failed_ = false; std::atomic<bool> stop_{false};
};

// ---- the decoder's streams.  The link step is the decoder's one serial path, and its workgroup wants 64 KiB of LDS -- on a chip
// whose LDS the decode wavefronts of the slabs ahead have filled it would wait tens of milliseconds for a CU to drain.  So a few
// CUs (one per XCD: mask bit b is a CU of XCD b mod 8) are kept free of decode work: the decode streams are masked off them, the
// link stream runs only there.  Where CU masks are not to be had, ordinary streams.  The sets are made once and handed from
// call to call: destroying a CU-masked stream right after use was seen to hang inside the runtime (ROCm 7.2), and they cost a
// few milliseconds to make.
constexpr uint32_t GZ_NSTREAM = 10;
struct StreamSet { int device = -1; hipStream_t sd[GZ_NSTREAM] = {}, link = nullptr, rest = nullptr; };
class StreamSets {
public:
    StreamSet *take(int device, std::string &err)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (size_t i = 0; i < free_.size(); i++) if (free_[i]->device == device) { StreamSet *s = free_[i]; free_.erase(free_.begin() + (long)i); return s; }
        }
        std::unique_ptr<StreamSet> s(new StreamSet());
        s->device = device;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) { err = "hipGetDeviceProperties failed"; return nullptr; }
        const int n_cu = prop.multiProcessorCount, words = (n_cu + 31) / 32;
        std::vector<uint32_t> m_dec((size_t)words, 0), m_link((size_t)words, 0);
        // A decode wavefront holds 128 vector registers and 9.9 KB of LDS for tens of milliseconds, sixteen of them fill a CU to the
        // last register: whatever else has to run meanwhile -- the link step, marker resolution, CRC, the consumer's line index and
        // pack kernels, all short and all on some host thread's critical path -- needs CUs of its own.  Four per XCD by default.
        int reserve = (int)env_u64("MF_GZDEV_RESERVED_CUS", 32);
        reserve = std::max(8, std::min(n_cu / 2, reserve)) & ~7;
        for (int b = 0; b < n_cu; b++) (b >= n_cu - reserve ? m_link : m_dec)[b / 32] |= 1u << (b % 32);
        const bool masks = n_cu >= 64 && !getenv("MF_GZDEV_NO_CUMASK");
        bool ok = true;
        for (auto &q : s->sd)
            if (!masks || hipExtStreamCreateWithCUMask(&q, (uint32_t)words, m_dec.data()) != hipSuccess) { (void)hipGetLastError(); ok = ok && hipStreamCreateWithFlags(&q, hipStreamNonBlocking) == hipSuccess; }
        if (!masks || hipExtStreamCreateWithCUMask(&s->link, (uint32_t)words, m_link.data()) != hipSuccess) { (void)hipGetLastError(); ok = ok && hipStreamCreateWithFlags(&s->link, hipStreamNonBlocking) == hipSuccess; }
        ok = ok && hipStreamCreateWithFlags(&s->rest, hipStreamNonBlocking) == hipSuccess;
        if (!ok) { err = "hipStreamCreate failed"; return nullptr; }           // (what was made stays behind: never destroyed, see above)
        return s.release();
    }
    void give(StreamSet *s) { if (s) { std::lock_guard<std::mutex> lk(mu_); free_.push_back(s); } }
private:
    std::mutex mu_; std::vector<StreamSet *> free_;
};
StreamSets g_streams;

// ---- the text of one input file on the device: ONE contiguous arena per mate, so that a record that spans two slabs, or the
// 32 KiB deflate window in front of a chunk, is simply the bytes in front.  A plain file's arena is the uploaded file; a .gz's
// grows (rarely) as the text does.  The producer thread writes behind `ready`, the consumer reads in front of it.
struct Arena {
    static constexpr size_t FRONT = 32768 + 256; // readable bytes in front of p: a damaged stream may point a full window back from its first byte
    uint8_t *p = nullptr; size_t cap = 0;       // cap bytes usable (+ 64 readable behind)
    uint8_t *raw = nullptr; size_t raw_bytes = 0; // the allocation, when the arena owns one
    std::mutex mu;                               // held by the consumer while it enqueues kernels that read the arena, by the producer while it moves it
    hipEvent_t read_ev = nullptr; bool read_pending = false;      // behind the consumer's last kernels that read it: the producer waits for it before it moves the arena
    ~Arena() { if (read_ev) (void)hipEventDestroy(read_ev); g_pool.put(raw, raw_bytes); }
};

// a range of an input's text that has become available, in order
struct TextPiece { uint64_t T0 = 0, len = 0; bool last = false; uint64_t est_total = 0; };      // est_total: the file's whole text, as far as one can tell now

// ---- one gzip file decoded on the device (runs on the mate's producer thread, on its own streams)
class GzStream {
public:
    ~GzStream()
    {
        TRACE("~GzStream");
        for (auto &s : sym_) { if (s.ev) (void)hipEventDestroy(s.ev); }
        if (streams_) {                           // nothing of this decoder may be in flight when its buffers go back to the pool
            for (auto &s : sd_) if (s) (void)hipStreamSynchronize(s);
            (void)hipStreamSynchronize(sp_); (void)hipStreamSynchronize(sr_);
            g_streams.give(streams_);
        }
        if (ev_link_) (void)hipEventDestroy(ev_link_);
        TRACE("streams destroyed");
        if (h_chain_) (void)hipHostFree(h_chain_);
        if (h_crc_) (void)hipHostFree(h_crc_);
        TRACE("~GzStream done");
    }
    // data: the mapped file; d_file: its copy on the device (size + 64 readable, being filled by `up`)
    int open(const uint8_t *data, size_t size, uint8_t *d_file, Uploader *up, Arena *arena, const std::string &path, std::string &err)
    {
        data_ = data; size_ = size; d_file_ = d_file; up_ = up; arena_ = arena; path_ = path;
        // chunks: large enough that the serial link step (a fixed cost per chunk) stays small, small enough that a file keeps the chip busy
        size_t dflt = size / 8192; dflt = std::min<size_t>(std::max<size_t>(dflt, (size_t)64 << 10), (size_t)256 << 10) & ~(size_t)4095;
        chunk_ = (size_t)env_u64("MF_GZDEV_CHUNK_BYTES", dflt);
        if (chunk_ < 1024) chunk_ = 1024;
        cps_ = (uint32_t)env_u64("MF_GZDEV_SLAB_CHUNKS", std::max<uint64_t>(256, ((uint64_t)128 << 20) / chunk_));
        if (cps_ < 1) cps_ = 1;
        expand_ = env_u64("MF_GZDEV_EXPAND", 8);
        size_t pos = 0;
        if (!member_header(pos, err)) return MF_E_FORMAT;
        base_byte_ = pos;
        n_chunks_ = (uint32_t)((size_ - base_byte_ + chunk_ - 1) / chunk_);
        if (n_chunks_ == 0) n_chunks_ = 1;
        if (cps_ > n_chunks_) cps_ = n_chunks_;
        n_slabs_ = (n_chunks_ + cps_ - 1) / cps_;
        sym_cap_ = chunk_ * expand_ + 262144;
        DCHK(d_chunks_.need(n_chunks_, false)); DCHK(d_out_off_.need(n_chunks_, false)); DCHK(d_chain_.need(1, false));
        {
            int dev = 0;
            DCHK(hipGetDevice(&dev));
            streams_ = g_streams.take(dev, err);
            if (!streams_) return MF_E_HIP;
            for (uint32_t i = 0; i < NSTREAM; i++) sd_[i] = streams_->sd[i];
            sp_ = streams_->link; sr_ = streams_->rest;
            DCHK(hipEventCreateWithFlags(&ev_link_, hipEventDisableTiming));
        }
        // (never the null stream: the CU-masked streams are blocking ones, a copy on the null stream would wait for every decode
        // kernel in flight -- of the other mate's file too)
        DCHK(hipMemsetAsync(d_chunks_.p, 0, n_chunks_ * sizeof(GzChunk), sr_));
        DCHK(hipMemsetAsync(d_out_off_.p, 0xFF, n_chunks_ * sizeof(uint64_t), sr_));
        DCHK(hipHostMalloc((void **)&h_chain_, sizeof(GzChain), hipHostMallocDefault));
        memset(h_chain_, 0, sizeof(GzChain));
        h_chain_->cur_bit = (uint64_t)base_byte_ * 8;
        DCHK(hipMemcpyAsync(d_chain_.p, h_chain_, sizeof(GzChain), hipMemcpyHostToDevice, sr_));
        DCHK(hipStreamSynchronize(sr_));
        for (auto &s : sym_) { DCHK(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming)); s.slab = ~0u; }
        h_chunks_.resize(n_chunks_);
        // the arena: a first guess at the size of the text (FASTQ compresses three- to fivefold); it is moved when it proves too small
        const int rc = grow_arena(std::max<size_t>(size_ * 4, (size_t)64 << 20), err);
        if (rc) return rc;
        in_member_ = true;
        return MF_OK;
    }
    // text of the next slab (possibly nothing) is in the arena when this returns
    // Marker resolution and the CRC of slab k run while slab k + 1 is waited for and linked (they are the link step's only
    // neighbours on the producer's critical path: 1.8 ms of 5.5 per slab), so the piece this returns is the one BEFORE the slab it
    // has just linked -- nothing the first time, the last piece in a call of its own.
    int next(TextPiece &out, std::string &err)
    {
        hipStream_t sp = sp_;
        out = TextPiece();
        if (done_ || next_slab_ >= n_slabs_) return finish_pending(out, err);
        const uint32_t k = next_slab_;
        // decode runs ahead of the text: this slab (waiting for its bytes if need be) and as many of the following ones as there are
        // symbol buffers and uploaded bytes for
        for (uint32_t j = k; j < k + NSYM && j < n_slabs_; j++) { const int rc = launch_decode(j, j == k, err); if (rc) return rc; }
        const uint32_t lo = k * cps_, hi = std::min(n_chunks_, lo + cps_);
        Sym &S = sym_[k % NSYM];
        TRACE("slab %u: chunks %u..%u, waiting for decode", k, lo, hi);
        DCHK(hipStreamWaitEvent(sp, S.ev, 0));
        DCHK(hipMemcpyAsync(h_chunks_.data() + lo, d_chunks_.p + lo, (hi - lo) * sizeof(GzChunk), hipMemcpyDeviceToHost, sp));
        DCHK(hipStreamSynchronize(sp));
        uint64_t sum = 0; uint32_t max_sym = 0;
        for (;;) {
            bool overflow = false;
            sum = 0; max_sym = 0;
            for (uint32_t c = lo; c < hi; c++) {
                const GzChunk &ch = h_chunks_[c];
                if (ch.status == GZ_OVERFLOW) overflow = true;
                if (ch.status == GZ_AT_BOUNDARY || ch.status == GZ_MEMBER_END) { sum += ch.n_sym; max_sym = std::max(max_sym, ch.n_sym); }
            }
            TRACE("slab %u decoded: %llu symbols, overflow %d", k, (unsigned long long)sum, (int)overflow);
            if (!overflow) break;
            // text that expands more than the symbol buffers allow for (a run of identical reads, say): this slab again, with four times the room
            if (S.cap > chunk_ * 2048) { err = "gzip data in " + path_ + " expands more than a thousandfold: not decoded on the device"; return MF_E_FORMAT; }
            S.cap *= 4;
            DCHK(S.p.need((size_t)(hi - lo) * S.cap, false));
            DCHK(launch_gz_decode(d_file_, size_, base_byte_, chunk_, lo, hi - lo, 0, (uint64_t)base_byte_ * 8, S.p.p, S.cap, d_chunks_.p, sd_[0]));
            DCHK(hipMemcpyAsync(h_chunks_.data() + lo, d_chunks_.p + lo, (hi - lo) * sizeof(GzChunk), hipMemcpyDeviceToHost, sd_[0]));
            DCHK(hipStreamSynchronize(sd_[0]));
        }
        const uint64_t T0 = h_chain_->total;
        int rc = grow_arena(T0 + sum + ((size_t)1 << 20), err);
        if (rc) return rc;
        const bool last_slab = k + 1 == n_slabs_;
        // link; the host steps in where the chain stops
        for (bool first = true;; first = false) {
            if (!done_ && in_member_) {
                DCHK(launch_gz_chain(d_chain_.p, d_chunks_.p, lo, hi, S.p.p, S.cap, d_out_off_.p, arena_->p, 0, sp));
                DCHK(hipMemcpyAsync(h_chain_, d_chain_.p, offsetof(GzChain, window), hipMemcpyDeviceToHost, sp));
            }
            if (first) { rc = finish_pending(out, err); if (rc) return rc; }       // (the slab before: its resolve and CRC kernels ran beside this slab's decode wait and link)
            if (!done_ && in_member_) DCHK(hipStreamSynchronize(sp));
            TRACE("chain: stop %u next %u cur_bit %llu total %llu linked %u", h_chain_->stop, h_chain_->next, (unsigned long long)h_chain_->cur_bit, (unsigned long long)h_chain_->total, h_chain_->linked);
            if (done_) break;
            uint64_t to_bit = 0;
            if (h_chain_->stop == GZ_STOP_MEMBER_END) { rc = member_end(max_sym, sp, err); if (rc) return rc; if (done_) break; }
            if (h_chain_->stop == GZ_STOP_GAP) to_bit = h_chunks_[h_chain_->next].start_bit;
            else if (h_chain_->stop == GZ_STOP_NONE) {
                if (!last_slab) break;
                to_bit = (uint64_t)size_ * 8;             // behind the last chunk: the host decodes to the end of the member
            }
            // ---- decode across the gap on the host, with the window the chain left
            DCHK(hipMemcpyAsync(h_chain_->window, d_chain_.p->window, GZ_WINDOW, hipMemcpyDeviceToHost, sp)); DCHK(hipStreamSynchronize(sp));
            std::vector<uint8_t> bytes; uint64_t end_bit = 0; bool mend = false; std::string why;
            if (!inflate_gap(data_, size_, h_chain_->cur_bit, to_bit, h_chain_->window, h_chain_->wlen, bytes, end_bit, mend, why)) {
                err = "gzip read error in " + path_ + ": " + why; return MF_E_FORMAT;
            }
            gap_bytes_ += bytes.size(); n_gaps_++;
            TRACE("gap: %zu bytes, ends at bit %llu (wanted %llu), member end %d", bytes.size(), (unsigned long long)end_bit, (unsigned long long)to_bit, (int)mend);
            rc = grow_arena(h_chain_->total + bytes.size() + sum + ((size_t)1 << 20), err);
            if (rc) return rc;
            if (!bytes.empty()) { DCHK(hipMemcpyAsync(arena_->p + h_chain_->total, bytes.data(), bytes.size(), hipMemcpyHostToDevice, sp)); DCHK(hipStreamSynchronize(sp)); }
            // the window behind the gap
            if (bytes.size() >= GZ_WINDOW) { memcpy(h_chain_->window, bytes.data() + bytes.size() - GZ_WINDOW, GZ_WINDOW); h_chain_->wlen = GZ_WINDOW; }
            else {
                const size_t keep = std::min<size_t>(h_chain_->wlen, GZ_WINDOW - bytes.size());
                memmove(h_chain_->window + GZ_WINDOW - keep - bytes.size(), h_chain_->window + GZ_WINDOW - keep, keep);
                memcpy(h_chain_->window + GZ_WINDOW - bytes.size(), bytes.data(), bytes.size());
                h_chain_->wlen = (uint32_t)(keep + bytes.size());
            }
            h_chain_->cur_bit = end_bit; h_chain_->total += bytes.size();
            if (h_chain_->stop == GZ_STOP_NONE && last_slab && !mend && bytes.empty()) { err = "gzip read error in " + path_ + ": truncated deflate stream"; return MF_E_FORMAT; }
            h_chain_->stop = mend ? GZ_STOP_MEMBER_END : GZ_STOP_NONE;
            DCHK(hipMemcpyAsync(d_chain_.p, h_chain_, sizeof(GzChain), hipMemcpyHostToDevice, sp)); DCHK(hipStreamSynchronize(sp));
            if (mend) { rc = member_end(max_sym, sp, err); if (rc) return rc; if (done_) break; }
        }
        DCHK(hipEventRecord(ev_link_, sp)); DCHK(hipStreamWaitEvent(sr_, ev_link_, 0));
        DCHK(launch_gz_resolve(d_chunks_.p, lo, hi, S.p.p, S.cap, d_out_off_.p, arena_->p, 0, max_sym, sr_));
        // the rest of the member's CRC over this slab: launched here, taken in by finish_pending
        if (h_chain_->total > crc_done_) { rc = crc_launch(crc_done_, h_chain_->total, sr_, err); if (rc) return rc; }
        next_slab_ = k + 1;
        if (next_slab_ == n_slabs_ && !done_) { err = "gzip read error in " + path_ + ": unexpected end of file"; return MF_E_FORMAT; }   // the data ran out inside a member
        pend_.T0 = T0; pend_.len = h_chain_->total - T0; pend_.last = done_;
        {   // the text so far over the compressed bytes it came from, times the file
            const double in = (double)std::min<uint64_t>(size_, base_byte_ + (uint64_t)hi * chunk_);
            pend_.est_total = done_ ? h_chain_->total : (uint64_t)((double)h_chain_->total * ((double)size_ / std::max(1.0, in)) * 1.03);
        }
        pending_ = true; pend_sym_ = k % NSYM;
        return MF_OK;
    }
    // the slab whose resolve and CRC kernels are in flight becomes text: its piece goes to `out`
    int finish_pending(TextPiece &out, std::string &err)
    {
        if (!pending_) return MF_OK;
        DCHK(hipStreamSynchronize(sr_));
        crc_finish();
        sym_[pend_sym_].slab = ~0u;                        // (its symbols are text now)
        out = pend_; pending_ = false;
        return MF_OK;
    }
    uint64_t gap_bytes() const { return gap_bytes_; }
    uint64_t gaps() const { return n_gaps_; }
    uint64_t chunks_linked() const { return h_chain_ ? h_chain_->linked : 0; }
    uint32_t chunks() const { return n_chunks_; }
    size_t chunk_bytes() const { return chunk_; }
private:
    struct Sym { DevBuf<uint16_t> p; size_t cap = 0; hipEvent_t ev = nullptr; uint32_t slab = ~0u; };     // cap: symbols of room per chunk
    // the arena holds at least `need` bytes (what is in it moves along)
    int grow_arena(size_t need, std::string &err)
    {
        if (need <= arena_->cap) return MF_OK;
        const size_t cap = std::max(need + need / 2, (size_t)64 << 20);
        uint8_t *raw = nullptr; size_t raw_bytes = 0;
        DCHK(g_pool.get((void **)&raw, Arena::FRONT + cap + 64, &raw_bytes));
        DCHK(hipMemsetAsync(raw, 0, Arena::FRONT, sr_)); DCHK(hipStreamSynchronize(sr_));
        uint8_t *p = raw + Arena::FRONT;
        std::lock_guard<std::mutex> lk(arena_->mu);
        if (arena_->read_pending) { DCHK(hipEventSynchronize(arena_->read_ev)); arena_->read_pending = false; }      // (no kernel of the consumer is reading the old one)
        const uint64_t have = h_chain_ ? h_chain_->total : 0;
        if (arena_->raw) {
            if (have) { DCHK(hipMemcpyAsync(p, arena_->p, have, hipMemcpyDeviceToDevice, sr_)); DCHK(hipStreamSynchronize(sr_)); }
            g_pool.put(arena_->raw, arena_->raw_bytes);       // (link, resolve and CRC of this stream are between slabs here; the consumer is held off by the lock)
        }
        arena_->raw = raw; arena_->raw_bytes = raw_bytes; arena_->p = p; arena_->cap = raw_bytes - Arena::FRONT - 64;
        return MF_OK;
    }
    int launch_decode(uint32_t j, bool must, std::string &err)
    {
        Sym &S = sym_[j % NSYM];
        if (j != launched_ || S.slab != ~0u) return MF_OK;     // launched already -- or its buffer still holds an earlier slab: launched when that one has become text
        const uint32_t lo = j * cps_, hi = std::min(n_chunks_, lo + cps_);
        hipStream_t st = sd_[j % NSTREAM];
        // the chunks read past their own range up to the end of a block, and the reader's ring a little further
        const size_t upto = base_byte_ + (size_t)hi * chunk_ + ((size_t)8 << 20);
        if (!must && !up_->issued(upto)) return MF_OK;
        S.cap = sym_cap_;
        DCHK(S.p.need((size_t)(hi - lo) * S.cap, false));
        if (!up_->wait_for(st, upto)) { err = "upload of " + path_ + " failed"; return MF_E_HIP; }
        DCHK(launch_gz_decode(d_file_, size_, base_byte_, chunk_, lo, hi - lo, 0, (uint64_t)base_byte_ * 8, S.p.p, S.cap, d_chunks_.p, st));
        DCHK(hipEventRecord(S.ev, st));
        S.slab = j; launched_ = j + 1;
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
    // the chain stands behind the final block of a member: check the trailer, look for another member
    int member_end(uint32_t max_sym, hipStream_t sp, std::string &err)
    {
        const size_t pos = (size_t)((h_chain_->cur_bit + 7) >> 3);
        if (pos + 8 > size_) { err = "gzip read error in " + path_ + ": truncated gzip trailer"; return MF_E_FORMAT; }
        uint32_t want_crc, want_len; memcpy(&want_crc, data_ + pos, 4); memcpy(&want_len, data_ + pos + 4, 4);
        // CRC of the member's text up to here (the resolve kernel has not run yet for this slab: do it for what is accepted)
        {
            const uint32_t k = next_slab_, lo = k * cps_, hi = std::min(n_chunks_, lo + cps_);
            DCHK(hipEventRecord(ev_link_, sp)); DCHK(hipStreamWaitEvent(sr_, ev_link_, 0));
            DCHK(launch_gz_resolve(d_chunks_.p, lo, hi, sym_[k % NSYM].p.p, sym_[k % NSYM].cap, d_out_off_.p, arena_->p, 0, max_sym, sr_));
        }
        if (h_chain_->total > crc_done_) { const int rc = crc_over(crc_done_, h_chain_->total, sr_, err); if (rc) return rc; }
        DCHK(hipStreamSynchronize(sr_));
        TRACE("member end: crc %08x want %08x", crc_, want_crc);
        if (crc_ != want_crc) { err = "gzip read error in " + path_ + ": incorrect data check"; return MF_E_FORMAT; }
        if ((uint32_t)(h_chain_->total - member_T0_) != want_len) { err = "gzip read error in " + path_ + ": incorrect length check"; return MF_E_FORMAT; }
        crc_ = 0; member_T0_ = h_chain_->total;
        size_t p = pos + 8;
        if (p >= size_ || size_ - p < 2 || data_[p] != 0x1f || data_[p + 1] != 0x8b) { done_ = true; in_member_ = false; return MF_OK; }   // trailing bytes that are no member: ignored
        if (!member_header(p, err)) return MF_E_FORMAT;
        h_chain_->cur_bit = (uint64_t)p * 8; h_chain_->wlen = 0; h_chain_->stop = GZ_STOP_NONE;
        DCHK(hipMemcpyAsync(d_chain_.p, h_chain_, offsetof(GzChain, window), hipMemcpyHostToDevice, sp)); DCHK(hipStreamSynchronize(sp));
        return MF_OK;
    }
    // running CRC of the member over the text [from, to): the kernel and the copy of its piece CRCs (crc_launch), the combination on the host (crc_finish)
    int crc_launch(uint64_t from, uint64_t to, hipStream_t sp, std::string &err)
    {
        if (crc_n_) { DCHK(hipStreamSynchronize(sp)); crc_finish(); }        // (an earlier launch on this stream that nobody has taken in)
        const uint64_t n = to - from;
        const size_t np = (size_t)((n + GZ_CRC_PIECE - 1) / GZ_CRC_PIECE);
        DCHK(d_crc_.need(np));
        if (np > h_crc_cap_) { if (h_crc_) (void)hipHostFree(h_crc_); h_crc_ = nullptr; h_crc_cap_ = 0; DCHK(hipHostMalloc((void **)&h_crc_, (np + np / 2 + 64) * 4, hipHostMallocDefault)); h_crc_cap_ = np + np / 2 + 64; }
        TRACE("crc over %llu bytes", (unsigned long long)n);
        DCHK(launch_gz_crc(arena_->p + from, n, d_crc_.p, sp));
        DCHK(hipMemcpyAsync(h_crc_, d_crc_.p, np * 4, hipMemcpyDeviceToHost, sp));
        crc_n_ = n; crc_done_ = to;
        return MF_OK;
    }
    void crc_finish() { if (crc_n_) { crc_ = gz_crc_combine(crc_, gz_crc_finish(h_crc_, crc_n_), crc_n_); crc_n_ = 0; } }      // (the stream of crc_launch has been synchronised)
    int crc_over(uint64_t from, uint64_t to, hipStream_t sp, std::string &err)
    {
        const int rc = crc_launch(from, to, sp, err); if (rc) return rc;
        DCHK(hipStreamSynchronize(sp));
        crc_finish();
        return MF_OK;
    }

    const uint8_t *data_ = nullptr; size_t size_ = 0; uint8_t *d_file_ = nullptr; Uploader *up_ = nullptr; Arena *arena_ = nullptr; std::string path_;
    size_t chunk_ = 0, base_byte_ = 0, sym_cap_ = 0; uint64_t expand_ = 8;
    uint32_t cps_ = 0, n_chunks_ = 0, n_slabs_ = 0, next_slab_ = 0, launched_ = 0;
    DevBuf<GzChunk> d_chunks_; DevBuf<uint64_t> d_out_off_; DevBuf<GzChain> d_chain_; DevBuf<uint32_t> d_crc_;
    GzChain *h_chain_ = nullptr; std::vector<GzChunk> h_chunks_; uint32_t *h_crc_ = nullptr; size_t h_crc_cap_ = 0; uint64_t crc_n_ = 0;      // h_crc_: pinned
    TextPiece pend_; bool pending_ = false; uint32_t pend_sym_ = 0;
    static constexpr uint32_t NSYM = 12, NSTREAM = GZ_NSTREAM;      // decode kernels in flight: enough wavefronts to fill the chip (a slab is a few hundred chunks)
    Sym sym_[NSYM]; StreamSet *streams_ = nullptr; hipStream_t sd_[NSTREAM] = {}, sp_ = nullptr, sr_ = nullptr; hipEvent_t ev_link_ = nullptr;     // sp_: the link stream (reserved CUs); sr_: resolve and CRC (the whole chip)
    bool in_member_ = false, done_ = false;
    uint32_t crc_ = 0; uint64_t crc_done_ = 0, member_T0_ = 0, gap_bytes_ = 0, n_gaps_ = 0;
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

// records of one piece of text, cut where they lie
struct Batch {
    uint64_t start = 0;                  // arena offset of the first record's header (the piece's T0 less the carry)
    DevBuf<uint64_t> line_start;         // offsets from `start`
    uint64_t n_rec = 0, rec_base = 0;
};

struct Mate {
    std::string path; Mapped map; bool gz = false;
    DevBuf<uint8_t> d_file; Uploader up; Arena arena; std::unique_ptr<GzStream> gzs;
    // producer: text pieces in order (a plain file: one piece per uploaded slab)
    std::thread prod; std::mutex mu; std::condition_variable cv; std::deque<TextPiece> ready; int prod_rc = MF_OK; std::string prod_err; bool prod_done = false;
    std::atomic<bool> stop{false};
    // consumer
    uint64_t carry = 0, rec_done = 0, bases = 0, n_npos = 0; bool eof = false;
    uint32_t min_len = ~0u, max_len = 0;
    std::vector<std::unique_ptr<Batch>> batches;
    mf_reads *reads = nullptr;           // the read set of the whole file, appended to batch by batch
    Writer out;
    DevBuf<uint32_t> tile_cnt, seq_len, inv_cnt, out_len, minmax; DevBuf<uint64_t> tile_base, scan_tmp, inv_base, out_off, offsets_tmp; DevBuf<uint8_t> d_out;
    // small results the host waits for (counts that size the next buffers), in pinned memory: a copy to pageable memory is a
    // synchronisation of its own.  [0] newlines [1] last byte [2] used [3] bases [4] min/max length [5] invalid bases [6] output bytes
    uint64_t *h_small = nullptr;
    ~Mate() { TRACE("~Mate"); stop = true; if (prod.joinable()) prod.join(); reads_release(reads); if (h_small) (void)hipHostFree(h_small); TRACE("~Mate body done"); }
};

// grow a device buffer, keeping what is in it (bytes)
// room for need_bytes (must_bytes, when given, is what has to fit now: need_bytes is then a wish -- the estimate for the whole file --
// that only counts when a new allocation has to be made anyway)
template <class T> int grow_keep(T *&p, size_t &cap_bytes, size_t used_bytes, size_t need_bytes, hipStream_t st, std::string &err, size_t must_bytes = 0)
{
    if ((must_bytes ? must_bytes : need_bytes) <= cap_bytes && p) return MF_OK;
    const size_t cap = must_bytes ? std::max(need_bytes + need_bytes / 16, must_bytes + must_bytes / 2) + 4096 : need_bytes + need_bytes / 2 + 4096;
    T *q = nullptr;
    DCHK(hipMalloc(&q, cap));
    if (p) { if (used_bytes) DCHK(hipMemcpyAsync(q, p, used_bytes, hipMemcpyDeviceToDevice, st)); DCHK(hipStreamSynchronize(st)); g_trash.add(p); }      // (hipMalloc'ed: an mf_reads owns these)
    p = q; cap_bytes = cap;
    return MF_OK;
}

struct Ingest {
    mf_kmerset *ks; uint32_t threshold; bool pair_both; int device; DevCtx *ctx; hipStream_t sp;
    Mate m[2]; int nm = 1;
    uint64_t kept = 0, total = 0;
    bool timing = false; double t_wait = 0, t_index = 0, t_pack = 0, t_filter = 0, t_emit = 0;

    void producer(Mate &M)
    {
        (void)hipSetDevice(phys(device));
        std::string err; int rc = MF_OK;
        if (M.gz) {
            for (;;) {
                TextPiece t;
                rc = M.gzs->next(t, err);
                if (rc || M.stop) break;
                if (!t.len && !t.last) continue;                       // (the first call: the decoder hands a slab over one call late)
                { std::lock_guard<std::mutex> lk(M.mu); M.ready.push_back(t); }
                M.cv.notify_all();
                if (t.last) break;
            }
        } else {
            // a plain file is its own text: pieces become available as they are uploaded
            const uint64_t slab = std::max<uint64_t>(env_u64("MF_INGEST_SLAB_BYTES", (uint64_t)256 << 20), 64);
            hipStream_t st = nullptr; hipEvent_t ev = nullptr;
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { rc = MF_E_HIP; err = "hipStreamCreate failed"; }
            for (uint64_t T0 = 0; !rc && T0 < M.map.n && !M.stop;) {
                const uint64_t T1 = std::min<uint64_t>(M.map.n, T0 + slab);
                if (!M.up.wait_for(st, T1) || hipStreamSynchronize(st) != hipSuccess) { rc = MF_E_HIP; err = "upload of " + M.path + " failed"; break; }
                TextPiece t; t.T0 = T0; t.len = T1 - T0; t.last = T1 == M.map.n; t.est_total = M.map.n;
                { std::lock_guard<std::mutex> lk(M.mu); M.ready.push_back(t); }
                M.cv.notify_all();
                T0 = T1;
            }
            if (ev) (void)hipEventDestroy(ev);
            if (st) (void)hipStreamDestroy(st);
        }
        TRACE("producer done rc %d", rc);
        { std::lock_guard<std::mutex> lk(M.mu); M.prod_rc = rc; M.prod_err = err; M.prod_done = true; }
        M.cv.notify_all();
    }

    // lines -> records -> appended to the mate's packed read set
    int ingest(Mate &M, const TextPiece &P, std::string &err)
    {
        const double t0 = now_s();
        std::lock_guard<std::mutex> alk(M.arena.mu);                 // the arena stays where it is while these kernels read it
        std::unique_ptr<Batch> B(new Batch());
        B->start = P.T0 - M.carry; B->rec_base = M.rec_done;
        const uint8_t *text = M.arena.p + B->start;
        const uint64_t n = M.carry + P.len;
        const uint64_t tiles = (n + INGEST_TILE - 1) / INGEST_TILE;
        if (!M.h_small) { DCHK(hipHostMalloc((void **)&M.h_small, 64, hipHostMallocDefault)); memset(M.h_small, 0, 64); }
        volatile uint64_t *hs = M.h_small;
        uint64_t n_lines = 0, used = 0;
        hs[2] = 0;
        if (n) {
            DCHK(M.tile_cnt.need(tiles)); DCHK(M.tile_base.need(tiles + 1)); DCHK(M.scan_tmp.need(tiles / 4096 + 4));
            DCHK(launch_count_newlines(text, n, M.tile_cnt.p, sp));
            DCHK(launch_scan_u32(M.tile_cnt.p, tiles, M.tile_base.p, M.scan_tmp.p, sp));
            hs[1] = 0;
            DCHK(hipMemcpyAsync(M.h_small + 0, M.tile_base.p + tiles, 8, hipMemcpyDeviceToHost, sp));
            DCHK(hipMemcpyAsync(M.h_small + 1, text + n - 1, 1, hipMemcpyDeviceToHost, sp));
            DCHK(hipStreamSynchronize(sp));
            const uint64_t newlines = hs[0]; const uint8_t last_byte = (uint8_t)hs[1];
            const bool open_line = P.last && last_byte != '\n';      // lines() yields an unterminated last line
            n_lines = newlines + (open_line ? 1 : 0);
            DCHK(B->line_start.need(n_lines + 2, false));
            DCHK(launch_line_starts(text, n, M.tile_base.p, B->line_start.p, sp));
            if (open_line) { const uint64_t v = n + 1; DCHK(hipMemcpyAsync(B->line_start.p + n_lines, &v, 8, hipMemcpyHostToDevice, sp)); DCHK(hipStreamSynchronize(sp)); }
            B->n_rec = n_lines / 4;
            DCHK(hipMemcpyAsync(M.h_small + 2, B->line_start.p + 4 * B->n_rec, 8, hipMemcpyDeviceToHost, sp));     // (read with the next synchronisation)
            if (!B->n_rec) DCHK(hipStreamSynchronize(sp));
        }
        if (timing) t_index += now_s() - t0;
        const double t1 = now_s();
        const uint64_t n_rec = B->n_rec;
        if (n_rec) {
            if (!M.reads) { M.reads = new (std::nothrow) mf_reads(); if (!M.reads) { err = "out of memory"; return MF_E_NOMEM; } M.reads->device = device; M.reads->lane = 0; }
            mf_reads *R = M.reads;
            // sequence lengths, the batch's own offsets
            DCHK(M.seq_len.need(n_rec)); DCHK(M.minmax.need(2)); DCHK(M.offsets_tmp.need(n_rec + 1)); DCHK(M.scan_tmp.need(n_rec / 4096 + 4));
            { const uint32_t init[2] = {~0u, 0u}; DCHK(hipMemcpyAsync(M.minmax.p, init, 8, hipMemcpyHostToDevice, sp)); }
            DCHK(launch_seq_lens(text, B->line_start.p, n_rec, M.seq_len.p, M.minmax.p, sp));
            DCHK(launch_scan_u32(M.seq_len.p, n_rec, M.offsets_tmp.p, M.scan_tmp.p, sp));
            DCHK(hipMemcpyAsync(M.h_small + 3, M.offsets_tmp.p + n_rec, 8, hipMemcpyDeviceToHost, sp));
            DCHK(hipMemcpyAsync(M.h_small + 4, M.minmax.p, 8, hipMemcpyDeviceToHost, sp));
            DCHK(hipStreamSynchronize(sp));
            const uint64_t nb = hs[3]; const uint32_t mm[2] = {(uint32_t)hs[4], (uint32_t)(hs[4] >> 32)};
            M.min_len = std::min(M.min_len, mm[0]); M.max_len = std::max(M.max_len, mm[1]);
            const uint32_t uniform = (mm[0] == mm[1] && mm[0] > 0) ? mm[0] : 0;
            // room in the file's read set: words (with the screen's padding), offsets, then pack behind what is there
            const uint64_t words_after = (M.bases + nb + 15) / 16;
            // (the read set is sized for the whole file from what this piece says about it -- bases and records per byte of text --
            // so that it is not moved to a larger allocation every few pieces: a hipMalloc and a copy of all there is so far each time)
            const uint64_t text_done = P.T0 + P.len;
            const double scale = text_done && P.est_total > text_done ? (double)P.est_total / (double)text_done : 1.0;
            const uint64_t words_est = (uint64_t)((double)words_after * scale), rec_est = (uint64_t)((double)(M.rec_done + n_rec) * scale);
            int rc = grow_keep(R->d_words, R->cap_words, ((M.bases + 15) / 16) * 4, padded_words_for(std::max(words_after, words_est)) * 4, sp, err, padded_words_for(words_after) * 4);
            if (rc) return rc;
            rc = grow_keep(R->d_offsets, R->cap_offsets, (M.rec_done + 1) * 8, (std::max(M.rec_done + n_rec, rec_est) + 1) * 8, sp, err, (M.rec_done + n_rec + 1) * 8);
            if (rc) return rc;
            DCHK(launch_add_base(R->d_offsets + M.rec_done, M.offsets_tmp.p, n_rec + 1, M.bases, sp));
            const uint64_t pb = pack_blocks(nb, M.bases);
            uint64_t inv = 0;
            if (pb) {
                DCHK(M.inv_cnt.need(pb)); DCHK(M.inv_base.need(pb + 1)); DCHK(M.scan_tmp.need(pb / 4096 + 4));
                DCHK(launch_pack(text, B->line_start.p, uniform ? nullptr : M.offsets_tmp.p, uniform, n_rec, nb, M.bases, R->d_words, M.inv_cnt.p, nullptr, nullptr, sp));
                DCHK(launch_scan_u32(M.inv_cnt.p, pb, M.inv_base.p, M.scan_tmp.p, sp));
                DCHK(hipMemcpyAsync(M.h_small + 5, M.inv_base.p + pb, 8, hipMemcpyDeviceToHost, sp));
                DCHK(hipStreamSynchronize(sp));
                inv = hs[5];
                if (inv) {
                    rc = grow_keep(R->d_npos, R->cap_npos, M.n_npos * 8, (M.n_npos + inv) * 8, sp, err);
                    if (rc) return rc;
                    DCHK(launch_pack(text, B->line_start.p, uniform ? nullptr : M.offsets_tmp.p, uniform, n_rec, nb, M.bases, R->d_words, M.inv_cnt.p, M.inv_base.p, R->d_npos + M.n_npos, sp));
                }
            }
            // (no synchronisation at the end: the next piece's kernels follow on the same stream, and the arena is not moved before
            // the event below has passed)
            M.bases += nb; M.n_npos += inv;
        }
        used = hs[2];                                                 // (arrived with one of the synchronisations above)
        if (used > n) used = n;                                       // (the virtual line end of an unterminated last line)
        M.carry = P.last ? 0 : n - used;                              // a partial record at the very end is dropped
        if (!M.arena.read_ev) DCHK(hipEventCreateWithFlags(&M.arena.read_ev, hipEventDisableTiming));
        DCHK(hipEventRecord(M.arena.read_ev, sp)); M.arena.read_pending = true;
        M.rec_done += n_rec;
        M.batches.push_back(std::move(B));
        if (timing) t_pack += now_s() - t1;
        return MF_OK;
    }

    // the whole file's read set through the filter: pass bits stay on the device (R->d_bits[R->cur])
    int filter(Mate &M, std::string &err)
    {
        if (!M.rec_done) return MF_OK;
        mf_reads *R = M.reads;
        const uint64_t n_words = (M.bases + 15) / 16, padded = padded_words_for(n_words);
        int rc = grow_keep(R->d_words, R->cap_words, n_words * 4, padded * 4, sp, err);
        if (rc) return rc;
        DCHK(hipMemsetAsync(R->d_words + n_words, 0, (padded - n_words) * 4, sp));
        if (!R->d_npos) { rc = grow_keep(R->d_npos, R->cap_npos, 0, 8, sp, err); if (rc) return rc; }
        const uint32_t uniform = (M.min_len == M.max_len && M.min_len > 0) ? M.min_len : 0;
        rc = reads_finish(R, false, n_words, M.rec_done, M.bases, uniform, M.n_npos, ctx);
        if (rc) { err = mf_thread_error(); return rc; }
        rc = filter_common(ks, R, threshold, MF_MODE_SCREENED, nullptr, nullptr, 1, nullptr);
        if (rc) { err = mf_thread_error(); return rc; }
        return MF_OK;
    }

    // survivors of the first n_emit records of batch B -> the mate's writer
    int emit(Mate &M, Batch &B, uint64_t n_emit, const uint32_t *bits_other, std::string &err)
    {
        if (n_emit > B.n_rec) n_emit = B.n_rec;
        if (!n_emit) return MF_OK;
        const uint8_t *text = M.arena.p + B.start;
        DCHK(M.out_len.need(n_emit)); DCHK(M.out_off.need(n_emit + 1)); DCHK(M.scan_tmp.need(n_emit / 4096 + 4));
        DCHK(launch_out_lens(text, B.line_start.p, n_emit, B.rec_base, M.reads->d_bits[M.reads->cur], bits_other, pair_both ? 1 : 0, M.out_len.p, sp));
        DCHK(launch_scan_u32(M.out_len.p, n_emit, M.out_off.p, M.scan_tmp.p, sp));
        uint64_t bytes = 0;
        DCHK(hipMemcpyAsync(&bytes, M.out_off.p + n_emit, 8, hipMemcpyDeviceToHost, sp));
        DCHK(hipStreamSynchronize(sp));
        if (bytes) {
            DCHK(M.d_out.need(bytes));
            DCHK(launch_gather(text, B.line_start.p, n_emit, M.out_len.p, M.out_off.p, M.d_out.p, sp));
            std::vector<char> host(bytes);
            DCHK(hipMemcpyAsync(host.data(), M.d_out.p, bytes, hipMemcpyDeviceToHost, sp));
            DCHK(hipStreamSynchronize(sp));
            M.out.push(std::move(host));
        }
        return MF_OK;
    }

    // number of survivors among pairs [0, n): population count over the combined bitmaps (read back in pieces)
    int count_kept(uint64_t n, std::string &err)
    {
        kept = 0;
        const uint64_t nw = (n + 31) / 32;
        std::vector<uint32_t> a((size_t)std::min<uint64_t>(nw, (uint64_t)1 << 22)), b(a.size());
        for (uint64_t w0 = 0; w0 < nw; w0 += a.size()) {
            const uint64_t k = std::min<uint64_t>(a.size(), nw - w0);
            DCHK(hipMemcpy(a.data(), m[0].reads->d_bits[m[0].reads->cur] + w0, k * 4, hipMemcpyDeviceToHost));
            if (nm == 2) DCHK(hipMemcpy(b.data(), m[1].reads->d_bits[m[1].reads->cur] + w0, k * 4, hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i < k; i++) {
                uint32_t v = nm == 2 ? (pair_both ? (a[i] & b[i]) : (a[i] | b[i])) : a[i];
                const uint64_t first = (w0 + i) * 32;
                if (first + 32 > n) v &= n > first ? ((1u << (n - first)) - 1) : 0u;
                kept += (uint64_t)__builtin_popcount(v);
            }
        }
        return MF_OK;
    }

    int run(std::string &err)
    {
        for (int i = 0; i < nm; i++) m[i].prod = std::thread([this, i] { producer(m[i]); });
        // ---- ingest the text as it becomes available, the mate that is behind first
        for (;;) {
            int pick = -1;
            for (int i = 0; i < nm; i++) if (!m[i].eof && (pick < 0 || m[i].rec_done < m[pick].rec_done)) pick = i;
            if (pick < 0) break;
            // (a mate whose text is not there yet does not hold up the other)
            bool got = false; TextPiece P;
            const double tw = now_s();
            for (;;) {
                for (int step = 0; step < nm && !got; step++) {
                    Mate &M = m[(pick + step) % nm];
                    if (M.eof) continue;
                    std::unique_lock<std::mutex> lk(M.mu);
                    if (!M.ready.empty()) { P = M.ready.front(); M.ready.pop_front(); got = true; pick = (pick + step) % nm; }
                    else if (M.prod_done) {
                        if (M.prod_rc) { err = M.prod_err; return M.prod_rc; }
                        M.eof = true;                                  // (an input without text: an empty file cannot get here, but a .gz of nothing can)
                    }
                }
                if (got) break;
                bool any = false; for (int i = 0; i < nm; i++) any = any || !m[i].eof;
                if (!any) break;
                std::unique_lock<std::mutex> lk(m[pick].mu);
                m[pick].cv.wait_for(lk, std::chrono::milliseconds(1));
            }
            if (timing) t_wait += now_s() - tw;
            if (!got) continue;
            Mate &M = m[pick];
            const int rc = ingest(M, P, err);
            if (rc) return rc;
            if (P.last) M.eof = true;
        }
        for (int i = 0; i < nm; i++) { Mate &M = m[i]; if (M.prod.joinable()) M.prod.join(); if (M.prod_rc) { err = M.prod_err; return M.prod_rc; } }
        // ---- filter, then write the survivors batch by batch
        const double tf = now_s();
        total = nm == 2 ? std::min(m[0].rec_done, m[1].rec_done) : m[0].rec_done;
        for (int i = 0; i < nm; i++) { const int rc = filter(m[i], err); if (rc) return rc; }
        if (timing) t_filter += now_s() - tf;
        const double te = now_s();
        if (total) {
            for (int i = 0; i < nm; i++) {
                Mate &M = m[i];
                const uint32_t *other = nm == 2 ? m[1 - i].reads->d_bits[m[1 - i].reads->cur] : nullptr;
                for (auto &B : M.batches) {
                    if (B->rec_base >= total) break;                  // pairs end with the shorter file
                    const int rc = emit(M, *B, total - B->rec_base, other, err);
                    if (rc) return rc;
                }
            }
            const int rc = count_kept(total, err);
            if (rc) return rc;
        }
        if (timing) t_emit += now_s() - te;
        return MF_OK;
    }
};

} // namespace

int run_device_ingest(mf_kmerset *ks, const char *fq1, const char *fq2, const char *out1, const char *out2, uint32_t threshold,
                      bool pair_both, int device, uint64_t *kept, uint64_t *total, std::string &err)
{
    struct EndOfCall { ~EndOfCall() { g_trash.empty(); const char *kb = getenv("MF_KEEP_BUFFERS"); g_pool.trim(kb && kb[0] == '0' ? 0 : (size_t)env_u64("MF_DEVPOOL_GB", 96) << 30); } } end_of_call;     // (declared first: runs after everything of this call is gone)
    Ingest I;
    I.ks = ks; I.threshold = threshold; I.pair_both = pair_both; I.device = device; I.nm = fq2 ? 2 : 1;
    I.timing = getenv("MF_PIPE_TIMING") != nullptr;
    const char *in_path[2] = {fq1, fq2}, *out_path[2] = {out1, out2};
    // ---- is this an input for the device path?  Everything of a file stays resident until its survivors are written: the
    // compressed bytes, the text, the line index, the packed reads.
    size_t need = (size_t)4 << 30;
    for (int i = 0; i < I.nm; i++) {
        Mate &M = I.m[i];
        M.path = in_path[i]; M.gz = has_gz_ext(in_path[i]);
        bool regular = false;
        if (!M.map.open(in_path[i], regular)) { err = std::string("Cannot open file ") + in_path[i]; return MF_E_IO; }
        if (!regular || M.map.n == 0) return MF_DEVINGEST_DECLINED;
        if (M.gz) {
            const uint8_t *d = M.map.p;
            if (M.map.n < 18 || d[0] != 0x1f || d[1] != 0x8b) return MF_DEVINGEST_DECLINED;      // (gzread hands such a file through; so does the host reader)
            if ((d[3] & 4) && M.map.n >= 18 && d[12] == 'B' && d[13] == 'C') return MF_DEVINGEST_DECLINED;   // BGZF: the host reader decodes its members side by side
        }
        need += M.gz ? M.map.n * 8 + ((size_t)32 << 30) : M.map.n + M.map.n / 2;      // (file, text arena, read set; twelve symbol buffers of 2.4 GB)
    }
    int rc = get_ctx(device, &I.ctx, 0);
    if (rc) { err = mf_thread_error(); return rc; }
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b + g_pool.held() < need) return MF_DEVINGEST_DECLINED;       // too large to keep resident: the host pipeline streams it (what the pool holds from earlier calls counts as free)
    }
    const double t_begin = now_s();
    I.sp = I.ctx->stream;
    for (int i = 0; i < I.nm; i++) {
        Mate &M = I.m[i];
        DCHK(M.d_file.need(M.map.n + 256, false));
        DCHK(hipMemsetAsync(M.d_file.p + M.map.n, 0, 256, I.sp));
        DCHK(hipStreamSynchronize(I.sp));
        rc = M.up.start(M.map.p, M.map.fd, M.map.n, M.d_file.p, phys(device), err);
        if (rc) return rc;
        if (M.gz) {
            M.gzs.reset(new GzStream());
            rc = M.gzs->open(M.map.p, M.map.n, M.d_file.p, &M.up, &M.arena, M.path, err);
            if (rc) return rc;
        } else { M.arena.p = M.d_file.p; M.arena.cap = M.map.n; }
    }
    for (int i = 0; i < I.nm; i++) if (!I.m[i].out.open(out_path[i])) { err = std::string("Cannot open file ") + out_path[i]; return MF_E_IO; }
    const double t_setup = now_s() - t_begin;
    rc = I.run(err);
    TRACE("run returned %d", rc);
    for (int i = 0; i < I.nm; i++) I.m[i].stop = true;
    bool wrote = true;
    for (int i = 0; i < I.nm; i++) wrote = I.m[i].out.close() && wrote;
    if (rc) return rc;
    if (!wrote) { err = std::string("write error on ") + out_path[0]; return MF_E_IO; }
    if (kept) *kept = I.kept;
    if (total) *total = I.total;
    if (I.timing) {
        fprintf(stderr, "[mf device ingest] wall %.3f s | set-up %.3f | waiting for text (upload, inflate, link, CRC on the producer threads) %.3f | line index %.3f | pack %.3f | filter %.3f | survivors %.3f",
                now_s() - t_begin, t_setup, I.t_wait, I.t_index, I.t_pack, I.t_filter, I.t_emit);
        for (int i = 0; i < I.nm; i++)
            if (I.m[i].gzs) fprintf(stderr, " | mate %d: %llu of %u chunks of %zu KiB linked, %llu gaps bridged on the host, %llu bytes decoded there", i + 1, (unsigned long long)I.m[i].gzs->chunks_linked(),
                                    I.m[i].gzs->chunks(), I.m[i].gzs->chunk_bytes() >> 10, (unsigned long long)I.m[i].gzs->gaps(), (unsigned long long)I.m[i].gzs->gap_bytes());
        fprintf(stderr, "\n");
    }
    return MF_OK;
}

} // namespace mf
