// Device ingest path of mf_filter_fastq_files (see mf_devingest.h).
//
// Per input file ("mate"): the mapped file is copied to the device by an uploader thread (pinned staging, its own copy stream).
// A plain FASTQ file then IS the text; a .gz is decoded in slabs of a few thousand speculative chunks (mf_gzdev.h): decode of
// slab k + 1 and k + 2 run on their own streams while slab k is linked, resolved, CRC-checked, indexed, packed, filtered and
// its survivors copied out.  Text is cut into records where it lies (mf_ingest.h); what is behind the last complete record of
// a slab (the carry) is the front of the next slab's text.  Pass bits go to a file-wide bitmap per mate, so that the pair rule
// can be applied to slabs of the two mates that do not cover the same records; the mate that is behind in records is advanced.
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
#include <memory>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace mf {
namespace {

#define DCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err = std::string(#call) + " failed: " + hipGetErrorString(e_); return MF_E_HIP; } } while (0)

uint64_t env_u64(const char *name, uint64_t dflt) { const char *v = getenv(name); return v && *v ? strtoull(v, nullptr, 10) : dflt; }
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class T> struct DevBuf {
    T *p = nullptr; size_t cap = 0;                      // cap in elements
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete; DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t need(size_t n, bool slack = true)          // contents are NOT kept
    {
        if (n <= cap && p) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = slack ? n + n / 8 + 1024 : (n ? n : 1);
        hipError_t e = hipMalloc(&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
};

struct Mapped {
    const uint8_t *p = nullptr; size_t n = 0;
    ~Mapped() { if (p) munmap(const_cast<uint8_t *>(p), n); }
    bool open(const char *path, bool &regular)
    {
        regular = false;
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); return true; }
        regular = true;
        n = (size_t)st.st_size;
        if (n) {
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ::close(fd); n = 0; return false; }
            p = (const uint8_t *)m;
            madvise(m, n, MADV_SEQUENTIAL);
        }
        ::close(fd);
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
        stop_ = true;
        if (th_.joinable()) th_.join();
        for (auto &e : ev_) if (e) (void)hipEventDestroy(e);
        for (auto &e : free_ev_) if (e) (void)hipEventDestroy(e);
        for (auto &s : stage_) if (s) (void)hipHostFree(s);
        if (st_) (void)hipStreamDestroy(st_);
    }
    int start(const uint8_t *src, size_t n, uint8_t *dst, int device, std::string &err)
    {
        src_ = src; n_ = n; dst_ = dst; device_ = device;
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
        const int nthr = 4;
        for (size_t i = 0; i < np && !stop_; i++) {
            const int b = (int)(i & 1);
            if (i >= 2 && hipEventSynchronize(free_ev_[b]) != hipSuccess) { fail_(); return; }       // the copy that read this staging buffer is done
            const size_t off = i * PIECE, len = std::min(PIECE, n_ - off);
            {   // page-cache -> pinned memory on a few threads (one memcpy stream does ~5 GB/s)
                std::vector<std::thread> th;
                for (int t = 1; t < nthr; t++) th.emplace_back([&, t] { const size_t a = len * t / nthr, e = len * (t + 1) / nthr; memcpy(stage_[b] + a, src_ + off + a, e - a); });
                memcpy(stage_[b], src_ + off, len / nthr);
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
    const uint8_t *src_ = nullptr; size_t n_ = 0; uint8_t *dst_ = nullptr; int device_ = 0;
    hipStream_t st_ = nullptr; uint8_t *stage_[2] = {nullptr, nullptr}; hipEvent_t free_ev_[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> ev_;
    std::thread th_; std::mutex mu_; std::condition_variable cv_; size_t enqueued_ = 0; bool failed_ = false; std::atomic<bool> stop_{false};
};

// ---- text of one slab on the device.  base[0 .. len) are the new bytes; base[-front .. 0) is the text in front of them.
struct Slab {
    DevBuf<uint8_t> buf;                 // owns the text of a .gz slab (a plain file's text is the resident file)
    uint8_t *base = nullptr; uint64_t len = 0; size_t front = 0;
    uint64_t carry = 0;                  // bytes in front of base that open the first record of this slab
    DevBuf<uint64_t> line_start;         // offsets from (base - carry)
    uint64_t n_rec = 0, rec_base = 0;
    bool last = false;                   // the input ends with this slab
};

// ---- one gzip member stream decoded on the device
class GzStream {
public:
    ~GzStream()
    {
        for (auto &s : sym_) { if (s.ev) (void)hipEventDestroy(s.ev); }
        for (auto &s : sd_) if (s) (void)hipStreamDestroy(s);
        if (h_chain_) (void)hipHostFree(h_chain_);
    }
    // data: the mapped file; d_file: its copy on the device (size + 64 readable, being filled by `up`)
    int open(const uint8_t *data, size_t size, uint8_t *d_file, Uploader *up, const std::string &path, std::string &err)
    {
        data_ = data; size_ = size; d_file_ = d_file; up_ = up; path_ = path;
        chunk_ = (size_t)env_u64("MF_GZDEV_CHUNK_BYTES", (size_t)128 << 10);
        if (chunk_ < 1024) chunk_ = 1024;
        cps_ = (uint32_t)env_u64("MF_GZDEV_SLAB_CHUNKS", 4096);
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
        DCHK(hipMemset(d_chunks_.p, 0, n_chunks_ * sizeof(GzChunk)));
        DCHK(hipMemset(d_out_off_.p, 0xFF, n_chunks_ * sizeof(uint64_t)));
        DCHK(hipHostMalloc((void **)&h_chain_, sizeof(GzChain), hipHostMallocDefault));
        memset(h_chain_, 0, sizeof(GzChain));
        h_chain_->cur_bit = (uint64_t)base_byte_ * 8;
        DCHK(hipMemcpy(d_chain_.p, h_chain_, sizeof(GzChain), hipMemcpyHostToDevice));
        for (auto &s : sd_) DCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (auto &s : sym_) { DCHK(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming)); s.slab = ~0u; }
        h_chunks_.resize(n_chunks_);
        in_member_ = true;
        return MF_OK;
    }
    bool done() const { return done_; }
    // text of the next slab (possibly empty) into `out`; sp: the stream the text is produced on.  Sets out.last at the end of the input.
    int next(Slab &out, hipStream_t sp, std::string &err)
    {
        const uint32_t k = next_slab_;
        for (uint32_t j = k; j < k + 3 && j < n_slabs_; j++) { const int rc = launch_decode(j, err); if (rc) return rc; }
        const uint32_t lo = k * cps_, hi = std::min(n_chunks_, lo + cps_);
        Sym &S = sym_[k % 3];
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
            if (!overflow) break;
            // text that expands more than the symbol buffers allow for (a run of identical reads, say): this slab again, with four times the room
            if (S.cap > chunk_ * 2048) { err = "gzip data in " + path_ + " expands more than a thousandfold: not decoded on the device"; return MF_E_FORMAT; }
            S.cap *= 4;
            DCHK(S.p.need((size_t)(hi - lo) * S.cap, false));
            DCHK(launch_gz_decode(d_file_, size_, base_byte_, chunk_, lo, hi - lo, 0, (uint64_t)base_byte_ * 8, S.p.p, S.cap, d_chunks_.p, sp));
            DCHK(hipMemcpyAsync(h_chunks_.data() + lo, d_chunks_.p + lo, (hi - lo) * sizeof(GzChunk), hipMemcpyDeviceToHost, sp));
            DCHK(hipStreamSynchronize(sp));
        }
        // the text buffer: the 32 KiB window (or the longer carry) in front, room for what the chunks hold and for gap fills
        const size_t front = ((std::max<uint64_t>(GZ_WINDOW, carry_in_) + 255) & ~(size_t)255) + 256;
        size_t room = front + sum + ((size_t)16 << 20);
        DCHK(out.buf.need(room + 64, false));
        room = out.buf.cap - 64;
        out.front = front; out.base = out.buf.p + front; out.len = 0;
        const uint64_t T0 = h_chain_->total;
        if (carry_in_ > GZ_WINDOW) {            // a record longer than the window: its head comes from the previous slab's buffer
            if (!prev_base_) { err = "internal: carry without a previous slab"; return MF_E_HIP; }
            DCHK(hipMemcpyAsync(out.base - carry_in_, prev_base_ + prev_len_ - carry_in_, carry_in_ - GZ_WINDOW, hipMemcpyDeviceToDevice, sp));
        }
        const bool last_slab = k + 1 == n_slabs_;
        // link; the host steps in where the chain stops
        for (;;) {
            if (!done_ && in_member_) {
                DCHK(launch_gz_chain(d_chain_.p, d_chunks_.p, lo, hi, S.p.p, S.cap, d_out_off_.p, out.base, T0, sp));
                DCHK(hipMemcpyAsync(h_chain_, d_chain_.p, offsetof(GzChain, window), hipMemcpyDeviceToHost, sp));
                DCHK(hipStreamSynchronize(sp));
            }
            if (done_) break;
            uint64_t to_bit = 0;
            if (h_chain_->stop == GZ_STOP_MEMBER_END) { const int rc = member_end(out, T0, sp, err); if (rc) return rc; if (done_) break; }
            if (h_chain_->stop == GZ_STOP_GAP) to_bit = h_chunks_[h_chain_->next].start_bit;
            else if (h_chain_->stop == GZ_STOP_NONE) {
                if (!last_slab) break;
                to_bit = (uint64_t)size_ * 8;             // behind the last chunk: the host decodes to the end of the member
            }
            // ---- decode across the gap on the host, with the window the chain left
            DCHK(hipMemcpy(h_chain_->window, d_chain_.p->window, GZ_WINDOW, hipMemcpyDeviceToHost));
            std::vector<uint8_t> bytes; uint64_t end_bit = 0; bool mend = false; std::string why;
            if (!inflate_gap(data_, size_, h_chain_->cur_bit, to_bit, h_chain_->window, h_chain_->wlen, bytes, end_bit, mend, why)) {
                err = "gzip read error in " + path_ + ": " + why; return MF_E_FORMAT;
            }
            gap_bytes_ += bytes.size();
            const uint64_t at = h_chain_->total - T0;
            if (front + at + bytes.size() > room) {       // (rare: a long gap) a larger buffer, what is there moves over
                DevBuf<uint8_t> nb; const size_t nroom = front + at + bytes.size() + sum + ((size_t)64 << 20);
                DCHK(nb.need(nroom + 64, false));
                DCHK(hipMemcpy(nb.p, out.buf.p, front + at, hipMemcpyDeviceToDevice));
                std::swap(nb.p, out.buf.p); std::swap(nb.cap, out.buf.cap);
                room = out.buf.cap - 64; out.base = out.buf.p + front;
            }
            if (!bytes.empty()) DCHK(hipMemcpy(out.base + at, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
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
            DCHK(hipMemcpy(d_chain_.p, h_chain_, sizeof(GzChain), hipMemcpyHostToDevice));
            if (mend) { const int rc = member_end(out, T0, sp, err); if (rc) return rc; if (done_) break; }
        }
        DCHK(launch_gz_resolve(d_chunks_.p, lo, hi, S.p.p, S.cap, d_out_off_.p, out.base, T0, max_sym, sp));
        out.len = h_chain_->total - T0;
        // the rest of the member's CRC over this slab
        if (out.len > crc_done_ - T0) { const int rc = crc_over(out, T0, crc_done_, h_chain_->total, sp, err); if (rc) return rc; }
        DCHK(hipStreamSynchronize(sp));
        S.slab = ~0u;                                     // (its symbols are text now)
        next_slab_ = k + 1;
        if (next_slab_ == n_slabs_ && !done_) {           // the data ran out inside a member
            err = "gzip read error in " + path_ + ": unexpected end of file"; return MF_E_FORMAT;
        }
        out.last = done_;
        prev_base_ = out.base; prev_len_ = out.len;
        return MF_OK;
    }
    void set_carry(uint64_t c) { carry_in_ = c; }
    uint64_t gap_bytes() const { return gap_bytes_; }
    uint64_t chunks_linked() const { return h_chain_ ? h_chain_->linked : 0; }
private:
    struct Sym { DevBuf<uint16_t> p; size_t cap = 0; hipEvent_t ev = nullptr; uint32_t slab = ~0u; };     // cap: symbols of room per chunk
    int launch_decode(uint32_t j, std::string &err)
    {
        Sym &S = sym_[j % 3];
        if (j != launched_ || S.slab != ~0u) return MF_OK;     // launched already -- or its buffer still holds an earlier slab: launched when that one has become text
        const uint32_t lo = j * cps_, hi = std::min(n_chunks_, lo + cps_);
        S.cap = sym_cap_;
        DCHK(S.p.need((size_t)(hi - lo) * S.cap, false));
        hipStream_t st = sd_[j % 2];
        // the chunks read past their own range up to the end of a block, and the reader's ring a little further
        if (!up_->wait_for(st, base_byte_ + (size_t)hi * chunk_ + ((size_t)8 << 20))) { err = "upload of " + path_ + " failed"; return MF_E_HIP; }
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
    int member_end(Slab &out, uint64_t T0, hipStream_t sp, std::string &err)
    {
        const size_t pos = (size_t)((h_chain_->cur_bit + 7) >> 3);
        if (pos + 8 > size_) { err = "gzip read error in " + path_ + ": truncated gzip trailer"; return MF_E_FORMAT; }
        uint32_t want_crc, want_len; memcpy(&want_crc, data_ + pos, 4); memcpy(&want_len, data_ + pos + 4, 4);
        // CRC of the member's text up to here (the resolve kernel has not run yet for this slab: do it for what is accepted)
        {
            const uint32_t k = next_slab_, lo = k * cps_, hi = std::min(n_chunks_, lo + cps_);
            uint32_t max_sym = 0; for (uint32_t c = lo; c < hi; c++) max_sym = std::max(max_sym, h_chunks_[c].n_sym);
            DCHK(launch_gz_resolve(d_chunks_.p, lo, hi, sym_[k % 3].p.p, sym_[k % 3].cap, d_out_off_.p, out.base, T0, max_sym, sp));
        }
        if (h_chain_->total > crc_done_) { const int rc = crc_over(out, T0, crc_done_, h_chain_->total, sp, err); if (rc) return rc; }
        if (crc_ != want_crc) { err = "gzip read error in " + path_ + ": incorrect data check"; return MF_E_FORMAT; }
        if ((uint32_t)(h_chain_->total - member_T0_) != want_len) { err = "gzip read error in " + path_ + ": incorrect length check"; return MF_E_FORMAT; }
        crc_ = 0; member_T0_ = h_chain_->total;
        size_t p = pos + 8;
        if (p >= size_ || size_ - p < 2 || data_[p] != 0x1f || data_[p + 1] != 0x8b) { done_ = true; in_member_ = false; return MF_OK; }   // trailing bytes that are no member: ignored
        if (!member_header(p, err)) return MF_E_FORMAT;
        h_chain_->cur_bit = (uint64_t)p * 8; h_chain_->wlen = 0; h_chain_->stop = GZ_STOP_NONE;
        DCHK(hipMemcpy(d_chain_.p, h_chain_, offsetof(GzChain, window), hipMemcpyHostToDevice));
        return MF_OK;
    }
    // running CRC of the member over the text [from, to) of this slab
    int crc_over(Slab &out, uint64_t T0, uint64_t from, uint64_t to, hipStream_t sp, std::string &err)
    {
        const uint64_t n = to - from;
        const size_t np = (size_t)((n + GZ_CRC_PIECE - 1) / GZ_CRC_PIECE);
        DCHK(d_crc_.need(np));
        h_crc_.resize(np);
        DCHK(launch_gz_crc(out.base + (from - T0), n, d_crc_.p, sp));
        DCHK(hipMemcpyAsync(h_crc_.data(), d_crc_.p, np * 4, hipMemcpyDeviceToHost, sp));
        DCHK(hipStreamSynchronize(sp));
        crc_ = gz_crc_combine(crc_, gz_crc_finish(h_crc_.data(), n), n);
        crc_done_ = to;
        return MF_OK;
    }

    const uint8_t *data_ = nullptr; size_t size_ = 0; uint8_t *d_file_ = nullptr; Uploader *up_ = nullptr; std::string path_;
    size_t chunk_ = 0, base_byte_ = 0, sym_cap_ = 0; uint64_t expand_ = 8;
    uint32_t cps_ = 0, n_chunks_ = 0, n_slabs_ = 0, next_slab_ = 0, launched_ = 0;
    DevBuf<GzChunk> d_chunks_; DevBuf<uint64_t> d_out_off_; DevBuf<GzChain> d_chain_; DevBuf<uint32_t> d_crc_;
    GzChain *h_chain_ = nullptr; std::vector<GzChunk> h_chunks_; std::vector<uint32_t> h_crc_;
    Sym sym_[3]; hipStream_t sd_[2] = {nullptr, nullptr};
    bool in_member_ = false, done_ = false;
    uint32_t crc_ = 0; uint64_t crc_done_ = 0, member_T0_ = 0, carry_in_ = 0, gap_bytes_ = 0;
    uint8_t *prev_base_ = nullptr; uint64_t prev_len_ = 0;
};

// ---- survivors on their way to the output file (one writer thread per mate; slabs arrive in order)
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

struct Mate {
    std::string path; Mapped map; bool gz = false;
    DevBuf<uint8_t> d_file; Uploader up; std::unique_ptr<GzStream> gzs;
    uint64_t text_pos = 0;               // plain files: bytes of the file handed out so far
    uint64_t carry = 0;                  // bytes in front of the next slab's text that belong to its first record
    uint64_t rec_done = 0; bool eof = false;
    std::deque<std::unique_ptr<Slab>> pending;      // filtered, waiting for the other mate / to be written
    std::vector<std::unique_ptr<Slab>> spare;
    DevBuf<uint32_t> file_bits; uint64_t bits_cap = 0;   // pass bit per record of the file
    mf_reads *reads = nullptr;
    Writer out;
    // scratch of the ingest kernels
    DevBuf<uint32_t> tile_cnt, seq_len, inv_cnt, out_len, minmax; DevBuf<uint64_t> tile_base, scan_tmp, inv_base, out_off, offsets_tmp; DevBuf<uint8_t> d_out;
    ~Mate() { reads_release(reads); }
};

struct Ingest {
    mf_kmerset *ks; uint32_t threshold; bool pair_both; int device; DevCtx *ctx; hipStream_t sp;
    Mate m[2]; int nm = 1;
    uint64_t kept = 0, total = 0;
    bool timing = false; double t_text = 0, t_index = 0, t_pack = 0, t_filter = 0, t_emit = 0;

    int grow_bits(Mate &M, uint64_t n_rec, std::string &err)
    {
        if (n_rec <= M.bits_cap) return MF_OK;
        uint64_t cap = std::max<uint64_t>(n_rec + n_rec / 2, (uint64_t)1 << 22);
        cap = (cap + 1023) & ~(uint64_t)1023;
        uint32_t *p = nullptr;
        DCHK(hipMalloc(&p, cap / 8 + 64));
        DCHK(hipMemsetAsync(p, 0, cap / 8 + 64, sp));
        if (M.file_bits.p) { DCHK(hipMemcpyAsync(p, M.file_bits.p, M.bits_cap / 8, hipMemcpyDeviceToDevice, sp)); DCHK(hipStreamSynchronize(sp)); (void)hipFree(M.file_bits.p); }
        M.file_bits.p = p; M.file_bits.cap = cap / 32; M.bits_cap = cap;
        return MF_OK;
    }

    // text of the next slab of mate M
    int next_text(Mate &M, Slab &S, std::string &err)
    {
        S.carry = M.carry; S.n_rec = 0; S.rec_base = M.rec_done; S.last = false;
        if (M.gz) {
            M.gzs->set_carry(M.carry);
            const int rc = M.gzs->next(S, sp, err);
            if (rc) return rc;
            return MF_OK;
        }
        const uint64_t slab = std::max<uint64_t>(env_u64("MF_INGEST_SLAB_BYTES", (uint64_t)1 << 30), 64);
        const uint64_t T0 = M.text_pos, T1 = std::min<uint64_t>(M.map.n, T0 + slab);
        if (!M.up.wait_for(sp, T1)) { err = "upload of " + M.path + " failed"; return MF_E_HIP; }
        S.base = M.d_file.p + T0; S.len = T1 - T0; S.front = T0;
        M.text_pos = T1; S.last = T1 == M.map.n;
        return MF_OK;
    }

    // lines -> records -> packed read set -> pass bits in the file-wide bitmap
    int ingest(Mate &M, Slab &S, std::string &err)
    {
        const double t0 = now_s();
        const uint8_t *text = S.base - S.carry;
        const uint64_t n = S.carry + S.len;
        const uint64_t tiles = (n + INGEST_TILE - 1) / INGEST_TILE;
        uint64_t n_lines = 0, used = 0;
        if (n) {
            DCHK(M.tile_cnt.need(tiles)); DCHK(M.tile_base.need(tiles + 1)); DCHK(M.scan_tmp.need(tiles / 4096 + 4));
            DCHK(launch_count_newlines(text, n, M.tile_cnt.p, sp));
            DCHK(launch_scan_u32(M.tile_cnt.p, tiles, M.tile_base.p, M.scan_tmp.p, sp));
            uint64_t newlines = 0; uint8_t last_byte = 0;
            DCHK(hipMemcpyAsync(&newlines, M.tile_base.p + tiles, 8, hipMemcpyDeviceToHost, sp));
            DCHK(hipMemcpyAsync(&last_byte, text + n - 1, 1, hipMemcpyDeviceToHost, sp));
            DCHK(hipStreamSynchronize(sp));
            const bool open_line = S.last && last_byte != '\n';      // lines() yields an unterminated last line
            n_lines = newlines + (open_line ? 1 : 0);
            DCHK(S.line_start.need(n_lines + 2));
            DCHK(launch_line_starts(text, n, M.tile_base.p, S.line_start.p, sp));
            if (open_line) { const uint64_t v = n + 1; DCHK(hipMemcpyAsync(S.line_start.p + n_lines, &v, 8, hipMemcpyHostToDevice, sp)); DCHK(hipStreamSynchronize(sp)); }
            S.n_rec = n_lines / 4;
            DCHK(hipMemcpyAsync(&used, S.line_start.p + 4 * S.n_rec, 8, hipMemcpyDeviceToHost, sp));
            DCHK(hipStreamSynchronize(sp));
            if (used > n) used = n;                                   // (the virtual line end of an unterminated last line)
        }
        M.carry = S.last ? 0 : n - used;                              // a partial record at the very end is dropped
        if (timing) { DCHK(hipStreamSynchronize(sp)); t_index += now_s() - t0; }
        const double t1 = now_s();
        const uint64_t n_rec = S.n_rec;
        int rc = grow_bits(M, S.rec_base + n_rec + 64, err);
        if (rc) return rc;
        if (n_rec == 0) return MF_OK;
        // sequence lengths, offsets
        DCHK(M.seq_len.need(n_rec)); DCHK(M.minmax.need(2)); DCHK(M.offsets_tmp.need(n_rec + 1)); DCHK(M.scan_tmp.need(n_rec / 4096 + 4));
        { const uint32_t init[2] = {~0u, 0u}; DCHK(hipMemcpyAsync(M.minmax.p, init, 8, hipMemcpyHostToDevice, sp)); }
        DCHK(launch_seq_lens(text, S.line_start.p, n_rec, M.seq_len.p, M.minmax.p, sp));
        DCHK(launch_scan_u32(M.seq_len.p, n_rec, M.offsets_tmp.p, M.scan_tmp.p, sp));
        uint64_t total_bases = 0; uint32_t mm[2] = {0, 0};
        DCHK(hipMemcpyAsync(&total_bases, M.offsets_tmp.p + n_rec, 8, hipMemcpyDeviceToHost, sp));
        DCHK(hipMemcpyAsync(mm, M.minmax.p, 8, hipMemcpyDeviceToHost, sp));
        DCHK(hipStreamSynchronize(sp));
        const uint32_t uniform = (mm[0] == mm[1] && mm[0] > 0) ? mm[0] : 0;
        const uint64_t n_words = (total_bases + 15) / 16;
        if (!M.reads) { M.reads = new (std::nothrow) mf_reads(); if (!M.reads) { err = "out of memory"; return MF_E_NOMEM; } M.reads->device = device; M.reads->lane = 0; }
        mf_reads *R = M.reads;
        rc = reads_reserve(R, true, n_words, n_rec, uniform, 0, ctx);
        if (rc) { err = mf_thread_error(); return rc; }
        if (!uniform) DCHK(hipMemcpyAsync(R->d_offsets, M.offsets_tmp.p, (n_rec + 1) * 8, hipMemcpyDeviceToDevice, sp));
        const uint64_t pack_blocks = (n_words + 255) / 256;
        uint64_t n_npos = 0;
        if (pack_blocks) {
            DCHK(M.inv_cnt.need(pack_blocks)); DCHK(M.inv_base.need(pack_blocks + 1)); DCHK(M.scan_tmp.need(pack_blocks / 4096 + 4));
            DCHK(launch_pack(text, S.line_start.p, uniform ? nullptr : R->d_offsets, uniform, n_rec, total_bases, R->d_words, M.inv_cnt.p, nullptr, nullptr, sp));
            DCHK(launch_scan_u32(M.inv_cnt.p, pack_blocks, M.inv_base.p, M.scan_tmp.p, sp));
            DCHK(hipMemcpyAsync(&n_npos, M.inv_base.p + pack_blocks, 8, hipMemcpyDeviceToHost, sp));
            DCHK(hipStreamSynchronize(sp));
            if (n_npos) {
                DCHK(dev_reserve(R->d_npos, R->cap_npos, n_npos * 8, true));
                DCHK(launch_pack(text, S.line_start.p, uniform ? nullptr : R->d_offsets, uniform, n_rec, total_bases, R->d_words, M.inv_cnt.p, M.inv_base.p, R->d_npos, sp));
            }
        }
        rc = reads_finish(R, true, n_words, n_rec, total_bases, uniform, n_npos, ctx);
        if (rc) { err = mf_thread_error(); return rc; }
        if (timing) t_pack += now_s() - t1;
        const double t2 = now_s();
        rc = filter_common(ks, R, threshold, MF_MODE_SCREENED, nullptr, nullptr, 1, nullptr);
        if (rc) { err = mf_thread_error(); return rc; }
        DCHK(launch_store_bits(R->d_bits[R->cur], n_rec, M.file_bits.p, S.rec_base, sp));
        DCHK(hipStreamSynchronize(sp));
        if (timing) t_filter += now_s() - t2;
        return MF_OK;
    }

    // survivors of the first n_emit records of S -> the mate's writer
    int emit(Mate &M, Slab &S, uint64_t n_emit, std::string &err)
    {
        const double t0 = now_s();
        if (n_emit > S.n_rec) n_emit = S.n_rec;
        if (n_emit) {
            const uint8_t *text = S.base - S.carry;
            DCHK(M.out_len.need(n_emit)); DCHK(M.out_off.need(n_emit + 1)); DCHK(M.scan_tmp.need(n_emit / 4096 + 4));
            const uint32_t *other = nm == 2 ? m[&M == &m[0] ? 1 : 0].file_bits.p : nullptr;
            DCHK(launch_out_lens(text, S.line_start.p, n_emit, S.rec_base, M.file_bits.p, other, pair_both ? 1 : 0, M.out_len.p, sp));
            DCHK(launch_scan_u32(M.out_len.p, n_emit, M.out_off.p, M.scan_tmp.p, sp));
            uint64_t bytes = 0;
            DCHK(hipMemcpyAsync(&bytes, M.out_off.p + n_emit, 8, hipMemcpyDeviceToHost, sp));
            DCHK(hipStreamSynchronize(sp));
            if (bytes) {
                DCHK(M.d_out.need(bytes));
                DCHK(launch_gather(text, S.line_start.p, n_emit, M.out_len.p, M.out_off.p, M.d_out.p, sp));
                std::vector<char> host(bytes);
                DCHK(hipMemcpyAsync(host.data(), M.d_out.p, bytes, hipMemcpyDeviceToHost, sp));
                DCHK(hipStreamSynchronize(sp));
                M.out.push(std::move(host));
            }
        }
        if (timing) t_emit += now_s() - t0;
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
            DCHK(hipMemcpy(a.data(), m[0].file_bits.p + w0, k * 4, hipMemcpyDeviceToHost));
            if (nm == 2) DCHK(hipMemcpy(b.data(), m[1].file_bits.p + w0, k * 4, hipMemcpyDeviceToHost));
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
        for (;;) {
            // the mate that is behind in records goes next
            int pick = -1;
            for (int i = 0; i < nm; i++) if (!m[i].eof && (pick < 0 || m[i].rec_done < m[pick].rec_done)) pick = i;
            if (pick < 0) break;
            Mate &M = m[pick];
            std::unique_ptr<Slab> S;
            if (!M.spare.empty()) { S = std::move(M.spare.back()); M.spare.pop_back(); } else S.reset(new Slab());
            const double t0 = now_s();
            int rc = next_text(M, *S, err);
            if (rc) return rc;
            if (timing) t_text += now_s() - t0;
            rc = ingest(M, *S, err);
            if (rc) return rc;
            M.rec_done += S->n_rec;
            if (S->last) M.eof = true;
            M.pending.push_back(std::move(S));
            // write what both mates have decided
            for (int i = 0; i < nm; i++) {
                Mate &A = m[i]; const Mate *B = nm == 2 ? &m[1 - i] : nullptr;
                while (!A.pending.empty()) {
                    Slab &P = *A.pending.front();
                    uint64_t n_emit = P.n_rec;
                    if (B) {
                        const uint64_t end = P.rec_base + P.n_rec;
                        if (B->rec_done < end) { if (!B->eof) break; n_emit = B->rec_done > P.rec_base ? B->rec_done - P.rec_base : 0; }     // pairs end with the shorter file
                    }
                    // the text in front of the NEXT slab of this mate may still be this slab's (a .gz slab's buffer is its own;
                    // a carry longer than the window is copied from it when the next slab is made): keep the newest slab until then
                    if (A.gz && A.pending.size() == 1 && !A.eof && A.carry > GZ_WINDOW) break;
                    rc = emit(A, P, n_emit, err);
                    if (rc) return rc;
                    A.spare.push_back(std::move(A.pending.front())); A.pending.pop_front();
                    if (A.spare.size() > 2) A.spare.erase(A.spare.begin());
                }
            }
        }
        for (int i = 0; i < nm; i++) if (!m[i].pending.empty()) { err = "internal: slabs left unwritten"; return MF_E_HIP; }
        total = nm == 2 ? std::min(m[0].rec_done, m[1].rec_done) : m[0].rec_done;
        return count_kept(total, err);
    }
};

} // namespace

int run_device_ingest(mf_kmerset *ks, const char *fq1, const char *fq2, const char *out1, const char *out2, uint32_t threshold,
                      bool pair_both, int device, uint64_t *kept, uint64_t *total, std::string &err)
{
    Ingest I;
    I.ks = ks; I.threshold = threshold; I.pair_both = pair_both; I.device = device; I.nm = fq2 ? 2 : 1;
    I.timing = getenv("MF_PIPE_TIMING") != nullptr;
    const char *in_path[2] = {fq1, fq2}, *out_path[2] = {out1, out2};
    // ---- is this an input for the device path?
    for (int i = 0; i < I.nm; i++) {
        Mate &M = I.m[i];
        M.path = in_path[i]; M.gz = has_gz_ext(in_path[i]);
        bool regular = false;
        if (!M.map.open(in_path[i], regular)) { err = std::string("Cannot open file ") + in_path[i]; return MF_E_IO; }
        if (!regular || M.map.n == 0 || M.map.n > ((size_t)48 << 30)) return MF_DEVINGEST_DECLINED;
        if (M.gz) {
            const uint8_t *d = M.map.p;
            if (M.map.n < 18 || d[0] != 0x1f || d[1] != 0x8b) return MF_DEVINGEST_DECLINED;      // (gzread hands such a file through; so does the host reader)
            if ((d[3] & 4) && M.map.n >= 18 && d[12] == 'B' && d[13] == 'C') return MF_DEVINGEST_DECLINED;   // BGZF: the host reader decodes its members side by side
        }
    }
    const double t_begin = now_s();
    int rc = get_ctx(device, &I.ctx, 0);
    if (rc) { err = mf_thread_error(); return rc; }
    I.sp = I.ctx->stream;
    for (int i = 0; i < I.nm; i++) {
        Mate &M = I.m[i];
        DCHK(M.d_file.need(M.map.n + 256, false));
        DCHK(hipMemsetAsync(M.d_file.p + M.map.n, 0, 256, I.sp));
        DCHK(hipStreamSynchronize(I.sp));
        rc = M.up.start(M.map.p, M.map.n, M.d_file.p, phys(device), err);
        if (rc) return rc;
        if (M.gz) {
            M.gzs.reset(new GzStream());
            rc = M.gzs->open(M.map.p, M.map.n, M.d_file.p, &M.up, M.path, err);
            if (rc) return rc;
        }
    }
    for (int i = 0; i < I.nm; i++) if (!I.m[i].out.open(out_path[i])) { err = std::string("Cannot open file ") + out_path[i]; return MF_E_IO; }
    rc = I.run(err);
    bool wrote = true;
    for (int i = 0; i < I.nm; i++) wrote = I.m[i].out.close() && wrote;
    if (rc) return rc;
    if (!wrote) { err = std::string("write error on ") + out_path[0]; return MF_E_IO; }
    if (kept) *kept = I.kept;
    if (total) *total = I.total;
    if (I.timing) {
        fprintf(stderr, "[mf device ingest] wall %.3f s | text (upload wait, inflate, link, CRC) %.3f | line index %.3f | pack %.3f | filter %.3f | survivors %.3f", now_s() - t_begin,
                I.t_text, I.t_index, I.t_pack, I.t_filter, I.t_emit);
        for (int i = 0; i < I.nm; i++) if (I.m[i].gzs) fprintf(stderr, " | %s: %llu chunks linked, %llu bytes decoded on the host", i ? "mate 2" : "mate 1", (unsigned long long)I.m[i].gzs->chunks_linked(), (unsigned long long)I.m[i].gzs->gap_bytes());
        fprintf(stderr, "\n");
    }
    return MF_OK;
}

} // namespace mf
