// Host-side plumbing of the quality filter's job on the device ingest path (mf_devingest.cpp: the reference's filter_v2,
// filter/filter_bin/src/main.rs:188-323): per-record arrays that several threads fill and read a piece at a time, the pool of
// chunks the output comes down through, the writer of an output file.  No GPU calls in here (the chunk pool takes its allocator from
// the caller): tests/native/qualsink_check.cpp runs it on the CPU, under ThreadSanitizer too.
#pragma once
#include "mf_host.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <errno.h>
#include <fcntl.h>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdint.h>
#include <string.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace mf {

inline double qs_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// one value per record of a file, written and read a piece at a time by several threads: segments of 2^20 records that never move
template <class T> class SegArray {
public:
    SegArray() : tab_(new std::atomic<T *>[NSEG]) { for (size_t i = 0; i < NSEG; i++) tab_[i] = nullptr; }
    ~SegArray() { for (size_t i = 0; i < NSEG; i++) delete[] tab_[i].load(); }
    bool put(uint64_t r0, uint64_t n, const T *src)
    {
        if (!n) return true;
        if ((r0 + n - 1) / SEG >= NSEG) return false;
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (uint64_t g = r0 / SEG; g <= (r0 + n - 1) / SEG; g++) if (!tab_[g].load()) { T *q = new (std::nothrow) T[SEG]; if (!q) return false; tab_[g] = q; }
        }
        for (uint64_t i = 0; i < n;) { const uint64_t g = (r0 + i) / SEG, o = (r0 + i) % SEG, c = std::min<uint64_t>(n - i, SEG - o); memcpy(tab_[g].load() + o, src + i, c * sizeof(T)); i += c; }
        return true;
    }
    void get(uint64_t r0, uint64_t n, T *dst) const          // (of records that have been put)
    {
        for (uint64_t i = 0; i < n;) { const uint64_t g = (r0 + i) / SEG, o = (r0 + i) % SEG, c = std::min<uint64_t>(n - i, SEG - o); memcpy(dst + i, tab_[g].load() + o, c * sizeof(T)); i += c; }
    }
private:
    static constexpr uint64_t SEG = (uint64_t)1 << 20; static constexpr size_t NSEG = (size_t)1 << 16;
    std::unique_ptr<std::atomic<T *>[]> tab_; std::mutex mu_;
};

// pinned buffers that the text of the kept records passes through on its way from the device to an output file: a consumer takes one,
// copies a chunk of a piece's output down into it and hands it to the file's writer, which gives it back
class OutChunks {
public:
    // (made one after the other by a thread of its own while the call is being set up and the first pieces are on their way: pinning memory
    // takes its time -- more of it while streams and device buffers are being made -- and the first chunk is wanted long before the last)
    // alloc(bytes) -> a buffer or nullptr; release(p)  (the device path: pinned host memory)
    void init(size_t chunk, int n, std::function<void *(size_t)> alloc, std::function<void(void *)> release)
    {
        chunk_ = chunk; release_ = release;
        maker_ = std::thread([this, n, alloc] {
            for (int i = 0; i < n; i++) {
                { std::lock_guard<std::mutex> lk(mu_); if (abort_) break; }
                void *q = alloc(chunk_);
                { std::lock_guard<std::mutex> lk(mu_); if (!q) { if (all_.empty()) failed_ = true; done_ = true; } else { all_.push_back((uint8_t *)q); free_.push_back((uint8_t *)q); } }
                cv_.notify_all();
                if (!q) return;
            }
            { std::lock_guard<std::mutex> lk(mu_); done_ = true; }
            cv_.notify_all();
        });
    }
    size_t chunk() const { return chunk_; }
    uint8_t *take(bool *alloc_failed)          // nullptr: not one chunk could be allocated (*alloc_failed), or the run is being abandoned
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return abort_ || !free_.empty() || (done_ && all_.empty()); });
        *alloc_failed = failed_;
        if (abort_ || free_.empty()) return nullptr;
        uint8_t *p = free_.back(); free_.pop_back();
        return p;
    }
    void give(uint8_t *p) { { std::lock_guard<std::mutex> lk(mu_); free_.push_back(p); } cv_.notify_one(); }
    void abort() { { std::lock_guard<std::mutex> lk(mu_); abort_ = true; } cv_.notify_all(); }
    ~OutChunks() { abort(); if (maker_.joinable()) maker_.join(); for (uint8_t *p : all_) release_(p); }
private:
    std::mutex mu_; std::condition_variable cv_; std::vector<uint8_t *> free_, all_; size_t chunk_ = 0;
    std::thread maker_; bool done_ = false, abort_ = false, failed_ = false; std::function<void(void *)> release_;
};

// An output file of the quality filter.  Nearly every record is written, so what goes out is as large as the text that came in, and
// writing it is what the job waits for: a thread per file does nothing else (a second one would queue behind the first on the
// inode's lock).  The consumers hand it chunks with their places in the file; a regular file takes them as they come (pwrite),
// anything else -- standard output, a pipe, a .gz (compressed by OutFile as the reference's GzEncoder would) -- in order.
class QSink {
public:
    bool open(const char *path, OutChunks *pool)
    {
        pool_ = pool;
        bool ok = false;
        // (what the path names is looked at BEFORE it is opened: opening a named pipe only to find out that it is one, and closing it again,
        // shows its reader an end of file -- the reader leaves, and the second open waits for a reader for ever; tests/native/ingest_check.cpp qual_pipes)
        struct stat sb;
        if (path && !has_gz_ext(path) && (stat(path, &sb) != 0 || S_ISREG(sb.st_mode))) {
            fd_ = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
            if (fd_ < 0) return false;
            if (fstat(fd_, &sb) == 0 && S_ISREG(sb.st_mode)) { direct_ = true; ok = true; }
            else { ::close(fd_); fd_ = -1; }
        }
        if (!ok) ok = of_.open(path);
        if (ok) th_ = std::thread([this] { run(); });
        return ok;
    }
    // A sink that writes in order wants its chunks TAKEN from the pool in order too: a consumer whose bytes lie further back could
    // otherwise fill the pool with chunks that cannot be written yet while the one whose bytes come next waits for a chunk.  So: before
    // taking chunks for the bytes at `off`, wait until everything in front of them has been pushed.  (false: the run is being abandoned)
    bool wait_turn(uint64_t off)
    {
        if (direct_) return true;
        std::unique_lock<std::mutex> lk(mu_);
        turn_.wait(lk, [&] { return abort_ || pushed_ == off; });
        return !abort_;
    }
    void push(uint64_t off, uint8_t *p, size_t n)          // p: a chunk of the pool, given back when written
    {
        { std::lock_guard<std::mutex> lk(mu_); q_.emplace(off, Item{p, n}); if (!direct_) pushed_ = off + n; }
        cv_.notify_one(); turn_.notify_all();
    }
    bool ok() { std::lock_guard<std::mutex> lk(mu_); return ok_; }
    void abort() { { std::lock_guard<std::mutex> lk(mu_); abort_ = true; } cv_.notify_all(); turn_.notify_all(); }
    double busy() const { return busy_; }
    bool close()          // everything pushed is written (in order: up to the first gap) unless aborted
    {
        if (th_.joinable()) { { std::lock_guard<std::mutex> lk(mu_); fin_ = true; } cv_.notify_all(); th_.join(); }
        if (direct_) { const bool c = fd_ < 0 || ::close(fd_) == 0; fd_ = -1; return c && ok_; }
        return of_.close() && ok_;
    }
    ~QSink() { if (th_.joinable()) { abort(); { std::lock_guard<std::mutex> lk(mu_); fin_ = true; } cv_.notify_all(); th_.join(); } if (fd_ >= 0) ::close(fd_); }
private:
    struct Item { uint8_t *p; size_t n; };
    void run()
    {
        for (;;) {
            uint64_t off = 0; Item it{nullptr, 0}; bool drop = false;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return abort_ || fin_ || (!q_.empty() && (direct_ || q_.begin()->first == next_)); });
                if (q_.empty()) { if (fin_ || abort_) return; continue; }
                if (!direct_ && !abort_ && q_.begin()->first != next_) { if (fin_) abort_ = true; else continue; }      // (closed with a gap: a failed run)
                off = q_.begin()->first; it = q_.begin()->second; q_.erase(q_.begin());
                drop = abort_ || !ok_;
            }
            if (!drop) {
                const double t0 = qs_now();
                bool w = true;
                if (direct_) {
                    const uint8_t *p = it.p; size_t n = it.n; uint64_t o = off;
                    while (n) {
                        const ssize_t k = pwrite(fd_, p, n, (off_t)o);
                        if (k < 0) { if (errno == EINTR) continue; w = false; break; }
                        p += k; n -= (size_t)k; o += (uint64_t)k;
                    }
                } else w = of_.write((const char *)it.p, it.n);
                busy_ += qs_now() - t0;
                std::lock_guard<std::mutex> lk(mu_);
                if (!w) ok_ = false;
                next_ = off + it.n;
            }
            pool_->give(it.p);
        }
    }
    int fd_ = -1; bool direct_ = false; OutFile of_; OutChunks *pool_ = nullptr;
    std::thread th_; std::mutex mu_; std::condition_variable cv_, turn_; std::map<uint64_t, Item> q_;
    uint64_t next_ = 0, pushed_ = 0; bool ok_ = true, abort_ = false, fin_ = false; double busy_ = 0;
};

} // namespace mf
