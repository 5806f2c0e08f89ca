#include "mf_pipeline.h"
#include "mf_inflate.h"
#include "mf_pinflate.h"
#include "../../include/mitofilter.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <poll.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <thread>
#include <zlib.h>

namespace mf {

// -------------------------------------------------------------- bounded channel
template <class T>
class Channel {
public:
    explicit Channel(size_t cap) : cap_(cap) {}
    bool push(T v)
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_space_.wait(lk, [&] { return q_.size() < cap_ || closed_; });
        if (closed_) return false;
        q_.push_back(std::move(v));
        cv_item_.notify_one();
        return true;
    }
    bool pop(T &v)
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_item_.wait(lk, [&] { return !q_.empty() || done_ || closed_; });
        if (closed_ || q_.empty()) return false;
        v = std::move(q_.front()); q_.pop_front();
        cv_space_.notify_one();
        return true;
    }
    void finish() { std::lock_guard<std::mutex> lk(mu_); done_ = true; cv_item_.notify_all(); }          // no more items
    void abort() { std::lock_guard<std::mutex> lk(mu_); closed_ = true; cv_item_.notify_all(); cv_space_.notify_all(); }
private:
    std::mutex mu_; std::condition_variable cv_item_, cv_space_;
    std::deque<T> q_; size_t cap_; bool done_ = false, closed_ = false;
};

// --------------------------------------------------------- batched FASTQ reader
// Hands out batches of up to `max_records` complete 4-line records.  The reference's conventions
// apply to the stream as a whole (filter/filter_bin/src/main.rs:287-321): a partial record at the
// very end is dropped, CR before LF is stripped.
// text is a malloc'd buffer (no zero fill on growth); recs point into it
struct MappedFile {                       // a plain input file mapped read-only for the run
    const char *p = nullptr; size_t n = 0;
    ~MappedFile() { if (p) munmap(const_cast<char *>(p), n); }
};

// Batch objects are recycled through a pool: their vectors keep their capacity, so a steady-state batch
// touches no freshly mapped memory (page faults on new mappings serialise on the process's mmap lock and
// were the limit of the pack and parse stages with many threads).
template <class T> class Pool : public std::enable_shared_from_this<Pool<T>> {
public:
    std::shared_ptr<T> get()
    {
        T *p = nullptr;
        { std::lock_guard<std::mutex> lk(mu_); if (!free_.empty()) { p = free_.back().release(); free_.pop_back(); } }
        if (!p) p = new T();
        auto self = this->shared_from_this();
        return std::shared_ptr<T>(p, [self](T *x) { x->recycle(); std::lock_guard<std::mutex> lk(self->mu_); self->free_.emplace_back(x); });
    }
    // keep at most `keep` idle objects (the rest is released)
    void trim(size_t keep) { std::lock_guard<std::mutex> lk(mu_); while (free_.size() > keep) free_.pop_back(); }
private:
    std::mutex mu_; std::vector<std::unique_ptr<T>> free_;
};

using RecVec = std::vector<FqRec, DefaultInitAlloc<FqRec>>;

struct MateBatch {
    char *text = nullptr; size_t len = 0, cap = 0;
    RecVec recs;
    std::shared_ptr<MappedFile> map;      // set when recs point into a mapped file instead of `text`
    ~MateBatch() { free(text); }
    void recycle() { recs.clear(); len = 0; map.reset(); }
    // the text buffer of a stream-mode batch (hundreds of MB): 2 MiB aligned and marked for huge pages, like the vectors
    bool reserve(size_t want)
    {
        if (want <= cap) return true;
        size_t nc = cap ? cap : ((size_t)64 << 20);
        while (nc < want) nc *= 2;
        const size_t huge = (size_t)2 << 20;
        nc = (nc + huge - 1) / huge * huge;
        char *p = (char *)aligned_alloc(huge, nc);
        if (!p) return false;
        madvise(p, nc, MADV_HUGEPAGE);
        if (len) memcpy(p, text, len);
        free(text);
        text = p; cap = nc;
        return true;
    }
};

class BatchReader {
public:
    bool open(const char *path, std::string &err)
    {
        if (!path) { f_ = stdin; own_f_ = false; path_ = "<stdin>"; return true; }
        gz_ = has_gz_ext(path);
        if (gz_ && !getenv("MF_ZLIB_INFLATE")) {             // regular .gz file: map it and decode with the streaming decoder
            int fd = ::open(path, O_RDONLY);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
                void *m = st.st_size > 0 ? mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
                if (st.st_size == 0 || m != MAP_FAILED) {
                    gzmap_ = std::make_shared<MappedFile>();
                    if (st.st_size > 0) { gzmap_->p = (const char *)m; gzmap_->n = (size_t)st.st_size; madvise(m, gzmap_->n, MADV_SEQUENTIAL); }
                    // big files: several chunks of the one stream are decoded at a time (mf_pinflate.h)
                    par_gz_ = parse_threads_ >= 3 && gzmap_->n >= ((size_t)16 << 20) && !getenv("MF_SERIAL_INFLATE");
                    if (par_gz_) pinflater_.open((const uint8_t *)gzmap_->p, gzmap_->n, parse_threads_);
                    else inflater_.open((const uint8_t *)gzmap_->p, gzmap_->n);
                    ::close(fd);
                    path_ = path;
                    return true;
                }
            }
            if (fd >= 0) ::close(fd);
        }
        if (gz_) { g_ = gzopen(path, "rb"); if (g_) gzbuffer(g_, 1 << 20); }
        else f_ = fopen(path, "rb");
        if (!g_ && !f_) { err = std::string("Cannot open file ") + path; return false; }
        if (f_) setvbuf(f_, nullptr, _IONBF, 0);             // we read in 16 MiB blocks ourselves
        path_ = path;
        if (f_ && parse_threads_ > 1) {                      // regular plain file: map it and parse segments in parallel
            struct stat st;
            if (fstat(fileno(f_), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
                void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(f_), 0);
                if (m != MAP_FAILED) {
                    map_ = std::make_shared<MappedFile>(); map_->p = (const char *)m; map_->n = (size_t)st.st_size;
                    madvise(m, map_->n, MADV_SEQUENTIAL);
                }
            }
        }
        return true;
    }
    void set_parse_threads(int t) { parse_threads_ = t < 1 ? 1 : t; }
    void set_stop_flag(const std::atomic<bool> *f) { stop_ = f; }
    ~BatchReader() { if (g_) gzclose(g_); if (f_ && own_f_) fclose(f_); }
    // One pass: bytes are read in blocks and lines are cut as they arrive; every fourth line closes a
    // record.  Record fields are kept as offsets while the buffer may still move, pointers afterwards.
    // false at end of stream (no records left) or on error (err set)
    bool next(MateBatch &b, uint64_t max_records, std::string &err)
    {
        if (map_) return next_mapped(b, max_records);
        if (parse_threads_ > 1 && !getenv("MF_SERIAL_PARSE")) return next_stream_indexed(b, max_records, err);
        b.recs.clear(); b.len = 0;
        if (!carry_.empty()) { if (!b.reserve(carry_.size())) { err = "out of memory"; return false; } memcpy(b.text, carry_.data(), carry_.size()); b.len = carry_.size(); carry_.clear(); }
        struct Off { size_t h, s, q; uint32_t hl, sl, ql; };
        std::vector<Off> off; off.reserve(max_records < (1u << 22) ? max_records : (1u << 22));
        size_t pos = 0, line_start = 0; int li = 0; size_t ls[4]; uint32_t ll[4]; size_t cut = 0, rec_end = 0; bool full = false, short_read = false;
        const size_t blk = 16u << 20;
        for (;;) {
            while (pos < b.len) {
                const char *nl = (const char *)memchr(b.text + pos, '\n', b.len - pos);
                if (!nl) { pos = b.len; break; }
                size_t e = (size_t)(nl - b.text), L = e - line_start;
                if (L && b.text[e - 1] == '\r') L--;          // lines() strips "\r\n"
                ls[li] = line_start; ll[li] = (uint32_t)L;
                line_start = pos = e + 1;
                if (++li == 4) {
                    li = 0;
                    off.push_back(Off{ls[0], ls[1], ls[3], ll[0], ll[1], ll[3]});
                    rec_end = pos;
                    if (off.size() == max_records) { cut = pos; full = true; break; }
                }
            }
            if (full || eof_) break;
            // a pipe or stdin whose producer has had nothing more for a while: hand over the complete records (see read_some)
            if (short_read && !off.empty()) { cut = rec_end; full = true; break; }
            if (stopped()) return false;
            if (!b.reserve(b.len + blk)) { err = "out of memory"; return false; }
            bool failed = false;
            idle_ = false;
            const size_t got = read_some(b.text + b.len, blk, err, failed);
            if (failed) return false;
            short_read = idle_;
            b.len += got;
            if (got == 0) eof_ = true;
        }
        if (full) { carry_.assign(b.text + cut, b.text + b.len); b.len = cut; }
        else if (eof_ && li == 3 && line_start < b.len) {
            // the stream ended inside the 4th line of a record without a final LF: lines() still yields it
            size_t L = b.len - line_start; if (L && b.text[b.len - 1] == '\r') L--;
            off.push_back(Off{ls[0], ls[1], line_start, ll[0], ll[1], (uint32_t)L});
        }
        b.recs.resize(off.size());
        for (size_t i = 0; i < off.size(); i++)
            b.recs[i] = FqRec{b.text + off[i].h, b.text + off[i].s, b.text + off[i].q, off[i].hl, off[i].sl, off[i].ql};
        return !b.recs.empty();
    }
private:
    // ---- indexed parsing of a text buffer (a mapped plain file, or the decoded text of a stream batch).
    // The buffer is cut into segments of SEG bytes.  Newlines are counted per segment in parallel; a
    // prefix sum gives every segment the index of the first line that starts in it, hence the index of
    // the first record whose header starts in it.  A range of record indices is then parsed by the
    // segments holding those headers, in parallel, every record written straight to its slot (a record
    // is followed across the segment border).
    size_t SEG = getenv("MF_PARSE_SEG") ? (size_t)strtoull(getenv("MF_PARSE_SEG"), nullptr, 10) : (size_t)(4u << 20);   // bytes per parse segment
    struct LineIndex { std::vector<uint64_t> first_line; uint64_t n_records = 0; };
    // final: the buffer ends where the input ends, so an unterminated last line is a line (lines() yields it)
    void build_index(const char *p, size_t n, bool final, LineIndex &ix)
    {
        const size_t nseg = (n + SEG - 1) / SEG;
        std::vector<uint64_t> cnt(nseg, 0);
        parallel_for(nseg, [&](size_t i) {
            const char *q = p + i * SEG, *e = p + std::min(n, (i + 1) * SEG); uint64_t c = 0;
            while (q < e) { const char *nl = (const char *)memchr(q, '\n', (size_t)(e - q)); if (!nl) break; c++; q = nl + 1; }
            cnt[i] = c;
        });
        // first_line[i]: index of the first line that STARTS inside segment i (a line starts after every '\n' and at 0)
        ix.first_line.assign(nseg + 1, 0);
        uint64_t newlines = 0;
        for (size_t i = 0; i < nseg; i++) {
            const size_t off = i * SEG;
            ix.first_line[i] = newlines + ((off > 0 && p[off - 1] != '\n') ? 1 : 0);
            newlines += cnt[i];
        }
        const uint64_t total_lines = newlines + ((final && n > 0 && p[n - 1] != '\n') ? 1 : 0);
        ix.first_line[nseg] = total_lines;
        ix.n_records = total_lines / 4;                     // a partial record at the very end is dropped / left for the next batch
    }
    // records [r0, r1) of the buffer -> out[0 .. r1-r0); *end_off (optional): offset just behind record r1-1
    void parse_records(const char *p, size_t n, const LineIndex &ix, uint64_t r0, uint64_t r1, FqRec *out, size_t *end_off)
    {
        const size_t nseg = (n + SEG - 1) / SEG;
        auto rec_base = [&](size_t i) { return (ix.first_line[i] + 3) / 4; };      // records whose header starts in segments before i
        // first segment holding record r0's header: the last i with rec_base(i) <= r0
        size_t lo = 0, hi = nseg;
        while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (rec_base(mid) <= r0) lo = mid; else hi = mid; }
        size_t s0 = lo, s1 = s0;
        while (s1 < nseg && rec_base(s1) < r1) s1++;
        parallel_for(s1 - s0, [&](size_t gi) {
            const size_t i = s0 + gi, off = i * SEG, end = std::min(n, off + SEG);
            uint64_t line = ix.first_line[i];
            size_t pos = off;
            if (off > 0 && p[off - 1] != '\n') {            // byte `off` continues an earlier line
                const char *nl = (const char *)memchr(p + off, '\n', n - off);
                if (!nl) return;
                pos = (size_t)(nl - p) + 1;
            }
            // skip to the next header line (index divisible by 4)
            while (pos < n && (line & 3)) {
                const char *nl = (const char *)memchr(p + pos, '\n', n - pos);
                if (!nl) return;
                pos = (size_t)(nl - p) + 1; line++;
            }
            uint64_t rec = line / 4;
            while (pos < end && rec < r1) {                 // this record's header starts inside the segment
                const char *ls[4]; uint32_t ll[4]; int li = 0;
                size_t q = pos;
                for (; li < 4 && q < n; li++) {
                    const char *nl = (const char *)memchr(p + q, '\n', n - q);
                    const size_t e = nl ? (size_t)(nl - p) : n;
                    size_t L = e - q; if (L && p[e - 1] == '\r') L--;      // lines() strips "\r\n"
                    ls[li] = p + q; ll[li] = (uint32_t)L;
                    if (!nl) { q = n; li++; break; }
                    q = e + 1;
                }
                if (li < 4) break;                          // partial record at the very end
                if (rec >= r0) out[rec - r0] = FqRec{ls[0], ls[1], ls[3], ll[0], ll[1], ll[3]};
                if (rec + 1 == r1 && end_off) *end_off = q;
                rec++;
                pos = q;
            }
        });
    }
    // ---- mapped mode: the whole file is indexed once, a batch is a range of record indices
    bool next_mapped(MateBatch &b, uint64_t max_records)
    {
        b.recs.clear(); b.map = map_;
        if (!indexed_) { build_index(map_->p, map_->n, true, map_ix_); indexed_ = true; }
        if (rec_pos_ >= map_ix_.n_records) return false;
        const uint64_t r0 = rec_pos_, r1 = std::min<uint64_t>(map_ix_.n_records, r0 + max_records);
        b.recs.resize((size_t)(r1 - r0));
        parse_records(map_->p, map_->n, map_ix_, r0, r1, b.recs.data(), nullptr);
        rec_pos_ = r1;
        return true;
    }
    // ---- stream mode with several threads (decoded .gz text, pipes): enough text for the batch is read
    // first, then indexed and parsed like a mapped file; what lies behind the last record goes to the next batch
    size_t read_some(char *dst, size_t want, std::string &err, bool &failed)
    {
        if (gzmap_) {
            std::string why;
            const long n = par_gz_ ? pinflater_.read((uint8_t *)dst, want, why) : inflater_.read((uint8_t *)dst, want, why);
            if (n < 0) { err = "gzip read error in " + path_ + ": " + why; failed = true; return 0; }
            return (size_t)n;
        }
        if (gz_) { const int n = gzread(g_, dst, (unsigned)std::min<size_t>(want, (size_t)1 << 30)); if (n < 0) { err = "gzip read error in " + path_; failed = true; return 0; } return (size_t)n; }
        // plain stream (pipe, stdin; the stream is unbuffered).  A pipe hands over at most 64 KiB per read(): keep reading until the
        // block is full, so that batch boundaries do not depend on how the producer's writes happen to arrive -- but if the
        // producer has had nothing for 50 ms, return what is there (idle_ set): the reference works line by line, and with a -t
        // budget it may never need the rest of a slow stream.  The poll also lets the stop flag be seen in good time.
        size_t total = 0;
        for (;;) {
            struct pollfd pfd; pfd.fd = fileno(f_); pfd.events = POLLIN; pfd.revents = 0;
            const int pr = poll(&pfd, 1, 50);                 // a blocked read() cannot be told to stop; a poll that times out can
            if (stopped()) { failed = true; return 0; }       // (no error text: the pipeline is winding down for its own reasons)
            if (pr == 0) { if (total) { idle_ = true; return total; } continue; }
            const ssize_t n = ::read(fileno(f_), dst + total, std::min<size_t>(want - total, (size_t)1 << 30));
            if (n > 0) { total += (size_t)n; if (total == want) return total; continue; }
            if (n == 0) return total;                         // end of the stream (the next call returns 0)
            if (errno != EINTR && errno != EAGAIN) { err = "read error in " + path_; failed = true; return 0; }
        }
    }
    // the pipeline is winding down (error elsewhere, the -t budget is spent, a mate file ran out): stop reading ahead
    bool stopped() const { return stop_ && stop_->load(std::memory_order_relaxed); }
    bool next_stream_indexed(MateBatch &b, uint64_t max_records, std::string &err)
    {
        b.recs.clear(); b.len = 0;
        if (!carry_.empty()) { if (!b.reserve(carry_.size())) { err = "out of memory"; return false; } memcpy(b.text, carry_.data(), carry_.size()); b.len = carry_.size(); carry_.clear(); }
        LineIndex ix;
        size_t target = (size_t)((double)max_records * rec_bytes_est_ * 1.02) + 4096;
        const bool live = !gz_ && !gzmap_;                 // a pipe or stdin: the producer may be slower than we are
        for (;;) {
            bool short_read = false;
            while (!eof_ && b.len < target) {
                if (stopped()) return false;
                const size_t want = std::min<size_t>((size_t)64 << 20, std::max<size_t>((size_t)1 << 16, target - b.len));
                if (!b.reserve(b.len + want)) { err = "out of memory"; return false; }
                bool failed = false;
                idle_ = false;
                const size_t got = read_some(b.text + b.len, want, err, failed);
                if (failed) return false;
                b.len += got;
                if (got == 0) eof_ = true;
                // the producer has had nothing more for a while: hand over the records that are complete instead of waiting for a
                // whole batch (the reference works line by line -- with a -t budget it may never need the rest)
                if (live && idle_ && got > 0) { short_read = true; break; }
            }
            build_index(b.text, b.len, eof_, ix);
            if (ix.n_records >= max_records || eof_ || (short_read && ix.n_records > 0)) break;
            const double per = ix.n_records ? (double)b.len / (double)ix.n_records : rec_bytes_est_ * 2;
            target = b.len + (size_t)((double)(max_records - ix.n_records + 1) * per * 1.05) + ((size_t)1 << 16);
        }
        const uint64_t take = std::min<uint64_t>(max_records, ix.n_records);
        if (take == 0) return false;                        // nothing but a partial record left: dropped
        b.recs.resize((size_t)take);
        size_t cut = b.len;
        parse_records(b.text, b.len, ix, 0, take, b.recs.data(), &cut);
        rec_bytes_est_ = 0.5 * rec_bytes_est_ + 0.5 * ((double)cut / (double)take);
        if (take < ix.n_records || !eof_) carry_.assign(b.text + cut, b.text + b.len);
        b.len = cut;
        return true;
    }
    template <class F> void parallel_for(size_t count, F f)
    {
        std::vector<std::thread> th;
        const size_t T = std::min<size_t>(count, (size_t)parse_threads_);
        std::atomic<size_t> nexti{0};
        for (size_t t = 1; t < T; t++) th.emplace_back([&] { for (size_t i; (i = nexti++) < count;) f(i); });
        for (size_t i; (i = nexti++) < count;) f(i);
        for (auto &x : th) x.join();
    }
    bool gz_ = false, eof_ = false, own_f_ = true; gzFile g_ = nullptr; FILE *f_ = nullptr; std::string path_;
    std::shared_ptr<MappedFile> gzmap_; GzInflater inflater_;   // .gz input: the compressed file, mapped, and its decoder
    ParallelGzReader pinflater_; bool par_gz_ = false;
    std::vector<char> carry_;
    int parse_threads_ = 1;
    const std::atomic<bool> *stop_ = nullptr;
    std::shared_ptr<MappedFile> map_;
    LineIndex map_ix_; bool indexed_ = false;
    uint64_t rec_pos_ = 0;
    double rec_bytes_est_ = 350.0;                      // bytes per record seen so far (stream mode read-ahead)
    bool idle_ = false;                                 // read_some returned early because a live producer had nothing for a while
};

struct PairBatch {
    uint64_t index = 0, n = 0;
    std::shared_ptr<MateBatch> mate[2];
    uint64_t first[2] = {0, 0};            // the pair batch holds records first[m] .. first[m] + n - 1 of mate[m]
    PackedHost packed[2];
    std::vector<uint8_t> keep;
    void recycle() { mate[0].reset(); mate[1].reset(); keep.clear(); index = n = 0; first[0] = first[1] = 0; }
};
using PairPtr = std::shared_ptr<PairBatch>;

int run_fastq_pipeline(const char *fq1, const char *fq2, const char *out1, const char *out2, bool pair_both,
                       int n_devices, int pack_threads, uint64_t batch_reads, const BatchFilterFn &filter,
                       PipelineStats &stats, std::string &err)
{
    const int nm = fq2 ? 2 : 1;
    const char *in_path[2] = {fq1, fq2}, *out_path[2] = {out1, out2};
    // MF_PIPE_TIMING=1: busy seconds of every stage on stderr (diagnostics)
    const bool timing = getenv("MF_PIPE_TIMING") != nullptr;
    std::atomic<uint64_t> t_read[2] = {{0}, {0}}, t_pack{0}, t_dev{0}, t_write[2] = {{0}, {0}};
    auto now_us = [] { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const uint64_t t_start = now_us();
    BatchReader rd[2];
    std::atomic<bool> winding_down{false};          // set with the queues' abort: the readers stop reading ahead
    for (int m = 0; m < nm; m++) {
        rd[m].set_stop_flag(&winding_down);
        rd[m].set_parse_threads(std::max(1, pack_threads / (2 * nm)));     // plain files: segments parsed in parallel
        if (!rd[m].open(in_path[m], err)) return MF_E_IO;
    }
    // create/truncate the outputs up front so that an empty input still leaves files behind
    std::mutex err_mu; int rc = MF_OK;
    auto set_err = [&](int code, const std::string &msg) { std::lock_guard<std::mutex> lk(err_mu); if (rc == MF_OK) { rc = code; err = msg; } };

    // Batch buffers outlive the call (a handful of idle batches, about 1.5 GB for 2 M-record batches): giving a
    // gigabyte back to the kernel and faulting it in again costs 0.1 s per call, which matters to callers that
    // filter file after file (the bim loop).  MF_KEEP_BUFFERS=0 releases everything at the end of every call.
    static std::shared_ptr<Pool<MateBatch>> g_mate_pool = std::make_shared<Pool<MateBatch>>();
    static std::shared_ptr<Pool<PairBatch>> g_pair_pool = std::make_shared<Pool<PairBatch>>();
    const char *kb = getenv("MF_KEEP_BUFFERS");
    const bool keep_buffers = !(kb && kb[0] == '0');
    auto mate_pool = keep_buffers ? g_mate_pool : std::make_shared<Pool<MateBatch>>();
    auto pair_pool = keep_buffers ? g_pair_pool : std::make_shared<Pool<PairBatch>>();
    Channel<std::shared_ptr<MateBatch>> q_read[2] = {Channel<std::shared_ptr<MateBatch>>(2), Channel<std::shared_ptr<MateBatch>>(2)};
    std::vector<std::unique_ptr<Channel<PairPtr>>> q_dev;
    for (int d = 0; d < n_devices; d++) q_dev.emplace_back(new Channel<PairPtr>(2));
    Channel<PairPtr> q_write[2] = {Channel<PairPtr>(4 * (size_t)n_devices + 4), Channel<PairPtr>(4 * (size_t)n_devices + 4)};
    auto abort_all = [&] { winding_down = true; for (auto &q : q_read) q.abort(); for (auto &q : q_dev) q->abort(); for (auto &q : q_write) q.abort(); };

    // ---- readers
    std::vector<std::thread> threads;
    for (int m = 0; m < nm; m++)
        threads.emplace_back([&, m] {
            for (;;) {
                auto b = mate_pool->get();
                std::string e;
                const uint64_t t0 = now_us();
                const bool got = rd[m].next(*b, batch_reads, e);
                t_read[m] += now_us() - t0;
                if (!got) { if (!e.empty()) { set_err(MF_E_IO, e); abort_all(); } break; }
                if (!q_read[m].push(b)) break;
            }
            q_read[m].finish();
        });

    // ---- zip mates, pack, deal to devices (whole pairs stay together).  Mates are paired BY RECORD COUNT: the batches of the two
    // readers need not be of equal size (a pipe whose producer pauses hands over what it has), so what is left of the longer
    // batch is paired with the other mate's next one.  A mate has run out only when its reader has (zip semantics: the shorter
    // file bounds the pair count, filter/filter_bin/src/main.rs:214).
    threads.emplace_back([&] {
        uint64_t idx = 0;
        std::shared_ptr<MateBatch> cur[2]; uint64_t pos[2] = {0, 0};
        for (;;) {
            bool ok = true;
            for (int m = 0; m < nm && ok; m++)
                while (ok && (!cur[m] || pos[m] == cur[m]->recs.size())) { cur[m].reset(); pos[m] = 0; ok = q_read[m].pop(cur[m]); }
            if (!ok) break;
            auto pb = pair_pool->get();
            pb->n = cur[0]->recs.size() - pos[0];
            if (nm == 2 && cur[1]->recs.size() - pos[1] < pb->n) pb->n = cur[1]->recs.size() - pos[1];
            for (int m = 0; m < nm; m++) { pb->mate[m] = cur[m]; pb->first[m] = pos[m]; pos[m] += pb->n; }
            pb->index = idx;
            const uint64_t t0 = now_us();
            if (nm == 2) {
                std::thread t([&] { pack_records(pb->mate[1]->recs.data() + pb->first[1], pb->n, pack_threads > 1 ? pack_threads / 2 : 1, pb->packed[1]); });
                pack_records(pb->mate[0]->recs.data() + pb->first[0], pb->n, pack_threads > 1 ? pack_threads - pack_threads / 2 : 1, pb->packed[0]);
                t.join();
            } else pack_records(pb->mate[0]->recs.data() + pb->first[0], pb->n, pack_threads, pb->packed[0]);
            t_pack += now_us() - t0;
            if (!q_dev[idx % n_devices]->push(pb)) break;
            idx++;
        }
        winding_down = true;                              // readers may still be producing past the end of the shorter file
        for (auto &q : q_read) q.abort();
        for (auto &q : q_dev) q->finish();
    });

    // ---- one worker per device: H2D + kernels + D2H, then the pair rule
    std::mutex stat_mu; int dev_alive = n_devices;
    for (int d = 0; d < n_devices; d++)
        threads.emplace_back([&, d] {
            PairPtr pb;
            while (q_dev[d]->pop(pb)) {
                std::vector<uint32_t> bits[2];
                bool ok = true;
                for (int m = 0; m < nm && ok; m++) {
                    std::string e;
                    const uint64_t t0 = now_us();
                    const int r = filter(d, pb->packed[m], pb->n, bits[m], e);
                    t_dev += now_us() - t0;
                    if (r != MF_OK) { set_err(r, e); abort_all(); ok = false; }
                }
                if (!ok) break;
                pb->keep.assign(pb->n ? pb->n : 1, 0);
                uint64_t kc = 0;
                for (uint64_t i = 0; i < pb->n; i++) {
                    const int a = (bits[0][i >> 5] >> (i & 31)) & 1, b = nm == 2 ? (bits[1][i >> 5] >> (i & 31)) & 1 : 0;
                    const uint8_t k = (uint8_t)(nm == 2 ? (pair_both ? (a & b) : (a | b)) : a);
                    pb->keep[i] = k; kc += k;
                }
                { std::lock_guard<std::mutex> lk(stat_mu); stats.kept += kc; stats.total += pb->n; stats.batches++; }
                for (int m = 0; m < nm; m++) if (!q_write[m].push(pb)) { ok = false; break; }
                if (!ok) break;
            }
            std::lock_guard<std::mutex> lk(stat_mu);
            if (--dev_alive == 0) for (int m = 0; m < nm; m++) q_write[m].finish();
        });

    // ---- ordered writers (devices may finish out of order)
    for (int m = 0; m < nm; m++)
        threads.emplace_back([&, m] {
            OutFile of;
            if (!of.open(out_path[m])) { set_err(MF_E_IO, std::string("Cannot open file ") + out_path[m]); abort_all(); return; }
            std::map<uint64_t, PairPtr> pending; uint64_t next = 0; bool ok = true;
            std::vector<char> buf; buf.reserve((1u << 22) + (1u << 16));
            auto drain = [&] {
                if (buf.empty()) return;
                if (!of.write(buf.data(), buf.size())) ok = false;
                buf.clear();
            };
            PairPtr pb;
            while (ok && q_write[m].pop(pb)) {
                pending[pb->index] = pb;
                while (ok && !pending.empty() && pending.begin()->first == next) {
                    const uint64_t t0w = now_us();
                    struct Acc { std::atomic<uint64_t> &a; uint64_t t0; std::function<uint64_t()> now; ~Acc() { a += now() - t0; } } acc{t_write[m], t0w, now_us};
                    PairPtr cur = pending.begin()->second; pending.erase(pending.begin()); next++;
                    const FqRec *recs = cur->mate[m]->recs.data() + cur->first[m];
                    for (uint64_t i = 0; i < cur->n && ok; i++) {
                        if (!cur->keep[i]) continue;
                        const FqRec &r = recs[i];                 // header / seq / "+" / qual (filter_bin main.rs:261-268)
                        buf.insert(buf.end(), r.h, r.h + r.hl); buf.push_back('\n');
                        buf.insert(buf.end(), r.s, r.s + r.sl); buf.push_back('\n'); buf.push_back('+'); buf.push_back('\n');
                        buf.insert(buf.end(), r.q, r.q + r.ql); buf.push_back('\n');
                        if (buf.size() > (1u << 22)) drain();
                    }
                }
            }
            drain();
            ok = of.close() && ok;
            if (!ok) { set_err(MF_E_IO, std::string("write error on ") + out_path[m]); abort_all(); }
        });

    for (auto &t : threads) t.join();
    const uint64_t t_joined = now_us();
    if (keep_buffers) { mate_pool->trim(8); pair_pool->trim(4); }   // every batch is back by now
    mate_pool.reset(); pair_pool.reset();                     // (not kept: the buffers are released here)
    if (timing)
        fprintf(stderr, "[mf pipeline] wall %.3f s (+%.3f s releasing buffers) | read %.3f %.3f | pack %.3f | device %.3f | write %.3f %.3f | batches %llu\n",
                (t_joined - t_start) / 1e6, (now_us() - t_joined) / 1e6, t_read[0] / 1e6, t_read[1] / 1e6, t_pack / 1e6, t_dev / 1e6, t_write[0] / 1e6, t_write[1] / 1e6,
                (unsigned long long)stats.batches);
    return rc;
}

// ================================================================ FASTQ quality filter (filter_v2)
// Same readers; the decision stage is one thread because the reference's -t budget is sequential (it stops at the first read
// that overflows it: filter/filter_bin/src/main.rs:254-259).  What is data parallel -- counting N and low-quality bytes,
// hashing the sequences, and the de-duplication set itself ("first occurrence wins" is a minimum over file indices, which needs
// no order of execution) -- runs on the GPU over the raw text of each batch.
bool utf8_valid(const char *p, size_t n)
{
    const unsigned char *s = (const unsigned char *)p; size_t i = 0;
    while (i < n) {
        if (i + 8 <= n) { uint64_t v; memcpy(&v, s + i, 8); if (!(v & 0x8080808080808080ULL)) { i += 8; continue; } }
        const unsigned char c = s[i];
        if (c < 0x80) { i++; continue; }
        int len; uint32_t cp, mn;
        if ((c & 0xE0) == 0xC0) { len = 2; cp = c & 0x1F; mn = 0x80; }
        else if ((c & 0xF0) == 0xE0) { len = 3; cp = c & 0x0F; mn = 0x800; }
        else if ((c & 0xF8) == 0xF0) { len = 4; cp = c & 0x07; mn = 0x10000; }
        else return false;
        if (i + len > n) return false;
        for (int k = 1; k < len; k++) { if ((s[i + k] & 0xC0) != 0x80) return false; cp = (cp << 6) | (s[i + k] & 0x3F); }
        if (cp < mn || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
        i += len;
    }
    return true;
}

namespace {

struct QualBatch {
    uint64_t index = 0, n = 0;
    std::shared_ptr<MateBatch> mate[2];
    uint64_t first[2] = {0, 0};            // records first[m] .. of mate[m] (mates are paired by record count, see run_fastq_pipeline)
    std::vector<uint8_t> keep;
    bool last = false;
    void recycle() { mate[0].reset(); mate[1].reset(); keep.clear(); index = n = 0; last = false; first[0] = first[1] = 0; }
};

// fn(lo, hi) over [0, n) cut into one contiguous chunk per worker
template <class F> void parallel_chunks(uint64_t n, int threads, F fn)
{
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > n / 4096 + 1) threads = (int)(n / 4096 + 1);      // not worth a thread below a few thousand records
    if (threads == 1) { fn((uint64_t)0, n); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < threads; t++) th.emplace_back(fn, n * (uint64_t)t / threads, n * (uint64_t)(t + 1) / threads);
    fn((uint64_t)0, n / threads);
    for (auto &x : th) x.join();
}
using QualPtr = std::shared_ptr<QualBatch>;

} // namespace

int run_qualfilter_pipeline(const char *fq1, const char *fq2, const char *out1, const char *out2, const QualParams &P,
                            int threads, uint64_t batch_reads, const QualScanFn &scan, const QualDedupFn &dedup, QualStats &stats,
                            std::string &err)
{
    const int nm = fq2 ? 2 : 1;
    const char *in_path[2] = {fq1, fq2}, *out_path[2] = {out1, out2};
    const bool timing = getenv("MF_PIPE_TIMING") != nullptr;      // busy seconds of every stage on stderr (diagnostics)
    std::atomic<uint64_t> t_read[2] = {{0}, {0}}, t_check{0}, t_scan{0}, t_decide{0}, t_dedup{0}, t_wait_in{0}, t_wait_out{0}, t_write[2] = {{0}, {0}};
    auto now_us = [] { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const uint64_t t_start = now_us();
    BatchReader rd[2];
    std::atomic<bool> winding_down{false};          // set when the decision loop stops (budget spent, panic, short mate file, error):
    for (int m = 0; m < nm; m++) {                  // the reference exits at its `break`, so the readers must not sit in a read
        rd[m].set_stop_flag(&winding_down);
        rd[m].set_parse_threads(std::max(1, threads / (2 * nm)));
        if (!rd[m].open(in_path[m], err)) return MF_E_IO;
    }
    std::mutex err_mu; int rc = MF_OK;
    auto set_err = [&](int code, const std::string &msg) { std::lock_guard<std::mutex> lk(err_mu); if (rc == MF_OK) { rc = code; err = msg; } };
    auto mate_pool = std::make_shared<Pool<MateBatch>>();
    auto qual_pool = std::make_shared<Pool<QualBatch>>();
    Channel<std::shared_ptr<MateBatch>> q_read[2] = {Channel<std::shared_ptr<MateBatch>>(2), Channel<std::shared_ptr<MateBatch>>(2)};
    Channel<QualPtr> q_write[2] = {Channel<QualPtr>(4), Channel<QualPtr>(4)};
    auto abort_all = [&] { winding_down = true; for (auto &q : q_read) q.abort(); for (auto &q : q_write) q.abort(); };

    std::vector<std::thread> th;
    for (int m = 0; m < nm; m++)
        th.emplace_back([&, m] {
            for (;;) {
                auto b = mate_pool->get();
                std::string e;
                const uint64_t t0 = now_us();
                const bool got = rd[m].next(*b, batch_reads, e);
                t_read[m] += now_us() - t0;
                if (!got) { if (!e.empty()) { set_err(MF_E_IO, e); abort_all(); } break; }
                if (!q_read[m].push(b)) break;
            }
            q_read[m].finish();
        });

    // ---- decisions (sequential semantics), GPU counting per batch
    th.emplace_back([&] {
        uint64_t budget = 0, idx = 0; bool stop = false;
        const uint64_t L = P.end - P.start;
        std::vector<uint32_t> nc[2], bc[2]; std::vector<uint8_t> alive, dupf;      // per-batch scratch, capacity kept
        std::vector<QualSpan, DefaultInitAlloc<QualSpan>> sp;
        std::shared_ptr<MateBatch> cur[2]; uint64_t pos[2] = {0, 0};
        while (!stop) {
            bool ok = true;
            const uint64_t tp0 = now_us();
            for (int m = 0; m < nm && ok; m++)
                while (ok && (!cur[m] || pos[m] == cur[m]->recs.size())) { cur[m].reset(); pos[m] = 0; ok = q_read[m].pop(cur[m]); }
            t_wait_in += now_us() - tp0;
            if (!ok) break;                                   // a reader has run out: the shorter file bounds the pair count
            auto qb = qual_pool->get();
            uint64_t n = cur[0]->recs.size() - pos[0];
            if (nm == 2 && cur[1]->recs.size() - pos[1] < n) n = cur[1]->recs.size() - pos[1];
            for (int m = 0; m < nm; m++) { qb->mate[m] = cur[m]; qb->first[m] = pos[m]; pos[m] += n; }
            FqRec *const mrec[2] = {qb->mate[0]->recs.data() + qb->first[0], nm == 2 ? qb->mate[1]->recs.data() + qb->first[1] : nullptr};
            // the cut (main.rs:222-233, 291-299) and the first record at which the reference would panic:
            // `drain(..start)` past the end of a string, or a line that is not valid UTF-8
            // Records are independent here, so the batch is cut into chunks; each chunk stops at its first
            // offender and the earliest one bounds the batch (records past it are never looked at again).
            const uint64_t tc0 = now_us();
            std::atomic<uint64_t> first_bad{n};
            parallel_chunks(n, threads, [&](uint64_t lo, uint64_t hi) {
                for (uint64_t i = lo; i < hi; i++) {
                    bool bad = false;
                    for (int m = 0; m < nm && !bad; m++) {
                        const FqRec &r = mrec[m][i];
                        // header, sequence and quality are unwrapped (main.rs:214-216, 287-289) and panic on invalid UTF-8; the
                        // '+' line is bound to `_` and never unwrapped: lines() yields an Err for it and carries on
                        bad = !utf8_valid(r.h, r.hl) || !utf8_valid(r.s, r.sl) || !utf8_valid(r.q, r.ql);
                    }
                    if (!bad && P.start) {
                        for (int m = 0; m < nm; m++) bad = bad || P.start > mrec[m][i].sl;      // seq1, seq2 first
                        for (int m = 0; m < nm; m++) bad = bad || P.start > mrec[m][i].ql;      // then qua1, qua2
                    }
                    if (bad) {
                        uint64_t cur = first_bad.load();
                        while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {}
                        return;
                    }
                    for (int m = 0; m < nm; m++) {
                        FqRec &r = mrec[m][i];
                        if (P.start) { r.s += P.start; r.sl -= (uint32_t)P.start; r.q += P.start; r.ql -= (uint32_t)P.start; }
                        if (P.end) { if (r.sl > L) r.sl = (uint32_t)L; if (r.ql > L) r.ql = (uint32_t)L; }
                    }
                }
            });
            const uint64_t n_ok = first_bad.load();
            t_check += now_us() - tc0;
            const uint64_t ts0 = now_us();
            const bool panicked = n_ok < n;
            // GPU: counts (and hashes of mate 1 when deduplicating)
            if (!P.trunc && n_ok) {
                for (int m = 0; m < nm && ok; m++) {
                    const FqRec *recs = mrec[m];
                    const char *base = recs[0].h, *endp = recs[n_ok - 1].q + recs[n_ok - 1].ql;
                    if ((size_t)(endp - base) >= 0xFFFFFFF0ull) { set_err(MF_E_ARG, "batch larger than 4 GiB: lower MF_BATCH_READS"); ok = false; break; }
                    sp.resize(n_ok);
                    parallel_chunks(n_ok, threads, [&](uint64_t lo, uint64_t hi) {
                        for (uint64_t i = lo; i < hi; i++)
                            sp[i] = QualSpan{(uint32_t)(recs[i].s - base), recs[i].sl, (uint32_t)(recs[i].q - base), recs[i].ql};
                    });
                    nc[m].resize(n_ok); bc[m].resize(n_ok);
                    std::string e;
                    const int r = scan(base, (size_t)(endp - base), sp.data(), (uint32_t)n_ok, P.quality, nc[m].data(), bc[m].data(),
                                       m == 0 && P.dedup, e);
                    if (r != MF_OK) { set_err(r, e); ok = false; }
                }
                if (!ok) { abort_all(); break; }
            }
            t_scan += now_us() - ts0;
            const uint64_t td0 = now_us();
            qb->keep.assign(n ? n : 1, 0);
            // what does not depend on other records: too many N, too many low qualities (in parallel)
            if (!P.trunc) {
                alive.resize(n_ok ? n_ok : 1);
                parallel_chunks(n_ok, threads, [&](uint64_t lo, uint64_t hi) {
                    for (uint64_t i = lo; i < hi; i++) {
                        const FqRec &r1 = mrec[0][i];
                        bool drop = nc[0][i] > P.ns || (nm == 2 && nc[1][i] > P.ns);                       // main.rs:236, 302
                        if (!drop) {
                            // PE: both mates against a cutoff from seq1's length (main.rs:239-243); SE: from the quality string (:305)
                            const float cf = (float)(nm == 2 ? r1.sl : r1.ql) * P.limit;
                            const uint64_t cutoff = !(cf > 0.0f) ? 0 : (cf >= 18446744073709551616.0f ? ~0ull : (uint64_t)cf);
                            drop = bc[0][i] >= cutoff || (nm == 2 && bc[1][i] >= cutoff);
                        }
                        alive[i] = !drop;
                    }
                });
                if (P.dedup && n_ok) {                                                                     // main.rs:244-250
                    dupf.resize(n_ok);
                    std::string e;
                    const uint64_t tdd = now_us();
                    const int r = dedup(alive.data(), (uint32_t)n_ok, dupf.data(), e);
                    t_dedup += now_us() - tdd;
                    if (r != MF_OK) { set_err(r, e); abort_all(); break; }
                }
            }
            for (uint64_t i = 0; i < n_ok; i++) {
                if (!P.trunc && (!alive[i] || (P.dedup && dupf[i]))) continue;
                if (P.trim) { budget += mrec[0][i].sl; if (budget > P.trim) { stop = true; break; } }    // main.rs:254-259
                qb->keep[i] = 1; stats.kept++;
            }
            stats.total += n_ok;
            t_decide += now_us() - td0;
            qb->n = n_ok; qb->index = idx++;
            if (panicked) { stats.panicked = true; stop = true; }
            const uint64_t tq0 = now_us();
            for (int m = 0; m < nm; m++) if (!q_write[m].push(qb)) { stop = true; break; }
            t_wait_out += now_us() - tq0;
        }
        winding_down = true;
        for (auto &q : q_read) q.abort();
        for (auto &q : q_write) q.finish();
    });

    // ---- writers: header / cut seq / "+" / cut qual (main.rs:261-268, 317-321); nullptr path = stdout
    for (int m = 0; m < nm; m++)
        th.emplace_back([&, m] {
            const bool to_stdout = out_path[m] == nullptr;
            OutFile of;
            if (!of.open(out_path[m])) { set_err(MF_E_IO, std::string("Cannot open file ") + out_path[m]); abort_all(); return; }
            bool ok = true;
            std::vector<char> buf; buf.reserve((1u << 22) + (1u << 16));
            auto drain = [&] {
                if (buf.empty()) return;
                if (!of.write(buf.data(), buf.size())) ok = false;
                buf.clear();
            };
            QualPtr qb;
            while (ok && q_write[m].pop(qb)) {
                const uint64_t tw0 = now_us();
                struct Acc { std::atomic<uint64_t> &a; uint64_t t0; std::function<uint64_t()> now; ~Acc() { a += now() - t0; } } acc{t_write[m], tw0, now_us};
                const FqRec *recs = qb->mate[m]->recs.data() + qb->first[m];
                for (uint64_t i = 0; i < qb->n && ok; i++) {
                    if (!qb->keep[i]) continue;
                    const FqRec &r = recs[i];
                    buf.insert(buf.end(), r.h, r.h + r.hl); buf.push_back('\n');
                    buf.insert(buf.end(), r.s, r.s + r.sl); buf.push_back('\n'); buf.push_back('+'); buf.push_back('\n');
                    buf.insert(buf.end(), r.q, r.q + r.ql); buf.push_back('\n');
                    if (buf.size() > (1u << 22)) drain();
                }
            }
            drain();
            ok = of.close() && ok;
            if (!ok) { set_err(MF_E_IO, std::string("write error on ") + (to_stdout ? "<stdout>" : out_path[m])); abort_all(); }
        });

    for (auto &t : th) t.join();
    if (timing)
        fprintf(stderr, "[mf qualfilter] wall %.3f s | read %.3f %.3f | decision thread: waiting for input %.3f, validate %.3f, scan (H2D, kernels, D2H) %.3f, decide %.3f (of which dedup on the device %.3f), waiting for the writers %.3f | write %.3f %.3f\n",
                (now_us() - t_start) / 1e6, t_read[0] / 1e6, t_read[1] / 1e6, t_wait_in / 1e6, t_check / 1e6, t_scan / 1e6, t_decide / 1e6, t_dedup / 1e6, t_wait_out / 1e6, t_write[0] / 1e6, t_write[1] / 1e6);
    return rc;
}

} // namespace mf
