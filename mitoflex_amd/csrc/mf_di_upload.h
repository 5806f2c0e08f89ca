// Device ingest path, part 3 of 8: the uploader -- a .gz file's bytes into the rings of the devices that decode it.
#pragma once
#include "mf_di_streams.h"

namespace mf {
namespace {

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
        n_bufs_ = (int)std::max<uint64_t>(2, std::min<uint64_t>(UP_BUFS_MAX, g_knobs.u64(KN_GZDEV_UPLOAD_BUFS, 2)));
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

} // namespace
} // namespace mf
