// Device ingest path, part 2 of 8: the streams of the path (made once per process and device by a maker thread), and the text
// buffers and pieces that travel from the producers to the consumers.
#pragma once
#include "mf_di_pool.h"

namespace mf {
namespace {

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
        int reserve = (int)g_knobs.u64(KN_GZDEV_RESERVED_CUS, 32);
        reserve = std::max(8, std::min(n_cu / 2, reserve)) & ~7;
        d->mask.assign((size_t)words, 0);
        for (int b = 0; b < n_cu - reserve; b++) d->mask[b / 32] |= 1u << (b % 32);
        d->mask_rest.resize(d->mask.size());
        for (size_t i = 0; i < d->mask.size(); i++) d->mask_rest[i] = ~d->mask[i];
        if (n_cu % 32) d->mask_rest.back() &= (1u << (n_cu % 32)) - 1;
        d->words = (uint32_t)words; d->n_cu = n_cu;
        d->masked = want_masks && n_cu >= 64 && !g_knobs.is_set(KN_GZDEV_NO_CUMASK);
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
        const bool profiled = (pre && strstr(pre, "rocprof")) || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") || getenv("ROCP_TOOL_LIBRARIES") || g_knobs.is_set(KN_GZDEV_DESTROY_STREAMS_AT_EXIT);
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

} // namespace
} // namespace mf
