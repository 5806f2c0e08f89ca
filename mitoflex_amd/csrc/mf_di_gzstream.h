// Device ingest path, part 4 of 8: one gzip file decoded on the device(s) -- set-up, the slabs in flight, windows, CRC (GzStream::next, the
// producer's step, is in mf_di_gznext.h).
#pragma once
#include "mf_di_upload.h"

namespace mf {
namespace {

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
        uint32_t NSLAB = std::max<uint32_t>(1, (uint32_t)g_knobs.u64(KN_GZDEV_SLABS_IN_FLIGHT, nslab));
        data_ = data; size_ = size; path_ = path; slots_ = slots; pad_ = TEXT_FRONT + carry_room; stop_ = stop;
        const uint32_t nl = (uint32_t)devices.size();
        const bool big = budget == 0 || budget >= ((uint64_t)2 << 30);          // (a mate's share of a call that plans for several gigabytes)
        dec_limit_ = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(GZ_NSTREAM, g_knobs.u64(KN_GZDEV_DEC_STREAMS, big ? GZ_NSTREAM : 3)));
        // chunks: large enough that a slab's fixed costs stay small, small enough that a file keeps the chip busy.  Measured over 0.1 / 0.3 / 1 / 3 GB of
        // .gz a mate x {64, 96, 128, 192, 256} KiB (profiles/r05/f_chunk_size_probe.txt, tools/chunk_size_probe.sh): 64 KiB is the fastest up to
        // 0.3 GB, 96 KiB at 1 GB (SE 0.087 s against 0.101 with 64 KiB and 0.123 with 256; PE 0.163 against 0.207), 192 KiB at 3 GB -- the file's
        // size / 10 923, between 64 and 192 KiB; the quality filter is fastest with 64 KiB at every size (0.44 s against 0.70-0.81 at 1 GB).
        size_t dflt = small_chunks ? (size_t)64 << 10 : size / 10923;
        dflt = std::min<size_t>(std::max<size_t>(dflt, (size_t)64 << 10), (size_t)192 << 10) & ~(size_t)4095;
        chunk_ = (size_t)g_knobs.u64(KN_GZDEV_CHUNK_BYTES, dflt);
        if (chunk_ < 1024) chunk_ = 1024;
        cps_ = (uint32_t)g_knobs.u64(KN_GZDEV_SLAB_CHUNKS, std::max<uint64_t>(256, ((uint64_t)128 << 20) / chunk_));
        if (cps_ < 1) cps_ = 1;
        // slabs in flight are counted in slabs of 512 chunks (the 256 KiB chunks of a large file): what fills the chip is chunks, and a file
        // of a gigabyte, with its smaller chunks and more of them to a slab, would hold twice the symbol room for nothing
        if (!g_knobs.is_set(KN_GZDEV_SLABS_IN_FLIGHT) && cps_ > 512) NSLAB = std::max<uint32_t>(2, (uint32_t)(((uint64_t)NSLAB * 512 + cps_ - 1) / cps_));
        // symbols of room per compressed byte: a first guess (FASTQ compresses three- to fivefold: 4.5, and 64 Ki symbols for the block behind the
        // range), then what the file has shown plus a quarter; a slab that overflows is decoded again with four times the room
        expand_ = g_knobs.is_set(KN_GZDEV_EXPAND) ? (double)g_knobs.u64(KN_GZDEV_EXPAND, 4) : 4.5;
        expand_fixed_ = g_knobs.is_set(KN_GZDEV_EXPAND);
        // The device memory of the path follows the INPUT.  What a chunk in flight holds: its symbol room (16-bit symbols, 4.5 : 1 and 64 Ki of
        // slack at first, then what the file has shown), 256 KiB of code lists, its bytes in the ring (twice: the ring is a power of two), and its share of
        // the text buffers (a slab's text each, 4.5 bytes per compressed byte, text_bufs of them over the slabs in flight).  The chunks in flight
        // are what the budget pays for -- in slabs small enough that four of them are in flight, so that upload, decode, link and the
        // consumers still overlap.  (Round 4 held 28 GB for a 0.6 GB pair: twelve slabs of a 5 GB file's size whatever the file.)
        if (budget && !g_knobs.is_set(KN_GZDEV_SLABS_IN_FLIGHT) && !g_knobs.is_set(KN_GZDEV_SLAB_CHUNKS)) {
            const uint64_t per_chunk = (uint64_t)sym_cap_first() * 2 + ((uint64_t)256 << 10) + (uint64_t)chunk_ * 2 + (uint64_t)chunk_ * 45 / 10 * std::max<uint32_t>(text_bufs, 1) / 4;
            const uint64_t fit = std::max<uint64_t>(64, budget / per_chunk);                     // chunks in flight the budget allows
            if ((uint64_t)NSLAB * cps_ > fit) {
                cps_ = (uint32_t)std::max<uint64_t>(16, std::min<uint64_t>(cps_, fit / 4));
                NSLAB = (uint32_t)std::max<uint64_t>(2, fit / cps_);
            }
        }
        text_piece_max_ = g_knobs.u64(KN_GZDEV_TEXT_PIECE, (uint64_t)1 << 30);
        size_t pos = 0;
        if (!member_header(pos, err)) return MF_E_FORMAT;
        base_byte_ = pos;
        n_chunks_ = (uint32_t)((size_ - base_byte_ + chunk_ - 1) / chunk_);
        if (n_chunks_ == 0) n_chunks_ = 1;
        if (cps_ > n_chunks_) cps_ = n_chunks_;
        // the ring: room for the slabs in flight, the bytes a slab's last chunk reads behind its range, and the uploader's pieces
        margin_ = (size_t)g_knobs.u64(KN_GZDEV_MARGIN, (size_t)8 << 20);
        const size_t slab_bytes = (size_t)cps_ * chunk_;
        size_t ring = pow2_ceil(std::max<size_t>(size_ + 512, 4096));
        {
            uint64_t want = g_knobs.u64(KN_GZDEV_RING_BYTES, 0);
            if (!want) want = (uint64_t)NSLAB * nl * slab_bytes + margin_ + 3 * std::min<uint64_t>((uint64_t)32 << 20, std::max<uint64_t>(slab_bytes, (uint64_t)4 << 20));      // (the slabs in flight, the read-ahead, three pieces of the uploader)
            want = pow2_ceil(std::max<uint64_t>(want, 4096));
            if (want < ring) ring = (size_t)want;
        }
        piece_ = std::min<size_t>(pow2_ceil((size_t)g_knobs.u64(KN_GZDEV_UPLOAD_PIECE_MB, 32)) << 20, std::max<size_t>(ring / 8, 512));
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
    int next(TextPiece &out, std::string &err);
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
        L.post_b = (g_knobs.is_set(KN_GZDEV_RESOLVE_STREAM) ? g_knobs.starts_1(KN_GZDEV_RESOLVE_STREAM) : (L.want_masked_post || n_chunks_ > 4 * cps_)) ? L.ds->take_post_b(L.post_slot) : nullptr;
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

} // namespace
} // namespace mf
