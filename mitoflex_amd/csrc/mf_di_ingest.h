// Device ingest path, part 7 of 8: the call -- producers, the consumers of the bait filter (index, filter, emit), the run.  The quality filter's
// consumers (q_*) are declared here and defined in mf_di_qual.h.
#pragma once
#include "mf_di_batch.h"

namespace mf {
namespace {


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
        const uint64_t slab = std::max<uint64_t>(g_knobs.u64(KN_INGEST_SLAB_BYTES, (uint64_t)256 << 20), 64);
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

    int q_scan(Worker &W, int mi, Batch &B, std::string &err);

    int q_dedup_room(DevScratch &S, uint64_t n, std::string &err);      // the set holds at most half its slots after n more keys

    int q_decide(Worker &W, Batch &B, uint64_t r0, uint64_t n, QPart &part, bool *stopped, std::string &err);

    int q_keep2(Worker &W, Batch &B, uint64_t n, QPart &part, std::string &err);

    int q_emit(Worker &W, int mi, std::shared_ptr<Batch> &B, const QPart &part, std::string &err);
    std::string out_path_[2];
    const char *out_name(int mi) const { return out_path_[mi].empty() ? "<stdout>" : out_path_[mi].c_str(); }

    bool q_progress(Worker &W, std::string &err, int &rc);

    void q_abandon() { qual->chunks.abort(); for (auto &sk : qual->sink) sk.abort(); }
    bool q_all_done()          // (takes mu)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (failed) return true;
        for (int i = 0; i < nm; i++) { Mate &M = m[i]; if (!M.eof || M.a_turn != M.taken || !M.batches.empty()) return false; }
        return qual->in_flight == 0;
    }

    void consume_q(Worker &W);

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
        const int nw = (int)std::max<uint64_t>(1, std::min<uint64_t>(16, g_knobs.u64(KN_INGEST_CONSUMERS, std::min<uint64_t>(qual ? 6 : 3, by_size))));
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
} // namespace mf
