// Device ingest path, part 5 of 8: GzStream::next -- the producer's step: wait for the front slab's descriptors, walk them, link, hand the text over.
#pragma once
#include "mf_di_gzstream.h"

namespace mf {
namespace {

// The next piece of text (possibly nothing: out.buf is null).  Nothing here waits for the link, resolve or CRC kernels of a piece: the
// piece is handed over with an event (TextBuf::ready) that the consumer's stream waits for; what the producer does wait for is the
// decode kernel of the front slab (it needs the chunks' descriptors) and a text buffer.
inline int GzStream::next(TextPiece &out, std::string &err)
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
            const size_t budget = (size_t)g_knobs.u64(KN_GZDEV_RETRY_BYTES, (size_t)4 << 30);
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

} // namespace
} // namespace mf
