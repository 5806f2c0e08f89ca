// Device ingest path, part 8 of 8: the quality filter's consumers (the reference's filter_v2 on this path): scan, de-duplication, decisions in file order, emit.
#pragma once
#include "mf_di_ingest.h"

namespace mf {
namespace {

inline int Ingest::q_scan(Worker &W, int mi, Batch &B, std::string &err)
{
    const double t0 = now_s();
    QualState &Q = *qual;
    DevScratch *Sp = scratch_for(W, B.ldev, err);
    if (!Sp) return MF_E_HIP;
    DevScratch &S = *Sp;
    const int dev = S.dev;
    hipStream_t sp = S.ctx->stream;
    volatile uint64_t *hs = S.h_small;
    const uint64_t n = B.n_rec;
    if (n >= 0xFFFFFFF0ull) { err = "a piece of text with 2^32 records"; return MF_E_ARG; }
    DCHK(B.q_bad.need(dev, n, false)); DCHK(B.q_sl.need(dev, n, false)); DCHK(B.q_ql.need(dev, n, false)); DCHK(B.q_olen.need(dev, n, false)); DCHK(B.q_fl.need(dev, n, false));
    DCHK(S.minmax.need(dev, 2));
    S.h_small[8] = ~0ull;
    DCHK(launch_bytes_from_host(S.minmax.p, S.h_small + 8, 8, sp));
    DCHK(launch_qual_scan(B.text, B.line_start.p, n, Q.P.start, Q.cap, Q.P.quality, Q.P.ns, B.q_bad.p, B.q_fl.p, B.q_sl.p, B.q_ql.p, B.q_olen.p, S.minmax.p, sp));
    if (mi == 0 && Q.P.dedup && !Q.P.trunc) { DCHK(B.q_hash.need(dev, n, false)); DCHK(launch_qual_hash(B.text, B.line_start.p, n, Q.P.start, B.q_sl.p, B.q_hash.p, sp)); }
    DCHK(launch_bytes_to_host(S.h_small + 4, S.minmax.p, 4, sp));
    uint32_t *h_bad = nullptr; uint8_t *h_fl = nullptr;
    if (mi == 1) {
        DCHK(S.stage(n * 5 + 16));
        h_bad = (uint32_t *)S.h_stage; h_fl = S.h_stage + n * 4;
        DCHK(hipMemcpyAsync(h_bad, B.q_bad.p, n * 4, hipMemcpyDeviceToHost, sp));
        DCHK(hipMemcpyAsync(h_fl, B.q_fl.p, n, hipMemcpyDeviceToHost, sp));
    }
    DCHK(hipStreamSynchronize(sp));
    const uint32_t first_flag = (uint32_t)hs[4];
    if (mi == 1 && (!Q.bad2.put(B.rec_base, n, h_bad) || !Q.fl2.put(B.rec_base, n, h_fl))) { err = "out of memory"; return MF_E_NOMEM; }
    uint64_t panic_at = ~0ull;
    if (first_flag != ~0u) {
        // rare: a byte that is not ASCII in a line the reference unwraps, or a string shorter than the cut's start.  The flagged
        // records are looked at on the host, in order, until one makes the reference panic (a header in UTF-8 does not).
        std::vector<uint8_t> fl(n), text(B.n_text + 1); std::vector<uint64_t> ls(4 * n + 1);
        // (on the consumer's own stream: a copy on the null stream would wait for every decode kernel in flight on the blocking CU-masked streams)
        DCHK(hipMemcpyAsync(fl.data(), B.q_fl.p, n, hipMemcpyDeviceToHost, sp));
        DCHK(hipMemcpyAsync(ls.data(), B.line_start.p, (4 * n + 1) * 8, hipMemcpyDeviceToHost, sp));
        DCHK(hipMemcpyAsync(text.data(), B.text, B.n_text, hipMemcpyDeviceToHost, sp));
        DCHK(hipStreamSynchronize(sp));
        auto line = [&](uint64_t k, const char *&p, size_t &len) {
            const uint64_t a = ls[k], b = std::min<uint64_t>(ls[k + 1], B.n_text + 1);
            len = (size_t)(b - a - 1); p = (const char *)text.data() + a;
            if (len && p[len - 1] == '\r') len--;
        };
        for (uint64_t r = first_flag; r < n && panic_at == ~0ull; r++) {
            if (!(fl[r] & (QF_HIGH | QF_SHORT | QF_LONG))) continue;
            if (fl[r] & QF_LONG) { err = "a FASTQ record of 4 GiB or more"; return MF_E_ARG; }
            if (fl[r] & QF_SHORT) { panic_at = r; break; }
            for (int k : {0, 1, 3}) { const char *p; size_t len; line(4 * r + k, p, len); if (!utf8_valid(p, len)) { panic_at = r; break; } }
        }
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        Mate &M = m[mi];
        if (panic_at != ~0ull) Q.panic_rec[mi] = std::min(Q.panic_rec[mi], B.rec_base + panic_at);
        B.filtered = true;
        update_scanned(M);
        Q.t_scan += now_s() - t0;
        size_t f = 0, t = 0;
        if (hipMemGetInfo(&f, &t) == hipSuccess) mem_used_max = std::max(mem_used_max, t - f);
    }
    return MF_OK;
}

inline int Ingest::q_dedup_room(DevScratch &S, uint64_t n, std::string &err)      // the set holds at most half its slots after n more keys
{
    QualState &Q = *qual;
    const int dev = S.dev; hipStream_t sp = S.ctx->stream;
    if (!Q.dd_slots) {
        // (as many slots as four times the records the input is likely to hold, at most 2^24 to begin with: the set doubles as it fills)
        uint64_t text_est = 0;
        for (int i = 0; i < nm; i++) text_est += m[i].gz ? m[i].map.n * 4 : m[i].map.n;
        uint64_t lg_est = 16; while (lg_est < 24 && ((uint64_t)1 << lg_est) < text_est / (uint64_t)nm / 300 * 4) lg_est++;
        uint64_t lg = g_knobs.u64(KN_DEDUP_LOG2_SLOTS, lg_est);
        lg = std::min<uint64_t>(std::max<uint64_t>(lg, 4), 34);
        Q.dd_slots = (uint64_t)1 << lg;
        DCHK(Q.dd_keys.need(dev, Q.dd_slots, false)); DCHK(Q.dd_first.need(dev, Q.dd_slots, false));
        DCHK(hipMemsetAsync(Q.dd_keys.p, 0, Q.dd_slots * 8, sp)); DCHK(hipMemsetAsync(Q.dd_first.p, 0xFF, Q.dd_slots * 8, sp));
    }
    while (2 * (Q.dd_n + n) > Q.dd_slots) {
        DevBuf<unsigned long long> k2, f2;
        DCHK(k2.need(dev, Q.dd_slots * 2, false)); DCHK(f2.need(dev, Q.dd_slots * 2, false));
        DCHK(hipMemsetAsync(k2.p, 0, Q.dd_slots * 16, sp)); DCHK(hipMemsetAsync(f2.p, 0xFF, Q.dd_slots * 16, sp));
        DCHK(launch_dedup_rehash(Q.dd_keys.p, Q.dd_first.p, Q.dd_slots, k2.p, f2.p, Q.dd_slots * 2, sp));
        DCHK(hipStreamSynchronize(sp));
        std::swap(Q.dd_keys.p, k2.p); std::swap(Q.dd_keys.cap, k2.cap); std::swap(Q.dd_keys.bytes_, k2.bytes_);
        std::swap(Q.dd_first.p, f2.p); std::swap(Q.dd_first.cap, f2.cap); std::swap(Q.dd_first.bytes_, f2.bytes_);
        Q.dd_slots *= 2;
    }
    return MF_OK;
}

// records [r0, r0 + n) of mate 1's piece B (emit_mu held): the tests, the de-duplication, the budget; where their output goes.
// *stopped: the -t budget ran out among them (part.n: the records in front of the one that overflowed it)
inline int Ingest::q_decide(Worker &W, Batch &B, uint64_t r0, uint64_t n, QPart &part, bool *stopped, std::string &err)
{
    const uint64_t g0 = B.rec_base + r0;          // file index of the first
    const double t0 = now_s();
    QualState &Q = *qual;
    DevScratch *Sp = scratch_for(W, B.ldev, err);
    if (!Sp) return MF_E_HIP;
    DevScratch &S = *Sp;
    const int dev = S.dev;
    hipStream_t sp = S.ctx->stream;
    volatile uint64_t *hs = S.h_small;
    uint64_t bytes = 0, kept_here = 0;
    if (n) {
        DCHK(S.q_alive.need(dev, n)); DCHK(S.q_keep.need(dev, n)); DCHK(S.out_len.need(dev, n)); DCHK(S.out_off.need(dev, n + 1)); DCHK(S.scan_tmp.need(dev, n / 4096 + 4));
        DCHK(S.stage(n * 5 + 16));
        if (Q.pe) {
            DCHK(S.q_bad2.need(dev, n)); DCHK(S.q_fl2.need(dev, n));
            Q.bad2.get(g0, n, (uint32_t *)S.h_stage); Q.fl2.get(g0, n, S.h_stage + n * 4);
            DCHK(launch_bytes_from_host(S.q_bad2.p, S.h_stage, n * 4, sp));
            DCHK(launch_bytes_from_host(S.q_fl2.p, S.h_stage + n * 4, n, sp));
        }
        DCHK(launch_qual_decide(n, Q.pe, Q.P.trunc, Q.P.limit, B.q_bad.p + r0, B.q_fl.p + r0, B.q_sl.p + r0, B.q_ql.p + r0, S.q_bad2.p, S.q_fl2.p, S.q_alive.p, sp));
        const bool dd = Q.P.dedup && !Q.P.trunc;
        if (!Q.dd_small.p) {          // [0] file index of the hash value 0, [1] keys in the set, [2] kept records of a piece
            DCHK(Q.dd_small.need(dev, 4, false));
            DCHK(hipMemsetAsync(Q.dd_small.p, 0xFF, 8, sp)); DCHK(hipMemsetAsync(Q.dd_small.p + 1, 0, 24, sp));
        }
        if (dd) { const int rc = q_dedup_room(S, n, err); if (rc) return rc; }
        if (dd) {
            DCHK(S.q_dup.need(dev, n));
            DCHK(launch_dedup(B.q_hash.p + r0, S.q_alive.p, (uint32_t)n, g0, Q.dd_keys.p, Q.dd_first.p, Q.dd_slots, Q.dd_small.p, Q.dd_small.p + 1, S.q_dup.p, sp));
        }
        DCHK(hipMemsetAsync(Q.dd_small.p + 2, 0, 8, sp));
        DCHK(launch_qual_keep(n, S.q_alive.p, dd ? S.q_dup.p : nullptr, B.q_olen.p + r0, S.q_keep.p, S.out_len.p, Q.dd_small.p + 2, sp));
        uint8_t *h_keep = S.h_stage;
        if (Q.P.trim) {
            // the budget is sequential (main.rs:254-259, 311-316): the first kept record that overflows it ends the run
            uint32_t *h_sl = (uint32_t *)(S.h_stage + ((n + 15) & ~(uint64_t)15));          // (n * 5 + 16 bytes are there)
            DCHK(hipMemcpyAsync(h_keep, S.q_keep.p, n, hipMemcpyDeviceToHost, sp));
            DCHK(hipMemcpyAsync(h_sl, B.q_sl.p + r0, n * 4, hipMemcpyDeviceToHost, sp));
            DCHK(hipStreamSynchronize(sp));
            uint64_t i = 0;
            for (; i < n; i++) {
                if (!h_keep[i]) continue;
                Q.budget += h_sl[i];
                if (Q.budget > Q.P.trim) { *stopped = true; break; }
                kept_here++;
            }
            n = i;
        }
        if (n) {
            DCHK(launch_scan_u32(S.out_len.p, n, S.out_off.p, S.scan_tmp.p, sp));
            DCHK(launch_bytes_to_host(S.h_small + 6, S.out_off.p + n, 8, sp));
            DCHK(launch_bytes_to_host(S.h_small + 3, Q.dd_small.p + 1, 16, sp));       // keys of the set, kept of the piece
            if (Q.pe && !Q.P.trim) DCHK(hipMemcpyAsync(h_keep, S.q_keep.p, n, hipMemcpyDeviceToHost, sp));
            DCHK(hipStreamSynchronize(sp));
            bytes = hs[6]; if (dd) Q.dd_n = hs[3];
            if (!Q.P.trim) kept_here = hs[4];
            if (Q.pe && !Q.keep.put(g0, n, h_keep)) { err = "out of memory"; return MF_E_NOMEM; }
        }
    }
    part.r0 = r0; part.n = n; part.bytes = bytes; part.out_at = Q.out_pos[0]; Q.out_pos[0] += bytes;
    Q.kept += kept_here;
    { std::lock_guard<std::mutex> lk(mu); Q.t_decide += now_s() - t0; }
    return MF_OK;
}

// mate 2's piece B, its first n records (emit_mu held): the keep flags mate 1's decisions left for them
inline int Ingest::q_keep2(Worker &W, Batch &B, uint64_t n, QPart &part, std::string &err)
{
    const double t0 = now_s();
    QualState &Q = *qual;
    DevScratch *Sp = scratch_for(W, B.ldev, err);
    if (!Sp) return MF_E_HIP;
    DevScratch &S = *Sp;
    const int dev = S.dev;
    hipStream_t sp = S.ctx->stream;
    uint64_t bytes = 0;
    if (n) {
        DCHK(S.q_keep.need(dev, n)); DCHK(S.out_len.need(dev, n)); DCHK(S.out_off.need(dev, n + 1)); DCHK(S.scan_tmp.need(dev, n / 4096 + 4));
        DCHK(S.stage(n + 16));
        Q.keep.get(B.rec_base, n, S.h_stage);
        DCHK(launch_bytes_from_host(S.q_keep.p, S.h_stage, n, sp));
        DCHK(launch_qual_keep(n, S.q_keep.p, nullptr, B.q_olen.p, nullptr, S.out_len.p, nullptr, sp));
        DCHK(launch_scan_u32(S.out_len.p, n, S.out_off.p, S.scan_tmp.p, sp));
        DCHK(launch_bytes_to_host(S.h_small + 6, S.out_off.p + n, 8, sp));
        DCHK(hipStreamSynchronize(sp));
        bytes = ((volatile uint64_t *)S.h_small)[6];
    }
    part.r0 = 0; part.n = n; part.bytes = bytes; part.out_at = Q.out_pos[1]; Q.out_pos[1] += bytes;
    { std::lock_guard<std::mutex> lk(mu); Q.t_decide += now_s() - t0; }
    return MF_OK;
}

// the kept records of a decided part -> its place in the output file (any number of parts at a time; S.out_len / S.out_off are
// still those of the part: the consumer that decided it is the one that gathers it, and does nothing in between).  The text is
// done with once the records are gathered: the last part's consumer lets go of the piece (B) there, and its buffer goes back
// to the producer while the output is on its way down and out.
inline int Ingest::q_emit(Worker &W, int mi, std::shared_ptr<Batch> &B, const QPart &part, std::string &err)
{
    QualState &Q = *qual;
    if (!part.bytes) { B.reset(); return MF_OK; }
    const double t0 = now_s();
    DevScratch *Sp = scratch_for(W, B->ldev, err);
    if (!Sp) return MF_E_HIP;
    DevScratch &S = *Sp;
    const int dev = S.dev;
    hipStream_t sp = S.ctx->stream;
    const uint64_t bytes = part.bytes;
    DCHK(S.d_out.need(dev, bytes));
    DCHK(launch_qual_gather(B->text, B->line_start.p + 4 * part.r0, part.n, Q.P.start, B->q_sl.p + part.r0, B->q_ql.p + part.r0, S.out_len.p, S.out_off.p, S.d_out.p, sp));
    DCHK(hipStreamSynchronize(sp));
    B.reset();
    const size_t chunk = Q.chunks.chunk();
    double tw = 0;
    { const double w0 = now_s(); if (!Q.sink[mi].wait_turn(part.out_at)) { err = "abandoned"; return MF_E_IO; } tw += now_s() - w0; }      // (standard output, a pipe, a .gz: the parts' chunks are taken in file order -- tests/native/qualsink_check.cpp hangs without it)      // (standard output, a pipe, a .gz: the parts' chunks are taken in file order)
    for (uint64_t off = 0; off < bytes; off += chunk) {
        const uint64_t len = std::min<uint64_t>(chunk, bytes - off);
        const double w0 = now_s();
        bool no_mem = false;
        uint8_t *p = Q.chunks.take(&no_mem);
        tw += now_s() - w0;
        if (!p) { if (no_mem) { err = "hipHostMalloc failed: no pinned memory for the output's chunks"; return MF_E_NOMEM; } err = "abandoned"; return MF_E_IO; }
        hipError_t c = hipMemcpyAsync(p, S.d_out.p + off, len, hipMemcpyDeviceToHost, sp);
        if (c == hipSuccess) c = hipStreamSynchronize(sp);
        if (c != hipSuccess) { Q.chunks.give(p); err = std::string("copy of the output failed: ") + hipGetErrorString(c); return MF_E_HIP; }
        wrote_any = true;                   // (before the first byte reaches the sink: a later failure must not hand the call to the host pipeline, which would write them again)
        Q.sink[mi].push(part.out_at + off, p, (size_t)len);
        if (!Q.sink[mi].ok()) { err = std::string("write error on ") + out_name(mi); return MF_E_IO; }
    }
    { std::lock_guard<std::mutex> lk(mu); Q.t_gather += now_s() - t0 - tw; Q.t_chunk += tw; }
    return MF_OK;
}

// decide and write what can be decided and written.  true: did something
inline bool Ingest::q_progress(Worker &W, std::string &err, int &rc)
{
    QualState &Q = *qual;
    bool did = false;
    for (;;) {
        std::shared_ptr<Batch> B; int mi = -1; uint64_t r0 = 0, n = 0; bool final = false, by_panic = false, stopped = false, whole = false;
        QPart part;
        {
            std::unique_lock<std::mutex> elk(emit_mu, std::try_to_lock);          // (somebody else is deciding: there is other work)
            if (!elk.owns_lock()) break;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (failed) break;
                Mate &A = m[0];
                while (Q.decided_final && !A.batches.empty() && A.batches.front()->filtered) { A.batches.pop_front(); did = true; }      // (nothing of them is wanted)
                if (!Q.decided_final && !A.batches.empty() && A.batches.front()->filtered) {
                    Batch &F = *A.batches.front();
                    const uint64_t end = F.rec_base + F.n_rec, cur = F.rec_base + F.q_done;
                    uint64_t limit = std::min(end, Q.panic_rec[0]), upto = limit;          // limit: what of the piece will ever be decided
                    bool ready = true;
                    if (Q.pe) {
                        const bool other_done = scans_done(m[1]);
                        limit = std::min(limit, Q.panic_rec[1]);
                        if (other_done) limit = std::min(limit, m[1].rec_indexed);          // (pairs end with the shorter file)
                        upto = other_done ? limit : std::min(limit, m[1].rec_filtered);     // ... and what can be now: the records mate 2's scanned pieces cover
                        // A part of the piece is decided only when waiting for the rest cannot end: the other mate holds all its text buffers
                        // (its pieces wait for THESE decisions before they are written and their buffers come back)
#ifdef MF_TEST_WITHOUT_PARTIAL_DECISIONS         // (tests/test_ingest_orchestration.py: the check must hang without this rule, as the path did before it had it)
                        ready = upto == limit;
#else
                        ready = upto == limit || (upto > cur && m[1].slots.none_free());
#endif
                    }
                    if (ready) {
                        whole = upto == limit;
                        final = whole && limit < end;
                        by_panic = final && std::min(Q.panic_rec[0], Q.panic_rec[1]) == limit;
                        r0 = F.q_done; n = upto > cur ? upto - cur : 0;
                        B = A.batches.front(); mi = 0;
                        if (whole) A.batches.pop_front();
                    }
                }
                if (mi < 0 && !Q.decided_final && A.eof && A.a_turn == A.taken && A.batches.empty()) { Q.decided_final = true; did = true; }      // mate 1 has been decided to its end
                if (mi < 0 && Q.pe && !m[1].batches.empty() && m[1].batches.front()->filtered) {
                    Batch &F = *m[1].batches.front();
                    const uint64_t end = F.rec_base + F.n_rec;
                    if (Q.decided >= end || Q.decided_final) {
                        const uint64_t upto = std::min(end, Q.decided);
                        n = upto > F.rec_base ? upto - F.rec_base : 0;
                        B = m[1].batches.front(); m[1].batches.pop_front(); mi = 1;
                    }
                }
                if (mi >= 0) Q.in_flight++;
            }
            if (mi < 0) break;
            rc = mi == 0 ? q_decide(W, *B, r0, n, part, &stopped, err) : q_keep2(W, *B, n, part, err);
            if (!rc && mi == 0) {
                std::lock_guard<std::mutex> lk(mu);
                B->q_done = r0 + part.n;
                Q.decided = B->rec_base + B->q_done;
                if (stopped && !whole) m[0].batches.pop_front();                            // (B is the front: nobody else decides)
                if (final || stopped) { Q.decided_final = true; if (by_panic && !stopped) Q.panicked = true; }
            }
        }
        cv.notify_all(); cv_all.notify_all();
        if (!rc) rc = q_emit(W, mi, B, part, err);     // (lets go of B as soon as the records are gathered: the text buffer goes back, a producer may be waiting for one)
        B.reset();
        { std::lock_guard<std::mutex> lk(mu); Q.in_flight--; }
        cv.notify_all(); cv_all.notify_all();
        did = true;
        if (rc) return true;
    }
    return did;
}

inline void Ingest::consume_q(Worker &W)
{
    std::string err;
    for (;;) {
        int rc = MF_OK;
        bool did = q_progress(W, err, rc);
        if (rc) { fail_with(rc, err); q_abandon(); return; }
        int mi = 0; TextPiece P; uint64_t seq = 0; bool again = false;
        if (take_piece(mi, P, seq, err, rc, &again)) {
            Mate &M = m[mi];
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return failed || M.a_turn == seq; });
                if (failed) return;
            }
            std::shared_ptr<Batch> B;
            rc = index_piece(W, M, P, B, err);
            Batch *Bp = nullptr;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!rc) {
                    B->rec_base = M.rec_indexed; M.rec_indexed += B->n_rec;
                    if (B->n_rec) { Bp = B.get(); M.batches.push_back(std::move(B)); }
                }
                M.a_turn = seq + 1;
            }
            cv.notify_all();
            if (rc) { fail_with(rc, err); q_abandon(); return; }
            B.reset();
            if (Bp) { rc = q_scan(W, mi, *Bp, err); if (rc) { fail_with(rc, err); q_abandon(); return; } }
            continue;
        }
        if (rc) { fail_with(rc, err); q_abandon(); return; }
        if (q_all_done()) return;
        if (!did && !again) { std::unique_lock<std::mutex> lk(mu_all); nap(cv_all, lk, 200); }
    }
}

} // namespace
} // namespace mf
