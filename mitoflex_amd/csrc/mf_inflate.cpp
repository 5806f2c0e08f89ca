// Streaming gzip decoder (see mf_inflate.h).  DEFLATE is RFC 1951, the gzip wrapper RFC 1952.
#include "mf_inflate.h"
#include "mf_inflate_core.h"

#include <string.h>
#include <zlib.h>      // crc32 only

namespace mf {

using namespace inflate_core;

GzInflater::~GzInflater()
{
    if (crc_thread_.joinable()) {
        { std::lock_guard<std::mutex> lk(crc_mu_); crc_stop_ = true; }
        crc_cv_.notify_all();
        crc_thread_.join();
    }
}

void GzInflater::crc_loop()
{
    std::unique_lock<std::mutex> lk(crc_mu_);
    for (;;) {
        crc_cv_.wait(lk, [&] { return crc_stop_ || !crc_jobs_.empty(); });
        if (crc_jobs_.empty()) return;                      // stop requested and nothing left
        const std::pair<const uint8_t *, size_t> job = crc_jobs_.front(); crc_jobs_.pop_front();
        crc_busy_ = true;
        lk.unlock();
        uint32_t c = crc_; size_t off = 0;                   // crc_ is only touched by this thread between crc_wait() calls
        while (off < job.second) { const size_t k = job.second - off < ((size_t)1 << 30) ? job.second - off : ((size_t)1 << 30); c = (uint32_t)crc32(c, job.first + off, (uInt)k); off += k; }
        lk.lock();
        crc_ = c; crc_busy_ = false;
        if (crc_jobs_.empty()) crc_idle_cv_.notify_all();
    }
}

void GzInflater::crc_push(const uint8_t *p, size_t n)
{
    if (!n) return;
    if (n < 65536 && !crc_thread_.joinable()) { crc_ = (uint32_t)crc32(crc_, p, (uInt)n); return; }   // small inputs never start the thread
    if (!crc_thread_.joinable()) crc_thread_ = std::thread([this] { crc_loop(); });
    { std::lock_guard<std::mutex> lk(crc_mu_); crc_jobs_.emplace_back(p, n); }
    crc_cv_.notify_one();
}

void GzInflater::crc_wait()
{
    if (!crc_thread_.joinable()) return;
    std::unique_lock<std::mutex> lk(crc_mu_);
    crc_idle_cv_.wait(lk, [&] { return crc_jobs_.empty() && !crc_busy_; });
}

void GzInflater::open(const uint8_t *data, size_t size)
{
    crc_wait();
    in_begin_ = in_ = data; in_end_ = data + size;
    bitbuf_ = 0; bitcnt_ = 0; overrun_ = 0;
    state_ = size ? MEMBER_HEADER : DONE;
    last_block_ = false; stored_left_ = 0; pend_len_ = pend_dist_ = 0;
    hist_.assign(32768, 0); hist_len_ = 0;
    crc_ = 0; member_out_ = 0; any_member_ = false; transparent_ = false;
}

inline void GzInflater::refill()
{
    if (in_end_ - in_ >= 8) {
        uint64_t v; memcpy(&v, in_, 8);
        bitbuf_ |= v << bitcnt_;
        in_ += (63 - bitcnt_) >> 3;
        bitcnt_ |= 56;
    } else {
        while (bitcnt_ <= 56) {
            uint64_t b = 0;
            if (in_ < in_end_) b = *in_++; else overrun_++;
            bitbuf_ |= b << bitcnt_;
            bitcnt_ += 8;
        }
    }
}

bool GzInflater::need_bits(unsigned n)
{
    if (bitcnt_ < n) refill();
    // bits that came from beyond the end of the file must not be used
    return overrun_ * 8 + n <= bitcnt_ || overrun_ == 0;
}

bool GzInflater::parse_member_header(std::string &err)
{
    // byte aligned here, bit buffer empty
    const size_t left = (size_t)(in_end_ - in_);
    if (left < 2 || in_[0] != 0x1f || in_[1] != 0x8b) {
        if (any_member_) { state_ = DONE; return true; }      // trailing bytes that are not a member: ignored, like gzread
        err = "not in gzip format"; return false;
    }
    if (left < 10) { err = "truncated gzip header"; return false; }
    if (in_[2] != 8) { err = "unknown gzip compression method"; return false; }
    const unsigned flg = in_[3];
    const uint8_t *p = in_ + 10;
    if (flg & 4) {
        if (in_end_ - p < 2) { err = "truncated gzip header"; return false; }
        const size_t xlen = p[0] | ((size_t)p[1] << 8);
        p += 2;
        if ((size_t)(in_end_ - p) < xlen) { err = "truncated gzip header"; return false; }
        p += xlen;
    }
    for (unsigned bit = 8; bit <= 16; bit <<= 1)
        if (flg & bit) {
            const uint8_t *z = (const uint8_t *)memchr(p, 0, (size_t)(in_end_ - p));
            if (!z) { err = "truncated gzip header"; return false; }
            p = z + 1;
        }
    if (flg & 2) { if (in_end_ - p < 2) { err = "truncated gzip header"; return false; } p += 2; }
    in_ = p;
    crc_ = 0; member_out_ = 0; any_member_ = true; last_block_ = false;
    state_ = BLOCK_HEADER;
    return true;
}

bool GzInflater::build_tables(const uint8_t *lens, unsigned n_litlen, unsigned n_dist, std::string &err)
{
    if (!build_table(lens, n_litlen, LIT_BITS, lit_, lit_entry) || !build_table(lens + n_litlen, n_dist, DIST_BITS, dist_, dist_entry)) {
        err = "invalid Huffman code in deflate stream"; return false;
    }
    pair_literals(lit_);
    return true;
}

bool GzInflater::parse_block_header(std::string &err)
{
    if (!need_bits(3)) { err = "truncated deflate stream"; return false; }
    last_block_ = peek(1); drop(1);
    const unsigned type = peek(2); drop(2);
    if (type == 0) {
        drop(bitcnt_ & 7);                                  // to the byte boundary
        if (!need_bits(32)) { err = "truncated deflate stream"; return false; }
        const unsigned len = peek(16); drop(16);
        const unsigned nlen = peek(16); drop(16);
        if ((len ^ nlen) != 0xFFFFu) { err = "invalid stored block length"; return false; }
        // give the whole bytes still in the bit buffer back to the input
        const size_t spare = bitcnt_ / 8;
        if (overrun_ > spare) { err = "truncated deflate stream"; return false; }
        in_ -= spare - overrun_; overrun_ = 0; bitbuf_ = 0; bitcnt_ = 0;
        stored_left_ = len;
        state_ = STORED;
        return true;
    }
    if (type == 1) {
        uint8_t lens[320];
        fixed_lengths(lens);
        if (!build_tables(lens, 288, 32, err)) return false;
        state_ = HUFFMAN;
        return true;
    }
    if (type == 3) { err = "invalid deflate block type"; return false; }
    if (!need_bits(14)) { err = "truncated deflate stream"; return false; }
    const unsigned hlit = peek(5) + 257; drop(5);
    const unsigned hdist = peek(5) + 1; drop(5);
    const unsigned hclen = peek(4) + 4; drop(4);
    if (hlit > 286 || hdist > 30) { err = "too many length or distance symbols"; return false; }
    uint8_t pre[19] = {0};
    for (unsigned i = 0; i < hclen; i++) {
        if (!need_bits(3)) { err = "truncated deflate stream"; return false; }
        pre[PRECODE_ORDER[i]] = (uint8_t)peek(3); drop(3);
    }
    std::vector<uint32_t> ptab;
    if (!build_table(pre, 19, PRE_BITS, ptab, [](unsigned s) { return entry(0, LITERAL, 0, s); })) { err = "invalid code lengths set"; return false; }
    uint8_t lens[286 + 30 + 138];
    unsigned i = 0;
    while (i < hlit + hdist) {
        if (!need_bits(7 + 7)) { err = "truncated deflate stream"; return false; }
        const uint32_t e = ptab[peek(PRE_BITS)];
        if (e_kind(e) != LITERAL) { err = "invalid code lengths set"; return false; }
        drop(e_len(e));
        const unsigned s = e_value(e);
        if (s < 16) { lens[i++] = (uint8_t)s; continue; }
        unsigned rep, val = 0;
        if (s == 16) {
            if (i == 0) { err = "invalid bit length repeat"; return false; }
            val = lens[i - 1]; rep = 3 + peek(2); drop(2);
        } else if (s == 17) { rep = 3 + peek(3); drop(3); }
        else { rep = 11 + peek(7); drop(7); }
        if (i + rep > hlit + hdist) { err = "invalid bit length repeat"; return false; }
        while (rep--) lens[i++] = (uint8_t)val;
    }
    if (lens[256] == 0) { err = "invalid code -- missing end-of-block"; return false; }
    if (!build_tables(lens, hlit, hdist, err)) return false;
    state_ = HUFFMAN;
    return true;
}

bool GzInflater::check_trailer(std::string &err)
{
    drop(bitcnt_ & 7);
    if (!need_bits(32)) { err = "truncated gzip trailer"; return false; }
    const uint32_t want_crc = peek(32) ; drop(32);
    if (bitcnt_ < 32) refill();
    if (overrun_ * 8 + 32 > bitcnt_ && overrun_) { err = "truncated gzip trailer"; return false; }
    const uint32_t want_size = (uint32_t)(bitbuf_ & 0xFFFFFFFFu); drop(32);
    crc_wait();
    if (want_crc != crc_) { err = "incorrect data check"; return false; }
    if (want_size != (uint32_t)member_out_) { err = "incorrect length check"; return false; }
    const size_t spare = bitcnt_ / 8;                       // whole bytes read ahead belong to what follows the member
    if (overrun_ > spare) { err = "truncated gzip trailer"; return false; }
    in_ -= spare - overrun_; overrun_ = 0; bitbuf_ = 0; bitcnt_ = 0;
    state_ = in_ < in_end_ ? MEMBER_HEADER : DONE;
    return true;
}

long GzInflater::read(uint8_t *out, size_t cap, std::string &err)
{
    // decode in slices so that the CRC thread can follow one slice behind; nothing is left pending on return
    // (the caller may move or reuse its buffer)
    const size_t slice = (size_t)1 << 20;
    size_t done = 0;
    while (done < cap && state_ != DONE) {
        const size_t want = cap - done < slice ? cap - done : slice;
        const long n = read_slice(out + done, want, err);
        if (n < 0) { crc_wait(); return -1; }
        done += (size_t)n;
        if ((size_t)n < want) break;
    }
    crc_wait();
    return (long)done;
}

long GzInflater::read_slice(uint8_t *out, size_t cap, std::string &err)
{
    uint8_t *o = out, *const o_end = out + cap;
    uint8_t *seg = out;                                      // start of the bytes not yet folded into the member's CRC
    auto fold = [&] { if (o > seg) { crc_push(seg, (size_t)(o - seg)); member_out_ += (uint64_t)(o - seg); seg = o; } };
    // byte `back` positions behind o (back >= 1), reaching into the history of earlier calls when needed
    auto byte_at = [&](size_t back, bool &ok) -> uint8_t {
        const size_t have = (size_t)(o - out);
        if (back <= have) return o[-(ptrdiff_t)back];
        const size_t h = back - have;
        if (h > hist_len_) { ok = false; return 0; }
        return hist_[hist_len_ - h];
    };
    while (o < o_end && state_ != DONE) {
        switch (state_) {
        case MEMBER_HEADER:
            if (transparent_) break;
            if (!any_member_ && (in_end_ - in_ < 2 || in_[0] != 0x1f || in_[1] != 0x8b)) { transparent_ = true; break; }   // gzread copies non-gzip input through
            if (!parse_member_header(err)) return -1;
            break;
        case BLOCK_HEADER:
            if (!parse_block_header(err)) return -1;
            break;
        case STORED: {
            size_t n = stored_left_;
            if (n > (size_t)(o_end - o)) n = (size_t)(o_end - o);
            if (n > (size_t)(in_end_ - in_)) { err = "truncated deflate stream"; return -1; }
            memcpy(o, in_, n); o += n; in_ += n; stored_left_ -= n;
            if (stored_left_ == 0) state_ = last_block_ ? MEMBER_TRAILER : BLOCK_HEADER;
            break;
        }
        case HUFFMAN: {
            // a match cut short by the end of the previous buffer
            while (pend_len_ && o < o_end) {
                bool ok = true; const uint8_t b = byte_at(pend_dist_, ok);
                if (!ok) { err = "invalid distance too far back"; return -1; }
                *o++ = b; pend_len_--;
            }
            if (pend_len_) break;
            const uint32_t *lit = lit_.data(), *dst = dist_.data();
            bool block_done = false;
            // ---- fast loop: plenty of input and output left, no bounds checks inside
            constexpr uint32_t LMASK = (1u << LIT_BITS) - 1, DMASK = (1u << DIST_BITS) - 1;
            if (in_end_ - in_ >= 16 && o_end - o >= 258 + 16) {
                // the bit reader lives in locals here: byte stores through `o` may alias anything, members included,
                // and a reload after every literal would sit on the critical path
                uint64_t bb = bitbuf_; unsigned bc = bitcnt_; const uint8_t *ip = in_;
                const uint8_t *const ip_safe = in_end_ - 16; uint8_t *const o_safe = o_end - (258 + 16);
                const char *bad = nullptr;
#define MF_DROP(n) do { const unsigned n_ = (n); bb >>= n_; bc -= n_; } while (0)
#define MF_PEEK(n) ((uint32_t)(bb & ((1ULL << (n)) - 1)))
                while (ip <= ip_safe && o <= o_safe) {
                    {   // refill, the branch-free way (at least 8 input bytes are there)
                        uint64_t v; memcpy(&v, ip, 8);
                        bb |= v << bc;
                        ip += (63 - bc) >> 3;
                        bc |= 56;
                    }
                    uint32_t e = lit[bb & LMASK];
                    if (e & LITERAL_FLAG) {                              // up to four lookups, one or two literals each, on one refill (<= 44 bits)
#define MF_PUT_LITERALS() do { o[0] = (uint8_t)(e >> 16); o[1] = (uint8_t)(e >> 24); o += 1 + ((e >> 14) & 1u); MF_DROP(e & 31u); } while (0)
                        MF_PUT_LITERALS();
                        e = lit[bb & LMASK];
                        if (e & LITERAL_FLAG) {
                            MF_PUT_LITERALS();
                            e = lit[bb & LMASK];
                            if (e & LITERAL_FLAG) {
                                MF_PUT_LITERALS();
                                e = lit[bb & LMASK];
                                if (e & LITERAL_FLAG) MF_PUT_LITERALS();
                            }
                        }
#undef MF_PUT_LITERALS
                        continue;
                    }
                    if (e_kind(e) == LINK) {
                        MF_DROP(LIT_BITS); e = lit[e_value(e) + MF_PEEK(e_extra(e))];
                        MF_DROP(e_len(e));
                        if (e & LITERAL_FLAG) { *o++ = (uint8_t)(e >> 16); continue; }
                    } else MF_DROP(e_len(e));
                    if (e_kind(e) == END_OF_BLOCK) { block_done = true; break; }
                    if (e_kind(e) != LENGTH) { bad = "invalid literal/length code"; break; }
                    unsigned len = e_value(e) + MF_PEEK(e_extra(e)); MF_DROP(e_extra(e));
                    uint32_t d = dst[bb & DMASK];
                    if (e_kind(d) == LINK) { MF_DROP(DIST_BITS); d = dst[e_value(d) + MF_PEEK(e_extra(d))]; }
                    MF_DROP(e_len(d));
                    if (e_kind(d) != DISTANCE) { bad = "invalid distance code"; break; }
                    const unsigned dist = e_value(d) + MF_PEEK(e_extra(d)); MF_DROP(e_extra(d));
                    if (dist <= (size_t)(o - out)) {
                        const uint8_t *s = o - dist; uint8_t *t = o; o += len;
                        if (dist >= 8) { do { memcpy(t, s, 8); t += 8; s += 8; } while (t < o); }
                        else if (dist == 1) memset(t, *s, len);
                        else { do { *t++ = *s++; } while (t < o); }
                    } else {
                        while (len--) { bool ok = true; const uint8_t b = byte_at(dist, ok); if (!ok) { bad = "invalid distance too far back"; break; } *o++ = b; }
                        if (bad) break;
                    }
                }
#undef MF_DROP
#undef MF_PEEK
                bitbuf_ = bb; bitcnt_ = bc; in_ = ip;
                if (bad) { err = bad; return -1; }
            }
            // ---- careful loop: near the end of the input or of the caller's buffer
            while (!block_done && o < o_end) {
                refill();
                uint32_t e = lit[bitbuf_ & ((1u << LIT_BITS) - 1)];
                if (e_kind(e) == LINK) { drop(LIT_BITS); e = lit[e_value(e) + peek(e_extra(e))]; }
                drop((e & DOUBLE_FLAG) ? e_extra(e) : e_len(e));               // of a two-literal entry only the first is taken here
                if (e_kind(e) == LITERAL) { *o++ = (uint8_t)(e >> 16); }
                else if (e_kind(e) == END_OF_BLOCK) block_done = true;
                else if (e_kind(e) == LENGTH) {
                    unsigned len = e_value(e) + peek(e_extra(e)); drop(e_extra(e));
                    uint32_t d = dst[bitbuf_ & ((1u << DIST_BITS) - 1)];
                    if (e_kind(d) == LINK) { drop(DIST_BITS); d = dst[e_value(d) + peek(e_extra(d))]; }
                    drop(e_len(d));
                    if (e_kind(d) != DISTANCE) { err = "invalid distance code"; return -1; }
                    const unsigned dist = e_value(d) + peek(e_extra(d)); drop(e_extra(d));
                    while (len && o < o_end) {
                        bool ok = true; const uint8_t b = byte_at(dist, ok);
                        if (!ok) { err = "invalid distance too far back"; return -1; }
                        *o++ = b; len--;
                    }
                    pend_len_ = len; pend_dist_ = dist;
                } else { err = "invalid literal/length code"; return -1; }
                if (overrun_ && overrun_ * 8 > bitcnt_) { err = "truncated deflate stream"; return -1; }   // used bits that were never in the file
                if (in_end_ - in_ >= 16 && o_end - o >= 258 + 16 && !pend_len_ && !block_done) break;          // back to the fast loop
            }
            if (block_done) state_ = last_block_ ? MEMBER_TRAILER : BLOCK_HEADER;
            break;
        }
        case MEMBER_TRAILER:
            fold();
            if (!check_trailer(err)) return -1;
            break;
        case DONE: break;
        }
        if (transparent_) {
            size_t n = (size_t)(in_end_ - in_);
            if (n > (size_t)(o_end - o)) n = (size_t)(o_end - o);
            memcpy(o, in_, n); o += n; in_ += n;
            if (in_ == in_end_) state_ = DONE;
            if (o == o_end) break;
        }
    }
    if (state_ == MEMBER_TRAILER) { fold(); if (!check_trailer(err)) return -1; }   // the output ended exactly at the end of a member
    if (!transparent_) fold();
    // remember the last 32 KiB for matches of the next call
    const size_t n = (size_t)(o - out);
    if (n >= 32768) { memcpy(hist_.data(), o - 32768, 32768); hist_len_ = 32768; }
    else if (n) {
        const size_t keep = hist_len_ + n > 32768 ? 32768 - n : hist_len_;
        memmove(hist_.data(), hist_.data() + (hist_len_ - keep), keep);
        memcpy(hist_.data() + keep, out, n);
        hist_len_ = keep + n;
    }
    return (long)n;
}

} // namespace mf
