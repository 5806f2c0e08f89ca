// TEST INFRASTRUCTURE: CPU stand-ins for everything mitoflex_amd/csrc/mf_devingest.cpp calls on the device -- the launch_* functions of
// mf_gzdev.h / mf_ingest.h / mf_kernels.h as obvious loops on the stub runtime's stream threads (tests/native/hipstub), and the pieces of
// mf_api.cpp it shares (device contexts, the refillable read set, the filter call) with a stand-in filter whose rule the check
// (tests/native/ingest_check.cpp) restates from the FASTQ text.  None of this is the product and nothing here is fast: it exists so that
// the ORCHESTRATION of the device ingest path -- producer, uploader, consumers, writers, ring, text-buffer pool, carry -- runs in the CPU
// suite, plain and under ThreadSanitizer, with tiny knobs.  The decode stand-in is the host decoder's speculative chunk (mf_pinflate.cpp).
#include "../../mitoflex_amd/csrc/mf_api_internal.h"
#include "../../mitoflex_amd/csrc/mf_gzdev.h"
#include "../../mitoflex_amd/csrc/mf_ingest.h"
#include "../../mitoflex_amd/csrc/mf_pinflate.h"
#include <algorithm>
#include <map>
#include <mutex>
#include <stdarg.h>
#include <string.h>
#include <vector>
#include <zlib.h>

// ------------------------------------------------------------------------------------------------ what mf_api.cpp provides in the product
static thread_local std::string t_err;
int fail(int code, const char *fmt, ...) { char b[512]; va_list ap; va_start(ap, fmt); vsnprintf(b, sizeof b, fmt, ap); va_end(ap); t_err = b; return code; }
const std::string &mf_thread_error() { return t_err; }
int phys(int device) { int n = 1; (void)hipGetDeviceCount(&n); return device % n; }
static std::mutex g_ctx_mu; static std::map<int, DevCtx> g_ctx;
int get_ctx(int device, DevCtx **out, int lane)
{
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    DevCtx &c = g_ctx[device + 4096 * lane];
    if (hipSetDevice(phys(device)) != hipSuccess) return fail(MF_E_ARG, "device %d out of range", device);
    if (!c.stream) { c.device = device; c.n_cu = 256; (void)hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking); }
    *out = &c;
    return MF_OK;
}
void reads_release(mf_reads *r)
{
    if (!r) return;
    (void)hipFree(r->d_words); (void)hipFree(r->d_offsets); (void)hipFree(r->d_npos); (void)hipFree(r->d_bits[0]);
    delete r;
}
int reads_reserve(mf_reads *r, bool reuse, uint64_t n_words, uint64_t n_reads, uint32_t uniform_len, uint64_t npos_cap, DevCtx *ctx)
{
    (void)ctx;
    if (dev_reserve(r->d_words, r->cap_words, (n_words + 64) * 4, reuse) != hipSuccess) return fail(MF_E_NOMEM, "read set: out of device memory");
    if (!uniform_len && dev_reserve(r->d_offsets, r->cap_offsets, (n_reads + 1) * 8, reuse) != hipSuccess) return fail(MF_E_NOMEM, "read set: out of device memory");
    if (dev_reserve(r->d_npos, r->cap_npos, (npos_cap ? npos_cap : 1) * 8, reuse) != hipSuccess) return fail(MF_E_NOMEM, "read set: out of device memory");
    return MF_OK;
}
int reads_finish(mf_reads *r, bool reuse, uint64_t n_words, uint64_t n_reads, uint64_t total_bases, uint32_t uniform_len, uint64_t n_npos, DevCtx *ctx)
{
    if (dev_reserve(r->d_bits[0], r->cap_bitmap, ((n_reads + 31) / 32 + 64) * 4, reuse) != hipSuccess) return fail(MF_E_NOMEM, "read set: out of device memory");
    r->v = mf::ReadsView{};
    r->v.words = r->d_words; r->v.n_words = n_words; r->v.offsets = uniform_len ? nullptr : r->d_offsets; r->v.uniform_len = uniform_len;
    r->v.n_reads = n_reads; r->v.total_bases = total_bases; r->v.npos = r->d_npos; r->v.n_npos = n_npos;
    (void)hipStreamSynchronize(ctx->stream);
    return MF_OK;
}
// The stand-in filter.  A read passes iff (sum of its 2-bit base codes + 7 * its invalid bases) is a multiple of 5 -- computed from the
// PACKED read set (words, offsets / uniform length, the list of invalid positions), so that a wrong pack, a wrong offset or a stale
// buffer changes the result; ingest_check.cpp computes the same from the text.
int filter_common(const mf_kmerset *, const mf_reads *r, uint32_t, int, uint32_t *out_bits, uint32_t *, int, mf_filter_stats_t *, uint64_t *)
{
    DevCtx *ctx; int rc = get_ctx(r->device, &ctx, r->lane); if (rc) return rc;
    const mf::ReadsView V = r->v;
    uint32_t *bits = r->d_bits[0];
    stub_enqueue(ctx->stream, [V, bits] {
        const uint64_t n = V.n_reads;
        for (uint64_t w = 0; w < (n + 31) / 32; w++) bits[w] = 0;
        uint64_t ip = 0;
        for (uint64_t i = 0; i < n; i++) {
            const uint64_t a = V.uniform_len ? i * V.uniform_len : V.offsets[i], b = V.uniform_len ? a + V.uniform_len : V.offsets[i + 1];
            uint64_t sum = 0;
            for (uint64_t g = a; g < b; g++) sum += (V.words[g >> 4] >> (2 * (g & 15))) & 3u;
            while (ip < V.n_npos && V.npos[ip] < a) ip++;
            while (ip < V.n_npos && V.npos[ip] < b) { sum += 7; ip++; }
            if (sum % 5 == 0) bits[i >> 5] |= 1u << (i & 31);
        }
    });
    if (out_bits) (void)hipMemcpyAsync(out_bits, bits, ((V.n_reads + 31) / 32) * 4, hipMemcpyDeviceToHost, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    return MF_OK;
}

namespace mf {

// ------------------------------------------------------------------------------------------------ mf_gzdev.h
void gz_preload() {}
void ingest_preload() {}
size_t gz_decode_scratch_bytes(uint32_t n_chunks) { return (size_t)n_chunks * 16; }
bool gz_decode_serial() { return false; }

hipError_t launch_gz_decode(const uint8_t *d_data, uint64_t ring_bytes, uint64_t size, uint64_t limit_bytes, uint64_t base_byte, uint64_t chunk_bytes,
                            uint32_t chunk_lo, uint32_t n_chunks, uint32_t exact_chunk, uint64_t exact_bit, uint16_t *d_sym, uint64_t sym_cap,
                            GzChunk *d_chunks, uint32_t *, hipStream_t st)
{
    stub_enqueue(st, [=] {
        const uint64_t limit = std::min(limit_bytes, size);
        for (uint32_t i = 0; i < n_chunks; i++) {
            const uint32_t c = chunk_lo + i;
            GzChunk res{0, 0, 0, GZ_NONE};
            const uint64_t nominal = (base_byte + (uint64_t)c * chunk_bytes) * 8, stop_bit = nominal + chunk_bytes * 8;
            const bool exact = c == exact_chunk;
            const uint64_t from = exact ? exact_bit : nominal;
            if (from < size * 8) {
                // the bytes the chunk may look at, out of the ring into a row (a chunk reads up to the end of a block behind its range)
                const uint64_t o = from >> 3;
                std::vector<uint8_t> lin((size_t)(limit > o ? limit - o : 0));
                for (size_t k = 0; k < lin.size(); k++) lin[k] = d_data[ring_bytes ? ((o + k) & (ring_bytes - 1)) : o + k];
                std::vector<uint16_t> sym; uint64_t s = 0, e = 0;
                const int st_ = speculative_chunk(lin.data(), lin.size(), o, from, std::min(stop_bit, size * 8), exact, sym, s, e);
                if (st_ == 1 || st_ == 2) {
                    res.start_bit = s; res.end_bit = e;
                    if (sym.size() > sym_cap) { res.status = GZ_OVERFLOW; res.n_sym = 0; res.end_bit = s; }          // (nothing fitted: the slab is decoded again with more room)
                    else { res.status = st_ == 2 ? GZ_MEMBER_END : GZ_AT_BOUNDARY; res.n_sym = (uint32_t)sym.size(); if (!sym.empty()) memcpy(d_sym + (uint64_t)i * sym_cap, sym.data(), sym.size() * 2); }
                } else if (st_ == 3) { res.start_bit = s; res.status = exact || limit < size ? GZ_FAILED : GZ_NONE; res.n_sym = limit < size ? 8 : 6; }
            }
            d_chunks[c] = res;
        }
    });
    return hipSuccess;
}

// the accepted chunks become text, in order, the obvious way: a marker points into the 32 KiB of text in front of its chunk (the first chunk's
// are the window's bytes, written in front of it first); the window behind the last chunk is left in d_window
hipError_t launch_gz_link(const uint32_t *d_acc, const uint64_t *d_acc_off, uint32_t n_acc, uint32_t, const GzChunk *chunks, uint32_t chunk_lo, uint16_t *sym, uint64_t sym_cap,
                          uint8_t *d_window, uint32_t wlen_before, uint8_t *, uint8_t *text, uint64_t text_base, uint64_t first_off, hipStream_t st)
{
    if (!n_acc) return hipSuccess;
    stub_enqueue(st, [=] {          // every chunk's tail (its last 32 Ki symbols): the markers in it point into the tails in front, which are text by now
        uint8_t *tp = text - text_base;                          // (indexed by absolute text offset)
        for (uint32_t i = 0; i < wlen_before; i++) tp[first_off - wlen_before + i] = d_window[GZ_WINDOW - wlen_before + i];
        uint64_t end = first_off; uint32_t wlen = wlen_before;
        for (uint32_t k = 0; k < n_acc; k++) {
            const uint32_t c = d_acc[k], n = chunks[c].n_sym, tail = n < GZ_WINDOW ? n : GZ_WINDOW;
            const uint64_t off = d_acc_off[k];
            const uint16_t *sp = sym + (uint64_t)(c - chunk_lo) * sym_cap;
            for (uint32_t i = n - tail; i < n; i++) tp[off + i] = (sp[i] & GZ_MARK) ? tp[off - GZ_WINDOW + (sp[i] & 0x7FFFu)] : (uint8_t)sp[i];
            end = off + n; wlen = wlen + n < GZ_WINDOW ? wlen + n : GZ_WINDOW;
        }
        for (uint32_t i = 0; i < wlen; i++) d_window[GZ_WINDOW - wlen + i] = tp[end - wlen + i];
    });
    return hipSuccess;
}
hipError_t launch_gz_resolve(const uint32_t *d_acc, const uint64_t *d_acc_off, uint32_t n_acc, uint32_t, const GzChunk *chunks, uint32_t chunk_lo, const uint16_t *sym, uint64_t sym_cap,
                             uint8_t *text, uint64_t text_base, hipStream_t st)
{
    if (!n_acc) return hipSuccess;
    stub_enqueue(st, [=] {          // every chunk's body: its window is the tail(s) in front of it
        uint8_t *tp = text - text_base;
        for (uint32_t k = 0; k < n_acc; k++) {
            const uint32_t c = d_acc[k], n = chunks[c].n_sym, body = n > GZ_WINDOW ? n - GZ_WINDOW : 0;
            const uint64_t off = d_acc_off[k];
            const uint16_t *sp = sym + (uint64_t)(c - chunk_lo) * sym_cap;
            for (uint32_t i = 0; i < body; i++) tp[off + i] = (sp[i] & GZ_MARK) ? tp[off - GZ_WINDOW + (sp[i] & 0x7FFFu)] : (uint8_t)sp[i];
        }
    });
    return hipSuccess;
}

// (the stand-in CRC of a piece is zlib's own value of the piece; finish / combine are zlib's)
hipError_t launch_gz_crc(const uint8_t *d_text, uint64_t n, uint32_t *d_piece, hipStream_t st)
{
    stub_enqueue(st, [=] { for (uint64_t p = 0; p * GZ_CRC_PIECE < n; p++) d_piece[p] = (uint32_t)crc32(0, d_text + p * GZ_CRC_PIECE, (uInt)std::min<uint64_t>(GZ_CRC_PIECE, n - p * GZ_CRC_PIECE)); });
    return hipSuccess;
}
uint32_t gz_crc_combine(uint32_t a, uint32_t b, uint64_t len_b) { return (uint32_t)crc32_combine(a, b, (z_off_t)len_b); }
uint32_t gz_crc_finish(const uint32_t *piece, uint64_t n)
{
    uint32_t r = 0;
    for (uint64_t p = 0; p * GZ_CRC_PIECE < n; p++) r = p ? gz_crc_combine(r, piece[p], std::min<uint64_t>(GZ_CRC_PIECE, n - p * GZ_CRC_PIECE)) : piece[0];
    return r;
}

// ------------------------------------------------------------------------------------------------ mf_ingest.h
hipError_t launch_bytes_from_host(void *dst, const void *src, uint64_t n, hipStream_t st) { stub_enqueue(st, [=] { if (n) memmove(dst, src, n); }); return hipSuccess; }
hipError_t launch_bytes_to_host(void *dst, const void *src, uint64_t n, hipStream_t st) { stub_enqueue(st, [=] { if (n) memmove(dst, src, n); }); return hipSuccess; }
hipError_t launch_scan_u32(const uint32_t *in, uint64_t n, uint64_t *out, uint64_t *, hipStream_t st)
{
    stub_enqueue(st, [=] { uint64_t s = 0; for (uint64_t i = 0; i < n; i++) { out[i] = s; s += in[i]; } out[n] = s; });
    return hipSuccess;
}
hipError_t launch_count_newlines(const uint8_t *text, uint64_t n, uint32_t *tile_cnt, hipStream_t st)
{
    stub_enqueue(st, [=] { for (uint64_t t = 0; t * INGEST_TILE < n; t++) { uint32_t c = 0; for (uint64_t i = t * INGEST_TILE; i < std::min<uint64_t>(n, (t + 1) * INGEST_TILE); i++) c += text[i] == '\n'; tile_cnt[t] = c; } });
    return hipSuccess;
}
hipError_t launch_line_starts(const uint8_t *text, uint64_t n, const uint64_t *, uint64_t *line_start, hipStream_t st)
{
    stub_enqueue(st, [=] { uint64_t k = 0; line_start[0] = 0; for (uint64_t i = 0; i < n; i++) if (text[i] == '\n') line_start[++k] = i + 1; });
    return hipSuccess;
}
static inline uint32_t line_len(const uint8_t *text, uint64_t a, uint64_t b) { uint64_t l = b - a - 1; if (l && text[b - 2] == '\r') l--; return (uint32_t)l; }
hipError_t launch_seq_lens(const uint8_t *text, const uint64_t *ls, uint64_t n_rec, uint32_t *seq_len, uint32_t *minmax, hipStream_t st)
{
    stub_enqueue(st, [=] { for (uint64_t r = 0; r < n_rec; r++) { const uint32_t l = line_len(text, ls[4 * r + 1], ls[4 * r + 2]); seq_len[r] = l; minmax[0] = std::min(minmax[0], l); minmax[1] = std::max(minmax[1], l); } });
    return hipSuccess;
}
uint64_t pack_blocks(uint64_t total_bases, uint64_t base)
{
    if (!total_bases) return 0;
    const uint64_t nw = ((base + total_bases + 15) >> 4) - (base >> 4);
    return (nw + 255) / 256;
}
hipError_t launch_pack(const uint8_t *text, const uint64_t *ls, const uint64_t *offsets, uint32_t uniform_len, uint64_t n_rec, uint64_t total_bases, uint64_t base,
                       uint32_t *words, uint32_t *inv_cnt, const uint64_t *inv_base, uint64_t *npos, hipStream_t st)
{
    stub_enqueue(st, [=] {
        const uint64_t nb = pack_blocks(total_bases, base), w0 = base >> 4;
        std::vector<uint64_t> k(nb, 0);
        if (!npos) { for (uint64_t b = 0; b < nb; b++) inv_cnt[b] = 0; for (uint64_t w = w0 + (base & 15 ? 1 : 0); w < ((base + total_bases + 15) >> 4); w++) words[w] = 0; }
        uint64_t g = base;
        for (uint64_t r = 0; r < n_rec; r++) {
            const uint64_t len = uniform_len ? uniform_len : offsets[r + 1] - offsets[r];
            const uint8_t *src = text + ls[4 * r + 1];
            for (uint64_t i = 0; i < len; i++, g++) {
                uint32_t code = 0; bool inv = false;
                switch (src[i] & 0xDF) { case 'A': code = 0; break; case 'C': code = 1; break; case 'G': code = 2; break; case 'T': code = 3; break; default: inv = true; }
                const uint64_t blk = ((g >> 4) - w0) / 256;
                if (!npos) { words[g >> 4] |= code << (2 * (g & 15)); if (inv) inv_cnt[blk]++; }
                else if (inv) npos[inv_base[blk] + k[blk]++] = g;
            }
        }
    });
    return hipSuccess;
}
struct RecSpan { uint64_t h, s, q; uint32_t hl, sl, ql; };
static inline RecSpan rec_span(const uint8_t *text, const uint64_t *ls, uint64_t r)
{
    RecSpan x; x.h = ls[4 * r]; x.s = ls[4 * r + 1]; x.q = ls[4 * r + 3];
    x.hl = line_len(text, ls[4 * r], ls[4 * r + 1]); x.sl = line_len(text, ls[4 * r + 1], ls[4 * r + 2]); x.ql = line_len(text, ls[4 * r + 3], ls[4 * r + 4]);
    return x;
}
hipError_t launch_sel_lens(const uint8_t *text, const uint64_t *ls, const uint32_t *sel, uint64_t n_sel, uint32_t *out_len, hipStream_t st)
{
    stub_enqueue(st, [=] { for (uint64_t i = 0; i < n_sel; i++) { const RecSpan x = rec_span(text, ls, sel[i]); out_len[i] = x.hl + x.sl + x.ql + 5; } });
    return hipSuccess;
}
static void write_record(uint8_t *o, const uint8_t *h, uint32_t hl, const uint8_t *s, uint32_t sl, const uint8_t *q, uint32_t ql)
{
    memcpy(o, h, hl); o += hl; *o++ = '\n'; memcpy(o, s, sl); o += sl; *o++ = '\n'; *o++ = '+'; *o++ = '\n'; memcpy(o, q, ql); o += ql; *o = '\n';
}
hipError_t launch_sel_gather(const uint8_t *text, const uint64_t *ls, const uint32_t *sel, uint64_t n_sel, const uint64_t *out_off, uint8_t *out, hipStream_t st)
{
    stub_enqueue(st, [=] { for (uint64_t i = 0; i < n_sel; i++) { const RecSpan x = rec_span(text, ls, sel[i]); write_record(out + out_off[i], text + x.h, x.hl, text + x.s, x.sl, text + x.q, x.ql); } });
    return hipSuccess;
}

// ---- the quality filter's kernels (the reference's rules: filter/filter_bin/src/main.rs:236-268, 302-321)
hipError_t launch_qual_scan(const uint8_t *text, const uint64_t *ls, uint64_t n_rec, uint64_t start, uint64_t cap, uint32_t quality, uint64_t ns, uint32_t *bad, uint8_t *flags,
                            uint32_t *cut_sl, uint32_t *cut_ql, uint32_t *olen, uint32_t *first_flag, hipStream_t st)
{
    stub_enqueue(st, [=] {
        for (uint64_t r = 0; r < n_rec; r++) {
            const RecSpan x = rec_span(text, ls, r);
            uint32_t fl = 0;
            auto high = [&](uint64_t a, uint32_t l) { for (uint32_t i = 0; i < l; i++) if (text[a + i] >= 0x80) return true; return false; };
            if (high(x.h, x.hl) || high(x.s, x.sl) || high(x.q, x.ql)) fl |= QF_HIGH;
            if (x.sl < start || x.ql < start) fl |= QF_SHORT;
            const uint32_t sl = x.sl < start ? 0 : (uint32_t)std::min<uint64_t>(x.sl - start, cap), ql = x.ql < start ? 0 : (uint32_t)std::min<uint64_t>(x.ql - start, cap);
            uint64_t n_count = 0; uint32_t b = 0;
            for (uint32_t i = 0; i < sl; i++) n_count += text[x.s + start + i] == 'N';
            for (uint32_t i = 0; i < ql; i++) b += text[x.q + start + i] <= quality;
            if (n_count > ns) fl |= QF_NFAIL;
            bad[r] = b; flags[r] = (uint8_t)fl; cut_sl[r] = sl; cut_ql[r] = ql; olen[r] = x.hl + sl + ql + 5;
            if ((fl & (QF_HIGH | QF_SHORT | QF_LONG)) && (uint32_t)r < *first_flag) *first_flag = (uint32_t)r;
        }
    });
    return hipSuccess;
}
// (any 64-bit hash of the cut sequence serves the de-duplication's semantics here; the real one is SipHash-1-3)
hipError_t launch_qual_hash(const uint8_t *text, const uint64_t *ls, uint64_t n_rec, uint64_t start, const uint32_t *cut_sl, uint64_t *hashes, hipStream_t st)
{
    stub_enqueue(st, [=] { for (uint64_t r = 0; r < n_rec; r++) { uint64_t h = 1469598103934665603ull; const uint8_t *s = text + ls[4 * r + 1] + start; for (uint32_t i = 0; i < cut_sl[r]; i++) h = (h ^ s[i]) * 1099511628211ull; hashes[r] = h; } });
    return hipSuccess;
}
hipError_t launch_qual_decide(uint64_t n, bool pe, bool trunc, float limit, const uint32_t *bad1, const uint8_t *fl1, const uint32_t *sl1, const uint32_t *ql1,
                              const uint32_t *bad2, const uint8_t *fl2, uint8_t *alive, hipStream_t st)
{
    stub_enqueue(st, [=] {
        for (uint64_t i = 0; i < n; i++) {
            bool drop = false;
            if (!trunc) {
                drop = (fl1[i] & QF_NFAIL) || (pe && (fl2[i] & QF_NFAIL));
                if (!drop) { const float cf = (float)(pe ? sl1[i] : ql1[i]) * limit; const uint64_t cutoff = !(cf > 0.0f) ? 0 : (uint64_t)cf; drop = bad1[i] >= cutoff || (pe && bad2[i] >= cutoff); }
            }
            alive[i] = drop ? 0 : 1;
        }
    });
    return hipSuccess;
}
hipError_t launch_qual_keep(uint64_t n, const uint8_t *alive, const uint8_t *dup, const uint32_t *olen, uint8_t *keep, uint32_t *out_len, unsigned long long *kept, hipStream_t st)
{
    stub_enqueue(st, [=] { for (uint64_t i = 0; i < n; i++) { const uint32_t k = alive[i] && !(dup && dup[i]); if (keep) keep[i] = (uint8_t)k; out_len[i] = k ? olen[i] : 0; if (k && kept) ++*kept; } });
    return hipSuccess;
}
hipError_t launch_qual_gather(const uint8_t *text, const uint64_t *ls, uint64_t n_rec, uint64_t start, const uint32_t *cut_sl, const uint32_t *cut_ql, const uint32_t *out_len,
                              const uint64_t *out_off, uint8_t *out, hipStream_t st)
{
    stub_enqueue(st, [=] {
        for (uint64_t r = 0; r < n_rec; r++) {
            if (!out_len[r]) continue;
            const uint64_t l0 = ls[4 * r], l1 = ls[4 * r + 1], l3 = ls[4 * r + 3];
            write_record(out + out_off[r], text + l0, line_len(text, l0, l1), text + l1 + start, cut_sl[r], text + l3 + start, cut_ql[r]);
        }
    });
    return hipSuccess;
}
// the de-duplication set: first occurrence by file index wins (keys zeroed, first all ones; the hash value 0 has a slot of its own)
hipError_t launch_dedup(const uint64_t *hashes, const uint8_t *alive, uint32_t n, uint64_t base, unsigned long long *keys, unsigned long long *first, uint64_t slots,
                        unsigned long long *zero_idx, unsigned long long *n_keys, uint8_t *dup, hipStream_t st)
{
    stub_enqueue(st, [=] {
        for (uint32_t i = 0; i < n; i++) {
            dup[i] = 0;
            if (!alive[i]) continue;
            const uint64_t h = hashes[i];
            if (!h) { if (*zero_idx == ~0ull) *zero_idx = base + i; else dup[i] = 1; continue; }
            uint64_t s = h & (slots - 1);
            while (keys[s] && keys[s] != h) s = (s + 1) & (slots - 1);
            if (!keys[s]) { keys[s] = h; first[s] = base + i; ++*n_keys; } else dup[i] = 1;
        }
    });
    return hipSuccess;
}
hipError_t launch_dedup_rehash(const unsigned long long *old_keys, const unsigned long long *old_first, uint64_t old_slots, unsigned long long *keys, unsigned long long *first,
                               uint64_t slots, hipStream_t st)
{
    stub_enqueue(st, [=] {
        for (uint64_t i = 0; i < old_slots; i++) {
            if (!old_keys[i]) continue;
            uint64_t s = old_keys[i] & (slots - 1);
            while (keys[s]) s = (s + 1) & (slots - 1);
            keys[s] = old_keys[i]; first[s] = old_first[i];
        }
    });
    return hipSuccess;
}

} // namespace mf
