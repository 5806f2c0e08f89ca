// Parallel decoding of one gzip stream (see mf_pinflate.h).
#include "mf_pinflate.h"
#include "mf_host.h"            // DefaultInitAlloc
#include "mf_inflate_core.h"

#include <atomic>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <immintrin.h>
#include <zlib.h>               // crc32, crc32_combine only

namespace mf {
using namespace inflate_core;

namespace {

constexpr uint32_t LMASK = (1u << LIT_BITS) - 1, DMASK = (1u << DIST_BITS) - 1;
constexpr size_t WINDOW = 32768;
constexpr uint16_t MARK = 0x8000;                     // symbol = MARK | index into the unknown window

// LSB-first bit reader over the mapped file; bits past the end read as zero and are counted
struct Bits {
    const uint8_t *base, *p, *end; uint64_t bb = 0; unsigned bc = 0; size_t over = 0;
    Bits(const uint8_t *b, size_t n) : base(b), p(b), end(b + n) {}
    inline void refill()
    {
        if (end - p >= 8) { uint64_t v; memcpy(&v, p, 8); bb |= v << bc; p += (63 - bc) >> 3; bc |= 56; }
        else while (bc <= 56) { uint64_t b = 0; if (p < end) b = *p++; else over++; bb |= b << bc; bc += 8; }
    }
    void seek(size_t bit) { p = base + (bit >> 3); if (p > end) p = end; bb = 0; bc = 0; over = 0; refill(); drop((unsigned)(bit & 7)); }
    size_t bitpos() const { return ((size_t)(p - base) + over) * 8 - bc; }
    inline uint32_t peek(unsigned n) const { return (uint32_t)(bb & ((1ULL << n) - 1)); }
    inline void drop(unsigned n) { bb >>= n; bc -= n; }
    bool need(unsigned n) { if (bc < n) refill(); return over == 0 || over * 8 + n <= bc; }
    bool sound() const { return over * 8 <= bc; }      // no bit from beyond the end of the file has been consumed
};

struct Tables { std::vector<uint32_t> lit, dist; };

template <class Sym> struct Out {
    std::vector<Sym, DefaultInitAlloc<Sym>> v; size_t n = 0, prefix = 0;
    void grow() { v.resize(v.size() * 2 + ((size_t)1 << 20)); }
};

uint64_t kraft(const uint8_t *lens, unsigned n) { uint64_t k = 0; for (unsigned i = 0; i < n; i++) if (lens[i]) k += (uint64_t)1 << (15 - lens[i]); return k; }

// Dynamic block header after the three type bits.  strict: only codes a compressor would emit (complete).
bool read_dynamic_header(Bits &in, Tables &t, bool strict, const char *&err)
{
    if (!in.need(14)) { err = "truncated deflate stream"; return false; }
    const unsigned hlit = in.peek(5) + 257; in.drop(5);
    const unsigned hdist = in.peek(5) + 1; in.drop(5);
    const unsigned hclen = in.peek(4) + 4; in.drop(4);
    if (hlit > 286 || hdist > 30) { err = "too many length or distance symbols"; return false; }
    uint8_t pre[19] = {0};
    for (unsigned i = 0; i < hclen; i++) { if (!in.need(3)) { err = "truncated deflate stream"; return false; } pre[PRECODE_ORDER[i]] = (uint8_t)in.peek(3); in.drop(3); }
    if (strict && kraft(pre, 19) != (1u << 15)) { err = "incomplete precode"; return false; }
    std::vector<uint32_t> ptab;
    if (!build_table(pre, 19, PRE_BITS, ptab, [](unsigned s) { return entry(0, LITERAL, 0, s); })) { err = "invalid code lengths set"; return false; }
    uint8_t lens[286 + 30 + 138];
    unsigned i = 0;
    while (i < hlit + hdist) {
        if (!in.need(14)) { err = "truncated deflate stream"; return false; }
        const uint32_t e = ptab[in.peek(PRE_BITS)];
        if (e_kind(e) != LITERAL || e_len(e) == 0) { err = "invalid code lengths set"; return false; }
        in.drop(e_len(e));
        const unsigned s = e_value(e);
        if (s < 16) { lens[i++] = (uint8_t)s; continue; }
        unsigned rep, val = 0;
        if (s == 16) { if (i == 0) { err = "invalid bit length repeat"; return false; } val = lens[i - 1]; rep = 3 + in.peek(2); in.drop(2); }
        else if (s == 17) { rep = 3 + in.peek(3); in.drop(3); }
        else { rep = 11 + in.peek(7); in.drop(7); }
        if (i + rep > hlit + hdist) { err = "invalid bit length repeat"; return false; }
        while (rep--) lens[i++] = (uint8_t)val;
    }
    if (lens[256] == 0) { err = "invalid code -- missing end-of-block"; return false; }
    if (strict) {
        if (kraft(lens, hlit) != (1u << 15)) { err = "incomplete literal/length code"; return false; }
        unsigned nd = 0; for (unsigned k = 0; k < hdist; k++) nd += lens[hlit + k] != 0;
        if (nd > 1 && kraft(lens + hlit, hdist) != (1u << 15)) { err = "incomplete distance code"; return false; }
    }
    if (!build_table(lens, hlit, LIT_BITS, t.lit, lit_entry) || !build_table(lens + hlit, hdist, DIST_BITS, t.dist, dist_entry)) {
        err = "invalid Huffman code in deflate stream"; return false;
    }
    pair_literals(t.lit);
    return true;
}

// One Huffman-coded block body up to its end-of-block symbol.  Sym = uint8_t: out holds real bytes and
// starts with the known window.  Sym = uint16_t: the window is unknown, matches into it become markers.
template <class Sym>
bool decode_block(Bits &in, const Tables &t, Out<Sym> &out, const char *&err)
{
    constexpr bool MARKING = sizeof(Sym) == 2;
    constexpr unsigned PER8 = 8 / sizeof(Sym);
    const uint32_t *lit = t.lit.data(), *dst = t.dist.data();
    uint64_t bb = in.bb; unsigned bc = in.bc; const uint8_t *p = in.p; size_t over = in.over;   // locals: stores through o may alias anything
    Sym *o = out.v.data() + out.n, *o_end = out.v.data() + out.v.size();
    const char *bad = nullptr;
    bool done = false;
#define PZ_DROP(n) do { const unsigned n_ = (n); bb >>= n_; bc -= n_; } while (0)
#define PZ_PEEK(n) ((uint32_t)(bb & ((1ULL << (n)) - 1)))
    while (!done) {
        if (o_end - o < 600) { out.n = (size_t)(o - out.v.data()); out.grow(); o = out.v.data() + out.n; o_end = out.v.data() + out.v.size(); }
        if (in.end - p >= 8) { uint64_t v; memcpy(&v, p, 8); bb |= v << bc; p += (63 - bc) >> 3; bc |= 56; }
        else {
            while (bc <= 56) { uint64_t b = 0; if (p < in.end) b = *p++; else over++; bb |= b << bc; bc += 8; }
            if (over > 16) { bad = "truncated deflate stream"; break; }
        }
        uint32_t e = lit[bb & LMASK];
        if (e & LITERAL_FLAG) {                                          // up to four lookups of one or two literals on one refill
#define PZ_PUT() do { o[0] = (Sym)((e >> 16) & 0xFFu); o[1] = (Sym)(e >> 24); o += 1 + ((e >> 14) & 1u); PZ_DROP(e & 31u); } while (0)
            PZ_PUT();
            e = lit[bb & LMASK];
            if (e & LITERAL_FLAG) {
                PZ_PUT();
                e = lit[bb & LMASK];
                if (e & LITERAL_FLAG) { PZ_PUT(); e = lit[bb & LMASK]; if (e & LITERAL_FLAG) PZ_PUT(); }
            }
#undef PZ_PUT
            continue;
        }
        if (e_kind(e) == LINK) {
            PZ_DROP(LIT_BITS); e = lit[e_value(e) + PZ_PEEK(e_extra(e))];
            PZ_DROP(e_len(e));
            if (e & LITERAL_FLAG) { *o++ = (Sym)((e >> 16) & 0xFFu); continue; }
        } else PZ_DROP(e_len(e));
        if (e_kind(e) == END_OF_BLOCK) { done = true; break; }
        if (e_kind(e) != LENGTH) { bad = "invalid literal/length code"; break; }
        unsigned len = e_value(e) + PZ_PEEK(e_extra(e)); PZ_DROP(e_extra(e));
        uint32_t d = dst[bb & DMASK];
        if (e_kind(d) == LINK) { PZ_DROP(DIST_BITS); d = dst[e_value(d) + PZ_PEEK(e_extra(d))]; }
        PZ_DROP(e_len(d));
        if (e_kind(d) != DISTANCE) { bad = "invalid distance code"; break; }
        const unsigned dist = e_value(d) + PZ_PEEK(e_extra(d)); PZ_DROP(e_extra(d));
        Sym *const base = out.v.data();
        if (dist <= (size_t)(o - base)) {
            const Sym *s = o - dist; Sym *q = o; o += len;
            if (dist >= PER8) { do { memcpy(q, s, 8); q += PER8; s += PER8; } while (q < o); }
            else if (dist == 1) { const Sym c = *s; do { *q++ = c; } while (q < o); }
            else { do { *q++ = *s++; } while (q < o); }
        } else if (MARKING) {
            while (len--) {
                const ptrdiff_t idx = (ptrdiff_t)(o - base) - (ptrdiff_t)dist;       // >= -32768: a distance never exceeds the window
                *o = idx >= 0 ? base[idx] : (Sym)(MARK | (uint16_t)((ptrdiff_t)WINDOW + idx));
                o++;
            }
        } else { bad = "invalid distance too far back"; break; }
    }
#undef PZ_DROP
#undef PZ_PEEK
    in.bb = bb; in.bc = bc; in.p = p; in.over = over;
    out.n = (size_t)(o - out.v.data());
    if (bad) { err = bad; return false; }
    if (!in.sound()) { err = "truncated deflate stream"; return false; }
    return true;
}

enum Stop { AT_BOUNDARY, MEMBER_END, FAILED };

// Decode whole blocks from the reader's position; stops in front of the first block that starts at
// or after stop_bit, or behind the final block of the member, or (memory bound: a few bytes of
// deflate can expand to megabytes) at the first block boundary after max_out symbols.
template <class Sym>
Stop decode_until(Bits &in, Tables &t, Out<Sym> &out, size_t stop_bit, size_t max_out, const char *&err)
{
    for (;;) {
        if (in.bitpos() >= stop_bit || out.n - out.prefix >= max_out) return AT_BOUNDARY;
        if (!in.need(3)) { err = "truncated deflate stream"; return FAILED; }
        const bool final = in.peek(1); in.drop(1);
        const unsigned type = in.peek(2); in.drop(2);
        if (type == 0) {
            in.drop(in.bc & 7);
            if (!in.need(32)) { err = "truncated deflate stream"; return FAILED; }
            const unsigned len = in.peek(16); in.drop(16);
            const unsigned nlen = in.peek(16); in.drop(16);
            if ((len ^ nlen) != 0xFFFFu) { err = "invalid stored block length"; return FAILED; }
            const size_t byte = in.bitpos() >> 3;
            if (byte + len > (size_t)(in.end - in.base)) { err = "truncated deflate stream"; return FAILED; }
            while (out.v.size() - out.n < len + 16) out.grow();
            for (unsigned i = 0; i < len; i++) out.v[out.n + i] = (Sym)in.base[byte + i];
            out.n += len;
            in.seek((byte + len) * 8);
        } else if (type == 3) { err = "invalid deflate block type"; return FAILED; }
        else {
            if (type == 1) {
                uint8_t lens[320]; fixed_lengths(lens);
                build_table(lens, 288, LIT_BITS, t.lit, lit_entry); build_table(lens + 288, 32, DIST_BITS, t.dist, dist_entry);
                pair_literals(t.lit);
            } else if (!read_dynamic_header(in, t, false, err)) return FAILED;
            if (!decode_block<Sym>(in, t, out, err)) return FAILED;
        }
        if (final) return MEMBER_END;
    }
}

// First bit position in [from_bit, to_bit) that parses as a non-final dynamic-Huffman block header with
// complete codes; SIZE_MAX if there is none.
size_t find_block(const uint8_t *data, size_t size, size_t from_bit, size_t to_bit)
{
    Tables scratch;
    Bits in(data, size);
    for (size_t byte = from_bit >> 3; byte * 8 < to_bit && byte + 16 <= size; byte++) {
        uint64_t lo, hi; memcpy(&lo, data + byte, 8); memcpy(&hi, data + byte + 8, 8);
        const unsigned __int128 w = (unsigned __int128)lo | ((unsigned __int128)hi << 64);
        for (unsigned r = (byte * 8 < from_bit ? (unsigned)(from_bit & 7) : 0); r < 8; r++) {
            const size_t bit = byte * 8 + r;
            if (bit >= to_bit) break;
            const unsigned __int128 v = w >> r;
            const uint32_t head = (uint32_t)v;
            if ((head & 7u) != 4u) continue;                             // BFINAL = 0, BTYPE = 10b (dynamic)
            if (((head >> 3) & 31u) > 29u || ((head >> 8) & 31u) > 29u) continue;
            const unsigned hclen = ((head >> 13) & 15u) + 4;
            unsigned k = 0;                                              // Kraft sum of the precode in units of 2^-7
            for (unsigned i = 0; i < hclen; i++) { const unsigned l = (unsigned)(v >> (17 + 3 * i)) & 7u; if (l) k += 128u >> l; }
            if (k != 128u) continue;
            in.seek(bit + 3);
            const char *why = nullptr;
            if (read_dynamic_header(in, scratch, true, why)) return bit;
        }
    }
    return SIZE_MAX;
}

} // namespace

// CRC-32 (the gzip polynomial, reflected) by carry-less multiplication: four 128-bit accumulators are folded forward 64 bytes
// at a time, then into one, then reduced to 32 bits (Barrett) -- the scheme of Intel's "Fast CRC computation for generic
// polynomials using PCLMULQDQ"; the constants are x^(n) mod P for the fold distances, bit-reflected.  ~10x zlib's table walk,
// which was a seventh of the parallel reader's CPU time.  Works on the raw register (no pre/post inversion), whole 16-byte blocks.
__attribute__((target("pclmul,sse4.1"))) static inline __m128i crc_ld(const uint8_t *q) { return _mm_loadu_si128(reinterpret_cast<const __m128i *>(q)); }
__attribute__((target("pclmul,sse4.1"))) static inline __m128i crc_fold(__m128i x, __m128i k, __m128i data)
{
    return _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x, k, 0x00), _mm_clmulepi64_si128(x, k, 0x11)), data);
}
__attribute__((target("pclmul,sse4.1")))
static uint32_t crc32_fold_pclmul(uint32_t reg, const uint8_t *p, size_t n)        // n >= 64, n % 16 == 0
{
#define ld crc_ld
#define fold crc_fold
    const __m128i k_r2r1 = _mm_set_epi64x(0x00000001c6e41596LL, 0x0000000154442bd4LL);
    const __m128i k_r4r3 = _mm_set_epi64x(0x00000000ccaa009eLL, 0x00000001751997d0LL);
    const __m128i k_r5 = _mm_set_epi64x(0, 0x0000000163cd6124LL);
    const __m128i k_poly = _mm_set_epi64x(0x00000001F7011641LL, 0x00000001DB710641LL);
    const __m128i mask32 = _mm_set_epi32(0, 0, 0, -1);
    __m128i x1 = _mm_xor_si128(ld(p), _mm_cvtsi32_si128((int)reg)), x2 = ld(p + 16), x3 = ld(p + 32), x4 = ld(p + 48);
    p += 64; n -= 64;
    while (n >= 64) {
        x1 = fold(x1, k_r2r1, ld(p)); x2 = fold(x2, k_r2r1, ld(p + 16)); x3 = fold(x3, k_r2r1, ld(p + 32)); x4 = fold(x4, k_r2r1, ld(p + 48));
        p += 64; n -= 64;
    }
    x1 = fold(x1, k_r4r3, x2); x1 = fold(x1, k_r4r3, x3); x1 = fold(x1, k_r4r3, x4);
    while (n >= 16) { x1 = fold(x1, k_r4r3, ld(p)); p += 16; n -= 16; }
    // 128 -> 64 bits (this also appends the 32 zero bits of the CRC definition), 64 -> 32, Barrett
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), _mm_clmulepi64_si128(k_r4r3, x1, 0x01));
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 4), _mm_clmulepi64_si128(_mm_and_si128(x1, mask32), k_r5, 0x00));
    __m128i t = _mm_and_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), k_poly, 0x10), mask32);
    t = _mm_clmulepi64_si128(t, k_poly, 0x00);
    return (uint32_t)_mm_extract_epi32(_mm_xor_si128(x1, t), 1);
#undef ld
#undef fold
}

// same contract as zlib's crc32()
uint32_t crc32_fast(uint32_t crc, const uint8_t *p, size_t n)
{
    static const bool hw = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1") && !getenv("MF_NO_SIMD");
    if (hw && n >= 64) {
        const size_t bulk = n & ~(size_t)15;
        crc = ~crc32_fold_pclmul(~crc, p, bulk);
        p += bulk; n -= bulk;
    }
    while (n) { const size_t k = n < ((size_t)1 << 30) ? n : ((size_t)1 << 30); crc = (uint32_t)crc32(crc, p, (uInt)k); p += k; n -= k; }
    return crc;
}

// 16-bit symbols -> bytes: a marker (MARK | index) is looked up in the 32 KiB window the chunk did not know, anything else is
// the byte itself.  Markers cluster (in FASTQ: the copied prefix of a header), so most 32-symbol blocks are plain narrowing.
__attribute__((target("avx2")))
static void resolve_symbols_avx2(const uint16_t *s, size_t n, const uint8_t *w, uint8_t *d)
{
    size_t k = 0;
    for (; k + 32 <= n; k += 32) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + k)), b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + k + 16));
        if ((uint32_t)_mm256_movemask_epi8(_mm256_or_si256(a, b)) & 0xAAAAAAAAu) {          // a MARK bit (the top bit of a high byte) somewhere
            for (size_t i = k; i < k + 32; i++) { const uint16_t v = s[i]; d[i] = (v & MARK) ? w[v & 0x7FFF] : (uint8_t)v; }
        } else
            _mm256_storeu_si256(reinterpret_cast<__m256i *>(d + k), _mm256_permute4x64_epi64(_mm256_packus_epi16(a, b), 0xD8));
    }
    for (; k < n; k++) { const uint16_t v = s[k]; d[k] = (v & MARK) ? w[v & 0x7FFF] : (uint8_t)v; }
}
void resolve_symbols(const uint16_t *s, size_t n, const uint8_t *w, uint8_t *d)
{
    static const bool simd = __builtin_cpu_supports("avx2") && !getenv("MF_NO_SIMD");
    if (simd) { resolve_symbols_avx2(s, n, w, d); return; }
    for (size_t k = 0; k < n; k++) { const uint16_t v = s[k]; d[k] = (v & MARK) ? w[v & 0x7FFF] : (uint8_t)v; }
}

namespace {

uint32_t crc32_parallel(uint32_t crc, const uint8_t *p, size_t n, int threads)
{
    const size_t slice = (size_t)2 << 20;
    if (n < 2 * slice || threads < 2) return crc32_fast(crc, p, n);
    size_t parts = n / slice; if (parts > (size_t)threads) parts = (size_t)threads;
    std::vector<uint32_t> c(parts); std::vector<size_t> len(parts);
    std::vector<std::thread> th;
    for (size_t i = 0; i < parts; i++) {
        const size_t a = n * i / parts, b = n * (i + 1) / parts;
        len[i] = b - a;
        auto job = [&c, p, a, i, b] { c[i] = crc32_parallel(0, p + a, b - a, 1); };
        if (i + 1 < parts) th.emplace_back(job); else job();
    }
    for (auto &t : th) t.join();
    for (size_t i = 0; i < parts; i++) crc = (uint32_t)crc32_combine(crc, c[i], (z_off_t)len[i]);
    return crc;
}

void slide_window(std::vector<uint8_t> &window, size_t &wlen, const uint8_t *bytes, size_t n)
{
    if (n >= WINDOW) { memcpy(window.data(), bytes + n - WINDOW, WINDOW); wlen = WINDOW; return; }
    const size_t keep = wlen + n > WINDOW ? WINDOW - n : wlen;           // the window is right-aligned in its buffer
    memmove(window.data() + WINDOW - keep - n, window.data() + WINDOW - keep, keep);
    memcpy(window.data() + WINDOW - n, bytes, n);
    wlen = keep + n;
}

} // namespace

// Serial decode across a gap for the device decoder (mf_devingest.cpp): from a known block boundary with the known window up
// to the first block boundary at or behind to_bit, the end of the member, or ~256 MB of output, whichever comes first.
bool inflate_gap(const uint8_t *data, size_t size, uint64_t from_bit, uint64_t to_bit, const uint8_t *window, size_t wlen,
                 std::vector<uint8_t> &out, uint64_t &end_bit, bool &member_end, std::string &err)
{
    Out<uint8_t> f;
    f.v.resize(wlen + ((size_t)1 << 20));
    if (wlen) memcpy(f.v.data(), window + WINDOW - wlen, wlen);
    f.n = f.prefix = wlen;
    Bits in(data, size); in.seek((size_t)from_bit);
    Tables t; const char *why = nullptr;
    const Stop st = decode_until<uint8_t>(in, t, f, (size_t)to_bit, (size_t)256 << 20, why);
    if (st == FAILED) { err = why ? why : "damaged deflate stream"; return false; }
    out.assign(f.v.begin() + (ptrdiff_t)f.prefix, f.v.begin() + (ptrdiff_t)f.n);
    end_bit = in.bitpos(); member_end = st == MEMBER_END;
    return true;
}

// One speculative chunk, decoded the way fill() decodes the chunks of a group: the CPU stand-in for the device decode kernel in
// tests/native/ingest_stub.cpp (the orchestration of the device ingest path under ThreadSanitizer) calls this.
int speculative_chunk(const uint8_t *data, size_t size, uint64_t origin, uint64_t from_bit, uint64_t to_bit, bool exact,
                      std::vector<uint16_t> &sym, uint64_t &start_bit, uint64_t &end_bit)
{
    const uint64_t o = origin * 8;
    size_t s = (size_t)(from_bit - o);
    if (!exact) { s = find_block(data, size, (size_t)(from_bit - o), (size_t)std::min<uint64_t>(to_bit - o, (uint64_t)size * 8)); if (s == SIZE_MAX) return 0; }
    Bits in(data, size); in.seek(s);
    Tables t; const char *why = nullptr;
    Out<uint16_t> out; out.v.resize((size_t)1 << 16);
    const Stop st = decode_until<uint16_t>(in, t, out, (size_t)(to_bit - o), (size_t)1 << 30, why);
    start_bit = o + s; end_bit = o + in.bitpos();
    if (st == FAILED) return 3;
    sym.assign(out.v.begin(), out.v.begin() + (ptrdiff_t)out.n);
    return st == MEMBER_END ? 2 : 1;
}

struct ChunkResult { bool valid = false; size_t start = 0, end = 0; Stop stop = FAILED; Out<uint16_t> sym; };
struct ParallelGzReader::Scratch { std::vector<ChunkResult> res; Out<uint8_t> first; };

ParallelGzReader::~ParallelGzReader() { if (pre_.joinable()) pre_.join(); delete scratch_; }

void ParallelGzReader::start_prefetch()
{
    pre_running_ = true;
    pre_ = std::thread([this] { pre_err_.clear(); pre_ok_ = fill(nbuf_, pre_err_); });
}

void ParallelGzReader::open(const uint8_t *data, size_t size, int threads, size_t chunk_bytes)
{
    if (pre_.joinable()) pre_.join();
    pre_running_ = false; pre_ok_ = true; nbuf_.clear();
    data_ = data; size_ = size; threads_ = threads < 2 ? 2 : threads; chunk_ = chunk_bytes < 4096 ? 4096 : chunk_bytes;
    cur_bit_ = 0; in_member_ = false; done_ = size == 0; any_member_ = false;
    transparent_ = size >= 1 && !(size >= 2 && data[0] == 0x1f && data[1] == 0x8b);       // not gzip: handed through, as gzread does
    window_.assign(WINDOW, 0); wlen_ = 0; crc_ = 0; member_out_ = 0;
    obuf_.clear(); opos_ = 0;
    chunks_linked = chunks_discarded = gap_fill_bytes = 0;
}

// BGZF (bgzip, htslib): a gzip file made of many small members, each announcing its own compressed size in an
// extra subfield 'B','C' (BSIZE = member bytes - 1).  Returns the member's total size, 0 if `pos` is not such a member.
static size_t bgzf_member_size(const uint8_t *d, size_t size, size_t pos, size_t &cdata_off)
{
    if (size - pos < 18 || d[pos] != 0x1f || d[pos + 1] != 0x8b || d[pos + 2] != 8 || (d[pos + 3] & 4) == 0) return 0;
    if (d[pos + 3] & ~4u) return 0;                                      // only FEXTRA set, as bgzip writes it
    const size_t xlen = d[pos + 10] | ((size_t)d[pos + 11] << 8);
    if (size - pos < 12 + xlen + 8) return 0;
    for (size_t q = pos + 12, e = pos + 12 + xlen; q + 4 <= e;) {
        const size_t slen = d[q + 2] | ((size_t)d[q + 3] << 8);
        if (d[q] == 'B' && d[q + 1] == 'C' && slen == 2 && q + 6 <= e) {
            const size_t total = (d[q + 4] | ((size_t)d[q + 5] << 8)) + 1;
            if (total < 12 + xlen + 8 || pos + total > size) return 0;
            cdata_off = 12 + xlen;
            return total;
        }
        q += 4 + slen;
    }
    return 0;
}

// A run of BGZF members starting at cur_bit_: every member is an independent deflate stream of known compressed
// and uncompressed size, so they are decoded side by side, each straight into its place of the output.
bool ParallelGzReader::fill_bgzf(ByteBuf &obuf_, std::string &err, bool &handled)
{
    handled = false;
    struct Member { size_t cdata, cend; uint32_t crc, isize; size_t out_off; };
    std::vector<Member> ms;
    size_t pos = cur_bit_ >> 3, total = 0;
    const size_t max_out = (size_t)threads_ * ((size_t)8 << 20);
    while (pos < size_ && total < max_out) {
        size_t coff = 0;
        const size_t sz = bgzf_member_size(data_, size_, pos, coff);
        if (!sz) break;
        Member m; m.cdata = pos + coff; m.cend = pos + sz - 8; m.out_off = total;
        memcpy(&m.crc, data_ + pos + sz - 8, 4); memcpy(&m.isize, data_ + pos + sz - 4, 4);
        // a BGZF block holds at most 64 KiB of text; a trailer that claims more is damage or not BGZF, and taking its word
        // would size buffers from a field an input file controls (up to 4 GiB a worker): leave it to the general decoder
        if (m.isize > 65536) break;
        total += m.isize;
        ms.push_back(m);
        pos += sz;
    }
    if (ms.empty()) return true;
    handled = true;
    obuf_.resize(total);
    std::atomic<size_t> next{0}; std::atomic<int> bad{0};
    auto work = [&] {
        Tables t; Out<uint8_t> o;
        for (size_t i; (i = next++) < ms.size();) {
            const Member &m = ms[i];
            Bits in(data_, m.cend); in.seek(m.cdata * 8);             // the member's deflate data ends where its trailer begins
            o.n = o.prefix = 0;
            if (o.v.size() < (size_t)m.isize + 1024) o.v.resize((size_t)m.isize + 1024);
            const char *why = nullptr;
            const Stop st = decode_until<uint8_t>(in, t, o, SIZE_MAX, SIZE_MAX, why);
            if (st != MEMBER_END || o.n != m.isize || crc32_fast(0, o.v.data(), o.n) != m.crc) { bad = 1; continue; }
            memcpy(obuf_.data() + m.out_off, o.v.data(), o.n);
        }
    };
    {
        std::vector<std::thread> th;
        const size_t T = ms.size() < (size_t)threads_ ? ms.size() : (size_t)threads_;
        for (size_t k = 1; k < T; k++) th.emplace_back(work);
        work();
        for (auto &x : th) x.join();
    }
    if (bad) { err = "damaged BGZF member (deflate data, length or CRC)"; return false; }
    cur_bit_ = pos * 8; in_member_ = false; any_member_ = true; wlen_ = 0; crc_ = 0; member_out_ = 0;
    if (pos >= size_) done_ = true;
    return true;
}

bool ParallelGzReader::begin_member(std::string &err)
{
    size_t pos = cur_bit_ >> 3;                                          // byte aligned between members
    const size_t left = size_ - pos;
    if (left < 2 || data_[pos] != 0x1f || data_[pos + 1] != 0x8b) {
        if (any_member_) { done_ = true; return true; }                  // bytes after the last member that are not a member: ignored
        err = "not in gzip format"; return false;
    }
    if (left < 10) { err = "truncated gzip header"; return false; }
    if (data_[pos + 2] != 8) { err = "unknown gzip compression method"; return false; }
    const unsigned flg = data_[pos + 3];
    const uint8_t *p = data_ + pos + 10, *end = data_ + size_;
    if (flg & 4) {
        if (end - p < 2) { err = "truncated gzip header"; return false; }
        const size_t xlen = p[0] | ((size_t)p[1] << 8);
        p += 2;
        if ((size_t)(end - p) < xlen) { err = "truncated gzip header"; return false; }
        p += xlen;
    }
    for (unsigned bit = 8; bit <= 16; bit <<= 1)
        if (flg & bit) {
            const uint8_t *z = (const uint8_t *)memchr(p, 0, (size_t)(end - p));
            if (!z) { err = "truncated gzip header"; return false; }
            p = z + 1;
        }
    if (flg & 2) { if (end - p < 2) { err = "truncated gzip header"; return false; } p += 2; }
    cur_bit_ = (size_t)(p - data_) * 8;
    in_member_ = true; any_member_ = true; wlen_ = 0;      // crc_ / member_out_ are reset where the previous member is checked
    return true;
}

bool ParallelGzReader::fill(ByteBuf &obuf_, std::string &err)
{
    obuf_.clear();
    if (done_) return true;
    if (transparent_) {
        const size_t pos = cur_bit_ >> 3, n = size_ - pos < ((size_t)64 << 20) ? size_ - pos : ((size_t)64 << 20);
        obuf_.resize(n); memcpy(obuf_.data(), data_ + pos, n);
        cur_bit_ = (pos + n) * 8;
        if (pos + n == size_) done_ = true;
        return true;
    }
    if (!in_member_) {
        bool handled = false;
        if (!fill_bgzf(obuf_, err, handled)) return false;
        if (handled) return true;
        if (!begin_member(err)) return false;
        if (done_) return true;
    }

    static const bool gz_timing = getenv("MF_GZ_TIMING") != nullptr;   // phase times of every group on stderr (diagnostics)
    auto now = [] { return std::chrono::steady_clock::now(); };
    const auto t_a = now();
    // ---- speculative decode of one group of chunks
    const size_t base_byte = cur_bit_ >> 3;
    size_t G = (size_t)threads_;
    while (G > 1 && base_byte + (G - 1) * chunk_ >= size_) G--;
    auto nominal_bit = [&](size_t j) { return (base_byte + j * chunk_) * 8; };
    const size_t max_out = chunk_ * 32 > ((size_t)16 << 20) ? chunk_ * 32 : ((size_t)16 << 20);   // symbols per piece before it is cut short
    using Res = ChunkResult;
    if (!scratch_) scratch_ = new Scratch();
    std::vector<Res> &res = scratch_->res;
    if (res.size() < G) res.resize(G);
    for (size_t j = 0; j < G; j++) { res[j].valid = false; res[j].start = res[j].end = 0; res[j].stop = FAILED; res[j].sym.n = res[j].sym.prefix = 0; }
    Out<uint8_t> &first = scratch_->first; first.n = first.prefix = 0;
    const char *first_err = nullptr; Stop first_stop = FAILED; size_t first_end = 0;
    {
        std::vector<std::thread> th;
        for (size_t j = 1; j < G; j++)
            th.emplace_back([&, j] {
                Res &r = res[j];
                const size_t s = find_block(data_, size_, nominal_bit(j), nominal_bit(j + 1) < size_ * 8 ? nominal_bit(j + 1) : size_ * 8);
                if (s == SIZE_MAX) return;
                Bits in(data_, size_); in.seek(s);
                Tables t; const char *why = nullptr;
                if (r.sym.v.size() < chunk_ * 3 + 4096) r.sym.v.resize(chunk_ * 3 + 4096);
                r.start = s;
                r.stop = decode_until<uint16_t>(in, t, r.sym, nominal_bit(j + 1), max_out, why);
                r.end = in.bitpos();
                r.valid = r.stop != FAILED;
            });
        {   // chunk 0: the true position, the true window
            Bits in(data_, size_); in.seek(cur_bit_);
            Tables t;
            if (first.v.size() < wlen_ + chunk_ * 3 + 4096) first.v.resize(wlen_ + chunk_ * 3 + 4096);
            memcpy(first.v.data(), window_.data() + WINDOW - wlen_, wlen_);
            first.n = first.prefix = wlen_;
            first_stop = decode_until<uint8_t>(in, t, first, nominal_bit(1), max_out, first_err);
            first_end = in.bitpos();
        }
        for (auto &t : th) t.join();
    }
    if (first_stop == FAILED) { err = first_err ? first_err : "damaged deflate stream"; return false; }

    const auto t_b = now();
    // ---- link (serial, cheap): which pieces make up the output, and the window in front of each
    struct Piece { const uint8_t *bytes = nullptr; const uint16_t *syms = nullptr; size_t n = 0, out_off = 0; std::vector<uint8_t> window; };
    struct Check { size_t out_off; uint32_t crc, isize; };
    std::vector<Piece> pieces; std::vector<Check> checks;
    std::vector<Out<uint8_t>> fills;                                    // gap fills own their bytes
    fills.reserve(2 * G + 2);
    size_t total = 0;
    auto add_bytes = [&](const uint8_t *b, size_t n) {
        Piece p; p.bytes = b; p.n = n; p.out_off = total; pieces.push_back(std::move(p));
        slide_window(window_, wlen_, b, n); total += n;
    };
    // the member ended at cur_bit_: take its trailer, then look for another member
    auto finish_member = [&]() -> bool {
        size_t pos = (cur_bit_ + 7) >> 3;
        if (pos + 8 > size_) { err = "truncated gzip trailer"; return false; }
        uint32_t c, s; memcpy(&c, data_ + pos, 4); memcpy(&s, data_ + pos + 4, 4);
        checks.push_back(Check{total, c, s});
        cur_bit_ = (pos + 8) * 8; in_member_ = false;
        if (pos + 8 >= size_) { done_ = true; return true; }
        return begin_member(err);
    };
    add_bytes(first.v.data() + first.prefix, first.n - first.prefix);
    cur_bit_ = first_end;
    bool ok = true;
    if (first_stop == MEMBER_END) ok = finish_member();
    for (size_t j = 1; ok && j < G && !done_; j++) {
        Res &r = res[j];
        if (!r.valid || r.start < cur_bit_) { chunks_discarded++; continue; }
        if (r.start > cur_bit_) {                                        // decode across the gap with the true window
            fills.emplace_back();
            Out<uint8_t> &f = fills.back();
            f.v.resize(wlen_ + ((size_t)1 << 20));
            memcpy(f.v.data(), window_.data() + WINDOW - wlen_, wlen_);
            f.n = f.prefix = wlen_;
            Bits in(data_, size_); in.seek(cur_bit_);
            Tables t; const char *why = nullptr;
            const Stop st = decode_until<uint8_t>(in, t, f, r.start, max_out, why);
            if (st == FAILED) { err = why ? why : "damaged deflate stream"; ok = false; break; }
            gap_fill_bytes += f.n - f.prefix;
            add_bytes(f.v.data() + f.prefix, f.n - f.prefix);
            cur_bit_ = in.bitpos();
            if (st == MEMBER_END) { ok = finish_member(); if (!ok || done_) break; }
            if (r.start != cur_bit_) {                                   // overshot (the candidate was not a block start), a new member began, or the fill hit its size bound
                if (r.start < cur_bit_) { chunks_discarded++; continue; }
                if (st == AT_BOUNDARY) break;                            // size bound: hand out what there is, the next group starts here
                j--; continue;                                           // a member ended in the gap: fill again from the next member's first block
            }
        }
        // r.start == cur_bit_: by induction a true block boundary -- accept the chunk
        Piece p; p.syms = r.sym.v.data(); p.n = r.sym.n; p.out_off = total;
        p.window.assign(window_.begin(), window_.end());
        // window behind this chunk: resolve its tail only
        {
            const size_t tail = p.n < WINDOW ? p.n : WINDOW;
            std::vector<uint8_t> tb(tail);
            const uint16_t *s = p.syms + (p.n - tail);
            for (size_t i = 0; i < tail; i++) tb[i] = (s[i] & MARK) ? p.window[s[i] & 0x7FFF] : (uint8_t)s[i];
            // a marker inside the tail may point into the part of the old window that slides out: resolved above, before sliding
            slide_window(window_, wlen_, tb.data(), tail);
            if (p.n > tail) wlen_ = WINDOW;
        }
        total += p.n;
        pieces.push_back(std::move(p));
        chunks_linked++;
        cur_bit_ = r.end;
        if (r.stop == MEMBER_END) ok = finish_member();
    }
    if (!ok) return false;

    const auto t_c = now();
    // ---- assemble: copy bytes, replace markers (parallel over pieces)
    obuf_.resize(total);
    {
        std::vector<std::thread> th;
        std::atomic<size_t> next{0};
        auto work = [&] {
            for (size_t i; (i = next++) < pieces.size();) {
                const Piece &p = pieces[i];
                uint8_t *d = obuf_.data() + p.out_off;
                if (p.bytes) { memcpy(d, p.bytes, p.n); continue; }
                const uint8_t *w = p.window.data();
                resolve_symbols(p.syms, p.n, w, d);
            }
        };
        const size_t T = pieces.size() < (size_t)threads_ ? pieces.size() : (size_t)threads_;
        for (size_t t = 1; t < T; t++) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
    }
    const auto t_d = now();
    // ---- member checks
    size_t seg = 0;
    for (const Check &c : checks) {
        crc_ = crc32_parallel(crc_, obuf_.data() + seg, c.out_off - seg, threads_);
        member_out_ += c.out_off - seg;
        if (crc_ != c.crc) { err = "incorrect data check"; return false; }
        if ((uint32_t)member_out_ != c.isize) { err = "incorrect length check"; return false; }
        crc_ = 0; member_out_ = 0; seg = c.out_off;
    }
    crc_ = crc32_parallel(crc_, obuf_.data() + seg, total - seg, threads_);
    member_out_ += total - seg;
    if (gz_timing) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[mf gz] group of %zu chunks -> %.1f MB: decode %.1f ms, link %.1f ms, assemble %.1f ms, crc %.1f ms\n", G, total / 1e6,
                ms(t_a, t_b), ms(t_b, t_c), ms(t_c, t_d), ms(t_d, now()));
    }
    return true;
}

long ParallelGzReader::read(uint8_t *out, size_t cap, std::string &err)
{
    size_t got = 0;
    while (got < cap) {
        if (opos_ == obuf_.size()) {
            if (!pre_running_) { if (done_) break; start_prefetch(); }
            pre_.join(); pre_running_ = false;
            if (!pre_ok_) { err = pre_err_; return -1; }
            obuf_.swap(nbuf_); opos_ = 0;
            if (!done_) start_prefetch();                    // the following group is decoded while this one is handed out
            continue;
        }
        size_t n = obuf_.size() - opos_;
        if (n > cap - got) n = cap - got;
        memcpy(out + got, obuf_.data() + opos_, n);
        opos_ += n; got += n;
    }
    return (long)got;
}

} // namespace mf
