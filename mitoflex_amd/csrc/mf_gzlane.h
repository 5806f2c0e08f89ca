// One lane's part of the device DEFLATE decoder (mf_gzdev.hip): the Huffman walk over a span of a block.
//
// Inside a deflate block all codes come from ONE pair of tables, and a Huffman stream decoded from a wrong bit offset falls
// into step with the true one after a few dozen codes.  So the 64 lanes of a wavefront walk 64 consecutive spans of the block
// at once, each from its own bit offset: lane i starts where lane i - 1 ended, which it first has to guess (the nominal border
// of its span) and then learns -- a lane whose start moves walks its span again, until every lane starts exactly where its
// predecessor stopped (mf_gzdev.hip, decode_block; the scheme is Weissenberger & Schmidt's self-synchronising Huffman decoding,
// ICPP 2018, with deflate's length / distance pairs and extra bits as part of a code).  By induction from lane 0, whose start is
// the true one, every lane of the chain then holds true codes.
//
// This header is the lane's walk alone -- table entry formats, bit buffer, the loop -- written so that it compiles for the host
// as well: tools/gzlane_model.cpp runs the same code over 64 emulated lanes against zlib (test infrastructure for the scheme;
// the product is the kernel).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GZL_HD __host__ __device__ __forceinline__
#else
#define GZL_HD inline
#endif

namespace mf {
namespace gzl {

constexpr int LIT_BITS = 10, DIST_BITS = 9, PRE_BITS = 7;
constexpr uint32_t LIT_SIZE = 1u << LIT_BITS, DIST_SIZE = 1u << DIST_BITS, PRE_SIZE = 1u << PRE_BITS;

// literal/length entry: bits 0-7 bits to drop; bit 31 set = not a literal (one signed compare in the walk).
//   literal:  bits 8-15 first byte, 16-23 second byte, bit 24 = two literals, bits 25-28 = bits of the FIRST literal's code then
//   other:    bits 29-30 kind; a length: bits 24-26 extra bits, bits 8-15 base - 3
constexpr uint32_t E_OTHER = 1u << 31, E_DOUBLE = 1u << 24;
constexpr uint32_t K_MASK = 3u << 29, K_LENGTH = 0u << 29, K_EOB = 1u << 29, K_LONG = 2u << 29, K_INVALID = 3u << 29;
// distance entry: bits 0-7 bits to drop; bit 31 set = long code (bit 30 clear) or invalid (bit 30 set); bits 24-27 extra bits, 8-22 base - 1
constexpr uint32_t D_INVALID = 1u << 30;
// list entry of a match: bit 31 | (length - 3) | (distance - 1) << 9; of literals: the table entry itself

// canonical code per length: first code, number of codes, offset of the length's first symbol in the sorted symbol list
struct Canon { uint16_t first[16], cnt[16], off[16]; };

GZL_HD uint32_t lit_entry(uint32_t s)
{
    if (s < 256) return s << 8;
    if (s == 256) return E_OTHER | K_EOB;
    if (s < 286) {
        const uint32_t k = s - 257;
        uint32_t extra = 0, base = 3 + k;
        if (k == 28) base = 258;
        else if (k >= 8) { extra = (k >> 2) - 1; base = 3 + ((4 + (k & 3)) << extra); }
        return E_OTHER | K_LENGTH | (extra << 24) | ((base - 3) << 8);      // the base is stored less 3 (what the list entry holds)
    }
    return E_OTHER | K_INVALID;
}
GZL_HD uint32_t dist_entry(uint32_t d)
{
    if (d >= 30) return E_OTHER | D_INVALID;
    uint32_t extra = 0, base = 1 + d;
    if (d >= 4) { extra = (d >> 1) - 1; base = 1 + ((2 + (d & 1)) << extra); }
    return (extra << 24) | ((base - 1) << 8);                                     // the base is stored less 1
}

GZL_HD uint32_t brev32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    return (x >> 16) | (x << 16);
#endif
}

// what a span's walk found
enum : uint32_t { SP_EOB = 1, SP_ERR = 2, SP_FULL = 4 };
struct Span {
    uint32_t end;           // bit (counted from the reader's origin) of the first code NOT taken: the first one at or behind `stop`; behind the end-of-block code with SP_EOB
    uint32_t n_code;        // list entries: one per literal entry (one or two literals), one per length / distance pair
    uint32_t n_sym;         // bytes of output they stand for
    uint32_t flags;         // SP_EOB: the block ended inside the span; SP_ERR: not deflate data (from a guessed start: means nothing); SP_FULL: max_codes reached
};

// Walk the codes that start in [start, stop).  lit / dist: the block's first-level tables (LIT_SIZE / DIST_SIZE entries), cl / cd
// and sorted_*: the canonical description for codes longer than the tables' index.  in(w): dword w of the stream, counted from the
// reader's origin (must be callable a few dwords past the span).  out.put(e): list entry number n_code (the caller flushes `out`).
template <class In, class Out>
GZL_HD Span walk_span(const uint32_t *lit, const uint32_t *dist, const Canon &cl, const Canon &cd, const uint16_t *sorted_lit,
                      const uint16_t *sorted_dist, const In &in, uint32_t start, uint32_t stop, uint32_t max_codes, Out &out)
{
    uint32_t w = start >> 5;
    uint64_t bb = (uint64_t)(in(w) >> (start & 31)); uint32_t bc = 32 - (start & 31);
    uint32_t nd = in(w + 1);                    // dword w - 1 from here on: read ahead, not yet in the buffer
    w += 2;
    Span r; r.end = start; r.n_code = 0; r.n_sym = 0; r.flags = 0;
    for (;;) {
        const uint32_t pos = (w - 1) * 32 - bc;
        if (pos >= stop) { r.end = pos; break; }
        if (r.n_code >= max_codes) { r.end = pos; r.flags |= SP_FULL; break; }
        if (bc <= 32) { bb |= (uint64_t)nd << bc; bc += 32; nd = in(w); w++; }          // 33 .. 64 valid bits: a length code, its extra bits: 20
        uint32_t e = lit[(uint32_t)bb & (LIT_SIZE - 1)];
        if ((int32_t)e >= 0) {                  // one or two literals
            uint32_t l = e & 255u;
            // A span ends at the FIRST code boundary at or behind `stop`, however the codes in front of it were taken two at a time:
            // that makes the end a function of the stream alone -- a lane that walked its span from a guessed start and fell into
            // step ends exactly where it will end when it walks from the true start.
            if ((e & E_DOUBLE) && pos + ((e >> 25) & 15u) >= stop) { l = (e >> 25) & 15u; e &= 0xFF00u; }
            bb >>= l; bc -= l;
            r.n_sym += 1 + ((e >> 24) & 1u);
            out.put(e);
            r.n_code++;
            continue;
        }
        if ((e & K_MASK) == K_LONG) {           // longer than the table's index: canonical decode
            const uint32_t rev = brev32((uint32_t)bb);
            uint32_t s = 0xFFFFu, len = 0;
            for (uint32_t l = LIT_BITS + 1; l <= 15; l++) {
                const uint32_t idx = (rev >> (32 - l)) - cl.first[l];
                if (idx < cl.cnt[l]) { len = l; s = sorted_lit[cl.off[l] + idx]; break; }
            }
            if (s == 0xFFFFu) { r.end = pos; r.flags |= SP_ERR; break; }
            bb >>= len; bc -= len;
            e = lit_entry(s);
            if ((int32_t)e >= 0) { r.n_sym++; out.put(e); r.n_code++; continue; }
        } else { const uint32_t l = e & 255u; bb >>= l; bc -= l; }
        const uint32_t kind = e & K_MASK;
        if (kind == K_EOB) { r.end = (w - 1) * 32 - bc; r.flags |= SP_EOB; break; }
        if (kind != K_LENGTH) { r.end = pos; r.flags |= SP_ERR; break; }
        const uint32_t xl = (e >> 24) & 7u;
        const uint32_t lenm3 = ((e >> 8) & 0xFFu) + ((uint32_t)bb & ((1u << xl) - 1u));
        bb >>= xl; bc -= xl;
        if (bc <= 32) { bb |= (uint64_t)nd << bc; bc += 32; nd = in(w); w++; }          // a distance code and its extra bits: 28
        uint32_t d = dist[(uint32_t)bb & (DIST_SIZE - 1)];
        if ((int32_t)d < 0) {
            if (d & D_INVALID) { r.end = pos; r.flags |= SP_ERR; break; }
            const uint32_t rev = brev32((uint32_t)bb);
            uint32_t s = 0xFFFFu, dl = 0;
            for (uint32_t l = DIST_BITS + 1; l <= 15; l++) {
                const uint32_t idx = (rev >> (32 - l)) - cd.first[l];
                if (idx < cd.cnt[l]) { dl = l; s = sorted_dist[cd.off[l] + idx]; break; }
            }
            if (s == 0xFFFFu) { r.end = pos; r.flags |= SP_ERR; break; }
            bb >>= dl; bc -= dl;
            d = dist_entry(s);
            if ((int32_t)d < 0) { r.end = pos; r.flags |= SP_ERR; break; }
        } else { const uint32_t l = d & 255u; bb >>= l; bc -= l; }
        const uint32_t xd = (d >> 24) & 15u;
        const uint32_t distm1 = ((d >> 8) & 0x7FFFu) + ((uint32_t)bb & ((1u << xd) - 1u));
        bb >>= xd; bc -= xd;
        r.n_sym += lenm3 + 3;
        out.put(0x80000000u | lenm3 | (distm1 << 9));
        r.n_code++;
    }
    return r;
}

} // namespace gzl
} // namespace mf
