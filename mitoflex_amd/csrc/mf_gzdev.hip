// DEFLATE decoding on gfx950 (see mf_gzdev.h for the scheme).  One wavefront per chunk of the compressed stream.
//
// Inside a wavefront the work is split by what is serial and what is not:
//   * the Huffman walk -- table lookup, drop the code's bits, next lookup -- is one dependent chain: up to 63 codes ("a round")
//     are walked and leave table entries / (length, distance) pairs in an LDS list.  It is hand-written assembly and scalar
//     where it is serial (bit buffer, bit count, entry fields).  First-level tables of 2^10 (literal/length) and 2^9 (distance)
//     entries are built in LDS and then held in 24 VECTOR REGISTERS for the block (a lookup is a v_readlane in GPR index mode);
//     an entry holds TWO literals where both codes fit into the index; codes longer than the index are decoded canonically
//     (first code / count per length), so there are no sub-tables and the footprint is fixed;
//   * everything else is done by all 64 lanes: the search for a block header (one bit offset per lane), building the decode
//     tables from the code lengths (ranks by ballot, table fill by symbol), and turning a round's list into output -- a
//     prefix sum gives every list entry its place, then every OUTPUT POSITION of the round finds its entry (binary search in
//     LDS) and fetches its symbol: a literal, a symbol written in an earlier round (global memory, all loads of the round in
//     flight together), a marker, or -- for a match that reaches into the round itself -- a reference that is chased
//     through the LDS staging buffer afterwards.  The round leaves as coalesced 16-bit stores.
// The kernel is bound by one wavefront's dependent chain (about 290 cycles per code, the all-lane expansion a third on top)
// times the wavefronts a CU holds: 9.9 KB of LDS and 128 VGPRs per wavefront, sixteen chunks in flight per CU.
#include "mf_gzdev.h"
#include "mf_gzlane.h"
#include <algorithm>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

namespace mf {
namespace {

using namespace gzl;                     // table sizes, entry formats, Canon, the lane's walk (mf_gzlane.h)
constexpr uint32_t ROUND = 64;           // list entries per round
constexpr uint32_t STG = 1024;           // output symbols per round (staging buffer)
constexpr uint32_t RING = 512;           // dwords of input held in LDS

constexpr uint32_t PENDING = 0x4000;     // staging value: reference to another staging slot (bit 15 clear, bit 14 set)

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
// the first-level tables of the block being walked: entry i of a table is lane i & 63 of element i >> 6 (in the walk's assembly: v[104:119], v[120:127])
struct TabRegs { u32x16 lit; u32x8 dist; };
struct Lds {                             // the walk's inline assembly relies on ring at LDS offset 0
    uint32_t ring[RING];
    // A block's first-level tables are built here by all lanes and then live in vector registers (TabRegs) while its codes are
    // walked; the rounds' list and staging buffers take the same bytes afterwards: 9.9 KB a wavefront, sixteen wavefronts a CU.
    union {
        struct { uint32_t lit[LIT_SIZE]; uint32_t dist[DIST_SIZE]; uint32_t pre[PRE_SIZE]; uint8_t lens[328]; uint8_t plens[24]; };
        struct {
            uint32_t sym[ROUND + 2 * 64];    // the round's list; behind it two slots per lane for the stores of the lanes that are not the writer
            uint16_t soff[ROUND];
            uint16_t stg[STG];
        };
    };
    uint16_t sorted_lit[288], sorted_dist[32];
    Canon clit, cdist;
};

// LSB-first bit reader.  The input reaches the wavefront through a ring of RING dwords in LDS that all 64 lanes top up
// together (one 16-byte load per lane, 1 KiB a time); the walk itself only ever reads LDS.  Every value in here is
// wave-uniform: all lanes execute the reader with the same state, so its branches are scalar branches.
struct BitRd {
    const uint4 *base; uint64_t vmax;        // vectors 0 .. vmax of the input are readable
    uint64_t vmask;                          // the input lives in a ring of vmask + 1 vectors (all ones: no ring)
    uint64_t origin_v;                       // rd_dw and ring_hi count from this vector of the input (a chunk reads a few MiB; the file may have more than 2^32 dwords)
    uint32_t *ring;                          // LDS; holds the dwords [ring_hi - RING, ring_hi)
    uint32_t rd_dw, ring_hi;                 // rd_dw: next dword to enter the bit buffer
    uint32_t nd;                             // ring[rd_dw], read ahead
    uint64_t bb; uint32_t bc;
    __device__ __forceinline__ uint32_t next_dword()
    {
        const uint32_t d = nd;
        rd_dw++;
        nd = ring[rd_dw & (RING - 1)];
        return d;
    }
    __device__ __forceinline__ void refill() { if (bc <= 30) { bb |= (uint64_t)next_dword() << bc; bc += 32; } }   // afterwards 31 <= bc <= 62
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return (uint32_t)bb & ((1u << n) - 1); }           // n <= 31
    __device__ __forceinline__ void drop(uint32_t n) { bb >>= n; bc -= n; }
    __device__ __forceinline__ uint64_t bitpos() const { return origin_v * 128 + (uint64_t)rd_dw * 32 - bc; }
    __device__ __forceinline__ void load_half(uint32_t lane)      // 256 more dwords behind ring_hi
    {
        const uint64_t v = origin_v + ring_hi / 4 + lane;
        const uint4 x = base[(v < vmax ? v : vmax) & vmask];
        *reinterpret_cast<uint4 *>(&ring[(ring_hi + 4 * lane) & (RING - 1)]) = x;
        ring_hi += RING / 2;
    }
    // at least RING / 2 dwords ahead of the reader afterwards: a block header needs ~150, a round of the walk at most 96
    __device__ __forceinline__ void top_up(uint32_t lane)
    {
        if (ring_hi - rd_dw < RING / 2) { load_half(lane); __syncthreads(); nd = ring[rd_dw & (RING - 1)]; }
    }
    __device__ __forceinline__ void seek(uint64_t bit, uint32_t lane)
    {
        const uint32_t d = (uint32_t)((bit - origin_v * 128) >> 5);
        if (!(d + RING / 2 <= ring_hi && d + RING >= ring_hi)) {
            __syncthreads();
            ring_hi = d & ~3u;
            load_half(lane); load_half(lane);
            __syncthreads();
        }
        rd_dw = d; bb = 0; bc = 0;
        nd = ring[rd_dw & (RING - 1)];
        refill();
        drop((uint32_t)bit & 31);
        refill();
    }
};

// Canonical Huffman decode table from code lengths, by the whole wavefront.  KIND 0: precode, 1: literal/length, 2: distance.
// Returns the Kraft sum in units of 2^-15 (32768 = complete; more = over-subscribed, nothing is built); *n_codes = codes in use.
template <int BITS, int KIND>
__device__ __forceinline__ uint32_t build_table(const uint8_t *lens, uint32_t n, uint32_t *tab, uint16_t *sorted, Canon &cn, uint32_t lane, uint32_t *n_codes)
{
    constexpr int R = KIND == 1 ? 5 : 1;
    constexpr uint32_t SIZE = 1u << BITS;
    uint32_t myl[R], rank[R], c[16];
#pragma unroll
    for (int l = 0; l < 16; l++) c[l] = 0;
    const uint64_t below = (1ull << lane) - 1;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t s = lane + 64 * r;
        myl[r] = s < n ? lens[s] : 0;
        rank[r] = 0;
#pragma unroll
        for (int l = 1; l < 16; l++) {
            const uint64_t m = __ballot(myl[r] == (uint32_t)l);
            if (myl[r] == (uint32_t)l) rank[r] = c[l] + __popcll(m & below);
            c[l] += __popcll(m);
        }
    }
    uint32_t code = 0, kraft = 0, o = 0, used = 0, fst[16], ofs[16];
    fst[0] = ofs[0] = 0;
#pragma unroll
    for (int l = 1; l < 16; l++) {
        code = (code + c[l - 1]) << 1; fst[l] = code; ofs[l] = o;
        kraft += c[l] << (15 - l); o += c[l]; used += c[l];
    }
    if (n_codes) *n_codes = used;
    if (kraft > 32768) return kraft;
    if (lane < 16) {
        uint32_t f = 0, cc = 0, oo = 0;
#pragma unroll
        for (int l = 1; l < 16; l++) if (lane == (uint32_t)l) { f = fst[l]; cc = c[l]; oo = ofs[l]; }
        cn.first[lane] = (uint16_t)f; cn.cnt[lane] = (uint16_t)cc; cn.off[lane] = (uint16_t)oo;
    }
    const uint32_t invalid = KIND == 1 ? (E_OTHER | K_INVALID) : (KIND == 2 ? (E_OTHER | D_INVALID) : 0u);
    for (uint32_t i = lane; i < SIZE; i += 64) tab[i] = invalid;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t l = myl[r], s = lane + 64 * r;
        if (l) {
            const uint32_t cd = (uint32_t)cn.first[l] + rank[r];
            if (KIND != 0) sorted[cn.off[l] + rank[r]] = (uint16_t)s;
            const uint32_t rev = __brev(cd) >> (32 - l);
            if (l <= (uint32_t)BITS) {
                const uint32_t e = (KIND == 1 ? lit_entry(s) : (KIND == 2 ? dist_entry(s) : (s << 8))) | l;
                for (uint32_t i = rev; i < SIZE; i += 1u << l) tab[i] = e;
            } else tab[rev & (SIZE - 1)] = KIND == 1 ? (E_OTHER | K_LONG) : E_OTHER;
        }
    }
    __syncthreads();
    return kraft;
}

// two literals per entry where both codes fit into the index (the entry of the second code is read from the single table)
__device__ __forceinline__ void pair_literals(uint32_t *lit, uint32_t lane)
{
    uint32_t ne[LIT_SIZE / 64];
#pragma unroll
    for (uint32_t j = 0; j < LIT_SIZE / 64; j++) {
        const uint32_t i = lane + 64 * j, e1 = lit[i];
        ne[j] = e1;
        if (!(e1 & E_OTHER)) {
            const uint32_t l1 = e1 & 255;
            if (l1 < (uint32_t)LIT_BITS) {
                const uint32_t e2 = lit[i >> l1];
                if (!(e2 & E_OTHER) && l1 + (e2 & 255) <= (uint32_t)LIT_BITS)
                    ne[j] = (l1 + (e2 & 255)) | E_DOUBLE | (l1 << 25) | (e1 & 0xFF00u) | ((e2 & 0xFF00u) << 8);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < LIT_SIZE / 64; j++) lit[lane + 64 * j] = ne[j];
    __syncthreads();
}

__device__ const uint8_t PRE_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Dynamic block header behind the three type bits: the code lengths of both alphabets into L.lens (uniform code; only the
// LDS stores are left to lane 0).  strict: only what a compressor emits (a complete precode).
template <class LDS>
__device__ __forceinline__ bool read_code_lengths(LDS &L, BitRd &rd, bool strict, uint32_t lane, uint32_t &hlit, uint32_t &hdist)
{
    rd.top_up(lane);
    rd.refill();
    hlit = rd.peek(5) + 257; rd.drop(5);
    hdist = rd.peek(5) + 1; rd.drop(5);
    const uint32_t hclen = rd.peek(4) + 4; rd.drop(4);
    if (hlit > 286 || hdist > 30) return false;
    if (lane < 24) L.plens[lane] = 0;
    __syncthreads();
    for (uint32_t i = 0; i < hclen; i++) { rd.refill(); const uint32_t v = rd.peek(3); rd.drop(3); if (lane == 0) L.plens[PRE_ORDER[i]] = (uint8_t)v; }
    __syncthreads();
    const uint32_t kraft = build_table<PRE_BITS, 0>(L.plens, 19, L.pre, nullptr, L.cdist, lane, nullptr);
    if (kraft > 32768 || (strict && kraft != 32768)) return false;
    const uint32_t total = hlit + hdist;
    uint32_t i = 0, prev = 0;
    // Running Kraft sums (units of 2^-15) of the two alphabets.  A speculative header is dropped the moment one of them passes 1:
    // nearly every false candidate -- there are hundreds per chunk -- dies within its first few dozen code lengths instead of
    // being parsed to the end and having its tables built.
    uint32_t kl = 0, kd = 0;
    while (i < total) {
        rd.refill();
        const uint32_t e = L.pre[(uint32_t)rd.bb & (PRE_SIZE - 1)];
        if ((e & 255) == 0) return false;
        rd.drop(e & 255);
        const uint32_t s = e >> 8;
        uint32_t rep = 1, val = s;
        if (s >= 16) {
            val = 0;
            if (s == 16) { if (i == 0) return false; val = prev; rep = 3 + rd.peek(2); rd.drop(2); }
            else if (s == 17) { rep = 3 + rd.peek(3); rd.drop(3); }
            else { rep = 11 + rd.peek(7); rd.drop(7); }
            if (i + rep > total) return false;
        }
        if (val) {
            const uint32_t in_lit = i >= hlit ? 0 : (i + rep <= hlit ? rep : hlit - i);
            kl += in_lit * (32768u >> val); kd += (rep - in_lit) * (32768u >> val);
            if (strict && (kl > 32768u || kd > 32768u)) return false;
        }
        if (lane < rep) L.lens[i + lane] = (uint8_t)val;
        if (lane + 64 < rep) L.lens[i + lane + 64] = (uint8_t)val;
        if (lane + 128 < rep) L.lens[i + lane + 128] = (uint8_t)val;
        i += rep; prev = val;
    }
    if (strict && kl != 32768u) return false;
    __syncthreads();
    return L.lens[256] != 0;
}

// both decode tables from L.lens[0 .. hlit + hdist)
template <class LDS>
__device__ __forceinline__ bool build_tables(LDS &L, uint32_t hlit, uint32_t hdist, bool strict, uint32_t lane)
{
    uint32_t kraft = build_table<LIT_BITS, 1>(L.lens, hlit, L.lit, L.sorted_lit, L.clit, lane, nullptr);
    if (kraft > 32768 || (strict && kraft != 32768)) return false;
    uint32_t nd = 0;
    kraft = build_table<DIST_BITS, 2>(L.lens + hlit, hdist, L.dist, L.sorted_dist, L.cdist, lane, &nd);
    if (kraft > 32768 || (strict && nd > 1 && kraft != 32768)) return false;
    pair_literals(L.lit, lane);
    return true;
}

// Candidates for a block start: the bit offsets in [pos, to_bit) that look like the header of a non-final dynamic block --
// type bits, symbol counts, Kraft sum of the precode -- 64 offsets a step, one per lane.  Returns the next one (~0: none) and
// keeps the rest of the current step in `mask`.
// The search looks at SEARCH_W windows of 64 bit offsets at a time (one offset per lane and window): the loads of all of them are in
// flight together -- a step of the search is a memory round trip, and with one window a step that was what it cost.
constexpr int SEARCH_W = 4;
// m0: candidates among the offsets b0 + lane of the current window; m1 .. m7: the windows behind it, `left` of them still to come.
// (Named members and a shift from one to the next, not an array picked by index: the masks are wave-uniform and stay in scalar registers.)
struct Search { uint64_t b0; uint64_t m0, m1, m2, m3, m4, m5, m6, m7; uint32_t left; };
__device__ __forceinline__ void search_init(Search &S, uint64_t first_bit) { S.b0 = first_bit - 64; S.m0 = S.m1 = S.m2 = S.m3 = S.m4 = S.m5 = S.m6 = S.m7 = 0; S.left = 0; }
__device__ __forceinline__ uint64_t next_candidate(Search &S, const uint32_t *words, uint64_t wmask, uint64_t to_bit, uint64_t size_bits, uint32_t lane)
{
    static_assert(SEARCH_W >= 1 && SEARCH_W <= 8, "the members of Search");
    for (;;) {
        if (S.m0) {
            const int idx = __ffsll((unsigned long long)S.m0) - 1;
            S.m0 &= S.m0 - 1;
            return S.b0 + (uint64_t)idx;
        }
        if (S.left) { S.m0 = S.m1; S.m1 = S.m2; S.m2 = S.m3; S.m3 = S.m4; S.m4 = S.m5; S.m5 = S.m6; S.m6 = S.m7; S.m7 = 0; S.b0 += 64; S.left--; continue; }
        S.b0 += 64;
        if (S.b0 >= to_bit) return ~0ull;
        uint32_t w[SEARCH_W][4]; bool in[SEARCH_W]; uint64_t k8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < SEARCH_W; j++) {
            const uint64_t bit = S.b0 + (uint64_t)(64 * j) + lane;
            in[j] = bit < to_bit && bit + 128 <= size_bits;
            const uint64_t wi = in[j] ? bit >> 5 : 0;
#pragma unroll
            for (int q = 0; q < 4; q++) w[j][q] = words[(wi + q) & wmask];
        }
#pragma unroll
        for (int j = 0; j < SEARCH_W; j++) {
            const uint32_t s = (uint32_t)(S.b0 + (uint64_t)(64 * j) + lane) & 31;
            const uint32_t h0 = __funnelshift_r(w[j][0], w[j][1], s), h1 = __funnelshift_r(w[j][1], w[j][2], s), h2 = __funnelshift_r(w[j][2], w[j][3], s);
            bool ok = in[j] && (h0 & 7u) == 4u && ((h0 >> 3) & 31u) <= 29u && ((h0 >> 8) & 31u) <= 29u;      // BFINAL = 0, BTYPE = 10b
            if (ok) {
                const uint32_t hclen = ((h0 >> 13) & 15u) + 4;
                const uint64_t a = h0 | ((uint64_t)h1 << 32), b = h1 | ((uint64_t)h2 << 32);
                uint32_t k = 0;
#pragma unroll
                for (uint32_t i = 0; i < 19; i++) {
                    const uint32_t pos = 17 + 3 * i;
                    const uint32_t l = (uint32_t)(pos < 32 ? (a >> pos) : (b >> (pos - 32))) & 7u;
                    if (i < hclen && l) k += 128u >> l;
                }
                ok = k == 128u;
            }
            k8[j] = __ballot(ok);
        }
        S.m0 = k8[0]; S.m1 = k8[1]; S.m2 = k8[2]; S.m3 = k8[3]; S.m4 = k8[4]; S.m5 = k8[5]; S.m6 = k8[6]; S.m7 = k8[7];
        S.left = SEARCH_W - 1;
    }
}

// ---- candidates are vetted 64 at a time, one per lane
// What the window test above lets through is still mostly noise: about one bit offset in five hundred, a thousand in a chunk that holds
// no block start at all -- and sending each through the wavefront's header reader (seek, precode table, code lengths until a Kraft sum
// runs over) cost ~20 k cycles apiece: 8 M cycles a chunk, more than half of what a 64 KiB chunk of a small file takes, 23 M in the
// chunks that find nothing.  So the next 64 candidates are collected first and every LANE reads the header of its own: the 19 precode
// lengths into one register, the code lengths decoded with the canonical rule bit by bit (no table), running Kraft sums of both
// alphabets -- the rules of read_code_lengths / build_tables for a strict header, nothing built.  The first lane that comes through
// gives the wavefront its start (the header is then read again by all lanes, which builds the tables); the others stay in the batch
// in case the data behind it does not decode.
__device__ __forceinline__ bool vet_header(const uint32_t *words, uint64_t wmask, uint64_t max_dw, uint64_t c, uint64_t size_bits)
{
    uint64_t w = c >> 5;
    auto ld = [&](uint64_t i) -> uint32_t { i = i < max_dw ? i : max_dw; return words[i & wmask]; };
    const uint32_t sh = (uint32_t)c & 31u;
    uint64_t bb = ((uint64_t)ld(w) | ((uint64_t)ld(w + 1) << 32)) >> sh; uint32_t bc = 64 - sh;
    w += 2;
    uint32_t used = 0;
    auto need = [&](uint32_t n) { if (bc < n) { bb |= (uint64_t)ld(w) << bc; bc += 32; w++; } };        // n <= 32
    auto take = [&](uint32_t n) -> uint32_t { const uint32_t v = (uint32_t)bb & ((1u << n) - 1u); bb >>= n; bc -= n; used += n; return v; };
    need(17);
    (void)take(3);
    const uint32_t hlit = take(5) + 257, hdist = take(5) + 1, hclen = take(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    // precode: length of symbol s in bits 3 s .. 3 s + 2 of pl
    uint64_t pl = 0;
    constexpr uint64_t ORDER_LO = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 | 10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
    constexpr uint64_t ORDER_HI = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
#pragma unroll
    for (uint32_t i = 0; i < 19; i++) {
        if (i < hclen) {
            need(3);
            const uint32_t sym = (uint32_t)((i < 12 ? ORDER_LO >> (5 * i) : ORDER_HI >> (5 * (i - 12))) & 31u);
            pl |= (uint64_t)take(3) << (3 * sym);
        }
    }
    // codes per length (5 bits each, length l in bits 5 (l - 1) ..) and the symbols sorted by (length, symbol), 5 bits each
    uint64_t cn = 0, sl_lo = 0, sl_hi = 0; uint32_t k = 0, kraft = 0;
    for (uint32_t l = 1; l <= 7; l++) {
        uint32_t c_l = 0;
#pragma unroll
        for (uint32_t sy = 0; sy < 19; sy++)
            if (((uint32_t)(pl >> (3 * sy)) & 7u) == l) {
                if (k < 12) sl_lo |= (uint64_t)sy << (5 * k); else sl_hi |= (uint64_t)sy << (5 * (k - 12));
                k++; c_l++;
            }
        cn |= (uint64_t)c_l << (5 * (l - 1));
        kraft += c_l << (7 - l);
    }
    if (kraft != 128u) return false;
    const uint32_t total = hlit + hdist;
    uint32_t i = 0, prev = 0, kl = 0, kd = 0, nd = 0; bool has_eob = false;
    while (i < total) {
        need(16);
        // one precode symbol, bit by bit (codes are packed most significant bit first)
        uint32_t code = 0, first = 0, index = 0, sym = 0xFFu;
        for (uint32_t l = 1; l <= 7; l++) {
            code |= (uint32_t)bb & 1u; bb >>= 1; bc--; used++;
            const uint32_t cnt = (uint32_t)(cn >> (5 * (l - 1))) & 31u;
            if (code - first < cnt) { const uint32_t at = index + (code - first); sym = (uint32_t)((at < 12 ? sl_lo >> (5 * at) : sl_hi >> (5 * (at - 12))) & 31u); break; }
            index += cnt; first = (first + cnt) << 1; code <<= 1;
        }
        if (sym == 0xFFu) return false;
        uint32_t rep = 1, val = sym;
        if (sym >= 16) {
            val = 0;
            if (sym == 16) { if (i == 0) return false; val = prev; rep = 3 + take(2); }
            else if (sym == 17) rep = 3 + take(3);
            else rep = 11 + take(7);
            if (i + rep > total) return false;
        }
        if (val) {
            const uint32_t in_lit = i >= hlit ? 0 : (i + rep <= hlit ? rep : hlit - i);
            kl += in_lit * (32768u >> val); kd += (rep - in_lit) * (32768u >> val); nd += rep - in_lit;
            if (kl > 32768u || kd > 32768u) return false;
            if (i <= 256 && 256 < i + rep) has_eob = true;
        }
        i += rep; prev = val;
    }
    if (kl != 32768u || !has_eob) return false;
    if (nd > 1 && kd != 32768u) return false;
    return c + used <= size_bits;
}
// V: a batch of vetted candidates in hand
struct Vetted { uint64_t cand; uint64_t vmask; bool dry; };      // cand: this lane's candidate of the batch; vmask: lanes whose candidate came through and has not been handed out

// ---- round 5: the same search with the lanes kept busy.  next_candidate gives every lane ONE bit offset of a window and lets all 64 run the
// 19-step Kraft sum of the precode whenever any of them got past the type bits -- one lane in nine does, so eight ninths of that work was
// idle lanes: 35 cycles a bit offset, 6 M of a 256 KiB chunk's 33 M cycles on a gzip -6 file (profiles/r04/c_gzdev_check_phases.log).  Now the
// survivors of the cheap test (type bits, symbol counts: a dozen instructions a window) are COMPACTED into a list in LDS, and the Kraft sum
// runs on 64 of them at a time, one per lane; what passes is compacted again into the batch of 64 candidates that vet_header takes, one per
// lane.  Same candidates in the same (ascending) order.  The search keeps no queues between calls: `pos` is the next bit offset to look at --
// behind a batch it is the bit after the batch's last candidate (the few survivors that had been collected behind it are found again), and a
// chunk whose speculative decode failed goes on from the bit behind that start.
struct Search2 { uint64_t pos; };
__device__ __forceinline__ uint32_t lanes_below(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
// list1: 128 entries, list2: 64 entries (LDS; free while a chunk searches).  Returns this lane's candidate of the batch (~0: none) and the batch's size in n.
__device__ __forceinline__ uint64_t collect_candidates(Search2 &S, uint32_t *list1, uint32_t *list2, const uint32_t *words, uint64_t wmask, uint64_t to_bit, uint64_t size_bits,
                                                       uint32_t lane, uint32_t &n_out, bool &dry)
{
    const uint64_t base = S.pos;                    // (offsets in the lists are relative to it: a chunk's range is far below 2^32 bits)
    uint64_t pos = S.pos;
    uint32_t n = 0, c1 = 0;
    for (;;) {
        // ---- the cheap test, four windows of 64 offsets at a time (their loads in flight together), survivors to list1
        while (c1 < 64 && pos < to_bit) {
            uint32_t w0[SEARCH_W], w1[SEARCH_W]; bool in[SEARCH_W];
            const uint64_t p0 = pos;
#pragma unroll
            for (int j = 0; j < SEARCH_W; j++) {
                const uint64_t bit = p0 + (uint64_t)(64 * j) + lane;
                in[j] = bit < to_bit && bit + 128 <= size_bits;
                const uint64_t wi = in[j] ? bit >> 5 : 0;
                w0[j] = words[wi & wmask]; w1[j] = words[(wi + 1) & wmask];
            }
#pragma unroll
            for (int j = 0; j < SEARCH_W; j++) {
                const uint64_t bit = p0 + (uint64_t)(64 * j) + lane;
                const uint32_t h0 = __funnelshift_r(w0[j], w1[j], (uint32_t)bit & 31u);
                const bool ok = in[j] && (h0 & 7u) == 4u && ((h0 >> 3) & 31u) <= 29u && ((h0 >> 8) & 31u) <= 29u;      // BFINAL = 0, BTYPE = 10b, HLIT, HDIST in range
                const uint64_t m = __ballot(ok);
                if (c1 < 64) {                      // (a window that would take the list past its 128 entries is looked at again next time)
                    if (ok) list1[c1 + lanes_below(m)] = (uint32_t)(bit - base);
                    c1 += (uint32_t)__popcll(m);
                    pos += 64;
                }
            }
        }
        const uint32_t take = c1 < 64 ? c1 : 64;
        if (!take) { dry = true; S.pos = to_bit; break; }
        __syncthreads();
        // ---- the Kraft sum of the precode, on 64 survivors at once
        bool ok = false; uint32_t off = 0;
        if (lane < take) {
            off = list1[lane];
            const uint64_t bit = base + off, wi = bit >> 5;
            const uint32_t s = (uint32_t)bit & 31u;
            const uint32_t x0 = words[wi & wmask], x1 = words[(wi + 1) & wmask], x2 = words[(wi + 2) & wmask], x3 = words[(wi + 3) & wmask];
            const uint32_t h0 = __funnelshift_r(x0, x1, s), h1 = __funnelshift_r(x1, x2, s), h2 = __funnelshift_r(x2, x3, s);
            const uint32_t hclen = ((h0 >> 13) & 15u) + 4;
            const uint64_t a = h0 | ((uint64_t)h1 << 32), b = h1 | ((uint64_t)h2 << 32);
            uint32_t k = 0;
#pragma unroll
            for (uint32_t i = 0; i < 19; i++) {
                const uint32_t p = 17 + 3 * i;
                const uint32_t l = (uint32_t)(p < 32 ? (a >> p) : (b >> (p - 32))) & 7u;
                if (i < hclen && l) k += 128u >> l;
            }
            ok = k == 128u;
        }
        const uint64_t m2 = __ballot(ok);
        const uint32_t r2 = n + lanes_below(m2), cnt2 = (uint32_t)__popcll(m2);
        if (ok && r2 < 64) list2[r2] = off;
        if (n + cnt2 >= 64) {                       // the batch is full: the search goes on behind its last candidate
            const uint64_t last = __ballot(ok && r2 == 63);
            const uint32_t last_off = __shfl(off, __ffsll((unsigned long long)last) - 1);
            S.pos = base + last_off + 1; n = 64;
            break;
        }
        n += cnt2;
        // what is left of list1 moves to its front
        __syncthreads();
        uint32_t mv0 = 0;
        if (lane + 64 < c1) mv0 = list1[lane + 64];
        __syncthreads();
        if (lane + 64 < c1) list1[lane] = mv0;
        c1 -= take;
        __syncthreads();
        if (pos >= to_bit && c1 == 0) { dry = true; S.pos = to_bit; break; }
    }
    __syncthreads();
    n_out = n;
    return lane < n ? base + list2[lane] : ~0ull;
}
__device__ __forceinline__ uint64_t next_vetted2(Search2 &S, Vetted &V, uint32_t *list1, uint32_t *list2, const uint32_t *words, uint64_t wmask, uint64_t max_dw, uint64_t to_bit,
                                                 uint64_t size_bits, uint32_t lane)
{
    for (;;) {
        if (V.vmask) {
            const int idx = __ffsll((unsigned long long)V.vmask) - 1;
            V.vmask &= V.vmask - 1;
            return __shfl(V.cand, idx);
        }
        if (V.dry) return ~0ull;
        uint32_t n = 0;
        const uint64_t mine = collect_candidates(S, list1, list2, words, wmask, to_bit, size_bits, lane, n, V.dry);
        const bool ok = lane < n && vet_header(words, wmask, max_dw, mine, size_bits);
        V.cand = mine; V.vmask = __ballot(ok);
    }
}

enum WalkEnd : uint32_t { W_MORE = 0, W_EOB = 1, W_ERROR = 2 };

// Walk codes until the round is full, the block ends or something is wrong.  One dependent chain.  Outside the assembly loop
// (the rare paths: long codes, end of block, the refill in front of them) every lane runs it with the same values in VECTOR
// registers -- `z` is a zero the compiler cannot see through; branch conditions go through a ballot, which makes them scalar
// branches without exec-mask juggling; only lane 0's stores reach the list, the other lanes store to slots of their own.
// The bit buffer is kept two bits up (bits 2.. of `lo` are the next bits of the stream).
#define GZ_UNI(cond) (__builtin_amdgcn_ballot_w64(cond) != 0)
__device__ __forceinline__ uint32_t lds_off(const void *p) { return (uint32_t)(size_t)p; }
__device__ __forceinline__ uint32_t walk(Lds &L, BitRd &rd, uint32_t lane, uint32_t &n_out, TabRegs &T)
{
    static_assert(LIT_SIZE == 1024 && DIST_SIZE == 512 && RING * 4 == 2048, "the assembly below has these sizes in its masks and register counts");
    uint32_t z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    uint32_t lo = (uint32_t)(rd.bb << 2) | z, hi = (uint32_t)(rd.bb >> 30) | z;
    uint32_t bc = rd.bc | z, nd = rd.nd | z, mtot = z;
    const uint32_t r4_in = (rd.rd_dw << 2);
    uint32_t r4 = r4_in | z;                    // 4 * (index of the dword nd holds), low 32 bits: the ring address and, by difference, the dwords taken
    uint32_t n = 0, end = W_MORE;
    uint32_t la = lds_off(L.sym) + (lane == 0 ? 0 : (ROUND + 2 * lane) * 4);
    const uint32_t step1 = lane == 0 ? 4 : 0;
    // refill below 30 valid bits (so that the shift of the incoming dword, bc + 2, stays below 32): 30 <= bc <= 61 afterwards
#define GZ_REFILL() do { if (GZ_UNI(bc < 30u)) { lo |= nd << (bc + 2); hi |= nd >> (30 - bc); bc += 32; r4 += 4; nd = L.ring[(r4 >> 2) & (RING - 1)] | z; } } while (0)
#define GZ_DROP(x) do { const uint32_t x_ = (x); lo = __builtin_amdgcn_alignbit(hi, lo, x_); hi >>= (x_ & 31u); bc -= x_; } while (0)
#define GZ_PEEK(nb) __builtin_amdgcn_ubfe(lo, 2u, (nb))
    const uint32_t k_len = 0xA0000000u, k_sign = 0x80000000u, k_limit = STG - 260 - 2 * ROUND;
    static_assert(offsetof(Lds, ring) == 0, "offset used by the assembly");
    while (n < ROUND - 1) {
        uint32_t e, t, d, lenm3, reason;
        n = (uint32_t)__builtin_amdgcn_readfirstlane(n);
        // The walk proper.  Literal entries two lookups per pass; a length entry takes its extra bits, the distance code and its
        // extra bits and leaves one list entry; the refill (ring read of the NEXT dword issued when the current one is taken)
        // is part of the loop.  It leaves for what is rare: reason 0 = the list (or the staging buffer) is full; 2 = e is an
        // end of block, a long code or invalid, nothing of it dropped; 3 = d is a long or invalid distance code, the length
        // (less 3, in lenm3) already taken.
        // Inside the loop everything that is serial is SCALAR: bit buffer (s[20:21], two bits up), bit count (s22), ring position
        // (s26), the fields of a table entry, the list entry of a match.  A drop is one s_lshr_b64 (its shift operand takes the
        // low six bits of the entry as they are) and a subtraction.  The two first-level tables are copied from LDS into 24 vector
        // registers (1024 + 512 entries = 16 + 8 registers of 64 lanes) when the loop is entered, and a lookup is a register read:
        // entry i sits in lane i & 63 of register i >> 6 -- the loop runs in GPR index mode (source 0 relative: s_set_gpr_idx_idx
        // picks the register) and v_readlane picks the lane.  EXEC is lane 0 alone in the loop, so the few vector instructions
        // left (entry to a vector register, list store, list address) are lane 0's; none of them has a vector register as source 0.
#define GZ_A_REFILL \
            "s_waitcnt lgkmcnt(0)\n\t" \
            "s_set_gpr_idx_idx s25\n\t" \
            "v_readfirstlane_b32 s24, %[nd]\n\t" \
            "s_add_u32 s28, s22, 2\n\t" \
            "s_lshl_b64 s[30:31], s[24:25], s28\n\t" \
            "s_or_b64 s[20:21], s[20:21], s[30:31]\n\t" \
            "s_add_u32 s22, s22, 32\n\t" \
            "s_add_u32 s26, s26, 4\n\t" \
            "s_and_b32 s28, s26, 0x7fc\n\t" \
            "v_mov_b32 %[t], s28\n\t" \
            "ds_read_b32 %[nd], %[t]\n"
#define GZ_A_DROP_CODE \
            "s_lshr_b64 s[20:21], s[20:21], s27\n\t" \
            "s_and_b32 s28, s27, 0xff\n\t" \
            "s_sub_u32 s22, s22, s28\n\t"
#define GZ_A_DROP_X \
            "s_lshr_b64 s[20:21], s[20:21], s29\n\t" \
            "s_sub_u32 s22, s22, s29\n\t"
#define GZ_A_LOOKUP(regbits, first) \
            "s_bfe_u32 s28, s20, " regbits "\n\t" \
            "s_bfe_u32 s23, s20, 0x60002\n\t" \
            "s_set_gpr_idx_idx s28\n\t" \
            "v_readlane_b32 s27, " first ", s23\n\t" \
            "s_cmp_lt_i32 s27, 0\n\t"
        /* x extra bits (count in s29) from the bit buffer into s28 */
#define GZ_A_EXTRA \
            "s_lshr_b32 s28, s20, 2\n\t" \
            "s_bfm_b32 s35, s29, 0\n\t" \
            "s_and_b32 s28, s28, s35\n\t"
        asm volatile(
            "v_readfirstlane_b32 s20, %[lo]\n\t"
            "v_readfirstlane_b32 s21, %[hi]\n\t"
            "v_readfirstlane_b32 s22, %[bc]\n\t"
            "v_readfirstlane_b32 s26, %[r4]\n\t"
            "v_readfirstlane_b32 s34, %[mtot]\n\t"
            "s_mov_b32 s25, 0\n\t"
            "s_mov_b32 s27, 0\n\t"
            "s_mov_b32 s38, 0\n\t"
            "s_mov_b32 %[reason], 0\n\t"
            "s_mov_b64 s[36:37], exec\n\t"
            "s_mov_b64 exec, 1\n\t"
            "s_set_gpr_idx_on s25, gpr_idx(SRC0)\n"
            ".Lgz_top_%=:\n\t"
            "s_cmp_lt_u32 s22, 30\n\t"
            "s_cbranch_scc0 .Lgz_pair_%=\n\t"
            GZ_A_REFILL
            ".Lgz_pair_%=:\n\t"
            GZ_A_LOOKUP("0x40008", "v104")
            "s_cbranch_scc1 .Lgz_nl1_%=\n\t"
            "v_mov_b32 %[e], s27\n\t"
            "ds_write_b32 %[la], %[e]\n\t"
            GZ_A_DROP_CODE
            GZ_A_LOOKUP("0x40008", "v104")
            "s_cbranch_scc1 .Lgz_nl2_%=\n\t"
            "v_mov_b32 %[e], s27\n\t"
            "ds_write_b32 %[la], %[e] offset:4\n\t"
            "v_add_u32 %[la], 8, %[la]\n\t"
            GZ_A_DROP_CODE
            "s_add_u32 %[n], %[n], 2\n\t"
            "s_cmp_lt_u32 %[n], 63\n\t"
            "s_cbranch_scc1 .Lgz_top_%=\n\t"
            "s_branch .Lgz_done_%=\n"
            ".Lgz_nl2_%=:\n\t"
            "v_add_u32 %[la], 4, %[la]\n\t"
            "s_add_u32 %[n], %[n], 1\n"
            ".Lgz_nl1_%=:\n\t"
            "s_cmp_lt_u32 s27, %[klen]\n\t"
            "s_cbranch_scc0 .Lgz_other_%=\n\t"
            GZ_A_DROP_CODE
            "s_bfe_u32 s29, s27, 0x30018\n\t"
            GZ_A_EXTRA
            "s_bfe_u32 s38, s27, 0x80008\n\t"
            "s_add_u32 s38, s38, s28\n\t"
            GZ_A_DROP_X
            "s_cmp_lt_u32 s22, 30\n\t"
            "s_cbranch_scc0 .Lgz_dist_%=\n\t"
            GZ_A_REFILL
            ".Lgz_dist_%=:\n\t"
            GZ_A_LOOKUP("0x30008", "v120")
            "s_cbranch_scc1 .Lgz_dlong_%=\n\t"
            GZ_A_DROP_CODE
            "s_bfe_u32 s29, s27, 0x40018\n\t"
            GZ_A_EXTRA
            "s_bfe_u32 s33, s27, 0xf0008\n\t"
            "s_add_u32 s33, s33, s28\n\t"
            GZ_A_DROP_X
            "s_lshl_b32 s33, s33, 9\n\t"
            "s_or_b32 s33, s33, s38\n\t"
            "s_or_b32 s33, s33, %[sign]\n\t"
            "v_mov_b32 %[t], s33\n\t"
            "ds_write_b32 %[la], %[t]\n\t"
            "v_add_u32 %[la], 4, %[la]\n\t"
            "s_add_u32 s34, s34, s38\n\t"
            "s_add_u32 s34, s34, 3\n\t"
            "s_add_u32 %[n], %[n], 1\n\t"
            "s_cmp_gt_u32 s34, %[limit]\n\t"
            "s_cbranch_scc1 .Lgz_done_%=\n\t"
            "s_cmp_lt_u32 %[n], 63\n\t"
            "s_cbranch_scc1 .Lgz_top_%=\n\t"
            "s_branch .Lgz_done_%=\n"
            ".Lgz_dlong_%=:\n\t"
            "s_mov_b32 %[reason], 3\n\t"
            "s_branch .Lgz_done_%=\n"
            ".Lgz_other_%=:\n\t"
            "s_mov_b32 %[reason], 2\n"
            ".Lgz_done_%=:\n\t"
            "s_set_gpr_idx_off\n\t"
            "s_mov_b64 exec, s[36:37]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_readfirstlane_b32 s24, %[nd]\n\t"
            "v_mov_b32 %[nd], s24\n\t"
            "v_mov_b32 %[lo], s20\n\t"
            "v_mov_b32 %[hi], s21\n\t"
            "v_mov_b32 %[bc], s22\n\t"
            "v_mov_b32 %[r4], s26\n\t"
            "v_mov_b32 %[mtot], s34\n\t"
            "v_mov_b32 %[e], s27\n\t"
            "v_mov_b32 %[d], s27\n\t"
            "v_mov_b32 %[len], s38\n"
            : [lo] "+v"(lo), [hi] "+v"(hi), [bc] "+v"(bc), [nd] "+v"(nd), [r4] "+v"(r4), [la] "+v"(la), [mtot] "+v"(mtot), [n] "+s"(n),
              [e] "=&v"(e), [t] "=&v"(t), [d] "=&v"(d), [len] "=&v"(lenm3), [reason] "=&s"(reason),
              "+{v[104:119]}"(T.lit), "+{v[120:127]}"(T.dist)              // (read only: in and out so that they stay where they are between the rounds)
            : [klen] "s"(k_len), [sign] "s"(k_sign), [limit] "s"(k_limit)
            : "vcc", "scc", "memory", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s33", "s34", "s35", "s36", "s37", "s38");
#undef GZ_A_EXTRA
#undef GZ_A_LOOKUP
#undef GZ_A_DROP_CODE
#undef GZ_A_DROP_X
#undef GZ_A_REFILL
        if (reason == 0) break;
        if (reason == 2) {
            GZ_REFILL();                             // (a second lookup may leave as few as 20 bits; a long code and its extra bits need 20)
            if (GZ_UNI((e & K_MASK) == K_LONG)) {
                const uint32_t rev = __brev(lo >> 2);
                uint32_t s = 0xFFFFu, len = 0;
                for (int l = LIT_BITS + 1; l <= 15; l++) {
                    const uint32_t idx = (rev >> (32 - l)) - L.clit.first[l];
                    if (GZ_UNI(idx < L.clit.cnt[l])) { len = (uint32_t)l; s = L.sorted_lit[L.clit.off[l] + idx]; break; }
                }
                if (GZ_UNI(s == 0xFFFFu)) { end = W_ERROR; break; }
                GZ_DROP(len);
                e = lit_entry(s);
                if (!GZ_UNI((int32_t)e < 0)) { asm volatile("ds_write_b32 %0, %1" :: "v"(la), "v"(e) : "memory"); la += step1; n++; continue; }
            } else GZ_DROP(e & 255u);
            const uint32_t kind = e & K_MASK;
            if (GZ_UNI(kind == K_EOB)) { end = W_EOB; break; }
            if (GZ_UNI(kind != K_LENGTH)) { end = W_ERROR; break; }
            const uint32_t xl = (e >> 24) & 7u;
            lenm3 = ((e >> 8) & 0xFFu) + GZ_PEEK(xl); GZ_DROP(xl);
            GZ_REFILL();
            {   // (the table is in registers: element idx >> 6, lane idx & 63)
                const uint32_t idx = (uint32_t)__builtin_amdgcn_readfirstlane((lo >> 2) & (DIST_SIZE - 1));
                uint32_t row = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) if ((idx >> 6) == (uint32_t)j) row = T.dist[j];          // (constant element indices: the tuple stays in registers)
                d = (uint32_t)__builtin_amdgcn_readlane(row, idx & 63) | z;
            }
        }
        if (GZ_UNI((int32_t)d < 0)) {
            if (GZ_UNI((d & D_INVALID) != 0)) { end = W_ERROR; break; }
            const uint32_t rev = __brev(lo >> 2);
            uint32_t s = 0xFFFFu, dl = 0;
            for (int l = DIST_BITS + 1; l <= 15; l++) {
                const uint32_t idx = (rev >> (32 - l)) - L.cdist.first[l];
                if (GZ_UNI(idx < L.cdist.cnt[l])) { dl = (uint32_t)l; s = L.sorted_dist[L.cdist.off[l] + idx]; break; }
            }
            if (GZ_UNI(s == 0xFFFFu)) { end = W_ERROR; break; }
            GZ_DROP(dl);
            d = dist_entry(s);
            if (GZ_UNI((int32_t)d < 0)) { end = W_ERROR; break; }
        } else GZ_DROP(d & 255u);
        const uint32_t xd = (d >> 24) & 15u;
        const uint32_t distm1 = ((d >> 8) & 0x7FFFu) + GZ_PEEK(xd); GZ_DROP(xd);
        const uint32_t m = 0x80000000u | lenm3 | (distm1 << 9);
        asm volatile("ds_write_b32 %0, %1" :: "v"(la), "v"(m) : "memory");
        la += step1; n++;
        mtot += lenm3 + 3;
        if (GZ_UNI(mtot > k_limit)) break;           // the staging buffer takes what is listed so far plus a round of literals
    }
#undef GZ_REFILL
#undef GZ_DROP
#undef GZ_PEEK
    // (the builtin returns int: without the casts the low half is sign-extended over the high one)
    const uint64_t bb4 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(hi) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(lo);
    rd.bb = bb4 >> 2;
    rd.bc = (uint32_t)__builtin_amdgcn_readfirstlane(bc); rd.nd = (uint32_t)__builtin_amdgcn_readfirstlane(nd);
    rd.rd_dw += ((uint32_t)__builtin_amdgcn_readfirstlane(r4) - r4_in) >> 2;
    n_out = (uint32_t)__builtin_amdgcn_readfirstlane(n);
    return (uint32_t)__builtin_amdgcn_readfirstlane(end);
}

// all lanes: the round's list -> symbols at out[opos ..]
// returns the number of symbols written
__device__ __forceinline__ uint32_t expand(Lds &L, uint32_t n, uint16_t *out, uint64_t opos, uint32_t lane)
{
    const uint32_t s0 = lane < n ? L.sym[lane] : 0;
    const uint32_t cnt = lane < n ? ((s0 >> 31) ? (s0 & 0x1FFu) + 3 : 1 + ((s0 >> 24) & 1u)) : 0;
    uint32_t inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d); if (lane >= (uint32_t)d) inc += t; }
    const uint32_t tot = __shfl(inc, 63);
    if (__ballot(s0 >> 31) == 0) {               // literals only (most rounds of FASTQ): every entry stores its one or two symbols itself
        if (lane < n) { out[opos + inc - cnt] = (uint16_t)((s0 >> 8) & 0xFFu); if (cnt == 2) out[opos + inc - 1] = (uint16_t)((s0 >> 16) & 0xFFu); }
        return tot;
    }
    L.soff[lane] = (uint16_t)(inc - cnt);
    __syncthreads();
    bool pending = false;
    for (uint32_t p = lane; p < tot; p += 64) {
        uint32_t lo = 0;
#pragma unroll
        for (uint32_t step = 32; step; step >>= 1) { const uint32_t j = lo + step; if (j < n && L.soff[j] <= p) lo = j; }
        const uint32_t s = L.sym[lo], o = L.soff[lo];
        uint32_t v;
        if (!(s >> 31)) v = (p == o ? (s >> 8) : (s >> 16)) & 0xFFu;
        else {
            const uint32_t len = (s & 0x1FFu) + 3, D = ((s >> 9) & 0x7FFFu) + 1, t = p - o;
            const uint32_t r = D < len ? t % D : t;
            const int32_t srel = (int32_t)o - (int32_t)D + (int32_t)r;
            if (srel >= 0) { v = PENDING | (uint32_t)srel; pending = true; }
            else {
                const int64_t g = (int64_t)opos + srel;
                v = g >= 0 ? out[g] : (uint32_t)(GZ_MARK | (uint32_t)((int64_t)GZ_WINDOW + g));
            }
        }
        L.stg[p] = (uint16_t)v;
    }
    __syncthreads();
    while (__any(pending)) {                     // references into the round itself: they point strictly backwards
        pending = false;
        for (uint32_t p = lane; p < tot; p += 64) {
            const uint32_t v = L.stg[p];
            if ((v & 0xC000u) == PENDING) {
                const uint32_t w = L.stg[v & 0x3FFFu];
                if ((w & 0xC000u) == PENDING) pending = true; else L.stg[p] = (uint16_t)w;
            }
        }
        __syncthreads();
    }
    for (uint32_t p = lane; p < tot; p += 64) out[opos + p] = L.stg[p];
    return tot;
}

__global__ __launch_bounds__(64) void gz_decode_kernel(const uint8_t *data, uint64_t ring_mask, uint64_t size, uint64_t limit_bytes, uint64_t base_byte,
                                                       uint64_t chunk_bytes, uint32_t chunk_lo, uint32_t exact_chunk, uint64_t exact_bit, uint16_t *sym,
                                                       uint64_t sym_cap, GzChunk *chunks)
{
    __shared__ Lds L;
    const uint32_t lane = threadIdx.x;
    const uint32_t c = chunk_lo + blockIdx.x;
    uint16_t *out = sym + (uint64_t)blockIdx.x * sym_cap;
    const uint64_t size_bits = size * 8, limit_bits = limit_bytes * 8;      // limit: what of the stream is on the device (the rest of the ring holds other bytes)
    const uint64_t nominal = (base_byte + (uint64_t)c * chunk_bytes) * 8, stop_bit = nominal + chunk_bytes * 8;
    const uint64_t search_end = stop_bit < size_bits ? stop_bit : size_bits;
    GzChunk res; res.start_bit = 0; res.end_bit = 0; res.n_sym = 0; res.status = GZ_NONE;
    BitRd rd; rd.base = reinterpret_cast<const uint4 *>(data); rd.vmax = (size + 48) / 16;      // readable (and zero) up to size + 64
    rd.vmask = ring_mask >> 4;
    rd.origin_v = (c == exact_chunk && exact_bit < nominal ? exact_bit : nominal) >> 7;
    rd.ring = L.ring; rd.rd_dw = 0; rd.ring_hi = 0; rd.nd = 0; rd.bb = 0; rd.bc = 0;
    // A speculative chunk tries the candidates of its range in order: one whose header parses strictly is decoded; if the data
    // behind it turns out not to decode (a false candidate) the search goes on behind it.
    TabRegs T;
    T.lit = (u32x16)(0u); T.dist = (u32x8)(0u);
    bool searching = c != exact_chunk;
    Search S; search_init(S, nominal);
    uint64_t start = exact_bit, opos = 0, blk_pos = exact_bit, blk_opos = 0;
    uint32_t status = GZ_NONE;
    if (!searching) rd.seek(start, lane);
    else if (nominal >= size_bits) { if (lane == 0) chunks[c] = res; return; }
    uint64_t behind_from = ~0ull;                   // the first block boundary behind the chunk's own range
    uint32_t why = 0;                               // what stopped a decode that failed (reported in n_sym of a GZ_FAILED chunk)
    for (;;) {
        uint32_t final = 0, type = 2;
        bool strict = false;
        why = 0;
        if (searching) {
            start = next_candidate(S, reinterpret_cast<const uint32_t *>(data), ring_mask >> 2, search_end, size_bits, lane);
            if (start == ~0ull) { status = GZ_NONE; break; }
            rd.seek(start + 3, lane);
            strict = true; opos = 0; blk_pos = start; blk_opos = 0;
        } else {
            blk_pos = rd.bitpos(); blk_opos = opos;
            // Behind its own range a chunk stops in front of the block its successor has started at -- and that can only be a
            // non-final dynamic block (what next_candidate looks for): stored and fixed blocks and the member's final block are
            // decoded here too (the empty stored blocks pigz and Z_SYNC_FLUSH writers leave every few ten kilobytes would each be
            // a gap for the host to bridge otherwise), for at most one chunk's worth of input behind the range.
            const bool behind = blk_pos >= stop_bit;
            if (behind && behind_from == ~0ull) behind_from = blk_pos;
            if (behind && (blk_pos + 3 > size_bits || blk_pos - behind_from >= chunk_bytes * 8)) { status = GZ_AT_BOUNDARY; break; }
            if (blk_pos + 3 > size_bits) why = 1;
            else {
                rd.top_up(lane);
                rd.refill(); final = rd.peek(1); type = rd.peek(3) >> 1;
                if (behind && !final && type == 2) { status = GZ_AT_BOUNDARY; break; }
                rd.drop(3);
                if (type == 3) why = 2;
            }
            if (!why && type == 0) {               // stored block: byte aligned LEN, ~LEN, then the bytes
                rd.drop(rd.bc & 7); rd.refill();
                const uint32_t len = rd.peek(16); rd.drop(16); rd.refill();
                const uint32_t nlen = rd.peek(16); rd.drop(16);
                const uint64_t byte = rd.bitpos() >> 3;
                if ((len ^ nlen) != 0xFFFFu || byte + len > size) why = 3;
                else if (byte + len > limit_bytes) why = 8;
                else {
                    if (opos + len > sym_cap) { status = GZ_OVERFLOW; break; }
                    for (uint32_t i = lane; i < len; i += 64) out[opos + i] = data[(byte + i) & ring_mask];
                    opos += len;
                    rd.seek((byte + len) * 8, lane);
                    if (final) { status = GZ_MEMBER_END; break; }
                    continue;
                }
            }
        }
        if (!why) {
            uint32_t hlit = 288, hdist = 32;
            bool ok = true;
            if (type == 1) {
                for (uint32_t i = lane; i < 320; i += 64) L.lens[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : (i < 288 ? 8 : 5)));
                __syncthreads();
            } else ok = read_code_lengths(L, rd, strict, lane, hlit, hdist);
            if (!ok) why = 4;
            else if (!build_tables(L, hlit, hdist, strict, lane)) why = 5;
            if (!why) {                      // the tables move into registers; their LDS bytes become the rounds' list and staging buffers
#pragma unroll
                for (int j = 0; j < 16; j++) T.lit[j] = L.lit[j * 64 + lane];
#pragma unroll
                for (int j = 0; j < 8; j++) T.dist[j] = L.dist[j * 64 + lane];
            }
            __syncthreads();
        }
        if (!why) {
            searching = false;
            uint32_t end = W_MORE;
            while (end == W_MORE) {
                if (opos + STG > sym_cap) { status = GZ_OVERFLOW; break; }
                rd.top_up(lane);
                uint32_t n = 0;
                end = walk(L, rd, lane, n, T);
                if (end == W_ERROR) why = 6;
                if (rd.bitpos() > size_bits) { end = W_ERROR; why = 7; }
                else if (rd.bitpos() > limit_bits) { end = W_ERROR; why = 8; }          // ran into bytes that are not on the device yet: the host bridges this stretch
                __syncthreads();
                if (end == W_ERROR) break;
                opos += expand(L, n, out, opos, lane);
                __syncthreads();
            }
            if (status == GZ_OVERFLOW) break;
        }
        if (why) {
            // not deflate data.  From a speculative start that only says the candidate was false: the search goes on behind it
            // (S still holds the rest of the candidates of its step)
            if (c != exact_chunk && why != 8) { searching = true; continue; }
            status = GZ_FAILED; break;
        }
        if (final) { status = GZ_MEMBER_END; break; }
    }
    if (lane == 0) {
        res.start_bit = start;
        if (status == GZ_OVERFLOW) { res.end_bit = blk_pos; res.n_sym = (uint32_t)blk_opos; }
        else if (status == GZ_AT_BOUNDARY) { res.end_bit = blk_pos; res.n_sym = (uint32_t)opos; }
        else { res.end_bit = rd.bitpos(); res.n_sym = (uint32_t)opos; }
        if (status == GZ_NONE) res.n_sym = 0;
        if (status == GZ_FAILED) res.n_sym = why;
        res.status = status;
        chunks[c] = res;
    }
}

// ------------------------------------------------------------------------------------------ the lane-parallel decoder
// gz_decode_kernel above walks a block's codes on ONE lane (290 cycles a code, and a CU's instruction issue is what its sixteen
// wavefronts share: the chip full, it streams 18-24 GB/s of text).  This kernel keeps everything around the walk -- one wavefront
// per chunk, the search for a block start, the header and table construction by all lanes, 16-bit symbols with markers -- and
// replaces the walk and the expansion:
//   * THE WALK IS DONE BY ALL 64 LANES AT ONCE.  A step covers 64 spans of SPAN bits of the block; lane i walks the codes that
//     start in span i (mf_gzlane.h), from where lane i - 1 stopped.  That it has to guess at first (the span's border); a Huffman
//     stream decoded from a wrong offset falls into step with the true one within a few dozen codes, so nearly every lane ENDS at
//     the right bit all the same.  Lanes whose start was wrong walk again from their predecessor's end, until no start moves
//     (measured on FASTQ at gzip -1 / -6 / -9: one re-walk, rarely two; tools/gzlane_model.cpp).  Lane 0 starts at the true
//     position, so by induction every lane of the chain holds true codes.  Each lane leaves its codes as list entries in its own
//     stretch of a scratch list in global memory (16-byte stores of four entries).
//   * THE EXPANSION takes the lists lane by lane, RN entries a round (twice the old round, from a list that is already there: its
//     loads are issued ahead), every output position of the round finding its entry by binary search in LDS, as before.
// The tables stay in LDS (the lanes index them on their own); ring, precode and code-length arrays share their bytes with the
// round's list and staging buffer.
// SPAN sets the LDS a wavefront holds (a step's staged rows) and with it how many wavefronts fit a CU: 1024 bits = 16.4 KB = 9 wavefronts,
// 768 = 15.2 KB = 10, 512 = 12.4 KB with 128-entry rounds = 12 (registers bind then).  Same box, the chip full, bytes equal to zlib's each
// (profiles/r05/e_gzdev_occupancy_variants.txt): 1024: 80.0 GB/s of text (a 1.3 GB gzip -6 file of 2 384 chunks, which 9 x 256 slots do not hold
// in one go: 54.2), 768: 89.2 (78.6), 512: 82.4 (73.1), 512 with 128-entry rounds: 83.2 (71.8).
#ifndef GZ_SPAN
#define GZ_SPAN 768
#endif
constexpr uint32_t SPAN = GZ_SPAN;       // bits of the block per lane and step
constexpr uint32_t LCAP = 512;           // list entries a lane may leave per step (more codes than that in a span: the lane stops early, the next step goes on from there)
#ifndef GZ_RN
#define GZ_RN 256
#endif
constexpr uint32_t RN = GZ_RN;           // list entries per round of the expansion
constexpr uint32_t STG2 = RN * 8;        // output symbols per round (staging buffer)
constexpr uint32_t MAX_REWALK = 6;       // re-walk rounds per step; what is not chained by then waits for the next step
// dwords of the stream a lane may touch in a step: its span, a code that starts in the span's last bit (48 bits), the bit buffer's
// 64 bits and one dword read ahead.  Odd, so that the lanes' rows start in different LDS banks.
constexpr uint32_t SPAN_DW = SPAN / 32, STAGE_DW = SPAN_DW + 5;
static_assert(SPAN % 32 == 0 && STAGE_DW % 2 == 1 && RN % 64 == 0 && STG2 % 512 == 0 && STG2 <= 0x10000, "layout of the rounds");
struct Lds2 {
    uint32_t lit[LIT_SIZE]; uint32_t dist[DIST_SIZE];
    union {
        struct { uint32_t ring[RING]; uint32_t pre[PRE_SIZE]; uint8_t lens[328]; uint8_t plens[24]; };      // between blocks: the header reader's
        uint32_t stage[64 * STAGE_DW];                                                                       // a step's walks: the stream, a row per lane
        uint32_t stg[STG2];                                                                                  // a step's expansion: the round's symbols, one 32-bit cell each
    };
    uint16_t sorted_lit[288], sorted_dist[32];
    Canon clit, cdist;
};

// dword w of the stream, counted from the chunk's origin (global memory, through the ring)
struct LaneIn {
    const uint32_t *words; uint64_t origin_dw, wmask, max_dw;
    __device__ __forceinline__ uint32_t operator()(uint32_t w) const { uint64_t i = origin_dw + w; i = i < max_dw ? i : max_dw; return words[i & wmask]; }
};
// ... from the lane's row of the step's staged input (LDS): the walk never waits for global memory
struct LaneInLds {
    const uint32_t *row; uint32_t w_first;
    __device__ __forceinline__ uint32_t operator()(uint32_t w) const { uint32_t k = w - w_first; k = k < STAGE_DW ? k : STAGE_DW - 1; return row[k]; }
};
// The stream from bit b_rel on, 64 spans and what a lane reads past its span, into LDS: row i = dwords (b_rel >> 5) + SPAN_DW i .. + STAGE_DW - 1.
// Loaded coalesced, all loads in flight together (one memory latency per step instead of one per few codes).
__device__ __forceinline__ void stage_input(Lds2 &L, const LaneIn &in, uint32_t b_rel, uint32_t lane)
{
    const uint32_t w0 = b_rel >> 5;
    constexpr uint32_t TOTAL = 64 * SPAN_DW + (STAGE_DW - SPAN_DW), ROUNDS = (TOTAL + 63) / 64, BATCH = 7;      // (seven loads in flight a thread, five batches)
#pragma unroll 1
    for (uint32_t i0 = 0; i0 < ROUNDS; i0 += BATCH) {
        uint32_t v[BATCH];
#pragma unroll
        for (uint32_t i = 0; i < BATCH; i++) { const uint32_t g = lane + 64 * (i0 + i); v[i] = g < TOTAL ? in(w0 + g) : 0u; }
#pragma unroll
        for (uint32_t i = 0; i < BATCH; i++) {
            const uint32_t g = lane + 64 * (i0 + i), row = g / SPAN_DW, col = g % SPAN_DW;
            if (g < TOTAL) {
                if (row < 64) L.stage[row * STAGE_DW + col] = v[i];
                if (col < STAGE_DW - SPAN_DW && row >= 1) L.stage[(row - 1) * STAGE_DW + SPAN_DW + col] = v[i];      // (the head of a span is the tail of the row before)
            }
        }
    }
    __syncthreads();
}
// a lane's list entries: four at a time, one 16-byte store
struct LaneOut {
    uint32_t *base; uint32_t n, e0, e1, e2;
    __device__ __forceinline__ void put(uint32_t e)
    {
        const uint32_t k = n & 3u;
        if (k == 3u) { uint4 v; v.x = e0; v.y = e1; v.z = e2; v.w = e; *reinterpret_cast<uint4 *>(base + (n - 3u)) = v; }
        else if (k == 0u) e0 = e; else if (k == 1u) e1 = e; else e2 = e;
        n++;
    }
    __device__ __forceinline__ void flush()
    {
        const uint32_t k = n & 3u, b = n - k;
        if (k > 0u) base[b] = e0;
        if (k > 1u) base[b + 1] = e1;
        if (k > 2u) base[b + 2] = e2;
    }
};

// The lane's walk of mf_gzlane.h (same codes, same list entries, same end: tools/gzdev_check holds it to zlib) written for the staged
// rows: there is no bit buffer to keep filled -- the 32 bits at the lane's position are two LDS dwords and one v_alignbit, taken
// afresh for every code (and once more for a match's distance), so the state of the walk is the position alone and everything
// is 32-bit arithmetic.  A length code and its extra bits are at most 20 bits, a distance code and its extra bits at most 28.
__device__ __forceinline__ uint32_t bits_at(const uint32_t *row, uint32_t w_first, uint32_t pos)
{
    const uint32_t k = (pos >> 5) - w_first;
    return __builtin_amdgcn_alignbit(row[k + 1], row[k], pos & 31u);
}
__device__ __forceinline__ Span walk_rows(const Lds2 &L, const uint32_t *row, uint32_t w_first, uint32_t start, uint32_t stop, uint32_t max_codes, LaneOut &out)
{
    Span r; r.end = start; r.n_code = 0; r.n_sym = 0; r.flags = 0;
    uint32_t pos = start;
    for (;;) {
        if (pos >= stop) { r.end = pos; break; }
        if (r.n_code >= max_codes) { r.end = pos; r.flags |= SP_FULL; break; }
        const uint32_t win = bits_at(row, w_first, pos);
        uint32_t e = L.lit[win & (LIT_SIZE - 1)];
        if ((int32_t)e >= 0) {                  // one or two literals
            uint32_t l = e & 255u;
            if ((e & E_DOUBLE) && pos + ((e >> 25) & 15u) >= stop) { l = (e >> 25) & 15u; e &= 0xFF00u; }      // (a span ends at the first code boundary at or behind `stop`: mf_gzlane.h)
            pos += l;
            r.n_sym += 1 + ((e >> 24) & 1u);
            out.put(e);
            r.n_code++;
            continue;
        }
        uint32_t used;                          // bits of `win` taken so far
        if ((e & K_MASK) == K_LONG) {           // longer than the table's index: canonical decode
            const uint32_t rev = __brev(win);
            uint32_t s = 0xFFFFu, len = 0;
            for (uint32_t l = LIT_BITS + 1; l <= 15; l++) {
                const uint32_t idx = (rev >> (32 - l)) - L.clit.first[l];
                if (idx < L.clit.cnt[l]) { len = l; s = L.sorted_lit[L.clit.off[l] + idx]; break; }
            }
            if (s == 0xFFFFu) { r.end = pos; r.flags |= SP_ERR; break; }
            e = lit_entry(s);
            if ((int32_t)e >= 0) { pos += len; r.n_sym++; out.put(e); r.n_code++; continue; }
            used = len;
        } else used = e & 255u;
        const uint32_t kind = e & K_MASK;
        if (kind == K_EOB) { r.end = pos + used; r.flags |= SP_EOB; break; }
        if (kind != K_LENGTH) { r.end = pos; r.flags |= SP_ERR; break; }
        const uint32_t xl = (e >> 24) & 7u;
        const uint32_t lenm3 = ((e >> 8) & 0xFFu) + __builtin_amdgcn_ubfe(win, used, xl);
        const uint32_t pos_d = pos + used + xl;
        const uint32_t win2 = bits_at(row, w_first, pos_d);
        uint32_t d = L.dist[win2 & (DIST_SIZE - 1)], used2;
        if ((int32_t)d < 0) {
            if (d & D_INVALID) { r.end = pos; r.flags |= SP_ERR; break; }
            const uint32_t rev = __brev(win2);
            uint32_t s = 0xFFFFu, dl = 0;
            for (uint32_t l = DIST_BITS + 1; l <= 15; l++) {
                const uint32_t idx = (rev >> (32 - l)) - L.cdist.first[l];
                if (idx < L.cdist.cnt[l]) { dl = l; s = L.sorted_dist[L.cdist.off[l] + idx]; break; }
            }
            if (s == 0xFFFFu) { r.end = pos; r.flags |= SP_ERR; break; }
            d = dist_entry(s);
            if ((int32_t)d < 0) { r.end = pos; r.flags |= SP_ERR; break; }
            used2 = dl;
        } else used2 = d & 255u;
        const uint32_t xd = (d >> 24) & 15u;
        const uint32_t distm1 = ((d >> 8) & 0x7FFFu) + __builtin_amdgcn_ubfe(win2, used2, xd);
        pos = pos_d + used2 + xd;
        r.n_sym += lenm3 + 3;
        out.put(0x80000000u | lenm3 | (distm1 << 9));
        r.n_code++;
    }
    return r;
}

// cycle counts per phase of a chunk (tools/gzdev_check built with -DGZ_PROFILE reads them from the head of the chunk's list scratch)
#ifdef GZ_PROFILE
struct Prof { unsigned long long t[16]; };      // 0 search + header reads, 1 table construction, 2 walks, 3 expansion, 4 steps, 5 walk rounds, 6 expansion rounds, 7 blocks, 8 round: list + sums, 9 round: cells, 13 round: loads of earlier output, 10 round: chase, 11 round: store, 12 chase passes
#define PROF_T0() const unsigned long long prof_t0_ = __builtin_readcyclecounter()
#define PROF_ADD(i) do { prof.t[i] += __builtin_readcyclecounter() - prof_t0_; } while (0)
#define PROF_STAMP(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); prof.t[i] += n_ - prof_last_; prof_last_ = n_; } while (0)
#define PROF_CNT(i, n) do { prof.t[i] += (n); } while (0)
#else
struct Prof {};
#define PROF_T0() do {} while (0)
#define PROF_ADD(i) do {} while (0)
#define PROF_STAMP(i) do {} while (0)
#define PROF_CNT(i, n) do {} while (0)
#endif

// One round of the expansion: up to RN entries of the step's list at cl[0 .. n) -> symbols at out[opos ..].  Takes as many entries
// as fit the staging buffer (at least one); returns their number in `taken` and the number of symbols written.
// Every thread has four consecutive entries of the list and expands them itself: a literal entry is one or two stores to the
// staging buffer; a match writes its symbols eight at a time -- a symbol of the round itself as a reference that is followed
// afterwards (pointer doubling: the passes grow with the logarithm of the longest chain), an earlier symbol of the chunk straight from
// global memory, all eight loads in flight together, a symbol from in front of the chunk as a marker.  (The first versions found the
// entry of every output POSITION -- by binary search, then by a running maximum over start marks: three and two times the
// instructions, most of them spent on positions that a literal entry settles with one store.)
__device__ __forceinline__ uint32_t expand4(Lds2 &L, const uint32_t *cl, uint32_t n, uint16_t *out, uint64_t opos, uint32_t lane, uint32_t &taken, Prof &prof)
{
    PROF_T0();
    constexpr uint32_t K = RN / 64;
    uint32_t s[K], cnt[K], end[K];
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t q = 0; q < K; q++) { const uint32_t j = lane * K + q; s[q] = j < n ? cl[j] : 0u; }
#pragma unroll
    for (uint32_t q = 0; q < K; q++) {
        const uint32_t j = lane * K + q;
        cnt[q] = j < n ? ((s[q] >> 31) ? (s[q] & 0x1FFu) + 3 : 1 + ((s[q] >> 24) & 1u)) : 0;
        sum += cnt[q]; end[q] = sum;
    }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(inc, d); if (lane >= (uint32_t)d) inc += u; }
    const uint32_t base = inc - sum;
    uint32_t okc = 0, my_end = 0; bool my_match = false;
#pragma unroll
    for (uint32_t q = 0; q < K; q++) {
        end[q] += base;
        const bool ok = lane * K + q < n && end[q] <= STG2;          // (a prefix of the list: the ends grow)
        if (ok) { okc++; my_end = end[q]; my_match = my_match || (s[q] >> 31); }
    }
    uint32_t n_ok = okc, tot = my_end;
    for (int d = 32; d; d >>= 1) { n_ok += __shfl_xor(n_ok, d); tot = max(tot, (uint32_t)__shfl_xor(tot, d)); }
    taken = n_ok;
    PROF_ADD(8);
#ifdef GZ_PROFILE
    unsigned long long prof_last_ = __builtin_readcyclecounter();
#endif
    if (!__any(my_match)) {                      // literals only: every entry stores its one or two symbols itself
#pragma unroll
        for (uint32_t q = 0; q < K; q++) {
            const uint32_t j = lane * K + q, o = end[q] - cnt[q];
            if (j < n_ok) { out[opos + o] = (uint16_t)((s[q] >> 8) & 0xFFu); if (cnt[q] == 2) out[opos + o + 1] = (uint16_t)((s[q] >> 16) & 0xFFu); }
        }
        return tot;
    }
    // The round's symbols are made in a staging buffer of 32-bit cells: a final symbol (kind 0: a literal byte or a marker), a
    // reference to an earlier cell of the round (kind 1), or a reference to earlier output of the chunk (kind 2: how far in front of the
    // round's first symbol, less one).  Symbol t of a match is the symbol D in front of it -- position first + t of the round, with
    // first = (the match's own first position) - D; for a match longer than its distance that is a symbol of the match itself: no modulo
    // anywhere -- so filling the cells is arithmetic and LDS stores only.  The loads from earlier output come afterwards, by POSITION:
    // consecutive lanes take consecutive cells, so the symbols one match copies are neighbours in one or two cache lines, eight loads a
    // lane are in flight together, and every lane loads unconditionally (the chunk's first symbol if its cell needs nothing) with the
    // cases told apart by selects -- a load whose register is also written on another path of a divergent branch makes that path wait
    // for it (s_waitcnt vmcnt(0) in front of the write), which is how the first versions came to pay a memory round trip per symbol.
    constexpr uint32_t C_REF = 1u << 16, C_FAR = 2u << 16;
    bool pending = false;
#pragma unroll
    for (uint32_t q = 0; q < K; q++) {
        const uint32_t j = lane * K + q, o = end[q] - cnt[q], e = s[q];
        if (j < n_ok) {
            if (!(e >> 31)) { L.stg[o] = (e >> 8) & 0xFFu; if (cnt[q] == 2) L.stg[o + 1] = (e >> 16) & 0xFFu; }
            else {
                const uint32_t len = cnt[q];
                const int32_t first = (int32_t)o - (int32_t)(((e >> 9) & 0x7FFFu) + 1);
                pending = pending || first + (int32_t)len > 0;
#pragma unroll
                for (uint32_t u = 0; u < 8; u++) {
                    const int32_t srel = first + (int32_t)u;
                    if (u < len) L.stg[o + u] = srel >= 0 ? (C_REF | (uint32_t)srel) : (C_FAR | (uint32_t)(-srel - 1));
                }
            }
        }
    }
    // what is left of the long matches (all lanes go along while one has something left)
#pragma unroll
    for (uint32_t q = 0; q < K; q++) {
        const uint32_t j = lane * K + q, o = end[q] - cnt[q], e = s[q];
        const uint32_t len = (j < n_ok && (e >> 31) && cnt[q] > 8) ? cnt[q] : 0;
        const int32_t first = (int32_t)o - (int32_t)(((e >> 9) & 0x7FFFu) + 1);
        for (uint32_t t0 = 8; __any(t0 < len); t0 += 8) {
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) {
                const uint32_t t = t0 + u;
                const int32_t srel = first + (int32_t)t;
                if (t < len) L.stg[o + t] = srel >= 0 ? (C_REF | (uint32_t)srel) : (C_FAR | (uint32_t)(-srel - 1));
            }
        }
    }
    __syncthreads();
    PROF_STAMP(9);
    // earlier output of the chunk, by position, eight cells a lane at a time
    for (uint32_t p0 = 0; p0 < tot; p0 += 512) {
        uint32_t c[8], ld[8];
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) {
            const uint32_t p = p0 + 64 * i + lane;
            c[i] = p < tot ? L.stg[p] : 0u;
            const int64_t g = (int64_t)opos - 1 - (int64_t)(c[i] & 0xFFFFu);
            ld[i] = out[((c[i] >> 16) == 2u && g >= 0) ? g : 0];
        }
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) {
            const uint32_t p = p0 + 64 * i + lane;
            const int64_t g = (int64_t)opos - 1 - (int64_t)(c[i] & 0xFFFFu);
            if (p < tot && (c[i] >> 16) == 2u) L.stg[p] = g >= 0 ? ld[i] : (uint32_t)(GZ_MARK | (uint32_t)((int64_t)GZ_WINDOW + g));
        }
    }
    __syncthreads();
    PROF_STAMP(13);
    while (__any(pending)) {                     // references into the round itself: they point strictly backwards
        PROF_CNT(12, 1);
        pending = false;
        for (uint32_t p = lane; p < tot; p += 64) {
            const uint32_t v = L.stg[p];
            if (v >> 16) {
                const uint32_t w = L.stg[v & 0xFFFFu];
                L.stg[p] = w;                    // the symbol -- or the reference it holds itself: the chain to follow halves with every pass
                if (w >> 16) pending = true;
            }
        }
        __syncthreads();
    }
    PROF_STAMP(10);
    for (uint32_t p = lane; p < tot; p += 64) out[opos + p] = (uint16_t)L.stg[p];
    __syncthreads();
    PROF_STAMP(11);
    return tot;
}

enum BlockEnd : uint32_t { B_EOB = 0, B_ERROR = 1, B_OVERFLOW = 2, B_PAST_SIZE = 3, B_PAST_LIMIT = 4 };

// The codes of one block (tables built in L), from bit `from` (absolute) on, by all lanes.  On B_EOB `from` is the bit behind the
// end-of-block code and opos the symbols written.  lst: the chunk's scratch -- a list of LCAP entries per lane, and behind the 64
// of them room for a step's list in one piece.
__device__ __forceinline__ uint32_t decode_block(Lds2 &L, const LaneIn &in, uint64_t origin_bits, uint64_t &from, uint64_t size_bits, uint64_t limit_bits,
                                                  uint16_t *out, uint64_t &opos, uint64_t sym_cap, uint32_t *lst, uint32_t lane, Prof &prof)
{
    uint32_t b_rel = (uint32_t)(from - origin_bits);
    uint32_t *cl = lst + 64 * LCAP;
    for (;;) {
        uint32_t start = b_rel + lane * SPAN;
        const uint32_t stop = b_rel + (lane + 1) * SPAN;
        LaneOut lo; lo.base = lst + lane * LCAP; lo.n = 0; lo.e0 = lo.e1 = lo.e2 = 0;
        Span sp; sp.end = start; sp.n_code = 0; sp.n_sym = 0; sp.flags = 0;
        bool need = true;                       // this lane has to walk (again): its start has moved
        PROF_CNT(4, 1);
        { PROF_T0();
        __syncthreads();
        stage_input(L, in, b_rel, lane);
        const uint32_t *row = &L.stage[lane * STAGE_DW]; const uint32_t w_first = (b_rel >> 5) + lane * SPAN_DW;
        for (uint32_t it = 0;; it++) {
            PROF_CNT(5, 1);
            if (need) {
                lo.n = 0;
                sp = walk_rows(L, row, w_first, start, stop, LCAP, lo);
                lo.flush();
            }
            const uint32_t prev_end = __shfl_up(sp.end, 1), prev_flags = __shfl_up(sp.flags, 1);
            need = lane > 0 && !(prev_flags & (SP_EOB | SP_ERR)) && prev_end != start;
            if (!__any(need) || it == MAX_REWALK) break;       // (a lane that still needs a walk keeps its old start and list: it is not part of the chain below)
            if (need) start = prev_end;
        }
        PROF_ADD(2); }
        // the chain: lanes 0 .. V - 1 each start where the lane before stopped
        const uint32_t prev_end = __shfl_up(sp.end, 1), prev_flags = __shfl_up(sp.flags, 1);
        const bool good = lane == 0 || (!(prev_flags & (SP_EOB | SP_ERR)) && prev_end == start);
        const uint64_t bad = ~__ballot(good);
        const uint32_t V = bad ? (uint32_t)__ffsll((unsigned long long)bad) - 1 : 64;
        if (__any(lane < V && (sp.flags & SP_ERR))) return B_ERROR;
        const uint32_t end_rel = (uint32_t)__shfl(sp.end, (int)V - 1), end_flags = (uint32_t)__shfl(sp.flags, (int)V - 1);
        if (origin_bits + end_rel > size_bits) return B_PAST_SIZE;
        if (origin_bits + end_rel > limit_bits) return B_PAST_LIMIT;
        uint32_t tot = lane < V ? sp.n_sym : 0;
        for (int d = 32; d; d >>= 1) tot += __shfl_xor(tot, d);
        if (opos + tot + 8 > sym_cap) return B_OVERFLOW;      // (+ 8: the expansion reads earlier output sixteen bytes at a time)
        { PROF_T0();
        // the lists of the chain's lanes, one behind the other (every lane moves its own: eight entries in flight)
        const uint32_t n_mine = lane < V ? sp.n_code : 0;
        uint32_t cinc = n_mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(cinc, d); if (lane >= (uint32_t)d) cinc += u; }
        const uint32_t n_list = (uint32_t)__shfl(cinc, 63);
        {
            const uint32_t *src = lst + lane * LCAP; uint32_t *dst = cl + (cinc - n_mine);
            for (uint32_t k = 0; k < n_mine; k += 8) {
                const uint4 a = *reinterpret_cast<const uint4 *>(src + k), b = *reinterpret_cast<const uint4 *>(src + k + 4);      // (inside the lane's LCAP entries: LCAP is a multiple of eight)
                const uint32_t e[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                for (uint32_t q = 0; q < 8; q++) if (k + q < n_mine) dst[k + q] = e[q];
            }
        }
        __threadfence_block();                   // the list is read by other lanes than the ones that wrote it
        __syncthreads();
        for (uint32_t e0 = 0; e0 < n_list;) {
            uint32_t taken = 0;
            opos += expand4(L, cl + e0, n_list - e0 < RN ? n_list - e0 : RN, out, opos, lane, taken, prof);
            e0 += taken;
            PROF_CNT(6, 1);
        }
        PROF_ADD(3); }
        b_rel = end_rel;
        if (end_flags & SP_EOB) { from = origin_bits + end_rel; return B_EOB; }
    }
}

__global__ __launch_bounds__(64) void gz_decode2_kernel(const uint8_t *data, uint64_t ring_mask, uint64_t size, uint64_t limit_bytes, uint64_t base_byte,
                                                        uint64_t chunk_bytes, uint32_t chunk_lo, uint32_t exact_chunk, uint64_t exact_bit, uint16_t *sym,
                                                        uint64_t sym_cap, GzChunk *chunks, uint32_t *lists)
{
    __shared__ Lds2 L;
    const uint32_t lane = threadIdx.x;
    const uint32_t c = chunk_lo + blockIdx.x;
    uint16_t *out = sym + (uint64_t)blockIdx.x * sym_cap;
    uint32_t *lst = lists + (uint64_t)blockIdx.x * (2 * 64 * LCAP);
    const uint64_t size_bits = size * 8, limit_bits = limit_bytes * 8;
    const uint64_t nominal = (base_byte + (uint64_t)c * chunk_bytes) * 8, stop_bit = nominal + chunk_bytes * 8;
    const uint64_t search_end = stop_bit < size_bits ? stop_bit : size_bits;
    GzChunk res; res.start_bit = 0; res.end_bit = 0; res.n_sym = 0; res.status = GZ_NONE;
    BitRd rd; rd.base = reinterpret_cast<const uint4 *>(data); rd.vmax = (size + 48) / 16;      // readable (and zero) up to size + 64
    rd.vmask = ring_mask >> 4;
    rd.origin_v = (c == exact_chunk && exact_bit < nominal ? exact_bit : nominal) >> 7;
    rd.ring = L.ring; rd.rd_dw = 0; rd.ring_hi = 0; rd.nd = 0; rd.bb = 0; rd.bc = 0;
    LaneIn in; in.words = reinterpret_cast<const uint32_t *>(data); in.origin_dw = rd.origin_v * 4; in.wmask = ring_mask >> 2; in.max_dw = rd.vmax * 4 + 3;
    const uint64_t origin_bits = rd.origin_v * 128;
    bool searching = c != exact_chunk;
    Search2 S; S.pos = nominal;
    Vetted V; V.cand = ~0ull; V.vmask = 0; V.dry = false;
    uint32_t *const cand1 = L.stg, *const cand2 = L.stg + 128;          // the search's lists: that part of LDS is the decoder's only once a start has been found
    uint64_t start = exact_bit, opos = 0, blk_pos = exact_bit, blk_opos = 0;
    uint32_t status = GZ_NONE;
    if (!searching) rd.seek(start, lane);
    else if (nominal >= size_bits) { if (lane == 0) chunks[c] = res; return; }
    uint64_t behind_from = ~0ull;                   // the first block boundary behind the chunk's own range
    uint32_t why = 0;                               // what stopped a decode that failed (reported in n_sym of a GZ_FAILED chunk)
    Prof prof;
#ifdef GZ_PROFILE
    for (int i = 0; i < 16; i++) prof.t[i] = 0;
    const unsigned long long prof_begin = __builtin_readcyclecounter();
#endif
    for (;;) {
        uint32_t final = 0, type = 2;
        bool strict = false;
        why = 0;
        PROF_T0();
        if (searching) {
            start = next_vetted2(S, V, cand1, cand2, reinterpret_cast<const uint32_t *>(data), ring_mask >> 2, in.max_dw, search_end, size_bits, lane);
            if (start == ~0ull) { status = GZ_NONE; break; }
            rd.seek(start + 3, lane);
            strict = true; opos = 0; blk_pos = start; blk_opos = 0;
        } else {
            blk_pos = rd.bitpos(); blk_opos = opos;
            // (the rules of gz_decode_kernel: behind its range a chunk stops in front of a non-final dynamic block; stored and fixed
            // blocks and the member's final block are decoded here too, for at most one chunk's worth of input behind the range)
            const bool behind = blk_pos >= stop_bit;
            if (behind && behind_from == ~0ull) behind_from = blk_pos;
            if (behind && (blk_pos + 3 > size_bits || blk_pos - behind_from >= chunk_bytes * 8)) { status = GZ_AT_BOUNDARY; break; }
            if (blk_pos + 3 > size_bits) why = 1;
            else {
                rd.top_up(lane);
                rd.refill(); final = rd.peek(1); type = rd.peek(3) >> 1;
                if (behind && !final && type == 2) { status = GZ_AT_BOUNDARY; break; }
                rd.drop(3);
                if (type == 3) why = 2;
            }
            if (!why && type == 0) {               // stored block: byte aligned LEN, ~LEN, then the bytes
                rd.drop(rd.bc & 7); rd.refill();
                const uint32_t len = rd.peek(16); rd.drop(16); rd.refill();
                const uint32_t nlen = rd.peek(16); rd.drop(16);
                const uint64_t byte = rd.bitpos() >> 3;
                if ((len ^ nlen) != 0xFFFFu || byte + len > size) why = 3;
                else if (byte + len > limit_bytes) why = 8;
                else {
                    if (opos + len > sym_cap) { status = GZ_OVERFLOW; break; }
                    for (uint32_t i = lane; i < len; i += 64) out[opos + i] = data[(byte + i) & ring_mask];
                    opos += len;
                    rd.seek((byte + len) * 8, lane);
                    if (final) { status = GZ_MEMBER_END; break; }
                    continue;
                }
            }
        }
        if (!why) {
            uint32_t hlit = 288, hdist = 32;
            bool ok = true;
            if (type == 1) {
                for (uint32_t i = lane; i < 320; i += 64) L.lens[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : (i < 288 ? 8 : 5)));
                __syncthreads();
            } else ok = read_code_lengths(L, rd, strict, lane, hlit, hdist);
            PROF_ADD(0);
            if (!ok) why = 4;
            else { PROF_T0(); if (!build_tables(L, hlit, hdist, strict, lane)) why = 5; PROF_ADD(1); }
            __syncthreads();
        } else PROF_ADD(0);
        if (!why) {
            searching = false;
            PROF_CNT(7, 1);
            uint64_t at = rd.bitpos();
            const uint32_t e = decode_block(L, in, origin_bits, at, size_bits, limit_bits, out, opos, sym_cap, lst, lane, prof);
            __syncthreads();
            rd.ring_hi = 0; rd.rd_dw = 0;            // (the ring's bytes were the expansion's in between: whoever seeks next loads it again)
            if (e == B_OVERFLOW) { status = GZ_OVERFLOW; break; }
            if (e == B_ERROR) why = 6; else if (e == B_PAST_SIZE) why = 7; else if (e == B_PAST_LIMIT) why = 8;
            if (!why) rd.seek(at, lane);
        }
        if (why) {
            // not deflate data.  From a speculative start that only says the candidate was false: the search goes on behind it
            // (from the bit behind the failed start: the search keeps no queue of its own)
            if (c != exact_chunk && why != 8) { searching = true; S.pos = start + 1; V.vmask = 0; V.dry = S.pos >= search_end; continue; }
            status = GZ_FAILED; break;
        }
        if (final) { status = GZ_MEMBER_END; break; }
    }
    if (lane == 0) {
        res.start_bit = start;
        if (status == GZ_OVERFLOW) { res.end_bit = blk_pos; res.n_sym = (uint32_t)blk_opos; }
        else if (status == GZ_AT_BOUNDARY) { res.end_bit = blk_pos; res.n_sym = (uint32_t)opos; }
        else { res.end_bit = rd.bitpos(); res.n_sym = (uint32_t)opos; }
        if (status == GZ_NONE) res.n_sym = 0;
        if (status == GZ_FAILED) res.n_sym = why;
        res.status = status;
        chunks[c] = res;
#ifdef GZ_PROFILE
        unsigned long long *pp = reinterpret_cast<unsigned long long *>(lst);
        for (int i = 0; i < 16; i++) pp[i] = prof.t[i];
        pp[16] = __builtin_readcyclecounter() - prof_begin;
#endif
    }
}

// ---------------------------------------------------------------------------------------------------------- link
// Which chunks are accepted, and where their text goes, is decided on the HOST (gz_link_walk below: a walk over the slab's 24-byte
// descriptors).  What is left for the device is the one true data dependency of the scheme: a chunk's markers point into the 32 KiB
// of text in front of it, i.e. into the resolved TAIL of the chunk before -- whose markers point into the tail before that.  Round 4
// walked that chain on one workgroup, a chunk at a time (3.3-4.5 us each, 0.07 s of configs[4]).  Now it is a two-level scan:
//   1. gz_link_tails_kernel, a workgroup per GROUP of consecutive accepted chunks, all groups side by side: the group's chunks in order,
//      against a SYMBOLIC window in LDS (16 bits an entry: a byte, or MARK | index into the window the group began with); the tails
//      are rewritten in place to that form and the window the group ends with is written out;
//   2. gz_link_groups_kernel, one workgroup: the groups in order against the real window (bytes, LDS) -- the only serial part, one step
//      per GROUP instead of one per chunk; leaves the window every group begins with, and the window behind the last chunk;
//   3. gz_link_text_kernel, all tails side by side: the rewritten tails become text through their group's window.
// Everything else of a chunk (its body) is gz_resolve_kernel's, which finds its windows in the text as before.
constexpr uint32_t LINK_THREADS = 1024, LINK_PER = GZ_WINDOW / LINK_THREADS;          // 32 entries of a window per thread

__global__ __launch_bounds__(LINK_THREADS) void gz_link_tails_kernel(const uint32_t *acc, uint32_t n_acc, uint32_t group, const GzChunk *chunks, uint32_t chunk_lo,
                                                                     uint16_t *sym, uint64_t sym_cap, uint16_t *gfinal)
{
    __shared__ uint16_t win[GZ_WINDOW];
    const uint32_t tid = threadIdx.x, g = blockIdx.x;
    const uint32_t k0 = g * group, k1 = k0 + group < n_acc ? k0 + group : n_acc;
    for (uint32_t i = tid; i < GZ_WINDOW; i += LINK_THREADS) win[i] = (uint16_t)(GZ_MARK | i);
    __syncthreads();
    for (uint32_t k = k0; k < k1; k++) {
        const uint32_t c = acc[k];
        const uint32_t n = chunks[c].n_sym, tail = n < GZ_WINDOW ? n : GZ_WINDOW, keep = GZ_WINDOW - tail;
        uint16_t *sp = sym + (uint64_t)(c - chunk_lo) * sym_cap + (n - tail);
        // read: this thread's share of the tail, looked up in the window as it is; and (a chunk shorter than the window) its share of
        // what slides -- all reads of the window first, then a barrier, then the writes: one window in LDS is enough
        uint16_t t[LINK_PER], sl[LINK_PER];
#pragma unroll
        for (uint32_t j = 0; j < LINK_PER / 8; j++) {
            const uint32_t i = tid * 8 + j * (LINK_THREADS * 8);
            uint16_t v[8];
            if (i + 8 <= tail) __builtin_memcpy(v, sp + i, 16);
            else {
#pragma unroll
                for (uint32_t q = 0; q < 8; q++) v[q] = i + q < tail ? sp[i + q] : (uint16_t)0;          // (the tail's last piece: never past the chunk's symbols; constant indices, so v stays in registers)
            }
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) t[j * 8 + q] = (v[q] & GZ_MARK) ? win[v[q] & 0x7FFFu] : v[q];
        }
        if (keep) {
#pragma unroll
            for (uint32_t j = 0; j < LINK_PER / 8; j++)
#pragma unroll
                for (uint32_t q = 0; q < 8; q++) { const uint32_t i = tid * 8 + j * (LINK_THREADS * 8) + q; sl[j * 8 + q] = i < keep ? win[i + tail] : 0; }
        }
        __syncthreads();
        if (keep) {
#pragma unroll
            for (uint32_t j = 0; j < LINK_PER / 8; j++)
#pragma unroll
                for (uint32_t q = 0; q < 8; q++) { const uint32_t i = tid * 8 + j * (LINK_THREADS * 8) + q; if (i < keep) win[i] = sl[j * 8 + q]; }
        }
#pragma unroll
        for (uint32_t j = 0; j < LINK_PER / 8; j++) {
            const uint32_t i = tid * 8 + j * (LINK_THREADS * 8);
            if (i >= tail) continue;
            if (i + 8 <= tail) {
                __builtin_memcpy(sp + i, &t[j * 8], 16);
                if (((keep + i) & 7) == 0) __builtin_memcpy(&win[keep + i], &t[j * 8], 16);
                else {
#pragma unroll
                    for (uint32_t q = 0; q < 8; q++) win[keep + i + q] = t[j * 8 + q];
                }
            } else {
#pragma unroll
                for (uint32_t q = 0; q < 8; q++) if (i + q < tail) { sp[i + q] = t[j * 8 + q]; win[keep + i + q] = t[j * 8 + q]; }
            }
        }
        __syncthreads();
    }
    uint16_t *out = gfinal + (uint64_t)g * GZ_WINDOW;
    for (uint32_t i = tid * 8; i < GZ_WINDOW; i += LINK_THREADS * 8) *reinterpret_cast<uint4 *>(out + i) = *reinterpret_cast<const uint4 *>(&win[i]);
}

// window: the last 32 KiB of text in front of the first accepted chunk (wlen valid bytes, right-aligned) -> behind the last one.
// gwin[g]: the window group g begins with.  The window in front of the first chunk is text too (a body's markers are looked up
// in the text): written to text_front, which ends where the first accepted chunk's text begins.
__global__ __launch_bounds__(LINK_THREADS) void gz_link_groups_kernel(uint8_t *window, uint32_t wlen, const uint16_t *gfinal, uint32_t n_groups, uint8_t *gwin, uint8_t *text_front)
{
    __shared__ uint8_t R[GZ_WINDOW];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid * 16; i < GZ_WINDOW; i += LINK_THREADS * 16) *reinterpret_cast<uint4 *>(&R[i]) = *reinterpret_cast<const uint4 *>(&window[i]);
    __syncthreads();
    for (uint32_t i = tid; i < wlen; i += LINK_THREADS) text_front[(int64_t)i - (int64_t)wlen] = R[GZ_WINDOW - wlen + i];
    for (uint32_t g = 0; g < n_groups; g++) {
        uint8_t *gw = gwin + (uint64_t)g * GZ_WINDOW;
        const uint16_t *F = gfinal + (uint64_t)g * GZ_WINDOW;
        uint8_t b[LINK_PER];
#pragma unroll
        for (uint32_t j = 0; j < LINK_PER / 8; j++) {
            const uint32_t i = tid * 8 + j * (LINK_THREADS * 8);
            uint16_t v[8];
            __builtin_memcpy(v, F + i, 16);
            uint64_t r; __builtin_memcpy(&r, &R[i], 8);
            __builtin_memcpy(gw + i, &r, 8);                              // the window this group begins with
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) b[j * 8 + q] = (v[q] & GZ_MARK) ? R[v[q] & 0x7FFFu] : (uint8_t)v[q];
        }
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < LINK_PER / 8; j++) __builtin_memcpy(&R[tid * 8 + j * (LINK_THREADS * 8)], &b[j * 8], 8);
        __syncthreads();
    }
    for (uint32_t i = tid * 16; i < GZ_WINDOW; i += LINK_THREADS * 16) *reinterpret_cast<uint4 *>(&window[i]) = *reinterpret_cast<const uint4 *>(&R[i]);
}

constexpr uint32_t RESOLVE_SEG = 8192;            // symbols per workgroup
__global__ __launch_bounds__(256) void gz_link_text_kernel(const uint32_t *acc, const uint64_t *acc_off, uint32_t group, const GzChunk *chunks, uint32_t chunk_lo,
                                                           const uint16_t *sym, uint64_t sym_cap, const uint8_t *gwin, uint8_t *text, uint64_t text_base)
{
    const uint32_t k = blockIdx.y, c = acc[k];
    const uint32_t n = chunks[c].n_sym, tail = n < GZ_WINDOW ? n : GZ_WINDOW;
    const uint32_t s0 = blockIdx.x * RESOLVE_SEG;
    if (s0 >= tail) return;
    const uint32_t s1 = s0 + RESOLVE_SEG < tail ? s0 + RESOLVE_SEG : tail;
    const uint16_t *sp = sym + (uint64_t)(c - chunk_lo) * sym_cap + (n - tail);
    const uint8_t *wp = gwin + (uint64_t)(k / group) * GZ_WINDOW;
    uint8_t *tp = text + (int64_t)(acc_off[k] - text_base) + (n - tail);
    for (uint32_t i = s0 + threadIdx.x * 8; i < s1; i += 256 * 8) {
        const uint32_t m = s1 - i < 8 ? s1 - i : 8;
        uint16_t v[8];
        if (m == 8) __builtin_memcpy(v, sp + i, 16);
        else {
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) v[q] = q < m ? sp[i + q] : (uint16_t)0;
        }
        uint8_t b[8];
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) b[q] = (v[q] & GZ_MARK) ? wp[v[q] & 0x7FFFu] : (uint8_t)v[q];
        if (m == 8) __builtin_memcpy(tp + i, b, 8);
        else {
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) if (q < m) tp[i + q] = b[q];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------- resolve
// the body of every accepted chunk (everything in front of its last 32 Ki symbols): markers are looked up in the text in front of the chunk
__global__ __launch_bounds__(256) void gz_resolve_kernel(const uint32_t *acc, const uint64_t *acc_off, const GzChunk *chunks, uint32_t chunk_lo, const uint16_t *sym,
                                                         uint64_t sym_cap, uint8_t *text, uint64_t text_base)
{
    const uint32_t c = acc[blockIdx.y];
    const uint64_t off = acc_off[blockIdx.y];
    const uint32_t n = chunks[c].n_sym;
    const uint32_t body = n > GZ_WINDOW ? n - GZ_WINDOW : 0;      // the link step has written the rest
    const uint32_t s0 = blockIdx.x * RESOLVE_SEG;
    if (s0 >= body) return;
    const uint32_t s1 = s0 + RESOLVE_SEG < body ? s0 + RESOLVE_SEG : body;
    const uint16_t *sp = sym + (uint64_t)(c - chunk_lo) * sym_cap;
    uint8_t *tp = text + (int64_t)(off - text_base);
    const uint8_t *wp = tp - GZ_WINDOW;                            // the 32 KiB in front of the chunk
    for (uint32_t i = s0 + threadIdx.x * 8; i < s1; i += 256 * 8) {
        const uint32_t m = s1 - i < 8 ? s1 - i : 8;
        uint16_t v[8];
        if (m == 8) __builtin_memcpy(v, sp + i, 16);
        else {
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) v[k] = k < m ? sp[i + k] : (uint16_t)0;
        }
        uint8_t b[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) b[k] = (v[k] & GZ_MARK) ? wp[v[k] & 0x7FFFu] : (uint8_t)v[k];          // (a marker of the last, partial group would read the window at index 0: readable)
        if (m == 8) __builtin_memcpy(tp + i, b, 8);
        else {
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) if (k < m) tp[i + k] = b[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------- CRC-32
// The gzip polynomial, reflected.  A thread takes 256 bytes with four table lookups per dword (tables built in LDS by the
// workgroup), then moves its remainder to the end of the 64 KiB piece -- multiplication by x^(8 * bytes behind it) modulo the
// polynomial, 32 shift-and-add steps -- and the workgroup XORs the 256 results: CRCs are linear once the register starts at 0.
constexpr uint32_t CRC_POLY = 0xEDB88320u, CRC_RUN = 256;
__host__ __device__ inline uint32_t crc_mulmod(uint32_t a, uint32_t b)       // a * b mod P; bit 31 is the coefficient of x^0
{
    uint32_t p = 0;
    for (uint32_t m = 0x80000000u; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
__host__ __device__ inline uint32_t crc_xpow8(uint64_t n_bytes)              // x^(8 * n_bytes) mod P
{
    uint32_t r = 0x80000000u, sq = 0x00800000u;                               // x^0, x^8
    for (; n_bytes; n_bytes >>= 1) { if (n_bytes & 1) r = crc_mulmod(r, sq); sq = crc_mulmod(sq, sq); }
    return r;
}
__global__ __launch_bounds__(256) void gz_crc_kernel(const uint8_t *text, uint64_t n, uint32_t *piece, uint32_t run_mult)
{
    __shared__ uint32_t T[4][256];
    __shared__ uint32_t part[4];
    const uint32_t tid = threadIdx.x;
    {
        uint32_t r = tid;
        for (int k = 0; k < 8; k++) r = (r & 1u) ? (r >> 1) ^ CRC_POLY : r >> 1;
        T[0][tid] = r;
    }
    __syncthreads();
    for (int j = 1; j < 4; j++) { const uint32_t p = T[j - 1][tid]; T[j][tid] = T[0][p & 0xFFu] ^ (p >> 8); __syncthreads(); }
    const uint64_t p0 = (uint64_t)blockIdx.x * GZ_CRC_PIECE, pn = n - p0 < GZ_CRC_PIECE ? n - p0 : GZ_CRC_PIECE;
    const uint64_t b0 = (uint64_t)tid * CRC_RUN;
    uint32_t r = 0;
    if (b0 < pn) {
        const uint8_t *q = text + p0 + b0;
        uint32_t len = pn - b0 < CRC_RUN ? (uint32_t)(pn - b0) : CRC_RUN;
        while (len && ((size_t)q & 3)) { r = T[0][(r ^ *q++) & 0xFFu] ^ (r >> 8); len--; }
        for (; len >= 4; len -= 4, q += 4) {
            r ^= *reinterpret_cast<const uint32_t *>(q);
            r = T[3][r & 0xFFu] ^ T[2][(r >> 8) & 0xFFu] ^ T[1][(r >> 16) & 0xFFu] ^ T[0][r >> 24];
        }
        while (len) { r = T[0][(r ^ *q++) & 0xFFu] ^ (r >> 8); len--; }
        // bytes of the piece behind this thread's run
        const uint64_t behind = pn - (b0 + CRC_RUN < pn ? b0 + CRC_RUN : pn);
        if (pn == GZ_CRC_PIECE) {                                    // whole piece: x^(8 * 256 * (255 - tid)) by repeated multiplication with x^(8 * 256)
            uint32_t mlt = 0x80000000u, sq = run_mult;
            for (uint32_t e = 255 - tid; e; e >>= 1) { if (e & 1) mlt = crc_mulmod(mlt, sq); sq = crc_mulmod(sq, sq); }
            r = crc_mulmod(r, mlt);
        } else r = crc_mulmod(r, crc_xpow8(behind));
    }
    for (int d = 32; d; d >>= 1) r ^= __shfl_xor(r, d);
    if ((tid & 63) == 0) part[tid >> 6] = r;
    __syncthreads();
    if (tid == 0) piece[blockIdx.x] = part[0] ^ part[1] ^ part[2] ^ part[3];
}

} // namespace

// scratch for the lane-parallel decoder: 64 lanes x LCAP list entries per chunk of a launch
size_t gz_decode_scratch_bytes(uint32_t n_chunks) { return (size_t)n_chunks * 2 * 64 * LCAP * sizeof(uint32_t); }
bool gz_decode_serial()       // MF_GZDEV_KERNEL=serial: the one-lane walk of round 3 (A/B runs); default: the lane-parallel kernel
{
    static const bool serial = [] { const char *v = getenv("MF_GZDEV_KERNEL"); return v && strcmp(v, "serial") == 0; }();
    return serial;
}

hipError_t launch_gz_decode(const uint8_t *d_data, uint64_t ring_bytes, uint64_t size, uint64_t limit_bytes, uint64_t base_byte, uint64_t chunk_bytes,
                            uint32_t chunk_lo, uint32_t n_chunks, uint32_t exact_chunk, uint64_t exact_bit, uint16_t *d_sym, uint64_t sym_cap,
                            GzChunk *d_chunks, uint32_t *d_scratch, hipStream_t st)
{
    if (!n_chunks) return hipSuccess;
    if (ring_bytes && ((ring_bytes & (ring_bytes - 1)) || ring_bytes < 4096)) return hipErrorInvalidValue;
    const uint64_t ring_mask = ring_bytes ? ring_bytes - 1 : ~0ull;
    if (gz_decode_serial() || !d_scratch)
        hipLaunchKernelGGL(gz_decode_kernel, dim3(n_chunks), dim3(64), 0, st, d_data, ring_mask, size, limit_bytes < size ? limit_bytes : size, base_byte, chunk_bytes,
                           chunk_lo, exact_chunk, exact_bit, d_sym, sym_cap, d_chunks);
    else
        hipLaunchKernelGGL(gz_decode2_kernel, dim3(n_chunks), dim3(64), 0, st, d_data, ring_mask, size, limit_bytes < size ? limit_bytes : size, base_byte, chunk_bytes,
                           chunk_lo, exact_chunk, exact_bit, d_sym, sym_cap, d_chunks, d_scratch);
    return hipGetLastError();
}

hipError_t launch_gz_link(const uint32_t *d_acc, const uint64_t *d_acc_off, uint32_t n_acc, uint32_t max_sym, const GzChunk *d_chunks, uint32_t chunk_lo, uint16_t *d_sym,
                          uint64_t sym_cap, uint8_t *d_window, uint32_t wlen_before, uint8_t *d_scratch, uint8_t *d_text, uint64_t text_base, uint64_t first_off, hipStream_t st)
{
    if (!n_acc) return hipSuccess;
    const uint32_t group = gz_link_group(n_acc), n_groups = (n_acc + group - 1) / group;
    uint16_t *gfinal = reinterpret_cast<uint16_t *>(d_scratch);
    uint8_t *gwin = d_scratch + (size_t)n_groups * GZ_WINDOW * 2;
    hipLaunchKernelGGL(gz_link_tails_kernel, dim3(n_groups), dim3(LINK_THREADS), 0, st, d_acc, n_acc, group, d_chunks, chunk_lo, d_sym, sym_cap, gfinal);
    hipLaunchKernelGGL(gz_link_groups_kernel, dim3(1), dim3(LINK_THREADS), 0, st, d_window, wlen_before, gfinal, n_groups, gwin, d_text + (int64_t)(first_off - text_base));
    hipLaunchKernelGGL(gz_link_text_kernel, dim3(GZ_WINDOW / RESOLVE_SEG, n_acc), dim3(256), 0, st, d_acc, d_acc_off, group, d_chunks, chunk_lo, d_sym, sym_cap, gwin, d_text, text_base);
    return hipGetLastError();
}

hipError_t launch_gz_resolve(const uint32_t *d_acc, const uint64_t *d_acc_off, uint32_t n_acc, uint32_t max_sym, const GzChunk *d_chunks, uint32_t chunk_lo, const uint16_t *d_sym,
                             uint64_t sym_cap, uint8_t *d_text, uint64_t text_base, hipStream_t st)
{
    if (!n_acc || max_sym <= GZ_WINDOW) return hipSuccess;
    hipLaunchKernelGGL(gz_resolve_kernel, dim3((max_sym - GZ_WINDOW + RESOLVE_SEG - 1) / RESOLVE_SEG, n_acc), dim3(256), 0, st, d_acc, d_acc_off, d_chunks, chunk_lo, d_sym, sym_cap, d_text, text_base);
    return hipGetLastError();
}

// The runtime loads a translation unit's code object when the first of its kernels is asked for (20-50 ms): asking for the attributes of one
// does that too, so a thread can get it out of the way while a cold call is still reading its first bytes (ingest_prefetch, mf_devingest.cpp)
void gz_preload() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&gz_decode2_kernel)); (void)hipGetLastError(); }

hipError_t launch_gz_crc(const uint8_t *d_text, uint64_t n, uint32_t *d_piece, hipStream_t st)
{
    if (!n) return hipSuccess;
    static const uint32_t run_mult = crc_xpow8(CRC_RUN);
    hipLaunchKernelGGL(gz_crc_kernel, dim3((uint32_t)((n + GZ_CRC_PIECE - 1) / GZ_CRC_PIECE)), dim3(256), 0, st, d_text, n, d_piece, run_mult);
    return hipGetLastError();
}

// crc32(A || B) = crc32(A) * x^(8 |B|) + crc32(B): the pre- and post-conditioning of the two values cancel out
uint32_t gz_crc_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) { return crc_mulmod(crc_a, crc_xpow8(len_b)) ^ crc_b; }

uint32_t gz_crc_finish(const uint32_t *piece, uint64_t n)
{
    // pure remainders of the pieces -> pure remainder of the text (Horner), then zlib's conditioning: a register that starts at
    // all ones and is inverted at the end adds the CRC of n zero bytes, which is a function of n alone
    if (!n) return 0;
    const uint64_t np = (n + GZ_CRC_PIECE - 1) / GZ_CRC_PIECE, last = n - (np - 1) * GZ_CRC_PIECE;
    const uint32_t full = crc_xpow8(GZ_CRC_PIECE);
    uint32_t r = 0;
    for (uint64_t i = 0; i + 1 < np; i++) r = (i ? crc_mulmod(r, full) : 0u) ^ piece[i];
    r = (np > 1 ? crc_mulmod(r, crc_xpow8(last)) : 0u) ^ piece[np - 1];
    // crc32 of n zero bytes: the all-ones start value moved n bytes along, inverted
    const uint32_t zeros = crc_mulmod(0xFFFFFFFFu, crc_xpow8(n)) ^ 0xFFFFFFFFu;
    return r ^ zeros;
}

} // namespace mf
