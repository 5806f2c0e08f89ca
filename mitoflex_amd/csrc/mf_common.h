// Shared definitions of libmitofilter_hip: hashing, table layout rules and the
// plain-data views handed to kernels.  Host and device code include this.
//
// Semantics (the reference has no counterpart, see
// DESIGN.md "Spec B"): 2-bit little-endian packing, canonical k-mer =
// min(fwd, revcomp) with the first base least significant, exact membership in
// an open-address table whose layout is history independent.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MF_HD __host__ __device__ __forceinline__
#else
#define MF_HD inline
#endif

namespace mf {

constexpr uint64_t EMPTY64 = ~0ULL;
constexpr uint32_t EMPTY32 = ~0u;

// ---- hashing (identical on host and device; restated by the test oracle) ----
MF_HD uint64_t mix64(uint64_t x)
{   // MurmurHash3 fmix64
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}
// Table hash: two 32-bit multiplies per 64-bit word (cheap on the GPU's 32-bit
// integer ALUs; a 64-bit multiply costs four), folded and finished with a
// xor-shift so the low bits (the slot index) depend on every input bit.
MF_HD uint32_t fold32(uint64_t x)
{
    const uint32_t a = (uint32_t)x * 0x9E3779B1u, b = (uint32_t)(x >> 32) * 0x85EBCA77u;
    uint32_t h = a ^ ((b << 15) | (b >> 17));
    return h ^ (h >> 15);
}
MF_HD uint64_t hash_key1(uint64_t lo) { return fold32(lo); }
MF_HD uint64_t hash_key2(uint64_t lo, uint64_t hi)
{
    uint32_t h = fold32(lo) ^ (fold32(hi) * 0xC2B2AE3Du);
    return h ^ (h >> 16);
}

// screen hashes (implementation detail of the s-mer screen; not part of the
// result semantics -- the screen is conservative for any choice)
//   stage 1: blocked bit table, one 128-bit block per s-mer (both strands
//            inserted), one bit in each of the block's four dwords
//   stage 2: classic Bloom filter over canonical s-mers, STAGE2_K probes
//   stage 3: exact ordered s-mer table in global memory (L2 resident)
// stage-1 hash: (low 24 bits of the s-mer) * 0x9E3779 + s-mer -- one full-rate v_mad_u32_u24 (a full 32-bit multiply is a
// quarter-rate instruction on CDNA).  The block index is the top bits of h.
MF_HD uint32_t bloom_hash(uint32_t smer) { return (smer & 0xFFFFFFu) * 0x9E3779u + smer; }
// stage-1 bit of dword i of the 128-bit block: 31 - field_i.  The fields are the low five bits of bytes 0, 1, 2 of the s-mer
// itself and of byte 1 of h -- every one a byte of a value the kernel already holds, so a field costs nothing: the kernel tests
// a bit by shifting the dword LEFT by field_i (the bit lands in the sign position) with the field taken straight from its byte
// through an SDWA operand selector.  Simulated on a random 16.5 kbp bait (32 969 s-mers, both strands, 8192 blocks): 0.052 %
// false positives for 14-base samples, 0.056 % for 16-base ones.  (Rounds 1-2 took the fields from the bytes of (h:smer) >> 11:
// the same 0.05 % for 16-base samples, but 0.27 % for 14-base ones -- byte 2 was smer[31:27], one live bit when s = 14 -- and one
// more instruction per sample.)
// stage-1 block index: idx_bits = log2w - 2 bits of h that end just below bit 2s, i.e. h[2s-1 : 2s-idx_bits] (for s = 16: the top
// bits).  h mod 2^(2s) only depends on the s-mer's own 2s bits even when the value hashed carries the following bases in the
// bits above them (the sum and the product only carry upwards), and the four fields sit below bit 24 <= 2s: so the screen
// hashes a sample AS IT LIES IN THE WORD, without masking it to 2s bits first -- one instruction per sample less for s < 16.
// (Simulated, 14-base samples: 0.065 % false positives against 0.052 % with the top bits.)  s >= 12 (screen_geom_for).
MF_HD uint32_t stage1_index_lo(int s, uint32_t log2w) { const uint32_t ib = log2w - 2; return (uint32_t)(2 * s) >= ib ? (uint32_t)(2 * s) - ib : 0u; }
MF_HD uint32_t stage1_field(uint32_t smer, uint32_t h, int i) { return ((i < 3 ? smer >> (8 * i) : h >> 8)) & 31u; }
MF_HD uint32_t stage1_bit(uint32_t smer, uint32_t h, int i) { return 31u - stage1_field(smer, h, i); }
// one-bit LDS table of mode 4: bit number among 1 << (log2w + 5), taken like a stage-1 block index (the bits of h that end just below bit 2s)
MF_HD uint32_t pre_index_lo(int s, uint32_t log2w) { const uint32_t ib = log2w + 5; return (uint32_t)(2 * s) >= ib ? (uint32_t)(2 * s) - ib : 0u; }
// the grid / record-list geometry a set's screen uses: the HBM-bound stride-16 screen leaves one CU in eight free (screen_grid_for);
// every other screen -- stride 8, and the gather- and issue-bound screens of large baits -- takes every CU
// (key 4: two workgroups a CU -- the front2-only screen of large baits holds few registers and no LDS table, and twice the gathers in flight)
// (mode 3 with stride 16 is screen_kernel's loop and keeps its HBM-bound grid)
#ifndef MF_CANON_KEY
#define MF_CANON_KEY 16
#endif
MF_HD int screen_grid_key(int stride, uint32_t front_mode, uint32_t canon = 0) { return front_mode == 2 ? 4 : (stride == 16 && (front_mode == 0 || front_mode == 3)) ? (canon ? MF_CANON_KEY : 16) : 8; }
// front2 at most 2 MiB: an XCD's L2 is 4 MiB and the read stream passes through it too -- a 4 MiB table is looked up at 150-180 G/s,
// a 2 MiB one at 205 (profiles/r06/c_front_variants.txt; the part's roof, nothing else running, is 265 G/s: tools/gather_roof.hip)
constexpr uint32_t FRONT2_MAX_LOG2B = 17;
constexpr int STAGE2_K = 4;
MF_HD uint32_t stage2_hash_a(uint32_t canon) { uint32_t h = canon * 0x85EBCA6Bu; return h ^ (h >> 13); }
MF_HD uint32_t stage2_hash_b(uint32_t canon) { uint32_t h = canon * 0xC2B2AE35u; return (h ^ (h >> 16)) | 1u; }
// k-mer bit table in front of the open-address table (exact kernel): one 64-bit block per key, one bit in each
// dword.  Its hash is one 24-bit multiply-add per 32 bits of key (full rate; a 32-bit multiply is a quarter-rate instruction) and xors -- the table hash proper is only computed for the
// positives; the block index comes from the top bits, the two bit positions from the low bits folded with the middle.
MF_HD uint32_t mad24(uint32_t x, uint32_t c) { return (x & 0xFFFFFFu) * c + x; }          // one full-rate v_mad_u32_u24
MF_HD uint32_t rot16(uint32_t x) { return (x << 16) | (x >> 16); }
MF_HD uint32_t kbit_hash1(uint64_t lo) { return mad24((uint32_t)lo, 0x9E3779u) ^ rot16(mad24((uint32_t)(lo >> 32), 0x85EBCBu)); }
MF_HD uint32_t kbit_hash2(uint64_t lo, uint64_t hi) { return kbit_hash1(lo) ^ mad24((uint32_t)hi, 0xC2B2AFu) ^ rot16(mad24((uint32_t)(hi >> 32), 0x27D4EBu)); }
MF_HD uint32_t kbit_pos(uint32_t hb) { return hb ^ (hb >> 16); }   // bit of dword 0: low five bits, of dword 1: the next five
MF_HD uint32_t smer_hash(uint32_t smer) { uint32_t h = smer * 0xC2B2AE35u; return h ^ (h >> 15); }

// ---- table sizing rules (the test oracle restates the same rule) -----------
MF_HD uint64_t table_slots_for(uint64_t n_windows)
{
    uint64_t s = 1024;
    while (s < 2 * n_windows) s <<= 1;
    return s;
}

// ---- screen geometry ---------------------------------------------------------
// Any window of k bases laid anywhere in the dense stream fully contains one
// s-mer that starts at a stream position divisible by `stride`, provided
// k >= s + stride - 1.  stride is 16 bases (one u32 word) or 8.
struct ScreenGeom { int s; int stride; };
MF_HD ScreenGeom screen_geom_for(int k)
{
    if (k >= 31) return {16, 16};
    if (k >= 28) return {k - 15, 16};   // 13..15-base s-mers, still one sample per word: a few more false candidates (an
                                         // s-mer of the bait matches by chance at 4^-s per sample) beat twice the samples;
                                         // measured pass at k = 28 / 29 / 30: 0.33 / 0.30 / 0.29 ms against 0.38 ms with stride 8
                                         // (at k = 27, s = 12, the false candidates cost more than they save: 0.48 ms)
    if (k >= 23) return {16, 8};
    if (k >= 19) return {k - 7, 8};
    return {0, 0};   // too short for a selective screen: exhaustive only
}

// ---- reverse complement of a little-endian 2-bit code ------------------------
MF_HD uint64_t swap_pairs_rev64(uint64_t x)
{   // reverse the order of the 32 two-bit groups
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t y = __brevll(x);
#else
    uint64_t y = x;
    y = ((y >> 1) & 0x5555555555555555ULL) | ((y & 0x5555555555555555ULL) << 1);
    y = ((y >> 2) & 0x3333333333333333ULL) | ((y & 0x3333333333333333ULL) << 2);
    y = ((y >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((y & 0x0F0F0F0F0F0F0F0FULL) << 4);
    y = __builtin_bswap64(y);
#endif
    return ((y >> 1) & 0x5555555555555555ULL) | ((y & 0x5555555555555555ULL) << 1);
}
MF_HD uint32_t swap_pairs_rev32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t y = __brev(x);
#else
    uint32_t y = x;
    y = ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
    y = ((y >> 2) & 0x33333333u) | ((y & 0x33333333u) << 2);
    y = ((y >> 4) & 0x0F0F0F0Fu) | ((y & 0x0F0F0F0Fu) << 4);
    y = __builtin_bswap32(y);
#endif
    return ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
}
// k <= 32
MF_HD uint64_t revcomp1(uint64_t fwd, int k) { return (~swap_pairs_rev64(fwd)) >> (64 - 2 * k); }
// s <= 16
MF_HD uint32_t revcomp_s(uint32_t fwd, int s) { return (~swap_pairs_rev32(fwd)) >> (32 - 2 * s); }
// 33 <= k <= 63: value = hi:lo (128 bit)
MF_HD void revcomp2(uint64_t lo, uint64_t hi, int k, uint64_t &rlo, uint64_t &rhi)
{
    uint64_t ylo = ~swap_pairs_rev64(hi), yhi = ~swap_pairs_rev64(lo);   // 128-bit reversed+complemented
    int sh = 128 - 2 * k;                                                 // 2..62
    rlo = (ylo >> sh) | (yhi << (64 - sh));
    rhi = yhi >> sh;
}

// ---- block index over the offsets of a ragged read set -----------------------
// entry of block b (bases B0 = b << 7 .. B0 + 127):
//   bits  0..30  r0    the read that holds base B0 (the last read that begins at or before it)
//   bits 31..38  back  B0 - (first base of r0), capped at 255
//   bits 39..46  fwd   (first read start at or behind B0 + 128) - (B0 + 128), capped at 255
//   bits 47..48  n     reads that begin inside the block behind B0 (1 .. 127): 0, 1, 2; 3 = more than two (the look-up falls back to a search)
//   bits 49..55  p1    where the first of them begins (offset inside the block)
//   bits 56..62  p2    ... the second
struct OffBlk {
    uint32_t r0, back, fwd, n, p1, p2;
    MF_HD static OffBlk unpack(uint64_t e) { return OffBlk{(uint32_t)(e & 0x7FFFFFFFu), (uint32_t)(e >> 31) & 255u, (uint32_t)(e >> 39) & 255u, (uint32_t)(e >> 47) & 3u, (uint32_t)(e >> 49) & 127u, (uint32_t)(e >> 56) & 127u}; }
    MF_HD uint64_t pack() const { return (uint64_t)r0 | ((uint64_t)back << 31) | ((uint64_t)fwd << 39) | ((uint64_t)n << 47) | ((uint64_t)p1 << 49) | ((uint64_t)p2 << 56); }
};

constexpr int OFF_BLK_SHIFT = 7;           // ragged read sets: one entry of the block index over `offsets` per 128 bases (8 bytes per 32 bytes of stream)

// the entry of block b (what build_off_blk_kernel stores; host code only in tests/native/offblk_check.cpp)
MF_HD OffBlk offblk_make(const uint64_t *offsets, uint64_t n_reads, uint64_t b)
{
    const uint64_t B0 = b << OFF_BLK_SHIFT, B1 = B0 + ((uint64_t)1 << OFF_BLK_SHIFT);
    uint64_t lo = 0, hi = n_reads + 1;                 // first index with offsets[i] > B0 (n_reads + 1: none)
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (offsets[mid] <= B0) lo = mid + 1; else hi = mid; }
    OffBlk e{};
    const uint64_t r0 = lo - 1;                        // (offsets[0] == 0 <= B0: lo >= 1)
    e.r0 = (uint32_t)r0;
    const uint64_t back = B0 - offsets[r0 <= n_reads ? r0 : n_reads];
    e.back = back > 255 ? 255u : (uint32_t)back;
    // the reads that begin inside the block behind its first base, and the first one that begins at or behind its end
    uint32_t n = 0; uint64_t pos[3] = {0, 0, 0}; uint64_t next = ~0ULL;
    for (uint64_t i = r0 + 1; i <= n_reads; i++) {
        const uint64_t o = offsets[i];
        if (o >= B1) { next = o; break; }
        if (n < 3) pos[n] = o - B0;
        n++;
        if (n > 2) break;
    }
    e.n = n > 2 ? 3u : n;
    e.p1 = (uint32_t)pos[0]; e.p2 = (uint32_t)pos[1];
    const uint64_t fwd = next == ~0ULL ? 255 : next - B1;          // (no read start behind the block: the last read's end is offsets[n_reads], the caller's total_bases check)
    e.fwd = fwd > 255 ? 255u : (uint32_t)fwd;
    return e;
}

// The read that holds base g of a ragged set (g + s <= total_bases = offsets[n_reads]; s <= 255), or ~0 if the s bases from g do not lie in one read.
// *start (optional): the read's first base -- exact, or a base at least 255 in front of g's block when the read begins further back than the entry can say;
// *end (optional): the read's end, exact (a read that runs on more than 255 bases behind the block costs a second load).
// ONE 8-byte load where a binary search over all offsets of a 5 Gbp set took twenty-six dependent ones (0.49 ms a pass against 0.22 for uniform reads), and
// an index of read numbers plus the candidates' offsets two round trips to tables far beyond the caches (0.32 ms).
MF_HD uint64_t offblk_lookup(const uint64_t *off_blk, const uint64_t *offsets, uint64_t n_reads, uint64_t g, uint32_t s, uint64_t *start, uint64_t *end)
{
    const uint64_t b = g >> OFF_BLK_SHIFT, B0 = b << OFF_BLK_SHIFT;
    const OffBlk e = OffBlk::unpack(off_blk[b]);
    const uint32_t rel = (uint32_t)(g - B0);
    if (e.n <= 2) {
        const uint32_t cnt = (e.n >= 1 && e.p1 <= rel ? 1u : 0u) + (e.n >= 2 && e.p2 <= rel ? 1u : 0u);
        const uint64_t r = (uint64_t)e.r0 + cnt;
        const uint64_t r_lo = cnt == 0 ? B0 - e.back : B0 + (cnt == 1 ? e.p1 : e.p2);
        uint64_t r_hi = cnt < e.n ? B0 + (cnt == 0 ? e.p1 : e.p2) : B0 + 128 + e.fwd;
        if (end && cnt == e.n && e.fwd == 255u) r_hi = offsets[r + 1];          // (a caller that wants the exact end of a long read)
        if (start) *start = r_lo;
        if (end) *end = r_hi;
        return g + s <= r_hi ? r : ~0ULL;                                          // (s <= 255: a capped end is far enough)
    }
    // more than two reads begin inside the block (reads of a few bases): first offset > g, minus one, searched between this block's and the next one's read
    uint64_t l = (uint64_t)e.r0 + 1, h = (uint64_t)(off_blk[b + 1] & 0x7FFFFFFFu) + 1;
    if (h > n_reads) h = n_reads;                                                  // (g < total_bases = offsets[n_reads]: never beyond the last entry)
    while (l < h) { const uint64_t mid = (l + h) >> 1; if (offsets[mid] <= g) l = mid + 1; else h = mid; }
    const uint64_t o_prev = offsets[l - 1], o_first = offsets[l];
    if (start) *start = o_prev;
    if (end) *end = o_first;
    return g + s <= o_first ? l - 1 : ~0ULL;
}

// ---- plain-data views passed to kernels ------------------------------------
constexpr int NPOS_BLK_SHIFT = 12;
struct ReadsView {
    const uint32_t *words;      // padded with zero words past n_words
    uint64_t        n_words;    // words holding bases
    uint64_t        n_vec;      // uint4 count the screen kernel walks (padded, zero tail)
    const uint64_t *offsets;    // n_reads+1 base offsets, nullptr when uniform_len > 0
    const uint64_t *off_blk;    // ragged sets: one 8-byte entry per block of 128 bases ((total_bases >> OFF_BLK_SHIFT) + 2 entries) that says which reads
                                // the block holds and where they begin (OffBlk below): the read of a base is ONE load, where a binary search over
                                // all offsets of a 5 Gbp set took twenty-six dependent ones
    uint32_t        uniform_len;
    uint64_t        len_magic;  // ceil(2^64 / uniform_len): g / uniform_len == umulhi64(g, len_magic) while g * len < 2^64
    uint32_t        len_magic32;// ceil(2^32 / uniform_len), used for 32-bit offsets when uniform_len <= 4096 (else 0)
    uint64_t        n_reads;
    uint64_t        total_bases;
    const uint64_t *npos;       // sorted invalid base positions
    uint64_t        n_npos;
    const uint32_t *npos_blk;   // npos_blk[b] = first index i with npos[i] >= b << NPOS_BLK_SHIFT (one entry per 4096 bases, plus
                                // two): a lookup is one load and a search over the few entries of a block instead of ~17 dependent loads
    const uint32_t *has_n;      // bit r set: read r holds an invalid base
};

struct KmerSetView {
    int32_t   k, kw;
    uint64_t  slot_mask;        // slots-1
    const uint64_t *keys;       // slots*kw
    // LDS front of the table: blocked bit table over canonical k-mers (128-bit blocks, one bit per dword)
    uint32_t  kb_log2w;         // 1 << kb_log2w words
    const uint32_t *kbloom;
    uint32_t  kb_co_log2w;      // the same table folded down to <= 16 KiB: what an exact kernel stages when it has to share the CU's LDS
    const uint32_t *kbloom_co;  // with a screen workgroup of the next pass (pipelined passes); == kbloom when that is small already
    // screen
    int32_t   s, stride;        // s == 0: disabled
    uint32_t  smask;            // (1 << 2s) - 1
    uint32_t  bloom_log2w;      // stage 1: 1 << bloom_log2w words (= 1 << (bloom_log2w-2) blocks of 128 bit)
    uint32_t  canon;            // != 0: the screen's tables (stage 1, front2, front3, the one-bit table) hold ONE key per bait s-mer -- the smaller of the s-mer and its
                                // reverse complement -- and the screen kernels ask them with the canonical sample: 1 sixteen-base samples (canon16), 2 shorter ones
    uint32_t  s8_finish;        // stride-8 set (k < 28) of a bait beyond ~20 kbp: threshold-1 passes take screen + finish instead of the candidate bitmap
    uint32_t  stage2_log2w;     // stage 2: 1 << stage2_log2w words, stored right behind stage 1
    const uint32_t *bloom;      // (1 << bloom_log2w) + (1 << stage2_log2w) words
    uint32_t  stab_mask;
    const uint32_t *stab;       // s-mer exact table (ordered linear probing, EMPTY32)
    uint32_t  stab_has_ones;    // the all-ones s-mer (poly-T, only possible for s == 16) is present
    uint32_t  use_stab;         // stage 3 on: stage 2 is too full to be trusted alone (large baits)
    // bait-sized fronts behind (or instead of) the LDS table, for baits the 128 KiB of LDS cannot screen (screen2_kernel): blocked bit
    // tables of the stage-1 kind (128-bit blocks, one bit per dword; both strands inserted, or one canonical key per s-mer: `canon`) in global memory.  front2 stays within
    // an XCD's L2 (<= 4 MiB); front3 (only where front2 itself is overloaded: baits of several Mbp) is as large as the bait asks.
    uint32_t  front_mode;       // 0: LDS table only (screen_kernel) | 1: LDS table, its positives through front2 turn by turn | 2: every sample through front2
                                // (no LDS table) | 3: LDS table, lone positives queued and looked up in front2 sixty-four at a time (screen3_kernel)
                                // | 4: a one-bit LDS table in front of mode 2's look-ups
    uint32_t  f2_log2b, f3_log2b;   // blocks = 1 << log2b; f3_log2b == 0: no front3
    const uint32_t *front2, *front3;
    // mode 4: a ONE-bit table for the LDS (1 << pre_log2w words, keys as in the other tables, bit = pre_bit(h)): at 100-500 kbp it still answers half to
    // two thirds of the samples itself, and only the rest is looked up in front2
    uint32_t  pre_log2w;
    const uint32_t *pre;
    // protein-space set (peptide k-mers, 5 bits per residue; k = residues per key, kw = 1, no screen)
    uint32_t  prot;             // 1: keys are peptide k-mers and reads are translated in six frames
    uint32_t  kb_in_lds;        // the k-mer bit table is small enough to be staged in LDS
    const uint32_t *plut;       // 64 codon entries of 4 dwords (see CodonLutEntry in mf_host.h)
};

} // namespace mf
