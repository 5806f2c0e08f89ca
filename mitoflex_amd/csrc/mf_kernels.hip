// HIP kernels of libmitofilter_hip for gfx950 (MI355X, CDNA4; 64-wide waves).
//
//   screen_kernel   streams the dense 2-bit read stream once from HBM with 16-byte coalesced non-temporal loads and
//                   tests every stream-aligned s-mer in a blocked bit table held in LDS (stage 1).  A lane with a
//                   positive only RECORDS it (16 bytes: chunk, lane, hit mask) -- no dependent load, no returning
//                   atomic, no division in the streaming loop.  Pure integer/indexing work, HBM-bound by design: persistent
//                   workgroups on seven CUs out of eight (stride 16; every CU for the VALU-bound stride-8 geometries).
//   finish_kernel   threshold 1, no hit counts (the default pass): one thread per record, two launches.  Phase 0 tries the
//                   first run of neighbouring positives of every record -- one canonical k-mer, one probe of the
//                   open-address bait table; phase 1 settles what is left (exact s-mer table, then the sixteen windows a
//                   sample owns; a read's pass bit is looked at before anything of the read is fetched).  For k > 32 a run may
//                   continue in the next lane's piece.  A pass is an atomicOr.  Runs on a second stream under the next pass's
//                   screen kernel; consecutive screens alternate between two streams.
//   mark_kernel     any threshold / hit counts: groups a record's positives by read, verifies lone ones (stage 2: Bloom over
//                   canonical s-mers in L2; stage 3 for large baits: exact s-mer table) and sets candidate bits.
//   exact_kernel    any threshold / hit counts, and the exhaustive mode: k-mer extract -> canonicalise -> LDS bit table ->
//                   open-address table in L2 -> threshold, for candidate reads (or every read); a candidate's k-mer positions
//                   are dealt to the lanes of a wave sixteen at a time, per-read counts live in LDS.  Pipelined like the default
//                   pass when no hit counts are wanted; beside a screen workgroup the exact kernel runs in a co-resident form
//                   (512 threads, bit table folded to 16 KiB).
//   build_*         device-side bait set builder (history-independent table), screen tables, k-mer bit table.
//   qualscan_kernel / seqhash_kernel   the FASTQ quality filter's counts and SipHash-1-3 (filter_v2 drop-in).
//
// Why the screen is exact (not a heuristic): a read can only have a k-mer hit if some window of k bases equals a bait
// k-mer (either strand).  That window fully contains a stream-aligned s-mer (k >= s + stride - 1), and that s-mer, read
// as it lies in the stream, is an s-mer of the bait or of its reverse complement -- both are in the stage-1 table.  So
// "no sampled s-mer of the read is positive" proves hits == 0 < T, and every read with a positive is decided exactly
// (finish_kernel or exact_kernel), so the emitted bits equal the brute-force oracle's.
#include "mf_common.h"
#include "mf_kernels.h"
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>

namespace mf {

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh)
{   // (hi:lo >> sh) & 0xffffffff, sh in [0,31]  -> v_alignbit_b32
    return __funnelshift_r(lo, hi, sh);
}

__device__ __forceinline__ uint64_t lower_bound_u64(const uint64_t *__restrict__ a, uint64_t n, uint64_t v)
{
    uint64_t lo = 0, hi = n;
    while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

// first index i with npos[i] >= v, through the block index
__device__ __forceinline__ uint64_t npos_lower_bound(const ReadsView &R, uint64_t v)
{
    const uint64_t b = v >> NPOS_BLK_SHIFT;
    uint64_t lo = R.npos_blk[b], hi = R.npos_blk[b + 1];          // the answer lies in [lo, hi]
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (R.npos[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

// read index holding global base g, or ~0 if the s bases from g do not lie in one read; *start (optional) = the read's first base
__device__ __forceinline__ uint64_t read_holding(const ReadsView &R, uint64_t g, uint32_t s, uint64_t *start = nullptr, uint64_t *end = nullptr)
{
    if (g + s > R.total_bases) return ~0ULL;
    if (R.uniform_len) {
        // exact floor division by multiplication (g * uniform_len < 2^64 always holds here)
        uint64_t r = R.uniform_len > 1 ? __umul64hi(g, R.len_magic) : g;
        uint64_t off = g - r * R.uniform_len;
        if (start) *start = g - off;
        if (end) *end = g - off + R.uniform_len;
        return off + s <= R.uniform_len ? r : ~0ULL;
    }
    // Ragged reads: the block index over the offsets (offblk_lookup, mf_common.h): one 8-byte load
    return offblk_lookup(R.off_blk, R.offsets, R.n_reads, g, s, start, end);
}

__device__ __forceinline__ bool stab_contains(const KmerSetView &S, uint32_t sm)
{
    if (sm == EMPTY32) return S.stab_has_ones != 0;
    uint32_t slot = smer_hash(sm) & S.stab_mask;
    uint32_t e = S.stab[slot];
    while (e < sm) { slot = (slot + 1) & S.stab_mask; e = S.stab[slot]; }   // ordered table: stop at first e >= sm
    return e == sm;
}

// ------------------------------------------------------------ screen kernel
// LDS image: the stage-1 table only (up to 128 KiB of the CU's 160) and the record counter.
//   stage 1 (every sampled s-mer, in the hot loop): one ds_read_b128 of the s-mer's 128-bit block, one bit tested in each
//     dword.  Both strands of the bait are inserted, so the s-mer is used exactly as it lies in the stream.
//   Positives are recorded, not followed up: the later stages (finish_kernel, or mark_kernel's stage 2 / 3) run in their
//   own launches, so global memory is touched in the loop only by the streaming loads and the record stores.
// SPW = samples per u32 word (1: stride 16 bases, 2: stride 8 bases)
// U   = uint4 loads per lane per chunk (two chunks are in flight: the one being examined and the next)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// pointers that keep the LDS address space (a generic pointer to LDS compiles to flat_* instructions, which count
// in vmcnt and would drain a streaming wave's prefetch queue)
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
__device__ __forceinline__ uint32_t lds_ld(const lds_u32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(lds_u32 *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ unsigned long long lds_ld(const lds_u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(lds_u64 *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t lds_add(lds_u32 *p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#define MF_COMPILER_FENCE() __atomic_signal_fence(__ATOMIC_SEQ_CST)

// data << (byte BYTE of amt): v_lshlrev_b32 uses the low five bits of its shift operand, and an
// SDWA source selector picks the byte, so no separate extract is issued (dst_sel is DWORD, so the
// partial-write forwarding hazard of SDWA destinations does not apply)
template <int BYTE>
__device__ __forceinline__ uint32_t lshl_by_byte(uint32_t amt, uint32_t data)
{
    uint32_t r;
    if (BYTE == 0) return data << (amt & 31u);
    if (BYTE == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(amt), "v"(data));
    if (BYTE == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(amt), "v"(data));
    if (BYTE == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(amt), "v"(data));
    return r;
}

// A 16-base sample or its reverse complement, whichever is smaller (revcomp_s(x, 16) in six vector instructions).  A set built CANONICAL
// (KmerSetView::canon) inserts one key per bait s-mer into its screen tables instead of one per strand: half the load on the 128 KiB LDS table --
// at 100 kbp it passes 1 % of the samples instead of 9 % -- for six instructions a sample.  What a screen kernel records is positions,
// so nothing behind the screen knows.
__device__ __forceinline__ uint32_t canon16(uint32_t x)
{
    const uint32_t y = __brev(~x);
    const uint32_t rc = (0xAAAAAAAAu & (y << 1)) | (0x55555555u & (y >> 1));
    return x < rc ? x : rc;
}
// The same for samples of fewer than sixteen bases (k < 31): the sample is the low 2s bits of `x` (what lies above rides along and falls out of the
// reverse complement with the shift), eight vector instructions.  CANON: 0 the sample as it lies in the stream | 1 sixteen bases | 2 fewer.
template <int CANON>
__device__ __forceinline__ uint32_t canon_key(uint32_t x, uint32_t smask, uint32_t rc_shift)
{
    if (CANON == 0) return x;
    if (CANON == 1) return canon16(x);
    const uint32_t y = __brev(~x);
    const uint32_t rc = ((0xAAAAAAAAu & (y << 1)) | (0x55555555u & (y >> 1))) >> rc_shift;
    const uint32_t xm = x & smask;
    return xm < rc ? xm : rc;
}

// A stage-1 positive record: everything mark_kernel needs to finish the job later.
// One per (lane, chunk) that saw at least one positive.  Sample i of the lane's chunk slice
// (i = (u*4+q)*SPW+j) is bit NS-1-i of hitmask, NS = U*4*SPW.
struct __attribute__((aligned(16))) ScreenRec { uint32_t chunk, tid, hitmask, pad; };

// SPW: samples per word (stride 16 -> 1, stride 8 -> 2).  For s < 16 a sample is the low 2s bits of what is hashed; the bits
// above them ride along (stage1_index_lo in mf_common.h).
template <int SPW, int U, int CANON = 0>
__global__ void __launch_bounds__(1024)
screen_kernel(ReadsView R, KmerSetView S, ScreenRec *__restrict__ recs, uint32_t rec_cap, uint32_t *__restrict__ rec_counts,
              uint4 *__restrict__ clear, uint64_t clear_vec4)
{
    extern __shared__ uint4 s_tab4[];                                       // stage-1 table, then the record counter
    const uint32_t nb4 = (1u << S.bloom_log2w) >> 2;
    uint32_t &s_nrec = *reinterpret_cast<uint32_t *>(s_tab4 + nb4);         // (all LDS in one array: the dynamic base stays 16-byte aligned)
    const u32x4 *__restrict__ w4 = reinterpret_cast<const u32x4 *>(R.words);
    const uint64_t chunk = (uint64_t)blockDim.x * U;
    const uint64_t n_chunks = R.n_vec / chunk;          // n_vec is padded to a whole number of chunks
    const uint32_t idx_lo = stage1_index_lo(S.s, S.bloom_log2w), idx_bits = S.bloom_log2w - 2;
    const uint64_t cstep = gridDim.x;
    ScreenRec *__restrict__ my_recs = recs + (size_t)blockIdx.x * rec_cap;

    auto load = [&](uint64_t c, u32x4 (&d)[U], uint32_t (&x)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t v = c * chunk + (uint64_t)u * blockDim.x + threadIdx.x;
            d[u] = __builtin_nontemporal_load(&w4[v]);
            if (SPW == 2) x[u] = R.words[4 * v + 4];
        }
    };
    // Stage 1 on one chunk slice held in registers.  Per sample: one multiply (hash), one
    // ds_read_b128 (the sample's 128-bit block), four left shifts that bring the tested bit of each
    // dword into the sign position (shift amounts are bytes of the hash / s-mer, picked by SDWA
    // selectors), two three-input ANDs, and one funnel shift that appends the sign bit to the hit
    // mask.  Positives are only RECORDED (LDS counter -> slot, one 16-byte store): no dependent load,
    // no returning global atomic, no division in the streaming loop -- mark_kernel finishes them.
    auto stage1 = [&](uint64_t c, const u32x4 (&d)[U], const uint32_t (&x)[U]) {
        uint32_t hitmask = 0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t wv[5] = {d[u].x, d[u].y, d[u].z, d[u].w, SPW == 2 ? x[u] : 0u};
#pragma unroll
            for (int q = 0; q < 4; q++) {
#pragma unroll
                for (int j = 0; j < SPW; j++) {
                    // the sample as it lies in the stream: for s < 16 the bases behind it ride along in the bits above 2s (stage1_index_lo)
                    static_assert(CANON != 1 || SPW == 1, "sixteen-base samples come with stride 16");
                    const uint32_t raw = (SPW == 1 || j == 0) ? wv[q] : alignbit(wv[q + 1], wv[q], 16u);
                    const uint32_t sm = canon_key<CANON>(raw, S.smask, 32u - 2u * (uint32_t)S.s);
                    const uint32_t h = bloom_hash(sm);
                    const uint4 blk = s_tab4[__builtin_amdgcn_ubfe(h, idx_lo, idx_bits)];        // one v_bfe_u32
                    const uint32_t t = lshl_by_byte<0>(sm, blk.x) & lshl_by_byte<1>(sm, blk.y) & lshl_by_byte<2>(sm, blk.z) & lshl_by_byte<1>(h, blk.w);   // stage1_field
                    hitmask = alignbit(hitmask, t, 31);             // (hitmask << 1) | sign(t)
                }
            }
        }
        if (hitmask) {
            const uint32_t slot = atomicAdd(&s_nrec, 1u);
            ScreenRec rec; rec.chunk = (uint32_t)c; rec.tid = threadIdx.x; rec.hitmask = hitmask; rec.pad = 0;
            my_recs[slot] = rec;
        }
    };

    // two register sets, ping-pong: the next chunk is in flight while this one is examined
    u32x4 a[U], b[U]; uint32_t ax[U], bx[U];
    uint64_t c = blockIdx.x;
    if (c < n_chunks) load(c, a, ax);                  // the first chunk is already on its way while the table is staged
    {
        // a threshold-1 pass sets its result bits with atomics: the bitmap of the pass after this one is cleared here
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < clear_vec4; i += cstep * blockDim.x) clear[i] = z;
        const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(S.bloom);
        for (uint32_t i = threadIdx.x; i < nb4; i += blockDim.x) s_tab4[i] = src[i];
        if (threadIdx.x == 0) s_nrec = 0;
    }
    __syncthreads();
    // The loads sit on unconditional paths (the last chunk is peeled), so the compiler's vmcnt counts are exact and
    // examining a chunk never waits for the loads issued right before it.
    if (c < n_chunks) for (;;) {
        if (c + cstep >= n_chunks) { stage1(c, a, ax); break; }
        load(c + cstep, b, bx);
        stage1(c, a, ax);
        c += cstep;
        if (c + cstep >= n_chunks) { stage1(c, b, bx); break; }
        load(c + cstep, a, ax);
        stage1(c, b, bx);
        c += cstep;
    }
    __syncthreads();
    if (threadIdx.x == 0) rec_counts[blockIdx.x] = s_nrec;
}

// ------------------------------------------------- screen for baits the LDS table cannot hold
// The 128 KiB stage-1 table screens a bait of one mitogenome (33 k s-mers on both strands: 0.06 % false positives); at two mitogenomes it
// passes 0.4 % of the samples, at 100 kbp 9 %, at 350 kbp three quarters -- and every positive used to become a record for the finish
// kernels (a pass fell from 0.24 ms to 1.3 ms at 100 kbp and 29 ms at 350 kbp, profiles/r06/a_bait_sweep_before.txt).  The s-mer test
// itself stays selective far beyond that (2 M s-mers of 4^16 at 1 Mbp); what is missing is a table of the bait's size to ask.  That table
// (front2: blocked bits of the stage-1 kind, <= 4 MiB so that it stays in every XCD's L2; front3 behind it for baits of several Mbp) is
// asked from INSIDE the screen, before anything is recorded:
//   front_mode 1 (LDS table still selective, baits of ~40 .. ~105 kbp): stage 1 as before; the positives of a wave's chunk are compacted
//     into a queue in LDS (ballot / mbcnt, one sample slot of the wave at a time) and looked up by ONE dense wave-wide 16-byte gather a
//     turn: issued at the top of the next turn, in front of the stream's loads, tested behind that chunk's stage 1.  Verified positives go
//     back to their lane through an LDS word per lane (ds_or), and a lane records its chunk's verified mask one turn late: the record
//     format, and everything behind it, is unchanged.  Bound by instruction issue (~185 vector instructions a wave and chunk at 100 kbp
//     against 80 of screen_kernel): 0.40 ms a pass at 100 kbp, where the records of screen_kernel cost 1.3 ms.
//   front_mode 2 (LDS table useless, larger baits): no LDS table; every sample is looked up in front2 (two workgroups a CU, four 16-byte
//     gathers in flight per lane), the survivors in front3 where the bait has one.  Bound by the part's random-gather rate
//     (tools/gather_roof.hip: 265 G lookups/s from a table of <= 4 MiB with nothing else running, 217 G/s beside a 16-byte read stream,
//     55-80 G/s beyond L2), not by HBM: 312.5 M samples a 5 Gbp pass = 1.4-1.5 ms.
// Conservative at every step (a table never loses an inserted s-mer; what a round cannot take is passed on unverified), so the records
// hold every true s-mer match and the emitted bits stay the brute-force oracle's.
constexpr int S2_QCAP = 64;           // queue entries per wave and chunk: ONE round of gathers a turn, issued and tested at fixed places of the turn (no load sits in a
                                      // branch or an inner loop, so the compiler's vmcnt counts are exact); more positives than that are passed on unverified.  A bait whose
                                      // LDS table passes more than ~10 % of the samples (46 a wave and chunk) takes front_mode 2 instead.
constexpr size_t S2_LDS_EXTRA = 16 + (size_t)(SCREEN_BLOCK / 64) * (64 * 4 + S2_QCAP * 8);          // record counter | verified masks | queues

__device__ __forceinline__ uint32_t lds_or(lds_u32 *p, uint32_t v) { return __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t lds_xchg(lds_u32 *p, uint32_t v) { return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// sign bit of the result: the s-mer's four bits are all set in its 128-bit block (stage1_field)
__device__ __forceinline__ uint32_t block_test(uint32_t sm, uint32_t h, const uint4 blk)
{
    return lshl_by_byte<0>(sm, blk.x) & lshl_by_byte<1>(sm, blk.y) & lshl_by_byte<2>(sm, blk.z) & lshl_by_byte<1>(h, blk.w);
}

template <int SPW, int U, bool LDSF, int G = 8, bool PRE = false, int CANON = 0>
__global__ void __launch_bounds__(1024, G == 8 ? 4 : 8)          // (G = 4, no LDS table: two workgroups share a CU)
screen2_kernel(ReadsView R, KmerSetView S, ScreenRec *__restrict__ recs, uint32_t rec_cap, uint32_t *__restrict__ rec_counts,
               uint4 *__restrict__ clear, uint64_t clear_vec4)
{
    extern __shared__ uint4 s_tab4[];                                       // [stage-1 table] record counter | verified masks | queues
    constexpr int NS = U * 4 * SPW;
    const uint32_t nb4 = LDSF ? (1u << S.bloom_log2w) >> 2 : PRE ? (1u << S.pre_log2w) >> 2 : 0u;          // the stage-1 table, or mode 4's one-bit table
    uint32_t &s_nrec = *reinterpret_cast<uint32_t *>(s_tab4 + nb4);
    const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const lds_u32 *s_pre = (const lds_u32 *)reinterpret_cast<const uint32_t *>(s_tab4);
    const uint32_t pre_lo = PRE ? pre_index_lo(S.s, S.pre_log2w) : 0u, pre_bits = PRE ? S.pre_log2w + 5 : 0u;
    lds_u32 *s_ver = (lds_u32 *)(reinterpret_cast<uint32_t *>(s_tab4 + nb4 + 1)) + wid * 64;
    // the queue as two arrays (s-mers, ids): one 8-byte store per entry wants its two halves in a register pair, and the register allocator
    // then keeps the chunk's words in pair positions of their own -- copied there from the stream loads' registers at the end of every
    // turn, behind a wait for loads issued a moment before (one chunk in flight instead of two)
    lds_u32 *s_qsm = (lds_u32 *)(reinterpret_cast<uint32_t *>(s_tab4 + nb4 + 1 + (SCREEN_BLOCK / 64) * 16)) + wid * (2 * S2_QCAP);
    lds_u32 *s_qid = s_qsm + S2_QCAP;
    const u32x4 *__restrict__ w4 = reinterpret_cast<const u32x4 *>(R.words);
    const uint4 *__restrict__ f2 = reinterpret_cast<const uint4 *>(S.front2);
    const uint4 *__restrict__ f3 = reinterpret_cast<const uint4 *>(S.front3);
    const uint64_t chunk = (uint64_t)blockDim.x * U;
    const uint64_t n_chunks = R.n_vec / chunk;
    const uint32_t idx_lo = LDSF ? stage1_index_lo(S.s, S.bloom_log2w) : 0u, idx_bits = LDSF ? S.bloom_log2w - 2 : 0u;
    const uint32_t lo2 = stage1_index_lo(S.s, S.f2_log2b + 2), b2 = S.f2_log2b;
    const uint32_t lo3 = stage1_index_lo(S.s, S.f3_log2b + 2), b3 = S.f3_log2b;
    const uint64_t cstep = gridDim.x;
    ScreenRec *__restrict__ my_recs = recs + (size_t)blockIdx.x * rec_cap;

    auto load = [&](uint64_t c, u32x4 (&d)[U], uint32_t (&x)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t v = c * chunk + (uint64_t)u * blockDim.x + threadIdx.x;
            d[u] = __builtin_nontemporal_load(&w4[v]);
            if (SPW == 2) x[u] = R.words[4 * v + 4];
        }
    };
    auto record = [&](uint64_t c, uint32_t mask) {
        const uint32_t slot = atomicAdd(&s_nrec, 1u);
        ScreenRec rec; rec.chunk = (uint32_t)c; rec.tid = threadIdx.x; rec.hitmask = mask; rec.pad = 0;
        my_recs[slot] = rec;
    };
    // the samples of a chunk slice as they lie in the stream; sample i is bit NS-1-i of a mask (ScreenRec)
    auto samples = [&](const u32x4 (&d)[U], const uint32_t (&x)[U], uint32_t (&sm)[NS]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t wv[5] = {d[u].x, d[u].y, d[u].z, d[u].w, SPW == 2 ? x[u] : 0u};
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int j = 0; j < SPW; j++) {
                    static_assert(CANON != 1 || SPW == 1, "sixteen-base samples come with stride 16");
                    const uint32_t raw = (SPW == 1 || j == 0) ? wv[q] : alignbit(wv[q + 1], wv[q], 16u);
                    sm[(u * 4 + q) * SPW + j] = canon_key<CANON>(raw, S.smask, 32u - 2u * (uint32_t)S.s);          // (the key every table of the screen is asked with)
                }
        }
    };

    // ---- front_mode 1: the first round of the LAST chunk's queue is issued at the top of a turn (in front of the next chunk's stream
    // loads, so that the wait for it does not include those), tested behind this chunk's stage 1, and the last chunk's verified masks are
    // recorded before this chunk's positives go into the queue.  Nothing a load returns lives across a turn: with two copies of the
    // loop body (the stream's two register sets) a loop-carried load result ends in register copies that wait for the load at once.
    uint32_t p_n = 0, p_sm = 0, p_id = 0, p_off = 0; uint4 p_blk = make_uint4(0, 0, 0, 0);
    uint64_t prev_c = 0;
    const char *__restrict__ f2_bytes = reinterpret_cast<const char *>(S.front2);
    auto issue_prev = [&] {
        // UNCONDITIONAL gather (a lane without an entry asks for block 0, one request for all of them): a load the compiler cannot count
        // makes its vmcnt waits conservative, and this chunk's stage 1 would wait for the gather instead of only for its own data.
        // Nothing is computed here: the byte offset was made at the end of the last turn, and the load takes it as it is (scalar base +
        // 32-bit vector offset).  A vector instruction at this spot got a register of the stream's next loads for its result and, in front
        // of it, a wait for every load in flight -- one chunk in flight instead of two, 0.36 ms a pass instead of 0.29.
        p_blk = *reinterpret_cast<const uint4 *>(f2_bytes + p_off);
    };
    auto settle_prev = [&] {          // test the round, then every lane takes (and clears) the verified mask of its slice of the last chunk
        // (the test itself is unconditional: a load whose only use sits in a branch is sunk into that branch by the compiler -- issued where it is waited for)
        const uint32_t pt = block_test(p_sm, bloom_hash(p_sm), p_blk);
        if (lane < p_n && (int32_t)pt < 0) lds_or(&s_ver[p_id & 63u], 1u << (p_id >> 8));
        MF_COMPILER_FENCE();          // (a wave's LDS operations execute in the order they were issued)
        const uint32_t v = lds_xchg(&s_ver[lane], 0u);
        if (v) record(prev_c, v);
    };

    auto process = [&](uint64_t c, const u32x4 (&d)[U], const uint32_t (&x)[U]) {
        uint32_t sm[NS];
        samples(d, x, sm);
        if (LDSF) {
            // stage 1 and the compaction of its positives into the wave's queue, eight samples at a time (one sample slot of the wave per ballot)
            uint32_t base = 0;
            uint32_t hms[NS / 8];
#pragma unroll
            for (int g0 = 0; g0 < NS; g0 += 8) {
                uint32_t hm = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint32_t h = bloom_hash(sm[g0 + i]);
                    const uint4 blk = s_tab4[__builtin_amdgcn_ubfe(h, idx_lo, idx_bits)];
                    hm = alignbit(hm, block_test(sm[g0 + i], h, blk), 31);          // sample g0 + i: bit 7-i
                }
                hms[g0 / 8] = hm;
            }
            settle_prev();                                              // the last chunk's first round (issued at the top of this turn)
            constexpr int GL = 8;
#pragma unroll
            for (int g0 = 0; g0 < NS; g0 += GL) {
                const uint32_t hm = hms[g0 / GL];
#pragma unroll
                for (int i = 0; i < GL; i++) {
                    const bool hit = (hm >> (GL - 1 - i)) & 1u;
                    const uint64_t bm = __builtin_amdgcn_ballot_w64(hit);
                    if (bm) {
                        const uint32_t off = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                        const uint32_t bit = (uint32_t)(NS - 1 - (g0 + i));
                        if (hit) {
                            if (off < (uint32_t)S2_QCAP) { lds_st(&s_qsm[off], sm[g0 + i]); lds_st(&s_qid[off], lane | (bit << 8)); }
                            else lds_or(&s_ver[lane], 1u << bit);          // no room: passed on unverified
                        }
                        base += (uint32_t)__popcll(bm);
                    }
                }
            }
            MF_COMPILER_FENCE();
            p_n = base < (uint32_t)S2_QCAP ? base : (uint32_t)S2_QCAP;      // the round is issued at the top of the next turn
            prev_c = c;
            p_sm = lds_ld(&s_qsm[lane]); p_id = lds_ld(&s_qid[lane]);       // (beyond the queue's fill: stale values, never used)
            p_off = lane < p_n ? __builtin_amdgcn_ubfe(bloom_hash(p_sm), lo2, b2) << 4 : 0u;
        } else {
            uint32_t mask = 0;                                          // G: gathers in flight per lane
#pragma unroll
            for (int g0 = 0; g0 < NS; g0 += G) {
                uint4 blk[G];                                               // (the hash is computed again where it is needed: a register each would cost the second workgroup of the CU)
                uint32_t pre = (1u << G) - 1u;                              // sample g0 + i: bit G-1-i
                if (PRE) {                                                  // mode 4: the one-bit LDS table first; only its positives are looked up
                    pre = 0;
#pragma unroll
                    for (int i = 0; i < G; i++) {
                        const uint32_t bi = __builtin_amdgcn_ubfe(bloom_hash(sm[g0 + i]), pre_lo, pre_bits);
                        const uint32_t w = lds_ld(&s_pre[bi >> 5]);
                        pre = alignbit(pre, w << (~bi & 31u), 31);          // the bit in the sign position
                    }
                }
#pragma unroll
                for (int i = 0; i < G; i++) if (!PRE || ((pre >> (G - 1 - i)) & 1u)) blk[i] = f2[__builtin_amdgcn_ubfe(bloom_hash(sm[g0 + i]), lo2, b2)];
                uint32_t m = 0;
#pragma unroll
                for (int i = 0; i < G; i++) {
                    const uint32_t t = (!PRE || ((pre >> (G - 1 - i)) & 1u)) ? block_test(sm[g0 + i], bloom_hash(sm[g0 + i]), blk[i]) : 0u;
                    m = alignbit(m, t, 31);
                }
                if (b3 && __ballot(m != 0)) {                           // the survivors through front3
#pragma unroll
                    for (int i = 0; i < G; i++) if ((m >> (G - 1 - i)) & 1u) blk[i] = f3[__builtin_amdgcn_ubfe(bloom_hash(sm[g0 + i]), lo3, b3)];
                    uint32_t m3 = 0;
#pragma unroll
                    for (int i = 0; i < G; i++) {
                        const uint32_t t = ((m >> (G - 1 - i)) & 1u) ? block_test(sm[g0 + i], bloom_hash(sm[g0 + i]), blk[i]) : 0u;
                        m3 = alignbit(m3, t, 31);
                    }
                    m = m3;
                }
                mask = (mask << G) | m;
            }
            if (mask) record(c, mask);
        }
    };

    u32x4 a[U], b[U]; uint32_t ax[U], bx[U];
    uint64_t c = blockIdx.x;
    if (c < n_chunks) load(c, a, ax);
    {
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < clear_vec4; i += cstep * blockDim.x) clear[i] = z;
        if (LDSF || PRE) {
            const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(LDSF ? S.bloom : S.pre);
            for (uint32_t i = threadIdx.x; i < nb4; i += blockDim.x) s_tab4[i] = src[i];
        }
        lds_st(&s_ver[lane], 0u); lds_st(&s_qsm[lane], 0u); lds_st(&s_qid[lane], 0u);
        if (threadIdx.x == 0) s_nrec = 0;
    }
    __syncthreads();
    // Two register sets in turn, as in screen_kernel (a single set that is copied at the end of a turn waits there for the loads it
    // has just issued: the copy needs the data).  The next chunk's loads are unconditional -- the last turn reads its own chunk again --
    // so the compiler's vmcnt counts stay exact without peeling the last chunk (two copies of this long body instead of four).
    if (c < n_chunks) for (;;) {
        {
            const bool last = c + cstep >= n_chunks;
            const uint64_t cn = last ? c : c + cstep;
            if (LDSF) issue_prev();
            load(cn, b, bx);
            process(c, a, ax);
            if (last) break;
            c = cn;
        }
        {
            const bool last = c + cstep >= n_chunks;
            const uint64_t cn = last ? c : c + cstep;
            if (LDSF) issue_prev();
            load(cn, a, ax);
            process(c, b, bx);
            if (last) break;
            c = cn;
        }
    }
    if (LDSF) { issue_prev(); settle_prev(); }                          // the last chunk's first round
    __syncthreads();
    if (threadIdx.x == 0) rec_counts[blockIdx.x] = s_nrec;
}

// ----------------------------------------------- screen for baits a little too large for the LDS table
// front_mode 3 (stride-16 geometries, baits of ~20 .. ~60 kbp: the LDS table passes 0.1 .. 2 % of the samples).  screen_kernel's loop with
// one change: a positive of a lane's chunk slice is not recorded at once -- the sample goes into a queue of the wave in LDS
// (ballot / mbcnt; what is selective here is the LDS table, so a wave sees a handful a chunk), and when 64 have gathered they are looked up
// in front2 by one dense 16-byte gather and only the survivors are recorded, each on behalf of the lane that queued it.  A slice with two
// NEIGHBOURING positives is recorded as ever: that is a bait read, and the finish kernel's run detection wants it in one record.  The per-chunk
// cost over screen_kernel is a dozen vector instructions; the look-ups cost one synchronous gather every tens of chunks.  The records, the
// finish kernels and the emitted bits are what they were -- minus the false positives that cost a 33 kbp bait 0.306 ms a pass instead of 0.24.
constexpr int S3_QN = 128;                 // queue entries per wave: fewer than 64 wait, a chunk adds at most 64
// (10 bytes an entry -- s-mer, chunk, 16-bit id -- and not 12: with 24 KiB of queues beside the 128 KiB table a finish workgroup's 8 KiB no longer
// fit on the CU, and the finish kernels of pass i, which run under the screen of pass i + 1, were left the one CU in eight that screen does not take)
constexpr size_t S3_LDS_EXTRA = 16 + (size_t)(SCREEN_BLOCK / 64) * (S3_QN * 10);

template <int U, int CANON = 0>
__global__ void __launch_bounds__(1024)
screen3_kernel(ReadsView R, KmerSetView S, ScreenRec *__restrict__ recs, uint32_t rec_cap, uint32_t *__restrict__ rec_counts,
               uint4 *__restrict__ clear, uint64_t clear_vec4)
{
    extern __shared__ uint4 s_tab4[];                                       // stage-1 table | record counter | per wave: queued s-mers, chunks, ids
    constexpr int NS = U * 4;
    static_assert(NS == 8, "the sample select below is written for eight samples a slice");
    const uint32_t nb4 = (1u << S.bloom_log2w) >> 2;
    uint32_t &s_nrec = *reinterpret_cast<uint32_t *>(s_tab4 + nb4);
    const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    typedef __attribute__((address_space(3))) unsigned short lds_u16;
    lds_u32 *s_qsm = (lds_u32 *)(reinterpret_cast<uint32_t *>(s_tab4 + nb4 + 1)) + wid * (S3_QN * 10 / 4);
    lds_u32 *s_qch = s_qsm + S3_QN;
    lds_u16 *s_qid = (lds_u16 *)(s_qch + S3_QN);          // lane-in-workgroup (10 bits) | bit of the hit mask (3 bits) << 10
    const u32x4 *__restrict__ w4 = reinterpret_cast<const u32x4 *>(R.words);
    const uint4 *__restrict__ f2 = reinterpret_cast<const uint4 *>(S.front2);
    const uint64_t chunk = (uint64_t)blockDim.x * U;
    const uint64_t n_chunks = R.n_vec / chunk;
    const uint32_t idx_lo = stage1_index_lo(S.s, S.bloom_log2w), idx_bits = S.bloom_log2w - 2;
    const uint32_t lo2 = stage1_index_lo(S.s, S.f2_log2b + 2), b2 = S.f2_log2b;
    const uint64_t cstep = gridDim.x;
    ScreenRec *__restrict__ my_recs = recs + (size_t)blockIdx.x * rec_cap;
    uint32_t q_n = 0;                                                       // entries in this wave's queue (wave-uniform)
    uint32_t e_w = 0, t_w = 0;                                              // what this wave has recorded and queued, and its turns (wave-uniform)

    auto load = [&](uint64_t c, u32x4 (&d)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) d[u] = __builtin_nontemporal_load(&w4[c * chunk + (uint64_t)u * blockDim.x + threadIdx.x]);
    };
    auto record = [&](uint32_t c32, uint32_t tid, uint32_t mask) {
        const uint32_t slot = atomicAdd(&s_nrec, 1u);
        ScreenRec rec; rec.chunk = c32; rec.tid = tid; rec.hitmask = mask; rec.pad = 0;
        my_recs[slot] = rec;
    };
    // the first 64 entries of the queue through front2; the rest moves to the front
    auto drain = [&] {
        const uint32_t cnt = q_n < 64u ? q_n : 64u;
        const uint32_t esm = lds_ld(&s_qsm[lane]), ech = lds_ld(&s_qch[lane]), eid = s_qid[lane];
        const uint32_t tsm = lds_ld(&s_qsm[64 + lane]), tch = lds_ld(&s_qch[64 + lane]), tid2 = s_qid[64 + lane];
        if (lane < cnt) {
            const uint32_t hh = bloom_hash(esm);
            const uint4 blk = f2[__builtin_amdgcn_ubfe(hh, lo2, b2)];
            if ((int32_t)block_test(esm, hh, blk) < 0) record(ech, eid & 1023u, 1u << (eid >> 10));
        }
        MF_COMPILER_FENCE();
        lds_st(&s_qsm[lane], tsm); lds_st(&s_qch[lane], tch); s_qid[lane] = (unsigned short)tid2;
        q_n -= cnt;
    };
    auto stage1 = [&](uint64_t c, const u32x4 (&d)[U]) {
        uint32_t hitmask = 0;
        uint32_t sm[NS] = {d[0].x, d[0].y, d[0].z, d[0].w, d[1].x, d[1].y, d[1].z, d[1].w};
        if (CANON) {
#pragma unroll
            for (int i = 0; i < NS; i++) sm[i] = canon_key<CANON>(sm[i], S.smask, 32u - 2u * (uint32_t)S.s);          // (the key the LDS table and front2 are asked with, and what is queued)
        }
#pragma unroll
        for (int i = 0; i < NS; i++) {
            const uint32_t h = bloom_hash(sm[i]);
            const uint4 blk = s_tab4[__builtin_amdgcn_ubfe(h, idx_lo, idx_bits)];
            hitmask = alignbit(hitmask, block_test(sm[i], h, blk), 31);          // sample i: bit NS-1-i
        }
        // A slice is recorded as it is only when two NEIGHBOURING samples are positive: what a bait read shows, and what the finish kernel's run detection
        // needs in one record (by chance: 7e-4 of the slices at a table that passes 1 %).  Every other positive -- the lone ones, and since the canonical
        // tables took the form to baits whose table passes 1 % and more, also two or three apart in one slice (0.3 % of the slices at 100 kbp: 0.6 M of a
        // pass's 1.3 M work items; 60 kbp with both strands: 1.86 M work items -> 1.19 M) -- is queued, one per lane and round.
        // (A workgroup's record list holds one record per lane and chunk, and a slice whose positives are queued one by one may come back as
        // several.  A wave keeps count: what it has recorded and queued so far, e_w, never exceeds 64 a turn -- slices with several positives
        // are queued only while the wave is 256 entries under that line, which at 1 % positives it is from its fifth turn on; on bait-rich
        // input it is not, and such slices are recorded as they are.)
        uint32_t todo = hitmask;
        {
            t_w++;
            const bool roomy = e_w + 256u <= 64u * t_w;
            const bool single = (hitmask & (hitmask - 1u)) == 0;
            const bool direct = hitmask != 0 && ((hitmask & (hitmask << 1)) != 0 || (!single && !roomy));
            if (direct) { record((uint32_t)c, threadIdx.x, hitmask); todo = 0; }
            e_w += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(direct));
        }
        for (;;) {
            const uint64_t bm = __builtin_amdgcn_ballot_w64(todo != 0);
            if (!bm) break;
            const uint32_t off = q_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
            if (todo) {
                const uint32_t bit = 31u - (uint32_t)__clz(todo), i = (uint32_t)(NS - 1) - bit;          // the (next) positive sample
                const uint32_t lo = (i & 1u) ? ((i & 2u) ? sm[3] : sm[1]) : ((i & 2u) ? sm[2] : sm[0]);
                const uint32_t hi = (i & 1u) ? ((i & 2u) ? sm[7] : sm[5]) : ((i & 2u) ? sm[6] : sm[4]);
                lds_st(&s_qsm[off], (i & 4u) ? hi : lo); lds_st(&s_qch[off], (uint32_t)c); s_qid[off] = (unsigned short)(threadIdx.x | (bit << 10));
                todo &= ~(1u << bit);
            }
            q_n += (uint32_t)__popcll(bm);
            e_w += (uint32_t)__popcll(bm);
            MF_COMPILER_FENCE();
            if (q_n >= 64u) drain();
        }
    };

    u32x4 a[U], b[U];
    uint64_t c = blockIdx.x;
    if (c < n_chunks) load(c, a);
    {
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < clear_vec4; i += cstep * blockDim.x) clear[i] = z;
        const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(S.bloom);
        for (uint32_t i = threadIdx.x; i < nb4; i += blockDim.x) s_tab4[i] = src[i];
        if (threadIdx.x == 0) s_nrec = 0;
    }
    __syncthreads();
    if (c < n_chunks) for (;;) {
        if (c + cstep >= n_chunks) { stage1(c, a); break; }
        load(c + cstep, b);
        stage1(c, a);
        c += cstep;
        if (c + cstep >= n_chunks) { stage1(c, b); break; }
        load(c + cstep, a);
        stage1(c, b);
        c += cstep;
    }
    while (q_n) drain();                                                    // what is left in the queue
    __syncthreads();
    if (threadIdx.x == 0) rec_counts[blockIdx.x] = s_nrec;
}

// Finishes the screen: for every recorded positive, stage 2 (canonical s-mer, Bloom probes),
// optional stage 3 (exact s-mer table, large baits), then the candidate bit of the read that holds
// the s-mer.  Dense: one record per lane, a few hundred thousand records per 5 Gbp.
constexpr int MARK_SPLIT = 1;       // mark workgroups per screen workgroup (workgroup dispatch costs ~7 ns each: few, fat workgroups)
constexpr int MARK_BLOCK = 1024;     // launch bound; the launch may use fewer threads

template <int SPW, int U>
__global__ void __launch_bounds__(MARK_BLOCK)
mark_kernel(ReadsView R, KmerSetView S, const ScreenRec *__restrict__ recs, uint32_t rec_cap, const uint32_t *__restrict__ rec_counts,
            uint32_t screen_block, uint32_t split, uint32_t *__restrict__ cand)
{
    // stage-2 table straight from global memory: <= 32 KiB, L2 resident, a few probes per record
    const uint32_t *__restrict__ s_st2 = S.bloom + ((size_t)1 << S.bloom_log2w);
    const uint32_t st2_shift = 32 - (S.stage2_log2w + 5);
    const uint32_t smask = S.smask;
    const uint64_t chunk = (uint64_t)screen_block * U;
    const bool fast = R.len_magic32 != 0;
    // `split` workgroups share one screen workgroup's record list.  Measured: the work per record is a chain of
    // latencies, but dispatching a workgroup costs about as much as a record does, so one 1024-thread workgroup
    // per list (two records per lane at 0.5 % bait reads) beats eight of 256 threads by 7 us.
    const uint32_t list = blockIdx.x / split, part = blockIdx.x % split;
    const uint32_t n = rec_counts[list];
    const ScreenRec *__restrict__ my = recs + (size_t)list * rec_cap;
    // Candidate bits are not written one atomic per positive: the records of a wave are neighbours in the
    // stream, so their reads share a handful of bitmap words, and same-line atomics of one wave instruction are
    // executed one after the other (measured: 85 ns per mark with 10 % bait reads).  Every lane collects its
    // marks in (pend_w, pend_b); at the end of the iteration lanes holding the same word OR their bits together
    // with a segmented scan and the last lane of each run issues one atomic.
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t base = part * blockDim.x + (threadIdx.x & ~63u); base < n; base += split * blockDim.x) {   // wave-uniform trip count
        const uint32_t i = base + lane;
        uint32_t pend_w = EMPTY32, pend_b = 0;
        auto mark = [&](uint64_t r) {
            const uint32_t w = (uint32_t)(r >> 5), b = 1u << (r & 31);
            if (w == pend_w) { pend_b |= b; return; }
            if (pend_b) atomicOr(&cand[pend_w], pend_b);              // a lane's reads span two bitmap words at most: rare
            pend_w = w; pend_b = b;
        };
        if (i < n) {
        const ScreenRec rec = my[i];
        // offset (bases, inside its chunk) of sample idx of the recording lane
        constexpr int NS = U * 4 * SPW;
        auto off_of = [&](int idx) -> uint32_t {
            const int j = idx % SPW, q = (idx / SPW) & 3, u = idx / (4 * SPW);
            return ((((uint32_t)u * screen_block + rec.tid) * 4 + q) << 4) + (uint32_t)j * 8;
        };
        // Read arithmetic: the chunk's first base cb = rq * L + rrem; a positive at 32-bit offset `off`
        // inside the chunk sits in read rq + (rrem + off) / L, a 32-bit division by multiplication.
        const uint64_t cb = (uint64_t)rec.chunk * chunk * 64;
        uint64_t rq = 0; uint32_t rrem = 0;
        if (fast) { rq = __umul64hi(cb, R.len_magic); rrem = (uint32_t)(cb - rq * R.uniform_len); }
        uint32_t m = rec.hitmask;                     // sample idx <-> bit NS-1-idx
        // finish one positive: fetch its s-mer, stage 2 (canonical s-mer, STAGE2_K Bloom probes), optional stage 3,
        // then the candidate bit of the read that holds it
        auto verify = [&](int idx, uint64_t r_known) {
            const uint64_t g0 = cb + off_of(idx);
            const uint64_t wi = g0 >> 4;
            uint32_t sm = R.words[wi];
            if (SPW == 2) sm = alignbit(R.words[wi + 1], sm, 16u * (idx % SPW));
            sm &= smask;                                               // all ones for s == 16
            const uint32_t rc = revcomp_s(sm, S.s);
            const uint32_t cn_ = sm < rc ? sm : rc;
            const uint32_t ha = stage2_hash_a(cn_), hb = stage2_hash_b(cn_);
            uint32_t ok = 1;
#pragma unroll
            for (int p = 0; p < STAGE2_K; p++) {
                const uint32_t pos = (ha + (uint32_t)p * hb) >> st2_shift;
                ok &= s_st2[pos >> 5] >> (pos & 31);
            }
            if (!(ok & 1u)) return;
            if (S.use_stab && !stab_contains(S, sm)) return;          // stage 3 (large baits only): exact s-mer table
            uint64_t r = r_known;
            if (r == ~0ULL) { r = read_holding(R, g0, (uint32_t)S.s); if (r == ~0ULL) return; }
            mark(r);
        };
        if (!fast) {
            // Ragged reads: the positives of a lane are walked in stream order; the read of the first is looked up (block index over the
            // offsets), the ones behind it inside the same read need no look-up, and -- as for uniform reads below -- two or more in one
            // read mark it without fetching anything, a lone one goes through stage 2.  (One look-up and one verification per positive,
            // nine for a bait read, made the mark kernel of a ragged 5 Gbp set 242 us against 76 for uniform reads.)
            uint64_t cur_r = ~0ULL, cur_lo = 0, cur_hi = 0; int cur_idx = 0; uint32_t cur_n = 0;
            auto flush = [&] {
                if (cur_n >= 2) mark(cur_r);
                else if (cur_n == 1) verify(cur_idx, cur_r);
            };
            while (m) {
                const int bit = 31 - __clz(m);
                m &= ~(1u << bit);
                const int idx = NS - 1 - bit;
                const uint64_t g0 = cb + off_of(idx);
                if (cur_r != ~0ULL && g0 >= cur_lo && g0 + (uint64_t)S.s <= cur_hi) { cur_n++; continue; }
                flush();
                cur_n = 0; cur_r = ~0ULL;
                if (g0 < cur_hi && cur_hi != 0 && g0 >= cur_lo) continue;                      // straddles the end of the read just left: not a sample of any read
                uint64_t lo_, hi_;
                const uint64_t r = read_holding(R, g0, (uint32_t)S.s, &lo_, &hi_);
                if (r == ~0ULL) continue;
                cur_r = r; cur_lo = lo_; cur_hi = hi_; cur_idx = idx; cur_n = 1;
            }
            flush();
        } else {
        // Uniform read length: the positives of a lane are walked in stream order and grouped by the read
        // they fall into (a division by multiplication each).  Two or more stage-1 positives of one lane
        // inside one read are practically always a bait read (two independent false positives there have
        // probability ~1e-4 per lane and chunk), so such a read is marked without fetching anything; a
        // lone positive goes through stage 2.  Marking is only ever conservative -- the exact kernel decides.
        uint64_t cur_r = ~0ULL; int cur_idx = 0; uint32_t cur_n = 0;
        auto flush = [&] {
            if (cur_n >= 2) mark(cur_r);
            else if (cur_n == 1) verify(cur_idx, cur_r);
        };
        while (m) {
            const int bit = 31 - __clz(m);
            m &= ~(1u << bit);
            const int idx = NS - 1 - bit;
            const uint32_t off = off_of(idx);
            const uint32_t t = rrem + off;
            const uint32_t dq = __umulhi(t, R.len_magic32);
            const uint32_t offr = t - dq * R.uniform_len;
            // straddles two reads / lies in the padding behind the last read: not a sample of any read
            if (offr + (uint32_t)S.s > R.uniform_len || cb + off + S.s > R.total_bases) continue;
            const uint64_t r = rq + dq;
            if (r == cur_r) { cur_n++; continue; }
            flush();
            cur_r = r; cur_idx = idx; cur_n = 1;
        }
        flush();
        }
        }
        // wave-level merge of the pending marks (all 64 lanes take part)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t ow = __shfl_up(pend_w, d), ob = __shfl_up(pend_b, d);
            if (lane >= (uint32_t)d && ow == pend_w) pend_b |= ob;
        }
        const uint32_t next_w = __shfl_down(pend_w, 1);
        if (pend_b && (lane == 63 || next_w != pend_w)) atomicOr(&cand[pend_w], pend_b);
    }
}

// ------------------------------------------------------------- exact kernel
template <int KW> struct Key;
template <> struct Key<1> { uint64_t lo; };
template <> struct Key<2> { uint64_t lo, hi; };

// canonical k-mer whose first base is global base g
template <int KW>
__device__ __forceinline__ Key<KW> canonical_at(const uint32_t *__restrict__ words, uint64_t g, int k);

template <>
__device__ __forceinline__ Key<1> canonical_at<1>(const uint32_t *__restrict__ words, uint64_t g, int k)
{
    const uint64_t bit = 2 * g;
    const uint64_t wi = bit >> 5; const uint32_t sh = (uint32_t)bit & 31;
    const uint32_t w0 = words[wi], w1 = words[wi + 1], w2 = words[wi + 2];
    uint64_t fwd = ((uint64_t)alignbit(w2, w1, sh) << 32) | alignbit(w1, w0, sh);
    if (k < 32) fwd &= (1ULL << (2 * k)) - 1;
    const uint64_t rc = revcomp1(fwd, k);
    return Key<1>{fwd < rc ? fwd : rc};
}

template <>
__device__ __forceinline__ Key<2> canonical_at<2>(const uint32_t *__restrict__ words, uint64_t g, int k)
{
    const uint64_t bit = 2 * g;
    const uint64_t wi = bit >> 5; const uint32_t sh = (uint32_t)bit & 31;
    const uint32_t w0 = words[wi], w1 = words[wi + 1], w2 = words[wi + 2], w3 = words[wi + 3], w4 = words[wi + 4];
    uint64_t lo = ((uint64_t)alignbit(w2, w1, sh) << 32) | alignbit(w1, w0, sh);
    uint64_t hi = ((uint64_t)alignbit(w4, w3, sh) << 32) | alignbit(w3, w2, sh);
    hi &= (1ULL << (2 * k - 64)) - 1;                 // 33 <= k <= 63
    uint64_t rlo, rhi; revcomp2(lo, hi, k, rlo, rhi);
    const bool f = (hi < rhi) || (hi == rhi && lo < rlo);
    return f ? Key<2>{lo, hi} : Key<2>{rlo, rhi};
}

__device__ __forceinline__ bool table_contains(const KmerSetView &S, Key<1> v)
{
    uint64_t slot = hash_key1(v.lo) & S.slot_mask;
    uint64_t e = S.keys[slot];
    while (e < v.lo) { slot = (slot + 1) & S.slot_mask; e = S.keys[slot]; }   // ordered table
    return e == v.lo;
}
__device__ __forceinline__ bool table_contains(const KmerSetView &S, Key<2> v)
{
    uint64_t slot = hash_key2(v.lo, v.hi) & S.slot_mask;
    for (;;) {
        const ulonglong2 e = reinterpret_cast<const ulonglong2 *>(S.keys)[slot];
        const bool less = (e.y < v.hi) || (e.y == v.hi && e.x < v.lo);
        if (!less) return e.x == v.lo && e.y == v.hi;
        slot = (slot + 1) & S.slot_mask;
    }
}

// Exact path: k-mer extract -> canonicalise -> bait table probe -> hit threshold,
// with the bait set staged through LDS.
//
// Each persistent 1024-thread workgroup first copies the k-mer bit table (a
// blocked Bloom filter over the canonical bait k-mers, <= 128 KiB) into LDS.
// A wave then takes 64 words of the candidate bitmap (2048 reads), gathers the
// candidate reads among them with a prefix scan over per-lane popcounts, and
// deals WORK ITEMS -- (candidate, run of ITEM_POS k-mer positions) -- to its
// lanes, so a read's ~120 positions are spread over 8 lanes.  An item builds
// its k-mers from one 128/192-bit window (a single unaligned 16-byte load plus
// a dword), tests each in the LDS table, and only the positives are verified in
// the open-address table in L2.  Measured before this structure existed: the
// kernel was bound by scattered L2 line requests (~200 per read); the LDS front
// leaves about one per item plus one per verified hit.  In threshold mode an
// item stops verifying as soon as its read's verified count (an LDS counter
// shared by the read's items) reaches the threshold.  Pass bits return to the
// owning bitmap word through LDS and are stored coalesced.
constexpr int ITEM_POS = 16;
constexpr int EXACT_BLOCK = 1024;
constexpr int EXACT_BLOCK_CO = 512;          // co-resident form: shares a CU with a screen workgroup (registers, wave slots, 32 KiB of LDS)

__device__ __forceinline__ uint64_t funnel64(uint64_t lo, uint64_t hi, int sh)   // (hi:lo >> sh), sh in [0,63]
{
    return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
}

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

template <int KW>
__device__ __forceinline__ bool table_has(const KmerSetView &S, uint64_t klo, uint64_t khi, uint32_t h)
{
    uint64_t slot = h & S.slot_mask;
    if (KW == 1) {
        uint64_t e = S.keys[slot];
        while (e < klo) { slot = (slot + 1) & S.slot_mask; e = S.keys[slot]; }       // ordered table
        return e == klo;
    } else {
        const ulonglong2 *__restrict__ kt = reinterpret_cast<const ulonglong2 *>(S.keys);
        ulonglong2 e = kt[slot];
        while ((e.y < khi) || (e.y == khi && e.x < klo)) { slot = (slot + 1) & S.slot_mask; e = kt[slot]; }
        return e.x == klo && e.y == khi;
    }
}

// One work item: k-mer positions [p0, p0 + ITEM_POS) of the read starting at base b0 with n_pos positions.
// Verified hits are added to *cnt (LDS, shared by the read's items).
template <int KW>
__device__ __forceinline__ void item_hits(const ReadsView &R, const KmerSetView &S, const uint2 *__restrict__ s_kb2,
                                          uint32_t kb_shift, uint64_t b0, uint64_t n_pos, uint64_t p0, bool hasn,
                                          uint32_t thr, lds_u32 *cnt, const bool COUNT_ALL)
{
    const int k = S.k;
    const uint64_t mask_lo = (KW == 1 && k < 32) ? (1ULL << (2 * k)) - 1 : ~0ULL;
    const uint64_t mask_hi = KW == 2 ? (1ULL << (2 * k - 64)) - 1 : 0;
    // a window of (ITEM_POS - 1 + k) bases starting at base b0 + p0, first base at bit 0
    const uint64_t bit = 2 * (b0 + p0);
    const uint32_t *__restrict__ w = R.words + (bit >> 5);
    const uint32_t sh = (uint32_t)bit & 31;
    constexpr int NW = KW == 1 ? 4 : 6;                       // aligned words of window; one more is read for the shift
    uint32_t raw[NW + 1];
    {
        const u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(w);
        raw[0] = v.x; raw[1] = v.y; raw[2] = v.z; raw[3] = v.w;
        if (KW == 1) raw[4] = w[4];
        else { const u32x2_a4 v2 = *reinterpret_cast<const u32x2_a4 *>(w + 4); raw[4] = v2.x; raw[5] = v2.y; raw[NW] = w[6]; }
    }
    uint32_t a[NW];
#pragma unroll
    for (int i = 0; i < NW; i++) a[i] = alignbit(raw[i + 1], raw[i], sh);
    const uint64_t x0 = (uint64_t)a[0] | ((uint64_t)a[1] << 32), x1 = (uint64_t)a[2] | ((uint64_t)a[3] << 32);
    uint64_t x2 = 0;
    if (KW == 2) x2 = (uint64_t)a[NW - 2] | ((uint64_t)a[NW - 1] << 32);

    // KW == 1: reverse-complement the whole 128-bit window once.  Base j of W is base 63 - j of
    // RCW, so the reverse complement of the k-mer at position i starts at RCW base 64 - k - i;
    // after dropping the first 64 - k - (ITEM_POS - 1) bases it is a funnel shift by 2*(ITEM_POS-1-i).
    uint64_t r0 = 0, r1 = 0;
    if (KW == 1) {
        const uint64_t y0 = ~swap_pairs_rev64(x1), y1 = ~swap_pairs_rev64(x0);     // RCW = y1:y0
        const int drop = 2 * (64 - k - (ITEM_POS - 1));                            // 2 .. 72 bits, uniform
        if (drop >= 64) { r0 = y1 >> (drop - 64); r1 = 0; }
        else { r0 = (y0 >> drop) | (y1 << (64 - drop)); r1 = y1 >> drop; }
    }
    // canonical key of position i of this item
    auto key_at = [&](int i, uint64_t &klo, uint64_t &khi) {
        if (KW == 1) {
            const uint64_t fwd = funnel64(x0, x1, 2 * i) & mask_lo;
            const uint64_t rc = funnel64(r0, r1, 2 * (ITEM_POS - 1 - i)) & mask_lo;
            klo = fwd < rc ? fwd : rc; khi = 0;
        } else {
            const uint64_t lo = funnel64(x0, x1, 2 * i), hi = funnel64(x1, x2, 2 * i) & mask_hi;
            uint64_t rlo, rhi; revcomp2(lo, hi, k, rlo, rhi);
            const bool f = (hi < rhi) || (hi == rhi && lo < rlo);
            klo = f ? lo : rlo; khi = f ? hi : rhi;
        }
    };

    // LDS stage: which of the ITEM_POS k-mers might be in the bait set
    uint32_t pos_mask = 0;
#pragma unroll
    for (int i = 0; i < ITEM_POS; i++) {
        uint64_t klo, khi;
        key_at(i, klo, khi);
        const uint32_t hb = KW == 1 ? kbit_hash1(klo) : kbit_hash2(klo, khi);
        const uint2 blk = s_kb2[hb >> kb_shift];
        const uint32_t g = kbit_pos(hb);
        const uint32_t t = (blk.x >> (g & 31)) & (blk.y >> ((g >> 5) & 31));
        pos_mask |= (t & 1u) << i;
    }
    // positions past the end of the read, and windows holding an invalid base
    const uint64_t left = n_pos - p0;
    if (left < ITEM_POS) pos_mask &= (1u << left) - 1;
    if (hasn && pos_mask) {
        const uint64_t len = n_pos + k - 1;
        const uint64_t n_lo = npos_lower_bound(R, b0), n_hi = npos_lower_bound(R, b0 + len);
        for (uint64_t n = n_lo; n < n_hi; n++) {
            const int64_t d = (int64_t)(R.npos[n] - (b0 + p0));       // window i holds it iff i <= d < i + k
#pragma unroll
            for (int i = 0; i < ITEM_POS; i++) if (d >= i && d < i + k) pos_mask &= ~(1u << i);
        }
    }
    // L2 stage: verify the positives (rebuild the key of position i from the window)
    while (pos_mask) {
        if (!COUNT_ALL && lds_ld(cnt) >= thr) break;
        const int i = __ffs(pos_mask) - 1;
        pos_mask &= pos_mask - 1;
        uint64_t klo, khi;
        key_at(i, klo, khi);
        const uint32_t h = (uint32_t)(KW == 1 ? hash_key1(klo) : hash_key2(klo, khi));
        if (table_has<KW>(S, klo, khi, h)) lds_add(cnt, 1u);
    }
}

__device__ __forceinline__ uint32_t nth_set_bit(uint32_t w, uint32_t n)
{   // position of the n-th (0-based) set bit of w; w has more than n bits set
    uint32_t pos = 0;
#pragma unroll
    for (int s = 16; s >= 1; s >>= 1) {
        const uint32_t c = __popc((w >> pos) & ((1u << s) - 1));
        if (n >= c) { n -= c; pos += s; }
    }
    return pos;
}

// Up to 64 candidate reads, one per lane (`owner` lanes, which must be lanes 0..n-1; b0 = first base, np = k-mer positions,
// hasn = holds an invalid base): deal their work items to the wave's lanes and leave each
// candidate's verified hit count in my_cnt[its lane].
// Phase 0 runs the first EARLY_ITEMS items of every candidate -- most bait reads reach the threshold
// there; phase 1 runs the remaining items only for the candidates that did not.  (One loop body for
// both phases, so the item code is instantiated once.)
__device__ __forceinline__ uint32_t nth_set_lane(uint64_t ballot, uint32_t m)
{   // position of the m-th (0-based) set bit of a 64-bit ballot (select by halving)
    uint32_t posn = 0;
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
        const uint32_t cnt_lo = (uint32_t)__popcll(ballot & ((1ull << sft) - 1));
        if (m >= cnt_lo) { m -= cnt_lo; ballot >>= sft; posn += sft; } else ballot &= (1ull << sft) - 1;
    }
    return posn;
}

template <int KW>
__device__ __forceinline__ void run_candidates(const ReadsView &R, const KmerSetView &S, const uint2 *__restrict__ s_kb2, uint32_t kb_shift,
                                               uint32_t thr, bool two_phase, int lane, bool owner, uint64_t b0, uint32_t np,
                                               uint32_t hasn, lds_u32 *my_cnt, const bool COUNT_ALL)
{
    my_cnt[lane] = 0;
    const uint32_t b0_lo = (uint32_t)b0, b0_hi = (uint32_t)(b0 >> 32);
    auto wave_max = [&](uint32_t v) -> uint32_t {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(v, o); v = t > v ? t : v; }
        return v;
    };
    constexpr uint32_t EARLY_ITEMS = 2;
    const uint32_t items_all = (wave_max(owner ? np : 0) + ITEM_POS - 1) / ITEM_POS;
    uint32_t it_lo = 0, it_hi = (COUNT_ALL || !two_phase || items_all <= EARLY_ITEMS) ? items_all : EARLY_ITEMS;
    uint64_t who = __ballot(owner);
    for (int phase = 0; phase < 2; phase++) {
        if (phase == 1) {
            if (it_hi >= items_all) break;                          // everything ran in phase 0
            const bool need = owner && my_cnt[lane] < thr && np > EARLY_ITEMS * ITEM_POS;
            who = __ballot(need);
            it_lo = EARLY_ITEMS;
            it_hi = (wave_max(need ? np : 0) + ITEM_POS - 1) / ITEM_POS;
        }
        const uint32_t nc = (uint32_t)__popcll(who);
        if (!nc) break;
        // lane m of `map` = lane number of the m-th candidate to run (phase 0: the owners are lanes 0..nc-1)
        const uint32_t map = phase == 0 ? (uint32_t)lane : nth_set_lane(who, (uint32_t)lane);
        for (uint32_t ib = it_lo; ib < it_hi; ib += 32) {           // passes of up to 32 items per candidate
            const uint32_t nit = it_hi - ib < 32 ? it_hi - ib : 32;
            int lg = 0; while (lg < 5 && (1u << lg) < nit) lg++;
            const uint32_t n_items = nc << lg;
            for (uint32_t t0 = 0; t0 < n_items; t0 += 64) {
                const uint32_t t = t0 + lane;
                const int ci = (int)__shfl(map, (int)((t >> lg) & 63)) & 63;
                const uint32_t it = ib + (t & ((1u << lg) - 1));
                const uint64_t cb0 = ((uint64_t)__shfl(b0_hi, ci) << 32) | __shfl(b0_lo, ci);
                const uint32_t cnp = __shfl(np, ci), chn = __shfl(hasn, ci);
                const uint32_t p0 = it * ITEM_POS;
                if (t < n_items && it < it_hi && p0 < cnp)
                    item_hits<KW>(R, S, s_kb2, kb_shift, cb0, cnp, p0, chn != 0, thr, &my_cnt[ci], COUNT_ALL);
            }
        }
    }
}

constexpr int WC_PER_LANE = 4;                     // candidate-bitmap words per lane
constexpr int WC_WORDS = 64 * WC_PER_LANE;         // ... per wave-chunk (8192 reads)

template <int KW, bool COUNT_ALL, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
exact_kernel(ReadsView R, KmerSetView S, uint32_t *__restrict__ cand, uint32_t thr,
             uint32_t *__restrict__ out_bits, uint32_t *__restrict__ hits_out, unsigned long long *__restrict__ partials, uint32_t merge)
{
    extern __shared__ uint4 s_mem4[];
    constexpr int WAVES = BLOCK / 64;
    uint32_t *s_res = reinterpret_cast<uint32_t *>(s_mem4);                  // [WAVES][WC_WORDS]
    uint32_t *s_cnt = s_res + WAVES * WC_WORDS;                              // [WAVES][64]
    constexpr int HEAD4 = (WAVES * WC_WORDS + WAVES * 64) / 4;
    const uint2 *s_kb2 = reinterpret_cast<const uint2 *>(s_mem4 + HEAD4);
    {
        const uint32_t nb4 = (1u << S.kb_log2w) >> 2;
        const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(S.kbloom);
        uint4 *dst = s_mem4 + HEAD4;
        for (uint32_t i = threadIdx.x; i < nb4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const uint32_t kb_shift = 32 - (S.kb_log2w - 1);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t *my_res = s_res + wid * WC_WORDS; lds_u32 *my_cnt = (lds_u32 *)(s_cnt + wid * 64);
    const uint64_t n_bw = (R.n_reads + 31) >> 5;
    const uint64_t n_wc = (n_bw + WC_WORDS - 1) / WC_WORDS;
    const int k = S.k;
    uint32_t tot_pass = 0, tot_cand = 0;

    const uint64_t wc_step = (uint64_t)gridDim.x * WAVES;
    // a lane owns WC_PER_LANE consecutive bitmap words (the buffers are padded past n_bw)
    // (candidate words are cleared as they are consumed, so the next pass needs no memset)
    // (the four words travel as a uint4 by value: as an array handed to the lambda by reference they lived in scratch memory)
    static_assert(WC_PER_LANE == 4, "a lane's words are a uint4");
    auto load_cw = [&](uint64_t wc) -> uint4 {
        const uint64_t wi = wc * WC_WORDS + (uint64_t)lane * WC_PER_LANE;
        uint32_t w[WC_PER_LANE];
#pragma unroll
        for (int j = 0; j < WC_PER_LANE; j++) {
            const bool in = wc < n_wc && wi + j < n_bw;
            w[j] = in ? (cand ? cand[wi + j] : 0xFFFFFFFFu) : 0u;
            if (in && cand && w[j]) cand[wi + j] = 0;
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    };
    uint4 cw_next = load_cw((uint64_t)blockIdx.x * WAVES + wid);
    for (uint64_t wc = (uint64_t)blockIdx.x * WAVES + wid; wc < n_wc; wc += wc_step) {
        const uint64_t wbase = wc * WC_WORDS;
        uint32_t cw[WC_PER_LANE] = {cw_next.x, cw_next.y, cw_next.z, cw_next.w}, pre[WC_PER_LANE + 1];
        pre[0] = 0;
#pragma unroll
        for (int j = 0; j < WC_PER_LANE; j++) {
            const uint64_t wi = wbase + (uint64_t)lane * WC_PER_LANE + j;
            if (wi < n_bw) { const uint64_t rem = R.n_reads - wi * 32; if (rem < 32) cw[j] &= (1u << rem) - 1; }
            pre[j + 1] = pre[j] + __popc(cw[j]);
            my_res[lane * WC_PER_LANE + j] = 0;
        }
        cw_next = load_cw(wc + wc_step);                                      // next chunk's words, a whole chunk early
        tot_cand += pre[WC_PER_LANE];
        // inclusive scan of the per-lane candidate counts
        uint32_t incl = pre[WC_PER_LANE];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        const uint32_t total = __shfl(incl, 63);
        // word-internal prefix counts of a lane, packed 4 x 8 bit (each <= 128)
        const uint32_t pre_packed = pre[1] | (pre[2] << 8) | (pre[3] << 16);
        for (uint32_t cbase = 0; cbase < total; cbase += 64) {
            // lane i decodes candidate cbase + i of this wave-chunk
            const uint32_t ncand = total - cbase < 64 ? total - cbase : 64;
            const bool active = (uint32_t)lane < ncand;
            const uint32_t c = active ? cbase + lane : total - 1;
            int lo = 0, hi = 63;                      // smallest lane whose inclusive count exceeds c
#pragma unroll
            for (int it = 0; it < 6; it++) { const int mid = (lo + hi) >> 1; if (__shfl(incl, mid) > c) hi = mid; else lo = mid + 1; }
            const int src = lo;
            const uint32_t sp = __shfl(pre_packed, src);
            const uint32_t w0 = __shfl(cw[0], src), w1 = __shfl(cw[1], src), w2 = __shfl(cw[2], src), w3 = __shfl(cw[3], src);
            const uint32_t tot_src = (sp >> 16 & 255u) + __popc(w3);
            uint32_t n = c - (__shfl(incl, src) - tot_src);                  // ordinal inside lane src
            const uint32_t p1 = sp & 255u, p2 = (sp >> 8) & 255u, p3 = (sp >> 16) & 255u;
            const int j = n >= p3 ? 3 : n >= p2 ? 2 : n >= p1 ? 1 : 0;
            const uint32_t w = j == 3 ? w3 : j == 2 ? w2 : j == 1 ? w1 : w0;
            n -= j == 3 ? p3 : j == 2 ? p2 : j == 1 ? p1 : 0u;
            const uint32_t b = nth_set_bit(w, n);
            const uint32_t wslot = (uint32_t)src * WC_PER_LANE + j;
            const uint64_t r = (wbase + wslot) * 32 + b;
            uint64_t b0 = 0; uint32_t np = 0, hasn = 0;
            if (active) {
                uint64_t len;
                if (R.uniform_len) { b0 = r * R.uniform_len; len = R.uniform_len; }
                else { b0 = R.offsets[r]; len = R.offsets[r + 1] - b0; }
                const uint64_t n_pos = len >= (uint64_t)k ? len - k + 1 : 0;
                np = n_pos > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)n_pos;
                hasn = (R.has_n[r >> 5] >> (r & 31)) & 1u;
            }
            run_candidates<KW>(R, S, s_kb2, kb_shift, thr, cand != nullptr, lane, active, b0, np, hasn, my_cnt, COUNT_ALL);
            if (active) {
                const uint32_t h = my_cnt[lane];
                if (COUNT_ALL) hits_out[r] = h;
                if (h >= thr) atomicOr(&my_res[wslot], 1u << b);
            }
        }
#pragma unroll
        for (int j = 0; j < WC_PER_LANE; j++) {
            const uint64_t wi = wbase + (uint64_t)lane * WC_PER_LANE + j;
            const uint32_t res = my_res[lane * WC_PER_LANE + j];
            if (wi < n_bw) { if (!merge) out_bits[wi] = res; else if (res) atomicOr(&out_bits[wi], res); }      // merge: behind the finish kernels, whose bits stay
            tot_pass += __popc(res);
        }
    }
    // pass / candidate tallies: one plain store pair per workgroup, summed by the host
    // (same-address atomics serialise at ~11 ns each, and a separate tally kernel costs a launch)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { tot_pass += __shfl_down(tot_pass, o); tot_cand += __shfl_down(tot_cand, o); }
    __syncthreads();
    if (lane == 0) { s_cnt[wid * 2] = tot_pass; s_cnt[wid * 2 + 1] = tot_cand; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long p = 0, c = 0;
        for (int w = 0; w < WAVES; w++) { p += s_cnt[w * 2]; c += s_cnt[w * 2 + 1]; }
        partials[2 * blockIdx.x] = p; partials[2 * blockIdx.x + 1] = c;
    }
    // slots of workgroups that do not exist in this launch must read as zero (no per-pass memset)
    if (blockIdx.x == 0)
        for (uint32_t i = 2 * gridDim.x + threadIdx.x; i < 2 * EXACT_MAX_GRID; i += blockDim.x) partials[i] = 0;
}

// Sixteen-window work item of the finish kernel: does one of the (<= 16) windows starting at read positions p0, p0 + 1, ...
// hold a bait k-mer?  A finisher's round is a chain of memory round trips under a saturated memory system, so the
// chain is kept at two: the window (and the read's has-N bit) -- then all sixteen k-mers go through the LDS bit table
// and the open-address table probes of the positives are issued four at a time, together.
template <int KW>
__device__ __forceinline__ bool sample_item(const ReadsView &R, const KmerSetView &S, const uint2 *__restrict__ s_kb2, uint32_t kb_shift,
                                            bool active, uint64_t r, uint64_t b0, uint64_t n_pos, uint64_t p0)
{
    const int k = S.k;
    const uint64_t mask_lo = (KW == 1 && k < 32) ? (1ULL << (2 * k)) - 1 : ~0ULL;
    const uint64_t mask_hi = KW == 2 ? (1ULL << (2 * k - 64)) - 1 : 0;
    constexpr int NW = KW == 1 ? 4 : 6;
    uint64_t x0 = 0, x1 = 0, x2 = 0, r0 = 0, r1 = 0;
    uint32_t hasn = 0, pm = 0;
    auto key_at = [&](int i, uint64_t &klo, uint64_t &khi) {
        if (KW == 1) {
            const uint64_t fwd = funnel64(x0, x1, 2 * i) & mask_lo;
            const uint64_t rc = funnel64(r0, r1, 2 * (ITEM_POS - 1 - i)) & mask_lo;
            klo = fwd < rc ? fwd : rc; khi = 0;
        } else {
            const uint64_t lo = funnel64(x0, x1, 2 * i), hi = funnel64(x1, x2, 2 * i) & mask_hi;
            uint64_t rlo, rhi; revcomp2(lo, hi, k, rlo, rhi);
            const bool fl = (hi < rhi) || (hi == rhi && lo < rlo);
            klo = fl ? lo : rlo; khi = fl ? hi : rhi;
        }
    };
    if (active) {
        const uint64_t bit = 2 * (b0 + p0);
        const uint32_t *__restrict__ w = R.words + (bit >> 5);
        const uint32_t sh = (uint32_t)bit & 31;
        uint32_t raw[NW + 1];
        {
            const u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(w);
            raw[0] = v.x; raw[1] = v.y; raw[2] = v.z; raw[3] = v.w;
            if (KW == 1) raw[4] = w[4];
            else { const u32x2_a4 v2 = *reinterpret_cast<const u32x2_a4 *>(w + 4); raw[4] = v2.x; raw[5] = v2.y; raw[NW] = w[6]; }
        }
        hasn = (R.has_n[r >> 5] >> (r & 31)) & 1u;
        uint32_t a[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) a[i] = alignbit(raw[i + 1], raw[i], sh);
        x0 = (uint64_t)a[0] | ((uint64_t)a[1] << 32); x1 = (uint64_t)a[2] | ((uint64_t)a[3] << 32);
        if (KW == 2) x2 = (uint64_t)a[NW - 2] | ((uint64_t)a[NW - 1] << 32);
        if (KW == 1) {                                  // reverse complement of the whole window, as in item_hits
            const uint64_t y0 = ~swap_pairs_rev64(x1), y1 = ~swap_pairs_rev64(x0);
            const int drop = 2 * (64 - k - (ITEM_POS - 1));
            if (drop >= 64) { r0 = y1 >> (drop - 64); r1 = 0; }
            else { r0 = (y0 >> drop) | (y1 << (64 - drop)); r1 = y1 >> drop; }
        }
        // LDS stage: which of the k-mers might be in the bait set
#pragma unroll
        for (int i = 0; i < ITEM_POS; i++) {
            uint64_t klo, khi;
            key_at(i, klo, khi);
            const uint32_t hb = KW == 1 ? kbit_hash1(klo) : kbit_hash2(klo, khi);
            const uint2 blk = s_kb2[hb >> kb_shift];
            const uint32_t g = kbit_pos(hb);
            const uint32_t t = (blk.x >> (g & 31)) & (blk.y >> ((g >> 5) & 31));
            pm |= (t & 1u) << i;
        }
        const uint64_t left = n_pos - p0;               // positions past the end of the read
        if (left < (uint64_t)ITEM_POS) pm &= (1u << left) - 1;
        if (hasn && pm) {                               // windows holding an invalid base
            const uint64_t len = n_pos + k - 1;
            const uint64_t n_lo = npos_lower_bound(R, b0), n_hi = npos_lower_bound(R, b0 + len);
            for (uint64_t n = n_lo; n < n_hi; n++) {
                const int64_t d = (int64_t)(R.npos[n] - (b0 + p0));           // window i holds it iff i <= d < i + k
#pragma unroll
                for (int i = 0; i < ITEM_POS; i++) if (d >= i && d < i + k) pm &= ~(1u << i);
            }
        }
    }
    // table stage: the probes of up to four positives are in flight together; an ordered probe sequence that has to
    // go on (first slot holds a smaller key) is followed on its own, which is rare
    bool hit = false;
    while (__ballot(pm != 0)) {
        uint64_t klo[4], khi[4], slot[4]; bool v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            v[j] = pm != 0;
            const int i = v[j] ? __ffs(pm) - 1 : 0;
            pm &= pm - 1;                               // (0 stays 0)
            key_at(i, klo[j], khi[j]);
            slot[j] = (KW == 1 ? hash_key1(klo[j]) : hash_key2(klo[j], khi[j])) & S.slot_mask;
        }
        if (KW == 1) {
            uint64_t e[4];
#pragma unroll
            for (int j = 0; j < 4; j++) e[j] = v[j] ? S.keys[slot[j]] : EMPTY64;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                while (e[j] < klo[j]) { slot[j] = (slot[j] + 1) & S.slot_mask; e[j] = S.keys[slot[j]]; }
                hit |= v[j] && e[j] == klo[j];
            }
        } else {
            const ulonglong2 *__restrict__ kt = reinterpret_cast<const ulonglong2 *>(S.keys);
            ulonglong2 e[4];
#pragma unroll
            for (int j = 0; j < 4; j++) e[j] = v[j] ? kt[slot[j]] : make_ulonglong2(EMPTY64, EMPTY64);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                while ((e[j].y < khi[j]) || (e[j].y == khi[j] && e[j].x < klo[j])) { slot[j] = (slot[j] + 1) & S.slot_mask; e[j] = kt[slot[j]]; }
                hit |= v[j] && e[j].x == klo[j] && e[j].y == khi[j];
            }
        }
        if (hit) pm = 0;
    }
    return hit;
}

// --------------------------------------------------------------- finish kernels
// Second stage of a threshold-1 pass (no hit counts wanted): one thread per stage-1 record of screen_kernel.
// A window of k bases that is a bait k-mer contains only true stage-1 positives, in particular the first stream-aligned
// s-mer at or after its start -- call that sample its OWNER.  Every window inside a read belongs to exactly one sample
// inside that read (the `stride` windows starting at g0 - stride + 1 .. g0), so a read passes iff some positive sample
// owns a window that is in the bait set, and a pass is an idempotent atomicOr: no claim, no candidate bitmap.
// Nearly every positive is settled cheaply, in two memory round trips, by two launches of this kernel:
//  PHASE 0, RUNS      a bait read shows as runs of neighbouring positives; a run of n_adj samples spans k bases and the window of
//                     k bases at its first sample is almost surely a bait k-mer: ONE canonical key, ONE table probe per record.
//                     49 registers, no table in LDS: it runs on the second stream under the next pass's screen kernel, in the
//                     registers and wave slots that one leaves free.
//  PHASE 1, THE REST  every read a run has passed is known now (kernel boundary), so a record's remaining positives mostly
//                     end at one look at their read's bit.  What is left (mostly stage-1 false positives) is looked up in the
//                     exact s-mer table, and a true bait s-mer outside any run gets its sixteen windows counted (sample_item).
constexpr int FINISH_BLOCK = 256;
constexpr int FINISH_GRID = 1024;            // one tally pair per workgroup in the first half of the tally buffer (MF_FINISH_GRID: up to EXACT_MAX_GRID; 2048 measured the same)
static_assert(FINISH_GRID <= EXACT_MAX_GRID, "tally buffer");          // phase 1's pairs go to the second half

template <int SPW, int U, int KW, int PHASE>
__global__ void __launch_bounds__(FINISH_BLOCK)
finish_kernel(ReadsView R, KmerSetView S, const ScreenRec *__restrict__ recs, uint32_t rec_cap, const uint32_t *__restrict__ rec_counts,
              uint32_t n_lists, uint32_t screen_block, uint32_t *__restrict__ bits, unsigned long long *__restrict__ partials, uint32_t *__restrict__ cand)
{
    __shared__ uint32_t s_pre[EXACT_MAX_GRID + 1];           // exclusive prefix of the record counts
    __shared__ uint32_t s_tot[2];
    {   // every thread sums the counts of four consecutive lists, a wave scan and a scan over the four waves do the rest
        uint32_t c[4], sum = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { const uint32_t i = threadIdx.x * 4 + j; c[j] = i < n_lists ? rec_counts[i] : 0u; sum += c[j]; }
        uint32_t incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if ((threadIdx.x & 63) >= (uint32_t)o) incl += t; }
        __shared__ uint32_t s_wave[FINISH_BLOCK / 64];
        if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
        if (threadIdx.x < 2) s_tot[threadIdx.x] = 0;
        __syncthreads();
        uint32_t base = incl - sum;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) base += s_wave[w];
#pragma unroll
        for (int j = 0; j < 4; j++) { const uint32_t i = threadIdx.x * 4 + j; if (i <= n_lists) s_pre[i] = base; base += c[j]; }
        __syncthreads();
    }
    const uint32_t total = s_pre[n_lists];
    constexpr int NSAMP = U * 4 * SPW, SPP = 4 * SPW;
    const int k = S.k;
    const uint32_t smask = S.smask;
    const bool fast = R.len_magic32 != 0;
    const uint64_t chunk_bases = (uint64_t)screen_block * U * 64;
    const uint64_t mask_lo = (KW == 1 && k < 32) ? (1ULL << (2 * k)) - 1 : ~0ULL;
    const uint64_t mask_hi = KW == 2 ? (1ULL << (2 * k - 64)) - 1 : 0;
    // a run of n_adj samples covers s + (n_adj - 1) * stride >= k bases; its samples lie in one 16-byte piece of the recording lane
    const uint32_t n_adj = (uint32_t)((S.k - S.s + S.stride - 1) / S.stride) + 1;
    const uint32_t run_span = (uint32_t)S.s + (n_adj - 1) * (uint32_t)S.stride;
    uint32_t run_ok = 0;                              // bits whose sample has n_adj - 1 successors inside its piece
    for (int i = 0; i < NSAMP; i++) if ((uint32_t)(i % SPP) + n_adj <= (uint32_t)SPP) run_ok |= 1u << (NSAMP - 1 - i);
    uint32_t n_pass = 0, n_items = 0;
    uint32_t pend_old = 0, pend_bit = 0;              // result of the last pass-bit atomic, looked at a turn later (never waited for)
    auto load_rec = [&](uint32_t ri) -> ScreenRec {
        uint32_t lo = 0, hi = n_lists;                // the list holding record ri: last l with s_pre[l] <= ri
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_pre[mid] <= ri) lo = mid; else hi = mid; }
        return recs[(size_t)lo * rec_cap + (ri - s_pre[lo])];
    };
    const uint32_t stride_r = gridDim.x * FINISH_BLOCK;
    uint32_t ri = blockIdx.x * FINISH_BLOCK + threadIdx.x;
    const ScreenRec no_rec{0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0};
    ScreenRec rec_next = no_rec;
    if (ri < total) rec_next = load_rec(ri);
    // (whole waves stay in the loop -- a lane past the end holds an empty record --: phase 0 looks at the neighbouring lane)
    for (uint32_t base = blockIdx.x * FINISH_BLOCK; base < total; base += stride_r, ri += stride_r) {
        const ScreenRec rec = rec_next;
        rec_next = no_rec;
        if (ri + stride_r < total) rec_next = load_rec(ri + stride_r);        // the next record is on its way while this one is settled
        const uint64_t cb = (uint64_t)rec.chunk * chunk_bases;
        uint64_t rq = 0; uint32_t rrem = 0;
        if (fast) { rq = __umul64hi(cb, R.len_magic); rrem = (uint32_t)(cb - rq * R.uniform_len); }
        uint32_t m = rec.hitmask;
        uint32_t runs = m & run_ok;                   // bit B (sample i) starts a run iff bits B, B-1, ..., B-(n_adj-1) are set (ragged reads too: whether a run lies
                                                      // inside one read is looked up through the offsets' block index where it is taken up)
        for (uint32_t j = 1; j < n_adj; j++) runs &= m << j;
        if (PHASE == 0 && SPW == 1 && KW == 2 && fast) {
            // Runs across the lane border (k > 32; for shorter k a run is two samples and the extra arithmetic does not pay): the records of a list are in lane order, so the next record is usually the next lane
            // of the same chunk, whose 16-byte pieces follow this lane's in the stream.  A run may then start in this lane's
            // piece and end in that one's (a run of k > 32 is three or four samples long and rarely fits one piece: 0.252 -> 0.237 ms
            // per step at k = 41).
            const uint32_t nb_chunk = __shfl_down(rec.chunk, 1), nb_tid = __shfl_down(rec.tid, 1), nb_mask = __shfl_down(rec.hitmask, 1);
            const bool nb = (threadIdx.x & 63) != 63 && nb_chunk == rec.chunk && nb_tid == rec.tid + 1 && rec.hitmask != 0;
            runs = 0;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int sh = 4 * (U - 1 - u);                               // piece u: samples 4u .. 4u+3 are bits sh+3 .. sh
                const uint32_t e = (((rec.hitmask >> sh) & 15u) << 4) | (nb ? (nb_mask >> sh) & 15u : 0u);
                uint32_t re = e;
                for (uint32_t j = 1; j < n_adj; j++) re &= e << j;
                runs |= ((re >> 4) & 15u) << sh;
            }
        }
        // A bait read shows in the records of two or three neighbouring lanes, and one confirmed run passes it: a run start is
        // left to the lane on the left when that lane (the record before this one, if it is the same chunk's previous lane) has a
        // run start of its own inside the same read.  (Should that one fail -- rare: its two samples are bait s-mers -- the read
        // goes to phase 1, which looks at every positive.)
        uint32_t lf_runs = 0; bool lf = false;
        if (PHASE == 0 && SPW == 1) {
            const uint32_t lc = __shfl_up(rec.chunk, 1), lt = __shfl_up(rec.tid, 1);
            lf_runs = __shfl_up(runs, 1);
            lf = (threadIdx.x & 63) != 0 && lc == rec.chunk && lt + 1 == rec.tid;
        }
        if (PHASE == 0) m = runs; else runs = 0;      // phase 0 looks at run starts only, phase 1 at every positive on its own
        uint64_t passed_r = ~0ULL, passed_r2 = ~0ULL; // the last two reads this record has passed
        while (m) {
            // ---- pick the next positive that needs work (run starts first, in stream order) and the read it lies in.  No
            // memory access for uniform read lengths, so positives that need nothing (outside any read, read already passed)
            // are skipped right here instead of costing the wave a turn of round trips.
            bool is_run = false, got = false; uint64_t g0 = 0, r = 0;
            constexpr int NKW = KW == 1 ? 3 : 5;
            uint32_t raw[NKW];                        // the words at the sample (ragged sets, phase 0: asked for beside the read's look-up)
            while (m) {
                is_run = PHASE == 0;                                             // (phase 0: m holds the run starts only)
                const int bit = 31 - __clz(m);
                m &= ~(1u << bit);
                const int idx = NSAMP - 1 - bit;
                const int sj = idx % SPW, sq = (idx / SPW) & 3, su = idx / (4 * SPW);
                const uint32_t off = ((((uint32_t)su * screen_block + rec.tid) * 4 + sq) << 4) + (uint32_t)sj * 8;
                g0 = cb + off;
                const uint32_t span = is_run ? run_span : (uint32_t)S.s;        // bases that have to lie inside one read
                if (fast) {
                    const uint32_t t = rrem + off;
                    const uint32_t dq = __umulhi(t, R.len_magic32);
                    const uint32_t offr = t - dq * R.uniform_len;
                    // straddles two reads / lies in the padding behind the last read: not a sample (or run) of any read
                    if (offr + span > R.uniform_len || g0 + span > R.total_bases) continue;
                    r = rq + dq;
                    if (PHASE == 0 && SPW == 1 && lf) {
                        const uint32_t nib = (lf_runs >> (4 * (U - 1 - su))) & 15u;           // the left lane's run starts in the same piece row
                        if (nib) {
                            const int qn = 3 - (__ffs(nib) - 1);                               // the one nearest to this lane
                            if ((uint32_t)(64 + 16 * (sq - qn)) <= offr) continue;             // ... lies inside this read: that lane's job
                        }
                    }
                } else {
                    if (PHASE == 0) {                 // the run's words do not wait for its read: one round trip less per run start (four -> three)
                        const uint32_t *__restrict__ wp = R.words + ((2 * g0) >> 5);
#pragma unroll
                        for (int i = 0; i < NKW; i++) raw[i] = wp[i];
                    }
                    uint64_t r_start;
                    r = read_holding(R, g0, span, &r_start);
                    if (r == ~0ULL) continue;
                    if (PHASE == 0 && SPW == 1 && lf) {                                          // (as above, with the sample's offset inside its read looked up)
                        const uint32_t nib = (lf_runs >> (4 * (U - 1 - su))) & 15u;
                        if (nib) {
                            const int qn = 3 - (__ffs(nib) - 1);
                            if ((uint64_t)(64 + 16 * (sq - qn)) <= g0 - r_start) continue;
                        }
                    }
                }
                if (r == passed_r || r == passed_r2) continue;
                got = true;
                break;
            }
            if (!got) break;
            n_items++;
            // ---- round trip 1, the same loads for both kinds: the words at the sample, the read's has-N and pass bits
            const uint32_t bitm = 1u << (r & 31);
            const uint64_t wbit = 2 * g0;
            const uint32_t *__restrict__ w = R.words + (wbit >> 5);
            const uint32_t sh = (uint32_t)wbit & 31;
            // the canonical k-mer at the sample, and whether it is a bait k-mer (phase 0: a run's window)
            auto run_hit = [&]() -> bool {
                uint64_t klo, khi = 0;
                const uint64_t lo = (uint64_t)alignbit(raw[1], raw[0], sh) | ((uint64_t)alignbit(raw[2], raw[1], sh) << 32);
                if (KW == 1) {
                    const uint64_t fwd = lo & mask_lo, rc = revcomp1(fwd, k);
                    klo = fwd < rc ? fwd : rc;
                } else {
                    const uint64_t hi = ((uint64_t)alignbit(raw[NKW - 2], raw[2], sh) | ((uint64_t)alignbit(raw[NKW - 1], raw[NKW - 2], sh) << 32)) & mask_hi;
                    uint64_t rlo, rhi; revcomp2(lo, hi, k, rlo, rhi);
                    const bool fl = (hi < rhi) || (hi == rhi && lo < rlo);
                    klo = fl ? lo : rlo; khi = fl ? hi : rhi;
                }
                return table_has<KW>(S, klo, khi, (uint32_t)(KW == 1 ? hash_key1(klo) : hash_key2(klo, khi)));
            };
            if (PHASE == 0 && !fast) {
                // Ragged reads: the words are here (asked for beside the read's look-up), so the probe goes out TOGETHER with the read's has-N and
                // pass bits instead of behind them -- a probe for a read that has passed meanwhile is wasted, a round trip per run start is not.
                const uint32_t hn_w = R.has_n[r >> 5];
                const uint32_t bw = __hip_atomic_load(&bits[r >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const bool hit = run_hit();
                if (pend_bit && !(pend_old & pend_bit)) n_pass++;
                pend_bit = 0;
                if (bw & bitm) { passed_r2 = passed_r; passed_r = r; continue; }
                if ((hn_w >> (r & 31)) & 1u) continue;
                if (hit) {
                    passed_r2 = passed_r; passed_r = r;
                    pend_bit = bitm;
                    pend_old = atomicOr(&bits[r >> 5], bitm);
                }
                continue;
            }
            if (PHASE == 1) {          // nearly every record left over belongs to a read phase 0 has passed: look at its bit before fetching anything else
                // (an ordinary cached load: phase 0 ended at a kernel boundary, so what it passed is visible; a read that another record
                // of this launch passes meanwhile may be missed -- that costs the work below, not correctness)
                const uint32_t bw1 = __hip_atomic_load(&bits[r >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (pend_bit && !(pend_old & pend_bit)) n_pass++;
                pend_bit = 0;
                if (bw1 & bitm) { passed_r2 = passed_r; passed_r = r; continue; }
            }
#pragma unroll
            for (int i = 0; i < NKW; i++) raw[i] = w[i];
            const uint32_t hn_w = R.has_n[r >> 5];
            const uint32_t bw = __hip_atomic_load(&bits[r >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (cached: other records pass reads meanwhile, seeing that late only costs a probe)
            if (pend_bit && !(pend_old & pend_bit)) n_pass++;
            pend_bit = 0;
            if (bw & bitm) { passed_r2 = passed_r; passed_r = r; continue; }     // passed meanwhile
            if (is_run && ((hn_w >> (r & 31)) & 1u)) continue;                   // invalid bases in the read: its positives come back as isolated ones
            // ---- round trip 2: one probe -- the k-mer table for a run, the exact s-mer table for an isolated positive
            if (PHASE == 0) {
                if (run_hit()) {
                    passed_r2 = passed_r; passed_r = r;
                    pend_bit = bitm;
                    pend_old = atomicOr(&bits[r >> 5], bitm);              // the first setter of the bit counts the pass
                }
            } else if (stab_contains(S, alignbit(raw[1], raw[0], sh) & smask)) {   // a true bait s-mer (either strand) outside any run
                // With a candidate bitmap (the default): the read is handed to an exact kernel behind this launch, which deals its windows to
                // eight lanes.  Counting a sample's sixteen windows right here, on this one lane, made the reads that have a bait s-mer but NO
                // matching window -- a bait read with a substitution in every window: 1-2 % of them at k = 41 -- the tail of the launch: eight
                // positives, eight serial sample_items, ~100 us (k = 41: phase 1 167 us against 105 without substitutions, profiles/r06/k41_phase_probe.txt).
                if (cand) { atomicOr(&cand[r >> 5], bitm); passed_r2 = passed_r; passed_r = r; continue; }
                // the windows this sample owns start at g0 - stride + 1 .. g0; the item tests 16 positions from the first of them
                // that lies inside the read (positions past g0 belong to the next sample: harmless for threshold 1)
                uint64_t b0, len;
                if (R.uniform_len) { b0 = r * R.uniform_len; len = R.uniform_len; } else { b0 = R.offsets[r]; len = R.offsets[r + 1] - b0; }
                const uint64_t own_lo = g0 + 1 > (uint64_t)S.stride ? g0 + 1 - S.stride : 0;
                const uint64_t p0 = own_lo > b0 ? own_lo - b0 : 0;
                const uint64_t n_pos = len >= (uint64_t)k ? len - k + 1 : 0;
                if (p0 < n_pos && sample_item<KW>(R, S, reinterpret_cast<const uint2 *>(S.kbloom), 32 - (S.kb_log2w - 1), true, r, b0, n_pos, p0)) {
                    passed_r2 = passed_r; passed_r = r;
                    pend_bit = bitm;
                    pend_old = atomicOr(&bits[r >> 5], bitm);
                }
            }
        }
    }
    if (pend_bit && !(pend_old & pend_bit)) n_pass++;
    // tallies: one plain store pair per workgroup, summed by the host
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { n_pass += __shfl_down(n_pass, o); n_items += __shfl_down(n_items, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_tot[0], n_pass); atomicAdd(&s_tot[1], n_items); }
    __syncthreads();
    if (threadIdx.x == 0) { const uint32_t o = PHASE * EXACT_MAX_GRID + blockIdx.x; partials[2 * o] = s_tot[0]; partials[2 * o + 1] = s_tot[1]; }
}

// ----------------------------------------------------------- bait builders
__global__ void build_keys1_kernel(BaitView B, int k, uint64_t *keys, uint64_t slot_mask)
{
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B.total || B.runlen[p] < k) return;
    uint64_t v = canonical_at<1>(B.words, p, k).lo;
    uint64_t slot = hash_key1(v) & slot_mask;
    // history-independent linear probing: keep the smaller key, carry the larger
    for (;;) {
        const uint64_t old = atomicMin(reinterpret_cast<unsigned long long *>(&keys[slot]), (unsigned long long)v);
        if (old == v) return;
        if (old > v) v = old;
        if (v == EMPTY64) return;
        slot = (slot + 1) & slot_mask;
    }
}

// wide keys (k > 32): slots hold bait positions, ordered by (key, position)
__global__ void build_pos2_kernel(BaitView B, int k, uint32_t *postab, uint64_t slot_mask)
{
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B.total || B.runlen[p] < k) return;
    uint32_t v = (uint32_t)p;
    Key<2> kv = canonical_at<2>(B.words, v, k);
    uint64_t slot = hash_key2(kv.lo, kv.hi) & slot_mask;
    for (;;) {
        const uint32_t c = __hip_atomic_load(&postab[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c == EMPTY32) {
            if (atomicCAS(&postab[slot], EMPTY32, v) == EMPTY32) return;
            continue;                                   // lost the race: look at the slot again
        }
        if (c == v) return;
        const Key<2> kc = canonical_at<2>(B.words, c, k);
        const bool eq = kc.lo == kv.lo && kc.hi == kv.hi;
        if (eq) {                                       // duplicate k-mer: keep the smaller position
            if (v < c) { if (atomicCAS(&postab[slot], c, v) != c) continue; }
            return;
        }
        const bool c_less = (kc.hi < kv.hi) || (kc.hi == kv.hi && kc.lo < kv.lo);
        if (c_less) { slot = (slot + 1) & slot_mask; continue; }
        if (atomicCAS(&postab[slot], c, v) != c) continue;
        v = c; kv = kc;                                 // carry the evicted (larger) key onward
        slot = (slot + 1) & slot_mask;
    }
}

__global__ void materialize2_kernel(BaitView B, int k, const uint32_t *postab, uint64_t slots, uint64_t *keys)
{
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= slots) return;
    const uint32_t p = postab[s];
    if (p == EMPTY32) { keys[2 * s] = EMPTY64; keys[2 * s + 1] = EMPTY64; return; }
    const Key<2> kv = canonical_at<2>(B.words, p, k);
    keys[2 * s] = kv.lo; keys[2 * s + 1] = kv.hi;
}

__device__ __forceinline__ void stage1_insert(uint32_t sm, int s, uint32_t *bloom, uint32_t log2w)
{
    const uint32_t h = bloom_hash(sm);
    uint32_t *blk = bloom + 4 * (size_t)((h >> stage1_index_lo(s, log2w)) & ((1u << (log2w - 2)) - 1));
#pragma unroll
    for (int i = 0; i < 4; i++) atomicOr(&blk[i], 1u << stage1_bit(sm, h, i));
}

__device__ __forceinline__ void stab_insert(uint32_t sm, uint32_t *stab, uint32_t stab_mask, uint32_t *has_ones)
{
    if (sm == EMPTY32) { *has_ones = 1u; return; }
    uint32_t v = sm, slot = smer_hash(sm) & stab_mask;
    for (;;) {                                          // history-independent ordered insertion
        const uint32_t old = atomicMin(&stab[slot], v);
        if (old == v) return;
        if (old > v) v = old;
        if (v == EMPTY32) return;
        slot = (slot + 1) & stab_mask;
    }
}

__global__ void build_screen_kernel(BaitView B, int s, uint32_t *bloom, uint32_t log2w, uint32_t log2w2, uint32_t *stab,
                                    uint32_t stab_mask, uint32_t *has_ones, uint32_t *front2, uint32_t f2_log2b, uint32_t *front3, uint32_t f3_log2b,
                                    uint32_t *pre, uint32_t pre_log2w, bool canon)
{
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B.total || B.runlen[p] < s) return;
    const uint64_t bit = 2 * p; const uint64_t wi = bit >> 5; const uint32_t sh = (uint32_t)bit & 31;
    uint32_t fwd = alignbit(B.words[wi + 1], B.words[wi], sh);
    if (s < 16) fwd &= (1u << (2 * s)) - 1;
    const uint32_t rc = revcomp_s(fwd, s);
    const uint32_t cn = fwd < rc ? fwd : rc;
    // the screen's tables: both strands as they lie in a read -- or, for a set built canonical (s == 16, larger baits), the smaller of the two only,
    // which the screen kernels then make of every sample (canon16)
    const uint32_t k0 = canon ? cn : fwd, k1 = canon ? cn : rc;
    stage1_insert(k0, s, bloom, log2w); if (!canon) stage1_insert(k1, s, bloom, log2w);
    if (front2) { stage1_insert(k0, s, front2, f2_log2b + 2); if (!canon) stage1_insert(k1, s, front2, f2_log2b + 2); }      // the bait-sized fronts: the same blocks, more of them
    if (front3) { stage1_insert(k0, s, front3, f3_log2b + 2); if (!canon) stage1_insert(k1, s, front3, f3_log2b + 2); }
    if (pre) {                                          // mode 4's one-bit table
        const uint32_t lo = pre_index_lo(s, pre_log2w), nb = pre_log2w + 5;
        for (uint32_t v : {k0, k1}) { const uint32_t bi = (bloom_hash(v) >> lo) & ((1u << nb) - 1u); atomicOr(&pre[bi >> 5], 1u << (bi & 31u)); }
    }
    stab_insert(fwd, stab, stab_mask, has_ones); stab_insert(rc, stab, stab_mask, has_ones);
    const uint32_t ha = stage2_hash_a(cn), hb = stage2_hash_b(cn);
    uint32_t *st2 = bloom + ((size_t)1 << log2w);
#pragma unroll
    for (int i = 0; i < STAGE2_K; i++) {
        const uint32_t pos = (ha + (uint32_t)i * hb) >> (32 - (log2w2 + 5));
        atomicOr(&st2[pos >> 5], 1u << (pos & 31));
    }
}

// LDS front of the bait table: one 64-bit block per canonical k-mer, one bit in each dword
__global__ void build_kbloom_kernel(const uint64_t *keys, uint64_t slots, int kw, uint32_t *kbloom, uint32_t kb_log2w)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slots) return;
    const uint64_t lo = keys[i * kw], hi = kw == 2 ? keys[i * kw + 1] : 0;
    if (lo == EMPTY64 && (kw == 1 || hi == EMPTY64)) return;
    const uint32_t hb = kw == 1 ? kbit_hash1(lo) : kbit_hash2(lo, hi), g = kbit_pos(hb);
    uint32_t *blk = kbloom + 2 * (size_t)(hb >> (32 - (kb_log2w - 1)));
    atomicOr(&blk[0], 1u << (g & 31));
    atomicOr(&blk[1], 1u << ((g >> 5) & 31));
}

__global__ void count_keys_kernel(const uint64_t *keys, uint64_t slots, int kw, const uint32_t *stab, uint64_t stab_slots,
                                  unsigned long long *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a = 0, b = 0;
    if (i < slots) a = !(keys[i * kw] == EMPTY64 && (kw == 1 || keys[i * kw + 1] == EMPTY64));
    if (stab && i < stab_slots) b = stab[i] != EMPTY32;
    const uint32_t ca = (uint32_t)__popcll(__ballot(a)), cb = (uint32_t)__popcll(__ballot(b));
    if ((threadIdx.x & 63) == 0) {
        if (ca) atomicAdd(&out[0], (unsigned long long)ca);
        if (cb) atomicAdd(&out[1], (unsigned long long)cb);
    }
}

// block index over the invalid-base list (one thread per block of 4096 bases)
__global__ void build_npos_blk_kernel(const uint64_t *__restrict__ npos, uint64_t n_npos, uint64_t n_blk, uint32_t *__restrict__ blk)
{
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blk) return;
    blk[b] = (uint32_t)lower_bound_u64(npos, n_npos, b << NPOS_BLK_SHIFT);
}

// block index over the offsets of a ragged read set (one thread per block of 128 bases): OffBlk, mf_common.h
__global__ void build_off_blk_kernel(const uint64_t *__restrict__ offsets, uint64_t n_reads, uint64_t n_blk, uint64_t *__restrict__ blk)
{
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blk) return;
    blk[b] = offblk_make(offsets, n_reads, b).pack();
}

// mark reads that hold an invalid base (one thread per invalid position)
__global__ void mark_has_n_kernel(ReadsView R, uint32_t *has_n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R.n_npos) return;
    const uint64_t r = read_holding(R, R.npos[i], 1);
    if (r != ~0ULL) atomicOr(&has_n[r >> 5], 1u << (r & 31));
}

// ------------------------------------------------- FASTQ quality filter (filter_v2)
// Per-read counts the reference's quality filter decides on (filter/filter_bin/src/main.rs:236-243,
// 302-307): number of 'N' in the (cut) sequence, number of quality bytes <= q in the (cut) quality
// string.  The FASTQ text is uploaded as it is; a record is four offsets into it.  Eight lanes per
// record, aligned 16-byte loads coalesced inside the record, counts reduced with 8-wide shuffles.
// bytes of w inside [lo, hi) (byte addresses; w sits at address `addr`) as a 0x80-per-byte mask
__device__ __forceinline__ uint32_t byte_window80(uint32_t addr, uint32_t lo, uint32_t hi)
{
    const uint32_t b0 = lo > addr ? lo - addr : 0u, b1 = hi < addr + 4 ? (hi > addr ? hi - addr : 0u) : 4u;   // valid bytes [b0, b1)
    if (b1 <= b0) return 0u;
    const uint32_t upto1 = b1 >= 4 ? 0xFFFFFFFFu : ((1u << (8 * b1)) - 1), upto0 = b0 >= 4 ? 0xFFFFFFFFu : ((1u << (8 * b0)) - 1);
    return upto1 & ~upto0 & 0x80808080u;
}

constexpr int QS_LANES = 8;         // lanes per record

__device__ __forceinline__ uint32_t count_n_bytes(const uint4 v, uint32_t blk, uint32_t lo, uint32_t hi)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t x = w[j] ^ 0x4E4E4E4Eu;                                      // zero byte <=> 'N'
        const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);  // 0x80 exactly in zero bytes
        c += __popc(z & byte_window80(blk * 16 + 4 * j, lo, hi));
    }
    return c;
}

__device__ __forceinline__ uint32_t count_lowq_bytes(const uint4 v, uint32_t blk, uint32_t lo, uint32_t hi, uint32_t q1)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t d = (w[j] | 0x80808080u) - q1;                               // per byte, no borrow: top bit clear <=> low7 <= q
        c += __popc(~d & ~w[j] & byte_window80(blk * 16 + 4 * j, lo, hi));
    }
    return c;
}

__global__ void __launch_bounds__(256)
qualscan_kernel(const uint8_t *__restrict__ text, const QualRec *__restrict__ recs, uint32_t n, uint32_t quality,
                uint32_t *__restrict__ n_count, uint32_t *__restrict__ bad_count)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t r = t / QS_LANES, l = t % QS_LANES;
    uint32_t nn = 0, nb = 0;
    if (r < n) {
        const QualRec rec = recs[r];
        const uint4 *__restrict__ t4 = reinterpret_cast<const uint4 *>(text);
        // 16-byte aligned blocks covering the two strings; bytes outside a string are masked.  Byte tests are done
        // four at a time on the dwords (SWAR), exact for every byte value (q <= 100 < 0x80, so a quality byte with
        // its top bit set never counts).  A step issues two sequence and two quality loads per lane before any
        // of them is used: a 150-byte string is 10 or 11 blocks, i.e. one step of the eight lanes.
        const uint32_t s_lo = rec.s_off, s_hi = rec.s_off + rec.s_len, q_lo = rec.q_off, q_hi = rec.q_off + rec.q_len;
        const uint32_t sb0 = s_lo >> 4, nsb = rec.s_len ? ((s_hi - 1) >> 4) - sb0 + 1 : 0;
        const uint32_t qb0 = q_lo >> 4, nqb = rec.q_len ? ((q_hi - 1) >> 4) - qb0 + 1 : 0;
        const uint32_t q1 = (quality + 1) * 0x01010101u;
        const uint32_t nmax = nsb > nqb ? nsb : nqb;
        const uint4 zero = make_uint4(0, 0, 0, 0);
        for (uint32_t i = l; i < nmax; i += 2 * QS_LANES) {
            const uint32_t i2 = i + QS_LANES;
            const uint4 s0 = i < nsb ? t4[sb0 + i] : zero, s1 = i2 < nsb ? t4[sb0 + i2] : zero;
            const uint4 q0 = i < nqb ? t4[qb0 + i] : zero, q2 = i2 < nqb ? t4[qb0 + i2] : zero;
            if (i < nsb) nn += count_n_bytes(s0, sb0 + i, s_lo, s_hi);
            if (i2 < nsb) nn += count_n_bytes(s1, sb0 + i2, s_lo, s_hi);
            if (i < nqb) nb += count_lowq_bytes(q0, qb0 + i, q_lo, q_hi, q1);
            if (i2 < nqb) nb += count_lowq_bytes(q2, qb0 + i2, q_lo, q_hi, q1);
        }
    }
#pragma unroll
    for (int o = QS_LANES / 2; o > 0; o >>= 1) { nn += __shfl_xor(nn, o, QS_LANES); nb += __shfl_xor(nb, o, QS_LANES); }
    if (r < n && l == 0) { n_count[r] = nn; bad_count[r] = nb; }
}

// SipHash-1-3 with keys (0, 0) of the (cut) sequence followed by 0xff: what Rust's DefaultHasher
// gives for `seq1.hash()` (main.rs:325-329), so the dedup set behaves exactly like the reference's
// HashSet<u64>.  One lane per record.
__device__ __forceinline__ void sip_round(uint64_t &v0, uint64_t &v1, uint64_t &v2, uint64_t &v3)
{
    auto rotl = [](uint64_t x, int b) { return (x << b) | (x >> (64 - b)); };
    v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32);
    v2 += v3; v3 = rotl(v3, 16); v3 ^= v2;
    v0 += v3; v3 = rotl(v3, 21); v3 ^= v0;
    v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32);
}

__global__ void __launch_bounds__(256)
seqhash_kernel(const uint8_t *__restrict__ text, const QualRec *__restrict__ recs, uint32_t n, uint64_t *__restrict__ hashes)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const QualRec rec = recs[r];
    // the message is the sequence followed by one 0xff byte; it is read as aligned dwords and the
    // 8-byte blocks are cut out with funnel shifts (two new dword loads per block instead of eight byte loads)
    const uint32_t *__restrict__ w = reinterpret_cast<const uint32_t *>(text) + (rec.s_off >> 2);
    const uint32_t sh = (rec.s_off & 3) * 8;
    const uint64_t len = (uint64_t)rec.s_len + 1;
    uint64_t v0 = 0x736F6D6570736575ULL, v1 = 0x646F72616E646F6DULL, v2 = 0x6C7967656E657261ULL, v3 = 0x7465646279746573ULL;
    uint32_t d0 = w[0];
    uint64_t i = 0;
    // 8 message bytes starting at byte i (i is a multiple of 8): dwords 2*(i/8) .. +2 of w, shifted by sh
    auto block = [&](uint64_t at) -> uint64_t {
        const uint32_t *p = w + (at >> 2);
        const uint32_t d1 = p[1], d2 = p[2];
        const uint64_t m = ((uint64_t)alignbit(d2, d1, sh) << 32) | alignbit(d1, d0, sh);
        d0 = d2;
        return m;
    };
    for (; i + 8 <= (uint64_t)rec.s_len; i += 8) {                 // whole blocks inside the sequence
        const uint64_t m = block(i);
        v3 ^= m; sip_round(v0, v1, v2, v3); v0 ^= m;
    }
    // tail: the remaining t < 8 sequence bytes, then 0xff; if t == 7 the 0xff completes a block
    const uint32_t t = (uint32_t)(rec.s_len - i);
    uint64_t tailv = block(i);
    tailv = t == 0 ? 0 : (tailv & (~0ULL >> (64 - 8 * t)));
    tailv |= 0xFFULL << (8 * t);                                    // t <= 7
    if (t == 7) { v3 ^= tailv; sip_round(v0, v1, v2, v3); v0 ^= tailv; tailv = 0; }
    const uint64_t b = ((len & 0xFF) << 56) | tailv;
    v3 ^= b; sip_round(v0, v1, v2, v3); v0 ^= b;
    v2 ^= 0xFF;
    sip_round(v0, v1, v2, v3); sip_round(v0, v1, v2, v3); sip_round(v0, v1, v2, v3);
    hashes[r] = v0 ^ v1 ^ v2 ^ v3;
}

// De-duplication set of the quality filter on the device: HashSet<u64> semantics over the SipHash values (main.rs:244-250: a
// record whose hash an EARLIER record that reached this point carried is dropped).  Open addressing, keys[] holds the hash
// (0 = empty; the hash value 0 itself lives in zero_idx), first[] the smallest file index of a live record that carried it
// (atomicMin): "first occurrence wins" without any order of execution.  Batches come in file order, so an entry left by an
// earlier batch always wins over the current one.
__device__ __forceinline__ uint64_t dedup_mix(uint64_t x) { x ^= x >> 32; x *= 0xD6E8FEB86659FD93ULL; x ^= x >> 32; return x; }

__global__ void __launch_bounds__(256)
dedup_insert_kernel(const uint64_t *__restrict__ hashes, const uint8_t *__restrict__ alive, uint32_t n, uint64_t base,
                    unsigned long long *__restrict__ keys, unsigned long long *__restrict__ first, uint64_t mask,
                    unsigned long long *__restrict__ zero_idx, unsigned long long *__restrict__ n_keys)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !alive[i]) return;
    const uint64_t h = hashes[i];
    if (h == 0) { atomicMin(zero_idx, (unsigned long long)(base + i)); return; }
    uint64_t slot = dedup_mix(h) & mask;
    for (;;) {
        unsigned long long k = keys[slot];
        if (k == 0) { k = atomicCAS(&keys[slot], 0ull, (unsigned long long)h); if (k == 0) { atomicAdd(n_keys, 1ull); k = h; } }
        if (k == h) { atomicMin(&first[slot], (unsigned long long)(base + i)); return; }
        slot = (slot + 1) & mask;
    }
}

__global__ void __launch_bounds__(256)
dedup_check_kernel(const uint64_t *__restrict__ hashes, const uint8_t *__restrict__ alive, uint32_t n, uint64_t base,
                   const unsigned long long *__restrict__ keys, const unsigned long long *__restrict__ first, uint64_t mask,
                   const unsigned long long *__restrict__ zero_idx, uint8_t *__restrict__ dup)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t d = 0;
    if (alive[i]) {
        const uint64_t h = hashes[i];
        if (h == 0) d = *zero_idx != base + i;
        else {
            uint64_t slot = dedup_mix(h) & mask;
            while (keys[slot] != h) slot = (slot + 1) & mask;            // (inserted by the launch before this one)
            d = first[slot] != base + i;
        }
    }
    dup[i] = d;
}

// moves every entry of a full table into one of twice the size
__global__ void __launch_bounds__(256)
dedup_rehash_kernel(const unsigned long long *__restrict__ old_keys, const unsigned long long *__restrict__ old_first, uint64_t old_slots,
                    unsigned long long *__restrict__ keys, unsigned long long *__restrict__ first, uint64_t mask)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= old_slots) return;
    const unsigned long long h = old_keys[i];
    if (h == 0) return;
    uint64_t slot = dedup_mix(h) & mask;
    while (atomicCAS(&keys[slot], 0ull, h) != 0) slot = (slot + 1) & mask;      // (keys are distinct)
    first[slot] = old_first[i];
}

hipError_t launch_dedup(const uint64_t *hashes, const uint8_t *alive, uint32_t n, uint64_t base, unsigned long long *keys,
                        unsigned long long *first, uint64_t slots, unsigned long long *zero_idx, unsigned long long *n_keys, uint8_t *dup,
                        hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(dedup_insert_kernel, dim3((n + 255) / 256), dim3(256), 0, st, hashes, alive, n, base, keys, first, slots - 1, zero_idx, n_keys);
    hipLaunchKernelGGL(dedup_check_kernel, dim3((n + 255) / 256), dim3(256), 0, st, hashes, alive, n, base, keys, first, slots - 1, zero_idx, dup);
    return hipGetLastError();
}

hipError_t launch_dedup_rehash(const unsigned long long *old_keys, const unsigned long long *old_first, uint64_t old_slots,
                               unsigned long long *keys, unsigned long long *first, uint64_t slots, hipStream_t st)
{
    hipLaunchKernelGGL(dedup_rehash_kernel, dim3((unsigned)((old_slots + 255) / 256)), dim3(256), 0, st, old_keys, old_first, old_slots, keys, first, slots - 1);
    return hipGetLastError();
}

hipError_t launch_qualscan(const uint8_t *text, const QualRec *recs, uint32_t n, uint32_t quality, uint32_t *n_count, uint32_t *bad_count,
                           uint64_t *hashes, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(qualscan_kernel, dim3((unsigned)(((uint64_t)n * QS_LANES + 255) / 256)), dim3(256), 0, st, text, recs, n, quality, n_count, bad_count);
    if (hashes) hipLaunchKernelGGL(seqhash_kernel, dim3((n + 255) / 256), dim3(256), 0, st, text, recs, n, hashes);
    return hipGetLastError();
}

// =================================================================== launchers
static inline unsigned grid_for(uint64_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

// Persistent: one 1024-thread workgroup per CU -- on seven CUs out of eight for the stride-16 geometries.  Their screen is bound
// by HBM, not by CUs (alone, 192 workgroups stream as fast as 256), and the CUs left over take the first workgroups of the
// next pass's screen and give the finish kernels a place where no streaming workgroup keeps the memory pipeline full
// (0.2205 -> 0.2166 ms per step).  The stride-8 screen is VALU bound and takes every CU.
uint64_t screen_grid_for(const ReadsView &R, int n_cu, int stride)
{
    const uint64_t n_chunks = R.n_vec / ((uint64_t)SCREEN_BLOCK * SCREEN_U);
    uint64_t wg = (uint64_t)n_cu;
    if (stride == 16 && n_cu >= 16) wg = (uint64_t)n_cu * SCREEN_CU_NUM / SCREEN_CU_DEN;
    if (stride == 4) wg = (uint64_t)n_cu * 2;          // (key 4: the front2-only screen of large baits, two workgroups a CU)
    return n_chunks < wg ? n_chunks : wg;
}
uint64_t screen_rec_cap_for(const ReadsView &R, int n_cu, int stride)
{   // records one workgroup can emit: one per lane per chunk it walks
    const uint64_t n_chunks = R.n_vec / ((uint64_t)SCREEN_BLOCK * SCREEN_U);
    const uint64_t grid = screen_grid_for(R, n_cu, stride);
    return grid ? (n_chunks + grid - 1) / grid * SCREEN_BLOCK : 0;
}

// Launch with or without kernel-attached timing events: hipExtLaunchKernelGGL stamps the events with the
// dispatch's own begin and end, so the elapsed time is the kernel's duration (what rocprofv3 reports) and no
// extra packet sits between two kernels of the pass.
#define MF_LAUNCH(kernel, grid, block, lds, st, tm, ...)                                                             \
    do {                                                                                                             \
        if (tm) hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)(lds), st, (tm)->start, (tm)->stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                           \
    } while (0)

// the dynamic-LDS ceiling of a kernel is raised once per (kernel, device), not on every launch
template <auto Kernel> static void raise_lds_limit_once(size_t max_bytes)
{
    static std::atomic<uint64_t> done{0};                     // one bit per device (0..63)
    int dev = 0; (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_bytes);
    done.fetch_or(bit, std::memory_order_release);
}

template <int SPW, int CANON>
static void launch_screen_spw(const ReadsView &R, const KmerSetView &S, void *recs, uint32_t *rec_counts, int n_cu, hipStream_t st,
                              const KernelTiming *tm, uint32_t *clear, uint64_t clear_vec4)
{
    const int key = screen_grid_key(S.stride, S.front_mode, S.canon);
    const uint64_t grid = screen_grid_for(R, n_cu, key);
    if (grid == 0) return;
    const uint32_t cap = (uint32_t)screen_rec_cap_for(R, n_cu, key);
    if constexpr (SPW == 1) if (S.front_mode == 3) {          // LDS table; lone positives queued and looked up in front2 sixty-four at a time (screen3_kernel)
        const size_t lds = (sizeof(uint32_t) << S.bloom_log2w) + S3_LDS_EXTRA;
        raise_lds_limit_once<&screen3_kernel<SCREEN_U, CANON>>(128 * 1024 + S3_LDS_EXTRA);
        MF_LAUNCH((screen3_kernel<SCREEN_U, CANON>), dim3((unsigned)grid), dim3(SCREEN_BLOCK), lds, st, tm, R, S,
                  static_cast<ScreenRec *>(recs), cap, rec_counts, reinterpret_cast<uint4 *>(clear), clear ? clear_vec4 : (uint64_t)0);
        return;
    }
    if (S.front_mode == 1 || S.front_mode == 3) {          // LDS table, its positives through front2 turn by turn (screen2_kernel; mode 3 falls here for the stride-8 geometries)
        const size_t lds = (sizeof(uint32_t) << S.bloom_log2w) + S2_LDS_EXTRA;
        raise_lds_limit_once<&screen2_kernel<SPW, SCREEN_U, true, 8, false, CANON>>(128 * 1024 + S2_LDS_EXTRA);
        MF_LAUNCH((screen2_kernel<SPW, SCREEN_U, true, 8, false, CANON>), dim3((unsigned)grid), dim3(SCREEN_BLOCK), lds, st, tm, R, S,
                  static_cast<ScreenRec *>(recs), cap, rec_counts, reinterpret_cast<uint4 *>(clear), clear ? clear_vec4 : (uint64_t)0);
        return;
    }
    if (S.front_mode == 4) {          // a one-bit LDS table, its positives through front2 (and front3): one workgroup a CU, eight gathers in flight per lane
        const size_t lds = (sizeof(uint32_t) << S.pre_log2w) + S2_LDS_EXTRA;
        raise_lds_limit_once<&screen2_kernel<SPW, SCREEN_U, false, 8, true, CANON>>(128 * 1024 + S2_LDS_EXTRA);
        MF_LAUNCH((screen2_kernel<SPW, SCREEN_U, false, 8, true, CANON>), dim3((unsigned)grid), dim3(SCREEN_BLOCK), lds, st, tm, R, S,
                  static_cast<ScreenRec *>(recs), cap, rec_counts, reinterpret_cast<uint4 *>(clear), clear ? clear_vec4 : (uint64_t)0);
        return;
    }
    if (S.front_mode == 2) {          // every sample through front2 (and front3): two workgroups a CU (its grid key says so), four gathers in flight per lane
        MF_LAUNCH((screen2_kernel<SPW, SCREEN_U, false, 4, false, CANON>), dim3((unsigned)grid), dim3(SCREEN_BLOCK), S2_LDS_EXTRA, st, tm, R, S,
                  static_cast<ScreenRec *>(recs), cap, rec_counts, reinterpret_cast<uint4 *>(clear), clear ? clear_vec4 : (uint64_t)0);
        return;
    }
    const size_t lds1 = (sizeof(uint32_t) << S.bloom_log2w) + 16;
    raise_lds_limit_once<&screen_kernel<SPW, SCREEN_U, CANON>>(128 * 1024 + 16);
    MF_LAUNCH((screen_kernel<SPW, SCREEN_U, CANON>), dim3((unsigned)grid), dim3(SCREEN_BLOCK), lds1, st, tm, R, S,
              static_cast<ScreenRec *>(recs), cap, rec_counts, reinterpret_cast<uint4 *>(clear), clear ? clear_vec4 : (uint64_t)0);
}

template <int SPW>
static void launch_mark_spw(const ReadsView &R, const KmerSetView &S, const void *recs, const uint32_t *rec_counts, uint32_t *cand, int n_cu,
                            hipStream_t st, const KernelTiming *tm)
{
    const int key = screen_grid_key(S.stride, S.front_mode, S.canon);
    const uint64_t grid = screen_grid_for(R, n_cu, key);
    if (grid == 0) return;
    const uint32_t cap = (uint32_t)screen_rec_cap_for(R, n_cu, key);
    static const uint32_t split = getenv("MF_MARK_SPLIT") ? (uint32_t)atoi(getenv("MF_MARK_SPLIT")) : (uint32_t)MARK_SPLIT;
    static const uint32_t mblock = getenv("MF_MARK_BLOCK") ? (uint32_t)atoi(getenv("MF_MARK_BLOCK")) : (uint32_t)MARK_BLOCK;
    MF_LAUNCH((mark_kernel<SPW, SCREEN_U>), dim3((unsigned)grid * split), dim3(mblock), 0, st, tm, R, S,
              static_cast<const ScreenRec *>(recs), cap, rec_counts, (uint32_t)SCREEN_BLOCK, split, cand);
}

hipError_t launch_screen(const ReadsView &R, const KmerSetView &S, void *recs, uint32_t *rec_counts, int n_cu, hipStream_t st,
                         const KernelTiming *tm, uint32_t *clear, uint64_t clear_vec4)
{
    // (canonical keys: KmerSetView::canon is 1 for sixteen-base samples, 2 for shorter ones)
    if (S.stride == 16 && S.canon == 1) launch_screen_spw<1, 1>(R, S, recs, rec_counts, n_cu, st, tm, clear, clear_vec4);
    else if (S.stride == 16 && S.canon) launch_screen_spw<1, 2>(R, S, recs, rec_counts, n_cu, st, tm, clear, clear_vec4);
    else if (S.stride == 16) launch_screen_spw<1, 0>(R, S, recs, rec_counts, n_cu, st, tm, clear, clear_vec4);
    else if (S.canon) launch_screen_spw<2, 2>(R, S, recs, rec_counts, n_cu, st, tm, clear, clear_vec4);
    else launch_screen_spw<2, 0>(R, S, recs, rec_counts, n_cu, st, tm, clear, clear_vec4);
    return hipGetLastError();
}

hipError_t launch_finish(const ReadsView &R, const KmerSetView &S, const void *recs, const uint32_t *rec_counts, uint32_t *bits,
                         unsigned long long *partials, int n_cu, hipStream_t st, const KernelTiming *tm, hipEvent_t done, uint32_t *cand)
{
    const int key = screen_grid_key(S.stride, S.front_mode, S.canon);
    const uint64_t lists = screen_grid_for(R, n_cu, key);
    if (lists == 0) {               // (an empty read set) nothing to settle: the tallies read zero and `done` still completes, as in launch_exact
        (void)hipMemsetAsync(partials, 0, 3 * EXACT_MAX_GRID * 16, st);
        if (done) (void)hipEventRecord(done, st);
        return hipGetLastError();
    }
    const uint32_t cap = (uint32_t)screen_rec_cap_for(R, n_cu, key);
    const ScreenRec *rc = static_cast<const ScreenRec *>(recs);
    static const bool time_phase1 = getenv("MF_TIME_PHASE1") != nullptr;          // the one event pair goes to phase 0 unless asked otherwise
    const KernelTiming *none = nullptr, *tm0 = time_phase1 ? none : tm, *tm1 = time_phase1 ? tm : none;
    // `done` rides on the last dispatch as its completion event (no marker packet of its own in the stream)
    KernelTiming last{nullptr, done};
    bool done_attached = false;
    if (done && !tm1) { tm1 = &last; done_attached = true; }
    static const unsigned fgrid = std::min<unsigned>(EXACT_MAX_GRID, std::max<unsigned>(64, getenv("MF_FINISH_GRID") ? (unsigned)atoi(getenv("MF_FINISH_GRID")) : FINISH_GRID));
#define MF_LAUNCH_FINISH(SPW, KW) do { \
        MF_LAUNCH((finish_kernel<SPW, SCREEN_U, KW, 0>), dim3(fgrid), dim3(FINISH_BLOCK), 0, st, tm0, R, S, rc, cap, rec_counts, \
                  (uint32_t)lists, (uint32_t)SCREEN_BLOCK, bits, partials, cand); \
        MF_LAUNCH((finish_kernel<SPW, SCREEN_U, KW, 1>), dim3(fgrid), dim3(FINISH_BLOCK), 0, st, tm1, R, S, rc, cap, rec_counts, \
                  (uint32_t)lists, (uint32_t)SCREEN_BLOCK, bits, partials, cand); } while (0)
    if (S.stride == 16) { if (S.kw == 1) MF_LAUNCH_FINISH(1, 1); else MF_LAUNCH_FINISH(1, 2); }
    else                { if (S.kw == 1) MF_LAUNCH_FINISH(2, 1); else MF_LAUNCH_FINISH(2, 2); }
#undef MF_LAUNCH_FINISH
    if (done && !done_attached) (void)hipEventRecord(done, st);
    return hipGetLastError();
}

hipError_t launch_mark(const ReadsView &R, const KmerSetView &S, const void *recs, const uint32_t *rec_counts, uint32_t *cand, int n_cu,
                       hipStream_t st, const KernelTiming *tm)
{
    if (S.stride == 16) launch_mark_spw<1>(R, S, recs, rec_counts, cand, n_cu, st, tm);
    else launch_mark_spw<2>(R, S, recs, rec_counts, cand, n_cu, st, tm);
    return hipGetLastError();
}

hipError_t launch_exact(const ReadsView &R, const KmerSetView &S_, uint32_t *cand, uint32_t thr, bool count_all,
                        uint32_t *out_bits, uint32_t *hits_out, unsigned long long *partials, int n_cu, hipStream_t st,
                        const KernelTiming *tm, bool coresident, hipEvent_t done, bool merge)
{
    const uint64_t n_bw = (R.n_reads + 31) >> 5;
    const uint64_t n_wc = (n_bw + WC_WORDS - 1) / WC_WORDS;
    if (n_wc == 0) { if (done) (void)hipEventRecord(done, st); return hipGetLastError(); }
    // co-resident form (pipelined passes): half the threads and the folded bit table, so that a workgroup fits beside the
    // screen workgroup of the next pass on every CU (32 KiB of LDS, eight wave slots, a quarter of the registers are free there)
    KmerSetView S = S_;
    if (coresident && !count_all) { S.kbloom = S_.kbloom_co; S.kb_log2w = S_.kb_co_log2w; } else coresident = false;
    const int waves = (coresident ? EXACT_BLOCK_CO : EXACT_BLOCK) / 64;
    const size_t lds = (sizeof(uint32_t) << S.kb_log2w) + (size_t)(waves * WC_WORDS + waves * 64) * sizeof(uint32_t);
    const int per_cu = coresident ? 1 : 2 * lds <= 160 * 1024 ? 2 : 1;       // persistent: 1-2 workgroups per CU
    uint64_t grid = (n_wc + waves - 1) / waves;
    if (grid > (uint64_t)n_cu * per_cu) grid = (uint64_t)n_cu * per_cu;
    if (grid > EXACT_MAX_GRID) grid = EXACT_MAX_GRID;
    // `done` rides on the dispatch as its completion event unless a timing pair is wanted
    KernelTiming with_done{nullptr, done};
    const KernelTiming *tmx = tm ? tm : (done ? &with_done : nullptr);
#define MF_LAUNCH_EXACT(KW, CA, BLK) do { \
        raise_lds_limit_once<&exact_kernel<KW, CA, BLK>>(160 * 1024); \
        MF_LAUNCH((exact_kernel<KW, CA, BLK>), dim3((unsigned)grid), dim3(BLK), lds, st, tmx, R, S, cand, thr, out_bits, hits_out, partials, merge ? 1u : 0u); } while (0)
    if (coresident) { if (S.kw == 1) MF_LAUNCH_EXACT(1, false, EXACT_BLOCK_CO); else MF_LAUNCH_EXACT(2, false, EXACT_BLOCK_CO); }
    else if (S.kw == 1) { if (count_all) MF_LAUNCH_EXACT(1, true, EXACT_BLOCK); else MF_LAUNCH_EXACT(1, false, EXACT_BLOCK); }
    else                { if (count_all) MF_LAUNCH_EXACT(2, true, EXACT_BLOCK); else MF_LAUNCH_EXACT(2, false, EXACT_BLOCK); }
#undef MF_LAUNCH_EXACT
    if (done && tm) (void)hipEventRecord(done, st);
    return hipGetLastError();
}

hipError_t launch_build_kbloom(const uint64_t *keys, uint64_t slots, int kw, uint32_t *kbloom, uint32_t kb_log2w, hipStream_t st)
{
    hipLaunchKernelGGL(build_kbloom_kernel, dim3(grid_for(slots, 256)), dim3(256), 0, st, keys, slots, kw, kbloom, kb_log2w);
    return hipGetLastError();
}

hipError_t launch_build_table(const BaitView &B, int k, int kw, uint64_t *keys, uint64_t slots, uint32_t *postab_scratch, hipStream_t st)
{
    if (B.total == 0) return hipSuccess;
    const unsigned grid = grid_for(B.total, 256);
    if (kw == 1) {
        hipLaunchKernelGGL(build_keys1_kernel, dim3(grid), dim3(256), 0, st, B, k, keys, slots - 1);
    } else {
        hipLaunchKernelGGL(build_pos2_kernel, dim3(grid), dim3(256), 0, st, B, k, postab_scratch, slots - 1);
        hipLaunchKernelGGL(materialize2_kernel, dim3(grid_for(slots, 256)), dim3(256), 0, st, B, k, postab_scratch, slots, keys);
    }
    return hipGetLastError();
}

hipError_t launch_build_screen(const BaitView &B, int s, uint32_t *bloom, uint32_t log2w, uint32_t log2w2, uint32_t *stab,
                               uint32_t stab_slots, uint32_t *has_ones, uint32_t *front2, uint32_t f2_log2b, uint32_t *front3, uint32_t f3_log2b,
                               uint32_t *pre, uint32_t pre_log2w, bool canon, hipStream_t st)
{
    if (B.total == 0) return hipSuccess;
    hipLaunchKernelGGL(build_screen_kernel, dim3(grid_for(B.total, 256)), dim3(256), 0, st, B, s, bloom, log2w, log2w2, stab, stab_slots - 1, has_ones,
                       front2, f2_log2b, front3, f3_log2b, pre, pre_log2w, canon);
    return hipGetLastError();
}

hipError_t launch_count_keys(const uint64_t *keys, uint64_t slots, int kw, const uint32_t *stab, uint64_t stab_slots,
                             unsigned long long *out2, hipStream_t st)
{
    const uint64_t n = slots > stab_slots ? slots : stab_slots;
    hipLaunchKernelGGL(count_keys_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, keys, slots, kw, stab, stab_slots, out2);
    return hipGetLastError();
}

hipError_t launch_build_npos_blk(const uint64_t *npos, uint64_t n_npos, uint64_t n_blk, uint32_t *blk, hipStream_t st)
{
    hipLaunchKernelGGL(build_npos_blk_kernel, dim3(grid_for(n_blk, 256)), dim3(256), 0, st, npos, n_npos, n_blk, blk);
    return hipGetLastError();
}

hipError_t launch_build_off_blk(const uint64_t *offsets, uint64_t n_reads, uint64_t n_blk, uint64_t *blk, hipStream_t st)
{
    hipLaunchKernelGGL(build_off_blk_kernel, dim3(grid_for(n_blk, 256)), dim3(256), 0, st, offsets, n_reads, n_blk, blk);
    return hipGetLastError();
}

hipError_t launch_mark_has_n(const ReadsView &R, uint32_t *has_n, hipStream_t st)
{
    if (R.n_npos == 0) return hipSuccess;
    hipLaunchKernelGGL(mark_has_n_kernel, dim3(grid_for(R.n_npos, 256)), dim3(256), 0, st, R, has_n);
    return hipGetLastError();
}

} // namespace mf
