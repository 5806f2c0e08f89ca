// HIP kernels of libmitofilter_hip for gfx950 (MI355X, CDNA4; 64-wide waves).
//
//   screen_kernel   streams the dense 2-bit read stream once from HBM with
//                   16-byte coalesced loads, probes stream-aligned s-mers in a
//                   blocked bit table held in LDS, verifies the few positives
//                   against an exact s-mer table in L2 and marks candidate
//                   reads.  Pure integer/indexing work, HBM-bound by design.
//   exact_kernel    one wave per candidate read: every lane extracts one
//                   k-mer, canonicalises it, probes the open-address bait
//                   table; hits are reduced with wave ballot + popcount and
//                   compared with the threshold.  With no candidate bitmap it
//                   scans every read (exhaustive mode).
//   build_*         device-side bait set builder (history-independent table).
//
// Why the screen is exact (not a heuristic): a read can only have a k-mer hit
// if some window of k bases equals a bait k-mer (either strand).  That window
// fully contains a stream-aligned s-mer (k >= s + stride - 1), and that s-mer,
// read as it lies in the stream, is an s-mer of the bait or of its reverse
// complement -- both are in the s-mer set.  So "no sampled s-mer of the read
// is in the set" proves hits == 0 < T.  Reads that survive the screen get the
// full per-k-mer count, so the emitted bits equal the brute-force oracle's.
#include "mf_common.h"
#include "mf_kernels.h"

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

// read index holding global base g, or ~0 if the s bases from g do not lie in one read
__device__ __forceinline__ uint64_t read_holding(const ReadsView &R, uint64_t g, uint32_t s)
{
    if (g + s > R.total_bases) return ~0ULL;
    if (R.uniform_len) {
        uint64_t r = g / R.uniform_len;
        uint64_t off = g - r * R.uniform_len;
        return off + s <= R.uniform_len ? r : ~0ULL;
    }
    // first offset > g, minus one
    uint64_t lo = 0, hi = R.n_reads + 1;
    while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (R.offsets[mid] <= g) lo = mid + 1; else hi = mid; }
    uint64_t r = lo - 1;
    return g + s <= R.offsets[r + 1] ? r : ~0ULL;
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
// SPW = samples per u32 word (1: stride 16 bases, 2: stride 8 bases)
// U   = uint4 loads in flight per lane per chunk
template <int SPW, int U>
__global__ void __launch_bounds__(1024)
screen_kernel(ReadsView R, KmerSetView S, uint32_t *__restrict__ cand)
{
    extern __shared__ uint32_t s_bloom[];
    {
        const uint32_t nb4 = (1u << S.bloom_log2w) >> 2;
        const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(S.bloom);
        uint4 *dst = reinterpret_cast<uint4 *>(s_bloom);
        for (uint32_t i = threadIdx.x; i < nb4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();

    const uint4 *__restrict__ w4 = reinterpret_cast<const uint4 *>(R.words);
    const uint64_t chunk = (uint64_t)blockDim.x * U;
    const uint64_t n_chunks = R.n_vec / chunk;          // n_vec is padded to a whole number of chunks
    const uint32_t shift = 32 - S.bloom_log2w;
    const uint32_t smask = S.smask;

    uint4 cur[U]; uint32_t curx[U];
    uint64_t c = blockIdx.x;
    if (c < n_chunks) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t v = c * chunk + (uint64_t)u * blockDim.x + threadIdx.x;
            cur[u] = w4[v];
            if (SPW == 2) curx[u] = R.words[4 * v + 4];
        }
    }
    for (; c < n_chunks; c += gridDim.x) {
        // prefetch the next chunk before touching this one
        uint4 nxt[U]; uint32_t nxtx[U];
        const uint64_t cn = c + gridDim.x;
        if (cn < n_chunks) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint64_t v = cn * chunk + (uint64_t)u * blockDim.x + threadIdx.x;
                nxt[u] = w4[v];
                if (SPW == 2) nxtx[u] = R.words[4 * v + 4];
            }
        }
        uint32_t hitmask = 0;    // bit (u*4+q)*SPW+j
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t wv[5] = {cur[u].x, cur[u].y, cur[u].z, cur[u].w, SPW == 2 ? curx[u] : 0u};
#pragma unroll
            for (int q = 0; q < 4; q++) {
#pragma unroll
                for (int j = 0; j < SPW; j++) {
                    uint32_t sm = (SPW == 1) ? wv[q] : (alignbit(wv[q + 1], wv[q], 16u * j) & smask);
                    const uint32_t h = bloom_hash(sm);
                    const uint32_t bw = s_bloom[h >> shift];
                    const uint32_t g = h ^ (h >> 15);
                    const uint32_t m = (1u << (g & 31)) | (1u << ((g >> 5) & 31));
                    hitmask |= ((bw & m) == m) ? (1u << ((u * 4 + q) * SPW + j)) : 0u;
                }
            }
        }
        // rare: verify LDS positives against the exact s-mer table, then mark the read
        while (hitmask) {
            const int idx = __ffs(hitmask) - 1;
            hitmask &= hitmask - 1;
            const int j = idx % SPW, q = (idx / SPW) & 3, u = idx / (4 * SPW);
            const uint64_t wi = 4 * (c * chunk + (uint64_t)u * blockDim.x + threadIdx.x) + q;
            uint32_t sm = R.words[wi];
            if (SPW == 2) sm = alignbit(R.words[wi + 1], sm, 16u * j) & smask;
            if (stab_contains(S, sm)) {
                const uint64_t g0 = wi * 16 + (uint64_t)j * 8;
                const uint64_t r = read_holding(R, g0, (uint32_t)S.s);
                if (r != ~0ULL) atomicOr(&cand[r >> 5], 1u << (r & 31));
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) { cur[u] = nxt[u]; if (SPW == 2) curx[u] = nxtx[u]; }
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

template <int KW, bool COUNT_ALL>
__device__ __forceinline__ uint32_t read_hits(const ReadsView &R, const KmerSetView &S, uint64_t r, uint32_t thr, int lane)
{
    uint64_t b0, len;
    if (R.uniform_len) { b0 = r * R.uniform_len; len = R.uniform_len; }
    else { b0 = R.offsets[r]; len = R.offsets[r + 1] - b0; }
    const int k = S.k;
    if (len < (uint64_t)k) return 0;
    const uint64_t n_pos = len - k + 1;
    const bool hasn = (R.has_n[r >> 5] >> (r & 31)) & 1u;
    uint64_t n_lo = 0, n_hi = 0;
    if (hasn) { n_lo = lower_bound_u64(R.npos, R.n_npos, b0); n_hi = lower_bound_u64(R.npos, R.n_npos, b0 + len); }
    uint32_t hits = 0;
    for (uint64_t p0 = 0; p0 < n_pos; p0 += 64) {
        const uint64_t p = p0 + lane;
        const bool active = p < n_pos;
        const uint64_t g = b0 + (active ? p : 0);
        const Key<KW> key = canonical_at<KW>(R.words, g, k);
        bool found = active && table_contains(S, key);
        if (hasn) {
            for (uint64_t i = n_lo; i < n_hi; i++) {
                const uint64_t np = R.npos[i];
                if (np >= g && np < g + k) found = false;
            }
        }
        hits += (uint32_t)__popcll(__ballot(found));
        if (!COUNT_ALL && hits >= thr) break;
    }
    return hits;
}

template <int KW, bool COUNT_ALL>
__global__ void __launch_bounds__(256)
exact_kernel(ReadsView R, KmerSetView S, const uint32_t *__restrict__ cand, uint32_t thr,
             uint32_t *__restrict__ out_bits, uint32_t *__restrict__ hits_out, unsigned long long *__restrict__ counters)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_bw = (R.n_reads + 31) >> 5;
    const uint64_t wbase = wave * 64;
    if (wbase >= n_bw) return;
    const uint64_t myw = wbase + lane;
    uint32_t cw = 0;
    if (myw < n_bw) {
        cw = cand ? cand[myw] : 0xFFFFFFFFu;
        const uint64_t rem = R.n_reads - myw * 32;
        if (rem < 32) cw &= (1u << rem) - 1;
    }
    uint32_t res = 0;
    uint64_t lanes = __ballot(cw != 0);
    while (lanes) {
        const int src = __ffsll((unsigned long long)lanes) - 1;
        lanes &= lanes - 1;
        uint32_t bits = __builtin_amdgcn_readlane(cw, src);
        while (bits) {
            const int b = __ffs(bits) - 1;
            bits &= bits - 1;
            const uint64_t r = (wbase + src) * 32 + b;
            const uint32_t h = read_hits<KW, COUNT_ALL>(R, S, r, thr, lane);
            if (COUNT_ALL && lane == 0) hits_out[r] = h;
            if (h >= thr && lane == src) res |= 1u << b;
        }
    }
    if (myw < n_bw) out_bits[myw] = res;
    // pass / candidate tallies: one atomic per wave
    const uint32_t np = __popc(res), nc = __popc(cw);
    uint32_t sp = np, sc = nc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sp += __shfl_down(sp, o); sc += __shfl_down(sc, o); }
    if (lane == 0 && (sp | sc)) {
        if (sp) atomicAdd(&counters[0], (unsigned long long)sp);
        if (sc) atomicAdd(&counters[1], (unsigned long long)sc);
    }
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

__device__ __forceinline__ void screen_insert(uint32_t sm, uint32_t *bloom, uint32_t log2w, uint32_t *stab, uint32_t stab_mask,
                                              uint32_t *has_ones)
{
    const uint32_t h = bloom_hash(sm);
    const uint32_t g = h ^ (h >> 15);
    atomicOr(&bloom[h >> (32 - log2w)], (1u << (g & 31)) | (1u << ((g >> 5) & 31)));
    if (sm == EMPTY32) { *has_ones = 1u; return; }
    uint32_t v = sm, slot = smer_hash(sm) & stab_mask;
    for (;;) {
        const uint32_t old = atomicMin(&stab[slot], v);
        if (old == v) return;
        if (old > v) v = old;
        if (v == EMPTY32) return;
        slot = (slot + 1) & stab_mask;
    }
}

__global__ void build_screen_kernel(BaitView B, int s, uint32_t *bloom, uint32_t log2w, uint32_t *stab, uint32_t stab_mask,
                                    uint32_t *has_ones)
{
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B.total || B.runlen[p] < s) return;
    const uint64_t bit = 2 * p; const uint64_t wi = bit >> 5; const uint32_t sh = (uint32_t)bit & 31;
    uint32_t fwd = alignbit(B.words[wi + 1], B.words[wi], sh);
    if (s < 16) fwd &= (1u << (2 * s)) - 1;
    screen_insert(fwd, bloom, log2w, stab, stab_mask, has_ones);
    screen_insert(revcomp_s(fwd, s), bloom, log2w, stab, stab_mask, has_ones);
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

// mark reads that hold an invalid base (one thread per invalid position)
__global__ void mark_has_n_kernel(ReadsView R, uint32_t *has_n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R.n_npos) return;
    const uint64_t r = read_holding(R, R.npos[i], 1);
    if (r != ~0ULL) atomicOr(&has_n[r >> 5], 1u << (r & 31));
}

// =================================================================== launchers
static inline unsigned grid_for(uint64_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

hipError_t launch_screen(const ReadsView &R, const KmerSetView &S, uint32_t *cand, int n_cu, hipStream_t st)
{
    const size_t lds = sizeof(uint32_t) << S.bloom_log2w;
    int blocks_per_cu = (int)((160 * 1024) / (lds ? lds : 1));
    if (blocks_per_cu > 2) blocks_per_cu = 2;          // 2 x 1024 threads = 32 waves/CU
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    uint64_t n_chunks = R.n_vec / ((uint64_t)SCREEN_BLOCK * SCREEN_U);
    uint64_t grid = (uint64_t)n_cu * blocks_per_cu;
    if (grid > n_chunks) grid = n_chunks;
    if (grid == 0) return hipSuccess;
    if (S.stride == 16) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&screen_kernel<1, SCREEN_U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((screen_kernel<1, SCREEN_U>), dim3((unsigned)grid), dim3(SCREEN_BLOCK), lds, st, R, S, cand);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&screen_kernel<2, SCREEN_U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((screen_kernel<2, SCREEN_U>), dim3((unsigned)grid), dim3(SCREEN_BLOCK), lds, st, R, S, cand);
    }
    return hipGetLastError();
}

hipError_t launch_exact(const ReadsView &R, const KmerSetView &S, const uint32_t *cand, uint32_t thr, bool count_all,
                        uint32_t *out_bits, uint32_t *hits_out, unsigned long long *counters, hipStream_t st)
{
    const uint64_t n_bw = (R.n_reads + 31) >> 5;
    const uint64_t waves = (n_bw + 63) / 64;
    if (waves == 0) return hipSuccess;
    const unsigned grid = grid_for(waves * 64, 256);
#define MF_LAUNCH_EXACT(KW, CA) hipLaunchKernelGGL((exact_kernel<KW, CA>), dim3(grid), dim3(256), 0, st, R, S, cand, thr, out_bits, hits_out, counters)
    if (S.kw == 1) { if (count_all) MF_LAUNCH_EXACT(1, true); else MF_LAUNCH_EXACT(1, false); }
    else           { if (count_all) MF_LAUNCH_EXACT(2, true); else MF_LAUNCH_EXACT(2, false); }
#undef MF_LAUNCH_EXACT
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

hipError_t launch_build_screen(const BaitView &B, int s, uint32_t *bloom, uint32_t log2w, uint32_t *stab, uint32_t stab_slots,
                               uint32_t *has_ones, hipStream_t st)
{
    if (B.total == 0) return hipSuccess;
    hipLaunchKernelGGL(build_screen_kernel, dim3(grid_for(B.total, 256)), dim3(256), 0, st, B, s, bloom, log2w, stab, stab_slots - 1, has_ones);
    return hipGetLastError();
}

hipError_t launch_count_keys(const uint64_t *keys, uint64_t slots, int kw, const uint32_t *stab, uint64_t stab_slots,
                             unsigned long long *out2, hipStream_t st)
{
    const uint64_t n = slots > stab_slots ? slots : stab_slots;
    hipLaunchKernelGGL(count_keys_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, keys, slots, kw, stab, stab_slots, out2);
    return hipGetLastError();
}

hipError_t launch_mark_has_n(const ReadsView &R, uint32_t *has_n, hipStream_t st)
{
    if (R.n_npos == 0) return hipSuccess;
    hipLaunchKernelGGL(mark_has_n_kernel, dim3(grid_for(R.n_npos, 256)), dim3(256), 0, st, R, has_n);
    return hipGetLastError();
}

} // namespace mf
