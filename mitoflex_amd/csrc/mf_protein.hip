// Protein-space baiting for gfx950 (SURVEY.md 8f "next" #4): reads are translated in six frames on
// the fly and their peptide k-mers looked up in the peptide k-mer set of a protein database such
// as the reference's profile/MT_database/*.fa (which the reference only ever gives to tblastn,
// annotation/annotation_tookit.py:55-97).  The semantics ("Spec P") are written down in DESIGN.md.
//
// Work mapping.  A codon phase of a read (codon starts p, p+3, p+6, ...) serves one forward frame
// and one reverse frame at once: the forward residue is lut[codon], the reverse-strand residue is
// lut[revcomp(codon)], and the reverse frame's peptide runs towards lower positions -- so the two
// keys roll in opposite directions over the same codon stream, exactly like a k-mer and its reverse
// complement.  Three lanes per read (one per phase), 256 reads per workgroup iteration, no idle
// lanes.  The codon table (64 entries, residues pre-shifted to the key's top position) and, when it
// fits, the key bit table in front of the open-address table live in LDS; the table itself is
// read through L2 only for bit-table positives.
//
// Validity costs nothing in the loop: a stop codon or a codon holding an invalid base translates to
// the non-residue code 31, and both keys start out as all 31s, so a window that is incomplete or
// broken carries a 31 somewhere and can never equal a database key (codes 0..19 only).
//
// The kernel is VALU-bound (296 codon-strand steps per 150-base read), so the loop is kept short:
// a three-instruction hash for the bit table (the table hash is computed for positives only), a
// two-bit test in one 64-bit block.
#include "mf_kernels.h"
#include <hip/hip_ext.h>

namespace mf {

constexpr int PF_BLOCK = 768;                 // 12 waves
constexpr int PF_READS = PF_BLOCK / 3;        // 256 reads = 8 bitmap words per iteration

__device__ __forceinline__ uint64_t lower_bound_npos(const uint64_t *__restrict__ a, uint64_t n, uint64_t v)
{
    uint64_t lo = 0, hi = n;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

__device__ __forceinline__ uint32_t alignbit32(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }

// bit-table hash of a peptide key: one multiply per half and a xor (block index from the top bits); the two bit
// positions come from the low bits folded with the middle ones, which do not take part in the block index alone
__device__ __forceinline__ uint32_t pep_hash(uint64_t key) { return kbit_hash1(key); }
__device__ __forceinline__ uint32_t pep_bits(uint32_t hb) { return kbit_pos(hb); }

// bit-table test of one key: 1 when both bits of its 64-bit block are set
template <bool LDS_KB>
__device__ __forceinline__ uint32_t bit_test(const KmerSetView &S, const uint2 *__restrict__ s_kb2, uint32_t kb_shift, uint32_t hb)
{
    const uint2 blk = LDS_KB ? s_kb2[hb >> kb_shift] : reinterpret_cast<const uint2 *>(S.kbloom)[hb >> kb_shift];
    const uint32_t g = pep_bits(hb);
    return (blk.x >> (g & 31)) & (blk.y >> ((g >> 5) & 31)) & 1u;
}

__device__ __forceinline__ uint32_t table_has_key(const KmerSetView &S, uint64_t key)
{
    uint64_t slot = hash_key1(key) & S.slot_mask;
    uint64_t e = S.keys[slot];
    while (e < key) { slot = (slot + 1) & S.slot_mask; e = S.keys[slot]; }       // ordered table
    return e == key ? 1u : 0u;
}

template <bool LDS_KB, bool COUNT_ALL>
__global__ void __launch_bounds__(PF_BLOCK)
pfilter_kernel(ReadsView R, KmerSetView S, uint32_t thr, uint32_t *__restrict__ out_bits, uint32_t *__restrict__ hits_out,
               unsigned long long *__restrict__ partials)
{
    extern __shared__ uint4 s_dyn[];
    uint4 *s_lut = s_dyn;                          // 64 codon entries
    uint2 *s_kb2 = reinterpret_cast<uint2 *>(s_dyn + 64);   // key bit table, 64-bit blocks (LDS_KB)
    __shared__ uint32_t s_hits[PF_READS];
    __shared__ unsigned long long s_tot;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid < 64) s_lut[tid] = reinterpret_cast<const uint4 *>(S.plut)[tid];
    if (LDS_KB)
        for (uint32_t i = tid; i < (1u << (S.kb_log2w - 2)); i += PF_BLOCK)
            reinterpret_cast<uint4 *>(s_kb2)[i] = reinterpret_cast<const uint4 *>(S.kbloom)[i];
    if (tid == 0) s_tot = 0;
    const uint32_t kb_shift = 32 - (S.kb_log2w - 1);
    const uint32_t kp = (uint32_t)S.k;
    const uint64_t mask = (1ULL << (5 * kp)) - 1;
    const uint64_t top31 = 31ULL << (5 * (kp - 1));            // the non-residue code in the key's top position
    const uint32_t rl = tid / 3, phase = tid - 3 * rl;
    const uint64_t n_groups = (R.n_reads + PF_READS - 1) / PF_READS;

    for (uint64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        if (tid < PF_READS) s_hits[tid] = 0;
        __syncthreads();
        const uint64_t r = grp * PF_READS + rl;
        if (r < R.n_reads) {
            uint64_t b0, b1;
            if (R.uniform_len) { b0 = r * R.uniform_len; b1 = b0 + R.uniform_len; }
            else { b0 = R.offsets[r]; b1 = R.offsets[r + 1]; }
            const bool hasn = (R.has_n[r >> 5] >> (r & 31)) & 1u;
            // positions are kept relative to the 16-base word that holds the read's first base (32-bit arithmetic)
            const uint32_t *__restrict__ wp = R.words + (b0 >> 4);
            const uint64_t g0 = b0 & ~15ULL;
            uint32_t q = (uint32_t)(b0 & 15) + phase;
            const uint32_t q_end = (uint32_t)(b0 & 15) + (uint32_t)(b1 - b0);
            uint64_t ni = 0, next_n = ~0ULL;
            if (hasn) { ni = lower_bound_npos(R.npos, R.n_npos, g0 + q); next_n = ni < R.n_npos ? R.npos[ni] : ~0ULL; }
            uint64_t kf = mask, kr = mask;                                     // all non-residues: no window yet
            uint32_t cnt = 0, run_f = 0, run_r = 0;                            // run counters: bit table in L2 only
            while (q + 3 <= q_end) {
                const uint32_t wi = q >> 4;
                const uint32_t lo = wp[wi], hi = wp[wi + 1];                       // 32 bases from an aligned word
                do {
                    const uint32_t c = alignbit32(hi, lo, 2 * (q & 15)) & 63u;
                    const uint4 e = s_lut[c];
                    uint64_t ef = (uint64_t)e.x | ((uint64_t)e.y << 32);
                    uint32_t er = e.z, fl = e.w;
                    if (hasn) {
                        const uint64_t g = g0 + q;
                        while (next_n < g) { ni++; next_n = ni < R.n_npos ? R.npos[ni] : ~0ULL; }
                        if (next_n <= g + 2) { ef = top31; er = 31u; fl = 0; }   // an invalid base inside this codon
                    }
                    kf = (kf >> 5) | ef;
                    kr = ((kr << 5) | er) & mask;
                    uint32_t tf, tr;
                    if (LDS_KB) {          // LDS tests are cheaper than knowing whether the window is whole
                        tf = bit_test<true>(S, s_kb2, kb_shift, pep_hash(kf));
                        tr = bit_test<true>(S, s_kb2, kb_shift, pep_hash(kr));
                    } else {               // bit table in L2: only whole windows are worth a load
                        run_f = (fl & 1u) ? run_f + 1 : 0;
                        run_r = (fl & 2u) ? run_r + 1 : 0;
                        tf = run_f >= kp ? bit_test<false>(S, s_kb2, kb_shift, pep_hash(kf)) : 0u;
                        tr = run_r >= kp ? bit_test<false>(S, s_kb2, kb_shift, pep_hash(kr)) : 0u;
                    }
                    if (tf | tr) {                                       // rare: the open-address table decides
                        if (tf) cnt += table_has_key(S, kf);
                        if (tr) cnt += table_has_key(S, kr);
                    }
                    q += 3;
                } while (q + 3 <= q_end && (q >> 4) == wi);
                if (!COUNT_ALL && cnt >= thr) break;
            }
            if (cnt) atomicAdd(&s_hits[rl], cnt);
        }
        __syncthreads();
        if (tid < PF_READS) {                      // waves 0..3: one read per lane, two bitmap words per wave
            const uint64_t rr = grp * PF_READS + tid;
            const uint32_t h = s_hits[tid];
            const bool pass = rr < R.n_reads && h >= thr;
            if (COUNT_ALL && rr < R.n_reads) hits_out[rr] = h;
            const uint64_t bal = __ballot(pass);
            if (lane == 0) {
                const uint64_t w0 = grp * (PF_READS / 32) + wid * 2;
                out_bits[w0] = (uint32_t)bal; out_bits[w0 + 1] = (uint32_t)(bal >> 32);
                if (bal) atomicAdd(&s_tot, (unsigned long long)__popcll(bal));
            }
        }
        __syncthreads();
    }
    __syncthreads();
    // tallies: one plain store per workgroup, summed by the host (slot layout shared with the exact kernel)
    if (tid == 0) { partials[2 * blockIdx.x] = s_tot; partials[2 * blockIdx.x + 1] = 0; }
    if (blockIdx.x == 0)
        for (uint32_t i = 2 * gridDim.x + tid; i < 2 * EXACT_MAX_GRID; i += PF_BLOCK) partials[i] = 0;
}

// peptide k-mer table: one thread per database residue that starts a valid window
__global__ void build_pkeys_kernel(const uint8_t *__restrict__ aa, const uint8_t *__restrict__ runlen, uint64_t total, int kp,
                                   uint64_t *keys, uint64_t slot_mask)
{
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total || runlen[p] < kp) return;
    uint64_t v = 0;
    for (int i = 0; i < kp; i++) v |= (uint64_t)aa[p + i] << (5 * i);
    uint64_t slot = hash_key1(v) & slot_mask;
    for (;;) {                                     // history-independent linear probing: keep the smaller key, carry the larger
        const uint64_t old = atomicMin(reinterpret_cast<unsigned long long *>(&keys[slot]), (unsigned long long)v);
        if (old == v) return;
        if (old > v) v = old;
        if (v == EMPTY64) return;
        slot = (slot + 1) & slot_mask;
    }
}

// key bit table of a peptide set: one 64-bit block per key, one bit in each dword
__global__ void build_pbits_kernel(const uint64_t *__restrict__ keys, uint64_t slots, uint32_t *kbloom, uint32_t kb_log2w)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slots) return;
    const uint64_t key = keys[i];
    if (key == EMPTY64) return;
    const uint32_t hb = pep_hash(key), g = pep_bits(hb);
    uint32_t *blk = kbloom + 2 * (size_t)(hb >> (32 - (kb_log2w - 1)));
    atomicOr(&blk[0], 1u << (g & 31));
    atomicOr(&blk[1], 1u << ((g >> 5) & 31));
}

hipError_t launch_build_pbits(const uint64_t *keys, uint64_t slots, uint32_t *kbloom, uint32_t kb_log2w, hipStream_t st)
{
    hipLaunchKernelGGL(build_pbits_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, st, keys, slots, kbloom, kb_log2w);
    return hipGetLastError();
}

hipError_t launch_build_ptable(const uint8_t *aa, const uint8_t *runlen, uint64_t total, int kp, uint64_t *keys, uint64_t slots,
                               hipStream_t st)
{
    if (total == 0) return hipSuccess;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(build_pkeys_kernel, dim3(grid), dim3(256), 0, st, aa, runlen, total, kp, keys, slots - 1);
    return hipGetLastError();
}

hipError_t launch_pfilter(const ReadsView &R, const KmerSetView &S, uint32_t thr, bool count_all, uint32_t *out_bits,
                          uint32_t *hits_out, unsigned long long *counters, int n_cu, hipStream_t st, const KernelTiming *tm)
{
    const uint64_t n_groups = (R.n_reads + PF_READS - 1) / PF_READS;
    uint64_t grid = (uint64_t)2 * (n_cu > 0 ? n_cu : 256);            // two 12-wave workgroups per CU
    if (grid > (uint64_t)EXACT_MAX_GRID) grid = EXACT_MAX_GRID;
    if (grid > n_groups) grid = n_groups ? n_groups : 1;
    const size_t lds = 64 * sizeof(uint4) + (S.kb_in_lds ? (sizeof(uint32_t) << S.kb_log2w) : 0);
#define PF_LAUNCH(LK, CA)                                                                                          \
    do {                                                                                                           \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&pfilter_kernel<LK, CA>),               \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
        if (e_ != hipSuccess) return e_;                                                                           \
        if (tm) hipExtLaunchKernelGGL((pfilter_kernel<LK, CA>), dim3((unsigned)grid), dim3(PF_BLOCK), (uint32_t)lds, st, tm->start,    \
                                      tm->stop, 0, R, S, thr, out_bits, hits_out, counters);                        \
        else hipLaunchKernelGGL((pfilter_kernel<LK, CA>), dim3((unsigned)grid), dim3(PF_BLOCK), lds, st, R, S, thr, out_bits,        \
                                hits_out, counters);                                                               \
    } while (0)
    if (S.kb_in_lds) { if (count_all) PF_LAUNCH(true, true); else PF_LAUNCH(true, false); }
    else { if (count_all) PF_LAUNCH(false, true); else PF_LAUNCH(false, false); }
#undef PF_LAUNCH
    return hipGetLastError();
}

} // namespace mf
