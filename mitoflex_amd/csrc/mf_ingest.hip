// FASTQ text in device memory -> line index -> packed read set -> survivors (see mf_ingest.h).
// Every kernel here streams its input once at 16 bytes per lane; none of them is a hot spot next to the decoder that
// produces the text (mf_gzdev.hip) -- they exist so that the text never has to leave the device.
#include "mf_ingest.h"

namespace mf {
namespace {

__device__ __forceinline__ uint4 load16(const uint8_t *p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }   // (unaligned: one global_load_dwordx4)

// 0x80 in every byte of w that equals the byte replicated in c4 (exact per byte, no borrow between bytes)
__device__ __forceinline__ uint32_t eq_flags(uint32_t w, uint32_t c4)
{
    const uint32_t x = w ^ c4, t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | x | 0x7F7F7F7Fu);
}

// sum of v over the workgroup (blockDim.x a multiple of 64, at most 1024); result valid in every thread
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t *lds)
{
    for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d);
    const uint32_t w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[w] = v;
    __syncthreads();
    uint32_t s = 0;
    for (uint32_t i = 0; i < nw; i++) s += lds[i];
    return s;
}
// exclusive prefix of v over the workgroup in thread order; *total = the workgroup's sum
template <class T> __device__ __forceinline__ T block_exclusive(T v, T *lds, T *total)
{
    T inc = v;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int d = 1; d < 64; d <<= 1) { const T t = __shfl_up(inc, d); if (lane >= (uint32_t)d) inc += t; }
    __syncthreads();
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    T base = 0, sum = 0;
    for (uint32_t i = 0; i < nw; i++) { if (i < w) base += lds[i]; sum += lds[i]; }
    if (total) *total = sum;
    return base + inc - v;
}

// ------------------------------------------------------------------------------------------------------------ scan
// (256-thread workgroups: they find room beside whatever else is on a CU; a 1024-thread one waits for a quarter of a CU to come free)
constexpr uint32_t SCAN_BLOCK = 256, SCAN_ITEMS = 16, SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;
__global__ __launch_bounds__(SCAN_BLOCK) void scan_reduce_kernel(const uint32_t *in, uint64_t n, uint64_t *partial)
{
    __shared__ uint64_t lds[16];
    const uint64_t i0 = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t s = 0;
    for (uint32_t k = 0; k < SCAN_ITEMS; k++) if (i0 + k < n) s += in[i0 + k];
    uint64_t tot;
    (void)block_exclusive<uint64_t>(s, lds, &tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}
__global__ __launch_bounds__(SCAN_BLOCK) void scan_partials_kernel(uint64_t *partial, uint64_t nb)      // in place, exclusive; partial[nb] = total
{
    __shared__ uint64_t lds[16];
    uint64_t carry = 0;
    for (uint64_t b0 = 0; b0 < nb; b0 += SCAN_BLOCK) {
        const uint64_t i = b0 + threadIdx.x;
        const uint64_t v = i < nb ? partial[i] : 0;
        uint64_t tot;
        const uint64_t ex = block_exclusive<uint64_t>(v, lds, &tot);
        if (i < nb) partial[i] = carry + ex;
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[nb] = carry;
}
__global__ __launch_bounds__(SCAN_BLOCK) void scan_final_kernel(const uint32_t *in, uint64_t n, const uint64_t *partial, uint64_t *out, uint64_t nb)
{
    __shared__ uint64_t lds[16];
    const uint64_t i0 = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS]; uint64_t s = 0;
    for (uint32_t k = 0; k < SCAN_ITEMS; k++) { v[k] = i0 + k < n ? in[i0 + k] : 0; s += v[k]; }
    uint64_t run = partial[blockIdx.x] + block_exclusive<uint64_t>(s, lds, nullptr);
    for (uint32_t k = 0; k < SCAN_ITEMS; k++) { if (i0 + k < n) out[i0 + k] = run; run += v[k]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = partial[nb];
}

// ------------------------------------------------------------------------------------------------------------ lines
// newline flags of the 16 bytes at `off` (bytes at or behind n do not count): 0x80 per byte, four dwords
__device__ __forceinline__ void nl16(const uint8_t *text, uint64_t off, uint64_t n, uint32_t f[4])
{
    f[0] = f[1] = f[2] = f[3] = 0;
    if (off >= n) return;
    const uint4 v = load16(text + off);
    f[0] = eq_flags(v.x, 0x0A0A0A0Au); f[1] = eq_flags(v.y, 0x0A0A0A0Au); f[2] = eq_flags(v.z, 0x0A0A0A0Au); f[3] = eq_flags(v.w, 0x0A0A0A0Au);
    const uint64_t left = n - off;
    if (left < 16) for (uint32_t j = 0; j < 4; j++) {
        const uint32_t valid = left > 4 * j ? (uint32_t)(left - 4 * j) : 0;           // bytes of dword j in front of n
        if (valid < 4) f[j] &= valid ? (0xFFFFFFFFu >> (8 * (4 - valid))) : 0u;
    }
}
__global__ __launch_bounds__(256) void count_newlines_kernel(const uint8_t *text, uint64_t n, uint32_t *tile_cnt)
{
    __shared__ uint32_t lds[4];
    uint32_t f[4];
    nl16(text, (uint64_t)blockIdx.x * INGEST_TILE + threadIdx.x * 16, n, f);
    const uint32_t c = __popc(f[0]) + __popc(f[1]) + __popc(f[2]) + __popc(f[3]);
    const uint32_t s = block_sum(c, lds);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void line_starts_kernel(const uint8_t *text, uint64_t n, const uint64_t *tile_base, uint64_t *line_start)
{
    __shared__ uint32_t lds[4];
    const uint64_t off = (uint64_t)blockIdx.x * INGEST_TILE + threadIdx.x * 16;
    uint32_t f[4];
    nl16(text, off, n, f);
    const uint32_t c = __popc(f[0]) + __popc(f[1]) + __popc(f[2]) + __popc(f[3]);
    uint64_t k = tile_base[blockIdx.x] + block_exclusive<uint32_t>(c, lds, nullptr);      // newlines in front of this thread's bytes
    if (blockIdx.x == 0 && threadIdx.x == 0) line_start[0] = 0;
    if (!c) return;
    for (uint32_t j = 0; j < 4; j++)
        for (uint32_t m = f[j]; m; m &= m - 1) {
            const uint32_t byte = (uint32_t)(__ffs(m) - 1) >> 3;
            line_start[++k] = off + 4 * j + byte + 1;
        }
}
__global__ __launch_bounds__(256) void seq_lens_kernel(const uint8_t *text, const uint64_t *line_start, uint64_t n_rec, uint32_t *seq_len, uint32_t *minmax)
{
    __shared__ uint32_t lmin[4], lmax[4];
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t L = 0, mn = ~0u, mx = 0;
    if (r < n_rec) {
        const uint64_t s = line_start[4 * r + 1], e = line_start[4 * r + 2];
        uint64_t l = e - s - 1;
        if (l && text[e - 2] == '\r') l--;
        L = l > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)l;
        seq_len[r] = L; mn = mx = L;
    }
    for (int d = 32; d; d >>= 1) { mn = min(mn, (uint32_t)__shfl_xor(mn, d)); mx = max(mx, (uint32_t)__shfl_xor(mx, d)); }
    if ((threadIdx.x & 63) == 0) { lmin[threadIdx.x >> 6] = mn; lmax[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(&minmax[0], min(min(lmin[0], lmin[1]), min(lmin[2], lmin[3])));
        atomicMax(&minmax[1], max(max(lmax[0], lmax[1]), max(lmax[2], lmax[3])));
    }
}

// ------------------------------------------------------------------------------------------------------------ pack
// four text bytes -> four 2-bit codes in the low byte (first base lowest) + a 4-bit mask of the invalid ones.
// Bits 1-2 of a letter tell A, C, T, G apart and x ^ (x >> 1) puts them in order (the host packer's trick, mf_host.cpp).
__device__ __forceinline__ uint32_t pack4(uint32_t w, uint32_t &invalid4)
{
    const uint32_t u = w & 0xDFDFDFDFu;                                   // case folded
    const uint32_t ok = eq_flags(u, 0x41414141u) | eq_flags(u, 0x43434343u) | eq_flags(u, 0x47474747u) | eq_flags(u, 0x54545454u);
    uint32_t code = ((w >> 1) & 0x03030303u) ^ ((w >> 2) & 0x01010101u);
    code &= (ok >> 7) * 3u;                                               // invalid bases are stored as 0
    const uint32_t bad = ~ok & 0x80808080u;
    invalid4 = ((bad >> 7) | (bad >> 14) | (bad >> 21) | (bad >> 28)) & 0xFu;
    return (code | (code >> 6) | (code >> 12) | (code >> 18)) & 0xFFu;
}
template <int MODE>
__global__ __launch_bounds__(256) void pack_kernel(const uint8_t *text, const uint64_t *line_start, const uint64_t *offsets, uint32_t uniform_len,
                                                   uint64_t n_rec, uint64_t total_bases, uint64_t base, uint32_t *words, uint32_t *inv_cnt,
                                                   const uint64_t *inv_base, uint64_t *npos)
{
    // The batch's bases take the places base .. base + total_bases - 1 of the read set's stream (a read set is appended to batch
    // by batch); offsets are the batch's own (they start at 0).  One thread per word of the stream that the batch touches.
    __shared__ uint32_t lds[4];
    if (MODE == 1 && inv_cnt[blockIdx.x] == 0) return;                    // (nearly every workgroup: invalid bases are rare)
    const uint64_t w = (base >> 4) + (uint64_t)blockIdx.x * 256 + threadIdx.x, g0 = w * 16;
    const uint64_t lo_g = g0 > base ? g0 : base, hi_g = g0 + 16 < base + total_bases ? g0 + 16 : base + total_bases;
    uint32_t word = 0, inv = 0;
    if (lo_g < hi_g) {
        const uint64_t l0 = lo_g - base;                                  // first base of this word, counted in the batch
        const uint32_t k0 = (uint32_t)(lo_g - g0), nb = (uint32_t)(hi_g - lo_g);
        uint64_t r, pos, len;
        if (uniform_len) { r = l0 / uniform_len; pos = l0 - r * uniform_len; len = uniform_len; }
        else {
            uint64_t lo = 0, hi = n_rec;                                  // last r with offsets[r] <= l0 (an empty read never is: its successor has the same offset)
            while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (offsets[mid] <= l0) lo = mid; else hi = mid; }
            r = lo; pos = l0 - offsets[r]; len = offsets[r + 1] - offsets[r];
        }
        const uint8_t *src = text + line_start[4 * r + 1];
        if (nb == 16 && pos + 16 <= len) {                                // the whole word lies in one read: one 16-byte load
            const uint4 v = load16(src + pos);
            uint32_t i0, i1, i2, i3;
            word = pack4(v.x, i0) | (pack4(v.y, i1) << 8) | (pack4(v.z, i2) << 16) | (pack4(v.w, i3) << 24);
            inv = i0 | (i1 << 4) | (i2 << 8) | (i3 << 12);
        } else {
            for (uint32_t k = k0; k < k0 + nb; k++) {
                while (pos >= len) {                                      // next read that has bases
                    r++; pos = 0;
                    len = uniform_len ? uniform_len : offsets[r + 1] - offsets[r];
                    src = text + line_start[4 * r + 1];
                }
                uint32_t i1;
                const uint32_t c = pack4(src[pos], i1) & 3u;
                word |= c << (2 * k); inv |= (i1 & 1u) << k;
                pos++;
            }
        }
    }
    if (MODE == 0) {
        if (lo_g < hi_g) { if (g0 < base) words[w] |= word; else words[w] = word; }      // (the first word may be shared with the batch before: that one is complete)
        const uint32_t s = block_sum(__popc(inv), lds);
        if (threadIdx.x == 0) inv_cnt[blockIdx.x] = s;
    } else {
        uint64_t k = inv_base[blockIdx.x] + block_exclusive<uint32_t>(__popc(inv), lds, nullptr);
        for (uint32_t m = inv; m; m &= m - 1) npos[k++] = g0 + (uint32_t)(__ffs(m) - 1);
    }
}
__global__ __launch_bounds__(256) void add_base_kernel(uint64_t *dst, const uint64_t *src, uint64_t n, uint64_t base)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i] + base;
}

// ------------------------------------------------------------------------------------------------------------ survivors
__global__ __launch_bounds__(256) void store_bits_kernel(const uint32_t *batch_bits, uint64_t n_rec, uint32_t *file_bits, uint64_t rec_base)
{
    // one thread per destination word of the file-wide bitmap that the batch touches; the first and the last are shared with
    // the neighbouring batches (atomic), the ones in between are written whole
    const uint64_t w_first = rec_base >> 5, w_last = (rec_base + n_rec - 1) >> 5;
    const uint64_t w = w_first + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (w > w_last) return;
    const uint32_t sh = (uint32_t)rec_base & 31u;
    const int64_t j = (int64_t)(w - w_first);                             // destination word j takes batch bits [32 j - sh, 32 j - sh + 32)
    const uint64_t nw = (n_rec + 31) >> 5;
    const uint32_t a = j >= 1 && (uint64_t)(j - 1) < nw ? batch_bits[j - 1] : 0, b = (uint64_t)j < nw ? batch_bits[j] : 0;
    uint32_t v = sh ? (a >> (32 - sh)) | (b << sh) : b;
    // bits of the destination word that belong to this batch
    const uint64_t lo = w * 32 > rec_base ? w * 32 : rec_base, hi = (w + 1) * 32 < rec_base + n_rec ? (w + 1) * 32 : rec_base + n_rec;
    const uint32_t nbits = (uint32_t)(hi - lo), first = (uint32_t)(lo - w * 32);
    const uint32_t mask = nbits == 32 ? 0xFFFFFFFFu : (((1u << nbits) - 1) << first);
    v &= mask;
    if (nbits == 32) file_bits[w] = v; else if (v) atomicOr(&file_bits[w], v);
}

struct RecSpan { uint64_t h, s, q; uint32_t hl, sl, ql; };
__device__ __forceinline__ uint32_t line_len(const uint8_t *text, uint64_t a, uint64_t b)     // line [a, b - 1) less a CR in front of the LF
{
    uint64_t l = b - a - 1;
    if (l && text[b - 2] == '\r') l--;
    return (uint32_t)l;
}
__device__ __forceinline__ RecSpan rec_span(const uint8_t *text, const uint64_t *ls, uint64_t r)
{
    RecSpan x;
    const uint64_t l0 = ls[4 * r], l1 = ls[4 * r + 1], l2 = ls[4 * r + 2], l3 = ls[4 * r + 3], l4 = ls[4 * r + 4];
    x.h = l0; x.s = l1; x.q = l3;
    x.hl = line_len(text, l0, l1); x.sl = line_len(text, l1, l2); x.ql = line_len(text, l3, l4);
    return x;
}
__global__ __launch_bounds__(256) void out_lens_kernel(const uint8_t *text, const uint64_t *line_start, uint64_t n_rec, uint64_t rec_base,
                                                       const uint32_t *bits_a, const uint32_t *bits_b, int both, uint32_t *out_len)
{
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rec) return;
    const uint64_t g = rec_base + r;
    const uint32_t a = (bits_a[g >> 5] >> (g & 31)) & 1u, b = bits_b ? (bits_b[g >> 5] >> (g & 31)) & 1u : a;
    const uint32_t keep = both ? (a & b) : (a | b);
    uint32_t n = 0;
    if (keep) { const RecSpan x = rec_span(text, line_start, r); n = x.hl + x.sl + x.ql + 5; }      // header LF seq LF '+' LF qual LF
    out_len[r] = n;
}
// the same for a short list of records (sel: their indices, ascending): what survives a bait filter is a fraction of a per cent of
// the records, and kernels over the list cost nothing beside kernels over every record
__global__ __launch_bounds__(256) void sel_lens_kernel(const uint8_t *text, const uint64_t *line_start, const uint32_t *sel, uint64_t n_sel, uint32_t *out_len)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_sel) return;
    const RecSpan x = rec_span(text, line_start, sel[i]);
    out_len[i] = x.hl + x.sl + x.ql + 5;
}
__global__ __launch_bounds__(256) void sel_gather_kernel(const uint8_t *text, const uint64_t *line_start, const uint32_t *sel, uint64_t n_sel, const uint64_t *out_off, uint8_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);        // one wavefront per record
    if (i >= n_sel) return;
    const uint32_t lane = threadIdx.x & 63;
    const RecSpan x = rec_span(text, line_start, sel[i]);
    uint8_t *o = out + out_off[i];
    for (uint32_t k = lane; k < x.hl; k += 64) o[k] = text[x.h + k];
    o += x.hl;
    for (uint32_t k = lane; k < x.sl; k += 64) o[1 + k] = text[x.s + k];
    if (lane == 0) { o[0] = '\n'; o[1 + x.sl] = '\n'; o[2 + x.sl] = '+'; o[3 + x.sl] = '\n'; o[4 + x.sl + x.ql] = '\n'; }
    o += x.sl + 4;
    for (uint32_t k = lane; k < x.ql; k += 64) o[k] = text[x.q + k];
}
__global__ __launch_bounds__(256) void gather_kernel(const uint8_t *text, const uint64_t *line_start, uint64_t n_rec, const uint32_t *out_len,
                                                     const uint64_t *out_off, uint8_t *out)
{
    const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);        // one wavefront per record
    if (r >= n_rec || out_len[r] == 0) return;
    const uint32_t lane = threadIdx.x & 63;
    const RecSpan x = rec_span(text, line_start, r);
    uint8_t *o = out + out_off[r];
    for (uint32_t i = lane; i < x.hl; i += 64) o[i] = text[x.h + i];
    o += x.hl;
    for (uint32_t i = lane; i < x.sl; i += 64) o[1 + i] = text[x.s + i];
    if (lane == 0) { o[0] = '\n'; o[1 + x.sl] = '\n'; o[2 + x.sl] = '+'; o[3 + x.sl] = '\n'; o[4 + x.sl + x.ql] = '\n'; }
    o += x.sl + 4;
    for (uint32_t i = lane; i < x.ql; i += 64) o[i] = text[x.q + i];
}


// ------------------------------------------------------------------------------------------------------------ quality filter
// The reference's FASTQ quality filter (filter/filter_bin/src/main.rs:188-323) over text that is already on the device: what it
// asks of a record -- the cut [start, end) of sequence and quality string (:222-233, :291-299), the 'N's of the cut sequence
// (:236, :302), the quality bytes at or below the threshold (:239-243, :305-309), whether a line that it unwraps holds a byte that
// is not ASCII (:214-216, :287-289: such a line may not be UTF-8, the host looks) or is shorter than the cut's start (`drain` panics) --
// comes from ONE pass over the record's bytes, eight lanes a record, sixteen bytes a lane and step.

// 0x80 in byte k of the dword at relative offset p for which lo <= p + k < hi
__device__ __forceinline__ uint32_t win80(uint32_t p, uint32_t lo, uint32_t hi)
{
    if (p + 4 <= lo || p >= hi) return 0;
    uint32_t m = 0x80808080u;
    if (p < lo) m &= 0xFFFFFFFFu << (8 * (lo - p));
    if (p + 4 > hi) m &= 0xFFFFFFFFu >> (8 * (p + 4 - hi));
    return m;
}
constexpr uint32_t QS_GROUP = 8;          // lanes per record

__global__ __launch_bounds__(256) void qual_scan_kernel(const uint8_t *text, const uint64_t *line_start, uint64_t n_rec, uint64_t start, uint64_t cap,
                                                        uint32_t quality, uint64_t ns, uint32_t *bad, uint8_t *flags, uint32_t *cut_sl, uint32_t *cut_ql,
                                                        uint32_t *olen, uint32_t *first_flag)
{
    const uint64_t r = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / QS_GROUP;
    const uint32_t l = threadIdx.x % QS_GROUP;
    uint32_t nn = 0, nb = 0, hi = 0, fl = 0, hl = 0, csl = 0, cql = 0;
    if (r < n_rec) {
        const uint64_t l0 = line_start[4 * r], l1 = line_start[4 * r + 1], l2 = line_start[4 * r + 2], l3 = line_start[4 * r + 3], l4 = line_start[4 * r + 4];
        if (l4 - l0 >= 0xFFFFFF00ull) fl |= QF_LONG;
        else {
            hl = line_len(text, l0, l1);
            const uint32_t sl = line_len(text, l1, l2), ql = line_len(text, l3, l4);
            uint32_t cs = (uint32_t)(l1 - l0), cq = (uint32_t)(l3 - l0);            // where the cut strings begin, counted from the record's first byte
            if (start > sl || start > ql) fl |= QF_SHORT;
            else {
                cs += (uint32_t)start; cq += (uint32_t)start;
                csl = sl - (uint32_t)start; cql = ql - (uint32_t)start;
                if (cap < csl) csl = (uint32_t)cap;
                if (cap < cql) cql = (uint32_t)cap;
            }
            const uint32_t total = (uint32_t)(l4 - l0), plus_lo = (uint32_t)(l2 - l0), plus_hi = (uint32_t)(l3 - l0);
            const uint32_t q1 = (quality + 1) * 0x01010101u;
            const uint8_t *base = text + l0;
            for (uint32_t b = l * 16; b < total; b += QS_GROUP * 16) {
                const uint4 v = load16(base + b);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    const uint32_t p = b + 4 * j, x = w[j];
                    hi |= x & (win80(p, 0, plus_lo) | win80(p, plus_hi, total));            // (the '+' line is never unwrapped)
                    nn += __popc(eq_flags(x, 0x4E4E4E4Eu) & win80(p, cs, cs + csl));
                    const uint32_t d = (x | 0x80808080u) - q1;                              // per byte, no borrow: top bit clear <=> low seven bits <= quality
                    nb += __popc(~d & ~x & win80(p, cq, cq + cql));
                }
            }
        }
    }
#pragma unroll
    for (int o = QS_GROUP / 2; o > 0; o >>= 1) { nn += __shfl_xor(nn, o, QS_GROUP); nb += __shfl_xor(nb, o, QS_GROUP); hi |= __shfl_xor(hi, o, QS_GROUP); }
    if (r < n_rec && l == 0) {
        if (hi) fl |= QF_HIGH;
        if ((uint64_t)nn > ns) fl |= QF_NFAIL;
        bad[r] = nb; flags[r] = (uint8_t)fl; cut_sl[r] = csl; cut_ql[r] = cql; olen[r] = hl + csl + cql + 5;
        if (fl & (QF_HIGH | QF_SHORT | QF_LONG)) atomicMin(first_flag, (uint32_t)r);
    }
}

// SipHash-1-3, keys (0, 0), of the cut sequence followed by 0xff: `seq1.hash()` of the reference's de-duplication
// (main.rs:244-250, 325-329; std's DefaultHasher).  One lane per record, the message read as aligned dwords.
__device__ __forceinline__ void sip_round(uint64_t &v0, uint64_t &v1, uint64_t &v2, uint64_t &v3)
{
    auto rotl = [](uint64_t x, int b) { return (x << b) | (x >> (64 - b)); };
    v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32);
    v2 += v3; v3 = rotl(v3, 16); v3 ^= v2;
    v0 += v3; v3 = rotl(v3, 21); v3 ^= v0;
    v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32);
}
__global__ __launch_bounds__(256) void qual_hash_kernel(const uint8_t *text, const uint64_t *line_start, uint64_t n_rec, uint64_t start, const uint32_t *cut_sl,
                                                        uint64_t *hashes)
{
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rec) return;
    const uint32_t s_len = cut_sl[r];
    const uint8_t *p = text + line_start[4 * r + 1] + (s_len ? start : 0);           // (a record the cut panics on has no cut sequence)
    const uint32_t *w = reinterpret_cast<const uint32_t *>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3) * 8;
    auto alignbit = [](uint32_t hi, uint32_t lo, uint32_t s) -> uint32_t { return s ? (lo >> s) | (hi << (32 - s)) : lo; };
    uint64_t v0 = 0x736F6D6570736575ULL, v1 = 0x646F72616E646F6DULL, v2 = 0x6C7967656E657261ULL, v3 = 0x7465646279746573ULL;
    uint32_t d0 = w[0];
    uint64_t i = 0;
    auto block = [&](uint64_t at) -> uint64_t {             // eight message bytes from byte `at` (a multiple of 8)
        const uint32_t *q = w + (at >> 2);
        const uint32_t d1 = q[1], d2 = q[2];
        const uint64_t m = ((uint64_t)alignbit(d2, d1, sh) << 32) | alignbit(d1, d0, sh);
        d0 = d2;
        return m;
    };
    for (; i + 8 <= (uint64_t)s_len; i += 8) { const uint64_t m = block(i); v3 ^= m; sip_round(v0, v1, v2, v3); v0 ^= m; }
    const uint32_t t = (uint32_t)(s_len - i);               // the t < 8 bytes that are left, then 0xff (str's Hash); t == 7: the 0xff completes a block
    uint64_t tailv = block(i);
    tailv = t == 0 ? 0 : (tailv & (~0ULL >> (64 - 8 * t)));
    tailv |= 0xFFULL << (8 * t);
    if (t == 7) { v3 ^= tailv; sip_round(v0, v1, v2, v3); v0 ^= tailv; tailv = 0; }
    const uint64_t b = ((((uint64_t)s_len + 1) & 0xFF) << 56) | tailv;
    v3 ^= b; sip_round(v0, v1, v2, v3); v0 ^= b;
    v2 ^= 0xFF;
    sip_round(v0, v1, v2, v3); sip_round(v0, v1, v2, v3); sip_round(v0, v1, v2, v3);
    hashes[r] = v0 ^ v1 ^ v2 ^ v3;
}

// the tests that need no other record (main.rs:236-243, 302-309): alive[i] = the record reaches the de-duplication / is kept
__global__ __launch_bounds__(256) void qual_decide_kernel(uint64_t n, int pe, int trunc, float limit, const uint32_t *bad1, const uint8_t *fl1, const uint32_t *sl1,
                                                          const uint32_t *ql1, const uint32_t *bad2, const uint8_t *fl2, uint8_t *alive)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    bool drop = false;
    if (!trunc) {
        drop = (fl1[i] & QF_NFAIL) || (pe && (fl2[i] & QF_NFAIL));
        if (!drop) {
            // PE: both mates against a cutoff from seq1's length; SE: from the quality string's.  f32 product, `as usize` (saturating, NaN -> 0)
            const float cf = (float)(pe ? sl1[i] : ql1[i]) * limit;
            const uint64_t cutoff = !(cf > 0.0f) ? 0 : (cf >= 18446744073709551616.0f ? ~0ull : (uint64_t)cf);
            drop = (uint64_t)bad1[i] >= cutoff || (pe && (uint64_t)bad2[i] >= cutoff);
        }
    }
    alive[i] = drop ? 0 : 1;
}
// keep = alive and not a duplicate; out_len = output bytes of a kept record; *kept += the kept ones
__global__ __launch_bounds__(256) void qual_keep_kernel(uint64_t n, const uint8_t *alive, const uint8_t *dup, const uint32_t *olen, uint8_t *keep, uint32_t *out_len,
                                                        unsigned long long *kept)
{
    __shared__ uint32_t lds[4];
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t k = 0;
    if (i < n) {
        k = alive[i] && !(dup && dup[i]);
        if (keep) keep[i] = (uint8_t)k;
        out_len[i] = k ? olen[i] : 0;
    }
    const uint32_t s = block_sum(k, lds);
    if (threadIdx.x == 0 && s && kept) atomicAdd(kept, (unsigned long long)s);
}
// header LF cut-sequence LF '+' LF cut-quality LF of the kept records (main.rs:261-268, 317-321), one wavefront a record
__global__ __launch_bounds__(256) void qual_gather_kernel(const uint8_t *text, const uint64_t *line_start, uint64_t n_rec, uint64_t start, const uint32_t *cut_sl,
                                                          const uint32_t *cut_ql, const uint32_t *out_len, const uint64_t *out_off, uint8_t *out)
{
    const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rec || out_len[r] == 0) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t l0 = line_start[4 * r], l1 = line_start[4 * r + 1], l3 = line_start[4 * r + 3];
    const uint32_t hl = line_len(text, l0, l1), sl = cut_sl[r], ql = cut_ql[r];
    const uint8_t *h = text + l0, *s = text + l1 + start, *q = text + l3 + start;
    uint8_t *o = out + out_off[r];
    for (uint32_t i = lane; i < hl; i += 64) o[i] = h[i];
    o += hl;
    for (uint32_t i = lane; i < sl; i += 64) o[1 + i] = s[i];
    if (lane == 0) { o[0] = '\n'; o[1 + sl] = '\n'; o[2 + sl] = '+'; o[3 + sl] = '\n'; o[4 + sl + ql] = '\n'; }
    o += sl + 4;
    for (uint32_t i = lane; i < ql; i += 64) o[i] = q[i];
}

} // namespace

void ingest_preload() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&count_newlines_kernel)); (void)hipGetLastError(); }      // (see gz_preload, mf_gzdev.hip)

uint64_t pack_blocks(uint64_t total_bases, uint64_t base)
{
    if (!total_bases) return 0;
    const uint64_t nw = ((base + total_bases + 15) >> 4) - (base >> 4);
    return (nw + 255) / 256;
}

hipError_t launch_scan_u32(const uint32_t *in, uint64_t n, uint64_t *out, uint64_t *scratch, hipStream_t st)
{
    const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nb) hipLaunchKernelGGL(scan_reduce_kernel, dim3((uint32_t)nb), dim3(SCAN_BLOCK), 0, st, in, n, scratch);
    hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(SCAN_BLOCK), 0, st, scratch, nb);
    hipLaunchKernelGGL(scan_final_kernel, dim3((uint32_t)(nb ? nb : 1)), dim3(SCAN_BLOCK), 0, st, in, n, scratch, out, nb);
    return hipGetLastError();
}

hipError_t launch_count_newlines(const uint8_t *text, uint64_t n, uint32_t *tile_cnt, hipStream_t st)
{
    const uint64_t nt = (n + INGEST_TILE - 1) / INGEST_TILE;
    if (!nt) return hipSuccess;
    hipLaunchKernelGGL(count_newlines_kernel, dim3((uint32_t)nt), dim3(256), 0, st, text, n, tile_cnt);
    return hipGetLastError();
}

hipError_t launch_line_starts(const uint8_t *text, uint64_t n, const uint64_t *tile_base, uint64_t *line_start, hipStream_t st)
{
    const uint64_t nt = (n + INGEST_TILE - 1) / INGEST_TILE;
    if (!nt) return hipSuccess;
    hipLaunchKernelGGL(line_starts_kernel, dim3((uint32_t)nt), dim3(256), 0, st, text, n, tile_base, line_start);
    return hipGetLastError();
}

hipError_t launch_seq_lens(const uint8_t *text, const uint64_t *line_start, uint64_t n_rec, uint32_t *seq_len, uint32_t *minmax, hipStream_t st)
{
    if (!n_rec) return hipSuccess;
    hipLaunchKernelGGL(seq_lens_kernel, dim3((uint32_t)((n_rec + 255) / 256)), dim3(256), 0, st, text, line_start, n_rec, seq_len, minmax);
    return hipGetLastError();
}

hipError_t launch_pack(const uint8_t *text, const uint64_t *line_start, const uint64_t *offsets, uint32_t uniform_len, uint64_t n_rec,
                       uint64_t total_bases, uint64_t base, uint32_t *words, uint32_t *inv_cnt, const uint64_t *inv_base, uint64_t *npos, hipStream_t st)
{
    const uint64_t nb = pack_blocks(total_bases, base);
    if (!nb) return hipSuccess;
    if (!npos) hipLaunchKernelGGL(pack_kernel<0>, dim3((uint32_t)nb), dim3(256), 0, st, text, line_start, offsets, uniform_len, n_rec, total_bases, base, words, inv_cnt, inv_base, npos);
    else hipLaunchKernelGGL(pack_kernel<1>, dim3((uint32_t)nb), dim3(256), 0, st, text, line_start, offsets, uniform_len, n_rec, total_bases, base, words, inv_cnt, inv_base, npos);
    return hipGetLastError();
}

// A few bytes to a few megabytes from PINNED host memory to the device, read by a kernel over the link instead of queued on the copy engine:
// the engine carries the uploads, and whatever small copy a consumer or a link step asks for in the same direction waits behind all of
// them (a plain pair's twelve 256 MiB slabs: 40-50 ms for a 300-byte carry, profiles/r05/g_pe_plain_trace_before.txt).
__global__ __launch_bounds__(256) void bytes_from_host_kernel(uint8_t *dst, const uint8_t *src, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256, t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if ((((uintptr_t)dst | (uintptr_t)src | n) & 15) == 0) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src); uint4 *d4 = reinterpret_cast<uint4 *>(dst);
        for (uint64_t i = t; i < n / 16; i += stride) d4[i] = s4[i];
    } else
        for (uint64_t i = t; i < n; i += stride) dst[i] = src[i];
}
hipError_t launch_bytes_from_host(void *dst, const void *pinned_src, uint64_t n, hipStream_t st)
{
    if (!n) return hipSuccess;
    static const bool on_engine = getenv("MF_SMALL_H2D_ON_ENGINE") != nullptr;          // (A/B runs)
    if (on_engine) return hipMemcpyAsync(dst, pinned_src, n, hipMemcpyHostToDevice, st);
    const uint64_t items = ((((uintptr_t)dst | (uintptr_t)pinned_src | n) & 15) == 0) ? n / 16 : n;
    hipLaunchKernelGGL(bytes_from_host_kernel, dim3((uint32_t)std::min<uint64_t>(1024, (items + 255) / 256)), dim3(256), 0, st, (uint8_t *)dst, (const uint8_t *)pinned_src, n);
    return hipGetLastError();
}

// ... and the other way: a few bytes to a few hundred kilobytes written by a kernel into pinned (coherent) host memory; the host reads them
// after it has waited for the stream.  The quality filter's output goes down through the copy engine a gigabyte at a time.
static const bool g_small_d2h_on_engine = getenv("MF_SMALL_D2H_ON_ENGINE") != nullptr;          // (A/B runs)
hipError_t launch_bytes_to_host(void *pinned_dst, const void *src, uint64_t n, hipStream_t st)
{
    if (!n) return hipSuccess;
    if (g_small_d2h_on_engine) return hipMemcpyAsync(pinned_dst, src, n, hipMemcpyDeviceToHost, st);
    const uint64_t items = ((((uintptr_t)pinned_dst | (uintptr_t)src | n) & 15) == 0) ? n / 16 : n;
    hipLaunchKernelGGL(bytes_from_host_kernel, dim3((uint32_t)std::min<uint64_t>(1024, (items + 255) / 256)), dim3(256), 0, st, (uint8_t *)pinned_dst, (const uint8_t *)src, n);
    return hipGetLastError();
}

hipError_t launch_add_base(uint64_t *dst, const uint64_t *src, uint64_t n, uint64_t base, hipStream_t st)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(add_base_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, dst, src, n, base);
    return hipGetLastError();
}

hipError_t launch_store_bits(const uint32_t *batch_bits, uint64_t n_rec, uint32_t *file_bits, uint64_t rec_base, hipStream_t st)
{
    if (!n_rec) return hipSuccess;
    const uint64_t nw = ((rec_base + n_rec - 1) >> 5) - (rec_base >> 5) + 1;
    hipLaunchKernelGGL(store_bits_kernel, dim3((uint32_t)((nw + 255) / 256)), dim3(256), 0, st, batch_bits, n_rec, file_bits, rec_base);
    return hipGetLastError();
}

hipError_t launch_out_lens(const uint8_t *text, const uint64_t *line_start, uint64_t n_rec, uint64_t rec_base, const uint32_t *bits_a,
                           const uint32_t *bits_b, int both, uint32_t *out_len, hipStream_t st)
{
    if (!n_rec) return hipSuccess;
    hipLaunchKernelGGL(out_lens_kernel, dim3((uint32_t)((n_rec + 255) / 256)), dim3(256), 0, st, text, line_start, n_rec, rec_base, bits_a, bits_b, both, out_len);
    return hipGetLastError();
}

hipError_t launch_sel_lens(const uint8_t *text, const uint64_t *line_start, const uint32_t *sel, uint64_t n_sel, uint32_t *out_len, hipStream_t st)
{
    if (!n_sel) return hipSuccess;
    hipLaunchKernelGGL(sel_lens_kernel, dim3((uint32_t)((n_sel + 255) / 256)), dim3(256), 0, st, text, line_start, sel, n_sel, out_len);
    return hipGetLastError();
}

hipError_t launch_sel_gather(const uint8_t *text, const uint64_t *line_start, const uint32_t *sel, uint64_t n_sel, const uint64_t *out_off, uint8_t *out, hipStream_t st)
{
    if (!n_sel) return hipSuccess;
    hipLaunchKernelGGL(sel_gather_kernel, dim3((uint32_t)((n_sel + 3) / 4)), dim3(256), 0, st, text, line_start, sel, n_sel, out_off, out);
    return hipGetLastError();
}

hipError_t launch_gather(const uint8_t *text, const uint64_t *line_start, uint64_t n_rec, const uint32_t *out_len, const uint64_t *out_off,
                         uint8_t *out, hipStream_t st)
{
    if (!n_rec) return hipSuccess;
    hipLaunchKernelGGL(gather_kernel, dim3((uint32_t)((n_rec + 3) / 4)), dim3(256), 0, st, text, line_start, n_rec, out_len, out_off, out);
    return hipGetLastError();
}

hipError_t launch_qual_scan(const uint8_t *text, const uint64_t *line_start, uint64_t n_rec, uint64_t start, uint64_t cap, uint32_t quality, uint64_t ns,
                            uint32_t *bad, uint8_t *flags, uint32_t *cut_sl, uint32_t *cut_ql, uint32_t *olen, uint32_t *first_flag, hipStream_t st)
{
    if (!n_rec) return hipSuccess;
    hipLaunchKernelGGL(qual_scan_kernel, dim3((uint32_t)((n_rec * QS_GROUP + 255) / 256)), dim3(256), 0, st, text, line_start, n_rec, start, cap, quality, ns, bad, flags,
                       cut_sl, cut_ql, olen, first_flag);
    return hipGetLastError();
}
hipError_t launch_qual_hash(const uint8_t *text, const uint64_t *line_start, uint64_t n_rec, uint64_t start, const uint32_t *cut_sl, uint64_t *hashes, hipStream_t st)
{
    if (!n_rec) return hipSuccess;
    hipLaunchKernelGGL(qual_hash_kernel, dim3((uint32_t)((n_rec + 255) / 256)), dim3(256), 0, st, text, line_start, n_rec, start, cut_sl, hashes);
    return hipGetLastError();
}
hipError_t launch_qual_decide(uint64_t n, bool pe, bool trunc, float limit, const uint32_t *bad1, const uint8_t *fl1, const uint32_t *sl1, const uint32_t *ql1,
                              const uint32_t *bad2, const uint8_t *fl2, uint8_t *alive, hipStream_t st)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(qual_decide_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, n, pe ? 1 : 0, trunc ? 1 : 0, limit, bad1, fl1, sl1, ql1, bad2, fl2, alive);
    return hipGetLastError();
}
hipError_t launch_qual_keep(uint64_t n, const uint8_t *alive, const uint8_t *dup, const uint32_t *olen, uint8_t *keep, uint32_t *out_len, unsigned long long *kept,
                            hipStream_t st)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(qual_keep_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, n, alive, dup, olen, keep, out_len, kept);
    return hipGetLastError();
}
hipError_t launch_qual_gather(const uint8_t *text, const uint64_t *line_start, uint64_t n_rec, uint64_t start, const uint32_t *cut_sl, const uint32_t *cut_ql,
                              const uint32_t *out_len, const uint64_t *out_off, uint8_t *out, hipStream_t st)
{
    if (!n_rec) return hipSuccess;
    hipLaunchKernelGGL(qual_gather_kernel, dim3((uint32_t)((n_rec + 3) / 4)), dim3(256), 0, st, text, line_start, n_rec, start, cut_sl, cut_ql, out_len, out_off, out);
    return hipGetLastError();
}

} // namespace mf
