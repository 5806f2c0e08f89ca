// Micro-benchmark (not part of the product): what the part does with INDEPENDENT random small gathers -- the roof of any
// bait-sized second front of the s-mer screen ("bait-size axis").
//   A  pure gathers: every lane issues G independent W-byte loads (W = 4 / 16) at hashed addresses inside a table of F bytes,
//      DEPTH of them in flight; persistent 1024-thread workgroups, one per CU.  -> lookups/s by footprint and width.
//   B  a 16-byte non-temporal streaming read of 1.25 GB (the screen kernel's loop shape: two loads a lane and chunk, the next
//      chunk in flight) with, per lane and chunk, `cnt` gathers issued by a fraction `p` of the lanes and consumed one chunk
//      later (software pipelined, so the wave never waits for a gather it has just issued).  -> ms a pass, lookups/s.
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_roof.hip -o tools/gather_roof
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t xs(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }

template <int W, int DEPTH>
__global__ void __launch_bounds__(1024) gather_kernel(const uint32_t *__restrict__ tab, uint32_t lg_items, uint32_t rounds, uint32_t *out)
{
    uint32_t x = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    const uint32_t sh = 32 - lg_items;
    for (uint32_t r = 0; r < rounds; r++) {
        uint32_t v[DEPTH][W / 4];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            x = xs(x);
            const uint32_t idx = x >> sh;
            if (W == 4) v[d][0] = tab[idx];
            else { const u32x4 t = reinterpret_cast<const u32x4 *>(tab)[idx]; v[d][0] = t.x; v[d][1 % (W / 4)] = t.y; v[d][2 % (W / 4)] = t.z; v[d][3 % (W / 4)] = t.w; }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
#pragma unroll
            for (int j = 0; j < W / 4; j++) acc ^= v[d][j];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// B: stream + pipelined gathers.  U = 2 uint4 per lane and chunk, as the screen kernel.
template <int CNT>
__global__ void __launch_bounds__(1024) stream_gather_kernel(const u32x4 *__restrict__ w4, uint64_t n_vec, const uint32_t *__restrict__ tab, uint32_t lg_items,
                                                             uint32_t p_thresh, uint32_t *out)
{
    const uint64_t chunk = 2048, n_chunks = n_vec / chunk, cstep = gridDim.x;
    const uint32_t sh = 32 - lg_items;
    uint32_t acc = 0;
    uint32_t g[CNT]; bool pend = false;
#pragma unroll
    for (int i = 0; i < CNT; i++) g[i] = 0;
    uint64_t c = blockIdx.x;
    u32x4 a0, a1, b0, b1;
    auto load = [&](uint64_t cc, u32x4 &d0, u32x4 &d1) {
        d0 = __builtin_nontemporal_load(&w4[cc * chunk + threadIdx.x]);
        d1 = __builtin_nontemporal_load(&w4[cc * chunk + 1024 + threadIdx.x]);
    };
    auto work = [&](const u32x4 &d0, const u32x4 &d1) {
        // consume the gathers issued one chunk ago, then issue this chunk's
        if (pend) {
#pragma unroll
            for (int i = 0; i < CNT; i++) acc ^= g[i];
        }
        const uint32_t h = xs(d0.x ^ d1.y ^ (d0.z * 0x9E3779B1u) ^ d1.w ^ threadIdx.x);
        pend = h <= p_thresh;
        acc += d0.y ^ d0.w ^ d1.x ^ d1.z;
        if (pend) {
            uint32_t x = h;
#pragma unroll
            for (int i = 0; i < CNT; i++) { x = xs(x + 0x9E3779B9u); g[i] = tab[x >> sh]; }
        }
    };
    if (c < n_chunks) {
        load(c, a0, a1);
        for (;;) {
            if (c + cstep >= n_chunks) { work(a0, a1); break; }
            load(c + cstep, b0, b1);
            work(a0, a1);
            c += cstep;
            if (c + cstep >= n_chunks) { work(b0, b1); break; }
            load(c + cstep, a0, a1);
            work(b0, b1);
            c += cstep;
        }
    }
    if (pend) {
#pragma unroll
        for (int i = 0; i < CNT; i++) acc ^= g[i];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char **argv)
{
    int n_cu = 256;
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0)); n_cu = prop.multiProcessorCount;
    const size_t TAB = (size_t)1 << 30;
    uint32_t *tab, *out; CHK(hipMalloc(&tab, TAB)); CHK(hipMemset(tab, 0x5A, TAB)); CHK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto time_ms = [&](auto &&launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            CHK(hipEventRecord(e0)); launch(); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
        }
        return best;
    };
    printf("# A: pure independent gathers, %d workgroups x 1024 threads, 8 in flight per lane\n", n_cu);
    printf("# footprint  width  lookups      ms     G lookups/s   GB/s(of W)\n");
    for (int lgF = 16; lgF <= 30; lgF += 2)
        for (int W : {4, 16}) {
            const uint32_t lg_items = (uint32_t)lgF - (W == 4 ? 2 : 4);
            const uint32_t rounds = 64;
            const double n = (double)n_cu * 1024 * rounds * 8;
            float ms;
            if (W == 4) ms = time_ms([&] { hipLaunchKernelGGL((gather_kernel<4, 8>), dim3(n_cu), dim3(1024), 0, 0, tab, lg_items, rounds, out); });
            else ms = time_ms([&] { hipLaunchKernelGGL((gather_kernel<16, 8>), dim3(n_cu), dim3(1024), 0, 0, (const uint32_t *)tab, lg_items, rounds, out); });
            printf("%8.2f MiB  %3d  %10.0f  %8.3f  %8.1f  %8.1f\n", (double)((size_t)1 << lgF) / 1048576.0, W, n, ms, n / ms / 1e6, n * W / ms / 1e6);
        }
    // B
    const uint64_t n_vec = (uint64_t)1250000000 / 16 / 2048 * 2048;
    u32x4 *stream; CHK(hipMalloc(&stream, n_vec * 16));
    {   // pseudo-random content (the lanes that gather are chosen by a hash of the data)
        std::vector<uint32_t> h(n_vec * 4);
        uint32_t x = 2463534242u; for (auto &v : h) { x = x * 1664525u + 1013904223u; v = x; }
        CHK(hipMemcpy(stream, h.data(), n_vec * 16, hipMemcpyHostToDevice));
    }
    const int grid = n_cu * 7 / 8;
    printf("# B: 1.25 GB 16-byte nt stream (%d persistent workgroups) + gathers: a fraction p of the lanes issues cnt 4-byte gathers per chunk (32 B of stream), consumed a chunk later\n", grid);
    printf("# footprint   p      cnt   ms/pass   stream TB/s   G lookups/s\n");
    for (int lgF : {20, 22, 25, 27, 29})
        for (double p : {0.0, 1.0 / 64, 1.0 / 16, 0.25, 1.0})
            for (int cnt : {1, 2, 8}) {
                if (p == 0.0 && (cnt != 1 || lgF != 20)) continue;
                const uint32_t pt = p >= 1.0 ? 0xFFFFFFFFu : (uint32_t)(p * 4294967296.0);
                const uint32_t lg_items = (uint32_t)lgF - 2;
                float ms;
                if (p == 0.0) ms = time_ms([&] { hipLaunchKernelGGL((stream_gather_kernel<1>), dim3(grid), dim3(1024), 0, 0, stream, n_vec, tab, lg_items, 0u, out); });
                else if (cnt == 1) ms = time_ms([&] { hipLaunchKernelGGL((stream_gather_kernel<1>), dim3(grid), dim3(1024), 0, 0, stream, n_vec, tab, lg_items, pt, out); });
                else if (cnt == 2) ms = time_ms([&] { hipLaunchKernelGGL((stream_gather_kernel<2>), dim3(grid), dim3(1024), 0, 0, stream, n_vec, tab, lg_items, pt, out); });
                else ms = time_ms([&] { hipLaunchKernelGGL((stream_gather_kernel<8>), dim3(grid), dim3(1024), 0, 0, stream, n_vec, tab, lg_items, pt, out); });
                const double look = (double)n_vec / 2 * p * cnt;
                printf("%8.2f MiB  %6.4f  %2d  %8.4f  %8.2f  %10.1f\n", (double)((size_t)1 << lgF) / 1048576.0, p, cnt, ms, n_vec * 16.0 / ms / 1e9, look / ms / 1e6);
            }
    return 0;
}
