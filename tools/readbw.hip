// Micro-benchmark (not part of the product): best achievable HBM READ bandwidth on this MI355X for
// a few streaming shapes, to know the real ceiling under the screen kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int U, bool NT, bool CONTIG>
__global__ void rd(const u32x4* __restrict__ p, uint64_t n_vec, uint32_t* out)
{
    const uint64_t chunk = (uint64_t)blockDim.x * U, n_chunks = n_vec / chunk;
    uint32_t acc = 0;
    uint64_t c0, c1, step;
    if (CONTIG) { const uint64_t per = (n_chunks + gridDim.x - 1) / gridDim.x; c0 = blockIdx.x * per; c1 = c0 + per < n_chunks ? c0 + per : n_chunks; step = 1; }
    else { c0 = blockIdx.x; c1 = n_chunks; step = gridDim.x; }
    u32x4 cur[U];
    if (c0 < c1) for (int u = 0; u < U; u++) { const u32x4* a = &p[c0 * chunk + (uint64_t)u * blockDim.x + threadIdx.x]; cur[u] = NT ? __builtin_nontemporal_load(a) : *a; }
    for (uint64_t c = c0; c < c1; c += step) {
        u32x4 nxt[U];
        if (c + step < c1) for (int u = 0; u < U; u++) { const u32x4* a = &p[(c + step) * chunk + (uint64_t)u * blockDim.x + threadIdx.x]; nxt[u] = NT ? __builtin_nontemporal_load(a) : *a; }
        for (int u = 0; u < U; u++) acc += cur[u].x ^ cur[u].y ^ cur[u].z ^ cur[u].w;
        for (int u = 0; u < U; u++) cur[u] = nxt[u];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// the screen kernel's own shape: the grid, block, loads per lane and LDS footprint it runs with (nothing else of it).
// LANE_CONTIG: a lane's U loads are neighbours in memory (32 contiguous bytes per lane for U = 2) instead of blockDim * 16 bytes apart
template <int U, bool LANE_CONTIG>
__global__ void __launch_bounds__(1024) rd_screen_shape(const u32x4* __restrict__ p, uint64_t n_vec, uint32_t* out)
{
    extern __shared__ uint4 lds[];
    const uint64_t chunk = (uint64_t)blockDim.x * U, n_chunks = n_vec / chunk, step = gridDim.x;
    if (threadIdx.x == 0) lds[0] = make_uint4(0, 0, 0, 0);
    uint32_t acc = 0;
    u32x4 cur[U];
    uint64_t c = blockIdx.x;
    auto at = [&](uint64_t cc, int u) { return LANE_CONTIG ? cc * chunk + (uint64_t)threadIdx.x * U + u : cc * chunk + (uint64_t)u * blockDim.x + threadIdx.x; };
    if (c < n_chunks) for (int u = 0; u < U; u++) cur[u] = __builtin_nontemporal_load(&p[at(c, u)]);
    for (; c < n_chunks; c += step) {
        u32x4 nxt[U];
        if (c + step < n_chunks) for (int u = 0; u < U; u++) nxt[u] = __builtin_nontemporal_load(&p[at(c + step, u)]);
        for (int u = 0; u < U; u++) acc += cur[u].x ^ cur[u].y ^ cur[u].z ^ cur[u].w;
        for (int u = 0; u < U; u++) cur[u] = nxt[u];
    }
    if (acc == 0x12345678u) out[0] = acc + lds[0].x;
}
template <int U, bool NT, bool CONTIG>
void run(const char* name, const u32x4* buf, uint64_t n_vec, uint32_t* out, int grid, int block)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int r = 0; r < 6; r++) {
        hipEventRecord(a); hipLaunchKernelGGL((rd<U, NT, CONTIG>), dim3(grid), dim3(block), 0, 0, buf, n_vec, out); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    printf("%-44s grid %5d block %4d : %.4f ms  %.0f GB/s\n", name, grid, block, best, n_vec * 16.0 / best / 1e6);
}
int main()
{
    const uint64_t n_vec = 78125056 / 4096 * 4096;   // ~1.25 GB
    u32x4* buf; uint32_t* out; hipMalloc(&buf, n_vec * 16); hipMemset(buf, 1, n_vec * 16); hipMalloc(&out, 64);
    {   // U = 2, 1024 threads, 128 KiB of LDS (one workgroup per CU), 224 workgroups: what screen_kernel<1, 2, false> launches with
        auto shape = [&](auto kern, const char* name) {
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 16);
            for (int grid : {224, 256}) {
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                float best = 1e9;
                for (int r = 0; r < 8; r++) {
                    hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), 131072 + 16, 0, buf, n_vec, out); hipEventRecord(b); hipEventSynchronize(b);
                    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
                }
                printf("%-44s grid %5d block %4d : %.4f ms  %.0f GB/s\n", name, grid, 1024, best, n_vec * 16.0 / best / 1e6);
            }
        };
        shape(rd_screen_shape<2, false>, "screen shape: U2 nt, 128 KiB LDS");
        shape(rd_screen_shape<2, true>, "same, 32 contiguous bytes per lane");
    }
    if (getenv("READBW_SHAPE_ONLY")) return 0;
    run<4, true, false>("U4 nt interleaved (screen kernel shape)", buf, n_vec, out, 256, 1024);
    run<4, false, false>("U4 plain interleaved", buf, n_vec, out, 256, 1024);
    run<4, true, true>("U4 nt contiguous per block", buf, n_vec, out, 256, 1024);
    run<8, true, false>("U8 nt interleaved", buf, n_vec, out, 256, 1024);
    run<2, true, false>("U2 nt interleaved", buf, n_vec, out, 256, 1024);
    run<4, true, false>("U4 nt interleaved 2 WG/CU", buf, n_vec, out, 512, 1024);
    run<4, true, false>("U4 nt interleaved 512-thr", buf, n_vec, out, 512, 512);
    run<4, true, false>("U4 nt interleaved 256-thr x8/CU", buf, n_vec, out, 2048, 256);
    run<8, true, false>("U8 nt 256-thr x8/CU", buf, n_vec, out, 2048, 256);
    run<4, false, false>("U4 plain 256-thr x8/CU", buf, n_vec, out, 2048, 256);
    run<1, true, false>("U1 nt 256-thr x16/CU", buf, n_vec, out, 4096, 256);
    return 0;
}
