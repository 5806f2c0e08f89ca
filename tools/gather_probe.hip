// Micro-benchmark (not part of the product): cost of N dependent-free random 4-byte gathers
// over a footprint of F bytes on MI355X.  Used to price the exact kernel's random reads.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
__global__ void gather(const uint32_t* __restrict__ buf, uint64_t n_words, uint32_t n, uint32_t* out, uint32_t seed, int chain)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t x = (i + 1) * 0x9E3779B97F4A7C15ULL + seed; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
    uint32_t acc = 0;
    for (int c = 0; c < chain; c++) {
        uint64_t idx = x % n_words;
        uint32_t v = buf[idx];
        acc += v;
        x = x * 6364136223846793005ULL + v + 1;   // next address depends on the loaded value
    }
    out[i] = acc;
}
int main()
{
    const uint64_t F = 1250000000ULL; uint32_t *buf, *out;
    hipMalloc(&buf, F); hipMemset(buf, 1, F); hipMalloc(&out, 1 << 24);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (uint64_t foot : {F / 4, (uint64_t)(64 << 20) / 4, (uint64_t)(2 << 20) / 4})
        for (uint32_t n : {10000u, 170000u, 1000000u})
            for (int chain : {1, 3}) {
                float best = 1e9;
                for (int rep = 0; rep < 5; rep++) {
                    hipEventRecord(a);
                    hipLaunchKernelGGL(gather, dim3((n + 255) / 256), dim3(256), 0, 0, buf, foot, n, out, rep * 77 + 1, chain);
                    hipEventRecord(b); hipEventSynchronize(b);
                    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
                }
                printf("footprint %8.1f MB  n=%7u chain=%d : %.1f us\n", foot * 4 / 1e6, n, chain, best * 1e3);
            }
    return 0;
}
