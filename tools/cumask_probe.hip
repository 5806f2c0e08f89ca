// probe: the CU mask of an ordinary stream, and which CUs / XCCs a masked stream's workgroups land on
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ void where(uint32_t *out)
{
    uint32_t hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hwid; out[2 * blockIdx.x + 1] = xcc; }
    // stay resident for a while so that workgroups spread over the CUs
    const long long t0 = clock64(); while (clock64() - t0 < 2000000) ;
}
static void run(const char *name, hipStream_t st, int grid)
{
    uint32_t *d; hipMalloc(&d, grid * 8); hipMemset(d, 0xFF, grid * 8);
    hipLaunchKernelGGL(where, dim3(grid), dim3(1024), 65536 * 2, st, d);
    hipStreamSynchronize(st);
    std::vector<uint32_t> h(grid * 2); hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
    int per_xcc[16] = {0}; std::vector<int> seen(16 * 4 * 32, 0); int distinct = 0;
    for (int i = 0; i < grid; i++) {
        const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 15;
        const uint32_t cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_xcc[xcc]++;
        const int key = (xcc * 8 + se) * 32 + sh * 16 + cu;
        if (key < (int)seen.size() && !seen[key]++) distinct++;
    }
    printf("%-28s grid %3d distinct CUs %3d  per XCC:", name, grid, distinct);
    for (int x = 0; x < 8; x++) printf(" %d", per_xcc[x]);
    printf("\n");
    hipFree(d);
}
int main()
{
    hipFuncSetAttribute((const void *)where, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipStream_t s0; hipStreamCreate(&s0);
    uint32_t m[16] = {0};
    hipError_t e = hipExtStreamGetCUMask(s0, 16, m);
    printf("default stream mask (%s):", hipGetErrorString(e)); for (int i = 0; i < 16; i++) printf(" %08x", m[i]); printf("\n");
    run("unmasked", s0, 256);
    for (int variant = 0; variant < 4; variant++) {
        uint32_t ms[8] = {0};
        const char *nm[4] = {"first 224 bits", "28 of every 32 bits", "bits 0..127", "even bits"};
        for (int i = 0; i < 256; i++) {
            bool on = variant == 0 ? i < 224 : variant == 1 ? (i % 32) < 28 : variant == 2 ? i < 128 : (i % 2 == 0);
            if (on) ms[i / 32] |= 1u << (i % 32);
        }
        hipStream_t s; e = hipExtStreamCreateWithCUMask(&s, 8, ms);
        if (e != hipSuccess) { printf("%s: %s\n", nm[variant], hipGetErrorString(e)); continue; }
        uint32_t g[16] = {0}; hipExtStreamGetCUMask(s, 16, g);
        printf("%s -> stream reports:", nm[variant]); for (int i = 0; i < 10; i++) printf(" %08x", g[i]); printf("\n");
        run(nm[variant], s, 256);
    }
    return 0;
}
