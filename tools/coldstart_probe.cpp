// What a process pays before its first useful kernel on this box: HIP start-up, code-object load, allocation, pinned memory, streams.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -x hip tools/coldstart_probe.cpp -ldl -o tools/coldstart_probe     (make -C mitoflex_amd/csrc tools)
//   tools/coldstart_probe [path/to/libmitofilter_hip.so]
// Every line is "seconds since main() | what | seconds it took".  The reference's boundary on this path is a process per call
// (/root/reference/utility/helper.py:78-86), so these are costs of EVERY call through the CLIs.
#include <hip/hip_runtime.h>
#include <chrono>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <thread>
#include <vector>

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double T0;
#define STEP(what, ...) do { const double a_ = now_s(); __VA_ARGS__; const double b_ = now_s(); printf("%8.4f | %-58s | %8.4f\n", b_ - T0, what, b_ - a_); fflush(stdout); } while (0)

__global__ void touch_kernel(unsigned *p) { if (p) p[threadIdx.x] = threadIdx.x; }

int main(int argc, char **argv)
{
    T0 = now_s();
    int n = 0;
    STEP("hipInit(0)", (void)hipInit(0));
    STEP("hipGetDeviceCount", (void)hipGetDeviceCount(&n));
    STEP("hipSetDevice(0) + hipFree(0) (context)", { (void)hipSetDevice(0); (void)hipFree(nullptr); });
    unsigned *d = nullptr;
    STEP("hipMalloc 4 KiB (first allocation)", (void)hipMalloc(&d, 4096));
    STEP("first kernel of this binary (code object load + launch + sync)", { hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, 0, d); (void)hipDeviceSynchronize(); });
    STEP("second kernel (launch + sync)", { hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, 0, d); (void)hipDeviceSynchronize(); });
    for (size_t mb : {64, 256, 1024, 4096, 16384}) {
        void *p = nullptr; char w[96];
        snprintf(w, sizeof w, "hipMalloc %zu MiB", mb);
        STEP(w, (void)hipMalloc(&p, mb << 20));
        snprintf(w, sizeof w, "hipMemsetAsync over it + sync (first touch)");
        STEP(w, { (void)hipMemsetAsync(p, 0, mb << 20, 0); (void)hipDeviceSynchronize(); });
        snprintf(w, sizeof w, "hipFree %zu MiB", mb);
        STEP(w, (void)hipFree(p));
    }
    {   // many small allocations against one arena
        std::vector<void *> v(200);
        STEP("200 x hipMalloc 8 MiB", for (auto &p : v) (void)hipMalloc(&p, 8 << 20));
        STEP("200 x hipFree", for (auto &p : v) (void)hipFree(p));
    }
    for (size_t mb : {1, 32, 256}) {
        void *p = nullptr; char w[96];
        snprintf(w, sizeof w, "hipHostMalloc %zu MiB (portable)", mb);
        STEP(w, (void)hipHostMalloc(&p, mb << 20, hipHostMallocPortable));
        snprintf(w, sizeof w, "hipHostFree %zu MiB", mb);
        STEP(w, (void)hipHostFree(p));
    }
    {
        void *p = nullptr; (void)posix_memalign(&p, 4096, (size_t)64 << 20); memset(p, 1, (size_t)64 << 20);
        STEP("hipHostRegister 64 MiB of touched malloc memory", (void)hipHostRegister(p, (size_t)64 << 20, hipHostRegisterPortable));
        STEP("hipHostUnregister", (void)hipHostUnregister(p));
        free(p);
    }
    if (const char *path = getenv("PROBE_MMAP")) {          // can the copy engine read a file's pages where the page cache holds them?  (register the mapping, copy from it)
        int fd = open(path, O_RDONLY); struct stat sb; fstat(fd, &sb);
        const size_t n = (size_t)sb.st_size & ~(size_t)((2 << 20) - 1);
        void *dev = nullptr; (void)hipMalloc(&dev, n);
        hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        for (int variant = 0; variant < 4; variant++) {
            const int prot = variant & 1 ? PROT_READ | PROT_WRITE : PROT_READ;
            const unsigned flags = variant & 2 ? hipHostRegisterPortable : (hipHostRegisterPortable | hipHostRegisterReadOnly);
            void *m = mmap(nullptr, n, prot, MAP_PRIVATE | MAP_POPULATE, fd, 0);
            if (m == MAP_FAILED) { printf("mmap failed\n"); continue; }
            char w[160];
            snprintf(w, sizeof w, "hipHostRegister of %zu MiB of a %s private file mapping, %s", n >> 20, prot & PROT_WRITE ? "read-write" : "read-only", variant & 2 ? "default flags" : "ReadOnly flag");
            hipError_t e = hipSuccess;
            STEP(w, e = hipHostRegister(m, n, flags));
            if (e != hipSuccess) { printf("         | -> %s\n", hipGetErrorString(e)); (void)hipGetLastError(); munmap(m, n); continue; }
            STEP("  hipMemcpyAsync of all of it from the registered mapping + sync", { (void)hipMemcpyAsync(dev, m, n, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); });
            STEP("  the same again", { (void)hipMemcpyAsync(dev, m, n, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); });
            STEP("  hipHostUnregister", (void)hipHostUnregister(m));
            munmap(m, n);
        }
        for (int variant = 0; variant < 3; variant++) {          // the way the product does it: no MAP_POPULATE, a window of 512 MiB at a time; variant 0: register cold, 1: madvise POPULATE_READ (8 threads) first, 2: the same, populate one window ahead in the background
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            const size_t WIN = (size_t)512 << 20; const size_t nw = (n + WIN - 1) / WIN;
            double t_pop = 0, t_reg = 0, t_copy = 0, t_unreg = 0; const double a0 = now_s();
            auto populate = [&](size_t w) { const size_t a = w * WIN, b = a + WIN < n ? a + WIN : n; std::vector<std::thread> th; const size_t part = (b - a + 7) / 8 + 4095 & ~(size_t)4095;
                for (size_t q = a; q < b; q += part) th.emplace_back([=] { madvise((char *)m + q, (q + part < b ? q + part : b) - q, 22); }); for (auto &t : th) t.join(); };
            std::thread ahead;
            for (size_t w = 0; w < nw; w++) {
                const size_t a = w * WIN, len = a + WIN < n ? WIN : n - a;
                double t = now_s();
                if (variant == 1) populate(w);
                if (variant == 2) { if (ahead.joinable()) ahead.join(); else populate(w); if (w + 1 < nw) ahead = std::thread([&, w] { populate(w + 1); }); }
                t_pop += now_s() - t; t = now_s();
                (void)hipHostRegister((char *)m + a, len, hipHostRegisterPortable | hipHostRegisterReadOnly);
                t_reg += now_s() - t; t = now_s();
                (void)hipMemcpyAsync((char *)dev + a, (char *)m + a, len, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st);
                t_copy += now_s() - t; t = now_s();
                (void)hipHostUnregister((char *)m + a);
                t_unreg += now_s() - t;
            }
            if (ahead.joinable()) ahead.join();
            printf("%8.4f | windows of 512 MiB, %s: populate %.4f register %.4f copy+sync %.4f unregister %.4f | %8.4f\n", now_s() - T0,
                   variant == 0 ? "registered cold" : variant == 1 ? "populated first (8 threads)" : "populated one window ahead", t_pop, t_reg, t_copy, t_unreg, now_s() - a0);
            munmap(m, n);
        }
        {   // the reference point: pread into pinned staging (what the uploader does), 8 threads, and the copy
            void *pin = nullptr; (void)hipHostMalloc(&pin, (size_t)32 << 20, hipHostMallocPortable);
            const double a = now_s(); size_t done = 0;
            while (done < n) {
                const size_t len = n - done < ((size_t)32 << 20) ? n - done : (size_t)32 << 20;
                std::vector<std::thread> th;
                for (int t = 0; t < 8; t++) th.emplace_back([&, t] { size_t x = len * t / 8, y = len * (t + 1) / 8; while (x < y) { ssize_t g = pread(fd, (char *)pin + x, y - x, (off_t)(done + x)); if (g <= 0) break; x += (size_t)g; } });
                for (auto &t : th) t.join();
                (void)hipMemcpyAsync((char *)dev + done, pin, len, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st);
                done += len;
            }
            printf("%8.4f | %-58s | %8.4f\n", now_s() - T0, "pread (8 threads) into one pinned buffer + copy, serial", now_s() - a);
        }
        printf("%8.4f | done (file of %zu MiB)\n", now_s() - T0, n >> 20);
        return 0;
    }
    if (getenv("PROBE_STREAMS")) {          // how the cost of a stream depends on how many there are already, and whether making one stalls another thread's launches
        hipStream_t q[16] = {};
        for (int i = 0; i < 12; i++) { char w[64]; snprintf(w, sizeof w, "plain non-blocking stream #%d", i + 1); STEP(w, (void)hipStreamCreateWithFlags(&q[i], hipStreamNonBlocking)); }
        STEP("12 kernels, one per stream, + device sync", { for (int i = 0; i < 12; i++) hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, q[i], d); (void)hipDeviceSynchronize(); });
        int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        STEP("stream with priority (high)", (void)hipStreamCreateWithPriority(&q[12], hipStreamNonBlocking, hi));
        STEP("hipStreamDestroy of one", (void)hipStreamDestroy(q[11]));
        STEP("plain stream again", (void)hipStreamCreateWithFlags(&q[11], hipStreamNonBlocking));
        std::atomic<bool> stop{false}; double worst = 0, sum = 0; int n_it = 0;
        std::thread other([&] {
            (void)hipSetDevice(0);
            while (!stop) { const double a = now_s(); hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, q[0], d); (void)hipStreamSynchronize(q[0]); const double dt = now_s() - a; worst = dt > worst ? dt : worst; sum += dt; n_it++; }
        });
        hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
        const int n_cu = prop.multiProcessorCount, words = (n_cu + 31) / 32;
        std::vector<uint32_t> mask((size_t)words, 0);
        for (int b = 0; b < n_cu - 32; b++) mask[b / 32] |= 1u << (b % 32);
        hipStream_t m[6];
        STEP("6 CU-masked streams while another thread launches + syncs in a loop", for (auto &x : m) (void)hipExtStreamCreateWithCUMask(&x, (uint32_t)words, mask.data()));
        stop = true; other.join();
        printf("         | the other thread: %d launch+sync rounds, mean %.6f s, worst %.6f s\n", n_it, n_it ? sum / n_it : 0.0, worst);
        {   // ... and copies?  a thread that copies 32 MiB pieces from pinned memory to the device on its own stream, while this one makes six more masked streams
            void *pin = nullptr, *dv = nullptr; (void)hipHostMalloc(&pin, (size_t)32 << 20, hipHostMallocPortable); (void)hipMalloc(&dv, (size_t)32 << 20);
            std::atomic<bool> stop2{false}; double worst2 = 0, sum2 = 0; int n2 = 0; double quiet = 0; int nq = 0;
            for (int i = 0; i < 20; i++) { const double a = now_s(); (void)hipMemcpyAsync(dv, pin, (size_t)32 << 20, hipMemcpyHostToDevice, q[1]); (void)hipStreamSynchronize(q[1]); quiet += now_s() - a; nq++; }
            std::thread copier([&] {
                (void)hipSetDevice(0);
                while (!stop2) { const double a = now_s(); (void)hipMemcpyAsync(dv, pin, (size_t)32 << 20, hipMemcpyHostToDevice, q[1]); (void)hipStreamSynchronize(q[1]); const double dt = now_s() - a; worst2 = dt > worst2 ? dt : worst2; sum2 += dt; n2++; }
            });
            hipStream_t m2[6];
            STEP("6 more CU-masked streams while another thread copies 32 MiB pieces H2D + syncs in a loop", for (auto &x : m2) (void)hipExtStreamCreateWithCUMask(&x, (uint32_t)words, mask.data()));
            stop2 = true; copier.join();
            printf("         | the copying thread: alone %.6f s a piece (%.1f GB/s); beside the stream maker %d pieces, mean %.6f s, worst %.6f s\n", quiet / nq, 0.0335544 / (quiet / nq), n2, n2 ? sum2 / n2 : 0.0, worst2);
        }
        printf("%8.4f | done\n", now_s() - T0);
        return 0;
    }
    hipStream_t s[4] = {};
    STEP("hipStreamCreateWithFlags (non-blocking) #1", (void)hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
    STEP("hipStreamCreateWithFlags (non-blocking) #2", (void)hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    STEP("kernel on the new stream #1 (queue creation is lazy)", { hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, s[0], d); (void)hipStreamSynchronize(s[0]); });
    STEP("kernel on the new stream #2", { hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, s[1], d); (void)hipStreamSynchronize(s[1]); });
    {
        hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
        const int n_cu = prop.multiProcessorCount, words = (n_cu + 31) / 32;
        std::vector<uint32_t> mask((size_t)words, 0);
        for (int b = 0; b < n_cu - 32; b++) mask[b / 32] |= 1u << (b % 32);
        STEP("hipExtStreamCreateWithCUMask #1", (void)hipExtStreamCreateWithCUMask(&s[2], (uint32_t)words, mask.data()));
        STEP("hipExtStreamCreateWithCUMask #2", (void)hipExtStreamCreateWithCUMask(&s[3], (uint32_t)words, mask.data()));
        STEP("kernel on CU-masked stream #1", { hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, s[2], d); (void)hipStreamSynchronize(s[2]); });
    }
    hipEvent_t e;
    STEP("hipEventCreate", (void)hipEventCreate(&e));
    if (argc > 1) {
        void *h = nullptr;
        STEP("dlopen libmitofilter_hip.so (fat binary registration)", h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL));
        if (!h) { printf("dlopen failed: %s\n", dlerror()); return 1; }
        auto dc = (int (*)())dlsym(h, "mf_device_count");
        STEP("mf_device_count()", if (dc) (void)dc());
    }
    printf("%8.4f | done\n", now_s() - T0);
    return 0;
}
