// TEST INFRASTRUCTURE: the CPU stand-in for the HIP runtime declared in hip/hip_runtime.h next to this file.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <thread>

struct StubStream {
    std::mutex mu; std::condition_variable cv, idle;
    std::deque<std::function<void()>> q; bool busy = false, stop = false;
    std::thread th;
    StubStream() { th = std::thread([this] { run(); }); }
    ~StubStream() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); th.join(); }
    void run()
    {
        for (;;) {
            std::function<void()> f;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || !q.empty(); }); if (q.empty()) return; f = std::move(q.front()); q.pop_front(); busy = true; }
            f();
            { std::lock_guard<std::mutex> lk(mu); busy = false; }
            idle.notify_all();
        }
    }
    void push(std::function<void()> f) { { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(f)); } cv.notify_one(); }
    void drain() { std::unique_lock<std::mutex> lk(mu); idle.wait(lk, [&] { return q.empty() && !busy; }); }
};
// (the state of an event is shared with the records and waits that are queued on streams: like the runtime's, an event may be destroyed
// while a stream still has a wait for it queued)
struct EventState {
    std::mutex mu; std::condition_variable cv;
    uint64_t recorded = 0, completed = 0;          // generations: record n is complete when completed >= n
    std::chrono::steady_clock::time_point when;
};
struct StubEvent { std::shared_ptr<EventState> s = std::make_shared<EventState>(); };

namespace {
std::mutex g_mu;
std::set<StubStream *> g_streams;
std::map<void *, size_t> g_allocs; size_t g_now = 0, g_peak = 0;
std::atomic<uint64_t> g_streams_made{0}, g_mallocs{0};
thread_local int t_device = 0;
void drain_all()
{
    std::vector<StubStream *> v;
    { std::lock_guard<std::mutex> lk(g_mu); v.assign(g_streams.begin(), g_streams.end()); }
    for (StubStream *s : v) s->drain();
}
uint64_t env_u64(const char *n, uint64_t d) { const char *v = getenv(n); return v && *v ? strtoull(v, nullptr, 10) : d; }
}

const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory" : e == hipErrorInvalidValue ? "invalid value" : "stub error"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipInit(unsigned) { return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { *n = (int)env_u64("STUB_DEVICES", 1); return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = t_device; return hipSuccess; }
hipError_t hipSetDevice(int d) { int n; hipGetDeviceCount(&n); if (d < 0 || d >= n) return hipErrorInvalidValue; t_device = d; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { memset(p, 0, sizeof *p); p->multiProcessorCount = 256; strcpy(p->gcnArchName, "gfx950:stub"); strcpy(p->name, "CPU stand-in"); return hipSuccess; }
hipError_t hipDeviceSynchronize() { drain_all(); return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = -1; return hipSuccess; }
hipError_t hipMemGetInfo(size_t *f, size_t *t) { std::lock_guard<std::mutex> lk(g_mu); *t = (size_t)288 << 30; *f = *t - g_now; return hipSuccess; }
hipError_t hipMalloc(void **p, size_t bytes)
{
    static const uint64_t fail_at = env_u64("STUB_FAIL_MALLOC_AT", 0);
    if (fail_at && ++g_mallocs == fail_at) { *p = nullptr; return hipErrorOutOfMemory; }
    void *q = nullptr;
    if (posix_memalign(&q, 256, bytes ? bytes : 1) != 0) { *p = nullptr; return hipErrorOutOfMemory; }
    { std::lock_guard<std::mutex> lk(g_mu); g_allocs[q] = bytes; g_now += bytes; if (g_now > g_peak) g_peak = g_now; }
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void *p)
{
    if (!p) return hipSuccess;
    drain_all();
    { std::lock_guard<std::mutex> lk(g_mu); auto it = g_allocs.find(p); if (it == g_allocs.end()) return hipErrorInvalidValue; g_now -= it->second; g_allocs.erase(it); }
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { void *q = nullptr; if (posix_memalign(&q, 4096, bytes ? bytes : 1) != 0) return hipErrorOutOfMemory; *p = q; return hipSuccess; }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static std::atomic<long> g_registered{0};
hipError_t hipHostRegister(void *, size_t, unsigned) { if (getenv("STUB_NO_REGISTER")) return hipErrorInvalidValue; g_registered++; return hipSuccess; }
hipError_t hipHostUnregister(void *) { drain_all(); g_registered--; return hipSuccess; }          // (like the runtime's: nothing may still read the range)
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new StubStream(); { std::lock_guard<std::mutex> lk(g_mu); g_streams.insert(*s); } g_streams_made++; return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t *s) { return hipStreamCreateWithFlags(s, 0); }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned f, int) { return hipStreamCreateWithFlags(s, f); }
hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, uint32_t, const uint32_t *) { return hipStreamCreateWithFlags(s, 0); }
hipError_t hipStreamDestroy(hipStream_t s) { if (!s) return hipErrorInvalidValue; s->drain(); { std::lock_guard<std::mutex> lk(g_mu); g_streams.erase(s); } delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { if (!s) drain_all(); else s->drain(); return hipSuccess; }
void stub_enqueue(hipStream_t s, std::function<void()> f) { if (!s) { drain_all(); f(); } else s->push(std::move(f)); }

hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new StubEvent(); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) { if (!e) return hipErrorInvalidValue; delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    std::shared_ptr<EventState> st = e->s;
    uint64_t gen;
    { std::lock_guard<std::mutex> lk(st->mu); gen = ++st->recorded; }
    stub_enqueue(s, [st, gen] { std::lock_guard<std::mutex> lk(st->mu); if (gen > st->completed) { st->completed = gen; st->when = std::chrono::steady_clock::now(); } st->cv.notify_all(); });
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
    std::shared_ptr<EventState> st = e->s;
    uint64_t gen;
    { std::lock_guard<std::mutex> lk(st->mu); gen = st->recorded; }
    if (!gen) return hipSuccess;                                   // (never recorded: no-op, as in the runtime)
    stub_enqueue(s, [st, gen] { std::unique_lock<std::mutex> lk(st->mu); st->cv.wait(lk, [&] { return st->completed >= gen; }); });
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) { std::shared_ptr<EventState> st = e->s; std::unique_lock<std::mutex> lk(st->mu); const uint64_t gen = st->recorded; st->cv.wait(lk, [&] { return st->completed >= gen; }); return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t e) { std::lock_guard<std::mutex> lk(e->s->mu); return e->s->completed >= e->s->recorded ? hipSuccess : hipErrorNotReady; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    std::unique_lock<std::mutex> la(a->s->mu, std::defer_lock), lb(b->s->mu, std::defer_lock);
    if (a->s == b->s) la.lock(); else std::lock(la, lb);
    if (!a->s->completed || !b->s->completed) return hipErrorNotReady;
    *ms = std::chrono::duration<float, std::milli>(b->s->when - a->s->when).count();
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t s) { stub_enqueue(s, [dst, src, n] { if (n) memmove(dst, src, n); }); return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) { drain_all(); if (n) memmove(dst, src, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t s) { stub_enqueue(s, [dst, v, n] { if (n) memset(dst, v, n); }); return hipSuccess; }
uint64_t stub_streams_made() { return g_streams_made; }
uint64_t stub_bytes_allocated_now() { std::lock_guard<std::mutex> lk(g_mu); return g_now; }
uint64_t stub_bytes_allocated_peak() { std::lock_guard<std::mutex> lk(g_mu); return g_peak; }
