// C ABI of libmitofilter_hip (include/mitofilter.h).  Host orchestration only:
// device memory, one stream per device, kernel launches, hipEvent timing.
// There is no CPU compute path in this file: without a gfx950 device every
// compute entry point fails with MF_E_NO_DEVICE.
#include "../../include/mitofilter.h"
#include "mf_common.h"
#include "mf_host.h"
#include "mf_kernels.h"
#include "mf_pipeline.h"
#include "mf_synth.h"
#include "mf_coldtrace.h"
#include "mf_api_internal.h"
#include "mf_devingest.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <string>
#include <thread>
#include <vector>

using namespace mf;

// ------------------------------------------------------------------ errors
static thread_local std::string t_err;
const std::string &mf_thread_error() { return t_err; }
int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    t_err = buf;
    return code;
}
#define HIPCHK(call)                                                                                  \
    do { hipError_t e_ = (call);                                                                      \
         if (e_ != hipSuccess) return fail(MF_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// --------------------------------------------------------------- device ctx
static std::mutex g_ctx_mu;
// "expect_files" (mf_set_option): this process is going to make a file-level call -- what that call's set-up needs of a device (the ingest
// path's streams, its pinned staging buffers) is started in the background the moment the device's first context exists, beside whatever
// the caller does first (the bait set's build).  The CLIs set it: a process per call is the reference's boundary (utility/helper.py:78-86).
static std::atomic<int> g_expect_files{0};
static std::map<int, DevCtx> g_ctx;

// MF_FAKE_DEVICES=N: the library reports N logical devices and maps logical device d onto physical device d mod <visible>.
// Every logical device has its own context (streams), bait tables and read sets, so the multi-device code paths -- batch
// dealing in the file pipeline, one worker per device -- run for real on a single-GPU box (tests; not a performance mode).
// A TEST HOOK: compiled into libmitofilter_hip_hooks.so (-DMF_TEST_HOOKS, what the multi-device tests load) and not into the shipped library.
static int fake_devices()
{
#ifdef MF_TEST_HOOKS
    static const int n = [] { const char *v = getenv("MF_FAKE_DEVICES"); const int k = v ? atoi(v) : 0; return k > 0 && k <= 64 ? k : 0; }();
    return n;
#else
    return 0;
#endif
}
static int physical_count() { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
int phys(int device) { const int n = physical_count(); return fake_devices() && n > 0 ? device % n : device; }

// lane: a device can have several independent contexts (own streams); the file pipeline runs two workers per device so
// that the host-to-device copy of one batch overlaps the kernels and the read-back of the other
int get_ctx(int device, DevCtx **out, int lane)
{
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    auto it = g_ctx.find(device + 4096 * lane);
    if (it != g_ctx.end()) { *out = &it->second; hipError_t e = hipSetDevice(phys(device)); if (e != hipSuccess) return fail(MF_E_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e)); return MF_OK; }
    int n = 0;
    mf::cold_mark("get_ctx: a new device context");
    hipError_t e = hipGetDeviceCount(&n);
    mf::cold_mark("get_ctx: HIP runtime answered (initialised)");
    if (e != hipSuccess || n <= 0) return fail(MF_E_NO_DEVICE, "no HIP device visible (%s); libmitofilter_hip has no CPU fallback", e == hipSuccess ? "count=0" : hipGetErrorString(e));
    const int logical = fake_devices() ? fake_devices() : n;
    if (device < 0 || device >= logical) return fail(MF_E_ARG, "device %d out of range (have %d)", device, logical);
    const int pdev = phys(device);
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, pdev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MF_E_NO_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code only", device, prop.gcnArchName);
    HIPCHK(hipSetDevice(pdev));
    DevCtx c; c.device = device; c.n_cu = prop.multiProcessorCount;
    HIPCHK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    mf::cold_mark("get_ctx: stream made");
    g_ctx[device + 4096 * lane] = c;
    *out = &g_ctx[device + 4096 * lane];
    if (g_expect_files && lane == 0) mf::ingest_prefetch(device);
    return MF_OK;
}

// The streams that only a call of several pipelined passes uses (mf_filter_resident with steps > 1: finish kernels under the next
// screen, every other screen) are made when such a call first comes.  A stream with a priority is a hardware queue of its own and
// takes 17-32 ms to make, a plain one 16-30 ms while the process has fewer than four queues (profiles/r05/a_stream_probe.log):
// a file-level call -- one pass per piece -- through the reference's process-per-call boundary must not pay for three of them per context.
static int ensure_pipeline_streams(DevCtx *c)
{
    // (made outside g_ctx_mu -- every other thread's get_ctx() would wait 50-100 ms behind them -- and published under it; a thread
    // that loses the race, or a creation that fails half way, destroys what it has made)
    { std::lock_guard<std::mutex> lk(g_ctx_mu); if (c->stream2) return MF_OK; }
    int lo = 0, hi = 0;
    hipStream_t s2 = nullptr, s3 = nullptr, s4 = nullptr;
    hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi);          // the short, latency-bound finish kernels that run under the next screen kernel: highest priority
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&s4, hipStreamNonBlocking, hi);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s3, hipStreamNonBlocking);
    bool publish = e == hipSuccess;
    if (publish) {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        if (c->stream2) publish = false;
        else { c->stream3 = s3; c->stream4 = s4; c->stream2 = s2; }
    }
    if (!publish) { for (hipStream_t s : {s2, s4, s3}) if (s) (void)hipStreamDestroy(s); }
    if (e != hipSuccess) return fail(MF_E_HIP, "creating the streams of a pipelined call failed: %s", hipGetErrorString(e));
    return MF_OK;
}

// ------------------------------------------------------------------ kmerset
struct DevTables {
    uint64_t *keys = nullptr;
    uint32_t *bloom = nullptr, *stab = nullptr, *kbloom = nullptr, *kbloom_co = nullptr, *plut = nullptr;    // kbloom_co: own allocation only when it differs from kbloom
    uint32_t *front2 = nullptr, *front3 = nullptr, *pre = nullptr;      // bait-sized fronts of the large-bait screen (front_mode 1 .. 4); mode 4's one-bit LDS table
    KmerSetView view{};
    uint64_t n_keys = 0, n_smers = 0;
};
struct mf_kmerset {
    int k = 0, kw = 1;
    int kind = MF_KIND_NUCLEOTIDE, genetic_code = 0;   // protein sets: k = residues per key, reads translated with genetic_code
    ProtBaitHost pbait;
    uint32_t codon_lut[256] = {0};
    bool kb_in_lds = true;
    BaitHost bait;
    uint64_t n_windows = 0, slots = 0;
    ScreenGeom geom{0, 0};
    uint32_t bloom_log2w = 0, stage2_log2w = 0, stab_slots = 0, kb_log2w = 0;
    uint32_t front_mode = 0, f2_log2b = 0, f3_log2b = 0, pre_log2w = 0;
    int canon = 0;              // != 0: the screen's tables hold one canonical key per bait s-mer (KmerSetView::canon: 1 sixteen-base samples, 2 shorter)
    bool s8_finish = false;     // a stride-8 set whose threshold-1 passes go through screen + finish (baits beyond ~20 kbp)
    size_t screen_words() const { return ((size_t)1 << bloom_log2w) + ((size_t)1 << stage2_log2w); }
    std::mutex mu;
    std::map<int, DevTables> dev;
};

// device temporaries of one build: released on every exit path
struct DevScratch {
    std::vector<void *> bufs;
    template <class T> hipError_t alloc(T *&p, size_t bytes) { hipError_t e = dev_malloc(&p, bytes); if (e == hipSuccess) bufs.push_back(p); return e; }
    ~DevScratch() { for (void *p : bufs) hipFree(p); }
};
// tables under construction: released unless the build commits them
struct TablesGuard {
    DevTables *t;
    ~TablesGuard() { if (t) { hipFree(t->keys); hipFree(t->bloom); hipFree(t->stab); hipFree(t->kbloom); hipFree(t->kbloom_co); hipFree(t->plut); hipFree(t->front2); hipFree(t->front3); hipFree(t->pre); } }
};
// events of one timing loop
struct EventList {
    std::vector<hipEvent_t> ev;
    hipError_t create(size_t n) { ev.reserve(n); for (size_t i = 0; i < n; i++) { hipEvent_t e; hipError_t r = hipEventCreate(&e); if (r != hipSuccess) return r; ev.push_back(e); } return hipSuccess; }
    ~EventList() { for (hipEvent_t e : ev) hipEventDestroy(e); }
};

// host threads for the one-shot host-side jobs (synthetic read sets, packing a FASTQ file): every hardware thread unless
// MF_HOST_THREADS says otherwise (bench.py sets it to its share when several ranks of one node pack their shards at once)
static int host_threads()
{
    const char *v = getenv("MF_HOST_THREADS");
    const int n = v && *v ? atoi(v) : (int)std::thread::hardware_concurrency();
    return n < 1 ? 1 : n;
}

static uint32_t env_u32(const char *name, uint32_t dflt)
{
    const char *v = getenv(name);
    return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}

// ----------------------------------------------------------------- options
// How a screened pass is run.  Default: screen_kernel, then finish_kernel when the threshold is 1 and no hit counts are wanted
// (mark_kernel + exact_kernel otherwise).  MF_PASS=split: always screen, mark, exact.  MF_PASS=serial: screen + finish
// without overlapping consecutive passes (for comparison).
// The switches that select WHICH KERNELS a pass runs live in one options block.  A production process never reads them from the
// environment: they are set through mf_set_option (the CLI's --option name=value), or -- for the test suite, bench.py and the profiling
// scripts -- taken from the MF_* variables when MF_ENV_KNOBS=1 says so.  Every variant is parity-tested (tests/test_gpu_parity.py).
struct PassOptions {
    std::atomic<int> pass{0};             // 0 default (screen + finish, pipelined) | 1 split | 2 serial            "pass"            MF_PASS
    std::atomic<int> adapt{1};            // switch the pass kind from the previous call's tallies                   "adapt"           MF_ADAPT
    std::atomic<int> finish_streams{0};   // 0 by tallies | 1 | 2                                                    "finish_streams"  MF_FINISH_STREAMS
    std::atomic<int> screen_streams{2};   // consecutive screens on one stream or on two in turn                     "screen_streams"  MF_SCREEN_STREAMS
    std::atomic<int> split_pipe{1};       // the candidate-bitmap pass pipelined                                     "split_pipe"      MF_SPLIT_PIPE
    std::atomic<int> exact_co{0};         // the co-resident exact kernel behind every screen (tests)                "exact_co"        MF_EXACT_CO
    // the large-bait screen (read when a k-mer set is BUILT; tests force every form on small baits)
    std::atomic<int> front{-1};           // -1 by the bait's size | 0 LDS table only | 1 LDS table + front2 | 2 front2 (+ front3) only | 3 LDS table, lone positives through front2   "front"   MF_FRONT
    std::atomic<int> front2_log2b{0};     // 0 by the bait's size | log2 of front2's 128-bit blocks (6..18)                         "front2_log2b"   MF_FRONT2_LOG2B
    std::atomic<int> s8_finish{-1};       // the stride-8 geometries (k < 28) through screen + finish instead of the candidate bitmap: -1 by the bait's size | 0 | 1   "s8_finish"   MF_S8_FINISH
    std::atomic<int> canon{-1};           // -1 by the bait's size | 0 both strands in the screen's tables | 1 one canonical key per s-mer (16-base samples only)             "canon"          MF_CANON
    std::atomic<int> front3_log2b{-1};    // -1 by the bait's size | 0 none | log2 of front3's blocks (6..27)                       "front3_log2b"   MF_FRONT3_LOG2B
};
static PassOptions g_opt;
static int set_option(const char *name, const char *value)
{
    const std::string n = name ? name : "", v = value ? value : "";
    if (n == "pass") { if (v == "" || v == "default") g_opt.pass = 0; else if (v == "split") g_opt.pass = 1; else if (v == "serial") g_opt.pass = 2; else return -1; return 0; }
    char *end = nullptr; const long x = strtol(v.c_str(), &end, 10);
    if (v.empty() || *end) return -1;
    if (n == "expect_files") g_expect_files = x != 0;
    else if (n == "short_lived") mf::ingest_short_lived(x != 0);
    else if (n == "adapt") g_opt.adapt = x != 0;
    else if (n == "finish_streams") { if (x < 0 || x > 2) return -1; g_opt.finish_streams = (int)x; }
    else if (n == "screen_streams") { if (x < 1 || x > 2) return -1; g_opt.screen_streams = (int)x; }
    else if (n == "split_pipe") g_opt.split_pipe = x != 0;
    else if (n == "exact_co") g_opt.exact_co = x != 0;
    else if (n == "front") { if (x < -1 || x > 4) return -1; g_opt.front = (int)x; }
    else if (n == "front2_log2b") { if (x != 0 && (x < 6 || x > 24)) return -1; g_opt.front2_log2b = (int)x; }
    else if (n == "s8_finish") { if (x < -1 || x > 1) return -1; g_opt.s8_finish = (int)x; }
    else if (n == "canon") { if (x < -1 || x > 1) return -1; g_opt.canon = (int)x; }
    else if (n == "front3_log2b") { if (x < -1 || (x > 0 && x < 6) || x > 27) return -1; g_opt.front3_log2b = (int)x; }
    else return -1;
    return 0;
}
static void options_from_env_once()
{
    static const bool done = [] {
        const char *k = getenv("MF_ENV_KNOBS");
        if (k && k[0] == '1') {
            static const char *const pairs[][2] = {{"pass", "MF_PASS"}, {"adapt", "MF_ADAPT"}, {"finish_streams", "MF_FINISH_STREAMS"}, {"screen_streams", "MF_SCREEN_STREAMS"},
                                                   {"split_pipe", "MF_SPLIT_PIPE"}, {"exact_co", "MF_EXACT_CO"},
                                                   {"front", "MF_FRONT"}, {"front2_log2b", "MF_FRONT2_LOG2B"}, {"front3_log2b", "MF_FRONT3_LOG2B"}, {"canon", "MF_CANON"}, {"s8_finish", "MF_S8_FINISH"}};
            for (auto &p : pairs) { const char *v = getenv(p[1]); if (v && *v && set_option(p[0], v) != 0) fprintf(stderr, "libmitofilter_hip: %s=%s is not a value of option '%s' (ignored)\n", p[1], v, p[0]); }
        }
        return true;
    }();
    (void)done;
}
static int pass_kind() { options_from_env_once(); return g_opt.pass; }          // (looked up on every pass: bench.py times the serial form next to the default one in one process)

static int build_on_device(mf_kmerset *ks, int device, DevTables **out)
{
    std::lock_guard<std::mutex> lk(ks->mu);
    auto it = ks->dev.find(device);
    if (it != ks->dev.end()) { *out = &it->second; return MF_OK; }
    DevCtx *ctx; int rc = get_ctx(device, &ctx); if (rc) return rc;
    hipStream_t st = ctx->stream;
    if (ks->kind == MF_KIND_PROTEIN) {
        const ProtBaitHost &P = ks->pbait;
        DevTables T; TablesGuard guard{&T}; DevScratch tmp;
        uint8_t *d_aa = nullptr, *d_run = nullptr; unsigned long long *d_cnt = nullptr;
        HIPCHK(tmp.alloc(d_aa, P.aa.size())); HIPCHK(tmp.alloc(d_run, P.runlen.size()));
        HIPCHK(hipMemcpyAsync(d_aa, P.aa.data(), P.aa.size(), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_run, P.runlen.data(), P.runlen.size(), hipMemcpyHostToDevice, st));
        HIPCHK(dev_malloc(&T.keys, ks->slots * sizeof(uint64_t)));
        HIPCHK(hipMemsetAsync(T.keys, 0xFF, ks->slots * sizeof(uint64_t), st));
        HIPCHK(launch_build_ptable(d_aa, d_run, P.total, ks->k, T.keys, ks->slots, st));
        HIPCHK(dev_malloc(&T.kbloom, sizeof(uint32_t) << ks->kb_log2w));
        HIPCHK(hipMemsetAsync(T.kbloom, 0, sizeof(uint32_t) << ks->kb_log2w, st));
        HIPCHK(launch_build_pbits(T.keys, ks->slots, T.kbloom, ks->kb_log2w, st));
        HIPCHK(dev_malloc(&T.plut, sizeof ks->codon_lut));
        HIPCHK(hipMemcpyAsync(T.plut, ks->codon_lut, sizeof ks->codon_lut, hipMemcpyHostToDevice, st));
        HIPCHK(tmp.alloc(d_cnt, 16)); HIPCHK(hipMemsetAsync(d_cnt, 0, 16, st));
        HIPCHK(launch_count_keys(T.keys, ks->slots, 1, nullptr, 0, d_cnt, st));
        unsigned long long cnt[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(cnt, d_cnt, 16, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        T.n_keys = cnt[0];
        KmerSetView &V = T.view;
        V.k = ks->k; V.kw = 1; V.slot_mask = ks->slots - 1; V.keys = T.keys;
        V.kb_log2w = ks->kb_log2w; V.kbloom = T.kbloom; V.kb_co_log2w = ks->kb_log2w; V.kbloom_co = T.kbloom;
        V.prot = 1; V.kb_in_lds = ks->kb_in_lds ? 1u : 0u; V.plut = T.plut;
        ks->dev[device] = T; guard.t = nullptr;
        *out = &ks->dev[device];
        return MF_OK;
    }
    const BaitHost &B = ks->bait;
    DevTables T; TablesGuard guard{&T}; DevScratch tmp;
    uint32_t *d_words = nullptr; uint8_t *d_run = nullptr; uint32_t *d_pos = nullptr, *d_flag = nullptr;
    unsigned long long *d_cnt = nullptr;
    HIPCHK(tmp.alloc(d_words, B.words.size() * 4));
    HIPCHK(tmp.alloc(d_run, B.runlen.size()));
    HIPCHK(hipMemcpyAsync(d_words, B.words.data(), B.words.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_run, B.runlen.data(), B.runlen.size(), hipMemcpyHostToDevice, st));
    const size_t key_bytes = ks->slots * ks->kw * sizeof(uint64_t);
    HIPCHK(dev_malloc(&T.keys, key_bytes));
    HIPCHK(hipMemsetAsync(T.keys, 0xFF, key_bytes, st));
    if (ks->kw == 2) {
        if (B.total > 0xFFFFFFF0ull) return fail(MF_E_ARG, "bait longer than 2^32 bases is not supported for k > 32");
        HIPCHK(tmp.alloc(d_pos, ks->slots * 4));
        HIPCHK(hipMemsetAsync(d_pos, 0xFF, ks->slots * 4, st));
    }
    HIPCHK(tmp.alloc(d_flag, 4)); HIPCHK(hipMemsetAsync(d_flag, 0, 4, st));
    HIPCHK(tmp.alloc(d_cnt, 16)); HIPCHK(hipMemsetAsync(d_cnt, 0, 16, st));
    BaitView bv{d_words, B.total, d_run};
    HIPCHK(launch_build_table(bv, ks->k, ks->kw, T.keys, ks->slots, d_pos, st));
    if (ks->geom.s) {
        HIPCHK(dev_malloc(&T.bloom, ks->screen_words() * 4));
        HIPCHK(hipMemsetAsync(T.bloom, 0, ks->screen_words() * 4, st));
        HIPCHK(dev_malloc(&T.stab, (size_t)ks->stab_slots * 4));
        HIPCHK(hipMemsetAsync(T.stab, 0xFF, (size_t)ks->stab_slots * 4, st));
        if (ks->front_mode) {
            HIPCHK(dev_malloc(&T.front2, (size_t)16 << ks->f2_log2b));
            HIPCHK(hipMemsetAsync(T.front2, 0, (size_t)16 << ks->f2_log2b, st));
            if (ks->f3_log2b) {
                HIPCHK(dev_malloc(&T.front3, (size_t)16 << ks->f3_log2b));
                HIPCHK(hipMemsetAsync(T.front3, 0, (size_t)16 << ks->f3_log2b, st));
            }
        }
        if (ks->front_mode == 4) {
            HIPCHK(dev_malloc(&T.pre, sizeof(uint32_t) << ks->pre_log2w));
            HIPCHK(hipMemsetAsync(T.pre, 0, sizeof(uint32_t) << ks->pre_log2w, st));
        }
        HIPCHK(launch_build_screen(bv, ks->geom.s, T.bloom, ks->bloom_log2w, ks->stage2_log2w, T.stab, ks->stab_slots, d_flag,
                                   T.front2, ks->f2_log2b, T.front3, ks->f3_log2b, T.pre, ks->pre_log2w, ks->canon != 0, st));
    }
    HIPCHK(dev_malloc(&T.kbloom, sizeof(uint32_t) << ks->kb_log2w));
    HIPCHK(hipMemsetAsync(T.kbloom, 0, sizeof(uint32_t) << ks->kb_log2w, st));
    HIPCHK(launch_build_kbloom(T.keys, ks->slots, ks->kw, T.kbloom, ks->kb_log2w, st));
    if (ks->kb_log2w > KB_CO_LOG2W) {        // the folded table of the co-resident exact kernel (pipelined passes)
        HIPCHK(dev_malloc(&T.kbloom_co, sizeof(uint32_t) << KB_CO_LOG2W));
        HIPCHK(hipMemsetAsync(T.kbloom_co, 0, sizeof(uint32_t) << KB_CO_LOG2W, st));
        HIPCHK(launch_build_kbloom(T.keys, ks->slots, ks->kw, T.kbloom_co, KB_CO_LOG2W, st));
    }
    HIPCHK(launch_count_keys(T.keys, ks->slots, ks->kw, T.stab, T.stab ? ks->stab_slots : 0, d_cnt, st));
    unsigned long long cnt[2] = {0, 0}; uint32_t flag = 0;
    HIPCHK(hipMemcpyAsync(cnt, d_cnt, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    T.n_keys = cnt[0]; T.n_smers = cnt[1] + (flag ? 1 : 0);
    KmerSetView &V = T.view;
    V.k = ks->k; V.kw = ks->kw; V.slot_mask = ks->slots - 1; V.keys = T.keys;
    V.kb_log2w = ks->kb_log2w; V.kbloom = T.kbloom;
    V.kb_co_log2w = T.kbloom_co ? KB_CO_LOG2W : ks->kb_log2w; V.kbloom_co = T.kbloom_co ? T.kbloom_co : T.kbloom;
    V.s = ks->geom.s; V.stride = ks->geom.stride;
    V.smask = ks->geom.s >= 16 ? 0xFFFFFFFFu : ((1u << (2 * ks->geom.s)) - 1);
    V.bloom_log2w = ks->bloom_log2w; V.stage2_log2w = ks->stage2_log2w; V.bloom = T.bloom;
    V.stab_mask = ks->stab_slots ? ks->stab_slots - 1 : 0; V.stab = T.stab; V.stab_has_ones = flag;
    V.canon = (uint32_t)ks->canon;
    V.s8_finish = ks->s8_finish ? 1u : 0u;
    V.front_mode = ks->front_mode; V.f2_log2b = ks->f2_log2b; V.f3_log2b = ks->f3_log2b; V.front2 = T.front2; V.front3 = T.front3;
    V.pre_log2w = ks->pre_log2w; V.pre = T.pre;
    // stage 2 holds STAGE2_K bits per canonical s-mer; past ~50 % fill its false-positive rate climbs fast
    V.use_stab = (T.n_smers / 2 * STAGE2_K > ((uint64_t)32 << ks->stage2_log2w) * 7 / 10) ? 1u : 0u;
#ifdef MF_DEBUG_KNOBS              // (experiment builds only: `make variant VARFLAGS=-DMF_DEBUG_KNOBS`)
    if (getenv("MF_USE_STAB")) V.use_stab = 1u;
#endif
    ks->dev[device] = T; guard.t = nullptr;
    *out = &ks->dev[device];
    return MF_OK;
}

static int kmerset_new(const char *text, size_t len, int k, int device, mf_kmerset **out)
{
    if (!out) return fail(MF_E_ARG, "out is NULL");
    *out = nullptr;
    if (k < 11 || k > 63) return fail(MF_E_ARG, "k=%d out of range [11,63]", k);
    mf_kmerset *ks = new (std::nothrow) mf_kmerset();
    if (!ks) return fail(MF_E_NOMEM, "out of memory");
    ks->k = k; ks->kw = k > 32 ? 2 : 1;
    parse_bait_fasta(text, len, ks->bait);
    ks->n_windows = ks->bait.n_windows(k);
    ks->slots = table_slots_for(ks->n_windows);
    {   // LDS k-mer bit table: about two k-mers per 128-bit block, 1 KiB .. 64 KiB (two workgroups per CU)
        uint32_t lg = 8;
        while (lg < 14 && (1ull << lg) < 2 * ks->n_windows) lg++;
        ks->kb_log2w = env_u32("MF_KBLOOM_LOG2W", lg);
        if (ks->kb_log2w < 8) ks->kb_log2w = 8;
        if (ks->kb_log2w > 15) ks->kb_log2w = 15;
    }
    ks->geom = screen_geom_for(k);
#ifdef MF_DEBUG_KNOBS
    if (getenv("MF_NO_SCREEN")) ks->geom = ScreenGeom{0, 0};
#endif
    if (ks->geom.s) {
        const uint64_t bound = 2 * ks->bait.n_swindows(ks->geom.s);
        // stage 1: about one inserted s-mer per 32 bits (4 per 128-bit block), 1 KiB .. 128 KiB of LDS
        uint32_t lg = 8;
        while (lg < 15 && (1ull << lg) < bound) lg++;
        ks->bloom_log2w = env_u32("MF_BLOOM_LOG2W", lg);
        if (ks->bloom_log2w < 8) ks->bloom_log2w = 8;
        if (ks->bloom_log2w > 15) ks->bloom_log2w = 15;
        // stage 2: >= 16 bits per canonical s-mer, 256 B .. 32 KiB (stage 1 + stage 2 <= 160 KiB of LDS)
        uint32_t lg2 = 6;
        while (lg2 < 13 && (32ull << lg2) < 8 * bound) lg2++;
        ks->stage2_log2w = env_u32("MF_STAGE2_LOG2W", lg2);
        if (ks->stage2_log2w < 6) ks->stage2_log2w = 6;
        if (ks->stage2_log2w > 13) ks->stage2_log2w = 13;
        uint64_t ss = 1024; while (ss < 2 * bound) ss <<= 1;
        if (ss > (1ull << 31)) { delete ks; return fail(MF_E_ARG, "bait too large for the s-mer screen table"); }
        ks->stab_slots = (uint32_t)ss;
        // Which screen (mf_kernels.hip, screen2_kernel), by the s-mers a 128-bit block of the LDS table holds.  Up to 9 (a bait of ~37 kbp:
        // the table passes up to 0.5 % of the samples) the records go straight to the finish kernels, as ever (mode 0).  Up to 26 (9 % of
        // the samples at 100 kbp: some 46 positives per wave and chunk of a queue's 64) the positives are looked up in a bait-sized table in
        // L2 before anything is recorded (mode 1).  Beyond that the LDS table passes so much that it is left out and every sample is looked
        // up (mode 2).  Measured where the forms meet (profiles/r06/e_front_variants3.txt, ms a pass): 25 kbp 0.261 (mode 0) against
        // 0.306 (mode 1), 33 kbp 0.306 / 0.320, 50 kbp 0.441 / 0.340; 100 kbp 0.396 (mode 1) against 1.34 (mode 2).
        // front2: about four s-mers a block (0.02 % false positives), at most 2 MiB; front3 behind it where front2 holds more than twelve
        // a block (from ~800 kbp).
        options_from_env_once();
        uint64_t per_lds_block = bound >> (ks->bloom_log2w - 2);
        ks->s8_finish = ks->geom.stride == 8 && per_lds_block > 5;          // (stride-8 sets beyond ~20 kbp: threshold-1 passes through screen + finish, enqueue_pass)
        // Canonical keys (16-base samples, i.e. k >= 31): from the size at which the LDS table stops screening a bait by itself, the screen's tables
        // hold one key per bait s-mer instead of one per strand and every sample is made canonical before it is looked up (six vector
        // instructions a sample, canon16) -- half the load on every table, so each form below reaches twice as far.  The six instructions cost the
        // LDS-table screens 0.03-0.04 ms a pass (16.5 kbp: 0.231 -> 0.266; 50 kbp, queued form: 0.295 -> 0.32), so the keys turn canonical where the
        // queued form with both strands ends (~61 kbp): 70 kbp 0.37 -> 0.32 ms a pass, 100 kbp 0.39 -> 0.36, 150 kbp 0.57 -> 0.40, 200 kbp 0.64 -> 0.45,
        // 350 kbp 0.86 -> 0.62, 1 Mbp 1.42 -> 1.09, 8.5 Mbp 4.6 -> 3.4 (profiles/r06/m_canon.txt).
        // Shorter samples (k < 31) take eight instructions; the stride-16 ones (k = 28 .. 30) turn canonical at the same size, the stride-8 ones (k < 28: twice the
        // samples to make canonical) where their both-strand per-turn form ends (~75 kbp): k = 21 at 100 kbp 0.85 -> 0.70 ms a pass, k = 29 0.45 -> 0.375 (profiles/r06/p_canon_short.txt).
        {
            const bool want = g_opt.canon < 0 ? per_lds_block > (ks->geom.stride == 8 ? 18u : 14u) : g_opt.canon == 1;
            ks->canon = !want ? 0 : ks->geom.s == 16 ? 1 : 2;
        }
        const uint64_t keys_bound = ks->canon ? bound / 2 : bound;          // keys the screen's tables hold
        per_lds_block = keys_bound >> (ks->bloom_log2w - 2);
        // (stride 8: sixteen samples a lane and chunk, so the per-turn form's queue of 64 overflows from ~5 % positives -- 18 keys a block, ~75 kbp -- and what
        // overflows is passed on unverified: k = 21 at 100 kbp 1.51 ms a pass with 20 M work items against 0.85 through the one-bit table, profiles/r06/o_stride8_finish.txt)
        const uint64_t mode1_max = ks->geom.stride == 8 ? 18 : 26;
        int mode = per_lds_block <= 9 ? 0 : per_lds_block <= mode1_max ? 1 : 2;
        if (ks->geom.stride == 16 && per_lds_block > 5 && per_lds_block <= (ks->canon ? 12u : 14u)) mode = 3;          // (canonical keys: 100 kbp 0.36 queued against 0.37 turn by turn, 150 kbp 0.51 against 0.40)
        // mode 2's range up to ~1 Mbp: a ONE-bit table of the LDS's size still answers e^(-keys / 2^20) of the samples itself -- 67 % at 200 kbp, 51 % at
        // 350 kbp, 14 % at 1 Mbp (both strands) -- and only the rest is looked up (mode 4): 0.57 / 0.64 / 0.86 / 1.06 ms a pass at 150 / 200 / 350 / 500 kbp against 1.4-1.5,
        // 1.28 against 1.59 at 700 kbp, 1.42 against 1.63 at 1 Mbp (profiles/r06/k_mode4.txt); beyond two million keys it passes everything and is left out
        if (mode == 2 && keys_bound <= (1u << 21)) mode = 4;
        if (g_opt.front >= 0) mode = g_opt.front;
        ks->front_mode = (uint32_t)mode;
        ks->pre_log2w = mode == 4 ? 15 : 0;
        if (mode) {
            uint32_t lg = 10;
            while (lg < FRONT2_MAX_LOG2B && (4ull << lg) < keys_bound) lg++;
            // (baits of several Mbp: a 2 MiB front2 with more than ~32 s-mers a block passes nearly everything on to front3, a table beyond L2 that
            // answers at a fifth of the rate -- a front2 of 4 or 8 MiB, slower per look-up, saves more of those than it costs:
            // 4 Mbp 3.87 -> 2.59 ms a pass, 8.5 Mbp 6.88 -> 4.64, profiles/r06/d_front_variants2.txt)
            while (lg < FRONT2_MAX_LOG2B + 2 && (keys_bound >> lg) > 32) lg++;
            if (g_opt.front2_log2b > 0) lg = (uint32_t)g_opt.front2_log2b;
            ks->f2_log2b = lg;
            uint32_t lg3 = 0;
            if ((keys_bound >> lg) > 12) { lg3 = lg + 1; while (lg3 < 27 && (4ull << lg3) < keys_bound) lg3++; }
            if (g_opt.front3_log2b >= 0) lg3 = (uint32_t)g_opt.front3_log2b;
            ks->f3_log2b = (mode == 2 || mode == 4) ? lg3 : 0;          // (modes 1 and 3 keep their LDS table and never see a bait that overloads front2)
        }
    }
    DevTables *T; int rc = build_on_device(ks, device, &T);
    if (rc) { delete ks; return rc; }
    *out = ks;
    return MF_OK;
}

static int protset_new(const char *text, size_t len, int kp, int genetic_code, int device, mf_kmerset **out)
{
    if (!out) return fail(MF_E_ARG, "out is NULL");
    *out = nullptr;
    if (kp < 4 || kp > 12) return fail(MF_E_ARG, "peptide k=%d out of range [4,12]", kp);
    mf_kmerset *ks = new (std::nothrow) mf_kmerset();
    if (!ks) return fail(MF_E_NOMEM, "out of memory");
    ks->kind = MF_KIND_PROTEIN; ks->k = kp; ks->kw = 1; ks->genetic_code = genetic_code;
    if (!codon_lut_for(genetic_code, kp, ks->codon_lut)) { delete ks; return fail(MF_E_ARG, "genetic code %d is not supported (1, 2, 3, 4, 5, 9, 11, 13, 14, 21)", genetic_code); }
    parse_bait_protein(text, len, ks->pbait);
    ks->n_windows = ks->pbait.n_windows(kp);
    ks->slots = table_slots_for(ks->n_windows);
    {   // k-mer bit table in front of the open-address table.  Small databases: staged in LDS (<= 64 KiB, about two keys
        // per 128-bit block; still worth it at 32 keys per block, where one probe in six goes on to the table).  Large
        // ones (a whole MT_database clade is ~1 M keys): in global memory at >= 16 bits per key, sized to stay in L2.
        uint32_t lg = 8;
        while (lg < 14 && (1ull << lg) < 2 * ks->n_windows) lg++;
        ks->kb_in_lds = ks->n_windows <= (1u << 17);
        if (!ks->kb_in_lds) { lg = 15; while (lg < 24 && (32ull << lg) < 16 * ks->n_windows) lg++; }
        if (getenv("MF_KBLOOM_LOG2W")) {
            lg = env_u32("MF_KBLOOM_LOG2W", lg);
            if (lg < 8) lg = 8;
            if (lg > 24) lg = 24;
            ks->kb_in_lds = lg <= 14;
        }
        ks->kb_log2w = lg;
    }
    DevTables *T; int rc = build_on_device(ks, device, &T);
    if (rc) { delete ks; return rc; }
    *out = ks;
    return MF_OK;
}

extern "C" {

int mf_abi_version(void) { return MF_ABI_VERSION; }
const char *mf_last_error(void) { return t_err.c_str(); }

int mf_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e == hipErrorNoDevice) return 0;
    if (e != hipSuccess) return fail(MF_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n > 0 && fake_devices() ? fake_devices() : n;
}

int mf_device_name(int device, char *buf, size_t buflen)
{
    if (!buf || !buflen) return fail(MF_E_ARG, "bad buffer");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, phys(device)));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return MF_OK;
}

int mf_device_synchronize(int device)
{
    HIPCHK(hipSetDevice(phys(device)));
    HIPCHK(hipDeviceSynchronize());
    return MF_OK;
}

int mf_kmerset_build_from_text(const char *text, size_t len, int k, int device, mf_kmerset **out)
{
    if (!text && len) return fail(MF_E_ARG, "fasta_text is NULL");
    return kmerset_new(text ? text : "", len, k, device, out);
}

int mf_kmerset_build_from_fasta(const char *path, int k, int device, mf_kmerset **out)
{
    if (!path) return fail(MF_E_ARG, "fasta_path is NULL");
    std::vector<char> buf; std::string err;
    if (!slurp_file(path, buf, err)) return fail(MF_E_IO, "%s", err.c_str());
    return kmerset_new(buf.data(), buf.size(), k, device, out);
}

int mf_kmerset_build_protein_from_text(const char *text, size_t len, int kp, int genetic_code, int device, mf_kmerset **out)
{
    if (!text && len) return fail(MF_E_ARG, "protein_fasta_text is NULL");
    return protset_new(text ? text : "", len, kp, genetic_code, device, out);
}

int mf_kmerset_build_protein_from_fasta(const char *path, int kp, int genetic_code, int device, mf_kmerset **out)
{
    if (!path) return fail(MF_E_ARG, "protein_fasta_path is NULL");
    std::vector<char> buf; std::string err;
    if (!slurp_file(path, buf, err)) return fail(MF_E_IO, "%s", err.c_str());
    return protset_new(buf.data(), buf.size(), kp, genetic_code, device, out);
}

int mf_kmerset_info(const mf_kmerset *ks, mf_kmerset_info_t *info)
{
    if (!ks || !info) return fail(MF_E_ARG, "NULL argument");
    memset(info, 0, sizeof *info);
    info->k = ks->k; info->key_words = ks->kw; info->slots = ks->slots; info->n_windows = ks->n_windows;
    info->screen_s = ks->geom.s; info->screen_stride = ks->geom.stride;
    info->bloom_words = ks->geom.s ? (uint32_t)ks->screen_words() : 0; info->smer_slots = ks->stab_slots;
    if (!ks->dev.empty()) { info->n_keys = ks->dev.begin()->second.n_keys; info->n_smers = ks->dev.begin()->second.n_smers; }
    info->kind = ks->kind; info->genetic_code = ks->genetic_code;
    info->front_mode = ks->front_mode; info->front2_log2_blocks = ks->front_mode ? ks->f2_log2b : 0; info->front3_log2_blocks = ks->f3_log2b; info->canonical_screen = ks->canon ? 1u : 0u;
    return MF_OK;
}

int mf_kmerset_export(const mf_kmerset *ks_, int device, uint64_t *keys_out, size_t n_u64)
{
    mf_kmerset *ks = const_cast<mf_kmerset *>(ks_);
    if (!ks || !keys_out) return fail(MF_E_ARG, "NULL argument");
    if (n_u64 < ks->slots * ks->kw) return fail(MF_E_ARG, "keys_out too small: need %llu u64", (unsigned long long)(ks->slots * ks->kw));
    DevTables *T; int rc = build_on_device(ks, device, &T); if (rc) return rc;
    HIPCHK(hipMemcpy(keys_out, T->keys, ks->slots * ks->kw * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return MF_OK;
}

int mf_kmerset_free(mf_kmerset *ks)
{
    if (!ks) return MF_OK;
    for (auto &kv : ks->dev) {
        if (hipSetDevice(phys(kv.first)) == hipSuccess) { hipFree(kv.second.keys); hipFree(kv.second.bloom); hipFree(kv.second.stab); hipFree(kv.second.kbloom); hipFree(kv.second.kbloom_co); hipFree(kv.second.plut); hipFree(kv.second.front2); hipFree(kv.second.front3); hipFree(kv.second.pre); }
    }
    delete ks;
    return MF_OK;
}

} // extern "C"

// -------------------------------------------------------------------- reads
void reads_release(mf_reads *r)
{
    if (!r) return;
    if (hipSetDevice(phys(r->device)) == hipSuccess) {
        hipFree(r->d_words); hipFree(r->d_offsets); hipFree(r->d_npos); hipFree(r->d_has_n);
        for (int i = 0; i < NSETS; i++) hipFree(r->d_cand[i]);
        for (int i = 0; i < NSETS; i++) {
            hipFree(r->d_bits[i]); hipFree(r->d_recs[i]); hipFree(r->d_rec_counts[i]); if (r->d_counters[i]) hipHostFree(r->d_counters[i]);
            if (r->ev_screen[i]) hipEventDestroy(r->ev_screen[i]);
            if (i < 2 && r->ev_call[i]) hipEventDestroy(r->ev_call[i]);
            if (r->ev_finish[i]) hipEventDestroy(r->ev_finish[i]);
        }
        hipFree(r->d_hits); hipFree(r->d_npos_blk); hipFree(r->d_off_blk);
    }
    delete r;
}

#define RCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(MF_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)
int reads_reserve(mf_reads *r, bool reuse, uint64_t n_words, uint64_t n_reads, uint32_t uniform_len, uint64_t npos_cap, DevCtx *ctx)
{
    const uint64_t padded = padded_words_for(n_words);                                   // readable and zero past the data
    RCHK(dev_reserve(r->d_words, r->cap_words, padded * 4, reuse));
    if (padded > n_words) RCHK(hipMemsetAsync(r->d_words + n_words, 0, (padded - n_words) * 4, ctx->stream));
    if (!uniform_len) RCHK(dev_reserve(r->d_offsets, r->cap_offsets, (n_reads + 1) * 8, reuse));
    RCHK(dev_reserve(r->d_npos, r->cap_npos, (npos_cap ? npos_cap : 1) * 8, reuse));
    return MF_OK;
}

int reads_finish(mf_reads *r, bool reuse, uint64_t n_words, uint64_t n_reads, uint64_t total_bases, uint32_t uniform_len, uint64_t n_npos, DevCtx *ctx)
{
    hipStream_t st = ctx->stream;
    const uint64_t padded = padded_words_for(n_words);
    if (n_npos >= 0xFFFFFFFFull) return fail(MF_E_ARG, "more than 2^32 invalid bases in one read set");
    const uint64_t n_blk = (total_bases >> NPOS_BLK_SHIFT) + 3;
    RCHK(dev_reserve(r->d_npos_blk, r->cap_npos_blk, n_blk * 4, reuse));
    RCHK(launch_build_npos_blk(r->d_npos, n_npos, n_blk, r->d_npos_blk, st));
    if (!uniform_len) {                        // ragged reads: the block index over the offsets (a read per 128 bases of the stream)
        if (n_reads >= 0x7FFFFFFFull) return fail(MF_E_ARG, "more than 2^31 reads in one ragged read set");
        const uint64_t n_oblk = (total_bases >> OFF_BLK_SHIFT) + 2;
        RCHK(dev_reserve(r->d_off_blk, r->cap_off_blk, n_oblk * 8, reuse));
        RCHK(launch_build_off_blk(r->d_offsets, n_reads, n_oblk, r->d_off_blk, st));
    }
    r->bitmap_bytes = ((n_reads + 31) / 32 + 64) * 4;
    if (r->bitmap_bytes > r->cap_bitmap || !r->d_has_n) {     // the four bitmaps share one capacity
        size_t c[2 + NSETS] = {};
        uint32_t **bm[2 + NSETS] = {&r->d_has_n, &r->d_cand[0]};
        for (int i = 0; i < NSETS; i++) bm[2 + i] = &r->d_bits[i];
        size_t least = ~(size_t)0;
        for (int i = 0; i < 2 + NSETS; i++) {
            if (*bm[i]) { hipFree(*bm[i]); *bm[i] = nullptr; }
            RCHK(dev_reserve(*bm[i], c[i], r->bitmap_bytes, reuse));
            if (c[i] < least) least = c[i];
        }
        r->cap_bitmap = least;
        for (int i = 1; i < NSETS; i++) { hipFree(r->d_cand[i]); r->d_cand[i] = nullptr; }      // (follow on their next use)
    }
    RCHK(hipMemsetAsync(r->d_has_n, 0, r->bitmap_bytes, st));
    RCHK(hipMemsetAsync(r->d_cand[0], 0, r->bitmap_bytes, st));
    r->cand_clean[0] = true;
    for (int i = 1; i < NSETS; i++) r->cand_clean[i] = false;          // (the twins are allocated and cleared on first use)
    for (int i = 0; i < NSETS; i++) {
        if (!r->d_counters[i]) RCHK(hipHostMalloc(reinterpret_cast<void **>(&r->d_counters[i]), 3 * EXACT_MAX_GRID * 16, hipHostMallocDefault));
        memset(r->d_counters[i], 0, 3 * EXACT_MAX_GRID * 16);             // (no kernel of this handle is in flight: every call ends synchronised)
        RCHK(hipMemsetAsync(r->d_bits[i], 0, r->bitmap_bytes, st));
        if (!r->ev_screen[i]) RCHK(hipEventCreate(&r->ev_screen[i]));         // (attached to dispatches as completion events)
        if (!r->ev_finish[i]) RCHK(hipEventCreate(&r->ev_finish[i]));
    }
    r->cur = 0;
    ReadsView &V = r->v;
    V = ReadsView{};
    V.words = r->d_words; V.n_words = n_words; V.n_vec = (padded - 16) / 4;
    V.offsets = uniform_len ? nullptr : r->d_offsets; V.off_blk = uniform_len ? nullptr : r->d_off_blk; V.uniform_len = uniform_len; V.n_reads = n_reads; V.total_bases = total_bases;
    V.len_magic = uniform_len > 1 ? ~0ULL / uniform_len + 1 : 0;
    V.len_magic32 = (uniform_len > 1 && uniform_len <= 4096) ? 0xFFFFFFFFu / uniform_len + 1 : 0;
    V.npos = r->d_npos; V.n_npos = n_npos; V.npos_blk = r->d_npos_blk; V.has_n = r->d_has_n;
    {   // worst case one 16-byte record per lane per chunk (a quarter of the packed stream); typical use is ~0.2 %.  The second
        // list is only needed by pipelined threshold-1 passes and is allocated on first use (enqueue_pass).
        // (the stride-16 and the stride-8 screens deal the chunks to different numbers of lists: room for either)
        // (key 4: two workgroups a CU -- the most lists, so its grid sizes the count array; room for the records of any of the three)
        uint64_t grid = screen_grid_for(V, ctx->n_cu, 4), cap = screen_rec_cap_for(V, ctx->n_cu, 4);
        for (int key : {8, 16}) { const uint64_t g2 = screen_grid_for(V, ctx->n_cu, key), c2 = screen_rec_cap_for(V, ctx->n_cu, key); if (g2 * c2 > grid * cap) cap = (g2 * c2 + grid - 1) / (grid ? grid : 1); }
        size_t c0 = r->cap_recs, c1 = r->cap_rec_counts;
        RCHK(dev_reserve(r->d_recs[0], c0, (grid * cap ? grid * cap : 1) * 16, reuse));
        RCHK(dev_reserve(r->d_rec_counts[0], c1, (grid ? grid : 1) * 4, reuse));
        if (c0 != r->cap_recs || c1 != r->cap_rec_counts)                        // grown: the other sets follow on their next use
            for (int i = 1; i < NSETS; i++) { hipFree(r->d_recs[i]); hipFree(r->d_rec_counts[i]); r->d_recs[i] = nullptr; r->d_rec_counts[i] = nullptr; }
        r->cap_recs = c0; r->cap_rec_counts = c1;
    }
    RCHK(launch_mark_has_n(V, r->d_has_n, st));
    RCHK(hipStreamSynchronize(st));
    return MF_OK;
}

// Fill `r` (fresh, or holding buffers of an earlier batch on the same device) with one packed read set from host memory.
// words: host buffer; already_padded = it extends to padded_words_for(n_words) with a zero tail.
static int reads_fill(mf_reads *r, bool reuse, const uint32_t *words, uint64_t n_words, bool already_padded, const uint64_t *offsets,
                      uint64_t n_reads, uint64_t total_bases, uint32_t uniform_len, const uint64_t *npos, uint64_t n_npos,
                      DevCtx *ctx)
{
    hipStream_t st = ctx->stream;
    int rc = reads_reserve(r, reuse, n_words, n_reads, uniform_len, n_npos, ctx);
    if (rc) return rc;
    const uint64_t copied = already_padded ? padded_words_for(n_words) : n_words;
    if (copied) RCHK(hipMemcpyAsync(r->d_words, words, copied * 4, hipMemcpyHostToDevice, st));
    if (!uniform_len) RCHK(hipMemcpyAsync(r->d_offsets, offsets, (n_reads + 1) * 8, hipMemcpyHostToDevice, st));
    if (n_npos) RCHK(hipMemcpyAsync(r->d_npos, npos, n_npos * 8, hipMemcpyHostToDevice, st));
    return reads_finish(r, reuse, n_words, n_reads, total_bases, uniform_len, n_npos, ctx);
}
#undef RCHK

static int reads_upload(const uint32_t *words, uint64_t n_words, bool already_padded, const uint64_t *offsets, uint64_t n_reads,
                        uint64_t total_bases, uint32_t uniform_len, const uint64_t *npos, uint64_t n_npos, int device,
                        mf_reads **out)
{
    DevCtx *ctx; int rc = get_ctx(device, &ctx); if (rc) return rc;
    mf_reads *r = new (std::nothrow) mf_reads();
    if (!r) return fail(MF_E_NOMEM, "out of memory");
    r->device = device;
    rc = reads_fill(r, false, words, n_words, already_padded, offsets, n_reads, total_bases, uniform_len, npos, n_npos, ctx);
    if (rc) { reads_release(r); return rc; }
    *out = r;
    return MF_OK;
}

extern "C" {

int mf_reads_from_packed(const uint32_t *words, const uint64_t *offsets, uint64_t n_reads,
                         const uint64_t *npos, uint64_t n_npos, int device, mf_reads **out)
{
    if (!out) return fail(MF_E_ARG, "out is NULL");
    *out = nullptr;
    if (!offsets) return fail(MF_E_ARG, "offsets is NULL");
    if (n_npos && !npos) return fail(MF_E_ARG, "npos is NULL");
    if (offsets[0] != 0) return fail(MF_E_ARG, "offsets[0] must be 0");
    for (uint64_t i = 0; i < n_reads; i++) if (offsets[i + 1] < offsets[i]) return fail(MF_E_ARG, "offsets must be non-decreasing");
    for (uint64_t i = 1; i < n_npos; i++) if (npos[i] <= npos[i - 1]) return fail(MF_E_ARG, "npos must be strictly ascending");
    const uint64_t total = offsets[n_reads];
    if (n_npos && npos[n_npos - 1] >= total) return fail(MF_E_ARG, "npos entry beyond the last base");
    const uint64_t n_words = (total + 15) / 16;
    if (n_words && !words) return fail(MF_E_ARG, "words is NULL");
    return reads_upload(words, n_words, false, offsets, n_reads, total, detect_uniform_len(offsets, n_reads), npos, n_npos, device, out);
}

int mf_reads_from_fastq(const char *path, int device, mf_reads **out)
{
    if (!out || !path) return fail(MF_E_ARG, "NULL argument");
    *out = nullptr;
    std::vector<char> buf; std::string err;
    if (!slurp_file(path, buf, err)) return fail(MF_E_IO, "%s", err.c_str());
    std::vector<FqRec> recs; parse_fastq(buf.data(), buf.size(), recs);
    PackedHost P; pack_records(recs.data(), recs.size(), host_threads(), P);
    return reads_upload(P.words.data(), P.n_words, true, P.offsets.data(), recs.size(), P.offsets.back(), P.uniform_len,
                        P.npos.data(), P.npos.size(), device, out);
}

int mf_reads_synth(uint64_t n_reads, uint32_t read_len, uint64_t seed, const char *bait_text, size_t bait_len,
                   uint32_t mito_ppm, uint32_t sub_ppm, uint32_t n_read_ppm, uint32_t n_base_ppm, int device, mf_reads **out,
                   uint32_t **host_words_out, uint64_t *host_n_words_out, uint64_t **host_npos_out, uint64_t *host_n_npos_out)
{
    return mf_reads_synth_ex(n_reads, read_len, seed, bait_text, bait_len, mito_ppm, sub_ppm, n_read_ppm, n_base_ppm, 0, 0, 0, device, out,
                             host_words_out, host_n_words_out, host_npos_out, host_n_npos_out);
}

int mf_reads_synth_ex(uint64_t n_reads, uint32_t read_len, uint64_t seed, const char *bait_text, size_t bait_len,
                      uint32_t mito_ppm, uint32_t sub_ppm, uint32_t n_read_ppm, uint32_t n_base_ppm,
                      uint32_t msat_ppm, uint32_t numt_ppm, uint32_t numt_div_ppm, int device, mf_reads **out,
                      uint32_t **host_words_out, uint64_t *host_n_words_out, uint64_t **host_npos_out, uint64_t *host_n_npos_out)
{
    if (!out) return fail(MF_E_ARG, "out is NULL");
    *out = nullptr;
    if (read_len == 0) return fail(MF_E_ARG, "read_len must be > 0");
    BaitHost B; parse_bait_fasta(bait_text ? bait_text : "", bait_text ? bait_len : 0, B);
    SynthOut S;
    std::string err;
    SynthExtra extra; extra.msat_ppm = msat_ppm; extra.numt_ppm = numt_ppm; extra.numt_div_ppm = numt_div_ppm;
    if (!synth_reads(n_reads, read_len, seed, B, mito_ppm, sub_ppm, n_read_ppm, n_base_ppm,
                     host_threads(), S, err, extra)) return fail(MF_E_ARG, "%s", err.c_str());
    int rc = reads_upload(S.words.data(), S.n_words, true, nullptr, n_reads, n_reads * (uint64_t)read_len, read_len,
                          S.npos.data(), S.npos.size(), device, out);
    if (rc) return rc;
    if (host_words_out) {
        *host_words_out = (uint32_t *)malloc(S.words.size() * 4);
        if (!*host_words_out) return fail(MF_E_NOMEM, "out of memory");
        memcpy(*host_words_out, S.words.data(), S.words.size() * 4);
        if (host_n_words_out) *host_n_words_out = S.n_words;
    }
    if (host_npos_out) {
        *host_npos_out = (uint64_t *)malloc((S.npos.size() + 1) * 8);
        if (!*host_npos_out) return fail(MF_E_NOMEM, "out of memory");
        memcpy(*host_npos_out, S.npos.data(), S.npos.size() * 8);
        if (host_n_npos_out) *host_n_npos_out = S.npos.size();
    }
    return MF_OK;
}

void mf_free_host(void *p) { free(p); }

int mf_reads_info(const mf_reads *r, mf_reads_info_t *info)
{
    if (!r || !info) return fail(MF_E_ARG, "NULL argument");
    info->n_reads = r->v.n_reads; info->total_bases = r->v.total_bases; info->n_invalid = r->v.n_npos;
    info->uniform_len = r->v.uniform_len; info->device = r->device;
    return MF_OK;
}

int mf_reads_free(mf_reads *r) { reads_release(r); return MF_OK; }

} // extern "C"

// ------------------------------------------------------------------- filter
static uint64_t algorithmic_bytes(const ReadsView &V) { return (2 * V.total_bases + 7) / 8 + (V.n_reads + 7) / 8; }

// enqueue one pass.  ev (when non-null) holds six events that are attached to the kernels themselves (start/stop of
// screen, mark, exact or finish): each pair reads that dispatch's own duration and the streams carry no extra packets.
// `overlap`: a threshold-1 pass may leave its finish kernel running on the second stream (filter_common joins the streams).
// `more`: another pass of the same call follows (its screen kernel is what this pass's later kernels run beside).
// where the kernels of a pass leave their tallies: the buffer set's pinned block -- or, when the caller wants every pass's
// tally (mf_filter_resident_passes), a block of that pass's own
static unsigned long long *tally_of(mf_reads *r, int set) { return r->tally_override ? r->tally_override : r->d_counters[set]; }

static int enqueue_pass(mf_reads *r, const KmerSetView &S, uint32_t thr, int mode, bool count_all, DevCtx *ctx, hipEvent_t *ev, bool overlap,
                        bool more = false)
{
    r->sample_pass = false;
    hipStream_t st = ctx->stream;
    const int n_cu = ctx->n_cu;
    KernelTiming tm[3]; const KernelTiming *t0 = nullptr, *t1 = nullptr, *t2 = nullptr;
    if (ev) { for (int i = 0; i < 3; i++) tm[i] = KernelTiming{ev[2 * i], ev[2 * i + 1]}; t0 = &tm[0]; t1 = &tm[1]; t2 = &tm[2]; }
    const int p = r->cur;
    if (S.prot) {          // protein-space set: one kernel translates and probes every read (no screen exists in residue space)
        HIPCHK(launch_pfilter(r->v, S, thr, count_all, r->d_bits[p], r->d_hits, tally_of(r, p), n_cu, st, t2));
        return MF_OK;
    }
    const bool screened = (mode == MF_MODE_SCREENED) && S.s > 0;
    // (stride-8 geometries, k < 28: twice the samples, several times the records -- measured faster through the candidate bitmap for a bait the LDS table
    // screens well, 16.5 kbp: k = 21 0.314 against 0.319 ms a pass, k = 25 / 27 0.292 against 0.298.  Beyond ~20 kbp the candidate bitmap's exact kernel is
    // what a pass waits for -- its LDS k-mer table fills up -- and screen + finish is faster: k = 21 33 kbp 0.50 -> 0.45, 50 kbp 0.74 -> 0.55, 100 kbp
    // 1.88 -> 1.51, 350 kbp 3.03 -> 1.98, k = 25 100 kbp 1.68 -> 1.21: KmerSetView::s8_finish, profiles/r06/o_stride8_finish.txt)
    // (round 3, after the stage-1 fields were fixed for 14-base samples: still the faster pass for stride 8 -- k = 21 0.299 vs 0.302-0.309 ms, k = 25 0.278-0.280 vs 0.281-0.283)
    if (screened && pass_kind() != 1 && !r->prefer_split && thr == 1 && !count_all && (S.stride == 16 || pass_kind() == 2 || (g_opt.s8_finish < 0 ? S.s8_finish != 0 : g_opt.s8_finish == 1))) {
        // Two launches: the screen records its stage-1 positives (and clears this pass's result bitmap on the side), the
        // finish kernel settles them and sets the pass bits with atomics.  Pass i works on buffer set i mod 2; its finish
        // kernel goes to the second stream and runs under the screen of pass i + 1, which uses the other set.
        // (three buffer sets for pipelined passes: the screen of pass i + 3 waits for the finish kernels of pass i, not of pass i + 1 -- with two sets a
        // finish chain that outlasts the next screen, as the two-word keys' does, held the screen after that: k = 41 0.243 -> 0.234 ms a pass, k = 63 0.257 -> 0.250 (k = 31
        // 0.224 -> 0.222, 33 kbp 0.243 -> 0.237), profiles/r06/n_three_sets.txt.  A call of one pass -- a file-level call's batches -- keeps to two.)
        // (Kept to the two-word keys: for k <= 32 it is worth 1-2 %, and with three sets two or three screens are in flight at a time, so that a launch
        // lasts twice what a pass takes -- the per-launch figure bench.py's `roofline` reports for the headline would no longer say what the pass does.)
        const bool two = overlap && pass_kind() == 0;
        const int q = (p + 1) % (two && S.kw == 2 ? NSETS : 2);
        if (!r->d_recs[q]) {
            size_t c0 = 0, c1 = 0;
            HIPCHK(dev_reserve(r->d_recs[q], c0, r->cap_recs, false));
            HIPCHK(dev_reserve(r->d_rec_counts[q], c1, r->cap_rec_counts, false));
        }
        // The finish kernels are chains of memory latencies.  With few records (the benchmark's 0.5 % bait reads) they are over long
        // before the next screen is and one stream carries them all; when they are what a pass waits for (bait-rich input: 2 % bait
        // reads and more, seen in the last call's tallies) those of consecutive passes go to two streams and run side by side --
        // 2 %: 0.303 -> 0.280 ms per pass, 10 %: 0.666 -> 0.596; at 0.5 % the same costs 2 % (0.208 -> 0.213).  MF_FINISH_STREAMS=1 / 2 forces.
        const uint32_t fin_streams = (uint32_t)g_opt.finish_streams;
        // (two-word keys, k >= 33: a finish kernel's probes are twice as long, and one stream's worth of them is not over when the next screen is --
        // k = 41 0.2369 -> 0.2331 ms a pass, k = 63 0.2960 -> 0.2817: profiles/r05/c_k41_finish_streams_probe.txt)
        const bool fin2 = fin_streams == 2 || (fin_streams == 0 && (r->finish_two || S.kw == 2));
        const int odd = (r->flip ^= 1);
        hipStream_t sf = two ? ((fin2 && odd) ? ctx->stream4 : ctx->stream2) : st;
        // consecutive screens go to two streams in turn: nothing orders them against each other (different buffer sets), so the
        // workgroups of the next screen take over the CUs as the last ones of this screen drain (MF_SCREEN_STREAMS=1: one stream)
        const bool alt = g_opt.screen_streams == 2;
        hipStream_t ss = (two && alt && odd) ? ctx->stream3 : st;
        if (two) HIPCHK(hipStreamWaitEvent(ss, r->ev_finish[q], 0));           // the finish kernels of NSETS passes ago read this set
        // cross-stream order without marker packets in the screen's stream: the events ride on the dispatches themselves
        // (hipExtLaunchKernelGGL completion events); a separately recorded event costs the next dispatch ~5 us
        KernelTiming scr_done{nullptr, r->ev_screen[q]};
        const KernelTiming *ts = t0 ? t0 : (two ? &scr_done : nullptr);
        HIPCHK(launch_screen(r->v, S, r->d_recs[q], r->d_rec_counts[q], n_cu, ss, ts, r->d_bits[q], ((r->v.n_reads + 31) / 32 + 3) / 4));
        if (two) { if (t0) HIPCHK(hipEventRecord(r->ev_screen[q], ss)); HIPCHK(hipStreamWaitEvent(sf, r->ev_screen[q], 0)); }
#ifdef MF_DEBUG_KNOBS              // timing experiment: the pass without its finish kernels (WRONG result bits)
        static const uint32_t nofin = env_u32("MF_NO_FINISH", 0);
        if (nofin) { if (two) HIPCHK(hipEventRecord(r->ev_finish[q], sf)); } else
#endif
        {
            // For k >= 48 (runs of four and more samples: fewer reads are settled by a run) phase 1 hands the reads that hold a bait s-mer outside
            // any run to an exact kernel behind it, which deals a read's windows to eight lanes, instead of counting them on the one lane that met
            // the s-mer: k = 63 0.283 -> 0.257 ms a pass.  Below that the third launch costs more than the tail it removes (k = 31 0.219 -> 0.226,
            // k = 41 0.235 -> 0.248, 33 kbp bait 0.228 -> 0.262: profiles/r06/j_finish_exact_ab.txt).  The candidate bitmap of set q is clean
            // (cleared when made, and the exact kernel clears what it consumes).
#ifndef MF_FINISH_EXACT
#define MF_FINISH_EXACT (S.k >= 48)
#endif
            uint32_t *cand = nullptr;
            if (MF_FINISH_EXACT) {
                if (!r->d_cand[q]) { size_t c = 0; HIPCHK(dev_reserve(r->d_cand[q], c, r->cap_bitmap, false)); r->cand_clean[q] = false; }
                if (!r->cand_clean[q]) { HIPCHK(hipMemsetAsync(r->d_cand[q], 0, r->bitmap_bytes, sf)); r->cand_clean[q] = true; }
                cand = r->d_cand[q];
            }
            HIPCHK(launch_finish(r->v, S, r->d_recs[q], r->d_rec_counts[q], r->d_bits[q], tally_of(r, q), n_cu, sf, t2, (two && !cand) ? r->ev_finish[q] : nullptr, cand));
            if (cand) HIPCHK(launch_exact(r->v, S, cand, 1, false, r->d_bits[q], nullptr, tally_of(r, q) + 4 * (size_t)EXACT_MAX_GRID, n_cu, sf, nullptr, more, two ? r->ev_finish[q] : nullptr, true));
        }
        r->sample_pass = true;
        r->cur = q;
        return MF_OK;
    }
    // split / exhaustive: no per-pass memsets -- the exact kernel clears the candidate words it consumes, writes every
    // result word and zeroes unused tally slots
    const bool split_pipe = g_opt.split_pipe != 0;
    const bool exact_co = g_opt.exact_co != 0;           // (tests: the co-resident exact kernel behind every screen)
    if (screened && !count_all && overlap && split_pipe && pass_kind() != 2 && !r->split_serial) {
        // The three-kernel pass, pipelined like the one above: pass i works on buffer set i mod 2 (records, candidate
        // bitmap, result bitmap, tallies); its mark and exact kernels go to the second stream and run beside the screen of
        // pass i + 1.  The exact kernel takes its co-resident form when a screen follows (a screen workgroup holds 128 KiB
        // of every CU's LDS for the whole pass), its full form behind the last screen of the call.
        const int q = (p + 1) % (S.kw == 2 ? NSETS : 2);
        if (!r->d_recs[q]) {
            size_t c0 = 0, c1 = 0;
            HIPCHK(dev_reserve(r->d_recs[q], c0, r->cap_recs, false));
            HIPCHK(dev_reserve(r->d_rec_counts[q], c1, r->cap_rec_counts, false));
        }
        if (!r->d_cand[q]) { size_t c = 0; HIPCHK(dev_reserve(r->d_cand[q], c, r->cap_bitmap, false)); r->cand_clean[q] = false; }
        const bool alt = g_opt.screen_streams == 2;
        const int odd = (r->flip ^= 1);
        hipStream_t ss = (alt && odd) ? ctx->stream3 : st, sf = ctx->stream2;
        HIPCHK(hipStreamWaitEvent(ss, r->ev_finish[q], 0));                    // the exact kernel of NSETS passes ago worked on this set
        if (!r->cand_clean[q]) { HIPCHK(hipMemsetAsync(r->d_cand[q], 0, r->bitmap_bytes, ss)); r->cand_clean[q] = true; }
        KernelTiming scr_done{nullptr, r->ev_screen[q]};
        HIPCHK(launch_screen(r->v, S, r->d_recs[q], r->d_rec_counts[q], n_cu, ss, t0 ? t0 : &scr_done));
        if (t0) HIPCHK(hipEventRecord(r->ev_screen[q], ss));
        HIPCHK(hipStreamWaitEvent(sf, r->ev_screen[q], 0));
        HIPCHK(launch_mark(r->v, S, r->d_recs[q], r->d_rec_counts[q], r->d_cand[q], n_cu, sf, t1));
        HIPCHK(launch_exact(r->v, S, r->d_cand[q], thr, false, r->d_bits[q], nullptr, tally_of(r, q), n_cu, sf, t2, more || exact_co, r->ev_finish[q]));
        r->cur = q;
        return MF_OK;
    }
    // one stream, one buffer set (hit counts wanted, the exhaustive mode, MF_PASS=serial)
    if (screened && !r->cand_clean[0]) { HIPCHK(hipMemsetAsync(r->d_cand[0], 0, r->bitmap_bytes, st)); r->cand_clean[0] = true; }
    if (screened) HIPCHK(launch_screen(r->v, S, r->d_recs[0], r->d_rec_counts[0], n_cu, st, t0));
    if (screened) HIPCHK(launch_mark(r->v, S, r->d_recs[0], r->d_rec_counts[0], r->d_cand[0], n_cu, st, t1));
    HIPCHK(launch_exact(r->v, S, screened ? r->d_cand[0] : nullptr, thr, count_all, r->d_bits[p], r->d_hits, tally_of(r, p), n_cu, st, t2));
    return MF_OK;
}

int filter_common(const mf_kmerset *ks_, const mf_reads *reads_, uint32_t thr, int mode, uint32_t *out_bits,
                  uint32_t *hits_out, int steps, mf_filter_stats_t *stats, uint64_t *pass_per_step)
{
    mf_kmerset *ks = const_cast<mf_kmerset *>(ks_);
    mf_reads *r = const_cast<mf_reads *>(reads_);
    if (!ks || !r) return fail(MF_E_ARG, "NULL handle");
    if (thr < 1) return fail(MF_E_ARG, "threshold must be >= 1");
    if (mode != MF_MODE_SCREENED && mode != MF_MODE_EXHAUSTIVE) return fail(MF_E_ARG, "bad mode %d", mode);
    if (steps < 1) return fail(MF_E_ARG, "steps must be >= 1");
    static const bool trace = getenv("MF_TRACE") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (trace) fprintf(stderr, "[mf trace] %-10s %8.1f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tp0).count()); };
    DevCtx *ctx; int rc = get_ctx(r->device, &ctx, r->lane); if (rc) return rc;
    DevTables *T; rc = build_on_device(ks, r->device, &T); if (rc) return rc;
    hipStream_t st = ctx->stream;
    const bool count_all = hits_out != nullptr;
    lap("setup");
    if (count_all) {
        HIPCHK(dev_reserve(r->d_hits, r->cap_hits, (r->v.n_reads ? r->v.n_reads : 1) * 4, false));
        HIPCHK(hipMemsetAsync(r->d_hits, 0, (r->v.n_reads ? r->v.n_reads : 1) * 4, st));
    }
    // per-kernel timing: events attached to the dispatches of every `stride`-th pass (profiling-enabled dispatches
    // cost a little command-processor work each); the whole loop is bracketed by its own pair of events
    int stride = steps <= 8 ? 1 : (int)env_u32("MF_EVENT_STRIDE", 8);
    if (stride < 1) stride = 1;
    const int n_sampled = stride > steps ? 0 : (steps + stride - 1) / stride;       // (a stride beyond the call: no per-kernel events at all)
    EventList events;
    HIPCHK(events.create((size_t)n_sampled * 6));
    hipEvent_t *ev = events.ev.data();
    for (int i = 0; i < 2; i++) if (!r->ev_call[i]) HIPCHK(hipEventCreate(&r->ev_call[i]));
    const hipEvent_t e_begin = r->ev_call[0], e_end = r->ev_call[1];
    lap("events");
    // (one pass: its three launches follow one another on the one stream -- nothing to overlap with, and no second stream to make)
    const bool pipelined = steps > 1;
    if (pipelined) { rc = ensure_pipeline_streams(ctx); if (rc) return rc; }
    HIPCHK(hipEventRecord(e_begin, st));
    if (pipelined) HIPCHK(hipStreamWaitEvent(ctx->stream3, e_begin, 0));
    // every pass's own tally block when the caller wants them all (tests: a buffer-set race that corrupted only the middle passes
    // of a pipelined call would not show in the last pass's tally)
    constexpr size_t TALLY_WORDS = 3 * (size_t)EXACT_MAX_GRID * 2;
    struct PinnedTmp { unsigned long long *p = nullptr; ~PinnedTmp() { if (p) (void)hipHostFree(p); } } all_tallies;
    if (pass_per_step) {
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&all_tallies.p), (size_t)steps * TALLY_WORDS * 8, hipHostMallocDefault));
        memset(all_tallies.p, 0, (size_t)steps * TALLY_WORDS * 8);
    }
    struct OverrideReset { mf_reads *r; ~OverrideReset() { r->tally_override = nullptr; } } override_reset{r};
    for (int i = 0; i < steps; i++) {
        if (pass_per_step) r->tally_override = all_tallies.p + (size_t)i * TALLY_WORDS;
        rc = enqueue_pass(r, T->view, thr, mode, count_all, ctx, n_sampled && i % stride == 0 ? &ev[(size_t)(i / stride) * 6] : nullptr, pipelined, i + 1 < steps);
        if (rc) return rc;
    }
    lap("enqueued");
    // join: finish kernels still running on the second stream belong to this call (unrecorded events are no-ops)
    for (int i = 0; i < NSETS; i++) HIPCHK(hipStreamWaitEvent(st, r->ev_finish[i], 0));
    HIPCHK(hipEventRecord(e_end, st));
    const bool two_halves = r->sample_pass;
    const unsigned long long *part = pass_per_step ? all_tallies.p + (size_t)(steps - 1) * TALLY_WORDS : r->d_counters[r->cur];          // pinned host memory, complete once the stream is
    if (out_bits) HIPCHK(hipMemcpyAsync(out_bits, r->d_bits[r->cur], ((r->v.n_reads + 31) / 32) * 4, hipMemcpyDeviceToHost, st));
    if (hits_out && r->v.n_reads) HIPCHK(hipMemcpyAsync(hits_out, r->d_hits, r->v.n_reads * 4, hipMemcpyDeviceToHost, st));
    lap("copies");
    HIPCHK(hipStreamSynchronize(st));
    lap("synced");
    unsigned long long cnt[2] = {0, 0};
    for (int i = 0; i < (two_halves ? 3 : 1) * EXACT_MAX_GRID; i++) { cnt[0] += part[2 * i]; cnt[1] += part[2 * i + 1]; }          // (a screen + finish pass: phase 0, phase 1, the exact kernel behind them)
    if (pass_per_step)
        for (int s = 0; s < steps; s++) {
            const unsigned long long *q = all_tallies.p + (size_t)s * TALLY_WORDS;
            uint64_t n = 0;
            for (int i = 0; i < (two_halves ? 3 : 1) * EXACT_MAX_GRID; i++) n += q[2 * i];
            pass_per_step[s] = n;
        }
    const bool adapt = g_opt.adapt != 0;          // (adapt = 0: measurements of the sample pass on bait-rich input)
    if (adapt && !T->view.prot && mode == MF_MODE_SCREENED && T->view.s > 0 && r->v.n_reads >= 100000) {
        // work items per read: ~0.025 at 0.5 % bait reads, 0.4 at 10 %, 0.8 at 20 %.  (Since a run start is left to the first lane that
        // holds one, the two kinds of pass are within 5 % of each other from 2 % to 100 % bait reads; the switch stays for inputs
        // that are nearly all bait.)
        if (r->sample_pass) { if (cnt[1] > r->v.n_reads) r->prefer_split = true; r->finish_two = cnt[1] > r->v.n_reads / 20; }
        else {                                                                                     // candidate reads per read
            if (r->prefer_split && cnt[1] < r->v.n_reads / 8) r->prefer_split = false;
            if (cnt[1] > r->v.n_reads / 20) r->split_serial = true; else if (cnt[1] < r->v.n_reads / 40) r->split_serial = false;
        }
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        float tot = 0, scr = 0, mrk = 0, exa = 0, t;
        HIPCHK(hipEventElapsedTime(&tot, e_begin, e_end));
        const bool prot = T->view.prot != 0, screened = !prot && mode == MF_MODE_SCREENED && T->view.s > 0;
        for (int i = 0; i < n_sampled; i++) {
            hipEvent_t *e = &ev[(size_t)i * 6];
            // a kernel that had nothing to do was not launched and its events were never recorded: that reads as zero
            auto span = [&](hipEvent_t a, hipEvent_t b) { t = 0; if (hipEventElapsedTime(&t, a, b) != hipSuccess) { (void)hipGetLastError(); t = 0; } return t; };
            if (screened) { scr += span(e[0], e[1]); mrk += span(e[2], e[3]); }
            exa += span(e[4], e[5]);
        }
        stats->n_reads = r->v.n_reads; stats->n_pass = cnt[0];
        stats->n_candidates = (mode == MF_MODE_SCREENED && T->view.s > 0) ? cnt[1] : r->v.n_reads;
        stats->ms_total = tot / steps;
        if (n_sampled) { stats->ms_screen = scr / n_sampled; stats->ms_mark = mrk / n_sampled; stats->ms_exact = exa / n_sampled; }
        stats->algorithmic_bytes = algorithmic_bytes(r->v);
    }
    lap("stats");
    return MF_OK;
}

extern "C" {

int mf_filter(const mf_kmerset *ks, const mf_reads *reads, uint32_t threshold, int mode,
              uint32_t *out_bits, uint32_t *hits_out, mf_filter_stats_t *stats)
{
    if (!out_bits && !hits_out && !stats) return fail(MF_E_ARG, "nothing to return: out_bits, hits_out and stats are all NULL");
    return filter_common(ks, reads, threshold, mode, out_bits, hits_out, 1, stats);
}

int mf_filter_resident(const mf_kmerset *ks, const mf_reads *reads, uint32_t threshold, int mode, int steps,
                       mf_filter_stats_t *stats)
{
    return filter_common(ks, reads, threshold, mode, nullptr, nullptr, steps, stats);
}

int mf_filter_resident_passes(const mf_kmerset *ks, const mf_reads *reads, uint32_t threshold, int mode, int steps,
                              uint64_t *n_pass_per_step, mf_filter_stats_t *stats)
{
    if (!n_pass_per_step) return fail(MF_E_ARG, "n_pass_per_step is NULL");
    return filter_common(ks, reads, threshold, mode, nullptr, nullptr, steps, stats, n_pass_per_step);
}

int mf_filter_packed(const mf_kmerset *ks, int device, const uint32_t *words, const uint64_t *offsets, uint64_t n_reads,
                     const uint64_t *npos, uint64_t n_npos, uint32_t threshold, uint32_t *out_bits)
{
    if (!out_bits) return fail(MF_E_ARG, "out_bits is NULL");
    mf_reads *r = nullptr;
    int rc = mf_reads_from_packed(words, offsets, n_reads, npos, n_npos, device, &r);
    if (rc) return rc;
    rc = filter_common(ks, r, threshold, MF_MODE_SCREENED, out_bits, nullptr, 1, nullptr);
    reads_release(r);
    return rc;
}

// ------------------------------------------------------------- file level
// what the calling thread's last file-level call did (mf_last_ingest_stats)
static thread_local mf_ingest_stats_t t_ingest_stats;
static thread_local bool t_ingest_stats_valid = false;

// the file-level call on a list of (logical) devices
static int filter_fastq_files_on(mf_kmerset *ks, const char *fq1, const char *fq2, const char *out1, const char *out2,
                                 uint32_t threshold, int pair_mode, const int *devices, int n_devices, uint64_t *kept, uint64_t *total)
{
    if (!ks || !fq1 || !out1) return fail(MF_E_ARG, "NULL argument");
    if ((fq2 == nullptr) != (out2 == nullptr)) return fail(MF_E_ARG, "fq2 and out2 must be given together");
    if (pair_mode != MF_PAIR_EITHER && pair_mode != MF_PAIR_BOTH) return fail(MF_E_ARG, "bad pair_mode %d", pair_mode);
    if (threshold < 1) return fail(MF_E_ARG, "threshold must be >= 1");
    const int have = mf_device_count();
    if (have <= 0) return fail(MF_E_NO_DEVICE, "no HIP device visible; libmitofilter_hip has no CPU fallback");
    if (!devices || n_devices < 1) return fail(MF_E_ARG, "empty device list");
    for (int i = 0; i < n_devices; i++) {
        if (devices[i] < 0 || devices[i] >= have) return fail(MF_E_ARG, "device %d out of range (have %d)", devices[i], have);
        for (int j = 0; j < i; j++) if (devices[j] == devices[i]) return fail(MF_E_ARG, "device %d listed twice", devices[i]);
    }

    // A regular .gz file: the bytes go to the GPU(s) as they are and inflate, line indexing, packing, filter and the copy of the
    // survivors run there (mf_devingest.cpp, streaming: device memory stays bounded whatever the file's size) -- the host's inflate is
    // what bounds the host pipeline on compressed input.  Plain files stay with the host pipeline by default (it packs them to a
    // quarter of their size before the copy to the device, and at ~45 GB/s of text); MF_INGEST=device sends them the same way,
    // MF_INGEST=host keeps everything on the host pipeline.  Pipes and BGZF files always take the host pipeline.
    {
        const char *ing = getenv("MF_INGEST");
        const bool force = ing && strcmp(ing, "device") == 0, any_gz = has_gz_ext(fq1) || (fq2 && has_gz_ext(fq2));
        // Plain regular files take the device path too (round 5): their bytes go up as they lie and are cut, packed and filtered there -- ahead of the
        // host pipeline at every size measured, 0.16 GB (7 ms against 20) to 10.7 GB (profiles/r05/e_masks_ab.txt, bench: extra.e2e_files.configs4_se_plain /
        // configs1_pe_plain).  MF_INGEST_PLAIN_MIN_MB sets a size below which plain files keep the host pipeline (default 0).
        bool big_plain = false;
        if (!any_gz && fq1) {
            struct stat sb; uint64_t sum = 0; bool regular = true;
            for (const char *f : {fq1, fq2}) if (f) { if (stat(f, &sb) == 0 && S_ISREG(sb.st_mode)) sum += (uint64_t)sb.st_size; else regular = false; }
            const char *mn = getenv("MF_INGEST_PLAIN_MIN_MB");
            big_plain = regular && sum > 0 && sum >= (mn && *mn ? strtoull(mn, nullptr, 10) : 0) << 20;
        }
        if (!(ing && strcmp(ing, "host") == 0) && (force || any_gz || big_plain)) {
            std::string derr; IngestStats is;
            const int drc = run_device_ingest(ks, fq1, fq2, out1, out2, threshold, pair_mode == MF_PAIR_BOTH, devices, n_devices, kept, total, derr, &is);
            if (drc == MF_OK) {
                mf_ingest_stats_t &o = t_ingest_stats;
                memset(&o, 0, sizeof o);
                o.path = MF_INGEST_PATH_DEVICE; o.n_devices = is.n_devices; o.consumers = is.consumers;
                o.input_bytes = is.input_bytes; o.text_bytes = is.text_bytes; o.records = is.records;
                o.seconds = is.seconds; o.decode_busy_seconds = is.decode_busy_seconds;
                o.pool_bytes_peak = is.pool_bytes_peak; o.device_bytes_peak = is.device_bytes_peak;
                o.chunks = is.chunks; o.chunks_linked = is.chunks_linked; o.gaps = is.gaps; o.gap_bytes = is.gap_bytes;
                t_ingest_stats_valid = true;
                return MF_OK;
            }
            if (drc != MF_DEVINGEST_DECLINED) return fail(drc, "%s", derr.c_str());
            if (getenv("MF_PIPE_TIMING")) fprintf(stderr, "[mf device ingest] declined%s%s: the host pipeline takes the input\n", derr.empty() ? "" : ": ", derr.c_str());
        }
    }
    // The host pipeline deals its batches to logical devices 0 .. n - 1 of a contiguous range: it is given the list's first device
    // as its base when the list is a run of consecutive devices, which is what the callers of this library pass.
    for (int i = 1; i < n_devices; i++)
        if (devices[i] != devices[0] + i) return fail(MF_E_ARG, "the host pipeline (plain files, pipes, BGZF) takes a run of consecutive devices; got %d after %d", devices[i], devices[i - 1]);
    const int dev_base = devices[0];
    const int hw = (int)std::thread::hardware_concurrency();
    // pack threads: what is left after the readers, writers and device workers
    int pack_threads = hw - 4 - n_devices; if (pack_threads < 1) pack_threads = 1; if (pack_threads > 64) pack_threads = 64;
    pack_threads = (int)env_u32("MF_PACK_THREADS", (uint32_t)pack_threads);
    if (pack_threads < 1) pack_threads = 1;
    const uint64_t batch_reads = env_u32("MF_BATCH_READS", 2000000);
    // Host-to-device overlap (configs[4]: "host-decompress overlapped with H2D on a side HIP stream").  The packers write
    // straight into pinned memory (hipHostMalloc through mf_host's DMA allocator), so a batch goes up as one asynchronous DMA;
    // and every device is served by TWO workers, each with its own stream pair and refillable device-side read set
    // (steady-state batches do not touch the allocator): while one worker's batch is in its kernels and read-back, the
    // other's is being copied up on its own stream -- and the readers, inflaters and packers of later batches run all along.
    // (for the time of this call only: other users of the host code in this process keep ordinary memory)
    set_dma_allocator([](size_t bytes) -> void * { void *p = nullptr; return hipHostMalloc(&p, bytes, hipHostMallocPortable) == hipSuccess ? p : nullptr; },
                      [](void *p) { (void)hipHostFree(p); });
    struct DmaScope { ~DmaScope() { set_dma_allocator(nullptr, nullptr); } } dma_scope;
    const int lanes = (int)env_u32("MF_WORKERS_PER_DEVICE", 2) < 1 ? 1 : (int)env_u32("MF_WORKERS_PER_DEVICE", 2);
    const int n_workers = n_devices * lanes;
    std::vector<mf_reads *> arena((size_t)n_workers, nullptr);
    BatchFilterFn fn = [ks, threshold, n_devices, dev_base, &arena](int worker, const PackedHost &P, uint64_t n, std::vector<uint32_t> &bits, std::string &err) -> int {
        bits.assign((n + 31) / 32 + 1, 0);
        if (n == 0) return MF_OK;
        const int device = dev_base + worker % n_devices, lane = worker / n_devices;
        DevCtx *ctx; int rc = get_ctx(device, &ctx, lane);
        if (rc == MF_OK && !arena[worker]) {
            arena[worker] = new (std::nothrow) mf_reads();
            if (!arena[worker]) rc = fail(MF_E_NOMEM, "out of memory"); else { arena[worker]->device = device; arena[worker]->lane = lane; }
        }
        if (rc == MF_OK)
            rc = reads_fill(arena[worker], true, P.words.data(), P.n_words, true, P.offsets.data(), n, P.offsets[n], P.uniform_len,
                            P.npos.data(), P.npos.size(), ctx);
        if (rc == MF_OK) rc = filter_common(ks, arena[worker], threshold, MF_MODE_SCREENED, bits.data(), nullptr, 1, nullptr);
        if (rc != MF_OK) err = t_err;
        return rc;
    };
    PipelineStats ps; std::string perr;
    const auto t_pipe0 = std::chrono::steady_clock::now();
    const int rc = run_fastq_pipeline(fq1, fq2, out1, out2, pair_mode == MF_PAIR_BOTH, n_workers, pack_threads, batch_reads, fn, ps, perr);
    {
        mf_ingest_stats_t &o = t_ingest_stats;
        memset(&o, 0, sizeof o);
        o.path = MF_INGEST_PATH_HOST; o.n_devices = n_devices; o.records = ps.total * (fq2 ? 2 : 1);
        o.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pipe0).count();
        t_ingest_stats_valid = rc == MF_OK;
    }
    {
        const auto t0 = std::chrono::steady_clock::now();
        for (mf_reads *a : arena) reads_release(a);
        if (getenv("MF_PIPE_TIMING"))
            fprintf(stderr, "[mf pipeline] device buffers released in %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    if (rc != MF_OK) return fail(rc, "%s", perr.c_str());
    if (kept) *kept = ps.kept;
    if (total) *total = ps.total;
    return MF_OK;
}

int mf_filter_fastq_files(mf_kmerset *ks, const char *fq1, const char *fq2, const char *out1, const char *out2,
                          uint32_t threshold, int pair_mode, int n_devices, uint64_t *kept, uint64_t *total)
{
    const int have = mf_device_count();
    if (have <= 0) return fail(MF_E_NO_DEVICE, "no HIP device visible; libmitofilter_hip has no CPU fallback");
    if (n_devices < 1) n_devices = 1;
    if (n_devices > have) n_devices = have;
    std::vector<int> devs((size_t)n_devices);
    for (int i = 0; i < n_devices; i++) devs[(size_t)i] = i;
    return filter_fastq_files_on(ks, fq1, fq2, out1, out2, threshold, pair_mode, devs.data(), n_devices, kept, total);
}

int mf_filter_fastq_files_on(mf_kmerset *ks, const char *fq1, const char *fq2, const char *out1, const char *out2,
                             uint32_t threshold, int pair_mode, const int *devices, int n_devices, uint64_t *kept, uint64_t *total)
{
    return filter_fastq_files_on(ks, fq1, fq2, out1, out2, threshold, pair_mode, devices, n_devices, kept, total);
}

int mf_set_option(const char *name, const char *value)
{
    options_from_env_once();          // (so that a later first pass does not overwrite what is set here)
    if (set_option(name, value) != 0) return fail(MF_E_ARG, "unknown option or value: %s=%s (options: pass=default|split|serial, adapt=0|1, finish_streams=0|1|2, screen_streams=1|2, split_pipe=0|1, exact_co=0|1, front=-1|0|1|2|3|4, canon=-1|0|1, s8_finish=-1|0|1, front2_log2b=0|6..24, front3_log2b=-1|0|6..27, expect_files=0|1, short_lived=0|1)", name ? name : "(null)", value ? value : "(null)");
    return MF_OK;
}

int mf_last_ingest_stats(mf_ingest_stats_t *out)
{
    if (!out) return fail(MF_E_ARG, "NULL argument");
    if (!t_ingest_stats_valid) return fail(MF_E_ARG, "no file-level call has succeeded on this thread");
    *out = t_ingest_stats;
    return MF_OK;
}

int mf_release_cached(uint64_t *bytes)
{
    const size_t n = mf::release_cached_device_memory(true);
    if (bytes) *bytes = n;
    return MF_OK;
}

// host-to-device copy rate of this box: `bytes` from pinned memory to device memory, the best of `reps` (the roof of the device ingest
// path, which sends the input file's bytes up as they are)
int mf_h2d_bandwidth(int device, size_t bytes, int reps, double *gb_per_s)
{
    if (!gb_per_s || !bytes || reps < 1) return fail(MF_E_ARG, "bad argument");
    DevCtx *ctx; int rc = get_ctx(device, &ctx); if (rc) return rc;
    void *h = nullptr, *d = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
    struct Free { void *&h, *&d; hipEvent_t &e0, &e1; ~Free() { if (h) (void)hipHostFree(h); if (d) (void)hipFree(d); if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); } } fr{h, d, e0, e1};
    HIPCHK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    memset(h, 0x5A, bytes);
    HIPCHK(dev_malloc(&d, bytes));
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    double best = 0;
    for (int i = 0; i <= reps; i++) {                     // (the first copy is a warm-up)
        HIPCHK(hipEventRecord(e0, ctx->stream));
        HIPCHK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipEventRecord(e1, ctx->stream));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        if (i && ms > 0) best = std::max(best, (double)bytes / (ms * 1e-3) / 1e9);
    }
    *gb_per_s = best;
    return MF_OK;
}

// ------------------------------------------------------------- quality filter
int mf_qualfilter_files(const char *fq1, const char *fq2, const char *out1, const char *out2,
                        uint64_t start, uint64_t end, uint64_t ns, uint32_t quality, float limit,
                        int dedup, uint64_t trim, int truncate_only, int device,
                        uint64_t *kept, uint64_t *total, int *panicked)
{
    if (!out1) return fail(MF_E_ARG, "out1 is NULL");
    if (start > end) return fail(MF_E_ARG, "start comes after end");
    if (quality == 0 || quality > 100) return fail(MF_E_ARG, "quality must be in 1..100");
    QualParams P; P.start = start; P.end = end; P.ns = ns; P.trim = trim; P.quality = quality; P.limit = limit;
    P.dedup = dedup != 0; P.trunc = truncate_only != 0;
    // Regular .gz files take the device ingest path (mf_devingest.cpp): the bytes go up as they lie, and inflate, line index, counting,
    // hashing, the de-duplication set, the decisions and the formatting of the kept records run there; what comes back is the output's
    // text.  Plain files stay with the host pipeline below by default, as in mf_filter_fastq_files (their text would cross PCIe twice, up as
    // input and down as output, where the host pipeline sends it up once and writes from its own memory: level on the box, the host
    // ahead on a few million pairs); MF_QUAL_INGEST=device sends them the same way, =host keeps everything on the host.  Standard input,
    // pipes, BGZF, empty files and .gz outputs always take the host pipeline.  Both write the same bytes: tests/test_filter_v2.py runs
    // the reference's vectors through both.
    {
        const char *ing = getenv("MF_QUAL_INGEST");
        const bool force = ing && strcmp(ing, "device") == 0, any_gz = (fq1 && has_gz_ext(fq1)) || (fq2 && has_gz_ext(fq2));
        if (!(ing && strcmp(ing, "host") == 0) && fq1 && (force || any_gz)) {
            if (mf_device_count() <= 0) return fail(MF_E_NO_DEVICE, "no HIP device visible; libmitofilter_hip has no CPU fallback");
            std::string derr; IngestStats is; bool pan = false;
            const int drc = run_device_qualfilter(fq1, fq2, out1, out2, P, device, kept, total, &pan, derr, &is);
            if (drc == MF_OK) {
                mf_ingest_stats_t &o = t_ingest_stats;
                memset(&o, 0, sizeof o);
                o.path = MF_INGEST_PATH_DEVICE; o.n_devices = is.n_devices; o.consumers = is.consumers;
                o.input_bytes = is.input_bytes; o.text_bytes = is.text_bytes; o.records = is.records;
                o.seconds = is.seconds; o.decode_busy_seconds = is.decode_busy_seconds;
                o.pool_bytes_peak = is.pool_bytes_peak; o.device_bytes_peak = is.device_bytes_peak;
                o.chunks = is.chunks; o.chunks_linked = is.chunks_linked; o.gaps = is.gaps; o.gap_bytes = is.gap_bytes;
                t_ingest_stats_valid = true;
                if (panicked) *panicked = pan ? 1 : 0;
                return MF_OK;
            }
            if (drc != MF_DEVINGEST_DECLINED) return fail(drc, "%s", derr.c_str());
            if (getenv("MF_PIPE_TIMING")) fprintf(stderr, "[mf device ingest] quality filter declined%s%s: the host pipeline takes the input\n", derr.empty() ? "" : ": ", derr.c_str());
        }
    }
    t_ingest_stats_valid = false;
    // The device is set up by the first batch that needs it (the decision thread), while the readers are already parsing: HIP's
    // start-up is a third of a second, as long as the whole pipeline takes for a few million records.  Without a gfx950 device the
    // call fails there with MF_E_NO_DEVICE (no CPU fallback) -- the output files have been created by then.
    DevCtx *ctx = nullptr; hipStream_t st = nullptr; int rc = MF_OK;
    auto ensure_device = [&](std::string &err) -> int {
        if (ctx) return MF_OK;
        const int r = get_ctx(device, &ctx);
        if (r) { ctx = nullptr; err = mf_last_error(); return r; }
        st = ctx->stream;
        return MF_OK;
    };
    struct Scratch { void *p = nullptr; size_t cap = 0; } d_text, d_recs, d_cnt, d_hash, d_flags;
    auto need = [&](Scratch &s, size_t bytes) -> hipError_t {
        if (bytes <= s.cap) return hipSuccess;
        if (s.p) hipFree(s.p);
        s.p = nullptr; s.cap = 0;
        hipError_t e = dev_malloc(&s.p, bytes + bytes / 4 + 4096);
        if (e == hipSuccess) s.cap = bytes + bytes / 4 + 4096;
        return e;
    };
#define QCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return MF_E_HIP; } } while (0)
    QualScanFn scan = [&](const char *text, size_t len, const QualSpan *recs, uint32_t n, uint32_t q, uint32_t *n_count,
                          uint32_t *bad_count, bool want_hashes, std::string &err) -> int {
        { const int r = ensure_device(err); if (r) return r; }
        QCHK(hipSetDevice(phys(device)));
        QCHK(need(d_text, len + 64)); QCHK(need(d_recs, (size_t)n * sizeof(QualSpan))); QCHK(need(d_cnt, (size_t)n * 8));
        if (want_hashes) QCHK(need(d_hash, (size_t)n * 8));
        QCHK(hipMemcpyAsync(d_text.p, text, len, hipMemcpyHostToDevice, st));
        QCHK(hipMemcpyAsync(d_recs.p, recs, (size_t)n * sizeof(QualSpan), hipMemcpyHostToDevice, st));
        uint32_t *dn = (uint32_t *)d_cnt.p, *db = dn + n;
        QCHK(launch_qualscan((const uint8_t *)d_text.p, (const QualRec *)d_recs.p, n, q, dn, db, want_hashes ? (uint64_t *)d_hash.p : nullptr, st));
        QCHK(hipMemcpyAsync(n_count, dn, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        QCHK(hipMemcpyAsync(bad_count, db, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        QCHK(hipStreamSynchronize(st));                   // (the hashes stay on the device for the dedup call)
        return MF_OK;
    };
    // The de-duplication set lives on the device for the whole file: keys + smallest file index per key.  Sized for the first batches
    // and doubled (rehashed by a kernel) before a batch that could fill it beyond a half.
    struct DedupSet { unsigned long long *keys = nullptr, *first = nullptr, *small = nullptr; uint64_t slots = 0, n_keys = 0, base = 0; } ds;     // small: [0] zero_idx, [1] n_keys
    auto ds_alloc = [&](uint64_t slots, unsigned long long **keys, unsigned long long **first, std::string &err) -> int {
        QCHK(dev_malloc((void **)keys, slots * 8)); QCHK(dev_malloc((void **)first, slots * 8));
        QCHK(hipMemsetAsync(*keys, 0, slots * 8, st)); QCHK(hipMemsetAsync(*first, 0xFF, slots * 8, st));
        return MF_OK;
    };
    QualDedupFn dedup_fn = [&](const uint8_t *alive, uint32_t n, uint8_t *dup, std::string &err) -> int {
        QCHK(hipSetDevice(phys(device)));
        if (!ds.keys) {
            uint64_t lg = env_u32("MF_DEDUP_LOG2_SLOTS", 24);
            if (lg < 4) lg = 4; if (lg > 34) lg = 34;
            ds.slots = (uint64_t)1 << lg;
            const int rc2 = ds_alloc(ds.slots, &ds.keys, &ds.first, err); if (rc2) return rc2;
            QCHK(dev_malloc((void **)&ds.small, 16));
            QCHK(hipMemsetAsync(ds.small, 0xFF, 8, st)); QCHK(hipMemsetAsync(ds.small + 1, 0, 8, st));
        }
        while (2 * (ds.n_keys + n) > ds.slots) {          // (every record of the batch may be a new key)
            unsigned long long *k2 = nullptr, *f2 = nullptr;
            const int rc2 = ds_alloc(ds.slots * 2, &k2, &f2, err); if (rc2) return rc2;
            QCHK(launch_dedup_rehash(ds.keys, ds.first, ds.slots, k2, f2, ds.slots * 2, st));
            QCHK(hipStreamSynchronize(st));
            QCHK(hipFree(ds.keys)); QCHK(hipFree(ds.first));
            ds.keys = k2; ds.first = f2; ds.slots *= 2;
        }
        QCHK(need(d_flags, (size_t)n * 2));
        uint8_t *d_alive = (uint8_t *)d_flags.p, *d_dup = d_alive + n;
        QCHK(hipMemcpyAsync(d_alive, alive, n, hipMemcpyHostToDevice, st));
        QCHK(launch_dedup((const uint64_t *)d_hash.p, d_alive, n, ds.base, ds.keys, ds.first, ds.slots, ds.small, ds.small + 1, d_dup, st));
        QCHK(hipMemcpyAsync(dup, d_dup, n, hipMemcpyDeviceToHost, st));
        unsigned long long nk = 0;
        QCHK(hipMemcpyAsync(&nk, ds.small + 1, 8, hipMemcpyDeviceToHost, st));
        QCHK(hipStreamSynchronize(st));
        ds.n_keys = nk; ds.base += n;
        return MF_OK;
    };
#undef QCHK
    static_assert(sizeof(QualSpan) == sizeof(QualRec), "host and device record layouts must match");
    int threads = (int)std::thread::hardware_concurrency() - 4; if (threads < 2) threads = 2; if (threads > 64) threads = 64;
    QualStats qs; std::string perr;
    rc = run_qualfilter_pipeline(fq1, fq2, out1, out2, P, threads, env_u32("MF_BATCH_READS", 2000000), scan, dedup_fn, qs, perr);
    if (ctx) { hipFree(d_text.p); hipFree(d_recs.p); hipFree(d_cnt.p); hipFree(d_hash.p); hipFree(d_flags.p); hipFree(ds.keys); hipFree(ds.first); hipFree(ds.small); }
    if (rc != MF_OK) return fail(rc, "%s", perr.c_str());
    if (kept) *kept = qs.kept;
    if (total) *total = qs.total;
    if (panicked) *panicked = qs.panicked ? 1 : 0;
    return MF_OK;
}

} // extern "C"
