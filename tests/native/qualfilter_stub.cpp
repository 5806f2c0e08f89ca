// TEST INFRASTRUCTURE ONLY -- never shipped, never loaded by the product.
// Stands in for libmitofilter_hip.so's mf_qualfilter_files so that the HOST side of the quality filter (the
// drop-in CLI's argument handling, the readers, the sequential decision stage, the writers: everything except
// the GPU counting and hashing kernels) can be tested without a GPU and under sanitizers: the per-record counts
// the kernels deliver are computed here by the most obvious loops.  The GPU tests (tests/test_filter_v2.py,
// -m gpu) run the same vectors through the real library.
#include "../../include/mitofilter.h"
#include "mf_pipeline.h"
#include <string.h>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

static thread_local std::string t_err;

static inline uint64_t rotl(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }
static uint64_t siphash13(const unsigned char *p, size_t n)        // keys (0, 0): Rust's DefaultHasher
{
    uint64_t v0 = 0x736f6d6570736575ULL, v1 = 0x646f72616e646f6dULL, v2 = 0x6c7967656e657261ULL, v3 = 0x7465646279746573ULL;
    auto round = [&] { v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32); v2 += v3; v3 = rotl(v3, 16); v3 ^= v2;
                       v0 += v3; v3 = rotl(v3, 21); v3 ^= v0; v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32); };
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t m; memcpy(&m, p + i, 8); v3 ^= m; round(); v0 ^= m; }
    uint64_t b = (uint64_t)(n & 0xff) << 56;
    for (size_t j = 0; i + j < n; j++) b |= (uint64_t)p[i + j] << (8 * j);
    v3 ^= b; round(); v0 ^= b;
    v2 ^= 0xff; round(); round(); round();
    return v0 ^ v1 ^ v2 ^ v3;
}

extern "C" {

int mf_abi_version(void) { return MF_ABI_VERSION; }
const char *mf_last_error(void) { return t_err.c_str(); }

int mf_qualfilter_files(const char *fq1, const char *fq2, const char *out1, const char *out2, uint64_t start, uint64_t end, uint64_t ns,
                        uint32_t quality, float limit, int dedup, uint64_t trim, int truncate_only, int, uint64_t *kept, uint64_t *total,
                        int *panicked)
{
    if (!out1) { t_err = "out1 is NULL"; return MF_E_ARG; }
    if (start > end) { t_err = "start comes after end"; return MF_E_ARG; }
    if (quality == 0 || quality > 100) { t_err = "quality must be in 1..100"; return MF_E_ARG; }
    std::vector<uint64_t> last_hashes;                 // of the batch scanned last (the real library keeps them on the device)
    std::unordered_set<uint64_t> seen;                 // the reference's HashSet<u64>, filled in file order
    mf::QualScanFn scan = [&](const char *text, size_t, const mf::QualSpan *recs, uint32_t n, uint32_t q, uint32_t *n_count, uint32_t *bad_count,
                              bool want_hashes, std::string &) -> int {
        if (want_hashes) last_hashes.resize(n);
        for (uint32_t i = 0; i < n; i++) {
            const unsigned char *s = (const unsigned char *)text + recs[i].s_off, *qq = (const unsigned char *)text + recs[i].q_off;
            uint32_t nn = 0, nb = 0;
            for (uint32_t j = 0; j < recs[i].s_len; j++) nn += s[j] == 'N';
            for (uint32_t j = 0; j < recs[i].q_len; j++) nb += qq[j] <= q;
            n_count[i] = nn; bad_count[i] = nb;
            if (want_hashes) { std::string m((const char *)s, recs[i].s_len); m.push_back((char)0xff); last_hashes[i] = siphash13((const unsigned char *)m.data(), m.size()); }
        }
        return MF_OK;
    };
    mf::QualDedupFn dedup_fn = [&](const uint8_t *alive, uint32_t n, uint8_t *dup, std::string &) -> int {
        for (uint32_t i = 0; i < n; i++) dup[i] = alive[i] ? !seen.insert(last_hashes[i]).second : 0;      // main.rs:244-250, literally
        return MF_OK;
    };
    mf::QualParams P; P.start = start; P.end = end; P.ns = ns; P.trim = trim; P.quality = quality; P.limit = limit;
    P.dedup = dedup != 0; P.trunc = truncate_only != 0;
    mf::QualStats qs; std::string perr;
    const char *b = getenv("MF_BATCH_READS");
    const int rc = mf::run_qualfilter_pipeline(fq1, fq2, out1, out2, P, 6, b ? strtoull(b, nullptr, 10) : 2000000, scan, dedup_fn, qs, perr);
    if (rc != MF_OK) { t_err = perr; return rc; }
    if (kept) *kept = qs.kept;
    if (total) *total = qs.total;
    if (panicked) *panicked = qs.panicked ? 1 : 0;
    return MF_OK;
}

} // extern "C"
