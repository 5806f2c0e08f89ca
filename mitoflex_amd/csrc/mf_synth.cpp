#include "mf_synth.h"
#include "mf_common.h"

#include <thread>

namespace mf {

static inline uint64_t h64(uint64_t seed, uint64_t salt, uint64_t i)
{
    return mix64(seed ^ salt ^ ((i + 1) * 0x9E3779B97F4A7C15ULL));
}

static inline void set_base(uint32_t *words, uint64_t g, uint32_t code)
{
    uint32_t &w = words[g >> 4];
    const int sh = 2 * (int)(g & 15);
    w = (w & ~(3u << sh)) | (code << sh);
}
static inline uint32_t get_base(const uint32_t *words, uint64_t g) { return (words[g >> 4] >> (2 * (g & 15))) & 3u; }

bool synth_reads(uint64_t n_reads, uint32_t L, uint64_t seed, const BaitHost &bait,
                 uint32_t mito_ppm, uint32_t sub_ppm, uint32_t n_read_ppm, uint32_t n_base_ppm,
                 int threads, SynthOut &out, std::string &err, const SynthExtra &extra)
{
    out = SynthOut();
    const uint64_t total = n_reads * (uint64_t)L;
    out.n_words = (total + 15) / 16;
    out.words.assign(padded_words_for(out.n_words), 0u);
    uint32_t *W = out.words.data();

    // background
    if (threads < 1) threads = 1;
    const uint64_t n_pairs = (out.n_words + 1) / 2;
    auto fill = [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) {
            const uint64_t x = h64(seed, 0, i);
            W[2 * i] = (uint32_t)x;
            if (2 * i + 1 < out.n_words) W[2 * i + 1] = (uint32_t)(x >> 32);
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < threads; t++) th.emplace_back(fill, n_pairs * t / threads, n_pairs * (t + 1) / threads);
        fill(0, n_pairs / threads);
        for (auto &x : th) x.join();
    }
    // clear the bits past the last base of the last word
    if (total & 15) W[out.n_words - 1] &= (1u << (2 * (total & 15))) - 1;

    // bait records usable as read sources
    std::vector<uint64_t> rstart, rlen; uint64_t s0 = 0;
    for (uint64_t l : bait.rec_len) { if (l >= L) { rstart.push_back(s0); rlen.push_back(l); } s0 += l; }
    if ((mito_ppm || extra.numt_ppm) && rstart.empty()) { err = "no bait record is as long as read_len"; return false; }

    for (uint64_t r = 0; r < n_reads; r++) {
        const uint64_t g0 = r * (uint64_t)L;
        const bool numt = extra.numt_ppm && h64(seed, 0x5EED0010ULL, r) % 1000000ULL < extra.numt_ppm;
        const uint32_t sub_here = numt ? extra.numt_div_ppm : sub_ppm;
        if (extra.msat_ppm && !numt && h64(seed, 0x5EED0011ULL, r) % 1000000ULL < extra.msat_ppm) {
            const uint64_t hm = h64(seed, 0x5EED0012ULL, r);
            const uint32_t mlen = 1 + (uint32_t)(hm % 6);                     // motif of 1..6 bases, repeated
            for (uint32_t j = 0; j < L; j++) set_base(W, g0 + j, (uint32_t)(hm >> (8 + 2 * (j % mlen))) & 3u);
        } else
        if (numt || (mito_ppm && h64(seed, 0xA5A5A5A5ULL, r) % 1000000ULL < mito_ppm)) {
            const uint64_t hr = h64(seed, 0x5EED0001ULL, r);
            const size_t rec = (size_t)(hr % rstart.size());
            const uint64_t pos = (hr >> 20) % (rlen[rec] - L + 1);
            const bool rc = (hr >> 63) & 1;
            for (uint32_t j = 0; j < L; j++) {
                const uint64_t src = rstart[rec] + pos + (rc ? (L - 1 - j) : j);
                uint32_t c;
                if (bait.runlen[src] == 0) c = (uint32_t)(h64(seed, 0x5EED0002ULL, g0 + j) & 3);   // invalid bait base: random
                else { c = get_base(bait.words.data(), src); if (rc) c = 3 - c; }
                const uint64_t hs = h64(seed, 0x5EED0003ULL, g0 + j);
                if (sub_here && hs % 1000000ULL < sub_here) c = (c + 1 + (uint32_t)((hs >> 32) % 3)) & 3;
                set_base(W, g0 + j, c);
            }
            if (!numt) out.n_mito++;
        }
        if (n_read_ppm && h64(seed, 0x5EED0004ULL, r) % 1000000ULL < n_read_ppm) {
            for (uint32_t j = 0; j < L; j++) {
                if (h64(seed, 0x5EED0005ULL, g0 + j) % 1000000ULL < n_base_ppm) {
                    set_base(W, g0 + j, 0);
                    out.npos.push_back(g0 + j);
                }
            }
        }
    }
    return true;
}

} // namespace mf
