// gz_link_walk (mitoflex_amd/csrc/mf_gzdev.h) against a model that is obviously right: the acceptance walk over chunk descriptors -- which chunks
// of a slab are accepted, where their text goes, where the walk stops (a gap, a member's end, the end of the range) -- on random descriptor
// sequences built from a known ground truth (a chain of blocks; chunks that found a true boundary, a false one inside accepted data, nothing
// at all, or the member's end), walked in pieces as the product does (a slab at a time, the state carried).
//   g++ -O1 -std=c++17 -I tests/native/hipstub -I mitoflex_amd/csrc tests/native/linkwalk_check.cpp -o linkwalk_check && ./linkwalk_check [seeds]
#include "mf_gzdev.h"
#include <stdio.h>
#include <stdlib.h>
#include <random>
using namespace mf;

int main(int argc, char **argv)
{
    const int seeds = argc > 1 ? atoi(argv[1]) : 2000;
    for (int seed = 0; seed < seeds; seed++) {
        std::mt19937_64 rng((uint64_t)seed * 7919 + 1);
        auto rnd = [&](uint64_t n) { return (uint64_t)(rng() % n); };
        const uint32_t n = 1 + (uint32_t)rnd(60);
        const uint64_t chunk_bits = 8 * (1024 + rnd(4096));
        // ground truth: where accepted data ends after each accepted chunk.  Chunk c "covers" [c * chunk_bits, (c + 1) * chunk_bits); an honest
        // chunk starts at the first boundary at or behind its range's start and ends at the first boundary at or behind its range's end
        std::vector<GzChunk> ch(n);
        std::vector<int> kind(n);          // 0 honest, 1 nothing found, 2 false start inside accepted data, 3 honest + member end, 4 starts late (a gap in front)
        uint64_t bit = rnd(64);            // the member's first block
        const uint64_t first_bit = bit;
        for (uint32_t c = 0; c < n; c++) {
            GzChunk &d = ch[c];
            const uint64_t r0 = (uint64_t)c * chunk_bits, r1 = r0 + chunk_bits;
            const uint64_t k = rnd(100);
            kind[c] = k < 70 ? 0 : k < 78 ? 1 : k < 88 ? 2 : k < 92 ? 3 : 4;
            if (bit >= r1 && kind[c] != 1) kind[c] = 2;          // the chunk before ran past this one's whole range: whatever this one found lies inside accepted data
            d.n_sym = 1 + (uint32_t)rnd(100000);
            if (kind[c] == 1) { d.status = rnd(2) ? GZ_FAILED : GZ_NONE; d.start_bit = r0 + rnd(chunk_bits); d.end_bit = d.start_bit; d.n_sym = 8; continue; }
            if (kind[c] == 2) { d.status = GZ_AT_BOUNDARY; d.start_bit = bit > 0 ? rnd(bit) : 0; if (d.start_bit == bit) kind[c] = 0; d.end_bit = d.start_bit + 1 + rnd(chunk_bits); if (kind[c] == 2) continue; }
            const uint64_t start = kind[c] == 4 ? bit + 1 + rnd(500) : bit;
            d.start_bit = start; d.end_bit = (start > r1 ? start : r1) + rnd(3000); d.status = kind[c] == 3 ? GZ_MEMBER_END : GZ_AT_BOUNDARY;
            if (kind[c] == 4) continue;          // (a gap: the truth stays where it is until the host has bridged it -- the test bridges by jumping)
            bit = d.end_bit;
        }
        // the product's walk, a slab at a time
        GzLinkState st; st.cur_bit = first_bit;
        std::vector<uint32_t> acc, got; std::vector<uint64_t> off, got_off;
        // the model: one chunk at a time
        uint64_t m_bit = first_bit, m_total = 0; std::vector<uint32_t> want; std::vector<uint64_t> want_off;
        uint32_t lo = 0;
        while (lo < n) {
            const uint32_t step = 1 + (uint32_t)rnd(8), hi = lo + step < n ? lo + step : n;
            for (;;) {
                if (st.next < lo) st.next = lo;
                gz_link_walk(ch.data(), hi, st, acc, off);
                got.insert(got.end(), acc.begin(), acc.end()); got_off.insert(got_off.end(), off.begin(), off.end());
                if (st.stop == GZ_STOP_NONE) break;
                if (st.stop == GZ_STOP_GAP) { st.cur_bit = ch[st.next].start_bit; st.total += 17; continue; }          // the host bridges: 17 bytes, ends where the next chunk begins
                if (st.stop == GZ_STOP_MEMBER_END) { st.cur_bit += 64 + 18 * 8; st.wlen = 0; continue; }                  // trailer + next header: the next member's first block
            }
            lo = hi;
        }
        for (uint32_t c = 0; c < n; c++) {
            const GzChunk &d = ch[c];
            const bool ok = d.status == GZ_AT_BOUNDARY || d.status == GZ_MEMBER_END;
            if (!ok || d.start_bit < m_bit) continue;
            if (d.start_bit > m_bit) { m_bit = d.start_bit; m_total += 17; }          // the gap, bridged as above
            want.push_back(c); want_off.push_back(m_total);
            m_total += d.n_sym; m_bit = d.end_bit;
            if (d.status == GZ_MEMBER_END) m_bit += 64 + 18 * 8;
        }
        if (got != want || got_off != want_off || st.total != m_total || st.cur_bit != m_bit || st.linked != want.size() || st.linked + st.discarded != n) {
            fprintf(stderr, "FAILED seed %d: %zu chunks accepted (model %zu), total %llu (model %llu), cur_bit %llu (model %llu), linked %u discarded %u of %u\n", seed, got.size(), want.size(),
                    (unsigned long long)st.total, (unsigned long long)m_total, (unsigned long long)st.cur_bit, (unsigned long long)m_bit, st.linked, st.discarded, n);
            return 1;
        }
    }
    printf("gz_link_walk: %d random descriptor sequences equal to the model\n", seeds);
    return 0;
}
