// The block index over the offsets of a ragged read set (mitoflex_amd/csrc/mf_common.h: OffBlk, offblk_make -- what build_off_blk_kernel stores -- and
// offblk_lookup -- what the kernels' read_holding does for a ragged set) against a search over the offsets that is obviously right, on random read-length
// mixes: empty reads, reads of one to five bases (more than two read starts inside a block of 128 bases: the look-up's search path), ordinary 60 .. 150
// base reads, reads of several hundred and several thousand bases (the capped distances, the second load for an exact end), every base of the stream
// (or a sample of them) at several spans.
//   g++ -O1 -std=c++17 -I tests/native/hipstub -I mitoflex_amd/csrc tests/native/offblk_check.cpp -o offblk_check && ./offblk_check [seeds]
#include "mf_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <random>
#include <vector>
using namespace mf;

int main(int argc, char **argv)
{
    const int seeds = argc > 1 ? atoi(argv[1]) : 300;
    unsigned long long checked = 0, searched = 0, capped = 0;
    for (int seed = 0; seed < seeds; seed++) {
        std::mt19937_64 rng((uint64_t)seed * 104729 + 7);
        auto rnd = [&](uint64_t n) { return (uint64_t)(rng() % n); };
        const uint64_t n_reads = 1 + rnd(seed % 7 == 0 ? 40 : 3000);
        const int mix = seed % 5;          // 0 ordinary | 1 many tiny and empty reads | 2 long reads | 3 everything | 4 all empty but a few
        std::vector<uint64_t> off(n_reads + 1, 0);
        for (uint64_t i = 0; i < n_reads; i++) {
            uint64_t len;
            const uint64_t k = rnd(100);
            if (mix == 0) len = 60 + rnd(91);
            else if (mix == 1) len = k < 30 ? 0 : k < 80 ? 1 + rnd(5) : 20 + rnd(200);
            else if (mix == 2) len = k < 50 ? 300 + rnd(700) : k < 60 ? 2000 + rnd(9000) : 100 + rnd(60);
            else if (mix == 3) len = k < 10 ? 0 : k < 30 ? 1 + rnd(4) : k < 70 ? 60 + rnd(91) : k < 90 ? 200 + rnd(400) : 1000 + rnd(3000);
            else len = k < 95 ? 0 : 1 + rnd(300);
            off[i + 1] = off[i] + len;
        }
        const uint64_t total = off[n_reads];
        const uint64_t n_blk = (total >> OFF_BLK_SHIFT) + 2;
        std::vector<uint64_t> blk(n_blk);
        for (uint64_t b = 0; b < n_blk; b++) {
            const OffBlk e = offblk_make(off.data(), n_reads, b);
            blk[b] = e.pack();
            const OffBlk u = OffBlk::unpack(blk[b]);
            if (u.r0 != e.r0 || u.back != e.back || u.fwd != e.fwd || u.n != e.n || u.p1 != e.p1 || u.p2 != e.p2) { printf("seed %d block %llu: pack / unpack differ\n", seed, (unsigned long long)b); return 1; }
        }
        const uint64_t step = total > 200000 ? 1 + total / 150000 : 1;
        for (uint64_t g = 0; g < total; g += step)
            for (uint32_t s : {1u, 14u, 16u, 32u, 64u, 255u}) {
                if (g + s > total) continue;          // (read_holding's own first test)
                // the model: the last read that begins at or before g (of several beginning there -- empty reads -- the last, the one that holds bases)
                const uint64_t r = (uint64_t)(std::upper_bound(off.begin(), off.end(), g) - off.begin()) - 1;
                const uint64_t want = g + s <= off[r + 1] ? r : ~0ULL;
                uint64_t st = ~0ULL, en = ~0ULL;
                const uint64_t got = offblk_lookup(blk.data(), off.data(), n_reads, g, s, &st, &en);
                const uint64_t got2 = offblk_lookup(blk.data(), off.data(), n_reads, g, s, nullptr, nullptr);
                const uint64_t B0 = g >> OFF_BLK_SHIFT << OFF_BLK_SHIFT;
                bool ok = got == want && got2 == want && en == off[r + 1];
                // the start: exact, or -- a read that begins 255 bases and more in front of the block -- some base at least 255 in front of the block and not before the read
                if (B0 >= off[r] && B0 - off[r] >= 255) { ok = ok && st >= off[r] && st + 255 <= B0; capped++; }
                else ok = ok && st == off[r];
                if (!ok) {
                    printf("seed %d (mix %d) g %llu s %u: read %lld start %llu end %llu, model read %lld start %llu end %llu\n", seed, mix, (unsigned long long)g, s, (long long)got,
                           (unsigned long long)st, (unsigned long long)en, (long long)want, (unsigned long long)off[r], (unsigned long long)off[r + 1]);
                    return 1;
                }
                if (OffBlk::unpack(blk[g >> OFF_BLK_SHIFT]).n == 3) searched++;
                checked++;
            }
    }
    printf("%llu look-ups over %d read sets equal to the model (%llu through the search path, %llu with a capped start)\n", checked, seeds, searched, capped);
    return 0;
}
