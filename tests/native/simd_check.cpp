// usage: simd_check SEED   -> "ok" when mf::crc32_fast equals zlib's crc32 and mf::resolve_symbols equals the scalar definition
// on random buffers of every small length, unaligned starts, running CRCs and marker densities from none to all
#include "mf_pinflate.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <zlib.h>
int main(int argc, char **argv)
{
    srand(argc > 1 ? atoi(argv[1]) : 1);
    std::vector<uint8_t> b((1 << 20) + 128);
    for (auto &x : b) x = (uint8_t)rand();
    int bad = 0;
    for (int t = 0; t < 4000; t++) {
        const size_t off = rand() % 64, n = t < 600 ? (size_t)t : (size_t)rand() % 300000;
        const uint32_t c0 = t % 3 ? (uint32_t)rand() * 2654435761u : 0u;
        if (mf::crc32_fast(c0, b.data() + off, n) != (uint32_t)crc32(c0, b.data() + off, (uInt)n)) { if (bad++ < 5) printf("crc mismatch n=%zu off=%zu\n", n, off); }
    }
    std::vector<uint8_t> w(32768);
    for (auto &x : w) x = (uint8_t)rand();
    for (int t = 0; t < 600; t++) {
        const size_t n = t < 200 ? (size_t)t : (size_t)rand() % 5000;
        const int dens = t % 5;                       // 0: no marker .. 4: every symbol a marker
        std::vector<uint16_t> s(n + 1);
        for (size_t i = 0; i < n; i++) s[i] = (rand() % 4 < dens) ? (uint16_t)(0x8000u | (rand() & 0x7FFF)) : (uint16_t)(rand() & 0xFF);
        std::vector<uint8_t> got(n + 1, 0xEE), want(n + 1, 0xEE);
        for (size_t i = 0; i < n; i++) want[i] = (s[i] & 0x8000u) ? w[s[i] & 0x7FFF] : (uint8_t)s[i];
        mf::resolve_symbols(s.data(), n, w.data(), got.data());
        if (got != want) { if (bad++ < 5) printf("resolve mismatch n=%zu dens=%d\n", n, dens); }
    }
    printf(bad ? "FAILED %d\n" : "ok\n", bad);
    return bad != 0;
}
