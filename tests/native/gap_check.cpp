// TEST INFRASTRUCTURE: mf::inflate_gap (mf_pinflate.cpp) -- what the device decoder calls where its chunks do not link -- against zlib.
//   gap_check file.gz [seed]
// zlib inflates the (single-member) file in Z_BLOCK mode, which stops at every block boundary and says how many bits of the last
// byte it has not used: that gives (bit position, text offset) of every boundary.  For random pairs of boundaries inflate_gap has
// to produce exactly the text between them from the 32 KiB in front, stop at the wanted bit, and report the member's end.
#include "mf_pinflate.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <random>
#include <string>
#include <vector>
#include <zlib.h>

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: gap_check file.gz [seed]\n"); return 2; }
    FILE *f = fopen(argv[1], "rb"); if (!f) { perror(argv[1]); return 2; }
    std::vector<uint8_t> gz; { uint8_t buf[65536]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) gz.insert(gz.end(), buf, buf + n); } fclose(f);
    if (gz.size() < 18 || gz[0] != 0x1f || gz[1] != 0x8b || gz[3] != 0) { fprintf(stderr, "plain 10-byte gzip header expected\n"); return 2; }
    const size_t hdr = 10;
    z_stream zs; memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return 2;
    std::vector<uint8_t> text; std::vector<uint8_t> obuf(1 << 16);
    struct Bd { uint64_t bit; size_t off; bool last; };
    std::vector<Bd> bd; bd.push_back(Bd{(uint64_t)hdr * 8, 0, false});
    zs.next_in = gz.data() + hdr; zs.avail_in = (uInt)(gz.size() - hdr - 8);
    for (;;) {
        zs.next_out = obuf.data(); zs.avail_out = (uInt)obuf.size();
        const int rc = inflate(&zs, Z_BLOCK);
        text.insert(text.end(), obuf.data(), obuf.data() + (obuf.size() - zs.avail_out));
        if (rc != Z_OK && rc != Z_STREAM_END) { fprintf(stderr, "zlib: %d\n", rc); return 2; }
        if ((zs.data_type & 128) && rc != Z_STREAM_END && !(zs.data_type & 64)) {     // at a block boundary, not behind the last block
            const uint64_t bit = (uint64_t)(zs.next_in - gz.data()) * 8 - (uint64_t)(zs.data_type & 7);
            if (bd.back().bit != bit) bd.push_back(Bd{bit, text.size(), false});
        }
        if (rc == Z_STREAM_END) break;
    }
    inflateEnd(&zs);
    const uint64_t end_bits = (uint64_t)(gz.size() - 8) * 8;
    std::mt19937_64 rng(argc > 2 ? strtoull(argv[2], nullptr, 10) : 1);
    int bad = 0, runs = 0;
    auto check = [&](size_t i, size_t j, bool to_end) {
        const Bd &a = bd[i];
        uint8_t window[32768]; memset(window, 0xEE, sizeof window);
        const size_t wlen = a.off < 32768 ? a.off : 32768;
        memcpy(window + 32768 - wlen, text.data() + a.off - wlen, wlen);
        std::vector<uint8_t> out; uint64_t end_bit = 0; bool mend = false; std::string err;
        const uint64_t to_bit = to_end ? (uint64_t)gz.size() * 8 : bd[j].bit;
        const size_t want_off = to_end ? text.size() : bd[j].off;
        runs++;
        if (!mf::inflate_gap(gz.data(), gz.size(), a.bit, to_bit, window, wlen, out, end_bit, mend, err)) { printf("boundary %zu -> %zu: error %s\n", i, j, err.c_str()); bad++; return; }
        if (out.size() != want_off - a.off || memcmp(out.data(), text.data() + a.off, out.size()) != 0) { printf("boundary %zu -> %zu: %zu bytes, want %zu, or they differ\n", i, j, out.size(), want_off - a.off); bad++; return; }
        if (to_end ? (!mend || end_bit > end_bits || end_bit + 8 <= end_bits) : (mend || end_bit != to_bit)) { printf("boundary %zu -> %zu: stopped at bit %llu (member end %d), wanted %llu\n", i, j, (unsigned long long)end_bit, (int)mend, (unsigned long long)to_bit); bad++; }
    };
    const size_t nb = bd.size();
    for (int k = 0; k < 40 && nb >= 2; k++) { const size_t i = rng() % (nb - 1), j = i + 1 + rng() % std::min<size_t>(nb - 1 - i, 6); check(i, j, false); }
    for (int k = 0; k < 6; k++) check(nb - 1 - std::min<size_t>(nb - 1, (size_t)(rng() % 4)), 0, true);
    check(0, 0, true);                                   // the whole member
    if (nb >= 2) check(0, 0 + 0, false);                 // an empty gap: nothing to decode, stops where it starts
    printf("%zu block boundaries, %d gaps checked, %d wrong\n", nb, runs, bad);
    return bad ? 1 : 0;
}
