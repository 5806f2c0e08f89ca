// usage: inflate_check FILE.gz EXPECTED_RAW CHUNK   -> prints "ok" when GzInflater(FILE.gz), read in pieces of CHUNK bytes, equals EXPECTED_RAW;
//        inflate_check FILE.gz - CHUNK              -> prints "error: <message>" or "ok <bytes>" (for damaged inputs)
//        inflate_check FILE.gz --time N            -> decode N times with 16 MiB pieces, print MB/s of output
#include "mf_inflate.h"
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
static std::vector<uint8_t> slurp(const char *p) { std::vector<uint8_t> v; FILE *f = fopen(p, "rb"); if (!f) { perror(p); exit(2); } uint8_t b[1 << 16]; size_t n; while ((n = fread(b, 1, sizeof b, f)) > 0) v.insert(v.end(), b, b + n); fclose(f); return v; }
int main(int argc, char **argv)
{
    if (argc != 4) return 2;
    std::vector<uint8_t> gz = slurp(argv[1]);
    if (!strcmp(argv[2], "--time")) {
        std::vector<uint8_t> buf((size_t)16 << 20);
        for (int it = 0; it < atoi(argv[3]); it++) {
            mf::GzInflater z; z.open(gz.data(), gz.size()); std::string err; size_t total = 0;
            auto t0 = std::chrono::steady_clock::now();
            for (;;) { long n = z.read(buf.data(), buf.size(), err); if (n < 0) { printf("error: %s\n", err.c_str()); return 1; } if (n == 0) break; total += (size_t)n; }
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("%.0f MB/s (%zu bytes in %.3f s)\n", total / dt / 1e6, total, dt);
        }
        return 0;
    }
    const size_t chunk = (size_t)atoll(argv[3]);
    mf::GzInflater z; z.open(gz.data(), gz.size());
    std::vector<uint8_t> out, buf(chunk); std::string err;
    for (;;) {
        long n = z.read(buf.data(), chunk, err);
        if (n < 0) { printf("error: %s\n", err.c_str()); return 0; }
        if (n == 0) { if (!z.eof()) { printf("error: zero bytes without eof\n"); return 0; } break; }
        out.insert(out.end(), buf.begin(), buf.begin() + n);
    }
    if (!strcmp(argv[2], "-")) { printf("ok %zu\n", out.size()); return 0; }
    std::vector<uint8_t> want = slurp(argv[2]);
    if (out == want) puts("ok"); else printf("MISMATCH got %zu want %zu\n", out.size(), want.size());
    return 0;
}
