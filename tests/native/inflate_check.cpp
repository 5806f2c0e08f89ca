// usage: inflate_check FILE.gz EXPECTED_RAW CHUNK   -> prints "ok" when GzInflater(FILE.gz), read in pieces of CHUNK bytes, equals EXPECTED_RAW;
//        inflate_check FILE.gz - CHUNK              -> prints "error: <message>" or "ok <bytes>" (for damaged inputs)
//        inflate_check FILE.gz --time N            -> decode N times with 16 MiB pieces, print MB/s of output
#include "mf_inflate.h"
#include "mf_pinflate.h"
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
static std::vector<uint8_t> slurp(const char *p) { std::vector<uint8_t> v; FILE *f = fopen(p, "rb"); if (!f) { perror(p); exit(2); } uint8_t b[1 << 16]; size_t n; while ((n = fread(b, 1, sizeof b, f)) > 0) v.insert(v.end(), b, b + n); fclose(f); return v; }
// inflate_check FILE.gz EXPECTED_RAW|- CHUNK --parallel THREADS COMPRESSED_CHUNK   -> the parallel reader instead
// inflate_check FILE.gz --ptime N THREADS COMPRESSED_CHUNK                      -> timing of the parallel reader
int main(int argc, char **argv)
{
    if (argc != 4 && argc != 7 && argc != 6) return 2;
    std::vector<uint8_t> gz = slurp(argv[1]);
    if (argc == 6 && !strcmp(argv[2], "--ptime")) {
        std::vector<uint8_t> buf((size_t)16 << 20);
        for (int it = 0; it < atoi(argv[3]); it++) {
            mf::ParallelGzReader z; z.open(gz.data(), gz.size(), atoi(argv[4]), (size_t)atoll(argv[5])); std::string err; size_t total = 0;
            auto t0 = std::chrono::steady_clock::now();
            for (;;) { long n = z.read(buf.data(), buf.size(), err); if (n < 0) { printf("error: %s\n", err.c_str()); return 1; } if (n == 0) break; total += (size_t)n; }
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("%.0f MB/s (%zu bytes in %.3f s; linked %llu discarded %llu gap bytes %llu)\n", total / dt / 1e6, total, dt,
                   (unsigned long long)z.chunks_linked, (unsigned long long)z.chunks_discarded, (unsigned long long)z.gap_fill_bytes);
        }
        return 0;
    }
    if (argc == 7) {
        const size_t chunk = (size_t)atoll(argv[3]);
        mf::ParallelGzReader z; z.open(gz.data(), gz.size(), atoi(argv[5]), (size_t)atoll(argv[6]));
        std::vector<uint8_t> out, buf(chunk); std::string err;
        for (;;) {
            long n = z.read(buf.data(), chunk, err);
            if (n < 0) { printf("error: %s\n", err.c_str()); return 0; }
            if (n == 0) { if (!z.eof()) { printf("error: zero bytes without eof\n"); return 0; } break; }
            out.insert(out.end(), buf.begin(), buf.begin() + n);
        }
        if (!strcmp(argv[2], "-")) { printf("ok %zu\n", out.size()); return 0; }
        std::vector<uint8_t> want = slurp(argv[2]);
        if (out == want) printf("ok linked %llu discarded %llu gap %llu\n", (unsigned long long)z.chunks_linked, (unsigned long long)z.chunks_discarded, (unsigned long long)z.gap_fill_bytes);
        else printf("MISMATCH got %zu want %zu\n", out.size(), want.size());
        return 0;
    }
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
