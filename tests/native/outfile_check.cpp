// usage: outfile_check IN OUT THREADS   -> writes IN's bytes through mf::OutFile (gzip when OUT ends in .gz) in odd-sized pieces
#include "mf_host.h"
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
int main(int argc, char **argv)
{
    if (argc != 4) return 2;
    FILE *f = fopen(argv[1], "rb"); if (!f) return 2;
    std::vector<char> d; char b[1 << 16]; size_t n;
    while ((n = fread(b, 1, sizeof b, f)) > 0) d.insert(d.end(), b, b + n);
    fclose(f);
    const auto t0 = std::chrono::steady_clock::now();
    mf::OutFile o;
    if (!o.open(argv[2], atoi(argv[3]))) { puts("cannot open"); return 1; }
    size_t off = 0, piece = 1;
    while (off < d.size()) { const size_t k = std::min(piece, d.size() - off); if (!o.write(d.data() + off, k)) { puts("write failed"); return 1; } off += k; piece = piece * 3 + 1; if (piece > (5u << 20)) piece = 7; }
    if (!o.close()) { puts("close failed"); return 1; }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("ok %.0f MB/s\n", d.size() / dt / 1e6);
    return 0;
}
