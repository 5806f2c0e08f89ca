#include "mf_host.h"
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); std::vector<char> d; char b[1 << 16]; size_t n; while ((n = fread(b, 1, sizeof b, f)) > 0) d.insert(d.end(), b, b + n); fclose(f);
    for (int th : {1, 8, 16, 32, 64}) {
        auto t0 = std::chrono::steady_clock::now();
        mf::OutFile o; o.open("/tmp/of_out.fq.gz", th);
        for (size_t off = 0; off < d.size(); off += 4 << 20) o.write(d.data() + off, std::min<size_t>(4 << 20, d.size() - off));
        o.close();
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("threads %d: %.2f s  %.0f MB/s\n", th, dt, d.size() / dt / 1e6);
    }
}
