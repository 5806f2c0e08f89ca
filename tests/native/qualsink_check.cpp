// TEST INFRASTRUCTURE: the host-side plumbing of the quality filter's device path (mitoflex_amd/csrc/mf_qualsink.h) without a GPU --
// per-record arrays filled and read by several threads, the chunk pool, the writer of an output file taking chunks out of order.
//   g++ -O1 -g -std=c++17 [-fsanitize=thread] -I mitoflex_amd/csrc tests/native/qualsink_check.cpp mitoflex_amd/csrc/mf_host.cpp -lz -lpthread
//   qualsink_check <dir>      prints "OK" and exits 0, or says what went wrong
#include "mf_qualsink.h"

#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <zlib.h>

using namespace mf;

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

static std::vector<uint8_t> pattern(size_t n) { std::vector<uint8_t> v(n); uint32_t x = 12345; for (size_t i = 0; i < n; i++) { x = x * 1664525u + 1013904223u; v[i] = (uint8_t)(x >> 24); } return v; }
static std::vector<uint8_t> slurp(const std::string &p) { std::vector<uint8_t> v; FILE *f = fopen(p.c_str(), "rb"); if (!f) return v; uint8_t b[65536]; size_t n; while ((n = fread(b, 1, sizeof b, f)) > 0) v.insert(v.end(), b, b + n); fclose(f); return v; }

// the pieces of `data` go to the sink from `threads` threads, each taking pieces in a scrambled order
static bool feed(QSink &sink, OutChunks &pool, const std::vector<uint8_t> &data, int threads)
{
    const size_t chunk = pool.chunk();
    const size_t n = (data.size() + chunk - 1) / chunk;
    std::atomic<size_t> next{0}; std::atomic<bool> ok{true};
    std::vector<size_t> order(n);
    for (size_t i = 0; i + 1 < n; i += 2) { order[i] = i + 1; order[i + 1] = i; }          // neighbours swapped: 1 0 3 2 ...
    if (n % 2) order[n - 1] = n - 1;
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&] {
            for (size_t k; (k = next++) < n;) {
                const size_t i = order[k], off = i * chunk, len = std::min(chunk, data.size() - off);
                if (!sink.wait_turn(off)) { ok = false; return; }
                bool no_mem = false;
                uint8_t *p = pool.take(&no_mem);
                if (!p) { ok = false; return; }
                memcpy(p, data.data() + off, len);
                sink.push(off, p, len);
            }
        });
    for (auto &x : th) x.join();
    return ok;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: qualsink_check <dir>\n"); return 2; }
    const std::string dir = argv[1];
    auto alloc = [](size_t n) -> void * { return malloc(n); };
    auto release = [](void *p) { free(p); };

    {   // ---- SegArray: ranges that cross segment borders, written by four threads, read back by four others
        SegArray<uint32_t> a;
        const uint64_t n = 5 * ((uint64_t)1 << 20) + 12345, piece = 700001;
        std::vector<std::thread> th;
        std::atomic<bool> ok{true};
        for (int t = 0; t < 4; t++)
            th.emplace_back([&, t] {
                std::vector<uint32_t> v;
                for (uint64_t r0 = (uint64_t)t * piece; r0 < n; r0 += 4 * piece) {
                    const uint64_t c = std::min(piece, n - r0);
                    v.resize(c);
                    for (uint64_t i = 0; i < c; i++) v[i] = (uint32_t)((r0 + i) * 2654435761u);
                    if (!a.put(r0, c, v.data())) ok = false;
                }
            });
        for (auto &x : th) x.join();
        th.clear();
        CHECK(ok);
        for (int t = 0; t < 4; t++)
            th.emplace_back([&, t] {
                std::vector<uint32_t> v;
                for (uint64_t r0 = (uint64_t)t * 333333; r0 < n; r0 += 4 * 333333) {
                    const uint64_t c = std::min<uint64_t>(333333, n - r0);
                    v.assign(c, 0);
                    a.get(r0, c, v.data());
                    for (uint64_t i = 0; i < c; i++) if (v[i] != (uint32_t)((r0 + i) * 2654435761u)) ok = false;
                }
            });
        for (auto &x : th) x.join();
        CHECK(ok);
        SegArray<uint8_t> b;
        CHECK(!b.put(((uint64_t)1 << 36) - 10, 20, (const uint8_t *)"01234567890123456789"));       // beyond what the table of segments holds: refused, not written
    }
    const std::vector<uint8_t> data = pattern(3 * 1000 * 1000 + 77);
    {   // ---- a regular file: chunks land at their offsets in whatever order they come
        OutChunks pool; pool.init(4096, 5, alloc, release);
        QSink sink;
        const std::string path = dir + "/direct.bin";
        CHECK(sink.open(path.c_str(), &pool));
        CHECK(feed(sink, pool, data, 3));
        CHECK(sink.close());
        CHECK(slurp(path) == data);
    }
    {   // ---- not a file to seek in (here: a .gz, compressed by OutFile): chunks are written in order however they arrive
        OutChunks pool; pool.init(65536, 2, alloc, release);          // (fewer chunks than threads: the chunks are taken in file order, or this hangs)
        QSink sink;
        const std::string path = dir + "/ordered.bin.gz";
        CHECK(sink.open(path.c_str(), &pool));
        CHECK(feed(sink, pool, data, 3));
        CHECK(sink.close());
        gzFile g = gzopen(path.c_str(), "rb");
        CHECK(g != nullptr);
        std::vector<uint8_t> back(data.size() + 10);
        const int got = gzread(g, back.data(), (unsigned)back.size());
        gzclose(g);
        CHECK(got == (int)data.size());
        back.resize(data.size());
        CHECK(back == data);
    }
    {   // ---- a run that is abandoned half way: nobody hangs, chunks come back
        OutChunks pool; pool.init(4096, 2, alloc, release);
        QSink sink;
        const std::string path = dir + "/gap.bin.gz";
        CHECK(sink.open(path.c_str(), &pool));
        bool no_mem = false;
        uint8_t *p = pool.take(&no_mem);
        CHECK(p && !no_mem);
        sink.push(4096, p, 100);                                      // (nothing at offset 0: an ordered sink waits for it)
        std::thread waiter([&] { bool f = false; uint8_t *q = pool.take(&f); if (q) { uint8_t *r = pool.take(&f); (void)r; } });      // the second take finds the pool empty
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
        pool.abort(); sink.abort();
        waiter.join();
        (void)sink.close();
    }
    {   // ---- no memory for a single chunk
        OutChunks pool; pool.init(4096, 3, [](size_t) -> void * { return nullptr; }, release);
        bool no_mem = false;
        CHECK(pool.take(&no_mem) == nullptr && no_mem);
    }
    puts("OK");
    return 0;
}
