#include "mf_host.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
using namespace mf;
int main() {
    srand(5);
    for (int iter = 0; iter < 300; iter++) {
        int n = rand() % 200;
        std::vector<std::string> seqs(n);
        for (auto &s : seqs) {
            int L = (rand() % 4 == 0) ? (rand() % 3) * 16 : rand() % 70;
            if (iter % 3 == 0) L = 32;
            if (iter % 5 == 0) L = 150 + rand() % 3;
            if (iter % 7 == 0) L = rand() % 400;
            for (int i = 0; i < L; i++) s.push_back(iter % 4 == 1 && rand() % 8 == 0 ? (char)(rand() % 256) : "ACGTNacgtx"[rand() % 10]);
        }
        std::vector<FqRec> recs(n);
        for (int i = 0; i < n; i++) recs[i] = FqRec{"h", seqs[i].data(), seqs[i].data(), 1, (uint32_t)seqs[i].size(), (uint32_t)seqs[i].size()};
        PackedHost a, b;
        pack_records(recs.data(), n, 1, a);
        int th = 1 + rand() % 17;
        // poison the heap so that unwritten words show up
        { std::vector<uint32_t> junk(1 << 16, 0xDEADBEEF); }
        pack_records(recs.data(), n, th, b);
        if (a.words.size() != b.words.size() || memcmp(a.words.data(), b.words.data(), a.words.size() * 4) || a.offsets != b.offsets || a.npos != b.npos || a.uniform_len != b.uniform_len) { printf("MISMATCH iter %d n %d th %d\n", iter, n, th); return 1; }
        // reference: naive
        uint64_t g = 0; std::vector<uint32_t> w(a.words.size(), 0); std::vector<uint64_t> np;
        for (auto &s : seqs) for (char c : s) { int code = -1; switch (c) { case 'A': case 'a': code = 0; break; case 'C': case 'c': code = 1; break; case 'G': case 'g': code = 2; break; case 'T': case 't': code = 3; break; } if (code < 0) np.push_back(g); else w[g >> 4] |= (uint32_t)code << (2 * (g & 15)); g++; }
        if (memcmp(w.data(), a.words.data(), w.size() * 4) || np != a.npos) { printf("NAIVE MISMATCH iter %d\n", iter); return 1; }
    }
    puts("pack ok");
}
