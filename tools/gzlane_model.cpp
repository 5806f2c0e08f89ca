// CPU model of the lane-parallel block decoder of mitoflex_amd/csrc/mf_gzdev.hip: the SAME per-lane walk (mf_gzlane.h) run over 64
// emulated lanes with the kernel's protocol -- nominal starts, re-walks until every lane starts where its predecessor ended, the
// confirmed prefix, lists per lane -- and the expansion of the step's joined list, compared with zlib byte for byte: either a plain LZ77
// copy (cells = 0) or the kernel's rounds (cells = 1, expand4 of mf_gzdev.hip restated: 256 entries and at most 2 048 symbols a round,
// a 32-bit cell per output position -- a final symbol, a reference to a cell of the round, a reference to earlier output --, symbol t of
// a match at position first + t whatever the period, the references into the round followed by pointer doubling).  Test infrastructure:
// it validates the scheme (and prints how many walks a step takes) where there is no GPU.
//   g++ -O2 -std=c++17 tools/gzlane_model.cpp -lz -o /tmp/gzlane_model && /tmp/gzlane_model file.gz [span_bits=2048] [max_iter=6] [cells=0]
#include "../mitoflex_amd/csrc/mf_gzlane.h"
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <zlib.h>

using namespace mf::gzl;

struct Tables {
    uint32_t lit[LIT_SIZE], dist[DIST_SIZE];
    Canon cl, cd;
    uint16_t sorted_lit[288], sorted_dist[32];
};

// canonical table from code lengths, the kernel's build_table + pair_literals in serial form
static bool build(const uint8_t *lens, uint32_t n, int bits, int kind, uint32_t *tab, uint16_t *sorted, Canon &cn)
{
    uint32_t c[16] = {0};
    for (uint32_t s = 0; s < n; s++) c[lens[s]]++;
    c[0] = 0;
    uint32_t code = 0, o = 0, kraft = 0;
    cn.first[0] = cn.cnt[0] = cn.off[0] = 0;
    uint32_t fst[16], ofs[16];
    for (int l = 1; l < 16; l++) { code = (code + c[l - 1]) << 1; fst[l] = code; ofs[l] = o; kraft += c[l] << (15 - l); o += c[l]; cn.first[l] = (uint16_t)fst[l]; cn.cnt[l] = (uint16_t)c[l]; cn.off[l] = (uint16_t)ofs[l]; }
    if (kraft > 32768) return false;
    const uint32_t size = 1u << bits, invalid = kind == 1 ? (E_OTHER | K_INVALID) : (E_OTHER | D_INVALID);
    for (uint32_t i = 0; i < size; i++) tab[i] = invalid;
    uint32_t rank[16] = {0};
    for (uint32_t s = 0; s < n; s++) {
        const uint32_t l = lens[s];
        if (!l) continue;
        const uint32_t cd = fst[l] + rank[l];
        sorted[ofs[l] + rank[l]] = (uint16_t)s;
        rank[l]++;
        const uint32_t rev = brev32(cd) >> (32 - l);
        if (l <= (uint32_t)bits) { const uint32_t e = (kind == 1 ? lit_entry(s) : dist_entry(s)) | l; for (uint32_t i = rev; i < size; i += 1u << l) tab[i] = e; }
        else tab[rev & (size - 1)] = kind == 1 ? (E_OTHER | K_LONG) : E_OTHER;
    }
    if (kind == 1) {
        std::vector<uint32_t> ne(size);
        for (uint32_t i = 0; i < size; i++) {
            const uint32_t e1 = tab[i]; ne[i] = e1;
            if (!(e1 & E_OTHER)) {
                const uint32_t l1 = e1 & 255;
                if (l1 < (uint32_t)bits) {
                    const uint32_t e2 = tab[i >> l1];
                    if (!(e2 & E_OTHER) && l1 + (e2 & 255) <= (uint32_t)bits) ne[i] = (l1 + (e2 & 255)) | E_DOUBLE | (l1 << 25) | (e1 & 0xFF00u) | ((e2 & 0xFF00u) << 8);
                }
            }
        }
        memcpy(tab, ne.data(), size * 4);
    }
    return true;
}

struct Bits {
    const uint8_t *d; size_t n; uint64_t pos;
    uint32_t get(int k) { uint32_t v = 0; for (int i = 0; i < k; i++, pos++) v |= (uint32_t)((pos >> 3) < n ? (d[pos >> 3] >> (pos & 7)) & 1 : 0) << i; return v; }
};

struct In {
    const uint8_t *d; size_t n;
    uint32_t operator()(uint32_t w) const { uint32_t v = 0; const size_t b = (size_t)w * 4; for (int i = 0; i < 4; i++) if (b + i < n) v |= (uint32_t)d[b + i] << (8 * i); return v; }
};
struct Out { std::vector<uint32_t> *v; void put(uint32_t e) { v->push_back(e); } };

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s file.gz [span_bits] [max_iter]\n", argv[0]); return 2; }
    const uint32_t S = argc > 2 ? (uint32_t)atoi(argv[2]) : 2048, MAXIT = argc > 3 ? (uint32_t)atoi(argv[3]) : 6, LCAP = 1024;
    const bool cells = argc > 4 && atoi(argv[4]) != 0;
    uint64_t n_rounds = 0, n_passes = 0;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END); const size_t size = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> gz(size + 64, 0);
    if (fread(gz.data(), 1, size, f) != size) return 2;
    fclose(f);
    size_t p = 10; const unsigned flg = gz[3];
    if (flg & 4) p += 2 + (gz[p] | (gz[p + 1] << 8));
    if (flg & 8) { while (gz[p]) p++; p++; }
    if (flg & 16) { while (gz[p]) p++; p++; }
    if (flg & 2) p += 2;
    std::vector<uint8_t> ref;
    {
        z_stream z; memset(&z, 0, sizeof z);
        inflateInit2(&z, -15);
        z.next_in = gz.data() + p; z.avail_in = (uInt)(size - p);
        ref.resize(size * 4 + (1 << 20));
        int rc;
        for (;;) {
            if (ref.size() - z.total_out < (1u << 20)) ref.resize(ref.size() * 2);
            z.next_out = ref.data() + z.total_out; z.avail_out = (uInt)(ref.size() - z.total_out);
            rc = inflate(&z, Z_NO_FLUSH);
            if (rc != Z_OK) break;
        }
        if (rc != Z_STREAM_END) { fprintf(stderr, "zlib: %d\n", rc); return 2; }
        ref.resize(z.total_out);
        inflateEnd(&z);
    }
    static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    std::vector<uint8_t> out; out.reserve(ref.size());
    Bits br{gz.data(), size, (uint64_t)p * 8};
    In in{gz.data(), size}; (void)in;
    Tables *T = new Tables();
    uint64_t n_steps = 0, n_walks = 0, n_blocks = 0, lanes_used = 0, iter_hist[16] = {0}, codes = 0, unconverged = 0;
    for (;;) {
        const uint32_t final = br.get(1), type = br.get(2);
        if (type == 0) {
            br.pos = (br.pos + 7) & ~7ull;
            const uint32_t len = br.get(16); br.get(16);
            for (uint32_t i = 0; i < len; i++) out.push_back(gz[(br.pos >> 3) + i]);
            br.pos += (uint64_t)len * 8;
        } else {
            uint8_t lens[320]; uint32_t hlit = 288, hdist = 32;
            if (type == 1) { for (uint32_t i = 0; i < 320; i++) lens[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : (i < 288 ? 8 : 5))); }
            else {
                hlit = br.get(5) + 257; hdist = br.get(5) + 1; const uint32_t hclen = br.get(4) + 4;
                uint8_t pl[19] = {0};
                for (uint32_t i = 0; i < hclen; i++) pl[ORDER[i]] = (uint8_t)br.get(3);
                // precode, bit-serial canonical decode
                uint32_t cnt[8] = {0}, fst[8]; for (int i = 0; i < 19; i++) cnt[pl[i]]++; cnt[0] = 0;
                uint32_t code = 0; for (int l = 1; l < 8; l++) { code = (code + cnt[l - 1]) << 1; fst[l] = code; }
                auto sym = [&]() -> int {
                    uint32_t c = 0;
                    for (int l = 1; l < 8; l++) { c = (c << 1) | br.get(1); uint32_t k = fst[l]; for (int s = 0; s < 19; s++) if (pl[s] == l) { if (k == c) return s; k++; } }
                    return -1;
                };
                uint32_t i = 0, prev = 0;
                while (i < hlit + hdist) {
                    const int s = sym(); if (s < 0) { fprintf(stderr, "bad precode\n"); return 1; }
                    if (s < 16) { lens[i++] = (uint8_t)s; prev = (uint32_t)s; }
                    else { uint32_t rep, val = 0; if (s == 16) { rep = 3 + br.get(2); val = prev; } else if (s == 17) rep = 3 + br.get(3); else rep = 11 + br.get(7); while (rep--) lens[i++] = (uint8_t)val; prev = val; }
                }
            }
            if (!build(lens, hlit, LIT_BITS, 1, T->lit, T->sorted_lit, T->cl) || !build(lens + hlit, hdist, DIST_BITS, 2, T->dist, T->sorted_dist, T->cd)) { fprintf(stderr, "bad code\n"); return 1; }
            n_blocks++;
            // ---- the block's codes, 64 spans a step
            uint64_t B = br.pos;
            for (bool eob = false; !eob;) {
                n_steps++;
                // the reader's origin: a multiple of 32 bits at or in front of B (the kernel's: the chunk's)
                const uint64_t origin = B & ~31ull;
                In rin{gz.data() + (origin >> 3), size - (size_t)(origin >> 3)};
                uint32_t s[64], stop[64]; Span sp[64]; std::vector<uint32_t> lst[64];
                const uint32_t b0 = (uint32_t)(B - origin);
                for (int i = 0; i < 64; i++) { s[i] = b0 + (uint32_t)i * S; stop[i] = b0 + (uint32_t)(i + 1) * S; }
                auto walk = [&](int i) { lst[i].clear(); Out o{&lst[i]}; sp[i] = walk_span(T->lit, T->dist, T->cl, T->cd, T->sorted_lit, T->sorted_dist, rin, s[i], stop[i], LCAP, o); n_walks++; };
                for (int i = 0; i < 64; i++) walk(i);
                uint32_t it = 0;
                for (; it < MAXIT; it++) {
                    bool any = false; uint32_t ns[64]; bool need[64];
                    for (int i = 1; i < 64; i++) { const bool prev_ok = !(sp[i - 1].flags & (SP_EOB | SP_ERR)); need[i] = prev_ok && sp[i - 1].end != s[i]; ns[i] = sp[i - 1].end; any = any || need[i]; }
                    if (!any) break;
                    for (int i = 1; i < 64; i++) if (need[i]) { s[i] = ns[i]; walk(i); }
                }
                iter_hist[it < 15 ? it : 15]++;
                int V = 1;
                while (V < 64 && !(sp[V - 1].flags & (SP_EOB | SP_ERR)) && s[V] == sp[V - 1].end) V++;
                if (it == MAXIT) unconverged++;
                lanes_used += (uint64_t)V;
                if (cells) {
                    // the kernel's way: the confirmed lanes' lists joined, taken RN entries a round
                    constexpr uint32_t RN = 256, STG2 = RN * 8, C_REF = 1u << 16, C_FAR = 2u << 16;
                    std::vector<uint32_t> cl; uint64_t want = 0;
                    for (int i = 0; i < V; i++) {
                        if (sp[i].flags & SP_ERR) { fprintf(stderr, "decode error in a confirmed lane (step %llu lane %d)\n", (unsigned long long)n_steps, i); return 1; }
                        cl.insert(cl.end(), lst[i].begin(), lst[i].end()); want += sp[i].n_sym; codes += lst[i].size();
                    }
                    const size_t before = out.size();
                    for (size_t e0 = 0; e0 < cl.size();) {
                        const size_t n = std::min<size_t>(RN, cl.size() - e0);
                        // the entries that fit the round's cells (a prefix; at least one: no entry stands for more than 258 symbols)
                        uint32_t stg[STG2]; uint32_t tot = 0; size_t taken = 0;
                        const size_t opos = out.size();
                        for (; taken < n; taken++) {
                            const uint32_t e = cl[e0 + taken], cnt = (e >> 31) ? (e & 0x1FFu) + 3 : 1 + ((e >> 24) & 1u);
                            if (tot + cnt > STG2) break;
                            const uint32_t o = tot;
                            if (!(e >> 31)) { stg[o] = (e >> 8) & 0xFFu; if (cnt == 2) stg[o + 1] = (e >> 16) & 0xFFu; }
                            else {
                                const int32_t first = (int32_t)o - (int32_t)(((e >> 9) & 0x7FFFu) + 1);
                                for (uint32_t t = 0; t < cnt; t++) { const int32_t srel = first + (int32_t)t; stg[o + t] = srel >= 0 ? (C_REF | (uint32_t)srel) : (C_FAR | (uint32_t)(-srel - 1)); }
                            }
                            tot += cnt;
                        }
                        if (!taken) { fprintf(stderr, "a round without an entry\n"); return 1; }
                        for (uint32_t p = 0; p < tot; p++)          // earlier output, by position
                            if ((stg[p] >> 16) == 2u) {
                                const int64_t g = (int64_t)opos - 1 - (int64_t)(stg[p] & 0xFFFFu);
                                if (g < 0) { fprintf(stderr, "distance too far\n"); return 1; }
                                stg[p] = out[(size_t)g];
                            }
                        for (bool pending = true; pending;) {          // references into the round: strictly backwards, halved by every pass
                            pending = false; n_passes++;
                            uint32_t nx[STG2];
                            for (uint32_t p = 0; p < tot; p++) { const uint32_t v = stg[p]; nx[p] = v; if (v >> 16) { nx[p] = stg[v & 0xFFFFu]; if (nx[p] >> 16) pending = true; } }
                            memcpy(stg, nx, tot * 4);              // (all lanes of a pass read the cells as the pass found them -- or as a neighbour has already left them: either way the chain shrinks)
                        }
                        for (uint32_t p = 0; p < tot; p++) out.push_back((uint8_t)stg[p]);
                        e0 += taken; n_rounds++;
                    }
                    if (out.size() - before != want) { fprintf(stderr, "symbol count mismatch\n"); return 1; }
                }
                else for (int i = 0; i < V; i++) {
                    if (sp[i].flags & SP_ERR) { fprintf(stderr, "decode error in a confirmed lane (step %llu lane %d)\n", (unsigned long long)n_steps, i); return 1; }
                    if (lst[i].size() != sp[i].n_code) { fprintf(stderr, "list size mismatch\n"); return 1; }
                    size_t before = out.size();
                    for (uint32_t e : lst[i]) {
                        codes++;
                        if (!(e >> 31)) { out.push_back((uint8_t)(e >> 8)); if (e & E_DOUBLE) out.push_back((uint8_t)(e >> 16)); }
                        else { const uint32_t len = (e & 0x1FF) + 3, D = ((e >> 9) & 0x7FFF) + 1; if (D > out.size()) { fprintf(stderr, "distance too far\n"); return 1; } for (uint32_t k = 0; k < len; k++) out.push_back(out[out.size() - D]); }
                    }
                    if (out.size() - before != sp[i].n_sym) { fprintf(stderr, "symbol count mismatch\n"); return 1; }
                }
                B = origin + sp[V - 1].end;
                eob = (sp[V - 1].flags & SP_EOB) != 0;
            }
            br.pos = B;
        }
        if (final) break;
    }
    const bool ok = out.size() == ref.size() && memcmp(out.data(), ref.data(), out.size()) == 0;
    printf("%s: %zu -> %zu bytes, %llu blocks, %llu steps of 64 x %u bits, %.2f walks per lane and step, %.1f of 64 lanes confirmed per step, %llu codes, %llu steps not converged in %u rounds\n",
           argv[1], size, ref.size(), (unsigned long long)n_blocks, (unsigned long long)n_steps, S, (double)n_walks / (64.0 * n_steps), (double)lanes_used / n_steps,
           (unsigned long long)codes, (unsigned long long)unconverged, MAXIT);
    if (cells) printf("expansion: %llu rounds, %.2f pointer-doubling passes a round\n", (unsigned long long)n_rounds, n_rounds ? (double)n_passes / n_rounds : 0.0);
    printf("re-walk rounds per step:"); for (int i = 0; i < 16; i++) if (iter_hist[i]) printf(" %d:%llu", i, (unsigned long long)iter_hist[i]); printf("\n");
    printf(ok ? "PASS\n" : "FAIL\n");
    return ok ? 0 : 1;
}
