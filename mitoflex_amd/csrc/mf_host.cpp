// Host-side parsing and packing for libmitofilter_hip.  See mf_host.h.
#include "mf_host.h"
#include <mutex>
#include <unordered_map>
#include "mf_kernels_cfg.h"

#include <algorithm>
#include <atomic>
#include <immintrin.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <zlib.h>

namespace mf {

// B1 alphabet: A/a=0 C/c=1 G/g=2 T/t=3, anything else 4 (invalid)
static const struct BaseLut {
    uint8_t t[256];
    BaseLut() {
        memset(t, 4, sizeof t);
        t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3;
    }
} g_lut;

bool has_gz_ext(const char *path)
{   // Path::extension() == "gz"  (filter/filter_bin/src/helper.rs:22)
    const char *slash = strrchr(path, '/');
    const char *name = slash ? slash + 1 : path;
    const char *dot = strrchr(name, '.');
    return dot && dot != name && strcmp(dot, ".gz") == 0;
}

bool slurp_file(const char *path, std::vector<char> &out, std::string &err)
{
    out.clear();
    if (has_gz_ext(path)) {
        gzFile g = gzopen(path, "rb");
        if (!g) { err = std::string("Cannot open file ") + path; return false; }
        gzbuffer(g, 1 << 20);
        size_t len = 0;
        out.resize(1 << 22);
        for (;;) {
            if (out.size() - len < (1 << 20)) out.resize(out.size() * 2);
            size_t want = out.size() - len; if (want > (1u << 30)) want = 1u << 30;
            int n = gzread(g, out.data() + len, (unsigned)want);
            if (n < 0) { gzclose(g); err = std::string("gzip read error in ") + path; return false; }
            if (n == 0) break;
            len += (size_t)n;
        }
        gzclose(g);
        out.resize(len);
        return true;
    }
    FILE *f = fopen(path, "rb");
    if (!f) { err = std::string("Cannot open file ") + path; return false; }
    if (fseeko(f, 0, SEEK_END) == 0) {
        off_t sz = ftello(f);
        if (sz > 0) out.reserve((size_t)sz);
        fseeko(f, 0, SEEK_SET);
    }
    size_t len = 0;
    out.resize(out.capacity() > (1 << 16) ? out.capacity() : (1 << 16));
    for (;;) {
        if (out.size() == len) out.resize(out.size() * 2);
        size_t n = fread(out.data() + len, 1, out.size() - len, f);
        if (n == 0) break;
        len += n;
    }
    fclose(f);
    out.resize(len);
    return true;
}

// ------------------------------------------------------------------- bait
uint64_t BaitHost::n_windows(int k) const
{
    uint64_t n = 0;
    for (uint64_t l : rec_len) if (l >= (uint64_t)k) n += l - k + 1;
    return n;
}
uint64_t BaitHost::n_swindows(int s) const { return n_windows(s); }

void parse_bait_fasta(const char *text, size_t len, BaitHost &out)
{
    out = BaitHost();
    std::vector<uint8_t> codes;            // 0..3 valid, 4 invalid, one per base
    std::vector<uint64_t> rec_start;       // base index where each record starts
    codes.reserve(len);
    bool at_line_start = true, in_header = false, open = false;
    for (size_t i = 0; i < len; i++) {
        const unsigned char c = (unsigned char)text[i];
        if (in_header) { if (c == '\n') { in_header = false; at_line_start = true; } continue; }
        if (at_line_start && c == '>') { rec_start.push_back(codes.size()); open = true; in_header = true; at_line_start = false; continue; }
        if (c == '\n') { at_line_start = true; continue; }
        at_line_start = false;
        if (c == '\r' || c == ' ' || c == '\t' || c == '\v' || c == '\f') continue;
        if (!open) { rec_start.push_back(codes.size()); open = true; }   // sequence before any header
        codes.push_back(g_lut.t[c]);
    }
    out.total = codes.size();
    rec_start.push_back(out.total);
    for (size_t r = 0; r + 1 < rec_start.size(); r++) out.rec_len.push_back(rec_start[r + 1] - rec_start[r]);
    out.words.assign((out.total + 15) / 16 + 8, 0u);
    out.runlen.assign(out.total + 1, 0);
    for (uint64_t g = 0; g < out.total; g++)
        if (codes[g] < 4) out.words[g >> 4] |= (uint32_t)codes[g] << (2 * (g & 15));
    // run lengths: walk each record backwards
    for (size_t r = 0; r + 1 < rec_start.size(); r++) {
        uint32_t run = 0;
        for (uint64_t g = rec_start[r + 1]; g-- > rec_start[r];) {
            run = codes[g] < 4 ? (run < 255 ? run + 1 : 255) : 0;
            out.runlen[g] = (uint8_t)run;
        }
    }
}

// ------------------------------------------------------------ protein bait
uint64_t ProtBaitHost::n_windows(int kp) const
{
    uint64_t n = 0;
    for (uint64_t L : rec_len) if (L >= (uint64_t)kp) n += L - kp + 1;
    return n;
}

static int residue_code(unsigned char c)
{
    static const char AA[] = "ACDEFGHIKLMNPQRSTVWY";
    if (c >= 'a' && c <= 'z') c = (unsigned char)(c - 'a' + 'A');
    for (int i = 0; i < 20; i++) if ((unsigned char)AA[i] == c) return i;
    return -1;
}

void parse_bait_protein(const char *text, size_t len, ProtBaitHost &out)
{
    out = ProtBaitHost();
    std::vector<int8_t> codes; codes.reserve(len);
    std::vector<uint64_t> rec_start;
    bool at_line_start = true, in_header = false, open = false;
    for (size_t i = 0; i < len; i++) {
        const unsigned char c = (unsigned char)text[i];
        if (in_header) { if (c == '\n') { in_header = false; at_line_start = true; } continue; }
        if (at_line_start && c == '>') { rec_start.push_back(codes.size()); open = true; in_header = true; at_line_start = false; continue; }
        if (c == '\n') { at_line_start = true; continue; }
        at_line_start = false;
        if (c == '\r' || c == ' ' || c == '\t' || c == '\v' || c == '\f') continue;
        if (!open) { rec_start.push_back(codes.size()); open = true; }
        codes.push_back((int8_t)residue_code(c));
    }
    out.total = codes.size();
    rec_start.push_back(out.total);
    for (size_t r = 0; r + 1 < rec_start.size(); r++) out.rec_len.push_back(rec_start[r + 1] - rec_start[r]);
    out.aa.assign(out.total + 16, 0);
    out.runlen.assign(out.total + 1, 0);
    for (uint64_t g = 0; g < out.total; g++) out.aa[g] = codes[g] < 0 ? 0 : (uint8_t)codes[g];
    for (size_t r = 0; r + 1 < rec_start.size(); r++) {
        uint32_t run = 0;
        for (uint64_t g = rec_start[r + 1]; g-- > rec_start[r];) {
            run = codes[g] >= 0 ? (run < 255 ? run + 1 : 255) : 0;
            out.runlen[g] = (uint8_t)run;
        }
    }
}

bool codon_lut_for(int genetic_code, int kp, uint32_t out[256])
{
    // NCBI transl_table strings, codons in the order TTT TTC TTA TTG TCT ... GGG (bases T, C, A, G)
    const char *tab = nullptr;
    switch (genetic_code) {
    case 1: case 11: tab = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"; break;
    case 2:  tab = "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSS**VVVVAAAADDEEGGGG"; break;
    case 3:  tab = "FFLLSSSSYY**CCWWTTTTPPPPHHQQRRRRIIMMTTTTNNKKSSRRVVVVAAAADDEEGGGG"; break;
    case 4:  tab = "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"; break;
    case 5:  tab = "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSSSVVVVAAAADDEEGGGG"; break;
    case 9:  tab = "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG"; break;
    case 13: tab = "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSGGVVVVAAAADDEEGGGG"; break;
    case 14: tab = "FFLLSSSSYYY*CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG"; break;
    case 21: tab = "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNNKSSSSVVVVAAAADDEEGGGG"; break;
    default: return false;
    }
    static const int tcag[4] = {2, 1, 3, 0};                       // our base codes A C G T -> position in "TCAG"
    auto residue = [&](int b1, int b2, int b3) { return residue_code((unsigned char)tab[16 * tcag[b1] + 4 * tcag[b2] + tcag[b3]]); };
    for (int c = 0; c < 64; c++) {
        const int b1 = c & 3, b2 = (c >> 2) & 3, b3 = (c >> 4) & 3;
        const int f = residue(b1, b2, b3), r = residue(3 - b3, 3 - b2, 3 - b1);
        const uint64_t top = (uint64_t)(f < 0 ? 31 : f) << (5 * (kp - 1));        // 31: the non-residue code (stop codon)
        out[4 * c] = (uint32_t)top; out[4 * c + 1] = (uint32_t)(top >> 32);
        out[4 * c + 2] = r < 0 ? 31u : (uint32_t)r;
        out[4 * c + 3] = (f >= 0 ? 1u : 0u) | (r >= 0 ? 2u : 0u);
    }
    return true;
}

// ------------------------------------------------------------------- FASTQ
void parse_fastq(const char *buf, size_t len, std::vector<FqRec> &recs)
{
    recs.clear();
    const char *p = buf, *end = buf + len;
    const char *ls[4]; uint32_t ll[4]; int li = 0;
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;            // final line may lack its LF
        size_t L = (size_t)(le - p);
        if (L && p[L - 1] == '\r') L--;            // lines() strips "\r\n"
        ls[li] = p; ll[li] = (uint32_t)L;
        if (++li == 4) { recs.push_back(FqRec{ls[0], ls[1], ls[3], ll[0], ll[1], ll[3]}); li = 0; }
        if (!nl) break;
        p = nl + 1;
    }
}

uint64_t padded_words_for(uint64_t n_words)
{
    const uint64_t chunk_vec = (uint64_t)SCREEN_BLOCK * SCREEN_U;
    uint64_t n_vec = (n_words + 3) / 4;
    n_vec = (n_vec + chunk_vec - 1) / chunk_vec * chunk_vec;
    return n_vec * 4 + 16;
}

uint32_t detect_uniform_len(const uint64_t *off, uint64_t n)
{
    if (n == 0) return 0;
    const uint64_t L = off[1] - off[0];
    if (L == 0 || L > 0xFFFFFFFFull) return 0;
    for (uint64_t i = 1; i < n; i++) if (off[i + 1] - off[i] != L) return 0;
    return (uint32_t)L;
}

// 32 sequence bytes -> 64 bits of 2-bit codes (invalid bytes code 0) + a bit mask of the invalid ones.
// Case is folded with & 0xDF; of a folded byte, bits 1-2 tell A, C, T, G apart as 0, 1, 2, 3, and
// x ^ (x >> 1) turns that into A, C, G, T = 0, 1, 2, 3.
#if defined(__x86_64__)
__attribute__((target("avx2,bmi2"))) static inline uint64_t codes32_avx2(const unsigned char *s, uint32_t &invalid)
{
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s));
    const __m256i f = _mm256_and_si256(v, _mm256_set1_epi8((char)0xDF));
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(f, _mm256_set1_epi8('A')), _mm256_cmpeq_epi8(f, _mm256_set1_epi8('C'))),
                                       _mm256_or_si256(_mm256_cmpeq_epi8(f, _mm256_set1_epi8('G')), _mm256_cmpeq_epi8(f, _mm256_set1_epi8('T'))));
    const __m256i x = _mm256_and_si256(_mm256_srli_epi16(f, 1), _mm256_set1_epi8(3));
    const __m256i c = _mm256_and_si256(_mm256_xor_si256(x, _mm256_and_si256(_mm256_srli_epi16(x, 1), _mm256_set1_epi8(1))), ok);
    invalid = ~(uint32_t)_mm256_movemask_epi8(ok);
    const uint64_t m = 0x0303030303030303ULL;
    return _pext_u64((uint64_t)_mm256_extract_epi64(c, 0), m) | (_pext_u64((uint64_t)_mm256_extract_epi64(c, 1), m) << 16) |
           (_pext_u64((uint64_t)_mm256_extract_epi64(c, 2), m) << 32) | (_pext_u64((uint64_t)_mm256_extract_epi64(c, 3), m) << 48);
}
#endif

// pack records [lo, hi).  The words that lie entirely inside the range are written plainly (and only
// once: nothing zero-fills the buffer first); the first and the last word may be shared with the
// neighbouring ranges -- those were zeroed by the caller and are OR-ed atomically.
// The same function body is compiled twice: plain, and for AVX2 + BMI2 (picked at run time).
#define MF_PACK_RANGE_BODY(SIMD_LOOP)                                                                            \
    if (lo >= hi || offsets[lo] == offsets[hi]) return;                                                          \
    uint64_t g = offsets[lo];                                                                                    \
    uint64_t wi = g >> 4;                                                                                        \
    const uint64_t first_w = wi, last_w = (offsets[hi] - 1) >> 4;   /* may be shared with neighbours */          \
    unsigned __int128 acc = 0; int nb = 2 * (int)(g & 15);          /* bits of the current word already taken */ \
    auto flush = [&](uint64_t w, uint32_t v) {                                                                   \
        if (w == first_w || w == last_w) __atomic_fetch_or(&words[w], v, __ATOMIC_RELAXED);                      \
        else words[w] = v;                                                                                       \
    };                                                                                                           \
    auto emit = [&](uint64_t bits, int n_bases) {                    /* append n_bases (<= 32) codes */           \
        acc |= (unsigned __int128)bits << nb;                                                                    \
        nb += 2 * n_bases;                                                                                       \
        while (nb >= 32) { flush(wi++, (uint32_t)acc); acc >>= 32; nb -= 32; }                                   \
    };                                                                                                           \
    for (uint64_t r = lo; r < hi; r++) {                                                                         \
        const unsigned char *s = (const unsigned char *)recs[r].s;                                               \
        const uint32_t L = recs[r].sl;                                                                           \
        uint32_t i = 0;                                                                                          \
        SIMD_LOOP                                                                                                \
        while (i < L) {                                              /* the tail (or everything without AVX2) */ \
            const uint32_t n = L - i < 32 ? L - i : 32;                                                          \
            uint64_t bits = 0;                                                                                   \
            for (uint32_t j = 0; j < n; j++) {                                                                   \
                const uint32_t c = g_lut.t[s[i + j]];                                                            \
                if (c == 4) npos.push_back(g + i + j); else bits |= (uint64_t)c << (2 * j);                      \
            }                                                                                                    \
            emit(bits, (int)n);                                                                                  \
            i += n;                                                                                              \
        }                                                                                                        \
        g += L;                                                                                                  \
    }                                                                                                            \
    if (nb) flush(wi, (uint32_t)acc);

static void pack_range_plain(const FqRec *recs, uint64_t lo, uint64_t hi, const uint64_t *offsets, uint32_t *words,
                             std::vector<uint64_t> &npos)
{
    MF_PACK_RANGE_BODY()
}

#if defined(__x86_64__)
__attribute__((target("avx2,bmi2")))
static void pack_range_avx2(const FqRec *recs, uint64_t lo, uint64_t hi, const uint64_t *offsets, uint32_t *words,
                            std::vector<uint64_t> &npos)
{
    MF_PACK_RANGE_BODY(
        for (; i + 32 <= L; i += 32) {
            uint32_t bad;
            const uint64_t bits = codes32_avx2(s + i, bad);
            for (; bad; bad &= bad - 1) npos.push_back(g + i + (uint32_t)__builtin_ctz(bad));
            emit(bits, 32);
        })
}
#endif

static void pack_range(const FqRec *recs, uint64_t lo, uint64_t hi, const uint64_t *offsets, uint32_t *words,
                       std::vector<uint64_t> &npos)
{
#if defined(__x86_64__)
    static const bool simd = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && !getenv("MF_NO_SIMD");
    if (simd) { pack_range_avx2(recs, lo, hi, offsets, words, npos); return; }
#endif
    pack_range_plain(recs, lo, hi, offsets, words, npos);
}

void pack_records(const FqRec *recs, uint64_t count, int threads, PackedHost &out)
{
    out.npos.clear(); out.uniform_len = 0; out.n_words = 0;     // buffers keep their capacity (batches are recycled)
    out.offsets.resize(count + 1);
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > count) threads = count ? (int)count : 1;
    auto run = [&](auto fn) {                               // fn(t) on `threads` workers
        if (threads == 1) { fn(0); return; }
        std::vector<std::thread> th;
        for (int t = 1; t < threads; t++) th.emplace_back(fn, t);
        fn(0);
        for (auto &x : th) x.join();
    };
    auto lo_of = [&](int t) { return count * (uint64_t)t / (uint64_t)threads; };
    // offsets: per-range sums, then the running offsets of every range; the uniform-length test rides along
    std::vector<uint64_t> sum(threads + 1, 0);
    std::vector<uint8_t> uni(threads, 1);
    const uint32_t L0 = count ? recs[0].sl : 0;
    run([&](int t) {
        uint64_t s = 0; bool u = true;
        for (uint64_t i = lo_of(t), e = lo_of(t + 1); i < e; i++) { s += recs[i].sl; u = u && recs[i].sl == L0; }
        sum[t + 1] = s; uni[t] = u;
    });
    for (int t = 0; t < threads; t++) sum[t + 1] += sum[t];
    const uint64_t total = sum[threads];
    out.offsets[count] = total;
    out.n_words = (total + 15) / 16;
    const uint64_t padded = padded_words_for(out.n_words);
    out.words.resize(padded);                                // not zero-filled (DefaultInitAlloc)
    bool uniform = count > 0 && L0 > 0;
    for (int t = 0; t < threads; t++) uniform = uniform && uni[t];
    out.uniform_len = uniform ? L0 : 0;
    run([&](int t) {
        uint64_t g = sum[t];
        for (uint64_t i = lo_of(t), e = lo_of(t + 1); i < e; i++) { out.offsets[i] = g; g += recs[i].sl; }
    });
    // words shared by two ranges (and the padding tail) must start from zero
    uint32_t *W = out.words.data();
    for (int t = 0; t <= threads; t++) {
        const uint64_t g = out.offsets[t == threads ? count : lo_of(t)];      // a range boundary, in bases
        W[g >> 4] = 0;                                       // first word of the range that starts here
        if (g) W[(g - 1) >> 4] = 0;                          // last word of the range that ends here
    }
    memset(W + out.n_words, 0, (padded - out.n_words) * sizeof(uint32_t));
    std::vector<std::vector<uint64_t>> np(threads);
    run([&](int t) { pack_range(recs, lo_of(t), lo_of(t + 1), out.offsets.data(), W, np[t]); });
    for (auto &v : np) out.npos.insert(out.npos.end(), v.begin(), v.end());   // ranges ascend with t
}

// ------------------------------------------------------------------ DMA-able blocks
static std::mutex g_dma_mu;
static void *(*g_dma_alloc)(size_t) = nullptr;
static void (*g_dma_release)(void *) = nullptr;
static std::unordered_map<void *, void (*)(void *)> g_dma_blocks;      // block -> how to release it (the allocator may have been taken away since)

void set_dma_allocator(void *(*alloc)(size_t), void (*release)(void *))
{
    std::lock_guard<std::mutex> lk(g_dma_mu);
    g_dma_alloc = alloc; g_dma_release = release;
}
void *dma_block_alloc(size_t bytes)
{
    void *(*alloc)(size_t); void (*release)(void *);
    { std::lock_guard<std::mutex> lk(g_dma_mu); alloc = g_dma_alloc; release = g_dma_release; }
    if (!alloc) return nullptr;
    void *p = alloc(bytes);                         // (pinning memory takes a while: not under the lock)
    if (p) { std::lock_guard<std::mutex> lk(g_dma_mu); g_dma_blocks.emplace(p, release); }
    return p;
}
bool dma_block_free(void *p)
{
    void (*release)(void *) = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_dma_mu);
        auto it = g_dma_blocks.find(p);
        if (it == g_dma_blocks.end()) return false;
        release = it->second;
        g_dma_blocks.erase(it);
    }
    if (release) release(p);
    return true;
}

// ------------------------------------------------------------------ output
OutFile::~OutFile() { if (f_) close(); }

bool OutFile::open(const char *path, int threads)
{
    if (threads < 1) { threads = (int)std::thread::hardware_concurrency() / 4; if (threads > 32) threads = 32; if (threads < 2) threads = 2; }
    threads_ = threads;
    if (!path) { f_ = stdout; own_ = false; gz_ = false; return true; }
    gz_ = has_gz_ext(path);
    f_ = fopen(path, "wb");
    if (f_ && gz_) setvbuf(f_, nullptr, _IOFBF, 1 << 20);
    return f_ != nullptr;
}

bool OutFile::write(const char *p, size_t n)
{
    if (!gz_) return fwrite(p, 1, n, f_) == n;
    pend_.insert(pend_.end(), p, p + n);
    if (pend_.size() >= (size_t)threads_ << 22) return flush_members();     // four slices per thread: thread start-up and the serial write amortised
    return true;
}

// Gzip output is ONE member, as flate2's GzEncoder writes and GzDecoder (the reference's own reader, helper.rs:22) expects --
// a single-member reader stops after the first member of a multi-member file -- but it is compressed on many threads the
// way pigz does it: the pending text is cut into 1 MiB slices, every slice becomes a run of non-final deflate blocks ending
// on a byte boundary (Z_SYNC_FLUSH from a reset state, so no back-reference crosses a slice), the runs are written in
// order behind one header, and close() appends an empty final block, the combined CRC-32 and the length.  No allocation
// per slice: every worker keeps one deflate state for the whole flush and compresses straight into its slice's place of
// one output buffer that lives as long as the file (fresh mappings per slice made 32 threads queue on the mmap lock).
bool OutFile::flush_members()
{
    if (pend_.empty()) return true;
    const size_t slice = (size_t)1 << 20, n = (pend_.size() + slice - 1) / slice;
    const size_t stride = slice + slice / 1000 + 128;               // deflateBound of a slice plus the sync marker, with room to spare
    if (obuf_.size() < n * stride) obuf_.resize(n * stride);
    std::vector<size_t> olen(n, 0);
    std::vector<uint32_t> crcs(n, 0);
    std::atomic<size_t> next{0}; std::atomic<int> bad{0};
    auto work = [&] {
        z_stream z; memset(&z, 0, sizeof z);
        if (deflateInit2(&z, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad = 1; return; }
        for (size_t i; (i = next++) < n;) {
            const size_t off = i * slice, len = std::min(slice, pend_.size() - off);
            unsigned char *o = obuf_.data() + i * stride;
            deflateReset(&z);
            z.next_in = (Bytef *)(pend_.data() + off); z.avail_in = (uInt)len;
            z.next_out = o; z.avail_out = (uInt)stride;
            if (deflate(&z, Z_SYNC_FLUSH) != Z_OK || z.avail_in != 0 || z.avail_out == 0) { bad = 1; continue; }
            olen[i] = z.total_out;
            crcs[i] = (uint32_t)crc32(0, (const Bytef *)(pend_.data() + off), (uInt)len);
        }
        deflateEnd(&z);
    };
    {
        std::vector<std::thread> th;
        const size_t T = std::min<size_t>(n, (size_t)threads_);
        for (size_t t = 1; t < T; t++) th.emplace_back(work);
        work();
        for (auto &x : th) x.join();
    }
    if (bad) { pend_.clear(); return false; }
    if (!wrote_) {
        static const unsigned char hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
        if (fwrite(hdr, 1, 10, f_) != 10) return false;
        wrote_ = true;
    }
    for (size_t i = 0; i < n; i++) {
        const size_t len = std::min(slice, pend_.size() - i * slice);
        crc_ = (uint32_t)crc32_combine(crc_, crcs[i], (z_off_t)len);
        total_ += len;
        if (fwrite(obuf_.data() + i * stride, 1, olen[i], f_) != olen[i]) return false;
    }
    pend_.clear();
    return true;
}

bool OutFile::close()
{
    if (!f_) return true;
    bool ok = true;
    if (gz_) {
        ok = flush_members();
        if (ok && !wrote_) {                                  // no text at all: header only so far
            static const unsigned char hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
            ok = fwrite(hdr, 1, 10, f_) == 10;
        }
        if (ok) {                                             // empty final block (fixed Huffman, end of block), CRC-32, ISIZE
            unsigned char tail[10] = {3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const uint32_t isz = (uint32_t)total_;
            memcpy(tail + 2, &crc_, 4); memcpy(tail + 6, &isz, 4);
            ok = fwrite(tail, 1, 10, f_) == 10;
        }
    }
    if (own_) ok = (fclose(f_) == 0) && ok; else ok = (fflush(f_) == 0) && ok;
    f_ = nullptr;
    return ok;
}

bool write_survivors(const char *path, const FqRec *recs, uint64_t n, const uint8_t *keep, std::string &err)
{
    OutFile out;
    if (!out.open(path)) { err = std::string("Cannot open file ") + path; return false; }
    std::vector<char> buf; buf.reserve(1 << 22);
    bool ok = true;
    for (uint64_t i = 0; i < n && ok; i++) {
        if (!keep[i]) continue;
        const FqRec &r = recs[i];
        buf.insert(buf.end(), r.h, r.h + r.hl); buf.push_back('\n');
        buf.insert(buf.end(), r.s, r.s + r.sl); buf.push_back('\n'); buf.push_back('+'); buf.push_back('\n');
        buf.insert(buf.end(), r.q, r.q + r.ql); buf.push_back('\n');
        if (buf.size() > (1u << 22) - 4096) { ok = out.write(buf.data(), buf.size()); buf.clear(); }
    }
    ok = ok && out.write(buf.data(), buf.size());
    ok = out.close() && ok;
    if (!ok) err = std::string("write error on ") + path;
    return ok;
}

} // namespace mf
