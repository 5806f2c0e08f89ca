// `fastfilter` drop-in for MitoFlex (installed as mitoflex_amd/assemble/fastfilter,
// the path MEGAHIT.FAST_FILTER resolves: assemble/assemble_wrapper.py:105-108).
//
// Personality 1 -- the reference's contig filter, argument for argument and
// quirk for quirk (assemble/fastfilter_src/src/main.rs:9-134, helper.rs:12-41):
//     fastfilter -i IN -o OUT -l MIN,MAX (-d INT | -m INT)
// stdout carries the kept count and nothing else (the caller does int(stdout),
// assemble_wrapper.py:326-339); a Rust panic is mirrored as exit code 101, a
// clap usage error as exit code 1.  This part is plain host C++: it is a
// kB-MB text filter and the reference itself is single-threaded host code.
//
// Personality 2 -- the read pre-filter the north star adds (no reference
// counterpart):
//     fastfilter bait --bait BAIT.fa -k 31 [-t 1] --fq1 R1.fq [--fq2 R2.fq]
//                     --out1 O1.fq [--out2 O2.fq] [--pair either|both] [--devices N]
// which loads libmitofilter_hip.so (HIP kernels, gfx950) and prints the kept
// read/pair count.  It has no CPU fallback: without the library or a GPU it
// exits non-zero, which shell_call turns into a RuntimeError (helper.py:82-86).
#include "../../include/mitofilter.h"
#include "mf_coldtrace.h"

#include <dlfcn.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <string>
#include <utility>
#include <vector>

// ------------------------------------------------------------ error exits
static void (*g_flush_on_panic)() = nullptr;   // BufWriter::drop flushes while the panic unwinds

[[noreturn]] static void rust_panic(const std::string &msg)
{   // mirrors `panic!`/`unwrap()` failures: message on stderr, exit code 101
    if (g_flush_on_panic) { void (*f)() = g_flush_on_panic; g_flush_on_panic = nullptr; f(); }
    fflush(stdout);
    fprintf(stderr, "thread 'main' panicked at '%s'\n", msg.c_str());
    exit(101);
}

static const char *USAGE = "USAGE:\n    fastfilter [OPTIONS] -i <PATH> -l <INT,INT> -o <PATH>\n\nFor more information try --help\n";

[[noreturn]] static void clap_error(const std::string &msg)
{   // clap 2.33 usage errors exit with code 1
    fprintf(stderr, "error: %s\n\n%s", msg.c_str(), USAGE);
    exit(1);
}

static void print_help()
{
    fputs("Length filter 0.1\nJunyu Li\nA simple filter for trimming intermediate contigs.\n\n"
          "USAGE:\n    fastfilter [OPTIONS] -i <PATH> -l <INT,INT> -o <PATH>\n\n"
          "FLAGS:\n    -h, --help       Prints help information\n    -V, --version    Prints version information\n\n"
          "OPTIONS:\n"
          "    -m <INT>            Use it to take only x sequence with highest depth.\n"
          "    -d <INT>            min depth\n"
          "    -i <PATH>           Input file path, accepts only one-line format\n"
          "    -l <INT,INT>        required min and max length\n"
          "    -o <PATH>           Output file path\n", stdout);
}

// ------------------------------------------------------------ Rust parsers
// str::parse::<usize>: optional '+', then ASCII digits only, no overflow
static bool parse_usize(const std::string &s, uint64_t &out)
{
    size_t i = 0;
    if (i < s.size() && s[i] == '+') i++;
    if (i >= s.size()) return false;
    uint64_t v = 0;
    for (; i < s.size(); i++) {
        if (s[i] < '0' || s[i] > '9') return false;
        const uint64_t d = (uint64_t)(s[i] - '0');
        if (v > (UINT64_MAX - d) / 10) return false;
        v = v * 10 + d;
    }
    out = v;
    return true;
}

// str::parse::<i32>: optional sign, digits, range checked
static bool parse_i32(const std::string &s, int32_t &out)
{
    size_t i = 0; bool neg = false;
    if (i < s.size() && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; i++; }
    if (i >= s.size()) return false;
    int64_t v = 0;
    for (; i < s.size(); i++) {
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (s[i] - '0');
        if (v > (int64_t)INT32_MAX + 1) return false;
    }
    if (neg) v = -v;
    if (v < INT32_MIN || v > INT32_MAX) return false;
    out = (int32_t)v;
    return true;
}

// str::parse::<f32> as the shipped binary does it (goldens X14-X17): [sign] then exactly "inf" or
// "NaN", or (digits [. digits] | . digits) [(e|E) [sign] digits]; no surrounding whitespace, no hex,
// no "infinity"/"nan".  Value correctly rounded to binary32 (strtof does the same).
static bool parse_f32(const std::string &s, float &out)
{
    size_t i = 0, n = s.size();
    bool neg = false;
    if (i < n && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; i++; }
    const std::string body = s.substr(i);
    if (body == "inf") { out = neg ? -INFINITY : INFINITY; return true; }
    if (body == "NaN") { out = NAN; return true; }
    size_t j = i, nd = 0;
    while (j < n && isdigit((unsigned char)s[j])) { j++; nd++; }
    if (j < n && s[j] == '.') { j++; while (j < n && isdigit((unsigned char)s[j])) { j++; nd++; } }
    if (nd == 0) return false;
    if (j < n && (s[j] == 'e' || s[j] == 'E')) {
        j++;
        if (j < n && (s[j] == '+' || s[j] == '-')) j++;
        size_t ne = 0;
        while (j < n && isdigit((unsigned char)s[j])) { j++; ne++; }
        if (ne == 0) return false;
    }
    if (j != n) return false;
    out = strtof(s.c_str(), nullptr);
    return true;
}

static bool valid_utf8(const char *p, size_t n)
{
    const unsigned char *s = (const unsigned char *)p;
    size_t i = 0;
    while (i < n) {
        if (i + 8 <= n) {                       // ASCII fast path, 8 bytes at a time
            uint64_t v; memcpy(&v, s + i, 8);
            if (!(v & 0x8080808080808080ULL)) { i += 8; continue; }
        }
        unsigned char c = s[i];
        if (c < 0x80) { i++; continue; }
        int len; uint32_t cp, min;
        if ((c & 0xE0) == 0xC0) { len = 2; cp = c & 0x1F; min = 0x80; }
        else if ((c & 0xF0) == 0xE0) { len = 3; cp = c & 0x0F; min = 0x800; }
        else if ((c & 0xF8) == 0xF0) { len = 4; cp = c & 0x07; min = 0x10000; }
        else return false;
        if (i + len > n) return false;
        for (int k = 1; k < len; k++) { if ((s[i + k] & 0xC0) != 0x80) return false; cp = (cp << 6) | (s[i + k] & 0x3F); }
        if (cp < min || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
        i += len;
    }
    return true;
}

// Clinger's fast path in binary32: a decimal with < 2^24 mantissa and <= 10 fractional digits is
// float(m) / 10^f with both operands exact, so the single division rounds correctly -- the same
// value Rust's (correctly rounded) parser and strtof produce.  Anything else goes to parse_f32.
static bool parse_f32_fast(const char *p, size_t n, float &out)
{
    static const float P10[11] = {1e0f, 1e1f, 1e2f, 1e3f, 1e4f, 1e5f, 1e6f, 1e7f, 1e8f, 1e9f, 1e10f};
    uint32_t m = 0; size_t i = 0, nd = 0, frac = 0;
    for (; i < n && p[i] >= '0' && p[i] <= '9'; i++, nd++) { if (m > 1677721) return false; m = m * 10 + (uint32_t)(p[i] - '0'); }
    if (i < n && p[i] == '.') {
        for (i++; i < n && p[i] >= '0' && p[i] <= '9'; i++, nd++, frac++) { if (m > 1677721) return false; m = m * 10 + (uint32_t)(p[i] - '0'); }
    }
    if (i != n || nd == 0 || frac > 10 || m >= (1u << 24)) return false;
    out = (float)m / P10[frac];
    return true;
}

// title.split_whitespace()[2].split('=')[1].parse::<f32>().unwrap()   (main.rs:86-91 / :120-124)
static float header_depth(const char *t, size_t n)
{
    // str::split_whitespace(): Unicode White_Space (the line is already known to be valid UTF-8)
    auto ws_len = [&](size_t i) -> size_t {       // length of the white-space character at i, 0 if none
        const unsigned char c = (unsigned char)t[i];
        if (c < 0x80) return (c == ' ' || (c >= 9 && c <= 13)) ? 1 : 0;
        if (c == 0xC2 && i + 1 < n && ((unsigned char)t[i + 1] == 0x85 || (unsigned char)t[i + 1] == 0xA0)) return 2;
        if (i + 2 < n) {
            const unsigned char d = (unsigned char)t[i + 1], e = (unsigned char)t[i + 2];
            if (c == 0xE1 && d == 0x9A && e == 0x80) return 3;                                   // U+1680
            if (c == 0xE2 && d == 0x80 && ((e >= 0x80 && e <= 0x8A) || e == 0xA8 || e == 0xA9 || e == 0xAF)) return 3;   // U+2000-200A, 2028, 2029, 202F
            if (c == 0xE2 && d == 0x81 && e == 0x9F) return 3;                                   // U+205F
            if (c == 0xE3 && d == 0x80 && e == 0x80) return 3;                                   // U+3000
        }
        return 0;
    };
    size_t i = 0, ntok = 0, tb = 0, tl = 0;
    while (i < n && ntok < 3) {
        size_t l;
        while (i < n && (l = ws_len(i)) > 0) i += l;
        const size_t b = i;
        while (i < n && ws_len(i) == 0) i++;
        if (i > b) { ntok++; tb = b; tl = i - b; }
    }
    if (ntok < 3) rust_panic("index out of bounds: the len is " + std::to_string(ntok) + " but the index is 2");
    const char *f = t + tb;
    const char *e1 = (const char *)memchr(f, '=', tl);
    if (!e1) rust_panic("index out of bounds: the len is 1 but the index is 1");
    const char *vb = e1 + 1;
    const char *e2 = (const char *)memchr(vb, '=', (size_t)(f + tl - vb));
    const size_t vl = (size_t)((e2 ? e2 : f + tl) - vb);
    float v;
    if (parse_f32_fast(vb, vl, v)) return v;
    if (!parse_f32(std::string(vb, vl), v)) rust_panic("called `Result::unwrap()` on an `Err` value: ParseFloatError { kind: Invalid }");
    return v;
}

// ----------------------------------------------------------------- file IO
static bool has_gz_ext(const std::string &path)
{   // Path::extension() == Some("gz")  (helper.rs:19,34)
    size_t slash = path.rfind('/');
    std::string name = slash == std::string::npos ? path : path.substr(slash + 1);
    size_t dot = name.rfind('.');
    return dot != std::string::npos && dot != 0 && name.substr(dot) == ".gz";
}

// Input bytes: plain files are mapped (no copy), .gz files are inflated into `owned`.
struct Input { const char *p = nullptr; size_t n = 0; std::string owned; };

static void read_all(const std::string &path, std::string &data);
static void open_input(const std::string &path, Input &in)
{
    if (!has_gz_ext(path)) {
        int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) rust_panic("Cannot open file " + path + "!");
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
            close(fd);
            if (m != MAP_FAILED) { in.p = (const char *)m; in.n = (size_t)st.st_size; return; }
        } else close(fd);
    }
    read_all(path, in.owned);
    in.p = in.owned.data(); in.n = in.owned.size();
}

static void read_all(const std::string &path, std::string &data)
{
    if (has_gz_ext(path)) {
        FILE *probe = fopen(path.c_str(), "rb");
        if (!probe) rust_panic("Cannot open file " + path + "!");
        fclose(probe);
        gzFile g = gzopen(path.c_str(), "rb");
        if (!g) rust_panic("Cannot open file " + path + "!");
        gzbuffer(g, 128 * 1024);
        char buf[1 << 16]; int n;
        while ((n = gzread(g, buf, sizeof buf)) > 0) data.append(buf, (size_t)n);
        const bool bad = n < 0;
        gzclose(g);
        if (bad) rust_panic("called `Result::unwrap()` on an `Err` value: Custom { kind: InvalidInput, error: \"corrupt deflate stream\" }");
        return;
    }
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) rust_panic("Cannot open file " + path + "!");
    struct stat st;
    size_t have = 0;
    if (fstat(fileno(f), &st) == 0 && st.st_size > 0) {      // regular file: one allocation, one read
        data.resize((size_t)st.st_size);
        have = fread(&data[0], 1, data.size(), f);
        data.resize(have);
    }
    char buf[1 << 16]; size_t n;                              // pipes / files that grew
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) data.append(buf, n);
    fclose(f);
}

struct Writer {
    bool gz = false; gzFile g = nullptr; FILE *f = nullptr; std::string buf;
    void open(const std::string &path)
    {
        gz = has_gz_ext(path);
        if (gz) { g = gzopen(path.c_str(), "wb6"); if (!g) rust_panic("Cannot open file " + path); }   // Compression::default() = level 6
        else { f = fopen(path.c_str(), "wb"); if (!f) rust_panic("Cannot open file " + path); }
        buf.reserve((1u << 20) + 65536);
    }
    void drain() { if (buf.empty()) return; if (gz) gzwrite(g, buf.data(), (unsigned)buf.size()); else fwrite(buf.data(), 1, buf.size(), f); buf.clear(); }
    void line(const char *p, size_t n) { buf.append(p, n); buf.push_back('\n'); if (buf.size() >= (1u << 20)) drain(); }
    void close() { drain(); if (gz && g) gzclose(g); if (f) fclose(f); g = nullptr; f = nullptr; }
};

struct Line { const char *p; size_t n; };
// BufRead::lines(): split at '\n', drop one trailing '\r', final line may lack '\n', empty trailing piece not yielded
static void split_lines(const Input &data, std::vector<Line> &lines)
{
    const char *p = data.p, *end = p + data.n;
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        size_t n = (size_t)(le - p);
        if (n && p[n - 1] == '\r') n--;
        lines.push_back(Line{p, n});
        if (!nl) break;
        p = nl + 1;
    }
}

// ------------------------------------------------------------ contig filter
static int contig_filter_main(int argc, char **argv)
{
    std::string v_l, v_d, v_i, v_o, v_m;
    bool has_l = false, has_d = false, has_i = false, has_o = false, has_m = false;
    for (int a = 1; a < argc; a++) {
        std::string arg = argv[a];
        if (arg == "-h" || arg == "--help") { print_help(); return 0; }
        if (arg == "-V" || arg == "--version") { puts("Length filter 0.1"); return 0; }
        if (arg.size() < 2 || arg[0] != '-' || arg[1] == '-')
            clap_error("Found argument '" + arg + "' which wasn't expected, or isn't valid in this context");
        const char o = arg[1];
        if (o != 'l' && o != 'd' && o != 'i' && o != 'o' && o != 'm')
            clap_error("Found argument '-" + std::string(1, o) + "' which wasn't expected, or isn't valid in this context");
        std::string val;
        if (arg.size() > 2) { val = arg.substr(2); if (val[0] == '=') val = val.substr(1); }
        else {
            if (a + 1 >= argc) clap_error("The argument '-" + std::string(1, o) + " <" + (o == 'i' || o == 'o' ? "PATH" : o == 'l' ? "INT,INT" : "INT") + ">' requires a value but none was supplied");
            val = argv[++a];
            if (!val.empty() && val[0] == '-' && val.size() > 1)
                clap_error("Found argument '" + val + "' which wasn't expected, or isn't valid in this context");
        }
        bool *has = o == 'l' ? &has_l : o == 'd' ? &has_d : o == 'i' ? &has_i : o == 'o' ? &has_o : &has_m;
        std::string *dst = o == 'l' ? &v_l : o == 'd' ? &v_d : o == 'i' ? &v_i : o == 'o' ? &v_o : &v_m;
        if (*has) clap_error("The argument '-" + std::string(1, o) + "' was provided more than once, but cannot be used multiple times");
        *has = true; *dst = val;
    }
    if (has_d && has_m) clap_error("The argument '-d <INT>' cannot be used with '-m <INT>'");
    if (!has_i || !has_l || !has_o) {
        std::string miss;
        if (!has_i) miss += "\n    -i <PATH>";
        if (!has_l) miss += "\n    -l <INT,INT>";
        if (!has_o) miss += "\n    -o <PATH>";
        clap_error("The following required arguments were not provided:" + miss);
    }

    // main.rs:57-67
    std::vector<uint64_t> lengths;
    {
        size_t b = 0;
        for (;;) {
            size_t c = v_l.find(',', b);
            std::string piece = v_l.substr(b, c == std::string::npos ? std::string::npos : c - b);
            uint64_t v;
            if (!piece.empty()) {                // clap's delimiter split drops empty pieces (golden X37)
                if (!parse_usize(piece, v)) rust_panic("called `Result::unwrap()` on an `Err` value: ParseIntError { kind: InvalidDigit }");
                lengths.push_back(v);
            }
            if (c == std::string::npos) break;
            b = c + 1;
        }
    }
    if (lengths.size() != 2) { puts("Input length string not valid, please input INT,INT."); }
    if (lengths.size() < 2) rust_panic("index out of bounds: the len is " + std::to_string(lengths.size()) + " but the index is " + std::to_string(lengths.size()));
    const uint64_t min = lengths[0], max = lengths[1];

    Input data; open_input(v_i, data);           // main.rs:69
    static Writer out; out.open(v_o);            // main.rs:70
    g_flush_on_panic = [] { out.close(); };
    uint64_t count = 0;
    std::vector<Line> lines;

    if (!has_m) {
        // main.rs:74-105 -- pairs of lines are taken straight off the buffer (lines().tuples())
        if (!has_d) rust_panic("called `Option::unwrap()` on a `None` value");
        int32_t depth;
        if (!parse_i32(v_d, depth)) rust_panic("called `Result::unwrap()` on an `Err` value: ParseIntError { kind: InvalidDigit }");
        const char *p = data.p, *end = p + data.n;
        auto next_line = [&](Line &l) -> bool {      // BufRead::lines(): split at LF, drop one trailing CR
            if (p >= end) return false;
            const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
            const char *le = nl ? nl : end;
            size_t n = (size_t)(le - p);
            if (n && p[n - 1] == '\r') n--;
            l = Line{p, n};
            p = nl ? nl + 1 : end;
            return true;
        };
        Line t, s;
        while (next_line(t) && next_line(s)) {
            if (!valid_utf8(t.p, t.n) || !valid_utf8(s.p, s.n)) rust_panic("called `Result::unwrap()` on an `Err` value: Custom { kind: InvalidData, error: \"stream did not contain valid UTF-8\" }");
            if (t.n == 0 || t.p[0] != '>') continue;
            if (depth != 0) {
                const float seq_depth = header_depth(t.p, t.n);
                if ((float)depth > seq_depth) continue;
            }
            const uint64_t length = (uint64_t)s.n - 1;            // wraps for an empty sequence, as the release binary does
            if (length < min || length > max) continue;
            out.line(t.p, t.n); out.line(s.p, s.n);
            count++;
        }
    } else {
        // main.rs:106-131
        split_lines(data, lines);
        uint64_t max_count;
        if (!parse_usize(v_m, max_count)) rust_panic("called `Result::unwrap()` on an `Err` value: ParseIntError { kind: InvalidDigit }");
        for (const Line &l : lines)
            if (!valid_utf8(l.p, l.n)) rust_panic("called `Result::unwrap()` on an `Err` value: Custom { kind: InvalidData, error: \"stream did not contain valid UTF-8\" }");
        std::vector<std::pair<Line, Line>> seqs;
        for (size_t i = 0; i + 1 < lines.size(); i += 2)
            if (lines[i + 1].n <= max && lines[i + 1].n >= min) seqs.emplace_back(lines[i], lines[i + 1]);
        // sort_by_cached_key with a unit key: stable no-op, but the key closure still
        // runs (and can panic) for every element when there are at least two
        if (seqs.size() >= 2) for (auto &p : seqs) (void)header_depth(p.first.p, p.first.n);
        for (size_t j = seqs.size(); j-- > 0 && count < max_count;) {
            out.line(seqs[j].first.p, seqs[j].first.n); out.line(seqs[j].second.p, seqs[j].second.n);
            count++;
        }
    }
    g_flush_on_panic = nullptr;
    out.close();
    printf("%llu\n", (unsigned long long)count);
    return 0;
}

// ---------------------------------------------------------------- bait mode
static std::string exe_dir()
{
    char buf[PATH_MAX]; ssize_t n = readlink("/proc/self/exe", buf, sizeof buf - 1);
    if (n <= 0) return ".";
    buf[n] = 0;
    std::string p(buf); size_t s = p.rfind('/');
    return s == std::string::npos ? "." : p.substr(0, s);
}

static int bait_main(int argc, char **argv)
{
    std::string bait, fq1, fq2, out1, out2, pair = "either", libpath;
    int k = 0, devices = 1, gcode = 5; unsigned thr = 1; bool protein = false;
    std::vector<int> device_list;              // --device-list 2,3: these devices instead of 0 .. N - 1
    std::vector<std::pair<std::string, std::string>> options;      // --option pass=serial: how a filter pass is run (mf_set_option)
    for (int a = 2; a < argc; a++) {
        std::string o = argv[a];
        auto need = [&](const char *name) -> std::string {
            if (a + 1 >= argc) { fprintf(stderr, "error: %s requires a value\n", name); exit(1); }
            return argv[++a];
        };
        if (o == "--bait") bait = need("--bait");
        else if (o == "--fq1") fq1 = need("--fq1");
        else if (o == "--fq2") fq2 = need("--fq2");
        else if (o == "--out1") out1 = need("--out1");
        else if (o == "--out2") out2 = need("--out2");
        else if (o == "--pair") pair = need("--pair");
        else if (o == "--lib") libpath = need("--lib");
        else if (o == "-k" || o == "--kmer") k = atoi(need("-k").c_str());
        else if (o == "-t" || o == "--threshold") thr = (unsigned)strtoul(need("-t").c_str(), nullptr, 10);
        else if (o == "--devices") devices = atoi(need("--devices").c_str());
        else if (o == "--device-list") {
            const std::string v = need("--device-list");
            for (size_t i = 0; i < v.size();) { size_t j = v.find(',', i); if (j == std::string::npos) j = v.size(); if (j > i) device_list.push_back(atoi(v.substr(i, j - i).c_str())); i = j + 1; }
            if (device_list.empty()) { fprintf(stderr, "error: --device-list wants device numbers separated by commas\n"); return 1; }
        }
        else if (o == "--option") {
            const std::string v = need("--option"); const size_t eq = v.find('=');
            if (eq == std::string::npos || eq == 0) { fprintf(stderr, "error: --option wants name=value\n"); return 1; }
            options.emplace_back(v.substr(0, eq), v.substr(eq + 1));
        }
        else if (o == "--protein") protein = true;                       // --bait is a protein FASTA (e.g. profile/MT_database/<clade>.fa)
        else if (o == "--code" || o == "--genetic-code") gcode = atoi(need("--code").c_str());
        else { fprintf(stderr, "error: unknown option '%s' for fastfilter bait\n", o.c_str()); return 1; }
    }
    if (bait.empty() || fq1.empty() || out1.empty() || (fq2.empty() != out2.empty()) || (pair != "either" && pair != "both")) {
        fputs("usage: fastfilter bait --bait BAIT.fa [-k 31] [-t 1] --fq1 R1.fq [--fq2 R2.fq] --out1 O1.fq [--out2 O2.fq]"
              " [--pair either|both] [--devices N | --device-list D0,D1,..] [--option name=value ..]\n"
              "       fastfilter bait --protein --bait PROTEINS.fa [--code 5] [-k 9] ...   (six-frame peptide k-mers)\n", stderr);
        return 1;
    }
    if (k == 0) k = protein ? 9 : 31;
    if (libpath.empty()) { const char *e = getenv("MITOFILTER_LIB"); if (e && *e) libpath = e; }          // (as the Python wrapper and filter_v2 do)
    if (libpath.empty()) libpath = exe_dir() + "/../libmitofilter_hip.so";
    mf::cold_mark("fastfilter bait: arguments read");
    void *h = dlopen(libpath.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "error: cannot load %s: %s (the bait filter has no CPU fallback)\n", libpath.c_str(), dlerror()); return 2; }
#define SYM(name) auto p_##name = (decltype(&name))dlsym(h, #name); if (!p_##name) { fprintf(stderr, "error: %s lacks symbol %s\n", libpath.c_str(), #name); return 2; }
    SYM(mf_abi_version) SYM(mf_last_error) SYM(mf_kmerset_build_from_fasta) SYM(mf_kmerset_build_protein_from_fasta)
    SYM(mf_filter_fastq_files) SYM(mf_filter_fastq_files_on) SYM(mf_kmerset_free) SYM(mf_set_option)
#undef SYM
    if (p_mf_abi_version() != MF_ABI_VERSION) { fprintf(stderr, "error: ABI version mismatch\n"); return 2; }
    (void)p_mf_set_option("expect_files", "1"); (void)p_mf_set_option("short_lived", "1");          // (the file-level call's set-up starts beside the bait set's build)
    for (auto &kv : options) if (p_mf_set_option(kv.first.c_str(), kv.second.c_str()) != MF_OK) { fprintf(stderr, "error: %s\n", p_mf_last_error()); return 1; }
    mf::cold_mark("library loaded");
    mf_kmerset *ks = nullptr;
    const int dev0 = device_list.empty() ? 0 : device_list[0];
    const int brc = protein ? p_mf_kmerset_build_protein_from_fasta(bait.c_str(), k, gcode, dev0, &ks)
                            : p_mf_kmerset_build_from_fasta(bait.c_str(), k, dev0, &ks);
    if (brc != MF_OK) { fprintf(stderr, "error: %s\n", p_mf_last_error()); return 3; }
    mf::cold_mark("bait set built");
    uint64_t kept = 0, total = 0;
    setenv("MF_DEVPOOL_GB", "4096", 0);          // (a process that ends with the call gives no device memory back in between: the runtime frees it all at once)
    int rc = device_list.empty()
        ? p_mf_filter_fastq_files(ks, fq1.c_str(), fq2.empty() ? nullptr : fq2.c_str(), out1.c_str(),
                                  out2.empty() ? nullptr : out2.c_str(), thr, pair == "both" ? MF_PAIR_BOTH : MF_PAIR_EITHER,
                                  devices, &kept, &total)
        : p_mf_filter_fastq_files_on(ks, fq1.c_str(), fq2.empty() ? nullptr : fq2.c_str(), out1.c_str(),
                                     out2.empty() ? nullptr : out2.c_str(), thr, pair == "both" ? MF_PAIR_BOTH : MF_PAIR_EITHER,
                                     device_list.data(), (int)device_list.size(), &kept, &total);
    if (rc != MF_OK) { fprintf(stderr, "error: %s\n", p_mf_last_error()); p_mf_kmerset_free(ks); return 3; }
    mf::cold_mark("files filtered");
    printf("%llu\n", (unsigned long long)kept);      // same stdout contract as the contig filter
    // (the outputs are written and closed; what is left is the GPU runtime's teardown -- queues, code objects, a tenth of a second -- which a
    // process that is about to be gone has no use for; a profiler writes its files in a finaliser, so not under one)
    fflush(stdout); fflush(stderr);
    const char *pre = getenv("LD_PRELOAD");
    if (!((pre && strstr(pre, "rocprof")) || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") || getenv("ROCP_TOOL_LIBRARIES"))) _exit(0);
    p_mf_kmerset_free(ks);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && strcmp(argv[1], "bait") == 0) return bait_main(argc, argv);
    return contig_filter_main(argc, argv);
}
