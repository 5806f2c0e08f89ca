// TEST INFRASTRUCTURE: the ORCHESTRATION of the device ingest path (mitoflex_amd/csrc/mf_devingest.cpp: producer, uploader, consumers,
// writers, ring, text-buffer pool, carry hand-off, the quality filter's decide / gather turns) on the CPU, against a stand-in device
// (tests/native/hipstub: streams are real in-order queues on threads of their own; tests/native/ingest_stub.cpp: the kernels as obvious
// loops), with knobs so small that every seam falls inside everything.  Built plain and with -fsanitize=thread by
// tests/test_ingest_orchestration.py.  What is checked: the output files equal what this file computes from the FASTQ text with a few plain
// loops (the stand-in filter's rule; the reference's filter_v2 rules, filter/filter_bin/src/main.rs:236-268); a hang is a failure
// (watchdog).  What is NOT checked here: the HIP kernels -- those have their own parity tests on the GPU (tests/test_gpu_devingest.py).
//   ingest_check <case> <scratch dir>      cases: see main()
#include "../../mitoflex_amd/csrc/mf_devingest.h"
#include "../../mitoflex_amd/csrc/mf_pipeline.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <set>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <initializer_list>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>
#include <zlib.h>

using std::string;
static uint64_t g_rng = 88172645463325252ull;
static uint32_t rnd() { g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17; return (uint32_t)(g_rng >> 20); }

struct Rec { string h, s, q; };
static std::vector<Rec> make_records(size_t n, uint32_t seed, bool uniform, const char *tag, uint32_t max_len = 220)
{
    g_rng = 88172645463325252ull ^ ((uint64_t)seed * 0x9E3779B97F4A7C15ull);
    std::vector<Rec> v(n);
    for (size_t i = 0; i < n; i++) {
        const uint32_t len = uniform ? 100 : 20 + rnd() % (max_len - 19);
        Rec &r = v[i];
        r.h = string("@") + tag + "." + std::to_string(i) + " x=" + std::to_string(rnd() % 1000);
        r.s.resize(len); r.q.resize(len);
        for (uint32_t k = 0; k < len; k++) { r.s[k] = "ACGT"[rnd() & 3]; r.q[k] = (char)('#' + rnd() % 40); }
        if (rnd() % 17 == 0) for (uint32_t k = 0, m = 1 + rnd() % 14; k < m; k++) r.s[rnd() % len] = 'N';
        if (rnd() % 41 == 0) r.s[rnd() % len] = "acgtRY"[rnd() % 6];
        if (i > 4 && rnd() % 23 == 0) { r.s = v[i - 1 - rnd() % 4].s; r.q.assign(r.s.size(), 'F'); }        // repeats: the de-duplication has something to find
    }
    return v;
}
static string fastq_text(const std::vector<Rec> &v, bool crlf, bool last_newline, const char *tail)
{
    const char *nl = crlf ? "\r\n" : "\n";
    string t;
    for (const Rec &r : v) { t += r.h; t += nl; t += r.s; t += nl; t += "+"; t += nl; t += r.q; t += nl; }
    if (tail) t += tail;
    if (!last_newline && !t.empty() && t.back() == '\n') { t.pop_back(); if (crlf && !t.empty() && t.back() == '\r') t.pop_back(); }
    return t;
}
static string gz_member(const string &text, int level, int flush_every = 0)
{
    z_stream z; memset(&z, 0, sizeof z);
    if (deflateInit2(&z, level, Z_DEFLATED, 31, 8, Z_DEFAULT_STRATEGY) != Z_OK) abort();
    string out(deflateBound(&z, (uLong)text.size()) + 64 + (flush_every ? text.size() / (size_t)flush_every * 16 : 0), '\0');
    z.next_out = (Bytef *)&out[0]; z.avail_out = (uInt)out.size();
    size_t pos = 0;
    while (pos < text.size()) {
        const size_t n = flush_every ? std::min<size_t>((size_t)flush_every, text.size() - pos) : text.size() - pos;
        z.next_in = (Bytef *)text.data() + pos; z.avail_in = (uInt)n; pos += n;
        if (deflate(&z, pos < text.size() ? Z_FULL_FLUSH : Z_NO_FLUSH) == Z_STREAM_ERROR) abort();
    }
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) abort();
    out.resize(out.size() - z.avail_out);
    deflateEnd(&z);
    return out;
}
static void write_file(const string &path, const string &bytes) { FILE *f = fopen(path.c_str(), "wb"); if (!f || fwrite(bytes.data(), 1, bytes.size(), f) != bytes.size()) { perror(path.c_str()); exit(2); } fclose(f); }
static string read_file(const string &path) { FILE *f = fopen(path.c_str(), "rb"); if (!f) return "<missing>"; string s; char b[65536]; size_t n; while ((n = fread(b, 1, sizeof b, f)) > 0) s.append(b, n); fclose(f); return s; }

// ---- what the path must produce, from the text alone
// the reference's FASTQ conventions (filter/filter_bin/src/main.rs:287-321): lines() (LF, a CR in front of it stripped, an unterminated last line counts), strict groups of four
static std::vector<Rec> parse(const string &t)
{
    std::vector<string> lines; size_t a = 0;
    while (a < t.size()) { size_t b = t.find('\n', a); string l = t.substr(a, b == string::npos ? string::npos : b - a); if (!l.empty() && l.back() == '\r') l.pop_back(); lines.push_back(l); if (b == string::npos) break; a = b + 1; }
    std::vector<Rec> v;
    for (size_t i = 0; i + 3 < lines.size(); i += 4) v.push_back(Rec{lines[i], lines[i + 1], lines[i + 3]});
    return v;
}
static bool stand_in_pass(const string &s)          // the rule of ingest_stub.cpp's filter_common, restated on the text
{
    uint64_t sum = 0;
    for (char c : s) switch (c & 0xDF) { case 'A': break; case 'C': sum += 1; break; case 'G': sum += 2; break; case 'T': sum += 3; break; default: sum += 7; }
    return sum % 5 == 0;
}
static string rec_text(const string &h, const string &s, const string &q) { return h + "\n" + s + "\n+\n" + q + "\n"; }

static int g_failed = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); g_failed++; } } while (0)

static void bait_case(const string &dir, const char *name, const string &t1, const string *t2, bool gz, int level, bool both, std::vector<int> devices, int members = 1, int flush_every = 0)
{
    auto to_file = [&](const string &t, const string &path) {
        if (!gz) { write_file(path, t); return; }
        string bytes;
        for (int m = 0; m < members; m++) { const size_t a = t.size() * (size_t)m / (size_t)members, b = t.size() * (size_t)(m + 1) / (size_t)members; bytes += gz_member(t.substr(a, b - a), level, flush_every); }
        if (members > 1) bytes += "trailing bytes that are no member";
        write_file(path, bytes);
    };
    const string ext = gz ? ".fq.gz" : ".fq";
    const string f1 = dir + "/" + name + "_1" + ext, f2 = dir + "/" + name + "_2" + ext, o1 = dir + "/" + name + "_o1.fq", o2 = dir + "/" + name + "_o2.fq";
    to_file(t1, f1); if (t2) to_file(*t2, f2);
    const std::vector<Rec> r1 = parse(t1), r2 = t2 ? parse(*t2) : std::vector<Rec>();
    const size_t n = t2 ? std::min(r1.size(), r2.size()) : r1.size();
    string e1, e2; uint64_t want_kept = 0;
    for (size_t i = 0; i < n; i++) {
        const bool a = stand_in_pass(r1[i].s), b = t2 ? stand_in_pass(r2[i].s) : a;
        if (both ? (a && b) : (a || b)) { want_kept++; e1 += rec_text(r1[i].h, r1[i].s, r1[i].q); if (t2) e2 += rec_text(r2[i].h, r2[i].s, r2[i].q); }
    }
    uint64_t kept = 0, total = 0; string err; mf::IngestStats st;
    const int rc = mf::run_device_ingest(nullptr, f1.c_str(), t2 ? f2.c_str() : nullptr, o1.c_str(), t2 ? o2.c_str() : nullptr, 1, both, devices.data(), (int)devices.size(), &kept, &total, err, &st);
    EXPECT(rc == 0, "%s: rc %d (%s)", name, rc, err.c_str());
    if (rc) return;
    EXPECT(total == n, "%s: total %llu, want %zu", name, (unsigned long long)total, n);
    EXPECT(kept == want_kept, "%s: kept %llu, want %llu", name, (unsigned long long)kept, (unsigned long long)want_kept);
    EXPECT(read_file(o1) == e1, "%s: output 1 differs (%zu bytes expected)", name, e1.size());
    if (t2) EXPECT(read_file(o2) == e2, "%s: output 2 differs (%zu bytes expected)", name, e2.size());
    fprintf(stderr, "ok %-28s %zu records, %llu kept, %d device(s), %llu chunks (%llu linked, %llu gaps)\n", name, n, (unsigned long long)kept, st.n_devices, (unsigned long long)st.chunks,
            (unsigned long long)st.chunks_linked, (unsigned long long)st.gaps);
}

// the reference's filter_v2 (main.rs:188-323) on parsed records
static void qual_case(const string &dir, const char *name, const string &t1, const string *t2, int level, mf::QualParams P, bool fifo = false)
{
    const string f1 = dir + "/" + name + "_1.fq.gz", f2 = dir + "/" + name + "_2.fq.gz", o1 = dir + "/" + name + "_o1.fq", o2 = dir + "/" + name + "_o2.fq";
    write_file(f1, gz_member(t1, level)); if (t2) write_file(f2, gz_member(*t2, level));
    const std::vector<Rec> r1 = parse(t1), r2 = t2 ? parse(*t2) : std::vector<Rec>();
    const size_t n = t2 ? std::min(r1.size(), r2.size()) : r1.size();
    auto cut = [&](const string &s) { if (s.size() < P.start) return string(); return P.end ? s.substr(P.start, P.end - P.start) : s.substr(P.start); };
    string e1, e2; uint64_t want_kept = 0, want_total = 0, budget = 0; std::set<string> seen;
    for (size_t i = 0; i < n; i++) {
        want_total = i + 1;
        const string s1 = cut(r1[i].s), q1 = cut(r1[i].q), s2 = t2 ? cut(r2[i].s) : "", q2 = t2 ? cut(r2[i].q) : "";
        bool keep = true;
        if (!P.trunc) {
            auto ns = [](const string &s) { return (uint64_t)std::count(s.begin(), s.end(), 'N'); };
            auto bad = [&](const string &q) { uint64_t b = 0; for (unsigned char c : q) b += c <= P.quality; return b; };
            if (ns(s1) > P.ns || (t2 && ns(s2) > P.ns)) keep = false;
            else {
                const float cf = (float)(t2 ? s1.size() : q1.size()) * P.limit; const uint64_t cutoff = !(cf > 0.0f) ? 0 : (uint64_t)cf;
                if (bad(q1) >= cutoff || (t2 && bad(q2) >= cutoff)) keep = false;
            }
            if (keep && P.dedup && !seen.insert(s1).second) keep = false;
        }
        if (!keep) continue;
        if (P.trim) { budget += s1.size(); if (budget > P.trim) { want_total = i; break; } }
        want_kept++;
        e1 += rec_text(r1[i].h, s1, q1); if (t2) e2 += rec_text(r2[i].h, s2, q2);
    }
    uint64_t kept = 0, total = 0; bool panicked = false; string err;
    // fifo: the outputs are named pipes with a slow reader each -- a sink that is no regular file takes its chunks in file order
    string got1, got2; std::vector<std::thread> readers;
    if (fifo) {
        auto reader = [](string path, string *into) { FILE *f = fopen(path.c_str(), "rb"); if (!f) return; char b[4096]; size_t n; while ((n = fread(b, 1, sizeof b, f)) > 0) { into->append(b, n); if (into->size() % 7 == 0) usleep(200); } fclose(f); };
        unlink(o1.c_str()); if (mkfifo(o1.c_str(), 0600) != 0) { perror("mkfifo"); exit(2); }
        readers.emplace_back(reader, o1, &got1);
        if (t2) { unlink(o2.c_str()); if (mkfifo(o2.c_str(), 0600) != 0) { perror("mkfifo"); exit(2); } readers.emplace_back(reader, o2, &got2); }
    }
    const int rc = mf::run_device_qualfilter(f1.c_str(), t2 ? f2.c_str() : nullptr, o1.c_str(), t2 ? o2.c_str() : nullptr, P, 0, &kept, &total, &panicked, err);
    for (auto &t : readers) t.join();
    EXPECT(rc == 0, "%s: rc %d (%s)", name, rc, err.c_str());
    if (rc) return;
    EXPECT(!panicked, "%s: reported a panic", name);
    EXPECT(kept == want_kept, "%s: kept %llu, want %llu", name, (unsigned long long)kept, (unsigned long long)want_kept);
    if (!P.trim) EXPECT(total == want_total, "%s: total %llu, want %llu", name, (unsigned long long)total, (unsigned long long)want_total);
    if (!fifo) { got1 = read_file(o1); if (t2) got2 = read_file(o2); }
    EXPECT(got1 == e1, "%s: output 1 differs (%zu bytes expected, %zu there)", name, e1.size(), got1.size());
    if (t2) EXPECT(got2 == e2, "%s: output 2 differs (%zu bytes expected)", name, e2.size());
    fprintf(stderr, "ok %-28s %zu records, %llu kept\n", name, n, (unsigned long long)kept);
}

static void tiny_knobs(bool carry_room)
{
    // (tests/test_gpu_devingest.py's STREAMING set: a 64 KiB ring under files of several hundred kilobytes, 4 KiB chunks, three-chunk slabs,
    // two text buffers a mate, pieces of at most 20 kB of text)
    setenv("MF_GZDEV_CHUNK_BYTES", "4096", 1); setenv("MF_GZDEV_SLAB_CHUNKS", "3", 1); setenv("MF_GZDEV_RING_BYTES", "65536", 1); setenv("MF_GZDEV_MARGIN", "8192", 1);
    setenv("MF_INGEST_TEXT_BUFS", "2", 1); setenv("MF_GZDEV_TEXT_PIECE", "20000", 1); setenv("MF_INGEST_SLAB_BYTES", "50001", 1);
    setenv("MF_INGEST_CARRY_ROOM", carry_room ? "1048576" : "0", 1);
    setenv("MF_QUAL_OUT_CHUNK", "4096", 1); setenv("MF_QUAL_OUT_CHUNKS", "6", 1); setenv("MF_DEDUP_LOG2_SLOTS", "4", 1);
    // INGEST_CHECK_KNOB_SEED: other sizes of everything, picked by a small generator from the seed (a case's own settings behind this call stay)
    if (const char *sd = getenv("INGEST_CHECK_KNOB_SEED")) {
        uint64_t x = 0x9E3779B97F4A7C15ull * (uint64_t)(atoi(sd) + 1);
        auto pick = [&](std::initializer_list<const char *> v) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return *(v.begin() + (size_t)(x % v.size())); };
        setenv("MF_GZDEV_CHUNK_BYTES", pick({"1024", "4096", "9000", "32768"}), 1); setenv("MF_GZDEV_SLAB_CHUNKS", pick({"1", "3", "7", "64"}), 1);
        setenv("MF_GZDEV_RING_BYTES", pick({"65536", "262144", "4194304"}), 1); setenv("MF_GZDEV_MARGIN", pick({"2048", "8192", "1048576"}), 1);
        setenv("MF_INGEST_TEXT_BUFS", pick({"2", "3", "6"}), 1); setenv("MF_GZDEV_TEXT_PIECE", pick({"20000", "300000", "1073741824"}), 1);
        setenv("MF_INGEST_SLAB_BYTES", pick({"50001", "400000", "268435456"}), 1); setenv("MF_INGEST_CONSUMERS", pick({"1", "2", "3", "5"}), 1);
        setenv("MF_GZDEV_UPLOAD_BUFS", pick({"2", "3", "4"}), 1); setenv("MF_GZDEV_RESOLVE_STREAM", pick({"0", "1"}), 1);
        setenv("MF_GZDEV_DEC_STREAMS", pick({"1", "2", "4"}), 1); setenv("MF_GZDEV_SLABS_IN_FLIGHT", pick({"1", "2", "5"}), 1);
        fprintf(stderr, "knobs of seed %s: chunk %s slab %s ring %s margin %s text bufs %s piece %s plain slab %s consumers %s upload bufs %s two post streams %s decode streams %s slabs in flight %s\n", sd,
                getenv("MF_GZDEV_CHUNK_BYTES"), getenv("MF_GZDEV_SLAB_CHUNKS"), getenv("MF_GZDEV_RING_BYTES"), getenv("MF_GZDEV_MARGIN"), getenv("MF_INGEST_TEXT_BUFS"), getenv("MF_GZDEV_TEXT_PIECE"),
                getenv("MF_INGEST_SLAB_BYTES"), getenv("MF_INGEST_CONSUMERS"), getenv("MF_GZDEV_UPLOAD_BUFS"), getenv("MF_GZDEV_RESOLVE_STREAM"), getenv("MF_GZDEV_DEC_STREAMS"), getenv("MF_GZDEV_SLABS_IN_FLIGHT"));
    }
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: ingest_check <case> <scratch dir>\n"); return 2; }
    const string what = argv[1], dir = argv[2];
    signal(SIGALRM, [](int) { static const char m[] = "FAILED: HANG (watchdog)\n"; (void)!write(2, m, sizeof m - 1); _exit(3); });
    alarm((unsigned)(getenv("INGEST_CHECK_TIMEOUT") ? atoi(getenv("INGEST_CHECK_TIMEOUT")) : 240));
    const size_t N = getenv("INGEST_CHECK_RECORDS") ? (size_t)atoi(getenv("INGEST_CHECK_RECORDS")) : 3000;
    const string a = fastq_text(make_records(N, 1, false, "a"), false, true, "@partial\nACGT\n");
    const string b = fastq_text(make_records(N - 100, 2, false, "b"), true, false, nullptr);          // CRLF, no newline at the very end, the shorter mate
    const string u = fastq_text(make_records(N, 3, true, "u"), false, true, nullptr);                 // uniform lengths: the read set without offsets
    if (what == "bait_se_gz") { tiny_knobs(true); bait_case(dir, "se6", a, nullptr, true, 6, false, {0}); bait_case(dir, "se1u", u, nullptr, true, 1, false, {0}); bait_case(dir, "se0", a, nullptr, true, 0, false, {0}); }
    else if (what == "bait_pe_gz") { tiny_knobs(true); bait_case(dir, "pe6", a, &b, true, 6, false, {0}); bait_case(dir, "pe9both", b, &a, true, 9, true, {0}); }
    else if (what == "bait_pe_gz_nocarry") { tiny_knobs(false); bait_case(dir, "pe6nc", a, &b, true, 6, false, {0}); }
    else if (what == "bait_plain") { tiny_knobs(true); bait_case(dir, "plain_se", a, nullptr, false, 0, false, {0}); bait_case(dir, "plain_pe", a, &b, false, 0, true, {0}); }
    else if (what == "bait_two_devices") { setenv("STUB_DEVICES", "2", 1); tiny_knobs(true); bait_case(dir, "pe6x2", a, &b, true, 6, false, {0, 1}); bait_case(dir, "plain_x2", a, &u, false, 0, false, {1, 0}); bait_case(dir, "members_x2", a, nullptr, true, 6, false, {0, 1}, 3); }
    else if (what == "bait_members_flush") { tiny_knobs(true); bait_case(dir, "members", a, &b, true, 6, false, {0}, 4); bait_case(dir, "flush", a, nullptr, true, 6, false, {0}, 1, 30000); }
    else if (what == "bait_margin") { tiny_knobs(true); setenv("MF_GZDEV_MARGIN", "1024", 1); bait_case(dir, "margin", a, nullptr, true, 9, false, {0}); }
    else if (what == "bait_default_knobs") { bait_case(dir, "dflt_pe", a, &b, true, 6, false, {0}); bait_case(dir, "dflt_plain", u, nullptr, false, 0, false, {0}); }
    else if (what == "bait_long_records") {
        tiny_knobs(false);
        const string l = fastq_text(make_records(40, 7, false, "l", 90000), false, true, nullptr);          // records longer than the deflate window and than a piece of text
        bait_case(dir, "long", l, nullptr, true, 6, false, {0});
    }
    else if (what == "bait_damaged") {
        tiny_knobs(true);
        string g = gz_member(a, 6); g[g.size() / 2] ^= 0x10;
        write_file(dir + "/dmg.fq.gz", g);
        uint64_t kept = 0, total = 0; string err; int dev = 0;
        const int rc = mf::run_device_ingest(nullptr, (dir + "/dmg.fq.gz").c_str(), nullptr, (dir + "/dmg_o.fq").c_str(), nullptr, 1, false, &dev, 1, &kept, &total, err);
        EXPECT(rc != 0 && rc != mf::MF_DEVINGEST_DECLINED, "a damaged stream must be an error (rc %d)", rc);
        fprintf(stderr, "ok damaged stream -> rc %d (%s)\n", rc, err.c_str());
    }
    else if (what == "qual_pe") {
        tiny_knobs(true);
        mf::QualParams P; P.quality = '+'; P.limit = 0.3f; P.ns = 3;
        qual_case(dir, "q_pe", a, &b, 6, P);
        P.dedup = true; qual_case(dir, "q_pe_dedup", a, &b, 6, P);
        P.start = 5; P.end = 60; qual_case(dir, "q_pe_cut", u, &a, 1, P);
    }
    else if (what == "qual_se") {
        tiny_knobs(true);
        mf::QualParams P; P.quality = '+'; P.limit = 0.3f; P.ns = 3; P.dedup = true;
        qual_case(dir, "q_se_dedup", a, nullptr, 6, P);
        P.trim = 40000; qual_case(dir, "q_se_budget", a, nullptr, 6, P);
        P = mf::QualParams(); P.trunc = true; P.end = 50; qual_case(dir, "q_se_trunc", u, nullptr, 6, P);
    }
    else if (what == "qual_pe_budget") {
        tiny_knobs(true); setenv("MF_INGEST_CONSUMERS", "5", 1);
        mf::QualParams P; P.quality = '+'; P.limit = 0.3f; P.ns = 3; P.trim = 90000;
        qual_case(dir, "q_pe_budget", a, &b, 6, P);
    }
    else if (what == "qual_pipes") {
        tiny_knobs(true); setenv("MF_INGEST_CONSUMERS", "5", 1); setenv("MF_QUAL_OUT_CHUNKS", "2", 1);          // (two chunks: a part further back that took them would leave none for the part whose turn it is)
        mf::QualParams P; P.quality = '+'; P.limit = 0.3f; P.ns = 3;
        qual_case(dir, "q_pipes_pe", a, &b, 6, P, true);
        P.dedup = true; qual_case(dir, "q_pipes_se", u, nullptr, 1, P, true);
    }
    else { fprintf(stderr, "unknown case %s\n", what.c_str()); return 2; }
    if (g_failed) { fprintf(stderr, "%d check(s) FAILED\n", g_failed); return 1; }
    fprintf(stderr, "all checks of %s passed; %llu streams made, %.1f MB of stand-in device memory at most\n", what.c_str(), (unsigned long long)stub_streams_made(), (double)stub_bytes_allocated_peak() / 1e6);
    fflush(stderr);
    _exit(0);          // (the library's process-wide caches hold stand-in streams whose threads nobody joins: leave like the CLIs do)
}
