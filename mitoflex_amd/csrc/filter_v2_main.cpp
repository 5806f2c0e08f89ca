// `filter_v2` drop-in for MitoFlex (installed as mitoflex_amd/filter/filter_v2, the path
// filter/filter.py:45,75 resolves).  Command line, validation, panic/usage exit codes and output
// bytes follow the reference (filter/filter_bin/src/main.rs:14-186, helper.rs:14-52; pinned by
// tests/golden/filter_v2_golden.json captured from the reference's ELF).  The filtering itself is
// mf_qualfilter_files of libmitofilter_hip.so: counting and hashing on the GPU, the order-dependent
// rules on the host.  No CPU fallback: without the library or a gfx950 device it exits non-zero.
#include "../../include/mitofilter.h"
#include "mf_coldtrace.h"

#include <dlfcn.h>
#include <time.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <map>
#include <string>

[[noreturn]] static void rust_panic(const char *msg) { fprintf(stderr, "thread 'main' panicked at '%s'\n", msg); exit(101); }
[[noreturn]] static void clap_error(const std::string &msg)
{
    fprintf(stderr, "error: %s\n\nUSAGE:\n    filter_v2 [FLAGS] [OPTIONS] --cleanq1 <CLEANQ1>\n\nFor more information try --help\n", msg.c_str());
    exit(1);
}

static void print_help()
{
    fputs("Fastq Filter 0.1\nJunyu Li, <2018301050@szu.edu.cn>\nFilter out unqualified fastq sequences\n\n"
          "USAGE:\n    filter_v2 [FLAGS] [OPTIONS] --cleanq1 <CLEANQ1>\n\n"
          "FLAGS:\n    -d, --deduplication    Filter out duplicated sequences\n    -h, --help             Prints help information\n"
          "        --truncate_only    Only truncates the file, no filtering.\n    -V, --version          Prints version information\n\n"
          "OPTIONS:\n"
          "    -3, --cleanq1 <CLEANQ1>    Output clean fastq file 1\n    -4, --cleanq2 <CLEANQ2>    Output clean fastq file 2\n"
          "    -e, --end <INT>            Cut suquences's end [default: 0]\n    -1, --fastq1 <FASTQ1>      Input raw data fastq file 1\n"
          "    -2, --fastq2 <FASTQ2>      Input raw data fastq file 2\n"
          "    -l, --limit <FLOAT>        Sequences will be filtered out if bad bases > limit * length [default: 0.2]\n"
          "    -n, --nvalues <INT>        Sequences having Ns more than this will be filtered out [default: 10]\n"
          "    -q, --quality <INT>        Quality under this will be considered as bad base [default: 55]\n"
          "    -s, --start <INT>          Cut sequence's start [default: 0]\n"
          "    -t, --trim <INT>           Only this bases of sequences will be filtered out [default: 0]\n", stdout);
}

// str::parse::<usize>
static bool parse_usize(const std::string &s, uint64_t &out)
{
    size_t i = 0; if (i < s.size() && s[i] == '+') i++;
    if (i >= s.size()) return false;
    uint64_t v = 0;
    for (; i < s.size(); i++) {
        if (s[i] < '0' || s[i] > '9') return false;
        const uint64_t d = (uint64_t)(s[i] - '0');
        if (v > (UINT64_MAX - d) / 10) return false;
        v = v * 10 + d;
    }
    out = v; return true;
}
// str::parse::<f32> as the shipped toolchain does it: [sign] "inf" | "NaN" | decimal [exponent]
static bool parse_f32(const std::string &s, float &out)
{
    size_t i = 0, n = s.size(); bool neg = false;
    if (i < n && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; i++; }
    const std::string body = s.substr(i);
    if (body == "inf") { out = neg ? -INFINITY : INFINITY; return true; }
    if (body == "NaN") { out = NAN; return true; }
    size_t j = i, nd = 0;
    while (j < n && isdigit((unsigned char)s[j])) { j++; nd++; }
    if (j < n && s[j] == '.') { j++; while (j < n && isdigit((unsigned char)s[j])) { j++; nd++; } }
    if (nd == 0) return false;
    if (j < n && (s[j] == 'e' || s[j] == 'E')) {
        j++; if (j < n && (s[j] == '+' || s[j] == '-')) j++;
        size_t ne = 0; while (j < n && isdigit((unsigned char)s[j])) { j++; ne++; }
        if (ne == 0) return false;
    }
    if (j != n) return false;
    out = strtof(s.c_str(), nullptr);
    return true;
}

static std::string exe_dir()
{
    char buf[PATH_MAX]; ssize_t n = readlink("/proc/self/exe", buf, sizeof buf - 1);
    if (n <= 0) return ".";
    buf[n] = 0; std::string p(buf); size_t s = p.rfind('/');
    return s == std::string::npos ? "." : p.substr(0, s);
}

int main(int argc, char **argv)
{
    static const std::map<std::string, char> LONG = {{"fastq1", '1'}, {"fastq2", '2'}, {"cleanq1", '3'}, {"cleanq2", '4'}, {"start", 's'},
                                                     {"end", 'e'}, {"quality", 'q'}, {"limit", 'l'}, {"nvalues", 'n'}, {"trim", 't'},
                                                     {"deduplication", 'd'}};
    std::map<char, std::string> opt;
    for (int a = 1; a < argc; a++) {
        std::string arg = argv[a];
        if (arg == "-h" || arg == "--help") { print_help(); return 0; }
        if (arg == "-V" || arg == "--version") { puts("Fastq Filter 0.1"); return 0; }
        char key = 0; std::string val; bool have_val = false;
        if (arg == "--truncate_only") { key = 'T'; }
        else if (arg.rfind("--", 0) == 0) {
            const size_t eq = arg.find('=');
            const std::string name = arg.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            auto it = LONG.find(name);
            if (it == LONG.end()) clap_error("Found argument '" + arg + "' which wasn't expected, or isn't valid in this context");
            key = it->second;
            if (key != 'd') {
                if (eq != std::string::npos) { val = arg.substr(eq + 1); have_val = true; }
                else { if (a + 1 >= argc) clap_error("The argument '--" + name + "' requires a value but none was supplied"); val = argv[++a]; have_val = true; }
            }
        } else if (arg.size() >= 2 && arg[0] == '-') {
            key = arg[1];
            if (key == 'd') { if (arg.size() > 2) clap_error("Found argument '" + arg + "' which wasn't expected, or isn't valid in this context"); }
            else if (strchr("1234seqlnt", key)) {
                if (arg.size() > 2) { val = arg.substr(2); if (val[0] == '=') val = val.substr(1); have_val = true; }
                else {
                    if (a + 1 >= argc) clap_error(std::string("The argument '-") + key + "' requires a value but none was supplied");
                    val = argv[++a]; have_val = true;
                    if (val.size() > 1 && val[0] == '-') clap_error("Found argument '" + val + "' which wasn't expected, or isn't valid in this context");
                }
            } else clap_error("Found argument '" + arg + "' which wasn't expected, or isn't valid in this context");
        } else clap_error("Found argument '" + arg + "' which wasn't expected, or isn't valid in this context");
        if (opt.count(key)) clap_error(std::string("The argument '") + arg + "' was provided more than once, but cannot be used multiple times");
        opt[key] = have_val ? val : "";
    }
    if (!opt.count('3')) clap_error("The following required arguments were not provided:\n    --cleanq1 <CLEANQ1>");
    if (opt.count('4') && !opt.count('2')) clap_error("The following required arguments were not provided:\n    --fastq2 <FASTQ2>");
    if (opt.count('d') && !opt.count('2')) clap_error("The following required arguments were not provided:\n    --fastq2 <FASTQ2>");

    auto get = [&](char k, const char *dflt) { return opt.count(k) ? opt[k] : std::string(dflt); };
    // main.rs:124-175, in the reference's order
    uint64_t start, end, ns, trim, q64; float limit;
    if (!parse_usize(get('s', "0"), start)) rust_panic("Cannot parse start position!");
    if (!parse_usize(get('e', "0"), end)) rust_panic("Cannot parse end position!");
    if (start > end) rust_panic("Start position comes after the end!");
    if (!parse_usize(get('q', "55"), q64) || q64 > 255) rust_panic("Canoot parse quality value!");
    if (q64 == 0 || q64 > 100) rust_panic("Wrong quality number!");
    if (!parse_f32(get('l', "0.2"), limit)) rust_panic("Cannot parse limit value!");
    if (limit <= 0.0f || limit >= 1.0f) rust_panic("Wrong percentage value!");
    if (!parse_usize(get('n', "10"), ns)) rust_panic("Cannot parse N value!");
    if (!parse_usize(get('t', "0"), trim)) rust_panic("Cannot parse a positive int to trimming!");

    const char *fq1 = opt.count('1') ? opt['1'].c_str() : nullptr;
    const char *fq2 = opt.count('2') ? opt['2'].c_str() : nullptr;
    const char *out2 = opt.count('4') ? opt['4'].c_str() : nullptr;
    // the reference opens its inputs first and panics when one is missing (helper.rs:17-20)
    for (const char *p : {fq1, fq2}) if (p) { FILE *f = fopen(p, "rb"); if (!f) rust_panic("Cannot open file"); fclose(f); }

    const char *libenv = getenv("MITOFILTER_LIB");
    const std::string libpath = libenv ? libenv : exe_dir() + "/../libmitofilter_hip.so";
    const bool timing = getenv("MF_PIPE_TIMING") != nullptr;
    auto now = [] { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
    const double t_load = now();
    mf::cold_mark("filter_v2: arguments read");
    void *h = dlopen(libpath.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "error: cannot load %s: %s (filter_v2 has no CPU fallback)\n", libpath.c_str(), dlerror()); return 2; }
    auto p_run = (decltype(&mf_qualfilter_files))dlsym(h, "mf_qualfilter_files");
    auto p_err = (decltype(&mf_last_error))dlsym(h, "mf_last_error");
    if (!p_run || !p_err) { fprintf(stderr, "error: %s lacks mf_qualfilter_files\n", libpath.c_str()); return 2; }
    if (auto p_opt = (decltype(&mf_set_option))dlsym(h, "mf_set_option")) { (void)p_opt("expect_files", "1"); (void)p_opt("short_lived", "1"); }
    uint64_t kept = 0, total = 0; int panicked = 0;
    setenv("MF_DEVPOOL_GB", "4096", 0);          // (a process that ends with the call gives no device memory back in between: the runtime frees it all at once)
    const double t_call = now();
    mf::cold_mark("library loaded");
    const int rc = p_run(fq1, fq2, opt['3'].c_str(), out2, start, end, ns, (uint32_t)q64, limit, opt.count('d') ? 1 : 0, trim,
                         opt.count('T') ? 1 : 0, 0, &kept, &total, &panicked);
    mf::cold_mark("files filtered");
    if (timing) fprintf(stderr, "[filter_v2] loading the library %.3f s, the call %.3f s\n", t_call - t_load, now() - t_call);
    if (rc != MF_OK) { fprintf(stderr, "error: %s\n", p_err()); return 3; }
    if (panicked) rust_panic("called `Result::unwrap()` on an `Err` value / drain out of range");
    // (the outputs are written and closed; what is left is the GPU runtime's teardown -- queues, code objects, a tenth of a second -- which a
    // process that is about to be gone has no use for)
    fflush(stdout); fflush(stderr);
    const char *pre = getenv("LD_PRELOAD");
    if (!((pre && strstr(pre, "rocprof")) || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") || getenv("ROCP_TOOL_LIBRARIES"))) _exit(0);      // (a profiler writes its files in a finaliser)
    return 0;
}
