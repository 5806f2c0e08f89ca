// MF_COLD_TRACE=1: a timeline of a process's start-up on stderr, "[cold +seconds since the process was started] what".
// The reference calls this path a process at a time (utility/helper.py:78-86), so what a process does before its first useful
// kernel is paid on every call; tools/cold_calls.sh reads these lines.  Header-only; the CLIs and the library both use it.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>

namespace mf {
inline double cold_now() { timespec ts; clock_gettime(CLOCK_BOOTTIME, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
// when the kernel started this process (field 22 of /proc/self/stat, clock ticks since boot: 10 ms resolution)
inline double cold_process_start()
{
    static const double t0 = [] {
        double t = cold_now();
        if (FILE *f = fopen("/proc/self/stat", "r")) {
            char buf[2048]; const size_t n = fread(buf, 1, sizeof buf - 1, f); fclose(f); buf[n] = 0;
            const char *p = buf + n; while (p > buf && *p != ')') p--;          // (the command name may hold spaces: count fields behind it)
            unsigned long long start = 0; int field = 2;
            for (; *p; p++) if (*p == ' ' && ++field == 22) { start = strtoull(p + 1, nullptr, 10); break; }
            const long hz = sysconf(_SC_CLK_TCK);
            if (start && hz > 0 && (double)start / (double)hz <= t) t = (double)start / (double)hz;
        }
        return t;
    }();
    return t0;
}
inline bool cold_trace_on() { static const bool on = [] { const char *v = getenv("MF_COLD_TRACE"); return v && *v; }(); return on; }
inline void cold_mark(const char *what) { if (cold_trace_on()) fprintf(stderr, "[cold +%.3f] %s\n", cold_now() - cold_process_start(), what); }
} // namespace mf
