// How many host cores this process may really use: the smallest of the online count, the affinity mask and the cgroup CPU quota
// (a container given 16 CPUs of a 256-thread host by quota still reports 256 from std::thread::hardware_concurrency(); sizing the
// reader's and set_reads' thread pools by that number oversubscribes the quota and the scheduler throttles every thread).
// UVC_CPUS=n overrides.
#pragma once
#include <sched.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
static inline int uvc_effective_cpus() {
    static const int n = [] {
        if (const char *e = getenv("UVC_CPUS")) { const int v = atoi(e); if (v > 0) return v; }
        long long v = (long long)std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) v = std::min<long long>(v, c); }
        long long quota = -1, period = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota|max> <period>"
            char q[64] = { 0 };
            if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
            fclose(f);
        } else {   // cgroup v1
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 0; fclose(g); }
        }
        if (quota > 0 && period > 0) v = std::min(v, (quota + period - 1) / period);
        return (int)std::max<long long>(1, v);
    }();
    return n;
}
