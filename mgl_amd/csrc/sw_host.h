// sw_host.h -- small host-side helpers shared by the front-ends of the one-pair entry (sw_batcher.cpp, sw_service.cpp).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include <sched.h>

namespace {

inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}

// CPUs this process may use: its affinity mask, cut by the cgroup's CPU quota (containers)
inline int usable_cpus()
{
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    long long quota = -1, period = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "max 100000" or "1600000 100000"
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { // v1
        if (fscanf(g, "%lld", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(h, "%lld", &period) != 1) period = 0;
            fclose(h);
        }
    }
    if (quota > 0 && period > 0) n = (int)std::min<long long>(n, std::max<long long>(1, quota / period));
    return std::max(1, n);
}

} // namespace
