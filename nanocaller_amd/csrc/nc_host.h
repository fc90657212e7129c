// Host threads this process may actually use: the hardware threads, capped by the scheduler affinity mask and by the
// cgroup CPU quota (cpu.max "quota period" of cgroup v2, cfs_quota_us / cfs_period_us of v1).  A container with 256
// visible CPUs and a quota of 16 runs 32 busy threads at half speed AND gets them throttled in bursts, so every host
// thread pool of the library (BGZF inflate, region decode, packer, VCF formatter) sizes itself with this.
#ifndef NC_HOST_H
#define NC_HOST_H

#include <sched.h>

#include <cstdio>
#include <cstdlib>
#include <thread>

static inline int nc_host_cpus()
{
    static int cached = 0;
    if (cached) return cached;
    int n = (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const int a = CPU_COUNT(&set);
        if (a >= 1 && a < n) n = a;
    }
    long long quota = -1, period = -1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm') quota = atoll(q);
        fclose(f);
    } else {
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = -1; fclose(g); }
    }
    if (quota > 0 && period > 0) {
        int c = (int)((quota + period - 1) / period);
        // one process per GPU: the ranks of a node draw from one quota (torchrun exports LOCAL_WORLD_SIZE); the affinity mask above is
        // already the rank's own share of its GPU's NUMA node (nanocaller_amd/numa.py)
        if (const char *lw = getenv("LOCAL_WORLD_SIZE")) {
            const int k = atoi(lw);
            if (k > 1) c = c / k > 1 ? c / k : 1;
        }
        if (c >= 1 && c < n) n = c;
    }
    cached = n;
    return n;
}

#endif
