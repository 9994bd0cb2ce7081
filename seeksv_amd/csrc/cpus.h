// cpus.h - how many CPUs this process can really use: the visible ones, cut down to the affinity mask and to the cgroup's CPU quota.
// A container may see 256 CPUs and be granted the time of 16 (cpu.max "1600000 100000"): a pool of 64 or 256 threads then spends its
// quota in the first milliseconds of every period and sits throttled for the rest.
#pragma once

#include <sched.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace ssv {

inline int effective_cpus()
{
	static const int cached = [] {
		int n = (int)std::thread::hardware_concurrency();
		if (n < 1) n = 1;
		cpu_set_t set;
		if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0 && a < n) n = a; }
		if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "<quota|max> <period>"
			char q[32]; long long period = 0;
			if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) { const long long c = atoll(q) / period; if (c >= 1 && c < n) n = (int)c; }
			fclose(f);
		} else {
			long long q = -1, p = 0;
			if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &q) != 1) q = -1; fclose(g); }
			if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &p) != 1) p = 0; fclose(g); }
			if (q > 0 && p > 0 && q / p >= 1 && q / p < n) n = (int)(q / p);
		}
		return n;
	}();
	return cached;
}

} // namespace ssv
