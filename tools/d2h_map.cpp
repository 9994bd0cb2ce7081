// d2h_map - which device memory copies to the host at what rate?  tools/d2h_probe.cpp: a 0.55 GB device buffer copies in 9.7 ms or in 18.2 ms, the same every time for that
// allocation, whatever the host memory.  Here: 1 GB allocations until 256 GB are taken, each one's first 128 MB copied to page-locked host memory (D2H) and back (H2D), timed with
// events; one character per allocation in address order: F = PCIe rate (>= 50 GB/s), s = slow (< 40), m = in between.
// build: hipcc -O2 tools/d2h_map.cpp -o tools/d2h_map
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
int main()
{
	const size_t piece = (size_t)1 << 30, sample = (size_t)128 << 20;
	hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	void *h = nullptr; hipHostMalloc(&h, sample, hipHostMallocDefault);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	struct A { void *p; float d2h, h2d; };
	std::vector<A> a;
	for (int k = 0; k < 256; ++k) {
		void *p = nullptr;
		if (hipMalloc(&p, piece) != hipSuccess) { (void)hipGetLastError(); break; }
		hipMemsetAsync(p, 1, sample, st);
		float best_d = 1e9f, best_h = 1e9f;
		for (int rep = 0; rep < 3; ++rep) {
			float ms;
			hipEventRecord(e0, st); hipMemcpyAsync(h, p, sample, hipMemcpyDeviceToHost, st); hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); best_d = std::min(best_d, ms);
			hipEventRecord(e0, st); hipMemcpyAsync(p, h, sample, hipMemcpyHostToDevice, st); hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); best_h = std::min(best_h, ms);
		}
		a.push_back(A{p, best_d, best_h});
	}
	std::sort(a.begin(), a.end(), [](const A &x, const A &y) { return x.p < y.p; });
	auto gbs = [&](float ms) { return (double)sample / ms / 1e6; };
	printf("%zu allocations of 1 GB, in address order (%p .. %p)\nD2H: ", a.size(), a.front().p, a.back().p);
	for (auto &x : a) putchar(gbs(x.d2h) >= 50 ? 'F' : gbs(x.d2h) < 40 ? 's' : 'm');
	printf("\nH2D: ");
	for (auto &x : a) putchar(gbs(x.h2d) >= 50 ? 'F' : gbs(x.h2d) < 40 ? 's' : 'm');
	double dmin = 1e9, dmax = 0, hmin = 1e9, hmax = 0;
	for (auto &x : a) { dmin = std::min(dmin, gbs(x.d2h)); dmax = std::max(dmax, gbs(x.d2h)); hmin = std::min(hmin, gbs(x.h2d)); hmax = std::max(hmax, gbs(x.h2d)); }
	printf("\nD2H %.1f .. %.1f GB/s, H2D %.1f .. %.1f GB/s\n", dmin, dmax, hmin, hmax);
	// in allocation order too (what a process gets first)
	return 0;
}
