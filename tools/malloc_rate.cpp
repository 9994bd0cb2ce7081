// malloc_rate - what does device memory cost to get and to give back?  (a command's first decode allocates ~8 GB in ~40 buffers: 0.2 s of a 1 s command)
// build: hipcc -O2 tools/malloc_rate.cpp -o tools/malloc_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	double t = now();
	hipFree(nullptr);
	printf("runtime start-up (first call): %.3f s\n", now() - t);
	hipStream_t st; t = now(); hipStreamCreateWithFlags(&st, hipStreamNonBlocking); printf("first stream: %.3f s\n", now() - t);
	for (int round = 0; round < 2; ++round) {
		for (size_t total_gb : {1, 8}) for (int pieces : {1, 8, 40}) {
			std::vector<void *> p((size_t)pieces, nullptr);
			const size_t each = (total_gb << 30) / (size_t)pieces;
			t = now();
			for (auto &q : p) if (hipMalloc(&q, each) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
			const double ta = now() - t;
			t = now();
			hipMemsetAsync(p[0], 0, each, st); hipStreamSynchronize(st);
			const double tm = now() - t;
			t = now();
			for (auto &q : p) hipFree(q);
			printf("round %d: %zu GB in %2d piece(s): hipMalloc %.4f s, first touch of one piece %.4f s, hipFree %.4f s\n", round, total_gb, pieces, ta, tm, now() - t);
		}
	}
	void *h = nullptr;
	for (size_t mb : {64, 512, 2048}) {
		t = now(); hipHostMalloc(&h, mb << 20, hipHostMallocDefault); const double ta = now() - t;
		t = now(); hipHostFree(h); printf("hipHostMalloc %4zu MB: %.4f s, hipHostFree %.4f s\n", mb, ta, now() - t);
	}
	return 0;
}
