// d2h_probe - how long does a 0.55 GB device-to-host copy take, and what does it depend on?  (bench.py: the same cluster table crossed PCIe in 6.5, 9.7 or 13.8-15.3 ms
// depending on the leg and the box.)  Host memory from hipHostMalloc, or anonymous huge pages registered with the runtime - on the GPU's NUMA node, on the other one, wherever
// the touching threads run -; the device buffer fresh, or allocated after 48 GB were allocated, written and freed.  HIP events around each copy, five in a row.
// build: hipcc -O2 tools/d2h_probe.cpp -o tools/d2h_probe
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static int gpu_node()
{
	char bdf[64] = {0};
	if (hipDeviceGetPCIBusId(bdf, 64, 0) != hipSuccess) return -1;
	for (char *q = bdf; *q; ++q) *q = (char)tolower((unsigned char)*q);
	int node = -1;
	if (FILE *f = fopen((std::string("/sys/bus/pci/devices/") + bdf + "/numa_node").c_str(), "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
	return node;
}

static void *huge(size_t n, int node)
{
	const size_t H = (size_t)2 << 20;
	uint8_t *m = static_cast<uint8_t *>(mmap(nullptr, n + H, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
	uint8_t *a = reinterpret_cast<uint8_t *>(((uintptr_t)m + H - 1) & ~(uintptr_t)(H - 1));
	madvise(a, n, MADV_HUGEPAGE);
	long rc = 0;
	if (node >= 0) { unsigned long mask[16] = {0}; mask[0] = 1ul << node; rc = syscall(SYS_mbind, a, n, 2 /* MPOL_BIND */, mask, 8 * sizeof(mask), 0u); }
	std::vector<std::thread> th;
	for (int t = 0; t < 8; ++t) th.emplace_back([=] { for (size_t o = n / 8 * (size_t)t; o < n / 8 * (size_t)(t + 1); o += 4096) a[o] = 1; });
	for (auto &x : th) x.join();
	if (hipHostRegister(a, n, hipHostRegisterPortable) != hipSuccess) { printf("register failed\n"); return nullptr; }
	if (rc != 0) printf("   (mbind to node %d refused: errno %ld)\n", node, rc);
	return a;
}

static void run(const char *what, void *h, void *d, size_t n, hipStream_t st)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	printf("%-78s [dev %p]", what, d);
	for (int rep = 0; rep < 5; ++rep) {
		hipEventRecord(e0, st);
		hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st);
		hipEventRecord(e1, st);
		hipEventSynchronize(e1);
		float ms = 0; hipEventElapsedTime(&ms, e0, e1);
		printf(" %6.2f", ms);
	}
	printf(" ms\n");
	fflush(stdout);
}

int main()
{
	const size_t n = 553303952;
	hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	const int node = gpu_node();
	printf("GPU 0 hangs on NUMA node %d\n", node);
	void *d = nullptr; hipMalloc(&d, n + 4096); hipMemset(d, 7, n);
	void *hm = nullptr; hipHostMalloc(&hm, n, hipHostMallocDefault);
	void *near = huge(n, node), *far = huge(n, node < 0 ? -1 : 1 - node), *any = huge(n, -1);
	for (int round = 0; round < 2; ++round) {
		run("hipHostMalloc", hm, d, n, st);
		if (near) run("huge pages, registered, bound to the GPU's node", near, d, n, st);
		if (far) run("huge pages, registered, bound to the OTHER node", far, d, n, st);
		if (any) run("huge pages, registered, unbound (wherever the touching threads ran)", any, d, n, st);
		if (round == 0) {
			// what bench.py's second leg does to the device side: a big sample leaves, another comes, the table's device buffer is allocated anew
			void *big = nullptr; hipMalloc(&big, (size_t)48 << 30); hipMemset(big, 1, (size_t)48 << 30); hipDeviceSynchronize(); hipFree(big);
			hipFree(d); hipMalloc(&big, (size_t)48 << 30); hipMalloc(&d, n + (200 << 20)); hipMemset(d, 9, n); hipDeviceSynchronize();
			printf("-- the device buffer allocated anew, behind 48 GB that stay allocated --\n");
		}
	}
	// the library's way: nine pieces (the table's columns and its string block) on a SECOND stream, behind an event of the first; host wall clock like ssv_clip_table_wait sees it
	{
		hipStream_t st2; hipStreamCreateWithFlags(&st2, hipStreamNonBlocking);
		hipEvent_t packed, copied; hipEventCreateWithFlags(&packed, hipEventDisableTiming); hipEventCreateWithFlags(&copied, hipEventDisableTiming);
		const size_t piece[9] = {22092000, 22092000, 11046000, 5523000, 5523000, 464930560, 26000000, 3200, 0};
		std::vector<void *> hp(9, nullptr), dp(9, nullptr);
		for (int k = 0; k < 9; ++k) if (piece[k]) { hp[k] = huge(piece[k] < ((size_t)2 << 20) ? ((size_t)2 << 20) : piece[k], node); hipMalloc(&dp[k], piece[k] + 4096); hipMemset(dp[k], k, piece[k]); }
		hipDeviceSynchronize();
		for (int variant = 0; variant < 3; ++variant) {
			printf("%-78s", variant == 0 ? "nine pieces on a second stream behind an event" : variant == 1 ? "the same, the big piece first" : "the same as ONE piece of the same total");
			for (int rep = 0; rep < 10; ++rep) {
				hipMemsetAsync(d, rep, 1 << 20, st);
				hipEventRecord(packed, st);
				hipStreamSynchronize(st);
				hipStreamWaitEvent(st2, packed, 0);
				const auto t0 = std::chrono::steady_clock::now();
				if (variant == 2) hipMemcpyAsync(any, d, n, hipMemcpyDeviceToHost, st2);
				else for (int q = 0; q < 9; ++q) { const int k = variant == 1 ? (q == 0 ? 5 : q <= 5 ? q - 1 : q) : q; if (piece[k]) hipMemcpyAsync(hp[k], dp[k], piece[k], hipMemcpyDeviceToHost, st2); }
				hipEventRecord(copied, st2);
				hipEventSynchronize(copied);
				printf(" %6.2f", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
			}
			printf(" ms\n");
			fflush(stdout);
		}
	}
	return 0;
}
