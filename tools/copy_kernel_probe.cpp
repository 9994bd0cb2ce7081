// copy_kernel_probe - a copy kernel of our own between device memory and page-locked host memory (the GPU reads / writes the host pages over PCIe itself) against hipMemcpyAsync,
// whose SDMA engine is a lottery on some boxes (tools/d2h_probe.cpp: 9.7 or 18.2 ms for the same 0.55 GB).  Grid sizes 16..512 workgroups of 256 lanes, 16 bytes a lane and step.
// build: hipcc -O2 tools/copy_kernel_probe.cpp -o tools/copy_kernel_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_copy16(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	// four loads in flight per lane and round
	for (; i + 3 * stride < n16; i += 4 * stride) {
		const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
		dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
	}
	for (; i < n16; i += stride) dst[i] = src[i];
}
int main()
{
	const size_t n = 553303952 & ~(size_t)15;
	hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	void *d = nullptr, *h = nullptr;
	hipMalloc(&d, n); hipMemset(d, 5, n);
	hipHostMalloc(&h, n, hipHostMallocDefault);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	auto timeit = [&](const char *what, auto fn) {
		printf("%-60s", what);
		for (int rep = 0; rep < 5; ++rep) { hipEventRecord(e0, st); fn(); hipEventRecord(e1, st); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); printf(" %6.2f", ms); }
		printf(" ms\n"); fflush(stdout);
	};
	timeit("hipMemcpyAsync device -> host", [&] { hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st); });
	timeit("hipMemcpyAsync host -> device", [&] { hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st); });
	for (int wg : {16, 32, 64, 128, 256, 512, 1024}) {
		char name[96];
		snprintf(name, sizeof(name), "copy kernel, %4d workgroups, device -> host", wg);
		timeit(name, [&] { hipLaunchKernelGGL(k_copy16, dim3(wg), dim3(256), 0, st, (uint4 *)h, (const uint4 *)d, n / 16); });
		snprintf(name, sizeof(name), "copy kernel, %4d workgroups, host -> device", wg);
		timeit(name, [&] { hipLaunchKernelGGL(k_copy16, dim3(wg), dim3(256), 0, st, (uint4 *)d, (const uint4 *)h, n / 16); });
	}
	return 0;
}
