// read_bw.hip - what a read-only streaming kernel can pull from HBM on this GPU (reference ceiling for the scan kernels).
// hipcc --offload-arch=gfx950 -O3 tools/read_bw.hip -o /tmp/read_bw && /tmp/read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_read(const u32x4 *__restrict__ p, size_t n16, unsigned *out)
{
	unsigned acc = 0;
	const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
	for (size_t base = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; base < n16; base += stride) {
		u32x4 v[UNROLL];
#pragma unroll
		for (int k = 0; k < UNROLL; ++k) {
			size_t i = base + (size_t)k * 256;
			if (i < n16) v[k] = NT ? __builtin_nontemporal_load(p + i) : p[i]; else v[k] = (u32x4){0, 0, 0, 0};
		}
#pragma unroll
		for (int k = 0; k < UNROLL; ++k) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
	}
	if (acc == 0x12345678u) *out = acc;
}

template <int UNROLL, bool NT>
void run(const u32x4 *p, size_t n16, unsigned *out, int blocks)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	for (int w = 0; w < 2; ++w) k_read<UNROLL, NT><<<blocks, 256>>>(p, n16, out);
	hipEventRecord(a);
	const int reps = 10;
	for (int r = 0; r < reps; ++r) k_read<UNROLL, NT><<<blocks, 256>>>(p, n16, out);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	printf("unroll %d nt %d blocks %5d : %7.3f ms  %7.1f GB/s\n", UNROLL, (int)NT, blocks, ms / reps, (double)n16 * 16 / (ms / reps * 1e-3) / 1e9);
}

int main()
{
	const size_t bytes = 4ull << 30; // 4 GiB, far beyond the 256 MiB Infinity Cache
	u32x4 *p; unsigned *out;
	hipMalloc(&p, bytes); hipMalloc(&out, 4);
	hipMemset(p, 1, bytes);
	const size_t n16 = bytes / 16;
	for (int blocks : {1024, 1536, 2048, 4096, 16384}) {
		run<4, false>(p, n16, out, blocks);
		run<4, true>(p, n16, out, blocks);
		run<8, true>(p, n16, out, blocks);
		run<16, true>(p, n16, out, blocks);
	}
	return 0;
}
