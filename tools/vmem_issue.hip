// vmem_issue.hip - what a CU's vector memory path charges per INSTRUCTION: loads / stores with K of 64 lanes active, 4 / 8 / 16 bytes a lane, every active lane its own
// line of a buffer that stays in the L2 (16 MB over the chip) - pass 2 of the device inflate (resolve_wave.h) issues ~40 such instructions per round of 64 tokens, most of
// them with a handful of lanes active.  hipcc --offload-arch=gfx950 -O3 tools/vmem_issue.hip -o tools/vmem_issue && tools/vmem_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int BYTES, bool STORE>
__global__ __launch_bounds__(256) void k_issue(uint8_t *buf, uint32_t mask_lines, int active, int iters, unsigned *out)
{
	const int lane = threadIdx.x & 63;
	const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
	uint32_t h = wave * 2654435761u + lane * 40503u + 12345u;
	unsigned acc = 0;
	if (lane < active) {
		for (int i = 0; i < iters; i += 8) {
			uint8_t *p[8];
#pragma unroll
			for (int k = 0; k < 8; ++k) { h = h * 1664525u + 1013904223u; p[k] = buf + (size_t)((h >> 8) & mask_lines) * 64 + 3; } // its own line, not aligned (like a match's bytes)
			if (STORE) {
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					if (BYTES == 4) asm volatile("global_store_dword %0, %1, off" : : "v"(p[k]), "v"(h) : "memory");
					else if (BYTES == 8) { u32x2 v = {h, h}; asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(p[k]), "v"(v) : "memory"); }
					else { u32x4 v = {h, h, h, h}; asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p[k]), "v"(v) : "memory"); }
				}
			} else {
				if (BYTES == 4) { uint32_t v[8];
#pragma unroll
					for (int k = 0; k < 8; ++k) __builtin_memcpy(&v[k], p[k], 4);
#pragma unroll
					for (int k = 0; k < 8; ++k) acc += v[k]; }
				else if (BYTES == 8) { u32x2 v[8];
#pragma unroll
					for (int k = 0; k < 8; ++k) __builtin_memcpy(&v[k], p[k], 8);
#pragma unroll
					for (int k = 0; k < 8; ++k) acc += v[k].x ^ v[k].y; }
				else { u32x4 v[8];
#pragma unroll
					for (int k = 0; k < 8; ++k) __builtin_memcpy(&v[k], p[k], 16);
#pragma unroll
					for (int k = 0; k < 8; ++k) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w; }
			}
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	if (acc == 0x12345678u) *out = acc;
}

template <int BYTES, bool STORE>
static void run(uint8_t *buf, uint32_t mask_lines, int active, int waves_per_cu, unsigned *out)
{
	const int iters = 2000, blocks = 256 * waves_per_cu / 4;
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	k_issue<BYTES, STORE><<<blocks, 256>>>(buf, mask_lines, active, 200, out);
	hipEventRecord(a);
	k_issue<BYTES, STORE><<<blocks, 256>>>(buf, mask_lines, active, iters, out);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	const double instr = (double)blocks * 4 * iters;
	printf("%s %2d bytes a lane, %2d of 64 lanes active, %2d wavefronts a CU: %6.1f ns per instruction and CU = %5.1f clocks at 2.4 GHz; %6.2f G lane accesses/s\n", STORE ? "store" : "load ", BYTES, active,
	       waves_per_cu, ms * 1e6 / (instr / 256), ms * 1e6 / (instr / 256) * 2.4, instr * active / (ms * 1e-3) / 1e9);
}

int main(int argc, char **argv)
{
	uint8_t *buf; unsigned *out;
	// buffer size in KB (default 16 MB: larger than one XCD's 4 MB L2, inside the Infinity Cache; 1024: every XCD's L2 holds it; 16: a CU's L1 does)
	const size_t bytes = (size_t)(argc > 1 ? atol(argv[1]) : 16384) << 10;
	hipMalloc(&buf, bytes + 4096); hipMalloc(&out, 4);
	hipMemset(buf, 1, bytes + 4096);
	const uint32_t mask_lines = (uint32_t)(bytes / 64 - 1);
	printf("buffer %zu KB\n", bytes >> 10);
	for (int wpc : {8, 32})
		for (int active : {1, 4, 16, 64}) {
			run<4, false>(buf, mask_lines, active, wpc, out);
			run<8, false>(buf, mask_lines, active, wpc, out);
			run<16, false>(buf, mask_lines, active, wpc, out);
			run<4, true>(buf, mask_lines, active, wpc, out);
			run<16, true>(buf, mask_lines, active, wpc, out);
		}
	return 0;
}
