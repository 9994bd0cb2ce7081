// gather_bw.hip - what the memory system serves when a kernel touches lines at RANDOM or SPARSELY instead of streaming (reference ceiling for the per-candidate
// kernels k_clip_filter / k_getsv_cand - one 64-byte record line per candidate, ~1 % of the records, in increasing order - and for k_pack3_direct - one read's
// 225-byte entry per cluster, the entries ~20 KB apart).  hipcc --offload-arch=gfx950 -O3 tools/gather_bw.hip -o tools/gather_bw && tools/gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

// item i -> byte offset of its first byte.  mode 0: a random 64-byte line of the buffer; 1: line i * gap + (random below gap) - sparse, increasing; 2: byte
// i * gap * 64 + (random below gap * 64 - span) - sparse, increasing, unaligned (an entry of `span` bytes)
__device__ inline uint64_t place(int mode, uint64_t i, uint64_t n_lines, uint64_t gap, uint32_t span)
{
	const uint64_t h = mix(i + 0x9e3779b97f4a7c15ull);
	if (mode == 0) return (h % n_lines) * 64;
	if (mode == 1) return (i * gap + h % gap) * 64;
	return i * gap * 64 + h % (gap * 64 - span - 64);
}

// LPI lanes per item, each lane 16 bytes (the item's bytes from the 16-byte boundary below its first one on); DEPTH items of a lane group in flight
template <int LPI, int DEPTH>
__global__ __launch_bounds__(256) void k_gather(const uint8_t *__restrict__ p, uint64_t n_items, int mode, uint64_t n_lines, uint64_t gap, uint32_t span, unsigned *out)
{
	const uint64_t group = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / LPI, n_groups = (uint64_t)gridDim.x * 256 / LPI;
	const int gl = (int)(threadIdx.x % LPI);
	unsigned acc = 0;
	for (uint64_t i0 = group * DEPTH; i0 < n_items; i0 += n_groups * DEPTH) {
		u32x4 v[DEPTH];
#pragma unroll
		for (int k = 0; k < DEPTH; ++k) {
			v[k] = (u32x4){0, 0, 0, 0};
			if (i0 + k < n_items) {
				const uint64_t a = place(mode, i0 + k, n_lines, gap, span) & ~15ull;
				if ((uint64_t)gl * 16 < (uint64_t)span + 16) v[k] = *reinterpret_cast<const u32x4 *>(p + a + (uint64_t)gl * 16);
			}
		}
#pragma unroll
		for (int k = 0; k < DEPTH; ++k) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
	}
	if (acc == 0x12345678u) *out = acc;
}

template <int LPI, int DEPTH>
static void run(const char *what, const uint8_t *p, uint64_t n_items, int mode, uint64_t n_lines, uint64_t gap, uint32_t span, unsigned *out, int blocks, double lines_per_item)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	k_gather<LPI, DEPTH><<<blocks, 256>>>(p, n_items, mode, n_lines, gap, span, out);
	hipEventRecord(a);
	const int reps = 5;
	for (int r = 0; r < reps; ++r) k_gather<LPI, DEPTH><<<blocks, 256>>>(p, n_items, mode, n_lines, gap, span, out);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	ms /= reps;
	printf("%-44s lanes/item %2d depth %d blocks %5d : %7.3f ms  %6.2f G items/s  %6.2f G 64-byte lines/s  %7.1f GB/s of lines\n", what, LPI, DEPTH, blocks, ms, n_items / (ms * 1e-3) / 1e9,
	       n_items * lines_per_item / (ms * 1e-3) / 1e9, n_items * lines_per_item * 64 / (ms * 1e-3) / 1e9);
}

int main()
{
	const uint64_t bytes = 32ull << 30; // the decoded batch of the bench's 617 M records is ~40 GB
	uint8_t *p; unsigned *out;
	if (hipMalloc(&p, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
	hipMalloc(&out, 4);
	hipMemset(p, 1, bytes);
	const uint64_t n_lines = bytes / 64, n = 6400000; // as many items as a step of the bench has soft-clip candidates
	const uint64_t gap = n_lines / n;                 // ~80 lines between neighbours
	for (int blocks : {2048, 4096, 8192}) {
		run<4, 1>("random 64-byte lines", p, n, 0, n_lines, gap, 48, out, blocks, 1.0);
		run<4, 4>("random 64-byte lines", p, n, 0, n_lines, gap, 48, out, blocks, 1.0);
		run<4, 1>("sparse increasing 64-byte lines", p, n, 1, n_lines, gap, 48, out, blocks, 1.0);
		run<4, 4>("sparse increasing 64-byte lines", p, n, 1, n_lines, gap, 48, out, blocks, 1.0);
		run<16, 1>("sparse increasing 225-byte entries", p, n, 2, n_lines, gap, 225, out, blocks, (225 + 63) / 64.0);
		run<16, 2>("sparse increasing 225-byte entries", p, n, 2, n_lines, gap, 225, out, blocks, (225 + 63) / 64.0);
		run<16, 4>("sparse increasing 225-byte entries", p, n, 2, n_lines, gap, 225, out, blocks, (225 + 63) / 64.0);
	}
	return 0;
}
