// tile_sort.h - sorting a NEARLY sorted list of (key, value) pairs in one pass: every element counts, inside a window of +-WS_W places,
// the earlier elements that are greater and the later ones that are smaller, and moves by the difference.
//
// The right-clip ('3') events of a coordinate-sorted BAM come out in record order, i.e. by start position, while their bin key is the
// alignment END (start + reference span): the list is sorted up to displacements of a few places (events whose reads start within one read span
// of each other: ~3 places at 300x).  A full LSD radix sort of it (five passes over keys and values) was 0.33 ms of the 0.6 ms event sort; two
// passes of 2048-element bitonic sorts in LDS still 0.22 ms (66 barrier stages each).  If no element is WS_W or more places from its sorted
// position the ranks are exact; that is CHECKED, not assumed (every output slot written once, the result strictly increasing): long
// reference skips or pathological depth fall back to the radix sort.  Order is by (key, value): values are the events' ordinals, so equal keys
// keep BAM order like a stable sort.
#pragma once

#include "common.h"

namespace ssv {

constexpr int WS_W = 32;      // window: displacements below this are repaired (the pass is bound by its LDS reads: 2 W pairs per element)
constexpr int WS_TILE = 1024; // elements per workgroup (4 per thread)

__device__ __forceinline__ bool ts_greater(uint64_t ka, uint32_t va, uint64_t kb, uint32_t vb) { return ka > kb || (ka == kb && va > vb); }

__global__ __launch_bounds__(BLOCK) void k_window_rank_sort(const uint64_t *__restrict__ key_in, const uint32_t *__restrict__ val_in, uint64_t *__restrict__ key_out, uint32_t *__restrict__ val_out, int64_t n)
{
	__shared__ uint64_t sk[WS_TILE + 2 * WS_W];
	__shared__ uint32_t sv[WS_TILE + 2 * WS_W];
	const int64_t base = (int64_t)blockIdx.x * WS_TILE - WS_W; // list index of sk[0]
	for (int i = (int)threadIdx.x; i < WS_TILE + 2 * WS_W; i += BLOCK) {
		const int64_t g = base + i;
		const bool in = g >= 0 && g < n;
		sk[i] = in ? key_in[g] : 0ull;
		sv[i] = in ? val_in[g] : 0u;
	}
	__syncthreads();
	for (int e = (int)threadIdx.x; e < WS_TILE; e += BLOCK) {
		const int i = e + WS_W;
		const int64_t g = base + i;
		if (g >= n) break;
		const uint64_t k = sk[i];
		const uint32_t v = sv[i];
		int shift = 0;
		const int lo = g < WS_W ? WS_W - (int)g : 0;                       // first window slot that exists
		const int hi = n - g <= WS_W ? WS_W + (int)(n - 1 - g) : 2 * WS_W; // last one
#pragma unroll 8
		for (int j = lo; j < WS_W; ++j) shift -= ts_greater(sk[e + j], sv[e + j], k, v) ? 1 : 0;            // earlier and greater: I move down
#pragma unroll 8
		for (int j = WS_W + 1; j <= hi; ++j) shift += ts_greater(k, v, sk[e + j], sv[e + j]) ? 1 : 0;       // later and smaller: I move up
		key_out[g + shift] = k;
		val_out[g + shift] = v;
	}
}

// every slot written (val_out was filled with 0xffffffff before) and the list strictly increasing by (key, value)
__global__ __launch_bounds__(BLOCK) void k_check_sorted_pairs(const uint64_t *__restrict__ key, const uint32_t *__restrict__ val, int64_t n, int *__restrict__ flag)
{
	const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	if (val[i] == 0xffffffffu || (i + 1 < n && !ts_greater(key[i + 1], val[i + 1], key[i], val[i]))) *flag = 1;
}

// the result is in key_out / val_out; *flag is raised when it is not the sorted list after all
inline hipError_t sort_nearly_sorted(hipStream_t st, const uint64_t *key_in, const uint32_t *val_in, uint64_t *key_out, uint32_t *val_out, int64_t n, int *flag)
{
	hipError_t e = hipMemsetAsync(val_out, 0xff, (size_t)n * 4, st);
	if (e != hipSuccess) return e;
	k_window_rank_sort<<<(unsigned)((n + WS_TILE - 1) / WS_TILE), BLOCK, 0, st>>>(key_in, val_in, key_out, val_out, n);
	k_check_sorted_pairs<<<(unsigned)((n + BLOCK - 1) / BLOCK), BLOCK, 0, st>>>(key_out, val_out, n, flag);
	return hipGetLastError();
}

} // namespace ssv
