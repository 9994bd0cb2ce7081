// scan.h - device-wide exclusive prefix sum (reduce / scan-of-sums / downsweep), order preserving.
// Used for ordered compaction (tile counts -> output offsets) and for variable-size blob offsets.
#pragma once

#include "common.h"

namespace ssv {

constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = BLOCK * SCAN_ITEMS; // 2048 elements per workgroup

template <typename TIn, typename TOut>
__global__ __launch_bounds__(BLOCK) void k_scan_reduce(const TIn *__restrict__ in, int64_t n, TOut *__restrict__ block_sums)
{
	__shared__ TOut lds[WAVES_PER_BLOCK];
	int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
	TOut s = 0;
#pragma unroll
	for (int k = 0; k < SCAN_ITEMS; ++k) {
		int64_t i = base + (int64_t)k * BLOCK + threadIdx.x;
		if (i < n) s += (TOut)in[i];
	}
	s = wave_sum(s);
	if (lane_id() == 0) lds[wave_id()] = s;
	__syncthreads();
	if (threadIdx.x == 0) {
		TOut t = 0;
		for (int w = 0; w < WAVES_PER_BLOCK; ++w) t += lds[w];
		block_sums[blockIdx.x] = t;
	}
}

// one workgroup: exclusive scan of sums[0..m) in place; *total = carry_in + sum of everything.  Every thread owns a contiguous run of
// ceil(m / BLOCK) sums, so the whole array takes one block-level scan (two barriers) whatever m is - this kernel runs once per
// device-wide scan, ~25 times per pass of the path, and a loop of 256-element rounds made it the longest part of the small scans.
template <typename TOut>
__global__ __launch_bounds__(BLOCK) void k_scan_sums(TOut *__restrict__ sums, int64_t m, TOut carry_in, TOut *__restrict__ total)
{
	__shared__ TOut lds[WAVES_PER_BLOCK + 1];
	const int64_t per = (m + BLOCK - 1) / BLOCK;
	const int64_t lo = (int64_t)threadIdx.x * per, hi = lo + per < m ? lo + per : m;
	TOut s = 0;
	for (int64_t i = lo; i < hi; ++i) s += sums[i];
	TOut tot;
	TOut ex = carry_in + block_exclusive_sum(s, lds, &tot);
	for (int64_t i = lo; i < hi; ++i) { const TOut v = sums[i]; sums[i] = ex; ex += v; }
	if (threadIdx.x == 0 && total) *total = carry_in + tot;
}

// out[i] = block_sums[block] + exclusive prefix of in within the block (block_sums already include carry_in)
template <typename TIn, typename TOut>
__global__ __launch_bounds__(BLOCK) void k_scan_down(const TIn *__restrict__ in, int64_t n, const TOut *__restrict__ block_sums, TOut *__restrict__ out)
{
	__shared__ TOut lds[WAVES_PER_BLOCK + 1];
	// blocked arrangement: thread t owns elements base + t*ITEMS .. +ITEMS so that order is preserved
	int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
	TOut v[SCAN_ITEMS];
	TOut s = 0;
#pragma unroll
	for (int k = 0; k < SCAN_ITEMS; ++k) {
		v[k] = (base + k < n) ? (TOut)in[base + k] : (TOut)0;
		s += v[k];
	}
	TOut tot;
	TOut ex = block_exclusive_sum(s, lds, &tot) + block_sums[blockIdx.x];
#pragma unroll
	for (int k = 0; k < SCAN_ITEMS; ++k) {
		if (base + k < n) out[base + k] = ex;
		ex += v[k];
	}
}

// Host driver.  block_sums: scratch of at least scan_scratch_elems(n) TOut elements.  total (device pointer, may be null)
// receives carry_in + sum(in).  in == out is allowed when TIn == TOut.
static inline int64_t scan_scratch_elems(int64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE + 1; }

template <typename TIn, typename TOut>
static inline void exclusive_scan(hipStream_t st, const TIn *in, TOut *out, int64_t n, TOut carry_in, TOut *block_sums, TOut *total)
{
	int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
	if (nb == 0) {
		k_scan_sums<TOut><<<1, BLOCK, 0, st>>>(block_sums, 0, carry_in, total);
		return;
	}
	k_scan_reduce<TIn, TOut><<<(unsigned)nb, BLOCK, 0, st>>>(in, n, block_sums);
	k_scan_sums<TOut><<<1, BLOCK, 0, st>>>(block_sums, nb, carry_in, total);
	k_scan_down<TIn, TOut><<<(unsigned)nb, BLOCK, 0, st>>>(in, n, block_sums, out);
}

} // namespace ssv
