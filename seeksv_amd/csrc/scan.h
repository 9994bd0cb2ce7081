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

// one workgroup: exclusive scan of sums[0..m) in place; *total = carry_in + sum of everything.  Chunks of 4096 sums go through LDS:
// loaded with coalesced, independent loads (one memory round trip per chunk), every thread then walks its own run of 16 in LDS (runs are
// padded by one slot so that the walks of a wavefront's lanes fall into different banks), one block-level scan, coalesced write-back.
// This kernel runs once per device-wide scan, ~25 times per pass of the path: walking the runs in global memory (a dependent load per
// element) took 13-40 us per call, 0.2 ms per pass.
template <typename TOut>
__device__ __forceinline__ void scan_sums_body(TOut *__restrict__ sums, int64_t m, TOut carry_in, TOut *__restrict__ total)
{
	constexpr int PER = 16, CH = BLOCK * PER;
	__shared__ TOut buf[CH + BLOCK];
	__shared__ TOut lds[WAVES_PER_BLOCK + 1];
	TOut carry = carry_in;
	// (the next chunk's sums are asked for while this one is scanned: a chunk was one trip to memory plus three barriers, 5 us, and a step's 25 K tile sums are seven chunks)
	TOut cur[PER];
#pragma unroll
	for (int k = 0; k < PER; ++k) { const int64_t i = (int64_t)k * BLOCK + threadIdx.x; cur[k] = i < m ? sums[i] : (TOut)0; }
	for (int64_t base = 0; base < m; base += CH) {
		const int64_t n = m - base < CH ? m - base : CH;
#pragma unroll
		for (int k = 0; k < PER; ++k) {
			const int i = k * BLOCK + (int)threadIdx.x;
			buf[i + i / PER] = cur[k];
		}
#pragma unroll
		for (int k = 0; k < PER; ++k) { const int64_t i = base + CH + (int64_t)k * BLOCK + threadIdx.x; cur[k] = i < m ? sums[i] : (TOut)0; }
		lds_barrier(); // (not __syncthreads(): that one also waits for the loads just issued - common.h)
		const int run = (int)threadIdx.x * (PER + 1);
		TOut s = 0;
#pragma unroll
		for (int j = 0; j < PER; ++j) s += buf[run + j];
		// block_exclusive_sum with LDS-only barriers
		const TOut inc = wave_inclusive_sum(s);
		if (lane_id() == 63) lds[wave_id()] = inc;
		lds_barrier();
		TOut before = 0, tot = 0;
#pragma unroll
		for (int w = 0; w < WAVES_PER_BLOCK; ++w) { const TOut x = lds[w]; if (w < wave_id()) before += x; tot += x; }
		TOut ex = carry + before + inc - s;
#pragma unroll
		for (int j = 0; j < PER; ++j) { const TOut v = buf[run + j]; buf[run + j] = ex; ex += v; }
		lds_barrier();
#pragma unroll
		for (int k = 0; k < PER; ++k) {
			const int i = k * BLOCK + (int)threadIdx.x;
			if (i < n) sums[base + i] = buf[i + i / PER];
		}
		carry += tot;
		lds_barrier(); // (buf and lds[] are written again)
	}
	if (threadIdx.x == 0 && total) *total = carry;
}

template <typename TOut>
__global__ __launch_bounds__(BLOCK) void k_scan_sums(TOut *__restrict__ sums, int64_t m, TOut carry_in, TOut *__restrict__ total) { scan_sums_body(sums, m, carry_in, total); }

// several independent lists of sums, `stride` elements apart: one workgroup each
template <typename TOut>
__global__ __launch_bounds__(BLOCK) void k_scan_sums_lists(TOut *__restrict__ sums, int64_t m, int64_t stride) { scan_sums_body<TOut>(sums + (int64_t)blockIdx.x * stride, m, (TOut)0, nullptr); }

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

// The same without the middle kernel, for up to SCAN_DIRECT_TILES tiles: every workgroup adds up the raw sums of the tiles before it
// itself (<= 16 coalesced loads per thread, one block-level reduction) - a few microseconds in parallel instead of a dependent launch of
// one workgroup.  The last workgroup also writes the total.
constexpr int64_t SCAN_DIRECT_TILES = 4096;

template <typename TIn, typename TOut>
__global__ __launch_bounds__(BLOCK) void k_scan_down_direct(const TIn *__restrict__ in, int64_t n, const TOut *__restrict__ raw_sums, TOut carry_in, TOut *__restrict__ out, TOut *__restrict__ total)
{
	__shared__ TOut lds[WAVES_PER_BLOCK + 1];
	__shared__ TOut lds2[WAVES_PER_BLOCK];
	TOut before = 0;
	for (int64_t i = threadIdx.x; i < (int64_t)blockIdx.x; i += BLOCK) before += raw_sums[i];
	before = wave_sum(before);
	if (lane_id() == 0) lds2[wave_id()] = before;
	int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
	TOut v[SCAN_ITEMS];
	TOut s = 0;
#pragma unroll
	for (int k = 0; k < SCAN_ITEMS; ++k) {
		v[k] = (base + k < n) ? (TOut)in[base + k] : (TOut)0;
		s += v[k];
	}
	TOut tot;
	TOut ex = block_exclusive_sum(s, lds, &tot); // (its barriers also publish lds2)
	TOut pre = carry_in;
#pragma unroll
	for (int w = 0; w < WAVES_PER_BLOCK; ++w) pre += lds2[w];
	ex += pre;
#pragma unroll
	for (int k = 0; k < SCAN_ITEMS; ++k) {
		if (base + k < n) out[base + k] = ex;
		ex += v[k];
	}
	if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = pre + tot;
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
	if (nb <= SCAN_DIRECT_TILES) {
		k_scan_down_direct<TIn, TOut><<<(unsigned)nb, BLOCK, 0, st>>>(in, n, block_sums, carry_in, out, total);
		return;
	}
	k_scan_sums<TOut><<<1, BLOCK, 0, st>>>(block_sums, nb, carry_in, total);
	k_scan_down<TIn, TOut><<<(unsigned)nb, BLOCK, 0, st>>>(in, n, block_sums, out);
}

} // namespace ssv
