// radix_sort.h - stable LSD radix sort of (u64 key, u32 value) pairs, 8 bits per pass.
// Per pass: LDS histogram per tile -> device-wide scan of the digit-major histogram -> stable scatter
// (per-wave match masks from ballots give each key its rank among equal digits; waves and rounds are
// ordered through LDS counters).  Bins the clip events by (contig, side, position) keeping BAM order
// inside a bin, which the greedy clustering depends on (clip_reads.cpp:260-283).
#pragma once

#include "common.h"
#include "scan.h"

namespace ssv {

constexpr int RS_ROUNDS = 8;
constexpr int RS_TILE = BLOCK * RS_ROUNDS; // 2048 keys per workgroup

__global__ __launch_bounds__(BLOCK) void k_rs_hist(const uint64_t *__restrict__ keys, int64_t n, int shift, int64_t ntiles, uint32_t *__restrict__ ghist)
{
	__shared__ uint32_t h[256];
	h[threadIdx.x] = 0;
	__syncthreads();
	int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
	for (int r = 0; r < RS_ROUNDS; ++r) {
		int64_t i = base + (int64_t)r * BLOCK + threadIdx.x;
		if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255], 1u);
	}
	__syncthreads();
	ghist[(int64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x]; // digit-major
}

__global__ __launch_bounds__(BLOCK) void k_rs_scatter(const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, int64_t n, int shift,
                                                      int64_t ntiles, const uint32_t *__restrict__ goff, uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out)
{
	__shared__ uint32_t running[256];                 // next free output slot per digit for this tile
	__shared__ uint32_t wcnt[WAVES_PER_BLOCK][256];   // (round tag << 8 | count) per wave and digit
	running[threadIdx.x] = goff[(int64_t)threadIdx.x * ntiles + blockIdx.x];
#pragma unroll
	for (int w = 0; w < WAVES_PER_BLOCK; ++w) wcnt[w][threadIdx.x] = 0xffffffffu;
	__syncthreads();
	const int64_t base = (int64_t)blockIdx.x * RS_TILE;
	const int w = wave_id();
	for (int r = 0; r < RS_ROUNDS; ++r) {
		int64_t i = base + (int64_t)r * BLOCK + threadIdx.x;
		bool valid = i < n;
		uint64_t key = valid ? keys_in[i] : 0;
		uint32_t val = valid ? vals_in[i] : 0;
		uint32_t d = (uint32_t)(key >> shift) & 255u;
		// lanes of this wave holding the same digit
		uint64_t m = __ballot(valid);
#pragma unroll
		for (int b = 0; b < 8; ++b) {
			uint64_t bal = __ballot((d >> b) & 1u);
			m &= ((d >> b) & 1u) ? bal : ~bal;
		}
		uint32_t rank = (uint32_t)__popcll(m & lanemask_lt());
		uint32_t cnt = (uint32_t)__popcll(m);
		if (valid && rank == 0) wcnt[w][d] = ((uint32_t)r << 8) | cnt;
		__syncthreads();
		uint32_t dst = 0;
		if (valid) {
			dst = running[d] + rank;
			for (int ww = 0; ww < w; ++ww) {
				uint32_t x = wcnt[ww][d];
				if ((x >> 8) == (uint32_t)r) dst += x & 255u;
			}
		}
		__syncthreads();
		if (valid && rank == 0) atomicAdd(&running[d], cnt);
		if (valid) { keys_out[dst] = key; vals_out[dst] = val; }
		// the barrier at the top of the next round orders the running[] updates before their next use
		__syncthreads();
	}
}

// iota for the value array
__global__ void k_iota(uint32_t *v, int64_t n)
{
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) v[i] = (uint32_t)i;
}

// Sorts by the low key_bits bits.  keys/vals: two buffers each (ping-pong); returns the index (0/1) holding the result.
// ghist: scratch of 256 * ntiles u32; scan_scratch: scan_scratch_elems(256 * ntiles) u32.
static inline int64_t rs_tiles(int64_t n) { return (n + RS_TILE - 1) / RS_TILE; }

static inline int radix_sort_pairs(hipStream_t st, uint64_t *keys[2], uint32_t *vals[2], int64_t n, int key_bits,
                                   uint32_t *ghist, uint32_t *scan_scratch)
{
	if (n == 0) return 0;
	int64_t nt = rs_tiles(n);
	int cur = 0;
	for (int shift = 0; shift < key_bits; shift += 8) {
		k_rs_hist<<<(unsigned)nt, BLOCK, 0, st>>>(keys[cur], n, shift, nt, ghist);
		exclusive_scan<uint32_t, uint32_t>(st, ghist, ghist, 256 * nt, 0u, scan_scratch, nullptr);
		k_rs_scatter<<<(unsigned)nt, BLOCK, 0, st>>>(keys[cur], vals[cur], n, shift, nt, ghist, keys[cur ^ 1], vals[cur ^ 1]);
		cur ^= 1;
	}
	return cur;
}

} // namespace ssv
