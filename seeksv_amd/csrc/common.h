// common.h - shared device helpers for the gfx950 kernels (64-wide wavefronts throughout).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ssv {

// sam/bam.h:97-124 flag bits, :133-155 CIGAR ops
enum : int { F_PAIRED = 1, F_PROPER = 2, F_UNMAP = 4, F_MUNMAP = 8, F_REV = 16, F_MREV = 32, F_SECONDARY = 256, F_QCFAIL = 512, F_DUP = 1024 };
enum : int { C_M = 0, C_I = 1, C_D = 2, C_N = 3, C_S = 4, C_H = 5, C_P = 6, C_EQ = 7, C_X = 8 };

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// 16-byte streaming load that does not displace the L2 working set (the record arrays are read exactly once)
__device__ __forceinline__ uint4 stream_load_u4(const void *p)
{
	u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
	return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int4 stream_load_i4(const void *p)
{
	i32x4 v = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(p));
	return make_int4(v.x, v.y, v.z, v.w);
}

// A pointer that was an integer (an event's `src`, a batch address plus an offset) is a FLAT pointer to the compiler: its loads are
// flat_load instructions, which count on BOTH memory counters - every wait for an LDS result (lgkmcnt) then also waits for the loads from
// memory still in flight, and a software pipeline that keeps loads in flight across LDS work does not overlap anything.  These say
// "global memory" explicitly: global_load, vmcnt only.
// A workgroup barrier for data exchanged through LDS only.  __syncthreads() is also a fence for global memory: hipcc puts `s_waitcnt vmcnt(0)` in front of every
// s_barrier, which ends every global load the wavefront has in flight - a streaming kernel's prefetched tiles (one barrier per tile: k_getsv_scan, k_clip_scan_ends).
// The instruction itself needs no such wait; what the other wavefronts must see is the LDS write: lgkmcnt(0).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// A value every lane reads from the same address in memory that no kernel writes while this one runs, through the scalar cache (s_load, lgkmcnt): a plain
// `p[i]` of a pointer the compiler cannot prove read-only is a VECTOR load even at a uniform address - and waiting for it (vmcnt counts in order) waits for every
// vector load issued before it, e.g. a streaming kernel's prefetched next tiles.
template <typename T> __device__ __forceinline__ T scalar_load(const T *p) { return *(const __attribute__((address_space(4))) T *)(uintptr_t)p; }
template <typename T> using gptr = const T __attribute__((address_space(1))) *;
template <typename T> __device__ __forceinline__ gptr<T> as_global(const T *p) { return (gptr<T>)p; }
template <typename T> __device__ __forceinline__ gptr<T> global_at(uint64_t addr) { return (gptr<T>)(uintptr_t)addr; }

constexpr int WAVE = 64;
constexpr int BLOCK = 256;            // 4 waves per workgroup
constexpr int WAVES_PER_BLOCK = BLOCK / WAVE;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// inclusive prefix sum across the 64 lanes of a wave
template <typename T>
__device__ __forceinline__ T wave_inclusive_sum(T v)
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		T o = __shfl_up(v, d, 64);
		if (lane_id() >= d) v += o;
	}
	return v;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
	return v;
}

template <typename T>
__device__ __forceinline__ T wave_max(T v)
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) { T o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
	return v;
}

// Exclusive prefix sum of one value per thread over a 256-thread block; *total = block sum.
// lds must hold WAVES_PER_BLOCK + 1 elements.  Contains two barriers.
template <typename T>
__device__ __forceinline__ T block_exclusive_sum(T v, T *lds, T *total)
{
	T inc = wave_inclusive_sum(v);
	if (lane_id() == 63) lds[wave_id()] = inc;
	__syncthreads();
	T base = 0, tot = 0;
#pragma unroll
	for (int w = 0; w < WAVES_PER_BLOCK; ++w) {
		T x = lds[w];
		if (w < wave_id()) base += x;
		tot += x;
	}
	__syncthreads();
	*total = tot;
	return base + inc - v;
}

} // namespace ssv
