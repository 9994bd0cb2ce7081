// resolve_wave.h - pass 2 of the device inflate by ONE WAVEFRONT per BGZF block: the holes that pass 1 left for the matches are filled from the block's own
// earlier bytes.  Used two ways: k_bgzf_resolve_wave (bamdec_kernels.h) walks a block's whole token stream; the fused form of pass 1 (inflate_wave.h,
// SSV_INFLATE_FUSED) calls wave_resolve_tokens for the tokens of every window right after it wrote them.
#pragma once

#include "common.h"
#include "inflate_core.h"

namespace ssv {

constexpr uint32_t TOKEN_NONE = 0xff800000u; // an escape that skips nothing

// Every lane its own short match (L = 0: none; else 3..32 bytes, source and destination disjoint): a match of 5..32 bytes goes as a head and a tail of
// 4 / 8 / 16 bytes that overlap (same bytes where they do) - two loads and two stores whatever the length, all loads first, one turn; three bytes: one dword
// read (its fourth byte is the hole's first), a short and a byte written.
// (Measured before this form, dword by dword with four dwords a turn: the same time on real reads, 5 % more on the low-entropy sample.  Also tried there:
// all eight dwords' loads before the first store - 15.7 -> 23.7 ms.)
__device__ __forceinline__ void st16u(uint8_t *p, uint32_t v) { asm volatile("global_store_short %0, %1, off" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st64u(uint8_t *p, uint64_t v) { asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st128u(uint8_t *p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ uint64_t ld64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
__device__ __forceinline__ u32x4 ld128(const uint8_t *p) { u32x4 v; memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void copy_own_wide(uint8_t *o, uint32_t src, uint32_t dst, uint32_t L)
{
	const bool c3 = L == 3u, c4 = L >= 4u && L <= 8u, c8 = L >= 9u && L <= 16u, c16 = L >= 17u;
	uint32_t a0 = 0, a1 = 0;
	uint64_t b0 = 0, b1 = 0;
	u32x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
	if (c3 || c4) a0 = ld32(o + src); // (three bytes: the dword's fourth byte is the hole's first)
	if (c4 && L > 4u) a1 = ld32(o + src + L - 4u);
	if (c8) { b0 = ld64(o + src); b1 = ld64(o + src + L - 8u); }
	if (c16) { d0 = ld128(o + src); d1 = ld128(o + src + L - 16u); }
	if (c3) { st16u(o + dst, a0); o[dst + 2u] = (uint8_t)(a0 >> 16); }
	if (c4) { st32u(o + dst, a0); if (L > 4u) st32u(o + dst + L - 4u, a1); }
	if (c8) { st64u(o + dst, b0); st64u(o + dst + L - 8u, b1); }
	if (c16) { st128u(o + dst, d0); st128u(o + dst + L - 16u, d1); }
}

// ---- pass 2 with ONE WAVEFRONT per block (round 4) ------------------------------------------------------------------------------------
//
// k_bgzf_resolve above gives a block 16 lanes, so a wavefront works on four blocks and 30-60 K blocks must be in flight to fill the chip: 2-4 GB
// of output windows, every match source a line nobody has in cache (PMC, profiles/r04_inflate_pmc.txt: 0.42 G L2 misses per 5.3 GB of output
// = 28 G lines a second, the rate the fabric serves lines touched at random - the pass is bound by that count, not by bytes or instructions).
// With 64 lanes on ONE block the same number of wavefronts keeps a quarter of the blocks in flight, and a round takes 64 tokens:
//   * places by a wavefront prefix sum;
//   * whose holes does my source touch?  The round's holes lie in increasing order, so the earlier tokens whose holes overlap [src, src + need)
//     are a RANGE of lanes [lo, hi]: two six-step binary searches over the lanes' hole ends / starts (ds_bpermute), once per round - and only
//     when some source reaches into the round at all;
//   * phases: a match is ready when no lane of its range is still open - one AND with the ballot of the open lanes; every phase copies all ready
//     short matches at once (each lane its own: head + tail, two loads then two stores) and the ready long ones / repeating patterns one after the
//     other with all 64 lanes (256 bytes a trip).  The first open lane is always ready, so a round ends after at most 64 phases (typically 2-4).
__device__ __forceinline__ void resolve_one_wave(uint8_t *o, uint32_t pos, uint32_t len, uint32_t dist, int lane)
{
	const uint32_t src = pos - dist;
	if (dist >= len) {
		if (len >= 4u) { // dwords; one that would reach past the end is moved back to end with the match (same bytes)
			const uint32_t last = len - 4u, off0 = 4u * (uint32_t)lane, off1 = off0 + 256u;
			const uint32_t q0 = off0 < last ? off0 : last, q1 = off1 < last ? off1 : last;
			uint32_t r0 = 0, r1 = 0;
			if (off0 < len) r0 = ld32(o + src + q0);
			if (off1 < len) r1 = ld32(o + src + q1);
			if (off0 < len) st32u(o + pos + q0, r0);
			if (off1 < len) st32u(o + pos + q1, r1);
		} else if ((uint32_t)lane < len) o[pos + (uint32_t)lane] = o[src + (uint32_t)lane];
	} else {
		for (uint32_t i = (uint32_t)lane; i < len; i += WAVE) o[pos + i] = o[src + i % dist];
	}
}

// where a block's resolution stands between calls
struct WaveResolveState {
	uint32_t pos = 0;              // output position behind the last token that was worked off
	uint32_t dirty = 0xffffffffu;  // positions from here on may hold stores of this pass that have not been waited for
};

// the tokens tk[0, n) of one block, all 64 lanes of the wavefront together; o = the block's output
__device__ __forceinline__ void wave_resolve_tokens(uint8_t *o, const uint32_t *tk, uint32_t n, WaveResolveState &R, int lane)
{
	uint32_t pos = R.pos, dirty = R.dirty;
	uint32_t next = (uint32_t)lane < n ? tk[lane] : TOKEN_NONE;
	for (uint32_t t0 = 0; t0 < n; t0 += WAVE) {
		const uint32_t w = next;
		next = t0 + WAVE + (uint32_t)lane < n ? tk[t0 + WAVE + lane] : TOKEN_NONE; // the next round's tokens travel with this round's loads
		const bool esc = (w >> 23) == 511u;
		const uint32_t lit = esc ? (w & 0x7fffffu) : (w >> 23);
		const uint32_t len = esc ? 0u : (w & 255u) + 3u, dist = ((w >> 8) & 0x7fffu) + 1u;
		const uint32_t inc = wave_inclusive_sum(lit + len); // where this lane's token ends, from the round's first byte
		const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
		const uint32_t hend = pos + inc, dst = hend - len, src = dst - dist;
		const uint32_t need = len < dist ? len : dist; // the source's bytes: [src, src + need)
		const bool small = !esc && dist >= len && len <= 32u;
		bool done = esc;
		// ---- the earlier lanes whose holes the source touches: [lo, hi] (empty: lo > hi) ----
		uint64_t deps = 0;
		if (__any(!done && src + need > pos)) {
			int c_end = 0, c_start = 0; // lanes whose hole ends at or before src / starts before src + need (both sequences ascend with the lane)
#pragma unroll
			for (int step = WAVE / 2; step >= 1; step >>= 1) {
				const uint32_t e = (uint32_t)__shfl((int)hend, c_end + step - 1, WAVE), st = (uint32_t)__shfl((int)dst, c_start + step - 1, WAVE);
				if (e <= src) c_end += step;
				if (st < src + need) c_start += step;
			}
			const int lo = c_end, hi = (c_start < lane ? c_start : lane) - 1; // (lane 63 is never somebody's earlier lane: the searches stop at 63 elements)
			if (!done && lo <= hi) deps = (hi >= 63 ? ~0ull : (1ull << (hi + 1)) - 1ull) & ~((1ull << lo) - 1ull);
		}
		// ---- phases ----
		for (;;) {
			const uint64_t open = __ballot(!done);
			if (!open) break;
			const bool ready = !done && (deps & open) == 0ull;
			if (__any(ready && src + need > dirty)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dirty = 0xffffffffu; }
			copy_own_wide(o, src, dst, ready && small ? len : 0u);
			for (uint64_t m = __ballot(ready && !small); m; m &= m - 1ull) {
				const int k = __ffsll((long long)m) - 1;
				resolve_one_wave(o, (uint32_t)__builtin_amdgcn_readlane((int)dst, k), (uint32_t)__builtin_amdgcn_readlane((int)len, k), (uint32_t)__builtin_amdgcn_readlane((int)dist, k), lane);
			}
			if (dirty > pos) dirty = pos;
			done = done || ready;
		}
		pos += total;
	}
	R.pos = pos; R.dirty = dirty;
}

} // namespace ssv
