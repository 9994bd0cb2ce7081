// resolve_wave.h - pass 2 of the device inflate by ONE WAVEFRONT per BGZF block: the holes that pass 1 left for the matches are filled from the block's own
// earlier bytes: k_bgzf_resolve_wave (bamdec_kernels.h) walks a block's whole token stream in global memory, k_bgzf_resolve_win inside a window of the block in LDS.
#pragma once

#include "common.h"
#include "inflate_core.h"

namespace ssv {

constexpr uint32_t TOKEN_NONE = 0xff800000u; // an escape that skips nothing
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

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


// ---- pass 2 with the block's recent bytes in LDS (round 4, late): k_bgzf_resolve_win ----------------------------------------------------
//
// What wave_resolve_tokens costs is its CU's vector memory path (profiles/r04_vmem_issue.txt, r04_inflate_pmc.txt): every match is ~2 scattered loads and
// 2 scattered stores, each a lane's own line - 2.3-3.8 clocks of the CU apiece when the L2 has the line, 7.7 when not - spread over ~40 instructions a round
// with a few lanes active in each (10-20 clocks apiece): ~20 clocks of a CU per match, 13.6 ms per 5.3 GB, whatever the occupancy.  Here a wavefront keeps a
// WINDOW of its block in LDS - RW_WIN bytes ending with the round at hand: the round's bytes (pass 1's literals, the holes open) come in with coalesced 16-byte
// loads, the matches are copied INSIDE the window with unaligned LDS accesses (gfx950 takes them at any address) when their source is in it - the usual case:
// a BAM record repeats fields of the records just before it -, from global memory into the window when it lies further back, and the round's bytes leave with
// coalesced 16-byte stores.  No scattered store is left and most scattered loads are gone.
//   * coordinates: A(p) = p + (address of the block & 15), so that A = 0 mod 16 is a 16-byte boundary of memory; the window holds A in [wbase, wend), whole chunks;
//   * a round takes the longest prefix of its 64 tokens whose bytes fit RW_ROUND (the rest waits for the next round), and the literals in front of a round's
//     first match are stepped over (pass 1 wrote them): a round's region is at most RW_ROUND + 32 bytes whatever the data;
//   * the first and the last chunk of a block share their 16 bytes with the neighbouring blocks - other wavefronts' - and are stored byte by byte.
#ifndef RW_WIN_BYTES
#define RW_WIN_BYTES 4096 // (8192: half the wavefronts per CU - pass 2 12.8 instead of 8.6 ms per 5.3 GB)
#endif
#ifndef RW_KEEP_BYTES
#define RW_KEEP_BYTES (RW_WIN_BYTES / 2 - 512)
#endif
constexpr uint32_t RW_WIN = RW_WIN_BYTES;        // window bytes per wavefront
constexpr uint32_t RW_KEEP = RW_KEEP_BYTES;      // history kept in front of the round when the window moves
constexpr uint32_t RW_ROUND = RW_WIN - RW_KEEP - 64; // bytes of a round
constexpr uint32_t RW_SKIP = 256;                // literals in front of a round's first match beyond this are stepped over
#ifndef RW_SMALL_BYTES
#define RW_SMALL_BYTES 79
#endif
constexpr uint32_t RW_SMALL = RW_SMALL_BYTES;    // a match of up to this many bytes is copied by its own lane (four whole 16-byte pieces and one that ends with the match)
static_assert(RW_KEEP + RW_ROUND + 64 <= RW_WIN && RW_ROUND >= 1024 && RW_KEEP % 16 == 0, "a round and its history fit the window");
static_assert(RW_SMALL >= 3 && RW_SMALL <= 79, "the own-lane copy is four 16-byte pieces and one that ends with the match (<= 79 bytes); a deflate match is 3 bytes or more");

__device__ __forceinline__ uint32_t lds_off(const void *p) { return (uint32_t)(uintptr_t)p; } // (the low half of a shared pointer is the LDS address)
// unaligned LDS accesses (the waits are the caller's: lds_wait makes the registers depend on it)
__device__ __forceinline__ u32x4 lds_r128(uint32_t a) { u32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory"); return v; }
__device__ __forceinline__ uint32_t lds_r32(uint32_t a) { uint32_t v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a) : "memory"); return v; }
__device__ __forceinline__ uint32_t lds_r8(uint32_t a) { uint32_t v; asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"(a) : "memory"); return v; }
__device__ __forceinline__ void lds_wait(u32x4 &a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a) : : "memory"); }
__device__ __forceinline__ void lds_wait(u32x4 &a, u32x4 &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) : : "memory"); }
__device__ __forceinline__ void lds_wait(uint32_t &a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a) : : "memory"); }
__device__ __forceinline__ void lds_w128(uint32_t a, u32x4 v) { asm volatile("ds_write_b128 %0, %1" : : "v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_w64(uint32_t a, uint32_t lo, uint32_t hi) { u32x2 v = {lo, hi}; asm volatile("ds_write_b64 %0, %1" : : "v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_w32(uint32_t a, uint32_t v) { asm volatile("ds_write_b32 %0, %1" : : "v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_w16(uint32_t a, uint32_t v) { asm volatile("ds_write_b16 %0, %1" : : "v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_w8(uint32_t a, uint32_t v) { asm volatile("ds_write_b8 %0, %1" : : "v"(a), "v"(v) : "memory"); }

// the first L (0: none; 3..15) bytes of h to LDS address a, exactly
__device__ __forceinline__ void lds_put_small(uint32_t a, uint32_t L, u32x4 h)
{
	if (L >= 8u) { // bytes [L - 8, L) of h: a window of its dwords
		const uint32_t k = L - 8u, sb = k & 3u;
		const uint32_t x0 = __builtin_amdgcn_alignbyte(h.y, h.x, sb), x1 = __builtin_amdgcn_alignbyte(h.z, h.y, sb), x2 = __builtin_amdgcn_alignbyte(h.w, h.z, sb);
		lds_w64(a, h.x, h.y);
		lds_w64(a + k, k >= 4u ? x1 : x0, k >= 4u ? x2 : x1);
	} else if (L >= 4u) { lds_w32(a, h.x); lds_w32(a + L - 4u, __builtin_amdgcn_alignbyte(h.y, h.x, L - 4u)); }
	else if (L == 3u) { lds_w16(a, h.x); lds_w8(a + 2u, h.x >> 16); }
}

// the tokens tk[0, n) of one block (ulen bytes at o), all 64 lanes of the wavefront together; win = the wavefront's RW_WIN + 64 bytes of LDS
// dbg (SSV_INFLATE_PHASES=1): [0] rounds, [1] phases, [2] all-lanes matches, [3] tokens, [4] window moves, [5..9] cycles: window, dependencies, own-lane copies, all-lanes copies, store
template <bool DBG>
__device__ __forceinline__ void wave_resolve_tokens_win(uint8_t *o, uint32_t ulen, const uint32_t *tk, uint32_t n, uint8_t *win, int lane, unsigned long long *dbg)
{
	unsigned long long d_cnt[5] = {0, 0, 0, 0, 0}, d_cyc[5] = {0, 0, 0, 0, 0}, tc = 0;
	auto lap = [&](int i) { if (DBG) { const unsigned long long now = __builtin_readcyclecounter(); d_cyc[i] += now - tc; tc = now; } };
	if (DBG) tc = __builtin_readcyclecounter();
	const uint32_t W0 = lds_off(win);
	const uint32_t al = (uint32_t)(reinterpret_cast<uintptr_t>(o) & 15u); // A(p) = p + al
	uint8_t *const oa = o - al;                                           // oa + A = the byte at aligned coordinate A
	const uint32_t aend = ulen + al;                                      // A of the block's end
	uint32_t wbase = 0, wend = 0; // the window holds A in [wbase, wend)
	uint32_t pos = 0, dirty = 0xffffffffu;
	uint32_t t0 = 0;
	uint32_t next = (uint32_t)lane < n ? tk[lane] : TOKEN_NONE;
	while (t0 < n) {
		if (DBG) { lap(4); ++d_cnt[0]; }
		const uint32_t w = next;
		const bool esc = (w >> 23) == 511u;
		uint32_t lit = esc ? (w & 0x7fffffu) : (w >> 23);
		const uint32_t len = esc ? 0u : (w & 255u) + 3u, dist = ((w >> 8) & 0x7fffu) + 1u;
		// the literals in front of the round's first match: pass 1 wrote them, nothing to do but to step over them
		const uint32_t lit0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lit);
		if (lit0 > RW_SKIP) { pos += lit0; if (lane == 0) lit = 0; }
		const uint32_t inc = wave_inclusive_sum(lit + len); // where this lane's token ends, from the round's first byte
		const int k = (int)__popcll(__ballot(inc <= RW_ROUND)); // the tokens of this round (>= 1: the first one is a match of <= 258 bytes behind <= RW_SKIP literals)
		const uint32_t total = (uint32_t)__shfl((int)inc, k - 1, WAVE);
		t0 += (uint32_t)k;
		if (DBG) d_cnt[3] += (unsigned long long)k;
		next = t0 + (uint32_t)lane < n ? tk[t0 + lane] : TOKEN_NONE; // the next round's tokens travel with this round's loads
		const bool mine = lane < k;
		const uint32_t hend = pos + inc, dst = hend - len, src = dst - dist;
		const uint32_t need = len < dist ? len : dist; // the source's bytes: [src, src + need)
		bool done = esc || !mine;
		// ---- the window: room for [A(pos) & ~15, need_end), the round's chunks loaded ----
		const uint32_t cs = (pos + al) & ~15u, need_end = (pos + total + al + 15u) & ~15u;
		if (cs >= wend) { wbase = cs; wend = cs; } // (the first round, or a long step over literals: nothing of the old window is of use)
		else if (need_end - wbase > RW_WIN) { // move the window: keep RW_KEEP bytes in front of the round (cs - wbase > RW_KEEP here, or the round would have fit)
			const uint32_t nb = cs - RW_KEEP, cnt = wend - nb, d = nb - wbase;
			for (uint32_t c0 = 0; c0 < cnt; c0 += 16u * WAVE) { // downwards, a KB at a time: every turn reads all it moves before it writes
				const uint32_t c = c0 + 16u * (uint32_t)lane;
				u32x4 v = {0, 0, 0, 0};
				if (c < cnt) v = lds_r128(W0 + d + c);
				lds_wait(v);
				if (c < cnt) lds_w128(W0 + c, v);
			}
			wbase = nb;
			if (DBG) ++d_cnt[4];
		}
		// (requesting the next round's chunks a round ahead was measured: no gain - the pass is bound by its instructions, not by this trip)
		for (uint32_t a = wend + 16u * (uint32_t)lane; a < need_end; a += 16u * WAVE) lds_w128(W0 + a - wbase, ld128(oa + a)); // (whole chunks: up to 15 bytes of the neighbours at the block's ends)
		wend = need_end;
		lap(0);
		// ---- the earlier lanes whose holes the source touches: [lo, hi] (empty: lo > hi) ----
		// (FLATTENING the chains - a source that lies inside ONE earlier hole of the round moves to that match's source - was built and measured: 4.7 -> 4.0 phases a round
		// and 3.6 x the cycles of this search: 10.8 instead of 8.8 ms.  Sources mostly span literals and several holes.)
		uint64_t deps = 0;
		if (__any(!done && src + need > pos)) {
			int c_end = 0, c_start = 0; // lanes whose hole ends at or before src / starts before src + need (both sequences ascend with the lane)
#pragma unroll
			for (int step = WAVE / 2; step >= 1; step >>= 1) {
				const uint32_t e = (uint32_t)__shfl((int)hend, c_end + step - 1, WAVE), st = (uint32_t)__shfl((int)dst, c_start + step - 1, WAVE);
				if (e <= src) c_end += step;
				if (st < src + need) c_start += step;
			}
			const int lo = c_end, hi = (c_start < lane ? c_start : lane) - 1;
			if (!done && lo <= hi) deps = (hi >= 63 ? ~0ull : (1ull << (hi + 1)) - 1ull) & ~((1ull << lo) - 1ull);
		}
		// where the source lies: inside the window (LDS), in front of it (memory), or across its start (byte by byte)
		const uint32_t wb = wbase > al ? wbase - al : 0u; // block position of the window's first byte
		const bool s_win = src >= wb, s_mem = src + need <= wb;
		const bool small = !done && dist >= len && len <= RW_SMALL && (s_win || s_mem);
		const uint32_t sa = W0 + src + al - wbase, da = W0 + dst + al - wbase; // LDS addresses of source (when in the window) and hole
		lap(1);
		// ---- phases ----
		for (;;) {
			const uint64_t open = __ballot(!done);
			if (!open) break;
			if (DBG) ++d_cnt[1];
			const bool ready = !done && (deps & open) == 0ull;
			if (__any(ready && !s_win && src + need > dirty)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dirty = 0xffffffffu; }
			{ // every lane its own match of up to RW_SMALL bytes: 16-byte pieces at 0, 16, 32, 48 as far as they fit, and one that ends with the match (what lies behind
			  // a source shorter than 16 bytes is read and not used).  Overlapping reads of a sorted BAM repeat 40-75 bytes of the read before: with 32 bytes as the limit
			  // a third of a round's matches would go one after the other through the all-lanes path below.
				const bool go = ready && small;
				const uint32_t L = go ? len : 0u;
				const bool tail = L > 16u && (L & 15u);
				{ // piece 0 and the one at the end
					u32x4 p0 = {0, 0, 0, 0}, t = p0;
					if (go && s_win) { p0 = lds_r128(sa); if (tail) t = lds_r128(sa + L - 16u); }
					if (go && !s_win) { p0 = ld128(o + src); if (tail) t = ld128(o + src + L - 16u); }
					lds_wait(p0, t);
					if (L >= 16u) lds_w128(da, p0); else lds_put_small(da, L, p0);
					if (tail) lds_w128(da + L - 16u, t);
				}
#pragma unroll
				for (uint32_t q = 1; q < 4u; ++q) // pieces 1..3, each in a turn of its own where a lane has one (few registers; no other lane's source lies in a hole written above: that lane would not be ready)
					if (__any(L >= 16u * q + 16u)) {
						u32x4 pq = {0, 0, 0, 0};
						if (L >= 16u * q + 16u) { if (s_win) pq = lds_r128(sa + 16u * q); else pq = ld128(o + src + 16u * q); }
						lds_wait(pq);
						if (L >= 16u * q + 16u) lds_w128(da + 16u * q, pq);
					}
			}
			lap(2);
			for (uint64_t m = __ballot(ready && !small); m; m &= m - 1ull) { // long matches, repeating patterns, sources across the window's start: all lanes on each
				const int q = __ffsll((long long)m) - 1;
				if (DBG) ++d_cnt[2];
				const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)dst, q), l0 = (uint32_t)__builtin_amdgcn_readlane((int)len, q), di = (uint32_t)__builtin_amdgcn_readlane((int)dist, q);
				const uint32_t s0 = d0 - di;
				if (di >= l0 && s0 >= wb) { // source and hole apart, both in the window: a dword a lane (one that would reach past the end is moved back to end with the match)
					const uint32_t last = l0 - 4u;
					for (uint32_t i4 = 4u * (uint32_t)lane; i4 < l0; i4 += 4u * WAVE) {
						const uint32_t qo = i4 < last ? i4 : last;
						uint32_t v = lds_r32(W0 + s0 + qo + al - wbase);
						lds_wait(v);
						lds_w32(W0 + d0 + qo + al - wbase, v);
					}
					continue;
				}
				const uint32_t rcp = 0xffffffffu / di + 1u; // (i < 2^9, 2 <= di < 2^15: i / di = the high half of i * rcp exactly; di = 1 does not fit and is taken by hand)
				for (uint32_t i = (uint32_t)lane; i < l0; i += WAVE) {
					const uint32_t p = s0 + (di >= l0 ? i : di == 1u ? 0u : i - __umulhi(i, rcp) * di);
					uint32_t b;
					if (p >= wb) { b = lds_r8(W0 + p + al - wbase); lds_wait(b); }
					else b = o[p];
					lds_w8(W0 + d0 + i + al - wbase, b);
				}
			}
			lap(3);
			done = done || ready;
		}
		// ---- the round's chunks leave (the first one again: it was the last one of the round before) ----
		for (uint32_t a = cs + 16u * (uint32_t)lane; a < need_end; a += 16u * WAVE) {
			u32x4 v = lds_r128(W0 + a - wbase);
			lds_wait(v);
			if (a >= al && a + 16u <= aend) st128u(oa + a, v);
			else { // a chunk shared with a neighbouring block: this block's bytes only
				const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
				for (uint32_t i = 0; i < 16u; ++i) if (a + i >= al && a + i < aend) oa[a + i] = (uint8_t)(x[i >> 2] >> (8u * (i & 3u)));
			}
		}
		if (dirty > (cs > al ? cs - al : 0u)) dirty = cs > al ? cs - al : 0u;
		pos += total;
	}
	if (DBG && dbg && lane == 0) { lap(4); for (int i = 0; i < 5; ++i) { atomicAdd(dbg + i, d_cnt[i]); atomicAdd(dbg + 5 + i, d_cyc[i]); } }
}

} // namespace ssv
