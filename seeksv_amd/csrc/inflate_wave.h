// inflate_wave.h - pass 1 of the device inflate (DEFLATE -> literals in place + match tokens, bamdec_kernels.h) with ONE WAVEFRONT PER BGZF BLOCK.
//
// k_bgzf_tokens gives every lane its own block: 64 unrelated Huffman tables and input windows per wavefront (31 KB of LDS: five wavefronts per CU), every
// memory instruction 64 unrelated addresses, every step the union of 64 lanes' paths.  Here the 64 lanes share one block, one set of tables (a 10-bit and an
// 8-bit look-up table in LDS, canonical walk for longer codes) and one 3 KB window of the input (7 KB of LDS in all: 22 wavefronts per CU), and decode it
// SPECULATIVELY in parallel:
//   * the window is cut into 64 segments of 384 bits; lane 0 starts at the true position, lane i at the first bit of segment i (a guess);
//   * every lane decodes symbols until it has left its segment and remembers where it ended (Huffman streams re-synchronise: a decoder that starts at a wrong
//     bit falls into step with the true symbol boundaries after a few symbols, so most of these ends are right);
//   * then lane i takes lane i-1's end as its start and decodes again if that differs from the start it used - repeated until no start changes.  Lane 0 is
//     right by construction, so at the fixed point every lane starts exactly where the one before it ended: the chain IS the sequential decode, whatever
//     the guesses were (no probabilistic step anywhere; the worst case is 64 rounds, the usual one three);
//   * the lanes' output sizes and token counts go through a prefix sum, and one more decode writes literals and tokens to their final places.
// The token stream is bit-identical to TokenOut's (inflate_core.h): pass 2 (k_bgzf_resolve) is unchanged.
// Code lengths of a dynamic block are read by all lanes in step (a serial chain by nature), the tables are built cooperatively (ballot ranks).
// Selected with SSV_TOKENS=wave; DESIGN.md section 9 has the measurements.
#pragma once

#include "common.h"
#include "inflate_core.h"

namespace ssv {

constexpr int WV_SEG_BITS = 384;                        // input bits per lane and window
// The first round's guesses start WV_TAIL_BITS before their segment's end.  A whole segment (the default) costs a full decode; 128-192 bits take a third to
// a half of that, but with so short a run-in a third of the lanes end wrong and the later rounds grow by what the first one saved (measured: 14.6 -> 14.7 ms).
#ifndef WV_TAIL_BITS
#define WV_TAIL_BITS WV_SEG_BITS
#endif
constexpr int WV_SEG_DW = WV_SEG_BITS / 32;
constexpr int WV_COLS = WAVE + 1;                       // the 64 lanes' segments side by side + one column for the last lane's overshoot (a symbol is at most 48 bits)
constexpr int WV_WIN_DW = WV_COLS * WV_SEG_DW;
constexpr int WV_LT = 10, WV_DT = 8, WV_CT = 7;         // look-up bits: literal/length, distance, code-length code

struct WaveLds {
	uint32_t win[WV_WIN_DW];
	// look-up entries (wv_entry): the code's length in bits 0-3 and what the symbol MEANS, worked out once when the table is built instead of by every lane at
	// every step; 0: no code of <= T bits starts like this (a longer code - rare symbols: a short walk finds it - or none).
	uint16_t lit[1 << WV_LT];
	uint16_t dst[1 << WV_DT];
	uint16_t clt[1 << WV_CT];
	uint16_t perm0[288], perm1[32], perm2[32];          // symbols by (length, symbol): the slow path of codes longer than the look-up
	__device__ __forceinline__ uint16_t *perm(int set) { return set == 0 ? perm0 : set == 1 ? perm1 : perm2; }
	__device__ __forceinline__ const uint16_t *perm(int set) const { return set == 0 ? perm0 : set == 1 ? perm1 : perm2; }
	uint16_t cnt[3][16], first[3][16], off[3][16];
	uint8_t lens[320];
	uint8_t cll[32];
};

// Where dword j of the window lies in LDS: dword k of lane i's segment at k * 65 + i (65 columns: the 65th is the room behind the last segment).  Lane i reads
// dwords 12 i + k; laid out as they come that is a stride of 12 dwords between neighbouring lanes - a bank conflict on every read of the input.  A lane's reader
// (WvBits) walks its column with an address and a row counter: no division per refill.
__device__ __forceinline__ uint32_t wv_at(uint32_t j) { return (j % WV_SEG_DW) * WV_COLS + j / WV_SEG_DW; }

// What a symbol of code `set` means, as a 16-bit table entry with the code's length l in bits 0-3:
//   literal / length code: bit 4 = 0: a literal, the byte in bits 8-15.  bit 4 = 1: bits 5-7 = the length's extra bits (0..5; 6: the end-of-block code, 7: a
//                          symbol that does not exist), bits 8-15 = the length's base - 3
//   distance code:         bits 4-7 = the distance's extra bits (0..13; 15: a symbol that does not exist), bits 8-12 the symbol (its base is two shifts away)
//   code-length code:      the symbol in bits 4-8
__device__ __forceinline__ uint32_t wv_entry(int set, uint32_t s, uint32_t l)
{
	if (set == 0) {
		if (s < 256u) return l | (s << 8);
		if (s == 256u) return l | 16u | (6u << 5);
		if (s > 285u) return l | 16u | (7u << 5);
		uint32_t xb = 0, base;
		if (s < 265u) base = s - 254u;
		else if (s == 285u) base = 258u;
		else { xb = (s - 261u) >> 2; base = ((4u + ((s - 265u) & 3u)) << xb) + 3u; }
		return l | 16u | (xb << 5) | ((base - 3u) << 8);
	}
	if (set == 1) return l | ((s > 29u ? 15u : s < 4u ? 0u : (s >> 1) - 1u) << 4) | (s << 8);
	return (s << 4) | l;
}

__device__ __forceinline__ uint32_t wv_fetch(const WaveLds &L, uint32_t w0_bits, uint32_t p) // the 32 bits from position p on
{
	const uint32_t rel = p - w0_bits, i = rel >> 5;
	return __builtin_amdgcn_alignbit(L.win[wv_at(i + 1)], L.win[wv_at(i)], rel & 31u);
}

// A lane's bit buffer over the window: 33..64 valid bits after need(); one LDS dword per refill (a symbol step then has ONE dependent trip to LDS, its
// table look-up, instead of three)
struct WvBits {
	uint64_t bb;
	int bc;
	uint32_t a, k, p; // the next window dword's place in LDS and its row (dword of its segment); bit position of bb's bit 0
	__device__ __forceinline__ void adv() { ++k; a += WV_COLS; if (k == (uint32_t)WV_SEG_DW) { k = 0; a -= (uint32_t)(WV_SEG_DW * WV_COLS - 1); } } // down the column, then the top of the next one
	__device__ __forceinline__ void start(const WaveLds &L, uint32_t w0_bits, uint32_t at)
	{
		const uint32_t rel = at - w0_bits, j = rel >> 5, col = j / WV_SEG_DW;
		k = j - col * WV_SEG_DW; a = k * WV_COLS + col; p = at;
		const uint32_t sh = rel & 31u;
		const uint32_t lo = L.win[a];
		adv();
		const uint32_t hi = L.win[a];
		adv();
		bb = (((uint64_t)hi << 32) | (uint64_t)lo) >> sh;
		bc = 64 - (int)sh;
	}
	__device__ __forceinline__ void need(const WaveLds &L) { if (bc <= 32) { bb |= (uint64_t)L.win[a] << bc; bc += 32; adv(); } }
	__device__ __forceinline__ uint32_t peek() const { return (uint32_t)bb; }
	__device__ __forceinline__ void drop(int n) { bb >>= n; bc -= n; p += (uint32_t)n; }
};

// win <- the payload's bytes from the dword at or below bit `cur` on; what lies behind the payload's end reads as zero bits
__device__ __forceinline__ uint32_t wv_load_window(WaveLds &L, const uint8_t *in, uint32_t in_len, uint32_t cur)
{
	const uint32_t w0 = (cur >> 3) & ~3u;
	__syncthreads(); // (everybody is done with the old window)
	for (int i = (int)threadIdx.x; i < WV_WIN_DW; i += WAVE) {
		const uint32_t off = w0 + 4u * (uint32_t)i;
		uint32_t v = 0;
		if (off < in_len) {
			memcpy(&v, in + off, 4); // (up to three bytes past the payload: the chunk buffer goes on behind every payload)
			if (off + 4u > in_len) v &= (1u << (8u * (in_len - off))) - 1u;
		}
		L.win[wv_at((uint32_t)i)] = v;
	}
	__syncthreads();
	return w0 * 8u;
}

// One Huffman code from its lengths lens[0, n): look-up table of T bits (entries of longer codes stay 0) + counts / first codes / offsets / permutation.
// All lanes call it together.  INF_E_OVERSUB for an over-subscribed code, INF_E_CODE for an incomplete one that `rule` does not allow (huff_construct's
// rules, inflate_core.h: both forms of pass 1 refuse the same headers); where it is allowed, its unused bit patterns decode to "invalid".
template <int T>
__device__ __forceinline__ int wv_build(WaveLds &L, int set, const uint8_t *lens, int n, uint16_t *tab, int rule)
{
	const int lane = (int)threadIdx.x;
	for (int i = lane; i < (1 << T) / 2; i += WAVE) reinterpret_cast<uint32_t *>(tab)[i] = 0u;
	uint32_t cnt[16];
#pragma unroll
	for (int l = 0; l < 16; ++l) cnt[l] = 0;
	for (int r = 0; r < n; r += WAVE) {
		const int s = r + lane;
		const int l = s < n ? (int)lens[s] : 0;
#pragma unroll
		for (int k = 1; k < 16; ++k) cnt[k] += (uint32_t)__popcll(__ballot(l == k));
	}
	int left = 1;
	uint32_t code = 0, offs = 0, cursor[16];
	bool over = false;
#pragma unroll
	for (int k = 1; k < 16; ++k) {
		left = (left << 1) - (int)cnt[k];
		over = over || left < 0;
		if (lane == 0) { L.cnt[set][k] = (uint16_t)cnt[k]; L.first[set][k] = (uint16_t)code; L.off[set][k] = (uint16_t)offs; }
		cursor[k] = offs;
		code = (code + cnt[k]) << 1; offs += cnt[k];
	}
	if (over) return INF_E_OVERSUB;
	if (left > 0 && offs > 0 && (rule == HUFF_CODES || (rule == HUFF_DATA && !(offs == 1 && cnt[1] == 1)))) return INF_E_CODE;
	__syncthreads();
	for (int r = 0; r < n; r += WAVE) {
		const int s = r + lane;
		const int l = s < n ? (int)lens[s] : 0;
		uint32_t idx = 0;
#pragma unroll
		for (int k = 1; k < 16; ++k) {
			const uint64_t m = __ballot(l == k);
			if (l == k) idx = cursor[k] + (uint32_t)__popcll(m & lanemask_lt());
			cursor[k] += (uint32_t)__popcll(m);
		}
		if (l) {
			L.perm(set)[idx] = (uint16_t)s;
			if (l <= T) {
				const uint32_t c = (uint32_t)L.first[set][l] + (idx - (uint32_t)L.off[set][l]);
				const uint32_t rev = __brev(c) >> (32 - l);
				const uint16_t ent = (uint16_t)wv_entry(set, (uint32_t)s, (uint32_t)l);
				for (uint32_t k = rev; k < (1u << T); k += 1u << l) tab[k] = ent;
			}
		}
	}
	__syncthreads();
	return INF_OK;
}

// the entry of the code that starts the 32 bits `bits` (0: none); T = the look-up's bits
template <int T>
__device__ __forceinline__ uint32_t wv_lookup(const WaveLds &L, int set, const uint16_t *tab, uint32_t bits)
{
	uint32_t e = tab[bits & ((1u << T) - 1u)];
	if (e == 0u && T < 15) { // a code longer than the table's index, or none: the canonical walk over the remaining lengths (kept small: it is rarely taken)
		const uint32_t rb = __brev(bits);
#pragma unroll 1
		for (int l = T + 1; l <= 15; ++l) {
			const uint32_t d = (rb >> (32 - l)) - (uint32_t)L.first[set][l];
			if (d < (uint32_t)L.cnt[set][l]) { e = wv_entry(set, (uint32_t)L.perm(set)[(uint32_t)L.off[set][l] + d], (uint32_t)l); break; }
		}
	}
	return e;
}

enum : int { WV_OK = 0, WV_EOB = 1, WV_BAD = 2, WV_DIST = 3, WV_DEAD = 4, WV_END = 5 }; // WV_END: the input is over (INF_E_INPUT when the sequential decode gets there)

struct WaveSeg {
	uint32_t end;      // where the next symbol starts (behind the end-of-block code when flag == WV_EOB)
	uint32_t bytes;    // output bytes of the segment
	uint32_t toks;     // tokens, the first match's run taken as the literals of THIS segment before it
	uint32_t lead;     // literals before the first match (~0: no match in the segment)
	uint32_t lastend;  // output bytes up to and including the last match
	uint32_t steps;    // symbols decoded (the phase counters' view of how even the lanes' work is)
	int flag;
};

// Decode from bit `start` until a symbol starts at or behind `seg_end` (or the end-of-block code, or nonsense).  EMIT: write the literals to out[o0 ...] and
// the tokens to tok[t0 ...]; `carry` = literals pending in front of the segment (they belong to the first match's run).
template <bool EMIT>
__device__ __forceinline__ WaveSeg wv_decode(const WaveLds &L, uint32_t w0_bits, uint32_t start, uint32_t seg_end, uint32_t lim_bits, uint8_t *out, uint32_t o0, uint32_t *tok, uint32_t t0,
                                             uint32_t carry)
{
	WaveSeg r;
	r.bytes = 0; r.toks = 0; r.lead = ~0u; r.lastend = 0; r.steps = 0; r.flag = WV_OK;
	uint32_t run = 0;
	WvBits B;
	B.start(L, w0_bits, start);
	while (B.p < seg_end) {
		if (B.p >= lim_bits) { r.flag = WV_END; break; } // (a guessed start behind the payload, or a chain that runs off its end)
		++r.steps;
		B.need(L);
		const uint32_t e = wv_lookup<WV_LT>(L, 0, L.lit, B.peek());
		if (e == 0u) { r.flag = WV_BAD; break; }
		B.drop((int)(e & 15u));
		if (!(e & 16u)) { // a literal
			if (EMIT) out[o0 + r.bytes] = (uint8_t)(e >> 8);
			++r.bytes; ++run;
			continue;
		}
		const int xb = (int)((e >> 5) & 7u);
		if (xb >= 6) { r.flag = xb == 6 ? WV_EOB : WV_BAD; break; }
		const uint32_t mlen = (e >> 8) + 3u + (B.peek() & ((1u << xb) - 1u)); // (>= 18 bits were left: xb <= 5)
		B.drop(xb);
		B.need(L);
		const uint32_t e2 = wv_lookup<WV_DT>(L, 1, L.dst, B.peek());
		const int xd = (int)((e2 >> 4) & 15u);
		if (e2 == 0u || xd == 15) { r.flag = WV_BAD; break; }
		B.drop((int)(e2 & 15u));
		uint32_t dist = 0;
		if (EMIT) { const uint32_t ds = e2 >> 8; dist = (ds < 4u ? ds : ((2u + (ds & 1u)) << xd)) + 1u + (B.peek() & ((1u << xd) - 1u)); } // (>= 18 bits were left: xd <= 13)
		B.drop(xd);
		const bool firstm = r.lead == ~0u;
		if (firstm) r.lead = run;
		uint32_t rr = run + (EMIT && firstm ? carry : 0u);
		if (EMIT) {
			if (dist > o0 + r.bytes) { r.flag = WV_DIST; break; }
			if (rr > 510u) { tok[t0 + r.toks] = 0xff800000u | (rr - 510u); ++r.toks; rr = 510u; } // (TokenOut::copy's escape; a block is < 8 M bytes: one is enough)
			tok[t0 + r.toks] = (mlen - 3u) | ((dist - 1u) << 8) | (rr << 23);
			++r.toks;
		} else r.toks += rr > 510u ? 2u : 1u;
		r.bytes += mlen; run = 0; r.lastend = r.bytes;
	}
	const uint32_t p = B.p;
	r.end = p;
	return r;
}

__device__ __forceinline__ uint32_t wv_excl_sum(uint32_t v, uint32_t &total)
{
	const uint32_t inc = wave_inclusive_sum(v);
	total = (uint32_t)__shfl((int)inc, WAVE - 1, WAVE);
	return inc - v;
}

__device__ __forceinline__ uint32_t wv_excl_max(uint32_t v) // max over the lanes below (0 for lane 0)
{
	uint32_t x = v;
#pragma unroll
	for (int d = 1; d < WAVE; d <<= 1) {
		const uint32_t y = (uint32_t)__shfl_up((int)x, d, WAVE);
		if ((int)threadIdx.x >= d) x = x > y ? x : y;
	}
	const uint32_t up = (uint32_t)__shfl_up((int)x, 1, WAVE);
	return threadIdx.x == 0 ? 0u : up;
}

// one raw deflate stream by one wavefront: literals to out, a token per match to tok (token_capacity(out_len) words); n_tok = tokens written
// dbg (SSV_INFLATE_PHASES=1): [0] deflate blocks, [1] windows, [2] decode rounds before the chain stood, [3] lanes that decoded in rounds after the second,
// [4..8] cycles: window loads, headers + tables, first round, later rounds, emit
template <bool DBG>
__device__ __forceinline__ int wave_inflate_tokens(WaveLds &L, const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_len, uint32_t *tok, uint32_t &n_tok, unsigned long long *dbg_arg)
{
	unsigned long long *const dbg = DBG ? dbg_arg : nullptr;
	unsigned long long d_blocks = 0, d_win = 0, d_rounds = 0, d_late = 0, c_load = 0, c_hdr = 0, c_r1 = 0, c_rn = 0, c_emit = 0;
	auto tick = [&]() { return dbg ? (unsigned long long)__builtin_readcyclecounter() : 0ull; };
	struct Flush {
		unsigned long long *dbg, *v[9];
		__device__ ~Flush() { if (dbg && threadIdx.x == 0) for (int k = 0; k < 9; ++k) atomicAdd(dbg + k, *v[k]); }
	} flush{dbg, {&d_blocks, &d_win, &d_rounds, &d_late, &c_load, &c_hdr, &c_r1, &c_rn, &c_emit}};

	const int lane = (int)threadIdx.x;
	const uint32_t lim_bits = in_len * 8u;
	uint32_t cur = 0, o = 0, nt = 0, last = 0;
	int final_block = 0;
	do {
		++d_blocks;
		unsigned long long tc = tick();
		uint32_t w0 = wv_load_window(L, in, in_len, cur);
		if (cur + 3u > lim_bits) return INF_E_INPUT;
		const uint32_t h = wv_fetch(L, w0, cur);
		final_block = (int)(h & 1u);
		const int type = (int)((h >> 1) & 3u);
		cur += 3u;
		if (type == 3) return INF_E_BTYPE;
		if (type == 0) { // stored: to the byte boundary, LEN, ~LEN, bytes (read where they lie)
			const uint32_t at = (cur + 7u) >> 3;
			if (at + 4u > in_len) return INF_E_INPUT;
			const uint32_t len = (uint32_t)in[at] | ((uint32_t)in[at + 1] << 8), nlen = (uint32_t)in[at + 2] | ((uint32_t)in[at + 3] << 8);
			if ((len ^ 0xffffu) != nlen) return INF_E_STORED;
			if (at + 4u + len > in_len) return INF_E_INPUT;
			if (o + len > out_len) return INF_E_OUTPUT;
			for (uint32_t i = (uint32_t)lane; i < len; i += WAVE) out[o + i] = in[at + 4u + i];
			o += len;
			cur = (at + 4u + len) * 8u;
			continue;
		}
		int nlen = 288, ndist = 30;
		if (type == 1) {
			for (int s = lane; s < 320; s += WAVE) L.lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5);
			__syncthreads();
		} else {
			const uint32_t hd = wv_fetch(L, w0, cur);
			nlen = (int)(hd & 31u) + 257; ndist = (int)((hd >> 5) & 31u) + 1;
			const int ncode = (int)((hd >> 10) & 15u) + 4;
			cur += 14u;
			if (nlen > 286 || ndist > 30) return INF_E_CODE;
			if (lane < 32) L.cll[lane] = 0;
			__syncthreads();
			if (lane < ncode) {
				// order of the code-length code lengths: 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
				const int k = lane, s = k < 3 ? 16 + k : k == 3 ? 0 : (k & 1) ? 8 - ((k - 3) >> 1) : 8 + ((k - 4) >> 1);
				L.cll[s] = (uint8_t)(wv_fetch(L, w0, cur + 3u * (uint32_t)k) & 7u);
			}
			cur += 3u * (uint32_t)ncode;
			__syncthreads();
			int rc = wv_build<WV_CT>(L, 2, L.cll, 19, L.clt, HUFF_CODES);
			if (rc != INF_OK) return rc;
			// the code lengths: a chain (every code's place depends on the one before), walked by all lanes in step
			int idx = 0, prev = 0;
			WvBits H;
			H.start(L, w0, cur);
			while (idx < nlen + ndist) {
				if (H.p >= lim_bits) return INF_E_INPUT;
				H.need(L); // (the header of a block - at most 2,300 bits - lies inside the window that was loaded at its first bit)
				const uint32_t e = L.clt[H.peek() & ((1u << WV_CT) - 1u)];
				const int l = (int)(e & 15u), s = (int)(e >> 4);
				if (l == 0) return INF_E_CODE;
				H.drop(l);
				const uint32_t x = H.peek();
				if (s < 16) { if (lane == 0) L.lens[idx] = (uint8_t)s; ++idx; prev = s; continue; }
				int rep, val = 0;
				if (s == 16) { if (idx == 0) return INF_E_REPEAT; val = prev; rep = 3 + (int)(x & 3u); H.drop(2); }
				else if (s == 17) { rep = 3 + (int)(x & 7u); H.drop(3); }
				else { rep = 11 + (int)(x & 127u); H.drop(7); }
				if (idx + rep > nlen + ndist) return INF_E_REPEAT;
				if (lane < rep) L.lens[idx + lane] = (uint8_t)val;
				if (lane + WAVE < rep) L.lens[idx + lane + WAVE] = (uint8_t)val; // (a run is at most 138 long)
				if (lane + 2 * WAVE < rep) L.lens[idx + lane + 2 * WAVE] = (uint8_t)val;
				idx += rep; prev = val;
			}
			cur = H.p;
			__syncthreads();
			if (L.lens[256] == 0) return INF_E_CODE; // no end-of-block code
		}
		int rc = wv_build<WV_DT>(L, 1, L.lens + nlen, ndist, L.dst, type == 1 ? HUFF_ANY : HUFF_DATA);
		if (rc != INF_OK) return rc;
		rc = wv_build<WV_LT>(L, 0, L.lens, nlen, L.lit, type == 1 ? HUFF_ANY : HUFF_DATA);
		if (rc != INF_OK) return rc;
		c_hdr += tick() - tc;
		// ---- the block's symbols, a window at a time ----
		for (;;) {
			++d_win;
			tc = tick();
			w0 = wv_load_window(L, in, in_len, cur);
			c_load += tick() - tc; tc = tick();
			const uint32_t seg_end = w0 + (uint32_t)(lane + 1) * WV_SEG_BITS;
			uint32_t start = lane == 0 ? cur : w0 + (uint32_t)lane * WV_SEG_BITS + (uint32_t)(WV_SEG_BITS - WV_TAIL_BITS);
			WaveSeg r = wv_decode<false>(L, w0, start, seg_end, lim_bits, nullptr, 0, nullptr, 0, 0);
			c_r1 += tick() - tc; tc = tick();
			if (DBG && dbg) { const uint32_t mx = wave_max(r.steps), sm = wave_sum(r.steps); if (lane == 0) { atomicAdd(dbg + 10, (unsigned long long)mx); atomicAdd(dbg + 11, (unsigned long long)sm); } }
			int round = 1;
			int t = WAVE - 1; // last lane whose symbols count
			bool stop = false;
			for (;;) {
				++round;
				// A lane is CONFIRMED when it started where the lane before it - confirmed itself, and simply going on - ended (lane 0 always is).  The confirmed
				// lanes are the sequential decode.  The window is done when they are all 64, or when one of them meets the end-of-block code (or nonsense):
				// what the lanes behind that one made of their guesses does not matter.  Every round confirms at least one lane more.
				const uint32_t pend = (uint32_t)__shfl_up((int)r.end, 1, WAVE);
				const int pflag = __shfl_up(r.flag, 1, WAVE);
				const uint64_t linked = __ballot(lane == 0 || (pflag == WV_OK && start == pend));
				const int nconf = ~linked ? __ffsll((long long)~linked) - 1 : WAVE; // lanes [0, nconf) are confirmed
				const uint64_t ends = __ballot(r.flag != WV_OK) & (nconf == WAVE ? ~0ull : (1ull << nconf) - 1ull);
				if (ends) { t = __ffsll((long long)ends) - 1; stop = true; break; }
				if (nconf == WAVE) break;
				// the first lane that is not confirmed takes its start from the confirmed one before it; the others behind it take theirs from lanes that
				// are guesses themselves - usually right already (a decoder falls into step within a few symbols), so that most of them are confirmed next round
				bool changed = false;
				if (lane >= nconf && pflag == WV_OK && start != pend) {
					start = pend; changed = true;
					r = wv_decode<false>(L, w0, start, seg_end, lim_bits, nullptr, 0, nullptr, 0, 0);
				}
				++d_rounds;
				if (round > 2) d_late += (unsigned long long)__popcll(__ballot(changed));
			}
			c_rn += tick() - tc; tc = tick();
			if (dbg && lane == 0) atomicMax(dbg + 9, (unsigned long long)round);
			const int tflag = __shfl(r.flag, t, WAVE);
			if (stop && tflag != WV_EOB) return tflag == WV_END ? INF_E_INPUT : INF_E_CODE;
			const bool mine = lane <= t;
			const uint32_t bytes = mine ? r.bytes : 0u;
			uint32_t total;
			const uint32_t o0 = o + wv_excl_sum(bytes, total);
			if (o + total > out_len) return INF_E_OUTPUT;
			const bool has = mine && r.lead != ~0u;
			const uint32_t after = has ? o0 + r.lastend : 0u;       // output position behind this lane's last match
			const uint32_t before = max(last, wv_excl_max(after));  // ... behind the last match in front of this lane
			const uint32_t carry = o0 - before;
			uint32_t toks = mine ? r.toks : 0u;
			if (has) toks = toks - (r.lead > 510u ? 1u : 0u) + (r.lead + carry > 510u ? 1u : 0u);
			uint32_t ttotal;
			const uint32_t t0 = nt + wv_excl_sum(toks, ttotal);
			int bad = 0;
			if (mine) {
				const WaveSeg e = wv_decode<true>(L, w0, start, seg_end, lim_bits, out, o0, tok, t0, carry);
				bad = e.flag == WV_DIST ? 1 : (e.bytes != r.bytes || e.toks != toks) ? 2 : 0;
			}
			if (__any(bad == 1)) return INF_E_DIST;
			if (__any(bad == 2)) return INF_E_CODE; // (cannot happen: both passes decode the same bits)
			c_emit += tick() - tc;
			o += total; nt += ttotal;
			last = max(last, wave_max(after));
			const uint32_t tend = (uint32_t)__shfl((int)r.end, t, WAVE);
			cur = tend;
			if (stop) break; // the end-of-block code
		}
	} while (!final_block);
	n_tok = nt;
	if (o != out_len) return INF_E_OUTPUT;
	if (((cur + 7u) >> 3) > in_len) return INF_E_INPUT;
	return INF_OK;
}

} // namespace ssv
