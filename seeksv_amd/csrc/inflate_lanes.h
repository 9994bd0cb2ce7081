// inflate_lanes.h - the DEFLATE decoder of inflate_core.h as a per-lane STATE MACHINE over two small rings (input, output).
//
// Why: with one lane per BGZF block every vector memory instruction of the decoder carries 64 unrelated addresses, and an LZ77 copy is a chain
// of loads that depend on stores the lane has just made - a round trip to memory each.  Real reads (mostly literals and short, near matches)
// inflate at 25-30 GB/s that way.  Here a lane touches global memory only for far matches:
//   * its compressed bytes come out of an input ring (IN_DW dwords) that the WAVEFRONT refills, half a ring at a time, with one coalesced
//     load per lane in need (sixteen lanes x 4 bytes of that lane's stream);
//   * its output goes into an output ring (OUT_DW dwords); matches up to NEAR_MAX bytes back are copied ring to ring; the wavefront writes
//     half a ring at a time to the inflated stream, again one coalesced store per lane in need;
//   * matches from further back read the inflated stream (those bytes were flushed steps ago).
// One `step()` = at most one symbol and at most PIECE bytes of a pending copy, so the lanes of a wavefront stay in the same few code paths
// (decode, copy) instead of each running its own loop nest; block headers are parsed in one go by the lanes that meet one.
// Everything is expressed against an `Io` policy (where the rings and the streams live), so that the same source runs on the GPU (rings in
// LDS, interleaved by lane) and in the CPU test against zlib (tests/native/inflate_check.cpp).
#pragma once

#include "inflate_core.h"

namespace ssv {

template <int IN_DW_, int OUT_DW_, int PIECE_ = 16> struct RingCfg {
	static constexpr int IN_DW = IN_DW_, OUT_DW = OUT_DW_;
	static constexpr int REFILL_DW = IN_DW_ / 2, FLUSH_DW = OUT_DW_ / 2;
	static constexpr int PIECE = PIECE_;                     // bytes of a match copied per step
	static constexpr int NEAR_MAX = 4 * OUT_DW_ - PIECE - 4; // a match from at most this far back is still in the ring while it is copied
	// a step starts with fewer than 4 * FLUSH_DW unflushed bytes and adds at most PIECE:
	static_assert(4 * FLUSH_DW - 1 + PIECE <= 4 * OUT_DW_, "the ring must hold them");
	static_assert(NEAR_MAX + PIECE <= 4 * OUT_DW_, "a near match must not be overwritten while it is copied");
	static_assert(4 * FLUSH_DW - 1 + 2 * PIECE <= NEAR_MAX + 1, "a far match (its piece and the next one) must lie in what has been written out");
	static_assert(IN_DW_ >= 4 && IN_DW_ % 2 == 0 && OUT_DW_ % 2 == 0 && PIECE_ % 4 == 0, "ring geometry");
};

enum : int { ST_HEADER = 0, ST_SYMBOL = 1, ST_COPY = 2, ST_STORED = 3, ST_FAR = 4, ST_FINISH = 5, ST_DONE = 6 };

SSV_HD uint32_t align_bytes(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8 * sh)); }

// What the helpers of a far move need to know of the lane they help (see LaneInflate, ST_FAR).
template <class Cfg> struct FarMove {
	static constexpr uint32_t FAR_DW = 64; // destination dwords of one move at most
	static_assert(4 * Cfg::FLUSH_DW + Cfg::PIECE + 8 <= 4 * 64 - 16, "a move must have room for match bytes behind the unwritten ring bytes");
	uint32_t f, o, rem, dist, o_begin;
	SSV_HD uint32_t chunk() const
	{
		uint32_t c = rem;
		const uint32_t lim = dist - (o - f);
		if (c > lim) c = lim;
		const uint32_t room = 4u * FAR_DW - (o - (f & ~3u)) - 4u;
		return c > room ? room : c;
	}
	SSV_HD uint32_t dwords() const { return ((o + chunk() + 3u) >> 2) - (f >> 2); }
	SSV_HD uint32_t pos(uint32_t j) const { return (f & ~3u) + 4u * j; } // position of helper j's dword
	SSV_HD bool needs_src(uint32_t j) const { const uint32_t q0 = pos(j); return q0 + 4u > o && q0 < o + chunk(); }
	// helper j: which bytes of its dword are written (bit b of the result) and their values; `ringv` = ring dword (pos(j) >> 2) % OUT_DW,
	// `srcv` = the four stream bytes at pos(j) - dist (only looked at where needs_src(j))
	SSV_HD uint32_t merge(uint32_t j, uint32_t ringv, uint32_t srcv, uint32_t &value) const
	{
		const uint32_t q0 = pos(j), o1 = o + chunk();
		uint32_t mask = 0, v = 0;
		for (uint32_t b = 0; b < 4; ++b) {
			const uint32_t q = q0 + b;
			if (q < o_begin || q >= o1) continue;
			mask |= 1u << b;
			v |= (q < o ? (ringv >> (8 * b)) & 0xffu : (srcv >> (8 * b)) & 0xffu) << (8 * b);
		}
		value = v;
		return mask;
	}
	// does helper j's dword go back into the ring (it holds match bytes and is among the last OUT_DW dwords of the output so far)?
	SSV_HD bool to_ring(uint32_t j) const
	{
		const uint32_t q0 = pos(j), o1 = o + chunk();
		return q0 + 4u > o && q0 < o1 && ((o1 - 1u) >> 2) - (q0 >> 2) < (uint32_t)Cfg::OUT_DW;
	}
};

// State of one stream.  Offsets on the input side count from `in_org`, the dword-aligned address at or below the stream's first byte;
// positions on the output side count from `out_org`, the dword-aligned address at or below the block's first output byte.
template <class Cfg> struct LaneInflate {
	// input
	uint32_t p = 0;         // offset of `ahead`
	uint32_t in_lim = 0;    // fetches start below this offset (stream end + 4): beyond it the decoder is fed zero bits
	uint32_t in_first = 0, in_len = 0;
	uint32_t rfill = 0;     // dwords [max(0, rfill - IN_DW), rfill) of the stream are in the ring, dword d at slot d % IN_DW
	uint64_t bb = 0;
	int bc = 0;
	uint32_t ahead = 0;
	// output
	uint32_t o = 0, o_begin = 0, o_end = 0; // next position, the block's first and end position
	uint32_t f = 0;                          // positions below f (a multiple of 4) have been written to the inflated stream
	// decoder
	int state = ST_DONE, last = 0, rc = INF_OK;
	uint32_t rem = 0, dist = 0;             // pending copy (ST_COPY) / stored bytes (ST_STORED)
	HuffCounts lit, dst;

	SSV_HD void start(uint32_t in_misalign, uint32_t in_len_, uint32_t out_misalign, uint32_t out_len)
	{
		p = in_first = in_misalign; in_len = in_len_; in_lim = in_misalign + in_len_ + 4; rfill = 0; bb = 0; bc = 0; ahead = 0;
		o = o_begin = out_misalign; o_end = out_misalign + out_len; f = out_misalign & ~3u;
		state = out_len ? ST_HEADER : ST_DONE; last = 0; rc = INF_OK; rem = dist = 0;
	}
	SSV_HD void fail(int code) { rc = code; state = ST_DONE; }
	SSV_HD bool wants_refill() const { return state < ST_FINISH && rfill - (p >> 2) <= (uint32_t)(Cfg::IN_DW - Cfg::REFILL_DW) && 4 * rfill < in_lim + 4; }
	SSV_HD bool wants_flush() const { return state == ST_FINISH || (state < ST_FAR && o - f >= 4u * Cfg::FLUSH_DW); }

	// ---- bits ----
	template <class Io> SSV_HD void fetch(Io &io)
	{
		const uint32_t d = p >> 2;
		if (p >= in_lim) ahead = 0;
		else if (d + 1 < rfill) ahead = align_bytes(io.in_get((int)((d + 1) % Cfg::IN_DW)), io.in_get((int)(d % Cfg::IN_DW)), p & 3u);
		else ahead = io.in_stream32(p); // the ring has run dry (block headers are read in one go): the lane helps itself
	}
	template <class Io> SSV_HD void refill(Io &io)
	{
		if (bc < 32) { bb |= (uint64_t)ahead << bc; bc += 32; p += 4; fetch(io); }
	}
	SSV_HD uint32_t peek(int n) const { return (uint32_t)bb & ((1u << n) - 1u); }
	SSV_HD void drop(int n) { bb >>= n; bc -= n; }
	SSV_HD uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
	SSV_HD uint32_t consumed() const { return (p - in_first) - (uint32_t)(bc >> 3); }

	// canonical decode, as huff_decode of inflate_core.h (which works on a BitReader)
	template <bool LIT, class Tab> SSV_HD int decode(const HuffCounts &h, const Tab &tab, int sym_base)
	{
		int code = 0, first = 0, index = 0;
		uint32_t bits = (uint32_t)bb;
#pragma unroll
		for (int len = 1; len <= 15; ++len) {
			code |= (int)(bits & 1u); bits >>= 1;
			const int count = (int)((h.c[len >> 1] >> ((len & 1) * 16)) & 0xffffu);
			if (code - count < first) { drop(len); const int k = sym_base + index + (code - first); return LIT ? tab.lit_get(k) : tab.dst_get(k); }
			index += count; first += count; first <<= 1; code <<= 1;
		}
		return -1;
	}

	// ---- one block header, in one go (inflate_stream of inflate_core.h, the part before the symbol loop) ----
	template <class Io, class Tab> SSV_HD void header(Io &io, Tab &tab)
	{
		if (p == in_first && bc == 0) fetch(io); // the stream's first bits
		refill(io);
		last = (int)take(1);
		const int type = (int)take(2);
		if (type == 0) {
			drop(bc & 7);
			refill(io);
			const uint32_t len = take(16);
			refill(io);
			const uint32_t nlen = take(16);
			if ((len ^ 0xffffu) != nlen) return fail(INF_E_STORED);
			if (o + len > o_end) return fail(INF_E_OUTPUT);
			rem = len;
			state = len ? ST_STORED : (last ? ST_FINISH : ST_HEADER);
			return;
		}
		if (type == 3) return fail(INF_E_BTYPE);
		if (type == 1) {
			for (int s = 0; s < 288; ++s) tab.len_set(s, s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
			for (int s = 0; s < 30; ++s) tab.len_set(288 + s, 5);
			huff_construct<true>(tab, 0, 288, 0, lit);
			huff_construct<false>(tab, 288, 30, 0, dst);
		} else {
			const int nlen = (int)take(5) + 257, ndist = (int)take(5) + 1, ncode = (int)take(4) + 4;
			if (nlen > 286 || ndist > 30) return fail(INF_E_CODE);
			for (int s = 0; s < 19; ++s) tab.len_set(s, 0);
			for (int k = 0; k < ncode; ++k) {
				refill(io);
				// order of the code-length code lengths: 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
				const int s = k < 3 ? 16 + k : k == 3 ? 0 : (k & 1) ? 8 - ((k - 3) >> 1) : 8 + ((k - 4) >> 1);
				tab.len_set(s, (int)take(3));
			}
			HuffCounts cl;
			int r = huff_construct<false>(tab, 0, 19, 12, cl, HUFF_CODES);
			if (r != INF_OK) return fail(r);
			int idx = 0, prev = 0;
			while (idx < nlen + ndist) {
				refill(io);
				const int s = decode<false>(cl, tab, 12);
				if (s < 0) return fail(INF_E_CODE);
				if (s < 16) { tab.len_set(idx++, s); prev = s; continue; }
				int rep, val = 0;
				if (s == 16) { if (idx == 0) return fail(INF_E_REPEAT); val = prev; rep = 3 + (int)take(2); }
				else if (s == 17) rep = 3 + (int)take(3);
				else rep = 11 + (int)take(7);
				if (idx + rep > nlen + ndist) return fail(INF_E_REPEAT);
				while (rep--) tab.len_set(idx++, val);
				prev = val;
			}
			if (tab.len_get(256) == 0) return fail(INF_E_CODE); // no end-of-block code
			r = huff_construct<false>(tab, nlen, ndist, 0, dst, HUFF_DATA);
			if (r != INF_OK) return fail(r);
			r = huff_construct<true>(tab, 0, nlen, 0, lit, HUFF_DATA);
			if (r != INF_OK) return fail(r);
		}
		state = ST_SYMBOL;
	}

	// ---- one symbol ----
	template <class Io, class Tab> SSV_HD void symbol(Io &io, const Tab &tab)
	{
		refill(io);
		const int s = decode<true>(lit, tab, 0);
		if (s < 0) return fail(INF_E_CODE);
		if (s < 256) {
			if (o >= o_end) return fail(INF_E_OUTPUT);
			io.out_set8(o % (4u * Cfg::OUT_DW), (uint8_t)s);
			++o;
			return;
		}
		if (s == 256) { state = last ? ST_FINISH : ST_HEADER; return; }
		if (s > 285) return fail(INF_E_CODE);
		// length: 257..264 -> 3..10; then groups of four symbols share an extra-bit count; 285 -> 258
		uint32_t len;
		if (s < 265) len = (uint32_t)s - 254u;
		else if (s == 285) len = 258;
		else { const int e = (s - 261) >> 2; len = ((4u + (uint32_t)((s - 265) & 3)) << e) + 3u + take(e); }
		refill(io);
		const int d = decode<false>(dst, tab, 0);
		if (d < 0 || d > 29) return fail(INF_E_CODE);
		uint32_t back;
		if (d < 4) back = (uint32_t)d + 1u;
		else { const int e = (d >> 1) - 1; back = ((2u + (uint32_t)(d & 1)) << e) + 1u + take(e); }
		if (back > o - o_begin) return fail(INF_E_DIST);
		if (o + len > o_end) return fail(INF_E_OUTPUT);
		rem = len; dist = back; state = back > (uint32_t)Cfg::NEAR_MAX ? ST_FAR : ST_COPY;
	}

	// ---- up to PIECE bytes of the pending match ----
	// four bytes of the ring from any byte position (two aligned dwords), n <= 4 bytes into the ring at any byte position
	template <class Io> SSV_HD uint32_t ring_get32u(const Io &io, uint32_t pos) const
	{
		const uint32_t idx = pos % (4u * Cfg::OUT_DW), d = idx >> 2;
		return align_bytes(io.out_get32((int)((d + 1) % Cfg::OUT_DW)), io.out_get32((int)d), idx & 3u);
	}
	template <class Io> SSV_HD void ring_put(Io &io, uint32_t pos, uint32_t v, uint32_t n)
	{
		const uint32_t idx = pos % (4u * Cfg::OUT_DW);
		if (n == 4 && (idx & 3u) == 0) io.out_set32((int)(idx >> 2), v);
		else for (uint32_t b = 0; b < n; ++b) io.out_set8((pos + b) % (4u * Cfg::OUT_DW), (uint8_t)(v >> (8 * b)));
	}
	template <class Io> SSV_HD void copy_piece(Io &io)
	{
		const uint32_t n = rem < (uint32_t)Cfg::PIECE ? rem : (uint32_t)Cfg::PIECE;
		constexpr uint32_t RB = 4u * Cfg::OUT_DW;
		if (dist == 1) { // a run: one read, then only writes
			const uint32_t v = (uint32_t)io.out_get8((o - 1) % RB) * 0x01010101u;
#pragma unroll
			for (uint32_t k = 0; k < (uint32_t)Cfg::PIECE / 4; ++k) if (4 * k < n) ring_put(io, o + 4 * k, v, n - 4 * k < 4 ? n - 4 * k : 4u);
		} else if (dist >= 4) {
			if (dist >= (uint32_t)Cfg::PIECE) { // source and destination of the piece are disjoint: all reads, then all writes
				uint32_t w[Cfg::PIECE / 4];
#pragma unroll
				for (uint32_t k = 0; k < (uint32_t)Cfg::PIECE / 4; ++k) w[k] = 4 * k < n ? ring_get32u(io, o - dist + 4 * k) : 0u;
#pragma unroll
				for (uint32_t k = 0; k < (uint32_t)Cfg::PIECE / 4; ++k) if (4 * k < n) ring_put(io, o + 4 * k, w[k], n - 4 * k < 4 ? n - 4 * k : 4u);
			} else { // every source dword lies behind the write position: dword steps, in order
#pragma unroll
				for (uint32_t k = 0; k < (uint32_t)Cfg::PIECE / 4; ++k) if (4 * k < n) ring_put(io, o + 4 * k, ring_get32u(io, o - dist + 4 * k), n - 4 * k < 4 ? n - 4 * k : 4u);
			}
		} else {
			for (uint32_t i = 0; i < n; ++i) io.out_set8((o + i) % RB, io.out_get8((o + i - dist) % RB)); // period 2 or 3: byte by byte
		}
		o += n; rem -= n;
		if (!rem) state = ST_SYMBOL;
	}

	// ---- up to PIECE bytes of a stored block ----
	template <class Io> SSV_HD void stored_piece(Io &io)
	{
		const uint32_t n = rem < (uint32_t)Cfg::PIECE ? rem : (uint32_t)Cfg::PIECE;
		for (uint32_t i = 0; i < n; ++i) { refill(io); io.out_set8((o + i) % (4u * Cfg::OUT_DW), (uint8_t)take(8)); }
		o += n; rem -= n;
		if (!rem) state = last ? ST_FINISH : ST_HEADER;
	}

	// ---- a far match (ST_FAR): moved by whoever owns the rings, inflated stream -> inflated stream, in whole dwords of the destination ----
	// The move covers positions [f, o + chunk): the lane's unwritten ring bytes [f, o) and the first `chunk` bytes of the match
	// (chunk <= dist - (o - f): the source then ends below f, in what has been written out; at most FAR_DW destination dwords).
	// Destination dword j of the move (j < far_dwords()) is position q0 = (f & ~3) + 4 j; its byte b comes from the ring when q0 + b < o,
	// from the stream at q0 + b - dist when q0 + b < o + chunk, and is not written otherwise (far_merge tells).  The dwords that hold
	// match bytes also go into the ring (the last OUT_DW of them), so that later near matches find them there.
	SSV_HD FarMove<Cfg> far_view() const { FarMove<Cfg> m; m.f = f; m.o = o; m.rem = rem; m.dist = dist; m.o_begin = o_begin; return m; }
	SSV_HD void far_done()
	{
		const uint32_t c = far_view().chunk();
		o += c; rem -= c; f = o & ~3u;
		if (!rem) state = ST_SYMBOL;
	}

	// one step of the machine (refill and flush of the rings, and far matches, happen between steps, by whoever owns the rings)
	template <class Io, class Tab> SSV_HD void step(Io &io, Tab &tab)
	{
		if (state == ST_HEADER) header(io, tab);
		if (state == ST_SYMBOL) symbol(io, tab);
		if (state == ST_COPY) copy_piece(io);
		else if (state == ST_STORED) stored_piece(io);
	}

	// after the last flush: did the stream deliver exactly the block, out of exactly its bytes?
	SSV_HD int verdict() const
	{
		if (rc != INF_OK) return rc;
		if (o != o_end) return INF_E_OUTPUT;
		if (consumed() > in_len) return INF_E_INPUT;
		return INF_OK;
	}
};

} // namespace ssv
