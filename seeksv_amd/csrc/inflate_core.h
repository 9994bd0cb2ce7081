// inflate_core.h - a DEFLATE (RFC 1951) decoder written for one GPU lane per BGZF block.
//
// the reference spends 78 % of getclip inside libbam's samread() = zlib inflate + record parse on one core.  BGZF
// blocks (<= 64 KB each) are independent deflate streams, so a BAM file offers millions of them; here every lane of a wavefront
// decodes its own block.  What shapes the code:
//   * no per-lane arrays in registers or scratch: the Huffman decode is canonical (count-per-length walk, as in Mark Adler's `puff`
//     description of the algorithm) with the 15 per-length counts packed into 8 registers and only the symbol permutation in a
//     table; tables live behind an accessor `Tab` (LDS interleaved by lane on the GPU, plain arrays in the CPU test build);
//   * length/distance bases are computed from the symbol, not looked up (a divergent constant-table index is a waterfall loop);
//   * the bit buffer is refilled 32 bits at a time with one unaligned load.
// The same source is compiled by g++ for tests/test_inflate_core.py (against zlib) - which is why nothing here is HIP specific.
#pragma once

#include <stdint.h>
#include <string.h>
#include <stddef.h>

#ifndef SSV_HD
#ifdef __HIPCC__
#define SSV_HD __host__ __device__ __forceinline__
#else
#define SSV_HD inline
#endif
#endif

namespace ssv {

enum : int { INF_OK = 0, INF_E_BTYPE = -1, INF_E_STORED = -2, INF_E_CODE = -3, INF_E_OVERSUB = -4, INF_E_DIST = -5, INF_E_OUTPUT = -6, INF_E_INPUT = -7, INF_E_REPEAT = -8 };

// Table storage for one stream: the literal/length permutation (288 symbols of 9 bits), the distance permutation (32 symbols; the
// code-length code borrows its slots [12,31) while a block header is read), 320 code lengths, 16 running offsets.
struct PlainTab {
	uint16_t lit[288]; uint8_t dst[32]; uint8_t len[320]; uint16_t off[16];
	SSV_HD uint16_t lit_get(int i) const { return lit[i]; }
	SSV_HD void lit_set(int i, uint16_t v) { lit[i] = v; }
	SSV_HD uint16_t dst_get(int i) const { return dst[i]; }
	SSV_HD void dst_set(int i, uint16_t v) { dst[i] = (uint8_t)v; }
	SSV_HD int len_get(int i) const { return len[i]; }
	SSV_HD void len_set(int i, int v) { len[i] = (uint8_t)v; }
	SSV_HD uint16_t off_get(int i) const { return off[i]; }
	SSV_HD void off_set(int i, uint16_t v) { off[i] = v; }
	static constexpr bool has_base = false; // true: the table also keeps base[set][length] and symbols are decoded by limit compares (huff_decode_lim)
};
struct PlainTabLim : PlainTab {
	int16_t base[2][16];
	static constexpr bool has_base = true;
	SSV_HD int base_get(int set, int l) const { return base[set][l]; }
	SSV_HD void base_set(int set, int l, int v) { base[set][l] = (int16_t)v; }
};

struct BitReader {
	const uint8_t *p;   // address of `ahead`
	const uint8_t *lim; // loads start below this address: the stream's end + 4 (buffers keep 8 spare bytes behind the last stream)
	uint64_t bb = 0;    // bit buffer, LSB first
	int bc = 0;         // valid bits in bb
	uint32_t ahead;     // the next 32 input bits, loaded one refill early: its latency hides behind the symbols decoded meanwhile
	bool ahead_in;      // ... and whether they lie inside the stream
	SSV_HD BitReader(const uint8_t *in, uint32_t in_len) : p(in), lim(in + in_len + 4) { fetch(); }
	// a damaged or crafted stream can ask for more input than it has: past the end it is fed zero bits (which every path of the decoder
	// turns into an error or into output that hits the output bound) instead of whatever lies behind the buffer
	SSV_HD void fetch() // (no branch around the load: it is on every symbol's path)
	{
		ahead_in = p < lim;
		memcpy(&ahead, ahead_in ? p : lim - 4, 4); // lim - 4 = the stream's end: readable (the buffers keep spare bytes behind the last stream)
		// (whether the word counts is decided where it is USED: a select right here would make the lane wait for the load at once)
	}
	SSV_HD void refill() // afterwards bc >= 32 (looks up to 8 bytes past the data: buffers are padded)
	{
		if (bc < 32) {
			bb |= (uint64_t)(ahead_in ? ahead : 0u) << bc;
			bc += 32; p += 4;
			fetch();
		}
	}
	SSV_HD uint32_t peek(int n) const { return (uint32_t)bb & ((1u << n) - 1u); }
	SSV_HD void drop(int n) { bb >>= n; bc -= n; }
	SSV_HD uint32_t take(int n) { uint32_t v = peek(n); drop(n); return v; }
	SSV_HD uint32_t consumed(const uint8_t *in) const { return (uint32_t)(p - in) - (uint32_t)(bc >> 3); } // whole bytes taken from the input
	SSV_HD void top_of_symbol() {}
};

// RingReader: the same bit buffer fed from a small window of the input that the lane keeps in fast storage (LDS on the device, `Ring` says where).
// Why: on gfx950 a wavefront's loads and stores share ONE counter (vmcnt), so a lane that waits for its next input word also waits for every
// store the wavefront has in flight - with BitReader that is each symbol's literal / token store, every step.  Here the symbol loop only touches
// the window; the window is topped up from memory at the top of a symbol (top_of_symbol) by all lanes of the wavefront together whenever ANY of
// them runs low, so the wait for memory comes once per a few dozen symbols instead of once per symbol.
//   Ring: uint32_t get(uint32_t j), void set(uint32_t j, uint32_t v) for j < CAP / 4; bool any(bool) = "any lane of the wavefront" (identity on a CPU).
// Loads are whole aligned 16-byte groups from the dword boundary below the stream's first byte and may reach 16 + 3 bytes past its end.
template <class Ring, uint32_t CAP = 128>
struct RingReader {
	static_assert(CAP >= 64 && (CAP & (CAP - 1)) == 0, "the window is a power of two, at least four groups");
	static constexpr uint32_t LOW = 24; // a symbol takes at most 48 bits: with this much left two symbols are safe
	Ring ring;
	const uint8_t *base; // the dword boundary at or below the stream's first byte
	uint32_t lim;        // groups start below this offset from base; what lies behind is fed as zero bits (see BitReader)
	uint32_t skew;       // bytes between base and the stream
	uint32_t rd = 0;     // offset of the next dword for the bit buffer
	uint32_t wr = 0;     // offset of the next group to load (wr - rd <= CAP)
	uint64_t bb = 0;
	int bc = 0;
	SSV_HD RingReader(const uint8_t *in, uint32_t in_len, Ring r) : ring(r)
	{
		skew = (uint32_t)(reinterpret_cast<uintptr_t>(in) & 3u);
		base = in - skew;
		lim = in_len + skew + 4u;
		top_up();
		refill();
		drop(8 * (int)skew);
	}
	SSV_HD void group(uint32_t at, uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d) const
	{
		a = b = c = d = 0u;
		if (at < lim) { const uint32_t *q = reinterpret_cast<const uint32_t *>(base + at); a = q[0]; b = q[1]; c = q[2]; d = q[3]; }
	}
	SSV_HD void put(uint32_t at, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
	{
		const uint32_t j = (at >> 2) & (CAP / 4u - 1u);
		ring.set(j, a); ring.set(j + 1u, b); ring.set(j + 2u, c); ring.set(j + 3u, d);
	}
	SSV_HD void top_up() // every group that fits, up to four: all the loads first
	{
		const uint32_t k = (CAP - (wr - rd)) >> 4;
		uint32_t a0, b0, c0, d0, a1, b1, c1, d1, a2, b2, c2, d2, a3, b3, c3, d3;
		if (k > 0u) group(wr, a0, b0, c0, d0);
		if (k > 1u) group(wr + 16u, a1, b1, c1, d1);
		if (k > 2u) group(wr + 32u, a2, b2, c2, d2);
		if (k > 3u) group(wr + 48u, a3, b3, c3, d3);
		if (k > 0u) put(wr, a0, b0, c0, d0);
		if (k > 1u) put(wr + 16u, a1, b1, c1, d1);
		if (k > 2u) put(wr + 32u, a2, b2, c2, d2);
		if (k > 3u) put(wr + 48u, a3, b3, c3, d3);
		wr += 16u * (k < 4u ? k : 4u);
	}
	SSV_HD void top_of_symbol() { if (ring.any(wr - rd < LOW)) top_up(); }
	SSV_HD void refill()
	{
		if (bc < 32) {
			if (rd == wr) { uint32_t a, b, c, d; group(wr, a, b, c, d); put(wr, a, b, c, d); wr += 16u; } // (outside the symbol loop: headers, stored blocks)
			bb |= (uint64_t)ring.get((rd >> 2) & (CAP / 4u - 1u)) << bc;
			bc += 32; rd += 4u;
		}
	}
	SSV_HD uint32_t peek(int n) const { return (uint32_t)bb & ((1u << n) - 1u); }
	SSV_HD void drop(int n) { bb >>= n; bc -= n; }
	SSV_HD uint32_t take(int n) { uint32_t v = peek(n); drop(n); return v; }
	SSV_HD uint32_t consumed(const uint8_t *) const { return rd - skew - (uint32_t)(bc >> 3); }
};

// LZ77 copy of len bytes from `dist` bytes back.  Measured on MI355X (profiles/r01_bamdec_*.json): the lane is latency bound and the
// wavefront runs every divergent path its lanes take, so the copy is kept to few, short paths:
//   head         up to three bytes bring the destination to a dword boundary (hipcc turns unaligned dword STORES into four byte
//                stores; unaligned dword loads stay single instructions)
//   dist >= len  source and destination are disjoint: 16 bytes per step, the four loads issued before the four (merged) stores
//   dist >= 4    overlapping, but every source dword lies behind the write position: dword steps
//   dist 1..3    a short period: the pattern is built in registers once and only stored
// Tried and slower here: issuing all loads of up to 32 bytes before the first store (more instructions per match: 37.6 ms vs 27.4 ms
// for 5.3 GB), and the same with loads running ahead of the write position (those lines are not in L2 yet: 45 ms).
SSV_HD uint32_t ld32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; } // unaligned load (one dword load on gfx950)

SSV_HD void lz_copy_disjoint(uint32_t *__restrict__ to, const uint8_t *__restrict__ from, uint32_t len)
{
	uint32_t i = 0;
	for (; i + 16 <= len; i += 16) {
		const uint32_t a = ld32(from + i), b = ld32(from + i + 4), c = ld32(from + i + 8), d = ld32(from + i + 12);
		to[i >> 2] = a; to[(i >> 2) + 1] = b; to[(i >> 2) + 2] = c; to[(i >> 2) + 3] = d;
	}
	for (; i + 4 <= len; i += 4) to[i >> 2] = ld32(from + i);
	uint8_t *tb = reinterpret_cast<uint8_t *>(to);
	for (; i < len; ++i) tb[i] = from[i];
}

SSV_HD void lz_copy(uint8_t *to_, uint32_t dist, uint32_t len);

// one dword store to ANY address: hipcc splits a store it cannot prove aligned into four byte stores, the hardware does not need that
SSV_HD void st32u(uint8_t *p, uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
#else
	memcpy(p, &v, 4);
#endif
}

SSV_HD void lz_copy(uint8_t *to_, uint32_t dist, uint32_t len)
{
	uint32_t head = (uint32_t)(0u - (uint32_t)(uintptr_t)to_) & 3u;
	if (head > len) head = len;
	const uint8_t *from_ = to_ - dist;
	for (uint32_t i = 0; i < head; ++i) to_[i] = from_[i];
	len -= head;
	if (!len) return;
	uint8_t *to = to_ + head;
	uint32_t *to4 = reinterpret_cast<uint32_t *>(to); // aligned: dword stores
	const uint8_t *from = to - dist;
	if (dist >= len) { lz_copy_disjoint(to4, from, len); return; }
	if (dist >= 4) {
		uint32_t i = 0;
		for (; i + 4 <= len; i += 4) to4[i >> 2] = ld32(from + i);
		for (; i < len; ++i) to[i] = from[i];
		return;
	}
	// period 1, 2 or 3: twelve bytes hold a whole number of periods (no runtime-indexed local arrays: they would live in scratch)
	const uint32_t a = from[0], b = dist > 1 ? from[1] : a, c = dist > 2 ? from[2] : a;
	uint32_t w0, w1, w2;
	if (dist == 3) { w0 = a | b << 8 | c << 16 | a << 24; w1 = b | c << 8 | a << 16 | b << 24; w2 = c | a << 8 | b << 16 | c << 24; }
	else { w0 = w1 = w2 = dist == 1 ? a * 0x01010101u : (a | b << 8) * 0x00010001u; }
	uint32_t i = 0;
	for (; i + 12 <= len; i += 12) { to4[i >> 2] = w0; to4[(i >> 2) + 1] = w1; to4[(i >> 2) + 2] = w2; }
	for (uint32_t k = 0; i < len; ++i, ++k) to[i] = (uint8_t)((k < 4 ? w0 : k < 8 ? w1 : w2) >> (8 * (k & 3)));
}

// ---- where the decoded bytes go ----
// DirectOut: straight into the inflated stream (a byte store per literal, lz_copy per match).
struct DirectOut {
	uint8_t *out;
	SSV_HD void put(uint32_t &o, uint8_t v) { out[o++] = v; }
	SSV_HD void copy(uint32_t &o, uint32_t dist, uint32_t len) { lz_copy(out + o, dist, len); o += len; }
	SSV_HD void finish(uint32_t) {}
};

// TokenOut: the decode split in two.  What makes a block slow is the chain decode -> copy -> decode: every match reads bytes of the
// block's own output, a trip to memory that the next symbol has to wait for.  Pass 1 (this sink) only DECODES: literals go to their final place,
// a match becomes a 32-bit token and leaves a hole; pass 2 (resolve_tokens / k_bgzf_resolve) fills the holes in order, several lanes per block.
//   token = len - 3 | (dist - 1) << 8 | literals since the previous token << 23     (literals 0..510)
//           0xff800000 | n: no match, n (< 2^23) more literals (a run of more than 510)
struct TokenOut {
	uint8_t *out;
	uint32_t *tok;       // room for token_capacity(out_len) words
	uint32_t n = 0;      // tokens written
	uint32_t last = 0;   // position behind the last token's match
	SSV_HD void put(uint32_t &o, uint8_t v) { out[o++] = v; }
	SSV_HD void copy(uint32_t &o, uint32_t dist, uint32_t len)
	{
		uint32_t run = o - last;
		while (run > 510u) { const uint32_t k = run - 510u < 0x7fffffu ? run - 510u : 0x7fffffu; tok[n++] = 0xff800000u | k; run -= k; }
		tok[n++] = (len - 3u) | ((dist - 1u) << 8) | (run << 23);
		o += len; last = o;
	}
	SSV_HD void finish(uint32_t) {}
};
// words a block of out_len bytes can need: a token per 3 bytes of output, an escape per 510 literals
SSV_HD uint32_t token_capacity(uint32_t out_len) { return out_len / 3u + out_len / 510u + 4u; }

// pass 2, one block, one thread (the CPU check; the kernel does the same with a group of lanes per block)
inline void resolve_tokens(uint8_t *out, const uint32_t *tok, uint32_t n)
{
	uint32_t pos = 0;
	for (uint32_t t = 0; t < n; ++t) {
		const uint32_t w = tok[t];
		if ((w >> 23) == 511u) { pos += w & 0x7fffffu; continue; }
		pos += w >> 23;
		const uint32_t len = (w & 255u) + 3u, dist = ((w >> 8) & 0x7fffu) + 1u;
		for (uint32_t i = 0; i < len; ++i) out[pos + i] = out[pos + i - dist];
		pos += len;
	}
}

struct HuffCounts { uint32_t c[8]; }; // count of codes of length L in bits [16*(L&1), +16) of c[L>>1]

// canonical decode: walk the lengths, one bit per step; the first length at which the code falls below first+count wins
// LIT selects the literal/length table; otherwise the distance table at slot offset sym_base
template <bool LIT, class Tab, class Reader>
SSV_HD int huff_decode(Reader &br, const HuffCounts &h, const Tab &tab, int sym_base)
{
	int code = 0, first = 0, index = 0, used = 0;
	uint32_t bits = (uint32_t)br.bb; // caller guarantees >= 15 valid bits
#pragma unroll
	for (int len = 1; len <= 15; ++len) {
		code |= (int)(bits & 1u); bits >>= 1;
		const int count = (int)((h.c[len >> 1] >> ((len & 1) * 16)) & 0xffffu);
		if (code - count < first) { used = len; break; }
		index += count; first += count; first <<= 1; code <<= 1;
	}
	// ONE table read behind the loop, whatever the length was: lanes of a wavefront leave the loop at different lengths, and a read inside each
	// exit would be a trip to the table per distinct length
	if (used == 0) return -1;
	br.drop(used);
	const int k = sym_base + index + (code - first);
	return LIT ? tab.lit_get(k) : tab.dst_get(k);
}

SSV_HD uint32_t brev32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __brev(x);
#else
	x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
	x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
	x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
	x = ((x >> 8) & 0x00ff00ffu) | ((x & 0x00ff00ffu) << 8);
	return (x >> 16) | (x << 16);
#endif
}
SSV_HD int popc32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __popc(x);
#else
	return __builtin_popcount(x);
#endif
}

// The same decode without the walk.  A wavefront executes the union of its lanes' paths: the bit-by-bit walk above costs it every length up to the
// longest code among 64 lanes plus an exit per distinct length (~140 instructions a symbol).  Canonical codes can be told apart by ONE comparison per
// length instead: with the next 15 bits read MSB first (`rev`), a code of length L is present iff rev < limit[L] = (first[L] + count[L]) << (15 - L),
// and the limits only grow with L - so the length is 1 + the number of limits <= rev.  The 15 limits (<= 0x8000) sit two to a register where the
// counts were (slot 0 holds limit 0: it always counts, that is the "1 +"); (rev | 0x8000) - limit has bit 15 set iff rev >= limit, in both halves of
// a register at once, no borrow between them; the 16 result bits are gathered and counted.  The symbol's index in the permutation is
// code + base[L], base[L] = (symbols shorter than L) - first[L], one small table read (Tab::base_get).
template <class Tab>
SSV_HD void huff_limits(const HuffCounts &h, Tab &tab, int set, HuffCounts &lim)
{
	int first = 0, off = 0;
#pragma unroll
	for (int k = 0; k < 8; ++k) lim.c[k] = 0;
#pragma unroll
	for (int l = 1; l <= 15; ++l) {
		const int count = (int)((h.c[l >> 1] >> ((l & 1) * 16)) & 0xffffu);
		const int end = first + count; // one past the last code of this length (<= 2^l: huff_construct has refused over-subscribed codes)
		lim.c[l >> 1] |= ((uint32_t)end << (15 - l)) << ((l & 1) * 16);
		tab.base_set(set, l, off - first);
		off += count; first = end << 1;
	}
}
template <bool LIT, class Tab, class Reader>
SSV_HD int huff_decode_lim(Reader &br, const HuffCounts &lim, const Tab &tab, int set, int sym_base)
{
	const uint32_t rev = brev32((uint32_t)br.bb) >> 17; // caller guarantees >= 15 valid bits
	const uint32_t x = (rev * 0x00010001u) | 0x80008000u;
	uint32_t m = 0;
#pragma unroll
	for (int k = 0; k < 8; ++k) m |= ((x - lim.c[k]) >> k) & (0x80008000u >> k);
	const int used = popc32(m);
	if (used > 15) return -1; // no code of any length: the code is incomplete and these bits are not in it
	br.drop(used);
	const int k = sym_base + (int)(rev >> (15 - used)) + tab.base_get(set, used);
	return LIT ? tab.lit_get(k) : tab.dst_get(k);
}
// one symbol, by whichever decode the table supports (h holds limits when Tab::has_base, counts otherwise)
template <bool LIT, class Tab, class Reader>
SSV_HD int huff_next(Reader &br, const HuffCounts &h, const Tab &tab, int set, int sym_base)
{
	if constexpr (Tab::has_base) return huff_decode_lim<LIT>(br, h, tab, set, sym_base);
	else return huff_decode<LIT>(br, h, tab, sym_base);
}
template <class Tab>
SSV_HD void huff_ready(HuffCounts &h, Tab &tab, int set) // counts -> what huff_next wants
{
	if constexpr (Tab::has_base) { const HuffCounts counts = h; huff_limits(counts, tab, set, h); }
}

// build the decoding tables of one code from the lengths at Tab::len[len_base, len_base + n): counts -> h, permutation -> Tab::sym[sym_base...)
// `rule`: what an INCOMPLETE code (bit patterns that belong to no symbol) means - zlib's inftrees.c: HUFF_ANY accepts it (the fixed distance code is
// one), HUFF_DATA (literal/length and distance codes of a dynamic block) only a code of a single 1-bit symbol or of no symbol at all, HUFF_CODES (the
// code-length code) only one of no symbol at all.  A damaged header rarely gives complete codes: refusing here is most of what stands between a
// flipped bit and wrong output, since - like libbam 0.1.16's reader - nothing checks a block's CRC32.
enum : int { HUFF_ANY = 0, HUFF_DATA = 1, HUFF_CODES = 2 };
template <bool LIT, class Tab>
SSV_HD int huff_construct(Tab &tab, int len_base, int n, int sym_base, HuffCounts &h, int rule = HUFF_ANY)
{
	for (int l = 0; l < 16; ++l) tab.off_set(l, 0);
	for (int s = 0; s < n; ++s) { const int l = tab.len_get(len_base + s); tab.off_set(l, (uint16_t)(tab.off_get(l) + 1)); }
	int left = 1, run = 0;
#pragma unroll
	for (int k = 0; k < 8; ++k) h.c[k] = 0;
#pragma unroll
	for (int l = 1; l <= 15; ++l) {
		const int count = tab.off_get(l);
		h.c[l >> 1] |= (uint32_t)count << ((l & 1) * 16);
		left <<= 1; left -= count;
		if (left < 0) return INF_E_OVERSUB;
		tab.off_set(l, (uint16_t)run); // offset of the first symbol of this length in the permutation
		run += count;
	}
	if (left > 0 && run > 0 && (rule == HUFF_CODES || (rule == HUFF_DATA && !(run == 1 && (h.c[0] >> 16) == 1u)))) return INF_E_CODE;
	for (int s = 0; s < n; ++s) {
		const int l = tab.len_get(len_base + s);
		if (l) {
			const int o = tab.off_get(l);
			if (LIT) tab.lit_set(sym_base + o, (uint16_t)s); else tab.dst_set(sym_base + o, (uint16_t)s);
			tab.off_set(l, (uint16_t)(o + 1));
		}
	}
	return INF_OK;
}

// Inflate one raw deflate stream of in_len bytes into exactly out_len bytes.  Returns INF_OK or an error (the lane's block is then bad).
template <class Tab, class Out, class Reader>
SSV_HD int inflate_stream_from(Reader &br, const uint8_t *in, uint32_t in_len, Out &out, uint32_t out_len, Tab &tab)
{
	uint32_t o = 0;
	int last;
	do {
		br.refill();
		last = (int)br.take(1);
		const int type = (int)br.take(2);
		if (type == 0) {
			// stored: to the byte boundary, LEN, ~LEN, bytes
			br.drop(br.bc & 7);
			br.refill();
			const uint32_t len = br.take(16);
			br.refill();
			const uint32_t nlen = br.take(16);
			if ((len ^ 0xffffu) != nlen) return INF_E_STORED;
			if (o + len > out_len) return INF_E_OUTPUT;
			for (uint32_t i = 0; i < len; ++i) { br.refill(); out.put(o, (uint8_t)br.take(8)); }
			continue;
		}
		if (type == 3) return INF_E_BTYPE;
		HuffCounts lit, dist;
		if (type == 1) {
			for (int s = 0; s < 288; ++s) tab.len_set(s, s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
			for (int s = 0; s < 30; ++s) tab.len_set(288 + s, 5);
			huff_construct<true>(tab, 0, 288, 0, lit);
			huff_construct<false>(tab, 288, 30, 0, dist);
			huff_ready(lit, tab, 0); huff_ready(dist, tab, 1);
		} else {
			const int nlen = (int)br.take(5) + 257, ndist = (int)br.take(5) + 1, ncode = (int)br.take(4) + 4;
			if (nlen > 286 || ndist > 30) return INF_E_CODE;
			for (int s = 0; s < 19; ++s) tab.len_set(s, 0);
			for (int k = 0; k < ncode; ++k) {
				br.refill();
				// order of the code-length code lengths: 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
				const int s = k < 3 ? 16 + k : k == 3 ? 0 : (k & 1) ? 8 - ((k - 3) >> 1) : 8 + ((k - 4) >> 1);
				tab.len_set(s, (int)br.take(3));
			}
			HuffCounts cl;
			int rc = huff_construct<false>(tab, 0, 19, 12, cl, HUFF_CODES);
			if (rc != INF_OK) return rc;
			huff_ready(cl, tab, 1); // (the distance code's slots: it is built after the lengths have been read)
			int idx = 0, prev = 0;
			while (idx < nlen + ndist) {
				br.refill();
				int s = huff_next<false>(br, cl, tab, 1, 12);
				if (s < 0) return INF_E_CODE;
				if (s < 16) { tab.len_set(idx++, s); prev = s; continue; }
				int rep, val = 0;
				if (s == 16) { if (idx == 0) return INF_E_REPEAT; val = prev; rep = 3 + (int)br.take(2); }
				else if (s == 17) rep = 3 + (int)br.take(3);
				else rep = 11 + (int)br.take(7);
				if (idx + rep > nlen + ndist) return INF_E_REPEAT;
				while (rep--) tab.len_set(idx++, val);
				prev = val;
			}
			if (tab.len_get(256) == 0) return INF_E_CODE; // no end-of-block code
			// the distance lengths sit right behind the literal/length lengths: build the distance code first (its lengths are read in
			// place), then the literal/length code
			rc = huff_construct<false>(tab, nlen, ndist, 0, dist, HUFF_DATA);
			if (rc != INF_OK) return rc;
			rc = huff_construct<true>(tab, 0, nlen, 0, lit, HUFF_DATA);
			if (rc != INF_OK) return rc;
			huff_ready(lit, tab, 0); huff_ready(dist, tab, 1);
		}
		for (;;) {
			br.top_of_symbol();
			br.refill();
			int s = huff_next<true>(br, lit, tab, 0, 0);
			if (s < 0) return INF_E_CODE;
			if (s < 256) {
				if (o >= out_len) return INF_E_OUTPUT;
				out.put(o, (uint8_t)s);
				continue;
			}
			if (s == 256) break;
			if (s > 285) return INF_E_CODE;
			// length: 257..264 -> 3..10; then groups of four symbols share an extra-bit count; 285 -> 258
			uint32_t len;
			if (s < 265) len = (uint32_t)s - 254u;
			else if (s == 285) len = 258;
			else { const int e = (s - 261) >> 2; len = ((4u + (uint32_t)((s - 265) & 3)) << e) + 3u + br.take(e); }
			br.refill();
			const int d = huff_next<false>(br, dist, tab, 1, 0);
			if (d < 0 || d > 29) return INF_E_CODE;
			uint32_t dst;
			if (d < 4) dst = (uint32_t)d + 1u;
			else { const int e = (d >> 1) - 1; dst = ((2u + (uint32_t)(d & 1)) << e) + 1u + br.take(e); }
			if (dst > o) return INF_E_DIST;
			if (o + len > out_len) return INF_E_OUTPUT;
			out.copy(o, dst, len);
		}
	} while (!last);
	out.finish(o);
	if (o != out_len) return INF_E_OUTPUT;
	// bytes taken from the input: everything up to p except the whole bytes still in the buffer
	if (br.consumed(in) > in_len) return INF_E_INPUT;
	return INF_OK;
}

template <class Tab, class Out>
SSV_HD int inflate_stream_to(const uint8_t *in, uint32_t in_len, Out &out, uint32_t out_len, Tab &tab)
{
	BitReader br(in, in_len);
	return inflate_stream_from(br, in, in_len, out, out_len, tab);
}

template <class Tab>
SSV_HD int inflate_stream(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_len, Tab &tab)
{
	DirectOut d{out};
	return inflate_stream_to(in, in_len, d, out_len, tab);
}

} // namespace ssv
