// huff_gz.h - a gzip member whose deflate stream is ONE dynamic-Huffman block of literals (RFC 1951 / 1952): no string matching at all.
//
// Why: the rows of `seeksv getclip` (prefix.clip.gz, prefix.clip.fq.gz: clipped bases, qualities, positions) are text whose redundancy is its small
// alphabet, not repeats: zlib level 1 packs them 2.7 x at 50-60 MB/s per core, and at whole-genome size (2.5 GB of rows) that was the largest consumer
// of host CPU time of the command.  An order-0 Huffman code packs the same rows 2.2-2.3 x at several hundred MB/s per core - a histogram pass and a
// table-driven bit-append pass.  The decompressed bytes are what the reference's readers (igzstream, bwa, zcat) see; they are identical.
// SSV_GZ_LEVEL=1..9 selects zlib at that level instead (6 = the reference's file sizes).
#pragma once

#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include <zlib.h> // crc32

namespace ssvh_huff {

// Huffman code lengths (<= maxlen) for n symbols with the given counts (0 = unused).  When the optimal code is deeper than maxlen the counts are
// halved (rounding up, so a used symbol stays used) and the code rebuilt: a few rounds at most, and only for extremely skewed inputs.
inline void code_lengths(const uint32_t *count, int n, int maxlen, uint8_t *len)
{
	std::vector<uint32_t> cnt(count, count + n);
	for (;;) {
		struct Node { uint64_t w; int left, right; };
		std::vector<Node> nodes;
		std::vector<int> leaves;
		for (int s = 0; s < n; ++s) { len[s] = 0; if (cnt[s]) { nodes.push_back(Node{cnt[s], -1, s}); leaves.push_back((int)nodes.size() - 1); } }
		const int used = (int)leaves.size();
		if (used == 0) return;
		if (used == 1) { len[nodes[0].right] = 1; return; }
		std::sort(leaves.begin(), leaves.end(), [&](int a, int b) { return nodes[a].w != nodes[b].w ? nodes[a].w < nodes[b].w : nodes[a].right < nodes[b].right; });
		// two queues: sorted leaves and the internal nodes in the order they are made (their weights never decrease)
		std::vector<int> inner;
		size_t li = 0, ii = 0;
		auto take = [&]() -> int {
			if (li < leaves.size() && (ii >= inner.size() || nodes[leaves[li]].w <= nodes[inner[ii]].w)) return leaves[li++];
			return inner[ii++];
		};
		for (int k = 0; k < used - 1; ++k) {
			const int a = take(), b = take();
			nodes.push_back(Node{nodes[a].w + nodes[b].w, a, b});
			inner.push_back((int)nodes.size() - 1);
		}
		// depths from the root down (children are made before their parent: walk the array backwards)
		std::vector<int> depth(nodes.size(), 0);
		int deepest = 0;
		for (int i = (int)nodes.size() - 1; i >= 0; --i) {
			if (nodes[i].left < 0) { len[nodes[i].right] = (uint8_t)depth[i]; deepest = std::max(deepest, depth[i]); }
			else { depth[nodes[i].left] = depth[i] + 1; depth[nodes[i].right] = depth[i] + 1; }
		}
		if (deepest <= maxlen) return;
		for (auto &c : cnt) if (c) c = (c + 1) / 2;
	}
}

// canonical codes of the lengths, bit-reversed (deflate sends Huffman codes most significant bit first into an LSB-first stream)
inline void canonical_codes(const uint8_t *len, int n, uint16_t *code)
{
	uint32_t bl_count[16] = {0}, next[16] = {0};
	for (int s = 0; s < n; ++s) bl_count[len[s]]++;
	bl_count[0] = 0;
	uint32_t c = 0;
	for (int b = 1; b <= 15; ++b) { c = (c + bl_count[b - 1]) << 1; next[b] = c; }
	for (int s = 0; s < n; ++s) {
		if (!len[s]) { code[s] = 0; continue; }
		uint32_t v = next[len[s]]++, r = 0;
		for (int k = 0; k < len[s]; ++k) r |= ((v >> k) & 1u) << (len[s] - 1 - k);
		code[s] = (uint16_t)r;
	}
}

struct BitSink {
	uint8_t *o;
	uint64_t bb = 0;
	int bc = 0;
	explicit BitSink(uint8_t *out) : o(out) {}
	inline void put(uint32_t v, int n) { bb |= (uint64_t)v << bc; bc += n; if (bc >= 32) { const uint32_t w = (uint32_t)bb; memcpy(o, &w, 4); o += 4; bb >>= 32; bc -= 32; } }
	inline void finish_bytes() { while (bc >= 8) { *o++ = (uint8_t)bb; bb >>= 8; bc -= 8; } } // afterwards fewer than 8 bits are pending
	inline uint8_t *finish() { while (bc > 0) { *o++ = (uint8_t)bb; bb >>= 8; bc -= 8; } bc = 0; return o; }
};

// upper bound of member(): header + block header + 15 bits a byte + trailer
inline size_t member_bound(size_t n) { return n * 2 + 1024; }

// the raw deflate stream for text[0, n) - one final dynamic-Huffman block of literals - into o (room for member_bound(n)); returns its end
inline uint8_t *deflate_literals(const uint8_t *text, size_t n, uint8_t *o)
{
	if (n == 0) { *o++ = 0x03; *o++ = 0x00; } // a final fixed-Huffman block holding only the end-of-block symbol
	else {
		// literal / length alphabet: the 256 byte values + end of block (257 codes, HLIT = 0); four interleaved histograms so that runs of one
		// character do not serialise on one counter
		uint32_t h[4][256];
		memset(h, 0, sizeof(h));
		size_t i = 0;
		for (; i + 4 <= n; i += 4) { h[0][text[i]]++; h[1][text[i + 1]]++; h[2][text[i + 2]]++; h[3][text[i + 3]]++; }
		for (; i < n; ++i) h[0][text[i]]++;
		uint32_t cnt[257];
		for (int s = 0; s < 256; ++s) cnt[s] = h[0][s] + h[1][s] + h[2][s] + h[3][s];
		cnt[256] = 1;
		uint8_t len[258];
		uint16_t code[257];
		code_lengths(cnt, 257, 15, len);
		canonical_codes(len, 257, code);
		len[257] = 0; // the one distance code, of zero bits: "the data is all literals" (RFC 1951, 3.2.7)
		// the code-length code over the values 0..15 that occur (no repeat codes: 258 lengths are ~100 bytes of a 256 KiB member)
		uint32_t ccnt[19] = {0};
		for (int s = 0; s < 258; ++s) ccnt[len[s]]++;
		uint8_t clen[19];
		uint16_t ccode[19];
		code_lengths(ccnt, 19, 7, clen);
		canonical_codes(clen, 19, ccode);
		static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
		int hclen = 19;
		while (hclen > 4 && clen[order[hclen - 1]] == 0) --hclen;
		BitSink bs(o);
		bs.put(1, 1); bs.put(2, 2);   // BFINAL, BTYPE = dynamic
		bs.put(0, 5); bs.put(0, 5);   // HLIT = 257 - 257, HDIST = 1 - 1
		bs.put((uint32_t)(hclen - 4), 4);
		for (int k = 0; k < hclen; ++k) bs.put(clen[order[k]], 3);
		for (int s = 0; s < 258; ++s) bs.put(ccode[len[s]], clen[len[s]]);
		// the literals: code and length of a byte in one table entry
		uint32_t tab[256];
		for (int s = 0; s < 256; ++s) tab[s] = (uint32_t)code[s] | ((uint32_t)len[s] << 16);
		// three codes (<= 45 bits) on top of at most 7 pending ones, then a branch-free flush of the whole bytes (the store reaches up to 8 bytes
		// past the stream's end: member_bound leaves room)
		bs.finish_bytes();
		uint64_t bb = bs.bb;
		int bc = bs.bc;
		uint8_t *p = bs.o;
		size_t k = 0;
		for (; k + 3 <= n; k += 3) {
			const uint32_t e0 = tab[text[k]], e1 = tab[text[k + 1]], e2 = tab[text[k + 2]];
			bb |= (uint64_t)(e0 & 0xffffu) << bc; bc += (int)(e0 >> 16);
			bb |= (uint64_t)(e1 & 0xffffu) << bc; bc += (int)(e1 >> 16);
			bb |= (uint64_t)(e2 & 0xffffu) << bc; bc += (int)(e2 >> 16);
			memcpy(p, &bb, 8);
			p += bc >> 3; bb >>= bc & ~7; bc &= 7;
		}
		bs.bb = bb; bs.bc = bc; bs.o = p;
		for (; k < n; ++k) bs.put(code[text[k]], len[text[k]]);
		bs.put(code[256], len[256]);
		o = bs.finish();
	}
	return o;
}

// The same block WITH string matching of the cheapest kind (bench.py's whole-genome file on a 16-CPU box: zlib level 4 deflates 40 MB/s per core, this
// several hundred): a hash of the four bytes at every position, ONE candidate per hash (the last position that hashed there), greedy, matches of 4..258
// bytes, no hash insertions inside a match; then the two Huffman codes of the tokens that came out.  Any inflater reads it (tests/test_gz_writer.py: zlib;
// the device decoder).  n < 65536 (a BGZF block).
inline uint8_t *deflate_fast(const uint8_t *text, size_t n, uint8_t *o)
{
	if (n < 32 || n >= 65536) return deflate_literals(text, n, o);
	constexpr int HB = 13;
	struct Tok { uint16_t a, d; };                // d == 0: the literal a; else a match of a bytes, d back
	struct Work { uint16_t head[1 << HB]; Tok toks[65536]; }; // head: position + 1 of the last occurrence of the hash (0: none)
	static thread_local Work *tls_work = nullptr;
	if (!tls_work) tls_work = new Work;           // (one per thread, kept; looked up ONCE per call: in a shared library every touch of a thread_local is a call)
	Work &W = *tls_work;
	uint16_t *const head = W.head;
	memset(head, 0, sizeof(W.head));
	Tok *const toks = W.toks;
	size_t nt = 0;
	uint32_t lcnt[286] = {0}, dcnt[30] = {0};
	auto len_sym = [](uint32_t l, uint32_t &xb, uint32_t &xv) -> uint32_t {
		const uint32_t y = l - 3;
		if (y < 8) { xb = 0; xv = 0; return 257 + y; }
		if (l == 258) { xb = 0; xv = 0; return 285; }
		const uint32_t hb = 31u - (uint32_t)__builtin_clz(y);
		xb = hb - 2; xv = y & ((1u << xb) - 1u);
		return 257 + 4 * (hb - 1) + ((y >> (hb - 2)) & 3u);
	};
	auto dist_sym = [](uint32_t d, uint32_t &xb, uint32_t &xv) -> uint32_t {
		const uint32_t x = d - 1;
		if (x < 4) { xb = 0; xv = 0; return x; }
		const uint32_t hb = 31u - (uint32_t)__builtin_clz(x);
		xb = hb - 1; xv = x & ((1u << xb) - 1u);
		return 2 * hb + ((x >> (hb - 1)) & 1u);
	};
	size_t i = 0;
	while (i + 4 <= n) {
		uint32_t v;
		memcpy(&v, text + i, 4);
		const uint32_t h = (v * 2654435761u) >> (32 - HB);
		const size_t cand = head[h];
		head[h] = (uint16_t)(i + 1);
		if (cand) {
			const size_t c = cand - 1;
			uint32_t w;
			memcpy(&w, text + c, 4);
			if (w == v && i - c <= 32768) { // (deflate's window; a BGZF block is up to 65280 bytes)
				const size_t maxl = std::min<size_t>(258, n - i);
				size_t len = 4;
				while (len + 8 <= maxl) {
					uint64_t x, y;
					memcpy(&x, text + c + len, 8); memcpy(&y, text + i + len, 8);
					if (x != y) { len += (size_t)(__builtin_ctzll(x ^ y) >> 3); goto matched; }
					len += 8;
				}
				while (len < maxl && text[c + len] == text[i + len]) ++len;
			matched:
				uint32_t xb, xv;
				toks[nt++] = Tok{(uint16_t)len, (uint16_t)(i - c)};
				lcnt[len_sym((uint32_t)len, xb, xv)]++; dcnt[dist_sym((uint32_t)(i - c), xb, xv)]++;
				i += len;
				continue;
			}
		}
		toks[nt++] = Tok{text[i], 0};
		lcnt[text[i]]++;
		++i;
	}
	for (; i < n; ++i) { toks[nt++] = Tok{text[i], 0}; lcnt[text[i]]++; }
	lcnt[256] = 1;
	uint8_t llen[286], dlen[30];
	uint16_t lcode[286], dcode[30];
	code_lengths(lcnt, 286, 15, llen);
	canonical_codes(llen, 286, lcode);
	code_lengths(dcnt, 30, 15, dlen);
	canonical_codes(dlen, 30, dcode);
	int hlit = 286, hdist = 30;
	while (hlit > 257 && llen[hlit - 1] == 0) --hlit;
	while (hdist > 1 && dlen[hdist - 1] == 0) --hdist; // (no match at all: one distance code of zero bits - "the data is all literals", RFC 1951 3.2.7)
	uint32_t ccnt[19] = {0};
	for (int s = 0; s < hlit; ++s) ccnt[llen[s]]++;
	for (int s = 0; s < hdist; ++s) ccnt[dlen[s]]++;
	uint8_t clen[19];
	uint16_t ccode[19];
	code_lengths(ccnt, 19, 7, clen);
	canonical_codes(clen, 19, ccode);
	static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
	int hclen = 19;
	while (hclen > 4 && clen[order[hclen - 1]] == 0) --hclen;
	BitSink bs(o);
	bs.put(1, 1); bs.put(2, 2); // BFINAL, BTYPE = dynamic
	bs.put((uint32_t)(hlit - 257), 5); bs.put((uint32_t)(hdist - 1), 5); bs.put((uint32_t)(hclen - 4), 4);
	for (int k = 0; k < hclen; ++k) bs.put(clen[order[k]], 3);
	for (int s = 0; s < hlit; ++s) bs.put(ccode[llen[s]], clen[llen[s]]);
	for (int s = 0; s < hdist; ++s) bs.put(ccode[dlen[s]], clen[dlen[s]]);
	for (size_t k = 0; k < nt; ++k) {
		const Tok t = toks[k];
		if (t.d == 0) { bs.put(lcode[t.a], llen[t.a]); continue; }
		uint32_t xb, xv;
		const uint32_t ls = len_sym(t.a, xb, xv);
		bs.put(lcode[ls], llen[ls]);
		if (xb) bs.put(xv, (int)xb);
		const uint32_t ds = dist_sym(t.d, xb, xv);
		bs.put(dcode[ds], dlen[ds]);
		if (xb) bs.put(xv, (int)xb);
	}
	bs.put(lcode[256], llen[256]);
	return bs.finish();
}

// one gzip member for text[0, n) into out (room for member_bound(n)); returns its size
inline size_t member(const uint8_t *text, size_t n, uint8_t *out)
{
	static const uint8_t gz_header[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3}; // deflate, no flags, no time, OS = unix
	memcpy(out, gz_header, 10);
	uint8_t *o = deflate_literals(text, n, out + 10);
	const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), text, (uInt)n), isize = (uint32_t)n;
	memcpy(o, &crc, 4); memcpy(o + 4, &isize, 4);
	return (size_t)(o + 8 - out);
}

} // namespace ssvh_huff
