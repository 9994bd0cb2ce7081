// CPU check of seeksv_amd/csrc/inflate_core.h (the per-lane DEFLATE decoder of the device BGZF reader) against zlib:
// raw deflate streams of several data shapes, every compression level, the fixed-Huffman and stored strategies, sizes up to 64 KB.
// Built and run by tests/test_inflate_core.py.  Exit code 0 = all streams decoded to the original bytes.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <zlib.h>

#include "inflate_core.h"

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 11); }

static std::vector<uint8_t> make_data(int shape, size_t n)
{
	std::vector<uint8_t> d(n);
	switch (shape) {
	case 0: for (auto &b : d) b = (uint8_t)rnd(); break;                                 // incompressible
	case 1: for (auto &b : d) b = "ACGT"[rnd() & 3]; break;                             // 2 bits of entropy per byte
	case 2: for (size_t i = 0; i < n; ++i) d[i] = (uint8_t)(i % 7 == 0 ? rnd() : 'x'); break; // long runs (distance 1 copies)
	case 3: for (size_t i = 0; i < n; ++i) d[i] = (uint8_t)((i * 2654435761u) >> 24); break;
	case 4: { // BAM-like: repeating record skeletons with small mutations, far back-references
		std::vector<uint8_t> rec(300);
		for (auto &b : rec) b = (uint8_t)rnd();
		for (size_t i = 0; i < n; ++i) { if (i % 300 == 0) for (int k = 0; k < 20; ++k) rec[rnd() % 300] = (uint8_t)rnd(); d[i] = rec[i % 300]; }
		break;
	}
	case 5: for (auto &b : d) b = 0; break;                                              // one symbol
	default: for (size_t i = 0; i < n; ++i) d[i] = (uint8_t)(rnd() % (1 + i % 251)); break;
	}
	return d;
}

int main()
{
	const size_t sizes[] = {0, 1, 2, 3, 17, 255, 256, 257, 4096, 30000, 65280, 65536};
	int n_ok = 0, n_bad = 0;
	ssv::PlainTab tab;
	ssv::PlainTabLim tabl; // the table of the two-pass kernel: symbols decoded by limit compares (huff_decode_lim)
	for (int shape = 0; shape < 7; ++shape)
		for (size_t n : sizes)
			for (int level = 0; level <= 9; ++level)
				for (int strategy : {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE}) {
					if (strategy != Z_DEFAULT_STRATEGY && level != 6) continue;
					std::vector<uint8_t> d = make_data(shape, n);
					d.reserve(n + 1); // (data() of an empty vector may be null: memcmp does not take that, even for zero bytes)
					z_stream zs = {};
					if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy) != Z_OK) return 2;
					std::vector<uint8_t> c(2 * n + 1024, 0xAA); // Z_FIXED on noisy data exceeds deflateBound
					zs.next_in = d.data(); zs.avail_in = (uInt)n; zs.next_out = c.data(); zs.avail_out = (uInt)c.size() - 8;
					{ int drc = deflate(&zs, Z_FINISH); if (drc != Z_STREAM_END) { fprintf(stderr, "deflate rc %d shape %d n %zu level %d strat %d\n", drc, shape, n, level, strategy); return 3; } }
					const uint32_t clen = (uint32_t)zs.total_out;
					deflateEnd(&zs);
					std::vector<uint8_t> o(n + 8, 0x55);
					int rc = ssv::inflate_stream(c.data(), clen, o.data(), (uint32_t)n, tab);
					bool same = rc == ssv::INF_OK && memcmp(o.data(), d.data(), n) == 0 && o[n] == 0x55;
					for (int mo = 0; mo < 4; ++mo) { // decode and copy split in two: tokens, then the holes filled in order; the output at every misalignment
						// (literals leave as aligned dwords: nothing may be written in front of the block or behind it)
						std::vector<uint8_t> o4(n + 24, 0x55);
						uint8_t *op = o4.data() + 8 + mo;
						std::vector<uint32_t> tk(ssv::token_capacity((uint32_t)n) + 1, 0xdeadbeefu);
						ssv::TokenOut to;
						to.out = op; to.tok = tk.data();
						int rc4 = ssv::inflate_stream_to(c.data(), clen, to, (uint32_t)n, tab);
						if (rc4 == ssv::INF_OK) ssv::resolve_tokens(op, tk.data(), to.n);
						bool guard = true;
						for (int g = 0; g < 8 + mo; ++g) guard = guard && o4[g] == 0x55;
						for (size_t g = 8 + mo + n; g < o4.size(); ++g) guard = guard && o4[g] == 0x55;
						if (!(rc4 == ssv::INF_OK && to.n <= ssv::token_capacity((uint32_t)n) && tk[ssv::token_capacity((uint32_t)n)] == 0xdeadbeefu && memcmp(op, d.data(), n) == 0 && guard)) { same = false; fprintf(stderr, "TOKEN sink (misalignment %d): ", mo); }
					}
					for (int mi = 0; mi < 4; ++mi) { // the window reader (RingReader) in front of the token sink, at every misalignment of the input
						struct CpuRing {
							uint32_t a[32];
							uint32_t get(uint32_t j) const { return a[j]; }
							void set(uint32_t j, uint32_t v) { a[j] = v; }
							bool any(bool c) const { return c; }
						};
						std::vector<uint8_t> cc(clen + 4 + 32, 0xAA); // (a group may reach 16 + 3 bytes past the stream's end)
						memcpy(cc.data() + mi, c.data(), clen);
						std::vector<uint8_t> o5(n + 8, 0x55);
						std::vector<uint32_t> tk(ssv::token_capacity((uint32_t)n) + 1, 0xdeadbeefu);
						ssv::TokenOut to;
						to.out = o5.data(); to.tok = tk.data();
						int rc5;
						if (mi & 1) { ssv::RingReader<CpuRing, 64> br(cc.data() + mi, clen, CpuRing()); rc5 = ssv::inflate_stream_from(br, cc.data() + mi, clen, to, (uint32_t)n, tab); }
						else { ssv::RingReader<CpuRing, 128> br(cc.data() + mi, clen, CpuRing()); rc5 = ssv::inflate_stream_from(br, cc.data() + mi, clen, to, (uint32_t)n, tabl); } // (as the kernel does)
						if (rc5 == ssv::INF_OK) ssv::resolve_tokens(o5.data(), tk.data(), to.n);
						if (!(rc5 == ssv::INF_OK && memcmp(o5.data(), d.data(), n) == 0 && o5[n] == 0x55)) { same = false; fprintf(stderr, "WINDOW reader (misalignment %d, rc %d): ", mi, rc5); }
						if (n > 16 && level > 0 && shape != 5 && mi == 1) { // a stream cut short is refused (fed zero bits behind its end, not the next stream's bytes)
							ssv::TokenOut t2;
							t2.out = o5.data(); t2.tok = tk.data();
							ssv::RingReader<CpuRing, 128> br(cc.data() + mi, clen - 3, CpuRing());
							if (ssv::inflate_stream_from(br, cc.data() + mi, clen - 3, t2, (uint32_t)n, tabl) == ssv::INF_OK) { same = false; fprintf(stderr, "WINDOW reader: truncated input accepted: "); }
						}
					}
					if (same) ++n_ok;
					else { ++n_bad; fprintf(stderr, "MISMATCH shape %d n %zu level %d strategy %d rc %d\n", shape, n, level, strategy, rc); }
					// a stream cut short or an output size that is wrong must be refused, not overrun
					if (n > 16) {
						int r2 = ssv::inflate_stream(c.data(), clen, o.data(), (uint32_t)n - 1, tab);
						if (r2 == ssv::INF_OK) { ++n_bad; fprintf(stderr, "short output accepted: shape %d n %zu level %d\n", shape, n, level); }
					}
				}
	// Damaged streams: whatever the decoder ACCEPTS, zlib accepts too, with the same bytes (the structural rules are zlib's: over-subscribed and
	// incomplete codes, a missing end-of-block code, distances in front of the block, symbols that do not exist, input or output that ends early)
	{
		int accepted = 0, refused = 0;
		for (int it = 0; it < 6000; ++it) {
			const int shape = it % 7;
			const size_t n = 200 + rnd() % 3000;
			std::vector<uint8_t> d = make_data(shape, n);
			z_stream zs = {};
			if (deflateInit2(&zs, 1 + it % 9, Z_DEFLATED, -15, 8, it % 11 == 0 ? Z_FIXED : Z_DEFAULT_STRATEGY) != Z_OK) return 2;
			std::vector<uint8_t> c(2 * n + 1024, 0);
			zs.next_in = d.data(); zs.avail_in = (uInt)n; zs.next_out = c.data(); zs.avail_out = (uInt)c.size() - 8;
			if (deflate(&zs, Z_FINISH) != Z_STREAM_END) return 3;
			const uint32_t clen = (uint32_t)zs.total_out;
			deflateEnd(&zs);
			// damage: mostly inside the first bytes (the block header with its code lengths), sometimes anywhere
			const int flips = 1 + (int)(rnd() % 3);
			for (int f = 0; f < flips; ++f) { const uint32_t at = (rnd() & 3) ? rnd() % (clen < 40 ? clen : 40) : rnd() % clen; c[at] ^= (uint8_t)(1u << (rnd() & 7)); }
			std::vector<uint8_t> o(n + 8, 0x55), zo(n + 8, 0);
			const int rc = ssv::inflate_stream(c.data(), clen, o.data(), (uint32_t)n, tab);
			std::vector<uint32_t> tk(ssv::token_capacity((uint32_t)n) + 1);
			std::vector<uint8_t> o2(n + 8, 0x55);
			ssv::TokenOut to;
			to.out = o2.data(); to.tok = tk.data();
			const int rc2 = ssv::inflate_stream_to(c.data(), clen, to, (uint32_t)n, tab);
			if ((rc == ssv::INF_OK) != (rc2 == ssv::INF_OK)) { ++n_bad; fprintf(stderr, "damaged stream %d: the two sinks disagree (%d, %d)\n", it, rc, rc2); }
			if (rc != ssv::INF_OK) { ++refused; continue; }
			++accepted;
			z_stream is = {};
			if (inflateInit2(&is, -15) != Z_OK) return 2;
			is.next_in = c.data(); is.avail_in = clen; is.next_out = zo.data(); is.avail_out = (uInt)n + 8;
			const int zr = inflate(&is, Z_FINISH);
			const bool zok = zr == Z_STREAM_END && is.total_out == n && memcmp(zo.data(), o.data(), n) == 0;
			inflateEnd(&is);
			if (!zok) { ++n_bad; fprintf(stderr, "damaged stream %d (shape %d, n %zu): accepted, but zlib says %d with %lu bytes\n", it, shape, n, zr, (unsigned long)is.total_out); }
		}
		printf("damaged streams: %d refused, %d accepted (each of those as zlib decodes it)\n", refused, accepted);
		if (refused < 1000) { ++n_bad; fprintf(stderr, "too few damaged streams refused\n"); }
	}
	printf("%d streams ok, %d bad\n", n_ok, n_bad);
	return n_bad ? 1 : 0;
}
