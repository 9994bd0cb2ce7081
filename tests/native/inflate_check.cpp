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
	for (int shape = 0; shape < 7; ++shape)
		for (size_t n : sizes)
			for (int level = 0; level <= 9; ++level)
				for (int strategy : {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE}) {
					if (strategy != Z_DEFAULT_STRATEGY && level != 6) continue;
					std::vector<uint8_t> d = make_data(shape, n);
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
					if (same) ++n_ok;
					else { ++n_bad; fprintf(stderr, "MISMATCH shape %d n %zu level %d strategy %d rc %d\n", shape, n, level, strategy, rc); }
					// a stream cut short or an output size that is wrong must be refused, not overrun
					if (n > 16) {
						int r2 = ssv::inflate_stream(c.data(), clen, o.data(), (uint32_t)n - 1, tab);
						if (r2 == ssv::INF_OK) { ++n_bad; fprintf(stderr, "short output accepted: shape %d n %zu level %d\n", shape, n, level); }
					}
				}
	printf("%d streams ok, %d bad\n", n_ok, n_bad);
	return n_bad ? 1 : 0;
}
