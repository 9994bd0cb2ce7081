// tools/inflate_match_stats.cpp - literal / match statistics of a BAM's BGZF blocks (first 400 blocks), decoded with the repository's own
// ring-machine decoder on the CPU.  g++ -O2 -std=c++17 -Iseeksv_amd/csrc tools/inflate_match_stats.cpp -o /tmp/inflate_match_stats && /tmp/inflate_match_stats file.bam
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string.h>
#include "inflate_core.h"
#include "inflate_lanes.h"
using Cfg = ssv::RingCfg<32, 32, 16>;
template <class C> struct CpuIo {
	uint32_t in_ring[C::IN_DW]; uint8_t out_ring[4 * C::OUT_DW];
	const uint8_t *in_org, *in_end; uint8_t *out_org;
	uint32_t in_get(int slot) const { return in_ring[slot]; }
	uint32_t in_stream32(uint32_t off) const { uint32_t v = 0; for (int b = 0; b < 4; ++b) if (in_org + off + b < in_end) v |= (uint32_t)in_org[off + b] << (8 * b); return v; }
	void out_set8(uint32_t idx, uint8_t v) { out_ring[idx] = v; }
	uint8_t out_get8(uint32_t idx) const { return out_ring[idx]; }
	uint32_t out_get32(int slot) const { uint32_t v; memcpy(&v, out_ring + 4 * slot, 4); return v; }
	void out_set32(int slot, uint32_t v) { memcpy(out_ring + 4 * slot, &v, 4); }
	uint32_t out_stream32(uint32_t pos) const { uint32_t v; memcpy(&v, out_org + pos, 4); return v; }
};
int main(int argc, char **argv)
{
	FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
	std::vector<uint8_t> buf(sz + 64); if (fread(buf.data(), 1, sz, f) != (size_t)sz) return 1;
	long off = 0; int nblk = 0;
	long lit = 0, nmatch = 0, dist_hist[20] = {0}, len_hist[10] = {0}, bytes_lit = 0, bytes_match = 0, hdrs = 0;
	long dist_bytes[20] = {0};
	ssv::PlainTab tab;
	while (off + 18 < sz && nblk < 400) {
		const uint8_t *h = buf.data() + off;
		unsigned xlen = h[10] | h[11] << 8; unsigned bsize = (h[16] | h[17] << 8) + 1;
		const uint8_t *in = h + 12 + xlen; unsigned clen = bsize - xlen - 19; unsigned isize; memcpy(&isize, h + bsize - 4, 4);
		std::vector<uint8_t> out(isize + 16);
		CpuIo<Cfg> io; unsigned mi = (uintptr_t)in & 3, mo = (uintptr_t)out.data() & 3;
		io.in_org = in - mi; io.in_end = buf.data() + buf.size(); io.out_org = out.data() - mo;
		ssv::LaneInflate<Cfg> L; L.start(mi, clen, mo, isize);
		while (L.state != ssv::ST_DONE) {
			if (L.wants_refill()) { for (int j = 0; j < Cfg::REFILL_DW; ++j) { uint32_t D = L.rfill + j; io.in_ring[D % Cfg::IN_DW] = io.in_stream32(4 * D); } L.rfill += Cfg::REFILL_DW; }
			int st0 = L.state; uint32_t o0 = L.o;
			L.step(io, tab);
			if (st0 == ssv::ST_HEADER) ++hdrs;
			if (st0 == ssv::ST_SYMBOL && L.o == o0 + 1 && L.state == ssv::ST_SYMBOL) { ++lit; ++bytes_lit; }
			if (st0 <= ssv::ST_SYMBOL && (L.state == ssv::ST_FAR || (L.rem + (L.o - o0) > 1 && (L.state == ssv::ST_COPY || (L.state == ssv::ST_SYMBOL && L.o - o0 >= 3))))) {
				uint32_t len = L.rem + (L.o - o0); ++nmatch; bytes_match += len;
				int db = 0; while ((1u << (db + 1)) <= L.dist) ++db; dist_hist[db]++; dist_bytes[db] += len;
				int lb = len <= 3 ? 0 : len <= 4 ? 1 : len <= 6 ? 2 : len <= 8 ? 3 : len <= 12 ? 4 : len <= 16 ? 5 : len <= 32 ? 6 : len <= 64 ? 7 : len <= 128 ? 8 : 9; len_hist[lb]++;
			}
			if (L.state == ssv::ST_FAR) {
				auto m = L.far_view(); uint32_t nd = m.dwords(); std::vector<uint32_t> val(nd), msk(nd);
				for (uint32_t j = 0; j < nd; ++j) { uint32_t q0 = m.pos(j), srcv = 0; if (m.needs_src(j)) for (int b = 0; b < 4; ++b) { long sp = (long)q0 + b - (long)m.dist; if (sp >= (long)L.o_begin) srcv |= (uint32_t)io.out_org[sp] << (8 * b); } msk[j] = m.merge(j, io.out_get32((q0 >> 2) % Cfg::OUT_DW), srcv, val[j]); }
				for (uint32_t j = 0; j < nd; ++j) { uint32_t q0 = m.pos(j); for (int b = 0; b < 4; ++b) if (msk[j] >> b & 1) io.out_org[q0 + b] = val[j] >> (8 * b); if (m.to_ring(j)) io.out_set32((q0 >> 2) % Cfg::OUT_DW, val[j]); }
				L.far_done();
			}
			if (L.wants_flush()) { bool fin = L.state == ssv::ST_FINISH; uint32_t hi = fin ? L.o : L.f + 4u * Cfg::FLUSH_DW; for (uint32_t q = L.f; q < hi; ++q) if (q >= L.o_begin && q < L.o_end) io.out_org[q] = io.out_ring[q % (4u * Cfg::OUT_DW)]; if (fin) L.state = ssv::ST_DONE; else L.f += 4u * Cfg::FLUSH_DW; }
		}
		if (L.verdict() != 0) { printf("block %d rc %d\n", nblk, L.verdict()); return 2; }
		off += bsize; ++nblk;
	}
	printf("blocks %d headers %ld literals %ld matches %ld bytes_lit %ld bytes_match %ld avg_len %.1f\n", nblk, hdrs, lit, nmatch, bytes_lit, bytes_match, (double)bytes_match / nmatch);
	printf("dist (log2 bins): "); for (int i = 0; i < 16; ++i) printf("%d:%ld(%ldB) ", 1 << i, dist_hist[i], dist_bytes[i]); printf("\nlen bins <=3,4,6,8,12,16,32,64,128,258: "); for (int i = 0; i < 10; ++i) printf("%ld ", len_hist[i]); printf("\n");
	return 0;
}
