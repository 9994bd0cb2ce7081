// table3_kernels.h - the compact cluster table (ssv_clip_table_format 3): what crosses PCIe at the end of getclip, cut to what the host
// cannot rebuild.  The reference holds this state in two multimaps of strings (clip_reads.cpp:57-108, 260-283) and prints it
// (clip_reads.h:300-345); here it is, per cluster:
//   fixed columns   pos i32 | left_len, right_len (u16 or u32 each) | support (u16 or u32) | n_cigar (u8 or u16) | flags u8  = 12 bytes
//                   (contig and side come from a handful of (contig, side, first cluster) runs; string / CIGAR offsets are prefix sums)
//   CIGAR           n_cigar operations
//   string block    [base stream | quality stream], each a whole number of dwords: the n = left_len + right_len bases of seq_left + seq_right
//                   at base_bits (2: index into "ACGT"; 4: BAM's code) each, base i at stream bits [i * base_bits, +base_bits) (bit b of a
//                   stream = bit b % 32 of dword b / 32), then the n qualities at qual_bits each the same way (8: characters, phred + 33),
//                   or in groups of qual_group qualities, qual_bits per group (below)
//   exceptions      with 2-bit bases, every base that is not A/C/G/T: (cluster, base index, BAM code) - the stream holds 0 there
// A single-event cluster (97 % of a WGS sample) is a contiguous piece of its read: n nibbles from nibble `begin` of the packed bases and n
// bytes from quality `begin`.  No piece of the block starts inside a dword, so every lane composes whole output dwords straight from the
// staged read: no second staging, no byte merging - half the instructions of the four-piece layout of formats 1 and 2, and 60 % of the bytes.
#pragma once

#include "clip_kernels.h"

namespace ssv {

struct TableRun {
	int32_t tid;
	uint8_t side, pad[3];
	int64_t first; // dense index of the run's first cluster
};
static_assert(sizeof(TableRun) == 16, "ssv_table_run");

struct Pack3Args {
	int32_t *pos;
	void *len;                // [n][2] u16 or u32
	void *support;            // [n] u16 or u32
	void *ncig;               // [n] u8 or u16
	uint8_t *flags;           // [n] bit 0: the reference prints "*" for the qualities
	int len_bytes, support_bytes, ncig_bytes, base_bits;
	TableRun *runs;
	unsigned int *run_count;
	uint64_t *exc;            // cluster << 28 | base index << 4 | code
	unsigned int *exc_count;
	uint32_t exc_cap;
	int *support_miss;        // a support count that does not fit support_bytes
	int cig_bytes;            // 2: a CIGAR operation as length << 4 | code in 16 bits (lengths below 4096); 4: BAM's 32 bits
	int *cig_miss;            // an operation that does not fit cig_bytes
	int *exc_miss;            // more exceptions than exc_cap (or a base index / cluster index beyond the entry's fields)
};

__host__ __device__ __forceinline__ uint64_t table3_block_bytes(uint64_t n, int base_bits, int qual_bits, int qual_group = 1)
{
	return 4ull * ((n * (uint64_t)base_bits + 31) / 32 + (qual_stream_bits(n, (uint64_t)qual_bits, (uint64_t)qual_group) + 31) / 32);
}

// Grouped qualities (round 4): with an alphabet of R values, k alphabet indices i_0 .. i_(k-1) go into the stream as ONE number i_0 + R i_1 + R^2 i_2 + ...
// of B bits - five values: three to 7 bits (2.33 bits a quality instead of 3), nine to eleven values: two to 7 bits (3.5 instead of 4).  Group g of a stream lies at stream bits [g B, g B + B); a last group that the qualities do not fill holds index 0
// in its unused places.  The table is what crosses PCIe and qualities are half of it.
// one group from its qualities, the generic way (the LDS-staged and the bytewise kernels): ph(i) = phred value of quality i of the stream
template <class PhredAt, class Seen>
__device__ __forceinline__ uint32_t qual_group_code(const uint8_t *s_lut, int K, uint32_t R, int i0, int n, uint32_t &miss, PhredAt ph, Seen seen)
{
	uint32_t code = 0, mul = 1;
	for (int j = 0; j < K && i0 + j < n; ++j) {
		const uint32_t v = ph(i0 + j), idx = s_lut[v];
		seen(v);
		miss |= idx;
		code += (idx & 0x7fu) * mul;
		mul *= R;
	}
	return code;
}

// one thread per sorted slot: the row's fixed columns, its CIGAR, the descriptor for the string kernels, and the (contig, side) runs
__global__ __launch_bounds__(BLOCK) void k_cluster_cols3(PackArgs p, Pack3Args q, PackDesc *__restrict__ desc, uint32_t *__restrict__ out_cig)
{
	const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= p.c.E) return;
	const uint64_t key = p.c.skey[j];
	if (j == 0 || (p.c.skey[j - 1] >> 32) != (key >> 32)) { // first slot of a (contig, side): the next cluster at or after it starts the run
		TableRun r;
		r.tid = (int32_t)(key >> 33); r.side = ((key >> 32) & 1ull) ? '3' : '5'; r.pad[0] = r.pad[1] = r.pad[2] = 0;
		r.first = (int64_t)(uint32_t)p.slot_cnt[j];
		q.runs[atomicAdd(q.run_count, 1u)] = r; // (a few per contig; the host orders them by `first`)
	}
	const SlotCluster s = slot_cluster_load(p, j);
	if (!s.cluster) return;
	const uint32_t c = s.c;
	q.pos[c] = (int32_t)(uint32_t)s.key;
	if (q.len_bytes == 2) reinterpret_cast<uint32_t *>(q.len)[c] = (uint32_t)s.ll | ((uint32_t)s.lr << 16);
	else reinterpret_cast<uint2 *>(q.len)[c] = make_uint2((uint32_t)s.ll, (uint32_t)s.lr);
	if (q.support_bytes == 2) {
		if (s.support > 0xffff) *q.support_miss = 1;
		reinterpret_cast<uint16_t *>(q.support)[c] = (uint16_t)s.support;
	} else reinterpret_cast<uint32_t *>(q.support)[c] = (uint32_t)s.support;
	if (q.ncig_bytes == 1) reinterpret_cast<uint8_t *>(q.ncig)[c] = (uint8_t)s.ncg;
	else reinterpret_cast<uint16_t *>(q.ncig)[c] = (uint16_t)s.ncg;
	if (!s.single) q.flags[c] = s.qmiss_multi ? 1 : 0;
	if (q.cig_bytes == 2) { // (150-base reads have no operation of 4096 bases; a table that has one is built again with 32-bit operations)
		uint16_t *dc = reinterpret_cast<uint16_t *>(out_cig) + s.cig_off;
		uint32_t wide = 0;
		if (s.ncg <= 5) {
			if (s.ncg > 0) { dc[0] = (uint16_t)s.cg0; wide |= s.cg0; }
			if (s.ncg > 1) { dc[1] = (uint16_t)s.cg1; wide |= s.cg1; }
			if (s.ncg > 2) { dc[2] = (uint16_t)s.cg2; wide |= s.cg2; }
			if (s.ncg > 3) { dc[3] = (uint16_t)s.cg3; wide |= s.cg3; }
			if (s.ncg > 4) { dc[4] = (uint16_t)s.cg4; wide |= s.cg4; }
		} else {
			const gptr<uint32_t> src = global_at<uint32_t>(s.cig_ptr);
			for (int i = 0; i < s.ncg; ++i) { dc[i] = (uint16_t)src[i]; wide |= src[i]; }
		}
		if (wide >> 16) *q.cig_miss = 1;
	} else {
		uint32_t *dc = out_cig + s.cig_off;
		if (s.ncg <= 5) {
			if (s.ncg > 0) dc[0] = s.cg0;
			if (s.ncg > 1) dc[1] = s.cg1;
			if (s.ncg > 2) dc[2] = s.cg2;
			if (s.ncg > 3) dc[3] = s.cg3;
			if (s.ncg > 4) dc[4] = s.cg4;
		} else {
			const gptr<uint32_t> src = global_at<uint32_t>(s.cig_ptr);
			for (int i = 0; i < s.ncg; ++i) dc[i] = src[i];
		}
	}
	uint4 *dd = reinterpret_cast<uint4 *>(desc + c);
	dd[0] = make_uint4((uint32_t)s.src, (uint32_t)(s.src >> 32), (uint32_t)s.str_off, (uint32_t)(s.str_off >> 32));
	dd[1] = make_uint4((uint32_t)s.lq, (uint32_t)s.ll, (uint32_t)s.lr, s.single ? (uint32_t)s.begin : (uint32_t)j);
}

// ---- the same with the scans over TILES instead of slots (round 6).  k_cluster_meta, the two device-wide scans behind it and k_cluster_cols3 carried two 8-byte
// words per slot through memory four times (written, scanned twice, read: 0.5 GB a step); here
//   k_cluster_tile_sums  adds up, per tile of 256 slots, what its clusters put into the table (clusters, CIGAR operations, string bytes) and lists the long reads,
//   k_scan_sums_lists    scans the three lists of tile sums (a workgroup each: 25 K tiles a step, 35 us - letting every workgroup of the third kernel add up the
//                        tiles before its own instead, through sums of groups of 64 tiles, cost the two other kernels what it saved),
//   k_cluster_cols3_tiles works the slot's words out again from the line it reads anyway, scans them inside the tile and writes the rows.
// (A single pass with a look-back from tile to tile was built first and measured: 331-397 us against 345 for what it replaced - 25 K tiles of 256 slots are
// 20 generations of workgroups, each waiting out four dependent trips to memory plus its predecessors'; handing tiles out by a ticket costs as much again,
// 25 K atomics on one address at ~90 a microsecond.)
struct TileSums {
	uint64_t *clusters, *cig, *bytes; // [tiles] each
};

// what a table row needs of slot j beyond SlotCluster: the words k_cluster_meta worked out
struct SlotWords { uint64_t cnt, bytes; bool lng; };
__device__ __forceinline__ SlotWords slot_words(const PackArgs &p, const SlotCluster &s)
{
	SlotWords w{0, 0, false};
	if (s.cluster) {
		w.cnt = 1ull | ((uint64_t)(uint32_t)s.ncg << 32);
		w.bytes = 4ull * (((uint64_t)(s.ll + s.lr) * (uint64_t)p.base_bits + 31) / 32 + (qual_stream_bits((uint64_t)(s.ll + s.lr), (uint64_t)p.qual_bits, (uint64_t)p.qual_group) + 31) / 32);
		w.lng = p.packed && s.single && s.lq > PACK_MAX_LQ;
	}
	return w;
}

__global__ __launch_bounds__(BLOCK) void k_cluster_tile_sums(PackArgs p, TileSums ts)
{
	__shared__ uint64_t s_cnt[WAVES_PER_BLOCK], s_bytes[WAVES_PER_BLOCK];
	const int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	uint64_t cnt = 0, bytes = 0;
	bool lng = false;
	if (j < p.c.E && p.c.support[j] > 0) { // (only what the sums need of the line: its second and third quarter)
		const bool single = !p.c.mflag[j] || p.c.support[j] == 1;
		const uint32_t e = single ? (uint32_t)j : p.c.c_cig_ev[j];
		const uint4 *ep = reinterpret_cast<const uint4 *>(p.c.ev + e);
		const uint4 eb = ep[1], ec = ep[2];
		const uint64_t n = (uint64_t)(single ? (int)eb.w : p.c.c_ll[j]) + (uint64_t)(single ? (int)ec.x : p.c.c_lr[j]);
		cnt = 1ull | ((uint64_t)ec.z << 32);
		bytes = 4ull * ((n * (uint64_t)p.base_bits + 31) / 32 + (qual_stream_bits(n, (uint64_t)p.qual_bits, (uint64_t)p.qual_group) + 31) / 32);
		lng = p.packed && single && (int)ec.y > PACK_MAX_LQ;
	}
	// one atomic per wavefront that holds long reads (same-address atomics run at ~90 per microsecond: with long-read data every slot is listed)
	const uint64_t lm = __ballot(lng);
	if (lm) {
		uint32_t base = 0;
		if (lane_id() == 0) base = atomicAdd(p.slow_count, (unsigned int)__popcll(lm));
		base = __shfl(base, 0, WAVE);
		if (lng) p.slow_list[base + (uint32_t)__popcll(lm & lanemask_lt())] = (uint32_t)j;
	}
	cnt = wave_sum(cnt); bytes = wave_sum(bytes);
	if (lane_id() == 0) { s_cnt[wave_id()] = cnt; s_bytes[wave_id()] = bytes; }
	__syncthreads();
	if (threadIdx.x == 0) {
		uint64_t c = 0, b = 0;
		for (int w = 0; w < WAVES_PER_BLOCK; ++w) { c += s_cnt[w]; b += s_bytes[w]; }
		ts.clusters[blockIdx.x] = c & 0xffffffffull; ts.cig[blockIdx.x] = c >> 32; ts.bytes[blockIdx.x] = b;
	}
}

// ts: the tile sums, scanned (exclusive); tot: [0] clusters | CIGAR operations << 32, [1] string bytes - the totals, written by the last tile
__global__ __launch_bounds__(BLOCK) void k_cluster_cols3_tiles(PackArgs p, Pack3Args q, PackDesc *__restrict__ desc, uint32_t *__restrict__ out_cig, TileSums ts, uint64_t *__restrict__ tot)
{
	__shared__ uint64_t lds[WAVES_PER_BLOCK + 1];
	const int64_t t = (int64_t)blockIdx.x;
	const int64_t j = t * BLOCK + threadIdx.x;
	const uint64_t s_pre[3] = {scalar_load(ts.clusters + t), scalar_load(ts.cig + t), scalar_load(ts.bytes + t)}; // (scanned: k_scan_sums_lists)
	SlotCluster s = {};
	uint64_t key = 0;
	bool run_start = false;
	if (j < p.c.E) {
		s = slot_cluster_load<false>(p, j);
		key = s.key;
		run_start = j == 0 || (p.c.skey[j - 1] >> 32) != (key >> 32);
	}
	const SlotWords sw = slot_words(p, s);
	const bool lng = sw.lng;
	uint64_t tot_cnt, tot_bytes;
	const uint64_t ex_cnt = block_exclusive_sum(sw.cnt, lds, &tot_cnt);
	const uint64_t ex_bytes = block_exclusive_sum(sw.bytes, lds, &tot_bytes);
	if (threadIdx.x == 0 && t == (int64_t)gridDim.x - 1) { tot[0] = (s_pre[0] + (tot_cnt & 0xffffffffull)) | ((s_pre[1] + (tot_cnt >> 32)) << 32); tot[1] = s_pre[2] + tot_bytes; }
	if (j >= p.c.E) return;
	const uint32_t c = (uint32_t)(s_pre[0] + (ex_cnt & 0xffffffffull));
	if (run_start) { // first slot of a (contig, side): the next cluster at or after it starts the run
		TableRun r;
		r.tid = (int32_t)(key >> 33); r.side = ((key >> 32) & 1ull) ? '3' : '5'; r.pad[0] = r.pad[1] = r.pad[2] = 0;
		r.first = (int64_t)c;
		q.runs[atomicAdd(q.run_count, 1u)] = r;
	}
	if (!s.cluster) return;
	s.cig_off = s_pre[1] + (ex_cnt >> 32);
	s.str_off = s_pre[2] + ex_bytes;
	if (!s.single || lng) { p.slot_cnt[j] = (uint64_t)c | (s.cig_off << 32); p.slot_bytes[j] = s.str_off; } // (the base-by-base kernel finds its clusters by slot: k_pack3_slow)
	q.pos[c] = (int32_t)(uint32_t)s.key;
	if (q.len_bytes == 2) reinterpret_cast<uint32_t *>(q.len)[c] = (uint32_t)s.ll | ((uint32_t)s.lr << 16);
	else reinterpret_cast<uint2 *>(q.len)[c] = make_uint2((uint32_t)s.ll, (uint32_t)s.lr);
	if (q.support_bytes == 2) {
		if (s.support > 0xffff) *q.support_miss = 1;
		reinterpret_cast<uint16_t *>(q.support)[c] = (uint16_t)s.support;
	} else reinterpret_cast<uint32_t *>(q.support)[c] = (uint32_t)s.support;
	if (q.ncig_bytes == 1) reinterpret_cast<uint8_t *>(q.ncig)[c] = (uint8_t)s.ncg;
	else reinterpret_cast<uint16_t *>(q.ncig)[c] = (uint16_t)s.ncg;
	if (!s.single) q.flags[c] = s.qmiss_multi ? 1 : 0;
	if (q.cig_bytes == 2) {
		uint16_t *dc = reinterpret_cast<uint16_t *>(out_cig) + s.cig_off;
		uint32_t wide = 0;
		if (s.ncg <= 5) {
			if (s.ncg > 0) { dc[0] = (uint16_t)s.cg0; wide |= s.cg0; }
			if (s.ncg > 1) { dc[1] = (uint16_t)s.cg1; wide |= s.cg1; }
			if (s.ncg > 2) { dc[2] = (uint16_t)s.cg2; wide |= s.cg2; }
			if (s.ncg > 3) { dc[3] = (uint16_t)s.cg3; wide |= s.cg3; }
			if (s.ncg > 4) { dc[4] = (uint16_t)s.cg4; wide |= s.cg4; }
		} else {
			const gptr<uint32_t> src = global_at<uint32_t>(s.cig_ptr);
			for (int i = 0; i < s.ncg; ++i) { dc[i] = (uint16_t)src[i]; wide |= src[i]; }
		}
		if (wide >> 16) *q.cig_miss = 1;
	} else {
		uint32_t *dc = out_cig + s.cig_off;
		if (s.ncg <= 5) {
			if (s.ncg > 0) dc[0] = s.cg0;
			if (s.ncg > 1) dc[1] = s.cg1;
			if (s.ncg > 2) dc[2] = s.cg2;
			if (s.ncg > 3) dc[3] = s.cg3;
			if (s.ncg > 4) dc[4] = s.cg4;
		} else {
			const gptr<uint32_t> src = global_at<uint32_t>(s.cig_ptr);
			for (int i = 0; i < s.ncg; ++i) dc[i] = src[i];
		}
	}
	uint4 *dd = reinterpret_cast<uint4 *>(desc + c);
	dd[0] = make_uint4((uint32_t)s.src, (uint32_t)(s.src >> 32), (uint32_t)s.str_off, (uint32_t)(s.str_off >> 32));
	dd[1] = make_uint4((uint32_t)s.lq, (uint32_t)s.ll, (uint32_t)s.lr, s.single ? (uint32_t)s.begin : (uint32_t)j);
}

__device__ __forceinline__ void exc_append(const Pack3Args &q, uint64_t cluster, int base, uint32_t code)
{
	if (base >= (1 << 24) || cluster >= (1ull << 36)) { *q.exc_miss = 1; return; }
	const uint32_t k = atomicAdd(q.exc_count, 1u);
	if (k < q.exc_cap) q.exc[k] = (cluster << 28) | ((uint64_t)(uint32_t)base << 4) | (uint64_t)code;
	else *q.exc_miss = 1;
}

// eight BAM codes (nibble k of w = base k) -> eight 2-bit indices into "ACGT" (bits [2k, 2k + 2) of the result); A = 1, C = 2, G = 4, T = 8
__device__ __forceinline__ uint32_t acgt2(uint32_t w)
{
	const uint32_t b0 = ((w >> 1) | (w >> 3)) & 0x11111111u; // C or T
	const uint32_t b1 = ((w >> 2) | (w >> 3)) & 0x11111111u; // G or T
	uint32_t x = b0 | (b1 << 1);
	x = (x | (x >> 2)) & 0x0f0f0f0fu;
	x = (x | (x >> 4)) & 0x00ff00ffu;
	return (x | (x >> 8)) & 0xffffu;
}

// nibbles of w (under `valid`, a mask of whole nibbles) that are not exactly one of A, C, G, T
__device__ __forceinline__ bool not_acgt(uint32_t w, uint32_t valid)
{
	const uint32_t M = 0x11111111u;
	const uint32_t s = (w & M) + ((w >> 1) & M) + ((w >> 2) & M) + ((w >> 3) & M); // bits set per nibble (<= 4: no carries)
	return s != (valid & M);
}

// the rare dword with something else in it: base by base (the stream gets 0 for such a base, the exception list says what it was)
__device__ __forceinline__ uint32_t acgt2_slow(const Pack3Args &q, uint64_t cluster, int base0, uint32_t w, int count)
{
	uint32_t out = 0;
	for (int k = 0; k < count; ++k) {
		const uint32_t n = (w >> (4 * k)) & 15u;
		if (n == 1u || n == 2u || n == 4u || n == 8u) out |= (uint32_t)(__ffs((int)n) - 1) << (2 * k);
		else exc_append(q, cluster, base0 + k, n);
	}
	return out;
}

__device__ __forceinline__ uint32_t swap_nibbles(uint32_t x) { return ((x & 0x0f0f0f0fu) << 4) | ((x >> 4) & 0x0f0f0f0fu); }

// 16 lanes per cluster, four clusters per wavefront.  Single-event clusters of reads up to PACK_MAX_LQ bases: the read's entry (packed bases,
// qualities) goes into LDS from the dword-aligned address below it, then every lane composes output dwords:
//   base dword t   = bases [16 t, 16 t + 16) (BB = 2; 8 t .. for BB = 4) = a 9-byte window of the packed bases, nibbles swapped into stream order
//                    and shifted by the odd nibble, then eight codes -> 16 bits twice
//   quality dword t = stream bits [32 t, 32 t + 32) of the W-bit alphabet indices = up to 12 source bytes through the look-up table
// and stores them: the block is its own staging.  TRACK: also note every phred value met (the second launch after a quality outside the guessed
// alphabet turned up).
template <int W, int BB, bool TRACK>
__global__ __launch_bounds__(BLOCK) void k_pack3_stream(PackArgs p, Pack3Args q, const PackDesc *__restrict__ desc, const unsigned int *__restrict__ n_clusters_dev, uint8_t *__restrict__ out_str)
{
	__shared__ uint32_t s_raw[GROUPS_PER_BLOCK][PACK_RAW_DWORDS + 12]; // (+ what the unconditional look-ahead of the last dwords can touch)
	__shared__ uint8_t s_lut[256]; // phred -> alphabet index
	__shared__ uint32_t s_seen[8];  // phred values met (bit set)
	if (W < 8) s_lut[threadIdx.x] = p.qlut[threadIdx.x]; // BLOCK == 256
	if (threadIdx.x < 8) s_seen[threadIdx.x] = 0;
	const int grp = (int)(threadIdx.x / GROUP);
	const int gl = (int)(threadIdx.x % GROUP);
	const int64_t nc = (int64_t)*n_clusters_dev;
	const int64_t c = (int64_t)blockIdx.x * GROUPS_PER_BLOCK + grp;
	const uint32_t *s4 = s_raw[grp];
	const PackDescR d0 = pack_desc_load(desc, c, nc);
	const bool fast = pack_desc_fast(d0);
	{
		uint32_t r[PACK_RAW_PER_LANE];
		pack_entry_load(d0, gl, r);
		const int mis0 = (int)(d0.src & 3ull);
		const int nraw = fast ? (mis0 + (d0.lq + 1) / 2 + d0.lq + 3) / 4 + 1 : 0;
#pragma unroll
		for (int u = 0; u < PACK_RAW_PER_LANE; ++u) { const int i = gl + GROUP * u; if (i < nraw && i < PACK_RAW_DWORDS) s_raw[grp][i] = r[u]; }
	}
	__syncthreads(); // s_raw, s_lut, s_seen
	if (fast) {
		const int n = d0.ll + d0.lr, lq = d0.lq, begin = d0.begin;
		const int mis = (int)(d0.src & 3ull);     // the entry starts mis bytes into s_raw
		const int qb = mis + (lq + 1) / 2;        // first quality byte of the entry inside s_raw
		const bool qmiss = lq > 0 && ((s4[qb >> 2] >> (8 * (qb & 3))) & 0xffu) == 0xffu;
		if (gl == 0) q.flags[c] = qmiss ? 1 : 0;
		constexpr int PER = 32 / BB; // bases per dword
		const int nDb = (n * BB + 31) / 32;
		const int nDq = (int)((qual_stream_bits((uint64_t)n, W ? (uint64_t)W : (uint64_t)p.qual_bits, W ? 1ull : (uint64_t)p.qual_group) + 31) / 32); // W == 0: grouped qualities, shape at run time
		uint32_t *d = reinterpret_cast<uint32_t *>(out_str + d0.str_off);
		for (int t = gl; t < nDb; t += GROUP) {
			const int nb0 = begin + PER * t;                  // first nibble of the read
			const int B = mis + (nb0 >> 1);
			const uint32_t *w = s4 + (B >> 2);
			const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];     // (s_raw holds one dword of read-ahead; beyond it stale LDS, masked below)
			const uint32_t sh = (uint32_t)(B & 3);
			uint32_t lo = swap_nibbles(__builtin_amdgcn_alignbyte(w1, w0, sh));
			uint32_t hi = swap_nibbles(__builtin_amdgcn_alignbyte(w2, w1, sh));
			if (nb0 & 1) {
				const uint32_t ex = BB == 2 ? (w2 >> (8 * sh)) >> 4 : 0u; // high nibble of byte B + 8 (sh <= 3: still inside w2)
				lo = __builtin_amdgcn_alignbit(hi, lo, 4);
				hi = (hi >> 4) | (ex << 28);
			}
			const int rem = n - PER * t; // bases of the stream from this dword on
			if (BB == 4) {
				if (rem < 8) lo &= (1u << (4 * rem)) - 1u;
				d[t] = lo;
			} else {
				const uint32_t vlo = rem >= 8 ? 0xffffffffu : (1u << (4 * rem)) - 1u;
				const uint32_t vhi = rem >= 16 ? 0xffffffffu : rem > 8 ? (1u << (4 * (rem - 8))) - 1u : 0u;
				lo &= vlo; hi &= vhi;
				uint32_t a = acgt2(lo), b = acgt2(hi);
				if (not_acgt(lo, vlo)) a = acgt2_slow(q, (uint64_t)c, PER * t, lo, rem < 8 ? rem : 8);
				if (not_acgt(hi, vhi)) b = acgt2_slow(q, (uint64_t)c, PER * t + 8, hi, rem < 16 ? rem - 8 : 8);
				d[t] = a | (b << 16);
			}
		}
		uint32_t seen_lo = 0, seen_hi = 0; // TRACK: phred 0..63 as a bit set in registers (anything higher goes straight to LDS)
		uint32_t miss = 0;                 // OR of the table look-ups: 0xff marks a value outside the alphabet
		if constexpr (W == 0) { // grouped qualities, group by group out of the staged bytes (this kernel is the direct one's stand-in for alphabets with a phred value >= 64 and the TRACK launch)
			const int B = p.qual_bits, K = p.qual_group;
			const uint8_t *qbytes = reinterpret_cast<const uint8_t *>(s4) + qb + begin;
			for (int t = gl; t < nDq; t += GROUP) {
				const int g0 = (32 * t) / B, off = 32 * t - B * g0;
				uint64_t acc = 0;
				for (int g = g0; B * g < 32 * t + 32 && K * g < n; ++g)
					acc |= (uint64_t)qual_group_code(s_lut, K, (uint32_t)p.qual_radix, K * g, n, miss, [&](int i) { return (uint32_t)qbytes[i]; },
					                                 [&](uint32_t ph) { if (TRACK) { if (ph < 32) seen_lo |= 1u << ph; else if (ph < 64) seen_hi |= 1u << (ph - 32); else atomicOr(&s_seen[ph >> 5], 1u << (ph & 31)); } }) << (B * (g - g0));
				d[nDb + t] = qmiss ? 0u : (uint32_t)(acc >> off);
			}
		} else
		for (int t = gl; t < nDq; t += GROUP) {
			constexpr int CNT = W == 8 ? 4 : (W == 3 ? 12 : 32 / W); // qualities that can touch one dword
			constexpr int NSRC = (CNT + 3) / 4;                      // source dwords they lie in, once aligned
			const int i0 = (32 * t) / W, off = 32 * t - W * i0;   // off != 0 only for W = 3
			const int a = qb + begin + i0;                         // first source byte
			const uint32_t *q4 = s4 + (a >> 2);
			const uint32_t sh = (uint32_t)(a & 3);
			const int rem = n - i0;                                // qualities of the stream from i0 on (>= 1)
			// no branches: all CNT look-ups are made (what lies behind the stream's end is stale LDS inside the array) and the tail is masked off
			uint32_t src[NSRC];
			{
				uint32_t prev = q4[0];
#pragma unroll
				for (int g = 0; g < NSRC; ++g) { const uint32_t next = q4[g + 1]; src[g] = __builtin_amdgcn_alignbyte(next, prev, sh); prev = next; }
			}
			uint32_t v;
			if (W == 8) {
				v = qmiss ? 0x2a2a2a2au : src[0] + 0x21212121u; // phred + 33 (no carries: qualities <= 93); '*' when absent
				if (rem < 4) v &= (1u << (8 * rem)) - 1u;
			} else {
				uint64_t acc = 0;
#pragma unroll
				for (int jq = 0; jq < CNT; ++jq) {
					const uint32_t ph = (src[jq >> 2] >> (8 * (jq & 3))) & 0xffu;
					const uint32_t idx = s_lut[ph];
					acc |= (uint64_t)(idx & ((1u << W) - 1u)) << (jq * W);
					miss |= jq < rem ? idx : 0u;
					if (TRACK && jq < rem) { if (ph < 32) seen_lo |= 1u << ph; else if (ph < 64) seen_hi |= 1u << (ph - 32); else atomicOr(&s_seen[ph >> 5], 1u << (ph & 31)); }
				}
				if (rem < CNT) acc &= (1ull << (W * rem)) - 1ull;
				v = qmiss ? 0u : (uint32_t)(acc >> off); // no qualities: the row prints "*", the stream stays zero
			}
			d[nDb + t] = v;
		}
		if (W < 8 && !qmiss) {
			if (miss & 0x80u) *p.lut_miss = 1;
			if (TRACK) {
				if (seen_lo) atomicOr(&s_seen[0], seen_lo);
				if (seen_hi) atomicOr(&s_seen[1], seen_hi);
			}
		}
	}
	if (!TRACK || W == 8) return;
	__syncthreads();
	if (threadIdx.x < 8 && s_seen[threadIdx.x]) { // one look per workgroup; an atomic only while the set still grows
		const uint32_t have = __atomic_load_n(&p.qual_seen[threadIdx.x], __ATOMIC_RELAXED);
		if (s_seen[threadIdx.x] & ~have) atomicOr(&p.qual_seen[threadIdx.x], s_seen[threadIdx.x]);
	}
}

// ---- the same blocks without the LDS stage: every lane reads the few source dwords of its output dwords itself ----
//
// k_pack3_stream above is bound by its vector instructions (eight wavefronts per SIMD, each ~400 instructions of 4 cycles; PMC: VALU busy, memory
// idle).  Most of them shuffle bytes: stage the entry, pick bytes out of LDS dwords, one table look-up per quality, SWAR per 8 bases.  Here
//   - a lane loads the 3 (bases) / up to 5 (qualities) aligned dwords its output dword is made of straight from the batch (the read's
//     225 bytes are four lines in L1 for all 16 lanes), nothing is staged and no barrier is needed;
//   - bases go through a 256-entry table: a BAM byte (two bases) -> their two 2-bit codes, bit 7 when one of them is not A/C/G/T;
//   - qualities go through a 4096-entry table indexed by TWO phred values (6 bits each): both alphabet indices at once, bit 15 when
//     one is outside the alphabet.  (A phred value >= 64 has no place in that table: the host then launches k_pack3_stream instead.)
// Half the instructions per block.

// pair table entry e = q0 | q1 << 6: index(q0) | index(q1) << W, or 0x8000
__device__ __forceinline__ uint32_t pair_index(uint32_t halfword) { return (halfword & 0x3fu) | ((halfword >> 2) & 0xfc0u); }

// W bits per quality (K == 1) or per group of K qualities (K > 1: W = 7, or 11 for two of 17 to 45 values)
template <int W, int K = 1> struct Q3 {
	static constexpr int CNTG = 32 % W == 0 ? 32 / W : (32 + 2 * (W - 1)) / W;     // groups that can touch one dword (a group may start up to W - 1 bits in front of it)
	static constexpr int CNT = K > 1 ? K * CNTG : W == 8 ? 4 : (W == 3 ? 12 : 32 / W); // qualities that can touch one dword
	static constexpr int NSRC = (CNT + 3) / 4;                      // source dwords they lie in, once aligned
	static_assert(K == 1 || CNT % 2 == 0, "grouped qualities are looked up in pairs");
	// first quality of the stream that touches dword t, and the dwords of a stream of n qualities
	static __device__ __forceinline__ int first(int t) { return K > 1 ? K * ((32 * t) / W) : (32 * t) / W; }
	static __device__ __forceinline__ int dwords(int n) { return K > 1 ? (((n + K - 1) / K) * W + 31) / 32 : (n * W + 31) / 32; }
};

// what one lane reads for base dword t and quality dword t of a cluster: aligned dwords, nothing beyond the one that holds the last byte
template <int W, int BB, int K = 1> struct Src3 {
	uint32_t w[3];
	uint32_t raw[Q3<W, K>::NSRC + 1];
};

template <int W, int BB>
__device__ __forceinline__ void src3_load_bases(const PackDescR &d0, int t, uint32_t (&w)[3])
{
	constexpr int PER = 32 / BB;
	const int n = d0.ll + d0.lr;
	const int nb0 = d0.begin + PER * t;
	const uint64_t A = d0.src + (uint64_t)(nb0 >> 1);
	const gptr<uint32_t> p = global_at<uint32_t>(A & ~3ull); // (global_load, not flat_load: the pair tables live in LDS - common.h)
	const int rem = n - PER * t;
	const int need = (int)(A & 3ull) + (((rem < PER ? rem : PER) + (nb0 & 1) + 1) >> 1); // bytes from p[0] on that hold the dword's bases
	w[0] = p[0]; w[1] = need > 4 ? p[1] : 0u; w[2] = need > 8 ? p[2] : 0u;
}

template <int W, int BB, int K = 1>
__device__ __forceinline__ void src3_load_quals(const PackDescR &d0, int t, uint32_t (&raw)[Q3<W, K>::NSRC + 1])
{
	const int n = d0.ll + d0.lr;
	const int i0 = Q3<W, K>::first(t);
	const uint64_t A = d0.src + (uint64_t)((d0.lq + 1) / 2 + d0.begin + i0); // first source byte
	const gptr<uint32_t> q4 = global_at<uint32_t>(A & ~3ull);
	const int bytes = (int)(A & 3ull) + (n - i0);
#pragma unroll
	for (int g = 0; g <= Q3<W, K>::NSRC; ++g) raw[g] = 4 * g < bytes ? q4[g] : 0u;
}

// base dword t of the cluster's block from its three source dwords
template <int BB>
__device__ __forceinline__ uint32_t base_dword3(const Pack3Args &q, const uint8_t *s_base, const PackDescR &d0, int64_t c, int t, const uint32_t (&w)[3])
{
	constexpr int PER = 32 / BB;
	const int n = d0.ll + d0.lr;
	const int nb0 = d0.begin + PER * t;                       // first nibble of the read
	const uint32_t sh = (uint32_t)((d0.src + (uint64_t)(nb0 >> 1)) & 3ull);
	uint32_t lo = __builtin_amdgcn_alignbyte(w[1], w[0], sh), hi = __builtin_amdgcn_alignbyte(w[2], w[1], sh);
	const int rem = n - PER * t;                              // bases of the stream from this dword on
	if (BB == 4) {
		lo = swap_nibbles(lo);
		if (nb0 & 1) lo = __builtin_amdgcn_alignbit(swap_nibbles(hi), lo, 4);
		if (rem < 8) lo &= (1u << (4 * rem)) - 1u;
		return lo;
	}
	if (nb0 & 1) { // pair the nibbles anew: byte k = low nibble of byte k, high nibble of byte k + 1
		const uint32_t ex = (w[2] >> (8 * sh)) & 0xffu;       // byte 8
		const uint32_t lo12 = __builtin_amdgcn_alignbit(hi, lo, 12), hi12 = (hi >> 12) | (ex << 20);
		lo = ((lo & 0x0f0f0f0fu) << 4) | (lo12 & 0x0f0f0f0fu);
		hi = ((hi & 0x0f0f0f0fu) << 4) | (hi12 & 0x0f0f0f0fu);
	}
	if (rem < 16) { // behind the stream's end: make it 'A' (valid for the table), cut the result below
		const int nb = (rem + 1) >> 1;                    // bytes that hold stream bases
		const uint64_t keep = nb >= 8 ? ~0ull : (1ull << (8 * nb)) - 1ull;
		uint64_t Y = ((uint64_t)hi << 32) | lo;
		Y = (Y & keep) | (0x1111111111111111ull & ~keep);
		if (rem & 1) Y = (Y & ~(0x0full << (8 * (nb - 1)))) | (0x01ull << (8 * (nb - 1)));
		lo = (uint32_t)Y; hi = (uint32_t)(Y >> 32);
	}
	uint32_t out = 0, inv = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const uint32_t e0 = s_base[(lo >> (8 * k)) & 0xffu], e1 = s_base[(hi >> (8 * k)) & 0xffu];
		out |= ((e0 & 15u) << (4 * k)) | ((e1 & 15u) << (4 * k + 16));
		inv |= e0 | e1;
	}
	if (inv & 0x80u) { // something else than A/C/G/T among the 16: base by base (nibbles into stream order first)
		const uint32_t slo = swap_nibbles(lo), shi = swap_nibbles(hi);
		const uint32_t a = acgt2_slow(q, (uint64_t)c, PER * t, slo, rem < 8 ? rem : 8);
		const uint32_t b = rem > 8 ? acgt2_slow(q, (uint64_t)c, PER * t + 8, shi, rem < 16 ? rem - 8 : 8) : 0u;
		out = a | (b << 16);
	} else if (rem < 16) out &= (1u << (2 * rem)) - 1u;
	return out;
}

// quality dword t of the cluster's block from its source dwords; `miss` collects what the pair table could not place
template <int W>
__device__ __forceinline__ uint32_t qual_dword3(const Pack3Args &q, const uint16_t *s_pair, const PackDescR &d0, int64_t c, int t, const uint32_t (&raw)[Q3<W>::NSRC + 1], uint32_t &miss)
{
	constexpr int CNT = Q3<W>::CNT, NSRC = Q3<W>::NSRC;
	const int n = d0.ll + d0.lr;
	const int i0 = (32 * t) / W, off = 32 * t - W * i0;   // off != 0 only for W = 3
	const uint32_t sh = (uint32_t)((d0.src + (uint64_t)((d0.lq + 1) / 2 + d0.begin + i0)) & 3ull);
	const int rem = n - i0;                                // qualities of the stream from i0 on (>= 1)
	uint32_t src[NSRC];
#pragma unroll
	for (int g = 0; g < NSRC; ++g) src[g] = __builtin_amdgcn_alignbyte(raw[g + 1], raw[g], sh);
	const bool qmiss = (src[0] & 0xffu) == 0xffu; // a read without qualities has 0xff in all of them
	if (t == 0) q.flags[c] = qmiss ? 1 : 0;
	if (W == 8) {
		uint32_t v = qmiss ? 0x2a2a2a2au : src[0] + 0x21212121u; // phred + 33 (no carries: qualities <= 93); '*' when absent
		if (rem < 4) v &= (1u << (8 * rem)) - 1u;
		return v;
	}
	const uint32_t fill = (src[0] & 0xffu) * 0x01010101u; // behind the stream's end: the first quality again (so that the pair table sees alphabet members only)
	uint64_t acc = 0;
	uint32_t bad = 0;
#pragma unroll
	for (int g = 0; g < NSRC; ++g) {
		const uint32_t valid = rem >= 4 * g + 4 ? 0xffffffffu : rem > 4 * g ? (1u << (8 * (rem - 4 * g))) - 1u : 0u;
		const uint32_t x = (src[g] & valid) | (fill & ~valid);
		const uint32_t e0 = s_pair[pair_index(x)], e1 = s_pair[pair_index(x >> 16)];
		acc |= (uint64_t)(e0 & 0xffu) << (4 * g * W);
		acc |= (uint64_t)(e1 & 0xffu) << ((4 * g + 2) * W);
		bad |= (x & 0xc0c0c0c0u) | ((e0 | e1) & 0x8000u); // the table knows nothing of phred >= 64, and says 0x8000 for a pair with a value outside the alphabet
	}
	if (rem < CNT) acc &= (1ull << (W * rem)) - 1ull;
	miss |= qmiss ? 0u : bad;
	return qmiss ? 0u : (uint32_t)(acc >> off); // no qualities: the row prints "*", the stream stays zero
}

// The same for grouped qualities (K alphabet indices as the digits of one W-bit number, radix R): the pair table's entry for two phred values is
//   index(q0) | index(q1) << 4 | (index(q0) + R index(q1)) << 8, bit 15: one of them is outside the alphabet
// and a group's number is put together from the pairs it is made of (two qualities: the pair's own sum; three: a pair and the half of the next one, ...).
// `fill` = the phred value of alphabet index 0 in all four bytes: what stands in behind the stream's end, so that a last group's unused places hold 0.
template <int W, int K>
__device__ __forceinline__ uint32_t qual_dword3g(const Pack3Args &q, const uint16_t *s_pair, const PackDescR &d0, int64_t c, int t, const uint32_t (&raw)[Q3<W, K>::NSRC + 1], uint32_t &miss,
                                                 uint32_t fill, uint32_t R)
{
	constexpr int CNTG = Q3<W, K>::CNTG, NSRC = Q3<W, K>::NSRC, NP = Q3<W, K>::CNT / 2;
	static_assert(K == 2 || K == 3, "group shapes of set_alphabet (seeksv_hip.hip)");
	static_assert(K == 2 || CNTG % 2 == 0, "odd groups are put together two at a time");
	const int n = d0.ll + d0.lr;
	const int g0 = (32 * t) / W, off = 32 * t - W * g0, i0 = K * g0;
	const uint32_t sh = (uint32_t)((d0.src + (uint64_t)((d0.lq + 1) / 2 + d0.begin + i0)) & 3ull);
	const int rem = n - i0;                                // qualities of the stream from i0 on (>= 1)
	uint32_t src[NSRC];
#pragma unroll
	for (int g = 0; g < NSRC; ++g) src[g] = __builtin_amdgcn_alignbyte(raw[g + 1], raw[g], sh);
	const bool qmiss = (src[0] & 0xffu) == 0xffu; // a read without qualities has 0xff in all of them
	if (t == 0) q.flags[c] = qmiss ? 1 : 0;
	uint32_t e[NP];
	uint32_t bad = 0;
#pragma unroll
	for (int g = 0; g < NSRC; ++g) {
		const uint32_t valid = rem >= 4 * g + 4 ? 0xffffffffu : rem > 4 * g ? (1u << (8 * (rem - 4 * g))) - 1u : 0u;
		const uint32_t x = (src[g] & valid) | (fill & ~valid);
		bad |= x & 0xc0c0c0c0u; // the table knows nothing of phred >= 64
		e[2 * g] = s_pair[pair_index(x)];
		bad |= e[2 * g] & 0x8000u;
		if (2 * g + 1 < NP) { e[2 * g + 1] = s_pair[pair_index(x >> 16)]; bad |= e[2 * g + 1] & 0x8000u; }
	}
	auto sum = [](uint32_t v) { return K == 2 ? v & 0x7fffu : (v >> 8) & 0x7fu; }; // (two to a group: the entry IS the group's number; three: index | index << 4 | pair's sum << 8)
	auto lo = [](uint32_t v) { return v & 15u; };
	auto hi = [](uint32_t v) { return (v >> 4) & 15u; };
	uint64_t acc = 0;
	if (K == 2) {
#pragma unroll
		for (int j = 0; j < CNTG; ++j) acc |= (uint64_t)sum(e[j]) << (W * j);
	} else {
		const uint32_t R2 = R * R;
#pragma unroll
		for (int m = 0; m < CNTG / 2; ++m) {
			const uint32_t a = sum(e[3 * m]) + R2 * lo(e[3 * m + 1]), b = hi(e[3 * m + 1]) + R * sum(e[3 * m + 2]);
			acc |= ((uint64_t)a << (W * 2 * m)) | ((uint64_t)b << (W * (2 * m + 1)));
		}
	}
	const int ng = (rem + K - 1) / K; // groups of the stream from g0 on
	if (ng < CNTG) acc &= (1ull << (W * ng)) - 1ull;
	miss |= qmiss ? 0u : bad;
	return qmiss ? 0u : (uint32_t)(acc >> off); // no qualities: the row prints "*", the stream stays zero
}

// Three qualities a group of seven bits (five-value alphabets: every bench number), round 5: the group's number out of ONE look-up.  The three phred bytes are
// hashed (a 24-bit multiply, the host found the multiplier) into a table of 4096 entries in which the alphabet's 125 triples have slots of their own; an entry
// holds its triple beside the number, so a value outside the alphabet shows as a mismatch (-> miss, like the pair table's 0x8000).  Against qual_dword3g's nine
// pair look-ups and the arithmetic that puts pairs together into threes: SQ_INSTS_VALU 390 -> 366 M a launch, LDS bank conflicts 56.7 -> 14.9 M cycles, the kernel
// 655-660 -> 637-649 us (same box, same run) - most of its instructions are not the qualities' (addresses, descriptors, the pipeline's register copies).
constexpr int TRI_BITS = 12;
__device__ __forceinline__ uint32_t qual_dword3h(const Pack3Args &q, const uint32_t *s_tri, const PackDescR &d0, int64_t c, int t, const uint32_t (&raw)[Q3<7, 3>::NSRC + 1], uint32_t &miss,
                                                 uint32_t fill, uint32_t mul)
{
	constexpr int W = 7, K = 3, CNTG = Q3<W, K>::CNTG, NSRC = Q3<W, K>::NSRC;
	static_assert(CNTG == 6 && NSRC == 5, "six groups = eighteen qualities = five source dwords");
	const int n = d0.ll + d0.lr;
	const int g0 = (32 * t) / W, off = 32 * t - W * g0, i0 = K * g0;
	const uint32_t sh = (uint32_t)((d0.src + (uint64_t)((d0.lq + 1) / 2 + d0.begin + i0)) & 3ull);
	const int rem = n - i0;                                // qualities of the stream from i0 on (>= 1)
	uint32_t x[NSRC];
	bool qmiss = false;
#pragma unroll
	for (int g = 0; g < NSRC; ++g) {
		const uint32_t src = __builtin_amdgcn_alignbyte(raw[g + 1], raw[g], sh);
		if (g == 0) qmiss = (src & 0xffu) == 0xffu; // a read without qualities has 0xff in all of them
		const uint32_t valid = rem >= 4 * g + 4 ? 0xffffffffu : rem > 4 * g ? (1u << (8 * (rem - 4 * g))) - 1u : 0u;
		x[g] = (src & valid) | (fill & ~valid);
	}
	if (t == 0) q.flags[c] = qmiss ? 1 : 0;
	uint64_t acc = 0;
	uint32_t bad = 0;
#pragma unroll
	for (int j = 0; j < CNTG; ++j) {
		const int k = (3 * j) >> 2, s = (3 * j) & 3; // (compile-time after unrolling; 3 j + 2 <= 17 < 20: x[k + 1] exists whenever s > 1)
		const uint32_t tri = (s == 0 ? x[k] : __builtin_amdgcn_alignbyte(k + 1 < NSRC ? x[k + 1] : 0u, x[k], (uint32_t)s)) & 0xffffffu;
		const uint32_t e = s_tri[(uint32_t)__umul24(tri, mul) >> (32 - TRI_BITS)]; // (__umul24 returns an int)
		bad |= (e >> 8) ^ tri;
		acc |= (uint64_t)(e & 0x7fu) << (W * j);
	}
	const int ng = (rem + K - 1) / K; // groups of the stream from g0 on
	if (ng < CNTG) acc &= (1ull << (W * ng)) - 1ull;
	miss |= qmiss ? 0u : bad;
	return qmiss ? 0u : (uint32_t)(acc >> off); // no qualities: the row prints "*", the stream stays zero
}

// Persistent grid; every group of LPC lanes walks its clusters with a two-deep software pipeline: while the dwords of cluster i are composed
// and stored, the source dwords of cluster i + 1 and the descriptor of cluster i + 2 are on their way (a cluster is descriptor -> source
// bytes -> output: two dependent trips to memory that a wavefront would otherwise sit out; at eight wavefronts per SIMD that wait,
// not the instructions, is what the unpipelined form spends its time on).
template <int W, int BB, int K = 1>
__global__ __launch_bounds__(BLOCK) void k_pack3_direct(PackArgs p, Pack3Args q, const PackDesc *__restrict__ desc, const unsigned int *__restrict__ n_clusters_dev, uint8_t *__restrict__ out_str,
                                                        const uint16_t *__restrict__ pair_lut, int LPC)
{
	constexpr bool TRI = W == 7 && K == 3; // (qual_dword3h: the table is 4096 dwords of triples instead of 4096 halves of pairs)
	__shared__ __attribute__((aligned(16))) uint32_t s_tab[TRI ? (1 << TRI_BITS) : 2048];
	const uint16_t *s_pair = reinterpret_cast<const uint16_t *>(s_tab);
	__shared__ uint8_t s_base[256];
	if (W != 8) {
		const uint4 *g = reinterpret_cast<const uint4 *>(pair_lut);
		uint4 *l = reinterpret_cast<uint4 *>(s_tab);
#pragma unroll
		for (int i = 0; i < (TRI ? 4 : 2); ++i) l[threadIdx.x + i * BLOCK] = g[threadIdx.x + i * BLOCK]; // 8 KB = 512 x 16 B (16 KB: 1024), BLOCK == 256
	}
	if (BB == 2) { // BAM byte -> code(high nibble) | code(low nibble) << 2, bit 7: not both of A, C, G, T
		const uint32_t hi = threadIdx.x >> 4, lo = threadIdx.x & 15u;
		const bool ok = __popc(hi) == 1 && __popc(lo) == 1;
		s_base[threadIdx.x] = ok ? (uint8_t)((uint32_t)(__ffs((int)hi) - 1) | ((uint32_t)(__ffs((int)lo) - 1) << 2)) : (uint8_t)0x80;
	}
	__syncthreads();
	// LPC lanes per cluster (the host picks it from the dwords of the pass's longest read: 12 for 150 bases with grouped qualities - five clusters per wavefront
	// and round instead of the four that 16 lanes each give: the kernel is bound by its vector instructions, and a round costs the same however many lanes work)
	const int lane = lane_id(), per = WAVE / LPC;
	const int grp = lane / LPC, gl = lane - grp * LPC;
	const bool on = grp < per;
	const int64_t nc = (int64_t)*n_clusters_dev;
	const int64_t wave = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave_id();
	const int64_t step = (int64_t)gridDim.x * WAVES_PER_BLOCK * per;
	int64_t c = on ? wave * per + grp : (int64_t)1 << 40; // (lanes without a cluster: beyond every table, so that their descriptors come back empty)
	int64_t first = wave * per;                          // the first cluster of the wavefront's round
	uint32_t miss = 0;
	auto counts = [](const PackDescR &d, int &nDb, int &nDq) {
		const int n = d.ll + d.lr;
		const bool fast = pack_desc_fast(d);
		nDb = fast ? (n * BB + 31) / 32 : 0; nDq = fast ? Q3<W, K>::dwords(n) : 0;
	};
	auto issue = [&](const PackDescR &d, Src3<W, BB, K> &S) {
		int nDb, nDq; counts(d, nDb, nDq);
		S.w[0] = S.w[1] = S.w[2] = 0u;
#pragma unroll
		for (int g = 0; g <= Q3<W, K>::NSRC; ++g) S.raw[g] = 0u;
		if (gl < nDb) src3_load_bases<W, BB>(d, gl, S.w);
		if (gl < nDq) src3_load_quals<W, BB, K>(d, gl, S.raw);
	};
	PackDescR d0 = pack_desc_load(desc, c, nc), d1 = pack_desc_load(desc, c + step, nc);
	Src3<W, BB, K> S0, S1;
	const uint32_t qfill = p.qual_fill, qradix = (uint32_t)p.qual_radix;
	auto qual_dword = [&](const PackDescR &d, int64_t cc, int t, const uint32_t (&raw)[Q3<W, K>::NSRC + 1]) {
		if constexpr (TRI) return qual_dword3h(q, s_tab, d, cc, t, raw, miss, qfill, p.tri_mul);
		else if constexpr (K > 1) return qual_dword3g<W, K>(q, s_pair, d, cc, t, raw, miss, qfill, qradix);
		else return qual_dword3<W>(q, s_pair, d, cc, t, raw, miss);
	};
	issue(d0, S0);
	for (; first < nc; first += step, c += step) {
		const PackDescR d2 = pack_desc_load(desc, c + 2 * step, nc);
		issue(d1, S1);
		int nDb, nDq; counts(d0, nDb, nDq);
		uint32_t *d = reinterpret_cast<uint32_t *>(out_str + d0.str_off);
		if (gl < nDb) d[gl] = base_dword3<BB>(q, s_base, d0, c, gl, S0.w);
		if (gl < nDq) d[nDb + gl] = qual_dword(d0, c, gl, S0.raw);
		// clipped sequences longer than one round of the group (2-bit bases: 256; 3-bit qualities: 170): the rest, unpipelined
		for (int t = gl + LPC; t < nDb; t += LPC) { uint32_t w[3]; src3_load_bases<W, BB>(d0, t, w); d[t] = base_dword3<BB>(q, s_base, d0, c, t, w); }
		for (int t = gl + LPC; t < nDq; t += LPC) { uint32_t raw[Q3<W, K>::NSRC + 1]; src3_load_quals<W, BB, K>(d0, t, raw); d[nDb + t] = qual_dword(d0, c, t, raw); }
		d0 = d1; d1 = d2; S0 = S1;
	}
	if (W != 8 && miss) *p.lut_miss = 1;
}

// The base-by-base path: the slots of multi-event bins (mlist; those without a cluster leave at once: consensus storage, left part kept
// reversed), then the listed single-event clusters of reads longer than PACK_MAX_LQ.
template <int W, int BB, bool TRACK>
__global__ __launch_bounds__(BLOCK) void k_pack3_slow(PackArgs p, Pack3Args q, uint8_t *__restrict__ out_str)
{
	__shared__ uint8_t s_lut[256];  // phred -> alphabet index
	__shared__ uint8_t s_code[256]; // character -> 4-bit code
	__shared__ uint32_t s_seen[8];  // phred values met (bit set)
	if (W < 8) s_lut[threadIdx.x] = p.qlut[threadIdx.x]; // BLOCK == 256
	s_code[threadIdx.x] = NT16_CODE_OF[threadIdx.x];
	if (threadIdx.x < 8) s_seen[threadIdx.x] = 0;
	__syncthreads();
	// A wavefront looks at 64 items, one per lane - five of six slots of multi-event bins hold no cluster -, and its four groups of lanes then work off the ones
	// that do, four at a time: with a group per ITEM the launch was six times as many workgroups, most of which came and went (0.12 ms for a step's 28 K clusters).
	const int lane = lane_id(), grp = lane / GROUP, gl = lane % GROUP;
	const int64_t n_items = p.c.M + (int64_t)*p.slow_count;
	const int64_t k_ = ((int64_t)blockIdx.x * WAVES_PER_BLOCK + wave_id()) * WAVE + lane;
	uint32_t mine = 0xffffffffu;
	if (k_ < n_items) {
		mine = k_ < p.c.M ? p.c.mlist[k_] : p.slow_list[k_ - p.c.M];
		const int sup = p.c.support[mine];
		if (sup <= 0 || (k_ < p.c.M && sup == 1)) mine = 0xffffffffu; // (a lone cluster of a multi-event bin is a single: the dword path's, or - a long read - listed in slow_list)
	}
	for (uint64_t todo = __ballot(mine != 0xffffffffu); todo;) {
	int pick = -1;
	for (int g = 0; g < GROUPS_PER_WAVE && todo; ++g) { if (g == grp) pick = __ffsll((long long)todo) - 1; todo &= todo - 1ull; }
	const int64_t j = (int64_t)(uint32_t)__shfl((int)mine, pick < 0 ? 0 : pick, WAVE);
	if (pick < 0) continue;
	const SlotCluster sc = slot_cluster_load(p, j);
	const int ll = sc.ll, lr = sc.lr, lq = sc.lq, begin = sc.begin, n = ll + lr;
	const bool single = lq >= 0;
	uint32_t *d = reinterpret_cast<uint32_t *>(out_str + sc.str_off);
	gptr<uint8_t> sp = nullptr, qp = nullptr;
	const uint8_t *cs = nullptr, *cq = nullptr, *rs = nullptr, *rq = nullptr;
	bool qm;
	if (single) {
		sp = global_at<uint8_t>(sc.src); qp = sp + (lq + 1) / 2;
		qm = lq > 0 && qp[0] == 0xff;
		if (gl == 0) q.flags[sc.c] = qm ? 1 : 0;
	} else {
		const int64_t stride = 2ll * (p.c.SL + p.c.SR);
		cs = p.c.strings + (int64_t)p.c.mslot[j] * stride;
		cq = cs + p.c.SL; rs = cs + 2 * p.c.SL; rq = rs + p.c.SR;
		qm = sc.qmiss_multi != 0;
	}
	auto code_at = [&](int i) -> uint32_t { // BAM code of base i of seq_left + seq_right
		if (single) { const int pn = begin + i; return (uint32_t)(sp[pn >> 1] >> ((~pn & 1) << 2)) & 15u; }
		return s_code[i < ll ? cs[ll - 1 - i] : rs[i - ll]];
	};
	auto phred_at = [&](int i) -> uint32_t {
		if (single) return qp[begin + i];
		return ((uint32_t)(i < ll ? cq[ll - 1 - i] : rq[i - ll]) - 33u) & 255u;
	};
	constexpr int PER = 32 / BB;
	const int nDb = (n * BB + 31) / 32, nDq = (int)((qual_stream_bits((uint64_t)n, W ? (uint64_t)W : (uint64_t)p.qual_bits, W ? 1ull : (uint64_t)p.qual_group) + 31) / 32);
	// (all of a dword's byte loads first, then the look-ups: a loop that stops at the stream's end makes every load wait for the one before - the kernel's time was
	// the latency of sixteen loads in a row, 0.12 ms for the 28 K clusters of a step's multi-event bins)
	for (int t = gl; t < nDb; t += GROUP) {
		uint32_t word = 0, codes[PER];
#pragma unroll
		for (int k = 0; k < PER; ++k) codes[k] = PER * t + k < n ? code_at(PER * t + k) : 1u; // (behind the end: 'A', cut off below)
#pragma unroll
		for (int k = 0; k < PER; ++k) {
			const uint32_t code = codes[k];
			if (PER * t + k >= n) continue;
			if (BB == 4) word |= code << (4 * k);
			else if (code == 1u || code == 2u || code == 4u || code == 8u) word |= (uint32_t)(__ffs((int)code) - 1) << (2 * k);
			else exc_append(q, (uint64_t)sc.c, PER * t + k, code);
		}
		d[t] = word;
	}
	for (int t = gl; t < nDq; t += GROUP) {
		uint32_t word = 0;
		if constexpr (W == 0) { // grouped qualities (shape at run time)
			if (!qm) { // the (up to 18) qualities whose groups touch the dword: all loads first; a group's number is a sum, so its terms go into the stream one by one
				const int B = p.qual_bits, K = p.qual_group; // K is 2 or 3 (set_alphabet, seeksv_hip.hip)
				const uint32_t R = (uint32_t)p.qual_radix;
				const int g0 = (32 * t) / B, off = 32 * t - B * g0, i0 = K * g0;
				uint32_t ph[18];
#pragma unroll
				for (int jq = 0; jq < 18; ++jq) ph[jq] = i0 + jq < n && (K == 3 || jq < 12) ? phred_at(i0 + jq) : 0xffffffffu;
				uint64_t acc = 0;
				uint32_t miss = 0;
#pragma unroll
				for (int jq = 0; jq < 18; ++jq) {
					if (ph[jq] == 0xffffffffu) continue;
					const int g = K == 3 ? jq / 3 : jq / 2, j = K == 3 ? jq % 3 : jq % 2;
					if (TRACK) { const uint32_t bit = 1u << (ph[jq] & 31); if (!(atomicOr(&s_seen[ph[jq] >> 5], bit) & bit)) atomicOr(&p.qual_seen[ph[jq] >> 5], bit); }
					const uint32_t idx = s_lut[ph[jq]];
					miss |= idx;
					acc += (uint64_t)((idx & 0x7fu) * (j == 0 ? 1u : j == 1 ? R : R * R)) << (B * g);
				}
				if (miss & 0x80u) *p.lut_miss = 1;
				word = (uint32_t)(acc >> off);
			}
		} else if (W == 8) {
			for (int k = 0; k < 4 && 4 * t + k < n; ++k) word |= (qm ? 0x2au : phred_at(4 * t + k) + 33u) << (8 * k);
		} else if (!qm) {
			const int i0 = (32 * t) / W, off = 32 * t - W * i0;
			constexpr int CNT = W == 3 ? 12 : 32 / W; // qualities that can touch one dword
			uint32_t phs[CNT];
#pragma unroll
			for (int jq = 0; jq < CNT; ++jq) phs[jq] = i0 + jq < n && W * jq < off + 32 ? phred_at(i0 + jq) : 0xffffffffu;
			uint64_t acc = 0;
#pragma unroll
			for (int jq = 0; jq < CNT; ++jq) {
				const uint32_t ph = phs[jq];
				if (ph == 0xffffffffu) continue;
				if (TRACK) {
					const uint32_t bit = 1u << (ph & 31);
					if (!(atomicOr(&s_seen[ph >> 5], bit) & bit)) atomicOr(&p.qual_seen[ph >> 5], bit); // the first time this workgroup meets the value
				}
				const uint32_t idx = s_lut[ph];
				if (idx & 0x80u) *p.lut_miss = 1;
				acc |= (uint64_t)(idx & ((1u << W) - 1u)) << (W * jq);
			}
			word = (uint32_t)(acc >> off);
		}
		d[nDb + t] = word;
	}
	} // the wavefront's clusters, four at a time
}

} // namespace ssv
