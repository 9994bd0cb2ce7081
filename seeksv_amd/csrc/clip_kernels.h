// clip_kernels.h - getclip on the GPU: CIGAR-end scan -> ordered clip events -> (contig, side, pos) bins
// -> greedy consensus clustering, one wavefront per bin -> cluster table cut straight out of the reads' own bytes.
//
// Reference behaviour being reproduced (file:line in /root/reference/seeksv):
//   record routing + contig-switch rule   clip_reads.h:410-440
//   GetSClipReads / GenerateCigar         clip_reads.cpp:112-192, 309-329
//   GetSeq                                clip_reads.cpp:286-306
//   InsertSeq / CompareString* / ChangeSeqAndQual   clip_reads.cpp:260-283, 194-217, 57-108
//
// Data layout: hot / cold split.  The streaming pass reads one hot column (n_cigar, 2 B/record); everything a
// candidate record needs afterwards is ONE 64-byte line (ssv_record); a clip event is ONE 64-byte line (ClipEvent) that points at
// the read's packed bases + qualities where they already lie in HBM.
#pragma once

#include "common.h"
#include "seeksv_hip.h"

namespace ssv {

static_assert(sizeof(ssv_record) == 64, "ssv_record must be one 64-byte line");

// device view of one batch (all pointers in HBM)
struct DevBatch {
	int64_t n;
	const int32_t *tid, *pos;   // hot columns (streamed)
	const uint16_t *n_cigar;    // hot column (streamed)
	const ssv_record *rec;      // cold fields, one line per record
	const uint32_t *cigar;      // all operations (a line holds the first five)
	const uint8_t *seqqual;
	int32_t max_ref_span;
};

// structure-of-arrays source of k_build_rec (batches that come without `rec`)
struct SoaCols {
	const int32_t *tid, *pos;
	const uint16_t *flag;
	const uint8_t *mapq;
	const uint16_t *n_cigar;
	const int32_t *l_qseq, *mtid, *mpos, *isize;
	const uint32_t *cigar_off, *cigar;
	const uint8_t *xc;
	const uint64_t *seq_off;
};

// one ssv_record in registers: four 16-byte loads of one line
struct RecLine {
	uint32_t w_tid, w_pos, w_fmx, w_nc, w_lq, w_mtid, w_mpos, w_isize, w_coff, h0, h1, h2, h3, h4, so_lo, so_hi;
	__device__ __forceinline__ int tid() const { return (int)w_tid; }
	__device__ __forceinline__ int pos() const { return (int)w_pos; }
	__device__ __forceinline__ int flag() const { return (int)(w_fmx & 0xffffu); }
	__device__ __forceinline__ int mapq() const { return (int)((w_fmx >> 16) & 0xffu); }
	__device__ __forceinline__ int xc() const { return (int)(w_fmx >> 24); }
	__device__ __forceinline__ int n_cigar() const { return (int)(w_nc & 0xffffu); }
	__device__ __forceinline__ int l_qseq() const { return (int)w_lq; }
	__device__ __forceinline__ int mtid() const { return (int)w_mtid; }
	__device__ __forceinline__ int mpos() const { return (int)w_mpos; }
	__device__ __forceinline__ int isize() const { return (int)w_isize; }
	__device__ __forceinline__ uint32_t cigar_off() const { return w_coff; }
	__device__ __forceinline__ uint64_t seq_off() const { return (uint64_t)so_lo | ((uint64_t)so_hi << 32); }
	// operation k < 5 from the line: 64-bit shifts pick one of a pair (a chain of selects over values that were loaded side by side is turned
	// into an indexed access to a stack copy by hipcc: 32 bytes of scratch per lane and a dependent scratch load)
	__device__ __forceinline__ uint32_t head(int k) const
	{
		const uint64_t p01 = (uint64_t)h0 | ((uint64_t)h1 << 32), p23 = (uint64_t)h2 | ((uint64_t)h3 << 32);
		const uint64_t pr = k < 2 ? p01 : (k < 4 ? p23 : (uint64_t)h4);
		return (uint32_t)(pr >> ((k & 1) << 5));
	}
	// any operation: the line for k < 5, the batch's cigar array beyond
	__device__ __forceinline__ uint32_t op(const uint32_t *cigar, int k) const { return k < 5 ? head(k) : cigar[cigar_off() + (uint32_t)k]; }
};

__device__ __forceinline__ RecLine rec_load(const ssv_record *rec, int64_t i)
{
	const uint4 *p = reinterpret_cast<const uint4 *>(rec + i);
	const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
	RecLine r;
	r.w_tid = a.x; r.w_pos = a.y; r.w_fmx = a.z; r.w_nc = a.w; r.w_lq = b.x; r.w_mtid = b.y; r.w_mpos = b.z; r.w_isize = b.w;
	r.w_coff = c.x; r.h0 = c.y; r.h1 = c.z; r.h2 = c.w; r.h3 = d.x; r.h4 = d.y; r.so_lo = d.z; r.so_hi = d.w;
	return r;
}

// structure of arrays -> one line per record (one thread per record; only batches whose producer did not write `rec` itself)
__global__ __launch_bounds__(BLOCK) void k_build_rec(SoaCols s, int64_t n, ssv_record *__restrict__ out)
{
	const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	const uint32_t nc = s.n_cigar[i], off = s.cigar_off[i];
	uint32_t h[5];
#pragma unroll
	for (uint32_t k = 0; k < 5; ++k) h[k] = k < nc ? s.cigar[off + k] : 0u;
	const uint64_t so = s.seq_off[i];
	uint4 *p = reinterpret_cast<uint4 *>(out + i);
	p[0] = make_uint4((uint32_t)s.tid[i], (uint32_t)s.pos[i], (uint32_t)s.flag[i] | ((uint32_t)s.mapq[i] << 16) | ((s.xc ? (uint32_t)(s.xc[i] != 0) : 0u) << 24), nc);
	p[1] = make_uint4((uint32_t)s.l_qseq[i], (uint32_t)s.mtid[i], (uint32_t)s.mpos[i], (uint32_t)s.isize[i]);
	p[2] = make_uint4(off, h[0], h[1], h[2]);
	p[3] = make_uint4(h[3], h[4], (uint32_t)so, (uint32_t)(so >> 32));
}

// one line per record -> structure of arrays (ssv_batch_to_host: tests and debugging)
__global__ __launch_bounds__(BLOCK) void k_unpack_rec(const ssv_record *__restrict__ rec, int64_t n, uint16_t *flag, uint8_t *mapq, int32_t *l_qseq, int32_t *mtid, int32_t *mpos, int32_t *isize,
                                                     uint32_t *cigar_off, uint8_t *xc, uint64_t *seq_off)
{
	const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	const RecLine r = rec_load(rec, i);
	flag[i] = (uint16_t)r.flag(); mapq[i] = (uint8_t)r.mapq(); l_qseq[i] = r.l_qseq(); mtid[i] = r.mtid(); mpos[i] = r.mpos(); isize[i] = r.isize();
	cigar_off[i] = r.cigar_off(); xc[i] = (uint8_t)r.xc(); seq_off[i] = r.seq_off();
}

constexpr int CS_ITEMS = 4;                           // getsv scan: records per lane per sub-tile (two 16-byte loads per lane)
constexpr int CS_SUB = 4;                             // sub-tiles per tile
constexpr int CS_TILE = BLOCK * CS_ITEMS * CS_SUB;    // 4096 records per workgroup iteration, one barrier each
constexpr int CC_ITEMS = 8;                           // clip scan: 8 x u16 = one 16-byte load per lane per sub-tile
constexpr int CC_TILE = BLOCK * CC_ITEMS * CS_SUB;    // 8192 records per workgroup iteration
constexpr int PACK_MAX_LQ = 320;                     // reads up to this length take the LDS-staged paths of the pack kernels
constexpr int CS_MAX_BLOCKS = 8192;                   // upper bound of the persistent grid (private staging regions are sized by the actual grid)

// One clip event = one 64-byte line.  Written once by the filter kernel (a wavefront's events side by side: whole lines), moved once
// into BAM order, then only ever fetched whole by the per-event kernels.
struct ClipEvent {
	uint64_t key;      // tid << 33 | side << 32 | pos1   (side 0 = '5' / breakpoint2read_l, 1 = '3' / breakpoint2read_r)
	uint64_t src;      // device address of the read's packed bases (ceil(lq / 2) bytes) followed by its lq quality bytes
	uint64_t cig_ptr;  // device address of all the record's CIGAR operations (read only when ncig > 5)
	int32_t begin;     // first query base of seq_left  (GetSeq's begin_pos)
	int32_t ll, lr;    // |seq_left|, |seq_right|
	int32_t lq;        // l_qseq
	uint32_t ncig;
	uint32_t cig[5];   // operations 0..4
};
static_assert(sizeof(ClipEvent) == 64, "ClipEvent must be one 64-byte line");

__device__ __forceinline__ void event_store(ClipEvent *dst, const ClipEvent &e)
{
	uint4 *p = reinterpret_cast<uint4 *>(dst);
	p[0] = make_uint4((uint32_t)e.key, (uint32_t)(e.key >> 32), (uint32_t)e.src, (uint32_t)(e.src >> 32));
	p[1] = make_uint4((uint32_t)e.cig_ptr, (uint32_t)(e.cig_ptr >> 32), (uint32_t)e.begin, (uint32_t)e.ll);
	p[2] = make_uint4((uint32_t)e.lr, (uint32_t)e.lq, e.ncig, e.cig[0]);
	p[3] = make_uint4(e.cig[1], e.cig[2], e.cig[3], e.cig[4]);
}

__device__ __forceinline__ ClipEvent event_load(const ClipEvent *src)
{
	const uint4 *p = reinterpret_cast<const uint4 *>(src);
	const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
	ClipEvent e;
	e.key = (uint64_t)a.x | ((uint64_t)a.y << 32); e.src = (uint64_t)a.z | ((uint64_t)a.w << 32);
	e.cig_ptr = (uint64_t)b.x | ((uint64_t)b.y << 32); e.begin = (int32_t)b.z; e.ll = (int32_t)b.w;
	e.lr = (int32_t)c.x; e.lq = (int32_t)c.y; e.ncig = c.z; e.cig[0] = c.w;
	e.cig[1] = d.x; e.cig[2] = d.y; e.cig[3] = d.z; e.cig[4] = d.w;
	return e;
}

struct ClipCounters {
	unsigned long long n_cand;       // candidates of the batch (total of the tile-count scan)
	unsigned long long n_new;        // low 32 bits: events of the batch, high 32 bits: how many of them are right-clip ('3') events
	unsigned long long max_key;      // filled by k_event_max (grid-level reduction, a few hundred atomics)
	unsigned long long seq_total;    // running totals written by the offset scans of the copying (non-persistent) path
	unsigned long long cig_total;
	unsigned long long sum_ncig;     // CIGAR operations of the batch's events (bounds the table's CIGAR blob)
	unsigned long long n_long;       // events of reads longer than PACK_MAX_LQ (they take the bytewise path of the pack kernels)
	int max_ll, max_lr;
	int max_lq, max_ncig;
	int overflow;
	int l_unsorted;                  // the '5' events did not come out in key order (unsorted input): the full sort takes over
};

// K1 clip_scan arguments: the streaming pass only needs n_cigar
struct ClipScanArgs {
	const uint16_t *n_cigar;
	int64_t n;
	uint32_t *tile_cnt;      // [ntiles] candidates per tile
	uint32_t *tile_off;      // [ntiles] where the tile's candidates sit in stage[]
	uint32_t *stage;         // record indices; workgroup b owns stage[b * block_cap .. (b + 1) * block_cap)
	int64_t block_cap;
	int *overflow;
	int64_t ntiles;
};

// per-candidate filter arguments (GetSClipReads' predicate chain)
struct ClipFilterArgs {
	DevBatch b;
	int min_mapq;
	int save_low_quality;
	int use_ownership;       // range-partitioned runs: keep only events with own_lo <= (tid << 32 | pos1) < own_hi
	long long own_lo, own_hi;
	const int *last_tid_in;  // tid of the last mapped-pair record before this batch (clip_reads.h:407: starts at 0)
};

// GenerateCigar's l: M, D, =, N advance the reference; X does not (clip_reads.cpp:322)
__device__ __forceinline__ int ref_advance(uint32_t c)
{
	const int op = (int)(c & 15u);
	return (op == C_M || op == C_D || op == C_EQ || op == C_N) ? (int)(c >> 4) : 0;
}

// Decide the events of candidate record i from its line: at most one '5' (left-clip) and one '3' (right-clip) event, in that order.
// Returns bit 0: evl is an event, bit 1: evr is one.  Beyond the line it touches: the line of the record before it (the contig-switch
// rule; usually the same or the neighbouring line) and, for CIGARs of more than five operations, the cigar array.
__device__ __forceinline__ int clip_events_of(const ClipFilterArgs &a, int64_t i, const RecLine &r, ClipEvent &evl, ClipEvent &evr)
{
	const DevBatch &b = a.b;
	const int nc = r.n_cigar();
	if (nc < 2) return 0;                               // a lone "nS" is skipped like in the oracle
	const uint32_t c0 = r.head(0), cl = r.op(b.cigar, nc - 1);
	const int op1 = (int)(c0 & 15u), op2 = (int)(cl & 15u);
	if (op1 != C_S && op2 != C_S) return 0;             // clip_reads.cpp:124,150
	const int flag = r.flag();
	if (flag & (F_UNMAP | F_MUNMAP)) return 0;          // unmapped-pair side channel (host), clip_reads.h:415
	if (op1 == C_H || op2 == C_H || (flag & F_DUP) || r.mapq() < a.min_mapq) return 0; // clip_reads.cpp:118
	const int tid = r.tid();
	if (tid < 0) return 0;
	// contig-switch rule: processed only if tid equals the tid of the previous mapped-pair record (clip_reads.h:423-438)
	int prev_tid = *a.last_tid_in;
	for (int64_t j = i - 1; j >= 0; --j) {
		const uint4 pa = reinterpret_cast<const uint4 *>(b.rec + j)[0];
		if (!(pa.z & (uint32_t)(F_UNMAP | F_MUNMAP))) { prev_tid = (int)pa.x; break; }
	}
	if (tid != prev_tid) return 0;
	const int xc = r.xc();
	const int lq = r.l_qseq();
	const int pos0 = r.pos();
	int ref_len = 0;                                    // only right-clip events need it
	if (op2 == C_S) for (int k = 0; k < nc; ++k) ref_len += ref_advance(r.op(b.cigar, k));
	const bool s1 = op1 == C_S, s2 = op2 == C_S;
	const uint64_t tkey = (uint64_t)(uint32_t)tid << 33;
	int m = 0;
	if (s1 != s2) {
		if (xc != 0 && !a.save_low_quality) return 0;   // clip_reads.cpp:129
		if (s1) {
			const int ll = (int)(c0 >> 4), lr = lq - ll;
			if (lr < 0) return 0;
			evl.key = tkey | (uint32_t)(pos0 + 1); evl.begin = 0; evl.ll = ll; evl.lr = lr; m = 1;
		} else {
			const int lr = (int)(cl >> 4), ll = lq - lr;
			if (ll < 0) return 0;
			evr.key = tkey | (1ull << 32) | (uint32_t)(pos0 + ref_len); evr.begin = 0; evr.ll = ll; evr.lr = lr; m = 2;
		}
	} else {
		const int ll = (int)(c0 >> 4), rc = (int)(cl >> 4), mid = lq - ll - rc;
		if (mid < 0) return 0;
		bool do_l = true, do_r = true;
		if (xc != 0 && !a.save_low_quality) { if (!(flag & F_REV)) do_r = false; else do_l = false; } // clip_reads.cpp:160-175
		if (do_l) { evl.key = tkey | (uint32_t)(pos0 + 1); evl.begin = 0; evl.ll = ll; evl.lr = mid; m |= 1; }
		if (do_r) { evr.key = tkey | (1ull << 32) | (uint32_t)(pos0 + ref_len); evr.begin = ll; evr.ll = mid; evr.lr = rc; m |= 2; }
	}
	if (a.use_ownership) {
		const long long kl = ((long long)tid << 32) | (long long)(uint32_t)evl.key, kr = ((long long)tid << 32) | (long long)(uint32_t)evr.key;
		if ((m & 1) && !(kl >= a.own_lo && kl < a.own_hi)) m &= ~1;
		if ((m & 2) && !(kr >= a.own_lo && kr < a.own_hi)) m &= ~2;
	}
	const uint64_t src = (uint64_t)reinterpret_cast<uintptr_t>(b.seqqual) + r.seq_off();
	const uint64_t cig_ptr = (uint64_t)reinterpret_cast<uintptr_t>(b.cigar + r.cigar_off());
	evl.src = evr.src = src; evl.cig_ptr = evr.cig_ptr = cig_ptr; evl.lq = evr.lq = lq; evl.ncig = evr.ncig = (uint32_t)nc;
#pragma unroll
	for (int q = 0; q < 5; ++q) evl.cig[q] = evr.cig[q] = r.head(q);
	return m;
}

// Shared tail of the two streaming passes: given each lane's candidate bits (bit sub * ITEMS + k) and its per-sub-tile counts
// packed as four 16-bit fields, give every candidate of the tile a slot in the workgroup's private staging region, in record order.
// One barrier per tile (double-buffered LDS); the cursor is workgroup-uniform state that every thread tracks from the block totals.
template <int ITEMS>
__device__ __forceinline__ void stage_tile_candidates(uint32_t mask, uint64_t packed, int64_t tile, int64_t first_rec_of_lane, uint64_t (&lds)[2][WAVES_PER_BLOCK], int parity,
                                                      uint32_t &cursor, int64_t region, int64_t block_cap, uint32_t *tile_cnt, uint32_t *tile_off, uint32_t *stage, int *overflow)
{
	uint64_t inc = wave_inclusive_sum(packed);
	if (lane_id() == 63) lds[parity][wave_id()] = inc;
	__syncthreads();
	uint64_t base = 0, tot = 0;
#pragma unroll
	for (int w = 0; w < WAVES_PER_BLOCK; ++w) {
		uint64_t x = lds[parity][w];
		if (w < wave_id()) base += x;
		tot += x;
	}
	const uint64_t ex = base + inc - packed;
	const uint32_t total = (uint32_t)(tot & 0xffff) + (uint32_t)((tot >> 16) & 0xffff) + (uint32_t)((tot >> 32) & 0xffff) + (uint32_t)(tot >> 48);
	const bool fits = (int64_t)cursor + total <= block_cap;
	if (threadIdx.x == 0) {
		tile_cnt[tile] = fits ? total : 0u;
		tile_off[tile] = (uint32_t)(region + cursor);
		if (!fits) *overflow = 1;
	}
	if (mask && fits) {
		// first slot of this lane inside each sub-tile
		uint32_t first[CS_SUB];
		uint32_t sub_base = 0;
#pragma unroll
		for (int sub = 0; sub < CS_SUB; ++sub) {
			first[sub] = cursor + sub_base + (uint32_t)((ex >> (16 * sub)) & 0xffff);
			sub_base += (uint32_t)((tot >> (16 * sub)) & 0xffff);
		}
		// one store per candidate: the loop runs max-popcount-in-the-wave times (1-3 at WGS rates) instead of one predicated
		// store instruction per record slot
		constexpr uint32_t SUBMASK = ITEMS == 32 ? 0xffffffffu : ((1u << ITEMS) - 1u);
		uint32_t m = mask;
		while (m) {
			const int b = __ffs((int)m) - 1;
			m &= m - 1;
			const int sub = b / ITEMS, k = b % ITEMS;
			const uint32_t below = mask & ((1u << b) - 1u) & (SUBMASK << (sub * ITEMS));
			uint32_t f = first[0];
#pragma unroll
			for (int q = 1; q < CS_SUB; ++q) f = sub == q ? first[q] : f;
			stage[region + f + (uint32_t)__popc(below)] = (uint32_t)(first_rec_of_lane + (int64_t)sub * (BLOCK * ITEMS) + k);
		}
	}
	if (fits) cursor += total;
}

// K1 clip_scan: the streaming pass.  A record can only carry a usable soft clip if its CIGAR has at least two operations (a lone
// "nS" is skipped like in the oracle), so the pass reads nothing but n_cigar - 2 B/record, one 16-byte load per lane per 8 records,
// four loads in flight per lane - and writes the indices of the records with n_cigar >= 2 (indels and clips: ~3 % of a WGS BAM).
// Their lines are looked at by k_clip_filter, one thread per candidate.  Persistent workgroups, private staging, no atomics.
__device__ __forceinline__ void clip_scan_load(const ClipScanArgs &a, int64_t tile, uint4 (&v)[CS_SUB])
{
	const int64_t t0 = tile * CC_TILE + (int64_t)threadIdx.x * CC_ITEMS;
	if ((tile + 1) * CC_TILE <= a.n) { // workgroup-uniform: the whole tile is in range, all four loads issue back to back
#pragma unroll
		for (int sub = 0; sub < CS_SUB; ++sub) v[sub] = stream_load_u4(a.n_cigar + t0 + (int64_t)sub * (BLOCK * CC_ITEMS));
	} else {
#pragma unroll
		for (int sub = 0; sub < CS_SUB; ++sub) {
			const int64_t i0 = t0 + (int64_t)sub * (BLOCK * CC_ITEMS);
			uint32_t h[CC_ITEMS];
#pragma unroll
			for (int k = 0; k < CC_ITEMS; ++k) h[k] = i0 + k < a.n ? a.n_cigar[i0 + k] : 0u;
			v[sub] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
		}
	}
}

__global__ __launch_bounds__(BLOCK) void k_clip_scan(ClipScanArgs a)
{
	__shared__ uint64_t lds[2][WAVES_PER_BLOCK];
	uint32_t cursor = 0;
	int parity = 0;
	const int64_t region = (int64_t)blockIdx.x * a.block_cap;
	uint4 v[CS_SUB], nxt[CS_SUB];
	if ((int64_t)blockIdx.x < a.ntiles) clip_scan_load(a, blockIdx.x, v);
	for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x, parity ^= 1) {
		const int64_t t0 = tile * CC_TILE + (int64_t)threadIdx.x * CC_ITEMS;
		// software pipeline: the next tile's loads are in flight while this tile is classified, scanned and staged
		const int64_t next = tile + gridDim.x;
		if (next < a.ntiles) clip_scan_load(a, next, nxt);
		uint32_t mask = 0;
		uint64_t packed = 0;
#pragma unroll
		for (int sub = 0; sub < CS_SUB; ++sub) {
			const uint32_t w4[4] = {v[sub].x, v[sub].y, v[sub].z, v[sub].w};
			uint32_t bits = 0;
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				bits |= ((w4[k] & 0xffffu) >= 2u ? 1u : 0u) << (2 * k);
				bits |= ((w4[k] >> 16) >= 2u ? 1u : 0u) << (2 * k + 1);
			}
			mask |= bits << (sub * CC_ITEMS);
			packed += (uint64_t)__popc(bits) << (16 * sub);
		}
		stage_tile_candidates<CC_ITEMS>(mask, packed, tile, t0, lds, parity, cursor, region, a.block_cap, a.tile_cnt, a.tile_off, a.stage, a.overflow);
#pragma unroll
		for (int sub = 0; sub < CS_SUB; ++sub) v[sub] = nxt[sub];
	}
}

// candidates of tile t: stage[tile_off[t] ..] -> cand[tile_base[t] ..]; one wavefront per tile
__global__ __launch_bounds__(BLOCK) void k_cand_place(const uint32_t *__restrict__ stage, const uint32_t *__restrict__ tile_cnt, const uint32_t *__restrict__ tile_off,
                                                      const uint32_t *__restrict__ tile_base, int64_t ntiles, uint32_t *__restrict__ cand)
{
	int64_t t = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave_id();
	if (t >= ntiles) return;
	const uint32_t n = tile_cnt[t], so = tile_off[t], db = tile_base[t];
	for (uint32_t k = lane_id(); k < n; k += WAVE) cand[db + k] = stage[so + k];
}

// K1b clip_filter: one thread per candidate record (n_cigar >= 2) fetches the record's line - one 64-byte sector holds the CIGAR ends
// and every field of GetSClipReads' predicate chain (flag, MAPQ, DUP, XC, hard clips, lengths) - and leaves its 0, 1 or 2 events in the
// candidate's two stash slots.  cnt[c] = events | right-clip events << 32.
__global__ __launch_bounds__(BLOCK) void k_clip_filter(ClipFilterArgs a, const uint32_t *__restrict__ cand, int64_t n_cand, ClipEvent *__restrict__ stash, uint64_t *__restrict__ cnt)
{
	const int64_t c = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	ClipEvent evl, evr;
	int m = 0;
	if (c < n_cand) {
		const int64_t i = cand[c];
		const RecLine r = rec_load(a.b.rec, i);
		m = clip_events_of(a, i, r, evl, evr);
		cnt[c] = (uint64_t)((m & 1) + (m >> 1)) | ((uint64_t)(m >> 1) << 32);
	}
	// the wavefront's events side by side (two slots per candidate are reserved, the wave fills its 128 from the front): a third of
	// the candidates emit, and an event is one whole 64-byte line
	const int n = (m & 1) + (m >> 1);
	const int ex = wave_inclusive_sum(n) - n;
	ClipEvent *dst = stash + 2 * (c - lane_id()) + ex;
	if (m & 1) event_store(dst, evl);
	if (m & 2) event_store(dst + (m & 1), evr);
}

// tid of the last mapped-pair record of the batch -> *last_tid (unchanged when there is none)
__global__ __launch_bounds__(BLOCK) void k_last_tid(DevBatch b, int *last_tid)
{
	__shared__ long long best;
	if (threadIdx.x == 0) best = -1;
	__syncthreads();
	for (int64_t hi = b.n; hi > 0; hi -= BLOCK) {
		int64_t i = hi - 1 - threadIdx.x;
		if (i >= 0 && !(b.rec[i].flag & (F_UNMAP | F_MUNMAP))) atomicMax(&best, (long long)i);
		__syncthreads();
		if (best >= 0) break;
	}
	if (threadIdx.x == 0 && best >= 0) *last_tid = b.tid[best];
}

// the pass's events: one array of lines in BAM order, and the sort keys of the two sides as separate compact lists (the '5' events
// come out of a coordinate-sorted BAM already in key order: only the '3' list needs sorting)
struct EventLists {
	ClipEvent *ev;           // [n_events] BAM order
	uint64_t *key_l, *key_r; // keys of the side-0 / side-1 events, BAM order
	uint32_t *val_l, *val_r; // their event indices
};

// candidate c's events -> final, BAM-ordered position ev_base + ev_off[c] + e; keys into the side lists
__global__ __launch_bounds__(BLOCK) void k_clip_place(const ClipEvent *__restrict__ stash, const uint64_t *__restrict__ cnt, const uint64_t *__restrict__ ev_off, int64_t n_cand,
                                                      EventLists L, int64_t ev_base, int64_t l_base, int64_t r_base)
{
	const int64_t c = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	const int n = c < n_cand ? (int)(uint32_t)cnt[c] : 0;
	const int ex = wave_inclusive_sum(n) - n; // same thread-to-candidate mapping as k_clip_filter: the wave's events are packed from its first slot
	const ClipEvent *src = stash + 2 * (c - lane_id()) + ex;
	if (n == 0) return;
	const uint64_t off = ev_off[c];
	const int64_t e0 = ev_base + (int64_t)(uint32_t)off;
	int64_t ir = r_base + (int64_t)(off >> 32), il = l_base + (int64_t)(uint32_t)off - (int64_t)(off >> 32);
	for (int k = 0; k < n; ++k) {
		const uint4 *p = reinterpret_cast<const uint4 *>(src + k);
		const uint4 a = p[0], b = p[1], cc = p[2], d = p[3];
		uint4 *q = reinterpret_cast<uint4 *>(L.ev + e0 + k);
		q[0] = a; q[1] = b; q[2] = cc; q[3] = d;
		const uint64_t key = (uint64_t)a.x | ((uint64_t)a.y << 32);
		if ((key >> 32) & 1ull) { L.key_r[ir] = key; L.val_r[ir] = (uint32_t)(e0 + k); ++ir; }
		else { L.key_l[il] = key; L.val_l[il] = (uint32_t)(e0 + k); ++il; }
	}
}

// largest key / slice lengths / CIGAR length of the batch's events and their CIGAR operations in total: grid-stride partials, one atomic per wave
__global__ __launch_bounds__(BLOCK) void k_event_max(const ClipEvent *__restrict__ ev, int64_t ev_base, ClipCounters *ctr)
{
	const int64_t n_new = (int64_t)(uint32_t)ctr->n_new; // written by the scan of the per-candidate counts
	unsigned long long mk = 0, sc = 0, nl = 0;
	int mll = 0, mlr = 0, mlq = 0, mnc = 0;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_new; i += (int64_t)gridDim.x * blockDim.x) {
		const uint4 *p = reinterpret_cast<const uint4 *>(ev + ev_base + i);
		const uint4 a = p[0], b = p[1], c = p[2];
		const unsigned long long k = (uint64_t)a.x | ((uint64_t)a.y << 32);
		mk = k > mk ? k : mk;
		mll = (int)b.w > mll ? (int)b.w : mll;
		mlr = (int)c.x > mlr ? (int)c.x : mlr;
		mlq = (int)c.y > mlq ? (int)c.y : mlq;
		mnc = (int)c.z > mnc ? (int)c.z : mnc;
		sc += c.z;
		nl += (int)c.y > PACK_MAX_LQ ? 1u : 0u;
	}
	mk = wave_max(mk); mll = wave_max(mll); mlr = wave_max(mlr); mlq = wave_max(mlq); mnc = wave_max(mnc); sc = wave_sum(sc); nl = wave_sum(nl);
	if (lane_id() == 0) {
		atomicMax(&ctr->max_key, mk); atomicMax(&ctr->max_ll, mll); atomicMax(&ctr->max_lr, mlr); atomicMax(&ctr->max_lq, mlq); atomicMax(&ctr->max_ncig, mnc);
		atomicAdd(&ctr->sum_ncig, sc);
		if (nl) atomicAdd(&ctr->n_long, nl);
	}
}

// ---- the copying path (batches without SSV_MEM_PERSISTENT): the bytes an event points at move into context memory ----

constexpr int GROUP = 16;                      // lanes that cooperate on one event / cluster in the gather and pack kernels
constexpr int GROUPS_PER_WAVE = WAVE / GROUP;  // 4 items in flight per wavefront: the per-item metadata loads overlap
constexpr int GROUPS_PER_BLOCK = BLOCK / GROUP;

// bytes to copy per new event: packed bases + qualities (padded to 4: entries of the context blob start 4-byte aligned), long CIGARs
__global__ __launch_bounds__(BLOCK) void k_gather_sizes(const ClipEvent *__restrict__ ev, int64_t ev_base, int64_t n_new, uint32_t *__restrict__ seq_bytes, uint32_t *__restrict__ cig_ops)
{
	const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n_new) return;
	const uint4 c = reinterpret_cast<const uint4 *>(ev + ev_base + i)[2];
	const uint32_t lq = c.y, nc = c.z;
	seq_bytes[i] = ((lq + 1) / 2 + lq + 3u) & ~3u;
	cig_ops[i] = nc > 5 ? nc : 0u;
}

// 16 lanes per event copy its packed bases, qualities and (if longer than five operations) CIGAR into the context blobs and point the
// event at the copies.  The source may start anywhere, so every lane assembles aligned output dwords from two aligned source dwords.
__global__ __launch_bounds__(BLOCK) void k_clip_gather(ClipEvent *__restrict__ ev, int64_t ev_base, int64_t n_new, const uint64_t *__restrict__ seq_off, const uint64_t *__restrict__ cig_off,
                                                       uint8_t *__restrict__ seq_blob, uint32_t *__restrict__ cig_blob)
{
	const int64_t w = (int64_t)blockIdx.x * GROUPS_PER_BLOCK + (threadIdx.x / GROUP);
	if (w >= n_new) return;
	const uint32_t gl = threadIdx.x % GROUP;
	ClipEvent *E = ev + ev_base + w;
	const uint4 *p = reinterpret_cast<const uint4 *>(E);
	const uint4 a = p[0], b = p[1], c = p[2];
	const uint8_t *src = reinterpret_cast<const uint8_t *>((uintptr_t)((uint64_t)a.z | ((uint64_t)a.w << 32)));
	const uint32_t *cs = reinterpret_cast<const uint32_t *>((uintptr_t)((uint64_t)b.x | ((uint64_t)b.y << 32)));
	const uint32_t lq = c.y, nc = c.z;
	const uint32_t nb = ((lq + 1) / 2 + lq + 3u) & ~3u;
	uint32_t *dst = reinterpret_cast<uint32_t *>(seq_blob + seq_off[w]);
	const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 3u);
	const uint32_t *s4 = reinterpret_cast<const uint32_t *>(src - mis);
	// every load of the event issued before the first store: a read of up to 320 bases is covered by the unrolled batch
	constexpr int BATCH = 8;
	uint32_t lo[BATCH], hi[BATCH];
#pragma unroll
	for (int u = 0; u < BATCH; ++u) {
		const uint32_t k = gl + GROUP * u;
		const bool in = k < nb / 4;
		lo[u] = in ? s4[k] : 0u;
		hi[u] = in && mis ? s4[k + 1] : 0u;       // may read up to 7 bytes past the entry: see the slack rule in seeksv_hip.h
	}
#pragma unroll
	for (int u = 0; u < BATCH; ++u) {
		const uint32_t k = gl + GROUP * u;
		if (k < nb / 4) dst[k] = mis ? __builtin_amdgcn_alignbyte(hi[u], lo[u], mis) : lo[u];
	}
	for (uint32_t k = gl + GROUP * BATCH; k < nb / 4; k += GROUP) { const uint32_t l = s4[k], h = mis ? s4[k + 1] : 0u; dst[k] = mis ? __builtin_amdgcn_alignbyte(h, l, mis) : l; }
	uint32_t *cd = cig_blob + cig_off[w];
	if (nc > 5) for (uint32_t k = gl; k < nc; k += GROUP) cd[k] = cs[k];
	if (gl == 0) {
		uint4 *q = reinterpret_cast<uint4 *>(E);
		const uint64_t ns = (uint64_t)reinterpret_cast<uintptr_t>(dst), ncp = (uint64_t)reinterpret_cast<uintptr_t>(cd);
		q[0] = make_uint4(a.x, a.y, (uint32_t)ns, (uint32_t)(ns >> 32));
		q[1] = make_uint4((uint32_t)ncp, (uint32_t)(ncp >> 32), b.z, b.w);
	}
}

// ---- binning: the '5' list is in key order already (checked), the '3' list is sorted, both are merged per contig ----

__global__ __launch_bounds__(BLOCK) void k_check_sorted(const uint64_t *__restrict__ key, int64_t n, int *__restrict__ flag)
{
	const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i + 1 < n && key[i] > key[i + 1]) *flag = 1;
}

// first element of each contig in the two sorted side lists: cum_l[t] = # side-0 keys of contigs < t, cum_r likewise; t in [0, T]
__global__ __launch_bounds__(BLOCK) void k_side_bounds(const uint64_t *__restrict__ key_l, int64_t nl, const uint64_t *__restrict__ key_r, int64_t nr, int64_t T,
                                                       uint32_t *__restrict__ cum_l, uint32_t *__restrict__ cum_r)
{
	const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (t > T) return;
	const uint64_t want = (uint64_t)t << 33;
	auto lower = [&](const uint64_t *k, int64_t n) { int64_t lo = 0, hi = n; while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (k[m] < want) lo = m + 1; else hi = m; } return lo; };
	cum_l[t] = t >= (1ll << 31) ? (uint32_t)nl : (uint32_t)lower(key_l, nl);
	cum_r[t] = t >= (1ll << 31) ? (uint32_t)nr : (uint32_t)lower(key_r, nr);
}

// (contig, side, position) order = per contig: its '5' events, then its '3' events
__global__ __launch_bounds__(BLOCK) void k_merge_sides(const uint64_t *__restrict__ key_l, const uint32_t *__restrict__ val_l, int64_t nl, const uint64_t *__restrict__ key_r,
                                                       const uint32_t *__restrict__ val_r, int64_t nr, const uint32_t *__restrict__ cum_l, const uint32_t *__restrict__ cum_r,
                                                       uint64_t *__restrict__ skey, uint32_t *__restrict__ perm)
{
	const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i < nl) {
		const uint64_t k = key_l[i];
		const int64_t o = i + cum_r[k >> 33];
		skey[o] = k; perm[o] = val_l[i];
	} else if (i < nl + nr) {
		const int64_t j = i - nl;
		const uint64_t k = key_r[j];
		const int64_t o = j + cum_l[(k >> 33) + 1];
		skey[o] = k; perm[o] = val_r[j];
	}
}

// unsorted input: both lists, '5' first, into one array for the full sort (equal keys are on one side, so BAM order inside a bin is kept)
__global__ __launch_bounds__(BLOCK) void k_concat_sides(const uint64_t *__restrict__ key_l, const uint32_t *__restrict__ val_l, int64_t nl, const uint64_t *__restrict__ key_r,
                                                        const uint32_t *__restrict__ val_r, int64_t nr, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
	const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i < nl) { keys[i] = key_l[i]; vals[i] = val_l[i]; }
	else if (i < nl + nr) { keys[i] = key_r[i - nl]; vals[i] = val_r[i - nl]; }
}

// ---------------------------------------------------------------------------------------------------------------------
// K3 cluster_bins
// ---------------------------------------------------------------------------------------------------------------------

__device__ __constant__ char NT16[16] = {'=', 'A', 'C', 'M', 'G', 'R', 'S', 'V', 'T', 'W', 'Y', 'H', 'K', 'D', 'B', 'N'}; // bam_nt16_rev_table

struct ClusterArgs {
	// sorted events
	const uint64_t *skey;  // [E] sorted keys
	const uint32_t *perm;  // [E] sorted position -> event index
	int64_t E;
	const ClipEvent *ev;
	double match_rate;
	// per sorted slot outputs
	int32_t *support;      // [E]; > 0 marks a cluster created by the event at this slot (single-event bins: 1, set by k_bin_mark)
	int32_t *c_ll, *c_lr;  // [E] clusters of multi-event bins only
	uint32_t *c_cig_ev;    // [E] event whose CIGAR the cluster carries (multi-event bins only)
	uint8_t *c_qmiss;      // [E] (multi-event bins only)
	const uint32_t *mflag; // [E] 1: the slot belongs to a bin with more than one event
	const uint32_t *mslot; // [E] exclusive scan of mflag: index of the slot's string storage
	const uint32_t *mlist; // [M] the slots with mflag set, ascending (inverse of mslot)
	int64_t M;
	uint8_t *strings;      // [M * stride]: left seq (reversed), left qual (reversed), right seq, right qual - multi-event bins only
	int32_t SL, SR;        // capacity of a left / right string
};

constexpr int CL_CACHE = 64; // clusters of the current bin tracked in LDS; deeper bins fall back to scanning the slots

struct EventView {
	const uint8_t *sp, *qp;
	int begin, ll, lr;
	bool qmiss;
	// i-th base of seq_left counted from its END (i = 0 is adjacent to the breakpoint side of the compare)
	__device__ __forceinline__ int lpos(int i) const { return begin + ll - 1 - i; }
	__device__ __forceinline__ int rpos(int i) const { return begin + ll + i; }
	__device__ __forceinline__ char base(int p) const { return NT16[(sp[p >> 1] >> ((~p & 1) << 2)) & 15]; }
	__device__ __forceinline__ char qual(int p) const { return qmiss ? '*' : (char)(qp[p] + 33); }
};

// Classify the sorted slots.  A bin with a single event (97 % of a WGS sample: random clips) needs no clustering: its cluster is
// the event itself; only bins with several events go through k_cluster_bins and get string storage.
__global__ void k_bin_mark(const uint64_t *__restrict__ skey, int64_t E, uint32_t *__restrict__ mflag, int32_t *__restrict__ support)
{
	int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= E) return;
	const uint64_t k = skey[j];
	const bool single = (j == 0 || skey[j - 1] != k) && (j + 1 == E || skey[j + 1] != k);
	mflag[j] = single ? 0u : 1u;
	support[j] = single ? 1 : 0;
}

// slots of multi-event bins, densely: mlist[mslot[j]] = j
__global__ void k_multi_list(const uint32_t *__restrict__ mflag, const uint32_t *__restrict__ mslot, int64_t E, uint32_t *__restrict__ mlist)
{
	int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j < E && mflag[j]) mlist[mslot[j]] = (uint32_t)j;
}

// One wavefront per bin (= run of equal keys in the sorted event list).  The wave walks the bin's events in BAM order -
// the order the reference's multimap::equal_range scan sees them - and keeps the evolving clusters in HBM; lanes are
// spread over bases, match counts come from ballots.  Bins are independent, so there is no cross-wave communication.
__global__ __launch_bounds__(BLOCK) void k_cluster_bins(ClusterArgs a)
{
	__shared__ int32_t s_slot[WAVES_PER_BLOCK][CL_CACHE];
	const int64_t m0 = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave_id();
	if (m0 >= a.M) return;
	const int64_t j0 = a.mlist[m0];                // slots of multi-event bins only (single-event bins were finished by k_bin_mark)
	const uint64_t key0 = a.skey[j0];
	if (j0 > 0 && a.skey[j0 - 1] == key0) return; // not the start of a bin
	const int lane = lane_id();
	const int w = wave_id();
	const bool left_clipped = ((key0 >> 32) & 1ull) == 0; // side '5' = breakpoint2read_l = LEFT_CLIPPED
	const int64_t stride = 2ll * (a.SL + a.SR);
	int nclu = 0;
	for (int64_t jj = j0; jj < a.E && a.skey[jj] == key0; ++jj) {
		const uint32_t e = a.perm[jj];
		const uint4 *ep = reinterpret_cast<const uint4 *>(a.ev + e); // one line, the same address in every lane
		const uint4 ea = ep[0], eb = ep[1], ec = ep[2];
		EventView v;
		const int lq = (int)ec.y;
		v.sp = reinterpret_cast<const uint8_t *>((uintptr_t)((uint64_t)ea.z | ((uint64_t)ea.w << 32)));
		v.qp = v.sp + (lq + 1) / 2;
		v.begin = (int)eb.z; v.ll = (int)eb.w; v.lr = (int)ec.x;
		v.qmiss = lq > 0 && v.qp[0] == 0xff; // no qualities: the row prints "*" (clip_reads.cpp:296)
		// ---- find the first cluster of the bin that absorbs this event (clip_reads.cpp:262-273) ----
		auto absorbs = [&](int64_t slot) -> bool {
			const uint8_t *cs = a.strings + (int64_t)a.mslot[slot] * stride;
			const int cll = a.c_ll[slot], clr = a.c_lr[slot];
			const int n1 = v.ll < cll ? v.ll : cll;
			int m1 = 0;
			for (int i0 = 0; i0 < n1; i0 += WAVE) {
				int i = i0 + lane;
				bool eq = i < n1 && v.base(v.lpos(i)) == (char)cs[i];
				m1 += (int)__popcll(__ballot(eq));
			}
			if (!((double)m1 / (double)n1 >= a.match_rate)) return false; // n1 == 0 -> NaN -> false, like the reference
			const int n2 = v.lr < clr ? v.lr : clr;
			int m2 = 0;
			const uint8_t *csr = cs + 2 * a.SL;
			for (int i0 = 0; i0 < n2; i0 += WAVE) {
				int i = i0 + lane;
				bool eq = i < n2 && v.base(v.rpos(i)) == (char)csr[i];
				m2 += (int)__popcll(__ballot(eq));
			}
			return (double)m2 / (double)n2 >= a.match_rate;
		};
		int64_t hit = -1;
		const int kmax = nclu < CL_CACHE ? nclu : CL_CACHE;
		for (int k = 0; k < kmax && hit < 0; ++k) {
			int64_t slot = j0 + s_slot[w][k];
			if (absorbs(slot)) hit = slot;
		}
		if (hit < 0 && nclu > CL_CACHE) {
			// clusters beyond the LDS cache: they were created after the last cached one, i.e. at later slots
			for (int64_t s = j0 + s_slot[w][CL_CACHE - 1] + 1; s < jj && hit < 0; ++s)
				if (a.support[s] > 0 && absorbs(s)) hit = s;
		}
		if (hit >= 0) {
			// ---- ReadsInfo::ChangeSeqAndQual (clip_reads.cpp:57-108) on the reversed-left / forward-right storage ----
			uint8_t *cs = a.strings + (int64_t)a.mslot[hit] * stride;
			uint8_t *cq = cs + a.SL, *rs = cs + 2 * a.SL, *rq = rs + a.SR;
			const int cll = a.c_ll[hit], clr = a.c_lr[hit];
			const int n1 = v.ll < cll ? v.ll : cll;
			for (int i = lane; i < v.ll; i += WAVE) {
				int p = v.lpos(i);
				char q = v.qual(p);
				if (i < n1) {
					if ((signed char)cq[i] < (signed char)q) { cq[i] = (uint8_t)q; cs[i] = (uint8_t)v.base(p); }
				} else if (cll <= v.ll) { cs[i] = (uint8_t)v.base(p); cq[i] = (uint8_t)q; } // prepend the extra prefix
			}
			if (cll <= v.ll) {
				a.c_ll[hit] = v.ll;
				if (!left_clipped) a.c_cig_ev[hit] = e;  // aa == RIGHT_CLIPPED (also when the lengths are equal)
			}
			const int n2 = v.lr < clr ? v.lr : clr;
			for (int i = lane; i < v.lr; i += WAVE) {
				int p = v.rpos(i);
				char q = v.qual(p);
				if (i < n2) {
					if ((signed char)rq[i] < (signed char)q) { rq[i] = (uint8_t)q; rs[i] = (uint8_t)v.base(p); }
				} else if (clr < v.lr) { rs[i] = (uint8_t)v.base(p); rq[i] = (uint8_t)q; } // append the extra suffix
			}
			if (clr < v.lr) {
				a.c_lr[hit] = v.lr;
				if (left_clipped) a.c_cig_ev[hit] = e;   // aa == LEFT_CLIPPED
			}
			a.support[hit] += 1; // every lane stores the same value; each lane later reads back what it stored
		} else {
			// ---- new cluster at this event's slot (clip_reads.cpp:276-281) ----
			uint8_t *cs = a.strings + (int64_t)a.mslot[jj] * stride;
			uint8_t *cq = cs + a.SL, *rs = cs + 2 * a.SL, *rq = rs + a.SR;
			for (int i = lane; i < v.ll; i += WAVE) { int p = v.lpos(i); cs[i] = (uint8_t)v.base(p); cq[i] = (uint8_t)v.qual(p); }
			for (int i = lane; i < v.lr; i += WAVE) { int p = v.rpos(i); rs[i] = (uint8_t)v.base(p); rq[i] = (uint8_t)v.qual(p); }
			a.c_ll[jj] = v.ll; a.c_lr[jj] = v.lr; a.c_cig_ev[jj] = e; a.c_qmiss[jj] = v.qmiss ? 1 : 0;
			a.support[jj] = 1;
			if (nclu < CL_CACHE) s_slot[w][nclu] = (int32_t)(jj - j0);
			++nclu;
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------------
// cluster table packing
// ---------------------------------------------------------------------------------------------------------------------

struct PackArgs {
	ClusterArgs c;
	// per sorted slot: what the scans turn into dense positions
	uint64_t *slot_cnt;       // [E] cluster at this slot ? 1 | CIGAR operations << 32 : 0          -> exclusive scan: cidx | cig_off << 32
	uint64_t *slot_bytes;     // [E] bytes of the cluster's string block (0 when there is none)      -> exclusive scan: str_off
	// dense outputs [n_clusters]
	int32_t *tid, *pos;
	uint8_t *side;
	int32_t *support, *ll, *lr;
	uint8_t *qmiss;
	int32_t *ncig;
	uint64_t *str_off, *cig_off;
	uint32_t *slot;           // dense index -> sorted slot
	int packed;               // 1: sequences as 4-bit codes (ssv_cluster_table.seq_packed)
	int qual_bits;            // 8: quality characters; 1, 2, 3, 4: indices into the table's quality alphabet
	const uint8_t *qlut;      // [256] phred -> index (0xff: not in the alphabet), when qual_bits < 8
	uint32_t *qual_seen;      // [8] bit set of the phred values met while packing (which alphabet the table really needs)
	// where a cluster's characters come from, resolved once per cluster by k_cluster_cols so that the pack kernel starts with
	// independent loads instead of a slot -> event -> address chain: src_lq >= 0: single-event cluster, the read's packed bases start at
	// address src_ptr, its seq_left at base src_begin; src_lq < 0: consensus storage of a multi-event bin
	uint64_t *src_ptr;
	int32_t *src_begin, *src_lq;
	uint32_t *cig_ev;         // the event whose CIGAR the cluster carries
	// clusters that need the bytewise path of the packed kernel (multi-event bins, reads longer than PACK_MAX_LQ), listed by k_cluster_cols:
	// they are packed by a launch of their own, so that a wavefront of the main launch never runs the long path for one of its four clusters
	uint32_t *slow_list;
	unsigned int *slow_count;
};

__host__ __device__ __forceinline__ uint64_t table_block_bytes(uint64_t L, uint64_t R, int packed, uint64_t W)
{
	return ((packed ? (L + 1) / 2 + (L * W + 7) / 8 + (R + 1) / 2 + (R * W + 7) / 8 : 2 * (L + R)) + 3ull) & ~3ull; // blocks start 4-byte aligned
}

// per sorted slot: is there a cluster, how many bytes / CIGAR operations does it put into the table
__global__ __launch_bounds__(BLOCK) void k_cluster_meta(PackArgs p)
{
	const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= p.c.E) return;
	uint64_t cnt = 0, bytes = 0;
	if (p.c.support[j] > 0) {
		const bool single = !p.c.mflag[j];
		const uint32_t e = single ? p.c.perm[j] : p.c.c_cig_ev[j];
		const uint4 *ep = reinterpret_cast<const uint4 *>(p.c.ev + e);
		const uint4 eb = ep[1], ec = ep[2];
		const int ll = single ? (int)eb.w : p.c.c_ll[j], lr = single ? (int)ec.x : p.c.c_lr[j];
		cnt = 1ull | ((uint64_t)ec.z << 32);
		bytes = table_block_bytes((uint64_t)ll, (uint64_t)lr, p.packed, (uint64_t)p.qual_bits);
	}
	p.slot_cnt[j] = cnt; p.slot_bytes[j] = bytes;
}

// per cluster slot: the dense columns (slot_cnt / slot_bytes now hold their exclusive scans)
__global__ __launch_bounds__(BLOCK) void k_cluster_cols(PackArgs p)
{
	const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= p.c.E || !(p.c.support[j] > 0)) return;
	const uint64_t sc = p.slot_cnt[j];
	const uint32_t c = (uint32_t)sc;
	const uint64_t key = p.c.skey[j];
	const bool single = !p.c.mflag[j];
	const uint32_t e = single ? p.c.perm[j] : p.c.c_cig_ev[j];
	const uint4 *ep = reinterpret_cast<const uint4 *>(p.c.ev + e);
	const uint4 ea = ep[0], eb = ep[1], ec = ep[2];
	const int ll = single ? (int)eb.w : p.c.c_ll[j], lr = single ? (int)ec.x : p.c.c_lr[j];
	const int lq = (int)ec.y;
	p.tid[c] = (int32_t)(key >> 33);
	p.pos[c] = (int32_t)(uint32_t)key;
	p.side[c] = ((key >> 32) & 1ull) ? '3' : '5';
	p.support[c] = p.c.support[j];
	p.ll[c] = ll; p.lr[c] = lr;
	p.ncig[c] = (int32_t)ec.z;
	p.str_off[c] = p.slot_bytes[j];
	p.cig_off[c] = sc >> 32;
	p.slot[c] = (uint32_t)j;
	p.cig_ev[c] = e;
	p.src_ptr[c] = single ? ((uint64_t)ea.z | ((uint64_t)ea.w << 32)) : 0ull;
	p.src_begin[c] = single ? (int32_t)eb.z : 0;
	p.src_lq[c] = single ? lq : -1;
	if (!single) p.qmiss[c] = p.c.c_qmiss[j]; // single-event clusters: the pack kernel sees the read's first quality byte and writes it
	if (!(single && lq <= PACK_MAX_LQ && ll + lr <= PACK_MAX_LQ)) p.slow_list[atomicAdd(p.slow_count, 1u)] = c; // a few thousand of millions: no hot spot
}

// which phred values occur among the qualities of the first events (BAM order): the first guess of the table's quality alphabet
__global__ __launch_bounds__(BLOCK) void k_qual_sample(const ClipEvent *__restrict__ ev, int64_t n, uint32_t *__restrict__ seen)
{
	__shared__ uint32_t s_seen[8];
	if (threadIdx.x < 8) s_seen[threadIdx.x] = 0;
	__syncthreads();
	const int64_t e = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave_id();
	if (e < n) {
		const uint4 *ep = reinterpret_cast<const uint4 *>(ev + e);
		const uint4 ea = ep[0], ec = ep[2];
		const int lq = (int)ec.y;
		const uint8_t *qp = reinterpret_cast<const uint8_t *>((uintptr_t)((uint64_t)ea.z | ((uint64_t)ea.w << 32))) + (lq + 1) / 2;
		if (lq > 0 && qp[0] != 0xff)
			for (int i = lane_id(); i < lq; i += WAVE) { const uint32_t q = qp[i]; atomicOr(&s_seen[q >> 5], 1u << (q & 31)); }
	}
	__syncthreads();
	if (threadIdx.x < 8 && s_seen[threadIdx.x]) atomicOr(&seen[threadIdx.x], s_seen[threadIdx.x]);
}

// bam_nt16_rev_table "=ACMGRSVTWYHKDBN" as two little-endian 64-bit words: nibble -> ASCII without touching memory
__device__ __forceinline__ uint32_t nt16_char(uint32_t nib)
{
	const uint64_t LO = ((uint64_t)'=') | ((uint64_t)'A' << 8) | ((uint64_t)'C' << 16) | ((uint64_t)'M' << 24) | ((uint64_t)'G' << 32) | ((uint64_t)'R' << 40) | ((uint64_t)'S' << 48) | ((uint64_t)'V' << 56);
	const uint64_t HI = ((uint64_t)'T') | ((uint64_t)'W' << 8) | ((uint64_t)'Y' << 16) | ((uint64_t)'H' << 24) | ((uint64_t)'K' << 32) | ((uint64_t)'D' << 40) | ((uint64_t)'B' << 48) | ((uint64_t)'N' << 56);
	return (uint32_t)(((nib & 8u) ? HI : LO) >> (8u * (nib & 7u))) & 0xffu;
}

// the inverse as a table (ASCII -> 4-bit code, anything else N): a compare chain costs 32 instructions per character, and a wavefront with
// one cluster of a multi-event bin runs its path for all its lanes
__device__ __constant__ uint8_t NT16_CODE_OF[256] = {15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 0, 15, 15, 15, 1, 14, 2, 13, 15, 15, 4, 11, 15, 15, 12, 15, 3, 15, 15, 15, 15, 5, 6, 8, 15, 7, 9, 15, 10, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15};

// the CIGAR the cluster carries: the event's line holds the first five operations, its cig_ptr all of them
__device__ __forceinline__ void pack_copy_cigar(const ClipEvent *ev, uint32_t e, int ncg, int gl, uint32_t *dc)
{
	const uint4 *ep = reinterpret_cast<const uint4 *>(ev + e);
	if (ncg <= 5) {
		const uint4 c = ep[2], d = ep[3];
		const uint32_t op = gl == 0 ? c.w : gl == 1 ? d.x : gl == 2 ? d.y : gl == 3 ? d.z : d.w;
		if (gl < ncg) dc[gl] = op;
	} else {
		const uint4 b = ep[1];
		const uint32_t *src = reinterpret_cast<const uint32_t *>((uintptr_t)((uint64_t)b.x | ((uint64_t)b.y << 32)));
		for (int i = gl; i < ncg; i += GROUP) dc[i] = src[i];
	}
}

// 16 lanes per cluster, four clusters per wavefront: strings and the CIGAR of the carrying event into dense blobs.  Every lane
// assembles whole output dwords (a cluster's block starts 4-byte aligned and is zero padded to a multiple of 4 bytes):
// [seq_left | qual_left | seq_right | qual_right].  Single-event clusters (97 %) are decoded from the read's packed bases /
// qualities where the batch (or the context's copy) holds them (GetSeq, clip_reads.cpp:286-306): the group expands the read once into
// LDS with dword loads (8 bases per packed dword, 4 qualities per dword via alignbyte and a packed +33) and then composes the output
// dwords from LDS bytes.  Clusters of multi-event bins come from their consensus storage (left part un-reversed); reads longer than
// PACK_MAX_LQ take the per-byte path.
// This is the ASCII layout (ssv_clip_table_format 0, the C ABI's default); the packed layouts have their own kernel below.
__global__ __launch_bounds__(BLOCK) void k_cluster_pack_ascii(PackArgs p, const unsigned int *__restrict__ n_clusters_dev, uint8_t *__restrict__ out_str, uint32_t *__restrict__ out_cig)
{
	const int64_t n_clusters = (int64_t)*n_clusters_dev; // the grid is an upper bound (one group per event)
	__shared__ uint32_t s_seq[GROUPS_PER_BLOCK][PACK_MAX_LQ / 4 + 4];
	__shared__ uint32_t s_qual[GROUPS_PER_BLOCK][PACK_MAX_LQ / 4 + 4];
	const int grp = (int)(threadIdx.x / GROUP);
	const int gl = (int)(threadIdx.x % GROUP);
	const int64_t c = (int64_t)blockIdx.x * GROUPS_PER_BLOCK + grp;
	const bool active = c < n_clusters;
	int64_t j = 0;
	int ll = 0, lr = 0, lq = -1, begin = 0;
	bool single = false, staged = false, qmiss = false;
	const uint8_t *sp = nullptr;
	if (active) {
		j = p.slot[c]; ll = p.ll[c]; lr = p.lr[c]; lq = p.src_lq[c];
		single = lq >= 0;
		if (single) {
			sp = reinterpret_cast<const uint8_t *>((uintptr_t)p.src_ptr[c]);
			begin = p.src_begin[c];
			qmiss = lq > 0 && sp[(lq + 1) / 2] == 0xff;
			staged = lq <= PACK_MAX_LQ;
		}
	}
	if (staged) {
		// the read's packed bases start at any byte: aligned dword loads + alignbyte
		const uint32_t smis = (uint32_t)(reinterpret_cast<uintptr_t>(sp) & 3u);
		const uint32_t *sp4 = reinterpret_cast<const uint32_t *>(sp - smis);
		const int nseq4 = ((lq + 1) / 2 + 3) / 4; // packed dwords holding the bases
		for (int w = gl; w < nseq4; w += GROUP) {
			const uint32_t l0 = sp4[w], h0 = smis ? sp4[w + 1] : 0u;
			const uint32_t pk = smis ? __builtin_amdgcn_alignbyte(h0, l0, smis) : l0;
			uint32_t lo = 0, hi = 0;
#pragma unroll
			for (int b = 0; b < 2; ++b) { // packed bytes 0,1 -> chars 0..3 ; bytes 2,3 -> chars 4..7
				const uint32_t b0 = (pk >> (16 * b)) & 0xffu, b1 = (pk >> (16 * b + 8)) & 0xffu;
				const uint32_t four = nt16_char(b0 >> 4) | (nt16_char(b0 & 15u) << 8) | (nt16_char(b1 >> 4) << 16) | (nt16_char(b1 & 15u) << 24);
				if (b == 0) lo = four; else hi = four;
			}
			s_seq[grp][2 * w] = lo; s_seq[grp][2 * w + 1] = hi;
		}
		const uint8_t *qp = sp + (lq + 1) / 2;
		const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(qp) & 3u);
		const uint32_t *q4 = reinterpret_cast<const uint32_t *>(qp - mis);
		const int nq4 = (lq + 3) / 4;
		for (int w = gl; w < nq4; w += GROUP) {
			uint32_t lo = q4[w];
			uint32_t hi = mis ? q4[w + 1] : 0u; // stays inside the entry or the 8 bytes of slack behind the last one
			uint32_t v = mis ? __builtin_amdgcn_alignbyte(hi, lo, mis) : lo;
			s_qual[grp][w] = qmiss ? 0x2a2a2a2au : v + 0x21212121u; // phred + 33 (qualities <= 93: no carry between bytes); '*' when absent
		}
	}
	__syncthreads();
	if (!active) return;
	if (single && gl == 0) p.qmiss[c] = qmiss ? 1 : 0;
	// one block = four pieces: sequence / quality of the left part, sequence / quality of the right part.  A piece position maps to a
	// character through seq_at / qual_at below (three sources: the LDS stage, the read's packed bytes, the consensus storage).
	const int total = 2 * (ll + lr);
	uint32_t *d = reinterpret_cast<uint32_t *>(out_str + p.str_off[c]);
	EventView v;
	const uint8_t *cs = nullptr, *cq = nullptr, *rs = nullptr, *rq = nullptr;
	const uint8_t *sq = reinterpret_cast<const uint8_t *>(s_seq[grp]);
	const uint8_t *qq = reinterpret_cast<const uint8_t *>(s_qual[grp]);
	if (!staged) {
		if (single) {
			v.sp = sp; v.qp = sp + (lq + 1) / 2; v.begin = begin; v.ll = ll; v.lr = lr; v.qmiss = qmiss;
		} else {
			const int64_t stride = 2ll * (p.c.SL + p.c.SR);
			cs = p.c.strings + (int64_t)p.c.mslot[j] * stride;
			cq = cs + p.c.SL; rs = cs + 2 * p.c.SL; rq = rs + p.c.SR;
		}
	}
	auto seq_at = [&](bool right, int i) -> uint32_t {
		if (staged) return sq[begin + (right ? ll : 0) + i];
		return single ? (uint32_t)(uint8_t)v.base(v.begin + (right ? ll : 0) + i) : (uint32_t)(right ? rs[i] : cs[ll - 1 - i]);
	};
	auto qual_at = [&](bool right, int i) -> uint32_t {
		if (staged) return qq[begin + (right ? ll : 0) + i];
		return single ? (uint32_t)(uint8_t)v.qual(v.begin + (right ? ll : 0) + i) : (uint32_t)(right ? rq[i] : cq[ll - 1 - i]);
	};
	for (int w = gl; w * 4 < total; w += GROUP) {
		uint32_t word = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int q = w * 4 + k; // byte position inside the cluster's block
			uint32_t ch = 0;
			if (q < total) {
				const bool right = q >= 2 * ll;
				const int r = right ? q - 2 * ll : q, S = right ? lr : ll; // offset inside the half, its sequence bytes
				ch = r >= S ? qual_at(right, r - S) : seq_at(right, r);
			}
			word |= ch << (8 * k);
		}
		d[w] = word;
	}
	pack_copy_cigar(p.c.ev, p.cig_ev[c], p.ncig[c], gl, out_cig + p.cig_off[c]);
}

// ---- packed table (ssv_clip_table_format 1 / 2): sequences as 4-bit codes, qualities W bits each ----
//
// A cluster's block is four pieces [seq_left | qual_left | seq_right | qual_right] at byte offsets that are not dword aligned, cut
// out of a read at arbitrary nibble / byte offsets.  Composing the block byte by byte costs ~100 instructions per byte; here every
// lane works on whole dwords twice:
//   0. the read's entry (packed bases + qualities, <= 480 B, wherever it lies in the batch) into LDS: one batch of independent loads
//      per cluster from the dword-aligned address below it; the misalignment is carried as a byte offset into everything that follows.
//   1. piece dwords into LDS: a sequence piece dword is 8 nibbles of the BAM-packed read = a 40-bit window of the source shifted by
//      0 or 4 bits (the table keeps BAM's nibble order); a quality piece dword is the next 32 bits of the stream of W-bit alphabet indices.
//      Tails are zeroed, and every piece sits between zero guard dwords.
//   2. output dword at block byte o = OR over the (1, rarely 2-3) pieces it overlaps of that piece's bytes [o - start, o - start + 4):
//      two LDS dwords and one alignbyte each; the guards supply the zeros on either side of a piece.
// 16 lanes per cluster.  Clusters of multi-event bins (consensus storage) and reads longer than PACK_MAX_LQ take the bytewise path.
constexpr int PACK_RAW_DWORDS = (PACK_MAX_LQ / 2 + PACK_MAX_LQ) / 4 + 3; // entry of a PACK_MAX_LQ read + misalignment + read-ahead
constexpr int PACK_LDS_DWORDS = (PACK_MAX_LQ / 8 + 2) + (PACK_MAX_LQ / 4 + 2) + 5 + 1; // two sequence + two quality pieces of left_len + right_len <= PACK_MAX_LQ (W = 8 worst case) + guards

template <int W, bool SLOW> // SLOW: the launch over p.slow_list (bytewise path); otherwise every cluster takes the dword path or is skipped
__global__ __launch_bounds__(BLOCK) void k_cluster_pack_codes(PackArgs p, const unsigned int *__restrict__ n_clusters_dev, uint8_t *__restrict__ out_str, uint32_t *__restrict__ out_cig)
{
	const int64_t n_clusters = (int64_t)*n_clusters_dev; // the grid is an upper bound (clusters: one group per event; slow list: multi-event slots + long reads)
	__shared__ uint32_t s_piece[GROUPS_PER_BLOCK][PACK_LDS_DWORDS];
	__shared__ uint32_t s_raw[GROUPS_PER_BLOCK][PACK_RAW_DWORDS]; // the read's entry (packed bases, qualities) as it lies in memory, from the aligned address below it
	__shared__ uint8_t s_lut[256]; // phred -> alphabet index
	__shared__ uint8_t s_code[256]; // character -> 4-bit code (bytewise path)
	__shared__ uint32_t s_seen[8];  // phred values met (bit set)
	if (W < 8) s_lut[threadIdx.x] = p.qlut[threadIdx.x]; // BLOCK == 256
	if (SLOW) s_code[threadIdx.x] = NT16_CODE_OF[threadIdx.x];
	if (threadIdx.x < 8) s_seen[threadIdx.x] = 0;
	const int grp = (int)(threadIdx.x / GROUP);
	const int gl = (int)(threadIdx.x % GROUP);
	const int64_t k_ = (int64_t)blockIdx.x * GROUPS_PER_BLOCK + grp;
	const int64_t c = SLOW ? (k_ < n_clusters ? (int64_t)p.slow_list[k_] : 0) : k_; // SLOW: n_clusters = entries of the list
	bool active = k_ < n_clusters;
	int ll = 0, lr = 0, lq = -1, begin = 0;
	uint64_t sptr = 0;
	int ncg = 0;
	uint32_t cev = 0;
	uint64_t dcig = 0, doff = 0;
	if (active) { ll = p.ll[c]; lr = p.lr[c]; lq = p.src_lq[c]; begin = p.src_begin[c]; sptr = p.src_ptr[c]; ncg = p.ncig[c]; cev = p.cig_ev[c]; dcig = p.cig_off[c]; doff = p.str_off[c]; }
	const bool fast = !SLOW && active && lq >= 0 && lq <= PACK_MAX_LQ && ll + lr <= PACK_MAX_LQ;
	if (!SLOW) active = fast; // the others are on the slow list
	// piece k: 0 seq_left, 1 qual_left, 2 seq_right, 3 qual_right; nB bytes, nD dwords, at block byte oP, at LDS dword st
	int nB[4], nD[4], oP[4], st[4];
	nB[0] = (ll + 1) / 2; nB[1] = (ll * W + 7) / 8; nB[2] = (lr + 1) / 2; nB[3] = (lr * W + 7) / 8;
	{
		int o = 0, s_ = 1;
#pragma unroll
		for (int k = 0; k < 4; ++k) { nD[k] = (nB[k] + 3) / 4; oP[k] = o; st[k] = s_; o += nB[k]; s_ += nD[k] + 1; }
	}
	const int total = oP[3] + nB[3];
	uint32_t *L = s_piece[grp];
	const uint32_t *s4 = s_raw[grp];
	const int mis = (int)(sptr & 3ull);     // the entry starts mis bytes into s_raw
	const int qb = mis + (lq + 1) / 2;      // first quality byte of the entry inside s_raw
	if (fast) {
		// 0. the whole entry into LDS with one batch of independent loads (one memory round trip per cluster; everything after reads LDS)
		const uint32_t *g4 = reinterpret_cast<const uint32_t *>((uintptr_t)(sptr - (uint64_t)mis));
		const int nraw = (qb + lq + 3) / 4 + 1; // + one dword of read-ahead for the unaligned windows below (8 bytes of slack behind seqqual, seeksv_hip.h)
		uint32_t r[PACK_RAW_DWORDS / GROUP + 1];
#pragma unroll
		for (int u = 0; u < PACK_RAW_DWORDS / GROUP + 1; ++u) { const int i = gl + GROUP * u; r[u] = i < nraw && i < PACK_RAW_DWORDS ? g4[i] : 0u; }
#pragma unroll
		for (int u = 0; u < PACK_RAW_DWORDS / GROUP + 1; ++u) { const int i = gl + GROUP * u; if (i < nraw && i < PACK_RAW_DWORDS) s_raw[grp][i] = r[u]; }
	}
	__syncthreads(); // s_raw, s_lut
	bool qmiss = false;
	if (fast) {
		qmiss = lq > 0 && ((s4[qb >> 2] >> (8 * (qb & 3))) & 0xffu) == 0xffu;
		if (gl == 0) { L[0] = 0u; p.qmiss[c] = qmiss ? 1 : 0; } // guards: before the first piece and after each piece
#pragma unroll
		for (int k = 0; k < 4; ++k) if (gl == k + 1) L[st[k] + nD[k]] = 0u;
		// sequence pieces (the dwords of both pieces share one index space: with 16 lanes and ~10 dwords per piece, a loop per piece
		// would leave half the lanes idle in each - and this kernel is VALU bound, 93 % busy in the PMC pass)
		for (int tt = gl; tt < nD[0] + nD[2]; tt += GROUP) {
			const bool h = tt >= nD[0];
			const int t = h ? tt - nD[0] : tt;
			const int nib0 = begin + (h ? ll : 0), len = h ? lr : ll;
			{
				const int B = mis + (nib0 >> 1) + 4 * t;
				const uint32_t d0 = s4[B >> 2], d1 = s4[(B >> 2) + 1];
				const uint64_t X = (((uint64_t)d1 << 32) | d0) >> (8 * (B & 3));
				const uint32_t lo = (uint32_t)X, nx = (uint32_t)(X >> 8);
				uint32_t v = (nib0 & 1) ? (((lo & 0x0f0f0f0fu) << 4) | ((nx >> 4) & 0x0f0f0f0fu)) : lo;
				const int rem = len - 8 * t; // nibbles of the piece in this dword
				if (rem < 8) v &= ((1u << (8 * (rem >> 1))) - 1u) | ((rem & 1) ? 0xf0u << (8 * (rem >> 1)) : 0u);
				L[(h ? st[2] : st[0]) + t] = v;
			}
		}
		// quality pieces: a piece is the bit stream of its W-bit indices, quality i at stream bit i * W (for W = 3 an index can straddle
		// a byte or a dword); dword t of the piece = stream bits [32 t, 32 t + 32) = the CNT qualities from i0 = 32 t / W on, shifted
		uint32_t seen_lo = 0, seen_hi = 0; // phred 0..63 as a bit set in registers (anything higher goes straight to LDS)
		for (int tt = gl; tt < nD[1] + nD[3]; tt += GROUP) {
			const bool h = tt >= nD[1];
			const int t = h ? tt - nD[1] : tt;
			const int q0 = begin + (h ? ll : 0), len = h ? lr : ll;
			{
				constexpr int CNT = W == 8 ? 4 : (W == 3 ? 12 : 32 / W); // qualities that can touch one dword
				const int i0 = (32 * t) / W, off = 32 * t - W * i0;   // off != 0 only for W = 3
				const int a = qb + q0 + i0;                            // first source byte
				const uint32_t *q4 = s4 + (a >> 2);
				const int sh = a & 3;
				const int rem = len - i0;                              // qualities of the piece from i0 on
				uint64_t acc = 0;
				uint32_t prev = q4[0];
#pragma unroll
				for (int g = 0; g < (CNT + 3) / 4; ++g) {
					if (4 * g >= rem) break; // nothing of the piece left (also keeps the reads inside the staged entry)
					const uint32_t next = q4[g + 1];
					const uint32_t four = __builtin_amdgcn_alignbyte(next, prev, sh);
					prev = next;
					if (W == 8) acc = qmiss ? 0x2a2a2a2au : four + 0x21212121u; // phred + 33 (no carries: qualities <= 93); '*' when absent
					else {
#pragma unroll
						for (int b = 0; b < 4; ++b) {
							const int j = 4 * g + b;
							const uint32_t ph = (four >> (8 * b)) & 0xffu;
							// every quality is encoded by exactly one dword of its piece's stream (the one holding its first bit): j < CNT' where CNT' stops
							// at the next dword's first quality; marking it "seen" here covers each quality at least once
							if (j < CNT && j < rem && !qmiss) {
								acc |= (uint64_t)(s_lut[ph] & ((1u << W) - 1u)) << (j * W);
								if (ph < 32) seen_lo |= 1u << ph; else if (ph < 64) seen_hi |= 1u << (ph - 32); else atomicOr(&s_seen[ph >> 5], 1u << (ph & 31));
							}
						}
					}
				}
				uint32_t v = (uint32_t)(acc >> off);
				if (W == 8 && rem < 4) v &= (1u << (8 * rem)) - 1u;
				L[(h ? st[3] : st[1]) + t] = v;
			}
		}
		if (W < 8) {
			if (seen_lo) atomicOr(&s_seen[0], seen_lo);
			if (seen_hi) atomicOr(&s_seen[1], seen_hi);
		}
	}
	__syncthreads();
	if (W < 8 && threadIdx.x < 8 && s_seen[threadIdx.x] && !SLOW) { // one look per workgroup; an atomic only while the set still grows
		const uint32_t have = __atomic_load_n(&p.qual_seen[threadIdx.x], __ATOMIC_RELAXED);
		if (s_seen[threadIdx.x] & ~have) atomicOr(&p.qual_seen[threadIdx.x], s_seen[threadIdx.x]);
	}
	if (!active) return;
	uint32_t *d = reinterpret_cast<uint32_t *>(out_str + doff);
	if (fast) {
		for (int w = gl; 4 * w < total; w += GROUP) {
			const int o = 4 * w;
			uint32_t word = 0;
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				if (o + 4 > oP[k] && o < oP[k] + nB[k]) {
					const int r0 = o - oP[k];     // -3 .. nB - 1
					const int i = st[k] + (r0 >> 2); // floor: -1 reads the guard in front of the piece
					word |= __builtin_amdgcn_alignbyte(L[i + 1], L[i], (uint32_t)(r0 & 3));
				}
			}
			d[w] = word;
		}
	} else if (SLOW) {
		// bytewise: from the read's packed bytes (long reads) or from the consensus storage (left part kept reversed)
		const int64_t j = p.slot[c];
		const bool single = lq >= 0;
		EventView v;
		const uint8_t *cs = nullptr, *cq = nullptr, *rs = nullptr, *rq = nullptr;
		bool qm;
		if (single) {
			v.sp = reinterpret_cast<const uint8_t *>((uintptr_t)sptr); v.qp = v.sp + (lq + 1) / 2; v.begin = begin; v.ll = ll; v.lr = lr; v.qmiss = lq > 0 && v.qp[0] == 0xff;
			qm = v.qmiss;
			if (gl == 0) p.qmiss[c] = qm ? 1 : 0;
		} else {
			const int64_t stride = 2ll * (p.c.SL + p.c.SR);
			cs = p.c.strings + (int64_t)p.c.mslot[j] * stride;
			cq = cs + p.c.SL; rs = cs + 2 * p.c.SL; rq = rs + p.c.SR;
			qm = p.c.c_qmiss[j] != 0;
		}
		auto seq_at = [&](bool right, int i) -> uint32_t {
			return s_code[single ? (uint32_t)(uint8_t)v.base(v.begin + (right ? ll : 0) + i) : (uint32_t)(right ? rs[i] : cs[ll - 1 - i])];
		};
		auto qual_at = [&](bool right, int i) -> uint32_t { // character
			return single ? (uint32_t)(uint8_t)v.qual(v.begin + (right ? ll : 0) + i) : (uint32_t)(right ? rq[i] : cq[ll - 1 - i]);
		};
		auto qual_index = [&](bool right, int i) -> uint32_t { // W < 8: alphabet index of quality i (0 when the cluster has no qualities)
			if (qm) return 0u;
			const uint32_t ph = (qual_at(right, i) - 33u) & 255u;
			atomicOr(&p.qual_seen[ph >> 5], 1u << (ph & 31)); // the slow list is a few thousand clusters
			return (uint32_t)(s_lut[ph] & ((1u << W) - 1u));
		};
		const int A = nB[0], QA = nB[1], C = nB[2];
		for (int w = gl; w * 4 < total; w += GROUP) {
			uint32_t word = 0;
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const int q = w * 4 + k;
				uint32_t ch = 0;
				if (q < total) {
					const bool right = q >= A + QA;
					const int r = right ? q - A - QA : q, S = right ? C : A, n = right ? lr : ll;
					if (r >= S) {
						if (W == 8) ch = qual_at(right, r - S);
						else { // stream bits [8 r', 8 r' + 8) of the quality piece
							const int bit0 = 8 * (r - S), i0 = bit0 / W, off = bit0 - W * i0;
							uint32_t acc = 0;
							for (int jq = 0, i = i0; W * jq < off + 8 && i < n; ++jq, ++i) acc |= qual_index(right, i) << (W * jq);
							ch = (acc >> off) & 0xffu;
						}
					} else ch = (seq_at(right, 2 * r) << 4) | (2 * r + 1 < n ? seq_at(right, 2 * r + 1) : 0u);
				}
				word |= ch << (8 * k);
			}
			d[w] = word;
		}
	}
	pack_copy_cigar(p.c.ev, cev, ncg, gl, out_cig + dcig);
}

} // namespace ssv
