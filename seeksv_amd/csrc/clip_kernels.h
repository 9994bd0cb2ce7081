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
	const uint8_t *ends;        // hot column, optional: first | last << 4 CIGAR operation codes (ssv_batch_t.cigar_ends)
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

// lane k of every aligned group of four lanes -> all four (DPP quad_perm broadcast: a plain VALU move, no LDS traffic)
template <int K> __device__ __forceinline__ uint32_t quad_bcast(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xf, 0xf, false); }

// The same line fetched by FOUR lanes, 16 bytes each: one vector memory instruction then carries 16 whole lines (64 contiguous bytes per
// quad) instead of 64 quarter lines, four instructions per record - the per-candidate kernels were bound by the issue of exactly those
// (PMC: 64 % of their wave cycles stalled on instruction issue).  All four lanes end up with the whole line.
__device__ __forceinline__ RecLine rec_load_quad(const ssv_record *rec, int64_t i, int q)
{
	const uint4 m = reinterpret_cast<const uint4 *>(rec + i)[q];
	RecLine r;
	r.w_tid = quad_bcast<0>(m.x); r.w_pos = quad_bcast<0>(m.y); r.w_fmx = quad_bcast<0>(m.z); r.w_nc = quad_bcast<0>(m.w);
	r.w_lq = quad_bcast<1>(m.x); r.w_mtid = quad_bcast<1>(m.y); r.w_mpos = quad_bcast<1>(m.z); r.w_isize = quad_bcast<1>(m.w);
	r.w_coff = quad_bcast<2>(m.x); r.h0 = quad_bcast<2>(m.y); r.h1 = quad_bcast<2>(m.z); r.h2 = quad_bcast<2>(m.w);
	r.h3 = quad_bcast<3>(m.x); r.h4 = quad_bcast<3>(m.y); r.so_lo = quad_bcast<3>(m.z); r.so_hi = quad_bcast<3>(m.w);
	return r;
}

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
constexpr int CC_TILE = BLOCK * 32;                   // clip scan: 8192 records per workgroup iteration
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
	int r_unsorted;                  // the '3' events are further from key order than the two-pass tile sort repairs: the radix sort takes over
};

// K1 clip_scan arguments: the streaming pass only needs the cigar_ends column
struct ClipScanArgs {
	const uint8_t *ends;     // the cigar_ends column (the batch's own, or built by k_build_ends)
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
	int64_t rec_begin;       // records before this index yield no events (ssv_clip_scan_range); they still count for the contig-switch rule
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
	if (nc < 1) return 0;                               // no CIGAR: the reference reads cigar[-1] (undefined there, nothing here)
	if (i < a.rec_begin) return 0;                      // ssv_clip_scan_range: the records before the range belong to the pass before
	const uint32_t c0 = r.head(0), cl = r.op(b.cigar, nc - 1); // (a lone "nS" is BOTH ends of its CIGAR, clip_reads.cpp:115)
	const int op1 = (int)(c0 & 15u), op2 = (int)(cl & 15u);
	if (op1 != C_S && op2 != C_S) return 0;             // clip_reads.cpp:124,150
	const int flag = r.flag();
	if (flag & (F_UNMAP | F_MUNMAP)) return 0;          // unmapped-pair side channel (host), clip_reads.h:415
	if (op1 == C_H || op2 == C_H || (flag & F_DUP) || r.mapq() < a.min_mapq) return 0; // clip_reads.cpp:118
	const int tid = r.tid();
	if (tid < 0) return 0;
	// contig-switch rule: processed only if tid equals the tid of the previous mapped-pair record (clip_reads.h:423-438)
	// (requested only for soft-clipped records: fetching it beside the line for every candidate cost more - two thirds of them are reads
	// with an indel that leave above - than the second round trip it saved)
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
		// A negative middle (a lone "nS", or clips that overlap): GetSeq's loops over it run zero times (clip_reads.cpp:293-294), the '5' event
		// is ([0, ll), ""), the '3' event ("", [l_qseq - rc, l_qseq)); an empty part makes every match rate 0 / 0 = NaN, so such an event
		// neither joins a cluster nor is joined (k_cluster_bins divides like the reference).  Clips longer than the read: the reference reads
		// outside the record (undefined there, nothing here).
		const int ll = (int)(c0 >> 4), rc = (int)(cl >> 4), mid = lq - ll - rc;
		if (ll > lq || rc > lq) return 0;
		const int midc = mid < 0 ? 0 : mid;
		bool do_l = true, do_r = true;
		if (xc != 0 && !a.save_low_quality) { if (!(flag & F_REV)) do_r = false; else do_l = false; } // clip_reads.cpp:160-175
		if (do_l) { evl.key = tkey | (uint32_t)(pos0 + 1); evl.begin = 0; evl.ll = ll; evl.lr = midc; m |= 1; }
		if (do_r) { evr.key = tkey | (1ull << 32) | (uint32_t)(pos0 + ref_len); evr.begin = ll + mid - midc; evr.ll = midc; evr.lr = rc; m |= 2; }
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
// SPARSE (the getsv scans: most wavefronts hold no candidate, and the caller keeps prefetched tiles in flight): a wavefront without a candidate skips the scan's
// 64-bit DPP steps, and the barrier is lds_barrier() - not __syncthreads(), which ends the loads in flight (nothing written to global memory here is read by this
// workgroup).  The clip scans (a candidate in nearly every wavefront, one tile ahead) were 6 us slower that way and keep the plain form.
template <int ITEMS, bool SPARSE = false>
__device__ __forceinline__ void stage_tile_candidates(uint32_t mask, uint64_t packed, int64_t tile, int64_t first_rec_of_lane, uint64_t (&lds)[2][WAVES_PER_BLOCK], int parity,
                                                      uint32_t &cursor, int64_t region, int64_t block_cap, uint32_t *tile_cnt, uint32_t *tile_off, uint32_t *stage, int *overflow)
{
	const uint64_t inc = !SPARSE || __ballot(mask != 0) ? wave_inclusive_sum(packed) : 0ull;
	if (lane_id() == 63) lds[parity][wave_id()] = inc;
	if (SPARSE) lds_barrier(); else __syncthreads();
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

// A batch that comes without the `cigar_ends` column gets one built here (one thread per record: its line, and the cigar array for CIGARs of
// more than five operations): the codes of the first and of the last operation, 0xff without CIGAR.  64 B/record instead of the 1 B/record
// the column costs when the batcher fills it while it parses the record anyway - all three batchers of this repository do.
__global__ __launch_bounds__(BLOCK) void k_build_ends(DevBatch b, uint8_t *__restrict__ ends)
{
	const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= b.n) return;
	const RecLine r = rec_load(b.rec, i);
	const int nc = r.n_cigar();
	ends[i] = nc ? (uint8_t)((r.head(0) & 15u) | ((r.op(b.cigar, nc - 1) & 15u) << 4)) : (uint8_t)0xff;
}

// K1 clip_scan: the streaming pass over the `cigar_ends` column: 1 B/record, sixteen records per 16-byte load, two loads per lane
// and tile (8192-record tiles); a record goes on when the code of its first or of its last CIGAR operation is S - the soft-clip
// test of GetSClipReads (clip_reads.cpp:124,150) applied to every record (a lone "nS" included: it is both ends of its CIGAR).
// Their lines are looked at by k_clip_filter, four lanes per candidate.  Persistent workgroups, private staging, no atomics.
constexpr int CE_ITEMS = 16, CE_SUB = 2;
static_assert(BLOCK * CE_ITEMS * CE_SUB == CC_TILE, "tile size of the clip scan");

__device__ __forceinline__ void clip_scan_ends_load(const ClipScanArgs &a, int64_t tile, uint4 (&v)[CE_SUB])
{
	const int64_t t0 = tile * CC_TILE + (int64_t)threadIdx.x * CE_ITEMS;
	if ((tile + 1) * CC_TILE <= a.n) {
#pragma unroll
		for (int sub = 0; sub < CE_SUB; ++sub) v[sub] = stream_load_u4(a.ends + t0 + (int64_t)sub * (BLOCK * CE_ITEMS));
	} else {
#pragma unroll
		for (int sub = 0; sub < CE_SUB; ++sub) {
			const int64_t i0 = t0 + (int64_t)sub * (BLOCK * CE_ITEMS);
			uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
			for (int k = 0; k < CE_ITEMS; ++k) w[k >> 2] |= (i0 + k < a.n ? (uint32_t)a.ends[i0 + k] : 0u) << (8 * (k & 3));
			v[sub] = make_uint4(w[0], w[1], w[2], w[3]);
		}
	}
}

// bit k of the result: byte k of w has the code 4 (S) in its low or in its high nibble
__device__ __forceinline__ uint32_t ends_have_s(uint32_t w)
{
	const uint32_t x = w ^ 0x44444444u;                                  // a nibble that was 4 is 0 now
	const uint32_t z = ~(((x & 0x77777777u) + 0x77777777u) | x | 0x77777777u); // bit 3 of every zero nibble (exact, no borrows)
	const uint32_t y = ((z | (z >> 4)) >> 3) & 0x01010101u;              // bit 0 of every byte with such a nibble
	return (y * 0x01020408u) >> 24;                                      // gathered into four bits
}

__global__ __launch_bounds__(BLOCK) void k_clip_scan_ends(ClipScanArgs a)
{
	__shared__ uint64_t lds[2][WAVES_PER_BLOCK];
	uint32_t cursor = 0;
	int parity = 0;
	const int64_t region = (int64_t)blockIdx.x * a.block_cap;
	uint4 v[CE_SUB], nxt[CE_SUB];
	if ((int64_t)blockIdx.x < a.ntiles) clip_scan_ends_load(a, blockIdx.x, v);
	for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x, parity ^= 1) {
		const int64_t t0 = tile * CC_TILE + (int64_t)threadIdx.x * CE_ITEMS;
		const int64_t next = tile + gridDim.x;
		if (next < a.ntiles) clip_scan_ends_load(a, next, nxt);
		uint32_t mask = 0;
		uint64_t packed = 0;
#pragma unroll
		for (int sub = 0; sub < CE_SUB; ++sub) {
			const uint32_t bits = ends_have_s(v[sub].x) | (ends_have_s(v[sub].y) << 4) | (ends_have_s(v[sub].z) << 8) | (ends_have_s(v[sub].w) << 12);
			mask |= bits << (sub * CE_ITEMS);
			packed += (uint64_t)__popc(bits) << (16 * sub);
		}
		stage_tile_candidates<CE_ITEMS>(mask, packed, tile, t0, lds, parity, cursor, region, a.block_cap, a.tile_cnt, a.tile_off, a.stage, a.overflow);
#pragma unroll
		for (int sub = 0; sub < CE_SUB; ++sub) v[sub] = nxt[sub];
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

// K1b clip_filter: four lanes per candidate record (an S at either end of its CIGAR) fetch the record's line - one 64-byte sector holds the CIGAR ends
// and every field of GetSClipReads' predicate chain (flag, MAPQ, DUP, XC, hard clips, lengths) - and leaves its 0, 1 or 2 events in the
// candidate's two event slots.  The slots are the events' final place (the event array has holes; everything downstream goes through
// indices), so an event line is written exactly once; what the ordered side lists need of an event - key, l_qseq, n_cigar: 16 bytes - is
// staged beside it for k_clip_place.  cnt[c] = bit 0: a '5' event, bit 1: a '3' event; wave_cnt[c / 64] = the wavefront's events | its '3' events << 32
// (the order-restoring scan runs over one value per wavefront, not per candidate).
__global__ __launch_bounds__(BLOCK) void k_clip_filter(ClipFilterArgs a, const uint32_t *__restrict__ cand, int64_t n_cand, ClipEvent *__restrict__ stash, uint4 *__restrict__ kv, uint8_t *__restrict__ cnt,
                                                       uint64_t *__restrict__ wave_cnt)
{
	// four lanes per candidate (rec_load_quad): a wavefront handles 16 candidates, every lane of a quad decides the same events and stores
	// its quarter of their lines
	const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	const int64_t c = t >> 2;
	const int q = (int)(t & 3);
	ClipEvent evl, evr;
	int m = 0;
	if (c < n_cand) {
		const int64_t i = cand[c];
		const RecLine r = rec_load_quad(a.b.rec, i, q);
		m = clip_events_of(a, i, r, evl, evr);
		if (q == 0) cnt[c] = (uint8_t)m;
	}
	// the wavefront's events side by side (two slots per candidate are reserved, the wave fills its 32 from the front): a third of
	// the candidates emit, and an event is one whole 64-byte line
	const int n = (m & 1) + (m >> 1);
	const int v = q == 0 ? n : 0;
	const int inc = wave_inclusive_sum(v);
	const int ex = (int)quad_bcast<0>((uint32_t)(inc - v));
	const uint64_t nr_wave = (uint64_t)__popcll(__ballot((m & 2) != 0 && q == 0));
	if (lane_id() == 63 && c - 15 < n_cand) wave_cnt[c >> 4] = (uint64_t)inc | (nr_wave << 32); // (whole wavefronts are launched: lane 63 exists also in the last one)
	const int64_t s0 = 2 * (c - (lane_id() >> 2)) + ex;
	auto quarter = [&](const ClipEvent &e) -> uint4 {
		const uint4 p0 = make_uint4((uint32_t)e.key, (uint32_t)(e.key >> 32), (uint32_t)e.src, (uint32_t)(e.src >> 32));
		const uint4 p1 = make_uint4((uint32_t)e.cig_ptr, (uint32_t)(e.cig_ptr >> 32), (uint32_t)e.begin, (uint32_t)e.ll);
		const uint4 p2 = make_uint4((uint32_t)e.lr, (uint32_t)e.lq, e.ncig, e.cig[0]);
		const uint4 p3 = make_uint4(e.cig[1], e.cig[2], e.cig[3], e.cig[4]);
		// (picked through 64-bit shifts: a plain chain of selects over these sixteen values becomes an indexed stack array under hipcc)
		auto pick = [&](uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) -> uint32_t {
			const uint64_t p01 = (uint64_t)w0 | ((uint64_t)w1 << 32), p23 = (uint64_t)w2 | ((uint64_t)w3 << 32);
			return (uint32_t)(((q & 2) ? p23 : p01) >> ((q & 1) << 5));
		};
		return make_uint4(pick(p0.x, p1.x, p2.x, p3.x), pick(p0.y, p1.y, p2.y, p3.y), pick(p0.z, p1.z, p2.z, p3.z), pick(p0.w, p1.w, p2.w, p3.w));
	};
	if (m & 1) {
		reinterpret_cast<uint4 *>(stash + s0)[q] = quarter(evl);
		if (q == 0) kv[s0] = make_uint4((uint32_t)evl.key, (uint32_t)(evl.key >> 32), (uint32_t)evl.lq, evl.ncig);
	}
	if (m & 2) {
		reinterpret_cast<uint4 *>(stash + s0 + (m & 1))[q] = quarter(evr);
		if (q == 0) kv[s0 + (m & 1)] = make_uint4((uint32_t)evr.key, (uint32_t)(evr.key >> 32), (uint32_t)evr.lq, evr.ncig);
	}
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
	uint64_t *key_l, *key_r; // keys of the side-0 / side-1 events, BAM order
	uint32_t *val_l, *val_r; // their event slots
	uint2 *meta;             // per event in BAM order (l_qseq, n_cigar): what the allocation bounds are reduced from without touching the lines again
	uint32_t *idx;           // per event in BAM order: its slot
};

// candidate c's events, in BAM order: keys and slots into the side lists, (l_qseq, n_cigar) and slot into the per-event arrays
__global__ __launch_bounds__(BLOCK) void k_clip_place(const uint4 *__restrict__ kv, const uint8_t *__restrict__ cnt, const uint64_t *__restrict__ wave_off, int64_t n_cand,
                                                      EventLists L, int64_t ev_base, int64_t l_base, int64_t r_base, int64_t slot_base)
{
	const int64_t c = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	const int m = c < n_cand ? (int)cnt[c] : 0;
	const int n = (m & 1) + (m >> 1);
	// k_clip_filter packed the events of every 16 candidates (one of its wavefronts) from the first of their 32 slots
	const int inc = wave_inclusive_sum(n);
	const int seg = lane_id() & ~15;
	const int before_seg = __shfl(inc, seg > 0 ? seg - 1 : 0, WAVE);
	const int ex = inc - n - (seg > 0 ? before_seg : 0);
	const uint64_t seg_mask = ~((1ull << seg) - 1ull);
	const int exr = (int)__popcll(__ballot((m & 2) != 0) & lanemask_lt() & seg_mask);
	if (n == 0) return;
	const int64_t s0 = 2 * (c - (lane_id() & 15)) + ex;
	const uint64_t off = wave_off[c >> 4];
	const int64_t e0 = ev_base + (int64_t)(uint32_t)off + ex;
	int64_t ir = r_base + (int64_t)(off >> 32) + exr, il = l_base + (int64_t)(uint32_t)off - (int64_t)(off >> 32) + (ex - exr);
	for (int k = 0; k < n; ++k) {
		const uint4 a = kv[s0 + k];
		const uint32_t slot = (uint32_t)(slot_base + s0 + k);
		L.meta[e0 + k] = make_uint2(a.z, a.w);
		L.idx[e0 + k] = slot;
		const uint64_t key = (uint64_t)a.x | ((uint64_t)a.y << 32);
		if ((key >> 32) & 1ull) { L.key_r[ir] = key; L.val_r[ir] = slot; ++ir; }
		else { L.key_l[il] = key; L.val_l[il] = slot; ++il; }
	}
}

// longest read / CIGAR of the batch's events, their CIGAR operations in total, the reads too long for the dword path of the pack kernels:
// grid-stride partials over the compact (l_qseq, n_cigar) pairs, reduced per workgroup; the maxima look before they touch the shared word
// (same-address atomics run at ~90 per microsecond)
__global__ __launch_bounds__(BLOCK) void k_event_max(const uint2 *__restrict__ meta, int64_t ev_base, ClipCounters *ctr)
{
	__shared__ unsigned long long s_sc[WAVES_PER_BLOCK], s_nl[WAVES_PER_BLOCK];
	__shared__ int s_lq[WAVES_PER_BLOCK], s_nc[WAVES_PER_BLOCK];
	const int64_t n_new = (int64_t)(uint32_t)ctr->n_new; // written by the scan of the per-wavefront counts
	unsigned long long sc = 0, nl = 0;
	int mlq = 0, mnc = 0;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_new; i += (int64_t)gridDim.x * blockDim.x) {
		const uint2 v = meta[ev_base + i];
		mlq = (int)v.x > mlq ? (int)v.x : mlq;
		mnc = (int)v.y > mnc ? (int)v.y : mnc;
		sc += v.y;
		nl += (int)v.x > PACK_MAX_LQ ? 1u : 0u;
	}
	mlq = wave_max(mlq); mnc = wave_max(mnc); sc = wave_sum(sc); nl = wave_sum(nl);
	if (lane_id() == 0) { s_sc[wave_id()] = sc; s_nl[wave_id()] = nl; s_lq[wave_id()] = mlq; s_nc[wave_id()] = mnc; }
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < WAVES_PER_BLOCK; ++w) { sc += s_sc[w]; nl += s_nl[w]; mlq = s_lq[w] > mlq ? s_lq[w] : mlq; mnc = s_nc[w] > mnc ? s_nc[w] : mnc; }
		if (mlq > __atomic_load_n(&ctr->max_lq, __ATOMIC_RELAXED)) atomicMax(&ctr->max_lq, mlq);
		if (mnc > __atomic_load_n(&ctr->max_ncig, __ATOMIC_RELAXED)) atomicMax(&ctr->max_ncig, mnc);
		if (sc) atomicAdd(&ctr->sum_ncig, sc);
		if (nl) atomicAdd(&ctr->n_long, nl);
	}
}

// largest key of a list (the '3' keys are not in order): *out = max
__global__ __launch_bounds__(BLOCK) void k_key_max(const uint64_t *__restrict__ key, int64_t n, unsigned long long *__restrict__ out)
{
	__shared__ unsigned long long s_mk[WAVES_PER_BLOCK];
	unsigned long long mk = 0;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { const unsigned long long k = key[i]; mk = k > mk ? k : mk; }
	mk = wave_max(mk);
	if (lane_id() == 0) s_mk[wave_id()] = mk;
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < WAVES_PER_BLOCK; ++w) mk = s_mk[w] > mk ? s_mk[w] : mk;
		if (mk > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, mk);
	}
}

// ---- the copying path (batches without SSV_MEM_PERSISTENT): the bytes an event points at move into context memory ----

constexpr int GROUP = 16;                      // lanes that cooperate on one event / cluster in the gather and pack kernels
constexpr int GROUPS_PER_WAVE = WAVE / GROUP;  // 4 items in flight per wavefront: the per-item metadata loads overlap
constexpr int GROUPS_PER_BLOCK = BLOCK / GROUP;

// bytes to copy per new event: packed bases + qualities (padded to 4: entries of the context blob start 4-byte aligned), long CIGARs
__global__ __launch_bounds__(BLOCK) void k_gather_sizes(const uint2 *__restrict__ meta, int64_t ev_base, int64_t n_new, uint32_t *__restrict__ seq_bytes, uint32_t *__restrict__ cig_ops)
{
	const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n_new) return;
	const uint2 v = meta[ev_base + i];
	const uint32_t lq = v.x, nc = v.y;
	seq_bytes[i] = ((lq + 1) / 2 + lq + 3u) & ~3u;
	cig_ops[i] = nc > 5 ? nc : 0u;
}

// 16 lanes per event copy its packed bases, qualities and (if longer than five operations) CIGAR into the context blobs and point the
// event at the copies.  The source may start anywhere, so every lane assembles aligned output dwords from two aligned source dwords.
__global__ __launch_bounds__(BLOCK) void k_clip_gather(ClipEvent *__restrict__ ev, const uint32_t *__restrict__ ev_idx, int64_t ev_base, int64_t n_new, const uint64_t *__restrict__ seq_off, const uint64_t *__restrict__ cig_off,
                                                       uint8_t *__restrict__ seq_blob, uint32_t *__restrict__ cig_blob)
{
	const int64_t w = (int64_t)blockIdx.x * GROUPS_PER_BLOCK + (threadIdx.x / GROUP);
	if (w >= n_new) return;
	const uint32_t gl = threadIdx.x % GROUP;
	ClipEvent *E = ev + ev_idx[ev_base + w];
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

// (contig, side, position) order = per contig: its '5' events, then its '3' events.  The events' lines move into that order too (one
// nearly sequential copy): everything downstream then reads slot j's line at evs[j] - coalesced, and one dependent load less per item than
// through a permutation.
__global__ __launch_bounds__(BLOCK) void k_merge_sides(const uint64_t *__restrict__ key_l, const uint32_t *__restrict__ val_l, int64_t nl, const uint64_t *__restrict__ key_r,
                                                       const uint32_t *__restrict__ val_r, int64_t nr, const uint32_t *__restrict__ cum_l, const uint32_t *__restrict__ cum_r,
                                                       const ClipEvent *__restrict__ ev, uint64_t *__restrict__ skey, ClipEvent *__restrict__ evs)
{
	// four lanes per event, each moves a quarter of the line: whole 64-byte lines per memory instruction on both sides
	const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	const int64_t i = t >> 2;
	const int q = (int)(t & 3);
	if (i >= nl + nr) return;
	uint64_t k;
	uint32_t v;
	int64_t o;
	if (i < nl) { k = key_l[i]; v = val_l[i]; o = i + cum_r[k >> 33]; }
	else { const int64_t j = i - nl; k = key_r[j]; v = val_r[j]; o = j + cum_l[(k >> 33) + 1]; }
	const uint4 w = reinterpret_cast<const uint4 *>(ev + v)[q];
	if (q == 0) skey[o] = k;
	reinterpret_cast<uint4 *>(evs + o)[q] = w;
}

// the same after the full sort (unsorted input)
__global__ __launch_bounds__(BLOCK) void k_gather_lines(const uint32_t *__restrict__ perm, int64_t n, const ClipEvent *__restrict__ ev, ClipEvent *__restrict__ evs)
{
	const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	const int64_t j = t >> 2;
	const int q = (int)(t & 3);
	if (j >= n) return;
	reinterpret_cast<uint4 *>(evs + j)[q] = reinterpret_cast<const uint4 *>(ev + perm[j])[q];
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
	int64_t E;
	const ClipEvent *ev;   // [E] the events' lines in sorted order
	double match_rate;
	// per sorted slot outputs
	int32_t *support;      // [E]; > 0 marks a cluster created by the event at this slot (single-event bins: 1, set by k_bin_mark)
	int32_t *c_ll, *c_lr;  // [E] clusters of multi-event bins only
	uint32_t *c_cig_ev;    // [E] sorted slot of the event whose CIGAR the cluster carries (multi-event bins only)
	uint8_t *c_qmiss;      // [E] (multi-event bins only)
	const uint32_t *mflag; // [E] 1: the slot belongs to a bin with more than one event
	const uint32_t *mslot; // [E] exclusive scan of mflag: index of the slot's string storage
	const uint32_t *mlist; // [M] the slots with mflag set, ascending (inverse of mslot)
	int64_t M;
	const uint32_t *blist; // the first slot of every multi-event bin
	const uint32_t *n_bins; // (device) their number; n_bins[1]: the number of deep bins
	const uint32_t *dlist; // the first slot of every bin of B4_DEEP events or more (k_cluster_bins4 starts these first)
	const uint16_t *tab4;  // k_bins4_tables' two tables
	int64_t deep_cap;      // workgroups of k_cluster_bins4 reserved for the deep list (an upper bound of its length)
	uint8_t *strings;      // [M * stride]: left seq (reversed), left qual (reversed), right seq, right qual - multi-event bins only
	int32_t SL, SR;        // capacity of a left / right string
};

constexpr int CL_CACHE = 64; // clusters of the current bin tracked in LDS; deeper bins fall back to scanning the slots

// bam_nt16_rev_table "=ACMGRSVTWYHKDBN" as two little-endian 64-bit words: nibble -> ASCII without touching memory
__device__ __forceinline__ uint32_t nt16_char(uint32_t nib)
{
	const uint64_t LO = ((uint64_t)'=') | ((uint64_t)'A' << 8) | ((uint64_t)'C' << 16) | ((uint64_t)'M' << 24) | ((uint64_t)'G' << 32) | ((uint64_t)'R' << 40) | ((uint64_t)'S' << 48) | ((uint64_t)'V' << 56);
	const uint64_t HI = ((uint64_t)'T') | ((uint64_t)'W' << 8) | ((uint64_t)'Y' << 16) | ((uint64_t)'H' << 24) | ((uint64_t)'K' << 32) | ((uint64_t)'D' << 40) | ((uint64_t)'B' << 48) | ((uint64_t)'N' << 56);
	return (uint32_t)(((nib & 8u) ? HI : LO) >> (8u * (nib & 7u))) & 0xffu;
}


struct EventView {
	const uint8_t *sp, *qp;
	int begin, ll, lr;
	bool qmiss;
	// i-th base of seq_left counted from its END (i = 0 is adjacent to the breakpoint side of the compare)
	__device__ __forceinline__ int lpos(int i) const { return begin + ll - 1 - i; }
	__device__ __forceinline__ int rpos(int i) const { return begin + ll + i; }
	// (arithmetic, not the NT16 table: a table in memory makes every base two dependent loads, and the per-bin walk is one long dependent chain)
	__device__ __forceinline__ char base(int p) const { return (char)nt16_char((uint32_t)(sp[p >> 1] >> ((~p & 1) << 2)) & 15u); }
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

// which entries of mlist start a bin (bflag), and after the scan of bflag (boff): the starts, densely - one wavefront of k_cluster_bins each.
// The flag is two counters in one 64-bit word: bin starts in the low half, starts of DEEP bins (B4_DEEP events or more: the planted
// breakpoints of a 300x sample) in the high half - one scan numbers both, and k_cluster_bins4 starts the deep list first.
constexpr int B4_DEEP = 16;
__global__ void k_bin_start_flags(const uint64_t *__restrict__ skey, int64_t E, const uint32_t *__restrict__ mlist, int64_t M, uint64_t *__restrict__ bflag)
{
	int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (m >= M) return;
	const int64_t j = mlist[m];
	const uint64_t k = skey[j];
	const bool start = j == 0 || skey[j - 1] != k;
	const bool deep = start && j + B4_DEEP - 1 < E && skey[j + B4_DEEP - 1] == k;
	bflag[m] = (start ? 1ull : 0ull) | (deep ? 1ull << 32 : 0ull);
}

__global__ void k_bin_start_list(const uint32_t *__restrict__ mlist, const uint64_t *__restrict__ bflag, const uint64_t *__restrict__ boff, int64_t M, uint32_t *__restrict__ blist, uint32_t *__restrict__ dlist)
{
	int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (m >= M) return;
	const uint64_t f = bflag[m];
	if (f == 0) return;
	const uint64_t o = boff[m];
	blist[(uint32_t)o] = mlist[m];
	if (f >> 32) dlist[o >> 32] = mlist[m];
}

// One wavefront per bin (= run of equal keys in the sorted event list).  The wave walks the bin's events in BAM order - the order the
// reference's multimap::equal_range scan sees them - and keeps the evolving clusters; lanes are spread over bases, match counts come from
// ballots.  Bins are independent, so there is no cross-wave communication.
// The walk is one long dependent chain (event -> its bytes -> compares against the clusters so far -> consensus update -> next event), and
// the kernel lasts as long as the longest bin (PMC: waves parked 85 % of their cycles).  So the chain stays on chip: the first BIN_KLDS
// clusters of a bin keep their consensus strings, and the first CL_CACHE their lengths / support, in LDS; an event's bytes are staged in
// LDS from loads issued one event ahead, its line is requested two events ahead.  Global memory sees the results once, at the end of the bin.
constexpr int BIN_ENT = 512;         // bytes of a read's entry staged in LDS (reads of up to PACK_MAX_LQ bases + misalignment)
constexpr int BIN_KLDS = 4;          // clusters per bin with LDS-resident strings (later ones: global storage)
constexpr int BIN_SL = PACK_MAX_LQ;  // capacity of an LDS-resident left / right string

struct BinMeta { int32_t ll, lr, support; uint32_t cig_ev; };

__global__ __launch_bounds__(BLOCK) void k_cluster_bins(ClusterArgs a)
{
	__shared__ int32_t s_slot[WAVES_PER_BLOCK][CL_CACHE];
	__shared__ BinMeta s_meta[WAVES_PER_BLOCK][CL_CACHE];
	__shared__ uint32_t s_ent[WAVES_PER_BLOCK][BIN_ENT / 4 + 4];
	__shared__ uint8_t s_str[WAVES_PER_BLOCK][BIN_KLDS][4 * BIN_SL];
	const int64_t m0 = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave_id();
	if (m0 >= (int64_t)*a.n_bins) return;         // the grid is an upper bound (every second slot of a multi-event bin could start one)
	const int64_t j0 = a.blist[m0];                // multi-event bins only (single-event bins were finished by k_bin_mark)
	const uint64_t key0 = a.skey[j0];
	const int lane = lane_id();
	const int w = wave_id();
	const bool left_clipped = ((key0 >> 32) & 1ull) == 0; // side '5' = breakpoint2read_l = LEFT_CLIPPED
	const int64_t stride = 2ll * (a.SL + a.SR);
	const bool lds_strings = a.SL <= BIN_SL && a.SR <= BIN_SL;
	// storage of cluster number k (creation order) created at `slot`: its four strings and their capacities
	struct Str { uint8_t *cs, *cq, *rs, *rq; };
	auto storage = [&](int k, int64_t slot) -> Str {
		Str t;
		if (lds_strings && k < BIN_KLDS) { t.cs = s_str[w][k]; t.cq = t.cs + BIN_SL; t.rs = t.cs + 2 * BIN_SL; t.rq = t.cs + 3 * BIN_SL; }
		else { t.cs = a.strings + (int64_t)a.mslot[slot] * stride; t.cq = t.cs + a.SL; t.rs = t.cs + 2 * a.SL; t.rq = t.rs + a.SR; }
		return t;
	};
	// per-cluster lengths / support / CIGAR carrier: LDS for the first CL_CACHE clusters, the per-slot arrays beyond
	auto get_ll = [&](int k, int64_t slot) -> int { return k < CL_CACHE ? s_meta[w][k].ll : a.c_ll[slot]; };
	auto get_lr = [&](int k, int64_t slot) -> int { return k < CL_CACHE ? s_meta[w][k].lr : a.c_lr[slot]; };
	int nclu = 0;
	// pipeline: line of event jj + 1 and bytes of event jj are in flight when event jj - 1 is being compared
	auto line_load = [&](int64_t j, uint4 &la, uint4 &lb, uint4 &lc) { const uint4 *ep = reinterpret_cast<const uint4 *>(a.ev + j); la = ep[0]; lb = ep[1]; lc = ep[2]; };
	auto bytes_load = [&](const uint4 &la, const uint4 &lc, uint32_t &b0, uint32_t &b1) {
		const uint64_t src = (uint64_t)la.z | ((uint64_t)la.w << 32);
		const int lq = (int)lc.y, mis = (int)(src & 3ull);
		const int nraw = (mis + (lq + 1) / 2 + lq + 3) / 4;
		const uint32_t *g4 = reinterpret_cast<const uint32_t *>((uintptr_t)(src - (uint64_t)mis));
		const bool staged = nraw <= BIN_ENT / 4;
		b0 = staged && lane < nraw ? g4[lane] : 0u;
		b1 = staged && lane + WAVE < nraw ? g4[lane + WAVE] : 0u;
	};
	uint4 ca, cb, cc, na, nb_, nc_;
	uint32_t b0, b1, nb0 = 0, nb1 = 0;
	line_load(j0, ca, cb, cc);
	bytes_load(ca, cc, b0, b1);
	bool more = j0 + 1 < a.E && a.skey[j0 + 1] == key0;
	if (more) line_load(j0 + 1, na, nb_, nc_);
	for (int64_t jj = j0;; ++jj) {
		const uint32_t e = (uint32_t)jj;
		const bool more2 = more && jj + 2 < a.E && a.skey[jj + 2] == key0;
		uint4 fa, fb, fc; // line of event jj + 2
		if (more) bytes_load(na, nc_, nb0, nb1);
		if (more2) line_load(jj + 2, fa, fb, fc);
		EventView v;
		const int lq = (int)cc.y;
		const uint64_t src = (uint64_t)ca.z | ((uint64_t)ca.w << 32);
		const int mis = (int)(src & 3ull);
		const bool staged = (mis + (lq + 1) / 2 + lq + 3) / 4 <= BIN_ENT / 4;
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the previous event's readers of s_ent are done
		if (staged) {
			s_ent[w][lane] = b0; s_ent[w][lane + WAVE] = b1;
			v.sp = reinterpret_cast<const uint8_t *>(s_ent[w]) + mis;
		} else v.sp = reinterpret_cast<const uint8_t *>((uintptr_t)src);
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		v.qp = v.sp + (lq + 1) / 2;
		v.begin = (int)cb.z; v.ll = (int)cb.w; v.lr = (int)cc.x;
		v.qmiss = lq > 0 && v.qp[0] == 0xff; // no qualities: the row prints "*" (clip_reads.cpp:296)
		// ---- find the first cluster of the bin that absorbs this event (clip_reads.cpp:262-273) ----
		auto absorbs = [&](int k, int64_t slot) -> bool {
			const Str t = storage(k, slot);
			const int cll = get_ll(k, slot), clr = get_lr(k, slot);
			const int n1 = v.ll < cll ? v.ll : cll;
			int m1 = 0;
			for (int i0 = 0; i0 < n1; i0 += WAVE) {
				int i = i0 + lane;
				bool eq = i < n1 && v.base(v.lpos(i)) == (char)t.cs[i];
				m1 += (int)__popcll(__ballot(eq));
			}
			if (!((double)m1 / (double)n1 >= a.match_rate)) return false; // n1 == 0 -> NaN -> false, like the reference
			const int n2 = v.lr < clr ? v.lr : clr;
			int m2 = 0;
			for (int i0 = 0; i0 < n2; i0 += WAVE) {
				int i = i0 + lane;
				bool eq = i < n2 && v.base(v.rpos(i)) == (char)t.rs[i];
				m2 += (int)__popcll(__ballot(eq));
			}
			return (double)m2 / (double)n2 >= a.match_rate;
		};
		int hit_k = -1;
		int64_t hit = -1;
		const int kmax = nclu < CL_CACHE ? nclu : CL_CACHE;
		for (int k = 0; k < kmax && hit < 0; ++k) {
			const int64_t slot = j0 + s_slot[w][k];
			if (absorbs(k, slot)) { hit = slot; hit_k = k; }
		}
		if (hit < 0 && nclu > CL_CACHE) {
			// clusters beyond the LDS cache: they were created after the last cached one, i.e. at later slots
			for (int64_t s_ = j0 + s_slot[w][CL_CACHE - 1] + 1; s_ < jj && hit < 0; ++s_)
				if (a.support[s_] > 0 && absorbs(CL_CACHE, s_)) { hit = s_; hit_k = CL_CACHE; }
		}
		if (hit >= 0) {
			// ---- ReadsInfo::ChangeSeqAndQual (clip_reads.cpp:57-108) on the reversed-left / forward-right storage ----
			const Str t = storage(hit_k, hit);
			const int cll = get_ll(hit_k, hit), clr = get_lr(hit_k, hit);
			const int n1 = v.ll < cll ? v.ll : cll;
			for (int i = lane; i < v.ll; i += WAVE) {
				int p = v.lpos(i);
				char q = v.qual(p);
				if (i < n1) {
					if ((signed char)t.cq[i] < (signed char)q) { t.cq[i] = (uint8_t)q; t.cs[i] = (uint8_t)v.base(p); }
				} else if (cll <= v.ll) { t.cs[i] = (uint8_t)v.base(p); t.cq[i] = (uint8_t)q; } // prepend the extra prefix
			}
			const int n2 = v.lr < clr ? v.lr : clr;
			for (int i = lane; i < v.lr; i += WAVE) {
				int p = v.rpos(i);
				char q = v.qual(p);
				if (i < n2) {
					if ((signed char)t.rq[i] < (signed char)q) { t.rq[i] = (uint8_t)q; t.rs[i] = (uint8_t)v.base(p); }
				} else if (clr < v.lr) { t.rs[i] = (uint8_t)v.base(p); t.rq[i] = (uint8_t)q; } // append the extra suffix
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			if (hit_k < CL_CACHE) {
				if (lane == 0) {
					BinMeta &mt = s_meta[w][hit_k];
					if (cll <= v.ll) { mt.ll = v.ll; if (!left_clipped) mt.cig_ev = e; } // aa == RIGHT_CLIPPED (also when the lengths are equal)
					if (clr < v.lr) { mt.lr = v.lr; if (left_clipped) mt.cig_ev = e; }    // aa == LEFT_CLIPPED
					mt.support += 1;
				}
			} else {
				if (cll <= v.ll) { a.c_ll[hit] = v.ll; if (!left_clipped) a.c_cig_ev[hit] = e; }
				if (clr < v.lr) { a.c_lr[hit] = v.lr; if (left_clipped) a.c_cig_ev[hit] = e; }
				a.support[hit] += 1; // every lane stores the same value; each lane later reads back what it stored
			}
		} else {
			// ---- new cluster at this event's slot (clip_reads.cpp:276-281) ----
			const int k = nclu < CL_CACHE ? nclu : CL_CACHE;
			const Str t = storage(k, jj);
			for (int i = lane; i < v.ll; i += WAVE) { int p = v.lpos(i); t.cs[i] = (uint8_t)v.base(p); t.cq[i] = (uint8_t)v.qual(p); }
			for (int i = lane; i < v.lr; i += WAVE) { int p = v.rpos(i); t.rs[i] = (uint8_t)v.base(p); t.rq[i] = (uint8_t)v.qual(p); }
			a.c_qmiss[jj] = v.qmiss ? 1 : 0;
			if (nclu < CL_CACHE) {
				if (lane == 0) { s_slot[w][nclu] = (int32_t)(jj - j0); BinMeta &mt = s_meta[w][nclu]; mt.ll = v.ll; mt.lr = v.lr; mt.support = 1; mt.cig_ev = e; }
			} else { a.c_ll[jj] = v.ll; a.c_lr[jj] = v.lr; a.c_cig_ev[jj] = e; a.support[jj] = 1; }
			++nclu;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		if (!more) break;
		ca = na; cb = nb_; cc = nc_; b0 = nb0; b1 = nb1;
		more = more2;
		if (more2) { na = fa; nb_ = fb; nc_ = fc; }
	}
	// ---- the bin's results to global memory: lengths / support / CIGAR carrier of the cached clusters, strings of the LDS-resident ones ----
	const int kc = nclu < CL_CACHE ? nclu : CL_CACHE;
	for (int k = lane; k < kc; k += WAVE) {
		const int64_t slot = j0 + s_slot[w][k];
		const BinMeta mt = s_meta[w][k];
		a.c_ll[slot] = mt.ll; a.c_lr[slot] = mt.lr; a.c_cig_ev[slot] = mt.cig_ev; a.support[slot] = mt.support;
	}
	if (lds_strings) {
		const int ks = nclu < BIN_KLDS ? nclu : BIN_KLDS;
		for (int k = 0; k < ks; ++k) {
			const int64_t slot = j0 + s_slot[w][k];
			const int ll = s_meta[w][k].ll, lr = s_meta[w][k].lr;
			uint8_t *g = a.strings + (int64_t)a.mslot[slot] * stride;
			const uint8_t *l = s_str[w][k];
			for (int i = lane; i < ll; i += WAVE) { g[i] = l[i]; g[a.SL + i] = l[BIN_SL + i]; }
			for (int i = lane; i < lr; i += WAVE) { g[2 * a.SL + i] = l[2 * BIN_SL + i]; g[2 * a.SL + a.SR + i] = l[3 * BIN_SL + i]; }
		}
	}
}

// ---- k_cluster_bins4: the same walk with FOUR positions per lane (round 5) ----
// At 300x (BASELINE config 3) a planted breakpoint's bin holds 50-150 reads and the walk above is bound by its vector instructions: a lane
// per base, every base a byte read through a pointer that may be LDS or global (flat instructions), three rounds per side for a 150-base
// read, two fp64 divides per compare - about a thousand instructions per event.  Here a lane owns the positions 4 lane .. 4 lane + 3 of
// either side (reads of up to B4_CAP = 256 bases: one round), an event's bases and qualities come out of its staged entry ONCE - one
// unaligned dword of packed bases through a byte -> two-characters table, one unaligned dword of qualities, +33 on all four at once - into
// four registers that every compare and the update share; the consensus strings are dwords in LDS, compared with an XOR and a zero-byte
// test, updated with a four-byte signed compare and a bit-field insert; the match counts are ballots over the bits of the lanes' counts;
// and `(double) m / (double) n >= rate` is looked up: the smallest m that passes for every n, found with that very division when the
// workgroup starts (the quotient grows with m, so the table says exactly what the division would).  Clusters beyond the B4_KLDS-th of a
// bin keep their strings in global memory as before (same code, instantiated for that address space).  The results are bit-identical to
// k_cluster_bins, which stays for passes with a read longer than 256 bases.
constexpr int B4_CAP = 256;   // positions of a side that one round of 64 lanes covers
constexpr int B4_KLDS = 3;    // clusters per bin with LDS-resident strings
constexpr int B4_ENT = 132;   // dwords of a staged entry: one in front (a left part's dword may begin up to three bases before the read), 3 + 128 + 256 bytes, slack behind

__device__ __forceinline__ uint32_t b4_below(int L, int lane) // byte k is 0xff where position 4 lane + k < L
{
	const int r = L - 4 * lane;
	return r >= 4 ? 0xffffffffu : r <= 0 ? 0u : (1u << (8 * r)) - 1u;
}
__device__ __forceinline__ uint32_t b4_zero_bytes(uint32_t x) { return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu); } // 0x80 in every byte that is 0
__device__ __forceinline__ uint32_t b4_lt_s8(uint32_t a, uint32_t b) // 0xff in every byte where (signed char) a < (signed char) b
{
	const uint32_t x = a ^ 0x80808080u, y = b ^ 0x80808080u; // signed order -> unsigned order
	const uint32_t d = (x | 0x80808080u) - (y & 0x7f7f7f7fu); // per byte 128 + low7(x) - low7(y): no borrow between bytes
	const uint32_t lt = ((~x & y) | (~(x ^ y) & ~d)) & 0x80808080u;
	return (lt << 1) - (lt >> 7);
}
__device__ __forceinline__ uint32_t b4_bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }
__device__ __forceinline__ uint32_t b4_lds_u32(const uint32_t *ent, int byte_off) // four bytes at any byte offset of a dword array
{
	const int d = byte_off >> 2;
	return __builtin_amdgcn_alignbyte(ent[d + 1], ent[d], (uint32_t)(byte_off & 3));
}

struct Ev4 { uint32_t lb, lq, rb, rq; int ll, lr; bool qmiss; };

constexpr int B4_TAB = 256 + B4_CAP + 2; // 16-bit entries: a BAM byte -> its two bases as characters (first base in the low byte) | the smallest m that passes for n = 0 .. 256
// the two tables of a pass, once: what the compare `(double) m / (double) n >= match_rate` says for every n, found with that division
__global__ __launch_bounds__(BLOCK) void k_bins4_tables(double match_rate, uint16_t *__restrict__ tab)
{
	const int t = (int)threadIdx.x;
	tab[t] = (uint16_t)(nt16_char((uint32_t)t >> 4) | (nt16_char((uint32_t)t & 15u) << 8));
	for (int n = t; n <= B4_CAP; n += BLOCK) {
		const double r = match_rate;
		int m = 0xffff;                                    // n == 0: 0 / 0 is NaN and compares false, like the reference's division
		if (n > 0 && r == r) {
			const double g = ceil(r * (double)n);
			m = g < 0.0 ? 0 : g > (double)(n + 1) ? n + 1 : (int)g;
			while (m > 0 && (double)(m - 1) / (double)n >= r) --m;
			while (m <= n && !((double)m / (double)n >= r)) ++m;
		}
		tab[256 + n] = (uint16_t)m;
	}
	if (t == 0) tab[B4_TAB - 1] = 0;
}

// One wavefront = one workgroup = one bin: a workgroup of four held its LDS until its deepest bin was done, and at 300x every sixth
// workgroup has a deep one.  The first deep_cap workgroups take the list of deep bins (they start first: the kernel ends with the shallow
// bins, not with a 150-event walk that began late), the others the list of all bins and leave when theirs is deep.  (Measured and
// dropped, profiles/r05_config3_notes.txt: a persistent grid whose wavefronts request the next bin's keys, lines and first bytes while
// they walk the present one - 1.06 against 0.95 ms at 300x: the kernel is bound by its vector instructions, not by a bin's start-up
// latency, which the other wavefronts of the CU cover, and the bookkeeping of the second bin in flight costs more than it hides.)
__global__ __launch_bounds__(WAVE) void k_cluster_bins4(ClusterArgs a)
{
	__shared__ int32_t s_slot[CL_CACHE];
	__shared__ BinMeta s_meta[CL_CACHE];
	__shared__ uint32_t s_ent[B4_ENT];
	__shared__ uint32_t s_str[B4_KLDS][4][WAVE]; // per cluster: left bases (reversed), left qualities (reversed), right bases, right qualities
	__shared__ uint32_t s_tab[B4_TAB / 2];
	const uint16_t *s_lut2 = reinterpret_cast<const uint16_t *>(s_tab), *s_minm = s_lut2 + 256;
	const int lane = lane_id();
	const bool deep_role = (int64_t)blockIdx.x < a.deep_cap;
	const int64_t m = deep_role ? (int64_t)blockIdx.x : (int64_t)blockIdx.x - a.deep_cap;
	if (m >= (int64_t)a.n_bins[deep_role ? 1 : 0]) return; // the grid is an upper bound (every second slot of a multi-event bin could start one)
	struct Line { uint2 s, bl, rq; };            // src | begin, ll | lr, lq of an event line
	struct Bin { int64_t j0; uint64_t key0, keys_after; Line L0, L1, L2; };
	auto line_load = [&](int64_t j, Line &L) { const uint2 *ep = reinterpret_cast<const uint2 *>(a.ev + j); L.s = ep[1]; L.bl = ep[3]; L.rq = ep[4]; };
	auto bin_request = [&](int64_t j0, Bin &B) {
		B.j0 = j0;
		B.key0 = a.skey[j0];
		B.keys_after = j0 + 1 + lane < a.E ? a.skey[j0 + 1 + lane] : ~0ull;
		line_load(j0, B.L0);
		line_load(j0 + 1, B.L1); // (a bin has at least two events)
		B.L2.s = B.L2.bl = B.L2.rq = make_uint2(0u, 0u);
		if (j0 + 2 < a.E) line_load(j0 + 2, B.L2);
	};
	const int64_t stride = 2ll * (a.SL + a.SR);            // (SL = SR = a multiple of four: the dwords of a string never reach into the next one)
	uint32_t *ent = s_ent;
	auto bytes_load = [&](const Line &L, uint32_t &b0, uint32_t &b1) {
		const uint64_t src = (uint64_t)L.s.x | ((uint64_t)L.s.y << 32);
		const int lq = (int)L.rq.y, mis = (int)(src & 3ull);
		const int nraw = (mis + (lq + 1) / 2 + lq + 3) / 4;   // <= 97 dwords for a read of 256 bases
		const gptr<uint32_t> g4 = global_at<uint32_t>(src - (uint64_t)mis);
		b0 = lane < nraw ? g4[lane] : 0u;
		b1 = lane + WAVE < nraw ? g4[lane + WAVE] : 0u;
	};
	// the event's four registers out of its staged entry (GetSeq, clip_reads.cpp:286-306, four positions at a time)
	auto extract = [&](const Line &L) -> Ev4 {
		Ev4 v;
		const int begin = (int)L.bl.x, lq = (int)L.rq.y;
		v.ll = (int)L.bl.y; v.lr = (int)L.rq.x;
		const int eb = 4 + (int)(L.s.x & 3u), nbytes = (lq + 1) / 2;
		v.qmiss = lq > 0 && (b4_lds_u32(ent, eb + nbytes) & 0xffu) == 0xffu; // no qualities: the row prints "*" (clip_reads.cpp:296)
		// right part, ascending: positions begin + ll + 4 lane + k
		const int p0 = 4 * lane < v.lr ? begin + v.ll + 4 * lane : begin;
		uint32_t N = __builtin_bswap32(b4_lds_u32(ent, eb + (p0 >> 1))) << (4 * (p0 & 1));
		v.rb = (uint32_t)s_lut2[N >> 24] | ((uint32_t)s_lut2[(N >> 16) & 0xffu] << 16);
		uint32_t q = b4_lds_u32(ent, eb + nbytes + p0);
		v.rq = v.qmiss ? 0x2a2a2a2au : (((q & 0x7f7f7f7fu) + 0x21212121u) ^ (q & 0x80808080u));
		// left part, reversed: positions begin + ll - 1 - 4 lane - k, i.e. the four that END at begin + ll - 1 - 4 lane, byte-swapped
		const int pb = 4 * lane < v.ll ? begin + v.ll - 4 - 4 * lane : begin; // (>= -3: the dword in front of the entry covers it)
		N = __builtin_bswap32(b4_lds_u32(ent, eb + (pb >> 1))) << (4 * (pb & 1));
		v.lb = __builtin_bswap32((uint32_t)s_lut2[N >> 24] | ((uint32_t)s_lut2[(N >> 16) & 0xffu] << 16));
		q = __builtin_bswap32(b4_lds_u32(ent, eb + nbytes + pb));
		v.lq = v.qmiss ? 0x2a2a2a2au : (((q & 0x7f7f7f7fu) + 0x21212121u) ^ (q & 0x80808080u));
		return v;
	};
	// one cluster against the event: its four dwords of this lane (kept for the update) and the verdict (clip_reads.cpp:194-217, 262-273)
	struct Clu { uint32_t cs, cq, rs, rq; };
	auto absorbs = [&](const Ev4 &v, const uint32_t *cs, const uint32_t *cq, const uint32_t *rs, const uint32_t *rq, int cll, int clr, Clu &c) __attribute__((always_inline)) -> bool {
		const bool inl = 4 * lane < cll, inr = 4 * lane < clr;
		c.cs = inl ? cs[lane] : 0u; c.cq = inl ? cq[lane] : 0u;
		c.rs = inr ? rs[lane] : 0u; c.rq = inr ? rq[lane] : 0u;
		const int n1 = v.ll < cll ? v.ll : cll, n2 = v.lr < clr ? v.lr : clr;
		const uint32_t zl = b4_zero_bytes(v.lb ^ c.cs) & b4_below(n1, lane), zr = b4_zero_bytes(v.rb ^ c.rs) & b4_below(n2, lane);
		const uint32_t cnt = (uint32_t)__popc(zl) | ((uint32_t)__popc(zr) << 3);
		const int m1 = (int)__popcll(__ballot((cnt & 1u) != 0)) + 2 * (int)__popcll(__ballot((cnt & 2u) != 0)) + 4 * (int)__popcll(__ballot((cnt & 4u) != 0));
		const int m2 = (int)__popcll(__ballot((cnt & 8u) != 0)) + 2 * (int)__popcll(__ballot((cnt & 16u) != 0)) + 4 * (int)__popcll(__ballot((cnt & 32u) != 0));
		return m1 >= (int)s_minm[n1] && m2 >= (int)s_minm[n2];
	};
	// ReadsInfo::ChangeSeqAndQual (clip_reads.cpp:57-108): below the shorter length the better quality wins, beyond it the longer string is taken over
	auto merge = [&](const Ev4 &v, uint32_t *cs, uint32_t *cq, uint32_t *rs, uint32_t *rq, int cll, int clr, const Clu &c) __attribute__((always_inline)) {
		const int n1 = v.ll < cll ? v.ll : cll, n2 = v.lr < clr ? v.lr : clr;
		const uint32_t bl = b4_below(n1, lane), br = b4_below(n2, lane);
		const uint32_t sl = (b4_lt_s8(c.cq, v.lq) & bl) | (b4_below(v.ll, lane) & ~bl);
		const uint32_t sr = (b4_lt_s8(c.rq, v.rq) & br) | (b4_below(v.lr, lane) & ~br);
		if (sl) { cs[lane] = b4_bfi(sl, v.lb, c.cs); cq[lane] = b4_bfi(sl, v.lq, c.cq); }
		if (sr) { rs[lane] = b4_bfi(sr, v.rb, c.rs); rq[lane] = b4_bfi(sr, v.rq, c.rq); }
	};
	auto create = [&](const Ev4 &v, uint32_t *cs, uint32_t *cq, uint32_t *rs, uint32_t *rq) __attribute__((always_inline)) {
		const uint32_t ml = b4_below(v.ll, lane), mr = b4_below(v.lr, lane);
		if (ml) { cs[lane] = v.lb & ml; cq[lane] = v.lq & ml; }
		if (mr) { rs[lane] = v.rb & mr; rq[lane] = v.rq & mr; }
	};
	auto gstr = [&](int64_t slot) -> uint32_t * { return reinterpret_cast<uint32_t *>(a.strings + (int64_t)a.mslot[slot] * stride); };
	const int gl = a.SL / 4, gr = a.SR / 4; // a global string's dwords

	// ---- start-up: the bin's keys and first lines, the two tables ----
	Bin cur;
	bin_request(deep_role ? (int64_t)a.dlist[m] : (int64_t)a.blist[m], cur);
	{
		const uint32_t *g = reinterpret_cast<const uint32_t *>(a.tab4);
		for (int i = lane; i < B4_TAB / 2; i += WAVE) s_tab[i] = g[i];
		if (lane == 0) ent[0] = 0u;
	}
	uint32_t f0, f1, b0, b1; // bytes of the bin's first and second event
	bytes_load(cur.L0, f0, f1);
	bytes_load(cur.L1, b0, b1);
	{
		const int64_t j0 = cur.j0;
		const uint64_t key0 = cur.key0;
		// the bin's extent: the keys behind its first slot, 64 at a time
		int64_t nb = 1;
		{
			uint64_t keys_after = cur.keys_after;
			for (;;) {
				const uint64_t same = __ballot(keys_after == key0);
				const int run = same == ~0ull ? WAVE : (int)__builtin_ctzll(~same);
				nb += run;
				if (run < WAVE || !deep_role) break; // (a shallow-list wavefront only needs to know that the bin is deep)
				keys_after = j0 + nb + lane < a.E ? a.skey[j0 + nb + lane] : ~0ull;
			}
		}
		if (deep_role || nb < B4_DEEP) { // (a deep bin met in the list of all bins belongs to the deep list's wavefronts)
			const bool left_clipped = ((key0 >> 32) & 1ull) == 0; // side '5' = breakpoint2read_l = LEFT_CLIPPED
			Line L1 = cur.L1, L2 = cur.L2;
			ent[1 + lane] = f0; ent[1 + lane + WAVE] = f1;
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			Ev4 v = extract(cur.L0);
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			int nclu = 0;
			for (int64_t t = 0; t < nb; ++t) {
				const int64_t jj = j0 + t;
				const uint32_t e = (uint32_t)jj;
				// bytes of event t + 1 into the staging (event t lives in registers), bytes of t + 2 and the line of t + 3 requested
				if (t + 1 < nb) { ent[1 + lane] = b0; ent[1 + lane + WAVE] = b1; }
				Line L3;
				L3.s = L3.bl = L3.rq = make_uint2(0u, 0u);
				if (t + 2 < nb) bytes_load(L2, b0, b1);
				if (t + 3 < nb) line_load(jj + 3, L3);
				// ---- the first cluster of the bin that takes this event in ----
				int hit_k = -1, hit_ll = 0, hit_lr = 0;
				int64_t hit = -1;
				Clu c;
				c.cs = c.cq = c.rs = c.rq = 0u;
				const int kmax = nclu < CL_CACHE ? nclu : CL_CACHE;
				for (int k = 0; k < kmax && hit < 0; ++k) {
					const int64_t slot = j0 + s_slot[k];
					const int cll = s_meta[k].ll, clr = s_meta[k].lr;
					bool ok;
					if (k < B4_KLDS) ok = absorbs(v, s_str[k][0], s_str[k][1], s_str[k][2], s_str[k][3], cll, clr, c);
					else { const uint32_t *g = gstr(slot); ok = absorbs(v, g, g + gl, g + 2 * gl, g + 2 * gl + gr, cll, clr, c); }
					if (ok) { hit = slot; hit_k = k; hit_ll = cll; hit_lr = clr; }
				}
				if (hit < 0 && nclu > CL_CACHE) {
					// clusters beyond the LDS cache: they were created after the last cached one, i.e. at later slots
					for (int64_t s_ = j0 + s_slot[CL_CACHE - 1] + 1; s_ < jj && hit < 0; ++s_)
						if (a.support[s_] > 0) {
							const uint32_t *g = gstr(s_);
							const int cll = a.c_ll[s_], clr = a.c_lr[s_];
							if (absorbs(v, g, g + gl, g + 2 * gl, g + 2 * gl + gr, cll, clr, c)) { hit = s_; hit_k = CL_CACHE; hit_ll = cll; hit_lr = clr; }
						}
				}
				if (hit >= 0) {
					const int cll = hit_ll, clr = hit_lr;
					if (hit_k < B4_KLDS) merge(v, s_str[hit_k][0], s_str[hit_k][1], s_str[hit_k][2], s_str[hit_k][3], cll, clr, c);
					else { uint32_t *g = gstr(hit); merge(v, g, g + gl, g + 2 * gl, g + 2 * gl + gr, cll, clr, c); }
					__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
					if (hit_k < CL_CACHE) {
						if (lane == 0) {
							BinMeta &mt = s_meta[hit_k];
							if (cll <= v.ll) { mt.ll = v.ll; if (!left_clipped) mt.cig_ev = e; } // aa == RIGHT_CLIPPED (also when the lengths are equal)
							if (clr < v.lr) { mt.lr = v.lr; if (left_clipped) mt.cig_ev = e; }    // aa == LEFT_CLIPPED
							mt.support += 1;
						}
					} else {
						if (cll <= v.ll) { a.c_ll[hit] = v.ll; if (!left_clipped) a.c_cig_ev[hit] = e; }
						if (clr < v.lr) { a.c_lr[hit] = v.lr; if (left_clipped) a.c_cig_ev[hit] = e; }
						a.support[hit] += 1; // every lane stores the same value; each lane later reads back what it stored
					}
				} else {
					// ---- new cluster at this event's slot (clip_reads.cpp:276-281) ----
					const int k = nclu < CL_CACHE ? nclu : CL_CACHE;
					if (k < B4_KLDS) create(v, s_str[k][0], s_str[k][1], s_str[k][2], s_str[k][3]);
					else { uint32_t *g = gstr(jj); create(v, g, g + gl, g + 2 * gl, g + 2 * gl + gr); }
					a.c_qmiss[jj] = v.qmiss ? 1 : 0;
					if (nclu < CL_CACHE) {
						if (lane == 0) { s_slot[nclu] = (int32_t)t; BinMeta &mt = s_meta[nclu]; mt.ll = v.ll; mt.lr = v.lr; mt.support = 1; mt.cig_ev = e; }
					} else { a.c_ll[jj] = v.ll; a.c_lr[jj] = v.lr; a.c_cig_ev[jj] = e; a.support[jj] = 1; }
					++nclu;
				}
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				if (t + 1 < nb) v = extract(L1);
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				L1 = L2; L2 = L3;
			}
			// ---- the bin's results to global memory: lengths / support / CIGAR carrier of the cached clusters, strings of the LDS-resident ones ----
			const int kc = nclu < CL_CACHE ? nclu : CL_CACHE;
			for (int k = lane; k < kc; k += WAVE) {
				const int64_t slot = j0 + s_slot[k];
				const BinMeta mt = s_meta[k];
				a.c_ll[slot] = mt.ll; a.c_lr[slot] = mt.lr; a.c_cig_ev[slot] = mt.cig_ev; a.support[slot] = mt.support;
			}
			// (a cluster that stayed alone IS its event: the pack kernels cut it out of the read like the cluster of a single-event bin - no strings)
			const int ks = nclu < B4_KLDS ? nclu : B4_KLDS;
			for (int k = 0; k < ks; ++k) {
				if (s_meta[k].support == 1) continue;
				uint32_t *g = gstr(j0 + s_slot[k]);
				const int ll = s_meta[k].ll, lr = s_meta[k].lr;
				if (4 * lane < ll) { g[lane] = s_str[k][0][lane]; g[gl + lane] = s_str[k][1][lane]; }
				if (4 * lane < lr) { g[2 * gl + lane] = s_str[k][2][lane]; g[2 * gl + gr + lane] = s_str[k][3][lane]; }
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------------
// cluster table packing
// ---------------------------------------------------------------------------------------------------------------------

struct PackArgs {
	ClusterArgs c;
	// per sorted slot: what the scans turn into dense positions
	uint64_t *slot_cnt;       // [E] cluster at this slot ? 1 | CIGAR operations << 32 : 0          -> exclusive scan: cluster index | cig_off << 32
	uint64_t *slot_bytes;     // [E] bytes of the cluster's string block (0 when there is none)      -> exclusive scan: str_off
	// dense outputs [n_clusters]: written by the pack kernels themselves (a slot's lanes know everything about its cluster)
	int32_t *tid, *pos;
	uint8_t *side;
	int32_t *support, *ll, *lr;
	uint8_t *qmiss;
	int32_t *ncig;
	uint64_t *str_off, *cig_off;
	int packed;               // 1: sequences as 4-bit codes (ssv_cluster_table.seq_packed)
	int format3, base_bits;   // the compact layout (table3_kernels.h): one base stream + one quality stream per block
	int qual_bits;            // 8: quality characters; 1, 2, 3, 4: indices into the table's quality alphabet (format 3 with qual_group > 1: bits per GROUP)
	int qual_group;           // format 3: qualities per group (1: every quality its own qual_bits; k > 1: k alphabet indices as digits of one number of qual_bits bits, radix qual_radix)
	int qual_radix;           // the alphabet's size (the base of a group's number)
	uint32_t qual_fill;       // phred value of alphabet index 0 in all four bytes (what the look-ups see behind a stream's end)
	uint32_t tri_mul;         // three qualities a group: the multiplier that hashes a triple of phred bytes into the table of k_pack3_direct (qual_dword3h)
	const uint8_t *qlut;      // [256] phred -> index (0xff: not in the alphabet), when qual_bits < 8
	uint32_t *qual_seen;      // [8] bit set of the phred values met while packing (TRACK launches: which alphabet the table really needs)
	int *lut_miss;            // raised when a quality outside the alphabet is met (qual_bits < 8)
	// slots whose cluster needs the bytewise path of the packed kernel (multi-event bins, reads longer than PACK_MAX_LQ), listed by
	// k_cluster_meta: they are packed by a launch of their own, so that a wavefront of the main launch never runs the long path for one of
	// its four clusters
	uint32_t *slow_list;
	unsigned int *slow_count;
};

// bits of a format-3 quality stream of n qualities: qual_bits each, or per group of `group` qualities
__host__ __device__ __forceinline__ uint64_t qual_stream_bits(uint64_t n, uint64_t bits, uint64_t group) { return group > 1 ? ((n + group - 1) / group) * bits : n * bits; }

__host__ __device__ __forceinline__ uint64_t table_block_bytes(uint64_t L, uint64_t R, int packed, uint64_t W)
{
	return ((packed ? (L + 1) / 2 + (L * W + 7) / 8 + (R + 1) / 2 + (R * W + 7) / 8 : 2 * (L + R)) + 3ull) & ~3ull; // blocks start 4-byte aligned
}

// per sorted slot: is there a cluster, how many bytes / CIGAR operations does it put into the table; single-event clusters of reads too
// long for the dword path are listed (the slots of multi-event bins, the other users of the bytewise path, are listed already: mlist)
__global__ __launch_bounds__(BLOCK) void k_cluster_meta(PackArgs p)
{
	const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t cnt = 0, bytes = 0;
	bool lng = false;
	if (j < p.c.E && p.c.support[j] > 0) {
		const bool single = !p.c.mflag[j] || p.c.support[j] == 1; // (a cluster of a multi-event bin that nothing joined is its event, like a single-event bin's)
		const uint32_t e = single ? (uint32_t)j : p.c.c_cig_ev[j];
		const uint4 *ep = reinterpret_cast<const uint4 *>(p.c.ev + e);
		const uint4 eb = ep[1], ec = ep[2];
		const int ll = single ? (int)eb.w : p.c.c_ll[j], lr = single ? (int)ec.x : p.c.c_lr[j];
		cnt = 1ull | ((uint64_t)ec.z << 32);
		bytes = p.format3 ? 4ull * (((uint64_t)(ll + lr) * (uint64_t)p.base_bits + 31) / 32 + (qual_stream_bits((uint64_t)(ll + lr), (uint64_t)p.qual_bits, (uint64_t)p.qual_group) + 31) / 32)
		                  : table_block_bytes((uint64_t)ll, (uint64_t)lr, p.packed, (uint64_t)p.qual_bits);
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
	if (j < p.c.E) { p.slot_cnt[j] = cnt; p.slot_bytes[j] = bytes; }
}

// which phred values occur among the qualities of the first events (BAM order): the first guess of the table's quality alphabet
__global__ __launch_bounds__(BLOCK) void k_qual_sample(const ClipEvent *__restrict__ ev, const uint32_t *__restrict__ ev_idx, int64_t n, uint32_t *__restrict__ seen)
{
	__shared__ uint32_t s_seen[8];
	if (threadIdx.x < 8) s_seen[threadIdx.x] = 0;
	__syncthreads();
	const int64_t o = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave_id();
	if (o < n) {
		const uint4 *ep = reinterpret_cast<const uint4 *>(ev + ev_idx[o]);
		const uint4 ea = ep[0], ec = ep[2];
		const int lq = (int)ec.y;
		const uint8_t *qp = reinterpret_cast<const uint8_t *>((uintptr_t)((uint64_t)ea.z | ((uint64_t)ea.w << 32))) + (lq + 1) / 2;
		if (lq > 0 && qp[0] != 0xff)
			for (int i = lane_id(); i < lq; i += WAVE) { const uint32_t q = qp[i]; atomicOr(&s_seen[q >> 5], 1u << (q & 31)); }
	}
	__syncthreads();
	if (threadIdx.x < 8 && s_seen[threadIdx.x]) atomicOr(&seen[threadIdx.x], s_seen[threadIdx.x]);
}

// the inverse as a table (ASCII -> 4-bit code, anything else N): a compare chain costs 32 instructions per character, and a wavefront with
// one cluster of a multi-event bin runs its path for all its lanes
__device__ __constant__ uint8_t NT16_CODE_OF[256] = {15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 0, 15, 15, 15, 1, 14, 2, 13, 15, 15, 4, 11, 15, 15, 12, 15, 3, 15, 15, 15, 15, 5, 6, 8, 15, 7, 9, 15, 10, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15};

// What a table row needs of the cluster at sorted slot j: its per-slot words and the carrying event's line.
struct SlotCluster {
	bool cluster, single;
	uint32_t c;             // dense cluster index
	uint64_t str_off, cig_off, key;
	int support, ll, lr, lq, begin, ncg;
	uint64_t src, cig_ptr;
	uint32_t cg0, cg1, cg2, cg3, cg4;
	uint8_t qmiss_multi;
};

// (dense == false: everything but the three scanned words c, cig_off, str_off - the one-pass kernel works those out itself)
template <bool dense = true>
__device__ __forceinline__ SlotCluster slot_cluster_load(const PackArgs &p, int64_t j)
{
	SlotCluster s;
	s.support = p.c.support[j];
	s.cluster = s.support > 0;
	s.single = !p.c.mflag[j] || s.support == 1;
	if (dense) { const uint64_t sc = p.slot_cnt[j]; s.c = (uint32_t)sc; s.cig_off = sc >> 32; s.str_off = p.slot_bytes[j]; }
	else { s.c = 0; s.cig_off = 0; s.str_off = 0; }
	s.key = p.c.skey[j];
	const uint32_t e = s.single ? (uint32_t)j : p.c.c_cig_ev[j];
	const uint4 *ep = reinterpret_cast<const uint4 *>(p.c.ev + (s.cluster ? e : 0u));
	const uint4 ea = ep[0], eb = ep[1], ec = ep[2], ed = ep[3];
	s.src = (uint64_t)ea.z | ((uint64_t)ea.w << 32); s.cig_ptr = (uint64_t)eb.x | ((uint64_t)eb.y << 32);
	s.begin = s.single ? (int)eb.z : 0;
	s.ll = s.single ? (int)eb.w : p.c.c_ll[j]; s.lr = s.single ? (int)ec.x : p.c.c_lr[j];
	s.lq = s.single ? (int)ec.y : -1; s.ncg = (int)ec.z;
	s.cg0 = ec.w; s.cg1 = ed.x; s.cg2 = ed.y; s.cg3 = ed.z; s.cg4 = ed.w;
	s.qmiss_multi = s.single ? 0 : p.c.c_qmiss[j];
	return s;
}

// What the string kernels need of a cluster, dense by cluster index: 32 bytes, read coalesced.  lq < 0: a cluster of a multi-event bin
// (consensus storage; `begin` then holds its sorted slot).
struct PackDesc {
	uint64_t src, str_off;
	int32_t lq, ll, lr, begin;
};
static_assert(sizeof(PackDesc) == 32, "PackDesc is half a line");

// one thread per sorted slot: the row's fixed columns, its CIGAR, and the descriptor for the string kernels (the `qual_missing` column
// of single-event clusters is theirs: it takes the read's first quality byte)
__global__ __launch_bounds__(BLOCK) void k_cluster_cols(PackArgs p, PackDesc *__restrict__ desc, uint32_t *__restrict__ out_cig)
{
	const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= p.c.E) return;
	const SlotCluster s = slot_cluster_load(p, j);
	if (!s.cluster) return;
	const uint32_t c = s.c;
	p.tid[c] = (int32_t)(s.key >> 33);
	p.pos[c] = (int32_t)(uint32_t)s.key;
	p.side[c] = ((s.key >> 32) & 1ull) ? '3' : '5';
	p.support[c] = s.support;
	p.ll[c] = s.ll; p.lr[c] = s.lr;
	p.ncig[c] = s.ncg;
	p.str_off[c] = s.str_off;
	p.cig_off[c] = s.cig_off;
	if (!s.single) p.qmiss[c] = s.qmiss_multi;
	uint32_t *dc = out_cig + s.cig_off;
	if (s.ncg <= 5) {
		if (s.ncg > 0) dc[0] = s.cg0;
		if (s.ncg > 1) dc[1] = s.cg1;
		if (s.ncg > 2) dc[2] = s.cg2;
		if (s.ncg > 3) dc[3] = s.cg3;
		if (s.ncg > 4) dc[4] = s.cg4;
	} else {
		const uint32_t *src = reinterpret_cast<const uint32_t *>((uintptr_t)s.cig_ptr);
		for (int i = 0; i < s.ncg; ++i) dc[i] = src[i];
	}
	uint4 *dd = reinterpret_cast<uint4 *>(desc + c);
	dd[0] = make_uint4((uint32_t)s.src, (uint32_t)(s.src >> 32), (uint32_t)s.str_off, (uint32_t)(s.str_off >> 32));
	dd[1] = make_uint4((uint32_t)s.lq, (uint32_t)s.ll, (uint32_t)s.lr, s.single ? (uint32_t)s.begin : (uint32_t)j);
}

// 16 lanes per sorted slot, four slots per wavefront: the strings of the slot's cluster (if it has one) into the dense blob.  Every lane
// assembles whole output dwords (a cluster's block starts 4-byte aligned and is zero padded to a multiple of 4 bytes):
// [seq_left | qual_left | seq_right | qual_right].  Single-event clusters (97 %) are decoded from the read's packed bases / qualities
// where the batch (or the context's copy) holds them (GetSeq, clip_reads.cpp:286-306): the group expands the read once into LDS with dword
// loads (8 bases per packed dword, 4 qualities per dword via alignbyte and a packed +33) and then composes the output dwords from LDS
// bytes.  Clusters of multi-event bins come from their consensus storage (left part un-reversed); reads longer than PACK_MAX_LQ take the
// per-byte path.  This is the ASCII layout (ssv_clip_table_format 0, the C ABI's default); the packed layouts have their own kernels below.
__global__ __launch_bounds__(BLOCK) void k_cluster_pack_ascii(PackArgs p, uint8_t *__restrict__ out_str)
{
	__shared__ uint32_t s_seq[GROUPS_PER_BLOCK][PACK_MAX_LQ / 4 + 4];
	__shared__ uint32_t s_qual[GROUPS_PER_BLOCK][PACK_MAX_LQ / 4 + 4];
	const int grp = (int)(threadIdx.x / GROUP);
	const int gl = (int)(threadIdx.x % GROUP);
	const int64_t j = (int64_t)blockIdx.x * GROUPS_PER_BLOCK + grp;
	SlotCluster sc = {};
	if (j < p.c.E) sc = slot_cluster_load(p, j);
	const bool active = sc.cluster;
	const int ll = sc.ll, lr = sc.lr, lq = sc.lq, begin = sc.begin;
	const bool single = active && sc.single;
	bool staged = false, qmiss = false;
	const uint8_t *sp = nullptr;
	if (single) {
		sp = reinterpret_cast<const uint8_t *>((uintptr_t)sc.src);
		qmiss = lq > 0 && sp[(lq + 1) / 2] == 0xff;
		staged = lq <= PACK_MAX_LQ;
	} else if (active) qmiss = sc.qmiss_multi != 0;
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
	if (single && gl == 0) p.qmiss[sc.c] = qmiss ? 1 : 0;
	// one block = four pieces: sequence / quality of the left part, sequence / quality of the right part.  A piece position maps to a
	// character through seq_at / qual_at below (three sources: the LDS stage, the read's packed bytes, the consensus storage).
	const int total = 2 * (ll + lr);
	uint32_t *d = reinterpret_cast<uint32_t *>(out_str + sc.str_off);
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
}

// ---- the LDS-staged composition of a packed block (round 1-2: table formats 1 / 2, removed in round 5; what k_pack3_stream, table3_kernels.h, still builds on) ----
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
// 16 lanes per cluster.  The kernel is bound by memory LATENCY, not by bytes or instructions (PMC: waves parked 68 % of their cycles,
// VALU busy 11 %, at full occupancy): per cluster it is descriptor -> read bytes -> LDS -> output, two dependent round trips to HBM.  So
// the grid is persistent and every group of lanes runs a software pipeline over its clusters: while cluster t is composed, the bytes of
// cluster t + 1 and the descriptor of cluster t + 2 are already on their way.
constexpr int PACK_RAW_DWORDS = (PACK_MAX_LQ / 2 + PACK_MAX_LQ) / 4 + 3; // entry of a PACK_MAX_LQ read + misalignment + read-ahead
constexpr int PACK_LDS_DWORDS = (PACK_MAX_LQ / 8 + 2) + (PACK_MAX_LQ / 4 + 2) + 5 + 1; // two sequence + two quality pieces of left_len + right_len <= PACK_MAX_LQ (W = 8 worst case) + guards
constexpr int PACK_RAW_PER_LANE = PACK_RAW_DWORDS / GROUP + 1;

struct PackDescR { uint64_t src, str_off; int lq, ll, lr, begin; };

__device__ __forceinline__ PackDescR pack_desc_load(const PackDesc *desc, int64_t c, int64_t nc)
{
	PackDescR d;
	d.src = 0; d.str_off = 0; d.lq = -1; d.ll = 0; d.lr = 0; d.begin = 0;
	if (c < nc) {
		const uint4 *p = reinterpret_cast<const uint4 *>(desc + c);
		const uint4 a = p[0], b = p[1];
		d.src = (uint64_t)a.x | ((uint64_t)a.y << 32); d.str_off = (uint64_t)a.z | ((uint64_t)a.w << 32);
		d.lq = (int)b.x; d.ll = (int)b.y; d.lr = (int)b.z; d.begin = (int)b.w;
	}
	return d;
}

__device__ __forceinline__ bool pack_desc_fast(const PackDescR &d) { return d.lq >= 0 && d.lq <= PACK_MAX_LQ; }

// the read's entry from the dword-aligned address below it: this lane's share of the loads, all issued at once
__device__ __forceinline__ void pack_entry_load(const PackDescR &d, int gl, uint32_t (&r)[PACK_RAW_PER_LANE])
{
	const int mis = (int)(d.src & 3ull);
	const uint32_t *g4 = reinterpret_cast<const uint32_t *>((uintptr_t)(d.src - (uint64_t)mis));
	const int nraw = pack_desc_fast(d) ? (mis + (d.lq + 1) / 2 + d.lq + 3) / 4 + 1 : 0; // + one dword of read-ahead for the unaligned windows (8 bytes of slack behind seqqual, seeksv_hip.h)
#pragma unroll
	for (int u = 0; u < PACK_RAW_PER_LANE; ++u) { const int i = gl + GROUP * u; r[u] = i < nraw && i < PACK_RAW_DWORDS ? g4[i] : 0u; }
}

} // namespace ssv
