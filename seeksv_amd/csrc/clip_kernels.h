// clip_kernels.h - getclip on the GPU: CIGAR-end scan -> ordered clip events -> (contig, side, pos) bins
// -> greedy consensus clustering, one wavefront per bin.
//
// Reference behaviour being reproduced (file:line in /root/reference/seeksv):
//   record routing + contig-switch rule   clip_reads.h:410-440
//   GetSClipReads / GenerateCigar         clip_reads.cpp:112-192, 309-329
//   GetSeq                                clip_reads.cpp:286-306
//   InsertSeq / CompareString* / ChangeSeqAndQual   clip_reads.cpp:260-283, 194-217, 57-108
#pragma once

#include "common.h"

namespace ssv {

// device view of one batch (all pointers in HBM)
struct DevBatch {
	int64_t n;
	const int32_t *tid, *pos;
	const uint16_t *flag;
	const uint8_t *mapq;
	const uint16_t *n_cigar;
	const int32_t *l_qseq, *mtid, *mpos, *isize;
	const uint32_t *cigar_off, *cigar;
	const uint8_t *xc;
	const uint64_t *seq_off;
	const uint8_t *seqqual;
	int32_t max_ref_span;
};

constexpr int CS_ITEMS = 4;                           // getsv scan: records per lane per sub-tile (two 16-byte loads per lane)
constexpr int CS_SUB = 4;                             // sub-tiles per tile
constexpr int CS_TILE = BLOCK * CS_ITEMS * CS_SUB;    // 4096 records per workgroup iteration, one barrier each
constexpr int CC_ITEMS = 8;                           // clip scan: 8 x u16 = one 16-byte load per lane per sub-tile
constexpr int CC_TILE = BLOCK * CC_ITEMS * CS_SUB;    // 8192 records per workgroup iteration
constexpr int CS_MAX_BLOCKS = 8192;                   // upper bound of the persistent grid (private staging regions are sized by the actual grid)

// one clip event as produced by the filter kernel (two stash slots per candidate record)
struct StagedEvent {
	uint64_t key;    // tid << 33 | side << 32 | pos1   (side 0 = '5' / breakpoint2read_l, 1 = '3' / breakpoint2read_r)
	uint32_t rec;    // record index inside the batch
	uint32_t src_cig; // the record's cigar_off
	int32_t begin;   // first query base of seq_left  (GetSeq's begin_pos)
	int32_t ll, lr;  // |seq_left|, |seq_right|
	int32_t lq;      // l_qseq
	uint32_t ncig;
	uint32_t pad;
	uint64_t src_seq; // the record's seq_off
};

struct ClipCounters {
	unsigned long long n_cand;       // candidates of the batch (total of the tile-count scan)
	unsigned long long n_new;        // events of the batch (total of the per-candidate count scan)
	unsigned long long max_key;      // filled by k_event_max (grid-level reduction, a few hundred atomics)
	unsigned long long seq_total;    // running totals written by the offset scans
	unsigned long long cig_total;
	int max_ll, max_lr;
	int overflow;
	int pad;
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

// Decide the 0/1/2 events of record i.  Only called for records whose first or last op is 'S' (about 1 % of a WGS BAM), so everything
// it touches beyond the CIGAR ends is a lazy, sparse load: one 64-byte sector per column.  The kernel is bound by the number of sectors it
// touches, not by their latency (issuing all loads before the first test made it slower), so the chain leaves as early as GetSClipReads.
__device__ __forceinline__ int clip_events_of(const ClipFilterArgs &a, int64_t i, int nc, uint32_t c0, uint32_t cl, const uint32_t *cig, uint32_t cig_off, uint64_t soff, StagedEvent ev[2])
{
	const DevBatch &b = a.b;
	const int op1 = (int)(c0 & 15u), op2 = (int)(cl & 15u);
	const int flag = b.flag[i];
	if (flag & (F_UNMAP | F_MUNMAP)) return 0;          // unmapped-pair side channel (host), clip_reads.h:415
	const int tid = b.tid[i];
	// contig-switch rule: processed only if tid equals the tid of the previous mapped-pair record
	int prev_tid = *a.last_tid_in;
	for (int64_t j = i - 1; j >= 0; --j) {
		if (!(b.flag[j] & (F_UNMAP | F_MUNMAP))) { prev_tid = b.tid[j]; break; }
	}
	if (tid != prev_tid || tid < 0) return 0;
	if (op1 == C_H || op2 == C_H || (flag & F_DUP) || (int)b.mapq[i] < a.min_mapq) return 0; // clip_reads.cpp:118
	const int xc = b.xc ? b.xc[i] : 0;
	const int lq = b.l_qseq[i];
	const int pos0 = b.pos[i];
	int ref_len = 0;                                    // only right-clip events need it
	if (op2 == C_S) {
		ref_len = ref_advance(c0) + ref_advance(cl);    // nc >= 2: first and last op are distinct
		for (int k = 1; k < nc - 1; ++k) ref_len += ref_advance(cig[k]);
	}
	bool s1 = op1 == C_S, s2 = op2 == C_S;
	int n = 0;
	uint64_t tkey = (uint64_t)(uint32_t)tid << 33;
	if (s1 != s2) {
		if (xc != 0 && !a.save_low_quality) return 0;   // clip_reads.cpp:129
		if (s1) {
			int ll = (int)(c0 >> 4), lr = lq - ll;
			if (lr < 0) return 0;
			ev[0].key = tkey | (uint32_t)(pos0 + 1); ev[0].begin = 0; ev[0].ll = ll; ev[0].lr = lr; n = 1;
		} else {
			int lr = (int)(cl >> 4), ll = lq - lr;
			if (ll < 0) return 0;
			ev[0].key = tkey | (1ull << 32) | (uint32_t)(pos0 + ref_len); ev[0].begin = 0; ev[0].ll = ll; ev[0].lr = lr; n = 1;
		}
	} else {
		int ll = (int)(c0 >> 4), rc = (int)(cl >> 4), mid = lq - ll - rc;
		if (mid < 0) return 0;
		bool do_l = true, do_r = true;
		if (xc != 0 && !a.save_low_quality) { if (!(flag & F_REV)) do_r = false; else do_l = false; } // clip_reads.cpp:160-175
		if (do_l) { ev[n].key = tkey | (uint32_t)(pos0 + 1); ev[n].begin = 0; ev[n].ll = ll; ev[n].lr = mid; ++n; }
		if (do_r) { ev[n].key = tkey | (1ull << 32) | (uint32_t)(pos0 + ref_len); ev[n].begin = ll; ev[n].ll = mid; ev[n].lr = rc; ++n; }
	}
	if (a.use_ownership) {
		int m = 0;
		for (int k = 0; k < n; ++k) {
			long long kk = ((long long)tid << 32) | (long long)(uint32_t)ev[k].key;
			if (kk >= a.own_lo && kk < a.own_hi) { if (m != k) ev[m] = ev[k]; ++m; }
		}
		n = m;
	}
	for (int k = 0; k < n; ++k) { ev[k].rec = (uint32_t)i; ev[k].lq = lq; ev[k].ncig = (uint32_t)nc; ev[k].pad = 0; ev[k].src_cig = cig_off; ev[k].src_seq = soff; }
	return n;
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
// Their CIGAR ends are looked at by k_clip_filter, one thread per candidate.  Persistent workgroups, private staging, no atomics.
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

// K1b clip_filter: one thread per candidate record (n_cigar >= 2) looks at the CIGAR ends and, for soft-clipped ones, runs GetSClipReads' predicate chain (flag, contig-switch rule, MAPQ, DUP, XC,
// hard clips) and leaves its 0, 1 or 2 events in the candidate's two stash slots.
__global__ __launch_bounds__(BLOCK) void k_clip_filter(ClipFilterArgs a, const uint32_t *__restrict__ cand, int64_t n_cand, StagedEvent *__restrict__ stash, uint32_t *__restrict__ cnt)
{
	const int64_t c = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	StagedEvent ev[2];
	int n = 0;
	if (c < n_cand) {
		const int64_t i = cand[c];
		// two thirds of the candidates are reads with an indel: no soft clip, so the batcher shipped no bases for them (the contract of
		// seq_off) and they cannot become events - one load settles them instead of three
		const uint64_t soff = a.b.seq_off[i];
		if (soff != ~0ull) {
			const int nc = a.b.n_cigar[i];
			const uint32_t off = a.b.cigar_off[i];
			const uint32_t c0 = a.b.cigar[off], cl = a.b.cigar[off + nc - 1];
			if ((c0 & 15u) == C_S || (cl & 15u) == C_S) n = clip_events_of(a, i, nc, c0, cl, a.b.cigar + off, off, soff, ev);
		}
		cnt[c] = (uint32_t)n;
	}
	// the wavefront's events side by side (two slots per candidate are reserved, the wave fills its 128 from the front): a third of
	// the candidates emit, and 48-byte stores scattered at a 96-byte stride cost four times their bytes in partial-line writes
	const int ex = wave_inclusive_sum(n) - n;
	StagedEvent *dst = stash + 2 * (c - lane_id()) + ex;
	for (int e = 0; e < n; ++e) dst[e] = ev[e];
}

// tid of the last mapped-pair record of the batch -> *last_tid (unchanged when there is none)
__global__ __launch_bounds__(BLOCK) void k_last_tid(DevBatch b, int *last_tid)
{
	__shared__ long long best;
	if (threadIdx.x == 0) best = -1;
	__syncthreads();
	for (int64_t hi = b.n; hi > 0; hi -= BLOCK) {
		int64_t i = hi - 1 - threadIdx.x;
		if (i >= 0 && !(b.flag[i] & (F_UNMAP | F_MUNMAP))) atomicMax(&best, (long long)i);
		__syncthreads();
		if (best >= 0) break;
	}
	if (threadIdx.x == 0 && best >= 0) *last_tid = b.tid[best];
}

// final, ordered event arrays (context owned, all batches)
struct EventArrays {
	uint64_t *key;
	int32_t *begin, *ll, *lr, *lq;
	uint32_t *ncig;
	uint32_t *seq_bytes;     // packed bases + qualities
	uint64_t *seq_off;       // into seq_blob
	uint64_t *cig_off;       // into cig_blob
	uint64_t *src_seq;       // scratch: offset in the batch's seqqual
	uint32_t *src_cig;       // scratch: offset in the batch's cigar
	uint8_t *qmiss;          // 1: the event's read carries no qualities (first quality byte 0xff; the row prints "*") - set by k_clip_gather
};

// candidate c's events -> final, BAM-ordered position ev_base + ev_off[c] + e
__global__ __launch_bounds__(BLOCK) void k_clip_place(const StagedEvent *__restrict__ stash, const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ ev_off, int64_t n_cand,
                                                      EventArrays ev, int64_t ev_base)
{
	const int64_t c = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	const int n = c < n_cand ? (int)cnt[c] : 0;
	const int ex = wave_inclusive_sum(n) - n; // same thread-to-candidate mapping as k_clip_filter: the wave's events are packed from its first slot
	const StagedEvent *src = stash + 2 * (c - lane_id()) + ex;
	for (int k = 0; k < n; ++k) {
		StagedEvent x = src[k];
		int64_t e = ev_base + ev_off[c] + k;
		ev.key[e] = x.key; ev.begin[e] = x.begin; ev.ll[e] = x.ll; ev.lr[e] = x.lr; ev.lq[e] = x.lq; ev.ncig[e] = x.ncig;
		ev.seq_bytes[e] = ((uint32_t)((x.lq + 1) / 2 + x.lq) + 3u) & ~3u; // entries of the context blob are 4-byte aligned
		ev.src_seq[e] = x.src_seq;
		ev.src_cig[e] = x.src_cig;
	}
}

// largest key / slice lengths of the batch's events: grid-stride partial maxima, one atomic per wave
__global__ __launch_bounds__(BLOCK) void k_event_max(EventArrays ev, int64_t ev_base, int64_t n_new, ClipCounters *ctr)
{
	unsigned long long mk = 0;
	int mll = 0, mlr = 0;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_new; i += (int64_t)gridDim.x * blockDim.x) {
		int64_t e = ev_base + i;
		unsigned long long k = ev.key[e];
		mk = k > mk ? k : mk;
		mll = ev.ll[e] > mll ? ev.ll[e] : mll;
		mlr = ev.lr[e] > mlr ? ev.lr[e] : mlr;
	}
	mk = wave_max(mk); mll = wave_max(mll); mlr = wave_max(mlr);
	if (lane_id() == 0) { atomicMax(&ctr->max_key, mk); atomicMax(&ctr->max_ll, mll); atomicMax(&ctr->max_lr, mlr); }
}

// K2 clip_gather: one wavefront per event copies its packed bases, qualities and CIGAR into context-owned blobs so that
// the batch buffers can be recycled.  Destination entries start 4-byte aligned (sizes are padded when the offsets are scanned);
// the source may start anywhere, so every lane assembles one aligned output dword from two aligned source dwords.
constexpr int GROUP = 16;                      // lanes that cooperate on one event / cluster in the gather and pack kernels
constexpr int GROUPS_PER_WAVE = WAVE / GROUP;  // 4 items in flight per wavefront: the per-item metadata loads overlap
constexpr int GROUPS_PER_BLOCK = BLOCK / GROUP;

__global__ __launch_bounds__(BLOCK) void k_clip_gather(DevBatch b, EventArrays ev, int64_t ev_base, int64_t n_new, uint8_t *__restrict__ seq_blob, uint32_t *__restrict__ cig_blob,
                                                       uint8_t *__restrict__ qual_present)
{
	// which base-quality values occur among the events (the cluster table can carry qualities as indices into that alphabet, see
	// ssv_clip_table_format): byte flags in LDS while the quality bytes pass through the registers anyway, one plain store per
	// occurring value and workgroup at the end.  Events without qualities (first quality byte 0xff, the row prints "*") do not count.
	__shared__ uint8_t s_present[BLOCK];
	s_present[threadIdx.x] = 0;
	__syncthreads();
	const int64_t w = (int64_t)blockIdx.x * GROUPS_PER_BLOCK + (threadIdx.x / GROUP);
	if (w < n_new) {
		const uint32_t gl = threadIdx.x % GROUP;
		int64_t e = ev_base + w;
		const uint8_t *src = b.seqqual + ev.src_seq[e];
		uint32_t *dst = reinterpret_cast<uint32_t *>(seq_blob + ev.seq_off[e]);
		const uint32_t nb = ev.seq_bytes[e];          // padded to a multiple of 4
		const uint32_t lq = (uint32_t)ev.lq[e];
		const uint32_t q0 = (lq + 1) / 2, q1 = q0 + lq; // the quality bytes of the entry
		const bool has_qual = lq > 0 && src[q0] != 0xff;
		const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 3u);
		const uint32_t *s4 = reinterpret_cast<const uint32_t *>(src - mis);
		const uint32_t *cs = b.cigar + ev.src_cig[e];
		uint32_t *cd = cig_blob + ev.cig_off[e];
		const uint32_t nc = ev.ncig[e];
		auto put = [&](uint32_t k, uint32_t lo, uint32_t hi) {
			const uint32_t word = mis ? __builtin_amdgcn_alignbyte(hi, lo, mis) : lo;
			dst[k] = word;
			if (has_qual && 4 * k + 4 > q0 && 4 * k < q1) {
#pragma unroll
				for (uint32_t j = 0; j < 4; ++j)
					if (4 * k + j >= q0 && 4 * k + j < q1) s_present[(word >> (8 * j)) & 0xffu] = 1;
			}
		};
		// every load of the event issued before the first store: the entry of a read of up to 320 bases (128 dwords) is covered by the
		// unrolled batch (128 dwords: 320 bases), so an event costs one memory round trip after its metadata; longer reads and CIGARs continue in the loops
		constexpr int BATCH = 8;
		uint32_t lo[BATCH], hi[BATCH];
#pragma unroll
		for (int u = 0; u < BATCH; ++u) {
			const uint32_t k = gl + GROUP * u;
			const bool in = k < nb / 4;
			lo[u] = in ? s4[k] : 0u;
			hi[u] = in && mis ? s4[k + 1] : 0u;       // may read up to 7 bytes past the entry: see the slack rule in seeksv_hip.h
		}
		const uint32_t c_first = gl < nc ? cs[gl] : 0u;
#pragma unroll
		for (int u = 0; u < BATCH; ++u) {
			const uint32_t k = gl + GROUP * u;
			if (k < nb / 4) put(k, lo[u], hi[u]);
		}
		if (gl < nc) cd[gl] = c_first;
		if (gl == 0) ev.qmiss[e] = (lq > 0 && !has_qual) ? 1 : 0; // (last: a store right after the test would make the loads above wait for it)
		for (uint32_t k = gl + GROUP * BATCH; k < nb / 4; k += GROUP) put(k, s4[k], mis ? s4[k + 1] : 0u);
		for (uint32_t k = gl + GROUP; k < nc; k += GROUP) cd[k] = cs[k];
	}
	__syncthreads();
	if (s_present[threadIdx.x]) qual_present[threadIdx.x] = 1;
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
	EventArrays ev;
	const uint8_t *seq_blob;
	double match_rate;
	// per sorted slot outputs
	int32_t *support;      // [E], zero initialised; > 0 marks a cluster created by the event at this slot
	int32_t *c_ll, *c_lr;  // [E]
	uint32_t *c_cig_ev;    // [E] event whose CIGAR the cluster carries
	uint8_t *c_qmiss;      // [E]
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
// the event itself, recorded here; only bins with several events go through k_cluster_bins and get string storage.
__global__ void k_bin_mark(const uint64_t *__restrict__ skey, const uint32_t *__restrict__ perm, int64_t E, EventArrays ev, const uint8_t *__restrict__ seq_blob,
                           uint32_t *__restrict__ mflag, int32_t *__restrict__ support, int32_t *__restrict__ c_ll, int32_t *__restrict__ c_lr,
                           uint32_t *__restrict__ c_cig_ev, uint8_t *__restrict__ c_qmiss)
{
	int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= E) return;
	const uint64_t k = skey[j];
	const bool single = (j == 0 || skey[j - 1] != k) && (j + 1 == E || skey[j + 1] != k);
	mflag[j] = single ? 0u : 1u;
	if (single) {
		const uint32_t e = perm[j];
		support[j] = 1; c_ll[j] = ev.ll[e]; c_lr[j] = ev.lr[e]; c_cig_ev[j] = e;
		c_qmiss[j] = ev.qmiss[e]; // (noted by the gather kernel: looking it up in the blob costs a scattered sector per event)
	} else support[j] = 0;
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
		EventView v;
		const int lq = a.ev.lq[e];
		v.sp = a.seq_blob + a.ev.seq_off[e];
		v.qp = v.sp + (lq + 1) / 2;
		v.begin = a.ev.begin[e]; v.ll = a.ev.ll[e]; v.lr = a.ev.lr[e];
		v.qmiss = a.ev.qmiss[e] != 0; // noted by the gather kernel (reading qp[0] here would be one more memory round trip in the per-read chain)
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

__global__ void k_cluster_flags(const int32_t *__restrict__ support, int64_t E, uint32_t *__restrict__ flag)
{
	int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j < E) flag[j] = support[j] > 0 ? 1u : 0u;
}

constexpr int PACK_MAX_LQ_ = 320; // = PACK_MAX_LQ below (reads up to this length take the LDS-staged paths of the pack kernels)

struct PackArgs {
	ClusterArgs c;
	const uint32_t *flag;     // [E]
	const uint32_t *cidx;     // [E] exclusive scan of flag
	// dense outputs [n_clusters]
	int32_t *tid, *pos;
	uint8_t *side;
	int32_t *support, *ll, *lr;
	uint8_t *qmiss;
	uint32_t *slot;           // dense index -> sorted slot
	uint64_t *str_bytes;      // bytes of the cluster's string block
	int packed;               // 1: sequences as 4-bit codes (ssv_cluster_table.seq_packed)
	int qual_bits;            // 8: quality characters; 1, 2, 4: indices into the table's quality alphabet
	const uint8_t *qlut;      // [256] quality character (phred + 33) -> index, when qual_bits < 8
	uint64_t *ncig64;
	int32_t *ncig;
	// where a cluster's characters come from, resolved once per cluster by k_cluster_pack_meta so that the pack kernel starts with
	// coalesced loads instead of a slot -> event -> offset chain: src_lq >= 0: single-event cluster, the event's entry in seq_blob starts
	// at src_off, its seq_left at base src_begin; src_lq < 0: consensus storage of a multi-event bin
	uint64_t *src_off;
	int32_t *src_begin, *src_lq;
	uint64_t *src_cig;        // where the carrying event's CIGAR starts in cig_blob
	// clusters that need the bytewise path of the packed kernel (multi-event bins, reads longer than PACK_MAX_LQ), listed by the meta kernel:
	// they are packed by a launch of their own, so that a wavefront of the main launch never runs the long path for one of its four clusters
	uint32_t *slow_list;
	unsigned int *slow_count;
};

__global__ void k_cluster_pack_meta(PackArgs p)
{
	int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= p.c.E || !p.flag[j]) return;
	uint32_t c = p.cidx[j];
	uint64_t key = p.c.skey[j];
	p.tid[c] = (int32_t)(key >> 33);
	p.pos[c] = (int32_t)(uint32_t)key;
	p.side[c] = ((key >> 32) & 1ull) ? '3' : '5';
	p.support[c] = p.c.support[j];
	int ll = p.c.c_ll[j], lr = p.c.c_lr[j];
	p.ll[c] = ll; p.lr[c] = lr;
	p.qmiss[c] = p.c.c_qmiss[j];
	p.slot[c] = (uint32_t)j;
	const uint64_t L = (uint64_t)ll, R = (uint64_t)lr, W = (uint64_t)p.qual_bits;
	p.str_bytes[c] = ((p.packed ? (L + 1) / 2 + (L * W + 7) / 8 + (R + 1) / 2 + (R * W + 7) / 8 : 2 * (L + R)) + 3ull) & ~3ull; // blocks start 4-byte aligned
	uint32_t nc = p.c.ev.ncig[p.c.c_cig_ev[j]];
	p.ncig[c] = (int32_t)nc;
	p.ncig64[c] = nc;
	if (p.src_off) {
		const uint32_t e = p.c.c_cig_ev[j];
		const bool single = !p.c.mflag[j];
		p.src_off[c] = single ? p.c.ev.seq_off[e] : 0ull;
		p.src_begin[c] = single ? p.c.ev.begin[e] : 0;
		p.src_lq[c] = single ? p.c.ev.lq[e] : -1;
		p.src_cig[c] = p.c.ev.cig_off[e];
		const int lq = single ? p.c.ev.lq[e] : -1;
		if (!(lq >= 0 && lq <= PACK_MAX_LQ_ && ll + lr <= PACK_MAX_LQ_)) p.slow_list[atomicAdd(p.slow_count, 1u)] = c; // a few thousand of millions: no hot spot
	}
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

constexpr int PACK_MAX_LQ = PACK_MAX_LQ_;

// 16 lanes per cluster, four clusters per wavefront: strings and the CIGAR of the carrying event into dense blobs.  Every lane
// assembles whole output dwords (a cluster's block starts 4-byte aligned and is zero padded to a multiple of 4 bytes):
// [seq_left | qual_left | seq_right | qual_right].  Single-event clusters (97 %) are decoded from the event's packed bases /
// qualities (GetSeq, clip_reads.cpp:286-306): the group expands the read once into LDS with dword loads (8 bases per packed dword,
// 4 qualities per dword via alignbyte and a packed +33) and then composes the output dwords from LDS bytes.  Clusters of multi-event
// bins come from their consensus storage (left part un-reversed); reads longer than PACK_MAX_LQ take the per-byte path.
// This is the ASCII layout (ssv_clip_table_format 0, the C ABI's default); the packed layouts have their own kernel below.
__global__ __launch_bounds__(BLOCK) void k_cluster_pack_ascii(PackArgs p, int64_t n_clusters, const uint64_t *__restrict__ str_off, const uint64_t *__restrict__ cig_off,
                                                                const uint32_t *__restrict__ cig_blob, uint8_t *__restrict__ out_str, uint32_t *__restrict__ out_cig)
{
	__shared__ uint32_t s_seq[GROUPS_PER_BLOCK][PACK_MAX_LQ / 4 + 2];
	__shared__ uint32_t s_qual[GROUPS_PER_BLOCK][PACK_MAX_LQ / 4 + 2];
	const int grp = (int)(threadIdx.x / GROUP);
	const int gl = (int)(threadIdx.x % GROUP);
	const int64_t c = (int64_t)blockIdx.x * GROUPS_PER_BLOCK + grp;
	const bool active = c < n_clusters;
	int64_t j = 0;
	int ll = 0, lr = 0, lq = 0, begin = 0;
	uint32_t e = 0;
	bool single = false, staged = false, qmiss = false;
	const uint8_t *sp = nullptr;
	if (active) {
		j = p.slot[c]; ll = p.ll[c]; lr = p.lr[c];
		e = p.c.c_cig_ev[j];
		single = !p.c.mflag[j];
		if (single) {
			lq = p.c.ev.lq[e];
			sp = p.c.seq_blob + p.c.ev.seq_off[e]; // 4-byte aligned
			begin = p.c.ev.begin[e];
			qmiss = lq > 0 && sp[(lq + 1) / 2] == 0xff;
			staged = lq <= PACK_MAX_LQ;
		}
	}
	if (staged) {
		const uint32_t *sp4 = reinterpret_cast<const uint32_t *>(sp);
		const int nseq4 = ((lq + 1) / 2 + 3) / 4; // packed dwords holding the bases
		for (int w = gl; w < nseq4; w += GROUP) {
			const uint32_t pk = sp4[w];
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
			uint32_t hi = mis ? q4[w + 1] : 0u; // stays inside the 4-byte padded entry (or the blob's slack) like k_clip_gather
			uint32_t v = mis ? __builtin_amdgcn_alignbyte(hi, lo, mis) : lo;
			s_qual[grp][w] = qmiss ? 0x2a2a2a2au : v + 0x21212121u; // phred + 33 (qualities <= 93: no carry between bytes); '*' when absent
		}
	}
	__syncthreads();
	if (!active) return;
	// one block = four pieces: sequence / quality of the left part, sequence / quality of the right part.  A piece position maps to a
	// character through seq_at / qual_at below (three sources: the LDS stage, the event's packed read, the consensus storage).
	const int total = 2 * (ll + lr);
	uint32_t *d = reinterpret_cast<uint32_t *>(out_str + str_off[c]);
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
	const uint32_t *src = cig_blob + p.c.ev.cig_off[e];
	uint32_t *dc = out_cig + cig_off[c];
	const int nc = p.ncig[c];
	for (int i = gl; i < nc; i += GROUP) dc[i] = src[i];
}

// ---- packed table (ssv_clip_table_format 1 / 2): sequences as 4-bit codes, qualities W bits each ----
//
// A cluster's block is four pieces [seq_left | qual_left | seq_right | qual_right] at byte offsets that are not dword aligned, cut
// out of a read at arbitrary nibble / byte offsets.  Composing the block byte by byte costs ~100 instructions per byte; here every
// lane works on whole dwords twice:
//   0. the event's entry (packed bases + qualities, <= 480 B) into LDS: one batch of independent, coalesced loads per cluster.
//   1. piece dwords into LDS: a sequence piece dword is 8 nibbles of the BAM-packed read = a 40-bit window of the source shifted by
//      0 or 4 bits (the table keeps BAM's nibble order); a quality piece dword is the next 32 bits of the stream of W-bit alphabet indices.
//      Tails are zeroed, and every piece sits between zero guard dwords.
//   2. output dword at block byte o = OR over the (1, rarely 2-3) pieces it overlaps of that piece's bytes [o - start, o - start + 4):
//      two LDS dwords and one alignbyte each; the guards supply the zeros on either side of a piece.
// 16 lanes per cluster.  Clusters of multi-event bins (consensus storage) and reads longer than PACK_MAX_LQ take the bytewise path.
constexpr int PACK_RAW_DWORDS = (PACK_MAX_LQ / 2 + PACK_MAX_LQ) / 4 + 2; // entry of a PACK_MAX_LQ read + read-ahead
constexpr int PACK_LDS_DWORDS = (PACK_MAX_LQ / 8 + 2) + (PACK_MAX_LQ / 4 + 2) + 5 + 1; // two sequence + two quality pieces of left_len + right_len <= PACK_MAX_LQ (W = 8 worst case) + guards

template <int W, bool SLOW> // SLOW: the launch over p.slow_list (bytewise path); otherwise every cluster takes the dword path or is skipped
__global__ __launch_bounds__(BLOCK) void k_cluster_pack_codes(PackArgs p, int64_t n_clusters, const uint64_t *__restrict__ str_off, const uint64_t *__restrict__ cig_off,
                                                              const uint32_t *__restrict__ cig_blob, uint8_t *__restrict__ out_str, uint32_t *__restrict__ out_cig)
{
	__shared__ uint32_t s_piece[GROUPS_PER_BLOCK][PACK_LDS_DWORDS];
	__shared__ uint32_t s_raw[GROUPS_PER_BLOCK][PACK_RAW_DWORDS]; // the event's entry (packed bases, qualities) as it lies in the blob
	__shared__ uint8_t s_lut[256]; // phred -> alphabet index
	__shared__ uint8_t s_code[256]; // character -> 4-bit code (bytewise path)
	if (W < 8) s_lut[threadIdx.x] = p.qlut[(threadIdx.x + 33u) & 255u]; // BLOCK == 256; p.qlut is indexed by character
	if (SLOW) s_code[threadIdx.x] = NT16_CODE_OF[threadIdx.x];
	const int grp = (int)(threadIdx.x / GROUP);
	const int gl = (int)(threadIdx.x % GROUP);
	const int64_t k_ = (int64_t)blockIdx.x * GROUPS_PER_BLOCK + grp;
	const int64_t c = SLOW ? (k_ < n_clusters ? (int64_t)p.slow_list[k_] : 0) : k_; // SLOW: n_clusters = entries of the list
	bool active = k_ < n_clusters;
	int ll = 0, lr = 0, lq = -1, begin = 0;
	uint64_t soff = 0;
	int ncg = 0;
	uint64_t scig = 0, dcig = 0, doff = 0;
	if (active) { ll = p.ll[c]; lr = p.lr[c]; lq = p.src_lq[c]; begin = p.src_begin[c]; soff = p.src_off[c]; ncg = p.ncig[c]; scig = p.src_cig[c]; dcig = cig_off[c]; doff = str_off[c]; }
	const uint32_t cig_first = gl < ncg ? cig_blob[scig + gl] : 0u; // issued with the entry's loads below; CIGARs longer than 16 ops finish at the end
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
	const int qb = (lq + 1) / 2; // first quality byte of the entry
	if (fast) {
		// 0. the whole entry into LDS with one batch of independent loads (one memory round trip per cluster; everything after reads LDS)
		const uint32_t *g4 = reinterpret_cast<const uint32_t *>(p.c.seq_blob + soff); // entries are 4-byte aligned
		const int nraw = (qb + lq + 3) / 4 + 1; // + one dword of read-ahead for the unaligned windows below (blob slack, see k_clip_gather)
		uint32_t r[PACK_RAW_DWORDS / GROUP + 1];
#pragma unroll
		for (int u = 0; u < PACK_RAW_DWORDS / GROUP + 1; ++u) { const int i = gl + GROUP * u; r[u] = i < nraw ? g4[i] : 0u; }
#pragma unroll
		for (int u = 0; u < PACK_RAW_DWORDS / GROUP + 1; ++u) { const int i = gl + GROUP * u; if (i < nraw) s_raw[grp][i] = r[u]; }
	}
	__syncthreads(); // s_raw, s_lut
	if (fast) {
		const bool qmiss = lq > 0 && ((s4[qb >> 2] >> (8 * (qb & 3))) & 0xffu) == 0xffu;
		if (gl == 0) L[0] = 0u; // guards: before the first piece and after each piece
#pragma unroll
		for (int k = 0; k < 4; ++k) if (gl == k + 1) L[st[k] + nD[k]] = 0u;
		// sequence pieces (the dwords of both pieces share one index space: with 16 lanes and ~10 dwords per piece, a loop per piece
		// would leave half the lanes idle in each - and this kernel is VALU bound, 93 % busy in the PMC pass)
		for (int tt = gl; tt < nD[0] + nD[2]; tt += GROUP) {
			const bool h = tt >= nD[0];
			const int t = h ? tt - nD[0] : tt;
			const int nib0 = begin + (h ? ll : 0), len = h ? lr : ll;
			{
				const int B = (nib0 >> 1) + 4 * t;
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
							const uint32_t ph = qmiss ? (uint32_t)('*' - 33) : (four >> (8 * b)) & 0xffu;
							if (j < CNT && j < rem) acc |= (uint64_t)s_lut[ph] << (j * W);
						}
					}
				}
				uint32_t v = (uint32_t)(acc >> off);
				if (W == 8 && rem < 4) v &= (1u << (8 * rem)) - 1u;
				L[(h ? st[3] : st[1]) + t] = v;
			}
		}
	}
	__syncthreads();
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
		// bytewise: from the event's packed read (long reads) or from the consensus storage (left part kept reversed)
		const int64_t j = p.slot[c];
		const bool single = lq >= 0;
		EventView v;
		const uint8_t *cs = nullptr, *cq = nullptr, *rs = nullptr, *rq = nullptr;
		if (single) {
			v.sp = p.c.seq_blob + soff; v.qp = v.sp + (lq + 1) / 2; v.begin = begin; v.ll = ll; v.lr = lr; v.qmiss = lq > 0 && v.qp[0] == 0xff;
		} else {
			const int64_t stride = 2ll * (p.c.SL + p.c.SR);
			cs = p.c.strings + (int64_t)p.c.mslot[j] * stride;
			cq = cs + p.c.SL; rs = cs + 2 * p.c.SL; rq = rs + p.c.SR;
		}
		auto seq_at = [&](bool right, int i) -> uint32_t {
			return s_code[single ? (uint32_t)(uint8_t)v.base(v.begin + (right ? ll : 0) + i) : (uint32_t)(right ? rs[i] : cs[ll - 1 - i])];
		};
		auto qual_at = [&](bool right, int i) -> uint32_t { // character
			return single ? (uint32_t)(uint8_t)v.qual(v.begin + (right ? ll : 0) + i) : (uint32_t)(right ? rq[i] : cq[ll - 1 - i]);
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
							for (int j = 0, i = i0; W * j < off + 8 && i < n; ++j, ++i) acc |= (uint32_t)s_lut[(qual_at(right, i) - 33u) & 255u] << (W * j);
							ch = (acc >> off) & 0xffu;
						}
					} else ch = (seq_at(right, 2 * r) << 4) | (2 * r + 1 < n ? seq_at(right, 2 * r + 1) : 0u);
				}
				word |= ch << (8 * k);
			}
			d[w] = word;
		}
	}
	uint32_t *dc = out_cig + dcig;
	if (gl < ncg) dc[gl] = cig_first;
	for (int i = gl + GROUP; i < ncg; i += GROUP) dc[i] = cig_blob[scig + i];
}

} // namespace ssv
