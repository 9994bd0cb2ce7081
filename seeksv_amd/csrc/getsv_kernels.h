// getsv_kernels.h - the BAM passes of `seeksv getsv` on the GPU.
//
// Reference behaviour being reproduced (file:line in /root/reference/seeksv):
//   CalculateInsertsizeDeviation      cluster.cpp:15-83
//   IsConcordant / IsHardClip         cluster.cpp:136-147, clip_reads.cpp:247-257
//   FindDiscordantReadPairs           getsv.cpp:990-1120   (candidate set = bam_iter_query/bam_iter_read of libbam 0.1.16:
//                                     records of tid with bam_calend > beg && pos < end; calend counts M, D, N)
//   main_depth + read_bam             bam2depth.cpp:17-142, bam2depth.h:29-35 (pileup mask sam/bam.h:124)
#pragma once

#include "clip_kernels.h"
#include "common.h"

namespace ssv {

__device__ __forceinline__ bool is_hard_clip(const DevBatch &b, const RecLine &r)
{
	const int n = r.n_cigar();
	if (n == 0) return false; // the reference reads cigar[-1] here (undefined); "not hard clipped" like the oracle
	return (r.head(0) & 15u) == C_H || (r.op(b.cigar, n - 1) & 15u) == C_H;
}

// ---------------------------------------------------------------------------------------------------------------------
// K4 insert-size statistics
// ---------------------------------------------------------------------------------------------------------------------

constexpr int ISZ_TILE = BLOCK;         // records per workgroup

// cluster.cpp:51-67: MAPQ >= q, not hard clipped, PAIRED && PROPER_PAIR && !DUP && isize > 0
__device__ __forceinline__ bool isize_qualifies(const DevBatch &b, const RecLine &r, int min_mapq)
{
	if (r.mapq() < min_mapq) return false;
	const int f = r.flag();
	if (!((f & F_PAIRED) && (f & F_PROPER) && !(f & F_DUP))) return false;
	if (r.isize() <= 0) return false;
	return !is_hard_clip(b, r);
}

// pass A: four lanes fetch a record's line (the pass covers a bounded prefix of the file - the first -n qualifying records - so whole lines
// are fine); val[i] = its insert size when it qualifies, else 0 (a qualifying insert size is > 0); qualifying records per tile of 256
__global__ __launch_bounds__(BLOCK) void k_isize_count(DevBatch b, int min_mapq, int32_t *__restrict__ val, uint32_t *__restrict__ tile_cnt)
{
	__shared__ uint32_t lds[WAVES_PER_BLOCK];
	const int q = (int)(threadIdx.x & 3);
	uint32_t c = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const int64_t i = (int64_t)blockIdx.x * ISZ_TILE + k * (BLOCK / 4) + (threadIdx.x >> 2);
		if (i < b.n) {
			const RecLine r = rec_load_quad(b.rec, i, q);
			const bool ok = isize_qualifies(b, r, min_mapq);
			if (q == 0) { val[i] = ok ? r.isize() : 0; c += ok ? 1u : 0u; }
		}
	}
	c = wave_sum(c);
	if (lane_id() == 0) lds[wave_id()] = c;
	__syncthreads();
	if (threadIdx.x == 0) tile_cnt[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

// pass B: the qualifying record with file-order ordinal o < max_pairs stores its isize at vals[o]
__global__ __launch_bounds__(BLOCK) void k_isize_collect(const int32_t *__restrict__ val, int64_t n, const uint32_t *__restrict__ tile_base, int64_t count_before, int64_t max_pairs,
                                                         int32_t *__restrict__ vals)
{
	__shared__ uint32_t lds[WAVES_PER_BLOCK + 1];
	const int64_t i = (int64_t)blockIdx.x * ISZ_TILE + threadIdx.x;
	const int v = i < n ? val[i] : 0;
	uint32_t tot;
	const uint32_t ex = block_exclusive_sum(v > 0 ? 1u : 0u, lds, &tot);
	const int64_t o = count_before + tile_base[blockIdx.x] + ex;
	if (v > 0 && o < max_pairs) vals[o] = v;
}

// sum of vals (mode 0) or of the int-wrapped squared deviations from mean (mode 1, cluster.cpp:77) into *acc
// mode 1: the mean is formed on the device from the sum the mode-0 launch left in mean_from[0] (cluster.cpp:72: unsigned long division)
__global__ __launch_bounds__(BLOCK) void k_isize_reduce(const int32_t *__restrict__ vals, int64_t n, int mode, const long long *__restrict__ mean_from, long long *__restrict__ acc)
{
	__shared__ long long lds[WAVES_PER_BLOCK];
	const int mean = mode ? (int)((unsigned long)*mean_from / (unsigned long)n) : 0;
	long long s = 0;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		int x = vals[i];
		if (mode == 0) s += x;
		else {
			int d = (int)((unsigned)x - (unsigned)mean);
			s += (int)((unsigned)d * (unsigned)d);
		}
	}
	s = wave_sum(s);
	if (lane_id() == 0) lds[wave_id()] = s;
	__syncthreads();
	if (threadIdx.x == 0) atomicAdd((unsigned long long *)acc, (unsigned long long)(lds[0] + lds[1] + lds[2] + lds[3]));
}

// ---------------------------------------------------------------------------------------------------------------------
// K5+K7 getsv_scan: one fused pass = discordant-pair tally per junction + coverage of the depth windows
// ---------------------------------------------------------------------------------------------------------------------

constexpr int TILE_SHIFT = 9;          // genome tiles of 512 bp
constexpr uint8_t TM_DEPTH = 1, TM_JUNC = 2;

struct DevJunction { // sorted by (up_tid, beg)
	int32_t up_tid, down_tid, up_pos, down_pos, beg, end;
	uint8_t up_strand, down_strand;
	uint16_t pad;
	int32_t orig; // index in the caller's junction array
};

// the batch's tid column as runs (ssv_batch_t.tid_runs), by value in the kernel arguments: run k = records [first[k], first[k + 1]) on contig tid[k]
constexpr int RUN_MAX = 48;
struct RunTab {
	int32_t n;               // 0: no runs given (or more than RUN_MAX): the tid column is read
	int32_t tid[RUN_MAX];
	int64_t first[RUN_MAX + 1]; // first[n] = records of the batch
};

struct GetsvArgs {
	DevBatch b;
	RunTab runs;
	// genome tile map: which 512-bp tiles can hold the START of a record that matters (windows are extended to the left by
	// the longest reference span when the map is built, so one lookup per record is enough)
	const uint8_t *tilemap;
	const uint32_t *tile_win;    // per tile with TM_DEPTH: the first window that ends at or after the tile's first column
	const uint32_t *tile_junc;   // per tile with TM_JUNC: the first junction window that begins after (tile start - junc_wmax)
	const int64_t *ctg_tile_off; // [n_targets + 1]
	int32_t n_targets;
	// discordant
	const DevJunction *junc;
	int64_t n_junc;
	int32_t junc_wmax;           // longest junction window
	int32_t mean, sd, times, min_ins, max_ins, disc_min_mapq;
	int32_t *counts;             // [n_junc] by orig index
	// depth
	const int32_t *win_tid, *win_beg, *win_end; // merged windows, 1-based inclusive, sorted, disjoint
	const int64_t *win_off;      // [n_win + 1] offset of the window's difference array (len + 1 entries each)
	int64_t n_win;
	int32_t depth_min_mapq;
	int32_t *diff;
	// pileup read cap (see the CapArgs block below): the streaming pass raises cap_flag[0] when a wavefront's records of one tile
	// all start within cap_span bp - a necessary condition for 8000 reads being alive anywhere in the tile's neighbourhood
	int *cap_flag;               // null: no depth pass
	int32_t cap_span;
};

// IsConcordant, cluster.cpp:136-147 (lower bound NOT clamped here)
__device__ __forceinline__ bool is_concordant(int flag, int isize, int mean, int sd, int times)
{
	int lo = mean - sd * times, hi = mean + sd * times;
	if (!(flag & F_REV) && (flag & F_MREV) && lo <= isize && isize <= hi) return true;
	if ((flag & F_REV) && !(flag & F_MREV) && isize < 0) {
		int a = isize < 0 ? -isize : isize;
		return lo <= a && a <= hi;
	}
	return false;
}

// the strand-specific geometry of getsv.cpp:1074-1113 for one candidate read and one junction
__device__ __forceinline__ bool discordant_geometry(const DevJunction &j, int flag, int pos, int mpos, int lq, int mtid, int min_ins, int max_ins)
{
	if (!(j.down_tid != -1 && j.down_tid == mtid)) return false;
	const int K = 5; // kCrossLength, getsv.cpp:15
	const int up = j.up_pos, down = j.down_pos;
	if (j.up_strand == '+' && j.down_strand == '+' && pos + lq <= up + K && mpos + 1 >= down - K) {
		if (!(flag & F_REV) && (flag & F_MREV)) {
			int ins = up - pos + mpos + lq - down + 1;
			if (j.up_tid == j.down_tid && up > down && up - down + 1 + 2 * lq <= max_ins) { // tandem duplication
				while (ins <= max_ins) {
					if (ins >= min_ins) return true;
					ins += up - down + 1;
				}
				return false;
			}
			return min_ins <= ins && ins <= max_ins;
		}
		return false;
	} else if (j.up_strand == '-' && j.down_strand == '+' && (flag & F_REV) && (flag & F_MREV) && mpos + 1 >= down - K) {
		int ins = pos + 1 - up + 1 + mpos + lq - down + 1;
		return min_ins <= ins && ins <= max_ins;
	} else if (j.up_strand == '+' && j.down_strand == '-' && !(flag & F_REV) && !(flag & F_MREV) && pos + lq <= up + K && mpos + lq <= down + K) {
		int ins = up - pos + down - (mpos + lq) + 1;
		return min_ins <= ins && ins <= max_ins;
	}
	return false;
}

// The fixed fields of a candidate record and its first five CIGAR operations are ONE 64-byte line (ssv_record): the per-candidate kernels
// are bound by the number and latency of scattered loads (VALU 6 % busy), and a line is one sector where the structure-of-arrays batch
// cost one sector per field.
struct CandRec {
	int mapq, flag, isize, nc, mtid, mpos, lq;
	const uint32_t *cig; // all operations (read beyond the fifth)
	RecLine line;
};

__device__ __forceinline__ CandRec cand_load(const DevBatch &b, int64_t i)
{
	CandRec r;
	r.line = rec_load(b.rec, i);
	r.mapq = r.line.mapq(); r.flag = r.line.flag(); r.isize = r.line.isize(); r.nc = r.line.n_cigar(); r.mtid = r.line.mtid(); r.mpos = r.line.mpos(); r.lq = r.line.l_qseq();
	r.cig = b.cigar + r.line.cigar_off();
	return r;
}

__device__ __forceinline__ uint32_t cand_op(const CandRec &r, int k) { return k < 5 ? r.line.head(k) : r.cig[k]; }

// the read's own part of the discordant test (everything that does not look at a junction)
__device__ __forceinline__ bool discordant_read(const GetsvArgs &a, const CandRec &r)
{
	if (r.mapq < a.disc_min_mapq) return false;
	const int flag = r.flag;
	if (flag & (F_DUP | F_UNMAP | F_MUNMAP)) return false;
	if (is_concordant(flag, r.isize, a.mean, a.sd, a.times)) return false;
	const int n = r.nc;
	if (n > 0 && ((cand_op(r, 0) & 15u) == C_H || (cand_op(r, n - 1) & 15u) == C_H)) return false; // IsHardClip (n == 0: the reference reads cigar[-1]; "not hard clipped" like the oracle)
	return true;
}

// junction windows that can overlap [pos, rend): beg in (pos - wmax, rend) on this contig.  `lo`: any junction index not behind the first of them
// (the loop skips what ends before the read); junction_at(m): junction m, from wherever the caller keeps it; hit(m, j): the read counts for junction m.
template <typename F, typename H> __device__ __forceinline__ void discordant_walk(const GetsvArgs &a, const CandRec &r, int tid, int pos, int64_t lo, F junction_at, H hit)
{
	// bam_calend of libbam 0.1.16: M, D, N advance; no CIGAR -> pos + 1
	const int n = r.nc;
	int rend = pos;
	if (n == 0) rend = pos + 1;
	else for (int k = 0; k < n; ++k) { const uint32_t c = cand_op(r, k); const int op = (int)(c & 15u); if (op == C_M || op == C_D || op == C_N) rend += (int)(c >> 4); }
	for (int64_t m = lo; m < a.n_junc; ++m) {
		const DevJunction j = junction_at(m);
		if (j.up_tid != tid || j.beg >= rend) break;
		if (!(rend > j.beg && pos < j.end)) continue;
		if (discordant_geometry(j, r.flag, pos, r.mpos, r.lq, r.mtid, a.min_ins, a.max_ins)) hit(m, j);
	}
}

__device__ __forceinline__ void discordant_record(const GetsvArgs &a, const CandRec &r, int tid, int pos, int64_t tile)
{
	if (!discordant_read(a, r)) return;
	// The tile of the record's start knows the first junction that begins after (tile start - wmax): a lower bound of the binary search's answer
	discordant_walk(a, r, tid, pos, a.tile_junc[tile], [&](int64_t m) { return a.junc[m]; }, [&](int64_t, const DevJunction &j) { atomicAdd(&a.counts[j.orig], 1); });
}

// one covered stretch [s, e] (1-based columns) of a read into the windows it meets: +sign where it enters a window, -sign behind where it leaves it.
// w: the caller's window cursor (-1: not looked up yet; the first window that ends at or after the tile's first column, then walked on)
__device__ __forceinline__ void depth_segment(const GetsvArgs &a, int tid, int s, int e, int64_t tile, int64_t &w, int sign)
{
	if (w < 0) w = a.tile_win[tile];
	while (w < a.n_win && a.win_tid[w] == tid && a.win_end[w] < s) ++w;
	for (int64_t x = w; x < a.n_win && a.win_tid[x] == tid && a.win_beg[x] <= e; ++x) {
		int wb = a.win_beg[x], we = a.win_end[x];
		int lo = s > wb ? s : wb, hi = e < we ? e : we;
		int32_t *d = a.diff + a.win_off[x];
		atomicAdd(&d[lo - wb], sign);
		atomicAdd(&d[hi + 1 - wb], -sign);
	}
}

// sign -1: the correction pass of the read cap takes a dropped read's coverage out again (it has applied the filter already)
__device__ __forceinline__ bool depth_counts(const GetsvArgs &a, const CandRec &r)
{
	if (r.mapq < a.depth_min_mapq) return false;                              // read_bam: MAPQ < mapQ -> treated as unmapped
	return !(r.flag & (F_UNMAP | F_SECONDARY | F_QCFAIL | F_DUP));            // BAM_DEF_MASK
}

__device__ __forceinline__ void depth_record(const GetsvArgs &a, const CandRec &r, int tid, int pos, int64_t tile, int sign = 1)
{
	if (sign > 0 && !depth_counts(a, r)) return;
	const int n = r.nc;
	int col = pos + 1; // 1-based
	int64_t w = -1;
	for (int k = 0; k < n; ++k) {
		const uint32_t c = cand_op(r, k);
		int op = (int)(c & 15u), len = (int)(c >> 4);
		if (op == C_M) { // libbam 0.1.16 pileup: only M covers, only M / D / N advance; '=' and 'X' are skipped like padding (tests/golden/getsv/eqx.*)
			if (len > 0) depth_segment(a, tid, col, col + len - 1, tile, w, sign);
			col += len;
		} else if (op == C_D || op == C_N) col += len;
	}
}

// tile-map lookup of one record: which slow paths (if any) it needs, and its tile
__device__ __forceinline__ uint32_t getsv_tile_bits(const GetsvArgs &a, int tid, int pos, int64_t &tile)
{
	tile = 0;
	if (tid < 0 || tid >= a.n_targets || pos < 0) return 0;
	int64_t t = a.ctg_tile_off[tid] + (pos >> TILE_SHIFT);
	tile = t;
	return t < a.ctg_tile_off[tid + 1] ? a.tilemap[t] : 0u;
}

// Built on the device at ssv_getsv_begin (and again when a batch reports a longer reference span): one thread per depth window /
// junction window marks the tiles that can hold the START (0-based pos) of a record overlapping it - columns [beg, end] 1-based <->
// pos + 1 <= end, pos + span >= beg - and leaves, per marked tile, where the per-record look-up starts, so that a candidate record
// needs one load instead of a binary search over the windows / junctions (the value does not depend on which window marks the tile).
__global__ void k_tile_mark_windows(const int32_t *__restrict__ win_tid, const int32_t *__restrict__ win_beg, const int32_t *__restrict__ win_end, int64_t n_win, int32_t span,
                                    const int64_t *__restrict__ ctg_tile_off, int32_t n_targets, uint32_t *__restrict__ map32, uint32_t *__restrict__ tile_win)
{
	const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= n_win) return;
	const int tid = win_tid[w];
	if (tid < 0 || tid >= n_targets) return;
	int64_t p0 = (int64_t)win_beg[w] - span, p1 = (int64_t)win_end[w] - 1;
	if (p1 < 0) return;
	if (p0 < 0) p0 = 0;
	const int64_t base = ctg_tile_off[tid], ntile = ctg_tile_off[tid + 1] - base;
	int64_t t0 = p0 >> TILE_SHIFT, t1 = p1 >> TILE_SHIFT;
	if (t0 >= ntile) return;
	if (t1 >= ntile) t1 = ntile - 1;
	for (int64_t t = t0; t <= t1; ++t) {
		const int col = (int)(t << TILE_SHIFT) + 1; // first column a record starting in this tile can cover
		int64_t lo = 0, hi = n_win;                  // first window with (tid, end) >= (tid, col)
		while (lo < hi) {
			const int64_t m = (lo + hi) >> 1;
			const int wt = win_tid[m];
			if (wt < tid || (wt == tid && win_end[m] < col)) lo = m + 1; else hi = m;
		}
		tile_win[base + t] = (uint32_t)lo;
		atomicOr(&map32[(base + t) >> 2], (uint32_t)TM_DEPTH << (8 * ((base + t) & 3)));
	}
}

__global__ void k_tile_mark_junctions(const DevJunction *__restrict__ junc, int64_t n_junc, int32_t span, int32_t wmax, const int64_t *__restrict__ ctg_tile_off, int32_t n_targets,
                                      uint32_t *__restrict__ map32, uint32_t *__restrict__ tile_junc)
{
	const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n_junc) return;
	const int tid = junc[k].up_tid;
	if (tid < 0 || tid >= n_targets) return;
	int64_t p0 = (int64_t)junc[k].beg - span, p1 = (int64_t)junc[k].end - 1; // rend > beg && pos < end
	if (p1 < 0) return;
	if (p0 < 0) p0 = 0;
	const int64_t base = ctg_tile_off[tid], ntile = ctg_tile_off[tid + 1] - base;
	int64_t t0 = p0 >> TILE_SHIFT, t1 = p1 >> TILE_SHIFT;
	if (t0 >= ntile) return;
	if (t1 >= ntile) t1 = ntile - 1;
	for (int64_t t = t0; t <= t1; ++t) {
		const int64_t want = (t << TILE_SHIFT) - (int64_t)wmax; // records of this tile have pos >= tile start
		int64_t lo = 0, hi = n_junc;                             // first junction with (up_tid, beg) > (tid, want)
		while (lo < hi) {
			const int64_t m = (lo + hi) >> 1;
			if (junc[m].up_tid < tid || (junc[m].up_tid == tid && (int64_t)junc[m].beg <= want)) lo = m + 1; else hi = m;
		}
		tile_junc[base + t] = (uint32_t)lo;
		atomicOr(&map32[(base + t) >> 2], (uint32_t)TM_JUNC << (8 * ((base + t) & 3)));
	}
}

struct GetsvStage {
	uint32_t *tile_cnt, *tile_off, *stage;
	int64_t block_cap;
	int *overflow;
	int64_t ntiles;
	struct DenseTile *dense_list; // the scan tiles that are dense with candidates (k_dense_tiles lists them, k_getsv_cand leaves them to k_getsv_cand_dense); null: none
	int *dense_n;
	unsigned long long *n_cand; // candidates staged by the scan (null: not counted)
};

// K5/K7 getsv_scan, the streaming pass: reads tid and pos of every record (8 B/record, eight 16-byte loads in flight per lane), looks
// the record's start tile up in the genome tile map (L2 resident, wave-coherent because the BAM is coordinate sorted) and writes the
// indices of the ~1 % of records that start near a depth window or a junction window.  Same persistent, atomic-free structure as
// k_clip_scan; the lookups are branch-free (clamped indices) so that they pipeline.
// Does the whole tile lie inside ONE run of the tid column (then its contig is known without reading the column)?  Workgroup-uniform: a binary
// search over the <= 48 run starts in the kernel arguments (scalar loads).  -1: no (no runs given, or a run boundary falls into the tile).
__device__ __forceinline__ int tile_run_tid(const RunTab &R, int64_t tile, int64_t n)
{
	if (R.n <= 0 || (tile + 1) * CS_TILE > n) return -1; // (the batch's last, partial tile reads the column)
	const int64_t i0 = tile * CS_TILE, i1 = (tile + 1) * CS_TILE;
	int lo = 0, hi = R.n - 1;
	while (lo < hi) { const int m = (lo + hi + 1) >> 1; if (R.first[m] <= i0) lo = m; else hi = m - 1; }
	return R.first[lo + 1] >= i1 && R.tid[lo] >= 0 ? R.tid[lo] : -1;
}

__device__ __forceinline__ void getsv_scan_load(const DevBatch &b, int64_t tile, int4 (&t4)[CS_SUB], int4 (&p4)[CS_SUB], int run_tid)
{
	const int64_t t0 = tile * CS_TILE + (int64_t)threadIdx.x * CS_ITEMS;
	if (run_tid >= 0) { // workgroup-uniform: the tile's contig is known from the runs: 4 B/record
#pragma unroll
		for (int sub = 0; sub < CS_SUB; ++sub) {
			t4[sub] = make_int4(run_tid, run_tid, run_tid, run_tid);
			p4[sub] = stream_load_i4(b.pos + t0 + (int64_t)sub * (BLOCK * CS_ITEMS));
		}
	} else if ((tile + 1) * CS_TILE <= b.n) { // workgroup-uniform
#pragma unroll
		for (int sub = 0; sub < CS_SUB; ++sub) {
			t4[sub] = stream_load_i4(b.tid + t0 + (int64_t)sub * (BLOCK * CS_ITEMS));
			p4[sub] = stream_load_i4(b.pos + t0 + (int64_t)sub * (BLOCK * CS_ITEMS));
		}
	} else {
#pragma unroll
		for (int sub = 0; sub < CS_SUB; ++sub) {
			const int64_t i0 = t0 + (int64_t)sub * (BLOCK * CS_ITEMS);
			int tt[CS_ITEMS], pp[CS_ITEMS];
#pragma unroll
			for (int k = 0; k < CS_ITEMS; ++k) { bool in = i0 + k < b.n; tt[k] = in ? b.tid[i0 + k] : -1; pp[k] = in ? b.pos[i0 + k] : 0; }
			t4[sub] = make_int4(tt[0], tt[1], tt[2], tt[3]); p4[sub] = make_int4(pp[0], pp[1], pp[2], pp[3]);
		}
	}
}

// SSV_VERIFY_RUNS=1 (always in the test suite): the batch's run list is the producer's statement about the tid column, and k_getsv_scan takes
// a tile's contig from it without reading the column.  One streaming pass holds the column against the list: every record of run k must carry
// tid[k].  *bad = 1 + index of a record that does not (any one of them).
__global__ __launch_bounds__(BLOCK) void k_verify_runs(const int32_t *__restrict__ tid, int64_t n, RunTab runs, unsigned long long *bad)
{
	for (int64_t i0 = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * 4; i0 < n; i0 += (int64_t)gridDim.x * BLOCK * 4) {
		int k = 0;
		while (k + 1 < runs.n && runs.first[k + 1] <= i0) ++k;
		if (i0 + 4 <= n) {
			const int4 t = stream_load_i4(tid + i0);
			const int v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				while (k + 1 < runs.n && runs.first[k + 1] <= i0 + j) ++k;
				if (v[j] != runs.tid[k]) atomicMax(bad, (unsigned long long)(i0 + j + 1));
			}
		} else for (int64_t i = i0; i < n; ++i) {
			while (k + 1 < runs.n && runs.first[k + 1] <= i) ++k;
			if (tid[i] != runs.tid[k]) atomicMax(bad, (unsigned long long)(i + 1));
		}
	}
}

__global__ __launch_bounds__(BLOCK) void k_getsv_scan(GetsvArgs a, GetsvStage g)
{
	__shared__ uint64_t lds[2][WAVES_PER_BLOCK];
	const DevBatch &b = a.b;
	uint32_t cursor = 0;
	int parity = 0;
	const int64_t region = (int64_t)blockIdx.x * g.block_cap;
	const int last_tid = a.n_targets - 1;
	int4 t4[CS_SUB], p4[CS_SUB], nt4[CS_SUB], np4[CS_SUB];
	int run = -1, nrun = -1; // the contig of the whole tile when the batch's runs say so (workgroup-uniform)
	if ((int64_t)blockIdx.x < g.ntiles) { run = tile_run_tid(a.runs, blockIdx.x, b.n); getsv_scan_load(b, blockIdx.x, t4, p4, run); }
	for (int64_t tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x, parity ^= 1) {
		const int64_t t0 = tile * CS_TILE + (int64_t)threadIdx.x * CS_ITEMS;
		const int64_t next = tile + gridDim.x;
		if (next < g.ntiles) { nrun = tile_run_tid(a.runs, next, b.n); getsv_scan_load(b, next, nt4, np4, nrun); } // software pipeline, as in k_clip_scan
		uint32_t mask = 0;
		// Fast path (coordinate-sorted input at WGS depth): every record of this wavefront's share of the tile is on one contig and
		// their start tiles span < 64 genome tiles.  Then 64 tile-map bytes are fetched once per wavefront (one byte per lane), turned
		// into a wave-uniform 64-bit "tile is interesting" mask by a ballot, and each record only shifts that mask - no per-record
		// memory access at all: a shift, a subtract, a 64-bit shift and an and-or per record, plus a min / max for the span.
		const int tid0 = run >= 0 ? run : __builtin_amdgcn_readfirstlane(t4[0].x);
		bool same = true;
		int pmin = 0x7fffffff, pmax = -1;
		if (run >= 0) {
#pragma unroll
			for (int sub = 0; sub < CS_SUB; ++sub) {
				pmin = min(pmin, min(min(p4[sub].x, p4[sub].y), min(p4[sub].z, p4[sub].w)));
				pmax = max(pmax, max(max(p4[sub].x, p4[sub].y), max(p4[sub].z, p4[sub].w)));
			}
			same = pmin >= 0;
		} else {
#pragma unroll
			for (int sub = 0; sub < CS_SUB; ++sub) {
				const int tid[CS_ITEMS] = {t4[sub].x, t4[sub].y, t4[sub].z, t4[sub].w};
				const int pos[CS_ITEMS] = {p4[sub].x, p4[sub].y, p4[sub].z, p4[sub].w};
#pragma unroll
				for (int k = 0; k < CS_ITEMS; ++k) {
					same = same && tid[k] == tid0 && pos[k] >= 0;
					pmin = pos[k] < pmin ? pos[k] : pmin;
					pmax = pos[k] > pmax ? pos[k] : pmax;
				}
			}
		}
		const bool uniform = __all(same) && tid0 >= 0 && tid0 <= last_tid;
		int wmin = 0, wmax = 0;
		if (uniform) {
			const int wpmin = -wave_max(-pmin), wpmax = wave_max(pmax);
			wmin = wpmin >> TILE_SHIFT; wmax = wpmax >> TILE_SHIFT;
			// 3,328 consecutive records (this wavefront's share of the tile spans that many) starting within one read span: the read cap of
			// the reference's pileup may bind near here (k_cap_*); never at WGS depths
			if (a.cap_flag && wpmax - wpmin <= a.cap_span && lane_id() == 0) *a.cap_flag = 1;
		}
		if (uniform && wmax - wmin < WAVE) {
			const int64_t lo = scalar_load(a.ctg_tile_off + tid0), hi = scalar_load(a.ctg_tile_off + tid0 + 1); // (wave-uniform; NOT a vector load: see scalar_load)
			const int64_t mine = lo + wmin + lane_id();
			// (bit k = genome tile wmin + k of this contig; tiles past the contig's end stay 0, so a position beyond the contig is no candidate)
			const uint64_t interesting = __ballot(mine < hi && a.tilemap[mine < hi ? mine : hi - 1] != 0);
			if (interesting) { // (wave-uniform)
#pragma unroll
				for (int sub = 0; sub < CS_SUB; ++sub) {
					const int pos[CS_ITEMS] = {p4[sub].x, p4[sub].y, p4[sub].z, p4[sub].w};
#pragma unroll
					for (int k = 0; k < CS_ITEMS; ++k) // 0 <= (pos >> 9) - wmin < 64 for every record of the wavefront
						mask |= ((uint32_t)(interesting >> ((pos[k] >> TILE_SHIFT) - wmin)) & 1u) << (sub * CS_ITEMS + k);
				}
			}
		} else {
#pragma unroll
			for (int sub = 0; sub < CS_SUB; ++sub) {
				const int tid[CS_ITEMS] = {t4[sub].x, t4[sub].y, t4[sub].z, t4[sub].w};
				const int pos[CS_ITEMS] = {p4[sub].x, p4[sub].y, p4[sub].z, p4[sub].w};
#pragma unroll
				for (int k = 0; k < CS_ITEMS; ++k) {
					// clamped, unconditional lookups; validity is applied afterwards
					const int tc = tid[k] < 0 ? 0 : (tid[k] > last_tid ? last_tid : tid[k]);
					const int64_t lo = a.ctg_tile_off[tc], hi = a.ctg_tile_off[tc + 1];
					int64_t t = lo + ((pos[k] < 0 ? 0 : pos[k]) >> TILE_SHIFT);
					const bool valid = tid[k] >= 0 && tid[k] <= last_tid && pos[k] >= 0 && t < hi;
					t = t < hi ? t : hi - 1;
					const bool cand = valid && a.tilemap[t] != 0;
					mask |= (cand ? 1u : 0u) << (sub * CS_ITEMS + k);
				}
			}
		}
		uint64_t packed = 0; // candidates per sub-tile, four 16-bit fields
#pragma unroll
		for (int sub = 0; sub < CS_SUB; ++sub) packed |= (uint64_t)__popc((mask >> (sub * CS_ITEMS)) & ((1u << CS_ITEMS) - 1u)) << (16 * sub);
		stage_tile_candidates<CS_ITEMS, true>(mask, packed, tile, t0, lds, parity, cursor, region, g.block_cap, g.tile_cnt, g.tile_off, g.stage, g.overflow);
#pragma unroll
		for (int sub = 0; sub < CS_SUB; ++sub) { t4[sub] = nt4[sub]; p4[sub] = np4[sub]; }
		run = nrun;
	}
	if (g.n_cand && threadIdx.x == 0 && cursor) atomicAdd(g.n_cand, (unsigned long long)cursor); // (the host sizes the dense pass's grid by it)
}

// The same pass for a batch that comes with its tid column as runs (ssv_batch_t.tid_runs: every batch of the product's own decoders): the column is never read - a
// tile inside one run has its contig from the list; a tile with a run boundary in it (at most one per run) and the batch's last, partial tile work their records'
// contigs out of the list, record by record - and a thread's prefetched tiles really stay in flight while it works.  k_getsv_scan took 3.4 us per tile and
// workgroup whatever its prefetch depth, its vector work (skipping it changed nothing) or the bytes in flight (555 us at 1024 workgroups, 623 at 768, 808 at 512:
// only more workgroups helped), because every tile ENDED the loads in flight (vmcnt counts in order, and the compiler's waits are static):
//   * `cur = next` at the end of a turn reads `next`: a wait for the loads just issued.  Here: CS_AHEAD + 1 register buffers with fixed roles per unrolled step;
//     the tile-map byte requested one tile ahead stays in its slot the same way;
//   * __syncthreads() = `s_waitcnt vmcnt(0)` + s_barrier (hipcc fences global memory at every barrier).  Here: lds_barrier() (common.h);
//   * a.ctg_tile_off[run] was a VECTOR load at a uniform address, waited for at once.  Here: scalar_load() (common.h);
//   * loads only some paths issue (the partial last tile's, the look-up's) leave the wait-count pass without the number of loads behind a buffer, and it waits
//     for all.  Here: every step issues the same four position loads and one tile-map byte (clamped addresses); the partial tile loads on demand;
//   * a spilled value's scratch reload is one more vector load to wait for (the record-by-record path, sixteen times unrolled or as a call, cost 30 registers).
//     Here: that path is a loop over the tile's positions parked in LDS.
// What is left: the per-tile stores of tile_cnt / tile_off / candidates (a later write to one of their source registers waits for them: vmcnt(0) again, behind
// the barrier) - taking that away too (the bookkeeping in LDS until the end) was 5 % SLOWER, profiles/r05_getsv_cand_notes.txt B.3.  Same box, same run: k_getsv_scan 542-552 us, this 404-439 us = 5.6-6.1 TB/s (the guide's float4 copy: 6.29).
// (always the same four loads: the caller clamps the tile to the batch's last whole one)
__device__ __forceinline__ void getsv_scan_load_pos(const DevBatch &b, int64_t tile, int4 (&p4)[CS_SUB])
{
	const int64_t t0 = tile * CS_TILE + (int64_t)threadIdx.x * CS_ITEMS;
#pragma unroll
	for (int sub = 0; sub < CS_SUB; ++sub) p4[sub] = stream_load_i4(b.pos + t0 + (int64_t)sub * (BLOCK * CS_ITEMS));
}

constexpr int CS_AHEAD = 2; // tiles of positions a thread has on their way while it looks at one (1 with six workgroups per CU, 3: within the noise of this)

__global__ __launch_bounds__(BLOCK, 4) void k_getsv_scan_runs(GetsvArgs a, GetsvStage g)
{
	__shared__ uint64_t lds[2][WAVES_PER_BLOCK];
	__shared__ int s_pos[CS_SUB * CS_ITEMS][BLOCK]; // a tile's positions, for the record-by-record path's loop
	const DevBatch &b = a.b;
	uint32_t cursor = 0;
	int parity = 0;
	const int64_t region = (int64_t)blockIdx.x * g.block_cap;
	const int last_tid = a.n_targets - 1;
	// Record by record (a tile with a run boundary in it, the batch's last tile, wavefronts whose records span 64 genome tiles or more): the contig is the tile's
	// run or the run the record's index falls into.  Rare, and the loop around it is unrolled over its buffers: a LOOP over the sixteen records, their positions
	// through LDS (registers cannot be indexed), so that it costs the kernel neither registers nor code.
	auto by_record = [&](const int4 (&q)[CS_SUB], int run_, int64_t tile, int64_t t0) -> uint32_t {
		const bool partial = (tile + 1) * CS_TILE > b.n; // the batch's last tile: its positions were not prefetched
#pragma unroll
		for (int sub = 0; sub < CS_SUB; ++sub) {
			s_pos[sub * CS_ITEMS + 0][threadIdx.x] = q[sub].x; s_pos[sub * CS_ITEMS + 1][threadIdx.x] = q[sub].y;
			s_pos[sub * CS_ITEMS + 2][threadIdx.x] = q[sub].z; s_pos[sub * CS_ITEMS + 3][threadIdx.x] = q[sub].w;
		}
		int r = 0;
		if (run_ < 0) { int hi = a.runs.n - 1; const int64_t i0 = tile * CS_TILE; while (r < hi) { const int m = (r + hi + 1) >> 1; if (a.runs.first[m] <= i0) r = m; else hi = m - 1; } }
		uint32_t mask = 0;
#pragma nounroll
		for (int j = 0; j < CS_SUB * CS_ITEMS; ++j) { // (records in increasing index order: the run cursor only moves forward)
			const int64_t i = t0 + (int64_t)(j / CS_ITEMS) * (BLOCK * CS_ITEMS) + (j % CS_ITEMS);
			const int pos = partial ? (i < b.n ? b.pos[i] : 0) : s_pos[j][threadIdx.x];
			int tid = run_;
			if (run_ < 0) {
				while (r + 1 < a.runs.n && a.runs.first[r + 1] <= i) ++r;
				tid = i < b.n ? a.runs.tid[r] : -1;
			}
			const int tc = tid < 0 ? 0 : (tid > last_tid ? last_tid : tid);
			const int64_t lo = a.ctg_tile_off[tc], hi = a.ctg_tile_off[tc + 1];
			int64_t t = lo + ((pos < 0 ? 0 : pos) >> TILE_SHIFT);
			const bool valid = tid >= 0 && tid <= last_tid && pos >= 0 && t < hi;
			t = t < hi ? t : hi - 1;
			if (valid && a.tilemap[t] != 0) mask |= 1u << j;
		}
		return mask;
	};
	// CS_AHEAD + 1 buffers of positions in registers, their roles fixed per unrolled step: a buffer is never COPIED - a `cur = next` at the end of a turn reads
	// `next`, so the turn ends by waiting for the loads it has just issued (vmcnt counts in order), and a tile then costs a memory round trip however far ahead
	// its loads were issued (k_getsv_scan, round 1-4: 3.4 us a tile per workgroup at any prefetch depth).
	constexpr int NB = CS_AHEAD + 1;
	int4 pq[NB][CS_SUB];
	// A tile's "head": its run, whether the wavefront's records start within 64 genome tiles of one contig (the fast path of k_getsv_scan), and then the request for
	// the wavefront's 64 tile-map bytes - issued one tile AHEAD, beside the tile in front of it.
	int run = -1, wmin = 0, nrun = -1, nwmin = 0;
	bool fast = false, nfast = false;
	bool cap_seen = false;
	uint8_t tmq[NB];  // (a tile's byte stays in the slot its load was issued into: a copy would wait for the load, like a copy of a position buffer)
	bool tm_in = false, ntm_in = false;
	const int full_tiles = __builtin_amdgcn_readfirstlane((int)(b.n / CS_TILE)); // (> 0: the launcher sends batches of less than a tile to k_getsv_scan)
	auto head = [&](int64_t tile, const int4 (&q)[CS_SUB], int &run_, bool &fast_, int &wmin_, uint8_t &tm_, bool &in_) {
		run_ = tile < g.ntiles ? tile_run_tid(a.runs, tile, b.n) : -1; // (-1: a boundary inside, the partial last tile, or no such tile)
		int pmin = 0x7fffffff, pmax = -1;
#pragma unroll
		for (int sub = 0; sub < CS_SUB; ++sub) {
			pmin = min(pmin, min(min(q[sub].x, q[sub].y), min(q[sub].z, q[sub].w)));
			pmax = max(pmax, max(max(q[sub].x, q[sub].y), max(q[sub].z, q[sub].w)));
		}
		const bool ok = run_ >= 0 && run_ <= last_tid && __all(pmin >= 0);
		const int wpmin = -wave_max(-pmin), wpmax = wave_max(pmax);
		wmin_ = wpmin >> TILE_SHIFT;
		cap_seen = cap_seen || (ok && wpmax - wpmin <= a.cap_span); // (see k_getsv_scan; stored once, at the end: a store in the loop is one more thing a wait can be for)
		fast_ = ok && (wpmax >> TILE_SHIFT) - wmin_ < WAVE;
		// the wavefront's 64 tile-map bytes: ALWAYS one load (an address inside the map when the tile takes the other path)
		const int rc = ok ? run_ : 0;
		const int64_t lo = scalar_load(a.ctg_tile_off + rc), hi = scalar_load(a.ctg_tile_off + rc + 1); // (wave-uniform; NOT a vector load: see scalar_load)
		const int64_t mine = lo + (fast_ ? wmin_ : 0) + lane_id();
		in_ = fast_ && mine < hi; // (tiles past the contig's end stay 0: a position beyond the contig is no candidate)
		tm_ = a.tilemap[mine < hi ? mine : (hi > 0 ? hi - 1 : 0)]; // (hi == 0: a header whose first contigs have no tiles; the map has 16 spare bytes)
	};
	auto step = [&](int64_t tile, const int4 (&cur)[CS_SUB], const int4 (&nxt)[CS_SUB], int4 (&far)[CS_SUB], const uint8_t &tm, uint8_t &ntm) {
		const int64_t t0 = tile * CS_TILE + (int64_t)threadIdx.x * CS_ITEMS;
		const int64_t next = tile + gridDim.x;
		const int ahead = (int)tile + CS_AHEAD * (int)gridDim.x; // (tiles: < 2^31)
		getsv_scan_load_pos(b, ahead < full_tiles ? ahead : full_tiles - 1, far);
		head(next, nxt, nrun, nfast, nwmin, ntm, ntm_in);
		uint32_t mask = 0;
		if (fast) {
			const uint64_t interesting = __ballot(tm_in && tm != 0);
			if (interesting) { // (wave-uniform; most wavefronts' 64 genome tiles hold no window: nothing to do per record)
#pragma unroll
				for (int sub = 0; sub < CS_SUB; ++sub) {
					const int pos[CS_ITEMS] = {cur[sub].x, cur[sub].y, cur[sub].z, cur[sub].w};
#pragma unroll
					for (int k = 0; k < CS_ITEMS; ++k) // 0 <= (pos >> 9) - wmin < 64 for every record of the wavefront
						mask |= ((uint32_t)(interesting >> ((pos[k] >> TILE_SHIFT) - wmin)) & 1u) << (sub * CS_ITEMS + k);
				}
			}
		} else mask = by_record(cur, run, tile, t0);
		uint64_t packed = 0; // candidates per sub-tile, four 16-bit fields
#pragma unroll
		for (int sub = 0; sub < CS_SUB; ++sub) packed |= (uint64_t)__popc((mask >> (sub * CS_ITEMS)) & ((1u << CS_ITEMS) - 1u)) << (16 * sub);
		stage_tile_candidates<CS_ITEMS, true>(mask, packed, tile, t0, lds, parity, cursor, region, g.block_cap, g.tile_cnt, g.tile_off, g.stage, g.overflow);
		run = nrun; fast = nfast; wmin = nwmin; tm_in = ntm_in;
		parity ^= 1;
	};
#pragma unroll
	for (int d = 0; d < CS_AHEAD; ++d) { const int t = (int)blockIdx.x + d * (int)gridDim.x; getsv_scan_load_pos(b, t < full_tiles ? t : full_tiles - 1, pq[d]); }
	head(blockIdx.x, pq[0], run, fast, wmin, tmq[0], tm_in);
	int64_t tile = blockIdx.x;
	while (tile < g.ntiles) {
#pragma unroll
		for (int j = 0; j < NB; ++j) {
			if (tile >= g.ntiles) break;
			step(tile, pq[j], pq[(j + 1) % NB], pq[(j + CS_AHEAD) % NB], tmq[j], tmq[(j + 1) % NB]);
			tile += gridDim.x;
		}
	}
	if (g.n_cand && threadIdx.x == 0 && cursor) atomicAdd(g.n_cand, (unsigned long long)cursor);
	if (cap_seen && a.cap_flag && lane_id() == 0) *a.cap_flag = 1;
}

// the slow paths, one thread per staged record: discordant tally and / or depth coverage.  Candidates come in runs - a scan tile near a junction holds hundreds
// (30x) or all of its 4096 records (300x, BASELINE config 3), its neighbours none: the tiles with GC_DENSE_MIN candidates or more (80 % of the candidates at 30x, all at
// 300x) are k_getsv_cand_dense's, one workgroup each, with what a record asks the tables for in LDS; k_getsv_cand walks what is left, a workgroup per four tiles.
// (A wavefront per tile, round 4, lasted as long as 64 dependent rounds of record line -> look-up -> junctions / windows -> atomics on the few wavefronts with work.)
constexpr uint32_t GC_DENSE_MIN = 128; // candidates in a scan tile from which on the tile is k_getsv_cand_dense's (its depth pass goes through LDS)
constexpr int GC_COLS = 4096;            // columns of the genome a dense workgroup's difference array in LDS covers

constexpr int GC_NTB = GC_COLS / (1 << TILE_SHIFT) + 1; // genome tiles a dense workgroup's range touches

// What a dense tile's workgroup needs before it can look at a record, fetched by the listing pass - one THREAD per tile, 150 K chains of dependent loads side by
// side - instead of by the workgroup, whose life is that chain (PMC at 300x: 89 % of the wave cycles waiting, a read request back after ~930 clocks).
struct DenseTile {
	uint32_t so, n;         // the tile's staged candidates
	int32_t T0, P0;         // contig and position of the first of them: the workgroup's range starts there
	int64_t tb0;            // the genome tile of P0 (index into the tile map)
	uint32_t jlo, wlo;      // the first junction window / depth window a read that starts in the range can meet (n_junc / n_win: none)
	uint8_t tb[GC_NTB + 7 - (GC_NTB + 7) % 8]; // tile bits of the range's tiles
};

// the scan tiles that hold GC_DENSE_MIN candidates or more, listed (in any order: integer sums downstream); one atomic per workgroup (one per wavefront: 2,356 atomics
// with return on one address, 26 us)
__global__ __launch_bounds__(BLOCK) void k_dense_tiles(GetsvArgs a, GetsvStage g)
{
	__shared__ int s_w[WAVES_PER_BLOCK + 1];
	const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	const uint32_t n = t < g.ntiles ? g.tile_cnt[t] : 0u;
	const bool dense = n >= GC_DENSE_MIN;
	int total;
	const int at = block_exclusive_sum<int>(dense ? 1 : 0, s_w, &total);
	if (!total) return;
	if (threadIdx.x == 0) s_w[WAVES_PER_BLOCK] = atomicAdd(g.dense_n, total);
	__syncthreads();
	if (!dense) return;
	DenseTile e;
	e.so = g.tile_off[t]; e.n = n;
	const RecLine l0 = rec_load(a.b.rec, g.stage[e.so]);
	e.T0 = l0.tid(); e.P0 = l0.pos();
	e.tb0 = a.ctg_tile_off[e.T0] + (e.P0 >> TILE_SHIFT);
	const int64_t tb_end = a.ctg_tile_off[e.T0 + 1];
	// The range's first tile with junction windows knows the first junction that begins after (its start - wmax): no read of this contig that starts in the
	// range and has to look at junctions at all meets an earlier one.  The same for the depth windows: the first tile with any knows the first window that
	// ends at or after its first column (an earlier window that reaches into the range would have marked an earlier tile of it).
	int dj = -1, dd = -1;
#pragma unroll
	for (int i = GC_NTB - 1; i >= 0; --i) {
		const uint8_t m = e.tb0 + i < tb_end ? a.tilemap[e.tb0 + i] : (uint8_t)0;
		e.tb[i] = m;
		if (m & TM_JUNC) dj = i;
		if (m & TM_DEPTH) dd = i;
	}
	e.jlo = dj >= 0 ? a.tile_junc[e.tb0 + dj] : (uint32_t)a.n_junc;
	e.wlo = dd >= 0 ? a.tile_win[e.tb0 + dd] : (uint32_t)a.n_win;
	g.dense_list[s_w[WAVES_PER_BLOCK] + at] = e;
}

__global__ __launch_bounds__(BLOCK) void k_getsv_cand(GetsvArgs a, GetsvStage g)
{
	const int64_t t0 = (int64_t)blockIdx.x * WAVES_PER_BLOCK;
	uint32_t n[WAVES_PER_BLOCK], so[WAVES_PER_BLOCK], total = 0;
#pragma unroll
	for (int w = 0; w < WAVES_PER_BLOCK; ++w) {
		const bool in = t0 + w < g.ntiles;
		n[w] = in ? g.tile_cnt[t0 + w] : 0u; so[w] = in ? g.tile_off[t0 + w] : 0u;
		if (g.dense_list && n[w] >= GC_DENSE_MIN) n[w] = 0; // (k_getsv_cand_dense's)
		total += n[w];
	}
	// the four tiles' candidates as ONE list: at 30x a tile holds ~40, and a pass per tile was four times the chain of dependent loads (line -> tile bits ->
	// junctions / windows) with a sixth of the lanes at work
	const uint32_t c1 = n[0], c2 = c1 + n[1], c3 = c2 + n[2];
	for (uint32_t k = threadIdx.x; k < total; k += BLOCK) {
		const uint32_t at = k < c1 ? so[0] + k : (k < c2 ? so[1] + (k - c1) : (k < c3 ? so[2] + (k - c2) : so[3] + (k - c3)));
		const CandRec r = cand_load(a.b, g.stage[at]);
		const int tid = r.line.tid(), pos = r.line.pos();
		int64_t tile;
		const uint32_t m = getsv_tile_bits(a, tid, pos, tile);
		if (m & TM_JUNC) discordant_record(a, r, tid, pos, tile);
		if (m & TM_DEPTH) depth_record(a, r, tid, pos, tile);
	}
}

// A DENSE tile's candidates (k_dense_tiles lists the tiles with their context).  Depth: the per-read scheme's two global atomics per read and window - every one its
// own trip to the L2 - were a third of the 300x kernel (profiles/r05_config3_notes.txt); tally: a planted breakpoint's discordant pairs are atomics on ONE count.
// The records are position sorted, so a workgroup's reads cover a few thousand consecutive columns of one contig: their +1 / -1 go into a difference array over
// COLUMNS in LDS (from the first candidate's column on, GC_COLS wide; LDS atomics), and when all reads are in, every window that reaches into the range gets what
// the per-read scheme would have given it, one atomic per column that is not zero: its first column the running sum up to there (the reads that entered before
// it), the columns behind it the array's own entries, the slot behind its last column minus the reads that leave beyond it.  A stretch that does not fit the range
// (another contig, a long N skip) goes the per-read way.  Integer sums: the result is the same whatever the path.
constexpr int GC_JC = 32, GC_WC = 16;          // junction windows / depth windows a dense workgroup keeps in LDS (beyond: global memory)
// measured at 300x (tools/gpu_variants.sh; us): one workgroup per group of four tiles, 8192 columns, one line in flight 1506; a workgroup per TILE, 4096 columns 1311,
// two lines in flight 1160 (four: 1241); the tables in LDS and the first lines requested first 959; the junction counts summed in LDS 629.
constexpr int GC_U = 2;     // record lines a thread has in flight
__global__ __launch_bounds__(BLOCK) void k_getsv_cand_dense(GetsvArgs a, GetsvStage g)
{
	__shared__ int32_t s_l[GC_COLS];
	__shared__ DevJunction s_j[GC_JC];
	__shared__ int32_t s_jc[GC_JC];          // reads counted for the junctions in s_j
	__shared__ int32_t s_wtid[GC_WC], s_wbeg[GC_WC], s_wend[GC_WC];
	__shared__ int64_t s_woff[GC_WC];
	__shared__ uint8_t s_tb[GC_NTB];
	__shared__ int32_t s_chunk[BLOCK];       // running sum in front of every thread's columns
	__shared__ int32_t s_w[WAVES_PER_BLOCK + 1];
	if ((int)blockIdx.x >= *g.dense_n) return; // (the grid is sized by a bound of the list's length that the host knows)
	const DenseTile *ep = g.dense_list + blockIdx.x;
	const uint32_t n1 = ep->n, so1 = ep->so;
	// the first records' lines are asked for before anything else; junction and depth windows come into LDS side by side
	CandRec r[GC_U];
#pragma unroll
	for (int u = 0; u < GC_U; ++u) { const uint32_t k = u * BLOCK + threadIdx.x; r[u] = cand_load(a.b, g.stage[so1 + (k < n1 ? k : n1 - 1)]); }
	for (int i = (int)threadIdx.x; i < GC_COLS; i += BLOCK) s_l[i] = 0;
	// the range: from the first candidate's first column on (candidates are in file order: no later one starts before it on that contig)
	const int T0 = ep->T0, P0 = ep->P0, C0 = P0 + 1;
	const int64_t tb0 = ep->tb0, jlo = ep->jlo, wlo = ep->wlo;
	if (threadIdx.x < GC_NTB) s_tb[threadIdx.x] = ep->tb[threadIdx.x];
	if (threadIdx.x < GC_JC) { s_jc[threadIdx.x] = 0; if (jlo + threadIdx.x < a.n_junc) s_j[threadIdx.x] = a.junc[jlo + threadIdx.x]; }
	else if (threadIdx.x < GC_JC + GC_WC) {
		const int i = (int)threadIdx.x - GC_JC;
		const int64_t x = wlo + i;
		const bool in = x < a.n_win;
		s_wtid[i] = in ? a.win_tid[x] : -1; s_wbeg[i] = in ? a.win_beg[x] : 0; s_wend[i] = in ? a.win_end[x] : 0; s_woff[i] = in ? a.win_off[x] : 0;
	}
	__syncthreads();
	auto one = [&](const CandRec &r) {
		const int tid = r.line.tid(), pos = r.line.pos();
		int64_t tile;
		uint32_t m;
		const int dt = (pos >> TILE_SHIFT) - (P0 >> TILE_SHIFT);
		const bool here = tid == T0 && pos >= P0 && dt < GC_NTB;
		if (here) { m = s_tb[dt]; tile = tb0 + dt; }
		else m = getsv_tile_bits(a, tid, pos, tile);
		if ((m & TM_JUNC) && discordant_read(a, r)) {
			// (a planted breakpoint's hundred-odd discordant pairs at 300x all count for ONE junction: as global atomics on one address they were 0.3 ms of this kernel)
			if (here) discordant_walk(a, r, tid, pos, jlo, [&](int64_t x) { return x - jlo < GC_JC ? s_j[x - jlo] : a.junc[x]; },
			                          [&](int64_t x, const DevJunction &j) { if (x - jlo < GC_JC) atomicAdd(&s_jc[x - jlo], 1); else atomicAdd(&a.counts[j.orig], 1); });
			else discordant_record(a, r, tid, pos, tile);
		}
		if (!(m & TM_DEPTH) || !depth_counts(a, r)) return;
		int col = pos + 1; // 1-based
		int64_t wcur = -1;
		for (int q = 0; q < r.nc; ++q) {
			const uint32_t c = cand_op(r, q);
			const int op = (int)(c & 15u), len = (int)(c >> 4);
			if (op == C_M) {
				if (len > 0) {
					const int s = col, e = col + len - 1;
					if (tid == T0 && s >= C0 && e + 1 - C0 < GC_COLS) { atomicAdd(&s_l[s - C0], 1); atomicAdd(&s_l[e + 1 - C0], -1); }
					else depth_segment(a, tid, s, e, tile, wcur, 1);
				}
				col += len;
			} else if (op == C_D || op == C_N) col += len;
		}
	};
#pragma unroll
	for (int u = 0; u < GC_U; ++u) if (u * BLOCK + threadIdx.x < n1) one(r[u]);
	for (uint32_t k0 = BLOCK * GC_U; k0 < n1; k0 += BLOCK * GC_U) { // GC_U lines requested before the first is looked at
#pragma unroll
		for (int u = 0; u < GC_U; ++u) { const uint32_t k = k0 + u * BLOCK + threadIdx.x; r[u] = cand_load(a.b, g.stage[so1 + (k < n1 ? k : n1 - 1)]); }
#pragma unroll
		for (int u = 0; u < GC_U; ++u) if (k0 + u * BLOCK + threadIdx.x < n1) one(r[u]);
	}
	__syncthreads();
	if (threadIdx.x < GC_JC && s_jc[threadIdx.x]) atomicAdd(&a.counts[s_j[threadIdx.x].orig], s_jc[threadIdx.x]);
	// running sums: every thread its columns, a scan over the threads' totals
	constexpr int PER = GC_COLS / BLOCK;
	int32_t mine = 0;
#pragma unroll 8
	for (int j = 0; j < PER; ++j) mine += s_l[(int)threadIdx.x * PER + j];
	int32_t tot;
	s_chunk[threadIdx.x] = block_exclusive_sum(mine, s_w, &tot);
	__syncthreads();
	auto pre = [&](int c) -> int32_t { // sum of the array up to and including column offset c
		int32_t v = s_chunk[c / PER];
		for (int j = (c / PER) * PER; j <= c; ++j) v += s_l[j];
		return v;
	};
	// the windows that reach into [C0, C0 + GC_COLS) on T0 (from wlo on; the first GC_WC of them are in LDS)
	for (int64_t x = wlo; x < a.n_win; ++x) {
		const bool c = x - wlo < GC_WC;
		const int i = (int)(x - wlo);
		const int wt = c ? s_wtid[i] : a.win_tid[x], wb = c ? s_wbeg[i] : a.win_beg[x], we = c ? s_wend[i] : a.win_end[x];
		if (wt != T0 || (int64_t)wb >= (int64_t)C0 + GC_COLS) break;
		if (we < C0) continue;
		const int lo = wb > C0 ? wb : C0, hi = (int64_t)we < (int64_t)C0 + GC_COLS - 1 ? we : C0 + GC_COLS - 1;
		int32_t *d = a.diff + (c ? s_woff[i] : a.win_off[x]);
		for (int col = lo + (int)threadIdx.x; col <= hi; col += BLOCK) {
			const int32_t v = col == wb ? pre(col - C0) : s_l[col - C0];
			if (v) atomicAdd(&d[col - wb], v);
		}
		if (threadIdx.x == 0 && hi == we) { const int32_t v = pre(we - C0); if (v) atomicAdd(&d[we + 1 - wb], -v); }
	}
}

// K8: per window, running sum of the difference array -> per-column depth (in place); one wavefront per window, 256 columns a round: four coalesced loads
// issued together, then four scans chained through the carry (a round is a trip to memory and back; four columns per LANE - strided loads and stores - was slower)
__global__ __launch_bounds__(BLOCK) void k_depth_prefix(const int64_t *__restrict__ win_off, int64_t n_win, int32_t *__restrict__ diff, int32_t *__restrict__ max_depth)
{
	int64_t w = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave_id();
	if (w >= n_win) return;
	int32_t *d = diff + win_off[w];
	const int64_t len = win_off[w + 1] - win_off[w] - 1; // columns (the extra slot absorbs the closing -1)
	int32_t carry = 0, mx = 0;
	for (int64_t base = 0; base < len; base += WAVE * 4) {
		int32_t v[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) { const int64_t i = base + k * WAVE + lane_id(); v[k] = i < len ? d[i] : 0; }
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int64_t i = base + k * WAVE + lane_id();
			const int32_t inc = wave_inclusive_sum(v[k]) + carry;
			if (i < len) { d[i] = inc; mx = inc > mx ? inc : mx; }
			carry = __shfl(inc, 63, 64);
		}
	}
	mx = wave_max(mx);
	// 20 K windows hitting one address with an atomic each cost 0.2 ms (~90 same-address atomics per microsecond): look first, most waves lose
	if (lane_id() == 0 && mx > __atomic_load_n(max_depth, __ATOMIC_RELAXED)) atomicMax(max_depth, mx);
}

// the last window with (tid, beg) <= (tid, pos), or -1; the whole wavefront searches: 64 evenly spaced probes a round (three rounds for 20 K windows, where a
// binary search is fifteen dependent loads)
__device__ __forceinline__ int64_t window_at_or_before(const int32_t *__restrict__ win_tid, const int32_t *__restrict__ win_beg, int64_t n_win, int tid, int pos)
{
	int64_t lo = 0, hi = n_win; // the answer + 1 lies in [lo, hi]: windows below lo are <= the key, windows from hi on are greater
	while (hi - lo > 0) {
		const int64_t step = (hi - lo + WAVE - 1) / WAVE;
		const int64_t m = lo + (int64_t)lane_id() * step; // probes lo, lo + step, ...
		const bool le = m < hi && (win_tid[m] < tid || (win_tid[m] == tid && win_beg[m] <= pos));
		const uint64_t b = __ballot(le); // a prefix of the lanes (the windows are sorted)
		const int k = __popcll(b);       // probes that are <= the key
		if (k == 0) { hi = lo; break; }
		const int64_t last_le = lo + (int64_t)(k - 1) * step;
		lo = last_le + 1;
		hi = last_le + step < hi ? last_le + step : hi;
	}
	return lo - 1;
}

// one wavefront per query range: sum of depth over [beg, end] (the range lies inside one window)
__global__ __launch_bounds__(BLOCK) void k_range_sum(const int32_t *__restrict__ win_tid, const int32_t *__restrict__ win_beg, const int32_t *__restrict__ win_end,
                                                     const int64_t *__restrict__ win_off, int64_t n_win, const int32_t *__restrict__ depth,
                                                     const int32_t *__restrict__ q_tid, const int32_t *__restrict__ q_beg, const int32_t *__restrict__ q_end, int64_t n_q,
                                                     unsigned long long *__restrict__ out)
{
	int64_t r = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave_id();
	if (r >= n_q) return;
	const int tid = q_tid[r], beg = q_beg[r], end = q_end[r];
	const int64_t w = window_at_or_before(win_tid, win_beg, n_win, tid, beg);
	unsigned long long s = 0;
	if (w >= 0 && win_tid[w] == tid && beg <= win_end[w]) {
		int e = end < win_end[w] ? end : win_end[w];
		const int32_t *d = depth + win_off[w] - win_beg[w];
		for (int64_t c = (int64_t)beg + lane_id(); c <= e; c += WAVE) s += (unsigned long long)(uint32_t)d[c];
	}
	s = wave_sum(s);
	if (lane_id() == 0) out[r] = s;
}

__global__ void k_point_depth(const int32_t *__restrict__ win_tid, const int32_t *__restrict__ win_beg, const int32_t *__restrict__ win_end,
                              const int64_t *__restrict__ win_off, int64_t n_win, const int32_t *__restrict__ depth,
                              const int32_t *__restrict__ q_tid, const int32_t *__restrict__ q_pos, int64_t n_q, int32_t *__restrict__ out)
{
	int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_q) return;
	const int tid = q_tid[r], c = q_pos[r];
	int64_t lo = 0, hi = n_win;
	while (lo < hi) {
		int64_t m = (lo + hi) >> 1;
		int wt = win_tid[m];
		if (wt < tid || (wt == tid && win_beg[m] <= c)) lo = m + 1; else hi = m;
	}
	int v = 0;
	if (lo > 0) {
		int64_t w = lo - 1;
		if (win_tid[w] == tid && c <= win_end[w]) v = depth[win_off[w] + (c - win_beg[w])];
	}
	out[r] = v;
}

// ---------------------------------------------------------------------------------------------------------------------
// The read cap of the reference's pileup (libbam 0.1.16, bam_plp_push: `iter->tid == b->core.tid && iter->pos == b->core.pos &&
// iter->mp->cnt > iter->maxcnt`, maxcnt = 8000, two pool nodes always allocated).  In file order, among the reads that pass the depth
// filter: a read that is not the first at its start position is dropped when 2 + L > 8000, L = accepted reads of the contig whose
// bam_calend end (M, D, N) is >= the start.  Pinned by the real reference on tests/golden/getsv/deep.*.  The rule is sequential, but it
// can only bind where >= 7,998 earlier records start within one read span of a record ("deep" records: impossible below ~5,000x), so:
//   * the streaming pass raises a flag when a wavefront's records of a tile start within one span (necessary for an interior deep record);
//   * k_cap_mark then marks the 4096-record tiles that contain deep records (the first and last two tiles of a batch are always looked at:
//     their look-back windows reach into the previous batch, whose last 8192 records are kept as (tid, pos, end, pass) in a tail buffer);
//   * k_cap_sweep, ONE wavefront, replays the pileup's bookkeeping over the marked tiles and two tiles either side of them (a sweep that
//     starts >= 7,999 records before the first deep record starts from the right state: everything before is accepted and no longer alive
//     there), and takes the contributions of the dropped reads out of the difference arrays again.  A sweep that is still running at the end
//     of a batch is carried into the next one (state + ring of ends in ctx memory).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int CAP_MAXCNT = 8000, CAP_LOOKBACK = 7998, CAP_TAIL = 8192, CAP_LDS_RING = 8192;

struct CapCarry { int32_t active, tid, pos, live, since_deep, pad[3]; };

struct CapArgs {
	GetsvArgs g;
	int32_t span;                 // >= the longest reference span of any record so far
	const int32_t *tail_tid, *tail_pos, *tail_end; // the last tail_n records of the stream before this batch
	const uint8_t *tail_pass;
	int32_t tail_n;
	uint8_t *deep;                // [ntiles] per CS_TILE records
	int64_t ntiles;
	int *flags;                   // [0] raised by the streaming pass, [1] raised by k_cap_mark
	CapCarry *carry;
	int32_t *ring;                // accepted reads by (end & ring_mask): carried state, and the working ring when it does not fit in LDS
	int32_t ring_mask;
	// next tail (k_cap_tail)
	int32_t *ntail_tid, *ntail_pos, *ntail_end;
	uint8_t *ntail_pass;
	int32_t ntail_n;
	int32_t prime;                // ssv_getsv_prime: the batch only rebuilds the pileup's state (it was counted by another rank): dropped reads give nothing back
};

__device__ __forceinline__ bool cap_pass(const GetsvArgs &a, const RecLine &r) // the depth pass's read filter (read_bam + BAM_DEF_MASK)
{
	return r.mapq() >= a.depth_min_mapq && !(r.flag() & (F_UNMAP | F_SECONDARY | F_QCFAIL | F_DUP)) && r.tid() >= 0;
}

__device__ __forceinline__ int cap_calend(const GetsvArgs &a, const RecLine &r) // bam_calend: M, D, N advance
{
	int end = r.pos();
	const int n = r.n_cigar();
	for (int k = 0; k < n; ++k) { const uint32_t c = r.op(a.b.cigar, k); const int op = (int)(c & 15u); if (op == C_M || op == C_D || op == C_N) end += (int)(c >> 4); }
	return end;
}

// the ring of read ends grows (a batch brought a read with a longer reference span): a carried sweep's live entries - ends in
// [pos, pos + old size) - move to their slots in the larger, zeroed ring
__global__ __launch_bounds__(BLOCK) void k_cap_regrow(const CapCarry *__restrict__ carry, const int32_t *__restrict__ old_ring, int32_t old_mask, int32_t *__restrict__ new_ring, int32_t new_mask)
{
	if (!carry->active || carry->pos < 0) return;
	const int64_t pos = carry->pos;
	for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= old_mask; k += (int64_t)gridDim.x * blockDim.x) {
		const int64_t e = pos + k;
		new_ring[e & new_mask] = old_ring[e & old_mask];
	}
}

// per CS_TILE records: does the tile hold a deep record?  (grid-stride over the tiles: in the usual case every tile but four leaves at once)
__global__ __launch_bounds__(BLOCK) void k_cap_mark(CapArgs c)
{
	const DevBatch &b = c.g.b;
	const bool all = c.flags[0] || c.carry->active;
	for (int64_t t = blockIdx.x; t < c.ntiles; t += gridDim.x) {
		const bool boundary = t < 2 || t + 2 >= c.ntiles;
		if (!(boundary || all)) { if (threadIdx.x == 0) c.deep[t] = 0; continue; }
		bool deep = false;
#pragma unroll
		for (int k = 0; k < CS_TILE / BLOCK; ++k) { // branch-free: all loads of the tile in flight at once
			const int64_t i = t * CS_TILE + (int64_t)k * BLOCK + threadIdx.x;
			const bool in = i < b.n;
			const int64_t ic = in ? i : 0;
			const int tid = b.tid[ic], pos = b.pos[ic];
			const int64_t j = ic - CAP_LOOKBACK, jt = (int64_t)c.tail_n + j;
			const bool from_batch = j >= 0, from_tail = !from_batch && jt >= 0;
			const int ptid = from_batch ? b.tid[j] : (from_tail ? c.tail_tid[jt] : -1);
			const int ppos = from_batch ? b.pos[j] : (from_tail ? c.tail_pos[jt] : 0);
			deep = deep || (in && tid >= 0 && ptid == tid && pos - ppos <= c.span);
		}
		const int any = __syncthreads_or(deep ? 1 : 0);
		if (threadIdx.x == 0) { c.deep[t] = any ? 1 : 0; if (any) c.flags[1] = 1; }
	}
}

// the pileup's bookkeeping, replayed by one wavefront (every lane keeps the same scalar state)
__global__ __launch_bounds__(WAVE) void k_cap_sweep(CapArgs c)
{
	__shared__ int32_t s_ring[CAP_LDS_RING];
	if (!(c.flags[0] | c.flags[1] | c.carry->active)) return;
	const GetsvArgs &a = c.g;
	const int lane = lane_id();
	const bool use_lds = c.ring_mask < CAP_LDS_RING;
	int32_t *ring = use_lds ? s_ring : c.ring;
	const int mask = c.ring_mask;
	bool active = c.carry->active != 0;
	int s_tid = c.carry->tid, s_pos = c.carry->pos, live = c.carry->live, since_deep = c.carry->since_deep;
	if (active && use_lds) for (int x = lane; x <= mask; x += WAVE) s_ring[x] = c.ring[x];
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	auto clear_ring = [&]() {
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		for (int x = lane; x <= mask; x += WAVE) ring[x] = 0;
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	};
	auto is_deep = [&](int64_t t) -> bool { return t >= 0 && t < c.ntiles && c.deep[t] != 0; };
	for (int64_t t = active ? 0 : -2; t < c.ntiles; ++t) {
		const bool need = is_deep(t) || is_deep(t + 1) || is_deep(t + 2) || (active && since_deep < 2);
		if (!need) { active = false; continue; }
		if (!active) { clear_ring(); live = 0; s_tid = -1; s_pos = -1; since_deep = 2; active = true; } // a sweep starts: nothing before this tile is alive at the first deep record
		for (int base = 0; base < CS_TILE; base += WAVE) {
			const int64_t i = t * CS_TILE + base + lane; // record index in the batch; negative: the previous records of the stream
			bool pass = false;
			int tid = -1, pos = 0, end = 0;
			if (i >= 0) {
				if (i < a.b.n) { const RecLine rl = rec_load(a.b.rec, i); if (cap_pass(a, rl)) { pass = true; tid = rl.tid(); pos = rl.pos(); end = cap_calend(a, rl); } }
			} else if ((int64_t)c.tail_n + i >= 0) {
				const int64_t j = (int64_t)c.tail_n + i;
				pass = c.tail_pass[j] != 0; tid = c.tail_tid[j]; pos = c.tail_pos[j]; end = c.tail_end[j];
			}
			const uint64_t any = __ballot(pass);
			if (!any) continue;
			uint64_t dropped = 0;
			// reads with the same start are decided together: in a stack there are a few starts per 64 records, not 64 decisions
			uint64_t todo = any;
			while (todo) {
				const int l = __ffsll((long long)todo) - 1;
				const int r_tid = __shfl(tid, l, WAVE), r_pos = __shfl(pos, l, WAVE);
				const uint64_t grp = __ballot(pass && tid == r_tid && pos == r_pos) & todo; // sorted input: the group is what is left of this start
				todo &= ~grp;
				if (r_tid != s_tid) { clear_ring(); live = 0; s_tid = r_tid; s_pos = -1; }
				int exempt = 0; // the first read at a new start is always taken
				if (r_pos != s_pos) {
					// the columns before r_pos have been emitted: reads that ended there are gone
					if (s_pos >= 0 && r_pos - s_pos <= mask) { for (int e = s_pos; e < r_pos; ++e) { live -= ring[e & mask]; ring[e & mask] = 0; } }
					else if (s_pos >= 0) { clear_ring(); live = 0; }
					s_pos = r_pos;
					exempt = 1;
				}
				// in file order: a read is taken while 2 + live <= CAP_MAXCNT (bam_plp_push); every taken read with a reference span adds one
				// to live.  Reads without any M / D / N operation (end == pos) do not: if the group holds one, decide it read by read.
				const bool in_grp = (grp >> lane) & 1ull;
				const bool spanless = __any(in_grp && end <= pos);
				if (!spanless) {
					const int n_g = (int)__popcll(grp);
					const int room_after_first = CAP_MAXCNT - 2 - (live + exempt) + 1; // reads that can still be taken once the exempt one is in
					int take = exempt + (room_after_first > 0 ? (n_g - exempt < room_after_first ? n_g - exempt : room_after_first) : 0);
					if (take > n_g) take = n_g;
					const int rank = (int)__popcll(grp & lanemask_lt());
					if (in_grp) {
						if (rank < take) atomicAdd(&ring[end & mask], 1);
						else dropped |= 1ull << lane;
					}
					live += take;
					__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				} else {
					uint64_t g = grp;
					bool first = exempt != 0;
					while (g) {
						const int q = __ffsll((long long)g) - 1;
						g &= g - 1;
						const int q_end = __shfl(end, q, WAVE);
						const bool accept = first || !(2 + live > CAP_MAXCNT);
						first = false;
						if (accept) { if (q_end > r_pos) { if (lane == q) ring[q_end & mask] += 1; ++live; } }
						else if (lane == q) dropped |= 1ull << lane;
						__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
					}
				}
			}
			dropped = __ballot((dropped >> lane) & 1ull); // every lane decided for itself above
			// a dropped read never reached the pileup: take its coverage out of the difference arrays again
			if (!c.prime && i >= 0 && ((dropped >> lane) & 1ull)) {
				int64_t tile;
				const uint32_t m = getsv_tile_bits(a, tid, pos, tile);
				if (m & TM_DEPTH) depth_record(a, cand_load(a.b, i), tid, pos, tile, -1);
			}
		}
		since_deep = is_deep(t) ? 0 : since_deep + 1;
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	if (active && use_lds) for (int x = lane; x <= mask; x += WAVE) c.ring[x] = s_ring[x];
	if (lane == 0) { c.carry->active = active ? 1 : 0; c.carry->tid = s_tid; c.carry->pos = s_pos; c.carry->live = live; c.carry->since_deep = since_deep; }
}

// the last (up to) CAP_TAIL records of the stream including this batch, for the next batch's look-backs and sweep starts
__global__ void k_cap_tail(CapArgs c)
{
	const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (j >= c.ntail_n) return;
	const int64_t s = (int64_t)c.tail_n + c.g.b.n - c.ntail_n + j; // index in (old tail ++ batch)
	if (s < c.tail_n) { c.ntail_tid[j] = c.tail_tid[s]; c.ntail_pos[j] = c.tail_pos[s]; c.ntail_end[j] = c.tail_end[s]; c.ntail_pass[j] = c.tail_pass[s]; }
	else {
		const int64_t i = s - c.tail_n;
		const RecLine rl = rec_load(c.g.b.rec, i);
		const bool p = cap_pass(c.g, rl);
		c.ntail_tid[j] = rl.tid(); c.ntail_pos[j] = rl.pos(); c.ntail_end[j] = p ? cap_calend(c.g, rl) : rl.pos(); c.ntail_pass[j] = p ? 1 : 0;
	}
}

} // namespace ssv
