// realign_kernels.h - clipped-sequence re-aligner on the GPU.
//
// The reference pipeline aligns the clipped sequences of `seeksv getclip` (clip.fq.gz, read name = sequence) with an EXTERNAL
// tool, `bwa mem` (README.md:22-34, example/seeksv.sh:3), and `seeksv getsv` reads the resulting clip.bam: per record it uses
// flag & {4, 16, 256}, MAPQ == 0 or not, tid, pos, the CIGAR with its S/H ends and the read name (getsv.cpp:25-71,
// getsv.h:445-527).  This is a stand-in for that step on hosts without bwa, for references that behave like random sequence
// (the synthetic genomes of bench / tests): a k-mer index of the reference in HBM, seed look-ups for every k-mer of a query on
// both strands, ungapped extension with bwa mem's default scores (match 1, mismatch 4, end clipping 5, report >= 30).
// It does not reproduce bwa's alignments bit for bit (no gapped extension, no chaining, no supplementary records; at most
// RA_MAX_CAND seed hits per query are followed, probe runs longer than RA_MAX_PROBE are cut).
//
// Index: every SAMPLE-th reference position p whose K-mer lies inside one contig is a slot of an open-addressing table
// (u32 slot = p / SAMPLE + 1, 0 = empty; linear probing; load <= 1/2); a slot does not hold its key - a look-up checks the K-mer
// at the slot's position against the 2-bit reference it has to read anyway.  Query: one wavefront per sequence.
#pragma once

#include "common.h"

namespace ssv {

constexpr int RA_K = 20;              // seed length (bwa mem's minimum seed length is 19)
constexpr int RA_SAMPLE = 4;          // indexed reference positions: p % 4 == 0; a query tries every offset, so any match of >= K + 3 bases has a seed
constexpr int RA_MAX_PROBE = 256;     // longest probe run followed / built (low-complexity sequence: the surplus is dropped)
constexpr int RA_MAX_CAND = 192;      // candidate diagonals per query kept for extension
constexpr int RA_MAX_Q = 1024;        // longest query (longer ones are reported unaligned)
constexpr int RA_MATCH = 1, RA_MISMATCH = 4, RA_CLIP = 5, RA_MIN_SCORE = 30;

struct RaIndex {
	const uint64_t *ref;      // 2-bit bases, base i at bits [2 (i % 32), +2) of word i / 32; A C G T = 0 1 2 3
	int64_t n_bases;
	const int64_t *ctg_off;   // [n_ctg + 1] first base of each contig
	int32_t n_ctg;
	uint32_t *table;
	uint64_t mask;            // slots - 1
};

__device__ __forceinline__ uint64_t ra_kmer_at(const uint64_t *ref, int64_t p)
{
	const int64_t w = p >> 5;
	const int sh = (int)(p & 31) * 2;
	uint64_t lo = ref[w] >> sh;
	if (sh > 64 - 2 * RA_K) lo |= ref[w + 1] << (64 - sh); // the reference array has one word of slack
	return lo & ((1ull << (2 * RA_K)) - 1ull);
}

__device__ __forceinline__ uint32_t ra_base_at(const uint64_t *ref, int64_t p) { return (uint32_t)(ref[p >> 5] >> ((p & 31) * 2)) & 3u; }

__device__ __forceinline__ uint64_t ra_hash(uint64_t k)
{
	k *= 0x9E3779B97F4A7C15ull;
	k ^= k >> 29;
	k *= 0xBF58476D1CE4E5B9ull;
	return k ^ (k >> 32);
}

__device__ __forceinline__ int ra_contig_of(const RaIndex &ix, int64_t p)
{
	int lo = 0, hi = ix.n_ctg - 1;
	while (lo < hi) {
		int m = (lo + hi + 1) >> 1;
		if (ix.ctg_off[m] <= p) lo = m; else hi = m - 1;
	}
	return lo;
}

// one thread per sampled position
__global__ __launch_bounds__(BLOCK) void k_ra_build(RaIndex ix, unsigned long long *dropped)
{
	const int64_t s = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	const int64_t p = s * RA_SAMPLE;
	if (p + RA_K > ix.n_bases) return;
	const int t = ra_contig_of(ix, p);
	if (p + RA_K > ix.ctg_off[t + 1]) return; // the K-mer would run into the next contig
	uint64_t slot = ra_hash(ra_kmer_at(ix.ref, p)) & ix.mask;
	for (int probe = 0; probe < RA_MAX_PROBE; ++probe) {
		if (atomicCAS(&ix.table[slot], 0u, (uint32_t)(s + 1)) == 0u) return;
		slot = (slot + 1) & ix.mask;
	}
	atomicAdd(dropped, 1ull);
}

struct RaHit { // = ssv_realign_hit
	int32_t tid, pos;
	int32_t q_beg, q_end;
	int32_t score, second;
	int32_t n_mismatch;
	uint8_t reverse, mapq, pad[2];
};

struct RaQueryArgs {
	RaIndex ix;
	const char *seqs;         // concatenated ASCII sequences
	const uint64_t *seq_off;  // [n + 1]
	int64_t n;
	RaHit *hits;
};

// ASCII -> 2-bit code, 4 for anything else (never matches)
__device__ __forceinline__ uint32_t ra_code(char ch)
{
	switch (ch) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': return 3;
	default: return 4;
	}
}

// One wavefront per query.  1. codes of both orientations into LDS.  2. every lane takes K-mer offsets lane, lane + 64, ... of both
// orientations, probes the table and records the diagonals (reference position of query base 0) of verified seeds.  3. duplicates
// out (a matching stretch of m bases yields ~(m - K) / SAMPLE seeds on one diagonal).  4. one lane per distinct diagonal scores the
// whole query along it: best local segment under +1 / -4, extended to an end of the query when that loses less than the clipping
// penalty (bwa mem's rule).  5. best and second best locus -> hit.
__global__ __launch_bounds__(BLOCK) void k_ra_query(RaQueryArgs a)
{
	__shared__ uint8_t s_code[WAVES_PER_BLOCK][2][RA_MAX_Q];
	__shared__ int64_t s_diag[WAVES_PER_BLOCK][RA_MAX_CAND];
	__shared__ uint16_t s_so[WAVES_PER_BLOCK][RA_MAX_CAND]; // seed offset in the query | strand << 15
	__shared__ uint16_t s_tid[WAVES_PER_BLOCK][RA_MAX_CAND]; // contig of the seed (a query can hang over a contig's end: same diagonal, two contigs)
	__shared__ int s_n[WAVES_PER_BLOCK];
	const int w = wave_id(), lane = lane_id();
	const int64_t q = (int64_t)blockIdx.x * WAVES_PER_BLOCK + w;
	if (q >= a.n) return;
	const RaIndex &ix = a.ix;
	const uint64_t o0 = a.seq_off[q];
	const int n = (int)(a.seq_off[q + 1] - o0);
	RaHit out;
	out.tid = -1; out.pos = -1; out.q_beg = 0; out.q_end = 0; out.score = 0; out.second = 0; out.n_mismatch = 0; out.reverse = 0; out.mapq = 0; out.pad[0] = out.pad[1] = 0;
	if (n < RA_K || n > RA_MAX_Q) { if (lane == 0) a.hits[q] = out; return; }
	for (int i = lane; i < n; i += WAVE) {
		const uint32_t c = ra_code(a.seqs[o0 + i]);
		s_code[w][0][i] = (uint8_t)c;
		s_code[w][1][n - 1 - i] = (uint8_t)(c < 4 ? 3 - c : 4);
	}
	if (lane == 0) s_n[w] = 0;
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	// ---- seeds ----
	for (int st = 0; st < 2; ++st) {
		const uint8_t *code = s_code[w][st];
		for (int o = lane; o + RA_K <= n; o += WAVE) {
			uint64_t km = 0;
			bool ok = true;
			for (int k = 0; k < RA_K; ++k) { const uint32_t c = code[o + k]; ok = ok && c < 4; km |= (uint64_t)(c & 3u) << (2 * k); }
			if (!ok) continue;
			uint64_t slot = ra_hash(km) & ix.mask;
			for (int probe = 0; probe < RA_MAX_PROBE; ++probe) {
				const uint32_t v = ix.table[slot];
				if (v == 0u) break;
				const int64_t p = (int64_t)(v - 1u) * RA_SAMPLE;
				if (ra_kmer_at(ix.ref, p) == km) {
					const int at = atomicAdd(&s_n[w], 1);
					if (at < RA_MAX_CAND) { s_diag[w][at] = p - o; s_so[w][at] = (uint16_t)(o | (st << 15)); s_tid[w][at] = (uint16_t)ra_contig_of(ix, p); }
				}
				slot = (slot + 1) & ix.mask;
			}
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	const int m = s_n[w] < RA_MAX_CAND ? s_n[w] : RA_MAX_CAND;
	// ---- distinct diagonals, scored one per lane ----
	int best_score = 0, best_beg = 0, best_end = 0, best_mm = 0, best_st = 0, best_tid = -1;
	int64_t best_diag = 0;
	for (int c0 = 0; c0 < m; c0 += WAVE) {
		const int c = c0 + lane;
		bool mine = c < m;
		int64_t d = 0;
		int st = 0, t = 0;
		if (mine) {
			d = s_diag[w][c]; st = s_so[w][c] >> 15; t = s_tid[w][c];
			for (int e = 0; e < c; ++e) if (s_diag[w][e] == d && (s_so[w][e] >> 15) == st && s_tid[w][e] == t) { mine = false; break; }
		}
		if (!mine) continue;
		const uint8_t *code = s_code[w][st];
		// the query is scored inside the contig of the seed only (a seed never spans two contigs)
		const int64_t c_lo = ix.ctg_off[t], c_hi = ix.ctg_off[t + 1];
		int i_lo = (int)(c_lo - d > 0 ? c_lo - d : 0), i_hi = (int)(c_hi - d < n ? c_hi - d : n); // query positions inside the contig
		if (i_hi - i_lo < RA_K) continue;
		// Kadane over [i_lo, i_hi): best segment; prefix scores for the end-extension rule
		int run = 0, run_beg = i_lo, bs = 0, bb = i_lo, be = i_lo;
		for (int i = i_lo; i < i_hi; ++i) {
			const bool eq = code[i] < 4 && (uint32_t)code[i] == ra_base_at(ix.ref, d + i);
			if (run <= 0) { run = 0; run_beg = i; }
			run += eq ? RA_MATCH : -RA_MISMATCH;
			if (run > bs) { bs = run; bb = run_beg; be = i + 1; }
		}
		if (bs < RA_MIN_SCORE) continue;
		// extension to the query's ends (only possible when the end lies inside the contig)
		if (bb > i_lo || be < i_hi) {
			int sc = 0;
			if (bb > 0 && i_lo == 0) {
				for (int i = bb - 1; i >= 0; --i) sc += (code[i] < 4 && (uint32_t)code[i] == ra_base_at(ix.ref, d + i)) ? RA_MATCH : -RA_MISMATCH;
				if (sc > -RA_CLIP) { bs += sc; bb = 0; }
			}
			sc = 0;
			if (be < n && i_hi == n) {
				for (int i = be; i < n; ++i) sc += (code[i] < 4 && (uint32_t)code[i] == ra_base_at(ix.ref, d + i)) ? RA_MATCH : -RA_MISMATCH;
				if (sc > -RA_CLIP) { bs += sc; be = n; }
			}
		}
		int mm = 0;
		for (int i = bb; i < be; ++i) mm += (code[i] < 4 && (uint32_t)code[i] == ra_base_at(ix.ref, d + i)) ? 0 : 1;
		const bool better = bs > best_score || (bs == best_score && best_tid >= 0 && (st < best_st || (st == best_st && d < best_diag)));
		if (better) { best_score = bs; best_beg = bb; best_end = be; best_mm = mm; best_st = st; best_diag = d; best_tid = t; }
	}
	// ---- best locus over the lanes; second best = best score among diagonals that are not the winner's neighbourhood ----
	int win_score = best_score, win_lane = lane;
	int64_t win_diag = best_diag;
	int win_st = best_st;
#pragma unroll
	for (int dlt = 32; dlt >= 1; dlt >>= 1) {
		const int os = __shfl_xor(win_score, dlt, 64), ol = __shfl_xor(win_lane, dlt, 64), ost = __shfl_xor(win_st, dlt, 64);
		const int64_t od = __shfl_xor(win_diag, dlt, 64);
		const bool take = os > win_score || (os == win_score && os > 0 && (ost < win_st || (ost == win_st && (od < win_diag || (od == win_diag && ol < win_lane)))));
		if (take) { win_score = os; win_lane = ol; win_diag = od; win_st = ost; }
	}
	if (win_score < RA_MIN_SCORE) { if (lane == 0) a.hits[q] = out; return; }
	const int win_tid = __shfl(best_tid, win_lane, 64);
	const bool same_locus = best_score > 0 && best_st == win_st && best_tid == win_tid && (best_diag - win_diag <= 32 && win_diag - best_diag <= 32);
	int second = (best_score > 0 && !same_locus) ? best_score : 0;
	second = wave_max(second);
	if (lane == win_lane) {
		out.tid = best_tid;
		out.pos = (int32_t)(best_diag + best_beg - ix.ctg_off[best_tid]);
		out.q_beg = best_beg; out.q_end = best_end; out.score = best_score; out.second = second; out.n_mismatch = best_mm;
		out.reverse = (uint8_t)best_st;
		const int gap = best_score - second;
		out.mapq = (uint8_t)(second >= best_score ? 0 : (gap >= 10 ? 60 : (gap * 6 > 1 ? gap * 6 : 1)));
		a.hits[q] = out;
	}
}

} // namespace ssv
