// synth_core.h - deterministic synthetic BAM-record generator (bench / test tooling, not the product path).
// Counter-based: every field of record g is a pure function of (config, g), so any shard [g0, g0+n) can be
// produced independently on any rank, on the GPU (synth.hip) or on the CPU (synth_cpu.cpp) with identical bytes.
// Follows SURVEY.md 8(d): coordinate-sorted 150 bp paired-end reads over hash-generated contigs, MAPQ / DUP /
// SECONDARY / indel / soft-clip mixtures, and planted structural variants whose spanning reads are soft clipped
// at the breakpoint (clipped bases = partner-side reference) and whose spanning pairs are discordant.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define SY_HD __host__ __device__ __forceinline__
#else
#define SY_HD static inline
#endif

#define SY_MAX_CONTIGS 64

typedef struct {
	uint64_t seed;
	int64_t n_total;              // records of the whole synthetic BAM
	int32_t n_contigs;
	int32_t read_len;             // 150
	int64_t contig_off[SY_MAX_CONTIGS + 1]; // linear genome offsets
	int32_t contig_len[SY_MAX_CONTIGS];
	uint64_t spacing_fp;          // start spacing in 1/2^20 bp
	int32_t clip_permille;        // random soft clips (C2: 10)
	int32_t indel_permille;       // 20
	int32_t dup_permille;         // 80
	int32_t sec_permille;         // 2
	int32_t improper_permille;    // 20: background pairs that are not concordant
	int32_t vaf_permille;         // 500
	int32_t n_breakends;
	int32_t qual_model;           // 0: five binned values {2, 11, 25, 37, 40} (NovaSeq-like); 1: forty values 2..41, skewed to the high end (HiSeq-like)
	int32_t unmap_permille;       // records (g, g + 1), g even, that are a pair with ONE unmapped end: g the mapped read (MUNMAP), g + 1 its unmapped mate (UNMAP) placed
	                              // beside it as aligners do - the input of getclip's unmapped-pair side channel (clip_reads.h:415-419); 0: none (real WGS BAMs: 10-30)
} sy_config;

// one side of a planted junction, sorted by lin
typedef struct {
	int64_t lin;      // linear coordinate of q
	int32_t tid;
	int32_t q;        // 0-based reference index: side 0 -> aligned part is [.., q), side 1 -> aligned part is [q, ..)
	int32_t ptid;     // partner: bases visited when walking away from the aligned part through the junction
	int32_t ppos;     //   start at ppos and move by pdir
	int8_t side;
	int8_t pdir;
	int8_t mate_rev;  // discordant mates: mate strand flag (1 = MREVERSE)
	int8_t is_up;     // 1: this breakend is the junction's "up" end (discordant pairs are generated on this side)
	int32_t mate_anchor; // 0-based position the mate cluster is anchored on (down_pos - 1)
} sy_breakend;

typedef struct {
	int32_t tid, pos, l_qseq, mtid, mpos, isize;
	uint16_t flag;
	uint8_t mapq;
	uint8_t n_cigar;
	uint32_t cigar[3];
	// soft-clip description (has_seq != 0 -> bases and qualities are shipped)
	uint8_t has_seq;
	int8_t be_side;      // -1 random clip, 0/1 planted breakend side
	int32_t clip_len;
	int32_t be_index;    // breakend index or -1
} sy_record;

SY_HD uint64_t sy_mix(uint64_t x)
{
	x += 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}

SY_HD uint64_t sy_hash(uint64_t seed, uint64_t a, uint64_t b) { return sy_mix(sy_mix(seed ^ (a * 0xD6E8FEB86659FD93ull)) ^ (b * 0xCA5A826395121157ull)); }

// reference base (BAM nibble code A=1 C=2 G=4 T=8) at (tid, p)
SY_HD uint32_t sy_ref_base(const sy_config *c, int32_t tid, int64_t p) { return 1u << (sy_hash(c->seed ^ 0x5245464241534531ull, (uint64_t)tid, (uint64_t)p) & 3u); }
SY_HD uint32_t sy_comp(uint32_t b) { return ((b & 1u) << 3) | ((b & 2u) << 1) | ((b & 4u) >> 1) | ((b & 8u) >> 3); }

SY_HD int32_t sy_contig_of(const sy_config *c, int64_t x)
{
	int32_t lo = 0, hi = c->n_contigs - 1;
	while (lo < hi) {
		int32_t m = (lo + hi + 1) >> 1;
		if (c->contig_off[m] <= x) lo = m; else hi = m - 1;
	}
	return lo;
}

// last breakend with lin <= x, or -1
SY_HD int32_t sy_breakend_at_or_before(const sy_breakend *be, int32_t n, int64_t x)
{
	int32_t lo = 0, hi = n;
	while (lo < hi) {
		int32_t m = (lo + hi) >> 1;
		if (be[m].lin <= x) lo = m + 1; else hi = m;
	}
	return lo - 1;
}

#define SY_SNAP 72 /* records starting within SNAP bp before a side-1 breakend are moved onto it (keeps the stream sorted) */

SY_HD void sy_decide(const sy_config *c, const sy_breakend *be, int64_t g, sy_record *r)
{
	const int32_t L = c->read_len;
	int64_t x = (int64_t)(((uint64_t)g * c->spacing_fp) >> 20);
	int32_t tid = sy_contig_of(c, x);
	int64_t p = x - c->contig_off[tid];
	if (p > (int64_t)c->contig_len[tid] - L - 8) p = (int64_t)c->contig_len[tid] - L - 8;
	if (p < 0) p = 0;
	x = c->contig_off[tid] + p;
	const uint64_t h0 = sy_hash(c->seed, (uint64_t)g, 1), h1 = sy_hash(c->seed, (uint64_t)g, 2), h2 = sy_hash(c->seed, (uint64_t)g, 3);
	// ---- pair geometry ----
	int32_t s4 = (int32_t)(h0 & 0xffff) + (int32_t)((h0 >> 16) & 0xffff) + (int32_t)((h0 >> 32) & 0xffff) + (int32_t)((h0 >> 48) & 0xffff);
	int32_t T = 350 + (int32_t)(((int64_t)(s4 - 131070) * 50) / 37837); // ~N(350, 50^2), integer only
	if (T < L) T = L;
	const int fwd = (int)(h1 & 1u), read1 = (int)((h1 >> 1) & 1u);
	uint32_t flag = 1u | 2u | (fwd ? 32u : 16u) | (read1 ? 64u : 128u);
	int32_t mtid = tid, mpos, isize;
	if (fwd) { mpos = (int32_t)p + T - L; isize = T; }
	else { mpos = (int32_t)p - T + L; isize = -T; if (mpos < 0) { mpos = 0; } }
	const uint32_t u_mapq = (uint32_t)((h1 >> 8) % 1000u), u_dup = (uint32_t)((h1 >> 20) % 1000u), u_sec = (uint32_t)((h1 >> 32) % 1000u), u_imp = (uint32_t)((h1 >> 44) % 1000u);
	uint32_t mapq = 60;
	if (u_mapq < 30) mapq = 0; else if (u_mapq < 70) mapq = 1 + (uint32_t)((h1 >> 54) % 59u);
	if ((int32_t)u_dup < c->dup_permille) flag |= 1024u;
	if ((int32_t)u_sec < c->sec_permille) flag |= 256u;
	if ((int32_t)u_imp < c->improper_permille) {
		// background non-concordant pair: wrong orientation, far mate or mate on another contig
		flag &= ~2u;
		uint32_t kind = (uint32_t)((h2 >> 3) % 3u);
		if (kind == 0) { flag = (flag & ~48u) | (fwd ? 0u : 48u); }   // ++ or --
		else if (kind == 1) { int32_t far = 2000 + (int32_t)((h2 >> 8) % 100000u); mpos = (int32_t)p + (fwd ? far : -far); if (mpos < 0) mpos = 0; isize = mpos - (int32_t)p; }
		else { mtid = (int32_t)((uint32_t)(tid + 1 + (int32_t)((h2 >> 8) % (uint32_t)(c->n_contigs > 1 ? c->n_contigs - 1 : 1))) % (uint32_t)c->n_contigs); mpos = (int32_t)((h2 >> 20) % (uint64_t)(c->contig_len[mtid] - L)); isize = 0; }
	}
	// ---- CIGAR ----
	r->n_cigar = 1; r->cigar[0] = ((uint32_t)L << 4) | 0u; r->cigar[1] = r->cigar[2] = 0;
	r->has_seq = 0; r->be_side = -1; r->clip_len = 0; r->be_index = -1;
	int planted = 0;
	if (c->n_breakends > 0) {
		// side-1 breakend shortly after the start: snap onto it; a VAF share of the snapped reads are left-clipped variant reads
		int32_t k = sy_breakend_at_or_before(be, c->n_breakends, x + SY_SNAP);
		if (k >= 0 && be[k].side == 1 && be[k].tid == tid && be[k].lin > x && be[k].lin - x <= SY_SNAP) {
			int32_t clip = (int32_t)(be[k].lin - x);
			p = be[k].q; x = be[k].lin;
			if ((int32_t)(h2 % 1000u) < c->vaf_permille && clip >= 5) {
				r->n_cigar = 2; r->cigar[0] = ((uint32_t)clip << 4) | 4u; r->cigar[1] = ((uint32_t)(L - clip) << 4) | 0u;
				r->has_seq = 1; r->be_side = 1; r->clip_len = clip; r->be_index = k; planted = 1;
			} else planted = 2; // snapped, reference allele
			if (fwd) mpos = (int32_t)p + T - L; else { mpos = (int32_t)p - T + L; if (mpos < 0) mpos = 0; }
		}
		if (!planted) {
			// side-0 breakend inside the read: a VAF share become right-clipped variant reads
			int32_t k0 = sy_breakend_at_or_before(be, c->n_breakends, x + L - 5);
			if (k0 >= 0 && be[k0].side == 0 && be[k0].tid == tid && be[k0].lin >= x + 5 && (int32_t)(h2 % 1000u) < c->vaf_permille) {
				int32_t al = (int32_t)(be[k0].lin - x);
				r->n_cigar = 2; r->cigar[0] = ((uint32_t)al << 4) | 0u; r->cigar[1] = ((uint32_t)(L - al) << 4) | 4u;
				r->has_seq = 1; r->be_side = 0; r->clip_len = L - al; r->be_index = k0; planted = 1;
			}
		}
		if (!planted) {
			// discordant pair across a junction: read entirely on the near side of the up end, mate beyond the down end
			int32_t ku = sy_breakend_at_or_before(be, c->n_breakends, x + 450);
			for (int t = 0; t < 2 && ku >= 0; ++t, --ku) {
				const sy_breakend *b = &be[ku];
				if (b->tid != tid || !b->is_up) continue;
				int64_t d = b->side == 0 ? (b->lin - (x + L)) : (x - b->lin); // gap between the read and the breakpoint
				if (d < 0 || d > 300 - 0) continue;
				if ((int32_t)((h2 >> 12) % 1000u) >= c->vaf_permille) break;
				int32_t rest = T - L - (int32_t)d - L; // what is left of the insert on the far side
				if (rest < 0) rest = 0;
				flag &= ~(2u | 16u | 32u);
				if (b->side == 1) flag |= 16u; // read on the right of the up breakpoint points left
				mtid = b->ptid;
				if (b->mate_rev) { flag |= 32u; mpos = b->mate_anchor + rest; }
				else { mpos = b->mate_anchor - L - rest + 1; if (mpos < 0) mpos = 0; }
				isize = (mtid == tid) ? (mpos - (int32_t)p) : 0;
				planted = 3;
				break;
			}
		}
	}
	if (!planted) {
		const uint32_t u_c = (uint32_t)(h2 % 1000u);
		if ((int32_t)u_c < c->clip_permille) {
			int32_t clip = 5 + (int32_t)((h2 >> 10) % 71u);
			r->n_cigar = 2; r->has_seq = 1; r->clip_len = clip;
			if ((h2 >> 20) & 1u) { r->cigar[0] = ((uint32_t)clip << 4) | 4u; r->cigar[1] = ((uint32_t)(L - clip) << 4) | 0u; r->be_side = -1; }
			else { r->cigar[0] = ((uint32_t)(L - clip) << 4) | 0u; r->cigar[1] = ((uint32_t)clip << 4) | 4u; r->be_side = -2; }
		} else if ((int32_t)u_c < c->clip_permille + c->indel_permille) {
			int32_t k = 1 + (int32_t)((h2 >> 10) % 5u), a = 20 + (int32_t)((h2 >> 16) % (uint32_t)(L - 60));
			r->n_cigar = 3;
			if ((h2 >> 30) & 1u) { r->cigar[0] = ((uint32_t)a << 4) | 0u; r->cigar[1] = ((uint32_t)k << 4) | 1u; r->cigar[2] = ((uint32_t)(L - a - k) << 4) | 0u; }
			else { r->cigar[0] = ((uint32_t)a << 4) | 0u; r->cigar[1] = ((uint32_t)k << 4) | 2u; r->cigar[2] = ((uint32_t)(L - a) << 4) | 0u; }
		}
	}
	if (c->unmap_permille > 0 && (g | 1) < c->n_total && (int32_t)(sy_hash(c->seed ^ 0x554E4D4150ull, (uint64_t)(g >> 1), 9) % 1000u) < c->unmap_permille) {
		// one end of the pair did not align: both records go to the side channel (neither is clipped-read material, whatever was decided above); the
		// unmapped read keeps a plain CIGAR (a record without one makes the reference's hard-clip test read past its CIGAR: IsHardClip, clip_reads.cpp:247)
		r->n_cigar = 1; r->cigar[0] = ((uint32_t)L << 4) | 0u; r->cigar[1] = r->cigar[2] = 0;
		r->has_seq = 0; r->be_side = -1; r->clip_len = 0; r->be_index = -1;
		const uint32_t strand = (uint32_t)(sy_hash(c->seed ^ 0x554E4D4150ull, (uint64_t)(g >> 1), 10) & 1u);
		if (g & 1) { flag = 1u | 4u | (strand ? 32u : 0u) | 128u; mapq = 0; }
		else flag = 1u | 8u | (strand ? 16u : 0u) | 64u | (flag & 1024u);
		mtid = tid; mpos = (int32_t)p; isize = 0;
	}
	r->tid = tid; r->pos = (int32_t)p; r->l_qseq = L; r->mtid = mtid; r->mpos = mpos; r->isize = isize;
	r->flag = (uint16_t)flag; r->mapq = (uint8_t)mapq;
}

// packed bases ((L+1)/2 bytes) and qualities (L bytes) of a soft-clipped record into dst
SY_HD void sy_fill_seq(const sy_config *c, const sy_breakend *be, int64_t g, const sy_record *r, uint8_t *dst)
{
	const int32_t L = r->l_qseq;
	uint8_t *q = dst + (L + 1) / 2;
	int32_t left_clip = 0, right_clip = 0;
	if (r->be_side == 1 || r->be_side == -1) left_clip = r->clip_len; else right_clip = r->clip_len;
	for (int32_t i = 0; i < L; ++i) {
		uint32_t b;
		const uint64_t hb = sy_hash(c->seed ^ 0x4241534553ull, (uint64_t)g, (uint64_t)i);
		if (i < left_clip || i >= L - right_clip) {
			if (r->be_index >= 0) {
				const sy_breakend *e = &be[r->be_index];
				int32_t step = i < left_clip ? (left_clip - 1 - i) : (i - (L - right_clip)); // distance from the junction
				b = sy_ref_base(c, e->ptid, (int64_t)e->ppos + (int64_t)e->pdir * step);
				int natural = e->side == 0 ? 1 : -1;
				if (e->pdir != natural) b = sy_comp(b);
			} else b = 1u << ((hb >> 40) & 3u);
		} else b = sy_ref_base(c, r->tid, (int64_t)r->pos + (i - left_clip));
		if ((uint32_t)(hb % 1000u) < 2u) b = ((b << 1) | (b >> 3)) & 15u; // 0.2 % substitutions
		if (i & 1) dst[i >> 1] = (uint8_t)(dst[i >> 1] | b); else dst[i >> 1] = (uint8_t)(b << 4);
		const uint32_t uq = (uint32_t)((hb >> 12) % 100u);
		if (c->qual_model == 1) { const uint32_t v = (uint32_t)((hb >> 12) & 1023u); q[i] = (uint8_t)(41u - ((((v * v) >> 10) * 40u) >> 10)); } // 41 .. 2, a sixth of them 41
		else q[i] = uq < 2 ? 2 : uq < 7 ? 11 : uq < 20 ? 25 : uq < 60 ? 37 : 40;
	}
}

// the record's 64-byte line (ssv_record, include/seeksv_hip.h) as 16 dwords: tid, pos, flag | mapq << 16 | xc << 24, n_cigar, l_qseq, mtid, mpos,
// isize, cigar_off, the first five CIGAR operations, seq_off (two dwords)
SY_HD void sy_fill_line(const sy_record *r, uint32_t cigar_off, uint64_t seq_off, uint32_t *w)
{
	w[0] = (uint32_t)r->tid; w[1] = (uint32_t)r->pos; w[2] = (uint32_t)r->flag | ((uint32_t)r->mapq << 16); w[3] = (uint32_t)r->n_cigar;
	w[4] = (uint32_t)r->l_qseq; w[5] = (uint32_t)r->mtid; w[6] = (uint32_t)r->mpos; w[7] = (uint32_t)r->isize;
	w[8] = cigar_off;
	for (int k = 0; k < 5; ++k) w[9 + k] = k < r->n_cigar && k < 3 ? r->cigar[k] : 0u;
	w[14] = (uint32_t)seq_off; w[15] = (uint32_t)(seq_off >> 32);
}

// 32 bases of the concatenated reference (linear coordinates 32 w ..) as one 2-bit word, A C G T = 0 1 2 3 (the re-aligner's layout)
SY_HD uint64_t sy_ref_word(const sy_config *c, int64_t w)
{
	const int64_t total = c->contig_off[c->n_contigs];
	uint64_t out = 0;
	int64_t lin = w * 32;
	if (lin >= total) return 0;
	int32_t tid = sy_contig_of(c, lin);
	for (int j = 0; j < 32 && lin < total; ++j, ++lin) {
		while (lin >= c->contig_off[tid + 1]) ++tid;
		const uint32_t b = sy_ref_base(c, tid, lin - c->contig_off[tid]); // nibble code 1 2 4 8
		const uint64_t two = b == 1 ? 0 : b == 2 ? 1 : b == 4 ? 2 : 3;
		out |= two << (2 * j);
	}
	return out;
}
