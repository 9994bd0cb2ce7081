// synth_cpu.cpp - host build of the synthetic generator (tests and the CPU-baseline leg of bench.py).
#include "synth.h"

#include <cstring>
#include <string>

static std::string g_err;

extern "C" {

const char *ssvs_last_error(void) { return g_err.c_str(); }

int ssvs_plan(const sy_config *cfg, const sy_breakend *be, int64_t g0, int64_t n, uint16_t *n_cigar, uint32_t *cigar_off, uint64_t *seq_off,
              int64_t *n_cigar_total, int64_t *seqqual_bytes)
{
	uint64_t co = 0, so = 0;
	for (int64_t i = 0; i < n; ++i) {
		sy_record r;
		sy_decide(cfg, be, g0 + i, &r);
		n_cigar[i] = r.n_cigar;
		cigar_off[i] = (uint32_t)co; co += r.n_cigar;
		if (r.has_seq) { seq_off[i] = so; so += (uint64_t)((r.l_qseq + 1) / 2 + r.l_qseq); }
		else seq_off[i] = UINT64_MAX;
	}
	*n_cigar_total = (int64_t)co; *seqqual_bytes = (int64_t)so;
	return 0;
}

int ssvs_fill(const sy_config *cfg, const sy_breakend *be, int64_t g0, int64_t n, int32_t *tid, int32_t *pos, uint16_t *flag, uint8_t *mapq,
              const uint16_t *n_cigar, int32_t *l_qseq, int32_t *mtid, int32_t *mpos, int32_t *isize, const uint32_t *cigar_off, uint32_t *cigar,
              const uint64_t *seq_off, uint8_t *seqqual, void *rec, uint8_t *cigar_ends)
{
	(void)n_cigar;
	for (int64_t i = 0; i < n; ++i) {
		sy_record r;
		sy_decide(cfg, be, g0 + i, &r);
		tid[i] = r.tid; pos[i] = r.pos;
		if (flag) flag[i] = r.flag;
		if (mapq) mapq[i] = r.mapq;
		if (l_qseq) l_qseq[i] = r.l_qseq;
		if (mtid) mtid[i] = r.mtid;
		if (mpos) mpos[i] = r.mpos;
		if (isize) isize[i] = r.isize;
		for (int k = 0; k < r.n_cigar; ++k) cigar[cigar_off[i] + k] = r.cigar[k];
		if (cigar_ends) cigar_ends[i] = r.n_cigar ? (uint8_t)((r.cigar[0] & 15u) | ((r.cigar[r.n_cigar - 1] & 15u) << 4)) : (uint8_t)0xff;
		if (r.has_seq) sy_fill_seq(cfg, be, g0 + i, &r, seqqual + seq_off[i]);
		if (rec) sy_fill_line(&r, cigar_off[i], seq_off[i], reinterpret_cast<uint32_t *>(rec) + 16 * i);
	}
	return 0;
}

// packed bases + qualities of EVERY record [g0, g0 + n), (read_len + 1) / 2 + read_len bytes each, one after the other: what a BAM file of the
// sample holds (bench.py's file leg).  A soft-clipped record's bytes are the ones ssvs_fill ships for it; the others read the reference
// at their position with the same 0.2 % substitutions and the same quality distribution ({2, 11, 25, 37, 40}).
int ssvs_fill_seq_all(const sy_config *cfg, const sy_breakend *be, int64_t g0, int64_t n, uint8_t *seqqual)
{
	const size_t entry = (size_t)((cfg->read_len + 1) / 2 + cfg->read_len);
	for (int64_t i = 0; i < n; ++i) {
		sy_record r;
		sy_decide(cfg, be, g0 + i, &r);
		sy_fill_seq(cfg, be, g0 + i, &r, seqqual + (size_t)i * entry);
	}
	return 0;
}

// reference bases of a contig as ASCII (FASTA export of the hash-generated genome; golden generation only)
int ssvs_ref_bases(const sy_config *cfg, int32_t tid, int64_t start, int64_t n, char *out)
{
	static const char T[16] = {'N', 'A', 'C', 'N', 'G', 'N', 'N', 'N', 'T', 'N', 'N', 'N', 'N', 'N', 'N', 'N'};
	for (int64_t i = 0; i < n; ++i) out[i] = T[sy_ref_base(cfg, tid, start + i)];
	return 0;
}

// the whole reference as 2-bit words (ssv_realign_index layout); out has (total + 31) / 32 + 1 words, the last one zero
int ssvs_ref_2bit(const sy_config *cfg, uint64_t *out, int64_t n_words)
{
	for (int64_t w = 0; w < n_words; ++w) out[w] = sy_ref_word(cfg, w);
	return 0;
}

} // extern "C"
