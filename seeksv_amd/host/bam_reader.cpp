// bam_reader.cpp - BGZF + BAM decoding into structure-of-arrays batches.
//
// Replaces, for the hot path, what the reference gets from samtools-0.1.16's libbam below its
// record loops: samopen()/samread() (sam/sam.h:59,73) and the bam1_core_t accessors
// (sam/bam.h:169-255).  Own implementation from the SAM/BAM specification; no libbam here.
#include "seeksv_host.h"

#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

struct Bgzf {
	FILE *fp = nullptr;
	std::vector<uint8_t> cbuf, ubuf;
	size_t upos = 0, ulen = 0;
	bool eof = false;

	// inflate the next BGZF block into ubuf; false at EOF or error (g_err set on error)
	bool next_block()
	{
		uint8_t hdr[18];
		size_t got = fread(hdr, 1, 18, fp);
		if (got == 0) { eof = true; return false; }
		if (got < 18 || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) { g_err = "not a BGZF block"; eof = true; return false; }
		unsigned xlen = hdr[10] | (hdr[11] << 8);
		// the BC subfield is normally first; walk the extra field to be safe
		std::vector<uint8_t> extra(xlen);
		memcpy(extra.data(), hdr + 12, xlen < 6 ? xlen : 6);
		if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, fp) != xlen - 6) { g_err = "truncated BGZF header"; eof = true; return false; }
		int bsize = -1;
		for (size_t off = 0; off + 4 <= xlen;) {
			unsigned slen = extra[off + 2] | (extra[off + 3] << 8);
			if (extra[off] == 'B' && extra[off + 1] == 'C' && slen == 2) bsize = extra[off + 4] | (extra[off + 5] << 8);
			off += 4 + slen;
		}
		if (bsize < 0) { g_err = "BGZF block without BC field"; eof = true; return false; }
		size_t clen = (size_t)bsize + 1 - 12 - xlen; // deflate data + crc32 + isize
		if (clen < 8) { g_err = "bad BGZF block size"; eof = true; return false; }
		cbuf.resize(clen);
		if (fread(cbuf.data(), 1, clen, fp) != clen) { g_err = "truncated BGZF block"; eof = true; return false; }
		uint32_t isize;
		memcpy(&isize, cbuf.data() + clen - 4, 4);
		ubuf.resize(isize ? isize : 1);
		z_stream zs;
		memset(&zs, 0, sizeof(zs));
		if (inflateInit2(&zs, -15) != Z_OK) { g_err = "inflateInit2 failed"; eof = true; return false; }
		zs.next_in = cbuf.data(); zs.avail_in = (uInt)(clen - 8);
		zs.next_out = ubuf.data(); zs.avail_out = (uInt)ubuf.size();
		int rc = inflate(&zs, Z_FINISH);
		inflateEnd(&zs);
		if (rc != Z_STREAM_END || zs.total_out != isize) { g_err = "BGZF inflate failed"; eof = true; return false; }
		upos = 0; ulen = isize;
		return true;
	}

	// read exactly n bytes of the uncompressed stream; returns bytes read (< n only at EOF)
	size_t read(void *dst, size_t n)
	{
		uint8_t *d = (uint8_t *)dst;
		size_t done = 0;
		while (done < n) {
			if (upos == ulen) {
				if (eof) break;
				if (!next_block()) { if (eof) break; else continue; }
				continue;
			}
			size_t take = ulen - upos < n - done ? ulen - upos : n - done;
			memcpy(d + done, ubuf.data() + upos, take);
			upos += take; done += take;
		}
		return done;
	}
};

struct Unmapped {
	std::string qname, seq, qual;
	int is_read1;
};

} // namespace

struct ssvh_bam {
	Bgzf z;
	std::vector<std::string> names;
	std::vector<int32_t> lens;
	// the current batch
	std::vector<int32_t> tid, pos, l_qseq, mtid, mpos, isize;
	std::vector<uint16_t> flag, n_cigar;
	std::vector<uint8_t> mapq, xc, seqqual;
	std::vector<uint32_t> cigar_off, cigar;
	std::vector<uint64_t> seq_off;
	std::vector<Unmapped> unmapped;
	std::vector<uint8_t> rec;
};

static const char NT16[] = "=ACMGRSVTWYHKDBN";

// bam_aux_get(b, "XC") + bam_aux2i (clip_reads.cpp:126-127): integer value of the XC tag, 0 when absent
// or not an integer type.
static int aux_xc(const uint8_t *p, const uint8_t *end)
{
	while (p + 3 <= end) {
		const uint8_t t0 = p[0], t1 = p[1], ty = p[2];
		p += 3;
		int64_t val = 0; bool isint = false;
		size_t sz = 0;
		switch (ty) {
		case 'A': sz = 1; break;
		case 'c': sz = 1; if (p + 1 <= end) { val = (int8_t)p[0]; isint = true; } break;
		case 'C': sz = 1; if (p + 1 <= end) { val = p[0]; isint = true; } break;
		case 's': sz = 2; if (p + 2 <= end) { int16_t v; memcpy(&v, p, 2); val = v; isint = true; } break;
		case 'S': sz = 2; if (p + 2 <= end) { uint16_t v; memcpy(&v, p, 2); val = v; isint = true; } break;
		case 'i': sz = 4; if (p + 4 <= end) { int32_t v; memcpy(&v, p, 4); val = v; isint = true; } break;
		case 'I': sz = 4; if (p + 4 <= end) { uint32_t v; memcpy(&v, p, 4); val = v; isint = true; } break;
		case 'f': sz = 4; break;
		case 'd': sz = 8; break;
		case 'Z': case 'H': { const uint8_t *q = p; while (q < end && *q) ++q; sz = (size_t)(q - p) + 1; break; }
		case 'B': {
			if (p + 5 > end) return 0;
			uint8_t sub = p[0]; uint32_t cnt; memcpy(&cnt, p + 1, 4);
			size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
			sz = 5 + es * cnt; break;
		}
		default: return 0; // unknown type: stop like a failed skip
		}
		if (t0 == 'X' && t1 == 'C') return isint ? (int)val : 0;
		p += sz;
	}
	return 0;
}

extern "C" {

const char *ssvh_last_error(void) { return g_err.c_str(); }

int ssvh_bam_open(const char *path, ssvh_bam **out)
{
	*out = nullptr;
	ssvh_bam *b = new ssvh_bam();
	b->z.fp = fopen(path, "rb");
	if (!b->z.fp) { g_err = std::string("cannot open ") + path; delete b; return -1; }
	char magic[4];
	int32_t l_text, n_ref;
	if (b->z.read(magic, 4) != 4 || memcmp(magic, "BAM\1", 4) != 0) { if (g_err.empty()) g_err = "not a BAM file"; fclose(b->z.fp); delete b; return -1; }
	if (b->z.read(&l_text, 4) != 4 || l_text < 0) { g_err = "bad BAM header"; fclose(b->z.fp); delete b; return -1; }
	std::vector<char> text((size_t)l_text);
	if (b->z.read(text.data(), (size_t)l_text) != (size_t)l_text || b->z.read(&n_ref, 4) != 4 || n_ref < 0) { g_err = "bad BAM header"; fclose(b->z.fp); delete b; return -1; }
	for (int32_t i = 0; i < n_ref; ++i) {
		int32_t l_name, l_ref;
		if (b->z.read(&l_name, 4) != 4 || l_name <= 0) { g_err = "bad BAM header"; fclose(b->z.fp); delete b; return -1; }
		std::string nm((size_t)l_name, '\0');
		if (b->z.read(&nm[0], (size_t)l_name) != (size_t)l_name || b->z.read(&l_ref, 4) != 4) { g_err = "bad BAM header"; fclose(b->z.fp); delete b; return -1; }
		nm.resize(strlen(nm.c_str()));
		b->names.push_back(nm);
		b->lens.push_back(l_ref);
	}
	*out = b;
	return 0;
}

int ssvh_bam_from_header(const char *const *names, const int32_t *lens, int32_t n, ssvh_bam **out)
{
	ssvh_bam *b = new ssvh_bam();
	for (int32_t i = 0; i < n; ++i) { b->names.push_back(names[i]); b->lens.push_back(lens[i]); }
	b->z.eof = true;
	*out = b;
	return 0;
}

void ssvh_bam_close(ssvh_bam *b)
{
	if (!b) return;
	if (b->z.fp) fclose(b->z.fp);
	delete b;
}

int32_t ssvh_bam_n_targets(const ssvh_bam *b) { return (int32_t)b->names.size(); }
const char *ssvh_bam_target_name(const ssvh_bam *b, int32_t tid) { return (tid >= 0 && (size_t)tid < b->names.size()) ? b->names[(size_t)tid].c_str() : nullptr; }
int32_t ssvh_bam_target_len(const ssvh_bam *b, int32_t tid) { return (tid >= 0 && (size_t)tid < b->lens.size()) ? b->lens[(size_t)tid] : -1; }
const int32_t *ssvh_bam_target_lens(const ssvh_bam *b) { return b->lens.data(); }

int ssvh_bam_read_batch(ssvh_bam *b, int64_t max_records, int keep_all_seq, ssv_batch_t *out)
{
	b->tid.clear(); b->pos.clear(); b->l_qseq.clear(); b->mtid.clear(); b->mpos.clear(); b->isize.clear();
	b->flag.clear(); b->n_cigar.clear(); b->mapq.clear(); b->xc.clear(); b->seqqual.clear();
	b->cigar_off.clear(); b->cigar.clear(); b->seq_off.clear(); b->unmapped.clear();
	g_err.clear();
	int64_t n = 0, max_span = 1;
	while (n < max_records) {
		int32_t block_size;
		size_t got = b->z.read(&block_size, 4);
		if (got == 0) break;
		if (got != 4 || block_size < 32) { if (g_err.empty()) g_err = "truncated BAM record"; return -1; }
		b->rec.resize((size_t)block_size);
		if (b->z.read(b->rec.data(), (size_t)block_size) != (size_t)block_size) { if (g_err.empty()) g_err = "truncated BAM record"; return -1; }
		const uint8_t *r = b->rec.data();
		int32_t refid, pos, l_seq, next_ref, next_pos, tlen;
		uint16_t ncig, flag;
		memcpy(&refid, r, 4); memcpy(&pos, r + 4, 4);
		uint8_t l_read_name = r[8], mapq = r[9];
		memcpy(&ncig, r + 12, 2); memcpy(&flag, r + 14, 2); memcpy(&l_seq, r + 16, 4);
		memcpy(&next_ref, r + 20, 4); memcpy(&next_pos, r + 24, 4); memcpy(&tlen, r + 28, 4);
		size_t o_name = 32, o_cig = o_name + l_read_name, o_seq = o_cig + 4 * (size_t)ncig;
		size_t o_qual = o_seq + ((size_t)l_seq + 1) / 2, o_aux = o_qual + (size_t)l_seq;
		if (l_seq < 0 || o_aux > (size_t)block_size) { g_err = "corrupt BAM record"; return -1; }
		b->tid.push_back(refid); b->pos.push_back(pos); b->flag.push_back(flag); b->mapq.push_back(mapq);
		b->n_cigar.push_back(ncig); b->l_qseq.push_back(l_seq); b->mtid.push_back(next_ref); b->mpos.push_back(next_pos); b->isize.push_back(tlen);
		b->cigar_off.push_back((uint32_t)b->cigar.size());
		bool soft = false;
		int64_t span = 0;
		for (unsigned k = 0; k < ncig; ++k) {
			uint32_t c; memcpy(&c, r + o_cig + 4 * k, 4);
			b->cigar.push_back(c);
			if ((k == 0 || k + 1 == ncig) && (c & 15) == 4) soft = true;
			unsigned op = c & 15;
			if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += c >> 4;
		}
		if (span > max_span) max_span = span;
		b->xc.push_back(soft ? (uint8_t)(aux_xc(r + o_aux, r + block_size) != 0) : (uint8_t)0);
		if (soft || keep_all_seq) {
			b->seq_off.push_back((uint64_t)b->seqqual.size());
			b->seqqual.insert(b->seqqual.end(), r + o_seq, r + o_aux);
		} else b->seq_off.push_back(SSV_NO_SEQ);
		if (flag & (4 | 8)) {
			// GetSeqAndQual (clip_reads.cpp:375-388): bases as stored, qualities +33, "*" when absent
			Unmapped u;
			u.qname.assign((const char *)r + o_name);
			u.seq.resize((size_t)l_seq);
			for (int32_t k = 0; k < l_seq; ++k) u.seq[(size_t)k] = NT16[(r[o_seq + (k >> 1)] >> ((~k & 1) << 2)) & 15];
			if (l_seq > 0 && r[o_qual] == 0xff) u.qual = "*";
			else { u.qual.resize((size_t)l_seq); for (int32_t k = 0; k < l_seq; ++k) u.qual[(size_t)k] = (char)(r[o_qual + k] + 33); }
			u.is_read1 = (flag & 64) ? 1 : 0;
			b->unmapped.push_back(std::move(u));
		}
		++n;
	}
	if (!g_err.empty()) return -1;
	memset(out, 0, sizeof(*out));
	out->n = n; out->mem = SSV_MEM_HOST;
	out->max_ref_span = (int32_t)(max_span > INT32_MAX ? INT32_MAX : max_span);
	out->tid = b->tid.data(); out->pos = b->pos.data(); out->flag = b->flag.data(); out->mapq = b->mapq.data();
	out->n_cigar = b->n_cigar.data(); out->l_qseq = b->l_qseq.data(); out->mtid = b->mtid.data(); out->mpos = b->mpos.data();
	out->isize = b->isize.data(); out->cigar_off = b->cigar_off.data(); out->cigar = b->cigar.data(); out->xc = b->xc.data();
	out->seq_off = b->seq_off.data(); out->seqqual = b->seqqual.data();
	out->n_cigar_total = (int64_t)b->cigar.size(); out->seqqual_bytes = (int64_t)b->seqqual.size();
	return 0;
}

int ssvh_bam_next_record(ssvh_bam *b, ssvh_record *out)
{
	g_err.clear();
	int32_t block_size;
	size_t got = b->z.read(&block_size, 4);
	if (got == 0) return g_err.empty() ? 0 : -1;
	if (got != 4 || block_size < 32) { if (g_err.empty()) g_err = "truncated BAM record"; return -1; }
	b->rec.resize((size_t)block_size + 4);
	if (b->z.read(b->rec.data(), (size_t)block_size) != (size_t)block_size) { if (g_err.empty()) g_err = "truncated BAM record"; return -1; }
	const uint8_t *r = b->rec.data();
	memcpy(&out->tid, r, 4); memcpy(&out->pos, r + 4, 4);
	const uint8_t l_read_name = r[8];
	out->mapq = r[9];
	memcpy(&out->n_cigar, r + 12, 2); memcpy(&out->flag, r + 14, 2); memcpy(&out->l_qseq, r + 16, 4);
	const size_t o_cig = 32 + (size_t)l_read_name, o_seq = o_cig + 4 * (size_t)out->n_cigar, o_qual = o_seq + ((size_t)out->l_qseq + 1) / 2;
	if (out->l_qseq < 0 || o_qual + (size_t)out->l_qseq > (size_t)block_size) { g_err = "corrupt BAM record"; return -1; }
	out->qname = (const char *)r + 32;
	b->cigar.assign((size_t)out->n_cigar, 0); // aligned copy
	if (out->n_cigar) memcpy(b->cigar.data(), r + o_cig, 4 * (size_t)out->n_cigar);
	out->cigar = b->cigar.data();
	out->seq = r + o_seq; out->qual = r + o_qual;
	return 1;
}

int64_t ssvh_bam_unmapped_count(const ssvh_bam *b) { return (int64_t)b->unmapped.size(); }

int ssvh_bam_unmapped_get(const ssvh_bam *b, int64_t k, const char **qname, const char **seq, const char **qual, int *is_read1)
{
	if (k < 0 || (size_t)k >= b->unmapped.size()) return -1;
	const Unmapped &u = b->unmapped[(size_t)k];
	*qname = u.qname.c_str(); *seq = u.seq.c_str(); *qual = u.qual.c_str(); *is_read1 = u.is_read1;
	return 0;
}

} // extern "C"
