// junction_stage.cpp - see junction_stage.h.  Restated from the behaviour of the reference, quirks included (they change the table):
//   * clip rows are grouped while consecutive rows carry the same clipped sequence; clip.bam records are consumed in file order
//     and belong to a group while their read name equals that sequence (the re-aligner must keep FASTQ order and use the sequence
//     as the read name, as `bwa mem` does with getclip's FASTQ);
//   * the first record that ends a group is filed under the PREVIOUS group's name (getsv.h:502), which decides its place in the
//     ordered alignment map and therefore the order in which junctions are created;
//   * only the first clip row of a group is paired with the group's alignments (the inner iterator is never reset, getsv.h:489-498);
//   * hard-clipped records are skipped in the main loop but not in the tail loop (getsv.h:476 vs :512-527).
#include "junction_stage.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <iostream>
#include <memory>
#include <climits>
#include <deque>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <thread>

#include "../csrc/cpus.h"
#include "../csrc/thp.h"

#include "seeksv_host.h"

namespace seeksv {

namespace {

// a piece of text that lives elsewhere (the rows of clip.gz in memory; the stream loop's strings in a store of their own)
// 64-bit hash of a piece of text.  The join compares every clipped sequence with the name of a re-alignment record - 5.5 M times on a whole-genome
// sample, nearly always unequal: two hashes (computed where the text is parsed anyway, by all threads) settle that without touching 2.5 GB of text
// on the one thread the join runs on; equal hashes are confirmed by comparing the text.
static inline uint64_t text_hash(const char *p, size_t n)
{ // (also clip_text_hash below)
	uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
	size_t i = 0;
	for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, p + i, 8); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 29; }
	uint64_t w = 0;
	if (i < n) { memcpy(&w, p + i, n - i); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 29; }
	return h * 0xC4CEB9FE1A85EC53ull;
}

struct AlignInfo { // getsv.h:27-44
	std::string chr; int pos = 0, len = 0; char strand = 0; CigarVec cigar_vec; std::string seq; int left_clipped = 0, right_clipped = 0; char type = 0;
};

typedef std::map<std::pair<std::string, std::pair<std::string, int>>, AlignInfo> AlignMap; // (name, (chr, pos)) -> alignment

void reverse_cigar(CigarVec &v) // ReverseCigar, getsv.cpp:453
{
	for (size_t i = 0, n = v.size(); i < n / 2; ++i) std::swap(v[i], v[n - 1 - i]);
}

// where the join's alignment records come from: clip.bam through the host reader, or the aligner step's results still in memory
struct RecSource {
	ssvh_bam *bam = nullptr;
	const AlnRecords *mem = nullptr;
	int64_t at = 0;
	std::vector<uint64_t> qhash; // mem: text_hash of every read name (made by all threads before the join starts)
	uint64_t h = 0;              // ... of the record next() handed out last
	void hash_names()
	{
		if (!mem || mem->n == 0) return;
		qhash.resize((size_t)mem->n);
		const int nt = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)ssv::effective_cpus(), 64, mem->n / 65536 + 1}));
		std::vector<std::thread> th;
		for (int w = 0; w < nt; ++w) th.emplace_back([this, w, nt] {
			for (int64_t i = mem->n * w / nt, e = mem->n * (w + 1) / nt; i < e; ++i) qhash[(size_t)i] = text_hash(mem->qname[i], strlen(mem->qname[i]));
		});
		for (auto &t : th) t.join();
	}
	int next(ssvh_record *out) // 1: a record, 0: the end, < 0: error (ssvh_last_error)
	{
		if (bam) { const int rc = ssvh_bam_next_record(bam, out); if (rc == 1) h = text_hash(out->qname, strlen(out->qname)); return rc; }
		if (at >= mem->n) return 0;
		const int64_t i = at++;
		h = qhash[(size_t)i];
		out->tid = mem->tid[i]; out->pos = mem->pos[i]; out->flag = mem->flag[i]; out->mapq = mem->mapq[i]; out->n_cigar = mem->n_cigar[i];
		out->cigar = mem->cigar + mem->cigar_off[i]; out->qname = mem->qname[i];
		out->l_qseq = 0; out->seq = nullptr; out->qual = nullptr;
		return 1;
	}
	const char *target_name(int32_t tid) const
	{
		if (bam) return ssvh_bam_target_name(bam, tid);
		return tid >= 0 && (size_t)tid < mem->target_names.size() ? mem->target_names[(size_t)tid].c_str() : nullptr;
	}
};

// GetAlignInfo, getsv.cpp:25-71
void align_info_of(const RecSource *bam, const ssvh_record &r, AlignInfo &a)
{
	if (r.flag & 4) {
		a = AlignInfo();
		a.chr = "Exogenous"; a.pos = -1; a.len = -1; a.strand = '*'; a.type = 'n';
		return;
	}
	a.type = ((r.flag & 256) || r.mapq == 0) ? 'r' : 'u';
	a.left_clipped = a.right_clipped = 0;
	if (r.n_cigar) {
		unsigned op1 = r.cigar[0] & 15u, op2 = r.cigar[r.n_cigar - 1] & 15u;
		if (op1 == 4 || op1 == 5) a.left_clipped = (int)(r.cigar[0] >> 4);
		if (op2 == 4 || op2 == 5) a.right_clipped = (int)(r.cigar[r.n_cigar - 1] >> 4);
	}
	a.seq = r.qname;
	a.cigar_vec.clear();
	a.len = 0;
	for (unsigned k = 0; k < r.n_cigar; ++k) { // GenerateCigar, clip_reads.cpp:309-329
		unsigned op = r.cigar[k] & 15u;
		int l = (int)(r.cigar[k] >> 4);
		if (op == 4 || op == 5) continue;
		if (op == 0 || op == 2 || op == 7 || op == 3) a.len += l;
		a.cigar_vec.push_back(std::make_pair(l, "MIDNSHP=X"[op]));
	}
	if (r.flag & 16) { a.strand = '-'; reverse_complement(a.seq); } else a.strand = '+';
	const char *nm = bam->target_name(r.tid);
	a.chr = nm ? nm : "";
	a.pos = r.pos + 1;
}

bool is_hard_clip(const ssvh_record &r) // IsHardClip, clip_reads.cpp:247 (no CIGAR: the reference reads neighbouring bytes, which never look like 'H')
{
	if (!r.n_cigar) return false;
	return (r.cigar[0] & 15u) == 5 || (r.cigar[r.n_cigar - 1] & 15u) == 5;
}

SeqInfo make_seq_info(const std::string &seq, const CigarVec &cig, int lc, int rc, int support, int uniq)
{
	SeqInfo s; s.seq = seq; s.cigar_vec = cig; s.left_clipped = lc; s.right_clipped = rc; s.support = support; s.uniq = uniq;
	return s;
}

// GetJunction, getsv.cpp:1705-1845
void add_junction(const ClipRow &row, AlignInfo &c, JunctionMap &j2o)
{
	int uniq;
	if (c.type == 'u') uniq = 2; else if (c.type == 'r') uniq = 1; else return; // type 'n': nothing is recorded at all
	std::string aligned_seq = row.aligned_seq.str(), clipped_seq = row.clipped_seq.str();
	CigarVec cigar_vec = parse_cigar(row.cigar.str());
	const std::string chr = row.chr.str();
	const int pos = row.pos, support = row.support;
	Junction j;
	SeqInfo up, down;
	if (c.strand == '+') {
		if (row.side == '5') {
			j = Junction{c.chr, c.pos + c.len - 1, '+', chr, pos, '+'};
			up = make_seq_info(clipped_seq, c.cigar_vec, c.left_clipped, c.right_clipped, 0, uniq);
			down = make_seq_info(aligned_seq, cigar_vec, 0, 0, support, 0);
		} else if (row.side == '3') {
			j = Junction{chr, pos, '+', c.chr, c.pos, '+'};
			up = make_seq_info(aligned_seq, cigar_vec, 0, 0, support, 0);
			down = make_seq_info(clipped_seq, c.cigar_vec, c.left_clipped, c.right_clipped, 0, uniq);
		}
	} else if (c.strand == '-') {
		if (row.side == '5') {
			if (std::make_pair(c.chr, c.pos) <= std::make_pair(chr, pos)) {
				j = Junction{c.chr, c.pos, '-', chr, pos, '+'};
				up = make_seq_info(clipped_seq, c.cigar_vec, c.left_clipped, c.right_clipped, 0, uniq);
				down = make_seq_info(aligned_seq, cigar_vec, 0, 0, support, 0);
			} else {
				j = Junction{chr, pos, '-', c.chr, c.pos, '+'};
				reverse_complement(aligned_seq); reverse_complement(clipped_seq);
				reverse_cigar(cigar_vec); reverse_cigar(c.cigar_vec);
				up = make_seq_info(aligned_seq, cigar_vec, 0, 0, support, 0);
				down = make_seq_info(clipped_seq, c.cigar_vec, c.right_clipped, c.left_clipped, 0, uniq);
			}
		} else if (row.side == '3') {
			if (std::make_pair(chr, pos) <= std::make_pair(c.chr, c.pos + c.len - 1)) {
				j = Junction{chr, pos, '+', c.chr, c.pos + c.len - 1, '-'};
				up = make_seq_info(aligned_seq, cigar_vec, 0, 0, support, 0);
				down = make_seq_info(clipped_seq, c.cigar_vec, c.left_clipped, c.right_clipped, 0, uniq);
			} else {
				j = Junction{c.chr, c.pos + c.len - 1, '+', chr, pos, '-'};
				reverse_complement(aligned_seq); reverse_complement(clipped_seq);
				reverse_cigar(c.cigar_vec); reverse_cigar(cigar_vec);
				up = make_seq_info(clipped_seq, c.cigar_vec, c.right_clipped, c.left_clipped, 0, uniq);
				down = make_seq_info(aligned_seq, cigar_vec, 0, 0, support, 0);
			}
		}
	}
	// entries with the same key merge when their clipped-length signatures agree (getsv.cpp:1806-1838)
	std::pair<JunctionMap::iterator, JunctionMap::iterator> range = j2o.equal_range(j);
	bool merged = false;
	for (JunctionMap::iterator it = range.first; it != range.second; ++it) {
		OtherInfo &o = it->second;
		if (o.up.right_clipped == down.left_clipped && o.down.left_clipped == up.right_clipped) {
			if (up.uniq > o.up.uniq) o.up.uniq = up.uniq;
			if (down.uniq > o.down.uniq) o.down.uniq = down.uniq;
			o.up.support += up.support;
			o.down.support += down.support;
			if (o.microhomology == -1) o.microhomology = it->first.up_pos - j.up_pos;
			merged = true;
		}
	}
	if (!merged) {
		OtherInfo o; o.up = up; o.down = down; o.microhomology = -1; o.abnormal = 0;
		j2o.insert(std::make_pair(j, o));
	}
}

// pair the group's rows with its alignments the way the reference's nested loops do: the alignment iterator is shared by all rows
void flush_group(const std::vector<ClipRow> &rows, AlignMap &aligns, JunctionMap &j2o)
{
	AlignMap::iterator a = aligns.begin();
	for (size_t r = 0; r < rows.size(); ++r)
		for (; a != aligns.end(); ++a) add_junction(rows[r], a->second, j2o);
}

} // namespace

uint64_t clip_text_hash(const char *p, size_t n) { return text_hash(p, n); }

double match_end_first(const std::string &a, const std::string &b) // CompareStringEndFirst, clip_reads.cpp:194
{
	int la = (int)a.size(), lb = (int)b.size(), n = la < lb ? la : lb, m = 0;
	for (int i = 0; i < n; ++i) if (a[(size_t)(la - 1 - i)] == b[(size_t)(lb - 1 - i)]) ++m;
	return (double)m / n;
}

double match_begin_first(const std::string &a, const std::string &b) // CompareStringBeginFirst, clip_reads.cpp:207
{
	int n = (int)(a.size() < b.size() ? a.size() : b.size()), m = 0;
	for (int i = 0; i < n; ++i) if (a[(size_t)i] == b[(size_t)i]) ++m;
	return (double)m / n;
}

// A .gz file written by ssvh_gz_append is a sequence of independent gzip members that all start with the same ten header bytes: the members
// are found by that pattern, their inflated sizes read from their trailers (ISIZE), and they are inflated side by side straight into place.
// Every member must end exactly where the next one starts, with its CRC and size right (zlib checks both) - otherwise, and for any other
// gzip or plain file, false: the caller reads the file the ordinary way (gzread).
// (the text lands in `buf`, allocated here without being cleared: its pages are first touched by the inflating threads, not by one thread's memset)
bool slurp_gz_members(const std::string &path, std::unique_ptr<char[]> &buf, size_t &buf_len)
{
	static const unsigned char head[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
	const int fd = open(path.c_str(), O_RDONLY);
	if (fd < 0) return false;
	struct stat st;
	if (fstat(fd, &st) != 0 || st.st_size < 36) { close(fd); return false; }
	const size_t size = (size_t)st.st_size;
	const unsigned char *raw = static_cast<const unsigned char *>(mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0));
	close(fd);
	if (raw == MAP_FAILED) return false;
	bool ok = memcmp(raw, head, 10) == 0;
	std::vector<size_t> start;
	if (ok) {
		const int nt = std::max(1, std::min(ssv::effective_cpus(), 64));
		std::vector<std::vector<size_t>> found((size_t)nt);
		std::vector<std::thread> th;
		for (int w = 0; w < nt; ++w) th.emplace_back([&, w] {
			const size_t lo = size * (size_t)w / (size_t)nt, hi = size * (size_t)(w + 1) / (size_t)nt;
			for (const unsigned char *p = raw + lo; p < raw + hi && (p = static_cast<const unsigned char *>(memchr(p, 0x1f, (size_t)(raw + hi - p)))) != nullptr; ++p)
				if ((size_t)(raw + size - p) >= 18 && memcmp(p, head, 10) == 0) found[(size_t)w].push_back((size_t)(p - raw));
		});
		for (auto &t : th) t.join();
		for (auto &f : found) start.insert(start.end(), f.begin(), f.end());
		if (start.size() < 2) ok = false; // one member: nothing to gain
	}
	if (ok) {
		const size_t nm = start.size();
		start.push_back(size);
		std::vector<size_t> at(nm + 1, 0);
		for (size_t k = 0; k < nm; ++k) {
			if (start[k + 1] - start[k] < 20) { ok = false; break; }
			uint32_t isize; memcpy(&isize, raw + start[k + 1] - 4, 4);
			at[k + 1] = at[k] + isize;
		}
		if (ok) {
			buf.reset(new char[at[nm] + 1]);
			ssv::thp_advise(buf.get(), at[nm] + 1); // (GBs of inflated rows, first touched by the inflating threads: thp.h)
			buf_len = at[nm];
			char *const out = buf.get();
			std::atomic<size_t> next{0};
			std::atomic<bool> good{true};
			const int nt = std::max(1, std::min<int>(ssv::effective_cpus(), 64));
			std::vector<std::thread> th;
			for (int w = 0; w < nt; ++w) th.emplace_back([&] {
				z_stream zs;
				memset(&zs, 0, sizeof(zs));
				if (inflateInit2(&zs, 15 + 16) != Z_OK) { good = false; return; }
				for (size_t k; good && (k = next++) < nm;) {
					inflateReset(&zs);
					zs.next_in = const_cast<Bytef *>(raw + start[k]); zs.avail_in = (uInt)(start[k + 1] - start[k]);
					unsigned char dummy;
					zs.next_out = at[k + 1] > at[k] ? reinterpret_cast<Bytef *>(out + at[k]) : &dummy; zs.avail_out = (uInt)(at[k + 1] - at[k]);
					const int rc = inflate(&zs, Z_FINISH);
					if (rc != Z_STREAM_END || zs.avail_in != 0 || zs.avail_out != 0) good = false; // (a false header match inside a member's data ends up here)
				}
				inflateEnd(&zs);
			});
			for (auto &t : th) t.join();
			ok = good;
			if (!ok) { buf.reset(); buf_len = 0; }
		}
	}
	munmap(const_cast<unsigned char *>(raw), size);
	return ok;
}

std::string slurp_gz(const std::string &path, std::string &out)
{
	static const bool serial = getenv("SSV_SERIAL") && strstr(getenv("SSV_SERIAL"), "gz"); // SSV_SERIAL=gz (tests): one inflate stream
	std::unique_ptr<char[]> whole;
	size_t n_whole = 0;
	if (!serial && slurp_gz_members(path, whole, n_whole)) { out.append(whole.get(), n_whole); return ""; }
	gzFile f = gzopen(path.c_str(), "rb"); // reads plain files too
	if (!f) return "Cannot open file " + path;
	char buf[1 << 16];
	int n;
	while ((n = gzread(f, buf, sizeof(buf))) > 0) out.append(buf, (size_t)n);
	gzclose(f);
	return "";
}


CigarVec parse_cigar(const std::string &cigar)
{
	CigarVec v;
	int len = 0;
	for (size_t i = 0; i < cigar.size(); ++i) {
		char ch = cigar[i];
		if (ch >= '0' && ch <= '9') len = len * 10 + (ch - '0');
		else { v.push_back(std::make_pair(len, ch)); len = 0; }
	}
	return v;
}

void reverse_complement(std::string &seq)
{
	auto comp = [](char c) -> char {
		switch (c) {
		case 'A': case 'a': return 'T';
		case 'T': case 't': return 'A';
		case 'C': case 'c': return 'G';
		case 'G': case 'g': return 'C';
		case 'N': case 'n': return 'N';
		default: return c;
		}
	};
	const size_t n = seq.size();
	for (size_t i = 0; i < n / 2; ++i) { char a = comp(seq[i]), b = comp(seq[n - 1 - i]); seq[i] = b; seq[n - 1 - i] = a; }
	if (n % 2) seq[n / 2] = comp(seq[n / 2]);
}

// the same with the rows of clip.gz already in memory (`seeksv run`: getclip has just written them)
// The rows of clip.gz, parsed by several threads: a row is a line of at least nine whitespace-separated fields whose second and ninth are plain
// integers and whose third is one character - what getclip writes.  For such text this is what the reference's `fin >> chr >> pos >> ...;
// getline(fin, rest)` loop (getsv.h:441-446) extracts; anything else (a short line would make operator>> run on into the next one, a field like
// "12x" would split) returns false and the caller parses the text with that very stream loop.

// texts: pieces of whole rows, in file order
bool parse_rows_parallel(const std::vector<TextView> &texts, std::vector<ClipRow> &rows)
{
	static const bool off = getenv("SSV_SERIAL") && strstr(getenv("SSV_SERIAL"), "rows"); // SSV_SERIAL=rows (tests): the stream loop
	if (off) return false;
	static const size_t per_thread = [] { const char *e = getenv("SSV_ROWS_CHUNK_KB"); const long kb = e ? atol(e) : 64; return (size_t)(kb < 1 ? 1 : kb) << 10; }(); // (tests: several threads on little text)
	size_t total_bytes = 0;
	for (const TextView &t : texts) total_bytes += t.size();
	const int nt = (int)std::max<size_t>(1, std::min<size_t>({(size_t)ssv::effective_cpus(), (size_t)64, total_bytes / per_thread + 1}));
	// segments of whole lines, a few per thread, handed out in turn; their rows are put together in segment order
	const size_t seg_bytes = std::max(per_thread, total_bytes / ((size_t)nt * 4) + 1);
	struct Seg { const char *p, *end; };
	std::vector<Seg> segs;
	for (const TextView &text : texts) {
		size_t at = 0;
		while (at < text.size()) {
			size_t stop = at + seg_bytes;
			if (stop >= text.size()) stop = text.size();
			else { const size_t nl = text.find('\n', stop); stop = nl == std::string::npos ? text.size() : nl + 1; }
			segs.push_back(Seg{text.data() + at, text.data() + stop});
			at = stop;
		}
	}
	std::vector<std::vector<ClipRow>> part(segs.size());
	std::atomic<size_t> next_seg{0};
	std::atomic<bool> good{true};
	auto is_space = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; };
	auto to_int = [](const char *p, size_t n, int &v) -> bool { // what operator>>(int&) takes from a token that is nothing but [+-]digits
		size_t i = 0;
		bool neg = false;
		if (n && (p[0] == '+' || p[0] == '-')) { neg = p[0] == '-'; i = 1; }
		if (i == n || n - i > 10) return false;
		long long x = 0;
		for (; i < n; ++i) { if (p[i] < '0' || p[i] > '9') return false; x = x * 10 + (p[i] - '0'); }
		if (neg) x = -x;
		if (x < INT_MIN || x > INT_MAX) return false;
		v = (int)x;
		return true;
	};
	std::vector<std::thread> th;
	for (int w = 0; w < nt; ++w) th.emplace_back([&] {
	  for (size_t sg; good && (sg = next_seg.fetch_add(1)) < segs.size();) {
		std::vector<ClipRow> &out = part[sg];
		const char *p = segs[sg].p, *end = segs[sg].end;
		out.reserve((size_t)(end - p) / 160 + 16); // (a row of getclip's is 300-500 bytes: no growing - a growing vector of this size is mapped, copied and unmapped again and again, under the address-space lock all threads share)
		while (p < end && good) {
			const char *eol = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
			if (!eol) eol = end;
			const char *tok[9]; size_t len[9];
			int nf = 0;
			for (const char *q = p; q < eol && nf < 9;) {
				while (q < eol && is_space(*q)) ++q;
				if (q == eol) break;
				const char *b = q;
				while (q < eol && !is_space(*q)) ++q;
				tok[nf] = b; len[nf] = (size_t)(q - b); ++nf;
			}
			if (nf == 9) {
				ClipRow row;
				if (len[2] != 1 || !to_int(tok[1], len[1], row.pos) || !to_int(tok[8], len[8], row.support)) { good = false; break; }
				row.chr = Str{tok[0], len[0]}; row.side = tok[2][0];
				row.cigar = Str{tok[3], len[3]};
				row.aligned_seq = Str{tok[4], len[4]}; row.clipped_seq = Str{tok[6], len[6]}; row.clipped_qual = Str{tok[7], len[7]};
				row.h = text_hash(tok[6], len[6]);
				out.push_back(row);
			} else if (nf != 0) { good = false; break; }
			p = eol < end ? eol + 1 : end;
		}
	  }
	});
	for (auto &t : th) t.join();
	if (!good) return false;
	size_t total = 0;
	for (auto &v : part) total += v.size();
	rows.reserve(total);
	ssv::thp_advise(rows.data(), rows.capacity() * sizeof(ClipRow));
	for (auto &v : part) { rows.insert(rows.end(), v.begin(), v.end()); std::vector<ClipRow>().swap(v); }
	return true;
}

typedef std::vector<const std::vector<ClipRow> *> RowParts;
static std::string assemble_junctions_view(const std::vector<TextView> &texts, const std::string &clip_bam, const AlnRecords *mem, JunctionMap &j2o, const RowParts *pre = nullptr);
std::string assemble_junctions_text(const std::vector<TextView> &text, const std::string &clip_bam, JunctionMap &j2o) { return assemble_junctions_view(text, clip_bam, nullptr, j2o); }
std::string assemble_junctions_records(const std::vector<TextView> &text, const AlnRecords &aln, JunctionMap &j2o) { return assemble_junctions_view(text, "", &aln, j2o); }
std::string assemble_junctions_rows(const RowParts &rows, const AlnRecords &aln, JunctionMap &j2o) { return assemble_junctions_view(std::vector<TextView>(), "", &aln, j2o, &rows); }

// texts: the rows in file order, every piece whole rows; pre: the rows as their writer kept them (no text to parse)
static std::string assemble_junctions_view(const std::vector<TextView> &texts, const std::string &clip_bam, const AlnRecords *mem, JunctionMap &j2o, const RowParts *pre)
{
	RecSource source;
	source.mem = mem;
	if (!mem && ssvh_bam_open(clip_bam.c_str(), &source.bam) != 0) return "[main_samview] fail to open file for reading.";
	source.hash_names();
	RecSource *const bam = &source;
	struct Closer { RecSource *s; ~Closer() { if (s->bam) ssvh_bam_close(s->bam); } } closer{bam};
	std::vector<ClipRow> rows; // the rows that share `current` (the reference's multimap is cleared at every group change)
	AlignMap aligns;
	Str current;               // last_clipped_seq (a view: the rows' text outlives the join)
	uint64_t cur_h = 0;        // its hash
	ssvh_record rec;
	std::string failure;
	// An alignment enters the group's map under (name, (chr, pos)).  An UNALIGNED record (flag 4: type 'n', "Exogenous" / -1) makes no junction
	// (GetJunction returns at once, getsv.cpp:1726), shares its key with every other unaligned record of the group and with no aligned one - leaving
	// it out of the map changes nothing in the table, and 5.5 M of a whole-genome sample's 5.52 M clipped sequences are of that kind: they only
	// drive the group bookkeeping (names compared, nothing built).
	auto file_under = [&](const Str &name) {
		if (rec.flag & 4) return;
		AlignInfo a;
		align_info_of(bam, rec, a);
		aligns.insert(std::make_pair(std::make_pair(name.str(), std::make_pair(a.chr, a.pos)), a));
	};
	// one row of clip.gz, in file order (false: clip.bam could not be read)
	auto on_row = [&](ClipRow &row) -> bool {
		if (current.n == 0 || (cur_h == row.h && current == row.clipped_seq)) { current = row.clipped_seq; cur_h = row.h; rows.push_back(row); return true; }
		// a new clipped sequence: consume the alignments of the current group
		int rc;
		while ((rc = bam->next(&rec)) == 1) {
			if (is_hard_clip(rec)) continue;
			if (cur_h == bam->h && current.equals(rec.qname)) { file_under(current); continue; }
			flush_group(rows, aligns, j2o);
			rows.clear(); aligns.clear();
			file_under(current); // filed under the OLD name
			current = row.clipped_seq; cur_h = row.h;
			rows.push_back(row);
			break;
		}
		if (rc < 0) { failure = ssvh_last_error(); return false; }
		// clip.bam exhausted before the group ended: like the reference, the row is dropped and nothing changes
		return true;
	};
	std::vector<ClipRow> parsed;
	std::deque<std::string> store; // (the rows are views: what the stream loop below extracts lives here until the join is over - the last group is flushed at the end)
	const bool timing = getenv("SSV_TIMING") != nullptr;
	const auto t0 = std::chrono::steady_clock::now();
	if (pre) {
		for (const std::vector<ClipRow> *part : *pre) for (const ClipRow &r : *part) { ClipRow row = r; if (!on_row(row)) return failure; }
		if (timing) std::cerr << "[timing] (junction stage: rows as their writer kept them, joined with the re-alignments in " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << " s)" << std::endl;
	} else if (parse_rows_parallel(texts, parsed)) {
		const auto t1 = std::chrono::steady_clock::now();
		for (auto &row : parsed) if (!on_row(row)) return failure;
		if (timing) std::cerr << "[timing] (junction stage: " << parsed.size() << " rows parsed in " << std::chrono::duration<double>(t1 - t0).count() << " s, joined with clip.bam in "
		                      << std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() << " s)" << std::endl;
		std::vector<ClipRow>().swap(parsed);
	} else {
		std::string whole;
		for (const TextView &text : texts) whole.append(text.data(), text.size());
		std::istringstream fin(whole);
		std::string chr, cigar, aligned_seq, aligned_qual, clipped_seq, clipped_qual, rest;
		auto keep = [&](const std::string &v) { store.push_back(v); return Str{store.back().data(), store.back().size()}; };
		while (fin >> chr) {
			ClipRow row;
			fin >> row.pos >> row.side >> cigar >> aligned_seq >> aligned_qual >> clipped_seq >> clipped_qual >> row.support;
			std::getline(fin, rest);
			row.chr = keep(chr); row.cigar = keep(cigar); row.aligned_seq = keep(aligned_seq); row.clipped_seq = keep(clipped_seq); row.clipped_qual = keep(clipped_qual);
			row.h = text_hash(row.clipped_seq.p, row.clipped_seq.n);
			if (!on_row(row)) return failure;
		}
	}
	int rc;
	while ((rc = bam->next(&rec)) == 1) { // tail: no hard-clip test here
		if (cur_h == bam->h && current.equals(rec.qname)) file_under(current);
		else break;
	}
	flush_group(rows, aligns, j2o);
	return "";
}

std::string assemble_junctions(const std::string &clipfile, const std::string &clip_bam, JunctionMap &j2o)
{
	const auto t0 = std::chrono::steady_clock::now();
	static const bool serial = getenv("SSV_SERIAL") && strstr(getenv("SSV_SERIAL"), "gz"); // SSV_SERIAL=gz (tests): one inflate stream
	std::unique_ptr<char[]> buf;
	size_t n_buf = 0;
	std::string text;
	TextView view{nullptr, 0};
	if (!serial && slurp_gz_members(clipfile, buf, n_buf)) view = TextView{buf.get(), n_buf};
	else {
		const std::string err = slurp_gz(clipfile, text);
		if (!err.empty()) return err;
		view = TextView{text.data(), text.size()};
	}
	if (getenv("SSV_TIMING")) std::cerr << "[timing] (junction stage: " << view.size() << " bytes of rows read in " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << " s)" << std::endl;
	return assemble_junctions_view(std::vector<TextView>(1, view), clip_bam, nullptr, j2o);
}

void merge_junctions(JunctionMap &j2o, int search_length)
{
	JunctionMap::iterator it = j2o.begin();
	while (it != j2o.end()) {
		if (it->second.up.right_clipped > 0 || it->second.up.left_clipped > 0) { ++it; continue; } // anchors need an unclipped up side
		JunctionMap::iterator nb = it;
		++nb;
		bool absorbed_into_neighbour = false;
		while (nb != j2o.end() && it->first.up_chr == nb->first.up_chr && it->first.down_chr == nb->first.down_chr && it->first.up_strand == nb->first.up_strand &&
		       it->first.down_strand == nb->first.down_strand && nb->first.up_pos - it->first.up_pos <= search_length) {
			if (!(std::abs(nb->first.down_pos - it->first.down_pos) <= search_length && nb->second.down.left_clipped == 0)) { ++nb; continue; }
			OtherInfo &A = it->second, &B = nb->second;
			std::string up1, down1, up2, down2;
			const bool plus = it->first.up_strand == '+', minus = it->first.up_strand == '-';
			if (A.up.cigar_vec.size() == 1 && B.up.cigar_vec.size() == 1) {
				// shift by the up-side offset: the later junction's up sequence carries `mh` extra bases that belong to the down side
				const int mh = nb->first.up_pos - it->first.up_pos;
				if ((plus && B.up.seq.length() < (size_t)(mh + 5)) || (minus && A.up.seq.length() < (size_t)(mh + 5))) { ++nb; continue; }
				if (plus) {
					up1 = A.up.seq; down1 = A.down.seq;
					up2 = B.up.seq.substr(0, B.up.seq.length() - (size_t)mh);
					down2 = B.up.seq.substr(B.up.seq.length() - (size_t)mh) + B.down.seq;
				} else {
					up1 = A.up.seq.substr(0, A.up.seq.length() - (size_t)mh);
					down1 = A.up.seq.substr(A.up.seq.length() - (size_t)mh) + A.down.seq;
					up2 = B.up.seq; down2 = B.down.seq;
				}
			} else if (A.down.cigar_vec.size() == 1 && B.down.cigar_vec.size() == 1) {
				const int mh = std::abs(nb->first.down_pos - it->first.down_pos);
				if ((plus && A.down.seq.length() < (size_t)(mh + 5)) || (minus && B.down.seq.length() < (size_t)(mh + 5))) { ++nb; continue; }
				if (plus) {
					down1 = A.down.seq.substr((size_t)mh); down2 = B.down.seq;
					up1 = A.up.seq + A.down.seq.substr(0, (size_t)mh); up2 = B.up.seq;
				} else {
					down1 = A.down.seq; down2 = B.down.seq.substr((size_t)mh);
					up1 = A.up.seq; up2 = B.up.seq + B.down.seq.substr(0, (size_t)mh);
				}
			}
			if (!(match_end_first(up1, up2) >= 0.85 && match_begin_first(down1, down2) >= 0.85)) { ++nb; continue; } // empty strings give NaN -> no merge
			if (B.up.uniq > A.up.uniq) A.up.uniq = B.up.uniq;
			if (B.down.uniq > A.down.uniq) A.down.uniq = B.down.uniq;
			const bool a_open = A.microhomology == -1, b_open = B.microhomology == -1;
			if (a_open && b_open) {
				A.up.support += B.up.support; A.down.support += B.down.support;
				if ((A.up.support != 0 && B.down.support != 0) || (A.down.support != 0 && B.up.support != 0)) A.microhomology = nb->first.up_pos - it->first.up_pos;
				j2o.erase(nb++);
			} else if (!a_open && b_open) {
				A.up.support += B.up.support; A.down.support += B.down.support;
				j2o.erase(nb++);
			} else if (a_open && !b_open) {
				B.up.support += A.up.support; B.down.support += A.down.support;
				absorbed_into_neighbour = true;
			} else {
				if (A.up.support > B.up.support || A.down.support == B.down.support) { A.up.support += B.up.support; j2o.erase(nb++); }
				else if (A.up.support == B.up.support || A.down.support > B.down.support) { A.down.support += B.down.support; j2o.erase(nb++); }
				else if (B.up.support > A.up.support && A.down.support == B.down.support) { B.up.support += A.up.support; absorbed_into_neighbour = true; }
				else if (B.down.support > A.down.support && B.up.support == A.up.support) { B.down.support += A.down.support; absorbed_into_neighbour = true; }
				else ++nb;
			}
			if (absorbed_into_neighbour) break;
		}
		if (absorbed_into_neighbour) j2o.erase(it++); else ++it;
	}
}

} // namespace seeksv
