// junction_stage.cpp - see junction_stage.h.  Restated from the behaviour of the reference, quirks included (they change the table):
//   * clip rows are grouped while consecutive rows carry the same clipped sequence; clip.bam records are consumed in file order
//     and belong to a group while their read name equals that sequence (the re-aligner must keep FASTQ order and use the sequence
//     as the read name, as `bwa mem` does with getclip's FASTQ);
//   * the first record that ends a group is filed under the PREVIOUS group's name (getsv.h:502), which decides its place in the
//     ordered alignment map and therefore the order in which junctions are created;
//   * only the first clip row of a group is paired with the group's alignments (the inner iterator is never reset, getsv.h:489-498);
//   * hard-clipped records are skipped in the main loop but not in the tail loop (getsv.h:476 vs :512-527).
#include "junction_stage.h"

#include <zlib.h>

#include <cstdlib>
#include <sstream>

#include "seeksv_host.h"

namespace seeksv {

namespace {

struct ClipRow { // one row of clip.gz as the join sees it: aligned part first, whatever the side (getsv.h:460-461)
	std::string chr; int pos = 0; char side = 0; CigarVec cigar_vec;
	std::string aligned_seq, clipped_seq, clipped_qual; int support = 0;
};

struct AlignInfo { // getsv.h:27-44
	std::string chr; int pos = 0, len = 0; char strand = 0; CigarVec cigar_vec; std::string seq; int left_clipped = 0, right_clipped = 0; char type = 0;
};

typedef std::map<std::pair<std::string, std::pair<std::string, int>>, AlignInfo> AlignMap; // (name, (chr, pos)) -> alignment

void reverse_cigar(CigarVec &v) // ReverseCigar, getsv.cpp:453
{
	for (size_t i = 0, n = v.size(); i < n / 2; ++i) std::swap(v[i], v[n - 1 - i]);
}

// GetAlignInfo, getsv.cpp:25-71
void align_info_of(const ssvh_bam *bam, const ssvh_record &r, AlignInfo &a)
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
	const char *nm = ssvh_bam_target_name(bam, r.tid);
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
	std::string aligned_seq = row.aligned_seq, clipped_seq = row.clipped_seq;
	CigarVec cigar_vec = row.cigar_vec;
	const std::string &chr = row.chr;
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

std::string slurp_gz(const std::string &path, std::string &out)
{
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

std::string assemble_junctions(const std::string &clipfile, const std::string &clip_bam, JunctionMap &j2o)
{
	std::string text, err = slurp_gz(clipfile, text);
	if (!err.empty()) return err;
	return assemble_junctions_text(text, clip_bam, j2o);
}

// the same with the rows of clip.gz already in memory (`seeksv run`: getclip has just written them)
std::string assemble_junctions_text(const std::string &text, const std::string &clip_bam, JunctionMap &j2o)
{
	ssvh_bam *bam = nullptr;
	if (ssvh_bam_open(clip_bam.c_str(), &bam) != 0) return "[main_samview] fail to open file for reading.";
	std::istringstream fin(text);
	std::vector<ClipRow> rows; // the rows that share `current` (the reference's multimap is cleared at every group change)
	AlignMap aligns;
	std::string current;       // last_clipped_seq
	ssvh_record rec;
	std::string chr, cigar, aligned_qual, rest;
	while (fin >> chr) {
		ClipRow row;
		row.chr = chr;
		fin >> row.pos >> row.side >> cigar >> row.aligned_seq >> aligned_qual >> row.clipped_seq >> row.clipped_qual >> row.support;
		std::getline(fin, rest);
		row.cigar_vec = parse_cigar(cigar);
		if (current.empty() || current == row.clipped_seq) { rows.push_back(row); current = row.clipped_seq; continue; }
		// a new clipped sequence: consume the alignments of the current group
		int rc;
		while ((rc = ssvh_bam_next_record(bam, &rec)) == 1) {
			if (is_hard_clip(rec)) continue;
			AlignInfo a;
			align_info_of(bam, rec, a);
			if (current == rec.qname) { aligns.insert(std::make_pair(std::make_pair(current, std::make_pair(a.chr, a.pos)), a)); continue; }
			flush_group(rows, aligns, j2o);
			rows.clear(); aligns.clear();
			rows.push_back(row);
			aligns.insert(std::make_pair(std::make_pair(current, std::make_pair(a.chr, a.pos)), a)); // filed under the OLD name
			current = row.clipped_seq;
			break;
		}
		if (rc < 0) { std::string e = ssvh_last_error(); ssvh_bam_close(bam); return e; }
		// clip.bam exhausted before the group ended: like the reference, the row is dropped and nothing changes
	}
	int rc;
	while ((rc = ssvh_bam_next_record(bam, &rec)) == 1) { // tail: no hard-clip test here
		AlignInfo a;
		align_info_of(bam, rec, a);
		if (current == rec.qname) aligns.insert(std::make_pair(std::make_pair(current, std::make_pair(a.chr, a.pos)), a));
		else break;
	}
	flush_group(rows, aligns, j2o);
	ssvh_bam_close(bam);
	return "";
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
