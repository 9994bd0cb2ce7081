// somatic_stage.cpp - see somatic_stage.h.  The reference spells the twelve (strand pair x microhomology x which side has no reads)
// cases out one by one; they are all made of two kinds of look-up, restated here once each:
//   exact probe  - the clusters of ONE (contig, pos) bin in file order; the first whose right string matches `right` from its beginning
//                  and whose left string matches `left` from its end (both at >= min_map_rate) gives its support;
//   range probe  - the clusters of a position interval in key order; the first for which the 10-base anchored comparison succeeds.
#include "somatic_stage.h"

#include <fstream>
#include <sstream>

#include "junction_stage.h"

namespace seeksv {

std::string load_normal_clusters(const std::string &clip_file, int min_len_of_clipped_seq, ClusterMap &clip3, ClusterMap &clip5, std::string &warnings)
{
	std::string text, err = slurp_gz(clip_file, text);
	if (!err.empty()) return err;
	std::istringstream fin(text);
	std::string chr, cigar, aligned_seq, aligned_qual, clipped_seq, clipped_qual, rest;
	int support = 0, pos = 0;
	char side = 0;
	// the same extraction sequence as the reference, so that a malformed row behaves the same (stream fails, loop ends after this row)
	while (fin >> chr) {
		fin >> pos >> side >> cigar >> aligned_seq >> aligned_qual >> clipped_seq >> clipped_qual >> support;
		std::getline(fin, rest);
		if (clipped_seq.length() < (size_t)min_len_of_clipped_seq) continue; // unsigned compare, somatic.h:53
		NormalCluster c;
		c.support = support;
		if (side == '3') { c.seq_left = aligned_seq; c.seq_right = clipped_seq; clip3.insert(std::make_pair(std::make_pair(chr, pos), c)); }
		else if (side == '5') { c.seq_left = clipped_seq; c.seq_right = aligned_seq; clip5.insert(std::make_pair(std::make_pair(chr, pos), c)); }
		else warnings += "Error:The orientation of soft-clipped reads must be 3 or 5 in position " + chr + ":" + std::to_string(pos) + "\n";
	}
	return "";
}

int anchored_compare(const std::string &seq1, const std::string &seq2, const std::string &seq3, const std::string &seq4, double match_rate)
{
	if ((int)seq2.length() < 10) return -1;
	const size_t hit = seq4.find(seq2.substr(0, 10));
	if (hit == std::string::npos) return -1;
	const std::string head = seq3 + seq4.substr(0, hit), tail = seq4.substr(hit);
	return (match_end_first(seq1, head) >= match_rate && match_begin_first(seq2, tail) >= match_rate) ? (int)hit : -1;
}

namespace {

int exact_probe(const ClusterMap &m, const std::string &chr, int pos, const std::string &left, const std::string &right, double rate)
{
	auto range = m.equal_range(std::make_pair(chr, pos));
	for (auto it = range.first; it != range.second; ++it)
		if (match_begin_first(right, it->second.seq_right) >= rate && match_end_first(left, it->second.seq_left) >= rate) return it->second.support;
	return 0;
}

// cluster_first: Compare(cluster.left, cluster.right, a, b); else Compare(a, b, cluster.left, cluster.right)
int range_probe(const ClusterMap &m, const std::string &chr, int lo, int hi, bool cluster_first, const std::string &a, const std::string &b, double rate)
{
	for (auto it = m.lower_bound(std::make_pair(chr, lo)); it != m.end() && it->first.first == chr && it->first.second <= hi; ++it) {
		const NormalCluster &c = it->second;
		const int r = cluster_first ? anchored_compare(c.seq_left, c.seq_right, a, b, rate) : anchored_compare(a, b, c.seq_left, c.seq_right, rate);
		if (r != -1) return c.support;
	}
	return 0;
}

} // namespace

std::string scan_tumor_table(const std::string &tumor_sv_file, const ClusterMap &clip3, const ClusterMap &clip5, int offset, double rate, int mean_insert_size,
                             std::vector<SomaticRow> &rows)
{
	std::ifstream fin(tumor_sv_file.c_str());
	if (!fin) return "Error: Cannot open output file " + tumor_sv_file; // (sic) somatic.cpp:19
	std::string up_chr, down_chr, sv_type, up_cigar, down_cigar, up_seq, down_seq, rest;
	int up_pos = 0, down_pos = 0, up_reads = 0, down_reads = 0, mh = 0, abnormal = 0, up_depth = 0, down_depth = 0, uu = 0, ud = 0, du = 0, dd = 0;
	double up_rate = 0, down_rate = 0;
	char us = 0, ds = 0;
	while (fin >> up_chr) {
		SomaticRow row;
		if (up_chr[0] == '@') {
			std::getline(fin, rest);
			row.kind = SomaticRow::HEADER;
			row.text = up_chr + rest + "\tleft_clip_read_NO_of_control\tright_clip_read_NO_of_control\tabnormal_read_pair_no_of_control";
			rows.push_back(row);
			continue;
		}
		fin >> up_pos >> us >> up_reads >> down_chr >> down_pos >> ds >> down_reads >> mh >> abnormal >> sv_type >> up_depth >> down_depth >> uu >> ud >> du >> dd >> up_rate >> down_rate >>
		    up_cigar >> down_cigar >> up_seq >> down_seq;
		std::getline(fin, rest);
		row.up_chr = up_chr; row.down_chr = down_chr; row.up_pos = up_pos; row.down_pos = down_pos; row.up_strand = us; row.down_strand = ds;
		{
			std::ostringstream o; // doubles go through the default ostream format again, like the reference's re-print
			o << up_chr << '\t' << up_pos << '\t' << us << '\t' << up_reads << '\t' << down_chr << '\t' << down_pos << '\t' << ds << '\t' << down_reads << '\t' << mh << '\t' << abnormal << '\t'
			  << sv_type << '\t' << up_depth << '\t' << down_depth << '\t' << uu << '\t' << ud << '\t' << du << '\t' << dd << '\t' << up_rate << '\t' << down_rate << '\t' << up_cigar << '\t'
			  << down_cigar << '\t' << up_seq << '\t' << down_seq;
			row.text = o.str();
		}
		const bool pp = us == '+' && ds == '+', pm = us == '+' && ds == '-', mp = us == '-' && ds == '+';
		if (!pp && !pm && !mp) {
			std::ostringstream o;
			o << "Error: Something error in line " << up_chr << '\t' << up_pos << '\t' << us << '\t' << up_reads << '\t' << down_chr << '\t' << down_pos << '\t' << ds << '\t' << down_reads << '\t'
			  << mh << '\t' << up_depth << '\t' << down_depth << '\t' << up_rate << '\t' << down_rate << '\t' << up_cigar << '\t' << down_cigar << '\t' << up_seq << '\t' << down_seq;
			row.kind = SomaticRow::MESSAGE; row.text = o.str();
			rows.push_back(row);
			continue;
		}
		std::string rc_up = up_seq, rc_down = down_seq;
		reverse_complement(rc_up);
		reverse_complement(rc_down);
		int left = 0, right = 0;
		bool tally = mean_insert_size != 0;
		if (mh != -1) {
			// both breakends were seen in the tumor: the normal's bins sit at the ends of the microhomology
			if (pp) {
				right = exact_probe(clip5, down_chr, down_pos, up_seq, down_seq, rate);
				if (down_seq.length() >= (size_t)mh) // unsigned compare, somatic.cpp:92
					left = exact_probe(clip3, up_chr, up_pos + mh, up_seq + down_seq.substr(0, (size_t)mh), down_seq.substr((size_t)mh), rate);
				tally = true; // somatic.cpp:111: not guarded by mean_insert_size != 0
			} else if (pm) {
				if ((size_t)mh > down_seq.length()) return "microhomology longer than the right sequence at " + up_chr + ":" + std::to_string(up_pos) + " (the reference aborts here)";
				left = exact_probe(clip3, up_chr, up_pos + mh, up_seq + down_seq.substr(0, (size_t)mh), down_seq.substr((size_t)mh), rate);
				right = exact_probe(clip3, down_chr, down_pos, rc_down, rc_up, rate);
			} else {
				if ((size_t)mh > up_seq.length()) return "microhomology longer than the left sequence at " + up_chr + ":" + std::to_string(up_pos) + " (the reference aborts here)";
				left = exact_probe(clip5, up_chr, up_pos, rc_down, rc_up, rate);
				const size_t cut = up_seq.length() - (size_t)mh;
				right = exact_probe(clip5, down_chr, down_pos - mh, up_seq.substr(0, cut), up_seq.substr(cut) + down_seq, rate);
			}
		} else if (up_reads == 0) {
			// only the right breakend was clipped in the tumor: exact bin on the right, anchored search near the left
			if (pp) { right = exact_probe(clip5, down_chr, down_pos, up_seq, down_seq, rate); left = range_probe(clip3, up_chr, up_pos, up_pos + offset, true, up_seq, down_seq, rate); }
			else if (pm) { right = exact_probe(clip3, down_chr, down_pos, rc_down, rc_up, rate); left = range_probe(clip3, up_chr, up_pos, up_pos + offset, true, up_seq, down_seq, rate); }
			else { right = exact_probe(clip5, down_chr, down_pos, up_seq, down_seq, rate); left = range_probe(clip5, up_chr, up_pos - offset, up_pos, false, rc_up, rc_down, rate); }
		} else if (down_reads == 0) {
			if (pp) { left = exact_probe(clip3, up_chr, up_pos, up_seq, down_seq, rate); right = range_probe(clip5, down_chr, down_pos - offset, down_pos, false, up_seq, down_seq, rate); }
			else if (pm) { left = exact_probe(clip3, up_chr, up_pos, up_seq, down_seq, rate); right = range_probe(clip3, down_chr, down_pos, down_pos + offset, true, rc_down, rc_up, rate); }
			else { left = exact_probe(clip5, up_chr, up_pos, rc_down, rc_up, rate); right = range_probe(clip5, down_chr, down_pos - offset, down_pos, false, up_seq, down_seq, rate); }
		} else {
			std::ostringstream o;
			o << "The tandem repeat length is error in postion: " << up_chr << '\t' << up_pos << '\t' << down_chr << '\t' << down_pos;
			row.kind = SomaticRow::MESSAGE; row.text = o.str();
			rows.push_back(row);
			continue;
		}
		row.normal_left_reads = left; row.normal_right_reads = right; row.tally = tally;
		rows.push_back(row);
	}
	return "";
}

} // namespace seeksv
