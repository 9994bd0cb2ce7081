// somatic_stage.cpp - see somatic_stage.h.  The reference spells the twelve (strand pair x microhomology x which side has no reads)
// cases out one by one; they are all made of two kinds of look-up, restated here once each:
//   exact probe  - the clusters of ONE (contig, pos) bin in file order; the first whose right string matches `right` from its beginning
//                  and whose left string matches `left` from its end (both at >= min_map_rate) gives its support;
//   range probe  - the clusters of a position interval in key order; the first for which the 10-base anchored comparison succeeds.
#include "somatic_stage.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <iostream>
#include <fstream>
#include <sstream>
#include <thread>

#include "../csrc/cpus.h"

namespace seeksv {

namespace {

template <class F> void on_threads(int nt, F f)
{
	if (nt <= 1) { f(0); return; }
	std::vector<std::thread> th;
	for (int w = 0; w < nt; ++w) th.emplace_back([&f, w] { f(w); });
	for (auto &t : th) t.join();
}

inline uint64_t key_of(uint32_t rank, int pos) { return (uint64_t)rank << 32 | (uint32_t)(pos + (int)0x80000000u); }

// (key, cluster) pairs in FILE order -> the multimap's order: by key, equal keys in file order.  getclip writes a contig's rows by ascending position
// (clip_reads.h:432-446), so a side's keys are a handful of ascending runs - one per contig, in the BAM's contig order, which is not the strings' -
// and putting the runs in order is all there is to do; any other file is sorted.
void order_side(std::vector<uint64_t> &key, std::vector<NormalCluster> &val, int nt)
{
	const size_t n = key.size();
	if (n < 2 || std::is_sorted(key.begin(), key.end())) return;
	struct Run { size_t first, last; }; // [first, last)
	std::vector<Run> runs;
	for (size_t i = 0, b = 0; i <= n; ++i)
		if (i == n || (i > b && key[i] < key[i - 1])) { runs.push_back(Run{b, i}); b = i; if (runs.size() > 65536) break; }
	std::vector<uint32_t> perm;
	bool by_runs = runs.size() <= 65536 && n < 0xffffffffull;
	if (by_runs) {
		std::stable_sort(runs.begin(), runs.end(), [&](const Run &a, const Run &b) { return key[a.first] < key[b.first]; });
		for (size_t r = 0; by_runs && r + 1 < runs.size(); ++r) {
			const uint64_t la = key[runs[r].last - 1], fb = key[runs[r + 1].first];
			by_runs = la < fb || (la == fb && runs[r].first < runs[r + 1].first);
		}
	}
	std::vector<uint64_t> k2(n);
	std::vector<NormalCluster> v2(n);
	if (by_runs) {
		std::vector<size_t> at(runs.size() + 1, 0);
		for (size_t r = 0; r < runs.size(); ++r) at[r + 1] = at[r] + (runs[r].last - runs[r].first);
		std::atomic<size_t> next{0};
		on_threads(std::min<int>(nt, (int)runs.size()), [&](int) {
			for (size_t r; (r = next++) < runs.size();) {
				std::copy(key.begin() + (ptrdiff_t)runs[r].first, key.begin() + (ptrdiff_t)runs[r].last, k2.begin() + (ptrdiff_t)at[r]);
				std::copy(val.begin() + (ptrdiff_t)runs[r].first, val.begin() + (ptrdiff_t)runs[r].last, v2.begin() + (ptrdiff_t)at[r]);
			}
		});
	} else {
		std::vector<std::pair<uint64_t, uint64_t>> o(n);
		for (size_t i = 0; i < n; ++i) o[i] = std::make_pair(key[i], (uint64_t)i);
		std::sort(o.begin(), o.end());
		for (size_t i = 0; i < n; ++i) { k2[i] = o[i].first; v2[i] = val[(size_t)o[i].second]; }
	}
	key.swap(k2);
	val.swap(v2);
}

} // namespace

int64_t ClusterIndex::rank_of(const std::string &chr) const
{
	const auto it = std::lower_bound(names_->begin(), names_->end(), chr);
	return it != names_->end() && *it == chr ? (int64_t)(it - names_->begin()) : -1;
}

std::pair<size_t, size_t> ClusterIndex::equal_range(const std::string &chr, int pos) const
{
	const int64_t r = rank_of(chr);
	if (r < 0) return std::make_pair((size_t)0, (size_t)0);
	const auto range = std::equal_range(key_.begin(), key_.end(), key_of((uint32_t)r, pos));
	return std::make_pair((size_t)(range.first - key_.begin()), (size_t)(range.second - key_.begin()));
}

// std::multimap::lower_bound((chr, lo)): a contig the file does not have still has its place among the names (the walk then stops at once: another contig)
size_t ClusterIndex::lower_bound(const std::string &chr, int lo) const
{
	const size_t r = (size_t)(std::lower_bound(names_->begin(), names_->end(), chr) - names_->begin());
	const bool present = r < names_->size() && (*names_)[r] == chr;
	return (size_t)(std::lower_bound(key_.begin(), key_.end(), present ? key_of((uint32_t)r, lo) : (uint64_t)r << 32) - key_.begin());
}

bool ClusterIndex::on_contig(size_t i, const std::string &chr) const { return (*names_)[(size_t)(key_[i] >> 32)] == chr; }

std::string load_normal_clusters(const std::string &clip_file, int min_len_of_clipped_seq, NormalClusters &out, std::string &warnings)
{
	static const bool serial_gz = getenv("SSV_SERIAL") && strstr(getenv("SSV_SERIAL"), "gz");
	const bool timing = getenv("SSV_TIMING") != nullptr;
	auto t_last = std::chrono::steady_clock::now();
	std::string laps;
	auto lap = [&](const char *what) {
		if (!timing) return;
		const auto now = std::chrono::steady_clock::now();
		char b[64]; snprintf(b, sizeof(b), " %s %.3f", what, std::chrono::duration<double>(now - t_last).count());
		laps += b; t_last = now;
	};
	TextView view{nullptr, 0};
	size_t n_text = 0;
	if (!serial_gz && slurp_gz_members(clip_file, out.text, n_text)) view = TextView{out.text.get(), n_text};
	else {
		const std::string err = slurp_gz(clip_file, out.text_s);
		if (!err.empty()) return err;
		view = TextView{out.text_s.data(), out.text_s.size()};
	}
	lap("inflate");
	std::vector<ClipRow> rows;
	if (!parse_rows_parallel(std::vector<TextView>(1, view), rows)) {
		// the same extraction sequence as the reference (somatic.h:45-52), so that a malformed row behaves the same (stream fails, loop ends after this row)
		rows.clear();
		std::istringstream fin(std::string(view.data(), view.size()));
		std::string chr, cigar, aligned_seq, aligned_qual, clipped_seq, clipped_qual, rest;
		auto keep = [&](const std::string &v) { out.store.push_back(v); return Str{out.store.back().data(), out.store.back().size()}; };
		int support = 0, pos = 0;
		char side = 0;
		while (fin >> chr) {
			fin >> pos >> side >> cigar >> aligned_seq >> aligned_qual >> clipped_seq >> clipped_qual >> support;
			std::getline(fin, rest);
			ClipRow row;
			row.chr = keep(chr); row.pos = pos; row.side = side; row.aligned_seq = keep(aligned_seq); row.clipped_seq = keep(clipped_seq); row.support = support;
			rows.push_back(row);
		}
	}
	lap("parse");
	out.rows = rows.size();
	const size_t n = rows.size();
	static const size_t rows_per_thread = [] { const char *e = getenv("SSV_ROWS_CHUNK_KB"); const long kb = e ? atol(e) : 64; return (size_t)(kb < 1 ? 1 : kb) * 64; }(); // (tests: several threads on few rows)
	const int nt = (int)std::max<size_t>(1, std::min<size_t>({(size_t)ssv::effective_cpus(), (size_t)64, n / rows_per_thread + 1}));
	// the file's contig names: a row's name is its predecessor's nearly always
	std::vector<std::vector<std::string>> seen((size_t)nt);
	on_threads(nt, [&](int w) {
		const size_t lo = n * (size_t)w / (size_t)nt, hi = n * (size_t)(w + 1) / (size_t)nt;
		std::vector<std::string> &mine = seen[(size_t)w];
		for (size_t i = lo; i < hi; ++i)
			if (i == lo || !(rows[i].chr == rows[i - 1].chr)) { const std::string name = rows[i].chr.str(); if (std::find(mine.begin(), mine.end(), name) == mine.end()) mine.push_back(name); }
	});
	for (auto &v : seen) out.names.insert(out.names.end(), v.begin(), v.end());
	std::sort(out.names.begin(), out.names.end());
	out.names.erase(std::unique(out.names.begin(), out.names.end()), out.names.end());
	out.clip3.names_ = out.clip5.names_ = &out.names;
	// per thread: how many rows of its share go to either side; then every thread writes its rows at its offsets - file order within a side is kept
	std::vector<size_t> n3((size_t)nt + 1, 0), n5((size_t)nt + 1, 0);
	std::vector<std::string> warn((size_t)nt);
	const size_t min_len = (size_t)min_len_of_clipped_seq; // unsigned compare, somatic.h:53
	on_threads(nt, [&](int w) {
		const size_t lo = n * (size_t)w / (size_t)nt, hi = n * (size_t)(w + 1) / (size_t)nt;
		size_t c3 = 0, c5 = 0;
		for (size_t i = lo; i < hi; ++i) {
			if (rows[i].clipped_seq.n < min_len) continue;
			if (rows[i].side == '3') ++c3; else if (rows[i].side == '5') ++c5;
		}
		n3[(size_t)w + 1] = c3; n5[(size_t)w + 1] = c5;
	});
	for (int w = 0; w < nt; ++w) { n3[(size_t)w + 1] += n3[(size_t)w]; n5[(size_t)w + 1] += n5[(size_t)w]; }
	out.clip3.key_.resize(n3[(size_t)nt]); out.clip3.val_.resize(n3[(size_t)nt]);
	out.clip5.key_.resize(n5[(size_t)nt]); out.clip5.val_.resize(n5[(size_t)nt]);
	on_threads(nt, [&](int w) {
		const size_t lo = n * (size_t)w / (size_t)nt, hi = n * (size_t)(w + 1) / (size_t)nt;
		size_t a3 = n3[(size_t)w], a5 = n5[(size_t)w];
		uint32_t rank = 0;
		for (size_t i = lo; i < hi; ++i) {
			const ClipRow &r = rows[i];
			if (i == lo || !(r.chr == rows[i - 1].chr)) rank = (uint32_t)(std::lower_bound(out.names.begin(), out.names.end(), r.chr.str()) - out.names.begin());
			if (r.clipped_seq.n < min_len) continue;
			NormalCluster c;
			c.support = r.support;
			if (r.side == '3') { c.seq_left = r.aligned_seq; c.seq_right = r.clipped_seq; out.clip3.key_[a3] = key_of(rank, r.pos); out.clip3.val_[a3++] = c; }
			else if (r.side == '5') { c.seq_left = r.clipped_seq; c.seq_right = r.aligned_seq; out.clip5.key_[a5] = key_of(rank, r.pos); out.clip5.val_[a5++] = c; }
			else warn[(size_t)w] += "Error:The orientation of soft-clipped reads must be 3 or 5 in position " + r.chr.str() + ":" + std::to_string(r.pos) + "\n";
		}
	});
	for (auto &t : warn) warnings += t;
	lap("sides");
	on_threads(2, [&](int w) { ClusterIndex &ix = w ? out.clip5 : out.clip3; order_side(ix.key_, ix.val_, std::max(1, nt / 2)); });
	lap("order");
	if (timing) std::cerr << "[timing] (normal clusters: " << n << " rows, " << view.size() << " bytes of text, " << nt << " threads:" << laps << ")" << std::endl;
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

// CompareStringEndFirst / CompareStringBeginFirst (clip_reads.cpp:194-217) with a cluster's string as a view into the rows' text
double end_first(const std::string &a, const Str &b)
{
	int la = (int)a.size(), lb = (int)b.n, n = la < lb ? la : lb, m = 0;
	for (int i = 0; i < n; ++i) if (a[(size_t)(la - 1 - i)] == b.p[lb - 1 - i]) ++m;
	return (double)m / n;
}
double begin_first(const std::string &a, const Str &b)
{
	int n = (int)(a.size() < b.n ? a.size() : b.n), m = 0;
	for (int i = 0; i < n; ++i) if (a[(size_t)i] == b.p[i]) ++m;
	return (double)m / n;
}

int exact_probe(const ClusterIndex &m, const std::string &chr, int pos, const std::string &left, const std::string &right, double rate)
{
	const auto range = m.equal_range(chr, pos);
	for (size_t i = range.first; i < range.second; ++i)
		if (begin_first(right, m.at(i).seq_right) >= rate && end_first(left, m.at(i).seq_left) >= rate) return m.at(i).support;
	return 0;
}

// cluster_first: Compare(cluster.left, cluster.right, a, b); else Compare(a, b, cluster.left, cluster.right)
int range_probe(const ClusterIndex &m, const std::string &chr, int lo, int hi, bool cluster_first, const std::string &a, const std::string &b, double rate)
{
	for (size_t i = m.lower_bound(chr, lo); i < m.size() && m.on_contig(i, chr) && m.pos(i) <= hi; ++i) {
		const NormalCluster &c = m.at(i);
		const std::string cl = c.seq_left.str(), cr = c.seq_right.str();
		const int r = cluster_first ? anchored_compare(cl, cr, a, b, rate) : anchored_compare(a, b, cl, cr, rate);
		if (r != -1) return c.support;
	}
	return 0;
}

} // namespace

std::string parse_tumor_table(const std::string &tumor_sv_file, int mean_insert_size, std::vector<SomaticRow> &rows)
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
		row.microhomology = mh; row.up_reads = up_reads; row.down_reads = down_reads; row.up_seq = up_seq; row.down_seq = down_seq;
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
		row.tally = mean_insert_size != 0;
		if (mh != -1) {
			if (pp) row.tally = true; // somatic.cpp:111: not guarded by mean_insert_size != 0
			else if (pm) { if ((size_t)mh > down_seq.length()) return "microhomology longer than the right sequence at " + up_chr + ":" + std::to_string(up_pos) + " (the reference aborts here)"; }
			else if ((size_t)mh > up_seq.length()) return "microhomology longer than the left sequence at " + up_chr + ":" + std::to_string(up_pos) + " (the reference aborts here)";
		} else if (up_reads != 0 && down_reads != 0) {
			std::ostringstream o;
			o << "The tandem repeat length is error in postion: " << up_chr << '\t' << up_pos << '\t' << down_chr << '\t' << down_pos;
			row.kind = SomaticRow::MESSAGE; row.text = o.str(); row.tally = false;
		}
		rows.push_back(row);
	}
	return "";
}

void probe_normal_clusters(std::vector<SomaticRow> &rows, const ClusterIndex &clip3, const ClusterIndex &clip5, int offset, double rate)
{
	for (SomaticRow &row : rows) {
		if (row.kind != SomaticRow::JUNCTION) continue;
		const std::string &up_chr = row.up_chr, &down_chr = row.down_chr, &up_seq = row.up_seq, &down_seq = row.down_seq;
		const int up_pos = row.up_pos, down_pos = row.down_pos, mh = row.microhomology;
		const bool pp = row.up_strand == '+' && row.down_strand == '+', pm = row.up_strand == '+' && row.down_strand == '-';
		std::string rc_up = up_seq, rc_down = down_seq;
		reverse_complement(rc_up);
		reverse_complement(rc_down);
		int left = 0, right = 0;
		if (mh != -1) {
			// both breakends were seen in the tumor: the normal's bins sit at the ends of the microhomology
			if (pp) {
				right = exact_probe(clip5, down_chr, down_pos, up_seq, down_seq, rate);
				if (down_seq.length() >= (size_t)mh) // unsigned compare, somatic.cpp:92
					left = exact_probe(clip3, up_chr, up_pos + mh, up_seq + down_seq.substr(0, (size_t)mh), down_seq.substr((size_t)mh), rate);
			} else if (pm) {
				left = exact_probe(clip3, up_chr, up_pos + mh, up_seq + down_seq.substr(0, (size_t)mh), down_seq.substr((size_t)mh), rate);
				right = exact_probe(clip3, down_chr, down_pos, rc_down, rc_up, rate);
			} else {
				left = exact_probe(clip5, up_chr, up_pos, rc_down, rc_up, rate);
				const size_t cut = up_seq.length() - (size_t)mh;
				right = exact_probe(clip5, down_chr, down_pos - mh, up_seq.substr(0, cut), up_seq.substr(cut) + down_seq, rate);
			}
		} else if (row.up_reads == 0) {
			// only the right breakend was clipped in the tumor: exact bin on the right, anchored search near the left
			if (pp) { right = exact_probe(clip5, down_chr, down_pos, up_seq, down_seq, rate); left = range_probe(clip3, up_chr, up_pos, up_pos + offset, true, up_seq, down_seq, rate); }
			else if (pm) { right = exact_probe(clip3, down_chr, down_pos, rc_down, rc_up, rate); left = range_probe(clip3, up_chr, up_pos, up_pos + offset, true, up_seq, down_seq, rate); }
			else { right = exact_probe(clip5, down_chr, down_pos, up_seq, down_seq, rate); left = range_probe(clip5, up_chr, up_pos - offset, up_pos, false, rc_up, rc_down, rate); }
		} else { // down_reads == 0 (rows with both counts and no microhomology became messages in parse_tumor_table)
			if (pp) { left = exact_probe(clip3, up_chr, up_pos, up_seq, down_seq, rate); right = range_probe(clip5, down_chr, down_pos - offset, down_pos, false, up_seq, down_seq, rate); }
			else if (pm) { left = exact_probe(clip3, up_chr, up_pos, up_seq, down_seq, rate); right = range_probe(clip3, down_chr, down_pos, down_pos + offset, true, rc_down, rc_up, rate); }
			else { left = exact_probe(clip5, up_chr, up_pos, rc_down, rc_up, rate); right = range_probe(clip5, down_chr, down_pos - offset, down_pos, false, up_seq, down_seq, rate); }
		}
		row.normal_left_reads = left; row.normal_right_reads = right;
	}
}

std::string scan_tumor_table(const std::string &tumor_sv_file, const ClusterIndex &clip3, const ClusterIndex &clip5, int offset, double rate, int mean_insert_size,
                             std::vector<SomaticRow> &rows)
{
	const std::string err = parse_tumor_table(tumor_sv_file, mean_insert_size, rows);
	if (err.empty()) probe_normal_clusters(rows, clip3, clip5, offset, rate);
	return err;
}

} // namespace seeksv
