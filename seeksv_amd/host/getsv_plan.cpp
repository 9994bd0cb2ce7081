// getsv_plan.cpp - the ordered-container bookkeeping the reference wraps around its two BAM passes.
//
// The GPU answers three primitive questions about a BAM: "how many reads pass the discordant
// predicate chain for junction j inside window [beg,end)", "what is the sum of per-column depth
// over [beg,end]" and "what is the depth at column c".  This file turns the reference's view
// (multimap<Junction,...>, GetBreak getsv.cpp:752-802, MergeOverlap getsv.cpp:804-835, the window
// arithmetic of FindDiscordantReadPairs getsv.cpp:1032-1060 and the per-column lookup rules of
// main_depth bam2depth.cpp:82-124) into those primitive queries and folds the answers back, keeping
// the reference's int/unsigned conversions so that wrapped and empty flank windows come out the same.
#include "seeksv_host.h"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <map>
#include <string>
#include <tuple>
#include <vector>

namespace {

using RangeKey = std::tuple<std::string, uint32_t, uint32_t>; // ChrRange::operator< == lexicographic (getsv.h:238-256)

struct Piece { int64_t range_idx; };

} // namespace

struct ssvh_plan {
	int64_t n_junctions = 0;
	std::vector<ssv_junction> dev_junctions;          // only junctions whose up_chr is in the header
	std::vector<int64_t> dev_junction_of;             // junction j -> index in dev_junctions or -1
	std::vector<ssv_interval> windows, ranges, points;
	// fold-back tables
	std::vector<std::vector<int64_t>> range_pieces;   // reference range r -> device range indices
	std::vector<RangeKey> ref_ranges;                 // sorted unique
	std::vector<int64_t> flank_of;                    // [4*j+k] -> index in ref_ranges
	std::vector<int64_t> up_point, down_point;        // junction -> device point index or -1 (depth stays 0)
	std::vector<int64_t> extra_point;                 // extra point -> device point index or -1
};

extern "C" {

int ssvh_plan_create(const ssvh_bam *bam, const ssvh_junction_in *junctions, int64_t n_junctions,
                     const char *const *extra_point_chr, const int32_t *extra_point_pos, int64_t n_extra_points,
                     int32_t mean, int32_t sd, int32_t times, int32_t flank_length, ssvh_plan **out)
{
	ssvh_plan *p = new ssvh_plan();
	p->n_junctions = n_junctions;
	std::map<std::string, int> name2tid; // StoreSeqName2Tid, cluster.cpp:206-216 (map::insert: first wins)
	const int32_t nt = ssvh_bam_n_targets(bam);
	for (int32_t i = 0; i < nt; ++i) name2tid.insert(std::make_pair(std::string(ssvh_bam_target_name(bam, i)), i));
	auto get_tid = [&](const char *name) -> int { auto it = name2tid.find(name); return it == name2tid.end() ? -1 : it->second; };

	// ---- discordant windows, getsv.cpp:1032-1060 ----
	int min_ins = mean - sd * times, max_ins = mean + sd * times;
	if (min_ins < 0) min_ins = 0;
	(void)min_ins;
	p->dev_junction_of.assign((size_t)n_junctions, -1);
	for (int64_t j = 0; j < n_junctions; ++j) {
		const ssvh_junction_in &J = junctions[j];
		int tid = get_tid(J.up_chr);
		if (tid == -1) continue; // keeps its previous count, getsv.cpp:1043
		unsigned chr_length = (unsigned)ssvh_bam_target_len(bam, tid);
		int beg, end;
		if (J.up_strand == '+') { end = J.up_pos; beg = end - max_ins; }
		else if (J.up_strand == '-') { beg = J.up_pos - 1 - 5; end = J.up_pos - 1 + max_ins; }
		else continue; // the reference loops forever here (getsv.cpp:1054-1058 `continue` without ++it)
		if (beg <= 0) beg = 1;
		if ((unsigned)end > chr_length) end = (int)chr_length; // int vs unsigned compare, getsv.cpp:1060
		ssv_junction d;
		d.up_tid = tid; d.down_tid = get_tid(J.down_chr);
		d.up_pos = J.up_pos; d.down_pos = J.down_pos; d.beg = beg; d.end = end;
		d.up_strand = (uint8_t)J.up_strand; d.down_strand = (uint8_t)J.down_strand; d.pad[0] = d.pad[1] = 0;
		p->dev_junction_of[(size_t)j] = (int64_t)p->dev_junctions.size();
		p->dev_junctions.push_back(d);
	}

	// ---- GetBreak, getsv.cpp:752-802 ----
	std::map<std::pair<std::string, int>, int> pos2depth;
	std::map<RangeKey, int> range2depth;
	std::vector<RangeKey> jr((size_t)n_junctions * 4, RangeKey());
	for (int64_t j = 0; j < n_junctions; ++j) {
		const ssvh_junction_in &J = junctions[j];
		std::string uc(J.up_chr), dc(J.down_chr);
		pos2depth.insert(std::make_pair(std::make_pair(uc, J.up_pos), 0));
		pos2depth.insert(std::make_pair(std::make_pair(dc, J.down_pos), 0));
		int l;
		if (uc == dc && J.up_strand == J.down_strand) l = std::abs(J.down_pos - 1 - J.up_pos) < flank_length ? std::abs(J.down_pos - 1 - J.up_pos) : flank_length;
		else l = flank_length;
		unsigned uub = (unsigned)(J.up_pos - l + 1), uue = (unsigned)J.up_pos;
		unsigned udb = (unsigned)(J.up_pos + 1), ude = (unsigned)(J.up_pos + l);
		unsigned dub = (unsigned)(J.down_pos - l), due = (unsigned)(J.down_pos - 1);
		unsigned ddb = (unsigned)J.down_pos, dde = (unsigned)(J.down_pos + l - 1);
		RangeKey r[4] = { RangeKey(uc, uub, uue), RangeKey(uc, udb, ude), RangeKey(dc, dub, due), RangeKey(dc, ddb, dde) };
		for (int k = 0; k < 4; ++k) { range2depth.insert(std::make_pair(r[k], 0)); jr[(size_t)j * 4 + (size_t)k] = r[k]; }
	}
	for (int64_t k = 0; k < n_extra_points; ++k) pos2depth.insert(std::make_pair(std::make_pair(std::string(extra_point_chr[k]), extra_point_pos[k]), 0));

	// ---- MergeOverlap, getsv.cpp:804-835 ----
	std::map<std::pair<std::string, int>, int> begin2end;
	if (!range2depth.empty()) {
		std::string chr; unsigned begin = 0, end = 0;
		bool first = true;
		for (auto &kv : range2depth) {
			const std::string &c = std::get<0>(kv.first); unsigned b = std::get<1>(kv.first), e = std::get<2>(kv.first);
			if (first) { chr = c; begin = b; end = e; first = false; continue; }
			if (chr == c && begin <= b && end + 1 >= b) { if (e > end) end = e; }
			else { begin2end.insert(std::make_pair(std::make_pair(chr, (int)begin), (int)end)); chr = c; begin = b; end = e; }
		}
		begin2end.insert(std::make_pair(std::make_pair(chr, (int)begin), (int)end));
	}
	// NB: with an empty junction list the reference still inserts one (chr="", 0) -> garbage entry; it matches nothing.

	// per contig: merged entries ordered by int begin (the order upper_bound sees, bam2depth.cpp:82)
	std::map<std::string, std::vector<std::pair<int, int>>> entries;
	for (auto &kv : begin2end) entries[kv.first.first].push_back(std::make_pair(kv.first.second, kv.second));

	// M(c): the entry a covered column (chr, c) falls in, bam2depth.cpp:82-85; returns false when the column is skipped
	auto owner = [&](const std::vector<std::pair<int, int>> &ev, int c, int &B, int &E) -> bool {
		auto it = std::upper_bound(ev.begin(), ev.end(), std::make_pair(c, INT_MAX));
		if (it == ev.begin()) return false;
		--it; B = it->first; E = it->second;
		return c <= E;
	};
	(void)owner;

	// the first range in map order: a column whose probe ChrRange(chr, c+1, c+1) sorts before it is skipped
	// entirely (`continue` at bam2depth.cpp:102), pos2depth update included
	const bool have_ranges = !range2depth.empty();
	RangeKey first_range;
	if (have_ranges) first_range = range2depth.begin()->first;
	auto probe_ok = [&](const std::string &chr, int c) -> bool {
		if (!have_ranges) return false;
		RangeKey probe(chr, (unsigned)(c + 1), (unsigned)(c + 1));
		return !(probe < first_range); // upper_bound(probe) != begin()  <=>  first_range <= probe
	};

	// ---- reference ranges -> device pieces ----
	p->ref_ranges.reserve(range2depth.size());
	for (auto &kv : range2depth) p->ref_ranges.push_back(kv.first);
	p->range_pieces.assign(p->ref_ranges.size(), std::vector<int64_t>());
	std::vector<ssv_interval> need; // every interval whose depth the device must know
	for (size_t r = 0; r < p->ref_ranges.size(); ++r) {
		const std::string &chr = std::get<0>(p->ref_ranges[r]);
		const unsigned b = std::get<1>(p->ref_ranges[r]), e = std::get<2>(p->ref_ranges[r]);
		int tid = get_tid(chr.c_str());
		if (tid < 0) continue; // no column of that contig is ever visited
		// columns c (positive ints) with: (b <= c or (b == c+1 and e <= c+1)) and c <= e   [bam2depth.cpp:101-122]
		if (b > (unsigned)INT_MAX) continue; // wrapped begin: no positive c qualifies -> stays 0
		int64_t lo = (int64_t)b, hi = (int64_t)std::min<unsigned>(e, (unsigned)INT_MAX - 1);
		if ((int64_t)e <= (int64_t)b) lo = (int64_t)b - 1; // single-column / empty range also takes column b-1
		if (lo < 1) lo = 1;
		if (hi < lo) continue;
		auto eit = entries.find(chr);
		if (eit == entries.end()) continue;
		const std::vector<std::pair<int, int>> &ev = eit->second;
		// split [lo,hi] by owning entry
		int64_t c = lo;
		while (c <= hi) {
			auto it = std::upper_bound(ev.begin(), ev.end(), std::make_pair((int)c, INT_MAX));
			int64_t next_begin = it == ev.end() ? (int64_t)INT_MAX : (int64_t)it->first;
			if (it == ev.begin()) { c = next_begin; continue; } // columns before the first entry are skipped
			--it;
			const int B = it->first, E = it->second;
			int64_t seg_hi = std::min<int64_t>(hi, next_begin - 1);
			int64_t ok_hi = std::min<int64_t>(seg_hi, (int64_t)E);     // c <= E
			if (ok_hi >= c && (unsigned)B <= b) {                      // r.begin >= (unsigned)M.begin
				// the probe rule can only fail on the lexicographically first contig; check the ends
				int64_t a = c;
				while (a <= ok_hi && !probe_ok(chr, (int)a)) ++a;
				if (a <= ok_hi) {
					ssv_interval iv; iv.tid = tid; iv.beg = (int32_t)a; iv.end = (int32_t)ok_hi;
					p->range_pieces[r].push_back((int64_t)p->ranges.size());
					p->ranges.push_back(iv); need.push_back(iv);
				}
			}
			c = seg_hi + 1;
		}
	}
	p->flank_of.assign((size_t)n_junctions * 4, -1);
	for (size_t k = 0; k < jr.size(); ++k) {
		auto it = std::lower_bound(p->ref_ranges.begin(), p->ref_ranges.end(), jr[k]);
		p->flank_of[k] = (int64_t)(it - p->ref_ranges.begin());
	}

	// ---- points ----
	auto add_point = [&](const std::string &chr, int c) -> int64_t {
		int tid = get_tid(chr.c_str());
		if (tid < 0 || c < 1) return -1;
		auto eit = entries.find(chr);
		if (eit == entries.end()) return -1;
		const std::vector<std::pair<int, int>> &ev = eit->second;
		auto it = std::upper_bound(ev.begin(), ev.end(), std::make_pair(c, INT_MAX));
		if (it == ev.begin()) return -1;
		--it;
		if (c > it->second) return -1;
		if (!probe_ok(chr, c)) return -1;
		ssv_interval iv; iv.tid = tid; iv.beg = c; iv.end = c;
		p->points.push_back(iv); need.push_back(iv);
		return (int64_t)p->points.size() - 1;
	};
	p->up_point.assign((size_t)n_junctions, -1); p->down_point.assign((size_t)n_junctions, -1);
	for (int64_t j = 0; j < n_junctions; ++j) {
		p->up_point[(size_t)j] = add_point(junctions[j].up_chr, junctions[j].up_pos);
		p->down_point[(size_t)j] = add_point(junctions[j].down_chr, junctions[j].down_pos);
	}
	p->extra_point.assign((size_t)n_extra_points, -1);
	for (int64_t k = 0; k < n_extra_points; ++k) p->extra_point[(size_t)k] = add_point(extra_point_chr[k], extra_point_pos[k]);

	// ---- device windows = union of everything that is queried, sorted and disjoint ----
	std::sort(need.begin(), need.end(), [](const ssv_interval &a, const ssv_interval &b) { return a.tid != b.tid ? a.tid < b.tid : (a.beg != b.beg ? a.beg < b.beg : a.end < b.end); });
	for (const ssv_interval &iv : need) {
		if (!p->windows.empty() && p->windows.back().tid == iv.tid && (int64_t)iv.beg <= (int64_t)p->windows.back().end + 1) {
			if (iv.end > p->windows.back().end) p->windows.back().end = iv.end;
		} else p->windows.push_back(iv);
	}
	*out = p;
	return 0;
}

void ssvh_plan_destroy(ssvh_plan *p) { delete p; }

const ssv_junction *ssvh_plan_junctions(const ssvh_plan *p, int64_t *n) { *n = (int64_t)p->dev_junctions.size(); return p->dev_junctions.data(); }
const ssv_interval *ssvh_plan_windows(const ssvh_plan *p, int64_t *n) { *n = (int64_t)p->windows.size(); return p->windows.data(); }
const ssv_interval *ssvh_plan_ranges(const ssvh_plan *p, int64_t *n) { *n = (int64_t)p->ranges.size(); return p->ranges.data(); }
const ssv_interval *ssvh_plan_points(const ssvh_plan *p, int64_t *n) { *n = (int64_t)p->points.size(); return p->points.data(); }

int ssvh_plan_fold(const ssvh_plan *p, const int32_t *counts, const int32_t *prev_counts,
                   const uint64_t *range_sum, const int32_t *point_depth,
                   int32_t *abnormal, int32_t *up_depth, int32_t *down_depth,
                   uint64_t *flank, uint32_t *flank_len, int32_t *extra_point_depth)
{
	for (int64_t j = 0; j < p->n_junctions; ++j) {
		if (abnormal) {
			int64_t d = p->dev_junction_of[(size_t)j];
			abnormal[j] = d >= 0 && counts ? counts[d] : (prev_counts ? prev_counts[j] : 0);
		}
		if (up_depth) up_depth[j] = p->up_point[(size_t)j] >= 0 ? point_depth[p->up_point[(size_t)j]] : 0;
		if (down_depth) down_depth[j] = p->down_point[(size_t)j] >= 0 ? point_depth[p->down_point[(size_t)j]] : 0;
		if (flank) for (int k = 0; k < 4; ++k) {
			int64_t r = p->flank_of[(size_t)j * 4 + (size_t)k];
			uint64_t s = 0;
			for (int64_t piece : p->range_pieces[(size_t)r]) s += range_sum[piece];
			flank[j * 4 + k] = s;
			if (flank_len) flank_len[j * 4 + k] = std::get<2>(p->ref_ranges[(size_t)r]) - std::get<1>(p->ref_ranges[(size_t)r]) + 1u;
		}
	}
	if (extra_point_depth) for (size_t k = 0; k < p->extra_point.size(); ++k) extra_point_depth[k] = p->extra_point[k] >= 0 ? point_depth[p->extra_point[k]] : 0;
	return 0;
}

} // extern "C"
