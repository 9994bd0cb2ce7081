// getsv_plan.cpp - the ordered-container bookkeeping the reference wraps around its two BAM passes.
//
// The GPU answers three primitive questions about a BAM: "how many reads pass the discordant
// predicate chain for junction j inside window [beg,end)", "what is the sum of per-column depth
// over [beg,end]" and "what is the depth at column c".  This file turns the reference's view
// (multimap<Junction,...>, GetBreak getsv.cpp:752-802, MergeOverlap getsv.cpp:804-835, the window
// arithmetic of FindDiscordantReadPairs getsv.cpp:1032-1060 and the per-column lookup rules of
// main_depth bam2depth.cpp:82-124) into those primitive queries and folds the answers back, keeping
// the reference's int/unsigned conversions so that wrapped and empty flank windows come out the same.
//
// The reference keys its std::maps by contig NAME (string order).  Here every name is replaced by
// its rank in the sorted set of names, which orders identically, and the maps become sorted vectors.
#include "seeksv_host.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

namespace {

using RangeKey = std::tuple<int32_t, uint32_t, uint32_t>; // (contig name rank, begin, end): ChrRange::operator< is lexicographic (getsv.h:238-256)

struct Entry { int32_t rank; int begin, end; }; // begin2end: (chr, (int)begin) -> (int)end

} // namespace

struct ssvh_plan {
	int64_t n_junctions = 0;
	std::vector<ssv_junction> dev_junctions;          // only junctions whose up_chr is in the header
	std::vector<uint32_t> dev_chr_length;             // target_len of each dev junction's up contig
	std::vector<int64_t> dev_junction_of;             // junction j -> index in dev_junctions or -1
	std::vector<ssv_interval> windows, ranges, points;
	// fold-back tables
	std::vector<int64_t> piece_first;                 // reference range r -> device ranges [piece_first[r], piece_first[r+1])
	std::vector<RangeKey> ref_ranges;                 // sorted unique
	std::vector<int64_t> flank_of;                    // [4*j+k] -> index in ref_ranges
	std::vector<int64_t> up_point, down_point;        // junction -> device point index or -1 (depth stays 0)
	std::vector<int64_t> extra_point;                 // extra point -> device point index or -1
};

extern "C" {

// discordant windows, getsv.cpp:1032-1060: the only part of the plan that depends on the insert-size statistics
int ssvh_plan_update_isize(ssvh_plan *p, int32_t mean, int32_t sd, int32_t times)
{
	const int max_ins = mean + sd * times;
	for (size_t k = 0; k < p->dev_junctions.size(); ++k) {
		ssv_junction &d = p->dev_junctions[k];
		int beg, end;
		if (d.up_strand == '+') { end = d.up_pos; beg = end - max_ins; }
		else { beg = d.up_pos - 1 - 5; end = d.up_pos - 1 + max_ins; }
		if (beg <= 0) beg = 1;
		if ((unsigned)end > p->dev_chr_length[k]) end = (int)p->dev_chr_length[k]; // int vs unsigned compare, getsv.cpp:1060
		// bam_iter_query on an empty interval visits no record (reg2bins returns no bins when beg >= end); this happens for '+'
		// junctions when mean = sd = 0 (somatic.cpp:111 calls the tally without insert-size statistics).  Empty window here too.
		if (beg >= end) beg = end = 0;
		d.beg = beg; d.end = end;
	}
	return 0;
}

int ssvh_plan_create(const ssvh_bam *bam, const ssvh_junction_in *junctions, int64_t n_junctions,
                     const char *const *extra_point_chr, const int32_t *extra_point_pos, int64_t n_extra_points,
                     int32_t mean, int32_t sd, int32_t times, int32_t flank_length, ssvh_plan **out)
{
	ssvh_plan *p = new ssvh_plan();
	p->n_junctions = n_junctions;
	static const bool timing = [] { const char *e = getenv("SSV_TIMING"); return e ? atoi(e) : 0; }() >= 3; // SSV_TIMING=3: the plan's phases
	auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double tl = tnow();
	auto lap = [&](const char *what) { if (timing) { const double t = tnow(); fprintf(stderr, "[plan] %s %.2f ms\n", what, t - tl); tl = t; } };

	// ---- contig names: header tid (StoreSeqName2Tid, cluster.cpp:206-216, first wins) and string rank ----
	std::map<std::string, int> name2tid;
	const int32_t nt = ssvh_bam_n_targets(bam);
	for (int32_t i = 0; i < nt; ++i) name2tid.insert(std::make_pair(std::string(ssvh_bam_target_name(bam, i)), i));
	// distinct names used by junctions / points.  Callers normally share one char* per contig, so names are first
	// de-duplicated by pointer (a handful of distinct pointers) and only those are compared as strings.
	std::map<const char *, int32_t> by_ptr;
	for (int64_t j = 0; j < n_junctions; ++j) { by_ptr.insert(std::make_pair(junctions[j].up_chr, 0)); by_ptr.insert(std::make_pair(junctions[j].down_chr, 0)); }
	for (int64_t k = 0; k < n_extra_points; ++k) by_ptr.insert(std::make_pair(extra_point_chr[k], 0));
	std::map<std::string, int32_t> rank_of;
	for (auto &kv : by_ptr) rank_of.insert(std::make_pair(std::string(kv.first), 0));
	std::vector<int> tid_of_rank;
	{
		int32_t r = 0;
		for (auto &kv : rank_of) {
			kv.second = r++;
			auto it = name2tid.find(kv.first);
			tid_of_rank.push_back(it == name2tid.end() ? -1 : it->second);
		}
	}
	for (auto &kv : by_ptr) kv.second = rank_of.find(std::string(kv.first))->second;
	// per junction ranks; consecutive junctions usually repeat the previous pointer
	std::vector<int32_t> urank((size_t)n_junctions), drank((size_t)n_junctions);
	{
		const char *lu = nullptr, *ld = nullptr;
		int32_t ru = 0, rd = 0;
		for (int64_t j = 0; j < n_junctions; ++j) {
			if (junctions[j].up_chr != lu) { lu = junctions[j].up_chr; ru = by_ptr.find(lu)->second; }
			if (junctions[j].down_chr != ld) { ld = junctions[j].down_chr; rd = by_ptr.find(ld)->second; }
			urank[(size_t)j] = ru; drank[(size_t)j] = rd;
		}
	}

	// ---- junctions the device will tally (window arithmetic in ssvh_plan_update_isize) ----
	p->dev_junction_of.assign((size_t)n_junctions, -1);
	p->dev_junctions.reserve((size_t)n_junctions);
	p->dev_chr_length.reserve((size_t)n_junctions);
	for (int64_t j = 0; j < n_junctions; ++j) {
		const ssvh_junction_in &J = junctions[j];
		int tid = tid_of_rank[(size_t)urank[(size_t)j]];
		if (tid == -1) continue; // keeps its previous count, getsv.cpp:1043
		if (J.up_strand != '+' && J.up_strand != '-') continue; // the reference loops forever here (getsv.cpp:1054-1058 `continue` without ++it)
		ssv_junction d;
		d.up_tid = tid; d.down_tid = tid_of_rank[(size_t)drank[(size_t)j]];
		d.up_pos = J.up_pos; d.down_pos = J.down_pos; d.beg = 0; d.end = 0;
		d.up_strand = (uint8_t)J.up_strand; d.down_strand = (uint8_t)J.down_strand; d.pad[0] = d.pad[1] = 0;
		p->dev_junction_of[(size_t)j] = (int64_t)p->dev_junctions.size();
		p->dev_junctions.push_back(d);
		p->dev_chr_length.push_back((unsigned)ssvh_bam_target_len(bam, tid));
	}
	lap("names + device junctions");
	ssvh_plan_update_isize(p, mean, sd, times);
	lap("update_isize");

	// ---- GetBreak, getsv.cpp:752-802: the four flank windows of every junction (unsigned arithmetic) ----
	std::vector<RangeKey> jr((size_t)n_junctions * 4);
	for (int64_t j = 0; j < n_junctions; ++j) {
		const ssvh_junction_in &J = junctions[j];
		const int32_t ur = urank[(size_t)j], dr = drank[(size_t)j];
		int l;
		if (ur == dr && J.up_strand == J.down_strand) l = std::abs(J.down_pos - 1 - J.up_pos) < flank_length ? std::abs(J.down_pos - 1 - J.up_pos) : flank_length;
		else l = flank_length;
		jr[(size_t)j * 4 + 0] = RangeKey(ur, (unsigned)(J.up_pos - l + 1), (unsigned)J.up_pos);
		jr[(size_t)j * 4 + 1] = RangeKey(ur, (unsigned)(J.up_pos + 1), (unsigned)(J.up_pos + l));
		jr[(size_t)j * 4 + 2] = RangeKey(dr, (unsigned)(J.down_pos - l), (unsigned)(J.down_pos - 1));
		jr[(size_t)j * 4 + 3] = RangeKey(dr, (unsigned)J.down_pos, (unsigned)(J.down_pos + l - 1));
	}
	// range2depth: the map's keys = the sorted unique ranges.  Sorted as packed words with their origin beside them, so that the position of
	// every junction's four windows among the unique ranges (flank_of) falls out of the same pass instead of 4 J binary searches over tuples.
	struct Packed { uint64_t a; uint32_t e, idx; }; // a = rank << 32 | begin (ranks are >= 0): (a, e) orders like (rank, begin, end)
	std::vector<Packed> pk(jr.size());
	for (size_t k = 0; k < jr.size(); ++k) pk[k] = Packed{((uint64_t)(uint32_t)std::get<0>(jr[k]) << 32) | std::get<1>(jr[k]), std::get<2>(jr[k]), (uint32_t)k};
	std::sort(pk.begin(), pk.end(), [](const Packed &x, const Packed &y) { return x.a != y.a ? x.a < y.a : x.e < y.e; });
	p->flank_of.assign((size_t)n_junctions * 4, -1);
	p->ref_ranges.clear(); p->ref_ranges.reserve(pk.size());
	for (size_t k = 0; k < pk.size(); ++k) {
		if (k == 0 || pk[k].a != pk[k - 1].a || pk[k].e != pk[k - 1].e) p->ref_ranges.push_back(jr[pk[k].idx]);
		p->flank_of[pk[k].idx] = (int64_t)p->ref_ranges.size() - 1;
	}
	const std::vector<RangeKey> &rr = p->ref_ranges;
	lap("flank ranges sorted");

	// ---- MergeOverlap, getsv.cpp:804-835 -> begin2end (map::insert: the first entry with a key wins) ----
	std::vector<Entry> entries;
	if (!rr.empty()) {
		int32_t chr = std::get<0>(rr[0]); unsigned begin = std::get<1>(rr[0]), end = std::get<2>(rr[0]);
		for (size_t k = 1; k < rr.size(); ++k) {
			const int32_t c = std::get<0>(rr[k]); const unsigned b = std::get<1>(rr[k]), e = std::get<2>(rr[k]);
			if (chr == c && begin <= b && end + 1 >= b) { if (e > end) end = e; }
			else { entries.push_back(Entry{chr, (int)begin, (int)end}); chr = c; begin = b; end = e; }
		}
		entries.push_back(Entry{chr, (int)begin, (int)end});
	}
	std::stable_sort(entries.begin(), entries.end(), [](const Entry &a, const Entry &b) { return a.rank != b.rank ? a.rank < b.rank : a.begin < b.begin; });
	entries.erase(std::unique(entries.begin(), entries.end(), [](const Entry &a, const Entry &b) { return a.rank == b.rank && a.begin == b.begin; }), entries.end());
	// per contig slice of the entries
	const size_t nrank = rank_of.size();
	std::vector<size_t> ent_first(nrank + 1, 0);
	for (const Entry &e : entries) ent_first[(size_t)e.rank + 1]++;
	for (size_t r = 0; r < nrank; ++r) ent_first[r + 1] += ent_first[r];

	// the first range in map order: a column whose probe ChrRange(chr, c+1, c+1) sorts before it is skipped entirely
	// (`continue` at bam2depth.cpp:102), pos2depth update included
	const bool have_ranges = !rr.empty();
	const RangeKey first_range = have_ranges ? rr[0] : RangeKey();
	auto probe_ok = [&](int32_t rank, int c) -> bool {
		if (!have_ranges) return false;
		return !(RangeKey(rank, (unsigned)(c + 1), (unsigned)(c + 1)) < first_range); // upper_bound(probe) != begin()
	};
	// the entry that owns column c of a contig (bam2depth.cpp:82-85): greatest (int)begin <= c; -1 if none
	auto owner = [&](int32_t rank, int64_t c, int64_t *next_begin) -> int64_t {
		size_t lo = ent_first[(size_t)rank], hi = ent_first[(size_t)rank + 1];
		size_t a = lo, z = hi;
		while (a < z) { size_t m = (a + z) / 2; if ((int64_t)entries[m].begin <= c) a = m + 1; else z = m; }
		*next_begin = a < hi ? (int64_t)entries[a].begin : (int64_t)INT_MAX;
		return a == lo ? -1 : (int64_t)a - 1;
	};

	lap("entries");
	// ---- reference ranges -> device pieces ----
	std::vector<ssv_interval> need; // every interval whose depth the device must know
	need.reserve(rr.size() + 2 * (size_t)n_junctions + (size_t)n_extra_points);
	p->ranges.reserve(rr.size() + 16);
	p->points.reserve(2 * (size_t)n_junctions + (size_t)n_extra_points);
	p->piece_first.assign(rr.size() + 1, 0);
	for (size_t r = 0; r < rr.size(); ++r) {
		p->piece_first[r] = (int64_t)p->ranges.size();
		const int32_t rank = std::get<0>(rr[r]);
		const unsigned b = std::get<1>(rr[r]), e = std::get<2>(rr[r]);
		const int tid = tid_of_rank[(size_t)rank];
		if (tid < 0) continue;               // no column of that contig is ever visited
		if (b > (unsigned)INT_MAX) continue; // wrapped begin: no positive column qualifies -> stays 0
		// columns c (positive ints) with: (b <= c or (b == c+1 and e <= c+1)) and c <= e   [bam2depth.cpp:101-122]
		int64_t lo = (int64_t)b, hi = (int64_t)std::min<unsigned>(e, (unsigned)INT_MAX - 1);
		if ((int64_t)e <= (int64_t)b) lo = (int64_t)b - 1; // single-column / empty range also takes column b-1
		if (lo < 1) lo = 1;
		int64_t c = lo;
		while (c <= hi) { // split [lo,hi] by owning entry
			int64_t next_begin;
			int64_t oi = owner(rank, c, &next_begin);
			if (oi < 0) { c = next_begin; continue; } // columns before the first entry are skipped
			const int B = entries[(size_t)oi].begin, E = entries[(size_t)oi].end;
			int64_t seg_hi = std::min<int64_t>(hi, next_begin - 1);
			int64_t ok_hi = std::min<int64_t>(seg_hi, (int64_t)E);     // c <= E
			if (ok_hi >= c && (unsigned)B <= b) {                      // r.begin >= (unsigned)M.begin
				int64_t a = c;
				while (a <= ok_hi && !probe_ok(rank, (int)a)) ++a;       // can only fail on the lexicographically first contig
				if (a <= ok_hi) {
					ssv_interval iv; iv.tid = tid; iv.beg = (int32_t)a; iv.end = (int32_t)ok_hi;
					p->ranges.push_back(iv); need.push_back(iv);
				}
			}
			c = seg_hi + 1;
		}
	}
	p->piece_first[rr.size()] = (int64_t)p->ranges.size();
	lap("pieces");

	lap("flank_of");
	// ---- points (pos2depth keys): depth is recorded only for columns inside their owning entry ----
	auto add_point = [&](int32_t rank, int c) -> int64_t {
		const int tid = tid_of_rank[(size_t)rank];
		if (tid < 0 || c < 1) return -1;
		int64_t nb;
		int64_t oi = owner(rank, c, &nb);
		if (oi < 0 || c > entries[(size_t)oi].end) return -1;
		if (!probe_ok(rank, c)) return -1;
		ssv_interval iv; iv.tid = tid; iv.beg = c; iv.end = c;
		p->points.push_back(iv); need.push_back(iv);
		return (int64_t)p->points.size() - 1;
	};
	p->up_point.assign((size_t)n_junctions, -1); p->down_point.assign((size_t)n_junctions, -1);
	for (int64_t j = 0; j < n_junctions; ++j) {
		p->up_point[(size_t)j] = add_point(urank[(size_t)j], junctions[j].up_pos);
		p->down_point[(size_t)j] = add_point(drank[(size_t)j], junctions[j].down_pos);
	}
	p->extra_point.assign((size_t)n_extra_points, -1);
	for (int64_t k = 0; k < n_extra_points; ++k) p->extra_point[(size_t)k] = add_point(by_ptr.find(extra_point_chr[k])->second, extra_point_pos[k]);

	lap("points");
	// ---- device windows = union of everything that is queried, sorted and disjoint ----
	{ // sorted by (tid, beg, end) as one 64-bit word + end (tid >= 0, beg >= 1 here)
		struct NeedKey { uint64_t a; int32_t end; };
		std::vector<NeedKey> nk(need.size());
		for (size_t k = 0; k < need.size(); ++k) nk[k] = NeedKey{((uint64_t)(uint32_t)need[k].tid << 32) | (uint32_t)need[k].beg, need[k].end};
		std::sort(nk.begin(), nk.end(), [](const NeedKey &x, const NeedKey &y) { return x.a != y.a ? x.a < y.a : x.end < y.end; });
		for (size_t k = 0; k < need.size(); ++k) { need[k].tid = (int32_t)(nk[k].a >> 32); need[k].beg = (int32_t)(uint32_t)nk[k].a; need[k].end = nk[k].end; }
	}
	for (const ssv_interval &iv : need) {
		if (!p->windows.empty() && p->windows.back().tid == iv.tid && (int64_t)iv.beg <= (int64_t)p->windows.back().end + 1) {
			if (iv.end > p->windows.back().end) p->windows.back().end = iv.end;
		} else p->windows.push_back(iv);
	}
	lap("windows");
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
			for (int64_t piece = p->piece_first[(size_t)r]; piece < p->piece_first[(size_t)r + 1]; ++piece) s += range_sum[piece];
			flank[j * 4 + k] = s;
			if (flank_len) flank_len[j * 4 + k] = std::get<2>(p->ref_ranges[(size_t)r]) - std::get<1>(p->ref_ranges[(size_t)r]) + 1u;
		}
	}
	if (extra_point_depth) for (size_t k = 0; k < p->extra_point.size(); ++k) extra_point_depth[k] = p->extra_point[k] >= 0 ? point_depth[p->extra_point[k]] : 0;
	return 0;
}

} // extern "C"
