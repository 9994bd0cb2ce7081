// somatic_stage.h - host stage of `seeksv somatic`: for every junction of the tumor's SV table, look the two breakends
// up among the NORMAL sample's soft-clip clusters (clip.gz of `getclip` on the normal BAM) and decide which junctions need the
// discordant-pair tally on the normal BAM.  The tally itself runs on the GPU (the same kernels as getsv's), once for all rows.
// Small, string-heavy work: stays on the host.  Reference: somatic.h:40-70 (ReadsClipReads), somatic.cpp:14-427.
#pragma once

#include <deque>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "junction_stage.h"

namespace seeksv {

// The normal sample's clusters of one side, the part of ReadsInfo (clip_reads.h:44-84) the look-ups read.  The reference keeps them in a
// std::multimap<pair<string, int>, ReadsInfo> (somatic.h:40-70): ordered by (contig name as a string, position), equal keys in file order.  Here:
// the rows stay views into clip.gz's inflated text (parsed by all threads, junction_stage.h), and a side is an array of (key, row) in that very
// order - contig names replaced by their rank among the file's names, which orders like the strings - searched by bisection.  A 30x normal has
// 5.5 M rows: the multimap of strings took 3 us a row on one thread, this takes the parse (all threads) + one pass over the rows.
struct NormalCluster { Str seq_left, seq_right; int support = 0; };
class ClusterIndex {
public:
	// the clusters of (chr, pos) in file order: [first, last)
	std::pair<size_t, size_t> equal_range(const std::string &chr, int pos) const;
	// the first cluster at or behind (chr, lo) in key order; walk while same_contig(i, chr) && pos(i) <= hi
	size_t lower_bound(const std::string &chr, int lo) const;
	size_t size() const { return key_.size(); }
	bool on_contig(size_t i, const std::string &chr) const;
	int pos(size_t i) const { return (int)(uint32_t)(key_[i] & 0xffffffffu) - (int)0x80000000; }
	const NormalCluster &at(size_t i) const { return val_[i]; }
private:
	friend std::string load_normal_clusters(const std::string &, int, struct NormalClusters &, std::string &);
	std::vector<uint64_t> key_;      // contig rank << 32 | (pos + 2^31), ascending; equal keys in file order
	std::vector<NormalCluster> val_;
	const std::vector<std::string> *names_ = nullptr; // sorted contig names; rank = index
	int64_t rank_of(const std::string &chr) const;    // -1: no such contig in the file
};
struct NormalClusters {
	ClusterIndex clip3, clip5;
	std::vector<std::string> names;          // the file's contig names, sorted
	std::unique_ptr<char[]> text;            // the inflated rows the views point into ...
	std::string text_s;                      // ... or, read the ordinary way, here
	std::deque<std::string> store;           // ... or (stream-loop fallback) the extracted fields
	size_t rows = 0;
};

// ReadsClipReads, somatic.h:40-70: rows whose clipped sequence is shorter than min_len_of_clipped_seq are dropped.
// Returns "" or the reference's error text.
std::string load_normal_clusters(const std::string &clip_file, int min_len_of_clipped_seq, NormalClusters &out, std::string &warnings);

// Compare, clip_reads.cpp:333-370: seq2's first 10 bases are searched in seq4; returns the hit position or -1
int anchored_compare(const std::string &seq1, const std::string &seq2, const std::string &seq3, const std::string &seq4, double match_rate);

struct SomaticRow {
	enum Kind { HEADER, JUNCTION, MESSAGE } kind = JUNCTION;
	std::string text;       // HEADER: the line to print; MESSAGE: the stderr text (no output row); JUNCTION: the 23 re-printed columns
	std::string up_chr, down_chr; int up_pos = 0, down_pos = 0; char up_strand = 0, down_strand = 0;
	int normal_left_reads = 0, normal_right_reads = 0;
	bool tally = false;     // FindDiscordantReadPairs is called for this row (somatic.cpp:111 calls it even when mean == 0)
	int microhomology = 0, up_reads = 0, down_reads = 0; std::string up_seq, down_seq; // what the look-ups read of the row
};

// The row loop of ReadTumorFileAndOutputSomaticInfo (somatic.cpp:58-427) without the BAM access, in two halves: what a row asks for (its text, whether it
// wants the discordant tally: needs the insert size only) and the look-ups among the normal's clusters - so that the BAM pass need not wait for clip.gz.
std::string parse_tumor_table(const std::string &tumor_sv_file, int mean_insert_size, std::vector<SomaticRow> &rows); // "" or an error text
void probe_normal_clusters(std::vector<SomaticRow> &rows, const ClusterIndex &clip3, const ClusterIndex &clip5, int offset, double min_map_rate);
// both, one after the other
std::string scan_tumor_table(const std::string &tumor_sv_file, const ClusterIndex &clip3, const ClusterIndex &clip5, int offset, double min_map_rate,
                             int mean_insert_size, std::vector<SomaticRow> &rows);

} // namespace seeksv
