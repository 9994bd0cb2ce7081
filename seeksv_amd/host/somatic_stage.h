// somatic_stage.h - host stage of `seeksv somatic`: for every junction of the tumor's SV table, look the two breakends
// up among the NORMAL sample's soft-clip clusters (clip.gz of `getclip` on the normal BAM) and decide which junctions need the
// discordant-pair tally on the normal BAM.  The tally itself runs on the GPU (the same kernels as getsv's), once for all rows.
// Small, string-heavy work: stays on the host.  Reference: somatic.h:40-70 (ReadsClipReads), somatic.cpp:14-427.
#pragma once

#include <map>
#include <string>
#include <utility>
#include <vector>

namespace seeksv {

struct NormalCluster { std::string seq_left, seq_right; int support = 0; }; // the part of ReadsInfo (clip_reads.h:44-84) the look-ups read
typedef std::multimap<std::pair<std::string, int>, NormalCluster> ClusterMap;   // (contig, pos) -> clusters in file order

// ReadsClipReads, somatic.h:40-70: rows whose clipped sequence is shorter than min_len_of_clipped_seq are dropped.
// Returns "" or the reference's error text.
std::string load_normal_clusters(const std::string &clip_file, int min_len_of_clipped_seq, ClusterMap &clip3, ClusterMap &clip5, std::string &warnings);

// Compare, clip_reads.cpp:333-370: seq2's first 10 bases are searched in seq4; returns the hit position or -1
int anchored_compare(const std::string &seq1, const std::string &seq2, const std::string &seq3, const std::string &seq4, double match_rate);

struct SomaticRow {
	enum Kind { HEADER, JUNCTION, MESSAGE } kind = JUNCTION;
	std::string text;       // HEADER: the line to print; MESSAGE: the stderr text (no output row); JUNCTION: the 23 re-printed columns
	std::string up_chr, down_chr; int up_pos = 0, down_pos = 0; char up_strand = 0, down_strand = 0;
	int normal_left_reads = 0, normal_right_reads = 0;
	bool tally = false;     // FindDiscordantReadPairs is called for this row (somatic.cpp:111 calls it even when mean == 0)
};

// The row loop of ReadTumorFileAndOutputSomaticInfo (somatic.cpp:58-427) without the BAM access.  Returns "" or an error text.
std::string scan_tumor_table(const std::string &tumor_sv_file, const ClusterMap &clip3, const ClusterMap &clip5, int offset, double min_map_rate,
                             int mean_insert_size, std::vector<SomaticRow> &rows);

} // namespace seeksv
