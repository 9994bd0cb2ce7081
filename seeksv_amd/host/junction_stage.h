// junction_stage.h - host stage of `seeksv getsv` between getclip's cluster table and the BAM passes: join the clip clusters
// (clip.gz) with the re-alignments of their clipped sequences (clip.bam) into junctions, then merge neighbouring junctions.
// Small, string-heavy, order-dependent work: stays on the host.
#pragma once

#include <map>
#include <string>
#include <utility>
#include <vector>

namespace seeksv {

typedef std::vector<std::pair<int, char>> CigarVec;

struct Junction { // getsv.h:149-227: ordered by (up_chr, down_chr, up_strand, down_strand, up_pos, down_pos), chromosomes as strings
	std::string up_chr; int up_pos; char up_strand; std::string down_chr; int down_pos; char down_strand;
	bool operator<(const Junction &o) const
	{
		if (up_chr != o.up_chr) return up_chr < o.up_chr;
		if (down_chr != o.down_chr) return down_chr < o.down_chr;
		if (up_strand != o.up_strand) return up_strand < o.up_strand;
		if (down_strand != o.down_strand) return down_strand < o.down_strand;
		if (up_pos != o.up_pos) return up_pos < o.up_pos;
		return down_pos < o.down_pos;
	}
};

struct SeqInfo { std::string seq; CigarVec cigar_vec; int left_clipped = 0, right_clipped = 0, support = 0, uniq = 0; }; // getsv.h:48-69
struct OtherInfo { SeqInfo up, down; int microhomology = 0, abnormal = 0; };                                             // getsv.h:89-108
typedef std::multimap<Junction, OtherInfo> JunctionMap;

double match_end_first(const std::string &a, const std::string &b);   // CompareStringEndFirst, clip_reads.cpp:194 (NaN for an empty string)
double match_begin_first(const std::string &a, const std::string &b); // CompareStringBeginFirst, clip_reads.cpp:207
std::string slurp_gz(const std::string &path, std::string &out);      // whole file (gzip or plain) into out; "" or an error text
CigarVec parse_cigar(const std::string &cigar);  // ChangeCigarType, getsv.cpp:433
void reverse_complement(std::string &seq);       // GetReverseComplementSeq, clip_reads.cpp:414

// InputSoftInfoStoreBreakpoint (getsv.h:423-541) + GetAlignInfo (getsv.cpp:25) + GetJunction (getsv.cpp:1705).  Returns "" or an error text.
std::string assemble_junctions(const std::string &clipfile, const std::string &clip_bam, JunctionMap &junction2other);
std::string assemble_junctions_text(const std::vector<std::string> &clip_rows, const std::string &clip_bam, JunctionMap &junction2other); // the rows of clip.gz in memory, in pieces of whole rows
// ... and the re-alignments too (`seeksv run`: the aligner step has just made them): what ssvh_bam_next_record would hand out for clip.bam, record by
// record in its order - only the fields GetAlignInfo looks at (getsv.cpp:25-71)
struct AlnRecords {
	int64_t n = 0;
	const int32_t *tid = nullptr, *pos = nullptr;
	const uint16_t *flag = nullptr, *n_cigar = nullptr;
	const uint8_t *mapq = nullptr;
	const uint32_t *cigar_off = nullptr, *cigar = nullptr;
	const char *const *qname = nullptr;
	std::vector<std::string> target_names;
};
std::string assemble_junctions_records(const std::vector<std::string> &clip_rows, const AlnRecords &aln, JunctionMap &junction2other);
// MergeJunction, getsv.cpp:1325-1482
void merge_junctions(JunctionMap &junction2other, int search_length);

} // namespace seeksv
