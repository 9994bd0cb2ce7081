// junction_stage.h - host stage of `seeksv getsv` between getclip's cluster table and the BAM passes: join the clip clusters
// (clip.gz) with the re-alignments of their clipped sequences (clip.bam) into junctions, then merge neighbouring junctions.
// Small, string-heavy, order-dependent work: stays on the host.
#pragma once

#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
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

// A piece of text that lives elsewhere (the rows of clip.gz in memory).
struct Str {
	const char *p = nullptr; size_t n = 0;
	std::string str() const { return std::string(p, n); }
	bool operator==(const Str &o) const { return n == o.n && (n == 0 || memcmp(p, o.p, n) == 0); }
	bool equals(const char *z) const { return strncmp(z, p, n) == 0 && z[n] == '\0'; }
};
// One row of clip.gz as views into the text: aligned part first, whatever the side (getsv.h:460-461; somatic.h:45-52 reads the same nine fields).
struct ClipRow {
	Str chr; int pos = 0; char side = 0; Str cigar;
	Str aligned_seq, clipped_seq, clipped_qual; int support = 0;
	uint64_t h = 0; // 64-bit hash of clipped_seq (the join's first compare)
};
struct TextView { // (just enough of std::string for the parser)
	const char *p; size_t n;
	size_t size() const { return n; }
	const char *data() const { return p; }
	size_t find(char c, size_t from) const { const void *q = from < n ? memchr(p + from, c, n - from) : nullptr; return q ? (size_t)(static_cast<const char *>(q) - p) : std::string::npos; }
};
// A .gz file of independent gzip members (what ssvh_gz_append writes) inflated by all threads straight into place; false: not such a file (read it with slurp_gz).
bool slurp_gz_members(const std::string &path, std::unique_ptr<char[]> &buf, size_t &buf_len);
// The rows of clip.gz parsed by all threads (texts: pieces of whole rows, in file order); false: text the reference's `fin >> ...` loop would not split the
// same way (short lines, non-numeric fields) - the caller then runs that very stream loop.
bool parse_rows_parallel(const std::vector<TextView> &texts, std::vector<ClipRow> &rows);

// InputSoftInfoStoreBreakpoint (getsv.h:423-541) + GetAlignInfo (getsv.cpp:25) + GetJunction (getsv.cpp:1705).  Returns "" or an error text.
std::string assemble_junctions(const std::string &clipfile, const std::string &clip_bam, JunctionMap &junction2other);
std::string assemble_junctions_text(const std::vector<TextView> &clip_rows, const std::string &clip_bam, JunctionMap &junction2other); // the rows of clip.gz in memory, in pieces of whole rows
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
std::string assemble_junctions_records(const std::vector<TextView> &clip_rows, const AlnRecords &aln, JunctionMap &junction2other);
// ... and the rows as the process that wrote clip.gz kept them (views into the text it wrote, h = clip_text_hash(clipped_seq)): nothing to parse.  Only for
// rows the text parser would split the same way - nine non-empty fields without white space (the caller checks while it writes them).
std::string assemble_junctions_rows(const std::vector<const std::vector<ClipRow> *> &rows, const AlnRecords &aln, JunctionMap &junction2other);
uint64_t clip_text_hash(const char *p, size_t n);
// MergeJunction, getsv.cpp:1325-1482
void merge_junctions(JunctionMap &junction2other, int search_length);

} // namespace seeksv
