// bam_partition.cpp - cutting a coordinate-sorted BAM FILE into N contiguous runs of records, one per GPU.
//
// The reference reaches a region of the BAM through its .bai (seeksv.cpp:272-280 bam_index_load, getsv.cpp:1063-1067 bam_iter_query /
// bam_iter_read).  Here the file itself is partitioned: boundaries are BGZF virtual offsets of record starts, chosen so that the parts hold
// about the same number of inflated bytes.  No index is read: a pass over the block headers (18 bytes each, through a read-only mapping of
// the file) gives every block's position and inflated size; only the blocks around a boundary are inflated.  A record start inside a block is found by speculation (a position from
// which a chain of plausible record headers follows) and - when walking backwards from a known start - verified exactly: the chain must
// land on the known start.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "seeksv_host.h"

namespace {

struct BlockPos { uint64_t coff; uint32_t csize, isize; };

struct BlockFile {
	int fd = -1;
	const uint8_t *map = nullptr; // the whole file, read-only: the block headers are walked through the mapping (two touched pages per
	size_t map_len = 0;           // block, no system call each), and a block's bytes are inflated from where they lie
	std::vector<BlockPos> blocks;
	std::vector<uint64_t> ucum; // inflated bytes before block k
	int32_t n_targets = 0;
	std::vector<int32_t> target_len;
	uint64_t header_len = 0;    // inflated bytes before the first record
	std::string err;
	~BlockFile() { if (map && map_len) munmap(const_cast<uint8_t *>(map), map_len); if (fd >= 0) close(fd); }

	bool open(const char *path)
	{
		fd = ::open(path, O_RDONLY);
		if (fd < 0) { err = std::string("cannot open ") + path; return false; }
		struct stat st;
		if (fstat(fd, &st) != 0) { err = "cannot stat"; return false; }
		map_len = (size_t)st.st_size;
		if (map_len) {
			void *m = mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE, fd, 0);
			if (m == MAP_FAILED) { map_len = 0; err = "cannot map the file"; return false; }
			map = static_cast<const uint8_t *>(m);
		}
		uint64_t at = 0;
		while (at < map_len) {
			const uint8_t *hdr = map + at;
			if (at + 18 > map_len || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) { err = "not a BGZF block"; return false; }
			const unsigned xlen = hdr[10] | (hdr[11] << 8);
			if (at + 12 + xlen > map_len) { err = "truncated BGZF header"; return false; }
			const uint8_t *extra = hdr + 12;
			int bsize = -1;
			for (size_t off = 0; off + 4 <= xlen;) {
				const unsigned slen = extra[off + 2] | (extra[off + 3] << 8);
				if (off + 4 + slen > xlen) break;
				if (extra[off] == 'B' && extra[off + 1] == 'C' && slen == 2) bsize = extra[off + 4] | (extra[off + 5] << 8);
				off += 4 + slen;
			}
			if (bsize < 0 || (size_t)bsize + 1 < 12 + (size_t)xlen + 8) { err = "bad BGZF block"; return false; }
			if (at + (uint64_t)bsize + 1 > map_len) { err = "truncated BGZF block"; return false; }
			uint32_t isize; memcpy(&isize, map + at + (uint64_t)bsize + 1 - 4, 4);
			if (isize > 65536) { err = "BGZF block that claims to inflate to more than 64 KB"; return false; }
			blocks.push_back(BlockPos{at, (uint32_t)bsize + 1, isize});
			at += (uint64_t)bsize + 1;
		}
		ucum.assign(blocks.size() + 1, 0);
		for (size_t k = 0; k < blocks.size(); ++k) ucum[k + 1] = ucum[k] + blocks[k].isize;
		return read_header();
	}

	// inflated bytes of blocks [k0, k1) appended to out
	bool inflate_blocks(size_t k0, size_t k1, std::vector<uint8_t> &out)
	{
		for (size_t k = k0; k < k1 && k < blocks.size(); ++k) {
			const BlockPos &b = blocks[k];
			const uint8_t *c = map + b.coff;
			const unsigned xlen = c[10] | (c[11] << 8);
			const size_t at = out.size();
			out.resize(at + b.isize);
			if (!b.isize) continue;
			z_stream zs;
			memset(&zs, 0, sizeof(zs));
			if (inflateInit2(&zs, -15) != Z_OK) { err = "zlib"; return false; }
			zs.next_in = const_cast<Bytef *>(c + 12 + xlen); zs.avail_in = (uInt)(b.csize - 12 - xlen - 8);
			zs.next_out = out.data() + at; zs.avail_out = b.isize;
			const int rc = inflate(&zs, Z_FINISH);
			inflateEnd(&zs);
			if (rc != Z_STREAM_END || zs.total_out != b.isize) { err = "BGZF inflate failed"; return false; }
		}
		return true;
	}

	bool read_header()
	{
		std::vector<uint8_t> u;
		size_t k = 0;
		auto need = [&](size_t n) { while (u.size() < n && k < blocks.size()) { if (!inflate_blocks(k, k + 1, u)) return false; ++k; } return u.size() >= n; };
		if (!need(12) || memcmp(u.data(), "BAM\1", 4) != 0) { if (err.empty()) err = "not a BAM file"; return false; }
		int32_t l_text; memcpy(&l_text, u.data() + 4, 4);
		if (l_text < 0 || !need(12 + (size_t)l_text)) { if (err.empty()) err = "bad BAM header"; return false; }
		int32_t n_ref; memcpy(&n_ref, u.data() + 8 + l_text, 4);
		size_t o = 12 + (size_t)l_text;
		for (int32_t i = 0; i < n_ref; ++i) {
			if (!need(o + 4)) { if (err.empty()) err = "bad BAM header"; return false; }
			int32_t l_name; memcpy(&l_name, u.data() + o, 4);
			if (l_name <= 0) { err = "bad BAM header"; return false; }
			o += 8 + (size_t)l_name;
			if (!need(o)) { if (err.empty()) err = "bad BAM header"; return false; }
			int32_t l_ref; memcpy(&l_ref, u.data() + o - 4, 4);
			target_len.push_back(l_ref);
		}
		if (!need(o)) { if (err.empty()) err = "bad BAM header"; return false; }
		n_targets = n_ref; header_len = o;
		return true;
	}

	// block that holds inflated offset g (the last block whose start is <= g)
	size_t block_of(uint64_t g) const { return (size_t)(std::upper_bound(ucum.begin(), ucum.end(), g) - ucum.begin()) - 1; }
};

// Could a record start at u + o?  The same predicate as the host reader's and the device decoder's (bam_reader.cpp, bamdec_kernels.h):
// a read name of at least one character + NUL (two bytes ahead of a true start the previous record's tail and the low half of block_size
// read as a ~17 MB record with an empty name, about once in 2,000 blocks of equal-sized records), and a position inside its contig.
bool plausible(const uint8_t *u, size_t o, size_t end, int32_t n_targets, const int32_t *target_len)
{
	if (o + 36 > end) return false;
	uint32_t bs; int32_t refid, pos, l_seq, next_ref, next_pos; uint16_t ncig;
	memcpy(&bs, u + o, 4);
	if (bs < 32 || bs > (1u << 28)) return false;
	const uint8_t *r = u + o + 4;
	memcpy(&refid, r, 4); memcpy(&pos, r + 4, 4); memcpy(&ncig, r + 12, 2); memcpy(&l_seq, r + 16, 4); memcpy(&next_ref, r + 20, 4); memcpy(&next_pos, r + 24, 4);
	const size_t l_name = r[8];
	if (refid < -1 || refid >= n_targets || next_ref < -1 || next_ref >= n_targets || pos < -1 || next_pos < -1 || l_seq < 0 || l_name < 2) return false;
	if (refid >= 0 && target_len && pos >= target_len[refid]) return false; // (the mate's position is left alone: this test must never fail a true record)
	if (32 + l_name + 4 * (size_t)ncig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq > (size_t)bs) return false;
	const size_t nul = o + 4 + 32 + l_name - 1;
	return nul >= end || u[nul] == 0;
}

// a record starts at global inflated offset `start`: the inflated stream from there (blocks appended on demand)
struct Window {
	BlockFile &f;
	std::vector<uint8_t> u;
	uint64_t g0 = 0;      // global inflated offset of u[0]
	size_t k_next = 0;    // next block to append
	explicit Window(BlockFile &bf) : f(bf) {}
	bool begin_at_block(size_t k) { u.clear(); g0 = f.ucum[k]; k_next = k; return true; }
	bool need(uint64_t g_end) // make bytes [g0, g_end) available (false at end of file)
	{
		while (g0 + u.size() < g_end) {
			if (k_next >= f.blocks.size()) return false;
			if (!f.inflate_blocks(k_next, k_next + 1, u)) return false;
			++k_next;
		}
		return true;
	}
};

// the first record that starts at or after global inflated offset g: a position from which `want` plausible headers chain (fewer only at
// the end of the file).  Returns false when there is none.
bool first_record_from(BlockFile &f, uint64_t g, int want, uint64_t *out)
{
	const uint64_t total = f.ucum.back();
	if (g < f.header_len) g = f.header_len;
	if (g >= total) return false;
	Window w(f);
	w.begin_at_block(f.block_of(g));
	for (uint64_t cand = g; cand + 36 <= total; ++cand) {
		uint64_t q = cand;
		int k = 0;
		for (; k < want; ++k) {
			if (q == total) break; // the chain ends exactly at the end of the file: good
			if (q + 36 > total) { k = -1; break; }
			w.need(std::min<uint64_t>(total, q + 4 + 36 + 256));
			if (!plausible(w.u.data(), (size_t)(q - w.g0), w.u.size(), f.n_targets, f.target_len.data())) { k = -1; break; }
			uint32_t bs; memcpy(&bs, w.u.data() + (q - w.g0), 4);
			q += 4 + (uint64_t)bs;
			if (q > total) { k = -1; break; }
		}
		if (k >= 0) { *out = cand; return true; }
		if (!f.err.empty()) return false;
	}
	return false;
}

// The first record that starts at or after g, VERIFIED.  A speculated start can be a false one whose chain joins the true records after a
// hop: two bytes ahead of a true start, the previous record's last two bytes and the low half of block_size read as a ~17 MB record, and
// with equal-sized records that hop can land exactly on a later true start - the boundary would sit two bytes off a record, or megabytes
// late.  False starts are rare and unrelated to each other, the file's chain is unique: so the answer must be given by TWO chains that
// were started independently - the local speculation from g and chains started 4, 8, 16 blocks (hundreds of records) earlier; each says
// where its first record at or after g is, and the first position two of them agree on is taken.  A chain from the end of the BAM
// header is the file's chain by definition and needs no second.
bool first_record_verified(BlockFile &f, uint64_t g, uint64_t *out)
{
	const uint64_t total = f.ucum.back();
	if (g < f.header_len) g = f.header_len;
	if (g >= total) return false;
	const size_t kb = f.block_of(g);
	std::vector<uint64_t> votes;
	bool any = false;
	for (size_t back : {(size_t)0, (size_t)4, (size_t)8, (size_t)16, (size_t)32}) {
		uint64_t from = back == 0 ? g : (kb > back ? f.ucum[kb - back] : f.header_len);
		if (from < f.header_len) from = f.header_len;
		uint64_t a = from;
		if (from > f.header_len && !first_record_from(f, from, 8, &a)) { if (!f.err.empty()) return false; continue; }
		any = true;
		// follow the chain from a to its first record at or after g
		Window w(f);
		w.begin_at_block(f.block_of(a));
		uint64_t q = a;
		bool ok = true;
		while (q < g) {
			if (q + 36 > total) { ok = false; break; }
			w.need(std::min<uint64_t>(total, q + 4 + 36 + 256));
			if (!plausible(w.u.data(), (size_t)(q - w.g0), w.u.size(), f.n_targets, f.target_len.data())) { ok = false; break; }
			uint32_t bs; memcpy(&bs, w.u.data() + (q - w.g0), 4);
			q += 4 + (uint64_t)bs;
		}
		if (!f.err.empty()) return false;
		if (!ok || q > total) continue; // a false anchor: its chain broke
		if (from == f.header_len) { if (q >= total) return false; *out = q; return true; }
		for (uint64_t v : votes) if (v == q) { if (q >= total) return false; *out = q; return true; }
		votes.push_back(q);
	}
	if (!any) return false; // nothing that looks like a record at or after g
	f.err = "no two record chains agree on the first record behind a partition boundary";
	return false;
}

bool from_voffset_of(const BlockFile &f, uint64_t coff, uint32_t uoff, uint64_t *g)
{
	auto it = std::lower_bound(f.blocks.begin(), f.blocks.end(), coff, [](const BlockPos &b, uint64_t c) { return b.coff < c; });
	if (it == f.blocks.end() || it->coff != coff || uoff > it->isize) return false;
	*g = f.ucum[(size_t)(it - f.blocks.begin())] + uoff;
	return true;
}

// The .bai next to the file, when there is one (the reference reaches regions through it: seeksv.cpp:272-280 bam_index_load,
// getsv.cpp:1063-1067 bam_iter_query): its LINEAR index holds, for every 16 kb window of every contig, the virtual offset of the first
// record that overlaps the window - record starts that need no speculation.  -> their global inflated offsets, ascending.
bool load_bai_starts(const char *bam_path, const BlockFile &f, std::vector<uint64_t> &starts)
{
	starts.clear();
	FILE *fp = fopen((std::string(bam_path) + ".bai").c_str(), "rb");
	if (!fp) {
		std::string alt(bam_path);
		if (alt.size() > 4 && alt.compare(alt.size() - 4, 4, ".bam") == 0) { alt.replace(alt.size() - 4, 4, ".bai"); fp = fopen(alt.c_str(), "rb"); }
		if (!fp) return false;
	}
	auto rd = [&](void *p, size_t n) { return fread(p, 1, n, fp) == n; };
	char magic[4]; int32_t n_ref = 0;
	bool ok = rd(magic, 4) && memcmp(magic, "BAI\1", 4) == 0 && rd(&n_ref, 4) && n_ref >= 0 && n_ref == f.n_targets;
	for (int32_t r = 0; ok && r < n_ref; ++r) {
		int32_t n_bin = 0;
		ok = rd(&n_bin, 4) && n_bin >= 0;
		for (int32_t b = 0; ok && b < n_bin; ++b) {
			uint32_t bin; int32_t n_chunk = 0;
			ok = rd(&bin, 4) && rd(&n_chunk, 4) && n_chunk >= 0 && fseek(fp, (long)n_chunk * 16, SEEK_CUR) == 0;
		}
		int32_t n_intv = 0;
		ok = ok && rd(&n_intv, 4) && n_intv >= 0;
		std::vector<uint64_t> io((size_t)n_intv);
		ok = ok && (n_intv == 0 || rd(io.data(), (size_t)n_intv * 8));
		for (uint64_t v : io) {
			if (!v) continue; // a window no record overlaps
			uint64_t g;
			if (!from_voffset_of(f, v >> 16, (uint32_t)(v & 0xffff), &g)) { ok = false; break; } // not a block of THIS file: a stale index
			starts.push_back(g);
		}
	}
	fclose(fp);
	if (!ok) { starts.clear(); return false; }
	std::sort(starts.begin(), starts.end());
	starts.erase(std::unique(starts.begin(), starts.end()), starts.end());
	// a stale or foreign index must not cut the file inside a record: every start has to look like one
	return !starts.empty();
}

struct RecHead { uint64_t g; int32_t tid, pos; uint16_t flag; };

// the records that start in [g_lo_hint .. g_known), in order, where g_known is a KNOWN record start: a chain is speculated from the start
// of the block holding g_lo_hint and accepted only if it lands exactly on g_known.  At the first block the chain starts at the header's end.
bool records_before(BlockFile &f, uint64_t g_known, size_t k_from, std::vector<RecHead> &out)
{
	out.clear();
	Window w(f);
	w.begin_at_block(k_from);
	if (!w.need(g_known)) { if (f.err.empty()) f.err = "walk back: short file"; return false; }
	const uint64_t lo = k_from == 0 ? f.header_len : f.ucum[k_from];
	const uint8_t *u = w.u.data();
	const size_t end = (size_t)(g_known - w.g0); // headers of records before the known one lie entirely below it
	auto follow = [&](uint64_t cand, std::vector<RecHead> &chain) -> bool {
		chain.clear();
		uint64_t q = cand;
		while (q < g_known) {
			if (!plausible(u, (size_t)(q - w.g0), end, f.n_targets, f.target_len.data())) return false;
			const uint8_t *r = u + (q - w.g0);
			uint32_t bs; memcpy(&bs, r, 4);
			RecHead h; h.g = q; memcpy(&h.tid, r + 4, 4); memcpy(&h.pos, r + 8, 4); memcpy(&h.flag, r + 18, 2);
			chain.push_back(h);
			q += 4 + (uint64_t)bs;
		}
		return q == g_known;
	};
	std::vector<RecHead> chain;
	if (k_from == 0) { // the stream's first record starts right behind the header: no guessing
		if (lo >= g_known) return true;
		if (!follow(lo, chain)) { f.err = "walk back: the record chain from the header does not reach the known record"; return false; }
		out.swap(chain);
		return true;
	}
	for (uint64_t cand = lo; cand + 36 <= g_known; ++cand) {
		if (!plausible(u, (size_t)(cand - w.g0), end, f.n_targets, f.target_len.data())) continue;
		if (follow(cand, chain)) { out.swap(chain); return true; }
	}
	return true; // no record starts between the block's start and the known record
}

void to_voffset(const BlockFile &f, uint64_t g, uint64_t *coff, uint32_t *uoff)
{
	if (g >= f.ucum.back()) { *coff = UINT64_MAX; *uoff = 0; return; }
	size_t k = f.block_of(g);
	while (k + 1 < f.blocks.size() && f.blocks[k].isize == 0) ++k; // (empty blocks carry nothing)
	*coff = f.blocks[k].coff; *uoff = (uint32_t)(g - f.ucum[k]);
}

bool from_voffset(const BlockFile &f, uint64_t coff, uint32_t uoff, uint64_t *g)
{
	if (coff == UINT64_MAX) { *g = f.ucum.back(); return true; }
	auto it = std::lower_bound(f.blocks.begin(), f.blocks.end(), coff, [](const BlockPos &b, uint64_t c) { return b.coff < c; });
	if (it == f.blocks.end() || it->coff != coff) return false;
	*g = f.ucum[(size_t)(it - f.blocks.begin())] + uoff;
	return true;
}

// Walk backwards from the known record start g_known, calling visit(record) for every earlier record, latest first, until it returns false
// or the file's first record has been visited.
template <class F>
bool walk_back(BlockFile &f, uint64_t g_known, F visit)
{
	uint64_t known = g_known;
	while (known > f.header_len) {
		size_t k = f.block_of(known - 1);
		std::vector<RecHead> recs;
		// widen the window backwards until it holds the start of a record before `known` (a record can span many blocks)
		for (;;) {
			if (!records_before(f, known, k, recs)) return false;
			if (!recs.empty() || k == 0) break;
			--k;
		}
		if (recs.empty()) break;
		for (size_t i = recs.size(); i-- > 0;) if (!visit(recs[i])) return true;
		known = recs[0].g;
	}
	return true;
}

thread_local std::string g_perr;
thread_local bool g_used_index = false;

} // namespace

extern "C" {

const char *ssvh_partition_last_error(void) { return g_perr.c_str(); }
int ssvh_partition_used_index(void) { return g_used_index ? 1 : 0; }

int ssvh_bam_partition(const char *path, int32_t n_parts, int32_t halo_bp, ssvh_bam_part *parts)
{
	g_perr.clear();
	if (n_parts < 1 || !parts) { g_perr = "bad arguments"; return -1; }
	BlockFile f;
	if (!f.open(path)) { g_perr = f.err; return -1; }
	const uint64_t total = f.ucum.back(), body = total > f.header_len ? total - f.header_len : 0;
	std::vector<uint64_t> own((size_t)n_parts + 1, total);
	own[0] = f.header_len < total ? f.header_len : total;
	// boundaries: with a .bai the record starts of its linear index (one per 16 kb window that holds reads: exact, nothing to verify beyond
	// a look at the header there); without one, speculation checked by two independent record chains (first_record_verified)
	std::vector<uint64_t> bai;
	g_used_index = getenv("SSV_NO_BAI") == nullptr && load_bai_starts(path, f, bai);
	for (int32_t r = 1; r < n_parts; ++r) {
		uint64_t g = f.header_len + body * (uint64_t)r / (uint64_t)n_parts, at = total;
		if (g < own[(size_t)r - 1]) g = own[(size_t)r - 1];
		if (g < total && g_used_index) {
			auto it = std::lower_bound(bai.begin(), bai.end(), g);
			if (it == bai.end()) {
				// behind the last window the linear index knows: the reads without a position at the end of the file have no entries there (and a
				// long tail of them would all go to one rank): this boundary is found like in a file without an index
				if (!first_record_verified(f, g, &at)) { if (!f.err.empty()) { g_perr = f.err; return -1; } at = total; }
			} else at = *it;
			if (it != bai.end() && at < total) { // (the index is the file's own if the bytes there are a record header; otherwise fall back)
				Window w(f);
				w.begin_at_block(f.block_of(at));
				w.need(std::min<uint64_t>(total, at + 4 + 36 + 256));
				if (!plausible(w.u.data(), (size_t)(at - w.g0), w.u.size(), f.n_targets, f.target_len.data())) { g_used_index = false; r = 0; continue; } // start over without the index
			}
		} else if (g < total && !first_record_verified(f, g, &at)) { if (!f.err.empty()) { g_perr = f.err; return -1; } at = total; }
		own[(size_t)r] = g < total ? at : total;
	}
	for (int32_t r = 0; r < n_parts; ++r) {
		ssvh_bam_part &p = parts[r];
		memset(&p, 0, sizeof(p));
		const uint64_t g = own[(size_t)r], ge = own[(size_t)r + 1];
		to_voffset(f, g, &p.own_coff, &p.own_uoff);
		to_voffset(f, ge, &p.end_coff, &p.end_uoff);
		p.scan_coff = p.own_coff; p.scan_uoff = p.own_uoff;
		p.own_tid = f.n_targets; p.own_pos = 0; p.initial_last_tid = 0; p.before_own_tid = 0;
		if (g >= total) continue;
		{ // the part's first record
			Window w(f);
			w.begin_at_block(f.block_of(g));
			w.need(std::min<uint64_t>(total, g + 40));
			const uint8_t *rec = w.u.data() + (g - w.g0);
			int32_t tid, pos; memcpy(&tid, rec + 4, 4); memcpy(&pos, rec + 8, 4);
			if (tid >= 0) { p.own_tid = tid; p.own_pos = pos; }
		}
		if (r == 0) continue;
		// halo: records before the part that start within halo_bp of its first record on the same contig; then the contig of the last
		// mapped-pair record before them
		uint64_t scan = g;
		int64_t halo = 0;
		bool in_halo = p.own_tid < f.n_targets, have_last = false, have_before = false;
		const int32_t t0 = p.own_tid, p0 = p.own_pos;
		const bool ok = walk_back(f, g, [&](const RecHead &h) {
			if (!have_before && !(h.flag & (4 | 8))) { p.before_own_tid = h.tid; have_before = true; }
			if (in_halo) {
				if (h.tid == t0 && (int64_t)h.pos >= (int64_t)p0 - halo_bp) { scan = h.g; ++halo; return true; }
				in_halo = false;
			}
			if (!(h.flag & (4 | 8))) { p.initial_last_tid = h.tid; have_last = true; return false; }
			return true;
		});
		if (!ok) { g_perr = f.err; return -1; }
		(void)have_last; // none before: the start of the file, where the reference's last_tid is 0 (clip_reads.h:407)
		to_voffset(f, scan, &p.scan_coff, &p.scan_uoff);
		p.halo_records = halo;
	}
	return 0;
}

int ssvh_bam_walk_back(const char *path, uint64_t coff, uint32_t uoff, int64_t n_back, uint64_t *out_coff, uint32_t *out_uoff, int64_t *n_found)
{
	g_perr.clear();
	BlockFile f;
	if (!f.open(path)) { g_perr = f.err; return -1; }
	uint64_t g;
	if (!from_voffset(f, coff, uoff, &g)) { g_perr = "not a block of this file"; return -1; }
	uint64_t at = g;
	int64_t n = 0;
	if (n_back > 0 && !walk_back(f, g, [&](const RecHead &h) { at = h.g; return ++n < n_back; })) { g_perr = f.err; return -1; }
	to_voffset(f, at, out_coff, out_uoff);
	if (n_found) *n_found = n;
	return 0;
}

} // extern "C"
