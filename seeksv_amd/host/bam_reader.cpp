// bam_reader.cpp - BGZF + BAM decoding into structure-of-arrays batches.
//
// Replaces, for the hot path, what the reference gets from samtools-0.1.16's libbam below its
// record loops: samopen()/samread() (sam/sam.h:59,73) and the bam1_core_t accessors
// (sam/bam.h:169-255).  Own implementation from the SAM/BAM specification; no libbam here.
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../csrc/cpus.h"
#include "seeksv_host.h"

#include <pthread.h>
#include <zlib.h>

#include "huff_gz.h"

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;

// A small fork-join pool: BGZF blocks are independent deflate streams, so a chunk of blocks is inflated (or deflated) by all
// host threads at once.  (The reference is single threaded and spends 78 % of getclip in zlib inflate + BAM parsing)
class Pool {
public:
	explicit Pool(int n, const char *name = "ssv-pool")
	{
		for (int i = 0; i < n; ++i) th_.emplace_back([this, name] { (void)pthread_setname_np(pthread_self(), name); work(); }); // (the names: SSV_TIMING=2's account of the CPU time, seeksv_cli.cpp)
	}
	~Pool()
	{
		{ std::lock_guard<std::mutex> l(m_); stop_ = true; ++gen_; }
		cv_.notify_all();
		for (auto &t : th_) t.join();
	}
	// run fn(i) for i in [0, n) on the pool (and on the calling thread), return when all are done
	void run(int n, const std::function<void(int)> &fn)
	{
		if (n <= 0) return;
		if (th_.empty() || n == 1) { for (int i = 0; i < n; ++i) fn(i); return; }
		std::lock_guard<std::mutex> one_at_a_time(run_m_);
		{
			std::lock_guard<std::mutex> l(m_);
			fn_ = &fn; n_ = n; next_.store(0); pending_ = n; ++gen_;
		}
		cv_.notify_all();
		drain();
		std::unique_lock<std::mutex> l(m_);
		done_.wait(l, [this] { return pending_ == 0; });
		fn_ = nullptr;
	}
private:
	void drain()
	{
		for (;;) {
			int i = next_.fetch_add(1);
			if (i >= n_) break;
			(*fn_)(i);
			std::lock_guard<std::mutex> l(m_);
			if (--pending_ == 0) done_.notify_all();
		}
	}
	void work()
	{
		uint64_t seen = 0;
		for (;;) {
			{
				std::unique_lock<std::mutex> l(m_);
				cv_.wait(l, [&] { return gen_ != seen; });
				seen = gen_;
				if (stop_) return;
			}
			drain();
		}
	}
	std::vector<std::thread> th_;
	std::mutex m_, run_m_;
	std::condition_variable cv_, done_;
	const std::function<void(int)> *fn_ = nullptr;
	std::atomic<int> next_{0};
	int n_ = 0, pending_ = 0;
	uint64_t gen_ = 0;
	bool stop_ = false;
};

static int host_threads()
{
	const char *e = getenv("SSV_HOST_THREADS");
	int n = e ? atoi(e) : ssv::effective_cpus();
	if (n < 1) n = 1;
	return n > 64 ? 64 : n;
}

static Pool &pool() // the reading side (inflate + decode); may run on the read-ahead thread
{
	static Pool p(host_threads() - 1, "ssv-read");
	return p;
}

static Pool &wpool() // the writing side (deflate), so that a read-ahead in flight and an output writer do not queue behind each other
{
	// SSV_WRITE_THREADS: beyond the readers' cap of 64 (writing a large BAM at deflate level 6 scales with every core of the box)
	static Pool p([] { const char *e = getenv("SSV_WRITE_THREADS"); const int n = e ? atoi(e) : host_threads(); return (n < 1 ? 1 : n > 1024 ? 1024 : n) - 1; }(), "ssv-write");
	return p;
}

struct Bgzf {
	FILE *fp = nullptr;
	static constexpr int CHUNK_BLOCKS = 1024; // up to 64 MB of uncompressed data per refill
	struct Block { std::vector<uint8_t> c; uint32_t isize = 0, use = 0; size_t uoff = 0; bool ok = true; }; // use: bytes of the inflated block that count (all of them, except in the block a range ends in)
	// a range of the file (ssvh_bam_set_range): reading stops at the record that starts end_uoff bytes into the block at file offset end_coff
	long end_coff = -1;
	uint32_t end_uoff = 0;
	bool end_seen = false;
	std::vector<Block> blocks;
	// the inflated window: a plain malloc'd buffer, never value-initialised (fresh pages are touched only by the inflating threads) and
	// reserved once at its working size so that growing it does not copy
	struct RawBuf {
		uint8_t *p = nullptr; size_t cap = 0;
		~RawBuf() { free(p); }
		uint8_t *data() { return p; }
		size_t size() const { return cap; }
		void reserve_keep(size_t need, size_t keep)
		{
			if (need <= cap) return;
			size_t ncap = std::max(need, cap + cap / 2);
			uint8_t *q = (uint8_t *)malloc(ncap);
			if (!q) throw std::bad_alloc();
			if (keep) memcpy(q, p, keep);
			free(p); p = q; cap = ncap;
		}
	} ubuf;
	size_t upos = 0, ulen = 0;
	bool eof = false;

	// read the next compressed block (header validated) into blk; false at EOF or error (g_err set on error)
	bool read_block(Block &blk)
	{
		uint8_t hdr[18];
		bool last_of_range = false;
		if (end_coff >= 0) {
			const long at = ftell(fp);
			if (at > end_coff || (at == end_coff && (end_uoff == 0 || end_seen))) { eof = true; return false; }
			if (at == end_coff) { end_seen = true; last_of_range = true; }
		}
		size_t got = fread(hdr, 1, 18, fp);
		if (got == 0) { eof = true; return false; }
		if (got < 18 || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) { g_err = "not a BGZF block"; eof = true; return false; }
		unsigned xlen = hdr[10] | (hdr[11] << 8);
		// the BC subfield is normally first; walk the extra field to be safe
		std::vector<uint8_t> extra(xlen);
		memcpy(extra.data(), hdr + 12, xlen < 6 ? xlen : 6);
		if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, fp) != xlen - 6) { g_err = "truncated BGZF header"; eof = true; return false; }
		int bsize = -1;
		for (size_t off = 0; off + 4 <= xlen;) {
			unsigned slen = extra[off + 2] | (extra[off + 3] << 8);
			if (off + 4 + slen > xlen) break; // a subfield that runs past the extra field: malformed, stop here
			if (extra[off] == 'B' && extra[off + 1] == 'C' && slen == 2) bsize = extra[off + 4] | (extra[off + 5] << 8);
			off += 4 + slen;
		}
		if (bsize < 0) { g_err = "BGZF block without BC field"; eof = true; return false; }
		if ((size_t)bsize + 1 < 12 + (size_t)xlen + 8) { g_err = "bad BGZF block size"; eof = true; return false; } // (the subtraction below would wrap)
		size_t clen = (size_t)bsize + 1 - 12 - xlen; // deflate data + crc32 + isize
		blk.c.resize(clen);
		if (fread(blk.c.data(), 1, clen, fp) != clen) { g_err = "truncated BGZF block"; eof = true; return false; }
		memcpy(&blk.isize, blk.c.data() + clen - 4, 4);
		blk.use = last_of_range ? std::min(blk.isize, end_uoff) : blk.isize;
		return true;
	}

	// the next block's raw deflate payload straight into dst (no inflate): 1 = ok, 0 = end of file, 2 = does not fit in `room`
	// (file position unchanged), -1 = error (g_err set)
	int read_block_raw(uint8_t *dst, size_t room, uint32_t *payload, uint32_t *isize)
	{
		uint8_t hdr[18];
		const long at = ftell(fp);
		size_t got = fread(hdr, 1, 18, fp);
		if (got == 0) { eof = true; return 0; }
		if (got < 18 || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) { g_err = "not a BGZF block"; eof = true; return -1; }
		const unsigned xlen = hdr[10] | (hdr[11] << 8);
		std::vector<uint8_t> extra(xlen);
		memcpy(extra.data(), hdr + 12, xlen < 6 ? xlen : 6);
		if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, fp) != xlen - 6) { g_err = "truncated BGZF header"; eof = true; return -1; }
		int bsize = -1;
		for (size_t off = 0; off + 4 <= xlen;) {
			unsigned slen = extra[off + 2] | (extra[off + 3] << 8);
			if (off + 4 + slen > xlen) break; // a subfield that runs past the extra field: malformed, stop here
			if (extra[off] == 'B' && extra[off + 1] == 'C' && slen == 2) bsize = extra[off + 4] | (extra[off + 5] << 8);
			off += 4 + slen;
		}
		if (bsize < 0) { g_err = "BGZF block without BC field"; eof = true; return -1; }
		if ((size_t)bsize + 1 < 12 + (size_t)xlen + 8) { g_err = "bad BGZF block size"; eof = true; return -1; } // (the subtraction below would wrap)
		const size_t clen = (size_t)bsize + 1 - 12 - xlen;
		if (clen - 8 + 8 > room) { fseek(fp, at, SEEK_SET); return 2; } // 8 spare bytes: the device bit reader looks 4 bytes ahead
		uint8_t tail[8];
		if (fread(dst, 1, clen - 8, fp) != clen - 8 || fread(tail, 1, 8, fp) != 8) { g_err = "truncated BGZF block"; eof = true; return -1; }
		*payload = (uint32_t)(clen - 8);
		memcpy(isize, tail + 4, 4);
		return 1;
	}

	// read up to CHUNK_BLOCKS compressed blocks and inflate them in parallel into ubuf
	bool refill()
	{
		if (blocks.empty()) blocks.resize(CHUNK_BLOCKS);
		int nb = 0;
		size_t total = 0;
		while (nb < CHUNK_BLOCKS) {
			if (!read_block(blocks[(size_t)nb])) break;
			blocks[(size_t)nb].uoff = total;
			total += blocks[(size_t)nb].use;
			++nb;
		}
		if (!g_err.empty()) return false;
		if (nb == 0) return false;
		ubuf.reserve_keep(total + 65536 + 1, 0);
		uint8_t *out = ubuf.data();
		std::vector<Block> &bl = blocks;
		pool().run(nb, [&bl, out](int i) {
			Block &b = bl[(size_t)i];
			b.ok = true;
			if (b.isize == 0) return;
			z_stream zs;
			memset(&zs, 0, sizeof(zs));
			if (inflateInit2(&zs, -15) != Z_OK) { b.ok = false; return; }
			zs.next_in = b.c.data(); zs.avail_in = (uInt)(b.c.size() - 8);
			zs.next_out = out + b.uoff; zs.avail_out = b.isize;
			int rc = inflate(&zs, Z_FINISH);
			inflateEnd(&zs);
			if (rc != Z_STREAM_END || zs.total_out != b.isize) b.ok = false;
		});
		for (int i = 0; i < nb; ++i) if (!blocks[(size_t)i].ok) { g_err = "BGZF inflate failed"; eof = true; return false; }
		upos = 0; ulen = total;
		return true;
	}

	// read exactly n bytes of the uncompressed stream; returns bytes read (< n only at EOF)
	size_t read(void *dst, size_t n)
	{
		uint8_t *d = (uint8_t *)dst;
		size_t done = 0;
		while (done < n) {
			if (upos == ulen) {
				if (eof && ulen == upos) { if (!refill_after_eof()) break; }
				else if (!refill()) break;
				continue;
			}
			size_t take = ulen - upos < n - done ? ulen - upos : n - done;
			memcpy(d + done, ubuf.data() + upos, take);
			upos += take; done += take;
		}
		return done;
	}
	bool refill_after_eof() { return false; }

	// drop the consumed bytes [0, upos) of the window; returns the shift applied to every offset into it
	size_t compact()
	{
		size_t shift = upos;
		if (shift) { memmove(ubuf.data(), ubuf.data() + upos, ulen - upos); ulen -= shift; upos = 0; }
		return shift;
	}

	// append the next chunk of blocks behind the unread data.  The chunk is cut into segments of SEG_BLOCKS consecutive blocks; one task
	// inflates a segment and then calls scan(segment index, begin, end) on it while its bytes are still in that core's cache.
	// Returns the number of segments (0 at end of file or on error, g_err set on error).
	static constexpr int SEG_BLOCKS = 8;
	size_t window_reserve = 0; // set by the batch reader: the size the window is expected to reach
	int grow(const std::function<void(int, size_t, size_t)> &scan)
	{
		if (eof) return 0;
		if (blocks.empty()) blocks.resize(CHUNK_BLOCKS);
		int nb = 0;
		size_t total = 0;
		while (nb < CHUNK_BLOCKS) {
			if (!read_block(blocks[(size_t)nb])) break;
			blocks[(size_t)nb].uoff = ulen + total;
			total += blocks[(size_t)nb].use;
			++nb;
		}
		if (!g_err.empty() || nb == 0) return 0;
		ubuf.reserve_keep(std::max(ulen + total + 65536 + 1, window_reserve), ulen);
		uint8_t *out = ubuf.data();
		std::vector<Block> &bl = blocks;
		const int nseg = (nb + SEG_BLOCKS - 1) / SEG_BLOCKS;
		pool().run(nseg, [&bl, out, nb, &scan](int sg) {
			const int b0 = sg * SEG_BLOCKS, b1 = std::min(nb, b0 + SEG_BLOCKS);
			bool ok = true;
			for (int i = b0; i < b1; ++i) {
				Block &b = bl[(size_t)i];
				b.ok = true;
				if (b.isize == 0) continue;
				z_stream zs;
				memset(&zs, 0, sizeof(zs));
				if (inflateInit2(&zs, -15) != Z_OK) { b.ok = ok = false; continue; }
				zs.next_in = b.c.data(); zs.avail_in = (uInt)(b.c.size() - 8);
				zs.next_out = out + b.uoff; zs.avail_out = b.isize;
				int rc = inflate(&zs, Z_FINISH);
				inflateEnd(&zs);
				if (rc != Z_STREAM_END || zs.total_out != b.isize) b.ok = ok = false;
			}
			if (ok) scan(sg, bl[(size_t)b0].uoff, bl[(size_t)b1 - 1].uoff + bl[(size_t)b1 - 1].use);
		});
		for (int i = 0; i < nb; ++i) if (!blocks[(size_t)i].ok) { g_err = "BGZF inflate failed"; eof = true; return 0; }
		ulen += total;
		return nseg;
	}
};

struct Unmapped {
	std::string qname, seq, qual;
	int is_read1;
};

} // namespace

// One inflated segment's speculative record list: offsets of the records that start in [begin, end) if the guess of the first record
// start is right, and the offset the chain leaves the segment at.  read_batch verifies every list against the true chain.
struct Segment { std::vector<size_t> list; size_t end = 0, exit = 0; };

// does a BAM record header that could be real start at u[o]?  (o + 36 <= end is the caller's business)
static inline bool plausible_record(const uint8_t *u, size_t o, size_t end, int32_t n_targets, const int32_t *tlen)
{
	uint32_t bs; int32_t refid, pos, l_seq, next_ref, next_pos; uint16_t ncig;
	memcpy(&bs, u + o, 4);
	if (bs < 32 || bs > (1u << 28)) return false;
	const uint8_t *r = u + o + 4;
	memcpy(&refid, r, 4); memcpy(&pos, r + 4, 4); memcpy(&ncig, r + 12, 2); memcpy(&l_seq, r + 16, 4); memcpy(&next_ref, r + 20, 4); memcpy(&next_pos, r + 24, 4);
	const size_t l_name = r[8];
	if (refid < -1 || refid >= n_targets || next_ref < -1 || next_ref >= n_targets || pos < -1 || next_pos < -1 || l_seq < 0 || l_name < 2) return false; // (a read name is at least one character and its NUL)
	if (refid >= 0 && pos >= tlen[refid]) return false; // a position lies inside its contig (the mate's is left alone: this test must never fail a true record)
	if (32 + l_name + 4 * (size_t)ncig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq > (size_t)bs) return false;
	const size_t nul = o + 4 + 32 + l_name - 1;
	return nul >= end || u[nul] == 0;
}

static void find_records(const uint8_t *u, size_t begin, size_t end, int32_t n_targets, const int32_t *tlen, Segment &S)
{
	S.list.clear(); S.end = end;
	size_t o = begin;
	// guess: the first offset from which three plausible headers follow one another
	for (; o + 36 <= end; ++o) {
		size_t q = o; int k = 0;
		for (; k < 3 && q + 36 <= end; ++k) {
			if (!plausible_record(u, q, end, n_targets, tlen)) break;
			uint32_t bs; memcpy(&bs, u + q, 4);
			q += 4 + (size_t)bs;
		}
		if (k == 3 || (k > 0 && q + 36 > end)) break;
	}
	if (o + 36 > end) { S.exit = begin; return; }
	while (o + 4 <= end) {
		uint32_t bs; memcpy(&bs, u + o, 4);
		if (bs < 32) break;
		S.list.push_back(o);
		o += 4 + (size_t)bs;
	}
	S.exit = o;
}

// Batch arrays: grown, never shrunk, contents not kept across a growth (every batch writes all it hands out).  With an allocator set
// (ssvh_bam_set_allocator: page-locked memory from the device library) the arrays can be copied to the GPU asynchronously.
struct HostAlloc { void *(*alloc)(size_t) = nullptr; void (*release)(void *) = nullptr; };

template <class T> struct Col {
	T *p = nullptr;
	size_t n = 0, cap = 0;
	const HostAlloc *a = nullptr;
	void (*release)(void *) = nullptr; // of the allocator `p` came from
	Col() = default;
	Col(const Col &) = delete;
	Col &operator=(const Col &) = delete;
	~Col() { drop(); }
	void drop() { if (p) { if (release) release(p); else free(p); } p = nullptr; cap = 0; }
	void resize(size_t m)
	{
		if (m > cap) {
			const size_t nc = std::max(m, cap + cap / 2);
			drop();
			release = nullptr;
			if (a && a->alloc && a->release && (p = (T *)a->alloc(nc * sizeof(T))) != nullptr) release = a->release;
			if (!p) p = (T *)malloc(nc * sizeof(T)); // (also when the page-locked allocator is exhausted: the copies then run synchronously)
			if (!p) throw std::bad_alloc();
			memset(p, 0, nc * sizeof(T));
			cap = nc;
		}
		n = m;
	}
	void assign(size_t m, T v) { resize(m); for (size_t i = 0; i < m; ++i) p[i] = v; }
	size_t size() const { return n; }
	T *data() { return p; }
	const T *data() const { return p; }
	T &operator[](size_t i) { return p[i]; }
};

struct ssvh_bam {
	Bgzf z;
	std::vector<std::string> names;
	std::vector<int32_t> lens, name_field_len;
	uint64_t header_len = 0; // bytes of the inflated stream before the first record
	// batch storage: three sets, so that with read-ahead the next batch is decoded while the caller still uses the current one AND the one
	// before it (whose copy to the GPU may still be running)
	struct BatchBuf {
		Col<int32_t> tid, pos, l_qseq, mtid, mpos, isize;
		Col<uint16_t> flag, n_cigar;
		Col<uint8_t> mapq, xc, seqqual, ends;
		Col<uint32_t> cigar_off, cigar;
		Col<uint64_t> seq_off;
		std::vector<uint8_t> unmapped_raw;  // the UNMAP|MUNMAP records as they lie in the stream (block_size prefixed), in order
		std::vector<size_t> unmapped_off;   // where each starts
		std::vector<ssv_tid_run> tid_runs; // the tid column as runs (ssv_batch_t.tid_runs)
		void use(const HostAlloc *a) { tid.a = pos.a = l_qseq.a = mtid.a = mpos.a = isize.a = a; flag.a = n_cigar.a = a; mapq.a = xc.a = seqqual.a = ends.a = a; cigar_off.a = cigar.a = a; seq_off.a = a; }
	} buf[3];
	HostAlloc alloc;
	int cur = 0;
	bool readahead = false;
	std::thread ra_thread;
	int ra_rc = 0; std::string ra_err; ssv_batch_t ra_batch; int64_t ra_max = 0; int ra_keep = 0;
	long raw_end_coff = -1;      // raw mode over a range (ssvh_bam_raw_begin_range)
	uint32_t raw_end_uoff = 0;
	uint64_t raw_limit = UINT64_MAX;
	bool raw_empty = false;
	// ssvh_bam_map_blocks: the file mapped three times over (a chunk's bytes are handed out as a pointer into one of the mappings, in turn: a caller
	// that page-locks the pages of the chunks it holds - three at most - never locks two ranges of ONE mapping that share a page)
	uint8_t *map[3] = {nullptr, nullptr, nullptr};
	size_t map_len = 0;
	int map_turn = 0;
	std::vector<uint8_t> rec;
	uint32_t small_cigar[8];
	std::vector<size_t> found; // located, not yet handed out record offsets in z.ubuf
	size_t found_pos = 0, chain_cur = 0;
	std::vector<Segment> segs;
};

static const char NT16[] = "=ACMGRSVTWYHKDBN";

// bam_aux_get(b, "XC") + bam_aux2i (clip_reads.cpp:126-127): integer value of the XC tag, 0 when absent
// or not an integer type.
static int aux_xc(const uint8_t *p, const uint8_t *end)
{
	while (p + 3 <= end) {
		const uint8_t t0 = p[0], t1 = p[1], ty = p[2];
		p += 3;
		int64_t val = 0; bool isint = false;
		size_t sz = 0;
		switch (ty) {
		case 'A': sz = 1; break;
		case 'c': sz = 1; if (p + 1 <= end) { val = (int8_t)p[0]; isint = true; } break;
		case 'C': sz = 1; if (p + 1 <= end) { val = p[0]; isint = true; } break;
		case 's': sz = 2; if (p + 2 <= end) { int16_t v; memcpy(&v, p, 2); val = v; isint = true; } break;
		case 'S': sz = 2; if (p + 2 <= end) { uint16_t v; memcpy(&v, p, 2); val = v; isint = true; } break;
		case 'i': sz = 4; if (p + 4 <= end) { int32_t v; memcpy(&v, p, 4); val = v; isint = true; } break;
		case 'I': sz = 4; if (p + 4 <= end) { uint32_t v; memcpy(&v, p, 4); val = v; isint = true; } break;
		case 'f': sz = 4; break;
		case 'd': sz = 8; break;
		case 'Z': case 'H': { const uint8_t *q = p; while (q < end && *q) ++q; sz = (size_t)(q - p) + 1; break; }
		case 'B': {
			if (p + 5 > end) return 0;
			uint8_t sub = p[0]; uint32_t cnt; memcpy(&cnt, p + 1, 4);
			size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
			sz = 5 + es * cnt; break;
		}
		default: return 0; // unknown type: stop like a failed skip
		}
		if (t0 == 'X' && t1 == 'C') return isint ? (int)val : 0;
		p += sz;
	}
	return 0;
}

extern "C" {

const char *ssvh_last_error(void) { return g_err.c_str(); }

int ssvh_bam_open(const char *path, ssvh_bam **out)
{
	*out = nullptr;
	g_err.clear();
	ssvh_bam *b = new ssvh_bam();
	b->z.fp = fopen(path, "rb");
	if (!b->z.fp) { g_err = std::string("cannot open ") + path; delete b; return -1; }
	char magic[4];
	int32_t l_text, n_ref;
	if (b->z.read(magic, 4) != 4 || memcmp(magic, "BAM\1", 4) != 0) { if (g_err.empty()) g_err = "not a BAM file"; fclose(b->z.fp); delete b; return -1; }
	if (b->z.read(&l_text, 4) != 4 || l_text < 0) { g_err = "bad BAM header"; fclose(b->z.fp); delete b; return -1; }
	std::vector<char> text((size_t)l_text);
	if (b->z.read(text.data(), (size_t)l_text) != (size_t)l_text || b->z.read(&n_ref, 4) != 4 || n_ref < 0) { g_err = "bad BAM header"; fclose(b->z.fp); delete b; return -1; }
	for (int32_t i = 0; i < n_ref; ++i) {
		int32_t l_name, l_ref;
		if (b->z.read(&l_name, 4) != 4 || l_name <= 0) { g_err = "bad BAM header"; fclose(b->z.fp); delete b; return -1; }
		std::string nm((size_t)l_name, '\0');
		if (b->z.read(&nm[0], (size_t)l_name) != (size_t)l_name || b->z.read(&l_ref, 4) != 4) { g_err = "bad BAM header"; fclose(b->z.fp); delete b; return -1; }
		b->name_field_len.push_back(l_name);
		nm.resize(strlen(nm.c_str()));
		b->names.push_back(nm);
		b->lens.push_back(l_ref);
	}
	b->header_len = 12 + (uint64_t)l_text;
	for (int32_t i = 0; i < n_ref; ++i) b->header_len += 8 + (uint64_t)b->name_field_len[(size_t)i];
	*out = b;
	return 0;
}

int ssvh_bam_raw_begin(ssvh_bam *b, uint64_t *first_record_offset)
{
	g_err.clear();
	if (!b->z.fp) { g_err = "no file behind this handle"; return -1; }
	if (b->ra_thread.joinable()) { g_err = "a read-ahead is in flight"; return -1; }
	if (fseek(b->z.fp, 0, SEEK_SET) != 0) { g_err = "cannot seek"; return -1; }
	b->z.eof = false; b->z.upos = b->z.ulen = 0; b->found.clear(); b->found_pos = 0;
	b->raw_end_coff = -1; b->raw_end_uoff = 0; b->raw_limit = UINT64_MAX; b->raw_empty = false;
	*first_record_offset = b->header_len;
	return 0;
}

// Raw mode over the records of [start, end) (virtual offsets as ssvh_bam_set_range takes them): ssvh_bam_read_blocks then starts at the block at
// start_coff and stops behind the block that holds the range's last record; ssvh_bam_raw_limit tells how many inflated bytes of the blocks of
// the latest read_blocks call belong to the range.
int ssvh_bam_raw_begin_range(ssvh_bam *b, uint64_t start_coff, uint32_t start_uoff, uint64_t end_coff, uint32_t end_uoff, uint64_t *first_record_offset)
{
	g_err.clear();
	if (!b->z.fp) { g_err = "no file behind this handle"; return -1; }
	if (b->ra_thread.joinable()) { g_err = "a read-ahead is in flight"; return -1; }
	b->z.upos = b->z.ulen = 0; b->found.clear(); b->found_pos = 0;
	b->raw_end_coff = end_coff == UINT64_MAX ? -1 : (long)end_coff; b->raw_end_uoff = end_uoff; b->raw_limit = UINT64_MAX;
	*first_record_offset = start_uoff;
	if (start_coff == UINT64_MAX || (start_coff == end_coff && start_uoff >= end_uoff)) { b->z.eof = true; b->raw_empty = true; return 0; }
	b->raw_empty = false;
	if (fseek(b->z.fp, (long)start_coff, SEEK_SET) != 0) { g_err = "cannot seek"; return -1; }
	b->z.eof = false;
	return 0;
}

int ssvh_bam_raw_limit(const ssvh_bam *b, uint64_t *inflated_bytes) { *inflated_bytes = b->raw_limit; return 0; }

int ssvh_bam_set_range(ssvh_bam *b, uint64_t start_coff, uint32_t start_uoff, uint64_t end_coff, uint32_t end_uoff)
{
	g_err.clear();
	if (!b->z.fp) { g_err = "no file behind this handle"; return -1; }
	if (b->ra_thread.joinable()) { g_err = "a read-ahead is in flight"; return -1; }
	b->z.upos = b->z.ulen = 0; b->found.clear(); b->found_pos = 0; b->chain_cur = 0;
	if (start_coff == UINT64_MAX) { b->z.eof = true; return 0; } // a range that starts at the end of the file: empty
	if (fseek(b->z.fp, (long)start_coff, SEEK_SET) != 0) { g_err = "cannot seek"; return -1; }
	b->z.eof = false;
	b->z.end_coff = end_coff == UINT64_MAX ? -1 : (long)end_coff; b->z.end_uoff = end_uoff; b->z.end_seen = false;
	if (start_coff == end_coff && start_uoff >= end_uoff && end_coff != UINT64_MAX) { b->z.eof = true; return 0; } // empty range
	// the first chunk of blocks is inflated here, so that the range's first record is where the window begins
	if (!b->z.refill()) { if (!g_err.empty()) return -1; b->z.eof = true; return 0; }
	if ((size_t)start_uoff > b->z.ulen) { g_err = "range start outside its block"; return -1; }
	b->z.upos = start_uoff;
	return 0;
}

// The file's bytes as they are - BGZF headers and trailers included - go into dst in large pieces, every piece read by all host threads at
// once (pread into the caller's buffer, usually page-locked); the block table is then read off the headers where they lie in dst
// (c_off = where a block's deflate payload starts inside dst).  A block at a time through stdio (four calls per block) fed the device
// decoder at 2.8 GB/s - a fifth of what it inflates; the file's pages are in the page cache or on NVMe, and many readers are what both want.
// dst == nullptr: the mapped form (ssvh_bam_map_blocks) - nothing is copied, the blocks are looked at where they lie in the mapping `mapped`, of which
// at most dst_bytes are handed out
// One BGZF block's header and trailer where they lie in memory: 1 = a whole block at d[p, p + size), 0 = it does not end inside d[0, have) yet, -1 = not a BGZF block
// (*err says what is wrong with it)
struct BgzfAt { size_t p; uint32_t hdr, size, isize; }; // hdr = 12 + xlen (where the deflate payload starts), size = BSIZE + 1
static int bgzf_block_at(const uint8_t *d, size_t p, size_t have, BgzfAt &o, const char **err)
{
	if (p + 18 > have) return 0;
	const uint8_t *h = d + p;
	if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { *err = "not a BGZF block"; return -1; }
	const unsigned xlen = h[10] | (h[11] << 8);
	if (p + 12 + xlen > have) return 0;
	int bsize = -1;
	for (size_t off = 0; off + 4 <= xlen;) {
		const unsigned slen = h[12 + off + 2] | (h[12 + off + 3] << 8);
		if (off + 4 + slen > xlen) break; // a subfield that runs past the extra field: malformed, stop here
		if (h[12 + off] == 'B' && h[12 + off + 1] == 'C' && slen == 2) bsize = h[12 + off + 4] | (h[12 + off + 5] << 8);
		off += 4 + slen;
	}
	if (bsize < 0) { *err = "BGZF block without BC field"; return -1; }
	if ((size_t)bsize + 1 < 12 + (size_t)xlen + 8) { *err = "bad BGZF block size"; return -1; }
	if (p + (size_t)bsize + 1 > have) return 0; // the block's tail is not there yet
	uint32_t isize; memcpy(&isize, h + bsize + 1 - 4, 4);
	if (isize > 65536) { *err = "BGZF block that claims to inflate to more than 64 KB"; return -1; }
	o.p = p; o.hdr = 12 + xlen; o.size = (uint32_t)bsize + 1; o.isize = isize;
	return 1;
}

// The headers of the blocks in d[p, have), found by all host threads at once.  Walking them one after the other is a chain of cache misses - a block's place follows from
// the size in the header before it -: 2 ms per 0.3 GB of file, a quarter of what the reader thread spends per chunk, and since pass 2 of the device inflate got its LDS
// window the reader is what a command waits for.  Here the range is cut into one segment per thread; a thread looks for the first block that begins in its segment (the
// gzip magic with the extra-field flag, a BC subfield, and a block behind it that looks the same) and walks on to the first block of the next segment.  The pieces are
// taken over as far as they JOIN - thread 0 starts at p, which is a block; a segment counts if the one before it counts and ends exactly where it begins - so a
// byte pattern that merely looks like a block cannot get in; what does not join is left to the caller's own walk.
static void bgzf_blocks_parallel(const uint8_t *d, size_t p, size_t have, std::vector<BgzfAt> &out)
{
	out.clear();
	const int nt = std::min(host_threads(), 16);
	if (have < p + ((size_t)16 << 20) || nt < 2) return;
	const size_t seg = (have - p) / (size_t)nt;
	std::vector<std::vector<BgzfAt>> part((size_t)nt);
	std::vector<size_t> first((size_t)nt, SIZE_MAX), last((size_t)nt, SIZE_MAX);
	pool().run(nt, [&](int i) {
		const size_t s0 = p + (size_t)i * seg, e0 = i + 1 < nt ? p + (size_t)(i + 1) * seg : have;
		const char *err = nullptr;
		size_t q = s0;
		BgzfAt a, b2;
		if (i > 0) {
			for (;; ++q) {
				if (q >= e0 || q + 18 > have) return; // (a segment without a block start: a block is at most 64 KB, this does not happen; its neighbours will not join)
				if (d[q] != 31 || d[q + 1] != 139 || d[q + 2] != 8 || !(d[q + 3] & 4)) continue;
				if (bgzf_block_at(d, q, have, a, &err) != 1) continue;
				if (q + a.size + 18 <= have && bgzf_block_at(d, q + a.size, have, b2, &err) < 0) continue;
				break;
			}
		}
		first[(size_t)i] = q;
		part[(size_t)i].reserve(seg / 12000 + 16);
		while (q < e0 && bgzf_block_at(d, q, have, a, &err) == 1) { part[(size_t)i].push_back(a); q += a.size; }
		last[(size_t)i] = q;
	});
	size_t total = 0;
	for (auto &v : part) total += v.size();
	out.reserve(total);
	for (int i = 0; i < nt; ++i) {
		if (i > 0 && (first[(size_t)i] == SIZE_MAX || last[(size_t)i - 1] != first[(size_t)i])) break; // does not join: the rest is the caller's
		out.insert(out.end(), part[(size_t)i].begin(), part[(size_t)i].end());
		if (last[(size_t)i] == SIZE_MAX) break;
	}
}

static int read_blocks_impl(ssvh_bam *b, void *dst, const uint8_t *mapped, size_t dst_bytes, uint64_t max_inflated, ssv_bgzf_block *blocks, int64_t max_blocks, int64_t *n_blocks, size_t *n_bytes,
                            const void **ptr)
{
	g_err.clear();
	*n_blocks = 0; *n_bytes = 0;
	if (ptr) *ptr = nullptr;
	b->raw_limit = UINT64_MAX;
	if (b->z.eof) return 0;
	const int fd = fileno(b->z.fp);
	struct stat st;
	if (fstat(fd, &st) != 0) { g_err = "cannot stat the BAM file"; return -1; }
	const uint64_t file_size = (uint64_t)st.st_size;
	const long at0 = ftell(b->z.fp);
	if (at0 < 0) { g_err = "cannot tell the file position"; return -1; }
	if (mapped && file_size != b->map_len) { // (a mapping past the file's present end faults - SIGBUS - where the header walk or the DMA touches it: refuse before)
		g_err = file_size > b->map_len ? "the BAM file grew while it was being read" : "the BAM file was truncated while it was being read";
		return -1;
	}
	const uint8_t *d = mapped ? mapped + at0 : static_cast<const uint8_t *>(dst);
	uint8_t *const dw = static_cast<uint8_t *>(dst);
	if (ptr) *ptr = d;
	const size_t PIECE = (size_t)256 << 20, SLICE = (size_t)4 << 20;
	static const bool timing = [] { const char *e = getenv("SSV_TIMING"); return e ? atoi(e) : 0; }() >= 2; // SSV_TIMING=2: per-chunk detail
	double t_read = 0, t_walk = 0;
	auto clk = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	struct Report { const bool on; double &r, &w; size_t &bytes; int64_t &nb; ~Report() { if (on) fprintf(stderr, "[timing] (read_blocks: %zu bytes, %lld blocks: pread %.4f s, header walk %.4f s)\n", bytes, (long long)nb, r, w); } };
	size_t have = 0;      // bytes of the file (from at0) that are in dst
	size_t p = 0;         // bytes of dst that whole, accepted blocks cover
	uint64_t inflated = 0;
	int64_t n = 0;
	bool stop = false, range_done = false;
	Report report{timing, t_read, t_walk, p, n};
	while (!stop) {
		// more of the file (leave 8 spare bytes in dst: the device bit reader looks 4 bytes past a payload)
		const uint64_t left_in_file = file_size - ((uint64_t)at0 + have);
		size_t want = (size_t)std::min<uint64_t>({(uint64_t)PIECE, left_in_file, dst_bytes > have + 8 ? (uint64_t)(dst_bytes - have - 8) : 0});
		if (b->raw_end_coff >= 0) { // a range of the file: nothing behind the block its last record lies in is needed
			const uint64_t last_needed = (uint64_t)b->raw_end_coff + 65536 + 1024;
			if ((uint64_t)at0 + have + want > last_needed) want = (uint64_t)at0 + have < last_needed ? (size_t)(last_needed - (uint64_t)at0 - have) : 0;
		}
		const double tr0 = timing ? clk() : 0;
		if (mapped) have += want; // (the bytes are there already)
		else if (want) {
			// (all threads of the pool copy: the copy out of the page cache is system time that grows with the threads on it - a quarter sample, 14.4 GB: 16 threads 4.0-7.6
			// CPU-seconds, 8 threads 2.5 - but on a box that grants 16 CPUs of host time eight threads do not keep ahead of the host link: round 6, the whole sample:
			// getsv 1.73-1.87 -> 2.12-2.30 s, the file leg with the read inside 1.23-1.37 -> 1.65-1.94 s)
			const int ns = (int)((want + SLICE - 1) / SLICE);
			std::vector<int> ok((size_t)ns, 1);
			pool().run(ns, [&](int i) {
				size_t off = (size_t)i * SLICE;
				const size_t end = std::min(want, off + SLICE);
				while (off < end) {
					const ssize_t got = pread(fd, dw + have + off, end - off, (off_t)((uint64_t)at0 + have + off));
					if (got <= 0) { ok[(size_t)i] = 0; return; }
					off += (size_t)got;
				}
			});
			for (int v : ok) if (!v) { g_err = "read error on the BAM file"; b->z.eof = true; return -1; }
			have += want;
		}
		const double tw0 = timing ? clk() : 0;
		t_read += tw0 - tr0;
		struct Lap { double &acc; double t0; bool on; std::function<double()> c; ~Lap() { if (on) acc += c() - t0; } } lap{t_walk, tw0, timing, clk};
		// the blocks that are whole in dst: their headers are looked up by all threads first (bgzf_blocks_parallel), the loop below then takes them in file order
		// and parses by itself whatever that did not cover
		std::vector<BgzfAt> pre;
		size_t pi = 0;
		static const bool par_walk = !(getenv("SSV_SERIAL") && strstr(getenv("SSV_SERIAL"), "walk")); // SSV_SERIAL=walk (tests): one header after the other
		if (par_walk) bgzf_blocks_parallel(d, p, have, pre);
		for (;;) {
			if (n >= max_blocks) { stop = true; break; }
			const uint64_t at = (uint64_t)at0 + p;
			if (b->raw_end_coff >= 0 && ((long)at > b->raw_end_coff || ((long)at == b->raw_end_coff && b->raw_end_uoff == 0))) { // the range ends before this block
				if (b->raw_limit == UINT64_MAX) b->raw_limit = inflated;
				range_done = stop = true; break;
			}
			if (p == have && (uint64_t)at0 + have == file_size) { range_done = stop = true; break; } // end of the file
			BgzfAt at1;
			while (pi < pre.size() && pre[pi].p < p) ++pi;
			if (pi < pre.size() && pre[pi].p == p) at1 = pre[pi];
			else {
				const char *perr = nullptr;
				const int rc = bgzf_block_at(d, p, have, at1, &perr);
				if (rc < 0) { g_err = perr; b->z.eof = true; return -1; }
				if (rc == 0) break; // the block's header or tail is not in dst yet
			}
			const uint32_t isize = at1.isize;
			const size_t xlen = at1.hdr - 12, bsize = (size_t)at1.size - 1;
			if (n > 0 && inflated + isize > max_inflated) { stop = true; break; }
			if (isize != 0) { // (empty blocks - the EOF marker - carry nothing)
				blocks[n].c_off = p + 12 + xlen; blocks[n].c_len = (uint32_t)((size_t)bsize + 1 - 12 - xlen - 8); blocks[n].u_len = isize;
				if (b->raw_end_coff >= 0 && (long)at == b->raw_end_coff) { b->raw_limit = inflated + b->raw_end_uoff; range_done = true; } // the range ends inside this block
				inflated += isize;
				++n;
			}
			p += (size_t)bsize + 1;
			if (range_done) { stop = true; break; }
		}
		if (stop) break;
		if (want == 0) { // nothing more can be read, and the next block is not whole
			if ((uint64_t)at0 + have == file_size && p < have) { g_err = "truncated BGZF block"; b->z.eof = true; return -1; }
			if (n == 0 && dst_bytes < 65536 + 1024) { g_err = "buffer smaller than one BGZF block"; return -1; }
			break; // dst is full
		}
	}
	if (fseek(b->z.fp, (long)((uint64_t)at0 + p), SEEK_SET) != 0) { g_err = "cannot seek"; return -1; }
	if (range_done) b->z.eof = true;
	*n_blocks = n; *n_bytes = p;
	return 0;
}

int ssvh_bam_read_blocks(ssvh_bam *b, void *dst, size_t dst_bytes, uint64_t max_inflated, ssv_bgzf_block *blocks, int64_t max_blocks, int64_t *n_blocks, size_t *n_bytes)
{
	if (!dst) { g_err = "ssvh_bam_read_blocks: no buffer"; return -1; }
	return read_blocks_impl(b, dst, nullptr, dst_bytes, max_inflated, blocks, max_blocks, n_blocks, n_bytes, nullptr);
}

// The same chunking without the copy: the chunk's bytes are handed out where they lie in a read-only mapping of the file (the page cache: a caller
// page-locks them and lets the GPU's DMA engines fetch them there - ssv_host_register, seeksv_hip.h - instead of copying 47 GB of a whole-genome
// file into staging buffers first).  The file is mapped three times and the chunks go round the mappings, so that the ranges of three chunks in
// flight never share a page of ONE mapping (page-locking works on whole pages, chunks end where BGZF blocks end).
int ssvh_bam_map_blocks(ssvh_bam *b, size_t max_bytes, uint64_t max_inflated, ssv_bgzf_block *blocks, int64_t max_blocks, int64_t *n_blocks, const void **ptr, size_t *n_bytes)
{
	g_err.clear();
	if (!b->z.fp || !ptr) { g_err = "no file behind this handle"; return -1; }
	if (!b->map[0]) {
		struct stat st;
		const int fd = fileno(b->z.fp);
		if (fstat(fd, &st) != 0 || st.st_size <= 0) { g_err = "cannot stat the BAM file"; return -1; }
		for (int k = 0; k < 3; ++k) {
			void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
			if (m == MAP_FAILED) {
				for (int j = 0; j < k; ++j) { munmap(b->map[j], (size_t)st.st_size); b->map[j] = nullptr; }
				g_err = std::string("cannot map the BAM file: ") + strerror(errno);
				return -2; // (the caller falls back to ssvh_bam_read_blocks)
			}
			b->map[k] = static_cast<uint8_t *>(m);
			(void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
		}
		b->map_len = (size_t)st.st_size;
	}
	const uint8_t *m = b->map[b->map_turn];
	b->map_turn = (b->map_turn + 1) % 3;
	return read_blocks_impl(b, nullptr, m, max_bytes + 8, max_inflated, blocks, max_blocks, n_blocks, n_bytes, ptr);
}

int ssvh_bam_from_header(const char *const *names, const int32_t *lens, int32_t n, ssvh_bam **out)
{
	ssvh_bam *b = new ssvh_bam();
	for (int32_t i = 0; i < n; ++i) { b->names.push_back(names[i]); b->lens.push_back(lens[i]); }
	b->z.eof = true;
	*out = b;
	return 0;
}

void ssvh_bam_close(ssvh_bam *b)
{
	if (!b) return;
	if (b->ra_thread.joinable()) b->ra_thread.join();
	for (uint8_t *m : b->map) if (m) munmap(m, b->map_len);
	if (b->z.fp) fclose(b->z.fp);
	delete b;
}

int32_t ssvh_bam_n_targets(const ssvh_bam *b) { return (int32_t)b->names.size(); }
const char *ssvh_bam_target_name(const ssvh_bam *b, int32_t tid) { return (tid >= 0 && (size_t)tid < b->names.size()) ? b->names[(size_t)tid].c_str() : nullptr; }
int32_t ssvh_bam_target_len(const ssvh_bam *b, int32_t tid) { return (tid >= 0 && (size_t)tid < b->lens.size()) ? b->lens[(size_t)tid] : -1; }
const int32_t *ssvh_bam_target_lens(const ssvh_bam *b) { return b->lens.data(); }

static int decode_batch(ssvh_bam *b, ssvh_bam::BatchBuf &B, int64_t max_records, int keep_all_seq, ssv_batch_t *out)
{
	g_err.clear();
	B.unmapped_raw.clear(); B.unmapped_off.clear();
	Bgzf &z = b->z;
	static const bool timing = [] { const char *e = getenv("SSV_TIMING"); return e ? atoi(e) : 0; }() >= 3; // SSV_TIMING=3: per-batch detail of the host reader
	auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t0 = now();
	// ---- record boundaries.  Walking the block_size chain is a chain of dependent cache misses, so it is done speculatively per
	// inflated segment by the task that just inflated it (find_records), and only stitched together here. ----
	const size_t BYTE_LIMIT = (size_t)384 << 20; // stop growing the window once a batch holds this much raw BAM
	z.window_reserve = BYTE_LIMIT + 2 * (size_t)Bgzf::CHUNK_BLOCKS * 65536;
	std::vector<size_t> &found = b->found;
	if (b->found_pos >= found.size()) { found.clear(); b->chain_cur = z.upos; }
	else found.erase(found.begin(), found.begin() + (ptrdiff_t)b->found_pos);
	b->found_pos = 0;
	{
		const size_t shift = z.compact();
		if (shift) { for (size_t &o : found) o -= shift; b->chain_cur -= shift; }
	}
	const int32_t n_targets = (int32_t)b->names.size();
	std::vector<Segment> &segs = b->segs;
	// whole records in already inflated bytes that no segment list covers (the window the header read left behind): follow the chain by hand
	auto hop_tail = [&]() -> bool {
		const uint8_t *u = z.ubuf.data();
		size_t cur = b->chain_cur;
		while (cur + 4 <= z.ulen) {
			uint32_t bs; memcpy(&bs, u + cur, 4);
			if (bs < 32) { g_err = "corrupt BAM record"; return false; }
			if (cur + 4 + (size_t)bs > z.ulen) break;
			found.push_back(cur);
			cur += 4 + (size_t)bs;
		}
		b->chain_cur = cur;
		return true;
	};
	if (!hop_tail()) return -1;
	while ((int64_t)found.size() < max_records && b->chain_cur < BYTE_LIMIT) {
		if (segs.size() < (size_t)(Bgzf::CHUNK_BLOCKS / Bgzf::SEG_BLOCKS + 1)) segs.resize((size_t)(Bgzf::CHUNK_BLOCKS / Bgzf::SEG_BLOCKS + 1));
		Bgzf *zp = &z;
		const double tg0 = now();
		const int32_t *tlen = b->lens.data();
		const int nseg = z.grow([&segs, zp, n_targets, tlen](int sg, size_t s0, size_t s1) { find_records(zp->ubuf.data(), s0, s1, n_targets, tlen, segs[(size_t)sg]); });
		if (nseg == 0) {
			if (!g_err.empty()) return -1;
			break;
		}
		const double tg1 = now();
		if (timing) fprintf(stderr, "  grow %.4f s nseg %d ulen %zu cap %zu found %zu\n", tg1 - tg0, nseg, z.ulen, z.ubuf.size(), found.size());
		const uint8_t *u = z.ubuf.data();
		size_t cur = b->chain_cur;
		bool window_end = false;
		for (int sg = 0; sg < nseg && !window_end; ++sg) {
			const Segment &S = segs[(size_t)sg];
			while (cur < S.end) {
				size_t k = 0;
				if (S.list.empty() || S.list[0] != cur) k = (size_t)(std::lower_bound(S.list.begin(), S.list.end(), cur) - S.list.begin());
				if (k < S.list.size() && S.list[k] == cur) { found.insert(found.end(), S.list.begin() + (ptrdiff_t)k, S.list.end()); cur = S.exit; break; }
				// not on the speculated chain (a record that straddles segments, or a wrong guess): follow the true chain by hand
				uint32_t bs;
				if (cur + 4 > z.ulen) { window_end = true; break; }
				memcpy(&bs, u + cur, 4);
				if (bs < 32) { g_err = "corrupt BAM record"; return -1; }
				if (cur + 4 + (size_t)bs > z.ulen) { window_end = true; break; }
				found.push_back(cur);
				cur += 4 + (size_t)bs;
			}
		}
		// records of the last list that run past the window are completed by the next chunk
		while (!found.empty()) {
			uint32_t bs; memcpy(&bs, u + found.back(), 4);
			if (found.back() + 4 + (size_t)bs <= z.ulen) break;
			cur = found.back(); found.pop_back();
		}
		b->chain_cur = cur;
		if (!hop_tail()) return -1;
	}
	if (z.eof && g_err.empty() && (int64_t)found.size() < max_records && b->chain_cur != z.ulen) { g_err = "truncated BAM record"; return -1; }
	const double t1 = now();
	const int64_t n = std::min<int64_t>((int64_t)found.size(), max_records);
	const std::vector<size_t> &off = found;
	const uint8_t *base = z.ubuf.data();
	// ---- pass 1 (parallel over record ranges, header fields only): sizes of the variable-length parts, then their prefix ----
	B.tid.resize((size_t)n); B.pos.resize((size_t)n); B.l_qseq.resize((size_t)n); B.mtid.resize((size_t)n); B.mpos.resize((size_t)n); B.isize.resize((size_t)n);
	B.flag.resize((size_t)n); B.n_cigar.resize((size_t)n); B.mapq.resize((size_t)n); B.xc.resize((size_t)n); B.ends.resize((size_t)n); B.cigar_off.resize((size_t)n); B.seq_off.resize((size_t)n);
	const int nt = (int)std::min<int64_t>(std::max<int64_t>(1, n / 8192), 128);
	std::vector<uint64_t> c_of((size_t)nt + 1, 0), s_of((size_t)nt + 1, 0);
	std::atomic<int> bad{0};
	pool().run(nt, [&](int t) {
		const int64_t i0 = n * t / nt, i1 = n * (t + 1) / nt;
		uint64_t ct = 0, st = 0;
		for (int64_t i = i0; i < i1; ++i) {
			const uint8_t *r = base + off[(size_t)i] + 4;
			uint32_t bs; memcpy(&bs, r - 4, 4);
			uint16_t ncig; int32_t l_seq;
			memcpy(&ncig, r + 12, 2); memcpy(&l_seq, r + 16, 4);
			const size_t o_cig = 32 + (size_t)r[8];
			if (l_seq < 0 || o_cig + 4 * (size_t)ncig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq > (size_t)bs) { bad.store(1); return; }
			bool soft = false;
			if (ncig) {
				uint32_t c0, cl; memcpy(&c0, r + o_cig, 4); memcpy(&cl, r + o_cig + 4 * ((size_t)ncig - 1), 4);
				soft = (c0 & 15) == 4 || (cl & 15) == 4;
			}
			ct += ncig;
			if (soft || keep_all_seq) st += ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
		}
		c_of[(size_t)t + 1] = ct; s_of[(size_t)t + 1] = st;
	});
	if (bad.load()) { g_err = "corrupt BAM record"; return -1; }
	for (int t = 0; t < nt; ++t) { c_of[(size_t)t + 1] += c_of[(size_t)t]; s_of[(size_t)t + 1] += s_of[(size_t)t]; }
	const uint64_t ctot = c_of[(size_t)nt], stot = s_of[(size_t)nt];
	if (ctot > 0xffffffffull) { g_err = "too many CIGAR operations in one batch"; return -1; }
	if (B.cigar.size() < (size_t)ctot + 1) B.cigar.resize((size_t)ctot + 1);
	if (B.seqqual.size() < (size_t)stot + 16) B.seqqual.resize((size_t)stot + 16);
	const double t2 = now();
	// ---- pass 2 (parallel): decode the records into the structure-of-arrays batch ----
	std::vector<int64_t> span_of((size_t)nt, 1);
	std::vector<std::vector<uint8_t>> un((size_t)nt); // per thread: its share's UNMAP|MUNMAP records, raw
	pool().run(nt, [&](int t) {
		const int64_t i0 = n * t / nt, i1 = n * (t + 1) / nt;
		int64_t max_span = 1;
		uint64_t c_run = c_of[(size_t)t], s_run = s_of[(size_t)t];
		for (int64_t i = i0; i < i1; ++i) {
			const uint8_t *r = base + off[(size_t)i] + 4;
			uint32_t bs; memcpy(&bs, r - 4, 4);
			int32_t refid, pos, l_seq, next_ref, next_pos, tlen;
			uint16_t ncig, flag;
			memcpy(&refid, r, 4); memcpy(&pos, r + 4, 4);
			const uint8_t l_read_name = r[8], mapq = r[9];
			memcpy(&ncig, r + 12, 2); memcpy(&flag, r + 14, 2); memcpy(&l_seq, r + 16, 4);
			memcpy(&next_ref, r + 20, 4); memcpy(&next_pos, r + 24, 4); memcpy(&tlen, r + 28, 4);
			const size_t o_name = 32, o_cig = o_name + l_read_name, o_seq = o_cig + 4 * (size_t)ncig;
			const size_t o_qual = o_seq + ((size_t)l_seq + 1) / 2, o_aux = o_qual + (size_t)l_seq;
			B.tid[(size_t)i] = refid; B.pos[(size_t)i] = pos; B.flag[(size_t)i] = flag; B.mapq[(size_t)i] = mapq; B.n_cigar[(size_t)i] = ncig;
			B.l_qseq[(size_t)i] = l_seq; B.mtid[(size_t)i] = next_ref; B.mpos[(size_t)i] = next_pos; B.isize[(size_t)i] = tlen;
			uint32_t *cd = B.cigar.data() + c_run;
			B.cigar_off[(size_t)i] = (uint32_t)c_run; c_run += ncig;
			if (ncig) memcpy(cd, r + o_cig, 4 * (size_t)ncig);
			int64_t span = 0;
			for (unsigned k = 0; k < ncig; ++k) { unsigned op = cd[k] & 15; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += cd[k] >> 4; }
			if (span > max_span) max_span = span;
			const bool soft = ncig && ((cd[0] & 15) == 4 || (cd[ncig - 1] & 15) == 4);
			B.ends[(size_t)i] = ncig ? (uint8_t)((cd[0] & 15u) | ((cd[ncig - 1] & 15u) << 4)) : (uint8_t)0xff; // the hot copy of the CIGAR's end operations (ssv_batch_t.cigar_ends)
			B.xc[(size_t)i] = soft ? (uint8_t)(aux_xc(r + o_aux, r + bs) != 0) : (uint8_t)0;
			if (soft || keep_all_seq) {
				B.seq_off[(size_t)i] = s_run;
				memcpy(B.seqqual.data() + s_run, r + o_seq, o_aux - o_seq);
				s_run += o_aux - o_seq;
			} else B.seq_off[(size_t)i] = SSV_NO_SEQ;
			if (flag & (4 | 8)) {
				un[(size_t)t].insert(un[(size_t)t].end(), r - 4, r + bs); // the unmapped-pair side channel decodes them (GetSeqAndQual, clip_reads.cpp:375-388): ssvh_raw_record_fastq
			}
		}
		span_of[(size_t)t] = max_span;
	});
	int64_t max_span = 1;
	{ size_t total = 0; for (auto &u : un) total += u.size(); B.unmapped_raw.reserve(total); }
	for (int t = 0; t < nt; ++t) {
		max_span = std::max(max_span, span_of[(size_t)t]);
		B.unmapped_raw.insert(B.unmapped_raw.end(), un[(size_t)t].begin(), un[(size_t)t].end());
	}
	if (timing) fprintf(stderr, "[read_batch] n=%lld inflate+find %.3f s, sizes %.3f s, decode %.3f s, window %zu MB\n", (long long)n, t1 - t0, t2 - t1, now() - t2, z.ulen >> 20);
	// the batch's raw bytes are consumed; records already located behind them stay queued for the next call
	b->found_pos = (size_t)n;
	z.upos = (size_t)n < found.size() ? found[(size_t)n] : b->chain_cur;
	memset(out, 0, sizeof(*out));
	out->n = n; out->mem = SSV_MEM_HOST;
	out->max_ref_span = (int32_t)(max_span > INT32_MAX ? INT32_MAX : max_span);
	out->tid = B.tid.data(); out->pos = B.pos.data(); out->flag = B.flag.data(); out->mapq = B.mapq.data();
	out->n_cigar = B.n_cigar.data(); out->l_qseq = B.l_qseq.data(); out->mtid = B.mtid.data(); out->mpos = B.mpos.data();
	out->isize = B.isize.data(); out->cigar_off = B.cigar_off.data(); out->cigar = B.cigar.data(); out->xc = B.xc.data();
	out->seq_off = B.seq_off.data(); out->seqqual = B.seqqual.data(); out->cigar_ends = B.ends.data();
	out->n_cigar_total = (int64_t)ctot; out->seqqual_bytes = (int64_t)stot;
	// the tid column as runs: one per contig in a coordinate-sorted file (more than the consumer takes: left out, it then reads the column)
	B.tid_runs.clear();
	for (int64_t i = 0; i < n && B.tid_runs.size() <= 64; ++i)
		if (i == 0 || B.tid.data()[i] != B.tid.data()[i - 1]) B.tid_runs.push_back(ssv_tid_run{i, B.tid.data()[i], 0});
	if (!B.tid_runs.empty() && B.tid_runs.size() <= 64) { out->tid_runs = B.tid_runs.data(); out->n_tid_runs = (int64_t)B.tid_runs.size(); }
	return 0;
}

int ssvh_bam_read_batch(ssvh_bam *b, int64_t max_records, int keep_all_seq, ssv_batch_t *out)
{
	if (!b->readahead) { b->cur = 0; return decode_batch(b, b->buf[0], max_records, keep_all_seq, out); }
	int rc;
	if (b->ra_thread.joinable()) {
		b->ra_thread.join();
		if (b->ra_max != max_records || b->ra_keep != keep_all_seq) { g_err = "read-ahead is on: every ssvh_bam_read_batch call must ask for the same batch"; return -1; }
		b->cur = (b->cur + 1) % 3;
		rc = b->ra_rc; g_err = b->ra_err; *out = b->ra_batch;
	} else rc = decode_batch(b, b->buf[b->cur], max_records, keep_all_seq, out);
	if (rc == 0 && out->n > 0) {
		b->ra_max = max_records; b->ra_keep = keep_all_seq;
		b->ra_thread = std::thread([b] {
			b->ra_rc = decode_batch(b, b->buf[(b->cur + 1) % 3], b->ra_max, b->ra_keep, &b->ra_batch);
			b->ra_err = g_err;
		});
	}
	return rc;
}

int ssvh_bam_set_allocator(ssvh_bam *b, void *(*alloc)(size_t), void (*release)(void *))
{
	if (b->ra_thread.joinable()) { g_err = "a read-ahead is in flight"; return -1; }
	b->alloc.alloc = alloc; b->alloc.release = release;
	for (auto &B : b->buf) B.use(&b->alloc); // arrays that exist keep their memory until they have to grow
	return 0;
}

int ssvh_bam_set_readahead(ssvh_bam *b, int on)
{
	if (b->ra_thread.joinable()) { g_err = "a read-ahead is in flight"; return -1; }
	b->readahead = on != 0;
	return 0;
}

int ssvh_bam_next_record(ssvh_bam *b, ssvh_record *out)
{
	g_err.clear();
	if (b->ra_thread.joinable()) { g_err = "a read-ahead is in flight: record-at-a-time reads cannot be mixed in"; return -1; }
	b->found.clear(); b->found_pos = 0; // record-at-a-time reads continue at z.upos; read_batch re-derives its chain from there
	int32_t block_size;
	const uint8_t *r;
	if (b->z.ulen - b->z.upos >= 4 && (memcpy(&block_size, b->z.ubuf.data() + b->z.upos, 4), block_size >= 32) && b->z.ulen - b->z.upos - 4 >= (size_t)block_size) {
		// the whole record lies in the inflated window (all but one in a few thousand do): handed out where it lies, valid until the next call
		r = b->z.ubuf.data() + b->z.upos + 4;
		b->z.upos += 4 + (size_t)block_size;
	} else {
		size_t got = b->z.read(&block_size, 4);
		if (got == 0) return g_err.empty() ? 0 : -1;
		if (got != 4 || block_size < 32) { if (g_err.empty()) g_err = "truncated BAM record"; return -1; }
		b->rec.resize((size_t)block_size + 4);
		if (b->z.read(b->rec.data(), (size_t)block_size) != (size_t)block_size) { if (g_err.empty()) g_err = "truncated BAM record"; return -1; }
		r = b->rec.data();
	}
	memcpy(&out->tid, r, 4); memcpy(&out->pos, r + 4, 4);
	const uint8_t l_read_name = r[8];
	out->mapq = r[9];
	memcpy(&out->n_cigar, r + 12, 2); memcpy(&out->flag, r + 14, 2); memcpy(&out->l_qseq, r + 16, 4);
	const size_t o_cig = 32 + (size_t)l_read_name, o_seq = o_cig + 4 * (size_t)out->n_cigar, o_qual = o_seq + ((size_t)out->l_qseq + 1) / 2;
	if (out->l_qseq < 0 || o_qual + (size_t)out->l_qseq > (size_t)block_size) { g_err = "corrupt BAM record"; return -1; }
	out->qname = (const char *)r + 32;
	if (out->n_cigar <= 8) { // aligned copy (a few operations: no vector is touched)
		if (out->n_cigar) memcpy(b->small_cigar, r + o_cig, 4 * (size_t)out->n_cigar);
		out->cigar = b->small_cigar;
	} else {
		b->buf[0].cigar.assign((size_t)out->n_cigar, 0);
		memcpy(b->buf[0].cigar.data(), r + o_cig, 4 * (size_t)out->n_cigar);
		out->cigar = b->buf[0].cigar.data();
	}
	out->seq = r + o_seq; out->qual = r + o_qual;
	return 1;
}

// ---- BAM writer (tooling) ----

// deflate level of the BAM writer: 1 (fast; the tests' fixtures) unless SSV_BGZF_LEVEL says otherwise (bench.py's file legs: 6, samtools' default, where
// >= 64 CPUs write the file; -2 where 16 CPUs have to write a whole-genome sample - the line's `coder` field says which)
// -1: no zlib at all - a block's payload is one literal-only Huffman block (huff_gz.h; what `seeksv realign` writes its clip.bam with unless the
// variable says otherwise: 5.5 M short records that are read back once, by the next command)
// -2: the same with string matching of the cheapest kind (huff_gz.h: deflate_fast; bench.py's whole-genome file where 16 CPUs have to write it)
static int bgzf_level()
{
	const char *e = getenv("SSV_BGZF_LEVEL");
	const int l = e ? atoi(e) : 1;
	return l < -2 ? -2 : l > 9 ? 9 : l;
}

static void bgzf_compress_blocks(const std::vector<uint8_t> &raw, std::vector<uint8_t> &out)
{
	const size_t BS = 0xff00;
	const size_t nb = (raw.size() + BS - 1) / BS;
	std::vector<std::vector<uint8_t>> comp(nb);
	const int level = bgzf_level();
	wpool().run((int)nb, [&](int i) {
		const size_t off = (size_t)i * BS, len = std::min(BS, raw.size() - off);
		std::vector<uint8_t> &c = comp[(size_t)i];
		const uint8_t hdr[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0};
		if (level < 0) { // literals only, or - when their codes do not even pay for the code table - stored (a block must stay below 64 KB)
			static thread_local std::vector<uint8_t> room;
			if (room.size() < ssvh_huff::member_bound(BS)) room.resize(ssvh_huff::member_bound(BS));
			size_t clen = (size_t)((level == -2 ? ssvh_huff::deflate_fast(raw.data() + off, len, room.data()) : ssvh_huff::deflate_literals(raw.data() + off, len, room.data())) - room.data());
			if (clen > len + 5) {
				uint8_t *p = room.data();
				const uint16_t l16 = (uint16_t)len, n16 = (uint16_t)~l16;
				p[0] = 1; memcpy(p + 1, &l16, 2); memcpy(p + 3, &n16, 2); memcpy(p + 5, raw.data() + off, len);
				clen = len + 5;
			}
			c.resize(18 + clen + 8);
			memcpy(c.data(), hdr, 16);
			const uint16_t bsize = (uint16_t)(clen + 25);
			memcpy(c.data() + 16, &bsize, 2);
			memcpy(c.data() + 18, room.data(), clen);
			const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), raw.data() + off, (uInt)len), isz = (uint32_t)len;
			memcpy(c.data() + 18 + clen, &crc, 4); memcpy(c.data() + 18 + clen + 4, &isz, 4);
			return;
		}
		c.resize(len + 1024);
		// one deflate state per thread, reset between blocks: deflateInit2 allocates ~270 KB in pieces that malloc serves by mmap, and a few
		// hundred threads mapping and unmapping at once queue on the process's address-space lock (the writer ran at 1.4 M records/s whatever
		// the thread count)
		struct Deflater { z_stream zs; int level = -1; ~Deflater() { if (level >= 0) deflateEnd(&zs); } };
		static thread_local Deflater D;
		if (D.level != level) {
			if (D.level >= 0) deflateEnd(&D.zs);
			memset(&D.zs, 0, sizeof(D.zs));
			deflateInit2(&D.zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
			D.level = level;
		} else deflateReset(&D.zs);
		z_stream &zs = D.zs;
		zs.next_in = const_cast<uint8_t *>(raw.data() + off); zs.avail_in = (uInt)len;
		zs.next_out = c.data() + 18; zs.avail_out = (uInt)(c.size() - 18 - 8);
		deflate(&zs, Z_FINISH);
		size_t clen = zs.total_out;
		memcpy(c.data(), hdr, 16);
		uint16_t bsize = (uint16_t)(clen + 25);
		memcpy(c.data() + 16, &bsize, 2);
		uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), raw.data() + off, (uInt)len), isz = (uint32_t)len;
		memcpy(c.data() + 18 + clen, &crc, 4); memcpy(c.data() + 18 + clen + 4, &isz, 4);
		c.resize(18 + clen + 8);
	});
	for (auto &c : comp) out.insert(out.end(), c.begin(), c.end());
}

static int reg2bin(int beg, int end)
{
	--end;
	if (beg >> 14 == end >> 14) return ((1 << 15) - 1) / 7 + (beg >> 14);
	if (beg >> 17 == end >> 17) return ((1 << 12) - 1) / 7 + (beg >> 17);
	if (beg >> 20 == end >> 20) return ((1 << 9) - 1) / 7 + (beg >> 20);
	if (beg >> 23 == end >> 23) return ((1 << 6) - 1) / 7 + (beg >> 23);
	if (beg >> 26 == end >> 26) return ((1 << 3) - 1) / 7 + (beg >> 26);
	return 0;
}

static int write_batch_impl(const char *path, const char *const *names, const int32_t *lens, int32_t n_targets, const ssv_batch_t *b,
                            const char *qname_prefix, int64_t first_index, const char *const *qnames, int append, int finish);

int ssvh_bam_write_batch(const char *path, const char *const *names, const int32_t *lens, int32_t n_targets, const ssv_batch_t *b,
                         const char *qname_prefix, int64_t first_index, int append, int finish)
{
	return write_batch_impl(path, names, lens, n_targets, b, qname_prefix, first_index, nullptr, append, finish);
}

int ssvh_bam_write_batch_named(const char *path, const char *const *names, const int32_t *lens, int32_t n_targets, const ssv_batch_t *b,
                               const char *const *qnames, int append, int finish)
{
	if (b && b->n > 0 && !qnames) { g_err = "ssvh_bam_write_batch_named: no read names"; return -1; }
	for (int64_t i = 0; b && i < b->n; ++i)
		if (strlen(qnames[i]) > 254) { g_err = "read name longer than 254 characters (BAM's limit)"; return -1; }
	return write_batch_impl(path, names, lens, n_targets, b, nullptr, 0, qnames, append, finish);
}

static int write_batch_impl(const char *path, const char *const *names, const int32_t *lens, int32_t n_targets, const ssv_batch_t *b,
                            const char *qname_prefix, int64_t first_index, const char *const *qnames, int append, int finish)
{
	FILE *f = fopen(path, append ? "ab" : "wb");
	if (!f) { g_err = std::string("cannot open ") + path; return -1; }
	static const bool timing = [] { const char *e = getenv("SSV_TIMING"); return e ? atoi(e) : 0; }() >= 3; // SSV_TIMING=3: the writer's phases
	double t_sizes = 0, t_fill = 0, t_comp = 0, t_io = 0;
	auto clk = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double tl = clk();
	auto lap = [&](double &acc) { const double now = clk(); acc += now - tl; tl = now; };
	std::vector<uint8_t> raw, out;
	auto put32 = [&](int32_t v) { uint8_t t[4]; memcpy(t, &v, 4); raw.insert(raw.end(), t, t + 4); };
	if (!append) {
		std::string text = "@HD\tVN:1.0\tSO:coordinate\n";
		for (int32_t i = 0; i < n_targets; ++i) text += std::string("@SQ\tSN:") + names[i] + "\tLN:" + std::to_string(lens[i]) + "\n";
		raw.insert(raw.end(), {'B', 'A', 'M', 1});
		put32((int32_t)text.size()); raw.insert(raw.end(), text.begin(), text.end());
		put32(n_targets);
		for (int32_t i = 0; i < n_targets; ++i) { size_t l = strlen(names[i]) + 1; put32((int32_t)l); raw.insert(raw.end(), names[i], names[i] + l); put32(lens[i]); }
		bgzf_compress_blocks(raw, out); // (the header in blocks of its own, as before)
		raw.clear();
	}
	// the records: sizes, their running sum, then every host thread serialises its share of the records straight into place
	const int64_t n = b ? b->n : 0;
	if (n > 0) {
		const std::string prefix = qname_prefix ? qname_prefix : "r";
		auto digits = [](int64_t v) { size_t d = 1; while (v >= 10) { v /= 10; ++d; } return d; };
		// a read with an unmapped end shares its name with its mate (the side channel of getclip pairs by name, clip_reads.h:172-219): the generator plants
		// such pairs on records 2k and 2k + 1 (synth_core.h: unmap_permille), and both are named after 2k
		auto name_index = [&](int64_t i) { const int64_t g = first_index + i; return (b->flag[i] & (4 | 8)) ? (g & ~(int64_t)1) : g; };
		std::vector<uint64_t> at((size_t)n + 1);
		const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n / 4096, 256));
		wpool().run(nt, [&](int w) {
			for (int64_t i = n * w / nt, e = n * (w + 1) / nt; i < e; ++i) {
				const size_t ln = qnames ? strlen(qnames[i]) : prefix.size() + digits(name_index(i));
				const size_t lq = (size_t)b->l_qseq[i];
				at[(size_t)i + 1] = 4 + 32 + ln + 1 + 4 * (size_t)b->n_cigar[i] + (lq + 1) / 2 + lq;
			}
		});
		at[0] = 0;
		for (int64_t i = 0; i < n; ++i) at[(size_t)i + 1] += at[(size_t)i];
		lap(t_sizes);
		// slices of ~128 MB of serialised records: serialise, compress, write, next - a multi-GB batch in one piece held twice
		// its inflated size in memory (the records, the compressed blocks and their copy in `out`)
		const uint64_t slice_bytes = (uint64_t)128 << 20;
		for (int64_t i0 = 0; i0 < n;) {
			int64_t i1 = (int64_t)(std::upper_bound(at.begin() + i0 + 1, at.end(), at[(size_t)i0] + slice_bytes) - at.begin()) - 1;
			if (i1 <= i0) i1 = i0 + 1; // (a single record larger than a slice)
			const int64_t ns = i1 - i0;
			const uint64_t base = at[(size_t)i0];
			const size_t head = raw.size(); // (the BAM header of a fresh file travels with the first slice)
			raw.resize(head + (size_t)(at[(size_t)i1] - base));
			const int nts = (int)std::max<int64_t>(1, std::min<int64_t>(ns / 4096, 256));
		wpool().run(nts, [&](int w) {
			char num[24];
			for (int64_t i = i0 + ns * w / nts, e = i0 + ns * (w + 1) / nts; i < e; ++i) {
				uint8_t *d = raw.data() + head + (at[(size_t)i] - base);
				const int lq = b->l_qseq[i], nc = b->n_cigar[i];
				const uint32_t *cig = b->cigar + b->cigar_off[i];
				int span = 0;
				for (int k = 0; k < nc; ++k) { unsigned op = cig[k] & 15; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += (int)(cig[k] >> 4); }
				const int32_t pos = b->pos[i];
				const int bin = pos >= 0 ? reg2bin(pos, pos + (span > 0 ? span : 1)) : 4680;
				const char *qn; size_t ln;
				std::string tmp;
				if (qnames) { qn = qnames[i]; ln = strlen(qn); }
				else { tmp = prefix; tmp.append(num, (size_t)snprintf(num, sizeof(num), "%lld", (long long)name_index(i))); qn = tmp.c_str(); ln = tmp.size(); }
				const int32_t body = (int32_t)(at[(size_t)i + 1] - at[(size_t)i] - 4);
				auto w32 = [&](int32_t v) { memcpy(d, &v, 4); d += 4; };
				w32(body); w32(b->tid[i]); w32(pos);
				const uint8_t hdr8[8] = {(uint8_t)(ln + 1), b->mapq[i], (uint8_t)(bin & 255), (uint8_t)(bin >> 8), (uint8_t)(nc & 255), (uint8_t)(nc >> 8), (uint8_t)(b->flag[i] & 255), (uint8_t)(b->flag[i] >> 8)};
				memcpy(d, hdr8, 8); d += 8;
				w32(lq); w32(b->mtid[i]); w32(b->mpos[i]); w32(b->isize[i]);
				memcpy(d, qn, ln + 1); d += ln + 1;
				memcpy(d, cig, 4 * (size_t)nc); d += 4 * (size_t)nc;
				const size_t sq = ((size_t)lq + 1) / 2 + (size_t)lq;
				if (b->seq_off[i] != SSV_NO_SEQ) memcpy(d, b->seqqual + b->seq_off[i], sq);
				else { memset(d, 0x11, ((size_t)lq + 1) / 2); memset(d + ((size_t)lq + 1) / 2, 30, (size_t)lq); } // no bases shipped: l_qseq 'A's of quality 30
			}
		});
			lap(t_fill);
			bgzf_compress_blocks(raw, out);
			raw.clear();
			lap(t_comp);
			if (fwrite(out.data(), 1, out.size(), f) != out.size()) { fclose(f); g_err = std::string("write error on ") + path; return -1; }
			out.clear();
			lap(t_io);
			i0 = i1;
		}
	}
	if (!raw.empty()) bgzf_compress_blocks(raw, out);
	if (finish) {
		static const uint8_t eofb[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
		out.insert(out.end(), eofb, eofb + 28);
	}
	if (fwrite(out.data(), 1, out.size(), f) != out.size()) { fclose(f); g_err = std::string("write error on ") + path; return -1; }
	if (fclose(f) != 0) { g_err = std::string("write error on ") + path; return -1; }
	lap(t_io);
	if (timing) fprintf(stderr, "[timing] (write_batch: %lld records: sizes %.3f s, serialise %.3f s, compress %.3f s, write %.3f s)\n", (long long)n, t_sizes, t_fill, t_comp, t_io);
	return 0;
}

int ssvh_gz_append_v(const char *path, const char *const *texts, const size_t *lens, int count, int append)
{
	FILE *f = fopen(path, append ? "ab" : "wb");
	if (!f) { g_err = std::string("cannot open ") + path; return -1; }
	static const size_t PIECE = [] { const char *e = getenv("SSV_GZ_PIECE_KB"); const long kb = e ? atol(e) : 256; return (size_t)(kb < 1 ? 1 : kb) << 10; }(); // (tests: many members from little text)
	struct Piece { const char *p; size_t n; };
	std::vector<Piece> pieces;
	for (int k = 0; k < count; ++k)
		for (size_t off = 0; off < lens[k]; off += PIECE) pieces.push_back(Piece{texts[k] + off, std::min(PIECE, lens[k] - off)});
	if (pieces.empty() && !append) pieces.push_back(Piece{"", 0}); // an empty file is one empty member, like gzstream's
	const size_t np = pieces.size();
	std::vector<std::vector<uint8_t>> comp(np);
	std::vector<int> ok(np, 1);
	// deflate level of the .gz outputs: 1 by default (rows of random-looking bases and qualities: level 6 - gzstream's default, the
	// reference's - packs them 3.4 x at 11 MB/s per core, level 1 2.9 x at 60; the files' decompressed bytes are what counts), SSV_GZ_LEVEL=6
	// writes files of the reference's size
	// ... and by default no zlib at all: members of one literal-only Huffman block (huff_gz.h): 2.2-2.3 x at several hundred MB/s per core
	static const int level = [] { const char *e = getenv("SSV_GZ_LEVEL"); const int l = e && *e ? atoi(e) : -1; return l < 0 ? -1 : l > 9 ? 9 : l; }();
	wpool().run((int)np, [&](int i) {
		const Piece &pc = pieces[(size_t)i];
		std::vector<uint8_t> &c = comp[(size_t)i];
		if (level < 0) {
			static thread_local std::vector<uint8_t> room; // (sized once per thread: a fresh vector would be zero-filled for every member)
			if (room.size() < ssvh_huff::member_bound(pc.n)) room.resize(ssvh_huff::member_bound(pc.n));
			const size_t got = ssvh_huff::member(reinterpret_cast<const uint8_t *>(pc.p), pc.n, room.data());
			c.assign(room.begin(), room.begin() + (ptrdiff_t)got);
			return;
		}
		// one deflate state per thread (see bgzf_compress_blocks)
		struct Deflater { z_stream zs; int level = -100; ~Deflater() { if (level != -100) deflateEnd(&zs); } };
		static thread_local Deflater D;
		if (D.level != level) {
			if (D.level != -100) deflateEnd(&D.zs);
			memset(&D.zs, 0, sizeof(D.zs));
			if (deflateInit2(&D.zs, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) { D.level = -100; ok[(size_t)i] = 0; return; }
			D.level = level;
		} else deflateReset(&D.zs);
		z_stream &zs = D.zs;
		c.resize(deflateBound(&zs, (uLong)pc.n) + 64);
		zs.next_in = (Bytef *)const_cast<char *>(pc.p); zs.avail_in = (uInt)pc.n;
		zs.next_out = c.data(); zs.avail_out = (uInt)c.size();
		if (deflate(&zs, Z_FINISH) != Z_STREAM_END) ok[(size_t)i] = 0;
		c.resize(zs.total_out);
	});
	int rc = 0;
	for (size_t i = 0; i < np; ++i) {
		if (!ok[i]) { g_err = "deflate failed"; rc = -1; break; }
		if (fwrite(comp[i].data(), 1, comp[i].size(), f) != comp[i].size()) { g_err = std::string("write error on ") + path; rc = -1; break; }
	}
	fclose(f);
	return rc;
}

int ssvh_gz_append(const char *path, const char *text, size_t n, int append) { return ssvh_gz_append_v(path, &text, &n, 1, append); }

size_t ssvh_raw_record_fastq(const uint8_t *raw, size_t raw_bytes, size_t offset, const char **qname, const char **seq, const char **qual, int *is_read1)
{
	static thread_local Unmapped u;
	if (offset + 36 > raw_bytes) return 0;
	uint32_t bs; memcpy(&bs, raw + offset, 4);
	if (bs < 32 || offset + 4 + (size_t)bs > raw_bytes) return 0;
	const uint8_t *r = raw + offset + 4;
	uint16_t ncig, flag; int32_t l_seq;
	memcpy(&ncig, r + 12, 2); memcpy(&flag, r + 14, 2); memcpy(&l_seq, r + 16, 4);
	const size_t o_seq = 32 + (size_t)r[8] + 4 * (size_t)ncig, o_qual = o_seq + ((size_t)l_seq + 1) / 2;
	if (l_seq < 0 || o_qual + (size_t)l_seq > bs) return 0;
	u.qname.assign((const char *)r + 32, strnlen((const char *)r + 32, r[8] ? (size_t)r[8] - 1 : 0));
	u.seq.resize((size_t)l_seq);
	for (int32_t k = 0; k < l_seq; ++k) u.seq[(size_t)k] = NT16[(r[o_seq + (k >> 1)] >> ((~k & 1) << 2)) & 15];
	if (l_seq > 0 && r[o_qual] == 0xff) u.qual = "*";
	else { u.qual.resize((size_t)l_seq); for (int32_t k = 0; k < l_seq; ++k) u.qual[(size_t)k] = (char)(r[o_qual + k] + 33); }
	*qname = u.qname.c_str(); *seq = u.seq.c_str(); *qual = u.qual.c_str(); *is_read1 = (flag & 64) ? 1 : 0;
	return offset + 4 + (size_t)bs;
}

static void unmapped_index(const ssvh_bam *b)
{
	auto &B = const_cast<ssvh_bam *>(b)->buf[b->cur];
	if (!B.unmapped_off.empty() || B.unmapped_raw.empty()) return;
	for (size_t off = 0; off + 4 <= B.unmapped_raw.size();) { uint32_t bs; memcpy(&bs, B.unmapped_raw.data() + off, 4); B.unmapped_off.push_back(off); off += 4 + (size_t)bs; }
}

int64_t ssvh_bam_unmapped_count(const ssvh_bam *b) { unmapped_index(b); return (int64_t)b->buf[b->cur].unmapped_off.size(); }

int ssvh_bam_unmapped_get(const ssvh_bam *b, int64_t k, const char **qname, const char **seq, const char **qual, int *is_read1)
{
	unmapped_index(b);
	const auto &B = b->buf[b->cur];
	if (k < 0 || (size_t)k >= B.unmapped_off.size()) return -1;
	return ssvh_raw_record_fastq(B.unmapped_raw.data(), B.unmapped_raw.size(), B.unmapped_off[(size_t)k], qname, seq, qual, is_read1) ? 0 : -1;
}

int ssvh_bam_unmapped_raw(const ssvh_bam *b, const uint8_t **raw, size_t *bytes)
{
	const auto &B = b->buf[b->cur];
	*raw = B.unmapped_raw.data(); *bytes = B.unmapped_raw.size();
	return 0;
}

} // extern "C"
