// unmapped_pairs.h - the unmapped-pair FASTQ side channel of getclip, off the thread that feeds the GPU.
// Reference: StoreUnmapSeqAndQual (clip_reads.h:172-219) + GetSeqAndQual (clip_reads.cpp:375-388), called for every record with UNMAP or MUNMAP
// (clip_reads.h:415-419): look the read name up in a std::map; absent -> keep (bases, qualities, '1' if READ1 else '2'); present with the OTHER end ->
// write `@name/1` (READ1's data, whichever came first) to prefix.unmapped_1.fq.gz and `@name/2` to prefix.unmapped_2.fq.gz and erase the entry; present
// with the SAME end -> the new record is dropped.  Bases as stored (no reverse complement), qualities + 33, "*" when the first quality byte is 0xff.
// What is left at the end is never written.
//
// A whole-genome BAM carries 1-3 % such records (6-18 M of 617 M): a string-keyed std::map and the FASTQ text built record by record on the thread
// that hands chunks to the GPU was ~1 us a record.  Here the feeder thread only hands over a pointer: a batch's records stay where the
// reader left them (raw BAM records, block_size prefixed: ssv_bamdec_info.unmapped_raw / ssvh_bam_unmapped_raw - both readers keep them valid while
// the NEXT batch is read); a first thread copies them (20 MB per 1 GB chunk at 2 %: the reader's buffer is free again within milliseconds, whatever
// the threads behind are doing), a second walks the copy, pairs the records through an open-addressing table keyed by a 64-bit hash of the name (the
// name is compared on a hit; mates lie side by side in a coordinate-sorted BAM, so the table stays small) and lets a few helper threads turn the pairs
// into FASTQ text, a third hands the text to the gzip writer.  A record whose mate has not come by the end of its batch moves into an arena.
#pragma once

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../csrc/thp.h"

namespace seeksv {

class UnmappedPairs {
public:
	// sink1 / sink2: called from a thread of their own with pieces of prefix.unmapped_1.fq / _2.fq text, in file order
	typedef std::function<void(std::vector<std::string> &)> Sink;
	UnmappedPairs(Sink sink1, Sink sink2, int format_threads) : sink1_(std::move(sink1)), sink2_(std::move(sink2)), nfmt_(format_threads < 1 ? 1 : format_threads)
	{
		table_.assign(1024, Slot{});
		copier_ = std::thread([this] { copy_in(); });
		worker_ = std::thread([this] { run(); });
		writer_ = std::thread([this] { write_out(); });
	}
	~UnmappedPairs() { finish(); }
	// The reader's thread: the UNMAP|MUNMAP records of the batch just read, in order.  Returns once the batch BEFORE this one has been copied (its bytes may
	// then be overwritten); this batch's bytes must stay valid until the next submit() or finish() returns.
	void submit(const uint8_t *raw, size_t bytes)
	{
		std::unique_lock<std::mutex> lk(mu_);
		cv_.wait(lk, [this] { return !busy_; });
		if (!bytes) return;
		raw_ = raw; bytes_ = bytes; busy_ = true;
		cv_.notify_all();
	}
	void finish()
	{
		{
			std::unique_lock<std::mutex> lk(mu_);
			if (done_) return;
			cv_.wait(lk, [this] { return !busy_; });
			done_ = true;
			cv_.notify_all();
		}
		copier_.join();
		{ std::lock_guard<std::mutex> lk(in_mu_); in_done_ = true; }
		in_cv_.notify_all();
		worker_.join();
		{ std::lock_guard<std::mutex> lk(out_mu_); out_done_ = true; }
		out_cv_.notify_all();
		writer_.join();
	}
	uint64_t records() const { return n_records_; }
	uint64_t pairs() const { return n_pairs_; }
	double busy_seconds() const { return busy_s_; }

private:
	struct Rec { // a raw BAM record, where it lies
		const uint8_t *r = nullptr; // behind block_size
		uint32_t name_len = 0, l_seq = 0, o_seq = 0;
		bool read1 = false;
		const char *name() const { return reinterpret_cast<const char *>(r + 32); }
	};
	struct Slot { uint64_t h = 0; Rec rec; bool used = false, in_batch = false; };
	struct Pair { Rec first, second; }; // first: READ1's

	static bool parse(const uint8_t *raw, size_t bytes, size_t off, Rec &out, size_t &next)
	{
		if (off + 36 > bytes) return false;
		uint32_t bs; memcpy(&bs, raw + off, 4);
		if (bs < 32 || off + 4 + (size_t)bs > bytes) return false;
		const uint8_t *r = raw + off + 4;
		uint16_t ncig, flag; int32_t l_seq;
		memcpy(&ncig, r + 12, 2); memcpy(&flag, r + 14, 2); memcpy(&l_seq, r + 16, 4);
		const size_t o_seq = 32 + (size_t)r[8] + 4 * (size_t)ncig;
		if (l_seq < 0 || o_seq + ((size_t)l_seq + 1) / 2 + (size_t)l_seq > bs) return false;
		out.r = r; out.l_seq = (uint32_t)l_seq; out.o_seq = (uint32_t)o_seq; out.read1 = (flag & 64) != 0;
		out.name_len = (uint32_t)strnlen(reinterpret_cast<const char *>(r + 32), r[8] ? (size_t)r[8] - 1 : 0); // bounded by l_read_name: a name without its NUL must not run on
		next = off + 4 + (size_t)bs;
		return true;
	}
	static uint64_t hash_name(const char *p, size_t n)
	{
		uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
		size_t i = 0;
		for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, p + i, 8); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 29; }
		uint64_t w = 0;
		if (i < n) { memcpy(&w, p + i, n - i); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 29; }
		h *= 0xC4CEB9FE1A85EC53ull;
		return h ^ (h >> 32);
	}
	static void fastq(const Rec &a, char end, std::string &out)
	{
		static const struct Tab { char t[256][2]; Tab() { static const char NT16[] = "=ACMGRSVTWYHKDBN"; for (int b = 0; b < 256; ++b) { t[b][0] = NT16[b >> 4]; t[b][1] = NT16[b & 15]; } } } tab;
		const size_t n = a.l_seq, at = out.size();
		const uint8_t *s = a.r + a.o_seq, *q = s + (n + 1) / 2;
		const bool no_qual = n > 0 && q[0] == 0xff;
		out.resize(at + 1 + a.name_len + 3 + n + 3 + (no_qual ? 1 : n) + 1);
		char *d = &out[at];
		*d++ = '@'; memcpy(d, a.name(), a.name_len); d += a.name_len; *d++ = '/'; *d++ = end; *d++ = '\n';
		for (size_t k = 0; k + 1 < n; k += 2) { d[k] = tab.t[s[k >> 1]][0]; d[k + 1] = tab.t[s[k >> 1]][1]; }
		if (n & 1) d[n - 1] = tab.t[s[n >> 1]][0];
		d += n; *d++ = '\n'; *d++ = '+'; *d++ = '\n';
		if (no_qual) *d++ = '*'; else { for (size_t k = 0; k < n; ++k) d[k] = (char)(q[k] + 33); d += n; }
		*d++ = '\n';
	}

	void grow()
	{
		std::vector<Slot> old;
		old.swap(table_);
		table_.assign(old.size() * 2, Slot{});
		for (const Slot &s : old) if (s.used) { size_t i = (size_t)s.h & (table_.size() - 1); while (table_[i].used) i = (i + 1) & (table_.size() - 1); table_[i] = s; }
	}
	// the slot that holds (h, name) - or the free one where it would go
	size_t find(uint64_t h, const Rec &rec) const
	{
		const size_t mask = table_.size() - 1;
		size_t i = (size_t)h & mask;
		for (; table_[i].used; i = (i + 1) & mask)
			if (table_[i].h == h && table_[i].rec.name_len == rec.name_len && memcmp(table_[i].rec.name(), rec.name(), rec.name_len) == 0) break;
		return i;
	}
	// backward-shift deletion: the probe sequences of the entries behind `i` stay unbroken
	void erase(size_t i)
	{
		const size_t mask = table_.size() - 1;
		for (size_t j = (i + 1) & mask; table_[j].used; j = (j + 1) & mask) {
			const size_t home = (size_t)table_[j].h & mask;
			if (((j - home) & mask) >= ((j - i) & mask)) { table_[i] = table_[j]; i = j; }
		}
		table_[i] = Slot{};
		--live_;
	}

	void one_batch(const uint8_t *raw, size_t bytes)
	{
		pairs_.clear();
		batch_new_.clear();
		Rec rec, mate;
		for (size_t off = 0, nxt = 0; parse(raw, bytes, off, rec, nxt); off = nxt) {
			++n_records_;
			// Mates lie side by side in a coordinate-sorted BAM (the unmapped end is placed on its mate): with nothing waiting in the table, a record followed by
			// the other end of its own pair is exactly "insert, find, write, erase" - done here without the table (a whole-genome sample's 12-25 M such
			// records were 100 ns each through it, on one thread)
			size_t nxt2;
			if (live_ == 0 && parse(raw, bytes, nxt, mate, nxt2) && mate.read1 != rec.read1 && mate.name_len == rec.name_len && memcmp(mate.name(), rec.name(), rec.name_len) == 0) {
				pairs_.push_back(rec.read1 ? Pair{rec, mate} : Pair{mate, rec});
				++n_records_;
				nxt = nxt2;
				continue;
			}
			const uint64_t h = hash_name(rec.name(), rec.name_len);
			const size_t i = find(h, rec);
			if (table_[i].used) {
				const Rec &kept = table_[i].rec;
				if (kept.read1 == rec.read1) continue; // the same end again: dropped (clip_reads.h:186-216 has no branch for it)
				pairs_.push_back(rec.read1 ? Pair{rec, kept} : Pair{kept, rec});
				erase(i);
				continue;
			}
			table_[i].used = true; table_[i].h = h; table_[i].rec = rec; table_[i].in_batch = true;
			batch_new_.push_back(std::make_pair(h, rec));
			++live_;
			if (live_ * 2 > table_.size()) grow();
		}
		// FASTQ text: the pairs in order, cut into pieces for the helper threads
		const size_t np = pairs_.size();
		n_pairs_ += np;
		if (np) {
			const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)nfmt_, np / 512));
			std::vector<std::string> t1((size_t)nt), t2((size_t)nt);
			auto piece = [&](int w) {
				const size_t lo = np * (size_t)w / (size_t)nt, hi = np * (size_t)(w + 1) / (size_t)nt;
				std::string &a = t1[(size_t)w], &b = t2[(size_t)w];
				size_t need = 0;
				for (size_t k = lo; k < hi; ++k) need += 2 * (size_t)pairs_[k].first.l_seq + pairs_[k].first.name_len + 16;
				a.reserve(need); b.reserve(need);
				ssv::thp_advise(a.data(), a.capacity()); ssv::thp_advise(b.data(), b.capacity());
				for (size_t k = lo; k < hi; ++k) { fastq(pairs_[k].first, '1', a); fastq(pairs_[k].second, '2', b); }
			};
			std::vector<std::thread> th;
			for (int w = 1; w < nt; ++w) th.emplace_back(piece, w);
			piece(0);
			for (auto &t : th) t.join();
			// compressing and writing is a third thread's: it may have to queue behind the row writer's large pieces (one job at a time on the
			// writers' pool), and the reader must not wait for that - the text needs no bytes of the batch any more
			{ std::lock_guard<std::mutex> lk(out_mu_); out_.emplace_back(std::move(t1), std::move(t2)); }
			out_cv_.notify_all();
		}
		// what waits for a mate in a later batch moves into memory of its own (the batch's bytes are the reader's again after this)
		for (const auto &hn : batch_new_) {
			Slot &s = table_[find(hn.first, hn.second)];
			if (!s.used || !s.in_batch) continue; // (its mate came within the batch)
			const uint8_t *r = s.rec.r;
			const size_t n = (size_t)s.rec.o_seq + ((size_t)s.rec.l_seq + 1) / 2 + (size_t)s.rec.l_seq;
			if (arena_.empty() || arena_.back().size() + n > arena_.back().capacity()) { arena_.emplace_back(); arena_.back().reserve(std::max<size_t>(n, (size_t)1 << 20)); }
			std::vector<uint8_t> &a = arena_.back();
			const size_t at = a.size();
			a.insert(a.end(), r, r + n); // (within its capacity: earlier records do not move)
			s.rec.r = a.data() + at;
			s.in_batch = false;
		}
	}

	void write_out()
	{
		for (;;) {
			std::pair<std::vector<std::string>, std::vector<std::string>> job;
			{
				std::unique_lock<std::mutex> lk(out_mu_);
				out_cv_.wait(lk, [this] { return !out_.empty() || out_done_; });
				if (out_.empty()) return;
				job = std::move(out_.front());
				out_.pop_front();
			}
			sink1_(job.first);
			sink2_(job.second);
		}
	}

	// the reader's bytes into memory of our own: the only thing the reader ever waits for
	void copy_in()
	{
		for (;;) {
			const uint8_t *raw; size_t bytes;
			{
				std::unique_lock<std::mutex> lk(mu_);
				cv_.wait(lk, [this] { return busy_ || done_; });
				if (!busy_) return;
				raw = raw_; bytes = bytes_;
			}
			std::unique_ptr<uint8_t[]> own(new uint8_t[bytes]);
			ssv::thp_advise(own.get(), bytes);
			memcpy(own.get(), raw, bytes);
			{ std::lock_guard<std::mutex> lk(in_mu_); in_.emplace_back(std::move(own), bytes); }
			in_cv_.notify_all();
			{ std::lock_guard<std::mutex> lk(mu_); busy_ = false; }
			cv_.notify_all();
		}
	}

	void run()
	{
		for (;;) {
			std::pair<std::unique_ptr<uint8_t[]>, size_t> batch;
			{
				std::unique_lock<std::mutex> lk(in_mu_);
				in_cv_.wait(lk, [this] { return !in_.empty() || in_done_; });
				if (in_.empty()) return;
				batch = std::move(in_.front());
				in_.pop_front();
			}
			const auto t0 = std::chrono::steady_clock::now();
			one_batch(batch.first.get(), batch.second);
			busy_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		}
	}

	Sink sink1_, sink2_;
	int nfmt_;
	std::thread copier_, worker_, writer_;
	std::mutex mu_, in_mu_, out_mu_;
	std::condition_variable cv_, in_cv_, out_cv_;
	std::deque<std::pair<std::unique_ptr<uint8_t[]>, size_t>> in_; // copied batches waiting for the pairing thread
	bool in_done_ = false;
	std::deque<std::pair<std::vector<std::string>, std::vector<std::string>>> out_;
	bool out_done_ = false;
	const uint8_t *raw_ = nullptr;
	size_t bytes_ = 0;
	bool busy_ = false, done_ = false;
	std::vector<Slot> table_;
	size_t live_ = 0;
	std::vector<Pair> pairs_;
	std::vector<std::pair<uint64_t, Rec>> batch_new_; // the names this batch put into the table
	std::deque<std::vector<uint8_t>> arena_;
	uint64_t n_records_ = 0, n_pairs_ = 0;
	double busy_s_ = 0;
};

} // namespace seeksv
