// seeksv_cli.cpp - the `seeksv` command line on top of libseeksv_host (BAM decoding, getsv bookkeeping) and
// libseeksv_hip (HIP kernels).  Keeps the reference's sub-commands, flags, defaults, file names, output columns and
// stderr / stdout messages for the hot path (seeksv.cpp:128-155 CallGetclip, :157-330 CallGetsv):
//
//   seeksv getclip [-t f] [-q n] [-s] [-o prefix] in.sorted.bam
//   seeksv getsv   [options] clip.bam in.sorted.bam clip.gz out.sv out.unmapped.clip.fq
//
// getsv: junctions come from `-B <table>` (ReadBreakpoint, getsv.cpp:1292) and / or from the join of clip.gz with clip.bam
// (junction_stage.cpp); the BAM passes run on the GPU; OutputBreakpoint's filter chain and columns are reproduced here.
// No .bai is needed: the discordant pass scans the BAM instead of seeking in it.
#include <getopt.h>
#include <zlib.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <deque>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <dirent.h>
#include <pthread.h>
#include <sys/resource.h>
#include <unistd.h>
#include <atomic>
#include "../csrc/cpus.h"
#include <condition_variable>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "junction_stage.h"
#include "somatic_stage.h"
#include "unmapped_pairs.h"
#include "../csrc/thp.h"
#include "seeksv_hip.h"
#include "seeksv_host.h"

using namespace std;
using seeksv::Junction;
using seeksv::JunctionMap;
using seeksv::OtherInfo;
using seeksv::SeqInfo;
using seeksv::parse_cigar;
using seeksv::NormalClusters;
using seeksv::SomaticRow;
using seeksv::load_normal_clusters;
using seeksv::parse_tumor_table;
using seeksv::probe_normal_clusters;
using seeksv::scan_tumor_table;

static const char *kVersion = "1.2.3-mi355x";

// SSV_TIMING=1: wall time of the phases on stderr
struct PhaseTimer {
	bool on = getenv("SSV_TIMING") != nullptr;
	std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
	std::map<std::string, double> acc;
	std::vector<std::string> order;
	void lap(const char *name)
	{
		if (!on) return;
		auto t1 = std::chrono::steady_clock::now();
		if (!acc.count(name)) order.push_back(name);
		acc[name] += std::chrono::duration<double>(t1 - t0).count();
		t0 = t1;
	}
	~PhaseTimer() { if (on) for (auto &n : order) std::cerr << "[timing] " << n << " " << acc[n] << " s" << std::endl; }
};

// A fatal error: the reference's message and exit status 1.  Reader, emitter and rank threads may be running (and the HIP context coming up on its own
// thread): the process leaves through _exit - no static destructor pulls a buffer away under a thread that is still copying into it - after everything
// written so far has been flushed.
[[noreturn]] static void die(const string &msg)
{
	cerr << msg << endl;
	cout.flush(); cerr.flush(); fflush(nullptr);
	_exit(1);
}

// The end of a command: every output file is closed by now.  Handing multi-GB device buffers and page-locked staging memory back one
// hipFree / hipHostFree at a time takes 0.3-0.4 s; the process is about to end and the driver reclaims all of it at once, so the context is
// only drained (SSV_CLEAN_EXIT=1 - sanitizer and leak-check runs - destroys it the long way, and main() returns instead of _exit).
static const bool kCleanExit = getenv("SSV_CLEAN_EXIT") != nullptr;
static bool g_verify_crc = false; // -C: the device decoder checks the CRC32 of every BGZF block it inflates
static void release_ctx(ssv_ctx *ctx);

[[noreturn]] static void usage_top()
{
	cerr << "Program: seeksv (structural variation / virus integration detection; MI355X build of the soft-clip hot path)\n"
	     << "Version: " << kVersion << "\n\n"
	     << "Usage: seeksv <command> [options]\n\n"
	     << "Command: getclip\tget soft-clipped reads\n"
	     << "         getsv  \tget final sv\n"
	     << "         somatic\tget somatic sv\n"
	     << "         realign\talign the clipped sequences of getclip to a reference (stand-in for the pipeline's external `bwa mem` step)\n"
	     << "         run    \tgetclip + realign + getsv in one process: the BAM is decoded once and stays in GPU memory" << endl;
	exit(1);
}

[[noreturn]] static void usage_getclip()
{
	cerr << "Usage: seeksv getclip [options] <input.sorted.bam>\n\n"
	     << "Options: -t <double>           Threshold of match rate while combining two soft-clipped reads [0.9]\n"
	     << "         -q <int>              Minimum mapping quality of soft-clipped reads [1]\n"
	     << "         -s                    Save the low quality sequence clipped before alignment by bwa.\n"
	     << "         -o <string>           Prefix of output files [output]\n"
	     << "         -G <int>[,<int>...]   GPU ordinal(s) [0; with -N: all GPUs of the machine in order]\n"
	     << "         -N <int>              cut the BAM into N runs of records, one per GPU (ranks share GPUs when there are fewer) [1]\n"
	     << "         -H <int>              with -N: halo in bp before a run's first record (>= the longest reference span of a read) [65536]\n"
	     << "         -Z                    inflate and decode the BAM on the GPU (compressed blocks over PCIe) instead of on the host threads" << endl
	     << "         -C                    with -Z: check every BGZF block's CRC32 on the GPU (off: like libbam 0.1.16, which checks none)" << endl;
	exit(1);
}

[[noreturn]] static void usage_getsv()
{
	cerr << "Usage: seeksv getsv [options] <input clipped sequence bam> <input orignal sorted bam> <soft-clipped reads file(*clip.gz)> <output SVs> <output unmaped clipped sequence fastq>\n"
	     << "Options: -B <FILE>             junction table (23 columns, as written by getsv) to evaluate in addition\n"
	     << "         -l <int>              Maximum search length to find microhomology [50]\n"
	     << "         -q <int>              Minimum mapping quality of discordant read pair [20]\n"
	     << "         -n <int>              Number of read pairs used to calculate insert size [5000000]; < 100000 switches the discordant pass off\n"
	     << "         -b <int>              Minimum number of soft clipping reads (left + right) [3]\n"
	     << "         -d <int>              Minimum distance between the two breakpoints [50]\n"
	     << "         -D                    Do not calculate depth of the breakpoints and their ajacency regions (sets -f 0)\n"
	     << "         -e <int>              Minimum number of read pairs which support the junction [0]\n"
	     << "         -f <double>           Minimum mutation frequency [0.1]\n"
	     << "         -T <int>              Maximum length of microhomology [50]\n"
	     << "         -m <int>              Minimum length of up_seq / down_seq when no read pair supports the junction [30]\n"
	     << "         -i <int>              Maximum indel number of up_seq / down_seq when no read pair supports the junction [1]\n"
	     << "         -L <int>              Flank length for the average depths [200]\n"
	     << "         -t <double> -Q <int> -w <int>   accepted for compatibility\n"
	     << "         -G <int>[,<int>...]   GPU ordinal(s) [0; with -N: all GPUs of the machine in order]\n"
	     << "         -N <int>              cut the BAM into N runs of records, one per GPU; the ranks' tallies and depths meet in one RCCL all-gather [1]\n"
	     << "         -Z                    inflate and decode the BAM on the GPU (compressed blocks over PCIe) instead of on the host threads" << endl
	     << "         -C                    with -Z: check every BGZF block's CRC32 on the GPU (off: like libbam 0.1.16, which checks none)" << endl;
	exit(1);
}

[[noreturn]] static void usage_somatic()
{
	cerr << "Usage: seeksv somatic [options] <input normal original bam> <input normal soft-clipped reads file(*.clip.gz)> <input tumor SV file> <output somatic SV file>\n\n"
	     << "         -t <double>           Threshold of match rate while comparing two soft-clipped reads [0.9]\n"
	     << "         -q <int>              Minimum mapping quality of discordant read pair [20]\n"
	     << "         -l <int>              Maximum search length to find microhomology [30]\n"
	     << "         -m <int>              Minimum length of the clipped sequence  in normal [10]\n"
	     << "         -n <int>              Number of read pairs used to calculate insert size [5000000]; < 100000 switches the insert-size pass off\n"
	     << "         -G <int>              GPU ordinal [0]\n"
	     << "         -Z                    inflate and decode the BAM on the GPU (compressed blocks over PCIe) instead of on the host threads" << endl
	     << "         -C                    with -Z: check every BGZF block's CRC32 on the GPU (off: like libbam 0.1.16, which checks none)" << endl;
	exit(1);
}

static const char CIGAR_CHARS[] = "MIDNSHP=X";

// .gz output: text is collected and written as gzip members compressed in parallel by libseeksv_host (ssvh_gz_append)
struct GzOut {
	string path, buf;
	bool started = false;
	bool open(const string &p)
	{
		path = p;
		FILE *f = fopen(p.c_str(), "wb"); // like ogzstream: fail early when the file cannot be created
		if (!f) return false;
		fclose(f);
		return true;
	}
	void flush(bool final)
	{
		if (buf.empty() && (started || !final)) return;
		if (ssvh_gz_append(path.c_str(), buf.data(), buf.size(), started ? 1 : 0) != 0) die(string("[seeksv] ") + ssvh_last_error());
		started = true;
		buf.clear();
	}
	void write(const string &s) { buf += s; if (buf.size() >= ((size_t)256 << 20)) flush(false); }
	void write_parts(const vector<string> &parts) // large, already formatted pieces: compressed from where they lie
	{
		flush(false);
		vector<const char *> ptr; vector<size_t> len;
		for (auto &s : parts) if (!s.empty()) { ptr.push_back(s.data()); len.push_back(s.size()); }
		if (ptr.empty()) return;
		if (ssvh_gz_append_v(path.c_str(), ptr.data(), len.data(), (int)ptr.size(), started ? 1 : 0) != 0) die(string("[seeksv] ") + ssvh_last_error());
		started = true;
	}
	void close() { flush(true); }
};

// ---------------------------------------------------------------------------------------------------------------------
// where the record batches come from: the host reader (inflate + decode on the host threads, batches uploaded by the kernels' H2D
// staging), or -Z: the compressed BGZF blocks go to the GPU as they are and are inflated and decoded there (ssv_bamdec_*); a reader
// thread fills one pinned staging buffer while the chunk in the other one is being decoded.
// ---------------------------------------------------------------------------------------------------------------------

// Page-locked batch arrays for the host reader (copies to the GPU out of them run asynchronously).
static void *pinned_alloc(size_t n) { void *p = nullptr; return ssv_host_alloc(n, &p) == SSV_OK ? p : nullptr; }
static void pinned_release(void *p) { ssv_host_free(p); }
static void use_pinned_batches(ssvh_bam *b)
{
	ssvh_bam_set_allocator(b, pinned_alloc, pinned_release);
	ssvh_bam_set_readahead(b, 1);
}

// The batches of a host reader (read-ahead on: three sets of arrays) through a context: batch k+1 is decoded, announced and on its way to the
// GPU (ssv_batch_prefetch, the upload stream) while batch k is scanned; `side` sees every batch right after it was read (the reader's
// per-batch side channel - unmapped reads - is valid only then), `scan` gets them in order.  Returns "" or the error text.
// records per batch of the host reader (SSV_HOST_BATCH_RECORDS: tests cut small files into many batches)
static int64_t host_batch_records()
{
	static const int64_t n = [] { const char *e = getenv("SSV_HOST_BATCH_RECORDS"); return e ? std::max<int64_t>(1, atoll(e)) : (int64_t)1 << 22; }();
	return n;
}

template <class Side, class Scan> static string pump_host_batches(ssvh_bam *rb, ssv_ctx *ctx, int keep_all_seq, Side side, Scan scan)
{
	ssv_batch_t b[2];
	auto read = [&](ssv_batch_t *out) -> string {
		if (ssvh_bam_read_batch(rb, host_batch_records(), keep_all_seq, out) != 0) return string("[seeksv] ") + ssvh_last_error();
		if (out->n == 0) return "";
		side(*out);
		if (ssv_batch_prefetch(ctx, out) != SSV_OK) return string("[seeksv] ") + ssv_last_error(ctx);
		return "";
	};
	string err = read(&b[0]);
	for (int k = 0; err.empty() && b[k & 1].n != 0; ++k) {
		err = read(&b[(k + 1) & 1]);
		if (!err.empty()) break;
		err = scan(b[k & 1]);
	}
	return err;
}

// `seeksv run`: one process, one context, one decode of the BAM.  While getclip scans the file, every decoded batch is copied into device
// memory of its own (ssv_batch_retain: hot columns, one 64-byte line per record, CIGARs, bases of the soft-clipped reads: 80 B/record -
// a 30x genome is 50 GB of the GPU's 288); the getsv passes of the same process then find the file's records there instead of reading,
// inflating and decoding the file a second and third time (the reference does: cluster.cpp:48, getsv.cpp:1067, bam2depth.h:29).
struct ResidentBam {
	string path;                 // the file these are the records of
	ssv_ctx *ctx = nullptr;      // the context every command of the process shares
	bool collect = false;        // getclip keeps what it decodes
	vector<ssv_batch_t> batches; // in file order
	// what getclip wrote, still in memory for the steps that follow (the files are written all the same: they are outputs)
	string clip_path, fq_path;   // prefix.clip.gz, prefix.clip.fq.gz
	// their decompressed contents, in the pieces the formatting threads made (whole rows / whole records each; never glued together: appending 2.5 GB to
	// one string on one thread was a second of `seeksv run`) - and, since round 6, what the formatter knew while it wrote them: every row as views into
	// its text (the junction stage parsed 5.5 M rows back out of it: 0.35 s) and where every FASTQ record's sequence and qualities lie (the aligner
	// step cut the text into lines again: 0.2 s).  `clean`: every row is what the text parser would make of it (nine non-empty fields, no white space).
	struct FqRef { uint32_t seq_off, seq_len, qual_off, qual_len; };
	struct Piece { string rows, fq; vector<seeksv::ClipRow> parsed; vector<FqRef> reads; bool clean = true; };
	std::deque<Piece> pieces;    // (a deque: a piece never moves, the views stay good)
	std::function<void(const vector<Piece *> &)> on_pass; // called by getclip's output thread with the pieces of the pass it has just written
	bool all_clean() const { for (const Piece &p : pieces) if (!p.clean) return false; return true; }
	vector<seeksv::TextView> row_views() const { vector<seeksv::TextView> v; for (const Piece &p : pieces) v.push_back(seeksv::TextView{p.rows.data(), p.rows.size()}); return v; }
	// what the aligner step made of the clipped sequences (clip.bam is written all the same); the read names point into the pieces' FASTQ text
	struct Aligned {
		string bam_path;
		const struct AlignedRecords *rec = nullptr; // (the run's aligner thread owns them)
		vector<string> names;
	} aln;
	std::thread bam_writer;      // clip.bam of `seeksv run`, written beside getsv
};
static ResidentBam g_resident;

static void release_ctx(ssv_ctx *ctx)
{
	if (ctx == g_resident.ctx) return; // shared by the commands of `seeksv run`, which lets go of it at its end
	if (kCleanExit) ssv_ctx_destroy(ctx);
	else ssv_sync(ctx);
}

// The HIP runtime takes 0.1-0.3 s to come up (first call of the process) - longer than a command spends on a 6 GB file's kernels.  main() starts the
// context on a thread of its own before the command parses its arguments; opening files, the junction stage of getsv and the reader's first chunks
// run beside it, and acquire_ctx joins it where the first kernel is needed.
struct EarlyCtx {
	std::thread th;
	ssv_ctx *ctx = nullptr;
	int device = -1, rc = SSV_OK;
	string err;
	bool started = false, taken = false;
};
static EarlyCtx g_early;
static void early_ctx_join() { if (g_early.th.joinable()) g_early.th.join(); }
static void early_ctx_start(int device)
{
	if (g_early.started) return;
	g_early.started = true; g_early.device = device;
	atexit(early_ctx_join); // (usage errors leave through exit(): not while a thread is inside the runtime's start-up)
	g_early.th = std::thread([] {
		g_early.rc = ssv_ctx_create(g_early.device, &g_early.ctx);
		if (g_early.rc != SSV_OK) g_early.err = ssv_last_error(nullptr);
	});
}

static ssv_ctx *acquire_ctx(int device)
{
	if (g_resident.ctx) return g_resident.ctx;
	if (g_early.started && !g_early.taken) {
		early_ctx_join();
		g_early.taken = true;
		if (g_early.device == device) {
			if (g_early.rc != SSV_OK) die(string("[seeksv] ") + g_early.err);
			return g_early.ctx;
		}
		if (g_early.ctx) ssv_ctx_destroy(g_early.ctx); // (another -G than the one main() saw)
	}
	ssv_ctx *ctx = nullptr;
	if (ssv_ctx_create(device, &ctx) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(nullptr));
	return ctx;
}

static const bool kTimingChunks = [] { const char *e = getenv("SSV_TIMING"); return e ? atoi(e) : 0; }() >= 2; // SSV_TIMING=2: per-chunk detail

// Staging memory for the chunks of a file in copy mode.  hipHostMalloc hands out page-locked memory at 0.25 s/GB (it allocates, clears and locks on
// one thread, and other HIP calls of the process queue behind it meanwhile): three 0.6-1.5 GB buffers were 0.5 s before the first kernel of a
// command.  Here a buffer is plain anonymous memory; the reader threads' pread calls fault its pages in on all cores while they fill it for the
// first time, and only then are the pages that hold bytes page-locked (ssv_host_register: 0.03 s/GB).  Buffers go back to a process-wide pool when a
// source closes (`seeksv run`, the ranks' halo and own passes: the next source finds them locked already).
struct StageBuf {
	uint8_t *p = nullptr;
	size_t cap = 0, locked = 0; // locked: bytes from p on that are page-locked
	void *map_base = nullptr; size_t map_len = 0; // the mapping p lies in (p is 2 MB aligned)
	int near_device = 0;                          // the GPU its pages should lie beside
	static size_t page() { static const size_t v = (size_t)sysconf(_SC_PAGESIZE); return v; }
	bool reserve(size_t want)
	{
		if (cap >= want) return true;
		release();
		// on transparent huge pages where the kernel grants them (2 MB aligned, madvise): 512 times fewer pages to fault in, to lock and - when the process ends -
		// to unlock (tools/exit_cost.cpp: 4 GB of locked 4 KB pages cost 0.4 s between _exit and the caller's wait() returning, huge pages 0.001 s)
		const size_t huge = (size_t)2 << 20;
		const size_t n = (want + huge - 1) & ~(huge - 1);
		void *m = mmap(nullptr, n + huge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
		if (m == MAP_FAILED) return false;
		map_base = m; map_len = n + huge;
		p = reinterpret_cast<uint8_t *>(((uintptr_t)m + huge - 1) & ~(uintptr_t)(huge - 1)); cap = n;
		(void)madvise(p, n, MADV_HUGEPAGE);
		ssv_host_bind_near(p, n, near_device); // (the reader's pread threads run on any CPU: the pages still land beside the GPU that fetches them)
		return true;
	}
	// the first `filled` bytes have just been written: make sure they are page-locked (a buffer that is mostly full is locked whole, its tail
	// touched on a few threads first, so that chunks of slightly different sizes do not lock again and again)
	bool lock(size_t filled)
	{
		if (filled <= locked) return true;
		if (locked) { ssv_host_unregister(p); locked = 0; }
		size_t n = (filled + page() - 1) & ~(page() - 1);
		if (filled * 2 > cap) {
			const int nt = 8;
			vector<std::thread> th;
			for (int t = 0; t < nt; ++t) th.emplace_back([&, t] {
				const size_t lo = n + (cap - n) / page() * (size_t)t / nt * page(), hi = n + (cap - n) / page() * (size_t)(t + 1) / nt * page();
				for (size_t o = lo; o < hi; o += page()) p[o] = 0;
			});
			for (auto &x : th) x.join();
			n = cap;
		}
		if (ssv_host_register(p, n) != SSV_OK) return false; // (the copies out of it then go through the runtime's own staging: slower, still right)
		locked = n;
		return true;
	}
	void release()
	{
		if (!p) return;
		if (locked) ssv_host_unregister(p);
		munmap(map_base, map_len);
		p = nullptr; cap = locked = 0; map_base = nullptr; map_len = 0;
	}
};
static int g_near_device = 0; // the command's GPU (-G; main() sets it): where page-locked staging memory should lie
static std::mutex g_stage_pool_mu;
static vector<StageBuf> g_stage_pool;
static StageBuf stage_from_pool(size_t want)
{
	std::lock_guard<std::mutex> lk(g_stage_pool_mu);
	for (size_t k = 0; k < g_stage_pool.size(); ++k)
		if (g_stage_pool[k].cap >= want) { StageBuf b = g_stage_pool[k]; g_stage_pool.erase(g_stage_pool.begin() + (long)k); return b; }
	return StageBuf();
}
static void stage_to_pool(StageBuf &b)
{
	if (!b.p) return;
	std::lock_guard<std::mutex> lk(g_stage_pool_mu);
	if (g_stage_pool.size() < 8) g_stage_pool.push_back(b); else b.release();
	b = StageBuf();
}

struct BatchSource {
	ssvh_bam *bam = nullptr;
	ssv_ctx *ctx = nullptr;
	bool on_device = false;
	bool resident = false;   // the records are in HBM already (g_resident)
	size_t resident_next = 0;
	// ---- device mode: a reader thread runs up to three chunks ahead of the decoder ----
	// A chunk = whole BGZF blocks, the file's bytes as they are.  Two ways to get them to the GPU:
	//   copy (default): all host threads pread the chunk into one of three staging buffers (StageBuf above: locked after their first fill);
	//   map  (SSV_READER=map): the file is mapped and the chunk's pages are page-locked where they lie in the page cache (ssv_host_register), the GPU's
	//         DMA engines fetch them there and no CPU copies the file.  Measured on a 11.8 GB file (profiles/r04_cli_reader_modes.txt): locking a fresh
	//         mapping's pages costs 0.04-0.1 s/GB on the one thread the driver lets do it (page-table entries for every 4 KB page first), i.e. 25-60 ms
	//         per 0.6 GB chunk where the copy takes 14 ms on 16 threads, and unmapping the file at the end 0.19 s: slower, kept for boxes with few CPUs.
	// Either way chunk k+1 is announced to the upload stream (ssv_bamdec_prefetch) before chunk k is decoded, so its bytes cross PCIe under chunk k's
	// kernels, while the reader prepares chunk k+2.
	static constexpr int NS = 3;
	struct Slot {
		StageBuf stage;                                       // copy mode: the slot's staging buffer
		void *reg_base = nullptr; size_t reg_bytes = 0;       // map mode: the page range that is locked for this slot's chunk
		const uint8_t *data = nullptr; size_t n_bytes = 0;    // the chunk
		vector<ssv_bgzf_block> blocks; int64_t n_blocks = 0;
		uint64_t limit = UINT64_MAX;                          // ssvh_bam_raw_limit: where a range of records ends inside the chunk
		bool last = false;                                    // nothing but the end-of-file block can follow
		double t_fill = 0;
		string err;
	} slot[NS];
	size_t stage_bytes = 0, first_bytes = 0;
	uint64_t chunk_inflated = 0;
	bool map_mode = true;
	std::thread reader;
	std::mutex mu;
	std::condition_variable cv;
	int64_t produced = 0, released = 0, cur = 0; // chunks the reader has finished / the decoder has let go of / next chunk to decode
	bool reader_exit = false, stop_reader = false;
	int64_t announced = -1;                      // highest chunk handed to ssv_bamdec_prefetch
	bool at_end = false, end_pending = false;
	uint64_t file_bytes = 0;
	ssv_bamdec_info info{};

	// a run of records [start, end) of the file instead of all of it (virtual offsets of ssvh_bam_partition); before open()
	bool ranged = false;
	uint64_t r_start_coff = 0, r_end_coff = 0;
	uint32_t r_start_uoff = 0, r_end_uoff = 0;
	void set_range(uint64_t sc, uint32_t su, uint64_t ec, uint32_t eu) { ranged = true; r_start_coff = sc; r_start_uoff = su; r_end_coff = ec; r_end_uoff = eu; }
	int32_t r_prev_tid = 0; // contig of the last mapped-pair record before the range

	void open(const string &path, ssv_ctx *c, bool device_inflate, const char *open_error) { open(path, [c] { return c; }, device_inflate, open_error); }
	// get_ctx: asked for the context as late as possible - the file is opened and the reader thread is on its first chunks before (the context of a
	// fresh process may still be coming up, acquire_ctx)
	void open(const string &path, const std::function<ssv_ctx *()> &get_ctx, bool device_inflate, const char *open_error)
	{
		on_device = device_inflate;
		if (ssvh_bam_open(path.c_str(), &bam) != 0) die(open_error);
		if (g_resident.ctx || !on_device) ctx = get_ctx();
		ssv_ctx *const c = ctx;
		if (!ranged && !g_resident.collect && c && g_resident.ctx == c && !g_resident.batches.empty() && path == g_resident.path) { resident = true; on_device = true; return; }
		if (!on_device) {
			if (ranged && ssvh_bam_set_range(bam, r_start_coff, r_start_uoff, r_end_coff, r_end_uoff) != 0) die(string("[seeksv] ") + ssvh_last_error());
			use_pinned_batches(bam);
			return;
		}
		const char *e1 = getenv("SSV_CHUNK_INFLATED_MB"), *e2 = getenv("SSV_STAGE_MB"), *e4 = getenv("SSV_READER");
		map_mode = e4 && !strcmp(e4, "map");
		{
			FILE *f = fopen(path.c_str(), "rb");
			if (f) { fseek(f, 0, SEEK_END); file_bytes = (uint64_t)ftell(f); fclose(f); }
			// What a command pays once grows with the chunk size: ~3.8 bytes of device memory per inflated byte and three page-locked staging buffers, to set
			// up AND to give back (the kernel takes 0.16-0.25 s to tear a process with 8 GB of device buffers and 2.3 GB of locked pages down, after exit and
			// before the caller's wait returns).  Since round 4 both inflate passes are a wavefront per block and their rate does not depend on the blocks in
			// flight: chunks of 1 GB inflated (0.3 GB of file) whatever the file's size (round 3: up to 4 GB, which the lane-per-block kernels needed; until late in
			// round 4 2 GB above 16 GB of file - measured again on the 23.7 GB and 47.4 GB files: 1 GB chunks take 9-11 % off both commands, half of it after exit).
			chunk_inflated = (uint64_t)(e1 ? atoll(e1) : 1024) << 20;
			stage_bytes = (size_t)(e2 ? atoll(e2) : 384) << 20;
			if (file_bytes && file_bytes + 65536 < stage_bytes) stage_bytes = (size_t)file_bytes + 65536;
		}
		uint64_t first = 0;
		if (ranged) { if (ssvh_bam_raw_begin_range(bam, r_start_coff, r_start_uoff, r_end_coff, r_end_uoff, &first) != 0) die(string("[seeksv] ") + ssvh_last_error()); file_bytes = 0; }
		else if (ssvh_bam_raw_begin(bam, &first) != 0) die(string("[seeksv] ") + ssvh_last_error());
		// The file's first chunk is the one nothing overlaps with: a small one (128 MB: read and locked in ~10 ms), so that the GPU starts early
		first_bytes = std::min(stage_bytes, first_bytes_default());
		reader = std::thread([this] { reader_main(); }); // (reads, and locks its buffers: needs no context)
		if (!ctx) ctx = get_ctx();
		if (ssv_bamdec_begin(ctx, ssvh_bam_n_targets(bam), first) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(ctx));
		if (ssv_bamdec_target_lens(ctx, ssvh_bam_target_lens(bam)) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(ctx));
		if (g_verify_crc) ssv_bamdec_verify_crc(ctx, 1);
		if (ranged) ssv_bamdec_prev_tid(ctx, r_prev_tid);
		if (!file_bytes || file_bytes > first_bytes_default()) ssv_bamdec_expect(ctx, chunk_inflated);
	}
	static size_t page_size() { static const size_t p = (size_t)sysconf(_SC_PAGESIZE); return p; }
	static size_t first_bytes_default() { return (size_t)128 << 20; }
	int near_device = g_near_device; // the GPU that fetches this source's chunks: their staging pages are asked for on its NUMA node
	bool take_stage(Slot &S, size_t want)
	{
		if (S.stage.cap >= want) return true;
		stage_to_pool(S.stage);
		S.stage = stage_from_pool(want);
		S.stage.near_device = near_device;
		return S.stage.reserve(want);
	}
	// the reader thread: chunk k into slot k % NS as soon as the decoder has let go of chunk k - NS
	void reader_main()
	{
		(void)pthread_setname_np(pthread_self(), "ssv-chunks");
		for (int64_t k = 0;; ++k) {
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&] { return stop_reader || k - released < NS; });
				if (stop_reader) break;
			}
			Slot &S = slot[k % NS];
			const auto t0 = std::chrono::steady_clock::now();
			if (S.reg_base) { ssv_host_unregister(S.reg_base); S.reg_base = nullptr; } // (the chunk that was here has been decoded)
			S.err.clear(); S.n_blocks = 0; S.n_bytes = 0; S.data = nullptr; S.limit = UINT64_MAX; S.last = false;
			const size_t want = k == 0 ? first_bytes : stage_bytes;
			if (S.blocks.empty()) S.blocks.resize((size_t)(std::min<uint64_t>(chunk_inflated, file_bytes ? file_bytes * 64 : chunk_inflated) >> 12) + 1024); // blocks are <= 64 KB but may be much smaller
			bool filled = false;
			if (map_mode) {
				const void *ptr = nullptr;
				const int rc = ssvh_bam_map_blocks(bam, want, chunk_inflated, S.blocks.data(), (int64_t)S.blocks.size(), &S.n_blocks, &ptr, &S.n_bytes);
				if (rc == -2) map_mode = false; // the file cannot be mapped: staging buffers from here on (nothing has been consumed)
				else if (rc != 0) S.err = ssvh_last_error();
				else {
					filled = true;
					S.data = static_cast<const uint8_t *>(ptr);
					if (S.n_bytes) {
						const uintptr_t lo = reinterpret_cast<uintptr_t>(ptr) & ~(uintptr_t)(page_size() - 1);
						const uintptr_t hi = (reinterpret_cast<uintptr_t>(ptr) + S.n_bytes + page_size() - 1) & ~(uintptr_t)(page_size() - 1);
						if (ssv_host_register(reinterpret_cast<void *>(lo), (size_t)(hi - lo)) == SSV_OK) { S.reg_base = reinterpret_cast<void *>(lo); S.reg_bytes = (size_t)(hi - lo); }
						else {
							// the runtime will not lock these pages (a file system it cannot pin, a memory-lock limit): this chunk is copied out of the
							// mapping, the following ones are read the other way
							map_mode = false;
							if (!take_stage(S, std::max(S.n_bytes + 64, stage_bytes))) S.err = "out of memory (staging buffer)";
							else { memcpy(S.stage.p, ptr, S.n_bytes); S.stage.lock(S.n_bytes); S.data = S.stage.p; }
						}
					}
				}
			}
			if (!filled && S.err.empty()) {
				// (the chunk that was in this buffer has been decoded: its bytes are on the device)
				if (!take_stage(S, stage_bytes)) S.err = "out of memory (staging buffer)";
				if (S.err.empty() && ssvh_bam_read_blocks(bam, S.stage.p, std::min(S.stage.cap, want), chunk_inflated, S.blocks.data(), (int64_t)S.blocks.size(), &S.n_blocks, &S.n_bytes) != 0) S.err = ssvh_last_error();
				if (S.err.empty() && S.n_bytes) S.stage.lock(S.n_bytes);
				S.data = S.stage.p;
			}
			if (S.err.empty()) ssvh_bam_raw_limit(bam, &S.limit);
			consumed_bytes += S.n_bytes;
			// (the chunk holds the file's bytes as they are, block headers and trailers included) when nothing but the 28-byte end-of-file block can be
			// left, no further chunk is read
			S.last = S.n_blocks == 0 || (file_bytes && consumed_bytes + 28 >= file_bytes);
			S.t_fill = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			const bool done = S.last || !S.err.empty();
			{
				std::lock_guard<std::mutex> lk(mu);
				produced = k + 1;
				if (done) reader_exit = true;
			}
			cv.notify_all();
			if (done) break;
		}
	}
	uint64_t consumed_bytes = 0; // (reader thread)
	// the next batch (valid until the following call); false at the end of the file
	bool next(ssv_batch_t *b, int keep_all_seq)
	{
		if (resident) {
			if (resident_next >= g_resident.batches.size()) return false;
			*b = g_resident.batches[resident_next++];
			return true;
		}
		if (!on_device) {
			if (ssvh_bam_read_batch(bam, host_batch_records(), keep_all_seq, b) != 0) die(string("[seeksv] ") + ssvh_last_error());
			return b->n != 0;
		}
		for (;;) {
			if (at_end) return false;
			if (end_pending) { // the last chunk has been handed out: tell the decoder the input is over (an unfinished record is an error)
				ssv_batch_t none;
				if (ssv_bamdec_decode(ctx, nullptr, 0, nullptr, 0, keep_all_seq, &none) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(ctx));
				at_end = true;
				return false;
			}
			const auto tw0 = std::chrono::steady_clock::now();
			bool next_ready = false;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&] { return produced > cur; });
				next_ready = produced > cur + 1;
			}
			const auto tw1 = std::chrono::steady_clock::now();
			Slot &S = slot[cur % NS];
			if (!S.err.empty()) die("[seeksv] " + S.err);
			const bool last_chunk = S.last;
			// the chunk behind this one starts its way over PCIe now, under this chunk's kernels
			if (!last_chunk && next_ready && announced < cur + 1) {
				Slot &N = slot[(cur + 1) % NS];
				if (N.err.empty() && N.n_blocks > 0 && ssv_bamdec_prefetch(ctx, N.data, N.n_bytes) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(ctx));
				announced = cur + 1;
			}
			if (S.limit != UINT64_MAX) ssv_bamdec_limit(ctx, S.limit);
			if (ssv_bamdec_decode(ctx, S.data, S.n_bytes, S.blocks.data(), S.n_blocks, keep_all_seq, b) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(ctx));
			if (kTimingChunks) cerr << "[timing] (chunk of " << S.n_bytes << " bytes, " << S.n_blocks << " blocks, " << (S.reg_base ? "page-locked in the mapping" : "copied to staging") << " in " << S.t_fill
			                        << " s: waited " << std::chrono::duration<double>(tw1 - tw0).count() << " s for the reader, " << (announced >= cur && cur > 0 ? "announced ahead, " : "") << "decoded in "
			                        << std::chrono::duration<double>(std::chrono::steady_clock::now() - tw1).count() << " s)" << endl;
			const int64_t nb = S.n_blocks;
			{
				std::lock_guard<std::mutex> lk(mu);
				released = ++cur; // the chunk's bytes are on the device and decoded: its slot is the reader's again
			}
			cv.notify_all();
			if (nb == 0) { at_end = true; return false; }
			if (last_chunk) end_pending = true;
			ssv_bamdec_last(ctx, &info);
			if (b->n) return true; // (a chunk can hold only the middle of one huge record)
		}
	}
	// every batch through side(b) (right after it was read) and scan(b); with the host reader the copy of the next batch runs ahead
	template <class Side, class Scan> void pump(int keep_all_seq, Side side, Scan scan)
	{
		if (!on_device) {
			const string err = pump_host_batches(bam, ctx, keep_all_seq, side, [&](const ssv_batch_t &b) { scan(b); return string(); });
			if (!err.empty()) die(err);
			return;
		}
		ssv_batch_t b;
		while (next(&b, keep_all_seq)) { side(b); scan(b); }
	}
	// UNMAP|MUNMAP records of the current batch as they lie in the BAM stream (block_size prefixed), in order; valid while the NEXT batch is read
	void unmapped_raw(const uint8_t **raw, size_t *bytes)
	{
		if (!on_device) { ssvh_bam_unmapped_raw(bam, raw, bytes); return; }
		*raw = info.unmapped_raw; *bytes = (size_t)info.unmapped_bytes;
	}
	// ... one by one, decoded
	template <class F> void for_each_unmapped(F fn)
	{
		const char *qname, *seq, *qual; int is_read1;
		if (!on_device) {
			for (int64_t k = 0, nu = ssvh_bam_unmapped_count(bam); k < nu; ++k) { ssvh_bam_unmapped_get(bam, k, &qname, &seq, &qual, &is_read1); fn(qname, seq, qual, is_read1); }
			return;
		}
		for (size_t off = 0, nxt; (nxt = ssvh_raw_record_fastq(info.unmapped_raw, info.unmapped_bytes, off, &qname, &seq, &qual, &is_read1)) != 0; off = nxt) fn(qname, seq, qual, is_read1);
	}
	// the contig changes among the other records of the current batch (the flush sequence of clip_reads.h:423-438): fn(record index, new contig)
	template <class F> void contig_changes(const ssv_batch_t &b, int32_t &last_tid, F fn)
	{
		if (!on_device) {
			for (int64_t i = 0; i < b.n; ++i) {
				if (b.flag[i] & (4 | 8)) continue;
				if (b.tid[i] != last_tid) { last_tid = b.tid[i]; fn(i, last_tid); }
			}
			return;
		}
		for (uint32_t k = 0; k < info.n_tid_runs; ++k) { last_tid = info.tid_run_tid[k]; fn((int64_t)info.tid_run_index[k], last_tid); }
	}
	void contig_runs(const ssv_batch_t &b, int32_t &last_tid, vector<int32_t> &run_tids)
	{
		int32_t prev = last_tid;
		contig_changes(b, last_tid, [&](int64_t, int32_t tid) { run_tids.push_back(prev); prev = tid; });
	}
	void close()
	{
		if (reader.joinable()) {
			{ std::lock_guard<std::mutex> lk(mu); stop_reader = true; }
			cv.notify_all();
			reader.join();
		}
		if (announced >= cur && ctx) ssv_bamdec_prefetch_drop(ctx); // a chunk on its way that nobody will decode: its copy must have left the host pages
		for (Slot &S : slot) { if (S.reg_base) { ssv_host_unregister(S.reg_base); S.reg_base = nullptr; } stage_to_pool(S.stage); }
		if (bam) ssvh_bam_close(bam);
		bam = nullptr;
	}
};

static bool device_inflate_default() { const char *e = getenv("SSV_DEVICE_INFLATE"); return e && atoi(e) != 0; }

static vector<int> parse_devices(const char *s) // "-G 0,2,5"
{
	vector<int> d;
	for (const char *p = s; *p;) { d.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
	return d;
}

// GPU of rank r: the -G list (cycled) or the machine's GPUs in order (cycled: more ranks than GPUs is allowed - a one-GPU box runs every
// rank on its one GPU, one after the other in effect)
static int device_of_rank(int r, const vector<int> &devices)
{
	if (!devices.empty()) return devices[(size_t)r % devices.size()];
	const int n = ssv_device_count();
	return n > 0 ? r % n : 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// getclip
// ---------------------------------------------------------------------------------------------------------------------

// DisplaySClipReadsAndClipFq (clip_reads.h:300-345) for the clusters [k0, k1) of a table: their clip.gz rows and clip.fq records
// (the cluster table crosses PCIe in the compact format 3, include/seeksv_hip.h; the text written is the ASCII table's)
static int table_format_default() { return 3; }

// where a row's fields lie in the text (offsets: the strings still grow)
struct RowAt { uint32_t row, chr_len, cigar, cigar_len, aligned, aligned_len, clipped, clipped_len, cqual, cqual_len, fq; int32_t pos, support; char side; };
static void format_clusters(const ssv_cluster_table &t, ssvh_bam *bam, int64_t k0, int64_t k1, string &row, string &fq, vector<RowAt> *index = nullptr)
{
		row.reserve(row.size() + (size_t)(k1 - k0) * 480); fq.reserve(fq.size() + (size_t)(k1 - k0) * 200);
		if (row.empty()) ssv::thp_advise(row.data(), row.capacity()); // (tens of MB of text per thread and pass: on 2 MB pages where the kernel grants them, thp.h)
		if (fq.empty()) ssv::thp_advise(fq.data(), fq.capacity());
		char num[16];
		// (decimal digits by hand: three snprintf calls a row were a tenth of the formatter's time on a whole-genome sample's 5.5 M rows)
		auto put_int = [&num](long long v) -> size_t {
			char tmp[24]; size_t n = 0;
			const bool neg = v < 0;
			unsigned long long u = neg ? 0ull - (unsigned long long)v : (unsigned long long)v;
			do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
			size_t o = 0;
			if (neg) num[o++] = '-';
			while (n) num[o++] = tmp[--n];
			return o;
		};
		string seqbuf;
		// a byte of the 2-bit base stream -> its four characters; of the 4-bit stream -> its two
		static const struct BaseTabs {
			uint32_t b2[256]; uint16_t b4[256];
			BaseTabs() {
				for (unsigned v = 0; v < 256; ++v) {
					char c[4];
					for (int i = 0; i < 4; ++i) c[i] = "ACGT"[(v >> (2 * i)) & 3];
					memcpy(&b2[v], c, 4);
					const char d[2] = {"=ACMGRSVTWYHKDBN"[v & 15], "=ACMGRSVTWYHKDBN"[v >> 4]};
					memcpy(&b4[v], d, 2);
				}
			}
		} base_tabs;
		std::vector<uint32_t> group_lut;   // grouped qualities: a group's number -> its (up to three) quality characters, as the low bytes of a dword
		const ssv_cluster_table *group_lut_for = nullptr;
		for (int64_t k = k0; k < k1; ++k) {
			const char *name = ssvh_bam_target_name(bam, t.tid[k]);
			const uint8_t *s = t.str + t.str_off[k];
			const size_t ll = (size_t)t.left_len[k], lr = (size_t)t.right_len[k];
			const char *sl, *ql, *sr, *qr;
			if (t.format == 3) { // compact: [base stream | quality stream] over seq_left + seq_right, whole 32-bit words each (seeksv_hip.h)
				const size_t n = ll + lr, W = (size_t)t.qual_bits, BBITS = (size_t)t.base_bits, QG = t.qual_group > 1 ? (size_t)t.qual_group : 1;
				const uint8_t *bs = s, *qs = s + 4 * ((n * BBITS + 31) / 32);
				seqbuf.resize(2 * n + 16); // (+ room for the whole dwords the table-driven loops write behind the last character)
				char *const sb = &seqbuf[0];
				if (BBITS == 2) {
					for (size_t i = 0, b = 0; i < n; i += 4, ++b) memcpy(sb + i, &base_tabs.b2[bs[b]], 4); // (the stream is whole dwords: the last byte is there)
					// the bases that are not A/C/G/T (sorted list; cluster << 28 | base index << 4 | code)
					const uint64_t *e0 = t.base_exc, *e1 = t.base_exc + t.n_base_exc;
					if (e0 != e1) for (const uint64_t *e = std::lower_bound(e0, e1, (uint64_t)k << 28); e < e1 && (*e >> 28) == (uint64_t)k; ++e) sb[(size_t)((*e >> 4) & 0xffffff)] = "=ACMGRSVTWYHKDBN"[*e & 15];
				} else for (size_t i = 0, b = 0; i < n; i += 2, ++b) memcpy(sb + i, &base_tabs.b4[bs[b]], 2);
				sl = sb; sr = sb + ll;
				if (W == 8) { ql = (const char *)qs; qr = ql + ll; }
				else if (QG > 1) { // groups of QG qualities: one number of W bits, its digits (radix = the alphabet's size) are the alphabet indices
					if (!group_lut_for || group_lut_for != &t) {
						unsigned radix = 0;
						while (radix < sizeof(t.qual_alphabet) && t.qual_alphabet[radix]) ++radix;
						group_lut.assign((size_t)1 << W, 0u);
						for (unsigned code = 0; code < (1u << W); ++code) { unsigned v = code; char c[4] = {0, 0, 0, 0}; for (size_t j = 0; j < 3; ++j) { c[j] = (char)t.qual_alphabet[radix ? v % radix : 0]; if (radix) v /= radix; } memcpy(&group_lut[code], c, 4); }
						group_lut_for = &t;
					}
					char *dq = sb + n;
					const unsigned mask = (1u << W) - 1u;
					const size_t ng = (n + QG - 1) / QG, qbytes = 4 * ((ng * W + 31) / 32);
					// (a group of up to 11 bits lies in at most three bytes: one unaligned dword from its first byte - read with bounds only for the stream's last three bytes)
					for (size_t g = 0; g < ng; ++g) {
						const size_t bit = g * W, b = bit >> 3;
						uint32_t w;
						if (b + 4 <= qbytes) memcpy(&w, qs + b, 4);
						else { w = 0; for (size_t x = b; x < qbytes; ++x) w |= (uint32_t)qs[x] << (8 * (x - b)); }
						const uint32_t chars = group_lut[(w >> (bit & 7)) & mask];
						memcpy(dq + g * QG, &chars, 4); // (the next group's characters overwrite the fourth byte; the last group's lands in the buffer's slack)
					}
					ql = dq; qr = dq + ll;
				} else {
					const unsigned mask = (1u << W) - 1u;
					char *dq = sb + n;
					const size_t qbytes = 4 * ((n * W + 31) / 32);
					for (size_t i = 0; i < n; ++i) { const size_t b = (i * W) >> 3; dq[i] = (char)t.qual_alphabet[((qs[b] | (b + 1 < qbytes ? (unsigned)qs[b + 1] << 8 : 0u)) >> ((i * W) & 7)) & mask]; }
					ql = dq; qr = dq + ll;
				}
			} else { sl = (const char *)s; ql = sl + ll; sr = sl + 2 * ll; qr = sl + 2 * ll + lr; }
			size_t lql = ll, lqr = lr;
			if (t.qual_missing[k]) { ql = qr = "*"; lql = lqr = 1; }
			RowAt at;
			at.row = (uint32_t)row.size(); at.fq = (uint32_t)fq.size(); at.pos = t.pos[k]; at.support = t.support[k]; at.side = (char)t.side[k];
			row += name ? name : ""; row += '\t';
			at.chr_len = (uint32_t)row.size() - 1 - at.row;
			row.append(num, put_int(t.pos[k])); row += '\t'; row += (char)t.side[k]; row += '\t';
			at.cigar = (uint32_t)row.size();
			for (int q = 0; q < t.n_cigar[k]; ++q) {
				const uint32_t op = t.cigar ? t.cigar[t.cigar_off[k] + q] : (uint32_t)reinterpret_cast<const uint16_t *>(t.c_cigar)[t.cigar_off[k] + q]; // (compact table: 16 bits an operation)
				if ((op & 15) == 4 || (op & 15) == 5) continue;
				row.append(num, put_int((long long)(op >> 4))); row += CIGAR_CHARS[op & 15];
			}
			at.cigar_len = (uint32_t)row.size() - at.cigar;
			row += '\t';
			at.aligned = (uint32_t)row.size();
			if (t.side[k] == '5') { at.aligned_len = (uint32_t)lr; at.clipped = at.aligned + (uint32_t)(lr + 1 + lqr + 1); at.clipped_len = (uint32_t)ll; at.cqual_len = (uint32_t)lql; }
			else { at.aligned_len = (uint32_t)ll; at.clipped = at.aligned + (uint32_t)(ll + 1 + lql + 1); at.clipped_len = (uint32_t)lr; at.cqual_len = (uint32_t)lqr; }
			at.cqual = at.clipped + at.clipped_len + 1;
			if (index) index->push_back(at);
			if (t.side[k] == '5') {
				row.append(sr, lr); row += '\t'; row.append(qr, lqr); row += '\t'; row.append(sl, ll); row += '\t'; row.append(ql, lql);
				fq += '@'; fq.append(sl, ll); fq += '\n'; fq.append(sl, ll); fq += "\n+\n"; fq.append(ql, lql); fq += '\n';
			} else {
				row.append(sl, ll); row += '\t'; row.append(ql, lql); row += '\t'; row.append(sr, lr); row += '\t'; row.append(qr, lqr);
				fq += '@'; fq.append(sr, lr); fq += '\n'; fq.append(sr, lr); fq += "\n+\n"; fq.append(qr, lqr); fq += '\n';
			}
			row += '\t'; row.append(num, put_int(t.support[k])); row += '\n';
		}
}

// `seeksv run`: a formatted piece's rows and FASTQ records as views into its text (the strings do not change any more)
static void index_piece(ResidentBam::Piece &p, const vector<RowAt> &ix)
{
	auto has_space = [](const char *q, size_t n) { for (size_t i = 0; i < n; ++i) if ((unsigned char)q[i] <= ' ') return true; return false; };
	const char *r = p.rows.data();
	p.parsed.reserve(ix.size()); p.reads.reserve(ix.size());
	if (p.rows.size() >= 0xffffffffull || p.fq.size() >= 0xffffffffull) p.clean = false;
	for (const RowAt &a : ix) {
		seeksv::ClipRow row;
		row.chr = seeksv::Str{r + a.row, a.chr_len}; row.pos = a.pos; row.side = a.side; row.cigar = seeksv::Str{r + a.cigar, a.cigar_len};
		row.aligned_seq = seeksv::Str{r + a.aligned, a.aligned_len}; row.clipped_seq = seeksv::Str{r + a.clipped, a.clipped_len}; row.clipped_qual = seeksv::Str{r + a.cqual, a.cqual_len};
		row.support = a.support;
		row.h = seeksv::clip_text_hash(row.clipped_seq.p, row.clipped_seq.n);
		// what `fin >> chr >> pos >> ...` (getsv.h:441-446) makes of the row's text: the same nine fields, unless one is empty or holds white space
		if (!a.chr_len || !a.cigar_len || !a.aligned_len || !a.clipped_len || !a.cqual_len || has_space(row.chr.p, row.chr.n)) p.clean = false;
		p.parsed.push_back(row);
		// the record's FASTQ text: '@' name '\n' sequence "\n+\n" qualities '\n' (clip_reads.h:320-321) - name = sequence
		p.reads.push_back(ResidentBam::FqRef{a.fq + 1 + a.clipped_len + 1, a.clipped_len, a.fq + 1 + a.clipped_len + 1 + a.clipped_len + 3, a.cqual_len});
	}
}

static int getclip_ranks(const string &bamfile, const string &prefix, double threshold, int min_mapQ, bool save_low_quality, int n_ranks, const vector<int> &devices, int halo_bp, bool device_inflate);
static int getclip_single(const string &bamfile, const string &prefix, double threshold, int min_mapQ, bool save_low_quality, int device, bool device_inflate);

static int cmd_getclip(int argc, char **argv)
{
	int c, min_mapQ = 1, device = 0, n_ranks = 1, halo_bp = 65536;
	double threshold = 0.9;
	string prefix = "output";
	bool save_low_quality = false, device_inflate = device_inflate_default();
	vector<int> devices;
	while ((c = getopt(argc, argv, "t:q:o:sG:ZCN:H:")) >= 0) {
		switch (c) {
		case 't': threshold = atof(optarg); break;
		case 'q': min_mapQ = atoi(optarg); break;
		case 's': save_low_quality = true; break;
		case 'o': prefix = optarg; break;
		case 'G': devices = parse_devices(optarg); device = devices.empty() ? 0 : devices[0]; break;
		case 'Z': device_inflate = true; break;
		case 'C': g_verify_crc = true; break;
		case 'N': n_ranks = atoi(optarg); break;
		case 'H': halo_bp = atoi(optarg); break;
		default: usage_getclip();
		}
	}
	if (argc != optind + 1 || n_ranks < 1 || halo_bp < 0) usage_getclip();
	const string bamfile = argv[optind];
	if (n_ranks > 1) return getclip_ranks(bamfile, prefix, threshold, min_mapQ, save_low_quality, n_ranks, devices, halo_bp, device_inflate);
	return getclip_single(bamfile, prefix, threshold, min_mapQ, save_low_quality, device, device_inflate);
}

static int getclip_single(const string &bamfile, const string &prefix, double threshold, int min_mapQ, bool save_low_quality, int device, bool device_inflate)
{
	PhaseTimer pt;
	{ // like the reference: complain about the input before anything is created
		ssvh_bam *probe = nullptr;
		if (ssvh_bam_open(bamfile.c_str(), &probe) != 0) die("[main_samview] fail to open file for reading.");
		ssvh_bam_close(probe);
	}
	GzOut softfout, fqfout, fuout1, fuout2;
	const string f_clip = prefix + ".clip.gz", f_fq = prefix + ".clip.fq.gz", f_u1 = prefix + ".unmapped_1.fq.gz", f_u2 = prefix + ".unmapped_2.fq.gz";
	if (!softfout.open(f_clip)) die("Cannot open file " + f_clip);
	if (!fqfout.open(f_fq)) die("Cannot open file " + f_fq);
	if (!fuout1.open(f_u1)) die("Cannot open file " + f_u1);
	if (!fuout2.open(f_u2)) die("Cannot open file " + f_u2);

	BatchSource src;
	src.open(bamfile, [device] { return acquire_ctx(device); }, device_inflate, "[main_samview] fail to open file for reading.");
	ssv_ctx *ctx = src.ctx;
	ssv_clip_params p;
	memset(&p, 0, sizeof(p));
	p.match_rate = threshold; p.min_mapq = min_mapQ; p.save_low_quality = save_low_quality ? 1 : 0;
	if (ssv_clip_begin(ctx, &p) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(ctx));
	ssv_clip_table_format(ctx, table_format_default()); // the compact table over PCIe; expanded by the formatting threads below
	ssvh_bam *bam = src.bam;
	pt.lap("open+gpu_init");

	// unmapped-pair side channel, StoreUnmapSeqAndQual (clip_reads.h:172-219): paired up, turned into FASTQ text and compressed by threads of its own
	// (unmapped_pairs.h); this thread hands every batch's raw UNMAP|MUNMAP records over by pointer
	seeksv::UnmappedPairs unmapped([&](vector<string> &parts) { fuout1.write_parts(parts); }, [&](vector<string> &parts) { fuout2.write_parts(parts); },
	                               std::max(1, std::min(8, ssv::effective_cpus() / 2)));
	// The flush sequence (clip_reads.h:428-438, :442): the reference writes out and clears its two maps at every change of contig among the
	// mapped-pair records.  A pass of the library bins by (contig, side, position), which is the same thing as long as every contig of the
	// pass is visited once, in ascending order - a coordinate-sorted BAM is ONE pass.  When a contig comes back (or one with a lower id
	// follows) the pass ends in front of that record (ssv_clip_scan_range), its rows go out, and the next pass starts there.
	vector<int32_t> pass_flushes;       // contigs flushed during the current pass, in order
	int32_t last_tid = 0, scan_last_tid = 0, pass_max_tid = 0; // (last_tid: the reader's side, one batch ahead of the scans)
	std::deque<vector<pair<int64_t, int32_t>>> changes_of; // per batch that was read and not yet scanned (the host reader runs one batch ahead)
	// The rows of a finished pass are formatted and compressed by a thread of their own while the main thread goes on reading and scanning the
	// next pass (the table sits in one of the context's two table sets: the emitter of pass k is joined before pass k + 1 is clustered, which is
	// when the library may touch that set again).  A coordinate-sorted BAM is one pass as far as the results go; it is cut at contig changes -
	// where the reference flushes anyway - whenever SSV_PASS_RECORDS records (default 16 M: about a contig of a 30x genome, so that only the last contig's rows are formatted after the last record) have gone into the pass, so that this overlap exists.
	static const int64_t pass_records_max = [] { const char *e = getenv("SSV_PASS_RECORDS"); return e ? std::max<int64_t>(1, atoll(e)) : (int64_t)16 << 20; }();
	int64_t pass_records = 0;
	std::thread emitter;
	double emit_format_s = 0, emit_gzip_s = 0;
	auto emit_pass = [&](bool last) {
		if (emitter.joinable()) emitter.join();
		pt.lap("wait for the previous pass's output");
		ssv_cluster_table t;
		if (ssv_clip_cluster(ctx, &t) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(ctx));
		if (t.format == 3 && ssv_clip_table_expand(ctx, &t, 0) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(ctx));
		pt.lap("gpu_cluster+table");
		// DisplaySClipReadsAndClipFq ('5' rows then '3' rows per contig run), clip_reads.h:300-345.  The rows of one cluster depend on nothing
		// else, so cluster ranges are formatted by several host threads and written out in order.
		int64_t k_end = 0;
		for (int32_t tid : pass_flushes) {
			const char *name = ssvh_bam_target_name(bam, tid);
			cerr << "Output merged soft-clipped reads of " << (name ? name : "") << endl;
			for (; k_end < t.n_clusters && t.tid[k_end] == tid; ++k_end) {}
		}
		pass_flushes.clear();
		pass_records = 0;
		emitter = std::thread([&, t, k_end] {
			const auto t0 = std::chrono::steady_clock::now();
			const int n_fmt = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)ssv::effective_cpus(), 32, k_end / 4096}));
			vector<string> rows((size_t)n_fmt), fqs((size_t)n_fmt);
			const bool keep = g_resident.collect && g_resident.ctx == ctx; // `seeksv run`: the aligner step and the junction stage take these from memory
			vector<vector<RowAt>> index(keep ? (size_t)n_fmt : 0);
			auto format_range = [&](int w) {
				const int64_t k0 = k_end * w / n_fmt, k1 = k_end * (w + 1) / n_fmt;
				format_clusters(t, bam, k0, k1, rows[(size_t)w], fqs[(size_t)w], keep ? &index[(size_t)w] : nullptr);
			};
			{
				vector<std::thread> th;
				for (int w = 1; w < n_fmt; ++w) th.emplace_back(format_range, w);
				format_range(0);
				for (auto &x : th) x.join();
			}
			const auto t1 = std::chrono::steady_clock::now();
			softfout.write_parts(rows); fqfout.write_parts(fqs);
			if (keep) {
				g_resident.clip_path = f_clip; g_resident.fq_path = f_fq;
				vector<ResidentBam::Piece *> fresh;
				vector<vector<RowAt> *> fresh_index;
				for (int w = 0; w < n_fmt; ++w) if (!rows[(size_t)w].empty()) {
					g_resident.pieces.emplace_back();
					ResidentBam::Piece &p = g_resident.pieces.back();
					p.rows = std::move(rows[(size_t)w]); p.fq = std::move(fqs[(size_t)w]);
					fresh.push_back(&p); fresh_index.push_back(&index[(size_t)w]);
				}
				{ // views and hashes, by as many threads as formatted (5.5 M rows of a whole-genome sample in all)
					vector<std::thread> th;
					for (size_t k = 1; k < fresh.size(); ++k) th.emplace_back([&, k] { index_piece(*fresh[k], *fresh_index[k]); });
					if (!fresh.empty()) index_piece(*fresh[0], *fresh_index[0]);
					for (auto &x : th) x.join();
				}
				if (g_resident.on_pass) g_resident.on_pass(fresh); // the aligner step starts on this pass's clipped sequences while the next pass is read
			}
			emit_format_s += std::chrono::duration<double>(t1 - t0).count();
			emit_gzip_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
		});
		if (last) { emitter.join(); pt.lap("wait for the last pass's output"); }
	};
	src.pump(0, [&](const ssv_batch_t &b) {
		pt.lap(device_inflate ? "bam_read(wait)+gpu_inflate+decode" : "bam_read(wait)");
		{ const uint8_t *raw = nullptr; size_t raw_bytes = 0; src.unmapped_raw(&raw, &raw_bytes); unmapped.submit(raw, raw_bytes); }
		changes_of.emplace_back();
		src.contig_changes(b, last_tid, [&](int64_t i, int32_t tid) { changes_of.back().emplace_back(i, tid); });
		pt.lap("host_side_channel");
	}, [&](const ssv_batch_t &decoded) {
		ssv_batch_t b = decoded;
		if (g_resident.collect && g_resident.ctx == ctx) { // `seeksv run`: the batch stays in HBM for the getsv passes
			if (ssv_batch_retain(ctx, &decoded, &b) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(ctx));
			g_resident.batches.push_back(b);
		}
		int64_t lo = 0;
		for (const auto &ch : changes_of.front()) {
			pass_flushes.push_back(scan_last_tid); // the visit that ends here is flushed
			if (ch.second <= pass_max_tid || pass_records + (ch.first - lo) >= pass_records_max) { // a contig that comes back (or a pass that is long enough): the pass ends in front of this record
				if (ssv_clip_scan_range(ctx, &b, lo, ch.first) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(ctx));
				emit_pass(false);
				p.initial_last_tid = scan_last_tid;
				if (ssv_clip_begin(ctx, &p) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(ctx));
				lo = ch.first;
			}
			pass_max_tid = scan_last_tid = ch.second;
		}
		changes_of.pop_front();
		if (ssv_clip_scan_range(ctx, &b, lo, b.n) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(ctx));
		pass_records += b.n - lo;
		pt.lap("gpu_scan(wait h2d+kernels)");
	});
	pass_flushes.push_back(scan_last_tid);
	emit_pass(true);
	cerr << "[GetSClipReads] finished!" << endl;
	unmapped.finish();
	softfout.close(); fqfout.close(); fuout1.close(); fuout2.close();
	pt.lap("gzip");
	if (pt.on) cerr << "[timing] (unmapped side channel, beside the reading: " << unmapped.records() << " records, " << unmapped.pairs() << " pairs written, its thread busy " << unmapped.busy_seconds() << " s)" << endl;
	if (pt.on) cerr << "[timing] (format, all passes, beside the reading: " << emit_format_s << " s; gzip: " << emit_gzip_s << " s)" << endl;
	src.close();
	release_ctx(ctx);
	pt.lap("teardown");
	return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// getclip over N runs of records, one per GPU (-N): the file is cut at record starts (ssvh_bam_partition); a rank scans its run plus the
// halo of records before it that can own a breakpoint inside it, keeps the clip events whose breakpoint lies in its interval, and clusters
// them: every (contig, side, position) bin lives on exactly one rank with its reads in file order, so the tables need no exchange - the
// rows are put together per contig: the '5' rows of all ranks in rank order, then the '3' rows (clip_reads.h:432-433).
// ---------------------------------------------------------------------------------------------------------------------

struct UnmappedRec { string qname, seq, qual; int is_read1; };

struct ClipRank {
	vector<int32_t> run_tids;                       // contig runs among the rank's own mapped-pair records
	int32_t last_tid = 0;
	vector<UnmappedRec> unmapped;                   // UNMAP|MUNMAP records of the rank's own run, in file order
	map<pair<int32_t, char>, pair<string, string>> seg; // (contig, side) -> rows, fastq records
	int32_t max_span = 0;
	string err;
};

static int getclip_ranks(const string &bamfile, const string &prefix, double threshold, int min_mapQ, bool save_low_quality, int n_ranks, const vector<int> &devices, int halo_bp, bool device_inflate)
{
	PhaseTimer pt;
	ssvh_bam *bam = nullptr;
	if (ssvh_bam_open(bamfile.c_str(), &bam) != 0) die("[main_samview] fail to open file for reading.");
	GzOut softfout, fqfout, fuout1, fuout2;
	const string f_clip = prefix + ".clip.gz", f_fq = prefix + ".clip.fq.gz", f_u1 = prefix + ".unmapped_1.fq.gz", f_u2 = prefix + ".unmapped_2.fq.gz";
	if (!softfout.open(f_clip)) die("Cannot open file " + f_clip);
	if (!fqfout.open(f_fq)) die("Cannot open file " + f_fq);
	if (!fuout1.open(f_u1)) die("Cannot open file " + f_u1);
	if (!fuout2.open(f_u2)) die("Cannot open file " + f_u2);
	vector<ssvh_bam_part> parts((size_t)n_ranks);
	if (ssvh_bam_partition(bamfile.c_str(), n_ranks, halo_bp, parts.data()) != 0) die(string("[seeksv] ") + ssvh_partition_last_error());
	pt.lap("open+partition");
	const int32_t n_targets = ssvh_bam_n_targets(bam);
	vector<ClipRank> R((size_t)n_ranks);
	auto rank_main = [&](int r) {
		ClipRank &out = R[(size_t)r];
		const ssvh_bam_part &P = parts[(size_t)r];
		ssv_ctx *ctx = nullptr;
		if (ssv_ctx_create(device_of_rank(r, devices), &ctx) != SSV_OK) { out.err = string("[seeksv] ") + ssv_last_error(nullptr); return; }
		ssv_clip_params p;
		memset(&p, 0, sizeof(p));
		p.match_rate = threshold; p.min_mapq = min_mapQ; p.save_low_quality = save_low_quality ? 1 : 0;
		// breakpoints (contig, 1-based position) owned by this rank: from its first record's 0-based start taken as a 1-based position - the
		// smallest key a record can give (a '5' event sits at start + 1, a '3' event at start + reference span, and the span is 0 for a CIGAR
		// that is one lone soft clip, clip_reads.cpp:150-190) - up to the next rank's
		p.use_ownership = 1; p.initial_last_tid = P.initial_last_tid;
		p.own_lo_tid = r == 0 ? 0 : P.own_tid; p.own_lo_pos = r == 0 ? 0 : P.own_pos;
		if (r + 1 < n_ranks) { p.own_hi_tid = parts[(size_t)r + 1].own_tid; p.own_hi_pos = parts[(size_t)r + 1].own_tid < n_targets ? parts[(size_t)r + 1].own_pos : 0; }
		else { p.own_hi_tid = INT32_MAX; p.own_hi_pos = 0; }
		if (ssv_clip_begin(ctx, &p) != SSV_OK) { out.err = string("[seeksv] ") + ssv_last_error(ctx); ssv_ctx_destroy(ctx); return; }
		ssv_clip_table_format(ctx, table_format_default());
		ssvh_bam *rb = nullptr; // (the header: contig names for the rows)
		if (ssvh_bam_open(bamfile.c_str(), &rb) != 0) { out.err = "[main_samview] fail to open file for reading."; ssv_ctx_destroy(ctx); return; }
		// the contig of the last mapped-pair record before the rank's OWN run (its run list continues from there)
		out.last_tid = r == 0 ? 0 : P.before_own_tid;
		for (int phase = 0; phase < 2 && out.err.empty(); ++phase) { // 0: the halo (scanned, nothing else), 1: the rank's own records
			if (phase == 0 && P.scan_coff == P.own_coff && P.scan_uoff == P.own_uoff) continue;
			BatchSource src; // host threads or the GPU (-Z) inflate and decode the run's blocks
			if (phase == 0) src.set_range(P.scan_coff, P.scan_uoff, P.own_coff, P.own_uoff);
			else { src.set_range(P.own_coff, P.own_uoff, P.end_coff, P.end_uoff); src.r_prev_tid = out.last_tid; }
			src.open(bamfile, ctx, device_inflate, "[main_samview] fail to open file for reading.");
			src.pump(0, [&](const ssv_batch_t &b) {
				out.max_span = std::max(out.max_span, b.max_ref_span);
				if (phase != 1) return;
				src.for_each_unmapped([&](const char *qname, const char *seq, const char *qual, int is_read1) { out.unmapped.push_back(UnmappedRec{qname, seq, qual, is_read1}); });
				src.contig_runs(b, out.last_tid, out.run_tids);
			}, [&](const ssv_batch_t &b) { if (out.err.empty() && ssv_clip_scan(ctx, &b) != SSV_OK) out.err = string("[seeksv] ") + ssv_last_error(ctx); });
			src.close();
		}
		ssv_cluster_table t;
		if (out.err.empty() && ssv_clip_cluster(ctx, &t) != SSV_OK) out.err = string("[seeksv] ") + ssv_last_error(ctx);
		if (out.err.empty() && t.format == 3 && ssv_clip_table_expand(ctx, &t, 2) != SSV_OK) out.err = string("[seeksv] ") + ssv_last_error(ctx);
		if (out.err.empty()) {
			for (int64_t k0 = 0; k0 < t.n_clusters;) { // the table is in (contig, side, position) order
				int64_t k1 = k0;
				while (k1 < t.n_clusters && t.tid[k1] == t.tid[k0] && t.side[k1] == t.side[k0]) ++k1;
				auto &sg = out.seg[make_pair(t.tid[k0], (char)t.side[k0])];
				format_clusters(t, rb, k0, k1, sg.first, sg.second);
				k0 = k1;
			}
		}
		ssvh_bam_close(rb);
		ssv_ctx_destroy(ctx);
	};
	{
		vector<std::thread> th;
		for (int r = 1; r < n_ranks; ++r) th.emplace_back(rank_main, r);
		rank_main(0);
		for (auto &x : th) x.join();
	}
	pt.lap("ranks(scan+cluster+format)");
	int32_t max_span = 0;
	for (auto &o : R) { if (!o.err.empty()) die(o.err); max_span = std::max(max_span, o.max_span); }
	if (max_span > halo_bp) die("[seeksv] a read spans " + to_string(max_span) + " reference bases, more than the halo of " + to_string(halo_bp) + " bp before a run of records: rerun with -H " + to_string(max_span));
	// the flush sequence of the whole file: the ranks' run lists one after the other (a rank's list continues the previous rank's)
	vector<int32_t> run_tids;
	for (auto &o : R) run_tids.insert(run_tids.end(), o.run_tids.begin(), o.run_tids.end());
	int32_t last_tid = 0;
	for (int r = n_ranks - 1; r >= 0; --r) if (parts[(size_t)r].own_coff != UINT64_MAX || r == 0) { last_tid = R[(size_t)r].last_tid; break; }
	run_tids.push_back(last_tid);
	for (size_t k = 1; k < run_tids.size(); ++k)
		if (run_tids[k] <= run_tids[k - 1]) {
			// Contigs that come back: the ranks' intervals of (contig, position) mean nothing for such a file.  The reference's result - one
			// flush per visit of a contig (clip_reads.h:428-438) - comes from the single-GPU path, which ends a pass at every such visit.
			ssvh_bam_close(bam);
			return getclip_single(bamfile, prefix, threshold, min_mapQ, save_low_quality, device_of_rank(0, devices), device_inflate);
		}
	// unmapped-pair side channel, StoreUnmapSeqAndQual (clip_reads.h:172-219): sequential over the file, i.e. over the ranks in order
	map<string, pair<pair<string, string>, char>> id2seq_qual;
	for (auto &o : R) for (auto &u : o.unmapped) {
		auto it = id2seq_qual.find(u.qname);
		if (it != id2seq_qual.end()) {
			if (u.is_read1 && it->second.second == '2') {
				fuout1.write(string("@") + it->first + "/1\n" + u.seq + "\n+\n" + u.qual + "\n");
				fuout2.write(string("@") + it->first + "/2\n" + it->second.first.first + "\n+\n" + it->second.first.second + "\n");
				id2seq_qual.erase(it);
			} else if (!u.is_read1 && it->second.second == '1') {
				fuout1.write(string("@") + it->first + "/1\n" + it->second.first.first + "\n+\n" + it->second.first.second + "\n");
				fuout2.write(string("@") + it->first + "/2\n" + u.seq + "\n+\n" + u.qual + "\n");
				id2seq_qual.erase(it);
			}
		} else id2seq_qual.insert(make_pair(u.qname, make_pair(make_pair(u.seq, u.qual), u.is_read1 ? '1' : '2')));
	}
	vector<string> rows, fqs;
	for (int32_t tid : run_tids) {
		const char *name = ssvh_bam_target_name(bam, tid);
		cerr << "Output merged soft-clipped reads of " << (name ? name : "") << endl;
		for (char side : {'5', '3'})
			for (auto &o : R) {
				auto it = o.seg.find(make_pair(tid, side));
				if (it == o.seg.end()) continue;
				rows.push_back(std::move(it->second.first)); fqs.push_back(std::move(it->second.second));
			}
	}
	softfout.write_parts(rows); fqfout.write_parts(fqs);
	cerr << "[GetSClipReads] finished!" << endl;
	softfout.close(); fqfout.close(); fuout1.close(); fuout2.close();
	pt.lap("output");
	ssvh_bam_close(bam);
	return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// getsv (junctions from -B)
// ---------------------------------------------------------------------------------------------------------------------

static string cigar_text(const SeqInfo &s) // DisplayCigarVector, clip_reads.h:489-505
{
	string o;
	if (s.left_clipped > 0) o += to_string(s.left_clipped) + "S";
	for (auto &pr : s.cigar_vec) { o += to_string(pr.first); o += pr.second; }
	if (s.right_clipped > 0) o += to_string(s.right_clipped) + "S";
	return o;
}

static string sv_type(const Junction &j) // GetSVType, clip_reads.cpp:572-581
{
	if (j.up_chr != j.down_chr) return "CTX";
	if (j.up_strand != j.down_strand) return "INV";
	if (j.up_pos < j.down_pos) return "DEL";
	if (j.up_pos > j.down_pos) return "INS";
	return "Unknown";
}

static double largest_base_frequency(const string &seq) // CountLargestBaseFrequency, getsv.cpp:1485
{
	int a = 0, t = 0, c = 0, g = 0, n = 0;
	for (char ch : seq) {
		if (ch == 'A' || ch == 'a') ++a; else if (ch == 'T' || ch == 't') ++t; else if (ch == 'C' || ch == 'c') ++c; else if (ch == 'G' || ch == 'g') ++g; else ++n;
	}
	int largest = max(max(max(a, t), max(c, g)), n);
	return largest / (double)seq.length();
}

// CalculateInsertsizeDeviation (cluster.cpp:15-83) over the head of a BAM, with the reference's stderr lines
// CalculateInsertsizeDeviation (cluster.cpp:15-83) reads the file until it has its pairs - the first chunk or the first few.  With the records decoded on the
// GPU their batches are copied into memory that stays (ssv_batch_retain; resident ones stay anyway): `keep` then holds the open source and those batches, and the
// fused pass scans them and goes on from the next chunk instead of reading, inflating and decoding the head of the file again.
struct IsizeCarry { BatchSource src; vector<ssv_batch_t> kept; bool owned = false, usable = false; };

static void insert_size_pass(const std::function<ssv_ctx *()> &get_ctx, const string &bamfile, bool device_inflate, int min_mapQ, int read_pair_used, int &mean_insert_size, int &deviation, IsizeCarry *keep = nullptr,
                             string *report = nullptr) // report: the pass's stderr lines go there instead (a caller that has earlier lines still to print)
{
	BatchSource local;
	BatchSource &src = keep ? keep->src : local;
	src.open(bamfile, get_ctx, device_inflate, "[main_samview] fail to open file for reading.");
	ssv_ctx *ctx = src.ctx ? src.ctx : get_ctx();
	ssv_isize_begin(ctx, min_mapQ, read_pair_used);
	int32_t done = 0;
	int chunks = 0;
	ssv_batch_t b;
	memset(&b, 0, sizeof(b));
	while (!done) {
		if (!src.next(&b, 0)) break;
		++chunks;
		if (ssv_isize_accumulate(ctx, &b, &done) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(ctx));
		if (keep && src.on_device) { // a decoder's batch is valid until the next decode: a copy that stays (80 B/record, device to device)
			ssv_batch_t k = b;
			if (!src.resident) { if (ssv_batch_retain(ctx, &b, &k) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(ctx)); keep->owned = true; }
			keep->kept.push_back(k);
		}
	}
	int64_t n; int32_t m = 0, sd = 0;
	ssv_isize_finish(ctx, &n, &m, &sd);
	if (n > 0) {
		mean_insert_size = m; deviation = sd;
		std::ostringstream o;
		o << "Bam/sam " << bamfile << "    Mean insert size : " << mean_insert_size << "\n" << "Mean deviation: " << deviation << "\n";
		if (report) *report += o.str(); else { cerr << o.str(); cerr.flush(); }
	}
	if (keep && src.on_device && chunks >= 1) { keep->usable = true; return; }
	src.close();
}

static void resident_alignments(seeksv::AlnRecords &R); // `seeksv run`: the aligner step's records as the join reads them

static int cmd_getsv(int argc, char **argv)
{
	string connect_bam, temp_breakpoint, dump_junctions;
	double frequency = 0.1;
	int c, min_mapQ = 20, read_pair_used = 5000000, sum_min_no_both_clipped_reads = 3, min_distance = 50, microhomology_length = 50, times = 4, device = 0,
	       min_abnormal_read_pair_no = 0, flank_length = 200, min_seq_len = 30, max_seq_indel_no = 1, flank = 50, n_ranks = 1;
	bool output_depth = true, device_inflate = device_inflate_default();
	vector<int> devices;
	PhaseTimer pt;
	while ((c = getopt(argc, argv, "F:B:t:l:q:Q:w:n:a:b:d:e:m:i:R:f:T:L:rDG:J:ZCN:")) >= 0) {
		switch (c) {
		case 'F': connect_bam = optarg; break;
		case 'B': temp_breakpoint = optarg; break;
		case 'l': flank = atoi(optarg); break;
		case 'q': min_mapQ = atoi(optarg); break;
		case 'n': read_pair_used = atoi(optarg); break;
		case 'b': sum_min_no_both_clipped_reads = atoi(optarg); break;
		case 'd': min_distance = atoi(optarg); break;
		case 'e': min_abnormal_read_pair_no = atoi(optarg); break;
		case 'm': min_seq_len = atoi(optarg); break;
		case 'i': max_seq_indel_no = atoi(optarg); break;
		case 'D': output_depth = false; break;
		case 'f': frequency = atof(optarg); break;
		case 'T': microhomology_length = atoi(optarg); break;
		case 'L': flank_length = atoi(optarg); break;
		case 'G': devices = parse_devices(optarg); device = devices.empty() ? 0 : devices[0]; break;
		case 'Z': device_inflate = true; break;
		case 'C': g_verify_crc = true; break;
		case 'N': n_ranks = atoi(optarg); break;
		case 'J': dump_junctions = optarg; break;
		default: break; // -t -Q -w -a -R -r: accepted, unused (as in the reference, where -t / -Q no longer reach the join)
		}
	}
	if (argc != optind + 5) usage_getsv();
	if (flank > 90 || flank < 0 || min_seq_len < 0 || n_ranks < 1) usage_getsv();
	const string clip_bam = argv[optind], original_bam = argv[optind + 1], clipfile = argv[optind + 2], breakpoint_file = argv[optind + 3], clip_unmap_fq_file = argv[optind + 4];
	if (!connect_bam.empty()) die("[seeksv] -F (bwasw read-through input) is not supported by this build");

	JunctionMap junction2other;
	if (!temp_breakpoint.empty()) { // ReadBreakpoint, getsv.cpp:1292-1323
		ifstream fin(temp_breakpoint.c_str());
		if (!fin) cerr << "Cannot open file " << temp_breakpoint << endl;
		string up_chr, down_chr, svt, up_cigar, down_cigar, up_seq, down_seq, temp;
		int up_pos, up_reads_no, down_pos, down_reads_no, micro, abnormal, d1, d2, d3, d4, d5, d6;
		char up_strand, down_strand;
		double r1, r2;
		while (fin >> up_chr) {
			if (up_chr[0] == '@') { getline(fin, temp); continue; }
			fin >> up_pos >> up_strand >> up_reads_no >> down_chr >> down_pos >> down_strand >> down_reads_no >> micro >> abnormal >> svt >> d1 >> d2 >> d3 >> d4 >> d5 >> d6 >> r1 >> r2 >>
			    up_cigar >> down_cigar >> up_seq >> down_seq;
			getline(fin, temp);
			Junction j{up_chr, up_pos, up_strand, down_chr, down_pos, down_strand};
			OtherInfo o;
			o.up.seq = up_seq; o.up.cigar_vec = parse_cigar(up_cigar); o.up.support = up_reads_no;
			o.down.seq = down_seq; o.down.cigar_vec = parse_cigar(down_cigar); o.down.support = down_reads_no;
			o.microhomology = micro; o.abnormal = abnormal;
			junction2other.insert(make_pair(j, o));
		}
		cerr << "[ReadBreakpoint] finish" << endl;
	}
	{ // InputSoftInfoStoreBreakpoint + GetJunction (getsv.h:423, getsv.cpp:1705): clip clusters x re-alignments of their clipped sequences
		string err;
		if (g_resident.ctx && clipfile == g_resident.clip_path && clip_bam == g_resident.aln.bam_path && g_resident.aln.rec) {
			seeksv::AlnRecords R; // rows and alignments are both still in memory
			resident_alignments(R);
			if (g_resident.all_clean() && !getenv("SSV_RUN_PARSE_ROWS")) { // the rows as the formatter kept them: nothing is parsed (SSV_RUN_PARSE_ROWS, tests: the text is)
				vector<const vector<seeksv::ClipRow> *> parts;
				for (const auto &p : g_resident.pieces) parts.push_back(&p.parsed);
				err = seeksv::assemble_junctions_rows(parts, R, junction2other);
			} else err = seeksv::assemble_junctions_records(g_resident.row_views(), R, junction2other);
		} else err = g_resident.ctx && clipfile == g_resident.clip_path ? seeksv::assemble_junctions_text(g_resident.row_views(), clip_bam, junction2other)
		                                                                : seeksv::assemble_junctions(clipfile, clip_bam, junction2other);
		if (!err.empty()) die(err);
	}
	cerr << "'InputSoftInfoStoreBreakpoint' finished" << endl;
	seeksv::merge_junctions(junction2other, flank); // MergeJunction(flank = -l)
	if (!dump_junctions.empty()) { // test hook: the junction table before any BAM pass (no GPU needed)
		ofstream jd(dump_junctions.c_str());
		for (auto &kv : junction2other) {
			const Junction &j = kv.first; const OtherInfo &o = kv.second;
			jd << j.up_chr << '\t' << j.up_pos << '\t' << j.up_strand << '\t' << o.up.support << '\t' << j.down_chr << '\t' << j.down_pos << '\t' << j.down_strand << '\t' << o.down.support << '\t'
			   << o.microhomology << '\t' << o.abnormal << '\t' << sv_type(j) << "\t0\t0\t0\t0\t0\t0\t0\t0\t" << cigar_text(o.up) << '\t' << cigar_text(o.down) << '\t' << o.up.seq << '\t' << o.down.seq
			   << '\t' << o.up.uniq << '\t' << o.down.uniq << '\n';
		}
		return 0;
	}

	pt.lap("junction_stage");
	ssvh_bam *bam = nullptr;
	if (ssvh_bam_open(original_bam.c_str(), &bam) != 0) die("[main_samview] fail to open file " + original_bam + "for reading.");
	ssv_ctx *ctx = nullptr;
	auto get_ctx = [&] { if (!ctx) ctx = acquire_ctx(device); return ctx; }; // (main() started it before the junction stage)

	int mean_insert_size = 0, deviation = 0;
	const bool do_discordant = read_pair_used >= 100000; // seeksv.cpp:246
	IsizeCarry carry;
	// SSV_GROUP_RCCL_ONE=1: `-N 1` takes the ranks' path too - one run of records, its vector through a one-rank RCCL communicator (what a
	// one-GPU box can execute of the exchange)
	const bool ranks_path = n_ranks > 1 || getenv("SSV_GROUP_RCCL_ONE");
	if (do_discordant) {
		insert_size_pass(get_ctx, original_bam, device_inflate, min_mapQ, read_pair_used, mean_insert_size, deviation, !ranks_path ? &carry : nullptr);
		cerr << "'CalculateInsertsizeDeviation' finished" << endl;
	} else min_abnormal_read_pair_no = 0; // seeksv.cpp:285
	get_ctx();

	pt.lap("isize_pass (+ what was left of the context's start-up)");
	// ---- the fused BAM pass: discordant tally + depth ----
	vector<ssvh_junction_in> J;
	vector<int32_t> prev;
	for (auto &kv : junction2other) {
		J.push_back(ssvh_junction_in{kv.first.up_chr.c_str(), kv.first.down_chr.c_str(), kv.first.up_pos, kv.first.down_pos, kv.first.up_strand, kv.first.down_strand});
		prev.push_back(kv.second.abnormal);
	}
	const int64_t nJ = (int64_t)J.size();
	ssvh_plan *plan = nullptr;
	ssvh_plan_create(bam, J.data(), nJ, nullptr, nullptr, 0, mean_insert_size, deviation, times, flank_length, &plan);
	int64_t nj, nw, nr, np;
	const ssv_junction *dj = ssvh_plan_junctions(plan, &nj);
	const ssv_interval *dw = ssvh_plan_windows(plan, &nw);
	const ssv_interval *dr = ssvh_plan_ranges(plan, &nr);
	const ssv_interval *dp = ssvh_plan_points(plan, &np);
	pt.lap("plan");
	vector<int32_t> counts((size_t)nj + 1, 0), pdepth((size_t)np + 1, 0);
	vector<uint64_t> rsum((size_t)nr + 1, 0);
	if ((do_discordant || output_depth) && ranks_path) {
		// ---- N runs of records, one per GPU: every rank scans its own records for ALL junctions and windows; the partial tallies, depth sums
		//      and point depths of the ranks meet in one all-gather (RCCL over xGMI between different GPUs) and are added up.  A rank replays
		//      the records before its run through the pileup's read-cap bookkeeping first (ssv_getsv_prime). ----
		vector<ssvh_bam_part> parts((size_t)n_ranks);
		if (ssvh_bam_partition(original_bam.c_str(), n_ranks, 0, parts.data()) != 0) die(string("[seeksv] ") + ssvh_partition_last_error());
		vector<ssv_ctx *> ctxs((size_t)n_ranks, nullptr);
		ctxs[0] = ctx;
		for (int r = 1; r < n_ranks; ++r) if (ssv_ctx_create(device_of_rank(r, devices), &ctxs[(size_t)r]) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(nullptr));
		ssv_group *group = nullptr;
		if (ssv_group_create(ctxs.data(), n_ranks, &group) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(nullptr));
		const size_t n_cnt = do_discordant ? (size_t)nj : 0, n_rs = output_depth ? (size_t)nr : 0, n_pd = output_depth ? (size_t)np : 0;
		const size_t vec_bytes = n_rs * 8 + (n_cnt + n_pd + 2) * 4; // [range sums u64 | counts i32 | point depths i32 | max depth, pad]
		vector<string> errs((size_t)n_ranks);
		vector<vector<uint8_t>> gathered((size_t)n_ranks);
		auto rank_main = [&](int r) {
			string &err = errs[(size_t)r];
			ssv_ctx *rc = ctxs[(size_t)r];
			const ssvh_bam_part &P = parts[(size_t)r];
			ssv_getsv_params gp;
			memset(&gp, 0, sizeof(gp));
			gp.junctions = dj; gp.n_junctions = do_discordant ? nj : 0;
			gp.mean = mean_insert_size; gp.sd = deviation; gp.times = times; gp.disc_min_mapq = min_mapQ;
			gp.windows = dw; gp.n_windows = output_depth ? nw : 0; gp.depth_min_mapq = min_mapQ;
			gp.n_targets = ssvh_bam_n_targets(bam); gp.target_len = ssvh_bam_target_lens(bam);
			ssvh_bam *rb = nullptr;
			if (ssvh_bam_open(original_bam.c_str(), &rb) != 0) { err = "[main_samview] fail to open file for reading."; }
			// the pileup's state at the rank's first record: replay the records before it (more of them while the replay itself starts inside a
			// > 8000x stack; back to the file's first record at most)
			for (int64_t back = 24576; err.empty(); back *= 4) {
				if (ssv_getsv_begin(rc, &gp) != SSV_OK) { err = string("[seeksv] ") + ssv_last_error(rc); break; }
				if (r == 0 || !output_depth || P.own_coff == UINT64_MAX) break;
				uint64_t co; uint32_t uo; int64_t found = 0;
				if (ssvh_bam_walk_back(original_bam.c_str(), P.own_coff, P.own_uoff, back, &co, &uo, &found) != 0) { err = string("[seeksv] ") + ssvh_partition_last_error(); break; }
				if (found == 0) break;
				int32_t sufficient = 1;
				bool primed = false;
				if (device_inflate && found <= (1 << 22)) {
					// -Z: the replayed records are inflated and decoded on the rank's GPU like the run itself (one chunk: they are a few MB)
					BatchSource rs;
					rs.set_range(co, uo, P.own_coff, P.own_uoff);
					rs.open(original_bam, rc, true, "[main_samview] fail to open file for reading.");
					ssv_batch_t db, more;
					if (rs.next(&db, 0)) {
						if (ssv_getsv_prime(rc, &db, &sufficient) != SSV_OK) { err = string("[seeksv] ") + ssv_last_error(rc); rs.close(); break; }
						primed = !rs.next(&more, 0); // (a replay that spans chunks goes through the host reader below: the bookkeeping takes one batch)
					}
					rs.close();
					if (!primed && ssv_getsv_begin(rc, &gp) != SSV_OK) { err = string("[seeksv] ") + ssv_last_error(rc); break; }
				}
				if (!primed) {
					if (ssvh_bam_set_range(rb, co, uo, P.own_coff, P.own_uoff) != 0) { err = string("[seeksv] ") + ssvh_last_error(); break; }
					ssv_batch_t hb;
					if (ssvh_bam_read_batch(rb, found + 16, 0, &hb) != 0) { err = string("[seeksv] ") + ssvh_last_error(); break; }
					if (ssv_getsv_prime(rc, &hb, &sufficient) != SSV_OK) { err = string("[seeksv] ") + ssv_last_error(rc); break; }
				}
				if (sufficient || found < back) break; // (found < back: the replay began at the file's first record - as far back as a replay can start)
			}
			if (err.empty()) { // the run itself: host threads or the GPU (-Z) inflate and decode its blocks
				BatchSource src;
				src.set_range(P.own_coff, P.own_uoff, P.end_coff, P.end_uoff);
				src.open(original_bam, rc, device_inflate, "[main_samview] fail to open file for reading.");
				src.pump(0, [](const ssv_batch_t &) {}, [&](const ssv_batch_t &b) { if (err.empty() && ssv_getsv_scan(rc, &b) != SSV_OK) err = string("[seeksv] ") + ssv_last_error(rc); });
				src.close();
			}
			vector<uint8_t> mine(vec_bytes, 0);
			uint64_t *v_rs = reinterpret_cast<uint64_t *>(mine.data());
			int32_t *v_cnt = reinterpret_cast<int32_t *>(mine.data() + n_rs * 8), *v_pd = v_cnt + n_cnt, *v_max = v_pd + n_pd;
			vector<int32_t> tmp_cnt(n_cnt + 1), tmp_pd(n_pd + 1);
			vector<uint64_t> tmp_rs(n_rs + 1);
			if (err.empty() && ssv_getsv_finish(rc, do_discordant ? tmp_cnt.data() : nullptr, dr, (int64_t)n_rs, tmp_rs.data(), dp, (int64_t)n_pd, tmp_pd.data(), v_max) != SSV_OK)
				err = string("[seeksv] ") + ssv_last_error(rc);
			if (n_rs) memcpy(v_rs, tmp_rs.data(), n_rs * 8);
			if (n_cnt) memcpy(v_cnt, tmp_cnt.data(), n_cnt * 4);
			if (n_pd) memcpy(v_pd, tmp_pd.data(), n_pd * 4);
			// the exchange: every rank takes part, also one that failed (its vector is zero; the error is reported afterwards)
			gathered[(size_t)r].assign(vec_bytes * (size_t)n_ranks, 0);
			if (ssv_group_allgather(group, r, mine.data(), vec_bytes, gathered[(size_t)r].data()) != SSV_OK && err.empty()) err = string("[seeksv] ") + ssv_last_error(rc);
			if (rb) ssvh_bam_close(rb);
		};
		{
			vector<std::thread> th;
			for (int r = 1; r < n_ranks; ++r) th.emplace_back(rank_main, r);
			rank_main(0);
			for (auto &x : th) x.join();
		}
		for (auto &e : errs) if (!e.empty()) die(e);
		// rank 0's copy of the gathered vectors: the sums are the whole file's tallies and depths
		for (int r = 0; r < n_ranks; ++r) {
			const uint8_t *v = gathered[0].data() + (size_t)r * vec_bytes;
			const uint64_t *v_rs = reinterpret_cast<const uint64_t *>(v);
			const int32_t *v_cnt = reinterpret_cast<const int32_t *>(v + n_rs * 8), *v_pd = v_cnt + n_cnt;
			for (size_t k = 0; k < n_rs; ++k) rsum[k] += v_rs[k];
			for (size_t k = 0; k < n_cnt; ++k) counts[k] += v_cnt[k];
			for (size_t k = 0; k < n_pd; ++k) pdepth[k] += v_pd[k];
		}
		if (getenv("SSV_TIMING")) cerr << "[timing] exchange over " << (ssv_group_uses_rccl(group) ? "RCCL (ncclAllGather)" : "host memory (ranks share a GPU)") << ", " << vec_bytes << " bytes per rank" << endl;
		ssv_group_destroy(group);
		for (int r = 1; r < n_ranks; ++r) ssv_ctx_destroy(ctxs[(size_t)r]);
	} else if (do_discordant || output_depth) {
		ssv_getsv_params gp;
		memset(&gp, 0, sizeof(gp));
		gp.junctions = dj; gp.n_junctions = do_discordant ? nj : 0;
		gp.mean = mean_insert_size; gp.sd = deviation; gp.times = times; gp.disc_min_mapq = min_mapQ;
		gp.windows = dw; gp.n_windows = output_depth ? nw : 0; gp.depth_min_mapq = min_mapQ;
		gp.n_targets = ssvh_bam_n_targets(bam); gp.target_len = ssvh_bam_target_lens(bam);
		if (ssv_getsv_begin(ctx, &gp) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(ctx));
		auto scan = [&](const ssv_batch_t &b) { if (ssv_getsv_scan(ctx, &b) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(ctx)); };
		if (carry.usable) { // the insert-size pass left its source open behind the file's first chunk, whose batch is still valid
			for (auto &k : carry.kept) { scan(k); if (carry.owned) ssv_batch_release(ctx, &k); }
			carry.kept.clear();
			carry.src.pump(0, [](const ssv_batch_t &) {}, scan);
			carry.src.close();
			carry.usable = false;
		} else {
			BatchSource src; // a second handle: `bam` keeps serving the header
			src.open(original_bam, ctx, device_inflate, "[main_samview] fail to open file for reading.");
			src.pump(0, [](const ssv_batch_t &) {}, scan);
			src.close();
		}
		int32_t maxd = 0;
		if (ssv_getsv_finish(ctx, do_discordant ? counts.data() : nullptr, dr, output_depth ? nr : 0, rsum.data(), dp, output_depth ? np : 0, pdepth.data(), &maxd) != SSV_OK)
			die(string("[seeksv] ") + ssv_last_error(ctx));
	}
	if (carry.usable) { if (carry.owned) for (auto &k : carry.kept) ssv_batch_release(ctx, &k); carry.kept.clear(); carry.src.close(); carry.usable = false; }
	pt.lap("fused_pass");
	if (do_discordant) { cerr << "'StoreSeqName2Tid' finished" << endl; cerr << "'FindDiscordantReadPairs' finished" << endl; }
	if (output_depth) { cerr << "'MergeOverlap' finished" << endl; cerr << "'main_depth' finished" << endl; }
	else frequency = 0; // seeksv.cpp:300
	vector<int32_t> abnormal((size_t)nJ + 1), up_depth((size_t)nJ + 1), down_depth((size_t)nJ + 1);
	vector<uint64_t> flank_sum((size_t)nJ * 4 + 4);
	vector<uint32_t> flank_len((size_t)nJ * 4 + 4);
	ssvh_plan_fold(plan, do_discordant ? counts.data() : nullptr, prev.data(), rsum.data(), pdepth.data(), abnormal.data(), up_depth.data(), down_depth.data(), flank_sum.data(),
	               flank_len.data(), nullptr);

	ofstream fout(breakpoint_file.c_str());
	if (!fout) die("Cannot open file " + breakpoint_file);
	fout << "@left_chr\tleft_pos\tleft_strand\tleft_clip_read_NO\tright_chr\tright_pos\tright_strand\tright_clip_read_NO\tmicrohomology_length\tabnormal_readpair_NO\tsvtype\tleft_pos_depth\t"
	        "right_pos_depth\taverage_depth_of_left_pos_5end\taverage_depth_of_left_pos_3end\taverage_depth_of_right_pos_5end\taverage_depth_of_right_pos_3end\tleft_pos_clip_percentage\t"
	        "right_pos_clip_percentage\tleft_seq_cigar\tright_seq_cigar\tleft_seq\tright_seq" << endl;
	// OutputBreakpoint, getsv.cpp:838-987
	int64_t idx = 0;
	for (auto it = junction2other.begin(); it != junction2other.end(); ++it, ++idx) {
		const Junction &j = it->first;
		OtherInfo &o = it->second;
		if (do_discordant) o.abnormal = abnormal[(size_t)idx];
		const int updepth = (output_depth ? up_depth[(size_t)idx] : 0) + o.down.support;     // without the depth pass pos2depth is empty: the reference prints an error and uses 0
		const int downdepth = (output_depth ? down_depth[(size_t)idx] : 0) + o.up.support;
		const int up_only = output_depth ? updepth : 0, down_only = output_depth ? downdepth : 0;
		if (!output_depth) {
			cerr << "Error: There is something wrong in upstream position " << j.up_chr << ":" << j.up_pos << endl;
			cerr << "Error: There is something wrong in downstream position " << j.down_chr << ":" << j.down_pos << endl;
		}
		const int ud = up_only, dd = down_only;
		const int junction_reads_no = o.up.support + o.down.support;
		const double rate1 = ud == 0 ? 0 : (double)junction_reads_no / ud, rate2 = dd == 0 ? 0 : (double)junction_reads_no / dd;
		auto filtered = [&](const char *reason) { // OutputFilteredBreakpoint, getsv.cpp:1846
			cout << reason << '\t' << j.up_chr << '\t' << j.up_pos << '\t' << j.up_strand << '\t' << o.up.support << '\t' << j.down_chr << '\t' << j.down_pos << '\t' << j.down_strand << '\t'
			     << o.down.support << '\t' << o.microhomology << '\t' << o.abnormal << '\t' << sv_type(j) << '\t' << ud << '\t' << dd << '\t' << rate1 << '\t' << rate2 << '\t'
			     << cigar_text(o.up) << '\t' << cigar_text(o.down) << '\t' << o.up.seq << '\t' << o.down.seq << endl;
		};
		if (!(o.up.uniq + o.down.uniq >= 2 || o.abnormal > 0)) { filtered("mappingQ_too_low"); continue; }
		if (j.up_chr == j.down_chr && abs(j.up_pos - j.down_pos) < min_distance) { filtered("distance_too_near"); continue; }
		if (o.microhomology > microhomology_length) { filtered("microhomology_len_too_long"); continue; }
		if (o.abnormal < min_abnormal_read_pair_no) { filtered("abnormal_read_pair_no_not_pass"); continue; }
		if ((o.up.support > 0 && o.down.support > 0 && rate1 < frequency && rate2 < frequency) || (o.up.support == 0 && rate2 < frequency) || (o.down.support == 0 && rate1 < frequency)) {
			filtered("frequency_too_low"); continue;
		}
		if (o.up.support + o.down.support < sum_min_no_both_clipped_reads) { filtered("total_clipped_reads_NO_not_pass"); continue; }
		if (o.abnormal == 0) {
			if (o.up.seq.length() < (size_t)(o.up.left_clipped + o.up.right_clipped + min_seq_len) || o.down.seq.length() < (size_t)(o.down.left_clipped + o.down.right_clipped + min_seq_len)) {
				filtered("seq_length_too_short"); continue;
			}
			if (o.up.cigar_vec.size() > (size_t)(2 * max_seq_indel_no + 1) || o.down.cigar_vec.size() > (size_t)(2 * max_seq_indel_no + 1)) { filtered("seq_with_too_many_indels"); continue; }
			if (largest_base_frequency(o.up.seq) >= 0.8 || largest_base_frequency(o.down.seq) >= 0.8) { filtered("repeat_bases"); continue; }
		}
		unsigned int fl[4] = {0, 0, 0, 0};
		if (output_depth) for (int k = 0; k < 4; ++k) fl[k] = (unsigned int)(flank_sum[(size_t)idx * 4 + k] / flank_len[(size_t)idx * 4 + k]); // getsv.cpp:946 (divides by zero like the reference if a window is empty)
		else cerr << "Error depth in the vicinity of junction " << j.up_chr << '\t' << j.up_pos << '\t' << j.up_strand << '\t' << j.down_chr << '\t' << j.down_pos << '\t' << j.down_strand << endl;
		fout << j.up_chr << '\t' << j.up_pos << '\t' << j.up_strand << '\t' << o.up.support << '\t' << j.down_chr << '\t' << j.down_pos << '\t' << j.down_strand << '\t' << o.down.support << '\t'
		     << o.microhomology << '\t' << o.abnormal << '\t' << sv_type(j) << '\t' << ud << '\t' << dd << '\t' << (int)fl[0] << '\t' << (int)fl[1] << '\t' << (int)fl[2] << '\t' << (int)fl[3] << '\t'
		     << rate1 << '\t' << rate2 << '\t' << cigar_text(o.up) << '\t' << cigar_text(o.down) << '\t' << o.up.seq << '\t' << o.down.seq << endl;
	}
	ofstream foutuq(clip_unmap_fq_file.c_str()); // OutputOneendUnmapBreakpoint: always empty (GetJunction returns early for type 'n')
	if (!foutuq) die("Cannot open file " + clip_unmap_fq_file);
	pt.lap("fold+output");
	ssvh_plan_destroy(plan);
	release_ctx(ctx);
	ssvh_bam_close(bam);
	pt.lap("teardown");
	return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// somatic (CallSomatic, seeksv.cpp:366-407; ReadTumorFileAndOutputSomaticInfo, somatic.cpp:14-427)
// ---------------------------------------------------------------------------------------------------------------------

static int cmd_somatic(int argc, char **argv)
{
	int offset = 30, c, min_len_of_clipped_seq = 10, read_pair_used = 5000000, min_mapQ = 20, device = 0;
	double min_map_rate = 0.9;
	string dump_lookups;
	bool device_inflate = device_inflate_default();
	PhaseTimer pt;
	while ((c = getopt(argc, argv, "t:q:l:m:n:G:J:ZC")) >= 0) {
		switch (c) {
		case 't': min_map_rate = atof(optarg); break;
		case 'q': min_mapQ = atoi(optarg); break;
		case 'l': offset = atoi(optarg); break;
		case 'm': min_len_of_clipped_seq = atoi(optarg); break;
		case 'n': read_pair_used = atoi(optarg); break;
		case 'G': device = atoi(optarg); break;
		case 'Z': device_inflate = true; break;
		case 'C': g_verify_crc = true; break;
		case 'J': dump_lookups = optarg; break; // test hook: the host look-ups only (no BAM pass, no GPU), one line per output row
		}
	}
	if (argc != optind + 4) { cerr << argc << '\t' << optind << endl; usage_somatic(); }
	else if (offset >= 90 || offset < 0) { cerr << "Error: value of -l must in range [0, 90) " << endl; usage_somatic(); }
	const string normal_bam_file = argv[optind++], clipped_file = argv[optind++], tumor_file = argv[optind++], somatic_file = argv[optind++];

	// The normal sample's clusters (all host threads: inflate + parse of a 30x sample's 5.5 M rows) are read BESIDE the GPU's work: what a row of the
	// tumor's table asks of the BAM (somatic.cpp:111,142: whether FindDiscordantReadPairs runs for it) depends on the insert size only, not on the look-ups.
	NormalClusters normal;
	string load_warnings, load_err;
	auto load = [&] { load_err = load_normal_clusters(clipped_file, min_len_of_clipped_seq, normal, load_warnings); };
	if (!dump_lookups.empty()) {
		load();
		if (!load_err.empty()) die(load_err);
		cerr << load_warnings;
		pt.lap("normal_clusters");
		vector<SomaticRow> rows;
		string err = scan_tumor_table(tumor_file, normal.clip3, normal.clip5, offset, min_map_rate, 1, rows);
		if (!err.empty()) die(err);
		ofstream jd(dump_lookups.c_str());
		for (auto &row : rows) {
			if (row.kind == SomaticRow::HEADER) jd << row.text << '\n';
			else if (row.kind == SomaticRow::MESSAGE) cerr << row.text << endl;
			else jd << row.text << '\t' << row.normal_left_reads << '\t' << row.normal_right_reads << '\n';
		}
		return 0;
	}
	std::thread loader(load);
	struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{loader}; // (die() leaves through _exit; a return must not leave the thread behind)
	ssv_ctx *ctx = nullptr;
	auto get_ctx = [&] { if (!ctx) ctx = acquire_ctx(device); return ctx; }; // (main() started it)
	int mean_insert_size = 0, deviation = 0;
	IsizeCarry carry;
	string isize_report; // ReadsClipReads' warnings come first on the reference's stderr (seeksv.cpp:392-398): the lines wait for the loader
	if (read_pair_used >= 100000) insert_size_pass(get_ctx, normal_bam_file, device_inflate, min_mapQ, read_pair_used, mean_insert_size, deviation, &carry, &isize_report);
	get_ctx();
	pt.lap("isize_pass (+ what was left of the context's start-up)");

	ofstream fout(somatic_file.c_str());
	if (!fout) die("Error: Cannot open output file " + somatic_file);
	ssvh_bam *bam = nullptr;
	if (ssvh_bam_open(normal_bam_file.c_str(), &bam) != 0) die("[main_samview] fail to open file for reading.");
	vector<SomaticRow> rows;
	{
		string err = parse_tumor_table(tumor_file, mean_insert_size, rows);
		if (!err.empty()) { loader.join(); if (!load_err.empty()) die(load_err); cerr << load_warnings << isize_report; die(err); }
	}
	pt.lap("tumor_table");

	// every row that asks for FindDiscordantReadPairs (getsv.cpp:1123-1247) becomes one junction of a single pass over the normal BAM
	vector<ssvh_junction_in> J;
	vector<size_t> row_of;
	for (size_t r = 0; r < rows.size(); ++r) {
		if (rows[r].kind != SomaticRow::JUNCTION || !rows[r].tally) continue;
		J.push_back(ssvh_junction_in{rows[r].up_chr.c_str(), rows[r].down_chr.c_str(), rows[r].up_pos, rows[r].down_pos, rows[r].up_strand, rows[r].down_strand});
		row_of.push_back(r);
	}
	vector<int32_t> abnormal(J.size() + 1, 0);
	if (!J.empty()) {
		const int times = 4; // seeksv.cpp:406
		ssvh_plan *plan = nullptr;
		ssvh_plan_create(bam, J.data(), (int64_t)J.size(), nullptr, nullptr, 0, mean_insert_size, deviation, times, 200, &plan);
		int64_t nj;
		const ssv_junction *dj = ssvh_plan_junctions(plan, &nj);
		vector<int32_t> counts((size_t)nj + 1, 0);
		ssv_getsv_params gp;
		memset(&gp, 0, sizeof(gp));
		gp.junctions = dj; gp.n_junctions = nj;
		gp.mean = mean_insert_size; gp.sd = deviation; gp.times = times; gp.disc_min_mapq = min_mapQ;
		gp.n_windows = 0; gp.depth_min_mapq = min_mapQ;
		gp.n_targets = ssvh_bam_n_targets(bam); gp.target_len = ssvh_bam_target_lens(bam);
		if (ssv_getsv_begin(ctx, &gp) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(ctx));
		auto scan = [&](const ssv_batch_t &b) { if (ssv_getsv_scan(ctx, &b) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(ctx)); };
		if (carry.usable) { // the insert-size pass left its source open behind the chunks it read, whose batches are still valid: ONE reading of the file
			for (auto &k : carry.kept) { scan(k); if (carry.owned) ssv_batch_release(ctx, &k); }
			carry.kept.clear();
			carry.src.pump(0, [](const ssv_batch_t &) {}, scan);
			carry.src.close();
			carry.usable = false;
		} else {
			BatchSource src; // a second handle: `bam` keeps serving the header
			src.open(normal_bam_file, ctx, device_inflate, "[main_samview] fail to open file for reading.");
			src.pump(0, [](const ssv_batch_t &) {}, scan);
			src.close();
		}
		int32_t maxd = 0;
		if (ssv_getsv_finish(ctx, counts.data(), nullptr, 0, nullptr, nullptr, 0, nullptr, &maxd) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(ctx));
		vector<int32_t> prev(J.size() + 1, 0); // a row whose left contig is not in the normal's header reports 0
		ssvh_plan_fold(plan, counts.data(), prev.data(), nullptr, nullptr, abnormal.data(), nullptr, nullptr, nullptr, nullptr, nullptr);
		ssvh_plan_destroy(plan);
	}
	if (carry.usable) { if (carry.owned) for (auto &k : carry.kept) ssv_batch_release(ctx, &k); carry.kept.clear(); carry.src.close(); carry.usable = false; }
	pt.lap("discordant_pass");
	loader.join();
	if (!load_err.empty()) die(load_err);
	cerr << load_warnings << isize_report;
	pt.lap("normal_clusters (wait: read beside the passes)");
	probe_normal_clusters(rows, normal.clip3, normal.clip5, offset, min_map_rate);
	pt.lap("cluster_lookups");
	pt.lap("discordant_pass");
	size_t k = 0;
	for (size_t r = 0; r < rows.size(); ++r) {
		const SomaticRow &row = rows[r];
		if (row.kind == SomaticRow::HEADER) { fout << row.text << endl; continue; }
		if (row.kind == SomaticRow::MESSAGE) { cerr << row.text << endl; continue; }
		int normal_abnormal = 0;
		if (row.tally) {
			if (k < row_of.size() && row_of[k] == r) normal_abnormal = abnormal[k++];
			bool known = false;
			for (int32_t t = 0, nt = ssvh_bam_n_targets(bam); t < nt && !known; ++t) known = row.up_chr == ssvh_bam_target_name(bam, t);
			if (!known) cerr << "Cannot find " << row.up_chr << ", please check whether you input a wrong file" << endl;
		}
		fout << row.text << '\t' << row.normal_left_reads << '\t' << row.normal_right_reads << '\t' << normal_abnormal << endl;
	}
	pt.lap("output");
	release_ctx(ctx);
	ssvh_bam_close(bam);
	pt.lap("teardown");
	return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// seeksv realign: prefix.clip.fq.gz -> prefix.clip.bam.  The reference's pipeline runs an external aligner here
// (README.md:22-34, example/seeksv.sh:3: `bwa mem ref.fa prefix.clip.fq.gz | samtools view -Sb - > prefix.clip.bam`); this is the
// stand-in for hosts without bwa (ssv_realign_*, include/seeksv_hip.h): one record per clipped sequence, in FASTQ order, the
// read name is the sequence.  Meant for references that behave like random sequence (synthetic genomes); no gapped alignment.
// ---------------------------------------------------------------------------------------------------------------------
[[noreturn]] static void usage_realign()
{
	cerr << "Usage: seeksv realign [options] <reference fasta(.gz)> <input clipped reads (*.clip.fq.gz)> <output clip.bam>\n\n"
	     << "         -G <int>              GPU ordinal [0]" << endl;
	exit(1);
}

static bool gz_getline(gzFile f, string &line)
{
	line.clear();
	char buf[1 << 16];
	while (gzgets(f, buf, sizeof(buf))) {
		line += buf;
		if (!line.empty() && line.back() == '\n') { line.pop_back(); if (!line.empty() && line.back() == '\r') line.pop_back(); return true; }
	}
	return !line.empty();
}

// A plain (not gzip'ed) FASTA file through all host threads: the file is mapped, cut into pieces at line starts; every piece counts its bases,
// a running sum gives each piece its first base index, then the pieces write their 2-bit codes (the words that two pieces share are OR-ed
// in atomically).  One thread parsing character by character took 2.8 s for an eighth of a human genome.  Same result as the serial loop
// (anything but ACGT becomes the same position-dependent pseudo-random base).  false: not a plain file (the caller reads it through zlib).
// base character -> 2-bit code, 4 for anything that is not A, C, G or T (either case)
static const struct BaseCode { uint8_t v[256]; BaseCode() { memset(v, 4, sizeof(v)); v['A'] = v['a'] = 0; v['C'] = v['c'] = 1; v['G'] = v['g'] = 2; v['T'] = v['t'] = 3; } uint8_t operator[](unsigned char c) const { return v[c]; } } kBaseCode;

static bool parse_fasta_parallel(const string &path, vector<string> &names, vector<int32_t> &lens, vector<int64_t> &offs, vector<uint64_t> &words)
{
	int fd = ::open(path.c_str(), O_RDONLY);
	if (fd < 0) return false;
	struct stat st;
	if (fstat(fd, &st) != 0 || st.st_size < 2) { ::close(fd); return false; }
	const size_t size = (size_t)st.st_size;
	const char *t = static_cast<const char *>(mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0));
	::close(fd);
	if (t == MAP_FAILED) return false;
	if ((unsigned char)t[0] == 0x1f && (unsigned char)t[1] == 0x8b) { munmap(const_cast<char *>(t), size); return false; } // gzip
	const int nt = std::max(1, std::min(ssv::effective_cpus(), 64));
	const size_t n_pieces = std::max<size_t>(1, std::min<size_t>((size_t)nt * 8, size / (1 << 20) + 1));
	vector<size_t> cut(n_pieces + 1, size);
	cut[0] = 0;
	for (size_t k = 1; k < n_pieces; ++k) { // the line start at or behind the k-th share of the file
		size_t p = size * k / n_pieces;
		const void *nl = p < size ? memchr(t + p, '\n', size - p) : nullptr;
		cut[k] = nl ? (size_t)(static_cast<const char *>(nl) - t) + 1 : size;
	}
	struct Hdr { size_t at; int64_t bases_before_in_piece; string name; };
	vector<int64_t> n_bases(n_pieces, 0);
	vector<vector<Hdr>> hdrs(n_pieces);
	auto run = [&](auto fn) {
		std::atomic<size_t> next{0};
		vector<std::thread> th;
		auto work = [&] { for (size_t k; (k = next.fetch_add(1)) < n_pieces;) fn(k); };
		for (int w = 1; w < nt; ++w) th.emplace_back(work);
		work();
		for (auto &x : th) x.join();
	};
	run([&](size_t k) { // pass 1: bases and header lines of piece k
		int64_t nb = 0;
		for (size_t p = cut[k]; p < cut[k + 1];) {
			const void *nl = memchr(t + p, '\n', cut[k + 1] - p);
			const size_t e = nl ? (size_t)(static_cast<const char *>(nl) - t) : cut[k + 1];
			if (t[p] == '>') {
				size_t w = p + 1;
				while (w < e && t[w] != ' ' && t[w] != '\t' && t[w] != '\r') ++w;
				hdrs[k].push_back(Hdr{p, nb, string(t + p + 1, w - p - 1)});
			} else {
				size_t len = e - p;
				if (len && t[e - 1] == '\r') --len;
				nb += (int64_t)len;
			}
			p = e + 1;
		}
		n_bases[k] = nb;
	});
	vector<int64_t> first(n_pieces + 1, 0);
	for (size_t k = 0; k < n_pieces; ++k) first[k + 1] = first[k] + n_bases[k];
	const int64_t total = first[n_pieces];
	if (hdrs[0].empty() || hdrs[0][0].at != 0) { munmap(const_cast<char *>(t), size); if (total == 0 && hdrs[0].empty()) return false; die("Reference file " + path + " does not start with a '>' line"); }
	for (size_t k = 0; k < n_pieces; ++k)
		for (const Hdr &h : hdrs[k]) {
			if (!names.empty()) { lens.push_back((int32_t)(first[k] + h.bases_before_in_piece - offs.back())); offs.push_back(first[k] + h.bases_before_in_piece); }
			names.push_back(h.name);
		}
	lens.push_back((int32_t)(total - offs.back())); offs.push_back(total);
	words.assign((size_t)((total + 31) / 32) + 1, 0);
	run([&](size_t k) { // pass 2: the codes of piece k
		int64_t n = first[k];
		const int64_t n_end = first[k + 1];
		uint64_t cur = 0;
		auto flush = [&](int64_t word) { if (cur) { if (word == first[k] / 32 || word == (n_end - 1) / 32) __atomic_fetch_or(&words[(size_t)word], cur, __ATOMIC_RELAXED); else words[(size_t)word] = cur; cur = 0; } };
		for (size_t p = cut[k]; p < cut[k + 1];) {
			const void *nl = memchr(t + p, '\n', cut[k + 1] - p);
			const size_t e = nl ? (size_t)(static_cast<const char *>(nl) - t) : cut[k + 1];
			if (t[p] != '>') {
				size_t stop = e;
				if (stop > p && t[stop - 1] == '\r') --stop;
				// (a table look-up per base: a switch on random bases is a mispredicted branch for three bases in four - 12 cycles a base, 14 CPU-seconds
				// for a human genome, beside getclip on the same 16 CPUs in `seeksv run`)
				for (size_t i = p; i < stop; ++i, ++n) {
					uint64_t two = kBaseCode[(unsigned char)t[i]];
					if (two > 3) { const uint64_t h = (uint64_t)n * 0x9E3779B97F4A7C15ull; two = (h >> 61) & 3; }
					cur |= two << (2 * (n & 31));
					if ((n & 31) == 31) flush(n / 32);
				}
			}
			p = e + 1;
		}
		if (n > first[k] && (n & 31) != 0) flush((n - 1) / 32);
	});
	munmap(const_cast<char *>(t), size);
	return true;
}

// `seeksv run`: the reference is read (and packed to 2 bits) by a thread of its own while getclip runs
struct PrereadFasta {
	string path;
	std::thread th;
	bool ok = false;
	vector<string> names; vector<int32_t> lens; vector<int64_t> offs; vector<uint64_t> words;
};
static PrereadFasta &g_preread = *new PrereadFasta; // (never destroyed: die() may exit while its thread runs)

// the reference: names, lengths, 2-bit bases (anything but ACGT becomes a position-dependent pseudo-random base, like bwa's index)
struct Reference { vector<string> names; vector<int32_t> lens; vector<int64_t> offs = vector<int64_t>(1, 0); vector<uint64_t> words; };
static void read_reference(const string &fasta, Reference &R)
{
	vector<string> &names = R.names; vector<int32_t> &lens = R.lens; vector<int64_t> &offs = R.offs; vector<uint64_t> &words = R.words;
	const bool fasta_check = getenv("SSV_FASTA_CHECK") != nullptr; // (tests: the serial reader runs too and must agree)
	vector<string> p_names; vector<int32_t> p_lens; vector<int64_t> p_offs(1, 0); vector<uint64_t> p_words;
	bool parsed;
	if (g_preread.th.joinable() && g_preread.path == fasta) { // `seeksv run` read the reference beside getclip
		g_preread.th.join();
		parsed = g_preread.ok;
		p_names.swap(g_preread.names); p_lens.swap(g_preread.lens); p_offs.swap(g_preread.offs); p_words.swap(g_preread.words);
	} else parsed = parse_fasta_parallel(fasta, p_names, p_lens, p_offs, p_words);
	if (parsed && !fasta_check) { names.swap(p_names); lens.swap(p_lens); offs.swap(p_offs); words.swap(p_words); }
	else {
		gzFile f = gzopen(fasta.c_str(), "rb");
		if (!f) die("Cannot open reference file " + fasta);
		string line;
		int64_t n = 0;
		auto push = [&](uint64_t two) { if ((n & 31) == 0) words.push_back(0); words.back() |= two << (2 * (n & 31)); ++n; };
		while (gz_getline(f, line)) {
			if (line.empty()) continue;
			if (line[0] == '>') {
				if (!names.empty()) { lens.push_back((int32_t)(n - offs.back())); offs.push_back(n); }
				size_t e = line.find_first_of(" \t");
				names.push_back(line.substr(1, e == string::npos ? string::npos : e - 1));
			} else {
				if (names.empty()) die("Reference file " + fasta + " does not start with a '>' line");
				for (char ch : line) {
					uint64_t two;
					switch (ch) {
					case 'A': case 'a': two = 0; break;
					case 'C': case 'c': two = 1; break;
					case 'G': case 'g': two = 2; break;
					case 'T': case 't': two = 3; break;
					default: { uint64_t h = (uint64_t)n * 0x9E3779B97F4A7C15ull; two = (h >> 61) & 3; }
					}
					push(two);
				}
			}
		}
		gzclose(f);
		if (names.empty()) die("No sequence in reference file " + fasta);
		lens.push_back((int32_t)(n - offs.back())); offs.push_back(n);
		words.push_back(0);
	}
	if (parsed && fasta_check && (names != p_names || lens != p_lens || offs != p_offs || words != p_words)) die("[seeksv] SSV_FASTA_CHECK: the parallel FASTA reader disagrees with the serial one");
}

// clip.bam's records as arrays (read name = sequence as given; SEQ / QUAL reverse-complemented / reversed for reverse-strand hits)
struct Line { char *p; int n; }; // a NUL-terminated line of FASTQ text, where it lies
struct AlignedRecords {
	vector<int32_t> tid, pos, lq, mtid, mpos, isz;
	vector<uint16_t> flag, ncig;
	vector<uint8_t> mapq, seqqual;
	vector<uint32_t> cig_off, cig;
	vector<uint64_t> seq_off;
	vector<const char *> qn;
	int64_t n_aligned = 0;
	int64_t size() const { return (int64_t)tid.size(); }
};
// the sequences through the aligner (ssv_realign_query) and their records appended to `out`
static void align_lines(ssv_ctx *ctx, const vector<Line> &seqs, const vector<Line> &quals, AlignedRecords &out)
{
	const int64_t n = (int64_t)seqs.size();
	if (!n) return;
	string blob;
	vector<uint64_t> soff(1, 0);
	{
		size_t total = 0;
		for (const Line &q : seqs) total += (size_t)q.n;
		blob.reserve(total);
		soff.reserve(seqs.size() + 1);
		for (const Line &q : seqs) { blob.append(q.p, (size_t)q.n); soff.push_back(blob.size()); }
	}
	vector<ssv_realign_hit> hits((size_t)n);
	if (ssv_realign_query(ctx, blob.data(), soff.data(), n, hits.data()) != SSV_OK) die(string("[seeksv] realign query: ") + ssv_last_error(ctx));
	const size_t at = (size_t)out.size();
	for (auto *v : {&out.tid, &out.pos, &out.lq}) v->resize(at + (size_t)n);
	out.mtid.resize(at + (size_t)n, -1); out.mpos.resize(at + (size_t)n, -1); out.isz.resize(at + (size_t)n, 0);
	out.flag.resize(at + (size_t)n); out.ncig.resize(at + (size_t)n); out.mapq.resize(at + (size_t)n); out.cig_off.resize(at + (size_t)n); out.seq_off.resize(at + (size_t)n); out.qn.resize(at + (size_t)n);
	int32_t *tid = out.tid.data() + at, *pos = out.pos.data() + at, *lq = out.lq.data() + at;
	uint16_t *flag = out.flag.data() + at, *ncig = out.ncig.data() + at;
	uint8_t *mapq = out.mapq.data() + at;
	uint32_t *cig_off = out.cig_off.data() + at;
	uint64_t *seq_off = out.seq_off.data() + at;
	const char **qn = out.qn.data() + at;
	auto code4 = [](char ch) -> uint8_t {
		switch (ch) { case 'A': case 'a': return 1; case 'C': case 'c': return 2; case 'G': case 'g': return 4; case 'T': case 't': return 8; default: return 15; }
	};
	auto comp4 = [](uint8_t b) -> uint8_t { return (uint8_t)(((b & 1) << 3) | ((b & 2) << 1) | ((b & 4) >> 1) | ((b & 8) >> 3)); };
	// sizes first (CIGAR operations and packed bytes per record, running sums), then every host thread fills its share of the records
	{
		uint64_t so = out.seqqual.size();
		uint32_t co = (uint32_t)out.cig.size();
		for (int64_t i = 0; i < n; ++i) {
			const ssv_realign_hit &h = hits[(size_t)i];
			const int L = seqs[(size_t)i].n;
			const bool al = h.tid >= 0;
			out.n_aligned += al ? 1 : 0;
			cig_off[i] = co; seq_off[i] = so;
			ncig[i] = (uint16_t)(al ? 1 + (h.q_beg > 0 ? 1 : 0) + (h.q_end < L ? 1 : 0) : 0);
			co += ncig[i]; so += ((uint64_t)L + 1) / 2 + (uint64_t)L;
		}
		out.cig.resize(co, 0);
		out.seqqual.resize(so, 0);
	}
	uint32_t *const cig = out.cig.data();
	uint8_t *const seqqual = out.seqqual.data();
	{
		const int nt = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)ssv::effective_cpus(), 64, n / 4096}));
		auto fill = [&](int w) {
			for (int64_t i = n * w / nt, e = n * (w + 1) / nt; i < e; ++i) {
				const ssv_realign_hit &h = hits[(size_t)i];
				const char *s = seqs[(size_t)i].p, *q = quals[(size_t)i].p;
				const int L = seqs[(size_t)i].n;
				const bool al = h.tid >= 0, rev = al && h.reverse, has_q = quals[(size_t)i].n == L;
				qn[i] = s;
				tid[i] = al ? h.tid : -1; pos[i] = al ? h.pos : -1; lq[i] = L;
				flag[i] = (uint16_t)(al ? (rev ? 16 : 0) : 4); mapq[i] = al ? h.mapq : 0;
				if (al) {
					uint32_t *c = cig + cig_off[i];
					if (h.q_beg > 0) *c++ = ((uint32_t)h.q_beg << 4) | 4u;
					*c++ = ((uint32_t)(h.q_end - h.q_beg) << 4) | 0u;
					if (h.q_end < L) *c++ = ((uint32_t)(L - h.q_end) << 4) | 4u;
				}
				uint8_t *sp = seqqual + seq_off[i], *qp = sp + ((size_t)L + 1) / 2;
				for (int k = 0; k < L; ++k) {
					const uint8_t b = rev ? comp4(code4(s[L - 1 - k])) : code4(s[k]);
					sp[k >> 1] |= (k & 1) ? b : (uint8_t)(b << 4);
					const char qc = has_q ? (rev ? q[L - 1 - k] : q[k]) : '!';
					qp[k] = (uint8_t)(qc - 33);
				}
			}
		};
		vector<std::thread> th;
		for (int w = 1; w < nt; ++w) th.emplace_back(fill, w);
		fill(0);
		for (auto &x : th) x.join();
	}
}

static void write_clip_bam(const string &out_bam, const Reference &R, AlignedRecords &A)
{
	const int64_t n = A.size();
	ssv_batch_t b;
	memset(&b, 0, sizeof(b));
	b.n = n; b.mem = SSV_MEM_HOST;
	b.tid = A.tid.data(); b.pos = A.pos.data(); b.flag = A.flag.data(); b.mapq = A.mapq.data(); b.n_cigar = A.ncig.data(); b.l_qseq = A.lq.data();
	b.mtid = A.mtid.data(); b.mpos = A.mpos.data(); b.isize = A.isz.data(); b.cigar_off = A.cig_off.data(); b.cigar = A.cig.data(); b.n_cigar_total = (int64_t)A.cig.size();
	b.seq_off = A.seq_off.data(); b.seqqual = A.seqqual.data(); b.seqqual_bytes = (int64_t)A.seqqual.size() - 16;
	vector<const char *> tn;
	for (const string &x : R.names) tn.push_back(x.c_str());
	if (ssvh_bam_write_batch_named(out_bam.c_str(), tn.data(), R.lens.data(), (int32_t)R.names.size(), &b, A.qn.data(), 0, 1) != 0) die(string("[seeksv] ") + ssvh_last_error());
}

static void resident_alignments(seeksv::AlnRecords &R)
{
	const AlignedRecords &A = *g_resident.aln.rec;
	R.n = A.size(); R.tid = A.tid.data(); R.pos = A.pos.data(); R.flag = A.flag.data(); R.n_cigar = A.ncig.data(); R.mapq = A.mapq.data();
	R.cigar_off = A.cig_off.data(); R.cigar = A.cig.data(); R.qname = A.qn.data(); R.target_names = g_resident.aln.names;
}

static int cmd_realign(int argc, char **argv)
{
	int gpu = 0, c;
	while ((c = getopt(argc, argv, "G:")) != -1) {
		if (c == 'G') gpu = atoi(optarg); else usage_realign();
	}
	if (argc - optind != 3) usage_realign();
	const string fasta = argv[optind], fq = argv[optind + 1], out_bam = argv[optind + 2];
	PhaseTimer pt;
	Reference R;
	read_reference(fasta, R);
	pt.lap("read reference");
	// ---- clipped sequences: the FASTQ file (its gzip members inflated side by side), cut into lines in place (the newline behind a line becomes its
	//      terminator: the sequences are the read names of the BAM records below) ----
	vector<Line> seqs, quals;
	string text;
	{ const string err = seeksv::slurp_gz(fq, text); if (!err.empty()) die("Cannot open clipped reads file " + fq); }
	{
		char *base = text.empty() ? nullptr : &text[0];
		const size_t size = text.size();
		size_t at = 0;
		auto line = [&](Line &out) { // like gz_getline: a line ends at '\n' (a '\r' in front of it is dropped); a last line without one counts when it is not empty
			if (at >= size) return false;
			char *b = base + at;
			char *e = static_cast<char *>(memchr(b, '\n', size - at));
			size_t len = e ? (size_t)(e - b) : size - at;
			at += len + (e ? 1 : 0);
			if (e) { *e = 0; if (len && b[len - 1] == '\r') b[--len] = 0; }
			out.p = b; out.n = (int)len;
			return e != nullptr || len != 0;
		};
		Line l1, l2, l3, l4;
		while (line(l1)) {
			if (!line(l2) || !line(l3) || !line(l4)) die("Truncated FASTQ record in " + fq);
			if (l1.n == 0 || l1.p[0] != '@' || l3.n == 0 || l3.p[0] != '+') die("Malformed FASTQ record in " + fq);
			seqs.push_back(l2); quals.push_back(l4);
		}
	}
	const int64_t n = (int64_t)seqs.size();
	pt.lap("read fastq");
	ssv_ctx *ctx = acquire_ctx(gpu);
	int64_t dropped = 0;
	if (ssv_realign_index(ctx, R.words.data(), SSV_MEM_HOST, R.offs.back(), R.offs.data(), (int32_t)R.names.size(), &dropped) != SSV_OK) die(string("[seeksv] realign index: ") + ssv_last_error(ctx));
	pt.lap("index");
	AlignedRecords A;
	align_lines(ctx, seqs, quals, A);
	pt.lap("align");
	A.seqqual.resize(A.seqqual.size() + 16, 0);
	// clip.bam is read back once, by getsv's join: its BGZF blocks are literal-only Huffman blocks (huff_gz.h: 4 x the speed of zlib level 1 on these
	// records, 1.6 x the bytes) unless SSV_BGZF_LEVEL asks for zlib
	setenv("SSV_BGZF_LEVEL", "-1", 0);
	write_clip_bam(out_bam, R, A);
	pt.lap("write bam");
	cerr << "[seeksv realign] " << n << " clipped sequences, " << A.n_aligned << " aligned" << (dropped ? ", " + to_string(dropped) + " repetitive index positions dropped" : string()) << endl;
	ssv_realign_free(ctx);
	release_ctx(ctx);
	return 0;
}

// `seeksv run`: the aligner step beside getclip.  A thread with a context of its own (a second stream on the same GPU) builds the index as soon as the
// reference is read - getclip is still inflating the BAM then - and takes every pass's clipped sequences as getclip's output thread writes them out:
// when the last record of the BAM has been scanned, all but the last pass's sequences are aligned (0.7 s of a 3.7 s whole-genome run came after getclip
// before).  The records equal `seeksv realign`'s on the finished prefix.clip.fq.gz: same sequences, same order, one query per sequence.
struct RunAligner {
	std::thread th;
	std::mutex mu;
	std::condition_variable cv;
	std::deque<vector<ResidentBam::Piece *>> todo;
	bool done = false;
	Reference R;
	AlignedRecords A;
	int64_t dropped = 0;
	double t_index = 0, t_align = 0, t_wait_ref = 0;
	void start(int device, const string &fasta)
	{
		th = std::thread([this, device, fasta] {
			auto now = [] { return std::chrono::steady_clock::now(); };
			auto t0 = now();
			ssv_ctx *ctx = nullptr;
			if (ssv_ctx_create(device, &ctx) != SSV_OK) die(string("[seeksv] ") + ssv_last_error(nullptr));
			read_reference(fasta, R);
			t_wait_ref = std::chrono::duration<double>(now() - t0).count();
			t0 = now();
			if (ssv_realign_index(ctx, R.words.data(), SSV_MEM_HOST, R.offs.back(), R.offs.data(), (int32_t)R.names.size(), &dropped) != SSV_OK) die(string("[seeksv] realign index: ") + ssv_last_error(ctx));
			t_index = std::chrono::duration<double>(now() - t0).count();
			for (;;) {
				vector<ResidentBam::Piece *> pass;
				{
					std::unique_lock<std::mutex> lk(mu);
					cv.wait(lk, [this] { return !todo.empty() || done; });
					if (todo.empty()) break;
					pass.swap(todo.front());
					todo.pop_front();
				}
				t0 = now();
				vector<Line> seqs, quals;
				for (ResidentBam::Piece *p : pass) {
					char *base = p->fq.empty() ? nullptr : &p->fq[0];
					for (const ResidentBam::FqRef &r : p->reads) {
						base[r.seq_off + r.seq_len] = 0; base[r.qual_off + r.qual_len] = 0; // (the newline behind a line becomes its terminator: the file is written)
						seqs.push_back(Line{base + r.seq_off, (int)r.seq_len}); quals.push_back(Line{base + r.qual_off, (int)r.qual_len});
					}
				}
				align_lines(ctx, seqs, quals, A);
				t_align += std::chrono::duration<double>(now() - t0).count();
			}
			A.seqqual.resize(A.seqqual.size() + 16, 0);
			ssv_realign_free(ctx);
			if (kCleanExit) ssv_ctx_destroy(ctx); else ssv_sync(ctx);
		});
	}
	void submit(const vector<ResidentBam::Piece *> &pass)
	{
		{ std::lock_guard<std::mutex> lk(mu); todo.push_back(pass); }
		cv.notify_all();
	}
	void finish()
	{
		{ std::lock_guard<std::mutex> lk(mu); done = true; }
		cv.notify_all();
		th.join();
	}
};

// ---------------------------------------------------------------------------------------------------------------------
// run: getclip -> realign -> getsv in one process (seeksv.cpp:128-329 + the pipeline's external aligner step, README.md:22-34)
// ---------------------------------------------------------------------------------------------------------------------

[[noreturn]] static void usage_run()
{
	cerr << "Usage: seeksv run [options] <input.sorted.bam> <reference fasta(.gz)> <output prefix>\n\n"
	     << "The three steps of the pipeline in one process on one GPU: `getclip` (prefix.clip.gz, prefix.clip.fq.gz, prefix.unmapped_[12].fq.gz),\n"
	     << "`realign` in the place of `bwa mem` (prefix.clip.bam), `getsv` (prefix.sv.txt, prefix.unmapped.clip.fq; filtered junctions on stdout).\n"
	     << "Every file is what the three commands write one after the other; the BAM is read, inflated and decoded ONCE, on the GPU, and its\n"
	     << "records stay in HBM for the getsv passes (80 bytes a record).\n\n"
	     << "Options: -c <string>           options handed to getclip, e.g. -c \"-q 5 -s\"\n"
	     << "         -v <string>           options handed to getsv, e.g. -v \"-b 5 -L 100\"\n"
	     << "         -G <int>              GPU ordinal [0]" << endl;
	exit(1);
}

static vector<string> split_words(const string &t)
{
	vector<string> w;
	std::istringstream in(t);
	for (string x; in >> x;) w.push_back(x);
	return w;
}

static int cmd_run(int argc, char **argv)
{
	int c, device = 0;
	string clip_opts, sv_opts;
	while ((c = getopt(argc, argv, "c:v:G:")) >= 0) {
		switch (c) {
		case 'c': clip_opts = optarg; break;
		case 'v': sv_opts = optarg; break;
		case 'G': device = atoi(optarg); break;
		default: usage_run();
		}
	}
	if (argc != optind + 3) usage_run();
	const string bam = argv[optind], fasta = argv[optind + 1], prefix = argv[optind + 2];
	PhaseTimer pt;
	g_resident.ctx = acquire_ctx(device);
	g_resident.path = bam;
	g_resident.collect = true;
	pt.lap("run: gpu_init");
	g_preread.path = fasta; g_preread.offs.assign(1, 0);
	g_preread.th = std::thread([] { g_preread.ok = parse_fasta_parallel(g_preread.path, g_preread.names, g_preread.lens, g_preread.offs, g_preread.words); });
	RunAligner &aligner = *new RunAligner; // (never destroyed: die() may exit while its thread runs)
	aligner.start(device, fasta);
	g_resident.on_pass = [&aligner](const vector<ResidentBam::Piece *> &pass) { aligner.submit(pass); };
	auto call = [&](int (*fn)(int, char **), vector<string> words) {
		vector<char *> av;
		for (auto &w : words) av.push_back(const_cast<char *>(w.c_str()));
		av.push_back(nullptr);
		optind = 1;
		return fn((int)words.size(), av.data());
	};
	vector<string> a = {"getclip"};
	for (auto &w : split_words(clip_opts)) a.push_back(w);
	a.insert(a.end(), {"-Z", "-G", to_string(device), "-o", prefix, bam});
	int rc = call(cmd_getclip, a);
	g_resident.collect = false;
	g_resident.on_pass = nullptr;
	pt.lap("run: getclip (decode once, records kept in HBM)");
	aligner.finish();
	pt.lap("run: realign (what was left of it behind getclip)");
	if (rc == 0) {
		// clip.bam is read back by nobody in this process (getsv's join takes the records from memory): it is written beside getsv, by a thread that is joined below;
		// its BGZF blocks are literal-only Huffman blocks (huff_gz.h) unless SSV_BGZF_LEVEL asks for zlib
		setenv("SSV_BGZF_LEVEL", "-1", 0);
		g_resident.aln.bam_path = prefix + ".clip.bam"; g_resident.aln.rec = &aligner.A; g_resident.aln.names = aligner.R.names;
		// (written under a temporary name and renamed when it is whole: a run that dies in getsv must not leave half a clip.bam where `seeksv getsv` would find it)
		g_resident.bam_writer = std::thread([&aligner] { write_clip_bam(g_resident.aln.bam_path + ".tmp", aligner.R, aligner.A); });
		cerr << "[seeksv realign] " << aligner.A.size() << " clipped sequences, " << aligner.A.n_aligned << " aligned" << (aligner.dropped ? ", " + to_string(aligner.dropped) + " repetitive index positions dropped" : string()) << endl;
		if (pt.on) cerr << "[timing] (aligner beside getclip: context + reference " << aligner.t_wait_ref << " s, index " << aligner.t_index << " s, align " << aligner.t_align << " s)" << endl;
		a = {"getsv"};
		for (auto &w : split_words(sv_opts)) a.push_back(w);
		a.insert(a.end(), {"-G", to_string(device), prefix + ".clip.bam", bam, prefix + ".clip.gz", prefix + ".sv.txt", prefix + ".unmapped.clip.fq"});
		rc = call(cmd_getsv, a);
	}
	pt.lap("run: getsv (records in HBM)");
	if (g_resident.bam_writer.joinable()) {
		g_resident.bam_writer.join();
		if (rename((g_resident.aln.bam_path + ".tmp").c_str(), g_resident.aln.bam_path.c_str()) != 0) die("Cannot write file " + g_resident.aln.bam_path);
		pt.lap("run: clip.bam written");
	}
	ssv_ctx *ctx = g_resident.ctx;
	if (kCleanExit) {
		for (auto &b : g_resident.batches) ssv_batch_release(ctx, &b);
		g_resident.ctx = nullptr;
		ssv_ctx_destroy(ctx);
	} else ssv_sync(ctx);
	return rc;
}

static double wall_now() { return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(); }

// SSV_TIMING=2: where the command's CPU time went - the process's user + system seconds (threads that have ended included) and, of the threads still alive, the ones
// that used more than 50 ms, by name (/proc/self/task).  A box that gives a command 16 CPUs of host time bounds it by this sum.
static void cpu_report()
{
	struct rusage ru;
	if (getrusage(RUSAGE_SELF, &ru) != 0) return;
	const double hz = (double)sysconf(_SC_CLK_TCK);
	fprintf(stderr, "[timing] (cpu: user %.2f s, system %.2f s in all;", ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6, ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6);
	std::map<string, std::pair<double, int>> by_name;
	if (DIR *d = opendir("/proc/self/task")) {
		while (struct dirent *e = readdir(d)) {
			if (e->d_name[0] == '.') continue;
			FILE *f = fopen((string("/proc/self/task/") + e->d_name + "/stat").c_str(), "r");
			if (!f) continue;
			char line[1024];
			if (fgets(line, sizeof line, f)) {
				const char *l = strchr(line, '('), *r = strrchr(line, ')');
				unsigned long ut = 0, st = 0;
				if (l && r && r > l && sscanf(r + 2, "%*c %*d %*d %*d %*d %*d %*u %*u %*u %*u %*u %lu %lu", &ut, &st) == 2) {
					auto &x = by_name[string(l + 1, r)];
					x.first += (double)(ut + st) / hz; ++x.second;
				}
			}
			fclose(f);
		}
		closedir(d);
	}
	fprintf(stderr, " threads alive at the end:");
	for (auto &kv : by_name) if (kv.second.first >= 0.05) fprintf(stderr, " %s x%d %.2f s;", kv.first.c_str(), kv.second.second, kv.second.first);
	fprintf(stderr, ")\n");
}

int main(int argc, char **argv)
{
	const bool stamp = getenv("SSV_TIMING") != nullptr; // with the caller's own clock around the process: exec -> main, main -> exit, exit -> reaped
	if (stamp) fprintf(stderr, "[stamp] wall clock at main: %.6f\n", wall_now());
	if (argc == 1) usage_top();
	const string cmd = argv[1];
	if (cmd != "getclip" && cmd != "getsv" && cmd != "somatic" && cmd != "realign" && cmd != "run") {
		cerr << "[seeksv] unrecognized command '" << argv[1] << "'" << endl;
		return 1;
	}
	if (argc == 2) { if (cmd == "getclip") usage_getclip(); else if (cmd == "getsv") usage_getsv(); else if (cmd == "realign") usage_realign(); else if (cmd == "run") usage_run(); else usage_somatic(); }
	{ // the GPU context starts coming up now (the device: the first number behind a -G)
		int dev = 0;
		for (int k = 2; k + 1 < argc; ++k) if (!strcmp(argv[k], "-G")) { dev = atoi(argv[k + 1]); break; }
		bool gpu_free = false; // getsv -J stops before any BAM pass (a test hook that needs no GPU)
		for (int k = 2; k < argc; ++k) if (!strcmp(argv[k], "-J")) gpu_free = true;
		g_near_device = dev;
		if (!gpu_free && argc > 3) early_ctx_start(dev);
	}
	optind = 1; // like SelectStep (seeksv.cpp:444-452): the sub-command becomes argv[0]
	int rc;
	if (cmd == "somatic") rc = cmd_somatic(argc - 1, argv + 1);
	else if (cmd == "realign") rc = cmd_realign(argc - 1, argv + 1);
	else if (cmd == "run") rc = cmd_run(argc - 1, argv + 1);
	else rc = cmd == "getclip" ? cmd_getclip(argc - 1, argv + 1) : cmd_getsv(argc - 1, argv + 1);
	if (kCleanExit) return rc;
	cout.flush(); cerr.flush(); fflush(nullptr);
	if (stamp && atoi(getenv("SSV_TIMING")) >= 2) cpu_report();
	if (stamp) { fprintf(stderr, "[stamp] wall clock at exit: %.6f\n", wall_now()); fflush(stderr); }
	_exit(rc); // (see release_ctx)
}
