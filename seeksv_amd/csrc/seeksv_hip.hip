// seeksv_hip.hip - context, memory management and the C ABI (include/seeksv_hip.h) over the gfx950 kernels.
// One context = one GPU = one HIP stream.  No CPU fallback: every entry point needs a live device.
#include "seeksv_hip.h"

#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <dlfcn.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <deque>
#include <functional>
#include <vector>

#include "bamdec_kernels.h"
#include "realign_kernels.h"
#include "clip_kernels.h"
#include "table3_kernels.h"
#include "tile_sort.h"
#include "common.h"
#include "cpus.h"
#include "getsv_kernels.h"
#include "radix_sort.h"
#include "scan.h"

using namespace ssv;

namespace {

thread_local std::string g_create_error; // (per thread: ssv_last_error(NULL) is asked by the thread whose call failed; rank and reader threads run side by side)

// timed kernel groups (ssv_prof_*)
enum ProfId { P_H2D, P_CLIP_SCAN, P_CLIP_PLACE, P_CLIP_GATHER, P_SORT, P_CLUSTER_BINS, P_CLUSTER_PACK, P_TABLE_D2H, P_ISIZE, P_GETSV_SCAN, P_GETSV_CAND, P_DEPTH_FINISH, P_BAM_INFLATE, P_BAM_RECORDS, P_BAM_DECODE, P_REALIGN_INDEX, P_REALIGN_QUERY, P_BAM_UPLOAD, P_BAM_RESOLVE, P_COUNT };
const char *const kProfNames[P_COUNT] = {"h2d", "clip_scan", "clip_place", "clip_gather", "event_sort", "cluster_bins", "cluster_pack", "table_d2h", "isize_stats", "getsv_scan", "getsv_cand", "depth_finish", "bam_inflate", "bam_records", "bam_decode", "realign_index", "realign_query", "bam_upload", "bam_resolve"};
const char kProfNameList[] = "h2d\nclip_scan\nclip_place\nclip_gather\nevent_sort\ncluster_bins\ncluster_pack\ntable_d2h\nisize_stats\ngetsv_scan\ngetsv_cand\ndepth_finish\nbam_inflate\nbam_records\nbam_decode\nrealign_index\nrealign_query\nbam_upload\nbam_resolve";

struct DBuf { // grow-only device buffer
	void *p = nullptr;
	size_t cap = 0;
};

struct HBuf { // grow-only pinned host buffer
	void *p = nullptr;
	size_t cap = 0;
};

// Device memory that is handed out in pieces and never moves (clip events hold addresses into it): a list of chunks with a cursor,
// rewound at the start of every getclip pass.
struct Arena {
	std::vector<DBuf> chunks;
	size_t cur = 0, used = 0;
};

struct ProfRec {
	int id;
	hipEvent_t a, b;
	int64_t units;
};

} // namespace

// A few host threads that stay around (the compact table's columns are rebuilt on them once per table: starting 64 threads costs more than
// the work they do).  run(n, fn) calls fn(0..n-1), fn(0) on the caller's thread, and returns when all are done.
struct HostPool {
	std::vector<std::thread> th;
	std::mutex mu;
	std::condition_variable cv_go, cv_done;
	const std::function<void(int)> *fn = nullptr;
	uint64_t generation = 0;
	int n_jobs = 0, n_left = 0;
	bool quit = false;
	void worker(int id)
	{
		uint64_t seen = 0;
		for (;;) {
			const std::function<void(int)> *f;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv_go.wait(lk, [&] { return quit || generation != seen; });
				if (quit) return;
				seen = generation;
				if (id >= n_jobs) continue;
				f = fn;
			}
			(*f)(id);
			std::unique_lock<std::mutex> lk(mu);
			if (--n_left == 0) cv_done.notify_all();
		}
	}
	void run(int n, const std::function<void(int)> &f)
	{
		if (n <= 1) { f(0); return; }
		while ((int)th.size() < n - 1) { const int id = (int)th.size() + 1; th.emplace_back([this, id] { worker(id); }); }
		{
			std::unique_lock<std::mutex> lk(mu);
			fn = &f; n_jobs = n; n_left = n - 1; ++generation;
		}
		cv_go.notify_all();
		f(0);
		std::unique_lock<std::mutex> lk(mu);
		cv_done.wait(lk, [&] { return n_left == 0; });
	}
	~HostPool()
	{
		{ std::unique_lock<std::mutex> lk(mu); quit = true; }
		cv_go.notify_all();
		for (auto &t : th) t.join();
	}
};

// ---- copies over the host link on an SDMA engine WE name ----
// hipMemcpyAsync lets the runtime pick the engine, and an MI355X has sixteen of which only four sit beside the PCIe root (tools/sdma_engine_probe.cpp: a 0.55 GB device-to-host copy takes
// 9.72 ms on engines 0-3, 43 ms on 4-7, 55-60 ms on 8-11, 72-78 ms on 12-15; host-to-device 9.65 / 10.9 / 13.7-15.2 / 18-19.5 ms).  On some boxes of the pool the runtime's choice (or the way
// it splits a large copy over several engines) made the SAME cluster table cross PCIe in 9.7 or in 12-18 ms from one copy to the next, and the compressed chunks of a BAM at 36-46 GB/s instead
// of 57 (round 6, profiles/r06_host_link.txt).  The two copies this path lives on - the cluster table to the host, a BAM's compressed chunks to the device - therefore go to the runtime's
// layer below HIP (hsa_amd_memory_async_copy_on_engine, found with dlsym: no link-time dependency) on the engine hsa_amd_memory_get_preferred_copy_engine names for the direction, with an HSA
// signal for the end.  Anything missing - the library, a symbol, the agents, a preferred engine - and the copy is hipMemcpyAsync's as before.  SSV_LINK_COPY=hip: that form always.
struct LinkCopy {
	bool ok = false;
	hsa_agent_t gpu{}, cpu{};
	uint32_t eng_to_host = 0, eng_to_device = 0; // the engine in use per direction, one bit each (hsa_amd_sdma_engine_id_t)
	// An engine is not ours alone: the kernel driver wipes freed device memory on one of them - for seconds behind a free of tens of GB, ours or that of the process before us -, and a
	// copy that shares it runs at half the link's rate (tools/free_wipe_probe.cpp, tools/sdma_engine_probe.cpp: engine 1 at 18.4 ms while 0, 2, 3 took 9.7).  So every large copy is
	// timed (the runtime's async-copy timestamps) and a direction moves on to the next of its candidate engines when a copy came in well below the best rate seen.
	struct Dir { std::vector<uint32_t> cand; std::vector<double> rate; size_t cur = 0; } to_host, to_device; // rate: GB/s of the candidate's last timed copy (0: not tried yet)
	double ticks_per_s = 0;
	decltype(&hsa_amd_profiling_get_async_copy_time) copy_time = nullptr;
	decltype(&hsa_signal_create) signal_create = nullptr;
	decltype(&hsa_signal_destroy) signal_destroy = nullptr;
	decltype(&hsa_signal_store_relaxed) signal_store = nullptr;
	decltype(&hsa_signal_wait_scacquire) signal_wait = nullptr;
	decltype(&hsa_signal_load_relaxed) signal_load = nullptr;
	decltype(&hsa_amd_memory_async_copy_on_engine) copy_on_engine = nullptr;
};

struct ssv_ctx {
	int device = 0;
	hipStream_t st = nullptr;
	std::string err;

	// ssv_batch_retain's memory: batches are cut out of arenas (a few large allocations instead of one per batch - with 320 batches of a whole-genome file kept,
	// `seeksv run` spent 0.9 s in allocations that grew slower with every one; an arena is given back when its last batch is released)
	struct RetainArena { uint8_t *base = nullptr; size_t cap = 0, used = 0; int64_t live = 0; std::vector<size_t> slabs; }; // slabs: where the arena's live batches begin
	std::vector<RetainArena> arenas;
	// arenas whose last batch was released: kept for the next ssv_batch_retain instead of handed back (round 6: a hipMalloc of 4 GB behind the release of tens of GB
	// took 0.6-1.7 s - the driver clears freed memory before it hands it out again -, once in every third run of bench.py's file leg); handed back when any
	// allocation of the context fails for lack of memory (dev_malloc), and with the context
	struct SpareArena { uint8_t *base; size_t cap; };
	std::vector<SpareArena> spare_arenas;
	// ... and the arena behind the one in use is allocated AHEAD, on a thread of its own, when the newest one is first cut into: a fresh process has no spares, and
	// its allocations may wait for memory that the process before it gave back (`seeksv run` behind `seeksv getsv`: 0.4-0.7 s of such waits inside getclip's
	// scan phase in two runs of three) - behind the decode of the batches that fill the current arena nobody waits for them
	struct ArenaAhead { std::thread th; bool pending = false; uint8_t *base = nullptr; size_t cap = 0; } ahead;

	// staging of host batches, and the record lines built for batches that come without them
	// host batches are copied into one of three staging sets: 0 and 1 take the batches announced with ssv_batch_prefetch (copied on st_h2d while
	// the kernels of the batch before run on st), 2 the ones that come unannounced (copied on st itself)
	struct StageSet { DBuf col[15], rec; hipEvent_t ready = nullptr; } ss[3];
	struct Prefetched { ssv_batch_t b; int set; };
	std::deque<Prefetched> pf;
	uint64_t pf_count = 0;
	hipStream_t st_h2d = nullptr;
	hipEvent_t ev_st = nullptr;

	// scratch shared by the passes
	DBuf tile_cnt, tile_off, tile_base, scan_scratch, scan_scratch64, counters;
	HBuf h_counters;

	// ---- getclip ----
	bool clip_active = false;
	ssv_clip_params clip_p{};
	DBuf d_last_tid, stage, cand, cand_cnt, cand_off, kv_stage, ends_buf;
	int64_t stage_cap = 0;
	DBuf ev, ev_meta, ev_idx, key_l, val_l, key_r[2], val_r[2];  // the pass's event lines (slots, with holes), per event in BAM order (l_qseq, n_cigar) and slot, and the sort keys / slots of the two sides
	int64_t ev_slots = 0;                                        // slots handed out so far
	int64_t ev_cap = 0, n_events = 0, n_l = 0, n_r = 0, n_long = 0;
	DBuf g_seq_bytes, g_cig_ops, g_seq_off, g_cig_off; // the copying path (batches without SSV_MEM_PERSISTENT)
	Arena blob;
	uint64_t sum_ncig = 0;
	int max_lq = 0, max_ncig = 0;
	// clustering temporaries / outputs
	DBuf keys2[2], vals2[2], evs, cum_l, cum_r, ghist, c_support, c_ll, c_lr, c_cig_ev, c_qmiss, c_mflag, c_mslot, c_mlist, c_bflag, c_boff, c_blist, c_dlist, bins4_tab, c_strings, slot_cnt, slot_bytes, tile_sums;
	DBuf o_slowlist, o_desc, totals;
	HBuf h_totals;
	// the dense cluster table: device columns + pinned host copy, double buffered so that the PCIe copy of one table can overlap
	// with whatever the caller runs next (ssv_clip_cluster_async / ssv_clip_table_wait)
	struct TableSet {
		DBuf o_tid, o_pos, o_side, o_support, o_ll, o_lr, o_qmiss, o_ncig, o_stroff, o_cigoff, o_str, o_cig;
		HBuf h_tid, h_pos, h_side, h_support, h_ll, h_lr, h_qmiss, h_stroff, h_cigoff, h_ncig, h_str, h_cig;
		// format 3 (compact): pos, flags (in o_qmiss / h_qmiss), str, cig as above, plus
		DBuf o_len, o_sup, o_nc, o_runs, o_exc;
		HBuf h_len, h_sup, h_nc, h_runs, h_exc;
		int format = 0, base_bits = 4, len_bytes = 4, support_bytes = 4, ncig_bytes = 4;
		int64_t n_runs = 0, n_exc = 0;
		uint64_t str_bytes = 0, cig_ops = 0;
		int64_t support_sum = 0;
		// the columns ssv_clip_table_expand rebuilds on the host
		std::vector<int32_t> x_tid, x_support, x_ll, x_lr, x_ncig;
		std::vector<uint8_t> x_side, x_qmiss;
		std::vector<uint64_t> x_stroff, x_cigoff;
		bool expanded = false, ordered = false;
		hipEvent_t copied = nullptr;
		hsa_signal_t copied_sig{};                      // ... or, when the copy went to a named SDMA engine (LinkCopy), its HSA signal: zero when every piece has landed
		bool via_link = false;
		hsa_signal_t big_sig{}; size_t big_bytes = 0;     // the largest piece (the string block) on a signal of its own: it is the one that is timed
		hipEvent_t packed_ev = nullptr;                 // the set's pack kernels are done (its copy waits for it): one event per set - a copy that is
		                                                // still queued behind the table before must not see the next pass's record of a shared event
		bool in_flight = false;
		int64_t n_clusters = 0, n_events = 0;
		int packed = 0, qual_bits = 8, qual_group = 1, qual_radix = 0, cig_bytes = 4; // qual_group > 1 (format 3): qual_bits per group of that many qualities, radix = the alphabet's size
		uint8_t qual_alphabet[64] = {0};
	} tab[2];
	HostPool pool;
	int tab_cur = 0;           // set of the most recent ssv_clip_cluster[_async]
	int table_mode = 0;        // ssv_clip_table_format: 0 ASCII, 3 compact
	DBuf qual_lut, qual_seen, pair_lut; HBuf h_qual_lut, h_pair_lut;
	hipStream_t st_copy = nullptr;
	hipEvent_t ev_packed = nullptr;
	LinkCopy link;             // the named SDMA engines of the host link (ok == false: hipMemcpyAsync)

	// ---- isize ----
	bool isz_active = false;
	int isz_min_mapq = 0;
	int64_t isz_max = 0, isz_count = 0;
	DBuf isz_vals, isz_acc, isz_tmp;

	// ---- getsv ----
	bool gs_active = false;
	ssv_getsv_params gs_p{};
	std::vector<DevJunction> gs_junc;
	std::vector<ssv_interval> gs_win;
	std::vector<int32_t> gs_tlen;
	std::vector<int64_t> gs_ctg_tile_off;
	int32_t gs_map_span = -1;
	int32_t gs_wmax = 0;
	int64_t gs_diff_len = 0;
	DBuf gs_djunc, gs_counts, gs_wtid, gs_wbeg, gs_wend, gs_woff, gs_diff, gs_tilemap, gs_tile_win, gs_tile_junc, gs_ctgoff, gs_maxdepth, gs_span;
	DBuf q_tid, q_beg, q_end, q_out64, q_out32;
	// read cap of the reference's pileup (k_cap_*): flags, per-tile marks, carried sweep state + ring, the stream's last records (ping-pong)
	DBuf dense_list, cap_flags, cap_deep, cap_carry, cap_ring, cap_ring_tmp, cap_tail[2][4];
	int32_t cap_tail_n = 0, cap_tail_cur = 0, cap_ring_mask = 0;
	HBuf h_q;

	// ---- device BGZF/BAM decoder (bamdec_api.inc) ----
	struct ssv_bamdec_state *bd = nullptr;
	// ---- clipped-sequence re-aligner (realign_api.inc) ----
	struct ssv_realign_state *ra = nullptr;
	HBuf h_batch;

	// ---- profiling ----
	int prof_mode = 0;
	std::vector<ProfRec> prof_recs;
	std::vector<hipEvent_t> prof_pool;
	double prof_ms[P_COUNT] = {0};
	int64_t prof_launches[P_COUNT] = {0};
	int64_t prof_units[P_COUNT] = {0};
};

namespace {

#define HIPCHECK(ctx, call)                                                                              \
	do {                                                                                                   \
		hipError_t e_ = (call);                                                                              \
		if (e_ != hipSuccess) {                                                                              \
			(ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                    \
			return e_ == hipErrorOutOfMemory ? SSV_E_NOMEM : SSV_E_HIP;                                        \
		}                                                                                                    \
	} while (0)

#define CHECK(expr)          \
	do {                       \
		int rc_ = (expr);        \
		if (rc_ != SSV_OK) return rc_; \
	} while (0)

// the arena that was asked for ahead, if any (null when there is none or its allocation failed); the caller owns it
static uint8_t *arena_ahead_take(ssv_ctx *c, size_t *cap)
{
	if (!c->ahead.pending) return nullptr;
	c->ahead.th.join();
	c->ahead.pending = false;
	uint8_t *b = c->ahead.base;
	*cap = c->ahead.cap;
	c->ahead.base = nullptr; c->ahead.cap = 0;
	return b;
}

static void arena_ahead_start(ssv_ctx *c, size_t cap)
{
	if (c->ahead.pending) return;
	c->ahead.pending = true;
	c->ahead.base = nullptr; c->ahead.cap = cap;
	const int device = c->device;
	ssv_ctx::ArenaAhead *a = &c->ahead;
	c->ahead.th = std::thread([a, device, cap] {
		void *p = nullptr;
		if (hipSetDevice(device) != hipSuccess || hipMalloc(&p, cap) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
		a->base = static_cast<uint8_t *>(p);
	});
}

// hipMalloc; out of memory: what the context keeps in reserve (ssv_batch_retain's spare arenas, the one asked for ahead) goes back first
hipError_t dev_malloc(ssv_ctx *c, void **p, size_t bytes)
{
	hipError_t e = hipMalloc(p, bytes);
	if (e == hipErrorOutOfMemory && (!c->spare_arenas.empty() || c->ahead.pending)) {
		(void)hipGetLastError();
		(void)hipStreamSynchronize(c->st);
		size_t cap = 0;
		if (uint8_t *b = arena_ahead_take(c, &cap)) (void)hipFree(b);
		for (auto &a : c->spare_arenas) (void)hipFree(a.base);
		c->spare_arenas.clear();
		e = hipMalloc(p, bytes);
	}
	return e;
}

int ensure(ssv_ctx *c, DBuf &b, size_t bytes, bool keep = false, size_t keep_bytes = 0)
{
	if (bytes <= b.cap) return SSV_OK;
	size_t ncap = std::max(bytes, b.cap + b.cap / 2);
	ncap = (ncap + 255) & ~(size_t)255;
	void *np = nullptr;
	HIPCHECK(c, dev_malloc(c, &np, ncap));
	if (keep && b.p && keep_bytes) {
		HIPCHECK(c, hipMemcpyAsync(np, b.p, keep_bytes, hipMemcpyDeviceToDevice, c->st));
		HIPCHECK(c, hipStreamSynchronize(c->st));
	}
	if (b.p) {
		HIPCHECK(c, hipStreamSynchronize(c->st));
		HIPCHECK(c, hipFree(b.p));
	}
	b.p = np; b.cap = ncap;
	return SSV_OK;
}

// Page-locked host memory.  hipHostMalloc hands out 4 KB pages: allocating, clearing and locking them runs on one thread at ~0.25 s/GB under the runtime's lock, and
// when the process ends the kernel takes every page's lock back one by one - tools/exit_cost.cpp: 4 GB of such memory cost 0.18 s to lock and 0.41-0.45 s between
// _exit and the parent's wait() returning, whatever else the process held (48 GB of device memory: 0.06 s).  That was the 0.2-0.6 s a `seeksv` command spent AFTER its
// last statement (round 6, bench.py: exit_to_reaped_s).  Buffers of 2 MB and more are therefore anonymous memory on TRANSPARENT HUGE PAGES (madvise: 2 MB pages where
// the kernel grants them - it falls back to 4 KB pages by itself), touched by a few threads, then registered with the runtime: 0.008 s to lock 4 GB, 0.001 s to leave.
// SSV_PINNED=malloc: hipHostMalloc as before (the form the tests compare with).
namespace {
struct PinnedMaps { std::mutex mu; std::unordered_map<void *, std::pair<void *, size_t>> m; }; // user pointer -> (mapping, its length)
PinnedMaps &pinned_maps() { static PinnedMaps *p = new PinnedMaps; return *p; } // (never destroyed: buffers may be freed from static destructors)
constexpr size_t HUGE_PAGE = (size_t)2 << 20;
}

// the NUMA node a GPU hangs on (sysfs numa_node of its PCI function; -1: unknown or a one-node box)
static int device_numa_node(int device)
{
	static std::mutex mu;
	static std::unordered_map<int, int> cache;
	std::lock_guard<std::mutex> lk(mu);
	auto it = cache.find(device);
	if (it != cache.end()) return it->second;
	int node = -1;
	char bdf[64] = {0};
	if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) == hipSuccess) {
		for (char *q = bdf; *q; ++q) *q = (char)tolower((unsigned char)*q);
		const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/numa_node";
		if (FILE *f = fopen(path.c_str(), "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
	} else (void)hipGetLastError();
	cache[device] = node;
	return node;
}

static void bind_near(void *p, size_t bytes, int device)
{
	const int node = device_numa_node(device);
	if (node < 0 || node >= 1024 || !p || !bytes) return;
	unsigned long mask[16] = {0};
	mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
	(void)syscall(SYS_mbind, p, (unsigned long)bytes, 1 /* MPOL_PREFERRED */, mask, (unsigned long)(8 * sizeof(mask)), 0u); // (refused by a cpuset that does not hold the node: the pages land where they land)
}

static hipError_t pinned_new(void **out, size_t bytes, unsigned malloc_flags)
{
	static const bool plain = [] { const char *e = getenv("SSV_PINNED"); return e && !strcmp(e, "malloc"); }();
	*out = nullptr;
	if (plain || bytes < HUGE_PAGE) return hipHostMalloc(out, bytes ? bytes : 1, malloc_flags);
	const size_t n = (bytes + HUGE_PAGE - 1) & ~(HUGE_PAGE - 1), len = n + HUGE_PAGE;
	void *m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
	if (m == MAP_FAILED) return hipHostMalloc(out, bytes, malloc_flags);
	uint8_t *a = reinterpret_cast<uint8_t *>(((uintptr_t)m + HUGE_PAGE - 1) & ~(uintptr_t)(HUGE_PAGE - 1));
	(void)madvise(a, n, MADV_HUGEPAGE);
	{ int dev = 0; if (hipGetDevice(&dev) == hipSuccess) bind_near(a, n, dev); else (void)hipGetLastError(); } // the pages on the GPU's own NUMA node, whichever CPUs touch them
	{ // first touch on a few threads (a fault clears 2 MB - or 4 KB, 512 times as often - before the runtime's call locks the pages one after the other)
		const int nt = (int)std::max<size_t>(1, std::min<size_t>({(size_t)8, n >> 25, (size_t)ssv::effective_cpus()}));
		auto touch = [a, n, nt](int t) { for (size_t o = n / HUGE_PAGE * (size_t)t / (size_t)nt * HUGE_PAGE, e = n / HUGE_PAGE * (size_t)(t + 1) / (size_t)nt * HUGE_PAGE; o < e; o += 4096) a[o] = 0; };
		std::vector<std::thread> th;
		for (int t = 1; t < nt; ++t) th.emplace_back(touch, t);
		touch(0);
		for (auto &x : th) x.join();
	}
	if (hipHostRegister(a, n, hipHostRegisterPortable) != hipSuccess) { // (a kernel or limit that refuses: the runtime's own allocator)
		(void)hipGetLastError();
		munmap(m, len);
		return hipHostMalloc(out, bytes, malloc_flags);
	}
	{ std::lock_guard<std::mutex> lk(pinned_maps().mu); pinned_maps().m[a] = std::make_pair(m, len); }
	*out = a;
	return hipSuccess;
}

static hipError_t pinned_delete(void *p)
{
	if (!p) return hipSuccess;
	std::pair<void *, size_t> mapping(nullptr, 0);
	{
		std::lock_guard<std::mutex> lk(pinned_maps().mu);
		auto it = pinned_maps().m.find(p);
		if (it != pinned_maps().m.end()) { mapping = it->second; pinned_maps().m.erase(it); }
	}
	if (!mapping.first) return hipHostFree(p);
	const hipError_t e = hipHostUnregister(p);
	munmap(mapping.first, mapping.second);
	return e;
}

// find the agents and the engines (once per context; quiet on failure: the context then copies the runtime's way)
static void link_init(ssv_ctx *c)
{
	LinkCopy &L = c->link;
	const char *e = getenv("SSV_LINK_COPY");
	if (e && !strcmp(e, "hip")) return;
	void *lib = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_GLOBAL);
	if (!lib) lib = dlopen("libhsa-runtime64.so", RTLD_NOW | RTLD_GLOBAL);
	if (!lib) return;
	auto sym = [&](const char *name) { return dlsym(lib, name); };
	auto f_init = reinterpret_cast<decltype(&hsa_init)>(sym("hsa_init"));
	auto f_iter = reinterpret_cast<decltype(&hsa_iterate_agents)>(sym("hsa_iterate_agents"));
	auto f_info = reinterpret_cast<decltype(&hsa_agent_get_info)>(sym("hsa_agent_get_info"));
	auto f_pref = reinterpret_cast<decltype(&hsa_amd_memory_get_preferred_copy_engine)>(sym("hsa_amd_memory_get_preferred_copy_engine"));
	auto f_stat = reinterpret_cast<decltype(&hsa_amd_memory_copy_engine_status)>(sym("hsa_amd_memory_copy_engine_status"));
	L.signal_create = reinterpret_cast<decltype(L.signal_create)>(sym("hsa_signal_create"));
	L.signal_destroy = reinterpret_cast<decltype(L.signal_destroy)>(sym("hsa_signal_destroy"));
	L.signal_store = reinterpret_cast<decltype(L.signal_store)>(sym("hsa_signal_store_relaxed"));
	L.signal_wait = reinterpret_cast<decltype(L.signal_wait)>(sym("hsa_signal_wait_scacquire"));
	L.signal_load = reinterpret_cast<decltype(L.signal_load)>(sym("hsa_signal_load_relaxed"));
	L.copy_on_engine = reinterpret_cast<decltype(L.copy_on_engine)>(sym("hsa_amd_memory_async_copy_on_engine"));
	if (!f_init || !f_iter || !f_info || !f_pref || !f_stat || !L.signal_create || !L.signal_destroy || !L.signal_store || !L.signal_wait || !L.signal_load || !L.copy_on_engine) return;
	if (f_init() != HSA_STATUS_SUCCESS) return; // (reference counted: HIP holds the runtime up already)
	// the GPU agent with this device's PCI address, and a CPU agent
	char bdf[64] = {0};
	if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), c->device) != hipSuccess) { (void)hipGetLastError(); return; }
	unsigned dom = 0, bus = 0, dev = 0, fn = 0;
	if (sscanf(bdf, "%x:%x:%x.%x", &dom, &bus, &dev, &fn) != 4) return;
	struct Find { decltype(f_info) info; uint32_t want_bdf, want_dom; hsa_agent_t gpu, cpu; bool have_gpu, have_cpu; } F{f_info, (bus << 8) | (dev << 3) | fn, dom, {}, {}, false, false};
	f_iter([](hsa_agent_t a, void *p) -> hsa_status_t {
		Find &F = *static_cast<Find *>(p);
		hsa_device_type_t t;
		if (F.info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
		if (t == HSA_DEVICE_TYPE_CPU && !F.have_cpu) { F.cpu = a; F.have_cpu = true; }
		if (t == HSA_DEVICE_TYPE_GPU && !F.have_gpu) {
			uint32_t b = 0, d = 0;
			if (F.info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &b) == HSA_STATUS_SUCCESS && F.info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &d) == HSA_STATUS_SUCCESS && b == F.want_bdf && d == F.want_dom) {
				F.gpu = a; F.have_gpu = true;
			}
		}
		return HSA_STATUS_SUCCESS;
	}, &F);
	if (!F.have_gpu || !F.have_cpu) return;
	// candidates per direction: the engines the runtime prefers for it, then the others among the four lowest it reports (the ones beside the PCIe root on this chip; an older
	// runtime - the one a PyTorch wheel carries - names no preference at all)
	auto candidates = [&](hsa_agent_t dst, hsa_agent_t src, LinkCopy::Dir &D) {
		uint32_t pref = 0, avail = 0;
		if (f_pref(dst, src, &pref) != HSA_STATUS_SUCCESS) pref = 0;
		if (f_stat(dst, src, &avail) != HSA_STATUS_SUCCESS) avail = 0;
		for (int pass = 0; pass < 2; ++pass)
			for (uint32_t b = 0; b < 16; ++b) {
				const uint32_t bit = 1u << b;
				const bool take = pass == 0 ? (pref & bit) != 0 : (!(pref & bit) && (avail & bit) && b < 4);
				if (take) { D.cand.push_back(bit); D.rate.push_back(0.0); }
			}
	};
	L.gpu = F.gpu; L.cpu = F.cpu;
	candidates(F.cpu, F.gpu, L.to_host);
	candidates(F.gpu, F.cpu, L.to_device);
	if (L.to_host.cand.empty() || L.to_device.cand.empty()) return;
	if (L.to_host.cand.size() > 1 && L.to_host.cand[0] == L.to_device.cand[0]) L.to_host.cur = 1; // the two directions start on engines of their own (both run at once: tables out, chunks in)
	L.eng_to_host = L.to_host.cand[L.to_host.cur]; L.eng_to_device = L.to_device.cand[L.to_device.cur];
	{ // timestamps of the copies (the runtime's own profiling of async copies): without them the engines stay where they start
		auto f_prof = reinterpret_cast<decltype(&hsa_amd_profiling_async_copy_enable)>(sym("hsa_amd_profiling_async_copy_enable"));
		auto f_sys = reinterpret_cast<decltype(&hsa_system_get_info)>(sym("hsa_system_get_info"));
		L.copy_time = reinterpret_cast<decltype(L.copy_time)>(sym("hsa_amd_profiling_get_async_copy_time"));
		uint64_t hz = 0;
		if (f_prof && f_sys && L.copy_time && f_prof(true) == HSA_STATUS_SUCCESS && f_sys(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &hz) == HSA_STATUS_SUCCESS && hz) L.ticks_per_s = (double)hz;
		else L.copy_time = nullptr;
	}
	L.ok = true;
	if (getenv("SSV_TIMING")) fprintf(stderr, "[timing] (host link: SDMA engines by name - to the host 0x%x of %zu candidates, to the device 0x%x of %zu%s)\n", L.eng_to_host, L.to_host.cand.size(), L.eng_to_device,
	                                  L.to_device.cand.size(), L.copy_time ? ", copies timed" : "");
}

// one copy on the direction's engine; `sig` loses one when it has landed.  false: not done (the caller copies the runtime's way)
static bool link_copy(ssv_ctx *c, void *dst, const void *src, size_t bytes, bool to_host, hsa_signal_t sig)
{
	const LinkCopy &L = c->link;
	if (!L.ok || !bytes) return false;
	return L.copy_on_engine(dst, to_host ? L.cpu : L.gpu, src, to_host ? L.gpu : L.cpu, bytes, 0, nullptr, sig, (hsa_amd_sdma_engine_id_t)(to_host ? L.eng_to_host : L.eng_to_device), true) == HSA_STATUS_SUCCESS;
}

// is [p, p + bytes) page-locked host memory the runtime knows (hipHostMalloc, hipHostRegister)?  Only such memory may be handed to an SDMA engine by address
static bool link_host_ok(const void *p)
{
	hipPointerAttribute_t a;
	memset(&a, 0, sizeof(a));
	if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
	return a.type == hipMemoryTypeHost;
}

static void link_wait(ssv_ctx *c, hsa_signal_t sig)
{
	while (c->link.signal_wait(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) >= 1) {}
}

// a large copy has landed (its own signal: `sig`): what rate did its engine give, and should the direction move on?
static void link_timed(ssv_ctx *c, hsa_signal_t sig, size_t bytes, bool to_host)
{
	LinkCopy &L = c->link;
	if (!L.copy_time || bytes < ((size_t)8 << 20)) return;
	hsa_amd_profiling_async_copy_time_t t{};
	if (L.copy_time(sig, &t) != HSA_STATUS_SUCCESS || t.end <= t.start) return;
	LinkCopy::Dir &D = to_host ? L.to_host : L.to_device;
	const double rate = (double)bytes / ((double)(t.end - t.start) / L.ticks_per_s) / 1e9;
	D.rate[D.cur] = rate;
	double best = 0;
	for (double r : D.rate) best = std::max(best, r);
	if (rate >= 0.85 * best && rate >= 40.0) return; // (a PCIe Gen5 x16 link gives 55-57 GB/s)
	// well below what this direction has seen (or what the link should give): an engine not tried yet, else the one whose last copy was the fastest
	size_t next = D.cur;
	for (size_t k = 0; k < D.cand.size(); ++k) if (D.rate[k] == 0.0) { next = k; break; }
	if (next == D.cur) for (size_t k = 0; k < D.cand.size(); ++k) if (D.rate[k] > D.rate[next]) next = k;
	if (next != D.cur) {
		if (getenv("SSV_TIMING")) fprintf(stderr, "[timing] (host link: %s at %.1f GB/s on engine 0x%x: on to engine 0x%x)\n", to_host ? "to the host" : "to the device", rate, D.cand[D.cur], D.cand[next]);
		D.cur = next;
		(to_host ? L.eng_to_host : L.eng_to_device) = D.cand[next];
	} else if (best > rate) D.rate[D.cur] = rate; // (everything is slow right now: stay, and let the stale best rates of the others age)
	for (size_t k = 0; k < D.cand.size(); ++k) if (k != D.cur && D.rate[k] > 0.0) D.rate[k] = std::max(rate, D.rate[k] * 0.9); // (what an engine gave a while ago counts for less and less)
}

static void table_link_wait(ssv_ctx *c, ssv_ctx::TableSet &T)
{
	link_wait(c, T.copied_sig);
	link_wait(c, T.big_sig);
	if (T.big_bytes) { link_timed(c, T.big_sig, T.big_bytes, true); T.big_bytes = 0; }
}

int ensure_host(ssv_ctx *c, HBuf &b, size_t bytes)
{
	if (bytes <= b.cap) return SSV_OK;
	if (b.p) HIPCHECK(c, pinned_delete(b.p));
	size_t ncap = (std::max(bytes, b.cap + b.cap / 2) + 255) & ~(size_t)255;
	b.p = nullptr; b.cap = 0;
	HIPCHECK(c, pinned_new(&b.p, ncap, hipHostMallocDefault));
	b.cap = ncap;
	return SSV_OK;
}

// `bytes` of device memory that stay where they are until the arena is rewound
int arena_alloc(ssv_ctx *c, Arena &a, size_t bytes, void **out)
{
	bytes = (bytes + 255) & ~(size_t)255;
	while (a.cur < a.chunks.size() && a.chunks[a.cur].cap - a.used < bytes) { ++a.cur; a.used = 0; }
	if (a.cur == a.chunks.size()) {
		DBuf b;
		const size_t cap = std::max<size_t>(bytes, (size_t)64 << 20);
		HIPCHECK(c, dev_malloc(c, &b.p, cap));
		b.cap = cap;
		a.chunks.push_back(b);
		a.used = 0;
	}
	*out = reinterpret_cast<uint8_t *>(a.chunks[a.cur].p) + a.used;
	a.used += bytes;
	return SSV_OK;
}

template <typename T> T *P(DBuf &b) { return reinterpret_cast<T *>(b.p); }
template <typename T> T *P(HBuf &b) { return reinterpret_cast<T *>(b.p); }

struct ProfScope {
	ssv_ctx *c;
	int id;
	bool on;
	hipEvent_t a{}, b{};
	int64_t units;
	ProfScope(ssv_ctx *ctx, int id_, int64_t units_) : c(ctx), id(id_), units(units_)
	{
		on = c->prof_mode == 1 || (c->prof_mode == 2 && (id == P_CLIP_SCAN || id == P_GETSV_SCAN));
		if (!on) return;
		for (hipEvent_t *e : {&a, &b}) {
			if (!c->prof_pool.empty()) { *e = c->prof_pool.back(); c->prof_pool.pop_back(); }
			else if (hipEventCreate(e) != hipSuccess) { on = false; return; }
		}
		(void)hipEventRecord(a, c->st);
	}
	~ProfScope()
	{
		if (!on) return;
		(void)hipEventRecord(b, c->st);
		c->prof_recs.push_back({id, a, b, units});
	}
};

void prof_collect(ssv_ctx *c)
{
	if (c->prof_recs.empty()) return;
	(void)hipStreamSynchronize(c->st);
	for (ProfRec &r : c->prof_recs) {
		float ms = 0;
		if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
			c->prof_ms[r.id] += ms; c->prof_launches[r.id] += 1; c->prof_units[r.id] += r.units;
		}
		c->prof_pool.push_back(r.a); c->prof_pool.push_back(r.b);
	}
	c->prof_recs.clear();
}

// persistent streaming kernels: the grid is the number of workgroups that are resident at once (a larger grid queues the surplus
// behind the first wave of workgroups and stretches the kernel).  clip_scan: 106 SGPRs -> 6 workgroups of 256 threads per CU;
// getsv_scan: 104 VGPRs (software-pipelined loads) -> 4 per CU.  Overrides for experiments: SSV_CLIP_SCAN_BLOCKS, SSV_GETSV_SCAN_BLOCKS.
inline unsigned scan_blocks(int64_t ntiles, const char *env, int64_t dflt)
{
	const char *e = getenv(env);
	int64_t cfg = e ? atoll(e) : dflt;
	return (unsigned)std::max<int64_t>(1, std::min<int64_t>(ntiles, std::min<int64_t>(cfg, CS_MAX_BLOCKS)));
}

inline unsigned grid_for(int64_t n, int per_block) { return (unsigned)std::max<int64_t>(1, (n + per_block - 1) / per_block); }

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
bool aligned64(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 63) == 0; }

// Copies of a host batch into staging set `set`, issued on `st`: with `rec` the batch ships lines + hot columns + variable parts; without,
// the classic columns (the lines are then built on the device, k_build_rec).
static int upload_host_batch(ssv_ctx *c, const ssv_batch_t *b, int set, hipStream_t st)
{
	const size_t n = (size_t)b->n;
	ssv_ctx::StageSet &S = c->ss[set];
	struct { const void *src; size_t bytes; } f[15] = {
		{b->tid, n * 4}, {b->pos, n * 4}, {b->rec ? nullptr : b->flag, n * 2}, {b->rec ? nullptr : b->mapq, n}, {b->n_cigar, n * 2}, {b->rec ? nullptr : b->l_qseq, n * 4},
		{b->rec ? nullptr : b->mtid, n * 4}, {b->rec ? nullptr : b->mpos, n * 4}, {b->rec ? nullptr : b->isize, n * 4}, {b->rec ? nullptr : b->cigar_off, n * 4},
		{b->cigar, (size_t)b->n_cigar_total * 4}, {b->rec ? nullptr : b->xc, b->xc ? n : 0}, {b->rec ? nullptr : b->seq_off, n * 8}, {b->seqqual, (size_t)b->seqqual_bytes},
		{b->cigar_ends, n}};
	for (int k = 0; k < 15; ++k) {
		if (!f[k].src) continue;
		CHECK(ensure(c, S.col[k], f[k].bytes + 16));
		if (f[k].bytes) HIPCHECK(c, hipMemcpyAsync(S.col[k].p, f[k].src, f[k].bytes, hipMemcpyHostToDevice, st));
	}
	CHECK(ensure(c, S.col[10], 16)); CHECK(ensure(c, S.col[13], 16));
	CHECK(ensure(c, S.rec, n * sizeof(ssv_record) + 64));
	if (b->rec && n) HIPCHECK(c, hipMemcpyAsync(S.rec.p, b->rec, n * sizeof(ssv_record), hipMemcpyHostToDevice, st));
	return SSV_OK;
}

// the device view of a host batch staged in set `set`
static void staged_view(ssv_ctx *c, const ssv_batch_t *b, int set, DevBatch &d, SoaCols &s)
{
	ssv_ctx::StageSet &S = c->ss[set];
	d.tid = P<int32_t>(S.col[0]); d.pos = P<int32_t>(S.col[1]); d.n_cigar = P<uint16_t>(S.col[4]); d.cigar = P<uint32_t>(S.col[10]); d.seqqual = P<uint8_t>(S.col[13]);
	d.ends = b->cigar_ends ? P<uint8_t>(S.col[14]) : nullptr;
	d.rec = P<ssv_record>(S.rec);
	s.tid = d.tid; s.pos = d.pos; s.flag = P<uint16_t>(S.col[2]); s.mapq = P<uint8_t>(S.col[3]); s.n_cigar = d.n_cigar; s.l_qseq = P<int32_t>(S.col[5]); s.mtid = P<int32_t>(S.col[6]);
	s.mpos = P<int32_t>(S.col[7]); s.isize = P<int32_t>(S.col[8]); s.cigar_off = P<uint32_t>(S.col[9]); s.cigar = d.cigar; s.xc = b->xc ? P<uint8_t>(S.col[11]) : nullptr;
	s.seq_off = P<uint64_t>(S.col[12]);
}

static int check_batch(ssv_ctx *c, const ssv_batch_t *b)
{
	if (!b || b->n < 0 || b->n >= (1ll << 31)) { c->err = "bad batch"; return SSV_E_ARG; }
	const bool has_soa = b->flag && b->mapq && b->l_qseq && b->mtid && b->mpos && b->isize && b->cigar_off && b->seq_off;
	if (b->n > 0 && (!b->tid || !b->pos || !b->n_cigar || (!b->rec && !has_soa))) { c->err = "batch with null arrays"; return SSV_E_ARG; }
	const int mem = b->mem & ~(int)SSV_MEM_PERSISTENT;
	if (mem != SSV_MEM_DEVICE && b->mem != SSV_MEM_HOST) { c->err = "bad batch.mem"; return SSV_E_ARG; }
	return SSV_OK;
}

// Make the batch visible to the kernels: device batches are used in place, host batches are copied to HBM (or were, ssv_batch_prefetch); a
// batch that comes as structure-of-arrays columns only is transposed into record lines (ssv_record) on the device.
int stage_batch(ssv_ctx *c, const ssv_batch_t *b, DevBatch &d, bool keep_announced = false)
{
	CHECK(check_batch(c, b));
	d.n = b->n; d.max_ref_span = b->max_ref_span;
	const size_t n = (size_t)b->n;
	SoaCols s{};
	ssv_record *rec_dst = nullptr;
	if ((b->mem & ~(int)SSV_MEM_PERSISTENT) == SSV_MEM_DEVICE) {
		if (!aligned16(b->tid) || !aligned16(b->pos) || !aligned16(b->n_cigar) || (b->rec && !aligned64(b->rec))) {
			c->err = "device batch arrays must be 16-byte aligned (rec: 64-byte aligned)"; return SSV_E_ARG;
		}
		if (b->cigar_ends && !aligned16(b->cigar_ends)) { c->err = "device batch arrays must be 16-byte aligned (rec: 64-byte aligned)"; return SSV_E_ARG; }
		d.tid = b->tid; d.pos = b->pos; d.n_cigar = b->n_cigar; d.cigar = b->cigar; d.seqqual = b->seqqual; d.rec = b->rec; d.ends = b->cigar_ends;
		if (d.rec || n == 0) return SSV_OK;
		s.tid = b->tid; s.pos = b->pos; s.flag = b->flag; s.mapq = b->mapq; s.n_cigar = b->n_cigar; s.l_qseq = b->l_qseq; s.mtid = b->mtid; s.mpos = b->mpos; s.isize = b->isize;
		s.cigar_off = b->cigar_off; s.cigar = b->cigar; s.xc = b->xc; s.seq_off = b->seq_off;
		CHECK(ensure(c, c->ss[2].rec, n * sizeof(ssv_record) + 64));
		rec_dst = P<ssv_record>(c->ss[2].rec);
	} else {
		if (!c->pf.empty()) {
			// announced batches are consumed in the order they were announced
			const ssv_ctx::Prefetched f = c->pf.front();
			if (memcmp(&f.b, b, sizeof(*b)) != 0) { c->err = "a prefetched batch is pending: the next scan call must be given that batch"; return SSV_E_STATE; }
			if (!keep_announced) c->pf.pop_front(); // (a scan of a leading part of the batch leaves it staged for the scan of the rest)
			staged_view(c, b, f.set, d, s);
			HIPCHECK(c, hipEventSynchronize(c->ss[f.set].ready)); // the caller may recycle the host arrays once the scan call returns
		} else {
			ProfScope ps(c, P_H2D, b->n);
			CHECK(upload_host_batch(c, b, 2, c->st));
			staged_view(c, b, 2, d, s);
			HIPCHECK(c, hipStreamSynchronize(c->st)); // same promise (copies from pinned host arrays are asynchronous)
		}
		if (b->rec || n == 0) return SSV_OK;
		rec_dst = const_cast<ssv_record *>(d.rec);
	}
	k_build_rec<<<grid_for(b->n, BLOCK), BLOCK, 0, c->st>>>(s, b->n, rec_dst);
	HIPCHECK(c, hipGetLastError());
	d.rec = rec_dst;
	return SSV_OK;
}

int ensure_events(ssv_ctx *c, int64_t need)
{
	if (need <= c->ev_cap) return SSV_OK;
	int64_t ncap = std::max<int64_t>(need, c->ev_cap + c->ev_cap / 2);
	ncap = std::max<int64_t>(ncap, 1 << 16);
	CHECK(ensure(c, c->ev_meta, (size_t)ncap * 8, true, (size_t)c->n_events * 8));
	CHECK(ensure(c, c->ev_idx, (size_t)ncap * 4, true, (size_t)c->n_events * 4));
	CHECK(ensure(c, c->key_l, (size_t)ncap * 8, true, (size_t)c->n_l * 8));
	CHECK(ensure(c, c->val_l, (size_t)ncap * 4, true, (size_t)c->n_l * 4));
	CHECK(ensure(c, c->key_r[0], (size_t)ncap * 8, true, (size_t)c->n_r * 8));
	CHECK(ensure(c, c->val_r[0], (size_t)ncap * 4, true, (size_t)c->n_r * 4));
	c->ev_cap = ncap;
	return SSV_OK;
}

// ssv_batch_t.tid_runs -> the kernel's table (checked: a wrong run list would silently move records to another contig)
int fill_runs(ssv_ctx *c, const ssv_batch_t *b, RunTab &R)
{
	memset(&R, 0, sizeof(R));
	if (!b->tid_runs || b->n_tid_runs <= 0 || b->n_tid_runs > RUN_MAX || getenv("SSV_NO_TID_RUNS")) return SSV_OK;
	const int64_t k = b->n_tid_runs;
	if (b->tid_runs[0].first != 0) { c->err = "tid_runs must start at record 0"; return SSV_E_ARG; }
	for (int64_t i = 0; i < k; ++i) {
		if (i && b->tid_runs[i].first <= b->tid_runs[i - 1].first) { c->err = "tid_runs must be strictly increasing"; return SSV_E_ARG; }
		if (b->tid_runs[i].first >= b->n) { c->err = "tid_runs reach past the batch"; return SSV_E_ARG; }
		R.first[i] = b->tid_runs[i].first; R.tid[i] = b->tid_runs[i].tid;
	}
	R.first[k] = b->n; R.n = (int32_t)k;
	return SSV_OK;
}

__global__ void k_max_span(DevBatch b, int *out)
{
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int span = 0;
	if (i < b.n) {
		const RecLine r = rec_load(b.rec, i);
		const int n = r.n_cigar();
		long long s = 0;
		for (int k = 0; k < n; ++k) { uint32_t x = r.op(b.cigar, k); int op = (int)(x & 15u); if (op == C_M || op == C_D || op == C_N || op == C_EQ || op == C_X) s += x >> 4; }
		span = s > 0x7fffffff ? 0x7fffffff : (int)s;
	}
	span = wave_max(span);
	if (lane_id() == 0 && span > 0 && span > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, span); // (look first: one address, a wavefront each)
}

} // namespace

// =====================================================================================================================
extern "C" {

int ssv_abi_version(void) { return SSV_ABI_VERSION; }

int ssv_device_count(void)
{
	int n = 0;
	return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int ssv_ctx_create(int device, ssv_ctx **out)
{
	if (!out) return SSV_E_ARG;
	*out = nullptr;
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev <= 0) {
		g_create_error = std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0") + "); libseeksv_hip has no CPU path";
		return SSV_E_NODEVICE;
	}
	if (device < 0 || device >= ndev) { g_create_error = "device ordinal out of range"; return SSV_E_ARG; }
	if ((e = hipSetDevice(device)) != hipSuccess) { g_create_error = hipGetErrorString(e); return SSV_E_NODEVICE; }
	ssv_ctx *c = new ssv_ctx();
	c->device = device;
	if ((e = hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking)) != hipSuccess) { g_create_error = hipGetErrorString(e); delete c; return SSV_E_NODEVICE; }
	if (hipStreamCreateWithFlags(&c->st_copy, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&c->st_h2d, hipStreamNonBlocking) != hipSuccess ||
	    hipEventCreateWithFlags(&c->ev_st, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ss[0].ready, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&c->ss[1].ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&c->tab[0].packed_ev, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->tab[1].packed_ev, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&c->tab[0].copied, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&c->tab[1].copied, hipEventDisableTiming) != hipSuccess) {
		g_create_error = "cannot create the copy stream / events"; ssv_ctx_destroy(c); return SSV_E_NODEVICE;
	}
	link_init(c);
	if (c->link.ok) for (auto &t : c->tab) if (c->link.signal_create(0, 0, nullptr, &t.copied_sig) != HSA_STATUS_SUCCESS || c->link.signal_create(0, 0, nullptr, &t.big_sig) != HSA_STATUS_SUCCESS) { c->link.ok = false; break; }
	*out = c;
	return SSV_OK;
}

static void bamdec_free(ssv_ctx *c); // bamdec_api.inc
static void realign_free(ssv_ctx *c); // realign_api.inc

void ssv_ctx_destroy(ssv_ctx *c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->st);

	for (auto &t : c->tab) if (t.in_flight && t.via_link) { table_link_wait(c, t); t.in_flight = false; } // (a table still on its way out: its buffers go below)
	bamdec_free(c);
	realign_free(c);
	if (c->h_batch.p) (void)pinned_delete(c->h_batch.p);
	if (c->h_qual_lut.p) (void)pinned_delete(c->h_qual_lut.p);
	if (c->h_pair_lut.p) (void)pinned_delete(c->h_pair_lut.p);
	// every DBuf / HBuf member
	if (c->st_h2d) { (void)hipStreamSynchronize(c->st_h2d); (void)hipStreamDestroy(c->st_h2d); }
	DBuf *dbufs[] = {&c->tile_cnt, &c->tile_off, &c->tile_base, &c->scan_scratch, &c->scan_scratch64, &c->counters, &c->d_last_tid, &c->stage, &c->cand, &c->cand_cnt, &c->cand_off, &c->kv_stage, &c->ends_buf,
	                 &c->ev, &c->ev_meta, &c->ev_idx, &c->key_l, &c->val_l, &c->key_r[0], &c->key_r[1], &c->val_r[0], &c->val_r[1], &c->g_seq_bytes, &c->g_cig_ops, &c->g_seq_off, &c->g_cig_off, &c->keys2[0],
	                 &c->keys2[1], &c->vals2[0], &c->vals2[1], &c->evs, &c->cum_l, &c->cum_r, &c->ghist, &c->c_support, &c->c_ll, &c->c_lr, &c->c_cig_ev, &c->c_qmiss, &c->c_mflag, &c->c_mslot, &c->c_mlist, &c->c_bflag, &c->c_boff, &c->c_blist, &c->c_dlist, &c->bins4_tab, &c->c_strings,
	                 &c->slot_cnt, &c->slot_bytes, &c->tile_sums, &c->o_slowlist, &c->o_desc, &c->totals, &c->qual_lut, &c->pair_lut, &c->qual_seen, &c->isz_vals, &c->isz_acc, &c->isz_tmp,
	                 &c->gs_djunc, &c->gs_counts, &c->gs_wtid, &c->gs_wbeg, &c->gs_wend, &c->gs_woff, &c->gs_diff, &c->gs_tilemap, &c->gs_tile_win, &c->gs_tile_junc, &c->dense_list, &c->cap_flags, &c->cap_deep, &c->cap_carry,
	                 &c->cap_ring, &c->cap_ring_tmp, &c->cap_tail[0][0], &c->cap_tail[0][1], &c->cap_tail[0][2], &c->cap_tail[0][3], &c->cap_tail[1][0], &c->cap_tail[1][1], &c->cap_tail[1][2], &c->cap_tail[1][3],
	                 &c->gs_ctgoff, &c->gs_maxdepth, &c->gs_span, &c->q_tid, &c->q_beg, &c->q_end, &c->q_out64, &c->q_out32};
	for (DBuf *b : dbufs) if (b->p) (void)hipFree(b->p);
	for (auto &a : c->arenas) if (a.base) (void)hipFree(a.base);
	for (auto &a : c->spare_arenas) (void)hipFree(a.base);
	{ size_t cap = 0; if (uint8_t *b = arena_ahead_take(c, &cap)) (void)hipFree(b); }
	for (auto &S : c->ss) {
		for (DBuf &b : S.col) if (b.p) (void)hipFree(b.p);
		if (S.rec.p) (void)hipFree(S.rec.p);
		if (S.ready) (void)hipEventDestroy(S.ready);
	}
	if (c->ev_st) (void)hipEventDestroy(c->ev_st);
	for (DBuf &b : c->blob.chunks) if (b.p) (void)hipFree(b.p);
	HBuf *hbufs[] = {&c->h_counters, &c->h_totals, &c->h_q};
	for (HBuf *b : hbufs) if (b->p) (void)pinned_delete(b->p);
	for (auto &t : c->tab) {
		DBuf *td[] = {&t.o_tid, &t.o_pos, &t.o_side, &t.o_support, &t.o_ll, &t.o_lr, &t.o_qmiss, &t.o_ncig, &t.o_stroff, &t.o_cigoff, &t.o_str, &t.o_cig, &t.o_len, &t.o_sup, &t.o_nc, &t.o_runs, &t.o_exc};
		HBuf *th[] = {&t.h_tid, &t.h_pos, &t.h_side, &t.h_support, &t.h_ll, &t.h_lr, &t.h_qmiss, &t.h_stroff, &t.h_cigoff, &t.h_ncig, &t.h_str, &t.h_cig, &t.h_len, &t.h_sup, &t.h_nc, &t.h_runs, &t.h_exc};
		for (DBuf *b : td) if (b->p) (void)hipFree(b->p);
		for (HBuf *b : th) if (b->p) (void)pinned_delete(b->p);
		if (t.copied) (void)hipEventDestroy(t.copied);
		if (c->link.signal_destroy && t.copied_sig.handle) (void)c->link.signal_destroy(t.copied_sig);
		if (c->link.signal_destroy && t.big_sig.handle) (void)c->link.signal_destroy(t.big_sig);
		if (t.packed_ev) (void)hipEventDestroy(t.packed_ev);
	}
	if (c->st_copy) { (void)hipStreamSynchronize(c->st_copy); (void)hipStreamDestroy(c->st_copy); }
	if (c->ev_packed) (void)hipEventDestroy(c->ev_packed);
	for (ProfRec &r : c->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
	for (hipEvent_t e : c->prof_pool) (void)hipEventDestroy(e);
	(void)hipStreamDestroy(c->st);
	delete c;
}

int ssv_sync(ssv_ctx *c)
{
	if (!c) return SSV_E_ARG;
	HIPCHECK(c, hipStreamSynchronize(c->st));
	return SSV_OK;
}

int ssv_host_alloc(size_t bytes, void **p)
{
	if (!p) return SSV_E_ARG;
	*p = nullptr;
	hipError_t e = pinned_new(p, bytes ? bytes : 1, hipHostMallocPortable);
	if (e != hipSuccess) { g_create_error = std::string("page-locked allocation: ") + hipGetErrorString(e); *p = nullptr; return SSV_E_NOMEM; }
	return SSV_OK;
}

int ssv_host_free(void *p)
{
	if (p && pinned_delete(p) != hipSuccess) return SSV_E_HIP;
	return SSV_OK;
}

int ssv_host_register(void *p, size_t bytes)
{
	if (!p || !bytes) return SSV_E_ARG;
	const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable);
	if (e != hipSuccess) { (void)hipGetLastError(); g_create_error = std::string("hipHostRegister: ") + hipGetErrorString(e); return SSV_E_HIP; }
	return SSV_OK;
}

int ssv_host_bind_near(void *p, size_t bytes, int device)
{
	bind_near(p, bytes, device);
	return SSV_OK;
}

int ssv_host_unregister(void *p)
{
	if (p && hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return SSV_E_HIP; }
	return SSV_OK;
}

int ssv_batch_prefetch(ssv_ctx *c, const ssv_batch_t *b)
{
	if (!c || !b) return SSV_E_ARG;
	CHECK(check_batch(c, b));
	if (b->mem != SSV_MEM_HOST) { c->err = "ssv_batch_prefetch takes host batches"; return SSV_E_ARG; }
	if (c->pf.size() >= 2) { c->err = "two prefetched batches are pending already"; return SSV_E_STATE; }
	HIPCHECK(c, hipSetDevice(c->device));
	const int set = (int)(c->pf_count & 1);
	// the set was last read by the kernels of the batch two announcements ago, all of them launched on st by now
	HIPCHECK(c, hipEventRecord(c->ev_st, c->st));
	HIPCHECK(c, hipStreamWaitEvent(c->st_h2d, c->ev_st, 0));
	CHECK(upload_host_batch(c, b, set, c->st_h2d));
	HIPCHECK(c, hipEventRecord(c->ss[set].ready, c->st_h2d));
	c->pf.push_back(ssv_ctx::Prefetched{*b, set});
	++c->pf_count;
	return SSV_OK;
}

int ssv_batch_prefetch_drop(ssv_ctx *c)
{
	if (!c) return SSV_E_ARG;
	HIPCHECK(c, hipSetDevice(c->device));
	if (!c->pf.empty()) HIPCHECK(c, hipStreamSynchronize(c->st_h2d)); // their host arrays are the caller's again when this returns
	c->pf.clear();
	return SSV_OK;
}

const char *ssv_last_error(const ssv_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }
void *ssv_stream(ssv_ctx *c) { return c ? (void *)c->st : nullptr; }

// ---------------------------------------------------------------------------------------------------------------------
// getclip
// ---------------------------------------------------------------------------------------------------------------------

int ssv_clip_begin(ssv_ctx *c, const ssv_clip_params *p)
{
	if (!c || !p) return SSV_E_ARG;
	// (the announced batches stay announced: a pass may end and the next begin in the middle of a stream of batches - and of a batch,
	// ssv_clip_scan_range; a caller that abandons a stream says so with ssv_batch_prefetch_drop)
	HIPCHECK(c, hipSetDevice(c->device));
	c->clip_p = *p;
	c->clip_active = true;
	c->n_events = 0; c->n_l = 0; c->n_r = 0; c->n_long = 0; c->sum_ncig = 0; c->max_lq = 0; c->max_ncig = 0; c->ev_slots = 0;
	c->blob.cur = 0; c->blob.used = 0;
	CHECK(ensure(c, c->d_last_tid, 16));
	CHECK(ensure(c, c->counters, sizeof(ClipCounters)));
	CHECK(ensure_host(c, c->h_counters, sizeof(ClipCounters)));
	HIPCHECK(c, hipMemsetAsync(c->d_last_tid.p, 0, 16, c->st));
	int *h_lt = P<int>(c->h_counters);
	*h_lt = p->initial_last_tid; // 0 in the reference, clip_reads.h:407
	HIPCHECK(c, hipMemcpyAsync(c->d_last_tid.p, h_lt, 4, hipMemcpyHostToDevice, c->st));
	HIPCHECK(c, hipStreamSynchronize(c->st));
	return SSV_OK;
}

int ssv_clip_scan(ssv_ctx *c, const ssv_batch_t *b) { return b ? ssv_clip_scan_range(c, b, 0, b->n) : SSV_E_ARG; }

int ssv_clip_scan_range(ssv_ctx *c, const ssv_batch_t *b, int64_t rec_begin, int64_t rec_end)
{
	if (!c || !b) return SSV_E_ARG;
	if (!c->clip_active) { c->err = "ssv_clip_scan before ssv_clip_begin"; return SSV_E_STATE; }
	if (rec_begin < 0 || rec_begin > rec_end || rec_end > b->n) { c->err = "ssv_clip_scan_range: bad record range"; return SSV_E_ARG; }
	HIPCHECK(c, hipSetDevice(c->device));
	if (b->n == 0) return SSV_OK;
	DevBatch d;
	CHECK(stage_batch(c, b, d, rec_end < b->n));
	if (!d.cigar) { c->err = "batch without cigar"; return SSV_E_ARG; }
	d.n = rec_end; // what lies behind the range is not looked at (nor does it move the contig-switch state)
	if (rec_end == 0) return SSV_OK;
	if (!d.ends) { // the batcher did not fill the cigar_ends column: built from the lines
		CHECK(ensure(c, c->ends_buf, (size_t)d.n + 16));
		k_build_ends<<<grid_for(d.n, BLOCK), BLOCK, 0, c->st>>>(d, P<uint8_t>(c->ends_buf));
		HIPCHECK(c, hipGetLastError());
		d.ends = P<uint8_t>(c->ends_buf);
	}
	const bool persistent = b->mem == (SSV_MEM_DEVICE | SSV_MEM_PERSISTENT);
	const int64_t ntiles = (d.n + CC_TILE - 1) / CC_TILE;
	const unsigned grid = scan_blocks(ntiles, "SSV_CLIP_SCAN_BLOCKS", 256 * 6);
	CHECK(ensure(c, c->tile_cnt, ntiles * 4));
	CHECK(ensure(c, c->tile_off, ntiles * 4));
	CHECK(ensure(c, c->tile_base, ntiles * 4));
	CHECK(ensure(c, c->scan_scratch, scan_scratch_elems(std::max<int64_t>(ntiles, 1)) * 4));
	if (c->stage_cap == 0) c->stage_cap = std::max<int64_t>(1 << 16, d.n / 8);
	ClipCounters *hc = P<ClipCounters>(c->h_counters);
	ClipCounters *dc = P<ClipCounters>(c->counters);
	for (int attempt = 0;; ++attempt) {
		const int64_t block_cap = (c->stage_cap + grid - 1) / grid;
		CHECK(ensure(c, c->stage, (size_t)block_cap * grid * 4));
		HIPCHECK(c, hipMemsetAsync(c->counters.p, 0, sizeof(ClipCounters), c->st));
		ClipScanArgs a;
		a.ends = d.ends; a.n = d.n;
		a.tile_cnt = P<uint32_t>(c->tile_cnt); a.tile_off = P<uint32_t>(c->tile_off); a.stage = P<uint32_t>(c->stage); a.block_cap = block_cap;
		a.overflow = &dc->overflow; a.ntiles = ntiles;
		{
			ProfScope ps(c, P_CLIP_SCAN, d.n);
			k_clip_scan_ends<<<grid, BLOCK, 0, c->st>>>(a);
		}
		HIPCHECK(c, hipGetLastError());
		// order across tiles: exclusive scan of the tile counts; its total is the number of candidates
		exclusive_scan<uint32_t, uint32_t>(c->st, P<uint32_t>(c->tile_cnt), P<uint32_t>(c->tile_base), ntiles, 0u, P<uint32_t>(c->scan_scratch), reinterpret_cast<uint32_t *>(&dc->n_cand));
		HIPCHECK(c, hipMemcpyAsync(hc, c->counters.p, sizeof(ClipCounters), hipMemcpyDeviceToHost, c->st));
		HIPCHECK(c, hipStreamSynchronize(c->st));
		if (!hc->overflow) break;
		if (attempt > 4) { c->err = "clip staging overflow"; return SSV_E_HIP; }
		c->stage_cap = std::max<int64_t>(c->stage_cap * 4, (int64_t)(uint32_t)hc->n_cand * 4); // a workgroup's private region was too small
	}
	const int64_t ncand = (int64_t)(uint32_t)hc->n_cand;
	if (ncand > 0) {
		int64_t nb = 0, slot_base = 0;
		(void)slot_base;
		{
			ProfScope ps(c, P_CLIP_PLACE, ncand);
			const int64_t nwave = (ncand + 15) / 16; // k_clip_filter: four lanes per candidate, 16 candidates per wavefront
			CHECK(ensure(c, c->cand, ncand * 4)); CHECK(ensure(c, c->cand_cnt, ncand + 64)); CHECK(ensure(c, c->cand_off, nwave * 8));
			const int64_t nslot = 2 * (((ncand + 63) / 64) * 64); // two per candidate, whole workgroups (64 candidates each)
			if (c->ev_slots + nslot >= (1ll << 32) - 1) { c->err = "more than 2^32 event slots in one pass (the sorted permutation is 32 bits wide)"; return SSV_E_RANGE; }
			CHECK(ensure(c, c->ev, (size_t)(c->ev_slots + nslot) * sizeof(ClipEvent), true, (size_t)c->ev_slots * sizeof(ClipEvent)));
			CHECK(ensure(c, c->kv_stage, (size_t)nslot * 16));
			CHECK(ensure(c, c->scan_scratch64, scan_scratch_elems(nwave) * 8));
			CHECK(ensure_events(c, c->n_events + 2 * ncand));
			k_cand_place<<<grid_for(ntiles, WAVES_PER_BLOCK), BLOCK, 0, c->st>>>(P<uint32_t>(c->stage), P<uint32_t>(c->tile_cnt), P<uint32_t>(c->tile_off), P<uint32_t>(c->tile_base), ntiles,
			                                                                    P<uint32_t>(c->cand));
			ClipFilterArgs f;
			f.b = d; f.min_mapq = c->clip_p.min_mapq; f.save_low_quality = c->clip_p.save_low_quality; f.last_tid_in = P<int>(c->d_last_tid);
			f.use_ownership = c->clip_p.use_ownership; f.rec_begin = rec_begin;
			f.own_lo = ((long long)c->clip_p.own_lo_tid << 32) | (long long)(uint32_t)c->clip_p.own_lo_pos;
			f.own_hi = ((long long)c->clip_p.own_hi_tid << 32) | (long long)(uint32_t)c->clip_p.own_hi_pos;
			k_clip_filter<<<grid_for(ncand, BLOCK / 4), BLOCK, 0, c->st>>>(f, P<uint32_t>(c->cand), ncand, P<ClipEvent>(c->ev) + c->ev_slots, P<uint4>(c->kv_stage), P<uint8_t>(c->cand_cnt), P<uint64_t>(c->cand_off));
			exclusive_scan<uint64_t, uint64_t>(c->st, P<uint64_t>(c->cand_off), P<uint64_t>(c->cand_off), nwave, 0ull, P<uint64_t>(c->scan_scratch64), reinterpret_cast<uint64_t *>(&dc->n_new));
			EventLists L;
			L.key_l = P<uint64_t>(c->key_l); L.val_l = P<uint32_t>(c->val_l); L.key_r = P<uint64_t>(c->key_r[0]); L.val_r = P<uint32_t>(c->val_r[0]);
			L.meta = P<uint2>(c->ev_meta); L.idx = P<uint32_t>(c->ev_idx);
			k_clip_place<<<grid_for(ncand, BLOCK), BLOCK, 0, c->st>>>(P<uint4>(c->kv_stage), P<uint8_t>(c->cand_cnt), P<uint64_t>(c->cand_off), ncand, L, c->n_events, c->n_l, c->n_r, c->ev_slots);
			k_event_max<<<512, BLOCK, 0, c->st>>>(P<uint2>(c->ev_meta), c->n_events, dc);
			HIPCHECK(c, hipGetLastError());
			HIPCHECK(c, hipMemcpyAsync(hc, c->counters.p, sizeof(ClipCounters), hipMemcpyDeviceToHost, c->st));
			HIPCHECK(c, hipStreamSynchronize(c->st));
			nb = (int64_t)(uint32_t)hc->n_new;
			slot_base = c->ev_slots;
			c->ev_slots += nslot;
		}
		if (nb > 0 && !persistent) {
			// the batch's buffers may be recycled after this call: the bytes its events point at move into context memory
			ProfScope ps(c, P_CLIP_GATHER, nb);
			CHECK(ensure(c, c->g_seq_bytes, nb * 4)); CHECK(ensure(c, c->g_cig_ops, nb * 4)); CHECK(ensure(c, c->g_seq_off, nb * 8)); CHECK(ensure(c, c->g_cig_off, nb * 8));
			CHECK(ensure(c, c->scan_scratch64, scan_scratch_elems(nb) * 8));
			k_gather_sizes<<<grid_for(nb, BLOCK), BLOCK, 0, c->st>>>(P<uint2>(c->ev_meta), c->n_events, nb, P<uint32_t>(c->g_seq_bytes), P<uint32_t>(c->g_cig_ops));
			exclusive_scan<uint32_t, uint64_t>(c->st, P<uint32_t>(c->g_seq_bytes), P<uint64_t>(c->g_seq_off), nb, 0ull, P<uint64_t>(c->scan_scratch64), reinterpret_cast<uint64_t *>(&dc->seq_total));
			exclusive_scan<uint32_t, uint64_t>(c->st, P<uint32_t>(c->g_cig_ops), P<uint64_t>(c->g_cig_off), nb, 0ull, P<uint64_t>(c->scan_scratch64), reinterpret_cast<uint64_t *>(&dc->cig_total));
			HIPCHECK(c, hipMemcpyAsync(hc, c->counters.p, sizeof(ClipCounters), hipMemcpyDeviceToHost, c->st));
			HIPCHECK(c, hipStreamSynchronize(c->st));
			void *seq_dst = nullptr, *cig_dst = nullptr;
			CHECK(arena_alloc(c, c->blob, (size_t)hc->seq_total + 16, &seq_dst));
			CHECK(arena_alloc(c, c->blob, (size_t)hc->cig_total * 4 + 16, &cig_dst));
			k_clip_gather<<<grid_for(nb, GROUPS_PER_BLOCK), BLOCK, 0, c->st>>>(P<ClipEvent>(c->ev), P<uint32_t>(c->ev_idx), c->n_events, nb, P<uint64_t>(c->g_seq_off), P<uint64_t>(c->g_cig_off),
			                                                                          reinterpret_cast<uint8_t *>(seq_dst), reinterpret_cast<uint32_t *>(cig_dst));
			HIPCHECK(c, hipGetLastError());
		}
		if (nb > 0) {
			const int64_t nbr = (int64_t)(hc->n_new >> 32);
			c->n_events += nb; c->n_r += nbr; c->n_l += nb - nbr; c->n_long += (int64_t)hc->n_long;
			c->max_lq = std::max(c->max_lq, hc->max_lq);
			c->max_ncig = std::max(c->max_ncig, hc->max_ncig);
			c->sum_ncig += hc->sum_ncig;
		}
	}
	k_last_tid<<<1, BLOCK, 0, c->st>>>(d, P<int>(c->d_last_tid));
	HIPCHECK(c, hipGetLastError());
	return SSV_OK;
}

// ---- a batch kept: the decoded records of a file stay in HBM for the passes that follow ----
int ssv_batch_retain(ssv_ctx *c, const ssv_batch_t *b, ssv_batch_t *out)
{
	if (!c || !b || !out) return SSV_E_ARG;
	HIPCHECK(c, hipSetDevice(c->device));
	if ((b->mem & ~(int)SSV_MEM_PERSISTENT) != SSV_MEM_DEVICE) { c->err = "ssv_batch_retain takes device batches"; return SSV_E_ARG; } // (so no announced host batch is consumed below)
	DevBatch d;
	CHECK(stage_batch(c, b, d)); // (builds the record lines when the batch has none)
	const size_t n = (size_t)b->n;
	auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
	const size_t sz[7] = {up(n * 4 + 16), up(n * 4 + 16), up(n * 2 + 16), up(n + 16), up(n * sizeof(ssv_record) + 64), up((size_t)b->n_cigar_total * 4 + 64), up((size_t)b->seqqual_bytes + 64)};
	size_t off[8]; off[0] = 0;
	for (int k = 0; k < 7; ++k) off[k + 1] = off[k] + sz[k];
	uint8_t *slab = nullptr;
	ssv_tid_run *runs = nullptr;
	if (b->tid_runs && b->n_tid_runs > 0) { // (host memory: a copy that lives as long as the batch)
		runs = static_cast<ssv_tid_run *>(malloc((size_t)b->n_tid_runs * sizeof(ssv_tid_run)));
		if (!runs) { c->err = "out of host memory (tid_runs)"; return SSV_E_NOMEM; }
		memcpy(runs, b->tid_runs, (size_t)b->n_tid_runs * sizeof(ssv_tid_run));
	}
	ssv_ctx::RetainArena *arena = nullptr;
	struct Guard { ssv_ctx::RetainArena *&arena; size_t need; ssv_tid_run *&runs; bool keep = false; ~Guard() { if (!keep) { if (arena) { --arena->live; arena->used -= need; arena->slabs.pop_back(); } free(runs); } } } guard{arena, off[7], runs};
	static const bool timing = [] { const char *e = getenv("SSV_TIMING"); return e ? atoi(e) : 0; }() >= 2; // SSV_TIMING=2: per-chunk detail
	const auto t0 = std::chrono::steady_clock::now();
	{ // room in the newest arena, or a new one: 256 MB first, doubling up to SSV_RETAIN_ARENA_MB (4 GB), never smaller than the batch
		if (!c->arenas.empty() && c->arenas.back().cap - c->arenas.back().used >= off[7]) arena = &c->arenas.back();
		else {
			static const size_t arena_max = []() { const char *e = getenv("SSV_RETAIN_ARENA_MB"); return (size_t)(e ? atoll(e) : 4096) << 20; }();
			size_t cap = c->arenas.empty() ? std::min(arena_max, (size_t)256 << 20) : std::min(arena_max, c->arenas.back().cap * 2);
			if (cap < off[7]) cap = off[7];
			ssv_ctx::RetainArena a;
			// a spare one that is large enough (the smallest such), else a new one
			size_t pick = c->spare_arenas.size();
			for (size_t k = 0; k < c->spare_arenas.size(); ++k)
				if (c->spare_arenas[k].cap >= off[7] && (pick == c->spare_arenas.size() || c->spare_arenas[k].cap < c->spare_arenas[pick].cap)) pick = k;
			if (pick < c->spare_arenas.size()) { a.base = c->spare_arenas[pick].base; cap = c->spare_arenas[pick].cap; c->spare_arenas.erase(c->spare_arenas.begin() + (long)pick); }
			else {
				size_t acap = 0;
				if (uint8_t *b = arena_ahead_take(c, &acap)) { // the one asked for ahead - if this batch fits (else it waits among the spares)
					if (acap >= off[7]) { a.base = b; cap = acap; }
					else c->spare_arenas.push_back({b, acap});
				}
				if (!a.base) HIPCHECK(c, dev_malloc(c, reinterpret_cast<void **>(&a.base), cap));
			}
			a.cap = cap;
			c->arenas.push_back(a);
			arena = &c->arenas.back();
			// the next one, while this one fills (not when spares are waiting: a context that has been through a release has its memory)
			if (c->spare_arenas.empty()) arena_ahead_start(c, std::min(arena_max, cap * 2));
		}
		slab = arena->base + arena->used;
		arena->slabs.push_back(arena->used);
		arena->used += off[7]; ++arena->live; // (off[] are multiples of 256: every batch starts 256-byte aligned)
	}
	const auto t1 = std::chrono::steady_clock::now();
	const void *src[7] = {d.tid, d.pos, d.n_cigar, d.ends, d.rec, d.cigar, d.seqqual};
	const size_t bytes[7] = {n * 4, n * 4, n * 2, n, n * sizeof(ssv_record), (size_t)b->n_cigar_total * 4, (size_t)b->seqqual_bytes};
	if (!d.ends && n) { // no cigar_ends column: built from the lines, straight into the slab
		k_build_ends<<<grid_for(d.n, BLOCK), BLOCK, 0, c->st>>>(d, slab + off[3]);
		HIPCHECK(c, hipGetLastError());
	}
	for (int k = 0; k < 7; ++k) if (src[k] && bytes[k]) HIPCHECK(c, hipMemcpyAsync(slab + off[k], src[k], bytes[k], hipMemcpyDeviceToDevice, c->st));
	HIPCHECK(c, hipStreamSynchronize(c->st)); // the source (the decoder's buffers) may be overwritten by the next decode
	if (timing) fprintf(stderr, "[timing] (retain: %lld records, %zu bytes: hipMalloc %.4f s, copies + wait %.4f s)\n", (long long)b->n, off[7], std::chrono::duration<double>(t1 - t0).count(),
	                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
	guard.keep = true;
	memset(out, 0, sizeof(*out));
	out->n = b->n; out->mem = SSV_MEM_DEVICE | SSV_MEM_PERSISTENT; out->max_ref_span = b->max_ref_span;
	out->tid = reinterpret_cast<const int32_t *>(slab + off[0]); out->pos = reinterpret_cast<const int32_t *>(slab + off[1]); out->n_cigar = reinterpret_cast<const uint16_t *>(slab + off[2]);
	out->cigar_ends = slab + off[3]; out->rec = reinterpret_cast<const ssv_record *>(slab + off[4]); out->cigar = reinterpret_cast<const uint32_t *>(slab + off[5]);
	out->seqqual = slab + off[6]; out->n_cigar_total = b->n_cigar_total; out->seqqual_bytes = b->seqqual_bytes;
	if (runs) { out->tid_runs = runs; out->n_tid_runs = b->n_tid_runs; }
	return SSV_OK;
}

int ssv_batch_release(ssv_ctx *c, ssv_batch_t *b)
{
	if (!c || !b) return SSV_E_ARG;
	if (b->mem != (SSV_MEM_DEVICE | SSV_MEM_PERSISTENT) || !b->tid) { c->err = "not a batch of ssv_batch_retain"; return SSV_E_ARG; }
	HIPCHECK(c, hipSetDevice(c->device));
	HIPCHECK(c, hipStreamSynchronize(c->st));
	{ // the batch's arena (the slab starts with the tid column): given back to the device when its last batch goes, unless it is the newest one - that one starts over
		const uint8_t *at = reinterpret_cast<const uint8_t *>(b->tid);
		size_t k = 0;
		while (k < c->arenas.size() && !(at >= c->arenas[k].base && at < c->arenas[k].base + c->arenas[k].cap)) ++k;
		if (k == c->arenas.size()) { c->err = "not a batch of ssv_batch_retain (no arena holds it)"; return SSV_E_ARG; }
		// ... and the slab must be one that is live: a copy of a batch released before (or any pointer into an arena) would count the arena down a second
		// time and hand its memory back while other batches still live in it
		auto &slabs = c->arenas[k].slabs;
		const auto it = std::find(slabs.begin(), slabs.end(), (size_t)(at - c->arenas[k].base));
		if (it == slabs.end()) { c->err = "not a live batch of ssv_batch_retain (released before, or not the start of a batch)"; return SSV_E_ARG; }
		slabs.erase(it);
		if (--c->arenas[k].live == 0) {
			if (k + 1 == c->arenas.size()) c->arenas[k].used = 0;
			else { c->spare_arenas.push_back({c->arenas[k].base, c->arenas[k].cap}); c->arenas.erase(c->arenas.begin() + (long)k); } // (kept for the next retain: ssv_ctx::spare_arenas)
		}
	}
	free(const_cast<ssv_tid_run *>(b->tid_runs));
	memset(b, 0, sizeof(*b));
	return SSV_OK;
}

int ssv_clip_event_count(ssv_ctx *c, int64_t *n)
{
	if (!c || !n) return SSV_E_ARG;
	HIPCHECK(c, hipStreamSynchronize(c->st));
	*n = c->n_events;
	return SSV_OK;
}

// entries of the list of bases outside A/C/G/T that a compact table may carry before it falls back to 4-bit bases
static int64_t exc_cap_of(int64_t E)
{
	const char *e = getenv("SSV_EXC_CAP"); // (tests: a tiny list)
	return e ? (int64_t)atoll(e) : std::min<int64_t>(E / 2 + 65536, 0xffffffffll);
}

int ssv_clip_cluster_async(ssv_ctx *c, int64_t *n_clusters, int64_t *n_events)
{
	if (!c) return SSV_E_ARG;
	if (!c->clip_active) { c->err = "ssv_clip_cluster before ssv_clip_begin"; return SSV_E_STATE; }
	HIPCHECK(c, hipSetDevice(c->device));
	// take the table set that is not the most recent one; its previous copy (two calls ago) must have landed
	const int s_ = c->tab_cur ^ 1;
	ssv_ctx::TableSet &T = c->tab[s_];
	if (T.in_flight) { if (T.via_link) table_link_wait(c, T); else HIPCHECK(c, hipEventSynchronize(T.copied)); T.in_flight = false; }
	c->tab_cur = s_;
	const int64_t E = c->n_events, EL = c->n_l, ER = c->n_r;
	T.n_events = E; T.n_clusters = 0;
	T.packed = c->table_mode ? 1 : 0; T.qual_bits = 8; T.qual_group = 1; T.qual_radix = 0; memset(T.qual_alphabet, 0, sizeof(T.qual_alphabet));
	const bool fmt3 = c->table_mode == 3;
	T.format = c->table_mode; T.base_bits = fmt3 ? 2 : 4; T.n_runs = 0; T.n_exc = 0; T.expanded = false; T.ordered = false;
	T.cig_bytes = fmt3 ? 2 : 4; T.len_bytes = fmt3 && c->max_lq < 65536 ? 2 : 4; T.support_bytes = fmt3 ? 2 : 4; T.ncig_bytes = fmt3 ? (c->max_ncig < 256 ? 1 : 2) : 4;
	if (fmt3 && getenv("SSV_TABLE_WIDE_COLUMNS")) { T.len_bytes = 4; T.support_bytes = 4; T.ncig_bytes = 2; } // (tests: the widths that only reads > 64 kb, > 65535-read clusters, > 255-operation CIGARs ask for)
	if (n_events) *n_events = E;
	if (n_clusters) *n_clusters = 0;
	if (E == 0) { HIPCHECK(c, hipStreamSynchronize(c->st)); return SSV_OK; }
	ClipCounters *hc = P<ClipCounters>(c->h_counters);
	ClipCounters *dc = P<ClipCounters>(c->counters);
	const ClipEvent *ev = P<ClipEvent>(c->ev);
	CHECK(ensure(c, c->totals, 128)); CHECK(ensure_host(c, c->h_totals, 128));
	CHECK(ensure(c, c->qual_seen, 32)); CHECK(ensure(c, c->qual_lut, 256)); CHECK(ensure_host(c, c->h_qual_lut, 256));
	CHECK(ensure(c, c->pair_lut, 16384)); CHECK(ensure_host(c, c->h_pair_lut, 16384)); // (pairs: 4096 halves; triples: 4096 dwords)
	// ---- bin the events by (contig, side, position), BAM order inside a bin.  A coordinate-sorted BAM emits its '5' events in key order
	//      already (key = start + 1): that is checked, not assumed; only the '3' events (key = start + reference span) need the sort, and
	//      the two sorted lists interleave per contig.  Unsorted input takes the full sort. ----
	int cur = 0;
	{
		ProfScope ps(c, P_SORT, E);
		HIPCHECK(c, hipMemsetAsync(c->counters.p, 0, sizeof(ClipCounters), c->st));
		if (EL > 1) k_check_sorted<<<grid_for(EL, BLOCK), BLOCK, 0, c->st>>>(P<uint64_t>(c->key_l), EL, &dc->l_unsorted);
		if (EL > 0) k_key_max<<<(unsigned)std::min<int64_t>(512, grid_for(EL, BLOCK)), BLOCK, 0, c->st>>>(P<uint64_t>(c->key_l), EL, &dc->max_key);
		if (ER > 0) k_key_max<<<(unsigned)std::min<int64_t>(512, grid_for(ER, BLOCK)), BLOCK, 0, c->st>>>(P<uint64_t>(c->key_r[0]), ER, &dc->max_key);
		// the '3' list is sorted up to small displacements: one windowed rank pass, checked (tile_sort.h); SSV_RADIX_ONLY=1 skips the attempt
		if (ER > 0) { CHECK(ensure(c, c->key_r[1], ER * 8)); CHECK(ensure(c, c->val_r[1], ER * 4)); }
		if (ER > 0) HIPCHECK(c, sort_nearly_sorted(c->st, P<uint64_t>(c->key_r[0]), P<uint32_t>(c->val_r[0]), P<uint64_t>(c->key_r[1]), P<uint32_t>(c->val_r[1]), ER, &dc->r_unsorted));
		uint32_t *h_seen = reinterpret_cast<uint32_t *>(P<uint8_t>(c->h_totals) + 64);
		if (fmt3) {
			// first guess of the table's quality alphabet: the qualities of the first events
			HIPCHECK(c, hipMemsetAsync(c->qual_seen.p, 0, 32, c->st));
			const int64_t ns = std::min<int64_t>(E, 4096); // (a value missed here is caught while packing, at the price of packing twice)
			k_qual_sample<<<grid_for(ns, WAVES_PER_BLOCK), BLOCK, 0, c->st>>>(ev, P<uint32_t>(c->ev_idx), ns, P<uint32_t>(c->qual_seen));
			HIPCHECK(c, hipMemcpyAsync(h_seen, c->qual_seen.p, 32, hipMemcpyDeviceToHost, c->st));
		}
		HIPCHECK(c, hipMemcpyAsync(hc, c->counters.p, sizeof(ClipCounters), hipMemcpyDeviceToHost, c->st));
		HIPCHECK(c, hipStreamSynchronize(c->st));
		const uint64_t max_key = hc->max_key;
		int key_bits = 1;
		while (key_bits < 64 && (max_key >> key_bits)) ++key_bits;
		CHECK(ensure(c, c->keys2[0], E * 8)); CHECK(ensure(c, c->evs, (size_t)E * sizeof(ClipEvent)));
		if (!hc->l_unsorted) {
			int rcur = ER > 0 && !hc->r_unsorted ? 1 : 0;
			if (ER > 0 && rcur == 0) {
				const int64_t nt = rs_tiles(ER);
				CHECK(ensure(c, c->ghist, 256 * nt * 4));
				CHECK(ensure(c, c->scan_scratch, scan_scratch_elems(256 * nt) * 4));
				uint64_t *keys[2] = {P<uint64_t>(c->key_r[0]), P<uint64_t>(c->key_r[1])};
				uint32_t *vals[2] = {P<uint32_t>(c->val_r[0]), P<uint32_t>(c->val_r[1])};
				rcur = radix_sort_pairs(c->st, keys, vals, ER, key_bits, P<uint32_t>(c->ghist), P<uint32_t>(c->scan_scratch));
			}
			const int64_t Tn = (int64_t)(max_key >> 33) + 1;
			CHECK(ensure(c, c->cum_l, (size_t)(Tn + 2) * 4)); CHECK(ensure(c, c->cum_r, (size_t)(Tn + 2) * 4));
			k_side_bounds<<<grid_for(Tn + 1, BLOCK), BLOCK, 0, c->st>>>(P<uint64_t>(c->key_l), EL, P<uint64_t>(c->key_r[rcur]), ER, Tn, P<uint32_t>(c->cum_l), P<uint32_t>(c->cum_r));
			k_merge_sides<<<grid_for(E, BLOCK / 4), BLOCK, 0, c->st>>>(P<uint64_t>(c->key_l), P<uint32_t>(c->val_l), EL, P<uint64_t>(c->key_r[rcur]), P<uint32_t>(c->val_r[rcur]), ER,
			                                                       P<uint32_t>(c->cum_l), P<uint32_t>(c->cum_r), ev, P<uint64_t>(c->keys2[0]), P<ClipEvent>(c->evs));
			cur = 0;
		} else {
			CHECK(ensure(c, c->keys2[1], E * 8)); CHECK(ensure(c, c->vals2[0], E * 4)); CHECK(ensure(c, c->vals2[1], E * 4));
			const int64_t nt = rs_tiles(E);
			CHECK(ensure(c, c->ghist, 256 * nt * 4));
			CHECK(ensure(c, c->scan_scratch, scan_scratch_elems(256 * nt) * 4));
			k_concat_sides<<<grid_for(E, BLOCK), BLOCK, 0, c->st>>>(P<uint64_t>(c->key_l), P<uint32_t>(c->val_l), EL, P<uint64_t>(c->key_r[0]), P<uint32_t>(c->val_r[0]), ER,
			                                                        P<uint64_t>(c->keys2[0]), P<uint32_t>(c->vals2[0]));
			uint64_t *keys[2] = {P<uint64_t>(c->keys2[0]), P<uint64_t>(c->keys2[1])};
			uint32_t *vals[2] = {P<uint32_t>(c->vals2[0]), P<uint32_t>(c->vals2[1])};
			cur = radix_sort_pairs(c->st, keys, vals, E, key_bits, P<uint32_t>(c->ghist), P<uint32_t>(c->scan_scratch));
			k_gather_lines<<<grid_for(E, BLOCK / 4), BLOCK, 0, c->st>>>(vals[cur], E, ev, P<ClipEvent>(c->evs));
		}
		HIPCHECK(c, hipGetLastError());
	}
	// ---- greedy consensus clustering, one wavefront per multi-event bin ----
	ClusterArgs ca;
	ca.skey = P<uint64_t>(c->keys2[cur]); ca.E = E; ca.ev = P<ClipEvent>(c->evs);
	ca.match_rate = c->clip_p.match_rate;
	ca.SL = ca.SR = (std::max(1, c->max_lq) + 3) & ~3; // |seq_left|, |seq_right| <= l_qseq, also after consensus growth; a multiple of four: k_cluster_bins4 moves dwords
	const size_t stride = 2 * ((size_t)ca.SL + (size_t)ca.SR);
	int64_t M = 0;
	{
		ProfScope ps(c, P_CLUSTER_BINS, E);
		CHECK(ensure(c, c->c_support, E * 4)); CHECK(ensure(c, c->c_ll, E * 4)); CHECK(ensure(c, c->c_lr, E * 4)); CHECK(ensure(c, c->c_cig_ev, E * 4));
		CHECK(ensure(c, c->c_qmiss, E)); CHECK(ensure(c, c->c_mflag, E * 4)); CHECK(ensure(c, c->c_mslot, E * 4));
		CHECK(ensure(c, c->scan_scratch, scan_scratch_elems(E) * 4));
		ca.support = P<int32_t>(c->c_support); ca.c_ll = P<int32_t>(c->c_ll); ca.c_lr = P<int32_t>(c->c_lr); ca.c_cig_ev = P<uint32_t>(c->c_cig_ev);
		ca.c_qmiss = P<uint8_t>(c->c_qmiss); ca.mflag = P<uint32_t>(c->c_mflag); ca.mslot = P<uint32_t>(c->c_mslot);
		k_bin_mark<<<grid_for(E, BLOCK), BLOCK, 0, c->st>>>(ca.skey, E, P<uint32_t>(c->c_mflag), ca.support);
		exclusive_scan<uint32_t, uint32_t>(c->st, P<uint32_t>(c->c_mflag), P<uint32_t>(c->c_mslot), E, 0u, P<uint32_t>(c->scan_scratch), P<uint32_t>(c->totals));
		HIPCHECK(c, hipMemcpyAsync(c->h_totals.p, c->totals.p, 4, hipMemcpyDeviceToHost, c->st));
		HIPCHECK(c, hipStreamSynchronize(c->st));
		M = *P<uint32_t>(c->h_totals); // events in multi-event bins: only they need consensus storage
		CHECK(ensure(c, c->c_strings, (size_t)std::max<int64_t>(M, 1) * stride));
		ca.strings = P<uint8_t>(c->c_strings);
		ca.M = M; ca.mlist = nullptr; ca.blist = nullptr; ca.n_bins = nullptr; ca.dlist = nullptr; ca.tab4 = nullptr; ca.deep_cap = 0;
		if (M > 0) {
			// one wavefront per slot of a multi-event bin (3 % of the slots; the waves that do not sit on a bin start leave at once)
			CHECK(ensure(c, c->c_mlist, M * 4));
			ca.mlist = P<uint32_t>(c->c_mlist);
			k_multi_list<<<grid_for(E, BLOCK), BLOCK, 0, c->st>>>(ca.mflag, ca.mslot, E, P<uint32_t>(c->c_mlist));
			// the bins' first slots, densely: every wavefront of the clustering kernel then has a bin (a bin has at least two slots)
			// (two counters in one 64-bit scan: the bins, and the deep ones among them - totals[1] = bins | deep bins << 32)
			ca.deep_cap = M / B4_DEEP + 1;
			CHECK(ensure(c, c->c_bflag, M * 8)); CHECK(ensure(c, c->c_boff, M * 8)); CHECK(ensure(c, c->c_blist, (M / 2 + 1) * 4)); CHECK(ensure(c, c->c_dlist, (size_t)ca.deep_cap * 4));
			CHECK(ensure(c, c->scan_scratch, scan_scratch_elems(M) * 8));
			k_bin_start_flags<<<grid_for(M, BLOCK), BLOCK, 0, c->st>>>(ca.skey, E, ca.mlist, M, P<uint64_t>(c->c_bflag));
			exclusive_scan<uint64_t, uint64_t>(c->st, P<uint64_t>(c->c_bflag), P<uint64_t>(c->c_boff), M, 0ull, P<uint64_t>(c->scan_scratch), P<uint64_t>(c->totals) + 1);
			k_bin_start_list<<<grid_for(M, BLOCK), BLOCK, 0, c->st>>>(ca.mlist, P<uint64_t>(c->c_bflag), P<uint64_t>(c->c_boff), M, P<uint32_t>(c->c_blist), P<uint32_t>(c->c_dlist));
			ca.blist = P<uint32_t>(c->c_blist); ca.dlist = P<uint32_t>(c->c_dlist); ca.n_bins = P<uint32_t>(c->totals) + 2;
			// reads of up to 256 bases: four positions per lane, a wavefront = a workgroup, deep bins first; longer ones: a base per lane
			if (c->max_lq <= B4_CAP) {
				CHECK(ensure(c, c->bins4_tab, B4_TAB * 2));
				ca.tab4 = P<uint16_t>(c->bins4_tab);
				k_bins4_tables<<<1, BLOCK, 0, c->st>>>(ca.match_rate, P<uint16_t>(c->bins4_tab));
				k_cluster_bins4<<<(unsigned)(ca.deep_cap + M / 2 + 1), WAVE, 0, c->st>>>(ca);
			} else k_cluster_bins<<<grid_for(M / 2 + 1, WAVES_PER_BLOCK), BLOCK, 0, c->st>>>(ca);
		}
		HIPCHECK(c, hipGetLastError());
	}
	// ---- the dense table, cut straight out of the reads' bytes.  Nothing below needs a size on the host before the kernels have run:
	//      the buffers are sized by upper bounds (E clusters, E x the largest block, the events' CIGAR operations), the kernels read the
	//      cluster count from device memory, and the one synchronisation comes after the pack kernels. ----
	int64_t nc = 0;
	uint64_t str_total = 0, cig_total = 0;
	{
		ProfScope ps(c, P_CLUSTER_PACK, E);
		uint32_t *h_seen = reinterpret_cast<uint32_t *>(P<uint8_t>(c->h_totals) + 64);
		auto set_alphabet = [&](const uint32_t seen[8]) { // -> T.qual_bits / T.qual_alphabet / the phred -> index table (0xff: not in the alphabet)
			uint8_t *lut = P<uint8_t>(c->h_qual_lut);
			memset(lut, 0xff, 256); memset(T.qual_alphabet, 0, sizeof(T.qual_alphabet));
			int n_vals = 0;
			for (int v = 0; v < 256; ++v) if ((seen[v >> 5] >> (v & 31)) & 1u) ++n_vals;
			T.qual_group = 1; T.qual_radix = n_vals;
			const char *ge = getenv("SSV_QUAL_GROUPS");
			const bool groups = !ge || atoi(ge) != 0;
			if (n_vals > 45 || (n_vals > 16 && !(fmt3 && groups))) { T.qual_bits = 8; return; } // (bytes: no index table)
			T.qual_bits = n_vals <= 2 ? 1 : n_vals <= 4 ? 2 : n_vals <= 8 ? 3 : 4;
			// format 3: alphabets whose size is far from a power of two go in groups - five values: three qualities as one number below 5^3 in 7 bits (2.33
			// bits a quality instead of 3), nine to eleven values: two in 7 bits (3.5 instead of 4); table3_kernels.h.  SSV_QUAL_GROUPS=0: one quality, one field.
			if (fmt3 && groups && n_vals == 5) { T.qual_bits = 7; T.qual_group = 3; }
			if (fmt3 && groups && n_vals >= 9 && n_vals <= 11) { T.qual_bits = 7; T.qual_group = 2; }
			// 17 to 45 values (a HiSeq-style 40-value alphabet): two to a group of 11 bits (45 x 45 = 2025 <= 2048) - 5.5 bits a quality instead of a byte
			if (fmt3 && groups && n_vals >= 17 && n_vals <= 45) { T.qual_bits = 11; T.qual_group = 2; }
			int k = 0;
			for (int v = 0; v < 256; ++v) if ((seen[v >> 5] >> (v & 31)) & 1u) { T.qual_alphabet[k] = (uint8_t)(v + 33); lut[v] = (uint8_t)k; ++k; } // increasing order; the table shows characters (phred + 33)
		};
		uint32_t guess[8] = {0};
		if (fmt3) { memcpy(guess, h_seen, 32); set_alphabet(guess); }
		CHECK(ensure(c, c->slot_cnt, E * 8)); CHECK(ensure(c, c->slot_bytes, E * 8));
		CHECK(ensure(c, c->scan_scratch64, scan_scratch_elems(E) * 8));
		DBuf *d4[] = {&T.o_tid, &T.o_pos, &T.o_support, &T.o_ll, &T.o_lr, &T.o_ncig, &c->o_slowlist};
		for (DBuf *b : d4) CHECK(ensure(c, *b, E * 4 + 16));
		CHECK(ensure(c, T.o_side, E + 16)); CHECK(ensure(c, T.o_qmiss, E + 16));
		DBuf *d8[] = {&T.o_stroff, &T.o_cigoff};
		for (DBuf *b : d8) CHECK(ensure(c, *b, E * 8 + 16));
		CHECK(ensure(c, c->o_desc, (size_t)E * sizeof(PackDesc) + 64));
		CHECK(ensure(c, T.o_cig, (size_t)c->sum_ncig * 4 + 16));
		const int64_t exc_cap = exc_cap_of(E);
		if (fmt3) {
			CHECK(ensure(c, T.o_len, (size_t)E * 8 + 16)); CHECK(ensure(c, T.o_sup, (size_t)E * 4 + 16)); CHECK(ensure(c, T.o_nc, (size_t)E * 2 + 16));
			CHECK(ensure(c, T.o_runs, (size_t)E * sizeof(TableRun) + 16)); CHECK(ensure(c, T.o_exc, (size_t)exc_cap * 8 + 16));
		}
		// [0] clusters | CIGAR operations << 32, [1] string bytes, [2] slow-list length, [3] "a quality outside the alphabet" flag,
		// format 3: [4] runs | base exceptions << 32, [5] "a support count too wide" | "too many base exceptions" << 32
		uint64_t *tot = P<uint64_t>(c->totals);
		bool track = false;
		for (int attempt = 0;; ++attempt) {
			const size_t str_cap = (size_t)E * (size_t)(fmt3 ? table3_block_bytes((uint64_t)ca.SL + (uint64_t)ca.SR, T.base_bits, T.qual_bits, T.qual_group)
			                                             : table_block_bytes((uint64_t)ca.SL, (uint64_t)ca.SR, T.packed, (uint64_t)T.qual_bits));
			CHECK(ensure(c, T.o_str, str_cap + 16));
			PackArgs pa;
			pa.c = ca; pa.slot_cnt = P<uint64_t>(c->slot_cnt); pa.slot_bytes = P<uint64_t>(c->slot_bytes);
			pa.tid = P<int32_t>(T.o_tid); pa.pos = P<int32_t>(T.o_pos); pa.side = P<uint8_t>(T.o_side); pa.support = P<int32_t>(T.o_support); pa.ll = P<int32_t>(T.o_ll);
			pa.lr = P<int32_t>(T.o_lr); pa.qmiss = P<uint8_t>(T.o_qmiss); pa.ncig = P<int32_t>(T.o_ncig); pa.str_off = P<uint64_t>(T.o_stroff); pa.cig_off = P<uint64_t>(T.o_cigoff);
			pa.packed = T.packed; pa.qual_bits = T.qual_bits; pa.qual_group = T.qual_group; pa.qual_radix = T.qual_radix;
			pa.qual_fill = T.qual_bits != 8 && T.qual_alphabet[0] ? (uint32_t)(T.qual_alphabet[0] - 33) * 0x01010101u : 0u; pa.qlut = P<uint8_t>(c->qual_lut); pa.qual_seen = P<uint32_t>(c->qual_seen);
			pa.lut_miss = reinterpret_cast<int *>(tot + 3);
			pa.slow_list = P<uint32_t>(c->o_slowlist); pa.slow_count = reinterpret_cast<unsigned int *>(tot + 2);
			pa.format3 = fmt3 ? 1 : 0; pa.base_bits = T.base_bits;
			Pack3Args p3{};
			if (fmt3) {
				p3.pos = P<int32_t>(T.o_pos); p3.len = T.o_len.p; p3.support = T.o_sup.p; p3.ncig = T.o_nc.p; p3.flags = P<uint8_t>(T.o_qmiss);
				p3.len_bytes = T.len_bytes; p3.support_bytes = T.support_bytes; p3.ncig_bytes = T.ncig_bytes; p3.base_bits = T.base_bits;
				p3.runs = P<TableRun>(T.o_runs); p3.run_count = reinterpret_cast<unsigned int *>(tot + 4);
				p3.exc = P<uint64_t>(T.o_exc); p3.exc_count = reinterpret_cast<unsigned int *>(tot + 4) + 1; p3.exc_cap = (uint32_t)exc_cap;
				p3.support_miss = reinterpret_cast<int *>(tot + 5); p3.exc_miss = reinterpret_cast<int *>(tot + 5) + 1;
				p3.cig_bytes = T.cig_bytes; p3.cig_miss = reinterpret_cast<int *>(tot + 6);
			}
			HIPCHECK(c, hipMemsetAsync(tot, 0, 64, c->st));
			if (track) HIPCHECK(c, hipMemsetAsync(c->qual_seen.p, 0, 32, c->st));
			if (T.packed && T.qual_bits != 8) HIPCHECK(c, hipMemcpyAsync(c->qual_lut.p, c->h_qual_lut.p, 256, hipMemcpyHostToDevice, c->st));
			// format 3, the kernel without the LDS stage: two qualities per table look-up - for alphabets below phred 64 (every sequencer's)
			bool direct = fmt3 && !track;
			if (direct && T.qual_bits != 8) {
				const uint8_t *lut = P<uint8_t>(c->h_qual_lut);
				for (int v = 64; v < 256; ++v) if (lut[v] != 0xff) direct = false;
				pa.tri_mul = 0;
				if (direct && T.qual_group == 3) {
					// three to a group: ONE look-up per group (qual_dword3h) - the alphabet's R^3 triples of phred bytes hashed into 4096 slots by a multiplier under which
					// no two of them meet (a few tries: 125 keys, 4096 slots); an entry = triple << 8 | the group's number, an empty slot matches no triple
					uint32_t *tl = P<uint32_t>(c->h_pair_lut);
					const int R = T.qual_radix;
					uint32_t mul = 0;
					for (uint32_t m = 0x9e3779u; m < 0x9e3779u + 4096u * 2u && !mul; m += 2u) {
						for (int i = 0; i < (1 << TRI_BITS); ++i) tl[i] = 0xffffffffu;
						bool ok = true;
						for (int i2 = 0; i2 < R && ok; ++i2)
							for (int i1 = 0; i1 < R && ok; ++i1)
								for (int i0 = 0; i0 < R && ok; ++i0) {
									const uint32_t tri = (uint32_t)(T.qual_alphabet[i0] - 33) | ((uint32_t)(T.qual_alphabet[i1] - 33) << 8) | ((uint32_t)(T.qual_alphabet[i2] - 33) << 16);
									const uint32_t slot = (uint32_t)((uint64_t)tri * (m & 0xffffffu)) >> (32 - TRI_BITS); // (v_mul_u32_u24: the low 32 bits of the 24 x 24 bit product)
									if (tl[slot] != 0xffffffffu) ok = false;
									else tl[slot] = (tri << 8) | (uint32_t)(i0 + R * i1 + R * R * i2);
								}
						if (ok) mul = m & 0xffffffu;
					}
					if (mul) { pa.tri_mul = mul; HIPCHECK(c, hipMemcpyAsync(c->pair_lut.p, c->h_pair_lut.p, 16384, hipMemcpyHostToDevice, c->st)); }
					else direct = false; // (never seen; the staged kernel takes the shape at run time)
				} else if (direct) {
					uint16_t *pl = P<uint16_t>(c->h_pair_lut);
					for (int q1 = 0; q1 < 64; ++q1)
						for (int q0 = 0; q0 < 64; ++q0) {
							const bool out = lut[q0] == 0xff || lut[q1] == 0xff;
							if (T.qual_group == 2) pl[q0 | (q1 << 6)] = out ? (uint16_t)0x8000 : (uint16_t)(lut[q0] + T.qual_radix * lut[q1]); // qual_dword3g: a pair IS a group
							else if (T.qual_group > 1) pl[q0 | (q1 << 6)] = out ? (uint16_t)0x8000 : (uint16_t)(lut[q0] | (lut[q1] << 4) | ((lut[q0] + T.qual_radix * lut[q1]) << 8));
							else pl[q0 | (q1 << 6)] = out ? (uint16_t)0x8000 : (uint16_t)(lut[q0] | (lut[q1] << T.qual_bits));
						}
					HIPCHECK(c, hipMemcpyAsync(c->pair_lut.p, c->h_pair_lut.p, 8192, hipMemcpyHostToDevice, c->st));
				}
			}
			uint8_t *os = P<uint8_t>(T.o_str);
			PackDesc *dsc = P<PackDesc>(c->o_desc);
			// format 3: the scans behind the rows' columns over tiles of 256 slots (k_cluster_tile_sums, k_cluster_cols3_tiles); SSV_PACK_COLS=split: two words per slot,
			// two device-wide scans, then the columns, as before round 6
			static const bool split_cols = [] { const char *e = getenv("SSV_PACK_COLS"); return e && !strcmp(e, "split"); }();
			if (fmt3 && !split_cols) {
				const unsigned tiles = grid_for(E, BLOCK);
				const int64_t stride = ((int64_t)tiles + 63) & ~63ll;
				CHECK(ensure(c, c->tile_sums, (size_t)stride * 3 * 8));
				TileSums ts;
				ts.clusters = P<uint64_t>(c->tile_sums); ts.cig = ts.clusters + stride; ts.bytes = ts.cig + stride;
				k_cluster_tile_sums<<<tiles, BLOCK, 0, c->st>>>(pa, ts);
				k_scan_sums_lists<uint64_t><<<3, BLOCK, 0, c->st>>>(ts.clusters, (int64_t)tiles, stride);
				k_cluster_cols3_tiles<<<tiles, BLOCK, 0, c->st>>>(pa, p3, dsc, P<uint32_t>(T.o_cig), ts, tot);
			} else {
				k_cluster_meta<<<grid_for(E, BLOCK), BLOCK, 0, c->st>>>(pa);
				exclusive_scan<uint64_t, uint64_t>(c->st, pa.slot_cnt, pa.slot_cnt, E, 0ull, P<uint64_t>(c->scan_scratch64), tot);
				exclusive_scan<uint64_t, uint64_t>(c->st, pa.slot_bytes, pa.slot_bytes, E, 0ull, P<uint64_t>(c->scan_scratch64), tot + 1);
				if (fmt3) k_cluster_cols3<<<grid_for(E, BLOCK), BLOCK, 0, c->st>>>(pa, p3, dsc, P<uint32_t>(T.o_cig));
				else k_cluster_cols<<<grid_for(E, BLOCK), BLOCK, 0, c->st>>>(pa, dsc, P<uint32_t>(T.o_cig));
			}
			const unsigned int *nc_dev = reinterpret_cast<const unsigned int *>(tot);
			if (fmt3) {
				// one group of lanes per cluster (the grid is an upper bound, the kernel reads the cluster count itself), then the base-by-base path
				const dim3 g(grid_for(E, GROUPS_PER_BLOCK));
				const dim3 gs3(grid_for(std::max<int64_t>(M + c->n_long, 1), BLOCK)); // k_pack3_slow: an item per lane (a wavefront then works off the ones with a cluster)
				const unsigned p3_blocks = 256u * 40u; // persistent (80 registers: six workgroups per CU resident); 2560 / 5120 / 10240 / 20480 workgroups measured in round 4: 1.20 / 1.18 / 1.14 / 1.16 ms for the group
				const dim3 gd((unsigned)std::max<int64_t>(1, std::min<int64_t>(p3_blocks, (E + GROUPS_PER_BLOCK - 1) / GROUPS_PER_BLOCK)));
				// lanes per cluster of the direct kernel: as many as the longest read's base / quality stream has dwords, rounded up to the next whole share of a wavefront
				// (150 bases, grouped qualities: 11 -> 12 lanes, five clusters a wavefront); streams of more than 32 dwords take 16 lanes and several rounds
				int lpc = 16;
				{
					const int n_fast = std::max(1, std::min(c->max_lq, PACK_MAX_LQ));
					const int nd = std::max((n_fast * T.base_bits + 31) / 32, (int)((qual_stream_bits((uint64_t)n_fast, (uint64_t)T.qual_bits, (uint64_t)T.qual_group) + 31) / 32));
					if (nd <= 32) lpc = WAVE / (WAVE / nd);
				}
#define SSV_P3D(W_, B_, K_) k_pack3_direct<W_, B_, K_><<<gd, BLOCK, 0, c->st>>>(pa, p3, dsc, nc_dev, os, P<uint16_t>(c->pair_lut), lpc)
#define SSV_P3B(W_, B_, T_) do { if (direct) SSV_P3D(W_, B_, 1); else k_pack3_stream<W_, B_, T_><<<g, BLOCK, 0, c->st>>>(pa, p3, dsc, nc_dev, os); \
			k_pack3_slow<W_, B_, T_><<<gs3, BLOCK, 0, c->st>>>(pa, p3, os); } while (0)
				// grouped qualities: the direct kernel knows the two shapes, the staged and the bytewise kernels take the shape at run time (W = 0)
#define SSV_P3G(B_, T_) do { if (direct) { if (pa.qual_group == 3) SSV_P3D(7, B_, 3); else if (pa.qual_bits == 11) SSV_P3D(11, B_, 2); else SSV_P3D(7, B_, 2); } \
			else k_pack3_stream<0, B_, T_><<<g, BLOCK, 0, c->st>>>(pa, p3, dsc, nc_dev, os); \
			k_pack3_slow<0, B_, T_><<<gs3, BLOCK, 0, c->st>>>(pa, p3, os); } while (0)
#define SSV_P3GT(T_) do { if (T.base_bits == 2) SSV_P3G(2, T_); else SSV_P3G(4, T_); } while (0)
#define SSV_P3T(W_, T_) do { if (T.base_bits == 2) SSV_P3B(W_, 2, T_); else SSV_P3B(W_, 4, T_); } while (0)
#define SSV_P3(W_) do { if (track) SSV_P3T(W_, true); else SSV_P3T(W_, false); } while (0)
				if (pa.qual_group > 1) { if (track) SSV_P3GT(true); else SSV_P3GT(false); }
				else if (pa.qual_bits == 8) SSV_P3T(8, false); else if (pa.qual_bits == 4) SSV_P3(4); else if (pa.qual_bits == 3) SSV_P3(3); else if (pa.qual_bits == 2) SSV_P3(2); else SSV_P3(1);
#undef SSV_P3
#undef SSV_P3T
#undef SSV_P3B
#undef SSV_P3D
#undef SSV_P3G
#undef SSV_P3GT
			} else k_cluster_pack_ascii<<<grid_for(E, GROUPS_PER_BLOCK), BLOCK, 0, c->st>>>(pa, os);
			HIPCHECK(c, hipGetLastError());
			HIPCHECK(c, hipMemcpyAsync(c->h_totals.p, c->totals.p, 64, hipMemcpyDeviceToHost, c->st));
			if (track) HIPCHECK(c, hipMemcpyAsync(h_seen, c->qual_seen.p, 32, hipMemcpyDeviceToHost, c->st));
			HIPCHECK(c, hipStreamSynchronize(c->st));
			if (fmt3 && attempt <= 6) {
				// (rare) a cluster with more than 65535 reads: the support column as u32; more bases outside A/C/G/T than the exception list takes:
				// the base streams at 4 bits
				const uint64_t m = P<uint64_t>(c->h_totals)[5];
				if ((uint32_t)m && T.support_bytes == 2) { T.support_bytes = 4; continue; }
				if ((uint32_t)(m >> 32) && T.base_bits == 2) { T.base_bits = 4; continue; }
				if ((uint32_t)P<uint64_t>(c->h_totals)[6] && T.cig_bytes == 2) { T.cig_bytes = 4; continue; } // an operation of 4096 bases or more (a long N / D)
			}
			if (track) { // the launch above met every quality value of the table's strings: that is the alphabet; pack once more with it
				memcpy(guess, h_seen, 32);
				set_alphabet(guess);
				track = false;
				continue;
			}
			if (fmt3 && T.qual_bits != 8 && (int)P<uint64_t>(c->h_totals)[3] != 0) {
				// the table's strings hold a quality value that the first events did not show: find out which values there are
				if (attempt > 8) { c->err = "quality alphabet did not settle"; return SSV_E_HIP; }
				track = true;
				continue;
			}
			break;
		}
		nc = (int64_t)(uint32_t)P<uint64_t>(c->h_totals)[0];
		cig_total = P<uint64_t>(c->h_totals)[0] >> 32;
		str_total = P<uint64_t>(c->h_totals)[1];
		if (fmt3) { T.n_runs = (int64_t)(uint32_t)P<uint64_t>(c->h_totals)[4]; T.n_exc = (int64_t)(P<uint64_t>(c->h_totals)[4] >> 32); }
	}
	T.n_clusters = nc; T.str_bytes = str_total; T.cig_ops = cig_total;
	if (n_clusters) *n_clusters = nc;
	if (nc == 0) return SSV_OK;
	struct CopyItem { HBuf *h; DBuf *d; size_t bytes; };
	std::vector<CopyItem> cp;
	if (fmt3)
		cp = {{&T.h_pos, &T.o_pos, (size_t)nc * 4}, {&T.h_len, &T.o_len, (size_t)nc * 2 * (size_t)T.len_bytes}, {&T.h_sup, &T.o_sup, (size_t)nc * (size_t)T.support_bytes},
		      {&T.h_nc, &T.o_nc, (size_t)nc * (size_t)T.ncig_bytes}, {&T.h_qmiss, &T.o_qmiss, (size_t)nc}, {&T.h_str, &T.o_str, (size_t)str_total}, {&T.h_cig, &T.o_cig, (size_t)cig_total * (size_t)T.cig_bytes},
		      {&T.h_runs, &T.o_runs, (size_t)T.n_runs * sizeof(TableRun)}, {&T.h_exc, &T.o_exc, (size_t)T.n_exc * 8}};
	else
		cp = {{&T.h_tid, &T.o_tid, (size_t)nc * 4}, {&T.h_pos, &T.o_pos, (size_t)nc * 4}, {&T.h_side, &T.o_side, (size_t)nc}, {&T.h_support, &T.o_support, (size_t)nc * 4},
		      {&T.h_ll, &T.o_ll, (size_t)nc * 4}, {&T.h_lr, &T.o_lr, (size_t)nc * 4}, {&T.h_qmiss, &T.o_qmiss, (size_t)nc}, {&T.h_stroff, &T.o_stroff, (size_t)nc * 8},
		      {&T.h_cigoff, &T.o_cigoff, (size_t)nc * 8}, {&T.h_ncig, &T.o_ncig, (size_t)nc * 4}, {&T.h_str, &T.o_str, (size_t)str_total}, {&T.h_cig, &T.o_cig, (size_t)cig_total * 4}};
	// the table goes to pinned host memory on the copy stream, behind the pack kernels; ssv_clip_table_wait() waits for it
	for (auto &x : cp) CHECK(ensure_host(c, *x.h, x.bytes + 16));
	T.via_link = false;
	if (c->link.ok) {
		// (the pack kernels are done: the loop above left through a synchronisation of the stream.)  Every piece on the engine of the direction; the signal counts them down.
		int64_t pieces = 0;
		const CopyItem *big = nullptr;
		for (auto &x : cp) if (x.bytes) { ++pieces; if (!big || x.bytes > big->bytes) big = &x; }
		T.big_bytes = big ? big->bytes : 0;
		c->link.signal_store(T.copied_sig, pieces - (big ? 1 : 0));
		c->link.signal_store(T.big_sig, big ? 1 : 0);
		int64_t started = 0;
		for (auto &x : cp) if (x.bytes) { if (!link_copy(c, x.h->p, x.d->p, x.bytes, true, &x == big ? T.big_sig : T.copied_sig)) break; ++started; }
		if (started == pieces) T.via_link = true;
		else { // an engine that refuses: let what was started land, then copy everything the runtime's way
			bool big_started = false;
			{ int64_t k = 0; for (auto &x : cp) if (x.bytes) { if (k < started && &x == big) big_started = true; ++k; } }
			c->link.signal_store(T.copied_sig, c->link.signal_load(T.copied_sig) - ((pieces - started) - (big && !big_started ? 1 : 0)));
			if (big && !big_started) c->link.signal_store(T.big_sig, 0);
			link_wait(c, T.copied_sig); link_wait(c, T.big_sig);
			c->link.ok = false;
		}
	}
	if (!T.via_link) {
		HIPCHECK(c, hipEventRecord(T.packed_ev, c->st));
		HIPCHECK(c, hipStreamWaitEvent(c->st_copy, T.packed_ev, 0));
		for (auto &x : cp) if (x.bytes) HIPCHECK(c, hipMemcpyAsync(x.h->p, x.d->p, x.bytes, hipMemcpyDeviceToHost, c->st_copy));
		HIPCHECK(c, hipEventRecord(T.copied, c->st_copy));
	}
	T.in_flight = true;
	// size the other table set like this one now (pinning ~2 GB of host memory takes ~100 ms: better here than in the caller's next pass)
	ssv_ctx::TableSet &O = c->tab[s_ ^ 1];
	if (!O.in_flight) {
		struct { HBuf *h; DBuf *d; size_t bytes, dbytes; } oc[] = {
			{&O.h_tid, &O.o_tid, (size_t)nc * 4, (size_t)E * 4}, {&O.h_pos, &O.o_pos, (size_t)nc * 4, (size_t)E * 4}, {&O.h_side, &O.o_side, (size_t)nc, (size_t)E}, {&O.h_support, &O.o_support, (size_t)nc * 4, (size_t)E * 4},
			{&O.h_ll, &O.o_ll, (size_t)nc * 4, (size_t)E * 4}, {&O.h_lr, &O.o_lr, (size_t)nc * 4, (size_t)E * 4}, {&O.h_qmiss, &O.o_qmiss, (size_t)nc, (size_t)E}, {&O.h_stroff, &O.o_stroff, (size_t)nc * 8, (size_t)E * 8},
			{&O.h_cigoff, &O.o_cigoff, (size_t)nc * 8, (size_t)E * 8}, {&O.h_ncig, &O.o_ncig, (size_t)nc * 4, (size_t)E * 4}, {&O.h_str, &O.o_str, (size_t)str_total, T.o_str.cap - 16},
			{&O.h_cig, &O.o_cig, (size_t)cig_total * 4, (size_t)c->sum_ncig * 4}};
		struct { HBuf *h; DBuf *d; size_t bytes, dbytes; } oc3[] = {
			{&O.h_pos, &O.o_pos, (size_t)nc * 4, (size_t)E * 4}, {&O.h_len, &O.o_len, (size_t)nc * 2 * (size_t)T.len_bytes, (size_t)E * 8}, {&O.h_sup, &O.o_sup, (size_t)nc * (size_t)T.support_bytes, (size_t)E * 4},
			{&O.h_nc, &O.o_nc, (size_t)nc * (size_t)T.ncig_bytes, (size_t)E * 2}, {&O.h_qmiss, &O.o_qmiss, (size_t)nc, (size_t)E}, {&O.h_str, &O.o_str, (size_t)str_total, T.o_str.cap - 16},
			{&O.h_cig, &O.o_cig, (size_t)cig_total * 4, (size_t)c->sum_ncig * 4}, {&O.h_runs, &O.o_runs, (size_t)T.n_runs * sizeof(TableRun), (size_t)E * sizeof(TableRun)},
			{&O.h_exc, &O.o_exc, (size_t)T.n_exc * 8, (size_t)exc_cap_of(E) * 8}};
		if (fmt3) for (auto &x : oc3) {
			if (x.h->cap < x.bytes + 16) CHECK(ensure_host(c, *x.h, x.bytes + 16));
			if (x.d->cap < x.dbytes + 16) { void *np = nullptr; HIPCHECK(c, dev_malloc(c, &np, x.dbytes + 16)); if (x.d->p) HIPCHECK(c, hipFree(x.d->p)); x.d->p = np; x.d->cap = x.dbytes + 16; }
		}
		else for (auto &x : oc) {
			if (x.h->cap < x.bytes + 16) CHECK(ensure_host(c, *x.h, x.bytes + 16));
			if (x.d->cap < x.dbytes + 16) { void *np = nullptr; HIPCHECK(c, dev_malloc(c, &np, x.dbytes + 16)); if (x.d->p) HIPCHECK(c, hipFree(x.d->p)); x.d->p = np; x.d->cap = x.dbytes + 16; }
		}
	}
	return SSV_OK;
}

static int table_wait(ssv_ctx *c, int which, ssv_cluster_table *out);
static void table_expanded_view(ssv_ctx::TableSet &T, ssv_cluster_table *out);

uint64_t ssv_table_block_bytes(int32_t left_len, int32_t right_len) { return table_block_bytes((uint64_t)left_len, (uint64_t)right_len, 0, 8); }

int ssv_clip_table_format(ssv_ctx *c, int packed)
{
	if (!c) return SSV_E_ARG;
	if (packed != 0 && packed != 3) { c->err = "ssv_clip_table_format: 0 (ASCII) or 3 (compact); the four-piece packed formats 1 and 2 of ABI versions < 8 are gone"; return SSV_E_ARG; }
	c->table_mode = packed;
	return SSV_OK;
}

int ssv_clip_table_wait(ssv_ctx *c, ssv_cluster_table *out) { return c && out ? table_wait(c, c->tab_cur, out) : SSV_E_ARG; }
int ssv_clip_table_wait_prev(ssv_ctx *c, ssv_cluster_table *out) { return c && out ? table_wait(c, c->tab_cur ^ 1, out) : SSV_E_ARG; }

static int table_wait(ssv_ctx *c, int which, ssv_cluster_table *out)
{
	HIPCHECK(c, hipSetDevice(c->device));
	ssv_ctx::TableSet &T = c->tab[which];
	memset(out, 0, sizeof(*out));
	if (T.in_flight) {
		ProfScope pd(c, P_TABLE_D2H, T.n_clusters); // what is left of the copy when the caller asks for the table
		if (T.via_link) table_link_wait(c, T); else HIPCHECK(c, hipEventSynchronize(T.copied));
		T.in_flight = false;
	}
	out->n_events = T.n_events; out->n_clusters = T.n_clusters; out->seq_packed = T.packed; out->qual_bits = T.qual_bits; out->qual_group = T.qual_group; memcpy(out->qual_alphabet, T.qual_alphabet, sizeof(out->qual_alphabet));
	out->format = T.format; out->base_bits = T.base_bits;
	if (T.n_clusters == 0) return SSV_OK;
	if (T.format == 3) {
		out->len_bytes = T.len_bytes; out->support_bytes = T.support_bytes; out->ncig_bytes = T.ncig_bytes;
		out->pos = P<int32_t>(T.h_pos); out->c_len = T.h_len.p; out->c_support = T.h_sup.p; out->c_ncig = T.h_nc.p; out->c_flags = P<uint8_t>(T.h_qmiss);
		out->str = P<uint8_t>(T.h_str); out->cigar = T.cig_bytes == 4 ? P<uint32_t>(T.h_cig) : nullptr; out->c_cigar = T.h_cig.p; out->cigar_bytes = T.cig_bytes; out->str_bytes = T.str_bytes; out->cigar_ops = T.cig_ops;
		out->runs = reinterpret_cast<const ssv_table_run *>(T.h_runs.p); out->n_runs = T.n_runs;
		out->base_exc = P<uint64_t>(T.h_exc); out->n_base_exc = T.n_exc;
		if (!T.ordered) { // once per table: put the runs (appended by whichever thread came first) and the exceptions in order
			ssv_table_run *r = reinterpret_cast<ssv_table_run *>(T.h_runs.p);
			std::sort(r, r + T.n_runs, [](const ssv_table_run &a, const ssv_table_run &b) { return a.first < b.first; });
			uint64_t *e = P<uint64_t>(T.h_exc);
			std::sort(e, e + T.n_exc);
			T.ordered = true;
		}
		if (T.expanded) { table_expanded_view(T, out); out->support_sum = T.support_sum; }
		return SSV_OK;
	}
	out->str_bytes = T.str_bytes; out->cigar_ops = T.cig_ops;
	out->tid = P<int32_t>(T.h_tid); out->pos = P<int32_t>(T.h_pos); out->side = P<uint8_t>(T.h_side); out->support = P<int32_t>(T.h_support);
	out->left_len = P<int32_t>(T.h_ll); out->right_len = P<int32_t>(T.h_lr); out->qual_missing = P<uint8_t>(T.h_qmiss); out->str_off = P<uint64_t>(T.h_stroff);
	out->str = P<uint8_t>(T.h_str); out->cigar_off = P<uint64_t>(T.h_cigoff); out->n_cigar = P<int32_t>(T.h_ncig); out->cigar = P<uint32_t>(T.h_cig);
	out->c_cigar = T.h_cig.p; out->cigar_bytes = 4;
	return SSV_OK;
}

// ---- the columns a compact table leaves to the host ----

static void table_expanded_view(ssv_ctx::TableSet &T, ssv_cluster_table *out)
{
	out->tid = T.x_tid.data(); out->side = T.x_side.data(); out->support = T.x_support.data(); out->left_len = T.x_ll.data(); out->right_len = T.x_lr.data();
	out->qual_missing = T.x_qmiss.data(); out->n_cigar = T.x_ncig.data(); out->str_off = T.x_stroff.data(); out->cigar_off = T.x_cigoff.data();
}

} // extern "C"

// one range of clusters: the widened columns (pass 1, also the range's string bytes / CIGAR operations / support sum), then the offsets (pass 2)
template <class LenT, class SupT, class NcT>
static void expand_range(ssv_ctx::TableSet &T, int64_t k0, int64_t k1, bool second, uint64_t &so, uint64_t &co, int64_t &ssum)
{
	const LenT *len = reinterpret_cast<const LenT *>(T.h_len.p);
	const SupT *sup = reinterpret_cast<const SupT *>(T.h_sup.p);
	const NcT *ncg = reinterpret_cast<const NcT *>(T.h_nc.p);
	const uint8_t *fl = P<uint8_t>(T.h_qmiss);
	const uint64_t bb = (uint64_t)T.base_bits, qb = (uint64_t)T.qual_bits, qg = (uint64_t)T.qual_group;
	int32_t *x_ll = T.x_ll.data(), *x_lr = T.x_lr.data(), *x_sup = T.x_support.data(), *x_nc = T.x_ncig.data();
	uint8_t *x_qm = T.x_qmiss.data();
	uint64_t *x_so = T.x_stroff.data(), *x_co = T.x_cigoff.data();
	if (!second) {
		int64_t sum = 0;
		for (int64_t k = k0; k < k1; ++k) {
			const uint32_t ll = len[2 * k], lr = len[2 * k + 1], nc1 = ncg[k], s1 = sup[k];
			x_ll[k] = (int32_t)ll; x_lr[k] = (int32_t)lr; x_sup[k] = (int32_t)s1; x_nc[k] = (int32_t)nc1; x_qm[k] = fl[k] & 1;
			const uint64_t n = (uint64_t)ll + lr;
			so += 4ull * ((n * bb + 31) / 32 + (qual_stream_bits(n, qb, qg) + 31) / 32); co += nc1; sum += s1;
		}
		ssum = sum;
		return;
	}
	for (int64_t k = k0; k < k1; ++k) {
		x_so[k] = so; x_co[k] = co;
		const uint64_t n = (uint64_t)(uint32_t)x_ll[k] + (uint32_t)x_lr[k];
		so += 4ull * ((n * bb + 31) / 32 + (qual_stream_bits(n, qb, qg) + 31) / 32); co += (uint32_t)x_nc[k];
	}
}

extern "C" {

int ssv_clip_table_expand(ssv_ctx *c, ssv_cluster_table *t, int32_t n_threads)
{
	if (!c || !t) return SSV_E_ARG;
	ssv_ctx::TableSet *Tp = nullptr;
	for (auto &x : c->tab) if (x.format == 3 && !x.in_flight && x.n_clusters == t->n_clusters && (t->n_clusters == 0 || t->pos == P<int32_t>(x.h_pos))) Tp = &x;
	if (t->format != 3 || !Tp) { c->err = "ssv_clip_table_expand takes a compact (format 3) table handed out by ssv_clip_table_wait"; return SSV_E_ARG; }
	ssv_ctx::TableSet &T = *Tp;
	const int64_t n = T.n_clusters;
	if (n == 0 || T.expanded) { if (n) table_expanded_view(T, t); t->support_sum = T.support_sum; return SSV_OK; }
	T.x_tid.resize((size_t)n); T.x_side.resize((size_t)n); T.x_support.resize((size_t)n); T.x_ll.resize((size_t)n); T.x_lr.resize((size_t)n); T.x_qmiss.resize((size_t)n);
	T.x_ncig.resize((size_t)n); T.x_stroff.resize((size_t)n); T.x_cigoff.resize((size_t)n);
	const int nt = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)(n_threads > 0 ? n_threads : (int32_t)effective_cpus()), 64, n / 32768 + 1}));
	const ssv_table_run *runs = reinterpret_cast<const ssv_table_run *>(T.h_runs.p);
	const int64_t n_runs = T.n_runs;
	std::vector<uint64_t> part_str((size_t)nt + 1, 0), part_cig((size_t)nt + 1, 0);
	std::vector<int64_t> part_sup((size_t)nt, 0);
	// pass 1 (widened columns, per-range sums, contig / side), the ranges' starting offsets, pass 2 (offsets)
	auto pass = [&](int w, int second) {
		const int64_t k0 = n * w / nt, k1 = n * (w + 1) / nt;
		uint64_t so = second ? part_str[(size_t)w] : 0, co = second ? part_cig[(size_t)w] : 0;
		int64_t ssum = 0;
		if (T.len_bytes == 2) {
			if (T.support_bytes == 2) { if (T.ncig_bytes == 1) expand_range<uint16_t, uint16_t, uint8_t>(T, k0, k1, second, so, co, ssum); else expand_range<uint16_t, uint16_t, uint16_t>(T, k0, k1, second, so, co, ssum); }
			else { if (T.ncig_bytes == 1) expand_range<uint16_t, uint32_t, uint8_t>(T, k0, k1, second, so, co, ssum); else expand_range<uint16_t, uint32_t, uint16_t>(T, k0, k1, second, so, co, ssum); }
		} else {
			if (T.support_bytes == 2) { if (T.ncig_bytes == 1) expand_range<uint32_t, uint16_t, uint8_t>(T, k0, k1, second, so, co, ssum); else expand_range<uint32_t, uint16_t, uint16_t>(T, k0, k1, second, so, co, ssum); }
			else { if (T.ncig_bytes == 1) expand_range<uint32_t, uint32_t, uint8_t>(T, k0, k1, second, so, co, ssum); else expand_range<uint32_t, uint32_t, uint16_t>(T, k0, k1, second, so, co, ssum); }
		}
		if (second) return;
		part_str[(size_t)w + 1] = so; part_cig[(size_t)w + 1] = co; part_sup[(size_t)w] = ssum;
		// contig / side of the range: whole runs at a time
		int64_t lo = 0, hi = n_runs;
		while (hi - lo > 1) { const int64_t m = (lo + hi) / 2; if (runs[m].first <= k0) lo = m; else hi = m; }
		for (int64_t r = lo; r < n_runs && runs[r].first < k1; ++r) {
			const int64_t a = std::max(k0, runs[r].first), b = std::min(k1, r + 1 < n_runs ? runs[r + 1].first : n);
			if (b > a) { std::fill(T.x_tid.begin() + a, T.x_tid.begin() + b, runs[r].tid); std::fill(T.x_side.begin() + a, T.x_side.begin() + b, runs[r].side); }
		}
	};
	c->pool.run(nt, [&](int w) { pass(w, 0); });
	for (int v = 0; v < nt; ++v) { part_str[(size_t)v + 1] += part_str[(size_t)v]; part_cig[(size_t)v + 1] += part_cig[(size_t)v]; }
	c->pool.run(nt, [&](int w) { pass(w, 1); });
	if (part_str[(size_t)nt] != T.str_bytes || part_cig[(size_t)nt] != T.cig_ops) { c->err = "compact table: the rebuilt offsets do not add up to the blob sizes"; return SSV_E_HIP; }
	T.support_sum = 0;
	for (int64_t v : part_sup) T.support_sum += v;
	T.expanded = true;
	table_expanded_view(T, t);
	t->support_sum = T.support_sum;
	return SSV_OK;
}

uint64_t ssv_table_block_bytes3(int64_t n_bases, int32_t base_bits, int32_t qual_bits) { return table3_block_bytes((uint64_t)n_bases, base_bits, qual_bits); }
uint64_t ssv_table_block_bytes3g(int64_t n_bases, int32_t base_bits, int32_t qual_bits, int32_t qual_group) { return table3_block_bytes((uint64_t)n_bases, base_bits, qual_bits, qual_group > 1 ? qual_group : 1); }

int ssv_clip_cluster(ssv_ctx *c, ssv_cluster_table *out)
{
	if (!c || !out) return SSV_E_ARG;
	CHECK(ssv_clip_cluster_async(c, nullptr, nullptr));
	return ssv_clip_table_wait(c, out);
}

// ---------------------------------------------------------------------------------------------------------------------
// insert-size statistics
// ---------------------------------------------------------------------------------------------------------------------

int ssv_isize_begin(ssv_ctx *c, int32_t min_mapq, int64_t max_pairs)
{
	if (!c) return SSV_E_ARG;
	HIPCHECK(c, hipSetDevice(c->device));
	c->pf.clear();
	c->isz_active = true; c->isz_min_mapq = min_mapq; c->isz_max = max_pairs; c->isz_count = 0;
	return SSV_OK;
}

int ssv_isize_accumulate(ssv_ctx *c, const ssv_batch_t *b, int32_t *done)
{
	if (!c || !b) return SSV_E_ARG;
	if (!c->isz_active) { c->err = "ssv_isize_accumulate before ssv_isize_begin"; return SSV_E_STATE; }
	HIPCHECK(c, hipSetDevice(c->device));
	// the reference tests `read_pair_number == read_pair_used` after every record (cluster.cpp:68): with max_pairs == 0 it stops at once
	if (c->isz_count >= c->isz_max || b->n == 0) { if (done) *done = c->isz_count >= c->isz_max; return SSV_OK; }
	DevBatch d;
	CHECK(stage_batch(c, b, d));
	ProfScope ps(c, P_ISIZE, d.n);
	const int64_t ntiles = (d.n + ISZ_TILE - 1) / ISZ_TILE;
	CHECK(ensure(c, c->tile_cnt, ntiles * 4)); CHECK(ensure(c, c->tile_base, ntiles * 4));
	CHECK(ensure(c, c->scan_scratch, scan_scratch_elems(ntiles) * 4));
	CHECK(ensure(c, c->totals, 64)); CHECK(ensure_host(c, c->h_totals, 64));
	const int64_t need = std::min<int64_t>(c->isz_max, c->isz_count + d.n);
	CHECK(ensure(c, c->isz_vals, (size_t)need * 4 + 16, true, (size_t)c->isz_count * 4));
	CHECK(ensure(c, c->isz_tmp, (size_t)d.n * 4 + 16));
	k_isize_count<<<(unsigned)ntiles, BLOCK, 0, c->st>>>(d, c->isz_min_mapq, P<int32_t>(c->isz_tmp), P<uint32_t>(c->tile_cnt));
	exclusive_scan<uint32_t, uint32_t>(c->st, P<uint32_t>(c->tile_cnt), P<uint32_t>(c->tile_base), ntiles, 0u, P<uint32_t>(c->scan_scratch), P<uint32_t>(c->totals));
	k_isize_collect<<<(unsigned)ntiles, BLOCK, 0, c->st>>>(P<int32_t>(c->isz_tmp), d.n, P<uint32_t>(c->tile_base), c->isz_count, c->isz_max, P<int32_t>(c->isz_vals));
	HIPCHECK(c, hipGetLastError());
	HIPCHECK(c, hipMemcpyAsync(c->h_totals.p, c->totals.p, 4, hipMemcpyDeviceToHost, c->st));
	HIPCHECK(c, hipStreamSynchronize(c->st));
	c->isz_count = std::min<int64_t>(c->isz_max, c->isz_count + *P<uint32_t>(c->h_totals));
	if (done) *done = c->isz_count >= c->isz_max;
	return SSV_OK;
}

int ssv_isize_finish(ssv_ctx *c, int64_t *n_pairs, int32_t *mean, int32_t *sd)
{
	if (!c || !n_pairs || !mean || !sd) return SSV_E_ARG;
	if (!c->isz_active) { c->err = "ssv_isize_finish before ssv_isize_begin"; return SSV_E_STATE; }
	HIPCHECK(c, hipSetDevice(c->device));
	c->isz_active = false;
	const int64_t n = c->isz_count;
	*n_pairs = n;
	if (n == 0) { HIPCHECK(c, hipStreamSynchronize(c->st)); return SSV_OK; } // cluster.cpp:71: mean / sd untouched
	ProfScope ps(c, P_ISIZE, 0);
	CHECK(ensure(c, c->isz_acc, 16)); CHECK(ensure_host(c, c->h_totals, 64));
	long long *acc = P<long long>(c->isz_acc);
	unsigned grid = (unsigned)std::min<int64_t>(1024, (n + BLOCK - 1) / BLOCK);
	HIPCHECK(c, hipMemsetAsync(acc, 0, 16, c->st));
	k_isize_reduce<<<grid, BLOCK, 0, c->st>>>(P<int32_t>(c->isz_vals), n, 0, nullptr, acc);
	k_isize_reduce<<<grid, BLOCK, 0, c->st>>>(P<int32_t>(c->isz_vals), n, 1, acc, acc + 1);
	HIPCHECK(c, hipMemcpyAsync(c->h_totals.p, acc, 16, hipMemcpyDeviceToHost, c->st));
	HIPCHECK(c, hipStreamSynchronize(c->st));
	const unsigned long total = (unsigned long)P<long long>(c->h_totals)[0];
	const int m = (int)(total / (unsigned long)n); // cluster.cpp:72
	// cluster.cpp:73-80 adds the (int) squares one by one into a double; the exact integer sum is the same value as long as it
	// stays below 2^53 (5e6 pairs * 2^31 is ~2^53.2: only reachable with absurd insert sizes)
	const double dsum = (double)P<long long>(c->h_totals)[1];
	*mean = m;
	*sd = (int)std::sqrt(dsum / (double)n);
	return SSV_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// getsv: discordant tally + depth
// ---------------------------------------------------------------------------------------------------------------------

static int gs_build_tilemap(ssv_ctx *c, int32_t span)
{
	// which 512-bp tiles can hold the start (0-based pos) of a record that overlaps a depth window / is a candidate of a junction window,
	// and where such a record's look-up starts: marked by the windows and junctions themselves, on the device
	const size_t ntile = (size_t)c->gs_ctg_tile_off.back();
	CHECK(ensure(c, c->gs_tilemap, ntile + 16)); CHECK(ensure(c, c->gs_tile_win, ntile * 4 + 16)); CHECK(ensure(c, c->gs_tile_junc, ntile * 4 + 16));
	HIPCHECK(c, hipMemsetAsync(c->gs_tilemap.p, 0, ntile + 16, c->st));
	const int64_t nw = (int64_t)c->gs_win.size(), nj = (int64_t)c->gs_junc.size();
	if (nw) k_tile_mark_windows<<<grid_for(nw, BLOCK), BLOCK, 0, c->st>>>(P<int32_t>(c->gs_wtid), P<int32_t>(c->gs_wbeg), P<int32_t>(c->gs_wend), nw, span, P<int64_t>(c->gs_ctgoff),
	                                                                        c->gs_p.n_targets, P<uint32_t>(c->gs_tilemap), P<uint32_t>(c->gs_tile_win));
	if (nj) k_tile_mark_junctions<<<grid_for(nj, BLOCK), BLOCK, 0, c->st>>>(P<DevJunction>(c->gs_djunc), nj, span, c->gs_wmax, P<int64_t>(c->gs_ctgoff), c->gs_p.n_targets,
	                                                                          P<uint32_t>(c->gs_tilemap), P<uint32_t>(c->gs_tile_junc));
	HIPCHECK(c, hipGetLastError());
	c->gs_map_span = span;
	return SSV_OK;
}

int ssv_getsv_begin(ssv_ctx *c, const ssv_getsv_params *p)
{
	if (!c || !p || p->n_targets < 0 || p->n_junctions < 0 || p->n_windows < 0) return SSV_E_ARG;
	if ((p->n_targets && !p->target_len) || (p->n_junctions && !p->junctions) || (p->n_windows && !p->windows)) return SSV_E_ARG;
	c->pf.clear();
	HIPCHECK(c, hipSetDevice(c->device));
	c->gs_p = *p;
	c->gs_tlen.assign(p->target_len, p->target_len + p->n_targets);
	c->gs_ctg_tile_off.assign((size_t)p->n_targets + 1, 0);
	for (int t = 0; t < p->n_targets; ++t) c->gs_ctg_tile_off[(size_t)t + 1] = c->gs_ctg_tile_off[(size_t)t] + ((int64_t)std::max(0, c->gs_tlen[(size_t)t]) >> TILE_SHIFT) + 1;
	c->gs_map_span = -1;
	// junctions sorted by (up_tid, beg); the window test itself is repeated exactly on the device
	c->gs_junc.clear();
	c->gs_wmax = 0;
	for (int64_t k = 0; k < p->n_junctions; ++k) {
		const ssv_junction &s = p->junctions[k];
		DevJunction j;
		j.up_tid = s.up_tid; j.down_tid = s.down_tid; j.up_pos = s.up_pos; j.down_pos = s.down_pos; j.beg = s.beg; j.end = s.end;
		j.up_strand = s.up_strand; j.down_strand = s.down_strand; j.pad = 0; j.orig = (int32_t)k;
		c->gs_junc.push_back(j);
		if ((int64_t)s.end - s.beg > c->gs_wmax) c->gs_wmax = (int32_t)std::min<int64_t>((int64_t)s.end - s.beg, 0x7fffffff);
	}
	std::stable_sort(c->gs_junc.begin(), c->gs_junc.end(), [](const DevJunction &a, const DevJunction &b) { return a.up_tid != b.up_tid ? a.up_tid < b.up_tid : a.beg < b.beg; });
	c->gs_win.assign(p->windows, p->windows + p->n_windows);
	for (size_t k = 0; k < c->gs_win.size(); ++k) {
		const ssv_interval &w = c->gs_win[k];
		if (w.end < w.beg || (k && (c->gs_win[k - 1].tid > w.tid || (c->gs_win[k - 1].tid == w.tid && c->gs_win[k - 1].end >= w.beg)))) {
			c->err = "depth windows must be sorted, disjoint and non-empty"; return SSV_E_ARG;
		}
	}
	const size_t nj = c->gs_junc.size(), nw = c->gs_win.size();
	std::vector<int32_t> wt(nw), wb(nw), we(nw);
	std::vector<int64_t> wo(nw + 1, 0);
	for (size_t k = 0; k < nw; ++k) { wt[k] = c->gs_win[k].tid; wb[k] = c->gs_win[k].beg; we[k] = c->gs_win[k].end; wo[k + 1] = wo[k] + ((int64_t)we[k] - wb[k] + 1) + 1; }
	c->gs_diff_len = wo[nw];
	CHECK(ensure(c, c->gs_djunc, nj * sizeof(DevJunction) + 16)); CHECK(ensure(c, c->gs_counts, nj * 4 + 16));
	CHECK(ensure(c, c->gs_wtid, nw * 4 + 16)); CHECK(ensure(c, c->gs_wbeg, nw * 4 + 16)); CHECK(ensure(c, c->gs_wend, nw * 4 + 16)); CHECK(ensure(c, c->gs_woff, (nw + 1) * 8));
	CHECK(ensure(c, c->gs_diff, (size_t)c->gs_diff_len * 4 + 16)); CHECK(ensure(c, c->gs_ctgoff, c->gs_ctg_tile_off.size() * 8)); CHECK(ensure(c, c->gs_maxdepth, 16));
	CHECK(ensure(c, c->gs_span, 16));
	if (nj) HIPCHECK(c, hipMemcpyAsync(c->gs_djunc.p, c->gs_junc.data(), nj * sizeof(DevJunction), hipMemcpyHostToDevice, c->st));
	if (nw) {
		HIPCHECK(c, hipMemcpyAsync(c->gs_wtid.p, wt.data(), nw * 4, hipMemcpyHostToDevice, c->st));
		HIPCHECK(c, hipMemcpyAsync(c->gs_wbeg.p, wb.data(), nw * 4, hipMemcpyHostToDevice, c->st));
		HIPCHECK(c, hipMemcpyAsync(c->gs_wend.p, we.data(), nw * 4, hipMemcpyHostToDevice, c->st));
	}
	HIPCHECK(c, hipMemcpyAsync(c->gs_woff.p, wo.data(), (nw + 1) * 8, hipMemcpyHostToDevice, c->st));
	HIPCHECK(c, hipMemcpyAsync(c->gs_ctgoff.p, c->gs_ctg_tile_off.data(), c->gs_ctg_tile_off.size() * 8, hipMemcpyHostToDevice, c->st));
	HIPCHECK(c, hipMemsetAsync(c->gs_counts.p, 0, nj * 4 + 16, c->st));
	HIPCHECK(c, hipMemsetAsync(c->gs_diff.p, 0, (size_t)c->gs_diff_len * 4 + 16, c->st));
	HIPCHECK(c, hipMemsetAsync(c->gs_maxdepth.p, 0, 16, c->st));
	CHECK(ensure(c, c->cap_carry, sizeof(CapCarry))); CHECK(ensure(c, c->cap_flags, 16));
	HIPCHECK(c, hipMemsetAsync(c->cap_carry.p, 0, sizeof(CapCarry), c->st));
	c->cap_tail_n = 0; c->cap_tail_cur = 0; c->cap_ring_mask = 0;
	HIPCHECK(c, hipStreamSynchronize(c->st)); // the host vectors above go out of scope
	c->gs_active = true;
	return SSV_OK;
}

// the read cap of the reference's pileup: three small launches that leave at once unless >= 8000 reads can be alive somewhere.
// prime != 0 (ssv_getsv_prime): the batch only rebuilds the bookkeeping (ring of read ends, live count, the stream's last records).
static int cap_launches(ssv_ctx *c, const GetsvArgs &a, const DevBatch &d, int64_t ntiles, int prime)
{
	ProfScope ps(c, P_GETSV_CAND, 0);
	// the ring of read ends covers one reference span; a later batch with a longer read (a long N skip or deletion) makes it grow: the
	// live entries of a sweep that is carried across the batch boundary move to their slots in the larger ring
	if (c->cap_ring_mask == 0 || (int64_t)c->gs_map_span + 2 > (int64_t)c->cap_ring_mask + 1) {
		int64_t e = CAP_LDS_RING;
		while (e < (int64_t)c->gs_map_span + 2) e <<= 1;
		if (e > (1ll << 30)) { c->err = "reference span of a read beyond 2^30"; return SSV_E_RANGE; }
		if (c->cap_ring_mask == 0) CHECK(ensure(c, c->cap_ring, (size_t)e * 4));
		else {
			CHECK(ensure(c, c->cap_ring_tmp, (size_t)e * 4));
			HIPCHECK(c, hipMemsetAsync(c->cap_ring_tmp.p, 0, (size_t)e * 4, c->st));
			k_cap_regrow<<<64, BLOCK, 0, c->st>>>(P<CapCarry>(c->cap_carry), P<int32_t>(c->cap_ring), c->cap_ring_mask, P<int32_t>(c->cap_ring_tmp), (int32_t)(e - 1));
			HIPCHECK(c, hipGetLastError());
			std::swap(c->cap_ring, c->cap_ring_tmp);
		}
		c->cap_ring_mask = (int32_t)(e - 1);
	}
	CHECK(ensure(c, c->cap_deep, (size_t)ntiles + 16));
	for (int s_ = 0; s_ < 2; ++s_) { CHECK(ensure(c, c->cap_tail[s_][0], CAP_TAIL * 4)); CHECK(ensure(c, c->cap_tail[s_][1], CAP_TAIL * 4)); CHECK(ensure(c, c->cap_tail[s_][2], CAP_TAIL * 4)); CHECK(ensure(c, c->cap_tail[s_][3], CAP_TAIL)); }
	CapArgs ca;
	ca.g = a; ca.span = c->gs_map_span; ca.prime = prime;
	DBuf *ot = c->cap_tail[c->cap_tail_cur], *nt = c->cap_tail[c->cap_tail_cur ^ 1];
	ca.tail_tid = P<int32_t>(ot[0]); ca.tail_pos = P<int32_t>(ot[1]); ca.tail_end = P<int32_t>(ot[2]); ca.tail_pass = P<uint8_t>(ot[3]); ca.tail_n = c->cap_tail_n;
	ca.deep = P<uint8_t>(c->cap_deep); ca.ntiles = ntiles; ca.flags = P<int>(c->cap_flags); ca.carry = P<CapCarry>(c->cap_carry);
	ca.ring = P<int32_t>(c->cap_ring); ca.ring_mask = c->cap_ring_mask;
	ca.ntail_tid = P<int32_t>(nt[0]); ca.ntail_pos = P<int32_t>(nt[1]); ca.ntail_end = P<int32_t>(nt[2]); ca.ntail_pass = P<uint8_t>(nt[3]);
	ca.ntail_n = (int32_t)std::min<int64_t>(CAP_TAIL, (int64_t)c->cap_tail_n + d.n);
	k_cap_mark<<<(unsigned)std::min<int64_t>(ntiles, 1024), BLOCK, 0, c->st>>>(ca);
	k_cap_sweep<<<1, WAVE, 0, c->st>>>(ca);
	k_cap_tail<<<grid_for(ca.ntail_n, BLOCK), BLOCK, 0, c->st>>>(ca);
	HIPCHECK(c, hipGetLastError());
	c->cap_tail_n = ca.ntail_n; c->cap_tail_cur ^= 1;
	return SSV_OK;
}

int ssv_getsv_scan(ssv_ctx *c, const ssv_batch_t *b)
{
	if (!c || !b) return SSV_E_ARG;
	if (!c->gs_active) { c->err = "ssv_getsv_scan before ssv_getsv_begin"; return SSV_E_STATE; }
	HIPCHECK(c, hipSetDevice(c->device));
	if (b->n == 0) return SSV_OK;
	DevBatch d;
	CHECK(stage_batch(c, b, d));
	if (!d.cigar) { c->err = "batch without cigar"; return SSV_E_ARG; }
	int32_t span = d.max_ref_span;
	if (span <= 0) { // unknown: measure it
		HIPCHECK(c, hipMemsetAsync(c->gs_span.p, 0, 16, c->st));
		k_max_span<<<grid_for(d.n, BLOCK), BLOCK, 0, c->st>>>(d, P<int>(c->gs_span));
		CHECK(ensure_host(c, c->h_totals, 64));
		HIPCHECK(c, hipMemcpyAsync(c->h_totals.p, c->gs_span.p, 4, hipMemcpyDeviceToHost, c->st));
		HIPCHECK(c, hipStreamSynchronize(c->st));
		span = std::max(1, *P<int>(c->h_totals));
	}
	if (span > c->gs_map_span) CHECK(gs_build_tilemap(c, span));
	GetsvArgs a;
	CHECK(fill_runs(c, b, a.runs));
	static const bool verify_runs = getenv("SSV_VERIFY_RUNS") && atoi(getenv("SSV_VERIFY_RUNS")) != 0;
	if (verify_runs && a.runs.n > 0) { // the run list against the column it stands for (the scan below never reads that column where a run covers a tile)
		CHECK(ensure(c, c->counters, sizeof(ClipCounters)));
		CHECK(ensure_host(c, c->h_counters, sizeof(ClipCounters)));
		HIPCHECK(c, hipMemsetAsync(c->counters.p, 0, sizeof(ClipCounters), c->st));
		k_verify_runs<<<(unsigned)std::min<int64_t>(256 * 8, (d.n + BLOCK * 4 - 1) / (BLOCK * 4)), BLOCK, 0, c->st>>>(d.tid, d.n, a.runs, &P<ClipCounters>(c->counters)->n_cand);
		HIPCHECK(c, hipGetLastError());
		HIPCHECK(c, hipMemcpyAsync(c->h_counters.p, c->counters.p, sizeof(ClipCounters), hipMemcpyDeviceToHost, c->st));
		HIPCHECK(c, hipStreamSynchronize(c->st));
		const unsigned long long badrec = P<ClipCounters>(c->h_counters)->n_cand;
		if (badrec) { c->err = "tid_runs disagree with the tid column at record " + std::to_string(badrec - 1) + " (SSV_VERIFY_RUNS)"; return SSV_E_ARG; }
	}
	a.b = d; a.tilemap = P<uint8_t>(c->gs_tilemap); a.tile_win = P<uint32_t>(c->gs_tile_win); a.tile_junc = P<uint32_t>(c->gs_tile_junc); a.ctg_tile_off = P<int64_t>(c->gs_ctgoff); a.n_targets = c->gs_p.n_targets;
	a.junc = P<DevJunction>(c->gs_djunc); a.n_junc = (int64_t)c->gs_junc.size(); a.junc_wmax = c->gs_wmax;
	a.mean = c->gs_p.mean; a.sd = c->gs_p.sd; a.times = c->gs_p.times; a.disc_min_mapq = c->gs_p.disc_min_mapq;
	a.min_ins = std::max(0, a.mean - a.sd * a.times); a.max_ins = a.mean + a.sd * a.times; // getsv.cpp:1032-1034
	a.counts = P<int32_t>(c->gs_counts);
	a.win_tid = P<int32_t>(c->gs_wtid); a.win_beg = P<int32_t>(c->gs_wbeg); a.win_end = P<int32_t>(c->gs_wend); a.win_off = P<int64_t>(c->gs_woff);
	a.n_win = (int64_t)c->gs_win.size(); a.depth_min_mapq = c->gs_p.depth_min_mapq; a.diff = P<int32_t>(c->gs_diff);
	a.cap_flag = nullptr; a.cap_span = c->gs_map_span;
	HIPCHECK(c, hipMemsetAsync(c->cap_flags.p, 0, 16, c->st)); // ([0], [1]: the read cap's flags, [2]: the length of the dense tiles' list)
	if (a.n_win > 0) a.cap_flag = P<int>(c->cap_flags);
	const int64_t ntiles = (d.n + CS_TILE - 1) / CS_TILE;
	const unsigned grid = scan_blocks(ntiles, "SSV_GETSV_SCAN_BLOCKS", 256 * 4);
	CHECK(ensure(c, c->tile_cnt, ntiles * 4));
	CHECK(ensure(c, c->tile_off, ntiles * 4));
	CHECK(ensure(c, c->counters, sizeof(ClipCounters)));
	CHECK(ensure_host(c, c->h_counters, sizeof(ClipCounters)));
	if (c->stage_cap == 0) c->stage_cap = std::max<int64_t>(1 << 16, d.n / 8);
	ClipCounters *hc = P<ClipCounters>(c->h_counters);
	ClipCounters *dc = P<ClipCounters>(c->counters);
	GetsvStage g;
	for (int attempt = 0;; ++attempt) {
		const int64_t block_cap = (c->stage_cap + grid - 1) / grid;
		CHECK(ensure(c, c->stage, (size_t)block_cap * grid * 4));
		HIPCHECK(c, hipMemsetAsync(c->counters.p, 0, sizeof(ClipCounters), c->st));
		g.tile_cnt = P<uint32_t>(c->tile_cnt); g.tile_off = P<uint32_t>(c->tile_off); g.stage = P<uint32_t>(c->stage); g.block_cap = block_cap;
		g.overflow = &dc->overflow; g.ntiles = ntiles;
		g.dense_list = nullptr; g.dense_n = nullptr; g.n_cand = &dc->n_cand;
		{
			ProfScope ps(c, P_GETSV_SCAN, d.n);
			if (a.runs.n > 0 && d.n >= CS_TILE) k_getsv_scan_runs<<<grid, BLOCK, 0, c->st>>>(a, g); // (the tid column as runs: never read)
			else k_getsv_scan<<<grid, BLOCK, 0, c->st>>>(a, g);
		}
		HIPCHECK(c, hipGetLastError());
		HIPCHECK(c, hipMemcpyAsync(hc, c->counters.p, sizeof(ClipCounters), hipMemcpyDeviceToHost, c->st));
		HIPCHECK(c, hipStreamSynchronize(c->st));
		if (!hc->overflow) break;
		if (attempt > 6) { c->err = "getsv staging overflow"; return SSV_E_HIP; }
		c->stage_cap *= 4; // a workgroup's private region was too small for the records near its windows
	}
	{
		ProfScope ps(c, P_GETSV_CAND, d.n);
		// tiles that are dense with candidates are listed and left to the second kernel; its grid: the scan's count bounds the list's length
		const int64_t dense_max = std::min<int64_t>(ntiles, (int64_t)(hc->n_cand / GC_DENSE_MIN));
		if (dense_max > 0) {
			CHECK(ensure(c, c->dense_list, (size_t)dense_max * sizeof(DenseTile)));
			g.dense_list = P<DenseTile>(c->dense_list); g.dense_n = P<int>(c->cap_flags) + 2; // (zeroed with the cap flags above)
			k_dense_tiles<<<grid_for(ntiles, BLOCK), BLOCK, 0, c->st>>>(a, g);
		}
		k_getsv_cand<<<grid_for(ntiles, WAVES_PER_BLOCK), BLOCK, 0, c->st>>>(a, g);
		if (g.dense_list) k_getsv_cand_dense<<<(unsigned)dense_max, BLOCK, 0, c->st>>>(a, g);
	}
	HIPCHECK(c, hipGetLastError());
	if (a.n_win > 0) CHECK(cap_launches(c, a, d, ntiles, 0));
	return SSV_OK;
}

int ssv_getsv_prime(ssv_ctx *c, const ssv_batch_t *b, int32_t *sufficient)
{
	if (!c || !b || !sufficient) return SSV_E_ARG;
	if (!c->gs_active) { c->err = "ssv_getsv_prime before ssv_getsv_begin"; return SSV_E_STATE; }
	if (c->cap_tail_n != 0) { c->err = "ssv_getsv_prime after records were scanned"; return SSV_E_STATE; }
	HIPCHECK(c, hipSetDevice(c->device));
	*sufficient = 1;
	if (b->n == 0 || c->gs_win.empty()) return SSV_OK; // no depth pass: nothing to rebuild
	DevBatch d;
	CHECK(stage_batch(c, b, d));
	int32_t span = d.max_ref_span;
	if (span <= 0) {
		HIPCHECK(c, hipMemsetAsync(c->gs_span.p, 0, 16, c->st));
		k_max_span<<<grid_for(d.n, BLOCK), BLOCK, 0, c->st>>>(d, P<int>(c->gs_span));
		CHECK(ensure_host(c, c->h_totals, 128));
		HIPCHECK(c, hipMemcpyAsync(c->h_totals.p, c->gs_span.p, 4, hipMemcpyDeviceToHost, c->st));
		HIPCHECK(c, hipStreamSynchronize(c->st));
		span = std::max(1, *P<int>(c->h_totals));
	}
	if (span > c->gs_map_span) CHECK(gs_build_tilemap(c, span));
	GetsvArgs a;
	memset(&a, 0, sizeof(a));
	a.b = d; a.depth_min_mapq = c->gs_p.depth_min_mapq; a.n_targets = c->gs_p.n_targets; a.cap_span = c->gs_map_span;
	a.tilemap = P<uint8_t>(c->gs_tilemap); a.tile_win = P<uint32_t>(c->gs_tile_win); a.ctg_tile_off = P<int64_t>(c->gs_ctgoff);
	a.win_tid = P<int32_t>(c->gs_wtid); a.win_beg = P<int32_t>(c->gs_wbeg); a.win_end = P<int32_t>(c->gs_wend); a.win_off = P<int64_t>(c->gs_woff);
	a.n_win = (int64_t)c->gs_win.size(); a.diff = P<int32_t>(c->gs_diff);
	// every tile is looked at (the streaming pass that usually raises this flag does not run over a replayed batch)
	int one[4] = {1, 0, 0, 0};
	HIPCHECK(c, hipMemcpyAsync(c->cap_flags.p, one, 16, hipMemcpyHostToDevice, c->st));
	HIPCHECK(c, hipStreamSynchronize(c->st)); // (`one` lives on this stack)
	const int64_t ntiles = (d.n + CS_TILE - 1) / CS_TILE;
	CHECK(cap_launches(c, a, d, ntiles, 1));
	// The replay leaves the right state if it started from one: a sweep that begins >= 7,999 records before the first "deep" record does
	// (getsv_kernels.h).  A record can be judged from index 7,998 of the batch on; so the records [7998, 15997) - inside tiles 1..3 - must not
	// be deep.  A batch that starts at the file's first record is exact anyway: the caller knows that case and ignores the answer.
	CHECK(ensure_host(c, c->h_totals, 128));
	const int64_t nt = std::min<int64_t>(ntiles, 4);
	HIPCHECK(c, hipMemcpyAsync(c->h_totals.p, c->cap_deep.p, (size_t)nt, hipMemcpyDeviceToHost, c->st));
	HIPCHECK(c, hipStreamSynchronize(c->st));
	for (int64_t t = 1; t < nt; ++t) if (P<uint8_t>(c->h_totals)[t]) *sufficient = 0;
	if (ntiles < 4) *sufficient = 0; // too short to tell
	return SSV_OK;
}

int ssv_getsv_finish(ssv_ctx *c, int32_t *counts, const ssv_interval *ranges, int64_t n_ranges, uint64_t *range_sum,
                     const ssv_interval *points, int64_t n_points, int32_t *point_depth, int32_t *max_depth)
{
	if (!c || n_ranges < 0 || n_points < 0 || (n_ranges && (!ranges || !range_sum)) || (n_points && (!points || !point_depth))) return SSV_E_ARG;
	if (!c->gs_active) { c->err = "ssv_getsv_finish before ssv_getsv_begin"; return SSV_E_STATE; }
	HIPCHECK(c, hipSetDevice(c->device));
	c->gs_active = false;
	const int64_t nw = (int64_t)c->gs_win.size(), nj = (int64_t)c->gs_junc.size();
	ProfScope ps(c, P_DEPTH_FINISH, nw);
	if (nw) k_depth_prefix<<<grid_for(nw, WAVES_PER_BLOCK), BLOCK, 0, c->st>>>(P<int64_t>(c->gs_woff), nw, P<int32_t>(c->gs_diff), P<int32_t>(c->gs_maxdepth));
	// queries up, answers down: one pinned buffer each way, one synchronisation
	//   up:   [range tid | range beg | range end | point tid | point beg]        (int32 each)
	//   down: [range sums u64 | point depths i32 | counts i32 | max depth i32]
	const size_t up_bytes = ((size_t)n_ranges * 3 + (size_t)n_points * 2) * 4;
	const size_t o_pd = (size_t)n_ranges * 8, o_cnt = o_pd + (size_t)n_points * 4, o_max = o_cnt + (size_t)nj * 4, down_bytes = o_max + 4;
	CHECK(ensure_host(c, c->h_q, up_bytes + down_bytes + 64));
	CHECK(ensure(c, c->q_tid, up_bytes + 16)); CHECK(ensure(c, c->q_out64, down_bytes + 16));
	int32_t *up = P<int32_t>(c->h_q);
	uint8_t *down = P<uint8_t>(c->h_q) + ((up_bytes + 15) & ~(size_t)15);
	int32_t *rt = up, *rb = rt + n_ranges, *re = rb + n_ranges, *pt = re + n_ranges, *pb = pt + n_points;
	for (int64_t k = 0; k < n_ranges; ++k) { rt[k] = ranges[k].tid; rb[k] = ranges[k].beg; re[k] = ranges[k].end; }
	for (int64_t k = 0; k < n_points; ++k) { pt[k] = points[k].tid; pb[k] = points[k].beg; }
	int32_t *d_up = P<int32_t>(c->q_tid);
	uint8_t *d_down = P<uint8_t>(c->q_out64);
	if (up_bytes) HIPCHECK(c, hipMemcpyAsync(d_up, up, up_bytes, hipMemcpyHostToDevice, c->st));
	if (n_ranges) k_range_sum<<<grid_for(n_ranges, WAVES_PER_BLOCK), BLOCK, 0, c->st>>>(P<int32_t>(c->gs_wtid), P<int32_t>(c->gs_wbeg), P<int32_t>(c->gs_wend), P<int64_t>(c->gs_woff), nw,
	                                                                                  P<int32_t>(c->gs_diff), d_up, d_up + n_ranges, d_up + 2 * n_ranges, n_ranges,
	                                                                                  reinterpret_cast<unsigned long long *>(d_down));
	if (n_points) k_point_depth<<<grid_for(n_points, BLOCK), BLOCK, 0, c->st>>>(P<int32_t>(c->gs_wtid), P<int32_t>(c->gs_wbeg), P<int32_t>(c->gs_wend), P<int64_t>(c->gs_woff), nw,
	                                                                          P<int32_t>(c->gs_diff), d_up + 3 * n_ranges, d_up + 3 * n_ranges + n_points, n_points,
	                                                                          reinterpret_cast<int32_t *>(d_down + o_pd));
	if (nj) HIPCHECK(c, hipMemcpyAsync(d_down + o_cnt, c->gs_counts.p, (size_t)nj * 4, hipMemcpyDeviceToDevice, c->st));
	HIPCHECK(c, hipMemcpyAsync(d_down + o_max, c->gs_maxdepth.p, 4, hipMemcpyDeviceToDevice, c->st));
	HIPCHECK(c, hipGetLastError());
	HIPCHECK(c, hipMemcpyAsync(down, d_down, down_bytes, hipMemcpyDeviceToHost, c->st));
	HIPCHECK(c, hipStreamSynchronize(c->st));
	if (n_ranges) memcpy(range_sum, down, (size_t)n_ranges * 8);
	if (n_points) memcpy(point_depth, down + o_pd, (size_t)n_points * 4);
	if (counts && nj) memcpy(counts, down + o_cnt, (size_t)nj * 4);
	if (max_depth) memcpy(max_depth, down + o_max, 4);
	return SSV_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// profiling
// ---------------------------------------------------------------------------------------------------------------------

int ssv_prof_enable(ssv_ctx *c, int on)
{
	if (!c) return SSV_E_ARG;
	prof_collect(c);
	c->prof_mode = on;
	return SSV_OK;
}

int ssv_prof_reset(ssv_ctx *c)
{
	if (!c) return SSV_E_ARG;
	prof_collect(c);
	for (int k = 0; k < P_COUNT; ++k) { c->prof_ms[k] = 0; c->prof_launches[k] = 0; c->prof_units[k] = 0; }
	return SSV_OK;
}

int ssv_prof_get(ssv_ctx *c, const char *name, double *total_ms, int64_t *launches, int64_t *units)
{
	if (!c || !name) return SSV_E_ARG;
	prof_collect(c);
	for (int k = 0; k < P_COUNT; ++k) {
		if (strcmp(name, kProfNames[k]) == 0) {
			if (total_ms) *total_ms = c->prof_ms[k];
			if (launches) *launches = c->prof_launches[k];
			if (units) *units = c->prof_units[k];
			return SSV_OK;
		}
	}
	return SSV_E_ARG;
}

const char *ssv_prof_names(void) { return kProfNameList; }

#include "bamdec_api.inc"
#include "realign_api.inc"
#include "group_api.inc"

} // extern "C"
