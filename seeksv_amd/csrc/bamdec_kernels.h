// bamdec_kernels.h - BGZF inflate and BAM record decode on the device.
//
// Replaces, for whole-file passes, libbam's samread() (sam/sam.h:73: bgzf inflate + bam_read1) that the reference calls once per
// record on one core.  Data flow of one chunk (a run of whole BGZF blocks, compressed bytes already in HBM):
//   k_bgzf_tokens(_wave) + k_bgzf_resolve_win   the inflate as two passes: a wavefront per BGZF block decodes its symbols (literals in place, a 32-bit token per
//                     match; a lane per block above 64 K blocks a chunk), a wavefront per block then fills the matches' holes inside a window of the block in LDS
//   k_find_records    one lane per block: speculate the first record start inside the block (three plausible headers in a row), follow
//                     the block_size chain to the block end: guess, exit offset, record count
//   k_stitch_blocks   one wavefront: verify every guess against the true chain (exit of the previous block), repair wrong ones by
//                     following the chain by hand; carries the position across blocks -> exact record starts, whatever the guesses
//   k_list_records    per block: write the record offsets at their ranks (exclusive scan of the counts)
//   k_record_fields   per record: fixed fields -> structure-of-arrays columns; CIGAR count, soft-clip test, bytes to ship
//   k_record_cigars   per record: CIGAR ops to their places (out of a two-operation stash the fields kernel left, or out of the stream); the list of records
//                     that ship bases / owe an XC flag
//   k_record_seqs     per listed record (16 lanes): packed bases + qualities, XC aux flag
//   k_raw_copy        UNMAP|MUNMAP records as raw bytes for the host's unmapped-FASTQ side channel (clip_reads.h:415-419)
//   k_tid_runs        contig changes among the mapped-pair records (the flush sequence of clip_reads.h:423-438) for the host
// The chain of record starts is a dependent-load chain; the speculation makes all but the (tiny) stitch parallel, and the stitch
// makes the result independent of the speculation.
#pragma once

#include "common.h"
#include "inflate_core.h"
#include "resolve_wave.h"
#include "inflate_wave.h"

namespace ssv {

constexpr uint32_t BD_NONE = 0xffffffffu;

// Huffman tables of the 64 lanes of a wavefront.  The decoder is latency bound (every symbol is a chain of dependent LDS and global
// accesses), so what matters is how many wavefronts a CU can hold, i.e. LDS bytes per lane.  Only what the symbol loop reads stays in
// LDS, element i of lane l at [i * 64 + l] (each lane its own bank column):
//   lit8  288 B   low byte of the literal/length permutation        hi  9 dwords  its ninth bit (symbols >= 256 are lengths / end of block)
//   dst8   32 B   distance permutation (also borrowed by the code-length code while a block header is read)
// = 356 B per lane, 22,784 B per wavefront: seven wavefronts per CU (the u16 table + lengths + offsets of the first version: three).
// The code lengths and the running offsets are only touched while a block header is parsed: they live in global scratch, same
// interleaving (coalesced 64-byte rows).
template <int STRIDE> struct LdsTabT {
	uint8_t *lit8; uint32_t *hi; uint8_t *dst8;  // LDS
	uint8_t *len8; uint16_t *off16;              // global scratch of this wavefront: 320 x stride bytes, 16 x stride halves
	int lane;
	static constexpr int stride = STRIDE;        // lanes that decode in this wavefront (element i of lane l at [i * stride + l])
	__device__ __forceinline__ uint16_t lit_get(int i) const { return (uint16_t)(lit8[i * stride + lane] | (((hi[(i >> 5) * stride + lane] >> (i & 31)) & 1u) << 8)); }
	__device__ __forceinline__ void lit_set(int i, uint16_t v)
	{
		lit8[i * stride + lane] = (uint8_t)v;
		uint32_t &w = hi[(i >> 5) * stride + lane];
		w = (w & ~(1u << (i & 31))) | ((uint32_t)(v >> 8) << (i & 31));
	}
	__device__ __forceinline__ uint16_t dst_get(int i) const { return dst8[i * stride + lane]; }
	__device__ __forceinline__ void dst_set(int i, uint16_t v) { dst8[i * stride + lane] = (uint8_t)v; }
	__device__ __forceinline__ int len_get(int i) const { return len8[i * stride + lane]; }
	__device__ __forceinline__ void len_set(int i, int v) { len8[i * stride + lane] = (uint8_t)v; }
	__device__ __forceinline__ uint16_t off_get(int i) const { return off16[i * stride + lane]; }
	__device__ __forceinline__ void off_set(int i, uint16_t v) { off16[i * stride + lane] = v; }
	static constexpr bool has_base = false;
};
using LdsTab = LdsTabT<64>;
// ... plus the two codes' base[length] for the decode by limit compares (huff_decode_lim, inflate_core.h): 2 x 16 halves per lane more in LDS
template <int STRIDE> struct LdsTabLim : LdsTabT<STRIDE> {
	int16_t *base16;
	static constexpr bool has_base = true;
	__device__ __forceinline__ int base_get(int set, int l) const { return base16[(set * 16 + l) * STRIDE + this->lane]; }
	__device__ __forceinline__ void base_set(int set, int l, int v) { base16[(set * 16 + l) * STRIDE + this->lane] = (int16_t)v; }
};
constexpr int TOKENS_BASE_BYTES = 2 * 16 * 2; // per lane

constexpr int INFLATE_LDS_BYTES = 288 * 64 + 9 * 64 * 4 + 32 * 64;      // 22,784 B per wavefront
constexpr int INFLATE_SCRATCH_BYTES = 320 * 64 + 16 * 64 * 2;           // global scratch per wavefront

struct BgzfBlock { uint64_t c_off; uint32_t c_len, u_len; }; // deflate payload inside the chunk buffer; inflated size

// ---- decode and copy split in two (TokenOut, inflate_core.h) ------------------------------------------------------------------------
//
// Pass 1, k_bgzf_tokens: the decoder as above, one lane per block, but a match is only RECORDED (a 32-bit token) - the lane never reads the
// block's output, so its chain per symbol is bits -> table -> store, no trip to memory to wait for.
// Pass 2 (k_bgzf_resolve_win / k_bgzf_resolve_wave, resolve_wave.h) fills the holes.  The forms that lost - one pass with every lane loading and storing for
// itself, a 64-byte line buffer per lane, LDS rings moved by the wavefront, pass 2 with 16 lanes per block or a workgroup per block in LDS - are in the history
// (HISTORY.md, git: rounds 1-4).
// the input window of one lane (RingReader, inflate_core.h): dword j of lane l at word j * LPW + l - a wavefront's lanes hit different banks
template <int LPW>
struct LdsRing {
	uint32_t *w; // this lane's dword 0
	__device__ __forceinline__ uint32_t get(uint32_t j) const { return w[j * LPW]; }
	__device__ __forceinline__ void set(uint32_t j, uint32_t v) { w[j * LPW] = v; }
	__device__ __forceinline__ bool any(bool c) const { return __any(c) != 0; }
};
constexpr uint32_t TOKENS_WINDOW = 64;  // bytes of input per lane in LDS

template <int LPW>
__global__ __launch_bounds__(WAVE) void k_bgzf_tokens(const uint8_t *__restrict__ comp, const BgzfBlock *__restrict__ blocks, const uint64_t *__restrict__ u_off, const uint64_t *__restrict__ tok_off,
                                                      int64_t n_blocks, uint8_t *__restrict__ out, uint32_t *__restrict__ tokens, uint32_t *__restrict__ n_tok, int *__restrict__ status,
                                                      uint8_t *__restrict__ scratch)
{
	extern __shared__ uint8_t lds_raw[];
	if ((int)threadIdx.x >= LPW) return;
	LdsTabLim<LPW> tab;
	tab.lit8 = lds_raw;
	tab.hi = reinterpret_cast<uint32_t *>(lds_raw + 288 * LPW);
	tab.dst8 = lds_raw + 288 * LPW + 9 * LPW * 4;
	tab.base16 = reinterpret_cast<int16_t *>(lds_raw + (INFLATE_LDS_BYTES / 64 + TOKENS_WINDOW) * LPW);
	tab.len8 = scratch + (size_t)blockIdx.x * (INFLATE_SCRATCH_BYTES / 64 * LPW);
	tab.off16 = reinterpret_cast<uint16_t *>(tab.len8 + 320 * LPW);
	tab.lane = (int)threadIdx.x;
	const int64_t b = (int64_t)blockIdx.x * LPW + threadIdx.x;
	if (b >= n_blocks) return;
	const BgzfBlock blk = blocks[b];
	TokenOut to;
	to.out = out + u_off[b];
	to.tok = tokens + tok_off[b];
	int rc = INF_OK;
	if (blk.u_len) {
		LdsRing<LPW> ring{reinterpret_cast<uint32_t *>(lds_raw + INFLATE_LDS_BYTES / 64 * LPW) + threadIdx.x};
		RingReader<LdsRing<LPW>, TOKENS_WINDOW> br(comp + blk.c_off, blk.c_len, ring);
		rc = inflate_stream_from(br, comp + blk.c_off, blk.c_len, to, blk.u_len, tab);
	}
	status[b] = rc;
	n_tok[b] = rc == INF_OK ? to.n : 0u;
}

// pass 1 with a wavefront per block (inflate_wave.h): same outputs as k_bgzf_tokens
template <bool DBG>
__global__ __launch_bounds__(WAVE, 5) void k_bgzf_tokens_wave(const uint8_t *__restrict__ comp, const BgzfBlock *__restrict__ blocks, const uint64_t *__restrict__ u_off, const uint64_t *__restrict__ tok_off,
                                                          int64_t n_blocks, uint8_t *__restrict__ out, uint32_t *__restrict__ tokens, uint32_t *__restrict__ n_tok, int *__restrict__ status,
                                                          unsigned long long *__restrict__ dbg)
{
	__shared__ WaveLds L;
	const int64_t b = blockIdx.x;
	if (b >= n_blocks) return;
	const BgzfBlock blk = blocks[b];
	uint32_t nt = 0;
	int rc = INF_OK;
	if (blk.u_len) rc = wave_inflate_tokens<DBG>(L, comp + blk.c_off, blk.c_len, out + u_off[b], blk.u_len, tokens + tok_off[b], nt, dbg);
	if (threadIdx.x == 0) { status[b] = rc; n_tok[b] = rc == INF_OK ? nt : 0u; }
}

// ---- pass 2 with ONE WAVEFRONT per block (round 4; resolve_wave.h) ----------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_bgzf_resolve_wave(const uint32_t *__restrict__ tokens, const uint64_t *__restrict__ tok_off, const uint32_t *__restrict__ n_tok, const uint64_t *__restrict__ u_off,
                                                             int64_t n_blocks, uint8_t *out)
{
	const int64_t b = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave_id();
	if (b >= n_blocks) return;
	const uint32_t n = n_tok[b];
	if (n == 0u) return;
	WaveResolveState R;
	wave_resolve_tokens(out + u_off[b], tokens + tok_off[b], n, R, lane_id());
}

// ---- pass 2 with a window of the block in LDS, a wavefront per block (round 4, late; resolve_wave.h) ---------------------------------------
template <bool DBG>
__global__ __launch_bounds__(BLOCK) void k_bgzf_resolve_win(const uint32_t *__restrict__ tokens, const uint64_t *__restrict__ tok_off, const uint32_t *__restrict__ n_tok, const uint64_t *__restrict__ u_off,
                                                            const BgzfBlock *__restrict__ blocks, int64_t n_blocks, uint8_t *out, unsigned long long *dbg)
{
	__shared__ __attribute__((aligned(16))) uint8_t s_win[WAVES_PER_BLOCK][RW_WIN + 64];
	const int64_t b = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave_id();
	if (b >= n_blocks) return;
	const uint32_t n = n_tok[b];
	if (n == 0u) return;
	wave_resolve_tokens_win<DBG>(out + u_off[b], blocks[b].u_len, tokens + tok_off[b], n, s_win[wave_id()], lane_id(), dbg);
}

// ---- optional: every inflated block's CRC32 against the one in its BGZF trailer (ssv_bamdec_verify_crc; libbam 0.1.16 checks none) ----------
//
// A wavefront per block.  The block is read in 1 KB chunks, 16 bytes a lane (coalesced; the next chunk's load is in flight while this one is worked on - a lane
// walking its own kilobyte four bytes at a time was a chain of 256 dependent trips to memory: 19.9 ms per 5.3 GB, as long as the inflate itself).  A lane's
// pieces lie 1024 bytes apart, so its value is kept Horner-wise in GF(2)[x] mod the CRC's polynomial: acc = acc x^(8 x 1024) + crc(piece) - the multiplication
// by that constant is four table look-ups (one per byte of acc), the piece's own value sixteen bytes through the slice-by-4 tables from a zero state.  At the
// end every lane multiplies by x^(8 x the bytes behind its last piece), the products are XORed over the wavefront, and the initial state's share
// (0xffffffff x^(8 len)) and the final complement make it the CRC-32 that zlib's crc32() gives.
constexpr int INF_E_CRC = -9; // (beside inflate_core.h's codes: the block's structure was fine, its bytes are not what the writer checksummed)
struct CrcTab {
	uint32_t slice[4][256];  // slice-by-4 tables of the reflected CRC-32 (slice[0] = the byte table)
	uint32_t adv[4][256];    // adv[k][v] = (v << 8 k) x^(8 x 1024): a state moved on by 1024 zero bytes, byte by byte of the state
	uint32_t w16[129];       // x^(8 x 16 m), m = 0 .. 128
	uint32_t xr[16];         // x^(8 r), r = 0 .. 15
	uint32_t x2n[32];        // x^(2^k) (zlib's x2n_table): any other power by square-and-multiply
};

__device__ __forceinline__ uint32_t crc_multmodp(uint32_t a, uint32_t b) // a(x) b(x) mod P, reflected (zlib's multmodp)
{
	uint32_t m = 1u << 31, p = 0;
	for (;;) {
		if (a & m) { p ^= b; if ((a & (m - 1u)) == 0) break; }
		m >>= 1;
		b = (b & 1u) ? (b >> 1) ^ 0xedb88320u : b >> 1;
	}
	return p;
}

__global__ __launch_bounds__(WAVE) void k_bgzf_crc(const uint8_t *__restrict__ comp, const BgzfBlock *__restrict__ blocks, const uint64_t *__restrict__ u_off, int64_t n_blocks,
                                                   const uint8_t *__restrict__ stream, const CrcTab *__restrict__ tab, int *__restrict__ status)
{
	__shared__ uint32_t s_sl[4][256];
	__shared__ uint32_t s_adv[4][256];
	const int lane = (int)threadIdx.x;
	for (int i = lane; i < 1024; i += WAVE) { (&s_sl[0][0])[i] = (&tab->slice[0][0])[i]; (&s_adv[0][0])[i] = (&tab->adv[0][0])[i]; }
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	const int64_t b = blockIdx.x;
	if (b >= n_blocks) return;
	const BgzfBlock blk = blocks[b];
	const uint32_t len = blk.u_len;
	const uint8_t *p = stream + u_off[b];
	auto word = [&](uint32_t c, uint32_t v) -> uint32_t { // four bytes into the state
		c ^= v;
		return s_sl[3][c & 0xffu] ^ s_sl[2][(c >> 8) & 0xffu] ^ s_sl[1][(c >> 16) & 0xffu] ^ s_sl[0][c >> 24];
	};
	auto advance = [&](uint32_t c) -> uint32_t { return s_adv[0][c & 0xffu] ^ s_adv[1][(c >> 8) & 0xffu] ^ s_adv[2][(c >> 16) & 0xffu] ^ s_adv[3][c >> 24]; };
	const uint32_t full = len >> 10, tail = len & 1023u; // whole 1 KB chunks, bytes of the last partial one
	uint32_t acc = 0, end = 0;                            // the lane's value so far; where its last piece ended
	uint4 v = make_uint4(0u, 0u, 0u, 0u);
	if (full) { const uint8_t *q = p + 16u * (uint32_t)lane; v = make_uint4(ld32(q), ld32(q + 4), ld32(q + 8), ld32(q + 12)); }
	for (uint32_t j = 0; j < full; ++j) {
		uint4 nv = make_uint4(0u, 0u, 0u, 0u);
		if (j + 1 < full) { const uint8_t *q = p + 1024u * (j + 1) + 16u * (uint32_t)lane; nv = make_uint4(ld32(q), ld32(q + 4), ld32(q + 8), ld32(q + 12)); }
		uint32_t c = word(0u, v.x);
		c = word(c, v.y); c = word(c, v.z); c = word(c, v.w);
		acc = (j ? advance(acc) : 0u) ^ c;
		end = 1024u * j + 16u * (uint32_t)lane + 16u;
		v = nv;
	}
	if (16u * (uint32_t)lane < tail) { // the last, partial chunk: this lane's piece of it (at most one lane's is shorter than 16 bytes)
		const uint32_t at = 1024u * full + 16u * (uint32_t)lane, r = min(16u, tail - 16u * (uint32_t)lane);
		uint32_t c = 0;
		for (uint32_t i = 0; i < r; ++i) c = s_sl[0][(c ^ p[at + i]) & 0xffu] ^ (c >> 8);
		if (full) acc = r == 16u ? advance(acc) : crc_multmodp(crc_multmodp(tab->w16[63], tab->xr[r]), acc); // the piece before ended 1008 + r bytes earlier
		acc ^= c;
		end = at + r;
	}
	// the bytes behind the lane's last piece: < 2048
	const uint32_t after = end ? len - end : 0u;
	uint32_t term = end ? crc_multmodp(crc_multmodp(tab->w16[after >> 4], tab->xr[after & 15u]), acc) : 0u;
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) term ^= (uint32_t)__shfl_xor((int)term, d, WAVE);
	if (lane == 0) {
		uint32_t n = len, k = 3, f = 1u << 31; // x^(8 len): zlib's x2nmodp(len, 3)
		while (n) { if (n & 1u) f = crc_multmodp(tab->x2n[k & 31u], f); n >>= 1; ++k; }
		const uint32_t crc = ~(term ^ crc_multmodp(f, 0xffffffffu));
		const uint8_t *t = comp + blk.c_off + blk.c_len; // the trailer: CRC32, ISIZE
		const uint32_t stored = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
		if (crc != stored && status[b] == 0) status[b] = INF_E_CRC;
	}
}

// ---- record boundaries ----------------------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t ld_u32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
__device__ __forceinline__ int32_t ld_i32(const uint8_t *p) { int32_t v; memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint16_t ld_u16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }

// could a BAM record start at u[o]?  (o + 36 <= total is the caller's business)
__device__ __forceinline__ bool plausible_record(const uint8_t *u, uint64_t o, uint64_t total, int32_t n_targets, const int32_t *__restrict__ tlen)
{
	const uint32_t bs = ld_u32(u + o);
	if (bs < 32 || bs > (1u << 28)) return false;
	const uint8_t *r = u + o + 4;
	const int32_t refid = ld_i32(r), pos = ld_i32(r + 4), l_seq = ld_i32(r + 16), next_ref = ld_i32(r + 20), next_pos = ld_i32(r + 24);
	const uint32_t l_name = r[8], ncig = ld_u16(r + 12);
	if (refid < -1 || refid >= n_targets || next_ref < -1 || next_ref >= n_targets || pos < -1 || next_pos < -1 || l_seq < 0 || l_name < 2) return false; // (a read name is at least one character and its NUL)
	// a position lies inside its contig (when the contig lengths are known: ssv_bamdec_target_lens) - what catches a word read two bytes ahead of a true
	// record, whose "position" is the true one times 65536
	if (tlen && refid >= 0 && pos >= tlen[refid]) return false; // (the mate's position is left alone: this test must never fail a true record)
	if (32ull + l_name + 4ull * ncig + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq > bs) return false;
	const uint64_t nul = o + 4 + 32 + l_name - 1;
	return nul >= total || u[nul] == 0;
}

// follow the chain from o while records START before `end` and lie completely inside [0, total); returns the count, *exit = where it stopped
__device__ __forceinline__ uint32_t follow_chain(const uint8_t *u, uint64_t o, uint64_t end, uint64_t total, uint64_t *exit, bool *corrupt)
{
	uint32_t n = 0;
	while (o < end) {
		if (o + 4 > total) break;
		const uint32_t bs = ld_u32(u + o);
		if (bs < 32) { *corrupt = true; break; }
		if (o + 4 + (uint64_t)bs > total) break;
		++n;
		o += 4 + (uint64_t)bs;
	}
	*exit = o;
	return n;
}

// plausible headers in a row that make a record start.  A wrong guess costs the stitch 0.6 ms of walking that block by hand (one lane-serial trip to HBM per
// record).  What used to go wrong about once in 2,000 blocks of equal-sized records: two bytes BEFORE a true start, a word made of the previous record's last
// bytes and the true block_size's low half reads as a 17 MB record with an empty name that ends, by the regular spacing, on a true start far ahead - hence the
// name length test in plausible_record.
constexpr int PLAUSIBLE_RUN = 3;
struct BlockChain { uint64_t guess, exit; uint32_t count, listed; }; // guess == ~0: no plausible start found in the block; listed: rel[] holds its record starts
// The record starts k_find_records walks over are kept, as 16-bit offsets from the block's first record, so that k_list_records need not walk the chain again:
// record i of block b at rel[((b / 64) * REL_CAP + i) * 64 + b % 64] (the 64 lanes of a wavefront write 128 contiguous bytes a step).  A block whose chain
// the stitch had to repair, or with more records / a longer reach than fits, is walked again instead (listed = 0).

// The record starts are searched in UNITS: a BGZF block each, or an n-th of one (unit s = part s % n of block s / n; a part behind the block's end is empty).
// Measured with n = 4 (a lane's walk along the block_size chain ~60 dependent loads instead of ~240): k_find_records no faster (0.92 -> 0.98 ms per 19.3 M
// records) - it is not the length of the chains but the number of scattered loads, one per record: 21 G a second is what the memory system gives for
// lines touched once.  So n = 1.
constexpr int UNITS_PER_BLOCK = 1;
constexpr uint32_t REL_CAP = (65536 / UNITS_PER_BLOCK) / 36 + 4; // the records a unit can hold
__device__ __forceinline__ size_t rel_index(int64_t b, uint32_t i) { return ((size_t)(b >> 6) * REL_CAP + i) * 64 + (size_t)(b & 63); }
constexpr uint32_t UNIT_BYTES = 65536 / UNITS_PER_BLOCK;
__device__ __forceinline__ void unit_range(const BgzfBlock *__restrict__ blocks, const uint64_t *__restrict__ u_off, int64_t s, uint64_t &begin, uint64_t &end)
{
	const int64_t b = s / UNITS_PER_BLOCK;
	const uint32_t j = (uint32_t)(s % UNITS_PER_BLOCK), len = blocks[b].u_len;
	const uint32_t lo = j * UNIT_BYTES < len ? j * UNIT_BYTES : len;
	const uint32_t hi = j == UNITS_PER_BLOCK - 1 || (j + 1) * UNIT_BYTES > len ? len : (j + 1) * UNIT_BYTES; // (the last quarter takes what a block of more than 64 KB has left)
	begin = u_off[b] + lo; end = u_off[b] + hi;
}
// stream = [carry bytes | inflated blocks]; block b covers [u_off[b], u_off[b] + u_len); n_blocks below counts UNITS
__global__ __launch_bounds__(BLOCK) void k_find_records(const uint8_t *__restrict__ u, const BgzfBlock *__restrict__ blocks, const uint64_t *__restrict__ u_off, int64_t n_blocks, uint64_t start,
                                                       uint64_t total, int32_t n_targets, const int32_t *__restrict__ tlen, BlockChain *__restrict__ chain, uint16_t *__restrict__ rel)
{
	const int64_t b = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (b >= n_blocks) return;
	// the first block also stands for what lies before it (the carried-over head of a record, or the BAM header): its chain begins at `start`
	uint64_t begin, end;
	unit_range(blocks, u_off, b, begin, end);
	if (b == 0 || begin < start) begin = start < end || b == 0 ? start : end; // nothing starts before `start` (the BAM header may fill several units)
	BlockChain c;
	c.guess = ~0ull; c.exit = begin; c.count = 0; c.listed = 0;
	uint64_t o = begin;
	for (; o < end && o + 36 <= total; ++o) {
		uint64_t q = o;
		int k = 0;
		for (; k < PLAUSIBLE_RUN && q + 36 <= total; ++k) {
			if (!plausible_record(u, q, total, n_targets, tlen)) break;
			q += 4 + (uint64_t)ld_u32(u + q);
		}
		if (k == PLAUSIBLE_RUN || (k > 0 && q + 36 > total)) { c.guess = o; break; }
	}
	if (c.guess != ~0ull) {
		// follow_chain, keeping the record starts
		uint32_t n = 0;
		bool fits = true;
		uint64_t q = c.guess;
		while (q < end) {
			if (q + 4 > total) break;
			const uint32_t bs = ld_u32(u + q);
			if (bs < 32) break; // (corrupt: the stitch walks this block again and reports it)
			if (q + 4 + (uint64_t)bs > total) break;
			if (n < REL_CAP && q - c.guess <= 0xffffull) rel[rel_index(b, n)] = (uint16_t)(q - c.guess); else fits = false;
			++n;
			q += 4 + (uint64_t)bs;
		}
		c.count = n; c.exit = q; c.listed = fits ? 1u : 0u;
	}
	chain[b] = c;
}

struct StitchOut { uint64_t tail; uint32_t n_records, n_repaired, corrupt, pad; };

// One wavefront walks the blocks in order (64 at a time, lane-serial inside the wavefront): `cur` is the true position of the next record
// start.  A block whose guess equals cur keeps its speculated count/exit; a block that cur has already passed holds no record start;
// anything else is repaired by following the chain by hand from cur.  first[b] = first record start in block b (or ~0), count[b] fixed up.
// one step of the stitch: the 64 blocks from `base` on, their guesses in c / end (one block a lane); cur = where the true chain stands (wave-uniform)
__device__ __forceinline__ void stitch_step(const uint8_t *__restrict__ u, int64_t base, int64_t n_blocks, uint64_t total, const BlockChain &c, uint64_t end, BlockChain *__restrict__ chain,
                                            uint32_t *__restrict__ count, uint64_t &cur, uint32_t &n_rec, uint32_t &n_lane, uint32_t &n_rep, uint32_t &bad)
{
	const int64_t b = base + lane_id();
	const int m = (int)(n_blocks - base < WAVE ? n_blocks - base : WAVE);
	// the usual case, checked for all 64 blocks at once: every guess is the exit of the block before it
	// (this wavefront is alone and every step hangs on the one before: the lane crossings are register moves - a DPP wavefront shift, v_readlane - not
	// trips through the LDS crossbar, and the record count is summed per lane, once at the end)
	const uint32_t pe_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)c.exit, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
	const uint32_t pe_hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(c.exit >> 32), 0x138, 0xf, 0xf, false);
	const uint64_t prev_exit = (uint64_t)pe_lo | ((uint64_t)pe_hi << 32);
	const bool ok = b >= n_blocks || (c.guess != ~0ull && c.guess == (lane_id() == 0 ? cur : prev_exit));
	if (__all(ok)) {
		if (b < n_blocks) { count[b] = c.count; n_lane += c.count; } // chain[b].guess already is the first record start
		cur = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)c.exit, m - 1) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(c.exit >> 32), m - 1) << 32);
		return;
	}
	uint64_t my_first = ~0ull;
	uint32_t my_count = 0;
	for (int l = 0; l < m; ++l) {
		const uint64_t g = __shfl(c.guess, l, 64), e = __shfl(c.exit, l, 64), bend = __shfl(end, l, 64);
		const uint32_t cn = __shfl(c.count, l, 64);
		uint64_t first = ~0ull, nxt = cur;
		uint32_t cnt = 0;
		if (cur < bend) {
			if (g == cur) { first = cur; cnt = cn; nxt = e; }
			else {
				// wave-uniform branch: every lane follows the same chain (same addresses: one load per step)
				bool corrupt = false;
				first = cur;
				cnt = follow_chain(u, cur, bend, total, &nxt, &corrupt);
				if (cnt == 0) first = ~0ull;
				if (corrupt) bad = 1;
				++n_rep;
			}
		}
		if (lane_id() == l) { my_first = first; my_count = cnt; }
		n_rec += cnt;
		cur = nxt;
	}
	if (b < n_blocks) {
		chain[b].guess = my_first; count[b] = my_count;
		if (my_first != c.guess || my_count != c.count) chain[b].listed = 0; // not the chain k_find_records walked: its list is of no use
	}
}

// The stitch's usual case needs no walk at all: every block's guess is the exit of the block before it (the first one's is `start`; an empty block - the
// BGZF end-of-file marker - has nothing to guess and is looked through).  One lane per block checks exactly that; only if some block fails (out->pad set) does the one-wavefront
// walk below run, and then it redoes everything.
__global__ __launch_bounds__(BLOCK) void k_stitch_check(const BgzfBlock *__restrict__ blocks, const uint64_t *__restrict__ u_off, int64_t n_blocks, uint64_t start, uint64_t total, const BlockChain *__restrict__ chain,
                                                       uint32_t *__restrict__ count, StitchOut *__restrict__ out)
{
	__shared__ uint32_t s_sum[WAVES_PER_BLOCK];
	const int64_t b = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	bool ok = true;
	uint32_t cnt = 0;
	if (b < n_blocks) {
		const BlockChain c = chain[b];
		// a unit without a guess holds no record start if the chain is past its end when it gets there (empty units, a first unit that is all BAM header, units
		// inside one long record); a unit with a guess must start exactly where the last unit with a guess lets the chain out
		uint64_t begin, end;
		unit_range(blocks, u_off, b, begin, end);
		// (the look back over units without a guess is bounded: a record longer than 64 units - 4 MB - sends the chunk to the walk below instead of
		// making every lane inside it walk to its start, quadratic in the record's length)
		int64_t p = b - 1;
		for (int steps = 0; p >= 0 && chain[p].guess == ~0ull && steps < 64; ++steps) --p;
		const bool capped = p >= 0 && chain[p].guess == ~0ull;
		const uint64_t before = p < 0 ? start : chain[p].exit;
		// (... or too close to the stream's end for a record header: the head of a record the next chunk completes)
		if (capped) ok = false;
		else if (c.guess == ~0ull) { ok = before >= end || before + 36 > total; if (ok) count[b] = 0; if (b == n_blocks - 1) out->tail = before; }
		else {
			ok = c.guess == before;
			cnt = c.count;
			if (ok) count[b] = cnt;
			if (b == n_blocks - 1) out->tail = c.exit;
		}
	}
	if (__any(!ok) && lane_id() == 0) atomicOr(&out->pad, 1u);
	const uint32_t ws = wave_sum(cnt);
	if (lane_id() == 0) s_sum[wave_id()] = ws;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t t = 0;
		for (int w2 = 0; w2 < WAVES_PER_BLOCK; ++w2) t += s_sum[w2];
		if (t) atomicAdd(&out->n_records, t);
	}
}

__global__ __launch_bounds__(WAVE) void k_stitch_blocks(const uint8_t *__restrict__ u, const BgzfBlock *__restrict__ blocks, const uint64_t *__restrict__ u_off, int64_t n_blocks, uint64_t start,
                                                        uint64_t total, BlockChain *__restrict__ chain, uint32_t *__restrict__ count, StitchOut *__restrict__ out)
{
	if (out->pad == 0) return; // k_stitch_check found every guess in place: counts, tail and record total are written
	uint64_t cur = start; // wave-uniform
	uint32_t n_rec = 0, n_lane = 0, n_rep = 0, bad = 0; // records: counted wave-uniformly on the repair path, per lane on the usual one
	// the kernel is one wavefront waiting for memory: the guesses of 8 x 64 blocks are loaded at once, then checked 64 at a time
	for (int64_t base = 0; base < n_blocks; base += WAVE * 8) {
#define SSV_LD(J) BlockChain c##J; uint64_t e##J = 0; { const int64_t b = base + (int64_t)J * WAVE + lane_id(); c##J.guess = ~0ull; c##J.exit = 0; c##J.count = 0; c##J.listed = 0; \
			if (b < n_blocks) { uint64_t bg##J; c##J = chain[b]; unit_range(blocks, u_off, b, bg##J, e##J); } }
		SSV_LD(0) SSV_LD(1) SSV_LD(2) SSV_LD(3) SSV_LD(4) SSV_LD(5) SSV_LD(6) SSV_LD(7)
#undef SSV_LD
		// all eight steps at once when nothing is wrong with any of them: step j's first block must start where step j-1's last block ends, and that is in
		// a register already - no step has to wait for the one before
		bool fine = true;
		uint64_t prev_last = cur;
#define SSV_OK(J) { const int64_t b = base + (int64_t)J * WAVE + lane_id(); \
			const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)c##J.exit, 0x138, 0xf, 0xf, false), hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(c##J.exit >> 32), 0x138, 0xf, 0xf, false); \
			const uint64_t before = lane_id() == 0 ? prev_last : ((uint64_t)lo | ((uint64_t)hi << 32)); \
			fine = fine && (b >= n_blocks || (c##J.guess != ~0ull && c##J.guess == before)); \
			prev_last = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)c##J.exit, 63) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(c##J.exit >> 32), 63) << 32); }
		SSV_OK(0) SSV_OK(1) SSV_OK(2) SSV_OK(3) SSV_OK(4) SSV_OK(5) SSV_OK(6) SSV_OK(7)
#undef SSV_OK
		if (__all(fine)) {
			int64_t last_b = base + (int64_t)WAVE * 8 - 1;
			if (last_b >= n_blocks) last_b = n_blocks - 1; // the batch's last block: lane last_b % 64 of step (last_b - base) / 64
			const int jl = (int)((last_b - base) >> 6), ll = (int)((last_b - base) & 63);
#define SSV_FIN(J) { const int64_t b = base + (int64_t)J * WAVE + lane_id(); if (b < n_blocks) { count[b] = c##J.count; n_lane += c##J.count; } \
			if (jl == J) cur = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)c##J.exit, ll) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(c##J.exit >> 32), ll) << 32); }
			SSV_FIN(0) SSV_FIN(1) SSV_FIN(2) SSV_FIN(3) SSV_FIN(4) SSV_FIN(5) SSV_FIN(6) SSV_FIN(7)
#undef SSV_FIN
			continue;
		}
#define SSV_ST(J) if (base + (int64_t)J * WAVE < n_blocks) stitch_step(u, base + (int64_t)J * WAVE, n_blocks, total, c##J, e##J, chain, count, cur, n_rec, n_lane, n_rep, bad);
		SSV_ST(0) SSV_ST(1) SSV_ST(2) SSV_ST(3) SSV_ST(4) SSV_ST(5) SSV_ST(6) SSV_ST(7)
#undef SSV_ST
	}
	n_rec += wave_sum(n_lane);
	if (lane_id() == 0) { out->tail = cur; out->n_records = n_rec; out->n_repaired = n_rep; out->corrupt = bad; out->pad = 0; }
}

// rec_off[rank] for every record: block b's records start at rank base[b]
__global__ __launch_bounds__(BLOCK) void k_list_records(const uint8_t *__restrict__ u, const BlockChain *__restrict__ chain, const uint32_t *__restrict__ count, const uint32_t *__restrict__ base,
                                                       int64_t n_blocks, const uint16_t *__restrict__ rel, uint64_t *__restrict__ rec_off)
{
	const int64_t b = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (b >= n_blocks) return;
	const BlockChain c = chain[b];
	const uint32_t n = count[b], r0 = base[b];
	if (c.listed) { // independent, coalesced loads: no chain to wait for
		for (uint32_t i = 0; i < n; ++i) rec_off[r0 + i] = c.guess + rel[rel_index(b, i)];
		return;
	}
	uint64_t o = c.guess;
	for (uint32_t i = 0; i < n; ++i) {
		rec_off[r0 + i] = o;
		o += 4 + (uint64_t)ld_u32(u + o);
	}
}

// ---- records -> structure of arrays ----------------------------------------------------------------------------------------------

struct RecColumns {
	int32_t *tid, *pos, *l_qseq, *mtid, *mpos, *isize;
	uint16_t *flag, *n_cigar;
	uint8_t *mapq, *xc;
	uint8_t *ends;        // first | last << 4 CIGAR operation codes (ssv_batch_t.cigar_ends)
	uint32_t *seq_bytes;  // bytes of packed bases + qualities to ship (0 when not shipped)
	uint32_t *raw_bytes;  // 4 + block_size for UNMAP|MUNMAP records (else 0)
	uint2 *stash;         // the first two CIGAR operations (k_record_cigars copies from here: most records have no more, and it need not touch the stream again)
	int32_t *max_span;    // one int: largest reference span (atomicMax)
	uint32_t *bad;        // one flag: a record whose fields overrun its block_size
};

// one record's fixed fields -> columns; returns its reference span.  rp = the record (its block_size word), in the stream or in a staged copy of it
__device__ __forceinline__ int record_fields_of(const uint8_t *rp, int64_t i, int keep_all_seq, const RecColumns &c)
{
	int span = 1;
	const uint8_t *r = rp + 4;
	const uint32_t bs = ld_u32(rp);
	const int32_t l_seq = ld_i32(r + 16);
	const uint32_t l_name = r[8], ncig = ld_u16(r + 12), flag = ld_u16(r + 14);
	c.tid[i] = ld_i32(r); c.pos[i] = ld_i32(r + 4); c.mapq[i] = r[9]; c.n_cigar[i] = (uint16_t)ncig; c.flag[i] = (uint16_t)flag; c.l_qseq[i] = l_seq;
	c.mtid[i] = ld_i32(r + 20); c.mpos[i] = ld_i32(r + 24); c.isize[i] = ld_i32(r + 28);
	const uint64_t o_cig = 32ull + l_name, need = o_cig + 4ull * ncig + ((uint64_t)(l_seq < 0 ? 0 : l_seq) + 1) / 2 + (uint64_t)(l_seq < 0 ? 0 : l_seq);
	bool soft = false;
	uint32_t ends = 0xffu;
	uint2 first2 = make_uint2(0u, 0u);
	if (l_seq < 0 || need > bs) { *c.bad = 1; c.seq_bytes[i] = 0; c.raw_bytes[i] = 0; c.n_cigar[i] = 0; }
	else {
		int s = 0;
		for (uint32_t k = 0; k < ncig; ++k) {
			const uint32_t op = ld_u32(r + o_cig + 4ull * k);
			const uint32_t t = op & 15u;
			if (k == 0) first2.x = op;
			if (k == 1) first2.y = op;
			if (t == 0 || t == 2 || t == 3 || t == 7 || t == 8) s += (int)(op >> 4);
			if ((k == 0 || k == ncig - 1) && t == 4) soft = true;
			if (k == 0) ends = t | (t << 4);
			if (k == ncig - 1) ends = (ends & 15u) | (t << 4);
		}
		if (s > span) span = s;
		c.seq_bytes[i] = (soft || keep_all_seq) ? (uint32_t)(((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq) : 0u;
		c.raw_bytes[i] = (flag & (F_UNMAP | F_MUNMAP)) ? 4u + bs : 0u;
	}
	c.ends[i] = (uint8_t)ends;
	c.stash[i] = first2;
	c.xc[i] = soft ? 2 : 0; // 2 = "soft clipped, aux not looked at yet": k_record_seqs turns it into the XC flag
	return span;
}

// (Tried: a wavefront copies its 64 records' whole run of the stream into LDS with coalesced 16-byte loads and the lanes parse out of the copy - the stream
// read once, in order: 3.5 -> 3.7 ms, with every load issued before the first LDS write 5.0 ms.  Lane-by-lane loads already pull each 128-byte line only
// once - consecutive lanes, consecutive records - so the copy saves no bytes and adds a trip through LDS.)
__global__ __launch_bounds__(BLOCK) void k_record_fields(const uint8_t *__restrict__ u, const uint64_t *__restrict__ rec_off, int64_t n, int keep_all_seq, RecColumns c)
{
	const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	int span = 1;
	if (i < n) span = record_fields_of(u + rec_off[i], i, keep_all_seq, c);
	span = wave_max(span);
	// every wavefront has SOME span > 1, and 300 K atomics on one address are served one after the other (3 of this kernel's 3.5 ms were that): look first
	// (a load that bypasses the CU's cache, or a stale small value would keep this CU's atomics coming), raise only what is not yet as large
	if (lane_id() == 0 && span > 1 && span > __hip_atomic_load(c.max_span, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(c.max_span, span);
}

// bam_aux_get(b, "XC") + bam_aux2i (clip_reads.cpp:126-127): integer value of the XC tag != 0
__device__ __forceinline__ int aux_xc_flag(const uint8_t *p, const uint8_t *end)
{
	while (p + 3 <= end) {
		const uint8_t t0 = p[0], t1 = p[1], ty = p[2];
		p += 3;
		const bool hit = t0 == 'X' && t1 == 'C';
		int64_t v = 0;
		uint32_t sz;
		switch (ty) {
		case 'A': case 'c': sz = 1; if (hit && ty == 'c') v = (int8_t)p[0]; break;
		case 'C': sz = 1; if (hit) v = p[0]; break;
		case 's': sz = 2; if (hit) v = (int16_t)ld_u16(p); break;
		case 'S': sz = 2; if (hit) v = ld_u16(p); break;
		case 'i': sz = 4; if (hit) v = ld_i32(p); break;
		case 'I': sz = 4; if (hit) v = ld_u32(p); break;
		case 'f': sz = 4; break;
		case 'd': sz = 8; break;
		case 'Z': case 'H': { const uint8_t *q = p; while (q < end && *q) ++q; sz = (uint32_t)(q - p) + 1; break; }
		case 'B': {
			if (p + 5 > end) return 0;
			const uint8_t st = p[0];
			const uint32_t cnt = ld_u32(p + 1);
			const uint32_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
			sz = 5 + cnt * es;
			break;
		}
		default: return 0; // unknown type: libbam stops here too
		}
		if (p + sz > end) return 0;
		if (hit) return v != 0;
		p += sz;
	}
	return 0;
}

// 16 lanes per record: CIGAR ops and (for the records that ship them) packed bases + qualities, byte for byte as they lie in the record
// CIGARs to their places (exclusive scan of n_cigar), one lane per record: out of the stash when the record has at most two operations, out of the stream
// otherwise.  Records that ship bases + qualities (or still owe their XC flag) go on a list for k_record_seqs - about one in a hundred.
constexpr int CIGARS_PER_THREAD = 16; // a workgroup takes BLOCK x 16 consecutive records: ONE atomic on the list's counter per 4096 records
__global__ __launch_bounds__(BLOCK) void k_record_cigars(const uint8_t *__restrict__ u, const uint64_t *__restrict__ rec_off, int64_t n, const uint16_t *__restrict__ n_cigar,
                                                        const uint32_t *__restrict__ cigar_off, const uint2 *__restrict__ stash, const uint32_t *__restrict__ seq_bytes,
                                                        const uint8_t *__restrict__ xc, uint64_t *__restrict__ seq_off, uint32_t *__restrict__ cigar, uint32_t *__restrict__ list,
                                                        uint32_t *__restrict__ n_list)
{
	__shared__ uint32_t mine[BLOCK * CIGARS_PER_THREAD];
	__shared__ uint32_t n_mine, base;
	if (threadIdx.x == 0) n_mine = 0;
	__syncthreads();
	const int64_t i0 = (int64_t)blockIdx.x * (BLOCK * CIGARS_PER_THREAD);
	for (int t = 0; t < CIGARS_PER_THREAD; ++t) {
		const int64_t i = i0 + (int64_t)t * BLOCK + threadIdx.x;
		bool has = false;
		if (i < n) {
			const uint32_t ncig = n_cigar[i], co = cigar_off[i];
			if (ncig <= 2u) {
				const uint2 f = stash[i];
				if (ncig > 0u) cigar[co] = f.x;
				if (ncig > 1u) cigar[co + 1] = f.y;
			} else {
				const uint8_t *r = u + rec_off[i] + 4;
				const uint64_t o_cig = 32ull + r[8];
				for (uint32_t k = 0; k < ncig; ++k) cigar[co + k] = ld_u32(r + o_cig + 4ull * k);
			}
			const bool ships = seq_bytes[i] != 0u;
			if (!ships) seq_off[i] = ~0ull; // SSV_NO_SEQ
			has = ships || xc[i] == 2;
		}
		const uint64_t m = __ballot(has);
		if (m) {
			const int leader = __ffsll((long long)m) - 1;
			uint32_t at = 0;
			if (lane_id() == leader) at = atomicAdd(&n_mine, (uint32_t)__popcll(m)); // (LDS)
			at = (uint32_t)__shfl((int)at, leader);
			if (has) mine[at + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull))] = (uint32_t)i;
		}
	}
	__syncthreads();
	if (threadIdx.x == 0 && n_mine) base = atomicAdd(n_list, n_mine);
	__syncthreads();
	for (uint32_t k = threadIdx.x; k < n_mine; k += BLOCK) list[base + k] = mine[k];
}

// packed bases + qualities of the listed records (16 lanes a record), and their XC aux flag
__global__ __launch_bounds__(BLOCK) void k_record_seqs(const uint8_t *__restrict__ u, const uint64_t *__restrict__ rec_off, const uint16_t *__restrict__ n_cigar,
                                                      const uint32_t *__restrict__ seq_bytes, const uint64_t *__restrict__ seq_off, const uint32_t *__restrict__ list,
                                                      const uint32_t *__restrict__ n_list, uint8_t *__restrict__ seqqual, uint8_t *__restrict__ xc)
{
	const uint32_t total = *n_list;
	const int sub = (int)(threadIdx.x & 15);
	for (uint32_t e = (blockIdx.x * BLOCK + threadIdx.x) >> 4; e < total; e += gridDim.x * (BLOCK / 16)) {
		const uint32_t i = list[e];
		const uint8_t *r = u + rec_off[i] + 4;
		const uint32_t bs = ld_u32(r - 4), l_name = r[8], ncig = n_cigar[i];
		const uint64_t o_seq = 32ull + l_name + 4ull * ncig;
		const uint32_t sb = seq_bytes[i];
		if (sb) {
			const uint64_t so = seq_off[i]; // exclusive scan of seq_bytes
			for (uint32_t k = (uint32_t)sub; k < sb; k += 16) seqqual[so + k] = r[o_seq + k];
		}
		if (sub == 0 && xc[i] == 2) {
			const int32_t l_seq = ld_i32(r + 16);
			xc[i] = (uint8_t)aux_xc_flag(r + o_seq + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq, r + bs);
		}
	}
}

__global__ __launch_bounds__(BLOCK) void k_raw_copy(const uint8_t *__restrict__ u, const uint64_t *__restrict__ rec_off, int64_t n, const uint32_t *__restrict__ raw_bytes,
                                                   const uint64_t *__restrict__ raw_off, uint8_t *__restrict__ raw)
{
	// a lane per record looks (such records are few: most wavefronts leave after one coalesced load); the wavefront then copies each one it found together
	const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	const uint32_t nb = i < n ? raw_bytes[i] : 0u;
	uint64_t m = __ballot(nb != 0u);
	while (m) {
		const int l = __ffsll((long long)m) - 1;
		m &= m - 1;
		const uint32_t knb = (uint32_t)__shfl((int)nb, l);
		const int64_t ki = i - lane_id() + l;
		const uint8_t *r = u + rec_off[ki];
		const uint64_t o = raw_off[ki];
		for (uint32_t k = (uint32_t)lane_id(); k < knb; k += WAVE) raw[o + k] = r[k];
	}
}

// The flush sequence of getclip's record loop (clip_reads.h:423-438) needs, in order, every change of contig among the records that
// are not UNMAP|MUNMAP ("qualifying").  Three small kernels: the last qualifying tid of every 256-record tile; one wavefront carries it
// across tiles; then every qualifying record compares its tid with the qualifying record before it (ballot + shuffle inside the
// wavefront, LDS across the workgroup's wavefronts, the carried value across tiles) and reports (index, tid) when it differs.
struct TidRun { uint32_t index; int32_t tid; };
constexpr int32_t TID_NONE = INT32_MIN;

// the tid column as runs (ssv_batch_t.tid_runs): the records where the contig changes, record 0 included; a handful in a sorted file
// (*count may exceed cap: then the list is incomplete and is not handed out)
__global__ __launch_bounds__(BLOCK) void k_tid_raw_runs(const int32_t *__restrict__ tid, int64_t n, TidRun *__restrict__ runs, uint32_t cap, uint32_t *__restrict__ count)
{
	const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	const int32_t t = tid[i];
	if (i == 0 || tid[i - 1] != t) {
		const uint32_t k = atomicAdd(count, 1u);
		if (k < cap) { runs[k].index = (uint32_t)i; runs[k].tid = t; }
	}
}


__global__ __launch_bounds__(BLOCK) void k_tid_tile_last(const int32_t *__restrict__ tid, const uint16_t *__restrict__ flag, int64_t n, int32_t *__restrict__ tile_last)
{
	__shared__ int lds[WAVES_PER_BLOCK];
	const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	const bool q = i < n && !(flag[i] & (F_UNMAP | F_MUNMAP));
	int idx = wave_max(q ? (int)threadIdx.x : -1);
	if (lane_id() == 0) lds[wave_id()] = idx;
	__syncthreads();
	if (threadIdx.x == 0) {
		int m = -1;
		for (int w = 0; w < WAVES_PER_BLOCK; ++w) m = lds[w] > m ? lds[w] : m;
		tile_last[blockIdx.x] = m >= 0 ? tid[(int64_t)blockIdx.x * BLOCK + m] : TID_NONE;
	}
}

__global__ __launch_bounds__(WAVE) void k_tid_tile_carry(const int32_t *__restrict__ tile_last, int64_t n_tiles, int32_t prev_tid, int32_t *__restrict__ tile_prev, int32_t *__restrict__ last_tid)
{
	int32_t cur = prev_tid; // wave-uniform
	constexpr int AHEAD = 16; // (one wavefront waiting for memory: sixteen steps' values are loaded at once)
	for (int64_t base0 = 0; base0 < n_tiles; base0 += WAVE * AHEAD) {
	int32_t vals[AHEAD];
#pragma unroll
	for (int j = 0; j < AHEAD; ++j) { const int64_t t = base0 + (int64_t)j * WAVE + lane_id(); vals[j] = t < n_tiles ? tile_last[t] : TID_NONE; }
#pragma unroll
	for (int j = 0; j < AHEAD; ++j) {
		const int64_t t = base0 + (int64_t)j * WAVE + lane_id(); // (steps past the end see only TID_NONE: they change nothing)
		const int32_t mine = vals[j];
		// exclusive "last value that is not NONE" over the 64 lanes
		const uint64_t have = __ballot(mine != TID_NONE);
		const uint64_t below = have & lanemask_lt();
		const int src = below ? 63 - __clzll((long long)below) : 0;
		const int32_t got = __shfl(mine, src, 64);
		if (t < n_tiles) tile_prev[t] = below ? got : cur;
		if (have) cur = __shfl(mine, 63 - __clzll((long long)have), 64);
	}
	}
	if (lane_id() == 0) *last_tid = cur;
}

__global__ __launch_bounds__(BLOCK) void k_tid_runs(const int32_t *__restrict__ tid, const uint16_t *__restrict__ flag, int64_t n, const int32_t *__restrict__ tile_prev,
                                                   TidRun *__restrict__ runs, uint32_t cap, uint32_t *__restrict__ n_runs)
{
	__shared__ int32_t wave_last[WAVES_PER_BLOCK];
	const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	const bool q = i < n && !(flag[i] & (F_UNMAP | F_MUNMAP));
	const int32_t mine = q ? tid[i] : TID_NONE;
	const uint64_t have = __ballot(q);
	const int32_t top = __shfl(mine, have ? 63 - __clzll((long long)have) : 0, 64); // all lanes take part in the shuffle
	if (lane_id() == 0) wave_last[wave_id()] = have ? top : TID_NONE;
	const uint64_t below = have & lanemask_lt();
	int32_t before = __shfl(mine, below ? 63 - __clzll((long long)below) : 0, 64);
	__syncthreads();
	if (!below) {
		before = tile_prev[blockIdx.x];
		for (int w = 0; w < wave_id(); ++w) if (wave_last[w] != TID_NONE) before = wave_last[w];
	}
	if (q && mine != before) {
		const uint32_t k = atomicAdd(n_runs, 1u);
		if (k < cap) { runs[k].index = (uint32_t)i; runs[k].tid = mine; }
	}
}

} // namespace ssv
