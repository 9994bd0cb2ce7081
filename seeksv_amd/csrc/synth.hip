// synth.hip - GPU build of the synthetic generator: fills a device-resident SoA batch for bench.py so that the
// timed region starts with its inputs in HBM.  Tooling, not the product path.
#include <hip/hip_runtime.h>

#include <string>

#include "scan.h"
#include "synth.h"

using namespace ssv;

static std::string g_err;

#define SY_CHECK(call)                                                         \
	do {                                                                         \
		hipError_t e_ = (call);                                                    \
		if (e_ != hipSuccess) { g_err = std::string(#call) + ": " + hipGetErrorString(e_); return -1; } \
	} while (0)

__global__ void k_sy_count(sy_config cfg, const sy_breakend *be, int64_t g0, int64_t n, uint16_t *n_cigar, uint32_t *seq_bytes)
{
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	sy_record r;
	sy_decide(&cfg, be, g0 + i, &r);
	n_cigar[i] = r.n_cigar;
	seq_bytes[i] = r.has_seq ? (uint32_t)((r.l_qseq + 1) / 2 + r.l_qseq) : 0u;
}

__global__ void k_sy_fix(int64_t n, const uint32_t *seq_bytes, uint64_t *seq_off)
{
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && seq_bytes[i] == 0) seq_off[i] = UINT64_MAX;
}

__global__ void k_sy_fill(sy_config cfg, const sy_breakend *be, int64_t g0, int64_t n, int32_t *tid, int32_t *pos, uint16_t *flag, uint8_t *mapq, int32_t *l_qseq,
                          int32_t *mtid, int32_t *mpos, int32_t *isize, const uint32_t *cigar_off, uint32_t *cigar, const uint64_t *seq_off, uint8_t *seqqual, uint4 *rec, uint8_t *ends)
{
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	sy_record r;
	sy_decide(&cfg, be, g0 + i, &r);
	tid[i] = r.tid; pos[i] = r.pos;
	if (flag) flag[i] = r.flag;
	if (mapq) mapq[i] = r.mapq;
	if (l_qseq) l_qseq[i] = r.l_qseq;
	if (mtid) mtid[i] = r.mtid;
	if (mpos) mpos[i] = r.mpos;
	if (isize) isize[i] = r.isize;
	for (int k = 0; k < r.n_cigar; ++k) cigar[cigar_off[i] + k] = r.cigar[k];
	if (ends) ends[i] = r.n_cigar ? (uint8_t)((r.cigar[0] & 15u) | ((r.cigar[r.n_cigar - 1] & 15u) << 4)) : (uint8_t)0xff;
	if (r.has_seq) sy_fill_seq(&cfg, be, g0 + i, &r, seqqual + seq_off[i]);
	if (rec) {
		uint32_t w[16];
		sy_fill_line(&r, cigar_off[i], seq_off[i], w);
		rec[4 * i] = make_uint4(w[0], w[1], w[2], w[3]); rec[4 * i + 1] = make_uint4(w[4], w[5], w[6], w[7]);
		rec[4 * i + 2] = make_uint4(w[8], w[9], w[10], w[11]); rec[4 * i + 3] = make_uint4(w[12], w[13], w[14], w[15]);
	}
}

__global__ void k_sy_ref2bit(sy_config cfg, uint64_t *out, int64_t n_words)
{
	int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w < n_words) out[w] = sy_ref_word(&cfg, w);
}

extern "C" {

// the whole reference as 2-bit words in device memory (ssv_realign_index layout, SSV_MEM_DEVICE); out has (total + 31) / 32 + 1 words
int ssvs_ref_2bit(const sy_config *cfg, uint64_t *out, int64_t n_words)
{
	if (n_words <= 0) return 0;
	k_sy_ref2bit<<<(unsigned)((n_words + 255) / 256), 256>>>(*cfg, out, n_words);
	SY_CHECK(hipGetLastError());
	SY_CHECK(hipDeviceSynchronize());
	return 0;
}

const char *ssvs_last_error(void) { return g_err.c_str(); }

int ssvs_plan(const sy_config *cfg, const sy_breakend *be, int64_t g0, int64_t n, uint16_t *n_cigar, uint32_t *cigar_off, uint64_t *seq_off,
              int64_t *n_cigar_total, int64_t *seqqual_bytes)
{
	if (n <= 0) { *n_cigar_total = 0; *seqqual_bytes = 0; return 0; }
	hipStream_t st = nullptr;
	uint32_t *seq_bytes = nullptr, *scr32 = nullptr;
	uint64_t *scr64 = nullptr, *tot = nullptr;
	SY_CHECK(hipMalloc(&seq_bytes, (size_t)n * 4));
	SY_CHECK(hipMalloc(&scr32, (size_t)scan_scratch_elems(n) * 4));
	SY_CHECK(hipMalloc(&scr64, (size_t)scan_scratch_elems(n) * 8));
	SY_CHECK(hipMalloc(&tot, 16));
	unsigned grid = (unsigned)((n + 255) / 256);
	k_sy_count<<<grid, 256, 0, st>>>(*cfg, be, g0, n, n_cigar, seq_bytes);
	exclusive_scan<uint16_t, uint32_t>(st, n_cigar, cigar_off, n, 0u, scr32, reinterpret_cast<uint32_t *>(tot));
	exclusive_scan<uint32_t, uint64_t>(st, seq_bytes, seq_off, n, 0ull, scr64, tot + 1);
	k_sy_fix<<<grid, 256, 0, st>>>(n, seq_bytes, seq_off);
	SY_CHECK(hipGetLastError());
	uint64_t h[2];
	SY_CHECK(hipMemcpy(h, tot, 16, hipMemcpyDeviceToHost));
	*n_cigar_total = (int64_t)(uint32_t)h[0]; *seqqual_bytes = (int64_t)h[1];
	SY_CHECK(hipFree(seq_bytes)); SY_CHECK(hipFree(scr32)); SY_CHECK(hipFree(scr64)); SY_CHECK(hipFree(tot));
	return 0;
}

int ssvs_fill(const sy_config *cfg, const sy_breakend *be, int64_t g0, int64_t n, int32_t *tid, int32_t *pos, uint16_t *flag, uint8_t *mapq,
              const uint16_t *n_cigar, int32_t *l_qseq, int32_t *mtid, int32_t *mpos, int32_t *isize, const uint32_t *cigar_off, uint32_t *cigar,
              const uint64_t *seq_off, uint8_t *seqqual, void *rec, uint8_t *cigar_ends)
{
	(void)n_cigar;
	if (n <= 0) return 0;
	k_sy_fill<<<(unsigned)((n + 255) / 256), 256, 0, nullptr>>>(*cfg, be, g0, n, tid, pos, flag, mapq, l_qseq, mtid, mpos, isize, cigar_off, cigar, seq_off, seqqual, reinterpret_cast<uint4 *>(rec), cigar_ends);
	SY_CHECK(hipGetLastError());
	SY_CHECK(hipDeviceSynchronize());
	return 0;
}

} // extern "C"
