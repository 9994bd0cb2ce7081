/* synth.h - C API of the synthetic BAM-record generator (bench / test tooling).  The same entry points exist in
 * libseeksv_synth.so (HIP: every pointer is device memory of the current device) and libseeksv_synth_cpu.so
 * (plain C++: every pointer is host memory).  Both produce identical bytes (csrc/synth_core.h). */
#ifndef SEEKSV_SYNTH_H_
#define SEEKSV_SYNTH_H_
#include <stdint.h>
#include "synth_core.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Phase 1: for records [g0, g0+n) write n_cigar[n], cigar_off[n], seq_off[n] (SSV_NO_SEQ where no bases are shipped) and
 * return the sizes the caller must allocate for cigar[] and seqqual[]. */
int ssvs_plan(const sy_config *cfg, const sy_breakend *be, int64_t g0, int64_t n, uint16_t *n_cigar, uint32_t *cigar_off, uint64_t *seq_off,
              int64_t *n_cigar_total, int64_t *seqqual_bytes);
/* Phase 2: fill every remaining array of the batch.  rec (optional): the records' 64-byte lines (ssv_record, include/seeksv_hip.h);
 * when it is given, the cold structure-of-arrays columns flag / mapq / l_qseq / mtid / mpos / isize may be NULL.  cigar_ends (optional): the hot
 * copy of the first and last CIGAR operation codes (ssv_batch_t.cigar_ends). */
int ssvs_fill(const sy_config *cfg, const sy_breakend *be, int64_t g0, int64_t n, int32_t *tid, int32_t *pos, uint16_t *flag, uint8_t *mapq,
              const uint16_t *n_cigar, int32_t *l_qseq, int32_t *mtid, int32_t *mpos, int32_t *isize, const uint32_t *cigar_off, uint32_t *cigar,
              const uint64_t *seq_off, uint8_t *seqqual, void *rec, uint8_t *cigar_ends);
const char *ssvs_last_error(void);
#ifdef __cplusplus
}
#endif
#endif
