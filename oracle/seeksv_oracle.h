/*
 * seeksv_oracle.h - TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded CPU restatement of the record-level arithmetic of seeksv v1.2.3's hot
 * path, written from the reference's behaviour (file:line cited at each function in the .c).
 * It is the checker for the HIP path and the "port" CPU baseline of bench.py.  Nothing in the
 * product (seeksv_amd/, include/) may call, link or import it; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg do.
 *
 * Parity status: PINNED - tests/test_oracle_golden.py checks this restatement against outputs of the
 * reference itself (oracle/_ref/seeksv_ref, built from /root/reference by oracle/Makefile) committed
 * under tests/golden/: clip tables of both example BAMs, SV tables, and -B junction-injection runs
 * on crafted and synthetic BAMs - including '=' / 'X' CIGAR operations in the depth pass (eqx.*) and the
 * read cap of the libbam 0.1.16 pileup (deep.*); tests/test_random_oracle_vs_reference.py runs the
 * reference binary itself on generated samples.  Not pinned (stated in DESIGN.md): records the
 * reference treats with undefined behaviour (n_cigar == 0, missing qualities inside a multi-read bin,
 * CIGAR shapes on which the 0.1.16 pileup asserts).
 *
 * It shares the batch / junction / interval struct definitions with include/seeksv_hip.h so that both
 * sides are fed byte-identical inputs.
 */
#ifndef SEEKSV_ORACLE_H_
#define SEEKSV_ORACLE_H_

#include "../include/seeksv_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* same field meaning as ssv_cluster_table, but every array is malloc'd and owned by the table */
typedef struct {
	int64_t n_clusters;
	int64_t n_events;
	int32_t *tid;
	int32_t *pos;
	uint8_t *side;
	int32_t *support;
	int32_t *left_len;
	int32_t *right_len;
	uint8_t *qual_missing;
	uint64_t *str_off;
	uint8_t *str;
	uint64_t *cigar_off;
	int32_t *n_cigar;
	uint32_t *cigar;
	int64_t str_bytes;
	int64_t cigar_ops;
} orc_cluster_table;

/* getclip over a list of batches (file order); all batches SSV_MEM_HOST */
int orc_getclip(const ssv_batch_t *batches, int n_batches, const ssv_clip_params *p, orc_cluster_table *out);
void orc_cluster_table_free(orc_cluster_table *t);

/* CalculateInsertsizeDeviation; returns 1 (and leaves mean/sd) when no pair qualifies, like the reference */
int orc_isize_stats(const ssv_batch_t *batches, int n_batches, int32_t min_mapq, int64_t max_pairs,
                    int64_t *n_pairs, int32_t *mean, int32_t *sd);

/* FindDiscordantReadPairs: counts[j] for every junction */
int orc_discordant(const ssv_batch_t *batches, int n_batches, const ssv_junction *junctions, int64_t n_junctions,
                   int32_t mean, int32_t sd, int32_t times, int32_t min_mapq, int32_t *counts);

/* main_depth: per-column depth restricted to windows, then range sums and point depths */
int orc_depth(const ssv_batch_t *batches, int n_batches, const ssv_interval *windows, int64_t n_windows,
              int32_t min_mapq, const ssv_interval *ranges, int64_t n_ranges, uint64_t *range_sum,
              const ssv_interval *points, int64_t n_points, int32_t *point_depth, int32_t *max_depth);

#ifdef __cplusplus
}
#endif
#endif
