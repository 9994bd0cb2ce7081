/*
 * seeksv_host.h - C ABI of libseeksv_host.so: the host side of the hot path, i.e. what replaces
 * the reference's use of samtools-0.1.16 libbam *below* the record loops (samopen/samread,
 * sam/sam.h:59,73) and the small ordered-container bookkeeping *around* the BAM passes of getsv
 * (GetBreak / MergeOverlap, getsv.cpp:752-835; the window arithmetic of FindDiscordantReadPairs,
 * getsv.cpp:1041-1060; the lookup rules of main_depth, bam2depth.cpp:82-124).
 * No GPU code here; it produces the structure-of-arrays batches that include/seeksv_hip.h consumes.
 */
#ifndef SEEKSV_HOST_H_
#define SEEKSV_HOST_H_

#include <stddef.h>
#include <stdint.h>
#include "seeksv_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ssvh_bam ssvh_bam; /* an open BAM file */

/* Open a BAM file and parse its header.  Returns 0, or <0 and a message in ssvh_last_error(). */
int ssvh_bam_open(const char *path, ssvh_bam **out);
/* Header-only handle (no file): contig names and lengths for ssvh_plan_create when the records do not come from a BAM file. */
int ssvh_bam_from_header(const char *const *names, const int32_t *lens, int32_t n, ssvh_bam **out);
void ssvh_bam_close(ssvh_bam *b);
const char *ssvh_last_error(void);

int32_t ssvh_bam_n_targets(const ssvh_bam *b);
const char *ssvh_bam_target_name(const ssvh_bam *b, int32_t tid);
int32_t ssvh_bam_target_len(const ssvh_bam *b, int32_t tid);
const int32_t *ssvh_bam_target_lens(const ssvh_bam *b);

/*
 * Decode up to max_records records (file order) into a structure-of-arrays batch (SSV_MEM_HOST).
 * The arrays are owned by the reader and stay valid until the next call on the same handle.
 * Bases + qualities are shipped only for records whose first or last CIGAR op is 'S'.
 * out->n == 0 at end of file.  keep_all_seq != 0 ships bases/qualities for every record.
 */
int ssvh_bam_read_batch(ssvh_bam *b, int64_t max_records, int keep_all_seq, ssv_batch_t *out);

/*
 * Range-partitioned reading (the reference reaches a region through the .bai: seeksv.cpp:272-280 bam_index_load,
 * getsv.cpp:1063-1067 bam_iter_query / bam_iter_read; here the FILE is cut into N contiguous runs of records, one per GPU, and no index
 * is needed).  Positions are BGZF virtual offsets given as (file offset of the block, offset inside its inflated bytes).
 *   ssvh_bam_partition   n_parts runs balanced by inflated bytes; every boundary is the exact start of a record (found by speculation inside
 *                        the block and verified by following the block_size chain over >= 8 records).  For the getclip pass a part also
 *                        knows where its HALO begins: the first record that starts within halo_bp before the part's first record on
 *                        the same contig - a right-clipped read's breakpoint lies at its start + reference span, so such a record can own
 *                        a breakpoint inside the part (halo_bp >= the longest reference span of a read) - and the contig of the last
 *                        mapped-pair record before the halo (the contig-switch rule of clip_reads.h:423-438 needs it).
 *   ssvh_bam_walk_back   the record n_back records before a position (the first record of the file if there are fewer): where the replay
 *                        of the pileup's read cap starts (bam2depth.cpp:72-75, libbam's bam_plp_push).
 *   ssvh_bam_set_range   ssvh_bam_read_batch / ssvh_bam_unmapped_* then yield exactly the records that start in [start, end)
 *                        (end_coff = UINT64_MAX: to the end of the file).
 */
typedef struct {
	uint64_t scan_coff; uint32_t scan_uoff;  /* getclip: first record to scan (halo start; = own for part 0) */
	uint64_t own_coff; uint32_t own_uoff;    /* first record of the part */
	uint64_t end_coff; uint32_t end_uoff;    /* first record of the next part (UINT64_MAX, 0 for the last part) */
	int32_t own_tid, own_pos;                /* contig and 0-based position of the part's first record (own_tid = n_targets when it is unplaced or the part is empty) */
	int32_t initial_last_tid;                /* contig of the last record without UNMAP|MUNMAP before the scan start (0 at the start of the file) */
	int32_t before_own_tid;                  /* the same before the part's first record */
	int64_t halo_records;                    /* records in [scan, own) */
} ssvh_bam_part;
int ssvh_bam_partition(const char *path, int32_t n_parts, int32_t halo_bp, ssvh_bam_part *parts);
int ssvh_bam_walk_back(const char *path, uint64_t coff, uint32_t uoff, int64_t n_back, uint64_t *out_coff, uint32_t *out_uoff, int64_t *n_found);
int ssvh_bam_set_range(ssvh_bam *b, uint64_t start_coff, uint32_t start_uoff, uint64_t end_coff, uint32_t end_uoff);
/* Raw mode (below) over the records of [start, end): read_blocks starts at the block at start_coff and stops behind the block that holds the
 * range's last record; *first_record_offset = where the first record begins in the first block's inflated bytes.  After every
 * ssvh_bam_read_blocks call ssvh_bam_raw_limit tells how many inflated bytes of that call's blocks belong to the range (UINT64_MAX: all):
 * pass it to ssv_bamdec_limit before decoding the chunk. */
int ssvh_bam_raw_begin_range(ssvh_bam *b, uint64_t start_coff, uint32_t start_uoff, uint64_t end_coff, uint32_t end_uoff, uint64_t *first_record_offset);
int ssvh_bam_raw_limit(const ssvh_bam *b, uint64_t *inflated_bytes);
const char *ssvh_partition_last_error(void); /* message of the last failed ssvh_bam_partition / ssvh_bam_walk_back of this thread */
/* 1 when this thread's last ssvh_bam_partition took its boundaries from the file's .bai (path + ".bai", or .bam replaced by .bai: the index
 * the reference loads, seeksv.cpp:272-280) - the record starts of its linear index -, 0 when it found them by speculation in the file itself. */
int ssvh_partition_used_index(void);

/* on != 0: after handing out a batch, ssvh_bam_read_batch decodes the following one on a background thread into another set
 * of arrays, so that inflate + decode overlap whatever the caller does with the current batch (upload, kernels, output); there are
 * three sets: a batch stays valid until the SECOND read_batch call after the one that returned it (its copy to the GPU may still be
 * running while the next batch is handed out).  While it is on, every read_batch call must pass the same max_records / keep_all_seq, and ssvh_bam_next_record is refused. */
int ssvh_bam_set_readahead(ssvh_bam *b, int on);
/* Where the batch arrays of ssvh_bam_read_batch live: pass the device library's page-locked allocator (ssv_host_alloc / ssv_host_free
 * behind two plain functions) so that the copies to the GPU run asynchronously; NULL, NULL = malloc / free (the default). */
int ssvh_bam_set_allocator(ssvh_bam *b, void *(*alloc)(size_t), void (*release)(void *));

/* Raw mode, for the device-side decoder (ssv_bamdec_*, seeksv_hip.h): rewind to the first BGZF block and hand out COMPRESSED blocks.
 * *first_record_offset = length of the BAM header inside the inflated stream.  Not to be mixed with read_batch / next_record. */
int ssvh_bam_raw_begin(ssvh_bam *b, uint64_t *first_record_offset);
/* Read whole BGZF blocks: the file's bytes as they are (block headers and trailers included; all host threads read at once) into dst
 * (dst_bytes; 8 spare bytes are kept behind the last block), at most max_blocks non-empty blocks and about max_inflated bytes of inflated
 * data; blocks[k].c_off / c_len = where block k's deflate payload lies in dst, u_len = what it inflates to; *n_bytes = bytes of dst in use.
 * *n_blocks == 0 at end of file. */
int ssvh_bam_read_blocks(ssvh_bam *b, void *dst, size_t dst_bytes, uint64_t max_inflated, ssv_bgzf_block *blocks, int64_t max_blocks, int64_t *n_blocks, size_t *n_bytes);
/* The same chunking without the copy: *ptr = the chunk's bytes (at most max_bytes) where they lie in a read-only mapping of the file - the page
 * cache -, valid until the handle is closed; blocks[k].c_off counts from *ptr.  The caller page-locks the chunk's pages (ssv_host_register,
 * seeksv_hip.h) and hands *ptr to ssv_bamdec_prefetch / ssv_bamdec_decode: the GPU's DMA engines fetch the file out of the page cache and no CPU
 * copies it (libbam's reader: read(2) into a buffer, inflate from there, one block at a time - sam/bgzf.c).  Consecutive chunks come out of three
 * mappings in turn, so the page ranges of up to three chunks in flight never overlap within one mapping.  Returns -2 when the file cannot be
 * mapped (the caller then uses ssvh_bam_read_blocks). */
int ssvh_bam_map_blocks(ssvh_bam *b, size_t max_bytes, uint64_t max_inflated, ssv_bgzf_block *blocks, int64_t max_blocks, int64_t *n_blocks, const void **ptr, size_t *n_bytes);
/* Decode the k-th record of a run of raw BAM records (block_size prefixed, as ssv_bamdec_info.unmapped_raw holds them) the way the
 * unmapped side channel wants it (GetSeqAndQual, clip_reads.cpp:375-388).  Returns the offset of the next record, 0 at the end. */
size_t ssvh_raw_record_fastq(const uint8_t *raw, size_t raw_bytes, size_t offset, const char **qname, const char **seq, const char **qual, int *is_read1);

/* One record at a time, with its read name and CIGAR (host-side consumers of small BAMs: the clip.bam join of getsv,
 * getsv.h:445-527).  Returns 1 and fills *out (pointers valid until the next call), 0 at end of file, <0 on error. */
typedef struct {
	int32_t tid, pos, l_qseq;
	uint16_t flag, n_cigar;
	uint8_t mapq;
	const char *qname;
	const uint32_t *cigar;
	const uint8_t *seq;   /* packed 4-bit bases */
	const uint8_t *qual;
} ssvh_record;
int ssvh_bam_next_record(ssvh_bam *b, ssvh_record *out);

/* Tooling: write a structure-of-arrays batch (SSV_MEM_HOST) as a BAM file (BGZF, deflate level 1 - SSV_BGZF_LEVEL in the environment picks another -,
 * records serialised and blocks compressed by all host threads).
 * Records without shipped bases get l_qseq 'A's with quality 30; read names are "<prefix><index>".  append != 0 continues a file
 * started by an earlier call (the header is written only when append == 0); finish != 0 writes the BGZF end-of-file block. */
int ssvh_bam_write_batch(const char *path, const char *const *names, const int32_t *lens, int32_t n_targets, const ssv_batch_t *b,
                         const char *qname_prefix, int64_t first_index, int append, int finish);
/* The same with explicit read names (qnames[b->n], each <= 254 characters): the clip.bam of `seeksv realign`, whose read names are
 * the clipped sequences like in the output of `bwa mem prefix.clip.fq.gz` (getsv.h:482-485 joins on them). */
int ssvh_bam_write_batch_named(const char *path, const char *const *names, const int32_t *lens, int32_t n_targets, const ssv_batch_t *b,
                               const char *const *qnames, int append, int finish);

/* Append text to a .gz file as independent gzip members compressed in parallel (256 KiB of text each).  By default a member is one
 * literal-only dynamic-Huffman block (seeksv_amd/host/huff_gz.h: the rows of getclip are text with a small alphabet and few repeats - 2.2-2.3 x
 * at several hundred MB/s per core, where zlib level 1 gives 2.7 x at 50-60 MB/s and level 6 - gzstream's default, the reference's - 3.2 x at
 * 10); SSV_GZ_LEVEL=0..9 selects zlib at that level.  The same bytes after decompression in every case.  Concatenated members are a valid gzip stream: zlib's gzread (igzstream, bwa, zcat) decompresses them to exactly the bytes
 * written.  append == 0 truncates the file first.  n == 0 with append == 0 creates an empty gzip stream like an ogzstream that is
 * closed without writes. */
int ssvh_gz_append(const char *path, const char *text, size_t n, int append);
/* the same for several buffers that follow one another in the output (one call keeps every host thread busy) */
int ssvh_gz_append_v(const char *path, const char *const *texts, const size_t *lens, int count, int append);

/* Records with UNMAP|MUNMAP seen in the last batch: qname / decoded bases / qualities for the
 * unmapped-pair FASTQ side channel (clip_reads.h:415-419).  Index k in [0, n).  Host only. */
int64_t ssvh_bam_unmapped_count(const ssvh_bam *b);
int ssvh_bam_unmapped_get(const ssvh_bam *b, int64_t k, const char **qname, const char **seq, const char **qual, int *is_read1);
/* The same records as they lie in the BAM stream (block_size prefixed, in order) - the form ssv_bamdec_info.unmapped_raw has, so that one consumer serves
 * both readers.  With read-ahead on (three batch sets) the bytes stay valid until the read after the next one. */
int ssvh_bam_unmapped_raw(const ssvh_bam *b, const uint8_t **raw, size_t *bytes);

/* ---- getsv bookkeeping around the BAM passes -------------------------------------------------- */

typedef struct ssvh_plan ssvh_plan; /* junction list -> windows / ranges / points, and back */

/* One junction as the reference keys it (Junction, getsv.h:149-227): contig NAMES, 1-based positions. */
typedef struct {
	const char *up_chr;
	const char *down_chr;
	int32_t up_pos;
	int32_t down_pos;
	char up_strand;
	char down_strand;
} ssvh_junction_in;

/*
 * Build the query plan for a junction multimap (given in Junction::operator< order, duplicates
 * allowed) plus extra one-end-unmapped points (GetBreak overload, getsv.cpp:791-802):
 *   - discordant windows per junction with mean/sd/times (getsv.cpp:1032-1060),
 *   - GetBreak's point set and the four flank windows per junction (unsigned arithmetic and its
 *     wrap-around included), MergeOverlap's begin2end, and from them the proper merged windows,
 *     the device range list and the device point list.
 * flank_length = -L (200).  do_discordant / do_depth mirror seeksv.cpp:246,288.
 */
int ssvh_plan_create(const ssvh_bam *bam, const ssvh_junction_in *junctions, int64_t n_junctions,
                     const char *const *extra_point_chr, const int32_t *extra_point_pos, int64_t n_extra_points,
                     int32_t mean, int32_t sd, int32_t times, int32_t flank_length, ssvh_plan **out);
/* New insert-size statistics: recompute the junction windows only (everything about depth is independent of them), so a
 * driver can build the plan while the GPU is still busy with the insert-size pass. */
int ssvh_plan_update_isize(ssvh_plan *p, int32_t mean, int32_t sd, int32_t times);
void ssvh_plan_destroy(ssvh_plan *p);

/* Device-facing tables (host memory owned by the plan). */
const ssv_junction *ssvh_plan_junctions(const ssvh_plan *p, int64_t *n);
const ssv_interval *ssvh_plan_windows(const ssvh_plan *p, int64_t *n);
const ssv_interval *ssvh_plan_ranges(const ssvh_plan *p, int64_t *n);
const ssv_interval *ssvh_plan_points(const ssvh_plan *p, int64_t *n);

/*
 * Fold device results back into the reference's per-junction view:
 *   counts[j]           -> abnormal_read_pair_no (junctions whose up_chr is not in the header keep
 *                          prev_counts[j], getsv.cpp:1043)
 *   up_depth/down_depth -> pos2depth values at both ends (before the support counts are added)
 *   flank[4*j + k]      -> range2depth sums of up_up, up_down, down_up, down_down, and
 *   flank_len[4*j + k]  -> the divisor (end - begin + 1) in unsigned arithmetic (getsv.cpp:946)
 * following main_depth's lookup rules for wrapped / empty windows.
 */
int ssvh_plan_fold(const ssvh_plan *p, const int32_t *counts, const int32_t *prev_counts,
                   const uint64_t *range_sum, const int32_t *point_depth,
                   int32_t *abnormal, int32_t *up_depth, int32_t *down_depth,
                   uint64_t *flank, uint32_t *flank_len, int32_t *extra_point_depth);

#ifdef __cplusplus
}
#endif
#endif
