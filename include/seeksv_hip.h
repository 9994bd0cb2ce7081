/*
 * seeksv_hip.h - C ABI of libseeksv_hip.so: the MI355X (gfx950) implementation of seeksv's
 * per-BAM-record hot path (getclip -> cluster -> getsv BAM passes).
 *
 * The reference (qiukunlong/seeksv v1.2.3) has no FFI/plugin API: the path sits behind four C++
 * call sites.  Each entry point below names the reference interface it replaces.
 * Conventions: plain C structs of pointers + counts over CALLER-OWNED buffers (structure of
 * arrays); every call returns int (0 = ok, <0 = ssv_status); nothing throws, nothing exits;
 * work is queued on the context's HIP stream and calls that return results synchronise that
 * stream before returning.  One context per GPU, not shared between threads.
 *
 * There is NO CPU fallback in this library: without a usable HIP device ssv_ctx_create fails
 * with SSV_E_NODEVICE and nothing else can be called.
 */
#ifndef SEEKSV_HIP_H_
#define SEEKSV_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSV_ABI_VERSION 9 /* v9: ssv_table_block_bytes(left_len, right_len) lost the parameters of the removed formats; ssv_bamdec_info.unmapped_raw stays valid for one more decode.
                             v8: table formats 1 and 2 (four-piece blocks with 4-bit bases) removed - 0 ASCII or 3 compact; ssv_group with SSV_GROUP_RCCL_ONE */

typedef enum {
	SSV_OK = 0,
	SSV_E_NODEVICE = -1, /* no HIP device / HIP runtime error at init */
	SSV_E_HIP = -2,      /* a HIP call failed; text in ssv_last_error() */
	SSV_E_ARG = -3,      /* bad argument */
	SSV_E_STATE = -4,    /* call out of sequence (e.g. finish before begin) */
	SSV_E_NOMEM = -5,    /* device allocation failed */
	SSV_E_RANGE = -6     /* an output buffer supplied by the caller is too small */
} ssv_status;

/* Where the arrays of a batch live. */
typedef enum {
	SSV_MEM_HOST = 0,   /* host memory (pinned preferred); the library stages it to the GPU */
	SSV_MEM_DEVICE = 1, /* already resident in this GPU's HBM; used in place, zero copies */
	/* SSV_MEM_DEVICE | SSV_MEM_PERSISTENT: a device batch whose `cigar` and `seqqual` arrays stay valid and unchanged until the
	 * getclip pass that scanned it has delivered its table (ssv_clip_cluster[_async] has returned): clip events then keep
	 * pointing at the batch's own bytes and the cluster table is cut straight out of them - nothing is copied in between.
	 * Without the flag ssv_clip_scan copies the bytes of the batch's clip events into context memory before it returns. */
	SSV_MEM_PERSISTENT = 2
} ssv_mem;

/*
 * The "cold" fields of one record as ONE 64-byte line: everything of BAM's fixed 32-byte record core (sam/bam.h:169-178)
 * plus the first five CIGAR operations and the offsets of the variable parts.  The streaming kernels never touch it (they read the
 * hot columns tid / pos / n_cigar below, 2-8 bytes per record); the per-candidate kernels (1-3 % of the records) fetch exactly this
 * one line instead of one 64-byte sector per structure-of-arrays column.  Arrays of ssv_record must be 64-byte aligned.
 */
typedef struct {
	int32_t tid, pos;         /* as in the hot columns */
	uint16_t flag;
	uint8_t mapq;
	uint8_t xc;               /* 1 if the XC:i aux value != 0 (clip_reads.cpp:126-129) */
	uint16_t n_cigar;
	uint16_t pad;
	int32_t l_qseq, mtid, mpos, isize;
	uint32_t cigar_off;       /* index of the record's first operation in cigar[] (all operations are there too) */
	uint32_t cigar_head[5];   /* operations 0..4 (0 beyond n_cigar): 99.9 % of short-read records need nothing else */
	uint64_t seq_off;         /* as seq_off[] below */
} ssv_record;

/*
 * One batch of decoded BAM records in file order.  Field meaning follows samtools' bam1_core_t (reference: sam/bam.h:169-178) -
 * the reference's samread() loop bodies (clip_reads.h:410, cluster.cpp:48, getsv.cpp:1067, bam2depth.h:29) consume exactly these.
 * Layout: the hot columns tid, pos, n_cigar (structure of arrays: what the streaming passes read) are always required; the other
 * fixed fields come either as the classic structure-of-arrays columns (flag ... seq_off) or as `rec`, one 64-byte line per record
 * (the device decoder and the GPU generator write it natively).  A batch without `rec` is transposed into one on the device first.
 *   seq_off[i] = byte offset into seqqual of record i's ceil(l_qseq/2) packed 4-bit bases followed
 *   by l_qseq quality bytes, or SSV_NO_SEQ when the batcher did not ship them.  The batcher must
 *   ship them for every record whose first or last CIGAR operation is 'S' (only those can become
 *   clip events, clip_reads.cpp:124,150); it may omit all others.
 *   A SSV_MEM_DEVICE batch must leave at least 8 readable bytes after seqqual[seqqual_bytes - 1] (whole aligned dwords are read).
 */
#define SSV_NO_SEQ UINT64_MAX
typedef struct {
	int64_t n;                 /* records in the batch (< 2^31) */
	int32_t mem;               /* ssv_mem */
	int32_t max_ref_span;      /* >= the longest reference span (sum of M,D,N,=,X lengths) of any record in the batch;
	                              0 = unknown (the library then measures it with an extra pass over the CIGARs) */
	const int32_t *tid;        /* [n] reference id, -1 = unplaced */
	const int32_t *pos;        /* [n] 0-based leftmost coordinate */
	const uint16_t *flag;      /* [n] */
	const uint8_t *mapq;       /* [n] */
	const uint16_t *n_cigar;   /* [n] */
	const int32_t *l_qseq;     /* [n] */
	const int32_t *mtid;       /* [n] */
	const int32_t *mpos;       /* [n] */
	const int32_t *isize;      /* [n] */
	const uint32_t *cigar_off; /* [n] index of the record's first op in cigar[] */
	const uint32_t *cigar;     /* [n_cigar_total] BAM encoding: len<<4 | op */
	const uint8_t *xc;         /* [n] 1 if the XC:i aux value != 0 (clip_reads.cpp:126-129); NULL = all 0 */
	const uint64_t *seq_off;   /* [n] see above */
	const uint8_t *seqqual;    /* [seqqual_bytes] */
	int64_t n_cigar_total;
	int64_t seqqual_bytes;
	const ssv_record *rec;     /* [n] or NULL; when given, flag / mapq / l_qseq / mtid / mpos / isize / cigar_off / xc / seq_off may be NULL */
	const uint8_t *cigar_ends; /* [n] or NULL: a hot copy of the two CIGAR operations the getclip pass looks at first - (code of the first
	                              operation) | (code of the last operation) << 4, BAM's operation codes; 0xff for a record without CIGAR.
	                              The streaming pass of getclip reads this byte per record and passes on the records with an `S` at either
	                              end (1 % of a WGS sample).  NULL: the library builds the column from the record lines first (64 B/record
	                              instead of 1: fill it where the records are parsed anyway). */
	const struct ssv_tid_run *tid_runs; /* [n_tid_runs] or NULL, HOST memory whatever `mem` says (it is a handful of entries): the tid column as runs -
	                              run k covers the records [tid_runs[k].first, tid_runs[k + 1].first) (the last one up to n), all on contig
	                              tid_runs[k].tid; tid_runs[0].first == 0, firsts strictly increasing.  A coordinate-sorted BAM has one run per
	                              contig, and the batcher sees the changes while it parses the records anyway.  With it (up to 48 runs) the
	                              streaming pass of getsv does not read the tid column at all - 4 of its 8 bytes per record - except in the
	                              few tiles a run boundary falls into; the column must still be there (other passes read single entries).
	                              The list's shape is always checked (SSV_E_ARG); with SSV_VERIFY_RUNS=1 in the environment ssv_getsv_scan also
	                              holds it against the tid column (one extra streaming pass) and refuses a list that disagrees. */
	int64_t n_tid_runs;
} ssv_batch_t;

typedef struct ssv_tid_run { int64_t first; int32_t tid; int32_t pad; } ssv_tid_run;

typedef struct ssv_ctx ssv_ctx; /* opaque */

/* ---- context ------------------------------------------------------------------------------- */

/* device = HIP device ordinal.  Fails (SSV_E_NODEVICE) when there is no GPU: no CPU path exists. */
int ssv_ctx_create(int device, ssv_ctx **out);
void ssv_ctx_destroy(ssv_ctx *ctx);
/* Block until everything queued on the context's stream has finished. */
int ssv_sync(ssv_ctx *ctx);
/* Text of the last error on this context (or of the last failed ssv_ctx_create when ctx == NULL). */
const char *ssv_last_error(const ssv_ctx *ctx);
int ssv_abi_version(void);
/* GPUs this process can see (0 when there is none or the HIP runtime is not usable). */
int ssv_device_count(void);
/* The hipStream_t the context launches on (for callers that time with HIP events). */
void *ssv_stream(ssv_ctx *ctx);

/* ---- host batches: pinned memory and the copy that runs ahead -------------------------------- */
/*
 * The reference's record loops read one bam1_t at a time through samread() (clip_reads.h:410, cluster.cpp:48, getsv.cpp:1067,
 * bam2depth.h:29); here the batcher decodes whole batches into two alternating sets of arrays and the library copies batch k+1 to HBM
 * while the kernels of batch k run:
 *     ssv_batch_prefetch(ctx, &b[1]);  ssv_clip_scan(ctx, &b[0]);  ssv_batch_prefetch(ctx, &b[2]);  ssv_clip_scan(ctx, &b[1]); ...
 * ssv_host_alloc hands out page-locked host memory (usable from every GPU); only copies out of such memory run asynchronously.
 * ssv_batch_prefetch starts copying a SSV_MEM_HOST batch on the context's upload stream into one of two staging sets and returns at once;
 * at most two batches may be announced and not yet scanned.  The scan calls (ssv_clip_scan, ssv_isize_scan, ssv_getsv_scan,
 * ssv_getsv_prime) must then be given exactly the announced batches, in that order (SSV_E_STATE otherwise).  With or without prefetch:
 * when a scan call returns, the batch's host arrays have been read and may be reused.
 * Announcements belong to the stream of batches, not to a pass: ssv_clip_begin leaves them alone (a driver ends a getclip pass and begins the
 * next in the middle of a stream - and of a batch, ssv_clip_scan_range - while batch k+1 is on its way already).  ssv_isize_begin and
 * ssv_getsv_begin start a new reading of the file and drop what is still announced; ssv_batch_prefetch_drop does so on request (a caller that
 * gives up a stream early) and returns once the copies in flight have read their host arrays.
 */
int ssv_host_alloc(size_t bytes, void **p);
int ssv_host_free(void *p);
/* Page-lock memory the caller already has - e.g. a range of a read-only file mapping (ssvh_bam_map_blocks, seeksv_host.h): the chunks of a BAM then
 * go from the page cache to the GPU by DMA, no staging copy (measured on the round-4 box: locking 0.03 s/GB on one thread, copies out of the locked
 * mapping at PCIe rate, 57 GB/s; allocating page-locked staging memory costs 0.25 s/GB before a byte is read).  p and bytes are rounded outwards to
 * whole pages by the caller.  SSV_E_HIP when the runtime refuses the range (callers fall back to staging buffers). */
int ssv_host_register(void *p, size_t bytes);
int ssv_host_unregister(void *p);
/* Ask for the pages of [p, p + bytes) - anonymous memory that has not been touched yet - on the NUMA node GPU `device` hangs on (mbind, MPOL_PREFERRED): a copy between the GPU
 * and host memory on the OTHER socket of a two-socket box crosses the sockets' link and runs at 60-70 % of the PCIe rate (round 6: the same 0.55 GB table took 9.7 or 13.8 ms
 * depending on where its pages had landed).  ssv_host_alloc does this for what it hands out (the device current at the call).  Best effort: SSV_OK whatever the kernel says. */
int ssv_host_bind_near(void *p, size_t bytes, int device);
int ssv_batch_prefetch(ssv_ctx *ctx, const ssv_batch_t *b);
int ssv_batch_prefetch_drop(ssv_ctx *ctx);

/* ---- batches that stay: one decode of the file for every pass ---------------------------------- */
/*
 * The reference reads the BAM once per command (getclip: clip_reads.h:410; getsv: cluster.cpp:48 for the insert sizes, getsv.cpp:1067 and
 * bam2depth.h:29 through the index) because each command is a process of its own.  What the passes look at is 80 bytes a record - the
 * hot columns, one 64-byte line, the CIGARs, the bases of the soft-clipped reads - so a 30x genome (617 M records) stays resident in 50 GB
 * of HBM: ssv_batch_retain copies a device batch (the decoder's, valid only until the next decode) into device memory of its own -
 * SSV_MEM_DEVICE | SSV_MEM_PERSISTENT, record lines built once - and every later pass scans it in place: the file is inflated once.
 * ssv_batch_release lets go of it: the memory is the context's again and the next ssv_batch_retain takes it from there (kept until the
 * context goes or any of its allocations runs out of device memory - a fresh allocation of GBs behind the release of tens of GB waits
 * for the driver to clear them, up to 1.7 s measured).  (`seeksv run` and bench.py's file leg work this way.)
 */
int ssv_batch_retain(ssv_ctx *ctx, const ssv_batch_t *device_batch, ssv_batch_t *out);
int ssv_batch_release(ssv_ctx *ctx, ssv_batch_t *retained);

/* ---- getclip: replaces InputBamOutputReads<>'s record loop (clip_reads.h:363, 410-446) ------ */

typedef struct {
	double match_rate;      /* -t, default 0.9  (seeksv.cpp:15,131) */
	int32_t min_mapq;       /* -q, default 1    (seeksv.cpp:130) */
	int32_t save_low_quality; /* -s             (seeksv.cpp:141) */
	/* Range-partitioned runs (one GPU per reference interval).  A rank scans its interval plus a halo of
	 * records that start before it, and keeps only the clip events whose breakpoint (tid, 1-based pos) lies in
	 * [own_lo, own_hi): every (contig, side, position) bin then lives on exactly one rank with its reads in BAM order.
	 * initial_last_tid = tid of the last mapped-pair record before the first scanned record (0 at the start of the
	 * file, clip_reads.h:407).  All zero = whole file on one GPU. */
	int32_t use_ownership;
	int32_t initial_last_tid;
	int32_t own_lo_tid, own_lo_pos;
	int32_t own_hi_tid, own_hi_pos;
} ssv_clip_params;

/* Start a getclip pass.  last_tid state starts at initial_last_tid (0 like the reference's, clip_reads.h:407). */
int ssv_clip_begin(ssv_ctx *ctx, const ssv_clip_params *p);
/*
 * Scan one batch (GetSClipReads, clip_reads.cpp:112-192, incl. the contig-switch rule of
 * clip_reads.h:423-438): appends the batch's clip events - key, slice lengths, CIGAR, where the read's packed bases and
 * qualities lie - to context-owned HBM.  A host batch's arrays may be reused when the call returns, a device batch's after ssv_sync(), except
 * the `cigar` and `seqqual` arrays of a SSV_MEM_PERSISTENT batch (see ssv_mem).
 */
int ssv_clip_scan(ssv_ctx *ctx, const ssv_batch_t *b);
/*
 * The same for the records [rec_begin, rec_end) of the batch only.  For input whose contigs come back (an unsorted BAM): the reference
 * flushes and clears its maps at EVERY change of contig among the mapped-pair records (clip_reads.h:423-438), so the reads of two
 * visits of one contig never share a cluster.  A pass of this library bins by (contig, side, position); a driver therefore ends the
 * pass (ssv_clip_cluster) in front of the first record of a visit whose contig is not greater than every contig of the pass so far
 * and starts the next one there (initial_last_tid = the contig before): scan [0, i) - cluster - begin - scan [i, n).  Records before
 * rec_begin still count as "the record before" of the contig-switch rule.  A batch announced with ssv_batch_prefetch stays announced
 * until a call with rec_end == n has consumed it.
 */
int ssv_clip_scan_range(ssv_ctx *ctx, const ssv_batch_t *b, int64_t rec_begin, int64_t rec_end);
/* Events collected so far (synchronises). */
int ssv_clip_event_count(ssv_ctx *ctx, int64_t *n_events);

/*
 * Cluster table = what the two multimaps hold when the reference flushes them
 * (InsertSeq/ChangeSeqAndQual, clip_reads.cpp:57-108,260-283), in the reference's emit order
 * (per contig: all '5' rows by position, then all '3' rows; ties in creation order,
 * clip_reads.h:432-433).  All pointers are HOST buffers owned by the context, valid until the
 * next ssv_clip_begin / ssv_ctx_destroy.
 */
typedef struct {
	int32_t tid;
	uint8_t side;              /* '5' or '3' */
	uint8_t pad[3];
	int64_t first;             /* index of the run's first cluster */
} ssv_table_run;

typedef struct {
	int64_t n_clusters;
	int64_t n_events;
	const int32_t *tid;        /* [n_clusters] */
	const int32_t *pos;        /* [n_clusters] 1-based breakpoint coordinate */
	const uint8_t *side;       /* [n_clusters] '5' or '3' */
	const int32_t *support;    /* [n_clusters] support_read_no */
	const int32_t *left_len;   /* [n_clusters] |seq_left|  */
	const int32_t *right_len;  /* [n_clusters] |seq_right| */
	const uint8_t *qual_missing; /* [n_clusters] 1: reference prints "*" for both qualities */
	const uint64_t *str_off;   /* [n_clusters] offset into str of: seq_left, qual_left, seq_right, qual_right; every block starts
	                              4-byte aligned and is zero padded to a multiple of 4 bytes */
	const uint8_t *str;        /* ASCII, not NUL terminated */
	const uint64_t *cigar_off; /* [n_clusters] offset into cigar */
	const int32_t *n_cigar;    /* [n_clusters] ops of the record whose CIGAR the cluster carries */
	const uint32_t *cigar;     /* BAM-encoded ops INCLUDING S/H (GenerateCigar drops those when printing) */
	int32_t seq_packed;        /* 0: the layout above (format 0).  1: format 3, below. */
	int32_t qual_bits;         /* 8: qualities are characters (phred + 33), one byte each; format 3: bits per quality (or per group of qualities) in the
	                              quality stream, below.  Lossless: base qualities are most of the table's bytes and come from a handful of values
	                              on current sequencers. */
	uint8_t qual_alphabet[64]; /* index -> quality character (v8: 64 places; the entries in use are non-zero and increasing) */
	/* ---- format 3, the compact table: what the host can rebuild does not cross PCIe ----
	 * Handed out by ssv_clip_table_wait: pos, c_cigar (cigar too when cigar_bytes is 4), str and the fields below; tid, side, support, left_len, right_len, qual_missing,
	 * n_cigar, str_off and cigar_off are NULL until ssv_clip_table_expand() has rebuilt them on the host.
	 *   c_len      [n_clusters][2] left_len, right_len, len_bytes (2 or 4) wide each
	 *   c_support  [n_clusters] support_bytes (2 or 4) wide;  c_ncig [n_clusters] ncig_bytes (1 or 2) wide;  c_flags bit 0 = qual_missing
	 *   runs       the table is a sequence of (contig, side) runs; run r covers clusters [runs[r].first, runs[r + 1].first)
	 *   a cluster's string block = [base stream | quality stream], each zero padded to whole 32-bit words; blocks follow each other in
	 *   cluster order (str_off = running sum of ssv_table_block_bytes3), CIGARs likewise (cigar_off = running sum of n_cigar):
	 *     base stream     the n = left_len + right_len bases of seq_left then seq_right, base_bits each, base i at stream bits
	 *                     [i * base_bits, +base_bits) (bit b of a stream = bit b % 8 of byte b / 8).  base_bits 2: index into "ACGT", and
	 *                     every base that is something else is listed in base_exc (the stream holds 0 there); base_bits 4 (only when
	 *                     that list would be too long): index into "=ACMGRSVTWYHKDBN"
	 *     quality stream  the n qualities the same way at qual_bits each (8: characters), all zero when qual_missing.  qual_group k > 1 (v7; alphabets
	 *                     whose size R is far from a power of two - five values: k = 3, qual_bits = 7; nine to eleven: k = 2, qual_bits = 7; v8: 17 to 45 values - a
	 *                     HiSeq-style 40-value alphabet -: k = 2, qual_bits = 11, i.e. 5.5 bits a quality instead of 8): the stream is
	 *                     ceil(n / k) groups of qual_bits bits, group g at stream bits [g * qual_bits, +qual_bits), holding qualities
	 *                     g k .. g k + k - 1 as the number i_0 + R i_1 + R^2 i_2 (i_j = index into qual_alphabet; places behind the stream's end: 0),
	 *                     R = the number of qual_alphabet entries in use (they are characters, so non-zero)
	 *   base_exc   sorted; cluster << 28 | base index << 4 | index into "=ACMGRSVTWYHKDBN" */
	int32_t format;            /* the ssv_clip_table_format the table was built with */
	int32_t base_bits;
	int32_t len_bytes, support_bytes, ncig_bytes;
	int32_t qual_group;        /* format 3: qualities per group of qual_bits bits (1: every quality its own field) */
	const void *c_len;
	const void *c_support;
	const void *c_ncig;
	const uint8_t *c_flags;
	const ssv_table_run *runs;
	int64_t n_runs;
	const uint64_t *base_exc;
	int64_t n_base_exc;
	uint64_t str_bytes;        /* bytes of str / operations of cigar (all formats) */
	uint64_t cigar_ops;
	int64_t support_sum;       /* format 3, after ssv_clip_table_expand: sum of the support column (= n_events: every clip event is in one cluster) */
	const void *c_cigar;       /* format 3 (v7): the CIGAR operations as they crossed PCIe, cigar_bytes wide each - 2: length << 4 | code in 16 bits (every
	                              length of the table is below 4096; `cigar` is NULL then: read (const uint16_t *)c_cigar + cigar_off[k]); 4: = cigar */
	int32_t cigar_bytes, pad4;
} ssv_cluster_table;

/* Table format of the following ssv_clip_cluster[_async] calls: 0 ASCII (default), 3 the compact table (see ssv_cluster_table: ~100
 * instead of ~330 bytes per 150-base cluster; the table is the path's output and PCIe bounds the step).  (1 and 2, the four-piece blocks
 * with 4-bit bases of ABI versions below 8, are gone: SSV_E_ARG.) */
int ssv_clip_table_format(ssv_ctx *ctx, int format);
/* Format 3: rebuild the columns that did not cross PCIe (contig and side from the runs, the widened support / lengths / flags, string
 * and CIGAR offsets as running sums) into context-owned host memory, on n_threads host threads (0: as many as the machine has), and
 * set the pointers in *t.  Valid as long as the table. */
int ssv_clip_table_expand(ssv_ctx *ctx, ssv_cluster_table *t, int32_t n_threads);
/* Bytes of one cluster's string block in format 3 (n_bases = left_len + right_len). */
uint64_t ssv_table_block_bytes3(int64_t n_bases, int32_t base_bits, int32_t qual_bits);
/* The same for a table whose qualities go in groups (ssv_cluster_table.qual_group; 1 = the function above). */
uint64_t ssv_table_block_bytes3g(int64_t n_bases, int32_t base_bits, int32_t qual_bits, int32_t qual_group);
/* Bytes of one cluster's string block in format 0: seq_left, qual_left, seq_right, qual_right as characters, padded to a multiple of 4 (v9: two parameters). */
uint64_t ssv_table_block_bytes(int32_t left_len, int32_t right_len);

/* Sort events into (contig, side, position) bins and run the greedy consensus clustering. */
int ssv_clip_cluster(ssv_ctx *ctx, ssv_cluster_table *out);
/*
 * The same in two halves: ssv_clip_cluster_async returns as soon as the clustering kernels are done and the copy of the table
 * to host memory has been queued on a second stream (n_clusters / n_events are known then); ssv_clip_table_wait blocks until
 * that copy has landed and hands the table out.  Two tables are kept, so the caller may start the next getclip pass (or the
 * getsv passes) while the previous table is still crossing PCIe: a table stays valid until the second-next
 * ssv_clip_cluster[_async] call.
 */
int ssv_clip_cluster_async(ssv_ctx *ctx, int64_t *n_clusters, int64_t *n_events);
int ssv_clip_table_wait(ssv_ctx *ctx, ssv_cluster_table *out);
/* The table of the call before the most recent ssv_clip_cluster[_async] (pipelined drivers: cluster k+1, then collect table k). */
int ssv_clip_table_wait_prev(ssv_ctx *ctx, ssv_cluster_table *out);

/* ---- getsv pass 1: replaces CalculateInsertsizeDeviation (cluster.cpp:15-83) ---------------- */

int ssv_isize_begin(ssv_ctx *ctx, int32_t min_mapq, int64_t max_pairs);
/* *done is set to 1 once max_pairs qualifying records have been seen (the reference breaks there). */
int ssv_isize_accumulate(ssv_ctx *ctx, const ssv_batch_t *b, int32_t *done);
/* Returns n_pairs==0 exactly when the reference returns 1 and leaves mean/sd untouched. */
int ssv_isize_finish(ssv_ctx *ctx, int64_t *n_pairs, int32_t *mean, int32_t *sd);

/* ---- getsv passes 2+3, fused: FindDiscordantReadPairs (getsv.cpp:990-1120) and
 *      main_depth (bam2depth.cpp:17-142) ------------------------------------------------------- */

/* One junction and its (already clamped) query window, getsv.cpp:1041-1060. */
typedef struct {
	int32_t up_tid;      /* tid of up_chr */
	int32_t down_tid;    /* tid of down_chr, -1 if absent from the header (never matches) */
	int32_t up_pos;      /* 1-based */
	int32_t down_pos;    /* 1-based */
	int32_t beg;         /* 0-based window start after clamping */
	int32_t end;         /* window end after clamping; candidates: ref_end > beg && pos < end */
	uint8_t up_strand;   /* '+' or '-' */
	uint8_t down_strand;
	uint8_t pad[2];
} ssv_junction;

/* 1-based inclusive interval on a contig. */
typedef struct {
	int32_t tid;
	int32_t beg;
	int32_t end;
} ssv_interval;

typedef struct {
	/* discordant tally (may be disabled with n_junctions = 0) */
	const ssv_junction *junctions; /* host */
	int64_t n_junctions;
	int32_t mean, sd, times;       /* from ssv_isize_finish; times = 4 (seeksv.cpp:161) */
	int32_t disc_min_mapq;         /* -q (20) */
	/* depth (may be disabled with n_windows = 0) */
	const ssv_interval *windows;   /* host; merged windows: sorted by (tid,beg), pairwise disjoint */
	int64_t n_windows;
	int32_t depth_min_mapq;        /* -q (20) */
	int32_t n_targets;             /* contigs in the BAM header */
	const int32_t *target_len;     /* host [n_targets] */
} ssv_getsv_params;

int ssv_getsv_begin(ssv_ctx *ctx, const ssv_getsv_params *p);
/* One fused pass over a batch: discordant-pair tally per junction + coverage of the windows. */
int ssv_getsv_scan(ssv_ctx *ctx, const ssv_batch_t *b);
/*
 * Range-partitioned runs (one GPU per run of records): the reference's pileup keeps at most ~8000 reads alive (bam_plp_push), a rule that
 * depends on the records BEFORE a rank's first one.  Right after ssv_getsv_begin a rank replays the last records before its range through
 * the pileup's bookkeeping only (nothing of them is counted: they belong to the rank before).  *sufficient = 0: the replayed batch itself
 * begins inside a > 8000x stack (or is shorter than 16,384 records without starting at the file's first record - the caller knows): call
 * ssv_getsv_begin again and replay a longer run.  At WGS depths the first 24,576 records before the range always suffice.
 */
int ssv_getsv_prime(ssv_ctx *ctx, const ssv_batch_t *b, int32_t *sufficient);
/*
 * counts[n_junctions]      = abnormal_read_pair_no per junction (getsv.cpp:1116)
 * range_sum[n_ranges]      = sum over the interval of per-column depth (bam2depth.cpp:101-122);
 *                            each range must lie inside one window
 * point_depth[n_points]    = depth at (tid, beg) (bam2depth.cpp:123-124); 0 outside every window
 * max_depth                = largest per-column depth inside the windows.  The depths follow libbam 0.1.16's pileup, which keeps at
 *                            most ~8000 reads alive: a read that is not the first at its start position is dropped when 2 + (accepted
 *                            reads of the contig ending at or after that start) > 8000 (bam_plp_push); batches must arrive in file order
 */
int ssv_getsv_finish(ssv_ctx *ctx, int32_t *counts,
                     const ssv_interval *ranges, int64_t n_ranges, uint64_t *range_sum,
                     const ssv_interval *points, int64_t n_points, int32_t *point_depth,
                     int32_t *max_depth);

/* ---- range-partitioned runs: the one exchange step ------------------------------------------------------------------
 * One context per GPU, one host thread per context.  Each rank scans its own run of records (libseeksv_host's ssvh_bam_partition cuts the
 * file) for ALL junctions and windows; the per-rank result vectors - discordant counts, depth sums, point depths: KBs to a few MB - are
 * all-gathered and added up by every rank.  RCCL (ncclAllGather over xGMI, loaded on first use) when the ranks sit on different GPUs;
 * ranks that share a GPU (tests on a one-GPU box) exchange through host memory.  SSV_GROUP_RCCL_ONE=1 makes a group of ONE rank an RCCL
 * communicator as well (ncclCommInitAll over one device), so that the RCCL branch runs end to end where there is a single GPU. */
typedef struct ssv_group ssv_group;
int ssv_group_create(ssv_ctx **ctxs, int n, ssv_group **out);
void ssv_group_destroy(ssv_group *g);
int ssv_group_uses_rccl(const ssv_group *g);
/* Called by every rank from its own thread: send = this rank's `bytes` bytes (host memory), recv = n * bytes, rank order.
 * No rank is ever left waiting: the ranks agree through the host - before anything is exchanged - that all of them arrived, prepared and
 * bring vectors of one size; a rank whose ncclAllGather fails aborts the group's communicators (ncclCommAbort) so that its peers come back;
 * a rank that does not arrive within SSV_GROUP_TIMEOUT_S (default 600) breaks the group.  In every such case EVERY rank returns an error
 * (SSV_E_HIP / SSV_E_ARG for the rank at fault, SSV_E_STATE for the others) and nothing in `recv` may be used; after an abort or a timeout
 * the group only answers SSV_E_STATE.  (SSV_GROUP_FAIL=<rank>:<1|2|3> injects a failure before the exchange / in the
 * collective's call / behind an issued collective: tests.) */
int ssv_group_allgather(ssv_group *g, int rank, const void *send, size_t bytes, void *recv);

/* ---- device-side BGZF inflate + BAM record decode ---------------------------- */

/*
 * What libbam's samread() does per record on one core (sam/sam.h:73: bgzf inflate + bam_read1) for a whole chunk of the file on
 * the GPU: the caller hands over the COMPRESSED bytes of a run of whole BGZF blocks plus their table, and gets the decoded records
 * as an SSV_MEM_DEVICE batch that ssv_clip_scan / ssv_isize_accumulate / ssv_getsv_scan consume directly - the inflated bytes
 * never cross PCIe.  libseeksv_host's ssvh_bam_read_blocks produces the input.
 *   one chunk = blocks in file order; a record may straddle chunks (its head is carried over inside the context);
 *   c_off/c_len = the block's raw deflate payload inside `comp` (after the 18-byte BGZF header, without the 8-byte trailer),
 *   u_len = its ISIZE.  A chunk must inflate to < 4 GB.
 */
typedef struct {
	uint64_t c_off;
	uint32_t c_len;
	uint32_t u_len;
} ssv_bgzf_block;

typedef struct {
	int64_t n_records;           /* records of the last chunk */
	uint64_t inflated_bytes;     /* bytes the chunk inflated to */
	uint64_t tail_offset;        /* internal: where the unfinished record starts */
	uint32_t repaired_blocks;    /* blocks whose speculated first record start had to be corrected (diagnostic) */
	int32_t last_tid;            /* contig of the last record without UNMAP|MUNMAP so far */
	/* host-side lists of the last chunk (pinned memory owned by the context, valid until the next decode; unmapped_raw: until the decode AFTER
	 * the next one - two buffers in turn, so that a host thread can pair a chunk's reads up while the next chunk is decoded): */
	const uint8_t *unmapped_raw; /* the UNMAP|MUNMAP records as they lie in the BAM stream (block_size prefixed), in order: */
	uint64_t unmapped_bytes;     /*   the unmapped-pair FASTQ side channel of getclip (clip_reads.h:415-419) decodes them on the host */
	uint32_t n_tid_runs;         /* contig changes among the other records, in order (the flush sequence of clip_reads.h:423-438): */
	const uint32_t *tid_run_index; /* record index inside the batch, */
	const int32_t *tid_run_tid;    /* its contig */
} ssv_bamdec_info;

/* Start a file: n_targets of its header (plausibility of speculated record starts), and the offset of the first record inside the
 * inflated stream of the first chunk (= the length of the BAM header: magic, text, reference list). */
int ssv_bamdec_begin(ssv_ctx *ctx, int32_t n_targets, uint64_t first_record_offset);
/* Optional, after ssv_bamdec_begin: the lengths of the n_targets contigs (the BAM header's l_ref values).  Record starts inside a chunk are found by
 * speculation and then verified; with the lengths a candidate whose position lies outside its contig is dismissed at once, which spares the verifier
 * the rare block it would have to walk record by record.  Results do not depend on it. */
int ssv_bamdec_target_lens(ssv_ctx *ctx, const int32_t *lens);
/* A run of records that ends inside the next chunk (one rank's share of a file, ssvh_bam_raw_begin_range): its records end `inflated_bytes`
 * into the chunk's blocks - a record boundary; what the blocks hold behind it belongs to the next run.  Applies to the next decode call only. */
int ssv_bamdec_limit(ssv_ctx *ctx, uint64_t inflated_bytes);
/* Optional, after ssv_bamdec_begin: chunks that inflate to up to `inflated_bytes` will follow - the decoder's buffers are then sized for them by the
 * first decode, however small its chunk (a driver starts a file with a small chunk so that the GPU gets to work early; growing ~40 buffers when the
 * first full-size chunk arrives cost 0.2 s). */
int ssv_bamdec_expect(ssv_ctx *ctx, uint64_t inflated_bytes);
/* Optional, after ssv_bamdec_begin: on != 0 makes every decode check each inflated block's CRC32 against the one in the block's BGZF trailer (one
 * more pass over bytes that are in HBM anyway) and refuse the chunk on a mismatch.  Off by default, like libbam 0.1.16's reader (sam/sam.h:73), which
 * checks no CRC: a block whose deflate structure is valid but whose bytes were damaged decodes silently there - and here, without this. */
int ssv_bamdec_verify_crc(ssv_ctx *ctx, int on);
/* ... and that starts inside the file: the contig of the last mapped-pair record before it (0 at the start of the file, clip_reads.h:407) -
 * ssv_bamdec_info's contig-change list continues from there.  After ssv_bamdec_begin, before the first decode. */
int ssv_bamdec_prev_tid(ssv_ctx *ctx, int32_t tid);
/* Pinned host buffers (three, which = 0 | 1 | 2; grow-only, owned by the context) to read the compressed bytes of a chunk into, so that a
 * reader thread can fill one while the chunk in the second is on its way to the GPU and the one in the third is being decoded; any host memory
 * works too (page-locked: ssv_host_register - e.g. the file's own pages in a mapping, no staging copy at all).  A buffer is free again when the
 * decode call that was given it returns. */
int ssv_bamdec_staging(ssv_ctx *ctx, int which, size_t bytes, void **host_ptr);
/* Optional: announce a chunk ahead.  Its compressed bytes start their way to the GPU at once, on the context's upload stream, while the chunk
 * before it is being inflated (two chunks may be announced at any time); ssv_bamdec_decode of the same (comp, comp_bytes) then finds them there instead
 * of copying.  The host buffer must stay untouched until that decode call returns; it should be page-locked (ssv_bamdec_staging, ssv_host_alloc) -
 * out of pageable memory the copy is not asynchronous.  Typical loop: prefetch(0); for k: prefetch(k+1); decode(k); ... */
int ssv_bamdec_prefetch(ssv_ctx *ctx, const void *comp, size_t comp_bytes);
/* Give up chunks that were announced and will not be decoded (a pass that stops early): returns when their copies have left the host memory. */
int ssv_bamdec_prefetch_drop(ssv_ctx *ctx);
/* Inflate + decode one chunk.  *out is an SSV_MEM_DEVICE batch owned by the context, valid until the next decode on it (stream
 * ordered: kernels already enqueued on the context's stream may still read it).  n_blocks == 0 = end of input (fails if a record
 * is unfinished).  keep_all_seq as in ssvh_bam_read_batch.  Synchronises the stream. */
int ssv_bamdec_decode(ssv_ctx *ctx, const void *comp, size_t comp_bytes, const ssv_bgzf_block *blocks, int64_t n_blocks, int keep_all_seq, ssv_batch_t *out);
int ssv_bamdec_last(ssv_ctx *ctx, ssv_bamdec_info *info);
/* Copy a device batch into host arrays owned by the context (valid until the next call): tests and debugging. */
int ssv_batch_to_host(ssv_ctx *ctx, const ssv_batch_t *device_batch, ssv_batch_t *host_batch);

/* ---- measurement --------------------------------------------------------------------------- */

/*
 * Per-kernel launch timing with HIP events on the context's stream.  Enable, run, then read back.
 * name = kernel name as in DESIGN.md ("clip_scan", "getsv_scan", ...).  Returns SSV_E_ARG for an
 * unknown name.  Reading synchronises and resets nothing; ssv_prof_reset clears.
 */
int ssv_prof_enable(ssv_ctx *ctx, int on);
int ssv_prof_reset(ssv_ctx *ctx);
int ssv_prof_get(ssv_ctx *ctx, const char *name, double *total_ms, int64_t *launches, int64_t *units);
/* Names of all timed kernels, '\n' separated. */
const char *ssv_prof_names(void);

/* ---- clipped-sequence re-aligner ----------------------------------------------------------------------
 * Stand-in for the EXTERNAL `bwa mem` step between `seeksv getclip` and `seeksv getsv` (README.md:22-34, example/seeksv.sh:3:
 * `bwa mem ref.fa prefix.clip.fq.gz | samtools view -Sb - > prefix.clip.bam`) on hosts without bwa, for references that behave
 * like random sequence (the synthetic genomes of bench.py / tests).  getsv consumes of each clip.bam record: flag & {4, 16, 256},
 * MAPQ == 0 or not, tid, pos, the CIGAR with its S ends, the read name = the sequence (getsv.cpp:25-71, getsv.h:445-527).
 * K-mer index of the reference in HBM + seed look-ups on both strands + ungapped extension with bwa mem's default scores
 * (match 1, mismatch 4, end clipping 5, minimum 30).  NOT bit-identical to bwa: no gaps, no chaining, one record per query. */
typedef struct {
	int32_t tid;         /* -1: unaligned (flag 4) */
	int32_t pos;         /* 0-based reference position of the aligned segment's first base */
	int32_t q_beg, q_end;/* aligned segment [q_beg, q_end) of the query as a BAM record stores it (reverse-complemented when `reverse`): CIGAR = q_beg S, M, rest S */
	int32_t score;       /* matches - 4 * mismatches of the segment */
	int32_t second;      /* best score at another locus (0: none) */
	int32_t n_mismatch;
	uint8_t reverse;     /* flag 16 */
	uint8_t mapq;        /* 0 when another locus scores as well, 60 when the runner-up is >= 10 behind */
	uint8_t pad[2];
} ssv_realign_hit;

/* Build the index.  ref2bit: base i of the concatenated contigs at bits [2 (i % 32), +2) of word i / 32, A C G T = 0 1 2 3 (the
 * caller decides what N becomes); target_off[n_targets + 1] = first base of every contig, target_off[0] = 0, target_off[n_targets] =
 * n_bases.  SSV_MEM_DEVICE arrays are used in place and need one readable word after the last one.  *n_dropped (optional) = sampled
 * positions that found no slot within the probe limit (low-complexity sequence). */
int ssv_realign_index(ssv_ctx *ctx, const uint64_t *ref2bit, int32_t mem, int64_t n_bases, const int64_t *target_off, int32_t n_targets, int64_t *n_dropped);
/* Align n ASCII sequences (host memory, concatenated; seq_off[n + 1]) -> hits[n] (host).  Queries shorter than 20 or longer than 1024
 * bases come back unaligned. */
int ssv_realign_query(ssv_ctx *ctx, const char *seqs, const uint64_t *seq_off, int64_t n, ssv_realign_hit *hits);
int ssv_realign_free(ssv_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* SEEKSV_HIP_H_ */
