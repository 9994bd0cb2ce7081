/*
 * seeksv_oracle.c - TEST INFRASTRUCTURE ONLY (see seeksv_oracle.h for the rules and parity status).
 *
 * CPU restatement of seeksv v1.2.3's per-record arithmetic.  Written from the behaviour of the
 * reference; every function cites the reference file:line it follows.  Straightforward, single
 * threaded, no cleverness: this is the thing the HIP kernels are compared against.
 */
#include "seeksv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* sam/bam.h:97-124 */
enum { F_PAIRED = 1, F_PROPER = 2, F_UNMAP = 4, F_MUNMAP = 8, F_REV = 16, F_MREV = 32, F_SECONDARY = 256, F_QCFAIL = 512, F_DUP = 1024 };
/* sam/bam.h:133-155 */
enum { C_M = 0, C_I = 1, C_D = 2, C_N = 3, C_S = 4, C_H = 5, C_P = 6, C_EQ = 7, C_X = 8 };

static const char NT16[] = "=ACMGRSVTWYHKDBN"; /* bam_nt16_rev_table */

/* ---------------------------------------------------------------------------------------------- */
/* getclip                                                                                        */
/* ---------------------------------------------------------------------------------------------- */

typedef struct {
	char *sl, *ql, *sr, *qr; /* forward strings like the reference's ReadsInfo (clip_reads.h:44-53) */
	int ll, lr, capl, capr;
	int batch, rec;          /* whose CIGAR the cluster carries (cigar_vec) */
	int support;
	int pos;
	int qual_missing;
	int next;                /* next cluster with the same (run, side, pos), creation order */
} cluster_t;

typedef struct {
	int pos;
	int head, tail;
	int used;
} slot_t;

typedef struct {
	slot_t *slots;
	int cap, count;
	int *members; /* cluster ids of this map in creation order */
	int nmem, capmem;
} posmap_t;

typedef struct {
	cluster_t *cl;
	int64_t ncl, capcl;
	posmap_t map[2]; /* 0 = breakpoint2read_l ('5'), 1 = breakpoint2read_r ('3') */
	/* emitted order */
	int *order; uint8_t *order_side; int32_t *order_tid;
	int64_t nord, capord;
	int64_t n_events;
	const ssv_clip_params *p;
	int32_t cur_tid;
} clip_state;

static void posmap_init(posmap_t *m)
{
	m->cap = 1024; m->count = 0;
	m->slots = (slot_t *)calloc((size_t)m->cap, sizeof(slot_t));
	m->members = NULL; m->nmem = 0; m->capmem = 0;
}

static void posmap_clear(posmap_t *m)
{
	memset(m->slots, 0, (size_t)m->cap * sizeof(slot_t));
	m->count = 0; m->nmem = 0;
}

static uint32_t hash_pos(int pos) { uint32_t x = (uint32_t)pos * 2654435761u; return x ^ (x >> 15); }

static slot_t *posmap_find(posmap_t *m, int pos, int create)
{
	if (create && (m->count + 1) * 2 > m->cap) {
		int ocap = m->cap; slot_t *os = m->slots;
		m->cap *= 2;
		m->slots = (slot_t *)calloc((size_t)m->cap, sizeof(slot_t));
		for (int i = 0; i < ocap; ++i) if (os[i].used) {
			uint32_t h = hash_pos(os[i].pos) & (uint32_t)(m->cap - 1);
			while (m->slots[h].used) h = (h + 1) & (uint32_t)(m->cap - 1);
			m->slots[h] = os[i];
		}
		free(os);
	}
	uint32_t h = hash_pos(pos) & (uint32_t)(m->cap - 1);
	while (m->slots[h].used) {
		if (m->slots[h].pos == pos) return &m->slots[h];
		h = (h + 1) & (uint32_t)(m->cap - 1);
	}
	if (!create) return NULL;
	m->slots[h].used = 1; m->slots[h].pos = pos; m->slots[h].head = m->slots[h].tail = -1;
	m->count++;
	return &m->slots[h];
}

/* CompareStringEndFirst, clip_reads.cpp:194-205: matches over the right-aligned overlap */
static int match_end(const char *a, int la, const char *b, int lb, int *len)
{
	int n = la < lb ? la : lb, m = 0;
	for (int i = 0; i < n; ++i) if (a[la - 1 - i] == b[lb - 1 - i]) ++m;
	*len = n; return m;
}

/* CompareStringBeginFirst, clip_reads.cpp:207-217 */
static int match_begin(const char *a, int la, const char *b, int lb, int *len)
{
	int n = la < lb ? la : lb, m = 0;
	for (int i = 0; i < n; ++i) if (a[i] == b[i]) ++m;
	*len = n; return m;
}

/* (double)total_match/len >= limit with len==0 giving NaN (false), clip_reads.cpp:204,216,266 */
static int rate_ok(int m, int n, double limit)
{
	volatile double r = (double)m / (double)n;
	return r >= limit;
}

/* GetSeq, clip_reads.cpp:286-306 */
static void get_seq(const ssv_batch_t *b, int64_t i, int begin, int ll, int lr, char *sl, char *ql, char *sr, char *qr, int *qual_missing)
{
	const uint8_t *s = b->seqqual + b->seq_off[i];
	int lq = b->l_qseq[i];
	const uint8_t *t = s + (lq + 1) / 2;
	for (int k = 0; k < ll; ++k) { int p = begin + k; sl[k] = NT16[(s[p >> 1] >> ((~p & 1) << 2)) & 15]; }
	for (int k = 0; k < lr; ++k) { int p = begin + ll + k; sr[k] = NT16[(s[p >> 1] >> ((~p & 1) << 2)) & 15]; }
	if (lq > 0 && t[0] == 0xff) {
		/* reference: qual_left = qual_right = "*" (one char).  We keep per-base '*' so that a merge
		 * (undefined behaviour in the reference) is at least deterministic; printing honours the flag. */
		*qual_missing = 1;
		for (int k = 0; k < ll; ++k) ql[k] = '*';
		for (int k = 0; k < lr; ++k) qr[k] = '*';
	} else {
		*qual_missing = 0;
		for (int k = 0; k < ll; ++k) ql[k] = (char)(t[begin + k] + 33);
		for (int k = 0; k < lr; ++k) qr[k] = (char)(t[begin + ll + k] + 33);
	}
}

static void grow(char **s, char **q, int *cap, int need)
{
	if (need <= *cap) return;
	int c = *cap ? *cap : 64;
	while (c < need) c *= 2;
	*s = (char *)realloc(*s, (size_t)c);
	*q = (char *)realloc(*q, (size_t)c);
	*cap = c;
}

/* ReadsInfo::ChangeSeqAndQual, clip_reads.cpp:57-108.  left_clipped = (aa == LEFT_CLIPPED) */
static void change_seq_and_qual(cluster_t *c, const char *sl, const char *ql, int l2, const char *sr, const char *qr, int r2, int batch, int rec, int left_clipped)
{
	int len1 = c->ll, len2 = l2;
	int len = len1 < len2 ? len1 : len2;
	for (int i = 0; i < len; ++i) {
		if ((signed char)c->ql[len1 - 1 - i] < (signed char)ql[len2 - 1 - i]) {
			c->ql[len1 - 1 - i] = ql[len2 - 1 - i];
			c->sl[len1 - 1 - i] = sl[len2 - 1 - i];
		}
	}
	if (len1 <= len2) {
		int extra = len2 - len1;
		grow(&c->sl, &c->ql, &c->capl, len2);
		memmove(c->sl + extra, c->sl, (size_t)len1); memcpy(c->sl, sl, (size_t)extra);
		memmove(c->ql + extra, c->ql, (size_t)len1); memcpy(c->ql, ql, (size_t)extra);
		c->ll = len2;
		if (!left_clipped) { c->batch = batch; c->rec = rec; }
	}
	len1 = c->lr; len2 = r2; len = len1 < len2 ? len1 : len2;
	for (int i = 0; i < len; ++i) {
		if ((signed char)c->qr[i] < (signed char)qr[i]) { c->qr[i] = qr[i]; c->sr[i] = sr[i]; }
	}
	if (len1 < len2) {
		grow(&c->sr, &c->qr, &c->capr, len2);
		memcpy(c->sr + len1, sr + len1, (size_t)(len2 - len1));
		memcpy(c->qr + len1, qr + len1, (size_t)(len2 - len1));
		c->lr = len2;
		if (left_clipped) { c->batch = batch; c->rec = rec; }
	}
}

/* InsertSeq, clip_reads.cpp:260-283 */
static void insert_seq(clip_state *st, int side, int pos, const char *sl, const char *ql, int ll, const char *sr, const char *qr, int lr,
                       int qual_missing, int batch, int rec, double limit)
{
	posmap_t *m = &st->map[side];
	if (st->p->use_ownership) {
		/* range-partitioned runs: only the events whose breakpoint lies in [own_lo, own_hi) belong to this rank */
		long long k = ((long long)st->cur_tid << 32) | (unsigned)pos;
		long long lo = ((long long)st->p->own_lo_tid << 32) | (unsigned)st->p->own_lo_pos, hi = ((long long)st->p->own_hi_tid << 32) | (unsigned)st->p->own_hi_pos;
		if (k < lo || k >= hi) return;
	}
	slot_t *slot = posmap_find(m, pos, 1);
	st->n_events++;
	for (int id = slot->head; id >= 0; id = st->cl[id].next) {
		cluster_t *c = &st->cl[id];
		int n1, n2;
		int m1 = match_end(sl, ll, c->sl, c->ll, &n1);
		if (!rate_ok(m1, n1, limit)) continue;
		int m2 = match_begin(sr, lr, c->sr, c->lr, &n2);
		if (!rate_ok(m2, n2, limit)) continue;
		change_seq_and_qual(c, sl, ql, ll, sr, qr, lr, batch, rec, side == 0);
		c->support++;
		return;
	}
	if (st->ncl == st->capcl) {
		st->capcl = st->capcl ? st->capcl * 2 : 1024;
		st->cl = (cluster_t *)realloc(st->cl, (size_t)st->capcl * sizeof(cluster_t));
		slot = posmap_find(m, pos, 0);
	}
	int id = (int)st->ncl++;
	cluster_t *c = &st->cl[id];
	memset(c, 0, sizeof(*c));
	grow(&c->sl, &c->ql, &c->capl, ll > 0 ? ll : 1);
	grow(&c->sr, &c->qr, &c->capr, lr > 0 ? lr : 1);
	memcpy(c->sl, sl, (size_t)ll); memcpy(c->ql, ql, (size_t)ll);
	memcpy(c->sr, sr, (size_t)lr); memcpy(c->qr, qr, (size_t)lr);
	c->ll = ll; c->lr = lr; c->batch = batch; c->rec = rec; c->support = 1; c->pos = pos; c->qual_missing = qual_missing; c->next = -1;
	if (slot->head < 0) slot->head = slot->tail = id;
	else { st->cl[slot->tail].next = id; slot->tail = id; }
	if (m->nmem == m->capmem) { m->capmem = m->capmem ? m->capmem * 2 : 1024; m->members = (int *)realloc(m->members, (size_t)m->capmem * sizeof(int)); }
	m->members[m->nmem++] = id;
}

static const cluster_t *g_sort_cl;
static int cmp_member(const void *a, const void *b)
{
	int x = *(const int *)a, y = *(const int *)b;
	if (g_sort_cl[x].pos != g_sort_cl[y].pos) return g_sort_cl[x].pos < g_sort_cl[y].pos ? -1 : 1;
	return x < y ? -1 : (x > y); /* multimap: equal keys stay in insertion order */
}

/* DisplaySClipReadsAndClipFq('5') then ('3') + clear, clip_reads.h:432-435,443-446 */
static void flush_run(clip_state *st, int32_t tid)
{
	for (int side = 0; side < 2; ++side) {
		posmap_t *m = &st->map[side];
		g_sort_cl = st->cl;
		if (m->nmem) qsort(m->members, (size_t)m->nmem, sizeof(int), cmp_member); /* (an empty side has no array yet) */
		for (int k = 0; k < m->nmem; ++k) {
			if (st->nord == st->capord) {
				st->capord = st->capord ? st->capord * 2 : 1024;
				st->order = (int *)realloc(st->order, (size_t)st->capord * sizeof(int));
				st->order_side = (uint8_t *)realloc(st->order_side, (size_t)st->capord);
				st->order_tid = (int32_t *)realloc(st->order_tid, (size_t)st->capord * sizeof(int32_t));
			}
			st->order[st->nord] = m->members[k];
			st->order_side[st->nord] = side ? '3' : '5';
			st->order_tid[st->nord] = tid;
			st->nord++;
		}
		posmap_clear(m);
	}
}

/* GenerateCigar's l, clip_reads.cpp:309-329: sum of M, D, =, N lengths (X is NOT counted) */
static int ref_len_generate_cigar(const uint32_t *cig, int n)
{
	int l = 0;
	for (int i = 0; i < n; ++i) {
		int op = (int)(cig[i] & 15);
		if (op == C_M || op == C_D || op == C_EQ || op == C_N) l += (int)(cig[i] >> 4);
	}
	return l;
}

/* GetSClipReads, clip_reads.cpp:112-192 */
static void get_sclip_reads(clip_state *st, const ssv_batch_t *b, int bi, int64_t i, const ssv_clip_params *p, char **buf, int *bufcap)
{
	int n = b->n_cigar[i];
	if (n == 0) return;                 /* reference reads cigar[-1]: undefined; we emit nothing */
	if (b->tid[i] < 0) return;          /* reference indexes target_name[-1]: undefined; we emit nothing */
	const uint32_t *cig = b->cigar + b->cigar_off[i];
	int op1 = (int)(cig[0] & 15), op2 = (int)(cig[n - 1] & 15);
	if (op1 == C_H || op2 == C_H || (int)b->mapq[i] < p->min_mapq || (b->flag[i] & F_DUP)) return;
	int s1 = op1 == C_S, s2 = op2 == C_S;
	if (!s1 && !s2) return;
	int xc = b->xc ? b->xc[i] : 0;
	int lq = b->l_qseq[i];
	int pos0 = b->pos[i];
	if (b->seq_off[i] == SSV_NO_SEQ) return; /* batcher contract violated; nothing sensible to do */
	if (4 * lq > *bufcap) { *bufcap = 4 * lq + 64; *buf = (char *)realloc(*buf, (size_t)*bufcap); }
	char *sl = *buf, *ql = sl + lq, *sr = ql + lq, *qr = sr + lq;
	int qm;
	if (s1 != s2) {
		if (xc != 0 && !p->save_low_quality) return;
		int ref_len = ref_len_generate_cigar(cig, n);
		if (s1) {
			int ll = (int)(cig[0] >> 4), lr = lq - ll;
			if (lr < 0) return;
			get_seq(b, i, 0, ll, lr, sl, ql, sr, qr, &qm);
			insert_seq(st, 0, pos0 + 1, sl, ql, ll, sr, qr, lr, qm, bi, (int)i, p->match_rate);
		} else {
			int lr = (int)(cig[n - 1] >> 4), ll = lq - lr;
			if (ll < 0) return;
			get_seq(b, i, 0, ll, lr, sl, ql, sr, qr, &qm);
			insert_seq(st, 1, pos0 + ref_len, sl, ql, ll, sr, qr, lr, qm, bi, (int)i, p->match_rate);
		}
	} else {
		/* A CIGAR that is one lone 'nS' lands here too: cig[0] and cig[n - 1] are the same operation (clip_reads.cpp:115,150).  The
		 * "middle" length l_qseq - ll - rc is then negative and GetSeq's loops over it run zero times (clip_reads.cpp:293-294): the '5'
		 * event is (seq_left = [0, ll), seq_right = ""), the '3' event (seq_left = "", seq_right = [l_qseq - rc, l_qseq)).  With an empty
		 * part CompareString* divide 0 by 0 (NaN >= limit is false): such an event never joins a cluster and nothing ever joins its own.
		 * Clips longer than the read make the reference read outside the record: undefined there, nothing here. */
		int ll = (int)(cig[0] >> 4), rc = (int)(cig[n - 1] >> 4), mid = lq - ll - rc;
		if (ll > lq || rc > lq) return;
		int midc = mid < 0 ? 0 : mid;
		int ref_len = ref_len_generate_cigar(cig, n);
		int do_l = 1, do_r = 1;
		if (xc != 0 && !p->save_low_quality) { if (!(b->flag[i] & F_REV)) do_r = 0; else do_l = 0; }
		if (do_l) {
			get_seq(b, i, 0, ll, midc, sl, ql, sr, qr, &qm);
			insert_seq(st, 0, pos0 + 1, sl, ql, ll, sr, qr, midc, qm, bi, (int)i, p->match_rate);
		}
		if (do_r) {
			get_seq(b, i, ll + mid - midc, midc, rc, sl, ql, sr, qr, &qm); /* seq_right = [ll + mid, ll + mid + rc) whatever the sign of mid */
			insert_seq(st, 1, pos0 + ref_len, sl, ql, midc, sr, qr, rc, qm, bi, (int)i, p->match_rate);
		}
	}
}

/* InputBamOutputReads' record loop, clip_reads.h:407-446 */
int orc_getclip(const ssv_batch_t *batches, int n_batches, const ssv_clip_params *p, orc_cluster_table *out)
{
	clip_state st;
	memset(&st, 0, sizeof(st));
	posmap_init(&st.map[0]); posmap_init(&st.map[1]);
	char *buf = NULL; int bufcap = 0;
	int32_t last_tid = p->initial_last_tid; /* 0 in the reference (clip_reads.h:407) */
	st.p = p;
	for (int bi = 0; bi < n_batches; ++bi) {
		const ssv_batch_t *b = &batches[bi];
		for (int64_t i = 0; i < b->n; ++i) {
			if (b->flag[i] & (F_UNMAP | F_MUNMAP)) continue; /* unmapped-pair FASTQ side channel: host only */
			st.cur_tid = b->tid[i];
			if (b->tid[i] == last_tid) get_sclip_reads(&st, b, bi, i, p, &buf, &bufcap);
			else { flush_run(&st, last_tid); last_tid = b->tid[i]; } /* this record is NOT processed */
		}
	}
	flush_run(&st, last_tid);

	memset(out, 0, sizeof(*out));
	int64_t nc = st.nord;
	out->n_clusters = nc; out->n_events = st.n_events;
	out->tid = (int32_t *)malloc((size_t)(nc + 1) * 4); out->pos = (int32_t *)malloc((size_t)(nc + 1) * 4);
	out->side = (uint8_t *)malloc((size_t)nc + 1); out->support = (int32_t *)malloc((size_t)(nc + 1) * 4);
	out->left_len = (int32_t *)malloc((size_t)(nc + 1) * 4); out->right_len = (int32_t *)malloc((size_t)(nc + 1) * 4);
	out->qual_missing = (uint8_t *)malloc((size_t)nc + 1);
	out->str_off = (uint64_t *)malloc((size_t)(nc + 1) * 8); out->cigar_off = (uint64_t *)malloc((size_t)(nc + 1) * 8);
	out->n_cigar = (int32_t *)malloc((size_t)(nc + 1) * 4);
	uint64_t sb = 0, co = 0;
	for (int64_t k = 0; k < nc; ++k) {
		const cluster_t *c = &st.cl[st.order[k]];
		out->str_off[k] = sb; sb += (2ull * (uint64_t)(c->ll + c->lr) + 3ull) & ~3ull; /* same layout as ssv_cluster_table: 4-byte aligned blocks */
		out->cigar_off[k] = co; co += batches[c->batch].n_cigar[c->rec];
	}
	out->str_bytes = (int64_t)sb; out->cigar_ops = (int64_t)co;
	out->str = (uint8_t *)calloc((size_t)sb + 1, 1); out->cigar = (uint32_t *)malloc((size_t)(co + 1) * 4);
	for (int64_t k = 0; k < nc; ++k) {
		const cluster_t *c = &st.cl[st.order[k]];
		out->tid[k] = st.order_tid[k]; out->pos[k] = c->pos; out->side[k] = st.order_side[k];
		out->support[k] = c->support; out->left_len[k] = c->ll; out->right_len[k] = c->lr; out->qual_missing[k] = (uint8_t)c->qual_missing;
		uint8_t *d = out->str + out->str_off[k];
		memcpy(d, c->sl, (size_t)c->ll); d += c->ll; memcpy(d, c->ql, (size_t)c->ll); d += c->ll;
		memcpy(d, c->sr, (size_t)c->lr); d += c->lr; memcpy(d, c->qr, (size_t)c->lr);
		const ssv_batch_t *b = &batches[c->batch];
		out->n_cigar[k] = b->n_cigar[c->rec];
		memcpy(out->cigar + out->cigar_off[k], b->cigar + b->cigar_off[c->rec], (size_t)b->n_cigar[c->rec] * 4);
	}
	for (int64_t k = 0; k < st.ncl; ++k) { free(st.cl[k].sl); free(st.cl[k].ql); free(st.cl[k].sr); free(st.cl[k].qr); }
	free(st.cl); free(st.order); free(st.order_side); free(st.order_tid);
	for (int s = 0; s < 2; ++s) { free(st.map[s].slots); free(st.map[s].members); }
	free(buf);
	return 0;
}

void orc_cluster_table_free(orc_cluster_table *t)
{
	free(t->tid); free(t->pos); free(t->side); free(t->support); free(t->left_len); free(t->right_len);
	free(t->qual_missing); free(t->str_off); free(t->str); free(t->cigar_off); free(t->n_cigar); free(t->cigar);
	memset(t, 0, sizeof(*t));
}

/* ---------------------------------------------------------------------------------------------- */
/* getsv BAM passes                                                                               */
/* ---------------------------------------------------------------------------------------------- */

/* IsHardClip, clip_reads.cpp:247-257 (n_cigar == 0 reads cigar[-1] in the reference: we say "no") */
static int is_hard_clip(const ssv_batch_t *b, int64_t i)
{
	int n = b->n_cigar[i];
	if (n == 0) return 0;
	const uint32_t *cig = b->cigar + b->cigar_off[i];
	return (cig[0] & 15) == C_H || (cig[n - 1] & 15) == C_H;
}

/* CalculateInsertsizeDeviation, cluster.cpp:15-83 */
int orc_isize_stats(const ssv_batch_t *batches, int n_batches, int32_t min_mapq, int64_t max_pairs,
                    int64_t *n_pairs, int32_t *mean, int32_t *sd)
{
	int64_t cap = 1 << 16, n = 0;
	int *v = (int *)malloc((size_t)cap * sizeof(int));
	unsigned long total = 0;
	int stop = 0;
	for (int bi = 0; bi < n_batches && !stop; ++bi) {
		const ssv_batch_t *b = &batches[bi];
		for (int64_t i = 0; i < b->n; ++i) {
			if ((int)b->mapq[i] < min_mapq) continue;              /* __g_skip_aln with g_min_mapQ = min_mapQ (cluster.cpp:41,51) */
			if (is_hard_clip(b, i)) continue;                      /* cluster.cpp:61 */
			int isz = b->isize[i];
			int f = b->flag[i];
			if ((f & F_PAIRED) && (f & F_PROPER) && !(f & F_DUP) && isz > 0) { /* cluster.cpp:62 */
				total += (unsigned long)isz;
				if (n == cap) { cap *= 2; v = (int *)realloc(v, (size_t)cap * sizeof(int)); }
				v[n++] = isz;
			}
			if (n == max_pairs) { stop = 1; break; }               /* cluster.cpp:68 */
		}
	}
	*n_pairs = n;
	if (n == 0) { free(v); return 1; }                              /* cluster.cpp:71 */
	int m = (int)(total / (unsigned long)n);                       /* cluster.cpp:72 */
	double d = 0;
	for (int64_t k = 0; k < n; ++k) {
		/* cluster.cpp:77: the product is formed in int (wraps for |x-mean| > 46340) and added to a double */
		int diff = (int)((unsigned)v[k] - (unsigned)m);
		int sq = (int)((unsigned)diff * (unsigned)diff);
		d += sq;
	}
	*mean = m;
	*sd = (int)sqrt(d / (double)n);                                 /* cluster.cpp:80 */
	free(v);
	return 0;
}

/* IsConcordant, cluster.cpp:136-147 */
static int is_concordant(int flag, int isize, int mean, int sd, int times)
{
	int lo = mean - sd * times, hi = mean + sd * times;
	if (!(flag & F_REV) && (flag & F_MREV) && lo <= isize && isize <= hi) return 1;
	if ((flag & F_REV) && !(flag & F_MREV) && isize < 0) {
		int a = abs(isize);
		return lo <= a && a <= hi;
	}
	return 0;
}

/* bam_calend of samtools 0.1.16 (binary in sam/libbam.a; disassembled: only M, D, N advance) and the
 * is_overlap() test of bam_iter_read: rend = n_cigar ? calend : pos + 1 */
static int calend(const ssv_batch_t *b, int64_t i)
{
	int n = b->n_cigar[i], end = b->pos[i];
	if (n == 0) return end + 1;
	const uint32_t *cig = b->cigar + b->cigar_off[i];
	for (int k = 0; k < n; ++k) {
		int op = (int)(cig[k] & 15);
		if (op == C_M || op == C_D || op == C_N) end += (int)(cig[k] >> 4);
	}
	return end;
}

/* one candidate record against one junction: the body of the bam_iter_read loop, getsv.cpp:1069-1113 */
static int discordant_hit(const ssv_batch_t *b, int64_t i, const ssv_junction *j, int mean, int sd, int times, int min_mapq, int min_ins, int max_ins)
{
	if ((int)b->mapq[i] < min_mapq) return 0;                      /* __g_skip_aln, g_min_mapQ set at getsv.cpp:1027 */
	if (is_hard_clip(b, i)) return 0;                              /* getsv.cpp:1071 */
	int f = b->flag[i];
	if ((f & F_DUP) || (f & F_UNMAP) || (f & F_MUNMAP) || is_concordant(f, b->isize[i], mean, sd, times)) return 0; /* :1072 */
	int mtid = j->down_tid;
	if (!(mtid != -1 && mtid == b->mtid[i])) return 0;             /* :1074 */
	int pos = b->pos[i], mpos = b->mpos[i], lq = b->l_qseq[i];
	int up = j->up_pos, down = j->down_pos;
	const int K = 5;                                                /* kCrossLength, getsv.cpp:15 */
	if (j->up_strand == '+' && j->down_strand == '+' && pos + lq <= up + K && mpos + 1 >= down - K) {
		if (!(f & F_REV) && (f & F_MREV)) {
			if (j->up_tid == mtid && up > down && up - down + 1 + 2 * lq <= max_ins) {      /* tandem duplication, :1081 */
				int ins = up - pos + mpos + lq - down + 1;
				while (ins <= max_ins) {
					if (ins >= min_ins) return 1;
					ins += up - down + 1;
				}
				return 0;
			}
			int ins = up - pos + mpos + lq - down + 1;
			return min_ins <= ins && ins <= max_ins;
		}
		return 0;
	} else if (j->up_strand == '-' && j->down_strand == '+' && (f & F_REV) && (f & F_MREV) && mpos + 1 >= down - K) {
		int ins = pos + 1 - up + 1 + mpos + lq - down + 1;             /* :1103 */
		return min_ins <= ins && ins <= max_ins;
	} else if (j->up_strand == '+' && j->down_strand == '-' && !(f & F_REV) && !(f & F_MREV) && pos + lq <= up + K && mpos + lq <= down + K) {
		int ins = up - pos + down - (mpos + lq) + 1;                   /* :1109 */
		return min_ins <= ins && ins <= max_ins;
	}
	return 0;
}

/* FindDiscordantReadPairs, getsv.cpp:990-1120.  The index query (bam_iter_query/bam_iter_read) yields the
 * records of tid with calend > beg && pos < end; we find them by a plain scan of the batches. */
int orc_discordant(const ssv_batch_t *batches, int n_batches, const ssv_junction *junctions, int64_t n_junctions,
                   int32_t mean, int32_t sd, int32_t times, int32_t min_mapq, int32_t *counts)
{
	int min_ins = mean - sd * times, max_ins = mean + sd * times;
	if (min_ins < 0) min_ins = 0;                                   /* getsv.cpp:1033-1034 */
	/* per batch: is it coordinate sorted, and the longest reference span (to bound the scan) */
	int *sorted = (int *)malloc((size_t)n_batches * sizeof(int));
	int *maxspan = (int *)malloc((size_t)n_batches * sizeof(int));
	for (int bi = 0; bi < n_batches; ++bi) {
		const ssv_batch_t *b = &batches[bi];
		sorted[bi] = 1; maxspan[bi] = 1;
		for (int64_t i = 0; i < b->n; ++i) {
			int sp = calend(b, i) - b->pos[i];
			if (sp > maxspan[bi]) maxspan[bi] = sp;
			if (i > 0) {
				uint32_t t0 = (uint32_t)b->tid[i - 1], t1 = (uint32_t)b->tid[i];
				if (t1 < t0 || (t1 == t0 && b->pos[i] < b->pos[i - 1])) sorted[bi] = 0;
			}
		}
	}
	for (int64_t jn = 0; jn < n_junctions; ++jn) {
		const ssv_junction *j = &junctions[jn];
		int c = 0;
		for (int bi = 0; bi < n_batches; ++bi) {
			const ssv_batch_t *b = &batches[bi];
			int64_t lo = 0, hi = b->n;
			if (sorted[bi]) {
				/* first record with (tid,pos) >= (up_tid, beg - maxspan) */
				int64_t a = 0, z = b->n;
				int64_t want = (int64_t)j->beg - maxspan[bi];
				while (a < z) {
					int64_t m = (a + z) / 2;
					uint32_t t = (uint32_t)b->tid[m];
					if (t < (uint32_t)j->up_tid || (t == (uint32_t)j->up_tid && (int64_t)b->pos[m] < want)) a = m + 1; else z = m;
				}
				lo = a;
			}
			for (int64_t i = lo; i < hi; ++i) {
				if (sorted[bi]) {
					uint32_t t = (uint32_t)b->tid[i];
					if (t > (uint32_t)j->up_tid || (t == (uint32_t)j->up_tid && b->pos[i] >= j->end)) break;
				}
				if (b->tid[i] != j->up_tid) continue;
				if (!(calend(b, i) > j->beg && b->pos[i] < j->end)) continue;
				c += discordant_hit(b, i, j, mean, sd, times, min_mapq, min_ins, max_ins);
			}
		}
		counts[jn] = c;
	}
	free(sorted); free(maxspan);
	return 0;
}

static int64_t find_window(const ssv_interval *w, int64_t n, int32_t tid, int32_t col)
{
	/* last window with (tid,beg) <= (tid,col) */
	int64_t a = 0, z = n;
	while (a < z) {
		int64_t m = (a + z) / 2;
		if (w[m].tid < tid || (w[m].tid == tid && w[m].beg <= col)) a = m + 1; else z = m;
	}
	if (a == 0) return -1;
	--a;
	if (w[a].tid != tid || col > w[a].end) return -1;
	return a;
}

/* main_depth, bam2depth.cpp:17-142 with read_bam (bam2depth.h:29-35) and libbam 0.1.16's pileup:
 * a read contributes iff MAPQ >= mapQ and none of UNMAP|SECONDARY|QCFAIL|DUP (BAM_DEF_MASK, sam/bam.h:124),
 * tid >= 0; it adds 1 to every column under an M base (deletion and ref-skip columns appear in the
 * pileup but are subtracted again at bam2depth.cpp:92-96; '=' / 'X' are skipped by the pileup).
 * The pileup's read cap (bam_plp_push of samtools 0.1.16: `iter->tid == b->core.tid && iter->pos == b->core.pos &&
 * iter->mp->cnt > iter->maxcnt` with maxcnt = 8000 and two nodes of the pool always allocated): a read that starts where the
 * previous accepted read of its contig started - i.e. not the first read at its start - is dropped when
 * 2 + (accepted reads of the contig whose bam_calend end is >= the start) > 8000.  The first read at a new start is always taken.
 * Pinned by tests/golden/getsv/deep*.  (Reads without any M/D/N operation are not counted as live here.) */
#define ORC_PLP_MAXCNT 8000
#define ORC_RING_BITS 22
int orc_depth(const ssv_batch_t *batches, int n_batches, const ssv_interval *windows, int64_t n_windows,
              int32_t min_mapq, const ssv_interval *ranges, int64_t n_ranges, uint64_t *range_sum,
              const ssv_interval *points, int64_t n_points, int32_t *point_depth, int32_t *max_depth)
{
	int64_t *off = (int64_t *)malloc((size_t)(n_windows + 1) * sizeof(int64_t));
	off[0] = 0;
	for (int64_t k = 0; k < n_windows; ++k) off[k + 1] = off[k] + (windows[k].end - windows[k].beg + 1);
	int32_t *depth = (int32_t *)calloc((size_t)off[n_windows] + 1, sizeof(int32_t));
	/* pileup state: accepted reads of the current contig that are still allocated, by their (exclusive, 0-based) end */
	const int64_t ring_mask = ((int64_t)1 << ORC_RING_BITS) - 1;
	int32_t *ring = (int32_t *)calloc((size_t)ring_mask + 1, sizeof(int32_t));
	int32_t plp_tid = -1;
	int64_t plp_pos = -1, live = 0;
	for (int bi = 0; bi < n_batches; ++bi) {
		const ssv_batch_t *b = &batches[bi];
		for (int64_t i = 0; i < b->n; ++i) {
			int f = b->flag[i];
			if ((int)b->mapq[i] < min_mapq) continue;
			if (f & (F_UNMAP | F_SECONDARY | F_QCFAIL | F_DUP)) continue;
			if (b->tid[i] < 0) continue;
			int n = b->n_cigar[i];
			const uint32_t *cig = b->cigar + b->cigar_off[i];
			{
				int64_t p0 = b->pos[i], end = p0;
				for (int k = 0; k < n; ++k) { int op = (int)(cig[k] & 15); if (op == C_M || op == C_D || op == C_N) end += (int64_t)(cig[k] >> 4); }
				if (b->tid[i] != plp_tid) { memset(ring, 0, ((size_t)ring_mask + 1) * sizeof(int32_t)); live = 0; plp_tid = b->tid[i]; plp_pos = -1; }
				if (p0 != plp_pos) {
					/* columns before p0 have been emitted: nodes with end <= p0 - 1 are gone */
					if (plp_pos >= 0 && p0 - plp_pos <= ring_mask) { for (int64_t e = plp_pos; e < p0; ++e) { live -= ring[e & ring_mask]; ring[e & ring_mask] = 0; } }
					else if (plp_pos >= 0) { memset(ring, 0, ((size_t)ring_mask + 1) * sizeof(int32_t)); live = 0; }
					plp_pos = p0;
				} else if (2 + live > ORC_PLP_MAXCNT) continue; /* dropped by bam_plp_push */
				if (end > p0) { ring[end & ring_mask]++; live++; }
			}
			int32_t col = b->pos[i] + 1; /* 1-based */
			{
				/* cheap skip of reads that touch no window (windows are disjoint and sorted, so also sorted by end) */
				int32_t span = 0;
				for (int k = 0; k < n; ++k) { int op = (int)(cig[k] & 15); if (op == C_M || op == C_EQ || op == C_X || op == C_D || op == C_N) span += (int32_t)(cig[k] >> 4); }
				int64_t a = 0, z = n_windows;
				while (a < z) {
					int64_t m = (a + z) / 2;
					if (windows[m].tid < b->tid[i] || (windows[m].tid == b->tid[i] && windows[m].end < col)) a = m + 1; else z = m;
				}
				if (a == n_windows || windows[a].tid != b->tid[i] || windows[a].beg > col + span - 1) continue;
			}
			for (int k = 0; k < n; ++k) {
				int op = (int)(cig[k] & 15), len = (int)(cig[k] >> 4);
				/* libbam 0.1.16 pileup (resolve_cigar / bam_calend): only M covers, only M, D, N advance the reference; '=' and 'X' are
				 * skipped like padding (pinned by tests/golden/getsv/eqx.*) */
				if (op == C_M) {
					for (int x = 0; x < len; ++x) {
						int64_t w = find_window(windows, n_windows, b->tid[i], col + x);
						if (w >= 0) depth[off[w] + (col + x - windows[w].beg)]++;
					}
					col += len;
				} else if (op == C_D || op == C_N) col += len;
			}
		}
	}
	int32_t mx = 0;
	for (int64_t k = 0; k < off[n_windows]; ++k) if (depth[k] > mx) mx = depth[k];
	if (max_depth) *max_depth = mx;
	for (int64_t r = 0; r < n_ranges; ++r) {
		uint64_t s = 0;
		for (int64_t c = ranges[r].beg; c <= ranges[r].end; ++c) {
			int64_t w = find_window(windows, n_windows, ranges[r].tid, (int32_t)c);
			if (w >= 0) s += (uint64_t)depth[off[w] + (c - windows[w].beg)];
		}
		range_sum[r] = s;
	}
	for (int64_t q = 0; q < n_points; ++q) {
		int64_t w = find_window(windows, n_windows, points[q].tid, points[q].beg);
		point_depth[q] = w >= 0 ? depth[off[w] + (points[q].beg - windows[w].beg)] : 0;
	}
	free(off); free(depth); free(ring);
	return 0;
}
