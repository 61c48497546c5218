/*
 * bsalign_compat.c -- the reference's single-pair hot-path functions (same names and signatures), as thin
 * C wrappers over the batch C-ABI (include/bsalign_hip.h).  See include/bsalign_compat.h.
 */
#include "../../include/bsalign_compat.h"
#include "../../include/bsalign_hip.h"
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* one context per calling thread: the reference's functions are re-entrant given distinct mempools, and a bsa_ctx_t
 * (error string, events, workspace) is not shared between threads (include/bsalign_hip.h) */
static __thread bsa_ctx_t *g_ctx = NULL;
static int g_device = 0;
/* a thread's context (streams, events, workspace, kept device buffers) goes with the thread: the key's destructor runs at thread exit */
static pthread_key_t g_ctx_key;
static pthread_once_t g_ctx_once = PTHREAD_ONCE_INIT;
static void ctx_at_thread_exit(void *p){ if(p) bsa_ctx_destroy((bsa_ctx_t*)p); }
static void ctx_key_make(void){ (void)pthread_key_create(&g_ctx_key, ctx_at_thread_exit); }

void bsalign_compat_set_device(int device){ g_device = device; }

void bsalign_compat_shutdown(void){
	if(g_ctx){
		(void)pthread_once(&g_ctx_once, ctx_key_make);
		(void)pthread_setspecific(g_ctx_key, NULL);
		bsa_ctx_destroy(g_ctx); g_ctx = NULL;
	}
}

static void die(const char *what, const char *func){
	fflush(stdout);
	fprintf(stderr, " -- %s in %s -- %s:%d --\n", what, func, __FILE__, __LINE__);
	fflush(stderr);
	abort();
}

static bsa_ctx_t *ctx(const char *func){
	if(!g_ctx){
		if(bsa_ctx_create(g_device, &g_ctx) != BSA_OK) die("no usable MI355X device (there is no CPU fallback)", func);
		(void)pthread_once(&g_ctx_once, ctx_key_make);
		(void)pthread_setspecific(g_ctx_key, g_ctx);
	}
	return g_ctx;
}

/* ---- list helpers (layout of list.h:116-122) ---- */
b1v *adv_init_b1v(u8i init_size, int mem_zero, int aligned_base, u4i n_head){
	b1v *l = (b1v*)malloc(sizeof(b1v));
	void *p = NULL;
	if(init_size == 0) init_size = 2;
	if(aligned_base < (int)sizeof(void*)) aligned_base = sizeof(void*);
	if(posix_memalign(&p, (size_t)aligned_base, (size_t)(init_size + n_head))) p = NULL;
	if(p && mem_zero) memset(p, 0, (size_t)(init_size + n_head));
	l->buffer = p ? (b1i*)p + n_head : NULL;
	l->size = 0; l->cap = init_size; l->mem_zero = mem_zero ? 1 : 0; l->n_head = n_head; l->aligned = (u8i)aligned_base; l->off = 0;
	return l;
}
void free_b1v(b1v *l){ if(l){ if(l->buffer) free(l->buffer - l->n_head); free(l); } }
void clear_b1v(b1v *l){ l->size = 0; l->off = 0; }
u4v *init_u4v(u8i init_size){
	u4v *l = (u4v*)malloc(sizeof(u4v));
	if(init_size == 0) init_size = 2;
	l->buffer = (u4i*)malloc((size_t)init_size * sizeof(u4i));
	l->size = 0; l->cap = init_size; l->mem_zero = 0; l->n_head = 0; l->aligned = 8; l->off = 0;
	return l;
}
void free_u4v(u4v *l){ if(l){ free(l->buffer); free(l); } }
void clear_u4v(u4v *l){ l->size = 0; l->off = 0; }
static void encap_u4v(u4v *l, u8i n){
	if(l->size + n <= l->cap) return;
	u8i cap = l->cap ? l->cap : 2;
	while(cap < l->size + n) cap += (cap < 0xFFFFFu) ? cap : 0xFFFFFu;
	l->buffer = (u4i*)realloc(l->buffer, (size_t)cap * sizeof(u4i));
	l->cap = cap;
}
void push_u4v(u4v *l, u4i e){ encap_u4v(l, 1); l->buffer[l->size++] = e; }

void banded_striped_epi8_seqalign_set_score_matrix(b1i matrix[16], b1i mat, b1i mis){ bsa_set_score_matrix(matrix, mat, mis); }

static void check_mempool(b1v *mempool, const char *func){
	if(mempool && mempool->aligned < 16){
		fflush(stdout);
		fprintf(stderr, " -- mempool should be aligned by (%d) but (%d) bytes in %s -- %s:%d --\n", 16, (int)mempool->aligned, func, __FILE__, __LINE__);
		fflush(stderr);
		abort();
	}
}

static void deliver_cigars(u4v *cigars, int mode, const uint32_t *words, uint64_t n){
	if(!cigars) return;
	if(!(mode & SEQALIGN_MODE_CIGRESV)) clear_u4v(cigars);
	encap_u4v(cigars, n);
	memcpy(cigars->buffer + cigars->size, words, (size_t)n * sizeof(u4i));
	cigars->size += n;
}

seqalign_result_t banded_striped_epi8_seqalign_pairwise(u1i *qseq, u4i qlen, u1i *tseq, u4i tlen, b1v *mempool, u4v *cigars,
		int mode, u4i bandwidth, b1i matrix[16], b1i gapo1, b1i gape1, b1i gapo2, b1i gape2, int verbose){
	seqalign_result_t rs;
	bsa_align_params_t par;
	bsa_result_t out;
	uint64_t qoff = 0, toff = qlen, off[2] = {0, 0};
	uint32_t status = 0, *cig;
	uint8_t *seqs;
	size_t cap = (size_t)qlen + tlen + 8;
	int rc;
	(void)verbose;
	memset(&rs, 0, sizeof(rs));
	check_mempool(mempool, __FUNCTION__);
	if(qlen == 0 || tlen == 0){ if(cigars && !(mode & SEQALIGN_MODE_CIGRESV)) clear_u4v(cigars); return rs; }
	par.mode = seqalign_mode_type(mode);
	par.bandwidth = bandwidth;      /* 0 = the whole query, as in bsalign.h:3861 */
	memcpy(par.matrix, matrix, 16);
	par.gapo1 = gapo1; par.gape1 = gape1; par.gapo2 = gapo2; par.gape2 = gape2;
	seqs = (uint8_t*)malloc((size_t)qlen + tlen + 1);
	cig = (uint32_t*)malloc(cap * sizeof(uint32_t));
	memcpy(seqs, qseq, qlen); memcpy(seqs + qlen, tseq, tlen);
	rc = bsa_align_batch(ctx(__FUNCTION__), seqs, (size_t)qlen + tlen, &qoff, &qlen, &toff, &tlen, 1, &par, &out, cig, cap, off, &status);
	if(rc != BSA_OK){
		fprintf(stderr, " -- device alignment failed (%d: %s)", rc, bsa_last_error(g_ctx));
		die("", __FUNCTION__);
	}
	if(status & BSA_ST_BAD_BASE) die("base code > 3 in input", __FUNCTION__);
	if(status & BSA_ST_TRACE) die("traceback left the band (the reference does not terminate on this input)", __FUNCTION__);
	memcpy(&rs, &out, sizeof(rs));
	deliver_cigars(cigars, mode, cig, off[1]);
	free(seqs); free(cig);
	return rs;
}

seqalign_result_t striped_seqedit_pairwise(u1i *qseq, u4i qlen, u1i *tseq, u4i tlen, int mode, u4i bandwidth,
		b1v *mempool, u4v *cigars, int verbose){
	seqalign_result_t rs;
	bsa_edit_params_t par;
	bsa_result_t out;
	uint64_t qoff = 0, toff = qlen, off[2] = {0, 0};
	uint32_t status = 0, *cig;
	uint8_t *seqs;
	size_t cap = (size_t)qlen + tlen + 8;
	int rc;
	(void)verbose;
	memset(&rs, 0, sizeof(rs));
	check_mempool(mempool, __FUNCTION__);
	if(qlen == 0 || tlen == 0) return rs;               /* bsalign.h:1051-1054 */
	par.mode = seqalign_mode_type(mode);
	par.bandwidth = bandwidth;
	seqs = (uint8_t*)malloc((size_t)qlen + tlen + 1);
	cig = (uint32_t*)malloc(cap * sizeof(uint32_t));
	memcpy(seqs, qseq, qlen); memcpy(seqs + qlen, tseq, tlen);
	rc = bsa_edit_batch(ctx(__FUNCTION__), seqs, (size_t)qlen + tlen, &qoff, &qlen, &toff, &tlen, 1, &par, &out, cig, cap, off, &status);
	if(rc != BSA_OK){
		fprintf(stderr, " -- device alignment failed (%d: %s)", rc, bsa_last_error(g_ctx));
		die("", __FUNCTION__);
	}
	if(status & BSA_ST_BAD_BASE) die("base code > 3 in input", __FUNCTION__);
	if(status & BSA_ST_TRACE) die("traceback left the band", __FUNCTION__);
	memcpy(&rs, &out, sizeof(rs));
	deliver_cigars(cigars, mode, cig, off[1]);
	free(seqs); free(cig);
	return rs;
}

seqalign_result_t kmer_striped_seqedit_pairwise(u1i ksz, u1i *qseq, u4i qlen, u1i *tseq, u4i tlen, b1v *mempool, u4v *cigars, int verbose){
	seqalign_result_t rs;
	bsa_kmer_params_t par;
	bsa_result_t out;
	uint64_t qoff = 0, toff = qlen, off[2] = {0, 0};
	uint32_t status = 0, *cig;
	uint8_t *seqs;
	size_t cap = (size_t)qlen + tlen + 8;
	int rc;
	(void)verbose;
	memset(&rs, 0, sizeof(rs));
	check_mempool(mempool, __FUNCTION__);
	if(cigars) clear_u4v(cigars);                        /* bsalign.h:1435 */
	if(qlen == 0 || tlen == 0) return rs;                /* falls through to the global alignment's empty case, bsalign.h:1436-1438 */
	if(ksz == 0) die("k-mer size 0", __FUNCTION__);
	par.ksz = ksz; par.threads = 1;
	seqs = (uint8_t*)malloc((size_t)qlen + tlen + 1);
	cig = (uint32_t*)malloc(cap * sizeof(uint32_t));
	memcpy(seqs, qseq, qlen); memcpy(seqs + qlen, tseq, tlen);
	rc = bsa_kmer_edit_batch(ctx(__FUNCTION__), seqs, (size_t)qlen + tlen, &qoff, &qlen, &toff, &tlen, 1, &par, &out, cig, cap, off, &status);
	if(rc != BSA_OK){
		fprintf(stderr, " -- device alignment failed (%d: %s)", rc, bsa_last_error(g_ctx));
		die("", __FUNCTION__);
	}
	if(status & BSA_ST_BAD_BASE) die("base code > 3 in input", __FUNCTION__);
	if(status & BSA_ST_TRACE) die("traceback left the band", __FUNCTION__);
	memcpy(&rs, &out, sizeof(rs));
	deliver_cigars(cigars, 0, cig, off[1]);
	free(seqs); free(cig);
	return rs;
}

u4i seqalign_cigar2alnstr(u1i *qseq, u1i *tseq, seqalign_result_t *rs, u4v *cigars, char *alnstr[3], u4i length){ /* bsalign.h:531-582 */
	static const char codes[] = "ACGTN-";
	u4i z = 0, x, y, k, j, op, sz;
	if(alnstr == NULL) return 0;
	if(length == 0){
		length = (u4i)rs->aln;
		for(k = 0; k < 3; k++) alnstr[k] = (char*)realloc(alnstr[k], (size_t)length + 1);
	}
	x = (u4i)rs->qb; y = (u4i)rs->tb;
	for(k = 0; k < cigars->size && z < length; k++){
		op = cigars->buffer[k] & 0xf;
		sz = cigars->buffer[k] >> 4;
		if(sz > length - z) sz = length - z;
		for(j = 0; j < sz; j++, z++){
			if(op == 0 || op == 7 || op == 8){          /* match / mismatch column */
				alnstr[2][z] = (qseq[x] == tseq[y]) ? '|' : '*';
				alnstr[0][z] = codes[qseq[x++]];
				alnstr[1][z] = codes[tseq[y++]];
			} else if(op == 1 || op == 4){               /* query-only column */
				alnstr[2][z] = '-';
				alnstr[0][z] = codes[qseq[x++]];
				alnstr[1][z] = '-';
			} else if(op == 2 || op == 3){               /* target-only column */
				alnstr[2][z] = '-';
				alnstr[0][z] = '-';
				alnstr[1][z] = codes[tseq[y++]];
			} else { z--; }                              /* other ops emit nothing */
		}
	}
	alnstr[0][z] = alnstr[1][z] = alnstr[2][z] = 0;
	return z;
}
