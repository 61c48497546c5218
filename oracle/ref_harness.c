/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin extern "C"-style wrappers that call the *real* reference
 * (ruanjue/bsalign) functions so that tests and the golden-vector generator
 * can drive them through ctypes.  The reference sources are NOT copied: this
 * translation unit is compiled with -I$(REF) where REF=/root/reference and
 * the output goes to oracle/_ref/libbsref.so (git-ignored).
 *
 * Wrapped reference entry points (file:line in /root/reference):
 *   banded_striped_epi8_seqalign_pairwise        bsalign.h:3854
 *   banded_striped_epi8_seqalign_set_score_matrix bsalign.h:323
 *   striped_seqedit_pairwise                     bsalign.h:1046
 *   kmer_striped_seqedit_pairwise                bsalign.h:1209
 *   banded_striped_epi8_seqalign_piecex_row_init bsalign.h:2094
 *   banded_striped_epi8_seqalign_piecex_row_movx bsalign.h:2244
 *   banded_striped_epi8_seqalign_piecex_row_cal  bsalign.h:3181
 *   banded_striped_epi8_seqalign_piecex_row_merge bsalign.h:2474
 *   banded_striped_epi8_seqalign_row_max         bsalign.h:3213
 *   banded_striped_epi8_seqalign_set_query_prof[_hpc] bsalign.h:2166,2194
 */
#include "bsalign.h"
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

typedef struct {
	b1v *mempool;
	u4v *cigars;
} ref_ctx_t;

void *ref_ctx_create(void){
	ref_ctx_t *c = (ref_ctx_t*)malloc(sizeof(ref_ctx_t));
	c->mempool = adv_init_b1v(1024, 0, WORDSIZE, 0);
	c->cigars = init_u4v(64);
	return c;
}

void ref_ctx_destroy(void *vc){
	ref_ctx_t *c = (ref_ctx_t*)vc;
	free_b1v(c->mempool);
	free_u4v(c->cigars);
	free(c);
}

/* res: 10 ints (seqalign_result_t); cig: caller buffer of cap words; returns n cigar words (or -needed) */
long ref_align_pairwise(void *vc, const uint8_t *q, uint32_t qlen, const uint8_t *t, uint32_t tlen,
		int mode, uint32_t bandwidth, int M, int X, int O, int E, int Q, int P,
		int32_t *res, uint32_t *cig, long cap){
	ref_ctx_t *c = (ref_ctx_t*)vc;
	b1i mtx[16];
	seqalign_result_t rs;
	banded_striped_epi8_seqalign_set_score_matrix(mtx, M, X);
	clear_b1v(c->mempool);
	rs = banded_striped_epi8_seqalign_pairwise((u1i*)q, qlen, (u1i*)t, tlen, c->mempool, c->cigars, mode, bandwidth, mtx, O, E, Q, P, 0);
	memcpy(res, &rs, sizeof(rs));
	if((long)c->cigars->size > cap) return -(long)c->cigars->size;
	memcpy(cig, c->cigars->buffer, c->cigars->size * sizeof(u4i));
	return (long)c->cigars->size;
}

/* same with an explicit 4x4 matrix */
long ref_align_pairwise_mtx(void *vc, const uint8_t *q, uint32_t qlen, const uint8_t *t, uint32_t tlen,
		int mode, uint32_t bandwidth, const int8_t *mtx, int O, int E, int Q, int P,
		int32_t *res, uint32_t *cig, long cap){
	ref_ctx_t *c = (ref_ctx_t*)vc;
	seqalign_result_t rs;
	clear_b1v(c->mempool);
	rs = banded_striped_epi8_seqalign_pairwise((u1i*)q, qlen, (u1i*)t, tlen, c->mempool, c->cigars, mode, bandwidth, (b1i*)mtx, O, E, Q, P, 0);
	memcpy(res, &rs, sizeof(rs));
	if((long)c->cigars->size > cap) return -(long)c->cigars->size;
	memcpy(cig, c->cigars->buffer, c->cigars->size * sizeof(u4i));
	return (long)c->cigars->size;
}

long ref_edit_pairwise(void *vc, const uint8_t *q, uint32_t qlen, const uint8_t *t, uint32_t tlen,
		int mode, uint32_t bandwidth, int32_t *res, uint32_t *cig, long cap){
	ref_ctx_t *c = (ref_ctx_t*)vc;
	seqalign_result_t rs;
	clear_b1v(c->mempool);
	rs = striped_seqedit_pairwise((u1i*)q, qlen, (u1i*)t, tlen, mode, bandwidth, c->mempool, c->cigars, 0);
	memcpy(res, &rs, sizeof(rs));
	if((long)c->cigars->size > cap) return -(long)c->cigars->size;
	memcpy(cig, c->cigars->buffer, c->cigars->size * sizeof(u4i));
	return (long)c->cigars->size;
}

/* kmer_striped_seqedit_pairwise (bsalign.h:1209); the reference reverses its inputs in place and back, so work on copies */
long ref_kmer_edit_pairwise(void *vc, int ksz, const uint8_t *q, uint32_t qlen, const uint8_t *t, uint32_t tlen,
		int32_t *res, uint32_t *cig, long cap){
	ref_ctx_t *c = (ref_ctx_t*)vc;
	seqalign_result_t rs;
	u1i *qq = (u1i*)malloc(qlen + 16), *tt = (u1i*)malloc(tlen + 16);
	memset(qq, 0, qlen + 16); memset(tt, 0, tlen + 16);
	memcpy(qq, q, qlen); memcpy(tt, t, tlen);
	clear_b1v(c->mempool);
	rs = kmer_striped_seqedit_pairwise(ksz, qq, qlen, tt, tlen, c->mempool, c->cigars, 0);
	free(qq); free(tt);
	memcpy(res, &rs, sizeof(rs));
	if((long)c->cigars->size > cap) return -(long)c->cigars->size;
	memcpy(cig, c->cigars->buffer, c->cigars->size * sizeof(u4i));
	return (long)c->cigars->size;
}

/* repeat-timing helper for the CPU baseline (kind="reference"): returns seconds for n pairs laid out back to back */
double ref_align_batch_time(void *vc, const uint8_t *seqs, const uint64_t *qoff, const uint32_t *qlen,
		const uint64_t *toff, const uint32_t *tlen, long n, int mode, uint32_t bandwidth,
		int M, int X, int O, int E, int Q, int P, int64_t *score_sum){
	ref_ctx_t *c = (ref_ctx_t*)vc;
	b1i mtx[16];
	seqalign_result_t rs;
	struct timespec t0, t1;
	long k;
	int64_t ss = 0;
	banded_striped_epi8_seqalign_set_score_matrix(mtx, M, X);
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for(k=0;k<n;k++){
		clear_b1v(c->mempool);
		rs = banded_striped_epi8_seqalign_pairwise((u1i*)(seqs + qoff[k]), qlen[k], (u1i*)(seqs + toff[k]), tlen[k], c->mempool, c->cigars, mode, bandwidth, mtx, O, E, Q, P, 0);
		ss += rs.score + (int64_t)c->cigars->size;
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	if(score_sum) *score_sum = ss;
	return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}

double ref_edit_batch_time(void *vc, const uint8_t *seqs, const uint64_t *qoff, const uint32_t *qlen,
		const uint64_t *toff, const uint32_t *tlen, long n, int mode, uint32_t bandwidth, int64_t *score_sum){
	ref_ctx_t *c = (ref_ctx_t*)vc;
	seqalign_result_t rs;
	struct timespec t0, t1;
	long k;
	int64_t ss = 0;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for(k=0;k<n;k++){
		clear_b1v(c->mempool);
		rs = striped_seqedit_pairwise((u1i*)(seqs + qoff[k]), qlen[k], (u1i*)(seqs + toff[k]), tlen[k], mode, bandwidth, c->mempool, c->cigars, 0);
		ss += rs.score + (int64_t)c->cigars->size;
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	if(score_sum) *score_sum = ss;
	return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}

/* ---- row-level wrappers (POA P4 parity).  Rows use the reference's striped layout:
 * us/es/qs: bw bytes each (index = (p % W)*16 + p / W), ubegs: 17 int32.  All buffers 16B-aligned by the caller. */

int ref_get_piecewise(int O, int E, int Q, int P, int bandwidth){
	return banded_striped_epi8_seqalign_get_piecewise(O, E, Q, P, bandwidth);
}

void ref_row_init(int8_t *us, int8_t *es, int8_t *qs, int32_t *ubegs, int mode, uint32_t bandwidth,
		int max_nt, int min_nt, int O, int E, int Q, int P){
	int rbeg;
	banded_striped_epi8_seqalign_piecex_row_init((b1i*)us, (b1i*)es, (b1i*)qs, ubegs, &rbeg, mode, bandwidth, max_nt, min_nt, O, E, Q, P);
}

/* src = [1] (previous row), dst = [0] */
void ref_row_movx(int8_t *us_dst, int8_t *es_dst, int8_t *qs_dst, int32_t *ub_dst,
		int8_t *us_src, int8_t *es_src, int8_t *qs_src, int32_t *ub_src,
		uint32_t W, uint32_t movx, int piecewise, int nt_max, int nt_min, int O, int E, int Q, int P){
	b1i *us[2], *es[2], *qs[2];
	int *ubegs[2];
	us[0] = us_dst; us[1] = us_src;
	es[0] = es_dst; es[1] = es_src;
	qs[0] = qs_dst; qs[1] = qs_src;
	ubegs[0] = ub_dst; ubegs[1] = ub_src;
	banded_striped_epi8_seqalign_piecex_row_movx(us, es, qs, ubegs, W, movx, piecewise, nt_max, nt_min, O, E, Q, P);
}

/* src = [0] (moved previous row), dst = [1]; qprof built by ref_set_query_prof */
int ref_row_cal(uint32_t rbeg, uint8_t base, int8_t *us_src, int8_t *es_src, int8_t *qs_src, int32_t *ub_src,
		int8_t *us_dst, int8_t *es_dst, int8_t *qs_dst, int32_t *ub_dst,
		int8_t *qprof, int O, int E, int Q, int P, uint32_t W, uint32_t mov, int rh, int piecewise){
	b1i *us[2], *es[2], *qs[2];
	int *ubegs[2];
	us[0] = us_src; us[1] = us_dst;
	es[0] = es_src; es[1] = es_dst;
	qs[0] = qs_src; qs[1] = qs_dst;
	ubegs[0] = ub_src; ubegs[1] = ub_dst;
	return banded_striped_epi8_seqalign_piecex_row_cal(rbeg, base, us, es, qs, ubegs, (b1i*)qprof, O, E, Q, P, W, mov, rh, piecewise);
}

void ref_row_merge(int8_t *us0, int8_t *es0, int8_t *qs0, int32_t *ub0,
		int8_t *us1, int8_t *es1, int8_t *qs1, int32_t *ub1,
		int8_t *us2, int8_t *es2, int8_t *qs2, int32_t *ub2, uint32_t W, int piecewise){
	b1i *us[3], *es[3], *qs[3];
	int *ubegs[3];
	us[0] = us0; us[1] = us1; us[2] = us2;
	es[0] = es0; es[1] = es1; es[2] = es2;
	qs[0] = qs0; qs[1] = qs1; qs[2] = qs2;
	ubegs[0] = ub0; ubegs[1] = ub1; ubegs[2] = ub2;
	banded_striped_epi8_seqalign_piecex_row_merge(us, es, qs, ubegs, W, piecewise);
}

uint32_t ref_row_max(int8_t *us, int32_t *ubegs, uint32_t W, int32_t *max_score){
	int ms;
	u4i x = banded_striped_epi8_seqalign_row_max((b1i*)us, ubegs, W, &ms);
	*max_score = ms;
	return x;
}

int ref_getscore(int8_t *us, int32_t *ubegs, uint32_t W, uint64_t pos){
	return banded_striped_epi8_seqalign_getscore((b1i*)us, ubegs, W, pos);
}

int ref_band_mov(int8_t *us, int32_t *ubegs, uint32_t W, uint32_t tidx, uint32_t qoff, uint32_t qlen){
	return banded_striped_epi8_seqalign_band_mov((b1i*)us, ubegs, W, tidx, qoff, qlen);
}

uint64_t ref_qprof_size(uint32_t qlen, uint32_t bandwidth){
	return banded_striped_epi8_seqalign_qprof_size(qlen, bandwidth);
}

void ref_set_query_prof(const uint8_t *q, uint32_t qlen, int8_t *qprof, uint32_t bandwidth, const int8_t *mtx){
	banded_striped_epi8_seqalign_set_query_prof((u1i*)q, qlen, (b1i*)qprof, bandwidth, (b1i*)mtx);
}

void ref_set_query_prof_hpc(const uint8_t *q, uint32_t qlen, int8_t *qprof, uint32_t bandwidth, const int8_t *mtx, int bonus){
	banded_striped_epi8_seqalign_set_query_prof_hpc((u1i*)q, qlen, (b1i*)qprof, bandwidth, (b1i*)mtx, bonus);
}
