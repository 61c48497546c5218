/*
 * bsalign_poa_diagdp.h -- reference-side binding of the MSA refinement's DP (libbsalign_hip: bsa_diagdp_batch).
 *
 * Compiled INSIDE the reference's tree: patches/bspoa_device_diagdp.diff includes it in front of remsa_pedits_bspoa
 * (bspoa.h:4178) and adds one field to BSPOA (devdiag).  With g->devdiag set, remsa_pedits_bspoa calls
 * bsa_poa_diagdp_window() once per window, before its loop over the reads: the planes of EVERY read of the window are
 * built here the way the loop builds them one read at a time (bspoa.h:4424-4440: the read's bases at their MSA
 * positions, the homopolymer counter behind each base; reading the nodes only -- the loop still cuts, traces and
 * reconnects every read in the reference's order), the window's shared planes (consensus, column profile,
 * bspoa.h:4236-4247, 4339-4345) are copied, and ONE bsa_diagdp_batch call fills the two difference planes of all reads
 * (maxmat_dp_diag_rowcal, bspoa.h:3856-3896).  The loop then hands remsa_pedit_rd_bspoacore (bspoa.h:3916) the read's
 * planes with `filled` set, which skips the fill (bspoa.h:3925-3935) and goes straight to the traceback.
 * Reads that were not part of the graph yet (rid >= nrds, the `all` pass) keep the host fill.
 *
 * Round 4, the walk as well (bsa_poa_diagdp_init_walk: bsa_diagdp_walk_batch): the planes never leave the device -- 2.1 bytes a cell, the
 * reason a host-pointer fill could not take more than a window per call -- and the traceback loop of remsa_pedit_rd_bspoacore
 * (bspoa.h:3965-4040) takes its decisions from the device's two bits a step (bsa_poa_diagdp_steps) while it does the graph work itself
 * (merge_nodes_bspoa at every diagonal step over a base).
 */
#ifndef BSALIGN_POA_DIAGDP_H
#define BSALIGN_POA_DIAGDP_H

#include "bsalign_hip.h"

typedef int (*bsa_poa_diagdp_fn)(void *user, const uint8_t *planes, size_t planes_bytes, const bsa_diagdp_prob_t *probs, size_t n,
		uint8_t *matrix, size_t matrix_bytes);

typedef int (*bsa_poa_diagdp_walk_fn)(void *user, const uint8_t *planes, size_t planes_bytes, const bsa_diagdp_prob_t *probs, size_t n,
		bsa_diagdp_walk_t *walks, uint32_t *steps, size_t steps_cap_words);

typedef struct {
	bsa_poa_diagdp_fn run;           /* bsa_diagdp_batch with user = bsa_ctx_t* */
	bsa_poa_diagdp_walk_fn run_walk; /* bsa_diagdp_walk_batch: fill and traceback on the device (run is then unused) */
	bsa_diagdp_walk_t *walks; uint32_t *words; size_t words_cap;
	uint32_t *slot;                  /* [nseq]: index of read rid among the problems of the call */
	void *user;
	uint8_t *planes, *matrix;        /* grown as needed, kept between windows */
	size_t planes_cap, matrix_cap;
	bsa_diagdp_prob_t *probs;
	uint8_t **ptrs;                  /* [2 * nseq]: the two planes of every read inside `matrix` (NULL: host fill) */
	size_t cap;
	size_t budget;                   /* bytes the matrices of one window may take on the host and on the device (0: BSA_POA_DIAGDP_BUDGET, 1 GiB);
	                                    the reads that do not fit keep the host fill */
	/* statistics */
	uint64_t calls, reads, steps;
} bsa_poa_diagdp_t;

static inline int bsa_poa_diagdp_hip(void *user, const uint8_t *planes, size_t planes_bytes, const bsa_diagdp_prob_t *probs, size_t n,
		uint8_t *matrix, size_t matrix_bytes){
	return bsa_diagdp_batch((bsa_ctx_t*)user, planes, planes_bytes, probs, n, matrix, matrix_bytes);
}
static inline int bsa_poa_diagdp_walk_hip(void *user, const uint8_t *planes, size_t planes_bytes, const bsa_diagdp_prob_t *probs, size_t n,
		bsa_diagdp_walk_t *walks, uint32_t *steps, size_t steps_cap_words){
	return bsa_diagdp_walk_batch((bsa_ctx_t*)user, planes, planes_bytes, probs, n, walks, steps, steps_cap_words);
}
static inline void bsa_poa_diagdp_init(bsa_poa_diagdp_t *dd, bsa_poa_diagdp_fn run, void *user){ memset(dd, 0, sizeof(*dd)); dd->run = run; dd->user = user; }
static inline void bsa_poa_diagdp_init_walk(bsa_poa_diagdp_t *dd, bsa_poa_diagdp_walk_fn run_walk, void *user){ memset(dd, 0, sizeof(*dd)); dd->run_walk = run_walk; dd->user = user; }
static inline void bsa_poa_diagdp_free(bsa_poa_diagdp_t *dd){ free(dd->planes); free(dd->matrix); free(dd->probs); free(dd->ptrs); free(dd->walks); free(dd->words); free(dd->slot); memset(dd, 0, sizeof(*dd)); }

/* walk mode: the steps of read rid (two bits each, sixteen a word), their number and the score remsa_pedit_rd_bspoacore returns; NULL in fill mode */
static inline const uint32_t* bsa_poa_diagdp_steps(bsa_poa_diagdp_t *dd, u4i rid, int mbeg, int mend, u4i *nsteps, int *score){
	const bsa_diagdp_walk_t *w;
	if(dd == NULL || dd->run_walk == NULL) return NULL;
	w = dd->walks + dd->slot[rid];
	if((int)dd->probs[dd->slot[rid]].mbeg != mbeg || (int)dd->probs[dd->slot[rid]].mend != mend || w->status != 0){
		fflush(stdout); fprintf(stderr, " -- device walk of read %u does not fit the loop's range (%d, %d) or failed (status %u) in %s -- %s:%d --\n", rid, mbeg, mend, w->status, __FUNCTION__, __FILE__, __LINE__); fflush(stderr);
		abort();
	}
	*nsteps = w->nsteps; *score = w->score;
	return dd->words + w->first_word;
}

/* seq1 / mats1: the window's shared planes as remsa_pedits_bspoa holds them (logical index 0, bandwidth / 2 bytes of
 * padding in front).  Returns dd->ptrs, or NULL when the band is not one the device kernel takes (the caller then fills
 * on the host, as without the patch). */
static inline uint8_t** bsa_poa_diagdp_window(BSPOA *g, bsa_poa_diagdp_t *dd, u1i *seq1, u1i *mats1[4], u4i mlen, u4i bandwidth, u4i nseq){
	const u4i W = bandwidth / WORDSIZE, HW = bandwidth / 2;
	const size_t PS = roundup_times(mlen + bandwidth, WORDSIZE);                       /* one plane */
	const size_t MS = roundup_times((size_t)(2 * mlen + 1) * (bandwidth + 2), WORDSIZE);    /* one matrix plane */
	size_t need_p, need_m, n = 0, slack, budget, nfit;
	uint8_t stale[4 * WORDSIZE];
	u4i rid, rdlen, b, i;
	bspoanode_t *v;
	if(!(W == 1 || W == 2 || W == 4) || nseq == 0 || mlen == 0) return NULL;
	/* The reference fills one read's matrix at a time; here the matrices of all reads of the window exist at once, so a deep or long
	 * window is cut at a byte budget: the first reads that fit go to the device, the others keep the host fill (ptrs stay NULL). */
	budget = dd->budget;
	if(budget == 0){ const char *e = getenv("BSA_POA_DIAGDP_BUDGET"); budget = e? (size_t)strtoull(e, NULL, 10) : 0; if(budget == 0) budget = dd->run_walk? (size_t)64 << 30 : (size_t)1 << 30; }      /* (walk mode: device memory only) */
	nfit = budget / (MS * 2);
	if(nfit > nseq) nfit = nseq;
	if(nfit == 0) return NULL;
	need_p = PS * 5 * (nfit + 1);
	need_m = dd->run_walk? 16 : MS * 2 * nfit;
	if(need_p > dd->planes_cap){ free(dd->planes); dd->planes = (uint8_t*)malloc(need_p); dd->planes_cap = dd->planes? need_p : 0; }
	if(need_m > dd->matrix_cap){ free(dd->matrix); dd->matrix = (uint8_t*)malloc(need_m); dd->matrix_cap = dd->matrix? need_m : 0; }
	if(nseq > dd->cap){
		free(dd->probs); free(dd->ptrs);
		dd->probs = (bsa_diagdp_prob_t*)malloc(sizeof(bsa_diagdp_prob_t) * nseq);
		dd->ptrs = (uint8_t**)malloc(sizeof(uint8_t*) * 2 * nseq);
		free(dd->walks); free(dd->slot);
		dd->walks = (bsa_diagdp_walk_t*)malloc(sizeof(bsa_diagdp_walk_t) * nseq);
		dd->slot = (uint32_t*)malloc(sizeof(uint32_t) * nseq);
		dd->cap = (dd->probs && dd->ptrs && dd->walks && dd->slot)? nseq : 0;
	}
	if(dd->planes == NULL || dd->matrix == NULL || dd->probs == NULL || dd->ptrs == NULL || dd->walks == NULL || dd->slot == NULL) return NULL;        /* no memory for the batch form: the host fill, as without the patch */
	memcpy(dd->planes, seq1 - HW, PS);
	for(b=0;b<4;b++) memcpy(dd->planes + PS * (1 + b), mats1[b] - HW, PS);
	/* The reference clears its four mats[0] planes with ONE memset of 4 * (mlen + bandwidth) bytes (bspoa.h:4349) although the
	 * planes lie roundup(mlen + bandwidth, 16) bytes apart: the last 4 * slack bytes of plane 3 are never cleared, and what an
	 * earlier read of the loop wrote there (counts behind T bases in the last columns) is still there for the later ones.  The
	 * DP reads those bytes, so they are carried from read to read here exactly as they survive there. */
	slack = PS - (mlen + bandwidth);
	memset(stale, 0, sizeof(stale));
	for(rid=0;rid<nseq;rid++){
		uint8_t *rp = dd->planes + PS * 5 * ((size_t)rid + 1);
		bsa_diagdp_prob_t *pb;
		u1i lc = 4, cc = 0;
		dd->ptrs[2 * rid] = dd->ptrs[2 * rid + 1] = NULL;
		rdlen = g->seqs->rdlens->buffer[rid];
		if(rdlen == 0) continue;
		if(rid >= nfit){
			/* over the budget: host fill.  What survives in the tail of plane 3 from read to read (see above) is only carried for
			 * the reads built here; the host fill of the others sees its own planes, exactly as without the patch. */
			continue;
		}
		memset(rp, 4, PS);
		memset(rp + PS, 0, 4 * PS);
		memcpy(rp + PS * 5 - 4 * slack, stale, 4 * slack);
		for(i=rdlen;i>0;i--){                                  /* bspoa.h:4426-4436, without the cut */
			v = get_rdnode_bspoa(g, rid, i - 1);
			rp[HW + v->mpos] = v->base;
			if(v->base == lc){
				if(cc < MAX_U1) cc ++;
				rp[PS * (1 + v->base) + HW + v->mpos] = cc;
			} else {
				lc = v->base;
				cc = 0;
			}
		}
		memcpy(stale, rp + PS * 5 - 4 * slack, 4 * slack);
		pb = dd->probs + n;
		pb->seq0 = PS * 5 * ((size_t)rid + 1) + HW;
		pb->seq1 = HW;
		for(b=0;b<4;b++){ pb->mats0[b] = PS * 5 * ((size_t)rid + 1) + PS * (1 + b) + HW; pb->mats1[b] = PS * (1 + b) + HW; }
		pb->out0 = dd->run_walk? 0 : MS * 2 * (size_t)rid; pb->out1 = dd->run_walk? 0 : pb->out0 + MS;
		dd->slot[rid] = (uint32_t)n;
		pb->mlen = mlen; pb->W = W;
		pb->mbeg = get_rdnode_bspoa(g, rid, 0)->mpos;
		pb->mend = get_rdnode_bspoa(g, rid, rdlen - 1)->mpos + 1;
		dd->ptrs[2 * rid] = dd->matrix + pb->out0; dd->ptrs[2 * rid + 1] = dd->matrix + pb->out1;
		dd->steps += 2 * (uint64_t)(pb->mend - pb->mbeg) - 1;
		n ++;
	}
	if(n && dd->run_walk){
		size_t words = 0, k;
		for(k=0;k<n;k++) words += (size_t)bsa_diagdp_walk_words(dd->probs[k].mbeg, dd->probs[k].mend);
		if(words > dd->words_cap){ free(dd->words); dd->words = (uint32_t*)malloc(sizeof(uint32_t) * words); dd->words_cap = dd->words? words : 0; }
		if(dd->words == NULL || dd->run_walk(dd->user, dd->planes, need_p, dd->probs, n, dd->walks, dd->words, dd->words_cap) != 0){
			for(rid=0;rid<nseq;rid++) dd->ptrs[2 * rid] = dd->ptrs[2 * rid + 1] = NULL;
			return NULL;
		}
	} else
	if(n && dd->run(dd->user, dd->planes, need_p, dd->probs, n, dd->matrix, need_m) != 0){
		/* device out of memory or a failed call: nothing was filled -- every read takes the host fill */
		for(rid=0;rid<nseq;rid++) dd->ptrs[2 * rid] = dd->ptrs[2 * rid + 1] = NULL;
		return NULL;
	}
	dd->calls ++; dd->reads += n;
	return dd->ptrs;
}

#endif
