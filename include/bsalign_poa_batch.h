/*
 * bsalign_poa_batch.h -- many POA windows at once, reference side.
 *
 * For a tree that has the reference's bspoa.h with patches/bspoa_device_sweep.diff applied (the patch adds one field,
 * BSPOA.devsweep, and makes align_rd_bspoa call include/bsalign_poa_adapter.h's bsa_poa_align_rd_core() when it is set).
 * Include it after bspoa.h; link libbsalign_hip.so and pthread.
 *
 *   beg_bspoa(g); push_bspoa(g, seq, len); ...        for every window, as before (bspoa.h:1775, 961)
 *   bsa_poa_end_many(gs, n, ctx);                      instead of n calls of end_bspoa (bspoa.h:4722)
 *   gs[k]->cns / qlt / alt / msacols                   as before
 *
 * One POA is sequential in its reads, windows are independent: every window runs the reference's
 * own end_bspoa on a host thread of its own, and wherever that would sweep the graph (align_rd_bspoacore,
 * bspoa.h:2515-2618) the program goes to the batcher of libbsalign_hip (bsa_sweep_batcher_submit), which runs read r of
 * ALL windows as one device launch.  bsa_poa_end_one() is the single-window form (bsa_sweep_host behind it).
 */
#ifndef BSALIGN_POA_BATCH_H
#define BSALIGN_POA_BATCH_H

#include <pthread.h>
#include "bsalign_hip.h"
#include "bsalign_poa_adapter.h"

#ifndef BSA_POA_WAVE
#define BSA_POA_WAVE 1024
#endif

typedef struct {
	BSPOA *g;
	bsa_sweep_batcher_t *batcher;
} bsa_poa_many_job_t;

static inline void bsa_poa_end_one(BSPOA *g, bsa_ctx_t *ctx);

static void *bsa_poa_many_thread(void *vp){
	bsa_poa_many_job_t *j = (bsa_poa_many_job_t*)vp;
	bsa_poa_adapter_t ad;
	bsa_sweep_batcher_enter(j->batcher);                 /* (runs while it holds one of the host slots) */
	bsa_poa_adapter_init_graph(&ad, bsa_poa_batcher_submit_graph, bsa_sweep_batcher_submit, j->batcher);
	j->g->devsweep = &ad;
	end_bspoa(j->g);
	j->g->devsweep = NULL;
	bsa_sweep_batcher_leave(j->batcher);                 /* this window submits nothing more */
	bsa_poa_adapter_free(&ad);
	return NULL;
}

/* end_bspoa for n windows in lock-step on the device.  Returns 0 or a BSA_E_* code (nothing was run then). */
static inline int bsa_poa_end_many(BSPOA **gs, int n, bsa_ctx_t *ctx){
	bsa_sweep_batcher_t *b;
	bsa_poa_many_job_t *jobs;
	pthread_t *th;
	char *up;
	int k, rc, w0, wn, started;
	if(n <= 0) return BSA_OK;
	jobs = (bsa_poa_many_job_t*)calloc((size_t)n, sizeof(bsa_poa_many_job_t));
	th = (pthread_t*)calloc((size_t)n, sizeof(pthread_t));
	up = (char*)calloc((size_t)n, 1);
	if(jobs == NULL || th == NULL || up == NULL){ free(jobs); free(th); free(up); return BSA_E_NOMEM; }
	cal_permutation_bspoa(MAX_LOG_CACHE, 0);             /* the reference fills this table lazily (bspoa.h:3391-3401): do it before any thread reads it */
	/* one host thread per window, at most BSA_POA_WAVE windows (threads) alive at once: a polisher hands over thousands */
	rc = BSA_OK;
	for(w0=0;w0<n && rc==BSA_OK;w0+=BSA_POA_WAVE){
		wn = (n - w0 < BSA_POA_WAVE)? n - w0 : BSA_POA_WAVE;
		b = NULL;
		rc = bsa_sweep_batcher_create(ctx, (uint32_t)wn, &b);
		if(rc != BSA_OK) break;
		started = 0;
		for(k=w0;k<w0+wn;k++){
			jobs[k].g = gs[k]; jobs[k].batcher = b;
			up[k] = (pthread_create(&th[k], NULL, bsa_poa_many_thread, &jobs[k]) == 0);
			if(up[k]) started ++;
			else bsa_sweep_batcher_leave(b);             /* nobody will submit for this window: the others must not wait for it */
		}
		for(k=w0;k<w0+wn;k++) if(up[k]) pthread_join(th[k], NULL);
		bsa_sweep_batcher_destroy(b);
		/* windows whose thread could not be started run here, one by one */
		for(k=w0;k<w0+wn;k++) if(!up[k]) bsa_poa_end_one(gs[k], ctx);
	}
	free(jobs); free(th); free(up);
	return rc;
}

/* end_bspoa of one window with its sweeps on the device */
static inline void bsa_poa_end_one(BSPOA *g, bsa_ctx_t *ctx){
	bsa_poa_adapter_t ad;
	bsa_poa_adapter_init_graph(&ad, bsa_poa_graph_backend_hip, bsa_poa_backend_hip, ctx);
	g->devsweep = &ad;
	end_bspoa(g);
	g->devsweep = NULL;
	bsa_poa_adapter_free(&ad);
}

#endif
