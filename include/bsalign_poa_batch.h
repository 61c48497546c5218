/*
 * bsalign_poa_batch.h -- many POA windows at once, reference side.
 *
 * For a tree that has the reference's bspoa.h with patches/bspoa_device_sweep.diff applied (the patch adds one field,
 * BSPOA.devsweep, and makes align_rd_bspoa go through include/bsalign_poa_adapter.h when it is set: bsa_poa_align_rd_pog() -- node
 * selection, band placement, the DP, the walk and the graph surgery on libbsalign_hip's own POA graph, include/bsalign_poa.h -- and,
 * for a read that declines, bsa_poa_align_rd_core() on the reference's own graph).
 * Include it after bspoa.h; link libbsalign_hip.so and pthread.
 *
 *   beg_bspoa(g); push_bspoa(g, seq, len); ...        for every window, as before (bspoa.h:1775, 961)
 *   bsa_poa_end_many(gs, n, ctx);                      instead of n calls of end_bspoa (bspoa.h:4722)
 *   gs[k]->cns / qlt / alt / msacols                   as before
 *
 * One POA is sequential in its reads, windows are independent: a pool of host threads runs the reference's
 * own end_bspoa window after window, and wherever that would sweep the graph (align_rd_bspoacore, bspoa.h:2515-2618) the program goes
 * to the batcher of libbsalign_hip (bsa_poa_batcher_submit_graph / bsa_sweep_batcher_submit), whose dispatcher thread runs whatever
 * programs are pending as one device launch.  bsa_poa_end_one() is the single-window form (bsa_poa_graph_host / bsa_sweep_host behind it).
 */
#ifndef BSALIGN_POA_BATCH_H
#define BSALIGN_POA_BATCH_H

#include <pthread.h>
#include <unistd.h>
#include "bsalign_hip.h"
#include "bsalign_poa_adapter.h"

#ifndef BSA_POA_WAVE
#define BSA_POA_WAVE 1024
#endif
#ifndef BSA_POA_POOL
#define BSA_POA_POOL 256         /* window threads alive at once: the batcher runs whatever is pending, so a pool that takes the windows one after the other keeps the host's working set to this many graphs */
#endif

/* WHEN the device is attached.  One POA is sequential in its reads and a lone wave of the device runs a read's DP several times slower than a host
 * core (HISTORY.md section 4b); the device pays by running many windows at once.  Measured on the MI355X box (16 host threads, windows of 12 reads x
 * 1.5 kbp, tests/test_poa_pog_gpu.py::test_where_many_windows_start_to_pay): 1 window 0.042 s on the device path against 0.024 s for the reference,
 * 16: 0.074 / 0.030, 32: 0.086 / 0.067, 64: 0.117 / 0.123, 128: 0.183 / 0.241.  Below BSA_POA_MIN_WINDOWS windows in flight this binding therefore
 * leaves BSPOA.devsweep NULL -- the reference's own end_bspoa on host threads, which is this header's caller's code, not a path of libbsalign_hip --
 * and a single window (C4 as BASELINE states it: 2.8 s on the host, 3.9 s with every read's DP on a lone wave) is never slower than before the patch.
 * $BSA_POA_MIN_WINDOWS overrides (1 = always the device). */
#ifndef BSA_POA_MIN_WINDOWS
#define BSA_POA_MIN_WINDOWS 64
#endif
static inline int bsa_poa_min_windows(void){
	const char *e = getenv("BSA_POA_MIN_WINDOWS");
	const int v = e ? atoi(e) : BSA_POA_MIN_WINDOWS;
	return v < 1 ? 1 : v;
}

typedef struct {
	BSPOA **gs;
	int n, t, nt;                                        /* this worker takes windows t, t + nt, ... */
	bsa_sweep_batcher_t *batcher;
} bsa_poa_many_job_t;

static void *bsa_poa_host_thread(void *vp){
	bsa_poa_many_job_t *j = (bsa_poa_many_job_t*)vp;
	int k;
	for(k=j->t;k<j->n;k+=j->nt) end_bspoa(j->gs[k]);        /* the reference, untouched (devsweep NULL) */
	return NULL;
}

/* too few windows for the device: the reference's end_bspoa on as many host threads as the process may run */
static inline int bsa_poa_end_on_host(BSPOA **gs, int n){
	long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
	int nt, k;
	bsa_poa_many_job_t *jobs; pthread_t *th; char *up;
	const char *ht = getenv("BSA_POA_HOST_THREADS");
	if(ht && atoi(ht) > 0) ncpu = atoi(ht);
	nt = (int)((ncpu < 1) ? 1 : (ncpu > n ? n : ncpu));
	if(nt <= 1){ for(k=0;k<n;k++) end_bspoa(gs[k]); return BSA_OK; }
	jobs = (bsa_poa_many_job_t*)calloc((size_t)nt, sizeof(bsa_poa_many_job_t)); th = (pthread_t*)calloc((size_t)nt, sizeof(pthread_t)); up = (char*)calloc((size_t)nt, 1);
	if(jobs == NULL || th == NULL || up == NULL){ free(jobs); free(th); free(up); for(k=0;k<n;k++) end_bspoa(gs[k]); return BSA_OK; }
	cal_permutation_bspoa(MAX_LOG_CACHE, 0);             /* (filled lazily by the reference, bspoa.h:3391-3401: before any thread reads it) */
	for(k=0;k<nt;k++){ jobs[k].gs = gs; jobs[k].n = n; jobs[k].t = k; jobs[k].nt = nt; up[k] = (pthread_create(&th[k], NULL, bsa_poa_host_thread, &jobs[k]) == 0); }
	for(k=0;k<nt;k++){ if(up[k]) pthread_join(th[k], NULL); else bsa_poa_host_thread(&jobs[k]); }
	free(jobs); free(th); free(up);
	return BSA_OK;
}

static inline void bsa_poa_end_one(BSPOA *g, bsa_ctx_t *ctx);

/* a worker: one window after the other.  The batcher runs whatever programs are pending whenever the device side is free (no lock-step since
 * round 4), so the windows need not be alive together: BSA_POA_POOL workers keep that many graphs in the host's caches instead of all of them. */
static void *bsa_poa_many_thread(void *vp){
	bsa_poa_many_job_t *j = (bsa_poa_many_job_t*)vp;
	bsa_poa_adapter_t ad;
	int k;
	bsa_poa_adapter_init_graph(&ad, bsa_poa_batcher_submit_graph, bsa_sweep_batcher_submit, j->batcher);
	bsa_poa_adapter_use_pog(&ad, 1);                     /* node selection, band placement, program building and graph surgery on the library's own graph (bsalign_poa.h) */
	for(k=j->t;k<j->n;k+=j->nt){
		bsa_sweep_batcher_enter(j->batcher);             /* (runs while it holds one of the host slots) */
		j->gs[k]->devsweep = &ad;
		end_bspoa(j->gs[k]);
		j->gs[k]->devsweep = NULL;
		bsa_sweep_batcher_leave(j->batcher);             /* gives the slot back; in lock-step mode: this participant submits nothing more */
	}
	bsa_poa_adapter_free(&ad);
	return NULL;
}

/* end_bspoa for n windows with their sweeps and walks on the device, batched.  Returns 0 or a BSA_E_* code (nothing was run then). */
static inline int bsa_poa_end_many(BSPOA **gs, int n, bsa_ctx_t *ctx){
	bsa_sweep_batcher_t *b = NULL;
	bsa_poa_many_job_t *jobs;
	pthread_t *th;
	char *up;
	const char *le = getenv("BSA_POA_BATCH_MIN");
	int k, rc, nt, started = 0;
	if(n <= 0) return BSA_OK;
	if(n < bsa_poa_min_windows()) return bsa_poa_end_on_host(gs, n);       /* too few windows in flight for the device to pay (see BSA_POA_MIN_WINDOWS above) */
	if(le && strcmp(le, "all") == 0 && n > BSA_POA_WAVE){        /* lock-step: at most BSA_POA_WAVE windows (threads) at a time */
		for(k=0;k<n;k+=BSA_POA_WAVE){ rc = bsa_poa_end_many(gs + k, (n - k < BSA_POA_WAVE)? n - k : BSA_POA_WAVE, ctx); if(rc != BSA_OK) return rc; }
		return BSA_OK;
	}
	/* lock-step (BSA_POA_BATCH_MIN=all) needs every window alive: a thread each, as in rounds 2-3 */
	nt = (le && strcmp(le, "all") == 0) ? ((n < BSA_POA_WAVE)? n : BSA_POA_WAVE) : ((n < BSA_POA_POOL)? n : BSA_POA_POOL);
	jobs = (bsa_poa_many_job_t*)calloc((size_t)nt, sizeof(bsa_poa_many_job_t));
	th = (pthread_t*)calloc((size_t)nt, sizeof(pthread_t));
	up = (char*)calloc((size_t)nt, 1);
	if(jobs == NULL || th == NULL || up == NULL){ free(jobs); free(th); free(up); return BSA_E_NOMEM; }
	cal_permutation_bspoa(MAX_LOG_CACHE, 0);             /* the reference fills this table lazily (bspoa.h:3391-3401): do it before any thread reads it */
	rc = bsa_sweep_batcher_create(ctx, (uint32_t)n, &b);
	if(rc != BSA_OK){ free(jobs); free(th); free(up); return rc; }
	for(k=0;k<nt;k++){
		jobs[k].gs = gs; jobs[k].n = n; jobs[k].t = k; jobs[k].nt = nt; jobs[k].batcher = b;
		up[k] = (pthread_create(&th[k], NULL, bsa_poa_many_thread, &jobs[k]) == 0);
		if(up[k]) started ++;
		else {
			/* a worker that could not be started: its windows leave the batcher NOW -- in lock-step mode the dispatcher waits for every
			 * participant that has not left, and the started workers would block in submit for ever if this waited for their join */
			int w;
			for(w=k;w<n;w+=nt) bsa_sweep_batcher_leave(b);
		}
	}
	for(k=0;k<nt;k++) if(up[k]) pthread_join(th[k], NULL);
	bsa_sweep_batcher_destroy(b);
	/* the windows of such a worker run here, one by one */
	for(k=0;k<nt;k++) if(!up[k]){
		int w;
		for(w=k;w<n;w+=nt) bsa_poa_end_one(gs[w], ctx);
	}
	free(jobs); free(th); free(up);
	(void)started;                                       /* every window was finished either way (with no worker at all: one by one, at the single-window rate) */
	return BSA_OK;
}

/* end_bspoa of one window: on the device only when a single window is what BSA_POA_MIN_WINDOWS allows (the default does not: one window is faster on the host) */
static inline void bsa_poa_end_one(BSPOA *g, bsa_ctx_t *ctx){
	bsa_poa_adapter_t ad;
	if(1 < bsa_poa_min_windows()){ end_bspoa(g); return; }
	bsa_poa_adapter_init_graph(&ad, bsa_poa_graph_backend_hip, bsa_poa_backend_hip, ctx);
	bsa_poa_adapter_use_pog(&ad, 1);
	g->devsweep = &ad;
	end_bspoa(g);
	g->devsweep = NULL;
	bsa_poa_adapter_free(&ad);
}

#endif
