/*
 * ref_poa_patched.c -- TEST INFRASTRUCTURE ONLY.  The reference's bspoa.h WITH patches/bspoa_device_sweep.diff applied
 * (oracle/Makefile patches a temporary copy outside the repository; nothing of the reference is kept) plus
 * include/bsalign_poa_batch.h: the real beg_bspoa / push_bspoa / end_bspoa surface with the per-read sweep on the
 * device -- what a maintainer who applies the patch gets.  The four libbsalign_hip entry points the headers need are
 * resolved at run time from addresses the GPU test hands in (this library must load without HIP).
 */
#include <stdint.h>
#include "bsalign_hip.h"

typedef int (*fn_sweep_host)(bsa_ctx_t*, const bsa_row_task_t*, size_t, const bsa_sweep_prog_t*, size_t, const uint8_t*, const uint64_t*,
		const uint32_t*, size_t, const bsa_sweep_params_t*, uint8_t*, size_t, bsa_sweep_result_t*);
typedef int (*fn_bcreate)(bsa_ctx_t*, uint32_t, bsa_sweep_batcher_t**);
typedef void (*fn_bvoid)(bsa_sweep_batcher_t*);
typedef int (*fn_bsubmit)(void*, const bsa_row_task_t*, size_t, const uint8_t*, uint32_t, const bsa_sweep_params_t*, uint8_t*, size_t, bsa_sweep_result_t*);
static fn_sweep_host p_sweep_host; static fn_bcreate p_bcreate; static fn_bvoid p_bdestroy, p_bleave; static fn_bsubmit p_bsubmit;
static bsa_ctx_t *p_ctx;
typedef int (*fn_diagdp)(bsa_ctx_t*, const uint8_t*, size_t, const bsa_diagdp_prob_t*, size_t, uint8_t*, size_t);
static fn_diagdp p_diagdp;
void refp_attach_diagdp(void *diagdp){ p_diagdp = (fn_diagdp)diagdp; }
typedef int (*fn_diagdp_walk)(bsa_ctx_t*, const uint8_t*, size_t, const bsa_diagdp_prob_t*, size_t, bsa_diagdp_walk_t*, uint32_t*, size_t);
static fn_diagdp_walk p_diagdp_walk;
void refp_attach_diagdp_walk(void *f){ p_diagdp_walk = (fn_diagdp_walk)f; }
void refp_attach(void *ctx, void *sweep_host, void *bcreate, void *bdestroy, void *bsubmit, void *bleave){
	p_ctx = (bsa_ctx_t*)ctx; p_sweep_host = (fn_sweep_host)sweep_host; p_bcreate = (fn_bcreate)bcreate; p_bdestroy = (fn_bvoid)bdestroy;
	p_bsubmit = (fn_bsubmit)bsubmit; p_bleave = (fn_bvoid)bleave;
}
typedef int (*fn_graph_host)(bsa_ctx_t*, const bsa_poa_node_t*, size_t, const bsa_poa_edge_t*, size_t, const bsa_poa_cand_t*, size_t, const bsa_poa_prog_t*, size_t,
		const uint8_t*, size_t, const bsa_sweep_params_t*, bsa_poa_result_t*, bsa_poa_event_t*, size_t, bsa_poa_cell_t*, int32_t*);
typedef int (*fn_bsubmit_graph)(void*, const bsa_poa_node_t*, size_t, const bsa_poa_edge_t*, size_t, const bsa_poa_cand_t*, size_t, const uint8_t*, uint32_t,
		const bsa_sweep_params_t*, bsa_poa_result_t*, bsa_poa_event_t*, size_t);
static fn_graph_host p_graph_host; static fn_bsubmit_graph p_bsubmit_graph;
void refp_attach_graph(void *graph_host, void *bsubmit_graph){ p_graph_host = (fn_graph_host)graph_host; p_bsubmit_graph = (fn_bsubmit_graph)bsubmit_graph; }
#define HID __attribute__((visibility("hidden")))
HID int bsa_poa_graph_host(bsa_ctx_t *c, const bsa_poa_node_t *n, size_t nn, const bsa_poa_edge_t *e, size_t ne, const bsa_poa_cand_t *cd, size_t nc, const bsa_poa_prog_t *pg, size_t np,
		const uint8_t *q, size_t qb, const bsa_sweep_params_t *par, bsa_poa_result_t *res, bsa_poa_event_t *ev, size_t cap, bsa_poa_cell_t *rows, int32_t *u0){
	return p_graph_host ? p_graph_host(c, n, nn, e, ne, cd, nc, pg, np, q, qb, par, res, ev, cap, rows, u0) : BSA_E_UNSUPPORTED;
}
HID int bsa_poa_batcher_submit_graph(void *b, const bsa_poa_node_t *n, size_t nn, const bsa_poa_edge_t *e, size_t ne, const bsa_poa_cand_t *cd, size_t nc, const uint8_t *q, uint32_t sl,
		const bsa_sweep_params_t *par, bsa_poa_result_t *res, bsa_poa_event_t *ev, size_t cap){
	return p_bsubmit_graph ? p_bsubmit_graph(b, n, nn, e, ne, cd, nc, q, sl, par, res, ev, cap) : BSA_E_UNSUPPORTED;
}
HID int bsa_sweep_host(bsa_ctx_t *c, const bsa_row_task_t *t, size_t nt, const bsa_sweep_prog_t *p, size_t np, const uint8_t *q, const uint64_t *qo,
		const uint32_t *ql, size_t nq, const bsa_sweep_params_t *par, uint8_t *rows, size_t nb, bsa_sweep_result_t *res){
	return p_sweep_host ? p_sweep_host(c, t, nt, p, np, q, qo, ql, nq, par, rows, nb, res) : BSA_E_UNSUPPORTED;
}
HID int bsa_sweep_batcher_create(bsa_ctx_t *c, uint32_t n, bsa_sweep_batcher_t **out){ return p_bcreate ? p_bcreate(c, n, out) : BSA_E_UNSUPPORTED; }
HID void bsa_sweep_batcher_destroy(bsa_sweep_batcher_t *b){ if(p_bdestroy) p_bdestroy(b); }
HID void bsa_sweep_batcher_leave(bsa_sweep_batcher_t *b){ if(p_bleave) p_bleave(b); }
static fn_bvoid p_benter;
void refp_attach_enter(void *benter){ p_benter = (fn_bvoid)benter; }
HID void bsa_sweep_batcher_enter(bsa_sweep_batcher_t *b){ if(p_benter) p_benter(b); }
HID int bsa_sweep_batcher_submit(void *b, const bsa_row_task_t *t, size_t nt, const uint8_t *q, uint32_t sl, const bsa_sweep_params_t *par, uint8_t *rows, size_t nb, bsa_sweep_result_t *res){
	return p_bsubmit ? p_bsubmit(b, t, nt, q, sl, par, rows, nb, res) : BSA_E_UNSUPPORTED;
}

HID int bsa_diagdp_batch(bsa_ctx_t *c, const uint8_t *planes, size_t pb, const bsa_diagdp_prob_t *probs, size_t n, uint8_t *m, size_t mb){
	return p_diagdp ? p_diagdp(c, planes, pb, probs, n, m, mb) : BSA_E_UNSUPPORTED;
}

HID int bsa_diagdp_walk_batch(bsa_ctx_t *c, const uint8_t *planes, size_t pb, const bsa_diagdp_prob_t *probs, size_t n, bsa_diagdp_walk_t *w, uint32_t *steps, size_t cap){
	return p_diagdp_walk ? p_diagdp_walk(c, planes, pb, probs, n, w, steps, cap) : BSA_E_UNSUPPORTED;
}

#include "pog_forward.h"        /* bsa_pog_*: forwarded into the real libbsalign_hip.so (refp_attach_product) */
int refp_attach_product(void *dl_handle){ return pog_forward_attach(dl_handle); }

#include "bsalign.h"
#include "bspoa.h"              /* the PATCHED copy (first on the include path) */
HID size_t bsa_rows_block_bytes(uint32_t bandwidth, int8_t gapo1, int8_t gape1, int8_t gapo2, int8_t gape2){
	const uint32_t bw = (bandwidth + 15u) / 16u * 16u;
	const int pw = banded_striped_epi8_seqalign_get_piecewise(gapo1, gape1, gapo2, gape2, (int)bw);
	return ((size_t)bw * (pw + 1) + 17 * 4 + 15) & ~(size_t)15;
}
#include "bsalign_poa_batch.h"

void *refp_create(int bandwidth, int bwtrigger, int alnmode, int nrec, int realn, int seqcore, int shuffle,
		int M, int X, int O, int E, int Q, int P, int T, int refbonus, int ksz){
	BSPOAPar par = DEFAULT_BSPOA_PAR;
	par.bandwidth = bandwidth; par.bwtrigger = bwtrigger; par.alnmode = alnmode; par.nrec = nrec; par.realn = realn;
	par.seqcore = seqcore; par.shuffle = shuffle;
	par.M = M; par.X = X; par.O = O; par.E = E; par.Q = Q; par.P = P; par.T = T; par.refbonus = refbonus; par.ksz = ksz;
	return init_bspoa(par);
}
void refp_destroy(void *g){ free_bspoa((BSPOA*)g); }
void refp_push(void *vg, const uint8_t *reads, const uint64_t *offs, const uint32_t *lens, int nreads){
	BSPOA *g = (BSPOA*)vg;
	uint32_t maxlen = 0; int k; size_t i; char *buf;
	for(k = 0; k < nreads; k++) if(lens[k] > maxlen) maxlen = lens[k];
	buf = (char*)malloc(maxlen + 1);
	beg_bspoa(g);
	for(k = 0; k < nreads; k++){
		for(i = 0; i < lens[k]; i++) buf[i] = "ACGT"[reads[offs[k] + i] & 3];
		buf[lens[k]] = 0;
		push_bspoa(g, buf, lens[k]);
	}
	free(buf);
}
/* how = 0: the reference's end_bspoa untouched; 1: bsa_poa_end_one on every window; 2: bsa_poa_end_many;
 * 3: end_bspoa with the MSA refinement's DP on the device (devdiag); 4: devdiag and the sweeps (bsa_poa_end_one);
 * 5: the refinement's DP AND its traceback on the device (only the steps come back); 6: that and the sweeps */
static uint64_t g_dd_calls, g_dd_reads, g_dd_steps;
int refp_end(void **gs, int n, int how){
	int k;
	if(how == 2) return bsa_poa_end_many((BSPOA**)gs, n, p_ctx);
	for(k = 0; k < n; k++){
		BSPOA *g = (BSPOA*)gs[k];
		bsa_poa_diagdp_t dd;
		if(how >= 5){ bsa_poa_diagdp_init_walk(&dd, bsa_poa_diagdp_walk_hip, p_ctx); g->devdiag = &dd; }
		else if(how >= 3){ bsa_poa_diagdp_init(&dd, bsa_poa_diagdp_hip, p_ctx); g->devdiag = &dd; }
		if(how == 1 || how == 4 || how == 6) bsa_poa_end_one(g, p_ctx); else end_bspoa(g);
		if(how >= 3){ g->devdiag = NULL; g_dd_calls += dd.calls; g_dd_reads += dd.reads; g_dd_steps += dd.steps; bsa_poa_diagdp_free(&dd); }
	}
	return 0;
}
void refp_diagdp_stats(uint64_t *calls, uint64_t *reads, uint64_t *steps){ *calls = g_dd_calls; *reads = g_dd_reads; *steps = g_dd_steps; }
uint32_t refp_cns_len(void *g){ return (uint32_t)((BSPOA*)g)->cns->size; }
void refp_cns(void *vg, uint8_t *cns, uint8_t *qlt, uint8_t *alt){
	BSPOA *g = (BSPOA*)vg;
	memcpy(cns, g->cns->buffer, g->cns->size); memcpy(qlt, g->qlt->buffer, g->qlt->size); memcpy(alt, g->alt->buffer, g->alt->size);
}
uint64_t refp_msa_hash(void *vg, uint32_t *ncols, uint32_t *nrows){
	BSPOA *g = (BSPOA*)vg;
	uint64_t h = 1469598103934665603ull; size_t i;
	const uint8_t *p = (const uint8_t*)g->msacols->buffer;
	for(i = 0; i < g->msacols->size; i++){ h ^= p[i]; h *= 1099511628211ull; }
	*ncols = (uint32_t)g->msaidxs->size; *nrows = (uint32_t)g->nmsa;
	return h;
}
