/*
 * bsalign_poa_adapter.h -- the reference-side binding for the POA seq->graph DP.
 *
 * This header is meant to be compiled INSIDE a tree that has the reference's bspoa.h (include it AFTER bspoa.h): it
 * uses the reference's own graph structures (BSPOA, bspoanode_t, bspoaedge_t) and replaces exactly one function,
 *
 *     int align_rd_bspoacore(BSPOA *g, BSPOAPar *par, u2i rid, u4i nhead, u4i ntail)        bspoa.h:2515-2618
 *
 * with bsa_poa_align_rd_core(): same arguments plus the adapter handle, same side effects on the graph (node mpos /
 * vst, g->maxscr / g->maxidx / g->maxoff, every selected node's DP row block in g->memp), so that the callers on
 * either side -- sel_nodes_bspoa / prepare_rd_align_bspoa before it and alignment2graph_bspoa after it
 * (bspoa.h:2643-2652) -- run unchanged.
 *
 * How: the reference's sweep pops nodes from a stack and pushes a node once all its selected in-edges were seen
 * (v->vst == v->nct).  That order depends on the graph only, never on DP values.  bsa_poa_flatten() walks the graph
 * in exactly that order and, instead of computing rows, records what the reference would compute as a program of
 * row tasks (include/bsalign_hip.h: INIT / UPDATE / MERGE / SCORE_TAIL / SCORE_END); the backend (bsa_sweep_host on
 * the GPU) executes the program and returns the row blocks in the reference's own block layout (bspoa.h:1787-1793).
 *
 * oracle/ref_harness.c compiles this header against the real reference to prove the equivalence (same consensus,
 * same per-read results, identical row blocks); INTEGRATION.md shows the three-line patch to align_rd_bspoa.
 */
#ifndef BSALIGN_POA_ADAPTER_H
#define BSALIGN_POA_ADAPTER_H

#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
#include "bsalign_hip.h"
#include "bsalign_poa.h"

/* executes one program; returns 0 or a BSA_E_* code.  rows_out receives nblocks row blocks. */
typedef int (*bsa_poa_backend_fn)(void *user, const bsa_row_task_t *tasks, size_t ntasks, const uint8_t *query, uint32_t slen,
                                  const bsa_sweep_params_t *par, uint8_t *rows_out, size_t nblocks, bsa_sweep_result_t *res);

/* Second backend form (the default on the device): the sweep AND the traceback.  It receives the selected sub-graph as nodes /
 * in-edges / candidates (bsa_poa_node_t ... in bsalign_hip.h) and returns the best end cell and the steps of the walk
 * alignment2graph_bspoa (bspoa.h:2274-2513) would take; no row block comes back.  Returns 0 or a BSA_E_* code; BSA_E_UNSUPPORTED
 * (parameters outside bsa_poa_graph_supported) sends the read down the first form. */
typedef int (*bsa_poa_graph_backend_fn)(void *user, const bsa_poa_node_t *nodes, size_t nnodes, const bsa_poa_edge_t *edges, size_t nedges,
                                        const bsa_poa_cand_t *cands, size_t ncands, const uint8_t *query, uint32_t slen,
                                        const bsa_sweep_params_t *par, bsa_poa_result_t *res, bsa_poa_event_t *events, size_t events_cap);

typedef struct { uint32_t src, toff; } bsa_poa_visit_t;

typedef struct {
	bsa_poa_backend_fn run;
	void *user;                 /* e.g. the bsa_ctx_t* for bsa_poa_backend_hip */
	bsa_row_task_t *tasks;      /* grown on demand, reused between reads */
	size_t ntasks, cap;
	/* graph form */
	bsa_poa_graph_backend_fn run_graph;     /* NULL: first form only */
	bsa_poa_node_t *nodes; size_t nnodes, capnodes;
	bsa_poa_edge_t *edges; size_t nedges, capedges;
	bsa_poa_cand_t *cands; size_t ncands, capcands;
	bsa_poa_event_t *events; size_t capevents;
	bsa_poa_visit_t *visits; uint32_t *voff, *loc; size_t capvisits, capblocks;
	bsa_poa_result_t res;
	int have_trace;             /* the last bsa_poa_align_rd_core went through the graph form: bsa_poa_apply_trace has its steps */
	unsigned long long graph_reads, rows_reads;     /* reads aligned through either form */
	double seconds[3];          /* host time spent building programs, inside the backend (waiting for the device), applying walks */
	/* the library's own POA graph (include/bsalign_poa.h): when use_pog is set, node selection, band placement, program building and the
	 * graph surgery run inside libbsalign_hip on ITS graph; this side only keeps the reference's BSPOA in step for msa_bspoa / cns */
	int use_pog;
	bsa_pog_t *pog;
	bsa_pog_params_t pog_par;
	unsigned long long pog_ncall;   /* g->ncall the mirror was built for (a new beg_bspoa = a new window) */
	void *pog_owner;                /* the BSPOA it mirrors */
	int pog_stale;                  /* a read went through the reference's own path: re-import before the next one */
	uint32_t *pog_gnodes; int32_t *pog_cpos; size_t cap_pog_gnodes, cap_pog_cpos;
	uint8_t *pog_bases; size_t cap_pog_bases;
	unsigned long long pog_reads, pog_imports, pog_declined;
	double pog_seconds[4];          /* keeping the mirror (add / import), gathering inputs (guide alignment, cpos), inside the library, applying the steps to the reference's graph */
} bsa_poa_adapter_t;

static inline double bsa_poa_now(void){ struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

static inline void bsa_poa_adapter_init(bsa_poa_adapter_t *ad, bsa_poa_backend_fn run, void *user){
	memset(ad, 0, sizeof(*ad));
	ad->run = run; ad->user = user;
}

/* both forms: the graph form is tried first, `run` takes the reads it declines */
static inline void bsa_poa_adapter_init_graph(bsa_poa_adapter_t *ad, bsa_poa_graph_backend_fn run_graph, bsa_poa_backend_fn run, void *user){
	bsa_poa_adapter_init(ad, run, user);
	ad->run_graph = run_graph;
}

static inline void bsa_poa_adapter_free(bsa_poa_adapter_t *ad){
	free(ad->tasks); free(ad->nodes); free(ad->edges); free(ad->cands); free(ad->events); free(ad->visits); free(ad->voff); free(ad->loc);
	free(ad->pog_gnodes); free(ad->pog_cpos); free(ad->pog_bases);
	if(ad->pog) bsa_pog_destroy(ad->pog);
	memset(ad, 0, sizeof(*ad));
}

/* the library's own graph for everything around the DP (default on; BSA_POA_POG=0 in the environment keeps the round-4 path: the reference's
 * sel_nodes / prepare_rd_align + bsa_poa_flatten_graph + bsa_poa_apply_trace) */
static inline void bsa_poa_adapter_use_pog(bsa_poa_adapter_t *ad, int on){
	const char *e = getenv("BSA_POA_POG");
	ad->use_pog = (e && e[0] == '0') ? 0 : on;
}

/* the GPU backend: one program through bsa_sweep_host */
static inline int bsa_poa_backend_hip(void *user, const bsa_row_task_t *tasks, size_t ntasks, const uint8_t *query, uint32_t slen,
		const bsa_sweep_params_t *par, uint8_t *rows_out, size_t nblocks, bsa_sweep_result_t *res){
	bsa_sweep_prog_t pg;
	uint64_t qoff = 0;
	pg.first_task = 0; pg.ntasks = (uint32_t)ntasks; pg.first_block = 0; pg.reserved = 0;
	return bsa_sweep_host((bsa_ctx_t*)user, tasks, ntasks, &pg, 1, query, &qoff, &slen, 1, par, rows_out, nblocks, res);
}

/* the GPU backend of the graph form: one program through bsa_poa_graph_host */
static inline int bsa_poa_graph_backend_hip(void *user, const bsa_poa_node_t *nodes, size_t nnodes, const bsa_poa_edge_t *edges, size_t nedges,
		const bsa_poa_cand_t *cands, size_t ncands, const uint8_t *query, uint32_t slen,
		const bsa_sweep_params_t *par, bsa_poa_result_t *res, bsa_poa_event_t *events, size_t events_cap){
	bsa_poa_prog_t pg;
	memset(&pg, 0, sizeof(pg));
	pg.nnodes = (uint32_t)nnodes; pg.nedges = (uint32_t)nedges; pg.ncands = (uint32_t)ncands; pg.slen = slen; pg.event_cap = (uint32_t)events_cap;
	return bsa_poa_graph_host((bsa_ctx_t*)user, nodes, nnodes, edges, nedges, cands, ncands, &pg, 1, query, slen, par, res, events, events_cap, NULL, NULL);
}

static inline bsa_row_task_t* bsa_poa_emit(bsa_poa_adapter_t *ad, uint32_t op){
	bsa_row_task_t *t;
	if(ad->ntasks == ad->cap){
		ad->cap = ad->cap ? ad->cap * 2 : 4096;
		ad->tasks = (bsa_row_task_t*)realloc(ad->tasks, ad->cap * sizeof(bsa_row_task_t));
		if(ad->tasks == NULL){ fprintf(stderr, " -- out of memory in %s -- %s:%d --\n", __FUNCTION__, __FILE__, __LINE__); abort(); }
	}
	t = ad->tasks + ad->ntasks ++;
	memset(t, 0, sizeof(*t));
	t->op = op;
	return t;
}

/* Walk the selected sub-graph in the reference's visiting order and record the row work.  Leaves the nodes' mpos and
 * vst exactly as the reference's sweep leaves them. */
static inline size_t bsa_poa_flatten(BSPOA *g, BSPOAPar *par, u4i nhead, u4i ntail, bsa_poa_adapter_t *ad){
	bspoanode_t *from, *to;
	bspoaedge_t *lnk;
	bsa_row_task_t *t;
	u4i k, cur, ei;
	const int type = seqalign_mode_type(par->alnmode);
	ad->ntasks = 0;
	for(k=0;k<g->sels->size;k++){
		ref_bspoanodev(g->nodes, g->sels->buffer[k])->mpos = MAX_B4 - 1;      /* "not reached yet" */
	}
	from = ref_bspoanodev(g->nodes, nhead);
	from->mpos = -1;
	t = bsa_poa_emit(ad, BSA_ROW_OP_INIT);                                    /* head row = row -1 of the read */
	t->dst = from->mmidx;
	clear_u4v(g->stack);
	push_u4v(g->stack, nhead);
	while(pop_u4v(g->stack, &cur)){
		from = ref_bspoanodev(g->nodes, cur);
		for(ei=from->edge;ei;ei=lnk->next){
			lnk = ref_bspoaedgev(g->edges, ei);
			if(get_bitvec(g->states, lnk->node) == 0) continue;               /* edge leaves the selection */
			to = ref_bspoanodev(g->nodes, lnk->node);
			if(from->mpos + 1 < to->mpos) to->mpos = from->mpos + 1;
			if(lnk->node == ntail){
				t = bsa_poa_emit(ad, BSA_ROW_OP_SCORE_TAIL);
				t->src = from->mmidx; t->qoff_src = from->rpos; t->toff = cur;
				to->vst ++;
				continue;
			}
			t = bsa_poa_emit(ad, BSA_ROW_OP_UPDATE);
			t->src = from->mmidx;
			t->dst = to->vst? 1 : to->mmidx;                                  /* later in-edges go through block 1 and are merged */
			t->qoff_src = from->rpos; t->qoff_dst = to->rpos;
			t->toff = to->mpos;
			t->base = to->base;
			t->prof = (to->base == from->base) * 2 + to->bonus;
			if(to->vst){
				t = bsa_poa_emit(ad, BSA_ROW_OP_MERGE);
				t->src = 1; t->dst = to->mmidx;
			}
			to->vst ++;
			if(to->vst == to->nct){
				if(type != SEQALIGN_MODE_GLOBAL && to->rpos + g->bandwidth >= g->slen){
					t = bsa_poa_emit(ad, BSA_ROW_OP_SCORE_END);
					t->src = to->mmidx; t->qoff_src = to->rpos; t->toff = lnk->node;
				}
				push_u4v(g->stack, lnk->node);
			}
		}
	}
	return ad->ntasks;
}


#define BSA_POA_GROW(ptr, cnt, cap, type, need) do { if((size_t)(need) > (cap)){ (cap) = (size_t)(need) * 2 + 1024; (ptr) = (type*)realloc((ptr), (cap) * sizeof(type)); \
	if((ptr) == NULL){ fprintf(stderr, " -- out of memory in %s -- %s:%d --\n", __FUNCTION__, __FILE__, __LINE__); abort(); } } } while(0)

static inline void bsa_poa_set_input(bsa_poa_input_t *in, uint32_t src, uint32_t movx, uint32_t toff_kind){ in->src = src; in->movx = movx; in->toff_kind = toff_kind; }

/* The selected sub-graph as a program of the graph form.  Same walk as bsa_poa_flatten (the reference's visiting order decides the
 * left-boundary row number of every edge and the order of the end-score candidates; it leaves mpos / vst as the reference's sweep
 * does), but what is recorded is the graph itself: nodes in the order they complete, every node's selected in-edges in the order
 * of its erev list with their coverage -- the order and the tie rule of the traceback (bspoa.h:2426-2477) -- and, for the forward
 * pass, its in-edges folded two at a time (a node with more than two is preceded by partial nodes). */
static inline void bsa_poa_flatten_graph(BSPOA *g, BSPOAPar *par, u4i nhead, u4i ntail, bsa_poa_adapter_t *ad){
	bspoanode_t *from, *to, *w;
	bspoaedge_t *lnk;
	bsa_poa_node_t *nd;
	u4i k, cur, ei, slot, cnt, first_edge, j;
	size_t tot;
	const int type = seqalign_mode_type(par->alnmode);
	ad->nnodes = ad->nedges = ad->ncands = 0;
	/* visit slots: node with block index b records its in-edges, in visiting order, at visits[voff[b] ..] (vst counts them) */
	BSA_POA_GROW(ad->voff, 0, ad->capblocks, uint32_t, g->mmcnt + 1);
	ad->loc = (uint32_t*)realloc(ad->loc, ad->capblocks * sizeof(uint32_t));
	if(ad->loc == NULL){ fprintf(stderr, " -- out of memory in %s -- %s:%d --\n", __FUNCTION__, __FILE__, __LINE__); abort(); }
	tot = 0;
	for(k=0;k<g->sels->size;k++){
		w = ref_bspoanodev(g->nodes, g->sels->buffer[k]);
		w->mpos = MAX_B4 - 1;
		ad->voff[w->mmidx] = (uint32_t)tot;
		ad->loc[w->mmidx] = MAX_U4;
		tot += w->nct;
	}
	BSA_POA_GROW(ad->visits, 0, ad->capvisits, bsa_poa_visit_t, tot + 1);
	BSA_POA_GROW(ad->nodes, 0, ad->capnodes, bsa_poa_node_t, g->sels->size + tot + 4);
	BSA_POA_GROW(ad->edges, 0, ad->capedges, bsa_poa_edge_t, tot + 1);
	BSA_POA_GROW(ad->cands, 0, ad->capcands, bsa_poa_cand_t, g->sels->size + 4);
	from = ref_bspoanodev(g->nodes, nhead);
	from->mpos = -1;
	nd = ad->nodes + ad->nnodes;
	memset(nd, 0, sizeof(*nd));
	nd->rpos = from->rpos; nd->gnode = nhead; nd->base = from->base; nd->flags = from->bonus;
	ad->loc[from->mmidx] = (uint32_t)ad->nnodes ++;
	clear_u4v(g->stack);
	push_u4v(g->stack, nhead);
	while(pop_u4v(g->stack, &cur)){
		from = ref_bspoanodev(g->nodes, cur);
		for(ei=from->edge;ei;ei=lnk->next){
			lnk = ref_bspoaedgev(g->edges, ei);
			if(get_bitvec(g->states, lnk->node) == 0) continue;
			to = ref_bspoanodev(g->nodes, lnk->node);
			if(from->mpos + 1 < to->mpos) to->mpos = from->mpos + 1;
			if(lnk->node == ntail){
				ad->cands[ad->ncands].node = ad->loc[from->mmidx]; ad->cands[ad->ncands].kind = 0; ad->ncands ++;
				to->vst ++;
				continue;
			}
			slot = ad->voff[to->mmidx] + to->vst;
			ad->visits[slot].src = ad->loc[from->mmidx]; ad->visits[slot].toff = (uint32_t)to->mpos;
			to->vst ++;
			if(to->vst != to->nct) continue;
			/* `to` is complete: its records */
			cnt = to->nct;
			first_edge = (u4i)ad->nedges;
			{
				bspoaedge_t *re; u4i ri, found = 0;
				for(ri=to->erev;ri;ri=re->next){
					re = ref_bspoaedgev(g->edges, ri);
					if(get_bitvec(g->states, re->node) == 0) continue;
					w = ref_bspoanodev(g->nodes, re->node);
					ad->edges[ad->nedges].src = ad->loc[w->mmidx]; ad->edges[ad->nedges].cov = re->cov;
					ad->edges[ad->nedges].src_rpos = w->rpos; ad->edges[ad->nedges].reserved = 0;
					ad->nedges ++; found ++;
				}
				if(found != cnt){
					fprintf(stderr, " -- selected in-edges (%u) and in-degree (%u) disagree in %s -- %s:%d --\n", found, cnt, __FUNCTION__, __FILE__, __LINE__); fflush(stderr);
					abort();
				}
			}
			/* forward view: visits 0 and 1 make the first (partial) row, every further visit merges one more in */
			for(j=0;j<cnt;j++){
				const bsa_poa_visit_t *vs = ad->visits + ad->voff[to->mmidx] + j;
				const bspoanode_t *sn = ref_bspoanodev(g->nodes, ad->nodes[vs->src].gnode);
				const uint32_t tk = BSA_POA_IN_PRESENT | ((to->base == sn->base)? BSA_POA_IN_SAME : 0) | (vs->toff & BSA_POA_IN_TOFF);
				const uint32_t movx = (uint32_t)(to->rpos - sn->rpos);
				if(j == 0 || (j >= 2)){
					if(j >= 2){
						/* the row so far becomes a partial node, the new one starts with it */
						ad->nodes[ad->nnodes].gnode = MAX_U4;
						ad->nnodes ++;
					}
					nd = ad->nodes + ad->nnodes;
					memset(nd, 0, sizeof(*nd));
					nd->rpos = to->rpos; nd->base = to->base; nd->flags = to->bonus; nd->first_in = first_edge; nd->n_in = 0;
					if(j >= 2){
						bsa_poa_set_input(nd->in + 0, (uint32_t)ad->nnodes - 1, 0, BSA_POA_IN_PRESENT | BSA_POA_IN_MERGE);
						bsa_poa_set_input(nd->in + 1, vs->src, movx, tk);
					} else {
						bsa_poa_set_input(nd->in + 0, vs->src, movx, tk);
					}
				} else {
					bsa_poa_set_input(nd->in + 1, vs->src, movx, tk);
				}
			}
			nd->gnode = lnk->node; nd->n_in = (uint16_t)cnt;
			ad->loc[to->mmidx] = (uint32_t)ad->nnodes ++;
			if(type != SEQALIGN_MODE_GLOBAL && to->rpos + g->bandwidth >= g->slen){
				ad->cands[ad->ncands].node = ad->loc[to->mmidx]; ad->cands[ad->ncands].kind = 1; ad->ncands ++;
			}
			push_u4v(g->stack, lnk->node);
		}
	}
}

/* drop-in for align_rd_bspoacore (bspoa.h:2515): call it between prepare_rd_align_bspoa and alignment2graph_bspoa */
static inline int bsa_poa_align_rd_core(BSPOA *g, BSPOAPar *par, u2i rid, u4i nhead, u4i ntail, bsa_poa_adapter_t *ad){
	bsa_sweep_params_t sp;
	bsa_sweep_result_t res;
	int rc;
	UNUSED(rid);
	memset(&sp, 0, sizeof(sp));
	sp.rows.mode = par->alnmode;
	sp.rows.bandwidth = g->bandwidth;
	sp.rows.M = par->M; sp.rows.X = par->X; sp.rows.refbonus = par->refbonus;
	sp.rows.gapo1 = par->O; sp.rows.gape1 = par->E; sp.rows.gapo2 = par->Q; sp.rows.gape2 = par->P;
	sp.T = par->T;
	ad->have_trace = 0;
	if(ad->run_graph && nhead != ntail && g->sels->size >= 2){
		const size_t ecap = 2 * ((size_t)g->slen + g->sels->size) + 64;
		double t0 = bsa_poa_now(), t1;
		bsa_poa_flatten_graph(g, par, nhead, ntail, ad);
		BSA_POA_GROW(ad->events, 0, ad->capevents, bsa_poa_event_t, ecap);
		t1 = bsa_poa_now(); ad->seconds[0] += t1 - t0;
		rc = ad->run_graph(ad->user, ad->nodes, ad->nnodes, ad->edges, ad->nedges, ad->cands, ad->ncands, g->qseq->buffer + g->qb, g->slen, &sp,
			&ad->res, ad->events, ecap);
		ad->seconds[1] += bsa_poa_now() - t1;
		if(rc == 0){
			if(ad->res.status != BSA_POA_ST_OK){
				/* the walk left the stored rows: the reference reads outside its arena or does not terminate on this input */
				fprintf(stderr, " -- device traceback stopped (status %d after %d steps at node %d, column %d; read of %u, %u nodes) in %s -- %s:%d --\n", ad->res.status, ad->res.nevents,
					ad->res.fin_node, ad->res.fin_x, (unsigned)g->slen, (unsigned)ad->nnodes, __FUNCTION__, __FILE__, __LINE__); fflush(stderr);
				abort();
			}
			g->maxscr = ad->res.maxscr;
			g->maxidx = (int)ad->nodes[ad->res.maxidx].gnode;
			g->maxoff = ad->res.maxoff;
			ad->have_trace = 1;
			ad->graph_reads ++;
			return g->maxscr;
		}
		if(rc != BSA_E_UNSUPPORTED || ad->run == NULL){
			fprintf(stderr, " -- device sweep failed (code %d, bandwidth %u) in %s -- %s:%d --\n", rc, g->bandwidth, __FUNCTION__, __FILE__, __LINE__); fflush(stderr);
			abort();
		}
		/* declined: undo the walk's visit counters and take the first form */
		{ u4i k; for(k=0;k<g->sels->size;k++) ref_bspoanodev(g->nodes, g->sels->buffer[k])->vst = 0; }
	}
	ad->rows_reads ++;
	bsa_poa_flatten(g, par, nhead, ntail, ad);
	memset(&sp, 0, sizeof(sp));
	sp.rows.mode = par->alnmode;
	sp.rows.bandwidth = g->bandwidth;
	sp.rows.M = par->M; sp.rows.X = par->X; sp.rows.refbonus = par->refbonus;
	sp.rows.gapo1 = par->O; sp.rows.gape1 = par->E; sp.rows.gapo2 = par->Q; sp.rows.gape2 = par->P;
	sp.T = par->T;
	if(bsa_rows_block_bytes(g->bandwidth, par->O, par->E, par->Q, par->P) != g->mmblk){
		fprintf(stderr, " -- row block size mismatch in %s -- %s:%d --\n", __FUNCTION__, __FILE__, __LINE__); fflush(stderr);
		abort();
	}
	rc = ad->run(ad->user, ad->tasks, ad->ntasks, g->qseq->buffer + g->qb, g->slen, &sp, (uint8_t*)g->memp->buffer, g->mmcnt, &res);
	if(rc){
		fprintf(stderr, " -- device sweep failed (code %d, bandwidth %u) in %s -- %s:%d --\n", rc, g->bandwidth, __FUNCTION__, __FILE__, __LINE__); fflush(stderr);
		abort();                                                              /* the reference's error convention */
	}
	g->maxscr = res.maxscr;
	g->maxidx = res.maxidx;
	g->maxoff = res.maxoff;
	return g->maxscr;
}

/* drop-in for alignment2graph_bspoa (bspoa.h:2274-2513) after a bsa_poa_align_rd_core that went through the graph form
 * (ad->have_trace): the device has walked, here the walk's steps are applied to the graph -- the read's bases are merged into
 * the nodes they matched, their column positions set, the read's nodes chained -- and the counters of the result filled in.
 * (The debug strings of the reference's `alnstrs` are not produced.) */
static inline seqalign_result_t bsa_poa_apply_trace(BSPOA *g, BSPOAPar *par, u4i rid, u4i rbeg, u4i nhead, u4i ntail, bsa_poa_adapter_t *ad){
	seqalign_result_t rs;
	bspoanode_t *gn, *rd;
	const bsa_poa_event_t *ev;
	int k, col;
	const double t0 = bsa_poa_now();
	UNUSED(par);
	nhead = ref_bspoanodev(g->nodes, nhead)->header;
	ntail = ref_bspoanodev(g->nodes, ntail)->header;
	ZEROS(&rs);
	for(k=0;k<Int(g->qlen);k++) get_rdnode_bspoa(g, rid, k)->cpos = 0;
	gn = ref_bspoanodev(g->nodes, ad->nodes[ad->res.maxidx].gnode);
	col = gn->cpos;
	rs.qe = ad->res.maxoff + 1;
	rs.te = col + 1;
	for(k=0;k<ad->res.nevents;k++){
		ev = ad->events + k;
		if(ev->bt == SEQALIGN_BT_I){ rs.ins ++; continue; }
		if(ev->bt != SEQALIGN_BT_M){ rs.del ++; continue; }
		gn = ref_bspoanodev(g->nodes, ad->nodes[ev->node].gnode);
		rd = get_rdnode_bspoa(g, rid, rbeg + g->qb + ev->x);
		rd->cpos = gn->cpos;
		if(ad->nodes[ev->node].gnode != nhead && ad->nodes[ev->node].gnode != ntail && rd->base == gn->base){
			merge_nodes_bspoa(g, gn, rd);
			rs.mat ++;
		} else {
			rs.mis ++;
		}
	}
	rs.qb = ad->res.fin_x;
	rs.tb = ref_bspoanodev(g->nodes, ad->nodes[ad->res.fin_node].gnode)->cpos;
	rs.qb += g->qb;
	rs.qe += g->qb;
	/* chain the read's nodes and give the unaligned ones the column of their right neighbour (bspoa.h:2501-2511) */
	connect_rdnode_bspoa(g, rid, rbeg + g->qlen);
	for(k=Int(g->qlen)-1;k>=0;k--){
		connect_rdnode_bspoa(g, rid, rbeg + k);
		rd = get_rdnode_bspoa(g, rid, rbeg + k);
		if(rd->cpos) col = rd->cpos; else rd->cpos = col;
	}
	ad->seconds[2] += bsa_poa_now() - t0;
	return rs;
}

/* ---- the reference's graph as the flat snapshot of include/bsalign_poa.h (bsa_pog_import; the fixtures of tests/golden) ---- */
typedef struct {
	bsa_pog_snapshot_t snap;
	bsa_pog_node_t *nodes; uint32_t *ndoff, *rdlen, *out_off, *out_to, *out_cov, *in_off, *in_from;
} bsa_poa_graph_export_t;

static inline void bsa_poa_graph_export_free(bsa_poa_graph_export_t *x){
	free(x->nodes); free(x->ndoff); free(x->rdlen); free(x->out_off); free(x->out_to); free(x->out_cov); free(x->in_off); free(x->in_from);
	memset(x, 0, sizeof(*x));
}

/* returns 0, or -1 when the host has no memory for the copy (nothing is left allocated) */
static inline int bsa_poa_graph_export(BSPOA *g, bsa_poa_graph_export_t *x){
	const u4i n = (u4i)g->nodes->size, nr = (u4i)g->seqs->nseq;
	u4i i, ei, ne = 0, k;
	bspoaedge_t *e;
	memset(x, 0, sizeof(*x));
	x->nodes = (bsa_pog_node_t*)calloc((size_t)n + 1, sizeof(bsa_pog_node_t));
	x->ndoff = (uint32_t*)calloc((size_t)nr + 1, 4); x->rdlen = (uint32_t*)calloc((size_t)nr + 1, 4);
	x->out_off = (uint32_t*)calloc((size_t)n + 2, 4); x->in_off = (uint32_t*)calloc((size_t)n + 2, 4);
	if(!x->nodes || !x->ndoff || !x->rdlen || !x->out_off || !x->in_off){ bsa_poa_graph_export_free(x); return -1; }
	for(i=0;i<n;i++){
		const bspoanode_t *u = ref_bspoanodev(g->nodes, i);
		bsa_pog_node_t *d = x->nodes + i;
		d->header = u->header; d->next = u->next; d->prev = u->prev; d->pos = u->pos; d->cpos = u->cpos; d->rid = u->rid; d->cov = u->cov; d->base = u->base;
		d->flags = (uint8_t)((u->bless? BSA_POG_F_BLESS : 0) | (u->rdc? BSA_POG_F_RDC : 0) | (u->rdd? BSA_POG_F_RDD : 0) | (u->ref? BSA_POG_F_REF : 0));
		for(ei=u->edge;ei;ei=e->next){ e = ref_bspoaedgev(g->edges, ei); ne ++; }
	}
	for(i=0;i<nr;i++){ x->ndoff[i] = g->ndoffs->buffer[i]; x->rdlen[i] = g->seqs->rdlens->buffer[i]; }
	x->out_to = (uint32_t*)calloc((size_t)ne + 1, 4); x->out_cov = (uint32_t*)calloc((size_t)ne + 1, 4); x->in_from = (uint32_t*)calloc((size_t)ne + 1, 4);
	if(!x->out_to || !x->out_cov || !x->in_from){ bsa_poa_graph_export_free(x); return -1; }
	for(i=0,k=0;i<n;i++){
		x->out_off[i] = k;
		for(ei=ref_bspoanodev(g->nodes, i)->edge;ei;ei=e->next){ e = ref_bspoaedgev(g->edges, ei); x->out_to[k] = e->node; x->out_cov[k] = e->cov; k ++; }
	}
	x->out_off[n] = k;
	for(i=0,k=0;i<n;i++){
		x->in_off[i] = k;
		for(ei=ref_bspoanodev(g->nodes, i)->erev;ei;ei=e->next){ e = ref_bspoaedgev(g->edges, ei); x->in_from[k ++] = e->node; }
	}
	x->in_off[n] = k;
	x->snap.nnodes = n; x->snap.nreads = nr; x->snap.head = g->HEAD; x->snap.tail = g->TAIL;
	x->snap.nodes = x->nodes; x->snap.ndoff = x->ndoff; x->snap.rdlen = x->rdlen;
	x->snap.out_off = x->out_off; x->snap.out_to = x->out_to; x->snap.out_cov = x->out_cov; x->snap.in_off = x->in_off; x->snap.in_from = x->in_from;
	return 0;
}

static inline void bsa_poa_pog_params(const BSPOAPar *par, bsa_pog_params_t *pp){
	memset(pp, 0, sizeof(*pp));
	pp->alnmode = seqalign_mode_type(par->alnmode); pp->bandwidth = par->bandwidth; pp->bwtrigger = par->bwtrigger; pp->nrec = par->nrec; pp->seqcore = (int32_t)par->seqcore;
	pp->M = par->M; pp->X = par->X; pp->O = par->O; pp->E = par->E; pp->Q = par->Q; pp->P = par->P; pp->T = par->T; pp->refbonus = par->refbonus;
}

/* the mirror of this window: built read by read when the window's first read comes (bsa_pog_add_read = _add_read_bspoa_core, bspoa.h:916-951),
 * re-imported whole after a read that went through the reference's own path */
static inline int bsa_poa_pog_sync(BSPOA *g, BSPOAPar *par, bsa_poa_adapter_t *ad){
	bsa_pog_params_t pp;
	u4i r, i;
	int rc;
	bsa_poa_pog_params(par, &pp);
	if(ad->pog && memcmp(&pp, &ad->pog_par, sizeof(pp))){ bsa_pog_destroy(ad->pog); ad->pog = NULL; }
	if(ad->pog == NULL){
		if((rc = bsa_pog_create(&pp, &ad->pog)) != BSA_OK) return rc;
		ad->pog_par = pp; ad->pog_owner = NULL;
	}
	if(ad->pog_owner == (void*)g && ad->pog_ncall == g->ncall && !ad->pog_stale) return BSA_OK;
	if(g->nrds == 1 && !ad->pog_stale){
		/* a fresh window: every read was given its nodes (end_bspoa, bspoa.h:4746-4748), none is aligned yet */
		bsa_pog_clear(ad->pog);
		for(r=0;r<g->seqs->nseq;r++){
			const u4i len = g->seqs->rdlens->buffer[r];
			BSA_POA_GROW(ad->pog_bases, 0, ad->cap_pog_bases, uint8_t, len + 1);
			for(i=0;i<len;i++) ad->pog_bases[i] = get_basebank(g->seqs->rdseqs, g->seqs->rdoffs->buffer[r] + i);
			if((rc = bsa_pog_add_read(ad->pog, ad->pog_bases, len, NULL)) != BSA_OK) return rc;
		}
	} else {
		bsa_poa_graph_export_t x;
		if(bsa_poa_graph_export(g, &x) != 0) return BSA_E_NOMEM;
		rc = bsa_pog_import(ad->pog, &x.snap, NULL);
		bsa_poa_graph_export_free(&x);
		if(rc != BSA_OK) return rc;
		ad->pog_imports ++;
	}
	ad->pog_owner = (void*)g; ad->pog_ncall = g->ncall; ad->pog_stale = 0;
	return BSA_OK;
}

/* Drop-in for the body of align_rd_bspoa (bspoa.h:2632-2657) between `if(rlen == 0) return rs` and its return: selection, band placement, DP,
 * walk and surgery through the library's own graph (bsa_pog_*), then the same steps applied to the reference's graph so that msa_bspoa / cns_bspoa
 * go on from it.  realn: align_rd_bspoa's own flag -- the stretch was just cut out of the reference's graph (bspoa.h:2626-2630) and is cut out of the
 * mirror here (bsa_pog_cut).  refmode: the band comes from the read's SAM CIGAR when one was pushed with it (bspoa.h:2055-2085).  Returns 1 with *out
 * filled, or 0 when the read has to take the reference's own path (a shape the kernel declines): the caller goes on with sel_nodes_bspoa etc. and
 * the mirror is re-imported before the next read. */
static inline int bsa_poa_align_rd_pog(BSPOA *g, BSPOAPar *par, int realn, u2i rid, int rbeg, int rlen, bsa_poa_adapter_t *ad, seqalign_result_t *out){
	bsa_pog_read_t rd;
	bsa_pog_guide_t gd;
	bsa_poa_result_t res;
	bsa_result_t brs;
	seqalign_result_t rs, krs;
	const bsa_poa_event_t *ev;
	const uint32_t *sel;
	const uint64_t *aux; size_t naux, a;
	bspoanode_t *gn, *rdn;
	u4i k, nhead, ntail;
	int rc, col, x;
	double t0, t1;
	if(!ad->use_pog || ad->run_graph == NULL) return 0;
	t0 = bsa_poa_now();
	if(bsa_poa_pog_sync(g, par, ad) != BSA_OK){ ad->pog_stale = 1; return 0; }
	if(realn && bsa_pog_cut(ad->pog, rid, (uint32_t)rbeg, (uint32_t)rlen) != BSA_OK){ ad->pog_stale = 1; ad->pog_declined ++; return 0; }          /* (a mirror re-imported just now holds the stretch cut already: cutting it again changes nothing) */
	t1 = bsa_poa_now(); ad->pog_seconds[0] += t1 - t0; t0 = t1;
	rc = bsa_pog_select(ad->pog, rid, (uint32_t)rbeg, (uint32_t)rlen, &rd, &sel);
	if(rc != BSA_OK){ ad->pog_stale = 1; ad->pog_declined ++; return 0; }
	if(rd.nhead == rd.ntail || rd.nsel < 2){ bsa_pog_abort(ad->pog); ad->pog_stale = 1; ad->pog_declined ++; return 0; }
	/* the read, and its guide alignment against the running consensus when the band is placed by one (bspoa.h:2033-2038, 2086-2106) */
	g->qlen = g->slen = (u4i)rlen; g->qb = 0; g->qe = g->qlen;
	clear_and_encap_u1v(g->qseq, g->qlen);
	bitseq_basebank(g->seqs->rdseqs, g->seqs->rdoffs->buffer[rid] + rbeg, g->qlen, g->qseq->buffer);
	g->qseq->size = g->qlen;
	memset(&gd, 0, sizeof(gd));
	gd.reflen = g->par->refmode ? (uint32_t)g->backbone : (uint32_t)g->cns->size;
	if(g->par->refmode && g->cges->buffer[rid] > g->cgbs->buffer[rid]){
		/* (bspoa.h:2073 reads the word behind the read's CIGAR; so does the library: the caller of push_bspoacore keeps one readable, as for the reference) */
		gd.sam = 1; gd.cigar = g->cigars->buffer + g->cgbs->buffer[rid]; gd.ncigar = (uint32_t)(g->cges->buffer[rid] - g->cgbs->buffer[rid]);
	} else
	if(bsa_pog_needs_guide(ad->pog, gd.reflen)){
		if(par->ksz) krs = kmer_striped_seqedit_pairwise(par->ksz, g->qseq->buffer, g->qseq->size, g->cns->buffer, g->cns->size, g->memp, g->stack, 0);
		else krs = striped_seqedit_pairwise(g->qseq->buffer, g->qseq->size, g->cns->buffer, g->cns->size, par->alnmode, 0, g->memp, g->stack, 0);
		gd.have = 1; gd.qb = krs.qb; gd.qe = krs.qe; gd.tb = krs.tb; gd.te = krs.te; gd.cigar = g->stack->buffer; gd.ncigar = (uint32_t)g->stack->size;
	}
	BSA_POA_GROW(ad->pog_cpos, 0, ad->cap_pog_cpos, int32_t, rd.nsel + 1);
	for(k=0;k<rd.nsel;k++) ad->pog_cpos[k] = ref_bspoanodev(g->nodes, sel[k])->cpos;
	t1 = bsa_poa_now(); ad->pog_seconds[1] += t1 - t0; t0 = t1;
	rc = bsa_pog_place(ad->pog, &gd, ad->pog_cpos, &rd);
	if(rc == BSA_OK) rc = bsa_pog_run(ad->pog, (bsa_pog_backend_fn)ad->run_graph, ad->user, &res, &ev);
	if(rc != BSA_OK){
		if(rc != BSA_E_UNSUPPORTED){
			fprintf(stderr, " -- device sweep failed (code %d, bandwidth %u) in %s -- %s:%d --\n", rc, rd.bandwidth, __FUNCTION__, __FILE__, __LINE__); fflush(stderr);
			abort();                                                              /* the reference's error convention */
		}
		bsa_pog_abort(ad->pog); ad->pog_stale = 1; ad->pog_declined ++;
		return 0;
	}
	/* the auxiliary edges of this alignment, taken over before bsa_pog_apply takes them back */
	bsa_pog_aux_edges(ad->pog, &aux, &naux);
	clear_u8v(g->todels);
	for(a=0;a<naux;a++) push_u8v(g->todels, aux[a]);
	BSA_POA_GROW(ad->pog_gnodes, 0, ad->cap_pog_gnodes, uint32_t, (size_t)res.nevents + 1);
	rc = bsa_pog_apply(ad->pog, &brs, ad->pog_gnodes);
	if(rc != BSA_OK){ fprintf(stderr, " -- bsa_pog_apply failed (code %d) in %s -- %s:%d --\n", rc, __FUNCTION__, __FILE__, __LINE__); fflush(stderr); abort(); }
	t1 = bsa_poa_now(); ad->pog_seconds[2] += t1 - t0; t0 = t1;
	/* ---- the same on the reference's graph: auxiliary edges in, the walk's merges, the read chained, auxiliary edges out (the order of the
	 * coverage-sorted edge lists depends on every one of these changes, bspoa.h:464-494) */
	nhead = rd.nhead; ntail = rd.ntail;
	g->bandwidth = rd.bandwidth; g->qb = rd.qb; g->qe = rd.qe; g->slen = rd.slen;
	g->maxscr = res.maxscr; g->maxoff = res.maxoff;
	for(a=0;a<g->todels->size;a++) chg_edge_bspoa(g, ref_bspoanodev(g->nodes, g->todels->buffer[a] >> 32), ref_bspoanodev(g->nodes, g->todels->buffer[a] & MAX_U4), 1, NULL);
	for(x=0;x<Int(g->qlen);x++) get_rdnode_bspoa(g, rid, x)->cpos = 0;
	col = 0;
	for(x=0;x<res.nevents;x++){
		if(ev[x].bt != SEQALIGN_BT_M) continue;
		gn = ref_bspoanodev(g->nodes, ad->pog_gnodes[x]);
		rdn = get_rdnode_bspoa(g, rid, rbeg + g->qb + ev[x].x);
		rdn->cpos = gn->cpos;
		if(ad->pog_gnodes[x] != nhead && ad->pog_gnodes[x] != ntail && rdn->base == gn->base) merge_nodes_bspoa(g, gn, rdn);
	}
	/* (the column the fill starts from: the best end cell's node, bspoa.h:2300-2301 -- its graph index is what the first step or, without steps, the end of the walk names) */
	col = brs.te - 1;
	connect_rdnode_bspoa(g, rid, rbeg + g->qlen);
	for(x=Int(g->qlen)-1;x>=0;x--){
		connect_rdnode_bspoa(g, rid, rbeg + x);
		rdn = get_rdnode_bspoa(g, rid, rbeg + x);
		if(rdn->cpos) col = rdn->cpos; else rdn->cpos = col;
	}
	for(a=0;a<g->todels->size;a++) chg_edge_bspoa(g, ref_bspoanodev(g->nodes, g->todels->buffer[a] >> 32), ref_bspoanodev(g->nodes, g->todels->buffer[a] & MAX_U4), -1, NULL);
	clear_u8v(g->todels);
	ZEROS(&rs);
	rs.score = brs.score; rs.qb = brs.qb; rs.qe = brs.qe; rs.tb = brs.tb; rs.te = brs.te; rs.mat = brs.mat; rs.mis = brs.mis; rs.ins = brs.ins; rs.del = brs.del; rs.aln = brs.aln;
	*out = rs;
	ad->pog_reads ++; ad->graph_reads ++;
	ad->pog_seconds[3] += bsa_poa_now() - t0;
	return 1;
}

#endif
