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
	memset(ad, 0, sizeof(*ad));
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

#endif
