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
#include "bsalign_hip.h"

/* executes one program; returns 0 or a BSA_E_* code.  rows_out receives nblocks row blocks. */
typedef int (*bsa_poa_backend_fn)(void *user, const bsa_row_task_t *tasks, size_t ntasks, const uint8_t *query, uint32_t slen,
                                  const bsa_sweep_params_t *par, uint8_t *rows_out, size_t nblocks, bsa_sweep_result_t *res);

typedef struct {
	bsa_poa_backend_fn run;
	void *user;                 /* e.g. the bsa_ctx_t* for bsa_poa_backend_hip */
	bsa_row_task_t *tasks;      /* grown on demand, reused between reads */
	size_t ntasks, cap;
} bsa_poa_adapter_t;

static inline void bsa_poa_adapter_init(bsa_poa_adapter_t *ad, bsa_poa_backend_fn run, void *user){
	memset(ad, 0, sizeof(*ad));
	ad->run = run; ad->user = user;
}

static inline void bsa_poa_adapter_free(bsa_poa_adapter_t *ad){
	free(ad->tasks);
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

/* drop-in for align_rd_bspoacore (bspoa.h:2515): call it between prepare_rd_align_bspoa and alignment2graph_bspoa */
static inline int bsa_poa_align_rd_core(BSPOA *g, BSPOAPar *par, u2i rid, u4i nhead, u4i ntail, bsa_poa_adapter_t *ad){
	bsa_sweep_params_t sp;
	bsa_sweep_result_t res;
	int rc;
	UNUSED(rid);
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

#endif
