/*
 * ref_poa_harness.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Drives the REAL reference POA (ruanjue/bsalign bspoa.h, compiled from /root/reference via -I, nothing copied) in
 * three ways so that the device sweep and its reference-side binding can be pinned:
 *
 *   mode 0  beg_bspoa / push_bspoa / end_bspoa untouched                                  bspoa.h:1775, 961, 4722
 *   mode 1  the same steps driven from here (poa_finish / poa_align_read below restate only the ORCHESTRATION of
 *           end_bspoa bspoa.h:4722-4776 and align_rd_bspoa bspoa.h:2620-2667 -- every step is a call into the
 *           reference), with the reference's own align_rd_bspoacore: must reproduce mode 0 exactly
 *   mode 2  as mode 1, but align_rd_bspoacore is replaced by bsa_poa_align_rd_core() from
 *           include/bsalign_poa_adapter.h with a backend supplied by the test (the oracle's orc_sweep_run here; the
 *           GPU's bsa_sweep_host in a deployment).  After every read the reference's core is re-run on the same graph
 *           state and its row blocks / best end cell are compared with what the backend returned.
 *   mode 3  as mode 2 with the DEVICE as backend: bsa_poa_backend_hip -> bsa_sweep_host of libbsalign_hip.so, whose
 *           address the GPU test hands in (ref_poa_set_device) -- the real end_bspoa with its sweep on the MI355X.
 *   mode 4  MANY windows in lock-step (ref_poa_run_many): every window runs the mode-1 orchestration on a host thread of
 *           its own, its sweeps go to the product's batcher (bsa_sweep_batcher_submit, attached with
 *           ref_poa_set_batcher), which runs read r of all windows as one device launch.  No re-run of the reference's
 *           core here: the test compares every window's consensus / MSA with a mode-0 run of the same reads.
 *
 *   mode 5  the GRAPH form of the binding in the shadow of the reference: bsa_poa_flatten_graph + a backend supplied by the test
 *           (the oracle's orc_wf_backend on the CPU, bsa_poa_graph_host on the GPU) give the best end cell and the steps of the
 *           traceback; then the reference's own align_rd_bspoacore and alignment2graph_bspoa run on the same graph state -- the
 *           latter with the TEST-ONLY recording hook of oracle/bspoa_trace_record.diff (build _ref/libbsref_trace.so) -- and
 *           best cell, every (node, x, bt) step and the end of the walk are compared; the reference's results are the ones kept.
 *   mode 6  the graph form as the product runs it: flatten, backend, bsa_poa_apply_trace; nothing of the reference's sweep or
 *           walk runs.  The test compares consensus / MSA with a mode-0 run.
 *   mode 8  the library's OWN graph (include/bsalign_poa.h, bsa_pog_*: container, node selection, band placement, program building,
 *           surgery) in the shadow of the reference, step by step on every read: its selection list and in-degrees against sel_nodes_bspoa's,
 *           its band width / read interval / every node's band offset / auxiliary edges against prepare_rd_align_bspoa's, its program byte
 *           for byte against bsa_poa_flatten_graph on the reference's graph, its result against the binding's, and after the surgery the WHOLE
 *           graph (rings, coverages, flags, every edge list in order) against the reference's.  The library's graph is never re-imported:
 *           it is built once with bsa_pog_add_read and evolves by its own surgery.
 *   mode 9  that path as a patched reference runs it (bsa_poa_align_rd_pog of include/bsalign_poa_adapter.h): nothing of sel_nodes /
 *           prepare_rd_align / align_rd_bspoacore / alignment2graph runs.  The test compares consensus / MSA with a mode-0 run.
 *
 * Recorded per read: the seqalign_result_t of align_rd_bspoa, the program (tasks), the best end cell and a hash of
 * the reference's row blocks -- tests/golden/make_golden_poa.py turns these into the committed fixtures.
 */
#include <stdint.h>
#include "bsalign.h"
/* The POA places the band of a long read by a k-mer anchored edit alignment against the current consensus
 * (bspoa.h:2087-2090, 4360-4361).  When a GPU test has attached the product's batch entry (ref_poa_set_kmer), those
 * calls go to the device instead -- a batch of one through bsa_kmer_edit_batch, the CIGAR pushed into the reference's
 * own vector -- so that the whole end_bspoa can be checked with BOTH of its alignment steps on the MI355X. */
typedef int (*kmer_batch_fn)(void *ctx, const uint8_t *seqs, size_t seqs_bytes, const uint64_t *qoff, const uint32_t *qlen,
		const uint64_t *toff, const uint32_t *tlen, size_t n, const void *par, void *out, uint32_t *cigar, size_t cap, uint64_t *cigar_off, uint32_t *status);
static kmer_batch_fn g_kmer_batch = NULL;
static void *g_kmer_ctx = NULL;
static long g_kmer_calls = 0, g_kmer_device_calls = 0;
void ref_poa_set_kmer(void *kmer_batch_addr, void *ctx){ g_kmer_batch = (kmer_batch_fn)kmer_batch_addr; g_kmer_ctx = ctx; }
long ref_poa_kmer_calls(int device){ return device ? g_kmer_device_calls : g_kmer_calls; }
static inline seqalign_result_t harness_kmer_edit(u1i ksz, u1i *qseq, u4i qlen, u1i *tseq, u4i tlen, b1v *mempool, u4v *cigars, int verbose){
	g_kmer_calls ++;
	if(g_kmer_batch == NULL || qlen == 0 || tlen == 0) return kmer_striped_seqedit_pairwise(ksz, qseq, qlen, tseq, tlen, mempool, cigars, verbose);
	seqalign_result_t rs;
	uint32_t par[2] = { ksz, 1 }, status = 0;
	uint64_t qoff = 0, toff = qlen, off[2] = {0, 0};
	size_t cap = (size_t)qlen + tlen + 8, k;
	uint8_t *seqs = (uint8_t*)malloc((size_t)qlen + tlen + 1);
	uint32_t *cig = (uint32_t*)malloc(cap * sizeof(uint32_t));
	memcpy(seqs, qseq, qlen); memcpy(seqs + qlen, tseq, tlen);
	int rc = g_kmer_batch(g_kmer_ctx, seqs, (size_t)qlen + tlen, &qoff, &qlen, &toff, &tlen, 1, par, &rs, cig, cap, off, &status);
	if(rc != 0 || status != 0){ fprintf(stderr, " -- device k-mer alignment failed (%d, status %u) --\n", rc, status); abort(); }
	clear_u4v(cigars);
	for(k = 0; k < off[1]; k++) push_u4v(cigars, cig[k]);
	free(seqs); free(cig);
	g_kmer_device_calls ++;
	return rs;
}
#define kmer_striped_seqedit_pairwise harness_kmer_edit
#include "bspoa.h"
#undef kmer_striped_seqedit_pairwise
#include "pog_forward.h"
#include "../include/bsalign_poa_adapter.h"
#include <time.h>
/* the product library's handle (ctypes CDLL._handle of libbsalign_hip.so): the bsa_pog_* entry points of modes 8 / 9 */
int ref_poa_attach_product(void *dl_handle){ return pog_forward_attach(dl_handle); }

/* the adapter's two link-time dependencies on libbsalign_hip.so, satisfied locally: this library must load without HIP.
 * bsa_sweep_host forwards to the real one when a GPU test has attached it (ref_poa_set_device). */
typedef int (*sweep_host_fn)(bsa_ctx_t*, const bsa_row_task_t*, size_t, const bsa_sweep_prog_t*, size_t, const uint8_t*, const uint64_t*,
		const uint32_t*, size_t, const bsa_sweep_params_t*, uint8_t*, size_t, bsa_sweep_result_t*);
static sweep_host_fn g_sweep_host = NULL;
static void *g_device_ctx = NULL;
void ref_poa_set_device(void *sweep_host_addr, void *ctx){ g_sweep_host = (sweep_host_fn)sweep_host_addr; g_device_ctx = ctx; }
__attribute__((visibility("hidden"))) size_t bsa_rows_block_bytes(uint32_t bandwidth, int8_t gapo1, int8_t gape1, int8_t gapo2, int8_t gape2){
	const uint32_t bw = (bandwidth + 15u) / 16u * 16u;
	const int pw = banded_striped_epi8_seqalign_get_piecewise(gapo1, gape1, gapo2, gape2, (int)bw);
	return ((size_t)bw * (pw + 1) + 17 * 4 + 15) & ~(size_t)15;
}
__attribute__((visibility("hidden"))) int bsa_sweep_host(bsa_ctx_t *ctx, const bsa_row_task_t *tasks, size_t ntasks, const bsa_sweep_prog_t *progs, size_t nprogs,
		const uint8_t *queries, const uint64_t *qoff, const uint32_t *qlen, size_t nqueries,
		const bsa_sweep_params_t *par, uint8_t *rows_out, size_t nblocks, bsa_sweep_result_t *results){
	if(g_sweep_host) return g_sweep_host(ctx, tasks, ntasks, progs, nprogs, queries, qoff, qlen, nqueries, par, rows_out, nblocks, results);
	return BSA_E_UNSUPPORTED;       /* no device library attached */
}

typedef int (*graph_host_fn)(bsa_ctx_t*, const bsa_poa_node_t*, size_t, const bsa_poa_edge_t*, size_t, const bsa_poa_cand_t*, size_t, const bsa_poa_prog_t*, size_t,
		const uint8_t*, size_t, const bsa_sweep_params_t*, bsa_poa_result_t*, bsa_poa_event_t*, size_t, bsa_poa_cell_t*, int32_t*);
static graph_host_fn g_graph_host = NULL;
void ref_poa_set_graph_host(void *graph_host_addr, void *ctx){ g_graph_host = (graph_host_fn)graph_host_addr; g_device_ctx = ctx; }
__attribute__((visibility("hidden"))) int bsa_poa_graph_host(bsa_ctx_t *ctx, const bsa_poa_node_t *nodes, size_t nnodes, const bsa_poa_edge_t *edges, size_t nedges,
		const bsa_poa_cand_t *cands, size_t ncands, const bsa_poa_prog_t *progs, size_t nprogs, const uint8_t *queries, size_t query_bytes,
		const bsa_sweep_params_t *par, bsa_poa_result_t *results, bsa_poa_event_t *events, size_t events_cap, bsa_poa_cell_t *rows_out, int32_t *u0_out){
	if(g_graph_host) return g_graph_host(ctx, nodes, nnodes, edges, nedges, cands, ncands, progs, nprogs, queries, query_bytes, par, results, events, events_cap, rows_out, u0_out);
	return BSA_E_UNSUPPORTED;
}
/* graph-form backend of modes 5 / 6: the oracle's orc_wf_backend (CPU tests), or NULL = the device through bsa_poa_graph_backend_hip */
static bsa_poa_graph_backend_fn g_graph_backend = NULL;
static void *g_graph_backend_user = NULL;
void ref_poa_set_graph_backend(void *fn, void *user){ g_graph_backend = (bsa_poa_graph_backend_fn)fn; g_graph_backend_user = user; }
/* the product's lock-step batcher, graph form (bsa_poa_batcher_submit_graph) */
static bsa_poa_graph_backend_fn g_batch_submit_graph = NULL;
void ref_poa_set_batcher_graph(void *submit_graph_addr){ g_batch_submit_graph = (bsa_poa_graph_backend_fn)submit_graph_addr; }

/* the product's batcher (bsalign_hip.h: bsa_sweep_batcher_submit / _leave), attached by the GPU test */
typedef void (*batch_leave_fn)(void *batcher);
static bsa_poa_backend_fn g_batch_submit = NULL;
static batch_leave_fn g_batch_leave = NULL;
static void *g_batcher = NULL;
void ref_poa_set_batcher(void *submit_addr, void *leave_addr, void *batcher){
	g_batch_submit = (bsa_poa_backend_fn)submit_addr; g_batch_leave = (batch_leave_fn)leave_addr; g_batcher = batcher;
}

typedef void (*orc_sweep_fn)(uint8_t *rows, const void *tasks, const void *progs, size_t nprogs,
		const uint8_t *queries, const uint64_t *qoff, const uint32_t *qlen,
		int mode, uint32_t bandwidth, int M, int X, int refbonus, int gapo1, int gape1, int gapo2, int gape2, int T, void *results);

typedef struct {
	seqalign_result_t rs;
	int maxscr, maxidx, maxoff;
	uint32_t bandwidth, slen, qb, nblocks, ntasks, piecewise;
	uint64_t task_off;          /* into poa->tasks */
	uint64_t query_off;         /* into poa->queries */
	uint64_t rows_hash;         /* FNV-1a over the used bytes of every written node block of the reference's memp */
	int mismatch;               /* mode 2: bit 0 best end cell differs, bit 1 some row block differs; mode 5: bit 2 a step of the walk differs, bit 3 its end */
	/* graph form (modes 5 / 6, when recording): slices of poa->gnodes / gedges / gcands / gtrace */
	uint64_t node_off, edge_off, cand_off, trace_off;
	uint32_t nnodes, nedges, ncands, ntrace;
	int32_t fin_gnode, fin_x, maxidx_local;
} poa_read_rec_t;

typedef struct {
	BSPOA *g;
	bsa_poa_adapter_t ad;
	orc_sweep_fn sweep;
	poa_read_rec_t *recs; size_t nrec, caprec;
	bsa_row_task_t *tasks; size_t ntasks, captasks;
	uint8_t *queries; size_t nq, capq;
	bsa_poa_node_t *gnodes; size_t ngn, capgn;
	bsa_poa_edge_t *gedges; size_t nge, capge;
	bsa_poa_cand_t *gcands; size_t ngc, capgc;
	bsa_poa_event_t *gtrace; size_t ngt, capgt;       /* the REFERENCE's steps (node = graph node index) */
	bsa_poa_event_t *cur_trace; size_t ncur, capcur;  /* steps of the read being aligned, filled by the recording hook */
	int mode, record_programs;
	double core_seconds;        /* mode 1: wall time inside the reference's align_rd_bspoacore */
	uint64_t core_updates;      /* mode 1: row updates (edges) those calls processed */
	uint64_t core_merges;
	uint64_t pog_checked[6];    /* mode 8: selected nodes, placed nodes, program bytes, steps, graph nodes, graph edges compared */
	/* record_programs & 4 (mode 5): the flat graph BEFORE every aligned read with what the reference then decided (tests/golden/make_golden_poa_pog.py) */
	uint8_t *snap; size_t nsnap, capsnap;
	uint64_t *snapoff; size_t nsnaprec, capsnaprec;
	/* refmode: the reads' SAM CIGARs against read 0 (read k: cigs[coffs[k] .. coffs[k + 1])), pushed with the reads; realn_pass (mode 8): after the first
	 * stage every aligned read has a stretch re-aligned through the realn entry of align_rd_bspoa (1: its middle half, 2: the whole read) */
	const uint32_t *cigs; const uint64_t *coffs;
	int realn_pass;
} ref_poa_t;

/* one snapshot record: 20 x u64 header [nnodes, nreads, nedges, ncigar, nsel, naux, head, tail, guide.have, qb, qe, tb, te, reflen, bandwidth, slen, rd.qb, rd.qe, rid, rlen]
 * then nodes (28 B each) | ndoff | rdlen | out_off (nnodes + 1) | out_to | out_cov | in_off (nnodes + 1) | in_from | cigar | sels (all u32) | aux (u64) */
static void snap_put(ref_poa_t *p, const void *src, size_t bytes){
	if(p->nsnap + bytes + 8 > p->capsnap){ p->capsnap = (p->nsnap + bytes) * 2 + 4096; p->snap = (uint8_t*)realloc(p->snap, p->capsnap); }
	if(bytes) memcpy(p->snap + p->nsnap, src, bytes);
	p->nsnap += (bytes + 7) & ~(size_t)7;
}
static void snap_record(ref_poa_t *p, const bsa_poa_graph_export_t *x, const uint64_t hdr[20], const uint32_t *cigar, const uint32_t *sels, const uint64_t *aux){
	const size_t n = x->snap.nnodes, nr = x->snap.nreads, ne = x->out_off[n];
	if(p->nsnaprec == p->capsnaprec){ p->capsnaprec = p->capsnaprec ? p->capsnaprec * 2 : 64; p->snapoff = (uint64_t*)realloc(p->snapoff, p->capsnaprec * sizeof(uint64_t)); }
	p->snapoff[p->nsnaprec ++] = p->nsnap;
	snap_put(p, hdr, 20 * sizeof(uint64_t));
	snap_put(p, x->nodes, n * sizeof(bsa_pog_node_t)); snap_put(p, x->ndoff, nr * 4); snap_put(p, x->rdlen, nr * 4);
	snap_put(p, x->out_off, (n + 1) * 4); snap_put(p, x->out_to, ne * 4); snap_put(p, x->out_cov, ne * 4);
	snap_put(p, x->in_off, (n + 1) * 4); snap_put(p, x->in_from, ne * 4);
	snap_put(p, cigar, (size_t)hdr[3] * 4); snap_put(p, sels, (size_t)hdr[4] * 4); snap_put(p, aux, (size_t)hdr[5] * 8);
}

static uint64_t fnv1a(uint64_t h, const void *p, size_t n){
	const uint8_t *b = (const uint8_t*)p; size_t i;
	for(i = 0; i < n; i++){ h ^= b[i]; h *= 0x100000001B3ULL; }
	return h;
}

/* the tail node's block is never written by the sweep (uninitialised arena bytes): skipped */
static uint64_t hash_node_blocks(BSPOA *g, u4i tail){
	const size_t used = (size_t)g->bandwidth * (g->piecewise + 1) + (WORDSIZE + 1) * sizeof(int);
	const u8i skip = ref_bspoanodev(g->nodes, tail)->mmidx;
	uint64_t h = 0xCBF29CE484222325ULL; u8i i;
	for(i = 2; i < g->mmcnt; i++) if(i != skip) h = fnv1a(h, g->memp->buffer + i * g->mmblk, used);
	return h;
}

/* adapter backend -> the oracle's sweep */
static int backend_oracle(void *user, const bsa_row_task_t *tasks, size_t ntasks, const uint8_t *query, uint32_t slen,
		const bsa_sweep_params_t *par, uint8_t *rows_out, size_t nblocks, bsa_sweep_result_t *res){
	ref_poa_t *p = (ref_poa_t*)user;
	bsa_sweep_prog_t pg; uint64_t qoff = 0;
	(void)nblocks;
	pg.first_task = 0; pg.ntasks = (uint32_t)ntasks; pg.first_block = 0; pg.reserved = 0;
	p->sweep(rows_out, tasks, &pg, 1, query, &qoff, &slen, par->rows.mode, par->rows.bandwidth, par->rows.M, par->rows.X,
		par->rows.refbonus, par->rows.gapo1, par->rows.gape1, par->rows.gapo2, par->rows.gape2, par->T, res);
	return 0;
}

void *ref_poa_create(int bandwidth, int bwtrigger, int alnmode, int nrec, int realn, int seqcore, int shuffle,
		int M, int X, int O, int E, int Q, int P, int T, int refbonus, int ksz){
	BSPOAPar par = DEFAULT_BSPOA_PAR;
	ref_poa_t *p = (ref_poa_t*)calloc(1, sizeof(ref_poa_t));
	par.bandwidth = bandwidth; par.bwtrigger = bwtrigger; par.alnmode = alnmode; par.nrec = nrec; par.realn = realn;
	par.seqcore = seqcore; par.shuffle = shuffle;
	par.M = M; par.X = X; par.O = O; par.E = E; par.Q = Q; par.P = P; par.T = T; par.refbonus = refbonus; par.ksz = ksz;
	p->g = init_bspoa(par);
	return p;
}

void ref_poa_destroy(void *vp){
	ref_poa_t *p = (ref_poa_t*)vp;
	free_bspoa(p->g);
	bsa_poa_adapter_free(&p->ad);
	free(p->recs); free(p->tasks); free(p->queries);
	free(p->gnodes); free(p->gedges); free(p->gcands); free(p->gtrace); free(p->cur_trace); free(p->snap); free(p->snapoff);
	free(p);
}

void ref_poa_set_refmode(void *vp, int refmode){ ((ref_poa_t*)vp)->g->par->refmode = refmode; }
void ref_poa_set_cigars(void *vp, const uint32_t *cigs, const uint64_t *coffs){ ((ref_poa_t*)vp)->cigs = cigs; ((ref_poa_t*)vp)->coffs = coffs; }
void ref_poa_set_realn_pass(void *vp, int how){ ((ref_poa_t*)vp)->realn_pass = how; }

static void record_read(ref_poa_t *p, seqalign_result_t rs, int mismatch, uint64_t rows_hash){
	BSPOA *g = p->g;
	poa_read_rec_t *r;
	if(p->nrec == p->caprec){ p->caprec = p->caprec ? p->caprec * 2 : 64; p->recs = (poa_read_rec_t*)realloc(p->recs, p->caprec * sizeof(poa_read_rec_t)); }
	r = p->recs + p->nrec ++;
	memset(r, 0, sizeof(*r));
	r->rs = rs; r->maxscr = g->maxscr; r->maxidx = g->maxidx; r->maxoff = g->maxoff;
	r->bandwidth = g->bandwidth; r->slen = g->slen; r->qb = g->qb; r->nblocks = (uint32_t)g->mmcnt; r->piecewise = (uint32_t)g->piecewise;
	r->rows_hash = rows_hash; r->mismatch = mismatch;
	r->task_off = p->ntasks; r->query_off = p->nq;
	if((p->mode == 5 && (p->record_programs & 1)) || (p->mode == 1 && (p->record_programs & 2))){
#define APPEND(dst, n, cap, src, cnt, type) do { if((n) + (cnt) > (cap)){ (cap) = ((n) + (cnt)) * 2 + 64; (dst) = (type*)realloc((dst), (cap) * sizeof(type)); } \
		memcpy((dst) + (n), (src), (cnt) * sizeof(type)); (n) += (cnt); } while(0)
		r->node_off = p->ngn; r->edge_off = p->nge; r->cand_off = p->ngc; r->trace_off = p->ngt;
		r->nnodes = (uint32_t)p->ad.nnodes; r->nedges = (uint32_t)p->ad.nedges; r->ncands = (uint32_t)p->ad.ncands; r->ntrace = (uint32_t)p->ncur;
		APPEND(p->gnodes, p->ngn, p->capgn, p->ad.nodes, p->ad.nnodes, bsa_poa_node_t);
		APPEND(p->gedges, p->nge, p->capge, p->ad.edges, p->ad.nedges, bsa_poa_edge_t);
		APPEND(p->gcands, p->ngc, p->capgc, p->ad.cands, p->ad.ncands, bsa_poa_cand_t);
		APPEND(p->gtrace, p->ngt, p->capgt, p->cur_trace, p->ncur, bsa_poa_event_t);
#undef APPEND
		if(p->nq + g->slen > p->capq){ p->capq = (p->nq + g->slen) * 2; p->queries = (uint8_t*)realloc(p->queries, p->capq); }
		memcpy(p->queries + p->nq, g->qseq->buffer + g->qb, g->slen);
		p->nq += g->slen;
	} else if(p->mode >= 1 && p->mode <= 4 && p->record_programs){      /* (mode 1: the program the adapter would submit, recorded beside the reference's own sweep) */
		r->ntasks = (uint32_t)p->ad.ntasks;
		if(p->ntasks + p->ad.ntasks > p->captasks){
			p->captasks = (p->ntasks + p->ad.ntasks) * 2;
			p->tasks = (bsa_row_task_t*)realloc(p->tasks, p->captasks * sizeof(bsa_row_task_t));
		}
		memcpy(p->tasks + p->ntasks, p->ad.tasks, p->ad.ntasks * sizeof(bsa_row_task_t));
		p->ntasks += p->ad.ntasks;
		if(p->nq + g->slen > p->capq){ p->capq = (p->nq + g->slen) * 2; p->queries = (uint8_t*)realloc(p->queries, p->capq); }
		memcpy(p->queries + p->nq, g->qseq->buffer + g->qb, g->slen);
		p->nq += g->slen;
	}
}

#ifdef REF_TRACE_RECORD
static void harness_trace_hook(void *user, u4i node, int x, u4i bt){
	ref_poa_t *p = (ref_poa_t*)user;
	if(p->ncur == p->capcur){ p->capcur = p->capcur ? p->capcur * 2 : 4096; p->cur_trace = (bsa_poa_event_t*)realloc(p->cur_trace, p->capcur * sizeof(bsa_poa_event_t)); }
	p->cur_trace[p->ncur].node = node; p->cur_trace[p->ncur].x = x; p->cur_trace[p->ncur].bt = bt; p->ncur ++;
}
#endif
int ref_poa_can_record_trace(void){
#ifdef REF_TRACE_RECORD
	return 1;
#else
	return 0;
#endif
}

/* mode 8: one read through the library's own graph AND through the reference, compared after every step.  mismatch bits: 16 selection, 32 placement,
 * 64 program, 128 result, 256 graph after the surgery, 512 the library declined / failed */
static int graphs_equal(BSPOA *g, bsa_pog_t *pog, uint64_t *nn, uint64_t *ne){
	bsa_poa_graph_export_t x;
	uint32_t n = 0, nr = 0, nedge = 0, hd = 0, tl = 0, i;
	bsa_pog_node_t *nodes; uint32_t *ndoff, *rdlen, *oo, *ot, *oc, *io, *inf;
	int ok = 1;
	if(bsa_poa_graph_export(g, &x) != 0){ fprintf(stderr, "ref_poa_harness: out of memory\n"); abort(); }
	if(bsa_pog_export(pog, &n, &nr, &nedge, &hd, &tl, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL) != BSA_OK){ bsa_poa_graph_export_free(&x); return 0; }
	if(n != x.snap.nnodes || nr != x.snap.nreads || hd != x.snap.head || tl != x.snap.tail || nedge != x.out_off[n]){ bsa_poa_graph_export_free(&x); return 0; }
	nodes = (bsa_pog_node_t*)calloc((size_t)n + 1, sizeof(bsa_pog_node_t)); ndoff = (uint32_t*)calloc((size_t)nr + 1, 4); rdlen = (uint32_t*)calloc((size_t)nr + 1, 4);
	oo = (uint32_t*)calloc((size_t)n + 2, 4); io = (uint32_t*)calloc((size_t)n + 2, 4);
	ot = (uint32_t*)calloc((size_t)nedge + 1, 4); oc = (uint32_t*)calloc((size_t)nedge + 1, 4); inf = (uint32_t*)calloc((size_t)nedge + 1, 4);
	bsa_pog_export(pog, NULL, NULL, NULL, NULL, NULL, nodes, ndoff, rdlen, oo, ot, oc, io, inf);
	for(i = 0; i < n && ok; i++){
		const bsa_pog_node_t *a = nodes + i, *b = x.nodes + i;
		/* (cpos is compared where it is defined for both: the library holds the columns of the nodes it selected and of the reads it placed, the reference
		 * recomputes every node's before each read) */
		if(a->header != b->header || a->next != b->next || a->prev != b->prev || a->pos != b->pos || a->rid != b->rid || a->base != b->base || a->flags != b->flags) ok = 0;
		if(a->header == i && a->cov != b->cov) ok = 0;
	}
	if(memcmp(ndoff, x.ndoff, (size_t)nr * 4) || memcmp(rdlen, x.rdlen, (size_t)nr * 4)) ok = 0;
	if(memcmp(oo, x.out_off, ((size_t)n + 1) * 4) || memcmp(io, x.in_off, ((size_t)n + 1) * 4)) ok = 0;
	if(ok && (memcmp(ot, x.out_to, (size_t)nedge * 4) || memcmp(oc, x.out_cov, (size_t)nedge * 4) || memcmp(inf, x.in_from, (size_t)nedge * 4))) ok = 0;
	*nn += n; *ne += nedge;
	free(nodes); free(ndoff); free(rdlen); free(oo); free(io); free(ot); free(oc); free(inf);
	bsa_poa_graph_export_free(&x);
	return ok;
}

/* one read (realn 0: the whole read rid, rbeg = 0) or one stretch of a read that is already in the graph (realn 1: cut out of both graphs first, bspoa.h:2626-2630) */
static seqalign_result_t poa_align_read_shadow_pog(ref_poa_t *p, u2i rid, int rbeg, int rlen, int realn){
	BSPOA *g = p->g;
	BSPOAPar *par = g->par;
	seqalign_result_t rs, krs;
	bsa_pog_read_t rd;
	bsa_pog_guide_t gd;
	const uint32_t *sel = NULL;
	uint32_t *kcig = NULL;
	int32_t *cps = NULL;
	const bsa_poa_node_t *pn; const bsa_poa_edge_t *pe; const bsa_poa_cand_t *pc; size_t npn = 0, npe = 0, npc = 0;
	const uint8_t *pq = NULL;
	const uint64_t *aux = NULL; size_t naux = 0;
	bsa_sweep_params_t sp;
	bsa_poa_result_t pres;
	const bsa_poa_event_t *pev = NULL;
	bsa_result_t brs;
	u4i head, tail, k;
	u2i rfirst;
	int mismatch = 0, rc, score, lib_ok = 0;
	ZEROS(&rs); ZEROS(&krs); memset(&gd, 0, sizeof(gd)); memset(&brs, 0, sizeof(brs)); memset(&pres, 0, sizeof(pres));
	clear_u8v(g->todels);
	/* ---- the library's selection, then the reference's */
	if(bsa_poa_pog_sync(g, par, &p->ad) != BSA_OK) mismatch |= 512;
	if(p->ad.pog_imports > p->ad.pog_declined) mismatch |= 512;            /* the mirror is only ever re-imported after a read the kernel declined (a whole-read band above 256 columns: a window's first read) */
	if(realn){
		int i;
		if(rid) for(i = rbeg; i < rbeg + rlen; i++) cut_rdnode_bspoa(g, rid, i, BSPOA_RDNODE_CUTALL);
		if(!mismatch){
			if(bsa_pog_cut(p->ad.pog, rid, (uint32_t)rbeg, (uint32_t)rlen) != BSA_OK) mismatch |= 512;
			else if(!graphs_equal(g, p->ad.pog, &p->pog_checked[4], &p->pog_checked[5])) mismatch |= 1024;          /* the two graphs after the cut */
		}
	}
	rc = mismatch ? BSA_E_ARG : bsa_pog_select(p->ad.pog, rid, (uint32_t)rbeg, (uint32_t)rlen, &rd, &sel);
	if(rc != BSA_OK) mismatch |= 512;
	head = get_rdnode_bspoa(g, rid, rbeg - 1)->header;
	tail = get_rdnode_bspoa(g, rid, rbeg + rlen)->header;
	rfirst = (!realn && par->nrec) ? num_max(0, Int(rid) - par->nrec - 1) : 0;
	sel_nodes_bspoa(g, head, tail, rfirst, (!realn && par->nrec) ? rid : MAX_U2);
	if(!(mismatch & 512)){
		if(rd.nhead != head || rd.ntail != tail || rd.nsel != g->sels->size) mismatch |= 16;
		else for(k = 0; k < g->sels->size; k++) if(sel[k] != g->sels->buffer[k]){ mismatch |= 16; break; }
		p->pog_checked[0] += g->sels->size;
	}
	/* ---- the guide alignment (the same call prepare_rd_align_bspoa makes, bspoa.h:2087-2091), the library's placement, then the reference's */
	if(!(mismatch & 512)){
		clear_and_encap_u1v(g->qseq, (u4i)rlen);
		bitseq_basebank(g->seqs->rdseqs, g->seqs->rdoffs->buffer[rid] + rbeg, (u4i)rlen, g->qseq->buffer);
		g->qseq->size = (u4i)rlen;
		gd.reflen = par->refmode ? (uint32_t)g->backbone : (uint32_t)g->cns->size;
		if(par->refmode && g->cges->buffer[rid] > g->cgbs->buffer[rid]){
			/* refmode: the read's SAM CIGAR places the band (bspoa.h:2055-2085); the word behind it is readable (ref_poa_run) */
			gd.sam = 1; gd.cigar = g->cigars->buffer + g->cgbs->buffer[rid]; gd.ncigar = (uint32_t)(g->cges->buffer[rid] - g->cgbs->buffer[rid]);
		} else
		if(bsa_pog_needs_guide(p->ad.pog, gd.reflen)){
			if(par->ksz) krs = kmer_striped_seqedit_pairwise(par->ksz, g->qseq->buffer, g->qseq->size, g->cns->buffer, g->cns->size, g->memp, g->stack, 0);
			else krs = striped_seqedit_pairwise(g->qseq->buffer, g->qseq->size, g->cns->buffer, g->cns->size, par->alnmode, 0, g->memp, g->stack, 0);
			kcig = (uint32_t*)malloc((g->stack->size + 1) * sizeof(uint32_t));
			memcpy(kcig, g->stack->buffer, g->stack->size * sizeof(uint32_t));
			gd.have = 1; gd.qb = krs.qb; gd.qe = krs.qe; gd.tb = krs.tb; gd.te = krs.te; gd.cigar = kcig; gd.ncigar = (uint32_t)g->stack->size;
		}
		cps = (int32_t*)malloc(((size_t)rd.nsel + 1) * sizeof(int32_t));
		for(k = 0; k < rd.nsel; k++) cps[k] = ref_bspoanodev(g->nodes, sel[k])->cpos;
		rc = bsa_pog_place(p->ad.pog, &gd, cps, &rd);
		if(rc != BSA_OK) mismatch |= 512;
	}
	prepare_rd_align_bspoa(g, par, head, tail, rid, rbeg, rbeg + rlen);
	if(!(mismatch & 512)){
		int32_t *rp = (int32_t*)malloc(((size_t)rd.nsel + 1) * sizeof(int32_t));
		if(rd.bandwidth != g->bandwidth || rd.qlen != g->qlen || rd.slen != g->slen || rd.qb != g->qb || rd.qe != g->qe) mismatch |= 32;
		bsa_pog_aux_edges(p->ad.pog, &aux, &naux);
		if(naux != g->todels->size) mismatch |= 32;
		else for(k = 0; k < naux; k++) if(aux[k] != g->todels->buffer[k]){ mismatch |= 32; break; }
		/* band offsets and in-degrees come out with the program; compare it with the binding's own flattening of the reference's graph */
		rc = bsa_pog_program(p->ad.pog, &pn, &npn, &pe, &npe, &pc, &npc, &pq, &sp);
		if(rc != BSA_OK) mismatch |= 512;
		else if(head != tail && g->sels->size >= 2){
			bsa_poa_flatten_graph(g, par, head, tail, &p->ad);
			for(k = 0; k < g->sels->size; k++) ref_bspoanodev(g->nodes, g->sels->buffer[k])->vst = 0;
			if(npn != p->ad.nnodes || npe != p->ad.nedges || npc != p->ad.ncands) mismatch |= 64;
			else if(memcmp(pn, p->ad.nodes, npn * sizeof(bsa_poa_node_t)) || memcmp(pe, p->ad.edges, npe * sizeof(bsa_poa_edge_t)) || memcmp(pc, p->ad.cands, npc * sizeof(bsa_poa_cand_t))) mismatch |= 64;
			if(memcmp(pq, g->qseq->buffer + g->qb, g->slen)) mismatch |= 64;
			if(sp.rows.bandwidth != g->bandwidth || sp.rows.mode != seqalign_mode_type(par->alnmode) || sp.T != par->T) mismatch |= 64;
			p->pog_checked[1] += g->sels->size; p->pog_checked[2] += npn * sizeof(bsa_poa_node_t) + npe * sizeof(bsa_poa_edge_t) + npc * sizeof(bsa_poa_cand_t);
		}
		free(rp);
	}
	/* ---- the DP + walk: the library runs its program through the test's backend and applies the steps to ITS graph; the binding does the same
	 * for the reference's graph (bsa_poa_align_rd_core + bsa_poa_apply_trace, the round-4 path, itself pinned against the reference's own walk) */
	if(!(mismatch & 512) && p->ad.run_graph && head != tail && g->sels->size >= 2){
		rc = bsa_pog_run(p->ad.pog, (bsa_pog_backend_fn)p->ad.run_graph, p->ad.user, &pres, &pev);
		if(rc == BSA_OK && bsa_pog_apply(p->ad.pog, &brs, NULL) == BSA_OK) lib_ok = 1;
		else { bsa_pog_abort(p->ad.pog); p->ad.pog_stale = 1; p->ad.pog_declined ++; }
	} else if(!(mismatch & 512)){ bsa_pog_abort(p->ad.pog); p->ad.pog_stale = 1; p->ad.pog_declined ++; }
	if(lib_ok) p->ad.pog_reads ++;
	score = bsa_poa_align_rd_core(g, par, rid, head, tail, &p->ad);
	if(p->ad.have_trace) rs = bsa_poa_apply_trace(g, par, rid, (u4i)rbeg, head, tail, &p->ad);
	else rs = alignment2graph_bspoa(g, par, rid, (u4i)rbeg, head, tail, g->maxidx, g->maxoff, NULL);
	rs.qb += g->qb; rs.qe += g->qb; rs.score = score;
	for(k=0;k<g->todels->size;k++){
		chg_edge_bspoa(g, ref_bspoanodev(g->nodes, g->todels->buffer[k] >> 32), ref_bspoanodev(g->nodes, g->todels->buffer[k] & MAX_U4), -1, NULL);
	}
	clear_u8v(g->todels);
	if(lib_ok){
		if(!p->ad.have_trace) mismatch |= 128;
		else {
			if(brs.score != rs.score || brs.qb != rs.qb || brs.qe != rs.qe || brs.tb != rs.tb || brs.te != rs.te || brs.mat != rs.mat || brs.mis != rs.mis || brs.ins != rs.ins || brs.del != rs.del) mismatch |= 128;
			if(pres.maxscr != p->ad.res.maxscr || pres.maxidx != p->ad.res.maxidx || pres.maxoff != p->ad.res.maxoff || pres.nevents != p->ad.res.nevents) mismatch |= 128;
			p->pog_checked[3] += (uint64_t)pres.nevents;
		}
		if(!graphs_equal(g, p->ad.pog, &p->pog_checked[4], &p->pog_checked[5])) mismatch |= 256;
	} else if(p->ad.have_trace) mismatch |= 512;          /* the binding's graph form took the read, the library did not */
	free(kcig); free(cps);
	record_read(p, rs, mismatch, 0);
	return rs;
}

/* orchestration of align_rd_bspoa (bspoa.h:2620-2667), realn == 0 entry only (the one end_bspoa uses) */
static seqalign_result_t poa_align_read(ref_poa_t *p, u2i rid){
	BSPOA *g = p->g;
	BSPOAPar *par = g->par;
	seqalign_result_t rs;
	const int rlen = g->seqs->rdlens->buffer[rid];
	u4i head, tail, k;
	u2i rfirst;
	int score, mismatch = 0;
	uint64_t rows_hash = 0;
	clear_u8v(g->todels);
	ZEROS(&rs);
	if(rlen == 0) return rs;
	if(p->mode == 9 || p->mode == 10){
		/* the product's path, as the patched align_rd_bspoa takes it */
		if(bsa_poa_align_rd_pog(g, par, 0, rid, 0, rlen, &p->ad, &rs)){ record_read(p, rs, 0, 0); return rs; }
	}
	if(p->mode == 8) return poa_align_read_shadow_pog(p, rid, 0, rlen, 0);
	head = get_rdnode_bspoa(g, rid, -1)->header;
	tail = get_rdnode_bspoa(g, rid, rlen)->header;
	rfirst = par->nrec ? num_max(0, Int(rid) - par->nrec - 1) : 0;
	{
		bsa_poa_graph_export_t snapx; uint64_t shdr[20]; uint32_t *scig = NULL; int snapping = (p->mode == 5 && (p->record_programs & 4));
		if(snapping){
			seqalign_result_t krs;
			memset(shdr, 0, sizeof(shdr)); ZEROS(&krs);
			if(bsa_poa_graph_export(g, &snapx) != 0){ fprintf(stderr, "ref_poa_harness: out of memory\n"); abort(); }
			shdr[0] = snapx.snap.nnodes; shdr[1] = snapx.snap.nreads; shdr[2] = snapx.out_off[snapx.snap.nnodes]; shdr[6] = g->HEAD; shdr[7] = g->TAIL;
			shdr[13] = g->cns->size; shdr[18] = rid; shdr[19] = (uint64_t)rlen;
			/* the guide alignment prepare_rd_align_bspoa is about to make (bspoa.h:2086-2091): the same call, kept */
			if(par->bwtrigger && head == g->HEAD && tail == g->TAIL && g->cns->size && Int(roundup_times(rlen, WORDSIZE)) > par->bandwidth){
				clear_and_encap_u1v(g->qseq, (u4i)rlen);
				bitseq_basebank(g->seqs->rdseqs, g->seqs->rdoffs->buffer[rid], (u4i)rlen, g->qseq->buffer);
				g->qseq->size = (u4i)rlen;
				if(par->ksz) krs = kmer_striped_seqedit_pairwise(par->ksz, g->qseq->buffer, g->qseq->size, g->cns->buffer, g->cns->size, g->memp, g->stack, 0);
				else krs = striped_seqedit_pairwise(g->qseq->buffer, g->qseq->size, g->cns->buffer, g->cns->size, par->alnmode, 0, g->memp, g->stack, 0);
				scig = (uint32_t*)malloc((g->stack->size + 1) * sizeof(uint32_t));
				memcpy(scig, g->stack->buffer, g->stack->size * sizeof(uint32_t));
				shdr[3] = g->stack->size; shdr[8] = 1; shdr[9] = (uint64_t)(int64_t)krs.qb; shdr[10] = (uint64_t)(int64_t)krs.qe; shdr[11] = (uint64_t)(int64_t)krs.tb; shdr[12] = (uint64_t)(int64_t)krs.te;
			}
		}
		sel_nodes_bspoa(g, head, tail, rfirst, par->nrec ? rid : MAX_U2);
		prepare_rd_align_bspoa(g, par, head, tail, rid, 0, rlen);
		if(snapping){
			shdr[4] = g->sels->size; shdr[5] = g->todels->size; shdr[14] = g->bandwidth; shdr[15] = g->slen; shdr[16] = g->qb; shdr[17] = g->qe;
			snap_record(p, &snapx, shdr, scig, g->sels->buffer, g->todels->buffer);
			bsa_poa_graph_export_free(&snapx); free(scig);
		}
	}
	if(p->mode == 6 || p->mode == 7 || p->mode == 9 || p->mode == 10){
		/* the product's path: graph form, the walk applied by the binding (modes 9 / 10: a read the library's own graph declined) */
		score = bsa_poa_align_rd_core(g, par, rid, head, tail, &p->ad);
		if(p->ad.have_trace){
			rs = bsa_poa_apply_trace(g, par, rid, 0, head, tail, &p->ad);
			rs.qb += g->qb; rs.qe += g->qb; rs.score = score;
			for(k=0;k<g->todels->size;k++){
				chg_edge_bspoa(g, ref_bspoanodev(g->nodes, g->todels->buffer[k] >> 32), ref_bspoanodev(g->nodes, g->todels->buffer[k] & MAX_U4), -1, NULL);
			}
			clear_u8v(g->todels);
			record_read(p, rs, 0, 0);
			return rs;
		}
	} else if(p->mode == 5){
		/* graph form in the shadow of the reference */
		int a_scr = 0, a_idx = -1, a_off = -1, have;
		p->ncur = 0;
		score = bsa_poa_align_rd_core(g, par, rid, head, tail, &p->ad);
		have = p->ad.have_trace;
		if(have){ a_scr = g->maxscr; a_idx = g->maxidx; a_off = g->maxoff; }
		for(k=0;k<g->sels->size;k++) ref_bspoanodev(g->nodes, g->sels->buffer[k])->vst = 0;
		g->maxscr = SEQALIGN_SCORE_MIN; g->maxidx = -1; g->maxoff = -1;
		score = align_rd_bspoacore(g, par, rid, head, tail);
		if(have && (a_scr != g->maxscr || a_idx != g->maxidx || a_off != g->maxoff)) mismatch |= 1;
		rows_hash = hash_node_blocks(g, tail);
#ifdef REF_TRACE_RECORD
		bspoa_trace_hook = harness_trace_hook; bspoa_trace_hook_user = p;
#endif
		rs = alignment2graph_bspoa(g, par, rid, 0, head, tail, g->maxidx, g->maxoff, NULL);
#ifdef REF_TRACE_RECORD
		bspoa_trace_hook = NULL;
		if(have){
			size_t i;
			if((size_t)p->ad.res.nevents != p->ncur) mismatch |= 4;
			else for(i = 0; i < p->ncur; i++){
				const bsa_poa_event_t *m = p->ad.events + i, *r = p->cur_trace + i;
				if(p->ad.nodes[m->node].gnode != r->node || m->x != r->x || m->bt != r->bt){ mismatch |= 4; break; }
			}
			/* the end of the walk is what alignment2graph_bspoa reports as rs.qb / rs.tb (bspoa.h:2307-2311) */
			if(p->ad.res.fin_x != rs.qb - Int(g->qb) || ref_bspoanodev(g->nodes, p->ad.nodes[p->ad.res.fin_node].gnode)->cpos != rs.tb) mismatch |= 8;
		}
#endif
		rs.qb += g->qb; rs.qe += g->qb; rs.score = score;
		for(k=0;k<g->todels->size;k++){
			chg_edge_bspoa(g, ref_bspoanodev(g->nodes, g->todels->buffer[k] >> 32), ref_bspoanodev(g->nodes, g->todels->buffer[k] & MAX_U4), -1, NULL);
		}
		clear_u8v(g->todels);
		if(!have){ p->ad.nnodes = p->ad.nedges = p->ad.ncands = 0; p->ncur = 0; }
		record_read(p, rs, mismatch, rows_hash);
		if(p->nrec){ poa_read_rec_t *r = p->recs + p->nrec - 1; r->fin_gnode = have ? (int32_t)p->ad.nodes[p->ad.res.fin_node].gnode : -1; r->fin_x = have ? p->ad.res.fin_x : -1; r->maxidx_local = have ? p->ad.res.maxidx : -1; }
		return rs;
	}
	if(p->mode == 4 || p->mode == 6 || p->mode == 7 || p->mode == 9 || p->mode == 10){
		if(p->mode == 4) score = bsa_poa_align_rd_core(g, par, rid, head, tail, &p->ad);
	} else if(p->mode >= 2){
		int a_scr, a_idx, a_off;
		const size_t used = (size_t)g->bandwidth * (g->piecewise + 1) + (WORDSIZE + 1) * sizeof(int);
		b1i *mine;
		u8i b;
		score = bsa_poa_align_rd_core(g, par, rid, head, tail, &p->ad);
		a_scr = g->maxscr; a_idx = g->maxidx; a_off = g->maxoff;
		/* run the reference's own sweep on the same graph state and compare */
		mine = (b1i*)malloc(g->mmcnt * g->mmblk);
		memcpy(mine, g->memp->buffer, g->mmcnt * g->mmblk);
		for(k=0;k<g->sels->size;k++) ref_bspoanodev(g->nodes, g->sels->buffer[k])->vst = 0;
		g->maxscr = SEQALIGN_SCORE_MIN; g->maxidx = -1; g->maxoff = -1;
		score = align_rd_bspoacore(g, par, rid, head, tail);
		if(a_scr != g->maxscr || a_idx != g->maxidx || a_off != g->maxoff) mismatch |= 1;
		for(b=2;b<g->mmcnt;b++){
			if(b == ref_bspoanodev(g->nodes, tail)->mmidx) continue;
			if(memcmp(mine + b * g->mmblk, g->memp->buffer + b * g->mmblk, used)){ mismatch |= 2; break; }
		}
		rows_hash = hash_node_blocks(g, tail);
		/* continue with the backend's rows: the traceback must work from them */
		memcpy(g->memp->buffer, mine, g->mmcnt * g->mmblk);
		g->maxscr = a_scr; g->maxidx = a_idx; g->maxoff = a_off;
		score = a_scr;
		free(mine);
	} else {
		struct timespec t0, t1;
		size_t i;
		/* count the work first (the flattening walk is the same traversal), then time the reference's sweep alone */
		bsa_poa_flatten(g, par, head, tail, &p->ad);
		for(i=0;i<p->ad.ntasks;i++){
			if(p->ad.tasks[i].op == BSA_ROW_OP_UPDATE) p->core_updates ++;
			else if(p->ad.tasks[i].op == BSA_ROW_OP_MERGE) p->core_merges ++;
		}
		for(k=0;k<g->sels->size;k++) ref_bspoanodev(g->nodes, g->sels->buffer[k])->vst = 0;
		p->ad.nnodes = p->ad.nedges = p->ad.ncands = 0; p->ncur = 0;
		if((p->record_programs & 2) && head != tail && g->sels->size >= 2){      /* the graph-form program the binding would build, recorded beside the reference's own sweep */
			bsa_poa_flatten_graph(g, par, head, tail, &p->ad);
			for(k=0;k<g->sels->size;k++) ref_bspoanodev(g->nodes, g->sels->buffer[k])->vst = 0;
		}
		clock_gettime(CLOCK_MONOTONIC, &t0);
		score = align_rd_bspoacore(g, par, rid, head, tail);
		clock_gettime(CLOCK_MONOTONIC, &t1);
		p->core_seconds += (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
		rows_hash = hash_node_blocks(g, tail);
	}
	rs = alignment2graph_bspoa(g, par, rid, 0, head, tail, g->maxidx, g->maxoff, NULL);
	rs.qb += g->qb;
	rs.qe += g->qb;
	rs.score = score;
	for(k=0;k<g->todels->size;k++){
		chg_edge_bspoa(g, ref_bspoanodev(g->nodes, g->todels->buffer[k] >> 32), ref_bspoanodev(g->nodes, g->todels->buffer[k] & MAX_U4), -1, NULL);
	}
	clear_u8v(g->todels);
	record_read(p, rs, mismatch, rows_hash);
	return rs;
}

/* orchestration of end_bspoa (bspoa.h:4722-4776) */
static void poa_finish(ref_poa_t *p){
	BSPOA *g = p->g;
	u4i n0;
	u2i rid;
	int round;
	clear_u1v(g->cns); clear_u1v(g->qlt); clear_u1v(g->alt);
	if(g->par->refmode){
		n0 = g->seqs->rdlens->buffer[0];
		resize_u1v(g->cns, n0); resize_u1v(g->qlt, n0); resize_u1v(g->alt, n0);
		bitseq_basebank(g->seqs->rdseqs, g->seqs->rdoffs->buffer[0], n0, g->cns->buffer);
		memset(g->qlt->buffer, 0, n0);
		memset(g->alt->buffer, 0, n0);
	}
	if(g->seqs->nseq <= 1) return;
	if(g->par->shuffle) shuffle_reads_by_kmers_bspoa(g);
	g->nmsa = g->par->seqcore ? num_min(g->seqs->nseq, g->par->seqcore) : g->seqs->nseq;
	for(rid=0;rid<g->seqs->nseq;rid++) _add_read_bspoa_core(g, rid);
	g->nrds = 1;
	for(rid=1;rid<g->nmsa;rid++){
		if(!g->par->refmode && g->par->bwtrigger){
			msa_bspoa(g);
			simple_cns_bspoa(g);
		}
		poa_align_read(p, rid);
		g->nrds ++;
	}
	if(p->mode == 8 && p->realn_pass){
		/* the realn entry of align_rd_bspoa (the reference's own caller of it, remsa_lsps_bspoa bspoa.h:5463-5556, is compiled out of main.c): a stretch of
		 * every aligned read is cut out of the graph and aligned again, against every read of the window */
		for(rid=1;rid<g->nmsa;rid++){
			const int len = (int)g->seqs->rdlens->buffer[rid];
			const int rb = p->realn_pass == 2 ? 0 : len / 4, rl = p->realn_pass == 2 ? len : len / 2;
			if(rl > 0) poa_align_read_shadow_pog(p, rid, rb, rl, 1);
		}
	}
	if(p->mode != 8 && p->realn_pass){
		/* the same pass outside the shadow: the product's path (modes 9 / 10: the patched align_rd_bspoa -- the reference's cut, then bsa_poa_align_rd_pog with
		 * its realn flag) or the untouched reference function */
		for(rid=1;rid<g->nmsa;rid++){
			const int len = (int)g->seqs->rdlens->buffer[rid];
			const int rb = p->realn_pass == 2 ? 0 : len / 4, rl = p->realn_pass == 2 ? len : len / 2;
			seqalign_result_t rs;
			int i;
			if(rl <= 0) continue;
			if(p->mode == 9 || p->mode == 10){
				for(i = rb; i < rb + rl; i++) cut_rdnode_bspoa(g, rid, i, BSPOA_RDNODE_CUTALL);
				clear_u8v(g->todels);
				if(bsa_poa_align_rd_pog(g, g->par, 1, rid, rb, rl, &p->ad, &rs)){ record_read(p, rs, 0, 0); continue; }
			}
			rs = align_rd_bspoa(g, g->par, 1, rid, rb, rl);
			record_read(p, rs, 0, 0);
		}
	}
	if(p->mode == 5 && (p->record_programs & 4)){
		/* the graph the last surgery left */
		bsa_poa_graph_export_t x; uint64_t hdr[20];
		memset(hdr, 0, sizeof(hdr));
		if(bsa_poa_graph_export(g, &x) != 0){ fprintf(stderr, "ref_poa_harness: out of memory\n"); abort(); }
		hdr[0] = x.snap.nnodes; hdr[1] = x.snap.nreads; hdr[2] = x.out_off[x.snap.nnodes]; hdr[6] = g->HEAD; hdr[7] = g->TAIL; hdr[18] = (uint64_t)-1;
		snap_record(p, &x, hdr, NULL, NULL, NULL);
		bsa_poa_graph_export_free(&x);
	}
	for(round=0;round<g->par->realn;round++){
		msa_bspoa(g);
		cns_bspoa(g);
		if(g->par->editbw < 0) remsa_edits_bspoa(g, - g->par->editbw);
		else remsa_pedits_bspoa(g, g->par->editbw / 2, 1, (round + 1 == g->par->realn));
	}
	if(g->par->shuffle) restore_rd_orders_bspoa(g);
	msa_bspoa(g);
	cns_bspoa(g);
}

/* reads: one base per byte (0..3), read k at reads[offs[k] .. +lens[k]).  mode as described in the file header.
 * sweep_fn = address of orc_sweep_run (mode 2).  Returns the number of reads whose backend results differed from the
 * reference's core (mode 2), else 0. */
int ref_poa_run(void *vp, const uint8_t *reads, const uint64_t *offs, const uint32_t *lens, int nreads, int mode,
		void *sweep_fn, int record_programs){
	ref_poa_t *p = (ref_poa_t*)vp;
	BSPOA *g = p->g;
	char *buf;
	uint32_t maxlen = 0;
	int k, bad = 0;
	size_t i;
	p->mode = mode; p->record_programs = record_programs;
	p->nrec = 0; p->ntasks = 0; p->nq = 0; p->ngn = p->nge = p->ngc = p->ngt = 0; p->nsnap = 0; p->nsnaprec = 0;
	p->core_seconds = 0; p->core_updates = 0; p->core_merges = 0;
	p->sweep = (orc_sweep_fn)sweep_fn;
	bsa_poa_adapter_free(&p->ad);
	if(mode == 3) bsa_poa_adapter_init(&p->ad, bsa_poa_backend_hip, g_device_ctx);
	else if(mode == 4) bsa_poa_adapter_init(&p->ad, g_batch_submit, g_batcher);
	else if(mode == 5 || mode == 6 || mode == 8 || mode == 9){
		/* graph form; reads it declines (whole-read bands) take the rows form: the oracle's sweep on the CPU, the device's otherwise */
		if(g_graph_backend) bsa_poa_adapter_init_graph(&p->ad, g_graph_backend, sweep_fn ? backend_oracle : bsa_poa_backend_hip, sweep_fn ? (void*)p : g_device_ctx);
		else bsa_poa_adapter_init_graph(&p->ad, bsa_poa_graph_backend_hip, bsa_poa_backend_hip, g_device_ctx);
		if(g_graph_backend && !sweep_fn) p->ad.user = g_graph_backend_user;
	} else if(mode == 7 || mode == 10) bsa_poa_adapter_init_graph(&p->ad, g_batch_submit_graph, g_batch_submit, g_batcher);
	else bsa_poa_adapter_init(&p->ad, backend_oracle, p);
	p->ad.use_pog = (mode == 8 || mode == 9 || mode == 10);          /* (mode 10 = mode 7, many windows through the batcher, on the library's own graph) */
	memset(p->pog_checked, 0, sizeof(p->pog_checked));
	for(k = 0; k < nreads; k++) if(lens[k] > maxlen) maxlen = lens[k];
	buf = (char*)malloc(maxlen + 1);
	beg_bspoa(g);
	for(k = 0; k < nreads; k++){
		for(i = 0; i < lens[k]; i++) buf[i] = "ACGT"[reads[offs[k] + i] & 3];
		buf[lens[k]] = 0;
		if(p->cigs && p->coffs) push_bspoacore(g, buf, lens[k], (u4i*)(p->cigs + p->coffs[k]), (u4i)(p->coffs[k + 1] - p->coffs[k]));
		else push_bspoa(g, buf, lens[k]);
	}
	if(p->cigs){ encap_u4v(g->cigars, 1); g->cigars->buffer[g->cigars->size] = 0; }          /* (bspoa.h:2073 reads the word behind a read's CIGAR) */
	free(buf);
	if(mode == 0) end_bspoa(g);
	else poa_finish(p);
	for(i = 0; i < p->nrec; i++) if(p->recs[i].mismatch) bad ++;
	return bad;
}

/* ---- many windows in lock-step (mode 4) ---- */
#include <pthread.h>
typedef struct {
	void *handle;
	const uint8_t *reads; const uint64_t *offs; const uint32_t *lens;
	int nreads, mode, record, rc;
} many_job_t;

static batch_leave_fn g_batch_enter = NULL;
void ref_poa_set_batcher_enter(void *enter_addr){ g_batch_enter = (batch_leave_fn)enter_addr; }
static void *many_thread(void *vp){
	many_job_t *j = (many_job_t*)vp;
	if((j->mode == 4 || j->mode == 7 || j->mode == 10) && g_batch_enter) g_batch_enter(g_batcher);
	j->rc = ref_poa_run(j->handle, j->reads, j->offs, j->lens, j->nreads, j->mode, NULL, j->record);
	if((j->mode == 4 || j->mode == 7 || j->mode == 10) && g_batch_leave) g_batch_leave(g_batcher);       /* this window submits nothing more */
	return NULL;
}

typedef struct { many_job_t *jobs; int nwin, t, nt; } pool_arg_t;
static void *pool_thread(void *vp){
	pool_arg_t *a = (pool_arg_t*)vp; int k;
	for(k = a->t; k < a->nwin; k += a->nt) many_thread(&a->jobs[k]);
	return NULL;
}

/* window w aligns reads first[w] .. first[w] + count[w] - 1 (indices into offs / lens); handles[w] from ref_poa_create.
 * mode 4: device batcher (one thread per window, all alive at once); mode 0 / 1: the same windows on `threads` host
 * threads with the reference's own core, for the CPU side of the comparison.  Returns 0 or the number of windows that failed. */
int ref_poa_run_many(void **handles, int nwin, const uint8_t *reads, const uint64_t *offs, const uint32_t *lens,
		const int *first, const int *count, int mode, int threads, int record){
	many_job_t *jobs = (many_job_t*)calloc((size_t)nwin, sizeof(many_job_t));
	pthread_t *th = (pthread_t*)calloc((size_t)nwin, sizeof(pthread_t));
	int w, bad = 0;
	if((mode == 4 || mode == 7 || mode == 10) && (!g_batch_submit || !g_batcher)){ free(jobs); free(th); return -1; }
	if((mode == 7 || mode == 10) && !g_batch_submit_graph){ free(jobs); free(th); return -1; }
	cal_permutation_bspoa(MAX_LOG_CACHE, 0);                             /* fill the reference's lazily built log table before any thread reads it (bspoa.h:3391-3401) */
	for(w = 0; w < nwin; w++){
		jobs[w].handle = handles[w]; jobs[w].reads = reads; jobs[w].offs = offs + first[w]; jobs[w].lens = lens + first[w];
		jobs[w].nreads = count[w]; jobs[w].mode = mode; jobs[w].record = record;
	}
	if(mode == 4 || mode == 7 || mode == 10){
		/* through the batcher: since it runs whatever is pending (round 4) the windows need not all be alive at once -- a pool of
		 * BSA_POA_POOL threads (default 256) takes them one after the other, which keeps the working set to that many graphs.  With
		 * BSA_POA_BATCH_MIN=all (lock-step) every window keeps a thread of its own. */
		const char *pe = getenv("BSA_POA_POOL"), *le = getenv("BSA_POA_BATCH_MIN");
		const int pool = (le && strcmp(le, "all") == 0) ? nwin : (pe && atoi(pe) > 0) ? atoi(pe) : 256;
		threads = pool < nwin ? pool : nwin;
	}
	if(threads >= nwin){
		for(w = 0; w < nwin; w++) pthread_create(&th[w], NULL, many_thread, &jobs[w]);
		for(w = 0; w < nwin; w++) pthread_join(th[w], NULL);
	} else {
		/* a bounded pool: thread t takes windows t, t + threads, ... */
		int t, nt = threads > 0 ? threads : 1;
		pool_arg_t *pa = (pool_arg_t*)calloc((size_t)nt, sizeof(pool_arg_t));
		for(t = 0; t < nt; t++){ pa[t].jobs = jobs; pa[t].nwin = nwin; pa[t].t = t; pa[t].nt = nt; pthread_create(&th[t], NULL, pool_thread, &pa[t]); }
		for(t = 0; t < nt; t++) pthread_join(th[t], NULL);
		free(pa);
	}
	for(w = 0; w < nwin; w++) if(jobs[w].rc) bad ++;
	free(jobs); free(th);
	return bad;
}

/* ---- result access ---- */
uint32_t ref_poa_cns_len(void *vp){ return (uint32_t)((ref_poa_t*)vp)->g->cns->size; }
void ref_poa_cns(void *vp, uint8_t *cns, uint8_t *qlt, uint8_t *alt){
	BSPOA *g = ((ref_poa_t*)vp)->g;
	if(cns) memcpy(cns, g->cns->buffer, g->cns->size);
	if(qlt) memcpy(qlt, g->qlt->buffer, g->qlt->size);
	if(alt) memcpy(alt, g->alt->buffer, g->alt->size);
}
uint64_t ref_poa_msa_hash(void *vp, uint32_t *ncols, uint32_t *nrows){
	BSPOA *g = ((ref_poa_t*)vp)->g;
	const u4i mrow = g->seqs->nseq + 3;
	uint64_t h = 0xCBF29CE484222325ULL; u4i c;
	if(ncols) *ncols = (uint32_t)g->msaidxs->size;
	if(nrows) *nrows = mrow;
	for(c = 0; c < g->msaidxs->size; c++) h = fnv1a(h, g->msacols->buffer + (size_t)g->msaidxs->buffer[c] * mrow, g->seqs->nseq);
	return h;
}
uint32_t ref_poa_nrec(void *vp){ return (uint32_t)((ref_poa_t*)vp)->nrec; }
/* out: 10 ints of rs, maxscr, maxidx, maxoff, bandwidth, slen, qb, nblocks, ntasks, piecewise, mismatch = 20 int32; hash separately */
void ref_poa_rec(void *vp, uint32_t k, int32_t *out, uint64_t *rows_hash, uint64_t *task_off, uint64_t *query_off){
	const poa_read_rec_t *r = ((ref_poa_t*)vp)->recs + k;
	memcpy(out, &r->rs, 10 * sizeof(int32_t));
	out[10] = r->maxscr; out[11] = r->maxidx; out[12] = r->maxoff; out[13] = (int32_t)r->bandwidth; out[14] = (int32_t)r->slen;
	out[15] = (int32_t)r->qb; out[16] = (int32_t)r->nblocks; out[17] = (int32_t)r->ntasks; out[18] = (int32_t)r->piecewise; out[19] = r->mismatch;
	*rows_hash = r->rows_hash; *task_off = r->task_off; *query_off = r->query_off;
}
uint64_t ref_poa_ntasks(void *vp){ return ((ref_poa_t*)vp)->ntasks; }
uint64_t ref_poa_nquery_bytes(void *vp){ return ((ref_poa_t*)vp)->nq; }
void ref_poa_programs(void *vp, void *tasks, uint8_t *queries){
	ref_poa_t *p = (ref_poa_t*)vp;
	memcpy(tasks, p->tasks, p->ntasks * sizeof(bsa_row_task_t));
	memcpy(queries, p->queries, p->nq);
}
uint64_t ref_poa_block_bytes(void *vp){ return ((ref_poa_t*)vp)->g->mmblk; }
void ref_poa_core_stats(void *vp, double *seconds, uint64_t *updates, uint64_t *merges){
	ref_poa_t *p = (ref_poa_t*)vp;
	*seconds = p->core_seconds; *updates = p->core_updates; *merges = p->core_merges;
}

/* ---- the anti-diagonal u8 DP of remsa_pedits: the REAL maxmat_dp_diag_rowcal_init / _prepare /
 * maxmat_dp_diag_rowcal (bspoa.h:3752-3896) driven by the loop of remsa_pedit_rd_bspoacore (bspoa.h:3925-3935) on arrays the
 * caller supplies in the layout of remsa_pedits_bspoa (bspoa.h:4213-4233): seq0 / seq1 / mats[s][b] point at logical index 0
 * of planes that carry bandwidth / 2 bytes of padding in front and behind; matrix[p] holds (2 mlen + 1) rows of
 * 16 W + 2 bytes.  Only the rows the loop writes (2 mbeg .. 2 mend - 1) are touched. */
void ref_diagdp_fill(uint8_t *seq0, uint8_t *seq1, uint8_t *m00, uint8_t *m01, uint8_t *m02, uint8_t *m03,
		uint8_t *m10, uint8_t *m11, uint8_t *m12, uint8_t *m13, int mlen, int mbeg, int mend, int W, uint8_t *matrix0, uint8_t *matrix1){
	u1i *matrix[2] = {matrix0, matrix1}, *_seqs[2] = {seq0, seq1}, *_mats[2][4] = {{m00, m01, m02, m03}, {m10, m11, m12, m13}};
	u1i *rows[2][2], *seqs[2], *mats[2][4];
	int i, x, y, dir;
	MM_EPI8_ALL0 = mm_set1_epi8(0); MM_EPI8_ALL1 = mm_set1_epi8(1); MM_EPI8_ALL2 = mm_set1_epi8(2); MM_EPI8_ALL3 = mm_set1_epi8(3);   /* bspoa.h:4192-4195 */
	maxmat_dp_diag_rowcal_init(W, mbeg, matrix);
	x = y = mbeg;
	for(i = x + y;; i++){
		dir = (i & 0x1);
		maxmat_dp_diag_rowcal_prepare(x, y, mlen, W, matrix, _seqs, _mats, rows, seqs, mats);
		maxmat_dp_diag_rowcal(W, dir, rows, seqs, mats);
		if(dir) y ++; else x ++;
		if(x >= mend) break;
	}
}

/* ---- the reference's MSA writers on a finished window (tests of bsa_msa_* in include/bsalign_msa.h) ----
 * dump_binary_msa_bspoa bspoa.h:1555-1586, print_msa_bspoa bspoa.h:1491-1553, through open_memstream */
void ref_poa_msa_dims(void *vp, uint32_t *mlen, uint32_t *mrow, uint32_t *nrds, uint64_t *cols_bytes, uint32_t *nvar){
	BSPOA *g = ((ref_poa_t*)vp)->g;
	if(mlen) *mlen = (uint32_t)g->msaidxs->size;
	if(mrow) *mrow = (uint32_t)g->seqs->nseq + 3;
	if(nrds) *nrds = (uint32_t)g->nrds;
	if(cols_bytes) *cols_bytes = (uint64_t)g->msacols->size;
	if(nvar) *nvar = (uint32_t)g->var->size;
}
void ref_poa_msa_cols(void *vp, uint8_t *cols, uint32_t *idxs, uint32_t *var_mpos){
	BSPOA *g = ((ref_poa_t*)vp)->g; u4i i;
	if(cols) memcpy(cols, g->msacols->buffer, g->msacols->size);
	if(idxs) memcpy(idxs, g->msaidxs->buffer, g->msaidxs->size * sizeof(u4i));
	if(var_mpos) for(i = 0; i < g->var->size; i++) var_mpos[i] = ref_bspoavarv(g->var, i)->mpos;
}
uint64_t ref_poa_msa_binary(void *vp, const char *meta, uint32_t metalen, uint8_t *out, uint64_t cap){
	BSPOA *g = ((ref_poa_t*)vp)->g;
	char *buf = NULL; size_t len = 0;
	FILE *f = open_memstream(&buf, &len);
	dump_binary_msa_bspoa(g, (char*)meta, metalen, f);
	fclose(f);
	if(out && len <= cap) memcpy(out, buf, len);
	free(buf);
	return (uint64_t)len;
}
uint64_t ref_poa_msa_text(void *vp, const char *label, uint32_t mbeg, uint32_t mend, uint32_t linewidth, uint8_t *out, uint64_t cap){
	BSPOA *g = ((ref_poa_t*)vp)->g;
	char *buf = NULL; size_t len = 0;
	FILE *f = open_memstream(&buf, &len);
	print_msa_bspoa(g, label, mbeg, mend, linewidth, 0, f);
	fclose(f);
	if(out && len <= cap) memcpy(out, buf, len);
	free(buf);
	return (uint64_t)len;
}
/* the reference's loader on a binary MSA: returns its return code, and what post_load_binary_msa_bspoa derives */
int ref_msa_load_binary(const uint8_t *in, uint64_t len, uint32_t *nrds, uint32_t *mlen, uint8_t *cols, uint64_t cols_cap,
		uint8_t *cns, uint8_t *qlt, uint8_t *alt, uint32_t *clen, char *meta, uint32_t meta_cap, uint32_t *metalen){
	BSPOAPar par = DEFAULT_BSPOA_PAR;
	BSPOA *g = init_bspoa(par);
	String *md = init_string(64);
	FILE *f = fmemopen((void*)in, (size_t)len, "rb");
	int rc = load_binary_msa_bspoa(g, f, md);
	fclose(f);
	if(rc == 0){
		if(nrds) *nrds = g->nrds;
		if(mlen) *mlen = (uint32_t)g->msaidxs->size;
		if(cols && g->msacols->size <= cols_cap) memcpy(cols, g->msacols->buffer, g->msacols->size);
		if(clen) *clen = (uint32_t)g->cns->size;
		if(cns) memcpy(cns, g->cns->buffer, g->cns->size);
		if(qlt) memcpy(qlt, g->qlt->buffer, g->qlt->size);
		if(alt) memcpy(alt, g->alt->buffer, g->alt->size);
		if(metalen) *metalen = (uint32_t)md->size;
		if(meta && md->size <= meta_cap) memcpy(meta, md->string, md->size);
	}
	free_string(md);
	free_bspoa(g);
	return rc;
}

/* ---- graph-form records (mode 5): per read the program the binding built and the steps the REFERENCE's walk took ---- */
/* out: node_off, edge_off, cand_off, trace_off (u64 x 4), then nnodes, nedges, ncands, ntrace, fin_gnode, fin_x, maxidx_local as int64 */
void ref_poa_graph_rec(void *vp, uint32_t k, int64_t *out){
	const poa_read_rec_t *r = ((ref_poa_t*)vp)->recs + k;
	out[0] = (int64_t)r->node_off; out[1] = (int64_t)r->edge_off; out[2] = (int64_t)r->cand_off; out[3] = (int64_t)r->trace_off;
	out[4] = r->nnodes; out[5] = r->nedges; out[6] = r->ncands; out[7] = r->ntrace; out[8] = r->fin_gnode; out[9] = r->fin_x; out[10] = r->maxidx_local;
}
void ref_poa_graph_sizes(void *vp, uint64_t *out){
	ref_poa_t *p = (ref_poa_t*)vp;
	out[0] = p->ngn; out[1] = p->nge; out[2] = p->ngc; out[3] = p->ngt;
}
void ref_poa_graph_data(void *vp, void *nodes, void *edges, void *cands, void *trace){
	ref_poa_t *p = (ref_poa_t*)vp;
	if(nodes) memcpy(nodes, p->gnodes, p->ngn * sizeof(bsa_poa_node_t));
	if(edges) memcpy(edges, p->gedges, p->nge * sizeof(bsa_poa_edge_t));
	if(cands) memcpy(cands, p->gcands, p->ngc * sizeof(bsa_poa_cand_t));
	if(trace) memcpy(trace, p->gtrace, p->ngt * sizeof(bsa_poa_event_t));
}
void ref_poa_form_counts(void *vp, uint64_t *graph_reads, uint64_t *rows_reads){
	ref_poa_t *p = (ref_poa_t*)vp;
	*graph_reads = p->ad.graph_reads; *rows_reads = p->ad.rows_reads;
}
/* record_programs & 4: the snapshot records (layout above snap_record) */
uint64_t ref_poa_snap_count(void *vp){ return ((ref_poa_t*)vp)->nsnaprec; }
uint64_t ref_poa_snap_bytes(void *vp){ return ((ref_poa_t*)vp)->nsnap; }
void ref_poa_snap_data(void *vp, uint8_t *blob, uint64_t *offs){
	ref_poa_t *p = (ref_poa_t*)vp;
	if(blob) memcpy(blob, p->snap, p->nsnap);
	if(offs) memcpy(offs, p->snapoff, p->nsnaprec * sizeof(uint64_t));
}
/* modes 8 - 10: reads through the library's own graph / re-imports of it / reads it declined, and what mode 8 compared */
void ref_poa_pog_counts(void *vp, uint64_t *out){
	ref_poa_t *p = (ref_poa_t*)vp; int k;
	out[0] = p->ad.pog_reads; out[1] = p->ad.pog_imports; out[2] = p->ad.pog_declined;
	for(k = 0; k < 6; k++) out[3 + k] = p->pog_checked[k];
}
void ref_poa_pog_seconds(void *vp, double *out){
	ref_poa_t *p = (ref_poa_t*)vp; int k;
	for(k = 0; k < 4; k++) out[k] = p->ad.pog_seconds[k];
	for(k = 0; k < 5; k++) out[4 + k] = 0;
	if(p->ad.pog) bsa_pog_seconds(p->ad.pog, out + 4);
}
/* the graph as it stands (after ref_poa_run: the finished window's; the fixtures of tests/golden/make_golden_poa_pog.py record it BEFORE reads through
 * ref_poa_snapshot_hook) */
void ref_poa_binding_seconds(void *vp, double *out){ ref_poa_t *p = (ref_poa_t*)vp; out[0] = p->ad.seconds[0]; out[1] = p->ad.seconds[1]; out[2] = p->ad.seconds[2]; }

/* ---- consensus calling (tests of bsa_msa_call_consensus in include/bsalign_msa.h): the REAL cns_bspoa (bspoa.h:3457-3733) re-run on the finished
 * window's MSA; its inputs besides the columns */
double ref_poa_cns_call(void *vp){ return cns_bspoa(((ref_poa_t*)vp)->g); }
void ref_poa_cns_inputs(void *vp, uint32_t *nmsa, uint32_t *nrds, uint32_t *nall, float *par7){
	BSPOA *g = ((ref_poa_t*)vp)->g;
	*nmsa = g->nmsa; *nrds = g->nrds; *nall = (uint32_t)g->seqs->nseq;
	par7[0] = g->par->psub; par7[1] = g->par->pins; par7[2] = g->par->pdel; par7[3] = g->par->piex; par7[4] = g->par->pdex; par7[5] = g->par->hins; par7[6] = g->par->hdel;
}
