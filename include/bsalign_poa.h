/*
 * bsalign_poa.h -- C-ABI of the POA's own graph surface in libbsalign_hip.so.
 *
 * The reference keeps a POA window in a BSPOA object (bspoa.h:103-140: nodes in rings of aligned bases, edge pairs in
 * coverage-ordered lists) and aligns a read against it in five steps (align_rd_bspoa, bspoa.h:2620-2667):
 *
 *     sel_nodes_bspoa           bspoa.h:1887-2020   which nodes the read is aligned against, in-degrees, auxiliary edges
 *     prepare_rd_align_bspoa    bspoa.h:2022-2230   band width, band offset (rpos) of every node from a guide alignment
 *     align_rd_bspoacore        bspoa.h:2515-2618   the banded DP over the sub-graph                 (device: k_poa_wf)
 *     alignment2graph_bspoa     bspoa.h:2274-2513   the walk back (device: k_poa_wf) and the graph surgery it causes
 *     chg_edge_bspoa(.., -1)    bspoa.h:2655-2657   auxiliary edges removed
 *
 * bsa_pog_t is this library's OWN graph container for that path -- flat arrays, node indices equal to the reference's
 * (read r's base p is node ndoff[r] + p, bspoa.h:408-410) so that a reference-side binding can translate nothing --, with
 * every step but the DP itself as host code of the library: bsa_pog_select (P2), bsa_pog_place (P3), bsa_pog_program (the
 * sub-graph as the kernel's nodes / in-edges / candidates, in the order the reference's sweep completes them),
 * bsa_pog_run (P4 / P5 / the walk of P6 on the device) and bsa_pog_apply (the surgery of P6: merging matched bases into
 * their rings, column positions, chaining the read, auxiliary edges removed).  The graph therefore evolves inside the
 * library from read to read; nothing of the reference's sel_nodes / prepare_rd_align / align_rd_bspoacore /
 * alignment2graph runs, and no graph is exported from the reference between reads.
 *
 * What stays outside (out of scope): the MSA column order and the running consensus (msa_bspoa,
 * simple_cns_bspoa) -- a caller supplies, per read, the column position `cpos` of the selected nodes and the guide alignment of
 * the read against its consensus (a seqalign_result_t + CIGAR: kmer_striped_seqedit_pairwise, or this library's
 * bsa_kmer_edit_batch).  refmode (read 0 a reference sequence, the other reads with their SAM CIGARs against it, bspoa.h:2039-2085):
 * the caller passes the backbone's length as the guide's `reflen` and the read's SAM CIGAR with `sam` set.  Re-alignment of a stretch that is
 * already in the graph (the realn entry of align_rd_bspoa, bspoa.h:2626-2630): bsa_pog_cut, then the five steps as for a new read.
 *
 * Plain C types only.  Every function returns BSA_OK (0) or a negative BSA_E_* code of bsalign_hip.h; no C++ exception leaves
 * the library (a failed host allocation is BSA_E_NOMEM).  After BSA_E_NOMEM from bsa_pog_apply the graph may hold half of a
 * read's surgery: bsa_pog_clear (or bsa_pog_import of the caller's copy) before going on; after it from any earlier step
 * bsa_pog_abort takes the read's auxiliary edges back and leaves the graph as it was.
 */
#ifndef BSALIGN_POA_H
#define BSALIGN_POA_H

#include "bsalign_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bsa_pog bsa_pog_t;

/* BSPOAPar (bspoa.h:55-77), the fields the alignment path reads */
typedef struct {
	int32_t alnmode;              /* SEQALIGN_MODE_* (0 global, 1 overlap, 2 extend) */
	int32_t bandwidth;            /* par->bandwidth: 0 = the whole read */
	int32_t bwtrigger;            /* place bands by a guide alignment against the consensus (bspoa.h:2054) */
	int32_t nrec;                 /* align against the last nrec + 1 reads only (bspoa.h:2636-2642); 0 = all */
	int32_t seqcore;              /* reads in the primary MSA (g->nmsa = min(reads, seqcore), bspoa.h:4741-4745); 0 = all */
	int32_t M, X, O, E, Q, P, T;  /* scores */
	int32_t refbonus;
} bsa_pog_params_t;

/* a node as import / export sees it (bspoanode_t, bspoa.h:28-38, without its per-alignment scratch) */
#define BSA_POG_F_BLESS 1u        /* bless: the homopolymer-free profile's bonus applies to its ring (bspoa.h:1965-1973) */
#define BSA_POG_F_RDC   2u        /* rdc: chained to the previous base of its read */
#define BSA_POG_F_RDD   4u        /* rdd: chained to the next one */
#define BSA_POG_F_REF   8u
typedef struct {
	uint32_t header, next, prev;  /* the ring of bases aligned to each other; header = its representative, which owns the edges */
	int32_t  pos, cpos;           /* position on the read (-1 / len: its two sentinels), column on the consensus */
	uint16_t rid, cov;            /* read, number of bases in the ring (kept on the header) */
	uint8_t  base, flags;         /* 0..3, 4 = sentinel; BSA_POG_F_* */
	uint16_t reserved;
} bsa_pog_node_t;                 /* 28 bytes */
/* the whole graph as flat arrays: nodes, reads, and every header's out- and in-edges IN LIST ORDER (the reference keeps them
 * ordered by coverage with ties in insertion order, bspoa.h:464-494; the order decides the sweep's visiting order and the
 * walk's tie rule, so it is part of the state).  out_off / in_off have nnodes + 1 entries. */
typedef struct {
	uint32_t nnodes, nreads, head, tail;
	const bsa_pog_node_t *nodes;
	const uint32_t *ndoff;        /* node index of position 0 of every read */
	const uint32_t *rdlen;
	const uint32_t *out_off, *out_to, *out_cov;
	const uint32_t *in_off, *in_from;
} bsa_pog_snapshot_t;

/* the guide of bspoa.h:2086-2106: the read aligned against the current consensus */
typedef struct {
	uint32_t reflen;              /* g->cns->size (0: no consensus yet) */
	int32_t  have;                /* an alignment is supplied (needed when bsa_pog_needs_guide says so) */
	int32_t  qb, qe, tb, te;      /* seqalign_result_t of the read (query) against the consensus (target) */
	const uint32_t *cigar;        /* its CIGAR words, len << 4 | op */
	uint32_t ncigar;
	int32_t  sam;                 /* refmode: `cigar` is the read's SAM CIGAR against the backbone (ops M I D N S H = X: 0 1 2 3 4 5 7 8), `have` and qb .. te are
	                                 not read, reflen = the backbone's length, and cigar[ncigar] must be readable (bspoa.h:2073 reads one word behind the one it tests) */
} bsa_pog_guide_t;

/* what select / place decided for the read (g->qb, g->qe, g->slen, g->bandwidth, bspoa.h:2035-2110) */
typedef struct {
	uint32_t nhead, ntail;        /* header nodes between which the read is aligned */
	uint32_t nsel;                /* selected nodes (g->sels->size) */
	uint32_t bandwidth, qlen, slen, qb, qe;
} bsa_pog_read_t;

/* same signature as bsa_poa_graph_backend_fn of include/bsalign_poa_adapter.h: bsa_poa_batcher_submit_graph fits (user = the batcher) */
typedef int (*bsa_pog_backend_fn)(void *user, const bsa_poa_node_t *nodes, size_t nnodes, const bsa_poa_edge_t *edges, size_t nedges,
                                  const bsa_poa_cand_t *cands, size_t ncands, const uint8_t *query, uint32_t slen,
                                  const bsa_sweep_params_t *par, bsa_poa_result_t *res, bsa_poa_event_t *events, size_t events_cap);

/* ---- container (P0) ---- */
int  bsa_pog_create(const bsa_pog_params_t *par, bsa_pog_t **out);                         /* init_bspoa, bspoa.h:206 */
void bsa_pog_destroy(bsa_pog_t *g);                                                          /* free_bspoa, bspoa.h:354 */
void bsa_pog_clear(bsa_pog_t *g);                                                            /* beg_bspoa's clear_bspoa, bspoa.h:309 (the caller pushes the empty read 0 of bspoa.h:1782-1784 itself) */
/* _add_read_bspoa_core (bspoa.h:916-951): the read's nodes (sentinel, bases, sentinel); read 0 becomes the chained backbone with
 * HEAD / TAIL, every later read's sentinels join their rings.  bases: one per byte, 0..3.  Call for every read of the window, in
 * the order the alignment will take them (after the caller's shuffle), before the first bsa_pog_select.  *rid_out = its index. */
int  bsa_pog_add_read(bsa_pog_t *g, const uint8_t *bases, uint32_t len, uint32_t *rid_out);
int  bsa_pog_import(bsa_pog_t *g, const bsa_pog_snapshot_t *snap, const uint8_t *const *read_bases);       /* replace the graph (read_bases may be NULL: taken from the nodes) */
/* export: sizes first (arrays NULL), then the arrays (caller-allocated: nnodes nodes, nreads, nnodes + 1 offsets, nedges entries) */
int  bsa_pog_export(const bsa_pog_t *g, uint32_t *nnodes, uint32_t *nreads, uint32_t *nedges, uint32_t *head, uint32_t *tail,
                    bsa_pog_node_t *nodes, uint32_t *ndoff, uint32_t *rdlen, uint32_t *out_off, uint32_t *out_to, uint32_t *out_cov, uint32_t *in_off, uint32_t *in_from);

/* ---- one read (P7 = these five in order) ---- */
/* P2 sel_nodes_bspoa: the sub-graph read `rid` [rbeg, rbeg + rlen) is aligned against -- the rings met by the reads
 * [max(0, rid - nrec - 1), rid) between the rings of its two ends --, in-degrees, auxiliary edges that make every selected node
 * reachable.  *sel (owned by g, valid until the next select) lists the selected headers in the reference's order. */
int  bsa_pog_select(bsa_pog_t *g, uint32_t rid, uint32_t rbeg, uint32_t rlen, bsa_pog_read_t *rd, const uint32_t **sel);
/* the realn entry of align_rd_bspoa (bspoa.h:2626-2630, cut_rdnode_bspoa :741-795): the bases [rbeg, rbeg + rlen) of a read that is already in the graph
 * leave their rings (a ring's representative hands its edges to the next member) and are unchained; the bsa_pog_select that follows takes every read of
 * the window into the selection, whatever nrec says (bspoa.h:2636-2642).  Read 0, the backbone, is never cut. */
int  bsa_pog_cut(bsa_pog_t *g, uint32_t rid, uint32_t rbeg, uint32_t rlen);
/* does prepare_rd_align consult a guide alignment for this read (bspoa.h:2054, 2086)?  reflen = the consensus' length */
int  bsa_pog_needs_guide(const bsa_pog_t *g, uint32_t reflen);
/* P3 prepare_rd_align_bspoa without its profiles and row arena (the device builds its own): band width, [qb, qe), every selected
 * node's band offset from the guide's CIGAR (bspoa.h:2112-2198), the two auxiliary edges at the guide's ends.  cpos[k] = column
 * of sel[k] on the consensus (node->cpos after simple_cns_bspoa / cns_bspoa); NULL: the graph's own (what bsa_pog_apply left). */
int  bsa_pog_place(bsa_pog_t *g, const bsa_pog_guide_t *guide, const int32_t *cpos, bsa_pog_read_t *rd);
/* the selected sub-graph as the device's program (bsa_poa_node_t ... of bsalign_hip.h), arrays owned by g */
int  bsa_pog_program(bsa_pog_t *g, const bsa_poa_node_t **nodes, size_t *nnodes, const bsa_poa_edge_t **edges, size_t *nedges,
                     const bsa_poa_cand_t **cands, size_t *ncands, const uint8_t **query, bsa_sweep_params_t *par);
/* P4 / P5 / the walk of P6: runs the program through `fn` (NULL: bsa_poa_graph_host on `user` = a bsa_ctx_t*).  BSA_E_UNSUPPORTED:
 * the kernel declines the shape (bsa_poa_graph_supported) -- the graph is untouched but for the auxiliary edges, which
 * bsa_pog_abort removes. */
int  bsa_pog_run(bsa_pog_t *g, bsa_pog_backend_fn fn, void *user, bsa_poa_result_t *res, const bsa_poa_event_t **events);
/* the surgery of P6 (bspoa.h:2393-2405, 2501-2511) + the end of align_rd_bspoa (bspoa.h:2652-2657): the steps of the last
 * bsa_pog_run applied to the graph; *rs = the seqalign_result_t align_rd_bspoa returns.  events_gnode (optional, nevents
 * entries): every step's node as a GRAPH node index, for a caller that keeps a graph of its own in step. */
int  bsa_pog_apply(bsa_pog_t *g, bsa_result_t *rs, uint32_t *events_gnode);
/* the auxiliary edges select / place have added so far (from << 32 | to, header nodes, in the order they were added: each is one
 * chg_edge_bspoa(from, to, +1); bsa_pog_apply / bsa_pog_abort take them back in the same order).  A caller that keeps a graph of its
 * own in step applies the same changes: an edge whose coverage changes moves behind its equals in both of its lists. */
int  bsa_pog_aux_edges(const bsa_pog_t *g, const uint64_t **list, size_t *n);
int  bsa_pog_abort(bsa_pog_t *g);          /* give the read up after select / place: auxiliary edges removed, nothing else changed */
/* column positions of nodes (bsa_pog_place's default input): n entries for the nodes idx[k] */
int  bsa_pog_set_cpos(bsa_pog_t *g, const uint32_t *idx, const int32_t *cpos, size_t n);
int  bsa_pog_get_cpos(const bsa_pog_t *g, const uint32_t *idx, int32_t *cpos, size_t n);
/* seconds spent in select / place / program / run / apply since creation (host profile of the path) */
void bsa_pog_seconds(const bsa_pog_t *g, double out[5]);

#ifdef __cplusplus
}
#endif
#endif
