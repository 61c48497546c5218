/*
 * bsalign_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement ("oracle") of the reference's banded striped DP hot
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load or call this; the product path (libbsalign_hip.so) never does.
 *
 * Parity status: PINNED -- every function here is checked against the real
 * reference compiled into oracle/_ref/libbsref.so (tests/test_oracle_vs_ref.py,
 * runs when /root/reference is present) and against the committed golden
 * vectors in tests/golden/ (generated from that reference build by
 * tests/golden/make_golden.py).
 */
#ifndef BSALIGN_ORACLE_H
#define BSALIGN_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MODE_GLOBAL   0
#define ORC_MODE_OVERLAP  1
#define ORC_MODE_EXTEND   2
#define ORC_LANES         16          /* lanes of the reference's SIMD word (bsalign.h:142) */
#define ORC_EPI8_MIN      (-63)       /* bsalign.h:56 */
#define ORC_EPI8_MAX      (63)        /* bsalign.h:57 */
#define ORC_SCORE_MIN     (-(0x7FFFFFFF >> 2)) /* bsalign.h:58 */

/* layout-compatible with seqalign_result_t (bsalign.h:213-218) */
typedef struct {
	int32_t score, qb, qe, tb, te, mat, mis, ins, del, aln;
} orc_result_t;

/* query description used to compute S(x, base) on the fly (replaces the striped profile, bsalign.h:2166-2221) */
typedef struct {
	const uint8_t *seq;
	uint32_t len;
	const int8_t *mtx;   /* 16 entries, mtx[q*4+t] */
	int hpc;             /* 0: plain profile; 1: +bonus where q[x] != q[x+1] (bsalign.h:2203-2205) */
	int bonus;
} orc_query_t;

void     orc_set_score_matrix(int8_t mtx[16], int mat, int mis);                        /* bsalign.h:323 */
int      orc_get_piecewise(int gapo1, int gape1, int gapo2, int gape2, int bandwidth); /* bsalign.h:2084 */

/* Row-level functions.  Row buffers use the reference's striped index
 * idx(p) = (p % W) * 16 + p / W; ubegs has 17 entries. */
void     orc_row_init(int8_t *us, int8_t *es, int8_t *qs, int32_t *ubegs, int mode, uint32_t bandwidth,
                      int max_nt, int min_nt, int gapo1, int gape1, int gapo2, int gape2);            /* bsalign.h:2094 */
void     orc_row_movx(int8_t *us_dst, int8_t *es_dst, int8_t *qs_dst, int32_t *ub_dst,
                      const int8_t *us_src, const int8_t *es_src, const int8_t *qs_src, const int32_t *ub_src,
                      uint32_t W, uint32_t movx, int piecewise, int nt_max, int nt_min,
                      int gapo1, int gape1, int gapo2, int gape2);                                   /* bsalign.h:2244 */
int      orc_row_cal(uint32_t rbeg, uint8_t base,
                     const int8_t *us_src, const int8_t *es_src, const int8_t *qs_src, const int32_t *ub_src,
                     int8_t *us_dst, int8_t *es_dst, int8_t *qs_dst, int32_t *ub_dst,
                     const orc_query_t *qry, int gapo1, int gape1, int gapo2, int gape2,
                     uint32_t W, int rh, int piecewise);                                             /* bsalign.h:2727,2885,3084 */
void     orc_row_merge(const int8_t *us0, const int8_t *es0, const int8_t *qs0, const int32_t *ub0,
                       const int8_t *us1, const int8_t *es1, const int8_t *qs1, const int32_t *ub1,
                       int8_t *us2, int8_t *es2, int8_t *qs2, int32_t *ub2, uint32_t W, int piecewise); /* bsalign.h:2474 */
int      orc_getscore(const int8_t *us, const int32_t *ubegs, uint32_t W, uint64_t pos);            /* bsalign.h:3187 */
uint32_t orc_row_max(const int8_t *us, const int32_t *ubegs, uint32_t W, int32_t *max_score);       /* bsalign.h:3213 */
int      orc_band_mov(const int32_t *ubegs, uint32_t W, uint32_t tidx, uint32_t qoff, uint32_t qlen); /* bsalign.h:3331 */

/* Whole-pair functions.  cig may be NULL.  Return value: number of CIGAR words
 * (len<<4|op, bsalign.h:61-69, 401-417), or -(needed) when cap is too small,
 * or a large negative code ORC_ERR_* on invalid input. */
#define ORC_ERR_INPUT (-(1L << 40))
#define ORC_ERR_TRACE (-(1L << 41))  /* input on which the reference itself does not terminate / leaves its arena */
long     orc_align_pairwise(const uint8_t *q, uint32_t qlen, const uint8_t *t, uint32_t tlen,
                            int mode, uint32_t bandwidth, const int8_t mtx[16],
                            int gapo1, int gape1, int gapo2, int gape2,
                            orc_result_t *res, uint32_t *cig, long cap);                             /* bsalign.h:3854 */
long     orc_edit_pairwise(const uint8_t *q, uint32_t qlen, const uint8_t *t, uint32_t tlen,
                           int mode, uint32_t bandwidth,
                           orc_result_t *res, uint32_t *cig, long cap);                              /* bsalign.h:1046 */

/* global alignment through 4-bit traceback codes (restatement of the device's compact path; piecewise <= 1) */
long     orc_align_pairwise_codes(const uint8_t *q, uint32_t qlen, const uint8_t *t, uint32_t tlen,
                            uint32_t bandwidth, const int8_t mtx[16], int gapo1, int gape1, int gapo2, int gape2,
                            orc_result_t *res, uint32_t *cig, long cap);

long     orc_align_pairwise_codes_mode(const uint8_t *q, uint32_t qlen, const uint8_t *t, uint32_t tlen, int mode,
                            uint32_t bandwidth, const int8_t mtx[16], int gapo1, int gape1, int gapo2, int gape2,
                            orc_result_t *res, uint32_t *cig, long cap);

/* optional tracing hook for band-trajectory goldens: begs[i] = band offset of row i (tlen entries) */
long     orc_align_pairwise_trace(const uint8_t *q, uint32_t qlen, const uint8_t *t, uint32_t tlen,
                            int mode, uint32_t bandwidth, const int8_t mtx[16],
                            int gapo1, int gape1, int gapo2, int gape2,
                            orc_result_t *res, uint32_t *cig, long cap, int32_t *begs_out);

/* row-record dump in the device layout ((tlen+1) records of rowb bytes, row -1 first): lets the GPU tests
 * locate the first differing DP row */
long     orc_align_pairwise_rows(const uint8_t *q, uint32_t qlen, const uint8_t *t, uint32_t tlen,
                            int mode, uint32_t bandwidth, const int8_t mtx[16],
                            int gapo1, int gape1, int gapo2, int gape2,
                            orc_result_t *res, uint8_t *rows_out, uint32_t rowb);

/* POA sweep programs: flattened align_rd_bspoacore (bspoa.h:2515-2618); structs are layout-identical to
 * bsa_row_task_t / bsa_sweep_prog_t / bsa_sweep_result_t of include/bsalign_hip.h */
#define ORC_ROW_OP_UPDATE     0u
#define ORC_ROW_OP_MERGE      1u
#define ORC_ROW_OP_INIT       2u
#define ORC_ROW_OP_SCORE_TAIL 3u
#define ORC_ROW_OP_SCORE_END  4u
typedef struct { uint32_t op, src, dst, qoff_src, qoff_dst, toff, query; uint8_t base, prof; uint16_t reserved; } orc_row_task_t;
typedef struct { uint32_t first_task, ntasks, first_block, reserved; } orc_sweep_prog_t;
typedef struct { int32_t maxscr, maxidx, maxoff, reserved; } orc_sweep_result_t;
void     orc_sweep_run(uint8_t *rows, const orc_row_task_t *tasks, const orc_sweep_prog_t *progs, size_t nprogs,
                       const uint8_t *queries, const uint64_t *qoff, const uint32_t *qlen,
                       int mode, uint32_t bandwidth, int M, int X, int refbonus, int gapo1, int gape1, int gapo2, int gape2, int T,
                       orc_sweep_result_t *results);


/* ---- POA sweep, second formulation (bsalign_oracle_wf.c): absolute scores on a node / in-edge description of the selected
 * sub-graph, plus the traceback into the graph as a list of steps.  Structs are layout-identical to bsa_poa_node_t /
 * bsa_poa_edge_t / bsa_poa_cand_t / bsa_poa_cell_t / bsa_poa_event_t / bsa_poa_params_t of include/bsalign_hip.h. */
typedef struct { int32_t h; int8_t e, q; uint16_t tag; } orc_wf_cell_t;
#define ORC_WF_IN_PRESENT 0x80000000u
#define ORC_WF_IN_MERGE   0x40000000u
#define ORC_WF_IN_SAME    0x20000000u
#define ORC_WF_IN_TOFF    0x0FFFFFFFu
typedef struct { uint32_t src, movx, toff_kind; } orc_wf_input_t;
typedef struct { uint32_t rpos, gnode, first_in; uint16_t n_in; uint8_t base, flags; orc_wf_input_t in[2]; uint32_t reserved[2]; } orc_wf_node_t;
typedef struct { uint32_t src, cov, src_rpos, reserved; } orc_wf_edge_t;
typedef struct { uint32_t node, kind; } orc_wf_cand_t;                         /* kind 0: edge node -> tail, 1: node reaches the read end */
typedef struct { uint32_t node; int32_t x; uint32_t bt; } orc_wf_event_t;
typedef struct { int32_t mode; uint32_t bandwidth; int8_t M, X, refbonus, gapo1, gape1, gapo2, gape2, pad; int32_t T; } orc_wf_params_t;   /* == bsa_sweep_params_t */
typedef struct { int32_t maxscr, maxidx, maxoff, status, nevents, fin_node, fin_x, reserved; } orc_wf_result_t;     /* == bsa_poa_result_t */
int orc_wf_backend(void *user, const orc_wf_node_t *nodes, size_t nnodes, const orc_wf_edge_t *edges, size_t nedges,
		const orc_wf_cand_t *cands, size_t ncands, const uint8_t *query, uint32_t slen,
		const orc_wf_params_t *par, orc_wf_result_t *res, orc_wf_event_t *events, size_t events_cap);
void orc_wf_init_row(const orc_wf_params_t *par, orc_wf_cell_t *row, int32_t *u0);
void orc_wf_forward(const orc_wf_node_t *nodes, uint32_t nnodes, const uint8_t *query, uint32_t slen,
		const orc_wf_params_t *par, orc_wf_cell_t *rows, int32_t *u0);
void orc_wf_row_to_block(const orc_wf_cell_t *row, int32_t u0, uint32_t bandwidth, int pw, uint8_t *block);
void orc_wf_best(const orc_wf_node_t *nodes, const orc_wf_cand_t *cands, uint32_t ncands, uint32_t slen, const orc_wf_params_t *par,
		const orc_wf_cell_t *rows, orc_sweep_result_t *res);
long orc_wf_trace(const orc_wf_node_t *nodes, const orc_wf_edge_t *edges, const uint8_t *query, uint32_t slen,
		const orc_wf_params_t *par, const orc_wf_cell_t *rows, const int32_t *u0, uint32_t head, uint32_t midx, int xe,
		orc_wf_event_t *ev, long cap, int32_t fin[2]);

/* timing helper for bench.py's cpu_baseline (kind = "port") */
double   orc_align_batch_time(const uint8_t *seqs, const uint64_t *qoff, const uint32_t *qlen,
                              const uint64_t *toff, const uint32_t *tlen, long n, int mode, uint32_t bandwidth,
                              const int8_t mtx[16], int gapo1, int gape1, int gapo2, int gape2, int64_t *checksum);
double   orc_edit_batch_time(const uint8_t *seqs, const uint64_t *qoff, const uint32_t *qlen,
                             const uint64_t *toff, const uint32_t *tlen, long n, int mode, uint32_t bandwidth,
                             int64_t *checksum);

#ifdef __cplusplus
}
#endif
#endif

/* anti-diagonal u8 DP of remsa_pedits (bspoa.h:3752-3896, loop :3925-3935): fills rows 2 mbeg .. 2 mend - 1 of the two planes
 * (row = 16 W + 2 bytes, cell c at byte 1 + c) */
void orc_diagdp_fill(const uint8_t *seq0, const uint8_t *seq1, const uint8_t *const mats0[4], const uint8_t *const mats1[4],
		int mlen, int mbeg, int mend, int W, uint8_t *matrix0, uint8_t *matrix1);
int orc_diagdp_walk(const uint8_t *seq0, const uint8_t *seq1, const uint8_t *const mats0[4], const uint8_t *const mats1[4],
		int mlen, int mbeg, int mend, int W, const uint8_t *matrix0, const uint8_t *matrix1, uint8_t *steps, uint32_t *nsteps, int *score, int *xend, int *yend);
