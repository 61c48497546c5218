/*
 * bsalign_hip.h -- C-ABI of libbsalign_hip.so: the MI355X (gfx950) implementation of
 * bsalign's banded striped DP hot path.
 *
 * Plain C types only: no HIP, torch or C++ types cross this boundary.  Every entry
 * point returns 0 on success or a negative BSA_E_* code (the reference abort()s
 * instead: bsalign.h:3882-3885, 1076-1079).  Nothing here falls back to a CPU
 * implementation: without a usable GPU every compute entry point fails with
 * BSA_E_NODEVICE.
 *
 * What each entry point replaces in the reference (/root/reference):
 *   bsa_align_*   <- banded_striped_epi8_seqalign_pairwise   bsalign.h:3854 (one call per pair, main.c:323-326)
 *   bsa_edit_*    <- striped_seqedit_pairwise                bsalign.h:1046 (main.c:196-204)
 *   bsa_rows_*    <- dpalign_row_update_bspoa / dpalign_row_merge_bspoa  bspoa.h:2232-2272
 *                    (= banded_striped_epi8_seqalign_piecex_row_movx + _row_cal + _row_merge,
 *                     bsalign.h:2244, 3181, 2474)
 *   bsa_result_t  <- seqalign_result_t                       bsalign.h:213-218
 *   CIGAR words   <- u4v of (len << 4 | op)                  bsalign.h:61-69, 401-417
 * The reference has no batch interface (one pair per call, one thread); the batch
 * forms below are the device-sized equivalent of its per-pair loop.  The drop-in
 * single-pair functions with the reference's exact signatures live in
 * include/bsalign_compat.h and are thin wrappers over these.
 */
#ifndef BSALIGN_HIP_H
#define BSALIGN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* modes (bsalign.h:30-38) */
#define BSA_MODE_GLOBAL   0
#define BSA_MODE_OVERLAP  1
#define BSA_MODE_EXTEND   2
#define BSA_MODE_ROWRECORDS 0x100 /* flag: keep the reference-layout row records and the literal traceback even where the
                                      compact 4-bit-code path applies (global mode, 1-piece gaps, small scores) */

/* CIGAR op codes (bsalign.h:61-69) */
#define BSA_CIGAR_M 0
#define BSA_CIGAR_I 1
#define BSA_CIGAR_D 2

/* error codes */
#define BSA_OK            0
#define BSA_E_NODEVICE   (-1)   /* no HIP device / runtime failure at context creation */
#define BSA_E_ARG        (-2)   /* invalid argument (NULL pointer, bad mode, W == 0 ...) */
#define BSA_E_NOMEM      (-3)   /* device allocation failed / workspace limit too small for one pair */
#define BSA_E_HIP        (-4)   /* a HIP call failed (see bsa_last_error) */
#define BSA_E_CIGAR_CAP  (-5)   /* cigar arena too small; cigar_off[n] holds the required number of words */
#define BSA_E_UNSUPPORTED (-6)  /* parameter combination not implemented on the device yet */

/* per-pair status bits written to the optional status array */
#define BSA_ST_OK          0u
#define BSA_ST_BAD_BASE    1u   /* a base code > 3 (the reference would index past matrix[16]) */
#define BSA_ST_EMPTY       2u   /* qlen == 0 or tlen == 0 */
#define BSA_ST_TRACE       4u   /* traceback left the stored band: the reference does not terminate on this input.
                                    With device pointers (bsa_align_run) the compact path also sets it for a pair its flags
                                    cannot decide (none seen in testing): resubmit such pairs with BSA_MODE_ROWRECORDS.
                                    bsa_align_batch does that itself. */
#define BSA_ST_DEVICE      8u   /* the device gave the pair up: a wave of the segmented forward pass waited for the previous segment's
                                    state longer than its bound (about two seconds) -- a fault, never an input property.  Result zeroed;
                                    bsa_align_batch returns BSA_E_HIP when any pair carries it.  bsa_align_run (device pointers) is
                                    asynchronous and returns BSA_OK: its caller finds the flag in the status array it passed. */

/* == seqalign_result_t (bsalign.h:213-218): 10 x int32, [qb,qe) x [tb,te) half-open */
typedef struct {
	int32_t score;
	int32_t qb, qe;
	int32_t tb, te;
	int32_t mat, mis, ins, del, aln;
} bsa_result_t;

/* arguments of banded_striped_epi8_seqalign_pairwise that are shared by a batch (bsalign.h:399) */
typedef struct {
	int32_t  mode;        /* BSA_MODE_* */
	uint32_t bandwidth;   /* 0 => qlen of each pair; rounded up to a multiple of 16 (bsalign.h:3861-3862) */
	int8_t   matrix[16];  /* matrix[q*4+t] (bsalign.h:323) */
	int8_t   gapo1, gape1, gapo2, gape2; /* negative penalties as the reference's CLI stores them (main.c:283-288) */
} bsa_align_params_t;

typedef struct {
	int32_t  mode;        /* BSA_MODE_* */
	uint32_t bandwidth;   /* rounded to a multiple of 64 with the rules of bsalign.h:1055-1067 */
} bsa_edit_params_t;

typedef struct bsa_ctx bsa_ctx_t;

/* ---- context ------------------------------------------------------------------------------ */
int         bsa_ctx_create(int device, bsa_ctx_t **out);
void        bsa_ctx_destroy(bsa_ctx_t *ctx);
/* run on a caller-owned hipStream_t (passed as void*); NULL = the context's own stream */
int         bsa_ctx_set_stream(bsa_ctx_t *ctx, void *hip_stream);
/* cap the device scratch (traceback rows) the context may allocate; 0 = 80% of free memory */
int         bsa_ctx_set_workspace_limit(bsa_ctx_t *ctx, size_t bytes);
int         bsa_ctx_sync(bsa_ctx_t *ctx);
const char *bsa_last_error(bsa_ctx_t *ctx);
/* average duration (ms) of the dominant kernel's launches in the last *_run call, measured with HIP
 * events on the launch stream; *launches = number of launches averaged, *cells = band cells they covered */
int         bsa_ctx_last_kernel_ms(bsa_ctx_t *ctx, double *ms, long *launches, double *cells);
/* the same for the traceback launches of the last *_run call, and the names of the kernels behind the two timings
 * (traceback = 0: forward DP, 1: traceback) */
int         bsa_ctx_last_trace_ms(bsa_ctx_t *ctx, double *ms, long *launches);
const char *bsa_ctx_last_kernel_name(bsa_ctx_t *ctx, int traceback);
/* pairs of the last bsa_align_batch call that the compact (code) path left undecided and the literal kernels re-ran
 * (the hand-over described at bsa_align_batch; with scores outside the static guard: the pairs the checked whole-query
 * kernel flagged) */
long        bsa_ctx_last_handover(bsa_ctx_t *ctx);
void        bsa_set_score_matrix(int8_t matrix[16], int8_t mat, int8_t mis);   /* bsalign.h:323 */

/* ---- 8-bit banded striped pairwise alignment (A-rows) ------------------------------------------
 * Sequences: one base per byte, codes 0..3 (the reference's u1i* qseq/tseq), all pairs in one blob;
 * pair k uses seqs[qoff[k] .. qoff[k]+qlen[k]) and seqs[toff[k] .. toff[k]+tlen[k]).
 * Outputs: out[k]; CIGAR words of pair k at cigar[cigar_off[k] .. cigar_off[k+1]) (cigar_off has n+1
 * entries); status[k] = 0 or BSA_ST_* flags.  status may be NULL, but it is the only place a pair that stays UNDECIDED
 * can be reported (BSA_ST_TRACE: the reference's own traceback does not terminate on it, result zeroed): with status == NULL
 * bsa_align_batch returns BSA_E_UNSUPPORTED if any such pair is left, instead of BSA_OK with silent zeroed records.
 *
 * bsa_align_batch      : every pointer is HOST memory (copies in, runs, copies out, synchronises).
 * bsa_align_plan_*     : two-phase form for resident data -- the plan takes the HOST metadata
 *                        (offsets, lengths) once; bsa_align_run takes DEVICE pointers for the
 *                        sequence blob and all outputs and is asynchronous on the context stream.
 *
 * Whole-query bands (bandwidth 0, the reference CLI's default, or a bandwidth no shorter than any query): a plan
 * runs at ONE kernel width; when no query of the plan is longer than 256 bases it runs at 64 / 128 / 256 columns on
 * the fast path (same results as the reference's own width), otherwise on the slower run-time-width kernel.
 * bsa_align_batch groups the pairs of a mixed batch by width class itself; with bsa_align_plan_* group short and long
 * queries into separate plans. */
int bsa_align_batch(bsa_ctx_t *ctx, const uint8_t *seqs, size_t seqs_bytes,
                    const uint64_t *qoff, const uint32_t *qlen,
                    const uint64_t *toff, const uint32_t *tlen, size_t n,
                    const bsa_align_params_t *par,
                    bsa_result_t *out, uint32_t *cigar, size_t cigar_cap_words,
                    uint64_t *cigar_off, uint32_t *status);

typedef struct bsa_align_plan bsa_align_plan_t;
int  bsa_align_plan_create(bsa_ctx_t *ctx, const uint64_t *qoff, const uint32_t *qlen,
                           const uint64_t *toff, const uint32_t *tlen, size_t n,
                           const bsa_align_params_t *par, bsa_align_plan_t **out);
void bsa_align_plan_destroy(bsa_align_plan_t *plan);
/* total band cells of the plan: sum over pairs of tlen * bw_eff (the GCUPS numerator) */
double bsa_align_plan_cells(const bsa_align_plan_t *plan);
int  bsa_align_run(bsa_align_plan_t *plan, const uint8_t *d_seqs,
                   bsa_result_t *d_out, uint32_t *d_cigar, size_t cigar_cap_words,
                   uint64_t *d_cigar_off, uint32_t *d_status);

/* ---- 2-bit striped edit-distance pairwise alignment (E-rows) --------------------------------- */
int bsa_edit_batch(bsa_ctx_t *ctx, const uint8_t *seqs, size_t seqs_bytes,
                   const uint64_t *qoff, const uint32_t *qlen,
                   const uint64_t *toff, const uint32_t *tlen, size_t n,
                   const bsa_edit_params_t *par,
                   bsa_result_t *out, uint32_t *cigar, size_t cigar_cap_words,
                   uint64_t *cigar_off, uint32_t *status);

typedef struct bsa_edit_plan bsa_edit_plan_t;
int  bsa_edit_plan_create(bsa_ctx_t *ctx, const uint64_t *qoff, const uint32_t *qlen,
                          const uint64_t *toff, const uint32_t *tlen, size_t n,
                          const bsa_edit_params_t *par, bsa_edit_plan_t **out);
void bsa_edit_plan_destroy(bsa_edit_plan_t *plan);
double bsa_edit_plan_cells(const bsa_edit_plan_t *plan);
int  bsa_edit_run(bsa_edit_plan_t *plan, const uint8_t *d_seqs,
                  bsa_result_t *d_out, uint32_t *d_cigar, size_t cigar_cap_words,
                  uint64_t *d_cigar_off, uint32_t *d_status);

/* ---- k-mer anchored edit alignment (reference: kmer_striped_seqedit_pairwise, bsalign.h:1209-1536; CLI `edit -m kmer`) ---
 * Unique same-strand k-mers (ksz <= 15) shared by the two sequences are chained on the host; only the stretches
 * between consecutive anchors are aligned, every one of them by the device edit path (two bsa_edit_batch calls for
 * the whole batch: reversed heads and tails in EXTEND mode, gaps in GLOBAL mode), and the CIGAR of each pair is stitched together exactly as the
 * reference does it (including where it puts the anchor matches).  A pair without a usable chain is aligned globally.
 * All pointers are HOST memory; cigar/cigar_off/status follow bsa_edit_batch. */
typedef struct {
	uint32_t ksz;        /* k-mer size, values above 15 mean 15 (bsalign.h:1217); the CLI default is 13 (main.c:141) */
	uint32_t threads;    /* host threads for chaining and stitching; 0 = all hardware threads (or $BSA_KMER_THREADS) */
} bsa_kmer_params_t;

int bsa_kmer_edit_batch(bsa_ctx_t *ctx, const uint8_t *seqs, size_t seqs_bytes,
                        const uint64_t *qoff, const uint32_t *qlen,
                        const uint64_t *toff, const uint32_t *tlen, size_t n,
                        const bsa_kmer_params_t *par,
                        bsa_result_t *out, uint32_t *cigar, size_t cigar_cap_words,
                        uint64_t *cigar_off, uint32_t *status);

/* The host-only pieces of the above, usable without a GPU.
 * bsa_kmer_chain   : anchors of one pair in query order, maps[i] = query offset << 32 | target offset of the k-mer
 *                    (the reference's `maps`, bsalign.h:1431-1433); returns their number, 0 when the pair has to be
 *                    aligned as a whole, 0xFFFFFFFF when cap is too small (min(qlen, tlen) always suffices).
 * bsa_kmer_segments: the alignments the reference would run for these anchors (bsalign.h:1451-1530), at most kmap + 1.
 * bsa_kmer_assemble: result and CIGAR of the pair from the results of its segments. */
#define BSA_KMER_SEG_REVERSED 0x100u   /* head segment: both sequences are aligned reversed, SEQALIGN_MODE_KMER (bsalign.h:1489-1499) */
typedef struct {
	uint32_t qb, qe, tb, te;   /* the segment aligns query[qb, qe) with target[tb, te); a reversed head has qb = tb = 0 */
	uint32_t mode;             /* BSA_MODE_GLOBAL | BSA_MODE_EXTEND, optionally | BSA_KMER_SEG_REVERSED */
	uint32_t ml;               /* anchor matches emitted in front of this segment's CIGAR */
} bsa_kmer_seg_t;
uint32_t bsa_kmer_chain(uint32_t ksz, const uint8_t *q, uint32_t qlen, const uint8_t *t, uint32_t tlen, uint64_t *maps, uint32_t cap);
uint32_t bsa_kmer_segments(uint32_t ksz, const uint64_t *maps, uint32_t kmap, uint32_t qlen, uint32_t tlen, bsa_kmer_seg_t *segs);
int      bsa_kmer_assemble(const bsa_kmer_seg_t *segs, uint32_t nseg, const bsa_result_t *seg_out, const uint32_t *seg_cigar,
                           const uint64_t *seg_cigar_off, bsa_result_t *out, uint32_t *cigar, uint64_t cigar_cap_words, uint64_t *cigar_words);

/* ---- row-level kernels for the POA seq->graph DP (P4; reference bspoa.h:2232-2272) ----------------------------
 * The POA sweep calls, per graph edge u -> v, row_movx + row_cal on u's DP row (dpalign_row_update_bspoa) and, per
 * extra in-edge, row_merge (dpalign_row_merge_bspoa).  bsa_rows_run executes a batch of such INDEPENDENT tasks (one
 * topological level of many reads / windows) on device-resident row blocks.  A row block has exactly the reference's
 * layout and size (bspoa.h:1787-1793, 2217): us[bw] | es[bw] if piecewise >= 1 | qs[bw] if piecewise == 2 |
 * int32 ubegs[17], striped index (p % W) * 16 + p / W, padded to 16 bytes (bsa_rows_block_bytes). */
#define BSA_ROW_OP_UPDATE 0u   /* rows[dst] = row_cal(row_movx(rows[src], qoff_dst - qoff_src))   bspoa.h:2232 */
#define BSA_ROW_OP_MERGE  1u   /* rows[dst] = cell-wise max(rows[src], rows[dst])                  bspoa.h:2263 */
#define BSA_ROW_OP_INIT   2u   /* rows[dst] = row -1 of the read (row_init, bspoa.h:2226)                      */
typedef struct {
	uint32_t op;                  /* BSA_ROW_OP_* */
	uint32_t src, dst;            /* row block indices (u->mmidx, v->mmidx) */
	uint32_t qoff_src, qoff_dst;  /* band offsets (u->rpos, v->rpos) */
	uint32_t toff;                /* v->mpos: row number used by the left-boundary score */
	uint32_t query;               /* index into the query table */
	uint8_t  base;                /* v->base */
	uint8_t  prof;                /* (v->base == u->base) * 2 + v->bonus: which of the 4 profiles of bspoa.h:2199-2213 */
	uint16_t reserved;
} bsa_row_task_t;
typedef struct {
	int32_t  mode;                /* par->alnmode */
	uint32_t bandwidth;           /* multiple of 16, bandwidth / 16 in {1,2,4,8,16} */
	int8_t   M, X, refbonus;      /* par->M, par->X, par->refbonus */
	int8_t   gapo1, gape1, gapo2, gape2;
} bsa_rows_params_t;
size_t bsa_rows_block_bytes(uint32_t bandwidth, int8_t gapo1, int8_t gape1, int8_t gapo2, int8_t gape2);
/* all pointers are DEVICE memory; asynchronous on the context stream; tasks of one call must not depend on each other */
int bsa_rows_run(bsa_ctx_t *ctx, uint8_t *d_rows, const bsa_row_task_t *d_tasks, size_t ntasks,
                 const uint8_t *d_queries, const uint64_t *d_qoff, const uint32_t *d_qlen, const bsa_rows_params_t *par);

/* ---- the whole per-read sweep on the device (align_rd_bspoacore, bspoa.h:2515-2618) ----------
 * The reference walks the selected sub-graph with a stack and per-node in-degree counters; the visiting order depends
 * on the graph only, so the host flattens it into a program of row tasks (include/bsalign_poa_adapter.h does that from
 * the reference's own graph structures).  One program = one read against one graph; its tasks run in order on one
 * 16-lane DPP row, and many programs (POA windows) run concurrently.  Besides INIT / UPDATE / MERGE a program holds
 * the two places where the reference samples an end-of-alignment score:
 *   SCORE_TAIL  edge u -> tail (bspoa.h:2547-2577): H at the last band cell of u's row + the unaligned-tail gap + T,
 *               and in overlap mode the row maximum (row_max, bsalign.h:3213);
 *   SCORE_END   node v complete and its band reaches the read end, non-global modes (bspoa.h:2597-2606).
 * For both: src = the node's row block, qoff_src = its band offset (rpos), toff = the node index reported as maxidx.
 * A candidate replaces the running best only if strictly greater, in program order -- the reference's rule. */
#define BSA_ROW_OP_SCORE_TAIL 3u
#define BSA_ROW_OP_SCORE_END  4u
typedef struct {
	uint32_t first_task, ntasks;  /* this program's slice of the task array */
	uint32_t first_block;         /* task block indices are relative to this row block (mmidx 0 of the read's memp) */
	uint32_t reserved;
} bsa_sweep_prog_t;
typedef struct {
	int32_t maxscr, maxidx, maxoff; /* g->maxscr, g->maxidx, g->maxoff (bspoa.h:2227-2229); -2^30-ish, -1, -1 when no candidate */
	int32_t reserved;
} bsa_sweep_result_t;
typedef struct {
	bsa_rows_params_t rows;
	int32_t T;                    /* par->T: bonus for reaching the read end */
} bsa_sweep_params_t;
/* all pointers are DEVICE memory; asynchronous on the context stream */
/* Preconditions of the device-pointer entries (bsa_rows_run, bsa_sweep_run; bsa_sweep_host checks them and returns
 * BSA_E_ARG): every program has ntasks >= 1; first_task + ntasks stays inside the task array; first_block + src and
 * first_block + dst address row blocks inside d_rows; query indexes the query table.  A program with ntasks == 0 is skipped. */
int bsa_sweep_run(bsa_ctx_t *ctx, uint8_t *d_rows, const bsa_row_task_t *d_tasks, const bsa_sweep_prog_t *d_progs,
                  size_t nprogs, const uint8_t *d_queries, const uint64_t *d_qoff, const uint32_t *d_qlen,
                  const bsa_sweep_params_t *par, bsa_sweep_result_t *d_results);
/* ---- the per-read seq->graph DP as an anti-diagonal wavefront with the traceback on the device (P5 + P6) --------------------
 * Second form of the sweep (align_rd_bspoacore bspoa.h:2515-2618) that also replaces alignment2graph_bspoa's walk
 * (bspoa.h:2274-2513): the row blocks never leave the device, what comes back per read is the best end cell and the list
 * of traceback steps.  The band offset of every node is fixed before the sweep (prepare_rd_align_bspoa, bspoa.h:2168-2174),
 * so -- unlike the pairwise DP, whose band moves with the scores -- the dependencies between rows are known up front: one
 * wave runs one read, lane l owns node i (nodes in the order the reference completes them), walks its row cell by cell and
 * trails the rows it depends on by movx + 1 cells; rows of the nodes in flight live in an LDS ring, finished rows are
 * drained to HBM in 4-byte cells (H relative to the row's first cell, e, q) for the traceback.  Scores are absolute
 * integers: inside bsa_poa_graph_supported()'s guard none of the reference's int8 operations saturates, and the rows
 * equal the reference's block for block (tests/test_poa_graph_gpu.py through bsa_poa_graph_host's rows_out).
 *
 * A program = the selected sub-graph of one read:
 *   nodes   in completion order (node 0 = the head; every input of a node has a lower index).  A node carries two views:
 *           FORWARD: at most two inputs in[0..1]; an input is the row of node `src` moved by movx = rpos - rpos(src) and
 *           extended by one DP row (dpalign_row_update_bspoa, bspoa.h:2232), or -- kind MERGE -- the finished row of a
 *           PARTIAL node at the same rpos taken as it is (dpalign_row_merge_bspoa, bspoa.h:2263: the cell-wise maximum).
 *           A graph node with more than two selected in-edges is preceded by partial nodes (gnode = 0xFFFFFFFF) that
 *           fold its first in-edges, two at a time.
 *           TRACEBACK: its selected in-edges in the order of the reference's erev list with their coverage (edges[]).
 *   cands   the places where the reference samples an end-of-alignment score, in its visiting order (bspoa.h:2549-2603)
 * include/bsalign_poa_adapter.h builds programs from the reference's own graph. */
#define BSA_POA_IN_PRESENT 0x80000000u
#define BSA_POA_IN_MERGE   0x40000000u
#define BSA_POA_IN_SAME    0x20000000u   /* v->base == u->base: the profile without the homopolymer bonus (bspoa.h:2588) */
#define BSA_POA_IN_TOFF    0x0FFFFFFFu   /* v->mpos when the reference took the edge (left-boundary score, bspoa.h:2246-2248) */
typedef struct { uint32_t src, movx, toff_kind; } bsa_poa_input_t;
typedef struct {
	uint32_t rpos;                /* u->rpos */
	uint32_t gnode;               /* index of the node in the caller's graph; 0xFFFFFFFF for a partial node */
	uint32_t first_in;            /* traceback view: first in-edge record, relative to the program's edges */
	uint16_t n_in;
	uint8_t  base;                /* u->base (0..3; 4 = HEAD / TAIL sentinel) */
	uint8_t  flags;               /* bit 0: u->bonus */
	bsa_poa_input_t in[2];        /* forward view */
	uint32_t reserved[2];
} bsa_poa_node_t;                 /* 48 bytes */
typedef struct { uint32_t src, cov, src_rpos, reserved; } bsa_poa_edge_t;     /* local index of the predecessor, e->cov */
typedef struct { uint32_t node, kind; } bsa_poa_cand_t;                       /* kind 0: edge node -> tail, 1: node complete and its band reaches the read end */
typedef struct { uint32_t node; int32_t x; uint32_t bt; } bsa_poa_event_t;    /* one step of alignment2graph_bspoa: bt 0 M, 1 I, 2 D, 4 D2 (bsalign.h:40-50) */
typedef struct {
	uint32_t first_node, nnodes, first_edge, nedges, first_cand, ncands;
	uint32_t slen;                /* g->slen */
	uint32_t event_cap;           /* room for this program's events */
	uint64_t query_off;           /* its read (one base per byte, g->qseq + g->qb) inside the query blob */
	uint64_t first_event;
} bsa_poa_prog_t;                 /* 48 bytes */
#define BSA_POA_ST_OK     0
#define BSA_POA_ST_TRACE  1       /* the walk left the stored band / found no predecessor: the reference reads outside its rows or does not terminate there */
#define BSA_POA_ST_EVENTS 2       /* event_cap too small */
#define BSA_POA_ST_NOCAND 3       /* no end-of-alignment candidate (the reference would start its walk at node -1) */
typedef struct {
	int32_t maxscr, maxidx, maxoff;   /* g->maxscr, LOCAL index of g->maxidx, g->maxoff */
	int32_t status;                   /* BSA_POA_ST_* */
	int32_t nevents;
	int32_t fin_node, fin_x;          /* where the walk stopped: rs.tb = cpos of that node, rs.qb = fin_x (bspoa.h:2307-2311) */
	int32_t reserved;
} bsa_poa_result_t;
typedef struct { int32_t h; int8_t e, q; uint16_t tag; } bsa_poa_cell_t;      /* a row cell as rows_out returns it: absolute H, e = E - H, q = Q - H */
/* lanes a read of `max_slen` bases can use at this parameter set (a power of two up to 64), 0 = not supported: bandwidth above 256,
 * scores outside the exactness guard, or a read too long for the LDS.  Callers fall back to bsa_sweep_* then. */
int bsa_poa_graph_supported(const bsa_sweep_params_t *par, uint32_t max_slen);
/* 1 when bsa_poa_graph_run / _host take this parameter set at ANY bandwidth (up to 32768 columns): scores inside the same exactness guard, the
 * generic-width kernel of bsa_poa_gen.hip behind the same entry points (a workgroup per read, rows of 8 bytes a cell in the context's scratch,
 * as many programs side by side as $BSA_POA_GEN_WS_GB -- default 48 -- holds; rows_out is not available there).  A window's first aligned read
 * has the whole read as its band (bspoa.h:2045-2054, 2109-2111) and goes this way. */
int bsa_poa_graph_gen_supported(const bsa_sweep_params_t *par);
/* all pointers DEVICE memory, asynchronous on the context stream.  The steps of a walk leave the device as one word each,
 * node << 3 | bt (x is implied: it starts at maxoff and moves left with every M and I step): d_steps is scratch, program k walks
 * into d_steps[first_event .. + event_cap); when it is done its steps are appended to d_packed (capacity: the sum of all
 * event_cap), result.reserved = where, *d_packed_used = words in use (zeroed by the call).  bsa_poa_expand_steps turns a
 * program's words into bsa_poa_event_t on the host.  d_rows: (total nodes) x bw uint32 cells, d_u0: total nodes int32 (scratch
 * the traceback reads; pass NULL for both to use the context's own buffer). */
int bsa_poa_graph_run(bsa_ctx_t *ctx, const bsa_poa_node_t *d_nodes, size_t nnodes, const bsa_poa_edge_t *d_edges, const bsa_poa_cand_t *d_cands,
                      const bsa_poa_prog_t *d_progs, size_t nprogs, const uint8_t *d_queries, uint32_t max_slen, const bsa_sweep_params_t *par,
                      bsa_poa_result_t *d_results, uint32_t *d_steps, uint32_t *d_packed, uint64_t *d_packed_used, uint32_t *d_rows, int32_t *d_u0);
void bsa_poa_expand_steps(const uint32_t *steps, bsa_poa_result_t *res, bsa_poa_event_t *events);
/* HOST buffers in and out (uploads, runs, downloads, synchronises).  rows_out / u0_out (optional, tests): every node's row as
 * absolute cells, nnodes x bw, and its ubegs[0]. */
int bsa_poa_graph_host(bsa_ctx_t *ctx, const bsa_poa_node_t *nodes, size_t nnodes, const bsa_poa_edge_t *edges, size_t nedges,
                       const bsa_poa_cand_t *cands, size_t ncands, const bsa_poa_prog_t *progs, size_t nprogs,
                       const uint8_t *queries, size_t query_bytes, const bsa_sweep_params_t *par,
                       bsa_poa_result_t *results, bsa_poa_event_t *events, size_t events_cap, bsa_poa_cell_t *rows_out, int32_t *u0_out);

/* ---- batch scatter, host side (bsalign_amd/csrc/bsa_shard.cpp) ----------------
 * The pairs of one rank's contiguous range packed into the blob that travels to it: target k, then query k, each
 * padded to 16 bytes.  bsa_shard_pack fills out[0 .. bsa_shard_bytes) and every pair's offsets inside it (host memory,
 * `threads` host threads, 0 = as many as the machine has, at most 16). */
size_t bsa_shard_bytes(const uint32_t *qlen, const uint32_t *tlen, size_t first, size_t count);
int    bsa_shard_pack(const uint8_t *seqs, const uint64_t *qoff, const uint32_t *qlen, const uint64_t *toff, const uint32_t *tlen,
                      size_t first, size_t count, uint8_t *out, size_t out_bytes, uint64_t *out_qoff, uint64_t *out_toff, unsigned threads);

/* ---- the exchange itself, for a C host: RCCL point-to-point messages over xGMI (bsalign_amd/csrc/bsa_shard_rccl.hip) -----------
 * One process per GPU.  Rank 0 calls bsa_shard_unique_id and ships the 128 bytes to the other ranks by whatever means the job has
 * (launcher environment, a file, MPI); every rank then creates its communicator on its own context.  RCCL (librccl.so.1) is loaded
 * at run time; with nranks == 1 nothing is loaded and both calls degenerate to copies.
 *   scatter   the root holds the batch in HOST memory; every rank gets its contiguous range [*first, *first + *count) -- ranges
 *             balanced by tlen x bandwidth --: the shard blob in DEVICE memory (*d_seqs, owned by the communicator, valid until the
 *             next scatter) with the pairs' lengths and offsets inside it (host arrays of `cap` entries), ready for
 *             bsa_align_plan_create / bsa_align_run.  Lengths travel as two broadcasts, the N - 1 shards as one ncclGroup of sends.
 *   gather    every rank hands in its range's results (DEVICE), CIGAR words (DEVICE) and CIGAR offsets (HOST, count + 1); the root
 *             receives all n records in input order, the CIGAR arena and its offsets (HOST).  Sizes travel as one ncclAllGather. */
typedef struct bsa_shard_comm bsa_shard_comm_t;
int  bsa_shard_unique_id(uint8_t id[128]);
int  bsa_shard_comm_create(bsa_ctx_t *ctx, int rank, int nranks, const uint8_t id[128], bsa_shard_comm_t **out);
void bsa_shard_comm_destroy(bsa_shard_comm_t *comm);
int  bsa_shard_scatter(bsa_shard_comm_t *comm, int root, const uint8_t *seqs, const uint64_t *qoff, const uint32_t *qlen, const uint64_t *toff, const uint32_t *tlen,
                       size_t n, uint32_t bandwidth, size_t *first, size_t *count, uint8_t **d_seqs, size_t *blob_bytes,
                       uint32_t *local_qlen, uint32_t *local_tlen, uint64_t *local_qoff, uint64_t *local_toff, size_t cap);
int  bsa_shard_gather(bsa_shard_comm_t *comm, int root, const bsa_result_t *d_out, const uint32_t *d_cigar, const uint64_t *cigar_off, size_t count,
                      bsa_result_t *out, uint32_t *cigar, size_t cigar_cap_words, uint64_t *out_cigar_off, size_t n);

/* ---- many windows in lock-step (bsalign_amd/csrc/bsa_batcher.hip) ----------------------------
 * One POA (end_bspoa, bspoa.h:4722-4776) is sequential in its reads, but windows are independent: a caller with many
 * of them runs each on a host thread of its own and lets every thread's sweep go through a batcher.  submit() has the
 * signature of the single-window backend of include/bsalign_poa_adapter.h (pass the batcher as `user`), is thread-safe
 * and BLOCKS until every participant that has not left is waiting in it; the last arrival executes all programs as one
 * device launch per distinct parameter set, then everybody returns with its results and row blocks.  A window that
 * has aligned its last read calls leave().  participants = number of windows (threads) that will submit. */
typedef struct bsa_sweep_batcher bsa_sweep_batcher_t;
int  bsa_sweep_batcher_create(bsa_ctx_t *ctx, uint32_t participants, bsa_sweep_batcher_t **out);
void bsa_sweep_batcher_destroy(bsa_sweep_batcher_t *b);
int  bsa_sweep_batcher_submit(void *batcher, const bsa_row_task_t *tasks, size_t ntasks, const uint8_t *query, uint32_t slen,
                              const bsa_sweep_params_t *par, uint8_t *rows_out, size_t nblocks, bsa_sweep_result_t *res);
void bsa_sweep_batcher_leave(bsa_sweep_batcher_t *b);
/* optional, first thing a window's thread does: from then on the thread computes only while it holds one of as many host slots
 * as the process has CPUs (cgroup quota, or $BSA_POA_HOST_THREADS); waiting in submit() hands the slot to another window */
void bsa_sweep_batcher_enter(bsa_sweep_batcher_t *b);
/* the same rendezvous for the graph form (sweep + traceback on the device, nothing but the steps comes back): signature of
 * bsa_poa_graph_backend_fn (include/bsalign_poa_adapter.h), `batcher` as user.  Declines with BSA_E_UNSUPPORTED, without
 * waiting, what bsa_poa_graph_supported declines; the window then submits the read through bsa_sweep_batcher_submit. */
int  bsa_poa_batcher_submit_graph(void *batcher, const bsa_poa_node_t *nodes, size_t nnodes, const bsa_poa_edge_t *edges, size_t nedges,
                                  const bsa_poa_cand_t *cands, size_t ncands, const uint8_t *query, uint32_t slen, const bsa_sweep_params_t *par,
                                  bsa_poa_result_t *res, bsa_poa_event_t *events, size_t events_cap);
/* out[0..7] = batches, device launches, programs, tasks, bytes uploaded, bytes downloaded, device microseconds, microseconds inside batches */
void bsa_sweep_batcher_stats(bsa_sweep_batcher_t *b, uint64_t out[8]);

/* HOST buffers in, HOST buffers out (uploads, runs, downloads, synchronises): what a single-window caller such as
 * the adapter uses.  rows_out (nblocks * bsa_rows_block_bytes, may be NULL) receives every row block so that host
 * traceback code (alignment2graph_bspoa, bspoa.h:2274) can read them as if the CPU had computed them. */
int bsa_sweep_host(bsa_ctx_t *ctx, const bsa_row_task_t *tasks, size_t ntasks, const bsa_sweep_prog_t *progs, size_t nprogs,
                   const uint8_t *queries, const uint64_t *qoff, const uint32_t *qlen, size_t nqueries,
                   const bsa_sweep_params_t *par, uint8_t *rows_out, size_t nblocks, bsa_sweep_result_t *results);

/* ---- synthetic read pairs (measurement inputs, SURVEY 8(d) / BASELINE.md 3) ---------------------
 * pair k: target = iid uniform ACGT of length L from splitmix64(seed ^ k*0x9E3779B97F4A7C15);
 * query = target with errors at rate err_q32 / 2^32 split sub:ins:del = 23:31:46.
 * Layout: target k at seqs[k*stride .. +L), query k at seqs[n*stride + k*stride .. +qlen[k]),
 * stride = bsa_synth_stride(L).  *_host fills host memory (no GPU needed); *_dev fills device
 * memory on the context stream and writes qlen to a device array. */
size_t bsa_synth_stride(uint32_t L);
/* ---- the anti-diagonal u8 DP of the MSA refinement (reference: maxmat_dp_diag_rowcal bspoa.h:3856-3896, driven by the
 * fill loop of remsa_pedit_rd_bspoacore bspoa.h:3925-3935; replaces that loop, the traceback that follows it stays the
 * caller's).  One problem = one read of a POA window against the window's column profile.  `planes` holds the ten byte
 * planes in the reference's own layout (bspoa.h:4213-4233): the offsets point at LOGICAL index 0 of a plane, which carries
 * 8 W bytes of padding in front of it and behind index mlen - 1 (seq planes: base codes, >= 4 = no base; mats planes: u8
 * counts).  `matrix` receives rows 2 mbeg .. 2 mend - 1 of the two difference planes (row r of a plane at out + r (16 W + 2),
 * cell c at byte 1 + c, guard cells at bytes 0 and 16 W + 1); other rows are not written.  All problems of a call share W
 * (1, 2 or 4: band of 16, 32 or 64 cells). */
typedef struct {
	uint64_t seq0, seq1;            /* read plane (x side), consensus plane (y side, reversed as the reference stores it) */
	uint64_t mats0[4], mats1[4];    /* mats[0][b], mats[1][b] */
	uint64_t out0, out1;            /* row 0 of matrix[0], matrix[1] inside `matrix` */
	uint32_t mlen, mbeg, mend, W;
} bsa_diagdp_prob_t;
int bsa_diagdp_batch(bsa_ctx_t *ctx, const uint8_t *planes, size_t planes_bytes, const bsa_diagdp_prob_t *probs, size_t n,
                     uint8_t *matrix, size_t matrix_bytes);
/* The same fill FOLLOWED BY THE TRACEBACK of remsa_pedit_rd_bspoacore (bspoa.h:3965-4040) on the device: the difference planes stay there
 * (2.1 bytes a cell: 12 GB for 64 windows of 64 reads x 22 k columns) and what comes back per problem is the walk from (mend - 1, mend - 1):
 * two bits a step -- 0 diagonal (x - 1, y - 1), 1 x - 1, 2 y - 1 -- sixteen steps a word from bit 0 up, problem k's words at
 * steps[first_word .. ], plus the score the reference returns (the sum over the diagonal steps) and where the walk stopped.  A caller
 * replays the steps to do what the reference does at every diagonal step (merge_nodes_bspoa, bspoa.h:4011-4022).  probs[k].out0 / out1 are
 * ignored (the planes are laid out by the library).  steps_cap_words >= sum over k of bsa_diagdp_walk_words(mbeg, mend). */
typedef struct { uint32_t nsteps; int32_t score; int32_t xi, yi; uint32_t status, reserved; uint64_t first_word; } bsa_diagdp_walk_t;    /* status 0 ok, 1 left the band, 2 no source explains a cell (the reference aborts) */
static inline uint64_t bsa_diagdp_walk_words(uint32_t mbeg, uint32_t mend){ return (2ull * (mend - mbeg) + 2 + 15) / 16; }
int bsa_diagdp_walk_batch(bsa_ctx_t *ctx, const uint8_t *planes, size_t planes_bytes, const bsa_diagdp_prob_t *probs, size_t n,
                          bsa_diagdp_walk_t *walks, uint32_t *steps, size_t steps_cap_words);
/* device time of the last bsa_diagdp_batch (staging + fill kernels), ms */
double bsa_diagdp_last_ms(bsa_ctx_t *ctx);

int bsa_synth_pairs_host(uint64_t seed, uint64_t first_pair, size_t n, uint32_t L, uint32_t err_q32,
                         uint8_t *seqs, uint32_t *qlen);
int bsa_synth_pairs_dev(bsa_ctx_t *ctx, uint64_t seed, uint64_t first_pair, size_t n, uint32_t L, uint32_t err_q32,
                        uint8_t *d_seqs, uint32_t *d_qlen);

#ifdef __cplusplus
}
#endif
#endif
