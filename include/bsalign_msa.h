/* bsalign_msa.h -- the POA's MSA formats on plain arrays (the data formats behind the path).
 *
 * The reference keeps a finished window's MSA as `msacols` (mrow = nseq + 3 bytes per column: one byte per read --
 * base 0..3, 4 = gap, 5 / 6 = outside the read --, then consensus, consensus quality, alternative-allele quality) and
 * `msaidxs` (the order of the columns), bspoa.h:131-132.  These functions take exactly those two arrays and produce /
 * parse the reference's two formats byte for byte; nothing here touches a BSPOA.
 *
 *   binary container   dump_binary_msa_bspoa        bspoa.h:1555-1586
 *                      load_binary_msa_bspoa_core   bspoa.h:1588-1650, post_load_binary_msa_bspoa :1652-1685
 *   text               print_msa_bspoa              bspoa.h:1491-1553 with its row builders :1329-1483 (colorful = 0)
 *
 * All functions return 0 or a negative BSA_E_* code (bsalign_hip.h); writers report the bytes they need in *need and
 * return BSA_E_CIGAR_CAP-style BSA_E_ARG only for bad arguments: a too small buffer is BSA_E_NOMEM with *need set.
 */
#ifndef BSALIGN_MSA_H
#define BSALIGN_MSA_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Binary container: [0x81, u32 metalen, meta]  0x22, u32 mlen, u32 nseq, mlen x (nseq + 1) bytes (reads + consensus of
 * every column, in msaidxs order), mlen consensus qualities, mlen alternative qualities, 0xFF.  `cols` / `idxs` as the
 * reference holds them (idxs == NULL: columns in storage order); mrow = nseq + 3. */
int bsa_msa_binary_write(const uint8_t *cols, const uint32_t *idxs, uint32_t nseq, uint32_t mlen,
                         const char *meta, uint32_t metalen, uint8_t *out, size_t cap, size_t *need);

/* Parses one container (records up to and including the 0xFF terminator).  Outputs are optional; `cols` receives
 * mlen x (nseq + 3) bytes in file order (the reference's loader sets msaidxs to the identity).  Returns 0, BSA_E_ARG on
 * a truncated / malformed stream, BSA_E_NOMEM when cols_cap or meta_cap is too small (sizes still reported). */
int bsa_msa_binary_read(const uint8_t *in, size_t len, size_t *consumed, uint32_t *nseq, uint32_t *mlen,
                        uint8_t *cols, size_t cols_cap, char *meta, size_t meta_cap, uint32_t *metalen);

/* What post_load_binary_msa_bspoa derives from a loaded MSA: the consensus with its two quality strings (columns
 * whose consensus is a base), and optionally every read's bases (rdseqs: concatenated, rdoffs[nseq + 1]). */
int bsa_msa_consensus(const uint8_t *cols, const uint32_t *idxs, uint32_t nseq, uint32_t mlen,
                      uint8_t *cns, uint8_t *qlt, uint8_t *alt, uint32_t *clen,
                      uint8_t *rdseqs, uint64_t *rdoffs);

/* print_msa_bspoa(g, label, mbeg, mend, linewidth, colorful = 0, out) into a buffer.  var_mpos: the MSA columns the
 * reference marks with '~' in the ruler (g->var, ascending; NULL / 0 = none).  cns / qlt / alt: the consensus arrays
 * (g->cns, g->qlt, g->alt) the trailer lines print. */
int bsa_msa_text(const uint8_t *cols, const uint32_t *idxs, uint32_t nseq, uint32_t mlen,
                 const uint8_t *cns, const uint8_t *qlt, const uint8_t *alt,
                 const uint32_t *var_mpos, uint32_t nvar,
                 const char *label, uint32_t mbeg, uint32_t mend, uint32_t linewidth,
                 char *out, size_t cap, size_t *need);

/* Consensus calling of a window's MSA: cns_bspoa (bspoa.h:3457-3733) with its alignment-event table
 * (gen_cns_aln_event_table_bspoa, bspoa.h:142-204), sum_log_nums (:3413-3453) and the binomial / normal tail of the
 * alternative-allele quality (:3391-3411), on the same plain arrays.  A column DP over five states (consensus base A, C, G, T or
 * gap) whose transition score sums, over the reads, the log probability of the event each read shows (match, substitution,
 * insertion / deletion opened or extended, homopolymer variants), followed by the traceback, the consensus quality (phred of the
 * posterior of the chosen state) and the quality against the most frequent other allele.  Double precision in the reference's
 * own order of operations: with the same libm the bytes are the reference's.
 *   cols / idxs   as above, mrow = nall + 3; the three consensus bytes of every column are (over)written, as the reference does
 *   nseq          reads that vote in the DP (the reference: min(g->nmsa, g->nrds));   nmax: reads counted for the alternative
 *                 allele (g->nrds);   nall: read rows of a column (g->seqs->nseq)
 *   cns/qlt/alt   receive the columns whose consensus is a base (room for mlen each), *clen their number; *score the DP's log
 *                 probability (cns_bspoa's return value) */
typedef struct { float psub, pins, pdel, piex, pdex, hins, hdel; } bsa_cns_params_t;     /* BSPOAPar, bspoa.h:70; defaults 0.10 0.10 0.15 0.15 0.20 0.20 0.40 */
int bsa_msa_call_consensus(uint8_t *cols, const uint32_t *idxs, uint32_t nall, uint32_t nseq, uint32_t nmax, uint32_t mlen,
                           const bsa_cns_params_t *par, uint8_t *cns, uint8_t *qlt, uint8_t *alt, uint32_t *clen, double *score);

/* The same for MANY windows on the device (bsa_cns_dev.hip; needs a context of libbsalign_hip, include/bsalign_hip.h): window k's columns
 * start at cols + win[k].cols_off (mrow = nall + 3 as above), its column order at idxs + win[k].idxs_off (words; ~0: storage order), its
 * consensus / qualities go to cns / qlt / alt + win[k].out_off (room for mlen each), clen[k] / score[k] as above.  One wave runs a window's
 * columns (the automaton is sequential in them), the windows run side by side.  Every sum over the reads is formed in the host form's order
 * from the host's own logarithm tables, so it has the host's bits; the merges log(exp a + exp b) use the device's exp / log, which may
 * differ from the host libm's in the last place: consensus, both quality strings and the three bytes of every column are the host form's
 * (and the reference's) byte for byte on everything tested (tests/test_cns_gpu.py), score[k] to a relative 1e-12.
 * Returns BSA_E_UNSUPPORTED above ~9900 reads in a window (the per-read state lives in LDS). */
typedef struct { uint64_t cols_off, idxs_off, out_off; uint32_t nall, nseq, nmax, mlen; } bsa_cns_window_t;
struct bsa_ctx;
int bsa_msa_call_consensus_batch(struct bsa_ctx *ctx, uint8_t *cols, size_t cols_bytes, const uint32_t *idxs, size_t idxs_words,
                                 const bsa_cns_window_t *win, size_t nwin, const bsa_cns_params_t *par,
                                 uint8_t *cns, uint8_t *qlt, uint8_t *alt, size_t out_bytes, uint32_t *clen, double *score);

#ifdef __cplusplus
}
#endif
#endif
