/*
 * bsalign_compat.h -- source-level drop-in for the hot-path part of the reference's bsalign.h.
 *
 * The reference is header-only ("copy bsalign.h, list.h, sort.h and mem_share.h", bsalign.h:13); callers use
 * static-inline functions.  This header declares the SAME names, argument order and result types for the hot
 * path, implemented in C (bsalign_amd/host/bsalign_compat.c -> libbsalign_compat.so) on top of the C-ABI batch
 * library (include/bsalign_hip.h): every call runs a batch of one pair on the GPU.  A caller with many pairs
 * should use the batch entry points directly.
 *
 * Kept from the reference (file:line in /root/reference):
 *   typedefs u1i/u4i/u8i/b1i            mem_share.h:40-64
 *   b1v / u4v layout                    list.h:116-122, 519-530   ({buffer, size, cap, mem_zero:1 n_head:6 aligned:6 off:51})
 *   seqalign_result_t                   bsalign.h:213-218
 *   SEQALIGN_MODE_* / SEQALIGN_CIGAR_*  bsalign.h:30-38, 61-69
 *   banded_striped_epi8_seqalign_set_score_matrix   bsalign.h:323
 *   banded_striped_epi8_seqalign_pairwise           bsalign.h:399 / 3854
 *   striped_seqedit_pairwise                        bsalign.h:232 / 1046
 *   kmer_striped_seqedit_pairwise                   bsalign.h:1209           (SEQALIGN_MODE_KMER, bsalign.h:33)
 *   seqalign_cigar2alnstr                           bsalign.h:394 / 531
 * Error behaviour follows the reference: a mempool whose `aligned` field is < 16 aborts with a message
 * (bsalign.h:3882-3885); "no alignment" is reported as rs.mat == 0 / zeroed result (main.c:206, 327).  Device or
 * input errors that the reference cannot have (no GPU, base code > 3) also print a message and abort().
 * SEQALIGN_MODE_CIGRESV appends to `cigars` instead of clearing it (bsalign.h:3713-3719); SEQALIGN_MODE_QPROF and
 * SEQALIGN_MODE_MEMRESV concern host scratch the device path does not use and are ignored.
 */
#ifndef BSALIGN_COMPAT_H
#define BSALIGN_COMPAT_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint8_t  u1i;
typedef uint32_t u4i;
typedef unsigned long long u8i;
typedef int8_t   b1i;

#define SEQALIGN_MODE_GLOBAL   0
#define SEQALIGN_MODE_OVERLAP  1
#define SEQALIGN_MODE_EXTEND   2
#define SEQALIGN_MODE_KMER     3   /* only as the CLI's mode switch and inside kmer_striped_seqedit_pairwise */
#define SEQALIGN_MODEMASK_TYPE 0x3
#define SEQALIGN_MODE_QPROF    4
#define SEQALIGN_MODE_MEMRESV  8
#define SEQALIGN_MODE_CIGRESV  16
#define seqalign_mode_type(mode) ((mode) & SEQALIGN_MODEMASK_TYPE)

#define SEQALIGN_CIGAR_M 0
#define SEQALIGN_CIGAR_I 1
#define SEQALIGN_CIGAR_D 2
#define SEQALIGN_CIGAR_N 3
#define SEQALIGN_CIGAR_S 4
#define SEQALIGN_CIGAR_H 5
#define SEQALIGN_CIGAR_P 6
#define SEQALIGN_CIGAR_E 7
#define SEQALIGN_CIGAR_X 8

typedef struct {
	int score;
	int qb, qe;
	int tb, te;
	int mat, mis, ins, del, aln;
} seqalign_result_t;

typedef struct { b1i *buffer; u8i size; u8i cap; u8i mem_zero:1, n_head:6, aligned:6, off:51; } b1v;
typedef struct { u4i *buffer; u8i size; u8i cap; u8i mem_zero:1, n_head:6, aligned:6, off:51; } u4v;

/* minimal list helpers with the reference's names (list.h:154-200) */
b1v  *adv_init_b1v(u8i init_size, int mem_zero, int aligned_base, u4i n_head);
void  free_b1v(b1v *list);
void  clear_b1v(b1v *list);
u4v  *init_u4v(u8i init_size);
void  free_u4v(u4v *list);
void  clear_u4v(u4v *list);
void  push_u4v(u4v *list, u4i e);

void banded_striped_epi8_seqalign_set_score_matrix(b1i matrix[16], b1i mat, b1i mis);

seqalign_result_t banded_striped_epi8_seqalign_pairwise(u1i *qseq, u4i qlen, u1i *tseq, u4i tlen, b1v *mempool, u4v *cigars,
		int mode, u4i bandwidth, b1i matrix[16], b1i gapo1, b1i gape1, b1i gapo2, b1i gape2, int verbose);

seqalign_result_t striped_seqedit_pairwise(u1i *qseq, u4i qlen, u1i *tseq, u4i tlen, int mode, u4i bandwidth,
		b1v *mempool, u4v *cigars, int verbose);

/* k-mer anchored edit alignment: anchors are chained on the host, everything between them is aligned on the GPU in
 * one go (bsa_kmer_edit_batch, include/bsalign_hip.h).  qseq / tseq are not modified (the reference reverses their
 * heads in place and back, bsalign.h:1490-1495). */
seqalign_result_t kmer_striped_seqedit_pairwise(u1i ksz, u1i *qseq, u4i qlen, u1i *tseq, u4i tlen, b1v *mempool, u4v *cigars, int verbose);

u4i seqalign_cigar2alnstr(u1i *qseq, u1i *tseq, seqalign_result_t *rs, u4v *cigars, char *alnstr[3], u4i length);

/* device selection for the process-wide context the wrappers create lazily (default 0) */
void bsalign_compat_set_device(int device);
void bsalign_compat_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif
