// bsa_common.h -- internal definitions shared by the HIP translation units of libbsalign_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/bsalign_hip.h"

#define BSA_LANES      16                      // running blocks per band row == lanes of one DPP row
#define BSA_EPI8_MIN   (-63)                   // bsalign.h:56
#define BSA_EPI8_MAX   (63)                    // bsalign.h:57
#define BSA_SCORE_MIN  (-(0x7FFFFFFF >> 2))    // bsalign.h:58
#define BSA_QPAD_CODE  4                       // staged query code for columns >= qlen (S = -63, bsalign.h:2157-2160)

// device-side description of one batch chunk of the 8-bit path
struct Align8Args {
	// staged sequences: query codes padded with BSA_QPAD_CODE, target bytes
	const uint8_t  *qst;        // staged queries
	const uint8_t  *tst;        // staged targets
	const uint64_t *qpoff;      // [n] offset of pair's staged query
	const uint64_t *tpoff;      // [n] offset of pair's staged target
	const uint32_t *qlen;       // [n]
	const uint32_t *tlen;       // [n]
	const uint32_t *order;      // [n] processing order (sorted by length); results go to original index
	const uint64_t *slot_off;   // [n] (indexed by processing position) byte offset of the pair's row slot in `rows`
	uint8_t        *rows;       // traceback rows workspace
	uint32_t       *status;     // [n] per original pair
	uint32_t first, count;      // processing positions [first, first+count) handled by this launch
	uint32_t bw;                // effective bandwidth (multiple of 16), uniform over the chunk
	uint32_t rowb;              // bytes per stored row record
	int32_t  mode;
	int32_t  gapo1, gape1, gapo2, gape2;
	int32_t  smax, smin;        // max / min of the score matrix (bsalign.h:3868-3873)
	uint32_t mrow[4];           // mrow[t] = 4 packed int8 scores {matrix[0*4+t], matrix[1*4+t], matrix[2*4+t], matrix[3*4+t]}
	int8_t   matrix[16];
};

// Traceback state of one pair ("slot") in the workspace:
//   [begs: (tlen + 2) int32, padded to 16 B]   begs[r + 1] = band offset of target row r (row -1 first, = 0)
//   [row records: tlen + 3 of rowb bytes]       record r + 1 = target row r; the 2 spare records + consumed rows
//                                               at the end are the CIGAR scratch of the traceback
// Row record, BLOCK-INTERLEAVED so that one traceback step touches one cache line: for running block y = 0..15
// (= lane y of the forward kernel):
//   [u: W int8][e: W int8 if pw >= 1][q: W int8 if pw == 2][pad to 4][ubegs[y]: int32]        = blk bytes
// followed by ubegs[16] (int32).  u[p] = H(p) - H(p-1) for band position p = y*W + k.
static inline __host__ __device__ uint32_t bsa_blk_cells(uint32_t W, int pw){ return ((uint32_t)(pw + 1) * W + 3u) & ~3u; }
static inline __host__ __device__ uint32_t bsa_blk_bytes(uint32_t W, int pw){ return bsa_blk_cells(W, pw) + 4u; }
static inline __host__ __device__ uint32_t bsa_row_bytes(uint32_t bw, int pw){
	uint32_t b = 16u * bsa_blk_bytes(bw / 16u, pw) + 4u;
	return (b + 15u) & ~15u;
}
static inline __host__ __device__ size_t bsa_begs_bytes(uint32_t tlen){ return (((size_t)tlen + 2) * 4 + 15) & ~(size_t)15; }
static inline __host__ __device__ size_t bsa_slot_bytes(uint32_t tlen, uint32_t rowb){ return bsa_begs_bytes(tlen) + ((size_t)tlen + 3) * rowb; }

static inline __host__ __device__ int bsa_get_piecewise(int gapo1, int gape1, int gapo2, int gape2, int bandwidth){ // bsalign.h:2084-2092
	if(gapo2 < gapo1 && gape2 > gape1 && gapo2 + gape2 < gapo1 + gape1 && (gapo1 - gapo2) / (gape1 - gape2) < bandwidth) return 2;
	return gapo1 ? 1 : 0;
}

// device-side description of one batch chunk of the 2-bit edit path
struct EditArgs {
	const uint8_t  *qst, *tst;      // staged query / target bytes (codes 0..3)
	const uint64_t *qpoff, *tpoff;  // [n]
	const uint64_t *qbits;          // staged query bit planes: plane0 then plane1, qwords[k] words each
	const uint64_t *qboff;          // [n] word offset of pair's plane0
	const uint32_t *qwords;         // [n]
	const uint32_t *qlen, *tlen;    // [n]
	const uint32_t *order;          // [n]
	const uint64_t *slot_off;       // [n] by processing position
	uint8_t        *rows;
	uint32_t       *status;         // [n] per original pair
	int32_t        *fwd_sbeg;       // [n] by processing position: H at the band start of the last row
	uint32_t first, count;
	uint32_t bw;                    // effective bandwidth of this launch (multiple of 64)
	uint32_t pad_rows;              // spare row records at the end of every slot (CIGAR scratch)
	int32_t  mode;
};

// launchers implemented in the kernel translation units
hipError_t bsa_launch_align8_fwd(const Align8Args &a, int pw, hipStream_t st);
hipError_t bsa_launch_align8_backcal(const Align8Args &a, int pw, bsa_result_t *out, uint32_t *cig_cnt, hipStream_t st);
bool bsa_align8_supported_bw(uint32_t bw);
bool bsa_align8_pk_supported(const Align8Args &a, int pw);
hipError_t bsa_launch_align8_fwd_pk(const Align8Args &a, int pw, hipStream_t st);
bool bsa_edit_supported_bw(uint32_t bw);
hipError_t bsa_launch_edit_stage(const uint8_t *seqs, const uint64_t *qoff, const uint32_t *qlen, const uint64_t *toff, const uint32_t *tlen,
		const uint64_t *qpoff, const uint64_t *tpoff, const uint64_t *qboff, const uint32_t *qwords,
		uint8_t *qst, uint8_t *tst, uint64_t *qbits, uint32_t *status, uint32_t n, hipStream_t st);
hipError_t bsa_launch_edit_fwd(const EditArgs &a, hipStream_t st);
hipError_t bsa_launch_edit_trace(const EditArgs &a, bsa_result_t *out, uint32_t *cig_cnt, hipStream_t st);
