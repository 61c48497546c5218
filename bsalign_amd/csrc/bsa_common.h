// bsa_common.h -- internal definitions shared by the HIP translation units of libbsalign_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/bsalign_hip.h"

// Environment knobs (BSA_*): read ONCE, when the first context is created, into a snapshot; bsa_env() looks a knob up there (no
// getenv on the launch path).  bsa_env_reload() takes a new snapshot (tests that flip a knob inside one process call it).
const char *bsa_env(const char *name);
extern "C" void bsa_env_reload(void);

#define BSA_LANES      16                      // running blocks per band row == lanes of one DPP row
#define BSA_EPI8_MIN   (-63)                   // bsalign.h:56
#define BSA_EPI8_MAX   (63)                    // bsalign.h:57
#define BSA_SCORE_MIN  (-(0x7FFFFFFF >> 2))    // bsalign.h:58
#define BSA_QPAD_CODE  4                       // staged query code for columns >= qlen (S = -63, bsalign.h:2157-2160)

// device-side description of one batch chunk of the 8-bit path
// bytes of padding behind the band of a staged query (bsa_api.hip: qpad = widest band + BSA_QPAD_TAIL): the forward kernels' LDS query window
// (bsa_align8_x.hip, x_qwin) reads up to 4 KD - W = 48 bytes behind the band's last block; one constant for the plan and the kernels' static_assert
#define BSA_QPAD_TAIL 96
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
	uint32_t bw;                // effective bandwidth (multiple of 16), uniform over the chunk; 0 = per pair roundup(qlen, 16)
	uint32_t rowb;              // bytes of one row group = 16 tiles (see the layout note below)
	uint32_t static_band;       // every pair's band covers its whole query (qlen <= bw): the band never moves (k_align8_fwd_x_static)
	uint32_t ref_bw;            // compact path, a whole-query band widened to `bw` (bsa_api.hip): the reference's own bandwidth (1 = per pair roundup(qlen, 16)); 0 = bw
	uint32_t code_fmt;          // compact slots, one-piece gaps at bandwidth 128: 0 = four flag planes a block, 1 = M, R and two-bit D / Od fields (below)
	uint32_t max_tlen;          // longest target of the launch (pairs are ordered by target length: that of position `first`)
	uint32_t sys_chk;           // systolic whole-query kernel: scores outside the static guard -- the kernel checks every pair and flags (BSA_ST_TRACE) the ones on which the reference's int8 arithmetic may clamp
	uint32_t *xq; uint64_t xq_bytes;    // control words + band states of the persistent forward kernel (k_align8_fwd_xq), or null
	int32_t  mode;
	int32_t  gapo1, gape1, gapo2, gape2;
	int32_t  smax, smin;        // max / min of the score matrix (bsalign.h:3868-3873)
	uint32_t mrow[4];           // mrow[t] = 4 packed int8 scores {matrix[0*4+t], matrix[1*4+t], matrix[2*4+t], matrix[3*4+t]}
	int8_t   matrix[16];
};

// Traceback state of one pair ("slot") in the workspace:
//   [begs: (tlen + 2) int32, padded to 16 B]   begs[r + 1] = band offset of target row r (row -1 first, = 0)
//   [row-group tiles]                           see below; the tiles of consumed rows + 1 spare group at the end are
//                                               the CIGAR scratch of the traceback
// Block record (what one lane of the forward kernel owns of one row): for running block y = 0..15
//   [u: W int8][e: W int8 if pw >= 1][q: W int8 if pw == 2][pad to 4][ubegs[y]: int32]        = blk bytes
// with u[p] = H(p) - H(p-1) for band position p = y*W + k (ubegs[16] is not stored: the traceback never reads it).
// TILING: the traceback walks up the rows while its band-relative position drifts slowly, so the records of the SAME
// block y of G consecutive rows share one tile (G = 64 / blk rows, tile = 64 bytes when blk <= 64): one 128-byte
// line (two neighbouring blocks) serves G steps of the walk instead of one.  The forward kernel keeps the last G-1
// records of its block in registers and writes whole tiles, so HBM only ever sees full 64-byte stores.
// Record (row r, block y), with rr = r + 1:
//   tiles + ((rr / G) * 16 + y) * tileb + (rr % G) * blk
static inline __host__ __device__ uint32_t bsa_blk_cells(uint32_t W, int pw){ return ((uint32_t)(pw + 1) * W + 3u) & ~3u; }
static inline __host__ __device__ uint32_t bsa_blk_bytes(uint32_t W, int pw){ return bsa_blk_cells(W, pw) + 4u; }
static inline __host__ __device__ uint32_t bsa_tile_rows(uint32_t W, int pw){ uint32_t g = 64u / bsa_blk_bytes(W, pw); return g ? g : 1u; }
static inline __host__ __device__ uint32_t bsa_tile_bytes(uint32_t W, int pw){ return (bsa_tile_rows(W, pw) * bsa_blk_bytes(W, pw) + 15u) & ~15u; }
// (a multiple of 256 bytes, and slots start at multiples of 256 -- bsa_api.hip rounds what a pair needs: a code row, a group of four and a line of sixteen band
// offsets then never straddle a 64-byte line; with 16-byte rounding three pairs in four had every row across two lines and the L2 wrote some of them back half filled)
static inline __host__ __device__ size_t bsa_begs_bytes(uint32_t tlen){ return (((size_t)tlen + 2) * 4 + 255) & ~(size_t)255; }
static inline __host__ __device__ size_t bsa_groups(uint32_t tlen, uint32_t G){ return ((size_t)tlen + 1 + G - 1) / G + 1; }   // + 1 spare group
static inline __host__ __device__ size_t bsa_slot_bytes(uint32_t tlen, uint32_t W, int pw){
	return bsa_begs_bytes(tlen) + bsa_groups(tlen, bsa_tile_rows(W, pw)) * 16 * (size_t)bsa_tile_bytes(W, pw);
}

// COMPACT (4-bit code) slot of the global-mode fast path (bsa_align8_pk.hip, CODES = true; HISTORY.md section 3):
//   int32 begs[tlen + 2]  (begs[tlen + 1] = final score, global mode)  |  code rows 0 .. tlen-1  |  BSA_CODE_SPARE_ROWS
//   spare rows: CIGAR scratch of the traceback and, in overlap / extend mode, the end record the forward pass leaves
//   at the start of the spare area (bsa_code_end_t followed by the last row's u bytes in natural band order)
// Code row = 16 blocks x CW dwords (CW = max(1, W / 8)), running block y of the row = dwords y*CW ...  The 4W bits of a
// block are four planes of W bits, plane n at bit n*W: M, D, R (insert opens here for the next cell), Od (stored e is
// a fresh opening); inside a plane cell k of the block is bit W-1-k.
// Format 1 (Align8Args::code_fmt, W = 8, one-piece gaps with 1 <= -gapo <= 3; k_align8_fwd_x* with DO2, k_align8_trace_codes_wave<8>): the block's
// dword is  M | R << 8 | F << 16,  F = eight two-bit fields, cell k at bits 15 - 2k, 14 - 2k: the cell's new e-difference min(h - (u + e), -gapo),
// so 0 means D and -gapo means Od (they exclude each other); where the forward pass clears D (cells at / beyond the previous row's band end) a
// zero field becomes a value that is neither.  One cell is literal: band position 0 of a row whose band offset is 0 (query column 0, where D
// follows the comparison of bsalign.h:3763-3767 and may hold together with Od): bit 14 = D, bit 15 = Od.
// Two-piece gaps (W = 8 only): two dwords per block, eight byte planes --
//   dword 0: A | D << 8 | D2 << 16 | B << 24        dword 1: R1 | R2 << 8 | Od1 << 16 | Od2 << 24
// D / D2: h == u + e / h == u + q.  A and B fold M and "which insertion chain equals h" (I1: h == f, I2: h == g), which are
// only consulted where D = D2 = 0:  D or D2 set: A is M.  Else (A, B) = (1, 0) M; (1, 1) not M, both chains; (0, 1) chain 1
// only; (0, 0) chain 2 only.  R1 / R2, Od1 / Od2: the flags R, Od of the two pieces (HISTORY.md section 3 states the rules).
// TILING: the traceback walks up the rows while its position inside the band drifts slowly (the band follows the
// diagonal), so it wants a few blocks of many rows, not whole rows.  Rows are stored in groups of four; inside a
// group the four rows of ONE block are adjacent (16 CW bytes), blocks follow each other:
//   dword offset of (row r, block y, dword d) = ((r / 4) * 64 + y * 4 + (r % 4)) * CW + d          (bsa_code_off)
// Three neighbouring blocks of four rows are then 48 contiguous bytes (one or two 64-byte lines instead of four), and
// the row count of a slot is rounded up to a multiple of four (bsa_code_rows).
// (two-piece gaps: 8 bits per cell -- the planes A, D, D2, B, R1, R2, Od1, Od2 of W bits each, plane j at bit j W of the block's W / 4 dwords:
// one dword at W = 4, the two dwords above at W = 8, four at W = 16)
static inline __host__ __device__ uint32_t bsa_code_words(uint32_t W, int pw = 1){ return pw == 2 ? (W >= 4u ? W / 4u : 1u) : (W >= 8u ? W / 8u : 1u); }
static inline __host__ __device__ uint32_t bsa_code_row_bytes(uint32_t W, int pw = 1){ return 64u * bsa_code_words(W, pw); }
#define BSA_CODE_SPARE_ROWS 7u
static inline __host__ __device__ size_t bsa_code_off(uint32_t r, uint32_t y, uint32_t CW){ return ((size_t)(r >> 2) * 64u + y * 4u + (r & 3u)) * CW; }
static inline __host__ __device__ uint32_t bsa_code_rows(uint32_t tlen){ return (tlen + 3u) & ~3u; }       // stored rows: whole groups of four
// end record of the non-global modes: per lane the best end-of-query score seen while the band touched the query end
// and its row (bsalign.h:4023-4032), and the last row itself for row_max (bsalign.h:4038-4046)
struct bsa_code_end_t { int32_t cand_sc[16], cand_te[16], ubegs[17], rbeg_last; };
static inline __host__ __device__ size_t bsa_code_slot_bytes(uint32_t tlen, uint32_t W, int pw = 1){
	return bsa_begs_bytes(tlen) + ((size_t)bsa_code_rows(tlen) + BSA_CODE_SPARE_ROWS) * bsa_code_row_bytes(W, pw);
}

static inline __host__ __device__ int bsa_get_piecewise(int gapo1, int gape1, int gapo2, int gape2, int bandwidth){ // bsalign.h:2084-2092
	if(gapo2 < gapo1 && gape2 > gape1 && gapo2 + gape2 < gapo1 + gape1 && (gapo1 - gapo2) / (gape1 - gape2) < bandwidth) return 2;
	return gapo1 ? 1 : 0;
}

// device-side description of one batch chunk of the 2-bit edit path
// effective bandwidth of one pair (bsalign.h:1055-1067)
#ifdef __HIPCC__
__host__ __device__
#endif
static inline uint32_t bsa_edit_bw_eff(uint32_t qlen, uint32_t tlen, int type, uint32_t bandwidth){
	const uint32_t qround = (qlen + 63u) / 64u * 64u;
	if(type == BSA_MODE_OVERLAP || type == BSA_MODE_EXTEND) return qround;
	uint32_t bw = (bandwidth + 63u) / 64u * 64u;
	if(bw == 0 || bw > qlen) bw = qround;
	if(bw < qlen){
		const uint32_t step = (qlen + tlen - 1) / tlen + 1;
		if(bw < step) bw = (step + 63u) / 64u * 64u;
	}
	return bw;
}
#define BSA_EDIT_REG_BW 1024u       // widest band of the register kernels (16 words)
// launch classes of the edit plan: the band itself up to BSA_EDIT_REG_BW; above, by the words per lane the wave-per-pair
// kernel needs (64 lanes x WPL words x 64 columns, WPL = 1, 2, 4, 8: up to 32768 columns), 0 = only the generic kernel is left
static inline uint32_t bsa_edit_class(uint32_t bw){
	if(bw <= BSA_EDIT_REG_BW) return bw;
	const uint32_t nw = bw / 64u;
	return nw <= 64u ? 0xFFFFFF01u : nw <= 128u ? 0xFFFFFF02u : nw <= 256u ? 0xFFFFFF04u : nw <= 512u ? 0xFFFFFF08u : 0xFFFFFF00u;
}

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
	int32_t        *fwd_smin, *fwd_ry;  // [n] overlap / extend: min over rows of H at the last query column and its (first) row,
	                                //   followed while the row is in registers (bsalign.h:1124-1139)
	uint32_t first, count;
	uint32_t bw;                    // effective bandwidth of this launch (multiple of 64); 0 = wide class, every pair has its own
	                                //   (bsa_edit_bw_eff of its lengths), all above 1024 -- one launch of the generic kernel
	uint32_t bandwidth;             // the caller's bandwidth parameter, for bsa_edit_bw_eff
	uint32_t wide;                  // bw == 0 only: words per lane of the wave-per-pair kernel for static bands (1, 2, 4, 8), 0 = none
	uint32_t pad_rows;              // spare row records at the end of every slot (CIGAR scratch)
	int32_t  mode;
	uint32_t row_fmt;               // 0: row r = [plane0: NW u64][plane1: NW u64]; 1: TILED (k_edit_fwd_grp32 + k_edit_trace_wave, bsa_edit_row_dword)
};

// Row format 1 of the edit path (round 6): the walk of a banded alignment needs, of every row, the 32 columns around its diagonal -- one or two of a
// row's 32-bit words per plane -- but a 64-byte row is one memory request whatever part of it is wanted (profiles/r06_fetch_size_calibration.txt),
// and the walker, reading every row whole, was at the HBM ceiling.  So rows are TILED eight at a time: tile t = rows 8 t .. 8 t + 7, inside it the
// 64-byte block of 32-bit column word h holds that word of both planes of the eight rows -- dword (r & 7) * 2 + plane -- and a lane of the forward
// kernel (which owns word h of its pair) fills one block over eight consecutive rows.  The walker fetches three blocks per tile instead of eight.
static inline __host__ __device__ size_t bsa_edit_row_dword(uint32_t NH, uint32_t r, uint32_t plane, uint32_t h){      // dword index of (row, plane, 32-bit word) in a tiled slot; NH = words per plane
	return (size_t)(r >> 3) * (16u * NH) + (size_t)h * 16u + (r & 7u) * 2u + plane;
}

#ifdef __HIPCC__
// Buffered tile writer used by the forward kernels: RW dwords per block record, TG rows per tile.
template<int RW, int TG>
struct TileWriter {
	uint32_t hist[(TG > 1) ? (TG - 1) : 1][RW];
	static constexpr uint32_t BLK = RW * 4u, TILEB = (TG * BLK + 15u) & ~15u;
	// push the record of row index rr (= target row + 1); `last` = this is the final row of the pair
	__device__ __forceinline__ void push(uint8_t *tiles, uint32_t j, uint32_t rr, bool active, bool last, const uint32_t (&rec)[RW]){
		const uint32_t slot = rr % (uint32_t)TG;
		if(active && (slot == (uint32_t)TG - 1u || last)){
			uint32_t *tile = (uint32_t*)(tiles + ((size_t)(rr / (uint32_t)TG) * 16u + j) * TILEB);
			if(slot == (uint32_t)TG - 1u){
#pragma unroll
				for(int m = 0; m < TG - 1; m++){
#pragma unroll
					for(int d = 0; d < RW; d++) tile[m * RW + d] = hist[m][d];
				}
#pragma unroll
				for(int d = 0; d < RW; d++) tile[(TG - 1) * RW + d] = rec[d];
			} else {           // partial last group of the pair
#pragma unroll
				for(int m = 0; m < TG - 1; m++){
					const int ds = m - (TG - 1 - (int)slot);
					if(ds >= 0){
#pragma unroll
						for(int d = 0; d < RW; d++) tile[ds * RW + d] = hist[m][d];
					}
				}
#pragma unroll
				for(int d = 0; d < RW; d++) tile[slot * RW + d] = rec[d];
			}
		}
		if(TG > 1){
#pragma unroll
			for(int m = 0; m + 1 < TG - 1; m++){
#pragma unroll
				for(int d = 0; d < RW; d++) hist[m][d] = hist[m + 1][d];
			}
#pragma unroll
			for(int d = 0; d < RW; d++) hist[(TG > 1) ? (TG - 2) : 0][d] = rec[d];
		}
	}
};
#endif

// launchers implemented in the kernel translation units
hipError_t bsa_launch_align8_fwd(const Align8Args &a, int pw, hipStream_t st);
hipError_t bsa_launch_align8_backcal(const Align8Args &a, int pw, bsa_result_t *out, uint32_t *cig_cnt, hipStream_t st);
bool bsa_align8_supported_bw(uint32_t bw);
size_t bsa_align8_gen_lds(uint32_t bw, int pw, uint32_t nbuf = 2u);
hipError_t bsa_launch_align8_fwd_gen(const Align8Args &a, int pw, uint32_t max_bw, hipStream_t st);
bool bsa_align8_pk_supported(const Align8Args &a, int pw);
hipError_t bsa_launch_align8_fwd_pk(const Align8Args &a, int pw, hipStream_t st);
bool bsa_align8_codes_supported(const Align8Args &a, int pw);          // global mode, piecewise <= 1, small scores
hipError_t bsa_launch_align8_fwd_codes(const Align8Args &a, int pw, hipStream_t st);
bool bsa_align8_x_supported(const Align8Args &a, int pw);              // exact-arithmetic forward kernel of the compact path (bsa_align8_x.hip)
bool bsa_align8_do2_supported(const Align8Args &a, int pw);            // ... able to write code format 1
bool bsa_align8_trace_reads_do2(const Align8Args &a, int pw);          // the traceback kernel the launcher would pick reads code format 1
hipError_t bsa_launch_diagdp(const uint8_t *d_planes, const bsa_diagdp_prob_t *d_probs, uint32_t *d_T, const uint64_t *d_toff, uint8_t *d_matrix,
		uint32_t n, uint32_t W, uint32_t max_len, hipStream_t st);
hipError_t bsa_launch_diagdp_walk(const uint8_t *d_planes, const bsa_diagdp_prob_t *d_probs, const uint32_t *d_T, const uint64_t *d_toff, const uint8_t *d_matrix,
		uint32_t n, bsa_diagdp_walk_t *d_walks, uint32_t *d_steps, const uint64_t *d_word_off, hipStream_t st);
// the kernels the launchers picked last (this thread), for bsa_ctx_last_kernel_name
extern thread_local const char *bsa_last_fwd_kernel, *bsa_last_trace_kernel;
hipError_t bsa_launch_align8_fwd_x(const Align8Args &a, int pw, hipStream_t st);
size_t bsa_align8_xq_bytes(uint32_t bw, int pw, uint32_t count);       // what Align8Args::xq must hold for a launch of `count` pairs (0: no persistent form)
// whole-query bands above 256 columns, global mode: systolic wavefront + its own code layout and traceback (bsa_align8_sys.hip)
int bsa_align8_sys_supported(const Align8Args &a, int pw);          // 0 no, 1 inside the static guard, 2 with the kernel checking every pair (Align8Args::sys_chk)
size_t bsa_align8_sys_slot_bytes(uint32_t qlen, uint32_t tlen, int pw);
hipError_t bsa_launch_align8_fwd_sys(const Align8Args &a, int pw, uint32_t max_qlen, hipStream_t st);
hipError_t bsa_launch_align8_trace_sys(const Align8Args &a, int pw, bsa_result_t *out, uint32_t *cig_cnt, const uint64_t *slot_end, hipStream_t st);
hipError_t bsa_launch_align8_trace_codes(const Align8Args &a, int pw, bsa_result_t *out, uint32_t *cig_cnt, hipStream_t st);
bool bsa_edit_supported_bw(uint32_t bw);
bool bsa_edit_tiled_ok(uint32_t bw, uint32_t count, int mode);       // row format 1 for a launch class that fills its chunk alone (bsa_edit.hip)
hipError_t bsa_launch_edit_stage(const uint8_t *seqs, const uint64_t *qoff, const uint32_t *qlen, const uint64_t *toff, const uint32_t *tlen,
		const uint64_t *qpoff, const uint64_t *tpoff, const uint64_t *qboff, const uint32_t *qwords,
		uint8_t *qst, uint8_t *tst, uint64_t *qbits, uint32_t *status, uint32_t n, hipStream_t st);
hipError_t bsa_launch_edit_fwd(const EditArgs &a, hipStream_t st);
hipError_t bsa_launch_edit_trace(const EditArgs &a, bsa_result_t *out, uint32_t *cig_cnt, hipStream_t st);
