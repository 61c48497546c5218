// bsa_align8_x.hip -- forward pass of the compact (4-bit code) path of the 8-bit banded DP in EXACT arithmetic.
//
// Same results as k_align8_fwd_pk<W, 1, true> (bsa_align8_pk.hip): the same code rows, band offsets and final score,
// bit for bit -- the traceback kernels (bsa_align8_codes.hip) do not know which forward kernel ran.  It is only
// dispatched inside a guard (bsa_align8_x_supported) under which none of the reference's int8-saturating operations
// can clamp (bsalign.h:2885-2960 row_cal, :2639-2652 F-penetration, :2618-2636 tail, :2244-2392 row_movx), so the
// recurrence may be restated freely.  What changed against the packed kernel, and why (tools/valu_rate_probe.hip:
// every packed / VOP3 / DPP instruction occupies its SIMD for 4 cycles and the old kernel issued 631 of them per row
// of 8 pairs, at 100 % of that rate):
//   * ONE pair per 8 lanes, both 16-bit halves of a register belong to the SAME pair: lane l owns the reference's
//     running blocks l (low half) and l + 8 (high half).  Per-pair scalars (band offset, steering, ubegs[0]) are
//     computed once, not once per half.
//   * frame shifted by the gap extension: with H~(x, y) = H(x, y) - gape * (x + y) extending a gap is free, which
//     removes the "+ gape" of e' and f' (U = u - gape, NE = gape - e, S~ = S - 2 gape, h~ = h - 2 gape, f~ = f - 2 gape):
//         ee = U - NE          m = max(ee, S~)          mg = m + gapo
//         pass 1:  f = max(f, mg) - U                                  (2 dependent ops per cell)
//         pass 2:  h = max(m, f)   fm = max(f, mg)   f = fm - U   n = h - ee   NE' = min(n, -gapo)   U' = h - v   v = h - U
//     and the four flags of a cell are differences the recurrence has anyway: M: h - S~ == 0, D: n == 0,
//     R: fm - mg == 0, Od: NE' == -gapo, each one packed min / saturating subtract + one v_pk_mad_u16.
//   * F-penetration as a prefix maximum: in the shifted frame an F value travels along the row unchanged, so with
//     X[b] = H~ at the end of block b (relative, int16) the value entering block b is max_{i<b}(fout[i] + X[i]) - X[b-1]:
//     a 3-step max scan over the 8 lanes plus one step from the low halves to the high ones, instead of the max-plus scan.
//   * the band slide is speculated: 96 % of the rows move the band by one cell, so pass 2 writes cell k of the new row
//     into slot k - 1 and the first cell of every block travels to the previous block's last slot with one DPP move.
//     Rows that do not move (or move by more) are corrected afterwards, under a wave-level branch.
//   * ubegs are kept as PN[b] = ubegs[b+1] - ubegs[0] - (b+1) W gape in packed int16 plus ubegs[0] in an int32.
// Values are value << 8 in each int16 half (as in the packed kernel), so S~ comes out of v_perm_b32 byte lookups.
#include "bsa_common.h"
#include "bsa_dpp.h"
#include <algorithm>
#ifndef XQ_BEGS16
#define XQ_BEGS16 0              // 1: band offsets leave sixteen rows (one 64-byte line) at a time instead of four (16 bytes at an odd dword).  Measured at C2 with
                                 // slots at multiples of 256 bytes: WRITE_SIZE 80.0 -> 69.7 GB (= the 69.6 GB the kernel has to store), forward 59.6 -> 60.2 ms (two-piece
                                 // gaps 123.9 -> 125.5): the partial lines cost traffic, not time, and the kernel is bound by its instructions -- off
#endif

typedef short xv2s __attribute__((ext_vector_type(2)));
typedef unsigned short xv2u __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ uint32_t x_add(uint32_t a, uint32_t b){ return __builtin_bit_cast(uint32_t, (xv2u)(__builtin_bit_cast(xv2u, a) + __builtin_bit_cast(xv2u, b))); }
static __device__ __forceinline__ uint32_t x_sub(uint32_t a, uint32_t b){ return __builtin_bit_cast(uint32_t, (xv2u)(__builtin_bit_cast(xv2u, a) - __builtin_bit_cast(xv2u, b))); }
static __device__ __forceinline__ uint32_t x_max(uint32_t a, uint32_t b){ return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(xv2s, a), __builtin_bit_cast(xv2s, b))); }
static __device__ __forceinline__ uint32_t x_minu(uint32_t a, uint32_t b){ return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(xv2u, a), __builtin_bit_cast(xv2u, b))); }
static __device__ __forceinline__ uint32_t x_satsubu(uint32_t a, uint32_t b){ return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(xv2u, a), __builtin_bit_cast(xv2u, b))); }
// acc * 2 + flag per half: one v_pk_mad_u16 as long as the compiler cannot see that `two` is the constant 0x00020002 (it
// would turn the product into a shift and need a second instruction).  No inline asm in the row loop: on gfx950 the
// hazard recognizer puts a wait state behind every asm statement whose result a VALU instruction reads.
static __device__ __forceinline__ uint32_t x_acc(uint32_t acc, uint32_t flag, uint32_t two){
	return __builtin_bit_cast(uint32_t, (xv2u)(__builtin_bit_cast(xv2u, acc) * __builtin_bit_cast(xv2u, two) + __builtin_bit_cast(xv2u, flag)));
}
static __device__ __forceinline__ uint32_t x_ashr8(uint32_t a){ return __builtin_bit_cast(uint32_t, (xv2s)(__builtin_bit_cast(xv2s, a) >> 8)); }
static __device__ __forceinline__ uint32_t x_shl8(uint32_t a){ return __builtin_bit_cast(uint32_t, (xv2u)(__builtin_bit_cast(xv2u, a) << 8)); }
// x = take ? y : x, in place (the band corrections: a plain C select makes the register allocator copy the whole band)
static __device__ __forceinline__ void x_sel(uint32_t &x, uint32_t y, uint64_t take){ asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "s"(take)); }
static __device__ __forceinline__ uint32_t x_q8(int v){ return (((uint32_t)v & 0xffu) << 8) * 0x00010001u; }        // value << 8 in both halves
static __device__ __forceinline__ uint32_t x_i16(int v){ return ((uint32_t)v & 0xffffu) * 0x00010001u; }            // plain int16 in both halves
static __device__ __forceinline__ int x_lo8(uint32_t x){ return __builtin_amdgcn_sbfe((int)x, 8, 8); }              // value of the low half (value << 8 form)
static __device__ __forceinline__ int x_hi8(uint32_t x){ return (int)x >> 24; }
static __device__ __forceinline__ int x_lo16(uint32_t x){ return __builtin_amdgcn_sbfe((int)x, 0, 16); }
static __device__ __forceinline__ int x_hi16(uint32_t x){ return (int)x >> 16; }

// (no dpp_keep here: every consumer of these moves is a packed, a three-operand or a select instruction, none of which can
// absorb a DPP operand, so the v_subrev_u32_dpp fold that bsa_dpp.h guards against cannot happen)
#define XDPP(old, x, ctrl, bank) ((uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(x), (ctrl), 0xf, (bank), false))
// lanes without a source read 0 (bound_ctrl): the same values as XDPP(0, ...), but the compiler need not put a zero into the destination first -- with
// old = 0 every one of these moves was preceded by a v_mov_b32 v, 0 (ten a row on the headline shape)
#define XDPPZ(x, ctrl, bank) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), 0xf, (bank), true))
#define XQP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define XROW_SHL(n) (0x100 + (n))
#define XROW_SHR(n) (0x110 + (n))
#define XHALF_MIRROR 0x141

// A pair owns a group of L = 8 or 4 consecutive lanes (two or four groups per 16-lane DPP row).
// value of the first / last lane of every group, in all its lanes
template<int L> static __device__ __forceinline__ uint32_t x_bcast_first(uint32_t x){
	const uint32_t s = XDPPZ(x, XQP(0, 0, 0, 0), 0xf);
	if constexpr (L == 4) return s;
	else return XDPP(s, s, XROW_SHR(4), 0xA);          // lanes 4..7 <- lanes 0..3
}
template<int L> static __device__ __forceinline__ uint32_t x_bcast_last(uint32_t x){
	const uint32_t s = XDPPZ(x, XQP(3, 3, 3, 3), 0xf);
	if constexpr (L == 4) return s;
	else return XDPP(s, s, XROW_SHL(4), 0x5);          // lanes 0..3 <- lanes 4..7
}
// block b receives the value of block b - 1 (low half: block l, high half: block l + L); block 0 receives fill_lo
template<int L> static __device__ __forceinline__ uint32_t x_shift_down(uint32_t x, uint32_t fill_lo, bool first){
	if constexpr (L == 4){
		// a pair is one DPP quad: ONE rotation brings every lane its neighbour and lane 0 the last lane's value, ONE byte permute (the
		// selector is a loop invariant of the lane) puts that value's low half above the fill in lane 0 and leaves the others as they are
		const uint32_t r = XDPPZ(x, XQP(3, 0, 1, 2), 0xf);
		return __builtin_amdgcn_perm(r, fill_lo, first ? 0x05040100u : 0x07060504u);
	}
	const uint32_t s = XDPPZ(x, XROW_SHR(1), 0xf);
	const uint32_t w = XDPPZ(x, XROW_SHL(L - 1), 0xf);  // first lane <- last lane
	const uint32_t fix = (w << 16) | (fill_lo & 0xffffu);
	return first ? fix : s;
}
// block b receives the value of block b + 1; block 2L - 1 receives fill (given in both halves)
template<int L> static __device__ __forceinline__ uint32_t x_shift_up(uint32_t x, uint32_t fill, bool last){
	if constexpr (L == 4){
		const uint32_t r = XDPPZ(x, XQP(1, 2, 3, 0), 0xf);             // (lane 3 receives lane 0's value: its high half goes below the fill)
		return __builtin_amdgcn_perm(fill, r, last ? 0x05040302u : 0x03020100u);
	}
	const uint32_t s = XDPPZ(x, XROW_SHL(1), 0xf);
	const uint32_t w = XDPPZ(x, XROW_SHR(L - 1), 0xf);  // last lane <- first lane
	const uint32_t fix = __builtin_amdgcn_alignbit(fill, w, 16);        // {fill.lo, w.hi}
	return last ? fix : s;
}
// inclusive prefix maximum over the lanes of a group, per half (Sklansky: nothing crosses a group)
template<int L> static __device__ __forceinline__ uint32_t x_scan_max(uint32_t x){
	uint32_t y = XDPP(x, x, XQP(0, 0, 2, 2), 0xf); x = x_max(x, y);
	y = XDPP(x, x, XQP(0, 1, 1, 1), 0xf); x = x_max(x, y);
	if constexpr (L == 8){
		const uint32_t z = XDPP(x, x, XQP(3, 3, 3, 3), 0xf);
		y = XDPP(x, z, XROW_SHR(4), 0xA); x = x_max(x, y);
	}
	return x;
}
// sum over the lanes of a group, per half, in every lane
template<int L> static __device__ __forceinline__ uint32_t x_sum(uint32_t x){
	x = x_add(x, XDPPZ(x, XQP(1, 0, 3, 2), 0xf));
	x = x_add(x, XDPPZ(x, XQP(2, 3, 0, 1), 0xf));
	if constexpr (L == 8) x = x_add(x, XDPPZ(x, XHALF_MIRROR, 0xf));
	return x;
}

template<int W>
static __device__ __forceinline__ void x_load_qcodes(const uint8_t *p, uint32_t *w){
	if constexpr (W >= 4) __builtin_memcpy(w, p, W);
	else if constexpr (W == 2){ uint16_t v; __builtin_memcpy(&v, p, 2); w[0] = v | 0x04040000u; }
	else w[0] = p[0] | 0x04040400u;
}

// W cells per lane and half, L lanes per pair: 2 L blocks of W cells; the reference's 16 running blocks have WR = 2 L W / 16
// cells, so a block here holds CR = 8 / L of them
// PW = 2: two-piece gaps.  The frame is shifted by the extension of piece 1, so piece 2 keeps a real step dP = gape2 - gape1 > 0:
//   qq = U - NQ      m = max(ee, qq, S~)      g = max(g, max(m, f) + gapo2) + dP - U      NQ' = min(h - qq, -gapo2) - dP
// and a cell has nine facts (bsa_common.h "COMPACT slot", 8 bits): M, D, D2, which chain equals h (I1, I2), R1, R2, Od1, Od2.
// STATIC: every pair of the launch has a band that covers its whole query (Align8Args::static_band): the band never moves, so
// the row is kept in place -- no speculative slide, no corrections of it, no band steering.
// Row segments (k_align8_fwd_xq): the rows [row0, row1) of the pairs (row0, row1 multiples of 8, so that no group of four code rows
// and no group of L band offsets is open at a cut); `st` = the wave's state block (XS_WORDS(W, PW) x 64 dwords, dword r of lane l at
// st[64 r + l]): loaded when row0 != 0, stored when a pair has rows left behind row1.  tid_base + threadIdx.x = the lane's index among
// the launch's lanes (L consecutive ones share a pair).  The plain kernels pass row0 = 0, row1 = 0xFFFFFFF8, st = nullptr.
#define XS_WORDS(W, PW) ((((PW) == 2) ? 3 : 2) * (W) + 11)
// NWV: waves per workgroup.  The code dwords of a group of four rows wait in LDS (a lane's own sixteen dwords, no synchronisation) until
// the group is complete and leaves as 16-byte pieces: kept in registers they were twelve of the 168 a wave may have at three per SIMD.
// (Stored row by row as four-byte pieces the same launch takes 40 % longer.)
// DO2 (one-piece gaps with 1 <= -gapo <= 3, bandwidth 128; Align8Args::code_fmt == 1): D and Od of a cell leave as ONE two-bit field,
// the new e-difference min(h - ee, -gapo) itself (0: D, -gapo: Od) -- one multiply-add a cell where the two planes take four
// instructions.  A reference block's dword is then M | R << 8 | field of cell c at bits 31 - 2c, 30 - 2c (bsa_common.h).
// EXT: the two LDS areas come from the caller (k_align8_fwd_x_mix runs two shapes of this function in one launch and a block only ever one of
// them: they share the areas instead of each instantiation carrying its own).
template<int W, int L, int PW = 1, bool STATIC = false, int NWV = 4, bool DO2 = false, bool EXT = false>
static __device__ __forceinline__ void x_forward(const Align8Args &a, const uint32_t first_pos, const uint32_t count, const uint32_t tid_base,
		const uint32_t row0 = 0u, const uint32_t row1 = 0xFFFFFFF8u, uint32_t *st = nullptr, uint32_t *ext_stage = nullptr, uint32_t *ext_qwin = nullptr){
	constexpr int BW = 2 * L * W;
	constexpr int WR = BW / 16, CR = 8 / L;
	static_assert((L == 8 || L == 4) && W * L / 8 == WR && (CR == 1 || (CR == 2 && (W == 16 || W == 8))), "supported shapes");
	constexpr int NQ = (W + 3) / 4;
	constexpr int NACC = (W + 7) / 8;                     // flag accumulators per plane (8 cells each)
	constexpr int TOPBIT = 8 + ((W < 8) ? W : 8) - 1;     // accumulator bit of the first cell it holds
	static_assert(PW == 0 || PW == 1 || (PW == 2 && ((W == 8 && (L == 8 || L == 4)) || (W == 16 && (L == 8 || L == 4)))), "two-piece gaps: bandwidth 128 (eight lanes per pair), 64 (four), 256 (eight, sixteen cells a half)");
	static_assert(!DO2 || (PW == 1 && WR == 8), "two-bit D / Od fields: one-piece gaps, bandwidth 128");
	constexpr int NDO = DO2 ? W / 4 : 1;                  // accumulators of the two-bit fields (four cells each, in the high byte of a half)
	constexpr int CWD = (PW == 2) ? WR / 4 : (WR >= 8) ? WR / 8 : 1;          // code dwords per reference block and row (two-piece gaps: eight bits a cell)
	constexpr int ND = (PW == 2) ? 2 * W / 4 : (WR == 8) ? 2 * NACC : (WR == 16) ? 4 : (W == 8) ? 4 : 2;      // code dwords per lane and row
	const int lt = threadIdx.x;
	const int jl = lt & (L - 1);
	const bool first = jl == 0, last = jl == L - 1;
	const uint32_t g = (tid_base + (uint32_t)lt) / (uint32_t)L;
	const bool live = g < count;
	const uint32_t ppos = first_pos + (live ? g : 0u);
	const uint32_t pair = a.order[ppos];
	const uint32_t qlen = a.qlen[pair];
	uint32_t tlen = a.tlen[pair];
	const uint8_t *qp = a.qst + a.qpoff[pair], *tp = a.tst + a.tpoff[pair];
	int *begs = (int*)(a.rows + a.slot_off[ppos]);
	uint8_t *rowp = (uint8_t*)begs + bsa_begs_bytes(tlen);
	if(!live || a.status[pair] != 0u) tlen = 0;

	const int mode = a.mode & 3;
	const int gapo1 = a.gapo1, gape1 = a.gape1;
	const int GE = gape1, GO = gapo1;                      // GO <= 0, GE <= 0 (bsa_align8_x_supported)
	const uint32_t GOQ = x_q8(GO);                         // gapo, value << 8
	const uint32_t NGOQ = x_q8(-GO), NGOQ1 = x_q8(-GO - 1);
	const uint32_t ONE = 0x01000100u;
	uint32_t TWO = 0x00020002u;
	asm volatile("" : "+s"(TWO));                          // opaque multiplier of x_acc
	// DO2: cell j of an accumulator's four goes to bits 15 - 2j, 14 - 2j of the half: multipliers 64, 16, 4 (and 1: an addition)
	uint32_t KDO0 = 0x00400040u, KDO1 = 0x00100010u, KDO2 = 0x00040004u;
	asm volatile("" : "+s"(KDO0), "+s"(KDO1), "+s"(KDO2));
	const uint32_t DOSP = (GO == -1) ? 2u : 1u;            // a field value that is neither 0 (D) nor -gapo (Od)
	const uint32_t MINF = x_q8(BSA_EPI8_MIN - 2 * GE);     // the -63 sentinel of f in the shifted frame
	const uint32_t NGEQ = x_q8(-GE);                       // a cell with u = 0
	const int gapo2 = a.gapo2, gape2 = a.gape2;
	const int DP = (PW == 2) ? gape2 - gape1 : 0;          // > 0
	const int gopen = (PW == 2) ? gapo2 + gape2 : gapo1 + gape1;
	const int cfirst = min(a.smin, gopen) - 1 - a.smax + gopen;                        // bsalign.h:2362
	const int dsw = (PW == 2) ? (gapo1 - gapo2) / (gape2 - gape1) : 0x7FFFFFFF;         // new cells beyond this distance extend with piece 2 (bsalign.h:2369-2389)
	// cells entering at the band end (bsalign.h:2357-2389): u = cfirst for the first one, gape1 (gape2) after it, e = q = 0
	const uint32_t NEWU0 = x_q8(cfirst - GE), NEWNE = (PW == 0) ? 0u : x_q8(GE);
	const uint32_t GQQ = x_q8(gapo2), NGQQ = x_q8(-gapo2), NGQQ1 = x_q8(-gapo2 - 1), DPQ = x_q8(DP);
	const uint32_t GE16 = x_i16(GE), WGE16 = x_i16(W * GE);
	const uint32_t PADS = (uint32_t)((BSA_EPI8_MIN - 2 * GE) & 0xff) * 0x01010101u;    // S~ beyond the query end
	// S~ rows per target base: four dwords in LDS, read by the lanes with their row's base as the index (every wave of the block writes the same values)
	__shared__ uint32_t x_mtab[4];
	if((lt & 63) < 4){
		const int t = lt & 3;
		uint32_t w = 0;
#pragma unroll
		for(int q = 0; q < 4; q++) w |= (uint32_t)(((int)a.matrix[q * 4 + t] - 2 * GE) & 0xff) << (8 * q);
		x_mtab[t] = w;
	}
	__builtin_amdgcn_wave_barrier();

	uint32_t U[W], NE[W], NQ2[(PW == 2) ? W : 1];     // NQ2: gape1 - q of piece 2
	uint32_t PN;                  // packed int16: ubegs[b+1] - ubegs[0] - (b+1) W gape for b = jl (low) and jl + L (high)
	uint32_t PM = 0;              // CR == 2: the same for the middle of the block (the end of its first reference block)
	int HB;                       // ubegs[0]
	uint32_t svU = 0, svNE = 0, svNQ = 0; // first cell of the row before the speculative slide (lane 0, low half: band position 0)
	uint32_t rbeg = 0, mov = 0, i = row0;
	int cand_sc = BSA_SCORE_MIN, cand_te = 0;       // overlap / extend: best end-of-query score this lane has seen, and its row
	// ---- row -1 (bsalign.h:2094-2140)
	if(row0 == 0u){
		const int first_u = (int)(int8_t)(gapo1 + gape1 + a.smin - a.smax);
		const int xp = (PW == 2) ? (gapo2 - gapo1) / (gape1 - gape2) : 0x7FFFFFFF;      // row -1 extends with piece 2 from here on (bsalign.h:2115-2125)
		auto u_init = [&](int p) -> int { return (mode == BSA_MODE_OVERLAP) ? 0 : (p == 0) ? first_u : (p < xp) ? gape1 : gape2; };
#pragma unroll
		for(int k = 0; k < W; k++){
			const int vlo = u_init(jl * W + k), vhi = u_init((jl + L) * W + k);
			U[k] = (((uint32_t)(vlo - GE) & 0xffu) << 8) | (((uint32_t)(vhi - GE) & 0xffu) << 24);
			NE[k] = (PW == 0) ? 0u : x_q8(GE - BSA_EPI8_MIN);      // linear gaps: e is the constant gape1 (bsalign.h:2779), NE stays 0
			if constexpr (PW == 2) NQ2[k] = x_q8(GE - BSA_EPI8_MIN);
		}
		if(mode == BSA_MODE_OVERLAP){
			PN = ((uint32_t)(-(jl + 1) * W * GE) & 0xffffu) | ((uint32_t)(-(jl + L + 1) * W * GE) << 16);
			PM = x_add(PN, x_i16((W / 2) * GE));
			HB = 0;
		} else {
			// sum of (u - gape1) over the cells in front of the block's end
			auto pn_at = [&](int cells) -> int { return (first_u - GE) + DP * max(0, cells - max(xp, 1)); };
			auto pn_init = [&](int b) -> int { return pn_at((b + 1) * W); };
			PN = ((uint32_t)pn_init(jl) & 0xffffu) | ((uint32_t)pn_init(jl + L) << 16);
			PM = ((uint32_t)pn_at(jl * W + W / 2) & 0xffffu) | ((uint32_t)pn_at((jl + L) * W + W / 2) << 16);       // (the same number with one piece; with two, piece 2's cells in the block's second half do not count yet)
			HB = a.smax - a.smin;
		}
	}
	// bring row -1 into the loop's form: slid by one cell, ubegs[0] not advanced
	if constexpr (!STATIC) if(row0 == 0u){
		const uint32_t t0u = U[0], t0e = NE[0];
#pragma unroll
		for(int k = 0; k + 1 < W; k++){ U[k] = U[k + 1]; NE[k] = NE[k + 1]; }
		const uint32_t inu = x_shift_up<L>(t0u, NEWU0, last), inne = x_shift_up<L>(t0e, NEWNE, last);
		U[W - 1] = inu; NE[W - 1] = inne;
		if constexpr (PW == 2){
			svNQ = NQ2[0];
#pragma unroll
			for(int k = 0; k + 1 < W; k++) NQ2[k] = NQ2[k + 1];
			NQ2[W - 1] = x_shift_up<L>(svNQ, NEWNE, last);
		}
		PN = x_add(x_add(PN, x_ashr8(inu)), GE16);
		if constexpr (CR == 2) PM = x_add(x_add(PM, x_ashr8(U[W / 2 - 1])), GE16);
		svU = t0u; svNE = t0e;
	}
	if(row0 != 0u){
		// the state the previous segment left (plain loads behind the acquire of k_align8_fwd_xq)
		const uint32_t *sp = st + (lt & 63);
		int r = 0;
#pragma unroll
		for(int k = 0; k < W; k++) U[k] = sp[64 * r++];
#pragma unroll
		for(int k = 0; k < W; k++) NE[k] = sp[64 * r++];
		if constexpr (PW == 2){
#pragma unroll
			for(int k = 0; k < W; k++) NQ2[k] = sp[64 * r++];
		}
		PN = sp[64 * r++]; PM = sp[64 * r++]; HB = (int)sp[64 * r++]; svU = sp[64 * r++]; svNE = sp[64 * r++]; svNQ = sp[64 * r++];
		rbeg = sp[64 * r++]; mov = sp[64 * r++]; cand_sc = (int)sp[64 * r++]; cand_te = (int)sp[64 * r++];
	}
	__shared__ uint32_t x_stage[EXT ? 1 : NWV][EXT ? 1 : 4 * ND][64];       // [wave][4 q + row of the group (CWD == 1) | ND row + dword (CWD >= 2)][lane]
	uint32_t *const stg = EXT ? ext_stage + (size_t)(lt >> 6) * (4 * ND * 64) + (lt & 63) : &x_stage[(!EXT && NWV > 1) ? (lt >> 6) : 0][0][lt & 63];
	// QUERY WINDOW (round 5).  A lane needs the W query codes of each of its two blocks on every row, at a band offset that moves by about one base a
	// row: loaded from memory row by row that is one full round trip the wave waits for per row -- and `vmcnt` also holds the row's wait back behind the
	// code-row stores in flight (a second such round trip per row, added as an experiment, cost 5.4 ms of the 61: 9 %).  Instead the lane keeps, in LDS
	// slots of its own (no synchronisation, like the staging above), KD dwords per block starting at the band offset of the last refill: a row reads
	// NQ + 1 of them at the dword its offset has reached and shifts them into place (v_alignbyte), and only every 4 KD - W - 3 bases of band movement
	// (about 45 rows) the window is loaded again.  The staged query is padded far enough behind its end (plan: bandwidth + 32 bytes).
#ifdef XQ_NO_QWIN
	constexpr bool QWIN = false;
#else
	constexpr bool QWIN = !STATIC && PW != 2 && W == 16;      // (measured: two-piece gaps at 251 registers lose 3 % with it, eight cells a half gain nothing)
#endif
	// band offsets leave sixteen rows at a time through four LDS dwords a lane (below, "band offsets"); the window gives them up -- 40 instead of 48 bytes of
	// band movement between refills -- so that the block's LDS stays what three waves per SIMD allow (one KB more a wave cost 5 % of the launch)
	constexpr bool BQ16 = XQ_BEGS16 && L == 4 && !EXT;
	__shared__ uint32_t x_bq[BQ16 ? NWV : 1][BQ16 ? 4 : 1][64];
	uint32_t *const bqp = &x_bq[(BQ16 && NWV > 1) ? (lt >> 6) : 0][0][lt & 63];
	constexpr int KD = NQ + ((BQ16 && QWIN) ? 10 : 12);   // dwords per block in the window
	constexpr uint32_t QOFFMAX = 4u * (uint32_t)(KD - NQ - 1) + 3u;      // the last offset at which dwords k .. k + NQ are all inside
	static_assert(!QWIN || 4 * KD - W <= BSA_QPAD_TAIL, "the window reads 4 KD - W bytes behind the band's last block: the staged query's padding (bsa_api.hip: qpad = bandwidth + BSA_QPAD_TAIL) must cover it");
	__shared__ uint32_t x_qwin[(QWIN && !EXT) ? NWV : 1][(QWIN && !EXT) ? 2 * KD : 1][64];
	uint32_t *const qwp = EXT ? ext_qwin + (size_t)(lt >> 6) * (2 * KD * 64) + (lt & 63) : &x_qwin[(QWIN && !EXT && NWV > 1) ? (lt >> 6) : 0][0][lt & 63];
	uint32_t wbase = 0x40000000u;                          // band offset the window starts at (this value: no window yet)
	int begq = 0;
	if(tlen != 0u && first && row0 == 0u) begs[0] = 0;
	uint64_t twin = 0;
	if(row0 < tlen){ __builtin_memcpy(&twin, tp + row0, 8); twin <<= 2; }
	// STATIC: the band never moves, so a lane's query codes are the same on every row -- loaded once
	uint32_t sqlo[STATIC ? NQ : 1], sqhi[STATIC ? NQ : 1];
	if constexpr (STATIC){
#pragma unroll
		for(int n = 0; n < NQ; n++){ sqlo[n] = 0x04040404u; sqhi[n] = 0x04040404u; }
		if(tlen){ x_load_qcodes<W>(qp + jl * W, sqlo); x_load_qcodes<W>(qp + (jl + L) * W, sqhi); }
	}
	const int rbz = 2 * max((int)(tlen / max(qlen, 1u)), 1);          // bsalign.h:4008
	const bool rush32 = (unsigned long long)(uint32_t)rbz * tlen + qlen + (uint32_t)BW + (uint32_t)rbz < 0xFFFFFFFFull;
	// rbz * (rows left after this one), kept by subtraction: a 32-bit multiplication is a quarter-rate instruction and this one sat in every row
	uint32_t rzl = rush32 ? (uint32_t)rbz * (tlen - min(row0, tlen) - 1u) : 0u;
	int rby_tab = 0;
	const int rby_lane = (lt & (64 - L)) << 2;                   // byte address of lane 0 of this group for ds_bpermute
	const uint32_t kd1 = last ? 0x01000000u : 0u;          // band cell bw - 1 after a slide by one: x == bw, no deletion there (bsalign.h:3672-3678)

	// Wave-level tests: ONE vector compare into a scalar mask, the pair's liveness ANDed in on the scalar unit.  (`__any(act && x)` is the library's
	// __ockl_wfany_i32 on an `and` of two conditions: the backend materialises the predicate as 0 / 1 in a register and compares it again -- two to three
	// vector instructions a test, six tests a row.)
	uint64_t actm = 0;
	while(i < row1 && (actm = __builtin_amdgcn_ballot_w64(i < tlen)) != 0ull){
		const bool act = i < tlen;
		if(!STATIC && mode == BSA_MODE_GLOBAL && (i & (uint32_t)(L - 1)) == 0u)
			rby_tab = (int)((1.0 * (double)(i + (uint32_t)jl) / (double)tlen) * (double)qlen);      // bsalign.h:4009, row i + jl (measured: even a three-instruction f32 stand-in for the fifteen f64 instructions would give 0.1 - 0.3 ms of 57)
		// ---- band offset of this row (bsalign.h:3932-3946)
		if constexpr (!STATIC){
			// mov = (mov != 0 && rbeg + BW < qlen) ? min(mov, qlen - (rbeg + BW)) : 0 -- the room as a saturating difference makes it one minimum
			mov = min(mov, __builtin_elementwise_sub_sat(qlen, rbeg + (uint32_t)BW));
			rbeg += mov;
		}
		int rh;
		if(rbeg) rh = BSA_SCORE_MIN;
		else if(mode == BSA_MODE_OVERLAP || i == 0) rh = 0;
		else if(PW < 2) rh = gapo1 + gape1 * (int)i;
		else rh = max(gapo1 + gape1 * (int)i, gapo2 + gape2 * (int)i);
		// ---- row_movx (bsalign.h:2244-2392): the row is held slid by one cell; correct what did not move that way
		if(!STATIC && __builtin_expect((__builtin_amdgcn_ballot_w64(mov != 1u) & actm) != 0ull, 0)){
			if(__any(act && mov >= (uint32_t)BW)){
				// the band jumped past everything it held: zero rows, every ubegs = SCORE_MIN (bsalign.h:2253-2259);
				// rh = H at the last cell of the previous row (getscore(bw - 1))
				const bool z = act && mov >= (uint32_t)BW;
				const uint32_t bc = x_bcast_last<L>(PN);
				const int rhz = HB + (x_hi16(bc) - cfirst) + BW * GE;
				if(z){
					rh = rhz;
#pragma unroll
					for(int k = 0; k < W; k++){ U[k] = NGEQ; NE[k] = NEWNE; if constexpr (PW == 2) NQ2[k] = NEWNE; }
					HB = BSA_SCORE_MIN;
					PN = ((uint32_t)(-(jl + 1) * W * GE) & 0xffffu) | ((uint32_t)(-(jl + L + 1) * W * GE) << 16);
					PM = x_add(PN, x_i16((W / 2) * GE));
				}
			}
			if(__any(act && mov == 0u)){
				// undo the slide
				const bool d = act && mov == 0u;
				const uint64_t dm = __ballot(d);
				const uint32_t outu = U[W - 1];
				const uint32_t pu = x_shift_down<L>(outu, svU, first), pe = x_shift_down<L>(NE[W - 1], svNE, first);
				const uint32_t npn = x_sub(x_sub(PN, x_ashr8(outu)), GE16);
				PN = d ? npn : PN;
				if constexpr (CR == 2){ const uint32_t npm = x_sub(x_sub(PM, x_ashr8(U[W / 2 - 1])), GE16); PM = d ? npm : PM; }
#pragma unroll
				for(int k = W - 1; k >= 1; k--){ x_sel(U[k], U[k - 1], dm); x_sel(NE[k], NE[k - 1], dm); }
				x_sel(U[0], pu, dm); x_sel(NE[0], pe, dm);
				if constexpr (PW == 2){
					const uint32_t pq = x_shift_down<L>(NQ2[W - 1], svNQ, first);
#pragma unroll
					for(int k = W - 1; k >= 1; k--) x_sel(NQ2[k], NQ2[k - 1], dm);
					x_sel(NQ2[0], pq, dm);
				}
			}
			// one more cell at a time for steps of two and more (the first new cell went in with the speculative slide)
			for(uint32_t s = 1; __any(act && mov < (uint32_t)BW && s < mov); s++){
				const bool d = act && mov < (uint32_t)BW && s < mov;
				const uint64_t dm = __ballot(d);
				const uint32_t bc = x_bcast_first<L>(U[0]);
				const uint32_t D0 = __builtin_amdgcn_perm(bc, bc, 0x01000100u);
				const uint32_t NEWU1 = ((int)s >= dsw) ? DPQ : 0u;          // u = gape1, or gape2 beyond the switch distance
				const uint32_t inu = x_shift_up<L>(U[0], NEWU1, last), inne = x_shift_up<L>(NE[0], NEWNE, last);
				const uint32_t npn = x_add(PN, x_ashr8(x_sub(inu, D0)));
				PN = d ? npn : PN;
				if constexpr (CR == 2){ const uint32_t npm = x_add(PM, x_ashr8(x_sub(U[W / 2], D0))); PM = d ? npm : PM; }
				HB += d ? (x_lo8(bc) + GE) : 0;
#pragma unroll
				for(int k = 0; k + 1 < W; k++){ x_sel(U[k], U[k + 1], dm); x_sel(NE[k], NE[k + 1], dm); }
				x_sel(U[W - 1], inu, dm); x_sel(NE[W - 1], inne, dm);
				if constexpr (PW == 2){
					const uint32_t innq = x_shift_up<L>(NQ2[0], NEWNE, last);
#pragma unroll
					for(int k = 0; k + 1 < W; k++) x_sel(NQ2[k], NQ2[k + 1], dm);
					x_sel(NQ2[W - 1], innq, dm);
				}
			}
		}
		if(!STATIC && mov != 0u && mov < (uint32_t)BW) rh = HB;   // getscore(mov - 1) of the previous row
		// ---- sequences, S~(x, y)
		uint32_t S[W];
		{
			// the target base of this row (i is uniform: the half of the eight-base window is picked on the scalar side), as the byte offset of its S~ row
			// (the window holds the bases times four: a base is 0 .. 3, so nothing crosses a byte)
			const uint32_t tw32 = (i & 4u) ? (uint32_t)(twin >> 32) : (uint32_t)twin;
			const uint32_t tb4 = __builtin_amdgcn_ubfe(tw32, 8u * (i & 3u), 4u);
			uint32_t qlo[NQ], qhi[NQ];
			if constexpr (STATIC){
#pragma unroll
				for(int n = 0; n < NQ; n++){ qlo[n] = act ? sqlo[n] : 0x04040404u; qhi[n] = act ? sqhi[n] : 0x04040404u; }
			} else if constexpr (QWIN){
				uint32_t off = rbeg - wbase;
				if(__builtin_expect((__builtin_amdgcn_ballot_w64(off > QOFFMAX) & actm) != 0ull, 0)){
					// refill: KD dwords of each block from the band offset of this row
					if(act && off > QOFFMAX){
						const uint8_t *pl = qp + rbeg + jl * W, *ph = qp + rbeg + (jl + L) * W;
						uint32_t bl[KD], bh[KD];
						__builtin_memcpy(bl, pl, 4 * KD); __builtin_memcpy(bh, ph, 4 * KD);
#pragma unroll
						for(int m = 0; m < KD; m++){ qwp[64 * m] = bl[m]; qwp[64 * (KD + m)] = bh[m]; }
						wbase = rbeg; off = 0u;
					}
				}
				const uint32_t kq = act ? (off >> 2) : 0u;
				const uint32_t *wl = qwp + 64u * kq;
				uint32_t dl[NQ + 1], dh[NQ + 1];
#pragma unroll
				for(int n = 0; n <= NQ; n++){ dl[n] = wl[64 * n]; dh[n] = wl[64 * (KD + n)]; }
#pragma unroll
				for(int n = 0; n < NQ; n++){
					const uint32_t vl = __builtin_amdgcn_alignbyte(dl[n + 1], dl[n], off), vh = __builtin_amdgcn_alignbyte(dh[n + 1], dh[n], off);
					qlo[n] = vl; qhi[n] = vh;          // (a lane whose pair has no row left computes on whatever its window holds: nothing of it is stored)
				}
			} else if(act){ x_load_qcodes<W>(qp + rbeg + jl * W, qlo); x_load_qcodes<W>(qp + rbeg + (jl + L) * W, qhi); }
			else {
#pragma unroll
				for(int n = 0; n < NQ; n++){ qlo[n] = 0x04040404u; qhi[n] = 0x04040404u; }
			}
			// (a four-way select among uniform values compiles to three nested exec-mask regions with v_readlane hazards: one LDS read instead)
			const uint32_t mr = *(const uint32_t*)((const uint8_t*)x_mtab + tb4);
			uint32_t slo[NQ], shi[NQ];
#pragma unroll
			for(int n = 0; n < NQ; n++){ slo[n] = __builtin_amdgcn_perm(PADS, mr, qlo[n]); shi[n] = __builtin_amdgcn_perm(PADS, mr, qhi[n]); }
#pragma unroll
			for(int k = 0; k < W; k++){
				const uint32_t sel = 0x000C000Cu | ((uint32_t)(k & 3) << 8) | ((uint32_t)(4 + (k & 3)) << 24);   // {0, lo.byte[k], 0, hi.byte[k]}
				S[k] = __builtin_amdgcn_perm(shi[k >> 2], slo[k >> 2], sel);
			}
		}
		// ---- first cell of the band (bsalign.h:2899-2907): h0 = rh - ubegs[0] + S, kept if >= u + e, else -63.  After a
		// slide inside the band rh == ubegs[0] and the rule changes neither h nor any flag, so only rows that stayed
		// (or jumped past the whole band: ubegs[0] = SCORE_MIN) need it.
		uint32_t hc0 = S[0];
		uint32_t q0m = 0, q0d = 0, q0d2 = 0;          // rows starting at query column 0: what h is compared with for M and D (bsalign.h:3763-3767)
		if(__builtin_expect((__builtin_amdgcn_ballot_w64(mov - 1u >= (uint32_t)BW - 1u) & actm) != 0ull, 0)){          // mov == 0 or mov >= BW
			const int s0 = x_lo8(S[0]) + 2 * GE, u0 = x_lo8(U[0]) + GE, e0 = GE - x_lo8(NE[0]);
			const int qq0 = (PW == 2) ? GE - x_lo8(NQ2[0]) : e0;
			const int t0 = u0 + max(e0, qq0);
			int hh = (rh - HB) + s0;
			hh = (hh >= t0) ? min(hh, BSA_EPI8_MAX) : BSA_EPI8_MIN;
			if(first && (mov == 0u || mov >= (uint32_t)BW)) hc0 = (hc0 & 0xffff0000u) | (((uint32_t)(hh - 2 * GE) & 0xffu) << 8);
			// compare values of the quirk, in the shifted frame; out of int8 range = never equal
			const int cm = rh - HB + s0 - 2 * GE, cd = u0 + e0 + rh - HB - 2 * GE, cd2 = u0 + qq0 + rh - HB - 2 * GE;
			q0m = (cm >= -128 && cm <= 127) ? (((uint32_t)cm & 0xffu) << 8) : 0x00ffu;
			q0d = (cd >= -128 && cd <= 127) ? (((uint32_t)cd & 0xffu) << 8) : 0x00ffu;
			q0d2 = (cd2 >= -128 && cd2 <= 127) ? (((uint32_t)cd2 & 0xffu) << 8) : 0x00ffu;
		}
		// ---- row_cal, pass 1 (bsalign.h:2911-2930): F leaving every block when nothing but the sentinel enters it
		uint32_t ee[W], m[W], mg[W], qq[(PW == 2) ? W : 1], Ud[(PW == 2) ? W : 1];
		uint32_t f = MINF, g2 = MINF;
#pragma unroll
		for(int k = 0; k < W; k++){
			ee[k] = x_sub(U[k], NE[k]);
			if constexpr (PW == 2){
				qq[k] = x_sub(U[k], NQ2[k]);
				Ud[k] = x_sub(U[k], DPQ);
				m[k] = x_max(x_max(ee[k], qq[k]), (k == 0) ? hc0 : S[k]);
				const uint32_t t = x_add(x_max(m[k], f), GQQ);
				g2 = x_sub(x_max(g2, t), Ud[k]);
			} else m[k] = x_max(ee[k], (k == 0) ? hc0 : S[k]);
			mg[k] = x_add(m[k], GOQ);
			f = x_sub(x_max(f, mg[k]), U[k]);
		}
		// ---- F-penetration (bsalign.h:2639-2652) as a prefix maximum over the blocks
		{
			const uint32_t Fa = x_add(x_ashr8(f), PN);
			const uint32_t P = x_scan_max<L>(Fa);
			const uint32_t bc = x_bcast_last<L>(P);
			const uint32_t Thi = (bc << 16) | 0x8000u;              // {-32768, max over blocks 0..7}
			const uint32_t G = x_sub(x_max(P, Thi), PN);
			f = x_shl8(x_shift_down<L>(G, x_i16(BSA_EPI8_MIN - 2 * GE), first));
			if constexpr (PW == 2){
				// the chain of piece 2 gains dP per cell on its way: the same scan on X - (b + 1) W dP
				const uint32_t PN2 = x_sub(PN, ((uint32_t)((jl + 1) * W * DP) & 0xffffu) | ((uint32_t)((jl + L + 1) * W * DP) << 16));
				const uint32_t Fb = x_add(x_ashr8(g2), PN2);
				const uint32_t P2 = x_scan_max<L>(Fb);
				const uint32_t bc2 = x_bcast_last<L>(P2);
				const uint32_t G2 = x_sub(x_max(P2, (bc2 << 16) | 0x8000u), PN2);
				g2 = x_shl8(x_shift_down<L>(G2, x_i16(BSA_EPI8_MIN - 2 * GE), first));
			}
		}
		// ---- pass 2 (bsalign.h:2932-2957), flags, new row written one slot to the left
		uint32_t accM[NACC], accD[NACC], accR[NACC], accO[NACC], accDO[NDO];
#pragma unroll
		for(int n = 0; n < NDO; n++) accDO[n] = 0;
		uint32_t accD2[NACC], accI1[NACC], accI2[NACC], accR2[NACC], accO2[NACC];          // PW == 2
#pragma unroll
		for(int n = 0; n < NACC; n++){ accD2[n] = 0; accI1[n] = 0; accI2[n] = 0; accR2[n] = 0; accO2[n] = 0; }
#pragma unroll
		for(int n = 0; n < NACC; n++){ accM[n] = 0; accD[n] = 0; accR[n] = 0; accO[n] = 0; }
		uint32_t tmpU0 = 0, tmpNE0 = 0, tmpNQ0 = 0, hfirst = 0;
		uint32_t v = 0, vmid = 0;
#pragma unroll
		for(int k = 0; k < W; k++){
			const uint32_t uk = U[k];
			uint32_t h = x_max(m[k], f);
			uint32_t nq = 0;
			if constexpr (PW == 2){
				const uint32_t t = x_add(h, GQQ);                  // max(m, f) + gapo2
				h = x_max(h, g2);
				accI1[k >> 3] = x_acc(accI1[k >> 3], x_minu(x_sub(h, f), ONE), TWO);
				accI2[k >> 3] = x_acc(accI2[k >> 3], x_minu(x_sub(h, g2), ONE), TWO);
				const uint32_t gm = x_max(g2, t);
				g2 = x_sub(gm, Ud[k]);
				accR2[k >> 3] = x_acc(accR2[k >> 3], x_minu(x_sub(gm, t), ONE), TWO);
				const uint32_t n2 = x_sub(h, qq[k]);
				accD2[k >> 3] = x_acc(accD2[k >> 3], x_minu(n2, ONE), TWO);
				const uint32_t nqd = x_minu(n2, NGQQ);
				accO2[k >> 3] = x_acc(accO2[k >> 3], x_satsubu(nqd, NGQQ1), TWO);
				nq = x_sub(nqd, DPQ);
			}
			const uint32_t fm = x_max(f, mg[k]);
			f = x_sub(fm, uk);
			if constexpr (PW != 0) accR[k >> 3] = x_acc(accR[k >> 3], x_minu(x_sub(fm, mg[k]), ONE), TWO);
			const uint32_t n = x_sub(h, ee[k]);
			const uint32_t ne = (PW == 0) ? 0u : x_minu(n, NGOQ);
			if constexpr (DO2){
				const uint32_t kk = ((k & 3) == 0) ? KDO0 : ((k & 3) == 1) ? KDO1 : KDO2;
				if((k & 3) == 3) accDO[k >> 2] = x_add(accDO[k >> 2], ne);
				else accDO[k >> 2] = x_acc(ne, accDO[k >> 2], kk);                 // ne * K + acc
			} else {
				accD[k >> 3] = x_acc(accD[k >> 3], x_minu(n, ONE), TWO);
				if constexpr (PW != 0) accO[k >> 3] = x_acc(accO[k >> 3], x_satsubu(ne, NGOQ1), TWO);
			}
			accM[k >> 3] = x_acc(accM[k >> 3], x_minu(x_sub(h, S[k]), ONE), TWO);
			const uint32_t un = x_sub(h, v);
			v = x_sub(h, uk);
			if(CR == 2 && k == W / 2 - 1) vmid = v;
			if(k == 0){ tmpU0 = un; tmpNE0 = ne; tmpNQ0 = nq; hfirst = h; }
			else if constexpr (STATIC){ U[k] = un; NE[k] = ne; if constexpr (PW == 2) NQ2[k] = nq; }       // in place (U[k] was read above)
			else { U[k - 1] = un; NE[k - 1] = ne; if constexpr (PW == 2) NQ2[k - 1] = nq; }
		}
		// ---- tail (bsalign.h:2618-2636): u of every block's first cell, ubegs of the new row, ubegs[0] re-based on cell 0
		const uint32_t vlast = v;
		tmpU0 = x_sub(tmpU0, x_shift_down<L>(vlast, NGEQ, first));
		const uint32_t bc0 = x_bcast_first<L>(tmpU0);
		{
			const uint32_t D0 = __builtin_amdgcn_perm(bc0, bc0, 0x01000100u);
			HB += x_lo8(bc0) + GE;
			PN = x_add(PN, x_ashr8(x_sub(vlast, D0)));
			if constexpr (CR == 2) PM = x_add(PM, x_ashr8(x_sub(vmid, D0)));
		}
		if(first) tmpU0 = (tmpU0 & 0xffff0000u) | (NGEQ & 0xffffu);
		const uint32_t Psh = x_shift_down<L>(PN, 0u, first);           // ubegs[b] - ubegs[0] - b W gape of the block's own start
		// ---- flags of special cells, then the code row (bsa_common.h "COMPACT slot"); M, D, R were accumulated inverted
		if(__builtin_expect((__builtin_amdgcn_ballot_w64(rbeg == 0u) & actm) != 0ull, 0)){
			if(first && rbeg == 0u){
				const uint32_t hl = hfirst & 0xffffu;
				const uint32_t b0 = 1u << TOPBIT;
				accM[0] = (accM[0] & ~b0) | ((hl == q0m) ? 0u : b0);
				if constexpr (DO2){
					// band position 0 of a row whose band starts at query column 0: D follows the quirk's comparison and need not agree with the
					// field, so this one cell carries D and Od literally (bit 14: D, bit 15: Od); the traceback knows it by x == 0 == band offset
					const uint32_t fld = (accDO[0] >> 14) & 3u;
					accDO[0] = (accDO[0] & ~0xC000u) | ((hl == q0d) ? 0x4000u : 0u) | ((fld == (uint32_t)(-GO)) ? 0x8000u : 0u);
				} else
				accD[0] = (accD[0] & ~b0) | ((hl == q0d) ? 0u : b0);
				if constexpr (PW == 2) accD2[0] = (accD2[0] & ~b0) | ((hl == q0d2) ? 0u : b0);
			}
		}
		if(__builtin_expect((__builtin_amdgcn_ballot_w64(mov > 1u) & actm) != 0ull, 0)){          // (the general form covers mov == 1 of the other pairs of the wave)
			// cells at / beyond the end of the previous row's band: x == bw decides M or I only, x > bw is always I
#pragma unroll
			for(int hf = 0; hf < 2; hf++){
				const int lim = BW - (int)mov - (jl + L * hf) * W;                 // cells k < lim have x < bw
				const int nd = min(max(lim, 0), W), nm = min(max(lim + 1, 0), W);
#pragma unroll
				for(int n = 0; n < NACC; n++){
					constexpr int CN = (W < 8) ? W : 8;
					// accumulator n holds cells 8n .. 8n + CN - 1, cell c at bit 8 + CN - 1 - (c - 8n)
					const int cd = min(max(nd - 8 * n, 0), CN), cm = min(max(nm - 8 * n, 0), CN);
					const uint32_t md = (((1u << (CN - cd)) - 1u) << 8) << (16 * hf), mm = (((1u << (CN - cm)) - 1u) << 8) << (16 * hf);
					if(mov != 0u){ if constexpr (!DO2) accD[n] |= md; accM[n] |= mm; if constexpr (PW == 2) accD2[n] |= md; }
				}
				if constexpr (DO2){
					// "no deletion here" = a zero field becomes the spare value: cells nd .. W - 1 of the half
#pragma unroll
					for(int n = 0; n < NDO; n++){
						const int c0 = min(max(nd - 4 * n, 0), 4);                    // the accumulator's cells c0 .. 3 (bits 15 - 2c, 14 - 2c)
						const uint32_t cm = (0x5500u & ((1u << (16 - 2 * c0)) - 1u)) << (16 * hf);
						const uint32_t z = ~(accDO[n] | (accDO[n] >> 1)) & cm;       // low bit of every zero field among them
						if(mov != 0u) accDO[n] |= z * DOSP;
					}
				}
			}
		} else if constexpr (DO2){
			const uint32_t t = accDO[NDO - 1];
			if(mov == 1u && last && (t & 0x03000000u) == 0u) accDO[NDO - 1] = t | (DOSP << 24);
		} else { accD[NACC - 1] |= (mov == 1u) ? kd1 : 0u; if constexpr (PW == 2) accD2[NACC - 1] |= (mov == 1u) ? kd1 : 0u; }
		if constexpr (PW == 0){
			// linear gaps: every gap is opened at length 1 (R and Od always set); accR is kept inverted, accO is not
#pragma unroll
			for(int n = 0; n < NACC; n++){ accR[n] = 0u; accO[n] = (W >= 8) ? 0xFF00FF00u : (((1u << W) - 1u) << 8) * 0x00010001u; }
		}
		// The code dwords of this row: ND per lane.  Rows are stored in groups of four (bsa_common.h): the lane keeps the
		// dwords of the group's rows in registers and stores whole 16-byte pieces when the group (or the pair) ends.
		{
			uint32_t cur[ND];
			if constexpr (PW == 2){
				// eight facts a cell (bsa_common.h): planes A, D, D2, B | R1, R2, Od1, Od2 of a reference block, W_R bits each.
				// M, D, D2, I1, I2, R1, R2 were accumulated inverted; the decision facts fold into A = M or (no D, no D2, I1 and I2),
				// B = not M and I1 (bit-wise on the planes, after the special cells were set above)
				const uint32_t F8 = 0xFF00FF00u;
				uint32_t pl[8][NACC];                 // the eight planes as they are stored (not inverted), cells in bits 15 .. 8 of each half
#pragma unroll
				for(int n = 0; n < NACC; n++){
					const uint32_t nA = accM[n] & (accI1[n] | accI2[n] | (~(accD[n] & accD2[n]) & F8));
					const uint32_t nB = (~accM[n] & F8) | accI1[n];
					pl[0][n] = ~nA & F8; pl[1][n] = ~accD[n] & F8; pl[2][n] = ~accD2[n] & F8; pl[3][n] = ~nB & F8;
					pl[4][n] = ~accR[n] & F8; pl[5][n] = ~accR2[n] & F8; pl[6][n] = accO[n] & F8; pl[7][n] = accO2[n] & F8;
				}
				if constexpr (WR == 8){
					// two dwords per reference block: A | D << 8 | D2 << 16 | B << 24 and R1 | R2 << 8 | Od1 << 16 | Od2 << 24; a half holds NACC blocks
#pragma unroll
					for(int n = 0; n < NACC; n++){
						const uint32_t t1 = __builtin_amdgcn_perm(pl[1][n], pl[0][n], 0x07030501u);        // {A.lo, D.lo, A.hi, D.hi}
						const uint32_t t2 = __builtin_amdgcn_perm(pl[3][n], pl[2][n], 0x07030501u);        // {D2.lo, B.lo, D2.hi, B.hi}
						const uint32_t t3 = __builtin_amdgcn_perm(pl[5][n], pl[4][n], 0x07030501u);
						const uint32_t t4 = __builtin_amdgcn_perm(pl[7][n], pl[6][n], 0x07030501u);
						cur[2 * n] = __builtin_amdgcn_perm(t2, t1, 0x05040100u); cur[2 * n + 1] = __builtin_amdgcn_perm(t4, t3, 0x05040100u);                         // block NACC jl + n
						cur[2 * (NACC + n)] = __builtin_amdgcn_perm(t2, t1, 0x07060302u); cur[2 * (NACC + n) + 1] = __builtin_amdgcn_perm(t4, t3, 0x07060302u);     // block NACC (jl + L) + n
					}
				} else if constexpr (WR == 4){
					// bandwidth 64: a half holds two reference blocks of four cells (bits 15 .. 12 and 11 .. 8); one dword a block, plane j at bits 4 j
					uint32_t pa = 0, pb = 0;              // per half 16 bits: planes 0 .. 3 of block a / b; then planes 4 .. 7
					uint32_t qa = 0, qb = 0;
#pragma unroll
					for(int j = 0; j < 4; j++){
						pa |= ((pl[j][0] >> 12) & 0x000F000Fu) << (4 * j); pb |= ((pl[j][0] >> 8) & 0x000F000Fu) << (4 * j);
						qa |= ((pl[4 + j][0] >> 12) & 0x000F000Fu) << (4 * j); qb |= ((pl[4 + j][0] >> 8) & 0x000F000Fu) << (4 * j);
					}
					cur[0] = (pa & 0xFFFFu) | (qa << 16); cur[1] = (pb & 0xFFFFu) | (qb << 16);                 // blocks 2 jl, 2 jl + 1
					cur[2] = (pa >> 16) | (qa & 0xFFFF0000u); cur[3] = (pb >> 16) | (qb & 0xFFFF0000u);       // blocks 2 (jl + L), 2 (jl + L) + 1
				} else {
					// bandwidth 256: a half is one reference block of sixteen cells; four dwords a block, plane j in the halfword j of them
					uint32_t x16[8];
#pragma unroll
					for(int j = 0; j < 8; j++) x16[j] = __builtin_amdgcn_perm(pl[j][0], pl[j][NACC - 1], 0x07030501u);      // {plane of the low block, plane of the high block}
#pragma unroll
					for(int d = 0; d < 4; d++){
						cur[d] = __builtin_amdgcn_perm(x16[2 * d + 1], x16[2 * d], 0x05040100u);         // block jl
						cur[4 + d] = __builtin_amdgcn_perm(x16[2 * d + 1], x16[2 * d], 0x07060302u);     // block jl + L
					}
				}
			} else if constexpr (DO2){
				// one dword per reference block: M | R << 8 | two-bit fields << 16 (cell c at bits 15 - 2c, 14 - 2c of those)
#pragma unroll
				for(int n = 0; n < NACC; n++){
					const uint32_t t1 = __builtin_amdgcn_perm(accR[n], accM[n], 0x07030501u);               // {M.lo, R.lo, M.hi, R.hi}
					const uint32_t t2 = __builtin_amdgcn_perm(accDO[2 * n], accDO[2 * n + 1], 0x07030501u);  // {cells 4-7 .lo, cells 0-3 .lo, 4-7 .hi, 0-3 .hi}
					cur[n] = __builtin_amdgcn_perm(t2, t1, 0x05040100u) ^ 0x0000FFFFu;            // block NACC jl + n
					cur[NACC + n] = __builtin_amdgcn_perm(t2, t1, 0x07060302u) ^ 0x0000FFFFu;     // block NACC (jl + L) + n
				}
			} else if constexpr (WR == 8){
				// one dword per reference block: M | D << 8 | R << 16 | Od << 24, cell k at bit 7 - k
#pragma unroll
				for(int n = 0; n < NACC; n++){
					const uint32_t t1 = __builtin_amdgcn_perm(accD[n], accM[n], 0x07030501u);   // {M.lo, D.lo, M.hi, D.hi}
					const uint32_t t2 = __builtin_amdgcn_perm(accO[n], accR[n], 0x07030501u);
					cur[n] = __builtin_amdgcn_perm(t2, t1, 0x05040100u) ^ 0x00FFFFFFu;            // block NACC jl + n
					cur[NACC + n] = __builtin_amdgcn_perm(t2, t1, 0x07060302u) ^ 0x00FFFFFFu;     // block NACC (jl + L) + n
				}
			} else if constexpr (WR == 4 && W == 8){
				// four lanes per pair at bandwidth 64: a half holds two reference blocks of four cells, the first in bits 15..12 of the
				// accumulators, the second in bits 11..8; one dword per reference block: M | D << 4 | R << 8 | Od << 12
				const uint32_t pa = (((accM[0] >> 12) & 0x000F000Fu) | ((accD[0] >> 8) & 0x00F000F0u) | ((accR[0] >> 4) & 0x0F000F00u) | (accO[0] & 0xF000F000u)) ^ 0x0FFF0FFFu;
				const uint32_t pb = (((accM[0] >> 8) & 0x000F000Fu) | ((accD[0] >> 4) & 0x00F000F0u) | (accR[0] & 0x0F000F00u) | ((accO[0] << 4) & 0xF000F000u)) ^ 0x0FFF0FFFu;
				cur[0] = pa & 0xFFFFu; cur[1] = pb & 0xFFFFu; cur[2] = pa >> 16; cur[3] = pb >> 16;      // blocks 2 jl, 2 jl + 1, 2 (jl + L), 2 (jl + L) + 1
			} else if constexpr (WR == 4){
				const uint32_t pk = ((accM[0] >> 8) | (accD[0] >> 4) | accR[0] | (accO[0] << 4)) ^ 0x0FFF0FFFu;
				cur[0] = pk & 0xFFFFu; cur[1] = pk >> 16;
			} else {
				static_assert(WR == 16 || WR == 8 || WR == 4, "code row layouts");
				// 16 cells per block: dword 0 = M | D << 16, dword 1 = R | Od << 16, cell k at bit 15 - k
				const uint32_t xm = __builtin_amdgcn_perm(accM[0], accM[NACC - 1], 0x07030501u) ^ 0xFFFFFFFFu;   // {M16 of the low block, M16 of the high block}
				const uint32_t xd = __builtin_amdgcn_perm(accD[0], accD[NACC - 1], 0x07030501u) ^ 0xFFFFFFFFu;
				const uint32_t xr = __builtin_amdgcn_perm(accR[0], accR[NACC - 1], 0x07030501u) ^ 0xFFFFFFFFu;
				const uint32_t xo = __builtin_amdgcn_perm(accO[0], accO[NACC - 1], 0x07030501u);
				cur[0] = __builtin_amdgcn_perm(xd, xm, 0x05040100u); cur[1] = __builtin_amdgcn_perm(xo, xr, 0x05040100u);     // block jl
				cur[2] = __builtin_amdgcn_perm(xd, xm, 0x07060302u); cur[3] = __builtin_amdgcn_perm(xo, xr, 0x07060302u);     // block jl + L
			}
			const uint32_t ri = i & 3u;                       // (uniform over the wave: all pairs are at row i)
			if constexpr (CWD == 1){
				uint32_t *const sr = stg + 64u * ri;
#pragma unroll
				for(int q = 0; q < ND; q++) sr[256 * q] = cur[q];
			} else {
				uint32_t *const sr = stg + (64u * ND) * ri;
#pragma unroll
				for(int q = 0; q < ND; q++) sr[64 * q] = cur[q];
			}
			if(act && (ri == 3u || i + 1u == tlen)){
				uint32_t *gp = (uint32_t*)rowp + bsa_code_off(i & ~3u, 0u, CWD);       // block y of this group: gp[4 CWD y], rows then dwords
				if constexpr (CWD == 1){
#pragma unroll
					for(int q = 0; q < ND; q++){
						constexpr int RBH = (WR == 8) ? NACC : (WR == 4 && W == 8) ? 2 : 0;      // reference blocks (= code dwords) per half, where a half holds whole ones
						const uint32_t blk = RBH ? (uint32_t)(RBH * (jl + L * (q / RBH)) + q % RBH) : (uint32_t)(jl + L * q);
						uint4 t; t.x = stg[256 * q]; t.y = stg[256 * q + 64]; t.z = stg[256 * q + 128]; t.w = stg[256 * q + 192];
						*(uint4*)(gp + 4u * blk) = t;
					}
				} else {
					// a block's four rows of CWD dwords are contiguous (row r at dword r CWD): 16-byte pieces of two rows (CWD == 2) or one (CWD == 4);
					// a half holds RBH whole reference blocks, the lane's dwords of block (hf, n) are cur[CWD (RBH hf + n) ..]
					constexpr int RBH = (WR == 8) ? NACC : 1;
#pragma unroll
					for(int hb = 0; hb < 2 * RBH; hb++){
						const int hf = hb / RBH, n = hb % RBH;
						const uint32_t blk = (uint32_t)(RBH * (jl + L * hf) + n);
						uint32_t *bp = gp + (4u * CWD) * blk;
#pragma unroll
						for(int pc = 0; pc < CWD; pc++){           // piece pc: dwords 4 pc .. 4 pc + 3 of the block's 4 CWD
							uint4 t;
							uint32_t *tw = (uint32_t*)&t;
#pragma unroll
							for(int e = 0; e < 4; e++){
								const int lin = 4 * pc + e, row = lin / CWD, d = lin % CWD;
								tw[e] = stg[64 * (ND * row + CWD * hb + d)];
							}
							*(uint4*)(bp + 4 * pc) = t;
						}
					}
				}
			}
		}
		if(act){
			// band offsets: lane (i mod L) keeps the offset of row i, the group stores them together
			constexpr uint32_t LM = (uint32_t)(L - 1);
			const bool lastrow = i + 1u == tlen;
			if constexpr (BQ16){
				// Four rows of offsets are 16 bytes at an odd dword: stored as they come, every piece is a partial sector that the L2 has written back long
				// before its neighbours arrive (48 us later).  The lane keeps its last four offsets in LDS slots of its own (slot (row >> 2) & 3) and the
				// pair stores begs[16 k .. 16 k + 15] -- rows 16 k - 1 .. 16 k + 14, one whole line -- when row 16 k + 14 is done; what is pending at the end of
				// the pair or of the segment leaves row by row.
				if((i & 3u) == (uint32_t)jl) bqp[64u * ((i >> 2) & 3u)] = rbeg;
				if((i & 15u) == 14u || lastrow || i + 1u == row1){
					const uint32_t pf = (i >= 15u) ? (((i - 15u) & ~15u) + 15u) : 0u;          // first row no earlier flush has taken
					const int lo = (int)max(row0, pf);
#pragma unroll
					for(int sl = 0; sl < 4; sl++){
						const int row = (int)i - (int)((i - (uint32_t)(4 * sl + jl)) & 15u);
						if(row >= lo) begs[row + 1] = (int)bqp[64 * sl];
					}
				}
			} else {
			if((i & LM) == (uint32_t)jl) begq = (int)rbeg;
			if(((i & LM) == LM || lastrow) && (uint32_t)jl <= (i & LM)) begs[(i & ~LM) + 1u + (uint32_t)jl] = begq;
			}
			// H at band position pos of the new row: ubegs of its block + the block's u up to it (getscore, bsalign.h:3187-3197);
			// meaningful in the lane that owns the block
			auto score_at = [&](uint32_t pos) -> int {
				const uint32_t b = pos / W, kk = pos % W;
				const bool hi = b >= (uint32_t)L;
				int sc = HB + (hi ? x_hi16(Psh) : x_lo16(Psh)) + (int)b * W * GE;
#pragma unroll
				for(int k = 0; k < W; k++){
					const uint32_t uu = (k == 0) ? tmpU0 : U[STATIC ? k : k - 1];
					sc += ((uint32_t)k <= kk) ? ((hi ? x_hi8(uu) : x_lo8(uu)) + GE) : 0;
				}
				return sc;
			};
			if(mode != BSA_MODE_GLOBAL && rbeg + BW >= qlen){
				// overlap / extend: while the band touches the query end, H at query column qlen - 1 is a candidate end
				// (bsalign.h:4023-4032); the lane that owns that cell keeps the best one it has seen (strictly greater wins)
				const uint32_t pos = qlen - 1u - rbeg;
				if(((pos / W) & (uint32_t)(L - 1)) == (uint32_t)jl){
					const int sc = score_at(pos);
					if(sc > cand_sc){ cand_sc = sc; cand_te = (int)i; }
				}
			}
			if(lastrow && mode == BSA_MODE_GLOBAL){
				// global score = H at query column qlen - 1 of the last row (bsalign.h:4034-4037), kept in begs[tlen + 1]
				const uint32_t pos = qlen - 1u - rbeg;
				if(pos >= (uint32_t)BW){ if(first) begs[tlen + 1u] = (int)0x80000000u; }        // band never reached the query end
				else if(((pos / W) & (uint32_t)(L - 1)) == (uint32_t)jl) begs[tlen + 1u] = score_at(pos);
			} else if(lastrow){
				// end record (bsa_common.h bsa_code_end_t): the candidates and the last row itself, per reference block, in natural
				// band order (row_max is taken by the traceback kernel)
				bsa_code_end_t *er = (bsa_code_end_t*)(rowp + (size_t)bsa_code_rows(tlen) * (64u * CWD));
#pragma unroll
				for(int q = 0; q < 16 / L; q++){ er->cand_sc[jl + L * q] = q ? BSA_SCORE_MIN : cand_sc; er->cand_te[jl + L * q] = q ? 0 : cand_te; }
				int8_t *ub = (int8_t*)(er + 1);
#pragma unroll
				for(int hf = 0; hf < 2; hf++){
					const int b = jl + L * hf;
					er->ubegs[b * CR] = HB + (hf ? x_hi16(Psh) : x_lo16(Psh)) + b * W * GE;
					if constexpr (CR == 2) er->ubegs[b * CR + 1] = HB + (hf ? x_hi16(PM) : x_lo16(PM)) + (b * W + W / 2) * GE;
#pragma unroll
					for(int k = 0; k < W; k++){
						const uint32_t uu = (k == 0) ? tmpU0 : U[STATIC ? k : k - 1];
						ub[b * W + k] = (int8_t)((hf ? x_hi8(uu) : x_lo8(uu)) + GE);
					}
				}
				if(last){ er->ubegs[16] = HB + x_hi16(PN) + BW * GE; er->rbeg_last = (int)rbeg; }
			}
		}
		// ---- adaptive band (bsalign.h:3331-3349) + global steering (bsalign.h:4006-4021)
		if constexpr (!STATIC){
			uint32_t x;
			{
				if constexpr (CR == 1){
					const uint32_t dl = x_add(x_sub(PN, Psh), WGE16);         // ubegs[b+1] - ubegs[b]
					x = x_max(dl, x_sub(0u, dl));
				} else {
					const uint32_t HGE16 = x_i16((W / 2) * GE);
					const uint32_t da = x_add(x_sub(PM, Psh), HGE16), db = x_add(x_sub(PN, PM), HGE16);     // the two reference blocks of this block
					x = x_add(x_max(da, x_sub(0u, da)), x_max(db, x_sub(0u, db)));
				}
				x = x_sum<L>(x);
			}
			const int nzsum = (int)((x & 0xffffu) + (x >> 16));
			const int d16 = x_hi16(x_bcast_last<L>(PN)) + BW * GE;             // ubegs[16] - ubegs[0]
			uint32_t nz = (uint32_t)(nzsum / 16);
			nz = nz / (uint32_t)WR * 16u / 2u;
			const int noisy = (int)((16u > nz) ? 16u : nz);
			int rbx;
			if(i <= (uint32_t)BW / 4u) rbx = 0;
			else if(rbeg + BW >= qlen) rbx = 0;
			else if(noisy < d16) rbx = 2;
			else if(d16 < -noisy) rbx = 0;
			else rbx = 1;
			if(mode == BSA_MODE_GLOBAL){
				const int rby = __builtin_amdgcn_ds_bpermute(rby_lane + (int)((i & (uint32_t)(L - 1)) << 2), rby_tab);
				const uint32_t left = tlen - i - 1u;
				bool rush;
				if(rush32) rush = act && rbeg + rzl + (uint32_t)BW <= qlen + (uint32_t)rbz - 1u;
				else {
					const unsigned long long lhs = (unsigned long long)rbeg + (unsigned long long)(uint32_t)rbz * left + (unsigned long long)BW;
					rush = act && lhs <= (unsigned long long)(uint32_t)(qlen + (uint32_t)rbz - 1u);
				}
				if((int)rbeg < rby - BW) mov = (uint32_t)(rbx + 1);
				else if((int)rbeg > rby) mov = (uint32_t)max(0, rbx - 1);
				else mov = (uint32_t)rbx;
				if(rush) mov = 1u + (uint32_t)(qlen - (rbeg + BW)) / max(left, 1u);
			} else mov = (uint32_t)rbx;
		}
		if constexpr (STATIC){ U[0] = tmpU0; NE[0] = tmpNE0; if constexpr (PW == 2) NQ2[0] = tmpNQ0; }
		// ---- speculative slide by one cell: the first cell of every block becomes the last cell of the block before it
		else {
			uint32_t inu;
			if constexpr (L == 4) inu = x_shift_up<L>(tmpU0, NEWU0, last);
			else {
				const uint32_t nxt = XDPPZ(tmpU0, XROW_SHL(1), 0xf);          // (a DPP move must not sit in an arm of ?: -- only one arm runs)
				inu = last ? __builtin_amdgcn_alignbit(NEWU0, bc0, 16) : nxt;
			}
			const uint32_t inne = x_shift_up<L>(tmpNE0, NEWNE, last);
			U[W - 1] = inu; NE[W - 1] = inne;
			if constexpr (PW == 2){ NQ2[W - 1] = x_shift_up<L>(tmpNQ0, NEWNE, last); svNQ = tmpNQ0; }
			PN = x_add(x_add(PN, x_ashr8(inu)), GE16);
			if constexpr (CR == 2) PM = x_add(x_add(PM, x_ashr8(U[W / 2 - 1])), GE16);
			svU = tmpU0; svNE = tmpNE0;
		}
		i++;
		rzl -= (uint32_t)rbz;          // (meaningless once the pair has no row left: only read under `act`)
		// (the next eight target bases.  The compiler waits for them at the top of the next row, with `s_waitcnt vmcnt(0)` on every row; forcing the wait
		// into this branch instead -- once per eight rows, fully exposed -- was measured slower: 60.0 against 59.8 ms, two-piece gaps 129.8 against 127.1)
		if((i & 7u) == 0u && i < tlen){ __builtin_memcpy(&twin, tp + i, 8); twin <<= 2; }
	}
	if(st != nullptr && row1 < tlen){
		// write-through stores (sc1): the next segment usually runs on another CU, often on another XCD, and a release fence here
		// would write back every dirty line of this XCD's L2 -- the half-filled lines of the code rows of 400 waves -- once per item
		uint32_t *sp = st + (lt & 63);
		int r = 0;
		auto put = [&](uint32_t v){ __hip_atomic_store(sp + 64 * r, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); r++; };
#pragma unroll
		for(int k = 0; k < W; k++) put(U[k]);
#pragma unroll
		for(int k = 0; k < W; k++) put(NE[k]);
		if constexpr (PW == 2){
#pragma unroll
			for(int k = 0; k < W; k++) put(NQ2[k]);
		}
		put(PN); put(PM); put((uint32_t)HB); put(svU); put(svNE); put(svNQ);
		put(rbeg); put(mov); put((uint32_t)cand_sc); put((uint32_t)cand_te);
	}
}

template<int W, int L, bool DO2 = false>
__global__ void __launch_bounds__(256) k_align8_fwd_x(const Align8Args a){
	x_forward<W, L, 1, false, 4, DO2>(a, a.first, a.count, blockIdx.x * 256u);
}
// linear gaps (piecewise 0)
template<int W, int L>
__global__ void __launch_bounds__(256) k_align8_fwd_x0(const Align8Args a){
	x_forward<W, L, 0>(a, a.first, a.count, blockIdx.x * 256u);
}
// two-piece gaps (bandwidth 128): 8 bits per band cell
__global__ void __launch_bounds__(256) k_align8_fwd_x2(const Align8Args a){
	x_forward<8, 8, 2>(a, a.first, a.count, blockIdx.x * 256u);
}
// ... at bandwidth 64 (four lanes per pair) and 256 (sixteen cells a half: 256 registers, two waves per SIMD)
template<int W, int L>
__global__ void __launch_bounds__(256) k_align8_fwd_x2w(const Align8Args a){
	x_forward<W, L, 2>(a, a.first, a.count, blockIdx.x * 256u);
}
// bands that cover their whole queries (Align8Args::static_band): the row stays in place, no steering
template<int W, int L, int PW, bool DO2 = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) k_align8_fwd_x_static(const Align8Args a){
	x_forward<W, L, PW, true, 4, DO2>(a, a.first, a.count, blockIdx.x * 256u);
}

// Bandwidth 128, a batch that is not a whole number of four-lane rounds: the first nb8 blocks take the last n8 pairs
// eight lanes per pair, the others the first a.count - n8 pairs four lanes per pair.  One launch: the short blocks start
// first and the dispatcher hands the long ones to whichever CU has room, so pairs of one length no longer finish in
// lock-step rounds with a nearly empty last one.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) k_align8_fwd_x_mix(const Align8Args a, const uint32_t nb8, const uint32_t n8){
	// one staging area and one query window for both shapes (four waves; <16, 4> is the larger: 16 code dwords and 2 x 16 window dwords a lane)
	__shared__ uint32_t mix_stage[4 * 16 * 64], mix_qwin[4 * 32 * 64];
	if(blockIdx.x < nb8) x_forward<8, 8, 1, false, 4, false, true>(a, a.first + (a.count - n8), n8, blockIdx.x * 256u, 0u, 0xFFFFFFF8u, nullptr, mix_stage, mix_qwin);
	else x_forward<16, 4, 1, false, 4, false, true>(a, a.first, a.count - n8, (blockIdx.x - nb8) * 256u, 0u, 0xFFFFFFF8u, nullptr, mix_stage, mix_qwin);
}

// PERSISTENT form for batches of more than one generation of resident waves.  A pair's rows are serial and pairs of one length finish
// together, so a launch of whole pairs ends with a generation that is mostly empty (100 000 pairs = 6250 waves of 16 pairs on 3072
// wave slots: two full generations and 106 waves alone on the chip for a third).  Here the work item is a SEGMENT of rows of the 16
// pairs of a wave: every wave of the launch takes ONE item (segment s of group g, all groups' segment s before any segment s + 1) off a
// counter, takes the band state the previous segment left in memory (XS_WORDS dwords a lane: the two row planes, the block offsets,
// the band position and its pending move), runs the rows and hands the state on.  The launch then ends within one segment of the
// ideal.  ctl[0] = next item, ctl[16 + g] = segments of group g that are done (release / acquire at agent scope: the next segment
// usually runs on another CU).  An item only ever waits for an item that was handed out before it, i.e. one that is running.
struct XQArgs { uint32_t *ctl; uint32_t *state; uint32_t ngroups, nseg, seg_rows, spin_cap; };          // ctl[0]: next ticket, ctl[1]: some wave gave up waiting, ctl[16 + g]: segments of group g done
template<int W, int L, int PW, bool DO2 = false, int WPS = 3>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPS))) k_align8_fwd_xq(const Align8Args a, const XQArgs q){
	// one item per wave (a block is a wave: the dispatcher refills a wave slot the moment it is free); the ticket, not the block
	// index, names the item, so that an item's predecessor is always one that has started
	uint32_t id = 0;
	if(threadIdx.x == 0u) id = atomicAdd(q.ctl, 1u);
	id = (uint32_t)__builtin_amdgcn_readfirstlane((int)id);
	const uint32_t s = id / q.ngroups, g = id - s * q.ngroups;
	uint32_t *done = q.ctl + 16u + g;
	if(s != 0u){
		// BOUNDED wait (an item's predecessor was handed out earlier, i.e. is running: in a healthy launch this loop hardly ever turns).  Should the
		// predecessor never arrive -- a fault in its wave, a future change of the ticket order -- the wave gives up after about two seconds
		// (2^22 turns of s_sleep 16 = 1024 cycles), raises q.ctl[1] and flags its pairs BSA_ST_DEVICE instead of hanging the device: flagged pairs
		// are skipped by every later segment and by the traceback (zeroed result).  bsa_align_batch scans the status words and returns BSA_E_HIP;
		// a caller of the device-pointer form (bsa_align_run is asynchronous) finds the flag in its status array (include/bsalign_hip.h).
		// A wave waits for ITS predecessor only: another group's give-up does not end the wait of a healthy one.
		uint32_t gaveup = 0;
		if(threadIdx.x == 0u){
			uint32_t turns = 0;
			while(__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < s){
				__builtin_amdgcn_s_sleep(16);
				if(++turns >= q.spin_cap){ gaveup = 1; break; }
			}
		}
		gaveup = (uint32_t)__builtin_amdgcn_readfirstlane((int)gaveup);
		if(gaveup){
			const uint32_t pg = (g * 64u + threadIdx.x) / (uint32_t)L;
			if((threadIdx.x & (uint32_t)(L - 1)) == 0u && pg < a.count) atomicOr(&a.status[a.order[a.first + pg]], BSA_ST_DEVICE);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // the flags are visible before the successor (which acquires after its wait) is let go
			if(threadIdx.x == 0u){
				atomicOr(q.ctl + 1, 1u);
				__hip_atomic_fetch_max(done, s + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);          // successors go on (and skip the flagged pairs)
			}
			return;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
	}
	x_forward<W, L, PW, false, 1, DO2>(a, a.first, a.count, g * 64u, s * q.seg_rows, (s + 1u) * q.seg_rows, q.state + (size_t)g * (XS_WORDS(W, PW) * 64u));
	if(s + 1u < q.nseg){
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the state's write-through stores have arrived
		if(threadIdx.x == 0u) __hip_atomic_fetch_max(done, s + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (max: a successor that gave up may have written s + 2)
	}
}
static int x_cus(){
	static const int cus = [](){
		int dev = 0, v = 0;
		if(hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
		return v;
	}();
	return cus;
}
// bytes of a.xq the persistent form needs for `count` pairs (0: the shape has no persistent form)
size_t bsa_align8_xq_bytes(uint32_t bw, int pw, uint32_t count){
	const uint32_t W = bw / 16u;
	if(pw > 2 || !(W == 4u || W == 8u || W == 16u)) return 0;
	const uint32_t L = (W == 16u || (pw == 2 && W == 8u)) ? 8u : 4u, Wl = (W == 4u || (pw == 2 && W == 8u)) ? 8u : 16u;          // lanes per pair, cells per lane and half
	const size_t groups = ((size_t)count * L + 63u) / 64u;
	return (16u + groups) * 4u + 256u + groups * (size_t)(XS_WORDS(Wl, pw) * 64u * 4u);
}
// true when the launch was made
template<int W, int L, int PW, bool DO2 = false, int WPS = 3>
static bool x_launch_xq(const Align8Args &a, hipStream_t st, hipError_t &err){
	const char *qe = bsa_env("BSA_ALIGN8_XQ");
	if(!a.xq || (qe && qe[0] == '0')) return false;
	const uint32_t groups = (uint32_t)(((size_t)a.count * L + 63u) / 64u);
	const uint32_t workers = (uint32_t)x_cus() * 4u * (uint32_t)WPS;          // waves per SIMD x SIMDs
	if(!(qe && qe[0] == '1') && groups <= workers) return false;       // one generation: whole pairs
	const size_t ctl_bytes = ((16u + (size_t)groups) * 4u + 255u) & ~(size_t)255u;
	if(ctl_bytes + (size_t)groups * (XS_WORDS(W, PW) * 64u * 4u) > a.xq_bytes) return false;
	// segments: the last item of the launch should be a few percent of a wave slot's share
	uint32_t nseg = (uint32_t)std::min<uint64_t>(64u, (48ull * workers + groups - 1u) / groups);
	uint32_t seg_rows = ((a.max_tlen + nseg - 1u) / std::max(nseg, 1u) + 7u) & ~7u;
	if(const char *se = bsa_env("BSA_ALIGN8_XQ_SEG")){ const long v = atol(se); if(v >= 8) seg_rows = ((uint32_t)v + 7u) & ~7u; }
	seg_rows = std::max(seg_rows, 64u);
	nseg = std::max(1u, (a.max_tlen + seg_rows - 1u) / seg_rows);
	XQArgs q;
	q.ctl = a.xq; q.state = (uint32_t*)((uint8_t*)a.xq + ctl_bytes); q.ngroups = groups; q.nseg = nseg; q.seg_rows = seg_rows;
	q.spin_cap = 1u << 22;
	if(const char *sc = bsa_env("BSA_ALIGN8_XQ_SPIN_CAP")){ const long v = atol(sc); if(v >= 1) q.spin_cap = (uint32_t)v; }          // (test hook: a cap of a few turns makes hand-over waits give up)
	err = hipMemsetAsync(a.xq, 0, ctl_bytes, st);
	if(err != hipSuccess) return true;
	hipLaunchKernelGGL((k_align8_fwd_xq<W, L, PW, DO2, WPS>), dim3(groups * nseg), dim3(64), 0, st, a, q);
	err = hipGetLastError();
	bsa_last_fwd_kernel = "k_align8_fwd_xq (exact-arithmetic forward DP in row segments, 4-bit traceback codes)";
	return true;
}

// Exact arithmetic is the reference's arithmetic only while nothing saturates: the guard of the compact path
// (bsa_align8_codes_supported) plus room for the frame shift by 2 |gape| and for the int16 block offsets.
bool bsa_align8_x_supported(const Align8Args &a, int pw){
	const uint32_t W = a.bw / 16;
	const int ge = -(int)(int8_t)a.gape1, go = -(int)(int8_t)a.gapo1, m = a.smax, n = -a.smin;
	if(ge < 0 || go < 0 || m < 0 || n < 0 || (go == 0) != (pw == 0)) return false;
	int g = go + ge;
	if(pw == 2){
		// two pieces: bandwidth 64, 128, 256 (all three modes: the end record of overlap / extend is the one-piece kernels'); piece 2 opens
		// dearer and extends cheaper (bsalign.h:2084-2092 guarantees it), the bound is taken with the dearer opening
		if(!(W == 4 || W == 8 || W == 16)) return false;
		const int ge2 = -(int)(int8_t)a.gape2, go2 = -(int)(int8_t)a.gapo2;
		if(ge2 < 0 || go2 <= go || ge2 >= ge) return false;
		g = std::max(g, go2 + ge2);
		if(m + 3 * g > 64 || n + m + g > 100) return false;      // the compact path's own bound (bsa_align8_codes_supported)
	} else {
		if(pw > 1 || !bsa_align8_codes_supported(a, pw)) return false;
		if(!(W == 4 || W == 8 || W == 16)) return false;
	}
	const int cfirst = std::min(a.smin, -g) - 1 - a.smax - g;
	if(cfirst < -100) return false;
	return m + 3 * g + 2 * ge <= 100 && n + m + g + 2 * ge <= 110 && 63 + 2 * ge + n + m + 2 * g <= 125;
}

// the code rows with two-bit D / Od fields (Align8Args::code_fmt 1): what the forward kernels need; the traceback side is asked in bsa_api.hip
bool bsa_align8_do2_supported(const Align8Args &a, int pw){
	const int go = -(int)(int8_t)a.gapo1;
	const char *e = bsa_env("BSA_ALIGN8_DO2");
	return pw == 1 && a.bw == 128u && go >= 1 && go <= 3 && !(e && e[0] == '0') && !bsa_env("BSA_ALIGN8_X_LANES") && !bsa_env("BSA_ALIGN8_X_N8") && bsa_align8_x_supported(a, pw);
}

static bool x8_at_64(){ const char *e = bsa_env("BSA_ALIGN8_X_LANES"); return e && e[0] == '8'; }
hipError_t bsa_launch_align8_fwd_x(const Align8Args &a, int pw, hipStream_t st){
	if(a.count == 0) return hipSuccess;
	const uint32_t b8 = (a.count + 31u) / 32u;
	if(a.code_fmt == 1u){
		// two-bit D / Od fields (bsa_align8_do2_supported): bandwidth 128, one-piece gaps, four lanes per pair
		if(pw != 1 || a.bw != 128u) return hipErrorInvalidValue;
		hipError_t qe = hipSuccess;
		if(a.static_band && !bsa_env("BSA_ALIGN8_NO_STATIC")){
			hipLaunchKernelGGL((k_align8_fwd_x_static<16, 4, 1, true>), dim3((a.count + 63u) / 64u), dim3(256), 0, st, a);
			bsa_last_fwd_kernel = "k_align8_fwd_x_static (exact-arithmetic forward DP, band in place, traceback codes with two-bit D/Od fields)";
		} else if(x_launch_xq<16, 4, 1, true>(a, st, qe)){
			bsa_last_fwd_kernel = "k_align8_fwd_xq (exact-arithmetic forward DP in row segments, traceback codes with two-bit D/Od fields)";
			return qe;
		} else {
			hipLaunchKernelGGL((k_align8_fwd_x<16, 4, true>), dim3((a.count + 63u) / 64u), dim3(256), 0, st, a);
			bsa_last_fwd_kernel = "k_align8_fwd_x (exact-arithmetic forward DP, traceback codes with two-bit D/Od fields)";
		}
		return hipGetLastError();
	}
	if(a.static_band && !bsa_env("BSA_ALIGN8_NO_STATIC")){
		const uint32_t b4 = (a.count + 63u) / 64u;
		if(pw == 2 && a.bw == 128u){ hipLaunchKernelGGL((k_align8_fwd_x_static<8, 8, 2>), dim3(b8), dim3(256), 0, st, a); return hipGetLastError(); }
		if(pw == 1 && a.bw == 64u){ hipLaunchKernelGGL((k_align8_fwd_x_static<8, 4, 1>), dim3(b4), dim3(256), 0, st, a); return hipGetLastError(); }
		if(pw == 1 && a.bw == 128u){ hipLaunchKernelGGL((k_align8_fwd_x_static<16, 4, 1>), dim3(b4), dim3(256), 0, st, a); return hipGetLastError(); }
		if(pw == 1 && a.bw == 256u){ hipLaunchKernelGGL((k_align8_fwd_x_static<16, 8, 1>), dim3(b8), dim3(256), 0, st, a); return hipGetLastError(); }
		if(pw == 0 && a.bw == 64u){ hipLaunchKernelGGL((k_align8_fwd_x_static<8, 4, 0>), dim3(b4), dim3(256), 0, st, a); return hipGetLastError(); }
		if(pw == 0 && a.bw == 128u){ hipLaunchKernelGGL((k_align8_fwd_x_static<16, 4, 0>), dim3(b4), dim3(256), 0, st, a); return hipGetLastError(); }
		if(pw == 0 && a.bw == 256u){ hipLaunchKernelGGL((k_align8_fwd_x_static<16, 8, 0>), dim3(b8), dim3(256), 0, st, a); return hipGetLastError(); }
	}
	if(pw == 2 && a.bw == 64u){
		hipError_t qe2 = hipSuccess;
		bsa_last_fwd_kernel = "k_align8_fwd_x2 (exact-arithmetic forward DP, two-piece gaps, 8-bit traceback codes)";
		if(x_launch_xq<8, 4, 2>(a, st, qe2)) return qe2;
		hipLaunchKernelGGL((k_align8_fwd_x2w<8, 4>), dim3((a.count + 63u) / 64u), dim3(256), 0, st, a);
		return hipGetLastError();
	}
	if(pw == 2 && a.bw == 256u){
		bsa_last_fwd_kernel = "k_align8_fwd_x2 (exact-arithmetic forward DP, two-piece gaps, 8-bit traceback codes)";
		hipLaunchKernelGGL((k_align8_fwd_x2w<16, 8>), dim3(b8), dim3(256), 0, st, a);
		return hipGetLastError();
	}
	if(pw == 2){
		if(a.bw != 128u) return hipErrorInvalidValue;
		hipError_t qe2 = hipSuccess;
		// row segments: four lanes per pair at two waves per SIMD (sixteen pairs a wave share the per-row work: 126 ms against 137 for the
		// eight-lane shape at three waves, C2's shape); BSA_ALIGN8_X2_LANES=8 keeps the eight-lane shape
		{ const char *le = bsa_env("BSA_ALIGN8_X2_LANES");
		  if(!(le && le[0] == '8') && x_launch_xq<16, 4, 2, false, 2>(a, st, qe2)){ bsa_last_fwd_kernel = "k_align8_fwd_xq (exact-arithmetic forward DP in row segments, two-piece gaps, four lanes per pair, 8-bit traceback codes)"; return qe2; } }
		if(x_launch_xq<8, 8, 2>(a, st, qe2)){ bsa_last_fwd_kernel = "k_align8_fwd_xq (exact-arithmetic forward DP in row segments, two-piece gaps, 8-bit traceback codes)"; return qe2; }
		hipLaunchKernelGGL(k_align8_fwd_x2, dim3(b8), dim3(256), 0, st, a);
		return hipGetLastError();
	}
	hipError_t qerr = hipSuccess;
	if(pw == 0){
		if(a.bw == 64u && !x8_at_64() && x_launch_xq<8, 4, 0>(a, st, qerr)) return qerr;
		if(a.bw == 128u && x_launch_xq<16, 4, 0>(a, st, qerr)) return qerr;
		if(a.bw == 256u && x_launch_xq<16, 8, 0>(a, st, qerr)) return qerr;
		switch(a.bw / 16){
			case 4:
				if(x8_at_64()) hipLaunchKernelGGL((k_align8_fwd_x0<4, 8>), dim3(b8), dim3(256), 0, st, a);
				else hipLaunchKernelGGL((k_align8_fwd_x0<8, 4>), dim3((a.count + 63u) / 64u), dim3(256), 0, st, a);
				break;
			case 8:  hipLaunchKernelGGL((k_align8_fwd_x0<16, 4>), dim3((a.count + 63u) / 64u), dim3(256), 0, st, a); break;
			case 16: hipLaunchKernelGGL((k_align8_fwd_x0<16, 8>), dim3(b8), dim3(256), 0, st, a); break;
			default: return hipErrorInvalidValue;
		}
		return hipGetLastError();
	}
	if(a.bw == 64u && !x8_at_64() && x_launch_xq<8, 4, 1>(a, st, qerr)) return qerr;
	if(a.bw == 128u && !bsa_env("BSA_ALIGN8_X_LANES") && !bsa_env("BSA_ALIGN8_X_N8") && x_launch_xq<16, 4, 1>(a, st, qerr)) return qerr;
	if(a.bw == 256u && x_launch_xq<16, 8, 1>(a, st, qerr)) return qerr;
	switch(a.bw / 16){
		case 4:      // bandwidth 64: four lanes per pair (16 pairs per wave, eight cells per half); BSA_ALIGN8_X_LANES=8: eight lanes
			if(x8_at_64()) hipLaunchKernelGGL((k_align8_fwd_x<4, 8>), dim3(b8), dim3(256), 0, st, a);
			else hipLaunchKernelGGL((k_align8_fwd_x<8, 4>), dim3((a.count + 63u) / 64u), dim3(256), 0, st, a);
			break;
		case 8: {
			// Four lanes per pair (16 pairs per wave) cost 620 instructions per row of a wave, eight lanes per pair 387.
			// Pairs of one length finish together, so what counts is the largest number of waves any SIMD gets: whole
			// rounds of one four-lane wave per SIMD, and a remainder of at most half a round as eight-lane waves (0.62 of
			// a round instead of a whole one).  BSA_ALIGN8_X_LANES=4 / 8 forces one shape.
			static const int cus = [](){
				int dev = 0, v = 0;
				if(hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
				return v;
			}();
			const char *le = bsa_env("BSA_ALIGN8_X_LANES");
			const uint32_t round4 = (uint32_t)cus * 4u * 16u;
			uint32_t n4 = a.count / round4 * round4;
			if(a.count - n4 > round4 / 2u) n4 = a.count;
			if(le && le[0] == '8') n4 = 0;
			if(le && le[0] == '4') n4 = a.count;
			if(const char *ne = bsa_env("BSA_ALIGN8_X_N8")){ const long v = atol(ne); if(v >= 0 && (uint32_t)v <= a.count) n4 = a.count - (uint32_t)v; }      // tuning: pairs that go eight lanes per pair
			if(n4 == a.count) hipLaunchKernelGGL((k_align8_fwd_x<16, 4>), dim3((n4 + 63u) / 64u), dim3(256), 0, st, a);
			else if(n4 == 0) hipLaunchKernelGGL((k_align8_fwd_x<8, 8>), dim3(b8), dim3(256), 0, st, a);
			else {
				const uint32_t n8 = a.count - n4, nb8 = (n8 + 31u) / 32u;
				hipLaunchKernelGGL(k_align8_fwd_x_mix, dim3(nb8 + (n4 + 63u) / 64u), dim3(256), 0, st, a, nb8, n8);
			}
			break;
		}
		case 16: hipLaunchKernelGGL((k_align8_fwd_x<16, 8>), dim3(b8), dim3(256), 0, st, a); break;
		default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}
