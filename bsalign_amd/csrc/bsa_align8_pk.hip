// bsa_align8_pk.hip -- packed variant of the 8-bit banded striped forward DP: TWO pairs per 16-lane DPP row.
//
// Same algorithm, same row records and same bit-exact results as k_align8_fwd (bsa_align8.hip; reference
// functions cited there), but every cell register holds two independent alignments: pair A in bits 15..0 and
// pair B in bits 31..16, each as value << 8 in an int16.  In that form the int16-saturating packed VALU ops
// (v_pk_add_i16 / v_pk_sub_i16 with clamp, v_pk_max_i16) ARE the reference's int8-saturating SSE ops:
//   (a << 8) + (b << 8) saturates at -32768 = (-128) << 8 and at 32767, which one AND with 0xFF00FF00 turns
//   into 127 << 8.  That normalisation is applied exactly where a positive overflow is possible (h - v, h - u,
//   f - u); it is provably unnecessary after adding the non-positive gap constants and for e + u (e <= 0 always).
// A wave64 therefore advances 8 pairs per row step with roughly the instruction count the int32 kernel needs
// for 4.  Per-pair scalars (band offset, ubegs, steering) stay 32-bit and are computed per half.
// Requires gape1, gapo1+gape1 (and gape2, gapo2+gape2 for 2-piece gaps) <= 0; the dispatcher falls back to the
// int32 kernel otherwise.
#include "bsa_common.h"
#include "bsa_dpp.h"

typedef short v2s __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ v2s as_v2s(uint32_t a){ return __builtin_bit_cast(v2s, a); }
static __device__ __forceinline__ uint32_t as_u32(v2s a){ return __builtin_bit_cast(uint32_t, a); }
static __device__ __forceinline__ uint32_t pk_adds(uint32_t a, uint32_t b){ return as_u32(__builtin_elementwise_add_sat(as_v2s(a), as_v2s(b))); }
static __device__ __forceinline__ uint32_t pk_subs(uint32_t a, uint32_t b){ return as_u32(__builtin_elementwise_sub_sat(as_v2s(a), as_v2s(b))); }
static __device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b){ return as_u32(__builtin_elementwise_max(as_v2s(a), as_v2s(b))); }
static __device__ __forceinline__ uint32_t pk_norm(uint32_t a){ return a & 0xFF00FF00u; }
typedef unsigned short v2us_ __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ uint32_t pk_addu(uint32_t a, uint32_t b){ return __builtin_bit_cast(uint32_t, (v2us_)(__builtin_bit_cast(v2us_, a) + __builtin_bit_cast(v2us_, b))); }
static __device__ __forceinline__ int pk_get(uint32_t x, int h){ return h ? ((int)x >> 24) : __builtin_amdgcn_sbfe((int)x, 8, 8); }
static __device__ __forceinline__ uint32_t pk_make(int a, int b){ return (((uint32_t)a & 0xffu) << 8) | ((uint32_t)b << 24); }
static __device__ __forceinline__ uint32_t pk_splat(int x){ return pk_make(x, x); }
static __device__ __forceinline__ uint32_t pk_make16(int a, int b){ return ((uint32_t)a & 0xffffu) | ((uint32_t)b << 16); }   // plain int16 halves
static __device__ __forceinline__ uint32_t pk_dpp_shr1(uint32_t fill, uint32_t x){ return (uint32_t)DPP_SHR((int)fill, (int)x, 1); }
static __device__ __forceinline__ uint32_t pk_dpp_shl1(uint32_t fill, uint32_t x){ return (uint32_t)DPP_SHL((int)fill, (int)x, 1); }

template<int W>
static __device__ __forceinline__ void load_qcodes_pk(const uint8_t *p, uint32_t *w){
	if constexpr (W >= 4) __builtin_memcpy(w, p, W);
	else if constexpr (W == 2){ uint16_t v; __builtin_memcpy(&v, p, 2); w[0] = v | 0x04040000u; }
	else w[0] = p[0] | 0x04040400u;
}

typedef unsigned short v2us __attribute__((ext_vector_type(2)));
// per 16-bit half: 1 where the halves of a and b are equal, else 0 (v_xor + v_pk_sub_u16 clamp)
static __device__ __forceinline__ uint32_t pk_eq(uint32_t a, uint32_t b){
	return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(v2us, 0x00010001u), __builtin_bit_cast(v2us, a ^ b)));
}
// acc = acc * 2 + flag per half: one v_pk_mad_u16 (the compiler splits the C expression into a shift and an or; the
// multiplier comes from a register because a VOP3P inline constant only reaches the low half)
static __device__ __forceinline__ uint32_t pk_acc(uint32_t acc, uint32_t flag){
	uint32_t r;
	asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(acc), "s"(0x00020002u), "v"(flag));
	return r;
}

// F-penetration (bsalign.h:2639-2652) of both pairs at once: the max-plus scan of bsa_dpp.h fpen() on plain int16
// halves.  f comes and goes in the value << 8 form.  Entering sums stay inside int16 for W <= 16 (|ubegs[j+1] -
// ubegs[j]| <= 128 W, 15 lanes); anything below -128 can never win against an int8 f, so negative saturation is
// harmless, and the only way the scan can differ from the reference's serial chain -- an entering value above 127
// being truncated to int8 -- sends the row through the literal chain, exactly as in fpen().
static __device__ __forceinline__ uint32_t fpen_pk(uint32_t f, const int (&ubA)[2], const int (&ubB)[2], int t, int j){
	const uint32_t NEG = 0x80008000u;
	uint32_t f16;
	asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(f16) : "s"(0x00080008u), "v"(f));
	const uint32_t fs = (uint32_t)DPP_SHR((int)pk_make16(BSA_EPI8_MIN, BSA_EPI8_MIN), (int)f16, 1);
	const uint32_t c = pk_make16(t - (ubB[0] - ubA[0]), t - (ubB[1] - ubA[1]));
	uint32_t A = (uint32_t)DPP_SHR((int)NEG, (int)c, 1);
	uint32_t B = fs;
#define FPK_STEP(n) { const uint32_t A1 = (uint32_t)DPP_SHR(0, (int)A, n); const uint32_t B1 = (uint32_t)DPP_SHR((int)NEG, (int)B, n); B = pk_max(pk_adds(B1, A), B); A = pk_adds(A1, A); }
	FPK_STEP(1) FPK_STEP(2) FPK_STEP(4) FPK_STEP(8)
#undef FPK_STEP
	const uint32_t sprev = (uint32_t)DPP_SHR((int)NEG, (int)pk_adds(B, c), 1);
	const uint32_t lim = pk_make16(127, 127);
	if(__any(pk_max(sprev, lim) != lim)){
		const int fa = fpen_serial(__builtin_amdgcn_sbfe((int)f16, 0, 16), ubA[0], ubB[0], t, j);
		const int fb = fpen_serial((int)f16 >> 16, ubA[1], ubB[1], t, j);
		return pk_make(fa, fb);
	}
	uint32_t r;
	asm("v_pk_lshlrev_b16 %0, %1, %2" : "=v"(r) : "s"(0x00080008u), "v"(B));
	return r;
}

// CODES = true: the compact traceback of the global mode.  Instead of the row records the kernel stores, per band
// cell, the outcome of the four equality tests the reference's backcal would make there (bsa_common.h, "COMPACT
// slot"; the test-only scalar restatement states the same rules, tests/test_oracle_codes.py), computed from
// the values the recurrence has in registers anyway:
//   M  = h == S          D = h == e + u          R = h + gapoe >= f + gape          Od = e' == gapoe
// with the reference's quirks at band position 0 (frame of ubegs[0], bsalign.h:3760-3771, 2632-2633) and beyond the
// previous row's band end (bsalign.h:3672-3678) applied when the row is assembled.  64 bytes per row at bw 128.
template<int W, int PW, bool CODES = false>
__global__ void __launch_bounds__(256, (W <= 8 && PW <= 1) ? 4 : 2) k_align8_fwd_pk(const Align8Args a){
	constexpr int BW = W * 16;
	constexpr int NQ = (W + 3) / 4;
	extern __shared__ __attribute__((aligned(16))) int8_t smem[];   // generic movx scratch: per pair (PW+1)*BW bytes + 17 ints
	constexpr int PAIR_LDS = ((PW + 1) * BW + 17 * 4 + 15) & ~15;
	const int lt = threadIdx.x;
	const int j = lt & 15;
	const uint32_t g = (blockIdx.x * 256u + lt) >> 4;
	uint32_t qlen[2], tlen[2];
	const uint8_t *qp[2], *tp[2];
	uint8_t *rowp[2];
	int *begs[2];
#pragma unroll
	for(int h = 0; h < 2; h++){
		const uint32_t idx = 2u * g + h;
		const bool live = idx < a.count;
		const uint32_t ppos = a.first + (live ? idx : 0u);
		const uint32_t pair = a.order[ppos];
		qlen[h] = a.qlen[pair]; tlen[h] = a.tlen[pair];
		qp[h] = a.qst + a.qpoff[pair]; tp[h] = a.tst + a.tpoff[pair];
		begs[h] = (int*)(a.rows + a.slot_off[ppos]);
		rowp[h] = (uint8_t*)begs[h] + bsa_begs_bytes(tlen[h]);
		if(!live || a.status[pair] != 0u) tlen[h] = 0;
	}
	int8_t *gl = smem + (lt >> 4) * (2 * PAIR_LDS);

	const int mode = a.mode & 3;
	const int gapo1 = a.gapo1, gape1 = a.gape1, gapo2 = a.gapo2, gape2 = a.gape2;
	const int GapE = trunc8(gape1), GapOE = trunc8(gapo1 + gape1);
	const int GapP = trunc8(gape2), GapQP = trunc8(gapo2 + gape2);
	const int GapOQ = sat8(GapOE - GapQP);
	const uint32_t GE = pk_splat(GapE), GOE = pk_splat(GapOE), GP = pk_splat(GapP), GQP = pk_splat(GapQP), GOQ = pk_splat(GapOQ);
	const uint32_t MIN63 = pk_splat(BSA_EPI8_MIN);
	const int cfirst = (PW == 2) ? (min(a.smin, gapo2 + gape2) - 1 - a.smax + (gapo2 + gape2))
	                             : (min(a.smin, gapo1 + gape1) - 1 - a.smax + (gapo1 + gape1));
	const int dsw = (PW == 2) ? (gapo1 - gapo2) / (gape2 - gape1) : (BW + 1);
	auto newcell_int = [&](int k) -> int { return (k == 0) ? cfirst : ((PW == 2 && k >= dsw) ? gape2 : gape1); };
	auto newcell_cum = [&](int n) -> int {
		int n1 = min(n, dsw);
		return cfirst + (n1 - 1) * gape1 + ((PW == 2) ? max(0, n - dsw) * gape2 : 0);
	};

	uint32_t u[W], e[W], q2[W];
	int ubA[2], ubB[2];
	// ---- row -1 (bsalign.h:2094-2140): identical for both halves
	{
		int bs = 0;
		const int first = trunc8(gapo1 + gape1 + a.smin - a.smax);
		const int xp = (PW == 2) ? (gapo2 - gapo1) / (gape1 - gape2) : 0;
#pragma unroll
		for(int k = 0; k < W; k++){
			int p = j * W + k, v;
			if(mode == BSA_MODE_OVERLAP) v = 0;
			else if(p == 0) v = first;
			else if(PW == 2) v = (p < xp) ? gape1 : gape2;
			else v = gape1;
			u[k] = pk_splat(v); bs += v;
			e[k] = MIN63; q2[k] = MIN63;
		}
		int inc = row_iscan16(bs);
		int base0 = (mode == BSA_MODE_OVERLAP) ? 0 : (a.smax - a.smin);
		ubB[0] = ubB[1] = base0 + inc;
		ubA[0] = ubA[1] = base0 + inc - bs;
	}
	// row record store (tiled block records, bsa_common.h): bytes of pair A sit in byte 1, of pair B in byte 3 of
	// every cell register; lane j owns block j.  Records go through a TileWriter per half (whole 64-byte tiles).
	constexpr uint32_t CELLS = ((uint32_t)(PW + 1) * W + 3u) & ~3u, BLK = CELLS + 4u;
	constexpr int RW = (int)(BLK / 4u), TG = (64u / BLK) ? (int)(64u / BLK) : 1;
	TileWriter<RW, TG> tw0, tw1;
	auto store_rows = [&](uint32_t row_index, const bool *act, const uint32_t *rbeg_v){
		uint32_t ra[RW], rb[RW];
#pragma unroll
		for(int d = 0; d < RW; d++){ ra[d] = 0u; rb[d] = 0u; }
		if constexpr (W >= 4){
#pragma unroll
			for(int n = 0; n < W / 4; n++){
#pragma unroll
				for(int arr = 0; arr <= PW; arr++){
					const uint32_t *x = (arr == 0) ? u : (arr == 1) ? e : q2;
					const uint32_t t01 = __builtin_amdgcn_perm(x[4*n+1], x[4*n],   0x07030501u);   // {A0, A1, B0, B1}
					const uint32_t t23 = __builtin_amdgcn_perm(x[4*n+3], x[4*n+2], 0x07030501u);   // {A2, A3, B2, B3}
					ra[arr * (W / 4) + n] = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
					rb[arr * (W / 4) + n] = __builtin_amdgcn_perm(t23, t01, 0x07060302u);
				}
			}
		} else {
#pragma unroll
			for(int k = 0; k < W; k++){
				ra[k >> 2] |= (uint32_t)(pk_get(u[k], 0) & 0xff) << (8 * (k & 3));
				rb[k >> 2] |= (uint32_t)(pk_get(u[k], 1) & 0xff) << (8 * (k & 3));
				if(PW >= 1){
					ra[(W + k) >> 2] |= (uint32_t)(pk_get(e[k], 0) & 0xff) << (8 * ((W + k) & 3));
					rb[(W + k) >> 2] |= (uint32_t)(pk_get(e[k], 1) & 0xff) << (8 * ((W + k) & 3));
				}
				if(PW == 2){
					ra[(2 * W + k) >> 2] |= (uint32_t)(pk_get(q2[k], 0) & 0xff) << (8 * ((2 * W + k) & 3));
					rb[(2 * W + k) >> 2] |= (uint32_t)(pk_get(q2[k], 1) & 0xff) << (8 * ((2 * W + k) & 3));
				}
			}
		}
		ra[RW - 1] = (uint32_t)ubA[0]; rb[RW - 1] = (uint32_t)ubA[1];
		tw0.push(rowp[0], (uint32_t)j, row_index, act[0], row_index == tlen[0], ra);
		tw1.push(rowp[1], (uint32_t)j, row_index, act[1], row_index == tlen[1], rb);
#pragma unroll
		for(int h = 0; h < 2; h++) if(act[h] && j == 0) begs[h][row_index] = (int)rbeg_v[h];
	};
	uint32_t rbeg[2] = {0, 0}, mov[2] = {0, 0}, i = 0;
	if constexpr (!CODES){
		bool act0[2] = { tlen[0] != 0, tlen[1] != 0 };
		store_rows(0, act0, rbeg);
	} else {
#pragma unroll
		for(int h = 0; h < 2; h++) if(tlen[h] != 0 && j == 0) begs[h][0] = 0;
	}
	constexpr uint32_t CW = (W >= 8) ? (uint32_t)W / 8u : 1u;
	int begq[2] = {0, 0};
	int cand_sc[2] = {BSA_SCORE_MIN, BSA_SCORE_MIN}, cand_te[2] = {0, 0};
	uint64_t twin[2] = {0, 0};       // 8 target bases per pair, reloaded every 8th row (the staged targets carry 16 bytes of padding)
	int rbz[2];     // 2 * max(tlen / qlen, 1): suggested max band step (bsalign.h:4008)
	bool rush32[2];
#pragma unroll
	for(int h = 0; h < 2; h++){
		if(tlen[h]) __builtin_memcpy(&twin[h], tp[h], 8);
		rbz[h] = 2 * max((int)(tlen[h] / max(qlen[h], 1u)), 1);
		// the rush test fits 32-bit arithmetic when rbz * tlen + qlen + bw cannot wrap
		rush32[h] = (unsigned long long)(uint32_t)rbz[h] * tlen[h] + qlen[h] + (uint32_t)BW + (uint32_t)rbz[h] < 0xFFFFFFFFull;
	}

	// rby = (int)((1.0 * i / tlen) * qlen) (bsalign.h:4009) depends on the row number only: every 16th row each lane
	// evaluates it for one of the next 16 rows (lane j: row i + j) and the row loop fetches its value with a lane
	// permute.  Same expression, same rounding, one double division per 16 rows and pair instead of one per row.
	int rby_tab[2] = {0, 0};
	const int rby_lane = (lt & 48) << 2;                 // byte address of lane 0 of this DPP row for ds_bpermute
	while(__any(i < tlen[0] || i < tlen[1])){
		bool act[2];
		int rh[2];
		bool slow = false;
		if(mode == BSA_MODE_GLOBAL && (i & 15u) == 0u){
#pragma unroll
			for(int h = 0; h < 2; h++)
				rby_tab[h] = (int)((1.0 * (double)(i + (uint32_t)j) / (double)tlen[h]) * (double)qlen[h]);
		}
#pragma unroll
		for(int h = 0; h < 2; h++){
			act[h] = i < tlen[h];
			// ---- band offset of this row (bsalign.h:3932-3946)
			const bool moved = (mov[h] != 0u) && (rbeg[h] + BW < qlen[h]);
			const uint32_t room = qlen[h] - (rbeg[h] + BW);
			mov[h] = moved ? min(mov[h], room) : 0u;
			rbeg[h] += mov[h];
			if(rbeg[h]) rh[h] = BSA_SCORE_MIN;
			else if(mode == BSA_MODE_OVERLAP || i == 0) rh[h] = 0;
			else if(PW < 2) rh[h] = gapo1 + gape1 * (int)i;
			else rh[h] = max(gapo1 + gape1 * (int)i, gapo2 + gape2 * (int)i);
			slow = slow || (act[h] && mov[h] >= (uint32_t)W);
		}
		// ---- row_movx (bsalign.h:2244-2392)
		if(__any(slow)){
			// generic path through LDS for both halves (rare: band jump of >= W cells)
#pragma unroll
			for(int h = 0; h < 2; h++){
				int8_t *su = gl + h * PAIR_LDS, *se = su + BW, *sq = su + 2 * BW;
				int *sub = (int*)(su + (PW + 1) * BW);
#pragma unroll
				for(int k = 0; k < W; k++){
					su[j * W + k] = (int8_t)pk_get(u[k], h);
					if(PW >= 1) se[j * W + k] = (int8_t)pk_get(e[k], h);
					if(PW == 2) sq[j * W + k] = (int8_t)pk_get(q2[k], h);
				}
				sub[j] = ubA[h]; if(j == 15) sub[16] = ubB[h];
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			// one half at a time (keeps the temporaries of only one pair live)
#pragma unroll
			for(int h = 0; h < 2; h++){
				const int8_t *su = gl + h * PAIR_LDS, *se = su + BW, *sq = su + 2 * BW;
				const int *sub = (const int*)(su + (PW + 1) * BW);
				const uint32_t mv = mov[h];
				const uint32_t hmask = h ? 0xFFFF0000u : 0x0000FFFFu;
				if(mv){
					uint32_t pp = min(mv - 1u, (uint32_t)BW - 1u), yy = pp / W, xx = pp % W;
					int s = sub[yy];
					for(uint32_t k = 0; k <= xx; k++) s += su[yy * W + k];
					rh[h] = s;
				}
				if(mv >= (uint32_t)BW){
#pragma unroll
					for(int k = 0; k < W; k++){ u[k] &= ~hmask; e[k] &= ~hmask; q2[k] &= ~hmask; }
					ubA[h] = ubB[h] = BSA_SCORE_MIN;
				} else if(mv){
					const uint32_t cyc = mv / W, m = mv % W, p0 = BW - mv;
#pragma unroll
					for(int k = 0; k < W; k++){
						const uint32_t src = j * W + k + mv;
						int vu, ve = 0, vq = 0;
						if(src < (uint32_t)BW){
							vu = su[src];
							if(PW >= 1) ve = se[src];
							if(PW == 2) vq = sq[src];
						} else vu = trunc8(newcell_int((int)(src - BW)));
						u[k] = (pk_splat(vu) & hmask) | (u[k] & ~hmask);
						if(PW >= 1) e[k] = (pk_splat(ve) & hmask) | (e[k] & ~hmask);
						if(PW == 2) q2[k] = (pk_splat(vq) & hmask) | (q2[k] & ~hmask);
					}
					auto new_ub = [&](uint32_t idx) -> int {
						int v;
						if(idx + cyc < 16u){
							uint32_t l = idx + cyc;
							v = sub[l];
							for(uint32_t k = 0; k < m; k++) v += su[l * W + k];
						} else v = sub[16];
						int nbefore = (int)(idx * W) - (int)p0;
						if(nbefore > 0) v += newcell_cum(nbefore);
						return v;
					};
					ubA[h] = new_ub((uint32_t)j);
					ubB[h] = new_ub((uint32_t)j + 1u);
				}
			}
			__builtin_amdgcn_wave_barrier();
		} else {
			int bacc[2] = {0, 0};
			for(uint32_t s = 0; __any((act[0] && s < mov[0]) || (act[1] && s < mov[1])); s++){
				const bool d0 = s < mov[0], d1 = s < mov[1];
				const uint32_t msk = (d0 ? 0x0000FFFFu : 0u) | (d1 ? 0xFFFF0000u : 0u);
				const int nci = newcell_int((int)s);
				const uint32_t dropped = u[0];
				const uint32_t in_u = pk_dpp_shl1(pk_splat(trunc8(nci)), u[0]);
#pragma unroll
				for(int k = 0; k + 1 < W; k++) u[k] = (u[k + 1] & msk) | (u[k] & ~msk);
				u[W - 1] = (in_u & msk) | (u[W - 1] & ~msk);
				if(PW >= 1){
					const uint32_t in_e = pk_dpp_shl1(0u, e[0]);
#pragma unroll
					for(int k = 0; k + 1 < W; k++) e[k] = (e[k + 1] & msk) | (e[k] & ~msk);
					e[W - 1] = (in_e & msk) | (e[W - 1] & ~msk);
				}
				if(PW == 2){
					const uint32_t in_q = pk_dpp_shl1(0u, q2[0]);
#pragma unroll
					for(int k = 0; k + 1 < W; k++) q2[k] = (q2[k + 1] & msk) | (q2[k] & ~msk);
					q2[W - 1] = (in_q & msk) | (q2[W - 1] & ~msk);
				}
				ubA[0] += d0 ? pk_get(dropped, 0) : 0; ubA[1] += d1 ? pk_get(dropped, 1) : 0;
				bacc[0] += d0 ? nci : 0; bacc[1] += d1 ? nci : 0;
			}
#pragma unroll
			for(int h = 0; h < 2; h++){
				const int nb = DPP_SHL(ubB[h] + bacc[h], ubA[h], 1);
				ubB[h] = mov[h] ? nb : ubB[h];
				if(mov[h]) rh[h] = ubA[h];
			}
		}
		// ---- sequences + S(x, y) for both halves: Sv[k] = (S_A << 8) | (S_B << 24)
		uint32_t s4[2][NQ];
#pragma unroll
		for(int h = 0; h < 2; h++){
			const int tb = (int)((twin[h] >> (8u * (i & 7u))) & 3u);
			uint32_t qc[NQ];
			if(act[h]) load_qcodes_pk<W>(qp[h] + rbeg[h] + j * W, qc);
			else { for(int n = 0; n < NQ; n++) qc[n] = 0x04040404u; }
			const uint32_t mr = (tb == 0) ? a.mrow[0] : (tb == 1) ? a.mrow[1] : (tb == 2) ? a.mrow[2] : a.mrow[3];
#pragma unroll
			for(int n = 0; n < NQ; n++) s4[h][n] = __builtin_amdgcn_perm(0xC1C1C1C1u, mr, qc[n]);
		}
		uint32_t Sv[W];
#pragma unroll
		for(int k = 0; k < W; k++){
			const uint32_t sel = 0x000C000Cu | ((uint32_t)(k & 3) << 8) | ((uint32_t)(4 + (k & 3)) << 24);   // {0, A.byte[k], 0, B.byte[k]}
			Sv[k] = __builtin_amdgcn_perm(s4[1][k >> 2], s4[0][k >> 2], sel);
		}
		// ---- row_cal (bsalign.h:2727-2793 / 2885-2960 / 3084-3179)
		uint32_t H0;
		{
			int h0[2];
#pragma unroll
			for(int h = 0; h < 2; h++){
				int hh = (rh[h] - ubA[h]) + pk_get(Sv[0], h);
				int t0 = pk_get(u[0], h) + ((PW == 0) ? gape1 : (PW == 1) ? pk_get(e[0], h) : max(pk_get(e[0], h), pk_get(q2[0], h)));
				hh = (hh >= t0) ? min(hh, BSA_EPI8_MAX) : BSA_EPI8_MIN;
				h0[h] = trunc8(hh);
			}
			H0 = pk_make(h0[0], h0[1]);
		}
		uint32_t f = MIN63, gq = MIN63;
		{
			uint32_t hc = (j == 0) ? H0 : Sv[0];
#pragma unroll
			for(int k = 0; k < W; k++){
				const uint32_t uk = u[k];
				uint32_t h;
				if(PW == 0){
					const uint32_t ee = pk_adds(uk, GE);
					h = pk_max(pk_max(ee, hc), f);
					f = pk_norm(pk_subs(pk_adds(h, GE), uk));
				} else if(PW == 1){
					const uint32_t ee = pk_adds(e[k], uk);
					h = pk_max(pk_max(ee, hc), f);
					f = pk_adds(f, GE);
					h = pk_adds(h, GOE);
					f = pk_norm(pk_subs(pk_max(f, h), uk));
				} else {
					const uint32_t ee = pk_adds(e[k], uk), qq = pk_adds(q2[k], uk);
					h = pk_max(pk_max(ee, hc), pk_max(qq, pk_max(f, gq)));
					f = pk_adds(f, GE);
					h = pk_adds(h, GOE);
					f = pk_norm(pk_subs(pk_max(f, h), uk));
					gq = pk_adds(gq, GP);
					h = pk_subs(h, GOQ);
					gq = pk_norm(pk_subs(pk_max(gq, h), uk));
				}
				if(k + 1 < W) hc = Sv[k + 1];
			}
		}
		{
			f = fpen_pk(f, ubA, ubB, W * gape1, j);
			if(PW == 2) gq = fpen_pk(gq, ubA, ubB, W * gape2, j);
		}
		uint32_t htail, ulast = 0;
		uint32_t accM = 0, accD = 0, accR = 0, accO = 0;        // CODES: flag planes, pair A in the low half, pair B in the high half
		const uint32_t u0_old = u[0], e0_old = e[0];
		uint32_t hfirst = 0;
		{
			uint32_t v = 0, z = (j == 0) ? H0 : Sv[0], h = 0;
#pragma unroll
			for(int k = 0; k < W; k++){
				const uint32_t uk = u[k];
				if(PW == 0){
					const uint32_t ee = pk_adds(uk, GE);
					h = pk_max(pk_max(ee, z), f);
					if constexpr (CODES){
						accM = pk_acc(accM, pk_eq(h, Sv[k]));
						accD = pk_acc(accD, pk_eq(h, ee));
						if(k == 0) hfirst = h;
					}
					u[k] = pk_norm(pk_subs(h, v));
					v = pk_norm(pk_subs(h, uk));
					f = pk_norm(pk_subs(pk_adds(h, GE), uk));
				} else if(PW == 1){
					uint32_t ee = pk_adds(e[k], uk);
					h = pk_max(pk_max(ee, z), f);
					if constexpr (CODES){
						accM = pk_acc(accM, pk_eq(h, Sv[k]));
						accD = pk_acc(accD, pk_eq(h, ee));
						if(k == 0) hfirst = h;
					}
					u[k] = pk_norm(pk_subs(h, v));
					v = pk_norm(pk_subs(h, uk));
					ee = pk_subs(pk_adds(ee, GE), h);
					e[k] = pk_max(ee, GOE);
					f = pk_adds(f, GE);
					h = pk_adds(h, GOE);
					const uint32_t fm = pk_max(f, h);
					if constexpr (CODES){
						accO = pk_acc(accO, pk_eq(e[k], GOE));
						accR = pk_acc(accR, pk_eq(fm, h));             // opening at this cell reaches the next one at least as well as extending
					}
					f = pk_norm(pk_subs(fm, uk));
				} else {
					uint32_t ee = pk_adds(e[k], uk), qq = pk_adds(q2[k], uk);
					h = pk_max(pk_max(ee, z), pk_max(qq, pk_max(f, gq)));
					u[k] = pk_norm(pk_subs(h, v));
					v = pk_norm(pk_subs(h, uk));
					ee = pk_subs(pk_adds(ee, GE), h);
					e[k] = pk_max(ee, GOE);
					qq = pk_subs(pk_adds(qq, GP), h);
					q2[k] = pk_max(qq, GQP);
					f = pk_adds(f, GE);
					h = pk_adds(h, GOE);
					f = pk_norm(pk_subs(pk_max(f, h), uk));
					gq = pk_adds(gq, GP);
					h = pk_subs(h, GOQ);
					gq = pk_norm(pk_subs(pk_max(gq, h), uk));
				}
				ulast = uk;
				if(k + 1 < W) z = Sv[k + 1];
			}
			htail = (PW == 0) ? h : (PW == 1) ? pk_subs(h, GOE) : pk_subs(h, GQP);
		}
		if constexpr (CODES){
			// ---- assemble and store the code row of both pairs (planes stay packed: pair A low half, pair B high half)
			constexpr uint32_t FULL = (W == 16) ? 0xFFFFu : ((1u << W) - 1u);
			if constexpr (PW == 0){ accR = FULL * 0x00010001u; accO = FULL * 0x00010001u; }
			// band position 0 of a band that starts at query column 0: backcal compares in the frame of the boundary
			// column (bsalign.h:3763-3767) -- 32-bit arithmetic on ubegs[0], rh and the raw score.  Only the first rows.
			if(__any(rbeg[0] == 0u || rbeg[1] == 0u)){
#pragma unroll
				for(int hf = 0; hf < 2; hf++){
					if(j == 0 && rbeg[hf] == 0u){
						const int hh0 = pk_get(hfirst, hf), s0 = pk_get(Sv[0], hf), uu0 = pk_get(u0_old, hf);
						const int ee0 = (PW == 0) ? (gapo1 + gape1) : pk_get(e0_old, hf);
						const bool m0 = hh0 == rh[hf] - ubA[hf] + s0;
						const bool d0 = ubA[hf] + hh0 - rh[hf] == uu0 + ee0;
						const uint32_t b0 = 1u << (W - 1 + 16 * hf);
						accM = (accM & ~b0) | (m0 ? b0 : 0u);
						accD = (accD & ~b0) | (d0 ? b0 : 0u);
					}
				}
			}
			// cells at / beyond the end of the previous row's band: x == bw decides M or I only, x > bw is always I
			{
				uint32_t kd = 0, km = 0;
#pragma unroll
				for(int hf = 0; hf < 2; hf++){
					const int lim = BW - (int)mov[hf] - j * W;                 // cells k < lim have x < bw
					const int nd = min(max(lim, 0), W), nm = min(max(lim + 1, 0), W);
					kd |= ((FULL << (W - nd)) & FULL) << (16 * hf);
					km |= ((FULL << (W - nm)) & FULL) << (16 * hf);
				}
				accD &= kd; accM &= km;
			}
			uint32_t dA0, dA1 = 0, dB0, dB1 = 0;
			if constexpr (W == 8){
				const uint32_t t1 = __builtin_amdgcn_perm(accD, accM, 0x06020400u);   // {M.A, D.A, M.B, D.B}
				const uint32_t t2 = __builtin_amdgcn_perm(accO, accR, 0x06020400u);   // {R.A, O.A, R.B, O.B}
				dA0 = __builtin_amdgcn_perm(t2, t1, 0x05040100u);
				dB0 = __builtin_amdgcn_perm(t2, t1, 0x07060302u);
			} else if constexpr (W == 4){
				const uint32_t lo = accM | (accD << 4) | (accR << 8) | (accO << 12);     // fields of 4 bits stay inside each half
				dA0 = lo & 0xFFFFu; dB0 = lo >> 16;
			} else {
				dA0 = (accM & 0xFFFFu) | (accD << 16); dA1 = (accR & 0xFFFFu) | (accO << 16);
				dB0 = (accM >> 16) | (accD & 0xFFFF0000u); dB1 = (accR >> 16) | (accO & 0xFFFF0000u);
			}
			// band offsets: lane (i mod 16) keeps the offset of row i, all 16 lanes store together every 16th row (one
			// 64-byte store instead of sixteen 4-byte ones) and at the pair's last row
#pragma unroll
			for(int hf = 0; hf < 2; hf++){
				if(act[hf]){
					uint32_t *rp = (uint32_t*)rowp[hf] + bsa_code_off(i, (uint32_t)j, CW);
					rp[0] = hf ? dB0 : dA0; if constexpr (CW > 1) rp[1] = hf ? dB1 : dA1;
					if((i & 15u) == (uint32_t)j) begq[hf] = (int)rbeg[hf];
					const bool lastrow = i + 1u == tlen[hf];
					if(((i & 15u) == 15u || lastrow) && (uint32_t)j <= (i & 15u)) begs[hf][(i & ~15u) + 1u + (uint32_t)j] = begq[hf];
				}
			}
		}
		// ---- tail (bsalign.h:2618-2636)
		{
			const uint32_t vlast = pk_norm(pk_subs(htail, ulast));
			const uint32_t vsh = pk_dpp_shr1(0u, vlast);
#ifdef BSA_DEBUG
			if(g == 0 && i == 0 && j < 2) printf("pk lane %d: htail %08x ulast %08x vlast %08x vsh %08x u0 %08x ubA %d %d ubB %d %d f %08x H0 %08x Sv0 %08x GOE %08x\n", j, htail, ulast, vlast, vsh, u[0], ubA[0], ubA[1], ubB[0], ubB[1], f, H0, Sv[0], GOE);
#endif
			u[0] = pk_norm(pk_subs(u[0], vsh));
#pragma unroll
			for(int h = 0; h < 2; h++){
				const int nB = ubB[h] + pk_get(vlast, h);
				int nA = DPP_SHR(0, nB, 1);
				if(j == 0) nA = ubA[h] + pk_get(u[0], h);
				ubA[h] = nA; ubB[h] = nB;
			}
			if(j == 0) u[0] = 0u;
		}
		if constexpr (!CODES) store_rows(i + 1, act, rbeg);
		else {
			if(mode != BSA_MODE_GLOBAL && __any((act[0] && rbeg[0] + BW >= qlen[0]) || (act[1] && rbeg[1] + BW >= qlen[1]))){
				// overlap / extend: while the band touches the query end, H at query column qlen - 1 is a candidate end
				// (bsalign.h:4023-4032); the lane that owns that cell keeps the best one it has seen (strictly greater wins)
#pragma unroll
				for(int hf = 0; hf < 2; hf++){
					const uint32_t pos = qlen[hf] - 1u - rbeg[hf];
					if(act[hf] && rbeg[hf] + BW >= qlen[hf] && (uint32_t)j == pos / W){
						int sc = ubA[hf];
#pragma unroll
						for(int k = 0; k < W; k++) sc += ((uint32_t)k <= pos % W) ? pk_get(u[k], hf) : 0;
						if(sc > cand_sc[hf]){ cand_sc[hf] = sc; cand_te[hf] = (int)i; }
					}
				}
			}
			if(__any((act[0] && i + 1u == tlen[0]) || (act[1] && i + 1u == tlen[1]))){
#pragma unroll
				for(int hf = 0; hf < 2; hf++){
					if(act[hf] && i + 1u == tlen[hf]){
						if(mode == BSA_MODE_GLOBAL){
							// global score = H at query column qlen - 1 of the last row (bsalign.h:4034-4037), kept in begs[tlen + 1]
							const uint32_t pos = qlen[hf] - 1u - rbeg[hf];
							int sc = ubA[hf];
#pragma unroll
							for(int k = 0; k < W; k++) sc += ((uint32_t)k <= pos % W) ? pk_get(u[k], hf) : 0;
							if(pos >= (uint32_t)BW){ if(j == 0) begs[hf][tlen[hf] + 1u] = (int)0x80000000u; }      // band never reached the query end
							else if((uint32_t)j == pos / W) begs[hf][tlen[hf] + 1u] = sc;
						} else {
							// end record: the candidates and the last row itself (row_max is taken by the traceback kernel)
							bsa_code_end_t *er = (bsa_code_end_t*)(rowp[hf] + (size_t)bsa_code_rows(tlen[hf]) * (64u * CW));
							er->cand_sc[j] = cand_sc[hf]; er->cand_te[j] = cand_te[hf];
							er->ubegs[j] = ubA[hf];
							if(j == 15){ er->ubegs[16] = ubB[hf]; er->rbeg_last = (int)rbeg[hf]; }
							int8_t *ub = (int8_t*)(er + 1) + j * W;
#pragma unroll
							for(int k = 0; k < W; k++) ub[k] = (int8_t)pk_get(u[k], hf);
						}
					}
				}
			}
		}
		// ---- adaptive band (bsalign.h:3331-3349) + global steering (bsalign.h:4006-4021)
		bool rush[2] = {false, false};
		// noisy = sum over the 16 blocks of |ubegs[j+1] - ubegs[j]|: both pairs in one packed u16 rotate-and-add
		// (|difference| <= 128 W, so 16 of them stay below 65536)
		uint32_t nzs;
		{
			const int d0 = ubB[0] - ubA[0], d1 = ubB[1] - ubA[1];
			uint32_t x = (uint32_t)max(d0, -d0) | ((uint32_t)max(d1, -d1) << 16);
			x = pk_addu(x, (uint32_t)DPP_ROR((int)x, 8)); x = pk_addu(x, (uint32_t)DPP_ROR((int)x, 4));
			x = pk_addu(x, (uint32_t)DPP_ROR((int)x, 2)); x = pk_addu(x, (uint32_t)DPP_ROR((int)x, 1));
			nzs = x;
		}
		const bool wide = __any(!rush32[0] || !rush32[1]);      // a pair too long for the 32-bit form of the rush test
#pragma unroll
		for(int h = 0; h < 2; h++){
			const int nzsum = (int)(h ? (nzs >> 16) : (nzs & 0xFFFFu));
			const int ub0 = DPP_BCAST(ubA[h], 0), ub16 = DPP_BCAST(ubB[h], 15);
			uint32_t nz = (uint32_t)(nzsum / 16);
			nz = nz / (uint32_t)W * 16u / 2u;
			const int noisy = (int)((16u > nz) ? 16u : nz);
			int rbx;
			if(i <= (uint32_t)BW / 4u) rbx = 0;
			else if(rbeg[h] + BW >= qlen[h]) rbx = 0;
			else if(ub0 + noisy < ub16) rbx = 2;
			else if(ub0 > ub16 + noisy) rbx = 0;
			else rbx = 1;
			if(mode == BSA_MODE_GLOBAL){
				const int rby = __builtin_amdgcn_ds_bpermute(rby_lane + (int)((i & 15u) << 2), rby_tab[h]);
				// "be quick to move to end": rbeg + rbz * (tlen - i - 1) + bw <= qlen + rbz - 1; its division is rare
				const uint32_t left = tlen[h] - i - 1u;
				if(!wide) rush[h] = act[h] && rbeg[h] + (uint32_t)rbz[h] * left + (uint32_t)BW <= qlen[h] + (uint32_t)rbz[h] - 1u;
				else {
					const unsigned long long lhs = (unsigned long long)rbeg[h] + (unsigned long long)(uint32_t)rbz[h] * left + (unsigned long long)BW;
					rush[h] = act[h] && lhs <= (unsigned long long)(uint32_t)(qlen[h] + (uint32_t)rbz[h] - 1u);
				}
				if((int)rbeg[h] < rby - BW) mov[h] = (uint32_t)(rbx + 1);
				else if((int)rbeg[h] > rby) mov[h] = (uint32_t)max(0, rbx - 1);
				else mov[h] = (uint32_t)rbx;
			} else mov[h] = (uint32_t)rbx;
		}
		if(__any(rush[0] || rush[1])){
#pragma unroll
			for(int h = 0; h < 2; h++){
				const uint32_t left = tlen[h] - i - 1u;
				if(rush[h]) mov[h] = 1u + (uint32_t)(qlen[h] - (rbeg[h] + BW)) / max(left, 1u);
			}
		}
		i++;
		if((i & 7u) == 0u){
#pragma unroll
			for(int h = 0; h < 2; h++) if(i < tlen[h]) __builtin_memcpy(&twin[h], tp[h] + i, 8);
		}
	}
}

template<int W, int PW>
static hipError_t launch_fwd_pk(const Align8Args &a, hipStream_t st){
	constexpr int BW = W * 16;
	constexpr int PAIR_LDS = ((PW + 1) * BW + 17 * 4 + 15) & ~15;
	const uint32_t groups = (a.count + 1) / 2;
	const uint32_t blocks = (groups + 15) / 16;
	if(blocks == 0) return hipSuccess;
	hipLaunchKernelGGL((k_align8_fwd_pk<W, PW>), dim3(blocks), dim3(256), 2 * PAIR_LDS * 16, st, a);
	return hipGetLastError();
}

template<int W, int PW>
static hipError_t launch_fwd_codes(const Align8Args &a, hipStream_t st){
	constexpr int BW = W * 16;
	constexpr int PAIR_LDS = ((PW + 1) * BW + 17 * 4 + 15) & ~15;
	const uint32_t groups = (a.count + 1) / 2;
	const uint32_t blocks = (groups + 15) / 16;
	if(blocks == 0) return hipSuccess;
	hipLaunchKernelGGL((k_align8_fwd_pk<W, PW, true>), dim3(blocks), dim3(256), 2 * PAIR_LDS * 16, st, a);
	return hipGetLastError();
}

// The compact path is sound only when no saturating operation can clamp and no int8 store can wrap: then the stored
// differences are exact and the flags equal backcal's tests on reconstructed scores (checked against the literal
// restatement on 60 k random pairs in all three modes, tests/test_oracle_codes.py).  With m = max score, g = |gapo + gape|, every
// intermediate of the recurrence lies within [-(m + 3g), m + g] and the synthetic band-edge cell is
// min(smin, gapoe) - 1 - smax + gapoe (bsalign.h:2362); the limits below keep both well inside int8 (with larger
// scores the reference's own traceback stops terminating on divergent inputs and the two paths can disagree on which
// pairs get flagged).
bool bsa_align8_codes_supported(const Align8Args &a, int pw){
	if(pw == 2) return bsa_align8_x_supported(a, 2);        // two-piece gaps: 8 bits per cell, exact-arithmetic forward kernel only
	if(!bsa_align8_pk_supported(a, pw) || pw > 1) return false;
	const int g = -((int)(int8_t)(a.gapo1 + a.gape1)), m = a.smax, n = -a.smin;
	if(m < 0 || n < 0 || g < 0) return false;
	// (m + 2 n <= 128: row 0's seed at column 0, (min - max) + S, is inserted as a byte (bsalign.h:2910); a mismatch there must not wrap it)
	return m + 3 * g <= 64 && n + m + g <= 100 && m + 2 * n <= 128;
}

hipError_t bsa_launch_align8_fwd_codes(const Align8Args &a, int pw, hipStream_t st){
#define CODES_CASE(WW) case WW: return pw == 0 ? launch_fwd_codes<WW, 0>(a, st) : launch_fwd_codes<WW, 1>(a, st);
	switch(a.bw / 16){
		CODES_CASE(4) CODES_CASE(8) CODES_CASE(16)
		default: return hipErrorInvalidValue;
	}
#undef CODES_CASE
}

template<int W>
static hipError_t launch_fwd_pk_pw(const Align8Args &a, int pw, hipStream_t st){
	if(pw == 0) return launch_fwd_pk<W, 0>(a, st);
	if(pw == 1) return launch_fwd_pk<W, 1>(a, st);
	return launch_fwd_pk<W, 2>(a, st);
}

bool bsa_align8_pk_supported(const Align8Args &a, int pw){
	const uint32_t W = a.bw / 16;
	if(!(W == 4 || W == 8 || W == 16)) return false;
	const int GapE = (int)(int8_t)a.gape1, GapOE = (int)(int8_t)(a.gapo1 + a.gape1);
	if(GapE > 0 || GapOE > 0) return false;
	if(pw == 2){
		const int GapP = (int)(int8_t)a.gape2, GapQP = (int)(int8_t)(a.gapo2 + a.gape2);
		if(GapP > 0 || GapQP > 0 || GapOE - GapQP < 0) return false;
	}
	return true;
}

hipError_t bsa_launch_align8_fwd_pk(const Align8Args &a, int pw, hipStream_t st){
	switch(a.bw / 16){
		case 4:  return launch_fwd_pk_pw<4>(a, pw, st);
		case 8:  return launch_fwd_pk_pw<8>(a, pw, st);
		case 16: return launch_fwd_pk_pw<16>(a, pw, st);
		default: return hipErrorInvalidValue;
	}
}
