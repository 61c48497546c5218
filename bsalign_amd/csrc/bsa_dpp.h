// bsa_dpp.h -- DPP row primitives and the active F-loop shared by the 8-bit kernels
#pragma once
#include "bsa_common.h"

// Every DPP move is made opaque to the optimiser.  ROCm 7.2's DPP combiner folds a v_mov_b32_dpp into a
// following subtraction as `v_subrev_u32_dpp vdst, vsrc(dpp), vdst`, and on gfx950 that instruction returns
// dpp(vdst) - vsrc instead of vdst - dpp(vsrc) (measured on hardware, scratch/dpp_test.hip); keeping the move
// explicit costs one VALU op and is always right.
static __device__ __forceinline__ int dpp_keep(int x){ asm("" : "+v"(x)); return x; }
// (a fill that is the constant 0 is what bound_ctrl gives a lane without a source: then the compiler need not put the fill into the destination first --
// with old = 0 every such move was preceded by a v_mov_b32 v, 0)
#define DPP_MOV_(fill, x, ctrl) ((__builtin_constant_p(fill) && (fill) == 0) ? __builtin_amdgcn_update_dpp(0, (x), (ctrl), 0xf, 0xf, true) : __builtin_amdgcn_update_dpp((fill), (x), (ctrl), 0xf, 0xf, false))
#define DPP_SHR(fill, x, n)  dpp_keep(DPP_MOV_((fill), (x), 0x110 + (n)))  // lane j <- lane j-n (row of 16)
#define DPP_SHL(fill, x, n)  dpp_keep(DPP_MOV_((fill), (x), 0x100 + (n)))  // lane j <- lane j+n
#define DPP_BCAST(x, n)      dpp_keep(__builtin_amdgcn_update_dpp(0, (x), 0x150 + (n), 0xf, 0xf, true))       // row_newbcast:n (every lane has a source)
#define DPP_ROR(x, n)        dpp_keep(__builtin_amdgcn_update_dpp(0, (x), 0x120 + (n), 0xf, 0xf, true))       // row rotate right

#define BIGNEG (-(1 << 28))

static __device__ __forceinline__ int sat8(int v){ return min(max(v, -128), 127); }     // _mm_adds_epi8 / _mm_subs_epi8
static __device__ __forceinline__ int trunc8(int v){ return (int)(int8_t)v; }            // int -> b1i store
static __device__ __forceinline__ int row_sum16(int x){                                   // sum over the 16 lanes of a DPP row, in every lane
	x += DPP_ROR(x, 8); x += DPP_ROR(x, 4); x += DPP_ROR(x, 2); x += DPP_ROR(x, 1);
	return x;
}
static __device__ __forceinline__ int row_iscan16(int x){                                 // inclusive prefix sum over the DPP row
	x += DPP_SHR(0, x, 1); x += DPP_SHR(0, x, 2); x += DPP_SHR(0, x, 4); x += DPP_SHR(0, x, 8);
	return x;
}

// active F-loop, literal serial form (bsalign.h:2639-2652): 15 dependent lane-to-lane steps
static __device__ __forceinline__ int fpen_serial(int f, int ubA, int ubB, int t, int j){
	int fs = DPP_SHR(BSA_EPI8_MIN, f, 1);       // fs[j] = f[j-1], fs[0] = -63
	const int dd = ubB - ubA;
#pragma unroll
	for(int step = 1; step < 16; step++){
		int sv  = t + fs - dd;                  // s leaving lane j
		int sin = DPP_SHR(0, sv, 1);            // s entering lane j
		int cand = (fs < sin) ? trunc8(sin) : fs;
		fs = (j == step) ? cand : fs;
	}
	return fs;
}

// same result through a 4-step max-plus scan; falls back to the serial form when an int->int8
// truncation could have fired (some entering s > 127), which is the only way the two can differ
static __device__ __forceinline__ int fpen(int f, int ubA, int ubB, int t, int j){
#ifdef BSA_FPEN_SERIAL
	return fpen_serial(f, ubA, ubB, t, j);
#else
	const int fs = DPP_SHR(BSA_EPI8_MIN, f, 1);
	const int c  = t - (ubB - ubA);             // fs'[j+1] = max(fs[j+1], fs'[j] + c[j])
	int A = DPP_SHR(BIGNEG, c, 1);              // map of lane j: x -> max(x + A, B); lane 0 ignores x
	int B = fs;
#define FPEN_STEP(n) { int A1 = DPP_SHR(0, A, n); int B1 = DPP_SHR(BIGNEG, B, n); B = max(B1 + A, B); A = max(A1 + A, 2 * BIGNEG); }
	FPEN_STEP(1) FPEN_STEP(2) FPEN_STEP(4) FPEN_STEP(8)
#undef FPEN_STEP
	const int sprev = DPP_SHR(BIGNEG, B + c, 1);
	if(__any(sprev > 127)) return fpen_serial(f, ubA, ubB, t, j);
	return B;
#endif
}

