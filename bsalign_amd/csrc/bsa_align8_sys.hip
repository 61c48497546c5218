// bsa_align8_sys.hip -- whole-query bands longer than 256 columns (the reference CLI's default `-W 0` on long reads, main.c:314-315;
// the first command of example/run.sh): the 8-bit DP of banded_striped_epi8_seqalign_pairwise (bsalign.h:3854-4050) with the band
// never moving, as a SYSTOLIC wavefront, and its traceback from 4-bit codes (two-piece gaps: 8-bit codes); all three modes.
//
// A band that covers the whole query never moves (bsalign.h:3338: qoff + bw >= qlen), every row has band offset 0, and inside the
// exact-arithmetic guard of the compact path (bsa_align8_sys_supported = the score bounds of bsa_align8_codes_supported) none of
// the reference's saturating int8 operations clamps: the stored differences are exact, i.e. the DP is the affine-gap recurrence on
// absolute scores with the reference's own boundary rules.  No striping is needed for that: a wave owns 64 consecutive target rows, lane l
// row 64 b + l, and walks the query; at step t it computes column x = t - l, so the row above (lane l - 1) delivered H(x, y-1) and
// E(x, y-1) exactly one step earlier -- a DPP wave shift per value and step, no LDS, no waiting.  The last row of a wave's block is the
// boundary of the rows below: it leaves through a 256-entry LDS ring -- to the next wave of the pair's workgroup (one or four waves per
// pair, each 192 steps behind the one above, a barrier per 64 steps), or, below the last wave, to HBM and back, 64 columns per
// coalesced access (lane 0's shift-in is the ring entry of the step: the LDS read is the `old` operand of the DPP move).
// Kept literally (absolute-score form of the rules, cf. the POA wavefront bsa_poa_wf.hip):
//   * row -1 (row_init, bsalign.h:2094-2140): H = gapo + gape (x + 1), e = -63; ubegs[0] = smax - smin with u[0] = gapo + gape + smin - smax
//   * band cell 0 (bsalign.h:2899-2907) with rh = 0 on row 0 and gapo + gape y below (bsalign.h:3932-3946): h0 = rh - ubegs[0] + S, kept
//     (clamped to 63) if >= u[0] + e[0], else -63; every later row has ubegs[0] = H(0, y-1), u[0] = 0 (the re-basing of :2632-2633)
//   * F restarts from "H above - 63" at every running block of W = bw / 16 cells (bsalign.h:2909-2931 + the F-penetration :2639-2652):
//     never binding inside the guard (see the kernel), so the striping of the reference's band does not enter
// Traceback codes (the four facts backcal tests per cell, HISTORY section 3): M (h == S-path; at column 0 against the UNclamped
// rh + S), D (h == u + e; at column 0 in the frame mismatch of the re-based row: h - rh == u[0] + e[0]), R (h + gapo + gape >= f + gape:
// an insertion reaching the next cell opens here), Od (the stored e is a fresh opening).  Row y, columns 32 k .. 32 k + 31: four
// dwords {M, D, R, Od} of the wavefront steps t = x + (y & 63), 32 k <= t < 32 k + 32, step t at bit 31 - (t & 31); the 64 rows of a block
// store their dwords of the same k side by side (one kilobyte per store instruction).  Overlap / extend mode: row -1 and the score left of
// column 0 as row_init / the driver set them, the end cell searched for (last query column, then row_max of the last row in the reference's
// striping).  Two-piece gaps: a second vertical state Q through the rings, a second horizontal chain G, eight facts per cell.
#include "bsa_common.h"
#include <type_traits>

static __device__ __forceinline__ int sys_shr1(int fill, int x){                 // lane l <- lane l - 1 over the whole wave; lane 0 <- fill
	return __builtin_amdgcn_update_dpp(fill, x, 0x138, 0xf, 0xf, false);         // DPP wave_shr:1, bound_ctrl off: lane 0 keeps `fill`
}

struct SysHdr { int32_t score, qe, te, reserved; };      // the alignment's end cell (global: the last cell)

// slot: header, the boundary row (qlen + 256 entries of {H * 32 | query code << 3, E * 32}), the code tiles, the CIGAR tail.
// Code tiles: block b of 64 rows, word w = t >> 5 of the wavefront step t = x + (y & 63): 64 lanes x 16 bytes {M, D, R, Od}, step t at
// bit 31 - (t & 31) -- every 32 steps the wave stores ONE contiguous kilobyte.
// Two-piece gaps: boundary entries carry Q as well (16 bytes), and a cell has eight facts {A, D, D2, B | R1, R2, Od1, Od2} (section 3: the
// 8-bit codes of the compact path) -- two kilobytes per tile, stored as two contiguous kilobytes.
static __host__ __device__ inline uint32_t bsa_sys_words(uint32_t qlen){ return (((qlen + 15u) & ~15u) + 63u + 31u) / 32u; }      // wavefront steps of the reference's whole band (roundup(qlen, 16) columns: the checked kernel runs the padding cells too)
static __host__ __device__ inline size_t bsa_sys_bnd_off(){ return sizeof(SysHdr); }
static __host__ __device__ inline size_t bsa_sys_lasth_off(uint32_t qlen, int pw){ return bsa_sys_bnd_off() + ((size_t)qlen + 256) * (pw == 2 ? 16 : 8); }      // H of the last target row (overlap / extend: row_max)
static __host__ __device__ inline size_t bsa_sys_codes_off(uint32_t qlen, int pw){ return (bsa_sys_lasth_off(qlen, pw) + ((size_t)qlen + 64) * 4 + 1023) & ~(size_t)1023; }
static __host__ __device__ inline size_t bsa_sys_codes_bytes(uint32_t qlen, uint32_t tlen, int pw){ return (size_t)((tlen + 63u) / 64u) * bsa_sys_words(qlen) * (pw == 2 ? 2048u : 1024u); }

size_t bsa_align8_sys_slot_bytes(uint32_t qlen, uint32_t tlen, int pw){
	return ((bsa_sys_codes_off(qlen, pw) + bsa_sys_codes_bytes(qlen, tlen, pw) + ((size_t)qlen + tlen + 16) * 4) + 1023) & ~(size_t)1023;
}

// 0: not for this kernel; 1: inside the static guard (nothing of the reference's int8 arithmetic can clamp); 2: outside it, but the
// kernel can CHECK, pair by pair, that nothing did (k_align8_fwd_sys<.., CHK = true>, below).
int bsa_align8_sys_supported(const Align8Args &a, int pw){
	int g = -((int)(int8_t)(a.gapo1 + a.gape1));
	const int m = a.smax, n = -a.smin;
	if(m < 0 || n < 0 || g < 0 || (int8_t)a.gape1 > 0 || (int8_t)a.gapo1 > 0 || ((int8_t)a.gapo1 == 0) != (pw == 0)) return 0;
	if(pw == 2){
		// piece 2 opens dearer and extends cheaper (bsalign.h:2084-2092); the bound is taken with the dearer opening
		const int ge = -(int)(int8_t)a.gape1, go = -(int)(int8_t)a.gapo1, ge2 = -(int)(int8_t)a.gape2, go2 = -(int)(int8_t)a.gapo2;
		if(ge2 < 0 || go2 <= go || ge2 >= ge) return 0;
		g = max(g, go2 + ge2);
	}
	// (m + 2 n <= 128: the seed of row 0, column 0 -- (min - max) + S in global / extend mode, bsalign.h:2899-2910 -- stays a byte even on a mismatch;
	// beyond, the reference's byte wraps: the kernel wraps it too, and the checked form then sees the differences that leave int8)
	if(m + 3 * g <= 64 && n + m + g <= 100 && m + 2 * n <= 128) return 1;
	// The checked form (see the kernel): substitution scores no lower than the -63 sentinel, gap costs that keep h + gapoe and the stored e / q
	// inside int8 whatever the cell holds, row -1's first difference (row_init, bsalign.h:2100) not truncated, penalties that are what their
	// int8 truncations say.
	if((int)a.gapo1 + (int)a.gape1 != (int)(int8_t)(a.gapo1 + a.gape1) || (int)a.gape1 != (int)(int8_t)a.gape1 || (int)a.gapo1 != (int)(int8_t)a.gapo1) return 0;
	if(pw == 2 && ((int)a.gape2 != (int)(int8_t)a.gape2 || (int)a.gapo2 != (int)(int8_t)a.gapo2 || (int)a.gapo2 + (int)a.gape2 < -63)) return 0;
	if(n > 63 || m > 63 || g > 63 || -g - n - m < -128) return 0;
	return 2;
}

// Scores are carried times 32: the five low bits of the H register hold the query code of the column (code << 3), which travels down
// the lanes with H in ONE wave shift and is the bit offset v_bfe_i32 takes the substitution score from (the matrix column of the
// lane's target base, four bytes in a register).  Per step and lane (affine gaps): 2 DPP moves, 1 and, 1 bfe, 1 shift-add, 1 max3,
// 3 adds, 2 max, 1 and-or, and two instructions per traceback fact (difference, v_alignbit of its sign into the lane's bit plane).
// The F restart of the reference's running blocks (F = max(F, H above - 63)) is left out: inside the guard g <= 21, and the F that
// reaches cell x is >= H(x - 1, y) + gapo + gape >= H(x - 1, y - 1) + 2 (gapo + gape) > H(x - 1, y - 1) - 63, so it never binds.
// One workgroup per pair, NWV waves (1 or 4): wave w owns rows 64 (NWV s + w) .. + 63 of super-block s and runs LAG = 192 steps behind
// wave w - 1, whose last row reaches it through an LDS ring (column c at entry c mod 256, written one step late so that a 32-step body
// never wraps); one barrier per 64 steps keeps writer and reader an epoch apart.  Wave 0's upper boundary and the last wave's lower one
// go through HBM as before, 64 columns per access.  Four waves cut a pair's latency (and the memory a full chip needs) by four.
#define SYS_LAG 192
//
// CHK (scores outside the static guard, bsa_align8_sys_supported() == 2): exact arithmetic is the reference's arithmetic on a pair as long as
// none of its saturating int8 operations changes a value that is used and no running block's F (G) restart at -63 (bsalign.h:2897, 2909)
// wins against the F that really arrives.  Going through row_cal's operations (bsalign.h:2885-2960; e, u the stored row, D = H(x-1, y-1)):
//   e + u (adds)                clamps at -128 only; h = max(e + u, S, f) >= S >= -63 hides it, and the stored e' = max(.. - h, gapoe) as well
//   f + gape, h + gapoe         the first clamps at -128 only and then loses against h + gapoe >= -63 - g >= -128; the second cannot (g <= 63)
//   (e + u + gape) - h          clamps at -128 only, below gapoe either way
//   u' = h - v, v' = h - u      the new row's horizontal and vertical differences H(x, y) - H(x-1, y), H(x, y) - H(x, y-1): CHECKED per cell
//   f' = max(..) - u            = F(x+1, y) - H(x, y-1) <= v' + gape from above; from below >= -2 g (row 0: >= -63 - g): inside int8
//   F-penetration (:2639-2652)  its int -> int8 truncation needs a value above 127; every candidate is <= the F that arrives <= h <= 127
//   block restart               F(x, y) - D >= -63 CHECKED per query column (x >= 1): then max(-63, F) = F at every block start, whatever W is
// (two pieces: the same for q / G with gapo2 + gape2; linear gaps: e + u is u + gape).  So a lane tracks the minimum and maximum of its cells' two
// differences and the minimum of F - D (G - D); a pair any of whose cells leaves [-128, 127] resp. goes below -63 is flagged BSA_ST_TRACE and
// left to the literal kernels (bsa_align_batch re-runs it there).  Overlap / extend: row_max (bsalign.h:3213) also looks at the band's padding
// cells behind the query (S = -63, up to 15 of them), so there the checked kernel computes and checks those as well (marker bit 2 of a
// boundary entry; nothing of theirs is kept -- in exact arithmetic a padding cell never exceeds its left neighbour).
#ifndef SYS_CHK_WAVES
#define SYS_CHK_WAVES 4
#endif
template<int PW, int NWV, bool CHK>
__global__ void __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(PW == 2 ? 2 : CHK ? SYS_CHK_WAVES : 4))) k_align8_fwd_sys(const Align8Args a){
	using Ent = typename std::conditional<PW == 2, int4, int2>::type;        // a boundary cell: {H * 32 | q << 3, E * 32 [, Q * 32, -]}
	auto ent = [](int p_, int e_, int q_) -> Ent { if constexpr(PW == 2) return make_int4(p_, e_, q_, 0); else return make_int2(p_, e_); };
	__shared__ Ent rg[NWV + 1][256 + 2];            // rg[w]: the row above wave w's rows, rg[w + 1]: its own last row; {H * 32 | q << 3, E * 32} per column
	__shared__ long long bestw[NWV];
	const uint32_t ppos = a.first + blockIdx.x;
	const uint32_t pair = a.order[ppos];
	const int wv = (NWV == 1) ? 0 : (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
	const int qlen = (int)a.qlen[pair], tlen = (int)a.tlen[pair];
	uint8_t *slot = a.rows + a.slot_off[ppos];
	SysHdr *hdr = (SysHdr*)slot;
	if(a.status[pair] != 0u || qlen == 0 || tlen == 0){ if(threadIdx.x == 0) hdr->score = (int)0x80000000u; return; }
	Ent *bnd = (Ent*)(slot + bsa_sys_bnd_off());
	uint8_t *codes = slot + bsa_sys_codes_off((uint32_t)qlen, PW);
	const int NW = (int)bsa_sys_words((uint32_t)qlen);
	const uint8_t *qp = a.qst + a.qpoff[pair], *tp = a.tst + a.tpoff[pair];
	const int GO = a.gapo1, GE = a.gape1, GOE = GO + GE, GE5 = GE * 32, GOE5 = GOE * 32;
	const int GO2 = a.gapo2, GP = a.gape2, GQP = GO2 + GP, GP5 = GP * 32, GQP5 = GQP * 32;        // two-piece gaps: the second piece
	const int xp = (PW == 2) ? (GO2 - GO) / (GE - GP) : 1;                                         // row_init: the first xp cells cost gape1, the rest gape2 (bsalign.h:2102-2112)
	const int type = a.mode & 3;
	const bool ovl = type == BSA_MODE_OVERLAP, ends = type != BSA_MODE_GLOBAL;       // overlap: row -1 is all zero and H left of column 0 is 0 (row_init bsalign.h:2094-2140, :3932-3946); ends: the end cell is searched for
	const int first_u = ovl ? 0 : (int)(int8_t)(GOE + a.smin - a.smax), B0 = ovl ? 0 : a.smax - a.smin;
	int32_t *lastH = (int32_t*)(slot + bsa_sys_lasth_off((uint32_t)qlen, PW));
	long long bestc = (long long)0x8000000000000000ull;                              // best cell of the last query column: (score, first row) as one key
	const int xlim = (CHK && ends) ? ((qlen + 15) & ~15) : qlen;                     // columns computed: the query's, or (checked, overlap / extend) the reference's whole band
	const int nsb = (tlen + 64 * NWV - 1) / (64 * NWV), nsteps = xlim + 63, cmax = xlim + 192;
	int badl = 0;                                                                    // CHK: this lane saw a cell the reference's int8 arithmetic may treat differently
	const int Ttot = nsteps + 1 + SYS_LAG * (NWV - 1);
	// row -1 (row_init): H = gapo + gape (x + 1), e = -63, with the query codes
	for(int c = (int)threadIdx.x; c < cmax; c += 64 * NWV){
		int h = ovl ? 0 : GOE + GE * c;
		if(PW == 2 && !ovl){ const int n1 = min(c, xp - 1); h = GOE + n1 * GE + (c - n1) * GP; }
		const int q = (c < qlen) ? ((int)qp[c] & 3) : 0;
		bnd[c] = ent(h * 32 + q * 8 + ((CHK && c >= qlen) ? 4 : 0), (h + BSA_EPI8_MIN) * 32, (h + BSA_EPI8_MIN) * 32);
	}
	for(int sb = 0; sb < nsb; sb++){
		__builtin_amdgcn_s_waitcnt(0);                           // the boundary row is in memory before this super-block reads it
		if(NWV > 1) __syncthreads();
		const int blk = sb * NWV + wv;
		const bool live = blk * 64 < tlen, lastsb = sb + 1 == nsb;
		const int y = blk * 64 + lane;
		const int tb = (y < tlen) ? (int)tp[y] & 3 : 0;
		const int mr = (int)((tb == 0) ? a.mrow[0] : (tb == 1) ? a.mrow[1] : (tb == 2) ? a.mrow[2] : a.mrow[3]);     // matrix[q * 4 + tb], q = 0..3, one byte each
		const int rh = (y == 0 || ovl) ? 0 : (PW == 2) ? max(GO + GE * y, GO2 + GP * y) : GO + GE * y;        // H left of column 0 (bsalign.h:3932-3946)
		const bool lastrow = ends && y == tlen - 1;
		Ent *irng = rg[wv], *orng = rg[wv + 1];
		int P = 0, E = 0, Q = 0, Hd = 0, F = 0, G = 0;
		uint32_t pM = 0, pD = 0, pR = 0, pO = 0;                 // NOT-facts, newest step in bit 0 (two pieces: pM = A, pR / pO = R1 / Od1)
		uint32_t pD2 = 0, pB = 0, pR2 = 0, pO2 = 0;             // two pieces: D2, B (as it is), R2, Od2
		int chi = 0, clo = 0, cfl = 0;                           // CHK: max / min of the cells' differences, min of F - D (G - D)
		uint4 *cp = (uint4*)codes + ((size_t)blk * NW * (PW == 2 ? 128 : 64) + lane);
		Ent nxt = ent(0, 0, 0);
		if(wv == 0){ irng[lane] = bnd[lane]; irng[64 + lane] = bnd[64 + lane]; nxt = bnd[128 + lane]; }
		int drained = 0;
		auto top = [&](int t){                                   // t % 64 == 0: ring maintenance against HBM, 64 columns per coalesced access
			if(wv == 0 && t >= 64){
				irng[(t + 64 + lane) & 255] = nxt;
				const int c = t + 128 + lane;
				nxt = (c < cmax) ? bnd[c] : ent(0, 0, 0);
			}
			if(wv == NWV - 1 && t >= 128 && !lastsb){         // (column c was written at step c + 64)
				const int c = t - 128 + lane;
				bnd[c] = orng[c & 255];
				drained = t - 64;
			}
		};
		auto flush = [&](int t){
			const uint32_t sh = 31u - ((uint32_t)t & 31u);
			if constexpr(PW == 2){
				cp[0] = make_uint4(~(pM << sh), ~(pD << sh), ~(pD2 << sh), pB << sh);
				cp[64] = make_uint4(~(pR << sh), ~(pR2 << sh), ~(pO << sh), ~(pO2 << sh));
				cp += 128;
			} else {
				*cp = (PW == 0) ? make_uint4(~(pM << sh), ~(pD << sh), 0xffffffffu, 0xffffffffu) : make_uint4(~(pM << sh), ~(pD << sh), ~(pR << sh), ~(pO << sh));
				cp += 64;
			}
		};
		// GEN: a step in which some lane is at column 0 or at / beyond the query's last column (the first 64 and the last 64 + steps of a
		// block): the steady step plus, for the lane at column 0, the seed rule and its two facts, and the captures at the last column.
		// Lanes left of column 0 or right of the last column compute on whatever arrives; nothing of theirs is kept.
		auto step = [&](auto gen, auto last, const int t, const Ent b, Ent *ob){         // b: the ring entry of the step (all lanes read the same one; lane 0 uses it); ob: where column t - 64 of the last row goes
			constexpr bool GEN = decltype(gen)::value, LAST = decltype(last)::value;     // LAST: the block holds the last target row and its H is wanted
			if(lane == 63 && (!GEN || t >= 64)) *ob = ent(P, E, Q);                   // the previous step's cell of the last row: column t - 64
			const int Pi = sys_shr1(b.x, P);
			const int Eu = (PW == 0) ? 0 : sys_shr1(b.y, E);
			int Qu = 0;
			if constexpr(PW == 2) Qu = sys_shr1(b.z, Q);
			const int Hu = Pi & ~31;
			int S = __builtin_amdgcn_sbfe(mr, (unsigned)Pi, 8u);
			const int Ein = (PW == 0) ? Hu + GE5 : Eu;
			if constexpr(CHK && GEN){ if(Pi & 4) S = BSA_EPI8_MIN; }                      // a padding cell behind the query (bsalign.h:2166-2191: profile rows beyond qlen hold -63)
			int diag = S * 32 + Hd, cmpM = 0, cmpD = 0, cmpD2 = 0;
			const bool col0 = GEN && t == lane;
			int dF = 0;
			if constexpr(CHK){ dF = F - Hd; if constexpr(PW == 2) dF = min(dF, G - Hd); }  // what arrives from the left, in the cell's frame
			if constexpr(GEN){
				if(col0){
					// band cell 0: the seed rule and the F restart; the M / D facts of column 0 in their own frames
					const int Hui = Hu >> 5, Eui = Eu >> 5;
					const int ub0 = (y == 0) ? B0 : Hui;
					const int u0 = (y == 0) ? first_u : 0;
					const int e0 = Eui - Hui, q0 = (Qu >> 5) - Hui;
					int h0 = rh - ub0 + S;
					const int tt = u0 + (PW == 0 ? GE : PW == 1 ? e0 : max(e0, q0));
					h0 = (h0 >= tt) ? min(h0, BSA_EPI8_MAX) : BSA_EPI8_MIN;
					h0 = (int)(int8_t)h0;                                      // mm_insert_epi8 (bsalign.h:2910): row 0 of global / extend mode can seed below -128 (a mismatch at (0, 0): min - max + min), and the byte wraps
					diag = (ub0 + h0) * 32;
					cmpM = (rh + S) * 32;
					cmpD = (rh + u0 + (PW == 0 ? GOE : e0)) * 32;
					cmpD2 = (rh + u0 + q0) * 32;
					F = (ub0 + BSA_EPI8_MIN) * 32;
					G = F;
				}
			}
			int H = max(max(diag, Ein), F);
			if constexpr(PW == 2) H = max(max(H, Qu), G);
			if constexpr(CHK){
				const int dU = H - (P & ~31), dV = H - Hu;                                // u' = H(x, y) - H(x-1, y), v' = H(x, y) - H(x, y-1)
				if constexpr(!GEN){
					chi = max(chi, max(dU, dV)); clo = min(clo, min(dU, dV)); cfl = min(cfl, dF);
				} else {
					const int x = t - lane;
					if(x >= 0 && x < xlim){ chi = max(chi, dV); clo = min(clo, dV); }
					if(x >= 1 && x < xlim){ chi = max(chi, dU); clo = min(clo, dU); }
					if(x >= 1 && x < qlen) cfl = min(cfl, dF);                             // (a padding cell is never a block start: all of them lie inside the last block, whose start is a query column)
				}
			}
			const int t1 = H + GOE5;
			if constexpr(PW == 2){
				// the eight facts of the two-piece codes (section 3): A = M or (neither D nor D2 and both insertion chains equal h), D, D2,
				// B = not M and chain 1 equals h; R1 / R2: the chain reaching the next cell opens here; Od1 / Od2: the stored e / q is an opening
				uint32_t nM = (uint32_t)(diag - H), nD = (uint32_t)(Ein - H), nD2 = (uint32_t)(Qu - H);
				const uint32_t nI1 = (uint32_t)(F - H), nI2 = (uint32_t)(G - H);
				if constexpr(GEN){
					if(col0){ nM = (H != cmpM) ? 0x80000000u : 0u; nD = (H != cmpD) ? 0x80000000u : 0u; nD2 = (H != cmpD2) ? 0x80000000u : 0u; }
				}
				const uint32_t xx = (nD & nD2) & ~(nI1 | nI2);                      // (sign bit:) neither D nor D2, both chains
				pM = __builtin_amdgcn_alignbit(pM, nM & ~xx, 31u);                  // not A
				pD = __builtin_amdgcn_alignbit(pD, nD, 31u);
				pD2 = __builtin_amdgcn_alignbit(pD2, nD2, 31u);
				pB = __builtin_amdgcn_alignbit(pB, nM & ~nI1, 31u);                 // B
				const int t2 = H + GQP5, tF = F + GE5, tE = Ein + GE5, tG = G + GP5, tQ = Qu + GP5;
				pR = __builtin_amdgcn_alignbit(pR, (uint32_t)(t1 - tF), 31u);
				pR2 = __builtin_amdgcn_alignbit(pR2, (uint32_t)(t2 - tG), 31u);
				pO = __builtin_amdgcn_alignbit(pO, (uint32_t)(t1 - tE), 31u);
				pO2 = __builtin_amdgcn_alignbit(pO2, (uint32_t)(t2 - tQ), 31u);
				E = max(tE, t1); Q = max(tQ, t2);
				F = max(tF, t1); G = max(tG, t2);
			} else {
			pM = __builtin_amdgcn_alignbit(pM, (uint32_t)(diag - H), 31u);
			pD = __builtin_amdgcn_alignbit(pD, (uint32_t)(Ein - H), 31u);
			if constexpr(GEN){
				if(col0){ pM = (pM & ~1u) | (H != cmpM ? 1u : 0u); pD = (pD & ~1u) | (H != cmpD ? 1u : 0u); }
			}
			if(PW != 0){
				const int tF = F + GE5, tE = Ein + GE5;
				pR = __builtin_amdgcn_alignbit(pR, (uint32_t)(t1 - tF), 31u);        // R: t1 >= tF
				pO = __builtin_amdgcn_alignbit(pO, (uint32_t)(t1 - tE), 31u);        // Od: tE <= t1
				E = max(tE, t1);
				F = max(tF, t1);
			} else F = H + GE5;
			}
			P = (Pi & 31) | H;
			Hd = Hu;
			if constexpr(GEN){
				const int x = t - lane;
				if(x == qlen - 1 && y < tlen){
					if(!ends){ if(y == tlen - 1){ hdr->score = H >> 5; hdr->qe = qlen - 1; hdr->te = tlen - 1; } }
					else {
						// the last query column, row after row: strictly greater replaces (bsalign.h:4023-4032)
						const long long key = ((long long)(H >> 5) << 32) | (long long)(0x7FFFFFFFu - (uint32_t)y);
						if(key > bestc) bestc = key;
					}
				}
				if(LAST && lastrow && x >= 0 && x < qlen) lastH[x] = H >> 5;
			} else {
				if(LAST && lastrow) lastH[t - lane] = H >> 5;
			}
		};
		// 32 steps from step t (a multiple of 32)
		auto body = [&](auto gen, auto last, const int t){
			int ro = t & 255, wo = (t - 64) & 255;                   // ring offsets of the step, kept in VGPRs (a uniform address would be
			asm volatile("" : "+v"(ro), "+v"(wo));                   // moved from an SGPR in front of every LDS instruction)
			const Ent *ib = irng + ro; Ent *ob = orng + wo;
			Ent b0 = ib[0], b1 = ib[1];                             // ring entries are read two steps ahead of their use
#pragma unroll 1
			for(int kk = 0; kk < 4; kk++, ib += 8, ob += 8){
#pragma unroll
				for(int k = 0; k < 8; k++){
					const Ent b = b0;
					b0 = b1;
					b1 = ib[k + 2];
					step(gen, last, t + kk * 8 + k, b, ob + k);
				}
			}
		};
		// 64 steps of this wave from its step t0 (a multiple of 64).  The last body of a block runs to its end (the steps behind the last
		// column change nothing that is kept; the one at nsteps hands the last row's last cell down); a word that holds a step is stored.
		auto run64 = [&](auto last, const int t0){
			for(int t = t0; t < t0 + 64 && t <= nsteps; t += 32){
				if((t & 63) == 0) top(t);
				if(t >= 64 && t + 32 <= qlen - 1) body(std::false_type(), last, t);       // steady state: 0 < x < qlen - 1 on every lane
				else body(std::true_type(), last, t);
				if(t < nsteps) flush(t + 31);
			}
		};
		for(int T = 0; T < Ttot; T += 64){
			const int t0 = T - SYS_LAG * wv;
			if(live && t0 >= 0 && t0 <= nsteps){
				if(ends && lastsb) run64(std::true_type(), t0); else run64(std::false_type(), t0);
			}
			if(NWV > 1) __syncthreads();
		}
		if(wv == NWV - 1 && !lastsb)
			for(int c0 = drained; c0 < xlim; c0 += 64){ const int c = c0 + lane; if(c < xlim) bnd[c] = orng[c & 255]; }
		if constexpr(CHK){
			if(live && y < tlen && (chi > 127 * 32 || clo < -128 * 32 || cfl < BSA_EPI8_MIN * 32)) badl = 1;
#ifdef SYS_DBG
			if(live && y < tlen && (chi > 127 * 32 || clo < -128 * 32 || cfl < BSA_EPI8_MIN * 32) && blockIdx.x < 4u) printf("pair %u row %d (qlen %d tlen %d): max diff %d min diff %d min F - D %d\n", pair, y, qlen, tlen, chi / 32, clo / 32, cfl / 32);
#endif
		}
	}
	if constexpr(CHK){
		if(__any(badl) && lane == 0) atomicOr(&a.status[pair], BSA_ST_TRACE);       // the traceback kernel leaves a flagged pair alone
	}
	if(ends){
		// overlap / extend: the best cell of the last query column (first row on ties), replaced by row_max of the last target row if that
		// is strictly greater (bsalign.h:4034-4048).  row_max (bsalign.h:3213-3329) in the reference's own striping: the best cell of every
		// running block of W = band / 16 cells, the first one on ties, blocks compared in the order of its register reduction.  Band cells
		// beyond the query end are never the maximum (each is smaller than its left neighbour), so the query's cells are all that matters.
		for(int o = 32; o > 0; o >>= 1){ const long long ok = __shfl_xor(bestc, o); if(ok > bestc) bestc = ok; }
		__builtin_amdgcn_s_waitcnt(0);
		if(NWV > 1){
			if(lane == 0) bestw[wv] = bestc;
			__syncthreads();
			if(wv != 0) return;
#pragma unroll
			for(int w = 1; w < NWV; w++) if(bestw[w] > bestc) bestc = bestw[w];
		}
		const int Wc = a.ref_bw ? (int)(a.ref_bw / 16u) : ((qlen + 15) / 16 * 16) / 16;
		const int j = lane >> 2, sub = lane & 3, qw = (Wc + 3) / 4;
		const int xb = j * Wc + sub * qw, xe = min(min(xb + qw, (j + 1) * Wc), qlen);
		long long bk = (long long)0x8000000000000000ull;
		for(int x = xb; x < xe; x++){
			const long long key = ((long long)*(const volatile int32_t*)&lastH[x] << 32) | (long long)(0x7FFFFFFFu - (uint32_t)x);
			if(key > bk) bk = key;
		}
		for(int o = 1; o <= 2; o <<= 1){ const long long ok = __shfl_xor(bk, o); if(ok > bk) bk = ok; }
		int bs = 0, bp = 0; bool any = false;
		for(int kk = 0; kk < 16; kk++){
			const int jj = (kk & 3) * 4 + (kk >> 2);
			const long long key = __shfl(bk, jj * 4);
			if(key == (long long)0x8000000000000000ull) continue;                  // (a block beyond the query end)
			const int mx = (int)(key >> 32), px = (int)(0x7FFFFFFFu - (uint32_t)(key & 0xFFFFFFFFll));
			if(!any || mx > bs){ bs = mx; bp = px; any = true; }
		}
		if(lane == 0){
			int sc = (int)(bestc >> 32), qe = qlen - 1, te = (int)(0x7FFFFFFFu - (uint32_t)(bestc & 0xFFFFFFFFll));
			if(any && bs > sc){ sc = bs; qe = bp; te = tlen - 1; }
			hdr->score = sc; hdr->qe = qe; hdr->te = te;
		}
	}
}

// ---- traceback: backcal's decisions read off the codes (oracle: backcal_codes; bsalign.h:3704-3852) ----
// WAVE = true (the default): one pair per wave.  Every lane carries the walk's state; a code tile (64 rows x 32 wavefront steps, one
// contiguous kilobyte) is fetched by the whole wave into one of four LDS slots and the tile to its left -- where a match / mismatch run
// goes next -- is requested at the same time and waits in registers, so a step costs an LDS read, not a trip to HBM (the walk of a
// 10 kbp pair: 20 k steps); the bases for the match / mismatch count come from 256-byte windows in LDS.  WAVE = false: a pair per lane.
template<int PW, bool WAVE>
__global__ void __launch_bounds__(64) k_align8_trace_sys(const Align8Args a, bsa_result_t *out, uint32_t *cig_cnt, const uint64_t *slot_end){
	constexpr int TU = (PW == 2) ? 128 : 64;              // 16-byte words of a code tile (two pieces: {A, D, D2, B} then {R1, R2, Od1, Od2})
	__shared__ uint4 tl[WAVE ? 4 : 1][WAVE ? TU : 1];
	__shared__ uint32_t qwn[WAVE ? 64 : 1], twn[WAVE ? 64 : 1];
	const int lane = threadIdx.x;
	const bool wr = !WAVE || lane == 0;                   // who writes results
	const uint32_t g = WAVE ? blockIdx.x : blockIdx.x * 64u + threadIdx.x;
	if(g >= a.count) return;
	const uint32_t ppos = a.first + g;
	const uint32_t pair = a.order[ppos];
	bsa_result_t rs;
	rs.score = 0; rs.qb = rs.qe = rs.tb = rs.te = 0; rs.mat = rs.mis = rs.ins = rs.del = rs.aln = 0;
	if(a.status[pair] != 0u){ if(wr){ out[pair] = rs; cig_cnt[ppos] = 0; } return; }
	const int qlen = (int)a.qlen[pair], tlen_ = (int)a.tlen[pair];
	const uint8_t *qseq = a.qst + a.qpoff[pair], *tseq = a.tst + a.tpoff[pair];
	const uint8_t *slot = a.rows + a.slot_off[ppos];
	const SysHdr *hdr = (const SysHdr*)slot;
	const uint8_t *codes = slot + bsa_sys_codes_off((uint32_t)qlen, PW);
	const int NW = (int)bsa_sys_words((uint32_t)qlen);
	uint32_t *cig_end = (uint32_t*)(a.rows + slot_end[ppos]);
	uint32_t ncig = 0;
	auto cig_push = [&](uint32_t w){ ncig++; if(wr) *(cig_end - ncig) = w; };
	auto cig_add = [&](uint32_t cg, uint32_t op, uint32_t sz) -> uint32_t {   // bsalign.h:409-417
		if(op == (cg & 0xf)) return cg + (sz << 4);
		if(cg) cig_push(cg);
		return (sz << 4) | op;
	};
	// the facts of cell (x, r): tile r >> 6, wavefront step t = x + (r & 63), word t >> 5, bit 31 - (t & 31)
	const uint4 *gt = (const uint4*)codes;
	int tg0 = -1, tg1 = -1, tg2 = -1, tg3 = -1, pf_id = -1;          // the tiles in the LDS slots (slot = word index mod 4), the tile waiting in `pf`
	uint4 pf = make_uint4(0, 0, 0, 0), pf2 = pf;
	auto getw = [&](int r, int tw, int half = 0) -> uint4 {
		const int id = (r >> 6) * NW + tw;
		if constexpr(!WAVE) return gt[(size_t)id * TU + half * 64 + (r & 63)];
		else {
			const int sl = tw & 3;
			const int have = sl == 0 ? tg0 : sl == 1 ? tg1 : sl == 2 ? tg2 : tg3;
			if(have != id){
				if(pf_id == id){ tl[sl][lane] = pf; if(PW == 2) tl[sl][64 + lane] = pf2; }
				else { tl[sl][lane] = gt[(size_t)id * TU + lane]; if(PW == 2) tl[sl][64 + lane] = gt[(size_t)id * TU + 64 + lane]; }
				if(sl == 0) tg0 = id; else if(sl == 1) tg1 = id; else if(sl == 2) tg2 = id; else tg3 = id;
				const int sp = (tw - 1) & 3, hp = sp == 0 ? tg0 : sp == 1 ? tg1 : sp == 2 ? tg2 : tg3;
				if(tw >= 1 && hp != id - 1){ pf = gt[(size_t)(id - 1) * TU + lane]; if(PW == 2) pf2 = gt[(size_t)(id - 1) * TU + 64 + lane]; pf_id = id - 1; }
			}
			return tl[sl][(PW == 2 ? half * 64 : 0) + (r & 63)];
		}
	};
	auto bit = [&](int r, int x, int pl) -> bool {              // plane pl of the cell's facts (two pieces: 0..3 first word, 4..7 second)
		const int t = x + (r & 63);
		const uint4 v = getw(r, t >> 5, pl >> 2);
		const uint32_t wd = (pl & 3) == 0 ? v.x : (pl & 3) == 1 ? v.y : (pl & 3) == 2 ? v.z : v.w;
		return (wd >> (31 - (t & 31))) & 1u;
	};
	// the two bases of a match / mismatch step
	int qb0 = 1 << 30, tb0 = 1 << 30;                            // first base in the LDS windows (WAVE)
	auto getq = [&](int x) -> uint32_t {
		if constexpr(!WAVE) return qseq[x];
		else {
			if(x < qb0 || x >= qb0 + 256){ qb0 = max(0, x - 252) & ~3; qwn[lane] = (qb0 + 4 * lane < qlen) ? *(const uint32_t*)(qseq + qb0 + 4 * lane) : 0u; }      // (the staged sequences are padded past their ends)
			return (qwn[(x - qb0) >> 2] >> (8 * ((x - qb0) & 3))) & 0xFFu;
		}
	};
	auto gett = [&](int y) -> uint32_t {
		if constexpr(!WAVE) return tseq[y];
		else {
			if(y < tb0 || y >= tb0 + 256){ tb0 = max(0, y - 252) & ~3; twn[lane] = (tb0 + 4 * lane < tlen_) ? *(const uint32_t*)(tseq + tb0 + 4 * lane) : 0u; }
			return (twn[(y - tb0) >> 2] >> (8 * ((y - tb0) & 3))) & 0xFFu;
		}
	};
	bool bad = false;
	rs.score = hdr->score;
	if(rs.score == (int)0x80000000u) bad = true;
	rs.qe = hdr->qe; rs.te = hdr->te;                     // (global: the last cell; overlap / extend: the end cell the forward pass found)
	rs.qb = rs.qe; rs.qe++;
	rs.tb = rs.te; rs.te++;
	int prior_match = 0;
	uint32_t cg = 0;
	while(!bad){
		if(rs.qb < 0 || rs.tb < 0) break;
		const int lq = rs.tb & 63, tq = rs.qb + lq;
		const uint4 cw = getw(rs.tb, tq >> 5);
		const uint32_t qc = getq(rs.qb), tc = gett(rs.tb);     // (pair per lane: requested with the code word, one memory latency per step, not two)
		const uint32_t sh = 31u - ((uint32_t)tq & 31u);
		int bt, dpl = 3, chains = 1;                          // dpl: the plane whose bit ends a deletion run; chains: which insertion chains equal h (two pieces)
		if constexpr(PW == 2){
			// backcal_cell with two pieces (bsalign.h:3679-3701) off the folded facts: D or D2 set: A is M; else (A, B) = (1, 0) M, (1, 1) both chains,
			// (0, 1) chain 1 only, (0, 0) chain 2 only
			const bool fA = (cw.x >> sh) & 1u, fD = (cw.y >> sh) & 1u, fD2 = (cw.z >> sh) & 1u, fB = (cw.w >> sh) & 1u;
			const bool fM = (fD || fD2) ? fA : (fA && !fB);
			const int d = fD ? 1 : fD2 ? 2 : 0;
			if(prior_match) bt = fM ? 0 : d ? 2 : 1;
			else bt = d ? 2 : fM ? 0 : 1;
			dpl = (d == 2) ? 7 : 6;
			chains = fA ? (fB ? 3 : 0) : (fB ? 1 : 2);
		} else {
			const bool fM = (cw.x >> sh) & 1u, fD = (cw.y >> sh) & 1u;
			if(prior_match) bt = fM ? 0 : fD ? 2 : 1;             // backcal_cell (bsalign.h:3679-3699): the order of the tests depends on prior_match
			else bt = fD ? 2 : fM ? 0 : 1;
		}
		prior_match = 1;
		if(bt == 0){
			if(qc == tc) rs.mat++; else rs.mis++;
			rs.qb--; rs.tb--; rs.aln++;
			cg = cig_add(cg, 0, 1);
		} else if(bt == 1){
			if(rs.qb <= 0){ cg = cig_add(cg, 1, 1); rs.qb--; rs.ins++; rs.aln++; }
			else {
				// the nearest cell to the left at which an insertion reaching its right neighbour opens (bsalign.h:3798-3814)
				int sz = 0;
				const uint32_t lmask = ~((2u << sh) - 1u);                              // the cells left of qb in this word
				auto rword = [&](int tw_) -> uint32_t {                                 // the R bits a scan may stop at
					if constexpr(PW == 2){ const uint4 v = getw(rs.tb, tw_, 1); return ((chains & 1) ? v.x : 0u) | ((chains & 2) ? v.y : 0u); }
					else return getw(rs.tb, tw_).z;
				};
				uint32_t w = rword(tq >> 5) & lmask;
				int tw = tq >> 5;
				for(;;){
					if(w){ const int c = tw * 32 + (31 - (int)__builtin_ctz(w)) - lq; if(c >= 0) sz = rs.qb - c; break; }      // (steps left of column 0 hold no cell)
					if(--tw < 0) break;
					w = rword(tw);
				}
				if constexpr(PW == 2){
					if(sz){
						// the reference tests H(x - sz) + max(cost1(sz), cost2(sz)) == H(x): the chain that is tight here must also be the one with the
						// larger cost at this length (backcal_codes); else its scan finds no length
						const bool h1 = (chains & 1) && bit(rs.tb, rs.qb - sz, 4), h2 = (chains & 2) && bit(rs.tb, rs.qb - sz, 5);
						const int c1 = (int)a.gapo1 + sz * (int)a.gape1, c2 = (int)a.gapo2 + sz * (int)a.gape2;
						if(!((h1 && c1 >= c2) || (h2 && c2 >= c1))) sz = 0;
					}
				}
				if(sz == 0){ bad = true; break; }                                       // the reference's scan finds no length either
				cg = cig_add(cg, 1, (uint32_t)sz);
				rs.qb -= sz; rs.ins += sz; rs.aln += sz;
			}
		} else {
			// deletion run: up the column until a row whose stored e is a fresh opening (bsalign.h:3730-3744)
			int len = 1;
			if(PW == 2 && rs.qb == 0){ bad = true; break; }          // (two pieces, column 0: the reference's own scan does not terminate there -- literal path)
			for(;;){
				const int r = rs.tb - len;
				if(r < -1){ bad = true; break; }
				if(r == -1){ if(PW != 0) bad = true; break; }       // linear gaps: an ordinary move; affine: the reference compares real scores there -- literal path
				if(bit(r, rs.qb, dpl)) break;
				len++;
			}
			if(bad) break;
			cg = cig_add(cg, 2, (uint32_t)len);
			rs.del += len; rs.aln += len;
			rs.tb -= len;
		}
	}
	if(!bad){
		if((a.mode & 3) == BSA_MODE_OVERLAP){ if(cg) cig_push(cg); }       // overlap: the alignment simply starts where the walk left the matrix
		else {
			uint32_t op = 0, sz = 0;          // global / extend: what is left at the top becomes a leading I / D (bsalign.h:3827-3842)
			if(rs.qb >= 0){ op = 1; sz = (uint32_t)rs.qb + 1u; rs.ins += (int)sz; rs.qb = -1; }
			else if(rs.tb >= 0){ op = 2; sz = (uint32_t)rs.tb + 1u; rs.del += (int)sz; rs.tb = -1; }
			rs.aln += (int)sz;
			cg = cig_add(cg, op, sz);
			if(cg) cig_push(cg);
		}
		rs.qb++; rs.tb++;
	} else {
		if(wr) atomicOr(&a.status[pair], BSA_ST_TRACE);
		rs.qb = rs.qe = rs.tb = rs.te = 0; rs.mat = rs.mis = rs.ins = rs.del = rs.aln = 0;
		ncig = 0;
	}
	if(wr){ out[pair] = rs; cig_cnt[ppos] = ncig; }
}

hipError_t bsa_launch_align8_fwd_sys(const Align8Args &a, int pw, uint32_t max_qlen, hipStream_t st){
	if(a.count == 0) return hipSuccess;
	// four waves per pair where a wave per pair would leave the chip short of waves (BSA_ALIGN8_SYS_WAVES=1 / 2 / 4 / 8 overrides); a wave per
	// pair for short queries, where the 576 steps the fourth wave lags by would not be small against the query, and for large batches
	// (2000 bp x 30000 pairs: 68 ms against 100; 10 kbp x 4096: 290 against 237)
	int nwv = (max_qlen >= 1024u && a.count < 5120u) ? 4 : 1;          // (87 VGPRs: five waves per SIMD, 5120 on the chip -- a wave per pair fills it from there on)
	if(const char *e = bsa_env("BSA_ALIGN8_SYS_WAVES")){ const int v = atoi(e); if(v == 1 || v == 2 || v == 4 || v == 8) nwv = v; }
#define SYS_LAUNCH_(N_, C_) do { if(pw == 0) hipLaunchKernelGGL((k_align8_fwd_sys<0, N_, C_>), dim3(a.count), dim3(64 * N_), 0, st, a); \
		else if(pw == 1) hipLaunchKernelGGL((k_align8_fwd_sys<1, N_, C_>), dim3(a.count), dim3(64 * N_), 0, st, a); \
		else hipLaunchKernelGGL((k_align8_fwd_sys<2, N_, C_>), dim3(a.count), dim3(64 * N_), 0, st, a); } while(0)
#define SYS_LAUNCH(N_) do { if(a.sys_chk) SYS_LAUNCH_(N_, true); else SYS_LAUNCH_(N_, false); } while(0)
	if(nwv == 8) SYS_LAUNCH(8); else if(nwv == 4) SYS_LAUNCH(4); else if(nwv == 2) SYS_LAUNCH(2); else SYS_LAUNCH(1);
#undef SYS_LAUNCH_
#undef SYS_LAUNCH
	return hipGetLastError();
}

hipError_t bsa_launch_align8_trace_sys(const Align8Args &a, int pw, bsa_result_t *out, uint32_t *cig_cnt, const uint64_t *slot_end, hipStream_t st){
	if(a.count == 0) return hipSuccess;
	bool wave = a.count < 16384u;            // a pair per wave while that leaves no SIMD idle, a pair per lane for large batches (1 kbp x 100 k pairs: 3.6 ms against 16.9)
	if(const char *e = bsa_env("BSA_ALIGN8_SYS_TRACE")) wave = e[0] != 'l';            // "lane" / "wave"
	if(wave){
		if(pw == 0) hipLaunchKernelGGL((k_align8_trace_sys<0, true>), dim3(a.count), dim3(64), 0, st, a, out, cig_cnt, slot_end);
		else if(pw == 1) hipLaunchKernelGGL((k_align8_trace_sys<1, true>), dim3(a.count), dim3(64), 0, st, a, out, cig_cnt, slot_end);
		else hipLaunchKernelGGL((k_align8_trace_sys<2, true>), dim3(a.count), dim3(64), 0, st, a, out, cig_cnt, slot_end);
	} else {
		const dim3 grid((a.count + 63u) / 64u);
		if(pw == 0) hipLaunchKernelGGL((k_align8_trace_sys<0, false>), grid, dim3(64), 0, st, a, out, cig_cnt, slot_end);
		else if(pw == 1) hipLaunchKernelGGL((k_align8_trace_sys<1, false>), grid, dim3(64), 0, st, a, out, cig_cnt, slot_end);
		else hipLaunchKernelGGL((k_align8_trace_sys<2, false>), grid, dim3(64), 0, st, a, out, cig_cnt, slot_end);
	}
	return hipGetLastError();
}
