// bsa_edit.hip -- 2-bit bit-parallel edit-distance pairwise DP on gfx950 (MI355X).
//
// Replaces striped_seqedit_pairwise and its helpers (/root/reference/bsalign.h):
//   _set_query_prof :612   _row_init :653   _row_movx :658   _row_cal :766   _rowmin :813
//   _backtrace :965        _pairwise :1046 (band rules :1055-1067, fixed diagonal band :1112-1114,
//   score by popcount :1189-1203)
//
// The reference keeps per row two 64-bit planes of u(p) = H(p,y) - H(p-1,y) in {-1,0,+1} (plane0 bit <=>
// u = -1, plane1 bit <=> u = +1) in Farrar-striped order and iterates the row to a fix-point.  That
// recurrence is exact integer arithmetic (no saturation), so any correct evaluation yields the same
// planes; here the band lives in NATURAL bit order (band position p = bit p%64 of word p/64) and one row
// step is the Myers/Hyyro block update with the carry (horizontal delta) chained across the W words:
//     Xv = Eq | Mv;  Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;  Ph = Mv | ~(Xh | Pv);  Mh = Pv & Xh;  ...
// with Pv = plane1 (u = +1), Mv = plane0 (u = -1), Ph/Mh = the reference's v.  Boundary rules follow the
// reference exactly: left of the band v = +1 (0 in overlap mode, bsalign.h:770), cells shifted in on the
// right get u = +1 (:683-718), sbeg tracks H at the band start (:667-676).
//
// Mapping: one pair per lane (64 pairs per wave), all state in VGPRs, no cross-lane traffic; the kernel is
// HBM-bound: per row and pair it writes the two planes (W*16 bytes = 2 bits per band cell), nothing else.
#include "bsa_common.h"
#include "bsa_dpp.h"

typedef uint64_t u64;

static __device__ __forceinline__ u64 fsr(u64 lo, u64 hi, uint32_t m){   // (hi:lo) >> m, low 64 bits, m in [0,64]
	if(m == 0) return lo;
	if(m >= 64) return hi;
	return (lo >> m) | (hi << (64 - m));
}
static __device__ __forceinline__ u64 lowmask(uint32_t n){ return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }
// row records: row r (r = 0 is the initial row, r = i+1 the row of target base i): [plane0: NW u64][plane1: NW u64]
// TRACK (overlap / extend): also follow H at the last query column from row to row (its minimum picks the end cell)
template<int NW, bool TRACK>
__global__ void __launch_bounds__(64) k_edit_fwd(const EditArgs a, uint32_t lanes){
	constexpr uint32_t BW = NW * 64;
	const uint32_t g = blockIdx.x * lanes + threadIdx.x;      // `lanes` pairs per wave (bsa_launch_edit_fwd)
	if(threadIdx.x >= lanes || g >= a.count) return;
	const uint32_t ppos = a.first + g, pair = a.order[ppos];
	if(a.status[pair] != 0u) return;
	const uint32_t qlen = a.qlen[pair], tlen = a.tlen[pair];
	const u64 *Q0m = a.qbits + a.qboff[pair];
	const u64 *Q1m = Q0m + a.qwords[pair];
	const uint8_t *tp = a.tst + a.tpoff[pair];
	u64 *rows = (u64*)(a.rows + a.slot_off[ppos]);
	const int type = a.mode & 3;
	const bool overlap = type == BSA_MODE_OVERLAP;
	const uint32_t qround = (qlen + 63u) / 64u * 64u;

	u64 Pv[NW], Mv[NW], Q0[NW], Q1[NW];
	u64 l0c, l0n, l1c, l1n;      // look-ahead of the query planes: next position to shift in = lword*64 + lbit
	uint32_t lword, lbit;
	auto load_window = [&](uint32_t rb){
		const uint32_t w0 = rb >> 6, sh = rb & 63u;
#pragma unroll
		for(int k = 0; k < NW; k++){
			Q0[k] = fsr(*(Q0m + w0 + k), *(Q0m + w0 + k + 1), sh);
			Q1[k] = fsr(*(Q1m + w0 + k), *(Q1m + w0 + k + 1), sh);
		}
		const uint32_t lp = rb + BW;
		lword = lp >> 6; lbit = lp & 63u;
		l0c = *(Q0m + lword); l0n = *(Q0m + lword + 1); l1c = *(Q1m + lword); l1n = *(Q1m + lword + 1);
	};
#pragma unroll
	for(int k = 0; k < NW; k++){ Pv[k] = ~0ull; Mv[k] = 0ull; rows[k] = 0ull; rows[NW + k] = ~0ull; }   // row_init (:653-656)
	load_window(0);
	int sbeg = 0;
	uint32_t rb0 = 0;
	// overlap / extend (the band never moves): H(qlen-1, row), its minimum over the rows and the first row that has it
	const uint32_t lastw = (qlen - 1u) >> 6, lastb = (qlen - 1u) & 63u;
	int slast = (int)qlen, smin = 0x7FFFFFFF, ry = (int)tlen - 1;
	// i*qlen/tlen kept incrementally: quo = floor(i*qlen/tlen), rem = (i*qlen) % tlen
	u64 quo = 0, rem = 0;
	const u64 qstep = qlen / tlen, rstep = qlen % tlen;
	u64 tw = 0;   // 8 target bases at a time
	for(uint32_t i = 0; i < tlen; i++){
		if((i & 7u) == 0){ tw = 0; __builtin_memcpy(&tw, tp + i, 8); }   // staged with >= 8 bytes of padding
		const uint32_t tb = (uint32_t)(tw >> (8u * (i & 7u))) & 3u;
		uint32_t rb1;
		if(type != BSA_MODE_GLOBAL) rb1 = 0;
		else {      // fixed diagonal band (:1112-1114)
			uint32_t c = (uint32_t)quo;
			c = (c < BW / 2) ? 0u : c - BW / 2;
			rb1 = (c + BW > qround) ? qround - BW : c;
		}
		uint32_t movx = rb1 - rb0;
		// ---- row_movx (:658-721)
		if(overlap) sbeg = 0;
		else if(movx >= BW){
			int s = 0;
#pragma unroll
			for(int k = 0; k < NW; k++){ s += __popcll(Pv[k]) - __popcll(Mv[k]); Pv[k] = ~0ull; Mv[k] = 0ull; }
			sbeg += s + 1;
			load_window(rb1);
		} else if(movx && movx < 64u){
			// the usual band step (a few cells): one funnel shift per word with 0 < m < 64, no special cases
			const uint32_t m = movx, r = 64u - movx;
			const u64 mk = (1ull << m) - 1ull;
			sbeg += __popcll(Pv[0] & mk) - __popcll(Mv[0] & mk) + 1;
			const u64 n0 = fsr(l0c, l0n, lbit), n1 = fsr(l1c, l1n, lbit);
#pragma unroll
			for(int k = 0; k < NW; k++){
				Pv[k] = (Pv[k] >> m) | (((k + 1 < NW) ? Pv[k + 1] : ~0ull) << r);
				Mv[k] = (Mv[k] >> m) | (((k + 1 < NW) ? Mv[k + 1] : 0ull) << r);
				Q0[k] = (Q0[k] >> m) | (((k + 1 < NW) ? Q0[k + 1] : n0) << r);
				Q1[k] = (Q1[k] >> m) | (((k + 1 < NW) ? Q1[k + 1] : n1) << r);
			}
			lbit += m;
			if(lbit >= 64u){
				lbit -= 64u; lword++;
				l0c = l0n; l1c = l1n;
				l0n = *(Q0m + lword + 1); l1n = *(Q1m + lword + 1);
			}
		} else {
			while(movx){
				const uint32_t m = movx < 64u ? movx : 64u;
				const u64 mk = lowmask(m);
				sbeg += __popcll(Pv[0] & mk) - __popcll(Mv[0] & mk);
				const u64 n0 = fsr(l0c, l0n, lbit), n1 = fsr(l1c, l1n, lbit);
#pragma unroll
				for(int k = 0; k < NW; k++){
					Pv[k] = fsr(Pv[k], (k + 1 < NW) ? Pv[k + 1] : ~0ull, m);
					Mv[k] = fsr(Mv[k], (k + 1 < NW) ? Mv[k + 1] : 0ull, m);
					Q0[k] = fsr(Q0[k], (k + 1 < NW) ? Q0[k + 1] : n0, m);
					Q1[k] = fsr(Q1[k], (k + 1 < NW) ? Q1[k + 1] : n1, m);
				}
				lbit += m;
				if(lbit >= 64u){
					lbit -= 64u; lword++;
					l0c = l0n; l1c = l1n;
					l0n = *(Q0m + lword + 1); l1n = *(Q1m + lword + 1);
				}
				movx -= m;
			}
			sbeg++;
		}
		// ---- row_cal (:766-810) as a chained Myers block update
		const u64 x0 = (tb & 1u) ? 0ull : ~0ull, x1 = (tb & 2u) ? 0ull : ~0ull;
		const uint32_t nvalid = (rb1 < qlen) ? qlen - rb1 : 0u;      // band cells that are real query columns
		int hin = overlap ? 0 : 1;
#pragma unroll
		for(int k = 0; k < NW; k++){
			u64 Eq = (Q0[k] ^ x0) & (Q1[k] ^ x1);
			if(nvalid < BW){ const uint32_t lo = (uint32_t)k * 64u; Eq &= (nvalid > lo) ? lowmask(nvalid - lo) : 0ull; }
			const u64 pv = Pv[k], mv = Mv[k];
			const u64 hneg = (hin < 0) ? 1ull : 0ull, hpos = (hin > 0) ? 1ull : 0ull;
			const u64 Xv = Eq | mv;
			const u64 Eq2 = Eq | hneg;
			const u64 Xh = (((Eq2 & pv) + pv) ^ pv) | Eq2;
			u64 Ph = mv | ~(Xh | pv);
			u64 Mh = pv & Xh;
			if(TRACK && (uint32_t)k == lastw) slast += (int)((Ph >> lastb) & 1ull) - (int)((Mh >> lastb) & 1ull);
			hin = (int)(Ph >> 63) - (int)(Mh >> 63);
			Ph = (Ph << 1) | hpos;
			Mh = (Mh << 1) | hneg;
			Pv[k] = Mh | ~(Xv | Ph);
			Mv[k] = Ph & Xv;
		}
		u64 *rp = rows + (size_t)(i + 1) * (2 * NW);
#pragma unroll
		for(int k = 0; k < NW; k++){ rp[k] = Mv[k]; rp[NW + k] = Pv[k]; }
		// score at the last query column (overlap / extend, :1124-1139): slast followed the row-to-row delta there
		if(TRACK && slast < smin){ smin = slast; ry = (int)i; }
		rb0 = rb1;
		quo += qstep; rem += rstep;
		if(rem >= tlen){ rem -= tlen; quo++; }
	}
	// H at the band start of the last row; the traceback kernel derives the scores from it
	a.fwd_sbeg[ppos] = sbeg;
	a.fwd_smin[ppos] = smin; a.fwd_ry[ppos] = ry;
}


// ---- any band width (multiple of 64): the generic forward kernel ------------------------------------------------
// Same recurrence and row records as k_edit_fwd, but the band state is not held in registers: the previous row is
// read back from its row record (it has to be written anyway), shifted by the band step on the fly, and the query
// planes are taken straight from the staged bit planes at the row's band offset.  One pair per lane, NW a run-time
// value; used when NW > 16, i.e. for the full-width bands of overlap / extend mode and of `bandwidth 0` on queries
// longer than 1024 bp (bsalign.h:1055-1067).  Streaming loads and stores, four words in and two out per 64 cells.
__global__ void __launch_bounds__(64) k_edit_fwd_gen(const EditArgs a, uint32_t lanes){
	const uint32_t g = blockIdx.x * lanes + threadIdx.x;
	if(threadIdx.x >= lanes || g >= a.count) return;
	const uint32_t ppos = a.first + g, pair = a.order[ppos];
	if(a.status[pair] != 0u) return;
	const uint32_t qlen = a.qlen[pair], tlen = a.tlen[pair];
	const uint32_t BW = a.bw ? a.bw : bsa_edit_bw_eff(qlen, tlen, a.mode & 3, a.bandwidth), NW = BW / 64u;
	if(a.wide != 0u && BW == (qlen + 63u) / 64u * 64u) return;      // static band: done by k_edit_fwd_wide in the same launch
	const u64 *Q0m = a.qbits + a.qboff[pair];
	const u64 *Q1m = Q0m + a.qwords[pair];
	const uint8_t *tp = a.tst + a.tpoff[pair];
	u64 *rows = (u64*)(a.rows + a.slot_off[ppos]);
	const int type = a.mode & 3;
	const bool overlap = type == BSA_MODE_OVERLAP;
	const uint32_t qround = (qlen + 63u) / 64u * 64u;
	for(uint32_t k = 0; k < NW; k++){ rows[k] = 0ull; rows[NW + k] = ~0ull; }      // row_init (:653-656)
	int sbeg = 0;
	uint32_t rb0 = 0;
	const uint32_t lastw = (qlen - 1u) >> 6, lastb = (qlen - 1u) & 63u;
	int slast = (int)qlen, smin = 0x7FFFFFFF, ry = (int)tlen - 1;
	u64 quo = 0, rem = 0;
	const u64 qstep = qlen / tlen, rstep = qlen % tlen;
	for(uint32_t i = 0; i < tlen; i++){
		const uint32_t tb = (uint32_t)tp[i] & 3u;
		uint32_t rb1;
		if(type != BSA_MODE_GLOBAL) rb1 = 0;
		else {
			uint32_t c = (uint32_t)quo;
			c = (c < BW / 2) ? 0u : c - BW / 2;
			rb1 = (c + BW > qround) ? qround - BW : c;
		}
		const uint32_t movx = rb1 - rb0;
		const u64 *pm = rows + (size_t)i * (2 * NW), *pp = pm + NW;      // previous row: Mv words, Pv words
		u64 *nm = rows + (size_t)(i + 1) * (2 * NW), *np = nm + NW;
		const uint32_t ws = movx >> 6, bs = movx & 63u;
		// ---- row_movx (:658-721): H at the new band start
		if(overlap) sbeg = 0;
		else {
			int s = 0;
			const uint32_t full = ws < NW ? ws : NW;
			for(uint32_t w = 0; w < full; w++) s += __popcll(pp[w]) - __popcll(pm[w]);
			if(ws < NW && bs){ const u64 mk = lowmask(bs); s += __popcll(pp[ws] & mk) - __popcll(pm[ws] & mk); }
			sbeg += s + 1;
		}
		auto prevP = [&](uint32_t idx) -> u64 { return idx < NW ? pp[idx] : ~0ull; };   // cells shifted in on the right: u = +1
		auto prevM = [&](uint32_t idx) -> u64 { return idx < NW ? pm[idx] : 0ull; };
		// ---- row_cal (:766-810), word by word
		const u64 x0 = (tb & 1u) ? 0ull : ~0ull, x1 = (tb & 2u) ? 0ull : ~0ull;
		const uint32_t nvalid = (rb1 < qlen) ? qlen - rb1 : 0u;
		const uint32_t w0 = rb1 >> 6, sh = rb1 & 63u;
		int hin = overlap ? 0 : 1;
		u64 plo = prevP(ws), mlo = prevM(ws), q0lo = Q0m[w0], q1lo = Q1m[w0];
		for(uint32_t k = 0; k < NW; k++){
			const u64 phi = prevP(ws + k + 1), mhi = prevM(ws + k + 1), q0hi = Q0m[w0 + k + 1], q1hi = Q1m[w0 + k + 1];
			const u64 pv = fsr(plo, phi, bs), mv = fsr(mlo, mhi, bs);
			u64 Eq = (fsr(q0lo, q0hi, sh) ^ x0) & (fsr(q1lo, q1hi, sh) ^ x1);
			if(nvalid < BW){ const uint32_t lo = k * 64u; Eq &= (nvalid > lo) ? lowmask(nvalid - lo) : 0ull; }
			const u64 hneg = (hin < 0) ? 1ull : 0ull, hpos = (hin > 0) ? 1ull : 0ull;
			const u64 Xv = Eq | mv;
			const u64 Eq2 = Eq | hneg;
			const u64 Xh = (((Eq2 & pv) + pv) ^ pv) | Eq2;
			u64 Ph = mv | ~(Xh | pv);
			u64 Mh = pv & Xh;
			if(type != BSA_MODE_GLOBAL && k == lastw) slast += (int)((Ph >> lastb) & 1ull) - (int)((Mh >> lastb) & 1ull);
			hin = (int)(Ph >> 63) - (int)(Mh >> 63);
			Ph = (Ph << 1) | hpos;
			Mh = (Mh << 1) | hneg;
			np[k] = Mh | ~(Xv | Ph);
			nm[k] = Ph & Xv;
			plo = phi; mlo = mhi; q0lo = q0hi; q1lo = q1hi;
		}
		if(type != BSA_MODE_GLOBAL && slast < smin){ smin = slast; ry = (int)i; }
		rb0 = rb1;
		quo += qstep; rem += rstep;
		if(rem >= tlen){ rem -= tlen; quo++; }
	}
	a.fwd_sbeg[ppos] = sbeg;
	a.fwd_smin[ppos] = smin; a.fwd_ry[ppos] = ry;
}

// ---- wide static bands: one pair per WAVE -----------------------------------------------------------------------
// Overlap / extend mode and `bandwidth 0` make the band the whole (rounded) query, so it never moves: no row_movx, the
// query planes of a word are the same for every row and the whole band state fits the wave's registers -- lane l owns
// words l*WPL .. l*WPL+WPL-1 (up to 64*WPL*64 = 32768 query columns for WPL = 8).  What is left of the serial chain
// across words is one number per word, the horizontal delta hin in {-1, 0, +1} entering it, and the block update only
// looks at its sign: Xh takes (hin < 0) as bit 0 of Eq, (hin > 0) is just shifted into Ph afterwards.  So every lane
// evaluates its words for both cases (a negative delta entering or not), the wave resolves the chain in scalar code
//     neg[k+1] = neg[k] ? B[k] : A[k],   A[k] = hout(k | not negative) < 0,  B[k] = hout(k | negative) < 0
// as the carry chain of a 64-bit addition (A generates, B propagates; lowering the delta entering a block never raises
// the one leaving it, so A implies B -- if a word ever violates that the chain is walked bit by bit instead), and each
// lane then keeps the variant that was right.  Row records are identical to the other forward kernels' (natural word
// order), written as one coalesced 8*WPL-byte store per lane and plane.
static __device__ __forceinline__ uint32_t mask_pick(uint32_t if0, uint32_t if1, u64 mask){   // per lane: bit `lane` of mask ? if1 : if0
	uint32_t r;
	asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(if0), "v"(if1), "s"(mask));
	return r;
}
static __device__ __forceinline__ u64 mask_pick64(u64 if0, u64 if1, u64 mask){
	const uint32_t lo = mask_pick((uint32_t)if0, (uint32_t)if1, mask), hi = mask_pick((uint32_t)(if0 >> 32), (uint32_t)(if1 >> 32), mask);
	return (u64)hi << 32 | lo;
}
static __device__ __forceinline__ u64 uniform64(u64 x){      // tell the compiler the value is wave-uniform (it is): keep it in SGPRs
	return (u64)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(x >> 32)) << 32 | (u64)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)x);
}
// carries of the chain neg[k+1] = neg[k] ? B[k] : A[k] with neg[0] = 0; bit k of the result = neg[k]
static __device__ __forceinline__ u64 chain_neg(u64 A, u64 B){
	if(__builtin_expect((A & ~B) == 0ull, 1)) return (A + B) ^ A ^ B;
	u64 C = 0, c = 0;
	for(int k = 0; k < 63; k++){ c = c ? (B >> k) & 1ull : (A >> k) & 1ull; C |= c << (k + 1); }
	return C;
}

template<int WPL>
__global__ void __launch_bounds__(256) k_edit_fwd_wide(const EditArgs a){
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t g = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
	if(g >= a.count) return;
	const uint32_t ppos = a.first + g, pair = a.order[ppos];
	if(a.status[pair] != 0u) return;
	const uint32_t qlen = a.qlen[pair], tlen = a.tlen[pair];
	const int type = a.mode & 3;
	const uint32_t BW = bsa_edit_bw_eff(qlen, tlen, type, a.bandwidth), NW = BW / 64u;
	const uint32_t qround = (qlen + 63u) / 64u * 64u;
	if(BW != qround || NW > 64u * WPL || (WPL > 1 && NW <= 32u * WPL)) return;      // moving or other-size band: another kernel of this launch
	const u64 *Q0m = a.qbits + a.qboff[pair];
	const u64 *Q1m = Q0m + a.qwords[pair];
	const uint8_t *tp = a.tst + a.tpoff[pair];
	u64 *rows = (u64*)(a.rows + a.slot_off[ppos]);
	const bool overlap = type == BSA_MODE_OVERLAP;
	const uint32_t w0 = lane * WPL;
	u64 q0[WPL], q1[WPL], vm[WPL], pv[WPL], mv[WPL];
	bool act[WPL];
#pragma unroll
	for(int j = 0; j < WPL; j++){
		const uint32_t w = w0 + j;
		act[j] = w < NW;
		q0[j] = act[j] ? Q0m[w] : 0ull; q1[j] = act[j] ? Q1m[w] : 0ull;
		vm[j] = (qlen > w * 64u) ? lowmask(qlen - w * 64u) : 0ull;          // band cells that are real query columns (:1112 nvalid)
		pv[j] = ~0ull; mv[j] = 0ull;
		if(act[j]){ rows[w] = 0ull; rows[NW + w] = ~0ull; }                    // row_init (:653-656)
	}
	const uint32_t lastw = (qlen - 1u) >> 6, lastb = (qlen - 1u) & 63u;      // H(qlen-1, row) is followed by the lane that owns that column
	const int lastj = (int)(lastw % WPL);
	int slast = (int)qlen, smin = 0x7FFFFFFF, ry = (int)tlen - 1;
	const u64 hin0_pos = overlap ? 0ull : 1ull;                                // left of the band v = +1, 0 in overlap mode (:770)
	u64 tw = 0;
	for(uint32_t i = 0; i < tlen; i++){
		if((i & 7u) == 0){
			const u64 *wp = (const u64*)(tp + i);                                // staged 16-byte aligned with >= 8 bytes of padding
			u64 v = *wp;
			tw = uniform64(v);
		}
		const uint32_t tb = (uint32_t)(tw >> (8u * (i & 7u))) & 3u;
		const u64 x0 = (tb & 1u) ? 0ull : ~0ull, x1 = (tb & 2u) ? 0ull : ~0ull;
		u64 Xv[WPL], t[2][WPL];
		bool n[2][WPL + 1], pz[2][WPL];          // n[s][j]: a negative delta enters word j when one enters the lane (s = 1) or not (s = 0)
		n[0][0] = false; n[1][0] = true;
#pragma unroll
		for(int j = 0; j < WPL; j++){
			const u64 Eq = (q0[j] ^ x0) & (q1[j] ^ x1) & vm[j];
			Xv[j] = Eq | mv[j];
			const u64 e1 = Eq | 1ull;
			const u64 t0 = (((Eq & pv[j]) + pv[j]) ^ pv[j]) | Eq;
			const u64 t1 = (((e1 & pv[j]) + pv[j]) ^ pv[j]) | e1;
#pragma unroll
			for(int sI = 0; sI < 2; sI++){
				const u64 xh = n[sI][j] ? t1 : t0;
				t[sI][j] = xh;
				const uint32_t ph = (uint32_t)((mv[j] | ~(xh | pv[j])) >> 63), mh = (uint32_t)((pv[j] & xh) >> 63);
				n[sI][j + 1] = act[j] ? (mh > ph) : n[sI][j];      // idle words pass the delta through (never used)
				pz[sI][j] = ph > mh;
			}
		}
		const u64 A = __ballot(n[0][WPL]), B = __ballot(n[1][WPL]);
		const u64 C = uniform64(chain_neg(A, B));                                // bit l: a negative delta enters lane l
		// the delta leaving each lane, as masks, for the shift-in of the next lane's first word
		const u64 PA = __ballot(pz[0][WPL - 1]), PB = __ballot(pz[1][WPL - 1]);
		const u64 Ppos = uniform64(((PA & ~C) | (PB & C)) << 1 | hin0_pos);    // bit l: a positive delta enters lane l
#pragma unroll
		for(int j = 0; j < WPL; j++){
			const u64 Xh = mask_pick64(t[0][j], t[1][j], C);
			u64 Ph = mv[j] | ~(Xh | pv[j]);
			u64 Mh = pv[j] & Xh;
			if(type != BSA_MODE_GLOBAL && j == lastj) slast += (int)((Ph >> lastb) & 1ull) - (int)((Mh >> lastb) & 1ull);    // only the owner lane's value is used
			u64 hneg, hpos;
			if(j == 0){ hneg = mask_pick(0u, 1u, C); hpos = mask_pick(0u, 1u, Ppos); }
			else { hneg = mask_pick((uint32_t)n[0][j], (uint32_t)n[1][j], C); hpos = mask_pick((uint32_t)pz[0][j - 1], (uint32_t)pz[1][j - 1], C); }
			Ph = (Ph << 1) | hpos;
			Mh = (Mh << 1) | hneg;
			pv[j] = Mh | ~(Xv[j] | Ph);
			mv[j] = Ph & Xv[j];
		}
		u64 *rp = rows + (size_t)(i + 1) * (2 * NW);
		if(WPL == 1){ if(act[0]){ rp[w0] = mv[0]; rp[NW + w0] = pv[0]; } }
		else {
#pragma unroll
			for(int j = 0; j < WPL; j++) if(act[j]){ rp[w0 + j] = mv[j]; rp[NW + w0 + j] = pv[j]; }
		}
		if(type != BSA_MODE_GLOBAL && slast < smin){ smin = slast; ry = (int)i; }
	}
	if(lane == lastw / WPL){ a.fwd_smin[ppos] = smin; a.fwd_ry[ppos] = ry; }
	if(lane == 0) a.fwd_sbeg[ppos] = overlap ? 0 : (int)tlen;                     // no band motion: H at the band start grows by one per row (:667-676)
}

// ---- bands up to 1024 columns, few pairs: G lanes per pair ------------------------------------------------------------
// The pair-per-lane kernel above walks its NW words one after the other, a dependent chain of some 75 instructions per
// word, and a batch that does not fill the chip (16384 pairs = one wave per CU) is bound by exactly that chain.  Here the
// words of a pair sit side by side in G = 2, 4, 8 or 16 lanes (lane gl owns word gl, 64 / G pairs per wave) and the chain
// across them is resolved as in k_edit_fwd_wide, per group: the top word of a group and idle lanes neither generate nor
// propagate, so no carry crosses into the next pair.  The band moves (global mode): the words shift right by movx bits
// with the next lane's word coming in through a DPP row shift (+1 cells shifted in on the right), lane 0 keeps H at the
// band start, and the query planes shift the same way, the top lane feeding them from a 128-bit look-ahead.
static __device__ __forceinline__ u64 dpp_next_lane64(u64 x){        // lane j <- lane j + 1 inside its row of 16
	const int lo = DPP_SHL(0, (int)(uint32_t)x, 1), hi = DPP_SHL(0, (int)(uint32_t)(x >> 32), 1);
	return (u64)(uint32_t)hi << 32 | (u64)(uint32_t)lo;
}

template<int G>
__global__ void __launch_bounds__(256) k_edit_fwd_grp(const EditArgs a){
	constexpr uint32_t PPW = 64u / G;
	constexpr u64 GS = G == 2 ? 0x5555555555555555ull : G == 4 ? 0x1111111111111111ull : G == 8 ? 0x0101010101010101ull : 0x0001000100010001ull;   // first lane of every group
	const uint32_t lane = threadIdx.x & 63u, gl = lane % G, grp = lane / G;
	const uint32_t g = (blockIdx.x * 4u + (threadIdx.x >> 6)) * PPW + grp;
	const bool valid = g < a.count;
	const uint32_t ppos = a.first + (valid ? g : 0u), pair = a.order[ppos];
	const bool live = valid && a.status[pair] == 0u;
	const uint32_t qlen = a.qlen[pair], tlen = a.tlen[pair];
	const uint32_t BW = a.bw, NW = BW / 64u;
	const u64 *Q0m = a.qbits + a.qboff[pair];
	const u64 *Q1m = Q0m + a.qwords[pair];
	const uint8_t *tp = a.tst + a.tpoff[pair];
	u64 *rows = (u64*)(a.rows + a.slot_off[ppos]);
	const int type = a.mode & 3;
	const bool overlap = type == BSA_MODE_OVERLAP;
	const uint32_t qround = (qlen + 63u) / 64u * 64u;
	const bool word = gl < NW, top = gl + 1u == NW;
	const uint32_t tl = live ? tlen : 0u;
	u64 pv = ~0ull, mv = 0ull;
	if(live && word){ rows[gl] = 0ull; rows[NW + gl] = ~0ull; }            // row_init (:653-656)
	u64 q0 = word ? Q0m[gl] : 0ull, q1 = word ? Q1m[gl] : 0ull;            // query planes at band offset 0
	// the top lane keeps the query bits behind the band end: position lpos sits in (l?c, l?n) at bit lpos & 63
	uint32_t lpos = BW;
	u64 l0c = 0, l0n = 0, l1c = 0, l1n = 0;
	if(top){ l0c = Q0m[NW]; l0n = Q0m[NW + 1u]; l1c = Q1m[NW]; l1n = Q1m[NW + 1u]; }
	int sbeg = 0;                                                           // lane 0 of the group: H at the band start
	uint32_t rb0 = 0;
	uint32_t quo = 0, rem = 0;                                              // floor(i * qlen / tlen) and the remainder, kept incrementally
	const uint32_t qstep = qlen / (tlen ? tlen : 1u), rstep = qlen % (tlen ? tlen : 1u);
	const uint32_t lastw = (qlen - 1u) >> 6, lastb = (qlen - 1u) & 63u;
	int slast = (int)qlen, smin = 0x7FFFFFFF, ry = (int)tlen - 1;
	const u64 hin0_mask = overlap ? 0ull : GS;
	u64 tw = 0;
	u64 *rp = rows + 2u * NW + gl;                                          // this lane's word of the row being written
	for(uint32_t i = 0; __any(i < tl); i++){
		const bool on = i < tl;
		if(on && (i & 7u) == 0u) tw = *(const u64*)(tp + i);      // staged 16-byte aligned with >= 8 bytes of padding
		const uint32_t tb = (((i & 4u) ? (uint32_t)(tw >> 32) : (uint32_t)tw) >> (8u * (i & 3u))) & 3u;
		uint32_t rb1 = 0;
		if(type == BSA_MODE_GLOBAL){                                         // fixed diagonal band (:1112-1114)
			uint32_t c = quo;
			c = (c < BW / 2) ? 0u : c - BW / 2;
			rb1 = (c + BW > qround) ? qround - BW : c;
		}
		uint32_t movx = on ? rb1 - rb0 : 0u;
		// ---- row_movx (:658-721)
		if(overlap) sbeg = 0; else if(on) sbeg += 1;
		if(__any(movx != 0u)){
			while(__any(movx >= 64u)){                                        // whole words (rare)
				u64 np = dpp_next_lane64(pv), nm = dpp_next_lane64(mv), nq0 = dpp_next_lane64(q0), nq1 = dpp_next_lane64(q1);
				if(top){ np = ~0ull; nm = 0ull; nq0 = fsr(l0c, l0n, lpos & 63u); nq1 = fsr(l1c, l1n, lpos & 63u); }
				if(movx >= 64u){
					if(!overlap) sbeg += __popcll(pv) - __popcll(mv);
					pv = np; mv = nm; q0 = nq0; q1 = nq1; movx -= 64u;
					lpos += 64u;
					if(top){ l0c = l0n; l1c = l1n; l0n = *(Q0m + (lpos >> 6) + 1u); l1n = *(Q1m + (lpos >> 6) + 1u); }
				}
			}
			u64 np = dpp_next_lane64(pv), nm = dpp_next_lane64(mv), nq0 = dpp_next_lane64(q0), nq1 = dpp_next_lane64(q1);
			if(top){ np = ~0ull; nm = 0ull; nq0 = fsr(l0c, l0n, lpos & 63u); nq1 = fsr(l1c, l1n, lpos & 63u); }
			if(movx){
				const u64 mk = (1ull << movx) - 1ull;
				const uint32_t r = 64u - movx;
				if(!overlap) sbeg += __popcll(pv & mk) - __popcll(mv & mk);
				pv = (pv >> movx) | (np << r);
				mv = (mv >> movx) | (nm << r);
				q0 = (q0 >> movx) | (nq0 << r);
				q1 = (q1 >> movx) | (nq1 << r);
				const uint32_t lw = lpos >> 6;
				lpos += movx;
				if(top && (lpos >> 6) != lw){ l0c = l0n; l1c = l1n; l0n = *(Q0m + (lpos >> 6) + 1u); l1n = *(Q1m + (lpos >> 6) + 1u); }
			}
		}
		// ---- row_cal (:766-810): this lane's word for both signs of the delta entering it, then the chain per group
		const bool act = on && word;
		const u64 x0 = (tb & 1u) ? 0ull : ~0ull, x1 = (tb & 2u) ? 0ull : ~0ull;
		const uint32_t nvalid = (rb1 < qlen) ? qlen - rb1 : 0u;          // band cells that are real query columns
		u64 Eq = (q0 ^ x0) & (q1 ^ x1);
		if(__any(nvalid < BW)){ const uint32_t lo = gl * 64u; Eq &= (nvalid > lo) ? lowmask(nvalid - lo) : 0ull; }
		const u64 Xv = Eq | mv;
		const u64 e1 = Eq | 1ull;
		const u64 t0 = (((Eq & pv) + pv) ^ pv) | Eq;
		const u64 t1 = (((e1 & pv) + pv) ^ pv) | e1;
		const uint32_t ph0 = (uint32_t)((mv | ~(t0 | pv)) >> 63), mh0 = (uint32_t)((pv & t0) >> 63);
		const uint32_t ph1 = (uint32_t)((mv | ~(t1 | pv)) >> 63), mh1 = (uint32_t)((pv & t1) >> 63);
		const bool link = act && !top;                                        // the top word's outgoing delta leaves the band
		const u64 A = __ballot(link && mh0 > ph0), B = __ballot(link && mh1 > ph1);
		const u64 C = uniform64(chain_neg(A, B));                             // bit l: a negative delta enters lane l
		const u64 PA = __ballot(link && ph0 > mh0), PB = __ballot(link && ph1 > mh1);
		const u64 Ppos = uniform64(((((PA & ~C) | (PB & C)) << 1) & ~GS) | hin0_mask);   // left of the band v = +1, 0 in overlap mode (:770)
		const u64 Xh = mask_pick64(t0, t1, C);
		u64 Ph = mv | ~(Xh | pv);
		u64 Mh = pv & Xh;
		if(type != BSA_MODE_GLOBAL && gl == lastw) slast += (int)((Ph >> lastb) & 1ull) - (int)((Mh >> lastb) & 1ull);
		Ph = (Ph << 1) | (u64)mask_pick(0u, 1u, Ppos);
		Mh = (Mh << 1) | (u64)mask_pick(0u, 1u, C);
		if(act){
			pv = Mh | ~(Xv | Ph);
			mv = Ph & Xv;
			rp[0] = mv; rp[NW] = pv;
			if(type != BSA_MODE_GLOBAL && slast < smin){ smin = slast; ry = (int)i; }
		}
		rb0 = on ? rb1 : rb0;
		quo += qstep;
		if(rem >= tlen - rstep){ rem -= tlen - rstep; quo++; } else rem += rstep;     // no 33-bit sum
		rp += 2u * NW;
	}
	if(live && gl == 0u) a.fwd_sbeg[ppos] = sbeg;
	if(live && gl == (type != BSA_MODE_GLOBAL ? lastw : 0u)){ a.fwd_smin[ppos] = smin; a.fwd_ry[ppos] = ry; }
}

// ---- the same with 32-bit words: G = 2 NW lanes per pair (bands up to 512 columns) --------------------------------------
// gfx950 has no 64-bit integer ALU: every u64 operation of k_edit_fwd_grp is two or three instructions (a funnel shift of the
// band by movx bits: six), and with one wave per SIMD the row time is the wave's own instruction latency.  With 32-bit words
// the block step is one instruction per operation, the band shift is one v_alignbit_b32 per plane, and the same batch is
// twice the waves (C3: two per SIMD), which interleave.  Row records are identical (natural bit order: the 32-bit word h of
// a plane is the low / high half of the 64-bit word h / 2).
// TILED: rows in format 1 (bsa_common.h).  A lane's word of both planes of eight consecutive rows is ONE 64-byte block: the row loop runs a tile at a
// time, the eight rows wait in registers and leave as four 16-byte stores -- written piecemeal (eight bytes a row) the blocks were merged by the L2 only
// while few waves were in flight (32768 pairs: 156 ms instead of 90), staged through LDS the kernel paid 12 % more instructions.
template<int G, bool TILED>
__global__ void __launch_bounds__(256) k_edit_fwd_grp32(const EditArgs a){
	constexpr uint32_t PPW = 64u / G;
	constexpr u64 GS = G == 2 ? 0x5555555555555555ull : G == 4 ? 0x1111111111111111ull : G == 8 ? 0x0101010101010101ull : 0x0001000100010001ull;   // first lane of every group
	const uint32_t lane = threadIdx.x & 63u, gl = lane % G, grp = lane / G;
	const uint32_t g = (blockIdx.x * 4u + (threadIdx.x >> 6)) * PPW + grp;
	const bool valid = g < a.count;
	const uint32_t ppos = a.first + (valid ? g : 0u), pair = a.order[ppos];
	const bool live = valid && a.status[pair] == 0u;
	const uint32_t qlen = a.qlen[pair], tlen = a.tlen[pair];
	const uint32_t BW = a.bw, NW = BW / 64u, NH = BW / 32u;                // NH 32-bit words per plane
	const uint32_t *Q0m = (const uint32_t*)(a.qbits + a.qboff[pair]);
	const uint32_t *Q1m = Q0m + 2u * a.qwords[pair];
	const uint8_t *tp = a.tst + a.tpoff[pair];
	uint32_t *rows = (uint32_t*)(a.rows + a.slot_off[ppos]);
	const int type = a.mode & 3;
	const bool overlap = type == BSA_MODE_OVERLAP;
	const uint32_t qround = (qlen + 63u) / 64u * 64u;
	const bool word = gl < NH, top = gl + 1u == NH;
	const u64 wordm = __builtin_amdgcn_ballot_w64(word && !top);             // lanes whose word hands a delta on
	const uint32_t tl = live ? tlen : 0u;
	uint32_t pv = ~0u, mv = 0u;
	if(live && word){                                                     // row_init (:653-656)
		if constexpr(!TILED){ rows[gl] = 0u; rows[NH + gl] = ~0u; }          // (TILED: the initial row leaves with tile 0)
	}
	uint32_t q0 = word ? Q0m[gl] : 0u, q1 = word ? Q1m[gl] : 0u;           // query planes at band offset 0
	// the top lane keeps the query bits behind the band end: position lpos sits in (l?c, l?n) at bit lpos & 31
	uint32_t lpos = BW;
	uint32_t l0c = 0, l0n = 0, l1c = 0, l1n = 0;
	if(top){ l0c = Q0m[NH]; l0n = Q0m[NH + 1u]; l1c = Q1m[NH]; l1n = Q1m[NH + 1u]; }
	int sbeg = 0;                                                           // lane 0 of the group: H at the band start
	uint32_t rb0 = 0;
	uint32_t quo = 0, rem = 0;                                              // floor(i * qlen / tlen) and the remainder, kept incrementally
	const uint32_t qstep = qlen / (tlen ? tlen : 1u), rstep = qlen % (tlen ? tlen : 1u);
	const uint32_t lastw = (qlen - 1u) >> 5, lastb = (qlen - 1u) & 31u;
	int slast = (int)qlen, smin = 0x7FFFFFFF, ry = (int)tlen - 1;
	const u64 hin0_mask = overlap ? 0ull : GS;
	u64 tw = 0;
	uint32_t *rp = rows + 2u * NH + gl;                                     // this lane's word of the row being written (format 0)
	// one row of the DP (everything but the row record's store); returns whether this lane's word took part
	auto row_step = [&](const uint32_t i) -> bool {
		const bool on = i < tl;
		if((i & 7u) == 0u){                                       // (uniform) the next eight target bases; waited for here, once per eight rows,
			if(on) tw = *(const u64*)(tp + i);                    // so that no row waits for the previous row's stores (vmcnt counts both)
			asm volatile("" : "+v"(tw));
		}
		const uint32_t tb = (((i & 4u) ? (uint32_t)(tw >> 32) : (uint32_t)tw) >> (8u * (i & 3u))) & 3u;
		uint32_t rb1 = 0;
		if(type == BSA_MODE_GLOBAL){                                         // fixed diagonal band (:1112-1114)
			uint32_t c = quo;
			c = (c < BW / 2) ? 0u : c - BW / 2;
			rb1 = (c + BW > qround) ? qround - BW : c;
		}
		uint32_t movx = on ? rb1 - rb0 : 0u;
		// ---- row_movx (:658-721)
		if(overlap) sbeg = 0; else if(on) sbeg += 1;
		if(__builtin_amdgcn_ballot_w64(movx >= 32u) != 0ull){
			while(__builtin_amdgcn_ballot_w64(movx >= 32u) != 0ull){                                        // whole words (rare)
				uint32_t np = (uint32_t)DPP_SHL(0, (int)pv, 1), nm = (uint32_t)DPP_SHL(0, (int)mv, 1), nq0 = (uint32_t)DPP_SHL(0, (int)q0, 1), nq1 = (uint32_t)DPP_SHL(0, (int)q1, 1);
				if(top){ np = ~0u; nm = 0u; nq0 = __builtin_amdgcn_alignbit(l0n, l0c, lpos & 31u); nq1 = __builtin_amdgcn_alignbit(l1n, l1c, lpos & 31u); }
				if(movx >= 32u){
					if(!overlap) sbeg += __popc(pv) - __popc(mv);
					pv = np; mv = nm; q0 = nq0; q1 = nq1; movx -= 32u;
					lpos += 32u;
					if(top){ l0c = l0n; l1c = l1n; l0n = Q0m[(lpos >> 5) + 1u]; l1n = Q1m[(lpos >> 5) + 1u]; }
				}
			}
		}
		if(type == BSA_MODE_GLOBAL){                                          // 0 .. 31 cells: no branch (a shift by 0 changes nothing)
			uint32_t np = (uint32_t)DPP_SHL(0, (int)pv, 1), nm = (uint32_t)DPP_SHL(0, (int)mv, 1), nq0 = (uint32_t)DPP_SHL(0, (int)q0, 1), nq1 = (uint32_t)DPP_SHL(0, (int)q1, 1);
			const uint32_t la0 = __builtin_amdgcn_alignbit(l0n, l0c, lpos & 31u), la1 = __builtin_amdgcn_alignbit(l1n, l1c, lpos & 31u);
			np = top ? ~0u : np; nm = top ? 0u : nm; nq0 = top ? la0 : nq0; nq1 = top ? la1 : nq1;
			const uint32_t mk = ~(~0u << movx);
			sbeg += __popc(pv & mk) - __popc(mv & mk);
			pv = __builtin_amdgcn_alignbit(np, pv, movx);
			mv = __builtin_amdgcn_alignbit(nm, mv, movx);
			q0 = __builtin_amdgcn_alignbit(nq0, q0, movx);
			q1 = __builtin_amdgcn_alignbit(nq1, q1, movx);
			const uint32_t lw = lpos >> 5;
			lpos += movx;
			if(top && (lpos >> 5) != lw){ l0c = l0n; l1c = l1n; l0n = Q0m[(lpos >> 5) + 1u]; l1n = Q1m[(lpos >> 5) + 1u]; asm volatile("" : "+v"(l0n), "+v"(l1n)); }
		}
		// ---- row_cal (:766-810): this lane's word for both signs of the delta entering it, then the chain per group
		const bool act = on && word;
		const uint32_t x0 = (tb & 1u) ? 0u : ~0u, x1 = (tb & 2u) ? 0u : ~0u;
		const uint32_t nvalid = (rb1 < qlen) ? qlen - rb1 : 0u;          // band cells that are real query columns
		uint32_t Eq = (q0 ^ x0) & (q1 ^ x1);
		if(__builtin_amdgcn_ballot_w64(nvalid < BW) != 0ull){ const uint32_t lo = gl * 32u; Eq &= (nvalid > lo) ? (nvalid - lo >= 32u ? ~0u : ((1u << (nvalid - lo)) - 1u)) : 0u; }
		const uint32_t Xv = Eq | mv;
		const uint32_t e1 = Eq | 1u;
		const uint32_t t0 = (((Eq & pv) + pv) ^ pv) | Eq;
		const uint32_t t1 = (((e1 & pv) + pv) ^ pv) | e1;
		const uint32_t ph0 = (mv | ~(t0 | pv)) >> 31, mh0 = (pv & t0) >> 31;
		const uint32_t ph1 = (mv | ~(t1 | pv)) >> 31, mh1 = (pv & t1) >> 31;
		// (the top word's outgoing delta leaves the band.  One vector compare a ballot, the lanes that take part ANDed in on the scalar side: a ballot of
		// `link && ...` is an `and` of conditions, which the backend materialises as 0 / 1 in a register and compares again -- eight vector instructions a row)
		const u64 linkm = __builtin_amdgcn_ballot_w64(on) & wordm;
		const u64 A = __builtin_amdgcn_ballot_w64(mh0 > ph0) & linkm, B = __builtin_amdgcn_ballot_w64(mh1 > ph1) & linkm;
		const u64 C = uniform64(chain_neg(A, B));                             // bit l: a negative delta enters lane l
		const u64 PA = __builtin_amdgcn_ballot_w64(ph0 > mh0) & linkm, PB = __builtin_amdgcn_ballot_w64(ph1 > mh1) & linkm;
		const u64 Ppos = uniform64(((((PA & ~C) | (PB & C)) << 1) & ~GS) | hin0_mask);   // left of the band v = +1, 0 in overlap mode (:770)
		const uint32_t Xh = mask_pick(t0, t1, C);
		uint32_t Ph = mv | ~(Xh | pv);
		uint32_t Mh = pv & Xh;
		if(type != BSA_MODE_GLOBAL && gl == lastw) slast += (int)((Ph >> lastb) & 1u) - (int)((Mh >> lastb) & 1u);
		Ph = (Ph << 1) | mask_pick(0u, 1u, Ppos);
		Mh = (Mh << 1) | mask_pick(0u, 1u, C);
		if(act){
			pv = Mh | ~(Xv | Ph);
			mv = Ph & Xv;
			if(type != BSA_MODE_GLOBAL && slast < smin){ smin = slast; ry = (int)i; }
		}
		rb0 = on ? rb1 : rb0;
		quo += qstep;
		if(rem >= tlen - rstep){ rem -= tlen - rstep; quo++; } else rem += rstep;     // no 33-bit sum
			return act;
	};
	if constexpr(!TILED){
		for(uint32_t i = 0; __any(i < tl); i++){
			if(row_step(i)){ rp[0] = mv; rp[NH] = pv; }
			rp += 2u * NH;
		}
	} else {
		// a tile at a time: rows 8 t .. 8 t + 7 (row 0 is the initial row, row r the one of target base r - 1), this lane's word of both planes of the
		// eight rows in sixteen registers, then ONE 64-byte block as four 16-byte stores (a lane whose pair ends inside the tile stores what it has;
		// the slots past its last row hold that row again -- nobody reads them)
		uint32_t bm[8], bp_[8];
		bm[0] = 0u; bp_[0] = ~0u;
		for(uint32_t t8 = 0; t8 == 0u || __builtin_amdgcn_ballot_w64(t8 - 1u < tl) != 0ull; t8 += 8u){
#pragma unroll
			for(uint32_t j = 0; j < 8u; j++){
				if(j == 0u && t8 == 0u) continue;                                // (the initial row)
				row_step(t8 + j - 1u);
				bm[j] = mv; bp_[j] = pv;
			}
			if(live && word && t8 <= tl){
				uint4 *dst = (uint4*)(rows + (size_t)(t8 >> 3) * (16u * NH) + (size_t)gl * 16u);
#pragma unroll
				for(int k = 0; k < 4; k++) dst[k] = make_uint4(bm[2 * k], bp_[2 * k], bm[2 * k + 1], bp_[2 * k + 1]);
			}
			if(__builtin_amdgcn_ballot_w64(t8 + 7u < tl) == 0ull) break;
		}
	}
	if(live && gl == 0u) a.fwd_sbeg[ppos] = sbeg;
	if(live && gl == (type != BSA_MODE_GLOBAL ? lastw : 0u)){ a.fwd_smin[ppos] = smin; a.fwd_ry[ppos] = ry; }
	(void)NW;
}

// ---------------------------------------------------------------------------------------------
// traceback (bsalign.h:965-1044) + end-cell / score selection of the driver (:1124-1139, 1180-1203)
// one pair per lane
// ---------------------------------------------------------------------------------------------
// The walk is latency-bound (dependent bit lookups) and every step of a wave waits for its slowest lane, so only a few
// lanes of every wave carry a pair: as few as still let ALL waves of the launch be resident at once (8 per SIMD).
// Measured on MI355X, 16384 pairs x 100 kbp (ms per launch): 32 lanes 151, 16 lanes 155, 8 lanes 137, 4 lanes 121,
// 2 lanes 114 (8192 waves = 8 per SIMD); one lane per wave would need two rounds.
// COOP (few pairs per wave, lanes <= 8): the lanes that carry no pair fetch rows for those that do.  Whenever a walker
// has moved RR - 1 rows (RR = 64 / lanes) the whole wave stops at a service point: each walker publishes the row it will look at next and a
// 64-column window centred on its column (a banded alignment runs along the middle of its band, which is a word border
// whenever the band has an even number of words), RR lanes per walker load one row each -- one load instruction, one
// latency for the next RR rows -- and leave the window's bits of both planes in an LDS ring.  A lookup the ring cannot
// serve (the walk left the window) takes the plain load, so the ring changes when data is read, never what is read.
template<bool COOP>
__global__ void __launch_bounds__(64) k_edit_trace(const EditArgs a, bsa_result_t *out, uint32_t *cig_cnt, uint32_t lanes){
	__shared__ u64 ring[COOP ? 128 : 2];                 // [walker][RR rows][plane]
	__shared__ u64 sh_ptr[8], sh_qp[8], sh_tp[8];
	__shared__ int sh_hi[8];
	__shared__ uint32_t sh_s[8], sh_nw[8], sh_ql[8], sh_tl[8];
	__shared__ __attribute__((aligned(8))) uint8_t sh_seq[COOP ? 8 : 1][64];      // per walker: 32 query bases up to x, 32 target bases up to y
	const uint32_t g0 = blockIdx.x * lanes + threadIdx.x;
	const bool walker = threadIdx.x < lanes && g0 < a.count;
	if(!COOP && !walker) return;
	const uint32_t g = walker ? g0 : blockIdx.x * lanes;    // COOP: the other lanes shadow the block's first pair and never step
	const uint32_t ppos = a.first + g, pair = a.order[ppos];
	bsa_result_t rs;
	rs.score = 0; rs.qb = rs.qe = rs.tb = rs.te = 0; rs.mat = rs.mis = rs.ins = rs.del = rs.aln = 0;
	const bool skip = a.status[pair] != 0u;                // flagged by an earlier stage: zero result
	if(!COOP && skip){ out[pair] = rs; cig_cnt[ppos] = 0; return; }
	const uint32_t qlen = a.qlen[pair], tlen = a.tlen[pair];
	const uint32_t BW = a.bw ? a.bw : bsa_edit_bw_eff(qlen, tlen, a.mode & 3, a.bandwidth), NW = BW / 64u;
	const uint8_t *qs = a.qst + a.qpoff[pair];
	const uint8_t *ts = a.tst + a.tpoff[pair];
	const u64 *rows = (const u64*)(a.rows + a.slot_off[ppos]);
	const int type = a.mode & 3;
	const uint32_t qround = (qlen + 63u) / 64u * 64u;
	auto plane_bit = [&](uint32_t row, int plane, long pos) -> int {
		// striped_seqedit_getval (:224) on `x - begs[..]`, which is UNSIGNED 32-bit arithmetic in the reference and whose
		// shift count x86 masks to 6 bits: striped word pos%W, bit (pos/W)&63  <=>  natural position below
		const uint32_t pu = (uint32_t)pos;
		const uint32_t p = ((pu / NW) & 63u) * NW + (pu % NW);
		return (int)((rows[(size_t)row * (2 * NW) + (size_t)plane * NW + (p >> 6)] >> (p & 63u)) & 1ull);
	};
	auto beg_of_row = [&](uint32_t r) -> uint32_t {    // begs[r]: r = 0 -> 0, r = i+1 -> band offset of target row i
		if(r == 0 || type != BSA_MODE_GLOBAL) return 0u;
		uint32_t c = (uint32_t)(((u64)(r - 1) * qlen) / tlen);
		c = (c < BW / 2) ? 0u : c - BW / 2;
		return (c + BW > qround) ? qround - BW : c;
	};
	int rx = (int)qlen - 1, ry = (int)tlen - 1, smin = 0x7FFFFFFF;
	const int sbeg_last = a.fwd_sbeg[ppos];
	int score = 0;
	if(type == BSA_MODE_GLOBAL){
		const uint32_t rbl = beg_of_row(tlen);
		const u64 *lr = rows + (size_t)tlen * (2 * NW);
		score = sbeg_last;
		for(uint32_t k = 0; k < NW; k++) score += __popcll(lr[NW + k]) - __popcll(lr[k]);
		for(uint32_t k = rbl + BW; k > qlen; k--){
			score += plane_bit(tlen, 0, (long)(k - 1 - rbl)) - plane_bit(tlen, 1, (long)(k - 1 - rbl));
		}
	} else {
		// full band, rbeg == 0: the minimum over the rows of H(qlen-1, i) and its first row come from the forward kernel
		// (:1124-1139), then (extend) the first strict minimum of the last row
		smin = a.fwd_smin[ppos]; rx = (int)qlen - 1; ry = a.fwd_ry[ppos];
		if(type == BSA_MODE_EXTEND){     // striped_seqedit_rowmin (:813-963)
			const u64 *lr = rows + (size_t)tlen * (2 * NW);
			int sc = (int)tlen, best = sc; uint32_t pmin = 0;
			for(uint32_t p = 0; p < BW; p++){
				sc += (int)((lr[NW + (p >> 6)] >> (p & 63u)) & 1ull) - (int)((lr[p >> 6] >> (p & 63u)) & 1ull);
				if(sc < best){ best = sc; pmin = p; }
			}
			if(best < smin){ smin = best; rx = (int)pmin; ry = (int)tlen - 1; }
		}
	}
	// ---- backtrace
	uint32_t *cig_end = (uint32_t*)((uint8_t*)rows + (size_t)(tlen + 1 + a.pad_rows) * (2 * NW) * 8);
	uint32_t ncig = 0, cg = 0, op = 0;
	auto cig_push = [&](uint32_t w){ ncig++; *(cig_end - ncig) = w; };
	int x = rx, y = ry;
	bool bad = (rx >= (int)qlen);
	rs.qe = x + 1; rs.te = y + 1;
	// begs[y+1], begs[y] kept incrementally (y only ever decreases by one): (bq, br) = divmod((y-1)*qlen, tlen)
	uint32_t b1 = beg_of_row((uint32_t)y + 1u), b0 = beg_of_row((uint32_t)y);
	const uint32_t qstep = qlen / tlen, rstep = qlen % tlen;
	uint32_t bq = 0, br = 0;                              // quotient < qlen, remainder < tlen
	if(y >= 1){ const u64 pr = (u64)(y - 1) * qlen; bq = (uint32_t)(pr / tlen); br = (uint32_t)(pr % tlen); }
	int cached_y = y;
	// 8 bases of each sequence in a register window: the common step (equal bases, no plane lookup) then touches
	// memory once per 8 steps instead of twice per step
	u64 qwin = 0, twin = 0; int qwb = -1000, twb = -1000;
	auto qbase_at = [&](int idx) -> int {
		if(idx < qwb || idx >= qwb + 8){ qwb = max(idx - 7, 0); __builtin_memcpy(&qwin, qs + qwb, 8); }
		return (int)((qwin >> (8 * (idx - qwb))) & 0xffu);
	};
	auto tbase_at = [&](int idx) -> int {
		if(idx < twb || idx >= twb + 8){ twb = max(idx - 7, 0); __builtin_memcpy(&twin, ts + twb, 8); }
		return (int)((twin >> (8 * (idx - twb))) & 0xffu);
	};
	// COOP service state
	// RR rows (one per lane) and 32 bases of each sequence per service point: a walker may move SVC rows / bases before
	// the next one (the lookups of a step touch rows y + 1 and y)
	const uint32_t RR = COOP ? 64u / lanes : 1u, SVC = COOP ? min(RR, 32u) - 1u : 1u;
	const uint32_t hw = threadIdx.x / RR, hl = threadIdx.x % RR;
	int ring_lo = 1, ring_hi = 0; uint32_t ring_s = 0;
	int sq0 = 0, st0 = 0;                                    // first base of the LDS sequence windows
	auto map_pos = [&](long pos) -> uint32_t {               // see plane_bit; the identity for positions inside the band
		const uint32_t pu = (uint32_t)pos;
		return pu < BW ? pu : ((pu / NW) & 63u) * NW + (pu % NW);
	};
	auto two_bits = [&](uint32_t row, long pos, int &bit0, int &bit1){
		const uint32_t p = map_pos(pos);
		u64 w0, w1; uint32_t sh;
		if(COOP && (int)row >= ring_lo && (int)row <= ring_hi && p >= ring_s && p < ring_s + 64u){
			const uint32_t slot = (threadIdx.x * RR + (row & (RR - 1u))) * 2u;
			w0 = ring[slot]; w1 = ring[slot + 1u]; sh = p - ring_s;
		} else {
			const u64 *rp = rows + (size_t)row * (2 * NW) + (p >> 6);
			w0 = rp[0]; w1 = rp[NW]; sh = p & 63u;
		}
		bit0 = (int)((w0 >> sh) & 1ull); bit1 = (int)((w1 >> sh) & 1ull);
	};
	bool active = walker && !skip && !bad && x >= 0 && y >= 0;
	uint32_t svc_left = 0, step = 1;      // svc_left: rows / bases a walker may still move before the next service point
	while(COOP ? __any(active) : active){
		if(active && type == BSA_MODE_GLOBAL){           // a run of matches moves several rows at once
			while(cached_y != y){
				b1 = b0; cached_y--;
				if(cached_y >= 1){
					bq -= qstep;
					if(br < rstep){ br += tlen; bq--; }
					br -= rstep;
					uint32_t c = bq;
					c = (c < BW / 2) ? 0u : c - BW / 2;
					b0 = (c + BW > qround) ? qround - BW : c;
				} else b0 = 0;
			}
		}
		if(COOP && __any(active && svc_left == 0u)){     // service point, the whole wave (some walker has used up its window)
			if(threadIdx.x < lanes){
				const uint32_t p = map_pos((long)x - (long)b1);
				ring_s = (p < 32u) ? 0u : min(p - 32u, BW - 64u);
				ring_hi = active ? y + 1 : -1; ring_lo = max(ring_hi - (int)RR + 1, 0);
				sq0 = max(x - 31, 0); st0 = max(y - 31, 0);
				sh_hi[threadIdx.x] = ring_hi; sh_s[threadIdx.x] = ring_s; sh_nw[threadIdx.x] = NW; sh_ptr[threadIdx.x] = (u64)(uintptr_t)rows;
				sh_qp[threadIdx.x] = (u64)(uintptr_t)(qs + sq0); sh_tp[threadIdx.x] = (u64)(uintptr_t)(ts + st0);
				sh_ql[threadIdx.x] = active ? qlen + 8u - (uint32_t)sq0 : 0u; sh_tl[threadIdx.x] = active ? tlen + 8u - (uint32_t)st0 : 0u;   // bytes that may be read
			}
			__syncthreads();
			if(hl < 8u){         // eight lanes per walker bring the 32 bases of each sequence up to (x, y)
				const bool isq = hl < 4u;
				const uint32_t off = (hl & 3u) * 8u, lim = isq ? sh_ql[hw] : sh_tl[hw];
				if(off <= lim && lim != 0u){
					const uint8_t *src = (const uint8_t*)(uintptr_t)(isq ? sh_qp[hw] : sh_tp[hw]) + off;     // staged with >= 16 bytes of padding
					u64 v; __builtin_memcpy(&v, src, 8);
					*(u64*)&sh_seq[hw][(isq ? 0u : 32u) + off] = v;
				}
			}
			{                    // and every lane one row: the window's two words of both planes
				const int hi = sh_hi[hw], r = hi - (int)hl;
				if(hi >= 0 && r >= 0){
					const uint32_t nw = sh_nw[hw], ws = sh_s[hw], wa = ws >> 6, fs = ws & 63u;
					const u64 *rp = (const u64*)(uintptr_t)sh_ptr[hw] + (size_t)r * (2 * nw) + wa;
					const bool two = wa + 1u < nw;
					const u64 l0 = rp[0], l1 = rp[nw], h0 = two ? rp[1] : 0ull, h1 = two ? rp[nw + 1u] : 0ull;
					const uint32_t slot = (hw * RR + ((uint32_t)r & (RR - 1u))) * 2u;
					ring[slot] = fsr(l0, h0, fs); ring[slot + 1u] = fsr(l1, h1, fs);
				}
			}
			__syncthreads();
			svc_left = SVC;
		}
		if(active){
			// the bases at (x, y) and up to seven before them, most recent in the top byte: equal top bytes = a match, and the
			// number of equal leading bytes is how far the diagonal run of matches goes (as far as both windows reach)
			u64 qv, tv; uint32_t reach;
			if(COOP){
				const uint32_t eq = (uint32_t)(x - sq0), et = (uint32_t)(y - st0);
				const u64 *sq = (const u64*)&sh_seq[threadIdx.x][0], *st = (const u64*)&sh_seq[threadIdx.x][32];
				const u64 q1 = sq[eq >> 3], q0 = (eq >> 3) ? sq[(eq >> 3) - 1u] : 0ull, t1 = st[et >> 3], t0 = (et >> 3) ? st[(et >> 3) - 1u] : 0ull;
				const uint32_t sq_ = 8u * (7u - (eq & 7u)), st_ = 8u * (7u - (et & 7u));
				qv = (q1 << sq_) | (sq_ ? q0 >> (64u - sq_) : 0ull);
				tv = (t1 << st_) | (st_ ? t0 >> (64u - st_) : 0ull);
				reach = min(min(eq, et) + 1u, svc_left);
			} else {
				(void)qbase_at(x); (void)tbase_at(y);        // windows now hold [qwb, qwb + 8) with x inside, same for y
				qv = qwin << (8u * (7u - (uint32_t)(x - qwb))); tv = twin << (8u * (7u - (uint32_t)(y - twb)));
				reach = (uint32_t)min(x - qwb, y - twb) + 1u;
			}
			const u64 df = qv ^ tv;
			if((df >> 56) == 0ull){
				const uint32_t lead = df ? (uint32_t)__clzll((long long)df) >> 3 : 8u;
				const uint32_t run = min(lead, reach);
				rs.mat += (int)run; op = 0; x -= (int)run; y -= (int)run; step = run;
			}
			else {
				step = 1u;
				const long p1 = (long)x - (long)b1;
				int u3, u4;
				two_bits((uint32_t)y + 1u, p1, u3, u4);
				if(u3 == 0 && u4 == 1){ rs.ins++; op = 1; x--; }
				else {
					const long p0 = (long)x - (long)b0;
					int u1, u2;
					two_bits((uint32_t)y, p0, u1, u2);
					if(u1 == 1 && u2 == 0){ rs.del++; op = 2; y--; }
					else { rs.mis++; op = 0; x--; y--; }
				}
			}
			if(op == (cg & 0xf)) cg += 0x10u * step;
			else { if(cg) cig_push(cg); cg = (0x10u * step) | op; }
			active = x >= 0 && y >= 0;
			svc_left -= min(step, svc_left);
		}
	}
	if(COOP && (!walker || skip)){                     // flagged by an earlier stage: the zero result
		if(walker){
			rs.score = 0; rs.qb = rs.qe = rs.tb = rs.te = 0; rs.mat = rs.mis = rs.ins = rs.del = rs.aln = 0;
			out[pair] = rs; cig_cnt[ppos] = 0;
		}
		return;
	}
	if(!bad){
		rs.qb = x + 1; rs.tb = y + 1;
		if(rs.qb){
			op = 1;
			if(op == (cg & 0xf)) cg += 0x10u * (uint32_t)rs.qb;
			else { if(cg) cig_push(cg); cg = (0x10u * (uint32_t)rs.qb) | op; }
			rs.ins += rs.qb; rs.qb = 0;
		}
		if((type == BSA_MODE_GLOBAL || type == BSA_MODE_EXTEND) && rs.tb){
			op = 2;
			if(op == (cg & 0xf)) cg += 0x10u * (uint32_t)rs.tb;
			else { if(cg) cig_push(cg); cg = (0x10u * (uint32_t)rs.tb) | op; }
			rs.del += rs.tb; rs.tb = 0;
		}
		rs.aln = rs.mat + rs.mis + rs.ins + rs.del;
		if(cg) cig_push(cg);
		if(type == BSA_MODE_OVERLAP) rs.score = smin + rs.te - rs.tb;
		else if(type == BSA_MODE_EXTEND) rs.score = smin;
		else rs.score = score;
	} else {
		atomicOr(&a.status[pair], BSA_ST_TRACE);
		rs.score = 0; rs.qb = rs.qe = rs.tb = rs.te = 0; rs.mat = rs.mis = rs.ins = rs.del = rs.aln = 0;
		ncig = 0;
	}
	out[pair] = rs;
	cig_cnt[ppos] = ncig;
}

// ---------------------------------------------------------------------------------------------
// traceback, ONE walk per wave (few long pairs: C3 has 16 walks per SIMD, far too few to hide a lane-per-walk chain)
// ---------------------------------------------------------------------------------------------
// Every decision of striped_seqedit_backtrace (bsalign.h:986-1010) is a function of the cell alone, so the 64 lanes
// evaluate the 63 cells that FOLLOW the walker on its diagonal -- lane i looks at (x - i, y - i): bases equal? if not,
// the two plane bits of rows y - i + 1 and y - i at that column -- and one ballot finds the first cell that is an
// insertion or a deletion.  Everything before it is a run of matches / mismatches taken in one step (a CIGAR run of op 0
// of that length), then the I or D moves the diagonal and the lanes behind it re-evaluate.  Steps per walk = indels +
// tiles instead of one per column.  The walker's state (x, y, CIGAR word, counters) is wave-uniform and lives in SGPRs.
// A tile = 64 consecutive rows (lane i owns row R_hi - i), a window of EW_WW words per plane of each row in LDS -- the
// whole row for bands up to 256 columns, else 256 columns centred on the diagonal -- 4 KB per tile; the target base of a
// lane's row is fixed for the tile; the query bases sit in a 512-byte LDS window refilled every few tiles.  A lookup
// outside the band (the reference's unsigned position arithmetic, plane_bit above) or outside the window takes the
// literal per-lane path with plain loads.
// TILED (row format 1, bsa_common.h; bands of 64 / 128 / 256 columns): the rows sit eight to a tile, a 64-byte block per 32-bit column word, and a lane
// fetches THREE dwords per plane of its row -- the 96 columns around where the walker's diagonal is expected to cross it (the estimate is made a tile
// ahead; the drift of a tile's indels is a few columns, and what falls outside is looked up literally) -- so a tile of 64 rows costs 24 blocks of 64
// bytes instead of 64 rows of 64 (of 32 * NW) bytes.
#define EW_WW 4
#define EW_QWIN 512
#define EW_TD 3               // TILED: dwords per plane and row in the window
template<bool TILED>
__global__ void __launch_bounds__(64) k_edit_trace_wave(const EditArgs a, bsa_result_t *out, uint32_t *cig_cnt){
	__shared__ uint32_t tile[64][4 * EW_WW + 1];     // rows padded to 17 dwords: lanes 64 bytes apart would meet in 4 of the 64 banks
	__shared__ uint32_t s_beg[64], s_ws[64];
	__shared__ __attribute__((aligned(8))) uint8_t s_q[EW_QWIN];
	const uint32_t lane = threadIdx.x;
	const uint32_t ppos = a.first + blockIdx.x, pair = a.order[ppos];
	bsa_result_t rs;
	rs.score = 0; rs.qb = rs.qe = rs.tb = rs.te = 0; rs.mat = rs.mis = rs.ins = rs.del = rs.aln = 0;
	if(a.status[pair] != 0u){ if(lane == 0){ out[pair] = rs; cig_cnt[ppos] = 0; } return; }
	const uint32_t qlen = a.qlen[pair], tlen = a.tlen[pair];
	const uint32_t BW = a.bw ? a.bw : bsa_edit_bw_eff(qlen, tlen, a.mode & 3, a.bandwidth), NW = BW / 64u;
	const uint8_t *qs = a.qst + a.qpoff[pair];
	const uint8_t *ts = a.tst + a.tpoff[pair];
	const u64 *rows = (const u64*)(a.rows + a.slot_off[ppos]);
	const uint32_t *rowsd = (const uint32_t*)rows;
	const uint32_t NH = 2u * NW;
	const int type = a.mode & 3;
	const uint32_t qround = (qlen + 63u) / 64u * 64u;
	auto plane_bit = [&](uint32_t row, int plane, long pos) -> int {         // as in k_edit_trace
		const uint32_t pu = (uint32_t)pos;
		const uint32_t p = ((pu / NW) & 63u) * NW + (pu % NW);
		if constexpr(TILED) return (int)((rowsd[bsa_edit_row_dword(NH, row, (uint32_t)plane, p >> 5)] >> (p & 31u)) & 1u);
		else return (int)((rows[(size_t)row * (2 * NW) + (size_t)plane * NW + (p >> 6)] >> (p & 63u)) & 1ull);
	};
	auto row_word = [&](uint32_t row, uint32_t plane, uint32_t w) -> u64 {    // 64-bit word w of a plane of a row
		if constexpr(TILED) return (u64)rowsd[bsa_edit_row_dword(NH, row, plane, 2u * w)] | ((u64)rowsd[bsa_edit_row_dword(NH, row, plane, 2u * w + 1u)] << 32);
		else return rows[(size_t)row * (2 * NW) + (size_t)plane * NW + w];
	};
	auto beg_of_row = [&](uint32_t r) -> uint32_t {
		if(r == 0 || type != BSA_MODE_GLOBAL) return 0u;
		uint32_t c = (uint32_t)(((u64)(r - 1) * qlen) / tlen);
		c = (c < BW / 2) ? 0u : c - BW / 2;
		return (c + BW > qround) ? qround - BW : c;
	};
	// ---- end cell and score (uniform; the same statements as k_edit_trace, the row scans spread over the lanes)
	int rx = (int)qlen - 1, ry = (int)tlen - 1, smin = 0x7FFFFFFF, score = 0;
	if(type == BSA_MODE_GLOBAL){
		const uint32_t rbl = beg_of_row(tlen);
		int part = 0;
		for(uint32_t k = lane; k < NW; k += 64u) part += __popcll(row_word(tlen, 1u, k)) - __popcll(row_word(tlen, 0u, k));
		for(uint32_t k = qlen + 1u + lane; k <= rbl + BW; k += 64u) part += plane_bit(tlen, 0, (long)(k - 1 - rbl)) - plane_bit(tlen, 1, (long)(k - 1 - rbl));
		for(int o = 32; o; o >>= 1) part += __shfl_xor(part, o);
		score = a.fwd_sbeg[ppos] + part;
	} else {
		smin = a.fwd_smin[ppos]; ry = a.fwd_ry[ppos];
		if(type == BSA_MODE_EXTEND){     // striped_seqedit_rowmin (:813-963): first strict minimum of the prefix sums of the last row
			// lane l owns words l, l + 64, ... : word totals first, then the scan inside the words
			int best = (int)tlen; uint32_t pmin = 0; int base = (int)tlen;
			for(uint32_t w0 = 0; w0 < NW; w0 += 64u){
				const uint32_t w = w0 + lane;
				const u64 pl0 = w < NW ? row_word(tlen, 0u, w) : 0ull, pl1 = w < NW ? row_word(tlen, 1u, w) : 0ull;
				int tot = __popcll(pl1) - __popcll(pl0), pre = tot;         // inclusive prefix over the lanes
				for(int o = 1; o < 64; o <<= 1){ const int v = __shfl_up(pre, o); if((int)lane >= o) pre += v; }
				int sc = base + pre - tot, lb = 0x7FFFFFFF; uint32_t lp = 0;
				if(w < NW){
					for(uint32_t b = 0; b < 64u; b++){
						sc += (int)((pl1 >> b) & 1ull) - (int)((pl0 >> b) & 1ull);
						if(sc < lb){ lb = sc; lp = w * 64u + b; }
					}
				}
				for(int o = 1; o < 64; o <<= 1){            // first strict minimum over the lanes: smaller value, then smaller position
					const int vb = __shfl_xor(lb, o); const uint32_t vp = __shfl_xor(lp, o);
					if(vb < lb || (vb == lb && vp < lp)){ lb = vb; lp = vp; }
				}
				if(lb < best){ best = lb; pmin = lp; }
				base += __shfl(pre, 63);
			}
			if(best < smin){ smin = best; rx = (int)pmin; ry = (int)tlen - 1; }
		}
	}
	rx = __builtin_amdgcn_readfirstlane(rx); ry = __builtin_amdgcn_readfirstlane(ry);
	// ---- backtrace
	uint32_t *cig_end = (uint32_t*)((uint8_t*)rows + (size_t)(tlen + 1 + a.pad_rows) * (2 * NW) * 8);
	// CIGAR: one token per gap event (matches / mismatches since the last event, op, length), one token per lane; a gap that goes
	// on (same op, no match in between) extends its own token, so neighbouring tokens never carry the same op and every 64 tokens
	// the lanes turn them into words side by side (an M word when the run is not empty, then the gap's word; positions from two
	// prefix popcounts; word m at cig_end - (m + 1)).  The same scheme as k_align8_trace_codes_wave (bsa_align8_codes.hip).
	uint32_t ncig = 0, tokN = 0, tokB = 0;            // words written; the lane's token: run length, len << 2 | op (op 0: none)
	// (uniform) tokens held; key = op of the last token << 28 | match / mismatch columns since it, so that "same op, nothing in between" is one
	// compare (28 bits: the length field of a CIGAR word, length << 4 | op in 32 bits -- a longer run has no word in the reference's format either)
	uint32_t ntok = 0, key = 0;
	constexpr uint32_t KEYM = 0x0FFFFFFFu;
	auto tok_flush = [&](uint32_t cnt){
		const bool in = lane < cnt;
		tokN &= KEYM;
		const bool hasA = in && tokN != 0u, hasB = in && (tokB & 3u) != 0u;
		const u64 mA = __ballot(hasA), mB = __ballot(hasB);
		const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mA, 0u))
			+ __builtin_amdgcn_mbcnt_hi((uint32_t)(mB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mB, 0u));
		uint32_t *wp = cig_end - (ncig + below + 1u);
		if(hasA){ *wp = tokN << 4; wp--; }
		if(hasB) *wp = ((tokB >> 2) << 4) | (tokB & 3u);
		ncig += (uint32_t)(__popcll(mA) + __popcll(mB));
	};
	auto emit = [&](uint32_t op, uint32_t len){
		if(op == 0u){ key += len; return; }
		if(key == (op << 28)){ if(lane + 1u == ntok) tokB += len << 2; return; }
		if(ntok == 64u){
			tok_flush(63u);
			tokN = (uint32_t)__builtin_amdgcn_readlane((int)tokN, 63); tokB = (uint32_t)__builtin_amdgcn_readlane((int)tokB, 63);
			ntok = 1u;
		}
		if(lane == ntok){ tokN = key; tokB = (len << 2) | op; }       // (the op field of the key is masked off when the token is written)
		ntok++; key = op << 28;
	};
	auto cig_finish = [&](){
		if(key & KEYM){
			if(ntok == 64u){ tok_flush(64u); ntok = 0u; }
			if(lane == ntok){ tokN = key; tokB = 0u; }
			ntok++; key = 0u;
		}
		tok_flush(ntok); ntok = 0u;
	};
	int x = rx, y = ry;
#ifdef EDIT_DBG
	uint32_t dbg_lit = 0, dbg_tiles = 0, dbg_demand = 0;
#endif
	uint32_t vmis = 0;                               // this lane's mismatch count (summed over the wave at the end)
	const bool bad = (rx >= (int)qlen);
	rs.qe = x + 1; rs.te = y + 1;
	const bool dbl_ok = (u64)tlen * (u64)qlen < (1ull << 52);
	const double inv_t = 1.0 / (double)tlen;
	int qw_lo = 0, qw_hi = -1;                       // query bases [qw_lo, qw_hi] are in s_q
	// a tile's rows for lane i: row R_hi - i, window start ws (in words), band offset beg, target base of the row
	struct TileRegs { u64 w[2][EW_WW]; uint32_t beg, ws, tb; };
	// The band offset of row r is (r - 1) qlen / tlen, clamped (beg_of_row).  A tile that follows the last one fetched 63 rows further up -- every tile
	// but the first, unless the walk left its tile early -- takes each lane's quotient and remainder from that tile's and the constants of 63 qlen / tlen:
	// a subtraction and a borrow where the division was a third of a tile's vector instructions (and in f64)
	const u64 step63 = 63ull * qlen;
	const uint32_t dq63 = (uint32_t)(step63 / tlen), dr63 = (uint32_t)(step63 % tlen);
	uint32_t bq = 0, br = 0;                         // the lane's quotient and remainder in the tile fetched last (rows r >= 1)
	auto tile_fetch = [&](int R_hi, int xs, TileRegs &t, bool follows){
		const int r = R_hi - (int)lane;
		uint32_t beg = 0;
		if(type == BSA_MODE_GLOBAL && r >= 1){
			uint32_t c;
			if(follows){
				const uint32_t borrow = br < dr63 ? 1u : 0u;
				br = br - dr63 + (borrow ? tlen : 0u);
				c = bq - dq63 - borrow;
			} else {
				const u64 n = (u64)(uint32_t)(r - 1) * qlen;
				if(dbl_ok){
					u64 qe = (u64)((double)n * inv_t);
					long rem = (long)(n - qe * (u64)tlen);
					if(rem < 0){ qe--; rem += (long)tlen; } else if(rem >= (long)tlen){ qe++; rem -= (long)tlen; }
					c = (uint32_t)qe; br = (uint32_t)rem;
				} else { c = (uint32_t)(n / tlen); br = (uint32_t)(n - (u64)c * tlen); }
			}
			bq = c;
			c = (c < BW / 2) ? 0u : c - BW / 2;
			beg = (c + BW > qround) ? qround - BW : c;
		}
		uint32_t ws = 0;
		if constexpr(TILED){
			// the window's first dword: the 32 columns the lane will look at start 16 left of the diagonal's column in this row; one more dword to either
			// side.  The row is a RING of NH dwords (NH a power of two): a position outside the band is looked up where the reference's unsigned
			// arithmetic lands (plane_bit), the position modulo BW -- a walk that runs along the band's edge stays on the fast path
			const int oe = (xs - (int)lane - 16) - (int)beg;
			ws = (uint32_t)((oe - 16) >> 5) & (NH - 1u);
		} else
		if(NW > EW_WW){
			const int pe = (xs - (int)lane) - (int)beg;
			int w = (pe >> 6) - 1;
			w = w < 0 ? 0 : w;
			ws = (uint32_t)w > NW - EW_WW ? NW - EW_WW : (uint32_t)w;
		}
		t.beg = beg; t.ws = ws;
		t.tb = r >= 1 ? (uint32_t)ts[r - 1] : 0xffu;
		if constexpr(TILED){
			// three unconditional loads (nothing after them depends on the data until the tile is used: the next tile's rows travel while this one is
			// walked): dwords ws, ws + 1, ws + 2 of the row's ring.  A row above the first is clamped: it belongs to a lane that stops the walk anyway.
			const uint32_t rr = r > 0 ? (uint32_t)r : 0u;
			const uint32_t *bp = rowsd + bsa_edit_row_dword(NH, rr, 0u, 0u);
			const uint32_t h1 = (ws + 1u) & (NH - 1u), h2 = (ws + 2u) & (NH - 1u);
			const uint2 d0 = *(const uint2*)(bp + ws * 16u), d1 = *(const uint2*)(bp + h1 * 16u), d2 = *(const uint2*)(bp + h2 * 16u);
			t.w[0][0] = (u64)d0.x | ((u64)d1.x << 32); t.w[0][1] = (u64)d2.x;
			t.w[1][0] = (u64)d0.y | ((u64)d1.y << 32); t.w[1][1] = (u64)d2.y;
#pragma unroll
			for(int w = 2; w < EW_WW; w++){ t.w[0][w] = 0ull; t.w[1][w] = 0ull; }
		} else
		if(r >= 0){
			const u64 *rp = rows + (size_t)(uint32_t)r * (2 * NW) + ws;
			if(NW == EW_WW){
				const uint4 *r4 = (const uint4*)rp;
				const uint4 v0 = r4[0], v1 = r4[1], v2 = r4[2], v3 = r4[3];
				t.w[0][0] = (u64)v0.x | ((u64)v0.y << 32); t.w[0][1] = (u64)v0.z | ((u64)v0.w << 32);
				t.w[0][2] = (u64)v1.x | ((u64)v1.y << 32); t.w[0][3] = (u64)v1.z | ((u64)v1.w << 32);
				t.w[1][0] = (u64)v2.x | ((u64)v2.y << 32); t.w[1][1] = (u64)v2.z | ((u64)v2.w << 32);
				t.w[1][2] = (u64)v3.x | ((u64)v3.y << 32); t.w[1][3] = (u64)v3.z | ((u64)v3.w << 32);
			} else {
#pragma unroll
				for(int w = 0; w < EW_WW; w++){
					const bool in = ws + (uint32_t)w < NW;
					t.w[0][w] = in ? rp[w] : 0ull;
					t.w[1][w] = in ? rp[NW + w] : 0ull;
				}
			}
		} else {
#pragma unroll
			for(int w = 0; w < EW_WW; w++){ t.w[0][w] = 0ull; t.w[1][w] = 0ull; }
		}
	};
	TileRegs pf;
	int pf_R = -1;                                   // the tile (its R_hi) whose rows are on their way in pf
	auto q_refill = [&](int xx){                     // query window [qw_lo, qw_lo + 512) with xx near its upper end
		int lo = (xx + 8 - EW_QWIN) & ~7;
		lo = lo < 0 ? 0 : lo;
		qw_lo = lo; qw_hi = lo + EW_QWIN - 1;
		__syncthreads();
		if((uint32_t)lo + 8u * lane < qlen + 8u){ u64 v; __builtin_memcpy(&v, qs + lo + 8 * lane, 8); *(u64*)&s_q[8 * lane] = v; }
		__syncthreads();
	};
	if(!bad && x >= 0) q_refill(x);
	while(!bad && x >= 0 && y >= 0){
		const int R_hi = y + 1;
#ifdef EDIT_DBG
		dbg_tiles++; if(pf_R != R_hi) dbg_demand++;
#endif
		if(pf_R != R_hi) tile_fetch(R_hi, x, pf, false);
		__syncthreads();
		if constexpr(TILED){
			// three dwords a plane: slots 0 .. 2 and 2 EW_WW .. 2 EW_WW + 2 (the window() of this format reads no others)
			tile[lane][0] = (uint32_t)pf.w[0][0]; tile[lane][1] = (uint32_t)(pf.w[0][0] >> 32); tile[lane][2] = (uint32_t)pf.w[0][1];
			tile[lane][2 * EW_WW] = (uint32_t)pf.w[1][0]; tile[lane][2 * EW_WW + 1] = (uint32_t)(pf.w[1][0] >> 32); tile[lane][2 * EW_WW + 2] = (uint32_t)pf.w[1][1];
		} else {
#pragma unroll
		for(int w = 0; w < EW_WW; w++){
			tile[lane][2 * w] = (uint32_t)pf.w[0][w]; tile[lane][2 * w + 1] = (uint32_t)(pf.w[0][w] >> 32);
			tile[lane][2 * EW_WW + 2 * w] = (uint32_t)pf.w[1][w]; tile[lane][2 * EW_WW + 2 * w + 1] = (uint32_t)(pf.w[1][w] >> 32);
		}
		}
		s_beg[lane] = pf.beg; s_ws[lane] = pf.ws;
		const uint32_t beg = pf.beg, ws = pf.ws, tb = pf.tb;
		const int r_own = R_hi - (int)lane;
		// lanes whose cell lies inside the target (y - d >= 0) with both rows in the tile: the others stop the walk (lane 63 always does).  Kept in
		// the two forms the lanes' tests use: the sign bit of the stop word, bit 2 of the event word
		const uint32_t TL = (lane <= 62u && r_own >= 1) ? 0u : 0x80000000u, TE = TL ? 0u : 4u;
		__syncthreads();
		const uint32_t begn = s_beg[(lane + 1u) & 63u], wsn = s_ws[(lane + 1u) & 63u];
		// the next tile starts 63 rows further up: its rows travel while this one is walked
		pf_R = R_hi - 63;
		if(pf_R >= 1) tile_fetch(pf_R, x - 63, pf, true);
		// ---- the lane's cells as 32-bit masks over query columns cbase .. cbase + 31 (cbase = its column on the walker's
		// diagonal - 16): IM = "insertion" (u3, u4) == (0, 1) on the lane's own row, DM = "deletion" (u1, u2) == (1, 0) on the
		// row above and not IM, VM = both lookups lie inside the band and inside the LDS windows.  A cell stops the diagonal
		// run when the bases differ and (IM | DM | ~VM); what ~VM stopped is looked up literally.
		const int cbase = (x - (int)lane) - 16;
		const bool pow2row = !TILED && NW <= (uint32_t)EW_WW && (NW & (NW - 1u)) == 0u;
		uint32_t IM, DM, VM;
		{
			auto window = [&](uint32_t row, int o, uint32_t wsr, uint32_t &p0w, uint32_t &p1w) -> uint32_t {   // bits o .. o + 31 of both planes of a row; returns their validity
				const uint32_t sft = (uint32_t)o & 31u;
				if constexpr(TILED){
					// EW_TD dwords per plane from dword wsr of the row's ring on (tile[row][0 ..] / tile[row][2 EW_WW ..]); bit o + c is there when its
					// dword, counted from wsr around the ring, is one of the EW_TD
					const uint32_t d0 = ((uint32_t)(o >> 5) - wsr) & (NH - 1u), d1 = (d0 + 1u) & (NH - 1u);
					const bool inA = d0 < (uint32_t)EW_TD, inB = d1 < (uint32_t)EW_TD;
					const uint32_t a0 = inA ? tile[row][d0 & 3u] : 0u, b0 = inB ? tile[row][d1 & 3u] : 0u;
					const uint32_t a1 = inA ? tile[row][2 * EW_WW + (d0 & 3u)] : 0u, b1 = inB ? tile[row][2 * EW_WW + (d1 & 3u)] : 0u;
					p0w = __builtin_amdgcn_alignbit(b0, a0, sft); p1w = __builtin_amdgcn_alignbit(b1, a1, sft);
					const uint32_t lowm = sft ? ((1u << (32u - sft)) - 1u) : ~0u;             // the bits that come from dword d0
					return (inA ? lowm : 0u) | (inB ? ~lowm : 0u);
				}
				if(pow2row){
					// the whole row is in LDS and NW is a power of two: a position outside the band is looked up where the reference's
					// unsigned arithmetic lands (plane_bit), which is then simply the position modulo BW -- the row read as a ring
					const uint32_t ia = (uint32_t)(o >> 5) & (2u * NW - 1u), ib = (ia + 1u) & (2u * NW - 1u);
					p0w = __builtin_amdgcn_alignbit(tile[row][ib], tile[row][ia], sft);
					p1w = __builtin_amdgcn_alignbit(tile[row][2 * EW_WW + ib], tile[row][2 * EW_WW + ia], sft);
					return ~0u;
				}
				const int d0 = (o >> 5) - 2 * (int)wsr;                    // dword of bit o inside the row's LDS window
				const bool inA = (uint32_t)d0 < 2u * EW_WW, inB = (uint32_t)(d0 + 1) < 2u * EW_WW;
				const uint32_t ia = (uint32_t)d0 & (2u * EW_WW - 1u), ib = (uint32_t)(d0 + 1) & (2u * EW_WW - 1u);
				const uint32_t a0 = inA ? tile[row][ia] : 0u, b0 = inB ? tile[row][ib] : 0u;
				const uint32_t a1 = inA ? tile[row][2 * EW_WW + ia] : 0u, b1 = inB ? tile[row][2 * EW_WW + ib] : 0u;
				p0w = __builtin_amdgcn_alignbit(b0, a0, sft); p1w = __builtin_amdgcn_alignbit(b1, a1, sft);
				// valid c: 0 <= o + c < BW and the bit's dword inside the window
				int lo = -o, hi = (int)BW - o;
				const int wl = 64 * (int)wsr - o, wh = wl + 64 * EW_WW;
				lo = lo > wl ? lo : wl; hi = hi < wh ? hi : wh;
				lo = lo < 0 ? 0 : lo; hi = hi > 32 ? 32 : hi;
				if(hi <= lo) return 0u;
				const uint32_t mh = hi >= 32 ? ~0u : ((1u << hi) - 1u);
				return mh & ~((1u << lo) - 1u);
			};
			uint32_t a3, a4, a1, a2;
			const uint32_t v1 = window(lane, cbase - (int)beg, ws, a3, a4);
			const uint32_t v0 = window((lane + 1u) & 63u, cbase - (int)begn, wsn, a1, a2);
			VM = v1 & v0;
			IM = ~a3 & a4;
			DM = a1 & ~a2 & ~IM;
		}
		const uint32_t SM = IM | DM | ~VM;
		const int shK = 31 + (int)lane + cbase;                       // 31 - c = shK - xs
		int k0 = 0;
		if(x - 63 < qw_lo && qw_lo > 0) q_refill(x);
		int qK = (int)lane + qw_lo;
		// ---- walk inside the tile
		while(true){
			const int xs = x + k0;
			const uint32_t sh = (uint32_t)(shK - xs);
			const uint32_t qb = (uint32_t)s_q[xs - qK];
			const uint32_t c = (31u - sh) & 31u;
			// Where the diagonal run from lane k0 ends: each lane's verdict in the sign bit of one word, one ballot.  A cell stops the run when its bases
			// differ and the masks say gap or cannot tell (outside the 32 columns: the literal step), when it is not the tile's, or left of the query;
			// lanes below k0 are behind the walk.  (All of it vector work: as three 64-bit mask operations and a select each on the scalar unit these tests were what the walk was bound by.)
			uint32_t st = sh < 32u ? SM << (sh & 31u) : 0x80000000u;
			const uint32_t nei = qb != tb ? 1u : 0u;
			st = nei ? st : 0u;
			st |= TL | (uint32_t)(xs - (int)lane);                   // (the sign of xs - lane: columns left of the query)
#ifdef EDIT_WALK_VMASK
			st = (int)lane < k0 ? 0u : st;
			const u64 stopm = __ballot((int)st < 0);                // lane 63 always stops
#else
			const u64 stopm = __builtin_amdgcn_ballot_w64((int)st < 0) & (~0ull << k0);                // lane 63 always stops; lanes below k0 masked on the scalar side (one compare instead of compare, select, 0 / 1, compare)
#endif
			// what the cell is if the walk stops on it: bit 0 insertion, bit 1 decided by the masks, bit 2 the tile's own
			const uint32_t ev = (sh < 32u ? ((IM >> c) & 1u) | (((VM >> c) & 1u) << 1) : 0u) | TE;
			const int k = __builtin_ctzll(stopm);
			const int n = k - k0;
			// n match / mismatch columns (possibly none): the count goes to the token key, the mismatches to the lanes' own counts
			vmis += (uint32_t)((int)lane - k0) < (uint32_t)n ? nei : 0u;
			key += (uint32_t)n;
			x -= n; y -= n;
			const uint32_t ek = (uint32_t)__builtin_amdgcn_readlane((int)ev, k);
			if(!(ek & 4u) || x < 0) break;                           // end of the tile / of the walk
			bool isI, isD;
			if(ek & 2u){ isI = ek & 1u; isD = !isI; }                  // (a stop inside VM is I or D)
			else {
#ifdef EDIT_DBG
				dbg_lit++;
#endif
				// literally (bsalign.h:986-1010), the lookups as plain loads: outside the band or the windows
				const long pb1 = (long)x - (long)__builtin_amdgcn_readlane((int)beg, k), pb0 = (long)x - (long)__builtin_amdgcn_readlane((int)begn, k);
				const int u3 = plane_bit((uint32_t)y + 1u, 0, pb1), u4 = plane_bit((uint32_t)y + 1u, 1, pb1);
				isI = u3 == 0 && u4 == 1; isD = false;
				if(!isI){ const int u1 = plane_bit((uint32_t)y, 0, pb0), u2 = plane_bit((uint32_t)y, 1, pb0); isD = u1 == 1 && u2 == 0; }
			}
			if(isI){
				rs.ins++; emit(1u, 1u); x--; k0 = k;
				if(x + k0 - 63 < qw_lo && qw_lo > 0){ q_refill(x); qK = (int)lane + qw_lo; }
			}
			else if(isD){ emit(2u, 1u); y--; k0 = k + 1; }
			else { vmis += (int)lane == k ? 1u : 0u; emit(0u, 1u); x--; y--; k0 = k + 1; }    // a mismatch the masks could not decide
			if(k0 > 62) break;
		}
	}
	if(!bad){
		rs.qb = x + 1; rs.tb = y + 1;
		{
			// rs.ins holds the inserted columns of the walk: the other totals follow from its two ends, the mismatches from the lanes' counts
			const int mcols = (rx - x) - rs.ins;
			uint32_t t = vmis;
			for(int o = 32; o; o >>= 1) t += (uint32_t)__shfl_xor((int)t, o);
			rs.mis = __builtin_amdgcn_readfirstlane((int)t); rs.mat = mcols - rs.mis;
			rs.del = (ry - y) - mcols;
		}
		if(rs.qb){ emit(1u, (uint32_t)rs.qb); rs.ins += rs.qb; rs.qb = 0; }
		if((type == BSA_MODE_GLOBAL || type == BSA_MODE_EXTEND) && rs.tb){ emit(2u, (uint32_t)rs.tb); rs.del += rs.tb; rs.tb = 0; }
		rs.aln = rs.mat + rs.mis + rs.ins + rs.del;
		cig_finish();
		if(type == BSA_MODE_OVERLAP) rs.score = smin + rs.te - rs.tb;
		else if(type == BSA_MODE_EXTEND) rs.score = smin;
		else rs.score = score;
	} else {
		if(lane == 0) atomicOr(&a.status[pair], BSA_ST_TRACE);
		rs.score = 0; rs.qb = rs.qe = rs.tb = rs.te = 0; rs.mat = rs.mis = rs.ins = rs.del = rs.aln = 0;
		ncig = 0;
	}
#ifdef EDIT_DBG
	if(lane == 0 && blockIdx.x < 12u) printf("walk %u (tiled %d): %u tiles, %u fetched on demand, %u literal steps, qlen %u tlen %u\n", blockIdx.x, (int)TILED, dbg_tiles, dbg_demand, dbg_lit, qlen, tlen);
#endif
	if(lane == 0){ out[pair] = rs; cig_cnt[ppos] = ncig; }
}

// stage one pair per block: query -> two bit planes (bit p of plane b = bit b of base p; zero beyond qlen),
// query and target bytes copied (the traceback compares bases), codes validated
// (TPP threads per pair: the block for few long pairs, a wave for batches of many -- the k-mer path stages 1.5 M pieces of ~26 bp)
template<int TPP>
__global__ void __launch_bounds__(256) k_edit_stage(const uint8_t *seqs, const uint64_t *qoff, const uint32_t *qlen,
		const uint64_t *toff, const uint32_t *tlen, const uint64_t *qpoff, const uint64_t *tpoff,
		const uint64_t *qboff, const uint32_t *qwords, uint8_t *qst, uint8_t *tst, u64 *qbits, uint32_t *status, uint32_t n){
	const uint32_t k = (TPP == 64) ? blockIdx.x * 4u + (threadIdx.x >> 6) : blockIdx.x, lane = threadIdx.x & (uint32_t)(TPP - 1);
	if(k >= n) return;
	const uint32_t ql = qlen[k], tl = tlen[k], nw = qwords[k];
	const uint8_t *q = seqs + qoff[k], *t = seqs + toff[k];
	uint8_t *dq = qst + qpoff[k], *dt = tst + tpoff[k];
	u64 *p0 = qbits + qboff[k], *p1 = p0 + nw;
	uint32_t bad = 0;
	// 16 bases per thread and trip: two 8-byte loads (the piece that holds the end byte by byte), the staged bytes as one 16-byte
	// store, the two planes as 16 bits each (bit 0 / bit 1 of eight bytes gathered by a multiplication)
	auto piece = [&](const uint8_t *src, uint32_t len, uint32_t i, u64 &v0, u64 &v1){
		if(i + 16u <= len){
			__builtin_memcpy(&v0, src + i, 8); __builtin_memcpy(&v1, src + i + 8, 8);
			if((v0 | v1) & 0xFCFCFCFCFCFCFCFCull){ bad = 1; v0 &= 0x0303030303030303ull; v1 &= 0x0303030303030303ull; }
		} else {
			v0 = v1 = 0;
#pragma unroll
			for(uint32_t b = 0; b < 16u; b++){
				uint8_t c = (i + b < len) ? src[i + b] : (uint8_t)0;
				if(c > 3){ bad = 1; c &= 3; }
				if(b < 8u) v0 |= (u64)c << (8u * b); else v1 |= (u64)c << (8u * (b - 8u));
			}
		}
	};
	auto gather = [](u64 v) -> uint32_t { return (uint32_t)(((v & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56); };   // bit 0 of byte j -> bit j
	const uint32_t qbytes = (ql + 16u + 15u) & ~15u, qplane = nw * 64u;
	for(uint32_t i = lane * 16u; i < max(qbytes, qplane); i += (uint32_t)TPP * 16u){
		u64 v0, v1;
		piece(q, ql, i, v0, v1);
		if(i < qbytes){ uint4 o; o.x = (uint32_t)v0; o.y = (uint32_t)(v0 >> 32); o.z = (uint32_t)v1; o.w = (uint32_t)(v1 >> 32); *(uint4*)(dq + i) = o; }
		if(i < qplane){
			((uint16_t*)p0)[i >> 4] = (uint16_t)(gather(v0) | (gather(v1) << 8));
			((uint16_t*)p1)[i >> 4] = (uint16_t)(gather(v0 >> 1) | (gather(v1 >> 1) << 8));
		}
	}
	const uint32_t tbytes = (tl + 16u + 15u) & ~15u;
	for(uint32_t i = lane * 16u; i < tbytes; i += (uint32_t)TPP * 16u){
		u64 v0, v1;
		piece(t, tl, i, v0, v1);
		uint4 o; o.x = (uint32_t)v0; o.y = (uint32_t)(v0 >> 32); o.z = (uint32_t)v1; o.w = (uint32_t)(v1 >> 32); *(uint4*)(dt + i) = o;
	}
	uint32_t st = 0;
	if((TPP == 64) ? __any((int)bad) : __syncthreads_or((int)bad)) st |= BSA_ST_BAD_BASE;
	if(ql == 0 || tl == 0) st |= BSA_ST_EMPTY;
	if(lane == 0) status[k] = st;
}

bool bsa_edit_supported_bw(uint32_t bw){       // register kernels up to 16 words, the generic kernel beyond
	return bw >= 64 && (bw % 64) == 0;
}

hipError_t bsa_launch_edit_stage(const uint8_t *seqs, const uint64_t *qoff, const uint32_t *qlen, const uint64_t *toff, const uint32_t *tlen,
		const uint64_t *qpoff, const uint64_t *tpoff, const uint64_t *qboff, const uint32_t *qwords,
		uint8_t *qst, uint8_t *tst, uint64_t *qbits, uint32_t *status, uint32_t n, hipStream_t st){
	if(n == 0) return hipSuccess;
	if(n >= 65536u) hipLaunchKernelGGL(k_edit_stage<64>, dim3((n + 3u) / 4u), dim3(256), 0, st, seqs, qoff, qlen, toff, tlen, qpoff, tpoff, qboff, qwords, qst, tst, (u64*)qbits, status, n);
	else hipLaunchKernelGGL(k_edit_stage<256>, dim3(n), dim3(256), 0, st, seqs, qoff, qlen, toff, tlen, qpoff, tpoff, qboff, qwords, qst, tst, (u64*)qbits, status, n);
	return hipGetLastError();
}

// Row format 1 (tiled, bsa_common.h) is a contract between ONE forward kernel and ONE traceback kernel: k_edit_fwd_grp32 and k_edit_trace_wave.  It is
// taken for a launch class that both launchers below give to exactly those two -- a band of 64 / 128 / 256 columns, few enough pairs for a walk per
// wave -- and that fills a chunk alone (bsa_edit_run sets EditArgs::row_fmt from this for the forward and the traceback launch of the chunk alike).
// BSA_EDIT_TILED=0 keeps format 0; any of the kernel-choice knobs does as well.
bool bsa_edit_tiled_ok(uint32_t bw, uint32_t count, int mode){
	(void)mode;
	const char *te = bsa_env("BSA_EDIT_TILED");
	if(te && te[0] == '0') return false;
	if(bsa_env("BSA_EDIT_GRP") || bsa_env("BSA_EDIT_GRP32") || bsa_env("BSA_EDIT_TRACE_WAVE") || bsa_env("BSA_EDIT_TRACE_LANES") || bsa_env("BSA_EDIT_TRACE_COOP") || bsa_env("BSA_EDIT_FWD_LANES")) return false;
	const uint32_t nw = bw / 64u;
	if(!(bw == 64u || bw == 128u || bw == 256u) || count == 0u) return false;
	const uint32_t G32 = nw <= 1u ? 2u : nw <= 2u ? 4u : 8u;
	if((uint64_t)count * G32 / 64u > 65536u) return false;                          // k_edit_fwd_grp32's own condition
	int dev = 0, cus = 256;
	if(hipGetDevice(&dev) == hipSuccess){ int v = 0; if(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
	return count <= 8u * (uint32_t)cus * 32u;                                         // k_edit_trace_wave's (narrow bands)
}

hipError_t bsa_launch_edit_fwd(const EditArgs &a, hipStream_t st){
	if(a.count == 0) return hipSuccess;
	// pairs per wave of the register kernels (BSA_EDIT_FWD_LANES overrides, for measurements).  64 is best: a wave's time
	// is its own serial chain (~600 instructions per row of 64-bit funnel shifts and block updates) whatever its lane
	// count -- 16384 pairs x 100 kbp, ms per launch: 64 lanes 102, 32 -> 104, 16 -> 136, 8 -> 261, 4 -> 374
	uint32_t lanes = 64;
	if(const char *e = bsa_env("BSA_EDIT_FWD_LANES")){ const int v = atoi(e); if(v >= 1 && v <= 64) lanes = (uint32_t)v; }
	const uint32_t fblocks = (a.count + lanes - 1) / lanes;
#define EDIT_CASE(N) case N: if(track) hipLaunchKernelGGL((k_edit_fwd<N, true>), dim3(fblocks), dim3(64), 0, st, a, lanes); \
		else hipLaunchKernelGGL((k_edit_fwd<N, false>), dim3(fblocks), dim3(64), 0, st, a, lanes); break;
	const bool track = (a.mode & 3) != BSA_MODE_GLOBAL;
	// few pairs: G lanes per pair (waves stay few, every row is a third of the serial chain); many pairs: one per lane
	{
		const uint32_t nw = a.bw / 64u;
		const uint32_t G = nw <= 2u ? 2u : nw <= 4u ? 4u : nw <= 8u ? 8u : 16u;
		bool grp = a.bw != 0u && nw >= 2u && nw <= 16u && (uint64_t)a.count * G / 64u <= 4096u;
		if(const char *e = bsa_env("BSA_EDIT_GRP")) grp = a.bw != 0u && nw >= 2u && nw <= 16u && e[0] == '1';
		// 32-bit words, twice the lanes per pair (bands up to 512 columns): BSA_EDIT_GRP32=0/1 overrides
		{
			const uint32_t G32 = nw <= 1u ? 2u : nw <= 2u ? 4u : nw <= 4u ? 8u : 16u;
			bool g32 = a.bw != 0u && nw >= 1u && nw <= 8u && (uint64_t)a.count * G32 / 64u <= 65536u;      // (faster than the other two kernels on every shape of tools/edit_shapes.sh)
			if(const char *e = bsa_env("BSA_EDIT_GRP32")) g32 = a.bw != 0u && nw >= 1u && nw <= 8u && e[0] == '1';
			if(bsa_env("BSA_EDIT_GRP")) g32 = false;
			if(g32){
				bsa_last_fwd_kernel = "k_edit_fwd_grp32 (forward DP, 32-bit words, 2 NW lanes per pair)";
				const uint32_t ppw = 64u / G32, gblocks = ((a.count + ppw - 1) / ppw + 3) / 4;
				if(a.row_fmt == 1u){
					bsa_last_fwd_kernel = "k_edit_fwd_grp32 (forward DP, 32-bit words, 2 NW lanes per pair, rows tiled eight at a time)";
					switch(G32){
						case 2: hipLaunchKernelGGL((k_edit_fwd_grp32<2, true>), dim3(gblocks), dim3(256), 0, st, a); break;
						case 4: hipLaunchKernelGGL((k_edit_fwd_grp32<4, true>), dim3(gblocks), dim3(256), 0, st, a); break;
						default: hipLaunchKernelGGL((k_edit_fwd_grp32<8, true>), dim3(gblocks), dim3(256), 0, st, a); break;
					}
					return hipGetLastError();
				}
				switch(G32){
					case 2: hipLaunchKernelGGL((k_edit_fwd_grp32<2, false>), dim3(gblocks), dim3(256), 0, st, a); break;
					case 4: hipLaunchKernelGGL((k_edit_fwd_grp32<4, false>), dim3(gblocks), dim3(256), 0, st, a); break;
					case 8: hipLaunchKernelGGL((k_edit_fwd_grp32<8, false>), dim3(gblocks), dim3(256), 0, st, a); break;
					default: hipLaunchKernelGGL((k_edit_fwd_grp32<16, false>), dim3(gblocks), dim3(256), 0, st, a); break;
				}
				return hipGetLastError();
			}
		}
		if(grp){
			bsa_last_fwd_kernel = "k_edit_fwd_grp (forward DP, NW lanes per pair)";
			const uint32_t ppw = 64u / G, gblocks = ((a.count + ppw - 1) / ppw + 3) / 4;
			switch(G){
				case 2: hipLaunchKernelGGL((k_edit_fwd_grp<2>), dim3(gblocks), dim3(256), 0, st, a); break;
				case 4: hipLaunchKernelGGL((k_edit_fwd_grp<4>), dim3(gblocks), dim3(256), 0, st, a); break;
				case 8: hipLaunchKernelGGL((k_edit_fwd_grp<8>), dim3(gblocks), dim3(256), 0, st, a); break;
				default: hipLaunchKernelGGL((k_edit_fwd_grp<16>), dim3(gblocks), dim3(256), 0, st, a); break;
			}
			return hipGetLastError();
		}
	}
	bsa_last_fwd_kernel = a.bw ? "k_edit_fwd (forward DP, one pair per lane)" : "k_edit_fwd_wide / k_edit_fwd_gen (forward DP, bands above 1024 columns)";
	switch(a.bw / 64){
		EDIT_CASE(1) EDIT_CASE(2) EDIT_CASE(3) EDIT_CASE(4) EDIT_CASE(5) EDIT_CASE(6) EDIT_CASE(7) EDIT_CASE(8)
		EDIT_CASE(9) EDIT_CASE(10) EDIT_CASE(11) EDIT_CASE(12) EDIT_CASE(13) EDIT_CASE(14) EDIT_CASE(15) EDIT_CASE(16)
		default: {
			// bands above 1024: static ones (overlap / extend / bandwidth 0) one pair per wave, the rest one pair per lane
			if(a.wide == 1u) hipLaunchKernelGGL((k_edit_fwd_wide<1>), dim3((a.count + 3) / 4), dim3(256), 0, st, a);
			else if(a.wide == 2u) hipLaunchKernelGGL((k_edit_fwd_wide<2>), dim3((a.count + 3) / 4), dim3(256), 0, st, a);
			else if(a.wide == 4u) hipLaunchKernelGGL((k_edit_fwd_wide<4>), dim3((a.count + 3) / 4), dim3(256), 0, st, a);
			else if(a.wide == 8u) hipLaunchKernelGGL((k_edit_fwd_wide<8>), dim3((a.count + 3) / 4), dim3(256), 0, st, a);
			if(a.wide == 0u || ((a.mode & 3) == BSA_MODE_GLOBAL && a.bandwidth != 0u)){
				uint32_t gl = 64;
				while(gl > 2u && (a.count + gl / 2 - 1) / (gl / 2) <= 8192u) gl >>= 1;
				if(const char *e = bsa_env("BSA_EDIT_GEN_LANES")){ const int v = atoi(e); if(v >= 1 && v <= 64) gl = (uint32_t)v; }
				hipLaunchKernelGGL(k_edit_fwd_gen, dim3((a.count + gl - 1) / gl), dim3(64), 0, st, a, gl);
			}
		} break;
	}
#undef EDIT_CASE
	return hipGetLastError();
}

hipError_t bsa_launch_edit_trace(const EditArgs &a, bsa_result_t *out, uint32_t *cig_cnt, hipStream_t st){
	if(a.count == 0) return hipSuccess;
	// fewest pairs per wave that keep every wave resident, a power of two in [2, 64]; up to 8 pairs per wave the idle
	// lanes prefetch rows for the walking ones (COOP)
	int dev = 0, cus = 256;
	if(hipGetDevice(&dev) == hipSuccess){
		int v = 0;
		if(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
	}
	static int per_cu[2] = {0, 0};         // resident blocks (= waves) per CU of the two variants
	if(per_cu[0] == 0){
		int v = 0;
		per_cu[0] = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, k_edit_trace<false>, 64, 0) == hipSuccess && v > 0) ? v : 32;
		per_cu[1] = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, k_edit_trace<true>, 64, 0) == hipSuccess && v > 0) ? v : 32;
		(void)hipGetLastError();
	}
	// few long walks: one walk per wave (k_edit_trace_wave); BSA_EDIT_TRACE_WAVE=0/1 overrides
	{
		// (measured, tools/edit_shapes.sh: 65536 x 20 kbp at bandwidth 256 16.9 ms against 23.5 pair-per-lane, at bandwidth 512 -- rows
		// wider than the LDS window -- 43 against 22; 262144 x 3 kbp 10.1 against 8.5)
		const bool narrow = a.bw != 0u && a.bw <= 64u * EW_WW;
		bool wave = a.count <= (narrow ? 8u : 2u) * (uint32_t)cus * 32u;
		if(const char *e = bsa_env("BSA_EDIT_TRACE_WAVE")) wave = e[0] == '1';
		if(bsa_env("BSA_EDIT_TRACE_LANES") || bsa_env("BSA_EDIT_TRACE_COOP")) wave = false;
		if(wave){
			bsa_last_trace_kernel = a.row_fmt == 1u ? "k_edit_trace_wave (rows tiled eight at a time)" : "k_edit_trace_wave";
			if(a.row_fmt == 1u) hipLaunchKernelGGL(k_edit_trace_wave<true>, dim3(a.count), dim3(64), 0, st, a, out, cig_cnt);
			else hipLaunchKernelGGL(k_edit_trace_wave<false>, dim3(a.count), dim3(64), 0, st, a, out, cig_cnt);
			return hipGetLastError();
		}
	}
	const char *ce = bsa_env("BSA_EDIT_TRACE_COOP");
	const bool coop_ok = !(ce && ce[0] == '0');
	uint32_t lanes = 2;
	const uint32_t slots_c = (uint32_t)cus * (uint32_t)per_cu[1], slots_p = (uint32_t)cus * (uint32_t)per_cu[0];
	while(lanes < 64u && (a.count + lanes - 1) / lanes > (lanes <= 8u && coop_ok ? slots_c : slots_p)) lanes <<= 1;
	if(const char *e = bsa_env("BSA_EDIT_TRACE_LANES")){ const int v = atoi(e); if(v >= 1 && v <= 64 && (v & (v - 1)) == 0) lanes = (uint32_t)v; }
	const uint32_t blocks = (a.count + lanes - 1) / lanes;
	bsa_last_trace_kernel = "k_edit_trace";
	if(lanes <= 8u && lanes >= 2u && coop_ok) hipLaunchKernelGGL(k_edit_trace<true>, dim3(blocks), dim3(64), 0, st, a, out, cig_cnt, lanes);
	else hipLaunchKernelGGL(k_edit_trace<false>, dim3(blocks), dim3(64), 0, st, a, out, cig_cnt, lanes);
	return hipGetLastError();
}
