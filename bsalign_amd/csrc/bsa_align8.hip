// bsa_align8.hip -- 8-bit banded striped pairwise DP on gfx950 (MI355X), lane-exact.
//
// Replaces, for a batch of independent pairs, the reference's per-pair row loop
//   banded_striped_epi8_seqalign_pairwise            /root/reference/bsalign.h:3854-4050
//     _piecex_row_init :2094   _piecex_row_movx :2244   _piece{0,1,2}_row_cal :2727/:2885/:3084
//     _row_cal_FPenetration_codes :2639   _row_cal_tail_codes :2618   _band_mov :3331
//     _piecex_backcal[_cell] :3667-3852   _getscore :3187   _row_max :3213
//
// Mapping.  The reference's SSE word has 16 int8 lanes; lane j owns "running block" j = W consecutive
// band cells (W = bandwidth / 16).  A DPP row on CDNA is exactly 16 lanes, so one pair occupies one DPP
// row: lane j keeps its W cells of u / e / q in VGPRs (one cell per register, int32 holding an int8
// value), the SSE byte shifts become row_shr/row_shl DPP moves, and one wave64 advances 4 pairs per row
// step.  Every saturating int8 operation of the reference is reproduced as add + v_med3_i32 clamp, every
// int->int8 truncation as v_bfe_i32, in the reference's order, so results are bit-identical including
// the saturation corner cases.  The row loop is strictly serial per pair (the next band offset depends
// on the whole row), parallelism is 16 lanes x W cells x batch.
//
// HBM traffic per row and pair: one row record (see bsa_common.h) = (pw+1)*bw + 72 bytes written once;
// the traceback kernel re-derives the path from those records exactly like the reference's backcal.
#include "bsa_common.h"
#include "bsa_dpp.h"
#include <cstdlib>

template<int W> struct QCodes { uint32_t w[(W + 3) / 4]; };

template<int W>
static __device__ __forceinline__ QCodes<W> load_qcodes(const uint8_t *p){
	QCodes<W> q;
	if constexpr (W >= 4){
		__builtin_memcpy(q.w, p, W);            // unaligned W-byte load (W multiple of 4)
	} else if constexpr (W == 2){
		uint16_t v; __builtin_memcpy(&v, p, 2); q.w[0] = v | 0x04040000u;
	} else {
		q.w[0] = p[0] | 0x04040400u;
	}
	return q;
}

template<int W, int PW>
__global__ void __launch_bounds__(256) k_align8_fwd(const Align8Args a){
	constexpr int BW = W * 16;
	extern __shared__ __attribute__((aligned(16))) int8_t smem[];   // slow-path movx scratch: per group (PW+1)*BW bytes + 17 ints
	constexpr int GROUP_LDS = ((PW + 1) * BW + 17 * 4 + 15) & ~15;
	const int lt = threadIdx.x;
	const int j = lt & 15;
	const uint32_t g = (blockIdx.x * 256u + lt) >> 4;
	const bool live = g < a.count;
	const uint32_t ppos = a.first + (live ? g : 0u);
	const uint32_t pair = a.order[ppos];
	const uint32_t qlen = a.qlen[pair];
	uint32_t tlen = a.tlen[pair];
	const uint8_t *qp = a.qst + a.qpoff[pair];
	const uint8_t *tp = a.tst + a.tpoff[pair];
	int *begs = (int*)(a.rows + a.slot_off[ppos]);
	uint8_t *rowp = (uint8_t*)begs + bsa_begs_bytes(tlen);
	if(!live || a.status[pair] != 0u) tlen = 0;
	int8_t *gl = smem + (lt >> 4) * GROUP_LDS;
	int8_t *su = gl, *se = gl + BW, *sq = gl + 2 * BW;
	int *sub = (int*)(gl + (PW + 1) * BW);

	const int mode = a.mode & 3;
	const int gapo1 = a.gapo1, gape1 = a.gape1, gapo2 = a.gapo2, gape2 = a.gape2;
	const int GapE = trunc8(gape1), GapOE = trunc8(gapo1 + gape1);
	const int GapP = trunc8(gape2), GapQP = trunc8(gapo2 + gape2);
	const int GapOQ = sat8(GapOE - GapQP);
	// synthetic values for cells entering the band at its right end (bsalign.h:2357-2369)
	const int cfirst = (PW == 2) ? (min(a.smin, gapo2 + gape2) - 1 - a.smax + (gapo2 + gape2))
	                             : (min(a.smin, gapo1 + gape1) - 1 - a.smax + (gapo1 + gape1));
	const int dsw = (PW == 2) ? (gapo1 - gapo2) / (gape2 - gape1) : (BW + 1);
	auto newcell_int = [&](int k) -> int { return (k == 0) ? cfirst : ((PW == 2 && k >= dsw) ? gape2 : gape1); };
	auto newcell_cum = [&](int n) -> int {   // sum of the first n (>= 1) synthetic cells, as the reference's running int `c`
		int n1 = min(n, dsw);
		return cfirst + (n1 - 1) * gape1 + ((PW == 2) ? max(0, n - dsw) * gape2 : 0);
	};

	int u[W], e[W], q2[W];
	int ubA, ubB;
	// ---- row -1 (bsalign.h:2094-2140)
	{
		int bs = 0;
		const int first = trunc8(gapo1 + gape1 + a.smin - a.smax);
		const int xp = (PW == 2) ? (gapo2 - gapo1) / (gape1 - gape2) : 0;
#pragma unroll
		for(int i = 0; i < W; i++){
			int p = j * W + i, v;
			if(mode == BSA_MODE_OVERLAP) v = 0;
			else if(p == 0) v = first;
			else if(PW == 2) v = (p < xp) ? gape1 : gape2;
			else v = gape1;
			u[i] = v; bs += v;
			e[i] = BSA_EPI8_MIN; q2[i] = BSA_EPI8_MIN;
		}
		int inc = row_iscan16(bs);
		int base0 = (mode == BSA_MODE_OVERLAP) ? 0 : (a.smax - a.smin);
		ubB = base0 + inc;
		ubA = ubB - bs;
	}
	constexpr uint32_t CELLS = ((uint32_t)(PW + 1) * W + 3u) & ~3u, BLK = CELLS + 4u;
	constexpr int RW = (int)(BLK / 4u), TG = (64u / BLK) ? (int)(64u / BLK) : 1;
	TileWriter<RW, TG> tw;
	auto store_row = [&](uint32_t row_index, uint32_t rbeg_v, bool active){
		// block record of this lane (bsa_common.h): u / e / q bytes then ubegs[j]
		uint32_t rec[RW];
#pragma unroll
		for(int d = 0; d < RW; d++) rec[d] = 0u;
#pragma unroll
		for(int k = 0; k < W; k++){
			rec[k >> 2] |= (uint32_t)(u[k] & 0xff) << (8 * (k & 3));
			if(PW >= 1) rec[(W + k) >> 2] |= (uint32_t)(e[k] & 0xff) << (8 * ((W + k) & 3));
			if(PW == 2) rec[(2 * W + k) >> 2] |= (uint32_t)(q2[k] & 0xff) << (8 * ((2 * W + k) & 3));
		}
		rec[RW - 1] = (uint32_t)ubA;
		tw.push(rowp, (uint32_t)j, row_index, active, row_index == tlen, rec);
		if(active && j == 0) begs[row_index] = (int)rbeg_v;
	};
	store_row(0u, 0u, tlen != 0);

	uint32_t rbeg = 0, mov = 0, i = 0;
	int tb_next = tlen ? (int)tp[0] : 0;
	const double dqlen = (double)qlen, dtlen = (double)tlen;

	while(__any(i < tlen)){
		const bool act = i < tlen;
		// ---- band offset of this row (bsalign.h:3932-3946)
		const bool moved = (mov != 0u) && (rbeg + BW < qlen);
		{
			uint32_t room = qlen - (rbeg + BW);
			mov = moved ? min(mov, room) : 0u;
			rbeg += mov;
		}
		int rh;
		if(rbeg) rh = BSA_SCORE_MIN;
		else if(mode == BSA_MODE_OVERLAP || i == 0) rh = 0;
		else if(PW < 2) rh = gapo1 + gape1 * (int)i;
		else rh = max(gapo1 + gape1 * (int)i, gapo2 + gape2 * (int)i);
		// ---- row_movx (bsalign.h:2244-2392)
		if(__any(act && mov >= (uint32_t)W)){
			// generic path through LDS, any movx (rare: band jumps by >= W cells)
#pragma unroll
			for(int k = 0; k < W; k++){
				su[j * W + k] = (int8_t)u[k];
				if(PW >= 1) se[j * W + k] = (int8_t)e[k];
				if(PW == 2) sq[j * W + k] = (int8_t)q2[k];
			}
			sub[j] = ubA; if(j == 15) sub[16] = ubB;
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			if(mov){
				// H(rbeg-1, y-1) = getscore(previous row, mov-1) (bsalign.h:3935)
				uint32_t pp = min(mov - 1u, (uint32_t)BW - 1u), yy = pp / W, xx = pp % W;
				int s = sub[yy];
				for(uint32_t k = 0; k <= xx; k++) s += su[yy * W + k];
				rh = s;
			}
			if(mov >= (uint32_t)BW){
#pragma unroll
				for(int k = 0; k < W; k++){ u[k] = 0; e[k] = 0; q2[k] = 0; }
				ubA = ubB = BSA_SCORE_MIN;
			} else if(mov){
				const uint32_t cyc = mov / W, m = mov % W, p0 = BW - mov;
#pragma unroll
				for(int k = 0; k < W; k++){
					uint32_t src = j * W + k + mov;
					if(src < (uint32_t)BW){
						u[k] = su[src];
						if(PW >= 1) e[k] = se[src];
						if(PW == 2) q2[k] = sq[src];
					} else {
						u[k] = trunc8(newcell_int((int)(src - BW)));
						e[k] = 0; q2[k] = 0;
					}
				}
				auto new_ub = [&](uint32_t idx) -> int {      // ubegs[idx] of the moved row, idx in 0..16
					int v;
					if(idx + cyc < 16u){
						uint32_t l = idx + cyc;
						v = sub[l];
						for(uint32_t k = 0; k < m; k++) v += su[l * W + k];
					} else v = sub[16];
					int nbefore = (int)(idx * W) - (int)p0;  // synthetic cells in blocks < idx
					if(nbefore > 0) v += newcell_cum(nbefore);
					return v;
				};
				ubA = new_ub((uint32_t)j);
				ubB = new_ub((uint32_t)j + 1u);
			}
			__builtin_amdgcn_wave_barrier();
		} else {
			// band slides by mov < W cells: `mov` predicated single-cell shifts (registers + one DPP per array)
			int bacc = 0;
			for(uint32_t s = 0; __any(act && s < mov); s++){
				const bool d = s < mov;
				const int nci = newcell_int((int)s);
				const int dropped = u[0];
				const int in_u = DPP_SHL(trunc8(nci), u[0], 1);   // lane 15 receives the synthetic cell
#pragma unroll
				for(int k = 0; k + 1 < W; k++) u[k] = d ? u[k + 1] : u[k];
				u[W - 1] = d ? in_u : u[W - 1];
				if(PW >= 1){
					const int in_e = DPP_SHL(0, e[0], 1);
#pragma unroll
					for(int k = 0; k + 1 < W; k++) e[k] = d ? e[k + 1] : e[k];
					e[W - 1] = d ? in_e : e[W - 1];
				}
				if(PW == 2){
					const int in_q = DPP_SHL(0, q2[0], 1);
#pragma unroll
					for(int k = 0; k + 1 < W; k++) q2[k] = d ? q2[k + 1] : q2[k];
					q2[W - 1] = d ? in_q : q2[W - 1];
				}
				ubA += d ? dropped : 0;
				bacc += d ? nci : 0;
			}
			const int nb = DPP_SHL(ubB + bacc, ubA, 1);            // ubegs[j+1] of the moved row; lane 15: old end + synthetic sum
			ubB = mov ? nb : ubB;
			if(mov) rh = ubA;                                      // lane 0: getscore(prev, mov-1) == new ubegs[0]
		}
		// ---- sequences for this row
		const int tb = tb_next;
		if(act && i + 1 < tlen) tb_next = tp[i + 1];
		QCodes<W> qc;
		if(act) qc = load_qcodes<W>(qp + rbeg + j * W);
		else { for(int n = 0; n < (W + 3) / 4; n++) qc.w[n] = 0x04040404u; }
		const uint32_t mr = (tb == 0) ? a.mrow[0] : (tb == 1) ? a.mrow[1] : (tb == 2) ? a.mrow[2] : a.mrow[3];
		uint32_t s4[(W + 3) / 4];
#pragma unroll
		for(int n = 0; n < (W + 3) / 4; n++) s4[n] = __builtin_amdgcn_perm(0xC1C1C1C1u, mr, qc.w[n]);
#define SCORE(ii) __builtin_amdgcn_sbfe((int)s4[(ii) >> 2], 8 * ((ii) & 3), 8)
		// ---- row_cal (bsalign.h:2727-2793 / 2885-2960 / 3084-3179)
		int h0;
		{
			int hh = (rh - ubA) + SCORE(0);
			int t0 = u[0] + ((PW == 0) ? gape1 : (PW == 1) ? e[0] : max(e[0], q2[0]));
			hh = (hh >= t0) ? min(hh, BSA_EPI8_MAX) : BSA_EPI8_MIN;
			h0 = trunc8(hh);
		}
		int f = BSA_EPI8_MIN, gq = BSA_EPI8_MIN;
		{
			int hc = (j == 0) ? h0 : SCORE(0);
#pragma unroll
			for(int k = 0; k < W; k++){
				const int uk = u[k];
				int h;
				if(PW == 0){
					int ee = sat8(uk + GapE);
					h = max(max(ee, hc), f);
					f = sat8(sat8(h + GapE) - uk);
				} else if(PW == 1){
					int ee = sat8(e[k] + uk);
					h = max(max(ee, hc), f);
					f = sat8(f + GapE);
					h = sat8(h + GapOE);
					f = sat8(max(f, h) - uk);
				} else {
					int ee = sat8(e[k] + uk), qq = sat8(q2[k] + uk);
					h = max(max(ee, hc), max(qq, max(f, gq)));
					f = sat8(f + GapE);
					h = sat8(h + GapOE);
					f = sat8(max(f, h) - uk);
					gq = sat8(gq + GapP);
					h = sat8(h - GapOQ);
					gq = sat8(max(gq, h) - uk);
				}
				if(k + 1 < W) hc = SCORE(k + 1);
			}
		}
#ifdef BSA_DEBUG
		if(g == 0 && i == 0 && j < 3) printf("lane %d row %u: tb %d mr %08x qc %08x s4 %08x S0 %d S1 %d h0 %d rh %d ubA %d ubB %d f(pass1) %d u0 %d e0 %d\n", j, i, tb, mr, qc.w[0], s4[0], SCORE(0), SCORE(1), h0, rh, ubA, ubB, f, u[0], e[0]);
#endif
		f = fpen(f, ubA, ubB, W * gape1, j);
		if(PW == 2) gq = fpen(gq, ubA, ubB, W * gape2, j);
#ifdef BSA_DEBUG
		if(g == 0 && i == 0 && j < 3) printf("lane %d after fpen f %d\n", j, f);
#endif
		int htail, ulast;
		{
			int v = 0, z = (j == 0) ? h0 : SCORE(0), h = 0;
			ulast = 0;
#pragma unroll
			for(int k = 0; k < W; k++){
				const int uk = u[k];
				if(PW == 0){
					int ee = sat8(uk + GapE);
					h = max(max(ee, z), f);
					u[k] = sat8(h - v);
					v = sat8(h - uk);
					f = sat8(sat8(h + GapE) - uk);
				} else if(PW == 1){
					int ee = sat8(e[k] + uk);
					h = max(max(ee, z), f);
					u[k] = sat8(h - v);
					v = sat8(h - uk);
					ee = sat8(ee + GapE); ee = sat8(ee - h); e[k] = max(ee, GapOE);
					f = sat8(f + GapE);
					h = sat8(h + GapOE);
					f = sat8(max(f, h) - uk);
				} else {
					int ee = sat8(e[k] + uk), qq = sat8(q2[k] + uk);
					h = max(max(ee, z), max(qq, max(f, gq)));
					u[k] = sat8(h - v);
					v = sat8(h - uk);
					ee = sat8(ee + GapE); ee = sat8(ee - h); e[k] = max(ee, GapOE);
					qq = sat8(qq + GapP); qq = sat8(qq - h); q2[k] = max(qq, GapQP);
					f = sat8(f + GapE);
					h = sat8(h + GapOE);
					f = sat8(max(f, h) - uk);
					gq = sat8(gq + GapP);
					h = sat8(h - GapOQ);
					gq = sat8(max(gq, h) - uk);
				}
				ulast = uk;
				if(k + 1 < W) z = SCORE(k + 1);
			}
			htail = (PW == 0) ? h : (PW == 1) ? sat8(h - GapOE) : sat8(h - GapQP);
		}
#undef SCORE
		// ---- tail (bsalign.h:2618-2636)
		{
			const int vlast = sat8(htail - ulast);
			const int nB = ubB + vlast;                       // ubegs[j+1] += v at the end of block j
			const int vsh = DPP_SHR(0, vlast, 1);
#ifdef BSA_DEBUG
			if(g == 0 && i == 0 && j < 3) printf("lane %d tail: htail %d ulast %d vlast %d vsh %d u0 %d u1 %d f %d\n", j, htail, ulast, vlast, vsh, u[0], u[1], f);
#endif
			u[0] = sat8(u[0] - vsh);
			int nA = DPP_SHR(0, nB, 1);
			if(j == 0){ nA = ubA + u[0]; u[0] = 0; }         // re-base: ubegs[0] = H(0), u[0] = 0
			ubA = nA; ubB = nB;
		}
		store_row(i + 1u, rbeg, act);
		// ---- adaptive band (bsalign.h:3331-3349) + global steering (bsalign.h:4006-4021)
		{
			int dsum = ubB - ubA; dsum = dsum < 0 ? -dsum : dsum;
			const int nzsum = row_sum16(dsum);
			const int ub0 = DPP_BCAST(ubA, 0), ub16 = DPP_BCAST(ubB, 15);
			uint32_t nz = (uint32_t)(nzsum / 16);
			nz = nz / (uint32_t)W * 16u / 2u;
			const int noisy = (int)((16u > nz) ? 16u : nz);
			int rbx;
			if(i <= (uint32_t)BW / 4u) rbx = 0;
			else if(rbeg + BW >= qlen) rbx = 0;
			else if(ub0 + noisy < ub16) rbx = 2;
			else if(ub0 > ub16 + noisy) rbx = 0;
			else rbx = 1;
			if(mode == BSA_MODE_GLOBAL){
				const int rbz = 2 * max((int)(tlen / max(qlen, 1u)), 1);
				const int rby = (int)((1.0 * (double)i / dtlen) * dqlen);
				const uint32_t left = tlen - i - 1u;
				if((long long)rbeg + (long long)rbz * (long long)left + (long long)BW <= (long long)(uint32_t)(qlen + (uint32_t)rbz - 1u)){
					mov = 1u + (uint32_t)(qlen - (rbeg + BW)) / max(left, 1u);
				} else if((int)rbeg < rby - BW){
					mov = (uint32_t)(rbx + 1);
				} else if((int)rbeg > rby){
					mov = (uint32_t)max(0, rbx - 1);
				} else mov = (uint32_t)rbx;
			} else mov = (uint32_t)rbx;
		}
		i++;
	}
}

// ---------------------------------------------------------------------------------------------
// Traceback: literal restatement of banded_striped_epi8_seqalign_piecex_backcal (bsalign.h:3704-3852)
// over the stored row records; also the end-score selection of the driver (bsalign.h:4023-4045).
// One thread per pair.  CIGAR words are produced back-to-front, so they are written downwards from the
// end of the pair's (already consumed) row slot and come out in forward order.
// ---------------------------------------------------------------------------------------------
struct RowView {
	const uint8_t *rows; const int *begs; uint32_t bw, W, cells, blk, tg, tileb; int pw;
	__device__ __forceinline__ void init(const uint8_t *slot, uint32_t tlen, uint32_t bw_, int pw_){
		begs = (const int*)slot; rows = slot + bsa_begs_bytes(tlen);
		bw = bw_; W = bw_ / 16; pw = pw_; cells = bsa_blk_cells(W, pw_); blk = cells + 4u;
		tg = bsa_tile_rows(W, pw_); tileb = bsa_tile_bytes(W, pw_);
	}
	__device__ __forceinline__ const int8_t* blkp(int row, uint32_t y) const {
		const uint32_t rr = (uint32_t)(row + 1);
		return (const int8_t*)(rows + ((size_t)(rr / tg) * 16u + y) * tileb + (rr % tg) * blk);
	}
	__device__ __forceinline__ const uint8_t* tilep(int row, uint32_t y) const { return rows + ((size_t)((uint32_t)(row + 1) / tg) * 16u + y) * tileb; }
	__device__ __forceinline__ int ubv(int row, uint32_t y) const { return *(const int*)((const uint8_t*)blkp(row, y) + cells); }
	__device__ __forceinline__ int beg(int row) const { return begs[row + 1]; }
	__device__ __forceinline__ uint32_t* cigar_end(uint32_t tlen) const { return (uint32_t*)(const_cast<uint8_t*>(rows) + bsa_groups(tlen, tg) * 16 * (size_t)tileb); }
	__device__ __forceinline__ int getscore(int row, long pos) const {       // bsalign.h:3187-3197
		uint32_t p = (uint32_t)pos;
		if(p >= bw) p = bw - 1;                                              // keep reads inside the record
		const uint32_t y = p / W, x = p % W;
		const int8_t *us = blkp(row, y);
		int s = *(const int*)((const uint8_t*)us + cells);
		for(uint32_t k = 0; k <= x; k++) s += us[k];
		return s;
	}
	__device__ __forceinline__ int mtx_getscore(int row, int col) const { return getscore(row, (long)col - beg(row)); }
};

static __device__ uint32_t row_max_dev(const RowView &R, int row, int *max_score){ // bsalign.h:3213-3329
	const uint32_t W = R.W, STEP = 32;
	int lmax[16]; uint32_t lchunk[16];
	for(uint32_t l = 0; l < 16; l++){
		const int8_t *us = R.blkp(row, l);
		int base = R.ubv(row, l);
		lmax[l] = BSA_SCORE_MIN; lchunk[l] = 0;
		for(uint32_t i = 0, c = 0; i < W; i += STEP, c++){
			uint32_t n = (i + STEP < W) ? STEP : W - i;
			int run = 0, cmax = -32767;
			for(uint32_t x = 0; x < n; x++){
				run += us[i + x];
				run = min(max(run, -32768), 32767);
				cmax = max(cmax, run);
			}
			if(base + cmax > lmax[l]){ lmax[l] = base + cmax; lchunk[l] = c; }
			base += run;
		}
	}
	int best = 0, lane = 0;
	{
		int mm[4], ii[4];
		for(int k = 0; k < 4; k++){
			int m01, i01, m23, i23;
			if(lmax[4 + k] > lmax[k]){ m01 = lmax[4 + k]; i01 = 4 + k; } else { m01 = lmax[k]; i01 = k; }
			if(lmax[12 + k] > lmax[8 + k]){ m23 = lmax[12 + k]; i23 = 12 + k; } else { m23 = lmax[8 + k]; i23 = 8 + k; }
			if(m23 > m01){ mm[k] = m23; ii[k] = i23; } else { mm[k] = m01; ii[k] = i01; }
		}
		best = mm[0]; lane = ii[0];
		for(int k = 1; k < 4; k++) if(mm[k] > best){ best = mm[k]; lane = ii[k]; }
	}
	*max_score = best;
	uint32_t x = lchunk[lane] * STEP, y = min(x + STEP, W), jj = x;
	int umax = BSA_SCORE_MIN, uscr = 0;
	const int8_t *usl = R.blkp(row, (uint32_t)lane);
	for(; x < y; x++){
		uscr += usl[x];
		if(uscr > umax){ jj = x; umax = uscr; }
	}
	return (uint32_t)lane * W + jj;
}

__global__ void __launch_bounds__(64) k_align8_backcal(const Align8Args a, int pw, bsa_result_t *out, uint32_t *cig_cnt){
	const uint32_t g = blockIdx.x * 64u + threadIdx.x;
	if(g >= a.count) return;
	const uint32_t ppos = a.first + g;
	const uint32_t pair = a.order[ppos];
	bsa_result_t rs;
	rs.score = 0; rs.qb = rs.qe = rs.tb = rs.te = 0; rs.mat = rs.mis = rs.ins = rs.del = rs.aln = 0;
	if(a.status[pair] != 0u){ out[pair] = rs; cig_cnt[ppos] = 0; return; }
	const uint32_t qlen = a.qlen[pair], tlen = a.tlen[pair];
	const uint8_t *qseq = a.qst + a.qpoff[pair];
	const uint8_t *tseq = a.tst + a.tpoff[pair];
	RowView R;
	const uint32_t bw = a.bw ? a.bw : ((qlen + 15u) / 16u * 16u);     // bandwidth 0 = the whole (rounded) query (bsalign.h:3861-3862)
	R.init(a.rows + a.slot_off[ppos], tlen, bw, pw);
	const uint32_t W = R.W;
	const int mode = a.mode & 3;
	const int gapo1 = a.gapo1, gape1 = a.gape1, gapo2 = a.gapo2, gape2 = a.gape2;
	// cigar scratch: the tail end of this pair's slot (tlen + 3 row records long)
	uint32_t *cig_end = R.cigar_end(tlen);
	uint32_t ncig = 0;
	auto cig_push = [&](uint32_t w){ ncig++; *(cig_end - ncig) = w; };
	auto cig_add = [&](uint32_t cg, uint32_t op, uint32_t sz) -> uint32_t {   // bsalign.h:409-417
		if(op == (cg & 0xf)) return cg + (sz << 4);
		if(cg) cig_push(cg);
		return (sz << 4) | op;
	};
	bool bad = false;
	// ---- end cell (bsalign.h:4023-4045)
	rs.score = BSA_SCORE_MIN;
	const int lastbeg = R.beg((int)tlen - 1);
	if(mode == BSA_MODE_GLOBAL){
		if(qlen - 1u - (uint32_t)lastbeg >= bw) bad = true;
		rs.score = R.getscore((int)tlen - 1, (long)qlen - 1 - lastbeg);
		rs.qe = (int)qlen - 1; rs.te = (int)tlen - 1;
	} else {
		for(uint32_t i = 0; i < tlen; i++){
			int rb = R.beg((int)i);
			if((uint32_t)rb + bw >= qlen){
				int sc = R.getscore((int)i, (long)qlen - 1 - rb);
				if(sc > rs.score){ rs.score = sc; rs.qe = (int)qlen - 1; rs.te = (int)i; }
			}
		}
		int ms; uint32_t rmax = row_max_dev(R, (int)tlen - 1, &ms);
		if(ms > rs.score){ rs.score = ms; rs.qe = lastbeg + (int)rmax; rs.te = (int)tlen - 1; }
	}
	if(!bad){ int b = R.beg(rs.te); if(rs.qe < b || rs.qe >= b + (int)bw) bad = true; }
	// ---- backcal (bsalign.h:3704-3852)
	if(!bad){
		int Hs0 = 0, Hs1, pend = 0, prior_match = 0;
		uint32_t cg = 0;
		rs.qb = rs.qe; rs.qe++;
		rs.tb = rs.te; rs.te++;
		Hs1 = R.mtx_getscore(rs.tb, rs.qb);
		for(;;){
			if((pend & 0xf) == 2 || (pend & 0xf) == 4){
				const int go = ((pend & 0xf) == 2) ? gapo1 : gapo2, ge = ((pend & 0xf) == 2) ? gape1 : gape2;
				Hs0 = R.mtx_getscore(rs.tb, rs.qb);
				long long t = go + (long long)(pend >> 4) * ge;
				if(Hs0 + t == Hs1){
					cg = cig_add(cg, 2, (uint32_t)(pend >> 4));
					rs.del += pend >> 4; rs.aln += pend >> 4;
					Hs1 = Hs0; pend = 0;
				} else {
					pend += 1 << 4; rs.tb--;
					if(rs.tb < -1){ bad = true; break; }
					continue;
				}
			}
			if(rs.qb < 0 || rs.tb < 0) break;
			const int pbeg = R.beg(rs.tb - 1);
			if(rs.qb == pbeg){
				if(rs.qb){ Hs0 = R.ubv(rs.tb - 1, 0u); prior_match = 0; }
				else if(mode == BSA_MODE_OVERLAP || rs.tb == 0) Hs0 = 0;
				else if(pw < 2) Hs0 = gapo1 + gape1 * rs.tb;
				else Hs0 = max(gapo1 + gape1 * rs.tb, gapo2 + gape2 * rs.tb);
			} else {
				Hs0 = R.mtx_getscore(rs.tb - 1, rs.qb - 1);
			}
			const int x = rs.qb - pbeg;
			int uu = 0, ee = 0, qq = 0;
			if(x >= 0 && x < (int)bw){
				const int8_t *pr = R.blkp(rs.tb - 1, (uint32_t)x / W);
				const uint32_t xk = (uint32_t)x % W;
				uu = pr[xk];
				ee = (pw >= 1) ? pr[W + xk] : gapo1 + gape1;
				qq = (pw == 2) ? pr[2 * W + xk] : 0;
			}
			const int s = a.matrix[qseq[rs.qb] * 4 + tseq[rs.tb]];
			const int h = Hs1 - Hs0;
			int bt;   // 0 M, 1 I, 2 D, 4 D2 (bsalign.h:3667-3702)
			if(x > (int)bw) bt = 1;
			else if(x == (int)bw) bt = (h == s) ? 0 : 1;
			else if(prior_match){
				if(h == s) bt = 0;
				else if(h == uu + ee) bt = 2;
				else if(pw == 2 && h == uu + qq) bt = 4;
				else bt = 1;
			} else {
				if(h == uu + ee) bt = 2;
				else if(pw == 2 && h == uu + qq) bt = 4;
				else if(h == s) bt = 0;
				else bt = 1;
			}
			prior_match = 1;
			if(bt == 0){
				if(qseq[rs.qb] == tseq[rs.tb]) rs.mat++; else rs.mis++;
				rs.qb--; rs.tb--; rs.aln++;
				cg = cig_add(cg, 0, 1);
				Hs1 = Hs0;
			} else if(bt == 1){
				if(rs.qb <= 0){
					cg = cig_add(cg, 1, 1);
					Hs1 = Hs0;
					rs.qb--; rs.ins++; rs.aln++;
				} else {
					const int cbeg = R.beg(rs.tb);
					bool found = false;
					for(int sz = 1; sz + cbeg <= rs.qb; sz++){
						long long t = (pw == 2) ? (long long)max(gapo1 + sz * gape1, gapo2 + sz * gape2) : (long long)(gapo1 + sz * gape1);
						Hs0 = R.mtx_getscore(rs.tb, rs.qb - sz);
						if(Hs0 + t == Hs1){
							cg = cig_add(cg, 1, (uint32_t)sz);
							Hs1 = Hs0;
							rs.qb -= sz; rs.ins += sz; rs.aln += sz;
							found = true;
							break;
						}
					}
					if(!found){ bad = true; break; }
				}
			} else {
				pend = (1 << 4) | bt;
				rs.tb--;
				continue;
			}
		}
		if(!bad){
			if(mode == BSA_MODE_OVERLAP){
				if(cg) cig_push(cg);
			} else {
				uint32_t op = 0, sz = 0;
				if(rs.qb >= 0){ op = 1; sz = (uint32_t)rs.qb + 1u; rs.ins += (int)sz; rs.qb = -1; }
				else if(rs.tb >= 0){ op = 2; sz = (uint32_t)rs.tb + 1u; rs.del += (int)sz; rs.tb = -1; }
				rs.aln += (int)sz;
				cg = cig_add(cg, op, sz);
				if(cg) cig_push(cg);
			}
			rs.qb++; rs.tb++;
		}
	}
	if(bad){
		atomicOr(&a.status[pair], BSA_ST_TRACE);
		ncig = 0;
	}
	out[pair] = rs;
	cig_cnt[ppos] = ncig;
}

// ---------------------------------------------------------------------------------------------
// Global-mode traceback, register-record form (bandwidth <= 128, i.e. W <= 8).
//
// Same decisions as k_align8_backcal.  The walk is a chain of dependent loads (band offset -> block record) and the
// loads of one wave return in order, so what hides the HBM latency is the NUMBER OF WAVES, not prefetching inside a
// wave (a helper wave that touched the rows ahead was measured: it only added traffic once the records were tiled).
// The kernel therefore runs one pair per lane but uses only TRACE_LANES lanes of every wave: a 50 k-pair launch
// becomes ~3000 waves instead of ~800 and a step waits for the slowest of 16 lanes instead of 64.  The cell step
// reads one whole block record (u, e, q bytes + ubegs of one running block) into registers instead of byte loads.
// ---------------------------------------------------------------------------------------------
#define TRACE_LANES 16u
struct BlkRec { uint32_t d[7]; };
template<int W>
static __device__ __forceinline__ int rec_prefix(const BlkRec &r, uint32_t cells, uint32_t x){   // ubegs[y] + u[0..x]
	const uint32_t ubw = cells / 4u;
	int s = (int)((ubw == 1) ? r.d[1] : (ubw == 2) ? r.d[2] : (ubw == 3) ? r.d[3] : (ubw == 4) ? r.d[4] : (ubw == 5) ? r.d[5] : r.d[6]);
	const uint32_t lo4 = r.d[0], hi4 = (W > 4) ? r.d[1] : 0u;
	const uint32_t nlo = x >= 3 ? 0xFFFFFFFFu : ((1u << (8 * (x + 1))) - 1u);
	const uint32_t nhi = x < 4 ? 0u : (x >= 7 ? 0xFFFFFFFFu : ((1u << (8 * (x - 3))) - 1u));
	const uint32_t lo = (lo4 & nlo) ^ 0x80808080u, hi = (hi4 & nhi) ^ 0x80808080u;
	s += (int)__builtin_amdgcn_sad_u8(lo, 0u, 0u) + (int)__builtin_amdgcn_sad_u8(hi, 0u, 0u) - 1024;
	return s;
}
static __device__ __forceinline__ int rec_byte(const BlkRec &r, uint32_t b){                        // signed byte b of the record
	const uint32_t w = b >> 2;
	const uint32_t v = (w == 0) ? r.d[0] : (w == 1) ? r.d[1] : (w == 2) ? r.d[2] : (w == 3) ? r.d[3] : (w == 4) ? r.d[4] : r.d[5];
	return __builtin_amdgcn_sbfe((int)v, 8u * (b & 3u), 8u);
}


// The walk is written as a flat state machine with exactly ONE block-record fetch per iteration, whatever the state
// of the lane (cell decision, block-boundary cell, insertion-length scan, deletion run): the 64 pairs of a wave are
// in different states at any time, and with nested loops every state's chain of dependent loads would be paid
// one after the other on every step.
template<int W, int PW>
__global__ void __launch_bounds__(64) k_align8_backcal_g(const Align8Args a, bsa_result_t *out, uint32_t *cig_cnt){
	constexpr int pw = PW;
	constexpr uint32_t CELLS = (((uint32_t)(PW + 1) * W + 3u) & ~3u), BLKW = (CELLS + 4u) / 4u;
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t g = blockIdx.x * TRACE_LANES + lane;
	const bool live = lane < TRACE_LANES && g < a.count;
	const uint32_t ppos = a.first + (live ? g : 0u);
	const uint32_t pair = a.order[ppos];
	const bool skip = !live || a.status[pair] != 0u;
	const uint32_t qlen = a.qlen[pair], tlen = a.tlen[pair];
	RowView R;
	R.init(a.rows + a.slot_off[ppos], tlen, a.bw, pw);
	bsa_result_t rs;
	rs.score = 0; rs.qb = rs.qe = rs.tb = rs.te = 0; rs.mat = rs.mis = rs.ins = rs.del = rs.aln = 0;
	if(skip){ if(live){ out[pair] = rs; cig_cnt[ppos] = 0; } return; }
	const uint8_t *qseq = a.qst + a.qpoff[pair];
	const uint8_t *tseq = a.tst + a.tpoff[pair];
	const uint32_t bw = a.bw;
	const int gapo1 = a.gapo1, gape1 = a.gape1, gapo2 = a.gapo2, gape2 = a.gape2;
	uint32_t *cig_end = R.cigar_end(tlen);
	uint32_t ncig = 0;
	auto cig_push = [&](uint32_t w){ ncig++; *(cig_end - ncig) = w; };
	auto cig_add = [&](uint32_t cg, uint32_t op, uint32_t sz) -> uint32_t {   // bsalign.h:409-417
		if(op == (cg & 0xf)) return cg + (sz << 4);
		if(cg) cig_push(cg);
		return (sz << 4) | op;
	};
	uint64_t qwin = 0, twin = 0; int qwb = -1000, twb = -1000;     // 8 bases of each sequence in a register window
	auto qbase_at = [&](int idx) -> int {
		if(idx < qwb || idx >= qwb + 8){ qwb = max(idx - 7, 0); __builtin_memcpy(&qwin, qseq + qwb, 8); }
		return (int)((qwin >> (8 * (idx - qwb))) & 0xffu);
	};
	auto tbase_at = [&](int idx) -> int {
		if(idx < twb || idx >= twb + 8){ twb = max(idx - 7, 0); __builtin_memcpy(&twin, tseq + twb, 8); }
		return (int)((twin >> (8 * (idx - twb))) & 0xffu);
	};
	bool bad = false;
	// ---- end cell (global, bsalign.h:4034-4037)
	const int lastbeg = R.beg((int)tlen - 1);
	if(qlen - 1u - (uint32_t)lastbeg >= bw) bad = true;
	rs.score = R.getscore((int)tlen - 1, (long)qlen - 1 - lastbeg);
	rs.qe = (int)qlen - 1; rs.te = (int)tlen - 1;
	// ---- backcal (bsalign.h:3704-3852) as a state machine
	enum { ST_CELL = 0, ST_CELLB = 1, ST_ISCAN = 2, ST_DRUN = 3, ST_DONE = 4 };
	int state = bad ? ST_DONE : ST_CELL;
	int Hs0 = 0, Hs1 = rs.score, pend = 0, prior_match = 0, isz = 0;
	uint32_t cg = 0;
	rs.qb = rs.qe; rs.qe++;
	rs.tb = rs.te; rs.te++;
	// band offsets of rows tb, tb-1, tb-2 live in registers
	int beg_c = lastbeg, beg_p = R.beg(rs.tb - 1), beg_pp = (rs.tb - 2 >= -1) ? R.beg(rs.tb - 2) : 0;
	auto row_down = [&](){ rs.tb--; beg_c = beg_p; beg_p = beg_pp; beg_pp = (rs.tb - 2 >= -1) ? R.beg(rs.tb - 2) : 0; };
	BlkRec r1;      // record of the cell's own block, kept across a ST_CELLB iteration
#pragma unroll
	for(int k = 0; k < 7; k++) r1.d[k] = 0u;
	int cx = 0;     // band position of the current cell in row tb-1
	while(state != ST_DONE){
		if(state == ST_CELL && (rs.qb < 0 || rs.tb < 0)) break;
		// ---- which record does this lane need?
		int frow; uint32_t fblk; uint32_t fx = 0;
		if(state == ST_CELL){
			cx = rs.qb - beg_p;
			const uint32_t xc = (cx < 0) ? 0u : ((uint32_t)cx >= bw ? bw - 1u : (uint32_t)cx);
			frow = rs.tb - 1; fblk = xc / W;
		} else if(state == ST_CELLB){
			frow = rs.tb - 1; fblk = (uint32_t)cx / W - 1u; fx = W - 1u;
		} else {
			uint32_t pp = (uint32_t)((state == ST_ISCAN ? rs.qb - isz : rs.qb) - beg_c);
			if(pp >= bw) pp = bw - 1u;
			frow = rs.tb; fblk = pp / W; fx = pp % W;
		}
		const uint32_t *rp = (const uint32_t*)R.blkp(frow, fblk);
		BlkRec rc;
#pragma unroll
		for(int k = 0; k < 7; k++) rc.d[k] = ((uint32_t)k < BLKW) ? rp[k] : 0u;
		// ---- act on it
		bool decide = false;
		if(state == ST_CELL){
			r1 = rc;
			if(rs.qb == beg_p){
				if(rs.qb){ Hs0 = (int)rc.d[BLKW - 1u]; prior_match = 0; }          // ubegs[0] of row tb-1 (fblk == 0 here)
				else if(rs.tb == 0) Hs0 = 0;
				else if(pw < 2) Hs0 = gapo1 + gape1 * rs.tb;
				else Hs0 = max(gapo1 + gape1 * rs.tb, gapo2 + gape2 * rs.tb);
				decide = true;
			} else {
				uint32_t pp = (uint32_t)(cx - 1);
				if(pp >= bw) pp = bw - 1u;
				const uint32_t y2 = pp / W, x2 = pp % W;
				if(y2 == fblk){ Hs0 = rec_prefix<W>(rc, CELLS, x2); decide = true; }
				else if(y2 + 1u == fblk) state = ST_CELLB;                           // H(x-1, y-1) lives in the previous block
				else { Hs0 = R.mtx_getscore(rs.tb - 1, rs.qb - 1); decide = true; }   // off-band corner, generic path
			}
		} else if(state == ST_CELLB){
			Hs0 = rec_prefix<W>(rc, CELLS, fx);
			decide = true;
		} else if(state == ST_ISCAN){
			const int hv = rec_prefix<W>(rc, CELLS, fx);
			const long long t = (pw == 2) ? (long long)max(gapo1 + isz * gape1, gapo2 + isz * gape2) : (long long)(gapo1 + isz * gape1);
			if(hv + t == Hs1){
				cg = cig_add(cg, 1, (uint32_t)isz);
				Hs1 = hv;
				rs.qb -= isz; rs.ins += isz; rs.aln += isz;
				state = ST_CELL;
			} else {
				isz++;
				if(isz + beg_c > rs.qb){ bad = true; state = ST_DONE; }
			}
		} else {   // ST_DRUN
			const int go = ((pend & 0xf) == 2) ? gapo1 : gapo2, ge = ((pend & 0xf) == 2) ? gape1 : gape2;
			const int hv = rec_prefix<W>(rc, CELLS, fx);
			const long long t = go + (long long)(pend >> 4) * ge;
			if(hv + t == Hs1){
				cg = cig_add(cg, 2, (uint32_t)(pend >> 4));
				rs.del += pend >> 4; rs.aln += pend >> 4;
				Hs1 = hv; pend = 0;
				state = ST_CELL;
			} else {
				pend += 1 << 4;
				row_down();
				if(rs.tb < -1){ bad = true; state = ST_DONE; }
			}
		}
		if(decide){
			int uu = 0, ee = 0, qq = 0;
			if(cx >= 0 && cx < (int)bw){
				const uint32_t xk = (uint32_t)cx % W;
				uu = rec_byte(r1, xk);
				ee = (pw >= 1) ? rec_byte(r1, W + xk) : gapo1 + gape1;
				qq = (pw == 2) ? rec_byte(r1, 2 * W + xk) : 0;
			}
			const int qbase = qbase_at(rs.qb), tbase = tbase_at(rs.tb);
			const uint32_t mr = (tbase == 0) ? a.mrow[0] : (tbase == 1) ? a.mrow[1] : (tbase == 2) ? a.mrow[2] : a.mrow[3];
			const int s = __builtin_amdgcn_sbfe((int)mr, 8u * (uint32_t)qbase, 8u);
			const int h = Hs1 - Hs0;
			int bt;   // 0 M, 1 I, 2 D, 4 D2 (bsalign.h:3667-3702)
			if(cx > (int)bw) bt = 1;
			else if(cx == (int)bw) bt = (h == s) ? 0 : 1;
			else if(prior_match){
				if(h == s) bt = 0;
				else if(h == uu + ee) bt = 2;
				else if(pw == 2 && h == uu + qq) bt = 4;
				else bt = 1;
			} else {
				if(h == uu + ee) bt = 2;
				else if(pw == 2 && h == uu + qq) bt = 4;
				else if(h == s) bt = 0;
				else bt = 1;
			}
			prior_match = 1;
			state = ST_CELL;
			if(bt == 0){
				if(qbase == tbase) rs.mat++; else rs.mis++;
				rs.qb--; rs.aln++;
				row_down();
				cg = cig_add(cg, 0, 1);
				Hs1 = Hs0;
			} else if(bt == 1){
				if(rs.qb <= 0){
					cg = cig_add(cg, 1, 1);
					Hs1 = Hs0;
					rs.qb--; rs.ins++; rs.aln++;
				} else {
					isz = 1;
					if(isz + beg_c > rs.qb){ bad = true; state = ST_DONE; }
					else state = ST_ISCAN;
				}
			} else {
				pend = (1 << 4) | bt;
				row_down();
				if(rs.tb < -1){ bad = true; state = ST_DONE; }
				else state = ST_DRUN;
			}
		}
	}
	if(!bad){
		uint32_t op = 0, sz = 0;      // global: leading clip becomes I / D (bsalign.h:3827-3842)
		if(rs.qb >= 0){ op = 1; sz = (uint32_t)rs.qb + 1u; rs.ins += (int)sz; rs.qb = -1; }
		else if(rs.tb >= 0){ op = 2; sz = (uint32_t)rs.tb + 1u; rs.del += (int)sz; rs.tb = -1; }
		rs.aln += (int)sz;
		cg = cig_add(cg, op, sz);
		if(cg) cig_push(cg);
		rs.qb++; rs.tb++;
	}
	if(bad){
		atomicOr(&a.status[pair], BSA_ST_TRACE);
		ncig = 0;
	}
	out[pair] = rs;
	cig_cnt[ppos] = ncig;
}

// ---------------------------------------------------------------------------------------------
template<int W, int PW>
static hipError_t launch_fwd(const Align8Args &a, hipStream_t st){
	constexpr int BW = W * 16;
	constexpr int GROUP_LDS = ((PW + 1) * BW + 17 * 4 + 15) & ~15;
	const uint32_t groups_per_block = 16;
	const uint32_t blocks = (a.count + groups_per_block - 1) / groups_per_block;
	if(blocks == 0) return hipSuccess;
	hipLaunchKernelGGL((k_align8_fwd<W, PW>), dim3(blocks), dim3(256), GROUP_LDS * 16, st, a);
	return hipGetLastError();
}

template<int W>
static hipError_t launch_fwd_pw(const Align8Args &a, int pw, hipStream_t st){
	if(pw == 0) return launch_fwd<W, 0>(a, st);
	if(pw == 1) return launch_fwd<W, 1>(a, st);
	return launch_fwd<W, 2>(a, st);
}

bool bsa_align8_supported_bw(uint32_t bw){
	switch(bw / 16){ case 1: case 2: case 4: case 8: case 16: case 32: return (bw % 16) == 0; default: return false; }
}

hipError_t bsa_launch_align8_fwd(const Align8Args &a, int pw, hipStream_t st){
	// packed two-pairs-per-row kernel where its preconditions hold (BSA_ALIGN8_I32=1 forces the int32 kernel)
	const bool force_i32 = [](){ const char *e = bsa_env("BSA_ALIGN8_I32"); return e && e[0] == '1'; }();
	if(!force_i32 && bsa_align8_pk_supported(a, pw)) return bsa_launch_align8_fwd_pk(a, pw, st);
	switch(a.bw / 16){
		case 1:  return launch_fwd_pw<1>(a, pw, st);
		case 2:  return launch_fwd_pw<2>(a, pw, st);
		case 4:  return launch_fwd_pw<4>(a, pw, st);
		case 8:  return launch_fwd_pw<8>(a, pw, st);
		case 16: return launch_fwd_pw<16>(a, pw, st);
		case 32: return launch_fwd_pw<32>(a, pw, st);
		default: return hipErrorInvalidValue;
	}
}

hipError_t bsa_launch_align8_backcal(const Align8Args &a, int pw, bsa_result_t *out, uint32_t *cig_cnt, hipStream_t st){
	// global mode, W <= 8: two-wave traceback (walker + prefetching helper); BSA_ALIGN8_TRACE1=1 forces the one-wave kernel
	const bool force_one = [](){ const char *e = bsa_env("BSA_ALIGN8_TRACE1"); return e && e[0] == '1'; }();
	if(!force_one && (a.mode & 3) == BSA_MODE_GLOBAL && a.bw / 16 <= 8 && a.count){
		const uint32_t nb = (a.count + TRACE_LANES - 1) / TRACE_LANES;
#define TRACE_CASE(WW) case WW: \
			if(pw == 0) hipLaunchKernelGGL((k_align8_backcal_g<WW, 0>), dim3(nb), dim3(64), 0, st, a, out, cig_cnt); \
			else if(pw == 1) hipLaunchKernelGGL((k_align8_backcal_g<WW, 1>), dim3(nb), dim3(64), 0, st, a, out, cig_cnt); \
			else hipLaunchKernelGGL((k_align8_backcal_g<WW, 2>), dim3(nb), dim3(64), 0, st, a, out, cig_cnt); \
			return hipGetLastError();
		switch(a.bw / 16){
			TRACE_CASE(1) TRACE_CASE(2) TRACE_CASE(4) TRACE_CASE(8)
			default: break;
		}
#undef TRACE_CASE
	}
	const uint32_t blocks = (a.count + 63) / 64;
	if(blocks == 0) return hipSuccess;
	hipLaunchKernelGGL(k_align8_backcal, dim3(blocks), dim3(64), 0, st, a, pw, out, cig_cnt);
	return hipGetLastError();
}
