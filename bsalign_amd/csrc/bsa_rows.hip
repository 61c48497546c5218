// bsa_rows.hip -- row-level kernels of the 8-bit path for the POA seq->graph DP.
//
// The POA sweep (/root/reference/bspoa.h:2515-2618) does not walk target rows: for every graph edge u -> v it
// re-aligns u's DP row to v's band offset and computes v's row (dpalign_row_update_bspoa, bspoa.h:2232-2261 =
// banded_striped_epi8_seqalign_piecex_row_movx bsalign.h:2244 + _piecex_row_cal bsalign.h:3181), and for every
// further in-edge of v it merges (dpalign_row_merge_bspoa, bspoa.h:2263-2272 = _piecex_row_merge bsalign.h:2474).
// This file is that pair of operations (plus row_init, bsalign.h:2094) as a batch kernel over independent tasks:
// all edges of one topological level, of many reads / many POA windows at once.  Row blocks keep the reference's
// memory layout (us | es | qs | ubegs[17], striped index (p % W) * 16 + p / W, block size = the reference's mmblk,
// bspoa.h:2217), so host graph code can read them back unchanged.
//
// Mapping as in bsa_align8.hip: one task per 16-lane DPP row, lane j = running block j, W cells per lane in VGPRs,
// lane-exact saturating arithmetic.  S(x, base) is evaluated from the query codes for the profile the task names
// (bspoa.h:2199-2213: matrix M or M+refbonus; homopolymer bonus 1 where q[x] != q[x+1]) instead of four stored profiles.
#include "bsa_common.h"
#include "bsa_dpp.h"
#include <algorithm>

struct RowsArgs {
	uint8_t *rows;                  // row blocks, blk bytes each
	const bsa_row_task_t *tasks;
	const uint8_t *queries;         // one base per byte, codes 0..3
	const uint64_t *qoff;
	const uint32_t *qlen;
	uint32_t ntasks, blk, bw;
	int32_t mode;
	int32_t M, X, refbonus;
	int32_t gapo1, gape1, gapo2, gape2;
};

// a DP row in registers: lane j holds the W cells of running block j and the absolute scores around it
template<int W>
struct RowRegs { int u[W], e[W], q2[W], ubA, ubB; };

template<int W, int PW>
static __device__ __forceinline__ void rows_load(RowRegs<W> &R, const int8_t *bp, const int j){
	constexpr int BW = W * 16;
	const int *ub = (const int*)(bp + (PW + 1) * BW);
#pragma unroll
	for(int k = 0; k < W; k++){
		R.u[k] = bp[k * 16 + j];
		R.e[k] = (PW >= 1) ? bp[BW + k * 16 + j] : 0;
		R.q2[k] = (PW == 2) ? bp[2 * BW + k * 16 + j] : 0;
	}
	R.ubA = ub[j]; R.ubB = ub[j + 1];
}

template<int W, int PW>
static __device__ __forceinline__ void rows_store(const RowRegs<W> &R, int8_t *bp, const int j){
	constexpr int BW = W * 16;
	int *ub = (int*)(bp + (PW + 1) * BW);
#pragma unroll
	for(int k = 0; k < W; k++){
		bp[k * 16 + j] = (int8_t)R.u[k];
		if(PW >= 1) bp[BW + k * 16 + j] = (int8_t)R.e[k];
		if(PW == 2) bp[2 * BW + k * 16 + j] = (int8_t)R.q2[k];
	}
	ub[j] = R.ubA;
	if(j == 15) ub[16] = R.ubB;
}

// row_init (bsalign.h:2094-2140) with max_nt = M + refbonus + 1, min_nt = X as the POA passes them (bspoa.h:2226)
template<int W, int PW>
static __device__ __forceinline__ void rows_init(const RowsArgs &a, RowRegs<W> &R, const int j){
	const int gapo1 = a.gapo1, gape1 = a.gape1, gapo2 = a.gapo2, gape2 = a.gape2;
	const int nt_max = a.M + a.refbonus + 1, nt_min = a.X;
	const int type = a.mode & 3;
	int bs = 0;
	const int first = trunc8(gapo1 + gape1 + nt_min - nt_max);
	const int xp = (PW == 2) ? (gapo2 - gapo1) / (gape1 - gape2) : 0;
#pragma unroll
	for(int k = 0; k < W; k++){
		int p = j * W + k, v;
		if(type == BSA_MODE_OVERLAP) v = 0;
		else if(p == 0) v = first;
		else if(PW == 2) v = (p < xp) ? gape1 : gape2;
		else v = gape1;
		R.u[k] = v; bs += v;
		R.e[k] = BSA_EPI8_MIN; R.q2[k] = BSA_EPI8_MIN;
	}
	const int inc = row_iscan16(bs);
	const int base0 = (type == BSA_MODE_OVERLAP) ? 0 : (nt_max - nt_min);
	R.ubB = base0 + inc; R.ubA = R.ubB - bs;
}

// row_merge (bsalign.h:2474-2616): R = cell-wise max(R, row at b1).  Lanes are independent; int16 offsets around a
// common base, re-based every 256 vectors as the reference does.
template<int W, int PW>
static __device__ __forceinline__ void rows_merge(RowRegs<W> &R, const int8_t *b1, const int j){
	constexpr int BW = W * 16;
	const int *ub1 = (const int*)(b1 + (PW + 1) * BW);
	int s0 = R.ubA, s1 = ub1[j];
	const int end0 = DPP_BCAST(R.ubB, 15), end1 = ub1[16];
	const int ubn = max(s0, s1);
	auto s16 = [](int v) -> int { return min(max(v, -32768), 32767); };
	int t0 = 0, t1 = 0, mprev = 0;
#pragma unroll
	for(int k = 0; k < W; k++){
		if((k & 255) == 0){
			if(k){ s0 += t0; s1 += t1; }
			int d = s0 - s1;
			d = min(max(d, -0x7FFF), 0x7FFF);
			const int x0 = d >> 1, x1 = x0 - d;
			s0 -= x0; s1 -= x1;
			t0 = s16(x0); t1 = s16(x1);
			mprev = max(t0, t1);
		}
		t0 = s16(t0 + R.u[k]);
		t1 = s16(t1 + b1[k * 16 + j]);
		const int m = max(t0, t1);
		R.u[k] = sat8(s16(m - mprev));
		mprev = m;
		if(PW >= 1){
			const int a0 = s16(t0 + R.e[k]), a1 = s16(t1 + b1[BW + k * 16 + j]);
			R.e[k] = sat8(s16(max(a0, a1) - m));
		}
		if(PW == 2){
			const int a0 = s16(t0 + R.q2[k]), a1 = s16(t1 + b1[2 * BW + k * 16 + j]);
			R.q2[k] = sat8(s16(max(a0, a1) - m));
		}
	}
	R.ubA = ubn;
	const int nxt = DPP_SHL(0, ubn, 1);
	R.ubB = (j == 15) ? max(end0, end1) : nxt;       // ubegs[j + 1]: the next lane's start, ubegs[16] for the last
}

// the W + 1 query codes lane j needs for a task with band offset qoff_dst (4 = beyond the read end)
template<int W>
static __device__ __forceinline__ void rows_fetch_codes(const uint8_t *qp, uint32_t qlen, uint32_t qoff_dst, int (&qc)[W + 1], const int j){
	const uint32_t x0 = qoff_dst + (uint32_t)j * W;
	constexpr int ND = (W + 4) / 4;                          // dwords that cover the W + 1 codes
	if(x0 + 4u * ND <= qlen){                                // the whole window lies inside the read: a few unaligned dword loads
		uint32_t w[ND];
#pragma unroll
		for(int i = 0; i < ND; i++) __builtin_memcpy(&w[i], qp + x0 + 4 * i, 4);
#pragma unroll
		for(int k = 0; k <= W; k++) qc[k] = (int)((w[k >> 2] >> (8 * (k & 3))) & 0xffu);
	} else {
#pragma unroll
		for(int k = 0; k <= W; k++) qc[k] = (x0 + k < qlen) ? (int)qp[x0 + k] : 4;
	}
}

// update: R = row_cal(row_movx(R, qoff_dst - qoff_src)) (bspoa.h:2232-2261); qc = rows_fetch_codes for this task
template<int W, int PW>
static __device__ __forceinline__ void rows_update(const RowsArgs &a, const bsa_row_task_t &tk, RowRegs<W> &R, const int (&qc)[W + 1], int8_t *gl, const int j){
	constexpr int BW = W * 16;
	int8_t *su = gl, *se = gl + BW, *sq = gl + 2 * BW;
	int *sub = (int*)(gl + (PW + 1) * BW);
	const int gapo1 = a.gapo1, gape1 = a.gape1, gapo2 = a.gapo2, gape2 = a.gape2;
	const int GapE = trunc8(gape1), GapOE = trunc8(gapo1 + gape1);
	const int GapP = trunc8(gape2), GapQP = trunc8(gapo2 + gape2);
	const int GapOQ = sat8(GapOE - GapQP);
	const int nt_max = a.M + a.refbonus + 1, nt_min = a.X;       // as the POA passes them (bspoa.h:2241, 2226)
	const int type = a.mode & 3;
	int (&u)[W] = R.u; int (&e)[W] = R.e; int (&q2)[W] = R.q2;
	int &ubA = R.ubA; int &ubB = R.ubB;
	// ---- update: row_movx(qoff_dst - qoff_src) then row_cal (bspoa.h:2232-2261)
	const uint32_t movx = tk.qoff_dst - tk.qoff_src;
	int rh;
	if(tk.qoff_src == tk.qoff_dst){
		if(tk.qoff_src) rh = BSA_SCORE_MIN;
		else if(type == BSA_MODE_OVERLAP || tk.toff == 0) rh = 0;
		else if(PW < 2) rh = gapo1 + gape1 * (int)tk.toff;
		else rh = max(gapo1 + gape1 * (int)tk.toff, gapo2 + gape2 * (int)tk.toff);
	} else rh = BSA_SCORE_MIN;                      // replaced below by the moved ubegs[0] when the bands overlap
	const int cfirst = (PW == 2) ? (min(nt_min, gapo2 + gape2) - 1 - nt_max + (gapo2 + gape2))
	                             : (min(nt_min, gapo1 + gape1) - 1 - nt_max + (gapo1 + gape1));
	const int dsw = (PW == 2) ? (gapo1 - gapo2) / (gape2 - gape1) : (BW + 1);
	auto newcell_int = [&](int k) -> int { return (k == 0) ? cfirst : ((PW == 2 && k >= dsw) ? gape2 : gape1); };
	auto newcell_cum = [&](int n) -> int { int n1 = min(n, dsw); return cfirst + (n1 - 1) * gape1 + ((PW == 2) ? max(0, n - dsw) * gape2 : 0); };
	if(movx && movx < (uint32_t)W){
		// the usual step of a few cells: single-cell shifts in registers, one DPP per plane (same as the pairwise kernel)
		int bacc = 0;
		for(uint32_t sft = 0; sft < movx; sft++){
			const int nci = newcell_int((int)sft);
			const int dropped = u[0];
			const int in_u = DPP_SHL(trunc8(nci), u[0], 1);       // lane 15 receives the synthetic cell
#pragma unroll
			for(int k = 0; k + 1 < W; k++) u[k] = u[k + 1];
			u[W - 1] = in_u;
			if(PW >= 1){
				const int in_e = DPP_SHL(0, e[0], 1);
#pragma unroll
				for(int k = 0; k + 1 < W; k++) e[k] = e[k + 1];
				e[W - 1] = in_e;
			}
			if(PW == 2){
				const int in_q = DPP_SHL(0, q2[0], 1);
#pragma unroll
				for(int k = 0; k + 1 < W; k++) q2[k] = q2[k + 1];
				q2[W - 1] = in_q;
			}
			ubA += dropped;
			bacc += nci;
		}
		ubB = DPP_SHL(ubB + bacc, ubA, 1);                        // ubegs[j+1] of the moved row; lane 15: old end + synthetic sum
		rh = DPP_BCAST(ubA, 0);                                    // "movx -> aligned" (bspoa.h:2252)
	} else if(movx){
		// generic movx through LDS (bsalign.h:2244-2392)
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
		if(movx >= (uint32_t)BW){
#pragma unroll
			for(int k = 0; k < W; k++){ u[k] = 0; e[k] = 0; q2[k] = 0; }
			ubA = ubB = BSA_SCORE_MIN;
		} else {
			const uint32_t cyc = movx / W, m = movx % W, p0 = BW - movx;
#pragma unroll
			for(int k = 0; k < W; k++){
				const uint32_t src = j * W + k + movx;
				if(src < (uint32_t)BW){
					u[k] = su[src];
					if(PW >= 1) e[k] = se[src];
					if(PW == 2) q2[k] = sq[src];
				} else {
					u[k] = trunc8(newcell_int((int)(src - BW)));
					e[k] = 0; q2[k] = 0;
				}
			}
			auto new_ub = [&](uint32_t idx) -> int {
				int v;
				if(idx + cyc < 16u){
					const uint32_t l = idx + cyc;
					v = sub[l];
					for(uint32_t k = 0; k < m; k++) v += su[l * W + k];
				} else v = sub[16];
				const int nbefore = (int)(idx * W) - (int)p0;
				if(nbefore > 0) v += newcell_cum(nbefore);
				return v;
			};
			ubA = new_ub((uint32_t)j);
			ubB = new_ub((uint32_t)j + 1u);
		}
		if(tk.qoff_src + (uint32_t)BW >= tk.qoff_dst) rh = DPP_BCAST(ubA, 0);     // "movx -> aligned" (bspoa.h:2252)
	}
	// ---- S(x, base) for this lane's cells under the task's profile (bspoa.h:2199-2213, 2588)
	const int mat = (tk.prof & 1) ? (a.M + a.refbonus) : a.M;
	const bool hpc = !(tk.prof & 2);
	int S[W];
	{
		int cprev = qc[0];
#pragma unroll
		for(int k = 0; k < W; k++){
			const int cnext = qc[k + 1];
			int s;
			if(cprev == 4) s = BSA_EPI8_MIN;
			else {
				s = (cprev == (int)tk.base) ? mat : a.X;
				if(hpc && cnext != 4 && cnext != cprev) s += 1;
				s = trunc8(s);
			}
			S[k] = s;
			cprev = cnext;
		}
	}
	// ---- row_cal (bsalign.h:2727-2793 / 2885-2960 / 3084-3179)
	int h0;
	{
		int hh = (rh - ubA) + S[0];
		int t0 = u[0] + ((PW == 0) ? gape1 : (PW == 1) ? e[0] : max(e[0], q2[0]));
		hh = (hh >= t0) ? min(hh, BSA_EPI8_MAX) : BSA_EPI8_MIN;
		h0 = trunc8(hh);
	}
	int f = BSA_EPI8_MIN, gq = BSA_EPI8_MIN;
	{
		int hc = (j == 0) ? h0 : S[0];
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
			if(k + 1 < W) hc = S[k + 1];
		}
	}
	f = fpen(f, ubA, ubB, W * gape1, j);
	if(PW == 2) gq = fpen(gq, ubA, ubB, W * gape2, j);
	int htail, ulast = 0;
	{
		int v = 0, z = (j == 0) ? h0 : S[0], h = 0;
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
			if(k + 1 < W) z = S[k + 1];
		}
		htail = (PW == 0) ? h : (PW == 1) ? sat8(h - GapOE) : sat8(h - GapQP);
	}
	{
		const int vlast = sat8(htail - ulast);
		const int nB = ubB + vlast;
		const int vsh = DPP_SHR(0, vlast, 1);
		u[0] = sat8(u[0] - vsh);
		int nA = DPP_SHR(0, nB, 1);
		if(j == 0){ nA = ubA + u[0]; u[0] = 0; }
		ubA = nA; ubB = nB;
	}
}

// one independent row task: load, operate, store
template<int W, int PW>
static __device__ __forceinline__ void rows_task(const RowsArgs &a, uint8_t *rows, const bsa_row_task_t &tk, int8_t *gl, const int j){
	auto blkp = [&](uint32_t idx) -> int8_t* { return (int8_t*)(rows + (size_t)idx * a.blk); };
	RowRegs<W> R;
	if(tk.op == BSA_ROW_OP_INIT) rows_init<W, PW>(a, R, j);
	else {
		rows_load<W, PW>(R, blkp(tk.src), j);
		if(tk.op == BSA_ROW_OP_MERGE) rows_merge<W, PW>(R, blkp(tk.dst), j);
		else {
			int qc[W + 1];
			rows_fetch_codes<W>(a.queries + a.qoff[tk.query], a.qlen[tk.query], tk.qoff_dst, qc, j);
			rows_update<W, PW>(a, tk, R, qc, gl, j);
		}
	}
	rows_store<W, PW>(R, blkp(tk.dst), j);
}

// ---- the same row task for ANY bandwidth (run-time W): cells stay in the row blocks in global memory, lane j walks
// its W cells (striped index k * 16 + j is the lane's own byte column, so no cross-lane traffic except ubegs).  A moved
// row is staged in the program's block 0 -- the reference uses block 0 for exactly that (bspoa.h:2236-2241).  Slow path:
// the first read of every POA (band = whole read, bspoa.h:2100-2101) and reads shorter than the band width land here.
template<int PW>
static __device__ __forceinline__ void rows_task_gen(const RowsArgs &a, uint8_t *rows, const bsa_row_task_t &tk, const int j, int *sub){
	const uint32_t BW = a.bw, W = BW / 16u;
	const int gapo1 = a.gapo1, gape1 = a.gape1, gapo2 = a.gapo2, gape2 = a.gape2;
	const int GapE = trunc8(gape1), GapOE = trunc8(gapo1 + gape1);
	const int GapP = trunc8(gape2), GapQP = trunc8(gapo2 + gape2);
	const int GapOQ = sat8(GapOE - GapQP);
	const int nt_max = a.M + a.refbonus + 1, nt_min = a.X;
	const int type = a.mode & 3;
	auto blkp = [&](uint32_t idx) -> int8_t* { return (int8_t*)(rows + (size_t)idx * a.blk); };
	auto ubp = [&](int8_t *bp) -> int* { return (int*)(bp + (size_t)(PW + 1) * BW); };
	if(tk.op == BSA_ROW_OP_INIT){
		int8_t *bp = blkp(tk.dst);
		int bs = 0;
		const int first = trunc8(gapo1 + gape1 + nt_min - nt_max);
		const int xp = (PW == 2) ? (gapo2 - gapo1) / (gape1 - gape2) : 0;
		for(uint32_t k = 0; k < W; k++){
			int p = (int)(j * W + k), v;
			if(type == BSA_MODE_OVERLAP) v = 0;
			else if(p == 0) v = first;
			else if(PW == 2) v = (p < xp) ? gape1 : gape2;
			else v = gape1;
			bp[k * 16 + j] = (int8_t)v; bs += v;
			if(PW >= 1) bp[BW + k * 16 + j] = BSA_EPI8_MIN;
			if(PW == 2) bp[2 * BW + k * 16 + j] = BSA_EPI8_MIN;
		}
		const int inc = row_iscan16(bs);
		const int base0 = (type == BSA_MODE_OVERLAP) ? 0 : (nt_max - nt_min);
		int *ub = ubp(bp);
		ub[j] = base0 + inc - bs;
		if(j == 15) ub[16] = base0 + inc;
		return;
	}
	if(tk.op == BSA_ROW_OP_MERGE){
		const int8_t *b0 = blkp(tk.src); int8_t *b1 = blkp(tk.dst);
		const int *ub0 = (const int*)(b0 + (size_t)(PW + 1) * BW); int *ub1 = ubp(b1);
		int s0 = ub0[j], s1 = ub1[j];
		const int end0 = ub0[16], end1 = ub1[16];
		auto s16 = [](int v) -> int { return min(max(v, -32768), 32767); };
		int t0 = 0, t1 = 0, mprev = 0;
		const int ubn = max(s0, s1);
		for(uint32_t k = 0; k < W; k++){
			if((k & 255u) == 0u){
				if(k){ s0 += t0; s1 += t1; }
				int d = s0 - s1;
				d = min(max(d, -0x7FFF), 0x7FFF);
				const int x0 = d >> 1, x1 = x0 - d;
				s0 -= x0; s1 -= x1;
				t0 = s16(x0); t1 = s16(x1);
				mprev = max(t0, t1);
			}
			t0 = s16(t0 + b0[k * 16 + j]);
			t1 = s16(t1 + b1[k * 16 + j]);
			const int m = max(t0, t1);
			if(PW >= 1){
				const int a0 = s16(t0 + b0[BW + k * 16 + j]), a1 = s16(t1 + b1[BW + k * 16 + j]);
				b1[BW + k * 16 + j] = (int8_t)sat8(s16(max(a0, a1) - m));
			}
			if(PW == 2){
				const int a0 = s16(t0 + b0[2 * BW + k * 16 + j]), a1 = s16(t1 + b1[2 * BW + k * 16 + j]);
				b1[2 * BW + k * 16 + j] = (int8_t)sat8(s16(max(a0, a1) - m));
			}
			b1[k * 16 + j] = (int8_t)sat8(s16(m - mprev));
			mprev = m;
		}
		ub1[j] = ubn;
		if(j == 15) ub1[16] = max(end0, end1);
		return;
	}
	// ---- update
	const int8_t *sp = blkp(tk.src);
	const int *sub_g = (const int*)(sp + (size_t)(PW + 1) * BW);
	int ubA = sub_g[j], ubB = sub_g[j + 1];
	const uint32_t movx = tk.qoff_dst - tk.qoff_src;
	int rh;
	if(movx == 0){
		if(tk.qoff_src) rh = BSA_SCORE_MIN;
		else if(type == BSA_MODE_OVERLAP || tk.toff == 0) rh = 0;
		else if(PW < 2) rh = gapo1 + gape1 * (int)tk.toff;
		else rh = max(gapo1 + gape1 * (int)tk.toff, gapo2 + gape2 * (int)tk.toff);
	} else rh = BSA_SCORE_MIN;
	if(movx){
		int8_t *mp = blkp(0);                       // scratch block of the program
		const int cfirst = (PW == 2) ? (min(nt_min, gapo2 + gape2) - 1 - nt_max + (gapo2 + gape2))
		                             : (min(nt_min, gapo1 + gape1) - 1 - nt_max + (gapo1 + gape1));
		const int dsw = (PW == 2) ? (gapo1 - gapo2) / (gape2 - gape1) : (int)(BW + 1);
		auto newcell_int = [&](int k) -> int { return (k == 0) ? cfirst : ((PW == 2 && k >= dsw) ? gape2 : gape1); };
		auto newcell_cum = [&](int n) -> int { int n1 = min(n, dsw); return cfirst + (n1 - 1) * gape1 + ((PW == 2) ? max(0, n - dsw) * gape2 : 0); };
		if(movx >= BW){
			for(uint32_t k = 0; k < W; k++){
				mp[k * 16 + j] = 0;
				if(PW >= 1) mp[BW + k * 16 + j] = 0;
				if(PW == 2) mp[2 * BW + k * 16 + j] = 0;
			}
			ubA = ubB = BSA_SCORE_MIN;
		} else {
			const uint32_t cyc = movx / W, m = movx % W, p0 = BW - movx;
			for(uint32_t k = 0; k < W; k++){
				const uint32_t src = (uint32_t)j * W + k + movx;
				if(src < BW){
					const uint32_t si = (src % W) * 16u + src / W;
					mp[k * 16 + j] = sp[si];
					if(PW >= 1) mp[BW + k * 16 + j] = sp[BW + si];
					if(PW == 2) mp[2 * BW + k * 16 + j] = sp[2 * BW + si];
				} else {
					mp[k * 16 + j] = (int8_t)trunc8(newcell_int((int)(src - BW)));
					if(PW >= 1) mp[BW + k * 16 + j] = 0;
					if(PW == 2) mp[2 * BW + k * 16 + j] = 0;
				}
			}
			auto new_ub = [&](uint32_t idx) -> int {
				int v;
				if(idx + cyc < 16u){
					const uint32_t l = idx + cyc;
					v = sub_g[l];
					for(uint32_t k = 0; k < m; k++) v += sp[k * 16 + l];
				} else v = sub_g[16];
				const int nbefore = (int)(idx * W) - (int)p0;
				if(nbefore > 0) v += newcell_cum(nbefore);
				return v;
			};
			ubA = new_ub((uint32_t)j);
			ubB = new_ub((uint32_t)j + 1u);
		}
		if(tk.qoff_src + BW >= tk.qoff_dst) rh = DPP_BCAST(ubA, 0);
		sp = mp;
	}
	int8_t *dp = blkp(tk.dst);
	const uint32_t qlen = a.qlen[tk.query];
	const uint8_t *qp = a.queries + a.qoff[tk.query];
	const int mat = (tk.prof & 1) ? (a.M + a.refbonus) : a.M;
	const bool hpc = !(tk.prof & 2);
	const uint32_t x0 = tk.qoff_dst + (uint32_t)j * W;
	auto score = [&](uint32_t k) -> int {
		const uint32_t x = x0 + k;
		if(x >= qlen) return BSA_EPI8_MIN;
		const int c = qp[x];
		int s = (c == (int)tk.base) ? mat : a.X;
		if(hpc && x + 1 < qlen && (int)qp[x + 1] != c) s += 1;
		return trunc8(s);
	};
	int h0;
	{
		int hh = (rh - ubA) + score(0);
		const int u0 = sp[j], e0 = (PW >= 1) ? sp[BW + j] : 0, q0 = (PW == 2) ? sp[2 * BW + j] : 0;
		const int t0 = u0 + ((PW == 0) ? gape1 : (PW == 1) ? e0 : max(e0, q0));
		hh = (hh >= t0) ? min(hh, BSA_EPI8_MAX) : BSA_EPI8_MIN;
		h0 = trunc8(hh);
	}
	int f = BSA_EPI8_MIN, gq = BSA_EPI8_MIN;
	{
		int hc = (j == 0) ? h0 : score(0);
		for(uint32_t k = 0; k < W; k++){
			const int uk = sp[k * 16 + j];
			int h;
			if(PW == 0){
				const int ee = sat8(uk + GapE);
				h = max(max(ee, hc), f);
				f = sat8(sat8(h + GapE) - uk);
			} else if(PW == 1){
				const int ee = sat8(sp[BW + k * 16 + j] + uk);
				h = max(max(ee, hc), f);
				f = sat8(f + GapE);
				h = sat8(h + GapOE);
				f = sat8(max(f, h) - uk);
			} else {
				const int ee = sat8(sp[BW + k * 16 + j] + uk), qq = sat8(sp[2 * BW + k * 16 + j] + uk);
				h = max(max(ee, hc), max(qq, max(f, gq)));
				f = sat8(f + GapE);
				h = sat8(h + GapOE);
				f = sat8(max(f, h) - uk);
				gq = sat8(gq + GapP);
				h = sat8(h - GapOQ);
				gq = sat8(max(gq, h) - uk);
			}
			if(k + 1 < W) hc = score(k + 1);
		}
	}
	f = fpen(f, ubA, ubB, (int)W * gape1, j);
	if(PW == 2) gq = fpen(gq, ubA, ubB, (int)W * gape2, j);
	int htail, ulast = 0, unew0 = 0;
	{
		int v = 0, z = (j == 0) ? h0 : score(0), h = 0;
		for(uint32_t k = 0; k < W; k++){
			const int uk = sp[k * 16 + j];
			int un;
			if(PW == 0){
				const int ee = sat8(uk + GapE);
				h = max(max(ee, z), f);
				un = sat8(h - v);
				v = sat8(h - uk);
				f = sat8(sat8(h + GapE) - uk);
			} else if(PW == 1){
				int ee = sat8(sp[BW + k * 16 + j] + uk);
				h = max(max(ee, z), f);
				un = sat8(h - v);
				v = sat8(h - uk);
				ee = sat8(ee + GapE); ee = sat8(ee - h);
				dp[BW + k * 16 + j] = (int8_t)max(ee, GapOE);
				f = sat8(f + GapE);
				h = sat8(h + GapOE);
				f = sat8(max(f, h) - uk);
			} else {
				int ee = sat8(sp[BW + k * 16 + j] + uk), qq = sat8(sp[2 * BW + k * 16 + j] + uk);
				h = max(max(ee, z), max(qq, max(f, gq)));
				un = sat8(h - v);
				v = sat8(h - uk);
				ee = sat8(ee + GapE); ee = sat8(ee - h);
				dp[BW + k * 16 + j] = (int8_t)max(ee, GapOE);
				qq = sat8(qq + GapP); qq = sat8(qq - h);
				dp[2 * BW + k * 16 + j] = (int8_t)max(qq, GapQP);
				f = sat8(f + GapE);
				h = sat8(h + GapOE);
				f = sat8(max(f, h) - uk);
				gq = sat8(gq + GapP);
				h = sat8(h - GapOQ);
				gq = sat8(max(gq, h) - uk);
			}
			if(k == 0) unew0 = un; else dp[k * 16 + j] = (int8_t)un;
			ulast = uk;
			if(k + 1 < W) z = score(k + 1);
		}
		htail = (PW == 0) ? h : (PW == 1) ? sat8(h - GapOE) : sat8(h - GapQP);
	}
	{
		const int vlast = sat8(htail - ulast);
		const int nB = ubB + vlast;
		const int vsh = DPP_SHR(0, vlast, 1);
		int u0 = sat8(unew0 - vsh);
		int nA = DPP_SHR(0, nB, 1);
		if(j == 0){ nA = ubA + u0; u0 = 0; }
		dp[j] = (int8_t)u0;
		int *ub = ubp(dp);
		ub[j] = nA;
		if(j == 15) ub[16] = nB;
	}
	(void)sub;
}

#define ROWS_GROUP_LDS(W, PW) ((((PW) + 1) * (W) * 16 + 17 * 4 + 15) & ~15)

// W == 0 selects the run-time-W variant
template<int W, int PW>
__global__ void __launch_bounds__(256) k_rows(const RowsArgs a){
	__shared__ __attribute__((aligned(16))) int8_t smem[16 * ROWS_GROUP_LDS(W, PW)];
	const int lt = threadIdx.x, j = lt & 15;
	const uint32_t g = (blockIdx.x * 256u + lt) >> 4;
	if(g >= a.ntasks) return;          // tasks are whole 16-lane rows: no partial DPP rows
	const bsa_row_task_t tk = a.tasks[g];
	int8_t *gl = smem + (lt >> 4) * ROWS_GROUP_LDS(W, PW);
	if constexpr(W != 0) rows_task<W, PW>(a, a.rows, tk, gl, j);
	else rows_task_gen<PW>(a, a.rows, tk, j, (int*)gl);
}

// ---- the whole sweep of one read over its sub-graph (align_rd_bspoacore, bspoa.h:2515-2618) as a device program.
// The order in which the reference visits edges depends only on the graph (stack + per-node in-degree counters), never
// on DP values, so the host flattens it into a task list; one 16-lane DPP row executes one program from start to end
// and keeps the running best end cell (g->maxscr / maxidx / maxoff) exactly as the reference updates it: strictly
// greater wins, in program order.  Thousands of programs (POA windows) run side by side.
struct SweepArgs {
	RowsArgs r;
	const bsa_sweep_prog_t *progs;
	bsa_sweep_result_t *results;
	uint32_t nprogs;
	int32_t T;
};

template<int WT, int PW>
__global__ void __launch_bounds__(256) k_sweep(const SweepArgs a){
	const int W = WT ? WT : (int)(a.r.bw / 16u);       // WT == 0: run-time W
	const int BW = W * 16;
	__shared__ __attribute__((aligned(16))) int8_t smem[16 * ROWS_GROUP_LDS(WT, PW)];
	const int lt = threadIdx.x, j = lt & 15;
	const uint32_t g = (blockIdx.x * 256u + lt) >> 4;
	if(g >= a.nprogs) return;
	const bsa_sweep_prog_t pg = a.progs[g];
	int8_t *gl = smem + (lt >> 4) * ROWS_GROUP_LDS(WT, PW);
	int *sub = (int*)(gl + (PW + 1) * WT * 16);
	uint8_t *rows = a.r.rows + (size_t)pg.first_block * a.r.blk;
	const int type = a.r.mode & 3;
	int maxscr = BSA_SCORE_MIN, maxidx = -1, maxoff = -1;
	auto getscore = [&](const int8_t *bp, uint32_t pos) -> int {      // bsalign.h:3187-3197; every lane computes the same value
		const uint32_t y = pos / (uint32_t)W, x = pos % (uint32_t)W;
		int s = ((const int*)(bp + (PW + 1) * BW))[y];
		for(uint32_t k = 0; k <= x; k++) s += bp[k * 16 + y];
		return s;
	};
	// The row a task produces is usually the next task's source (chains of the graph): it stays in registers and the
	// reload is skipped.  Stores are fire-and-forget; a fence is paid only before a task that reads row blocks from
	// memory after something was stored (other lanes' ubegs words are involved).  The next task record is fetched
	// while the current one computes.
	RowRegs<(WT ? WT : 1)> R;
	uint32_t held = 0xFFFFFFFFu;                 // block index whose row R holds
	bool dirty = false;
	auto sync_rows = [&](){
		if(dirty){
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			dirty = false;
		}
	};
	auto blkp = [&](uint32_t idx) -> int8_t* { return (int8_t*)(rows + (size_t)idx * a.r.blk); };
	// software pipeline: tk = current task, tk1 = next (already in registers), tk2 = the one after (load in flight);
	// the query codes of tk1 are requested while tk computes
	if(pg.ntasks == 0u){                         // nothing to do (whole DPP rows leave together: one program per row)
		if(j == 0){ bsa_sweep_result_t rs; rs.maxscr = BSA_SCORE_MIN; rs.maxidx = -1; rs.maxoff = -1; rs.reserved = 0; a.results[g] = rs; }
		return;
	}
	const uint32_t last = pg.ntasks - 1u;
	bsa_row_task_t tk = a.r.tasks[pg.first_task];
	bsa_row_task_t tk1 = a.r.tasks[pg.first_task + min(1u, last)];
	uint32_t cq = 0xFFFFFFFFu, cq_len = 0; const uint8_t *cq_ptr = nullptr;      // the program's current query
	auto use_query = [&](uint32_t qi){
		if(qi != cq){ cq = qi; cq_len = a.r.qlen[qi]; cq_ptr = a.r.queries + a.r.qoff[qi]; }
	};
	int qc[(WT ? WT : 1) + 1], qc1[(WT ? WT : 1) + 1];
	if constexpr(WT != 0){
		use_query(tk.query);
		rows_fetch_codes<WT>(cq_ptr, cq_len, tk.qoff_dst, qc, j);
	}
	for(uint32_t t = 0; t < pg.ntasks; t++){
		const bsa_row_task_t tk2 = a.r.tasks[pg.first_task + min(t + 2u, last)];
		if constexpr(WT != 0){
			if(tk1.op == BSA_ROW_OP_UPDATE){ use_query(tk1.query); rows_fetch_codes<WT>(cq_ptr, cq_len, tk1.qoff_dst, qc1, j); }
		}
		if(tk.op == BSA_ROW_OP_SCORE_TAIL || tk.op == BSA_ROW_OP_SCORE_END){
			sync_rows();
			const int8_t *bp = (const int8_t*)(rows + (size_t)tk.src * a.r.blk);
			const int slen = (int)a.r.qlen[tk.query], rpos = (int)tk.qoff_src;
			if(tk.op == BSA_ROW_OP_SCORE_END){                       // bspoa.h:2597-2606
				const int smax = getscore(bp, (uint32_t)(slen - 1 - rpos)) + a.T;
				if(smax > maxscr){ maxscr = smax; maxidx = (int)tk.toff; maxoff = slen - 1; }
			} else {                                                 // edge into the tail node, bspoa.h:2547-2577
				const int mo = min(slen, rpos + BW) - 1;
				int smax = getscore(bp, (uint32_t)(mo - rpos));
				if(slen > mo + 1){
					const int n = slen - mo - 1;
					smax += (PW < 2) ? (a.r.gapo1 + a.r.gape1 * n) : max(a.r.gapo1 + a.r.gape1 * n, a.r.gapo2 + a.r.gape2 * n);
				}
				smax += a.T;
				if(smax > maxscr){ maxscr = smax; maxidx = (int)tk.toff; maxoff = mo; }
				if(type == BSA_MODE_OVERLAP){
					// row_max (bsalign.h:3213-3329): per lane the best 32-vector chunk, lanes reduced in the reference's register order
					const int *ub = (const int*)(bp + (PW + 1) * BW);
					int base = ub[j], lmax = BSA_SCORE_MIN, lchunk = 0;
					for(int i = 0, c = 0; i < W; i += 32, c++){
						const int n = (i + 32 < W) ? 32 : W - i;
						int run = 0, cmax = -32767;
						for(int x = 0; x < n; x++){
							run += bp[(i + x) * 16 + j];
							run = min(max(run, -32768), 32767);
							cmax = max(cmax, run);
						}
						if(base + cmax > lmax){ lmax = base + cmax; lchunk = c; }
						base += run;
					}
					sub[j] = lmax;
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
					__builtin_amdgcn_wave_barrier();
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
					int best, lane;
					{
						int mm[4], ii[4];
						for(int k = 0; k < 4; k++){
							int m01, i01, m23, i23;
							if(sub[4 + k] > sub[k]){ m01 = sub[4 + k]; i01 = 4 + k; } else { m01 = sub[k]; i01 = k; }
							if(sub[12 + k] > sub[8 + k]){ m23 = sub[12 + k]; i23 = 12 + k; } else { m23 = sub[8 + k]; i23 = 8 + k; }
							if(m23 > m01){ mm[k] = m23; ii[k] = i23; } else { mm[k] = m01; ii[k] = i01; }
						}
						best = mm[0]; lane = ii[0];
						for(int k = 1; k < 4; k++) if(mm[k] > best){ best = mm[k]; lane = ii[k]; }
					}
					const int wchunk = __builtin_amdgcn_ds_bpermute(((lt & ~15) + lane) << 2, lchunk);
					int x = wchunk * 32, jj = x, umax = BSA_SCORE_MIN, uscr = 0;
					const int y = min(x + 32, W);
					for(; x < y; x++){
						uscr += bp[x * 16 + lane];
						if(uscr > umax){ jj = x; umax = uscr; }
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
					__builtin_amdgcn_wave_barrier();
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
					if(best > maxscr){ maxscr = best; maxidx = (int)tk.toff; maxoff = lane * W + jj + rpos; }
				}
			}
		} else if constexpr(WT != 0){
			if(tk.op == BSA_ROW_OP_INIT) rows_init<WT, PW>(a.r, R, j);
			else {
				if(tk.src != held){ sync_rows(); rows_load<WT, PW>(R, blkp(tk.src), j); }
				if(tk.op == BSA_ROW_OP_MERGE){ sync_rows(); rows_merge<WT, PW>(R, blkp(tk.dst), j); }
				else rows_update<WT, PW>(a.r, tk, R, qc, gl, j);
			}
			rows_store<WT, PW>(R, blkp(tk.dst), j);
			held = tk.dst; dirty = true;
		} else {
			rows_task_gen<PW>(a.r, rows, tk, j, sub);
			// the next task may read, from other lanes, what this one stored (ubegs[j + 1]); same wave, so program order + fence
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		}
		tk = tk1; tk1 = tk2;
#pragma unroll
		for(int k = 0; k <= (WT ? WT : 1); k++) qc[k] = qc1[k];
	}
	if(j == 0){
		bsa_sweep_result_t rs;
		rs.maxscr = maxscr; rs.maxidx = maxidx; rs.maxoff = maxoff; rs.reserved = 0;
		a.results[g] = rs;
	}
}

template<int W>
static hipError_t launch_rows_pw(const RowsArgs &a, int pw, hipStream_t st){
	const uint32_t blocks = (a.ntasks + 15) / 16;
	if(blocks == 0) return hipSuccess;
	if(pw == 0) hipLaunchKernelGGL((k_rows<W, 0>), dim3(blocks), dim3(256), 0, st, a);
	else if(pw == 1) hipLaunchKernelGGL((k_rows<W, 1>), dim3(blocks), dim3(256), 0, st, a);
	else hipLaunchKernelGGL((k_rows<W, 2>), dim3(blocks), dim3(256), 0, st, a);
	return hipGetLastError();
}

extern "C" int bsa_ctx_get_stream_internal(bsa_ctx_t *ctx, hipStream_t *st);
extern "C" int bsa_ctx_time_begin_internal(bsa_ctx_t *ctx, double cells, void **stop_event);
extern "C" int bsa_ctx_time_end_internal(bsa_ctx_t *ctx, void *stop_event);

extern "C" size_t bsa_rows_block_bytes(uint32_t bandwidth, int8_t gapo1, int8_t gape1, int8_t gapo2, int8_t gape2){   // == mmblk, bspoa.h:2217
	const uint32_t bw = (bandwidth + 15u) / 16u * 16u;
	const int pw = bsa_get_piecewise(gapo1, gape1, gapo2, gape2, (int)bw);
	return ((size_t)bw * (pw + 1) + 17 * 4 + 15) & ~(size_t)15;
}

extern "C" int bsa_rows_run(bsa_ctx_t *ctx, uint8_t *d_rows, const bsa_row_task_t *d_tasks, size_t ntasks,
		const uint8_t *d_queries, const uint64_t *d_qoff, const uint32_t *d_qlen, const bsa_rows_params_t *par){
	if(!ctx || !par || (ntasks && (!d_rows || !d_tasks || !d_queries || !d_qoff || !d_qlen))) return BSA_E_ARG;
	if(ntasks > 0xFFFFFFF0ull) return BSA_E_ARG;
	hipStream_t st;
	int rc = bsa_ctx_get_stream_internal(ctx, &st);
	if(rc != BSA_OK) return rc;
	const uint32_t bw = (par->bandwidth + 15u) / 16u * 16u;
	if(bw == 0) return BSA_E_UNSUPPORTED;
	RowsArgs a;
	a.rows = d_rows; a.tasks = d_tasks; a.queries = d_queries; a.qoff = d_qoff; a.qlen = d_qlen;
	a.ntasks = (uint32_t)ntasks; a.bw = bw; a.mode = par->mode;
	a.M = par->M; a.X = par->X; a.refbonus = par->refbonus;
	a.gapo1 = par->gapo1; a.gape1 = par->gape1; a.gapo2 = par->gapo2; a.gape2 = par->gape2;
	const int pw = bsa_get_piecewise(par->gapo1, par->gape1, par->gapo2, par->gape2, (int)bw);
	a.blk = (uint32_t)bsa_rows_block_bytes(bw, par->gapo1, par->gape1, par->gapo2, par->gape2);
	hipError_t e;
	switch(bw / 16){
		case 1:  e = launch_rows_pw<1>(a, pw, st); break;
		case 2:  e = launch_rows_pw<2>(a, pw, st); break;
		case 4:  e = launch_rows_pw<4>(a, pw, st); break;
		case 8:  e = launch_rows_pw<8>(a, pw, st); break;
		case 16: e = launch_rows_pw<16>(a, pw, st); break;
		default: e = launch_rows_pw<0>(a, pw, st); break;
	}
	return e == hipSuccess ? BSA_OK : BSA_E_HIP;
}

template<int W>
static hipError_t launch_sweep_pw(const SweepArgs &a, int pw, hipStream_t st){
	const uint32_t blocks = (a.nprogs + 15) / 16;
	if(blocks == 0) return hipSuccess;
	if(pw == 0) hipLaunchKernelGGL((k_sweep<W, 0>), dim3(blocks), dim3(256), 0, st, a);
	else if(pw == 1) hipLaunchKernelGGL((k_sweep<W, 1>), dim3(blocks), dim3(256), 0, st, a);
	else hipLaunchKernelGGL((k_sweep<W, 2>), dim3(blocks), dim3(256), 0, st, a);
	return hipGetLastError();
}

extern "C" int bsa_sweep_run(bsa_ctx_t *ctx, uint8_t *d_rows, const bsa_row_task_t *d_tasks, const bsa_sweep_prog_t *d_progs,
		size_t nprogs, const uint8_t *d_queries, const uint64_t *d_qoff, const uint32_t *d_qlen,
		const bsa_sweep_params_t *par, bsa_sweep_result_t *d_results){
	if(!ctx || !par || (nprogs && (!d_rows || !d_tasks || !d_progs || !d_queries || !d_qoff || !d_qlen || !d_results))) return BSA_E_ARG;
	if(nprogs > 0x0FFFFFF0ull) return BSA_E_ARG;
	hipStream_t st;
	int rc = bsa_ctx_get_stream_internal(ctx, &st);
	if(rc != BSA_OK) return rc;
	const bsa_rows_params_t *rp = &par->rows;
	const uint32_t bw = (rp->bandwidth + 15u) / 16u * 16u;
	if(bw == 0) return BSA_E_UNSUPPORTED;
	SweepArgs a;
	a.r.rows = d_rows; a.r.tasks = d_tasks; a.r.queries = d_queries; a.r.qoff = d_qoff; a.r.qlen = d_qlen;
	a.r.ntasks = 0; a.r.bw = bw; a.r.mode = rp->mode;
	a.r.M = rp->M; a.r.X = rp->X; a.r.refbonus = rp->refbonus;
	a.r.gapo1 = rp->gapo1; a.r.gape1 = rp->gape1; a.r.gapo2 = rp->gapo2; a.r.gape2 = rp->gape2;
	a.r.blk = (uint32_t)bsa_rows_block_bytes(bw, rp->gapo1, rp->gape1, rp->gapo2, rp->gape2);
	a.progs = d_progs; a.results = d_results; a.nprogs = (uint32_t)nprogs; a.T = par->T;
	const int pw = bsa_get_piecewise(rp->gapo1, rp->gape1, rp->gapo2, rp->gape2, (int)bw);
	void *stop = nullptr;
	rc = bsa_ctx_time_begin_internal(ctx, 0.0, &stop);
	if(rc != BSA_OK) return rc;
	hipError_t e;
	switch(bw / 16){
		case 1:  e = launch_sweep_pw<1>(a, pw, st); break;
		case 2:  e = launch_sweep_pw<2>(a, pw, st); break;
		case 4:  e = launch_sweep_pw<4>(a, pw, st); break;
		case 8:  e = launch_sweep_pw<8>(a, pw, st); break;
		case 16: e = launch_sweep_pw<16>(a, pw, st); break;
		default: e = launch_sweep_pw<0>(a, pw, st); break;
	}
	if(e != hipSuccess) return BSA_E_HIP;
	return bsa_ctx_time_end_internal(ctx, stop);
}

namespace {
struct DevBuf {
	void *p = nullptr;
	~DevBuf(){ if(p) (void)hipFree(p); }
	hipError_t alloc(size_t n){ return hipMalloc(&p, n ? n : 16); }
};
}

extern "C" int bsa_sweep_host(bsa_ctx_t *ctx, const bsa_row_task_t *tasks, size_t ntasks, const bsa_sweep_prog_t *progs, size_t nprogs,
		const uint8_t *queries, const uint64_t *qoff, const uint32_t *qlen, size_t nqueries,
		const bsa_sweep_params_t *par, uint8_t *rows_out, size_t nblocks, bsa_sweep_result_t *results){
	if(!ctx || !par || !results || (nprogs && (!tasks || !progs || !queries || !qoff || !qlen || !nqueries || !nblocks))) return BSA_E_ARG;
	if(nprogs == 0) return BSA_OK;
	hipStream_t st;
	int rc = bsa_ctx_get_stream_internal(ctx, &st);
	if(rc != BSA_OK) return rc;
	const bsa_rows_params_t *rp = &par->rows;
	const size_t blk = bsa_rows_block_bytes(rp->bandwidth, rp->gapo1, rp->gape1, rp->gapo2, rp->gape2);
	size_t qbytes = 0;
	for(size_t k = 0; k < nqueries; k++) qbytes = std::max(qbytes, (size_t)qoff[k] + qlen[k]);
	// every program has at least one task, stays inside the task array and only touches row blocks inside rows_out
	for(size_t k = 0; k < nprogs; k++){
		if(progs[k].ntasks == 0 || (size_t)progs[k].first_task + progs[k].ntasks > ntasks || progs[k].first_block >= nblocks) return BSA_E_ARG;
		const size_t room = nblocks - progs[k].first_block;
		for(size_t t = progs[k].first_task; t < (size_t)progs[k].first_task + progs[k].ntasks; t++)
			if(tasks[t].src >= room || tasks[t].dst >= room) return BSA_E_ARG;
	}
	for(size_t k = 0; k < ntasks; k++) if(tasks[k].query >= nqueries) return BSA_E_ARG;
	DevBuf d_rows, d_tasks, d_progs, d_q, d_qoff, d_qlen, d_res;
#define SWCHK(x) do { if((x) != hipSuccess) return BSA_E_HIP; } while(0)
	SWCHK(d_rows.alloc(nblocks * blk)); SWCHK(d_tasks.alloc(ntasks * sizeof(bsa_row_task_t)));
	SWCHK(d_progs.alloc(nprogs * sizeof(bsa_sweep_prog_t))); SWCHK(d_q.alloc(qbytes + 64));
	SWCHK(d_qoff.alloc(nqueries * 8)); SWCHK(d_qlen.alloc(nqueries * 4)); SWCHK(d_res.alloc(nprogs * sizeof(bsa_sweep_result_t)));
	SWCHK(hipMemsetAsync(d_rows.p, 0, nblocks * blk, st));
	SWCHK(hipMemcpyAsync(d_tasks.p, tasks, ntasks * sizeof(bsa_row_task_t), hipMemcpyHostToDevice, st));
	SWCHK(hipMemcpyAsync(d_progs.p, progs, nprogs * sizeof(bsa_sweep_prog_t), hipMemcpyHostToDevice, st));
	SWCHK(hipMemcpyAsync(d_q.p, queries, qbytes, hipMemcpyHostToDevice, st));
	SWCHK(hipMemcpyAsync(d_qoff.p, qoff, nqueries * 8, hipMemcpyHostToDevice, st));
	SWCHK(hipMemcpyAsync(d_qlen.p, qlen, nqueries * 4, hipMemcpyHostToDevice, st));
	rc = bsa_sweep_run(ctx, (uint8_t*)d_rows.p, (const bsa_row_task_t*)d_tasks.p, (const bsa_sweep_prog_t*)d_progs.p, nprogs,
		(const uint8_t*)d_q.p, (const uint64_t*)d_qoff.p, (const uint32_t*)d_qlen.p, par, (bsa_sweep_result_t*)d_res.p);
	if(rc != BSA_OK) return rc;
	SWCHK(hipMemcpyAsync(results, d_res.p, nprogs * sizeof(bsa_sweep_result_t), hipMemcpyDeviceToHost, st));
	if(rows_out) SWCHK(hipMemcpyAsync(rows_out, d_rows.p, nblocks * blk, hipMemcpyDeviceToHost, st));
	SWCHK(hipStreamSynchronize(st));
#undef SWCHK
	return BSA_OK;
}
