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

template<int W, int PW>
__global__ void __launch_bounds__(256) k_rows(const RowsArgs a){
	constexpr int BW = W * 16;
	__shared__ __attribute__((aligned(16))) int8_t smem[16 * (((PW + 1) * BW + 17 * 4 + 15) & ~15)];
	constexpr int GROUP_LDS = ((PW + 1) * BW + 17 * 4 + 15) & ~15;
	const int lt = threadIdx.x, j = lt & 15;
	const uint32_t g = (blockIdx.x * 256u + lt) >> 4;
	const bool live = g < a.ntasks;
	const bsa_row_task_t tk = a.tasks[live ? g : 0u];
	int8_t *gl = smem + (lt >> 4) * GROUP_LDS;
	int8_t *su = gl, *se = gl + BW, *sq = gl + 2 * BW;
	int *sub = (int*)(gl + (PW + 1) * BW);
	const int gapo1 = a.gapo1, gape1 = a.gape1, gapo2 = a.gapo2, gape2 = a.gape2;
	const int GapE = trunc8(gape1), GapOE = trunc8(gapo1 + gape1);
	const int GapP = trunc8(gape2), GapQP = trunc8(gapo2 + gape2);
	const int GapOQ = sat8(GapOE - GapQP);
	const int nt_max = a.M + a.refbonus + 1, nt_min = a.X;       // as the POA passes them (bspoa.h:2241, 2226)
	const int type = a.mode & 3;
	auto blkp = [&](uint32_t idx) -> int8_t* { return (int8_t*)(a.rows + (size_t)idx * a.blk); };
	int u[W], e[W], q2[W], ubA = 0, ubB = 0;
	auto load_row = [&](const int8_t *bp){
		const int *ub = (const int*)(bp + (PW + 1) * BW);
#pragma unroll
		for(int k = 0; k < W; k++){
			u[k] = bp[k * 16 + j];
			e[k] = (PW >= 1) ? bp[BW + k * 16 + j] : 0;
			q2[k] = (PW == 2) ? bp[2 * BW + k * 16 + j] : 0;
		}
		ubA = ub[j]; ubB = ub[j + 1];
	};
	auto store_row = [&](int8_t *bp){
		int *ub = (int*)(bp + (PW + 1) * BW);
#pragma unroll
		for(int k = 0; k < W; k++){
			bp[k * 16 + j] = (int8_t)u[k];
			if(PW >= 1) bp[BW + k * 16 + j] = (int8_t)e[k];
			if(PW == 2) bp[2 * BW + k * 16 + j] = (int8_t)q2[k];
		}
		ub[j] = ubA;
		if(j == 15) ub[16] = ubB;
	};
	if(!live) return;          // tasks are whole 16-lane rows: no partial DPP rows
	if(tk.op == BSA_ROW_OP_INIT){
		// ---- row_init (bsalign.h:2094-2140) with max_nt = M + refbonus + 1, min_nt = X
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
			u[k] = v; bs += v;
			e[k] = BSA_EPI8_MIN; q2[k] = BSA_EPI8_MIN;
		}
		const int inc = row_iscan16(bs);
		const int base0 = (type == BSA_MODE_OVERLAP) ? 0 : (nt_max - nt_min);
		ubB = base0 + inc; ubA = ubB - bs;
		store_row(blkp(tk.dst));
		return;
	}
	if(tk.op == BSA_ROW_OP_MERGE){
		// ---- row_merge (bsalign.h:2474-2616): lanes are independent; int16 offsets around a common base, 256 vectors per chunk
		const int8_t *b0 = blkp(tk.src), *b1 = blkp(tk.dst);
		const int *ub0 = (const int*)(b0 + (PW + 1) * BW), *ub1 = (const int*)(b1 + (PW + 1) * BW);
		int s0 = ub0[j], s1 = ub1[j];
		const int end0 = ub0[16], end1 = ub1[16];
		ubA = max(s0, s1);
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
			t0 = s16(t0 + b0[k * 16 + j]);
			t1 = s16(t1 + b1[k * 16 + j]);
			const int m = max(t0, t1);
			u[k] = sat8(s16(m - mprev));
			mprev = m;
			if(PW >= 1){
				const int a0 = s16(t0 + b0[BW + k * 16 + j]), a1 = s16(t1 + b1[BW + k * 16 + j]);
				e[k] = sat8(s16(max(a0, a1) - m));
			}
			if(PW == 2){
				const int a0 = s16(t0 + b0[2 * BW + k * 16 + j]), a1 = s16(t1 + b1[2 * BW + k * 16 + j]);
				q2[k] = sat8(s16(max(a0, a1) - m));
			}
		}
		ubB = max(end0, end1);                      // only lane 15's value is stored (ubegs[16])
		store_row(blkp(tk.dst));
		return;
	}
	// ---- update: row_movx(qoff_dst - qoff_src) then row_cal (bspoa.h:2232-2261)
	load_row(blkp(tk.src));
	const uint32_t movx = tk.qoff_dst - tk.qoff_src;
	int rh;
	if(tk.qoff_src == tk.qoff_dst){
		if(tk.qoff_src) rh = BSA_SCORE_MIN;
		else if(type == BSA_MODE_OVERLAP || tk.toff == 0) rh = 0;
		else if(PW < 2) rh = gapo1 + gape1 * (int)tk.toff;
		else rh = max(gapo1 + gape1 * (int)tk.toff, gapo2 + gape2 * (int)tk.toff);
	} else rh = BSA_SCORE_MIN;                      // replaced below by the moved ubegs[0] when the bands overlap
	if(movx){
		// generic movx through LDS (bsalign.h:2244-2392)
		const int cfirst = (PW == 2) ? (min(nt_min, gapo2 + gape2) - 1 - nt_max + (gapo2 + gape2))
		                             : (min(nt_min, gapo1 + gape1) - 1 - nt_max + (gapo1 + gape1));
		const int dsw = (PW == 2) ? (gapo1 - gapo2) / (gape2 - gape1) : (BW + 1);
		auto newcell_int = [&](int k) -> int { return (k == 0) ? cfirst : ((PW == 2 && k >= dsw) ? gape2 : gape1); };
		auto newcell_cum = [&](int n) -> int { int n1 = min(n, dsw); return cfirst + (n1 - 1) * gape1 + ((PW == 2) ? max(0, n - dsw) * gape2 : 0); };
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
	const uint32_t qlen = a.qlen[tk.query];
	const uint8_t *qp = a.queries + a.qoff[tk.query];
	const int mat = (tk.prof & 1) ? (a.M + a.refbonus) : a.M;
	const bool hpc = !(tk.prof & 2);
	int S[W];
	{
		const uint32_t x0 = tk.qoff_dst + (uint32_t)j * W;
		int cprev = (x0 < qlen) ? (int)qp[x0] : 4;
#pragma unroll
		for(int k = 0; k < W; k++){
			const uint32_t x = x0 + k;
			const int cnext = (x + 1 < qlen) ? (int)qp[x + 1] : 4;
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
	store_row(blkp(tk.dst));
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
	if(bw == 0 || !bsa_align8_supported_bw(bw) || bw / 16 > 16) return BSA_E_UNSUPPORTED;
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
		default: return BSA_E_UNSUPPORTED;
	}
	return e == hipSuccess ? BSA_OK : BSA_E_HIP;
}
