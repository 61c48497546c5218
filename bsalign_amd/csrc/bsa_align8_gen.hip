// bsa_align8_gen.hip -- generic-bandwidth forward kernel of the 8-bit banded path.
//
// Same algorithm and row records as k_align8_fwd (bsa_align8.hip; reference functions cited there) for ANY bandwidth
// that is a multiple of 16, including the reference's default "bandwidth 0 = whole query" (bsalign.h:3861, where
// every pair has its own W = roundup(qlen, 16) / 16).  The register kernels need W in {1,2,4,8,16,32}; here W is a
// run-time value and the band rows live in LDS (two buffers of (pw+1)*bw int8 per pair: the moved previous row and
// the row being computed), lane j of the pair's 16-lane DPP row walking its W cells one at a time.  It is the
// fall-back, not the fast path: byte-wide LDS traffic per cell instead of registers.
#include "bsa_common.h"
#include "bsa_dpp.h"
#include <algorithm>

template<int PW>
__global__ void __launch_bounds__(256) k_align8_fwd_gen(const Align8Args a, uint32_t lds_per_group, uint32_t gpb, uint32_t nbuf){
	extern __shared__ __attribute__((aligned(16))) int8_t gsm[];
	const int lt = threadIdx.x, j = lt & 15;
	const uint32_t grp = (uint32_t)(lt >> 4);
	const uint32_t g = blockIdx.x * gpb + grp;
	const bool live = grp < gpb && g < a.count;          // gpb = pairs per block that fit the LDS; other DPP rows idle
	const uint32_t ppos = a.first + (live ? g : 0u);
	const uint32_t pair = a.order[ppos];
	const uint32_t qlen = a.qlen[pair];
	uint32_t tlen = a.tlen[pair];
	const uint32_t BW = !live ? 0u : a.bw ? a.bw : ((qlen + 15u) / 16u * 16u);
	const uint32_t W = BW / 16u;
	const uint8_t *qp = a.qst + a.qpoff[pair];
	const uint8_t *tp = a.tst + a.tpoff[pair];
	int *begs = (int*)(a.rows + a.slot_off[ppos]);
	uint8_t *rowp = (uint8_t*)begs + bsa_begs_bytes(tlen);
	if(!live || a.status[pair] != 0u || W == 0u) tlen = 0;
	int8_t *buf0 = gsm + (size_t)(live ? grp : 0u) * lds_per_group;   // [PW+1][BW]
	int8_t *buf1 = buf0 + (size_t)(nbuf - 1u) * (PW + 1) * BW;      // nbuf == 1: the band is the whole query and never moves, no second row
	const uint32_t cells = ((uint32_t)(PW + 1) * W + 3u) & ~3u, blk = cells + 4u;
	const uint32_t tg = (64u / blk) ? (64u / blk) : 1u, tileb = (tg * blk + 15u) & ~15u;
	const int mode = a.mode & 3;
	const int gapo1 = a.gapo1, gape1 = a.gape1, gapo2 = a.gapo2, gape2 = a.gape2;
	const int GapE = trunc8(gape1), GapOE = trunc8(gapo1 + gape1);
	const int GapP = trunc8(gape2), GapQP = trunc8(gapo2 + gape2);
	const int GapOQ = sat8(GapOE - GapQP);
	const int cfirst = (PW == 2) ? (min(a.smin, gapo2 + gape2) - 1 - a.smax + (gapo2 + gape2))
	                             : (min(a.smin, gapo1 + gape1) - 1 - a.smax + (gapo1 + gape1));
	const int dsw = (PW == 2) ? (gapo1 - gapo2) / (gape2 - gape1) : (int)(BW + 1);
	auto newcell_int = [&](int k) -> int { return (k == 0) ? cfirst : ((PW == 2 && k >= dsw) ? gape2 : gape1); };
	auto newcell_cum = [&](int n) -> int { int n1 = min(n, dsw); return cfirst + (n1 - 1) * gape1 + ((PW == 2) ? max(0, n - dsw) * gape2 : 0); };
	auto wave_sync = [&](){
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	};
	int8_t *cur = buf0, *nxt = buf1;        // cur = previous row (natural order), nxt = scratch for the moved row
	int ubA = 0, ubB = 0;
	// ---- row -1 (bsalign.h:2094-2140)
	if(tlen){
		int bs = 0;
		const int first = trunc8(gapo1 + gape1 + a.smin - a.smax);
		const int xp = (PW == 2) ? (gapo2 - gapo1) / (gape1 - gape2) : 0;
		for(uint32_t k = 0; k < W; k++){
			const int p = (int)(j * W + k);
			int v;
			if(mode == BSA_MODE_OVERLAP) v = 0;
			else if(p == 0) v = first;
			else if(PW == 2) v = (p < xp) ? gape1 : gape2;
			else v = gape1;
			cur[p] = (int8_t)v; bs += v;
			if(PW >= 1) cur[BW + p] = BSA_EPI8_MIN;
			if(PW == 2) cur[2 * BW + p] = BSA_EPI8_MIN;
		}
		const int inc = row_iscan16(bs);
		const int base0 = (mode == BSA_MODE_OVERLAP) ? 0 : (a.smax - a.smin);
		ubB = base0 + inc; ubA = ubB - bs;
	}
	// tile record of this lane's block: written straight from LDS (generic path: no register tile buffer; W >= 1 any)
	auto store_row = [&](const int8_t *row, uint32_t rr, uint32_t rbeg_v){
		uint8_t *bp = rowp + ((size_t)(rr / tg) * 16u + (uint32_t)j) * tileb + (rr % tg) * blk;
		// the record is the byte stream u[0..W) | e[0..W) | q[0..W) (padded to `cells`): gathered from LDS four bytes at a
		// time and written as dwords -- byte stores to HBM were the whole run time of this kernel
		{
			uint32_t plane = 0, cell = 0;
			const int8_t *pl = row + (size_t)j * W;
			for(uint32_t d = 0; d < cells / 4u; d++){
				uint32_t v = 0;
#pragma unroll
				for(uint32_t b = 0; b < 4u; b++){
					if(plane <= (uint32_t)PW){
						v |= (uint32_t)(uint8_t)pl[cell] << (8u * b);
						if(++cell == W){ cell = 0; plane++; pl += BW; }
					}
				}
				((uint32_t*)bp)[d] = v;
			}
		}
		*(int*)(bp + cells) = ubA;
		if(j == 0) begs[rr] = (int)rbeg_v;
	};
	wave_sync();
	if(tlen) store_row(cur, 0u, 0u);
	uint32_t rbeg = 0, mov = 0, i = 0;
	while(__any(i < tlen)){
		const bool act = i < tlen;
		const bool moved = (mov != 0u) && (rbeg + BW < qlen);
		{
			const uint32_t room = qlen - (rbeg + BW);
			mov = moved ? min(mov, room) : 0u;
			rbeg += mov;
		}
		int rh;
		if(rbeg) rh = BSA_SCORE_MIN;
		else if(mode == BSA_MODE_OVERLAP || i == 0) rh = 0;
		else if(PW < 2) rh = gapo1 + gape1 * (int)i;
		else rh = max(gapo1 + gape1 * (int)i, gapo2 + gape2 * (int)i);
		// ---- row_movx (bsalign.h:2244-2392): cur -> nxt, then roles swap
		// ubegs exchange area: 17 ints after the two row buffers (other lanes' block scores are needed by row_movx)
		int *sub = (int*)(buf0 + (size_t)nbuf * (PW + 1) * BW);
		if(act){ sub[j] = ubA; if(j == 15) sub[16] = ubB; }
		wave_sync();
		if(act && mov){
			const uint32_t pp = min(mov - 1u, BW - 1u), yy = pp / W, xx = pp % W;
			int s = sub[yy];
			for(uint32_t k = 0; k <= xx; k++) s += cur[yy * W + k];
			rh = s;
			if(mov >= BW){
				for(uint32_t k = 0; k < W; k++){
					const uint32_t p = j * W + k;
					nxt[p] = 0; if(PW >= 1) nxt[BW + p] = 0; if(PW == 2) nxt[2 * BW + p] = 0;
				}
				ubA = ubB = BSA_SCORE_MIN;
			} else {
				const uint32_t cyc = mov / W, m = mov % W, p0 = BW - mov;
				for(uint32_t k = 0; k < W; k++){
					const uint32_t p = j * W + k, src = p + mov;
					if(src < BW){
						nxt[p] = cur[src];
						if(PW >= 1) nxt[BW + p] = cur[BW + src];
						if(PW == 2) nxt[2 * BW + p] = cur[2 * BW + src];
					} else {
						nxt[p] = (int8_t)trunc8(newcell_int((int)(src - BW)));
						if(PW >= 1) nxt[BW + p] = 0;
						if(PW == 2) nxt[2 * BW + p] = 0;
					}
				}
				auto new_ub = [&](uint32_t idx) -> int {
					int v;
					if(idx + cyc < 16u){
						const uint32_t l = idx + cyc;
						v = sub[l];
						for(uint32_t k = 0; k < m; k++) v += cur[l * W + k];
					} else v = sub[16];
					const int nbefore = (int)(idx * W) - (int)p0;
					if(nbefore > 0) v += newcell_cum(nbefore);
					return v;
				};
				ubA = new_ub((uint32_t)j);
				ubB = new_ub((uint32_t)j + 1u);
			}
		}
		wave_sync();
		int8_t *src = (act && mov) ? nxt : cur;       // the (moved) previous row; the new row is written over it in place
		// ---- row_cal (bsalign.h:2727-2793 / 2885-2960 / 3084-3179), S(x, base) from the staged codes
		const int tb = act ? (int)tp[i] : 0;
		const uint32_t mr = (tb == 0) ? a.mrow[0] : (tb == 1) ? a.mrow[1] : (tb == 2) ? a.mrow[2] : a.mrow[3];
		// The query codes of this lane's block sit in LDS behind the exchange area and are re-read from the staged sequence
		// only when the band has moved (never with the whole query as band): a global load inside the cell loops costs its
		// full latency every time (measured: two thirds of this kernel's run time).
		int8_t *qcl = buf0 + (size_t)nbuf * (PW + 1) * BW + 80 + (size_t)j * W;
		if(act && (i == 0u || mov)){
			const uint8_t *qc = qp + rbeg + (size_t)j * W;
			for(uint32_t k0 = 0; k0 < W; k0 += 32u){
				uint32_t w8[8];
#pragma unroll
				for(uint32_t b = 0; b < 8u; b++) if(k0 + 4u * b < W) __builtin_memcpy(&w8[b], qc + k0 + 4u * b, 4);      // the staged sequence is padded
#pragma unroll
				for(uint32_t b = 0; b < 8u; b++){
#pragma unroll
					for(uint32_t c = 0; c < 4u; c++) if(k0 + 4u * b + c < W) qcl[k0 + 4u * b + c] = (int8_t)(w8[b] >> (8u * c));
				}
			}
		}
		auto score = [&](uint32_t k) -> int {
			const uint32_t c = act ? (uint32_t)(uint8_t)qcl[k] : 4u;
			return (c >= 4u) ? BSA_EPI8_MIN : __builtin_amdgcn_sbfe((int)mr, 8u * c, 8u);
		};
		const uint32_t base = j * W;
		int h0;
		{
			int hh = (rh - ubA) + score(0);
			const int u0 = src[base], e0 = (PW >= 1) ? src[BW + base] : 0, q0 = (PW == 2) ? src[2 * BW + base] : 0;
			const int t0 = u0 + ((PW == 0) ? gape1 : (PW == 1) ? e0 : max(e0, q0));
			hh = (hh >= t0) ? min(hh, BSA_EPI8_MAX) : BSA_EPI8_MIN;
			h0 = trunc8(hh);
		}
		int f = BSA_EPI8_MIN, gq = BSA_EPI8_MIN;
		{
			int hc = (j == 0) ? h0 : score(0);
			// Four cells per trip.  Everything a trip needs from LDS -- the cells' bytes and the query codes of the cells after
			// them -- is requested up front, so a trip waits for LDS once, and cells beyond the block's end are computed and
			// thrown away by selects rather than skipped by branches (W differs between the pairs of a wave).
			for(uint32_t k0 = 0; __any(k0 < W); k0 += 4u){
				int u4[4], e4[4], q4[4], s4[4];
#pragma unroll
				for(uint32_t b = 0; b < 4u; b++){
					// up to four bytes past the block are read and not used: still inside this pair's LDS
					u4[b] = (int)src[base + k0 + b];
					e4[b] = (PW >= 1) ? (int)src[BW + base + k0 + b] : 0;
					q4[b] = (PW == 2) ? (int)src[2 * BW + base + k0 + b] : 0;
					s4[b] = score(k0 + b + 1u);
				}
#pragma unroll
				for(uint32_t b = 0; b < 4u; b++){
					const bool ok = k0 + b < W;
					const int uk = u4[b];
					int h, fn, gn = gq;
					if(PW == 0){
						const int ee = sat8(uk + GapE);
						h = max(max(ee, hc), f);
						fn = sat8(sat8(h + GapE) - uk);
					} else if(PW == 1){
						const int ee = sat8(e4[b] + uk);
						h = max(max(ee, hc), f);
						fn = sat8(f + GapE);
						h = sat8(h + GapOE);
						fn = sat8(max(fn, h) - uk);
					} else {
						const int ee = sat8(e4[b] + uk), qq = sat8(q4[b] + uk);
						h = max(max(ee, hc), max(qq, max(f, gq)));
						fn = sat8(f + GapE);
						h = sat8(h + GapOE);
						fn = sat8(max(fn, h) - uk);
						gn = sat8(gq + GapP);
						h = sat8(h - GapOQ);
						gn = sat8(max(gn, h) - uk);
					}
					f = ok ? fn : f; gq = ok ? gn : gq;
					hc = s4[b];                                   // only used by a cell that exists
				}
			}
		}
		f = fpen(f, ubA, ubB, (int)W * gape1, j);
		if(PW == 2) gq = fpen(gq, ubA, ubB, (int)W * gape2, j);
		int htail, ulast = 0, unew0 = 0;
		{
			int v = 0, z = (j == 0) ? h0 : score(0), h = 0;
			for(uint32_t k0 = 0; __any(k0 < W); k0 += 4u){
				int u4[4], e4[4], q4[4], s4[4];
#pragma unroll
				for(uint32_t b = 0; b < 4u; b++){
					u4[b] = (int)src[base + k0 + b];
					e4[b] = (PW >= 1) ? (int)src[BW + base + k0 + b] : 0;
					q4[b] = (PW == 2) ? (int)src[2 * BW + base + k0 + b] : 0;
					s4[b] = score(k0 + b + 1u);
				}
#pragma unroll
				for(uint32_t b = 0; b < 4u; b++){
					const uint32_t k = k0 + b;
					const bool ok = k < W;
					const int uk = u4[b];
					int hn, un, vn, fn, gn = gq, en = 0, qn = 0;
					if(PW == 0){
						const int ee = sat8(uk + GapE);
						hn = max(max(ee, z), f);
						un = sat8(hn - v);
						vn = sat8(hn - uk);
						fn = sat8(sat8(hn + GapE) - uk);
					} else if(PW == 1){
						int ee = sat8(e4[b] + uk);
						hn = max(max(ee, z), f);
						un = sat8(hn - v);
						vn = sat8(hn - uk);
						ee = sat8(ee + GapE); ee = sat8(ee - hn);
						en = max(ee, GapOE);
						fn = sat8(f + GapE);
						hn = sat8(hn + GapOE);
						fn = sat8(max(fn, hn) - uk);
					} else {
						int ee = sat8(e4[b] + uk), qq = sat8(q4[b] + uk);
						hn = max(max(ee, z), max(qq, max(f, gq)));
						un = sat8(hn - v);
						vn = sat8(hn - uk);
						ee = sat8(ee + GapE); ee = sat8(ee - hn);
						en = max(ee, GapOE);
						qq = sat8(qq + GapP); qq = sat8(qq - hn);
						qn = max(qq, GapQP);
						fn = sat8(f + GapE);
						hn = sat8(hn + GapOE);
						fn = sat8(max(fn, hn) - uk);
						gn = sat8(gq + GapP);
						hn = sat8(hn - GapOQ);
						gn = sat8(max(gn, hn) - uk);
					}
					if(act && ok){
						if(PW >= 1) src[BW + base + k] = (int8_t)en;
						if(PW == 2) src[2 * BW + base + k] = (int8_t)qn;
						if(k != 0) src[base + k] = (int8_t)un;
					}
					unew0 = (k == 0) ? un : unew0;
					h = ok ? hn : h; v = ok ? vn : v; f = ok ? fn : f; gq = ok ? gn : gq;
					ulast = ok ? uk : ulast;
					z = s4[b];
				}
			}
			htail = (PW == 0) ? h : (PW == 1) ? sat8(h - GapOE) : sat8(h - GapQP);
		}
		// ---- tail (bsalign.h:2618-2636)
		{
			const int vlast = sat8(htail - ulast);
			const int nB = ubB + vlast;
			const int vsh = DPP_SHR(0, vlast, 1);
			int u0 = sat8(unew0 - vsh);
			int nA = DPP_SHR(0, nB, 1);
			if(j == 0){ nA = ubA + u0; u0 = 0; }
			if(act) src[base] = (int8_t)u0;      // idle DPP rows alias pair 0's buffers: never write them
			ubA = nA; ubB = nB;
		}
		wave_sync();
		if(act) store_row(src, i + 1u, rbeg);
		if(act && mov){ int8_t *t = cur; cur = nxt; nxt = t; }
		// ---- adaptive band (bsalign.h:3331-3349) + global steering (bsalign.h:4006-4021)
		{
			int dsum = ubB - ubA; dsum = dsum < 0 ? -dsum : dsum;
			const int nzsum = row_sum16(dsum);
			const int ub0 = DPP_BCAST(ubA, 0), ub16 = DPP_BCAST(ubB, 15);
			uint32_t nz = (uint32_t)(nzsum / 16);
			nz = nz / max(W, 1u) * 16u / 2u;
			const int noisy = (int)((16u > nz) ? 16u : nz);
			int rbx;
			if(i <= BW / 4u) rbx = 0;
			else if(rbeg + BW >= qlen) rbx = 0;
			else if(ub0 + noisy < ub16) rbx = 2;
			else if(ub0 > ub16 + noisy) rbx = 0;
			else rbx = 1;
			if(mode == BSA_MODE_GLOBAL){
				const int rbz = 2 * max((int)(tlen / max(qlen, 1u)), 1);
				const int rby = (int)((1.0 * (double)i / (double)tlen) * (double)qlen);
				const uint32_t left = tlen - i - 1u;
				if((long long)rbeg + (long long)rbz * (long long)left + (long long)BW <= (long long)(uint32_t)(qlen + (uint32_t)rbz - 1u)){
					mov = 1u + (uint32_t)(qlen - (rbeg + BW)) / max(left, 1u);
				} else if((int)rbeg < rby - (int)BW){
					mov = (uint32_t)(rbx + 1);
				} else if((int)rbeg > rby){
					mov = (uint32_t)max(0, rbx - 1);
				} else mov = (uint32_t)rbx;
			} else mov = (uint32_t)rbx;
		}
		i++;
	}
}

// LDS needed by one pair: two row buffers + the 17-int ubegs exchange area
// two row buffers, the 17-int ubegs exchange area (padded to 80 bytes), the query codes of the band
size_t bsa_align8_gen_lds(uint32_t bw, int pw, uint32_t nbuf){ return ((size_t)nbuf * (pw + 1) * bw + 80 + bw + 15) & ~(size_t)15; }

hipError_t bsa_launch_align8_fwd_gen(const Align8Args &a, int pw, uint32_t max_bw, hipStream_t st){
	if(a.count == 0) return hipSuccess;
	const uint32_t nbuf = (a.bw == 0u) ? 1u : 2u;              // bandwidth 0: every pair's band is its whole query
	const size_t per = bsa_align8_gen_lds(max_bw, pw, nbuf);
	if(per > 160 * 1024) return hipErrorInvalidValue;
	const size_t budget = (per > 64 * 1024) ? 160 * 1024 : 64 * 1024;
	uint32_t gpb = (uint32_t)std::min<size_t>(16, budget / per);
	if(gpb == 0) gpb = 1;
	const uint32_t threads = ((gpb * 16u + 63u) / 64u) * 64u;       // whole waves; surplus DPP rows idle
	const uint32_t blocks = (a.count + gpb - 1) / gpb;
	const size_t lds = (size_t)gpb * per;
	hipError_t e = hipSuccess;
	if(lds > 64 * 1024){
		const void *fn = (pw == 0) ? (const void*)k_align8_fwd_gen<0> : (pw == 1) ? (const void*)k_align8_fwd_gen<1> : (const void*)k_align8_fwd_gen<2>;
		e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if(e != hipSuccess) return e;
	}
	if(pw == 0) hipLaunchKernelGGL((k_align8_fwd_gen<0>), dim3(blocks), dim3(threads), lds, st, a, (uint32_t)per, gpb, nbuf);
	else if(pw == 1) hipLaunchKernelGGL((k_align8_fwd_gen<1>), dim3(blocks), dim3(threads), lds, st, a, (uint32_t)per, gpb, nbuf);
	else hipLaunchKernelGGL((k_align8_fwd_gen<2>), dim3(blocks), dim3(threads), lds, st, a, (uint32_t)per, gpb, nbuf);
	return hipGetLastError();
}
