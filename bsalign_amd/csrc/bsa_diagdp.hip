// bsa_diagdp.hip -- the anti-diagonal u8 DP of the reference's MSA refinement:
// maxmat_dp_diag_rowcal bspoa.h:3856-3896 with maxmat_dp_diag_rowcal_init :3752-3761, _prepare :3763-3787 and the fill loop of
// remsa_pedit_rd_bspoacore :3925-3935 (one read of a POA window against the window's column profile, band of 16 W cells
// around the main diagonal of the (MSA column) x (MSA column) matrix).  Step i = x + y takes row i of the two difference
// planes to row i + 1:
//     s = sat_u8(mats0[seq1[c]][c] + mats1[seq0[c]][c])           (0 where the base code is >= 4)
//     even i ("down"):  u = r0[i][c], v = r1[i][c - 1]       odd i ("left"):  u = r0[i][c + 1], v = r1[i][c]
//     h = max(s, u, v)        r0[i + 1][c] = h - v        r1[i + 1][c] = h - u
// Every cell of a row depends only on the row before it: one lane per band cell, a problem = 16 W lanes, 64 / (16 W)
// problems per wave, the neighbour's value through a wave shuffle, the row itself never leaves registers.  A read is a
// chain of 2 (mend - mbeg) - 1 dependent steps of a dozen instructions, so what matters is that nothing in the chain waits
// for memory: the four inputs of a lane (two base codes, two dwords of four counts from the transposed planes that
// k_diagdp_stage builds) are requested eight step-pairs ahead, and the window slides by one cell per step-pair, so only one
// new value per plane and pair is loaded.  All reads of all windows of a batch run side by side.
#include "bsa_common.h"

// planes -> per position one dword of the four counts (mats[s][0..3][j]) for both sides; T holds (mlen + 16 W) dwords per
// side and problem, logical index j at T[j + 8 W]
__global__ void __launch_bounds__(256) k_diagdp_stage(const uint8_t *planes, const bsa_diagdp_prob_t *probs, uint32_t *T, const uint64_t *toff, uint32_t n){
	const uint32_t k = blockIdx.y;
	if(k >= n) return;
	const bsa_diagdp_prob_t pb = probs[k];
	const int bw = 16 * (int)pb.W, half = bw / 2;
	const uint32_t len = pb.mlen + (uint32_t)bw;
	uint32_t *t0 = T + toff[k], *t1 = t0 + len;
	for(uint32_t j = blockIdx.x * 256u + threadIdx.x; j < len; j += gridDim.x * 256u){
		const long lj = (long)j - half;
		uint32_t a = 0, b = 0;
#pragma unroll
		for(int q = 0; q < 4; q++){
			a |= (uint32_t)planes[pb.mats0[q] + lj] << (8 * q);
			b |= (uint32_t)planes[pb.mats1[q] + lj] << (8 * q);
		}
		t0[j] = a; t1[j] = b;
	}
}

template<int W, int PF>
__global__ void __launch_bounds__(64) k_diagdp_fill(const uint8_t *planes, const bsa_diagdp_prob_t *probs, const uint32_t *T, const uint64_t *toff,
		uint8_t *matrix, uint32_t n){
	constexpr int G = 16 * W, half = G / 2, rowlen = G + 2;
	const int lane = threadIdx.x, c = lane % G;
	const uint32_t k = blockIdx.x * (64u / G) + (uint32_t)(lane / G);
	const bool live = k < n;
	const bsa_diagdp_prob_t pb = probs[live ? k : 0u];
	const int mlen = (int)pb.mlen, mbeg = (int)pb.mbeg, mend = live ? (int)pb.mend : mbeg;
	const uint32_t len = pb.mlen + (uint32_t)G;
	const uint32_t *t0 = T + toff[live ? k : 0u] + half, *t1 = t0 + len;           // logical index 0
	const uint8_t *s0 = planes + pb.seq0, *s1 = planes + pb.seq1;
	uint8_t *o0 = matrix + pb.out0, *o1 = matrix + pb.out1;
	const int npairs = mend - mbeg;                   // pair p: the even step i = 2 (mbeg + p) and (all but the last pair) the odd step i + 1
	// row 2 mbeg (maxmat_dp_diag_rowcal_init)
	uint32_t m0 = (c == half - 1) ? 255u : 0u, m1 = (c == half) ? 255u : 0u;
	if(npairs > 0){
		const size_t r = (size_t)(2 * mbeg) * rowlen;
		o0[r + 1 + c] = (uint8_t)m0; o1[r + 1 + c] = (uint8_t)m1;
		if(c == 0){ o0[r] = 0; o1[r] = 0; }
		if(c == G - 1){ o0[r + 1 + G] = 0; o1[r + 1 + G] = 0; }
	}
	// inputs of pair p (even step: x = y = mbeg + p): side 0 at xb + c = mbeg + p - half + c, side 1 at yb + c = mlen - 1 - mbeg - p - half + c;
	// the odd step looks one further on side 0 (x + 1), which is the even step of pair p + 1
	const int xb0 = mbeg - half + c, yb0 = mlen - 1 - mbeg - half + c;
	uint32_t rb0[PF], rb1[PF], rt0[PF], rt1[PF];       // ring: side-0 values of index xb0 + p + 1 (the odd step of pair p), side-1 values of index yb0 - p
	uint32_t b0e = 4, t0e = 0;                        // side-0 values of the even step of the current pair
	if(npairs > 0){ b0e = s0[xb0]; t0e = t0[xb0]; }
#pragma unroll
	for(int q = 0; q < PF; q++){
		const bool in = q < npairs;
		rb0[q] = in ? (uint32_t)s0[xb0 + q + 1] : 4u; rt0[q] = in ? t0[xb0 + q + 1] : 0u;
		rb1[q] = in ? (uint32_t)s1[yb0 - q] : 4u; rt1[q] = in ? t1[yb0 - q] : 0u;
	}
	auto count = [](uint32_t t, uint32_t b) -> uint32_t { return b < 4u ? (t >> (8u * b)) & 0xffu : 0u; };
	for(int p0 = 0; __any(p0 < npairs); p0 += PF){
#pragma unroll
		for(int q = 0; q < PF; q++){
			const int p = p0 + q;
			const bool on = p < npairs;
			const uint32_t b0o = rb0[q], t0o = rt0[q], b1 = rb1[q], tt1 = rt1[q];
			// refill this ring slot for pair p + PF
			{
				const bool in = p + PF < npairs;
				rb0[q] = in ? (uint32_t)s0[xb0 + p + PF + 1] : 4u; rt0[q] = in ? t0[xb0 + p + PF + 1] : 0u;
				rb1[q] = in ? (uint32_t)s1[yb0 - p - PF] : 4u; rt1[q] = in ? t1[yb0 - p - PF] : 0u;
			}
			// even step ("down"): u = r0[c], v = r1[c - 1]
			{
				const uint32_t s = min(count(t0e, b1) + count(tt1, b0e), 255u);
				const uint32_t vn = (uint32_t)__shfl_up((int)m1, 1, G);
				const uint32_t u = m0, v = (c == 0) ? 0u : vn;
				const uint32_t h = max(max(s, u), v);
				const uint32_t n0 = h - v, n1 = h - u;
				if(on){
					const size_t r = (size_t)(2 * (mbeg + p) + 1) * rowlen;
					o0[r + 1 + c] = (uint8_t)n0; o1[r + 1 + c] = (uint8_t)n1;
					if(c == 0){ o0[r] = 0; o1[r] = 0; }
					if(c == G - 1){ o0[r + 1 + G] = 0; o1[r + 1 + G] = 255; }
					m0 = n0; m1 = n1;
				}
			}
			// odd step ("left"): u = r0[c + 1], v = r1[c]  (not after the last pair: the loop ends when x reaches mend)
			{
				const bool on2 = p + 1 < npairs;
				const uint32_t s = min(count(t0o, b1) + count(tt1, b0o), 255u);
				const uint32_t un = (uint32_t)__shfl_down((int)m0, 1, G);
				const uint32_t u = (c == G - 1) ? 0u : un, v = m1;
				const uint32_t h = max(max(s, u), v);
				const uint32_t n0 = h - v, n1 = h - u;
				if(on2){
					const size_t r = (size_t)(2 * (mbeg + p) + 2) * rowlen;
					o0[r + 1 + c] = (uint8_t)n0; o1[r + 1 + c] = (uint8_t)n1;
					if(c == 0){ o0[r] = 255; o1[r] = 0; }
					if(c == G - 1){ o0[r + 1 + G] = 0; o1[r + 1 + G] = 0; }
					m0 = n0; m1 = n1;
				}
			}
			b0e = b0o; t0e = t0o;
		}
	}
}

// The traceback of remsa_pedit_rd_bspoacore (bspoa.h:3965-4040) on the device, so that the two difference planes -- 2.1 bytes a cell -- never
// leave it: from (mend - 1, mend - 1) every step reads the cell's two rows, recomputes the column score and takes the reference's choice in
// the reference's order (left when f alone explains the cell and it is not the band's first cell of an even row; up when e does; diagonal when
// the column score does).  What goes back is two bits a step (0 diagonal, 1 x - 1, 2 y - 1), the score the reference returns (the sum over
// the diagonal steps) and where the walk ended; the caller replays the steps to merge the nodes.  One lane per read: a walk is a chain of
// dependent loads from rows the fill kernel has just written, all reads of all windows of the call side by side.
__global__ void __launch_bounds__(64) k_diagdp_walk(const uint8_t *planes, const bsa_diagdp_prob_t *probs, const uint32_t *T, const uint64_t *toff,
		const uint8_t *matrix, uint32_t n, bsa_diagdp_walk_t *walks, uint32_t *steps, const uint64_t *word_off){
	const uint32_t k = blockIdx.x * 64u + threadIdx.x;
	if(k >= n) return;
	const bsa_diagdp_prob_t pb = probs[k];
	const int G = 16 * (int)pb.W, half = G / 2, rowlen = G + 2;
	const int mlen = (int)pb.mlen, mbeg = (int)pb.mbeg, mend = (int)pb.mend;
	const uint32_t len = pb.mlen + (uint32_t)G;
	const uint32_t *t0 = T + toff[k] + half, *t1 = t0 + len;
	const uint8_t *s0 = planes + pb.seq0, *s1 = planes + pb.seq1;
	const uint8_t *o0 = matrix + pb.out0, *o1 = matrix + pb.out1;
	uint32_t *out = steps + word_off[k];
	auto count = [](uint32_t t, uint32_t b) -> uint32_t { return b < 4u ? (t >> (8u * b)) & 0xffu : 0u; };
	int xi = mend - 1, yi = mend - 1, scr = 0;
	uint32_t nst = 0, acc = 0, status = 0;
	while(xi >= 0 && yi >= 0){
		const int i = xi + yi;
		if(i < mbeg + mbeg) break;
		const int dir = i & 1;
		const int xx = (xi - yi - dir) / 2 + half;
		if(xx < 0 || xx >= G){ status = 1u; break; }
		const uint32_t b0 = s0[xi], b1 = s1[mlen - 1 - yi];
		const int h = (int)min(count(t0[xi], b1) + count(t1[mlen - 1 - yi], b0), 255u);
		const uint8_t *r0 = o0 + (size_t)i * rowlen + 1, *r1 = o1 + (size_t)i * rowlen + 1;
		const int e = dir ? r0[xx + 1] : r0[xx], f = dir ? r1[xx] : r1[xx - 1];
		const int s = f + (int)o0[(size_t)(i + 1) * rowlen + 1 + xx];
		uint32_t bt;
		if(s == f && !(xx == 0 && dir == 0)){ bt = 1u; xi--; }
		else if(s == e){ bt = 2u; yi--; }
		else if(s == h){ bt = 0u; scr += s; xi--; yi--; }
		else { status = 2u; break; }
		acc |= bt << (2u * (nst & 15u));
		nst++;
		if((nst & 15u) == 0u){ out[(nst >> 4) - 1u] = acc; acc = 0; }
	}
	if(nst & 15u) out[nst >> 4] = acc;
	bsa_diagdp_walk_t w;
	w.nsteps = nst; w.score = scr; w.xi = xi; w.yi = yi; w.status = status; w.reserved = 0; w.first_word = word_off[k];
	walks[k] = w;
}

hipError_t bsa_launch_diagdp_walk(const uint8_t *d_planes, const bsa_diagdp_prob_t *d_probs, const uint32_t *d_T, const uint64_t *d_toff, const uint8_t *d_matrix,
		uint32_t n, bsa_diagdp_walk_t *d_walks, uint32_t *d_steps, const uint64_t *d_word_off, hipStream_t st){
	if(n == 0) return hipSuccess;
	hipLaunchKernelGGL(k_diagdp_walk, dim3((n + 63u) / 64u), dim3(64), 0, st, d_planes, d_probs, d_T, d_toff, d_matrix, n, d_walks, d_steps, d_word_off);
	return hipGetLastError();
}

hipError_t bsa_launch_diagdp(const uint8_t *d_planes, const bsa_diagdp_prob_t *d_probs, uint32_t *d_T, const uint64_t *d_toff, uint8_t *d_matrix,
		uint32_t n, uint32_t W, uint32_t max_len, hipStream_t st){
	if(n == 0) return hipSuccess;
	const uint32_t bx = std::min<uint32_t>((max_len + 255u) / 256u, 64u);
	hipLaunchKernelGGL(k_diagdp_stage, dim3(bx ? bx : 1u, n), dim3(256), 0, st, d_planes, d_probs, d_T, d_toff, n);
	const uint32_t ppw = 64u / (16u * W), blocks = (n + ppw - 1) / ppw;
	switch(W){
		case 1: hipLaunchKernelGGL((k_diagdp_fill<1, 8>), dim3(blocks), dim3(64), 0, st, d_planes, d_probs, d_T, d_toff, d_matrix, n); break;
		case 2: hipLaunchKernelGGL((k_diagdp_fill<2, 8>), dim3(blocks), dim3(64), 0, st, d_planes, d_probs, d_T, d_toff, d_matrix, n); break;
		case 4: hipLaunchKernelGGL((k_diagdp_fill<4, 8>), dim3(blocks), dim3(64), 0, st, d_planes, d_probs, d_T, d_toff, d_matrix, n); break;
		default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}
