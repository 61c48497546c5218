// bsa_poa_gen.hip -- the POA's per-read seq->graph DP, best end cell and traceback for bands WIDER than the wavefront kernel takes
// (bsa_poa_wf.hip stops at 256 columns): a window's first aligned read has no consensus to place a band by, so its band is the whole
// read (prepare_rd_align_bspoa, bspoa.h:2045-2054, 2109-2111: bandwidth = roundup(seqlen, 16) -- 20 000 columns at C4), and
// BSPOAPar::bandwidth itself may be any width.  Same program (nodes in completion order / in-edges / candidates, include/bsalign_hip.h),
// same results and step words as k_poa_wf: bsa_poa_graph_run dispatches here when bsa_poa_graph_supported declines for the width alone.
//
// Reference: align_rd_bspoacore bspoa.h:2515-2618 (dpalign_row_update_bspoa :2232-2261 = row_movx bsalign.h:2244-2392 + row_cal
// :2885-2960 / :3084-3179, dpalign_row_merge_bspoa :2263-2272 = row_merge bsalign.h:2474-2616), the end-of-alignment candidates
// bspoa.h:2547-2606 with row_max bsalign.h:3213-3329, and the walk of alignment2graph_bspoa bspoa.h:2274-2513.
//
// Formulation: ABSOLUTE scores (inside the score guard of bsa_poa_graph_supported none of the reference's int8 operations saturates, so
// its stored differences are exact -- the same argument and the same guard as k_poa_wf), one WORKGROUP of 1024 threads per read, one graph
// node per trip: thread t owns C consecutive cells of the node's row.  What the inputs offer a cell (diagonal + S, E, Q, merged rows) is
// cell-local; the two serial chains of a row -- F and G, the horizontal gap states -- are max-plus recurrences
//     F(p) = max(inj(p), F(p-1) + gape1, N(p-1) + gapo1 + gape1),      inj = "H of the previous row - 63" at the start of every running
// block of the reference's striping (bsalign.h:2909-2931), i.e. prefix maxima: a thread folds its C cells, one exclusive scan over the
// workgroup (DPP-free: wave shuffles + one LDS hop) gives every thread the chain entering its cells.  (Paths from one chain through H
// into the other are dropped as in k_poa_wf: never better than staying, gapo <= 0 and gape1 <= gape2.)  Rows stay in HBM as
// {H int32, e | q << 8}: 8 bytes a cell -- a 20 kbp read against a 20 k-node chain is 3.2 GB, so the launcher runs as many programs
// side by side as BSA_POA_GEN_WS_GB (default 48) holds.  Then, by the same workgroup: the candidates in program order (runs of
// "node complete" candidates evaluated a thread each, the first strictly greatest kept; an edge into the tail with row_max in the
// reference's block order), and the walk by ONE thread, every step's loads issued together.
#include "bsa_common.h"
#include <algorithm>
#include <vector>
#include <cstring>
#include <cstdlib>

extern "C" int bsa_ctx_get_stream_internal(bsa_ctx_t *ctx, hipStream_t *st);
extern "C" int bsa_ctx_time_begin_internal(bsa_ctx_t *ctx, double cells, void **stop_event);
extern "C" int bsa_ctx_time_end_internal(bsa_ctx_t *ctx, void *stop_event);
extern "C" int bsa_ctx_scratch_internal(bsa_ctx_t *ctx, int slot, size_t bytes, void **out);

#define GEN_NT 1024
#define GEN_NEG (2 * BSA_SCORE_MIN)

struct GenArgs {
	const bsa_poa_node_t *nodes; const bsa_poa_edge_t *edges; const bsa_poa_cand_t *cands; const bsa_poa_prog_t *progs;
	const uint8_t *queries;
	int2 *rows; int32_t *u0;                          // rows of the programs of this launch back to back: program k's node i at rowbase[k] + i
	const uint64_t *rowbase;
	bsa_poa_result_t *res; uint32_t *steps; uint32_t *packed; unsigned long long *packed_used;
	uint32_t first_prog;
	uint32_t bw, W;
	int32_t mode, M, X, refbonus, O, E, Q, P, T;
	int32_t c0, d, head_u0, xp;
};

struct GenCell { int h, e, q; };
static __device__ __forceinline__ GenCell gen_unpack(int2 v){ GenCell c; c.h = v.x; c.e = (int)(int8_t)(v.y & 0xFF); c.q = (int)(int8_t)((v.y >> 8) & 0xFF); return c; }
static __device__ __forceinline__ int2 gen_pack(int h, int e, int q){ return make_int2(h, (e & 0xFF) | ((q & 0xFF) << 8)); }

// S(x) of the profile (v.base == u.base) * 2 + v.bonus (bspoa.h:2199-2215, 2588; the hpc bonus of bsalign.h:2194-2221)
static __device__ __forceinline__ int gen_score(const GenArgs &a, const uint8_t *q, uint32_t slen, uint32_t x, int base, int bonus, int same){
	if(x >= slen) return BSA_EPI8_MIN;
	const int qx = q[x];
	int s = (qx == base) ? a.M + (bonus ? a.refbonus : 0) : a.X;
	if(!same && x + 1 < slen && qx != (int)q[x + 1]) s += 1;
	return s;
}

// cell p of row `ru` after row_movx(movx): a real cell, a synthetic cell behind the row's end (bsalign.h:2357-2389), or the dead row of a
// move by at least the bandwidth (:2253-2259)
template<int PW>
static __device__ __forceinline__ GenCell gen_post(const GenArgs &a, const int2 *ru, uint32_t movx, uint32_t p){
	GenCell c;
	if(movx >= a.bw){ c.h = BSA_SCORE_MIN; c.e = 0; c.q = 0; return c; }
	if(p + movx < a.bw) return gen_unpack(ru[p + movx]);
	const int k = (int)(p + movx - a.bw);
	int v = ru[a.bw - 1].x + a.c0;
	if(k < a.d) v += k * a.E; else v += (a.d - 1) * a.E + (k - a.d + 1) * a.P;
	c.h = v; c.e = 0; c.q = 0;
	return c;
}

// exclusive max-plus scan over the workgroup: thread t gets max over s < t of (v_s), GEN_NEG for t = 0
static __device__ __forceinline__ int gen_excl_max(int v, int *lds, int t){
	int inc = v;
#pragma unroll
	for(int o = 1; o < 64; o <<= 1){ const int x = __shfl_up(inc, o); if((t & 63) >= o) inc = max(inc, x); }
	const int wv = t >> 6;
	if((t & 63) == 63) lds[wv] = inc;
	__syncthreads();
	int carry = GEN_NEG;
	for(int w = 0; w < wv; w++) carry = max(carry, lds[w]);
	int ex = __shfl_up(inc, 1);
	if((t & 63) == 0) ex = GEN_NEG;
	__syncthreads();
	return max(ex, carry);
}

template<int PW, int C>
__global__ void __launch_bounds__(GEN_NT) k_poa_gen(const GenArgs a){
	__shared__ int sF[GEN_NT / 64], sG[GEN_NT / 64];
	__shared__ long long sbest[GEN_NT / 64];
	__shared__ int sres[8];
	const int t = threadIdx.x;
	const bsa_poa_prog_t pg = a.progs[a.first_prog + blockIdx.x];
	const bsa_poa_node_t *nodes = a.nodes + pg.first_node;
	const bsa_poa_edge_t *edges = a.edges + pg.first_edge;
	const bsa_poa_cand_t *cands = a.cands + pg.first_cand;
	const uint8_t *query = a.queries + pg.query_off;
	const uint32_t slen = pg.slen, bw = a.bw, W = a.W;
	int2 *rows = a.rows + a.rowbase[blockIdx.x] * (size_t)bw;
	int32_t *u0 = a.u0 + a.rowbase[blockIdx.x];
	bsa_poa_result_t *res = a.res + a.first_prog + blockIdx.x;
	const int mode = a.mode & 3;
	const int OE = a.O + a.E, QP = a.Q + a.P;
	const uint32_t p0 = (uint32_t)t * C;
#define ROW(i) (rows + (size_t)(i) * bw)
	// ---- the head's row: row_init (bsalign.h:2094-2140) as absolute cells
	for(uint32_t p = t; p < bw; p += GEN_NT){
		int h;
		if(mode == BSA_MODE_OVERLAP) h = 0;
		else if(PW == 2){ const int n1 = min((int)p, a.xp - 1); h = OE + n1 * a.E + ((int)p - n1) * a.P; }
		else h = OE + (int)p * a.E;
		ROW(0)[p] = gen_pack(h, PW >= 1 ? BSA_EPI8_MIN : 0, PW == 2 ? BSA_EPI8_MIN : 0);
	}
	if(t == 0) u0[0] = a.head_u0;
	__syncthreads();
	// ---- forward: one node per trip
	for(uint32_t i = 1; i < pg.nnodes; i++){
		const bsa_poa_node_t v = nodes[i];
		int N[C], aF[C], aG[C], Ei[C], Qi[C], EX[C], QX[C];
#pragma unroll
		for(int c = 0; c < C; c++){ N[c] = GEN_NEG; aF[c] = GEN_NEG; aG[c] = GEN_NEG; Ei[c] = GEN_NEG; Qi[c] = GEN_NEG; EX[c] = GEN_NEG; QX[c] = GEN_NEG; }
		int Nprev = GEN_NEG;                                     // N of the cell in front of this thread's first one
#pragma unroll
		for(int k = 0; k < 2; k++){
			const bsa_poa_input_t in = v.in[k];
			if(!(in.toff_kind & BSA_POA_IN_PRESENT)) continue;
			const int2 *ru = ROW(in.src);
			if(in.toff_kind & BSA_POA_IN_MERGE){
				// the finished row of a partial node at the same band offset, taken as it is: cell-wise maximum of H, E, Q (row_merge)
#pragma unroll
				for(int c = 0; c < C; c++){
					const uint32_t p = p0 + c;
					if(p < bw){ const GenCell x = gen_unpack(ru[p]); N[c] = max(N[c], x.h); EX[c] = max(EX[c], x.h + x.e); QX[c] = max(QX[c], x.h + x.q); }
				}
				if(p0 >= 1 && p0 - 1 < bw) Nprev = max(Nprev, ru[p0 - 1].x);
				continue;
			}
			const uint32_t movx = in.movx, urpos = v.rpos - movx, toff = in.toff_kind & BSA_POA_IN_TOFF;
			const int same = (in.toff_kind & BSA_POA_IN_SAME) != 0;
			// cells p0 - 2 .. p0 + C - 1 of the moved row (p0 - 2: the diagonal of the cell in front, for Nprev)
			GenCell prev2, prev1;
			prev2.h = prev2.e = prev2.q = 0; prev1 = prev2;
			if(p0 >= 2 && p0 - 2 < bw) prev2 = gen_post<PW>(a, ru, movx, p0 - 2);
			if(p0 >= 1 && p0 - 1 < bw) prev1 = gen_post<PW>(a, ru, movx, p0 - 1);
			// what this input offers the cell in front (p0 - 1 >= 1 or == 0 with the seed rule): only its N is needed
			if(p0 >= 1 && p0 - 1 < bw){
				const uint32_t p = p0 - 1, x = v.rpos + p;
				const int S = gen_score(a, query, slen, x, v.base, v.flags & 1, same);
				int mc;
				if(p == 0){
					int ub0, rh;
					if(movx == 0) ub0 = u0[in.src]; else if(movx < bw) ub0 = ru[movx - 1].x; else ub0 = BSA_SCORE_MIN;
					if(movx == 0){
						if(urpos) rh = BSA_SCORE_MIN;
						else if(mode == BSA_MODE_OVERLAP || toff == 0) rh = 0;
						else if(PW < 2) rh = a.O + a.E * (int)toff;
						else rh = max(a.O + a.E * (int)toff, a.Q + a.P * (int)toff);
					} else if(movx <= bw) rh = ub0; else rh = BSA_SCORE_MIN;
					int h0 = rh - ub0 + S;
					const int tt = (prev1.h - ub0) + (PW == 0 ? a.E : PW == 1 ? prev1.e : max(prev1.e, prev1.q));
					if(h0 >= tt){ if(h0 > BSA_EPI8_MAX) h0 = BSA_EPI8_MAX; } else h0 = BSA_EPI8_MIN;
					mc = ub0 + h0;
				} else mc = prev2.h + S;
				int n = max(mc, prev1.h + (PW == 0 ? a.E : prev1.e));
				if(PW == 2) n = max(n, prev1.h + prev1.q);
				Nprev = max(Nprev, n);
			}
			GenCell left = prev1;                                    // the moved row's cell p - 1
#pragma unroll
			for(int c = 0; c < C; c++){
				const uint32_t p = p0 + c;
				if(p >= bw) continue;
				const uint32_t x = v.rpos + p;
				const GenCell cur = gen_post<PW>(a, ru, movx, p);
				const int S = gen_score(a, query, slen, x, v.base, v.flags & 1, same);
				int mc, fl = GEN_NEG;
				if(p == 0){
					// the seed of band cell 0 (bsalign.h:2899-2907) with rh as dpalign_row_update_bspoa picks it (bspoa.h:2242-2254)
					int ub0, rh;
					if(movx == 0) ub0 = u0[in.src]; else if(movx < bw) ub0 = ru[movx - 1].x; else ub0 = BSA_SCORE_MIN;
					if(movx == 0){
						if(urpos) rh = BSA_SCORE_MIN;
						else if(mode == BSA_MODE_OVERLAP || toff == 0) rh = 0;
						else if(PW < 2) rh = a.O + a.E * (int)toff;
						else rh = max(a.O + a.E * (int)toff, a.Q + a.P * (int)toff);
					} else if(movx <= bw) rh = ub0; else rh = BSA_SCORE_MIN;
					int h0 = rh - ub0 + S;
					const int tt = (cur.h - ub0) + (PW == 0 ? a.E : PW == 1 ? cur.e : max(cur.e, cur.q));
					if(h0 >= tt){ if(h0 > BSA_EPI8_MAX) h0 = BSA_EPI8_MAX; } else h0 = BSA_EPI8_MIN;
					mc = ub0 + h0;
					fl = ub0 + BSA_EPI8_MIN;
				} else {
					mc = left.h + S;
					if(p % W == 0) fl = left.h + BSA_EPI8_MIN;
				}
				const int ein = cur.h + (PW == 0 ? a.E : cur.e);
				N[c] = max(N[c], max(mc, ein));
				Ei[c] = max(Ei[c], ein);
				if(PW == 2){ const int qin = cur.h + cur.q; N[c] = max(N[c], qin); Qi[c] = max(Qi[c], qin); }
				aF[c] = max(aF[c], fl);
				if(PW == 2) aG[c] = max(aG[c], fl);
				left = cur;
			}
		}
		// ---- the two chains: fold this thread's cells, scan the workgroup, fix up
		// a(p) = max(inj(p), N(p - 1) + gapoe); F_in(p) = max(a(p), F_in(p - 1) + gape)
		int Fl[C], Gl[C];
		{
			int np = Nprev, f = GEN_NEG, g = GEN_NEG;
#pragma unroll
			for(int c = 0; c < C; c++){
				const uint32_t p = p0 + c;
				const int af = (p == 0) ? aF[c] : max(aF[c], np + (PW == 0 ? a.E : OE));
				f = max(af, f + a.E); Fl[c] = f;
				if(PW == 2){ const int ag = (p == 0) ? aG[c] : max(aG[c], np + QP); g = max(ag, g + a.P); Gl[c] = g; }
				np = N[c];
			}
		}
		const int KF = C * a.E, KG = C * a.P;
		// the chain entering this thread's first cell by pure extension from the threads in front: (t - 1) K + max_{s < t} (L_s - s K), L_s = Fl_s[C - 1] + gape
		const int vF = (Fl[C - 1] + a.E) - t * KF;
		int FcIn = gen_excl_max(vF, sF, t);
		FcIn = (t == 0) ? GEN_NEG : FcIn + (t - 1) * KF;
		int GcIn = GEN_NEG;
		if(PW == 2){
			const int vG = (Gl[C - 1] + a.P) - t * KG;
			GcIn = gen_excl_max(vG, sG, t);
			GcIn = (t == 0) ? GEN_NEG : GcIn + (t - 1) * KG;
		}
		int2 *rv = ROW(i);
		{
			int fc = FcIn, gc = GcIn;
#pragma unroll
			for(int c = 0; c < C; c++){
				const uint32_t p = p0 + c;
				if(p >= bw) continue;
				const int F = max(Fl[c], fc), G = (PW == 2) ? max(Gl[c], gc) : GEN_NEG;
				int H = max(N[c], F);
				if(PW == 2) H = max(H, G);
				int e1 = 0, q1 = 0;
				if(PW >= 1) e1 = max(max(Ei[c] + a.E, H + OE), EX[c]) - H;
				if(PW == 2) q1 = max(max(Qi[c] + a.P, H + QP), QX[c]) - H;
				rv[p] = gen_pack(H, e1, q1);
				if(p == 0) u0[i] = H;
				fc += a.E; gc += a.P;
			}
		}
		__syncthreads();
	}
	// ---- the end-of-alignment candidates in the reference's visiting order (bspoa.h:2549-2578, 2593-2603): strictly greater replaces
	int maxscr = BSA_SCORE_MIN, maxidx = -1, maxoff = -1;
	{
		const int sl = (int)slen;
		uint32_t k = 0;
		while(k < pg.ncands){
			if(cands[k].kind == 1){
				// a run of "node complete, its band reaches the read end" candidates: a thread each, the first of the greatest
				uint32_t k2 = k;
				while(k2 < pg.ncands && k2 - k < GEN_NT && cands[k2].kind == 1) k2++;
				long long key = (long long)0x8000000000000000ull;
				if(k + t < k2){
					const uint32_t nd = cands[k + t].node;
					const int s = ROW(nd)[sl - 1 - (int)nodes[nd].rpos].x + a.T;
					key = ((long long)s << 32) | (long long)(0x7FFFFFFFu - (uint32_t)t);
				}
				for(int o = 32; o > 0; o >>= 1){ const long long ok = __shfl_xor(key, o); if(ok > key) key = ok; }
				if((t & 63) == 0) sbest[t >> 6] = key;
				__syncthreads();
				for(int w = 0; w < GEN_NT / 64; w++) if(sbest[w] > key) key = sbest[w];
				__syncthreads();
				if(key != (long long)0x8000000000000000ull){
					const int s = (int)(key >> 32); const uint32_t who = 0x7FFFFFFFu - (uint32_t)(key & 0xFFFFFFFFll);
					if(s > maxscr){ maxscr = s; maxidx = (int)cands[k + who].node; maxoff = sl - 1; }
				}
				k = k2;
			} else {
				const uint32_t nd = cands[k].node;
				const int rpos = (int)nodes[nd].rpos;
				const int2 *ru = ROW(nd);
				const int mo = min(sl, rpos + (int)bw) - 1;
				int s = ru[mo - rpos].x;
				if(sl > mo + 1){ const int n = sl - mo - 1; s += (PW < 2) ? a.O + a.E * n : max(a.O + a.E * n, a.Q + a.P * n); }
				s += a.T;
				if(s > maxscr){ maxscr = s; maxidx = (int)nd; maxoff = mo; }
				if(mode == BSA_MODE_OVERLAP){
					// row_max (bsalign.h:3213-3329): the best cell of every running block (the first on ties), blocks compared in the order of the
					// reference's register reduction (0, 4, 8, 12, 1, 5, ...); 64 threads a block
					const int j = t >> 6, sub = t & 63;
					long long bk = (long long)0x8000000000000000ull;
					for(uint32_t c = sub; c < W; c += 64){
						const long long key = ((long long)ru[(uint32_t)j * W + c].x << 32) | (long long)(0x7FFFFFFFu - c);
						if(key > bk) bk = key;
					}
					for(int o = 32; o > 0; o >>= 1){ const long long ok = __shfl_xor(bk, o); if(ok > bk) bk = ok; }
					if(sub == 0) sbest[j] = bk;
					__syncthreads();
					int bs = 0; uint32_t bp = 0;
					for(int kk = 0; kk < 16; kk++){
						const int jj = (kk & 3) * 4 + (kk >> 2);
						const long long key = sbest[jj];
						const int mx = (int)(key >> 32); const uint32_t ai = 0x7FFFFFFFu - (uint32_t)(key & 0xFFFFFFFFll);
						if(kk == 0 || mx > bs){ bs = mx; bp = (uint32_t)jj * W + ai; }
					}
					__syncthreads();
					if(bs > maxscr){ maxscr = bs; maxidx = (int)nd; maxoff = (int)bp + rpos; }
				}
				k++;
			}
		}
	}
	// ---- the walk of alignment2graph_bspoa (bspoa.h:2274-2513): one thread; one word a step (node << 3 | bt)
	if(t == 0){
		uint32_t *ev = a.steps + pg.first_event;
		const long cap = (long)pg.event_cap;
		long ne = 0;
		int status = BSA_POA_ST_OK, fin_node = 0, fin_x = 0;
		if(maxidx < 0) status = BSA_POA_ST_NOCAND;
		else {
			uint32_t nidx = (uint32_t)maxidx, n = (uint32_t)maxidx, bt = 0xFFFFFFFFu;
			int x = maxoff, Hs0 = 0, Hs1, Hs2 = 0;
#define USX(i, p) ((p) == 0 ? ROW(i)[0].x - u0[i] : ROW(i)[p].x - ROW(i)[(p) - 1].x)
#define EMIT(nn, bb) do { if(ne >= cap){ status = BSA_POA_ST_EVENTS; goto done; } ev[ne++] = ((uint32_t)(nn) << 3) | (uint32_t)(bb); } while(0)
			if(x - (int)nodes[n].rpos < 0 || x - (int)nodes[n].rpos >= (int)bw){ status = BSA_POA_ST_TRACE; goto done; }
			Hs1 = ROW(n)[x - (int)nodes[n].rpos].x;
			for(;;){
				if(n == 0u || x < 0) break;                          // the head is node 0 of a program
				if(bt == 2u || bt == 4u){
					bool found = false;
					EMIT(n, bt);
					const uint32_t nin = nodes[n].n_in, fin = nodes[n].first_in;
					for(uint32_t k = 0; k < nin; k++){
						const uint32_t w = edges[fin + k].src; const int wr = (int)nodes[w].rpos;
						if(x < wr || x >= wr + (int)bw) continue;
						const GenCell cw = gen_unpack(ROW(w)[x - wr]);
						Hs0 = cw.h;
						const int q = (bt == 2u) ? (PW ? cw.e : OE) : cw.q;
						if(Hs0 + q != Hs1) continue;
						n = w;
						if(q == ((bt == 2u) ? OE : QP)){ bt = 0xFFFFFFFFu; Hs1 = Hs0; Hs2 = 0; }
						else { Hs1 -= (bt == 2u) ? a.E : a.P; Hs2++; }
						found = true;
						break;
					}
					if(!found){ status = BSA_POA_ST_TRACE; goto done; }    // (the reference loops on the same state for ever: bspoa.h:2351-2357 is DEBUG only)
					continue;
				} else if(bt == 1u){
					EMIT(n, bt);
					const int tt = (PW == 2) ? max(a.O + a.E * Hs2, a.Q + a.P * Hs2) : a.O + a.E * Hs2;
					x--;
					if(Hs0 + tt == Hs1){ bt = 0xFFFFFFFFu; Hs1 = Hs0; Hs2 = 0; }
					else if(x >= 0){
						const int p = x - (int)nodes[n].rpos;
						if(p < 0){ status = BSA_POA_ST_TRACE; goto done; }
						Hs0 -= USX(n, p);
						Hs2++;
					}
					continue;
				} else if(bt == 0u){
					EMIT(n, bt);
					x--;
					n = nidx;
					bt = 0xFFFFFFFFu;
				} else {
					uint32_t btc = 0, bti = 0xFFFFFFFFu, bnode = 0; int bh = 0;
					const uint32_t nin = nodes[n].n_in, fin = nodes[n].first_in;
					const int nbase = nodes[n].base, nbonus = nodes[n].flags & 1;
					for(uint32_t k = 0; k < nin; k++){
						const bsa_poa_edge_t ed = edges[fin + k];
						const uint32_t w = ed.src, cov = ed.cov; const int wr = (int)nodes[w].rpos;
						int ft = 0;
						if(x < wr || x > (int)bw + wr) continue;
						const int p = x - wr;
						// (everything the three tests read, requested together)
						const int2 cL = (p >= 1) ? ROW(w)[p - 1] : make_int2(0, 0);
						const int2 cC = (p < (int)bw) ? ROW(w)[p] : make_int2(0, 0);
						const int uw = u0[w];
						if(p == (int)bw){ Hs0 = cL.x; ft |= (1 << 2) | (1 << 4); }
						else if(p == 0){
							Hs0 = uw;
							if(wr == 0 && (mode == BSA_MODE_OVERLAP || w == 0u)) ft |= 1 << 15;
							else ft |= 1 << 0;
						} else Hs0 = cL.x;
						int s = gen_score(a, query, slen, (uint32_t)x, nbase, nbonus, (int)nodes[w].base == nbase);
						if(ft & (1 << 15)) s -= uw;
						const GenCell cc = gen_unpack(cC);
						const int us = (p == 0) ? cc.h - uw : cc.h - cL.x;
						int scr[3];
						scr[0] = (ft & (1 << 0)) ? BSA_SCORE_MIN : s;
						scr[1] = (ft & (1 << 2)) ? BSA_SCORE_MIN : us + (PW ? cc.e : a.E);
						scr[2] = (ft & (1 << 4)) ? BSA_SCORE_MIN : (PW == 2 ? us + cc.q : -BSA_SCORE_MIN);
#pragma unroll
						for(int ii = 0; ii < 3; ii++){
							if(Hs0 + scr[ii] == Hs1){
								if(cov > btc || (cov == btc && ii == 0 && (bti & 0xFFu) != 0u)){ bti = (uint32_t)ii; btc = cov; bnode = w; bh = Hs0; }
							}
						}
					}
					if(bti == 0xFFFFFFFFu){
						const int p = x - (int)nodes[n].rpos;
						if(p < 0 || p >= (int)bw){ status = BSA_POA_ST_TRACE; goto done; }
						bt = 1u; Hs2 = 1;
						Hs0 = Hs1 - USX(n, p);
					} else if(bti == 0u){ bt = 0u; nidx = bnode; Hs1 = bh; Hs2 = 0; }
					else if(bti == 1u){ bt = 2u; Hs2 = 1; }
					else { bt = 4u; Hs2 = 1; }
				}
			}
			fin_node = (int)n; fin_x = x;
#undef USX
#undef EMIT
		}
done:
		sres[0] = status; sres[1] = (int)ne; sres[2] = fin_node; sres[3] = fin_x;
		const unsigned long long at = (status == BSA_POA_ST_OK && ne > 0) ? atomicAdd(a.packed_used, (unsigned long long)ne) : 0ull;
		sres[4] = (int)(uint32_t)at;
		bsa_poa_result_t r;
		r.maxscr = maxscr; r.maxidx = maxidx; r.maxoff = maxoff; r.status = status; r.nevents = (status == BSA_POA_ST_OK) ? (int)ne : 0;
		r.fin_node = fin_node; r.fin_x = fin_x; r.reserved = (int)(uint32_t)at;
		*res = r;
	}
	__syncthreads();
	// the steps of a finished walk are appended to the packed output (the order programs finish in)
	if(sres[0] == BSA_POA_ST_OK){
		const uint32_t ne = (uint32_t)sres[1], at = (uint32_t)sres[4];
		const uint32_t *ev = a.steps + pg.first_event;
		for(uint32_t k = t; k < ne; k += GEN_NT) a.packed[at + k] = ev[k];
	}
#undef ROW
}

// the score guard of bsa_poa_graph_supported (exact arithmetic = the reference's int8 arithmetic), without its width and LDS limits
extern "C" int bsa_poa_graph_gen_supported(const bsa_sweep_params_t *par){
	if(!par) return 0;
	const bsa_rows_params_t *rp = &par->rows;
	const uint32_t bw = (rp->bandwidth + 15u) / 16u * 16u;
	if(bw < 16u || bw > 32u * GEN_NT) return 0;
	const int pw = bsa_get_piecewise(rp->gapo1, rp->gape1, rp->gapo2, rp->gape2, (int)bw);
	const int m = rp->M + rp->refbonus + 1, n = -rp->X, ge = -rp->gape1, go = -rp->gapo1;
	if(m < 0 || n < 0 || ge < 0 || go < 0 || rp->M < 0 || rp->refbonus < 0) return 0;
	int g = go + ge;
	if(pw == 2){
		const int ge2 = -rp->gape2, go2 = -rp->gapo2;
		if(ge2 < 0 || go2 < 0 || rp->gape1 > rp->gape2) return 0;
		g = std::max(g, go2 + ge2);
	}
	if(m + 3 * g > 64 || n + m + g > 100 || m + 2 * n > 128) return 0;      // (m + 2 n: the head row's seed (min - max) + S stays a byte on a mismatch, bsalign.h:2899-2910)
	if(std::min((int)rp->X, -g) - 1 - m - g < -100) return 0;
	return 1;
}

// same contract as bsa_poa_graph_run (device pointers, asynchronous on the context stream but for the table of row bases it uploads)
int bsa_poa_graph_gen_run(bsa_ctx_t *ctx, const bsa_poa_node_t *d_nodes, size_t nnodes, const bsa_poa_edge_t *d_edges, const bsa_poa_cand_t *d_cands,
		const bsa_poa_prog_t *d_progs, size_t nprogs, const uint8_t *d_queries, const bsa_sweep_params_t *par,
		bsa_poa_result_t *d_results, uint32_t *d_steps, uint32_t *d_packed, uint64_t *d_packed_used){
	(void)nnodes;
	if(!bsa_poa_graph_gen_supported(par)) return BSA_E_UNSUPPORTED;
	hipStream_t st;
	int rc = bsa_ctx_get_stream_internal(ctx, &st);
	if(rc != BSA_OK) return rc;
	const bsa_rows_params_t *rp = &par->rows;
	const uint32_t bw = (rp->bandwidth + 15u) / 16u * 16u;
	const int pw = bsa_get_piecewise(rp->gapo1, rp->gape1, rp->gapo2, rp->gape2, (int)bw);
	// the programs' node counts (the caller's tables live on the device)
	std::vector<bsa_poa_prog_t> progs(nprogs);
	if(hipMemcpyAsync(progs.data(), d_progs, nprogs * sizeof(bsa_poa_prog_t), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return BSA_E_HIP;
	if(hipMemsetAsync(d_packed_used, 0, 8, st) != hipSuccess) return BSA_E_HIP;
	size_t budget = (size_t)48 << 30;
	if(const char *e = bsa_env("BSA_POA_GEN_WS_GB")){ const double v = atof(e); if(v > 0) budget = (size_t)(v * 1073741824.0); }
	GenArgs a;
	memset(&a, 0, sizeof(a));
	a.nodes = d_nodes; a.edges = d_edges; a.cands = d_cands; a.progs = d_progs; a.queries = d_queries;
	a.res = d_results; a.steps = d_steps; a.packed = d_packed; a.packed_used = (unsigned long long*)d_packed_used;
	a.bw = bw; a.W = bw / 16u;
	a.mode = rp->mode; a.M = rp->M; a.X = rp->X; a.refbonus = rp->refbonus;
	a.O = rp->gapo1; a.E = rp->gape1; a.Q = rp->gapo2; a.P = rp->gape2; a.T = par->T;
	{
		const int nt_max = a.M + a.refbonus + 1, nt_min = a.X;
		const int goe = (pw == 2) ? a.Q + a.P : a.O + a.E;
		a.c0 = std::min(nt_min, goe) - 1 - nt_max + goe;
		a.d = (pw == 2) ? (a.O - a.Q) / (a.P - a.E) : (int)bw + 1;
		a.xp = (pw == 2) ? (a.Q - a.O) / (a.E - a.P) : 1;
		a.head_u0 = ((a.mode & 3) == BSA_MODE_OVERLAP) ? 0 : nt_max - nt_min;
	}
	const uint32_t cells = (bw + GEN_NT - 1) / GEN_NT;
	const int C = cells <= 1 ? 1 : cells <= 2 ? 2 : cells <= 4 ? 4 : cells <= 8 ? 8 : cells <= 16 ? 16 : 32;
	void *stop = nullptr;
	rc = bsa_ctx_time_begin_internal(ctx, 0.0, &stop);
	if(rc != BSA_OK) return rc;
	auto fail = [&](int code){ (void)bsa_ctx_time_end_internal(ctx, stop); return code; };
	size_t k0 = 0;
	while(k0 < nprogs){
		// as many programs side by side as the row budget holds (at least one)
		size_t k1 = k0, rowsn = 0;
		std::vector<uint64_t> base;
		while(k1 < nprogs){
			const size_t need = (size_t)progs[k1].nnodes;
			if(k1 > k0 && (rowsn + need) * ((size_t)bw * 8 + 4) > budget) break;
			base.push_back(rowsn); rowsn += need; k1++;
		}
		const size_t o_rows = 0, o_u0 = (rowsn * (size_t)bw * 8 + 255) & ~(size_t)255, o_base = (o_u0 + rowsn * 4 + 255) & ~(size_t)255, total = o_base + base.size() * 8 + 256;
		void *ws = nullptr;
		rc = bsa_ctx_scratch_internal(ctx, 0, total, &ws);
		if(rc != BSA_OK) return fail(rc);
		if(hipMemcpyAsync((uint8_t*)ws + o_base, base.data(), base.size() * 8, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail(BSA_E_HIP);
		a.rows = (int2*)((uint8_t*)ws + o_rows); a.u0 = (int32_t*)((uint8_t*)ws + o_u0); a.rowbase = (const uint64_t*)((uint8_t*)ws + o_base);
		a.first_prog = (uint32_t)k0;
		const dim3 grid((uint32_t)(k1 - k0)), block(GEN_NT);
#define GEN_LAUNCH(PWV, CV) hipLaunchKernelGGL((k_poa_gen<PWV, CV>), grid, block, 0, st, a)
#define GEN_LAUNCH_C(PWV) do { switch(C){ case 1: GEN_LAUNCH(PWV, 1); break; case 2: GEN_LAUNCH(PWV, 2); break; case 4: GEN_LAUNCH(PWV, 4); break; case 8: GEN_LAUNCH(PWV, 8); break; \
			case 16: GEN_LAUNCH(PWV, 16); break; default: GEN_LAUNCH(PWV, 32); break; } } while(0)
		if(pw == 0) GEN_LAUNCH_C(0); else if(pw == 1) GEN_LAUNCH_C(1); else GEN_LAUNCH_C(2);
#undef GEN_LAUNCH_C
#undef GEN_LAUNCH
		if(hipGetLastError() != hipSuccess) return fail(BSA_E_HIP);
		k0 = k1;
		if(k0 < nprogs && hipStreamSynchronize(st) != hipSuccess) return fail(BSA_E_HIP);          // (the next group reuses the rows)
	}
	return bsa_ctx_time_end_internal(ctx, stop);
}
