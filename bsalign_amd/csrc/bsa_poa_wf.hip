// bsa_poa_wf.hip -- the POA's per-read seq->graph DP as an anti-diagonal wavefront, with the traceback on the device.
//
// Reference: align_rd_bspoacore bspoa.h:2515-2618 (dpalign_row_update_bspoa :2232-2261 = row_movx bsalign.h:2244 + row_cal
// bsalign.h:2727/2885/3084, dpalign_row_merge_bspoa :2263-2272 = row_merge bsalign.h:2474) and alignment2graph_bspoa
// bspoa.h:2274-2513.  C-ABI: bsa_poa_graph_run / bsa_poa_graph_host (include/bsalign_hip.h), programs built by
// include/bsalign_poa_adapter.h from the reference's own graph.
//
// Why a wavefront: the band offset of every node is fixed before the sweep (prepare_rd_align_bspoa bspoa.h:2168-2174), so the cell
// (node v, column x) depends on cells (u, x - 1) and (u, x) of v's predecessors and on (v, x - 1) -- nothing else.  One wave runs
// one read.  Lane l owns node i (nodes in the order the reference completes them, i mod lanes = l), walks its row cell by cell
// and trails each predecessor by movx + 1 cells; the rows of the nodes in flight are an LDS ring of 4-byte cells
// {int16 H - H(first cell), e, q} with one word per row {node tag, cells written} that tells a reader whether the cell it needs is
// there yet, so lanes never wait for each other: a lane whose inputs are not there skips the step.  The nodes in flight are always 64 consecutive ones (a
// lane takes its next node only when every node before the window is complete), which bounds the ring: a predecessor at most
// NEAR nodes back is read from the ring, anything further from the drained rows in HBM.  Finished rows leave the ring in
// batches, as they are, + one int32 per node (H of the first cell), coalesced -- the only HBM traffic of the forward pass and
// exactly what the traceback reads.
//
// Arithmetic: absolute int32 scores.  The reference keeps int8 differences (u = H(p) - H(p-1), e = E - H, q = Q - H) and block
// start scores; inside bsa_poa_graph_supported()'s guard none of its saturating operations clamps, so the two are the same
// numbers (the test suite holds a scalar statement of this kernel, checked byte for byte against the lane-exact
// restatement of the reference's rows and against the reference itself).  The rules that are not plain affine-gap DP are kept
// literally: the seed of band cell 0 (bsalign.h:2899-2907, rh as bspoa.h:2242-2254 picks it), F and G restarting from
// "predecessor's H - 63" at every running block (bsalign.h:2909-2931, 2639-2652), the synthetic cells behind a moved row's end
// (bsalign.h:2347-2391), the dead row of a move by >= bandwidth (:2253-2259), S = -63 beyond the read end (:2157-2160).
#include "bsa_common.h"
#include <algorithm>
#include <type_traits>
#include <vector>
#include <cstring>
#include <cstdlib>

#define POA_NEAR   12      // predecessors at most this many nodes back are read from the LDS ring
#define POA_DRAIN  8       // finished rows leave the ring in batches of this many
#define POA_NEG    (2 * BSA_SCORE_MIN)
#define POA_NQ     192     // node records staged in LDS ahead of the window
#define POA_TN     32      // traceback: nodes in the ring (a power of two >= POA_TNEAR + 1 + the nodes per refill: 16, or 8 above 128 columns)
#define POA_TILE   16      // ... nodes of a tile of decisions (four columns each: one lane per (node, column))
#define POA_QW     128     // ... columns of the read kept (a power of two)
#define POA_TW     32      // ... cells of a row kept per node (a window around the walk's path; a power of two, a multiple of 4)
#define POA_TNEAR  7       // ... predecessors at most this many nodes back are kept in the ring (0.3 % are further: read from HBM)
#define POA_TE     64      // ... in-edges in the ring (a power of two), refilled half a ring at a time

#define POA_TILE_BYTES (POA_TN * POA_TW * 4 + 256 + POA_TN * 16 + POA_TE * 16)      // the traceback's ring: rows, ubegs[0] + window starts, record heads, in-edges
struct PoaArgs {
	const bsa_poa_node_t *nodes; const bsa_poa_edge_t *edges; const bsa_poa_cand_t *cands; const bsa_poa_prog_t *progs;
	const uint8_t *queries;
	uint32_t *rows; int32_t *u0;
	bsa_poa_result_t *res; uint32_t *steps;          // per program a scratch region of event_cap step words (node << 3 | bt)
	uint32_t *packed; unsigned long long *packed_used;   // all programs' steps back to back, in the order the programs finish
	uint32_t bw, W, nl, R, ri_off, qn_off, nq_off;
	int32_t mode, M, X, refbonus, O, E, Q, P, T;
	int32_t c0, d, head_u0, xp;
	int32_t win_shift;                                // test knob (BSA_POA_WIN_SHIFT): the traceback's row windows that many cells off their place, so that its fall-backs run
};

struct PoaNodeHead { uint32_t rpos, first_in, n_in, base, flags; };
struct PoaTileNode {              // a bsa_poa_node_t as three uint4 in LDS
	uint4 r0, r1, r2;
	__device__ __forceinline__ uint32_t flags_word() const { return r0.w; }
	__device__ __forceinline__ PoaNodeHead head() const { PoaNodeHead h; h.rpos = r0.x; h.first_in = r0.z; h.n_in = r0.w & 0xFFFFu; h.base = (r0.w >> 16) & 0xFFu; h.flags = r0.w >> 24; return h; }
};
__device__ __forceinline__ uint32_t poa_tag(int node){ return ((uint32_t)node & 0x7FFFu) + 1u; }     // never 0 (a cleared cell), distinct within any 32768 consecutive nodes
__device__ __forceinline__ int sx8(uint32_t v){ return (int)(int8_t)(v & 0xFFu); }

template<int PW>
__device__ __forceinline__ int poa_init_h(const PoaArgs &a, int p){     // row_init (bsalign.h:2094-2140) as absolute scores
	if((a.mode & 3) == BSA_MODE_OVERLAP) return 0;
	if(PW == 2){
		const int n1 = min(p, a.xp - 1);
		return a.O + a.E + n1 * a.E + (p - n1) * a.P;
	}
	return a.O + a.E + p * a.E;
}

// ---- the forward pass, a row at a time (the default) ----------------------------------------------------------------------------
// Nodes in the reference's completion order, ONE node per trip of the wave: lane l holds cells l, l + 64, ... of the node's row, the
// node record is uniform (scalar registers: one or two inputs, update or merge, far or near are real branches that cost nothing),
// predecessor rows come from an LDS ring of the last 64 rows (older ones from the rows already in HBM), and the two serial chains of a
// row -- F and G, the horizontal gap states -- are max-plus prefix scans over the wave (DPP row shifts and broadcasts):
//     H(p) = max(N(p), F(p), G(p)),   N = what the inputs offer (diagonal, E, Q, merged rows)
//     F(p) = max(inj(p), max_{j<p} (max(inj(j), N(j) + gapo1) + (p - j) gape1)),   G the same with gapo2 / gape2
// inj(p) = "H of the previous row - 63" at the start of every running block of the reference's striping (bsalign.h:2909-2931).  This is
// the literal recurrence F(p) = max(F(p-1) + gape1, H(p-1) + gapo1 + gape1) with every path through an H that itself came from F or G
// dropped: such a path is never better than staying in the chain it came from (gapo <= 0, and gapo1 + gape1 <= gape1 <= gape2,
// bsa_poa_rows_supported), so the maxima -- the only thing stored -- are the same numbers.  Same rows in HBM as the wavefront below.
// inclusive prefix maximum over the 64 lanes of two values at once: six DPP steps each (row shifts by 1, 2, 4, 8, then the row
// broadcasts), the DPP operand folded into v_max_i32.  The two chains alternate, which leaves one wait state of the two a DPP read of
// a just-written register needs to an s_nop (inline assembly is not seen by the compiler's hazard recogniser).
static __device__ __forceinline__ void poa_scan_max2(int &f, int &g){
	asm volatile(
		"s_nop 1\n\t"
		"v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
		"v_max_i32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
		"s_nop 0\n\t"
		"v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
		"v_max_i32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
		"s_nop 0\n\t"
		"v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
		"v_max_i32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
		"s_nop 0\n\t"
		"v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
		"v_max_i32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
		"s_nop 0\n\t"
		"v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
		"v_max_i32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
		"s_nop 0\n\t"
		"v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
		"v_max_i32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
		"s_nop 1"
		: "+v"(f), "+v"(g));
}

#ifdef POA_PROF
// section timers of the row-at-a-time forward pass (make EXTRA=-DPOA_PROF): every mark waits for everything outstanding and adds the
// shader clocks since the previous mark to the section that ends there
#define POA_PROF_MARK(k_) { __builtin_amdgcn_s_waitcnt(0xC07F); const long long t_ = clock64(); prof_acc[k_] += t_ - prof_t; prof_t = t_; }
#else
#define POA_PROF_MARK(k_)
#endif
#define POA_ROWS_PAD 8        // ring rows are (cells of a wave) + 8 cells apart: a lane whose read would pass the end of a row holds synthetic cells only, its base is clamped

template<int PW, int CPL>
static __device__ __forceinline__ int poa_forward_rows(const PoaArgs &a, const bsa_poa_prog_t &pg, uint8_t *lds, const int lane){
	uint32_t *ring = (uint32_t*)lds;                    // R rows, RS cells apart, of bw cells {int16 H - base, e, q}
	uint8_t *qb = lds + a.nq_off;                       // the read as a profile: bit b = "base b matches", bit 4 = differs from the next base, bit 5 = beyond the end, bits 6-7 = the base
	const int bw = (int)a.bw, W = (int)a.W, RS = CPL * 64 + 2 * POA_ROWS_PAD, RM = (int)a.R - 1, BC = bw + POA_ROWS_PAD;      // BC: the cell of a ring row that holds its base
	const int nn = (int)pg.nnodes, slen = (int)pg.slen;
	const bsa_poa_node_t *nodes = a.nodes + pg.first_node;
	uint32_t *grows = a.rows + (size_t)pg.first_node * bw;
	int32_t *gu0 = a.u0 + pg.first_node;
	const int mode = a.mode & 3;
	const int E = a.E, O = a.O, OE = a.O + a.E, P = a.P, Q = a.Q, QP = a.Q + a.P;
	const int NEG = POA_NEG;
	const int p0 = CPL * lane;                          // the lane's cells: p0 .. p0 + CPL - 1
	int pE[CPL], pP[CPL]; bool blk0[CPL], live[CPL];
#pragma unroll
	for(int j = 0; j < CPL; j++){ const int p = p0 + j; pE[j] = p * E; pP[j] = p * P; blk0[j] = (p % W) == 0; live[j] = p < bw; }
	const int h0init = poa_init_h<PW>(a, 0);
#ifdef POA_PROF
	long long prof_acc[5] = {0, 0, 0, 0, 0}, prof_t = clock64();
#endif
	{
		const uint8_t *q = a.queries + pg.query_off;
		for(int x = lane; x < slen + CPL * 64 + 8; x += 64){
			uint32_t v = 0x20u;
			if(x < slen){ const uint32_t c = q[x] & 3u; v = (1u << c) | (c << 6); if(x + 1 < slen && q[x + 1] != q[x]) v |= 0x10u; }      // (bits 6-7: the base itself, for the traceback)
			qb[x] = (uint8_t)v;
		}
	}
	const int synk = a.c0 + ((lane < a.d) ? lane * E : (a.d - 1) * E + (lane - a.d + 1) * P);      // synthetic cell `lane` behind a row's end, from the row's last H
	// the head: row_init, in ring row 0 already (the caller wrote it at the wavefront's stride bw; RS >= bw and row 0 starts at 0); its copy in HBM
	for(int p = lane; p < bw; p += 64) grows[p] = ring[p];
	if(lane < POA_ROWS_PAD) ring[bw + lane] = (uint32_t)((poa_init_h<PW>(a, bw - 1) + synk - h0init) & 0xFFFF);
	if(lane == 0){ gu0[0] = a.head_u0; ring[BC] = (uint32_t)h0init; }
	__syncthreads();
	for(int i0 = 0; i0 < nn; i0 += 64){
		// 64 node records, one per lane; the fields of node i0 + k come out with v_readlane
		uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0, r2 = r0;
		if(i0 + lane < nn){ const uint4 *rec = (const uint4*)(nodes + i0 + lane); r0 = rec[0]; r1 = rec[1]; r2 = rec[2]; }
		// the records are waited for HERE: a wait inside the node loop would count the rows stored since (vmcnt is one in-order counter
		// for loads and stores) and make every node wait for the previous node's stores to be acknowledged
		__builtin_amdgcn_s_waitcnt(0x0F70);
		// what a node's scalar code would work out of its record, worked out here by the lane that holds the record (1 / 64 of the cost):
		// per input the byte offset of the predecessor's ring row | "predecessor is the head" << 15 | move << 16, the kind word with bit 28 =
		// "further back than the ring", and the diagonal score left of band cell 0 when the band did not move (bspoa.h:2242-2249)
		uint32_t dA[2], dK[2]; int dR[2];
#pragma unroll
		for(int kk = 0; kk < 2; kk++){
			const uint32_t kind = (kk == 0) ? r1.z : r2.y, src = (kk == 0) ? r1.x : r1.w, mv = (kk == 0) ? r1.y : r2.x;
			const bool present = (kind & BSA_POA_IN_PRESENT) != 0u, merge = (kind & BSA_POA_IN_MERGE) != 0u;
			const uint32_t srcE = present ? src : 0u, mvE = (present && !merge) ? min(mv, 0xFFFFu) : 0u;
			dA[kk] = ((srcE & (uint32_t)RM) * (uint32_t)RS * 4u) | ((present && src == 0u) ? 0x8000u : 0u) | (mvE << 16);
			dK[kk] = (kind & ~0x10000000u) | ((present && (i0 + lane) - (int)src > RM) ? 0x10000000u : 0u);
			const int toff = (int)(kind & 0x0FFFFFFFu);
			dR[kk] = r0.x ? BSA_SCORE_MIN : (mode == BSA_MODE_OVERLAP || toff == 0) ? 0 : (PW < 2) ? O + E * toff : max(O + E * toff, Q + P * toff);
		}
		asm volatile("" : "+v"(r0.x), "+v"(r0.w), "+v"(dA[0]), "+v"(dA[1]), "+v"(dK[0]), "+v"(dK[1]), "+v"(dR[0]), "+v"(dR[1]), "+v"(r1.x), "+v"(r1.w));
		const int kend = min(64, nn - i0);
		for(int k = (i0 == 0) ? 1 : 0; k < kend; k++){
			const int i = i0 + k;
			POA_PROF_MARK(0)
			const int rpos = (int)__builtin_amdgcn_readlane((int)r0.x, k);
			const uint32_t w3 = (uint32_t)__builtin_amdgcn_readlane((int)r0.w, k);
			const uint32_t nbase = (w3 >> 16) & 0xFFu;
			const int Mv = a.M + (((w3 >> 24) & 1u) ? a.refbonus : 0);
			// The node's kind decides how much of the general row is needed.  Most nodes are PLAIN: one or two inputs, each a row update (not
			// a merge) of a row in the ring moved by at most the pad, the whole band inside the read -- for them the body below is compiled
			// again (once for one input, once for two) with everything else cut out (no merged rows, no synthetic cells, no clamps, no
			// input that may be absent), the same expressions otherwise.
			uint32_t kinds[2];
			kinds[0] = (uint32_t)__builtin_amdgcn_readlane((int)dK[0], k); kinds[1] = (uint32_t)__builtin_amdgcn_readlane((int)dK[1], k);
			const uint32_t dA0 = (uint32_t)__builtin_amdgcn_readlane((int)dA[0], k);
			const uint32_t kmask = BSA_POA_IN_PRESENT | BSA_POA_IN_MERGE | 0x10000000u;
			int plain = 0;
			if((kinds[0] & kmask) == BSA_POA_IN_PRESENT && (dA0 >> 16) <= (uint32_t)POA_ROWS_PAD && CPL * 64 == bw && rpos + CPL * 64 <= slen){
				if(!(kinds[1] & BSA_POA_IN_PRESENT)) plain = 1;
				else if((kinds[1] & kmask) == BSA_POA_IN_PRESENT && ((uint32_t)__builtin_amdgcn_readlane((int)dA[1], k) >> 16) <= (uint32_t)POA_ROWS_PAD) plain = 2;
			}
			auto node_body = [&](auto plain_tag){
				constexpr int NINP = decltype(plain_tag)::value;
				constexpr bool SIMPLE = NINP > 0;
				// phase 1: everything the node reads from LDS, requested at once (one round trip per node); an input further back than the
				// ring (rare) is read from HBM into the same registers afterwards
				int mvs[2], sbs[2]; uint32_t dAs[2], cwv[2][CPL], cmv[2];
				uint32_t qv[CPL];
#pragma unroll
				for(int j = 0; j < CPL; j++) qv[j] = qb[rpos + p0 + j];
#pragma unroll
				for(int kk = 0; kk < 2; kk++){
					dAs[kk] = 0; mvs[kk] = 0; sbs[kk] = 0; cmv[kk] = 0;
#pragma unroll
					for(int j = 0; j < CPL; j++) cwv[kk][j] = 0;
					if(kk == 1 && (SIMPLE ? NINP < 2 : !(kinds[1] & BSA_POA_IN_PRESENT))) continue;          // (most nodes have one input: nothing to fetch for the other)
					dAs[kk] = (kk == 0) ? dA0 : (uint32_t)__builtin_amdgcn_readlane((int)dA[kk], k);
					mvs[kk] = (int)(dAs[kk] >> 16);
					const uint32_t *lrow = (const uint32_t*)((const uint8_t*)ring + (dAs[kk] & 0x7FFFu));
					sbs[kk] = (int)lrow[BC];
					const int bi = SIMPLE ? p0 + mvs[kk] : min(p0 + mvs[kk], RS - CPL);          // (a lane whose base is clamped holds synthetic cells only; a move within the pad never gets there)
#pragma unroll
					for(int j = 0; j < CPL; j++) cwv[kk][j] = lrow[bi + j];
					cmv[kk] = lrow[max(bi - 1, 0)];
				}
#pragma unroll
				for(int kk = 0; kk < (SIMPLE ? 0 : 2); kk++){
					if(kinds[kk] & 0x10000000u){
						// further back than the ring: the rows stored so far have landed, and nothing stale is in this CU's vector cache
						const int src = __builtin_amdgcn_readlane((int)(kk == 0 ? r1.x : r1.w), k);
						__builtin_amdgcn_s_waitcnt(0);
						__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
						const uint32_t *grow = grows + (size_t)src * bw;
						sbs[kk] = (src == 0) ? h0init : gu0[src];
#pragma unroll
						for(int j = 0; j < CPL; j++) cwv[kk][j] = grow[min(p0 + j + mvs[kk], bw - 1)];
						cmv[kk] = grow[min(max(p0 + mvs[kk] - 1, 0), bw - 1)];
						__builtin_amdgcn_s_waitcnt(0x0F70);          // (waited for here, so that the common path never waits on the vector-memory counter)
					}
				}
				POA_PROF_MARK(1)
				// the substitution scores of the node's base along the lane's cells (bspoa.h:2199-2215; bsalign.h:2166-2221)
				int Sb[CPL], hpc[CPL];
#pragma unroll
				for(int j = 0; j < CPL; j++){
					Sb[j] = ((qv[j] >> nbase) & 1u) ? Mv : a.X;
					hpc[j] = (int)((qv[j] >> 4) & 1u);
				}
				if(!SIMPLE && rpos + CPL * 64 > slen){
#pragma unroll
					for(int j = 0; j < CPL; j++) if(qv[j] & 0x20u) Sb[j] = BSA_EPI8_MIN;
				}
				int N[CPL], Ein[CPL], Qin[CPL], inj[CPL], HX[CPL], EX[CPL], QX[CPL];
#pragma unroll
				for(int j = 0; j < CPL; j++){ N[j] = NEG; Ein[j] = NEG; Qin[j] = NEG; inj[j] = NEG; HX[j] = NEG; EX[j] = NEG; QX[j] = NEG; }
				bool has_merge = false;
				// phase 2: what the inputs offer (uniform branches on the node record, selects per cell)
#pragma unroll
				for(int kk = 0; kk < (SIMPLE ? NINP : 2); kk++){
					const uint32_t kind = kinds[kk];
					if(!SIMPLE && !(kind & BSA_POA_IN_PRESENT)) continue;
					const int mv = mvs[kk], sbase = sbs[kk];
					const bool far = !SIMPLE && (kind & 0x10000000u) != 0u, src0 = (dAs[kk] & 0x8000u) != 0u;
					if(!SIMPLE && (kind & BSA_POA_IN_MERGE)){
						has_merge = true;
#pragma unroll
						for(int j = 0; j < CPL; j++){
							const uint32_t cw = cwv[kk][j];
							const int h = sbase + (int)(int16_t)(cw & 0xFFFFu);
							HX[j] = max(HX[j], h); EX[j] = max(EX[j], h + sx8(cw >> 16)); QX[j] = max(QX[j], h + sx8(cw >> 24));
						}
						continue;
					}
					const bool same = (kind & BSA_POA_IN_SAME) != 0u;
					const bool dead = mv >= bw;
					int h1[CPL], b0[CPL], ee[CPL], qq[CPL];
#pragma unroll
					for(int j = 0; j < CPL; j++){
						const uint32_t cw = cwv[kk][j];
						h1[j] = sbase + (int)(int16_t)(cw & 0xFFFFu);
						ee[j] = sx8(cw >> 16); qq[j] = sx8(cw >> 24);
						b0[j] = (j == 0) ? sbase + (int)(int16_t)(cmv[kk] & 0xFFFFu) : h1[j - 1];
					}
					if(!SIMPLE && (mv > POA_ROWS_PAD || (far && mv > 0))){
						// synthetic cells behind the moved row's end (bsalign.h:2357-2389): c0, then gape1 up to distance d, then gape2.  (A row in
						// the ring carries its first POA_ROWS_PAD synthetic cells behind its end, so a move by up to that many cells -- all but
						// 0.04 % -- reads them like any other cell and only longer moves and rows read back from HBM come here.)
						uint32_t hl_;
						if(far){ const int src = __builtin_amdgcn_readlane((int)(kk == 0 ? r1.x : r1.w), k); hl_ = grows[(size_t)src * bw + bw - 1]; __builtin_amdgcn_s_waitcnt(0x0F70); }
						else hl_ = *(const uint32_t*)((const uint8_t*)ring + (dAs[kk] & 0x7FFFu) + (bw - 1) * 4);
						const int hlast = sbase + (int)(int16_t)(hl_ & 0xFFFFu) + a.c0;
						auto synth = [&](int kx) -> int { return hlast + ((kx < a.d) ? kx * E : (a.d - 1) * E + (kx - a.d + 1) * P); };
#pragma unroll
						for(int j = 0; j < CPL; j++){
							const int idx = p0 + j + mv;
							if(idx >= bw){ h1[j] = synth(idx - bw); ee[j] = 0; qq[j] = 0; }
							if(idx - 1 >= bw) b0[j] = synth(idx - 1 - bw);
						}
						if(dead){
#pragma unroll
							for(int j = 0; j < CPL; j++){ h1[j] = BSA_SCORE_MIN; b0[j] = BSA_SCORE_MIN; ee[j] = 0; qq[j] = 0; }
						}
					} else if(mv == 0 && lane == 0) b0[0] = src0 ? a.head_u0 : sbase;       // ubegs[0] of the predecessor
#pragma unroll
					for(int j = 0; j < CPL; j++){
						const int S = Sb[j] + (same ? 0 : hpc[j]);
						int mc = b0[j] + S;
						if(j == 0){
							// band cell 0: the seed rule (bsalign.h:2899-2907), rh as dpalign_row_update_bspoa picks it (bspoa.h:2242-2254)
							const int rh = (mv == 0) ? __builtin_amdgcn_readlane(dR[kk], k) : b0[0];
							int h0 = rh - b0[0] + S;
							const int tt = (h1[0] - b0[0]) + (PW == 0 ? E : PW == 1 ? ee[0] : max(ee[0], qq[0]));
							h0 = (h0 >= tt) ? min(h0, BSA_EPI8_MAX) : BSA_EPI8_MIN;
							if(lane == 0) mc = b0[0] + h0;
						}
						if(SIMPLE && kk == 0){
							N[j] = mc; inj[j] = blk0[j] ? b0[j] + BSA_EPI8_MIN : NEG; Ein[j] = h1[j] + (PW == 0 ? E : ee[j]);
							if(PW == 2) Qin[j] = h1[j] + qq[j];
						} else {
							N[j] = max(N[j], mc);
							inj[j] = max(inj[j], blk0[j] ? b0[j] + BSA_EPI8_MIN : NEG);
							Ein[j] = max(Ein[j], h1[j] + (PW == 0 ? E : ee[j]));
							if(PW == 2) Qin[j] = max(Qin[j], h1[j] + qq[j]);
						}
					}
				}
				POA_PROF_MARK(2)
				// the chains: per lane the maximum of its cells' sources, one scan over the lanes, then cell by cell inside the lane
				int Nc[CPL], af[CPL], ag[CPL];
				int mf = NEG, mg = NEG;
#pragma unroll
				for(int j = 0; j < CPL; j++){
					Nc[j] = max(N[j], Ein[j]);
					if(!SIMPLE && has_merge) Nc[j] = max(Nc[j], HX[j]);
					if(PW == 2) Nc[j] = max(Nc[j], Qin[j]);
					if(!SIMPLE && CPL * 64 != bw && !live[j]){ Nc[j] = NEG; inj[j] = NEG; }
					af[j] = max(inj[j], Nc[j] + O) - pE[j]; mf = max(mf, af[j]);
					if(PW == 2){ ag[j] = max(inj[j], Nc[j] + Q) - pP[j]; mg = max(mg, ag[j]); }
				}
				poa_scan_max2(mf, mg);
				int exf = __builtin_amdgcn_update_dpp(NEG, mf, 0x138, 0xf, 0xf, false);       // wave_shr:1: what the lanes before offer
				int exg = (PW == 2) ? __builtin_amdgcn_update_dpp(NEG, mg, 0x138, 0xf, 0xf, false) : NEG;
				int H[CPL];
#pragma unroll
				for(int j = 0; j < CPL; j++){
					H[j] = max(Nc[j], max(exf + pE[j], inj[j]));
					if(PW == 2) H[j] = max(H[j], exg + pP[j]);
					exf = max(exf, af[j]);
					if(PW == 2) exg = max(exg, ag[j]);
				}
				POA_PROF_MARK(3)
				const int hb = __builtin_amdgcn_readlane(H[0], 0);
				uint32_t cwo[CPL];
#pragma unroll
				for(int j = 0; j < CPL; j++){
					int e1 = 0, q1 = 0;
					// e = max(E-path of the inputs + gape1, H + gapo1 + gape1, merged rows' E) - H, taken relative to H from the start
					if(PW >= 1){ e1 = max(Ein[j] - H[j] + E, OE); if(!SIMPLE && has_merge) e1 = max(e1, EX[j] - H[j]); }
					if(PW == 2){ q1 = max(Qin[j] - H[j] + P, QP); if(!SIMPLE && has_merge) q1 = max(q1, QX[j] - H[j]); }
					// {int16 H - base, e, q}: the two low bytes of e and q side by side with one v_perm, then under the 16 bits of H
					cwo[j] = (((uint32_t)(H[j] - hb)) & 0xFFFFu) | __builtin_amdgcn_perm((uint32_t)q1, (uint32_t)e1, 0x04000c0cu);
				}
				{
					uint32_t *lrow = ring + (i & RM) * RS + p0;
					uint32_t *grow = grows + (size_t)i * bw + p0;
					if(SIMPLE || CPL * 64 == bw){
						if constexpr(CPL == 2){ *(uint2*)lrow = make_uint2(cwo[0], cwo[1]); *(uint2*)grow = make_uint2(cwo[0], cwo[1]); }
						else if constexpr(CPL == 4){ *(uint4*)lrow = make_uint4(cwo[0], cwo[1], cwo[2], cwo[3]); *(uint4*)grow = make_uint4(cwo[0], cwo[1], cwo[2], cwo[3]); }
						else { lrow[0] = cwo[0]; grow[0] = cwo[0]; }
					} else {
#pragma unroll
						for(int j = 0; j < CPL; j++) if(live[j]){ lrow[j] = cwo[j]; grow[j] = cwo[j]; }
					}
				}
				{
					// the row's first synthetic cells (e = q = 0) behind its end, ring only
					const int hl = __builtin_amdgcn_readlane(H[(bw - 1) % CPL], (bw - 1) / CPL);
					if(lane < POA_ROWS_PAD) ring[(i & RM) * RS + bw + lane] = (uint32_t)((hl + synk - hb) & 0xFFFF);
				}
				if(lane == 0){ ring[(i & RM) * RS + BC] = (uint32_t)hb; gu0[i] = hb; }
				POA_PROF_MARK(4)
			};
			if(plain == 1) node_body(std::integral_constant<int, 1>{}); else if(plain == 2) node_body(std::integral_constant<int, 2>{}); else node_body(std::integral_constant<int, 0>{});
		}
	}
#ifdef POA_PROF
	if(lane == 0 && blockIdx.x == 0) printf("poa_forward_rows profile: %d nodes; clocks per node: setup+issue %.0f, wait+far %.0f, inputs %.0f, chains %.0f, store %.0f\n", nn,
		(double)prof_acc[1] / nn, (double)prof_acc[2] / nn, (double)prof_acc[3] / nn, (double)prof_acc[4] / nn, (double)prof_acc[0] / nn);
#endif
	return nn;
}

template<int PW, int ROWS>                  // ROWS: 0 = the wavefront forward pass, else the row-at-a-time one with ROWS cells per lane
__global__ void __launch_bounds__(64) k_poa_wf(const PoaArgs a){
	extern __shared__ __align__(16) uint8_t lds[];
	uint32_t *ring = (uint32_t*)lds;                // R rows of bw cells {int16 H - base, e, q}: exactly the cells the traceback reads from HBM
	uint2 *rinfo = (uint2*)(lds + a.ri_off);        // per ring row: {base = H of its first cell, tag << 16 | cells written}
	uint32_t *qn = (uint32_t*)(lds + a.qn_off);
	uint4 *nq = (uint4*)(lds + a.nq_off);           // node records of the next POA_NQ nodes, record i at slot i % POA_NQ (three uint4 each)
	const bsa_poa_prog_t pg = a.progs[blockIdx.x];
	const int lane = threadIdx.x;
	const int bw = (int)a.bw, W = (int)a.W, NL = (int)a.nl, R = (int)a.R;
	const int nn = (int)pg.nnodes, slen = (int)pg.slen;
	const bsa_poa_node_t *nodes = a.nodes + pg.first_node;
	uint32_t *grows = a.rows + (size_t)pg.first_node * bw;
	int32_t *gu0 = a.u0 + pg.first_node;
	const uint64_t lmask = (NL == 64) ? ~0ull : ((1ull << NL) - 1ull);
	const int mode = a.mode & 3;
	const int E = a.E, OE = a.O + a.E, P = a.P, QP = a.Q + a.P;
	if(nn == 0){ if(lane == 0){ bsa_poa_result_t r; r.maxscr = BSA_SCORE_MIN; r.maxidx = -1; r.maxoff = -1; r.status = BSA_POA_ST_NOCAND; r.nevents = 0; r.fin_node = -1; r.fin_x = -1; r.reserved = 0; a.res[blockIdx.x] = r; } return; }

	// ring tags cleared, query as nibbles (code | differs-from-next << 2 | beyond-the-read << 3), head row in slot 0
	for(int i = lane; i < R; i += 64) rinfo[i] = make_uint2(0u, 0u);
	if constexpr(ROWS == 0){
		const uint8_t *q = a.queries + pg.query_off;
		const int nqw = (slen + bw + 16) / 8 + 1;
		for(int w = lane; w < nqw; w += 64){
			uint32_t v = 0;
			for(int k = 0; k < 8; k++){
				const int x = w * 8 + k; uint32_t nb;
				if(x >= slen) nb = 8u;
				else { const uint32_t c = q[x]; nb = c & 3u; if(x + 1 < slen && q[x + 1] != c) nb |= 4u; }
				v |= nb << (4 * k);
			}
			qn[w] = v;
		}
	}
	__syncthreads();
	{
		const uint32_t eq = ((PW >= 1) ? 0xC10000u : 0u) | ((PW == 2) ? 0xC1000000u : 0u);      // e = q = -63
		const int h0 = poa_init_h<PW>(a, 0);
		for(int p = lane; p < bw; p += 64) ring[p] = (uint32_t)((poa_init_h<PW>(a, p) - h0) & 0xFFFF) | eq;
		if(lane == 0) rinfo[0] = make_uint2((uint32_t)h0, (poa_tag(0) << 16) | (uint32_t)bw);
	}
	__syncthreads();

	// ---- forward pass ----
	const unsigned long long tick0 = wall_clock64(), clk0 = clock64();
	int iters = 0;
	if constexpr(ROWS != 0){
		iters = poa_forward_rows<PW, ROWS>(a, pg, lds, lane);
	} else {
	int m = -NL;                 // nodes below m are complete; the nodes in flight are m .. m + NL - 1
	int mr = 0;                  // m mod NL
	int myslot = lane - NL;      // cur mod R, kept incrementally (no divisions in the loop)
	while(myslot < 0) myslot += R;
	int drslot = 0;              // dr mod R
	int dr = 0;                  // rows below dr are in HBM
	int cur = lane - NL, p = 0;
	bool fin = true;             // finished `cur`, waiting for the window to take the next node
	int rposv = 0, Mv = 0, F = POA_NEG, G = POA_NEG, blk = 0;
	uint32_t basev = 0, mytag = 0; int myrow = 0, mybase = 0;
	uint32_t qw = 0;
	uint32_t fl0 = 0, fl1 = 0;   // input flags: 1 present, 2 merge, 4 same base, 8 far (HBM), 16 dead (moved by >= bandwidth)
	int ad0 = 0, ad1 = 0, lim0 = 0, lim1 = 0, hp0 = 0, hp1 = 0, src0 = 0, src1 = 0, mv0 = 0, mv1 = 0, to0 = 0, to1 = 0;
	uint32_t tg0 = 0, tg1 = 0; int fu0 = 0, fu1 = 0;       // far inputs: the predecessor's ubegs[0]
	int rh00 = 0, rh01 = 0;                                // the diagonal score left of band cell 0 when the band did not move (bspoa.h:2242-2249)

	auto drain = [&](int upto){
		// rows dr .. upto - 1 -> HBM as they are; four cells per lane and store
		for(int r = dr; r < upto; r++){
			const uint32_t *src = ring + drslot * bw;
			for(int c = lane * 4; c < bw; c += 256) *(uint4*)&grows[(size_t)r * bw + c] = *(const uint4*)&src[c];
			if(lane == 0) gu0[r] = (r == 0) ? a.head_u0 : (int)rinfo[drslot].x;
			if(++drslot == R) drslot = 0;
		}
		dr = upto;
	};

	int qfill = 0;               // node records below qfill are (or were) in the LDS queue
	auto refill = [&](int upto){
		// records [qfill, upto): 3 x 16 bytes each, coalesced
		const uint4 *src = (const uint4*)nodes;
		for(int i = qfill * 3 + lane; i < upto * 3; i += 64) nq[i % (POA_NQ * 3)] = src[i];
		qfill = upto;
	};
	refill(min(nn, POA_NQ));
	__syncthreads();
	while(m < nn){
		iters++;
		if(qfill < nn && qfill < m + 2 * NL + 32){ refill(min(nn, qfill + 32)); __syncthreads(); }
		// (A) the window: lanes whose finished node heads the window take their next node
		{
			const uint64_t B = __ballot(fin) & lmask;
			const int r = mr;
			uint64_t rot = B;
			if(r) rot = ((B >> r) | (B << (NL - r))) & lmask;
			const int t = (rot == lmask) ? NL : __builtin_ctzll(~rot);
			int dd = lane - r; if(dd < 0) dd += NL;
			if(lane < NL && fin && dd < t){
				cur += NL; p = 0;
				myslot += NL; if(myslot >= R) myslot -= R;
				if(cur == 0){ fin = true; }                 // the head's row is there already
				else if(cur >= nn){ fin = true; }           // past the end: a virtual node, complete at once
				else {
					bsa_poa_node_t nd;
					{
						const uint4 *rec = nq + (cur % POA_NQ) * 3;
						const uint4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
						nd.rpos = r0.x; nd.gnode = r0.y; nd.first_in = r0.z; nd.n_in = (uint16_t)(r0.w & 0xFFFFu); nd.base = (uint8_t)((r0.w >> 16) & 0xFFu); nd.flags = (uint8_t)(r0.w >> 24);
						nd.in[0].src = r1.x; nd.in[0].movx = r1.y; nd.in[0].toff_kind = r1.z; nd.in[1].src = r1.w; nd.in[1].movx = r2.x; nd.in[1].toff_kind = r2.y;
					}
					fin = false;
					rposv = (int)nd.rpos; basev = nd.base; Mv = a.M + ((nd.flags & 1) ? a.refbonus : 0);
					F = POA_NEG; G = POA_NEG; blk = 0;
					myrow = myslot * bw; mytag = poa_tag(cur) << 16;
					qw = qn[rposv >> 3];
#define POA_SETUP(k, FL, AD, LIM, SRC, MV, TO, TG, RH0)                                                              \
					{                                                                                                      \
						const bsa_poa_input_t in = nd.in[k];                                                               \
						FL = 0; LIM = 0;                                                                                   \
						if(in.toff_kind & BSA_POA_IN_PRESENT){                                                             \
							SRC = (int)in.src; MV = (int)in.movx; TO = (int)(in.toff_kind & BSA_POA_IN_TOFF);              \
							FL = 1u | ((in.toff_kind & BSA_POA_IN_MERGE) ? 2u : 0u) | ((in.toff_kind & BSA_POA_IN_SAME) ? 4u : 0u); \
							if(cur - SRC > POA_NEAR) FL |= 8u;                                                             \
							if(MV >= bw) FL |= 16u;                                                                        \
							LIM = bw - MV; TG = poa_tag(SRC);                                                              \
							{ int sl = myslot - (cur - SRC); if(sl < 0) sl += R; if(FL & 8u) sl = 0; AD = sl * bw + MV; TG = (TG << 16) | (uint32_t)sl; }   \
							if(rposv) RH0 = BSA_SCORE_MIN;                                                                 \
							else if(mode == BSA_MODE_OVERLAP || TO == 0) RH0 = 0;                                          \
							else if(PW < 2) RH0 = a.O + E * TO;                                                            \
							else RH0 = max(a.O + E * TO, a.Q + P * TO);                                                    \
						}                                                                                                  \
					}
					POA_SETUP(0, fl0, ad0, lim0, src0, mv0, to0, tg0, rh00)
					POA_SETUP(1, fl1, ad1, lim1, src1, mv1, to1, tg1, rh01)
#undef POA_SETUP
				}
			}
			m += t; mr += t; if(mr >= NL) mr -= NL;
		}
		// (B) finished rows leave the ring; a lane waiting for a far predecessor forces them out
		{
			const bool farwait = !fin && (((fl0 & 9u) == 9u && src0 >= dr) || ((fl1 & 9u) == 9u && src1 >= dr));
			const int done = min(m, nn);
			if(done - dr >= POA_DRAIN || (__ballot(farwait) != 0ull && done > dr)){
				drain(done);
				if(__ballot(farwait) != 0ull) __builtin_amdgcn_s_waitcnt(0);      // the stores have to land before they are read back
			}
		}
		// (C) one cell for every lane whose inputs are there.  Straight-line code: what a lane is (one or two inputs, update or merge, a
		// moved row's synthetic tail, band cell 0) is decided by selects, not branches -- a branch costs a wave more than the few
		// instructions it would skip.  Only predecessors read back from HBM (further than POA_NEAR nodes away: rare) take a branch.
		if(!fin){
			const bool is0 = (p == 0);
			int h10, e10, q10, hq0, h11, e11, q11, hq1;
			bool ok0, ok1;
#define POA_FETCH(FL, AD, LIM, TG, HPK, H1, E1, Q1, HQ, OK)                                                              \
			{                                                                                                        \
				const bool real = (p < LIM) && !(FL & 8u);                                                           \
				const int adc = real ? AD : 0;                                                                       \
				const uint32_t cw = ring[adc];                                                                       \
				const uint32_t cm = ring[adc - (adc > 0 ? 1 : 0)];                                                   \
				const uint2 ri = rinfo[TG & 0xFFFFu];                                                                \
				const int ov = p - LIM;                                                                              \
				const int inc = (ov == 0) ? a.c0 : ((ov < a.d) ? E : P);                                             \
				/* the cell is there when the row's owner has written past it (and the row is still this node's) */  \
				OK = !real || (ri.y >> 16 == TG >> 16 && (int)(ri.y & 0xFFFFu) > adc - (int)(TG & 0xFFFFu) * bw);     \
				HQ = (FL & 16u) ? BSA_SCORE_MIN : (is0 ? (int)ri.x + (int)(int16_t)(cm & 0xFFFFu) : HPK);            \
				H1 = real ? (int)ri.x + (int)(int16_t)(cw & 0xFFFFu) : ((FL & 16u) ? BSA_SCORE_MIN : HPK + inc);     \
				E1 = real ? sx8(cw >> 16) : 0; Q1 = real ? sx8(cw >> 24) : 0;                                        \
			}
			POA_FETCH(fl0, ad0, lim0, tg0, hp0, h10, e10, q10, hq0, ok0)
			POA_FETCH(fl1, ad1, lim1, tg1, hp1, h11, e11, q11, hq1, ok1)
#undef POA_FETCH
			if(__ballot(((fl0 | fl1) & 8u) != 0u) != 0ull){
				// a predecessor further back than the ring holds: its finished row from HBM (drained first, see (B))
#define POA_FAR(FL, LIM, SRC, MV, FU, H1, E1, Q1, HQ, OK)                                                                \
				if((FL & 9u) == 9u && !(FL & 16u) && p < LIM){                                                       \
					if(SRC < dr){                                                                                    \
						if(is0) FU = *(const volatile int32_t*)&gu0[SRC];                                            \
						const uint32_t cw = *(const volatile uint32_t*)&grows[(size_t)SRC * bw + p + MV];            \
						H1 = (SRC == 0) ? poa_init_h<PW>(a, p + MV) : FU + (int)(int16_t)(cw & 0xFFFFu);             \
						E1 = sx8(cw >> 16); Q1 = sx8(cw >> 24);                                                      \
						if(is0 && MV > 0){                                                                           \
							const uint32_t cp = *(const volatile uint32_t*)&grows[(size_t)SRC * bw + MV - 1];        \
							HQ = (SRC == 0) ? poa_init_h<PW>(a, MV - 1) : FU + (int)(int16_t)(cp & 0xFFFFu);         \
						}                                                                                            \
						OK = true;                                                                                   \
					} else OK = false;                                                                               \
				}
				POA_FAR(fl0, lim0, src0, mv0, fu0, h10, e10, q10, hq0, ok0)
				POA_FAR(fl1, lim1, src1, mv1, fu1, h11, e11, q11, hq1, ok1)
#undef POA_FAR
			}
			const bool ready = ok0 && ok1;
			{
				const int x = rposv + p;
				const uint32_t nb = (qw >> ((x & 7) * 4)) & 0xFu;
				const bool beyond = (nb & 8u) != 0u;
				const int Sb = beyond ? BSA_EPI8_MIN : (((nb & 3u) == basev) ? Mv : a.X);
				const int hpc = beyond ? 0 : (int)((nb >> 2) & 1u);
				const bool bs = is0 || (blk == 0);
				int mm = POA_NEG, Ein = POA_NEG, Qin = POA_NEG, fl = POA_NEG, HX = POA_NEG, EX = POA_NEG, QX = POA_NEG;
#define POA_INPUT(FL, SRC, MV, RH0, H1, E1, Q1, HQ)                                                                      \
				{                                                                                                    \
					const bool upd = (FL & 3u) == 1u, mrg = (FL & 3u) == 3u;                                         \
					const int S = Sb + ((FL & 4u) ? 0 : hpc);                                                        \
					const int ub0 = (FL & 16u) ? BSA_SCORE_MIN : ((MV == 0) ? ((SRC == 0) ? a.head_u0 : H1) : HQ);   \
					const int rh = (MV == 0) ? RH0 : ((MV <= bw) ? ub0 : BSA_SCORE_MIN);                             \
					int h0 = rh - ub0 + S;                                                                           \
					const int t = (H1 - ub0) + (PW == 0 ? E : PW == 1 ? E1 : max(E1, Q1));                           \
					h0 = (h0 >= t) ? min(h0, BSA_EPI8_MAX) : BSA_EPI8_MIN;                                           \
					const int b0 = is0 ? ub0 : HQ;                                                                   \
					const int mc = b0 + (is0 ? h0 : S);                                                              \
					mm = max(mm, upd ? mc : POA_NEG);                                                                \
					fl = max(fl, (upd && bs) ? b0 + BSA_EPI8_MIN : POA_NEG);                                         \
					Ein = max(Ein, upd ? H1 + (PW == 0 ? E : E1) : POA_NEG);                                         \
					if(PW == 2) Qin = max(Qin, upd ? H1 + Q1 : POA_NEG);                                             \
					HX = max(HX, mrg ? H1 : POA_NEG);                                                                \
					if(PW >= 1) EX = max(EX, mrg ? H1 + E1 : POA_NEG);                                               \
					if(PW == 2) QX = max(QX, mrg ? H1 + Q1 : POA_NEG);                                               \
				}
				POA_INPUT(fl0, src0, mv0, rh00, h10, e10, q10, hq0)
				POA_INPUT(fl1, src1, mv1, rh01, h11, e11, q11, hq1)
#undef POA_INPUT
				const int Fc = max(F, fl), Gc = max(G, fl);
				int H = max(max(mm, Ein), max(Fc, HX));
				if(PW == 2) H = max(max(H, Qin), Gc);
				int e1 = 0, q1 = 0, Fn, Gn = Gc;
				if(PW == 0) Fn = H + E;
				else {
					e1 = max(max(Ein + E, H + OE), EX) - H;
					Fn = max(Fc + E, H + OE);
					if(PW == 2){
						q1 = max(max(Qin + P, H + QP), QX) - H;
						Gn = max(Gc + P, H + QP);
					}
				}
				if(ready){
					if(is0) mybase = H;
					ring[myrow + p] = (uint32_t)((H - mybase) & 0xFFFF) | (((uint32_t)e1 & 0xFFu) << 16) | ((uint32_t)q1 << 24);
					rinfo[myslot] = make_uint2((uint32_t)mybase, mytag | (uint32_t)(p + 1));
					F = Fn; G = Gn;
					hp0 = h10; hp1 = h11;
					ad0++; ad1++;
					p++;
					blk = (blk + 1 == W) ? 0 : blk + 1;
					qw = qn[(rposv + p) >> 3];
					fin = (p == bw);
				}
			}
		}
	}
	drain(nn);
	}
	__builtin_amdgcn_s_waitcnt(0);
	__syncthreads();
	const int fwd_ticks = (int)(wall_clock64() - tick0), fwd_clk = (int)((clock64() - clk0) >> 6);

	// ---- the best end cell (bspoa.h:2549-2603): candidates in the reference's visiting order, strictly greater replaces ----
	const bsa_poa_cand_t *cands = a.cands + pg.first_cand;
	long long bkey = (long long)0x8000000000000000ull; int boff = -1;
	auto cellH = [&](int node, int pp) -> int {
		if(node == 0) return poa_init_h<PW>(a, pp);
		const uint32_t cw = *(const volatile uint32_t*)&grows[(size_t)node * bw + pp];
		return *(const volatile int32_t*)&gu0[node] + (int)(int16_t)(cw & 0xFFFFu);
	};
	for(int k = lane; k < (int)pg.ncands; k += 64){
		const int node = (int)cands[k].node, rp = (int)nodes[node].rpos;
		if(cands[k].kind == 1){
			const int s = cellH(node, slen - 1 - rp) + a.T;
			const long long key = ((long long)s << 32) | (long long)(0xFFFFFFFFu - (uint32_t)(2 * k));
			if(key > bkey){ bkey = key; boff = slen - 1; }
		} else {
			const int mo = min(slen, rp + bw) - 1;
			int s = cellH(node, mo - rp);
			if(slen > mo + 1){ const int n = slen - mo - 1; s += (PW < 2) ? a.O + E * n : max(a.O + E * n, a.Q + P * n); }
			s += a.T;
			long long key = ((long long)s << 32) | (long long)(0xFFFFFFFFu - (uint32_t)(2 * k));
			if(key > bkey){ bkey = key; boff = mo; }
			if(mode == BSA_MODE_OVERLAP){
				// row_max (bsalign.h:3213-3329): best cell of every running block, first one on ties; blocks in the order of the register reduction
				int bs = 0, bp = 0;
				for(int kk = 0; kk < 16; kk++){
					const int j = (kk & 3) * 4 + (kk >> 2);
					int mx = cellH(node, j * W), ai = 0;
					for(int i = 1; i < W; i++){ const int v = cellH(node, j * W + i); if(v > mx){ mx = v; ai = i; } }
					if(kk == 0 || mx > bs){ bs = mx; bp = j * W + ai; }
				}
				key = ((long long)bs << 32) | (long long)(0xFFFFFFFFu - (uint32_t)(2 * k + 1));
				if(key > bkey){ bkey = key; boff = bp + rp; }
			}
		}
	}
	for(int o = 32; o > 0; o >>= 1){
		const long long ok = __shfl_xor(bkey, o); const int oo = __shfl_xor(boff, o);
		if(ok > bkey){ bkey = ok; boff = oo; }
	}
	bsa_poa_result_t rs;
	rs.reserved = (int)(wall_clock64() - tick0);        // forward pass + best end cell, in ticks of the 100 MHz counter
	if(a.mode & 0x100) rs.reserved = (ROWS != 0) ? fwd_ticks : iters;
	rs.nevents = 0; rs.fin_node = -1; rs.fin_x = -1; rs.status = BSA_POA_ST_OK;
	if(bkey == (long long)0x8000000000000000ull){
		rs.maxscr = BSA_SCORE_MIN; rs.maxidx = -1; rs.maxoff = -1; rs.status = BSA_POA_ST_NOCAND;
		if(lane == 0) a.res[blockIdx.x] = rs;
		return;
	}
	rs.maxscr = (int)(bkey >> 32);
	rs.maxidx = (int)cands[(0xFFFFFFFFu - (uint32_t)(bkey & 0xFFFFFFFFll)) >> 1].node;
	rs.maxoff = boff;

	if(a.mode & 0x200){ if(a.mode & 0x100){ rs.fin_node = fwd_ticks; rs.fin_x = fwd_clk; } if(lane == 0) a.res[blockIdx.x] = rs; return; }      // (measurement: forward pass and best end cell only)
	// ---- traceback: the walk's state is uniform; the wave decides 64 places at a time (a tile) from a ring of nodes, row windows and in-edges in LDS ----
	{
		const bsa_poa_edge_t *gedges = a.edges + pg.first_edge;
		uint32_t *ev = a.steps + pg.first_event;
		const int ecap = (int)pg.event_cap;
		// The walk's window of the graph: a RING of the last POA_TN nodes at and below the walker (node i at slot i mod POA_TN: record,
		// ubegs[0], and POA_TW cells of its row around the column the walk will pass it at -- cell p at position p mod POA_TW, the window's
		// first cell beside it, so that a reader fetches cell and window start in one round trip and checks afterwards) and of POA_TE
		// in-edges, refilled sixteen nodes / sixty-four edges at a time.  A refill is requested well before the walker needs it and kept
		// in registers until it does, so its memory latency passes while the walk goes on.
		uint32_t *t_rows = (uint32_t*)lds;
		int32_t *t_u0 = (int32_t*)(lds + (size_t)POA_TN * POA_TW * 4);
		int32_t *t_c0 = (int32_t*)(lds + (size_t)POA_TN * POA_TW * 4 + 128);
		uint4 *t_r0 = (uint4*)(lds + (size_t)POA_TN * POA_TW * 4 + 256);           // the traceback's view of a node record: {rpos, gnode, first_in, n_in | base << 16 | flags << 24}
		bsa_poa_edge_t *t_edges = (bsa_poa_edge_t*)(lds + (size_t)POA_TN * POA_TW * 4 + 256 + POA_TN * 16);
		// the read's base at a column as code | differs-from-the-next << 2 | beyond-the-read << 3: the wavefront's nibbles of the whole read,
		// else a window of POA_QW columns that slides down with the walker (column c at position c mod POA_QW; the forward pass's
		// profile of the whole read lies where the ring is now)
		uint8_t *t_q = lds + POA_TILE_BYTES;
		int qlo = 0;
		auto q_fill = [&](int c0_, int cnt_){
			const uint8_t *q = a.queries + pg.query_off;
			for(int c = c0_ + lane; c < c0_ + cnt_; c += 64){
				uint32_t nb = 8u;
				if(c < slen){ const uint32_t b = q[c]; nb = b & 3u; if(c + 1 < slen && q[c + 1] != b) nb |= 4u; }
				t_q[c & (POA_QW - 1)] = (uint8_t)nb;
			}
		};
		auto QCODE = [&](int xx) -> uint32_t {
			if constexpr(ROWS == 0) return (qn[xx >> 3] >> ((xx & 7) * 4)) & 0xFu;
			else return t_q[xx & (POA_QW - 1)];
		};
		int n = rs.maxidx, nidx = rs.maxidx, x = rs.maxoff, ne = 0, status = BSA_POA_ST_OK;
		uint32_t bt = 0xFFFFFFFFu;
		int Hs0 = 0, Hs1 = 0, Hs2 = 0;
		bool done = false, first = true;
		constexpr int RQ = POA_TW / 4;                      // 16-byte pieces of a ring row
		constexpr int TC = 16;                              // nodes per refill
		constexpr int NRQ = TC * RQ / 64;                   // 16-byte pieces of a refill, per lane
		// The window of a node: centred on the cell the walk is expected to pass it at.  The reference places a node's band around the read
		// position it expects there, pinned to the read's two ends (bspoa.h:2168-2174), so the walker's cell runs along
		// cell_of(x) = x - min(max(x - bw / 2, 0), slen - bw) -- down a cell a step near the ends, level in between -- plus whatever the
		// walk has wandered off it so far; the column a node further down is passed at follows from the steps per node the walk has shown.
		auto window_at = [&](int centre) -> int { return min(max((centre + a.win_shift - POA_TW / 2) & ~3, 0), max(bw - POA_TW, 0)); };
		auto cell_of = [&](int xx) -> int { return xx - min(max(xx - bw / 2, 0), max(slen - bw, 0)); };
		int lo = max(0, n - (POA_TN - 1));                  // the ring holds nodes lo .. (the walker never goes up)
		int elo, ehi;                                       // ... and edges elo .. ehi - 1
		{
			__syncthreads();
			if constexpr(ROWS != 0){ qlo = max(0, x + 4 - POA_QW); q_fill(qlo, POA_QW); }
			const int cnt = n - lo + 1;
			const int off0 = x - (int)nodes[n].rpos - cell_of(x);          // (no history yet: a step a node)
			if(lane < cnt) t_r0[(lo + lane) & (POA_TN - 1)] = *(const uint4*)(nodes + lo + lane);
			for(int i = lane; i < cnt * RQ; i += 64){
				const int nd_ = lo + i / RQ, c = window_at(cell_of(x - (n - nd_)) + off0) + (i % RQ) * 4;
				if(c < bw) *(uint4*)(t_rows + (nd_ & (POA_TN - 1)) * POA_TW + (c & (POA_TW - 1))) = *(const uint4*)(grows + (size_t)nd_ * bw + c);
			}
			if(lane < cnt){ t_u0[(lo + lane) & (POA_TN - 1)] = gu0[lo + lane]; t_c0[(lo + lane) & (POA_TN - 1)] = window_at(cell_of(x - (n - lo - lane)) + off0); }
			__syncthreads();
			const uint4 hh = t_r0[n & (POA_TN - 1)];
			ehi = (int)hh.z + (int)(hh.w & 0xFFFFu);
			elo = max(0, ehi - POA_TE);
			for(int i = elo + lane; i < ehi; i += 64) ((uint4*)t_edges)[i & (POA_TE - 1)] = ((const uint4*)gedges)[i];
			__syncthreads();
		}
		bool npend = false, epend = false;                   // a refill of nodes / edges is in flight (in the registers below)
		int plo = 0, pelo = 0, pc0 = 0;
		uint4 prow[NRQ], pnode = make_uint4(0, 0, 0, 0), pedge = make_uint4(0, 0, 0, 0); int pu0 = 0;
#pragma unroll
		for(int k = 0; k < NRQ; k++) prow[k] = make_uint4(0, 0, 0, 0);
		auto node_request = [&](int pp){
			// (every lane loads -- an index past the refill's end reads its last piece again --: a load under a condition would have to be
			// merged into the register it waits in, and the merge waits for it)
			plo = max(0, lo - TC); pc0 = window_at(pp);
			const int cnt = lo - plo;
#pragma unroll
			for(int k = 0; k < NRQ; k++){
				const int i = min(k * 64 + lane, cnt * RQ - 1), c = min(pc0 + (i % RQ) * 4, bw - 4);
				prow[k] = *(const uint4*)(grows + (size_t)(plo + i / RQ) * bw + c);
			}
			pnode = *(const uint4*)(nodes + plo + min(lane, cnt - 1));
			pu0 = gu0[plo + min(lane, cnt - 1)];
			npend = true;
		};
		auto node_commit = [&](){
			const int cnt = lo - plo;
#pragma unroll
			for(int k = 0; k < NRQ; k++){
				const int i = k * 64 + lane, c = pc0 + (i % RQ) * 4;
				if(i < cnt * RQ && c < bw) *(uint4*)(t_rows + ((plo + i / RQ) & (POA_TN - 1)) * POA_TW + (c & (POA_TW - 1))) = prow[k];
			}
			if(lane < cnt){ t_r0[(plo + lane) & (POA_TN - 1)] = pnode; t_u0[(plo + lane) & (POA_TN - 1)] = pu0; t_c0[(plo + lane) & (POA_TN - 1)] = pc0; }
			lo = plo; npend = false;
		};
		auto edge_request = [&](){
			pelo = max(0, elo - POA_TE / 2);
			pedge = ((const uint4*)gedges)[min(pelo + lane, elo - 1)];
			epend = true;
		};
		auto edge_commit = [&](){
			if(pelo + lane < elo) ((uint4*)t_edges)[(pelo + lane) & (POA_TE - 1)] = pedge;
			elo = pelo; ehi = min(ehi, elo + POA_TE); epend = false;
		};
		{
			{
				auto in_tile = [&](int i) -> bool { return i >= lo; };
				auto U0 = [&](int i) -> int { if(in_tile(i)) return t_u0[i & (POA_TN - 1)]; return *(const volatile int32_t*)&gu0[i]; };
				auto CELL = [&](int i, int pp) -> uint32_t {
					if(in_tile(i) && (unsigned)(pp - t_c0[i & (POA_TN - 1)]) < (unsigned)POA_TW) return t_rows[(i & (POA_TN - 1)) * POA_TW + (pp & (POA_TW - 1))];
					return *(const volatile uint32_t*)&grows[(size_t)i * bw + pp];
				};
				auto HH = [&](int i, int pp, uint32_t cw, int u0v) -> int { return (i == 0) ? poa_init_h<PW>(a, pp) : u0v + (int)(int16_t)(cw & 0xFFFFu); };
#define EMIT(nn_, xx_, bb_) do{ if(ne >= ecap){ status = BSA_POA_ST_EVENTS; done = true; } else { if(lane == 0) ev[ne] = ((uint32_t)(nn_) << 3) | (bb_); ne++; } }while(0)
				if(first){
					first = false;
					const int pp = x - (int)nodes[n].rpos;
					if(pp < 0 || pp >= bw){ status = BSA_POA_ST_TRACE; done = true; }
					else Hs1 = HH(n, pp, CELL(n, pp), U0(n));
				}
#ifdef POA_PROF
				long long tq[6] = {0, 0, 0, 0, 0, 0}, tq_t = clock64(); int tq_n = 0, tq_chase = 0, tq_build = 0, tq_slow = 0; const long long tq_0 = clock64(); long long tq_c[6] = {0, 0, 0, 0, 0, 0}, tq_l = clock64(); int d_dbg = 0, d_why = 0, tq_why[10] = {0,0,0,0,0,0,0,0,0,0}; int tq_cat = 5, tq_k[6] = {0, 0, 0, 0, 0, 0}, tq_end[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, tq_endL[12] = {0,0,0,0,0,0,0,0,0,0,0,0};
#define POA_TRK(k_) { __builtin_amdgcn_s_waitcnt(0xC07F); const long long t_ = clock64(); tq[k_] += t_ - tq_t; tq_t = t_; }
#else
#define POA_TRK(k_)
#endif
				// The walk's state is uniform, and kept so for the compiler: every value that comes out of memory or out of one lane goes
				// through v_readfirstlane / v_readlane, so that the loop's tests are scalar compares and branches and the lanes' work below has
				// no divergent branch in it.
#define POA_UNI(v_) __builtin_amdgcn_readfirstlane((int)(v_))
				n = POA_UNI(n); nidx = n; x = POA_UNI(x); Hs1 = POA_UNI(Hs1); ne = 0; done = POA_UNI(done) != 0; status = POA_UNI(status);
				lo = POA_UNI(lo); elo = POA_UNI(elo); ehi = POA_UNI(ehi);
				int wpp = x - (int)nodes[n].rpos, wpx = x;                     // the walker's cell, as of the last node whose record was looked at (then at column wpx)
				wpp = POA_UNI(wpp); wpx = POA_UNI(wpx);
				int h_n = -1; uint32_t h_rpos = 0, h_first = 0, h_w3 = 0;       // the record of node h_n (a step that chooses a node brings its record along)
				const int h0init = poa_init_h<PW>(a, 0);
				const bool ovl = mode == BSA_MODE_OVERLAP;
				auto start_insertion = [&](int nrpos_){
					// no predecessor explains the cell: an insertion run along the node's own row (bspoa.h:2412-2440)
					const int pp = x - nrpos_;
					if(pp < 0 || pp >= bw){ status = BSA_POA_ST_TRACE; done = true; }
					else {
						const int u0v = POA_UNI(U0(n));
						const int hmn = (pp == 0) ? u0v : POA_UNI(HH(n, pp - 1, CELL(n, pp - 1), u0v));
						bt = 1u; Hs2 = 1; Hs0 = Hs1 - (POA_UNI(HH(n, pp, CELL(n, pp), u0v)) - hmn);
					}
				};
				// THE TILE.  What the walk does at (node, column) in its plain state -- which predecessor explains the cell, as a match /
				// mismatch, a deletion, or none (an insertion) -- depends on the rows alone, not on how the walk got there.  So the wave works
				// it out for 64 places at once, the ones the walk is about to pass: lane 4 j + d holds the decision at node t_top - j, column
				// t_x - j + d (a match / mismatch step goes down one node or more and left one column; d = the nodes skipped so far), each lane
				// with the reference's own loop over its node's in-edges (bspoa.h:2360-2392, 2466-2476).  The walk then follows the lanes'
				// decisions with v_readlane -- a dozen scalar instructions a step -- until it leaves the tile, meets a place a lane could not
				// decide from the ring (7), or changes state; the states of a deletion / insertion run and everything odd take the steps
				// further down, one at a time, as before.
				int t_top = -1, t_x = 0;
				uint32_t d_cat = 7u, d_word = 0; int d_w = 0, d_sx = 0, d_h = 0, d_H = 0, t_j[5] = {64, 64, 64, 64, 64};
				// The four columns of a tile's node j start at t_base(j) = the skew the walk has shown so far: it loses columns against the
				// nodes wherever it skips nodes (a graph of many reads: more than one node a step) or takes a deletion, and gains one with
				// an insertion.  t_sig = that drift per node in 1/256 (of the tile), sig = the running estimate (halved into every tile's).
				int t_sig = 0, sig = 0;
				auto t_base = [&](int j) -> int { return j ? ((j * t_sig) >> 8) - 1 : 0; };       // (one column to the left of the expected one: an insertion on the way)
				// Follow the tile from lane `id`, all of its steps at once.  Every lane knows the lane its decision leads to (t_j[0]; 64 = out of
				// the tile, itself = the walk changes state or cannot be decided here); t_j[k] is that map applied 2^k times.  Lane s composes the
				// maps of s's bits and so stands on the place the walk reaches after s steps; the first s whose place does not go on is the
				// number of steps taken, their words go out in one store, and the state after them is read off one or two lanes.  Steps that
				// go on: match / mismatch, and deletions / insertions of length one (the runs the walk meets most).  -> false: the trip is over
				auto chase = [&](int id) -> bool {
					int pos = id;
#pragma unroll
					for(int k = 0; k < 5; k++){
						const int nx = __builtin_amdgcn_ds_bpermute((pos & 63) << 2, t_j[k]);
						if((lane >> k) & 1) pos = (pos >= 64) ? 64 : nx;
					}
					const uint32_t cat = (pos >= 64) ? 8u : (uint32_t)__builtin_amdgcn_ds_bpermute((pos & 63) << 2, (int)d_cat);
					const uint32_t wd = (uint32_t)__builtin_amdgcn_ds_bpermute((pos & 63) << 2, (int)d_word);
					const uint32_t stop = (uint32_t)__ballot(cat != 0u);
					const int L = stop ? __builtin_ctz(stop) : 31;                              // (lanes 0 .. 31 stand for steps: a longer stay in the tile goes on next trip)
					const int Lc = min(L, ecap - ne);
					if(lane < Lc) ev[ne + lane] = wd;
					ne += Lc;
#ifdef POA_PROF
					tq_chase += Lc; tq_cat = 0;
#endif
					if(Lc < L){ status = BSA_POA_ST_EVENTS; done = true; return false; }
					const int pe = __builtin_amdgcn_readlane(pos, L);
					const uint32_t ce = stop ? (uint32_t)__builtin_amdgcn_readlane((int)cat, L) : 9u;
					if(L > 0){
						const int n_was = n, x_was = x;
						// (what the last step taken leads to -- not what the place it leads to holds: that lane may be one that could not be decided,
						// its own cell outside the window kept of its row)
						const int last = __builtin_amdgcn_readlane(pos, L - 1);
						n = __builtin_amdgcn_readlane(d_w, last); x = __builtin_amdgcn_readlane(d_sx, last); Hs1 = __builtin_amdgcn_readlane(d_h, last);
						nidx = n; Hs2 = 0;
						// nodes gone down against columns gone left, in 1/256 per node
						if(n_was - n >= 4) sig = (sig + min(max(((n_was - n) + (x - x_was)) * 256 / (n_was - n), 0), 224)) >> 1;
					}
#ifdef POA_PROF
					if(ce == 7u){ tq_why[__builtin_amdgcn_readlane(d_why, pe) % 10]++; if(__builtin_amdgcn_readlane(d_why, pe) == 7 && lane == 0 && blockIdx.x == 0) printf("  window miss: window start %d, cell %d, node %d of tile at %d (column %d), lo %d, walker's cell %d, x %d, slen %d\n", __builtin_amdgcn_readlane(d_dbg, pe) >> 16, (int)(int16_t)(__builtin_amdgcn_readlane(d_dbg, pe) & 0xFFFF), t_top - (pe >> 2), t_top, t_x, lo, wpp, x, slen); }
					{ int why = (int)ce; if(ce == 8u){ const int jj = t_top - n; why = (jj >= POA_TILE) ? 8 : (x - t_x + jj - t_base(jj) < 0) ? 10 : 11; } tq_end[why]++; tq_endL[why] += L; }
#endif
					if(ce == 7u) return !(n == 0 || x < 0);                                     // undecided here: the step below takes it
					if(ce == 3u){ bt = 1u; Hs2 = 1; Hs0 = __builtin_amdgcn_readlane(d_h, pe); }
					else if(ce == 1u || ce == 2u){ bt = (ce == 1u) ? 2u : 4u; Hs2 = 1; }
					return false;                                                               // (out of the tile or paused: the next trip goes on)
				};
				while(!done){
					POA_TRK(0)
#ifdef POA_PROF
					{ __builtin_amdgcn_s_waitcnt(0xC07F); const long long t_ = clock64(); tq_c[tq_cat] += t_ - tq_l; tq_k[tq_cat]++; tq_l = t_; tq_cat = (bt == 0xFFFFFFFFu) ? 2 : (bt == 1u) ? 3 : 4; }
#endif
					if(n == 0 || x < 0){ done = true; break; }
					if constexpr(ROWS != 0){
						// the read's window: the places looked at lie between x - POA_TILE and x + 3
						if(__builtin_expect(qlo > 0 && x - POA_TILE - 4 < qlo, 0)){ const int nl_ = max(0, qlo - POA_QW / 2); q_fill(nl_, qlo - nl_); qlo = nl_; }
					}
					bool build = false;
#ifndef POA_NO_TILE
					if(bt == 0xFFFFFFFFu){
						const int tj = t_top - n, td = (t_top >= 0 && (unsigned)tj < (unsigned)POA_TILE) ? x - t_x + tj - t_base(tj) : -1;
						if((unsigned)td >= 4u) build = true;
						else if(__builtin_amdgcn_readlane(d_H, tj * 4 + td) == Hs1){          // (always, by construction; if not, the step below finds out why)
							if(!chase(tj * 4 + td)) continue;
						}
#ifdef POA_PROF
						else { tq_why[0]++; if(lane == 0 && blockIdx.x == 0 && tq_why[0] < 6) printf("  H differs at lane %d of tile (%d, %d): walker (%d, %d) carries %d, lane has %d, code %d, why %d, window start %d cell %d, lo %d\n", tj * 4 + td, t_top, t_x, n, x, Hs1, __builtin_amdgcn_readlane(d_H, tj * 4 + td), __builtin_amdgcn_readlane((int)d_cat, tj * 4 + td), __builtin_amdgcn_readlane(d_why, tj * 4 + td), __builtin_amdgcn_readlane(d_dbg, tj * 4 + td) >> 16, (int)(int16_t)(__builtin_amdgcn_readlane(d_dbg, tj * 4 + td) & 0xFFFF), lo); }
#endif
					}
#endif
					// The ring: the walker's node and its predecessors (at most POA_TNEAR nodes back; further ones are read from HBM) have to be
					// in it -- a step may have gone down further than that --; a new tile asks for the next nodes below it every time and takes
					// them when their slots are free.  (ONE place requests: the registers a refill waits in are then written at one point of
					// the loop and nothing has to move -- and wait for -- them.  The window of cells follows the walker's last known cell.)
					{
						const bool need = lo > 0 && n < lo + POA_TNEAR + 1;
						if(npend && (need || (build && n < plo + POA_TN))) node_commit();
						const bool need2 = lo > 0 && n < lo + POA_TNEAR + 1;
						if(!npend && lo > 0 && (need2 || build || n < lo + POA_TNEAR + 1 + TC)) node_request(cell_of(x - (((n - lo + TC / 2) * (256 - sig)) >> 8)) + wpp - cell_of(wpx));       // (the refill's nodes lie that far below the walker, on average)
						if(need2) continue;
					}
					// the walker's node: inside the ring now
					if(h_n != n){ const uint4 r0 = t_r0[n & (POA_TN - 1)]; h_rpos = (uint32_t)POA_UNI(r0.x); h_first = (uint32_t)POA_UNI(r0.z); h_w3 = (uint32_t)POA_UNI(r0.w); h_n = n; }
					const int nrpos = (int)h_rpos, nin = (int)(h_w3 & 0xFFFFu), nfirst = (int)h_first;
					const uint32_t nbase = (h_w3 >> 16) & 0xFFu, nflags = h_w3 >> 24;
					wpp = x - nrpos; wpx = x;
					POA_TRK(1)
					{
						const int ef = build ? max(nfirst - 24, 0) : nfirst;       // (a tile looks at the in-edges of the nodes below the walker as well)
						if(epend && elo > 0 && ef < elo) edge_commit();
						const bool eneed = elo > 0 && ef < elo;
						if(!epend && elo > 0 && ef < elo + POA_TE / 2) edge_request();
						if(eneed) continue;
					}
					POA_TRK(2)
					if(build){
#ifdef POA_PROF
							tq_build++;
#endif
							t_sig = sig;
							const int j = lane >> 2;
							const int m = n - j, xm = x - j + (lane & 3) + t_base(j), ms = m & (POA_TN - 1);
							const uint4 r0m = t_r0[ms];
							const int mfirst = (int)r0m.z, mnin = (int)(r0m.w & 0xFFFFu), ppm = xm - (int)r0m.x;
							const uint32_t mbase = (r0m.w >> 16) & 0xFFu;
							bool ok = m >= 1 && m >= lo && xm >= 0 && ppm >= 0 && ppm < bw && mfirst >= elo && mfirst + mnin <= ehi;
#ifdef POA_PROF
							d_why = (m < 1) ? 1 : (m < lo) ? 2 : (xm < 0) ? 3 : (ppm < 0 || ppm >= bw) ? 4 : (mfirst < elo) ? 5 : (mfirst + mnin > ehi) ? 6 : 0;
#endif
							const int pc = min(max(ppm, 0), bw - 1), pcm = max(pc - 1, 0), xq = max(xm, 0);
							const uint32_t cwm = t_rows[ms * POA_TW + (pc & (POA_TW - 1))], cmm = t_rows[ms * POA_TW + (pcm & (POA_TW - 1))];
							const int u0m = t_u0[ms], c0m = t_c0[ms];
							const uint32_t nbv = QCODE(xq);
							uint4 ed = ((const uint4*)t_edges)[mfirst & (POA_TE - 1)];
							ok = ok && (unsigned)(pc - c0m) < (unsigned)POA_TW && (unsigned)(pcm - c0m) < (unsigned)POA_TW;
#ifdef POA_PROF
							if(d_why == 0 && !ok) d_why = 7;
							d_dbg = (c0m << 16) | (ppm & 0xFFFF);
#endif
							const int Hm = u0m + (int)(int16_t)(cwm & 0xFFFFu);
							const int hmn = (ppm >= 1) ? u0m + (int)(int16_t)(cmm & 0xFFFFu) : u0m;
							const int sbv = (nbv & 8u) ? BSA_EPI8_MIN : (((nbv & 3u) == mbase) ? a.M + ((r0m.w >> 24) & 1u ? a.refbonus : 0) : a.X);
							const bool hpv = (nbv & 12u) == 4u;
							uint32_t btc = 0, bti = 0xFFFFFFFFu; int bnode = 0, bh = 0;
							bool g1 = false, g2 = false, end1 = false, end2 = false; int w1 = 0, w2 = 0, h1 = 0, h2 = 0;       // the first in-edge a deletion run would take, and whether it ends there
							for(int k = 0; __ballot(ok && k < mnin) != 0ull; k++){
								const int w = (int)ed.x, wr = (int)ed.z; const uint32_t cov = ed.y;
								ed = ((const uint4*)t_edges)[(mfirst + k + 1) & (POA_TE - 1)];         // (the next edge travels with this one's cells)
								const int pp = xm - wr;
								const bool valid = ok && k < mnin && pp >= 0 && pp <= bw;
								const int ws = w & (POA_TN - 1);
								const int p1 = min(max(pp, 0), bw - 1), p0 = max(min(pp, bw) - 1, 0);
								const uint32_t cw = t_rows[ws * POA_TW + (p1 & (POA_TW - 1))], cm = t_rows[ws * POA_TW + (p0 & (POA_TW - 1))];
								const int u0w = t_u0[ws]; int c0w = t_c0[ws];
								const uint32_t wbase = (t_r0[ws].w >> 16) & 0xFFu;
								asm volatile("" : "+v"(c0w));                                   // (read with the cells, not in a branch after them)
								// a predecessor below the ring, or its cells outside the window kept of it
								const bool bad = w < lo || (unsigned)(p1 - c0w) >= (unsigned)POA_TW || (unsigned)(p0 - c0w) >= (unsigned)POA_TW;
								ok = ok && !(valid && bad);
#ifdef POA_PROF
								if(d_why == 0 && valid && bad) d_why = (w < lo) ? 8 : 9;
#endif
								const int rbase = (w == 0) ? h0init : u0w;
								const int hm = (pp >= 1) ? rbase + (int)(int16_t)(cm & 0xFFFFu) : u0w;
								const int hc = rbase + (int)(int16_t)(cw & 0xFFFFu);
								const bool f15 = pp == 0 && wr == 0 && (ovl || w == 0);
								int sc = sbv + ((hpv && wbase != mbase) ? 1 : 0);
								if(f15) sc -= u0w;
								const bool inb = valid && pp < bw;
								const bool m0 = valid && (pp != 0 || f15) && hm + sc == Hm;
								const bool m1 = inb && hc + (PW ? sx8(cw >> 16) : E) == Hm;
								const bool m2 = (PW == 2) && inb && hc + sx8(cw >> 24) == Hm;
								if(m0 && (cov > btc || (cov == btc && (bti & 0xFFu) != 0u))){ bti = 0u; btc = cov; bnode = w; bh = hm; }
								if(m1 && cov > btc){ bti = 1u; btc = cov; bnode = w; bh = hm; }
								if(m2 && cov > btc){ bti = 2u; btc = cov; bnode = w; bh = hm; }
								if(m1 && !g1){ g1 = true; w1 = w; h1 = hc; end1 = (PW ? sx8(cw >> 16) : a.O + E) == a.O + E; }
								if(m2 && !g2){ g2 = true; w2 = w; h2 = hc; end2 = sx8(cw >> 24) == a.Q + P; }
							}
							{
								const uint32_t code = !ok ? 7u : (bti == 0xFFFFFFFFu) ? 3u : bti;
								const int cost1 = (PW == 2) ? max(a.O + E, a.Q + P) : a.O + E;
								bool on = code == 0u;
								d_word = (uint32_t)m << 3; d_w = bnode; d_sx = xm - 1; d_h = bh; d_H = Hm;
								if(code == 1u && g1 && end1){ on = true; d_word |= 2u; d_w = w1; d_sx = xm; d_h = h1; }          // a deletion of one node (bspoa.h:2325-2358)
								else if(code == 2u && g2 && end2){ on = true; d_word |= 4u; d_w = w2; d_sx = xm; d_h = h2; }
								else if(code == 3u){ d_h = hmn; if(hmn + cost1 == Hm){ on = true; d_word |= 1u; d_w = m; } }       // an insertion of one base (bspoa.h:2412-2440)
								d_cat = on ? 0u : code;
								t_top = n; t_x = x;
								// where the decision leads: a lane of this tile, out of it (64), or nowhere (the walk stops here: the lane itself)
								int nid = lane;
								if(on){
									const int wj = n - d_w;
									if((unsigned)wj < (unsigned)POA_TILE){ const int dd = d_sx - (x - wj) - t_base(wj); nid = ((unsigned)dd < 4u) ? wj * 4 + dd : 64; }
									else nid = 64;
								}
								t_j[0] = nid;
#pragma unroll
								for(int k = 1; k < 5; k++){
									const int nx = __builtin_amdgcn_ds_bpermute((t_j[k - 1] & 63) << 2, t_j[k - 1]);
									t_j[k] = (t_j[k - 1] >= 64) ? 64 : nx;
								}
							}
#ifdef POA_PROF
							tq_cat = 1;
#endif
							if(__builtin_amdgcn_readlane(d_H, 0) == Hs1) (void)chase(0);       // (what is left undecided there: the next trip's step, with that node's record)
							continue;
					}
#ifdef POA_PROF
					tq_slow++;
#endif
					if(__builtin_expect(bt == 0xFFFFFFFFu, 1)){
						bool coop = nin <= 64 && nfirst >= elo && nfirst + nin <= ehi;
						uint32_t nb = 0; int sbase = 0;
						if(coop){
							// one in-edge per lane: what the loop further down does edge after edge (bspoa.h:2360-2392), then its choice -- the
							// reference keeps the candidate with the largest coverage, the first one on ties unless a later one is a match /
							// mismatch move and the kept one is not; coverage 0 is only ever taken as a match / mismatch move.  Two LDS round
							// trips: the edge (and the read's base), then everything about the predecessor at once -- a lane without an edge
							// reads what some edge slot and ring row hold and is masked afterwards.
							const uint4 ed = ((const uint4*)t_edges)[(nfirst + lane) & (POA_TE - 1)];
							const uint32_t qword = QCODE(x);
							const int w = (int)ed.x, wr = (int)ed.z; const uint32_t cov = ed.y;
							const int pp = x - wr;
							const bool valid = lane < nin && pp >= 0 && pp <= bw;
							const int ws = w & (POA_TN - 1);
							const int p1 = min(max(pp, 0), bw - 1), p0 = max(min(pp, bw) - 1, 0);
							const uint32_t cw = t_rows[ws * POA_TW + (p1 & (POA_TW - 1))], cm = t_rows[ws * POA_TW + (p0 & (POA_TW - 1))];
							const int u0w = t_u0[ws], c0w = t_c0[ws];
							const uint4 r0w = t_r0[ws];
							// a predecessor below the ring (0.3 %) or its cells outside the window kept of it: edge after edge
							if(__builtin_expect(__ballot(valid && (w < lo || (unsigned)(p1 - c0w) >= (unsigned)POA_TW || (unsigned)(p0 - c0w) >= (unsigned)POA_TW)) != 0ull, 0)) coop = false;
							nb = (uint32_t)POA_UNI(qword);
							sbase = (nb & 8u) ? BSA_EPI8_MIN : (((nb & 3u) == nbase) ? a.M + ((nflags & 1u) ? a.refbonus : 0) : a.X);
							if(coop){
								const int rbase = (w == 0) ? h0init : u0w;                  // (the head's row is row_init; its ubegs[0] is not its first cell)
								const int hm = (pp >= 1) ? rbase + (int)(int16_t)(cm & 0xFFFFu) : u0w;
								const int hc = rbase + (int)(int16_t)(cw & 0xFFFFu);
								const uint32_t wbase = (r0w.w >> 16) & 0xFFu;
								const bool f15 = pp == 0 && wr == 0 && (ovl || w == 0);
								int sc = sbase + (((nb & 12u) == 4u && wbase != nbase) ? 1 : 0);
								if(f15) sc -= u0w;
								const bool inb = valid && pp < bw;
								const bool m0 = valid && (pp != 0 || f15) && hm + sc == Hs1;
								const bool m1 = inb && hc + (PW ? sx8(cw >> 16) : E) == Hs1;
								const bool m2 = (PW == 2) && inb && hc + sx8(cw >> 24) == Hs1;
								POA_TRK(3)
								const unsigned long long bany = __ballot(m0 || m1 || m2);
								uint32_t C = 0;
								for(unsigned long long r = bany; r; r &= r - 1) C = max(C, (uint32_t)__builtin_amdgcn_readlane((int)cov, __builtin_ctzll(r)));
								const bool atc = (m0 || m1 || m2) && cov == C;
								const unsigned long long b0c = __ballot(atc && m0), bc = __ballot(atc);
								if(b0c){
									// a match / mismatch column: the step itself (bspoa.h:2394-2410) taken at once, and the next node's record with it
									const int win = __builtin_ctzll(b0c);
									EMIT(n, x, 0u);
									x--; n = __builtin_amdgcn_readlane(w, win); nidx = n; Hs1 = __builtin_amdgcn_readlane(hm, win); Hs2 = 0;
									h_rpos = (uint32_t)__builtin_amdgcn_readlane((int)r0w.x, win); h_first = (uint32_t)__builtin_amdgcn_readlane((int)r0w.z, win);
									h_w3 = (uint32_t)__builtin_amdgcn_readlane((int)r0w.w, win); h_n = n;
								} else if(bc && C > 0){
									const int win = __builtin_ctzll(bc);
									bt = ((__ballot(m1) >> win) & 1ull) ? 2u : 4u; Hs2 = 1;
								} else start_insertion(nrpos);
								POA_TRK(4)
#ifdef POA_PROF
								tq_n++;
#endif
								continue;
							}
						} else {
							nb = (uint32_t)POA_UNI(QCODE(x));
							sbase = (nb & 8u) ? BSA_EPI8_MIN : (((nb & 3u) == nbase) ? a.M + ((nflags & 1u) ? a.refbonus : 0) : a.X);
						}
						uint32_t btc = 0, bti = 0xFFFFFFFFu; int bnode = 0, bh = 0;
						for(int k = 0; k < nin; k++){
							const int ek = nfirst + k;
							bsa_poa_edge_t ed; if(ek >= elo && ek < ehi) ed = t_edges[ek & (POA_TE - 1)]; else ed = gedges[ek];
							const int w = POA_UNI(ed.src), wr = POA_UNI(ed.src_rpos);
							const uint32_t cov = (uint32_t)POA_UNI(ed.cov);
							if(x < wr || x > bw + wr) continue;
							const int pp = x - wr;
							const int u0w = POA_UNI(U0(w));
							const uint32_t wbase = (w >= lo) ? (uint32_t)POA_UNI((t_r0[w & (POA_TN - 1)].w >> 16) & 0xFFu) : (uint32_t)POA_UNI(nodes[w].base);
							const int hm = (pp >= 1) ? POA_UNI(HH(w, pp - 1, CELL(w, pp - 1), u0w)) : u0w;          // H(pp - 1); at pp = 0 the block start ubegs[0]
							int hc = 0, ec = 0, qc = 0;
							if(pp < bw){ const uint32_t cw = (uint32_t)POA_UNI(CELL(w, pp)); hc = HH(w, pp, cw, u0w); ec = sx8(cw >> 16); qc = sx8(cw >> 24); }
							int ft = 0, s, scr0, scr1 = BSA_SCORE_MIN, scr2 = BSA_SCORE_MIN;
							if(pp == bw) ft |= (1 << 2) | (1 << 4);
							else if(pp == 0){ if(wr == 0 && (ovl || w == 0)) ft |= 1 << 15; else ft |= 1; }
							Hs0 = hm;
							s = sbase;
							if(!(nb & 8u) && (nb & 4u) && wbase != nbase) s += 1;
							if(ft & (1 << 15)) s -= u0w;
							scr0 = (ft & 1) ? BSA_SCORE_MIN : s;
							if(pp < bw){
								const int us = hc - hm;
								scr1 = us + (PW ? ec : E);
								scr2 = (PW == 2) ? us + qc : -BSA_SCORE_MIN;
							}
#define POA_PICK(i_, sc_) if(Hs0 + (sc_) == Hs1){ if(cov > btc || (cov == btc && (i_) == 0 && (bti & 0xFFu) != 0u)){ bti = (i_); btc = cov; bnode = w; bh = Hs0; } }
							POA_PICK(0u, scr0) POA_PICK(1u, scr1) POA_PICK(2u, scr2)
#undef POA_PICK
						}
						if(bti == 0xFFFFFFFFu) start_insertion(nrpos);
						else if(bti == 0u){
							EMIT(n, x, 0u);
							x--; n = bnode; nidx = bnode; Hs1 = bh; Hs2 = 0;
						}
						else if(bti == 1u){ bt = 2u; Hs2 = 1; }
						else { bt = 4u; Hs2 = 1; }
					} else if(bt == 2u || bt == 4u){
						// a deletion run: the first predecessor whose E (Q) continues it (bspoa.h:2325-2358)
						EMIT(n, x, bt);
						bool found = false;
						if(nin <= 64 && nfirst >= elo && nfirst + nin <= ehi){
							// one in-edge per lane, the first lane that continues the run
							const uint4 ed = ((const uint4*)t_edges)[(nfirst + lane) & (POA_TE - 1)];
							const int w = (int)ed.x, pp = x - (int)ed.z, ws = w & (POA_TN - 1);
							const bool inb = lane < nin && pp >= 0 && pp < bw;
							const int pq = min(max(pp, 0), bw - 1);
							const uint32_t cw = t_rows[ws * POA_TW + (pq & (POA_TW - 1))];
							const int u0w = t_u0[ws], c0w = t_c0[ws];
							if(__ballot(inb && (w < lo || (unsigned)(pq - c0w) >= (unsigned)POA_TW)) == 0ull){
								const int hw = ((w == 0) ? h0init : u0w) + (int)(int16_t)(cw & 0xFFFFu);
								const int qv = (bt == 2u) ? (PW ? sx8(cw >> 16) : a.O + E) : sx8(cw >> 24);
								const unsigned long long hit = __ballot(inb && hw + qv == Hs1);
								if(hit){
									const int win = __builtin_ctzll(hit);
									const int qw = __builtin_amdgcn_readlane(qv, win);
									Hs0 = __builtin_amdgcn_readlane(hw, win); n = __builtin_amdgcn_readlane(w, win);
									if(qw == ((bt == 2u) ? a.O + E : a.Q + P)){ bt = 0xFFFFFFFFu; Hs1 = Hs0; Hs2 = 0; }
									else { Hs1 -= (bt == 2u) ? E : P; Hs2++; }
									continue;
								}
								status = BSA_POA_ST_TRACE; done = true;
								continue;
							}
						}
						for(int k = 0; k < nin && !found; k++){
							const int ek = nfirst + k;
							bsa_poa_edge_t ed; if(ek >= elo && ek < ehi) ed = t_edges[ek & (POA_TE - 1)]; else ed = gedges[ek];
							const int w = POA_UNI(ed.src), wr = POA_UNI(ed.src_rpos);
							if(x < wr || x >= wr + bw) continue;
							const uint32_t cw = (uint32_t)POA_UNI(CELL(w, x - wr));
							Hs0 = HH(w, x - wr, cw, POA_UNI(U0(w)));
							int qv;
							if(bt == 2u) qv = PW ? sx8(cw >> 16) : a.O + E;
							else qv = sx8(cw >> 24);
							if(Hs0 + qv != Hs1) continue;
							n = w;
							if(qv == ((bt == 2u) ? a.O + E : a.Q + P)){ bt = 0xFFFFFFFFu; Hs1 = Hs0; Hs2 = 0; }
							else { Hs1 -= (bt == 2u) ? E : P; Hs2++; }
							found = true;
						}
						if(!found){ status = BSA_POA_ST_TRACE; done = true; }
					} else if(bt == 1u){
						{
							// an insertion run along the node's own row, all of its lengths at once: lane l tries length Hs2 + l (bspoa.h:2412-2440)
							const int ns = n & (POA_TN - 1), ppx = x - nrpos, c = ppx - 1 - lane;
							const uint32_t cwc = t_rows[ns * POA_TW + (max(c, 0) & (POA_TW - 1))];
							const int u0v = t_u0[ns], c0v = t_c0[ns];
							const bool there = lane <= x && c >= -1;                          // (the walk gets as far as this length)
							const bool inw = c < 0 || (unsigned)(c - c0v) < (unsigned)POA_TW;
							const int V = (c >= 0) ? u0v + (int)(int16_t)(cwc & 0xFFFFu) : u0v;
							const int D = Hs0 - __builtin_amdgcn_readlane(V, 0);              // (0: what the walk carries is H of the cell left of it)
							const int Ln = Hs2 + lane;
							const int t = (PW == 2) ? max(a.O + E * Ln, a.Q + P * Ln) : a.O + E * Ln;
							const unsigned long long hit = __ballot(there && V + D + t == Hs1), bad = __ballot(!there || !inw);
							if(ppx >= 0 && ppx <= bw && hit && !(bad & ((hit & (0ull - hit)) * 2ull - 1ull))){
								const int f = __builtin_ctzll(hit);
								const int Lc = min(f + 1, ecap - ne);
								if(lane < Lc) ev[ne + lane] = ((uint32_t)n << 3) | 1u;
								ne += Lc;
								if(Lc < f + 1){ status = BSA_POA_ST_EVENTS; done = true; continue; }
								x -= f + 1; bt = 0xFFFFFFFFu; Hs1 = __builtin_amdgcn_readlane(V, f) + D; Hs0 = Hs1; Hs2 = 0;
								continue;
							}
						}
						EMIT(n, x, bt);
						const int t = (PW == 2) ? max(a.O + E * Hs2, a.Q + P * Hs2) : a.O + E * Hs2;
						x--;
						if(Hs0 + t == Hs1){ bt = 0xFFFFFFFFu; Hs1 = Hs0; Hs2 = 0; }
						else if(x >= 0){
							const int pp = x - nrpos;
							if(pp < 0){ status = BSA_POA_ST_TRACE; done = true; }
							else {
								// us[pp] = H(pp) - H(pp - 1), us[0] = H(0) - ubegs[0]; Hs0 is H(pp) here
								const int u0v = POA_UNI(U0(n));
								const int hm = (pp == 0) ? u0v : POA_UNI(HH(n, pp - 1, CELL(n, pp - 1), u0v));
								Hs0 -= POA_UNI(HH(n, pp, CELL(n, pp), u0v)) - hm;
								Hs2++;
							}
						}
					} else {
						EMIT(n, x, bt);
						x--;
						n = nidx;
						bt = 0xFFFFFFFFu;
					}
				}
#undef POA_UNI
#ifdef POA_PROF
				if(lane == 0 && blockIdx.x == 0) printf("poa walk: %d steps in %d chases (%lld clocks), %d tiles (%lld), plain steps one at a time %d (%lld), insertion %d (%lld), deletion %d (%lld); %lld clocks in all\n", tq_chase, tq_k[0], tq_c[0], tq_k[1], tq_c[1], tq_k[2], tq_c[2], tq_k[3], tq_c[3], tq_k[4], tq_c[4], (long long)(clock64() - tq_0));
				if(lane == 0 && blockIdx.x == 0) printf("poa chases end at: deletion %d (%d steps before), q-deletion %d (%d), insertion %d (%d), undecided %d (%d), below the tile's nodes %d (%d), left of its columns %d (%d), right of them %d (%d), pause %d (%d); drift estimate %d/256\n",
					tq_end[1], tq_endL[1], tq_end[2], tq_endL[2], tq_end[3], tq_endL[3], tq_end[7], tq_endL[7], tq_end[8], tq_endL[8], tq_end[10], tq_endL[10], tq_end[11], tq_endL[11], tq_end[9], tq_endL[9], sig);
				if(lane == 0 && blockIdx.x == 0) printf("poa undecided places: node 0 %d, node below the ring %d, column < 0 %d, cell outside the band %d, in-edges below the edge ring %d, above it %d, own cells outside the window %d, predecessor below the ring %d, its cells outside the window %d, other %d\n",
					tq_why[1], tq_why[2], tq_why[3], tq_why[4], tq_why[5], tq_why[6], tq_why[7], tq_why[8], tq_why[9], tq_why[0]);
				if(lane == 0 && blockIdx.x == 0) printf("poa walk profile: %d cooperative steps; clocks per step: loop top %.0f, node ring %.0f, record + edge ring %.0f, edges evaluated %.0f, choice + move %.0f\n", tq_n,
					(double)tq[0] / max(tq_n, 1), (double)tq[1] / max(tq_n, 1), (double)tq[2] / max(tq_n, 1), (double)tq[3] / max(tq_n, 1), (double)tq[4] / max(tq_n, 1));
#endif
#undef EMIT
			}
			n = __shfl(n, 0); done = __shfl((int)done, 0) != 0;
		}
		// the steps join the other programs' in one packed array (what the host downloads): reserve, then copy as a wave
		ne = __shfl(ne, 0);
		unsigned long long off = 0;
		if(lane == 0 && ne > 0) off = atomicAdd(a.packed_used, (unsigned long long)ne);
		off = __shfl(off, 0);
		__builtin_amdgcn_s_waitcnt(0);
		for(int i = lane; i < ne; i += 64) a.packed[off + i] = *(const volatile uint32_t*)&ev[i];
		if(lane == 0){
			rs.status = status; rs.nevents = ne; rs.fin_node = n; rs.fin_x = x; rs.reserved = (int)(uint32_t)off;
			a.res[blockIdx.x] = rs;
		}
	}
}

// ---- host side ---------------------------------------------------------------------------------------------------------
extern "C" int bsa_ctx_get_stream_internal(bsa_ctx_t *ctx, hipStream_t *st);
extern "C" int bsa_ctx_time_begin_internal(bsa_ctx_t *ctx, double cells, void **stop_event);
extern "C" int bsa_ctx_time_end_internal(bsa_ctx_t *ctx, void *stop_event);
extern "C" int bsa_ctx_scratch_internal(bsa_ctx_t *ctx, int slot, size_t bytes, void **out);
// bands wider than 256 columns (bsa_poa_gen.hip)
extern "C" int bsa_poa_graph_gen_supported(const bsa_sweep_params_t *par);
int bsa_poa_graph_gen_run(bsa_ctx_t *ctx, const bsa_poa_node_t *d_nodes, size_t nnodes, const bsa_poa_edge_t *d_edges, const bsa_poa_cand_t *d_cands,
		const bsa_poa_prog_t *d_progs, size_t nprogs, const uint8_t *d_queries, const bsa_sweep_params_t *par,
		bsa_poa_result_t *d_results, uint32_t *d_steps, uint32_t *d_packed, uint64_t *d_packed_used);

static const size_t POA_LDS_MAX = 160u * 1024u - 1024u;
static size_t poa_tile_bytes(uint32_t bw){ (void)bw; return (size_t)POA_TILE_BYTES; }
static size_t poa_qn_bytes(uint32_t bw, uint32_t max_slen){ return ((((size_t)max_slen + bw + 16) / 8 + 2) * 4 + 15) & ~(size_t)15; }
static size_t poa_ring_bytes(uint32_t bw, uint32_t nl){ return (size_t)(nl + POA_NEAR + POA_DRAIN) * bw * 4; }
static size_t poa_front_bytes(uint32_t bw, uint32_t nl){ return (std::max(poa_ring_bytes(bw, nl) + (size_t)(nl + POA_NEAR + POA_DRAIN) * 8, poa_tile_bytes(bw)) + 15) & ~(size_t)15; }

static uint32_t poa_rows_cpl(uint32_t bw){ return bw <= 64 ? 1u : bw <= 128 ? 2u : 4u; }
// Ring rows of the row-at-a-time forward pass (a power of two; inputs further back are read from HBM, after a wait for the stores and a
// cache invalidate: microseconds).  8 rows keep a read's LDS at the traceback's 10 KB -- 16 reads per CU, what a launch of thousands of
// windows wants; a launch that cannot fill the CUs anyway takes 16 rows: in the deep graph of a window with 64 reads 2 % of the inputs lie
// more than 7 nodes back, none more than 15.
static uint32_t poa_rows_r(size_t nprogs){
	const char *e = bsa_env("BSA_POA_FWD_RING");          // test knob: 2 / 4 / 8 / 16 ring rows (2: nearly every two-input node reads a row back from HBM)
	if(e){ const int r = atoi(e); if(r == 2 || r == 4 || r == 8 || r == 16) return (uint32_t)r; }
	return nprogs > 2048 ? 8u : 16u;
}
static size_t poa_rows_ring_bytes(uint32_t bw, uint32_t R){ return (size_t)R * (poa_rows_cpl(bw) * 64 + 2 * POA_ROWS_PAD) * 4; }
static size_t poa_rows_front_bytes(uint32_t bw, uint32_t R){ return (poa_rows_ring_bytes(bw, R) + R * 8 + 15) & ~(size_t)15; }       // (the forward pass's part; the traceback's ring + POA_QW bytes take its place afterwards)
static size_t poa_rows_qb_bytes(uint32_t bw, uint32_t max_slen){ return ((size_t)max_slen + poa_rows_cpl(bw) * 64 + 8 + 15) & ~(size_t)15; }
// the row-at-a-time forward pass (poa_forward_rows): its scans need gapo <= 0 and gapo1 + gape1 <= gape1 <= gape2 <= 0 (the guard of
// bsa_poa_graph_supported has the signs), and its ring rows + the read's profile (or the traceback's ring) in LDS
static bool poa_rows_supported(const bsa_rows_params_t *rp, int pw, uint32_t bw, uint32_t max_slen){
	if(pw == 2 && rp->gape1 > rp->gape2) return false;
	return std::max(poa_rows_front_bytes(bw, 16) + poa_rows_qb_bytes(bw, max_slen), poa_tile_bytes(bw) + POA_QW) <= POA_LDS_MAX;
}

extern "C" int bsa_poa_graph_supported(const bsa_sweep_params_t *par, uint32_t max_slen){
	if(!par) return 0;
	const bsa_rows_params_t *rp = &par->rows;
	const uint32_t bw = (rp->bandwidth + 15u) / 16u * 16u;
	if(bw < 16u || bw > 256u) return 0;
	const int pw = bsa_get_piecewise(rp->gapo1, rp->gape1, rp->gapo2, rp->gape2, (int)bw);
	// exact arithmetic is the reference's int8 arithmetic only while nothing saturates (cf. bsa_align8_x_supported): scores and
	// gap costs small against the int8 range, the synthetic cell behind a moved row's end (bsalign.h:2357-2389) representable
	const int m = rp->M + rp->refbonus + 1, n = -rp->X, ge = -rp->gape1, go = -rp->gapo1;
	if(m < 0 || n < 0 || ge < 0 || go < 0 || rp->M < 0 || rp->refbonus < 0) return 0;
	int g = go + ge;
	if(pw == 2){
		const int ge2 = -rp->gape2, go2 = -rp->gapo2;
		if(ge2 < 0 || go2 < 0) return 0;
		g = std::max(g, go2 + ge2);
	}
	if(m + 3 * g > 64 || n + m + g > 100 || m + 2 * n > 128) return 0;      // (m + 2 n: the head row's seed (min - max) + S stays a byte on a mismatch, bsalign.h:2899-2910)
	if(std::min((int)rp->X, -g) - 1 - m - g < -100) return 0;
	if((int)(bw / 16) * ge > 60) return 0;
	for(uint32_t nl = 64; nl >= 8; nl >>= 1)
		if(poa_front_bytes(bw, nl) + poa_qn_bytes(bw, max_slen) + POA_NQ * sizeof(bsa_poa_node_t) <= POA_LDS_MAX) return (int)nl;
	return 0;
}

extern "C" int bsa_poa_graph_run(bsa_ctx_t *ctx, const bsa_poa_node_t *d_nodes, size_t nnodes, const bsa_poa_edge_t *d_edges, const bsa_poa_cand_t *d_cands,
		const bsa_poa_prog_t *d_progs, size_t nprogs, const uint8_t *d_queries, uint32_t max_slen, const bsa_sweep_params_t *par,
		bsa_poa_result_t *d_results, uint32_t *d_steps, uint32_t *d_packed, uint64_t *d_packed_used, uint32_t *d_rows, int32_t *d_u0){
	if(!ctx || !par || (nprogs && (!d_nodes || !d_progs || !d_queries || !d_results || !d_steps || !d_packed || !d_packed_used))) return BSA_E_ARG;
	if(nprogs == 0) return BSA_OK;
	if(nprogs > 0x0FFFFFF0ull || (d_rows == nullptr) != (d_u0 == nullptr)) return BSA_E_ARG;
	const int nl = bsa_env("BSA_POA_FORCE_GEN") ? 0 : bsa_poa_graph_supported(par, max_slen);          // (test knob: every program through the generic-width kernel)
	if(nl == 0){
		// bands wider than this kernel takes (a window's first read: the whole read) with scores inside the guard: the generic-width kernel
		// (bsa_poa_gen.hip), same program, results and step words; it keeps its rows to itself
		if(d_rows == nullptr && bsa_poa_graph_gen_supported(par) && !bsa_env("BSA_POA_NO_GEN"))
			return bsa_poa_graph_gen_run(ctx, d_nodes, nnodes, d_edges, d_cands, d_progs, nprogs, d_queries, par, d_results, d_steps, d_packed, d_packed_used);
		return BSA_E_UNSUPPORTED;
	}
	hipStream_t st;
	int rc = bsa_ctx_get_stream_internal(ctx, &st);
	if(rc != BSA_OK) return rc;
	const bsa_rows_params_t *rp = &par->rows;
	const uint32_t bw = (rp->bandwidth + 15u) / 16u * 16u;
	const int pw = bsa_get_piecewise(rp->gapo1, rp->gape1, rp->gapo2, rp->gape2, (int)bw);
	if(!d_rows){
		void *ws = nullptr;
		const size_t rb = ((size_t)nnodes * bw * 4 + 255) & ~(size_t)255;
		rc = bsa_ctx_scratch_internal(ctx, 0, rb + (size_t)nnodes * 4 + 256, &ws);
		if(rc != BSA_OK) return rc;
		d_rows = (uint32_t*)ws; d_u0 = (int32_t*)((uint8_t*)ws + rb);
	}
	PoaArgs a;
	a.nodes = d_nodes; a.edges = d_edges; a.cands = d_cands; a.progs = d_progs; a.queries = d_queries;
	a.rows = d_rows; a.u0 = d_u0; a.res = d_results; a.steps = d_steps; a.packed = d_packed; a.packed_used = (unsigned long long*)d_packed_used;
	if(hipMemsetAsync(d_packed_used, 0, 8, st) != hipSuccess) return BSA_E_HIP;
	// forward pass: a row at a time wherever its preconditions hold (BSA_POA_FWD=wf keeps the wavefront)
	bool rows_fwd = poa_rows_supported(rp, pw, bw, max_slen);
	{ const char *fe = bsa_env("BSA_POA_FWD"); if(fe && fe[0] == 'w') rows_fwd = false; }
	a.bw = bw; a.W = bw / 16; a.nl = (uint32_t)nl;
	if(rows_fwd){
		a.R = poa_rows_r(nprogs);
		a.ri_off = (uint32_t)poa_rows_ring_bytes(bw, a.R);
		a.qn_off = (uint32_t)poa_rows_front_bytes(bw, a.R);
	} else {
		a.R = (uint32_t)nl + POA_NEAR + POA_DRAIN;
		a.ri_off = (uint32_t)poa_ring_bytes(bw, (uint32_t)nl);
		a.qn_off = (uint32_t)poa_front_bytes(bw, (uint32_t)nl);
	}
	a.nq_off = a.qn_off + (rows_fwd ? 0u : (uint32_t)poa_qn_bytes(bw, max_slen));     // (the row-at-a-time pass keeps the read as bytes only)
	a.mode = rp->mode; a.M = rp->M; a.X = rp->X; a.refbonus = rp->refbonus;
	a.O = rp->gapo1; a.E = rp->gape1; a.Q = rp->gapo2; a.P = rp->gape2; a.T = par->T;
	{
		// row_movx's synthetic cells (bsalign.h:2357-2389) and row_init's start (bsalign.h:2094-2126), as the POA passes the score range (bspoa.h:2226, 2240)
		const int nt_max = a.M + a.refbonus + 1, nt_min = a.X;
		const int goe = (pw == 2) ? a.Q + a.P : a.O + a.E;
		a.c0 = std::min(nt_min, goe) - 1 - nt_max + goe;
		a.d = (pw == 2) ? (a.O - a.Q) / (a.P - a.E) : (int)bw + 1;
		a.xp = (pw == 2) ? (a.Q - a.O) / (a.E - a.P) : 1;
		const int type = a.mode & 3;
		a.head_u0 = (type == BSA_MODE_OVERLAP) ? 0 : nt_max - nt_min;
		{ const char *ws = bsa_env("BSA_POA_WIN_SHIFT"); a.win_shift = ws ? atoi(ws) : 0; }
	}
	const size_t lds = rows_fwd ? std::max((size_t)a.nq_off + poa_rows_qb_bytes(bw, max_slen), poa_tile_bytes(bw) + POA_QW) : (size_t)a.nq_off + POA_NQ * sizeof(bsa_poa_node_t);
	void *stop = nullptr;
	rc = bsa_ctx_time_begin_internal(ctx, 0.0, &stop);
	if(rc != BSA_OK) return rc;
	// (the attribute is raised once per instantiation and size, and a failure closes the timing scope)
#define POA_LAUNCH(PWV, RV) do {                                                                                                          \
		static size_t lds_set = 0;                                                                                                        \
		if(lds > lds_set){                                                                                                                \
			if(hipFuncSetAttribute((const void*)k_poa_wf<PWV, RV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess){ (void)bsa_ctx_time_end_internal(ctx, stop); return BSA_E_HIP; } \
			lds_set = lds; }                                                                                                              \
		hipLaunchKernelGGL((k_poa_wf<PWV, RV>), dim3((uint32_t)nprogs), dim3(64), lds, st, a); } while(0)
#define POA_LAUNCH_R(RV) do { if(pw == 0) POA_LAUNCH(0, RV); else if(pw == 1) POA_LAUNCH(1, RV); else POA_LAUNCH(2, RV); } while(0)
	if(!rows_fwd) POA_LAUNCH_R(0);
	else if(bw <= 64) POA_LAUNCH_R(1);
	else if(bw <= 128) POA_LAUNCH_R(2);
	else POA_LAUNCH_R(4);
#undef POA_LAUNCH_R
#undef POA_LAUNCH
	if(hipGetLastError() != hipSuccess){ (void)bsa_ctx_time_end_internal(ctx, stop); return BSA_E_HIP; }
	return bsa_ctx_time_end_internal(ctx, stop);
}

// step words (node << 3 | bt) -> (node, x, bt): x starts at the best end cell's column and moves left with every M and I step
extern "C" void bsa_poa_expand_steps(const uint32_t *steps, bsa_poa_result_t *res, bsa_poa_event_t *events){
	int x = res->maxoff;
	for(int i = 0; i < res->nevents; i++){
		events[i].node = steps[i] >> 3; events[i].bt = steps[i] & 7u; events[i].x = x;
		if(events[i].bt <= 1u) x--;
	}
	res->reserved = 0;
}

namespace {
struct PoaDevBuf {
	void *p = nullptr;
	~PoaDevBuf(){ if(p) (void)hipFree(p); }
	hipError_t alloc(size_t n){ return hipMalloc(&p, n ? n : 16); }
};
}

extern "C" int bsa_poa_graph_host(bsa_ctx_t *ctx, const bsa_poa_node_t *nodes, size_t nnodes, const bsa_poa_edge_t *edges, size_t nedges,
		const bsa_poa_cand_t *cands, size_t ncands, const bsa_poa_prog_t *progs, size_t nprogs,
		const uint8_t *queries, size_t query_bytes, const bsa_sweep_params_t *par,
		bsa_poa_result_t *results, bsa_poa_event_t *events, size_t events_cap, bsa_poa_cell_t *rows_out, int32_t *u0_out){
	if(!ctx || !par || !results || (nprogs && (!nodes || !progs || !queries || !events))) return BSA_E_ARG;
	if(nprogs == 0) return BSA_OK;
	uint32_t max_slen = 0;
	for(size_t k = 0; k < nprogs; k++){
		const bsa_poa_prog_t &pg = progs[k];
		if((size_t)pg.first_node + pg.nnodes > nnodes || (size_t)pg.first_edge + pg.nedges > nedges || (size_t)pg.first_cand + pg.ncands > ncands) return BSA_E_ARG;
		if(pg.query_off + pg.slen > query_bytes || pg.first_event + pg.event_cap > events_cap) return BSA_E_ARG;
		max_slen = std::max(max_slen, pg.slen);
		for(size_t i = 0; i < pg.nnodes; i++){
			const bsa_poa_node_t &nd = nodes[pg.first_node + i];
			if(nd.rpos > pg.slen) return BSA_E_ARG;                 // (the kernels index the read's profile by rpos)
			for(int j = 0; j < 2; j++) if((nd.in[j].toff_kind & BSA_POA_IN_PRESENT) && nd.in[j].src >= i) return BSA_E_ARG;
			if((size_t)nd.first_in + nd.n_in > pg.nedges) return BSA_E_ARG;
			for(size_t e = 0; e < nd.n_in; e++) if(edges[pg.first_edge + nd.first_in + e].src >= i) return BSA_E_ARG;
		}
		for(size_t i = 0; i < pg.ncands; i++) if(cands[pg.first_cand + i].node >= pg.nnodes) return BSA_E_ARG;
	}
	if((bsa_env("BSA_POA_FORCE_GEN") || bsa_poa_graph_supported(par, max_slen) == 0) && (rows_out || u0_out || !bsa_poa_graph_gen_supported(par) || bsa_env("BSA_POA_NO_GEN"))) return BSA_E_UNSUPPORTED;
	hipStream_t st;
	int rc = bsa_ctx_get_stream_internal(ctx, &st);
	if(rc != BSA_OK) return rc;
	const uint32_t bw = (par->rows.bandwidth + 15u) / 16u * 16u;
#define PCHK(x) do { if((x) != hipSuccess) return BSA_E_HIP; } while(0)
	// ONE device buffer that stays the context's (this is what a single window calls once per read: ten hipMalloc / hipFree pairs a call cost
	// more than the kernel), ONE upload of the tables packed side by side, the rows only when the caller wants them
	auto up = [](size_t b){ return (b + 255) & ~(size_t)255; };
	const bool want_rows = rows_out && u0_out;
	const size_t o_n = 0, o_e = o_n + up(nnodes * sizeof(bsa_poa_node_t)), o_c = o_e + up(nedges * sizeof(bsa_poa_edge_t)), o_p = o_c + up(ncands * sizeof(bsa_poa_cand_t)),
		o_q = o_p + up(nprogs * sizeof(bsa_poa_prog_t)), in_bytes = o_q + up(query_bytes + 64);
	const size_t o_r = in_bytes, o_v = o_r + up(nprogs * sizeof(bsa_poa_result_t)), o_pk = o_v + up(events_cap * 4), o_rows = o_pk + up(events_cap * 4 + 16),
		o_u0 = o_rows + (want_rows ? up(nnodes * bw * 4) : 0), total_bytes = o_u0 + (want_rows ? up(nnodes * 4) : 0);
	void *wsv = nullptr;
	rc = bsa_ctx_scratch_internal(ctx, 2, total_bytes + 256, &wsv);
	if(rc != BSA_OK) return rc;
	uint8_t *ws = (uint8_t*)wsv;
	static thread_local std::vector<uint8_t> stage;
	if(stage.size() < in_bytes) stage.resize(in_bytes + in_bytes / 4);
	memcpy(stage.data() + o_n, nodes, nnodes * sizeof(bsa_poa_node_t));
	if(nedges) memcpy(stage.data() + o_e, edges, nedges * sizeof(bsa_poa_edge_t));
	if(ncands) memcpy(stage.data() + o_c, cands, ncands * sizeof(bsa_poa_cand_t));
	memcpy(stage.data() + o_p, progs, nprogs * sizeof(bsa_poa_prog_t));
	memcpy(stage.data() + o_q, queries, query_bytes);
	PCHK(hipMemcpyAsync(ws, stage.data(), in_bytes, hipMemcpyHostToDevice, st));
	rc = bsa_poa_graph_run(ctx, (const bsa_poa_node_t*)(ws + o_n), nnodes, (const bsa_poa_edge_t*)(ws + o_e), (const bsa_poa_cand_t*)(ws + o_c), (const bsa_poa_prog_t*)(ws + o_p), nprogs,
		ws + o_q, max_slen, par, (bsa_poa_result_t*)(ws + o_r), (uint32_t*)(ws + o_v), (uint32_t*)(ws + o_pk) + 4, (uint64_t*)(ws + o_pk),
		want_rows ? (uint32_t*)(ws + o_rows) : nullptr, want_rows ? (int32_t*)(ws + o_u0) : nullptr);
	if(rc != BSA_OK) return rc;
	PCHK(hipMemcpyAsync(results, ws + o_r, nprogs * sizeof(bsa_poa_result_t), hipMemcpyDeviceToHost, st));
	PCHK(hipStreamSynchronize(st));
	{
		// the packed steps of all programs in one copy, expanded into (node, x, bt) here
		size_t total = 0;
		for(size_t k = 0; k < nprogs; k++) if(results[k].nevents > 0) total = std::max(total, (size_t)(uint32_t)results[k].reserved + (size_t)results[k].nevents);
		static thread_local std::vector<uint32_t> pk;
		if(pk.size() < total) pk.resize(total + total / 4);
		if(total){ PCHK(hipMemcpyAsync(pk.data(), (const uint32_t*)(ws + o_pk) + 4, total * 4, hipMemcpyDeviceToHost, st)); PCHK(hipStreamSynchronize(st)); }
		for(size_t k = 0; k < nprogs; k++) bsa_poa_expand_steps(pk.data() + (uint32_t)results[k].reserved, &results[k], events + progs[k].first_event);
	}
	if(want_rows){
		std::vector<uint32_t> cells(nnodes * bw);
		PCHK(hipMemcpyAsync(cells.data(), ws + o_rows, nnodes * bw * 4, hipMemcpyDeviceToHost, st));
		PCHK(hipMemcpyAsync(u0_out, ws + o_u0, nnodes * 4, hipMemcpyDeviceToHost, st));
		PCHK(hipStreamSynchronize(st));
		for(size_t k = 0; k < nprogs; k++){
			for(size_t i = 0; i < progs[k].nnodes; i++){
				const size_t nidx = progs[k].first_node + i;
				// the head's cells are relative to its H(0) (row_init: 0 in overlap mode, gapo1 + gape1 otherwise), every other row's to its ubegs[0] (the same number there)
				const int base = (i == 0) ? (((par->rows.mode & 3) == BSA_MODE_OVERLAP) ? 0 : par->rows.gapo1 + par->rows.gape1) : u0_out[nidx];
				for(uint32_t p = 0; p < bw; p++){
					const uint32_t cw = cells[nidx * bw + p];
					bsa_poa_cell_t &o = rows_out[nidx * bw + p];
					o.h = base + (int)(int16_t)(cw & 0xFFFFu); o.e = (int8_t)(cw >> 16); o.q = (int8_t)(cw >> 24); o.tag = 0;
				}
			}
		}
	}
#undef PCHK
	return BSA_OK;
}
