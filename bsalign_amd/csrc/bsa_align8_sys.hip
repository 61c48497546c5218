// bsa_align8_sys.hip -- whole-query bands longer than 256 columns (the reference CLI's default `-W 0` on long reads, main.c:314-315;
// the first command of example/run.sh): the 8-bit DP of banded_striped_epi8_seqalign_pairwise (bsalign.h:3854-4050) with the band
// never moving, as a SYSTOLIC wavefront, and its traceback from 4-bit codes.
//
// A band that covers the whole query never moves (bsalign.h:3338: qoff + bw >= qlen), every row has band offset 0, and inside the
// exact-arithmetic guard of the compact path (bsa_align8_sys_supported = the score bounds of bsa_align8_codes_supported) none of
// the reference's saturating int8 operations clamps: the stored differences are exact, i.e. the DP is the affine-gap recurrence on
// absolute scores with the reference's own boundary rules.  No striping is needed for that: one wave runs one pair, lane l owns
// target row 64 b + l of the current block of 64 rows and walks the query; at step t it computes column x = t - l, so the row above
// (lane l - 1) delivered H(x, y-1) and E(x, y-1) exactly one step earlier -- a DPP wave shift per value and step, no LDS, no
// waiting.  The last row of a block is the boundary of the next one: it leaves through a 64-entry LDS ring, 64 columns per
// coalesced store, and comes back 64 columns per coalesced load (lane 0 picks its column with v_readlane).
// Kept literally (absolute-score form of the rules, cf. the POA wavefront bsa_poa_wf.hip):
//   * row -1 (row_init, bsalign.h:2094-2140): H = gapo + gape (x + 1), e = -63; ubegs[0] = smax - smin with u[0] = gapo + gape + smin - smax
//   * band cell 0 (bsalign.h:2899-2907) with rh = 0 on row 0 and gapo + gape y below (bsalign.h:3932-3946): h0 = rh - ubegs[0] + S, kept
//     (clamped to 63) if >= u[0] + e[0], else -63; every later row has ubegs[0] = H(0, y-1), u[0] = 0 (the re-basing of :2632-2633)
//   * F restarts from "H above - 63" at every running block of W = bw / 16 cells (bsalign.h:2909-2931 + the F-penetration :2639-2652)
// Traceback codes (the four facts backcal tests per cell): M (h == S-path; at column 0 against the UNclamped
// rh + S), D (h == u + e; at column 0 in the frame mismatch of the re-based row: h - rh == u[0] + e[0]), R (h + gapo + gape >= f + gape:
// an insertion reaching the next cell opens here), Od (the stored e is a fresh opening).  Row y, columns 32 k .. 32 k + 31: four
// dwords {M, D, R, Od}, column c at bit 31 - (c & 31); rows are roundup(qlen, 32) / 2 bytes apart.
#include "bsa_common.h"

static __device__ __forceinline__ int sys_shr1(int fill, int x){                 // lane l <- lane l - 1 over the whole wave; lane 0 <- fill
	int r = __builtin_amdgcn_update_dpp(fill, x, 0x138, 0xf, 0xf, false);        // DPP wave_shr:1
	asm("" : "+v"(r));
	return r;
}

struct SysHdr { int32_t score, reserved[3]; };

static __host__ __device__ inline size_t bsa_sys_row_bytes(uint32_t qlen){ return ((size_t)qlen + 31) / 32 * 16; }
static __host__ __device__ inline size_t bsa_sys_bnd_off(){ return sizeof(SysHdr); }
static __host__ __device__ inline size_t bsa_sys_codes_off(uint32_t qlen){ return (bsa_sys_bnd_off() + ((size_t)qlen + 64) * 8 + 63) & ~(size_t)63; }

size_t bsa_align8_sys_slot_bytes(uint32_t qlen, uint32_t tlen){
	return ((bsa_sys_codes_off(qlen) + (size_t)tlen * bsa_sys_row_bytes(qlen) + ((size_t)qlen + tlen + 16) * 4) + 255) & ~(size_t)255;
}

bool bsa_align8_sys_supported(const Align8Args &a, int pw){
	if(pw > 1 || (a.mode & 3) != BSA_MODE_GLOBAL) return false;
	const int g = -((int)(int8_t)(a.gapo1 + a.gape1)), m = a.smax, n = -a.smin;
	if(m < 0 || n < 0 || g < 0 || (int8_t)a.gape1 > 0 || (int8_t)a.gapo1 > 0 || ((int8_t)a.gapo1 == 0) != (pw == 0)) return false;
	return m + 3 * g <= 64 && n + m + g <= 100;
}

template<int PW>
__global__ void __launch_bounds__(64) k_align8_fwd_sys(const Align8Args a){
	extern __shared__ __align__(16) uint8_t lds[];
	int2 *oring = (int2*)lds;                        // the block's last row on its way out: 64 columns
	uint8_t *qs = lds + 64 * sizeof(int2);           // the query
	const uint32_t ppos = a.first + blockIdx.x;
	const uint32_t pair = a.order[ppos];
	const int lane = threadIdx.x;
	const int qlen = (int)a.qlen[pair], tlen = (int)a.tlen[pair];
	uint8_t *slot = a.rows + a.slot_off[ppos];
	SysHdr *hdr = (SysHdr*)slot;
	if(a.status[pair] != 0u || qlen == 0 || tlen == 0){ if(lane == 0) hdr->score = (int)0x80000000u; return; }
	int2 *bnd = (int2*)(slot + bsa_sys_bnd_off());
	uint8_t *codes = slot + bsa_sys_codes_off((uint32_t)qlen);
	const size_t rb = bsa_sys_row_bytes((uint32_t)qlen);
	const uint8_t *qp = a.qst + a.qpoff[pair], *tp = a.tst + a.tpoff[pair];
	for(int i = lane * 16; i < qlen; i += 64 * 16) *(uint4*)(qs + i) = *(const uint4*)(qp + i);      // (the staged query is padded)
	__syncthreads();
	const int GO = a.gapo1, GE = a.gape1, GOE = GO + GE;
	const int Wc = a.ref_bw ? (int)(a.ref_bw / 16u) : ((qlen + 15) / 16 * 16) / 16;       // cells per running block of the reference's striping of its band
	const int first_u = (int)(int8_t)(GOE + a.smin - a.smax), B0 = a.smax - a.smin;
	const int nblk = (tlen + 63) / 64;
	for(int blk = 0; blk < nblk; blk++){
		const int y = blk * 64 + lane;
		const bool rowok = y < tlen;
		const int tb = rowok ? (int)tp[y] & 3 : 0;
		const uint32_t mr = (tb == 0) ? a.mrow[0] : (tb == 1) ? a.mrow[1] : (tb == 2) ? a.mrow[2] : a.mrow[3];      // matrix[q * 4 + tb], q = 0..3, one byte each
		const int rh = (y == 0) ? 0 : GO + GE * y;               // H left of column 0 (bsalign.h:3932-3946)
		int Hout = 0, Eout = 0, Hd = 0, F = 0, wblk = 0;
		uint32_t pM = 0, pD = 0, pR = 0, pO = 0;
		uint8_t *myrow = codes + (size_t)(rowok ? y : 0) * rb;
		// boundary chunk: columns [c0, c0 + 64) of the row above the block, one column per lane
		int2 cur = make_int2(0, 0), nxt = make_int2(0, 0);
		auto load_chunk = [&](int c0) -> int2 {
			const int x = c0 + lane;
			if(blk == 0){ const int h = GOE + GE * x; return make_int2(h, h + BSA_EPI8_MIN); }      // row -1: e = -63
			return (x < qlen) ? bnd[x] : make_int2(0, 0);
		};
		cur = load_chunk(0);
		const int nsteps = qlen + 63;
		for(int t = 0; t < nsteps; t++){
			if((t & 63) == 0) nxt = load_chunk(t + 64);
			const int x = t - lane;
			// from the row above: lane l - 1's cell of the previous step is this lane's column
			int Hu = sys_shr1(0, Hout), Eu = sys_shr1(0, Eout);
			{
				const int bh = __builtin_amdgcn_readlane(cur.x, t & 63), be = __builtin_amdgcn_readlane(cur.y, t & 63);
				if(lane == 0){ Hu = bh; Eu = be; }
			}
			if((t & 63) == 63) cur = nxt;
			const bool on = x >= 0 && x < qlen;
			const int qb = on ? (int)qs[x] : 4;
			const int S = (int)(int8_t)((mr >> (8 * (qb & 3))) & 0xffu);
			int diag = Hd + S, cmpM = diag, cmpD = Eu;
			if(__any(x == 0)){
				if(x == 0){
					// band cell 0: the seed rule and the F restart; the M / D facts of column 0 in their own frames
					const int ub0 = (y == 0) ? B0 : Hu;
					const int u0 = (y == 0) ? first_u : 0;
					const int e0 = Eu - Hu;
					int h0 = rh - ub0 + S;
					const int tt = u0 + (PW == 0 ? GE : e0);
					h0 = (h0 >= tt) ? min(h0, BSA_EPI8_MAX) : BSA_EPI8_MIN;
					diag = ub0 + h0;
					cmpM = rh + S;
					cmpD = rh + u0 + (PW == 0 ? GOE : e0);
					F = ub0 + BSA_EPI8_MIN;
					wblk = 0;
				}
			}
			if(wblk == Wc){ F = max(F, Hd + BSA_EPI8_MIN); wblk = 0; }       // a new running block: F restarts from the sentinel
			const int Ein = (PW == 0) ? Hu + GE : Eu;
			const int H = max(max(diag, Ein), F);
			const int t1 = H + GOE, tF = F + GE, tE = Ein + GE;
			const uint32_t fM = (H == cmpM) ? 1u : 0u;
			const uint32_t fD = (H == ((PW == 0 && x != 0) ? Hu + GOE : cmpD)) ? 1u : 0u;
			const uint32_t fR = (PW == 0 || t1 >= tF) ? 1u : 0u;
			const uint32_t fO = (PW == 0 || tE <= t1) ? 1u : 0u;
			if(on){
				pM = (pM << 1) | fM; pD = (pD << 1) | fD; pR = (pR << 1) | fR; pO = (pO << 1) | fO;
				Hout = H; Eout = (PW == 0) ? H : max(tE, t1);
				F = (PW == 0) ? H + GE : max(tF, t1);
				Hd = Hu;
				wblk++;
				if((x & 31) == 31 || x == qlen - 1){
					const int sh = 31 - (x & 31);
					if(rowok) *(uint4*)(myrow + (size_t)(x >> 5) * 16) = make_uint4(pM << sh, pD << sh, pR << sh, pO << sh);
					pM = pD = pR = pO = 0;
				}
				if(x == qlen - 1 && y == tlen - 1) hdr->score = H;
			}
			// the block's last row leaves: column x of lane 63 into the ring, a full ring to HBM
			if(lane == 63 && on) oring[x & 63] = make_int2(Hout, Eout);
			if(blk + 1 < nblk && t >= 63 && (((t - 63) & 63) == 63 || t == nsteps - 1)){
				const int c0 = (t - 63) & ~63;
				__builtin_amdgcn_s_waitcnt(0xc07f);             // lgkmcnt(0): the ring entry just written
				if(c0 + lane <= t - 63) bnd[c0 + lane] = oring[lane];
			}
		}
		__builtin_amdgcn_s_waitcnt(0);                           // the boundary row is in HBM before the next block reads it
	}
}

// ---- traceback: one pair per lane, backcal's decisions read off the codes (oracle: backcal_codes; bsalign.h:3704-3852) ----
template<int PW>
__global__ void __launch_bounds__(64) k_align8_trace_sys(const Align8Args a, bsa_result_t *out, uint32_t *cig_cnt, const uint64_t *slot_end){
	const uint32_t g = blockIdx.x * 64u + threadIdx.x;
	if(g >= a.count) return;
	const uint32_t ppos = a.first + g;
	const uint32_t pair = a.order[ppos];
	bsa_result_t rs;
	rs.score = 0; rs.qb = rs.qe = rs.tb = rs.te = 0; rs.mat = rs.mis = rs.ins = rs.del = rs.aln = 0;
	if(a.status[pair] != 0u){ out[pair] = rs; cig_cnt[ppos] = 0; return; }
	const int qlen = (int)a.qlen[pair], tlen = (int)a.tlen[pair];
	const uint8_t *qseq = a.qst + a.qpoff[pair], *tseq = a.tst + a.tpoff[pair];
	const uint8_t *slot = a.rows + a.slot_off[ppos];
	const SysHdr *hdr = (const SysHdr*)slot;
	const uint8_t *codes = slot + bsa_sys_codes_off((uint32_t)qlen);
	const size_t rb = bsa_sys_row_bytes((uint32_t)qlen);
	uint32_t *cig_end = (uint32_t*)(a.rows + slot_end[ppos]);
	uint32_t ncig = 0;
	auto cig_push = [&](uint32_t w){ ncig++; *(cig_end - ncig) = w; };
	auto cig_add = [&](uint32_t cg, uint32_t op, uint32_t sz) -> uint32_t {   // bsalign.h:409-417
		if(op == (cg & 0xf)) return cg + (sz << 4);
		if(cg) cig_push(cg);
		return (sz << 4) | op;
	};
	auto plane = [&](int r, int x, int pl) -> uint32_t { return *(const uint32_t*)(codes + (size_t)r * rb + (size_t)(x >> 5) * 16 + pl * 4); };
	auto bit = [&](int r, int x, int pl) -> bool { return (plane(r, x, pl) >> (31 - (x & 31))) & 1u; };
	bool bad = false;
	rs.score = hdr->score;
	if(rs.score == (int)0x80000000u) bad = true;
	rs.qe = qlen - 1; rs.te = tlen - 1;
	rs.qb = rs.qe; rs.qe++;
	rs.tb = rs.te; rs.te++;
	int prior_match = 0;
	uint32_t cg = 0;
	while(!bad){
		if(rs.qb < 0 || rs.tb < 0) break;
		const uint4 cw = *(const uint4*)(codes + (size_t)rs.tb * rb + (size_t)(rs.qb >> 5) * 16);
		const uint32_t sh = 31u - ((uint32_t)rs.qb & 31u);
		const bool fM = (cw.x >> sh) & 1u, fD = (cw.y >> sh) & 1u;
		int bt;                                               // backcal_cell (bsalign.h:3679-3699): the order of the tests depends on prior_match
		if(prior_match) bt = fM ? 0 : fD ? 2 : 1;
		else bt = fD ? 2 : fM ? 0 : 1;
		prior_match = 1;
		if(bt == 0){
			if(qseq[rs.qb] == tseq[rs.tb]) rs.mat++; else rs.mis++;
			rs.qb--; rs.tb--; rs.aln++;
			cg = cig_add(cg, 0, 1);
		} else if(bt == 1){
			if(rs.qb <= 0){ cg = cig_add(cg, 1, 1); rs.qb--; rs.ins++; rs.aln++; }
			else {
				// the nearest cell to the left at which an insertion reaching its right neighbour opens (bsalign.h:3798-3814)
				int sz = 0;
				uint32_t w = cw.z & ~((2u << sh) - 1u);                                 // R bits of the cells left of qb in this word
				int xw = rs.qb >> 5;
				for(;;){
					if(w){ const int c = xw * 32 + (31 - (int)__builtin_ctz(w)); sz = rs.qb - c; break; }
					if(--xw < 0) break;
					w = plane(rs.tb, xw * 32, 2);
				}
				if(sz == 0){ bad = true; break; }                                       // the reference's scan finds no length either
				cg = cig_add(cg, 1, (uint32_t)sz);
				rs.qb -= sz; rs.ins += sz; rs.aln += sz;
			}
		} else {
			// deletion run: up the column until a row whose stored e is a fresh opening (bsalign.h:3730-3744)
			int len = 1;
			for(;;){
				const int r = rs.tb - len;
				if(r < -1){ bad = true; break; }
				if(r == -1){ if(PW != 0) bad = true; break; }       // linear gaps: an ordinary move; affine: the reference compares real scores there -- literal path
				if(bit(r, rs.qb, 3)) break;
				len++;
			}
			if(bad) break;
			cg = cig_add(cg, 2, (uint32_t)len);
			rs.del += len; rs.aln += len;
			rs.tb -= len;
		}
	}
	if(!bad){
		uint32_t op = 0, sz = 0;          // global: what is left at the top becomes a leading I / D (bsalign.h:3827-3842)
		if(rs.qb >= 0){ op = 1; sz = (uint32_t)rs.qb + 1u; rs.ins += (int)sz; rs.qb = -1; }
		else if(rs.tb >= 0){ op = 2; sz = (uint32_t)rs.tb + 1u; rs.del += (int)sz; rs.tb = -1; }
		rs.aln += (int)sz;
		cg = cig_add(cg, op, sz);
		if(cg) cig_push(cg);
		rs.qb++; rs.tb++;
	} else {
		atomicOr(&a.status[pair], BSA_ST_TRACE);
		rs.qb = rs.qe = rs.tb = rs.te = 0; rs.mat = rs.mis = rs.ins = rs.del = rs.aln = 0;
		ncig = 0;
	}
	out[pair] = rs;
	cig_cnt[ppos] = ncig;
}

hipError_t bsa_launch_align8_fwd_sys(const Align8Args &a, int pw, uint32_t max_qlen, hipStream_t st){
	if(a.count == 0) return hipSuccess;
	const size_t lds = 64 * sizeof(int2) + (((size_t)max_qlen + 15) & ~(size_t)15) + 16;
	if(lds > 64 * 1024){
		if(hipFuncSetAttribute((const void*)k_align8_fwd_sys<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return hipErrorInvalidValue;
		if(hipFuncSetAttribute((const void*)k_align8_fwd_sys<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return hipErrorInvalidValue;
	}
	if(pw == 0) hipLaunchKernelGGL(k_align8_fwd_sys<0>, dim3(a.count), dim3(64), lds, st, a);
	else hipLaunchKernelGGL(k_align8_fwd_sys<1>, dim3(a.count), dim3(64), lds, st, a);
	return hipGetLastError();
}

hipError_t bsa_launch_align8_trace_sys(const Align8Args &a, int pw, bsa_result_t *out, uint32_t *cig_cnt, const uint64_t *slot_end, hipStream_t st){
	if(a.count == 0) return hipSuccess;
	const dim3 grid((a.count + 63u) / 64u);
	if(pw == 0) hipLaunchKernelGGL(k_align8_trace_sys<0>, grid, dim3(64), 0, st, a, out, cig_cnt, slot_end);
	else hipLaunchKernelGGL(k_align8_trace_sys<1>, grid, dim3(64), 0, st, a, out, cig_cnt, slot_end);
	return hipGetLastError();
}
