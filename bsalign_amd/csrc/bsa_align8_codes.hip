// bsa_align8_codes.hip -- traceback of the compact (4-bit code) path of the 8-bit global alignment.
//
// The forward kernel (k_align8_fwd_pk<W, PW, true>, bsa_align8_pk.hip) leaves, per target row, one code row of 16
// lanes x CW dwords (bsa_common.h "COMPACT slot"): for every band cell the outcome of the equality tests the
// reference's backcal would make there (bsalign.h:3667-3852).  These kernels make the same walk from those bits alone:
//   cell (qb, tb):  prior_match ? (M ? match : D ? delete : insert) : (D ? delete : M ? match : insert)
//   insert:         length = distance to the nearest cell on the left whose R bit is set (bsalign.h:3798-3814)
//   delete:         walk up the column until a row whose Od bit is set (bsalign.h:3730-3744)
// The test-only scalar restatement states the same rules (tests/test_oracle_codes.py); both are checked against the
// literal backcal.  Whatever the bits cannot decide (a scan leaves the band; the cases in which the reference itself
// does not terminate) is reported as BSA_ST_TRACE.
//
// Start and end of the walk follow the mode: global starts at (qlen-1, tlen-1) and turns what is left at the top into
// a leading insertion / deletion; overlap and extend start at the best end cell -- the end-of-query candidates the
// forward pass recorded or the maximum of the last row (row_max, bsalign.h:3213-3329, taken here from the end record) --
// and overlap stops without a leading gap (bsalign.h:3822-3842).
#include "bsa_common.h"

// end cell of the non-global modes from the forward pass's end record (bsa_common.h bsa_code_end_t): the best
// end-of-query candidate (strictly greater wins, in row order) against the maximum of the last row, whose tie rules are
// the reference's: per lane the first best 32-vector chunk, lanes reduced in its register order, then the first
// maximum inside the winning chunk (bsalign.h:3213-3329).
// ref_bw != 0 (a whole-query band run at a wider register-kernel width, bsa_api.hip): the row maximum is taken over the
// REFERENCE's band -- ref_bw columns, or roundup(qlen, 16) when ref_bw == 1 -- in the reference's striping of it (WR cells per
// lane), whose tie rules depend on that striping; the end record is in natural order, so only the lane boundaries move.
template<int WK>
static __device__ void codes_end_cell(const uint8_t *rows, uint32_t RB, uint32_t qlen, uint32_t tlen, uint32_t ref_bw, int &score, int &qe, int &te){
	const bsa_code_end_t *er = (const bsa_code_end_t*)(rows + (size_t)bsa_code_rows(tlen) * RB);
	const int8_t *us = (const int8_t*)(er + 1);           // natural band order: lane l, cell x at l * W + x
	const uint32_t W = ref_bw == 0u ? (uint32_t)WK : ref_bw == 1u ? (max(qlen, 1u) + 15u) / 16u : ref_bw / 16u;
	// H in front of band position p, from the kernel's own 16 block starts
	auto before = [&](uint32_t p) -> int { const uint32_t b = p / (uint32_t)WK; int h = er->ubegs[b]; for(uint32_t i = b * (uint32_t)WK; i < p; i++) h += us[i]; return h; };
	int best = BSA_SCORE_MIN, bte = 0;
	for(int l = 0; l < 16; l++){
		const int sc = er->cand_sc[l], t = er->cand_te[l];
		if(sc > best || (sc == best && sc != BSA_SCORE_MIN && t < bte)){ best = sc; bte = t; }
	}
	int lmax[16]; uint32_t lchunk[16];
	for(uint32_t l = 0; l < 16; l++){
		int base = (W == (uint32_t)WK) ? er->ubegs[l] : before(l * W);
		lmax[l] = BSA_SCORE_MIN; lchunk[l] = 0;
		for(uint32_t i = 0, c = 0; i < (uint32_t)W; i += 32, c++){
			const uint32_t n = (i + 32 < (uint32_t)W) ? 32 : (uint32_t)W - i;
			int run = 0, cmax = -32767;
			for(uint32_t x = 0; x < n; x++){
				run += us[l * W + i + x];
				run = min(max(run, -32768), 32767);
				cmax = max(cmax, run);
			}
			if(base + cmax > lmax[l]){ lmax[l] = base + cmax; lchunk[l] = c; }
			base += run;
		}
	}
	int ms, lane;
	{
		int mm[4], ii[4];
		for(int k = 0; k < 4; k++){
			int m01, i01, m23, i23;
			if(lmax[4 + k] > lmax[k]){ m01 = lmax[4 + k]; i01 = 4 + k; } else { m01 = lmax[k]; i01 = k; }
			if(lmax[12 + k] > lmax[8 + k]){ m23 = lmax[12 + k]; i23 = 12 + k; } else { m23 = lmax[8 + k]; i23 = 8 + k; }
			if(m23 > m01){ mm[k] = m23; ii[k] = i23; } else { mm[k] = m01; ii[k] = i01; }
		}
		ms = mm[0]; lane = ii[0];
		for(int k = 1; k < 4; k++) if(mm[k] > ms){ ms = mm[k]; lane = ii[k]; }
	}
	uint32_t x = lchunk[lane] * 32u, jj = x;
	const uint32_t y = min(x + 32u, (uint32_t)W);
	int umax = BSA_SCORE_MIN, uscr = 0;
	for(; x < y; x++){
		uscr += us[lane * W + x];
		if(uscr > umax){ jj = x; umax = uscr; }
	}
	if(ms > best){ score = ms; qe = er->rbeg_last + (int)((uint32_t)lane * W + jj); te = (int)tlen - 1; }
	else { score = best; qe = (int)qlen - 1; te = bte; }
}


// ---- the stall-free walker --------------------------------------------------------------------------------------
// A wave waits (s_waitcnt is a counter, not a per-register flag) whenever ANY lane uses an older load while another
// lane has just issued a new one, so with 64 desynchronised walks per wave "prefetch into registers" still stalls on
// almost every step (a register-window variant was measured: no faster than the plain kernel).  Here all lanes fetch at the
// SAME iterations: every CODE_SVC steps each lane requests the 8 rows above what it already holds, and the rows
// requested at the previous service point -- long since arrived -- are filed into the lane's own LDS ring, from
// where the walk reads them with dynamic indexing and no memory wait.  Ring entry (16 bytes) of row r:
//   x, y, z = code dwords of blocks base .. base+2 (the path's block and its neighbours when the row was requested)
//   w       = band offset of the row | target base << 26 | base << 28
// The query is held as three 16-byte chunks refreshed on the same schedule.  A lane that outruns its ring (a long
// insertion or deletion) simply idles until the next service point.
#define CODE_SVC 8
#ifndef CODE_RING_ROWS
#define CODE_RING_ROWS 32
#endif

// LPW = pairs (active lanes) per wave: the walk is a chain of dependent ALU ops, so a SIMD needs several waves to keep
// issuing; fewer lanes per wave = more waves for the same batch
template<int W, int LPW>
__global__ void __launch_bounds__(64) k_align8_trace_codes_lds(const Align8Args a, bsa_result_t *out, uint32_t *cig_cnt){
	static_assert(W == 4 || W == 8, "one code dword per block");
	constexpr uint32_t RB = 64u;
	constexpr uint32_t FULL = (1u << W) - 1u;
	constexpr int bw = W * 16;
	const int type = a.mode & 3;
	const bool lin = a.gapo1 == 0;                                 // linear gaps (piecewise 0)
	__shared__ uint4 ring[CODE_RING_ROWS * LPW];
	const uint32_t lane = threadIdx.x;
	const uint32_t g = blockIdx.x * (uint32_t)LPW + lane;
	const bool live = lane < (uint32_t)LPW && g < a.count;
	const uint32_t ppos = a.first + (live ? g : 0u);
	const uint32_t pair = a.order[ppos];
	bsa_result_t rs;
	rs.score = 0; rs.qb = rs.qe = rs.tb = rs.te = 0; rs.mat = rs.mis = rs.ins = rs.del = rs.aln = 0;
	const bool skip = !live || a.status[pair] != 0u;
	const uint32_t qlen = a.qlen[pair], tlen = a.tlen[pair];
	const uint8_t *qseq = a.qst + a.qpoff[pair];
	const uint8_t *tseq = a.tst + a.tpoff[pair];
	const int *begs = (const int*)(a.rows + a.slot_off[ppos]);
	const uint8_t *rows = (const uint8_t*)begs + bsa_begs_bytes(tlen);
	uint32_t *cig_end = (uint32_t*)(rows + ((size_t)bsa_code_rows(tlen) + BSA_CODE_SPARE_ROWS) * RB);
	uint32_t ncig = 0;
	auto cig_push = [&](uint32_t w){ ncig++; *(cig_end - ncig) = w; };
	auto cig_add = [&](uint32_t cg, uint32_t op, uint32_t sz) -> uint32_t {   // bsalign.h:409-417
		if(op == (cg & 0xf)) return cg + (sz << 4);
		if(cg) cig_push(cg);
		return (sz << 4) | op;
	};
	auto base_for = [&](uint32_t y) -> uint32_t { return (y == 0u) ? 0u : ((y >= 14u) ? 13u : y - 1u); };
	auto slot = [&](int r) -> uint4& { return ring[((uint32_t)r & (CODE_RING_ROWS - 1u)) * (uint32_t)LPW + (live ? lane : 0u)]; };
	// requested at the last service point, still in registers: two row groups (bsa_common.h: four rows of a block are 16
	// adjacent bytes) of the blocks base .. base+2 -- 48 contiguous bytes each --, their band offsets and target bases
	uint4 pt[2][3], pb[2];
	uint32_t ptq[2] = {0, 0}, pend_base = 0;
	int pend_gtop = 0; bool pend = false;
	int have_lo = 0x7FFFFFFF;                     // lowest row filed in the ring (rows have_lo .. are readable)
	uint4 qc0 = {0, 0, 0, 0}, qc1 = {0, 0, 0, 0}, qc2 = {0, 0, 0, 0}, qp0 = {0, 0, 0, 0}, qp1 = {0, 0, 0, 0}, qp2 = {0, 0, 0, 0};
	int qc_ch = -1000, qp_ch = -1000;             // qc0 = chunk qc_ch, qc1 = qc_ch - 1, qc2 = qc_ch - 2
	auto request = [&](int gtop, uint32_t base){        // row groups gtop and gtop - 1 = rows 4 gtop + 3 .. 4 gtop - 4
#pragma unroll
		for(int u = 0; u < 2; u++){
			const uint32_t gg = (uint32_t)max(gtop - u, 0);
			const uint4 *tp4 = (const uint4*)((const uint32_t*)rows + bsa_code_off(4u * gg, base, 1u));
			pt[u][0] = tp4[0]; pt[u][1] = tp4[1]; pt[u][2] = tp4[2];
			__builtin_memcpy(&pb[u], begs + 4u * gg + 1u, 16);
			__builtin_memcpy(&ptq[u], tseq + 4u * gg, 4);
		}
		pend_gtop = gtop; pend_base = base; pend = true;
	};
	auto comp = [](const uint4 &v, int k) -> uint32_t { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; };
	auto file = [&](){
		if(pend){
#pragma unroll
			for(int u = 0; u < 2; u++){
				if(pend_gtop - u >= 0){
#pragma unroll
					for(int k = 0; k < 4; k++){
						uint4 e;
						e.x = comp(pt[u][0], k); e.y = comp(pt[u][1], k); e.z = comp(pt[u][2], k);
						e.w = comp(pb[u], k) | (((ptq[u] >> (8 * k)) & 3u) << 26) | (pend_base << 28);
						slot(4 * (pend_gtop - u) + k) = e;
					}
				}
			}
			have_lo = 4 * max(pend_gtop - 1, 0);
			pend = false;
		}
	};
	auto chunk = [&](int ch) -> uint4 { return *(const uint4*)(qseq + (size_t)max(ch, 0) * 16u); };
	auto base_in = [&](const uint4 &w, int idx) -> int {
		const uint32_t d = (idx & 12) == 0 ? w.x : (idx & 12) == 4 ? w.y : (idx & 12) == 8 ? w.z : w.w;
		return (int)((d >> (8 * (idx & 3))) & 0xffu);
	};
	bool bad = false, done = skip;
	uint32_t cury = 0;
	if(!skip){
		if(type == BSA_MODE_GLOBAL){
			rs.score = begs[tlen + 1];
			if(rs.score == (int)0x80000000u) bad = true;            // band never reached the query end (bsalign.h:4034)
			rs.qe = (int)qlen - 1; rs.te = (int)tlen - 1;
		} else codes_end_cell<W>(rows, RB, qlen, tlen, a.ref_bw, rs.score, rs.qe, rs.te);
		if(qlen >= (1u << 26)) bad = true;                          // band offsets are kept in 26 bits here
		const int lastbeg = begs[rs.te + 1];
		if(rs.qe < lastbeg || rs.qe >= lastbeg + bw) bad = true;    // end cell outside the stored band
		rs.qb = rs.qe; rs.qe++;
		rs.tb = rs.te; rs.te++;
		cury = (uint32_t)max(min(rs.qb - lastbeg, bw - 1), 0) / W;
		// prologue: two windows, filed immediately
		request(rs.tb >> 2, base_for(cury)); file();
		{ const int lo = have_lo; if(lo > 0){ request((lo - 1) >> 2, base_for(cury)); file(); } }
		qc_ch = rs.qb >> 4; qc0 = chunk(qc_ch); qc1 = chunk(qc_ch - 1); qc2 = chunk(qc_ch - 2);
	}
	int prior_match = 0, dlen = 0;                // dlen != 0: inside a deletion run
	uint32_t cg = 0;
	// Two kinds of iteration.  Seven out of eight handle only the common step -- a match / mismatch cell reached after
	// another cell (prior_match set, M flag set): ~40 instructions.  Everything else (insertions, deletions, band start,
	// window misses, termination) is handled by the general step, which all lanes run together at the service point; a
	// lane that needs it waits for that iteration.  The 16 walks of a wave diverge constantly, so running the general
	// step on every iteration would cost its full instruction count every time (measured: 370 instructions per step).
	bool slow = true;                                 // this lane needs the general step
	for(uint32_t it = 0; __any(!done); it++){
		const bool svc = (it & (CODE_SVC - 1u)) == 0u;
		if(svc){
			// ---- service point (all lanes together): file what was requested last time, request what comes next
			file();
			if(qp_ch != -1000){ qc0 = qp0; qc1 = qp1; qc2 = qp2; qc_ch = qp_ch; qp_ch = -1000; }
			if(!done){
				if(have_lo > 0 && rs.tb - have_lo + 1 < CODE_RING_ROWS - CODE_SVC) request((have_lo - 1) >> 2, base_for(cury));   // the ring holds the live rows + the 8 new ones
				if(rs.qb >= 0 && (rs.qb >> 4) != qc_ch){ qp_ch = rs.qb >> 4; qp0 = chunk(qp_ch); qp1 = chunk(qp_ch - 1); qp2 = chunk(qp_ch - 2); }
			}
		}
		if(done) continue;
		if(!svc){
			// ---- common step: a cell inside the ring and the window, decided from one ring entry.  Written with
			// predicates instead of branches: the 32 walks of a wave are in different cases on every step, and each
			// divergent branch costs its scalar bookkeeping whether or not any lane takes it.
			if(slow) continue;
			const int tbm = max(rs.tb, 1);
			const uint4 e = slot(tbm);
			const uint32_t ew1 = slot(tbm - 1).w;
			const int beg_c = (int)(e.w & 0x3FFFFFFu), beg_p = (int)(ew1 & 0x3FFFFFFu);
			const uint32_t wbase = e.w >> 28;
			const int p = rs.qb - beg_c;
			const uint32_t y = (uint32_t)p / W, off = y - wbase, kk = (uint32_t)p % W, bit = 1u << (W - 1 - kk);
			const int dch = qc_ch - (rs.qb >> 4);
			const bool ok = rs.qb > 0 && rs.tb > 0 && rs.tb - 1 >= have_lo && (uint32_t)p < (uint32_t)bw && off <= 2u && (uint32_t)dch <= 2u;
			const uint32_t wc = off == 0u ? e.x : off == 1u ? e.y : e.z;
			const bool fm = (wc & bit) != 0u, fd = ((wc >> W) & bit) != 0u, fo = ((wc >> (3 * W)) & bit) != 0u;
			const uint32_t cand = ((wc >> (2 * W)) & FULL) & ~((bit << 1) - 1u);
			const bool run_cont = dlen != 0 && !fo;                    // deletion run goes on through this row
			const bool pmatch = (rs.qb != beg_p) && prior_match;
			const bool is_m = !run_cont && fm && (pmatch || !fd);
			const bool is_d = !run_cont && !is_m && fd;
			const bool is_i = !run_cont && !is_m && !is_d;
			if(!ok || (is_i && cand == 0u)){ slow = true; continue; }   // general step: edges, window misses, insertions crossing a block
			cury = y;
			const int sz = (int)__builtin_ctz(cand | 0x80000000u) - (int)(W - 1 - kk);
			const int qbase = base_in(dch == 0 ? qc0 : dch == 1 ? qc1 : qc2, rs.qb & 15);
			const bool same = qbase == (int)((e.w >> 26) & 3u);
			// CIGAR: a deletion is counted row by row (its length is only known when the run ends), so every step adds
			// to exactly one run
			const uint32_t op = is_m ? 0u : is_i ? 1u : 2u;
			const uint32_t n = is_i ? (uint32_t)sz : 1u;
			const bool ext = op == (cg & 0xfu);
			if(!ext && cg) cig_push(cg);
			cg = ext ? cg + (n << 4) : ((n << 4) | op);
			rs.mat += (is_m && same) ? 1 : 0;
			rs.mis += (is_m && !same) ? 1 : 0;
			rs.ins += is_i ? sz : 0;
			rs.del += (is_d || run_cont) ? 1 : 0;
			rs.aln += (int)n;
			rs.qb -= is_m ? 1 : is_i ? sz : 0;
			rs.tb -= is_i ? 0 : 1;
			dlen = (is_d || run_cont) ? 1 : 0;
			prior_match = run_cont ? prior_match : 1;
			continue;
		}
		// ---- general step (service iterations only)
		slow = false;
		if(bad || rs.qb < 0 || rs.tb < 0){
			// a deletion run that reached row -1: an ordinary move with linear gaps; with affine gaps only the e = -63 sentinel of
			// row -1 leads here and the reference then compares real scores -- literal path (BSA_ST_TRACE, hand-over)
			if(dlen && !bad && (!lin || rs.qb >= bw)) bad = true;
			dlen = 0;
			done = true;
			continue;
		}
		if(rs.tb < have_lo || (rs.tb > 0 && rs.tb - 1 < have_lo)){ slow = true; continue; }     // outran the ring: wait for the next service point
		const uint4 e = slot(rs.tb);
		const int beg_c = (int)(e.w & 0x3FFFFFFu);
		const int beg_p = rs.tb > 0 ? (int)(slot(rs.tb - 1).w & 0x3FFFFFFu) : 0;
		const uint32_t wbase = e.w >> 28;
		const int p = rs.qb - beg_c;
		if(p < 0 || p >= bw){ bad = true; slow = true; continue; }
		const uint32_t y = (uint32_t)p / W, k = (uint32_t)p % W, bit = 1u << (W - 1 - k);
		cury = y;
		uint32_t wc;
		if(y >= wbase && y <= wbase + 2u){ const uint32_t off = y - wbase; wc = off == 0u ? e.x : off == 1u ? e.y : e.z; }
		else wc = ((const uint32_t*)rows)[bsa_code_off((uint32_t)rs.tb, y, 1u)];           // drifted two blocks inside one ring: plain load
		const uint32_t pm = wc & FULL, pd = (wc >> W) & FULL, pr = (wc >> (2 * W)) & FULL, po = (wc >> (3 * W)) & FULL;
		if(dlen){
			// deletion run (bsalign.h:3730-3744), counted row by row: this row ends it if its stored e is a fresh opening
			if(po & bit) dlen = 0;
			else { cg = cig_add(cg, 2, 1); rs.del++; rs.aln++; rs.tb--; continue; }
		}
		const bool pmatch = !(rs.qb == beg_p && rs.qb) && prior_match;      // bsalign.h:3761-3764
		const bool fm = (pm & bit) != 0u, fd = (pd & bit) != 0u;
		int bt;                                                       // 0 M, 1 I, 2 D
		if(pmatch) bt = fm ? 0 : fd ? 2 : 1;
		else bt = fd ? 2 : fm ? 0 : 1;
		if(bt == 0){
			const int ch = rs.qb >> 4, dch = qc_ch - ch;
			if(dch < 0 || dch > 2){ slow = true; continue; }           // query chunk not here yet (prior_match untouched)
			const int qbase = base_in(dch == 0 ? qc0 : dch == 1 ? qc1 : qc2, rs.qb & 15);
			const int tbase = (int)((e.w >> 26) & 3u);
			if(qbase == tbase) rs.mat++; else rs.mis++;
			rs.qb--; rs.aln++; rs.tb--;
			cg = cig_add(cg, 0, 1);
		} else if(bt == 1){
			if(rs.qb <= 0){
				cg = cig_add(cg, 1, 1);
				rs.qb--; rs.ins++; rs.aln++;
			} else {
				int sz = 0;
				const uint32_t cand = pr & ~((bit << 1) - 1u);
				if(cand) sz = (int)__builtin_ctz(cand) - (int)(W - 1 - k);
				else {
					int left = (int)k;
					for(int yy = (int)y - 1; yy >= 0 && sz == 0; yy--){
						uint32_t wl;
						if(yy >= (int)wbase && yy <= (int)wbase + 2) wl = ((uint32_t)yy == wbase) ? e.x : ((uint32_t)yy == wbase + 1u) ? e.y : e.z;
						else wl = ((const uint32_t*)rows)[bsa_code_off((uint32_t)rs.tb, (uint32_t)yy, 1u)];
						const uint32_t r2 = (wl >> (2 * W)) & FULL;
						if(r2) sz = left + 1 + (int)__builtin_ctz(r2);
						else left += W;
					}
					if(sz == 0){ bad = true; slow = true; continue; }  // the reference's scan finds no length either: it never terminates
				}
				cg = cig_add(cg, 1, (uint32_t)sz);
				rs.qb -= sz; rs.ins += sz; rs.aln += sz;
			}
		} else {
			cg = cig_add(cg, 2, 1); rs.del++; rs.aln++;
			dlen = 1; rs.tb--;                                           // the rows above decide how far the run goes
		}
		prior_match = 1;
		if(rs.qb < 0 || rs.tb < 0) slow = true;                          // termination is a general step
	}
	if(skip){ if(live){ out[pair] = rs; cig_cnt[ppos] = 0; } return; }
	if(!bad){
		if(type == BSA_MODE_OVERLAP){ if(cg) cig_push(cg); }         // overlap: the alignment simply starts here (bsalign.h:3822-3826)
		else {
			uint32_t op = 0, sz = 0;      // global / extend: what is left at the top becomes a leading I / D (bsalign.h:3827-3842)
			if(rs.qb >= 0){ op = 1; sz = (uint32_t)rs.qb + 1u; rs.ins += (int)sz; rs.qb = -1; }
			else if(rs.tb >= 0){ op = 2; sz = (uint32_t)rs.tb + 1u; rs.del += (int)sz; rs.tb = -1; }
			rs.aln += (int)sz;
			cg = cig_add(cg, op, sz);
			if(cg) cig_push(cg);
		}
		rs.qb++; rs.tb++;
	}
	if(bad){
		atomicOr(&a.status[pair], BSA_ST_TRACE);
		rs.qb = rs.qe = rs.tb = rs.te = 0; rs.mat = rs.mis = rs.ins = rs.del = rs.aln = 0;
		ncig = 0;
	}
	out[pair] = rs;
	cig_cnt[ppos] = ncig;
}

// ---- one walk per wave (bandwidth 128) ---------------------------------------------------------------------------
// The rules above make a cell's outcome a function of the cell alone once the previous step was a match / mismatch
// (prior_match set, no deletion run open): lane i evaluates the cell i steps further up the walker's diagonal --
// (x - i, y - i) -- and one ballot finds the first cell that is not a match.  The run before it is taken in one step;
// the cell that ended it (an insertion with its length scan, a row of a deletion run, a cell the lanes could not
// decide) is then handled by the literal single-cell step, executed once for the wave on values read from that lane
// (the walker's state lives in SGPRs).  Steps per walk = gap events + tiles, not cells.
// A tile is 64 rows (lane i = row T - i, T = 3 mod 4 so that it is sixteen whole row groups): per row the band offsets
// of the row and of the row above, the target base, and a window of four blocks (32 cells) of its code row around the
// diagonal, one coalesced 16-byte load per lane (four rows of one block, bsa_code_off) transposed into LDS (stride 5
// dwords: conflict-free).  The next tile's codes and the band offsets of the one after travel while a tile is walked.
// Anything outside the window is read with a plain load; the query bases sit in a 512-byte LDS window.
#define CWV_QWIN 512
// W = 4, 8, 16 (bandwidth 64, 128, 256).  The window is always 32 cells: eight blocks of one code dword at W = 4 (two 16-byte
// loads per lane), four at W = 8, two blocks of two dwords at W = 16; ND dwords per row in LDS, stride ND + 1.
// FMT 1 (W = 8): code format 1 of bsa_common.h -- M | R << 8 | eight two-bit fields << 16 (min(h - (u + e), -gapo): 0 = D, -gapo = Od).  The fields
// of the window's 32 cells stay interleaved in two registers (bit-reversed, cell c of its half at bits 2c, 2c + 1: the field's high bit first) and
// a step shifts the lane's field out of them; the one literal cell (query column 0 of a row whose band starts there) is answered from a per-tile copy.
template<int W, int FMT = 0>
__global__ void __launch_bounds__(64) k_align8_trace_codes_wave(const Align8Args a, bsa_result_t *out, uint32_t *cig_cnt){
	static_assert(W == 4 || W == 8 || W == 16, "bandwidth 64, 128, 256");
	static_assert(FMT == 0 || W == 8, "code format 1: bandwidth 128");
	const uint32_t NGO = (uint32_t)(-a.gapo1), NGOS = ((NGO & 1u) << 1) | (NGO >> 1);         // -gapo, and with its two bits swapped
	constexpr int bw = 16 * W, ND = (W == 4) ? 8 : 4, STR = ND + 1;
	constexpr uint32_t CW = (W == 16) ? 2u : 1u, RB = 64u * CW, FULL = (W == 16) ? 0xFFFFu : ((1u << W) - 1u);
	constexpr int NB = 32 / W, B0MAX = 16 - NB;                        // blocks in the window, last window start
	__shared__ uint32_t tile[64 * STR];
	__shared__ int s_b0[16];
	__shared__ __attribute__((aligned(8))) uint8_t s_q[CWV_QWIN];
	const uint32_t lane = threadIdx.x;
	const uint32_t ppos = a.first + blockIdx.x;
	const uint32_t pair = a.order[ppos];
	bsa_result_t rs;
	rs.score = 0; rs.qb = rs.qe = rs.tb = rs.te = 0; rs.mat = rs.mis = rs.ins = rs.del = rs.aln = 0;
	if(a.status[pair] != 0u){ if(lane == 0){ out[pair] = rs; cig_cnt[ppos] = 0; } return; }
	const int type = a.mode & 3;
	const bool lin = a.gapo1 == 0;
	const uint32_t qlen = a.qlen[pair], tlen = a.tlen[pair];
	const uint8_t *qseq = a.qst + a.qpoff[pair];
	const uint8_t *tseq = a.tst + a.tpoff[pair];
	const int *begs = (const int*)(a.rows + a.slot_off[ppos]);
	const uint8_t *rows = (const uint8_t*)begs + bsa_begs_bytes(tlen);
	const uint32_t *codes = (const uint32_t*)rows;
	uint32_t *cig_end = (uint32_t*)(rows + ((size_t)bsa_code_rows(tlen) + BSA_CODE_SPARE_ROWS) * RB);
	// CIGAR.  The walker does not merge runs step by step (that logic was a third of the scalar instructions of a step): it records
	// one TOKEN per gap event -- (matches / mismatches since the last event, op, length), one token per lane -- and a gap that simply
	// goes on (no match in between, same op: the rows of a deletion run, consecutive insertions) extends its own token, so that
	// neighbouring tokens never carry the same op.  Every 64 tokens (the last one stays: it may still grow) the lanes turn their
	// tokens into words side by side: an M word when the run is not empty, then the gap's word, positions from two prefix popcounts
	// (word m at cig_end - (m + 1)).  The word sequence is the reference's run-length merge (bsalign.h:409-417) of the same op stream.
	uint32_t ncig = 0, tokN = 0, tokB = 0;            // words written; the lane's token: run length, len << 2 | op (op 0: none)
	// (uniform) tokens held; key = op of the last token << 28 | match / mismatch columns since it, so that "same op, nothing in between" is one
	// compare (a walk over codes has fewer than 2^26 columns: bsa_api.hip sends longer queries to the literal kernels)
	uint32_t ntok = 0, key = 0;
	constexpr uint32_t KEYM = 0x0FFFFFFFu;
	auto tok_flush = [&](uint32_t cnt){
		const bool in = lane < cnt;
		tokN &= KEYM;
		const bool hasA = in && tokN != 0u, hasB = in && (tokB & 3u) != 0u;
		const uint64_t mA = __ballot(hasA), mB = __ballot(hasB);
		const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mA, 0u))
			+ __builtin_amdgcn_mbcnt_hi((uint32_t)(mB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mB, 0u));
		uint32_t *wp = cig_end - (ncig + below + 1u);
		if(hasA){ *wp = tokN << 4; wp--; }                // (tokN: already without the key's op field)
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
	auto cig_finish = [&](){                          // the matches behind the last event, then everything out
		if(key & KEYM){
			if(ntok == 64u){ tok_flush(64u); ntok = 0u; }
			if(lane == ntok){ tokN = key; tokB = 0u; }
			ntok++; key = 0u;
		}
		tok_flush(ntok); ntok = 0u;
	};
	bool bad = false;
	if(type == BSA_MODE_GLOBAL){
		rs.score = begs[tlen + 1];
		if(rs.score == (int)0x80000000u) bad = true;               // band never reached the query end (bsalign.h:4034)
		rs.qe = (int)qlen - 1; rs.te = (int)tlen - 1;
	} else codes_end_cell<W>(rows, RB, qlen, tlen, a.ref_bw, rs.score, rs.qe, rs.te);
	rs.score = __builtin_amdgcn_readfirstlane(rs.score); rs.qe = __builtin_amdgcn_readfirstlane(rs.qe); rs.te = __builtin_amdgcn_readfirstlane(rs.te);
	{ const int lb = begs[rs.te + 1]; if(rs.qe < lb || rs.qe >= lb + bw) bad = true; }
	int x = rs.qe, y = rs.te;
	const int x_start = x, y_start = y;
	rs.qe++; rs.te++;
	// prior: prior_match of the reference (0 only before the first step); dl31: bit 31 set while a deletion run is open.  Both are kept in the
	// form the lanes' tests use them in, so that a step of the walk costs the scalar unit no selects
	int prior = 0;
	uint32_t dl31 = 0u, plim = 0u;                     // plim: 32 once prior_match holds (window cells c < plim take the fast paths)
	uint32_t vmis = 0;                                 // this lane's mismatch count (summed over the wave at the end)
	// ---- tiles
	struct TB { int bc, bp; uint32_t tb; };
	struct TC { uint32_t v0, v1, v2, v3, v4, v5, v6, v7; int b0; };    // W = 8: rows 0..3 of one block; W = 4: of two blocks; W = 16: two rows x two dwords
	auto fetch_begs = [&](int T, TB &t){
		const int r = T - (int)lane;
		if(r >= 0){ t.bc = begs[r + 1]; t.bp = begs[r]; t.tb = (uint32_t)tseq[r]; }
		else { t.bc = 0; t.bp = 0; t.tb = 0xffu; }
	};
	const uint32_t gi = lane >> 2, sl4 = lane & 3u;
	auto fetch_codes = [&](int T, const TB &tb_, int xT, TC &c){      // xT: the diagonal's column at row T
		const int G = (T >> 2) - (int)gi;
		const int bcg = __shfl(tb_.bc, (int)(gi * 4u));              // band offset of the group's top row
		int pp = (xT - 4 * (int)gi) - bcg;
		pp = pp < 0 ? 0 : pp > bw - 1 ? bw - 1 : pp;
		int b0 = (W == 16) ? (pp - 8) >> 4 : (W == 8) ? (pp >> 3) - 1 : (pp >> 2) - 3;       // the diagonal in the middle of the window
		b0 = b0 < 0 ? 0 : b0 > B0MAX ? B0MAX : b0;
		c.b0 = b0;
		c.v0 = c.v1 = c.v2 = c.v3 = c.v4 = c.v5 = c.v6 = c.v7 = 0u;
		if(G >= 0){
			if constexpr (W == 8){
				const uint4 g0 = *(const uint4*)(codes + ((size_t)G * 64u + (size_t)((uint32_t)b0 + sl4) * 4u));
				c.v0 = g0.x; c.v1 = g0.y; c.v2 = g0.z; c.v3 = g0.w;
			} else if constexpr (W == 4){
				const uint4 *gp = (const uint4*)(codes + ((size_t)G * 64u + (size_t)((uint32_t)b0 + 2u * sl4) * 4u));     // blocks b0 + 2 s, b0 + 2 s + 1
				const uint4 g0 = gp[0], g1 = gp[1];
				c.v0 = g0.x; c.v1 = g0.y; c.v2 = g0.z; c.v3 = g0.w; c.v4 = g1.x; c.v5 = g1.y; c.v6 = g1.z; c.v7 = g1.w;
			} else {
				// two dwords per block and row: block b0 + (s >> 1), rows 2 (s & 1) and 2 (s & 1) + 1
				const uint4 g0 = *(const uint4*)(codes + ((size_t)G * 128u + (size_t)((uint32_t)b0 + (sl4 >> 1)) * 8u + (size_t)(sl4 & 1u) * 4u));
				c.v0 = g0.x; c.v1 = g0.y; c.v2 = g0.z; c.v3 = g0.w;
			}
		}
	};
	int qw_lo = 0;
	auto q_refill = [&](int xx){
		int lo = (xx + 8 - CWV_QWIN) & ~7;
		lo = lo < 0 ? 0 : lo;
		qw_lo = lo;
		__syncthreads();
		if((uint32_t)lo + 8u * lane < qlen + 8u){ uint64_t v; __builtin_memcpy(&v, qseq + lo + 8 * lane, 8); *(uint64_t*)&s_q[8 * lane] = v; }
		__syncthreads();
	};
	TB curB, nxtB, nx2B; TC curC, nxtC;
	int T = y | 3;
	if(!bad){
		q_refill(x);
		fetch_begs(T, curB);
		fetch_codes(T, curB, x + (T - y), curC);
		fetch_begs(T - 64, nxtB);
	}
	bool walking = !bad && x >= 0 && y >= 0;
	while(walking){
		// ---- the tile of rows T - 63 .. T
		__syncthreads();
		if constexpr (W == 8){               // row 4 G + m sits at lane index 4 gi + 3 - m
			tile[(4u * gi + 3u) * STR + sl4] = curC.v0; tile[(4u * gi + 2u) * STR + sl4] = curC.v1;
			tile[(4u * gi + 1u) * STR + sl4] = curC.v2; tile[(4u * gi + 0u) * STR + sl4] = curC.v3;
		} else if constexpr (W == 4){
			tile[(4u * gi + 3u) * STR + 2u * sl4] = curC.v0; tile[(4u * gi + 2u) * STR + 2u * sl4] = curC.v1;
			tile[(4u * gi + 1u) * STR + 2u * sl4] = curC.v2; tile[(4u * gi + 0u) * STR + 2u * sl4] = curC.v3;
			tile[(4u * gi + 3u) * STR + 2u * sl4 + 1u] = curC.v4; tile[(4u * gi + 2u) * STR + 2u * sl4 + 1u] = curC.v5;
			tile[(4u * gi + 1u) * STR + 2u * sl4 + 1u] = curC.v6; tile[(4u * gi + 0u) * STR + 2u * sl4 + 1u] = curC.v7;
		} else {
			const uint32_t ra = 4u * gi + 3u - 2u * (sl4 & 1u), co = 2u * (sl4 >> 1);
			tile[ra * STR + co] = curC.v0; tile[ra * STR + co + 1u] = curC.v1;
			tile[(ra - 1u) * STR + co] = curC.v2; tile[(ra - 1u) * STR + co + 1u] = curC.v3;
		}
		if(sl4 == 0u) s_b0[gi] = curC.b0;
		const int bc = curB.bc, bp = curB.bp;
		const uint32_t tbs = curB.tb;
		__syncthreads();
		const int b0 = s_b0[lane >> 2];
		int k0 = T - y;
		// the next tile's codes (their band offsets arrived a tile ago) and the band offsets of the one after
		if(T - 64 >= 0){ fetch_codes(T - 64, nxtB, x + (T - y) - 64, nxtC); fetch_begs(T - 128, nx2B); }
		// the lane's row as four 32-bit planes in CELL order (bit c = window cell c = band cell 8 b0 + c): the bytes of a plane
		// gathered last block first, then all 32 bits reversed (inside a block cell k is bit 7 - k)
		uint32_t RM, RD = 0, RR, RO = 0;
		uint32_t X0 = 0, X1 = 0;                                     // FMT 1
		uint64_t E = 0;                                              // FMT 1: D, Od of the window's 32 cells, two bits a cell
		{
			const uint32_t *mr = &tile[lane * STR];
			if constexpr (W == 8 && FMT == 1){
				const uint32_t d0 = mr[0], d1 = mr[1], d2 = mr[2], d3 = mr[3];
				auto plane = [&](uint32_t j) -> uint32_t {
					const uint32_t lo = __builtin_amdgcn_perm(d2, d3, 0x0c0c0000u | ((4u + j) << 8) | j);
					const uint32_t hi = __builtin_amdgcn_perm(d0, d1, ((4u + j) << 24) | (j << 16) | 0x0c0cu);
					return __builtin_bitreverse32(lo | hi);
				};
				RM = plane(0); RR = plane(1);
				X0 = __builtin_bitreverse32(__builtin_amdgcn_perm(d0, d1, 0x07060302u));      // cells 0 .. 7 from d0's fields (low 16 bits), 8 .. 15 from d1's
				X1 = __builtin_bitreverse32(__builtin_amdgcn_perm(d2, d3, 0x07060302u));
			} else if constexpr (W == 8){
				const uint32_t d0 = mr[0], d1 = mr[1], d2 = mr[2], d3 = mr[3];
				auto plane = [&](uint32_t j) -> uint32_t {
					const uint32_t lo = __builtin_amdgcn_perm(d2, d3, 0x0c0c0000u | ((4u + j) << 8) | j);
					const uint32_t hi = __builtin_amdgcn_perm(d0, d1, ((4u + j) << 24) | (j << 16) | 0x0c0cu);
					return __builtin_bitreverse32(lo | hi);
				};
				RM = plane(0); RD = plane(1); RR = plane(2); RO = plane(3);
			} else if constexpr (W == 4){
				// eight blocks of four cells: nibble j of block b goes to nibble 7 - b, then all 32 bits reversed (cell k is bit 3 - k)
				uint32_t xm = 0, xd = 0, xr = 0, xo = 0;
#pragma unroll
				for(int b = 0; b < 8; b++){
					const uint32_t d = mr[b];
					xm |= (d & 0xFu) << (4 * (7 - b)); xd |= ((d >> 4) & 0xFu) << (4 * (7 - b));
					xr |= ((d >> 8) & 0xFu) << (4 * (7 - b)); xo |= ((d >> 12) & 0xFu) << (4 * (7 - b));
				}
				RM = __builtin_bitreverse32(xm); RD = __builtin_bitreverse32(xd); RR = __builtin_bitreverse32(xr); RO = __builtin_bitreverse32(xo);
			} else {
				// two blocks of sixteen cells: dword 0 = M | D << 16, dword 1 = R | Od << 16, cell k at bit 15 - k
				const uint32_t a0 = mr[0], a1 = mr[1], c0 = mr[2], c1 = mr[3];
				RM = __builtin_bitreverse32(__builtin_amdgcn_perm(a0, c0, 0x05040100u)); RD = __builtin_bitreverse32(__builtin_amdgcn_perm(a0, c0, 0x07060302u));
				RR = __builtin_bitreverse32(__builtin_amdgcn_perm(a1, c1, 0x05040100u)); RO = __builtin_bitreverse32(__builtin_amdgcn_perm(a1, c1, 0x07060302u));
			}
		}
		const int cb = bc + W * b0;                                     // column of window cell 0
		if constexpr (FMT == 1){
			// The two-bit fields decoded once per tile, all 32 cells at a time (bit 2 c: D, the field is 0; bit 2 c + 1: Od, the field is -gapo), where a
			// step of the walk used to decode its own cell -- a quarter of the vector instructions of a step, and the walk is bound by them now
			auto fields = [&](uint32_t X) -> uint32_t {
				const uint32_t z = ~(X | (X >> 1)) & 0x55555555u, y = X ^ (NGOS * 0x55555555u);
				return z | ((~(y | (y >> 1)) & 0x55555555u) << 1);
			};
			E = (uint64_t)fields(X0) | ((uint64_t)fields(X1) << 32);
			// the literal cell: window cell 0 when it is query column 0 of a row whose band starts there (bit 0: Od, bit 1: D after the reversal)
			if(cb == 0 && bc == 0) E = (E & ~3ull) | (uint64_t)(((X0 >> 1) & 1u) | ((X0 & 1u) << 1));
		}
		if(T - (int)lane < 0) RM = 0u;                                  // rows above the target: never a match (the walk ends before them)
		// prior_match is dropped at the first column of the previous row's band (bsalign.h:3761-3764): that cell is taken out of
		// the M plane and left to the literal step
		const uint32_t cpm = (uint32_t)(bp - cb);
		if(cpm < 32u && bp != 0) RM &= ~(1u << cpm);                    // (column 0 keeps prior_match: `... && qb`)
		const int shK = 31 + (int)lane + cb;                            // 31 - c = shK - xs for the lane's window cell c = xs - lane - cb
		if(x + k0 - 63 < qw_lo && qw_lo > 0) q_refill(x);               // the query bases of all 64 cells are in the window
		int qKr = (int)lane + qw_lo;                                    // index into s_q = xs - qKr
		uint32_t RMe = prior ? RM : 0u;                                 // the M plane as the walk sees it: empty until prior_match holds
		while(true){
			const int xs = x + k0;
			const uint32_t sh = (uint32_t)(shK - xs);                     // 31 - c; c < 32 <=> sh < 32
			const uint32_t qb = (uint32_t)s_q[xs - qKr];
			const uint32_t nei = qb != tbs ? 1u : 0u;
			const uint32_t c = (31u - sh) & 31u;
			uint32_t info;                                                          // D, Od of the lane's cell
			if constexpr (FMT == 1) info = (uint32_t)(E >> (2u * c)) & 3u;
			else info = ((RD >> c) & 1u) | (((RO >> c) & 1u) << 1);
			// Where the run of matches from lane k0 ends: each lane's verdict in the sign bit of one word (the walk passes the cell: set), one ballot.
			// Lanes below k0 are behind the walk; lane k0 under an open deletion run passes only where Od closes the run (bsalign.h:3789-3797).
			uint32_t g = sh < 32u ? RMe << (sh & 31u) : 0u;
			const uint32_t od = sh < 32u ? info << 30 : 0u;
			const uint32_t gk = g & (od | ~dl31);
			g = (int)lane == k0 ? gk : g;
			g = (int)lane < k0 ? 0x80000000u : g;
			const uint64_t stopm = __ballot((int)g >= 0);               // (the lanes-below-k0 mask on the scalar side instead -- one vector compare, two scalar operations -- was measured slower: 8.85 against 8.78 ms, two-piece 17.1 against 16.4: the walk is as close to the scalar unit's rate as to the vector units')
			dl31 &= ~(uint32_t)__builtin_amdgcn_readlane((int)od, k0);      // the run ends at this cell, which is then an ordinary one
			const int k = stopm ? (int)__builtin_ctzll(stopm) : 64;
			const int n = k - k0;
			// n match / mismatch columns (possibly none): the count goes to the token key, the mismatches to the lanes' own counts (vector work on a
			// unit that had room; masks, popcounts and additions of the scalar unit were a sixth of the walk's instructions when it was bound by them)
			vmis += (uint32_t)((int)lane - k0) < (uint32_t)n ? nei : 0u;
			key += (uint32_t)n;
			x -= n; y -= n;
			if(k == 64) break;
			if(x < 0 || y < 0){ walking = false; break; }
			// ---- the cell at lane k.  The common cases first: an open deletion run goes on, a deletion opens (not M and D set, in
			// both tie orders), an insertion whose opening cell lies inside the window
			const uint32_t shk = (uint32_t)__builtin_amdgcn_readlane((int)sh, k);
			if(shk < plim){
				if(dl31){
					emit(2u, 1u); y--; k0 = k + 1;
					if(k0 > 63) break;
					continue;
				}
				const uint32_t ik = (uint32_t)__builtin_amdgcn_readlane((int)info, k);
				if(ik & 1u){
					emit(2u, 1u); y--; dl31 = 0x80000000u; k0 = k + 1;
					if(k0 > 63) break;
					continue;
				}
				const uint32_t ck = 31u - shk;
				if(x > 0 && ck != (uint32_t)__builtin_amdgcn_readlane((int)cpm, k)){
					const uint32_t cand = (uint32_t)__builtin_amdgcn_readlane((int)RR, k) & ((1u << ck) - 1u);
					if(cand){
						const int sz = (int)ck - (31 - (int)__builtin_clz(cand));
						emit(1u, (uint32_t)sz);
						x -= sz; rs.ins += sz; k0 = k;
						if(x + k0 - 63 < qw_lo && qw_lo > 0){ q_refill(x); qKr = (int)lane + qw_lo; }
						continue;
					}
				}
			}
			// ---- everything else, literally (first cell, prior_match column, cells outside the window, insertions leaving it, column 0)
			const uint32_t pk = (uint32_t)(x - __builtin_amdgcn_readlane(bc, k));
			if(pk >= (uint32_t)bw){ bad = true; walking = false; break; }
			const int bpk = __builtin_amdgcn_readlane(bp, k);
			const uint32_t yb = pk / W, kk = pk % W, bit = 1u << (W - 1 - kk);
			struct Code { uint32_t m, d, r, o; };
			auto code_at = [&](uint32_t blk) -> Code {                       // the four planes of block blk of row y, plain loads (uniform)
				const uint32_t *rp = codes + bsa_code_off((uint32_t)y, blk, CW);
				const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rp[0]);
				Code cc;
				if constexpr (FMT == 1){
					// (d, o: only the bit of cell kk of block yb is ever looked at -- the walker's own cell)
					cc.m = w0 & 0xFFu; cc.r = (w0 >> 8) & 0xFFu;
					const uint32_t fld = (w0 >> (30u - 2u * kk)) & 3u;
					if(pk == 0u && x == 0) { cc.d = (fld & 1u) ? bit : 0u; cc.o = (fld & 2u) ? bit : 0u; }      // the literal cell (bsa_common.h)
					else { cc.d = (fld == 0u) ? bit : 0u; cc.o = (fld == NGO) ? bit : 0u; }
				} else
				if constexpr (W == 4){ cc.m = w0 & 0xFu; cc.d = (w0 >> 4) & 0xFu; cc.r = (w0 >> 8) & 0xFu; cc.o = (w0 >> 12) & 0xFu; }
				else if constexpr (W == 8){ cc.m = w0 & 0xFFu; cc.d = (w0 >> 8) & 0xFFu; cc.r = (w0 >> 16) & 0xFFu; cc.o = w0 >> 24; }
				else { const uint32_t w1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rp[1]); cc.m = w0 & 0xFFFFu; cc.d = w0 >> 16; cc.r = w1 & 0xFFFFu; cc.o = w1 >> 16; }
				return cc;
			};
			const Code wck = code_at(yb);
			if(dl31){
				if(wck.o & bit) dl31 = 0u;
				else { emit(2u, 1u); y--; k0 = k + 1; if(k0 > 63) break; continue; }
			}
			const bool pmatch = prior && !(x == bpk && x != 0);
			const bool fm = (wck.m & bit) != 0u, fd = (wck.d & bit) != 0u;
			int bt;
			if(pmatch) bt = fm ? 0 : fd ? 2 : 1;
			else bt = fd ? 2 : fm ? 0 : 1;
			prior = 1; plim = 32u; RMe = RM;
			if(bt == 0){
				vmis += (int)lane == k ? nei : 0u;
				emit(0u, 1u);
				x--; y--; k0 = k + 1;
			} else if(bt == 1){
				if(x <= 0){ emit(1u, 1u); x--; rs.ins++; }
				else {
					int sz = 0;
					const uint32_t cand = wck.r & ~((bit << 1) - 1u) & FULL;
					if(cand) sz = (int)__builtin_ctz(cand) - (int)(W - 1 - kk);
					else {
						int left = (int)kk;
						for(int yy = (int)yb - 1; yy >= 0 && sz == 0; yy--){
							const uint32_t r2 = code_at((uint32_t)yy).r & FULL;
							if(r2) sz = left + 1 + (int)__builtin_ctz(r2);
							else left += W;
						}
						if(sz == 0){ bad = true; walking = false; break; }  // the reference's scan finds no length either
					}
					emit(1u, (uint32_t)sz);
					x -= sz; rs.ins += sz;
				}
				k0 = k;
				if(x >= 0 && x + k0 - 63 < qw_lo && qw_lo > 0){ q_refill(x); qKr = (int)lane + qw_lo; }
			} else {
				emit(2u, 1u);
				y--; dl31 = 0x80000000u; k0 = k + 1;
			}
			if(k0 > 63) break;
		}
		if(x < 0 || y < 0) walking = false;
		T -= 64;
		curB = nxtB; curC = nxtC; nxtB = nx2B;
	}
	if(!bad && dl31 && y < 0 && (!lin || x >= bw)) bad = true;        // a deletion run that reached row -1: see the general step of the LDS kernel
	if(!bad){
		rs.qb = x; rs.tb = y;
		{
			// rs.ins holds the inserted columns of the walk: the other totals follow from its two ends, the mismatches from the lanes' counts
			const int mcols = (x_start - x) - rs.ins;
			uint32_t t = vmis;
			t += (uint32_t)__shfl_xor((int)t, 32); t += (uint32_t)__shfl_xor((int)t, 16); t += (uint32_t)__shfl_xor((int)t, 8);
			t += (uint32_t)__shfl_xor((int)t, 4); t += (uint32_t)__shfl_xor((int)t, 2); t += (uint32_t)__shfl_xor((int)t, 1);
			rs.mis = __builtin_amdgcn_readfirstlane((int)t); rs.mat = mcols - rs.mis;
			rs.del = (y_start - y) - mcols;
		}
		if(type != BSA_MODE_OVERLAP){
			uint32_t op = 0, sz = 0;
			if(rs.qb >= 0){ op = 1; sz = (uint32_t)rs.qb + 1u; rs.ins += (int)sz; rs.qb = -1; }
			else if(rs.tb >= 0){ op = 2; sz = (uint32_t)rs.tb + 1u; rs.del += (int)sz; rs.tb = -1; }
			if(sz) emit(op, sz);
		}
		cig_finish();
		rs.qb++; rs.tb++;
		rs.aln = rs.mat + rs.mis + rs.ins + rs.del;
	} else {
		if(lane == 0) atomicOr(&a.status[pair], BSA_ST_TRACE);
		rs.qb = rs.qe = rs.tb = rs.te = 0; rs.mat = rs.mis = rs.ins = rs.del = rs.aln = 0;
		ncig = 0;
	}
	if(lane == 0){ out[pair] = rs; cig_cnt[ppos] = ncig; }
}

// plain version: every access is a load (W = 16, and the reference point for the prefetching kernel)
template<int W>
__global__ void __launch_bounds__(64) k_align8_trace_codes_simple(const Align8Args a, bsa_result_t *out, uint32_t *cig_cnt){
	constexpr uint32_t CW = (W >= 8) ? (uint32_t)W / 8u : 1u, RB = 64u * CW;
	constexpr uint32_t FULL = (W == 16) ? 0xFFFFu : ((1u << W) - 1u);
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
	const int *begs = (const int*)(a.rows + a.slot_off[ppos]);
	const uint8_t *rows = (const uint8_t*)begs + bsa_begs_bytes(tlen);
	uint32_t *cig_end = (uint32_t*)(rows + ((size_t)bsa_code_rows(tlen) + BSA_CODE_SPARE_ROWS) * RB);
	const int bw = W * 16;
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
	// flag planes of block y of row r: (M, D, R, Od), each W bits, cell k = bit W-1-k
	struct Code { uint32_t m, d, r, o; };
	auto unpack = [&](uint32_t w0, uint32_t w1) -> Code {
		Code c;
		if(W == 4){ c.m = w0 & 0xF; c.d = (w0 >> 4) & 0xF; c.r = (w0 >> 8) & 0xF; c.o = (w0 >> 12) & 0xF; }
		else if(W == 8){ c.m = w0 & 0xFF; c.d = (w0 >> 8) & 0xFF; c.r = (w0 >> 16) & 0xFF; c.o = w0 >> 24; }
		else { c.m = w0 & 0xFFFF; c.d = w0 >> 16; c.r = w1 & 0xFFFF; c.o = w1 >> 16; }
		return c;
	};
	auto load_code = [&](int r, uint32_t y) -> Code {
		const uint32_t *rp = (const uint32_t*)rows + bsa_code_off((uint32_t)r, y, CW);
		return unpack(rp[0], (CW > 1) ? rp[CW - 1] : 0u);
	};
	bool bad = false;
	const int type = a.mode & 3;
	const bool lin = a.gapo1 == 0;                                 // linear gaps (piecewise 0)
	if(type == BSA_MODE_GLOBAL){
		rs.score = begs[tlen + 1];
		if(rs.score == (int)0x80000000u) bad = true;               // band never reached the query end (bsalign.h:4034)
		rs.qe = (int)qlen - 1; rs.te = (int)tlen - 1;
	} else codes_end_cell<W>(rows, RB, qlen, tlen, a.ref_bw, rs.score, rs.qe, rs.te);
	if(rs.qe < begs[rs.te + 1] || rs.qe >= begs[rs.te + 1] + bw) bad = true;
	rs.qb = rs.qe; rs.qe++;
	rs.tb = rs.te; rs.te++;
	int prior_match = 0;
	uint32_t cg = 0;
	// band offsets: beg_c = row tb, beg_p = row tb-1 (begs[r + 1] = offset of row r, begs[0] = 0 for row -1)
	int beg_c = begs[rs.tb + 1], beg_p = begs[rs.tb], beg_pp = (rs.tb >= 1) ? begs[rs.tb - 1] : 0;
	auto row_up = [&](){ rs.tb--; beg_c = beg_p; beg_p = beg_pp; beg_pp = (rs.tb >= 1) ? begs[rs.tb - 1] : 0; };
	while(!bad){
		if(rs.qb < 0 || rs.tb < 0) break;
		if(rs.qb == beg_p && rs.qb) prior_match = 0;                // bsalign.h:3761-3764
		const int p = rs.qb - beg_c;
		if(p < 0 || p >= bw){ bad = true; break; }
		const uint32_t y = (uint32_t)p / W, k = (uint32_t)p % W, bit = 1u << (W - 1 - k);
		const Code c = load_code(rs.tb, y);
		const bool fm = (c.m & bit) != 0u, fd = (c.d & bit) != 0u;
		int bt;                                                       // 0 M, 1 I, 2 D
		if(prior_match) bt = fm ? 0 : fd ? 2 : 1;
		else bt = fd ? 2 : fm ? 0 : 1;
		prior_match = 1;
		if(bt == 0){
			const int qbase = qbase_at(rs.qb), tbase = tbase_at(rs.tb);
			if(qbase == tbase) rs.mat++; else rs.mis++;
			rs.qb--; rs.aln++;
			row_up();
			cg = cig_add(cg, 0, 1);
		} else if(bt == 1){
			if(rs.qb <= 0){
				cg = cig_add(cg, 1, 1);
				rs.qb--; rs.ins++; rs.aln++;
			} else {
				// nearest cell to the left with R set: cells left of k are the bits above `bit`
				int sz = 0;
				uint32_t cand = c.r & ~((bit << 1) - 1u) & FULL;
				if(cand){ sz = (int)(__builtin_ctz(cand) - (W - 1 - k)); }
				else {
					// continue in the blocks to the left (rare: an insertion crossing a block boundary)
					int left = (int)k;                                    // cells already scanned without success
					for(int yy = (int)y - 1; yy >= 0 && sz == 0; yy--){
						const Code c2 = load_code(rs.tb, (uint32_t)yy);
						if(c2.r & FULL){ sz = left + 1 + (int)__builtin_ctz(c2.r & FULL); }
						else left += W;
					}
					if(sz == 0){ bad = true; break; }                     // the reference's scan finds no length either: it never terminates
				}
				cg = cig_add(cg, 1, (uint32_t)sz);
				rs.qb -= sz; rs.ins += sz; rs.aln += sz;
			}
		} else {
			// deletion run: up the column qb until a row whose Od bit is set; row -1 always ends it
			int len = 1;
			for(;;){
				const int r = rs.tb - len;
				if(r == -1){ if(!lin || rs.qb >= bw) bad = true; break; }      // see the general step of the LDS kernel
				const int pr = rs.qb - begs[r + 1];
				if(pr < 0 || pr >= bw){ bad = true; break; }
				const Code c2 = load_code(r, (uint32_t)pr / W);
				if(c2.o & (1u << (W - 1 - (uint32_t)pr % W))) break;
				len++;
			}
			if(bad) break;
			cg = cig_add(cg, 2, (uint32_t)len);
			rs.del += len; rs.aln += len;
			rs.tb -= len;
			beg_c = begs[rs.tb + 1]; beg_p = (rs.tb >= 0) ? begs[rs.tb] : 0; beg_pp = (rs.tb >= 1) ? begs[rs.tb - 1] : 0;
		}
	}
	if(!bad){
		if(type == BSA_MODE_OVERLAP){ if(cg) cig_push(cg); }         // overlap: the alignment simply starts here (bsalign.h:3822-3826)
		else {
			uint32_t op = 0, sz = 0;      // global / extend: what is left at the top becomes a leading I / D (bsalign.h:3827-3842)
			if(rs.qb >= 0){ op = 1; sz = (uint32_t)rs.qb + 1u; rs.ins += (int)sz; rs.qb = -1; }
			else if(rs.tb >= 0){ op = 2; sz = (uint32_t)rs.tb + 1u; rs.del += (int)sz; rs.tb = -1; }
			rs.aln += (int)sz;
			cg = cig_add(cg, op, sz);
			if(cg) cig_push(cg);
		}
		rs.qb++; rs.tb++;
	}
	if(bad){
		atomicOr(&a.status[pair], BSA_ST_TRACE);
		rs.qb = rs.qe = rs.tb = rs.te = 0; rs.mat = rs.mis = rs.ins = rs.del = rs.aln = 0;
		ncig = 0;
	}
	out[pair] = rs;
	cig_cnt[ppos] = ncig;
}

// two-piece gaps (8 bits per cell, bsa_common.h; bandwidth 128): the plain walker with the nine facts of a cell
// (W = 4, 8, 16: a reference block's eight planes of W bits -- A, D, D2, B, R1, R2, Od1, Od2 -- in W / 4 dwords, plane j at bit j W of them)
template<int W>
__global__ void __launch_bounds__(64) k_align8_trace_codes2(const Align8Args a, bsa_result_t *out, uint32_t *cig_cnt){
	static_assert(W == 4 || W == 8 || W == 16, "bandwidth 64, 128, 256");
	constexpr uint32_t CW = (uint32_t)W / 4u, RB = 64u * CW, FULLW = (W == 16) ? 0xFFFFu : ((1u << W) - 1u);
	struct Blk { uint32_t w[CW]; };
	auto plane = [](const Blk &b, int j) -> uint32_t { return (b.w[(j * W) >> 5] >> ((j * W) & 31)) & FULLW; };
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
	const int *begs = (const int*)(a.rows + a.slot_off[ppos]);
	const uint8_t *rows = (const uint8_t*)begs + bsa_begs_bytes(tlen);
	uint32_t *cig_end = (uint32_t*)(rows + ((size_t)bsa_code_rows(tlen) + BSA_CODE_SPARE_ROWS) * RB);
	const int bw = W * 16;
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
	auto load_code = [&](int r, uint32_t y) -> Blk {
		Blk b; const uint32_t *p = (const uint32_t*)rows + bsa_code_off((uint32_t)r, y, CW);
#pragma unroll
		for(uint32_t d = 0; d < CW; d++) b.w[d] = p[d];
		return b;
	};
	bool bad = false;
	const int type = a.mode & 3;
	if(type == BSA_MODE_GLOBAL){
		rs.score = begs[tlen + 1];
		if(rs.score == (int)0x80000000u) bad = true;           // band never reached the query end (bsalign.h:4034)
		rs.qe = (int)qlen - 1; rs.te = (int)tlen - 1;
	} else codes_end_cell<W>(rows, RB, qlen, tlen, a.ref_bw, rs.score, rs.qe, rs.te);      // overlap / extend: the best end cell (bsalign.h:4023-4046)
	if(rs.qe < begs[rs.te + 1] || rs.qe >= begs[rs.te + 1] + bw) bad = true;
	rs.qb = rs.qe; rs.qe++;
	rs.tb = rs.te; rs.te++;
	int prior_match = 0;
	uint32_t cg = 0;
	int beg_c = begs[rs.tb + 1], beg_p = begs[rs.tb], beg_pp = (rs.tb >= 1) ? begs[rs.tb - 1] : 0;
	auto row_up = [&](){ rs.tb--; beg_c = beg_p; beg_p = beg_pp; beg_pp = (rs.tb >= 1) ? begs[rs.tb - 1] : 0; };
	while(!bad){
		if(rs.qb < 0 || rs.tb < 0) break;
		if(rs.qb == beg_p && rs.qb) prior_match = 0;                // bsalign.h:3761-3764
		const int p = rs.qb - beg_c;
		if(p < 0 || p >= bw){ bad = true; break; }
		const uint32_t y = (uint32_t)p / W, k = (uint32_t)p % W, bit = 1u << (W - 1 - k);
		const Blk c = load_code(rs.tb, y);
		const bool fA = (plane(c, 0) & bit) != 0u, fD = (plane(c, 1) & bit) != 0u, fD2 = (plane(c, 2) & bit) != 0u, fB = (plane(c, 3) & bit) != 0u;
		const bool fM = (fD || fD2) ? fA : (fA && !fB);
		const int d = fD ? 1 : fD2 ? 2 : 0;                          // backcal_cell, bsalign.h:3679-3701
		int bt;                                                       // 0 M, 1 I, 2 D
		if(prior_match) bt = fM ? 0 : d ? 2 : 1;
		else bt = d ? 2 : fM ? 0 : 1;
		prior_match = 1;
		if(bt == 0){
			const int qbase = qbase_at(rs.qb), tbase = tbase_at(rs.tb);
			if(qbase == tbase) rs.mat++; else rs.mis++;
			rs.qb--; rs.aln++;
			row_up();
			cg = cig_add(cg, 0, 1);
		} else if(bt == 1){
			if(rs.qb <= 0){
				cg = cig_add(cg, 1, 1);
				rs.qb--; rs.ins++; rs.aln++;
			} else {
				// the chains that equal h here (only reached with M = D = D2 = 0): the nearest cell to the left at which one of
				// them was opened (bsalign.h:3798-3814: the smallest length whose cost -- the larger of the two pieces' -- closes the gap)
				const bool ch1 = fB, ch2 = fA == fB;
				auto rplane = [&](const Blk &cc) -> uint32_t { return (ch1 ? plane(cc, 4) : 0u) | (ch2 ? plane(cc, 5) : 0u); };
				int sz = 0;
				Blk hc = c; uint32_t hb = 0;                            // block and bit of the cell the scan stops at
				const uint32_t cand = rplane(c) & ~((bit << 1) - 1u);
				if(cand){ hb = cand & (0u - cand); sz = (int)(__builtin_ctz(cand) - (W - 1 - k)); }
				else {
					int left = (int)k;
					for(int yy = (int)y - 1; yy >= 0 && sz == 0; yy--){
						hc = load_code(rs.tb, (uint32_t)yy);
						const uint32_t r2 = rplane(hc);
						if(r2){ hb = r2 & (0u - r2); sz = left + 1 + (int)__builtin_ctz(r2); }
						else left += W;
					}
					if(sz == 0){ bad = true; break; }                     // the reference's scan finds no length either
				}
				{
					// the reference tests H(x - sz) + max(cost1, cost2) == H(x): the chain that is tight here must also have the
					// larger cost at this length (always true between real DP cells; next to cells that entered the band with
					// synthetic values it can fail, and the reference's scan then finds no length: literal path)
					const int c1 = a.gapo1 + sz * a.gape1, c2 = a.gapo2 + sz * a.gape2;
					const bool h1 = ch1 && (plane(hc, 4) & hb) != 0u, h2 = ch2 && (plane(hc, 5) & hb) != 0u;
					if(!((h1 && c1 >= c2) || (h2 && c2 >= c1))){ bad = true; break; }
				}
				cg = cig_add(cg, 1, (uint32_t)sz);
				rs.qb -= sz; rs.ins += sz; rs.aln += sz;
			}
		} else {
			// deletion run of piece d: up the column until a row whose Od bit of that piece is set (bsalign.h:3730-3760).
			// Not at query column 0: the D / D2 test there compares scores of two frames (the row is re-based at its first cell,
			// bsalign.h:2632-2633), so it can fire where no deletion ends, and the reference's run-length scan, which works on real
			// scores, then finds no opening and does not terminate -- the literal path reproduces (and flags) that
			if(rs.qb == 0){ bad = true; break; }
			const int opl = (d == 2) ? 7 : 6;
			int len = 1;
			for(;;){
				const int r = rs.tb - len;
				if(r == -1){ bad = true; break; }                        // row -1: the reference compares real scores there -- literal path
				const int pr = rs.qb - begs[r + 1];
				if(pr < 0 || pr >= bw){ bad = true; break; }
				const Blk c2 = load_code(r, (uint32_t)pr / W);
				if(plane(c2, opl) & (1u << (W - 1 - (uint32_t)pr % W))) break;
				len++;
			}
			if(bad) break;
			cg = cig_add(cg, 2, (uint32_t)len);
			rs.del += len; rs.aln += len;
			rs.tb -= len;
			beg_c = begs[rs.tb + 1]; beg_p = (rs.tb >= 0) ? begs[rs.tb] : 0; beg_pp = (rs.tb >= 1) ? begs[rs.tb - 1] : 0;
		}
	}
	if(!bad){
		if(type == BSA_MODE_OVERLAP){ if(cg) cig_push(cg); }     // overlap: the alignment simply starts here (bsalign.h:3822-3826)
		else {
			uint32_t op = 0, sz = 0;      // global / extend: what is left at the top becomes a leading I / D (bsalign.h:3827-3842)
			if(rs.qb >= 0){ op = 1; sz = (uint32_t)rs.qb + 1u; rs.ins += (int)sz; rs.qb = -1; }
			else if(rs.tb >= 0){ op = 2; sz = (uint32_t)rs.tb + 1u; rs.del += (int)sz; rs.tb = -1; }
			rs.aln += (int)sz;
			cg = cig_add(cg, op, sz);
			if(cg) cig_push(cg);
		}
		rs.qb++; rs.tb++;
	}
	if(bad){
		atomicOr(&a.status[pair], BSA_ST_TRACE);
		rs.qb = rs.qe = rs.tb = rs.te = 0; rs.mat = rs.mis = rs.ins = rs.del = rs.aln = 0;
		ncig = 0;
	}
	out[pair] = rs;
	cig_cnt[ppos] = ncig;
}

// ---- two-piece gaps, one walk per wave (the scheme of k_align8_trace_codes_wave with the nine facts of a cell) -------
// Row = 16 blocks x two dwords (A | D << 8 | D2 << 16 | B << 24, R1 | R2 << 8 | Od1 << 16 | Od2 << 24); a lane's window of four
// blocks is 32 bytes per row in LDS (stride 9 dwords), turned into eight 32-bit planes in cell order per tile.  A cell is a
// match iff  M = (D | D2) ? A : (A & ~B);  with prior_match set M decides first, so the run mask is A & (D | D2 | ~B) with the
// prior_match column taken out.  The cell that ends a run: a deletion of piece 1 / 2 opens where D / D2 is set, an open run
// of piece d goes on until the row whose Od_d bit is set, everything else is an insertion, decided by the literal rules on
// the planes of that lane (chains, nearest opening cell, cost comparison) when it closes inside the window, else by the
// literal single-cell step with plain loads.
__global__ void __launch_bounds__(64) k_align8_trace_codes2_wave(const Align8Args a, bsa_result_t *out, uint32_t *cig_cnt){
	constexpr int W = 8, bw = 128, STR = 9;
	constexpr uint32_t CW = 2u, RB = 64u * CW;
	__shared__ uint32_t tile[64 * STR];
	__shared__ int s_b0[16];
	__shared__ __attribute__((aligned(8))) uint8_t s_q[CWV_QWIN];
	const uint32_t lane = threadIdx.x;
	const uint32_t ppos = a.first + blockIdx.x;
	const uint32_t pair = a.order[ppos];
	bsa_result_t rs;
	rs.score = 0; rs.qb = rs.qe = rs.tb = rs.te = 0; rs.mat = rs.mis = rs.ins = rs.del = rs.aln = 0;
	if(a.status[pair] != 0u){ if(lane == 0){ out[pair] = rs; cig_cnt[ppos] = 0; } return; }
	const uint32_t qlen = a.qlen[pair], tlen = a.tlen[pair];
	const uint8_t *qseq = a.qst + a.qpoff[pair];
	const uint8_t *tseq = a.tst + a.tpoff[pair];
	const int *begs = (const int*)(a.rows + a.slot_off[ppos]);
	const uint8_t *rows = (const uint8_t*)begs + bsa_begs_bytes(tlen);
	const uint32_t *codes = (const uint32_t*)rows;
	uint32_t *cig_end = (uint32_t*)(rows + ((size_t)bsa_code_rows(tlen) + BSA_CODE_SPARE_ROWS) * RB);
	const int go1 = a.gapo1, ge1 = a.gape1, go2 = a.gapo2, ge2 = a.gape2;
	// CIGAR.  The walker does not merge runs step by step (that logic was a third of the scalar instructions of a step): it records
	// one TOKEN per gap event -- (matches / mismatches since the last event, op, length), one token per lane -- and a gap that simply
	// goes on (no match in between, same op: the rows of a deletion run, consecutive insertions) extends its own token, so that
	// neighbouring tokens never carry the same op.  Every 64 tokens (the last one stays: it may still grow) the lanes turn their
	// tokens into words side by side: an M word when the run is not empty, then the gap's word, positions from two prefix popcounts
	// (word m at cig_end - (m + 1)).  The word sequence is the reference's run-length merge (bsalign.h:409-417) of the same op stream.
	uint32_t ncig = 0, tokN = 0, tokB = 0;            // words written; the lane's token: run length, len << 2 | op (op 0: none)
	// (uniform) tokens held; key = op of the last token << 28 | match / mismatch columns since it (k_align8_trace_codes_wave)
	uint32_t ntok = 0, key = 0;
	constexpr uint32_t KEYM = 0x0FFFFFFFu;
	auto tok_flush = [&](uint32_t cnt){
		const bool in = lane < cnt;
		tokN &= KEYM;
		const bool hasA = in && tokN != 0u, hasB = in && (tokB & 3u) != 0u;
		const uint64_t mA = __ballot(hasA), mB = __ballot(hasB);
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
	auto cig_finish = [&](){                          // the matches behind the last event, then everything out
		if(key & KEYM){
			if(ntok == 64u){ tok_flush(64u); ntok = 0u; }
			if(lane == ntok){ tokN = key; tokB = 0u; }
			ntok++; key = 0u;
		}
		tok_flush(ntok); ntok = 0u;
	};
	bool bad = false;
	const int type = a.mode & 3;
	if(type == BSA_MODE_GLOBAL){
		rs.score = begs[tlen + 1];
		if(rs.score == (int)0x80000000u) bad = true;               // band never reached the query end (bsalign.h:4034)
		rs.qe = (int)qlen - 1; rs.te = (int)tlen - 1;
	} else {
		// overlap / extend: the best end cell (bsalign.h:4023-4046), worked out by one lane from the end record behind the code rows
		int sc = 0, qe = 0, te = 0;
		if(lane == 0) codes_end_cell<8>((const uint8_t*)codes, 64u * CW, qlen, tlen, a.ref_bw, sc, qe, te);
		rs.score = __shfl(sc, 0); rs.qe = __shfl(qe, 0); rs.te = __shfl(te, 0);
	}
	{ const int lb = begs[rs.te + 1]; if(rs.qe < lb || rs.qe >= lb + bw) bad = true; }
	int x = rs.qe, y = rs.te;
	const int x_start = x, y_start = y;
	rs.qe++; rs.te++;
	// prior: prior_match of the reference (0 only before the first step).  An open deletion run is held as the Od bit of its piece in the lanes' od
	// word (piece 1: bit 31, piece 2: bit 30; 0: no run) next to its complement in the sign bit (ndl), the forms the lanes' tests use (k_align8_trace_codes_wave)
	int prior = 0;
	constexpr uint32_t DL1 = 0x80000000u, DL2 = 0x40000000u;
	uint32_t dl = 0u, ndl = 0x80000000u, plim = 0u;                 // plim: 32 once prior_match holds
	uint32_t vmis = 0;                                              // this lane's mismatch count (summed over the wave at the end)
	struct TB { int bc, bp; uint32_t tb; };
	struct TC { uint32_t v0, v1, v2, v3, v4, v5, v6, v7; int b0; };    // rows 0..3 of one block, two dwords each
	auto fetch_begs = [&](int T, TB &t){
		const int r = T - (int)lane;
		if(r >= 0){ t.bc = begs[r + 1]; t.bp = begs[r]; t.tb = (uint32_t)tseq[r]; }
		else { t.bc = 0; t.bp = 0; t.tb = 0xffu; }
	};
	const uint32_t gi = lane >> 2, sl4 = lane & 3u;
	auto fetch_codes = [&](int T, const TB &tb_, int xT, TC &c){
		const int G = (T >> 2) - (int)gi;
		const int bcg = __shfl(tb_.bc, (int)(gi * 4u));
		int pp = (xT - 4 * (int)gi) - bcg;
		pp = pp < 0 ? 0 : pp > bw - 1 ? bw - 1 : pp;
		int b0 = (pp >> 3) - 1;
		b0 = b0 < 0 ? 0 : b0 > 12 ? 12 : b0;
		c.b0 = b0;
		if(G >= 0){
			const uint4 *gp = (const uint4*)(codes + ((size_t)G * 64u + (size_t)((uint32_t)b0 + sl4) * 4u) * CW);     // rows 0, 1 | rows 2, 3 of the block
			const uint4 g0 = gp[0], g1 = gp[1];
			c.v0 = g0.x; c.v1 = g0.y; c.v2 = g0.z; c.v3 = g0.w; c.v4 = g1.x; c.v5 = g1.y; c.v6 = g1.z; c.v7 = g1.w;
		} else { c.v0 = c.v1 = c.v2 = c.v3 = c.v4 = c.v5 = c.v6 = c.v7 = 0u; }
	};
	int qw_lo = 0;
	auto q_refill = [&](int xx){
		int lo = (xx + 8 - CWV_QWIN) & ~7;
		lo = lo < 0 ? 0 : lo;
		qw_lo = lo;
		__syncthreads();
		if((uint32_t)lo + 8u * lane < qlen + 8u){ uint64_t v; __builtin_memcpy(&v, qseq + lo + 8 * lane, 8); *(uint64_t*)&s_q[8 * lane] = v; }
		__syncthreads();
	};
	TB curB, nxtB, nx2B; TC curC, nxtC;
	int T = y | 3;
	if(!bad){
		q_refill(x);
		fetch_begs(T, curB);
		fetch_codes(T, curB, x + (T - y), curC);
		fetch_begs(T - 64, nxtB);
	}
	bool walking = !bad && x >= 0 && y >= 0;
	while(walking){
		__syncthreads();
		{
			uint32_t *t3 = &tile[(4u * gi + 3u) * STR + 2u * sl4], *t2 = &tile[(4u * gi + 2u) * STR + 2u * sl4];
			uint32_t *t1 = &tile[(4u * gi + 1u) * STR + 2u * sl4], *t0 = &tile[(4u * gi + 0u) * STR + 2u * sl4];
			t3[0] = curC.v0; t3[1] = curC.v1; t2[0] = curC.v2; t2[1] = curC.v3;       // row 4G + m sits at lane index 4 gi + 3 - m
			t1[0] = curC.v4; t1[1] = curC.v5; t0[0] = curC.v6; t0[1] = curC.v7;
		}
		if(sl4 == 0u) s_b0[gi] = curC.b0;
		const int bc = curB.bc, bp = curB.bp;
		const uint32_t tbs = curB.tb;
		__syncthreads();
		const int b0 = s_b0[lane >> 2];
		int k0 = T - y;
		if(T - 64 >= 0){ fetch_codes(T - 64, nxtB, x + (T - y) - 64, nxtC); fetch_begs(T - 128, nx2B); }
		uint32_t PA, PD, PD2, PB, PR1, PR2, PO1, PO2;                  // the lane's row: eight planes in cell order
		{
			const uint32_t *mr = &tile[lane * STR];
			const uint32_t a0 = mr[0], a1 = mr[2], a2 = mr[4], a3 = mr[6], r0 = mr[1], r1 = mr[3], r2 = mr[5], r3 = mr[7];
			auto plane = [&](uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3, uint32_t j) -> uint32_t {
				const uint32_t lo = __builtin_amdgcn_perm(d2, d3, 0x0c0c0000u | ((4u + j) << 8) | j);
				const uint32_t hi = __builtin_amdgcn_perm(d0, d1, ((4u + j) << 24) | (j << 16) | 0x0c0cu);
				return __builtin_bitreverse32(lo | hi);
			};
			PA = plane(a0, a1, a2, a3, 0); PD = plane(a0, a1, a2, a3, 1); PD2 = plane(a0, a1, a2, a3, 2); PB = plane(a0, a1, a2, a3, 3);
			PR1 = plane(r0, r1, r2, r3, 0); PR2 = plane(r0, r1, r2, r3, 1); PO1 = plane(r0, r1, r2, r3, 2); PO2 = plane(r0, r1, r2, r3, 3);
		}
		uint32_t RM = PA & (PD | PD2 | ~PB);
		const int cb = bc + 8 * b0;
		if(T - (int)lane < 0) RM = 0u;
		const uint32_t cpm = (uint32_t)(bp - cb);
		if(cpm < 32u && bp != 0) RM &= ~(1u << cpm);                    // prior_match column (bsalign.h:3761-3764): the literal step; column 0 keeps prior_match (`... && qb`)
		const int shK = 31 + (int)lane + cb;
		if(x + k0 - 63 < qw_lo && qw_lo > 0) q_refill(x);
		int qKr = (int)lane + qw_lo;
		uint32_t RMe = prior ? RM : 0u;                                 // the M plane as the walk sees it: empty until prior_match holds
		while(true){
			const int xs = x + k0;
			const uint32_t sh = (uint32_t)(shK - xs);
			const uint32_t qb = (uint32_t)s_q[xs - qKr];
			const uint32_t nei = qb != tbs ? 1u : 0u;
			const uint32_t c = (31u - sh) & 31u;
			const uint32_t info = ((PD >> c) & 1u) | (((PD2 >> c) & 1u) << 1);            // D, D2 of the lane's cell
			// the lanes' verdicts in the sign bit of one word, one ballot (k_align8_trace_codes_wave); lane k0 under an open run passes only where
			// the Od of the run's piece closes it
			uint32_t g = sh < 32u ? RMe << (sh & 31u) : 0u;
			const uint32_t od = sh < 32u ? ((PO1 << (sh & 31u)) & DL1) | (((PO2 << (sh & 31u)) >> 1) & DL2) : 0u;
			const uint32_t odr = od & dl;
			const uint32_t gk = g & (odr | (odr << 1) | ndl);
			g = (int)lane == k0 ? gk : g;
			g = (int)lane < k0 ? 0x80000000u : g;
			const uint64_t stopm = __ballot((int)g >= 0);               // (the lanes-below-k0 mask on the scalar side instead -- one vector compare, two scalar operations -- was measured slower: 8.85 against 8.78 ms, two-piece 17.1 against 16.4: the walk is as close to the scalar unit's rate as to the vector units')
			if((uint32_t)__builtin_amdgcn_readlane((int)od, k0) & dl){ dl = 0u; ndl = 0x80000000u; }      // Od of the run's piece: the run ends at this cell
			const int k = stopm ? (int)__builtin_ctzll(stopm) : 64;
			const int n = k - k0;
			// n match / mismatch columns (possibly none): the count goes to the token key, the mismatches to the lanes' own counts
			vmis += (uint32_t)((int)lane - k0) < (uint32_t)n ? nei : 0u;
			key += (uint32_t)n;
			x -= n; y -= n;
			if(k == 64) break;
			if(x < 0 || y < 0){ walking = false; break; }
			const uint32_t shk = (uint32_t)__builtin_amdgcn_readlane((int)sh, k);
			if(shk < plim){
				if(dl){
					emit(2u, 1u); y--; k0 = k + 1;
					if(k0 > 63) break;
					continue;
				}
				const uint32_t ik = (uint32_t)__builtin_amdgcn_readlane((int)info, k);
				if(ik & 3u){                                                // not M and D or D2 set: a deletion of that piece opens
					if(x == 0){ bad = true; walking = false; break; }         // (at query column 0 the flags cannot tell: k_align8_trace_codes2)
					emit(2u, 1u); y--; dl = (ik & 1u) ? DL1 : DL2; ndl = 0u; k0 = k + 1;
					if(k0 > 63) break;
					continue;
				}
				const uint32_t ck = 31u - shk;
				if(x > 0 && ck != (uint32_t)__builtin_amdgcn_readlane((int)cpm, k)){
					// insertion (M = D = D2 = 0): the chains that equal h, the nearest cell to the left at which one of them opens,
					// the cost comparison of bsalign.h:3798-3814 (k_align8_trace_codes2 states the rules)
					const uint32_t cbit = 1u << ck;
					const bool fA = ((uint32_t)__builtin_amdgcn_readlane((int)PA, k) & cbit) != 0u, fB = ((uint32_t)__builtin_amdgcn_readlane((int)PB, k) & cbit) != 0u;
					const bool ch1 = fB, ch2 = fA == fB;
					const uint32_t r1k = (uint32_t)__builtin_amdgcn_readlane((int)PR1, k), r2k = (uint32_t)__builtin_amdgcn_readlane((int)PR2, k);
					const uint32_t cand = ((ch1 ? r1k : 0u) | (ch2 ? r2k : 0u)) & (cbit - 1u);
					if(cand){
						const uint32_t hp = 31u - (uint32_t)__builtin_clz(cand);
						const int sz = (int)ck - (int)hp;
						const int c1 = go1 + sz * ge1, c2 = go2 + sz * ge2;
						const bool h1 = ch1 && ((r1k >> hp) & 1u), h2 = ch2 && ((r2k >> hp) & 1u);
						if(!((h1 && c1 >= c2) || (h2 && c2 >= c1))){ bad = true; walking = false; break; }
						emit(1u, (uint32_t)sz);
						x -= sz; rs.ins += sz; k0 = k;
						if(x + k0 - 63 < qw_lo && qw_lo > 0){ q_refill(x); qKr = (int)lane + qw_lo; }
						continue;
					}
				}
			}
			// ---- everything else, literally
			const uint32_t pk = (uint32_t)(x - __builtin_amdgcn_readlane(bc, k));
			if(pk >= (uint32_t)bw){ bad = true; walking = false; break; }
			const int bpk = __builtin_amdgcn_readlane(bp, k);
			const int b0k = __builtin_amdgcn_readlane(b0, k);
			const uint32_t yb = pk >> 3, kk = pk & 7u, bit = 1u << (7u - kk);
			auto code_at = [&](uint32_t blk) -> uint2 {
				const uint32_t s_ = blk - (uint32_t)b0k;
				uint32_t v0 = tile[(uint32_t)k * STR + 2u * (s_ & 3u)], v1 = tile[(uint32_t)k * STR + 2u * (s_ & 3u) + 1u];
				asm volatile("" : "+v"(v0), "+v"(v1));
				if(s_ >= 4u){ const uint2 g = *(const uint2*)(codes + bsa_code_off((uint32_t)y, blk, CW)); v0 = g.x; v1 = g.y; }
				uint2 r; r.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)v0); r.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)v1);
				return r;
			};
			const uint2 cc = code_at(yb);
			if(dl){
				if((cc.y >> (dl == DL2 ? 24 : 16)) & bit){ dl = 0u; ndl = 0x80000000u; }
				else { emit(2u, 1u); y--; k0 = k + 1; if(k0 > 63) break; continue; }
			}
			const bool pmatch = prior && !(x == bpk && x != 0);
			const bool fA = (cc.x & bit) != 0u, fD = ((cc.x >> 8) & bit) != 0u, fD2 = ((cc.x >> 16) & bit) != 0u, fB = ((cc.x >> 24) & bit) != 0u;
			const bool fM = (fD || fD2) ? fA : (fA && !fB);
			const int d = fD ? 1 : fD2 ? 2 : 0;
			int bt;
			if(pmatch) bt = fM ? 0 : d ? 2 : 1;
			else bt = d ? 2 : fM ? 0 : 1;
			prior = 1; plim = 32u; RMe = RM;
			if(bt == 0){
				vmis += (int)lane == k ? nei : 0u;
				emit(0u, 1u);
				x--; y--; k0 = k + 1;
			} else if(bt == 1){
				if(x <= 0){ emit(1u, 1u); x--; rs.ins++; }
				else {
					const bool ch1 = fB, ch2 = fA == fB;
					auto rplane = [&](const uint2 &w) -> uint32_t { return ((ch1 ? w.y : 0u) | (ch2 ? (w.y >> 8) : 0u)) & 0xFFu; };
					int sz = 0;
					uint2 hc = cc; uint32_t hb = 0;
					const uint32_t cand = rplane(cc) & ~((bit << 1) - 1u);
					if(cand){ hb = cand & (0u - cand); sz = (int)__builtin_ctz(cand) - (int)(7u - kk); }
					else {
						int left = (int)kk;
						for(int yy = (int)yb - 1; yy >= 0 && sz == 0; yy--){
							hc = code_at((uint32_t)yy);
							const uint32_t r2 = rplane(hc);
							if(r2){ hb = r2 & (0u - r2); sz = left + 1 + (int)__builtin_ctz(r2); }
							else left += W;
						}
						if(sz == 0){ bad = true; walking = false; break; }
					}
					{
						const int c1 = go1 + sz * ge1, c2 = go2 + sz * ge2;
						const bool h1 = ch1 && (hc.y & hb) != 0u, h2 = ch2 && ((hc.y >> 8) & hb) != 0u;
						if(!((h1 && c1 >= c2) || (h2 && c2 >= c1))){ bad = true; walking = false; break; }
					}
					emit(1u, (uint32_t)sz);
					x -= sz; rs.ins += sz;
				}
				k0 = k;
				if(x >= 0 && x + k0 - 63 < qw_lo && qw_lo > 0){ q_refill(x); qKr = (int)lane + qw_lo; }
			} else {
				if(x == 0){ bad = true; walking = false; break; }             // (query column 0: k_align8_trace_codes2)
				emit(2u, 1u);
				y--; dl = d == 1 ? DL1 : DL2; ndl = 0u; k0 = k + 1;
			}
			if(k0 > 63) break;
		}
		if(x < 0 || y < 0) walking = false;
		T -= 64;
		curB = nxtB; curC = nxtC; nxtB = nx2B;
	}
	if(!bad && dl && y < 0) bad = true;                            // a deletion run that reached row -1: the reference compares real scores there -- literal path
	if(!bad){
		rs.qb = x; rs.tb = y;
		{
			// rs.ins holds the inserted columns of the walk: the other totals follow from its two ends, the mismatches from the lanes' counts
			const int mcols = (x_start - x) - rs.ins;
			uint32_t t = vmis;
			t += (uint32_t)__shfl_xor((int)t, 32); t += (uint32_t)__shfl_xor((int)t, 16); t += (uint32_t)__shfl_xor((int)t, 8);
			t += (uint32_t)__shfl_xor((int)t, 4); t += (uint32_t)__shfl_xor((int)t, 2); t += (uint32_t)__shfl_xor((int)t, 1);
			rs.mis = __builtin_amdgcn_readfirstlane((int)t); rs.mat = mcols - rs.mis;
			rs.del = (y_start - y) - mcols;
		}
		if(type != BSA_MODE_OVERLAP){              // global / extend: what is left at the top becomes a leading I / D (bsalign.h:3827-3842)
			uint32_t op = 0, sz = 0;
			if(rs.qb >= 0){ op = 1; sz = (uint32_t)rs.qb + 1u; rs.ins += (int)sz; rs.qb = -1; }
			else if(rs.tb >= 0){ op = 2; sz = (uint32_t)rs.tb + 1u; rs.del += (int)sz; rs.tb = -1; }
			if(sz) emit(op, sz);
		}
		cig_finish();
		rs.qb++; rs.tb++;
		rs.aln = rs.mat + rs.mis + rs.ins + rs.del;
	} else {
		if(lane == 0) atomicOr(&a.status[pair], BSA_ST_TRACE);
		rs.qb = rs.qe = rs.tb = rs.te = 0; rs.mat = rs.mis = rs.ins = rs.del = rs.aln = 0;
		ncig = 0;
	}
	if(lane == 0){ out[pair] = rs; cig_cnt[ppos] = ncig; }
}

template<int W>
static void launch_trace_lds(const Align8Args &a, bsa_result_t *out, uint32_t *cig_cnt, hipStream_t st){
	// Pairs per wave: 32 where the batch is large enough (100 k pairs on MI355X, ms per launch: 64 -> 46, 32 -> 31.0,
	// 16 -> 31.2, 8 -> 57: instruction issue binds at few lanes per wave, the walk's own dependent chain at many); smaller
	// batches take fewer, so that there are still a few waves per SIMD to interleave.
	int dev = 0, cus = 256;
	if(hipGetDevice(&dev) == hipSuccess){
		int v = 0;
		if(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
	}
	const uint32_t want_waves = (uint32_t)cus * 4u * 3u;            // about three waves per SIMD
	uint32_t lanes = 4;
	while(lanes < 32u && a.count / lanes > want_waves) lanes <<= 1;
	const dim3 grid((a.count + lanes - 1) / lanes);
	switch(lanes){
		case 4:  hipLaunchKernelGGL((k_align8_trace_codes_lds<W, 4>), grid, dim3(64), 0, st, a, out, cig_cnt); break;
		case 8:  hipLaunchKernelGGL((k_align8_trace_codes_lds<W, 8>), grid, dim3(64), 0, st, a, out, cig_cnt); break;
		case 16: hipLaunchKernelGGL((k_align8_trace_codes_lds<W, 16>), grid, dim3(64), 0, st, a, out, cig_cnt); break;
		default: hipLaunchKernelGGL((k_align8_trace_codes_lds<W, 32>), grid, dim3(64), 0, st, a, out, cig_cnt); break;
	}
}

// code format 1 (bsa_common.h) is read by the one-walk-per-wave kernel at bandwidth 128 only
bool bsa_align8_trace_reads_do2(const Align8Args &a, int pw){
	const char *se = bsa_env("BSA_ALIGN8_TRACE_SIMPLE"), *we = bsa_env("BSA_ALIGN8_TRACE_WAVE");
	return pw == 1 && a.bw == 128u && !(se && se[0] == '1') && !(we && we[0] == '0');
}

hipError_t bsa_launch_align8_trace_codes(const Align8Args &a, int pw, bsa_result_t *out, uint32_t *cig_cnt, hipStream_t st){
	if(a.count == 0) return hipSuccess;
	if(a.code_fmt == 1u && !bsa_align8_trace_reads_do2(a, pw)) return hipErrorInvalidValue;
	if(pw == 2){
		if(a.bw == 64u){ bsa_last_trace_kernel = "k_align8_trace_codes2"; hipLaunchKernelGGL((k_align8_trace_codes2<4>), dim3((a.count + 63u) / 64u), dim3(64), 0, st, a, out, cig_cnt); return hipGetLastError(); }
		if(a.bw == 256u){ bsa_last_trace_kernel = "k_align8_trace_codes2"; hipLaunchKernelGGL((k_align8_trace_codes2<16>), dim3((a.count + 63u) / 64u), dim3(64), 0, st, a, out, cig_cnt); return hipGetLastError(); }
		if(a.bw != 128u) return hipErrorInvalidValue;
		const char *we = bsa_env("BSA_ALIGN8_TRACE_WAVE");              // =0: the pair-per-lane walker
		if(we && we[0] == '0'){ bsa_last_trace_kernel = "k_align8_trace_codes2"; hipLaunchKernelGGL((k_align8_trace_codes2<8>), dim3((a.count + 63u) / 64u), dim3(64), 0, st, a, out, cig_cnt); }
		else { bsa_last_trace_kernel = "k_align8_trace_codes2_wave"; hipLaunchKernelGGL(k_align8_trace_codes2_wave, dim3(a.count), dim3(64), 0, st, a, out, cig_cnt); }
		return hipGetLastError();
	}
	// BSA_ALIGN8_TRACE_SIMPLE=1: the plain kernel (kept as the reference point)
	const bool simple = [](){ const char *e = bsa_env("BSA_ALIGN8_TRACE_SIMPLE"); return e && e[0] == '1'; }();
	const uint32_t blocks = (a.count + 63u) / 64u;
	bsa_last_trace_kernel = simple ? "k_align8_trace_codes_simple" : (a.bw == 256u) ? "k_align8_trace_codes_simple" : "k_align8_trace_codes_lds";
	// one walk per wave (BSA_ALIGN8_TRACE_WAVE=0: the pair-per-lane kernels)
	const char *we = bsa_env("BSA_ALIGN8_TRACE_WAVE");
	const bool wave = !simple && !(we && we[0] == '0');
	if(wave) bsa_last_trace_kernel = "k_align8_trace_codes_wave";
	switch(a.bw / 16){
		case 4:
			if(simple) hipLaunchKernelGGL((k_align8_trace_codes_simple<4>), dim3(blocks), dim3(64), 0, st, a, out, cig_cnt);
			else if(wave) hipLaunchKernelGGL((k_align8_trace_codes_wave<4>), dim3(a.count), dim3(64), 0, st, a, out, cig_cnt);
			else launch_trace_lds<4>(a, out, cig_cnt, st);
			break;
		case 8:
			if(simple) hipLaunchKernelGGL((k_align8_trace_codes_simple<8>), dim3(blocks), dim3(64), 0, st, a, out, cig_cnt);
			else if(wave && a.code_fmt == 1u) hipLaunchKernelGGL((k_align8_trace_codes_wave<8, 1>), dim3(a.count), dim3(64), 0, st, a, out, cig_cnt);
			else if(wave) hipLaunchKernelGGL((k_align8_trace_codes_wave<8>), dim3(a.count), dim3(64), 0, st, a, out, cig_cnt);
			else launch_trace_lds<8>(a, out, cig_cnt, st);
			break;
		case 16:
			if(wave) hipLaunchKernelGGL((k_align8_trace_codes_wave<16>), dim3(a.count), dim3(64), 0, st, a, out, cig_cnt);
			else hipLaunchKernelGGL((k_align8_trace_codes_simple<16>), dim3(blocks), dim3(64), 0, st, a, out, cig_cnt);
			break;
		default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}
