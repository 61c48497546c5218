// bsa_align8_codes.hip -- traceback of the compact (4-bit code) path of the 8-bit global alignment.
//
// The forward kernel (k_align8_fwd_pk<W, PW, true>, bsa_align8_pk.hip) leaves, per target row, one code row of 16
// lanes x CW dwords (bsa_common.h "COMPACT slot"): for every band cell the outcome of the equality tests the
// reference's backcal would make there (bsalign.h:3667-3852).  This kernel makes the same walk from those bits alone:
//   cell (qb, tb):  prior_match ? (M ? match : D ? delete : insert) : (D ? delete : M ? match : insert)
//   insert:         length = distance to the nearest cell on the left whose R bit is set (bsalign.h:3798-3814)
//   delete:         walk up the column until a row whose Od bit is set (bsalign.h:3730-3744)
// The test-only scalar restatement states the same rules (tests/test_oracle_codes.py); both are checked against the
// literal backcal.  Whatever the bits cannot decide (a scan leaves the band; the cases in which the reference itself does not
// terminate) is reported as BSA_ST_TRACE.
//
// One pair per lane.  Rows are 64 bytes and are visited strictly upwards, and the band follows the path, so the
// dword a lane needs from the next rows is almost always the same block's: the lane keeps the dwords of the next
// RING rows of its current block in registers, requested RING steps ahead (loads of a wave return in order, so the
// request stream simply runs ahead of the walk); only a block change costs a memory round trip.
#include "bsa_common.h"

#define CODE_RING 8

template<int W>
__global__ void __launch_bounds__(64) k_align8_trace_codes(const Align8Args a, bsa_result_t *out, uint32_t *cig_cnt){
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
	uint32_t *cig_end = (uint32_t*)(rows + ((size_t)tlen + 1) * RB);
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
		const uint32_t *rp = (const uint32_t*)(rows + (size_t)r * RB) + y * CW;
		return unpack(rp[0], (CW > 1) ? rp[CW - 1] : 0u);
	};
	// ring of the next CODE_RING rows (r = ring_top, ring_top - 1, ...) of block ring_y
	uint32_t ring0[CODE_RING], ring1[CODE_RING];
	int ring_top = -1000; uint32_t ring_y = 0xFFFFu;
	auto ring_fill = [&](int r, uint32_t y){
#pragma unroll
		for(int k = 0; k < CODE_RING; k++){
			const int rr = max(r - k, 0);
			const uint32_t *rp = (const uint32_t*)(rows + (size_t)rr * RB) + y * CW;
			ring0[k] = rp[0]; ring1[k] = (CW > 1) ? rp[CW - 1] : 0u;
		}
		ring_top = r; ring_y = y;
	};
	auto ring_get = [&](int r, uint32_t y) -> Code {          // r must be ring_top or ring_top - 1 for the fast path
		if(y != ring_y || r > ring_top || r < ring_top - 1) ring_fill(r, y);
		if(r == ring_top - 1){
			// advance by one row: drop the top entry, request the row that enters at the bottom
#pragma unroll
			for(int k = 0; k + 1 < CODE_RING; k++){ ring0[k] = ring0[k + 1]; ring1[k] = ring1[k + 1]; }
			const int rr = max(r - (CODE_RING - 1), 0);
			const uint32_t *rp = (const uint32_t*)(rows + (size_t)rr * RB) + y * CW;
			ring0[CODE_RING - 1] = rp[0]; ring1[CODE_RING - 1] = (CW > 1) ? rp[CW - 1] : 0u;
			ring_top = r;
		}
		return unpack(ring0[0], ring1[0]);
	};
	bool bad = false;
	const int score = begs[tlen + 1];
	if(score == (int)0x80000000u) bad = true;                      // band never reached the query end (bsalign.h:4034)
	rs.score = score;
	rs.qe = (int)qlen - 1; rs.te = (int)tlen - 1;
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
		const Code c = ring_get(rs.tb, y);
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
				if(r == -1){ if(rs.qb >= bw) bad = true; break; }
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
		rs.qb = rs.qe = rs.tb = rs.te = 0; rs.mat = rs.mis = rs.ins = rs.del = rs.aln = 0;
		ncig = 0;
	}
	out[pair] = rs;
	cig_cnt[ppos] = ncig;
}

hipError_t bsa_launch_align8_trace_codes(const Align8Args &a, int pw, bsa_result_t *out, uint32_t *cig_cnt, hipStream_t st){
	(void)pw;
	const uint32_t blocks = (a.count + 63u) / 64u;
	if(blocks == 0) return hipSuccess;
	switch(a.bw / 16){
		case 4:  hipLaunchKernelGGL((k_align8_trace_codes<4>), dim3(blocks), dim3(64), 0, st, a, out, cig_cnt); break;
		case 8:  hipLaunchKernelGGL((k_align8_trace_codes<8>), dim3(blocks), dim3(64), 0, st, a, out, cig_cnt); break;
		case 16: hipLaunchKernelGGL((k_align8_trace_codes<16>), dim3(blocks), dim3(64), 0, st, a, out, cig_cnt); break;
		default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}
