/*
 * bsalign_oracle.c -- TEST INFRASTRUCTURE ONLY (see bsalign_oracle.h).
 *
 * Scalar, lane-exact CPU restatement of the reference's 8-bit banded striped
 * pairwise DP (ruanjue/bsalign, bsalign.h).  No SIMD: every SSE operation of
 * the reference is expressed as a loop over the 16 lanes with explicit
 * saturation / truncation so that results are bit-identical, including the
 * corner cases where int8 saturation or int->int8 truncation fires.
 *
 * Each function cites the reference file:line it follows.  Parity is pinned
 * against the real reference (oracle/_ref/libbsref.so) and tests/golden/.
 */
#include "bsalign_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define NL ORC_LANES

/* ---- int8 / int16 lane arithmetic with the x86 semantics the reference relies on ---- */
static inline int sat8(int v){ return v > 127 ? 127 : (v < -128 ? -128 : v); }      /* _mm_adds_epi8 / _mm_subs_epi8 / _mm_packs_epi16 */
static inline int sat16(int v){ return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); } /* _mm_adds_epi16 / _mm_packs_epi32 */
static inline int trunc8(int v){ return (int)(int8_t)(uint8_t)(v & 0xFF); }          /* int stored into b1i / _mm_set1_epi8 / _mm_insert_epi8 */
static inline int imax(int a, int b){ return a > b ? a : b; }
static inline int imin(int a, int b){ return a < b ? a : b; }
static inline uint32_t sidx(uint32_t W, uint32_t p){ return (p % W) * NL + p / W; } /* bsalign.h:321 */

void orc_set_score_matrix(int8_t mtx[16], int mat, int mis){ /* bsalign.h:323 */
	int i;
	for(i = 0; i < 16; i++) mtx[i] = (int8_t)(((i >> 2) == (i & 3)) ? mat : mis);
}

int orc_get_piecewise(int gapo1, int gape1, int gapo2, int gape2, int bandwidth){ /* bsalign.h:2084-2092 */
	if(gapo2 < gapo1 && gape2 > gape1 && gapo2 + gape2 < gapo1 + gape1){
		if((gapo1 - gapo2) / (gape1 - gape2) < bandwidth) return 2;
	}
	return gapo1 ? 1 : 0;
}

/* S(x, base): what the striped profile would hold for query column x (bsalign.h:2166-2221) */
static inline int score_at(const orc_query_t *qy, uint64_t x, uint8_t base){
	int c;
	if(x >= qy->len) return ORC_EPI8_MIN;
	c = qy->mtx[qy->seq[x] * 4 + base];
	if(qy->hpc && x + 1 < qy->len && qy->seq[x] != qy->seq[x + 1]) c += qy->bonus;
	return trunc8(c); /* stored into a b1i profile cell */
}

void orc_row_init(int8_t *us, int8_t *es, int8_t *qs, int32_t *ubegs, int mode, uint32_t bandwidth,
		int max_nt, int min_nt, int gapo1, int gape1, int gapo2, int gape2){ /* bsalign.h:2094-2140 */
	uint32_t W = bandwidth / NL, k;
	int two = (gapo2 < gapo1 && gape2 > gape1 && gapo2 + gape2 < gapo1 + gape1
	           && (gapo1 - gapo2) / (gape1 - gape2) < (int)bandwidth);
	int type = mode & 3;
	if(type == ORC_MODE_GLOBAL || type == ORC_MODE_EXTEND){
		int run, tmp;
		if(two){
			/* first xp cells cost gape1, the rest gape2 (bsalign.h:2102-2112) */
			uint32_t xp = (uint32_t)((gapo2 - gapo1) / (gape1 - gape2));
			for(k = 0; k < bandwidth; k++) us[k] = (int8_t)gape2;
			for(k = 0; k < NL; k++) ubegs[k] = gape2 * (int)W;
			us[0] = (int8_t)trunc8(gapo1 + gape1 + min_nt - max_nt);
			ubegs[0] += us[0] - gape2;
			for(k = 1; k < xp; k++){
				us[sidx(W, k)] = (int8_t)gape1;
				ubegs[k / W] += gape1 - gape2;
			}
		} else {
			for(k = 0; k < bandwidth; k++) us[k] = (int8_t)gape1;
			us[0] = (int8_t)trunc8(gapo1 + gape1 + min_nt - max_nt);
			for(k = 0; k < NL; k++) ubegs[k] = gape1 * (int)W;
			ubegs[0] += us[0] - gape1;
		}
		/* per-block sums -> exclusive prefix starting at max_nt - min_nt (bsalign.h:2120-2126) */
		run = max_nt - min_nt;
		for(k = 0; k < NL; k++){ tmp = ubegs[k]; ubegs[k] = run; run += tmp; }
		ubegs[NL] = run;
	} else {
		memset(us, 0, bandwidth);
		memset(ubegs, 0, (NL + 1) * sizeof(int32_t));
	}
	if(two){
		memset(es, ORC_EPI8_MIN & 0xFF, bandwidth);
		memset(qs, ORC_EPI8_MIN & 0xFF, bandwidth);
	} else if(gapo1){
		memset(es, ORC_EPI8_MIN & 0xFF, bandwidth);
	}
}

/* shift one striped byte row: dst vector i <- src vector (i+mov) lanes down by cyc, wrapping vectors take cyc+1 (bsalign.h:2271-2345) */
static void movx_bytes(int8_t *dst, const int8_t *src, uint32_t W, uint32_t cyc, uint32_t mov){
	uint32_t i, j, div = W - mov;
	for(i = 0; i < div; i++){
		for(j = 0; j < NL; j++) dst[i * NL + j] = (j + cyc < NL) ? src[(i + mov) * NL + j + cyc] : 0;
	}
	for(i = div; i < W; i++){
		for(j = 0; j < NL; j++) dst[i * NL + j] = (j + cyc + 1 < NL) ? src[(i - div) * NL + j + cyc + 1] : 0;
	}
}

void orc_row_movx(int8_t *us_dst, int8_t *es_dst, int8_t *qs_dst, int32_t *ub_dst,
		const int8_t *us_src, const int8_t *es_src, const int8_t *qs_src, const int32_t *ub_src,
		uint32_t W, uint32_t movx, int piecewise, int nt_max, int nt_min,
		int gapo1, int gape1, int gapo2, int gape2){ /* bsalign.h:2244-2392 */
	uint32_t bw = W * NL, i, j, cyc, mov, div;
	int32_t tmp[NL + 1];
	if(movx >= bw){ /* bsalign.h:2253-2259 */
		memset(us_dst, 0, bw);
		if(piecewise) memset(es_dst, 0, bw);
		if(piecewise == 2) memset(qs_dst, 0, bw);
		for(i = 0; i <= NL; i++) ub_dst[i] = ORC_SCORE_MIN;
		return;
	}
	if(movx == 0){ /* bsalign.h:2260-2269 */
		memcpy(us_dst, us_src, bw);
		if(piecewise) memcpy(es_dst, es_src, bw);
		if(piecewise == 2) memcpy(qs_dst, qs_src, bw);
		memcpy(ub_dst, ub_src, (NL + 1) * sizeof(int32_t));
		return;
	}
	cyc = movx / W; mov = movx % W; div = W - mov;
	movx_bytes(us_dst, us_src, W, cyc, mov);
	if(piecewise) movx_bytes(es_dst, es_src, W, cyc, mov);
	if(piecewise == 2) movx_bytes(qs_dst, qs_src, W, cyc, mov);
	/* block start scores: add the `mov` skipped vectors lane-wise, then drop `cyc` blocks (bsalign.h:2310-2355) */
	for(j = 0; j < NL; j++){
		int32_t s = ub_src[j];
		for(i = 0; i < mov; i++) s += us_src[i * NL + j];
		tmp[j] = s;
	}
	for(j = 0; j + cyc < NL; j++) ub_dst[j] = tmp[j + cyc];
	for(j = NL - cyc; j <= NL; j++) ub_dst[j] = ub_src[NL];
	{
		/* synthetic overhang for the movx new cells at the band end (bsalign.h:2357-2389) */
		uint32_t d, p0, a, a2, b, b2;
		int c;
		(void)div;
		d = (piecewise == 2) ? (uint32_t)((gapo1 - gapo2) / (gape2 - gape1)) : bw + 1;
		p0 = bw - movx;
		a = p0 % W; a2 = (p0 + d) % W;
		b = p0 / W; b2 = (p0 + d) / W;
		if(piecewise == 2) c = imin(nt_min, gapo2 + gape2) - 1 - nt_max + (gapo2 + gape2);
		else               c = imin(nt_min, gapo1 + gape1) - 1 - nt_max + (gapo1 + gape1);
		us_dst[a * NL + b] = (int8_t)trunc8(c);
		a++;
		for(; b < NL && b <= b2; b++){
			if(b == b2){
				c += (int)(a2 - a) * gape1; /* unsigned wrap of (a2 - a) when a2 < a is reproduced by the int cast */
				for(; a < a2; a++) us_dst[a * NL + b] = (int8_t)gape1;
				a = a2;
				if(a2 < W) break;
			}
			c += (int)(W - a) * gape1;
			for(; a < W; a++) us_dst[a * NL + b] = (int8_t)gape1;
			ub_dst[b + 1] += c;
			a = 0;
		}
		for(; b < NL; b++){
			c += (int)(W - a) * gape2;
			for(; a < W; a++) us_dst[a * NL + b] = (int8_t)gape2;
			ub_dst[b + 1] += c;
			a = 0;
		}
	}
}

/* active F-loop across running blocks (bsalign.h:2639-2652): scalar int chain, truncated to int8 on store */
static void f_penetrate(uint32_t W, int f[NL], const int32_t *ub, int gape){
	int fs[NL], i, s, t;
	for(i = NL - 1; i > 0; i--) fs[i] = f[i - 1];
	fs[0] = ORC_EPI8_MIN;
	t = (int)((uint32_t)W * (uint32_t)gape);
	s = t + fs[0] - (ub[1] - ub[0]);
	for(i = 1; i < NL; i++){
		if(fs[i] < s) fs[i] = trunc8(s);
		s = t + fs[i] - (ub[i + 1] - ub[i]);
	}
	for(i = 0; i < NL; i++) f[i] = fs[i];
}

/* ---- 4-bit traceback codes (prototype of the device's compact traceback; see orc_align_pairwise_codes below) ----
 * When tl_codes is set, orc_row_cal also emits, per band cell p of the row it computes, the outcome of the equality
 * tests the reference's backcal would make there, from values the row kernel has at hand:
 *   bit 0  M   h == s                     (backcal_cell, bsalign.h:3679-3699)
 *   bit 1  D   h == u + e                 (same; u, e of the previous row at this column)
 *   bit 2  R   the insertion reaching cell p+1 is opened at p (h + gapoe >= f + gape): the insert-length scan
 *              bsalign.h:3798-3814 stops at the first such cell to the left
 *   bit 3  Od  the stored e of this cell equals gapo+gape: the delete-length scan bsalign.h:3730-3744 stops here
 * Sound only when no saturation fires anywhere (stored differences exact).
 * piecewise == 2 (8 bits per cell): backcal needs nine facts there -- M, D (h == u + e), D2 (h == u + q), which of the two
 * insertion chains equals h (I1: h == f, I2: h == g; only consulted when none of M, D, D2 holds), R1 / R2 (the chain of piece
 * 1 / 2 reaching cell p+1 is opened at p) and Od1 / Od2 (the stored e / q is a fresh opening).  The five decision facts fold
 * into four bits because I1 / I2 matter only where M = D = D2 = 0 and at least one of the five always holds:
 *   bit 0  A   M, or (no D, no D2, I1 and I2)          bit 1  D          bit 2  D2          bit 3  B   not M and I1
 *   D or D2 set: A is M.  Else (A, B) = (1, 0) M; (1, 1) both chains; (0, 1) chain 1 only; (0, 0) chain 2 only.
 *   bit 4  R1      bit 5  R2      bit 6  Od1      bit 7  Od2 */
static __thread uint8_t *tl_codes = NULL;   /* bw bytes for the row being computed, natural band order */
static __thread uint32_t tl_mov = 0;

int orc_row_cal(uint32_t rbeg, uint8_t base,
		const int8_t *us0, const int8_t *es0, const int8_t *qs0, const int32_t *ub0,
		int8_t *us1, int8_t *es1, int8_t *qs1, int32_t *ub1,
		const orc_query_t *qy, int gapo1, int gape1, int gapo2, int gape2,
		uint32_t W, int rh, int piecewise){ /* bsalign.h:2727-2793 (pw0), 2885-2960 (pw1), 3084-3179 (pw2) */
	int h[NL], f[NL], g[NL], v[NL], z[NL], ulast[NL];
	int GapE = trunc8(gape1), GapOE = trunc8(gapo1 + gape1);
	int GapP = trunc8(gape2), GapQP = trunc8(gapo2 + gape2);
	int GapOQ = sat8(GapOE - GapQP);
	int h0, t, j;
	uint32_t i;
	/* seed for band cell 0 (bsalign.h:2899-2907) */
	h0 = (rh - ub0[0]) + score_at(qy, rbeg, base);
	if(piecewise == 0)      t = us0[0] + gape1;
	else if(piecewise == 1) t = us0[0] + es0[0];
	else                    t = us0[0] + imax(es0[0], qs0[0]);
	if(h0 >= t){ if(h0 > ORC_EPI8_MAX) h0 = ORC_EPI8_MAX; }
	else h0 = ORC_EPI8_MIN;
	h0 = trunc8(h0);
	/* pass 1: f (and g) leaving every running block when it starts from -63 */
	for(j = 0; j < NL; j++){ f[j] = g[j] = ORC_EPI8_MIN; h[j] = score_at(qy, (uint64_t)rbeg + (uint64_t)j * W, base); }
	h[0] = h0;
	for(i = 0; i < W; i++){
		for(j = 0; j < NL; j++){
			int u = us0[i * NL + j], e, q, hh = h[j];
			if(piecewise == 0){
				e = sat8(u + GapE);
				hh = imax(e, hh); hh = imax(f[j], hh);
				f[j] = sat8(sat8(hh + GapE) - u);
			} else if(piecewise == 1){
				e = sat8(es0[i * NL + j] + u);
				hh = imax(e, hh); hh = imax(f[j], hh);
				f[j] = sat8(f[j] + GapE);
				hh = sat8(hh + GapOE);
				f[j] = sat8(imax(f[j], hh) - u);
			} else {
				e = sat8(es0[i * NL + j] + u);
				q = sat8(qs0[i * NL + j] + u);
				hh = imax(e, hh); hh = imax(q, hh); hh = imax(f[j], hh); hh = imax(g[j], hh);
				f[j] = sat8(f[j] + GapE);
				hh = sat8(hh + GapOE);
				f[j] = sat8(imax(f[j], hh) - u);
				g[j] = sat8(g[j] + GapP);
				hh = sat8(hh - GapOQ);
				g[j] = sat8(imax(g[j], hh) - u);
			}
			h[j] = score_at(qy, (uint64_t)rbeg + i + 1 + (uint64_t)j * W, base);
		}
	}
	f_penetrate(W, f, ub0, gape1);
	if(piecewise == 2) f_penetrate(W, g, ub0, gape2);
	/* pass 2: the row itself */
	for(j = 0; j < NL; j++){ v[j] = 0; z[j] = score_at(qy, (uint64_t)rbeg + (uint64_t)j * W, base); ulast[j] = 0; h[j] = 0; }
	z[0] = h0;
	for(i = 0; i < W; i++){
		for(j = 0; j < NL; j++){
			int u = us0[i * NL + j], e, q, hh;
			if(piecewise == 0){
				e = sat8(u + GapE);
				hh = imax(e, z[j]); hh = imax(f[j], hh);
				if(tl_codes){
					const uint32_t pp = (uint32_t)j * W + i, xx = pp + tl_mov;
					const int sraw = score_at(qy, (uint64_t)rbeg + pp, base);
					int code = 4 | 8, lhs = hh, zc = sraw;                 /* linear gaps: every gap is "opened" at length 1 */
					if(pp == 0 && rbeg == 0){ zc = rh - ub0[0] + sraw; lhs = ub0[0] + hh - rh; }
					if(xx <= W * NL && hh == zc) code |= 1;
					if(xx < W * NL && lhs == (int)u + gapo1 + gape1) code |= 2;
					tl_codes[pp] = (uint8_t)code;
				}
				v[j] = sat8(hh - v[j]); us1[i * NL + j] = (int8_t)v[j];
				v[j] = sat8(hh - u);
				f[j] = sat8(sat8(hh + GapE) - u);
			} else if(piecewise == 1){
				e = sat8(es0[i * NL + j] + u);
				hh = imax(e, z[j]); hh = imax(f[j], hh);
				if(tl_codes){
					const uint32_t pp = (uint32_t)j * W + i, xx = pp + tl_mov;
					const int sraw = score_at(qy, (uint64_t)rbeg + pp, base);
					int code = 0, lhs = hh, zc = sraw;
					if(pp == 0 && rbeg == 0){ zc = rh - ub0[0] + sraw; lhs = ub0[0] + hh - rh; }
					if(xx <= W * NL && hh == zc) code |= 1;
					if(xx < W * NL && lhs == (int)u + (int)es0[i * NL + j]) code |= 2;
					if(sat8(hh + GapOE) >= sat8(f[j] + GapE)) code |= 4;
					if(imax(sat8(sat8(e + GapE) - hh), GapOE) == GapOE) code |= 8;
					tl_codes[pp] = (uint8_t)code;
				}
				v[j] = sat8(hh - v[j]); us1[i * NL + j] = (int8_t)v[j];
				v[j] = sat8(hh - u);
				e = sat8(e + GapE); e = sat8(e - hh); e = imax(e, GapOE);
				es1[i * NL + j] = (int8_t)e;
				f[j] = sat8(f[j] + GapE);
				hh = sat8(hh + GapOE);
				f[j] = sat8(imax(f[j], hh) - u);
			} else {
				e = sat8(es0[i * NL + j] + u);
				hh = imax(e, z[j]);
				q = sat8(qs0[i * NL + j] + u);
				hh = imax(q, hh); hh = imax(f[j], hh); hh = imax(g[j], hh);
				if(tl_codes){
					const uint32_t pp = (uint32_t)j * W + i, xx = pp + tl_mov;
					const int sraw = score_at(qy, (uint64_t)rbeg + pp, base);
					int lhs = hh, zc = sraw, code = 0;
					int fM, fD, fD2, fI1, fI2;
					if(pp == 0 && rbeg == 0){ zc = rh - ub0[0] + sraw; lhs = ub0[0] + hh - rh; }
					fM = xx <= W * NL && hh == zc;
					fD = xx < W * NL && lhs == (int)u + (int)es0[i * NL + j];
					fD2 = xx < W * NL && lhs == (int)u + (int)qs0[i * NL + j];
					fI1 = hh == f[j]; fI2 = hh == g[j];
					if(fM || (!fD && !fD2 && fI1 && fI2)) code |= 1;
					if(fD) code |= 2;
					if(fD2) code |= 4;
					if(!fM && fI1) code |= 8;
					if(sat8(hh + GapOE) >= sat8(f[j] + GapE)) code |= 16;
					if(sat8(hh + GapQP) >= sat8(g[j] + GapP)) code |= 32;
					if(imax(sat8(sat8(e + GapE) - hh), GapOE) == GapOE) code |= 64;
					if(imax(sat8(sat8(q + GapP) - hh), GapQP) == GapQP) code |= 128;
					tl_codes[pp] = (uint8_t)code;
				}
				v[j] = sat8(hh - v[j]); us1[i * NL + j] = (int8_t)v[j];
				v[j] = sat8(hh - u);
				e = sat8(e + GapE); e = sat8(e - hh); e = imax(e, GapOE);
				es1[i * NL + j] = (int8_t)e;
				q = sat8(q + GapP); q = sat8(q - hh); q = imax(q, GapQP);
				qs1[i * NL + j] = (int8_t)q;
				f[j] = sat8(f[j] + GapE);
				hh = sat8(hh + GapOE);
				f[j] = sat8(imax(f[j], hh) - u);
				g[j] = sat8(g[j] + GapP);
				hh = sat8(hh - GapOQ);
				g[j] = sat8(imax(g[j], hh) - u);
			}
			h[j] = hh; ulast[j] = u;
			z[j] = score_at(qy, (uint64_t)rbeg + i + 1 + (uint64_t)j * W, base);
		}
	}
	/* undo the gap offsets folded into h by the last iteration (bsalign.h:2958, 3177) */
	if(piecewise == 1){ for(j = 0; j < NL; j++) h[j] = sat8(h[j] - GapOE); }
	else if(piecewise == 2){ for(j = 0; j < NL; j++) h[j] = sat8(h[j] - GapQP); }
	/* tail (bsalign.h:2618-2636): vertical deltas at block ends -> next ubegs; fix the first vector of u; re-base */
	{
		int vv[NL];
		for(j = 0; j < NL; j++) vv[j] = sat8(h[j] - ulast[j]);
		for(j = 1; j <= NL; j++) ub1[j] = ub0[j] + vv[j - 1];
		for(j = NL - 1; j > 0; j--) us1[j] = (int8_t)sat8(us1[j] - vv[j - 1]);
		/* lane 0 subtracts the shifted-in zero */
		ub1[0] = ub0[0] + us1[0];
		us1[0] = 0;
	}
	return ub1[0];
}

void orc_row_merge(const int8_t *us0, const int8_t *es0, const int8_t *qs0, const int32_t *ub0,
		const int8_t *us1, const int8_t *es1, const int8_t *qs1, const int32_t *ub1,
		int8_t *us2, int8_t *es2, int8_t *qs2, int32_t *ub2, uint32_t W, int piecewise){ /* bsalign.h:2474-2616 */
	uint32_t i, ie, j;
	for(j = 0; j < NL; j++){
		int s0 = ub0[j], s1 = ub1[j];
		ub2[j] = imax(s0, s1);
		for(i = 0; i < W; ){
			int d, x0, x1, t0, t1, mprev;
			ie = (i + 256 < W) ? i + 256 : W;
			/* common base + int16 offsets (bsalign.h:2499-2533) */
			d = s0 - s1;
			if(d < -0x7FFF) d = -0x7FFF;
			if(d >  0x7FFF) d =  0x7FFF;
			x0 = d >> 1;             /* arithmetic shift: -1 >> 1 == -1 */
			x1 = x0 - d;
			s0 -= x0; s1 -= x1;
			t0 = sat16(x0); t1 = sat16(x1);
			mprev = imax(t0, t1);
			for(; i < ie; i++){
				int m, me, a0, a1;
				t0 = sat16(t0 + us0[i * NL + j]);
				t1 = sat16(t1 + us1[i * NL + j]);
				m = imax(t0, t1);
				us2[i * NL + j] = (int8_t)sat8(sat16(m - mprev));
				mprev = m;
				if(piecewise == 0) continue;
				a0 = sat16(t0 + es0[i * NL + j]);
				a1 = sat16(t1 + es1[i * NL + j]);
				me = imax(a0, a1);
				es2[i * NL + j] = (int8_t)sat8(sat16(me - m));
				if(piecewise == 1) continue;
				a0 = sat16(t0 + qs0[i * NL + j]);
				a1 = sat16(t1 + qs1[i * NL + j]);
				me = imax(a0, a1);
				qs2[i * NL + j] = (int8_t)sat8(sat16(me - m));
			}
			s0 += t0; s1 += t1;
		}
	}
	ub2[NL] = imax(ub0[NL], ub1[NL]);
}

int orc_getscore(const int8_t *us, const int32_t *ubegs, uint32_t W, uint64_t pos){ /* bsalign.h:3187-3197 */
	uint32_t x = (uint32_t)(pos % W), y = (uint32_t)(pos / W), i;
	int s = ubegs[y];
	for(i = 0; i <= x; i++) s += us[i * NL + y];
	return s;
}

uint32_t orc_row_max(const int8_t *us, const int32_t *ubegs, uint32_t W, int32_t *max_score){ /* bsalign.h:3213-3329 */
	/* per lane: best chunk (32 vectors per chunk, first chunk wins ties), then lanes are reduced in the
	 * reference's register order: (k vs 4+k), (8+k vs 12+k), pairs, then k ascending -- lower wins ties */
	static const int lane_order[NL] = {0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15};
	const uint32_t STEP = 32;
	int lmax[NL]; uint32_t lchunk[NL];
	uint32_t i, j, x, y, c;
	int best, lane, uscr, umax;
	for(j = 0; j < NL; j++){
		int base = ubegs[j], run, cmax;
		lmax[j] = ORC_SCORE_MIN; lchunk[j] = 0;
		for(i = 0, c = 0; i < W; i += STEP, c++){
			uint32_t n = (i + STEP < W) ? STEP : W - i;
			run = 0; cmax = -32767;
			for(x = 0; x < n; x++){
				run = sat16(run + us[(i + x) * NL + j]);
				if(run > cmax) cmax = run;
			}
			if(base + cmax > lmax[j]){ lmax[j] = base + cmax; lchunk[j] = c; }
			base += run;
		}
	}
	/* reduction order of bsalign.h:3264-3277 */
	{
		int m01[4], i01[4], m23[4], i23[4], mm[4], ii[4], k;
		for(k = 0; k < 4; k++){
			if(lmax[4 + k] > lmax[k]){ m01[k] = lmax[4 + k]; i01[k] = 4 + k; } else { m01[k] = lmax[k]; i01[k] = k; }
			if(lmax[12 + k] > lmax[8 + k]){ m23[k] = lmax[12 + k]; i23[k] = 12 + k; } else { m23[k] = lmax[8 + k]; i23[k] = 8 + k; }
			if(m23[k] > m01[k]){ mm[k] = m23[k]; ii[k] = i23[k]; } else { mm[k] = m01[k]; ii[k] = i01[k]; }
		}
		best = mm[0]; lane = ii[0];
		for(k = 1; k < 4; k++){ if(mm[k] > best){ best = mm[k]; lane = ii[k]; } }
	}
	(void)lane_order;
	*max_score = best;
	x = lchunk[lane] * STEP;
	y = (x + STEP < W) ? x + STEP : W;
	j = x; umax = ORC_SCORE_MIN; uscr = 0;
	for(; x < y; x++){
		uscr += us[x * NL + lane];
		if(uscr > umax){ j = x; umax = uscr; }
	}
	return (uint32_t)lane * W + j;
}

int orc_band_mov(const int32_t *ubegs, uint32_t W, uint32_t tidx, uint32_t qoff, uint32_t qlen){ /* bsalign.h:3331-3349 */
	uint32_t i;
	int noisy = 0;
	if(tidx <= W * NL / 4) return 0;
	if(qoff + W * NL >= qlen) return 0;
	for(i = 1; i <= NL; i++){
		int a = ubegs[i], b = ubegs[i - 1];
		noisy += (a < b) ? (b - a) : (a - b);
	}
	{
		/* the reference's expression mixes int and u4i: the divisions happen in unsigned arithmetic */
		uint32_t nz = (uint32_t)(noisy / NL);
		nz = nz / W * NL / 2;
		noisy = (int)((2 * NL / 2 > nz) ? (uint32_t)(2 * NL / 2) : nz);
	}
	if(ubegs[0] + noisy < ubegs[NL]) return 2;
	if(ubegs[0] > ubegs[NL] + noisy) return 0;
	return 1;
}

/* ---------------- whole-pair driver + traceback ---------------- */

typedef struct {
	uint32_t bw, W; int pw;
	int8_t *ups, *eps, *qps;  /* (tlen+1) rows, row r stored at index r+1 (row -1 first) */
	int32_t *ubs;             /* (tlen+1) x 17 */
	int32_t *begs;            /* (tlen+1), begs[0] is the reference's begs[-1] = 0 */
} rows_t;

static inline int mtx_getscore(const rows_t *R, long row, long col){ /* bsalign.h:3199-3202 */
	return orc_getscore(R->ups + (row + 1) * (long)R->bw, R->ubs + (row + 1) * (NL + 1), R->W, (uint64_t)(col - R->begs[row + 1]));
}

/* cigar accumulation (bsalign.h:409-417) */
typedef struct { uint32_t *buf; long n, cap; } cigv_t;
static inline void cig_push(cigv_t *c, uint32_t w){ if(c->buf && c->n < c->cap) c->buf[c->n] = w; c->n++; }
static inline uint32_t cig_add(cigv_t *c, uint32_t cg, uint32_t op, uint32_t sz){
	if(op == (cg & 0xf)) return cg + (sz << 4);
	if(cg) cig_push(c, cg);
	return (sz << 4) | op;
}

#define BT_M 0
#define BT_I 1
#define BT_D 2
#define BT_D2 4

static int backcal_cell(int x, int s, const int Hs[2], int u, int e, int q, uint32_t bw, int pw, int prior_match){ /* bsalign.h:3667-3702 */
	int h = Hs[1] - Hs[0];
	if(x > (int)bw) return BT_I;
	if(x == (int)bw) return (h == s) ? BT_M : BT_I;
	if(prior_match){
		if(h == s) return BT_M;
		if(h == u + e) return BT_D;
		if(pw == 2 && h == u + q) return BT_D2;
		return BT_I;
	}
	if(h == u + e) return BT_D;
	if(pw == 2 && h == u + q) return BT_D2;
	if(h == s) return BT_M;
	return BT_I;
}

static int backcal(const uint8_t *qseq, const uint8_t *tseq, const rows_t *R, int mode, const int8_t *mtx,
		int gapo1, int gape1, int gapo2, int gape2, orc_result_t *rs, cigv_t *cv){ /* bsalign.h:3704-3852 */
	const int pw = R->pw; const uint32_t W = R->W, bw = R->bw;
	int Hs0, Hs1, pend = 0 /* (len<<4)|op of a vertical gap being extended */, prior_match = 0;
	uint32_t cg = 0;
	long long t;
	int type = mode & 3;
	rs->qb = rs->qe; rs->qe++;
	rs->tb = rs->te; rs->te++;
	rs->mat = rs->mis = rs->ins = rs->del = rs->aln = 0;
	Hs0 = 0;
	Hs1 = mtx_getscore(R, rs->tb, rs->qb);
	for(;;){
		if((pend & 0xf) == BT_D || (pend & 0xf) == BT_D2){
			int go = ((pend & 0xf) == BT_D) ? gapo1 : gapo2, ge = ((pend & 0xf) == BT_D) ? gape1 : gape2;
			Hs0 = mtx_getscore(R, rs->tb, rs->qb);
			t = go + (long long)(pend >> 4) * ge;
			if(Hs0 + t == Hs1){
				cg = cig_add(cv, cg, BT_D, (uint32_t)(pend >> 4));
				rs->del += pend >> 4; rs->aln += pend >> 4;
				Hs1 = Hs0; pend = 0;
			} else {
				pend += 1 << 4; rs->tb--;
				if(rs->tb < -1) return -1; /* the reference would walk off the matrix here (undefined) */
				continue;
			}
		}
		if(rs->qb < 0 || rs->tb < 0) break;
		if(rs->qb == R->begs[rs->tb]){ /* roffs[tb-1] */
			if(rs->qb){
				Hs0 = R->ubs[(long)rs->tb * (NL + 1)]; /* ubegs[0] of row tb-1 */
				prior_match = 0;
			} else if(type == ORC_MODE_OVERLAP || rs->tb == 0) Hs0 = 0;
			else if(pw < 2) Hs0 = gapo1 + gape1 * rs->tb;
			else Hs0 = imax(gapo1 + gape1 * rs->tb, gapo2 + gape2 * rs->tb);
		} else {
			Hs0 = mtx_getscore(R, rs->tb - 1, rs->qb - 1);
		}
		{
			int x = rs->qb - R->begs[rs->tb], bt, Hs[2], u, e, q;
			long ci = (long)rs->tb * (long)bw + (x % (int)W) * NL + (x / (int)W); /* row tb-1 is stored at index tb */
			/* the reference reads ups[t] even when x >= bw (value unused there); keep the read in range */
			if(x >= 0 && x < (int)bw){
				u = R->ups[ci];
				e = R->eps ? R->eps[ci] : gapo1 + gape1;
				q = R->qps ? R->qps[ci] : 0;
			} else { u = e = q = 0; }
			Hs[0] = Hs0; Hs[1] = Hs1;
			bt = backcal_cell(x, mtx[qseq[rs->qb] * 4 + tseq[rs->tb]], Hs, u, e, q, bw, pw, prior_match);
			prior_match = 1;
			if(bt == BT_M){
				if(qseq[rs->qb] == tseq[rs->tb]) rs->mat++; else rs->mis++;
				rs->qb--; rs->tb--; rs->aln++;
				cg = cig_add(cv, cg, 0, 1);
				Hs1 = Hs0;
			} else if(bt == BT_I){
				if(rs->qb <= 0){
					cg = cig_add(cv, cg, 1, 1);
					Hs1 = Hs0;
					rs->qb--; rs->ins++; rs->aln++;
				} else {
					int sz, found = 0;
					for(sz = 1; sz + R->begs[rs->tb + 1] <= rs->qb; sz++){
						if(pw == 2) t = imax(gapo1 + sz * gape1, gapo2 + sz * gape2);
						else t = gapo1 + sz * gape1;
						Hs0 = mtx_getscore(R, rs->tb, rs->qb - sz);
						if(Hs0 + t == Hs1){
							cg = cig_add(cv, cg, 1, (uint32_t)sz);
							Hs1 = Hs0;
							rs->qb -= sz; rs->ins += sz; rs->aln += sz;
							found = 1;
							break;
						}
					}
					if(!found) return -1; /* the reference loops forever here (no consistent insertion length) */
				}
			} else {
				pend = (1 << 4) | bt;
				rs->tb--;
				continue;
			}
		}
	}
	if(type == ORC_MODE_OVERLAP){
		if(cg) cig_push(cv, cg);
	} else {
		uint32_t op = 0, sz = 0;
		if(rs->qb >= 0){ op = 1; sz = (uint32_t)rs->qb + 1; rs->ins += sz; rs->qb = -1; }
		else if(rs->tb >= 0){ op = 2; sz = (uint32_t)rs->tb + 1; rs->del += sz; rs->tb = -1; }
		rs->aln += sz;
		cg = cig_add(cv, cg, op, sz);
		if(cg) cig_push(cv, cg);
	}
	rs->qb++; rs->tb++;
	/* reverse in place (bsalign.h:3850) */
	if(cv->buf){
		long a = 0, b = (cv->n < cv->cap ? cv->n : cv->cap) - 1;
		if(cv->n <= cv->cap){ while(a < b){ uint32_t w = cv->buf[a]; cv->buf[a] = cv->buf[b]; cv->buf[b] = w; a++; b--; } }
	}
	return 0;
}

static long align_core(const uint8_t *q, uint32_t qlen, const uint8_t *tq, uint32_t tlen,
		int mode, uint32_t bandwidth, const int8_t mtx[16],
		int gapo1, int gape1, int gapo2, int gape2,
		orc_result_t *res, uint32_t *cig, long cap, int32_t *begs_out, uint8_t *rows_out, uint32_t rowb){ /* bsalign.h:3854-4050 */
	rows_t R;
	orc_query_t qy;
	orc_result_t rs;
	cigv_t cv;
	int8_t *us0, *es0, *qs0;
	int32_t ub0[NL + 1];
	uint32_t bw, W, i, rbeg, mov;
	int pw, smax = -127, smin = 127, rh, type = mode & 3;
	memset(&rs, 0, sizeof(rs));
	if(qlen == 0 || tlen == 0){ /* the reference divides by W == 0 / reads row -1 here: undefined; the restatement refuses */
		if(res) *res = rs;
		return ORC_ERR_INPUT;
	}
	for(i = 0; i < qlen; i++) if(q[i] > 3){ if(res) *res = rs; return ORC_ERR_INPUT; }
	for(i = 0; i < tlen; i++) if(tq[i] > 3){ if(res) *res = rs; return ORC_ERR_INPUT; }
	bw = bandwidth ? bandwidth : qlen;
	bw = (bw + NL - 1) / NL * NL;
	W = bw / NL;
	pw = orc_get_piecewise(gapo1, gape1, gapo2, gape2, (int)bw);
	for(i = 0; i < 16; i++){ smax = imax(smax, mtx[i]); smin = imin(smin, mtx[i]); }
	R.bw = bw; R.W = W; R.pw = pw;
	R.ups = (int8_t*)malloc((size_t)bw * (tlen + 1));
	R.eps = pw ? (int8_t*)malloc((size_t)bw * (tlen + 1)) : NULL;
	R.qps = (pw == 2) ? (int8_t*)malloc((size_t)bw * (tlen + 1)) : NULL;
	R.ubs = (int32_t*)malloc(sizeof(int32_t) * (NL + 1) * ((size_t)tlen + 1));
	R.begs = (int32_t*)malloc(sizeof(int32_t) * ((size_t)tlen + 1));
	us0 = (int8_t*)malloc(bw); es0 = (int8_t*)malloc(bw); qs0 = (int8_t*)malloc(bw);
	qy.seq = q; qy.len = qlen; qy.mtx = mtx; qy.hpc = 0; qy.bonus = 0;
	rs.score = ORC_SCORE_MIN;
	orc_row_init(R.ups, R.eps, R.qps, R.ubs, mode, bw, smax, smin, gapo1, gape1, gapo2, gape2);
	R.begs[0] = 0;
	rbeg = 0; mov = 0;
	for(i = 0; i < tlen; i++){
		const int8_t *pu = R.ups + (size_t)i * bw, *pe = R.eps ? R.eps + (size_t)i * bw : NULL, *pq = R.qps ? R.qps + (size_t)i * bw : NULL;
		const int32_t *pb = R.ubs + (size_t)i * (NL + 1);
		int8_t *cu = R.ups + (size_t)(i + 1) * bw, *ce = R.eps ? R.eps + (size_t)(i + 1) * bw : NULL, *cq = R.qps ? R.qps + (size_t)(i + 1) * bw : NULL;
		int32_t *cb = R.ubs + (size_t)(i + 1) * (NL + 1);
		int rbx;
		/* band offset of this row + H(rbeg-1, i-1) (bsalign.h:3932-3946) */
		if(mov && rbeg + bw < qlen){
			uint32_t room = qlen - (rbeg + bw);
			if(mov > room) mov = room;
			rbeg += mov;
			rh = orc_getscore(pu, pb, W, mov - 1);
		} else {
			mov = 0;
			if(rbeg) rh = ORC_SCORE_MIN;
			else if(type == ORC_MODE_OVERLAP || i == 0) rh = 0;
			else if(pw < 2) rh = gapo1 + gape1 * (int)i;
			else rh = imax(gapo1 + gape1 * (int)i, gapo2 + gape2 * (int)i);
		}
		orc_row_movx(us0, es0, qs0, ub0, pu, pe, pq, pb, W, mov, pw, smax, smin, gapo1, gape1, gapo2, gape2);
		orc_row_cal(rbeg, tq[i], us0, es0, qs0, ub0, cu, ce, cq, cb, &qy, gapo1, gape1, gapo2, gape2, W, rh, pw);
		/* adaptive band + global steering (bsalign.h:4006-4021) */
		rbx = orc_band_mov(cb, W, i, rbeg, qlen);
		if(type == ORC_MODE_GLOBAL){
			int rbz = 2 * imax((int)(tlen / qlen), 1);
			int rby = (int)((1.0 * i / tlen) * qlen);
			uint32_t left = tlen - i - 1;
			if((long long)rbeg + (long long)rbz * (long long)left + (long long)bw <= (long long)(uint32_t)(qlen + (uint32_t)rbz - 1)){
				mov = 1 + ((uint32_t)(qlen - (rbeg + bw)) / (left > 1 ? left : 1));
			} else if((int)rbeg < rby - (int)bw){
				mov = (uint32_t)(rbx + 1);
			} else if((int)rbeg > rby){
				mov = (uint32_t)imax(0, rbx - 1);
			} else mov = (uint32_t)rbx;
		} else mov = (uint32_t)rbx;
		R.begs[i + 1] = (int32_t)rbeg;
		if(type != ORC_MODE_GLOBAL && rbeg + bw >= qlen){ /* bsalign.h:4023-4032 */
			int sc = orc_getscore(cu, cb, W, qlen - 1 - rbeg);
			if(sc > rs.score){ rs.score = sc; rs.qe = (int)qlen - 1; rs.te = (int)i; }
		}
	}
	{
		const int8_t *lu = R.ups + (size_t)tlen * bw;
		const int32_t *lb = R.ubs + (size_t)tlen * (NL + 1);
		if(type == ORC_MODE_GLOBAL){ /* bsalign.h:4034-4037 */
			if(qlen - 1 - rbeg >= bw){ /* band never reached the query end: the reference reads outside the row here */
				free(R.ups); free(R.eps); free(R.qps); free(R.ubs); free(R.begs); free(us0); free(es0); free(qs0);
				if(res) *res = rs;
				return ORC_ERR_TRACE;
			}
			rs.score = orc_getscore(lu, lb, W, qlen - 1 - rbeg);
			rs.qe = (int)qlen - 1; rs.te = (int)tlen - 1;
		} else {
			int32_t ms; uint32_t rmax = orc_row_max(lu, lb, W, &ms);
			if(ms > rs.score){ rs.score = ms; rs.qe = (int)(rbeg + rmax); rs.te = (int)tlen - 1; }
		}
	}
	if(begs_out) memcpy(begs_out, R.begs + 1, sizeof(int32_t) * tlen);
	if(rows_out){ /* dump in the device's slot layout (bsalign_amd/csrc/bsa_common.h): begs array, then row-group tiles of block records */
		uint32_t r, y, k2;
		const uint32_t cells = ((uint32_t)(pw + 1) * W + 3u) & ~3u, blk = cells + 4u;
		const uint32_t tg = (64u / blk) ? (64u / blk) : 1u, tileb = (tg * blk + 15u) & ~15u;
		const size_t begs_bytes = (((size_t)tlen + 2) * 4 + 15) & ~(size_t)15;
		int32_t *bg = (int32_t*)rows_out;
		(void)rowb;
		for(r = 0; r <= tlen; r++) bg[r] = R.begs[r];
		for(r = 0; r <= tlen; r++){
			for(y = 0; y < NL; y++){
				uint8_t *bp = rows_out + begs_bytes + ((size_t)(r / tg) * NL + y) * tileb + (size_t)(r % tg) * blk;
				for(k2 = 0; k2 < W; k2++){
					bp[k2] = (uint8_t)R.ups[(size_t)r * bw + k2 * NL + y];
					if(pw >= 1) bp[W + k2] = (uint8_t)R.eps[(size_t)r * bw + k2 * NL + y];
					if(pw == 2) bp[2 * W + k2] = (uint8_t)R.qps[(size_t)r * bw + k2 * NL + y];
				}
				memcpy(bp + cells, R.ubs + (size_t)r * (NL + 1) + y, 4);
			}
		}
	}
	cv.buf = cig; cv.n = 0; cv.cap = cig ? cap : 0;
	{
		int bad = 0;
		/* end cell must lie inside the stored band, else the reference reads outside its arena */
		if(rs.qe < R.begs[rs.te + 1] || rs.qe >= R.begs[rs.te + 1] + (int)bw) bad = 1;
		if(!bad) bad = backcal(q, tq, &R, mode, mtx, gapo1, gape1, gapo2, gape2, &rs, &cv);
		if(bad){
			free(R.ups); free(R.eps); free(R.qps); free(R.ubs); free(R.begs); free(us0); free(es0); free(qs0);
			if(res) *res = rs;
			return ORC_ERR_TRACE;
		}
	}
	free(R.ups); free(R.eps); free(R.qps); free(R.ubs); free(R.begs); free(us0); free(es0); free(qs0);
	if(res) *res = rs;
	if(cig && cv.n > cap) return -cv.n;
	return cv.n;
}

/* Traceback from the 4-bit codes alone (global mode): same decisions as backcal() above, every equality test replaced
 * by the bit the forward pass recorded.  codes[(r + 1) * bw + p] = code of band cell p of row r; begs[r + 1] = band
 * offset of row r (begs[0] = 0 for row -1).  Returns 0, or -1 where the literal traceback is needed (a scan leaves the
 * band, or the situation in which the reference itself does not terminate). */
static __thread int g_gapo1, g_gape1, g_gapo2, g_gape2;     /* gap costs of the alignment backcal_codes is walking */
static int backcal_codes(const uint8_t *qseq, const uint8_t *tseq, const uint8_t *codes, const int32_t *begs, uint32_t bw,
		int type, int pw, orc_result_t *rs, cigv_t *cv){
	int prior_match = 0;
	uint32_t cg = 0;
	rs->qb = rs->qe; rs->qe++;
	rs->tb = rs->te; rs->te++;
	rs->mat = rs->mis = rs->ins = rs->del = rs->aln = 0;
	for(;;){
		int p, code, bt;
		if(rs->qb < 0 || rs->tb < 0) break;
		if(rs->qb == begs[rs->tb] && rs->qb) prior_match = 0;
		p = rs->qb - begs[rs->tb + 1];                       /* position in row tb's own band */
		if(p < 0 || p >= (int)bw) return -1;
		code = codes[((size_t)rs->tb + 1) * bw + (size_t)p];
		int chains = 1;                                      /* insertion: which chains equal h (bit 0: piece 1, bit 1: piece 2) */
		int dtype = 8;                                       /* deletion: the Od bit that ends the run */
		if(pw == 2){
			const int fD = (code >> 1) & 1, fD2 = (code >> 2) & 1, fA = code & 1, fB = (code >> 3) & 1;
			const int fM = (fD || fD2) ? fA : (fA && !fB);
			int d = fD ? 1 : fD2 ? 2 : 0;                    /* backcal_cell, bsalign.h:3679-3701 */
			if(prior_match) bt = fM ? BT_M : d ? BT_D : BT_I;
			else bt = d ? BT_D : fM ? BT_M : BT_I;
			dtype = (d == 2) ? 128 : 64;
			chains = fA ? (fB ? 3 : 0) : (fB ? 1 : 2);       /* (only read when bt == BT_I, where M = D = D2 = 0) */
		} else if(prior_match) bt = (code & 1) ? BT_M : (code & 2) ? BT_D : BT_I;
		else bt = (code & 2) ? BT_D : (code & 1) ? BT_M : BT_I;
		prior_match = 1;
		if(bt == BT_M){
			if(qseq[rs->qb] == tseq[rs->tb]) rs->mat++; else rs->mis++;
			rs->qb--; rs->tb--; rs->aln++;
			cg = cig_add(cv, cg, 0, 1);
		} else if(bt == BT_I){
			if(rs->qb <= 0){
				cg = cig_add(cv, cg, 1, 1);
				rs->qb--; rs->ins++; rs->aln++;
			} else {
				int sz, found = 0;
				/* the nearest cell to the left at which a chain that equals h here was opened (bsalign.h:3798-3814: the smallest
				 * length whose cost, the larger of the two pieces', closes the gap) */
				const int rmask = (pw == 2) ? (((chains & 1) ? 16 : 0) | ((chains & 2) ? 32 : 0)) : 4;
				for(sz = 1; sz <= p; sz++){
					if(codes[((size_t)rs->tb + 1) * bw + (size_t)(p - sz)] & rmask){ found = 1; break; }
				}
				if(!found) return -1;
				if(pw == 2){
					/* the reference tests H(x - sz) + max(cost1(sz), cost2(sz)) == H(x): the chain that is tight here must also be the
					 * one with the larger (less negative) cost at this length.  Between real DP cells that always holds; next to cells
					 * that entered the band with synthetic values it can fail, and the reference's scan then finds no length at all */
					const int hit = codes[((size_t)rs->tb + 1) * bw + (size_t)(p - sz)] & rmask;
					const int c1 = g_gapo1 + sz * g_gape1, c2 = g_gapo2 + sz * g_gape2;
					if(!(((hit & 16) && c1 >= c2) || ((hit & 32) && c2 >= c1))) return -1;
				}
				cg = cig_add(cv, cg, 1, (uint32_t)sz);
				rs->qb -= sz; rs->ins += sz; rs->aln += sz;
			}
		} else {
			/* deletion: walk up the column until the row whose stored e is a fresh opening.  Two pieces, query column 0: the D / D2
			 * test there compares scores of two frames (the row is re-based at its first cell, bsalign.h:2632-2633) and can fire where
			 * no deletion ends; the reference's run-length scan (bsalign.h:3730-3760) works on real scores, finds no opening then and
			 * does not terminate: hand over to the literal traceback, which says so */
			int len = 1;
			if(pw == 2 && rs->qb == 0) return -1;
			for(;;){
				int r = rs->tb - len, pr;
				if(r < -1) return -1;
				if(r == -1){
					/* linear gaps: a deletion out of row -1 is an ordinary move.  Affine gaps: row -1 carries the e = -63 sentinel, so only
					 * a coincidence (h == u - 63 in row 0) sends a run here; the reference then compares real scores -- literal path */
					if(pw != 0 || rs->qb >= (int)bw) return -1;
					break;
				}
				pr = rs->qb - begs[r + 1];
				if(pr < 0 || pr >= (int)bw) return -1;
				if(codes[((size_t)r + 1) * bw + (size_t)pr] & dtype) break;
				len++;
			}
			cg = cig_add(cv, cg, BT_D, (uint32_t)len);
			rs->del += len; rs->aln += len;
			rs->tb -= len;
		}
	}
	if(type == ORC_MODE_OVERLAP){ if(cg) cig_push(cv, cg); }
	else {
		uint32_t op = 0, sz = 0;
		if(rs->qb >= 0){ op = 1; sz = (uint32_t)rs->qb + 1; rs->ins += sz; rs->qb = -1; }
		else if(rs->tb >= 0){ op = 2; sz = (uint32_t)rs->tb + 1; rs->del += sz; rs->tb = -1; }
		rs->aln += sz;
		cg = cig_add(cv, cg, op, sz);
		if(cg) cig_push(cv, cg);
	}
	rs->qb++; rs->tb++;
	if(cv->buf){
		long a = 0, b = (cv->n < cv->cap ? cv->n : cv->cap) - 1;
		if(cv->n <= cv->cap){ while(a < b){ uint32_t w = cv->buf[a]; cv->buf[a] = cv->buf[b]; cv->buf[b] = w; a++; b--; } }
	}
	return 0;
}

/* global alignment through the compact path: forward pass recording 4-bit codes, traceback from the codes.
 * Returns the CIGAR word count, ORC_ERR_TRACE when the compact traceback hands over to the literal one, ORC_ERR_INPUT
 * for inputs outside its domain (piecewise 2, non-global modes). */
long orc_align_pairwise_codes(const uint8_t *q, uint32_t qlen, const uint8_t *tq, uint32_t tlen,
		uint32_t bandwidth, const int8_t mtx[16], int gapo1, int gape1, int gapo2, int gape2,
		orc_result_t *res, uint32_t *cig, long cap){
	return orc_align_pairwise_codes_mode(q, qlen, tq, tlen, ORC_MODE_GLOBAL, bandwidth, mtx, gapo1, gape1, gapo2, gape2, res, cig, cap);
}

long orc_align_pairwise_codes_mode(const uint8_t *q, uint32_t qlen, const uint8_t *tq, uint32_t tlen, int mode,
		uint32_t bandwidth, const int8_t mtx[16], int gapo1, int gape1, int gapo2, int gape2,
		orc_result_t *res, uint32_t *cig, long cap){
	const int type = mode & 3;
	orc_query_t qy;
	orc_result_t rs;
	cigv_t cv;
	uint32_t bw, W, i, rbeg = 0, mov = 0;
	int pw, smax = -127, smin = 127, rh;
	int8_t *rowbuf;           /* two rows (u, e) + moved scratch */
	int32_t ubA[NL + 1], ubB[NL + 1], ub0[NL + 1], *begs;
	uint8_t *codes;
	memset(&rs, 0, sizeof(rs));
	if(res) *res = rs;
	if(qlen == 0 || tlen == 0) return ORC_ERR_INPUT;
	bw = bandwidth ? bandwidth : qlen;
	bw = (bw + NL - 1) / NL * NL;
	W = bw / NL;
	pw = orc_get_piecewise(gapo1, gape1, gapo2, gape2, (int)bw);
	for(i = 0; i < 16; i++){ smax = imax(smax, mtx[i]); smin = imin(smin, mtx[i]); }
	rowbuf = (int8_t*)calloc((size_t)bw * 9, 1);
	codes = (uint8_t*)calloc((size_t)bw * ((size_t)tlen + 1), 1);
	begs = (int32_t*)calloc((size_t)tlen + 2, sizeof(int32_t));
	{
		int8_t *pu = rowbuf, *pe = rowbuf + bw, *pq = rowbuf + 2 * bw, *cu = rowbuf + 3 * bw, *ce = rowbuf + 4 * bw, *cq = rowbuf + 5 * bw;
		int8_t *mu = rowbuf + 6 * bw, *me = rowbuf + 7 * bw, *mq = rowbuf + 8 * bw;
		int32_t *pb = ubA, *cb = ubB;
		qy.seq = q; qy.len = qlen; qy.mtx = mtx; qy.hpc = 0; qy.bonus = 0;
		orc_row_init(pu, pe, pq, pb, mode, bw, smax, smin, gapo1, gape1, gapo2, gape2);
		rs.score = ORC_SCORE_MIN;
		for(i = 0; i < tlen; i++){
			int rbx;
			int8_t *t8; int32_t *t32;
			if(mov && rbeg + bw < qlen){
				uint32_t room = qlen - (rbeg + bw);
				if(mov > room) mov = room;
				rbeg += mov;
				rh = orc_getscore(pu, pb, W, mov - 1);
			} else {
				mov = 0;
				if(rbeg) rh = ORC_SCORE_MIN;
				else if(type == ORC_MODE_OVERLAP || i == 0) rh = 0;
				else if(pw < 2) rh = gapo1 + gape1 * (int)i;
				else rh = imax(gapo1 + gape1 * (int)i, gapo2 + gape2 * (int)i);
			}
			orc_row_movx(mu, me, mq, ub0, pu, pe, pq, pb, W, mov, pw, smax, smin, gapo1, gape1, gapo2, gape2);
			tl_codes = codes + ((size_t)i + 1) * bw; tl_mov = mov;
			orc_row_cal(rbeg, tq[i], mu, me, mq, ub0, cu, ce, cq, cb, &qy, gapo1, gape1, gapo2, gape2, W, rh, pw);
			tl_codes = NULL;
			rbx = orc_band_mov(cb, W, i, rbeg, qlen);
			if(type == ORC_MODE_GLOBAL){
				int rbz = 2 * imax((int)(tlen / qlen), 1);
				int rby = (int)((1.0 * i / tlen) * qlen);
				uint32_t left = tlen - i - 1;
				if((long long)rbeg + (long long)rbz * (long long)left + (long long)bw <= (long long)(uint32_t)(qlen + (uint32_t)rbz - 1)){
					mov = 1 + ((uint32_t)(qlen - (rbeg + bw)) / (left > 1 ? left : 1));
				} else if((int)rbeg < rby - (int)bw) mov = (uint32_t)(rbx + 1);
				else if((int)rbeg > rby) mov = (uint32_t)imax(0, rbx - 1);
				else mov = (uint32_t)rbx;
			} else mov = (uint32_t)rbx;
			begs[i + 1] = (int32_t)rbeg;
			if(type != ORC_MODE_GLOBAL && rbeg + bw >= qlen){ /* end-of-query score of this row (bsalign.h:4023-4032) */
				int sc = orc_getscore(cu, cb, W, qlen - 1 - rbeg);
				if(sc > rs.score){ rs.score = sc; rs.qe = (int)qlen - 1; rs.te = (int)i; }
			}
			t8 = pu; pu = cu; cu = t8; t8 = pe; pe = ce; ce = t8; t8 = pq; pq = cq; cq = t8;
			t32 = pb; pb = cb; cb = t32;
		}
		if(type == ORC_MODE_GLOBAL){
			if(qlen - 1 - rbeg >= bw){ free(rowbuf); free(codes); free(begs); return ORC_ERR_TRACE; }
			rs.score = orc_getscore(pu, pb, W, qlen - 1 - rbeg);
			rs.qe = (int)qlen - 1; rs.te = (int)tlen - 1;
		} else {
			int32_t ms; uint32_t rmax = orc_row_max(pu, pb, W, &ms);
			if(ms > rs.score){ rs.score = ms; rs.qe = (int)(rbeg + rmax); rs.te = (int)tlen - 1; }
		}
	}
	cv.buf = cig; cv.n = 0; cv.cap = cig ? cap : 0;
	{
		int bad = 0;
		if(rs.qe < begs[rs.te + 1] || rs.qe >= begs[rs.te + 1] + (int)bw) bad = 1;
		g_gapo1 = gapo1; g_gape1 = gape1; g_gapo2 = gapo2; g_gape2 = gape2;
		if(!bad) bad = backcal_codes(q, tq, codes, begs, bw, type, pw, &rs, &cv);
		free(rowbuf); free(codes); free(begs);
		if(bad){ memset(&rs, 0, sizeof(rs)); if(res) *res = rs; return ORC_ERR_TRACE; }
	}
	if(res) *res = rs;
	if(cig && cv.n > cap) return -cv.n;
	return cv.n;
}

long orc_align_pairwise_trace(const uint8_t *q, uint32_t qlen, const uint8_t *t, uint32_t tlen,
		int mode, uint32_t bandwidth, const int8_t mtx[16], int gapo1, int gape1, int gapo2, int gape2,
		orc_result_t *res, uint32_t *cig, long cap, int32_t *begs_out){
	return align_core(q, qlen, t, tlen, mode, bandwidth, mtx, gapo1, gape1, gapo2, gape2, res, cig, cap, begs_out, NULL, 0);
}

long orc_align_pairwise(const uint8_t *q, uint32_t qlen, const uint8_t *t, uint32_t tlen,
		int mode, uint32_t bandwidth, const int8_t mtx[16], int gapo1, int gape1, int gapo2, int gape2,
		orc_result_t *res, uint32_t *cig, long cap){
	return align_core(q, qlen, t, tlen, mode, bandwidth, mtx, gapo1, gape1, gapo2, gape2, res, cig, cap, NULL, NULL, 0);
}

long orc_align_pairwise_rows(const uint8_t *q, uint32_t qlen, const uint8_t *t, uint32_t tlen,
		int mode, uint32_t bandwidth, const int8_t mtx[16], int gapo1, int gape1, int gapo2, int gape2,
		orc_result_t *res, uint8_t *rows_out, uint32_t rowb){
	return align_core(q, qlen, t, tlen, mode, bandwidth, mtx, gapo1, gape1, gapo2, gape2, res, NULL, 0, NULL, rows_out, rowb);
}

double orc_align_batch_time(const uint8_t *seqs, const uint64_t *qoff, const uint32_t *qlen,
		const uint64_t *toff, const uint32_t *tlen, long n, int mode, uint32_t bandwidth,
		const int8_t mtx[16], int gapo1, int gape1, int gapo2, int gape2, int64_t *checksum){
	struct timespec t0, t1;
	long k; int64_t cs = 0;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for(k = 0; k < n; k++){
		orc_result_t rs;
		long nc = orc_align_pairwise(seqs + qoff[k], qlen[k], seqs + toff[k], tlen[k], mode, bandwidth, mtx, gapo1, gape1, gapo2, gape2, &rs, NULL, 0);
		cs += rs.score + nc;
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	if(checksum) *checksum = cs;
	return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ======================= 2-bit edit-distance path (bsalign.h:612-1206) =======================
 *
 * The reference keeps, per row, two 64-bit bit-planes in striped order (word = p % W, bit = p / W)
 * holding u(p) = H(p, y) - H(p - 1, y) in {-1, 0, +1}: plane0 bit <=> u == -1, plane1 bit <=> u == +1
 * (bsalign.h:723-765).  Its row update (bsalign.h:766-810) is an exact fix-point of
 *     h = min(s, u + 1, v + 1),  u' = h - v,  v' = h - u
 * so this restatement evaluates the same recurrence cell by cell in plain ints and only then packs
 * the planes; band placement, boundary values, scoring by popcount and the traceback follow the
 * reference line by line.
 */

static inline int plane_bit(const uint64_t *pl, uint32_t W, long pos){ /* striped_seqedit_getval, bsalign.h:224 */
	/* the reference evaluates `x - begs[..]` in unsigned 32-bit arithmetic and x86 masks the shift count to 6 bits */
	uint32_t pu = (uint32_t)pos;
	return (int)((pl[pu % W] >> ((pu / W) & 63u)) & 1);
}

static int edit_rowmin(int sbeg, const int8_t *u, uint32_t bw, uint32_t *whence){ /* bsalign.h:813-963: first strict minimum of the running score */
	int sc = sbeg, smin = sbeg; uint32_t p, pmin = 0;
	for(p = 0; p < bw; p++){
		sc += u[p];
		if(sc < smin){ smin = sc; pmin = p; }
	}
	*whence = pmin;
	return smin;
}

long orc_edit_pairwise(const uint8_t *q, uint32_t qlen, const uint8_t *tq, uint32_t tlen,
		int mode, uint32_t bandwidth, orc_result_t *res, uint32_t *cig, long cap){ /* bsalign.h:1046-1206 */
	orc_result_t rs;
	cigv_t cv;
	uint64_t *pl0, *pl1;   /* (tlen+1) x W words each; row index 0 = initial row */
	uint32_t *begs;        /* tlen+1 */
	int8_t *uprev, *ucur;
	uint32_t bw, W, i, p, rb0, rb1, qround;
	int type = mode & 3, sbeg = 0, smin = 0x7FFFFFFF, srow, rx, ry;
	memset(&rs, 0, sizeof(rs));
	if(qlen == 0 || tlen == 0){ if(res) *res = rs; return 0; } /* bsalign.h:1051-1054 */
	for(i = 0; i < qlen; i++) if(q[i] > 3){ if(res) *res = rs; return ORC_ERR_INPUT; }
	for(i = 0; i < tlen; i++) if(tq[i] > 3){ if(res) *res = rs; return ORC_ERR_INPUT; }
	qround = (qlen + 63) / 64 * 64;
	if(type == ORC_MODE_OVERLAP || type == ORC_MODE_EXTEND){ /* bsalign.h:1055-1067 */
		bw = qround;
	} else {
		bw = (bandwidth + 63) / 64 * 64;
		if(bw == 0 || bw > qlen) bw = qround;
		if(bw < qlen){
			uint32_t step = (qlen + tlen - 1) / tlen + 1;
			if(bw < step) bw = (step + 63) / 64 * 64;
		}
	}
	W = bw / 64;
	pl0 = (uint64_t*)calloc((size_t)W * (tlen + 1), 8);
	pl1 = (uint64_t*)calloc((size_t)W * (tlen + 1), 8);
	begs = (uint32_t*)calloc((size_t)tlen + 1, 4);
	uprev = (int8_t*)malloc(bw); ucur = (int8_t*)malloc(bw);
	for(p = 0; p < bw; p++) uprev[p] = 1;                 /* row_init: u = +1 (bsalign.h:653-656) */
	for(p = 0; p < W; p++){ pl0[p] = 0; pl1[p] = ~0ULL; }
	rx = (int)qlen - 1; ry = (int)tlen - 1;
	rb0 = 0; begs[0] = 0;
	for(i = 0; i < tlen; i++){
		uint32_t movx;
		int v;
		if(type == ORC_MODE_OVERLAP || type == ORC_MODE_EXTEND) rb1 = 0;
		else { /* fixed diagonal band (bsalign.h:1112-1114) */
			rb1 = (uint32_t)(((uint64_t)i * qlen) / tlen);
			rb1 = (rb1 < bw / 2) ? 0 : rb1 - bw / 2;
			if(rb1 + bw > qround) rb1 = qround - bw;
		}
		begs[i + 1] = rb1;
		movx = rb1 - rb0;
		/* row_movx (bsalign.h:658-721): score at band start, then shift in +1 cells */
		if(type == ORC_MODE_OVERLAP) sbeg = 0;
		else {
			uint32_t m = movx < bw ? movx : bw;
			for(p = 0; p < m; p++) sbeg += uprev[p];
			sbeg++;
		}
		if(movx){
			if(movx >= bw){ for(p = 0; p < bw; p++) uprev[p] = 1; }
			else {
				memmove(uprev, uprev + movx, bw - movx);
				for(p = bw - movx; p < bw; p++) uprev[p] = 1;
			}
		}
		/* row_cal (bsalign.h:766-810) */
		v = (type == ORC_MODE_OVERLAP) ? 0 : 1;
		for(p = 0; p < bw; p++){
			uint32_t x = rb1 + p;
			int s = (x < qlen && q[x] == tq[i]) ? 0 : 1;
			int u = uprev[p], h = s;
			if(u + 1 < h) h = u + 1;
			if(v + 1 < h) h = v + 1;
			ucur[p] = (int8_t)(h - v);
			v = h - u;
		}
		{
			uint64_t *r0 = pl0 + (size_t)(i + 1) * W, *r1 = pl1 + (size_t)(i + 1) * W;
			for(p = 0; p < bw; p++){
				if(ucur[p] < 0) r0[p % W] |= 1ULL << (p / W);
				else if(ucur[p] > 0) r1[p % W] |= 1ULL << (p / W);
			}
		}
		if(type == ORC_MODE_OVERLAP || type == ORC_MODE_EXTEND){ /* bsalign.h:1124-1139 */
			uint32_t k;
			srow = sbeg;
			for(p = 0; p < bw; p++) srow += ucur[p];
			for(k = rb1 + bw; k > qlen; k--) srow -= ucur[k - 1 - rb1];
			if(srow < smin){ smin = srow; rx = (int)qlen - 1; ry = (int)i; }
		}
		{ int8_t *tmp = uprev; uprev = ucur; ucur = tmp; }
		rb0 = rb1;
	}
	if(type == ORC_MODE_EXTEND){ /* bsalign.h:1180-1187 */
		uint32_t k;
		srow = edit_rowmin(sbeg, uprev, bw, &k);
		if(srow < smin){ smin = srow; rx = (int)k; ry = (int)tlen - 1; }
	}
	/* backtrace (bsalign.h:965-1044) */
	cv.buf = cig; cv.n = 0; cv.cap = cig ? cap : 0;
	{
		int x = rx, y = ry; uint32_t cg = 0, op = 0;
		rs.qe = x + 1; rs.te = y + 1;
		while(x >= 0 && y >= 0){
			if(q[x] == tq[y]){ rs.mat++; op = 0; x--; y--; }
			else {
				long pos = (long)x - (long)begs[y + 1];
				int u3 = plane_bit(pl0 + (size_t)(y + 1) * W, W, pos), u4 = plane_bit(pl1 + (size_t)(y + 1) * W, W, pos);
				if(u3 == 0 && u4 == 1){ rs.ins++; op = 1; x--; }
				else {
					long pos0 = (long)x - (long)begs[y];
					int u1 = plane_bit(pl0 + (size_t)y * W, W, pos0), u2 = plane_bit(pl1 + (size_t)y * W, W, pos0);
					if(u1 == 1 && u2 == 0){ rs.del++; op = 2; y--; }
					else { rs.mis++; op = 0; x--; y--; }
				}
			}
			if(op == (cg & 0xf)) cg += 0x10;
			else { if(cg) cig_push(&cv, cg); cg = 0x10 | op; }
		}
		rs.qb = x + 1; rs.tb = y + 1;
		if(rs.qb){
			op = 1;
			if(op == (cg & 0xf)) cg += 0x10 * (uint32_t)rs.qb;
			else { if(cg) cig_push(&cv, cg); cg = (0x10 * (uint32_t)rs.qb) | op; }
			rs.ins += rs.qb; rs.qb = 0;
		}
		if((type == ORC_MODE_GLOBAL || type == ORC_MODE_EXTEND) && rs.tb){
			op = 2;
			if(op == (cg & 0xf)) cg += 0x10 * (uint32_t)rs.tb;
			else { if(cg) cig_push(&cv, cg); cg = (0x10 * (uint32_t)rs.tb) | op; }
			rs.del += rs.tb; rs.tb = 0;
		}
		rs.aln = rs.mat + rs.mis + rs.ins + rs.del;
		if(cg) cig_push(&cv, cg);
		if(cv.buf && cv.n <= cv.cap){
			long a = 0, b = cv.n - 1;
			while(a < b){ uint32_t w = cv.buf[a]; cv.buf[a] = cv.buf[b]; cv.buf[b] = w; a++; b--; }
		}
	}
	if(type == ORC_MODE_OVERLAP) rs.score = smin + rs.te - rs.tb; /* bsalign.h:1189-1203 */
	else if(type == ORC_MODE_EXTEND) rs.score = smin;
	else {
		uint32_t k;
		rs.score = sbeg;
		for(p = 0; p < bw; p++) rs.score += uprev[p];
		for(k = rb0 + bw; k > qlen; k--) rs.score -= uprev[k - 1 - rb0];
	}
	free(pl0); free(pl1); free(begs); free(uprev); free(ucur);
	if(res) *res = rs;
	if(cig && cv.n > cap) return -cv.n;
	return cv.n;
}

double orc_edit_batch_time(const uint8_t *seqs, const uint64_t *qoff, const uint32_t *qlen,
		const uint64_t *toff, const uint32_t *tlen, long n, int mode, uint32_t bandwidth, int64_t *checksum){
	struct timespec t0, t1;
	long k; int64_t cs = 0;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for(k = 0; k < n; k++){
		orc_result_t rs;
		long nc = orc_edit_pairwise(seqs + qoff[k], qlen[k], seqs + toff[k], tlen[k], mode, bandwidth, &rs, NULL, 0);
		cs += rs.score + nc;
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	if(checksum) *checksum = cs;
	return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- POA sweep programs (device counterpart: k_sweep in bsalign_amd/csrc/bsa_rows.hip) ----------------------
 * Executes flattened align_rd_bspoacore programs (bspoa.h:2515-2618) with the row functions above.  Row blocks use
 * the reference's layout: us[bw] | es[bw] if piecewise >= 1 | qs[bw] if piecewise == 2 | int32 ubegs[17], block
 * size roundup(bw * (piecewise + 1) + 68, 16) (bspoa.h:1787-1793, 2217). */
void orc_sweep_run(uint8_t *rows, const orc_row_task_t *tasks, const orc_sweep_prog_t *progs, size_t nprogs,
		const uint8_t *queries, const uint64_t *qoff, const uint32_t *qlen,
		int mode, uint32_t bandwidth, int M, int X, int refbonus, int gapo1, int gape1, int gapo2, int gape2, int T,
		orc_sweep_result_t *results){
	const uint32_t bw = (bandwidth + NL - 1) / NL * NL, W = bw / NL;
	const int pw = orc_get_piecewise(gapo1, gape1, gapo2, gape2, (int)bw);
	const size_t blk = ((size_t)bw * (pw + 1) + 17 * 4 + 15) & ~(size_t)15;
	const int type = mode & 3;
	const int nt_max = M + refbonus + 1, nt_min = X;       /* as the POA passes them, bspoa.h:2226, 2241 */
	int8_t mtx0[16], mtx1[16];
	int8_t *tmp = (int8_t*)calloc((size_t)bw * 6 + 16, 1);
	int8_t *mu = tmp, *me = tmp + bw, *mq = tmp + 2 * bw, *dz0 = tmp + 3 * bw, *dz1 = tmp + 4 * bw, *dz2 = tmp + 5 * bw;
	int32_t mb[NL + 1];
	size_t p;
	orc_set_score_matrix(mtx0, M, X);
	orc_set_score_matrix(mtx1, M + refbonus, X);
	for(p = 0; p < nprogs; p++){
		uint8_t *base = rows + (size_t)progs[p].first_block * blk;
		int maxscr = ORC_SCORE_MIN, maxidx = -1, maxoff = -1;
		uint32_t t;
#define BLK_US(i) ((int8_t*)(base + (size_t)(i) * blk))
#define BLK_ES(i) ((pw >= 1) ? BLK_US(i) + bw : dz0)
#define BLK_QS(i) ((pw == 2) ? BLK_US(i) + 2 * bw : dz1)
#define BLK_UB(i) ((int32_t*)(BLK_US(i) + (size_t)bw * (pw + 1)))
		for(t = 0; t < progs[p].ntasks; t++){
			const orc_row_task_t *tk = tasks + progs[p].first_task + t;
			if(tk->op == ORC_ROW_OP_INIT){
				orc_row_init(BLK_US(tk->dst), (pw >= 1) ? BLK_ES(tk->dst) : dz0, (pw == 2) ? BLK_QS(tk->dst) : dz1, BLK_UB(tk->dst),
					mode, bw, nt_max, nt_min, gapo1, gape1, gapo2, gape2);
			} else if(tk->op == ORC_ROW_OP_MERGE){               /* dst = max(src, dst), dpalign_row_merge_bspoa bspoa.h:2263 */
				orc_row_merge(BLK_US(tk->src), BLK_ES(tk->src), BLK_QS(tk->src), BLK_UB(tk->src),
					BLK_US(tk->dst), BLK_ES(tk->dst), BLK_QS(tk->dst), BLK_UB(tk->dst),
					BLK_US(tk->dst), (pw >= 1) ? BLK_ES(tk->dst) : dz2, (pw == 2) ? BLK_QS(tk->dst) : dz2, BLK_UB(tk->dst), W, pw);
			} else if(tk->op == ORC_ROW_OP_UPDATE){              /* dpalign_row_update_bspoa bspoa.h:2232-2261 */
				const uint32_t movx = tk->qoff_dst - tk->qoff_src;
				orc_query_t qy;
				int rh;
				orc_row_movx(mu, me, mq, mb, BLK_US(tk->src), BLK_ES(tk->src), BLK_QS(tk->src), BLK_UB(tk->src),
					W, movx, pw, nt_max, nt_min, gapo1, gape1, gapo2, gape2);
				if(movx == 0){
					if(tk->qoff_src) rh = ORC_SCORE_MIN;
					else if(type == ORC_MODE_OVERLAP || tk->toff == 0) rh = 0;
					else if(pw < 2) rh = gapo1 + gape1 * (int)tk->toff;
					else { int a = gapo1 + gape1 * (int)tk->toff, b = gapo2 + gape2 * (int)tk->toff; rh = a > b ? a : b; }
				} else if(movx <= bw) rh = mb[0];
				else rh = ORC_SCORE_MIN;
				qy.seq = queries + qoff[tk->query]; qy.len = qlen[tk->query];
				qy.mtx = (tk->prof & 1) ? mtx1 : mtx0;
				qy.hpc = (tk->prof & 2) ? 0 : 1; qy.bonus = 1;
				orc_row_cal(tk->qoff_dst, tk->base, mu, me, mq, mb,
					BLK_US(tk->dst), (pw >= 1) ? BLK_ES(tk->dst) : dz2, (pw == 2) ? BLK_QS(tk->dst) : dz2, BLK_UB(tk->dst),
					&qy, gapo1, gape1, gapo2, gape2, W, rh, pw);
			} else if(tk->op == ORC_ROW_OP_SCORE_END){           /* bspoa.h:2597-2606 */
				const int slen = (int)qlen[tk->query], rpos = (int)tk->qoff_src;
				const int smax = orc_getscore(BLK_US(tk->src), BLK_UB(tk->src), W, (uint64_t)(slen - 1 - rpos)) + T;
				if(smax > maxscr){ maxscr = smax; maxidx = (int)tk->toff; maxoff = slen - 1; }
			} else if(tk->op == ORC_ROW_OP_SCORE_TAIL){          /* bspoa.h:2547-2577 */
				const int slen = (int)qlen[tk->query], rpos = (int)tk->qoff_src;
				int mo = (slen < rpos + (int)bw ? slen : rpos + (int)bw) - 1;
				int smax = orc_getscore(BLK_US(tk->src), BLK_UB(tk->src), W, (uint64_t)(mo - rpos));
				if(slen > mo + 1){
					const int n = slen - mo - 1;
					if(pw < 2) smax += gapo1 + gape1 * n;
					else { int a = gapo1 + gape1 * n, b = gapo2 + gape2 * n; smax += a > b ? a : b; }
				}
				smax += T;
				if(smax > maxscr){ maxscr = smax; maxidx = (int)tk->toff; maxoff = mo; }
				if(type == ORC_MODE_OVERLAP){
					int32_t ms;
					const uint32_t rmax = orc_row_max(BLK_US(tk->src), BLK_UB(tk->src), W, &ms);
					if(ms > maxscr){ maxscr = ms; maxidx = (int)tk->toff; maxoff = (int)rmax + rpos; }
				}
			}
		}
		results[p].maxscr = maxscr; results[p].maxidx = maxidx; results[p].maxoff = maxoff; results[p].reserved = 0;
	}
	free(tmp);
}

/* ---------------------------------------------------------------------------------------------
 * anti-diagonal u8 DP of remsa_pedits (TEST infrastructure like everything here)
 * Follows maxmat_dp_diag_rowcal_init bspoa.h:3752-3761, maxmat_dp_diag_rowcal_prepare :3763-3787, maxmat_dp_diag_rowcal
 * :3856-3896 and the fill loop of remsa_pedit_rd_bspoacore :3925-3935.  Step i = x + y (x == y on even steps, x == y + 1 on odd
 * ones, so the window always starts at x - half / mlen - 1 - y - half) computes row i + 1 from row i:
 *   s = sat_u8(mats0[seq1[c]][c] + mats1[seq0[c]][c])      (0 where the base code is >= 4)
 *   even i ("down"):  u = r0[i][c], v = r1[i][c - 1];   odd i ("left"):  u = r0[i][c + 1], v = r1[i][c]
 *   h = max(s, u, v);   r0[i + 1][c] = h - v;   r1[i + 1][c] = h - u
 * and the two guard cells of the new row: odd i: r0[-1] = 255, the other three 0; even i: r1[16 W] = 255, the others 0.
 * --------------------------------------------------------------------------------------------- */
void orc_diagdp_fill(const uint8_t *seq0, const uint8_t *seq1, const uint8_t *const mats0[4], const uint8_t *const mats1[4],
		int mlen, int mbeg, int mend, int W, uint8_t *matrix0, uint8_t *matrix1){
	const int bw = W * 16, rowlen = bw + 2, half = bw / 2;
	{       /* init: the row of step 2 mbeg */
		uint8_t *r0 = matrix0 + (size_t)(2 * mbeg) * rowlen, *r1 = matrix1 + (size_t)(2 * mbeg) * rowlen;
		memset(r0, 0, (size_t)rowlen); memset(r1, 0, (size_t)rowlen);
		r0[1 + half - 1] = 255; r1[1 + half] = 255;
	}
	int x = mbeg, y = mbeg;
	for(int i = x + y;; i++){
		const int dir = i & 1;
		const int xb = x - half, yb = mlen - 1 - (y + half);
		const uint8_t *p0 = matrix0 + (size_t)i * rowlen + 1, *p1 = matrix1 + (size_t)i * rowlen + 1;
		uint8_t *n0 = matrix0 + (size_t)(i + 1) * rowlen + 1, *n1 = matrix1 + (size_t)(i + 1) * rowlen + 1;
		for(int c = 0; c < bw; c++){
			const int b1 = seq1[yb + c], b0 = seq0[xb + c];
			int s = (b1 < 4 ? mats0[b1][xb + c] : 0) + (b0 < 4 ? mats1[b0][yb + c] : 0);
			if(s > 255) s = 255;
			const int u = dir ? p0[c + 1] : p0[c], v = dir ? p1[c] : p1[c - 1];
			int h = s > u ? s : u;
			if(v > h) h = v;
			n0[c] = (uint8_t)(h - v); n1[c] = (uint8_t)(h - u);
		}
		if(dir){ n0[-1] = 255; n1[-1] = 0; n0[bw] = 0; n1[bw] = 0; }
		else { n0[-1] = 0; n1[-1] = 0; n0[bw] = 0; n1[bw] = 255; }
		if(dir) y++; else x++;
		if(x >= mend) break;
	}
}

/* The traceback of remsa_pedit_rd_bspoacore (bspoa.h:3965-4040) over the planes orc_diagdp_fill left, without the graph work: from
 * (mend - 1, mend - 1), per step the cell's two rows (maxmat_dp_diag_rowcal_prepare bspoa.h:3763-3787: row x + y and the one after it,
 * band cell midx), the column score h, and the reference's choice in its order -- bt 1 (x - 1) when f alone explains the cell and it is not
 * band cell 0 of an even row, bt 2 (y - 1) when e does, bt 0 (diagonal, score += s) when h does.  steps: one byte per step.
 * Returns 0, 1 (left the band: the reference aborts) or 2 (nothing explains the cell: the reference aborts). */
int orc_diagdp_walk(const uint8_t *seq0, const uint8_t *seq1, const uint8_t *const mats0[4], const uint8_t *const mats1[4],
		int mlen, int mbeg, int mend, int W, const uint8_t *matrix0, const uint8_t *matrix1, uint8_t *steps, uint32_t *nsteps, int *score, int *xend, int *yend){
	const int bw = W * 16, rowlen = bw + 2, half = bw / 2;
	int xi = mend - 1, yi = mend - 1, scr = 0, status = 0;
	uint32_t n = 0;
	while(xi >= 0 && yi >= 0){
		const int i = xi + yi;
		if(i < mbeg + mbeg) break;
		const int dir = i & 1;
		const int xx = (xi - yi - dir) / 2 + half;
		if(xx < 0 || xx >= bw){ status = 1; break; }
		/* seqs[0][xx] = _seqs[0][xi], seqs[1][xx] = _seqs[1][mlen - 1 - yi] (prepare: xb = x - midx, yb = mlen - 1 - (y + midx)) */
		const int b0 = seq0[xi], b1 = seq1[mlen - 1 - yi];
		int h = (b1 < 4 ? mats0[b1][xi] : 0) + (b0 < 4 ? mats1[b0][mlen - 1 - yi] : 0);
		if(h > 255) h = 255;
		const uint8_t *r00 = matrix0 + (size_t)i * rowlen + 1, *r01 = matrix1 + (size_t)i * rowlen + 1, *r10 = matrix0 + (size_t)(i + 1) * rowlen + 1;
		const int e = dir ? r00[xx + 1] : r00[xx], f = dir ? r01[xx] : r01[xx - 1];
		const int s = f + r10[xx];
		int bt;
		if(s == f && !(xx == 0 && dir == 0)){ bt = 1; xi--; }
		else if(s == e){ bt = 2; yi--; }
		else if(s == h){ bt = 0; scr += s; xi--; yi--; }
		else { status = 2; break; }
		steps[n++] = (uint8_t)bt;
	}
	*nsteps = n; *score = scr; *xend = xi; *yend = yi;
	return status;
}
