/*
 * bsalign_oracle_wf.c -- TEST INFRASTRUCTURE ONLY (see bsalign_oracle.h).
 *
 * Second, independent restatement of the POA's per-read seq->graph DP (align_rd_bspoacore bspoa.h:2515-2618 with
 * dpalign_row_update_bspoa :2232-2261 = row_movx + row_cal and dpalign_row_merge_bspoa :2263-2272) and of the
 * traceback into the graph (alignment2graph_bspoa bspoa.h:2274-2513), in ABSOLUTE scores on a node / in-edge
 * description of the selected sub-graph -- the formulation the device's wavefront kernel (bsalign_amd/csrc/
 * bsa_poa_wf.hip) uses.  The lane-exact restatement in bsalign_oracle.c (orc_sweep_run: the reference's striped int8
 * differences, operation by operation) stays the authority: tests/test_oracle_wf.py converts the rows computed here
 * back into the reference's block layout and compares them byte for byte with orc_sweep_run on recorded programs of
 * the real end_bspoa, and the traceback with the triples the reference itself took (oracle/_ref/libbsref_wf.so).
 *
 * Why absolute scores are the same thing (inside the no-saturation guard of bsa_align8_x_supported): a row block
 * stores u[p] = H(p) - H(p-1), e[p] = E(p) - H(p), q[p] = Q(p) - H(p) and ubegs[j] = H(j W - 1) (ubegs[0] = H(0) after
 * the re-basing of bsalign.h:2632-2633, with u[0] = 0); row_movx shifts the row and continues it with synthetic gap
 * cells (bsalign.h:2347-2391); row_cal is the affine / two-piece recurrence with three rules of its own -- the seed
 * of band cell 0 (bsalign.h:2899-2907), F and G restarting from "H of the previous row - 63" at every running block
 * (bsalign.h:2909-2931 + the F-penetration :2639-2652), S = -63 beyond the read end (:2157-2160) -- and row_merge is
 * the cell-wise maximum of H, E and Q (bsalign.h:2474-2616).  Maximum and the F / G chains commute, so a node with
 * several in-edges is computed in one pass over the maximum of its candidates.
 */
#include "bsalign_oracle.h"
#include <stdlib.h>
#include <string.h>

#define NL ORC_LANES
static inline int imax(int a, int b){ return a > b ? a : b; }
static inline int imin(int a, int b){ return a < b ? a : b; }

typedef struct {
	int mode; uint32_t bw, W; int pw;
	int M, X, refbonus, O, E, Q, P, T;
	int nt_max, nt_min;
	const uint8_t *query; uint32_t slen;
} wf_ctx_t;

static void wf_ctx(wf_ctx_t *c, const orc_wf_params_t *par, const uint8_t *query, uint32_t slen){
	c->mode = par->mode & 3; c->bw = (par->bandwidth + NL - 1) / NL * NL; c->W = c->bw / NL;
	c->M = par->M; c->X = par->X; c->refbonus = par->refbonus; c->O = par->gapo1; c->E = par->gape1; c->Q = par->gapo2; c->P = par->gape2; c->T = par->T;
	c->pw = orc_get_piecewise(c->O, c->E, c->Q, c->P, (int)c->bw);
	c->nt_max = c->M + c->refbonus + 1; c->nt_min = c->X;           /* bspoa.h:2226, 2240 */
	c->query = query; c->slen = slen;
}

/* S(x) of the profile (v.base == u.base) * 2 + v.bonus (bspoa.h:2199-2215, 2588; bsalign.h:2166-2221) */
static inline int wf_score(const wf_ctx_t *c, uint32_t x, int base, int bonus, int same){
	int s;
	if(x >= c->slen) return ORC_EPI8_MIN;
	s = (c->query[x] == base) ? c->M + (bonus ? c->refbonus : 0) : c->X;
	if(!same && x + 1 < c->slen && c->query[x] != c->query[x + 1]) s += 1;
	return s;
}

/* the head's row: row_init (bsalign.h:2094-2140) turned into absolute cells; *u0 = its ubegs[0] */
void orc_wf_init_row(const orc_wf_params_t *par, orc_wf_cell_t *row, int32_t *u0){
	wf_ctx_t c; uint32_t p;
	int8_t *us, *es, *qs; int32_t ub[NL + 1];
	wf_ctx(&c, par, NULL, 0);
	us = (int8_t*)calloc(3 * (size_t)c.bw, 1); es = us + c.bw; qs = es + c.bw;
	orc_row_init(us, es, qs, ub, c.mode, c.bw, c.nt_max, c.nt_min, c.O, c.E, c.Q, c.P);
	for(p = 0; p < c.bw; p++){
		row[p].h = orc_getscore(us, ub, c.W, p);
		row[p].e = c.pw >= 1 ? es[(p % c.W) * NL + p / c.W] : 0;
		row[p].q = c.pw == 2 ? qs[(p % c.W) * NL + p / c.W] : 0;
		row[p].tag = 0;
	}
	*u0 = ub[0];
	free(us);
}

/* cell p of row u after row_movx(movx): real cell, synthetic overhang (bsalign.h:2357-2389) or the dead row of movx >= bw (:2253-2259) */
static inline void wf_post(const wf_ctx_t *c, const orc_wf_cell_t *ru, uint32_t movx, uint32_t p, int *h, int *e, int *q){
	if(movx >= c->bw){ *h = ORC_SCORE_MIN; *e = 0; *q = 0; return; }
	if(p + movx < c->bw){ *h = ru[p + movx].h; *e = ru[p + movx].e; *q = ru[p + movx].q; return; }
	{
		const int k = (int)(p + movx - c->bw);                        /* 0 = first synthetic cell */
		const int goe = (c->pw == 2) ? c->Q + c->P : c->O + c->E;
		const int d = (c->pw == 2) ? (c->O - c->Q) / (c->P - c->E) : (int)c->bw + 1;
		int v = ru[c->bw - 1].h + imin(c->nt_min, goe) - 1 - c->nt_max + goe;
		if(k < d) v += k * c->E; else v += (d - 1) * c->E + (k - d + 1) * c->P;
		*h = v; *e = 0; *q = 0;
	}
}

/* rows[i * bw + p] for node i, u0[i]; node 0 is the head (orc_wf_init_row).  Every input of a node has a lower index.  A node's
 * forward view is at most two inputs: the row of `src` moved by movx and extended by one DP row, or (MERGE) the finished row of a
 * partial node at the same band offset, taken as it is (row_merge = cell-wise maximum, bsalign.h:2474-2616). */
void orc_wf_forward(const orc_wf_node_t *nodes, uint32_t nnodes, const uint8_t *query, uint32_t slen,
		const orc_wf_params_t *par, orc_wf_cell_t *rows, int32_t *u0){
	wf_ctx_t c; uint32_t i, p, k;
	const int NEG = ORC_SCORE_MIN * 2;
	wf_ctx(&c, par, query, slen);
	if(nnodes == 0) return;
	orc_wf_init_row(par, rows, u0);
	for(i = 1; i < nnodes; i++){
		const orc_wf_node_t *v = nodes + i;
		orc_wf_cell_t *rv = rows + (size_t)i * c.bw;
		int F = NEG, G = NEG;
		for(p = 0; p < c.bw; p++){
			const uint32_t x = v->rpos + p;
			int m = NEG, Ein = NEG, Qin = NEG, fl = NEG, HX = NEG, EX = NEG, QX = NEG, H, e1, q1;
			for(k = 0; k < 2; k++){
				const orc_wf_input_t *in = v->in + k;
				const orc_wf_cell_t *ru;
				int h1, ee, qq, hp, ep, qp, mc;
				if(!(in->toff_kind & ORC_WF_IN_PRESENT)) continue;
				ru = rows + (size_t)in->src * c.bw;
				if(in->toff_kind & ORC_WF_IN_MERGE){
					HX = imax(HX, ru[p].h); EX = imax(EX, ru[p].h + ru[p].e); QX = imax(QX, ru[p].h + ru[p].q);
					continue;
				}
				{
					const uint32_t movx = in->movx, urpos = v->rpos - movx, toff = in->toff_kind & ORC_WF_IN_TOFF;
					const int same = (in->toff_kind & ORC_WF_IN_SAME) != 0, S = wf_score(&c, x, v->base, v->flags & 1, same);
					wf_post(&c, ru, movx, p, &h1, &ee, &qq);
					if(p == 0){
						/* seed of band cell 0 (bsalign.h:2899-2907) with rh as dpalign_row_update_bspoa picks it (bspoa.h:2242-2254) */
						int ub0, rh, h0, t;
						if(movx == 0) ub0 = u0[in->src];
						else if(movx < c.bw) ub0 = ru[movx - 1].h;
						else ub0 = ORC_SCORE_MIN;
						if(movx == 0){
							if(urpos) rh = ORC_SCORE_MIN;
							else if(c.mode == ORC_MODE_OVERLAP || toff == 0) rh = 0;
							else if(c.pw < 2) rh = c.O + c.E * (int)toff;
							else rh = imax(c.O + c.E * (int)toff, c.Q + c.P * (int)toff);
						} else if(movx <= c.bw) rh = ub0;
						else rh = ORC_SCORE_MIN;
						h0 = rh - ub0 + S;
						t = (h1 - ub0) + (c.pw == 0 ? c.E : c.pw == 1 ? ee : imax(ee, qq));
						if(h0 >= t){ if(h0 > ORC_EPI8_MAX) h0 = ORC_EPI8_MAX; } else h0 = ORC_EPI8_MIN;
						mc = ub0 + h0;
						fl = imax(fl, ub0 + ORC_EPI8_MIN);
					} else {
						wf_post(&c, ru, movx, p - 1, &hp, &ep, &qp);
						mc = hp + S;
						if(p % c.W == 0) fl = imax(fl, hp + ORC_EPI8_MIN);
					}
					m = imax(m, mc);
					Ein = imax(Ein, h1 + (c.pw == 0 ? c.E : ee));
					if(c.pw == 2) Qin = imax(Qin, h1 + qq);
				}
			}
			if(p % c.W == 0){ F = imax(F, fl); if(c.pw == 2) G = imax(G, fl); }
			H = imax(imax(m, Ein), imax(F, HX));
			if(c.pw == 2) H = imax(imax(H, Qin), G);
			if(c.pw == 0){
				e1 = 0; q1 = 0;
				F = H + c.E;
			} else {
				e1 = imax(imax(Ein + c.E, H + c.O + c.E), EX) - H;
				F = imax(F + c.E, H + c.O + c.E);
				q1 = 0;
				if(c.pw == 2){
					q1 = imax(imax(Qin + c.P, H + c.Q + c.P), QX) - H;
					G = imax(G + c.P, H + c.Q + c.P);
				}
			}
			rv[p].h = H; rv[p].e = (int8_t)e1; rv[p].q = (int8_t)q1; rv[p].tag = 0;
		}
		u0[i] = rv[0].h;
	}
}

/* absolute row -> the reference's row block (bspoa.h:1787-1793): us | es | qs | int32 ubegs[17] */
void orc_wf_row_to_block(const orc_wf_cell_t *row, int32_t u0, uint32_t bandwidth, int pw, uint8_t *block){
	const uint32_t bw = (bandwidth + NL - 1) / NL * NL, W = bw / NL;
	int8_t *us = (int8_t*)block, *es = us + bw, *qs = es + bw;
	int32_t *ub = (int32_t*)(block + (size_t)bw * (pw + 1));
	uint32_t p;
	for(p = 0; p < bw; p++){
		const uint32_t s = (p % W) * NL + p / W;
		const int prev = (p == 0) ? u0 : row[p - 1].h;
		us[s] = (int8_t)(row[p].h - prev);
		if(pw >= 1) es[s] = row[p].e;
		if(pw == 2) qs[s] = row[p].q;
	}
	ub[0] = u0;
	for(p = 1; p <= NL; p++) ub[p] = row[p * W - 1].h;
}

/* row_max (bsalign.h:3213-3329) on an absolute row: best cell of every running block (first one on ties), blocks
 * compared in the order of the reference's register reduction */
static uint32_t wf_row_max(const orc_wf_cell_t *row, uint32_t W, int32_t *best){
	static const int order[NL] = {0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15};
	int bs = 0, k; uint32_t bp = 0;
	for(k = 0; k < NL; k++){
		const uint32_t j = (uint32_t)order[k]; uint32_t i, ai = 0;
		int mx = row[j * W].h;
		for(i = 1; i < W; i++) if(row[j * W + i].h > mx){ mx = row[j * W + i].h; ai = i; }
		if(k == 0 || mx > bs){ bs = mx; bp = j * W + ai; }
	}
	*best = bs;
	return bp;
}

/* the end-of-alignment candidates in the reference's visiting order (bspoa.h:2549-2578, 2593-2603): strictly greater replaces */
void orc_wf_best(const orc_wf_node_t *nodes, const orc_wf_cand_t *cands, uint32_t ncands, uint32_t slen, const orc_wf_params_t *par,
		const orc_wf_cell_t *rows, orc_sweep_result_t *res){
	wf_ctx_t c; uint32_t k;
	int maxscr = ORC_SCORE_MIN, maxidx = -1, maxoff = -1;
	wf_ctx(&c, par, NULL, slen);
	for(k = 0; k < ncands; k++){
		const orc_wf_node_t *u = nodes + cands[k].node;
		const orc_wf_cell_t *ru = rows + (size_t)cands[k].node * c.bw;
		const int rpos = (int)u->rpos, sl = (int)slen;
		if(cands[k].kind == 1){
			const int s = ru[sl - 1 - rpos].h + c.T;
			if(s > maxscr){ maxscr = s; maxidx = (int)cands[k].node; maxoff = sl - 1; }
		} else {
			const int mo = imin(sl, rpos + (int)c.bw) - 1;
			int s = ru[mo - rpos].h;
			if(sl > mo + 1){
				const int n = sl - mo - 1;
				s += (c.pw < 2) ? c.O + c.E * n : imax(c.O + c.E * n, c.Q + c.P * n);
			}
			s += c.T;
			if(s > maxscr){ maxscr = s; maxidx = (int)cands[k].node; maxoff = mo; }
			if(c.mode == ORC_MODE_OVERLAP){
				int32_t ms; const uint32_t rm = wf_row_max(ru, c.W, &ms);
				if(ms > maxscr){ maxscr = ms; maxidx = (int)cands[k].node; maxoff = (int)rm + rpos; }
			}
		}
	}
	res->maxscr = maxscr; res->maxidx = maxidx; res->maxoff = maxoff; res->reserved = 0;   /* maxidx: LOCAL node index */
}

/* alignment2graph_bspoa (bspoa.h:2274-2513) as a walk that only reports: one event per step the reference takes,
 * (local node, x, bt) with bt = 0 M, 1 I, 2 D, 4 D2 (bsalign.h:40-50); the merges, cpos updates and counters are the
 * binding's (include/bsalign_poa_adapter.h).  fin[0..1] = node and x at which the walk stopped (rs.tb, rs.qb).
 * Returns the number of events, -1 when the walk leaves the stored band (the reference reads outside its row there),
 * -2 when cap is too small. */
long orc_wf_trace(const orc_wf_node_t *nodes, const orc_wf_edge_t *edges, const uint8_t *query, uint32_t slen,
		const orc_wf_params_t *par, const orc_wf_cell_t *rows, const int32_t *u0, uint32_t head, uint32_t midx, int xe,
		orc_wf_event_t *ev, long cap, int32_t fin[2]){
	wf_ctx_t c; long ne = 0;
	uint32_t nidx = midx, n = midx, bt = 0xFFFFFFFFu;
	int x = xe, Hs[3];
	wf_ctx(&c, par, query, slen);
#define ROW(i) (rows + (size_t)(i) * c.bw)
#define US(i, p) ((p) == 0 ? ROW(i)[0].h - u0[i] : ROW(i)[p].h - ROW(i)[(p) - 1].h)
#define EMIT(nn, xx, bb) do{ if(ne >= cap) return -2; ev[ne].node = (nn); ev[ne].x = (xx); ev[ne].bt = (bb); ne++; }while(0)
	if(x - (int)nodes[n].rpos < 0 || x - (int)nodes[n].rpos >= (int)c.bw) return -1;
	Hs[0] = 0; Hs[1] = ROW(n)[x - (int)nodes[n].rpos].h; Hs[2] = 0;
	while(1){
		if(n == head || x < 0) break;
		if(bt == 2 || bt == 4){
			uint32_t k, found = 0;
			EMIT(n, x, bt);
			for(k = 0; k < nodes[n].n_in; k++){
				const orc_wf_edge_t *ed = edges + nodes[n].first_in + k;
				const uint32_t w = ed->src; const int wr = (int)nodes[w].rpos;
				int q;
				if(x < wr || x >= wr + (int)c.bw) continue;
				Hs[0] = ROW(w)[x - wr].h;
				if(bt == 2) q = c.pw ? ROW(w)[x - wr].e : c.O + c.E;
				else q = ROW(w)[x - wr].q;
				if(Hs[0] + q != Hs[1]) continue;
				n = w;
				if(q == ((bt == 2) ? c.O + c.E : c.Q + c.P)){ bt = 0xFFFFFFFFu; Hs[1] = Hs[0]; Hs[2] = 0; }
				else { Hs[1] -= (bt == 2) ? c.E : c.P; Hs[2] ++; }
				found = 1;
				break;
			}
			(void)found;         /* not found: the reference loops on the same state for ever (bspoa.h:2351-2357 is DEBUG only) */
			if(!found) return -1;
			continue;
		} else if(bt == 1){
			int t;
			EMIT(n, x, bt);
			t = (c.pw == 2) ? imax(c.O + c.E * Hs[2], c.Q + c.P * Hs[2]) : c.O + c.E * Hs[2];
			x --;
			if(Hs[0] + t == Hs[1]){ bt = 0xFFFFFFFFu; Hs[1] = Hs[0]; Hs[2] = 0; }
			else if(x >= 0){
				const int p = x - (int)nodes[n].rpos;
				if(p < 0) return -1;
				Hs[0] -= US(n, p);
				Hs[2] ++;
			}
			continue;
		} else if(bt == 0){
			EMIT(n, x, bt);
			x --;
			n = nidx;
			bt = 0xFFFFFFFFu;
		} else {
			uint32_t k, btc = 0, bti = 0xFFFFFFFFu, bnode = 0; int bh = 0;
			for(k = 0; k < nodes[n].n_in; k++){
				const orc_wf_edge_t *ed = edges + nodes[n].first_in + k;
				const uint32_t w = ed->src, cov = ed->cov; const int wr = (int)nodes[w].rpos;
				int ft = 0, s, scr[3], i, p;
				if(x < wr || x > (int)c.bw + wr) continue;
				else if(x == (int)c.bw + wr){ Hs[0] = ROW(w)[x - wr - 1].h; ft |= (1 << 2) | (1 << 4); }
				else if(x == wr){
					Hs[0] = u0[w];
					if(wr == 0 && (c.mode == ORC_MODE_OVERLAP || w == head)) ft |= 1 << 15;
					else ft |= 1 << 0;
				} else Hs[0] = ROW(w)[x - wr - 1].h;
				s = wf_score(&c, (uint32_t)x, nodes[n].base, nodes[n].flags & 1, nodes[w].base == nodes[n].base);
				if(ft & (1 << 15)) s -= u0[w];
				p = x - wr;
				scr[0] = (ft & (1 << 0)) ? ORC_SCORE_MIN : s;
				scr[1] = (ft & (1 << 2)) ? ORC_SCORE_MIN : US(w, p) + (c.pw ? ROW(w)[p].e : c.E);
				scr[2] = (ft & (1 << 4)) ? ORC_SCORE_MIN : (c.pw == 2 ? US(w, p) + ROW(w)[p].q : -ORC_SCORE_MIN);
				for(i = 0; i < 3; i++){
					if(Hs[0] + scr[i] == Hs[1]){
						if(cov > btc || (cov == btc && i == 0 && (bti & 0xFF) != 0)){ bti = (uint32_t)i; btc = cov; bnode = w; bh = Hs[0]; }
					}
				}
			}
			if(bti == 0xFFFFFFFFu){
				const int p = x - (int)nodes[n].rpos;
				if(p < 0 || p >= (int)c.bw) return -1;
				bt = 1; Hs[2] = 1;
				Hs[0] = Hs[1] - US(n, p);
			} else if(bti == 0){ bt = 0; nidx = bnode; Hs[1] = bh; Hs[2] = 0; }
			else if(bti == 1){ bt = 2; Hs[2] = 1; }
			else { bt = 4; Hs[2] = 1; }
		}
	}
	fin[0] = (int32_t)n; fin[1] = x;
	return ne;
#undef ROW
#undef US
#undef EMIT
}

/* the three steps above behind the signature of the binding's graph backend (bsa_poa_graph_backend_fn of
 * include/bsalign_poa_adapter.h): what the device's bsa_poa_graph_host does, on the CPU, for the tests */
int orc_wf_backend(void *user, const orc_wf_node_t *nodes, size_t nnodes, const orc_wf_edge_t *edges, size_t nedges,
		const orc_wf_cand_t *cands, size_t ncands, const uint8_t *query, uint32_t slen,
		const orc_wf_params_t *par, orc_wf_result_t *res, orc_wf_event_t *events, size_t events_cap){
	const uint32_t bw = (par->bandwidth + NL - 1) / NL * NL;
	orc_wf_cell_t *rows;
	int32_t *u0, fin[2] = {-1, -1};
	orc_sweep_result_t best;
	long n;
	(void)user; (void)nedges;
	/* (round 6: any width -- the device takes bands above 256 columns through its generic-width kernel, bsa_poa_gen.hip) */
	rows = (orc_wf_cell_t*)malloc((size_t)nnodes * bw * sizeof(orc_wf_cell_t));
	u0 = (int32_t*)malloc((size_t)nnodes * sizeof(int32_t));
	orc_wf_forward(nodes, (uint32_t)nnodes, query, slen, par, rows, u0);
	orc_wf_best(nodes, cands, (uint32_t)ncands, slen, par, rows, &best);
	memset(res, 0, sizeof(*res));
	res->maxscr = best.maxscr; res->maxidx = best.maxidx; res->maxoff = best.maxoff;
	res->fin_node = -1; res->fin_x = -1;
	if(best.maxidx < 0) res->status = 3;
	else {
		n = orc_wf_trace(nodes, edges, query, slen, par, rows, u0, 0, (uint32_t)best.maxidx, best.maxoff, events, (long)events_cap, fin);
		if(n == -1) res->status = 1; else if(n == -2) res->status = 2;
		else { res->nevents = (int32_t)n; res->fin_node = fin[0]; res->fin_x = fin[1]; }
	}
	free(rows); free(u0);
	return 0;
}
