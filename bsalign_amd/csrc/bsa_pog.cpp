// bsa_pog.cpp -- the POA's own graph surface (include/bsalign_poa.h), host C++.
//
// What the reference does with a BSPOA object around its per-read DP (bspoa.h; line numbers of /root/reference):
//   graph container            :28-46 nodes in rings of aligned bases (header / next / prev), edge pairs in coverage-ordered lists
//   new_node / _add_read       :394-406, :916-951
//   edge lists                 :428-606 get_edge, new_edge, _add_edge (ordered insert), _del_edge, chg_edge (remove + re-insert)
//   connect_rdnode             :622-631, merge_nodes :797-894 (+ _mov_node_edges :689-735)
//   sel_nodes_bspoa            :1887-2020
//   prepare_rd_align_bspoa     :2022-2230 (without the four query profiles and the row arena: the device kernel builds its own)
//   the traversal order of align_rd_bspoacore :2515-2618 (what include/bsalign_poa_adapter.h::bsa_poa_flatten_graph read off the
//                                reference's graph: here it runs on this container)
//   the surgery of alignment2graph_bspoa :2393-2405, :2501-2511 and the end of align_rd_bspoa :2652-2657
// Own organisation: structure-of-arrays nodes, ONE record per edge with both list links (the reference keeps a forward and a
// reverse half), per-read scratch kept in arrays indexed by node and reset through the selection list, the selection, band
// placement and program building as three passes over flat arrays.  What must be the reference's, because results depend on it:
// node numbering (read r's position p is node ndoff[r] + p), the ORDER of every edge list (coverage descending, a changed edge
// re-inserted behind its equals), the order of the selection list, the traversal order, the uint32 arithmetic of the column map.
#include "../../include/bsalign_poa.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace {

struct Edge { uint32_t from, to, cov, next_out, next_in; };       // edge 0 is "none"

inline double now_s(){ return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// No exception crosses the C ABI: the containers below allocate (a window's graph grows with every read), and a failed allocation is BSA_E_NOMEM
// to the caller, who may bsa_pog_abort / bsa_pog_clear and go on.
template<class F> inline int guarded(F &&f){
	try { return f(); }
	catch(const std::bad_alloc&){ return BSA_E_NOMEM; }
	catch(...){ return BSA_E_ARG; }
}

}  // namespace

struct bsa_pog {
	bsa_pog_params_t par;
	// ---- the graph (P0)
	std::vector<uint32_t> header, next, prev, out, in;
	std::vector<int32_t> pos, cpos;
	std::vector<uint16_t> rid, cov;
	std::vector<uint8_t> base, flags;
	std::vector<Edge> edges;
	std::vector<uint32_t> efree;
	std::vector<uint32_t> ndoff, rdlen;
	uint32_t HEAD = 0, TAIL = 0;
	// ---- one read's alignment (scratch indexed by node is only valid for selected nodes)
	std::vector<uint8_t> sel_bit, bonus;
	std::vector<uint32_t> nct, vst, loc, voff;
	std::vector<int32_t> rpos, mpos;
	std::vector<uint32_t> sels, stack;
	std::vector<int32_t> rdreg[2];
	std::vector<uint64_t> todels;                     // auxiliary edges (from << 32 | to), in the order they were added
	std::vector<uint32_t> rmap;
	std::vector<uint8_t> qseq;                         // the read, one base per byte
	struct Visit { uint32_t src, toff; };
	std::vector<Visit> visits;
	std::vector<bsa_poa_node_t> pnodes;
	std::vector<bsa_poa_edge_t> pedges;
	std::vector<bsa_poa_cand_t> pcands;
	std::vector<bsa_poa_event_t> events;
	bsa_poa_result_t res;
	bsa_pog_read_t rd;
	uint32_t cur_rid = 0, cur_rbeg = 0;
	bool realn = false;                               // bsa_pog_cut ran: the next selection is over every read (bspoa.h:2636-2642)
	int stage = 0;                                    // 0 idle, 1 selected, 2 placed, 3 program built, 4 run
	double secs[5] = {0, 0, 0, 0, 0};

	uint32_t nnodes() const { return (uint32_t)header.size(); }
	uint32_t rdnode(uint32_t r, int p) const { return (uint32_t)((int64_t)ndoff[r] + p); }

	// ---- nodes (new_node_bspoa, bspoa.h:394-406)
	uint32_t new_node(uint16_t r, int p, uint8_t b){
		const uint32_t i = nnodes();
		header.push_back(i); next.push_back(i); prev.push_back(i); out.push_back(0); in.push_back(0);
		pos.push_back(p); cpos.push_back(0); rid.push_back(r); cov.push_back(1); base.push_back(b); flags.push_back(0);
		return i;
	}
	// ---- edge lists.  A list is ordered by coverage, descending; a new edge goes in front of the first edge with a SMALLER coverage
	// (bspoa.h:464-494), i.e. behind its equals.
	void link(uint32_t e){
		Edge &E = edges[e];
		{
			uint32_t *p = &out[E.from];
			while(*p && edges[*p].cov >= E.cov) p = &edges[*p].next_out;
			E.next_out = *p; *p = e;
		}
		{
			uint32_t *p = &in[E.to];
			while(*p && edges[*p].cov >= E.cov) p = &edges[*p].next_in;
			E.next_in = *p; *p = e;
		}
	}
	void unlink(uint32_t e){
		Edge &E = edges[e];
		uint32_t *p = &out[E.from];
		while(*p != e) p = &edges[*p].next_out;
		*p = E.next_out;
		p = &in[E.to];
		while(*p != e) p = &edges[*p].next_in;
		*p = E.next_in;
		E.next_out = E.next_in = 0;
	}
	uint32_t find_edge(uint32_t u, uint32_t v) const {
		for(uint32_t e = out[u]; e; e = edges[e].next_out) if(edges[e].to == v) return e;
		return 0;
	}
	// chg_edge_bspoa (bspoa.h:561-606): the coverage of u -> v (headers of the given nodes) changes by d; the edge leaves its lists and, if it
	// still has coverage, comes back at the place its new coverage gives it.  Returns whether the edge existed.
	bool chg_edge(uint32_t _u, uint32_t _v, int d){
		if(d == 0) return false;
		const uint32_t u = header[_u], v = header[_v];
		if(u == v) return false;
		uint32_t e = find_edge(u, v);
		const bool existed = e != 0;
		int ncov = d;
		if(e){ ncov = (int)edges[e].cov + d; unlink(e); efree.push_back(e); }
		if(ncov > 0){
			if(!efree.empty()){ e = efree.back(); efree.pop_back(); }
			else { e = (uint32_t)edges.size(); edges.push_back(Edge()); }
			edges[e].from = u; edges[e].to = v; edges[e].cov = (uint32_t)ncov; edges[e].next_out = edges[e].next_in = 0;
			link(e);
		}
		return existed;
	}
	// connect_rdnode_bspoa (bspoa.h:622-631)
	void connect_rdnode(uint32_t r, int p){
		const uint32_t u = rdnode(r, p - 1), v = rdnode(r, p);
		if(flags[v] & BSA_POG_F_RDC) return;
		chg_edge(u, v, 1);
		flags[u] |= BSA_POG_F_RDD; flags[v] |= BSA_POG_F_RDC;
	}
	// _mov_node_edges_bspoacore (bspoa.h:683-735): the edges of header `a` on one side (dir 0: out, 1: in) are split between `a` and header `b`.  movtype is the
	// reference's nibble matrix: what of an edge's coverage stays (column 1) or moves (column 0), for ordinary edges (row 0) and for the edge to / from the
	// ring of `spec` (row 1) -- F: all of it, E: all but one, 1: one, 0: nothing.  MOVALL 0x0F0F, KPTONE 0x1E0F (the spec edge keeps one), MOVONE 0xE1F0 (one
	// of the spec edge moves, everything else stays).  The changes are listed first, then applied in list order.
	static constexpr int MOVALL = 0x0F0F, KPTONE = 0x1E0F, MOVONE = 0xE1F0;
	void move_edges(uint32_t a, uint32_t b, int dir, uint32_t spec = 0xFFFFFFFFu, int movtype = MOVALL){
		const uint32_t spec_node = spec < nnodes() ? header[spec] : spec;
		struct Chg { uint32_t other; int keep_delta, moved; };
		std::vector<Chg> lst;
		for(uint32_t e = dir ? in[a] : out[a]; e; e = dir ? edges[e].next_in : edges[e].next_out){
			const uint32_t other = dir ? edges[e].from : edges[e].to;
			const int ecov = (int)edges[e].cov;
			int covs[4] = {other == spec_node ? 0 : ecov, other == spec_node ? ecov : 0, 0, 0};
			for(int i = 0; i < 2; i++) for(int j = 0; j < 2; j++){
				switch((movtype >> (4 * (i * 2 + j))) & 0xF){
					case 0xF: covs[3 - j] += covs[i]; break;
					case 0xE: covs[3 - j] += std::max(covs[i] - 1, 0); break;
					case 0x1: covs[3 - j] += std::min(covs[i], 1); break;
					default: break;
				}
			}
			lst.push_back({other, covs[2] - ecov, covs[3]});
		}
		for(const Chg &x : lst){
			if(dir){ chg_edge(x.other, a, x.keep_delta); chg_edge(x.other, b, x.moved); }
			else { chg_edge(a, x.other, x.keep_delta); chg_edge(b, x.other, x.moved); }
		}
	}
	// disconnect_rdnode_bspoa (bspoa.h:640-649)
	void disconnect_rdnode(uint32_t r, int p){
		const uint32_t u = rdnode(r, p - 1);
		if(!(flags[u] & BSA_POG_F_RDD)) return;
		const uint32_t v = rdnode(r, p);
		chg_edge(u, v, -1);
		flags[u] &= (uint8_t)~BSA_POG_F_RDD; flags[v] &= (uint8_t)~BSA_POG_F_RDC;
	}
	// cut_rdnode_bspoa(.., BSPOA_RDNODE_CUTALL) (bspoa.h:741-795): the base leaves its ring -- taking along one unit of the two edges that chain it to its
	// read's neighbours, or, when it was the ring's representative, leaving everything but those to the ring's new representative (its former `prev`) --
	// and is unchained from its read
	void cut_rdnode(uint32_t r, int p){
		const uint32_t u = rdnode(r, p);
		const uint32_t nb[2] = {u + 1u, u - 1u};
		const uint32_t h0 = header[u], h1 = prev[u];
		const uint16_t nodecov = cov[h0];
		if(next[u] != u){
			next[prev[u]] = next[u]; prev[next[u]] = prev[u];
			next[u] = prev[u] = header[u] = u;
			uint32_t x;
			if(h0 == u){
				x = h1;
				for(;;){ header[x] = h1; if(next[x] == h1) break; x = next[x]; }
				x = h1;
				move_edges(u, x, 0, nb[0], (flags[u] & BSA_POG_F_RDD) ? KPTONE : MOVALL);
				move_edges(u, x, 1, nb[1], (flags[u] & BSA_POG_F_RDC) ? KPTONE : MOVALL);
			} else {
				x = h0;
				if(flags[u] & BSA_POG_F_RDD) move_edges(x, u, 0, nb[0], MOVONE);
				if(flags[u] & BSA_POG_F_RDC) move_edges(x, u, 1, nb[1], MOVONE);
			}
			cov[header[x]] = (uint16_t)(nodecov - 1);
			cov[header[u]] = 1;
		}
		disconnect_rdnode(r, p);
		disconnect_rdnode(r, p + 1);
	}
	// merge_nodes_bspoa (bspoa.h:797-894): the rings of n1 and n2 become one; the representative is the ring with more bases, the lower read on a tie
	uint32_t merge_nodes(uint32_t n1, uint32_t n2){
		uint32_t a = header[n1], b = header[n2];
		if(a == b) return a;
		const uint16_t total = (uint16_t)(cov[a] + cov[b]);
		if(cov[a] < cov[b]) std::swap(a, b);
		else if(cov[a] == cov[b] && rid[a] > rid[b]) std::swap(a, b);
		if(out[b]) move_edges(b, a, 0);
		if(in[b]) move_edges(b, a, 1);
		cov[a] = total;
		uint32_t x = b;
		do { header[x] = a; x = next[x]; } while(x != b);
		const uint32_t pa = prev[a], pb = prev[b];
		prev[a] = pb; prev[b] = pa; next[pb] = a; next[pa] = b;
		return a;
	}

	void grow_scratch(){
		const size_t n = header.size();
		if(sel_bit.size() < n){
			sel_bit.resize(n, 0); bonus.resize(n, 0); nct.resize(n, 0); vst.resize(n, 0); loc.resize(n, 0); voff.resize(n, 0); rpos.resize(n, 0); mpos.resize(n, 0);
		}
	}
	void drop_aux(){
		for(uint64_t t : todels) chg_edge((uint32_t)(t >> 32), (uint32_t)(t & 0xFFFFFFFFu), -1);
		todels.clear();
	}
};

extern "C" int bsa_pog_create(const bsa_pog_params_t *par, bsa_pog_t **out){
	return guarded([&]() -> int {
	if(!par || !out) return BSA_E_ARG;
	if(par->alnmode < 0 || par->alnmode > 2 || par->bandwidth < 0 || par->nrec < 0 || par->seqcore < 0) return BSA_E_ARG;
	bsa_pog *g = new (std::nothrow) bsa_pog();
	if(!g) return BSA_E_NOMEM;
	g->par = *par;
	g->edges.push_back(Edge());
	memset(&g->res, 0, sizeof(g->res)); memset(&g->rd, 0, sizeof(g->rd));
	*out = g;
	return BSA_OK;
	});
}

extern "C" void bsa_pog_destroy(bsa_pog_t *g){ delete g; }

extern "C" void bsa_pog_clear(bsa_pog_t *g){
	if(!g) return;
	g->header.clear(); g->next.clear(); g->prev.clear(); g->out.clear(); g->in.clear(); g->pos.clear(); g->cpos.clear(); g->rid.clear(); g->cov.clear();
	g->base.clear(); g->flags.clear(); g->edges.resize(1); g->efree.clear(); g->ndoff.clear(); g->rdlen.clear(); g->HEAD = g->TAIL = 0;
	g->sel_bit.clear(); g->bonus.clear(); g->nct.clear(); g->vst.clear(); g->loc.clear(); g->voff.clear(); g->rpos.clear(); g->mpos.clear();
	g->sels.clear(); g->todels.clear(); g->stage = 0; g->realn = false;
}

extern "C" int bsa_pog_add_read(bsa_pog_t *g, const uint8_t *bases, uint32_t len, uint32_t *rid_out){
	return guarded([&]() -> int {
	if(!g || (len && !bases) || g->stage != 0) return BSA_E_ARG;
	if(g->ndoff.size() >= 0x3FFFu || len > 0x0FFFFFFFu) return BSA_E_ARG;                    // BSPOA_RDCNT_MAX, BSPOA_RDLEN_MAX (bspoa.h:22-23)
	for(uint32_t i = 0; i < len; i++) if(bases[i] > 3) return BSA_E_ARG;
	const uint32_t r = (uint32_t)g->ndoff.size();
	g->new_node((uint16_t)r, -1, 4);
	g->ndoff.push_back(g->nnodes());
	g->rdlen.push_back(len);
	for(uint32_t i = 0; i < len; i++) g->new_node((uint16_t)r, (int)i, bases[i]);
	g->new_node((uint16_t)r, (int)len, 4);
	if(r == 0){
		// the backbone: its bases keep the bonus (bless), its column is its position, and it is chained at once (bspoa.h:925-941)
		g->HEAD = g->ndoff[0] - 1; g->TAIL = g->ndoff[0] + len;
		g->cpos[g->HEAD] = 0; g->cpos[g->TAIL] = (int32_t)len;
		for(uint32_t i = 0; i < len; i++){
			const uint32_t v = g->rdnode(0, (int)i);
			g->flags[v] |= BSA_POG_F_REF | BSA_POG_F_BLESS; g->cpos[v] = (int32_t)i;
			g->connect_rdnode(0, (int)i);
		}
		g->connect_rdnode(0, (int)len);
	} else {
		g->merge_nodes(g->HEAD, g->rdnode(r, -1));
		g->merge_nodes(g->TAIL, g->rdnode(r, (int)len));
	}
	if(rid_out) *rid_out = r;
	return BSA_OK;
	});
}

extern "C" int bsa_pog_import(bsa_pog_t *g, const bsa_pog_snapshot_t *s, const uint8_t *const *read_bases){
	return guarded([&]() -> int {
	(void)read_bases;
	if(!g || !s || !s->nodes || !s->ndoff || !s->rdlen || !s->out_off || !s->in_off) return BSA_E_ARG;
	bsa_pog_clear(g);
	const uint32_t n = s->nnodes;
	if(s->head >= n || s->tail >= n) return BSA_E_ARG;
	g->header.resize(n); g->next.resize(n); g->prev.resize(n); g->out.assign(n, 0); g->in.assign(n, 0); g->pos.resize(n); g->cpos.resize(n);
	g->rid.resize(n); g->cov.resize(n); g->base.resize(n); g->flags.resize(n);
	for(uint32_t i = 0; i < n; i++){
		const bsa_pog_node_t &x = s->nodes[i];
		if(x.header >= n || x.next >= n || x.prev >= n) return BSA_E_ARG;
		g->header[i] = x.header; g->next[i] = x.next; g->prev[i] = x.prev; g->pos[i] = x.pos; g->cpos[i] = x.cpos; g->rid[i] = x.rid; g->cov[i] = x.cov;
		g->base[i] = x.base; g->flags[i] = x.flags;
	}
	g->ndoff.assign(s->ndoff, s->ndoff + s->nreads); g->rdlen.assign(s->rdlen, s->rdlen + s->nreads);
	for(uint32_t r = 0; r < s->nreads; r++) if(g->ndoff[r] == 0 || (uint64_t)g->ndoff[r] + g->rdlen[r] >= n) return BSA_E_ARG;
	g->HEAD = s->head; g->TAIL = s->tail;
	// out-lists first (they make the edges), then every in-list links the same edges in its own order
	const uint32_t ne = s->out_off[n];
	if(s->in_off[n] != ne || (ne && (!s->out_to || !s->out_cov || !s->in_from))) return BSA_E_ARG;
	// a snapshot comes from outside (fixtures, a binding's copy of another graph): nothing below may trust it.  Offsets start at zero and never
	// decrease (so no list reaches past `ne`); every ring's header is a header and the ring closes; every edge is linked into ONE in-list.
	if(s->out_off[0] != 0 || s->in_off[0] != 0) return BSA_E_ARG;
	for(uint32_t u = 0; u < n; u++) if(s->out_off[u + 1] < s->out_off[u] || s->in_off[u + 1] < s->in_off[u] || s->out_off[u + 1] > ne || s->in_off[u + 1] > ne) return BSA_E_ARG;
	{
		std::vector<uint8_t> seen(n, 0);
		for(uint32_t i = 0; i < n; i++){
			const uint32_t h = g->header[i];
			if(g->header[h] != h) return BSA_E_ARG;
			if(i != h || seen[h]) continue;
			uint32_t x = h, steps = 0;
			do {                                                        // the ring of h: every member names h, next / prev agree, back at h within n steps
				if(g->header[x] != h || g->prev[g->next[x]] != x || seen[x] || ++steps > n) return BSA_E_ARG;
				seen[x] = 1; x = g->next[x];
			} while(x != h);
		}
		for(uint32_t i = 0; i < n; i++) if(!seen[i]) return BSA_E_ARG;   // (a node whose ring never reaches its header)
	}
	std::vector<uint8_t> linked((size_t)ne + 1, 0);
	g->edges.resize((size_t)ne + 1);
	for(uint32_t u = 0; u < n; u++){
		uint32_t *p = &g->out[u];
		for(uint32_t k = s->out_off[u]; k < s->out_off[u + 1]; k++){
			if(s->out_to[k] >= n) return BSA_E_ARG;
			Edge &E = g->edges[k + 1];
			E.from = u; E.to = s->out_to[k]; E.cov = s->out_cov[k]; E.next_out = E.next_in = 0;
			*p = k + 1; p = &E.next_out;
		}
	}
	for(uint32_t v = 0; v < n; v++){
		uint32_t *p = &g->in[v];
		for(uint32_t k = s->in_off[v]; k < s->in_off[v + 1]; k++){
			const uint32_t u = s->in_from[k];
			if(u >= n) return BSA_E_ARG;
			uint32_t e = 0;
			for(uint32_t j = s->out_off[u]; j < s->out_off[u + 1]; j++) if(s->out_to[j] == v){ e = j + 1; break; }
			if(!e || linked[e]) return BSA_E_ARG;                 // no such edge, or already in an in-list (a duplicate would close the list on itself)
			linked[e] = 1;
			*p = e; p = &g->edges[e].next_in;
		}
	}
	return BSA_OK;
	});
}

extern "C" int bsa_pog_export(const bsa_pog_t *g, uint32_t *nnodes, uint32_t *nreads, uint32_t *nedges, uint32_t *head, uint32_t *tail,
		bsa_pog_node_t *nodes, uint32_t *ndoff, uint32_t *rdlen, uint32_t *out_off, uint32_t *out_to, uint32_t *out_cov, uint32_t *in_off, uint32_t *in_from){
	if(!g) return BSA_E_ARG;
	const uint32_t n = g->nnodes();
	uint32_t ne = 0;
	for(uint32_t u = 0; u < n; u++) for(uint32_t e = g->out[u]; e; e = g->edges[e].next_out) ne++;
	if(nnodes) *nnodes = n;
	if(nreads) *nreads = (uint32_t)g->ndoff.size();
	if(nedges) *nedges = ne;
	if(head) *head = g->HEAD;
	if(tail) *tail = g->TAIL;
	if(nodes) for(uint32_t i = 0; i < n; i++){
		bsa_pog_node_t &x = nodes[i];
		x.header = g->header[i]; x.next = g->next[i]; x.prev = g->prev[i]; x.pos = g->pos[i]; x.cpos = g->cpos[i]; x.rid = g->rid[i]; x.cov = g->cov[i];
		x.base = g->base[i]; x.flags = g->flags[i]; x.reserved = 0;
	}
	if(ndoff) memcpy(ndoff, g->ndoff.data(), g->ndoff.size() * 4);
	if(rdlen) memcpy(rdlen, g->rdlen.data(), g->rdlen.size() * 4);
	if(out_off && out_to && out_cov){
		uint32_t k = 0;
		for(uint32_t u = 0; u < n; u++){ out_off[u] = k; for(uint32_t e = g->out[u]; e; e = g->edges[e].next_out){ out_to[k] = g->edges[e].to; out_cov[k] = g->edges[e].cov; k++; } }
		out_off[n] = k;
	}
	if(in_off && in_from){
		uint32_t k = 0;
		for(uint32_t v = 0; v < n; v++){ in_off[v] = k; for(uint32_t e = g->in[v]; e; e = g->edges[e].next_in) in_from[k++] = g->edges[e].from; }
		in_off[n] = k;
	}
	return BSA_OK;
}

extern "C" int bsa_pog_set_cpos(bsa_pog_t *g, const uint32_t *idx, const int32_t *cpos, size_t n){
	if(!g || (n && (!idx || !cpos))) return BSA_E_ARG;
	for(size_t k = 0; k < n; k++){ if(idx[k] >= g->nnodes()) return BSA_E_ARG; g->cpos[idx[k]] = cpos[k]; }
	return BSA_OK;
}
extern "C" int bsa_pog_get_cpos(const bsa_pog_t *g, const uint32_t *idx, int32_t *cpos, size_t n){
	if(!g || (n && (!idx || !cpos))) return BSA_E_ARG;
	for(size_t k = 0; k < n; k++){ if(idx[k] >= g->nnodes()) return BSA_E_ARG; cpos[k] = g->cpos[idx[k]]; }
	return BSA_OK;
}
extern "C" void bsa_pog_seconds(const bsa_pog_t *g, double out[5]){ if(g && out) for(int k = 0; k < 5; k++) out[k] = g->secs[k]; }

// ---- the realn entry of align_rd_bspoa (bspoa.h:2626-2630): the stretch leaves the graph base by base
extern "C" int bsa_pog_cut(bsa_pog_t *g, uint32_t rid, uint32_t rbeg, uint32_t rlen){
	return guarded([&]() -> int {
	if(!g || rid >= g->ndoff.size() || (uint64_t)rbeg + rlen > g->rdlen[rid] || g->stage != 0) return BSA_E_ARG;
	if(rid) for(uint32_t p = rbeg; p < rbeg + rlen; p++) g->cut_rdnode(rid, (int)p);          // (read 0, the backbone, stays: `if(realn && rid)`)
	g->realn = true;
	return BSA_OK;
	});
}

// ---- P2: sel_nodes_bspoa (bspoa.h:1887-2020)
extern "C" int bsa_pog_select(bsa_pog_t *g, uint32_t rid, uint32_t rbeg, uint32_t rlen, bsa_pog_read_t *rd, const uint32_t **sel){
	return guarded([&]() -> int {
	if(!g || rid >= g->ndoff.size() || (uint64_t)rbeg + rlen > g->rdlen[rid]) return BSA_E_ARG;
	if(g->stage != 0) return BSA_E_ARG;
	// a stretch that is still chained into the graph has to be cut out first (bsa_pog_cut: the realn entry of align_rd_bspoa, bspoa.h:2626-2630)
	for(uint32_t p = rbeg; p <= rbeg + rlen; p++) if(g->flags[g->rdnode(rid, (int)p)] & BSA_POG_F_RDC) return BSA_E_ARG;
	const double t0 = now_s();
	g->grow_scratch();
	const uint32_t nreads = (uint32_t)g->ndoff.size();
	const uint32_t nmsa = g->par.seqcore ? std::min<uint32_t>(nreads, (uint32_t)g->par.seqcore) : nreads;
	const bool recent = g->par.nrec && !g->realn;                            // (a re-aligned stretch is selected against every read)
	g->realn = false;
	const uint32_t r0 = recent ? (uint32_t)std::max(0, (int)rid - g->par.nrec - 1) : 0u;
	const uint32_t r1 = recent ? rid : 0xFFFFu;                              // reads [r0, r1) (bspoa.h:2636-2642)
	const uint32_t nhead = g->header[g->rdnode(rid, (int)rbeg - 1)], ntail = g->header[g->rdnode(rid, (int)(rbeg + rlen))];
	for(uint32_t s : g->sels) g->sel_bit[s] = 0;
	g->sels.clear(); g->todels.clear();
	memset(&g->rd, 0, sizeof(g->rd));
	g->rd.nhead = nhead; g->rd.ntail = ntail;
	g->cur_rid = rid; g->cur_rbeg = rbeg;
	g->rd.qlen = g->rd.slen = rlen; g->rd.qb = 0; g->rd.qe = rlen;
	g->qseq.resize(rlen);
	for(uint32_t i = 0; i < rlen; i++) g->qseq[i] = g->base[g->rdnode(rid, (int)(rbeg + i))];
	if(nhead != ntail){
		// where the two end rings sit on every read of the range
		g->rdreg[0].assign(nreads, INT_MAX); g->rdreg[1].assign(nreads, -1);
		for(int side = 0; side < 2; side++){
			const uint32_t u = side ? ntail : nhead;
			uint32_t x = u;
			do { if(g->rid[x] >= r0 && g->rid[x] < r1) g->rdreg[side][g->rid[x]] = g->pos[x]; x = g->next[x]; } while(x != u);
		}
		// every ring those reads pass between them, once, in read / position order
		for(uint32_t r = 0; r < nmsa; r++){
			const int rb = g->rdreg[0][r], re = g->rdreg[1][r];
			if(rb >= re) continue;
			for(int j = rb; j <= re; j++){
				const uint32_t h = g->header[g->rdnode(r, j)];
				if(g->sel_bit[h]) continue;
				g->sel_bit[h] = 1; g->sels.push_back(h);
				g->nct[h] = 0; g->vst[h] = 0;
			}
		}
		// a selected node without a selected predecessor hangs off the head, one without a selected successor leads to the tail
		// (the edges are real for the length of this alignment: they take part in every list order)
		for(size_t k = 0; k < g->sels.size(); k++){
			const uint32_t u = g->sels[k];
			if(u == nhead) continue;
			int j = 0;
			for(uint32_t e = g->out[u]; e; e = g->edges[e].next_out) if(g->sel_bit[g->edges[e].to]){ j |= 1; break; }
			for(uint32_t e = g->in[u]; e; e = g->edges[e].next_in) if(g->sel_bit[g->edges[e].from]){ j |= 2; break; }
			if(j == 3) continue;
			if(j == 1 || u == ntail){ g->chg_edge(nhead, u, 1); g->todels.push_back(((uint64_t)nhead << 32) | u); }
			else if(j == 2){ g->chg_edge(u, ntail, 1); g->todels.push_back(((uint64_t)u << 32) | ntail); }
		}
		// the ring's bonus flag and the in-degrees inside the selection
		for(size_t k = 0; k < g->sels.size(); k++){
			const uint32_t u = g->sels[k];
			uint8_t b = 0;
			uint32_t x = u;
			do { b |= (g->flags[x] & BSA_POG_F_BLESS) ? 1 : 0; x = g->next[x]; } while(x != u && !b);
			g->bonus[u] = b;
			for(uint32_t e = g->out[u]; e; e = g->edges[e].next_out) if(g->sel_bit[g->edges[e].to]) g->nct[g->edges[e].to]++;
		}
	}
	g->rd.nsel = (uint32_t)g->sels.size();
	g->stage = 1;
	if(rd) *rd = g->rd;
	if(sel) *sel = g->sels.data();
	g->secs[0] += now_s() - t0;
	return BSA_OK;
	});
}

static inline uint32_t roundup16(uint32_t v){ return (v + 15u) / 16u * 16u; }

extern "C" int bsa_pog_needs_guide(const bsa_pog_t *g, uint32_t reflen){
	if(!g || g->stage < 1) return 0;
	return g->par.bwtrigger && g->rd.nhead == g->HEAD && g->rd.ntail == g->TAIL && reflen && (int64_t)roundup16(g->rd.qlen) > (int64_t)g->par.bandwidth;
}

// ---- P3: prepare_rd_align_bspoa (bspoa.h:2022-2230)
extern "C" int bsa_pog_place(bsa_pog_t *g, const bsa_pog_guide_t *gd, const int32_t *cpos_sel, bsa_pog_read_t *rd){
	return guarded([&]() -> int {
	if(!g || g->stage != 1) return BSA_E_ARG;
	const double t0 = now_s();
	const uint32_t seqlen = g->rd.qlen;
	const uint32_t reflen = gd ? gd->reflen : 0u;
	uint32_t bw = g->par.bandwidth == 0 ? roundup16(seqlen) : roundup16(std::min<uint32_t>((uint32_t)g->par.bandwidth, seqlen));
	int tb = 0, te = (int)reflen;
	const uint32_t *cgs = nullptr; uint32_t ncg = 0;
	uint32_t x = 0, y = 0;
	const bool ends = g->par.bwtrigger && g->rd.nhead == g->HEAD && g->rd.ntail == g->TAIL;
	if(ends && gd && gd->sam && gd->ncigar){
		// refmode: the read's SAM CIGAR against the backbone places the band (bspoa.h:2055-2085).  Leading D / N / H and I / S runs are margins; so are the
		// trailing ones -- where the reference adds the length of the word BEHIND the one it tests (`cgs[i]` for `cgs[i - 1]`, bspoa.h:2073): kept, so
		// cigar[ncigar] must be readable (the reference reads the next read's first word, or whatever follows, there)
		if(!gd->cigar) return BSA_E_ARG;
		const uint32_t *c0 = gd->cigar; uint32_t nc = gd->ncigar, i;
		x = y = 0;
		for(i = 0; i < nc; i++){
			const uint32_t op = c0[i] & 0xfu;
			if(op == 2u || op == 3u || op == 5u) y += c0[i] >> 4;
			else if(op == 1u || op == 4u) x += c0[i] >> 4;
			else break;
		}
		c0 += i; nc -= i;
		g->rd.qb = x; tb = (int)y;
		x = y = 0;
		for(i = nc; i; i--){
			const uint32_t op = c0[i - 1] & 0xfu;
			if(op == 2u || op == 3u || op == 5u) y += c0[i - 1] >> 4;
			else if(op == 1u || op == 4u) x += c0[i] >> 4;
			else break;
		}
		nc = i;
		g->rd.qe = g->rd.qlen - x; g->rd.slen = g->rd.qe - g->rd.qb;
		te = (int)(reflen - y);                                                   // (g->backbone - y: the caller passes the backbone's length as reflen in refmode)
		x = 0; y = (uint32_t)tb;
		tb = (tb >= (int)(bw / 2)) ? tb - (int)(bw / 4) : 0;
		te = (reflen - (uint32_t)te >= bw / 2) ? te + (int)(bw / 4) : (int)reflen;
		cgs = c0; ncg = nc;
	} else
	if(bsa_pog_needs_guide(g, reflen)){
		if(!gd || !gd->have) return BSA_E_ARG;
		g->rd.qb = (uint32_t)gd->qb; g->rd.qe = (uint32_t)gd->qe; g->rd.slen = g->rd.qe - g->rd.qb;
		tb = (gd->tb >= (int)(bw / 2)) ? gd->tb - (int)(bw / 4) : 0;
		te = ((uint64_t)reflen - (uint64_t)(int64_t)gd->te >= (uint64_t)(bw / 2)) ? (int)(gd->te + (int)(bw / 4)) : (int)reflen;
		cgs = gd->cigar; ncg = gd->ncigar;
		x = 0; y = (uint32_t)gd->tb;
	} else bw = roundup16(seqlen);
	g->rd.bandwidth = bw;
	const uint32_t slen = g->rd.slen, nhead = g->rd.nhead, ntail = g->rd.ntail;
	if(cgs && ncg){
		// column of the consensus -> position on the read, from the guide's CIGAR; columns in front of / behind the alignment spread evenly.
		// (all in uint32 arithmetic, as the reference's u4i expressions, bspoa.h:2112-2152)
		g->rmap.assign((size_t)reflen + 1, 0u);
		uint32_t *rmap = g->rmap.data();
		rmap[0] = 0;
		for(uint32_t i = 1; i < y && i <= reflen; i++) rmap[i] = i * g->rd.qb / (y + 1u);
		for(uint32_t i = 0; i < ncg; i++){
			const uint32_t op = cgs[i] & 0xfu, sz = cgs[i] >> 4;
			switch(op){
				case 0: case 7: case 8: for(uint32_t j = 0; j < sz; j++){ if(y > reflen) return BSA_E_ARG; rmap[y++] = x++; } break;
				case 1: case 4: x += sz; break;
				case 2: case 3: case 5: for(uint32_t j = 0; j < sz; j++){ if(y > reflen) return BSA_E_ARG; rmap[y++] = x; } break;
				default: break;
			}
		}
		for(uint32_t i = y; i < reflen; i++) rmap[i] = x + (i - y + 1u) * (slen - x) / (reflen - y + 1u);
		rmap[reflen] = slen;
		for(size_t k = 0; k < g->sels.size(); k++){
			const uint32_t u = g->sels[k];
			const int cp = cpos_sel ? cpos_sel[k] : g->cpos[u];
			if(cp < 0 || (uint32_t)cp > reflen) return BSA_E_ARG;
			g->cpos[u] = cp;
			int rp = (int)(rmap[cp] - bw / 2u);
			if(rp < 0) rp = 0;
			else if(bw >= slen) rp = 0;
			else if((uint32_t)rp + bw > slen) rp = (int)(slen - bw);
			g->rpos[u] = rp;
			// the guide's two ends: the node in their column hangs off the head / leads to the tail (once each)
			if(cp == tb && tb){
				const bool ex = g->chg_edge(nhead, u, 1);
				g->todels.push_back(((uint64_t)nhead << 32) | u);
				tb = 0;
				if(!ex && g->sel_bit[nhead] && g->sel_bit[u]) g->nct[u]++;
			}
			if(cp == te && te != (int)reflen){
				const bool ex = g->chg_edge(u, ntail, 1);
				g->todels.push_back(((uint64_t)g->header[u] << 32) | ntail);
				te = (int)reflen;
				if(!ex && g->sel_bit[ntail] && g->sel_bit[u]) g->nct[ntail]++;
			}
		}
	} else {
		for(size_t k = 0; k < g->sels.size(); k++){ g->rpos[g->sels[k]] = 0; if(cpos_sel) g->cpos[g->sels[k]] = cpos_sel[k]; }
	}
	g->stage = 2;
	if(rd) *rd = g->rd;
	g->secs[1] += now_s() - t0;
	return BSA_OK;
	});
}

// ---- the program: the traversal of align_rd_bspoacore (bspoa.h:2515-2618) recorded instead of computed.  A stack of complete nodes, a node is
// complete when all its selected in-edges were taken; an edge's left-boundary row number (mpos) and the order of the end-score candidates follow the
// order of the visits.  Every node's selected in-edges, in the order of its in-list with their coverage, are the traceback's view; the forward view
// folds them two at a time in visiting order (a node with more than two is preceded by partial nodes).
extern "C" int bsa_pog_program(bsa_pog_t *g, const bsa_poa_node_t **nodes, size_t *nnodes, const bsa_poa_edge_t **pedges, size_t *nedges,
		const bsa_poa_cand_t **cands, size_t *ncands, const uint8_t **query, bsa_sweep_params_t *par){
	return guarded([&]() -> int {
	if(!g || g->stage < 2) return BSA_E_ARG;
	const double t0 = now_s();
	const uint32_t nhead = g->rd.nhead, ntail = g->rd.ntail, bw = g->rd.bandwidth, slen = g->rd.slen;
	const bool global = g->par.alnmode == BSA_MODE_GLOBAL;
	g->pnodes.clear(); g->pedges.clear(); g->pcands.clear(); g->visits.clear();
	if(nhead != ntail && g->sels.size() >= 2){
		size_t tot = 0;
		for(uint32_t u : g->sels){ g->mpos[u] = INT_MAX - 1; g->voff[u] = (uint32_t)tot; g->loc[u] = 0xFFFFFFFFu; g->vst[u] = 0; tot += g->nct[u]; }
		g->visits.resize(tot + 1);
		g->pnodes.reserve(g->sels.size() + tot + 4); g->pedges.reserve(tot + 1);
		bsa_poa_node_t nd;
		memset(&nd, 0, sizeof(nd));
		nd.rpos = (uint32_t)g->rpos[nhead]; nd.gnode = nhead; nd.base = g->base[nhead]; nd.flags = g->bonus[nhead];
		g->mpos[nhead] = -1;
		g->loc[nhead] = 0;
		g->pnodes.push_back(nd);
		g->stack.clear();
		g->stack.push_back(nhead);
		while(!g->stack.empty()){
			const uint32_t cur = g->stack.back(); g->stack.pop_back();
			for(uint32_t e = g->out[cur]; e; e = g->edges[e].next_out){
				const uint32_t v = g->edges[e].to;
				if(!g->sel_bit[v]) continue;
				if(g->mpos[cur] + 1 < g->mpos[v]) g->mpos[v] = g->mpos[cur] + 1;
				if(v == ntail){
					bsa_poa_cand_t c; c.node = g->loc[cur]; c.kind = 0;
					g->pcands.push_back(c);
					g->vst[v]++;
					continue;
				}
				bsa_pog::Visit &vs = g->visits[g->voff[v] + g->vst[v]];
				vs.src = g->loc[cur]; vs.toff = (uint32_t)g->mpos[v];
				g->vst[v]++;
				if(g->vst[v] != g->nct[v]) continue;
				// v is complete
				const uint32_t cnt = g->nct[v], first_edge = (uint32_t)g->pedges.size();
				uint32_t found = 0;
				for(uint32_t r = g->in[v]; r; r = g->edges[r].next_in){
					const uint32_t w = g->edges[r].from;
					if(!g->sel_bit[w]) continue;
					bsa_poa_edge_t pe; pe.src = g->loc[w]; pe.cov = g->edges[r].cov; pe.src_rpos = (uint32_t)g->rpos[w]; pe.reserved = 0;
					g->pedges.push_back(pe); found++;
				}
				if(found != cnt) return BSA_E_ARG;          // the selection's in-degrees and the lists disagree: a corrupt graph
				for(uint32_t j = 0; j < cnt; j++){
					const bsa_pog::Visit &q = g->visits[g->voff[v] + j];
					const uint32_t sn = g->pnodes[q.src].gnode;
					const uint32_t tk = BSA_POA_IN_PRESENT | ((g->base[v] == g->base[sn]) ? BSA_POA_IN_SAME : 0u) | (q.toff & BSA_POA_IN_TOFF);
					const uint32_t movx = (uint32_t)(g->rpos[v] - g->rpos[sn]);
					if(j == 0 || j >= 2){
						if(j >= 2){ nd.gnode = 0xFFFFFFFFu; g->pnodes.push_back(nd); }          // the row so far: a partial node
						const uint32_t prevn = (uint32_t)g->pnodes.size() - 1u;
						memset(&nd, 0, sizeof(nd));
						nd.rpos = (uint32_t)g->rpos[v]; nd.base = g->base[v]; nd.flags = g->bonus[v]; nd.first_in = first_edge; nd.n_in = 0;
						if(j >= 2){
							nd.in[0].src = prevn; nd.in[0].movx = 0; nd.in[0].toff_kind = BSA_POA_IN_PRESENT | BSA_POA_IN_MERGE;
							nd.in[1].src = q.src; nd.in[1].movx = movx; nd.in[1].toff_kind = tk;
						} else { nd.in[0].src = q.src; nd.in[0].movx = movx; nd.in[0].toff_kind = tk; }
					} else { nd.in[1].src = q.src; nd.in[1].movx = movx; nd.in[1].toff_kind = tk; }
				}
				nd.gnode = v; nd.n_in = (uint16_t)cnt;
				g->loc[v] = (uint32_t)g->pnodes.size();
				g->pnodes.push_back(nd);
				if(!global && (uint32_t)g->rpos[v] + bw >= slen){ bsa_poa_cand_t c; c.node = g->loc[v]; c.kind = 1; g->pcands.push_back(c); }
				g->stack.push_back(v);
			}
		}
	}
	g->stage = 3;
	if(nodes) *nodes = g->pnodes.data();
	if(nnodes) *nnodes = g->pnodes.size();
	if(pedges) *pedges = g->pedges.data();
	if(nedges) *nedges = g->pedges.size();
	if(cands) *cands = g->pcands.data();
	if(ncands) *ncands = g->pcands.size();
	if(query) *query = g->qseq.data() + g->rd.qb;
	if(par){
		memset(par, 0, sizeof(*par));
		par->rows.mode = g->par.alnmode; par->rows.bandwidth = bw;
		par->rows.M = (int8_t)g->par.M; par->rows.X = (int8_t)g->par.X; par->rows.refbonus = (int8_t)g->par.refbonus;
		par->rows.gapo1 = (int8_t)g->par.O; par->rows.gape1 = (int8_t)g->par.E; par->rows.gapo2 = (int8_t)g->par.Q; par->rows.gape2 = (int8_t)g->par.P;
		par->T = g->par.T;
	}
	g->secs[2] += now_s() - t0;
	return BSA_OK;
	});
}

extern "C" int bsa_pog_run(bsa_pog_t *g, bsa_pog_backend_fn fn, void *user, bsa_poa_result_t *res, const bsa_poa_event_t **events){
	return guarded([&]() -> int {
	if(!g || g->stage < 2) return BSA_E_ARG;
	bsa_sweep_params_t sp;
	const uint8_t *q = nullptr;
	int rc = bsa_pog_program(g, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &q, &sp);
	if(rc != BSA_OK) return rc;
	if(g->pnodes.empty()) return BSA_E_UNSUPPORTED;             // nothing to align against (head == tail): the caller's business
	const double t0 = now_s();
	const size_t ecap = 2 * ((size_t)g->rd.slen + g->sels.size()) + 64;
	if(g->events.size() < ecap) g->events.resize(ecap);
	memset(&g->res, 0, sizeof(g->res));
	if(fn) rc = fn(user, g->pnodes.data(), g->pnodes.size(), g->pedges.data(), g->pedges.size(), g->pcands.data(), g->pcands.size(), q, g->rd.slen, &sp, &g->res, g->events.data(), ecap);
	else {
		if(!user) return BSA_E_ARG;
		bsa_poa_prog_t pg;
		memset(&pg, 0, sizeof(pg));
		pg.nnodes = (uint32_t)g->pnodes.size(); pg.nedges = (uint32_t)g->pedges.size(); pg.ncands = (uint32_t)g->pcands.size(); pg.slen = g->rd.slen; pg.event_cap = (uint32_t)ecap;
		rc = bsa_poa_graph_host((bsa_ctx_t*)user, g->pnodes.data(), g->pnodes.size(), g->pedges.data(), g->pedges.size(), g->pcands.data(), g->pcands.size(), &pg, 1,
			q, g->rd.slen, &sp, &g->res, g->events.data(), ecap, nullptr, nullptr);
	}
	g->secs[3] += now_s() - t0;
	if(rc != BSA_OK) return rc;
	if(g->res.status != BSA_POA_ST_OK) return BSA_E_UNSUPPORTED;          // the walk left the stored rows (the reference reads outside its arena there): no result to apply
	g->stage = 4;
	if(res) *res = g->res;
	if(events) *events = g->events.data();
	return BSA_OK;
	});
}

extern "C" int bsa_pog_aux_edges(const bsa_pog_t *g, const uint64_t **list, size_t *n){
	if(!g || !list || !n) return BSA_E_ARG;
	*list = g->todels.data(); *n = g->todels.size();
	return BSA_OK;
}

extern "C" int bsa_pog_abort(bsa_pog_t *g){
	return guarded([&]() -> int {
	if(!g) return BSA_E_ARG;
	g->realn = false;                                           // (a cut that no selection followed)
	if(g->stage == 0 && g->todels.empty()) return BSA_OK;        // (a selection that failed half-way has left its auxiliary edges)
	g->drop_aux();
	g->stage = 0;
	return BSA_OK;
	});
}

// ---- the surgery (alignment2graph_bspoa bspoa.h:2286, 2393-2405, 2501-2511; align_rd_bspoa :2652-2657)
extern "C" int bsa_pog_apply(bsa_pog_t *g, bsa_result_t *rs_out, uint32_t *events_gnode){
	return guarded([&]() -> int {
	if(!g || g->stage != 4) return BSA_E_ARG;
	const double t0 = now_s();
	const uint32_t rid = g->cur_rid, rbeg = g->cur_rbeg, qlen = g->rd.qlen, qb = g->rd.qb;
	const uint32_t nhead = g->rd.nhead, ntail = g->rd.ntail;
	bsa_result_t rs;
	memset(&rs, 0, sizeof(rs));
	for(uint32_t k = 0; k < qlen; k++) g->cpos[g->rdnode(rid, (int)k)] = 0;
	int col = g->cpos[g->pnodes[g->res.maxidx].gnode];
	rs.qe = g->res.maxoff + 1;
	rs.te = col + 1;
	for(int k = 0; k < g->res.nevents; k++){
		const bsa_poa_event_t &ev = g->events[k];
		const uint32_t gn = g->pnodes[ev.node].gnode;
		if(events_gnode) events_gnode[k] = gn;
		if(ev.bt == 1u){ rs.ins++; continue; }
		if(ev.bt != 0u){ rs.del++; continue; }
		const uint32_t rdn = g->rdnode(rid, (int)(rbeg + qb) + ev.x);
		g->cpos[rdn] = g->cpos[gn];
		if(gn != nhead && gn != ntail && g->base[rdn] == g->base[gn]){ g->merge_nodes(gn, rdn); rs.mat++; }
		else rs.mis++;
	}
	rs.qb = g->res.fin_x;
	rs.tb = g->cpos[g->pnodes[g->res.fin_node].gnode];
	// chain the read; a base that met no column takes the column of its right neighbour
	g->connect_rdnode(rid, (int)(rbeg + qlen));
	for(int k = (int)qlen - 1; k >= 0; k--){
		g->connect_rdnode(rid, (int)rbeg + k);
		const uint32_t v = g->rdnode(rid, (int)rbeg + k);
		if(g->cpos[v]) col = g->cpos[v]; else g->cpos[v] = col;
	}
	// align_rd_bspoa: the walk's ends are on the whole read, the score is the best end cell's, the auxiliary edges go
	rs.qb += (int32_t)qb; rs.qe += (int32_t)qb;           // (alignment2graph_bspoa adds g->qb ...
	rs.qb += (int32_t)qb; rs.qe += (int32_t)qb;           //  ... and so does align_rd_bspoa, bspoa.h:2489-2490 and :2652-2653)
	rs.score = g->res.maxscr;
	rs.aln = 0;
	g->drop_aux();
	g->stage = 0;
	if(rs_out) *rs_out = rs;
	g->secs[4] += now_s() - t0;
	return BSA_OK;
	});
}
