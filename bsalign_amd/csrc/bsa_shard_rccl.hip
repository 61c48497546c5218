// bsa_shard_rccl.hip -- the one exchange each way of a sharded batch for a C host: RCCL point-to-point
// messages over xGMI behind bsa_shard_* (include/bsalign_hip.h).  Pairs are independent (bsalign.h:3854-4050 touches only its
// arguments), so nothing here sits inside the DP: rank `root` holds the batch, cuts it into contiguous ranges balanced by band
// cells, and
//   scatter   broadcasts the lengths (two ncclBroadcast), packs every rank's shard (bsa_shard_pack) and posts the N - 1 shards as ONE
//             group of ncclSend (ncclGroupStart / ncclGroupEnd: on MI355X one message per xGMI link, all seven links busy at once);
//   gather    an ncclAllGather of two sizes per rank, then result records, CIGAR counts and CIGAR words as grouped receives on the root.
// RCCL is loaded at run time (dlopen of librccl.so.1): a process that already holds a copy -- PyTorch ships one -- keeps using that one,
// and libbsalign_hip.so has no link-time dependency on it.
// The exchange is written against two small interfaces (bsa_shard_transport.h): the wire (RCCL here; BSA_SHARD_TRANSPORT=shm: shared memory
// between processes of one host, bsa_shard_shm.cpp, so that the rank arithmetic runs at world size 2 without a GPU) and the memory the
// message buffers live in (device / host).  Errors are AGREED before any point-to-point message is posted: a rank that cannot go on
// (shard larger than the caller's arrays, arena too small, allocation failure) says so in a status exchange, and every rank returns
// instead of leaving its peers waiting for a message that never comes.
#include "bsa_common.h"
#include "bsa_shard_transport.h"
#include <dlfcn.h>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

extern "C" int bsa_ctx_get_stream_internal(bsa_ctx_t *ctx, hipStream_t *st);

namespace {
typedef struct { char internal[128]; } RcclId;
typedef void *RcclComm;
struct Rccl {
	void *h = nullptr;
	int (*GetUniqueId)(RcclId*) = nullptr;
	int (*CommInitRank)(RcclComm*, int, RcclId, int) = nullptr;
	int (*CommDestroy)(RcclComm) = nullptr;
	int (*GroupStart)() = nullptr; int (*GroupEnd)() = nullptr;
	int (*Send)(const void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
	int (*Recv)(void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
	int (*Broadcast)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
	int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
	bool ok = false;
};
const int RCCL_UINT8 = 1;        // ncclUint8 (rccl.h: ncclDataType_t): every message travels as bytes
Rccl &rccl(){
	static Rccl r;
	if(r.h) return r;
	for(const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}){ r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if(r.h) break; }
	if(!r.h) return r;
#define SYM(field, sym) *(void**)&r.field = dlsym(r.h, sym)
	SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
	SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
	SYM(Broadcast, "ncclBroadcast"); SYM(AllGather, "ncclAllGather");
#undef SYM
	r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.Broadcast && r.AllGather;
	return r;
}
// ---- RCCL over xGMI, device buffers
struct DeviceSpace : BsaShardSpace {
	hipStream_t st;
	explicit DeviceSpace(hipStream_t s) : st(s) {}
	void *alloc(size_t bytes) override { void *p = nullptr; if(hipMalloc(&p, bytes ? bytes : 16) != hipSuccess){ (void)hipGetLastError(); return nullptr; } return p; }
	void release(void *p) override { if(p) (void)hipFree(p); }
	int to_space(void *d, const void *s, size_t n) override { return (!n || hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, st) == hipSuccess) ? BSA_OK : BSA_E_HIP; }
	int to_host(void *d, const void *s, size_t n) override { return (!n || hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, st) == hipSuccess) ? BSA_OK : BSA_E_HIP; }
	int within(void *d, const void *s, size_t n) override { return (!n || hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, st) == hipSuccess) ? BSA_OK : BSA_E_HIP; }
	int sync() override { return hipStreamSynchronize(st) == hipSuccess ? BSA_OK : BSA_E_HIP; }
};
struct RcclTransport : BsaShardTransport {
	RcclComm comm = nullptr; hipStream_t st; int nranks;
	RcclTransport(RcclComm c, hipStream_t s, int n) : comm(c), st(s), nranks(n) {}
	~RcclTransport() override { if(comm) (void)rccl().CommDestroy(comm); }
	int broadcast(void *buf, size_t bytes, int root) override { return (!bytes || rccl().Broadcast(buf, buf, bytes, RCCL_UINT8, root, comm, st) == 0) ? BSA_OK : BSA_E_HIP; }
	int allgather(const void *mine, void *all, size_t each) override { return rccl().AllGather(mine, all, each, RCCL_UINT8, comm, st) == 0 ? BSA_OK : BSA_E_HIP; }
	int group_begin() override { return rccl().GroupStart() == 0 ? BSA_OK : BSA_E_HIP; }
	int send(const void *buf, size_t bytes, int peer) override { return rccl().Send(buf, bytes, RCCL_UINT8, peer, comm, st) == 0 ? BSA_OK : BSA_E_HIP; }
	int recv(void *buf, size_t bytes, int peer) override { return rccl().Recv(buf, bytes, RCCL_UINT8, peer, comm, st) == 0 ? BSA_OK : BSA_E_HIP; }
	int group_end() override { return rccl().GroupEnd() == 0 ? BSA_OK : BSA_E_HIP; }
};
bool want_shm(){ const char *e = getenv("BSA_SHARD_TRANSPORT"); return e && strcmp(e, "shm") == 0; }
size_t pad16(size_t n){ return (n + 15) & ~(size_t)15; }
}

// a buffer of the communicator's space, grown on demand
struct ShardBuf {
	void *p = nullptr; size_t cap = 0;
	bool need(BsaShardSpace *sp, size_t n){ if(n <= cap) return true; if(p) sp->release(p); p = sp->alloc(n + 256); cap = p ? n + 256 : 0; return p != nullptr; }
};

struct bsa_shard_comm {
	bsa_ctx_t *ctx = nullptr; int rank = 0, nranks = 1;
	BsaShardTransport *wire = nullptr; BsaShardSpace *space = nullptr;
	ShardBuf b_len, b_blob, b_sizes, b_res, b_cnt, b_cig, b_stage, b_words;
	std::vector<uint64_t> bounds;        // the last scatter's ranges (nranks + 1), known on every rank
};

extern "C" int bsa_shard_unique_id(uint8_t id[128]){
	if(!id) return BSA_E_ARG;
	if(want_shm()) return bsa_shm_unique_id(id);
	Rccl &r = rccl();
	if(!r.ok) return BSA_E_UNSUPPORTED;
	RcclId u;
	if(r.GetUniqueId(&u) != 0) return BSA_E_HIP;
	memcpy(id, u.internal, 128);
	return BSA_OK;
}

extern "C" void bsa_shard_comm_destroy(bsa_shard_comm_t *c);
extern "C" int bsa_shard_comm_create(bsa_ctx_t *ctx, int rank, int nranks, const uint8_t id[128], bsa_shard_comm_t **out){
	const bool shm = want_shm();
	if((!ctx && !shm) || !out || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !id)) return BSA_E_ARG;
	*out = nullptr;
	bsa_shard_comm *c = new (std::nothrow) bsa_shard_comm();
	if(!c) return BSA_E_NOMEM;
	c->ctx = ctx; c->rank = rank; c->nranks = nranks;
	if(shm){
		// host buffers, shared-memory wire: no device is touched (ctx may be null)
		c->space = bsa_host_space_create();
		if(nranks > 1){ c->wire = bsa_shm_transport_create(rank, nranks, id); if(!c->wire){ delete c->space; delete c; return BSA_E_HIP; } }
	} else {
		hipStream_t st; if(bsa_ctx_get_stream_internal(ctx, &st) != BSA_OK){ delete c; return BSA_E_ARG; }      // (also selects the context's device)
		c->space = new DeviceSpace(st);
		if(nranks > 1){
			Rccl &r = rccl();
			if(!r.ok){ delete c->space; delete c; return BSA_E_UNSUPPORTED; }
			RcclId u; memcpy(u.internal, id, 128);
			RcclComm comm = nullptr;
			if(r.CommInitRank(&comm, nranks, u, rank) != 0){ delete c->space; delete c; return BSA_E_HIP; }
			c->wire = new RcclTransport(comm, st, nranks);
		}
	}
	// the status words of agree(): allocated here so that agreeing on an error can never itself fail on one rank alone
	if(nranks > 1 && !c->b_words.need(c->space, 8 * ((size_t)nranks + 1))){ bsa_shard_comm_destroy(c); return BSA_E_NOMEM; }
	*out = c;
	return BSA_OK;
}

extern "C" void bsa_shard_comm_destroy(bsa_shard_comm_t *c){
	if(!c) return;
	for(ShardBuf *b : {&c->b_len, &c->b_blob, &c->b_sizes, &c->b_res, &c->b_cnt, &c->b_cig, &c->b_stage, &c->b_words}) if(b->p) c->space->release(b->p);
	delete c->wire;
	delete c->space;
	delete c;
}

// contiguous ranges balanced by band cells, sum over the pairs of tlen * bw
static void cut(const uint32_t *tlen, size_t n, uint32_t bw, int nranks, std::vector<uint64_t> &b){
	b.assign((size_t)nranks + 1, 0);
	double tot = 0; for(size_t k = 0; k < n; k++) tot += (double)tlen[k] * (bw ? bw : 1u);
	double acc = 0; int r = 1;
	for(size_t k = 0; k < n && r < nranks; k++){
		while(r < nranks && acc >= tot * r / nranks){ b[r++] = k; }
		acc += (double)tlen[k] * (bw ? bw : 1u);
	}
	while(r < nranks) b[r++] = n;
	b[nranks] = n;
}

#define TRY(x) do { const int _rc = (x); if(_rc != BSA_OK) return _rc; } while(0)
// inside a group_begin / group_end section: remember the first error, post nothing more, but ALWAYS reach group_end (an RCCL group left
// open keeps every peer blocked in its receive); GEND closes the group and returns what was collected
#define GTRY(x) do { if(_grc == BSA_OK) _grc = (x); } while(0)
#define GEND() do { const int _erc = c->wire->group_end(); if(_grc != BSA_OK) return _grc; if(_erc != BSA_OK) return _erc; } while(0)

// every rank contributes a status word; all learn all of them.  Returns the first non-zero one in rank order (0: everybody can go on).
static int agree(bsa_shard_comm *c, int mine, int *agreed){
	*agreed = mine;
	if(c->nranks == 1) return BSA_OK;
	const size_t nr = (size_t)c->nranks;
	if(!c->b_words.p) return BSA_E_NOMEM;                  // (allocated by bsa_shard_comm_create, 8 bytes a rank: cannot fail here)
	uint8_t *w = (uint8_t*)c->b_words.p;
	uint64_t v = (uint64_t)(uint32_t)mine;
	std::vector<uint64_t> all(nr);
	// a rank whose own word cannot be staged still takes part in the collective -- the others must not hang in it -- and then reports
	const int staged = c->space->to_space(w + 8 * nr, &v, 8);
	TRY(c->wire->allgather(w + 8 * nr, w, 8));
	if(staged != BSA_OK) return staged;
	TRY(c->space->to_host(all.data(), w, 8 * nr));
	TRY(c->space->sync());
	*agreed = BSA_OK;
	for(size_t k = 0; k < nr; k++) if(all[k]){ *agreed = (int)(int32_t)(uint32_t)all[k]; break; }
	return BSA_OK;
}

extern "C" int bsa_shard_scatter(bsa_shard_comm_t *c, int root, const uint8_t *seqs, const uint64_t *qoff, const uint32_t *qlen, const uint64_t *toff, const uint32_t *tlen,
		size_t n, uint32_t bandwidth, size_t *first, size_t *count, uint8_t **d_seqs, size_t *blob_bytes, uint32_t *lqlen, uint32_t *ltlen, uint64_t *lqoff, uint64_t *ltoff, size_t cap){
	if(!c || root < 0 || root >= c->nranks || !first || !count || !d_seqs || !blob_bytes) return BSA_E_ARG;
	const bool isroot = c->rank == root;
	BsaShardSpace *sp = c->space;
	// 1. the lengths of all pairs to everybody: [n, bandwidth, root's argument check] then qlen | tlen
	uint64_t head[3] = {(uint64_t)n, bandwidth, (uint64_t)((isroot && n && (!seqs || !qoff || !qlen || !toff || !tlen)) ? 1 : 0)};
	if(!c->b_sizes.need(sp, 32)) return BSA_E_NOMEM;
	if(c->nranks > 1){
		if(isroot) TRY(sp->to_space(c->b_sizes.p, head, 24));
		TRY(c->wire->broadcast(c->b_sizes.p, 24, root));
		TRY(sp->to_host(head, c->b_sizes.p, 24)); TRY(sp->sync());
	}
	if(head[2]) return BSA_E_ARG;                                  // (every rank has the root's verdict)
	const size_t N = (size_t)head[0]; const uint32_t bw = (uint32_t)head[1];
	std::vector<uint32_t> lens(2 * N);
	if(isroot && N){ memcpy(lens.data(), qlen, N * 4); memcpy(lens.data() + N, tlen, N * 4); }
	if(c->nranks > 1 && N){
		int st = c->b_len.need(sp, 8 * N) ? BSA_OK : BSA_E_NOMEM, all = BSA_OK;
		TRY(agree(c, st, &all)); if(all != BSA_OK) return all;
		if(isroot) TRY(sp->to_space(c->b_len.p, lens.data(), 8 * N));
		TRY(c->wire->broadcast(c->b_len.p, 8 * N, root));
		TRY(sp->to_host(lens.data(), c->b_len.p, 8 * N)); TRY(sp->sync());
	}
	cut(lens.data() + N, N, bw, c->nranks, c->bounds);
	const size_t a = (size_t)c->bounds[c->rank], b = (size_t)c->bounds[c->rank + 1], mine = b - a;
	*first = a; *count = mine;
	int status = (mine > cap) ? BSA_E_NOMEM : BSA_OK;
	size_t acc = 0;
	for(size_t i = 0; i < mine; i++){
		if(status == BSA_OK){
			if(ltoff) ltoff[i] = acc;
			if(lqlen) lqlen[i] = lens[a + i];
			if(ltlen) ltlen[i] = lens[N + a + i];
		}
		acc += pad16(lens[N + a + i]);
		if(status == BSA_OK && lqoff) lqoff[i] = acc;
		acc += pad16(lens[a + i]);
	}
	*blob_bytes = acc;
	if(status == BSA_OK && !c->b_blob.need(sp, acc + 64)) status = BSA_E_NOMEM;
	*d_seqs = (uint8_t*)c->b_blob.p;
	// 2. the shards: packed on the root -- and only when every rank can take its own, one grouped set of sends
	std::vector<size_t> soff((size_t)c->nranks + 1, 0);
	std::vector<uint8_t> host;
	if(isroot && status == BSA_OK){
		for(int k = 0; k < c->nranks; k++) soff[k + 1] = soff[k] + pad16(bsa_shard_bytes(lens.data(), lens.data() + N, (size_t)c->bounds[k], (size_t)(c->bounds[k + 1] - c->bounds[k])));
		host.resize(soff[c->nranks] + 16);
		std::vector<uint64_t> tq, tt;
		for(int k = 0; k < c->nranks && status == BSA_OK; k++){
			const size_t f = (size_t)c->bounds[k], cn = (size_t)(c->bounds[k + 1] - c->bounds[k]);
			tq.resize(cn + 1); tt.resize(cn + 1);
			status = bsa_shard_pack(seqs, qoff, qlen, toff, tlen, f, cn, host.data() + soff[k], soff[k + 1] - soff[k], tq.data(), tt.data(), 0);
		}
		if(status == BSA_OK && !c->b_stage.need(sp, soff[c->nranks] + 64)) status = BSA_E_NOMEM;
	}
	int all = BSA_OK;
	TRY(agree(c, status, &all));
	if(all != BSA_OK) return all;
	if(isroot){
		TRY(sp->to_space(c->b_stage.p, host.data(), soff[c->nranks]));
		TRY(sp->within(c->b_blob.p, (const uint8_t*)c->b_stage.p + soff[root], soff[root + 1] - soff[root]));
		if(c->nranks > 1){
			TRY(c->wire->group_begin());
			int _grc = BSA_OK;
			for(int k = 0; k < c->nranks; k++) if(k != root && soff[k + 1] > soff[k]) GTRY(c->wire->send((const uint8_t*)c->b_stage.p + soff[k], soff[k + 1] - soff[k], k));
			GEND();
		}
		TRY(sp->sync());
	} else if(acc){          // (acc == the root's soff[rank + 1] - soff[rank]: both sides derive it from the same lengths)
		TRY(c->wire->recv(c->b_blob.p, acc, root));
		TRY(sp->sync());
	}
	return BSA_OK;
}

extern "C" int bsa_shard_gather(bsa_shard_comm_t *c, int root, const bsa_result_t *d_out, const uint32_t *d_cigar, const uint64_t *cigar_off, size_t count,
		bsa_result_t *out, uint32_t *cigar, size_t cigar_cap_words, uint64_t *out_cigar_off, size_t n){
	if(!c || root < 0 || root >= c->nranks) return BSA_E_ARG;
	const bool isroot = c->rank == root;
	BsaShardSpace *sp = c->space;
	const size_t nr = (size_t)c->nranks;
	// 1. sizes and every rank's own verdict on its arguments: [count, CIGAR words, status]
	int status = (count && (!d_out || !cigar_off)) ? BSA_E_ARG : BSA_OK;
	const uint64_t nwords = (status == BSA_OK && count) ? cigar_off[count] : 0;          // cigar_off: HOST array of count + 1 offsets into d_cigar
	if(nwords && !d_cigar) status = BSA_E_ARG;                                          // (what is announced is what is sent: the root posts its receives from these sizes)
	std::vector<uint64_t> sizes(3 * nr, 0);
	sizes[3 * c->rank] = count; sizes[3 * c->rank + 1] = nwords; sizes[3 * c->rank + 2] = (uint64_t)(uint32_t)status;
	if(c->nranks > 1){
		if(!c->b_sizes.need(sp, 24 * nr + 32)) return BSA_E_NOMEM;
		uint8_t *ds = (uint8_t*)c->b_sizes.p;
		TRY(sp->to_space(ds + 24 * nr, &sizes[3 * c->rank], 24));
		TRY(c->wire->allgather(ds + 24 * nr, ds, 24));
		TRY(sp->to_host(sizes.data(), ds, 24 * nr)); TRY(sp->sync());
	}
	for(size_t k = 0; k < nr; k++) if(sizes[3 * k + 2]) return (int)(int32_t)(uint32_t)sizes[3 * k + 2];
	// 2. the root's verdict (totals against what its caller expects, the arena, its buffers) and this rank's count buffer, agreed
	size_t tot = 0, totw = 0;
	std::vector<size_t> p0(nr + 1, 0), w0(nr + 1, 0);
	for(size_t k = 0; k < nr; k++){ p0[k + 1] = p0[k] + (size_t)sizes[3 * k]; w0[k + 1] = w0[k] + (size_t)sizes[3 * k + 1]; }
	tot = p0[nr]; totw = w0[nr];
	status = c->b_cnt.need(sp, 8 * std::max<size_t>(count, 1)) ? BSA_OK : BSA_E_NOMEM;
	if(isroot && status == BSA_OK){
		if(tot != n || !out || !out_cigar_off) status = BSA_E_ARG;
		else if(cigar && totw > cigar_cap_words){ out_cigar_off[n] = totw; status = BSA_E_CIGAR_CAP; }
		else if(!c->b_res.need(sp, tot * sizeof(bsa_result_t) + 64) || !c->b_len.need(sp, 8 * tot + 64) || !c->b_cig.need(sp, 4 * totw + 64)) status = BSA_E_NOMEM;
	}
	int all = BSA_OK;
	TRY(agree(c, status, &all));
	if(all != BSA_OK) return all;
	// 3. the per-pair CIGAR word counts travel as u64 (count of them per rank); results, counts, words: one group
	std::vector<uint64_t> cnt(count);
	for(size_t i = 0; i < count; i++) cnt[i] = cigar_off[i + 1] - cigar_off[i];
	if(count) TRY(sp->to_space(c->b_cnt.p, cnt.data(), 8 * count));
	if(!isroot){
		TRY(c->wire->group_begin());
		int _grc = BSA_OK;
		if(count){ GTRY(c->wire->send(d_out, count * sizeof(bsa_result_t), root)); GTRY(c->wire->send(c->b_cnt.p, 8 * count, root)); }
		if(nwords) GTRY(c->wire->send(d_cigar, nwords * 4, root));
		GEND();
		TRY(sp->sync());
		return BSA_OK;
	}
	uint8_t *dr = (uint8_t*)c->b_res.p, *dc = (uint8_t*)c->b_len.p, *dw = (uint8_t*)c->b_cig.p;
	if(count){
		TRY(sp->within(dr + p0[root] * sizeof(bsa_result_t), d_out, count * sizeof(bsa_result_t)));
		TRY(sp->within(dc + 8 * p0[root], c->b_cnt.p, 8 * count));
		if(nwords) TRY(sp->within(dw + 4 * w0[root], d_cigar, 4 * nwords));
	}
	if(c->nranks > 1){
		TRY(c->wire->group_begin());
		int _grc = BSA_OK;
		for(int k = 0; k < c->nranks; k++){
			if(k == root) continue;
			const size_t ck = (size_t)sizes[3 * k], wk = (size_t)sizes[3 * k + 1];
			if(ck){ GTRY(c->wire->recv(dr + p0[k] * sizeof(bsa_result_t), ck * sizeof(bsa_result_t), k)); GTRY(c->wire->recv(dc + 8 * p0[k], 8 * ck, k)); }
			if(wk) GTRY(c->wire->recv(dw + 4 * w0[k], wk * 4, k));
		}
		GEND();
	}
	std::vector<uint64_t> allcnt(tot);
	TRY(sp->to_host(out, dr, tot * sizeof(bsa_result_t)));
	if(tot) TRY(sp->to_host(allcnt.data(), dc, 8 * tot));
	if(cigar && totw) TRY(sp->to_host(cigar, dw, 4 * totw));
	TRY(sp->sync());
	uint64_t acc = 0;
	for(size_t i = 0; i < tot; i++){ out_cigar_off[i] = acc; acc += allcnt[i]; }
	out_cigar_off[tot] = acc;
	return acc == totw ? BSA_OK : BSA_E_HIP;
}
