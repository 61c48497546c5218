// bsa_shard_rccl.hip -- the one exchange each way of a sharded batch for a C host: RCCL point-to-point
// messages over xGMI behind bsa_shard_* (include/bsalign_hip.h).  Pairs are independent (bsalign.h:3854-4050 touches only its
// arguments), so nothing here sits inside the DP: rank `root` holds the batch, cuts it into contiguous ranges balanced by band
// cells, and
//   scatter   broadcasts the lengths (two ncclBroadcast), packs every rank's shard (bsa_shard_pack) and posts the N - 1 shards as ONE
//             group of ncclSend (ncclGroupStart / ncclGroupEnd: on MI355X one message per xGMI link, all seven links busy at once);
//   gather    an ncclAllGather of two sizes per rank, then result records, CIGAR counts and CIGAR words as grouped receives on the root.
// RCCL is loaded at run time (dlopen of librccl.so.1): a process that already holds a copy -- PyTorch ships one -- keeps using that one,
// and libbsalign_hip.so has no link-time dependency on it.
#include "bsa_common.h"
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
const int RCCL_UINT8 = 1, RCCL_UINT64 = 5;        // ncclUint8, ncclUint64 (rccl.h: ncclDataType_t)
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
struct Dev {
	void *p = nullptr; size_t cap = 0;
	~Dev(){ if(p) (void)hipFree(p); }
	bool need(size_t n){ if(n <= cap) return true; if(p) (void)hipFree(p); p = nullptr; cap = 0; if(hipMalloc(&p, n + 256) != hipSuccess){ (void)hipGetLastError(); return false; } cap = n + 256; return true; }
};
size_t pad16(size_t n){ return (n + 15) & ~(size_t)15; }
}

struct bsa_shard_comm {
	bsa_ctx_t *ctx = nullptr; RcclComm comm = nullptr; int rank = 0, nranks = 1;
	Dev d_len, d_blob, d_sizes, d_res, d_cnt, d_cig, d_stage;
	std::vector<uint64_t> bounds;        // the last scatter's ranges (nranks + 1), known on every rank
};

extern "C" int bsa_shard_unique_id(uint8_t id[128]){
	if(!id) return BSA_E_ARG;
	Rccl &r = rccl();
	if(!r.ok) return BSA_E_UNSUPPORTED;
	RcclId u;
	if(r.GetUniqueId(&u) != 0) return BSA_E_HIP;
	memcpy(id, u.internal, 128);
	return BSA_OK;
}

extern "C" int bsa_shard_comm_create(bsa_ctx_t *ctx, int rank, int nranks, const uint8_t id[128], bsa_shard_comm_t **out){
	if(!ctx || !out || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !id)) return BSA_E_ARG;
	*out = nullptr;
	bsa_shard_comm *c = new (std::nothrow) bsa_shard_comm();
	if(!c) return BSA_E_NOMEM;
	c->ctx = ctx; c->rank = rank; c->nranks = nranks;
	if(nranks > 1){
		Rccl &r = rccl();
		if(!r.ok){ delete c; return BSA_E_UNSUPPORTED; }
		hipStream_t st; if(bsa_ctx_get_stream_internal(ctx, &st) != BSA_OK){ delete c; return BSA_E_ARG; }      // (also selects the context's device)
		RcclId u; memcpy(u.internal, id, 128);
		if(r.CommInitRank(&c->comm, nranks, u, rank) != 0){ delete c; return BSA_E_HIP; }
	}
	*out = c;
	return BSA_OK;
}

extern "C" void bsa_shard_comm_destroy(bsa_shard_comm_t *c){
	if(!c) return;
	if(c->comm) (void)rccl().CommDestroy(c->comm);
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

#define RC(x) do { if((x) != 0) return BSA_E_HIP; } while(0)
#define HC(x) do { if((x) != hipSuccess) return BSA_E_HIP; } while(0)

extern "C" int bsa_shard_scatter(bsa_shard_comm_t *c, int root, const uint8_t *seqs, const uint64_t *qoff, const uint32_t *qlen, const uint64_t *toff, const uint32_t *tlen,
		size_t n, uint32_t bandwidth, size_t *first, size_t *count, uint8_t **d_seqs, size_t *blob_bytes, uint32_t *lqlen, uint32_t *ltlen, uint64_t *lqoff, uint64_t *ltoff, size_t cap){
	if(!c || root < 0 || root >= c->nranks || !first || !count || !d_seqs || !blob_bytes) return BSA_E_ARG;
	const bool isroot = c->rank == root;
	if(isroot && n && (!seqs || !qoff || !qlen || !toff || !tlen)) return BSA_E_ARG;
	hipStream_t st; int rc = bsa_ctx_get_stream_internal(c->ctx, &st); if(rc != BSA_OK) return rc;
	Rccl &r = rccl();
	// 1. the lengths of all pairs to everybody: [n, bandwidth] then qlen | tlen
	uint64_t head[2] = {(uint64_t)n, bandwidth};
	if(!c->d_sizes.need(16)) return BSA_E_NOMEM;
	if(c->nranks > 1){
		if(isroot) HC(hipMemcpyAsync(c->d_sizes.p, head, 16, hipMemcpyHostToDevice, st));
		RC(r.Broadcast(c->d_sizes.p, c->d_sizes.p, 2, RCCL_UINT64, root, c->comm, st));
		HC(hipMemcpyAsync(head, c->d_sizes.p, 16, hipMemcpyDeviceToHost, st)); HC(hipStreamSynchronize(st));
	}
	const size_t N = (size_t)head[0]; const uint32_t bw = (uint32_t)head[1];
	std::vector<uint32_t> lens(2 * N);
	if(isroot){ memcpy(lens.data(), qlen, N * 4); memcpy(lens.data() + N, tlen, N * 4); }
	if(c->nranks > 1 && N){
		if(!c->d_len.need(8 * N)) return BSA_E_NOMEM;
		if(isroot) HC(hipMemcpyAsync(c->d_len.p, lens.data(), 8 * N, hipMemcpyHostToDevice, st));
		RC(r.Broadcast(c->d_len.p, c->d_len.p, 8 * N, RCCL_UINT8, root, c->comm, st));
		HC(hipMemcpyAsync(lens.data(), c->d_len.p, 8 * N, hipMemcpyDeviceToHost, st)); HC(hipStreamSynchronize(st));
	}
	cut(lens.data() + N, N, bw, c->nranks, c->bounds);
	const size_t a = (size_t)c->bounds[c->rank], b = (size_t)c->bounds[c->rank + 1], mine = b - a;
	*first = a; *count = mine;
	if(mine > cap) return BSA_E_NOMEM;
	size_t acc = 0;
	for(size_t i = 0; i < mine; i++){
		if(ltoff) ltoff[i] = acc; acc += pad16(lens[N + a + i]);
		if(lqoff) lqoff[i] = acc; acc += pad16(lens[a + i]);
		if(lqlen) lqlen[i] = lens[a + i];
		if(ltlen) ltlen[i] = lens[N + a + i];
	}
	*blob_bytes = acc;
	if(!c->d_blob.need(acc + 64)) return BSA_E_NOMEM;
	*d_seqs = (uint8_t*)c->d_blob.p;
	// 2. the shards: packed on the root, one grouped set of sends
	if(isroot){
		std::vector<size_t> soff((size_t)c->nranks + 1, 0);
		for(int k = 0; k < c->nranks; k++) soff[k + 1] = soff[k] + pad16(bsa_shard_bytes(lens.data(), lens.data() + N, (size_t)c->bounds[k], (size_t)(c->bounds[k + 1] - c->bounds[k])));
		std::vector<uint8_t> host(soff[c->nranks] + 16);
		std::vector<uint64_t> tq, tt;
		for(int k = 0; k < c->nranks; k++){
			const size_t f = (size_t)c->bounds[k], cn = (size_t)(c->bounds[k + 1] - c->bounds[k]);
			tq.resize(cn + 1); tt.resize(cn + 1);
			rc = bsa_shard_pack(seqs, qoff, qlen, toff, tlen, f, cn, host.data() + soff[k], soff[k + 1] - soff[k], tq.data(), tt.data(), 0);
			if(rc != BSA_OK) return rc;
		}
		if(!c->d_stage.need(soff[c->nranks] + 64)) return BSA_E_NOMEM;
		HC(hipMemcpyAsync(c->d_stage.p, host.data(), soff[c->nranks], hipMemcpyHostToDevice, st));
		HC(hipMemcpyAsync(c->d_blob.p, (const uint8_t*)c->d_stage.p + soff[root], soff[root + 1] - soff[root], hipMemcpyDeviceToDevice, st));
		if(c->nranks > 1){
			RC(r.GroupStart());
			for(int k = 0; k < c->nranks; k++) if(k != root && soff[k + 1] > soff[k]) RC(r.Send((const uint8_t*)c->d_stage.p + soff[k], soff[k + 1] - soff[k], RCCL_UINT8, k, c->comm, st));
			RC(r.GroupEnd());
		}
		HC(hipStreamSynchronize(st));
	} else if(acc){
		RC(r.Recv(c->d_blob.p, acc, RCCL_UINT8, root, c->comm, st));
		HC(hipStreamSynchronize(st));
	}
	return BSA_OK;
}

extern "C" int bsa_shard_gather(bsa_shard_comm_t *c, int root, const bsa_result_t *d_out, const uint32_t *d_cigar, const uint64_t *cigar_off, size_t count,
		bsa_result_t *out, uint32_t *cigar, size_t cigar_cap_words, uint64_t *out_cigar_off, size_t n){
	if(!c || root < 0 || root >= c->nranks || (count && (!d_out || !cigar_off))) return BSA_E_ARG;
	const bool isroot = c->rank == root;
	hipStream_t st; int rc = bsa_ctx_get_stream_internal(c->ctx, &st); if(rc != BSA_OK) return rc;
	Rccl &r = rccl();
	const uint64_t nwords = count ? cigar_off[count] : 0;          // cigar_off: HOST array of count + 1 offsets into d_cigar
	std::vector<uint64_t> sizes(2 * (size_t)c->nranks, 0);
	sizes[2 * c->rank] = count; sizes[2 * c->rank + 1] = nwords;
	if(c->nranks > 1){
		if(!c->d_sizes.need(16 * (size_t)c->nranks + 16)) return BSA_E_NOMEM;
		uint8_t *ds = (uint8_t*)c->d_sizes.p;
		HC(hipMemcpyAsync(ds + 16 * c->nranks, &sizes[2 * c->rank], 16, hipMemcpyHostToDevice, st));
		RC(r.AllGather(ds + 16 * c->nranks, ds, 2, RCCL_UINT64, c->comm, st));
		HC(hipMemcpyAsync(sizes.data(), ds, 16 * (size_t)c->nranks, hipMemcpyDeviceToHost, st)); HC(hipStreamSynchronize(st));
	}
	// the per-pair CIGAR word counts travel as u64 (count of them per rank)
	std::vector<uint64_t> cnt(count);
	for(size_t i = 0; i < count; i++) cnt[i] = cigar_off[i + 1] - cigar_off[i];
	if(!c->d_cnt.need(8 * std::max<size_t>(count, 1))) return BSA_E_NOMEM;
	if(count) HC(hipMemcpyAsync(c->d_cnt.p, cnt.data(), 8 * count, hipMemcpyHostToDevice, st));
	if(!isroot){
		RC(r.GroupStart());
		if(count){ RC(r.Send(d_out, count * sizeof(bsa_result_t), RCCL_UINT8, root, c->comm, st)); RC(r.Send(c->d_cnt.p, count, RCCL_UINT64, root, c->comm, st)); }
		if(nwords && d_cigar) RC(r.Send(d_cigar, nwords * 4, RCCL_UINT8, root, c->comm, st));
		RC(r.GroupEnd());
		HC(hipStreamSynchronize(st));
		return BSA_OK;
	}
	size_t tot = 0, totw = 0;
	std::vector<size_t> p0((size_t)c->nranks + 1, 0), w0((size_t)c->nranks + 1, 0);
	for(int k = 0; k < c->nranks; k++){ p0[k + 1] = p0[k] + (size_t)sizes[2 * k]; w0[k + 1] = w0[k] + (size_t)sizes[2 * k + 1]; }
	tot = p0[c->nranks]; totw = w0[c->nranks];
	if(tot != n || !out || !out_cigar_off) return BSA_E_ARG;
	if(cigar && totw > cigar_cap_words){ out_cigar_off[n] = totw; return BSA_E_CIGAR_CAP; }
	if(!c->d_res.need(tot * sizeof(bsa_result_t) + 64) || !c->d_len.need(8 * tot + 64) || !c->d_cig.need(4 * totw + 64)) return BSA_E_NOMEM;
	uint8_t *dr = (uint8_t*)c->d_res.p, *dc = (uint8_t*)c->d_len.p, *dw = (uint8_t*)c->d_cig.p;
	if(count){
		HC(hipMemcpyAsync(dr + p0[root] * sizeof(bsa_result_t), d_out, count * sizeof(bsa_result_t), hipMemcpyDeviceToDevice, st));
		HC(hipMemcpyAsync(dc + 8 * p0[root], c->d_cnt.p, 8 * count, hipMemcpyDeviceToDevice, st));
		if(nwords && d_cigar) HC(hipMemcpyAsync(dw + 4 * w0[root], d_cigar, 4 * nwords, hipMemcpyDeviceToDevice, st));
	}
	if(c->nranks > 1){
		RC(r.GroupStart());
		for(int k = 0; k < c->nranks; k++){
			if(k == root) continue;
			const size_t ck = (size_t)sizes[2 * k], wk = (size_t)sizes[2 * k + 1];
			if(ck){ RC(r.Recv(dr + p0[k] * sizeof(bsa_result_t), ck * sizeof(bsa_result_t), RCCL_UINT8, k, c->comm, st)); RC(r.Recv(dc + 8 * p0[k], ck, RCCL_UINT64, k, c->comm, st)); }
			if(wk) RC(r.Recv(dw + 4 * w0[k], wk * 4, RCCL_UINT8, k, c->comm, st));
		}
		RC(r.GroupEnd());
	}
	std::vector<uint64_t> allcnt(tot);
	HC(hipMemcpyAsync(out, dr, tot * sizeof(bsa_result_t), hipMemcpyDeviceToHost, st));
	if(tot) HC(hipMemcpyAsync(allcnt.data(), dc, 8 * tot, hipMemcpyDeviceToHost, st));
	if(cigar && totw) HC(hipMemcpyAsync(cigar, dw, 4 * totw, hipMemcpyDeviceToHost, st));
	HC(hipStreamSynchronize(st));
	uint64_t acc = 0;
	for(size_t i = 0; i < tot; i++){ out_cigar_off[i] = acc; acc += allcnt[i]; }
	out_cigar_off[tot] = acc;
	return acc == totw ? BSA_OK : BSA_E_HIP;
}
