// bsa_api.hip -- C-ABI of libbsalign_hip.so (include/bsalign_hip.h): context, plans, batch drivers.
//
// The drivers play the role of the reference's per-pair loop (main.c:311-326 / main.c:194-205): they
// stage the sequences, order the pairs by length so the pairs sharing a wavefront finish together, cut the
// batch into chunks whose traceback rows fit half of the device workspace, and pipeline the chunks over
// two HIP streams: forward DP of chunk k+1 (compute / HBM-write bound) runs on the context stream while
// traceback + CIGAR compaction of chunk k (latency bound, few waves) runs on an auxiliary stream, the two
// halves of the workspace alternating.
#include "bsa_common.h"
#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <string>
#include <vector>
#include <chrono>
#include <thread>
#include <cstdio>
#include <cstring>
#include <cstdlib>

struct bsa_ctx {
	int device = 0;
	hipStream_t own_stream = nullptr;
	hipStream_t stream = nullptr;
	hipStream_t aux_stream = nullptr;        // traceback side of the chunk pipeline
	std::vector<hipEvent_t> sev;             // ordering events of the pipeline (no timing)
	size_t sev_used = 0;
	size_t ws_limit = 0;
	uint8_t *ws = nullptr; size_t ws_bytes = 0;
	std::string err;
	// timing of the dominant kernel in the last run
	std::vector<hipEvent_t> ev;      // pairs (start, stop)
	size_t ev_used = 0;
	double last_cells = 0;
	std::vector<hipEvent_t> tev;     // the same for the traceback launches (on the stream they run on)
	size_t tev_used = 0;
	std::string fwd_name, trace_name;    // kernels behind those two timings
	long last_handover = 0;              // pairs the last bsa_align_batch re-ran through the literal kernels
	double diagdp_ms = 0;
	// small device buffers kept between calls (slot 0: a plan's metadata pool, slot 1: the host-pointer wrapper's buffers): a batch
	// of one pair otherwise spends more time in hipMalloc / hipFree than in its kernels
	size_t budget_last = 0;          // last answer of ctx_ws_budget
	void *keep[2] = {nullptr, nullptr}; size_t keep_bytes[2] = {0, 0}; bool keep_busy[2] = {false, false};
	void *scratch[3] = {nullptr, nullptr, nullptr}; size_t scratch_bytes[3] = {0, 0, 0};      // grown on demand, kept (bsa_ctx_scratch_internal): the POA rows; 2: bsa_poa_graph_host's tables
	void *xq = nullptr; size_t xq_bytes = 0;          // control words + band states of the persistent 8-bit forward kernel (k_align8_fwd_xq), grown on demand
};

static const size_t BSA_KEEP_MAX = (size_t)64 << 20;      // slot 0 (a plan's metadata): larger requests are plain allocations
// slot 1, the host-pointer wrappers' buffers (sequences up, records and CIGAR words down), stays with the context up to 16 GB: a caller that sends batch after
// batch otherwise pays the allocation every time -- 0.4 ms when the memory is at hand, 770 ms measured for the 11.5 GB of the second C3-sized call on a device
// that is 80 % full.  (BSA_KEEP_HOST_MB overrides; the workspace budget is taken from what is free, so a kept buffer only ever makes chunks smaller.)
static size_t keep_max(int slot){
	if(slot != 1) return BSA_KEEP_MAX;
	if(const char *e = bsa_env("BSA_KEEP_HOST_MB")){ const long v = atol(e); if(v >= 0) return (size_t)v << 20; }
	return (size_t)16 << 30;
}
// a device buffer of at least `bytes`; *kept says whether it is the context's (released with ctx_buf_put) or the caller's to free
static hipError_t ctx_buf_get(bsa_ctx *c, int slot, size_t bytes, void **out, bool *kept){
	*kept = false;
	const size_t kmax = keep_max(slot);
	if(bytes <= kmax && !c->keep_busy[slot]){
		if(c->keep_bytes[slot] < bytes){
			if(c->keep[slot]){ (void)hipFree(c->keep[slot]); c->keep[slot] = nullptr; c->keep_bytes[slot] = 0; }
			// small buffers double (a batch of one pair after another), large ones are rounded up to 256 MB
			const size_t want = bytes <= BSA_KEEP_MAX ? std::min(std::max<size_t>(bytes * 2, (size_t)1 << 20), BSA_KEEP_MAX) : ((bytes + (((size_t)256 << 20) - 1)) & ~(((size_t)256 << 20) - 1));
			hipError_t e = hipMalloc(&c->keep[slot], want);
			if(e != hipSuccess){
				// (not that much at hand: a plain allocation of what was asked for)
				c->keep[slot] = nullptr; (void)hipGetLastError();
				return hipMalloc(out, bytes);
			}
			c->keep_bytes[slot] = want;
		}
		c->keep_busy[slot] = true; *kept = true; *out = c->keep[slot];
		return hipSuccess;
	}
	return hipMalloc(out, bytes);
}
static void ctx_buf_put(bsa_ctx *c, int slot, void *ptr, bool kept){
	if(!ptr) return;
	if(kept) c->keep_busy[slot] = false; else (void)hipFree(ptr);
}

// ---- environment snapshot ----
extern char **environ;
// Snapshots are immutable and never freed (a reload makes a new one and leaves the old ones to the pointers already handed out), the
// current one is published through an atomic pointer: bsa_env() takes no lock and what it returns stays valid for the life of the process.
namespace {
typedef std::unordered_map<std::string, std::string> EnvMap;
std::mutex g_env_m;
std::atomic<const EnvMap*> g_env_cur{nullptr};
}
extern "C" void bsa_env_reload(void){
	std::unique_ptr<EnvMap> m(new EnvMap());
	for(char **e = environ; e && *e; e++){
		if(strncmp(*e, "BSA_", 4) != 0) continue;
		const char *eq = strchr(*e, '=');
		if(eq) m->emplace(std::string(*e, (size_t)(eq - *e)), std::string(eq + 1));
	}
	std::lock_guard<std::mutex> lk(g_env_m);
	// deliberately leaked: no owner whose destructor runs at process exit, so a thread (the POA dispatcher) that reads the
	// environment while the process is going down never sees freed memory
	g_env_cur.store(m.release(), std::memory_order_release);
}
const char *bsa_env(const char *name){
	const EnvMap *m = g_env_cur.load(std::memory_order_acquire);
	if(!m){ bsa_env_reload(); m = g_env_cur.load(std::memory_order_acquire); }
	auto it = m->find(name);
	return it == m->end() ? nullptr : it->second.c_str();
}

#define HIPCHK(ctx, call) do { hipError_t _e = (call); if(_e != hipSuccess){ (ctx)->err = std::string(#call) + ": " + hipGetErrorString(_e); return BSA_E_HIP; } } while(0)

extern "C" void bsa_set_score_matrix(int8_t m[16], int8_t mat, int8_t mis){ // bsalign.h:323
	for(int i = 0; i < 16; i++) m[i] = ((i >> 2) == (i & 3)) ? mat : mis;
}

extern "C" int bsa_ctx_create(int device, bsa_ctx_t **out){
	if(!out) return BSA_E_ARG;
	*out = nullptr;
	int ndev = 0;
	if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return BSA_E_NODEVICE;
	if(hipSetDevice(device) != hipSuccess) return BSA_E_NODEVICE;
	bsa_ctx *c = new bsa_ctx();
	(void)bsa_env("BSA_PIPELINE");          // (takes the snapshot of the BSA_* knobs if there is none yet)
	c->device = device;
	if(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess){ delete c; return BSA_E_NODEVICE; }
	if(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) != hipSuccess){ (void)hipStreamDestroy(c->own_stream); delete c; return BSA_E_NODEVICE; }
	c->stream = c->own_stream;
	*out = c;
	return BSA_OK;
}

extern "C" void bsa_ctx_destroy(bsa_ctx_t *c){
	if(!c) return;
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->stream);
	(void)hipStreamSynchronize(c->aux_stream);
	for(hipEvent_t e : c->ev) (void)hipEventDestroy(e);
	for(hipEvent_t e : c->sev) (void)hipEventDestroy(e);
	for(hipEvent_t e : c->tev) (void)hipEventDestroy(e);
	if(c->ws) (void)hipFree(c->ws);
	for(int k = 0; k < 2; k++) if(c->keep[k]) (void)hipFree(c->keep[k]);
	for(int k = 0; k < 3; k++) if(c->scratch[k]) (void)hipFree(c->scratch[k]);
	if(c->xq) (void)hipFree(c->xq);
	if(c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
	if(c->own_stream) (void)hipStreamDestroy(c->own_stream);
	delete c;
}

extern "C" int bsa_ctx_set_stream(bsa_ctx_t *c, void *s){ if(!c) return BSA_E_ARG; c->stream = s ? (hipStream_t)s : c->own_stream; return BSA_OK; }
extern "C" int bsa_ctx_set_workspace_limit(bsa_ctx_t *c, size_t b){ if(!c) return BSA_E_ARG; c->ws_limit = b; return BSA_OK; }
extern "C" int bsa_ctx_sync(bsa_ctx_t *c){ if(!c) return BSA_E_ARG; (void)hipSetDevice(c->device); HIPCHK(c, hipStreamSynchronize(c->stream)); return BSA_OK; }
extern "C" const char *bsa_last_error(bsa_ctx_t *c){ return c ? c->err.c_str() : "no context"; }

extern "C" int bsa_ctx_last_kernel_ms(bsa_ctx_t *c, double *ms, long *launches, double *cells){
	if(!c) return BSA_E_ARG;
	(void)hipSetDevice(c->device);
	HIPCHK(c, hipStreamSynchronize(c->stream));
	double tot = 0; long n = 0;
	for(size_t i = 0; i + 1 < c->ev_used; i += 2){
		float t = 0;
		HIPCHK(c, hipEventElapsedTime(&t, c->ev[i], c->ev[i + 1]));
		tot += t; n++;
	}
	if(ms) *ms = n ? tot / (double)n : 0.0;
	if(launches) *launches = n;
	if(cells) *cells = c->last_cells;
	return BSA_OK;
}

extern "C" int bsa_ctx_last_trace_ms(bsa_ctx_t *c, double *ms, long *launches){
	if(!c) return BSA_E_ARG;
	(void)hipSetDevice(c->device);
	HIPCHK(c, hipStreamSynchronize(c->stream));
	HIPCHK(c, hipStreamSynchronize(c->aux_stream));
	double tot = 0; long n = 0;
	for(size_t i = 0; i + 1 < c->tev_used; i += 2){
		float t = 0;
		HIPCHK(c, hipEventElapsedTime(&t, c->tev[i], c->tev[i + 1]));
		tot += t; n++;
	}
	if(ms) *ms = n ? tot / (double)n : 0.0;
	if(launches) *launches = n;
	return BSA_OK;
}
thread_local const char *bsa_last_fwd_kernel = nullptr, *bsa_last_trace_kernel = nullptr;
extern "C" long bsa_ctx_last_handover(bsa_ctx_t *c){ return c ? c->last_handover : 0; }
extern "C" const char *bsa_ctx_last_kernel_name(bsa_ctx_t *c, int traceback){ return !c ? "" : traceback ? c->trace_name.c_str() : c->fwd_name.c_str(); }

static int ctx_trace_event_pair(bsa_ctx *c, hipEvent_t *a, hipEvent_t *b){
	while(c->tev.size() < c->tev_used + 2){
		hipEvent_t e;
		HIPCHK(c, hipEventCreate(&e));
		c->tev.push_back(e);
	}
	*a = c->tev[c->tev_used]; *b = c->tev[c->tev_used + 1];
	c->tev_used += 2;
	return BSA_OK;
}

static int ctx_event_pair(bsa_ctx *c, hipEvent_t *a, hipEvent_t *b){
	while(c->ev.size() < c->ev_used + 2){
		hipEvent_t e;
		HIPCHK(c, hipEventCreate(&e));
		c->ev.push_back(e);
	}
	*a = c->ev[c->ev_used]; *b = c->ev[c->ev_used + 1];
	c->ev_used += 2;
	return BSA_OK;
}

static int ctx_sync_event(bsa_ctx *c, hipEvent_t *e){
	if(c->sev.size() <= c->sev_used){
		hipEvent_t ne;
		HIPCHK(c, hipEventCreateWithFlags(&ne, hipEventDisableTiming));
		c->sev.push_back(ne);
	}
	*e = c->sev[c->sev_used++];
	return BSA_OK;
}

// `need`: what the plan will ask for in total; a plan that needs less than an eighth of the last answer takes that answer
// (hipMemGetInfo costs more than all the kernels of a one-pair batch)
static size_t ctx_ws_budget(bsa_ctx *c, size_t need = ~(size_t)0){
	if(c->ws_limit) return c->ws_limit;
	if(c->budget_last && need < c->budget_last / 8) return c->budget_last;
	size_t fr = 0, tot = 0;
	if(hipMemGetInfo(&fr, &tot) != hipSuccess) return (size_t)8 << 30;
	// what is free now plus what the context already holds, keep 20% headroom for the caller
	c->budget_last = (size_t)((double)(fr + c->ws_bytes) * 0.8);
	return c->budget_last;
}

static int ctx_ws_reserve(bsa_ctx *c, size_t bytes){
	if(c->ws_bytes >= bytes) return BSA_OK;
	if(c->ws){ HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipStreamSynchronize(c->aux_stream)); (void)hipFree(c->ws); c->ws = nullptr; c->ws_bytes = 0; }
	if(hipMalloc((void**)&c->ws, bytes) != hipSuccess){ c->err = "workspace allocation failed"; (void)hipGetLastError(); return BSA_E_NOMEM; }
	c->ws_bytes = bytes;
	return BSA_OK;
}

// ------------------------------------------------------------------------------------------------
// small utility kernels
// ------------------------------------------------------------------------------------------------

// stage one pair per TPP threads -- a wave for short pairs (a block per pair spent most of its time being launched), the whole block
// for long ones: copy query codes (padded with BSA_QPAD_CODE) and target bytes, validate codes
template<int TPP>
__global__ void __launch_bounds__(256) k_stage(const uint8_t *seqs, const uint64_t *qoff, const uint32_t *qlen,
		const uint64_t *toff, const uint32_t *tlen, const uint64_t *qpoff, const uint64_t *tpoff,
		uint8_t *qst, uint8_t *tst, uint32_t qpad, uint32_t tpad, uint32_t *status, uint32_t n){
	const uint32_t k = (TPP == 64) ? blockIdx.x * 4u + (threadIdx.x >> 6) : blockIdx.x, lane = threadIdx.x & (uint32_t)(TPP - 1);
	if(k >= n) return;
	const uint32_t ql = qlen[k], tl = tlen[k];
	const uint8_t *q = seqs + qoff[k], *t = seqs + toff[k];
	uint8_t *dq = qst + qpoff[k], *dt = tst + tpoff[k];
	uint32_t bad = 0;
	// 16 bytes per thread and trip (the staged regions are 16-byte aligned and a whole number of 16-byte pieces, the source is
	// not aligned and ends with the sequence): whole pieces with two 8-byte loads, the piece that holds the end byte by byte
	auto copy = [&](const uint8_t *src, uint8_t *dst, uint32_t len, uint32_t pad, uint8_t padcode){
		const uint32_t total = (len + pad + 15u) & ~15u;
		for(uint32_t i = lane * 16u; i < total; i += (uint32_t)TPP * 16u){
			uint64_t v0, v1;
			if(i + 16u <= len){
				__builtin_memcpy(&v0, src + i, 8); __builtin_memcpy(&v1, src + i + 8, 8);
				if((v0 | v1) & 0xFCFCFCFCFCFCFCFCull){ bad = 1; v0 &= 0x0303030303030303ull; v1 &= 0x0303030303030303ull; }
			} else {
				v0 = v1 = 0;
#pragma unroll
				for(uint32_t b = 0; b < 16u; b++){
					uint8_t c = (i + b < len) ? src[i + b] : padcode;
					if(i + b < len && c > 3){ bad = 1; c &= 3; }
					if(b < 8u) v0 |= (uint64_t)c << (8u * b); else v1 |= (uint64_t)c << (8u * (b - 8u));
				}
			}
			uint4 o; o.x = (uint32_t)v0; o.y = (uint32_t)(v0 >> 32); o.z = (uint32_t)v1; o.w = (uint32_t)(v1 >> 32);
			*(uint4*)(dst + i) = o;
		}
	};
	copy(q, dq, ql, qpad, (uint8_t)BSA_QPAD_CODE);
	copy(t, dt, tl, tpad, (uint8_t)0);
	uint32_t st = 0;
	if((TPP == 64) ? __any((int)bad) : __syncthreads_or((int)bad)) st |= BSA_ST_BAD_BASE;
	if(ql == 0 || tl == 0) st |= BSA_ST_EMPTY;
	if(lane == 0) status[k] = st;
}

// single-block exclusive scan of cnt[0..n) into off[0..n], off[n] = total; *carry (optional) is a running base
__global__ void __launch_bounds__(1024) k_excl_scan(const uint32_t *cnt, uint64_t *off, uint32_t n, uint64_t *carry){
	__shared__ uint64_t part[1024];
	__shared__ uint64_t base;
	const uint32_t t = threadIdx.x;
	if(t == 0) base = carry ? *carry : 0ull;
	__syncthreads();
	for(uint32_t s = 0; s < n; s += 1024u * 8u){
		uint64_t loc[8]; uint64_t sum = 0;
		for(int k = 0; k < 8; k++){
			uint32_t idx = s + t * 8u + k;
			loc[k] = sum;
			sum += (idx < n) ? cnt[idx] : 0u;
		}
		part[t] = sum;
		__syncthreads();
		for(uint32_t d = 1; d < 1024; d <<= 1){   // Hillis-Steele inclusive scan of the per-thread sums
			uint64_t v = (t >= d) ? part[t - d] : 0ull;
			__syncthreads();
			part[t] += v;
			__syncthreads();
		}
		const uint64_t excl = part[t] - sum + base;
		for(int k = 0; k < 8; k++){
			uint32_t idx = s + t * 8u + k;
			if(idx < n) off[idx] = excl + loc[k];
		}
		__syncthreads();
		if(t == 1023) base += part[1023];
		__syncthreads();
	}
	if(t == 0){ off[n] = base; if(carry) *carry = base; }
}

// The same scan over many blocks for batches of millions of pairs (one block reads 4 bytes per pair at the speed of one CU: 2 ms
// for 2 M pairs, twice per step): sums of tiles of SCAN_TILE counts, a scan of the sums by one block, then every tile scanned with
// its base.  tmp: (tiles + 1) uint64.
#define SCAN_TILE 4096u
__global__ void __launch_bounds__(256) k_scan_tile_sums(const uint32_t *cnt, uint32_t n, uint64_t *tmp){
	__shared__ uint64_t red[4];
	const uint32_t b = blockIdx.x, t = threadIdx.x;
	uint64_t s = 0;
	for(uint32_t i = t; i < SCAN_TILE; i += 256u){ const uint32_t idx = b * SCAN_TILE + i; s += (idx < n) ? cnt[idx] : 0u; }
	for(int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d);
	if((t & 63u) == 0u) red[t >> 6] = s;
	__syncthreads();
	if(t == 0) tmp[b] = red[0] + red[1] + red[2] + red[3];
}
__global__ void __launch_bounds__(1024) k_scan_tile_bases(uint64_t *tmp, uint32_t tiles, uint64_t *carry){
	__shared__ uint64_t part[1024];
	__shared__ uint64_t base;
	const uint32_t t = threadIdx.x;
	if(t == 0) base = carry ? *carry : 0ull;
	__syncthreads();
	for(uint32_t s = 0; s < tiles; s += 1024u){
		const uint64_t v = (s + t < tiles) ? tmp[s + t] : 0ull;
		part[t] = v;
		__syncthreads();
		for(uint32_t d = 1; d < 1024; d <<= 1){
			const uint64_t w = (t >= d) ? part[t - d] : 0ull;
			__syncthreads();
			part[t] += w;
			__syncthreads();
		}
		if(s + t < tiles) tmp[s + t] = part[t] - v + base;
		__syncthreads();
		if(t == 1023) base += part[1023];
		__syncthreads();
	}
	if(t == 0){ tmp[tiles] = base; if(carry) *carry = base; }
}
__global__ void __launch_bounds__(256) k_scan_tile_apply(const uint32_t *cnt, uint64_t *off, uint32_t n, const uint64_t *tmp, uint32_t tiles){
	__shared__ uint64_t wsum[4];
	const uint32_t b = blockIdx.x, t = threadIdx.x;
	const uint32_t i0 = b * SCAN_TILE + t * 16u;            // 16 consecutive counts per thread
	uint32_t loc[16]; uint64_t s = 0;
	for(int k = 0; k < 16; k++){ const uint32_t idx = i0 + k; loc[k] = (uint32_t)s; s += (idx < n) ? cnt[idx] : 0u; }
	uint64_t inc = s;                                       // inclusive scan of the thread sums inside the wave, then across the four waves
	for(int d = 1; d < 64; d <<= 1){ const uint64_t v = __shfl_up(inc, d); if((t & 63u) >= (uint32_t)d) inc += v; }
	if((t & 63u) == 63u) wsum[t >> 6] = inc;
	__syncthreads();
	uint64_t wb = 0;
	for(uint32_t w = 0; w < (t >> 6); w++) wb += wsum[w];
	const uint64_t excl = tmp[b] + wb + inc - s;
	for(int k = 0; k < 16; k++){ const uint32_t idx = i0 + k; if(idx < n) off[idx] = excl + loc[k]; }
	if(b == tiles - 1u && t == 0) off[n] = tmp[tiles];
}

static hipError_t launch_excl_scan(hipStream_t st, const uint32_t *cnt, uint64_t *off, uint32_t n, uint64_t *carry, uint64_t *tmp){
	// (one block takes 111 us for the 100 000 counts of C2, twice a step; in tiles 3 x 5 us)
	if(n <= 16384u || !tmp){ hipLaunchKernelGGL(k_excl_scan, dim3(1), dim3(1024), 0, st, cnt, off, n, carry); return hipGetLastError(); }
	const uint32_t tiles = (n + SCAN_TILE - 1u) / SCAN_TILE;
	hipLaunchKernelGGL(k_scan_tile_sums, dim3(tiles), dim3(256), 0, st, cnt, n, tmp);
	hipLaunchKernelGGL(k_scan_tile_bases, dim3(1), dim3(1024), 0, st, tmp, tiles, carry);
	hipLaunchKernelGGL(k_scan_tile_apply, dim3(tiles), dim3(256), 0, st, cnt, off, n, (const uint64_t*)tmp, tiles);
	return hipGetLastError();
}

// one wave per pair: copy the CIGAR words a traceback kernel left at the tail of the pair's row slot
__global__ void __launch_bounds__(256) k_cigar_collect(const uint8_t *rows, const uint64_t *slot_end,
		uint32_t first, uint32_t count, const uint32_t *cnt, const uint64_t *off, uint32_t *tmp, uint64_t cap){
	const uint32_t g = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = threadIdx.x & 63;
	if(g >= count) return;
	const uint32_t ppos = first + g;
	const uint32_t c = cnt[ppos];
	const uint64_t o = off[ppos];
	if(o + c > cap) return;
	const uint32_t *src = (const uint32_t*)(rows + slot_end[ppos]) - c;
	for(uint32_t i = lane; i < c; i += 64) tmp[o + i] = src[i];
}

__global__ void k_cnt_by_pair(const uint32_t *order, const uint32_t *cnt_pos, const uint64_t *off_pos,
		uint32_t *cnt_pair, uint64_t *src_pair, uint32_t n){
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if(p >= n) return;
	const uint32_t pair = order[p];
	cnt_pair[pair] = cnt_pos[p];
	src_pair[pair] = off_pos ? off_pos[p] : (uint64_t)p;
}

// a plan of ONE chunk: the slots are all still there when the offsets by pair are known, so the words go from the tails of the slots straight to their
// place (src_pair[pair] = the pair's position: k_cnt_by_pair with off_pos == nullptr) -- no pass through the scratch arena (0.25 ms of C2's step)
__global__ void __launch_bounds__(256) k_cigar_final_direct(const uint8_t *rows, const uint64_t *slot_end, const uint32_t *cnt_pair, const uint64_t *pos_pair,
		const uint64_t *dst_off, uint32_t *dst, uint64_t cap, uint32_t n){
	const uint32_t g = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = threadIdx.x & 63;
	if(g >= n) return;
	const uint32_t c = cnt_pair[g];
	const uint64_t d = dst_off[g];
	if(d + c > cap) return;
	const uint32_t *src = (const uint32_t*)(rows + slot_end[pos_pair[g]]) - c;
	for(uint32_t i = lane; i < c; i += 64) dst[d + i] = src[i];
}

__global__ void __launch_bounds__(256) k_cigar_final(const uint32_t *tmp, const uint32_t *cnt_pair, const uint64_t *src_pair,
		const uint64_t *dst_off, uint32_t *dst, uint64_t cap, uint32_t n){
	const uint32_t g = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = threadIdx.x & 63;
	if(g >= n) return;
	const uint32_t c = cnt_pair[g];
	const uint64_t s = src_pair[g], d = dst_off[g];
	if(d + c > cap || s + c > cap) return;
	for(uint32_t i = lane; i < c; i += 64) dst[d + i] = tmp[s + i];
}


// ------------------------------------------------------------------------------------------------
// plans: metadata + staging buffers shared by both paths, chunk pipeline
// ------------------------------------------------------------------------------------------------
struct Sub   { uint32_t first, count, bw; };               // a run of one launch class inside a chunk (forward launches)
struct Chunk { uint32_t first, count, bw; size_t bytes; uint32_t sub0 = 0, nsub = 0; uint32_t max_tlen = 0; };      // bw = BSA_MIXED_BW when the chunk holds several classes
#define BSA_MIXED_BW 0xFFFFFFFFu

struct PlanBase {
	bsa_ctx *ctx = nullptr;
	size_t n = 0;
	double cells = 0;
	std::vector<Chunk> chunks;
	std::vector<Sub> subs;          // edit plans: forward launches per class, one traceback per chunk
	size_t half_bytes = 0;          // size of one workspace half (0 or 1 chunk in flight per half)
	bool two_halves = false;
	uint32_t nbuf = 1;              // workspace regions of half_bytes each; chunk k uses region k % nbuf
	// device metadata
	uint64_t *d_qoff = nullptr, *d_toff = nullptr, *d_qpoff = nullptr, *d_tpoff = nullptr, *d_slot = nullptr, *d_slot_end = nullptr;
	uint32_t *d_qlen = nullptr, *d_tlen = nullptr, *d_order = nullptr;
	uint8_t *d_qst = nullptr, *d_tst = nullptr;
	uint32_t *d_cnt_pos = nullptr, *d_cnt_pair = nullptr, *d_status_own = nullptr;
	uint64_t *d_off_pos = nullptr, *d_src_pair = nullptr, *d_carry = nullptr, *d_scan_tmp = nullptr;
	uint32_t *d_tmp = nullptr; size_t tmp_words = 0;
	void *pool = nullptr;           // one allocation behind all the metadata pointers above (plan_common_alloc)
	bool pool_kept = false;         // ... which is the context's kept buffer (ctx_buf_get)
	std::vector<void*> extra;       // path-specific device allocations
	virtual ~PlanBase(){}
};

template<typename T> static int dev_upload(bsa_ctx *c, T **dst, const std::vector<T> &src){
	size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
	if(hipMalloc((void**)dst, bytes) != hipSuccess){ c->err = "metadata allocation failed"; (void)hipGetLastError(); return BSA_E_NOMEM; }
	if(!src.empty()) HIPCHK(c, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
	return BSA_OK;
}
template<typename T> static int dev_alloc(bsa_ctx *c, T **dst, size_t count){
	if(hipMalloc((void**)dst, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess){ c->err = "device allocation failed"; (void)hipGetLastError(); return BSA_E_NOMEM; }
	return BSA_OK;
}

static void plan_free(PlanBase *p){
	if(!p) return;
	(void)hipSetDevice(p->ctx->device);
	(void)hipStreamSynchronize(p->ctx->stream);
	(void)hipStreamSynchronize(p->ctx->aux_stream);
	if(p->pool){ ctx_buf_put(p->ctx, 0, p->pool, p->pool_kept); if(p->d_tmp) (void)hipFree(p->d_tmp); }
	else {
		void *ptrs[] = { p->d_qoff, p->d_toff, p->d_qpoff, p->d_tpoff, p->d_slot, p->d_slot_end, p->d_qlen, p->d_tlen, p->d_order, p->d_qst, p->d_tst,
		                 p->d_cnt_pos, p->d_cnt_pair, p->d_status_own, p->d_off_pos, p->d_src_pair, p->d_carry, p->d_tmp };
		for(void *q : ptrs) if(q) (void)hipFree(q);
	}
	for(void *q : p->extra) if(q) (void)hipFree(q);
	delete p;
}

// order[] (initially 0..n-1) sorted by key, stable: LSD radix sort on 16-bit digits, digits that are the same for all
// keys are skipped (batches of millions of short pairs: a comparison sort costs more than the kernels)
static void radix_sort_order(std::vector<uint64_t> &key, std::vector<uint32_t> &order){
	const size_t n = key.size();
	if(n < 2) return;
	uint64_t all_or = 0, all_and = ~0ull;
	for(size_t k = 0; k < n; k++){ all_or |= key[k]; all_and &= key[k]; }
	const uint64_t varying = all_or ^ all_and;
	std::vector<uint64_t> k2(n); std::vector<uint32_t> o2(n);
	std::vector<uint32_t> cnt(65537);
	uint64_t *ks = key.data(), *kd = k2.data(); uint32_t *os = order.data(), *od = o2.data();
	for(int sh = 0; sh < 64; sh += 16){
		if(((varying >> sh) & 0xFFFFull) == 0) continue;
		std::fill(cnt.begin(), cnt.end(), 0u);
		for(size_t k = 0; k < n; k++) cnt[((ks[k] >> sh) & 0xFFFFull) + 1] ++;
		for(size_t v = 0; v < 65536; v++) cnt[v + 1] += cnt[v];
		for(size_t k = 0; k < n; k++){ const uint32_t d = cnt[(ks[k] >> sh) & 0xFFFFull] ++; kd[d] = ks[k]; od[d] = os[k]; }
		std::swap(ks, kd); std::swap(os, od);
	}
	if(os != order.data()) memcpy(order.data(), os, n * sizeof(uint32_t));
}

// cut the processing order into chunks: every chunk fits one workspace half, has a single bandwidth, and large
// batches are cut into >= 4 chunks so that the traceback of one chunk hides behind the forward pass of the next
static int plan_chunks(PlanBase *p, const std::vector<uint32_t> &order, const std::vector<size_t> &need, const std::vector<uint32_t> &bwv,
		std::vector<uint64_t> &slot, std::vector<uint64_t> &slot_end, bool mix_classes = false, bool pipe_default = false){
	bsa_ctx *c = p->ctx;
	const size_t n = order.size();
	size_t total = 0, biggest = 0;
	// (slots start at multiples of 256 bytes -- bsa_begs_bytes; a slot's END, where the walkers leave their CIGAR words, stays where the pair's own need puts it)
	auto up256 = [](size_t b){ return (b + 255) & ~(size_t)255; };
	for(size_t pos = 0; pos < n; pos++){ total += up256(need[pos]); biggest = std::max(biggest, up256(need[pos])); }
	const size_t budget = ctx_ws_budget(c, total);
	if(biggest > budget){ c->err = "workspace limit too small for one pair"; return BSA_E_NOMEM; }
	// Both kernels are row-serial per pair, so throughput = pairs in flight / per-pair latency: chunks are made as
	// large as memory allows and run back to back on the context stream.  Splitting the workspace in two halves and
	// running the traceback of chunk k beside the forward pass of chunk k+1 on a second stream is SLOWER on MI355X
	// (100k x 10 kbp, bw 128: 284-302 ms per step for chunk sizes 16k..33k pairs against 256 ms back to back; kernel
	// trace: the forward pass of a third of the batch takes 86-88 ms beside a traceback instead of 57 ms, and that
	// traceback 61-89 ms instead of 34 ms).  The packed forward kernel fills the register file (4 waves x 128 VGPRs
	// per SIMD), so every resident traceback wave evicts a forward wave, and raising the traceback's wave priority
	// changes nothing.  The mode stays opt-in (BSA_PIPELINE=1); BSA_CHUNK_PAIRS caps the pairs per chunk (tuning knob).
	size_t cap;            // bytes per chunk
	const char *pe = bsa_env("BSA_PIPELINE");
	const bool want_pipe = pipe_default ? !(pe && pe[0] == '0') : (pe && pe[0] == '1');
	size_t pipe_chunks = 4;
	if(const char *ke = bsa_env("BSA_PIPE_CHUNKS")){ const long v = atol(ke); if(v > 0) pipe_chunks = (size_t)v; }
	size_t cap_pairs = n ? n : 1;
	if(const char *ce = bsa_env("BSA_CHUNK_PAIRS")){ const long v = atol(ce); if(v > 0) cap_pairs = (size_t)v; }
	bool all_resident = false;      // pipelined and everything fits: every chunk gets a region of its own
	if(!want_pipe){ cap = std::max(std::min(total, budget), biggest); p->two_halves = false; }
	else if(total + (total / std::max<size_t>(n, 1)) * 64 <= budget && pipe_chunks > 1 && n >= 4096){
		cap = total; p->two_halves = true; all_resident = true;
		if(cap_pairs >= n) cap_pairs = (n + pipe_chunks - 1) / pipe_chunks;
	} else if(total <= budget){ cap = std::max(total, biggest); p->two_halves = false; }
	else {
		cap = std::max(budget / 2, biggest);
		p->two_halves = (2 * cap <= budget);
		if(!p->two_halves) cap = std::max(budget, biggest);
	}
	slot.assign(n, 0); slot_end.assign(n, 0);
	size_t acc = 0, maxacc = 0; uint32_t first = 0;
	// mix_classes (edit plans): a chunk is whatever fits the workspace; its launch classes become forward launches
	// (subs) and the whole chunk is traced in one launch -- many small classes would otherwise each pay the latency of
	// a row-serial kernel of their own
	auto close = [&](uint32_t end){
		Chunk ch{first, end - first, bwv[first], acc};
		ch.sub0 = (uint32_t)p->subs.size();
		for(uint32_t b = first; b < end; ){
			uint32_t e = b + 1;
			while(e < end && bwv[e] == bwv[b]) e++;
			p->subs.push_back({b, e - b, bwv[b]});
			b = e;
		}
		ch.nsub = (uint32_t)p->subs.size() - ch.sub0;
		if(ch.nsub > 1) ch.bw = BSA_MIXED_BW;
		p->chunks.push_back(ch);
		maxacc = std::max(maxacc, acc);
	};
	for(size_t pos = 0; pos < n; pos++){
		if(pos > first && (acc + up256(need[pos]) > cap || (!mix_classes && bwv[pos] != bwv[first]) || pos - first >= cap_pairs)){
			close((uint32_t)pos);
			first = (uint32_t)pos; acc = 0;
		}
		slot[pos] = acc; slot_end[pos] = acc + need[pos]; acc += up256(need[pos]);
	}
	if(n > first) close((uint32_t)n);
	p->half_bytes = (maxacc + 255) & ~(size_t)255;
	if(p->chunks.size() < 2) p->two_halves = false;
	// every chunk in a region of its own: the regions are all as large as the largest chunk, which with mixed lengths (chunks are cut by
	// pair count over pairs sorted by length) can be several times the total -- then the chunks share one region, back to back
	if(all_resident && p->half_bytes * p->chunks.size() > budget){ all_resident = false; p->two_halves = (2 * p->half_bytes <= budget); }
	p->nbuf = !p->two_halves ? 1u : all_resident ? (uint32_t)p->chunks.size() : 2u;
	return BSA_OK;
}

static int plan_common_alloc(PlanBase *p, const uint64_t *qoff, const uint32_t *qlen, const uint64_t *toff, const uint32_t *tlen,
		const std::vector<uint64_t> &qpoff, const std::vector<uint64_t> &tpoff, const std::vector<uint64_t> &slot,
		const std::vector<uint64_t> &slot_end, const std::vector<uint32_t> &order, size_t qst_bytes, size_t tst_bytes){
	bsa_ctx *c = p->ctx; const size_t n = p->n; int rc;
	// ONE device allocation and ONE copy for all of a plan's metadata (a batch of one pair -- the reference-named single-pair
	// functions, the POA's per-read calls -- used to pay seventeen hipMalloc and nine hipMemcpy calls here)
	struct Part { void **dst; const void *src; size_t bytes, off; };
	const size_t m = std::max<size_t>(n, 1);
	Part parts[] = {
		{(void**)&p->d_qoff, qoff, n * 8, 0}, {(void**)&p->d_toff, toff, n * 8, 0}, {(void**)&p->d_qlen, qlen, n * 4, 0}, {(void**)&p->d_tlen, tlen, n * 4, 0},
		{(void**)&p->d_qpoff, qpoff.data(), n * 8, 0}, {(void**)&p->d_tpoff, tpoff.data(), n * 8, 0}, {(void**)&p->d_slot, slot.data(), n * 8, 0},
		{(void**)&p->d_slot_end, slot_end.data(), n * 8, 0}, {(void**)&p->d_order, order.data(), n * 4, 0},
		// (no content to upload from here on)
		{(void**)&p->d_qst, nullptr, std::max<size_t>(qst_bytes, 1), 0}, {(void**)&p->d_tst, nullptr, std::max<size_t>(tst_bytes, 1), 0},
		{(void**)&p->d_cnt_pos, nullptr, m * 4, 0}, {(void**)&p->d_cnt_pair, nullptr, m * 4, 0}, {(void**)&p->d_status_own, nullptr, m * 4, 0},
		{(void**)&p->d_off_pos, nullptr, (m + 1) * 8, 0}, {(void**)&p->d_src_pair, nullptr, m * 8, 0}, {(void**)&p->d_carry, nullptr, 8, 0},
		{(void**)&p->d_scan_tmp, nullptr, (m / SCAN_TILE + 2) * 8, 0},
	};
	size_t total = 0, upload = 0;
	for(Part &q : parts){ q.off = total; total += (std::max<size_t>(q.bytes, 8) + 255) & ~(size_t)255; if(q.src) upload = total; }
	if(ctx_buf_get(c, 0, total, &p->pool, &p->pool_kept) != hipSuccess){ p->pool = nullptr; c->err = "metadata allocation failed"; (void)hipGetLastError(); return BSA_E_NOMEM; }
	if(n){
		std::vector<uint8_t> stage(upload);
		for(const Part &q : parts) if(q.src && q.bytes) memcpy(stage.data() + q.off, q.src, q.bytes);
		HIPCHK(c, hipMemcpy(p->pool, stage.data(), upload, hipMemcpyHostToDevice));
	}
	for(Part &q : parts) *q.dst = (uint8_t*)p->pool + q.off;
	rc = ctx_ws_reserve(c, p->half_bytes * p->nbuf);
	return rc;
}

// The chunk pipeline.  fwd(chunk, ws_half, stream) launches the forward DP; trace(chunk, ws_half, stream) launches the
// traceback that fills d_out / d_cnt_pos.  Ordering: fwd(k) waits for trace(k-2) (same half), trace(k) waits for fwd(k).
template<class FwdFn, class TraceFn>
static int run_pipeline(PlanBase *p, bool want_cig, uint32_t *d_cigar, size_t cigar_cap_words, uint64_t *d_cigar_off, FwdFn fwd, TraceFn trace){
	bsa_ctx *c = p->ctx;
	hipStream_t sf = c->stream, stt = p->two_halves ? c->aux_stream : c->stream;
	const uint32_t n = (uint32_t)p->n;
	int rc;
	c->sev_used = 0;
	const bool direct = want_cig && p->chunks.size() == 1 && !p->two_halves && !bsa_env("BSA_CIGAR_VIA_ARENA");      // (see k_cigar_final_direct)
	HIPCHK(c, hipMemsetAsync(p->d_carry, 0, sizeof(uint64_t), sf));
	std::vector<hipEvent_t> trace_done(p->chunks.size());
	if(p->two_halves){
		// the auxiliary stream must not start before the staging work already queued on the main stream
		hipEvent_t e; rc = ctx_sync_event(c, &e); if(rc != BSA_OK) return rc;
		HIPCHK(c, hipEventRecord(e, sf)); HIPCHK(c, hipStreamWaitEvent(stt, e, 0));
	}
	for(size_t k = 0; k < p->chunks.size(); k++){
		const Chunk &ch = p->chunks[k];
		uint8_t *half = c->ws + (k % p->nbuf) * p->half_bytes;
		if(p->two_halves && k >= p->nbuf) HIPCHK(c, hipStreamWaitEvent(sf, trace_done[k - p->nbuf], 0));
		hipEvent_t e0, e1;
		rc = ctx_event_pair(c, &e0, &e1); if(rc != BSA_OK) return rc;
		HIPCHK(c, hipEventRecord(e0, sf));
		rc = fwd(ch, half, sf); if(rc != BSA_OK) return rc;
		HIPCHK(c, hipEventRecord(e1, sf));
		if(p->two_halves) HIPCHK(c, hipStreamWaitEvent(stt, e1, 0));
		hipEvent_t t0, t1;
		rc = ctx_trace_event_pair(c, &t0, &t1); if(rc != BSA_OK) return rc;
		HIPCHK(c, hipEventRecord(t0, stt));
		rc = trace(ch, half, stt); if(rc != BSA_OK) return rc;
		HIPCHK(c, hipEventRecord(t1, stt));
		if(want_cig && !direct){
			HIPCHK(c, launch_excl_scan(stt, p->d_cnt_pos + ch.first, p->d_off_pos + ch.first, ch.count, p->d_carry, p->d_scan_tmp));
			hipLaunchKernelGGL(k_cigar_collect, dim3((ch.count + 3) / 4), dim3(256), 0, stt, half, p->d_slot_end,
				ch.first, ch.count, p->d_cnt_pos, p->d_off_pos, p->d_tmp, (uint64_t)cigar_cap_words);
			HIPCHK(c, hipGetLastError());
		}
		if(p->two_halves){
			rc = ctx_sync_event(c, &trace_done[k]); if(rc != BSA_OK) return rc;
			HIPCHK(c, hipEventRecord(trace_done[k], stt));
		}
	}
	if(p->two_halves && !p->chunks.empty()){
		for(size_t k = p->chunks.size() > p->nbuf ? p->chunks.size() - p->nbuf : 0; k < p->chunks.size(); k++)
			HIPCHK(c, hipStreamWaitEvent(sf, trace_done[k], 0));
	}
	c->last_cells = p->cells;
	if(want_cig){
		hipLaunchKernelGGL(k_cnt_by_pair, dim3((n + 255) / 256), dim3(256), 0, sf, p->d_order, p->d_cnt_pos, direct ? (const uint64_t*)nullptr : p->d_off_pos, p->d_cnt_pair, p->d_src_pair, n);
		HIPCHK(c, hipGetLastError());
		HIPCHK(c, launch_excl_scan(sf, p->d_cnt_pair, d_cigar_off, n, (uint64_t*)nullptr, p->d_scan_tmp));
		if(direct) hipLaunchKernelGGL(k_cigar_final_direct, dim3((n + 3) / 4), dim3(256), 0, sf, c->ws, p->d_slot_end, p->d_cnt_pair, p->d_src_pair, d_cigar_off, d_cigar, (uint64_t)cigar_cap_words, n);
		else hipLaunchKernelGGL(k_cigar_final, dim3((n + 3) / 4), dim3(256), 0, sf, p->d_tmp, p->d_cnt_pair, p->d_src_pair, d_cigar_off, d_cigar, (uint64_t)cigar_cap_words, n);
		HIPCHK(c, hipGetLastError());
	} else if(d_cigar_off){
		HIPCHK(c, hipMemsetAsync(d_cigar_off, 0, sizeof(uint64_t) * ((size_t)n + 1), sf));
	}
	return BSA_OK;
}

static int run_prologue(PlanBase *p, bool want_cig, size_t cigar_cap_words){
	bsa_ctx *c = p->ctx;
	(void)hipSetDevice(c->device);
	c->ev_used = 0; c->tev_used = 0; c->last_cells = 0;
	int rc = ctx_ws_reserve(c, p->half_bytes * p->nbuf);
	if(rc != BSA_OK) return rc;
	const bool direct = p->chunks.size() == 1 && !p->two_halves && !bsa_env("BSA_CIGAR_VIA_ARENA");          // (run_pipeline: no pass through the arena)
	if(want_cig && !direct && p->tmp_words < cigar_cap_words){
		if(p->d_tmp){ HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipStreamSynchronize(c->aux_stream)); (void)hipFree(p->d_tmp); p->d_tmp = nullptr; }
		if(hipMalloc((void**)&p->d_tmp, std::max<size_t>(cigar_cap_words, 1) * 4) != hipSuccess){ c->err = "cigar staging allocation failed"; (void)hipGetLastError(); return BSA_E_NOMEM; }
		p->tmp_words = cigar_cap_words;
	}
	return BSA_OK;
}

static hipError_t par_copy(int device, void *dst, const void *src, size_t bytes, hipMemcpyKind kind);      // (below: large pageable copies in pieces on threads of their own)
// host-pointer convenience wrapper shared by both paths: copy in, run, copy out, synchronise.  prep() is the host-side planning (bsa_*_plan_create:
// tens of milliseconds for 100 k pairs): for a large blob it runs on the calling thread WHILE a helper thread feeds the upload -- a copy from pageable
// memory keeps its host thread busy for its whole duration -- so the two no longer add up (C2: 29 ms of planning under a 50 ms upload).
template<class PrepFn, class RunFn>
static int batch_host(bsa_ctx *c, const uint8_t *seqs, size_t seqs_bytes, size_t n, bsa_result_t *out, uint32_t *cigar,
		size_t cigar_cap_words, uint64_t *cigar_off, uint32_t *status, PrepFn prep, RunFn run){
	uint8_t *d_seqs = nullptr; bsa_result_t *d_out = nullptr; uint32_t *d_cig = nullptr, *d_status = nullptr; uint64_t *d_off = nullptr;
	uint8_t *pool = nullptr; bool pool_kept = false;            // one allocation for the five buffers
	auto cleanup = [&](){ ctx_buf_put(c, 1, pool, pool_kept); pool = nullptr; };
	const bool want_cig = cigar && cigar_off;
#define TRYH(call) do { hipError_t _e = (call); if(_e != hipSuccess){ c->err = std::string(#call) + ": " + hipGetErrorString(_e); cleanup(); return BSA_E_HIP; } } while(0)
	auto up = [](size_t b){ return (std::max<size_t>(b, 8) + 255) & ~(size_t)255; };
	const size_t o_seqs = 0, o_out = o_seqs + up(seqs_bytes), o_st = o_out + up(n * sizeof(bsa_result_t)), o_off = o_st + up(n * sizeof(uint32_t));
	const size_t o_cig = o_off + (want_cig ? up((n + 1) * sizeof(uint64_t)) : 0), total = o_cig + (want_cig ? up(cigar_cap_words * 4) : 0);
	const bool tmg = bsa_env("BSA_API_TIMING") != nullptr;          // (stderr: where a host-pointer batch spends its wall time)
	const auto ts0 = std::chrono::steady_clock::now();
	auto since = [&](){ return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts0).count(); };
	double t_buf = 0, t_prep = 0, t_run = 0, t_res = 0;
	TRYH(ctx_buf_get(c, 1, total, (void**)&pool, &pool_kept));
	t_buf = since();
	d_seqs = pool + o_seqs; d_out = (bsa_result_t*)(pool + o_out); d_status = (uint32_t*)(pool + o_st);
	if(want_cig){ d_off = (uint64_t*)(pool + o_off); d_cig = (uint32_t*)(pool + o_cig); }
	int rc;
	if(seqs_bytes >= ((size_t)64 << 20) && !bsa_env("BSA_BATCH_NO_UPLOAD_THREAD")){
		hipError_t uperr = hipSuccess;
		std::thread upl([&]{ (void)hipSetDevice(c->device); uperr = par_copy(c->device, d_seqs, seqs, seqs_bytes, hipMemcpyHostToDevice); });      // (arrived when it returns: the run's kernels need no event)
		rc = prep();
		upl.join();
		if(uperr != hipSuccess){ c->err = std::string("upload of the sequences: ") + hipGetErrorString(uperr); cleanup(); return BSA_E_HIP; }
	} else {
		TRYH(hipMemcpyAsync(d_seqs, seqs, seqs_bytes, hipMemcpyHostToDevice, c->stream));
		rc = prep();
	}
	if(rc != BSA_OK){ (void)hipStreamSynchronize(c->stream); cleanup(); return rc; }
	t_prep = since();
	rc = run(d_seqs, d_out, d_cig, d_off, d_status);
	if(rc != BSA_OK){ cleanup(); return rc; }
	t_run = since();
	if(total - o_out <= ((size_t)1 << 20)){
		// a small batch: everything that goes back in ONE copy (results, status, offsets, the whole arena), then handed out
		std::vector<uint8_t> back(total - o_out);
		TRYH(hipMemcpyAsync(back.data(), pool + o_out, back.size(), hipMemcpyDeviceToHost, c->stream));
		TRYH(hipStreamSynchronize(c->stream));
		memcpy(out, back.data(), n * sizeof(bsa_result_t));
		if(status) memcpy(status, back.data() + (o_st - o_out), n * sizeof(uint32_t));
		if(want_cig){
			memcpy(cigar_off, back.data() + (o_off - o_out), (n + 1) * sizeof(uint64_t));
			if(cigar_off[n] > cigar_cap_words){ cleanup(); c->err = "cigar arena too small"; return BSA_E_CIGAR_CAP; }
			memcpy(cigar, back.data() + (o_cig - o_out), cigar_off[n] * 4);
		}
		cleanup();
		return BSA_OK;
	}
	TRYH(hipMemcpyAsync(out, d_out, n * sizeof(bsa_result_t), hipMemcpyDeviceToHost, c->stream));
	if(status) TRYH(hipMemcpyAsync(status, d_status, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
	if(want_cig) TRYH(hipMemcpyAsync(cigar_off, d_off, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
	TRYH(hipStreamSynchronize(c->stream));
	t_res = since();
	if(want_cig){
		if(cigar_off[n] > cigar_cap_words){ cleanup(); c->err = "cigar arena too small"; return BSA_E_CIGAR_CAP; }
		TRYH(par_copy(c->device, cigar, d_cig, cigar_off[n] * 4, hipMemcpyDeviceToHost));
	}
#undef TRYH
	const double t_cig = since();
	cleanup();
	if(tmg) fprintf(stderr, "[host-pointer batch] %zu pairs, %.1f MB up, %.1f MB of CIGAR words down: ms since entry -- buffers %.1f, plan + upload %.1f, launched %.1f, kernels done + records back %.1f, CIGAR words back %.1f, released %.1f\n",
		n, seqs_bytes / 1e6, want_cig ? cigar_off[n] * 4 / 1e6 : 0.0, t_buf, t_prep, t_run, t_res, t_cig, since());
	return BSA_OK;
}

// ------------------------------------------------------------------------------------------------
// 8-bit alignment plan (banded_striped_epi8_seqalign_pairwise, bsalign.h:3854)
// ------------------------------------------------------------------------------------------------
struct bsa_align_plan : PlanBase {
	bsa_align_params_t par;
	uint32_t bw = 0, rowb = 0; int pw = 0;       // bw == 0: per-pair bandwidth = roundup(qlen, 16)
	bool generic = false;                        // run the LDS-resident generic kernels
	bool codes = false;                          // compact 4-bit-code traceback (global mode, bsa_align8_pk.hip CODES)
	uint32_t max_bw = 0;
	uint32_t ref_bw = 0;                         // a whole-query band widened to bw: the reference's own bandwidth (1 = per pair), see bsa_align_plan_create
	bool static_band = false;                    // no query is longer than the band: it never moves
	bool sys = false;                            // whole-query bands above 256 columns, global mode: the systolic wavefront (bsa_align8_sys.hip)
	bool sys_chk = false;                        // ... with scores outside the static guard: the kernel checks every pair (Align8Args::sys_chk)
	uint32_t max_qlen = 0;
	size_t stage_bytes = 0;                      // staged bytes of all pairs (which staging kernel)
	uint32_t qpad = 0, tpad = 16;
};

extern "C" void bsa_align_plan_destroy(bsa_align_plan_t *p){ plan_free(p); }
extern "C" double bsa_align_plan_cells(const bsa_align_plan_t *p){ return p ? p->cells : 0.0; }

// Can a whole-query band of `cols` columns (a multiple of 16, at most 256) run at a register-kernel width on the compact path?
// Returns that width, or 0.  (See bsa_align_plan_create.)
static uint32_t align8_widened_bw(const bsa_align_params_t *par, uint32_t cols){
	const int type = par->mode & 3;
	const char *we = bsa_env("BSA_ALIGN8_WIDEN");
	const char *le = bsa_env("BSA_ALIGN8_LITERAL");
	if((par->mode & BSA_MODE_ROWRECORDS) || (we && we[0] == '0') || (le && le[0] == '1') || cols == 0u || cols > 256u) return 0u;
	// the gap model must not depend on the width (bsalign.h:2084-2092 compares a ratio of the penalties with it)
	const int pwa = bsa_get_piecewise(par->gapo1, par->gape1, par->gapo2, par->gape2, 16);
	const int pwb = bsa_get_piecewise(par->gapo1, par->gape1, par->gapo2, par->gape2, 256);
	if(pwa != pwb) return 0u;
	// (two-piece gaps: the compact path exists at bandwidth 128 in global mode only -- the checks below say so)
	const uint32_t kbw = (cols <= 64u && pwa <= 1) ? 64u : cols <= 128u ? 128u : 256u;
	Align8Args t;
	memset(&t, 0, sizeof(t));
	t.bw = kbw; t.mode = type; t.gapo1 = par->gapo1; t.gape1 = par->gape1; t.gapo2 = par->gapo2; t.gape2 = par->gape2;
	int smax = -127, smin = 127;
	for(int i = 0; i < 16; i++){ smax = std::max(smax, (int)par->matrix[i]); smin = std::min(smin, (int)par->matrix[i]); }
	t.smax = smax; t.smin = smin;
	return (bsa_align8_codes_supported(t, pwa) && bsa_align8_x_supported(t, pwa)) ? kbw : 0u;
}

extern "C" int bsa_align_plan_create(bsa_ctx_t *c, const uint64_t *qoff, const uint32_t *qlen,
		const uint64_t *toff, const uint32_t *tlen, size_t n, const bsa_align_params_t *par, bsa_align_plan_t **out){
	if(!c || !out || !par || (n && (!qoff || !qlen || !toff || !tlen))) return BSA_E_ARG;
	*out = nullptr;
	if(n > 0xFFFFFFF0ull) { c->err = "too many pairs"; return BSA_E_ARG; }
	const int type = par->mode & 3;
	if(type != BSA_MODE_GLOBAL && type != BSA_MODE_OVERLAP && type != BSA_MODE_EXTEND){ c->err = "bad mode"; return BSA_E_ARG; }
	(void)hipSetDevice(c->device);
	const uint32_t bw_req = (par->bandwidth + 15u) / 16u * 16u;   // bsalign.h:3862; 0 = per pair roundup(qlen, 16) (bsalign.h:3861)
	uint32_t bw = bw_req;
	uint32_t max_bw = bw;
	if(bw == 0) for(size_t k = 0; k < n; k++) max_bw = std::max(max_bw, (qlen[k] + 15u) / 16u * 16u);
	// Whole-query bands in global mode (the reference CLI's default `-W 0`, or a bandwidth no shorter than any query): the band
	// never moves (bsalign.h:3338: qoff + bw >= qlen), cells beyond the query end hold the -63 padding and nothing flows from them
	// to the cells in front, so inside the exact-arithmetic guard the result does not depend on how wide the band is or how it is
	// striped.  Such a batch runs at the next width the register kernels have (64 / 128 / 256) on the compact path instead of the
	// LDS-resident run-time-width kernel; `cells` keeps the reference's own band widths.  Overlap / extend: the one thing that sees
	// the striping is row_max on the last row (its tie rules, bsalign.h:3213-3329) -- the traceback kernel takes it over the
	// reference's band in the reference's striping (Align8Args::ref_bw, codes_end_cell).  BSA_ALIGN8_WIDEN=0 keeps the old dispatch.
	bool widened = false;
	{
		bool full = (bw == 0 || !bsa_align8_supported_bw(bw)) && max_bw <= 256u && n > 0;
		if(full && bw != 0) for(size_t k = 0; k < n && full; k++) full = qlen[k] <= bw;
		const uint32_t kbw = full ? align8_widened_bw(par, max_bw) : 0u;
		if(kbw){ bw = kbw; max_bw = kbw; widened = true; }
	}
	// Whole-query bands above 256 columns (the reference CLI's default on long reads; `bsalign align` defaults to overlap mode): all three modes, linear or affine gaps, scores inside
	// the exact-arithmetic guard -> the systolic wavefront with its own code rows and traceback (bsa_align8_sys.hip) instead of the
	// LDS-resident run-time-width kernel.  BSA_ALIGN8_SYS=0 keeps the old dispatch.
	bool sys = false, sys_chk = false; uint32_t max_qlen = 0;
	for(size_t k = 0; k < n; k++) max_qlen = std::max(max_qlen, qlen[k]);
	// (two-piece gaps: the register kernels stop at 128 columns, so the systolic kernel takes over from there)
	const uint32_t sys_from = (bsa_get_piecewise(par->gapo1, par->gape1, par->gapo2, par->gape2, (int)std::max(max_bw, 16u)) == 2) ? 128u : 256u;
	if(!widened && n > 0 && (bw == 0 || !bsa_align8_supported_bw(bw)) && max_bw > sys_from && max_qlen <= 60000u && !(par->mode & BSA_MODE_ROWRECORDS)){
		bool full = true;
		if(bw != 0) for(size_t k = 0; k < n && full; k++) full = qlen[k] <= bw;
		const char *se = bsa_env("BSA_ALIGN8_SYS"), *le = bsa_env("BSA_ALIGN8_LITERAL");
		if(full && !(se && se[0] == '0') && !(le && le[0] == '1')){
			Align8Args t;
			memset(&t, 0, sizeof(t));
			t.mode = type; t.gapo1 = par->gapo1; t.gape1 = par->gape1; t.gapo2 = par->gapo2; t.gape2 = par->gape2;
			int smax = -127, smin = 127;
			for(int i = 0; i < 16; i++){ smax = std::max(smax, (int)par->matrix[i]); smin = std::min(smin, (int)par->matrix[i]); }
			t.smax = smax; t.smin = smin;
			const int pwa = bsa_get_piecewise(par->gapo1, par->gape1, par->gapo2, par->gape2, 16), pwb = bsa_get_piecewise(par->gapo1, par->gape1, par->gapo2, par->gape2, (int)max_bw);
			const int lvl = pwa == pwb ? bsa_align8_sys_supported(t, pwb) : 0;
			const char *ce = bsa_env("BSA_ALIGN8_SYS_CHK");                 // =0: scores outside the static guard keep the run-time-width kernel; =1: the checked kernel inside the guard as well (tests)
			sys = lvl == 1 || (lvl == 2 && !(ce && ce[0] == '0'));
			sys_chk = sys && (lvl == 2 || (ce && ce[0] == '1'));
			if(sys){
				// the kernel carries scores times 32 in 32-bit registers: |H| <= (qlen + tlen) x the largest step
				uint32_t max_tlen = 0;
				for(size_t k = 0; k < n; k++) max_tlen = std::max(max_tlen, tlen[k]);
				const long long step = std::max(std::max(-(long long)par->gapo1 - par->gape1, -(long long)par->gapo2 - par->gape2), std::max((long long)smax, -(long long)smin));
				if(((long long)max_qlen + max_tlen) * std::max(step, 1ll) >= (1ll << 25)) sys = false;
			}
		}
	}
	bsa_align_plan *p = new bsa_align_plan();
	p->ctx = c; p->n = n; p->par = *par; p->bw = bw;
	p->sys = sys; p->sys_chk = sys && sys_chk; p->max_qlen = max_qlen;
	p->ref_bw = widened ? (bw_req ? bw_req : 1u) : sys ? bw_req : 0u;
	p->static_band = bw != 0 && n > 0;
	for(size_t k = 0; k < n && p->static_band; k++) p->static_band = qlen[k] <= bw;
	p->generic = (bw == 0) || !bsa_align8_supported_bw(bw);
	p->max_bw = max_bw;
	p->pw = bsa_get_piecewise(par->gapo1, par->gape1, par->gapo2, par->gape2, (int)std::max(max_bw, 16u));
	if(p->sys) p->generic = false;
	if(p->generic && bsa_align8_gen_lds(max_bw, p->pw, bw == 0 ? 1u : 2u) > 160 * 1024){        // a whole-query band never moves: one row buffer
		c->err = "bandwidth too large for the device's generic kernel (two band rows must fit 160 KB of LDS)";
		delete p; return BSA_E_UNSUPPORTED;
	}
	p->rowb = bw ? 16u * bsa_tile_bytes(bw / 16u, p->pw) : 0u;
	p->qpad = max_bw + BSA_QPAD_TAIL;          // (the forward kernels' query window reads up to 48 bytes behind the band's last block: bsa_align8_x.hip, x_qwin)
	{
		// compact traceback where its preconditions hold (BSA_ALIGN8_LITERAL=1 keeps the row-record path)
		const char *le = bsa_env("BSA_ALIGN8_LITERAL");
		Align8Args t;
		memset(&t, 0, sizeof(t));
		t.bw = bw; t.mode = par->mode; t.gapo1 = par->gapo1; t.gape1 = par->gape1; t.gapo2 = par->gapo2; t.gape2 = par->gape2;
		int smax = -127, smin = 127;
		for(int i = 0; i < 16; i++){ smax = std::max(smax, (int)par->matrix[i]); smin = std::min(smin, (int)par->matrix[i]); }
		t.smax = smax; t.smin = smin;
		p->codes = !p->sys && !p->generic && !(le && le[0] == '1') && !(par->mode & BSA_MODE_ROWRECORDS) && bsa_align8_codes_supported(t, p->pw);
		// the compact traceback packs band offsets into 26 bits of its ring entries
		for(size_t k = 0; k < n && p->codes; k++) if(qlen[k] >= (1u << 26)) p->codes = false;
	}
	std::vector<uint32_t> order(n);
	for(size_t k = 0; k < n; k++) order[k] = (uint32_t)k;
	std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y){ return tlen[x] > tlen[y]; });
	std::vector<uint64_t> qpoff(n), tpoff(n), slot, slot_end;
	std::vector<size_t> need(n); std::vector<uint32_t> bwv(n, bw);
	size_t qacc = 0, tacc = 0;
	double cells = 0;
	auto bw_of = [&](size_t k) -> uint32_t { return bw ? bw : std::max(16u, (qlen[k] + 15u) / 16u * 16u); };                 // the width the kernels run
	auto bw_ref = [&](size_t k) -> uint32_t { return bw_req ? bw_req : std::max(16u, (qlen[k] + 15u) / 16u * 16u); };       // the reference's own width (cells)
	(void)widened;
	for(size_t k = 0; k < n; k++){
		qpoff[k] = qacc; qacc += ((size_t)qlen[k] + p->qpad + 15) & ~(size_t)15;
		tpoff[k] = tacc; tacc += ((size_t)tlen[k] + p->tpad + 15) & ~(size_t)15;
		if(qlen[k] && tlen[k]) cells += (double)tlen[k] * (double)bw_ref(k);
	}
	for(size_t pos = 0; pos < n; pos++)
		need[pos] = p->sys ? bsa_align8_sys_slot_bytes(qlen[order[pos]], tlen[order[pos]], p->pw)
			: p->codes ? bsa_code_slot_bytes(tlen[order[pos]], bw / 16u, p->pw) : bsa_slot_bytes(tlen[order[pos]], bw_of(order[pos]) / 16u, p->pw);
	p->cells = cells;
	p->stage_bytes = qacc + tacc;
	int rc = plan_chunks(p, order, need, bwv, slot, slot_end);
	for(Chunk &ch : p->chunks) if(ch.count) ch.max_tlen = tlen[order[ch.first]];          // (ordered by target length, longest first)
	if(rc == BSA_OK) rc = plan_common_alloc(p, qoff, qlen, toff, tlen, qpoff, tpoff, slot, slot_end, order, qacc, tacc);
	if(rc != BSA_OK){ plan_free(p); return rc; }
	*out = p;
	return BSA_OK;
}

extern "C" int bsa_align_run(bsa_align_plan_t *p, const uint8_t *d_seqs, bsa_result_t *d_out, uint32_t *d_cigar,
		size_t cigar_cap_words, uint64_t *d_cigar_off, uint32_t *d_status){
	if(!p || !d_out) return BSA_E_ARG;
	bsa_ctx *c = p->ctx;
	const uint32_t n = (uint32_t)p->n;
	const bool want_cig = d_cigar != nullptr && d_cigar_off != nullptr;
	int rc = run_prologue(p, want_cig, cigar_cap_words);
	if(rc != BSA_OK) return rc;
	hipStream_t st = c->stream;
	if(n == 0){
		if(d_cigar_off) HIPCHK(c, hipMemsetAsync(d_cigar_off, 0, sizeof(uint64_t), st));
		return BSA_OK;
	}
	if(!d_seqs) return BSA_E_ARG;
	uint32_t *status = d_status ? d_status : p->d_status_own;
	if(p->stage_bytes / std::max<size_t>(n, 1) >= 8192)         // long pairs: a block per pair
		hipLaunchKernelGGL(k_stage<256>, dim3(n), dim3(256), 0, st, d_seqs, p->d_qoff, p->d_qlen, p->d_toff, p->d_tlen,
			p->d_qpoff, p->d_tpoff, p->d_qst, p->d_tst, p->qpad, p->tpad, status, n);
	else
		hipLaunchKernelGGL(k_stage<64>, dim3((n + 3u) / 4u), dim3(256), 0, st, d_seqs, p->d_qoff, p->d_qlen, p->d_toff, p->d_tlen,
			p->d_qpoff, p->d_tpoff, p->d_qst, p->d_tst, p->qpad, p->tpad, status, n);
	HIPCHK(c, hipGetLastError());
	Align8Args a;
	memset(&a, 0, sizeof(a));
	a.qst = p->d_qst; a.tst = p->d_tst; a.qpoff = p->d_qpoff; a.tpoff = p->d_tpoff;
	a.qlen = p->d_qlen; a.tlen = p->d_tlen; a.order = p->d_order; a.slot_off = p->d_slot;
	a.status = status; a.bw = p->bw; a.rowb = p->rowb; a.mode = p->par.mode; a.ref_bw = p->ref_bw; a.static_band = p->static_band ? 1u : 0u; a.sys_chk = p->sys_chk ? 1u : 0u;
	a.gapo1 = p->par.gapo1; a.gape1 = p->par.gape1; a.gapo2 = p->par.gapo2; a.gape2 = p->par.gape2;
	int smax = -127, smin = 127;
	for(int i = 0; i < 16; i++){ smax = std::max(smax, (int)p->par.matrix[i]); smin = std::min(smin, (int)p->par.matrix[i]); a.matrix[i] = p->par.matrix[i]; }
	a.smax = smax; a.smin = smin;
	for(int t = 0; t < 4; t++){
		uint32_t w = 0;
		for(int q = 0; q < 4; q++) w |= (uint32_t)(uint8_t)p->par.matrix[q * 4 + t] << (8 * q);
		a.mrow[t] = w;
	}
	const int pw = p->pw;
	uint32_t *cnt = p->d_cnt_pos;
	const bool generic = p->generic, codes = p->codes, sys = p->sys; const uint32_t max_bw = p->max_bw;
	// forward kernel of the compact path: the exact-arithmetic one wherever its guard holds (BSA_ALIGN8_FWD=pk keeps the
	// saturating packed kernel: same code rows, the reference point of the tests)
	bool fwd_x = false;
	if(codes){
		const char *fe = bsa_env("BSA_ALIGN8_FWD");
		const bool force_pk = fe && fe[0] == 'p';
		fwd_x = (pw == 2) || (!force_pk && bsa_align8_x_supported(a, pw));       // (two-piece gaps: the only forward kernel of the compact path)
	}
	if(codes && fwd_x && bsa_align8_do2_supported(a, pw) && bsa_align8_trace_reads_do2(a, pw)) a.code_fmt = 1u;      // two-bit D / Od fields (bsa_common.h)
	c->fwd_name = (sys && p->sys_chk) ? "k_align8_fwd_sys<CHK> (whole-query band, systolic wavefront checking every pair against the int8 range, 4-bit traceback codes)" : sys ? "k_align8_fwd_sys (whole-query band, systolic wavefront, 4-bit traceback codes)" : (fwd_x && pw == 2) ? "k_align8_fwd_x2 (exact-arithmetic forward DP, two-piece gaps, 8-bit traceback codes)"
		: fwd_x ? "k_align8_fwd_x (exact-arithmetic forward DP, 4-bit traceback codes)" : codes ? "k_align8_fwd_pk<.,.,true> (packed forward DP, 4-bit traceback codes)"
		: generic ? "k_align8_fwd_gen (run-time bandwidth, row records)" : "k_align8_fwd_pk / k_align8_fwd (row records)";
	c->trace_name = sys ? "k_align8_trace_sys" : codes ? "" : "k_align8_backcal";
	bsa_last_trace_kernel = nullptr; bsa_last_fwd_kernel = nullptr;
	if(codes && fwd_x && !p->static_band){
		// the persistent form of the forward kernel hands band states from one row segment to the next through this buffer
		size_t need = 0;
		for(const Chunk &ch : p->chunks) need = std::max(need, bsa_align8_xq_bytes(p->bw, pw, ch.count));
		if(need > c->xq_bytes){
			if(c->xq){ HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipStreamSynchronize(c->aux_stream)); (void)hipFree(c->xq); c->xq = nullptr; c->xq_bytes = 0; }
			if(hipMalloc(&c->xq, need) == hipSuccess) c->xq_bytes = need; else { c->xq = nullptr; (void)hipGetLastError(); }      // (without it the launcher takes the plain kernels)
		}
		a.xq = (uint32_t*)c->xq; a.xq_bytes = c->xq_bytes;
	}
	auto fwd = [&](const Chunk &ch, uint8_t *half, hipStream_t s) -> int {
		Align8Args b = a; b.first = ch.first; b.count = ch.count; b.rows = half; b.max_tlen = ch.max_tlen;
		if(sys) HIPCHK(c, bsa_launch_align8_fwd_sys(b, pw, p->max_qlen, s));
		else if(codes && fwd_x) HIPCHK(c, bsa_launch_align8_fwd_x(b, pw, s));
		else if(codes) HIPCHK(c, bsa_launch_align8_fwd_codes(b, pw, s));
		else if(generic) HIPCHK(c, bsa_launch_align8_fwd_gen(b, pw, max_bw, s));
		else HIPCHK(c, bsa_launch_align8_fwd(b, pw, s));
		return BSA_OK;
	};
	auto trace = [&](const Chunk &ch, uint8_t *half, hipStream_t s) -> int {
		Align8Args b = a; b.first = ch.first; b.count = ch.count; b.rows = half;
		if(sys) HIPCHK(c, bsa_launch_align8_trace_sys(b, pw, d_out, cnt, p->d_slot_end, s));
		else if(codes) HIPCHK(c, bsa_launch_align8_trace_codes(b, pw, d_out, cnt, s));
		else HIPCHK(c, bsa_launch_align8_backcal(b, pw, d_out, cnt, s));
		return BSA_OK;
	};
	const int rc8 = run_pipeline(p, want_cig, d_cigar, cigar_cap_words, d_cigar_off, fwd, trace);
	if(codes && bsa_last_trace_kernel) c->trace_name = bsa_last_trace_kernel;
	if(codes && fwd_x && bsa_last_fwd_kernel) c->fwd_name = bsa_last_fwd_kernel;
	return rc8;
}


// A copy between PAGEABLE host memory and the device keeps its host thread busy staging (measured on the MI355X box: one thread moves about 30 GB/s up
// and 10 GB/s down), so a large one is cut into pieces that travel on threads and streams of their own; returns when all of it has arrived.
static hipError_t par_copy(int device, void *dst, const void *src, size_t bytes, hipMemcpyKind kind){
	unsigned T = 4;
	if(const char *e = bsa_env("BSA_COPY_THREADS")){ const int v = atoi(e); if(v >= 1 && v <= 16) T = (unsigned)v; }
	if(bytes < ((size_t)32 << 20) || T == 1){
		hipStream_t st = nullptr;
		hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
		if(e != hipSuccess) return e;
		e = hipMemcpyAsync(dst, src, bytes, kind, st);
		if(e == hipSuccess) e = hipStreamSynchronize(st);
		(void)hipStreamDestroy(st);
		return e;
	}
	const size_t piece = ((bytes + T - 1) / T + 4095) & ~(size_t)4095;
	std::vector<hipError_t> errs(T, hipSuccess);
	std::vector<std::thread> th;
	for(unsigned t = 0; t < T; t++){
		const size_t lo = (size_t)t * piece;
		if(lo >= bytes) break;
		const size_t len = std::min(piece, bytes - lo);
		th.emplace_back([=, &errs]{
			(void)hipSetDevice(device);
			hipStream_t st = nullptr;
			hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
			if(e == hipSuccess) e = hipMemcpyAsync((uint8_t*)dst + lo, (const uint8_t*)src + lo, len, kind, st);
			if(e == hipSuccess) e = hipStreamSynchronize(st);
			if(st) (void)hipStreamDestroy(st);
			errs[t] = e;
		});
	}
	for(std::thread &x : th) x.join();
	for(hipError_t e : errs) if(e != hipSuccess) return e;
	return hipSuccess;
}

// ---- a large host-pointer batch in TWO slices (round 6) --------------------------------------------------------------
// A caller with host buffers pays the upload in front of the kernels and the download behind them: C2 176-180 ms against 72 ms on resident
// inputs.  With the batch cut in two -- 50 000 pairs still fill the chip, smaller launches do not -- slice B's sequences
// travel while slice A's kernels run, and slice A's results go back while slice B's run: a helper thread feeds the uploads (a copy from
// pageable memory keeps its host thread busy), the calling thread plans both slices under the first upload, launches, and collects.
// Each slice is an ordinary plan + run on the context's stream; both read the ONE device copy of the blob at the caller's offsets and write
// their own results, offsets and CIGAR arena.  Returns BSA_OK with *done = true, an error, or *done = false when the batch is not for this
// path (small, a bandwidth that makes width classes, byte ranges of the slices that cannot be told apart): the caller goes on as before.
static int align_batch_sliced(bsa_ctx *c, const uint8_t *seqs, size_t seqs_bytes, const uint64_t *qoff, const uint32_t *qlen, const uint64_t *toff,
		const uint32_t *tlen, size_t n, const bsa_align_params_t *par, bsa_result_t *out, uint32_t *cigar, size_t cigar_cap_words, uint64_t *cigar_off,
		uint32_t *st, bool *done, bool *codes_out){
	*done = false;
	const char *se = bsa_env("BSA_BATCH_SLICES");
	const bool forced = se && se[0] == '2';
	if(se && se[0] == '1') return BSA_OK;
	const uint32_t bw_req = (par->bandwidth + 15u) / 16u * 16u;
	if(bw_req == 0 || !bsa_align8_supported_bw(bw_req) || (par->mode & BSA_MODE_ROWRECORDS)) return BSA_OK;
	if(!forced && (n < 80000 || seqs_bytes < ((size_t)512 << 20))) return BSA_OK;
	if(n < 2 || (cigar && cigar_off && cigar_cap_words > ((size_t)1 << 31))) return BSA_OK;
	// the cut: half of the bytes
	size_t tot = 0, h = 0;
	for(size_t k = 0; k < n; k++) tot += (size_t)qlen[k] + tlen[k];
	{ size_t acc = 0; while(h < n && acc < tot / 2){ acc += (size_t)qlen[h] + tlen[h]; h++; } }
	if(h == 0 || h >= n) return BSA_OK;
	// the byte intervals a slice reads, merged (gaps below 64 KB travel along); many scattered intervals: not for this path
	struct Iv { size_t lo, hi; };
	auto intervals = [&](size_t k0, size_t k1, std::vector<Iv> &v) -> bool {
		std::vector<Iv> raw; raw.reserve(2 * (k1 - k0));
		for(size_t k = k0; k < k1; k++){ if(qlen[k]) raw.push_back({(size_t)qoff[k], (size_t)qoff[k] + qlen[k]}); if(tlen[k]) raw.push_back({(size_t)toff[k], (size_t)toff[k] + tlen[k]}); }
		std::sort(raw.begin(), raw.end(), [](const Iv &a, const Iv &b){ return a.lo < b.lo; });
		for(const Iv &x : raw){
			if(!v.empty() && x.lo <= v.back().hi + ((size_t)64 << 10)) v.back().hi = std::max(v.back().hi, x.hi);
			else { if(v.size() >= 64) return false; v.push_back(x); }
		}
		return true;
	};
	std::vector<Iv> ivA, ivB;
	if(!intervals(0, h, ivA) || !intervals(h, n, ivB)) return BSA_OK;
	// what slice A uploaded need not travel again
	{
		std::vector<Iv> rest;
		for(Iv x : ivB){
			for(const Iv &a : ivA){
				if(a.hi <= x.lo || a.lo >= x.hi) continue;
				if(a.lo > x.lo) rest.push_back({x.lo, a.lo});
				x.lo = std::min(x.hi, a.hi);
			}
			if(x.lo < x.hi) rest.push_back(x);
		}
		ivB.swap(rest);
	}
	(void)hipSetDevice(c->device);
	const bool tmg = bsa_env("BSA_API_TIMING") != nullptr;
	const auto ts0 = std::chrono::steady_clock::now();
	auto since = [&](){ return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts0).count(); };
	double t_alloc = 0, t_plan = 0, t_join = 0, t_launch = 0, t_colA = 0, t_colB = 0;
	const bool want_cig = cigar && cigar_off;
	const size_t nA = h, nB = n - h;
	auto up = [](size_t b){ return (std::max<size_t>(b, 8) + 255) & ~(size_t)255; };
	const size_t o_seqs = 0, o_out = o_seqs + up(seqs_bytes), o_st = o_out + up(n * sizeof(bsa_result_t)), o_offA = o_st + up(n * 4), o_offB = o_offA + up((nA + 1) * 8),
		o_cigA = o_offB + up((nB + 1) * 8), o_cigB = o_cigA + (want_cig ? up(cigar_cap_words * 4) : 0), total = o_cigB + (want_cig ? up(cigar_cap_words * 4) : 0);
	uint8_t *pool = nullptr; bool kept = false;
	if(ctx_buf_get(c, 1, total, (void**)&pool, &kept) != hipSuccess){ (void)hipGetLastError(); return BSA_OK; }      // (not enough memory for two arenas: the plain path)
	hipStream_t ups = nullptr, dns = nullptr;
	hipEvent_t evA = nullptr, evB = nullptr, doneA = nullptr, doneB = nullptr;
	bsa_align_plan_t *pA = nullptr, *pB = nullptr;
	int rc = BSA_OK;
	auto cleanup = [&](){
		(void)hipStreamSynchronize(c->stream);
		if(ups){ (void)hipStreamSynchronize(ups); (void)hipStreamDestroy(ups); }
		if(dns){ (void)hipStreamSynchronize(dns); (void)hipStreamDestroy(dns); }
		for(hipEvent_t e : {evA, evB, doneA, doneB}) if(e) (void)hipEventDestroy(e);
		if(pA) bsa_align_plan_destroy(pA);
		if(pB) bsa_align_plan_destroy(pB);
		ctx_buf_put(c, 1, pool, kept);
	};
#define SL(call) do { if(rc == BSA_OK){ hipError_t _e = (call); if(_e != hipSuccess){ c->err = std::string(#call) + ": " + hipGetErrorString(_e); rc = BSA_E_HIP; } } } while(0)
	SL(hipStreamCreateWithFlags(&ups, hipStreamNonBlocking)); SL(hipStreamCreateWithFlags(&dns, hipStreamNonBlocking));
	SL(hipEventCreateWithFlags(&evA, hipEventDisableTiming)); SL(hipEventCreateWithFlags(&evB, hipEventDisableTiming));
	SL(hipEventCreateWithFlags(&doneA, hipEventDisableTiming)); SL(hipEventCreateWithFlags(&doneB, hipEventDisableTiming));
	if(rc != BSA_OK){ cleanup(); return rc; }
	t_alloc = since();
	uint8_t *d_seqs = pool + o_seqs;
	bsa_result_t *d_out = (bsa_result_t*)(pool + o_out); uint32_t *d_st = (uint32_t*)(pool + o_st);
	uint64_t *d_offA = want_cig ? (uint64_t*)(pool + o_offA) : nullptr, *d_offB = want_cig ? (uint64_t*)(pool + o_offB) : nullptr;
	uint32_t *d_cigA = want_cig ? (uint32_t*)(pool + o_cigA) : nullptr, *d_cigB = want_cig ? (uint32_t*)(pool + o_cigB) : nullptr;
	// the uploads on a thread of their own: A's intervals, event, B's, event
	hipError_t uperr = hipSuccess;
	std::atomic<int> recorded{0};                       // 1: evA is recorded (slice A's copies are in the upload stream), 2: evB as well; -1: failed
	std::thread upl([&]{
		(void)hipSetDevice(c->device);
		for(const Iv &x : ivA) if(uperr == hipSuccess) uperr = par_copy(c->device, d_seqs + x.lo, seqs + x.lo, x.hi - x.lo, hipMemcpyHostToDevice);
		if(uperr == hipSuccess) uperr = hipEventRecord(evA, ups);
		recorded.store(uperr == hipSuccess ? 1 : -1, std::memory_order_release);
		for(const Iv &x : ivB) if(uperr == hipSuccess) uperr = par_copy(c->device, d_seqs + x.lo, seqs + x.lo, x.hi - x.lo, hipMemcpyHostToDevice);
		if(uperr == hipSuccess) uperr = hipEventRecord(evB, ups);
		recorded.store(uperr == hipSuccess ? 2 : -1, std::memory_order_release);
	});
	auto wait_recorded = [&](int want){ for(;;){ const int r = recorded.load(std::memory_order_acquire); if(r < 0 || r >= want) return r; std::this_thread::yield(); } };
	rc = bsa_align_plan_create(c, qoff, qlen, toff, tlen, nA, par, &pA);
	t_plan = since();
	if(rc != BSA_OK){ upl.join(); cleanup(); return rc; }
	*codes_out = pA->codes || pA->sys;
	// slice A as soon as its plan stands and its bytes are on their way (a wait on an event that is not recorded yet would be no wait); slice B is
	// planned while A's kernels run and goes behind them on the same stream: B's kernels wait for B's bytes only
	if(wait_recorded(1) < 0) rc = BSA_E_HIP;
	SL(hipStreamWaitEvent(c->stream, evA, 0));
	if(rc == BSA_OK) rc = bsa_align_run(pA, d_seqs, d_out, d_cigA, cigar_cap_words, d_offA, d_st);
	SL(hipEventRecord(doneA, c->stream));
	if(rc == BSA_OK) rc = bsa_align_plan_create(c, qoff + h, qlen + h, toff + h, tlen + h, nB, par, &pB);
	if(rc != BSA_OK){ upl.join(); cleanup(); return rc; }
	if(wait_recorded(2) < 0) rc = BSA_E_HIP;
	t_join = since();
	SL(hipStreamWaitEvent(c->stream, evB, 0));
	if(rc == BSA_OK) rc = bsa_align_run(pB, d_seqs, d_out + nA, d_cigB, cigar_cap_words, d_offB, d_st + nA);
	SL(hipEventRecord(doneB, c->stream));
	upl.join();
	if(uperr != hipSuccess && rc == BSA_OK){ c->err = std::string("upload of the sequences: ") + hipGetErrorString(uperr); rc = BSA_E_HIP; }
	// slice A's results go back while slice B runs
	uint64_t totA = 0, totB = 0;
	auto collect = [&](hipEvent_t ev, size_t k0, size_t m, const uint64_t *d_off, const uint32_t *d_cig, uint64_t base, uint64_t *tot_out){
		SL(hipStreamWaitEvent(dns, ev, 0));
		SL(hipMemcpyAsync(out + k0, d_out + k0, m * sizeof(bsa_result_t), hipMemcpyDeviceToHost, dns));
		SL(hipMemcpyAsync(st + k0, d_st + k0, m * 4, hipMemcpyDeviceToHost, dns));
		if(want_cig){
			SL(hipMemcpyAsync(cigar_off + k0 + (k0 ? 1 : 0), d_off + (k0 ? 1 : 0), (m + (k0 ? 0 : 1)) * 8, hipMemcpyDeviceToHost, dns));      // (slice B's first offset is slice A's total)
			SL(hipStreamSynchronize(dns));
			if(rc != BSA_OK) return;
			const uint64_t t = cigar_off[k0 + m];
			*tot_out = t;
			if(base + t > cigar_cap_words){ c->err = "cigar arena too small"; rc = BSA_E_CIGAR_CAP; return; }
			SL(par_copy(c->device, cigar + base, d_cig, t * 4, hipMemcpyDeviceToHost));
			if(base) for(size_t k = k0 + 1; k <= k0 + m; k++) cigar_off[k] += base;
		}
		SL(hipStreamSynchronize(dns));
	};
	t_launch = since();
	collect(doneA, 0, nA, d_offA, d_cigA, 0, &totA);
	t_colA = since();
	if(rc == BSA_OK) collect(doneB, nA, nB, d_offB, d_cigB, totA, &totB);
	t_colB = since();
	if(tmg) fprintf(stderr, "[bsa_align_batch] %zu pairs in two slices of %zu and %zu (uploads of %zu and %zu intervals): ms since entry -- buffers %.1f, plan of slice A %.1f, slice A launched + slice B planned + uploads issued %.1f, both runs launched %.1f, slice A back %.1f, slice B back %.1f\n",
		n, nA, nB, ivA.size(), ivB.size(), t_alloc, t_plan, t_join, t_launch, t_colA, t_colB);
#undef SL
	cleanup();
	if(tmg) fprintf(stderr, "[bsa_align_batch] ... buffers, plans, streams released %.1f\n", since());
	if(rc == BSA_OK) *done = true;
	return rc;
}

extern "C" int bsa_align_batch(bsa_ctx_t *c, const uint8_t *seqs, size_t seqs_bytes,
		const uint64_t *qoff, const uint32_t *qlen, const uint64_t *toff, const uint32_t *tlen, size_t n,
		const bsa_align_params_t *par, bsa_result_t *out, uint32_t *cigar, size_t cigar_cap_words,
		uint64_t *cigar_off, uint32_t *status){
	if(!c || !out || !par) return BSA_E_ARG;
	if(n == 0){ if(cigar_off) cigar_off[0] = 0; return BSA_OK; }
	if(!seqs || !qoff || !qlen || !toff || !tlen) return BSA_E_ARG;
	for(size_t k = 0; k < n; k++)            // the staging kernel reads seqs + qoff[k] .. + qlen[k] unconditionally
		if(qoff[k] + qlen[k] > seqs_bytes || toff[k] + tlen[k] > seqs_bytes){ c->err = "sequence offsets outside the blob"; return BSA_E_ARG; }
	(void)hipSetDevice(c->device);
	// Whole-query bands (bandwidth 0, or one the register kernels do not have): a plan runs at ONE width, so a batch whose pairs
	// fall into different width classes of the widened dispatch (bsa_align_plan_create) -- or some of whose queries are too long
	// for it -- goes down as one sub-batch per class: a single long query must not send a million short ones to the
	// run-time-width kernel.  (Host-pointer entry only; with device pointers the caller groups its pairs.)
	{
		const uint32_t bw_req = (par->bandwidth + 15u) / 16u * 16u;
		if(n >= 2 && (bw_req == 0 || !bsa_align8_supported_bw(bw_req))){
			const uint32_t wide[3] = {align8_widened_bw(par, 64u), align8_widened_bw(par, 128u), align8_widened_bw(par, 256u)};
			auto klass = [&](size_t k) -> uint32_t {           // the width the pair would run at alone; 0 = run-time-width kernel
				const uint32_t cols = bw_req ? (qlen[k] <= bw_req ? bw_req : 0u) : (std::max(qlen[k], 1u) + 15u) / 16u * 16u;
				return cols == 0u || cols > 256u ? 0u : cols <= 64u ? wide[0] : cols <= 128u ? wide[1] : wide[2];
			};
			size_t cnt[4] = {0, 0, 0, 0};
			auto slot_of = [](uint32_t w) -> int { return w == 64u ? 1 : w == 128u ? 2 : w == 256u ? 3 : 0; };
			if(wide[0] | wide[1] | wide[2]) for(size_t k = 0; k < n; k++) cnt[slot_of(klass(k))]++;
			const int present = (cnt[0] != 0) + (cnt[1] != 0) + (cnt[2] != 0) + (cnt[3] != 0);
			if(present > 1){
				std::vector<std::vector<uint32_t>> pc(n);          // per pair CIGAR words
				size_t still_flagged = 0; long handed = 0;
				for(int cl = 0; cl < 4; cl++){
					if(!cnt[cl]) continue;
					const size_t m = cnt[cl];
					std::vector<size_t> idx; idx.reserve(m);
					for(size_t k = 0; k < n; k++) if(slot_of(klass(k)) == cl) idx.push_back(k);
					std::vector<uint64_t> sq(m), stt(m), soff(m + 1);
					std::vector<uint32_t> sql(m), stl(m), sst(m);
					std::vector<bsa_result_t> sout(m);
					size_t scap = 16;
					for(size_t j = 0; j < m; j++){ sq[j] = qoff[idx[j]]; stt[j] = toff[idx[j]]; sql[j] = qlen[idx[j]]; stl[j] = tlen[idx[j]]; scap += (size_t)sql[j] + stl[j] + 2; }
					std::vector<uint32_t> scig(cigar ? scap : 0);
					const int rcs = bsa_align_batch(c, seqs, seqs_bytes, sq.data(), sql.data(), stt.data(), stl.data(), m, par, sout.data(),
						cigar ? scig.data() : nullptr, scap, (cigar && cigar_off) ? soff.data() : nullptr, sst.data());
					if(rcs != BSA_OK) return rcs;
					handed += c->last_handover;
					for(size_t j = 0; j < m; j++){
						out[idx[j]] = sout[j];
						if(sst[j] & BSA_ST_TRACE) still_flagged++;
						if(status) status[idx[j]] = sst[j];
						if(cigar && cigar_off) pc[idx[j]].assign(scig.begin() + soff[j], scig.begin() + soff[j + 1]);
					}
				}
				if(cigar && cigar_off){
					uint64_t w = 0;
					for(size_t k = 0; k < n; k++){ cigar_off[k] = w; w += pc[k].size(); }
					cigar_off[n] = w;
					if(w > cigar_cap_words){ c->err = "cigar arena too small"; return BSA_E_CIGAR_CAP; }
					for(size_t k = 0; k < n; k++) if(!pc[k].empty()) memcpy(cigar + cigar_off[k], pc[k].data(), pc[k].size() * 4);
				}
				c->last_handover = handed;
				// the rule of the single-plan path below (finish()): without a status array an undecided pair is an error, never a silent zeroed record
				if(status == nullptr && still_flagged){ c->err = "pairs left undecided (BSA_ST_TRACE) and no status array to report them in"; return BSA_E_UNSUPPORTED; }
				return BSA_OK;
			}
		}
	}
	bsa_align_plan_t *p = nullptr;
	if(!(par->mode & BSA_MODE_ROWRECORDS)) c->last_handover = 0;
	const bool timing = bsa_env("BSA_API_TIMING") != nullptr;          // (stderr: where a host-pointer batch spends its wall time)
	const auto tm0 = std::chrono::steady_clock::now();
	std::vector<uint32_t> st_own(status ? 0 : n);      // (a path that reads the traceback off codes may hand a pair over: it needs the flags even when the caller does not)
	uint32_t *st = status ? status : st_own.data();
	auto tm1 = tm0;
	bool sliced = false, codes = false;
	int rc = align_batch_sliced(c, seqs, seqs_bytes, qoff, qlen, toff, tlen, n, par, out, cigar, cigar_cap_words, cigar_off, st, &sliced, &codes);
	if(rc != BSA_OK) return rc;
	if(!sliced){
	rc = batch_host(c, seqs, seqs_bytes, n, out, cigar, cigar_cap_words, cigar_off, st,
		[&]{ const int r = bsa_align_plan_create(c, qoff, qlen, toff, tlen, n, par, &p); tm1 = std::chrono::steady_clock::now(); return r; },
		[&](uint8_t *ds, bsa_result_t *dout, uint32_t *dc, uint64_t *doff, uint32_t *dst){ return bsa_align_run(p, ds, dout, dc, cigar_cap_words, doff, dst); });
	if(!p) return rc;
	codes = p->codes || p->sys;           // both read the traceback off codes and may hand a pair over
	if(timing){
		const auto tm2 = std::chrono::steady_clock::now();
		fprintf(stderr, "[bsa_align_batch] %zu pairs, mode %d, bandwidth %u: plan %.3f s (workspace %.1f GB, %zu chunks), staging + kernels + copies %.3f s, forward kernel %s\n", n, par->mode & 3, par->bandwidth,
			std::chrono::duration<double>(tm1 - tm0).count(), (double)c->ws_bytes / 1e9, p->chunks.size(), std::chrono::duration<double>(tm2 - tm1).count(), c->fwd_name.c_str());
	}
	bsa_align_plan_destroy(p);
	}
	if(rc != BSA_OK || !codes) return rc;
	for(size_t k = 0; k < n; k++) if(st[k] & BSA_ST_DEVICE){ c->err = "forward pass: a row-segment hand-over timed out (BSA_ST_DEVICE)"; return BSA_E_HIP; }
	// ---- hand-over: pairs the compact traceback could not decide go through the literal kernels, so that a flag that
	// survives means what it means for the reference (its own traceback does not terminate there)
	std::vector<size_t> idx;
	const char *dbg = bsa_env("BSA_DEBUG_HANDOVER");          // test hook: treat every N-th pair as undecided
	const long every = dbg ? atol(dbg) : 0;
	// (only what the literal kernels can take: a pair whose band -- with bandwidth 0, its whole query -- does not fit the run-time-width
	// kernel's LDS keeps its flag instead of failing the batch after all the work is done; the systolic kernel accepts longer queries)
	auto literal_can_take = [&](size_t k) -> bool {
		const uint32_t bw_req = (par->bandwidth + 15u) / 16u * 16u;
		const uint32_t width = bw_req ? bw_req : std::max(16u, (qlen[k] + 15u) / 16u * 16u);
		if(bsa_align8_supported_bw(width)) return true;
		const int pwk = bsa_get_piecewise(par->gapo1, par->gape1, par->gapo2, par->gape2, (int)width);
		return bsa_align8_gen_lds(width, pwk, bw_req ? 2u : 1u) <= 160 * 1024;
	};
	size_t left_flagged = 0;                           // undecided pairs the literal kernels cannot take: they keep BSA_ST_TRACE and a zeroed result
	for(size_t k = 0; k < n; k++){
		const bool want = (st[k] & BSA_ST_TRACE) || (every > 0 && k % (size_t)every == 0 && st[k] == 0);
		if(!want) continue;
		if(literal_can_take(k)) idx.push_back(k);
		else if(st[k] & BSA_ST_TRACE) left_flagged ++;
	}
	// a caller that passed no status array cannot see a flag: undecided pairs are an error for it, never a silent zeroed result
	auto finish = [&](size_t still_flagged) -> int {
		if(status == nullptr && still_flagged){ c->err = "pairs left undecided (BSA_ST_TRACE) and no status array to report them in"; return BSA_E_UNSUPPORTED; }
		return BSA_OK;
	};
	if(idx.empty()) return finish(left_flagged);
	const size_t m = idx.size();
	c->last_handover = (long)m;
	if(timing) fprintf(stderr, "[bsa_align_batch] %zu pairs handed over to the literal kernels\n", m);
	std::vector<uint64_t> sq(m), stt(m), soff(m + 1);
	std::vector<uint32_t> sql(m), stl(m), sst(m);
	std::vector<bsa_result_t> sout(m);
	size_t scap = 16;
	for(size_t k = 0; k < m; k++){ sq[k] = qoff[idx[k]]; stt[k] = toff[idx[k]]; sql[k] = qlen[idx[k]]; stl[k] = tlen[idx[k]]; scap += (size_t)sql[k] + stl[k] + 2; }
	std::vector<uint32_t> scig(cigar ? scap : 0);
	bsa_align_params_t lp = *par;
	lp.mode |= BSA_MODE_ROWRECORDS;
	const std::string keep_fwd = c->fwd_name, keep_trace = c->trace_name;         // the batch's kernels stay the ones reported, not the re-run's
	rc = bsa_align_batch(c, seqs, seqs_bytes, sq.data(), sql.data(), stt.data(), stl.data(), m, &lp, sout.data(),
		cigar ? scig.data() : nullptr, scap, (cigar && cigar_off) ? soff.data() : nullptr, sst.data());
	c->fwd_name = keep_fwd; c->trace_name = keep_trace;
	// (literal_can_take filtered what the literal kernels decline, so BSA_E_UNSUPPORTED here is a real error of the re-run, not a pair
	// to leave flagged: it is returned like any other)
	if(rc != BSA_OK) return rc;
	for(size_t k = 0; k < m; k++){ out[idx[k]] = sout[k]; st[idx[k]] = sst[k]; if(sst[k] & BSA_ST_TRACE) left_flagged ++; }
	if(cigar && cigar_off){
		// splice the re-run pairs' CIGARs into the arena (the compact pass left them empty or, in the debug hook, filled)
		std::vector<uint32_t> merged;
		std::vector<uint64_t> noff(n + 1);
		merged.reserve((size_t)cigar_off[n] + (size_t)soff[m]);
		size_t j = 0;
		for(size_t k = 0; k < n; k++){
			noff[k] = merged.size();
			if(j < m && idx[j] == k){ merged.insert(merged.end(), scig.begin() + soff[j], scig.begin() + soff[j + 1]); j++; }
			else merged.insert(merged.end(), cigar + cigar_off[k], cigar + cigar_off[k + 1]);
		}
		noff[n] = merged.size();
		memcpy(cigar_off, noff.data(), (n + 1) * sizeof(uint64_t));
		if(merged.size() > cigar_cap_words){ c->err = "cigar arena too small"; return BSA_E_CIGAR_CAP; }
		if(!merged.empty()) memcpy(cigar, merged.data(), merged.size() * 4);
	}
	return finish(left_flagged);
}

// debug / test hook: copy the stored row records of `pair` to host.  Only meaningful right after a single-chunk run.
extern "C" int bsa_align_debug_rows(bsa_align_plan_t *p, uint32_t pair, uint8_t *host, size_t bytes, uint32_t *rowb_out){
	if(!p || !host) return BSA_E_ARG;
	bsa_ctx *c = p->ctx;
	(void)hipSetDevice(c->device);
	HIPCHK(c, hipStreamSynchronize(c->stream));
	HIPCHK(c, hipStreamSynchronize(c->aux_stream));
	std::vector<uint32_t> order(p->n);
	std::vector<uint64_t> slot(p->n);
	HIPCHK(c, hipMemcpy(order.data(), p->d_order, p->n * 4, hipMemcpyDeviceToHost));
	HIPCHK(c, hipMemcpy(slot.data(), p->d_slot, p->n * 8, hipMemcpyDeviceToHost));
	for(size_t pos = 0; pos < p->n; pos++){
		if(order[pos] == pair){
			if(bytes) HIPCHK(c, hipMemcpy(host, c->ws + slot[pos], bytes, hipMemcpyDeviceToHost));   // begs array, then the row records
			if(rowb_out) *rowb_out = p->rowb;
			return BSA_OK;
		}
	}
	return BSA_E_ARG;
}

// kernel timing for launches made outside this file (bsa_rows.hip): one (start, stop) event pair around the launch
extern "C" int bsa_ctx_time_begin_internal(bsa_ctx_t *c, double cells, void **stop_event){
	if(!c || !stop_event) return BSA_E_ARG;
	c->ev_used = 0; c->last_cells = cells;
	hipEvent_t e0, e1;
	int rc = ctx_event_pair(c, &e0, &e1);
	if(rc != BSA_OK) return rc;
	HIPCHK(c, hipEventRecord(e0, c->stream));
	*stop_event = (void*)e1;
	return BSA_OK;
}
extern "C" int bsa_ctx_time_end_internal(bsa_ctx_t *c, void *stop_event){
	if(!c || !stop_event) return BSA_E_ARG;
	HIPCHK(c, hipEventRecord((hipEvent_t)stop_event, c->stream));
	return BSA_OK;
}

// a device buffer of at least `bytes` that stays the context's (stream-ordered users only: the previous user's kernels are on the same stream)
extern "C" int bsa_ctx_scratch_internal(bsa_ctx_t *c, int slot, size_t bytes, void **out){
	if(!c || !out || slot < 0 || slot > 2) return BSA_E_ARG;
	(void)hipSetDevice(c->device);
	if(c->scratch_bytes[slot] < bytes){
		if(c->scratch[slot]){ HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->scratch[slot]); c->scratch[slot] = nullptr; c->scratch_bytes[slot] = 0; }
		const size_t want = bytes + bytes / 4 + ((size_t)1 << 20);
		if(hipMalloc(&c->scratch[slot], want) != hipSuccess){ c->scratch[slot] = nullptr; (void)hipGetLastError(); c->err = "scratch allocation failed"; return BSA_E_NOMEM; }
		c->scratch_bytes[slot] = want;
	}
	*out = c->scratch[slot];
	return BSA_OK;
}

extern "C" int bsa_ctx_get_stream_internal(bsa_ctx_t *c, hipStream_t *st){
	if(!c || !st) return BSA_E_ARG;
	(void)hipSetDevice(c->device);
	*st = c->stream;
	return BSA_OK;
}

// ------------------------------------------------------------------------------------------------
// 2-bit edit alignment plan (striped_seqedit_pairwise, bsalign.h:1046)
// ------------------------------------------------------------------------------------------------
struct bsa_edit_plan : PlanBase {
	bsa_edit_params_t par;
	uint32_t pad_rows = 12;         // (row format 1 rounds a slot's rows up to a tile of eight: at least four spare rows stay behind them)
	uint64_t *d_qboff = nullptr, *d_qbits = nullptr;
	uint32_t *d_qwords = nullptr;
	int32_t *d_sbeg = nullptr;
};

extern "C" void bsa_edit_plan_destroy(bsa_edit_plan_t *p){ plan_free(p); }
extern "C" double bsa_edit_plan_cells(const bsa_edit_plan_t *p){ return p ? p->cells : 0.0; }

extern "C" int bsa_edit_plan_create(bsa_ctx_t *c, const uint64_t *qoff, const uint32_t *qlen,
		const uint64_t *toff, const uint32_t *tlen, size_t n, const bsa_edit_params_t *par, bsa_edit_plan_t **out){
	if(!c || !out || !par || (n && (!qoff || !qlen || !toff || !tlen))) return BSA_E_ARG;
	*out = nullptr;
	if(n > 0xFFFFFFF0ull){ c->err = "too many pairs"; return BSA_E_ARG; }
	const int type = par->mode & 3;
	if(type != BSA_MODE_GLOBAL && type != BSA_MODE_OVERLAP && type != BSA_MODE_EXTEND){ c->err = "bad mode"; return BSA_E_ARG; }
	(void)hipSetDevice(c->device);
	bsa_edit_plan *p = new bsa_edit_plan();
	p->ctx = c; p->n = n; p->par = *par;
	const auto tp0 = std::chrono::steady_clock::now();
	std::vector<uint32_t> bwk(n), order(n), qwords(n);
	double cells = 0;
	for(size_t k = 0; k < n; k++){
		bwk[k] = (qlen[k] && tlen[k]) ? bsa_edit_bw_eff(qlen[k], tlen[k], type, par->bandwidth) : 64u;
		if(!bsa_edit_supported_bw(bwk[k])){
			c->err = "effective edit bandwidth must be a multiple of 64";
			plan_free(p); return BSA_E_UNSUPPORTED;
		}
		if(qlen[k] && tlen[k]) cells += (double)tlen[k] * (double)bwk[k];
		order[k] = (uint32_t)k;
	}
	p->cells = cells;
	// launch class of every pair.  A static band (the band is the whole rounded query: overlap / extend / bandwidth 0)
	// can also run on the wave-per-pair kernel, which is much the faster one while its waves all fit the chip: the
	// widest sparsely populated register classes are moved there, together at most 16384 pairs.
	std::vector<uint32_t> cls(n);
	{
		size_t cnt[17] = {0};
		auto is_static = [&](size_t k){ return bwk[k] == (qlen[k] + 63u) / 64u * 64u; };
		for(size_t k = 0; k < n; k++) if(bwk[k] <= BSA_EDIT_REG_BW && is_static(k) && qlen[k] && tlen[k]) cnt[bwk[k] / 64u] ++;
		bool moved[17] = {false};
		size_t total = 0;
		for(int w = 16; w >= 2; w--) if(cnt[w] && cnt[w] <= 8192 && total + cnt[w] <= 16384){ moved[w] = true; total += cnt[w]; }
		if(bsa_env("BSA_EDIT_NO_MERGE")) for(int w = 0; w < 17; w++) moved[w] = false;
		for(size_t k = 0; k < n; k++){
			cls[k] = bsa_edit_class(bwk[k]);
			if(bwk[k] <= BSA_EDIT_REG_BW && moved[bwk[k] / 64u] && is_static(k) && qlen[k] && tlen[k]) cls[k] = bsa_edit_class(64u * 64u);
		}
	}
	// processing order: by class, then band, then longest target first (ties keep the caller's order)
	{
		std::vector<uint64_t> key(n);
		for(size_t k = 0; k < n; k++){
			const uint32_t rank = cls[k] <= BSA_EDIT_REG_BW ? cls[k] / 64u : 17u + ((cls[k] & 0xFFu) == 0u ? 0u : (cls[k] & 0xFFu) == 1u ? 1u : (cls[k] & 0xFFu) == 2u ? 2u : (cls[k] & 0xFFu) == 4u ? 3u : 4u);
			key[k] = (uint64_t)rank << 56 | (uint64_t)(bwk[k] / 64u) << 32 | (uint64_t)(0xFFFFFFFFu - tlen[k]);
		}
		radix_sort_order(key, order);
	}
	const auto tp1 = std::chrono::steady_clock::now();
	std::vector<uint64_t> qpoff(n), tpoff(n), qboff(n), slot, slot_end;
	std::vector<size_t> need(n); std::vector<uint32_t> bwv(n);
	size_t qacc = 0, tacc = 0, bacc = 0;
	for(size_t k = 0; k < n; k++){
		qpoff[k] = qacc; qacc += ((size_t)qlen[k] + 16 + 15) & ~(size_t)15;
		tpoff[k] = tacc; tacc += ((size_t)tlen[k] + 16 + 15) & ~(size_t)15;
		qwords[k] = (qlen[k] + bwk[k]) / 64u + 4u;
		qboff[k] = bacc; bacc += 2 * (size_t)qwords[k];
	}
	for(size_t pos = 0; pos < n; pos++){
		const uint32_t k = order[pos];
		bwv[pos] = cls[k];
		need[pos] = ((size_t)tlen[k] + 1 + p->pad_rows) * (size_t)(bwk[k] / 64u) * 16;
	}
	int rc = plan_chunks(p, order, need, bwv, slot, slot_end, true);
	const auto tp2 = std::chrono::steady_clock::now();
	if(rc == BSA_OK) rc = plan_common_alloc(p, qoff, qlen, toff, tlen, qpoff, tpoff, slot, slot_end, order, qacc, tacc);
	if(rc == BSA_OK) rc = dev_upload(c, &p->d_qboff, qboff);
	if(rc == BSA_OK) rc = dev_upload(c, &p->d_qwords, qwords);
	if(rc == BSA_OK) rc = dev_alloc(c, &p->d_qbits, bacc);
	if(rc == BSA_OK) rc = dev_alloc(c, &p->d_sbeg, 3 * n);      // sbeg | smin | ry
	p->extra = { p->d_qboff, p->d_qwords, p->d_qbits, p->d_sbeg };
	if(rc != BSA_OK){ plan_free(p); return rc; }
	if(bsa_env("BSA_BATCH_TIMING")){
		auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b){ return std::chrono::duration<double, std::milli>(b - a).count(); };
		fprintf(stderr, "[bsa_edit_plan] %zu pairs: classes + order %.1f ms, layout + chunks %.1f ms, device metadata %.1f ms\n", n, ms(tp0, tp1), ms(tp1, tp2), ms(tp2, std::chrono::steady_clock::now()));
	}
	*out = p;
	return BSA_OK;
}

extern "C" int bsa_edit_run(bsa_edit_plan_t *p, const uint8_t *d_seqs, bsa_result_t *d_out, uint32_t *d_cigar,
		size_t cigar_cap_words, uint64_t *d_cigar_off, uint32_t *d_status){
	if(!p || !d_out) return BSA_E_ARG;
	bsa_ctx *c = p->ctx;
	const uint32_t n = (uint32_t)p->n;
	const bool want_cig = d_cigar != nullptr && d_cigar_off != nullptr;
	int rc = run_prologue(p, want_cig, cigar_cap_words);
	if(rc != BSA_OK) return rc;
	hipStream_t st = c->stream;
	if(n == 0){
		if(d_cigar_off) HIPCHK(c, hipMemsetAsync(d_cigar_off, 0, sizeof(uint64_t), st));
		return BSA_OK;
	}
	if(!d_seqs) return BSA_E_ARG;
	uint32_t *status = d_status ? d_status : p->d_status_own;
	HIPCHK(c, bsa_launch_edit_stage(d_seqs, p->d_qoff, p->d_qlen, p->d_toff, p->d_tlen, p->d_qpoff, p->d_tpoff, p->d_qboff, p->d_qwords,
		p->d_qst, p->d_tst, p->d_qbits, status, n, st));
	EditArgs a;
	memset(&a, 0, sizeof(a));
	a.qst = p->d_qst; a.tst = p->d_tst; a.qpoff = p->d_qpoff; a.tpoff = p->d_tpoff;
	a.qbits = p->d_qbits; a.qboff = p->d_qboff; a.qwords = p->d_qwords;
	a.qlen = p->d_qlen; a.tlen = p->d_tlen; a.order = p->d_order; a.slot_off = p->d_slot;
	a.status = status; a.fwd_sbeg = p->d_sbeg; a.fwd_smin = p->d_sbeg + n; a.fwd_ry = p->d_sbeg + 2 * (size_t)n; a.pad_rows = p->pad_rows; a.mode = p->par.mode; a.bandwidth = p->par.bandwidth;
	uint32_t *cnt = p->d_cnt_pos;
	auto fwd = [&](const Chunk &ch, uint8_t *half, hipStream_t s) -> int {
		for(uint32_t x = ch.sub0; x < ch.sub0 + ch.nsub; x++){      // one forward launch per class
			const Sub &sb = p->subs[x];
			EditArgs b = a; b.first = sb.first; b.count = sb.count; b.rows = half;
			b.bw = sb.bw <= BSA_EDIT_REG_BW ? sb.bw : 0u; b.wide = sb.bw <= BSA_EDIT_REG_BW ? 0u : (sb.bw & 0xFFu);
			b.row_fmt = (ch.nsub == 1u && sb.count == ch.count && ch.bw == sb.bw && bsa_edit_tiled_ok(b.bw, ch.count, b.mode)) ? 1u : 0u;
			HIPCHK(c, bsa_launch_edit_fwd(b, s));
		}
		return BSA_OK;
	};
	auto trace = [&](const Chunk &ch, uint8_t *half, hipStream_t s) -> int {
		EditArgs b = a; b.first = ch.first; b.count = ch.count; b.rows = half;
		b.bw = ch.bw <= BSA_EDIT_REG_BW ? ch.bw : 0u;               // several classes or wide bands: every pair works out its own
		b.row_fmt = (ch.nsub == 1u && p->subs[ch.sub0].count == ch.count && p->subs[ch.sub0].bw == ch.bw && bsa_edit_tiled_ok(b.bw, ch.count, b.mode)) ? 1u : 0u;      // (the same test as the forward launch of this chunk)
		HIPCHK(c, bsa_launch_edit_trace(b, d_out, cnt, s));
		return BSA_OK;
	};
	bsa_last_fwd_kernel = bsa_last_trace_kernel = nullptr;
	const int rce = run_pipeline(p, want_cig, d_cigar, cigar_cap_words, d_cigar_off, fwd, trace);
	c->fwd_name = bsa_last_fwd_kernel ? bsa_last_fwd_kernel : "k_edit_fwd*";         // (the last launch class of the batch)
	c->trace_name = bsa_last_trace_kernel ? bsa_last_trace_kernel : "k_edit_trace";
	return rce;
}

extern "C" int bsa_edit_batch(bsa_ctx_t *c, const uint8_t *seqs, size_t seqs_bytes,
		const uint64_t *qoff, const uint32_t *qlen, const uint64_t *toff, const uint32_t *tlen, size_t n,
		const bsa_edit_params_t *par, bsa_result_t *out, uint32_t *cigar, size_t cigar_cap_words,
		uint64_t *cigar_off, uint32_t *status){
	if(!c || !out || !par) return BSA_E_ARG;
	if(n == 0){ if(cigar_off) cigar_off[0] = 0; return BSA_OK; }
	if(!seqs || !qoff || !qlen || !toff || !tlen) return BSA_E_ARG;
	for(size_t k = 0; k < n; k++)
		if(qoff[k] + qlen[k] > seqs_bytes || toff[k] + tlen[k] > seqs_bytes){ c->err = "sequence offsets outside the blob"; return BSA_E_ARG; }
	(void)hipSetDevice(c->device);
	bsa_edit_plan_t *p = nullptr;
	const bool timing = bsa_env("BSA_BATCH_TIMING") != nullptr;        // host-side phase times on stderr
	const auto t0 = std::chrono::steady_clock::now();
	auto t1 = t0;
	int rc = batch_host(c, seqs, seqs_bytes, n, out, cigar, cigar_cap_words, cigar_off, status,
		[&]{ const int r = bsa_edit_plan_create(c, qoff, qlen, toff, tlen, n, par, &p); t1 = std::chrono::steady_clock::now(); return r; },
		[&](uint8_t *ds, bsa_result_t *dout, uint32_t *dc, uint64_t *doff, uint32_t *dst){ return bsa_edit_run(p, ds, dout, dc, cigar_cap_words, doff, dst); });
	const auto t2 = std::chrono::steady_clock::now();
	if(!p) return rc;
	const size_t nch = p->chunks.size(), nsub = p->subs.size();
	bsa_edit_plan_destroy(p);
	if(timing) fprintf(stderr, "[bsa_edit_batch] %zu pairs: plan %.1f ms (%zu chunks, %zu forward launches), copy + run + copy %.1f ms, destroy %.1f ms\n", n,
		std::chrono::duration<double, std::milli>(t1 - t0).count(), nch, nsub, std::chrono::duration<double, std::milli>(t2 - t1).count(),
		std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count());
	return rc;
}

// ---- anti-diagonal u8 DP of the MSA refinement (bsa_diagdp.hip), host pointers
// fill + traceback, nothing but the steps comes back (include/bsalign_hip.h)
extern "C" int bsa_diagdp_walk_batch(bsa_ctx_t *c, const uint8_t *planes, size_t planes_bytes, const bsa_diagdp_prob_t *probs_in, size_t n,
		bsa_diagdp_walk_t *walks, uint32_t *steps, size_t steps_cap_words){
	if(!c) return BSA_E_ARG;
	if(n == 0) return BSA_OK;
	if(!planes || !probs_in || !walks || !steps || n > 0x7fffffffu){ c->err = "bsa_diagdp_walk_batch: null argument"; return BSA_E_ARG; }
	(void)hipSetDevice(c->device);
	const uint32_t W = probs_in[0].W;
	if(!(W == 1 || W == 2 || W == 4)){ c->err = "bsa_diagdp_walk_batch: W must be 1, 2 or 4"; return BSA_E_ARG; }
	const uint64_t pad = 8ull * W, rowlen = 16ull * W + 2;
	std::vector<bsa_diagdp_prob_t> probs(probs_in, probs_in + n);
	std::vector<uint64_t> toff(n), woff(n);
	uint64_t tacc = 0, macc = 0, wacc = 0; uint32_t max_len = 0;
	for(size_t k = 0; k < n; k++){
		bsa_diagdp_prob_t &p = probs[k];
		if(p.W != W || p.mbeg > p.mend || p.mend > p.mlen){ c->err = "bsa_diagdp_walk_batch: bad problem"; return BSA_E_ARG; }
		const uint64_t offs[10] = {p.seq0, p.seq1, p.mats0[0], p.mats0[1], p.mats0[2], p.mats0[3], p.mats1[0], p.mats1[1], p.mats1[2], p.mats1[3]};
		for(uint64_t o : offs) if(o < pad || o + p.mlen + pad > planes_bytes){ c->err = "bsa_diagdp_walk_batch: plane outside the blob (8 W bytes of padding on both sides)"; return BSA_E_ARG; }
		// the device's own layout of the difference planes: only the rows 2 mbeg .. 2 mend - 1 exist, row r at (r - 2 mbeg) * rowlen
		const uint64_t rows = 2ull * (p.mend - p.mbeg) + 1, ms = (rows * rowlen + 15) & ~15ull;
		p.out0 = macc - 2ull * p.mbeg * rowlen; p.out1 = macc + ms - 2ull * p.mbeg * rowlen;        // (row 0's address; wraps below the buffer, never dereferenced there)
		macc += 2 * ms;
		toff[k] = tacc; tacc += 2ull * (p.mlen + 16ull * W);
		woff[k] = wacc; wacc += bsa_diagdp_walk_words(p.mbeg, p.mend);
		max_len = std::max<uint32_t>(max_len, p.mlen + 16u * W);
	}
	if(wacc > steps_cap_words){ c->err = "bsa_diagdp_walk_batch: steps buffer too small"; return BSA_E_ARG; }
	int rc = BSA_OK;
	auto fail = [&](const char *what, hipError_t e){ c->err = std::string(what) + ": " + hipGetErrorString(e); rc = (e == hipErrorOutOfMemory) ? BSA_E_NOMEM : BSA_E_HIP; };
	hipError_t e;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	do {
		const size_t a256 = 255;
		auto up = [&](size_t b){ return (b + a256) & ~a256; };
		const size_t o_planes = 0, o_probs = up(planes_bytes), o_T = o_probs + up(n * sizeof(bsa_diagdp_prob_t)), o_toff = o_T + up(tacc * 4), o_woff = o_toff + up(n * 8),
			o_walks = o_woff + up(n * 8), o_steps = o_walks + up(n * sizeof(bsa_diagdp_walk_t)), o_matrix = o_steps + up(wacc * 4 + 16) + 4096, total = o_matrix + macc + 4096;
		void *ws = nullptr;
		if((rc = bsa_ctx_scratch_internal(c, 1, total, &ws)) != BSA_OK) break;
		uint8_t *b = (uint8_t*)ws;
#define WCHK(x) if((e = (x)) != hipSuccess){ fail(#x, e); break; }
		WCHK(hipMemcpyAsync(b + o_planes, planes, planes_bytes, hipMemcpyHostToDevice, c->stream));
		WCHK(hipMemcpyAsync(b + o_probs, probs.data(), n * sizeof(bsa_diagdp_prob_t), hipMemcpyHostToDevice, c->stream));
		WCHK(hipMemcpyAsync(b + o_toff, toff.data(), n * 8, hipMemcpyHostToDevice, c->stream));
		WCHK(hipMemcpyAsync(b + o_woff, woff.data(), n * 8, hipMemcpyHostToDevice, c->stream));
		(void)hipEventCreate(&ev0); (void)hipEventCreate(&ev1);
		(void)hipEventRecord(ev0, c->stream);
		WCHK(bsa_launch_diagdp(b + o_planes, (const bsa_diagdp_prob_t*)(b + o_probs), (uint32_t*)(b + o_T), (const uint64_t*)(b + o_toff), b + o_matrix, (uint32_t)n, W, max_len, c->stream));
		WCHK(bsa_launch_diagdp_walk(b + o_planes, (const bsa_diagdp_prob_t*)(b + o_probs), (const uint32_t*)(b + o_T), (const uint64_t*)(b + o_toff), b + o_matrix, (uint32_t)n,
			(bsa_diagdp_walk_t*)(b + o_walks), (uint32_t*)(b + o_steps), (const uint64_t*)(b + o_woff), c->stream));
		(void)hipEventRecord(ev1, c->stream);
		WCHK(hipMemcpyAsync(walks, b + o_walks, n * sizeof(bsa_diagdp_walk_t), hipMemcpyDeviceToHost, c->stream));
		WCHK(hipMemcpyAsync(steps, b + o_steps, wacc * 4, hipMemcpyDeviceToHost, c->stream));
		WCHK(hipStreamSynchronize(c->stream));
#undef WCHK
		float ms = 0; if(hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) c->diagdp_ms = ms;
	} while(0);
	if(ev0) (void)hipEventDestroy(ev0);
	if(ev1) (void)hipEventDestroy(ev1);
	return rc;
}

extern "C" double bsa_diagdp_last_ms(bsa_ctx_t *c){ return c ? c->diagdp_ms : 0.0; }
extern "C" int bsa_diagdp_batch(bsa_ctx_t *c, const uint8_t *planes, size_t planes_bytes, const bsa_diagdp_prob_t *probs, size_t n,
		uint8_t *matrix, size_t matrix_bytes){
	if(!c) return BSA_E_ARG;
	if(n == 0) return BSA_OK;
	if(!planes || !probs || !matrix || n > 0x7fffffffu){ c->err = "bsa_diagdp_batch: null argument"; return BSA_E_ARG; }
	(void)hipSetDevice(c->device);
	const uint32_t W = probs[0].W;
	if(!(W == 1 || W == 2 || W == 4)){ c->err = "bsa_diagdp_batch: W must be 1, 2 or 4"; return BSA_E_ARG; }
	const uint64_t pad = 8ull * W, rowlen = 16ull * W + 2;
	std::vector<uint64_t> toff(n);
	uint64_t tacc = 0; uint32_t max_len = 0;
	for(size_t k = 0; k < n; k++){
		const bsa_diagdp_prob_t &p = probs[k];
		if(p.W != W || p.mbeg > p.mend || p.mend > p.mlen){ c->err = "bsa_diagdp_batch: bad problem"; return BSA_E_ARG; }
		const uint64_t offs[10] = {p.seq0, p.seq1, p.mats0[0], p.mats0[1], p.mats0[2], p.mats0[3], p.mats1[0], p.mats1[1], p.mats1[2], p.mats1[3]};
		for(uint64_t o : offs) if(o < pad || o + p.mlen + pad > planes_bytes){ c->err = "bsa_diagdp_batch: plane outside the blob (8 W bytes of padding on both sides)"; return BSA_E_ARG; }
		const uint64_t rows = 2ull * p.mlen + 1;
		if(p.out0 + rows * rowlen > matrix_bytes || p.out1 + rows * rowlen > matrix_bytes){ c->err = "bsa_diagdp_batch: matrix plane outside the buffer"; return BSA_E_ARG; }
		toff[k] = tacc; tacc += 2ull * (p.mlen + 16ull * W);
		max_len = std::max<uint32_t>(max_len, p.mlen + 16u * W);
	}
	uint8_t *d_planes = nullptr, *d_matrix = nullptr; bsa_diagdp_prob_t *d_probs = nullptr; uint32_t *d_T = nullptr; uint64_t *d_toff = nullptr;
	int rc = BSA_OK;
	auto fail = [&](const char *what, hipError_t e){ c->err = std::string(what) + ": " + hipGetErrorString(e); rc = (e == hipErrorOutOfMemory) ? BSA_E_NOMEM : BSA_E_HIP; };
	hipError_t e;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	do {
		// one buffer kept by the context between windows (an end_bspoa makes three calls per window, a polisher thousands of windows)
		const size_t a256 = 255;
		const size_t o_planes = 0, o_matrix = (o_planes + planes_bytes + a256) & ~a256, o_probs = (o_matrix + matrix_bytes + a256) & ~a256,
			o_T = (o_probs + n * sizeof(bsa_diagdp_prob_t) + a256) & ~a256, o_toff = (o_T + tacc * sizeof(uint32_t) + a256) & ~a256, total = o_toff + n * sizeof(uint64_t) + 256;
		void *ws = nullptr;
		if((rc = bsa_ctx_scratch_internal(c, 1, total, &ws)) != BSA_OK) break;
		d_planes = (uint8_t*)ws + o_planes; d_matrix = (uint8_t*)ws + o_matrix; d_probs = (bsa_diagdp_prob_t*)((uint8_t*)ws + o_probs);
		d_T = (uint32_t*)((uint8_t*)ws + o_T); d_toff = (uint64_t*)((uint8_t*)ws + o_toff);
		if((e = hipMemcpyAsync(d_planes, planes, planes_bytes, hipMemcpyHostToDevice, c->stream)) != hipSuccess){ fail("hipMemcpyAsync", e); break; }
		if((e = hipMemcpyAsync(d_probs, probs, n * sizeof(bsa_diagdp_prob_t), hipMemcpyHostToDevice, c->stream)) != hipSuccess){ fail("hipMemcpyAsync", e); break; }
		if((e = hipMemcpyAsync(d_toff, toff.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream)) != hipSuccess){ fail("hipMemcpyAsync", e); break; }
		(void)hipEventCreate(&ev0); (void)hipEventCreate(&ev1);
		(void)hipEventRecord(ev0, c->stream);
		if((e = bsa_launch_diagdp(d_planes, d_probs, d_T, d_toff, d_matrix, (uint32_t)n, W, max_len, c->stream)) != hipSuccess){ fail("bsa_launch_diagdp", e); break; }
		(void)hipEventRecord(ev1, c->stream);
		// only the rows the DP wrote travel back (the reference never reads others of this read)
		for(size_t k = 0; k < n && rc == BSA_OK; k++){
			const bsa_diagdp_prob_t &p = probs[k];
			if(p.mend == p.mbeg) continue;
			const uint64_t first = 2ull * p.mbeg * rowlen, bytes = (2ull * (p.mend - p.mbeg)) * rowlen;
			if((e = hipMemcpyAsync(matrix + p.out0 + first, d_matrix + p.out0 + first, bytes, hipMemcpyDeviceToHost, c->stream)) != hipSuccess){ fail("hipMemcpyAsync", e); break; }
			if((e = hipMemcpyAsync(matrix + p.out1 + first, d_matrix + p.out1 + first, bytes, hipMemcpyDeviceToHost, c->stream)) != hipSuccess){ fail("hipMemcpyAsync", e); break; }
		}
		if(rc != BSA_OK) break;
		if((e = hipStreamSynchronize(c->stream)) != hipSuccess){ fail("hipStreamSynchronize", e); break; }
		float ms = 0; if(hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) c->diagdp_ms = ms;
	} while(0);
	if(ev0) (void)hipEventDestroy(ev0);
	if(ev1) (void)hipEventDestroy(ev1);
	return rc;
}
