// bsa_cns_dev.hip -- consensus calling of MANY windows' MSAs on the device (include/bsalign_msa.h: bsa_msa_call_consensus_batch;
// cns_bspoa, bspoa.h:3457-3733, with sum_log_nums :3413-3453 and the tail probabilities :3391-3411).
//
// The model and the order of every floating-point addition are those of the host form (bsa_cns.cpp, which is the reference's bit for
// bit): a left-to-right automaton over the MSA columns with five states; entering state a from state e is worth the sum over the reads,
// IN ROW ORDER FROM ZERO, of the log probability of what each read does in the column.  A window is sequential in its columns (the
// masses, the anchors and every read's last event come from the column before), so the device's parallelism is
//   * the 25 (source, target) sums of a column, one lane each (k_cns_dp: a wave per window; a lane walks the reads in row order through
//     one byte per read and source state -- the read's last event and symbol folded into a table index -- and one 8-byte table entry);
//   * the reads of a column for everything that is per read (symbols, votes, the events of the next column);
//   * the columns for everything that is per column once the automaton has run (k_cns_finish: qualities, compaction);
//   * the windows of the batch.
// The logarithm TABLES come from the host (bsa_cns_model_internal: the host's libm), so every sum is the host's bit for bit.  The merges
// log(exp(a) + exp(b)) and the quality formulas call the device's exp / log / erfc: those may differ from glibc's in the last place, so
// the masses can differ from the host's in their last bits -- the consensus, both quality strings and every column's three bytes are
// compared byte for byte in tests/test_cns_gpu.py, the log probability with a relative tolerance of 1e-12 (stated there).
// Compiled with -ffp-contract=off: no multiply-add the host does not make.
#include "../../include/bsalign_msa.h"
#include "../../include/bsalign_hip.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

extern "C" int bsa_ctx_get_stream_internal(bsa_ctx_t *c, hipStream_t *st);
extern "C" int bsa_ctx_scratch_internal(bsa_ctx_t *c, int slot, size_t bytes, void **out);
extern "C" void bsa_cns_model_internal(const bsa_cns_params_t *par, double *cost, uint8_t *next, double *lf, uint32_t nlf, double *consts);

namespace {

constexpr double kDead = -1000000000.0;
constexpr double kPhredCap = 90.0;
constexpr int kGap = 4;
constexpr int kAbsentIdx = 25;                 // table index of a read that is not in the column: five zeros

struct CnsWinDev {
	uint64_t cols_off, idxs_off;               // window's columns in the blob; its column order in the index blob (words), ~0: storage order
	uint64_t out_off;                          // where its consensus / qualities go
	uint64_t mass_off, org_off;                // workspace: (mlen + 1) x 5 doubles; (mlen + 1) packed origins (3 bits a state)
	uint32_t nall, nseq, nmax, mlen;
};

struct CnsModelDev {
	double cost[5][26][5];                     // [anchor][event * 5 + symbol | 25][target]
	uint8_t next[5][25][5];
	uint8_t pad[7];
	double ln10, logp, log1mp, psub;
};

__device__ __forceinline__ double merge_log(double acc, double v){
	if(v == kDead) return acc;
	const double top = (v > acc) ? v : acc, low = (v > acc) ? acc : v;
	if(top >= low + 40) return top;
	return top + log(1 + exp(low - top));
}

__device__ __forceinline__ void wave_sync(){
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The automaton: one wave per window.
__global__ void __launch_bounds__(64) k_cns_dp(uint8_t *cols, const uint32_t *idxs, const CnsWinDev *wins, const CnsModelDev *model, uint8_t *ws, uint32_t nsp){
	extern __shared__ __attribute__((aligned(16))) uint8_t sm[];
	double (*tab)[26][5] = (double (*)[26][5])sm;                                   // 5 x 26 x 5 doubles
	uint8_t (*nxt)[25][5] = (uint8_t (*)[25][5])(sm + sizeof(double) * 5 * 26 * 5);  // 5 x 25 x 5 bytes
	uint8_t *sym = sm + sizeof(double) * 650 + 640;                                  // [nsp]
	uint8_t *evA = sym + nsp, *evB = evA + 5 * (size_t)nsp, *idx = evB + 5 * (size_t)nsp;       // events behind / being built [5][nsp]; table indices [5][nsp]
	const int lane = threadIdx.x;
	const CnsWinDev w = wins[blockIdx.x];
	for(int i = lane; i < 650; i += 64) ((double*)tab)[i] = ((const double*)model->cost)[i];
	for(int i = lane; i < 625; i += 64) ((uint8_t*)nxt)[i] = ((const uint8_t*)model->next)[i];
	const uint32_t nseq = w.nseq;
	for(uint32_t i = lane; i < 5 * nsp; i += 64){ evA[i] = 0; evB[i] = 0; idx[i] = kAbsentIdx; }
	double *mass = (double*)(ws + w.mass_off);
	uint16_t *org = (uint16_t*)(ws + w.org_off);
	const size_t stride = (size_t)w.nall + 3;
	const uint32_t *ix = (w.idxs_off == ~0ull) ? nullptr : idxs + w.idxs_off;
	// lanes 0..4 carry the states (mass, anchor); lane L < 25 the sum source L / 5 -> target L % 5
	double m = (lane == kGap) ? 0.0 : kDead;
	int anc = kGap;
	if(lane < 5) mass[lane] = m;
	if(lane == 0) org[0] = (uint16_t)(kGap | kGap << 3 | kGap << 6 | kGap << 9 | kGap << 12);
	const int se = lane / 5, ta = lane % 5;
	uint8_t *ev = evA, *ev2 = evB;
	wave_sync();
	for(uint32_t pos = 0; pos < w.mlen; pos++){
		const uint8_t *colp = cols + w.cols_off + (size_t)(ix ? ix[pos] : pos) * stride;
		// the column's symbols, the votes, the table index of every read for every source state
		uint32_t votes[5] = {0, 0, 0, 0, 0}, present = 0;
		for(uint32_t r0 = 0; r0 < nseq; r0 += 64){
			const uint32_t r = r0 + lane;
			const uint32_t b = (r < nseq) ? colp[r] : 255u;
			if(r < nseq){
				sym[r] = (uint8_t)b;
#pragma unroll
				for(int e = 0; e < 5; e++) idx[e * nsp + r] = (b <= (uint32_t)kGap) ? (uint8_t)(ev[e * nsp + r] * 5 + b) : (uint8_t)kAbsentIdx;
			}
#pragma unroll
			for(int a = 0; a < 5; a++) votes[a] += __popcll(__ballot(b == (uint32_t)a));
			present += __popcll(__ballot(b <= (uint32_t)kGap));
		}
		wave_sync();
		// the 25 sums, each over the reads in row order from zero (an absent read adds +0.0: the sum's bits do not change)
		const double m0e = __shfl(m, se < 5 ? se : 0);
		const int a0e = __shfl(anc, se < 5 ? se : 0);
		double arrive = kDead;
		if(lane < 25 && m0e != kDead){
			double run = 0;
			const uint8_t *ip = idx + se * nsp;
			const double *tb = &tab[a0e][0][ta];
			uint32_t r = 0;
			for(; r + 4 <= nseq; r += 4){
				const uint32_t i4 = *(const uint32_t*)(ip + r);
				const double c0 = tb[(i4 & 255u) * 5], c1 = tb[((i4 >> 8) & 255u) * 5], c2 = tb[((i4 >> 16) & 255u) * 5], c3 = tb[(i4 >> 24) * 5];
				run += c0; run += c1; run += c2; run += c3;
			}
			for(; r < nseq; r++) run += tb[ip[r] * 5];
			arrive = run + m0e;
		}
		// state a (lanes 0..4): the five arrivals in source order, the mass, the best source, the anchor
		double in[5];
#pragma unroll
		for(int e = 0; e < 5; e++) in[e] = __shfl(arrive, (e * 5 + lane) & 63);
		const uint32_t quorum = (uint32_t)(0.1 * present);
		int best = kGap, nanc = kGap; bool sup = true;
		double m1 = kDead;
		if(lane < 5){
			const uint32_t mine = lane == 0 ? votes[0] : lane == 1 ? votes[1] : lane == 2 ? votes[2] : lane == 3 ? votes[3] : votes[4];
			sup = !(present && mine < max(quorum, 1u));
			if(sup){
				double acc = kDead;
#pragma unroll
				for(int e = 0; e < 5; e++) acc = merge_log(acc, in[e]);
				m1 = acc;
#pragma unroll
				for(int e = 0; e < kGap; e++) if(in[e] > in[best]) best = e;          // the empty state wins ties, then the lower base
			}
		}
		const int abest = __shfl(anc, best);                                       // anchor of the best source (of the column behind)
		if(lane < 5){
			nanc = !sup ? kGap : (lane < kGap) ? lane : abest;
			mass[(size_t)(pos + 1) * 5 + lane] = m1;
		}
		// packed origins of the column; every lane learns (supported, best, anchor of best) of the five states
		uint32_t o15 = 0; int bs[5], ab[5]; bool sp[5];
#pragma unroll
		for(int a = 0; a < 5; a++){ bs[a] = __shfl(best, a); ab[a] = __shfl(abest, a); sp[a] = __shfl((int)sup, a) != 0; o15 |= (uint32_t)bs[a] << (3 * a); }
		if(lane == 0) org[pos + 1] = (uint16_t)o15;
		// the reads' events on the best path into every state
		for(uint32_t r = lane; r < nseq; r += 64){
			const uint32_t b = sym[r];
#pragma unroll
			for(int a = 0; a < 5; a++){
				uint8_t v;
				if(!sp[a]) v = 0;                                                      // unsupported state: all aligned
				else if(b > (uint32_t)kGap) v = 4;                                     // not in the column
				else v = nxt[ab[a]][ev[bs[a] * nsp + r] * 5 + b][a];
				ev2[a * nsp + r] = v;
			}
		}
		m = m1; anc = nanc;
		uint8_t *t = ev; ev = ev2; ev2 = t;
		wave_sync();
	}
}

__device__ __forceinline__ uint8_t phred_byte(double q){ return (uint8_t)(int)(q < kPhredCap ? q : kPhredCap); }

// probability that fewer than k of n reads show an error of rate p (bsa_cns.cpp: fewer_errors)
__device__ double fewer_errors(uint32_t n, uint32_t k, double p, double logp, double log1mp, const double *lf){
	if(n > 50 && n * p > 5 && n * (1 - p) > 5)
		return erfc(-((k - n * p) / sqrt(n * p * (1 - p))) / 1.4142135623731) / 2;
	double tail = 0;
	for(uint32_t e = 0; e < k; e++){
		const double ways = (n <= 1000) ? lf[n] - lf[e] - lf[n - e] : 1;
		tail += exp(logp * e + log1mp * (n - e) + ways);
	}
	return tail;
}

// After the automaton: the best final state and the chain of origins behind it (one lane, the origins through LDS in pieces), then per
// column the two qualities and the compaction of the called bases.  One wave per window.
#define CNS_CHUNK 8192
__global__ void __launch_bounds__(64) k_cns_finish(uint8_t *cols, const uint32_t *idxs, const CnsWinDev *wins, const CnsModelDev *model, const double *lf,
		uint8_t *ws, uint8_t *cns, uint8_t *qlt, uint8_t *alt, uint32_t *clen, double *score){
	__shared__ uint16_t so[CNS_CHUNK];
	__shared__ uint8_t sc[CNS_CHUNK];
	const int lane = threadIdx.x;
	const CnsWinDev w = wins[blockIdx.x];
	const double *mass = (const double*)(ws + w.mass_off);
	const uint16_t *org = (const uint16_t*)(ws + w.org_off);
	const size_t stride = (size_t)w.nall + 3;
	const uint32_t *ix = (w.idxs_off == ~0ull) ? nullptr : idxs + w.idxs_off;
	if(w.mlen == 0){ if(lane == 0){ clen[blockIdx.x] = 0; score[blockIdx.x] = 0; } return; }
	int state = kGap;
	{
		const double *mf = mass + (size_t)w.mlen * 5;
		for(int a = 0; a < kGap; a++) if(mf[a] > mf[state]) state = a;
		if(lane == 0) score[blockIdx.x] = mf[state];
	}
	// column pos is called `state`, then state = origin[pos + 1][state]; pieces of CNS_CHUNK columns from the end
	for(uint32_t hi = w.mlen; hi > 0; ){
		const uint32_t lo = hi > CNS_CHUNK ? hi - CNS_CHUNK : 0;
		for(uint32_t i = lo + lane; i < hi; i += 64) so[i - lo] = org[i + 1];
		__syncthreads();
		if(lane == 0){
			for(uint32_t pos = hi; pos-- > lo; ){
				sc[pos - lo] = (uint8_t)state;
				state = (so[pos - lo] >> (3 * state)) & 7;
			}
		}
		state = __shfl(state, 0);
		__syncthreads();
		for(uint32_t i = lo + lane; i < hi; i += 64) cols[w.cols_off + (size_t)(ix ? ix[i] : i) * stride + w.nall] = sc[i - lo];
		__syncthreads();
		hi = lo;
	}
	__threadfence();                                   // (the called states are read back from memory below)
	// qualities, a lane per column; the called bases of 64 columns leave together
	const double ln10 = model->ln10, psub = model->psub, logp = model->logp, log1mp = model->log1mp;
	uint32_t nc = 0;
	for(uint32_t p0 = 0; p0 < w.mlen; p0 += 64){
		const uint32_t pos = p0 + lane;
		uint32_t called = kGap; uint8_t q1 = 0, q2b = 0;
		if(pos < w.mlen){
			uint8_t *colp = cols + w.cols_off + (size_t)(ix ? ix[pos] : pos) * stride;
			called = colp[w.nall];
			const double *m1 = mass + (size_t)(pos + 1) * 5;
			double all = kDead;
			for(int i = 0; i < 5; i++) all = merge_log(all, m1[i]);
			const double others = log(1 - exp(m1[called] - all));
			q1 = phred_byte(-(10 * (others) / ln10));
			uint32_t votes[5] = {0, 0, 0, 0, 0}, total = 0;
			for(uint32_t r = 0; r < w.nmax; r++){ const uint32_t b = colp[r]; if(b <= (uint32_t)kGap){ votes[b]++; total++; } }
			uint32_t rival = (called + 1) % 5;
			for(uint32_t e = 0; e < 5; e++) if(e != called && votes[e] > votes[rival]) rival = e;
			const double by_error = fewer_errors(total, votes[rival], psub, logp, log1mp, lf);
			const double q2 = (by_error == 0) ? 0 : -(10 * log(1 - by_error) / ln10);
			q2b = phred_byte(q2);
			colp[w.nall + 1] = q1; colp[w.nall + 2] = q2b;
		}
		const uint64_t bm = __ballot(pos < w.mlen && called < (uint32_t)kGap);
		if(pos < w.mlen && called < (uint32_t)kGap){
			const uint32_t at = nc + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull));
			if(cns) cns[w.out_off + at] = (uint8_t)called;
			if(qlt) qlt[w.out_off + at] = q1;
			if(alt) alt[w.out_off + at] = q2b;
		}
		nc += (uint32_t)__popcll(bm);
	}
	if(lane == 0) clen[blockIdx.x] = nc;
}

}  // namespace

extern "C" int bsa_msa_call_consensus_batch(bsa_ctx_t *c, uint8_t *cols, size_t cols_bytes, const uint32_t *idxs, size_t idxs_words,
		const bsa_cns_window_t *win, size_t nwin, const bsa_cns_params_t *par, uint8_t *cns, uint8_t *qlt, uint8_t *alt, size_t out_bytes,
		uint32_t *clen, double *score){
	if(!c || !par || (nwin && (!win || !cols))) return BSA_E_ARG;
	if(nwin == 0) return BSA_OK;
	if(nwin > 0x7fffffffu) return BSA_E_ARG;
	hipStream_t st;
	int rc = bsa_ctx_get_stream_internal(c, &st);
	if(rc != BSA_OK) return rc;
	std::vector<CnsWinDev> wd(nwin);
	uint32_t max_nseq = 1, max_nmax = 0;
	size_t wsb = 0;
	for(size_t k = 0; k < nwin; k++){
		const bsa_cns_window_t &w = win[k];
		const size_t stride = (size_t)w.nall + 3;
		if(w.nseq > w.nall || w.nmax > w.nall) return BSA_E_ARG;
		if(w.mlen){
			if(w.idxs_off == ~0ull){ if(w.cols_off + (size_t)w.mlen * stride > cols_bytes) return BSA_E_ARG; }
			else {
				if(!idxs || w.idxs_off + w.mlen > idxs_words) return BSA_E_ARG;
				for(uint32_t i = 0; i < w.mlen; i++) if(w.cols_off + ((size_t)idxs[w.idxs_off + i] + 1) * stride > cols_bytes) return BSA_E_ARG;
			}
			if((cns || qlt || alt) && w.out_off + w.mlen > out_bytes) return BSA_E_ARG;
		}
		CnsWinDev &d = wd[k];
		d.cols_off = w.cols_off; d.idxs_off = w.idxs_off; d.out_off = w.out_off; d.nall = w.nall; d.nseq = w.nseq; d.nmax = w.nmax; d.mlen = w.mlen;
		d.mass_off = wsb; wsb += (((size_t)w.mlen + 1) * 5 * sizeof(double) + 255) & ~(size_t)255;
		d.org_off = wsb; wsb += (((size_t)w.mlen + 1) * sizeof(uint16_t) + 255) & ~(size_t)255;
		max_nseq = std::max(max_nseq, w.nseq); max_nmax = std::max(max_nmax, w.nmax);
	}
	const uint32_t nsp = (max_nseq + 3u) & ~3u;
	const size_t lds = sizeof(double) * 650 + 640 + (size_t)nsp * 16;
	if(lds > 160 * 1024) return BSA_E_UNSUPPORTED;                           // (about 9900 reads in a window)
	// the model: tables from the host's libm
	std::vector<uint8_t> mh(sizeof(CnsModelDev), 0);
	CnsModelDev *mod = (CnsModelDev*)mh.data();
	const uint32_t nlf = std::min<uint32_t>(std::max<uint32_t>(max_nmax, 1u), 1000u) + 1u;
	std::vector<double> lf(nlf);
	{
		double cost[5][5][5][5]; uint8_t next[5][5][5][5]; double consts[4];
		bsa_cns_model_internal(par, &cost[0][0][0][0], &next[0][0][0][0], lf.data(), nlf, consts);
		for(int a0 = 0; a0 < 5; a0++){
			for(int d = 0; d < 5; d++) for(int b = 0; b < 5; b++) for(int a = 0; a < 5; a++){ mod->cost[a0][d * 5 + b][a] = cost[a0][d][b][a]; mod->next[a0][d * 5 + b][a] = next[a0][d][b][a]; }
			for(int a = 0; a < 5; a++) mod->cost[a0][kAbsentIdx][a] = 0.0;
		}
		mod->ln10 = consts[0]; mod->logp = consts[1]; mod->log1mp = consts[2]; mod->psub = consts[3];
	}
	auto up = [](size_t b){ return (b + 255) & ~(size_t)255; };
	const size_t outb = (cns || qlt || alt) ? out_bytes : 0;
	const size_t o_cols = 0, o_idx = o_cols + up(cols_bytes), o_win = o_idx + up(idxs_words * 4 + 4), o_mod = o_win + up(nwin * sizeof(CnsWinDev)),
		o_lf = o_mod + up(sizeof(CnsModelDev)), o_cns = o_lf + up(nlf * 8), o_qlt = o_cns + up(outb), o_alt = o_qlt + up(outb), o_clen = o_alt + up(outb),
		o_score = o_clen + up(nwin * 4), o_ws = o_score + up(nwin * 8), total = o_ws + wsb + 256;
	void *buf = nullptr;
	rc = bsa_ctx_scratch_internal(c, 1, total, &buf);
	if(rc != BSA_OK) return rc;
	uint8_t *d = (uint8_t*)buf;
#define CNS_TRY(call) do { hipError_t _e = (call); if(_e != hipSuccess) return (_e == hipErrorOutOfMemory) ? BSA_E_NOMEM : BSA_E_HIP; } while(0)
	CNS_TRY(hipMemcpyAsync(d + o_cols, cols, cols_bytes, hipMemcpyHostToDevice, st));
	if(idxs && idxs_words) CNS_TRY(hipMemcpyAsync(d + o_idx, idxs, idxs_words * 4, hipMemcpyHostToDevice, st));
	CNS_TRY(hipMemcpyAsync(d + o_win, wd.data(), nwin * sizeof(CnsWinDev), hipMemcpyHostToDevice, st));
	CNS_TRY(hipMemcpyAsync(d + o_mod, mod, sizeof(CnsModelDev), hipMemcpyHostToDevice, st));
	CNS_TRY(hipMemcpyAsync(d + o_lf, lf.data(), nlf * 8, hipMemcpyHostToDevice, st));
	if(lds > 64 * 1024) CNS_TRY(hipFuncSetAttribute((const void*)k_cns_dp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipLaunchKernelGGL(k_cns_dp, dim3((uint32_t)nwin), dim3(64), lds, st, d + o_cols, (const uint32_t*)(d + o_idx), (const CnsWinDev*)(d + o_win), (const CnsModelDev*)(d + o_mod), d + o_ws, nsp);
	CNS_TRY(hipGetLastError());
	hipLaunchKernelGGL(k_cns_finish, dim3((uint32_t)nwin), dim3(64), 0, st, d + o_cols, (const uint32_t*)(d + o_idx), (const CnsWinDev*)(d + o_win), (const CnsModelDev*)(d + o_mod),
		(const double*)(d + o_lf), d + o_ws, cns ? d + o_cns : nullptr, qlt ? d + o_qlt : nullptr, alt ? d + o_alt : nullptr, (uint32_t*)(d + o_clen), (double*)(d + o_score));
	CNS_TRY(hipGetLastError());
	CNS_TRY(hipMemcpyAsync(cols, d + o_cols, cols_bytes, hipMemcpyDeviceToHost, st));           // the three consensus bytes of every column
	if(cns) CNS_TRY(hipMemcpyAsync(cns, d + o_cns, out_bytes, hipMemcpyDeviceToHost, st));
	if(qlt) CNS_TRY(hipMemcpyAsync(qlt, d + o_qlt, out_bytes, hipMemcpyDeviceToHost, st));
	if(alt) CNS_TRY(hipMemcpyAsync(alt, d + o_alt, out_bytes, hipMemcpyDeviceToHost, st));
	if(clen) CNS_TRY(hipMemcpyAsync(clen, d + o_clen, nwin * 4, hipMemcpyDeviceToHost, st));
	if(score) CNS_TRY(hipMemcpyAsync(score, d + o_score, nwin * 8, hipMemcpyDeviceToHost, st));
	CNS_TRY(hipStreamSynchronize(st));
#undef CNS_TRY
	return BSA_OK;
}
