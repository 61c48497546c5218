// bsa_synth.hip -- synthetic read-pair generator (measurement inputs).
//
// pair k: target = iid uniform ACGT of length L; query = target with errors at total rate
// err_q32 / 2^32 split substitution : insertion : deletion = 23 : 31 : 46 (the ONT profile of the
// reference's benchmark recipe, /root/reference/example/ScriptsForPaper.txt:9).
//
// All draws are position-indexed outputs of splitmix64, so the host, device and numpy
// (tests/support.py) forms produce identical bytes:
//   G = 0x9E3779B97F4A7C15;  draw(s0, n) = mix(s0 + n*G)            (n-th output of splitmix64(s0))
//   sT = seed ^ (k*G);  sQ = ~sT
//   T[i] = (draw(sT, i/32 + 1) >> (2*(i%32))) & 3
//   z = draw(sQ, i + 1);  error at i  <=>  (z >> 32) < err_q32
//        kind = (z & 0xFFFF) % 100;  aux = (z >> 16) & 0xFFFF
//        kind < 23 : substitute (T[i] + 1 + aux % 3) & 3
//        kind < 54 : insert base (aux & 3) before T[i]
//        else      : delete T[i]
#include "bsa_common.h"

#define SYNTH_G 0x9E3779B97F4A7C15ull

static inline __host__ __device__ uint64_t synth_mix(uint64_t z){
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

static inline __host__ __device__ uint32_t synth_one_pair(uint64_t seed, uint64_t k, uint32_t L, uint32_t err_q32,
		uint8_t *T, uint8_t *Q, uint32_t qcap){
	const uint64_t sT = seed ^ (k * SYNTH_G), sQ = ~sT;
	uint32_t qn = 0;
	uint64_t zt = 0;
	for(uint32_t i = 0; i < L; i++){
		if((i & 31u) == 0) zt = synth_mix(sT + (uint64_t)(i / 32u + 1u) * SYNTH_G);
		const uint8_t tb = (uint8_t)((zt >> (2u * (i & 31u))) & 3u);
		T[i] = tb;
		const uint64_t z = synth_mix(sQ + (uint64_t)(i + 1u) * SYNTH_G);
		if((uint32_t)(z >> 32) >= err_q32){
			if(qn < qcap) Q[qn++] = tb;
		} else {
			const uint32_t kind = (uint32_t)(z & 0xFFFFu) % 100u, aux = (uint32_t)(z >> 16) & 0xFFFFu;
			if(kind < 23u){ if(qn < qcap) Q[qn++] = (uint8_t)((tb + 1u + aux % 3u) & 3u); }
			else if(kind < 54u){
				if(qn < qcap) Q[qn++] = (uint8_t)(aux & 3u);
				if(qn < qcap) Q[qn++] = tb;
			}
		}
	}
	return qn;
}

extern "C" size_t bsa_synth_stride(uint32_t L){ return (((size_t)L + (size_t)L / 4u + 64u) + 63u) & ~(size_t)63u; }

extern "C" int bsa_synth_pairs_host(uint64_t seed, uint64_t first_pair, size_t n, uint32_t L, uint32_t err_q32, uint8_t *seqs, uint32_t *qlen){
	if(!seqs || !qlen) return BSA_E_ARG;
	const size_t stride = bsa_synth_stride(L);
	for(size_t k = 0; k < n; k++)
		qlen[k] = synth_one_pair(seed, first_pair + k, L, err_q32, seqs + k * stride, seqs + (n + k) * stride, (uint32_t)stride);
	return BSA_OK;
}

__global__ void __launch_bounds__(64) k_synth(uint64_t seed, uint64_t first_pair, uint32_t n, uint32_t L, uint32_t err_q32,
		uint8_t *seqs, uint32_t *qlen, uint64_t stride){
	const uint32_t k = blockIdx.x * 64u + threadIdx.x;
	if(k >= n) return;
	qlen[k] = synth_one_pair(seed, first_pair + k, L, err_q32, seqs + (size_t)k * stride, seqs + ((size_t)n + k) * stride, (uint32_t)stride);
}

// the context type is opaque here; only its stream is needed
extern "C" int bsa_ctx_get_stream_internal(bsa_ctx_t *ctx, hipStream_t *st);

extern "C" int bsa_synth_pairs_dev(bsa_ctx_t *ctx, uint64_t seed, uint64_t first_pair, size_t n, uint32_t L, uint32_t err_q32,
		uint8_t *d_seqs, uint32_t *d_qlen){
	if(!ctx || !d_seqs || !d_qlen || n > 0xFFFFFFF0ull) return BSA_E_ARG;
	hipStream_t st;
	int rc = bsa_ctx_get_stream_internal(ctx, &st);
	if(rc != BSA_OK) return rc;
	if(n == 0) return BSA_OK;
	hipLaunchKernelGGL(k_synth, dim3(((uint32_t)n + 63u) / 64u), dim3(64), 0, st, seed, first_pair, (uint32_t)n, L, err_q32, d_seqs, d_qlen, (uint64_t)bsa_synth_stride(L));
	return hipGetLastError() == hipSuccess ? BSA_OK : BSA_E_HIP;
}
