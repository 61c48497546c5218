// fetch_probe.hip -- calibration of the FETCH_SIZE counter for the READ patterns of the traceback walkers: a wave reads
// PIECE contiguous bytes (16 bytes a lane, PIECE / 16 lanes together) from a distant place, then the next piece far away -- the code rows of
// the 8-bit walker are 64-byte pieces (one row of one pair), the plane rows of the edit walker 64-byte pieces of two planes, against the wide
// coalesced stream (1024 contiguous bytes a wave) the guide calibrates (counter = 1/2 of the bytes there).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/fetch_probe tools/fetch_probe.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o f -- /tmp/fetch_probe
// prints the bytes each kernel really requested; the counter (KB) of the same dispatch divided by that is the calibration factor.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// every group of LANES = PIECE / 16 lanes reads one piece per round at a pseudo-random 64-byte-aligned (PIECE-aligned when ALIGNED) place of a 16 GiB buffer
template<int PIECE, bool ALIGNED>
__global__ void k_fetch_probe(const uint4 *src, size_t n16, int rounds, uint32_t *sink){
	constexpr int LANES = PIECE / 16;
	const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	const size_t grp = t / LANES, in = t % LANES;
	uint32_t acc = 0;
	for(int r = 0; r < rounds; r++){
		size_t at = ((grp * 2654435761ull + (size_t)r * 1000003ull) * 40503ull) % (n16 - 64);
		at &= ALIGNED ? ~(size_t)(LANES - 1) : ~(size_t)3;          // piece-aligned, or 64-byte-aligned only
		const uint4 v = src[at + in];
		acc += v.x ^ v.y ^ v.z ^ v.w;
	}
	if(acc == 0xDEADBEEFu) sink[0] = acc;
}
int main(){
	const size_t bytes = (size_t)16 << 30, n16 = bytes / 16;
	uint4 *d = nullptr; uint32_t *sink = nullptr;
	if(hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess){ printf("alloc failed\n"); return 1; }
	(void)hipMemset(d, 1, bytes);
	const int blocks = 16384, threads = 256, rounds = 64;
	const double req = (double)blocks * threads * rounds * 16.0;
#define RUNK(P, A) hipLaunchKernelGGL((k_fetch_probe<P, A>), dim3(blocks), dim3(threads), 0, 0, d, n16, rounds, sink); (void)hipDeviceSynchronize(); printf("k_fetch_probe<%d, %s> requested %.0f bytes\n", P, #A, req);
	RUNK(16, true) RUNK(32, true) RUNK(64, true) RUNK(64, false) RUNK(128, true) RUNK(128, false) RUNK(256, true) RUNK(1024, true)
	(void)hipFree(d); (void)hipFree(sink);
	return 0;
}
