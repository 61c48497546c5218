// write_probe.hip -- calibration of the WRITE_SIZE counter for the store pattern of the compact forward kernels: 16-byte stores, runs of
// RUN consecutive lanes writing contiguous bytes, runs scattered over a large buffer (every run in another pair's slot).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/write_probe tools/write_probe.hip
//   rocprofv3 --pmc WRITE_SIZE --output-format csv -d out -o w -- /tmp/write_probe
// prints the bytes each kernel really stored; the counter (KB) of the same dispatch divided by that is the calibration factor.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template<int RUN>
__global__ void k_write_probe(uint4 *dst, size_t slots, size_t slot16, int rounds){
	const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	const size_t run = t / RUN, in = t % RUN;
	for(int r = 0; r < rounds; r++){
		// run `run` writes into slot (run * 2654435761 + r) % slots: a different distant slot every round, 16 * RUN contiguous bytes
		const size_t slot = (run * 2654435761ull + (size_t)r * 40503ull) % slots;
		dst[slot * slot16 + (size_t)(r % (int)(slot16 / RUN)) * RUN + in] = make_uint4((uint32_t)t, (uint32_t)r, 0u, 0u);
	}
}
int main(){
	const size_t slots = 1 << 16, slot16 = 4096;      // 65536 slots of 64 KB = 4 GB
	uint4 *d = nullptr;
	if(hipMalloc(&d, slots * slot16 * 16) != hipSuccess){ printf("alloc failed\n"); return 1; }
	hipMemset(d, 0, slots * slot16 * 16);
	const int blocks = 16384, threads = 256, rounds = 64;
	const double bytes = (double)blocks * threads * rounds * 16.0;
#define RUNK(R) hipLaunchKernelGGL(k_write_probe<R>, dim3(blocks), dim3(threads), 0, 0, d, slots, slot16, rounds); hipDeviceSynchronize(); printf("k_write_probe<%d> stored %.0f bytes\n", R, bytes);
	RUNK(1) RUNK(4) RUNK(8) RUNK(16) RUNK(64)
	hipFree(d);
	return 0;
}
