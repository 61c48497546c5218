// write_bw_probe.hip -- what HBM write bandwidth many concurrent streams reach as a function of the contiguous chunk a stream writes at a time
// (the edit forward kernel: 32768 pairs, each appending 512 bytes -- a tile of eight rows -- every eight rows).
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/write_bw_probe.bin tools/write_bw_probe.hip ; run on the GPU box
// STREAMS streams of SLOT bytes each; a group of LPS = CHUNK / 16 lanes owns a stream and appends CHUNK bytes per round (16 bytes a lane), all streams
// advancing together.  Prints GB/s per chunk size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_streams(uint4 *dst, size_t slot16, uint32_t lps, int rounds){
	const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	const size_t stream = t / lps, in = t % lps;
	uint4 *p = dst + stream * slot16 + in;
	for(int r = 0; r < rounds; r++){ *p = make_uint4((uint32_t)t, (uint32_t)r, 1u, 2u); p += lps; }
}
int main(){
	const size_t streams = 32768, slot = (size_t)2 << 20;          // 64 GB in all
	uint4 *d = nullptr;
	if(hipMalloc(&d, streams * slot) != hipSuccess){ printf("alloc failed\n"); return 1; }
	hipMemset(d, 0, streams * slot);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const uint32_t chunks[] = {64, 128, 256, 512, 1024, 2048, 4096};
	for(uint32_t c : chunks){
		const uint32_t lps = c / 16;
		const size_t threads = streams * lps;
		const int rounds = (int)(slot / c);
		for(int rep = 0; rep < 2; rep++){
			hipEventRecord(e0);
			hipLaunchKernelGGL(k_streams, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, d, slot / 16, lps, rounds);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms = 0; hipEventElapsedTime(&ms, e0, e1);
			if(rep) printf("chunk %5u B: %6.1f ms, %7.1f GB/s (%zu streams, %d rounds)\n", c, ms, (double)streams * slot / ms / 1e6, streams, rounds);
		}
	}
	return 0;
}
