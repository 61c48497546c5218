#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define BSA_COMMON_ONLY
static __device__ __forceinline__ int dpp_keep(int x){ asm("" : "+v"(x)); return x; }
typedef short xv2s __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ uint32_t x_max(uint32_t a, uint32_t b){ return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(xv2s, a), __builtin_bit_cast(xv2s, b))); }
#define XDPP(old, x, ctrl, bank) ((uint32_t)dpp_keep(__builtin_amdgcn_update_dpp((int)(old), (int)(x), (ctrl), 0xf, (bank), false)))
#define XQP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define XROW_SHL(n) (0x100 + (n))
#define XROW_SHR(n) (0x110 + (n))
static __device__ __forceinline__ uint32_t x_bcast_first(uint32_t x){
	const uint32_t s = XDPP(0, x, XQP(0, 0, 0, 0), 0xf);
	return XDPP(s, s, XROW_SHR(4), 0xA);
}
static __device__ __forceinline__ uint32_t x_bcast_last(uint32_t x){
	const uint32_t s = XDPP(0, x, XQP(3, 3, 3, 3), 0xf);
	return XDPP(s, s, XROW_SHL(4), 0x5);
}
static __device__ __forceinline__ uint32_t x_shift_down(uint32_t x, uint32_t fill_lo, bool first){
	const uint32_t s = XDPP(0, x, XROW_SHR(1), 0xf);
	const uint32_t w = XDPP(0, x, XROW_SHL(7), 0xf);
	const uint32_t fix = (w << 16) | (fill_lo & 0xffffu);
	return first ? fix : s;
}
static __device__ __forceinline__ uint32_t x_shift_up(uint32_t x, uint32_t fill, bool last){
	const uint32_t s = XDPP(0, x, XROW_SHL(1), 0xf);
	const uint32_t w = XDPP(0, x, XROW_SHR(7), 0xf);
	const uint32_t fix = __builtin_amdgcn_alignbit(fill, w, 16);
	return last ? fix : s;
}
static __device__ __forceinline__ uint32_t x_scan_max8(uint32_t x){
	uint32_t y = XDPP(x, x, XQP(0, 0, 2, 2), 0xf); x = x_max(x, y);
	y = XDPP(x, x, XQP(0, 1, 1, 1), 0xf); x = x_max(x, y);
	const uint32_t z = XDPP(x, x, XQP(3, 3, 3, 3), 0xf);
	y = XDPP(x, z, XROW_SHR(4), 0xA); x = x_max(x, y);
	return x;
}
__global__ void k(const uint32_t *in, uint32_t *out){
	int t = threadIdx.x; int jl = t & 7;
	uint32_t x = in[t];
	out[t] = x_bcast_first(x); out[64 + t] = x_bcast_last(x);
	out[128 + t] = x_shift_down(x, 0xAAAA, jl == 0); out[192 + t] = x_shift_up(x, 0xBBBBBBBB, jl == 7);
	out[256 + t] = x_scan_max8(x);
}
int main(){
	uint32_t h[64], o[320];
	for(int i = 0; i < 64; i++){ int lo = (i * 37) % 23 - 11, hi = 100 + (i * 13) % 17; h[i] = ((uint32_t)lo & 0xffff) | ((uint32_t)hi << 16); }
	uint32_t *a, *b; hipMalloc(&a, 256); hipMalloc(&b, 1280);
	hipMemcpy(a, h, 256, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b);
	hipMemcpy(o, b, 1280, hipMemcpyDeviceToHost);
	int bad = 0;
	for(int i = 0; i < 64; i++){
		int g0 = i & ~7, jl = i & 7;
		uint32_t ef = h[g0], el = h[g0 + 7];
		uint32_t ed = jl ? h[i - 1] : ((h[g0 + 7] << 16) | 0xAAAA);
		uint32_t eu = jl < 7 ? h[i + 1] : ((0xBBBBu << 16) | (h[g0] >> 16));
		short mlo = -32768, mhi = -32768;
		for(int j = g0; j <= i; j++){ short l = (short)(h[j] & 0xffff), hh = (short)(h[j] >> 16); if(l > mlo) mlo = l; if(hh > mhi) mhi = hh; }
		uint32_t es = ((uint32_t)(uint16_t)mlo) | ((uint32_t)(uint16_t)mhi << 16);
		if(o[i] != ef){ bad++; printf("bcast_first lane %d got %08x want %08x\n", i, o[i], ef); }
		if(o[64 + i] != el){ bad++; printf("bcast_last lane %d got %08x want %08x\n", i, o[64 + i], el); }
		if(o[128 + i] != ed){ bad++; printf("shift_down lane %d got %08x want %08x\n", i, o[128 + i], ed); }
		if(o[192 + i] != eu){ bad++; printf("shift_up lane %d got %08x want %08x\n", i, o[192 + i], eu); }
		if(o[256 + i] != es){ bad++; printf("scan lane %d got %08x want %08x\n", i, o[256 + i], es); }
	}
	printf("bad %d\n", bad);
	return 0;
}
