// pcie_probe.hip -- what a host-pointer call can expect of this box: pageable against registered (pinned in place) caller buffers.
// hipcc --offload-arch=gfx950 -O2 -o tools/probes/pcie_probe.bin tools/pcie_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now(){ return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do{ hipError_t e_ = (x); if(e_ != hipSuccess){ printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } }while(0)
int main(){
	const size_t UP = (size_t)2500 << 20, DN = (size_t)540 << 20;
	void *d = nullptr; CK(hipMalloc(&d, UP));
	char *h = (char*)malloc(UP); memset(h, 1, UP);
	CK(hipMemcpy(d, h, 64 << 20, hipMemcpyHostToDevice));
	for(int rep = 0; rep < 2; rep++){
		double t0 = now(); CK(hipMemcpy(d, h, UP, hipMemcpyHostToDevice)); double t1 = now();
		printf("H2D pageable 2500 MB: %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, UP / (t1 - t0) / 1e9);
	}
	for(int rep = 0; rep < 2; rep++){
		double t0 = now(); CK(hipHostRegister(h, UP, hipHostRegisterDefault)); double t1 = now();
		CK(hipMemcpy(d, h, UP, hipMemcpyHostToDevice)); double t2 = now();
		CK(hipHostUnregister(h)); double t3 = now();
		printf("H2D registered 2500 MB: register %.1f ms, copy %.1f ms (%.1f GB/s), unregister %.1f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, UP / (t2 - t1) / 1e9, (t3 - t2) * 1e3);
	}
	{
		char *o = (char*)malloc(DN);                 // never touched
		double t0 = now(); CK(hipMemcpy(o, d, DN, hipMemcpyDeviceToHost)); double t1 = now();
		printf("D2H pageable untouched 540 MB: %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, DN / (t1 - t0) / 1e9);
		t0 = now(); CK(hipMemcpy(o, d, DN, hipMemcpyDeviceToHost)); t1 = now();
		printf("D2H pageable touched 540 MB: %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, DN / (t1 - t0) / 1e9);
		t0 = now(); CK(hipHostRegister(o, DN, hipHostRegisterDefault)); t1 = now();
		CK(hipMemcpy(o, d, DN, hipMemcpyDeviceToHost)); double t2 = now();
		CK(hipHostUnregister(o)); double t3 = now();
		printf("D2H registered (touched) 540 MB: register %.1f ms, copy %.1f ms (%.1f GB/s), unregister %.1f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, DN / (t2 - t1) / 1e9, (t3 - t2) * 1e3);
		free(o);
		o = (char*)malloc(DN);
		t0 = now(); CK(hipHostRegister(o, DN, hipHostRegisterDefault)); t1 = now();
		CK(hipMemcpy(o, d, DN, hipMemcpyDeviceToHost)); t2 = now();
		CK(hipHostUnregister(o)); t3 = now();
		printf("D2H registered (fresh malloc) 540 MB: register %.1f ms, copy %.1f ms (%.1f GB/s), unregister %.1f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, DN / (t2 - t1) / 1e9, (t3 - t2) * 1e3);
		free(o);
	}
	{
		void *p = nullptr; double t0 = now(); CK(hipHostMalloc(&p, DN, hipHostMallocDefault)); double t1 = now();
		CK(hipMemcpy(p, d, DN, hipMemcpyDeviceToHost)); double t2 = now();
		char *o = (char*)malloc(DN); memset(o, 0, DN); double t3 = now();
		memcpy(o, p, DN); double t4 = now();
		printf("hipHostMalloc 540 MB %.1f ms; D2H pinned %.1f ms (%.1f GB/s); memcpy pinned -> touched pageable one thread %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, DN / (t2 - t1) / 1e9, (t4 - t3) * 1e3, DN / (t4 - t3) / 1e9);
		CK(hipHostFree(p)); free(o);
	}
	return 0;
}
