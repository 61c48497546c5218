#include <hip/hip_runtime.h>
#include <stdio.h>
#define DPP_SHR(fill, x, n)  __builtin_amdgcn_update_dpp((fill), (x), 0x110 + (n), 0xf, 0xf, false)
__global__ void k1(const int *a, const int *b, int *o1){
  int t = threadIdx.x;
  int av = a[t], bv = b[t];
  int vsh = DPP_SHR(0, bv, 1);
  o1[t] = min(max(av - vsh, -128), 127);
}
__global__ void k2(const int *a, const int *b, int *o2){
  int t = threadIdx.x;
  int av = a[t], bv = b[t];
  int vsh2 = DPP_SHR(0, bv, 1);
  asm volatile("" : "+v"(vsh2));
  o2[t] = min(max(av - vsh2, -128), 127);
}
int main(){
  int ha[64], hb[64], h1[64], h2[64];
  for(int i=0;i<64;i++){ ha[i] = 10+i; hb[i] = 3*i+1; }
  int *a,*b,*o1,*o2;
  hipMalloc(&a,256); hipMalloc(&b,256); hipMalloc(&o1,256); hipMalloc(&o2,256);
  hipMemcpy(a,ha,256,hipMemcpyHostToDevice); hipMemcpy(b,hb,256,hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k1, dim3(1), dim3(64), 0, 0, a,b,o1);
  hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, a,b,o2);
  hipMemcpy(h1,o1,256,hipMemcpyDeviceToHost); hipMemcpy(h2,o2,256,hipMemcpyDeviceToHost);
  for(int i=0;i<20;i++) printf("lane %d a %d b %d fused %d unfused %d expect %d\n", i, ha[i], hb[i], h1[i], h2[i], ha[i] - ((i%16)? hb[i-1] : 0));
  return 0;
}
