#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* p, float* o1, float* o2) {
  float v = p[threadIdx.x];
  unsigned u = __builtin_bit_cast(unsigned, v);
  unsigned xa = u, ya = u;
  asm volatile("v_mov_b32 %1, %2\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(xa), "=&v"(ya) : "v"(u));
  unsigned r[2] = {xa, ya};
  o1[threadIdx.x] = __builtin_bit_cast(float, r[0]);
  o2[threadIdx.x] = __builtin_bit_cast(float, r[1]);
  unsigned xb = u, yb = u;
  asm volatile("v_mov_b32 %1, %2\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(xb), "=&v"(yb) : "v"(u));
  unsigned r2[2] = {xb, yb};
  o1[64 + threadIdx.x] = __builtin_bit_cast(float, r2[0]);
  o2[64 + threadIdx.x] = __builtin_bit_cast(float, r2[1]);
}
int main() {
  float h[64], a[128], b[128]; for (int i = 0; i < 64; ++i) h[i] = i;
  float *d, *o1, *o2; hipMalloc(&d, 256); hipMalloc(&o1, 512); hipMalloc(&o2, 512);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o1, o2);
  hipMemcpy(a, o1, 512, hipMemcpyDeviceToHost); hipMemcpy(b, o2, 512, hipMemcpyDeviceToHost);
  for (int s = 0; s < 2; ++s) { printf("%s r0:", s ? "swap32" : "swap16"); for (int i = 0; i < 64; i += 4) printf(" %g", a[64*s+i]); printf("\n       r1:"); for (int i = 0; i < 64; i += 4) printf(" %g", b[64*s+i]); printf("\n"); }
}
