// Micro-benchmark: issue cost (shader cycles per wave-instruction) of the instructions the TK / TKL
// epilogue and the split-bf16 path are made of, one wavefront per SIMD (the occupancy those kernels
// run at).  Build: hipcc --offload-arch=gfx950 -O3 -o inst_cost inst_cost.hip ; run on an MI355X.
// Every sequence is 8 independent dependency chains x 32 = 256 instructions between two s_memtime.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(B) B B B B B B B B B B B B B B B B B B B B B B B B B B B B B B B B

__device__ __forceinline__ uint64_t now() {
  uint64_t t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

#define KERNEL1(NAME, ASM)                                                                    \
  __global__ void __launch_bounds__(64) NAME(float* out, uint64_t* cyc, float seed) {         \
    float v0 = seed, v1 = seed + 1, v2 = seed + 2, v3 = seed + 3, v4 = seed + 4, v5 = seed + 5, \
          v6 = seed + 6, v7 = seed + 7;                                                       \
    float c = seed * 0.5f;                                                                    \
    const uint64_t t0 = now();                                                                \
    asm volatile(REP32(ASM) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(c)); \
    const uint64_t t1 = now();                                                                \
    out[blockIdx.x * 64 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;               \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                          \
  }

KERNEL1(k_fma, "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n")
KERNEL1(k_mul, "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
KERNEL1(k_exp, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
KERNEL1(k_log, "v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n v_log_f32 %4, %4\n v_log_f32 %5, %5\n v_log_f32 %6, %6\n v_log_f32 %7, %7\n")
KERNEL1(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
KERNEL1(k_cvtpk, "v_cvt_pk_bf16_f32 %0, %0, %8\n v_cvt_pk_bf16_f32 %1, %1, %8\n v_cvt_pk_bf16_f32 %2, %2, %8\n v_cvt_pk_bf16_f32 %3, %3, %8\n v_cvt_pk_bf16_f32 %4, %4, %8\n v_cvt_pk_bf16_f32 %5, %5, %8\n v_cvt_pk_bf16_f32 %6, %6, %8\n v_cvt_pk_bf16_f32 %7, %7, %8\n")
KERNEL1(k_and, "v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n")
KERNEL1(k_cnd, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
// exp interleaved 1:1 with fma (does the transcendental overlap plain VALU?)
KERNEL1(k_exp_fma, "v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %8\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %8, %8\n v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %8, %8\n v_exp_f32 %6, %6\n v_fma_f32 %7, %7, %8, %8\n")
// exp : 3 fma
KERNEL1(k_exp_3fma, "v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n")

#define KERNEL2(NAME, ASM)                                                                    \
  __global__ void __launch_bounds__(64) NAME(float* out, uint64_t* cyc, float seed) {         \
    f32x2 v0 = {seed, seed}, v1 = v0 + 1.0f, v2 = v0 + 2.0f, v3 = v0 + 3.0f, v4 = v0 + 4.0f, v5 = v0 + 5.0f, \
          v6 = v0 + 6.0f, v7 = v0 + 7.0f;                                                     \
    f32x2 c = {seed * 0.5f, seed * 0.25f};                                                    \
    const uint64_t t0 = now();                                                                \
    asm volatile(REP32(ASM) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(c)); \
    const uint64_t t1 = now();                                                                \
    const f32x2 s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;                                    \
    out[blockIdx.x * 64 + threadIdx.x] = s[0] + s[1];                                         \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                          \
  }
KERNEL2(k_pkfma, "v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n")
KERNEL2(k_pkmul, "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
KERNEL2(k_pkadd, "v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n")

// MFMA bf16 32x32x16: 4 independent accumulators x 8 = 32 MFMAs; optional VALU fillers per MFMA
template <int FILL>
__global__ void __launch_bounds__(64) k_mfma(float* out, uint64_t* cyc, float seed) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  bf16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(seed + i); y[i] = (__bf16)(seed - i); }
  float f0 = seed, f1 = seed + 1, f2 = seed + 2, f3 = seed + 3;
  const uint64_t t0 = now();
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < FILL; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f0) : "v"(f1));
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < FILL; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f1) : "v"(f2));
    a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < FILL; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f2) : "v"(f3));
    a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < FILL; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f3) : "v"(f0));
  }
  const uint64_t t1 = now();
  float s = f0 + f1 + f2 + f3;
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// LDS-DMA issue cost: NI x global_load_lds_dwordx4 (1 KiB each) issued back to back, then drained;
// reports cycles until the LAST instruction has ISSUED (t1) and until all data landed (t2).
template <int NI>
__global__ void __launch_bounds__(64) k_ldsdma(const char* src, float* out, uint64_t* cyc, uint64_t* cyc2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const char* g = src + (size_t)blockIdx.x * NI * 1024 * 8;
  uint32_t voff = threadIdx.x * 16;
  uint64_t t0 = 0, t1 = 0, t2 = 0;
  for (int rep = 0; rep < 8; ++rep) {
    const char* gg = g + rep * NI * 1024;
    t0 = now();
    asm volatile("s_mov_b32 m0, %0" ::"s"(lds0));
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 0x400" ::"v"(voff + i * 1024), "s"(gg) : "memory", "scc");
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    t2 = now();
  }
  out[blockIdx.x * 64 + threadIdx.x] = *(float*)(smem + threadIdx.x * 4);
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc2[blockIdx.x] = t2 - t0; }
}

// LDS read latency: 13 ds_read_b128 then wait
__global__ void __launch_bounds__(64) k_ldsread(float* out, uint64_t* cyc) {
  __shared__ __attribute__((aligned(16))) char smem[16384];
  for (int i = threadIdx.x; i < 4096; i += 64) ((float*)smem)[i] = i;
  __syncthreads();
  const char* p = smem + (threadIdx.x & 31) * 400 + (threadIdx.x >> 5) * 32;
  const uint64_t t0 = now();
  f32x4 x[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) x[i] = *(const volatile f32x4*)(p + i * 32);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const uint64_t t1 = now();
  float s = 0;
  for (int i = 0; i < 13; ++i) s += x[i][0] + x[i][3];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static double avg(const std::vector<uint64_t>& v) { double s = 0; for (auto x : v) s += x; return s / v.size(); }

int main() {
  const int NB = 1024;  // 4 waves per CU
  float* out; uint64_t *cyc, *cyc2; char* src;
  hipMalloc(&out, NB * 64 * 4); hipMalloc(&cyc, NB * 8); hipMalloc(&cyc2, NB * 8);
  hipMalloc(&src, (size_t)NB * 13 * 1024 * 8); hipMemset(src, 1, (size_t)NB * 13 * 1024 * 8);
  std::vector<uint64_t> h(NB), h2(NB);
  auto report = [&](const char* name, int ninst, double base) {
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, NB * 8, hipMemcpyDeviceToHost);
    printf("%-22s %8.1f cycles total  %6.2f cycles/inst\n", name, avg(h), (avg(h) - base) / ninst);
  };
#define RUN1(K, N) for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(K, dim3(NB), dim3(64), 0, 0, out, cyc, 1.0f); report(#K, N, 0);
  RUN1(k_fma, 256) RUN1(k_mul, 256) RUN1(k_exp, 256) RUN1(k_log, 256) RUN1(k_rcp, 256) RUN1(k_cvtpk, 256) RUN1(k_and, 256) RUN1(k_cnd, 256)
  RUN1(k_exp_fma, 256) RUN1(k_exp_3fma, 256) RUN1(k_pkfma, 256) RUN1(k_pkmul, 256) RUN1(k_pkadd, 256)
  RUN1(k_mfma<0>, 32) RUN1(k_mfma<2>, 32) RUN1(k_mfma<4>, 32) RUN1(k_mfma<6>, 32) RUN1(k_mfma<8>, 32) RUN1(k_mfma<12>, 32)
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_ldsread, dim3(NB), dim3(64), 0, 0, out, cyc);
  report("k_ldsread(13xb128)", 13, 0);
#define RUND(NI) for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_ldsdma<NI>, dim3(NB), dim3(64), NI * 1024, 0, src, out, cyc, cyc2); \
  hipDeviceSynchronize(); hipMemcpy(h.data(), cyc, NB * 8, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), cyc2, NB * 8, hipMemcpyDeviceToHost); \
  printf("k_ldsdma<%d>  issue %8.1f cycles (%6.1f / inst)   landed %8.1f cycles\n", NI, avg(h), avg(h) / NI, avg(h2));
  RUND(1) RUND(4) RUND(8) RUND(13)
  hipError_t e = hipGetLastError();
  printf("status: %s\n", hipGetErrorString(e));
  return 0;
}
