// Internal helpers shared by the HIP translation units of libmm_native.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/mm_native.h"

namespace mm {

typedef __attribute__((ext_vector_type(8))) short short8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kWave = 64;
constexpr int kCUs = 256;  // MI355X

int set_error(int code, const char* fmt, ...);

// Tuning / A-B knobs read from the environment ONCE per process.  env() returns a function-local static const
// (C++11 guarantees thread-safe initialisation), so the library holds no unsynchronised mutable state: the
// operators are called concurrently from nn.DataParallel's per-GPU threads (matchmaker/train.py:201).
struct EnvCfg {
  int maxsim_nbuf = 2;      // MM_MAXSIM_NBUF: LDS ring depth of the MaxSim roofline kernel (2..4)
  int maxsim_wpc = 4;       // MM_MAXSIM_WPC: wavefronts per CU to launch (0 = what LDS allows)
  int maxsim_nt = 1;        // MM_MAXSIM_NT: non-temporal LDS-DMA
  int maxsim_generic = 0;   // MM_MAXSIM_GENERIC: force the generic MaxSim kernel
  int maxsim_inb_untiled = 0;  // MM_MAXSIM_INB_UNTILED: all-pairs MaxSim with one query per wavefront (A/B runs)
  int maxsim_f32_terms = 3; // MM_MAXSIM_F32_TERMS: 2 = two-term split for fp32 MaxSim (A/B), default three terms
  int kp_generic = 0;       // MM_KP_GENERIC: force the generic pooling kernel
  int kp_f32mfma = 0;       // MM_KP_F32MFMA: exact-f32 MFMA pooling kernel instead of split-bf16
  int dot_prof = 0;         // MM_DOT_PROF: in-kernel phase counters of the dot top-k kernel
};
const EnvCfg& env();

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(MM_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return MM_OK;
}

// ---- mask handling -------------------------------------------------------------------------
// The kernels consume, per mask row, an int32 "effective length" (positions >= len are padding
// and never need to be loaded) and optionally one validity bit per position (for non-prefix
// masks).  LEN_I32 masks are used as-is; U8/I64/F32 masks are packed by pack_mask_kernel.
struct PackedMask {
  const int32_t* len = nullptr;    // [rows] or null (all positions real)
  const uint32_t* bits = nullptr;  // [rows, ceil(L/32)] or null
};

size_t packed_mask_bytes(int kind, int64_t rows, int L);
// Resolves a user mask to a PackedMask, launching the pack kernel into `ws` when needed.
// Advances *ws / *ws_left.  Returns MM_OK or an error code.
// Dense masks may be strided: row i starts at mask + i*row_stride + col0 (row_stride 0 = L).
int resolve_mask(const void* mask, int kind, int64_t rows, int L, char** ws, size_t* ws_left,
                 hipStream_t stream, PackedMask* out, int64_t row_stride = 0, int col0 = 0);

constexpr int kK = 11;   // RBF kernels of TK / TKL (tk.yaml:18-19, tkl.yaml)
constexpr int kKC = 12;  // K + the non-zero-count channel of TKL's pair sums

struct TklParams {            // offsets into the packed float parameter vector (see mm_native.h)
  __host__ __device__ static int mu() { return 0; }
  __host__ __device__ static int sigma() { return kK; }
  __host__ __device__ static int dense() { return 2 * kK; }
  __host__ __device__ static int kmult() { return 3 * kK; }
  __host__ __device__ static int sat() { return 4 * kK; }       // w1[2] b1 w2[2] b2 w3[2] b3 lnw[2] lnb[2]
  __host__ __device__ static int chunk_scoring() { return 4 * kK + 13; }
  __host__ __device__ static int emb() { return 4 * kK + 13 + 15; }
};

// kernel_pool.hip exports used by tkl.hip
bool kp_stream_supported(int Q, int E);
int tkl_stage1_stream(const float* q_ctx, const float* chunks, PackedMask dm, const int32_t* q_len,
                      const int32_t* chunk_slot, int C, const float* mu, const float* sigma, float* ps_out, int64_t P,
                      int Q, int E, int32_t* slot2p, int64_t n_slots, hipStream_t stream);

// kernel_pool128.hip: fp32 MaxSim on the split-bf16 streaming kernel (E = 64n <= 384, 512, 768; Q <= 32), called from maxsim.hip
bool kp128_maxsim_supported(int Q, int E);
int kp128_maxsim_f32(const float* q, const float* d, PackedMask qm, PackedMask dm, float* out, int64_t n_pairs,
                     int64_t pairs_per_query, int Q, int D, int E, hipStream_t stream);

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float neg_inf() { return -__builtin_huge_valf(); }

}  // namespace mm
