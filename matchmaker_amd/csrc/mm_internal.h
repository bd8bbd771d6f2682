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
  int maxsim_no_inline_masks = 0;  // MM_MAXSIM_NO_INLINE_MASKS: pack int64 masks in their own launch even for one-pair-per-wavefront calls (A/B runs)
  int maxsim_no_wpp2 = 0;          // MM_MAXSIM_NO_WPP2: one wavefront per pair also in eval.py-sized calls (A/B runs)
  int maxsim_inb_nowg = 0;     // MM_MAXSIM_INB_NOWG: all-pairs MaxSim without the workgroup-shared ring (A/B runs)
  int maxsim_f32_terms = 3; // MM_MAXSIM_F32_TERMS: 2 = two-term split for fp32 MaxSim (A/B), default three terms
  int kp_generic = 0;       // MM_KP_GENERIC: force the generic pooling kernel
  int kp_f32mfma = 0;       // MM_KP_F32MFMA: exact-f32 MFMA pooling kernel instead of split-bf16
  int dot_prof = 0;         // MM_DOT_PROF: in-kernel phase counters of the dot top-k kernel
  int kp_bwd_threads = 1024; // MM_KP_BWD_THREADS: 1024 (when its LDS fits) | 512 threads per pair in kernel_pool_bwd_tiled_kernel (A/B runs)
  int tkl_stage1_ksplit = 0;  // MM_TKL_STAGE1_KSPLIT=1: TKL stage 1 on the two-wavefront K-split kernel (tkl_stage1_ksplit.hip) instead of the row-streaming one (A/B runs)
  int tkl_stage1_slices = 0;  // MM_TKL_STAGE1_SLICES=1: TKL stage 1 (cosine hand-off) on the K-sliced ring of rounds 2-5 instead of the row-streaming kernel (A/B runs)
  int kp_bwd_nsplit = 0;    // MM_KP_BWD_NSPLIT=n: workgroups per pair of the split backward (0: by batch size — 4 up to 128 pairs, else 1)
  int kp_bwd_f32 = 0;       // MM_KP_BWD_F32=1: pooling backward on the exact-f32 tiled kernel of rounds 4-5 instead of the split-bf16 streaming kernel (parity twin, A/B runs)
  int kp_bwd_untiled = 0;   // MM_KP_BWD_UNTILED: pooling backward on the per-element kernel of rounds 1-3 (A/B runs)
  int kp_multi_wg = 0;        // MM_KP_MULTI_WG=1: the multi launch as one workgroup per (pair range, document tensor) with a wavefront per query
                              // tensor and a rate barrier per block, instead of independent workgroups in flat XCD-grouped order (A/B runs:
                              // measured SLOWER, 9.19 vs 8.40 ms for Conv-KNRM 3 x 3 — six wavefronts per CU instead of eight)
  int kp_multi_loop = -1;     // MM_KP_MULTI_LOOP: Conv-KNRM's multi launch as one wavefront per (pair range, document tensor) looping over the query
                              // tensors (kernel_pool_multi128_kernel: every document block crosses HBM once, 22.5 GB fetched instead of 47.3).
                              // -1 = by launch size (>= 2,048 pairs), 0 / 1 = never / always (A/B runs)
  int kp_multi_2d = 0;        // MM_KP_MULTI_2D=1: Conv-KNRM's multi launch on the 2-D grid of rounds 1-4 instead of the flat XCD-grouped order (A/B runs)
  int kp128_occ = 0;          // MM_KP128_OCC: 0 = choose by shape, 1 / 2 = wavefronts per SIMD of the 64n-wide pooling kernel (A/B runs)
  int tkl_bwd_nosplit = 0;    // MM_TKL_BWD_NOSPLIT=1: TKL's backward with one workgroup per document at every batch size (A/B runs)
  int tkl_fold_regions = 0;   // MM_TKL_FOLD_REGIONS=1: TKL's region top-k in the last window workgroup of each document (round 4's default) instead of
                              // its own launch (A/B runs: the hand-off written to the memory model costs +50 us per 256-document call, see tkl.hip)
  int tkl_pairsums = 0;     // MM_TKL_PAIRSUMS: TKL stage 1 emits pair sums (round-2 data path) instead of cosines (A/B runs)
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

// Both masks of a call: two dense masks of one element type are packed by ONE launch (else one resolve_mask each).
int resolve_mask_pair(const void* m0, int kind0, int64_t rows0, int L0, PackedMask* out0, const void* m1, int kind1,
                      int64_t rows1, int L1, PackedMask* out1, char** ws, size_t* ws_left, hipStream_t stream);

constexpr int kK = 11;   // RBF kernels of TK / TKL (tk.yaml:18-19, tkl.yaml)
constexpr int kKC = 12;  // K + the non-zero-count channel of TKL's pair sums

// RBF kernels by middle-out recurrence where the kernel set allows it (kp_device.h rbf_geo_one, tkl.hip's window staging);
// -DMM_RBF_GEO=0 builds the direct form everywhere (A/B libraries).
#ifndef MM_RBF_GEO
#define MM_RBF_GEO 1
#endif
constexpr float kGeoClamp = 13.0f;   // exp2(-13^2) = 0 exactly (below the smallest denormal): a clamped masked row adds exactly 0

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
                      int Q, int E, int32_t* slot2p, int64_t n_slots, hipStream_t stream, float* cos_out = nullptr);
bool tkl_cos_supported(int Q, int E);

// kernel_pool128.hip: fp32 MaxSim on the split-bf16 streaming kernel (E = 64n <= 384, 512, 768; Q <= 32), called from maxsim.hip
bool kp128_maxsim_supported(int Q, int E);
int kp128_maxsim_f32(const float* q, const float* d, PackedMask qm, PackedMask dm, float* out, int64_t n_pairs,
                     int64_t pairs_per_query, int Q, int D, int E, hipStream_t stream);

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n.  n >= 63 needs no wait: the counter has 6 bits, so at most 63
// vector-memory operations are outstanding and anything with that many younger ones behind it has completed.
// On gfx9-family hardware loads and stores of a wavefront retire in issue order (one counter, one event class in
// LLVM's SIInsertWaitcnts for targets without vscnt), so stores may be counted like the LDS-DMA loads around them.
__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
    case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
    case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
    case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
    case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break;
    case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
    case 31: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
    case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    case 33: asm volatile("s_waitcnt vmcnt(33)" ::: "memory"); break;
    case 34: asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); break;
    case 35: asm volatile("s_waitcnt vmcnt(35)" ::: "memory"); break;
    case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
    case 37: asm volatile("s_waitcnt vmcnt(37)" ::: "memory"); break;
    case 38: asm volatile("s_waitcnt vmcnt(38)" ::: "memory"); break;
    case 39: asm volatile("s_waitcnt vmcnt(39)" ::: "memory"); break;
    case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    case 41: asm volatile("s_waitcnt vmcnt(41)" ::: "memory"); break;
    case 42: asm volatile("s_waitcnt vmcnt(42)" ::: "memory"); break;
    case 43: asm volatile("s_waitcnt vmcnt(43)" ::: "memory"); break;
    case 44: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break;
    case 45: asm volatile("s_waitcnt vmcnt(45)" ::: "memory"); break;
    case 46: asm volatile("s_waitcnt vmcnt(46)" ::: "memory"); break;
    case 47: asm volatile("s_waitcnt vmcnt(47)" ::: "memory"); break;
    case 48: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
    case 49: asm volatile("s_waitcnt vmcnt(49)" ::: "memory"); break;
    case 50: asm volatile("s_waitcnt vmcnt(50)" ::: "memory"); break;
    case 51: asm volatile("s_waitcnt vmcnt(51)" ::: "memory"); break;
    case 52: asm volatile("s_waitcnt vmcnt(52)" ::: "memory"); break;
    case 53: asm volatile("s_waitcnt vmcnt(53)" ::: "memory"); break;
    case 54: asm volatile("s_waitcnt vmcnt(54)" ::: "memory"); break;
    case 55: asm volatile("s_waitcnt vmcnt(55)" ::: "memory"); break;
    case 56: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
    case 57: asm volatile("s_waitcnt vmcnt(57)" ::: "memory"); break;
    case 58: asm volatile("s_waitcnt vmcnt(58)" ::: "memory"); break;
    case 59: asm volatile("s_waitcnt vmcnt(59)" ::: "memory"); break;
    case 60: asm volatile("s_waitcnt vmcnt(60)" ::: "memory"); break;
    case 61: asm volatile("s_waitcnt vmcnt(61)" ::: "memory"); break;
    case 62: asm volatile("s_waitcnt vmcnt(62)" ::: "memory"); break;
    default: break;
  }
}

__device__ __forceinline__ float neg_inf() { return -__builtin_huge_valf(); }

}  // namespace mm
