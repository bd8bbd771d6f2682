// Shared by the translation units of the kernel-pooling backward (kernel_pool_bwd.hip: ABI entry, per-element and exact-f32
// tiled kernels; kernel_pool_bwd_split.hip: the split-bf16 streaming kernel).
#pragma once
#include "mm_internal.h"

namespace mm {

constexpr int kBK = 32;  // max kernels

struct KpBwdArgs {
  const float* q;
  const float* d;
  PackedMask qm, dm;
  const float* mu;
  const float* sigma;
  const float* alpha;
  const float* w;
  const float* go;
  const float* dw;  // optional gate [n_pairs, D]
  float* gdw;       // optional grad of the gate [n_pairs, D]
  float clamp_min;
  float* gq;
  float* gd;
  float* galpha;  // [n_pairs, K]
  float* gw;      // [n_pairs, K]
  int64_t n_pairs;
  int Q, D, E, K;
  // small batches (kernel_pool_bwd_split.hip): S > 1 workgroups per pair, workgroup s takes the blocks s, s + S, ... and leaves its
  // share of grad_q (raw accumulators [Q][E]) + of the own-direction sums (+ workgroup 0: 1 / (|q| + tiny), |q|) in `part`:
  // [n_pairs][S][Q * E + 96] floats; kp_bwd_combine_kernel finishes grad_q
  int nsplit;
  float* part;
};

__device__ __forceinline__ bool mask_bit(const PackedMask& m, int64_t row, int words, int pos, int L) {
  int len = m.len ? m.len[row] : L;
  if (pos >= len) return false;
  if (m.bits) return (m.bits[row * words + (pos >> 5)] >> (pos & 31)) & 1u;
  return true;
}

__device__ __forceinline__ constexpr int mrow(int i) { return (i & 3) + 8 * (i >> 2); }   // C/D layout of the 32x32 MFMA: acc[i] of lane l = row mrow(i) + 4 (l >> 5), column l & 31

// kernel_pool_bwd_split.hip: Q <= 32, K = 11, E = 4n <= 384.  pkq_in: the forward's pooled kernel sums [n_pairs, Q, K]
// (NULL: a pooling pre-pass of the same kernel writes them to pkq_ws first — the document then crosses HBM twice).
bool kp_bwd_split_supported(int Q, int E, int K);
size_t kp_bwd_split_ws_bytes(int64_t n_pairs, int Q, int K);
// workgroups per pair the launch would use for n_pairs (1 from 129 pairs on), and the bytes of the partial buffer that needs
int kp_bwd_split_nsplit(int64_t n_pairs, int D);
size_t kp_bwd_split_part_bytes(int64_t n_pairs, int Q, int D, int E);
int kp_bwd_split_launch(const KpBwdArgs& a, const float* pkq_in, float* pkq_ws, float* part, size_t part_bytes, hipStream_t stream);

}  // namespace mm
