// Backward of TK kernel pooling (training path: train.py:503-524 through ecai20_tk.py:105-124).
//
//   qh_i = q_i / (|q_i| + tiny), dh_j = d_j / (|d_j| + tiny), c_ij = <qh_i, dh_j>
//   e_ijk = exp(-(c_ij - mu_k)^2 / (2 sigma_k^2)),  pkq_ik = sum_j dmask_j e_ijk
//   out   = sum_k w_k sum_i qmask_i log(max(alpha_k pkq_ik, 1e-10))
//
// With g = d loss / d out:
//   A_ik  = g qmask_i w_k [alpha_k pkq_ik >= 1e-10] / pkq_ik                       (through log o clamp)
//   G_ij  = dmask_j sum_k A_ik e_ijk (-(c_ij - mu_k) / sigma_k^2)                   (= d loss / d c_ij)
//   grad_q_i = rq_i (sum_j G_ij dh_j - (sum_j G_ij c_ij) q_i / |q_i|)              rq_i = 1/(|q_i| + tiny)
//   grad_d_j = rd_j (sum_i G_ij qh_i - (sum_i G_ij c_ij) d_j / |d_j|)
//   grad_w_k     = g sum_i qmask_i log(max(alpha_k pkq_ik, 1e-10))                  (per pair; host sums)
//   grad_alpha_k = g w_k sum_i qmask_i [alpha_k pkq_ik >= 1e-10] / alpha_k
// (the norm's gradient at a zero vector is 0, as torch.norm's backward defines it).
//
// Variants (mm_kernel_pool_ex_bwd): a document-token gate s_j >= 0 (TK-Sparse, cikm20_tk_sparse.py:133-135)
// replaces dmask_j by dmask_j s_j in pkq and G, and gets grad_s_j = dmask_j sum_ik A_ik e_ijk; the floor
// 1e-10 inside the log is a parameter (1e-4: IDCM sampler, sigir21_idcm.py:185).
//
// One workgroup (256 threads) per pair, fp32 VALU throughout: training batches are tens of pairs, so
// this is a correctness path — what matters is that loss.backward() stays on the device without a
// [B,Q,D,K] tensor.  Pair-per-row layout (the one train.py feeds: neuralIR_encoder.py:86-87).
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
};

__device__ __forceinline__ bool mask_bit(const PackedMask& m, int64_t row, int words, int pos, int L) {
  int len = m.len ? m.len[row] : L;
  if (pos >= len) return false;
  if (m.bits) return (m.bits[row * words + (pos >> 5)] >> (pos & 31)) & 1u;
  return true;
}

__global__ void __launch_bounds__(256) kernel_pool_bwd_kernel(const KpBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t pair = blockIdx.x;
  const int Q = a.Q, D = a.D, E = a.E, K = a.K;
  float* C = (float*)smem;            // [Q][D] cosines
  float* G = C + Q * D;               // [Q][D] d loss / d c
  float* A = G + Q * D;               // [Q][K]
  float* rq = A + Q * kBK;            // [Q] 1/(|q|+tiny)
  float* nq = rq + Q;                 // [Q] |q|
  float* rd = nq + Q;                 // [D]
  float* nd = rd + D;                 // [D]
  float* sq = nd + D;                 // [Q] sum_j G c
  float* td = sq + Q;                 // [D] sum_i G c
  float* qmf = td + D;                // [Q] 0/1
  float* dmf = qmf + Q;               // [D] 0/1
  float* pw = dmf + D;                // [Q][K] per-(i,k) term of grad_w
  float* pa = pw + Q * kBK;           // [Q][K] per-(i,k) term of grad_alpha
  const float* qb = a.q + pair * Q * (int64_t)E;
  const float* db = a.d + pair * D * (int64_t)E;
  const float g = a.go[pair];
  const int qwords = (Q + 31) >> 5, dwords = (D + 31) >> 5;

  // 1. norms and masks
  for (int row = wave; row < Q + D; row += 4) {
    const float* x = row < Q ? qb + (int64_t)row * E : db + (int64_t)(row - Q) * E;
    float ss = 0.0f;
    for (int e = lane; e < E; e += 64) ss += x[e] * x[e];
    ss = wave_sum(ss);
    if (lane == 0) {
      const float n = sqrtf(ss);
      if (row < Q) { nq[row] = n; rq[row] = 1.0f / (n + 1e-13f); qmf[row] = mask_bit(a.qm, pair, qwords, row, Q) ? 1.0f : 0.0f; }
      else {
        const int j = row - Q;
        nd[j] = n; rd[j] = 1.0f / (n + 1e-13f);
        const float gate = a.dw ? fmaxf(a.dw[pair * D + j], 0.0f) : 1.0f;   // dmf = mask x gate
        dmf[j] = mask_bit(a.dm, pair, dwords, j, D) ? gate : 0.0f;
      }
    }
  }
  __syncthreads();
  // 2. cosine matrix (same factor order as the forward: (dot * rq) * rd)
  for (int idx = tid; idx < Q * D; idx += 256) {
    const int i = idx / D, j = idx - i * D;
    const float* x = qb + (int64_t)i * E;
    const float* y = db + (int64_t)j * E;
    float dot = 0.0f;
    for (int e = 0; e < E; ++e) dot += x[e] * y[e];
    C[idx] = (dot * rq[i]) * rd[j];
  }
  __syncthreads();
  // 3. pooled kernels -> A, parameter gradients of this pair
  for (int idx = tid; idx < Q * K; idx += 256) {
    const int i = idx / K, k = idx - i * K;
    const float mu = a.mu[k], sg = a.sigma[k];
    const float c2 = -1.0f / (2.0f * sg * sg);
    float pk = 0.0f;
    for (int j = 0; j < D; ++j) {
      const float t = C[i * D + j] - mu;
      pk += dmf[j] * __expf(t * t * c2);
    }
    const float al = a.alpha[k];
    const bool live = al * pk >= a.clamp_min;
    A[i * kBK + k] = live ? g * qmf[i] * a.w[k] / pk : 0.0f;
    pw[i * kBK + k] = qmf[i] * __logf(fmaxf(al * pk, a.clamp_min));   // -> grad_w
    pa[i * kBK + k] = live ? qmf[i] * a.w[k] / al : 0.0f;        // -> grad_alpha
  }
  __syncthreads();
  if (tid < K) {
    float sw = 0.0f, sa = 0.0f;
    for (int i = 0; i < Q; ++i) {
      sw += pw[i * kBK + tid];
      sa += pa[i * kBK + tid];
    }
    a.gw[pair * K + tid] = g * sw;
    a.galpha[pair * K + tid] = g * sa;
  }
  __syncthreads();
  // 4. G = d loss / d c
  for (int idx = tid; idx < Q * D; idx += 256) {
    const int i = idx / D, j = idx - i * D;
    const float c = C[idx];
    float s = 0.0f;
    if (dmf[j] != 0.0f) {
      for (int k = 0; k < K; ++k) {
        const float sg = a.sigma[k];
        const float t = c - a.mu[k];
        const float inv = 1.0f / (sg * sg);
        s += A[i * kBK + k] * __expf(-0.5f * t * t * inv) * (-t * inv);
      }
    }
    G[idx] = s * dmf[j];
  }
  __syncthreads();
  // 4b. gradient of the gate: sum_ik A_ik e_ijk on real tokens whose gate is open (relu'(x) = 0 at x <= 0 is
  // the caller's chain rule; a closed gate still gets the gradient of the product, as autograd gives it)
  if (a.gdw) {
    for (int j = tid; j < D; j += 256) {
      float s = 0.0f;
      if (mask_bit(a.dm, pair, dwords, j, D)) {
        for (int i = 0; i < Q; ++i) {
          const float c = C[i * D + j];
          for (int k = 0; k < K; ++k) {
            const float sg = a.sigma[k];
            const float t = c - a.mu[k];
            s += A[i * kBK + k] * __expf(-0.5f * t * t / (sg * sg));
          }
        }
      }
      a.gdw[pair * D + j] = s;
    }
  }
  // 5. sum_j G c (per query token) and sum_i G c (per document token)
  for (int i = tid; i < Q; i += 256) {
    float s = 0.0f;
    for (int j = 0; j < D; ++j) s += G[i * D + j] * C[i * D + j];
    sq[i] = s;
  }
  for (int j = tid; j < D; j += 256) {
    float s = 0.0f;
    for (int i = 0; i < Q; ++i) s += G[i * D + j] * C[i * D + j];
    td[j] = s;
  }
  __syncthreads();
  // 6. grad_q
  float* gq = a.gq + pair * Q * (int64_t)E;
  for (int idx = tid; idx < Q * E; idx += 256) {
    const int i = idx / E, e = idx - i * E;
    float s = 0.0f;
    for (int j = 0; j < D; ++j) s += G[i * D + j] * rd[j] * db[(int64_t)j * E + e];
    const float self = nq[i] > 0.0f ? sq[i] * qb[idx] / nq[i] : 0.0f;
    gq[idx] = rq[i] * (s - self);
  }
  // 7. grad_d
  float* gd = a.gd + pair * D * (int64_t)E;
  for (int idx = tid; idx < D * E; idx += 256) {
    const int j = idx / E, e = idx - j * E;
    float s = 0.0f;
    for (int i = 0; i < Q; ++i) s += G[i * D + j] * rq[i] * qb[(int64_t)i * E + e];
    const float self = nd[j] > 0.0f ? td[j] * db[idx] / nd[j] : 0.0f;
    gd[idx] = rd[j] * (s - self);
  }
}

}  // namespace mm

using namespace mm;

extern "C" size_t mm_kernel_pool_bwd_workspace_bytes(int64_t n_pairs, int Q, int D, int q_mask_kind, int d_mask_kind) {
  return packed_mask_bytes(q_mask_kind, n_pairs, Q) + packed_mask_bytes(d_mask_kind, n_pairs, D);
}

extern "C" int mm_kernel_pool_ex_bwd(const void* q, const void* d, const void* q_mask, int q_mask_kind, const void* d_mask,
                                     int d_mask_kind, const float* d_gate, const float* mu, const float* sigma,
                                     const float* alpha, const float* w, float clamp_min, const float* grad_out,
                                     float* grad_q, float* grad_d, float* grad_gate, float* grad_alpha, float* grad_w,
                                     int64_t n_pairs, int Q, int D, int E, int K, void* workspace, size_t workspace_bytes,
                                     void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!q || !d || !mu || !sigma || !alpha || !w || !grad_out || !grad_q || !grad_d || !grad_alpha || !grad_w)
    return set_error(MM_EINVAL, "kernel_pool_bwd: null pointer");
  if (grad_gate && !d_gate) return set_error(MM_EINVAL, "kernel_pool_bwd: grad_gate without d_gate");
  if (n_pairs < 0 || Q <= 0 || D <= 0 || E <= 0 || K <= 0) return set_error(MM_EINVAL, "kernel_pool_bwd: bad shape");
  if (!(clamp_min > 0.0f)) return set_error(MM_EINVAL, "kernel_pool_bwd: clamp_min must be > 0");
  if (K > kBK) return set_error(MM_EUNSUPPORTED, "kernel_pool_bwd: K=%d kernels (max %d)", K, kBK);
  if (n_pairs == 0) return MM_OK;
  if (n_pairs > 0x7fffffffLL) return set_error(MM_EUNSUPPORTED, "kernel_pool_bwd: too many pairs for one launch");
  const size_t lds = ((size_t)2 * Q * D + (size_t)3 * Q * kBK + 4 * (size_t)Q + 4 * (size_t)D) * 4;
  if (lds > 160 * 1024) return set_error(MM_EUNSUPPORTED, "kernel_pool_bwd: Q x D = %d x %d exceeds the LDS tile", Q, D);
  KpBwdArgs a{};
  a.q = (const float*)q; a.d = (const float*)d; a.mu = mu; a.sigma = sigma; a.alpha = alpha; a.w = w; a.go = grad_out;
  a.dw = d_gate; a.gdw = grad_gate; a.clamp_min = clamp_min;
  a.gq = grad_q; a.gd = grad_d; a.galpha = grad_alpha; a.gw = grad_w; a.n_pairs = n_pairs; a.Q = Q; a.D = D; a.E = E; a.K = K;
  char* ws = (char*)workspace;
  size_t left = workspace ? workspace_bytes : 0;
  if (int e = resolve_mask(q_mask, q_mask_kind, n_pairs, Q, &ws, &left, stream, &a.qm)) return e;
  if (int e = resolve_mask(d_mask, d_mask_kind, n_pairs, D, &ws, &left, stream, &a.dm)) return e;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kernel_pool_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kernel_pool_bwd_kernel, dim3((unsigned)n_pairs), dim3(256), lds, stream, a);
  return check_launch("kernel_pool_bwd_kernel");
}

extern "C" int mm_kernel_pool_bwd(const void* q, const void* d, const void* q_mask, int q_mask_kind, const void* d_mask,
                                  int d_mask_kind, const float* mu, const float* sigma, const float* alpha, const float* w,
                                  const float* grad_out, float* grad_q, float* grad_d, float* grad_alpha, float* grad_w,
                                  int64_t n_pairs, int Q, int D, int E, int K, void* workspace, size_t workspace_bytes,
                                  void* stream_) {
  return mm_kernel_pool_ex_bwd(q, d, q_mask, q_mask_kind, d_mask, d_mask_kind, nullptr, mu, sigma, alpha, w, 1e-10f, grad_out,
                               grad_q, grad_d, nullptr, grad_alpha, grad_w, n_pairs, Q, D, E, K, workspace, workspace_bytes,
                               stream_);
}
