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
// [B,Q,D,K] tensor, for ANY document length (the [Q,D] tiles are swept DT positions at a time).
// Pair-per-row layout (the one train.py feeds: neuralIR_encoder.py:86-87).
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

// LDS: cosine + gradient tiles [Q][DT] (DT document positions at a time) + per-pair vectors.  Documents longer than
// one tile (max_doc_length 2000 in tk.yaml-style training configs) take two sweeps over their tiles: the first pools
// the kernels (pkq needs every position), the second recomputes each tile's cosines and finishes the gradients.
__host__ __device__ inline size_t kp_bwd_lds_bytes(int Q, int DT) {
  return ((size_t)2 * Q * DT + (size_t)4 * Q * kBK + 5 * (size_t)Q + 5 * (size_t)DT) * 4;
}

__global__ void __launch_bounds__(256) kernel_pool_bwd_kernel(const KpBwdArgs a, const int DT) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t pair = blockIdx.x;
  const int Q = a.Q, D = a.D, E = a.E, K = a.K;
  float* C = (float*)smem;            // [Q][DT] cosines of the current tile
  float* G = C + Q * DT;              // [Q][DT] d loss / d c
  float* A = G + Q * DT;              // [Q][K]
  float* PK = A + Q * kBK;            // [Q][K] pooled kernels, accumulated over the tiles
  float* pw = PK + Q * kBK;           // [Q][K] per-(i,k) term of grad_w
  float* pa = pw + Q * kBK;           // [Q][K] per-(i,k) term of grad_alpha
  float* rq = pa + Q * kBK;           // [Q] 1/(|q|+tiny)
  float* nq = rq + Q;                 // [Q] |q|
  float* sq = nq + Q;                 // [Q] sum_j G c (over all tiles)
  float* qmf = sq + Q;                // [Q] 0/1
  float* rd = qmf + Q;                // [DT]
  float* nd = rd + DT;                // [DT]
  float* td = nd + DT;                // [DT] sum_i G c
  float* dmf = td + DT;               // [DT] mask x gate
  float* dmb = dmf + DT;              // [DT] mask alone (0/1)
  const float* qb = a.q + pair * Q * (int64_t)E;
  const float* db = a.d + pair * D * (int64_t)E;
  float* gq = a.gq + pair * Q * (int64_t)E;
  float* gd = a.gd + pair * D * (int64_t)E;
  const float g = a.go[pair];
  const int qwords = (Q + 31) >> 5, dwords = (D + 31) >> 5;
  const bool one_tile = DT >= D;

  // query norms and mask
  for (int row = wave; row < Q; row += 4) {
    const float* x = qb + (int64_t)row * E;
    float ss = 0.0f;
    for (int e = lane; e < E; e += 64) ss += x[e] * x[e];
    ss = wave_sum(ss);
    if (lane == 0) {
      const float n = sqrtf(ss);
      nq[row] = n; rq[row] = 1.0f / (n + 1e-13f); qmf[row] = mask_bit(a.qm, pair, qwords, row, Q) ? 1.0f : 0.0f;
      sq[row] = 0.0f;
    }
  }
  for (int idx = tid; idx < Q * kBK; idx += 256) PK[idx] = 0.0f;
  __syncthreads();

  // document-tile norms, masks and cosines (same factor order as the forward: (dot * rq) * rd)
  auto load_tile = [&](int j0, int nj) {
    for (int row = wave; row < nj; row += 4) {
      const int j = j0 + row;
      const float* x = db + (int64_t)j * E;
      float ss = 0.0f;
      for (int e = lane; e < E; e += 64) ss += x[e] * x[e];
      ss = wave_sum(ss);
      if (lane == 0) {
        const float n = sqrtf(ss);
        nd[row] = n; rd[row] = 1.0f / (n + 1e-13f);
        const float gate = a.dw ? fmaxf(a.dw[pair * D + j], 0.0f) : 1.0f;   // dmf = mask x gate
        const bool real = mask_bit(a.dm, pair, dwords, j, D);
        dmb[row] = real ? 1.0f : 0.0f;
        dmf[row] = real ? gate : 0.0f;
      }
    }
    __syncthreads();
    for (int idx = tid; idx < Q * nj; idx += 256) {
      const int i = idx / nj, jj = idx - i * nj;
      const float* x = qb + (int64_t)i * E;
      const float* y = db + (int64_t)(j0 + jj) * E;
      float dot = 0.0f;
      for (int e = 0; e < E; ++e) dot += x[e] * y[e];
      C[i * DT + jj] = (dot * rq[i]) * rd[jj];
    }
    __syncthreads();
  };

  // ---- sweep 1: pooled kernels pkq_ik = sum_j dmf_j e_ijk -------------------------------------------------
  for (int j0 = 0; j0 < D; j0 += DT) {
    const int nj = D - j0 < DT ? D - j0 : DT;
    load_tile(j0, nj);
    for (int idx = tid; idx < Q * K; idx += 256) {      // (i, k) is owned by one thread across the tiles
      const int i = idx / K, k = idx - i * K;
      const float mu = a.mu[k], sg = a.sigma[k];
      const float c2 = -1.0f / (2.0f * sg * sg);
      float pk = 0.0f;
      for (int jj = 0; jj < nj; ++jj) {
        const float t = C[i * DT + jj] - mu;
        pk += dmf[jj] * __expf(t * t * c2);
      }
      PK[i * kBK + k] += pk;
    }
    __syncthreads();
  }
  // A, parameter gradients of this pair
  for (int idx = tid; idx < Q * K; idx += 256) {
    const int i = idx / K, k = idx - i * K;
    const float pk = PK[i * kBK + k];
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

  // ---- sweep 2: G = d loss / d c per tile, gate gradient, grad_d (complete per tile), grad_q (accumulated) --
  for (int j0 = 0; j0 < D; j0 += DT) {
    const int nj = D - j0 < DT ? D - j0 : DT;
    if (!one_tile) load_tile(j0, nj);       // a single tile is still resident from sweep 1
    for (int idx = tid; idx < Q * nj; idx += 256) {
      const int i = idx / nj, jj = idx - i * nj;
      const float c = C[i * DT + jj];
      float s = 0.0f;
      if (dmf[jj] != 0.0f) {
        for (int k = 0; k < K; ++k) {
          const float sg = a.sigma[k];
          const float t = c - a.mu[k];
          const float inv = 1.0f / (sg * sg);
          s += A[i * kBK + k] * __expf(-0.5f * t * t * inv) * (-t * inv);
        }
      }
      G[i * DT + jj] = s * dmf[jj];
    }
    __syncthreads();
    // gradient of the gate: sum_ik A_ik e_ijk on real tokens (relu'(x) = 0 at x <= 0 is the caller's chain rule;
    // a closed gate still gets the gradient of the product, as autograd gives it)
    if (a.gdw) {
      for (int jj = tid; jj < nj; jj += 256) {
        float s = 0.0f;
        if (dmb[jj] != 0.0f) {
          for (int i = 0; i < Q; ++i) {
            const float c = C[i * DT + jj];
            for (int k = 0; k < K; ++k) {
              const float sg = a.sigma[k];
              const float t = c - a.mu[k];
              s += A[i * kBK + k] * __expf(-0.5f * t * t / (sg * sg));
            }
          }
        }
        a.gdw[pair * D + j0 + jj] = s;
      }
    }
    // sum_j G c (per query token, over all tiles) and sum_i G c (per document token)
    for (int i = tid; i < Q; i += 256) {
      float s = 0.0f;
      for (int jj = 0; jj < nj; ++jj) s += G[i * DT + jj] * C[i * DT + jj];
      sq[i] += s;
    }
    for (int jj = tid; jj < nj; jj += 256) {
      float s = 0.0f;
      for (int i = 0; i < Q; ++i) s += G[i * DT + jj] * C[i * DT + jj];
      td[jj] = s;
    }
    __syncthreads();
    // grad_q: the sum over this tile's document tokens (element idx is owned by one thread across the tiles)
    for (int idx = tid; idx < Q * E; idx += 256) {
      const int i = idx / E, e = idx - i * E;
      float s = j0 == 0 ? 0.0f : gq[idx];
      for (int jj = 0; jj < nj; ++jj) s += G[i * DT + jj] * rd[jj] * db[(int64_t)(j0 + jj) * E + e];
      gq[idx] = s;
    }
    // grad_d of this tile's rows
    for (int idx = tid; idx < nj * E; idx += 256) {
      const int jj = idx / E, e = idx - jj * E;
      float s = 0.0f;
      for (int i = 0; i < Q; ++i) s += G[i * DT + jj] * rq[i] * qb[(int64_t)i * E + e];
      const float x = db[(int64_t)(j0 + jj) * E + e];
      const float self = nd[jj] > 0.0f ? td[jj] * x / nd[jj] : 0.0f;
      gd[(int64_t)(j0 + jj) * E + e] = rd[jj] * (s - self);
    }
    __syncthreads();
  }
  // grad_q: the norm term needs sum_j G c over the whole document
  for (int idx = tid; idx < Q * E; idx += 256) {
    const int i = idx / E;
    const float self = nq[i] > 0.0f ? sq[i] * qb[idx] / nq[i] : 0.0f;
    gq[idx] = rq[i] * (gq[idx] - self);
  }
}

// ---------------------------------------------------------------------------------------------
// Tiled backward (round 4): the same arithmetic with every operand of the three small matrix products of a pair —
// cosines C = Dh Qh^T, grad_d = G^T Qh, grad_q = G Dh — staged in LDS and register-blocked.
//
// kernel_pool_bwd_kernel above takes its dot products straight from global memory, one thread per output element and
// one dependent load per FMA: 1.5-2 ms per PAIR (bench.py extra.train_step, round 4: 16.6 ms for 2,048 pairs, 2.4 x
// slower than torch's eager ops on the same GPU).  Here: the normalised query tile [Q][E] and one 32-row document block
// live in LDS, the cosines of the whole document stay in LDS between the two sweeps (pooled sums need every position
// before any gradient can be formed), and every thread owns a 4 x TQ (cosines) or 4 x 4 (gradients) register tile fed
// by 16-byte LDS reads.  The document is read from HBM/L2 twice and its gradient written once; grad_q accumulates in
// registers across the blocks.  Q <= 32, E <= 384 (16-byte rows), K <= 16, LDS permitting; everything else takes the
// kernel above.
// ---------------------------------------------------------------------------------------------
constexpr int kTK = 16;
constexpr int kTT = 512;   // threads of the tiled kernel: eight wavefronts, two per SIMD, around ONE set of LDS tiles per CU

__host__ __device__ inline size_t kp_bwd_tiled_lds_bytes(int Q, int E, int Dpad) {
  const int ES = E + 4, QS = (Q + 3) & ~3;
  return ((size_t)Q * ES + 32 * (size_t)ES + (size_t)Dpad * QS + 3 * 32 * (size_t)QS + 2 * (size_t)Q * kTK + 4 * (size_t)Dpad +
          5 * 32 + 3 * kTK) * 4;
}

__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

template <bool GATE>
__global__ void __launch_bounds__(kTT) kernel_pool_bwd_tiled_kernel(const KpBwdArgs a, const int Dpad) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int64_t pair = blockIdx.x;
  const int Q = a.Q, D = a.D, E = a.E, K = a.K;
  const int ES = E + 4, QS = (Q + 3) & ~3, NC = E >> 2;
  float* QH = (float*)smem;          // [Q][ES]    q_i / (|q_i| + tiny)
  float* DB = QH + Q * ES;           // [32][ES]   the current document block (raw rows)
  float* CT = DB + 32 * ES;          // [Dpad][QS] cosines of the whole document, [position][query token]
  float* GJ = CT + Dpad * QS;        // [32][QS]   d loss / d c of the block, [position][token]
  float* GI = GJ + 32 * QS;          // [QS][32]   the same, [token][position]
  float* SG = GI + QS * 32;          // [32][QS]   gate-gradient terms of the block (and scratch for grad_alpha)
  float* PK = SG + 32 * QS;          // [Q][kTK]   pooled kernels
  float* A = PK + Q * kTK;           // [Q][kTK]
  float* RD = A + Q * kTK;           // [Dpad] 1 / (|d| + tiny)  (0 past the document)
  float* ND = RD + Dpad;             // [Dpad] |d|
  float* DMF = ND + Dpad;            // [Dpad] mask x gate
  float* DMB = DMF + Dpad;           // [Dpad] mask alone
  float* rq = DMB + Dpad;            // [32]
  float* nq = rq + 32;
  float* sq = nq + 32;               // sum_j G c per query token
  float* qmf = sq + 32;
  float* td = qmf + 32;              // sum_i G c per position of the block
  float* kc = td + 32;               // [3][kTK]: mu, -log2(e) / (2 sigma^2), 1 / sigma^2
  const float* qb = a.q + pair * Q * (int64_t)E;
  const float* db = a.d + pair * D * (int64_t)E;
  float* gq = a.gq + pair * Q * (int64_t)E;
  float* gd = a.gd + pair * D * (int64_t)E;
  const float g = a.go[pair];
  const int qwords = (Q + 31) >> 5, dwords = (D + 31) >> 5;

  // ---- query tile, constants ---------------------------------------------------------------------------------
  for (int idx = tid; idx < Q * NC; idx += kTT) {
    const int i = idx / NC, c = idx - i * NC;
    *(f32x4*)(QH + i * ES + 4 * c) = *(const f32x4*)(qb + (int64_t)i * E + 4 * c);
  }
  if (tid < K) {
    const float sg = a.sigma[tid];
    kc[tid] = a.mu[tid];
    kc[kTK + tid] = -1.4426950408889634f / (2.0f * sg * sg);
    kc[2 * kTK + tid] = 1.0f / (sg * sg);
  }
  for (int idx = tid; idx < Q * kTK; idx += kTT) PK[idx] = 0.0f;
  if (tid < 32) sq[tid] = 0.0f;
  __syncthreads();
  {  // norms: sixteen threads per query token; the tile is stored normalised
    const int i = tid >> 4, sub = tid & 15;
    float ss = 0.0f;
    if (i < Q)
      for (int c = sub; c < NC; c += 16) {
        const f32x4 v = *(const f32x4*)(QH + i * ES + 4 * c);
        ss += dot4(v, v);
      }
    ss += __shfl_xor(ss, 1, 64);
    ss += __shfl_xor(ss, 2, 64);
    ss += __shfl_xor(ss, 4, 64);
    ss += __shfl_xor(ss, 8, 64);
    const float n = sqrtf(ss), r = 1.0f / (n + 1e-13f);
    if (i < Q) {
      for (int c = sub; c < NC; c += 16) {
        f32x4* p = (f32x4*)(QH + i * ES + 4 * c);
        *p = *p * r;
      }
      if (sub == 0) {
        nq[i] = n;
        rq[i] = r;
        qmf[i] = mask_bit(a.qm, pair, qwords, i, Q) ? 1.0f : 0.0f;
      }
    }
  }
  __syncthreads();

  auto load_block = [&](int j0, int nj) {
    for (int idx = tid; idx < 32 * NC; idx += kTT) {
      const int row = idx / NC, c = idx - row * NC;
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (row < nj) v = *(const f32x4*)(db + (int64_t)(j0 + row) * E + 4 * c);
      *(f32x4*)(DB + row * ES + 4 * c) = v;
    }
  };

  // ---- sweep 1: cosines of every block, pooled kernels ---------------------------------------------------------
  const int rg = tid >> 6, tg8 = (tid >> 3) & 7, ks = tid & 7;    // 8 row groups x 8 token groups x 8 K slices
  const int TQ = (Q + 7) >> 3;                         // query tokens per thread of the cosine tile (<= 4)
  for (int j0 = 0; j0 < D; j0 += 32) {
    const int nj = D - j0 < 32 ? D - j0 : 32;
    load_block(j0, nj);
    __syncthreads();
    {  // row norms and masks: sixteen threads per row
      const int row = tid >> 4, sub = tid & 15;
      float ss = 0.0f;
      for (int c = sub; c < NC; c += 16) {
        const f32x4 v = *(const f32x4*)(DB + row * ES + 4 * c);
        ss += dot4(v, v);
      }
      ss += __shfl_xor(ss, 1, 64);
      ss += __shfl_xor(ss, 2, 64);
      ss += __shfl_xor(ss, 4, 64);
      ss += __shfl_xor(ss, 8, 64);
      if (sub == 0) {
        const float n = sqrtf(ss);
        const bool real = row < nj && mask_bit(a.dm, pair, dwords, j0 + row, D);
        float gate = 1.0f;
        if (GATE) gate = row < nj ? fmaxf(a.dw[pair * D + j0 + row], 0.0f) : 0.0f;
        RD[j0 + row] = row < nj ? 1.0f / (n + 1e-13f) : 0.0f;
        ND[j0 + row] = n;
        DMB[j0 + row] = real ? 1.0f : 0.0f;
        DMF[j0 + row] = real ? gate : 0.0f;
      }
    }
    __syncthreads();
    {  // cosine tile: thread = (4 rows, TQ tokens, every 8th 16-byte chunk of E); the eight K slices meet by shuffles
      float acc[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[r][t] = 0.0f;
      for (int c = ks; c < NC; c += 8) {
        f32x4 dv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) dv[r] = *(const f32x4*)(DB + (4 * rg + r) * ES + 4 * c);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (t < TQ) {
            int i = tg8 * TQ + t;
            i = i < Q ? i : Q - 1;
            const f32x4 qv = *(const f32x4*)(QH + i * ES + 4 * c);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r][t] += dot4(dv[r], qv);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[r][t] += __shfl_xor(acc[r][t], 1, 64);
          acc[r][t] += __shfl_xor(acc[r][t], 2, 64);
          acc[r][t] += __shfl_xor(acc[r][t], 4, 64);
        }
      const int row = 4 * rg + (ks & 3);                // lanes ks = 0..3 of the eight write rows 0..3 of the thread tile
      const float rdv = RD[j0 + row];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = tg8 * TQ + t;
        if (ks < 4 && t < TQ && i < Q) {
          const float v = ks == 0 ? acc[0][t] : (ks == 1 ? acc[1][t] : (ks == 2 ? acc[2][t] : acc[3][t]));
          CT[(j0 + row) * QS + i] = v * rdv;
        }
      }
    }
    __syncthreads();
    for (int idx = tid; idx < Q * K; idx += kTT) {      // (i, k) is owned by one thread across the blocks
      const int i = idx / K, k = idx - i * K;
      const float mu = kc[k], c2 = kc[kTK + k];
      float pk = 0.0f;
      for (int jj = 0; jj < 32; ++jj) {
        const float t = CT[(j0 + jj) * QS + i] - mu;
        pk += DMF[j0 + jj] * __builtin_amdgcn_exp2f(t * t * c2);
      }
      PK[i * kTK + k] += pk;
    }
    __syncthreads();
  }

  // ---- A_ik and the parameter gradients of this pair -----------------------------------------------------------
  for (int idx = tid; idx < Q * K; idx += kTT) {
    const int i = idx / K, k = idx - i * K;
    const float pk = PK[i * kTK + k];
    const float al = a.alpha[k];
    const bool live = al * pk >= a.clamp_min;
    A[i * kTK + k] = live ? g * qmf[i] * a.w[k] / pk : 0.0f;
    PK[i * kTK + k] = qmf[i] * __logf(fmaxf(al * pk, a.clamp_min));   // -> grad_w
    SG[i * kTK + k] = live ? qmf[i] * a.w[k] / al : 0.0f;             // -> grad_alpha (scratch: SG is free until sweep 2)
  }
  __syncthreads();
  if (tid < K) {
    float sw = 0.0f, sa = 0.0f;
    for (int i = 0; i < Q; ++i) {
      sw += PK[i * kTK + tid];
      sa += SG[i * kTK + tid];
    }
    a.gw[pair * K + tid] = g * sw;
    a.galpha[pair * K + tid] = g * sa;
  }
  __syncthreads();

  // ---- sweep 2: G per block, grad_d (complete per block), grad_q (register accumulators across the blocks) -------
  const int TG = QS >> 2;                               // groups of four query tokens
  f32x4 accq[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int t = 0; t < 4; ++t) accq[s][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  for (int j0 = 0; j0 < D; j0 += 32) {
    const int nj = D - j0 < 32 ? D - j0 : 32;
    load_block(j0, nj);
    for (int idx = tid; idx < 32 * QS; idx += kTT) {
      const int jj = idx / QS, i = idx - jj * QS;
      float gs = 0.0f, sg = 0.0f;
      if (i < Q && DMB[j0 + jj] != 0.0f) {
        const float c = CT[(j0 + jj) * QS + i];
        for (int k = 0; k < K; ++k) {
          const float t = c - kc[k];
          const float ae = A[i * kTK + k] * __builtin_amdgcn_exp2f(t * t * kc[kTK + k]);
          sg += ae;
          gs -= ae * t * kc[2 * kTK + k];
        }
      }
      const float G = gs * DMF[j0 + jj];
      GJ[jj * QS + i] = G;
      GI[i * 32 + jj] = G;
      if (GATE) SG[jj * QS + i] = sg;
    }
    __syncthreads();
    if (tid < 32) {                                     // sum_i G c of every position of the block
      float s = 0.0f;
      for (int i = 0; i < Q; ++i) s += GJ[tid * QS + i] * CT[(j0 + tid) * QS + i];
      td[tid] = s;
    } else if (tid < 64) {                              // sum_j G c of every query token, over all blocks
      const int i = tid - 32;
      if (i < Q) {
        float s = 0.0f;
        for (int jj = 0; jj < 32; ++jj) s += GI[i * 32 + jj] * CT[(j0 + jj) * QS + i];
        sq[i] += s;
      }
    } else if (GATE && tid < 96) {                      // gate gradient: sum_ik A_ik e_ijk on real tokens
      const int jj = tid - 64;
      if (jj < nj && a.gdw) {
        float s = 0.0f;
        for (int i = 0; i < Q; ++i) s += SG[jj * QS + i];
        a.gdw[pair * D + j0 + jj] = s;
      }
    }
    __syncthreads();
    // grad_d of the block: item = (4 rows, one 16-byte chunk of E)
    for (int it = tid; it < 8 * NC; it += kTT) {
      const int rgp = it / NC, c = it - rgp * NC;
      f32x4 acc[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      for (int i = 0; i < Q; ++i) {
        const f32x4 qv = *(const f32x4*)(QH + i * ES + 4 * c);
        const f32x4 g4 = *(const f32x4*)(GI + i * 32 + 4 * rgp);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += qv * g4[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * rgp + r;
        if (row < nj) {
          const f32x4 x = *(const f32x4*)(DB + row * ES + 4 * c);
          const float n = ND[j0 + row];
          const float self = n > 0.0f ? td[row] / n : 0.0f;
          *(f32x4*)(gd + (int64_t)(j0 + row) * E + 4 * c) = (acc[r] - x * self) * RD[j0 + row];
        }
      }
    }
    // grad_q: item = (4 query tokens, one 16-byte chunk of E), accumulated over the blocks
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int it = tid + kTT * s;
      if (it < TG * NC) {
        const int tg = it / NC, c = it - tg * NC;
        for (int jj = 0; jj < 32; ++jj) {
          const f32x4 dv = *(const f32x4*)(DB + jj * ES + 4 * c) * RD[j0 + jj];
          const f32x4 g4 = *(const f32x4*)(GJ + jj * QS + 4 * tg);
#pragma unroll
          for (int t = 0; t < 4; ++t) accq[s][t] += dv * g4[t];
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int it = tid + kTT * s;
    if (it < TG * NC) {
      const int tg = it / NC, c = it - tg * NC;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = 4 * tg + t;
        if (i < Q) {
          const f32x4 qv = *(const f32x4*)(QH + i * ES + 4 * c);       // q_i / |q_i| (to 1e-13)
          const float self = nq[i] > 0.0f ? sq[i] : 0.0f;
          *(f32x4*)(gq + (int64_t)i * E + 4 * c) = (accq[s][t] - qv * self) * rq[i];
        }
      }
    }
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
  // the tiled kernel whenever its tiles fit (every shape the reference's configs train at); the per-element kernel otherwise
  {
    const int Dpad = (D + 31) & ~31, QS = (Q + 3) & ~3, NC = E >> 2;
    const size_t tl = kp_bwd_tiled_lds_bytes(Q, E, Dpad);
    if (Q <= 32 && !(E & 3) && K <= kTK && (QS >> 2) * NC <= 2 * kTT && tl <= 150 * 1024 &&
        !(((uintptr_t)q | (uintptr_t)d | (uintptr_t)grad_q | (uintptr_t)grad_d) & 15) && !env().kp_bwd_untiled) {
      KpBwdArgs a{};
      a.q = (const float*)q; a.d = (const float*)d; a.mu = mu; a.sigma = sigma; a.alpha = alpha; a.w = w; a.go = grad_out;
      a.dw = d_gate; a.gdw = grad_gate; a.clamp_min = clamp_min;
      a.gq = grad_q; a.gd = grad_d; a.galpha = grad_alpha; a.gw = grad_w; a.n_pairs = n_pairs; a.Q = Q; a.D = D; a.E = E; a.K = K;
      char* ws = (char*)workspace;
      size_t left = workspace ? workspace_bytes : 0;
      if (int e = resolve_mask(q_mask, q_mask_kind, n_pairs, Q, &ws, &left, stream, &a.qm)) return e;
      if (int e = resolve_mask(d_mask, d_mask_kind, n_pairs, D, &ws, &left, stream, &a.dm)) return e;
      if (d_gate) {
        if (tl > 64 * 1024)
          (void)hipFuncSetAttribute((const void*)kernel_pool_bwd_tiled_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl);
        hipLaunchKernelGGL(kernel_pool_bwd_tiled_kernel<true>, dim3((unsigned)n_pairs), dim3(kTT), tl, stream, a, Dpad);
      } else {
        if (tl > 64 * 1024)
          (void)hipFuncSetAttribute((const void*)kernel_pool_bwd_tiled_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl);
        hipLaunchKernelGGL(kernel_pool_bwd_tiled_kernel<false>, dim3((unsigned)n_pairs), dim3(kTT), tl, stream, a, Dpad);
      }
      return check_launch("kernel_pool_bwd_tiled_kernel");
    }
  }
  // document tile: the whole document when it fits 150 KiB of LDS, else the largest multiple of 32 positions that does
  int DT = D;
  if (kp_bwd_lds_bytes(Q, DT) > 150 * 1024) {
    const size_t fixed = ((size_t)4 * Q * kBK + 5 * (size_t)Q) * 4;
    if (fixed + (size_t)(2 * Q + 5) * 32 * 4 > 150 * 1024)
      return set_error(MM_EUNSUPPORTED, "kernel_pool_bwd: Q = %d query tokens exceed the LDS tile", Q);
    DT = (int)((150 * 1024 - fixed) / ((size_t)(2 * Q + 5) * 4)) & ~31;
  }
  const size_t lds = kp_bwd_lds_bytes(Q, DT);
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
  hipLaunchKernelGGL(kernel_pool_bwd_kernel, dim3((unsigned)n_pairs), dim3(256), lds, stream, a, DT);
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
