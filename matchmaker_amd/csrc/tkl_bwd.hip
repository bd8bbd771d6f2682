// Backward of mm_tkl_fwd (training path: train.py:503-524 through sigir20_tkl.py:180-286).
//
// The document score is sum_j chunk_scoring_j * window_score[idx_j] over the 15 indices the region search picks
// (three arg-max regions x {0, -1, +1, -2, +2}, :257-286).  The index choice is piecewise constant, so the exact
// gradient only involves those windows: 15 x 30 document tokens per document instead of up to 2,000.  One workgroup
// per document repeats the region search on the forward's window scores and, window by window,
//
//   c_it   = <q_i, d_t> / ((|q_i| + 1e-13)(|d_t| + 1e-13))                       :184   (t = 30 window positions)
//   e_itk  = m_t exp(-(c_it - mu_k)^2 / (2 sigma_k^2))                           :192-194
//   pk_ik  = sum_t e_itk,  len_i = #{t : sum_k e_itk != 0}                        :210-211
//   sat_ik = s1_i max(pk_ik, 1e-10)^{s2_i} - s3_i   with (s1, 1/s2, s3) = Linear_{1,2,3}(LayerNorm_2([emb . q_i, len_i]))
//            ("embedding", :224-234)   or   log(max(pk_ik km_k, 1e-10))   ("log", :245-246)
//   w_j    = sum_i qmask_i [len_i > 0] sum_k dense_k sat_ik                        :248-252
//
// recomputes that chain in fp32 and differentiates it: gradients w.r.t. the contextualised query, the contextualised
// chunk rows of the selected windows, and every trainable scoring parameter (dense, chunk_scoring, the three saturation
// layers, the 2-element LayerNorm, sat_emb_reduce1 / kernel_mult).  Exact zeros are constants as in the reference
// (:257, :282: a window whose score is 0 is rewritten to -9900 and contributes nothing).
// Per-document parameter gradients are written as rows [B, MM_TKL_NPARAMS] (summed on the host: deterministic);
// chunk-row gradients are accumulated window after window by one workgroup (no atomics).
// Correctness path: training batches are tens of documents.
#include "mm_internal.h"

namespace mm {

constexpr int kBwdT = 30;      // positions per window
constexpr int kBwdQ = 32;      // query tokens held in LDS
constexpr int kBwdThreads = 256;

struct TklBwdArgs {
  const float* q_ctx;       // [B, Q, E]
  const float* chunks;      // [P, 50, E]
  const float* chunk_mask;  // [P, 50]
  const int32_t* slot2p;    // [B, C]: (packed chunk << 2 | x) or < 0
  const float* q_mask;      // [B, Q]
  const float* prm;
  const float* win;         // [B, W] forward window scores (0 = empty window)
  const float* go;          // [B]
  float* gq;                // [B, Q, E]
  float* gchunks;           // [P, 50, E] (zero-initialised by the caller)
  float* gprm;              // [B, NP]
  int C, Q, E, W, NP, sat;
};

__global__ void __launch_bounds__(kBwdThreads) tkl_bwd_kernel(const TklBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int C = a.C, Q = a.Q, E = a.E, W = a.W;
  const int Wp = W < 3 ? 3 : W;
  const float* prm = a.prm;
  const float* sp = prm + TklParams::sat();

  // ---- LDS map ----------------------------------------------------------------------------------------------------
  float* orig = (float*)smem;                  // [Wp] window scores with 0 -> -9900
  float* work = orig + Wp;                     // [Wp]
  float* cosm = work + Wp;                     // [kBwdQ][kBwdT]
  float* Gm = cosm + kBwdQ * kBwdT;            // [kBwdQ][kBwdT]
  float* pk = Gm + kBwdQ * kBwdT;              // [kBwdQ][kK]
  float* dpk = pk + kBwdQ * kK;                // [kBwdQ][kK]
  float* red = dpk + kBwdQ * kK;               // [kBwdQ][40] per-token parameter-gradient partials (this document)
  float* rq = red + kBwdQ * 40;                // [kBwdQ] 1/(|q|+tiny)
  float* nq = rq + kBwdQ;                      // |q|
  float* embv = nq + kBwdQ;                    // emb . q_i
  float* lens = embv + kBwdQ;                  // window lengths
  float* vals = lens + kBwdQ;                  // per-token window value
  float* dev = vals + kBwdQ;                   // d loss / d (emb . q_i), summed over the windows
  float* sq = dev + kBwdQ;                     // sum_t G c
  float* rd = sq + kBwdQ;                      // [kBwdT]
  float* nd = rd + kBwdT + 2;
  float* mt = nd + kBwdT + 2;                  // mask x presence of position t
  float* td = mt + kBwdT + 2;                  // sum_i G c
  float* csg = td + kBwdT + 2;                 // [16] chunk_scoring gradients
  int* prow = (int*)(csg + 16);                // [kBwdT] flat row (p * 50 + row) of position t, or -1
  __shared__ float rv[4];
  __shared__ int ri[4];
  __shared__ int top_s[3];

  const float* qb = a.q_ctx + (int64_t)b * Q * E;
  float* gq = a.gq + (int64_t)b * Q * E;
  const float g = a.go[b];

  // ---- query norms, emb . q_i; zero the accumulators -----------------------------------------------------------------
  for (int i = wv; i < Q; i += 4) {
    const float* x = qb + (int64_t)i * E;
    float ss = 0.0f, se = 0.0f;
    for (int e = lane; e < E; e += 64) {
      ss += x[e] * x[e];
      se += x[e] * prm[TklParams::emb() + e];
    }
    ss = wave_sum(ss);
    se = wave_sum(se);
    if (lane == 0) {
      const float n = sqrtf(ss);
      nq[i] = n;
      rq[i] = 1.0f / (n + 1e-13f);
      embv[i] = se;
      dev[i] = 0.0f;
    }
  }
  for (int idx = tid; idx < kBwdQ * 40; idx += kBwdThreads) red[idx] = 0.0f;
  if (tid < 16) csg[tid] = 0.0f;
  for (int idx = tid; idx < Q * E; idx += kBwdThreads) gq[idx] = 0.0f;

  // ---- region search on the forward's window scores (:257, :268-273; ties -> lowest index, as tkl_region_kernel) -------
  for (int w = tid; w < Wp; w += kBwdThreads) {
    float s = w < W ? a.win[(int64_t)b * W + w] : 0.0f;
    if (s == 0.0f) s = -9900.0f;
    orig[w] = s;
    work[w] = s;
  }
  __syncthreads();
  for (int c = 0; c < 3; ++c) {
    float bv = -__builtin_huge_valf();
    int bi = 0x7fffffff;
    for (int w = tid; w < Wp; w += kBwdThreads) {
      const float v = work[w];
      if (v > bv) { bv = v; bi = w; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { rv[wv] = bv; ri[wv] = bi; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float ov = rv[k];
      const int oi = ri[k];
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (tid == 0) top_s[c] = bi;
    __syncthreads();
    for (int w = tid; w < Wp; w += kBwdThreads) {
      const int dlt = w > bi ? w - bi : bi - w;
      if (dlt < 15) work[w] = -10001.0f - (float)c;
    }
    __syncthreads();
  }

  // per-thread accumulators of the vector-valued parameter gradient (sat_emb_reduce1.weight[e], e = tid + 256 n)
  const int offs[5] = {0, -1, 1, -2, 2};
  for (int j = 0; j < 15; ++j) {
    int idx = top_s[j % 3] + offs[j / 3];                       // :276 order: peaks, -1, +1, -2, +2
    idx = idx < 0 ? 0 : (idx >= Wp ? Wp - 1 : idx);            // :277-278
    const float wfwd = orig[idx];
    const float cs = prm[TklParams::chunk_scoring() + j];
    if (wfwd <= -9900.0f) continue;                             // :282 an empty window is the constant 0 (uniform branch)
    // ---- the window's 30 positions -> chunk rows -----------------------------------------------------------------------
    if (tid < kBwdT) {
      const int pos = 2 * idx + tid;
      int flat = -1;
      float m = 0.0f;
      if (pos < C * 40) {
        const int c = pos / 40;
        const int info = a.slot2p[(int64_t)b * C + c];
        if (info >= 0) {
          flat = (info >> 2) * 50 + 5 + (pos - 40 * c);
          m = a.chunk_mask[flat] != 0.0f ? 1.0f : 0.0f;
        }
      }
      prow[tid] = flat;
      mt[tid] = m;
    }
    __syncthreads();
    for (int t = wv; t < kBwdT; t += 4) {                       // document-token norms
      const int flat = prow[t];
      float ss = 0.0f;
      if (flat >= 0) {
        const float* x = a.chunks + (int64_t)flat * E;
        for (int e = lane; e < E; e += 64) ss += x[e] * x[e];
      }
      ss = wave_sum(ss);
      if (lane == 0) {
        const float n = sqrtf(ss);
        nd[t] = n;
        rd[t] = 1.0f / (n + 1e-13f);
      }
    }
    __syncthreads();
    for (int e2 = tid; e2 < Q * kBwdT; e2 += kBwdThreads) {     // cosines (same factor order as the forward)
      const int i = e2 / kBwdT, t = e2 - i * kBwdT;
      const int flat = prow[t];
      float dot = 0.0f;
      if (flat >= 0) {
        const float* x = qb + (int64_t)i * E;
        const float* y = a.chunks + (int64_t)flat * E;
        for (int e = 0; e < E; ++e) dot += x[e] * y[e];
      }
      cosm[i * kBwdT + t] = (dot * rq[i]) * rd[t];
    }
    __syncthreads();
    for (int e2 = tid; e2 < Q * kK; e2 += kBwdThreads) {        // pooled kernels of the window
      const int i = e2 / kK, k = e2 - i * kK;
      const float mu = prm[TklParams::mu() + k], sg = prm[TklParams::sigma() + k];
      const float c2 = -1.0f / (2.0f * sg * sg);
      float s = 0.0f;
      for (int t = 0; t < kBwdT; ++t) {
        const float d = cosm[i * kBwdT + t] - mu;
        s += mt[t] * __expf(d * d * c2);
      }
      pk[i * kK + k] = s;
    }
    if (tid < Q) {                                              // window lengths (:210)
      int n = 0;
      for (int t = 0; t < kBwdT; ++t) {
        float any = 0.0f;
        for (int k = 0; k < kK; ++k) {
          const float sg = prm[TklParams::sigma() + k];
          const float d = cosm[tid * kBwdT + t] - prm[TklParams::mu() + k];
          any += mt[t] * __expf(-d * d / (2.0f * sg * sg));
        }
        n += any != 0.0f ? 1 : 0;
      }
      lens[tid] = (float)n;
    }
    __syncthreads();
    // ---- saturation forward + backward per query token -------------------------------------------------------------
    const float gw = g * cs;                                     // d loss / d w_j
    if (tid < Q) {
      const int i = tid;
      const float len = lens[i];
      const float f = a.q_mask[(int64_t)b * Q + i] * (len > 0.0f ? 1.0f : 0.0f);      // :248
      float* rr = red + i * 40;      // [0..10] dense, [11..21] kernel_mult, [22..34] saturation block (13)
      float val = 0.0f;
      if (a.sat == MM_TKL_SAT_EMBEDDING) {
        const float x0 = embv[i], x1 = len;
        const float mean = (x0 + x1) * 0.5f;
        const float d0 = x0 - mean, d1 = x1 - mean;
        const float rstd = 1.0f / sqrtf((d0 * d0 + d1 * d1) * 0.5f + 1e-5f);
        const float xh0 = d0 * rstd, xh1 = d1 * rstd;
        const float n0 = xh0 * sp[9] + sp[11], n1 = xh1 * sp[10] + sp[12];
        const float s1 = n0 * sp[0] + n1 * sp[1] + sp[2];
        const float u = n0 * sp[3] + n1 * sp[4] + sp[5];
        const float s2 = 1.0f / u;
        const float s3 = n0 * sp[6] + n1 * sp[7] + sp[8];
        float ds1 = 0.0f, ds2 = 0.0f, ds3 = 0.0f;
        for (int k = 0; k < kK; ++k) {
          const float p = pk[i * kK + k];
          const float x = fmaxf(p, 1e-10f);
          const float lx = __logf(x);
          const float xp = __expf(s2 * lx);
          const float sat = s1 * xp - s3;
          const float dk = prm[TklParams::dense() + k];
          val += dk * (sat * f);
          const float dsat = gw * f * dk;
          rr[k] += gw * f * sat;                                  // d dense_k
          ds1 += dsat * xp;
          ds3 -= dsat;
          ds2 += dsat * s1 * xp * lx;
          dpk[i * kK + k] = p >= 1e-10f ? dsat * s1 * s2 * xp / x : 0.0f;
        }
        const float du = -ds2 * s2 * s2;
        const float dn0 = ds1 * sp[0] + du * sp[3] + ds3 * sp[6];
        const float dn1 = ds1 * sp[1] + du * sp[4] + ds3 * sp[7];
        rr[22 + 0] += ds1 * n0; rr[22 + 1] += ds1 * n1; rr[22 + 2] += ds1;
        rr[22 + 3] += du * n0;  rr[22 + 4] += du * n1;  rr[22 + 5] += du;
        rr[22 + 6] += ds3 * n0; rr[22 + 7] += ds3 * n1; rr[22 + 8] += ds3;
        rr[22 + 9] += dn0 * xh0; rr[22 + 10] += dn1 * xh1;       // LayerNorm weight
        rr[22 + 11] += dn0;      rr[22 + 12] += dn1;             // LayerNorm bias
        const float dx0h = dn0 * sp[9], dx1h = dn1 * sp[10];
        const float m1 = (dx0h + dx1h) * 0.5f, m2 = (dx0h * xh0 + dx1h * xh1) * 0.5f;
        dev[i] += rstd * (dx0h - m1 - xh0 * m2);                 // d loss / d (emb . q_i); the length carries no gradient
      } else {
        for (int k = 0; k < kK; ++k) {
          const float p = pk[i * kK + k];
          const float km = prm[TklParams::kmult() + k];
          const bool live = p * km >= 1e-10f;
          const float sat = __logf(fmaxf(p * km, 1e-10f));
          const float dk = prm[TklParams::dense() + k];
          val += dk * (sat * f);
          const float dsat = gw * f * dk;
          rr[k] += gw * f * sat;
          rr[11 + k] += live ? dsat / km : 0.0f;
          dpk[i * kK + k] = live ? dsat / p : 0.0f;
        }
      }
      vals[i] = val;
    }
    __syncthreads();
    if (tid == 0) {                                             // d chunk_scoring_j = g * w_j (:249 sum in index order)
      float wj = 0.0f;
      for (int i = 0; i < Q; ++i) wj += vals[i];
      csg[j] += g * wj;
    }
    // ---- G = d loss / d c, its row / column sums --------------------------------------------------------------------
    for (int e2 = tid; e2 < Q * kBwdT; e2 += kBwdThreads) {
      const int i = e2 / kBwdT, t = e2 - i * kBwdT;
      const float c = cosm[e2];
      float s = 0.0f;
      if (mt[t] != 0.0f) {
        for (int k = 0; k < kK; ++k) {
          const float sg = prm[TklParams::sigma() + k];
          const float d = c - prm[TklParams::mu() + k];
          const float inv = 1.0f / (sg * sg);
          s += dpk[i * kK + k] * __expf(-0.5f * d * d * inv) * (-d * inv);
        }
      }
      Gm[e2] = s;
    }
    __syncthreads();
    if (tid < Q) {
      float s = 0.0f;
      for (int t = 0; t < kBwdT; ++t) s += Gm[tid * kBwdT + t] * cosm[tid * kBwdT + t];
      sq[tid] = s;
    }
    if (tid >= 64 && tid < 64 + kBwdT) {
      const int t = tid - 64;
      float s = 0.0f;
      for (int i = 0; i < Q; ++i) s += Gm[i * kBwdT + t] * cosm[i * kBwdT + t];
      td[t] = s;
    }
    __syncthreads();
    // ---- gradients of the vectors: grad_q accumulates over the windows, chunk rows over overlapping windows ---------------
    for (int e2 = tid; e2 < Q * E; e2 += kBwdThreads) {
      const int i = e2 / E, e = e2 - i * E;
      float s = 0.0f;
      for (int t = 0; t < kBwdT; ++t) {
        const int flat = prow[t];
        if (flat >= 0) s += Gm[i * kBwdT + t] * rd[t] * a.chunks[(int64_t)flat * E + e];
      }
      const float self = nq[i] > 0.0f ? sq[i] * qb[e2] / nq[i] : 0.0f;
      gq[e2] += rq[i] * (s - self);
    }
    for (int e2 = tid; e2 < kBwdT * E; e2 += kBwdThreads) {
      const int t = e2 / E, e = e2 - t * E;
      const int flat = prow[t];
      if (flat < 0 || mt[t] == 0.0f) continue;
      float s = 0.0f;
      for (int i = 0; i < Q; ++i) s += Gm[i * kBwdT + t] * rq[i] * qb[(int64_t)i * E + e];
      const float x = a.chunks[(int64_t)flat * E + e];
      const float self = nd[t] > 0.0f ? td[t] * x / nd[t] : 0.0f;
      a.gchunks[(int64_t)flat * E + e] += rd[t] * (s - self);
    }
    __syncthreads();   // the next window may touch the same chunk rows and reuses the LDS tiles
  }

  // ---- emb . q path: grad_q += dev_i * emb_w, d emb_w = sum_i dev_i q_i; parameter rows of this document ----------------
  float* gp = a.gprm + (int64_t)b * a.NP;
  if (a.sat == MM_TKL_SAT_EMBEDDING) {
    for (int e2 = tid; e2 < Q * E; e2 += kBwdThreads) {
      const int i = e2 / E, e = e2 - i * E;
      gq[e2] += dev[i] * prm[TklParams::emb() + e];
    }
    for (int e = tid; e < E; e += kBwdThreads) {
      float s = 0.0f;
      for (int i = 0; i < Q; ++i) s += dev[i] * qb[(int64_t)i * E + e];
      gp[TklParams::emb() + e] = s;
    }
  } else {
    for (int e = tid; e < E; e += kBwdThreads) gp[TklParams::emb() + e] = 0.0f;
  }
  for (int k = tid; k < 2 * kK; k += kBwdThreads) gp[k] = 0.0f;                 // mu, sigma are not trained
  if (tid < 35) {                                                              // sum over the query tokens in index order
    float s = 0.0f;
    for (int i = 0; i < Q; ++i) s += red[i * 40 + tid];
    const int dst = tid < 11 ? TklParams::dense() + tid : (tid < 22 ? TklParams::kmult() + (tid - 11) : TklParams::sat() + (tid - 22));
    gp[dst] = s;
  }
  if (tid < 15) gp[TklParams::chunk_scoring() + tid] = csg[tid];
}

// ---------------------------------------------------------------------------------------------
// Tiled variant (round 4).  tkl_bwd_kernel above takes every dot product straight from global memory, one dependent load
// per FMA: 4.6 ms per DOCUMENT (bench.py extra.train_step: 22 ms for 2,048 documents, 24 x the forward).  Here the
// normalised query tile and the 30 rows of the current window are staged in LDS and the three small products of a window
// (cosines, chunk-row gradients, query gradient) are register-blocked exactly as in kernel_pool_bwd_tiled_kernel
// (kernel_pool_bwd.hip); the query gradient accumulates in registers over the 15 windows.  The scalar chain between them
// (pooled kernels, lengths, saturation forward + backward, parameter gradients) is the code above, re-indexed.
// 512 threads, Q <= 32, E <= 384 (16-byte rows); other shapes take the kernel above.
// ---------------------------------------------------------------------------------------------
constexpr int kTT = 512;

__host__ __device__ inline size_t tkl_bwd_tiled_lds_bytes(int Wp, int Q, int E) {
  const int ES = E + 4, QS = (Q + 3) & ~3;
  return ((size_t)2 * Wp + (size_t)Q * ES + 32 * (size_t)ES + 3 * 32 * (size_t)QS + 2 * kBwdQ * kK + kBwdQ * 40 + 8 * kBwdQ + 4 * 32 +
          16 + 32) * 4 + 64;
}

__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

__global__ void __launch_bounds__(kTT) tkl_bwd_tiled_kernel(const TklBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int C = a.C, Q = a.Q, E = a.E, W = a.W;
  const int Wp = W < 3 ? 3 : W;
  const int ES = E + 4, QS = (Q + 3) & ~3, NC = E >> 2;
  const float* prm = a.prm;
  const float* sp = prm + TklParams::sat();

  float* QH = (float*)smem;                    // [Q][ES]  q_i / (|q_i| + tiny)
  float* DB = QH + Q * ES;                     // [32][ES] the window's rows (raw; rows 30, 31 and absent rows are zeros)
  float* CT = DB + 32 * ES;                    // [32][QS] cosines, [position][token]
  float* GJ = CT + 32 * QS;                    // [32][QS] d loss / d c
  float* GI = GJ + 32 * QS;                    // [QS][32]
  float* pk = GI + QS * 32;                    // [kBwdQ][kK]
  float* dpk = pk + kBwdQ * kK;                // [kBwdQ][kK]
  float* red = dpk + kBwdQ * kK;               // [kBwdQ][40]
  float* rq = red + kBwdQ * 40;                // [kBwdQ] each:
  float* nq = rq + kBwdQ;
  float* embv = nq + kBwdQ;
  float* lens = embv + kBwdQ;
  float* vals = lens + kBwdQ;
  float* dev = vals + kBwdQ;
  float* sq = dev + kBwdQ;                     // sum_t G c of the current window
  float* sqs = sq + kBwdQ;                     // ... summed over the windows
  float* rd = sqs + kBwdQ;                     // [32] each:
  float* nd = rd + 32;
  float* mt = nd + 32;
  float* td = mt + 32;
  float* csg = td + 32;                        // [16]
  int* prow = (int*)(csg + 16);                // [32]
  float* orig = (float*)(prow + 32);           // [Wp]
  float* work = orig + Wp;                     // [Wp]
  __shared__ float rv[kTT / 64];
  __shared__ int ri[kTT / 64];
  __shared__ int top_s[3];

  const float* qb = a.q_ctx + (int64_t)b * Q * E;
  float* gq = a.gq + (int64_t)b * Q * E;
  const float g = a.go[b];

  // ---- query tile: raw rows -> norms, emb . q_i -> normalised in place ----------------------------------------------
  for (int idx = tid; idx < Q * NC; idx += kTT) {
    const int i = idx / NC, c = idx - i * NC;
    *(f32x4*)(QH + i * ES + 4 * c) = *(const f32x4*)(qb + (int64_t)i * E + 4 * c);
  }
  for (int idx = tid; idx < kBwdQ * 40; idx += kTT) red[idx] = 0.0f;
  if (tid < 16) csg[tid] = 0.0f;
  if (tid < kBwdQ) { dev[tid] = 0.0f; sqs[tid] = 0.0f; }
  for (int w = tid; w < Wp; w += kTT) {
    float s = w < W ? a.win[(int64_t)b * W + w] : 0.0f;
    if (s == 0.0f) s = -9900.0f;
    orig[w] = s;
    work[w] = s;
  }
  __syncthreads();
  {
    const int i = tid >> 4, sub = tid & 15;
    float ss = 0.0f, se = 0.0f;
    if (i < Q)
      for (int c = sub; c < NC; c += 16) {
        const f32x4 v = *(const f32x4*)(QH + i * ES + 4 * c);
        const f32x4 w4 = *(const f32x4*)(prm + TklParams::emb() + 4 * c);
        ss += dot4(v, v);
        se += dot4(v, w4);
      }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      ss += __shfl_xor(ss, o, 64);
      se += __shfl_xor(se, o, 64);
    }
    const float n = sqrtf(ss), r = 1.0f / (n + 1e-13f);
    if (i < Q) {
      for (int c = sub; c < NC; c += 16) {
        f32x4* p = (f32x4*)(QH + i * ES + 4 * c);
        *p = *p * r;
      }
      if (sub == 0) { nq[i] = n; rq[i] = r; embv[i] = se; }
    }
  }
  // ---- region search (as above) ---------------------------------------------------------------------------------------
  for (int c = 0; c < 3; ++c) {
    float bv = -__builtin_huge_valf();
    int bi = 0x7fffffff;
    for (int w = tid; w < Wp; w += kTT) {
      const float v = work[w];
      if (v > bv) { bv = v; bi = w; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { rv[wv] = bv; ri[wv] = bi; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kTT / 64; ++k) {
      const float ov = rv[k];
      const int oi = ri[k];
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (tid == 0) top_s[c] = bi;
    __syncthreads();
    for (int w = tid; w < Wp; w += kTT) {
      const int dlt = w > bi ? w - bi : bi - w;
      if (dlt < 15) work[w] = -10001.0f - (float)c;
    }
    __syncthreads();
  }

  const int rg = tid >> 6, tg8 = (tid >> 3) & 7, ks = tid & 7;      // cosine tile: 8 row groups x 8 token groups x 8 K slices
  const int TQ = (Q + 7) >> 3;
  const int TG = QS >> 2;
  f32x4 accq[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int t = 0; t < 4; ++t) accq[s][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const int offs[5] = {0, -1, 1, -2, 2};
  for (int j = 0; j < 15; ++j) {
    int idx = top_s[j % 3] + offs[j / 3];                       // :276 order: peaks, -1, +1, -2, +2
    idx = idx < 0 ? 0 : (idx >= Wp ? Wp - 1 : idx);            // :277-278
    const float wfwd = orig[idx];
    const float cs = prm[TklParams::chunk_scoring() + j];
    if (wfwd <= -9900.0f) continue;                             // :282 an empty window is the constant 0 (uniform branch)
    if (tid < 32) {                                             // the window's 30 positions -> chunk rows
      const int pos = 2 * idx + tid;
      int flat = -1;
      float m = 0.0f;
      if (tid < kBwdT && pos < C * 40) {
        const int c = pos / 40;
        const int info = a.slot2p[(int64_t)b * C + c];
        if (info >= 0) {
          flat = (info >> 2) * 50 + 5 + (pos - 40 * c);
          m = a.chunk_mask[flat] != 0.0f ? 1.0f : 0.0f;
        }
      }
      prow[tid] = flat;
      mt[tid] = m;
    }
    __syncthreads();
    for (int e2 = tid; e2 < 32 * NC; e2 += kTT) {               // rows -> LDS
      const int row = e2 / NC, c = e2 - row * NC;
      const int flat = prow[row];
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (flat >= 0) v = *(const f32x4*)(a.chunks + (int64_t)flat * E + 4 * c);
      *(f32x4*)(DB + row * ES + 4 * c) = v;
    }
    __syncthreads();
    {  // row norms: sixteen threads per row
      const int row = tid >> 4, sub = tid & 15;
      float ss = 0.0f;
      for (int c = sub; c < NC; c += 16) {
        const f32x4 v = *(const f32x4*)(DB + row * ES + 4 * c);
        ss += dot4(v, v);
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) ss += __shfl_xor(ss, o, 64);
      if (sub == 0) {
        const float n = sqrtf(ss);
        nd[row] = n;
        rd[row] = 1.0f / (n + 1e-13f);
      }
    }
    __syncthreads();
    {  // cosines: thread = (4 rows, TQ tokens, every 8th chunk of E)
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
      const int row = 4 * rg + (ks & 3);
      const float rdv = rd[row];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = tg8 * TQ + t;
        if (ks < 4 && t < TQ && i < Q) {
          const float v = (ks & 3) == 0 ? acc[0][t] : ((ks & 3) == 1 ? acc[1][t] : ((ks & 3) == 2 ? acc[2][t] : acc[3][t]));
          CT[row * QS + i] = v * rdv;
        }
      }
    }
    __syncthreads();
    for (int e2 = tid; e2 < Q * kK; e2 += kTT) {                // pooled kernels of the window
      const int i = e2 / kK, k = e2 - i * kK;
      const float mu = prm[TklParams::mu() + k], sg = prm[TklParams::sigma() + k];
      const float c2 = -1.0f / (2.0f * sg * sg);
      float s = 0.0f;
      for (int t = 0; t < kBwdT; ++t) {
        const float d = CT[t * QS + i] - mu;
        s += mt[t] * __expf(d * d * c2);
      }
      pk[i * kK + k] = s;
    }
    if (tid >= 448 && tid - 448 < Q) {                          // window lengths (:210), on the last wavefront
      const int i = tid - 448;
      int n = 0;
      for (int t = 0; t < kBwdT; ++t) {
        float any = 0.0f;
        for (int k = 0; k < kK; ++k) {
          const float sg = prm[TklParams::sigma() + k];
          const float d = CT[t * QS + i] - prm[TklParams::mu() + k];
          any += mt[t] * __expf(-d * d / (2.0f * sg * sg));
        }
        n += any != 0.0f ? 1 : 0;
      }
      lens[i] = (float)n;
    }
    __syncthreads();
    // ---- saturation forward + backward per query token (as above) -------------------------------------------------------
    const float gw = g * cs;                                     // d loss / d w_j
    if (tid < Q) {
      const int i = tid;
      const float len = lens[i];
      const float f = a.q_mask[(int64_t)b * Q + i] * (len > 0.0f ? 1.0f : 0.0f);      // :248
      float* rr = red + i * 40;      // [0..10] dense, [11..21] kernel_mult, [22..34] saturation block (13)
      float val = 0.0f;
      if (a.sat == MM_TKL_SAT_EMBEDDING) {
        const float x0 = embv[i], x1 = len;
        const float mean = (x0 + x1) * 0.5f;
        const float d0 = x0 - mean, d1 = x1 - mean;
        const float rstd = 1.0f / sqrtf((d0 * d0 + d1 * d1) * 0.5f + 1e-5f);
        const float xh0 = d0 * rstd, xh1 = d1 * rstd;
        const float n0 = xh0 * sp[9] + sp[11], n1 = xh1 * sp[10] + sp[12];
        const float s1 = n0 * sp[0] + n1 * sp[1] + sp[2];
        const float u = n0 * sp[3] + n1 * sp[4] + sp[5];
        const float s2 = 1.0f / u;
        const float s3 = n0 * sp[6] + n1 * sp[7] + sp[8];
        float ds1 = 0.0f, ds2 = 0.0f, ds3 = 0.0f;
        for (int k = 0; k < kK; ++k) {
          const float p = pk[i * kK + k];
          const float x = fmaxf(p, 1e-10f);
          const float lx = __logf(x);
          const float xp = __expf(s2 * lx);
          const float sat = s1 * xp - s3;
          const float dk = prm[TklParams::dense() + k];
          val += dk * (sat * f);
          const float dsat = gw * f * dk;
          rr[k] += gw * f * sat;                                  // d dense_k
          ds1 += dsat * xp;
          ds3 -= dsat;
          ds2 += dsat * s1 * xp * lx;
          dpk[i * kK + k] = p >= 1e-10f ? dsat * s1 * s2 * xp / x : 0.0f;
        }
        const float du = -ds2 * s2 * s2;
        const float dn0 = ds1 * sp[0] + du * sp[3] + ds3 * sp[6];
        const float dn1 = ds1 * sp[1] + du * sp[4] + ds3 * sp[7];
        rr[22 + 0] += ds1 * n0; rr[22 + 1] += ds1 * n1; rr[22 + 2] += ds1;
        rr[22 + 3] += du * n0;  rr[22 + 4] += du * n1;  rr[22 + 5] += du;
        rr[22 + 6] += ds3 * n0; rr[22 + 7] += ds3 * n1; rr[22 + 8] += ds3;
        rr[22 + 9] += dn0 * xh0; rr[22 + 10] += dn1 * xh1;       // LayerNorm weight
        rr[22 + 11] += dn0;      rr[22 + 12] += dn1;             // LayerNorm bias
        const float dx0h = dn0 * sp[9], dx1h = dn1 * sp[10];
        const float m1 = (dx0h + dx1h) * 0.5f, m2 = (dx0h * xh0 + dx1h * xh1) * 0.5f;
        dev[i] += rstd * (dx0h - m1 - xh0 * m2);                 // d loss / d (emb . q_i); the length carries no gradient
      } else {
        for (int k = 0; k < kK; ++k) {
          const float p = pk[i * kK + k];
          const float km = prm[TklParams::kmult() + k];
          const bool live = p * km >= 1e-10f;
          const float sat = __logf(fmaxf(p * km, 1e-10f));
          const float dk = prm[TklParams::dense() + k];
          val += dk * (sat * f);
          const float dsat = gw * f * dk;
          rr[k] += gw * f * sat;
          rr[11 + k] += live ? dsat / km : 0.0f;
          dpk[i * kK + k] = live ? dsat / p : 0.0f;
        }
      }
      vals[i] = val;
    }
    __syncthreads();
    if (tid == 0) {                                             // d chunk_scoring_j = g * w_j (:249 sum in index order)
      float wj = 0.0f;
      for (int i = 0; i < Q; ++i) wj += vals[i];
      csg[j] += g * wj;
    }
    for (int e2 = tid; e2 < 32 * QS; e2 += kTT) {               // G = d loss / d c
      const int t = e2 / QS, i = e2 - t * QS;
      float s = 0.0f;
      if (i < Q && mt[t] != 0.0f) {
        const float c = CT[t * QS + i];
        for (int k = 0; k < kK; ++k) {
          const float sg = prm[TklParams::sigma() + k];
          const float d = c - prm[TklParams::mu() + k];
          const float inv = 1.0f / (sg * sg);
          s += dpk[i * kK + k] * __expf(-0.5f * d * d * inv) * (-d * inv);
        }
      }
      GJ[t * QS + i] = s;
      GI[i * 32 + t] = s;
    }
    __syncthreads();
    if (tid < Q) {
      float s = 0.0f;
      for (int t = 0; t < kBwdT; ++t) s += GI[tid * 32 + t] * CT[t * QS + tid];
      sqs[tid] += s;
    } else if (tid >= 64 && tid < 96) {
      const int t = tid - 64;
      float s = 0.0f;
      for (int i = 0; i < Q; ++i) s += GJ[t * QS + i] * CT[t * QS + i];
      td[t] = s;
    }
    __syncthreads();
    // chunk-row gradients: item = (4 rows, one 16-byte chunk of E); overlapping windows accumulate into the same rows, one after
    // the other inside this workgroup
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
        const int flat = prow[row];
        if (flat >= 0 && mt[row] != 0.0f) {
          const f32x4 x = *(const f32x4*)(DB + row * ES + 4 * c);
          const float self = nd[row] > 0.0f ? td[row] / nd[row] : 0.0f;
          f32x4* dst = (f32x4*)(a.gchunks + (int64_t)flat * E + 4 * c);
          *dst = *dst + (acc[r] - x * self) * rd[row];
        }
      }
    }
    // query gradient: item = (4 tokens, one chunk), accumulated over the windows
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int it = tid + kTT * s;
      if (it < TG * NC) {
        const int tg = it / NC, c = it - tg * NC;
        for (int t = 0; t < kBwdT; ++t) {
          const f32x4 dv = *(const f32x4*)(DB + t * ES + 4 * c) * rd[t];
          const f32x4 g4 = *(const f32x4*)(GJ + t * QS + 4 * tg);
#pragma unroll
          for (int u = 0; u < 4; ++u) accq[s][u] += dv * g4[u];
        }
      }
    }
    __syncthreads();   // the next window may touch the same chunk rows and reuses the LDS tiles
  }

  // ---- grad_q = rq (sum_w sum_t G dh - (sum G c) q / |q|) + dev emb_w; parameter rows of this document -----------------
  const bool emb_sat = a.sat == MM_TKL_SAT_EMBEDDING;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int it = tid + kTT * s;
    if (it < TG * NC) {
      const int tg = it / NC, c = it - tg * NC;
      f32x4 ew = {0.0f, 0.0f, 0.0f, 0.0f};
      if (emb_sat) ew = *(const f32x4*)(prm + TklParams::emb() + 4 * c);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = 4 * tg + u;
        if (i < Q) {
          const f32x4 qv = *(const f32x4*)(QH + i * ES + 4 * c);
          const float self = nq[i] > 0.0f ? sqs[i] : 0.0f;
          *(f32x4*)(gq + (int64_t)i * E + 4 * c) = (accq[s][u] - qv * self) * rq[i] + ew * dev[i];
        }
      }
    }
  }
  float* gp = a.gprm + (int64_t)b * a.NP;
  if (emb_sat) {
    for (int e = tid; e < E; e += kTT) {
      float s = 0.0f;
      for (int i = 0; i < Q; ++i) s += dev[i] * QH[i * ES + e] * (nq[i] + 1e-13f);     // q_i = qh_i (|q_i| + tiny)
      gp[TklParams::emb() + e] = s;
    }
  } else {
    for (int e = tid; e < E; e += kTT) gp[TklParams::emb() + e] = 0.0f;
  }
  for (int k = tid; k < 2 * kK; k += kTT) gp[k] = 0.0f;                           // mu, sigma are not trained
  if (tid < 35) {                                                              // sum over the query tokens in index order
    float s = 0.0f;
    for (int i = 0; i < Q; ++i) s += red[i * 40 + tid];
    const int dst = tid < 11 ? TklParams::dense() + tid : (tid < 22 ? TklParams::kmult() + (tid - 11) : TklParams::sat() + (tid - 22));
    gp[dst] = s;
  }
  if (tid < 15) gp[TklParams::chunk_scoring() + tid] = csg[tid];
}

// slot2p for the backward (the forward's preparation kernels live in tkl.hip)
__global__ void __launch_bounds__(256) tkl_bwd_fill_kernel(int32_t* slot2p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) slot2p[i] = -1;
}
__global__ void __launch_bounds__(256) tkl_bwd_slot_kernel(const int32_t* __restrict__ chunk_slot, int64_t P, int64_t BC,
                                                           int32_t* __restrict__ slot2p) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p < P) {
    const int32_t s = chunk_slot[p];
    if (s >= 0 && s < BC) slot2p[s] = (int32_t)(p << 2);
  }
}

}  // namespace mm

using namespace mm;

extern "C" size_t mm_tkl_bwd_workspace_bytes(int64_t B, int C) { return ((size_t)B * C * 4 + 255) & ~(size_t)255; }

extern "C" int mm_tkl_bwd(const void* q_ctx, const void* chunks, const float* chunk_mask, const int32_t* chunk_slot,
                          const float* q_mask, const float* params, const float* win_scores, const float* grad_out,
                          float* grad_q, float* grad_chunks, float* grad_params, int64_t B, int64_t P, int C, int Q, int E,
                          int K, int saturation, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B == 0) return MM_OK;  // an empty batch has empty gradients (as the other operators: nothing to launch)
  if (!q_ctx || !q_mask || !params || !win_scores || !grad_out || !grad_q || !grad_params)
    return set_error(MM_EINVAL, "tkl_bwd: null pointer");
  if (P > 0 && (!chunks || !chunk_mask || !chunk_slot || !grad_chunks)) return set_error(MM_EINVAL, "tkl_bwd: null chunk pointer");
  if (B <= 0 || P < 0 || C <= 0 || Q <= 0 || E <= 0) return set_error(MM_EINVAL, "tkl_bwd: bad shape");
  if (K != kK) return set_error(MM_EUNSUPPORTED, "tkl_bwd: K=%d kernels (only the reference's 11 are instantiated)", K);
  if (Q > kBwdQ) return set_error(MM_EUNSUPPORTED, "tkl_bwd: Q=%d query tokens (max %d)", Q, kBwdQ);
  if (saturation != MM_TKL_SAT_EMBEDDING && saturation != MM_TKL_SAT_LOG)
    return set_error(MM_EUNSUPPORTED, "tkl_bwd: saturation %d", saturation);
  if (P >= (1LL << 29)) return set_error(MM_EUNSUPPORTED, "tkl_bwd: too many packed chunks");
  const size_t need = mm_tkl_bwd_workspace_bytes(B, C);
  if (!workspace || workspace_bytes < need) return set_error(MM_EWORKSPACE, "tkl_bwd: workspace needs %zu bytes", need);
  const int W = ((C * 40 > 30 ? C * 40 : 30) - 30) / 2 + 1;
  const int Wp = W < 3 ? 3 : W;
  int32_t* slot2p = (int32_t*)workspace;
  hipLaunchKernelGGL(tkl_bwd_fill_kernel, dim3((unsigned)((B * (int64_t)C + 255) / 256)), dim3(256), 0, stream, slot2p, B * (int64_t)C);
  if (P > 0) {
    hipLaunchKernelGGL(tkl_bwd_slot_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, chunk_slot, P, B * (int64_t)C, slot2p);
    if (hipMemsetAsync(grad_chunks, 0, (size_t)P * 50 * E * sizeof(float), stream) != hipSuccess)
      return set_error(MM_ELAUNCH, "tkl_bwd: memset failed");
  }
  TklBwdArgs a{};
  a.q_ctx = (const float*)q_ctx; a.chunks = (const float*)chunks; a.chunk_mask = chunk_mask; a.slot2p = slot2p;
  a.q_mask = q_mask; a.prm = params; a.win = win_scores; a.go = grad_out; a.gq = grad_q; a.gchunks = grad_chunks;
  a.gprm = grad_params; a.C = C; a.Q = Q; a.E = E; a.W = W; a.NP = MM_TKL_NPARAMS(K, E); a.sat = saturation;
  {
    const size_t tl = tkl_bwd_tiled_lds_bytes(Wp, Q, E);
    const int QS = (Q + 3) & ~3;
    if (!(E & 3) && (QS >> 2) * (E >> 2) <= 2 * kTT && tl <= 150 * 1024 && !env().kp_bwd_untiled &&
        !(((uintptr_t)q_ctx | (uintptr_t)chunks | (uintptr_t)grad_q | (uintptr_t)grad_chunks | (uintptr_t)params) & 15)) {
      if (tl > 64 * 1024) (void)hipFuncSetAttribute((const void*)tkl_bwd_tiled_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl);
      hipLaunchKernelGGL(tkl_bwd_tiled_kernel, dim3((unsigned)B), dim3(kTT), tl, stream, a);
      return check_launch("tkl_bwd_tiled_kernel");
    }
  }
  const size_t lds = ((size_t)2 * Wp + 2 * kBwdQ * kBwdT + 2 * kBwdQ * kK + kBwdQ * 40 + 7 * kBwdQ + 4 * (kBwdT + 2) + 16 + kBwdT + 2) * 4;
  if (lds > 160 * 1024) return set_error(MM_EUNSUPPORTED, "tkl_bwd: %d windows per document exceed the LDS", W);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)tkl_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(tkl_bwd_kernel, dim3((unsigned)B), dim3(kBwdThreads), lds, stream, a);
  return check_launch("tkl_bwd_kernel");
}
