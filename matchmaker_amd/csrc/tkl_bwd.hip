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
// chunk-row gradients are accumulated by the one workgroup that owns the document — window after window (untiled kernel) or
// region after region (tiled kernel: plain stores for a region that touches no earlier region's rows, else read-add-write) —
// without atomics: deterministic.
// tkl_bwd_tiled_kernel (below) is the product path; tkl_bwd_kernel (first) the general fallback for shapes its tiles do not fit.
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
  // small batches (tkl_bwd_tiled_kernel): nsplit = 3 workgroups per document, one per arg-max region; each leaves its share of
  // grad_q in part_gq [B, 3, Q, E] and of the parameter row in part_gp [B, 3, NP]; tkl_bwd_combine_kernel adds them in region order
  int nsplit;
  float* part_gq;
  float* part_gp;
  int32_t* flags;           // [B, 3] zero before the launch: region r of document b has written its chunk rows
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
// per FMA: 4.6 ms per DOCUMENT (bench.py extra.train_step: 22 ms for 2,048 documents, 24 x the forward).  Here, per
// document (one 512-thread workgroup):
//   * the normalised query tile and the 30 rows of the current window live in LDS; the three small products of a window
//     (cosines, chunk-row gradients, query gradient) run on the matrix pipe in exact fp32 exactly as in
//     kernel_pool_bwd_tiled_kernel (kernel_pool_bwd.hip: K split over the eight wavefronts for the cosines, 32-column tiles
//     of E for the gradients); the query gradient accumulates in registers over the 15 windows;
//   * the 15 windows, their row tables (document position -> packed chunk row, two dependent global reads each) and masks are
//     resolved ONCE, in parallel, before the window loop; the rows of window n + 1 are fetched into registers while window n
//     is computed (one workgroup per CU: nothing else hides a load issued at its point of use);
//   * the scalar chain between the products (pooled kernels, lengths, saturation forward + backward, parameter gradients)
//     reads its constants from LDS; the window lengths come out of the pooling threads' own activations (a 30-bit set per
//     query token) instead of a second 30 x 11 exponential loop on one wavefront.
// (The first tiled version — VALU FMA products, loads at the point of use, global parameter reads inside the loops — took
// 4.9 ms for 2,048 documents.)  Q <= 32, E <= 512 (16-byte rows); other shapes take the kernel above.
// ---------------------------------------------------------------------------------------------
constexpr int kKS = 12;        // row stride of the [token][kernel] tables (16-byte rows, the twelfth entry is zero)

__host__ __device__ inline int tkl_bwd_row_stride(int E) { return ((E >> 2) & 1) ? E + 8 : E + 4; }   // odd number of 16-byte units

__host__ __device__ inline size_t tkl_bwd_tiled_lds_bytes(int Wp, int Q, int E, int nthr = 512) {
  const int ES = tkl_bwd_row_stride(E), QS = (Q + 3) & ~3;
  return ((size_t)Q * ES + 32 * (size_t)ES + 2 * 32 * (size_t)QS + 2 * kBwdQ * kKS + kBwdQ * 40 + 8 * kBwdQ + 4 * 32 + 16 + 4 * kKS +
          2 * kKS + 16 + (nthr / 128) * 1024 + 2 * 15 * 32 + 3 * 64 + 44 + 2 * 64 * (size_t)QS + 2 * (size_t)Wp) * 4 + 64;
}

__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
__device__ __forceinline__ constexpr int mrow(int i) { return (i & 3) + 8 * (i >> 2); }   // C/D layout of the 32x32 MFMA: acc[i] of lane l = row mrow(i) + 4 (l >> 5), column l & 31

// Phase clocks (tools/build_variant.sh phases tkl_bwd -DMM_TKL_BWD_PHASE_TIMES=1): thread 0 of document 0 sums s_memtime
// deltas per phase and overwrites the first entries of its parameter-gradient row with them.
#ifndef MM_TKL_BWD_PHASE_TIMES
#define MM_TKL_BWD_PHASE_TIMES 0
#endif
#if MM_TKL_BWD_PHASE_TIMES
#define TKL_PH(k) do { const long long t_ = clock64(); ph[k] += (float)(t_ - t_last); t_last = t_; } while (0)
#else
#define TKL_PH(k) do { } while (0)
#endif
#define TKL_KEEP8(v) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]))
#define TKL_KEEP16(v)                                                                                                         \
  asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), \
               "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]))

template <int kLB, int NTHR>   // kLB: 16-byte chunks of a window's rows per thread = ceil(32 (E / 4) / NTHR)
__global__ void __launch_bounds__(NTHR) tkl_bwd_tiled_kernel(const TklBwdArgs a) {
  constexpr int NW = NTHR / 64;          // wavefronts: 16 (four per SIMD, <= 128 registers) or 8
  constexpr int NS = NW / 2;             // slots of partial cosine tiles (wavefronts w and w + NS share one)
  constexpr int TPW = 16 / NW;           // 32-column tiles of E per wavefront (E <= 512)
  constexpr int PP = NTHR / 256;         // threads per (token, kernel) in the pooling phase: 2 x 15 positions, or 4 x 8
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // Small batches (the reference trains 32 x 2 = 64 documents, defaults.yaml:114: 64 workgroups on 256 CUs, each walking its
  // document's three regions one after the other — 91 % of a document's 515 k cycles are per-region work): three workgroups per
  // document, workgroup r0 takes arg-max region r0.  Everything a region adds to — grad_q, the parameter row — is linear in the
  // regions' sums, so each workgroup writes its share and tkl_bwd_combine_kernel adds the three in region order.  The chunk rows
  // of regions may overlap by a few positions (peaks 15 .. 18 windows apart, or a document of < 40 windows: a few per cent of the
  // documents).  The region order of the one-workgroup launch is kept for those rows: a workgroup whose region touches an earlier
  // region's rows waits for the earlier workgroups' flags (raised after their last row store, behind an agent-scope release)
  // before its read-add-write — same sums in the same order, bit for bit, and no atomics.  Workgroups are dispatched in index
  // order and a region-0 workgroup never waits, so the wait cannot deadlock.  (Tried first: workgroup 0 walking all three regions
  // of such a document — one such document in the batch and the launch is as long as before; float atomics onto the zero-filled
  // rows — a document of < 40 windows has three addends per row and the bits changed from run to run.)
  const int S = a.nsplit;
  const int b = S > 1 ? (int)(blockIdx.x / 3u) : (int)blockIdx.x;
  const int r0 = S > 1 ? (int)(blockIdx.x - 3u * (unsigned)b) : 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = tid & 31, lh = (tid >> 5) & 1;          // MFMA lane coordinates within the wavefront
  const int C = a.C, Q = a.Q, E = a.E, W = a.W;
  const int Wp = W < 3 ? 3 : W;
  const int ES = tkl_bwd_row_stride(E), QS = (Q + 3) & ~3, NC = E >> 2;
  const float* prm = a.prm;

  float* QH = (float*)smem;                    // [Q][ES]  q_i / (|q_i| + tiny)
  float* DB = QH + Q * ES;                     // [32][ES] the window's rows (raw; rows 30, 31 and absent rows are zeros)
  float* CT = DB + 32 * ES;                    // [32][QS] cosines, [position][token]
  float* GI = CT + 32 * QS;                    // [QS][32] d loss / d c of the current block of a region, [token][position]
  float* pk = GI + QS * 32;                    // [kBwdQ][kKS]
  float* dpk = pk + kBwdQ * kKS;               // [kBwdQ][kKS]
  float* red = dpk + kBwdQ * kKS;              // [kBwdQ][40]
  float* rq = red + kBwdQ * 40;                // [kBwdQ] each:
  float* nq = rq + kBwdQ;
  float* embv = nq + kBwdQ;
  float* qmk = embv + kBwdQ;                   // query mask
  int* lmask = (int*)(qmk + kBwdQ);            // positions of the window with a non-zero activation, one bit each
  float* dev = (float*)(lmask + kBwdQ);
  float* sqs = dev + kBwdQ;                    // sum_t G c, summed over the windows
  float* vals = sqs + kBwdQ;                   // the window's value per query token
  float* rd = vals + kBwdQ;                   // [32] each:
  float* nd = rd + 32;
  float* td = nd + 32;
  float* wcs = td + 32;                        // [15] chunk_scoring of the window list (+ padding)
  float* csg = wcs + 32;                       // [16]
  float* kc = csg + 16;                        // [kKS][4]: mu, -log2(e) / (2 sigma^2), 1 / sigma^2, -
  float* dkm = kc + 4 * kKS;                   // [2][kKS]: dense, kernel_mult
  float* spl = dkm + 2 * kKS;                  // [16] the saturation block's 13 parameters
  float* PS = spl + 16;                        // [NS][32][32] partial cosine tiles (wavefronts w and w + NS share a slot)
  int* prowA = (int*)(PS + NS * 1024);          // [15][32] flat chunk row (p * 50 + row) of every position of every window, or -1
  float* mtA = (float*)(prowA + 15 * 32);      // [15][32] mask x presence
  int* prowR = (int*)(mtA + 15 * 32);          // [3][64] flat chunk row of every position of every region, -1 = none / a padding token
  int* wlist = prowR + 3 * 64;                 // [15] the windows that carry gradient, in region order; [15] = their number
  int* winf = wlist + 16;                      // [15] window index, -1 = an empty window (the constant 0, :282)
  int* rinf = winf + 16;                       // [3] first position of region r, [4 + r] its number of positions, [8 + r] it overlaps an earlier region
  float* Gsum = (float*)(rinf + 12);           // [64][QS] d loss / d c summed over the windows of the current region, by position
  float* CTR = Gsum + 64 * QS;                 // [64][QS] cosines of the region's positions
  float* orig = CTR + 64 * QS;                 // [Wp]
  float* work = orig + Wp;                     // [Wp]
  __shared__ float rv[NW];
  __shared__ int ri[NW];
  __shared__ int top_s[3];

  const float* qb = a.q_ctx + (int64_t)b * Q * E;
  float* gq = S > 1 ? a.part_gq + ((int64_t)b * 3 + r0) * Q * E : a.gq + (int64_t)b * Q * E;
  const float g = a.go[b];
#if MM_TKL_BWD_PHASE_TIMES
  float ph[16] = {0};
  long long t_last = clock64();
#endif

  // ---- query tile: raw rows -> norms, emb . q_i -> normalised in place; constants -----------------------------------
  for (int idx = tid; idx < Q * NC; idx += NTHR) {
    const int i = idx / NC, c = idx - i * NC;
    *(f32x4*)(QH + i * ES + 4 * c) = *(const f32x4*)(qb + (int64_t)i * E + 4 * c);
  }
  for (int idx = tid; idx < kBwdQ * 40; idx += NTHR) red[idx] = 0.0f;
  for (int idx = tid; idx < 2 * kBwdQ * kKS; idx += NTHR) pk[idx] = 0.0f;      // pk and dpk (their padding stays zero)
  if (tid < 16) {
    csg[tid] = 0.0f;
    spl[tid] = tid < 13 ? prm[TklParams::sat() + tid] : 0.0f;
  }
  if (tid < kBwdQ) {
    dev[tid] = 0.0f;
    sqs[tid] = 0.0f;
    qmk[tid] = tid < Q ? a.q_mask[(int64_t)b * Q + tid] : 0.0f;
  }
  if (tid >= 64 && tid < 64 + kKS) {
    const int k = tid - 64;
    f32x4 kp = {0.0f, 0.0f, 0.0f, 0.0f};
    float dn = 0.0f, km = 0.0f;
    if (k < kK) {
      const float sg = prm[TklParams::sigma() + k];
      kp = f32x4{prm[TklParams::mu() + k], -1.4426950408889634f / (2.0f * sg * sg), 1.0f / (sg * sg), 0.0f};
      dn = prm[TklParams::dense() + k];
      km = prm[TklParams::kmult() + k];
    }
    *(f32x4*)(kc + 4 * k) = kp;
    dkm[k] = dn;
    dkm[kKS + k] = km;
  }
  for (int w = tid; w < Wp; w += NTHR) {
    float s = w < W ? a.win[(int64_t)b * W + w] : 0.0f;
    if (s == 0.0f) s = -9900.0f;
    orig[w] = s;
    work[w] = s;
  }
  __syncthreads();
  if (tid < 512) {                                      // (whole wavefronts)
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
    for (int w = tid; w < Wp; w += NTHR) {
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
    for (int k = 0; k < NW; ++k) {
      const float ov = rv[k];
      const int oi = ri[k];
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (tid == 0) top_s[c] = bi;
    __syncthreads();
    for (int w = tid; w < Wp; w += NTHR) {
      const int dlt = w > bi ? w - bi : bi - w;
      if (dlt < 15) work[w] = -10001.0f - (float)c;
    }
    __syncthreads();
  }
  // ---- the window list (:276 order: peaks, -1, +1, -2, +2), every window's row table and every region's, resolved once --
  if (wv == 0) {
    const int j = lane;
    bool valid = false;
    int idx = 0;
    if (j < 15) {
      const int off = j < 3 ? 0 : (j < 6 ? -1 : (j < 9 ? 1 : (j < 12 ? -2 : 2)));
      idx = top_s[j % 3] + off;
      idx = idx < 0 ? 0 : (idx >= Wp ? Wp - 1 : idx);            // :277-278
      valid = orig[idx] > -9900.0f;                              // :282 an empty window is the constant 0
      wcs[j] = prm[TklParams::chunk_scoring() + j];
      winf[j] = valid ? idx : -1;
    }
    // the list in REGION order (region r = the windows j = r, r + 3, ..., r + 12 around peak r): the five windows of a region
    // overlap in 28 of their 30 positions, and d loss / d c is summed over them per POSITION before the two gradient
    // products run once per region (below)
    const int jr = j < 15 ? (j % 3) * 5 + j / 3 : 63;            // rank of window j in region order
    const unsigned long long m = __ballot(valid);
    if (valid) {
      int before = 0;
      for (int o = 0; o < 15; ++o) {
        const int rank_o = (o % 3) * 5 + o / 3;
        before += (((m >> o) & 1ull) && rank_o < jr) ? 1 : 0;
      }
      wlist[before] = j;
    }
    if (lane == 0) wlist[15] = __popcll(m);
    if (lane < 3) {                                              // region r: first position, number of positions (<= 38)
      int lo = 0x7fffffff, hi = -1;
      for (int o = 0; o < 5; ++o) {
        const int jj = lane + 3 * o;
        int ix = top_s[lane] + (o == 0 ? 0 : (o == 1 ? -1 : (o == 2 ? 1 : (o == 3 ? -2 : 2))));
        ix = ix < 0 ? 0 : (ix >= Wp ? Wp - 1 : ix);
        if (orig[ix] > -9900.0f) { lo = ix < lo ? ix : lo; hi = ix > hi ? ix : hi; }
        (void)jj;
      }
      rinf[lane] = hi >= 0 ? 2 * lo : 0;
      rinf[4 + lane] = hi >= 0 ? 2 * (hi - lo) + kBwdT : 0;
      // does region r touch positions of an earlier region?  (peaks are >= 15 windows = 30 positions apart and a region
      // spans <= 38: rarely) — only then must its chunk rows be read before they are added to
      const int p0 = hi >= 0 ? 2 * lo : 0, p1 = hi >= 0 ? 2 * hi + kBwdT : 0;
      int ov = 0, ova = 0;
      for (int o = 0; o < 3; ++o) {
        const int q0 = __shfl(p0, o, 64), q1 = __shfl(p1, o, 64);
        const bool hit = o != lane && q1 > q0 && p1 > p0 && q0 < p1 && p0 < q1;
        if (hit && o < lane) ov = 1;
        if (hit) ova = 1;
      }
      rinf[8 + lane] = ov;
      // ... or of ANY other region (bit r of rinf[3]): with one workgroup per region (a.nsplit = 3) such a workgroup publishes its
      // rows with a release before it raises its flag (below)
      const unsigned long long am = __ballot(ova != 0);
      if (lane == 0) rinf[3] = (int)(am & 7ull);
    }
  }
  __syncthreads();
  auto resolve = [&](int pos, bool in, int& flat, float& m) {      // document position -> packed chunk row and its mask
    flat = -1;
    m = 0.0f;
    if (in && pos < C * 40) {
      const int c = pos / 40;
      const int info = a.slot2p[(int64_t)b * C + c];
      if (info >= 0) {
        flat = (info >> 2) * 50 + 5 + (pos - 40 * c);
        m = a.chunk_mask[flat] != 0.0f ? 1.0f : 0.0f;
      }
    }
  };
  for (int e2 = tid; e2 < 15 * 32 + 3 * 64; e2 += NTHR) {
    int flat;
    float m;
    if (e2 < 15 * 32) {
      const int j = e2 >> 5, t = e2 & 31;
      const int ix = winf[j];
      resolve(2 * ix + t, ix >= 0 && t < kBwdT, flat, m);
      prowA[e2] = flat;
      mtA[e2] = m;
    } else {
      const int r = (e2 - 15 * 32) >> 6, p2 = (e2 - 15 * 32) & 63;
      resolve(rinf[r] + p2, p2 < rinf[4 + r], flat, m);
      prowR[r * 64 + p2] = m != 0.0f ? flat : -1;
    }
  }
  __syncthreads();
  const int nv = wlist[15];

  // A block of rows travels global -> registers -> LDS in two steps (see kernel_pool_bwd_tiled_kernel): fetch() is called
  // one window ahead inside a region, commit() once the current window's arithmetic has read DB.  A thread's chunks are the
  // same (row, column) in every block.
  int frow[kLB], fcol[kLB];
#pragma unroll
  for (int u = 0; u < kLB; ++u) {
    const int idx = tid + u * NTHR;
    frow[u] = 32;
    fcol[u] = 0;
    if (idx < 32 * NC) {
      frow[u] = idx / NC;
      fcol[u] = 4 * (idx - frow[u] * NC);
    }
  }
  f32x4 nxt[kLB];
  auto fetch = [&](const int* table) {
#pragma unroll
    for (int u = 0; u < kLB; ++u) {
      nxt[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      const int flat = frow[u] < 32 ? table[frow[u]] : -1;
      if (flat >= 0) nxt[u] = *(const f32x4*)(a.chunks + (int64_t)flat * E + fcol[u]);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < kLB; ++u)
      if (frow[u] < 32) *(f32x4*)(DB + frow[u] * ES + fcol[u]) = nxt[u];
  };
  auto row_norms = [&]() {                              // sixteen threads per row of DB (the first eight wavefronts)
    if (tid >= 512) return;
    const int row = tid >> 4, sub = tid & 15;
    float ss = 0.0f;
    for (int c = sub; c < NC; c += 16) {
      const f32x4 v = *(const f32x4*)(DB + row * ES + 4 * c);
      ss += dot4(v, v);
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) ss += __shfl_xor(ss, o, 64);
    if (sub == 0) {
      const float nn = sqrtf(ss);
      nd[row] = nn;
      rd[row] = 1.0f / (nn + 1e-13f);
    }
  };

  const int NT = (E + 31) >> 5;                         // 32-column tiles of E; wavefront w owns tiles w and w + 8
  const int KQ = (Q + 1) >> 1;                          // MFMA steps over the query tokens
  f32x16 accq[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) accq[t] = f32x16{0};
  TKL_PH(0);
  for (int n = 0; n < nv; ++n) {
    const int j = wlist[n];
    const int r = j % 3;
    if (S > 1 && r != r0) continue;                                 // (wave-uniform) one workgroup per region
    const bool first = n == 0 || wlist[n - 1] % 3 != r;            // (wave-uniform: LDS values)
    const bool last = n + 1 == nv || wlist[n + 1] % 3 != r;
    const int* prow = prowA + j * 32;
    const float* mt = mtA + j * 32;
    const float cs = wcs[j];
    const int poff = 2 * winf[j] - rinf[r];                        // first position of the window inside its region
    if (first) {
      fetch(prow);                                                 // (not prefetched: the registers carried the previous region's rows)
      for (int idx = tid; idx < 64 * QS; idx += NTHR) Gsum[idx] = 0.0f;
    }
    commit();
    __syncthreads();
    if (!last) fetch(prowA + wlist[n + 1] * 32);
    TKL_PH(1);
    row_norms();
    if (tid < kBwdQ) lmask[tid] = 0;
    int lw = ln, hw = lh;                                // (opaque copies: the addresses below are recomputed per window, not kept)
    asm volatile("" : "+v"(lw), "+v"(hw));
    {  // cosine tile: this wavefront's K slice of the 32 x 32 tile; the eight partial tiles meet in LDS in a fixed order
      f32x16 acc = {0};
      const float* arow = DB + lw * ES;
      const float* brow = QH + (lw < Q ? lw : Q - 1) * ES;
      for (int p = wv; 2 * p < NC; p += NW) {
        const int cc = 2 * p + hw;
        const int cl = cc < NC ? cc : NC - 1;            // (odd NC: the last pair's upper half multiplies zeros)
        f32x4 av = *(const f32x4*)(arow + 4 * cl);
        const f32x4 bv = *(const f32x4*)(brow + 4 * cl);
        if (cc >= NC) av = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q4], bv[q4], acc, 0, 0, 0);
      }
      float* ps = PS + (wv % NS) * 1024 + 4 * hw * 32 + lw;
      if (wv >= NS) {
#pragma unroll
        for (int i = 0; i < 16; ++i) ps[mrow(i) * 32] = acc[i];
      }
      __syncthreads();
      if (wv < NS) {
#pragma unroll
        for (int i = 0; i < 16; ++i) ps[mrow(i) * 32] += acc[i];
      }
    }
    __syncthreads();
    TKL_PH(2);
    for (int idx = tid; idx < 1024; idx += NTHR) {
      const int row = idx >> 5, i = idx & 31;
      float v = PS[idx];
#pragma unroll
      for (int w2 = 1; w2 < NS; ++w2) v += PS[w2 * 1024 + idx];
      v *= rd[row];
      if (i < Q) {
        CT[row * QS + i] = v;
        if (row < kBwdT) CTR[(poff + row) * QS + i] = v;           // (the same bits from every window that holds the position)
      }
    }
    __syncthreads();
    TKL_PH(3);
    for (int idx = tid / PP; idx < Q * kK; idx += NTHR / PP) {   // pooled kernels of the window: (i, k) on PP threads, 32 / PP positions each
      const int i = idx / kK, k = idx - i * kK;
      const float mu = kc[4 * k], c2 = kc[4 * k + 1];
      const int t0 = (32 / PP) * (tid % PP);                     // (positions 30, 31: zero mask)
      float s = 0.0f;
      int bits = 0;
#pragma unroll
      for (int tt = 0; tt < 32 / PP; ++tt) {
        const float d = CT[(t0 + tt) * QS + i] - mu;
        const float e = mt[t0 + tt] * __builtin_amdgcn_exp2f(d * d * c2);
        s += e;
        bits |= (e != 0.0f ? 1 : 0) << (t0 + tt);
      }
#pragma unroll
      for (int m = 1; m < PP; m <<= 1) {
        s += __shfl_xor(s, m, 64);
        bits |= __shfl_xor(bits, m, 64);
      }
      if (!(tid % PP)) {
        pk[i * kKS + k] = s;
        atomicOr(&lmask[i], bits);                                // window length (:210) = positions with any non-zero activation
      }
    }
    __syncthreads();
    TKL_PH(4);
    // ---- saturation forward + backward (as above): thread = (query token, kernel), sixteen lanes per token --------------------
    // (On the first wavefront alone — one lane per token looping over the kernels — this chain of logs, exponentials and
    // LDS read-modify-writes was 20 % of the kernel, the other seven wavefronts waiting at the barrier.)
    const float gw = g * cs;                                     // d loss / d w_j
    if (tid < 512) {                                      // (whole wavefronts)
      const int i = tid >> 4, k = tid & 15;
      const bool tin = i < Q, kin = k < kK;
      const int ic = tin ? i : 0;
      const float len = (float)__popc(lmask[ic]);
      const float f = tin ? qmk[ic] * (len > 0.0f ? 1.0f : 0.0f) : 0.0f;      // :248
      float* rr = red + ic * 40;       // [0..10] dense, [11..21] kernel_mult, [22..34] saturation block (13)
      const float p = pk[ic * kKS + (kin ? k : kKS - 1)];
      const float dk = dkm[kin ? k : kKS - 1];                   // (the twelfth entries are zero)
      const float dsat = gw * f * dk;
      float val = 0.0f;
      if (a.sat == MM_TKL_SAT_EMBEDDING) {
        const float x0 = embv[ic], x1 = len;
        const float mean = (x0 + x1) * 0.5f;
        const float d0 = x0 - mean, d1 = x1 - mean;
        const float rstd = 1.0f / sqrtf((d0 * d0 + d1 * d1) * 0.5f + 1e-5f);
        const float xh0 = d0 * rstd, xh1 = d1 * rstd;
        const float n0 = xh0 * spl[9] + spl[11], n1 = xh1 * spl[10] + spl[12];
        const float s1 = n0 * spl[0] + n1 * spl[1] + spl[2];
        const float u = n0 * spl[3] + n1 * spl[4] + spl[5];
        const float s2 = 1.0f / u;
        const float s3 = n0 * spl[6] + n1 * spl[7] + spl[8];
        const float x = fmaxf(p, 1e-10f);
        const float lx = __logf(x);
        const float xp = __expf(s2 * lx);
        const float sat = s1 * xp - s3;
        const bool on = tin && kin;                               // (idle lanes must contribute exact zeros, not 0 x inf)
        val = on ? dk * (sat * f) : 0.0f;
        float ds1 = on ? dsat * xp : 0.0f, ds3 = on ? -dsat : 0.0f, ds2 = on ? dsat * s1 * xp * lx : 0.0f;
        if (on) {
          rr[k] += gw * f * sat;                                  // d dense_k
          dpk[i * kKS + k] = p >= 1e-10f ? dsat * s1 * s2 * xp / x : 0.0f;
        }
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {                        // over the token's kernels (lanes 11..15 hold zeros)
          ds1 += __shfl_xor(ds1, m, 64);
          ds2 += __shfl_xor(ds2, m, 64);
          ds3 += __shfl_xor(ds3, m, 64);
          val += __shfl_xor(val, m, 64);
        }
        if (tin && k == 0) {
          const float du = -ds2 * s2 * s2;
          const float dn0 = ds1 * spl[0] + du * spl[3] + ds3 * spl[6];
          const float dn1 = ds1 * spl[1] + du * spl[4] + ds3 * spl[7];
          rr[22 + 0] += ds1 * n0; rr[22 + 1] += ds1 * n1; rr[22 + 2] += ds1;
          rr[22 + 3] += du * n0;  rr[22 + 4] += du * n1;  rr[22 + 5] += du;
          rr[22 + 6] += ds3 * n0; rr[22 + 7] += ds3 * n1; rr[22 + 8] += ds3;
          rr[22 + 9] += dn0 * xh0; rr[22 + 10] += dn1 * xh1;       // LayerNorm weight
          rr[22 + 11] += dn0;      rr[22 + 12] += dn1;             // LayerNorm bias
          const float dx0h = dn0 * spl[9], dx1h = dn1 * spl[10];
          const float m1 = (dx0h + dx1h) * 0.5f, m2 = (dx0h * xh0 + dx1h * xh1) * 0.5f;
          dev[i] += rstd * (dx0h - m1 - xh0 * m2);                 // d loss / d (emb . q_i); the length carries no gradient
        }
      } else {
        const float km = dkm[kKS + (kin ? k : kKS - 1)];
        const bool live = p * km >= 1e-10f;
        const float sat = __logf(fmaxf(p * km, 1e-10f));
        val = tin && kin ? dk * (sat * f) : 0.0f;
        if (tin && kin) {
          rr[k] += gw * f * sat;
          rr[11 + k] += live ? dsat / km : 0.0f;
          dpk[i * kKS + k] = live ? dsat / p : 0.0f;
        }
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) val += __shfl_xor(val, m, 64);
      }
      if (k == 0) vals[i & 31] = tin ? val : 0.0f;
    }
    __syncthreads();
    TKL_PH(5);
    if (wv == 0) {                                              // d chunk_scoring_j = g * w_j (:249 sum in index order)
      const float val = vals[lane & 31];
      float wj = 0.0f;
      for (int i = 0; i < Q; ++i) wj += __shfl(val, i, 64);
      if (lane == 0) csg[j] += g * wj;
    }
    for (int idx = tid; idx < 1024; idx += NTHR) {               // G = d loss / d c, added to the region's sum at the window's positions
      const int t = idx >> 5, i = idx & 31;
      const int ic = i < Q ? i : Q - 1;
      const float c = CT[t * QS + ic];
      const bool live = i < Q && mt[t] != 0.0f;                  // (mt[30], mt[31] are zero)
      float s = 0.0f;
#pragma unroll
      for (int k4 = 0; k4 < kKS; k4 += 4) {
        const f32x4 a4 = *(const f32x4*)(dpk + ic * kKS + k4);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const f32x4 kp = *(const f32x4*)(kc + 4 * (k4 + kk));      // the twelfth row is zeros, and so is dpk's twelfth entry
          const float d = c - kp[0];
          s -= a4[kk] * __builtin_amdgcn_exp2f(d * d * kp[1]) * d * kp[2];
        }
      }
      if (live) Gsum[(poff + t) * QS + i] += s;
    }
    __syncthreads();
    TKL_PH(6);
    if (!last) continue;                                          // (wave-uniform)

    // ---- the region's gradient products: its <= 38 positions as two blocks of 32 rows -------------------------------------
    if (S > 1 && rinf[8 + r] != 0) {                               // (wave-uniform) the earlier regions' rows first: see the top
      if (tid == 0) {
        for (int e2 = 0; e2 < r0; ++e2)
          while (__hip_atomic_load(a.flags + (int64_t)b * 3 + e2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(8);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
    }
    for (int hb = 0; hb < 2; ++hb) {
      if (32 * hb >= rinf[4 + r]) break;
      const int* ptab = prowR + r * 64 + 32 * hb;
      const float* gs = Gsum + 32 * hb * QS;                     // [32][QS] d loss / d c of the block's positions
      const float* ctr = CTR + 32 * hb * QS;
      int fmin = 0x7fffffff;                                     // lowest chunk row of the block (wave-uniform after the reduction)
      {
        const int v = ptab[ln];
        fmin = v >= 0 ? v : fmin;
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) {
          const int w2 = __shfl_xor(fmin, o, 64);
          fmin = w2 < fmin ? w2 : fmin;
        }
        fmin = __builtin_amdgcn_readfirstlane(fmin == 0x7fffffff ? 0 : fmin);
      }
      float* gblk = a.gchunks + (int64_t)fmin * E;
      fetch(ptab);
      for (int idx = tid; idx < 1024; idx += NTHR) {              // the token-major copy the query-gradient product reads
        const int t = idx >> 5, i = idx & 31;
        if (i < QS) GI[i * 32 + t] = gs[t * QS + i];
      }
      commit();
      __syncthreads();
      TKL_PH(7);
      row_norms();
      __syncthreads();
      if (tid < 512) {  // sixteen threads per position / per query token, two terms each (whole wavefronts)
        const int r16 = tid >> 4, sub = tid & 15;
        const bool rin = 32 * hb + r16 < rinf[4 + r];            // (positions past the region hold stale cosines)
        float s = 0.0f, u = 0.0f;
        if (rin)
          for (int i = sub; i < Q; i += 16) s += gs[r16 * QS + i] * ctr[r16 * QS + i];
        if (r16 < Q)
          for (int t = sub; t < 32; t += 16)
            if (32 * hb + t < rinf[4 + r]) u += gs[t * QS + r16] * ctr[t * QS + r16];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
          s += __shfl_xor(s, m, 64);
          u += __shfl_xor(u, m, 64);
        }
        if (sub == 0) {
          const float nn = nd[r16];
          td[r16] = nn > 0.0f ? s / nn : 0.0f;           // the row's own-direction term of the norm's gradient
          if (r16 < Q) sqs[r16] += u;
        }
      }
      __syncthreads();
      TKL_PH(8);
      // (lane coordinates behind an opaque copy: everything derived from them below — some 80 LDS addresses — is then computed
      // here, per block, instead of once before the window loop and kept in spilled registers)
      int lq = ln, hq = lh;
      asm volatile("" : "+v"(lq), "+v"(hq));
      {  // chunk-row gradients: A[row][token] = G (K = tokens, two per step; zero past Q), shared by this wavefront's tiles.
         // Regions may overlap each other by a few positions: rows are read, added to and written back; the regions follow
         // each other inside this one workgroup.
        float ga[16];
#pragma unroll
        for (int st = 0; st < 16; ++st) {
          const int i = 2 * st + hq;
          ga[st] = gs[lq * QS + (i < QS ? i : QS - 1)];
          if (i >= Q) ga[st] = 0.0f;
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          const int nt = wv + NW * t;
          if (nt >= NT) break;                             // wave-uniform
          const int col = 32 * nt + lq;
          const bool cin = col < E;
          const float* qcol = QH + (cin ? col : 0);
          const float* dcol = DB + (cin ? col : 0);
          float bq[16];
#pragma unroll
          for (int st = 0; st < 16; ++st) {
            const int i = 2 * st + hq;
            bq[st] = qcol[(i < Q ? i : Q - 1) * ES];       // (multiplied by ga = 0 past Q)
          }
          f32x16 acc = {0};
#pragma unroll
          for (int g2 = 0; g2 < 8; ++g2) {
            if (2 * g2 < KQ) {                             // wave-uniform
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[2 * g2], bq[2 * g2], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[2 * g2 + 1], bq[2 * g2 + 1], acc, 0, 0, 0);
            }
          }
          // rows are addressed as 32-bit offsets from the block's lowest chunk row (one address register per store
          // instead of a 64-bit pair and two wide multiplies: the block ran on spilled registers before)
          float* gcol = gblk + (cin ? col : 0);
          const bool rmw = rinf[8 + r] != 0;               // (wave-uniform) an earlier region wrote some of these rows
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {                 // eight rows at a time
            float ov[8];
            uint32_t of[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int row = mrow(8 * hf + i) + 4 * hq;
              const int fl = ptab[row];                    // (-1: no such position, or a padding token)
              ov[i] = (acc[8 * hf + i] - dcol[row * ES] * td[row]) * rd[row];
              of[i] = (cin && fl >= 0) ? (uint32_t)(fl - fmin) * (uint32_t)E : 0xffffffffu;
            }
            if (rmw) {
#pragma unroll
              for (int i = 0; i < 8; ++i)                  // eight reads in flight (lanes without a row read the block's first row and
                ov[i] += gcol[of[i] != 0xffffffffu ? of[i] : 0u];      // drop it: no lane-conditional block around a load), then the stores
            }
            TKL_KEEP8(ov);
            TKL_KEEP8(of);
#pragma unroll
            for (int i = 0; i < 8; ++i)
              if (of[i] != 0xffffffffu) gcol[of[i]] = ov[i];
          }
        }
      }
      TKL_PH(9);
      {  // query gradient: A[token][t] = G / (|d_t| + tiny) (K = the block's positions), B = the raw rows
        float gi[16];
        const int tk = lq < Q ? lq : Q - 1;
#pragma unroll
        for (int st = 0; st < 16; ++st) {
          const int i = 2 * st + hq;
          gi[st] = GI[tk * 32 + i] * rd[i];
          if (lq >= Q) gi[st] = 0.0f;
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          const int nt = wv + NW * t;
          if (nt >= NT) break;
          const int col = 32 * nt + lq;
          const float* dcol = DB + (col < E ? col : 0);
          float bd[16];
#pragma unroll
          for (int st = 0; st < 16; ++st) bd[st] = dcol[(2 * st + hq) * ES];
#pragma unroll
          for (int st = 0; st < 16; ++st) accq[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gi[st], bd[st], accq[t], 0, 0, 0);
        }
      }
      __syncthreads();   // the next block / region reuses the LDS tiles and may touch the same chunk rows
      TKL_PH(10);
    }
  }

  if (S > 1) {                                                     // this region's chunk rows are written (the block loop ends with a barrier)
    __syncthreads();
    if (tid == 0) {
      if ((rinf[3] >> r0) & 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // another region's workgroup reads them
      __hip_atomic_store(a.flags + (int64_t)b * 3 + r0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // ---- grad_q = rq (sum_w sum_t G dh - (sum G c) q / |q|) + dev emb_w; parameter rows of this document -----------------
  const bool emb_sat = a.sat == MM_TKL_SAT_EMBEDDING;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int nt = wv + NW * t;
    const int col = 32 * nt + ln;
    if (nt < NT && col < E) {
      const float ew = emb_sat ? prm[TklParams::emb() + col] : 0.0f;
      float ov[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int tok = mrow(i) + 4 * lh, tc = tok < Q ? tok : Q - 1;
        const float self = nq[tc] > 0.0f ? sqs[tc] : 0.0f;            // QH holds q_i / |q_i| (to 1e-13)
        ov[i] = (accq[t][i] - QH[tc * ES + col] * self) * rq[tc] + ew * dev[tc];
      }
      TKL_KEEP16(ov);
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (mrow(i) + 4 * lh < Q) gq[(uint32_t)((mrow(i) + 4 * lh) * E + col)] = ov[i];
    }
  }
  float* gp = S > 1 ? a.part_gp + ((int64_t)b * 3 + r0) * a.NP : a.gprm + (int64_t)b * a.NP;
  if (emb_sat) {
    for (int e = tid; e < E; e += NTHR) {
      float s = 0.0f;
      for (int i = 0; i < Q; ++i) s += dev[i] * QH[i * ES + e] * (nq[i] + 1e-13f);     // q_i = qh_i (|q_i| + tiny)
      gp[TklParams::emb() + e] = s;
    }
  } else {
    for (int e = tid; e < E; e += NTHR) gp[TklParams::emb() + e] = 0.0f;
  }
  for (int k = tid; k < 2 * kK; k += NTHR) gp[k] = 0.0f;                           // mu, sigma are not trained
  if (tid < 35) {                                                              // sum over the query tokens in index order
    float s = 0.0f;
    for (int i = 0; i < Q; ++i) s += red[i * 40 + tid];
    const int dst = tid < 11 ? TklParams::dense() + tid : (tid < 22 ? TklParams::kmult() + (tid - 11) : TklParams::sat() + (tid - 22));
    gp[dst] = s;
  }
  if (tid < 15) gp[TklParams::chunk_scoring() + tid] = csg[tid];
#if MM_TKL_BWD_PHASE_TIMES
  TKL_PH(11);
  __syncthreads();
  if (tid == 0 && b == 0)
    for (int k = 0; k < 12; ++k) gp[k] = ph[k];
#endif
}

// slot2p for the backward (the forward's preparation kernels live in tkl.hip)
__global__ void __launch_bounds__(256) tkl_bwd_fill_kernel(int32_t* slot2p, int64_t n, int32_t* flags, int64_t nflags) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) slot2p[i] = -1;
  if (i < nflags) flags[i] = 0;                            // (region flags of the three-workgroup launch; nflags <= 3 * 85 < n's grid)
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

// grad_q and the parameter row of a document whose three regions went to three workgroups: the shares added in region order
__global__ void __launch_bounds__(256) tkl_bwd_combine_kernel(const float* __restrict__ part_gq, const float* __restrict__ part_gp,
                                                              float* __restrict__ gq, float* __restrict__ gp, int QE, int NP) {
  const int64_t b = blockIdx.x;
  const float* pq = part_gq + b * 3 * QE;
  for (int i = 4 * threadIdx.x; i < QE; i += 1024) {
    const f32x4 x0 = *(const f32x4*)(pq + i), x1 = *(const f32x4*)(pq + QE + i), x2 = *(const f32x4*)(pq + 2 * QE + i);
    *(f32x4*)(gq + b * QE + i) = (x0 + x1) + x2;
  }
  const float* pp = part_gp + b * 3 * NP;
  for (int i = threadIdx.x; i < NP; i += 256) gp[b * NP + i] = (pp[i] + pp[NP + i]) + pp[2 * NP + i];
}

extern "C" size_t mm_tkl_bwd_workspace_bytes(int64_t B, int C) { return ((size_t)B * C * 4 + 255) & ~(size_t)255; }

// ... + the per-region shares of grad_q and of the parameter rows when the batch is small enough for three workgroups per document
// (3 B <= the device's CUs).  Optional: with the smaller workspace above every document gets one workgroup.
// The region workgroups of a document wait for each other's flags, so all 3 B workgroups must be able to be resident at once (one
// per CU: ~100 KB of LDS each): the device's CU count is asked from the runtime (a partitioned MI355X shows fewer than 256), never
// assumed.
static int device_cus() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = -1;
    cus[dev] = n;
  }
  return cus[dev] > 0 ? cus[dev] : 0;
}
static bool tkl_bwd_splits(int64_t B) { return 3 * B <= device_cus() && !env().tkl_bwd_nosplit; }
extern "C" size_t mm_tkl_bwd_workspace_bytes2(int64_t B, int C, int Q, int E) {
  const size_t base = mm_tkl_bwd_workspace_bytes(B, C);
  if (B <= 0 || !tkl_bwd_splits(B) || Q <= 0 || E <= 0) return base;
  return base + (((size_t)B * 3 * ((size_t)Q * E + MM_TKL_NPARAMS(kK, E)) * 4 + 255) & ~(size_t)255) + (((size_t)B * 3 * 4 + 255) & ~(size_t)255);
}

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
  // (the flags of the three-workgroups-per-document launch sit behind the per-region shares; zeroed here whenever the workspace has them)
  const size_t ws_base = mm_tkl_bwd_workspace_bytes(B, C);
  const size_t ws_shares = (size_t)B * 3 * ((size_t)Q * E + MM_TKL_NPARAMS(K, E)) * 4, ws_flags = ((size_t)B * 3 * 4 + 255) & ~(size_t)255;
  const bool can_split = tkl_bwd_splits(B) && !(E & 3) && workspace_bytes >= ws_base + ws_shares + ws_flags;
  int32_t* flags = can_split ? (int32_t*)((char*)workspace + ws_base + ws_shares) : nullptr;
  {
    const int64_t nfill = B * (int64_t)C > 3 * B ? B * (int64_t)C : 3 * B;
    hipLaunchKernelGGL(tkl_bwd_fill_kernel, dim3((unsigned)((nfill + 255) / 256)), dim3(256), 0, stream, slot2p, B * (int64_t)C, flags,
                       flags ? 3 * B : (int64_t)0);
  }
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
    // 512 threads.  (The kernel is written for NTHR = 512 | 1024 like kernel_pool_bwd_tiled_kernel; here the sixteen-wavefront
    // form loses — 2,155 vs 2,059 us for 2,048 documents, 29 spilled registers at the 128-register budget — and is not instantiated.)
    const int nthr = 512;
    const size_t tl = tkl_bwd_tiled_lds_bytes(Wp, Q, E, nthr);
    if (!(E & 3) && E <= 512 && tl <= 150 * 1024 && !env().kp_bwd_untiled &&
        !(((uintptr_t)q_ctx | (uintptr_t)chunks | (uintptr_t)grad_q | (uintptr_t)grad_chunks | (uintptr_t)params) & 15)) {
      // three workgroups per document when the batch leaves CUs idle and the caller sized the workspace for the shares
      a.nsplit = 1;
      if (can_split) {
        a.nsplit = 3;
        a.part_gq = (float*)((char*)workspace + ws_base);
        a.part_gp = a.part_gq + (size_t)B * 3 * Q * E;
        a.flags = flags;
      }
      auto go = [&](auto kern) {
        if (tl > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl);
        hipLaunchKernelGGL(kern, dim3((unsigned)(B * a.nsplit)), dim3((unsigned)nthr), tl, stream, a);
      };
      // a thread's share of a block of rows in 16-byte chunks: five (E <= 320) or eight
      E <= 320 ? go(tkl_bwd_tiled_kernel<5, 512>) : go(tkl_bwd_tiled_kernel<8, 512>);
      if (int e = check_launch("tkl_bwd_tiled_kernel")) return e;
      if (a.nsplit > 1) {
        hipLaunchKernelGGL(tkl_bwd_combine_kernel, dim3((unsigned)B), dim3(256), 0, stream, (const float*)a.part_gq, (const float*)a.part_gp,
                           grad_q, grad_params, Q * E, a.NP);
        return check_launch("tkl_bwd_combine_kernel");
      }
      return MM_OK;
    }
  }
  const size_t lds = ((size_t)2 * Wp + 2 * kBwdQ * kBwdT + 2 * kBwdQ * kK + kBwdQ * 40 + 7 * kBwdQ + 4 * (kBwdT + 2) + 16 + kBwdT + 2) * 4;
  if (lds > 160 * 1024) return set_error(MM_EUNSUPPORTED, "tkl_bwd: %d windows per document exceed the LDS", W);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)tkl_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(tkl_bwd_kernel, dim3((unsigned)B), dim3(kBwdThreads), lds, stream, a);
  return check_launch("tkl_bwd_kernel");
}
