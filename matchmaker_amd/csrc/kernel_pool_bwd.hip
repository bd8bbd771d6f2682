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
// Two kernels.  kernel_pool_bwd_tiled_kernel (below; every shape the reference's configs train at) keeps the pair's tiles
// in LDS and runs its three products on the matrix pipe in exact fp32.  kernel_pool_bwd_kernel (first) is the general
// fallback — one 256-thread workgroup per pair, fp32 VALU, operands straight from global memory — for ANY document length
// and width (the [Q,D] tiles are swept DT positions at a time): what matters there is that loss.backward() stays on the
// device without a [B,Q,D,K] tensor.  Pair-per-row layout (the one train.py feeds: neuralIR_encoder.py:86-87).
#include "mm_internal.h"
#include "kp_bwd.h"

namespace mm {

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
// cosines C = Dh Qh^T, grad_d = G^T Qh, grad_q = G Dh — staged in LDS and run on the matrix pipe in exact float32
// (v_mfma_f32_32x32x2_f32: no operand splitting, the products and sums are fp32 like torch's).
//
// kernel_pool_bwd_kernel above takes its dot products straight from global memory, one thread per output element and
// one dependent load per FMA: 1.5-2 ms per PAIR (bench.py extra.train_step, round 4: 16.6 ms for 2,048 pairs, 2.4 x
// slower than torch's eager ops on the same GPU).  Here: the normalised query tile [Q][E] and one 32-row document block
// live in LDS, the cosines of the whole document stay in LDS between the two sweeps (pooled sums need every position
// before any gradient can be formed).  Per 32-row block the eight wavefronts of the workgroup
//   * cosines: split K = E between them (wavefront w takes the 32-byte chunk pairs w, w + 8, ...: A = document rows,
//     B = query tokens, both read as 16 bytes per lane and four MFMA steps), the eight partial 32 x 32 tiles meet in LDS;
//   * grad_d block [32 rows x E]: wavefront w owns the 32-column tiles w, w + 8 of E; K = query tokens (A = G[row][token],
//     B = Qh[token][column]: consecutive lanes read consecutive floats);
//   * grad_q [Q x E]: the same column tiles, K = the block's 32 rows (A = G[token][row] / (|d_row| + tiny), B = the raw
//     document block), accumulated in registers across the blocks.
// (The first version of this kernel ran the three products as register-blocked VALU FMAs out of the same LDS tiles:
// 1.12 ms for 2,048 pairs, 17 % VALU utilisation — every FMA needed its operands from LDS.)
// Two forms: 1,024 threads (sixteen wavefronts, <= 128 registers: one column tile per wavefront, twice the wavefronts to
// hide the phases' latencies; 0.79 ms for 2,048 pairs) when its 8 partial-tile slots fit the LDS, else 512 threads (0.85 ms).
// The document is read from HBM/L2 twice and its gradient written once.  Q <= 32, E <= 384 (16-byte rows), K <= 16, LDS
// permitting; everything else takes the kernel above.
// ---------------------------------------------------------------------------------------------
constexpr int kTK = 16;
constexpr int kTT = 512;   // threads of the tiled kernel's smaller form (eight wavefronts, two per SIMD, around ONE set of LDS tiles per CU); the default form has 1,024

// Row stride of the [row][E] tiles in floats: an ODD number of 16-byte units, so that the sixteen lanes of a ds_read_b128
// phase that read one column chunk of sixteen consecutive rows (the A / B operands of the cosine MFMAs) cover all 64 banks.
// (E + 4 alone is even for E = 300: 304 floats = 48 banks apart, rows r and r + 4 on the same banks, 8-way conflicts.)
__host__ __device__ inline int kp_bwd_row_stride(int E) { return ((E >> 2) & 1) ? E + 8 : E + 4; }

__host__ __device__ inline size_t kp_bwd_tiled_lds_bytes(int Q, int E, int Dpad, int nthr = 512) {
  const int ES = kp_bwd_row_stride(E), QS = (Q + 3) & ~3;
  return ((size_t)Q * ES + 32 * (size_t)ES + (size_t)Dpad * QS + 3 * 32 * (size_t)QS + 2 * (size_t)Q * kTK + 4 * (size_t)Dpad +
          5 * 32 + 4 * kTK + (nthr / 128) * 1024) * 4;
}


__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

// Phase clocks (tools/build_variant.sh phases kernel_pool_bwd -DMM_KP_BWD_PHASE_TIMES=1; tools/bench_kp_bwd_phases.py): thread 0
// of every pair sums s_memtime deltas per phase and pair 0 overwrites its grad_w / grad_alpha rows with them.
#ifndef MM_KP_BWD_PHASE_TIMES
#define MM_KP_BWD_PHASE_TIMES 0
#endif
// KP_KEEP16(v): the sixteen values are computed HERE.  Without it the compiler sinks each value's LDS reads and arithmetic
// into the lane-conditional block of the store that uses it: per store two dependent LDS round trips behind a branch.  (One
// statement for all sixteen: a volatile asm per value orders them and serialises the reads just the same — both measured, 35 %
// and 17 % of the kernel.)
#define KP_KEEP16(v)                                                                                                          \
  asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), \
               "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]))
#if MM_KP_BWD_PHASE_TIMES
#define KP_PH(k) do { const long long t_ = clock64(); ph[k] += (float)(t_ - t_last); t_last = t_; } while (0)
#else
#define KP_PH(k) do { } while (0)
#endif

template <bool GATE, int kLB, int NTHR>   // kLB: 16-byte chunks of a document block per thread = ceil(32 (E / 4) / NTHR)
__global__ void __launch_bounds__(NTHR) kernel_pool_bwd_tiled_kernel(const KpBwdArgs a, const int Dpad) {
  constexpr int NW = NTHR / 64;          // wavefronts: 8 (two per SIMD, 256 registers each) or 16 (four per SIMD, 128 each)
  constexpr int NS = NW / 2;             // slots of partial cosine tiles (wavefronts w and w + NS share one)
  constexpr int TPW = 16 / NW;           // 32-column tiles of E per wavefront (E <= 512)
  constexpr int PP = NTHR / 256;         // threads per (token, kernel) in the pooling phase
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int64_t pair = blockIdx.x;
  const int Q = a.Q, D = a.D, E = a.E, K = a.K;
  const int ES = kp_bwd_row_stride(E), QS = (Q + 3) & ~3, NC = E >> 2;
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
  float* kc = td + 32;               // [kTK][4]: mu, -log2(e) / (2 sigma^2), 1 / sigma^2, -
  float* PS = kc + 4 * kTK;          // [4][32][32] partial cosine tiles (wavefronts w and w + 4 share a slot), [row][token]
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = tid & 31, lh = (tid >> 5) & 1;           // MFMA lane coordinates within the wavefront
  const float* qb = a.q + pair * Q * (int64_t)E;
  const float* db = a.d + pair * D * (int64_t)E;
  float* gq = a.gq + pair * Q * (int64_t)E;
  float* gd = a.gd + pair * D * (int64_t)E;
  const float g = a.go[pair];
  const int qwords = (Q + 31) >> 5, dwords = (D + 31) >> 5;
#if MM_KP_BWD_PHASE_TIMES
  float ph[12] = {0};
  long long t_last = clock64();
#endif

  // ---- query tile, constants ---------------------------------------------------------------------------------
  for (int idx = tid; idx < Q * NC; idx += NTHR) {
    const int i = idx / NC, c = idx - i * NC;
    *(f32x4*)(QH + i * ES + 4 * c) = *(const f32x4*)(qb + (int64_t)i * E + 4 * c);
  }
  if (tid < kTK) {                                      // (entries past K are zeros: the G loop runs in fours)
    f32x4 kp = {0.0f, 0.0f, 0.0f, 0.0f};
    if (tid < K) {
      const float sg = a.sigma[tid];
      kp = f32x4{a.mu[tid], -1.4426950408889634f / (2.0f * sg * sg), 1.0f / (sg * sg), 0.0f};
    }
    *(f32x4*)(kc + 4 * tid) = kp;
  }
  for (int idx = tid; idx < Q * kTK; idx += NTHR) PK[idx] = 0.0f;
  if (tid < 32) sq[tid] = 0.0f;
  __syncthreads();
  if (tid < 512) {  // norms: sixteen threads per query token; the tile is stored normalised
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

  // A document block travels global -> registers -> LDS in two steps: fetch() issues every load of a thread's share at
  // once (kLB <= 8 chunks of 16 bytes) and is called one block AHEAD, right after the barrier that opens the
  // current block's arithmetic; commit() stores them to DB once that arithmetic has read the current block.  With one
  // workgroup per CU nothing else hides the load latency: issued at the point of use (round 4's first version) it was half
  // of the kernel's time.  A thread's chunks are the same (row, column) in every block.
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
  auto fetch = [&](int j0) {
    const int nj = D - j0 < 32 ? D - j0 : 32;
#pragma unroll
    for (int u = 0; u < kLB; ++u) {
      nxt[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (frow[u] < nj) nxt[u] = *(const f32x4*)(db + (uint32_t)((j0 + frow[u]) * E + fcol[u]));
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < kLB; ++u)
      if (frow[u] < 32) *(f32x4*)(DB + frow[u] * ES + fcol[u]) = nxt[u];
  };
  int dlen = a.dm.len ? a.dm.len[pair] : D;
  dlen = dlen < D ? dlen : D;

  // ---- sweep 1: cosines of every block, pooled kernels ---------------------------------------------------------
  fetch(0);
  KP_PH(0);
  for (int j0 = 0; j0 < D; j0 += 32) {
    const int nj = D - j0 < 32 ? D - j0 : 32;
    commit();
    __syncthreads();
    KP_PH(1);
    if (tid < 512) {  // row norms and masks: sixteen threads per row (whole wavefronts)
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
        const int j = j0 + row;
        bool real = row < nj && j < dlen;
        if (a.dm.bits) real = real && ((a.dm.bits[pair * dwords + (j >> 5)] >> (j & 31)) & 1u);
        float gate = 1.0f;
        if (GATE) gate = row < nj ? fmaxf(a.dw[pair * D + j], 0.0f) : 0.0f;
        RD[j] = row < nj ? 1.0f / (n + 1e-13f) : 0.0f;
        ND[j] = n;
        DMB[j] = real ? 1.0f : 0.0f;
        DMF[j] = real ? gate : 0.0f;
      }
    }
    __syncthreads();
    KP_PH(2);
    fetch(j0 + 32 < D ? j0 + 32 : 0);                   // the next block, or sweep 2's first
    {  // cosine tile: this wavefront's K slice of the 32 x 32 tile; the eight partial tiles meet in LDS in a fixed order
      f32x16 acc = {0};
      const float* arow = DB + ln * ES;
      const float* brow = QH + (ln < Q ? ln : Q - 1) * ES;
      for (int p = wv; 2 * p < NC; p += NW) {
        const int cc = 2 * p + lh;
        const int cl = cc < NC ? cc : NC - 1;            // (odd NC: the last pair's upper half multiplies zeros)
        f32x4 av = *(const f32x4*)(arow + 4 * cl);
        const f32x4 bv = *(const f32x4*)(brow + 4 * cl);
        if (cc >= NC) av = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
      }
      float* ps = PS + (wv % NS) * 1024 + 4 * lh * 32 + ln;
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
    KP_PH(3);
    for (int idx = tid; idx < 1024; idx += NTHR) {
      const int row = idx >> 5, i = idx & 31;
      float v = PS[idx];
#pragma unroll
      for (int w = 1; w < NS; ++w) v += PS[w * 1024 + idx];
      if (i < Q) CT[(j0 + row) * QS + i] = v * RD[j0 + row];
    }
    __syncthreads();
    KP_PH(4);
    for (int idx = tid / PP; idx < Q * K; idx += NTHR / PP) {   // (i, k): PP threads, 32 / PP positions each; owned across the blocks
      const int i = idx / K, k = idx - i * K;
      const float mu = kc[4 * k], c2 = kc[4 * k + 1];
      const int jb = j0 + (32 / PP) * (tid % PP);
      float pk = 0.0f;
#pragma unroll
      for (int jj = 0; jj < 32 / PP; ++jj) {
        const float t = CT[(jb + jj) * QS + i] - mu;
        pk += DMF[jb + jj] * __builtin_amdgcn_exp2f(t * t * c2);
      }
#pragma unroll
      for (int m = 1; m < PP; m <<= 1) pk += __shfl_xor(pk, m, 64);
      if (!(tid % PP)) PK[i * kTK + k] += pk;
    }
    __syncthreads();
    KP_PH(5);
  }

  // ---- A_ik and the parameter gradients of this pair -----------------------------------------------------------
  for (int idx = tid; idx < Q * K; idx += NTHR) {
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

  KP_PH(6);
  // ---- sweep 2: G per block, grad_d (complete per block), grad_q (register accumulators across the blocks) -------
  // Everything the matrix pipe is fed with is read unconditionally (clamped addresses, zeros selected afterwards): a
  // conditional LDS read in front of an MFMA becomes a branch, a wait and an exposed latency per step.
  const int NT = (E + 31) >> 5;                         // 32-column tiles of E; wavefront w owns tiles w and w + 8
  const int KQ = (Q + 1) >> 1;                          // MFMA steps over the query tokens
  f32x16 accq[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) accq[t] = f32x16{0};
  for (int j0 = 0; j0 < D; j0 += 32) {
    const int nj = D - j0 < 32 ? D - j0 : 32;
    commit();
    if (j0 + 32 < D) fetch(j0 + 32);
    for (int idx = tid; idx < 1024; idx += NTHR) {
      const int jj = idx >> 5, i = idx & 31;
      // every lane runs the arithmetic on a clamped token (reads ahead of the exp chain, no lane-conditional block)
      const int ic = i < Q ? i : Q - 1;
      const float c = CT[(j0 + jj) * QS + ic];
      const bool live = i < Q && DMB[j0 + jj] != 0.0f;
      float gs = 0.0f, sg = 0.0f;
      for (int k4 = 0; k4 < K; k4 += 4) {
        const f32x4 a4 = *(const f32x4*)(A + ic * kTK + k4);        // (rows of A are kTK = 16 floats; K <= kTK)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const f32x4 kp = *(const f32x4*)(kc + 4 * (k4 + kk));      // zeros past K (below): ae = 0
          const float t = c - kp[0];
          const float ae = (k4 + kk < K ? a4[kk] : 0.0f) * __builtin_amdgcn_exp2f(t * t * kp[1]);
          sg += ae;
          gs -= ae * t * kp[2];
        }
      }
      const float G = live ? gs * DMF[j0 + jj] : 0.0f;
      if (i < QS) {
        GJ[jj * QS + i] = G;
        GI[i * 32 + jj] = G;
        if (GATE) SG[jj * QS + i] = live ? sg : 0.0f;
      }
    }
    __syncthreads();
    KP_PH(7);
    if (tid < 512) {  // sixteen threads per row / per query token, two terms each (whole wavefronts)
      const int r16 = tid >> 4, sub = tid & 15;
      float s = 0.0f, u = 0.0f, sgs = 0.0f;
      for (int i = sub; i < Q; i += 16) {                // sum_i G c of every position of the block
        s += GJ[r16 * QS + i] * CT[(j0 + r16) * QS + i];
        if (GATE) sgs += SG[r16 * QS + i];
      }
      if (r16 < Q)                                       // sum_j G c of every query token, over all blocks
        for (int jj = sub; jj < 32; jj += 16) u += GI[r16 * 32 + jj] * CT[(j0 + jj) * QS + r16];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) {
        s += __shfl_xor(s, m, 64);
        u += __shfl_xor(u, m, 64);
        if (GATE) sgs += __shfl_xor(sgs, m, 64);
      }
      if (sub == 0) {
        const float n = ND[j0 + r16];
        td[r16] = n > 0.0f ? s / n : 0.0f;               // the row's own-direction term of the norm's gradient
        if (r16 < Q) sq[r16] += u;
        if (GATE && r16 < nj && a.gdw) a.gdw[pair * D + j0 + r16] = sgs;   // gate gradient: sum_ik A_ik e_ijk
      }
    }
    __syncthreads();
    KP_PH(8);
    // (lane coordinates behind an opaque copy: the ~80 per-lane LDS addresses derived from them below are then computed
    // here, per block, instead of once before the loop and kept in registers the 1,024-thread form does not have)
    int lq = ln, hq = lh;
    asm volatile("" : "+v"(lq), "+v"(hq));
    {  // grad_d of the block: A[row][token] = G (K = tokens, two per step; zero past Q), shared by this wavefront's tiles
      float ga[16];
#pragma unroll
      for (int st = 0; st < 16; ++st) {
        const int i = 2 * st + hq;
        ga[st] = GJ[lq * QS + (i < QS ? i : QS - 1)];
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
        const uint32_t o0 = (uint32_t)((j0 + 4 * hq) * E + col);
        float ov[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = mrow(i) + 4 * hq;
          ov[i] = (acc[i] - dcol[row * ES] * td[row]) * RD[j0 + row];
        }
        if (nj == 32) {                                  // wave-uniform: a full block stores under ONE lane mask (the tile's columns < E)
          if (cin) {
#pragma unroll
            for (int i = 0; i < 16; ++i) gd[o0 + (uint32_t)(mrow(i) * E)] = ov[i];   // 128-byte row segments per store
          }
        } else {
          KP_KEEP16(ov);
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (cin && mrow(i) + 4 * hq < nj) gd[o0 + (uint32_t)(mrow(i) * E)] = ov[i];
        }
      }
    }
    KP_PH(9);
    {  // grad_q: A[token][row] = G / (|d_row| + tiny) (K = the block's rows), B = the raw document block
      float gi[16];
      const int tk = lq < Q ? lq : Q - 1;
#pragma unroll
      for (int st = 0; st < 16; ++st) {
        const int i = 2 * st + hq;
        gi[st] = GI[tk * 32 + i] * RD[j0 + i];
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
    __syncthreads();
    KP_PH(10);
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int nt = wv + NW * t;
    const int col = 32 * nt + ln;
    if (nt < NT && col < E) {
      float ov[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int tok = mrow(i) + 4 * lh, tc = tok < Q ? tok : Q - 1;
        const float self = nq[tc] > 0.0f ? sq[tc] : 0.0f;            // QH holds q_i / |q_i| (to 1e-13)
        ov[i] = (accq[t][i] - QH[tc * ES + col] * self) * rq[tc];
      }
      KP_KEEP16(ov);
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (mrow(i) + 4 * lh < Q) gq[(uint32_t)((mrow(i) + 4 * lh) * E + col)] = ov[i];
    }
  }
#if MM_KP_BWD_PHASE_TIMES
  KP_PH(11);
  __syncthreads();
  if (tid == 0 && pair == 0)
    for (int k = 0; k < 12; ++k) (k < K ? a.gw[k] : a.galpha[k - K]) = ph[k];
#endif
}

}  // namespace mm

using namespace mm;

// masks as the kernels read them + the pooled kernel sums [n_pairs, Q, K <= 16] the split kernel's pooling pre-pass hands to its
// gradient pass (kernel_pool_bwd_split.hip; unused when the caller passes the forward's sums, mm_kernel_pool_ex_bwd2)
extern "C" size_t mm_kernel_pool_bwd_workspace_bytes(int64_t n_pairs, int Q, int D, int q_mask_kind, int d_mask_kind) {
  const size_t masks = packed_mask_bytes(q_mask_kind, n_pairs, Q) + packed_mask_bytes(d_mask_kind, n_pairs, D);
  return ((masks + 255) & ~(size_t)255) + kp_bwd_split_ws_bytes(n_pairs, Q, 16);
}

// ... + the partial grad_q buffers of a small batch, whose pairs are shared by several workgroups each (kp_bwd_split_launch;
// 0 bytes from 129 pairs on).  A workspace of the size above still works: one workgroup per pair then.
extern "C" size_t mm_kernel_pool_bwd_workspace_bytes2(int64_t n_pairs, int Q, int D, int E, int q_mask_kind, int d_mask_kind) {
  const size_t base = mm_kernel_pool_bwd_workspace_bytes(n_pairs, Q, D, q_mask_kind, d_mask_kind);
  const size_t part = (E > 0 && kp_bwd_split_supported(Q, E, 11)) ? kp_bwd_split_part_bytes(n_pairs, Q, D, E) : 0;
  return ((base + 255) & ~(size_t)255) + (part ? part + 256 : 0);
}

extern "C" int mm_kernel_pool_ex_bwd2(const void* q, const void* d, const void* q_mask, int q_mask_kind, const void* d_mask,
                                      int d_mask_kind, const float* d_gate, const float* mu, const float* sigma,
                                      const float* alpha, const float* w, float clamp_min, const float* pooled,
                                      const float* grad_out, float* grad_q, float* grad_d, float* grad_gate, float* grad_alpha,
                                      float* grad_w, int64_t n_pairs, int Q, int D, int E, int K, void* workspace,
                                      size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!q || !d || !mu || !sigma || !alpha || !w || !grad_out || !grad_q || !grad_d || !grad_alpha || !grad_w)
    return set_error(MM_EINVAL, "kernel_pool_bwd: null pointer");
  if (grad_gate && !d_gate) return set_error(MM_EINVAL, "kernel_pool_bwd: grad_gate without d_gate");
  if (n_pairs < 0 || Q <= 0 || D <= 0 || E <= 0 || K <= 0) return set_error(MM_EINVAL, "kernel_pool_bwd: bad shape");
  if (!(clamp_min > 0.0f)) return set_error(MM_EINVAL, "kernel_pool_bwd: clamp_min must be > 0");
  if (K > kBK) return set_error(MM_EUNSUPPORTED, "kernel_pool_bwd: K=%d kernels (max %d)", K, kBK);
  if (n_pairs == 0) return MM_OK;
  if (n_pairs > 0x7fffffffLL) return set_error(MM_EUNSUPPORTED, "kernel_pool_bwd: too many pairs for one launch");
  // The split-bf16 streaming kernel (kernel_pool_bwd_split.hip) for every shape the reference's configs train at (Q <= 32,
  // 11 kernels, E <= 384); MM_KP_BWD_F32=1 keeps the exact-f32 tiled kernel below as its parity twin (A/B runs).
  if (kp_bwd_split_supported(Q, E, K) && !env().kp_bwd_f32 && !env().kp_bwd_untiled &&
      !(((uintptr_t)q | (uintptr_t)d | (uintptr_t)grad_q | (uintptr_t)grad_d) & 15)) {
    KpBwdArgs a{};
    a.q = (const float*)q; a.d = (const float*)d; a.mu = mu; a.sigma = sigma; a.alpha = alpha; a.w = w; a.go = grad_out;
    a.dw = d_gate; a.gdw = grad_gate; a.clamp_min = clamp_min;
    a.gq = grad_q; a.gd = grad_d; a.galpha = grad_alpha; a.gw = grad_w; a.n_pairs = n_pairs; a.Q = Q; a.D = D; a.E = E; a.K = K;
    char* ws = (char*)workspace;
    size_t left = workspace ? workspace_bytes : 0;
    if (int e = resolve_mask_pair(q_mask, q_mask_kind, n_pairs, Q, &a.qm, d_mask, d_mask_kind, n_pairs, D, &a.dm, &ws, &left, stream)) return e;
    float* pkq_ws = nullptr;
    if (!pooled) {
      const size_t skip = (size_t)(-(intptr_t)ws) & 255, need = kp_bwd_split_ws_bytes(n_pairs, Q, K);
      if (left < skip + need)
        return set_error(MM_EINVAL, "kernel_pool_bwd: workspace too small for the pooled sums (%zu bytes left, %zu needed; "
                                    "mm_kernel_pool_bwd_workspace_bytes)", left, skip + need);
      pkq_ws = (float*)(ws + skip);
      ws += skip + need;
      left -= skip + need;
    }
    const size_t pskip = (size_t)(-(intptr_t)ws) & 255;
    float* part = left > pskip ? (float*)(ws + pskip) : nullptr;
    return kp_bwd_split_launch(a, pooled, pkq_ws, part, part ? left - pskip : 0, stream);
  }
  // the exact-f32 tiled kernel whenever its tiles fit; the per-element kernel otherwise
  {
    const int Dpad = (D + 31) & ~31;
    const int nthr = (env().kp_bwd_threads == 1024 && kp_bwd_tiled_lds_bytes(Q, E, Dpad, 1024) <= 150 * 1024) ? 1024 : 512;
    const size_t tl = kp_bwd_tiled_lds_bytes(Q, E, Dpad, nthr);
    if (Q <= 32 && !(E & 3) && E <= 512 && K <= kTK && tl <= 150 * 1024 &&
        !(((uintptr_t)q | (uintptr_t)d | (uintptr_t)grad_q | (uintptr_t)grad_d) & 15) && !env().kp_bwd_untiled) {
      KpBwdArgs a{};
      a.q = (const float*)q; a.d = (const float*)d; a.mu = mu; a.sigma = sigma; a.alpha = alpha; a.w = w; a.go = grad_out;
      a.dw = d_gate; a.gdw = grad_gate; a.clamp_min = clamp_min;
      a.gq = grad_q; a.gd = grad_d; a.galpha = grad_alpha; a.gw = grad_w; a.n_pairs = n_pairs; a.Q = Q; a.D = D; a.E = E; a.K = K;
      char* ws = (char*)workspace;
      size_t left = workspace ? workspace_bytes : 0;
      if (int e = resolve_mask_pair(q_mask, q_mask_kind, n_pairs, Q, &a.qm, d_mask, d_mask_kind, n_pairs, D, &a.dm, &ws, &left, stream)) return e;
      auto go = [&](auto kern) {
        if (tl > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl);
        hipLaunchKernelGGL(kern, dim3((unsigned)n_pairs), dim3((unsigned)nthr), tl, stream, a, Dpad);
      };
      // a thread's share of a 32-row block in 16-byte chunks: 512 threads: five (E <= 320) or eight; 1,024: three (E <= 384) or four
      if (nthr == 1024) {
        const bool small = E <= 384;
        if (d_gate) small ? go(kernel_pool_bwd_tiled_kernel<true, 3, 1024>) : go(kernel_pool_bwd_tiled_kernel<true, 4, 1024>);
        else small ? go(kernel_pool_bwd_tiled_kernel<false, 3, 1024>) : go(kernel_pool_bwd_tiled_kernel<false, 4, 1024>);
      } else {
        const bool small = E <= 320;
        if (d_gate) small ? go(kernel_pool_bwd_tiled_kernel<true, 5, 512>) : go(kernel_pool_bwd_tiled_kernel<true, 8, 512>);
        else small ? go(kernel_pool_bwd_tiled_kernel<false, 5, 512>) : go(kernel_pool_bwd_tiled_kernel<false, 8, 512>);
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
  if (int e = resolve_mask_pair(q_mask, q_mask_kind, n_pairs, Q, &a.qm, d_mask, d_mask_kind, n_pairs, D, &a.dm, &ws, &left, stream)) return e;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kernel_pool_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kernel_pool_bwd_kernel, dim3((unsigned)n_pairs), dim3(256), lds, stream, a, DT);
  return check_launch("kernel_pool_bwd_kernel");
}

extern "C" int mm_kernel_pool_ex_bwd(const void* q, const void* d, const void* q_mask, int q_mask_kind, const void* d_mask,
                                     int d_mask_kind, const float* d_gate, const float* mu, const float* sigma,
                                     const float* alpha, const float* w, float clamp_min, const float* grad_out,
                                     float* grad_q, float* grad_d, float* grad_gate, float* grad_alpha, float* grad_w,
                                     int64_t n_pairs, int Q, int D, int E, int K, void* workspace, size_t workspace_bytes,
                                     void* stream_) {
  return mm_kernel_pool_ex_bwd2(q, d, q_mask, q_mask_kind, d_mask, d_mask_kind, d_gate, mu, sigma, alpha, w, clamp_min, nullptr,
                                grad_out, grad_q, grad_d, grad_gate, grad_alpha, grad_w, n_pairs, Q, D, E, K, workspace,
                                workspace_bytes, stream_);
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
