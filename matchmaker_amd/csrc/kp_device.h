// Device helpers shared by the kernel-pooling translation units (kernel_pool.hip, kernel_pool128.hip).
#pragma once
#include "mm_internal.h"

namespace mm {

constexpr int kMaxK = 32;  // array bound; the streaming kernels are instantiated for the reference's K = 11

struct KpArgs {
  const float* q;
  const float* d;
  PackedMask qm, dm;
  const float* mu;
  const float* sigma;
  const float* alpha;
  const float* w;
  float* out;
  float* per_kernel;  // optional [n_pairs, K]
  // optional [n_pairs, Q, K]: the pooled kernel sums pkq[i][k] BEFORE alpha / clamp / log (ecai20_tk.py:120), one row per query
  // token — what the backward needs before it can form any gradient (mm_kernel_pool_ex_bwd2: saves its pooling pre-pass, i.e. the
  // second trip of the document through HBM).  Rows of padded / masked query tokens may stay unwritten (the backward never reads them).
  float* pooled;
  int64_t n_pairs;
  int64_t ppq;
  int Q, D, E, K;
  int64_t pairs_per_wave;
  // document addressing: document p starts at row p*d_doc_rows + d_row0 (TK: D, 0; TKL chunks: 50, 5)
  int64_t d_doc_rows;
  int d_row0;
  // TKL mode (sigir20_tkl.py:180-199): "documents" are packed chunks, the query of chunk p is
  // chunk_slot[p] / C, and instead of pooling over the document the kernel emits, per position pair
  // u (positions 2u, 2u+1 of the chunk's 40 centre tokens), the K summed activations + the number of
  // positions whose activation is non-zero: ps_out[p][20][Q][K+1].
  const int32_t* chunk_slot;
  int C;
  float* ps_out;
  // slot map of stage 2 (tkl.hip), published by stage 1 itself — one launch less than a separate scatter kernel:
  // slot2p[chunk_slot[p]] = (p << 2) | number of 32-row blocks whose pair rows this kernel writes for chunk p
  // (entries of dropped chunks were set to -1 by the preparation launch, earlier in the stream)
  int32_t* slot2p;
  int64_t n_slots;
  // TKL, cosine hand-off (tkl_stage1_run_kernel<COS>): instead of the pair sums stage 1 stores the scaled, masked
  // cosines themselves, cos_out[b * C * 40 * Q + (c * 40 + position) * ql + query token] (ql = the query's effective
  // length: only real tokens are stored; masked position: 1e5, which underflows every
  // kernel to exactly 0), and the window kernel evaluates the RBF kernels while it stages its tile — 4 bytes per
  // (position, token) across HBM instead of 24, and the transcendental work runs on sixteen thin wavefronts per CU
  // instead of one fat wavefront per SIMD
  float* cos_out;
  // variants of the pooling block (same arithmetic family):
  //   dw != nullptr: per document token gate >= 0 multiplying all its activations (TK-Sparse stop-word
  //   vector, cikm20_tk_sparse.py:133-135), [n_pairs, D] float32;
  //   clamp_min: floor inside the log (1e-10 TK :121; 1e-4 IDCM sampler, sigir21_idcm.py:185)
  const float* dw;
  float clamp_min;
  //   pair_q != nullptr: the query of pair p is row pair_q[p] of q (ragged groups: IDCM's packed passages);
  //   overrides ppq
  const int32_t* pair_q;
  // several (query tensor, document tensor) combinations in ONE launch (Conv-KNRM's n_grams^2 match matrices,
  // conv_knrm.py:130-132): blockIdx.y = i * n_md + t scores q = mq[i] against d = md[t] with the bin weights
  // w + y * K and writes its scores to out + y * n_pairs (summed in block order afterwards).  n_md = 0: one combination.
  // eval.py-sized calls (two wavefronts per pair, one pair per workgroup): the float {0, 1} masks as the caller passes them,
  // [q rows, Q] / [n_pairs, D] with Q <= 32, D <= 256 — read by the kernel itself (lane loads + ballots) instead of by a
  // mask-packing launch in front of it; qm / dm stay empty then
  const float* fqm;
  const float* fdm;
  const float* mq[4];
  const float* md[4];
  int n_md;
  int n_mblk;  // n_mq * n_md = grid.y (0 when n_md = 0)
  // Flat, XCD-grouped order of a multi launch (kernel_pool128.hip, round 5).  The 2-D grid ran combination after combination:
  // each document tensor crossed HBM once per QUERY tensor (3 x for Conv-KNRM's 3 x 3), a whole collection apart.  Flat: the
  // n_mq workgroups that score the same pair range of the same document tensor get the linear ids g, g + 8, g + 16 — same XCD
  // (workgroup b runs on XCD b % 8: observed, used for speed only), dispatched together — so the second and third reader of a
  // block find it in that XCD's L2 (a follower cannot overtake the leader without taking over its misses: the three stay
  // together).  m_ranges = pair ranges per combination (the 2-D grid's x extent); block_x = this workgroup's range.
  // m_flat = 2 (round 5, MM_KP_MULTI_WG=1, A/B only): ONE workgroup per (pair range, document tensor) whose n_mq wavefronts
  // are the query tensors — each with its own ring and query tile, no shared state — and meet at an s_barrier once per
  // 32-token block: a rate limiter that keeps the n_mq readers of a block within one block of each other (in the flat order
  // the three drift apart: FETCH_SIZE 46.9 GB per launch against the 2-D grid's 49.9, 22.9 with the barrier).  Bit-equal, and SLOWER than the flat order (9.19 vs 8.40 ms):
  // workgroups of three wavefronts fill six of a CU's eight two-per-SIMD slots, and the launch is bound by the RBF
  // evaluations, not by the bytes the barrier saves.
  int m_flat;
  int m_ranges;
  int block_x;
};

// wq: this wavefront's index in a workgroup of the m_flat = 2 form (0 otherwise)
__device__ __forceinline__ KpArgs kp_block_args(const KpArgs& a, int wq = 0) {
  KpArgs b = a;
  b.block_x = (int)blockIdx.x;
  if (a.n_md > 0) {
    int y = blockIdx.y;
    if (a.m_flat == 2) {
      const int fr = (int)blockIdx.x;                                // flat (pair range, document tensor)
      const int x = fr / a.n_md, t = fr - x * a.n_md;
      y = __builtin_amdgcn_readfirstlane(wq * a.n_md + t);
      b.block_x = __builtin_amdgcn_readfirstlane(x);
    } else if (a.m_flat) {
      const int n_mq = a.n_mblk / a.n_md, period = 8 * n_mq;
      const int g = (int)blockIdx.x, blk = g / period, r = g - blk * period;
      const int64_t fr = (int64_t)blk * 8 + (r & 7);                 // flat (pair range, document tensor)
      const int x = (int)(fr / a.n_md), t = (int)(fr - (int64_t)x * a.n_md);
      // (the divisions run on the VALU: tell the compiler the results are wave-uniform again — scalar loads hang off them)
      y = __builtin_amdgcn_readfirstlane((r >> 3) * a.n_md + t);
      b.block_x = __builtin_amdgcn_readfirstlane(x < a.m_ranges ? x : 0x3fffffff);   // padding workgroups: a range past the pairs, they leave at once
    }
    const int i = y / a.n_md, t = y - i * a.n_md;
    b.q = a.mq[i];
    b.d = a.md[t];
    b.w = a.w + y * a.K;
    b.out = a.out + (int64_t)y * a.n_pairs;
  }
  return b;
}

__device__ __forceinline__ constexpr int rowof(int i) { return (i & 3) + 8 * (i >> 2); }

__device__ __forceinline__ uint32_t sload_u32(const void* base, int64_t idx) {
  uint32_t v;
  const uint32_t* p = (const uint32_t*)base + idx;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p));
  return v;
}

// RBF constants in exp2 form: exp(-(c-mu)^2/(2 s^2)) = exp2((c-mu)^2 * c2), c2 = -log2(e)/(2 s^2)
typedef __attribute__((ext_vector_type(2))) float f32x2;

struct Rbf {
  float mu[kMaxK];
  float c2[kMaxK];
  float alpha[kMaxK];
  float w[kMaxK];
  // packed form for v_pk_* math: exp(-(c-mu)^2/(2 s^2)) = exp2(-(c*sq - mu*sq)^2), sq = sqrt(log2(e)/(2 s^2));
  // kernels are processed two at a time, an odd K gets a dummy partner (sq = 0) whose sum is ignored
  f32x2 sq2[kMaxK / 2];
  f32x2 msq2[kMaxK / 2];
  // Equally spaced kernels of one width (the reference's ten soft-match kernels, mu = 0.9 .. -0.9, sigma = 0.1, behind the
  // exact-match kernel 0): evaluated middle-out by recurrence instead of one exp2 each (rbf_block_geo below).  `geo` is decided
  // per launch from the parameter VALUES (load_rbf): any other kernel set runs the direct form.  The recurrence's constants
  // take the PLACE of the direct form's (no register beside them — the two-wavefront E = 128 kernel has none to spare):
  //   sq2[0], msq2[0][0]: as in the direct form (kernel 0 stays direct; sq2[0][1] = s of kernels 1 .. 10)
  //   sq2[1] = {-mu5 s, -mu6 s}: t = c s + sq2[1] = the scaled distances to the two middle kernels
  //   sq2[2] = {2 delta, -2 delta}, delta = (mu5 - mu6) s: exponent slope of the neighbour ratios;  sq2[3][0] = -delta^2
  //   msq2[1][0], msq2[1][1], msq2[2][0] = g, g^3, g^6, g = 2^(-2 delta^2): what the running sums of steps 3 .. 5 are short of
  bool geo = false;
};

// nk <= K real kernels (nk < K: the generic kernel's run-time kernel count); the rest are dummies
template <int K>
__device__ __forceinline__ void pack_rbf(Rbf& rbf, int nk = K) {
#pragma unroll
  for (int kp = 0; kp < (K + 1) / 2; ++kp) {
    float sq[2], msq[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = 2 * kp + u;
      sq[u] = (k < K && k < nk) ? sqrtf(-rbf.c2[k]) : 0.0f;
      msq[u] = (k < K && k < nk) ? rbf.mu[k] * sq[u] : 1.0e3f;  // dummy: exp2(-(0*c - 1e3)^2) = 0 for every c
    }
    rbf.sq2[kp] = f32x2{sq[0], sq[1]};
    rbf.msq2[kp] = f32x2{msq[0], msq[1]};
  }
}

__device__ __forceinline__ float sload_f32(const float* base, int idx) {
  return __builtin_bit_cast(float, sload_u32(base, idx));
}

// K wave-uniform floats through the scalar cache with ONE wait (a wait per value made the 44 parameter loads of a
// wavefront's prologue a chain of 44 scalar-cache round trips — microseconds before the first LDS-DMA was issued,
// which shows on eval.py-sized calls and on TKL's short stage-1 launches)
template <int K>
__device__ __forceinline__ void sload_vec(const float* base, float (&out)[kMaxK]) {
  static_assert(K == 11, "instantiated for the reference's 11 kernels");
  uint32_t v[11];
  asm volatile(
      "s_load_dword %0, %11, 0x0\n\t"
      "s_load_dword %1, %11, 0x4\n\t"
      "s_load_dword %2, %11, 0x8\n\t"
      "s_load_dword %3, %11, 0xc\n\t"
      "s_load_dword %4, %11, 0x10\n\t"
      "s_load_dword %5, %11, 0x14\n\t"
      "s_load_dword %6, %11, 0x18\n\t"
      "s_load_dword %7, %11, 0x1c\n\t"
      "s_load_dword %8, %11, 0x20\n\t"
      "s_load_dword %9, %11, 0x24\n\t"
      "s_load_dword %10, %11, 0x28\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(v[0]), "=&s"(v[1]), "=&s"(v[2]), "=&s"(v[3]), "=&s"(v[4]), "=&s"(v[5]), "=&s"(v[6]), "=&s"(v[7]), "=&s"(v[8]),
        "=&s"(v[9]), "=&s"(v[10])
      : "s"(base));
#pragma unroll
  for (int k = 0; k < 11; ++k) out[k] = __builtin_bit_cast(float, v[k]);
}

// Kernel parameters are wave-uniform: fetch them through the scalar cache (SGPRs, no vmcnt traffic
// that would make the compiler drain the LDS-DMA queue inside the block loop).
// Is the kernel set the middle-out recurrence's (K = 11: kernel 0 anything, kernels 1 .. 10 of ONE width and equally spaced,
// descending)?  Wave-uniform, decided from the parameter values once per wavefront.  Also required: the two middle kernels
// must not underflow for any cosine in [-1, 1] (every other kernel is reached from them by multiplication: a zero there would
// zero kernels that are not), and the neighbour ratios must stay finite under the clamp |t| <= kGeoClamp.
template <int K>
__device__ __forceinline__ void detect_geo(Rbf& rbf, const float (&sg)[kMaxK]) {
  rbf.geo = false;
  if constexpr (K == 11 && MM_RBF_GEO) {
    const float dmu = rbf.mu[5] - rbf.mu[6];
    bool ok = dmu > 0.0f;
#pragma unroll
    for (int k = 1; k < 10; ++k) ok = ok && fabsf((rbf.mu[k] - rbf.mu[k + 1]) - dmu) <= 1.0e-6f && sg[k] == sg[k + 1];
    const float s = rbf.sq2[0][1];
    const float delta = dmu * s;
    const float reach = (1.0f + fmaxf(fabsf(rbf.mu[5]), fabsf(rbf.mu[6]))) * s;   // largest |t| of a middle kernel over [-1, 1]
    ok = ok && reach * reach < 120.0f && 2.0f * delta * kGeoClamp < 100.0f;
    const bool on = __builtin_amdgcn_readfirstlane((int)ok) != 0;
    rbf.geo = on;
    if (on) {   // wave-uniform
      rbf.sq2[1] = f32x2{-rbf.mu[5] * s, -rbf.mu[6] * s};
      rbf.sq2[2] = f32x2{2.0f * delta, -2.0f * delta};
      rbf.sq2[3][0] = -delta * delta;
      const float l2g = -2.0f * delta * delta;
      rbf.msq2[1] = f32x2{__builtin_amdgcn_exp2f(l2g), __builtin_amdgcn_exp2f(3.0f * l2g)};
      rbf.msq2[2][0] = __builtin_amdgcn_exp2f(6.0f * l2g);
    }
  }
}

template <int K, bool GEO = true>
__device__ __forceinline__ void load_rbf(const float* mu, const float* sigma, const float* alpha, const float* w, Rbf& rbf) {
  float sg[kMaxK];
  if constexpr (K == 11) {
    sload_vec<K>(sigma, sg);
    sload_vec<K>(mu, rbf.mu);
    if (alpha) sload_vec<K>(alpha, rbf.alpha);
    if (w) sload_vec<K>(w, rbf.w);
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if constexpr (K != 11) {
      sg[k] = sload_f32(sigma, k);
      rbf.mu[k] = sload_f32(mu, k);
      rbf.alpha[k] = alpha ? sload_f32(alpha, k) : 1.0f;
      rbf.w[k] = w ? sload_f32(w, k) : 0.0f;
    } else {
      if (!alpha) rbf.alpha[k] = 1.0f;
      if (!w) rbf.w[k] = 0.0f;
    }
    rbf.c2[k] = -1.4426950408889634f / (2.0f * sg[k] * sg[k]);
  }
  pack_rbf<K>(rbf);
  if constexpr (GEO) detect_geo<K>(rbf, sg);
}

// Epilogue of one 32-token document block: cosine scaling + K RBF kernels, summed into pk[k].
// acc[i]: raw dot of document row rowof(i)+4h with this lane's query token; rdr[i]: 1/(|d|+tiny) of
// that row; vbits (already shifted by 4h): bit rowof(i) set <=> the row is a real token.
// Packed: per row one select (a masked row gets cosine 1e5, which underflows every kernel to exactly
// 0, the same contribution as the reference's multiply by the 0 mask) and per kernel PAIR
// v_pk_fma + v_pk_mul + 2 v_exp + v_pk_add.  The fp32 MFMA shares the SIMD's FMA lanes with the
// VALU (measured: zero overlap, profiles/r01_kernel_pool_pmc.json), so every VALU op removed here
// is wall time.
// W: lw[i] = log2(gate of row i) rides in the exponent (exp2(x + log2 g) = g exp2(x); g = 0 -> -inf -> 0),
// which turns the pk_mul of the square into a pk_fma: the gate costs no instruction.
// KP0..KP1: the kernel pairs this call evaluates, G0..G1: the 8-row groups (a K-split workgroup shares the
// epilogue between its two waves either by kernel pairs or by rows).
// the packed constants alone (for callers that fetch them per block instead of holding a whole Rbf in registers)
struct RbfPk {
  f32x2 sq2[kMaxK / 2];
  f32x2 msq2[kMaxK / 2];
};


// ---- middle-out recurrence over equally spaced kernels (Rbf::Geo) ----------------------------------------------------------
// exp2(-(t - j delta)^2) = exp2(-t^2) * u^j * g^(j (j - 1) / 2) with u = exp2(2 delta t - delta^2), g = exp2(-2 delta^2): from the
// two middle kernels (5, 6) the pairs (4, 7), (3, 8), (2, 9), (1, 10) are each ONE packed multiply by the running ratio pair
// (u, d) away — the constant factor g^(j (j - 1) / 2) is left out of the running sums and applied once per pair (pk_get) —
// so a cosine costs 4 + 1 v_exp_f32 and 12 packed + 5 plain VALU instead of 12 v_exp_f32 and 18 packed (by the per-instruction
// issue costs of profiles/r01_inst_cost_ubench.txt: ~134 instead of ~206 cycles).  Kernel 0 (the exact-match kernel, sigma
// 1e-3) keeps the direct form.  Sums land in geo order: pk2[0] = (k0, -), pk2[j] = (k(6 - j), k(5 + j)), j = 1 .. 5.
// A masked row arrives as cosine 1e5: its t is clamped to kGeoClamp, where exp2(-t^2) is exactly 0 and the ratios are finite:
// 0 x finite = exactly 0 in every kernel, as in the direct form.  Real cosines (|c| <= 1) never reach the clamp.
// Accuracy: the kernel j steps from the middle carries j ratio roundings, ~5e-6 relative at j = 4 against ~4e-7 direct — a
// twentieth of what the split-bf16 cosine itself contributes to an activation (1e-4), above the exact-fp32 twins' own error:
// those load their constants with load_rbf<K, false> and stay direct.
template <bool W>
__device__ __forceinline__ void rbf_geo_one(f32x2 (&pk2)[kMaxK / 2], float c, float lw, const Rbf& rbf) {
  const f32x2 cc = {c, c};
  const float s = rbf.sq2[0][1];
  f32x2 t = cc * f32x2{s, s} + rbf.sq2[1];
  t = f32x2{__builtin_amdgcn_fmed3f(t[0], -kGeoClamp, kGeoClamp), __builtin_amdgcn_fmed3f(t[1], -kGeoClamp, kGeoClamp)};
  const f32x2 lwv = {lw, lw};
  const f32x2 av = W ? lwv - t * t : -(t * t);
  const f32x2 rv = t * rbf.sq2[2] + f32x2{rbf.sq2[3][0], rbf.sq2[3][0]};
  f32x2 e = {__builtin_amdgcn_exp2f(av[0]), __builtin_amdgcn_exp2f(av[1])};
  const f32x2 ud = {__builtin_amdgcn_exp2f(rv[0]), __builtin_amdgcn_exp2f(rv[1])};
  pk2[1] += e;
#pragma unroll
  for (int j = 2; j <= 5; ++j) {
    e *= ud;
    pk2[j] += e;
  }
  const float sv = c * rbf.sq2[0][0] - rbf.msq2[0][0];
  const float a0 = W ? lw - sv * sv : -(sv * sv);
  pk2[0][0] += __builtin_amdgcn_exp2f(a0);
}

template <bool W, int G0 = 0, int G1 = 4>
__device__ __forceinline__ void rbf_block_geo(f32x2 (&pk2)[kMaxK / 2], const f32x16& acc, const float (&rdr)[16], float rq,
                                              uint32_t va, int h, const Rbf& rbf, const float* lw) {
  const uint32_t vbits = va >> (4 * h);
#pragma unroll
  for (int g = G0; g < G1; ++g) {
    const uint32_t gm = (va >> (8 * g)) & 0xffu;
    if (gm == 0) continue;
    const bool full = gm == 0xffu;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const int i = 4 * g + ii;
      float c = (acc[i] * rq) * rdr[i];
      if (!full) c = ((vbits >> rowof(i)) & 1u) ? c : 1.0e5f;
      rbf_geo_one<W>(pk2, c, W ? lw[i] : 0.0f, rbf);
    }
  }
}

// kernel k's pooled sum out of the running sums, in whichever order the epilogue filed them
template <int K>
__device__ __forceinline__ void pk_get(float (&pk)[kMaxK], const f32x2 (&pk2)[kMaxK / 2], const Rbf& rbf) {
  if constexpr (K == 11 && MM_RBF_GEO) {
    if (rbf.geo) {
      const float gs[5] = {1.0f, 1.0f, rbf.msq2[1][0], rbf.msq2[1][1], rbf.msq2[2][0]};
      pk[0] = pk2[0][0];
#pragma unroll
      for (int j = 1; j <= 5; ++j) {
        pk[6 - j] = j > 2 ? pk2[j][0] * gs[j - 1] : pk2[j][0];
        pk[5 + j] = j > 2 ? pk2[j][1] * gs[j - 1] : pk2[j][1];
      }
      return;
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) pk[k] = pk2[k >> 1][k & 1];
}

template <int K, bool W = false, int KP0 = 0, int KP1 = (K + 1) / 2, int G0 = 0, int G1 = 4, typename R = Rbf>
__device__ __forceinline__ void rbf_block(f32x2 (&pk2)[kMaxK / 2], const f32x16& acc, const float (&rdr)[16], float rq,
                                          uint32_t va, int h, const R& rbf, const float* lw = nullptr) {
  if constexpr (K == 11 && MM_RBF_GEO && KP0 == 0 && KP1 == (K + 1) / 2 && __is_same(R, Rbf)) {
    if (rbf.geo) {   // wave-uniform
      rbf_block_geo<W, G0, G1>(pk2, acc, rdr, rq, va, h, rbf, lw);
      return;
    }
  }
  // va (wave-uniform): bit r set <=> row r of the block is a real token.  Accumulator registers
  // 4g..4g+3 hold rows 8g..8g+7 (both lane halves), so a group with no real row is skipped as a
  // whole (the last block of a document: D = 200 -> 8 of 32 rows) and a group of 8 real rows needs
  // no per-row select.
  const uint32_t vbits = va >> (4 * h);
#pragma unroll
  for (int g = G0; g < G1; ++g) {
    const uint32_t gm = (va >> (8 * g)) & 0xffu;
    if (gm == 0) continue;
    const bool full = gm == 0xffu;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const int i = 4 * g + ii;
      float c = (acc[i] * rq) * rdr[i];
      if (!full) c = ((vbits >> rowof(i)) & 1u) ? c : 1.0e5f;
      const f32x2 cc = {c, c};
      const f32x2 lwv = W ? f32x2{lw[i], lw[i]} : f32x2{0.0f, 0.0f};
#pragma unroll
      for (int kp = KP0; kp < KP1; ++kp) {
        const f32x2 sv = cc * rbf.sq2[kp] - rbf.msq2[kp];
        const f32x2 av = W ? lwv - sv * sv : -(sv * sv);
        const f32x2 e = {__builtin_amdgcn_exp2f(av[0]), __builtin_amdgcn_exp2f(av[1])};
        pk2[kp] += e;
      }
    }
  }
}

// log2 of a TK-Sparse gate (a ReLU output, so >= 0; anything below 0 is treated as 0)
__device__ __forceinline__ float gate_log2(float g) { return __builtin_amdgcn_logf(fmaxf(g, 0.0f)); }

// The same epilogue on a REDISTRIBUTED tile: this lane evaluates ROWS consecutive document rows of ONE query token
// (c[j]: scaled cosines; bit j of `bits`: row j is a real token; lw: log2 gates of those rows when W).
// Used when the query is short: a Q-token tile keeps only Q of 32 lanes busy in rbf_block, so the tile is
// transposed through LDS and NP = 32 / ROWS lanes share each query token.
template <int K, bool W, int ROWS, typename R = Rbf>
__device__ __forceinline__ void rbf_rows(f32x2 (&pk2)[kMaxK / 2], const float (&c)[ROWS], uint32_t bits, const R& rbf,
                                         const float (&lw)[ROWS]) {
  if constexpr (K == 11 && MM_RBF_GEO && __is_same(R, Rbf)) {
    if (rbf.geo) {   // wave-uniform
#pragma unroll
      for (int j = 0; j < ROWS; ++j) rbf_geo_one<W>(pk2, ((bits >> j) & 1u) ? c[j] : 1.0e5f, lw[j], rbf);
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < ROWS; ++j) {
    const float cj = ((bits >> j) & 1u) ? c[j] : 1.0e5f;
    const f32x2 cc = {cj, cj};
    const f32x2 lwv = W ? f32x2{lw[j], lw[j]} : f32x2{0.0f, 0.0f};
#pragma unroll
    for (int kp = 0; kp < K / 2; ++kp) {
      const f32x2 sv = cc * rbf.sq2[kp] - rbf.msq2[kp];
      const f32x2 av = W ? lwv - sv * sv : -(sv * sv);
      const f32x2 e = {__builtin_amdgcn_exp2f(av[0]), __builtin_amdgcn_exp2f(av[1])};
      pk2[kp] += e;
    }
    if constexpr (K & 1) {   // the last kernel of an odd K alone: 4 scalar instructions instead of 5 with a dummy partner's v_exp
      const float sv = cj * rbf.sq2[K / 2][0] - rbf.msq2[K / 2][0];
      const float av = W ? lw[j] - sv * sv : -(sv * sv);
      pk2[K / 2][0] += __builtin_amdgcn_exp2f(av);
    }
  }
}

// Token stride of the transposed tile in floats.  With 32 the 3-lanes-per-token form (ROWS = 11, single-float reads
// at T[t * 32 + 11 s + j]) put all even tokens on one bank and all odd tokens on another: ~10-way conflicts,
// SQ_LDS_BANK_CONFLICT = 198.9 M cycles per dispatch in profiles/r02_tk_pmc.json (~444 cycles per 32-row block).
// 36 keeps the rows 16-byte aligned (the f32x4 writes of the MFMA layout and the ROWS = 4 / 8 reads) and spreads
// the tokens: 36 t mod 64 walks the multiples of 4 with period 16, and the row-group offsets 0 / 11 / 22 fall
// into different classes mod 4, so at most two of the 63 lanes share a bank.
constexpr int kTS = 36;

// read this lane's ROWS values of the transposed tile T[token][kTS >= 32 rows] (and of the gate vector) and evaluate them:
// lane = (token t, row group s), rows s * ROWS .. s * ROWS + ROWS - 1 (rows >= 32 do not exist: their bits are 0).
// ROWS need not divide 32: a 20-token query puts 3 lanes on a token (11 + 11 + 10 rows).
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};

// `loaded()` runs after this lane's values have been read from T and before they are evaluated (a caller that borrowed a
// ring slot for T can hand it back to its producer there)
template <int K, bool W, int ROWS, typename Hook = NoHook, typename R = Rbf>
__device__ __forceinline__ void rbf_redistributed(f32x2 (&pk2)[kMaxK / 2], const float* T, const float* lwrow, int t, int s,
                                                  uint32_t va, const R& rbf, Hook loaded = Hook()) {
  const int row0 = s * ROWS;
  const float* src = T + t * kTS + row0;
  float c[ROWS], lw[ROWS];
  if constexpr (ROWS % 4 == 0) {
#pragma unroll
    for (int v = 0; v < ROWS / 4; ++v) {
      const f32x4 x = *(const f32x4*)(src + 4 * v);
      c[4 * v] = x[0]; c[4 * v + 1] = x[1]; c[4 * v + 2] = x[2]; c[4 * v + 3] = x[3];
      if (W) {
        const f32x4 y = *(const f32x4*)(lwrow + row0 + 4 * v);
        lw[4 * v] = y[0]; lw[4 * v + 1] = y[1]; lw[4 * v + 2] = y[2]; lw[4 * v + 3] = y[3];
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      const int rr = row0 + j < 32 ? row0 + j : 31;      // stay inside the tile / the gate row; the bit of a row >= 32 is 0
      c[j] = T[t * kTS + rr];
      if (W) lw[j] = lwrow[rr];
    }
  }
  if (!W) {
#pragma unroll
    for (int j = 0; j < ROWS; ++j) lw[j] = 0.0f;
  }
  const uint32_t bits = (row0 < 32 ? (va >> row0) : 0u) & ((ROWS == 32) ? 0xffffffffu : ((1u << ROWS) - 1u));
  loaded();
  rbf_rows<K, W, ROWS>(pk2, c, bits, rbf, lw);
}

// rows per lane of the redistributed epilogue for a query of qn real tokens: the smallest ROWS of the instantiated set
// whose ceil(32 / ROWS) lanes per token fit all qn tokens into the 64 lanes (0 = keep the MFMA layout: 16 rows per lane)
__device__ __forceinline__ int redist_rows(int qn) {
  if (qn <= 0 || qn > 21) return 0;
  if (qn <= 2) return 1;
  if (qn <= 4) return 2;
  if (qn <= 8) return 4;
  if (qn <= 16) return 8;      // (5 / 6 / 7 rows on 7 / 6 / 5 lanes for qn = 9 .. 12 measured slower than 8 rows on 4: unaligned LDS reads)
  return 11;                   // 17 .. 21 tokens: 3 lanes per token, +1.3 % at Q = 20
}

template <int K, bool W, typename Hook = NoHook, typename R = Rbf>
__device__ __forceinline__ void rbf_redistributed_rows(int rows, f32x2 (&pk2)[kMaxK / 2], const float* T, const float* lwrow,
                                                       int t, int s, uint32_t va, const R& rbf, Hook loaded = Hook()) {
  switch (rows) {
    case 1: rbf_redistributed<K, W, 1>(pk2, T, lwrow, t, s, va, rbf, loaded); break;
    case 2: rbf_redistributed<K, W, 2>(pk2, T, lwrow, t, s, va, rbf, loaded); break;
    case 4: rbf_redistributed<K, W, 4>(pk2, T, lwrow, t, s, va, rbf, loaded); break;
    case 8: rbf_redistributed<K, W, 8>(pk2, T, lwrow, t, s, va, rbf, loaded); break;
    default: rbf_redistributed<K, W, 11>(pk2, T, lwrow, t, s, va, rbf, loaded); break;
  }
}

// partial sums of the np lanes of a token -> its first lane (s == 0); np = lanes per token (any value <= 32)
template <int K>
__device__ __forceinline__ void redist_reduce(float (&pk)[kMaxK], int np, int lane) {
  if ((np & (np - 1)) == 0) {
    for (int o = np >> 1; o >= 1; o >>= 1) {
#pragma unroll
      for (int k = 0; k < K; ++k) pk[k] += __shfl_xor(pk[k], o, 64);
    }
  } else {
    float acc[kMaxK];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = pk[k];
    for (int o = 1; o < np; ++o) {
#pragma unroll
      for (int k = 0; k < K; ++k) acc[k] += __shfl(pk[k], lane + o, 64);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) pk[k] = acc[k];
  }
}


// lane i of a 16-lane row += lane i + n (DPP row_shl:n, lanes past the row read 0) for n = 8, 4, 2, 1: the row's sum in its lane 0,
// added in the order of the xor butterfly as lane 0 sees it
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_sum_to_lane0(float v) {
  v += dpp_row<0x108>(v);
  v += dpp_row<0x104>(v);
  v += dpp_row<0x102>(v);
  v += dpp_row<0x101>(v);
  return v;
}

// log-sum pooling of one pair over kernels K0..K1-1: pk[k] (this lane's query token, both halves already
// combined); writes per_kernel, returns sum_k w_k * pooled_k (valid in LANE 0, the lane every caller stores from).
// tok: the query token whose sums this lane holds (-1: a lane that holds a copy or nothing) — only used to hand the sums out
template <int K, int K0 = 0, int K1 = K>
__device__ __forceinline__ float pool_partial(const KpArgs& a, int64_t pair, const float (&pk)[kMaxK], bool count_lane,
                                              int lane, const Rbf& rbf, int tok) {
  if (a.pooled && tok >= 0 && tok < a.Q) {
    float* dst = a.pooled + (pair * a.Q + tok) * K;
#pragma unroll
    for (int k = K0; k < K1; ++k) dst[k] = pk[k];
  }
  // The K wave sums run TOGETHER, butterfly step by butterfly step: K independent exchanges per step overlap their latencies.
  // (One wave_sum per kernel inside the loop — with the per_kernel store's branch between them — was six dependent cross-lane
  // round trips per kernel: 20 k cycles per pair for the three combinations of Conv-KNRM's loop kernel, a fifth of its time, and
  // 6-7 k of a TK pair's ~85 k; tools/build_variant.sh mprof kernel_pool128 -DMM_KP_MULTI_PROF.)  Every value goes through the
  // same additions in the same order as wave_sum: the sums are the same bits.
#if defined(MM_POOL_SERIAL)   // A/B builds: one wave_sum per kernel, as up to round 6
  float total_s = 0.0f;
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    float lgs = __logf(fmaxf(pk[k] * rbf.alpha[k], a.clamp_min));
    lgs = count_lane ? lgs : 0.0f;
    const float ssum = wave_sum(lgs);
    if (a.per_kernel && lane == 0) a.per_kernel[pair * K + k] = ssum;
    total_s += rbf.w[k] * ssum;
  }
  return total_s;
#endif
  float lg[K1 - K0];
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    const float v = __logf(fmaxf(pk[k] * rbf.alpha[k], a.clamp_min));
    lg[k - K0] = count_lane ? v : 0.0f;  // exactly one lane per real query token counts (the others hold copies)
  }
  // ... and only LANE 0's sums are used: the exchanges over 32 and 16 lanes stay butterflies (ds_bpermute), the steps inside a
  // row of 16 are v_add_f32 with a DPP row_shl operand (no LDS round trip) — in lane 0 the same additions of the same partial sums.
#pragma unroll
  for (int o = 32; o >= 16; o >>= 1) {
#pragma unroll
    for (int k = 0; k < K1 - K0; ++k) lg[k] += __shfl_xor(lg[k], o, 64);
  }
#pragma unroll
  for (int k = 0; k < K1 - K0; ++k) lg[k] = row_sum_to_lane0(lg[k]);
  if (a.per_kernel && lane == 0) {
#pragma unroll
    for (int k = K0; k < K1; ++k) a.per_kernel[pair * K + k] = lg[k - K0];
  }
  float total = 0.0f;
#pragma unroll
  for (int k = K0; k < K1; ++k) total += rbf.w[k] * lg[k - K0];
  return total;
}

template <int K>
__device__ __forceinline__ void finish_pool(const KpArgs& a, int64_t pair, const float (&pk)[kMaxK], bool count_lane,
                                            int lane, const Rbf& rbf, int tok) {
  const float total = pool_partial<K>(a, pair, pk, count_lane, lane, rbf, tok);
  if (lane == 0) a.out[pair] = total;
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
  const bf16x2 v = __builtin_convertvector(f32x2{a, b}, bf16x2);  // v_cvt_pk_bf16_f32 (RNE)
  return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, bf16x8& hi, bf16x8& lo) {
  u32x4 hw, lw;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = j < 2 ? x0[2 * j] : x1[2 * j - 4];
    const float b = j < 2 ? x0[2 * j + 1] : x1[2 * j - 3];
    const uint32_t w = cvt_pk_bf16(a, b);
    const float ha = __uint_as_float(w << 16), hb = __uint_as_float(w & 0xffff0000u);
    hw[j] = w;
    lw[j] = cvt_pk_bf16(a - ha, b - hb);
  }
  hi = __builtin_bit_cast(bf16x8, hw);
  lo = __builtin_bit_cast(bf16x8, lw);
}

// Three-term split x = hi + lo + c (3 x 8 significand bits = fp32's 24): the fp32 MaxSim path ranks candidates, and
// the two-term split's 2^-17 operand error moves ~10x more near-ties than a true fp32 evaluation would.
__device__ __forceinline__ void split8x3(const f32x4& x0, const f32x4& x1, bf16x8& hi, bf16x8& lo, bf16x8& c) {
  u32x4 hw, lw, cw;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = j < 2 ? x0[2 * j] : x1[2 * j - 4];
    const float b = j < 2 ? x0[2 * j + 1] : x1[2 * j - 3];
    const uint32_t w = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(w << 16), rb = b - __uint_as_float(w & 0xffff0000u);
    const uint32_t l = cvt_pk_bf16(ra, rb);
    hw[j] = w;
    lw[j] = l;
    cw[j] = cvt_pk_bf16(ra - __uint_as_float(l << 16), rb - __uint_as_float(l & 0xffff0000u));
  }
  hi = __builtin_bit_cast(bf16x8, hw);
  lo = __builtin_bit_cast(bf16x8, lw);
  c = __builtin_bit_cast(bf16x8, cw);
}

__device__ __forceinline__ float sumsq4(const f32x4& v) { return v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]; }


// Pin a fragment in accumulator registers: the MFMA reads its B operand from AGPRs directly.  Left to itself the
// register allocator keeps query fragments in VGPRs, runs out, spills the overflow to AGPRs and reloads it with four
// v_accvgpr_read before every use (dot_topk.hip: 192 of 387 instructions of the K loop).
__device__ __forceinline__ bf16x8 to_agpr(bf16x8 v) {
  bf16x8 r;
  asm volatile("" : "=a"(r) : "0"(v));
  return r;
}

__device__ __forceinline__ f32x16 mfma_bf16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// TKL: the wavefront that processes packed chunk p publishes its slot-map entry (see KpArgs::slot2p)
__device__ __forceinline__ void tkl_publish_slot(const KpArgs& a, int64_t p, int blocks, int lane) {
  if (a.slot2p && lane == 0) {
    const int sl = a.chunk_slot[p];
    if (sl >= 0 && sl < a.n_slots) a.slot2p[sl] = (int32_t)((p << 2) | blocks);
  }
}

// tkl_stage1_rows.hip: TKL stage 1 with the cosine hand-off, whole chunk rows streamed (E = 100 / 200 / 300, Q <= 32)
bool tkl_stage1_rows_supported(int Q, int E);
int tkl_stage1_rows_launch(const KpArgs& a, hipStream_t stream);
// tkl_stage1_ksplit.hip: the same with two wavefronts per workgroup sharing every tile along K (two wavefronts per SIMD)
bool tkl_stage1_ksplit_supported(int Q, int E);
int tkl_stage1_ksplit_launch(const KpArgs& a, hipStream_t stream);

// kernel_pool128.hip: streaming kernels for E = 64n <= 384 (Q <= 32)
bool kp128_supported(int Q, int D, int E, bool gated);
int kp128_launch(const KpArgs& a, hipStream_t stream);


}  // namespace mm
