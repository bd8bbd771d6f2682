// Backward of TK kernel pooling on the bf16 matrix pipe: kp_bwd_split_kernel (round 6).
//
// The exact-f32 tiled kernel (kernel_pool_bwd.hip) keeps ONE 1,024-thread workgroup per CU around 113 KB of LDS and walks a
// pair through ~30 dependent phases and 63 barriers: 253 k cycles per pair, the matrix pipe busy for 45 k of them
// (profiles/r04_experiments/tk_bwd_phases.txt).  This kernel is built the other way round:
//
//   * one 4-wavefront workgroup per pair, <= 74 KB of LDS and <= 256 registers: TWO workgroups per CU, so the barrier and
//     latency stalls of one pair are the other pair's issue slots;
//   * wavefront t owns the 16-column granules t, t + 4, t + 8, ... of E in ALL THREE products (E = 300: 5, 5, 5 and 4 granules),
//     so its slice of a document block ([32 rows][its columns], fp32, LDS-DMA: global -> LDS without registers) is
//     wavefront-private — no barrier guards it;
//   * the three products run as split-bf16 MFMAs (x = hi + lo; hi.hi + lo.hi + hi.lo with fp32 accumulation: operand error
//     2^-17, the forward's own scheme): cosines on v_mfma_f32_32x32x16_bf16 (one K step per granule), the two gradient
//     products on v_mfma_f32_16x16x32_bf16 (one 16-column tile per granule, K = all 32 query tokens / all 32 rows of the block)
//     — 15 + 60 MFMAs of 32 / 16 cycles per block and wavefront instead of 600 f32 MFMAs of 64 per block;
//   * the query operands live in registers for the whole pair (B fragments of the cosine product, K = E, and of
//     grad_d = G^T Q, K = query tokens: 8 + 8 registers per granule, + 8 accumulators of grad_q);
//   * cosines: K = E is split over the four wavefronts; the four partial 32 x 32 tiles meet in LDS (each wavefront keeps the
//     quarter it evaluates in registers and publishes the other three: 12 KB), barrier 1;
//   * wavefront t then owns rows 8t..8t+7 of the block for the elementwise part: lane (query token i, half h) holds rows
//     8t + 4h + 0..3: cosine scaling, the K RBF kernels and their derivative, G = d loss / d c, the two norm-gradient sums;
//     G is published as MFMA A operands (bf16 hi / lo) in BOTH contraction orders — GT[row][token] for grad_d, GS[token][row]
//     for grad_q (10 KB) — barrier 2;
//   * grad_d tile = GT x Q-fragments, minus the row's own-direction term, stored straight from the accumulators;
//     grad_q tile += GS x (the block's rows, read as 8 floats per lane in the accumulator's row order — the SAME eight values
//     serve grad_d's own-direction term and, split to bf16, grad_q's B operand: the K order of an MFMA is free as long as
//     both operands agree, so the document block is never transposed);
//   * the next block's LDS-DMA is issued as soon as the last of those reads has returned.
//
// The pooled kernel sums pkq[i][k] (every position of the document enters each of them) must exist before any gradient:
// MODE 1 takes them as an input — mm_kernel_pool_ex_fwd2 hands them out of the forward, 44 bytes per query token — and
// MODE 0 is the same kernel's pooling-only form that produces them when the caller has none (the document then crosses HBM
// twice, as in the tiled kernel).
//
//   c_ij = ((q_i . d_j) rq_i) rd_j,   e_ijk = exp2((c_ij - mu_k)^2 c2_k),   pkq_ik = sum_j m_j e_ijk     (m = mask x gate)
//   A_ik = g qmask_i w_k [alpha_k pkq_ik >= floor] / pkq_ik
//   G_ij = m_j sum_k A_ik e_ijk (-(c_ij - mu_k) / sigma_k^2)
//   grad_d_j = sum_i (G_ij rq_i rd_j) q_i - (sum_i G_ij c_ij) rd_j / |d_j| d_j
//   grad_q_i = rq_i (sum_j (G_ij rd_j) d_j - (sum_j G_ij c_ij) / |q_i| q_i)
#include "mm_internal.h"
#include "kp_device.h"
#include "kp_bwd.h"

namespace mm {

constexpr int kGRow = 80;                      // bytes per row of the published G operands: 32 bf16 + 16 (odd number of 16-B units)
constexpr int kGBytes = 32 * kGRow;            // one operand array (hi or lo)
constexpr int kPBytes = 4 * 3 * 8 * 32 * 4;    // partial cosine tiles: [quarter][3 foreign wavefronts][8 rows][32 tokens] fp32

// A wavefront's slice of a 32-row block in LDS: CHUNK-major, [NGW * 4 chunks of 16 bytes][32 row slots], row j of chunk c in slot
// j ^ (c & 3).  LDS-DMA writes lane-linearly (instruction n fills chunks 2n, 2n + 1: lane L -> chunk 2n + (L >> 5), slot L & 31),
// so the swizzle sits on the SOURCE side: the lane fetches row (L & 31) ^ (c & 3).  Reads: the cosine A operand of row j takes
// 16 bytes of chunks c, c + 1 — the sixteen lanes of a ds_read_b128 phase hold sixteen distinct slots of ONE chunk: 64 distinct
// banks, no padding; the column reads of the gradient products (lane = column e of a granule and row group kg: 4 bytes of rows
// 4 kg + x) would put the four chunks of a granule on one bank without the swizzle; with it the 32 lanes of a ds_read_b32
// phase cover 32 banks.  No table of source offsets in registers: two per-lane offsets (even / odd instruction) + 256 bytes per granule.
template <int NGW>
struct BwdGeo {
  static constexpr int NC = NGW * 4;                   // chunks per row of the slice
  static constexpr int NI = NGW * 2;                   // LDS-DMA instructions per slice (two chunks each)
  static constexpr int SLICE = NC * 32 * 16;
  static constexpr int LDS = 4 * SLICE + kPBytes + 512 /*NP*/ + 4 * kGBytes + 128 /*S2*/ + 3 * 128 /*RQ NQ F2*/ + 32 * 48 /*AK*/;
};

// ONE LDS-DMA instruction: 64 lanes x 16 bytes, global (sbase + voff) -> LDS (m0 + 16 lane)
__device__ __forceinline__ void glds16(const char* sbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}

// barrier for LDS traffic only: __syncthreads() would also drain vmcnt — the LDS-DMA prefetch and the gradient stores in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Sum over the 32 lanes of a half-wavefront, delivered to its LAST lane (31 / 63) only: five DPP adds (row_shr 1, 2, 4, 8 inside
// the 16-lane rows, then row_bcast:15 into rows 1 and 3) instead of five dependent ds_bpermute round trips (~120 cycles each).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float half_sum_last(float v) {
  v = dpp_add<0x111, 0xf>(v);   // row_shr:1
  v = dpp_add<0x112, 0xf>(v);   // row_shr:2
  v = dpp_add<0x114, 0xf>(v);   // row_shr:4
  v = dpp_add<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of every row holds the row's sum
  v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1, 3 -> lanes 31, 63 hold their half's sum
  return v;
}

__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// Phase clocks (tools/build_variant.sh phases kernel_pool_bwd_split -DMM_KP_BWD_PHASE_TIMES=1; tools/bench_kp_bwd.py --phases): lane 0
// of wavefronts 0 and 3 of the MIDDLE pair sums s_memtime deltas per phase and overwrites that pair's grad_w / grad_alpha rows.
#ifndef MM_KP_BWD_PHASE_TIMES
#define MM_KP_BWD_PHASE_TIMES 0
#endif
#if MM_KP_BWD_PHASE_TIMES
#define KPS_PH(k) do { const long long t_ = __builtin_readcyclecounter(); ph[k] += (float)(t_ - t_last); t_last = t_; } while (0)
#else
#define KPS_PH(k) do { } while (0)
#endif

template <int MODE, bool GATE, int NGW>
__global__ void __launch_bounds__(256, 2) kp_bwd_split_kernel(const KpBwdArgs a, const float* __restrict__ pkq_in, float* __restrict__ pkq_out) {
  constexpr int K = 11;
  using Geo = BwdGeo<NGW>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int t = __builtin_amdgcn_readfirstlane(tid >> 6);     // wavefront = granule owner = row-quarter owner
  const int lane = tid & 63, r_ = lane & 31, h_ = lane >> 5;
  // small batches: S workgroups per pair, this one takes the blocks sp, sp + S, ... (S = 1: the whole pair)
  const int S = MODE ? a.nsplit : 1;
  const int64_t pair = S > 1 ? (int64_t)(blockIdx.x / (unsigned)S) : (int64_t)blockIdx.x;
  const int sp = S > 1 ? (int)(blockIdx.x - (unsigned)pair * (unsigned)S) : 0;
  const int Q_ = a.Q, D = a.D, E_ = a.E;
  const int NG_ = (E_ + 15) >> 4;                             // 16-column granules of E; wavefront t owns t, t + 4, ...
  const int Q = Q_, E = E_, NG = NG_;
  const int RB = E * 4;

  char* DS = smem + t * Geo::SLICE;                          // this wavefront's slice of the current block
  float* P = (float*)(smem + 4 * Geo::SLICE);                // partial cosine tiles
  float* NP = (float*)((char*)P + kPBytes);                  // [32 rows][4 wavefronts] partial squared norms
  char* GTh = (char*)NP + 512;                               // GT[row][token]  hi
  char* GTl = GTh + kGBytes;
  char* GSh = GTl + kGBytes;                                 // GS[token][row group kg][rows 4 kg + 0..3, 16 + 4 kg + 0..3]  hi
  char* GSl = GSh + kGBytes;
  float* S2 = (float*)(GSl + kGBytes);                       // [32] (sum_i G c) rd / |d| of the block's rows
  float* RQ = S2 + 32;                                       // [32] 1 / (|q_i| + tiny)   (0 past Q)
  float* NQ = RQ + 32;                                       // [32] |q_i|
  float* F2 = NQ + 32;                                       // [32] (sum_j G c) rq / |q|  (end of the pair)
  float* AK = F2 + 32;                                       // [32][12] A_ik of every token (MODE 1)
  const uint32_t ds_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)DS;

  const float* qb = a.q + pair * Q * (int64_t)E;
  const char* db = (const char*)(a.d + pair * D * (int64_t)E);
  int dlen = a.dm.len ? (int)sload_u32(a.dm.len, pair) : D;
  dlen = dlen < 0 ? 0 : (dlen > D ? D : dlen);
  const int nb = (dlen + 31) >> 5;                            // blocks that hold a real row
  const int dwords = (D + 31) >> 5;

  // ---- LDS-DMA of this wavefront's slice (see BwdGeo) ---------------------------------------------------------------------
  const int rows_last = D - 32 * (dwords - 1);
  // per block: the two per-lane source offsets (even / odd instruction) of this wavefront's slice, then per granule v the two
  // instructions that fill its chunks 4v..4v+3
  auto block_offsets = [&](int b, int E, uint32_t& o0, uint32_t& o1, int& r0, int& r1) {
    const int RB = E * 4;
    const int rmax = (b == dwords - 1) ? rows_last - 1 : 31;  // rows past D: clamped re-reads of the last row (their G is 0)
    r0 = r_ ^ h_;                                             // instruction n, chunk 2n + h: row slot ^ (chunk & 3) = r ^ h ^ 2 (n & 1)
    r1 = r0 ^ 2;
    r0 = r0 < rmax ? r0 : rmax;
    r1 = r1 < rmax ? r1 : rmax;
    o0 = (uint32_t)(r0 * RB + (16 * t + 4 * h_) * 4);
    o1 = (uint32_t)(r1 * RB + (16 * t + 8 + 4 * h_) * 4);
  };
  auto issue_granule = [&](int b, int v, int NG, int E, uint32_t o0, uint32_t o1, int r0, int r1) {
    const int RB = E * 4;
    const char* g = db + (int64_t)b * 32 * RB;
    const int gi = t + 4 * v;                                 // (wave-uniform)
    if (gi < NG) {
#pragma unroll
      for (int odd = 0; odd < 2; ++odd) {
        uint32_t vo = (odd ? o1 : o0) + (uint32_t)(256 * v);
        if (16 * gi + 16 > E) {                               // the partial last granule: columns past E re-read its last chunk
          const int col = 16 * gi + 8 * odd + 4 * h_;         // (finite filler: their query operands are zeros)
          vo = (uint32_t)((odd ? r1 : r0) * RB + (col < E - 4 ? col : E - 4) * 4);
        }
        glds16(g, vo, ds_lds + (2 * v + odd) * 1024);
      }
    }
  };
  auto issue_block = [&](int b, int NG, int E) {
    uint32_t o0, o1;
    int r0, r1;
    block_offsets(b, E, o0, o1, r0, r1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // every LDS read of the slice has returned
#pragma unroll
    for (int v = 0; v < NGW; ++v) issue_granule(b, v, NG, E, o0, o1, r0, r1);
  };
#if MM_KP_BWD_PHASE_TIMES
  float ph[11] = {0};
  long long t_last = __builtin_readcyclecounter();
#endif
  if (sp < nb) issue_block(sp, NG, E);


  // ---- per-lane validity of the 4-float windows it squares for the row norms (columns < E) -------------------------------------
  uint32_t nvalid = 0;
#pragma unroll
  for (int v = 0; v < NGW; ++v)
#pragma unroll
    for (int hq = 0; hq < 2; ++hq)
      if (16 * (t + 4 * v) + 8 * h_ + 4 * hq < E) nvalid |= 1u << (2 * v + hq);

  // ---- query: the two sets of B fragments; norms from the very values of the first set (halves, then wavefronts through LDS).
  // Every load is unconditional (clamped token / column, zero selected afterwards): one batch in flight, one wait.
  bf16x8 qah[NGW], qal[NGW];     // cosine product: B[k = column][n = token]: this lane = token r, columns 16 g + 8 h + 0..7
  bf16x8 qbh[NGW], qbl[NGW];     // grad_d product (16x16x32): B[k = token][n = column]: this lane = column lane & 15 of the granule, tokens 8 (lane >> 4) + 0..7
  Rbf rbf;
  load_rbf<K, false>(a.mu, a.sigma, a.alpha, a.w, rbf);
  const float g = MODE ? a.go[pair] : 0.0f;
  float pkv[K];                                               // MODE 1: the forward's pooled sums of this lane's token
#pragma unroll
  for (int k = 0; k < K; ++k) pkv[k] = 1.0f;
  {
    const int e16 = lane & 15, kg = lane >> 4;
    const int rc = r_ < Q ? r_ : Q - 1;
    if (MODE) {                                               // (same batch of loads as the query rows: one memory round trip)
      const float* pp = pkq_in + (pair * Q + rc) * K;
#pragma unroll
      for (int k = 0; k < K; ++k) pkv[k] = pp[k];
    }
    f32x4 xa[NGW][2];
    float yb[NGW][8];
#pragma unroll
    for (int v = 0; v < NGW; ++v) {
      const int col = 16 * (t + 4 * v) + 8 * h_;
      const float* p = qb + (uint32_t)(rc * E);
      xa[v][0] = *(const f32x4*)(p + (col < E - 4 ? col : E - 4));
      xa[v][1] = *(const f32x4*)(p + (col + 4 < E - 4 ? col + 4 : E - 4));
      const int cb = 16 * (t + 4 * v) + e16;
      const float* pc = qb + (cb < E ? cb : E - 1);
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int i = 8 * kg + rr;
        yb[v][rr] = pc[(uint32_t)((i < Q ? i : Q - 1) * E)];
      }
    }
    float ss = 0.0f;
#pragma unroll
    for (int v = 0; v < NGW; ++v) {
      const int col = 16 * (t + 4 * v) + 8 * h_;
      if (!(r_ < Q && col < E)) xa[v][0] = f32x4{0, 0, 0, 0};
      if (!(r_ < Q && col + 4 < E)) xa[v][1] = f32x4{0, 0, 0, 0};
      ss += sumsq4(xa[v][0]) + sumsq4(xa[v][1]);
      split8(xa[v][0], xa[v][1], qah[v], qal[v]);
      const bool cv = 16 * (t + 4 * v) + e16 < E;
      f32x4 y0, y1;
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const float x = (cv && 8 * kg + rr < Q) ? yb[v][rr] : 0.0f;
        if (rr < 4) y0[rr] = x; else y1[rr - 4] = x;
      }
      split8(y0, y1, qbh[v], qbl[v]);
    }
    ss += __shfl_xor(ss, 32, 64);
    if (h_ == 0) NP[r_ * 4 + t] = ss;
  }
  __syncthreads();
  float rq, nqv;
  {
    const f32x4 np = *(const f32x4*)(NP + r_ * 4);
    nqv = sqrtf(((np[0] + np[1]) + np[2]) + np[3]);
    rq = r_ < Q ? 1.0f / (nqv + 1e-13f) : 0.0f;
    if (t == 0 && h_ == 0) {
      NQ[r_] = nqv;
      RQ[r_] = rq;
    }
  }
  bool qvalid = r_ < Q;
  {
    const int qlen = a.qm.len ? (int)sload_u32(a.qm.len, pair) : Q;
    qvalid = qvalid && r_ < qlen;
    if (a.qm.bits) qvalid = qvalid && ((sload_u32(a.qm.bits, pair) >> r_) & 1u);
  }
  float pk0[K];                                               // MODE 0: pooled sums of this lane's token over its rows
#pragma unroll
  for (int k = 0; k < K; ++k) pk0[k] = 0.0f;
  if (MODE) {
    // A_ik of token r -> LDS (read back per block: eleven registers less across the block loop), parameter gradients of the pair
    float lw[K], la[K], av[12];
    av[11] = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float pk = pkv[k];
      const bool live = qvalid && rbf.alpha[k] * pk >= a.clamp_min;
      av[k] = live ? g * rbf.w[k] / pk : 0.0f;
      lw[k] = qvalid ? __logf(fmaxf(rbf.alpha[k] * pk, a.clamp_min)) : 0.0f;
      la[k] = live ? rbf.w[k] / rbf.alpha[k] : 0.0f;
    }
    if (t == 0 && h_ == 0) {
#pragma unroll
      for (int v = 0; v < 3; ++v) *(f32x4*)(AK + r_ * 12 + 4 * v) = f32x4{av[4 * v], av[4 * v + 1], av[4 * v + 2], av[4 * v + 3]};
    }
    if (t == 0 && sp == 0) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float sw = half_sum_last(lw[k]), sa = half_sum_last(la[k]);
        if (lane == 31) {
          a.gw[pair * K + k] = g * sw;
          a.galpha[pair * K + k] = g * sa;
        }
      }
    }
    lds_barrier();                                            // A_ik (and RQ / NQ) are in LDS
  }

  KPS_PH(0);                                                  // prologue
  f32x4 accq[NGW][2];                                         // grad_q tiles (transposed): token 16 mt + (lane & 15), columns 16 g + 4 (lane >> 4) + 0..3
#pragma unroll
  for (int v = 0; v < NGW; ++v) accq[v][0] = accq[v][1] = f32x4{0, 0, 0, 0};
  float sqacc = 0.0f;                                         // sum_j G_ij c_ij over this lane's rows of every block
  float* gd = a.gd + pair * D * (int64_t)E;
  const int Pq = 3 * 8 * 32;                                  // floats per quarter of P

  for (int b = sp; b < nb; b += S) {
    const int j0 = 32 * b;
    // lane coordinates behind an opaque copy: the per-lane LDS / global addresses derived from them are then computed inside the
    // block, not once before the loop and kept in (spilled) registers across it (the lesson of kernel_pool_bwd.hip / tkl_bwd.hip)
    int r = r_, h = h_;
    asm volatile("" : "+v"(r), "+v"(h));
    // ... and the uniform shape values: the dozens of loop-invariant wave-uniform conditions derived from them (granule valid, last
    // granule, Q > 16, ...) are then scalar compares where they are used, not SGPR pairs held — and spilled to VGPR lanes — across the loop
    int NG = NG_, E = E_, Q = Q_;
    asm volatile("" : "+s"(NG), "+s"(E), "+s"(Q));
    // row validity word of the block; rows of quarter t: bits 8t + 4h + x
    const int rem = dlen - j0;
    uint32_t va = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
    if (a.dm.bits) va &= sload_u32(a.dm.bits, pair * dwords + b);
    float gate[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    if (GATE) {
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int j = j0 + 8 * t + 4 * h + x;
        gate[x] = j < D ? fmaxf(a.dw[pair * D + j], 0.0f) : 0.0f;
      }
    }
    KPS_PH(1);                                                // block head (masks, gate)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the slice has landed
    KPS_PH(2);                                                // wait for the slice (and the previous block's stores)

    // ---- cosines: this wavefront's K range; norms from the very A operands -----------------------------------------------
    f32x16 acc = {0}, accx = {0};                             // hi.hi, and the two cross terms: two dependency chains instead of one
    float ss = 0.0f;
#pragma unroll
    for (int v = 0; v < NGW; ++v) {
      if (t + 4 * v < NG) {
        // chunks 4v + 2h, 4v + 2h + 1 of row r: slots r ^ 2h, r ^ (2h + 1)
        const f32x4 x0 = *(const f32x4*)(DS + ((4 * v + 2 * h) * 32 + (r ^ (2 * h))) * 16);
        const f32x4 x1 = *(const f32x4*)(DS + ((4 * v + 2 * h + 1) * 32 + (r ^ (2 * h + 1))) * 16);
        bf16x8 ah, al;
        split8(x0, x1, ah, al);
        acc = mfma_bf16(ah, qah[v], acc);
        accx = mfma_bf16(al, qah[v], accx);
        accx = mfma_bf16(ah, qal[v], accx);
        const float s0 = sumsq4(x0), s1 = sumsq4(x1);
        ss += ((nvalid >> (2 * v)) & 1u) ? s0 : 0.0f;
        ss += ((nvalid >> (2 * v + 1)) & 1u) ? s1 : 0.0f;
      }
    }
#pragma unroll
    for (int x = 0; x < 16; ++x) acc[x] += accx[x];
    ss += __shfl_xor(ss, 32, 64);
    if (h == 0) NP[r * 4 + t] = ss;
    // publish the three foreign quarters of the partial tile; keep the own one
    float own[4];
#pragma unroll
    for (int tq = 0; tq < 4; ++tq) {
      if (tq == t) {
#pragma unroll
        for (int x = 0; x < 4; ++x) own[x] = acc[4 * tq + x];
      } else {
        float* dst = P + tq * Pq + (t < tq ? t : t - 1) * 256 + 4 * h * 32 + r;
#pragma unroll
        for (int x = 0; x < 4; ++x) dst[x * 32] = acc[4 * tq + x];
      }
    }
    if (MODE == 0 && b + S < nb) issue_block(b + S, NG, E);          // pooling pass: the slice is consumed
    KPS_PH(3);                                                // cosines + publish
    lds_barrier();                                            // barrier 1: partial tiles and norms are in LDS
    KPS_PH(4);                                                // barrier 1

    // ---- rows 8t + 4h + 0..3 of the block x token r: cosines, kernels, G -------------------------------------------------
    float c4[4], rd4[4], nd4[4];
    {
      const float* src = P + t * Pq + 4 * h * 32 + r;
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const float dot = ((own[x] + src[x * 32]) + src[256 + x * 32]) + src[512 + x * 32];
        const f32x4 np = *(const f32x4*)(NP + (8 * t + 4 * h + x) * 4);
        const float n = sqrtf(((np[0] + np[1]) + np[2]) + np[3]);
        nd4[x] = n;
        rd4[x] = 1.0f / (n + 1e-13f);
        c4[x] = (dot * rq) * rd4[x];
      }
    }
    const uint32_t vb = va >> (8 * t + 4 * h);
    if (MODE == 0) {
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const float m = ((vb >> x) & 1u) ? gate[x] : 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const float tt = c4[x] - rbf.mu[k];
          pk0[k] += m * __builtin_amdgcn_exp2f(tt * tt * rbf.c2[k]);
        }
      }
    } else {
      float G4[4], sg4[4], Ak[12];
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        const f32x4 w4 = *(const f32x4*)(AK + r * 12 + 4 * v);
        Ak[4 * v] = w4[0]; Ak[4 * v + 1] = w4[1]; Ak[4 * v + 2] = w4[2]; Ak[4 * v + 3] = w4[3];
      }
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        float gs = 0.0f, sg = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const float tt = c4[x] - rbf.mu[k];
          const float ae = Ak[k] * __builtin_amdgcn_exp2f(tt * tt * rbf.c2[k]);
          if (GATE) sg += ae;
          gs += (ae * tt) * rbf.c2[k];                         // -(c - mu) / sigma^2 = (c - mu) c2 2 ln 2: the constant once per row, below
                                                               // (c2 stays an SGPR operand: a product with a constant would live in a VGPR)
        }
        const bool real = (vb >> x) & 1u;
        G4[x] = real ? (gs * 1.3862943611198906f) * gate[x] : 0.0f;
        sg4[x] = real ? sg : 0.0f;
      }
      // own-direction terms of the two norms: sum_i G c per row (over the 32 token lanes), sum_j G c per token (kept per lane)
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const float gc = G4[x] * c4[x];
        sqacc += gc;
        const float td = half_sum_last(gc);
        if (r == 31) S2[8 * t + 4 * h + x] = nd4[x] > 0.0f ? td * rd4[x] / nd4[x] : 0.0f;
        if (GATE) {
          const float sgs = half_sum_last(sg4[x]);
          const int j = j0 + 8 * t + 4 * h + x;
          if (r == 31 && a.gdw && j < D) a.gdw[pair * D + j] = sgs;
        }
      }
      // GS[token r][kg][rr]: the K order of the 16x16x32 products is rr < 4: row 4 kg + rr, rr >= 4: row 16 + 4 kg + rr - 4
      // (the accumulator rows of a lane); rows 8t + 4h + 0..3 are kg = 2 (t & 1) + h, rr = 4 (t >> 1) + 0..3
      {
        uint32_t hw[2], lw2[2];
#pragma unroll
        for (int y = 0; y < 2; ++y) {
          const float v0 = G4[2 * y] * rd4[2 * y], v1 = G4[2 * y + 1] * rd4[2 * y + 1];
          const uint32_t w = cvt_pk_bf16(v0, v1);
          hw[y] = w;
          lw2[y] = cvt_pk_bf16(v0 - __uint_as_float(w << 16), v1 - __uint_as_float(w & 0xffff0000u));
        }
        const int off = r * kGRow + (2 * (t & 1) + h) * 16 + (t >> 1) * 8;
        *(uint2*)(GSh + off) = uint2{hw[0], hw[1]};
        *(uint2*)(GSl + off) = uint2{lw2[0], lw2[1]};
      }
      // GT[row][token]: tokens r and r ^ 1 share a 32-bit word; the even lane stores rows 0, 1 of the four, the odd lane rows 2, 3
      {
        uint32_t w4[4];                                       // per row: hi (low half) | lo (high half) of G rq rd
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const float v = (G4[x] * rq) * rd4[x];
          const uint32_t wh = cvt_pk_bf16(v, 0.0f) & 0xffffu;
          const float hv = __uint_as_float(wh << 16);
          w4[x] = wh | (cvt_pk_bf16(v - hv, 0.0f) << 16);
        }
        const bool odd = r & 1;
        const uint32_t s0 = odd ? w4[0] : w4[2], s1 = odd ? w4[1] : w4[3];     // what the partner stores
        const uint32_t p0 = __shfl_xor(s0, 1, 64), p1 = __shfl_xor(s1, 1, 64);
        const uint32_t m0 = odd ? w4[2] : w4[0], m1 = odd ? w4[3] : w4[1];     // my values of the rows I store
        const uint32_t e0 = odd ? p0 : m0, o0 = odd ? m0 : p0, e1 = odd ? p1 : m1, o1 = odd ? m1 : p1;   // (even token, odd token)
        const int row = 8 * t + 4 * h + (odd ? 2 : 0);
        const int off = row * kGRow + (r & ~1) * 2;
        *(uint32_t*)(GTh + off) = (e0 & 0xffffu) | (o0 << 16);
        *(uint32_t*)(GTl + off) = (e0 >> 16) | (o0 & 0xffff0000u);
        *(uint32_t*)(GTh + off + kGRow) = (e1 & 0xffffu) | (o1 << 16);
        *(uint32_t*)(GTl + off + kGRow) = (e1 >> 16) | (o1 & 0xffff0000u);
      }
      KPS_PH(5);                                              // G
      lds_barrier();                                          // barrier 2: G operands and S2 are in LDS
      KPS_PH(6);                                              // barrier 2

      // ---- the two gradient products over this wavefront's granules (16x16x32: lane = (column e16, row / token group kg)) ----
      // The G operands of the block serve every granule (40 registers); the column values of granule v + 1 are fetched while
      // granule v is multiplied, and the slice goes back to the LDS-DMA right after its last read.
      const int nj = D - j0 < 32 ? D - j0 : 32;
      int e16 = lane & 15, kg = lane >> 4;
      asm volatile("" : "+v"(e16), "+v"(kg));
      const int eq = e16 >> 2;
      const int vlast = t < NG ? (NG - 1 - t) >> 2 : -1;      // this wavefront's last granule (wave-uniform)
      const int goff = e16 * kGRow + kg * 16;
      // Both products are computed TRANSPOSED (operands swapped: A = the column-side fragment, B = the G fragment), so that a
      // lane's four accumulators are four CONSECUTIVE COLUMNS 16 g + 4 kg + 0..3 of ONE row / token (lane & 15 of tile mt): 16-byte
      // stores (a quarter of the vector-memory instructions of a column-per-lane layout, which ran into the address unit's
      // issue rate) and the row's own-direction term reads its four document values as ONE ds_read_b128.
      // Two passes over the granules — grad_d with the GT operands, then grad_q with the GS operands — so that only one set of G
      // fragments (16 registers) is live beside the 120 registers of query fragments and grad_q accumulators.
      {
        const bf16x8 gth0 = *(const bf16x8*)(GTh + goff), gtl0 = *(const bf16x8*)(GTl + goff);
        const bf16x8 gth1 = *(const bf16x8*)(GTh + goff + 16 * kGRow), gtl1 = *(const bf16x8*)(GTl + goff + 16 * kGRow);
        const float s20 = S2[e16], s21 = S2[16 + e16];
        // chunk 4v + kg (columns 16 g + 4 kg..) of rows e16 and 16 + e16: slots row ^ kg
        const char* cbase = DS + (kg * 32 + (e16 ^ kg)) * 16;
        f32x4 dc0 = {0, 0, 0, 0}, dc1 = {0, 0, 0, 0};
        if (vlast >= 0) {
          dc0 = *(const f32x4*)cbase;
          dc1 = *(const f32x4*)(cbase + 256);
        }
        const uint32_t lo = (uint32_t)(e16 * E + 4 * kg);     // + 16 g; rows j0 + e16 and j0 + 16 + e16
        float* g0 = gd + (uint32_t)(j0 * E) + 16 * t, * g1 = gd + (uint32_t)((j0 + 16) * E) + 16 * t;
#pragma unroll
        for (int v = 0; v < NGW; ++v) {
          if (v <= vlast) {
            f32x4 dn0 = {0, 0, 0, 0}, dn1 = {0, 0, 0, 0};
            if (v + 1 < NGW && v + 1 <= vlast) {
              dn0 = *(const f32x4*)(cbase + (v + 1) * 2048);
              dn1 = *(const f32x4*)(cbase + (v + 1) * 2048 + 256);
            }
            f32x4 ad0 = {0, 0, 0, 0}, ad1 = {0, 0, 0, 0};     // grad_d[row 16 mt + e16][columns 16 g + 4 kg + 0..3]
            ad0 = mfma16(qbh[v], gth0, ad0);
            ad1 = mfma16(qbh[v], gth1, ad1);
            ad0 = mfma16(qbh[v], gtl0, ad0);
            ad1 = mfma16(qbh[v], gtl1, ad1);
            ad0 = mfma16(qbl[v], gth0, ad0);
            ad1 = mfma16(qbl[v], gth1, ad1);
            const f32x4 o0 = ad0 - dc0 * s20, o1 = ad1 - dc1 * s21;
            const bool cin = 16 * (t + 4 * v) + 4 * kg < E;   // (E = 4n: a lane's four columns are valid together)
            if (nj == 32 && 16 * (t + 4 * v) + 16 <= E) {     // wave-uniform: a full block and a full granule store without lane conditions
              *(f32x4*)(g0 + 64 * v + lo) = o0;
              *(f32x4*)(g1 + 64 * v + lo) = o1;
            } else {
              if (cin && e16 < nj) *(f32x4*)(g0 + 64 * v + lo) = o0;
              if (cin && 16 + e16 < nj) *(f32x4*)(g1 + 64 * v + lo) = o1;
            }
            dc0 = dn0;
            dc1 = dn1;
          }
        }
      }
      KPS_PH(9);                                              // grad_d pass
      {
        // grad_q^T tile: A[m = column e16][k = rows in accumulator order] = the block's column values, B[k][n = token] = GS
        const bf16x8 gsh0 = *(const bf16x8*)(GSh + goff), gsl0 = *(const bf16x8*)(GSl + goff);
        const bf16x8 gsh1 = *(const bf16x8*)(GSh + goff + 16 * kGRow), gsl1 = *(const bf16x8*)(GSl + goff + 16 * kGRow);
        // rows 4 kg + x and 16 + 4 kg + x of column e16 of granule v: chunk 4v + eq, slots (4 kg + x) ^ eq
        const char* dbase = DS + (eq * 32 + 4 * kg) * 16 + (e16 & 3) * 4;
        const int sx0 = (0 ^ eq) << 4, sx1 = (1 ^ eq) << 4, sx2 = (2 ^ eq) << 4, sx3 = (3 ^ eq) << 4;
        auto load_dv = [&](int v, float (&dv)[8]) {
          const char* p = dbase + v * 2048;
          dv[0] = *(const float*)(p + sx0); dv[4] = *(const float*)(p + sx0 + 256);
          dv[1] = *(const float*)(p + sx1); dv[5] = *(const float*)(p + sx1 + 256);
          dv[2] = *(const float*)(p + sx2); dv[6] = *(const float*)(p + sx2 + 256);
          dv[3] = *(const float*)(p + sx3); dv[7] = *(const float*)(p + sx3 + 256);
        };
        float dvc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (vlast >= 0) load_dv(0, dvc);
        // The slice goes back to the LDS-DMA granule by granule: once granule v's column values are in registers (the split
        // below consumed them, and the grad_d pass is done with the whole slice) its four chunks are refilled with the next
        // block's — two LDS-DMA instructions per granule between the MFMAs instead of a burst of ten behind one wait (the
        // burst: 2.0 k cycles per block and wavefront, and the next block's rows requested a whole pass later).
        uint32_t o0 = 0, o1 = 0;
        int r0 = 0, r1 = 0;
        const bool refill = b + S < nb;
        if (refill) block_offsets(b + S, E, o0, o1, r0, r1);
#pragma unroll
        for (int v = 0; v < NGW; ++v) {
          if (v <= vlast) {
            float dvn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (v + 1 < NGW && v + 1 <= vlast) load_dv(v + 1, dvn);
            bf16x8 bh, bl;
            split8(f32x4{dvc[0], dvc[1], dvc[2], dvc[3]}, f32x4{dvc[4], dvc[5], dvc[6], dvc[7]}, bh, bl);
            if (refill) {
              asm volatile("" : "+v"(bh), "+v"(bl));          // (the refill below is ordered behind the values' arrival)
              KPS_PH(7);
              issue_granule(b + S, v, NG, E, o0, o1, r0, r1);
              KPS_PH(10);                                     // LDS-DMA issue
            }
            accq[v][0] = mfma16(bh, gsh0, accq[v][0]);
            accq[v][0] = mfma16(bh, gsl0, accq[v][0]);
            accq[v][0] = mfma16(bl, gsh0, accq[v][0]);
            if (Q > 16) {
              accq[v][1] = mfma16(bh, gsh1, accq[v][1]);
              accq[v][1] = mfma16(bh, gsl1, accq[v][1]);
              accq[v][1] = mfma16(bl, gsh1, accq[v][1]);
            }
#pragma unroll
            for (int x = 0; x < 8; ++x) dvc[x] = dvn[x];
          }
        }
      }
      KPS_PH(7);                                              // gradient products
    }
  }

  const int r = r_, h = h_;
  KPS_PH(7);                                                  // (last block's) gradient products
  if (MODE == 0) {
    // pooled sums of token r: both halves, then the four wavefronts (their row quarters) through LDS
    float* R = P;                                             // [4][32][K]
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float v = pk0[k] + __shfl_xor(pk0[k], 32, 64);
      if (h == 0) R[(t * 32 + r) * K + k] = v;
    }
    __syncthreads();
    for (int idx = tid; idx < Q * K; idx += 256) {
      const int i = idx / K, k = idx - i * K;
      pkq_out[(pair * Q + i) * K + k] = ((R[i * K + k] + R[(32 + i) * K + k]) + R[(64 + i) * K + k]) + R[(96 + i) * K + k];
    }
    return;
  }

  // ---- rows past the last real block: zeros (every byte of grad_d is written by this launch) -------------------------------
  if (sp == 0) {
    const int64_t z0 = (int64_t)nb * 32 * E, z1 = (int64_t)D * E;
    for (int64_t idx = z0 + 4 * tid; idx < z1; idx += 1024) *(f32x4*)(gd + idx) = f32x4{0, 0, 0, 0};
    if (GATE && a.gdw)
      for (int j = nb * 32 + tid; j < D; j += 256) a.gdw[pair * D + j] = 0.0f;
  }
  // ---- grad_q: own-direction term needs sum_j G c over every row: halves, then wavefronts ------------------------------------
  {
    float* R = P;                                             // [4][32]
    lds_barrier();                                            // (LDS only: the last block's gradient stores stay in flight)
    const float v = sqacc + __shfl_xor(sqacc, 32, 64);
    if (h == 0) R[t * 32 + r] = v;
    const int e16 = lane & 15, kg = lane >> 4;
    if (S > 1) {
      // this workgroup's share of the pair: raw grad_q accumulators + its own-direction sums (+ the query norms once) -> `part`
      float* pw = a.part + (pair * S + sp) * ((int64_t)Q * E + 96);
#pragma unroll
      for (int v2 = 0; v2 < NGW; ++v2) {
        const int col = 16 * (t + 4 * v2) + 4 * kg;
        if (t + 4 * v2 < NG && col < E) {
          if (e16 < Q) *(f32x4*)(pw + (uint32_t)(e16 * E + col)) = accq[v2][0];
          if (16 + e16 < Q) *(f32x4*)(pw + (uint32_t)((16 + e16) * E + col)) = accq[v2][1];
        }
      }
      lds_barrier();
      if (tid < 32) {
        float* tail = pw + (int64_t)Q * E;
        tail[tid] = ((R[tid] + R[32 + tid]) + R[64 + tid]) + R[96 + tid];
        tail[32 + tid] = RQ[tid];
        tail[64 + tid] = NQ[tid];
      }
      return;
    }
    // this lane's 16-byte pieces of q (unconditional, clamped): in flight across the two barriers
    f32x4 qv[NGW][2];
#pragma unroll
    for (int v2 = 0; v2 < NGW; ++v2) {
      const int col = 16 * (t + 4 * v2) + 4 * kg;
      const float* pc = qb + (col < E - 4 ? col : E - 4);
      qv[v2][0] = *(const f32x4*)(pc + (uint32_t)((e16 < Q ? e16 : Q - 1) * E));
      qv[v2][1] = *(const f32x4*)(pc + (uint32_t)((16 + e16 < Q ? 16 + e16 : Q - 1) * E));
    }
    lds_barrier();
    if (tid < 32) {
      const float s2 = ((R[tid] + R[32 + tid]) + R[64 + tid]) + R[96 + tid];
      const float n = NQ[tid];
      F2[tid] = n > 0.0f ? s2 * RQ[tid] / n : 0.0f;
    }
    lds_barrier();
    float* gq = a.gq + pair * Q * (int64_t)E;
    const float rq0 = RQ[e16], rq1 = RQ[16 + e16], f20 = F2[e16], f21 = F2[16 + e16];
#pragma unroll
    for (int v2 = 0; v2 < NGW; ++v2) {
      const int col = 16 * (t + 4 * v2) + 4 * kg;
      if (t + 4 * v2 < NG && col < E) {
        if (e16 < Q) *(f32x4*)(gq + (uint32_t)(e16 * E + col)) = accq[v2][0] * rq0 - qv[v2][0] * f20;
        if (16 + e16 < Q) *(f32x4*)(gq + (uint32_t)((16 + e16) * E + col)) = accq[v2][1] * rq1 - qv[v2][1] * f21;
      }
    }
  }
#if MM_KP_BWD_PHASE_TIMES
  KPS_PH(8);                                                  // epilogue
  __syncthreads();
  if (pair == (int64_t)(gridDim.x / 2) && lane == 0 && (t == 0 || t == 3))
    for (int k = 0; k < 11; ++k) (t == 0 ? a.gw : a.galpha)[pair * K + k] = ph[k];
#endif
}

// grad_q of a pair whose blocks were shared by S <= 8 workgroups: the partial accumulators and own-direction sums in workgroup
// order (deterministic), then the norm's chain rule as in the one-workgroup epilogue.  One 16-byte piece of grad_q per lane,
// every load of it issued before the first use (the launch is a latency chain otherwise: 10 us for 64 pairs).
__global__ void __launch_bounds__(256) kp_bwd_combine_kernel(const KpBwdArgs a) {
  const int Q = a.Q, E = a.E, S = a.nsplit, E4 = E >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.n_pairs * Q * E4) return;
  const int64_t row = idx / E4;                                // pair * Q + i
  const int c = 4 * (int)(idx - row * E4);
  const int64_t pair = row / Q;
  const int i = (int)(row - pair * Q);
  const int64_t stride = (int64_t)Q * E + 96;
  const float* pw = a.part + pair * S * stride;
  f32x4 acc[8];
  float sq[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const float* ps = pw + (s < S ? s : 0) * stride;
    acc[s] = *(const f32x4*)(ps + i * E + c);
    sq[s] = ps[(int64_t)Q * E + i];
  }
  const float rq = pw[(int64_t)Q * E + 32 + i], nq = pw[(int64_t)Q * E + 64 + i];
  const f32x4 qv = *(const f32x4*)(a.q + row * E + c);
  f32x4 sum = acc[0];
  float ssq = sq[0];
#pragma unroll
  for (int s = 1; s < 8; ++s)
    if (s < S) {
      sum += acc[s];
      ssq += sq[s];
    }
  const float f2 = nq > 0.0f ? ssq * rq / nq : 0.0f;
  *(f32x4*)(a.gq + row * E + c) = sum * rq - qv * f2;
}

bool kp_bwd_split_supported(int Q, int E, int K) { return Q >= 1 && Q <= 32 && K == 11 && E >= 4 && !(E & 3) && E <= 320; }

size_t kp_bwd_split_ws_bytes(int64_t n_pairs, int Q, int K) { return (size_t)n_pairs * Q * K * sizeof(float); }

// Small batches — batch_size_train 32 x 2 = 64 pairs is what the reference trains with (defaults.yaml:114) — leave most of the
// 512 workgroup slots empty and a pair's seven blocks in ONE workgroup are a serial chain (66 us at 64 pairs): S workgroups
// share a pair's blocks (MM_KP_BWD_NSPLIT forces S; 0 = by batch size), never more than it has blocks.
// Measured (ragged MSMARCO lengths, D = 200; gradient launch + combine launch, us): 64 pairs 42.7 -> 22.3 + 4.8; 128 pairs
// 44.3 -> 29.9 + 5.2 (S = 4); 192 / 256 pairs no gain with S = 2 (45.6 / 49 vs 45.6 / 48) -> one workgroup per pair from 129 on.
int kp_bwd_split_nsplit(int64_t n_pairs, int D) {
  const int nb = (D + 31) >> 5;
  int S = env().kp_bwd_nsplit > 0 ? (env().kp_bwd_nsplit > 8 ? 8 : env().kp_bwd_nsplit) : (n_pairs <= 128 ? 4 : 1);
  return S < nb ? S : (nb > 0 ? nb : 1);
}

size_t kp_bwd_split_part_bytes(int64_t n_pairs, int Q, int D, int E) {
  const int S = kp_bwd_split_nsplit(n_pairs, D);
  return S > 1 ? (size_t)n_pairs * S * ((size_t)Q * E + 96) * sizeof(float) : 0;
}

template <int NGW>
static int launch_ngw(const KpBwdArgs& a, const float* pkq_in, float* pkq_ws, hipStream_t stream) {
  const size_t lds = BwdGeo<NGW>::LDS;
  const dim3 block(256);
  auto go = [&](auto kern, const float* pin, float* pout, unsigned wgs) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(wgs), block, lds, stream, a, pin, pout);
  };
  const unsigned np = (unsigned)a.n_pairs;
  if (!pkq_in) {
    a.dw ? go(kp_bwd_split_kernel<0, true, NGW>, nullptr, pkq_ws, np) : go(kp_bwd_split_kernel<0, false, NGW>, nullptr, pkq_ws, np);
    if (int e = check_launch("kp_bwd_split_kernel<pool>")) return e;
    pkq_in = pkq_ws;
  }
  a.dw ? go(kp_bwd_split_kernel<1, true, NGW>, pkq_in, nullptr, np * (unsigned)a.nsplit)
       : go(kp_bwd_split_kernel<1, false, NGW>, pkq_in, nullptr, np * (unsigned)a.nsplit);
  if (int e = check_launch("kp_bwd_split_kernel")) return e;
  if (a.nsplit > 1) {
    const int64_t pieces = a.n_pairs * a.Q * (a.E >> 2);
    hipLaunchKernelGGL(kp_bwd_combine_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, stream, a);
    return check_launch("kp_bwd_combine_kernel");
  }
  return MM_OK;
}

int kp_bwd_split_launch(const KpBwdArgs& a0, const float* pkq_in, float* pkq_ws, float* part, size_t part_bytes, hipStream_t stream) {
  if (!pkq_in && !pkq_ws) return set_error(MM_EINVAL, "kernel_pool_bwd: no pooled sums and no workspace for them");
  KpBwdArgs a = a0;
  a.nsplit = kp_bwd_split_nsplit(a.n_pairs, a.D);
  a.part = part;
  if (a.nsplit > 1 && (!part || part_bytes < kp_bwd_split_part_bytes(a.n_pairs, a.Q, a.D, a.E))) a.nsplit = 1;   // (an old-size workspace: one workgroup per pair)
  const int NG = (a.E + 15) >> 4;
  if (NG <= 8) return launch_ngw<2>(a, pkq_in, pkq_ws, stream);
  return launch_ngw<5>(a, pkq_in, pkq_ws, stream);
}

}  // namespace mm
