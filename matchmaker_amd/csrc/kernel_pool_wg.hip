// TK kernel pooling, shared-query lists at E = 300: TWO wavefronts per SIMD.
//
// kernel_pool_split_kernel (kernel_pool.hip) keeps the query tile as 152 registers of bf16 hi / lo MFMA B fragments and a
// 39 KiB LDS ring per wavefront: one wavefront per SIMD.  Measured by removal its stream alone runs at
// 6.46 TB/s, and the 0.7 ms of VALU / MFMA / LDS-DMA issue work of a 64,000-pair launch is simply ADDED, because a
// wavefront blocked on the LDS-DMA queue cannot issue anything else and nobody else is resident on its SIMD.
//
// In a "1 query x C candidates" list every pair of a workgroup's range scores the SAME query, so here the query tile
// lives ONCE per workgroup in LDS (19 K-steps x {hi, lo} x 1 KiB, lane-linear: 38 KiB) and every K-step reads its two B
// fragments with ds_read_b128 (the address pattern maxsim_pair.hip uses for its ring-resident query tile).  That
// frees the 152 registers: eight wavefronts of <= 256 registers per workgroup, two per SIMD, each with its own pairs and
// its own two-slot ring of SMALLER slices — 32 tokens x 15 chunks (240-B row pieces: odd chunk count, so the 16 rows of a
// ds_read_b128 group sit on 16 distinct bank slots, no swizzle) = 7.5 KiB = 7.5 LDS-DMA instructions (the eighth runs
// with the upper half of EXEC off).  LDS: 38,912 (query) + 8 x 2 x 7,680 (rings) + 8 x 128 (row norms) = 162,816 B.
//
// K order of a 32-token block (5 slices x 15 chunks = 75 chunks = 300 floats, 19 steps of 16):
//   slice s, step p = 0..2: chunks 15 s + 4 p + 2 h, + 1 for lane half h (8 consecutive floats)      -> steps 0..14
//   chunks 12..14 of every slice are parked: parked chunk g = 3 s + m joins step 15 + g / 4 at position g % 4
//   (lane half (g % 4) / 2, first / second chunk g % 2); position 3 of step 18 does not exist (zeros).
// A slice's slot goes back to the producer when its steps are done, so during a block's epilogue both slots of the
// wavefront are in flight (8 x 15 KiB per CU) and during the K loop one is.
#include "mm_internal.h"
#include "kp_device.h"

namespace mm {

constexpr int kWgWaves = 8;
constexpr int kWC = 15;                        // 16-B chunks per row per slice
constexpr int kWRowB = kWC * 16;               // 240
constexpr int kWSlice = 32 * kWRowB;           // 7,680
constexpr int kWNbuf = 2;
constexpr int kWSl = 5;                        // slices per 32-token block
constexpr int kWSteps = 19;
constexpr int kWQt = kWSteps * 2 * 1024;       // query tile bytes
constexpr int kWE = 300;
constexpr int kWRB = kWE * 4;

// first chunk of this lane half's pair of chunks in K-step j (second = the next parked / row chunk); -1: none
__device__ __forceinline__ int wg_chunk(int j, int h, int which) {
  if (j < 15) return 15 * (j / 3) + 4 * (j % 3) + 2 * h + which;
  const int g = 4 * (j - 15) + 2 * h + which;
  return g < 15 ? 15 * (g / 3) + 12 + g % 3 : -1;
}

// eight LDS-DMA instructions = one slice; the last one moves only 32 x 16 B (EXEC upper half off)
__device__ __forceinline__ void wg_issue_slice(const char* gbase, const uint32_t (&v)[8], uint32_t lds_dst) {
  uint32_t keep;
  uint64_t ex;
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %11\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %10\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %3, %10\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %4, %10\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %5, %10\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %6, %10\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %7, %10\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %8, %10\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_mov_b64 %1, exec\n\t"
      "s_mov_b32 exec_hi, 0\n\t"
      "global_load_lds_dwordx4 %9, %10\n\t"
      "s_mov_b64 exec, %1\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep), "=&s"(ex)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "s"(gbase), "s"(lds_dst)
      : "memory", "scc");
}

__device__ __forceinline__ void wg_wait(int younger) {
  if (younger <= 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

// log-sum pooling of one pair (kp_device.h pool_partial) with alpha / w fetched through the scalar cache per pair
// instead of living in registers for the whole kernel
template <int K>
__device__ __forceinline__ void wg_finish(const KpArgs& a, int64_t pair, const float (&pk)[kMaxK], bool count_lane, int lane) {
  float total = 0.0f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float lg = __logf(fmaxf(pk[k] * sload_f32(a.alpha, k), a.clamp_min));
    lg = count_lane ? lg : 0.0f;
    const float sm = wave_sum(lg);
    if (a.per_kernel && lane == 0) a.per_kernel[pair * K + k] = sm;
    total += sload_f32(a.w, k) * sm;
  }
  if (lane == 0) a.out[pair] = total;
}

template <int K>
__global__ void __launch_bounds__(kWgWaves * 64) kernel_pool_wg_kernel(const KpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform for the compiler too (scalar loads, "s" asm operands)
  const int r = lane & 31, h = lane >> 5;
  const int D = a.D, Q = a.Q;
  const int nblk_tot = (D + 31) >> 5;
  const int rows_last = D - 32 * (nblk_tot - 1);
  char* qt = smem;                                              // [19][hi, lo][64 lanes] x 16 B
  char* ring = smem + kWQt + wv * (kWNbuf * kWSlice);
  float* rdbuf = (float*)(smem + kWQt + kWgWaves * kWNbuf * kWSlice) + wv * 32;
  // packed RBF constants (sq2 / msq2 of kp_device.h's Rbf, 6 + 6 pairs): parked in LDS and fetched at every block's
  // epilogue, so that they do not occupy 24 registers across the K loop (two wavefronts per SIMD: 256 registers each,
  // and a spill reload is a scratch load that drains the LDS-DMA queue with vmcnt(0))
  float* rbfc = (float*)(smem + kWQt + kWgWaves * kWNbuf * kWSlice + kWgWaves * 128);
  const uint32_t lds_ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring;

  // LDS-DMA source offsets of a slice image [32 rows][15 chunks]: slot = 64 n + lane (n = 7: lanes 0..31 only)
  uint32_t voff[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    int s = 64 * n + lane;
    if (s > 32 * kWC - 1) s = 32 * kWC - 1;
    const int row = s / kWC, c = s - row * kWC;
    voff[n] = (uint32_t)(row * kWRB + c * 16);
  }
  const uint32_t vmax_tail = (uint32_t)((rows_last - 1) * kWRB + (kWC - 1) * 16);
  const uint32_t a_off = (uint32_t)(r * kWRowB + h * 32);      // this lane's 32-B A window of step 0 of a slice

  if (wv == 0 && lane < (K + 1) / 2) {   // lane kp packs kernels 2 kp, 2 kp + 1 — the arithmetic of load_rbf / pack_rbf
    float sq[2], msq[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = 2 * lane + u;
      if (k < K) {
        const float sg = a.sigma[k];
        const float c2 = -1.4426950408889634f / (2.0f * sg * sg);
        sq[u] = sqrtf(-c2);
        msq[u] = a.mu[k] * sq[u];
      } else {
        sq[u] = 0.0f;
        msq[u] = 1.0e3f;
      }
    }
    *(f32x2*)(rbfc + 2 * lane) = f32x2{sq[0], sq[1]};
    *(f32x2*)(rbfc + 16 + 2 * lane) = f32x2{msq[0], msq[1]};
  }

  const char* dbase = (const char*)a.d;
  auto doc_len = [&](int64_t p) -> int {
    int len = a.dm.len ? (int)sload_u32(a.dm.len, p) : D;
    return len < 0 ? 0 : (len > D ? D : len);
  };

  const int64_t g0 = (int64_t)blockIdx.x * a.pairs_per_wave;   // (pairs per WORKGROUP in this kernel)
  const int64_t g1 = (g0 + a.pairs_per_wave < a.n_pairs) ? g0 + a.pairs_per_wave : a.n_pairs;

  for (int64_t s0 = g0; s0 < g1;) {
    // ---- segment = the pairs of this workgroup's range that score query qi ---------------------------------------
    const int64_t qi = s0 / a.ppq;
    const int64_t qend = (qi + 1) * a.ppq;
    const int64_t s1 = qend < g1 ? qend : g1;
    __syncthreads();                                            // everyone is done with the previous query tile
    {
      const int qr = r < Q ? r : Q - 1;
      const char* qrow = (const char*)a.q + (qi * Q + qr) * kWRB;
      for (int j = wv; j < kWSteps; j += kWgWaves) {
        const int cA = wg_chunk(j, h, 0), cB = wg_chunk(j, h, 1);
        f32x4 xa = {0, 0, 0, 0}, xb = {0, 0, 0, 0};
        if (cA >= 0) xa = *(const f32x4*)(qrow + cA * 16);
        if (cB >= 0) xb = *(const f32x4*)(qrow + cB * 16);
        bf16x8 hi, lo;
        split8(xa, xb, hi, lo);
        *(bf16x8*)(qt + (2 * j) * 1024 + lane * 16) = hi;
        *(bf16x8*)(qt + (2 * j + 1) * 1024 + lane * 16) = lo;
      }
    }
    // every wavefront needs 1 / (|q_r| + tiny) of its lanes' query row: 38 L2-hot loads per lane, once per segment
    float rq;
    {
      const int qr = r < Q ? r : Q - 1;
      const char* qrow = (const char*)a.q + (qi * Q + qr) * kWRB;
      float ss = 0.0f;
      for (int c = h; c < kWE / 4; c += 2) ss += sumsq4(*(const f32x4*)(qrow + c * 16));
      ss += __shfl_xor(ss, 32, 64);
      rq = 1.0f / (sqrtf(ss) + 1e-13f);
    }
    const int qlen = a.qm.len ? (int)sload_u32(a.qm.len, qi) : Q;
    bool qvalid = r < Q && r < qlen;
    const uint32_t qbits = a.qm.bits ? sload_u32(a.qm.bits, qi) : 0xffffffffu;
    if (a.qm.bits) qvalid = qvalid && ((qbits >> r) & 1u);
    const int qn = qlen < Q ? (qlen < 0 ? 0 : qlen) : Q;
    const int rrows = redist_rows(qn);
    const int np = rrows ? (32 + rrows - 1) / rrows : 2;
    const int rtk = lane / np, rsub = lane - rtk * np;
    __syncthreads();                                            // the query tile is complete (and all plain loads have landed)

    // this wavefront's pairs of the segment
    const int64_t nseg = s1 - s0;
    const int64_t p0 = s0 + nseg * wv / kWgWaves, p1 = s0 + nseg * (wv + 1) / kWgWaves;

    // producer cursor over (pair, block, slice)
    int64_t pp = p0;
    int pt = 0, ps = 0, pn = 0;
    while (pp < p1 && (pn = (doc_len(pp) + 31) >> 5) == 0) ++pp;
    int pbuf = 0, cbuf = 0, inflight = 0;
    auto top_up = [&]() {
      while (pp < p1 && inflight < kWNbuf) {
        const char* g = dbase + (pp * (int64_t)D + (int64_t)pt * 32) * kWRB + ps * kWRowB;
        uint32_t v[8];
        const bool clamp = pt == nblk_tot - 1 && rows_last != 32;
#pragma unroll
        for (int n = 0; n < 8; ++n) v[n] = clamp ? (voff[n] < vmax_tail ? voff[n] : vmax_tail) : voff[n];
        wg_issue_slice(g, v, lds_ring + (uint32_t)pbuf * kWSlice);
        pbuf ^= 1;
        ++inflight;
        if (++ps == kWSl) {
          ps = 0;
          if (++pt == pn) {
            pt = 0;
            ++pp;
            while (pp < p1 && (pn = (doc_len(pp) + 31) >> 5) == 0) ++pp;
          }
        }
      }
    };
    top_up();

    for (int64_t pair = p0; pair < p1; ++pair) {
      const int len = doc_len(pair);
      const int nb = (len + 31) >> 5;
      f32x2 pk2[kMaxK / 2];
#pragma unroll
      for (int k = 0; k < kMaxK / 2; ++k) pk2[k] = f32x2{0.0f, 0.0f};

      for (int t = 0; t < nb; ++t) {
        f32x16 acc_hh = {0}, acc_lh = {0}, acc_xl = {0};
        f32x4 park[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};  // this lane half's two chunks of the open parked step
        f32x2 ss2 = {0.0f, 0.0f};
        // operands of one K-step: 8 fp32 values of this lane's document row + the hi / lo B fragments of the query tile
        struct StepIn {
          f32x4 xa, xb;
          bf16x8 bh, bl;
        };
        auto load_b = [&](StepIn& in, int j) {
          in.bh = *(const bf16x8*)(qt + (2 * j) * 1024 + lane * 16);
          in.bl = *(const bf16x8*)(qt + (2 * j + 1) * 1024 + lane * 16);
        };
        auto compute = [&](const StepIn& in) {
          bf16x8 ah, al;
          split8(in.xa, in.xb, ah, al);
          acc_hh = mfma_bf16(ah, in.bh, acc_hh);
          acc_lh = mfma_bf16(al, in.bh, acc_lh);
          acc_xl = mfma_bf16(ah, in.bl, acc_xl);
          acc_xl = mfma_bf16(al, in.bl, acc_xl);
          const f32x2 a0 = {in.xa[0], in.xa[1]}, a1 = {in.xa[2], in.xa[3]}, b0 = {in.xb[0], in.xb[1]}, b1 = {in.xb[2], in.xb[3]};
          ss2 += a0 * a0;
          ss2 += a1 * a1;
          ss2 += b0 * b0;
          ss2 += b1 * b1;
          __builtin_amdgcn_sched_barrier(0);                     // keep the next step's loads where they are issued: one step ahead
        };
#pragma unroll 1      // (unrolled five times the compiler carries every slice's addresses and spills; a spill reload is a
                      // scratch load whose vmcnt(0) drains the LDS-DMA queue)
        for (int s = 0; s < kWSl; ++s) {
          wg_wait(inflight - 1);
          const char* buf = ring + cbuf * kWSlice;
          // software pipeline, one step deep: the operands of step i + 1 are requested before step i is computed
          StepIn cur, nxt, pst;
          cur.xa = *(const f32x4*)(buf + a_off);
          cur.xb = *(const f32x4*)(buf + a_off + 16);
          load_b(cur, 3 * s);
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            if (p < 2) {
              nxt.xa = *(const f32x4*)(buf + a_off + (p + 1) * 64);
              nxt.xb = *(const f32x4*)(buf + a_off + (p + 1) * 64 + 16);
              load_b(nxt, 3 * s + p + 1);
            }
            compute(cur);
            if (p < 2) cur = nxt;
          }
          // parked chunks 12..14 of the slice; the parked step 14 + s closes with this slice (s >= 1): its B fragments too
          f32x4 pc[3];
#pragma unroll
          for (int m = 0; m < 3; ++m) pc[m] = *(const f32x4*)(buf + r * kWRowB + (12 + m) * 16);
          if (s >= 1) load_b(pst, 14 + s);
          // in g order: parked chunk g = 3 s + m goes to lane half (g % 4) / 2, position g % 2 of parked step 15 + g / 4;
          // g = 3, 7, 11, 14 close steps 15..18 (a later chunk of the same slice already belongs to the next step)
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const int g = 3 * s + m;                             // wave-uniform
            const bool mine = h == ((g & 3) >> 1);
            const bool first = (g & 1) == 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              park[0][e] = (mine && first) ? pc[m][e] : park[0][e];
              park[1][e] = (mine && !first) ? pc[m][e] : park[1][e];
            }
            if ((g & 3) == 3 || g == 14) {
              pst.xa = park[0];
              pst.xb = park[1];
              compute(pst);
              park[0] = f32x4{0, 0, 0, 0};
              park[1] = f32x4{0, 0, 0, 0};
            }
          }
          cbuf ^= 1;
          --inflight;
          if (!(s == kWSl - 1 && np > 2)) top_up();
        }
        float ss = ss2[0] + ss2[1];
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = acc_hh[i] + (acc_lh[i] + acc_xl[i]);
        ss += __shfl_xor(ss, 32, 64);
        if (h == 0) rdbuf[r] = 1.0f / (sqrtf(ss) + 1e-13f);
        float rdr[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = *(const f32x4*)(rdbuf + 8 * g + 4 * h);
          rdr[4 * g + 0] = v[0]; rdr[4 * g + 1] = v[1]; rdr[4 * g + 2] = v[2]; rdr[4 * g + 3] = v[3];
        }
        const int rem = len - 32 * t;
        const uint32_t ex = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
        const uint32_t va = a.dm.bits ? (sload_u32(a.dm.bits, pair * nblk_tot + t) & ex) : ex;
        RbfPk rbf;
#pragma unroll
        for (int kp = 0; kp < (K + 1) / 2; kp += 2) {
          const f32x4 u = *(const f32x4*)(rbfc + 2 * kp), v = *(const f32x4*)(rbfc + 16 + 2 * kp);
          rbf.sq2[kp] = f32x2{u[0], u[1]};
          rbf.sq2[kp + 1] = f32x2{u[2], u[3]};
          rbf.msq2[kp] = f32x2{v[0], v[1]};
          rbf.msq2[kp + 1] = f32x2{v[2], v[3]};
        }
        if (np > 2) {
          float* T = (float*)(ring + (cbuf ^ 1) * kWSlice);      // the slot the block's last slice just left
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *(f32x4*)(T + r * kTS + 8 * g + 4 * h) = f32x4{(acc[4 * g] * rq) * rdr[4 * g], (acc[4 * g + 1] * rq) * rdr[4 * g + 1],
                                                         (acc[4 * g + 2] * rq) * rdr[4 * g + 2], (acc[4 * g + 3] * rq) * rdr[4 * g + 3]};
          // this lane's rows are read first, then the slot goes back to the producer, then they are evaluated
          rbf_redistributed_rows<K, false>(rrows, pk2, T, nullptr, rtk, rsub, va, rbf, top_up);
        } else {
          rbf_block<K, false, 0, (K + 1) / 2, 0, 4, RbfPk>(pk2, acc, rdr, rq, va, h, rbf);
        }
      }
      float pk[kMaxK];
      if (np > 2) {
#pragma unroll
        for (int k = 0; k < K; ++k) pk[k] = pk2[k >> 1][k & 1];
        redist_reduce<K>(pk, np, lane);
        const bool count = rsub == 0 && rtk < qn && ((qbits >> rtk) & 1u);
        wg_finish<K>(a, pair, pk, count, lane);
      } else {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          pk[k] = pk2[k >> 1][k & 1];
          pk[k] += __shfl_xor(pk[k], 32, 64);
        }
        wg_finish<K>(a, pair, pk, qvalid && lane < 32, lane);
      }
    }
    s0 = s1;
  }
}

bool kp_wg_supported(const KpArgs& a) {
  return !env().kp_no_wg && a.E == kWE && a.Q <= 32 && a.K == 11 && !a.dw && !a.pair_q && a.n_md == 0 && a.ppq >= 64 &&
         a.n_pairs >= 1024;
}

int kp_wg_launch(const KpArgs& a0, hipStream_t stream) {
  KpArgs a = a0;
  int64_t wgs = kCUs;
  if (wgs * 16 > a.n_pairs) wgs = (a.n_pairs + 15) / 16;
  a.pairs_per_wave = (a.n_pairs + wgs - 1) / wgs;               // pairs per workgroup
  wgs = (a.n_pairs + a.pairs_per_wave - 1) / a.pairs_per_wave;
  const int lds = kWQt + kWgWaves * kWNbuf * kWSlice + kWgWaves * 128 + 128;
  static_assert(kWQt + kWgWaves * kWNbuf * kWSlice + kWgWaves * 128 + 128 <= 160 * 1024, "LDS budget");
  (void)hipFuncSetAttribute((const void*)kernel_pool_wg_kernel<11>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((kernel_pool_wg_kernel<11>), dim3((unsigned)wgs), dim3(kWgWaves * 64), lds, stream, a);
  return check_launch("kernel_pool_wg_kernel");
}

}  // namespace mm
