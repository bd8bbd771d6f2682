// TKL stage 1, K-split form (round 6): the cosine match of sigir20_tkl.py:184-194 for the cosine hand-off to the window kernel
// (KpArgs::cos_out, tkl.hip), by workgroups of TWO wavefronts that share every 16-row tile of a chunk along K.
//
// Why: tkl_stage1_rows.hip showed with its phase clocks that stage 1 is not bound by its HBM stream (the stream alone: 50 us at
// 256 documents; waiting for it: 8 k of a wavefront's 140 k cycles) but by ONE wavefront per SIMD issuing ~520 instructions per
// tile at ~9 cycles each — nothing else is resident to fill its stalls, because a wavefront that holds the whole query tile
// (2 N-tiles x 10 k-steps x hi / lo = 160 registers) and a whole tile of rows cannot share its SIMD.  Here wavefront w of a
// workgroup owns k-steps [5 w, 5 w + 5) of every row: 80 registers of query fragments, 40 of rows, half of the hi / lo split and of
// the LDS-DMA issue per tile — two wavefronts per SIMD (eight per CU, the same 38.4 KB ring per workgroup as before).
//
//   * stream: each wavefront requests exactly the row PIECES it multiplies (E = 300: 640 B / 560 B of every 1,200-byte row, one
//     LDS-DMA instruction per row piece, four per M0 / base pair through the instruction offset) into its own half of the ring,
//     so requesting, waiting (vmcnt) and handing a slot back involve no other wavefront;
//   * LDS image: piece rows at a stride of 640 B would put rows r and r + 2 on the same banks; there is no room for padding
//     (4 x 38.4 KB rings + 2.2 KB of exchange = the CU's 160 KB), so row m of a tile is stored ROTATED by m 16-byte chunks
//     (source-side: lane j loads chunk (j + m) mod 40), which tiles the 64 banks exactly for the sixteen rows of a k-group;
//     the 560-B pieces (35 chunks, odd) need nothing;
//   * products: v_mfma_f32_16x16x32_bf16, A = 16 rows x 32 k, B = 16 query tokens x 32 k, the three-product bf16 split of the other
//     kernels (hi·hi + lo·hi + hi·lo, fp32 accumulation), norms in fp32 from the same registers;
//   * exchange: per tile the wavefronts swap the halves of their partial sums through LDS (wavefront w finishes rows 8 w .. 8 w + 7:
//     cosine = (dot · 1 / (|q| + 1e-13)) · 1 / (|d| + 1e-13), masked position = 1e5) — two LDS-only barriers per tile.
#include <type_traits>

#include "kp_device.h"

namespace mm {

namespace {

constexpr int kRing = 4;       // ring slots = units of 8 chunk rows in flight per wavefront

__device__ __forceinline__ f32x4 mfma16x32(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// barrier for LDS traffic only (__syncthreads() would drain vmcnt: the LDS-DMA units and the stores in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// FOUR LDS-DMA instructions, one row piece each: global (sbase + v_r + r RS) -> LDS (lds_dst + r RS + 16 lane).  The instruction
// offset advances both addresses by RS; the global row stride RB comes from v_r = chunk offset + r (RB - RS).
template <int RS>
__device__ __forceinline__ void glds_rows4(const char* sbase, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t lds_dst) {
  static_assert(3 * RS < 4096 && RS % 16 == 0, "instruction offsets reach 4095");
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %5\n\t"
               "global_load_lds_dwordx4 %2, %5 offset:%c7\n\t"
               "global_load_lds_dwordx4 %3, %5 offset:%c8\n\t"
               "global_load_lds_dwordx4 %4, %5 offset:%c9\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(lds_dst), "n"(RS), "n"(2 * RS), "n"(3 * RS)
               : "memory");
}

}  // namespace

#ifndef MM_S1K_PHASES
#define MM_S1K_PHASES 0    // 1: lane 0 of some wavefronts prints cycle counts per phase (A/B builds; tools/build_variant.sh)
#endif
#if MM_S1K_PHASES
#define S1K_PH(k) do { const long long t_ = __builtin_readcyclecounter(); ph[k] += t_ - t_last; t_last = t_; } while (0)
#else
#define S1K_PH(k) do { } while (0)
#endif

// E = 100 / 200 / 300.  NSTEP = ceil(E / 32) k-steps; wavefront 0 takes the first S0 = ceil(NSTEP / 2), wavefront 1 the rest.
template <int E>
__global__ void __launch_bounds__(128, 2) tkl_stage1_ksplit_kernel(const KpArgs a) {
  constexpr int NSTEP = (E + 31) / 32, S0 = (NSTEP + 1) / 2;
  constexpr int KN0 = 32 * S0 < E ? 32 * S0 : E, KN1 = E - KN0;      // floats per row piece
  constexpr int RS0 = 4 * KN0, RS1 = 4 * KN1;                      // piece bytes = LDS row strides
  constexpr int RB = 4 * E;
  static_assert(KN1 > 0 && E % 4 == 0, "both wavefronts have a piece");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int m = lane & 15, kg = lane >> 4;       // A / B fragments: row (token) m, k-group kg; accumulators: rows 4 kg + x, token m
  const int64_t p0 = (int64_t)blockIdx.x * a.pairs_per_wave;
  const int64_t p1 = (p0 + a.pairs_per_wave < a.n_pairs) ? p0 + a.pairs_per_wave : a.n_pairs;
  if (p0 >= p1) return;
  const int Q = a.Q;
  const int kb = w ? KN0 : 0;                    // first float of this wavefront's K range
  const int kn = w ? KN1 : KN0;                  // floats in it
  const int PBc = kn >> 2;                       // 16-byte chunks per row piece
  const uint32_t RS = w ? RS1 : RS0;
  const uint32_t UB = 8u * RS;                   // a unit (8 rows) in this wavefront's half of the ring
  char* ring = smem + (w ? kRing * 8 * RS0 : 0);
  const uint32_t ring0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
  float* exa = (float*)(smem + kRing * 8 * RB);                    // [dest wavefront][N-tile][k-group of the dest's rows][token] f32x4
  float* exs = exa + 2 * 2 * 2 * 16 * 4;                           // [wavefront][16 rows] partial sums of squares
  const bool rot = !(PBc & 1);                   // even piece: rows rotated by their tile row (see the header)

  // per-lane source offsets of the 16 tile rows: chunk (lane + rotation) mod PBc of the piece + (row & 3) (RB - RS)
  uint32_t vsrc[16];
#pragma unroll
  for (int mr = 0; mr < 16; ++mr) {
    int g = lane + (rot ? mr : 0);
    g = g >= PBc ? g - PBc : g;
    g = g >= PBc ? g - PBc : g;
    vsrc[mr] = (uint32_t)(g * 16) + (uint32_t)(mr & 3) * (RB - RS);
  }
  // per-lane LDS offsets of the two 16-byte chunks of every k-step (tile row m is rotated by m)
  uint32_t pos[S0][2];
#pragma unroll
  for (int s = 0; s < S0; ++s)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int c = 8 * s + 2 * kg + u - (rot ? m : 0);
      c = c < 0 ? c + PBc : c;
      c = c >= PBc ? c - PBc : c;                // (chunks past the piece: any position inside the row, zeroed after the read)
      c = c >= PBc ? c - PBc : c;
      pos[s][u] = (uint32_t)(c * 16);
    }

  // ---- producer: units in stream order (chunk pp, unit pu), ring slot pslot; `young` as in tkl_stage1_rows.hip ----------------------
  const char* dbase = (const char*)a.d + (int64_t)kb * 4;
  int64_t pp = p0;
  int pu = 0, pslot = 0, cslot = 0, inflight = 0;
  uint32_t young = 0;
#if MM_S1K_PHASES
  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long t_last = __builtin_readcyclecounter();
  const long long t_begin = t_last;
#endif
  auto issue_unit = [&](const char* g, uint32_t dst, auto par) __attribute__((always_inline)) {
    constexpr int P8 = 8 * decltype(par)::value;                   // tile rows 8 (pu & 1) + 0..7
    if (lane < PBc) {
      if (w == 0) {
        glds_rows4<RS0>(g, vsrc[P8 + 0], vsrc[P8 + 1], vsrc[P8 + 2], vsrc[P8 + 3], dst);
        glds_rows4<RS0>(g + 4 * RB, vsrc[P8 + 4], vsrc[P8 + 5], vsrc[P8 + 6], vsrc[P8 + 7], dst + 4u * RS0);
      } else {
        glds_rows4<RS1>(g, vsrc[P8 + 0], vsrc[P8 + 1], vsrc[P8 + 2], vsrc[P8 + 3], dst);
        glds_rows4<RS1>(g + 4 * RB, vsrc[P8 + 4], vsrc[P8 + 5], vsrc[P8 + 6], vsrc[P8 + 7], dst + 4u * RS1);
      }
    }
  };
  auto top_up = [&]() __attribute__((always_inline)) {
    if (!(pp < p1 && inflight < kRing)) return;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the LDS reads of the slots about to be overwritten have returned
    while (pp < p1 && inflight < kRing) {
      const char* g = dbase + (pp * 50 + 5 + 8 * pu) * (int64_t)RB;
      const uint32_t dst = ring0 + (uint32_t)pslot * UB;
      if (pu & 1)
        issue_unit(g, dst, std::integral_constant<int, 1>());
      else
        issue_unit(g, dst, std::integral_constant<int, 0>());
      young = (young + 0x01010101u * 8u) << 8;
      pslot = (pslot + 1) & (kRing - 1);
      ++inflight;
      if (++pu == 5) {
        pu = 0;
        ++pp;
      }
    }
  };
  auto wait_oldest = [&](int n_units) __attribute__((always_inline)) {    // the n_units oldest units in flight have landed
    int n = (int)((young >> (8 * (inflight - n_units))) & 0xffu);
    wait_vm(n < 62 ? n : 62);
  };

  // ---- chunk metadata, 64 chunks at a time (tkl_stage1_rows.hip) --------------------------------------------------------------------
  int64_t mbase = p0;
  uint32_t m_slot = 0, m_w0 = 0, m_w1 = 0;
  auto load_meta = [&]() __attribute__((always_inline)) {
    const int64_t pm = mbase + lane < p1 ? mbase + lane : p1 - 1;
    const uint32_t* ps = (const uint32_t*)a.chunk_slot + pm;
    const uint32_t* pb = a.dm.bits + pm * 2;
    asm volatile("global_load_dword %0, %3, off\n\tglobal_load_dword %1, %4, off\n\tglobal_load_dword %2, %4, off offset:4\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(m_slot), "=&v"(m_w0), "=&v"(m_w1) : "v"(ps), "v"(pb) : "memory");
  };
  load_meta();
  top_up();

  // ---- this wavefront's half of the query tile: 2 N-tiles x S0 k-steps, hi / lo, in accumulator registers ----------------------------
  bf16x8 qhi[2][S0], qlo[2][S0];
  float rq[2] = {0.0f, 0.0f};
  int64_t cur_q = -1;
  int qlim = Q;
  int ntile = 1;
  unsigned long long vb = 0;
  float* cbase = nullptr;

  // x of a k-step whose floats run past the piece -> 0 (wavefront 1's last step(s); compile-time: only the steps that can)
  auto clip = [&](f32x4 (&x)[S0][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < S0; ++s) {
      if (32 * (s + 1) > KN1 || 32 * (s + 1) > KN0) {
        const int rem = kn - (32 * s + 8 * kg);
        if (rem < 4) x[s][0] = f32x4{0, 0, 0, 0};
        if (rem < 8) x[s][1] = f32x4{0, 0, 0, 0};
      }
    }
  };

  auto chunk_head = [&](int64_t p) __attribute__((always_inline)) {
    if (p - mbase >= 64) {                                     // (every 64 chunks of a workgroup: drains the stream once)
      mbase = p;
      load_meta();
    }
    const int ml = (int)(p - mbase);
    const int slot = __builtin_amdgcn_readlane((int)m_slot, ml);
    if (w == 0 && a.slot2p && slot >= 0 && slot < a.n_slots) { // (KpArgs::slot2p; wave-uniform condition)
      if (lane == 0) a.slot2p[slot] = (int32_t)((p << 2) | 2);
      young += 0x01010101u;
    }
    const int64_t qi = (int64_t)(slot / a.C);
    const int cpos = slot - (int)qi * a.C;
    if (qi != cur_q) {
      cur_q = qi;
      if (a.qm.len) {
        const int ql = (int)sload_u32(a.qm.len, qi);
        qlim = ql < 0 ? 0 : (ql > Q ? Q : ql);
      }
      float ssq[2] = {0.0f, 0.0f};
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (16 * t >= qlim) break;                             // (wave-uniform: tokens 16..31 only for queries that have them)
        const int tok = 16 * t + m;
        const float* qrow = a.q + (qi * Q + (tok < Q ? tok : Q - 1)) * (int64_t)E + kb + 8 * kg;
        f32x4 raw[S0][2];
#pragma unroll
        for (int s = 0; s < S0; ++s) {
          const int rem = kn - (32 * s + 8 * kg);
          raw[s][0] = *(const f32x4*)(qrow + (rem >= 4 ? 32 * s : 0));
          raw[s][1] = *(const f32x4*)(qrow + (rem >= 8 ? 32 * s + 4 : 0));
        }
        clip(raw);
        float ss = 0.0f;
#pragma unroll
        for (int s = 0; s < S0; ++s) {
          ss += sumsq4(raw[s][0]) + sumsq4(raw[s][1]);
          split8(raw[s][0], raw[s][1], qhi[t][s], qlo[t][s]);
          qhi[t][s] = to_agpr(qhi[t][s]);
          qlo[t][s] = to_agpr(qlo[t][s]);
        }
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        ssq[t] = ss;
      }
      // |q|^2 of a token = this wavefront's half + the other's (through the exchange area, free between tiles)
      if (kg == 0) {
        exa[w * 32 + m] = ssq[0];
        exa[w * 32 + 16 + m] = ssq[1];
      }
      lds_barrier();
#pragma unroll
      for (int t = 0; t < 2; ++t) rq[t] = 1.0f / (sqrtf(exa[16 * t + m] + exa[32 + 16 * t + m]) + 1e-13f);
      lds_barrier();
    }
    ntile = qlim > 16 ? 2 : 1;                                 // wave-uniform
    const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)m_w0, ml), w1 = (uint32_t)__builtin_amdgcn_readlane((int)m_w1, ml);
    vb = ((unsigned long long)(w1 & 0xffu) << 32) | w0;
    cbase = a.cos_out + qi * ((int64_t)a.C * 40 * Q) + (int64_t)cpos * 40 * qlim;
  };

  const bool own = (kg >> 1) == w;                             // this lane's accumulator rows 4 kg + 0..3 are finished by this wavefront
  const int kgl = kg & 1;                                      // ... k-group inside the owner's half

  int64_t p = p0;
  int tau = 0;
#pragma unroll 1
  for (;;) {
    if (tau == 0) {
      S1K_PH(0);
      chunk_head(p);
      S1K_PH(5);
    }
    // ---- the tile's rows (this wavefront's k range) -> registers, its units back to the producer ------------------------------------
    const int nun = tau < 2 ? 2 : 1;
    top_up();
    S1K_PH(1);
    wait_oldest(nun);
    S1K_PH(2);
    f32x4 x[S0][2];
    {
      // row m of the tile: unit (m >> 3) of the tile (tau = 2: its only unit; rows 8..15 repeat rows 0..7 and are not stored)
      const int us = (cslot + (nun == 2 ? (m >> 3) : 0)) & (kRing - 1);
      const char* row = ring + (uint32_t)us * UB + (uint32_t)(m & 7) * RS;
#pragma unroll
      for (int s = 0; s < S0; ++s) {
        x[s][0] = *(const f32x4*)(row + pos[s][0]);
        x[s][1] = *(const f32x4*)(row + pos[s][1]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    cslot = (cslot + nun) & (kRing - 1);
    inflight -= nun;
    S1K_PH(3);
    top_up();                                                  // (waits for the reads above, then refills the slots)
    S1K_PH(4);

    if (qlim > 0) {
      clip(x);
      f32x4 hh[2], lh[2], hl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) hh[t] = lh[t] = hl[t] = f32x4{0, 0, 0, 0};
      f32x2 ss2 = {0.0f, 0.0f};
      auto products = [&](auto nt) __attribute__((always_inline)) {
        constexpr int NT = decltype(nt)::value;
#pragma unroll
        for (int s = 0; s < S0; ++s) {
          bf16x8 ah, al;
          split8(x[s][0], x[s][1], ah, al);
          {
            const f32x2 a0 = {x[s][0][0], x[s][0][1]}, a1 = {x[s][0][2], x[s][0][3]};
            const f32x2 b0 = {x[s][1][0], x[s][1][1]}, b1 = {x[s][1][2], x[s][1][3]};
            ss2 += a0 * a0;
            ss2 += a1 * a1;
            ss2 += b0 * b0;
            ss2 += b1 * b1;
          }
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            hh[t] = mfma16x32(ah, qhi[t][s], hh[t]);
            lh[t] = mfma16x32(al, qhi[t][s], lh[t]);
            hl[t] = mfma16x32(ah, qlo[t][s], hl[t]);
          }
        }
      };
      if (ntile == 2)
        products(std::integral_constant<int, 2>());
      else
        products(std::integral_constant<int, 1>());
      S1K_PH(6);
      float ss = ss2[0] + ss2[1];
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      f32x4 acc[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t] = hh[t] + (lh[t] + hl[t]);
      // ---- exchange: the other wavefront's rows out, this wavefront's rows in ---------------------------------------------------------
      if (kg == 0) exs[w * 16 + m] = ss;
      if (!own) {
        f32x4* dst = (f32x4*)exa + (((1 - w) * 2 + 0) * 2 + kgl) * 16 + m;
        dst[0] = acc[0];
        if (ntile == 2) dst[2 * 16] = acc[1];
      }
      lds_barrier();
      f32x4 oth[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
      f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
      if (own) {
        const f32x4* src = (const f32x4*)exa + ((w * 2 + 0) * 2 + kgl) * 16 + m;
        oth[0] = src[0];
        if (ntile == 2) oth[1] = src[2 * 16];
        s0 = *(const f32x4*)(exs + 4 * kg);
        s1 = *(const f32x4*)(exs + 16 + 4 * kg);
      }
      lds_barrier();
      S1K_PH(7);
      const int row0 = 16 * tau + 4 * kg;
      const bool rows_exist = row0 < 40;                       // tau = 2: k-groups 0 and 1 (wavefront 0)
      const bool stores = tau < 2 || w == 0;                   // wave-uniform: this wavefront finishes rows of this tile
      if (stores) {
        const uint32_t bits = (uint32_t)(vb >> (row0 < 40 ? row0 : 0)) & 0xfu;
        f32x4 rd;
#pragma unroll
        for (int i = 0; i < 4; ++i) rd[i] = 1.0f / (sqrtf(s0[i] + s1[i]) + 1e-13f);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (t < ntile) {
            const int tok = 16 * t + m;
            if (own && tok < qlim && rows_exist) {
              float* dst = cbase + (int64_t)row0 * qlim + tok;
              const f32x4 tot = w == 0 ? acc[t] + oth[t] : oth[t] + acc[t];   // (wavefront 0's half first, either way)
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float c = (tot[i] * rq[t]) * rd[i];
                dst[i * qlim] = ((bits >> i) & 1u) ? c : 1.0e5f;
              }
            }
          }
        }
        young += 0x01010101u * (uint32_t)(4 * ntile);          // the stores above are in the vmcnt queue too
      }
    }
    const bool last = p + 1 >= p1 && tau == 2;
    if (last) break;
    if (++tau == 3) {
      tau = 0;
      ++p;
    }
  }
#if MM_S1K_PHASES
  S1K_PH(0);
  if ((blockIdx.x % 61 == 0) && lane == 0)
    printf("S1K wg %4d w %d chunks %d | finish+stores+loop %lld | top-up %lld | wait units %lld | LDS reads issue %lld | reads back + DMA issue %lld | chunk head %lld | products %lld | exchange %lld | total %lld\n",
           (int)blockIdx.x, w, (int)(p1 - p0), ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7], t_last - t_begin);
#endif
}

bool tkl_stage1_ksplit_supported(int Q, int E) { return Q <= 32 && (E == 100 || E == 200 || E == 300); }

int tkl_stage1_ksplit_launch(const KpArgs& a0, hipStream_t stream) {
  KpArgs a = a0;
  int64_t wgs = (int64_t)kCUs * 4;
  if (wgs > a.n_pairs) wgs = a.n_pairs;
  if (wgs <= 0) return MM_OK;
  a.pairs_per_wave = (a.n_pairs + wgs - 1) / wgs;
  wgs = (a.n_pairs + a.pairs_per_wave - 1) / a.pairs_per_wave;
  const int lds = kRing * 8 * a.E * 4 + 2 * 2 * 2 * 16 * 16 + 2 * 16 * 4;
  const dim3 grid((unsigned)wgs), block(128);
  if (a.E == 100)
    hipLaunchKernelGGL((tkl_stage1_ksplit_kernel<100>), grid, block, lds, stream, a);
  else if (a.E == 200)
    hipLaunchKernelGGL((tkl_stage1_ksplit_kernel<200>), grid, block, lds, stream, a);
  else
    hipLaunchKernelGGL((tkl_stage1_ksplit_kernel<300>), grid, block, lds, stream, a);
  return check_launch("tkl_stage1_ksplit_kernel");
}

}  // namespace mm
