// TKL stage 1, row-streaming form (round 6): the cosine match of sigir20_tkl.py:184-194 for the cosine hand-off to the window
// kernel (KpArgs::cos_out, tkl.hip) — what tkl_stage1_run_kernel<COS> of kernel_pool.hip computes, with another HBM stream.
//
// tkl_stage1_run_kernel cuts the 1,200-byte rows of a chunk (E = 300) into three 400-byte K-slices so that a ring slot can go
// back to its producer after a third of a block: every LDS-DMA instruction then gathers 2.56 row PIECES (13 cache lines for
// 1,024 bytes, neighbouring slices sharing lines), the three pieces of a row are requested a slice period apart, and stage 1
// streamed 4.7 TB/s.  Here the 40 centre rows of a chunk — ONE contiguous 48,000-byte run — are streamed as
// such: five units of 8 rows (9,600 B = 9 full LDS-DMA instructions + one of 24 lanes), each a linear image of global memory in
// its ring slot.  A row is whole in LDS when its unit lands, so the matrix products run over M = 16 document rows at full K
// (v_mfma_f32_16x16x32_bf16: A = 16 rows x 32 k, B = 16 query tokens x 32 k; row stride 1,200 B = 75 x 16 B, odd: the b128 reads
// of the sixteen rows of a k-group are conflict-free) and a tile's two units go back to the producer as soon as their rows
// sit in registers.  Ring: 4 units = 38,400 B per one-wavefront workgroup, four workgroups per CU as before.
//
// Arithmetic: the three-product bf16 split of the TK / TKL kernels (x = hi + lo: hi·hi + lo·hi + hi·lo, fp32 accumulation), the
// norms in fp32 from the same registers, cosine = (dot · 1 / (|q| + 1e-13)) · 1 / (|d| + 1e-13) in that order (CosineMatrixAttention
// as the other kernels apply it), masked positions = 1e5 (every RBF kernel underflows to exactly 0 in the window kernel).
#include <type_traits>

#include "kp_device.h"

namespace mm {

namespace {

constexpr int kRing = 4;       // ring slots = units of 8 chunk rows in flight per wavefront
constexpr int kUnitRows = 8;

// sum over the four k-groups of a row (lanes l, l ^ 16, l ^ 32, l ^ 48), in every lane: two gfx950 lane swaps + two adds on the
// VALU (two ds_bpermute round trips before: ~200 cycles each with nothing else resident on the SIMD)
__device__ __forceinline__ float kgroup_sum(float v) {
  // v_permlane16_swap x, y: rows 1, 3 of x <-> rows 0, 2 of y; v_permlane32_swap: the upper half of x <-> the lower half of y.
  // (Inline assembly: __builtin_amdgcn_permlane16_swap(u, u) came back with the same value in both results on ROCm 7.2 —
  // tools/scratch/swap.hip.)
  float x = v, y;
  asm volatile("v_mov_b32 %1, %2\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "=&v"(y) : "v"(v));
  const float s = x + y;                                       // rows 0, 1: r0 + r1; rows 2, 3: r2 + r3
  float x2 = s, y2;
  asm volatile("v_mov_b32 %1, %2\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x2), "=&v"(y2) : "v"(s));
  return x2 + y2;
}

__device__ __forceinline__ f32x4 mfma16x32(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// LDS-DMA instructions of 64 lanes x 16 bytes each, global (sbase + voff + 1024 n) -> LDS (lds_dst + 1024 n + 16 lane).  The
// instruction offset advances the global AND the LDS address, so four instructions share one M0 / base pair: a 4-KiB group costs
// 4 + 3 instructions instead of the 5 per instruction of a one-at-a-time wrapper (the wavefront is alone on its SIMD: every
// instruction it issues is ~10 cycles of its critical path, profiles/r06_experiments/tkl_stage1_rows_phases.txt).
template <int N>
__device__ __forceinline__ void glds_group(const char* sbase, uint32_t voff, uint32_t lds_dst) {
  static_assert(N >= 1 && N <= 4, "instruction offsets reach 3 x 1024");
  uint32_t keep;
  if constexpr (N == 4)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
  else if constexpr (N == 3)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
  else if constexpr (N == 2)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// the NFULL full-width instructions of a unit
template <int NFULL>
__device__ __forceinline__ void glds_run(const char* sbase, uint32_t voff, uint32_t lds_dst) {
  if constexpr (NFULL >= 4) {
    glds_group<4>(sbase, voff, lds_dst);
    glds_run<NFULL - 4>(sbase + 4096, voff, lds_dst + 4096u);
  } else if constexpr (NFULL >= 1) {
    glds_group<NFULL>(sbase, voff, lds_dst);
  }
}

}  // namespace

#ifndef MM_S1_PHASES
#define MM_S1_PHASES 0     // 1: lane 0 of the middle wavefront prints its cycle counts per phase (A/B builds; tools/build_variant.sh)
#endif
#if MM_S1_PHASES
#define S1_PH(k) do { const long long t_ = __builtin_readcyclecounter(); ph[k] += t_ - t_last; t_last = t_; } while (0)
#else
#define S1_PH(k) do { } while (0)
#endif
#ifndef MM_S1_CUT
#define MM_S1_CUT 0        // A/B builds (timing only, wrong results): 1 = no matrix products, 2 = no hi / lo split, 3 = no norms
#endif
#ifndef MM_S1_PROBE
#define MM_S1_PROBE 0      // A/B builds: 1 = the stream alone (units requested, awaited, released); 2 = + the LDS reads; 3 = + the arithmetic, no stores
#endif

// NSTEP = ceil(E / 32) K-steps of 32 = LDS-DMA instructions per unit (a unit is 8 rows x 4 E bytes = 32 E bytes = E / 32 KiB)
template <int NSTEP>
__global__ void __launch_bounds__(64) tkl_stage1_rows_kernel(const KpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int m = lane & 15, kg = lane >> 4;       // A / B fragments: row (token) m, k-group kg; accumulators: rows 4 kg + x, token m
  const int64_t p0 = (int64_t)blockIdx.x * a.pairs_per_wave;
  const int64_t p1 = (p0 + a.pairs_per_wave < a.n_pairs) ? p0 + a.pairs_per_wave : a.n_pairs;
  if (p0 >= p1) return;
  const int E = a.E, Q = a.Q;
  const uint32_t RB = (uint32_t)E * 4u, UB = kUnitRows * RB;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  float* rdbuf = (float*)(smem + kRing * UB + 128);           // (128 B behind the ring: the last row's reads past E stay inside)
  float* timg = rdbuf + 16;                                   // [16 rows][qlim <= 32] image of a tile's cosines
  const int tail_lanes = (E & 31) ? 2 * (E & 31) : 64;         // lanes of a unit's last LDS-DMA instruction
  const uint32_t vlane = (uint32_t)lane * 16u;

  // ---- producer: units in stream order (chunk pp, unit pu), ring slot pslot ---------------------------------------------------
  // `young`: one byte per unit in flight, youngest in byte 0 — the vector-memory operations issued AFTER that unit's last
  // LDS-DMA instruction (vmcnt retires in order: the unit has landed once no more than that many operations are outstanding).
  const char* dbase = (const char*)a.d;
  int64_t pp = p0;
  int pu = 0, pslot = 0, cslot = 0, inflight = 0;
  uint32_t young = 0;
  auto top_up = [&]() __attribute__((always_inline)) {
    if (!(pp < p1 && inflight < kRing)) return;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the LDS reads of the slots about to be overwritten have returned
    while (pp < p1 && inflight < kRing) {
      const char* g = dbase + (pp * 50 + 5 + kUnitRows * pu) * (int64_t)RB;
      const uint32_t dst = lds0 + (uint32_t)pslot * UB;
      glds_run<NSTEP - 1>(g, vlane, dst);
      if (lane < tail_lanes) glds_group<1>(g + 1024 * (NSTEP - 1), vlane, dst + 1024u * (NSTEP - 1));
      young = (young + 0x01010101u * NSTEP) << 8;
      pslot = (pslot + 1) & (kRing - 1);
      ++inflight;
      if (++pu == 5) {
        pu = 0;
        ++pp;
      }
    }
  };
#if MM_S1_PHASES
  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long t_last = __builtin_readcyclecounter();
  const long long t_begin = t_last;
#endif
  auto wait_oldest = [&](int n_units) __attribute__((always_inline)) {    // the n_units oldest units in flight have landed
    // (vmcnt takes an immediate: the count is floored to a unit boundary — at most one store or two short of exact, i.e. a few
    // hundred bytes of the next unit awaited with it — instead of a 63-way jump table, 340 cycles per tile)
    const int n = (int)((young >> (8 * (inflight - n_units))) & 0xffu);
    if (n >= 3 * NSTEP)
      wait_vm(3 * NSTEP);
    else if (n >= 2 * NSTEP)
      wait_vm(2 * NSTEP);
    else if (n >= NSTEP)
      wait_vm(NSTEP);
    else
      wait_vm(0);
  };
  // ---- chunk metadata, 64 chunks at a time: lane l holds the slot and the 40 validity bits of chunk mbase + l, fetched by ONE
  // round of loads BEFORE the stream starts and read per chunk with v_readlane.  (Per chunk: three dependent scalar loads and —
  // through the slot-map publish — a vector load whose compiler-placed s_waitcnt vmcnt(0) drained the whole LDS-DMA queue:
  // 4.6 k of a chunk's 21 k cycles.)
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

  // ---- query tile: 2 N-tiles (tokens 0..15, 16..31) x NSTEP k-steps, split hi / lo, in accumulator registers --------------------
  bf16x8 qhi[2][NSTEP], qlo[2][NSTEP];
  float rq[2] = {0.0f, 0.0f};
  int64_t cur_q = -1;
  int qlim = Q;
  // floats of this lane's k-group that exist in the LAST k-step (0, 4 or 8: E is a multiple of 4)
  const int last_valid = E - (32 * (NSTEP - 1) + 8 * kg);

  // The 16 rows of a tile -> registers (A fragments of all k-steps), its units back to the producer.  Software pipeline: the rows
  // of tile i + 1 are fetched from LDS BEFORE tile i is multiplied, so a landed unit waits in LDS for one LDS read, not for a
  // tile's arithmetic, and all four ring slots are in flight while the matrix pipe works (the stream alone, MM_S1_PROBE: 51.7 us
  // at 256 documents; with the reads behind the arithmetic 75.8: half the ring sat landed and idle).
  auto read_tile = [&](f32x4 (&x)[NSTEP][2], int tau) __attribute__((always_inline)) {
    const int nun = tau < 2 ? 2 : 1;
    S1_PH(0);                                                  // (arithmetic + stores of the previous tile, loop overhead)
    top_up();
    S1_PH(1);
    wait_oldest(nun);
    S1_PH(2);                                                  // waiting for the units
#if MM_S1_PROBE != 1
    {
      // row m of the tile: unit (m >> 3) of the tile (tau = 2: its only unit, rows 8..15 repeat rows 0..7 and are not stored)
      const int us = (cslot + (nun == 2 ? (m >> 3) : 0)) & (kRing - 1);
      const char* row = smem + (uint32_t)us * UB + (uint32_t)(m & 7) * RB + kg * 32;
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        x[s][0] = *(const f32x4*)(row + 128 * s);
        x[s][1] = *(const f32x4*)(row + 128 * s + 16);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#endif
    cslot = (cslot + nun) & (kRing - 1);
    inflight -= nun;
    S1_PH(3);                                                  // LDS reads issued
    // (the freed slots are refilled from inside the k-loop of the tile multiplied next: compute_tile)
  };

  // per-chunk state of the tile being multiplied
  int ntile = 1;
  unsigned long long vb = 0;
  float* cbase = nullptr;
  auto compute_tile = [&](f32x4 (&x)[NSTEP][2], int tau) __attribute__((always_inline)) {
#if MM_S1_PROBE == 0 || MM_S1_PROBE == 3
    if (qlim <= 0) return;
    if (last_valid < 4) x[NSTEP - 1][0] = f32x4{0, 0, 0, 0};
    if (last_valid < 8) x[NSTEP - 1][1] = f32x4{0, 0, 0, 0};
    f32x4 hh[2], lh[2], hl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) hh[t] = lh[t] = hl[t] = f32x4{0, 0, 0, 0};
    f32x2 ssa[4] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}};
    // (the choice between one and two N-tiles OUTSIDE the k-loop: a branch per step made the register allocator shuffle the
    // accumulators through v_accvgpr moves at every step)
    // The refill of the slots the pre-read just freed is issued from INSIDE the k-loop, two LDS-DMA instructions per k-step.
    // Issuing is the transfer (the queue to the memory system is shallow: an instruction is accepted at the CU's share of the
    // HBM stream, ~80 cycles per KiB) and four wavefronts share that path: with a 20-instruction burst per tile each was in its
    // issue phase ~30 % of the time and the path idle whenever all four computed at once ((1 - 0.3)^4 = 24 %: the measured 74 % of
    // the stream-only rate).  Interleaved, every wavefront offers the path work throughout its tile.
    bool act = false;
    const char* ug = nullptr;
    uint32_t ud = 0;
    auto dma_step = [&](int s) __attribute__((always_inline)) {          // (s: a constant after unrolling)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int i0 = (2 * s) % NSTEP;
        const bool paired = i0 + 1 < NSTEP - 1;                // instructions i0, i0 + 1: both full-width, one M0 / base pair
        if (jj == 1 && paired) continue;
        const int i = (2 * s + jj) % NSTEP;
        if (i == 0) {
          act = pp < p1 && inflight < kRing;
          if (act) {
            ug = dbase + (pp * 50 + 5 + kUnitRows * pu) * (int64_t)RB;
            ud = lds0 + (uint32_t)pslot * UB;
          }
        }
        if (act) {
          if (i < NSTEP - 1) {
            if (jj == 0 && paired)
              glds_group<2>(ug + 1024 * i, vlane, ud + 1024u * i);
            else
              glds_group<1>(ug + 1024 * i, vlane, ud + 1024u * i);
          } else {
            if (lane < tail_lanes) glds_group<1>(ug + 1024 * i, vlane, ud + 1024u * i);
            young = (young + 0x01010101u * NSTEP) << 8;
            pslot = (pslot + 1) & (kRing - 1);
            ++inflight;
            if (++pu == 5) {
              pu = 0;
              ++pp;
            }
          }
        }
      }
    };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the pre-read of the next tile has returned: its slots may be overwritten
    auto products = [&](auto nt) __attribute__((always_inline)) {
      constexpr int NT = decltype(nt)::value;
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        dma_step(s);
        bf16x8 ah, al;
#if MM_S1_CUT == 2
        ah = __builtin_bit_cast(bf16x8, x[s][0]);
        al = __builtin_bit_cast(bf16x8, x[s][1]);
#else
        split8(x[s][0], x[s][1], ah, al);
#endif
#if MM_S1_CUT != 3
        {
          const f32x2 a0 = {x[s][0][0], x[s][0][1]}, a1 = {x[s][0][2], x[s][0][3]};
          const f32x2 b0 = {x[s][1][0], x[s][1][1]}, b1 = {x[s][1][2], x[s][1][3]};
          ssa[0] += a0 * a0;                                    // (four chains: one accumulator was 40 dependent v_pk_fma per tile,
          ssa[1] += a1 * a1;                                    //  ~13 cycles each with nothing else resident on the SIMD)
          ssa[2] += b0 * b0;
          ssa[3] += b1 * b1;
        }
#endif
#if MM_S1_CUT == 1
        asm volatile("" ::"v"(ah), "v"(al));
#else
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          hh[t] = mfma16x32(ah, qhi[t][s], hh[t]);
          lh[t] = mfma16x32(al, qhi[t][s], lh[t]);
          hl[t] = mfma16x32(ah, qlo[t][s], hl[t]);
        }
#endif
      }
    };
    S1_PH(0);
    if (ntile == 2)
      products(std::integral_constant<int, 2>());
    else
      products(std::integral_constant<int, 1>());
    S1_PH(6);                                                  // k-loop: split, norms, products
    const f32x2 ss2 = (ssa[0] + ssa[1]) + (ssa[2] + ssa[3]);
    const float ss = kgroup_sum(ss2[0] + ss2[1]);
    if (kg == 0) rdbuf[m] = 1.0f / (sqrtf(ss) + 1e-13f);       // row m of the tile
    const f32x4 rd = *(const f32x4*)(rdbuf + 4 * kg);          // rows 4 kg + 0..3 (same wavefront: program order)
    const int row0 = 16 * tau + 4 * kg;
    const bool rows_exist = row0 < 40;                         // tau = 2: k-groups 0 and 1
    S1_PH(7);                                                  // norm reduction + redistribution
    const uint32_t bits = (uint32_t)(vb >> (row0 < 40 ? row0 : 0)) & 0xfu;
    // The tile's cosines leave as ONE contiguous run: rows 16 tau .. of the chunk x qlim tokens are nrows x qlim consecutive
    // floats of the hand-off buffer.  From the accumulator layout (lane = token, four rows 4 apart in position) that was 4 store
    // instructions per N-tile of <= 16 x 4 B pieces; through a wavefront-private LDS image of the tile it is one 16-byte store
    // per lane (a second instruction only past 256 floats): the vector-memory queue is what this wavefront waits on.
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t < ntile) {
        const int tok = 16 * t + m;
        if (tok < qlim && rows_exist) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float c = ((hh[t][i] + (lh[t][i] + hl[t][i])) * rq[t]) * rd[i];
            timg[(4 * kg + i) * qlim + tok] = ((bits >> i) & 1u) ? c : 1.0e5f;
          }
        }
      }
    }
    const int nfl = (tau < 2 ? 16 : 8) * qlim;                 // floats of the run (a multiple of 4)
    float* run = cbase + (int64_t)(16 * tau) * qlim;
#if MM_S1_PROBE == 0
    if (4 * lane < nfl) {
      const f32x4 v = *(const f32x4*)(timg + 4 * lane);
      float* d = run + 4 * lane;
      asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(d), "v"(v) : "memory");
    }
    if (nfl > 256) {
      if (4 * lane + 256 < nfl) {
        const f32x4 v = *(const f32x4*)(timg + 256 + 4 * lane);
        float* d = run + 256 + 4 * lane;
        asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(d), "v"(v) : "memory");
      }
    }
#endif
#if MM_S1_PROBE == 0
    young += 0x01010101u * (nfl > 256 ? 2u : 1u);              // the stores above are in the vmcnt queue too
#endif
#endif
  };

  auto chunk_head = [&](int64_t p) __attribute__((always_inline)) {
    if (p - mbase >= 64) {                                     // (every 64 chunks of a wavefront: drains the stream once)
      mbase = p;
      load_meta();
    }
    const int ml = (int)(p - mbase);
    const int slot = __builtin_amdgcn_readlane((int)m_slot, ml);
    if (a.slot2p && slot >= 0 && slot < a.n_slots) {           // (KpArgs::slot2p; wave-uniform condition)
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
#if MM_S1_PROBE == 0 || MM_S1_PROBE == 3
#pragma unroll
      for (int t = 0; t < 2; ++t) {                            // (one N-tile at a time: 20 x 16 B in flight beside a pre-read tile)
        if (16 * t >= qlim) break;                             // (wave-uniform: tokens 16..31 only for queries that have them)
        const int tok = 16 * t + m;
        const float* qrow = a.q + (qi * Q + (tok < Q ? tok : Q - 1)) * (int64_t)E + 8 * kg;
        f32x4 raw[NSTEP][2];
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
          const bool v0 = s + 1 < NSTEP || last_valid >= 4, v1 = s + 1 < NSTEP || last_valid >= 8;
          raw[s][0] = *(const f32x4*)(qrow + (v0 ? 32 * s : 0));
          raw[s][1] = *(const f32x4*)(qrow + (v1 ? 32 * s + 4 : 0));
        }
        float ss = 0.0f;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
          if (s + 1 == NSTEP) {
            if (last_valid < 4) raw[s][0] = f32x4{0, 0, 0, 0};
            if (last_valid < 8) raw[s][1] = f32x4{0, 0, 0, 0};
          }
          ss += sumsq4(raw[s][0]) + sumsq4(raw[s][1]);
          split8(raw[s][0], raw[s][1], qhi[t][s], qlo[t][s]);
          qhi[t][s] = to_agpr(qhi[t][s]);
          qlo[t][s] = to_agpr(qlo[t][s]);
        }
        rq[t] = 1.0f / (sqrtf(kgroup_sum(ss)) + 1e-13f);
      }
#endif
    }
    ntile = qlim > 16 ? 2 : 1;                                 // wave-uniform
    // validity bits of the chunk's 40 centre tokens
    const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)m_w0, ml), w1 = (uint32_t)__builtin_amdgcn_readlane((int)m_w1, ml);
    vb = ((unsigned long long)(w1 & 0xffu) << 32) | w0;
    cbase = a.cos_out + qi * ((int64_t)a.C * 40 * Q) + (int64_t)cpos * 40 * qlim;
  };

  // Two tiles per loop trip, the register images alternating (xa multiplied while xb is fetched, then the reverse): no copy, two
  // copies of the body.  (The six-fold unrolled form — a chunk pair's six tiles — was 13 k instructions for E = 300, more than the
  // instruction cache holds: 84 us; one copy with a 40-register move per tile: ~200 cycles of a tile's 5.7 k.)
  f32x4 xa[NSTEP][2], xb[NSTEP][2];
  int64_t p = p0;
  int tau = 0;
  bool done = false;
  auto step = [&](f32x4 (&xc)[NSTEP][2], f32x4 (&xn)[NSTEP][2]) __attribute__((always_inline)) {
    if (tau == 0) {
      S1_PH(0);
      chunk_head(p);
      S1_PH(5);                                                // chunk head (query tile when the document changes)
    }
    const bool last = p + 1 >= p1 && tau == 2;
    if (!last) read_tile(xn, tau == 2 ? 0 : tau + 1);
#if MM_S1_PROBE == 2
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) asm volatile("" ::"v"(xc[s][0]), "v"(xc[s][1]));
#endif
    compute_tile(xc, tau);
    S1_PH(0);
    top_up();                                                  // (whatever the k-loop did not issue: a skipped tile, the ramp)
    S1_PH(4);
    done = last;
    if (++tau == 3) {
      tau = 0;
      ++p;
    }
  };
  read_tile(xa, 0);
#pragma unroll 1
  for (;;) {
    step(xa, xb);
    if (done) break;
    step(xb, xa);
    if (done) break;
  }
#if MM_S1_PHASES
  S1_PH(0);
  if ((blockIdx.x % 37 == 0 || blockIdx.x == gridDim.x / 2) && lane == 0)
    printf("S1PH wave %4d chunks %d | copy+stores+loop %lld | top-up before wait %lld | wait units %lld | LDS reads issue %lld | reads back + DMA issue %lld | chunk head %lld | k-loop %lld | norm %lld | total %lld\n",
           (int)blockIdx.x, (int)(p1 - p0), ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7], t_last - t_begin);
#endif
}

bool tkl_stage1_rows_supported(int Q, int E) { return Q <= 32 && (E == 100 || E == 200 || E == 300); }

int tkl_stage1_rows_launch(const KpArgs& a0, hipStream_t stream) {
  KpArgs a = a0;
#ifndef MM_S1_WAVES_PER_CU   // A/B builds only: 4 are resident (LDS); more give shorter ranges handed out as wavefronts retire
#define MM_S1_WAVES_PER_CU 4
#endif
  int64_t waves = (int64_t)kCUs * MM_S1_WAVES_PER_CU;
  if (waves > a.n_pairs) waves = a.n_pairs;
  if (waves <= 0) return MM_OK;
  a.pairs_per_wave = (a.n_pairs + waves - 1) / waves;
  waves = (a.n_pairs + a.pairs_per_wave - 1) / a.pairs_per_wave;
  const int lds = kRing * kUnitRows * a.E * 4 + 128 + 64 + 16 * 32 * 4;
  const dim3 grid((unsigned)waves), block(64);
  if (a.E == 100)
    hipLaunchKernelGGL((tkl_stage1_rows_kernel<4>), grid, block, lds, stream, a);
  else if (a.E == 200)
    hipLaunchKernelGGL((tkl_stage1_rows_kernel<7>), grid, block, lds, stream, a);
  else
    hipLaunchKernelGGL((tkl_stage1_rows_kernel<10>), grid, block, lds, stream, a);
  return check_launch("tkl_stage1_rows_kernel");
}

}  // namespace mm
