// Device helpers shared by the MaxSim translation units (maxsim.hip, maxsim_pair.hip).  gfx950 only.
#pragma once
#include "mm_internal.h"

namespace mm {

struct MaxsimArgs {
  const void* q;
  const void* d;
  PackedMask qm;  // rows = queries
  PackedMask dm;  // rows = documents
  float* out;
  int64_t n_pairs;
  int64_t ppq;     // pairs per query (paired mode)
  int64_t inb_bd;  // > 0: all-pairs mode, pair p = (query p / Bd, doc p % Bd)
  int inb_bug;     // all-pairs: mask with the document row of the *query* index (colbert.py:158)
  int64_t inb_bq;  // all-pairs: number of queries
  int inb_t, inb_gw;  // all-pairs tiled over queries: document slices per XCD, query-group lanes (maxsim.hip, INB = 2)
  int Q, D, E;
  int64_t pairs_per_wave;
  // ragged (CSR) documents: document p = rows [rag_begin[p], rag_end[p]) of the token matrix `d`
  // (the reference's on-disk store: token_reps_N.npy + doc_infos, dense_retrieval.py:201-280)
  const int64_t* rag_begin;
  const int64_t* rag_end;
  // pair-per-row kernel (maxsim_pair.hip): the tokenizer's int64 masks, read in the kernel itself
  const int64_t* qm64;
  const int64_t* dm64;
  int rnd;         // MM_SIM_ROUND | MM_SUM_ROUND: the reference's dtype flow for 16-bit inputs (round_like below)
};

// maxsim_pair.hip
bool maxsim_pair_supported(int Q, int E, int dtype);
bool maxsim_pair_i64_supported(int Q, int D, const void* qm, const void* dm);
int maxsim_pair_launch(const MaxsimArgs& a, int dtype, bool i64, hipStream_t stream);

template <int DT>
struct Mfma32x16;
template <>
struct Mfma32x16<MM_BF16> {
  static __device__ __forceinline__ f32x16 run(short8 a, short8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <>
struct Mfma32x16<MM_F16> {
  static __device__ __forceinline__ f32x16 run(short8 a, short8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

// C/D layout of the 32x32 MFMA: lane l holds column (l & 31), rows rowof(i) + 4*(l >> 5).
__device__ __forceinline__ constexpr int rowof(int i) { return (i & 3) + 8 * (i >> 2); }

// Running max of one 32-row document block into m[16].
//   ex: bit r set <=> row r of the block is below the document's effective length
//   va: bit r set <=> row r is a real token (va is a subset of ex)
//   fill: value of rows outside ex (-1000 if the document has padding at all, else -inf: rows
//         past D do not exist and must not take part in the max).
__device__ __forceinline__ void block_max(float (&m)[16], const f32x16& acc, uint32_t ex, uint32_t va, float fill, int h) {
  if (va == 0xffffffffu) {
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = fmaxf(m[i], acc[i]);
  } else {
    const uint32_t exs = ex >> (4 * h), vas = va >> (4 * h);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int bit = rowof(i);
      float v = ((vas >> bit) & 1u) ? acc[i] : (((exs >> bit) & 1u) ? -1000.0f : fill);
      m[i] = fmaxf(m[i], v);
    }
  }
}

// Wave-uniform 32-bit load through the scalar cache.  The compiler cannot use s_load here on its
// own (the asm "memory" clobbers of the LDS-DMA pipeline make every global look written), and a
// vector load would make it wait vmcnt(0) and drain the D stream once per pair.  Lengths / mask
// words are never written by these kernels, so the (non-coherent) scalar cache is safe.
__device__ __forceinline__ uint32_t sload_u32(const void* base, int64_t idx) {
  uint32_t v;
  const uint32_t* p = (const uint32_t*)base + idx;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p));
  return v;
}

__device__ __forceinline__ int64_t sload_i64(const int64_t* base, int64_t idx) {
  int64_t v;
  const int64_t* p = base + idx;
  asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p));
  return v;
}

// Query-tile B fragments: 8 x 16 B per lane at base + kk*32.  Loaded in asm together with their
// own vmcnt(0) so the compiler never plants vmcnt(N) waits for them inside the block loop (those
// would also drain the hidden LDS-DMA queue).  Runs once per query, so the drain is harmless.
__device__ __forceinline__ void load_q_frags(const char* base, short8 (&qf)[8]) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off\n\t"
      "global_load_dwordx4 %1, %8, off offset:32\n\t"
      "global_load_dwordx4 %2, %8, off offset:64\n\t"
      "global_load_dwordx4 %3, %8, off offset:96\n\t"
      "global_load_dwordx4 %4, %8, off offset:128\n\t"
      "global_load_dwordx4 %5, %8, off offset:160\n\t"
      "global_load_dwordx4 %6, %8, off offset:192\n\t"
      "global_load_dwordx4 %7, %8, off offset:224\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(qf[0]), "=&v"(qf[1]), "=&v"(qf[2]), "=&v"(qf[3]), "=&v"(qf[4]), "=&v"(qf[5]), "=&v"(qf[6]),
        "=&v"(qf[7])
      : "v"(base)
      : "memory");
}

// The same into ACCUMULATOR registers: the fragments stay there and the MFMA reads its B operand from AGPRs directly.  Two
// query tiles at dim 768 are 384 registers of B fragments; left to itself the allocator parks the overflow in AGPRs and
// fetches every parked fragment back with four v_accvgpr_read before each use (284 of them per 32-token block).
__device__ __forceinline__ void load_q_frags_agpr(const char* base, short8 (&qf)[8]) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off\n\t"
      "global_load_dwordx4 %1, %8, off offset:32\n\t"
      "global_load_dwordx4 %2, %8, off offset:64\n\t"
      "global_load_dwordx4 %3, %8, off offset:96\n\t"
      "global_load_dwordx4 %4, %8, off offset:128\n\t"
      "global_load_dwordx4 %5, %8, off offset:160\n\t"
      "global_load_dwordx4 %6, %8, off offset:192\n\t"
      "global_load_dwordx4 %7, %8, off offset:224\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&a"(qf[0]), "=&a"(qf[1]), "=&a"(qf[2]), "=&a"(qf[3]), "=&a"(qf[4]), "=&a"(qf[5]), "=&a"(qf[6]),
        "=&a"(qf[7])
      : "v"(base)
      : "memory");
}

// The same with ONE running maximum per lane (the 16 rows a lane holds all belong to its query token, and max is
// exact in any order): 8 v_max3 per block instead of 16 v_max, 1 register instead of 16 — what lets the all-pairs
// kernel keep four query tiles AND their accumulators in VGPRs.
__device__ __forceinline__ void block_max1(float& m, const f32x16& acc, uint32_t ex, uint32_t va, float fill, int h) {
  if (va == 0xffffffffu) {
#pragma unroll
    for (int i = 0; i < 16; i += 2) m = __builtin_fmaxf(__builtin_fmaxf(m, acc[i]), acc[i + 1]);
  } else {
    const uint32_t exs = ex >> (4 * h), vas = va >> (4 * h);
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int bit = rowof(i);
      v[i] = ((vas >> bit) & 1u) ? acc[i] : (((exs >> bit) & 1u) ? -1000.0f : fill);
    }
#pragma unroll
    for (int i = 0; i < 16; i += 2) m = __builtin_fmaxf(__builtin_fmaxf(m, v[i]), v[i + 1]);
  }
}

// The reference's similarity matrix has the dtype of its inputs: under torch.cuda.amp.autocast (colbert.py:60,
// defaults.yaml:21 use_fp16: True) `bmm` returns fp16 — fp32 accumulation, ONE rounding per element — the -1000 fill and
// `max` stay in fp16 and only `sum` is promoted to fp32 (:68-75).  Rounding is monotone, so
// max_j round(s_ij) = round(max_j s_ij): one conversion of the per-token maximum reproduces that arithmetic exactly.
// -1000 is representable in fp16 and bf16.  (bf16: what torch.bmm does on bf16 tensors.)
template <int DT>
__device__ __forceinline__ float round_like(float x) {
  if constexpr (DT == MM_F16) {
    return (float)(_Float16)x;                                 // v_cvt_f16_f32: RNE, overflow -> inf as torch's cast
  } else if constexpr (DT == MM_BF16) {
    const uint32_t u = __float_as_uint(x);
    return __uint_as_float((u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u);   // RNE (no NaNs reach here)
  } else {
    return x;
  }
}

template <int DT>
__device__ __forceinline__ float finish_pair1(float m, bool qvalid, int h, int rnd) {
  m = fmaxf(m, __shfl_xor(m, 32, 64));  // other half holds the other 16 rows of every block
  if (rnd & MM_SIM_ROUND) m = round_like<DT>(m);
  return wave_sum((qvalid && h == 0) ? m : 0.0f);
}

template <int DT>
__device__ __forceinline__ float finish_pair(const float (&m)[16], bool qvalid, int h, int rnd) {
  float mx = m[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) mx = fmaxf(mx, m[i]);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));  // other half holds the other 16 rows of every block
  if (rnd & MM_SIM_ROUND) mx = round_like<DT>(mx);
  return wave_sum((qvalid && h == 0) ? mx : 0.0f);
}

// the pair's score: fp32 (autocast promotes `sum`, colbert.py:75), or — all-fp16 tensors outside autocast, the dynamic
// teacher's all-pairs call, dynamic_teacher.py:245-246 — rounded like the inputs (torch sums 16-bit tensors in fp32 and
// rounds the result once)
template <int DT>
__device__ __forceinline__ float finish_sum(float s, int rnd) {
  return (rnd & MM_SUM_ROUND) ? round_like<DT>(s) : s;
}

// ---------------------------------------------------------------------------------------------
// Roofline path.
// ---------------------------------------------------------------------------------------------
constexpr int kBlkBytes = 32 * 256;  // 32 document tokens x 128 dims x 2 B

// 8 LDS-DMA instructions = one 8 KiB document block.  Instruction k moves rows 4k..4k+3:
// 16 lanes per row, each lane one 16-B chunk.  LDS destination is lane-linear (M0 + lane*16), so
// the bank swizzle is applied on the SOURCE side: the chunk stored at slot p of row r is chunk
// p ^ (r & 15).  A later ds_read_b128 of chunk c of row (lane & 31) then reads slot c ^ (r & 15):
// every 16-lane service group of the read covers 16 distinct slots of the 256-B bank row.
template <bool NT>
__device__ __forceinline__ void issue_block(const char* gbase, const uint32_t (&voff)[8], uint32_t lds_dst) {
  uint32_t keep;
  if (NT) {
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %10\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %9 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %9 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %9 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %9 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %5, %9 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %6, %9 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %7, %9 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %8, %9 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "v"(voff[4]), "v"(voff[5]), "v"(voff[6]),
          "v"(voff[7]), "s"(gbase), "s"(lds_dst)
        : "memory", "scc");
  } else {
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %10\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %9\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %9\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %9\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %9\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %5, %9\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %6, %9\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %7, %9\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %8, %9\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "v"(voff[4]), "v"(voff[5]), "v"(voff[6]),
          "v"(voff[7]), "s"(gbase), "s"(lds_dst)
        : "memory", "scc");
  }
}

// Wait until at most `younger` blocks (8 LDS-DMA each) issued after the one we need are pending.
__device__ __forceinline__ void wait_block(int younger) {
  switch (younger) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
  }
}

}  // namespace mm
