// Brute-force inner-product top-k (dense retrieval) for MI355X (gfx950 / CDNA4).
//
// Replaces, for one GPU's shard of the collection, the faiss flat inner-product index the reference
// searches (matchmaker/retrieval/faiss_indices.py:22-36 `search`, :49-74 FaissIdIndexer =
// IndexIDMap(IndexFlatIP), sharded over all GPUs with useFloat16; called from
// matchmaker/dense_retrieval.py:391 `indexer.search(output, top_n)`); the score is BERT_DOT's dot
// product (matchmaker/models/bert_dot.py:62).  faiss itself (faiss-gpu==1.7.0,
// conda-requirements.txt:1) is a third-party dependency absent from the reference tree: the
// semantics restated here are those of IndexFlatIP.search — for every query the k largest inner
// products, in descending order, with their ids.
//
// Three phases, all enqueued on the caller's stream (no host synchronisation inside):
//   1. dot_stream_kernel<SAMPLE>: scores of every query against a strided SAMPLE of the shard
//      (S = 16384 documents), written densely; sort_rows_kernel sorts each query's sample and takes
//      the m-th largest as the query's threshold tau (m chosen so that ~2k documents of the shard
//      are expected above it).
//   2. dot_stream_kernel<FILTER>: the full Q x C^T product on the bf16/fp16 matrix pipe; the
//      epilogue keeps only scores >= tau[q] (one atomic slot claim per survivor) in a per-query
//      candidate list of `cap` entries.  Nothing of the [nq, N] score matrix ever reaches HBM.
//   3. sort_rows_kernel: exact top-k of each candidate list (score descending, document index
//      ascending on ties), status[q] = 0 | 1 (fewer than k survivors: tau too high) | 2 (list
//      overflowed: tau too low).  The host re-runs queries whose status is not 0 with another m,
//      so the result is always the exact top-k.
//
// dot_stream_kernel: MaxSim's streaming structure turned into a GEMM.  A workgroup of 4 wavefronts
// (one per SIMD, 512 registers each) keeps 4 x 32 x NQT queries as MFMA B fragments in registers for
// its whole life and streams its slice of the collection through an LDS ring shared by the four
// wavefronts (LDS-DMA, 32 documents x E per block, source-side bank swizzle as in maxsim.hip).
// Every block is read once from LDS by each wavefront and multiplied against that wavefront's own
// queries: documents on the MFMA M axis, queries on N, so one lane owns one query per tile and the
// threshold test is a lane-local compare of its 16 accumulator registers.
// Work map: the shard's blocks are split 8 ways by XCD (blockIdx % 8); inside an XCD the
// workgroups are (query group g, sub-slice t): the CUs of one XCD sweep the same documents for
// different query groups at about the same time, so the collection is fetched from HBM once per
// XCD sweep and re-read from that XCD's L2.
#include "mm_internal.h"
#include <type_traits>

namespace mm {

constexpr int kDotSample = 16384;  // sample size (documents) of phase 1 when the shard is larger
constexpr int kSortMax = 16384;    // rows of sort_rows_kernel are padded to a power of two <= this

enum { DOT_SAMPLE = 0, DOT_FILTER = 1 };
constexpr int kStageW = 256;        // LDS staging entries (score, code: 8 bytes) per wavefront
constexpr int64_t kStageRel = (1 << 21) - 1;  // a staged code counts blocks from b_base in 21 bits

struct DotArgs {
  const void* q;      // [nq, E]
  const void* c;      // [N, E] the shard
  int64_t ndocs;      // documents visited by this launch: doc(i) = i * stride, i < ndocs
  int64_t stride;     // 1 = every document, > 1 = strided sample
  int nq, E;
  int G, T;           // query groups per launch, sub-slices per XCD  (grid = 8 * G * T)
  int q_base;         // first query of group 0
  // SAMPLE
  float* all_out;     // [nq, ld_all] scores of the visited documents
  int64_t ld_all;
  // FILTER
  const float* tau;   // [nq]
  int32_t* count;     // [nq] survivors (may exceed cap: overflow)
  float* cand_score;  // [nq, cap]
  int32_t* cand_idx;  // [nq, cap] document index inside the shard
  int cap;
  int flush_every;    // FILTER: blocks between the workgroup's common staging flushes (launch_dot: from `expect`)
  double expect;      // FILTER: expected fraction of a query's scores above its threshold
  unsigned long long* prof;  // optional [grid * 4 wavefronts][8] cycle counters (MM_DOT_PROF=1, tools only)
};

template <int DT>
struct DotMfma;
template <>
struct DotMfma<MM_BF16> {
  static __device__ __forceinline__ f32x16 run(short8 a, short8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <>
struct DotMfma<MM_F16> {
  static __device__ __forceinline__ f32x16 run(short8 a, short8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

__device__ __forceinline__ constexpr int drowof(int i) { return (i & 3) + 8 * (i >> 2); }

// 8 x 16 B of one query row slice (256 B): chunks (2kk + h).  NO wait inside: the prologue requests every slice of both query
// tiles (96 loads per lane at dim 768) and waits once (dot_q_landed) — a wait per slice made the prologue twelve dependent round
// trips, ~20 us per workgroup: 1 % of a filter workgroup's life, a third of a sampling workgroup's (16 blocks).
#define MM_DOT_LOADQ(C)                                                                                 \
  asm volatile(                                                                                         \
      "global_load_dwordx4 %0, %8, off\n\t"                                                             \
      "global_load_dwordx4 %1, %8, off offset:32\n\t"                                                   \
      "global_load_dwordx4 %2, %8, off offset:64\n\t"                                                   \
      "global_load_dwordx4 %3, %8, off offset:96\n\t"                                                   \
      "global_load_dwordx4 %4, %8, off offset:128\n\t"                                                  \
      "global_load_dwordx4 %5, %8, off offset:160\n\t"                                                  \
      "global_load_dwordx4 %6, %8, off offset:192\n\t"                                                  \
      "global_load_dwordx4 %7, %8, off offset:224"                                                       \
      : "=&" C(qf[0]), "=&" C(qf[1]), "=&" C(qf[2]), "=&" C(qf[3]), "=&" C(qf[4]), "=&" C(qf[5]),       \
        "=&" C(qf[6]), "=&" C(qf[7])                                                                    \
      : "v"(base)                                                                                       \
      : "memory")
// the slice's registers are defined (again) HERE: placed behind the one s_waitcnt of the prologue, every use of the fragments
// depends on this statement and so stays behind the wait (volatile statements keep their order; no instruction is emitted)
#define MM_DOT_LANDQ(C)                                                                                 \
  asm volatile("" : "+" C(qf[0]), "+" C(qf[1]), "+" C(qf[2]), "+" C(qf[3]), "+" C(qf[4]), "+" C(qf[5]), "+" C(qf[6]), "+" C(qf[7]))
// AGPR = true: the fragments are loaded straight into accumulator registers and STAY there — the MFMA reads its B operand
// from AGPRs directly.  Left to itself the register allocator treats the fragments that do not fit the 256 VGPRs as
// spills and reloads them with four v_accvgpr_read before every use (192 of the 387 instructions of the K loop).
template <bool AGPR>
__device__ __forceinline__ void dot_load_q(const char* base, short8 (&qf)[8]) {
  if constexpr (AGPR) {
#define MM_C_A(x) "a"(x)
    MM_DOT_LOADQ(MM_C_A);
#undef MM_C_A
  } else {
#define MM_C_V(x) "v"(x)
    MM_DOT_LOADQ(MM_C_V);
#undef MM_C_V
  }
}
template <bool AGPR>
__device__ __forceinline__ void dot_q_landed(short8 (&qf)[8]) {
  if constexpr (AGPR) {
#define MM_C_A(x) "a"(x)
    MM_DOT_LANDQ(MM_C_A);
#undef MM_C_A
  } else {
#define MM_C_V(x) "v"(x)
    MM_DOT_LANDQ(MM_C_V);
#undef MM_C_V
  }
}
#undef MM_DOT_LOADQ
#undef MM_DOT_LANDQ

// NSL LDS-DMA instructions: 4 document rows x 256 B of every 128-dim slice -> 1 KiB of LDS each, the
// slices 8 KiB apart (m0 walks).  The per-slice source offsets come in VGPRs: an instruction offset
// would also be added to the LDS address (LDS_addr = M0 + inst_offset + lane * 16).
#define MM_DOT_LD(N) "s_nop 0\n\tglobal_load_lds_dwordx4 %" #N ", %7\n\ts_add_u32 m0, m0, 0x2000\n\t"
#define MM_DOT_ISSUE(BODY)                                                                              \
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %8\n\t" BODY "s_mov_b32 m0, %0" \
               : "=&s"(keep)                                                                            \
               : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "s"(gbase), "s"(lds_dst) \
               : "memory", "scc")
template <int NSL>
__device__ __forceinline__ void dot_issue(const char* gbase, uint32_t voff, uint32_t lds_dst) {
  static_assert(NSL == 1 || NSL == 2 || NSL == 3 || NSL == 4 || NSL == 6, "E = 128 * NSL");
  uint32_t keep;
  uint32_t v[6];
#pragma unroll
  for (int sl = 0; sl < 6; ++sl) v[sl] = voff + (sl < NSL ? sl : 0) * 256;
  if constexpr (NSL == 1) MM_DOT_ISSUE(MM_DOT_LD(1));
  else if constexpr (NSL == 2) MM_DOT_ISSUE(MM_DOT_LD(1) MM_DOT_LD(2));
  else if constexpr (NSL == 3) MM_DOT_ISSUE(MM_DOT_LD(1) MM_DOT_LD(2) MM_DOT_LD(3));
  else if constexpr (NSL == 4) MM_DOT_ISSUE(MM_DOT_LD(1) MM_DOT_LD(2) MM_DOT_LD(3) MM_DOT_LD(4));
  else MM_DOT_ISSUE(MM_DOT_LD(1) MM_DOT_LD(2) MM_DOT_LD(3) MM_DOT_LD(4) MM_DOT_LD(5) MM_DOT_LD(6));
}
#undef MM_DOT_ISSUE
#undef MM_DOT_LD

// ONE LDS-DMA instruction (4 rows x 256 B of one slice), for the K loop: issued right after the block barrier the 2 NSL
// instructions of a wavefront queue behind those of the three other wavefronts at the CU's one address unit (48 KiB per
// block at 64 B per cycle: ~0.9 k cycles of every wavefront's 5.6 k per block, MM_DOT_PROF); one every second K step
// rides behind the MFMAs instead.  No lgkmcnt wait: the slot written is the one block b - 1 used, every read of which
// was consumed before the barrier.
__device__ __forceinline__ void dot_issue_one(const char* gbase, uint32_t voff, uint32_t lds_dst) {
#if defined(MM_DOT_M0_KEEP)
  // M0 is left holding the destination (declared clobbered): a write to M0 straight behind the request — the restore of the
  // old value — has to wait until the request has read it
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :
               : "v"(voff), "s"(gbase), "s"(lds_dst)
               : "memory", "m0");
#else
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(gbase), "s"(lds_dst)
               : "memory");
#endif
}

template <int N>
__device__ __forceinline__ void dot_wait() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#ifndef MM_DOT_AHEAD
#define MM_DOT_AHEAD 4   // A fragments in flight ahead of the MFMAs that use them (A/B builds: -DMM_DOT_AHEAD=n)
#endif
template <int DT, int NSL, int NQT, int MODE, bool PROF = false>
__global__ void __launch_bounds__(256) dot_stream_kernel(const DotArgs a) {
  // Ring of TWO blocks (measured: 15.20 ms against 15.33 ms with three — block b + 1 has the whole of block b's MFMA
  // loop to land), which leaves LDS for the accumulator parking area of the FILTER epilogue (see below).
  constexpr int NBUF = 2;
  constexpr int RB = NSL * 256;        // bytes per document row
  constexpr int BLK = 32 * RB;         // bytes per 32-document block
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: feeds SGPR operands (m0)
  const int r = lane & 31, h = lane >> 5;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // ---- work map ---------------------------------------------------------------------------------
  const int xcd = blockIdx.x & 7;
  const int j = blockIdx.x >> 3;
  const int g = j % a.G, t = j / a.G;
  const int64_t nblk = (a.ndocs + 31) >> 5;
  const int64_t x_lo = nblk * xcd / 8, x_hi = nblk * (xcd + 1) / 8;
  const int64_t b_lo = x_lo + (x_hi - x_lo) * t / a.T, b_hi = x_lo + (x_hi - x_lo) * (t + 1) / a.T;
  if (b_lo >= b_hi) return;
  const int q0 = a.q_base + g * (128 * NQT) + w * (32 * NQT);
  // Survivors (~5 per wavefront-block at the working threshold) are staged in a wavefront-PRIVATE LDS
  // area: positions come from the ballot of the compare (popcount + mbcnt), the fill level lives in a
  // scalar register — no atomics, no waits.  A wavefront flushes its own area to the per-query
  // candidate lists when it runs full; only that flush uses returning global atomics (which force
  // vmcnt(0), i.e. drain this wavefront's LDS-DMA prefetch: once every ~25 blocks instead of on
  // nearly every block).  An entry is 8 bytes: the score and a CODE (block - b_base) << 11 | element << 6 | lane;
  // query and document are decoded by the flush, 64 entries per instruction, instead of being selected and
  // multiplied per survivor in the filing loop, which one wavefront walks serially (round 6: that loop was
  // 1.2 k of a block's 5.9 k cycles, MM_DOT_PROF).
  uint2* st = (uint2*)(smem + NBUF * BLK) + w * kStageW;   // [4][kStageW] {score bits, code}
  // one slot per lane that takes the store of a lane without a survivor (the in-loop filing round below is branch-free)
  uint2* trash = (uint2*)(smem + NBUF * BLK + 4 * kStageW * 8) + w * 64 + lane;
  // accumulator parking: [element e = 16 n + i][lane] floats = NQT x 4 KiB per wavefront (FILTER epilogue)
  char* park = smem + NBUF * BLK + 4 * kStageW * 8 + 4 * 64 * 8 + w * (NQT * 4096);
  int scnt = 0;  // wave-uniform fill level
  int64_t b_base = b_lo;   // block the staged codes count from (moved by the per-block flush: codes stay below 2^21 blocks)
  auto flush_wave = [&]() {
    for (int i = lane; i < scnt; i += 64) {
      const uint2 ent = st[i];
      const uint32_t code = ent.y;
      const int ls = (int)(code & 63u), e = (int)((code >> 6) & 31u);
      const int qq = q0 + 32 * (e >> 4) + (ls & 31);
      if (qq >= a.nq) continue;   // (a query past the end has tau = +inf: only an infinite score gets here)
      const int doc = (int32_t)(((b_base + (int64_t)(code >> 11)) * 32 + 4 * (ls >> 5) + drowof(e & 15)) * a.stride);
      const int slot = atomicAdd(a.count + qq, 1);
      if ((unsigned)slot < (unsigned)a.cap) {
        a.cand_score[(int64_t)qq * a.cap + slot] = __uint_as_float(ent.x);
        a.cand_idx[(int64_t)qq * a.cap + slot] = doc;
      }
    }
    scnt = 0;
  };

  // ---- this wavefront's queries as MFMA B fragments ----------------------------------------------
  short8 qf[NQT][NSL][8];
  int qid[NQT];
#pragma unroll
  for (int n = 0; n < NQT; ++n) {
    const int qq = q0 + 32 * n + r;
    qid[n] = qq < a.nq ? qq : -1;
    const char* qrow = (const char*)a.q + (int64_t)(qq < a.nq ? qq : a.nq - 1) * RB + h * 16;
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
      if (NSL == 6 && NQT == 2 && n == 1) dot_load_q<true>(qrow + sl * 256, qf[n][sl]);
      else dot_load_q<false>(qrow + sl * 256, qf[n][sl]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every fragment of the query tiles
#pragma unroll
  for (int n = 0; n < NQT; ++n)
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
      if (NSL == 6 && NQT == 2 && n == 1) dot_q_landed<true>(qf[n][sl]);
      else dot_q_landed<false>(qf[n][sl]);
    }
  float tau[NQT];
#pragma unroll
  for (int n = 0; n < NQT; ++n) tau[n] = (MODE == DOT_FILTER && qid[n] >= 0) ? a.tau[qid[n]] : __builtin_huge_valf();
  // The thresholds are in their registers HERE: left to the first use — the in-loop test of the DEEP form — the compiler's wait
  // for this load is an s_waitcnt vmcnt(0) inside the K loop, behind the first LDS-DMA request of every block.
#pragma unroll
  for (int n = 0; n < NQT; ++n) asm volatile("" : "+v"(tau[n]));

  // ---- LDS-DMA: this wavefront moves rows 4k..4k+3 (k = w, w + 4) of every slice of a block --------
  const int64_t rowstep = a.stride * RB;  // bytes between consecutive visited documents
  int lrow[2];
  uint32_t lslot[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k = w + 4 * u;
    lrow[u] = 4 * k + (lane >> 4);
    lslot[u] = (uint32_t)(((lane & 15) ^ (lrow[u] & 15)) * 16);  // source-side swizzle (see maxsim.hip)
  }
  auto issue = [&](int64_t blk, int slot) {
    const int64_t left = a.ndocs - blk * 32;  // documents of this block that exist (>= 1)
    const char* gb = (const char*)a.c + blk * 32 * rowstep;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int row = lrow[u];
      if (left < 32 && row >= left) row = (int)left - 1;  // never read past the shard
      const uint32_t voff = (uint32_t)(row * rowstep) + lslot[u];
      dot_issue<NSL>(gb, voff, lds0 + (uint32_t)(slot * BLK + (w + 4 * u) * 1024));
    }
  };
  // the per-lane source offsets of a whole block (every block but a partial last one), for the in-loop issue
  uint32_t nvo_full[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) nvo_full[u] = (uint32_t)(lrow[u] * rowstep) + lslot[u];
  // A-fragment read offsets inside a slice: chunk (2kk + h) of row r at slot chunk ^ (r & 15) — = lo0 ^ (32 kk): the XOR with
  // 2 kk only reaches bits 5..7 of the byte offset.  One register and one v_xor per read (in the shadow of the MFMAs) instead
  // of eight offsets + eight per-block addresses held through the K loop: the DEEP form needs those registers for its second
  // accumulator set.  (lds0 and the slot bases are multiples of 256: the XOR commutes with adding them.)
  const uint32_t lo0 = (uint32_t)(r * 256 + ((h ^ (r & 15)) << 4));
  auto a_frag = [&](uint32_t base, int s) -> short8 {
    asm("" : "+v"(base));   // a fresh value per read: keeps the compiler from holding the eight XORs in registers
    const uint32_t off = base ^ (uint32_t)(32 * (s & 7));
    return *(const short8*)((const __attribute__((address_space(3))) char*)(uintptr_t)off + (s >> 3) * 8192);
  };

  unsigned long long tp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto now = [&]() -> unsigned long long {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
  };
  int slot_i = 0;  // ring slot of block `b`
  if (b_lo < b_hi) issue(b_lo, 0);
  // FILTER, software-pipelined over the blocks: block b's accumulators are tested and parked after its K loop (pmask_prev: bit
  // e = 16 n + i set <=> element e passed); the FIRST filing round of those survivors (every lane stages its lowest pending
  // element: all there is for 85 % of the blocks) runs as ~20 branch-free instructions in the shadow of block b + 1's MFMAs;
  // what is left after that round is filed by the loop behind the K loop.  (The whole filing used to run between the K loops:
  // a serial chain of ~300 cycles per round in a kernel with one wavefront per SIMD, plus the barrier wait its uneven round
  // counts caused: 1.2 k of a block's 5.9 k cycles, by removal, MM_DOT_CUT.)
  uint32_t pmask_prev = 0;
  // every pending element of the parked block, round after round (wave-uniform flush when a round would not fit)
  auto file_rounds = [&](uint32_t& pm, uint32_t c0) {
    while (true) {
      const bool pend = pm != 0;
      const unsigned long long bal = __builtin_amdgcn_ballot_w64(pend);
      if (bal == 0) break;
      const int cnt = __builtin_popcountll(bal);
      if (scnt + cnt > kStageW) flush_wave();
#ifdef MM_DOT_PROF_ROUNDS   // (counting build: "wait_vm" becomes leftover rounds per block x 1000, "barrier" their entries x 1000)
      if (PROF) { tp[0] += 1000; tp[1] += 1000 * cnt; }
#endif
      if (pend) {
        const int e = (int)__builtin_ctz(pm);
        const float val = *(const float*)(park + (e * 64 + lane) * 4);
        const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, (uint32_t)scnt));
        st[pos] = uint2{__float_as_uint(val), c0 | ((uint32_t)e << 6)};
      }
      scnt += cnt;
      pm &= pm - 1;
    }
  };

  // DEEP (FILTER, 64 queries per wavefront, >= 40 K steps: the dim-640 / 768 instantiations): TWO accumulator sets.  Block b
  // accumulates into one while the other — block b - 1's finished scores — is tested, parked and filed in the shadow of block
  // b's MFMAs: one element pair per K step for the test (4 VALU + one 8-byte-per-lane LDS store), then the two filing rounds.
  // Nothing of the threshold test is left between the K loops (it was ~0.75 k of a block's 5.3 k cycles, a serial stretch
  // that one wavefront per SIMD cannot hide); the last block of the range is tested behind the loop (tail).  The other
  // instantiations keep one set and test behind each K loop.
  // (The MM_DOT_PROF instantiation keeps the one-set form: with the stamps' registers the two-set form spills.)
  constexpr bool DEEP = MODE == DOT_FILTER && NQT == 2 && NSL >= 5 && !PROF;
  constexpr int STEPS = NSL * 8, AHEAD = MM_DOT_AHEAD;
  // in-loop filing rounds: round rd reads its parked values at K step kFileStep + 8 rd and stores four steps later
  constexpr int kFileStep = DEEP ? 17 : 1, kFileRounds = STEPS >= 16 ? 2 : 1;
  static_assert(kFileStep + 8 * (kFileRounds - 1) + 4 < STEPS, "the in-loop filing rounds do not fit the K loop");

  // test + park of a finished accumulator set behind a K loop (every block of the other instantiations; DEEP: the tail)
  auto test_park = [&](const f32x16 (&acc)[NQT], int64_t b) -> uint32_t {
    const int64_t d0 = b * 32 + 4 * h;
    const bool whole = b * 32 + 32 <= a.ndocs;  // only the last block of the shard can be partial
    // pm[n] = 2 pm[n] + (v >= tau): the compare's lane mask enters a per-lane 16-bit mask as the carry — two instructions per
    // element; one chain per query tile, each with its own scalar pair for the carry (through VCC the tiles' chains
    // serialised on the one register)
    uint32_t pm[NQT];
#pragma unroll
    for (int n = 0; n < NQT; ++n) pm[n] = 0;
#pragma unroll
    for (int i = 15; i >= 0; --i) {   // highest element first: its bit is shifted up by the ones after it
#pragma unroll
      for (int n = 0; n < NQT; ++n) *(float*)(park + ((16 * n + i) * 64 + lane) * 4) = acc[n][i];
      if constexpr (NQT == 2) {
        unsigned long long cy0, cy1;   // both compares first: the add of one tile does not wait behind its own compare
        asm("v_cmp_ge_f32_e64 %2, %4, %5\n\tv_cmp_ge_f32_e64 %3, %6, %7\n\t"
            "v_addc_co_u32_e64 %0, %2, %0, %0, %2\n\tv_addc_co_u32_e64 %1, %3, %1, %1, %3"
            : "+v"(pm[0]), "+v"(pm[1]), "=&s"(cy0), "=&s"(cy1)
            : "v"(acc[0][i]), "v"(tau[0]), "v"(acc[1][i]), "v"(tau[1]));
      } else {
        unsigned long long cy;
        asm("v_cmp_ge_f32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, %0, %0, %1" : "+v"(pm[0]), "=&s"(cy) : "v"(acc[0][i]), "v"(tau[0]));
      }
    }
    uint32_t pmask = pm[0];
    if constexpr (NQT == 2) pmask |= pm[1] << 16;
    if (!whole) {   // wave-uniform, the shard's last block only: documents past the end do not exist
      uint32_t exist = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) exist |= (d0 + drowof(i) < a.ndocs) ? (0x00010001u << i) : 0u;
      pmask &= exist;
    }
#if defined(MM_DOT_CUT) && MM_DOT_CUT == 1   // by-removal timing builds (tools/build_variant.sh; results are wrong)
    pmask = 0;
#endif
    return pmask;
  };

  // one block: acc = this block's accumulators; prv (DEEP) = block b - 1's, complete
  auto run_block = [&](const int64_t b, f32x16 (&acc)[NQT], f32x16 (&prv)[NQT]) {
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    if (PROF) t0 = now();
    // This wavefront's part of block b has landed: FILTER waited for it behind the previous K loop, in front of the flush — a
    // wait here would also wait for that flush's stores to be acknowledged (~1 k cycles in every flush block).
    if (MODE != DOT_FILTER || b == b_lo) dot_wait<0>();
    if (PROF) t1 = now();
    __syncthreads();  // every part of block b landed; every wavefront is done with block b - 1
    if (PROF) t2 = now();
    // Block b + 1 goes into the slot block b - 1 used, one LDS-DMA instruction every second K step (dot_issue_one).  The last
    // block of the range requests ITS OWN rows again (valid addresses, an idle slot, one block of ~540): the K loop carries no
    // branch, so it is one scheduling region.  (Until round 6 a run-time switch, MM_DOT_NO_SPREAD, kept the burst form for A/B.)
    const bool more = b + 1 < b_hi;
    const int64_t bn = more ? b + 1 : b;
    const char* gbn = (const char*)a.c + bn * 32 * rowstep;
    uint32_t nvo[2] = {nvo_full[0], nvo_full[1]};
    const uint32_t ndst = lds0 + (uint32_t)((slot_i ^ 1) * BLK + w * 1024);
    {
      const int64_t left = a.ndocs - bn * 32;
      if (left < 32) {   // the shard's last block only: rows past the end are redirected (64-bit multiplies: ~300 cycles)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          int row = lrow[u];
          if (row >= left) row = (int)left - 1;
          nvo[u] = (uint32_t)(row * rowstep) + lslot[u];
        }
      }
    }

    if (PROF) t3 = now();
#pragma unroll
    for (int n = 0; n < NQT; ++n) acc[n] = f32x16{0};
    const uint32_t abase = lds0 + (uint32_t)(slot_i * BLK) + lo0;
    // A fragments are fetched AHEAD steps ahead of the MFMAs that use them (one wavefront per SIMD:
    // nothing else hides the ~100-cycle LDS latency); the group barriers pin the order
    // {1 LDS read, NQT MFMAs} so the compiler does not fold the reads back next to their uses
    bool f_pend = false;
    unsigned long long f_bal = 0;
    uint32_t f_e = 0;
    float f_val = 0.0f;
    uint32_t dpm[2] = {0, 0};   // DEEP: the per-tile pass masks of block b - 1 as they are built
    // the code of block b - 1's entries: blocks since b_base, lane (b_base only moves in the flush below, behind the rounds)
    const uint32_t code_prev = ((uint32_t)(b - 1 - b_base) << 11) | (uint32_t)lane;
    short8 av[AHEAD + 1];
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) av[s] = a_frag(abase, s);
#if defined(MM_DOT_CUT) && MM_DOT_CUT == 10
    av[AHEAD] = av[0];
#endif
    __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);  // the first AHEAD reads go out together
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
#if !defined(MM_DOT_CUT) || MM_DOT_CUT != 10
      if (s + AHEAD < STEPS) av[(s + AHEAD) % (AHEAD + 1)] = a_frag(abase, s + AHEAD);
#endif
#pragma unroll
#if defined(MM_DOT_CUT) && MM_DOT_CUT == 7   // no matrix work: every fourth step keeps the operands alive
      for (int n = 0; n < NQT; ++n) if ((s & 3) == 0) acc[n] = DotMfma<DT>::run(av[s % (AHEAD + 1)], qf[n][s >> 3][s & 7], acc[n]); else asm volatile("" :: "v"(av[s % (AHEAD + 1)]), "v"(qf[n][s >> 3][s & 7]));
#else
      for (int n = 0; n < NQT; ++n) acc[n] = DotMfma<DT>::run(av[s % (AHEAD + 1)], qf[n][s >> 3][s & 7], acc[n]);
#endif
      __builtin_amdgcn_sched_group_barrier(0x008, NQT, 0);  // NQT MFMAs of step s
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // then the read for step s + AHEAD
      if ((s & 1) == 0 && (s >> 1) < 2 * NSL) {
        const int u = (s >> 1) / NSL, sl = (s >> 1) % NSL;
        // (fenced: without the two barriers the scheduler gathers the twelve requests at the head of the loop)
        __builtin_amdgcn_sched_barrier(0);
#if !defined(MM_DOT_CUT) || (MM_DOT_CUT != 8 && MM_DOT_CUT != 11)
        dot_issue_one(gbn, nvo[u] + (uint32_t)(sl * 256), ndst + (uint32_t)(u * 4096 + sl * 0x2000));
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (DEEP) {
        // test + park of block b - 1, element pair i = 15 - s (highest first: its bit is shifted up by the ones after it)
#if !defined(MM_DOT_CUT) || (MM_DOT_CUT != 6 && MM_DOT_CUT != 9)
        if (s < 16) {
          const int i = 15 - s;
          // [e = 16 n + i][lane]: the two tiles' slots are 16 x 256 B apart — one ds_write2st64_b32.  (Tried: the store and the
          // compare fed straight from the accumulator registers through "a" operands, so that the finished set need not be
          // copied out behind the K loop — the allocator answered with 600 v_accvgpr moves and scratch.)
          *(float*)(park + (i * 64 + lane) * 4) = prv[0][i];
          *(float*)(park + ((16 + i) * 64 + lane) * 4) = prv[1][i];
          unsigned long long cy0, cy1;
          asm("v_cmp_ge_f32_e64 %2, %4, %5\n\tv_cmp_ge_f32_e64 %3, %6, %7\n\t"
              "v_addc_co_u32_e64 %0, %2, %0, %0, %2\n\tv_addc_co_u32_e64 %1, %3, %1, %1, %3"
              : "+v"(dpm[0]), "+v"(dpm[1]), "=&s"(cy0), "=&s"(cy1)
              : "v"(prv[0][i]), "v"(tau[0]), "v"(prv[1][i]), "v"(tau[1]));
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
#endif
        if (s == 16) {
          pmask_prev = dpm[0] | (dpm[1] << 16);
          if (b == b_lo) pmask_prev = 0;   // (no block behind the first one: the other set holds zeros)
#if defined(MM_DOT_CUT) && (MM_DOT_CUT == 1 || MM_DOT_CUT == 9 || MM_DOT_CUT == 10 || MM_DOT_CUT == 11)
          pmask_prev = 0;
#endif
        }
      }
      if constexpr (MODE == DOT_FILTER) {
        // filing rounds of block b - 1 (pmask_prev = 0 before the first block: every lane stores to its trash slot).
        // No branch: a lane without a pending element reads element 31's slot and stores to its trash slot.
#pragma unroll
        for (int rd = 0; rd < kFileRounds; ++rd) {
          if (s == kFileStep + 8 * rd) {
            f_pend = pmask_prev != 0;
            f_bal = __builtin_amdgcn_ballot_w64(f_pend);
            f_e = (uint32_t)__builtin_ctz(pmask_prev | 0x80000000u);     // 31 for an empty mask
            f_val = *(const float*)(park + (f_e * 64 + lane) * 4);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // (the parked value's read)
          }
          if (s == kFileStep + 8 * rd + 4) {
            const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(f_bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)f_bal, (uint32_t)scnt));
            uint2* dst = f_pend ? st + pos : trash;
            *dst = uint2{__float_as_uint(f_val), code_prev | (f_e << 6)};
            scnt += __builtin_popcountll(f_bal);
            pmask_prev &= pmask_prev - 1;
            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
        }
      }
    }
    slot_i ^= 1;
    if (PROF) {
      // force the accumulators to be complete before the stamp
      float sink = 0.0f;
#pragma unroll
      for (int n = 0; n < NQT; ++n) sink += acc[n][0];
      asm volatile("" ::"v"(sink));
      t4 = now();
#ifndef MM_DOT_PROF_ROUNDS
      tp[0] += t1 - t0; tp[1] += t2 - t1;
#endif
      tp[2] += t3 - t2; tp[3] += t4 - t3;
    }

    // ---- epilogue: acc[n][i] = <document b*32 + drowof(i) + 4h, query qid[n]> ----------------------
    const int64_t d0 = b * 32 + 4 * h;
    if (MODE == DOT_SAMPLE) {
#pragma unroll
      for (int n = 0; n < NQT; ++n) {
        if (qid[n] < 0) continue;
        float* dst = a.all_out + (int64_t)qid[n] * a.ld_all + d0;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)  // rows 8*g4 .. 8*g4+3 (+4h) are consecutive documents
          if (d0 + 8 * g4 + 3 < a.ndocs)
            *(f32x4*)(dst + 8 * g4) = f32x4{acc[n][4 * g4], acc[n][4 * g4 + 1], acc[n][4 * g4 + 2], acc[n][4 * g4 + 3]};
          else
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (d0 + 8 * g4 + e < a.ndocs) dst[8 * g4 + e] = acc[n][4 * g4 + e];
      }
    } else {
      // Threshold test without a scalar branch per element: every lane collects the elements that pass in a bit mask
      // (bit e = 16 n + i), the accumulators are parked in LDS ([e][lane]), and the survivors (~5 per wavefront-block at
      // the working threshold) are filed from there: every lane takes its lowest pending element per round.
      dot_wait<0>();   // block b + 1's rows (requested in the first half of the K loop) — and nothing else is in flight here
      // What the in-loop rounds left of block b - 1 (a lane with three survivors):
#if !defined(MM_DOT_CUT) || MM_DOT_CUT != 4
      file_rounds(pmask_prev, code_prev);
#endif
      // The flush: on a schedule the four wavefronts of the workgroup share (every a.flush_every blocks, ~96 staged entries each at
      // the expected survivor rate), so that they pay its two returning-atomic round trips (~2.3 k cycles) in the SAME block —
      // flushing whenever the own area was half full put one of the four into a flush in every fifth block and the other three
      // at the barrier (440 cycles per block).  The fill-level test stays as the guard: the in-loop rounds need room for 128.
#if defined(MM_DOT_CUT) && MM_DOT_CUT == 5
      if (scnt > kStageW / 2) {
#else
      if (b - b_base >= a.flush_every || scnt > kStageW / 2) {
#endif
        unsigned long long tf = 0;
        if (PROF) tf = now();
        flush_wave();
        b_base = b;
        if (PROF) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          tp[6] += now() - tf;
        }
      }
      if constexpr (!DEEP) {
        unsigned long long tq = 0;
        if (PROF) tq = now();
        pmask_prev = test_park(acc, b);
        if (PROF) tp[7] += now() - tq;   // park + test (now() waits for the LDS stores)
      }
    }
    if (PROF) tp[4] += now() - t4;
  };

  f32x16 accA[NQT];
  if constexpr (DEEP) {
    f32x16 accB[NQT];
#pragma unroll
    for (int n = 0; n < NQT; ++n) accB[n] = f32x16{0};
    int64_t b = b_lo;
    bool last_in_a;
    while (true) {
      run_block(b, accA, accB);
      if (++b == b_hi) { last_in_a = true; break; }
      run_block(b, accB, accA);
      if (++b == b_hi) { last_in_a = false; break; }
    }
    dot_wait<0>();   // the last block's redundant request has landed before the workgroup's LDS can go to another one
    // tail: the last block of the range has no K loop behind it
    uint32_t pm = last_in_a ? test_park(accA, b_hi - 1) : test_park(accB, b_hi - 1);
    file_rounds(pm, ((uint32_t)(b_hi - 1 - b_base) << 11) | (uint32_t)lane);
    flush_wave();
  } else {
    for (int64_t b = b_lo; b < b_hi; ++b) run_block(b, accA, accA);
    dot_wait<0>();   // the last block's redundant request has landed before the workgroup's LDS can go to another one
    if (MODE == DOT_FILTER) {   // the last block has no K loop behind it
      file_rounds(pmask_prev, ((uint32_t)(b_hi - 1 - b_base) << 11) | (uint32_t)lane);
      flush_wave();
    }
  }
  if (PROF && lane == 0) {
    unsigned long long* o = a.prof + ((int64_t)blockIdx.x * 4 + w) * 8;
    o[0] = tp[0]; o[1] = tp[1]; o[2] = tp[2]; o[3] = tp[3]; o[4] = tp[4]; o[5] = (unsigned long long)(b_hi - b_lo);
    o[6] = tp[6]; o[7] = tp[7];
  }
}

// ---------------------------------------------------------------------------------------------
// Row sort: one workgroup per row; (score descending, index ascending) bitonic network in LDS.
//   SAMPLE rows: keys only, writes tau[row] = m-th largest.
//   CANDIDATE rows: n = min(count[row], cap) valid entries; writes the first k as the result.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool before(float ka, int va, float kb, int vb) {  // a sorts before b
  return ka > kb || (ka == kb && va < vb);
}

template <bool HAS_VAL>
__device__ __forceinline__ void bitonic_desc(float* key, int* val, int n2, int tid, int nthreads) {
  for (int size = 2; size <= n2; size <<= 1) {
    for (int strd = size >> 1; strd > 0; strd >>= 1) {
      __syncthreads();
      for (int idx = tid; idx < (n2 >> 1); idx += nthreads) {
        const int lo = 2 * idx - (idx & (strd - 1));  // element with bit `strd` clear
        const int hi = lo + strd;
        const bool desc = (lo & size) == 0;            // direction of this sub-sequence
        const float ka = key[lo], kb = key[hi];
        const int va = HAS_VAL ? val[lo] : 0, vb = HAS_VAL ? val[hi] : 0;
        const bool a_first = before(ka, va, kb, vb);
        if (a_first != desc) {
          key[lo] = kb; key[hi] = ka;
          if (HAS_VAL) { val[lo] = vb; val[hi] = va; }
        }
      }
    }
  }
  __syncthreads();
}

// m-th largest of a row of n <= 16384 sample scores -> tau[row].  Values stay in registers (16 per
// thread) as order-preserving integers.
__device__ __forceinline__ uint32_t f2ord(float f) {  // larger float <=> larger unsigned
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// The m-th largest of NK keys per thread x 1,024 threads, one byte per pass: a pass counts the keys that match the prefix found
// so far into a 256-bin LDS histogram, wavefront 0 scans it from the top.  Bytes on which all keys agree (kmax / kmin: bounds
// of the key set in unsigned order) are skipped.  hist[256] / misc[8]: LDS scratch; every thread returns the same key.
template <int NK>
__device__ __forceinline__ uint32_t radix_mth_largest(const uint32_t (&key)[NK], int need, uint32_t kmax, uint32_t kmin, int* hist,
                                                      int* misc, int tid, int lane) {
  const int common = kmax == kmin ? 32 : __builtin_clz(kmax ^ kmin);
  uint32_t prefix = common >= 32 ? kmax : (common == 0 ? 0u : (kmax & ~(0xffffffffu >> common)));
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (common >= 32 - shift) {                               // every key has the same byte here
      prefix = (prefix & ~(0xffu << shift)) | (kmax & (0xffu << shift));
      continue;
    }
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const uint32_t hi_mask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
#pragma unroll
    for (int e = 0; e < NK; ++e)
      if ((key[e] & hi_mask) == (prefix & hi_mask)) atomicAdd(&hist[(key[e] >> shift) & 0xffu], 1);
    __syncthreads();
    if (tid < 64) {                                           // suffix scan: lane l owns bins 4l .. 4l + 3
      const int b0 = hist[4 * lane], b1 = hist[4 * lane + 1], b2 = hist[4 * lane + 2], b3 = hist[4 * lane + 3];
      int above = b0 + b1 + b2 + b3;
      int tot = above;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_down(tot, o, 64);
        tot += (lane + o < 64) ? t : 0;
      }
      above = tot - above;                                    // keys in bins of higher lanes
      const int c3 = above + b3, c2 = c3 + b2, c1 = c2 + b1, c0 = c1 + b0;
      if (above < need && c0 >= need) {
        int bin, gt;
        if (c3 >= need) { bin = 4 * lane + 3; gt = above; }
        else if (c2 >= need) { bin = 4 * lane + 2; gt = c3; }
        else if (c1 >= need) { bin = 4 * lane + 1; gt = c2; }
        else { bin = 4 * lane; gt = c1; }
        misc[3] = bin;
        misc[4] = gt;
      }
    }
    __syncthreads();
    prefix = (prefix & ~(0xffu << shift)) | ((uint32_t)misc[3] << shift);
    need -= misc[4];
    __syncthreads();
  }
  return prefix;
}

__global__ void __launch_bounds__(1024) sample_tau_kernel(const float* __restrict__ all, int64_t ld, int n, int m,
                                                          float* __restrict__ tau) {
  // Radix select (round 2 built the answer bit by bit: 32 count + barrier rounds, 36 us per row): the 16 keys of a thread stay
  // in registers.  Round 6: the working point asks for the m = 37th largest of 16,384 — the select first runs over the 1,024
  // per-thread MAXIMA (one key per thread: a sixteenth of the histogram atomics, which all land in the few bins of the score
  // distribution's top bytes).  Its answer T0 is a lower bound of the true m-th largest (m distinct positions are >= T0), the
  // keys >= T0 — m plus the few threads that hold two of them — are gathered into LDS and ranked there: the same threshold, bit for
  // bit, as the full select, which stays as the path for m > 256 and for tie-heavy rows (more than 1,024 keys >= T0).
  __shared__ int hist[256];
  __shared__ int misc[8];
  __shared__ uint32_t cand[1024];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  uint32_t key[16];
  uint32_t kmax = 0u, kmin = 0xffffffffu;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int i = tid + 1024 * e;
    key[e] = i < n ? f2ord(all[(int64_t)row * ld + i]) : 0u;  // 0 sorts below every real value (incl. -inf)
    kmax = key[e] > kmax ? key[e] : kmax;
    kmin = key[e] < kmin ? key[e] : kmin;
  }
  const uint32_t tmax = kmax;                                 // this thread's largest key
  if (tid < 8) misc[tid] = tid == 1 ? -1 : 0;                 // [0] max key, [1] min key (unsigned order), [5] candidates
  __syncthreads();
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const uint32_t a = (uint32_t)__shfl_xor((int)kmax, o, 64), b2 = (uint32_t)__shfl_xor((int)kmin, o, 64);
    kmax = a > kmax ? a : kmax;
    kmin = b2 < kmin ? b2 : kmin;
  }
  if (lane == 0) {
    atomicMax((unsigned int*)&misc[0], kmax);
    atomicMin((unsigned int*)&misc[1], kmin);
  }
  __syncthreads();
  kmax = (uint32_t)misc[0];
  kmin = (uint32_t)misc[1];
  if (m > n) m = n;
  if (m < 1) m = 1;
  if (m <= 256) {
    const uint32_t one[1] = {tmax};
    const uint32_t t0 = radix_mth_largest<1>(one, m, kmax, kmin, hist, misc, tid, lane);   // (kmin <= every maximum <= kmax)
#pragma unroll
    for (int e = 0; e < 16; ++e)
      if (key[e] >= t0) {
        const int pos = atomicAdd(&misc[5], 1);
        if (pos < 1024) cand[pos] = key[e];
      }
    __syncthreads();
    const int cnt = misc[5];
    if (cnt <= 1024) {                                        // (wave-uniform; cnt >= m)
      if (tid < cnt) {
        const uint32_t mine = cand[tid];
        int rank = 0;                                          // keys that sort before mine (ties: lower position first)
        for (int j = 0; j < cnt; ++j) {
          const uint32_t o = cand[j];
          rank += (o > mine || (o == mine && j < tid)) ? 1 : 0;
        }
        if (rank == m - 1) tau[row] = ord2f(mine);
      }
      return;
    }
    __syncthreads();
  }
  const uint32_t prefix = radix_mth_largest<16>(key, m, kmax, kmin, hist, misc, tid, lane);
  if (tid == 0) tau[row] = ord2f(prefix);
}

// (score descending, index ascending) bitonic sort of 1024 entries held ONE PER THREAD: the 45 stages whose partner
// lives in the same wavefront (stride < 64) exchange through lane shuffles, only the 10 strides >= 64 go through LDS
// (two barriers each) — against 55 barrier-separated LDS passes of the in-LDS network.
__device__ __forceinline__ void bitonic1024_desc(float& k, int& v, float* xk, int* xv, int tid) {
  for (int size = 2; size <= 1024; size <<= 1) {
    const bool desc = (tid & size) == 0;
    for (int strd = size >> 1; strd > 0; strd >>= 1) {
      float pk;
      int pv;
      if (strd >= 64) {
        xk[tid] = k;
        xv[tid] = v;
        __syncthreads();
        pk = xk[tid ^ strd];
        pv = xv[tid ^ strd];
        __syncthreads();
      } else {
        pk = __shfl_xor(k, strd, 64);
        pv = __shfl_xor(v, strd, 64);
      }
      const bool me_lo = (tid & strd) == 0;
      const bool lo_first = me_lo ? before(k, v, pk, pv) : before(pk, pv, k, v);
      if (lo_first != desc) {  // the pair is in the wrong order for this sub-sequence: both sides take the partner's entry
        k = pk;
        v = pv;
      }
    }
  }
}

constexpr int kSelMax = 1024;   // entries the selection may keep (k plus the ties of the k-th score)

// Exact top-k of one candidate row.  The ~2.5 k survivors of a row used to be padded to 4,096 and sorted whole (78
// barrier-separated bitonic passes with 2,048 comparators each, 146.8 M LDS bank-conflict cycles per launch in
// profiles/r02_dot_pmc.json: 0.93 ms of the call).  Now: (1) radix SELECT of the k-th largest score — keys as
// order-preserving integers, one 256-bin LDS histogram per byte, bytes on which all keys agree (sign / exponent of
// scores above one threshold) skipped; (2) the entries >= that score (k plus ties, deterministic whatever order the
// filter appended them in) are compacted; (3) those <= 1,024 entries are sorted one per thread (bitonic1024_desc).
// Rows with massive ties (> 1,024 entries at or above the k-th score) or k > 1,024 fall back to the full sort.
__global__ void __launch_bounds__(1024) topk_rows_kernel(const float* __restrict__ cand_score,
                                                         const int32_t* __restrict__ cand_idx,
                                                         const int32_t* __restrict__ count, int cap, int k,
                                                         int64_t n_total, float* __restrict__ out_score,
                                                         int64_t* __restrict__ out_idx, int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* key = (float*)smem;                 // [cap]
  int* val = (int*)(key + cap);              // [cap]
  float* skey = (float*)(val + cap);         // [kSelMax]
  int* sval = (int*)(skey + kSelMax);        // [kSelMax]
  int* hist = sval + kSelMax;                // [256]
  int* misc = hist + 256;                    // [8]
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int cnt = count[row];
  const int n = cnt < cap ? cnt : cap;
  const int64_t want = k < n_total ? k : n_total;  // a shard smaller than k returns everything it has
  for (int i = tid; i < n; i += 1024) {
    key[i] = cand_score[(int64_t)row * cap + i];
    val[i] = cand_idx[(int64_t)row * cap + i];
  }
  if (tid < 8) misc[tid] = tid == 1 ? -1 : 0;             // [0] max key, [1] min key (unsigned order), [2] compaction counter
  __syncthreads();
  float* okey = key;
  int* oval = val;
  int m = n;                                               // entries that are sorted
  bool in_regs = false;
  float rk = -__builtin_huge_valf();
  int rv = 0x7fffffff;
  if (n > kSelMax && k <= kSelMax) {
    // ---- (1) select the k-th largest key --------------------------------------------------------
    uint32_t kmax = 0u, kmin = 0xffffffffu;
    for (int i = tid; i < n; i += 1024) {
      const uint32_t o = f2ord(key[i]);
      kmax = o > kmax ? o : kmax;
      kmin = o < kmin ? o : kmin;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const uint32_t a = (uint32_t)__shfl_xor((int)kmax, o, 64), b2 = (uint32_t)__shfl_xor((int)kmin, o, 64);
      kmax = a > kmax ? a : kmax;
      kmin = b2 < kmin ? b2 : kmin;
    }
    if (lane == 0) {
      atomicMax((unsigned int*)&misc[0], kmax);
      atomicMin((unsigned int*)&misc[1], kmin);
    }
    __syncthreads();
    kmax = (uint32_t)misc[0];
    kmin = (uint32_t)misc[1];
    const int common = kmax == kmin ? 32 : __builtin_clz(kmax ^ kmin);   // leading bits every key shares
    uint32_t prefix = common >= 32 ? kmax : (common == 0 ? 0u : (kmax & ~(0xffffffffu >> common)));
    int need = k;                                          // rank still wanted among the keys matching `prefix`
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (common >= 32 - shift) {                          // every key has the same byte here: nothing to count
        prefix = (prefix & ~(0xffu << shift)) | (kmax & (0xffu << shift));
        continue;
      }
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      const uint32_t hi_mask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
      for (int i = tid; i < n; i += 1024) {
        const uint32_t o = f2ord(key[i]);
        if ((o & hi_mask) == (prefix & hi_mask)) atomicAdd(&hist[(o >> shift) & 0xffu], 1);
      }
      __syncthreads();
      if (tid < 64) {                                      // suffix scan of the 256 bins: lane l owns bins 4l .. 4l + 3
        const int b0 = hist[4 * lane], b1 = hist[4 * lane + 1], b2 = hist[4 * lane + 2], b3 = hist[4 * lane + 3];
        int above = b0 + b1 + b2 + b3;                     // -> keys in bins of HIGHER lanes
        int tot = above;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int t = __shfl_down(tot, o, 64);
          tot += (lane + o < 64) ? t : 0;
        }
        above = tot - above;                               // strictly above this lane's four bins
        // the wanted key lies in the first bin (from the top) at which the running count reaches `need`
        int c3 = above + b3, c2 = c3 + b2, c1 = c2 + b1, c0 = c1 + b0;
        int bin = -1, gt = 0;
        if (above < need && c0 >= need) {
          if (c3 >= need) { bin = 4 * lane + 3; gt = above; }
          else if (c2 >= need) { bin = 4 * lane + 2; gt = c3; }
          else if (c1 >= need) { bin = 4 * lane + 1; gt = c2; }
          else { bin = 4 * lane; gt = c1; }
          misc[3] = bin;
          misc[4] = gt;
        }
      }
      __syncthreads();
      prefix = (prefix & ~(0xffu << shift)) | ((uint32_t)misc[3] << shift);
      need -= misc[4];
      __syncthreads();
    }
    // prefix is the k-th largest key; every entry >= it is kept (k plus ties)
    const uint32_t thr = prefix;
    for (int i = tid; i < n; i += 1024) {
      const float kf = key[i];
      if (f2ord(kf) >= thr) {
        const int pos = atomicAdd(&misc[2], 1);
        if (pos < kSelMax) {
          skey[pos] = kf;
          sval[pos] = val[i];
        }
      }
    }
    __syncthreads();
    m = misc[2];
    if (m <= kSelMax) {
      in_regs = true;
      if (tid < m) {
        rk = skey[tid];
        rv = sval[tid];
      }
      __syncthreads();
      bitonic1024_desc(rk, rv, skey, sval, tid);
    } else {
      m = n;                                               // massive ties: the whole row is sorted below
    }
  }
  if (!in_regs) {
    int n2 = 2;
    while (n2 < n) n2 <<= 1;  // sort only as much as survived
    for (int i = n + tid; i < n2; i += 1024) {
      key[i] = -__builtin_huge_valf();
      val[i] = 0x7fffffff;
    }
    bitonic_desc<true>(key, val, n2, tid, 1024);
  }
  for (int i = tid; i < k; i += 1024) {
    const bool ok = i < m && i < want;
    float ks;
    int vs;
    if (in_regs) {           // k <= 1024: thread i holds rank i
      ks = rk;
      vs = rv;
    } else {
      ks = ok ? okey[i] : 0.0f;
      vs = ok ? oval[i] : 0;
    }
    out_score[(int64_t)row * k + i] = ok ? ks : -__builtin_huge_valf();
    out_idx[(int64_t)row * k + i] = ok ? (int64_t)vs : -1;  // faiss pads missing results with -1
  }
  if (tid == 0) status[row] = cnt > cap ? 2 : (cnt < want ? 1 : 0);
}

__global__ void __launch_bounds__(256) fill_tau_kernel(float* tau, int n, float v) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) tau[i] = v;
}

static size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }
static int pow2_ge(int x) { int p = 1; while (p < x) p <<= 1; return p; }

// Launch geometry of one call: NQT query tiles per wavefront, query groups of 128 * NQT, at most 32
// groups per launch (one per CU of an XCD), T sub-slices per XCD.
struct DotGeom {
  int nqt, qpw, T;
};
static DotGeom dot_geom(int64_t n_docs, int nq) {
  DotGeom g;
  g.nqt = nq > 128 ? 2 : 1;
  g.qpw = 128 * g.nqt;
  int G = (nq + g.qpw - 1) / g.qpw;
  if (G > 32) G = 32;
  // G * T workgroups run on the 32 CUs of an XCD: T = 32 / gcd(G, 32) makes that a whole number of rounds (6,980 queries
  // = 28 groups: T = 1 left 4 of every 32 CUs idle for the whole launch, T = 8 gives 7 full rounds: 15.3 -> 13.6 ms)
  int gcd = G, b32 = 32;
  while (b32) { const int r = gcd % b32; gcd = b32; b32 = r; }
  g.T = 32 / gcd;
  const int64_t per_xcd = ((n_docs + 31) / 32 + 7) / 8;
  while (g.T > 1 && per_xcd / g.T < 16) g.T >>= 1;   // keep >= 16 blocks per workgroup (the query tile load is a prologue)
  if (g.T > per_xcd) g.T = per_xcd > 0 ? (int)per_xcd : 1;
  return g;
}

template <int DT, int NSL, int NQT, int MODE>
static int launch_dot(const DotArgs& a0, int nq_launch, int T, hipStream_t stream) {
  DotArgs a = a0;
  constexpr int QPW = 128 * NQT;  // queries per workgroup
  const int lds = 2 * 32 * NSL * 256 + 4 * kStageW * 8 + 4 * 64 * 8 + 4 * NQT * 4096;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)dot_stream_kernel<DT, NSL, NQT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (31.0 * (double)a.stride * (NSL * 256) >= 4294967296.0)
    return set_error(MM_EUNSUPPORTED, "dot_topk: sample stride too large for 32-bit row offsets");
  a.T = T;
  {  // ~96 staged entries per wavefront between two common flushes (32 NQT queries x 32 documents per wavefront-block)
    const double rate = a.expect * 32.0 * (32 * NQT);
    const double n = rate > 0.0 ? 96.0 / rate : (double)kStageRel;
    a.flush_every = n < 1.0 ? 1 : (n > (double)kStageRel ? (int)kStageRel : (int)n);
  }
  for (int qb = 0; qb < nq_launch; qb += 32 * QPW) {  // at most 32 query groups (one per CU of an XCD) per launch
    const int nq_here = nq_launch - qb < 32 * QPW ? nq_launch - qb : 32 * QPW;
    a.G = (nq_here + QPW - 1) / QPW;
    a.q_base = a0.q_base + qb;
    if constexpr (NSL == 6 && NQT == 2 && MODE == DOT_FILTER) {
      if (a.prof) {  // tools only (MM_DOT_PROF=1): the instantiation that carries the s_memtime stamps
        (void)hipFuncSetAttribute((const void*)dot_stream_kernel<DT, NSL, NQT, MODE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL((dot_stream_kernel<DT, NSL, NQT, MODE, true>), dim3(8 * a.G * a.T), dim3(256), lds, stream, a);
        if (int e = check_launch("dot_stream_kernel<prof>")) return e;
        continue;
      }
    }
    hipLaunchKernelGGL((dot_stream_kernel<DT, NSL, NQT, MODE>), dim3(8 * a.G * a.T), dim3(256), lds, stream, a);
    if (int e = check_launch("dot_stream_kernel")) return e;
  }
  return MM_OK;
}

template <int DT, int MODE>
static int launch_dot_e(const DotArgs& a, int nq_launch, const DotGeom& g, hipStream_t stream) {
  // NQT = 2 (64 queries per wavefront: every LDS read feeds two MFMAs) once there are enough queries
  const bool two = g.nqt == 2;
  switch (a.E / 128) {
    case 1: return two ? launch_dot<DT, 1, 2, MODE>(a, nq_launch, g.T, stream) : launch_dot<DT, 1, 1, MODE>(a, nq_launch, g.T, stream);
    case 2: return two ? launch_dot<DT, 2, 2, MODE>(a, nq_launch, g.T, stream) : launch_dot<DT, 2, 1, MODE>(a, nq_launch, g.T, stream);
    case 3: return two ? launch_dot<DT, 3, 2, MODE>(a, nq_launch, g.T, stream) : launch_dot<DT, 3, 1, MODE>(a, nq_launch, g.T, stream);
    case 4: return two ? launch_dot<DT, 4, 2, MODE>(a, nq_launch, g.T, stream) : launch_dot<DT, 4, 1, MODE>(a, nq_launch, g.T, stream);
    case 6: return two ? launch_dot<DT, 6, 2, MODE>(a, nq_launch, g.T, stream) : launch_dot<DT, 6, 1, MODE>(a, nq_launch, g.T, stream);
    default: return set_error(MM_EUNSUPPORTED, "dot_topk: E=%d (supported: 128, 256, 384, 512, 768; pad the vectors)", a.E);
  }
}

}  // namespace mm

using namespace mm;

static int dot_cap(int64_t n_docs, int k) {  // candidate list capacity per query
  int c = pow2_ge(4 * k);
  if (c < 1024) c = 1024;
  if (n_docs <= 4096 && c < 4096) c = 4096;  // small shards: everything is a candidate
  return c;
}

extern "C" size_t mm_dot_topk_workspace_bytes(int64_t n_docs, int nq, int k) {
  if (n_docs <= 0 || nq <= 0 || k <= 0) return 0;
  const int cap = dot_cap(n_docs, k);
  const int64_t s = n_docs < kDotSample ? n_docs : kDotSample;
  // sample scores | tau | survivor counts | candidate scores + indices
  return a256((size_t)nq * s * 4) + 2 * a256((size_t)nq * 4) + 2 * a256((size_t)nq * cap * 4);
}

extern "C" int mm_dot_topk_fwd(const void* queries, const void* corpus, int64_t n_docs, int nq, int E, int dtype, int k,
                               float m_scale, float* out_scores, int64_t* out_idx, int32_t* status, void* workspace,
                               size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!queries || !corpus || !out_scores || !out_idx || !status) return set_error(MM_EINVAL, "dot_topk: null pointer");
  if (n_docs <= 0 || nq <= 0 || E <= 0 || k <= 0) return set_error(MM_EINVAL, "dot_topk: non-positive shape");
  if (dtype != MM_F16 && dtype != MM_BF16)
    return set_error(MM_EUNSUPPORTED, "dot_topk: float16 / bfloat16 vectors only (the reference's GPU flat index stores fp16)");
  if (n_docs >= (1LL << 31)) return set_error(MM_EUNSUPPORTED, "dot_topk: more than 2^31-1 documents in one shard");
  if (((uintptr_t)queries | (uintptr_t)corpus) & 15) return set_error(MM_EINVAL, "dot_topk: 16-byte alignment required");
  if (k > kSortMax / 4) return set_error(MM_EUNSUPPORTED, "dot_topk: k=%d exceeds the candidate sorter (k <= %d)", k, kSortMax / 4);
  const size_t need = mm_dot_topk_workspace_bytes(n_docs, nq, k);
  if (!workspace || workspace_bytes < need) return set_error(MM_EWORKSPACE, "dot_topk: workspace needs %zu bytes", need);
  if (!(m_scale > 0.0f)) m_scale = 1.0f;

  const DotGeom g = dot_geom(n_docs, nq);
  const bool small = n_docs <= 4096;  // everything is a candidate: no sampling
  const int cap = dot_cap(n_docs, k);
  const int64_t S = n_docs < kDotSample ? n_docs : kDotSample;
  char* ws = (char*)workspace;
  float* all = (float*)ws;            ws += a256((size_t)nq * S * 4);
  float* tau = (float*)ws;            ws += a256((size_t)nq * 4);
  int32_t* count = (int32_t*)ws;      ws += a256((size_t)nq * 4);
  float* cand_score = (float*)ws;     ws += a256((size_t)nq * cap * 4);
  int32_t* cand_idx = (int32_t*)ws;

  DotArgs a{};
  a.q = queries; a.c = corpus; a.nq = nq; a.E = E; a.q_base = 0;
  a.tau = tau; a.count = count; a.cand_score = cand_score; a.cand_idx = cand_idx; a.cap = cap;

  // phase 1: threshold per query
  if (small) {
    hipLaunchKernelGGL(fill_tau_kernel, dim3((nq + 255) / 256), dim3(256), 0, stream, tau, nq, -__builtin_huge_valf());
  } else {
    a.ndocs = S; a.stride = n_docs / S; a.all_out = all; a.ld_all = S;
    const DotGeom gs = dot_geom(S, nq);
    const int e = dtype == MM_BF16 ? launch_dot_e<MM_BF16, DOT_SAMPLE>(a, nq, gs, stream) : launch_dot_e<MM_F16, DOT_SAMPLE>(a, nq, gs, stream);
    if (e) return e;
    // expected survivors of the full pass ~ m * n_docs / S: aim at 2.5 k of the 4 k capacity (x the caller's retry factor)
    double mt = 2.5 * k * m_scale * (double)S / (double)n_docs;
    int m = (int)(mt + 0.5);
    if (m < 4) m = 4;
    if (m > S) m = (int)S;
    hipLaunchKernelGGL(sample_tau_kernel, dim3(nq), dim3(1024), 0, stream, all, S, (int)S, m, tau);
  }
  if (int e = check_launch("dot_topk threshold")) return e;

  // phase 2: full product + threshold filter
  if (hipMemsetAsync(count, 0, (size_t)nq * 4, stream) != hipSuccess) return set_error(MM_ELAUNCH, "dot_topk: memset failed");
  a.ndocs = n_docs; a.stride = 1;
  a.expect = small ? 1.0 : 2.5 * k * m_scale / (double)n_docs;
  if (env().dot_prof) {  // MM_DOT_PROF=1: per-wavefront phase cycle counters (single-GPU profiling runs only:
                         // the buffer lives on the device that was current at the first call)
    static unsigned long long* const prof_buf = [] {
      void* p = nullptr;
      if (hipMalloc(&p, 8 * 32 * 32 * 4 * 8 * 8) != hipSuccess) p = nullptr;
      return (unsigned long long*)p;
    }();
    a.prof = prof_buf;
  }
  {
    const int e = dtype == MM_BF16 ? launch_dot_e<MM_BF16, DOT_FILTER>(a, nq, g, stream) : launch_dot_e<MM_F16, DOT_FILTER>(a, nq, g, stream);
    if (e) return e;
  }
  // phase 3: exact top-k of the survivors
  const size_t lds_rows = (size_t)cap * 8 + kSelMax * 8 + 256 * 4 + 32;
  if (lds_rows > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)topk_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rows);
  hipLaunchKernelGGL(topk_rows_kernel, dim3(nq), dim3(1024), lds_rows, stream, cand_score, cand_idx, count, cap, k,
                     n_docs, out_scores, out_idx, status);
  if (int e = check_launch("topk_rows_kernel")) return e;
  if (a.prof) {  // tools only: synchronous dump of the phase counters
    static unsigned long long host[8 * 32 * 32 * 4 * 8];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(host, a.prof, sizeof(host), hipMemcpyDeviceToHost);
    const int nw = 8 * 32 * 4;  // first launch's workgroups x 4 wavefronts at most
    double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int cntw = 0;
    for (int i = 0; i < nw; ++i) {
      if (host[i * 8 + 5] == 0) continue;
      for (int j = 0; j < 8; ++j)
        if (j != 5) sum[j] += (double)host[i * 8 + j] / (double)host[i * 8 + 5];
      ++cntw;
    }
    if (cntw)
      fprintf(stderr, "[MM_DOT_PROF] cycles per block (avg over %d wavefronts): wait_vm %.0f | barrier %.0f | flush+issue %.0f | mfma loop %.0f | epilogue %.0f (of which: staging flush %.0f, park + test %.0f)\n",
              cntw, sum[0] / cntw, sum[1] / cntw, sum[2] / cntw, sum[3] / cntw, sum[4] / cntw, sum[6] / cntw, sum[7] / cntw);
  }
  return MM_OK;
}

// Merge of per-shard results (the sharded index's final step): rows of n_in (score, id) pairs ->
// the k best per row.  ids < 0 are padding.  n_in <= 16384.
__global__ void __launch_bounds__(1024) merge_rows_kernel(const float* __restrict__ in_score, const int64_t* __restrict__ in_id,
                                                          int n_in, int n2, int k, float* __restrict__ out_score,
                                                          int64_t* __restrict__ out_id) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* key = (float*)smem;
  int* val = (int*)(key + n2);  // position in the input row: ties keep shard order (then ascending position)
  const int row = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < n2; i += 1024) {
    const bool ok = i < n_in && in_id[(int64_t)row * n_in + i] >= 0;
    key[i] = ok ? in_score[(int64_t)row * n_in + i] : -__builtin_huge_valf();
    val[i] = ok ? i : 0x7fffffff;
  }
  bitonic_desc<true>(key, val, n2, tid, 1024);
  for (int i = tid; i < k; i += 1024) {
    const bool ok = i < n2 && val[i] != 0x7fffffff;
    out_score[(int64_t)row * k + i] = ok ? key[i] : -__builtin_huge_valf();
    out_id[(int64_t)row * k + i] = ok ? in_id[(int64_t)row * n_in + val[i]] : -1;
  }
}

extern "C" int mm_topk_merge(const float* in_scores, const int64_t* in_ids, int nq, int n_in, int k, float* out_scores,
                             int64_t* out_ids, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!in_scores || !in_ids || !out_scores || !out_ids) return set_error(MM_EINVAL, "topk_merge: null pointer");
  if (nq <= 0 || n_in <= 0 || k <= 0) return set_error(MM_EINVAL, "topk_merge: non-positive shape");
  const int n2 = pow2_ge(n_in);
  if (n2 > kSortMax) return set_error(MM_EUNSUPPORTED, "topk_merge: %d inputs per row exceed %d", n_in, kSortMax);
  if ((size_t)n2 * 8 > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)merge_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, n2 * 8);
  hipLaunchKernelGGL(merge_rows_kernel, dim3(nq), dim3(1024), (size_t)n2 * 8, stream, in_scores, in_ids, n_in, n2, k, out_scores, out_ids);
  return check_launch("merge_rows_kernel");
}
