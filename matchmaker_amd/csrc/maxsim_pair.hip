// ColBERT MaxSim in the REFERENCE's batch layout: one query tile per pair (eval.py:108 -> colbert.py:68-75 hands
// ColBERT.forward pair-per-row batches: query_vecs [B,Q,E] replicated per candidate, int64 HF attention masks).
//
// The shared-query kernel (maxsim.hip) reloads the query fragments with plain vector loads + vmcnt(0) whenever the
// query changes — harmless once per 1000 candidates, a drain of the LDS-DMA prefetch once per PAIR here — and needs
// a separate pack_mask_kernel pass over the int64 masks.  This kernel keeps everything in ONE in-order stream:
//
//   per pair:  [mask DMA of the NEXT pair: 3 LDS-DMA instructions]  [query tile: NSL ring slots]  [document blocks]
//
//   * the query tile travels through the same LDS ring as the document blocks (same 8-instruction unit, same
//     source-side swizzle) and is read into the MFMA B-fragment registers with ds_read_b128 when its slot comes up:
//     no vector load, no vmcnt(0);
//   * the int64 masks of the next pair are fetched by LDS-DMA one pair ahead (raw image in LDS), turned into
//     {effective length, validity words, query bits} with v_cmp + ballot when the producer reaches that pair (it
//     needs the block count to issue), and handed to the consumer through an 8-entry LDS ring;
//   * waits stay COUNTED: every in-flight unit carries its instruction count (8, or 8 + 3 when a mask fetch rode in
//     front of it) in a 6-bit FIFO, and the wait for the oldest unit is vmcnt(sum of the younger ones).
#include "mm_internal.h"
#include "maxsim_device.h"

namespace mm {

constexpr int kMaskRaw = 3072;     // raw int64 image of one pair's masks: document [0, 2048) (D <= 256), query [2048, 3072)
constexpr int kMaskEntry = 12;     // dwords per converted entry: len, qbits, 8 validity words, 2 pad
constexpr int kMaskEntries = 8;    // producer runs at most NBUF + 1 pairs ahead of the consumer

// three LDS-DMA instructions: 2 KiB of the document mask row + 1 KiB slot for the query mask row
__device__ __forceinline__ void issue_masks(const char* dm_row, const uint32_t (&voff_d)[2], const char* qm_row, uint32_t voff_q,
                                            uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %6\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %4\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %4\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %3, %5\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff_d[0]), "v"(voff_d[1]), "v"(voff_q), "s"(dm_row), "s"(qm_row), "s"(lds_dst)
      : "memory", "scc");
}

template <int DT, int NBUF, int NSL, bool I64>
__device__ __forceinline__ void maxsim_pair_body(const MaxsimArgs& a, const int block_x) {
  static_assert(NBUF >= 2 && NBUF <= 4, "6-bit x 4 unit FIFO");
  constexpr int RB = NSL * 256;  // bytes per token row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int r = lane & 31, h = lane >> 5;
  const int64_t p0 = (int64_t)block_x * a.pairs_per_wave;
  const int64_t p1 = (p0 + a.pairs_per_wave < a.n_pairs) ? p0 + a.pairs_per_wave : a.n_pairs;
  if (p0 >= p1) return;
  const int D = a.D, Q = a.Q;
  const int nblk_tot = (D + 31) >> 5;
  const int rows_last = D - 32 * (nblk_tot - 1);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  char* mraw = smem + NBUF * kBlkBytes;
  uint32_t* ment = (uint32_t*)(mraw + kMaskRaw);
  const uint32_t lds_mraw = lds0 + NBUF * kBlkBytes;

  // per-lane LDS-DMA source offsets of a full block, of a document's last block and of the query tile (rows past the
  // end are redirected to the last real row: never read past the tensors); chunk p of row r is stored at p ^ (r & 15)
  uint32_t voff[8], voff_tail[8], voff_q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int row = 4 * k + (lane >> 4);
    const int c = (lane & 15) ^ (row & 15);
    const int rowt = row < rows_last ? row : rows_last - 1;
    const int rowq = row < Q ? row : Q - 1;
    voff[k] = (uint32_t)(row * RB + c * 16);
    voff_tail[k] = (uint32_t)(rowt * RB + c * 16);
    voff_q[k] = (uint32_t)(rowq * RB + c * 16);
  }
  uint32_t lo[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) lo[kk] = (uint32_t)(r * 256 + ((((2 * kk) | h) ^ (r & 15)) << 4));
  // mask rows: lane l of instruction n fetches bytes 16 (64 n + l) .. +15, clamped into the row
  uint32_t moff_d[2], moff_q = 0;
  if (I64) {
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int b = 16 * (64 * n + lane);
      moff_d[n] = (uint32_t)(b < D * 8 - 16 ? b : D * 8 - 16);
    }
    moff_q = (uint32_t)(16 * lane < Q * 8 - 16 ? 16 * lane : Q * 8 - 16);
  } else {
    moff_d[0] = moff_d[1] = 0;
  }

  const char* dbase = (const char*)a.d;
  const char* qbase = (const char*)a.q;

  // ---- producer: next (pair, query slice | block, slice) to put in flight --------------------------------------
  int64_t pp = p0;
  bool pnew = true;        // pp has not been opened yet (masks not converted, block count unknown)
  int pphase = 0;          // 0: query-tile slices, 1: document blocks
  int pt = 0, psl = 0, pn = 0;
  int pbuf = 0, cbuf = 0, inflight = 0;
  uint32_t fifo = 0;       // instruction counts of the in-flight units, 6 bits each, oldest lowest
  int total_ops = 0;       // their sum
  int pend_extra = 0;      // mask-fetch instructions issued after the youngest unit (ride on the next unit's count)
  int ops_since_pm = 0;    // loads issued after the mask fetch in flight

  if (I64) {
    issue_masks((const char*)(a.dm64 + pp * D), moff_d, (const char*)(a.qm64 + pp * Q), moff_q, lds_mraw);
  }

  auto open_pair = [&]() {   // producer reaches pair pp: its block count, and (I64) its converted masks
    int len;
    if (I64) {
      wait_vm(ops_since_pm);                                  // the raw masks of pp have landed
      const int64_t* md = (const int64_t*)mraw;
      const int64_t* mq = (const int64_t*)(mraw + 2048);
      unsigned long long b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        b[i] = 0;
        if (64 * i < D) {
          const int j = 64 * i + lane;
          b[i] = __ballot(j < D && md[j < D ? j : D - 1] != 0);
        }
      }
      const uint32_t qb = (uint32_t)__ballot(lane < Q && mq[lane < Q ? lane : Q - 1] != 0);
      len = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (b[i]) len = 64 * i + 64 - __builtin_clzll(b[i]);
      if (lane == 0) {
        uint32_t* e = ment + (int)(pp & (kMaskEntries - 1)) * kMaskEntry;
        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
        ((u32x4*)e)[0] = u32x4{(uint32_t)len, qb, (uint32_t)b[0], (uint32_t)(b[0] >> 32)};
        ((u32x4*)e)[1] = u32x4{(uint32_t)b[1], (uint32_t)(b[1] >> 32), (uint32_t)b[2], (uint32_t)(b[2] >> 32)};
        ((u32x4*)e)[2] = u32x4{(uint32_t)b[3], (uint32_t)(b[3] >> 32), 0u, 0u};
      }
      if (pp + 1 < p1) {   // fetch the next pair's masks (the raw image is free again: converted above)
        issue_masks((const char*)(a.dm64 + (pp + 1) * D), moff_d, (const char*)(a.qm64 + (pp + 1) * Q), moff_q, lds_mraw);
        pend_extra += 3;
        ops_since_pm = 0;
      }
    } else {
      len = a.dm.len ? (int)sload_u32(a.dm.len, pp) : D;
      len = len < 0 ? 0 : (len > D ? D : len);
    }
    pn = (len + 31) >> 5;
    pnew = false;
  };

  auto top_up = [&]() {
    while (pp < p1 && inflight < NBUF) {
      if (pnew) open_pair();
      const uint32_t dst = lds0 + (uint32_t)pbuf * kBlkBytes;
      if (pphase == 0) {
        issue_block<true>(qbase + pp * Q * (int64_t)RB + psl * 256, voff_q, dst);
      } else {
        const char* g = dbase + (pp * D + (int64_t)pt * 32) * RB + psl * 256;
        if (pt == nblk_tot - 1 && rows_last != 32)
          issue_block<true>(g, voff_tail, dst);
        else
          issue_block<true>(g, voff, dst);
      }
      const int cnt = 8 + pend_extra;
      pend_extra = 0;
      fifo |= (uint32_t)cnt << (6 * inflight);
      total_ops += cnt;
      ops_since_pm += 8;
      pbuf = (pbuf + 1 == NBUF) ? 0 : pbuf + 1;
      ++inflight;
      if (NSL > 1 && ++psl < NSL) continue;
      psl = 0;
      bool next_pair = false;
      if (pphase == 0) {
        pphase = 1;
        pt = 0;
        next_pair = pn == 0;
      } else {
        next_pair = ++pt == pn;
      }
      if (next_pair) {
        ++pp;
        pphase = 0;
        pnew = true;
      }
    }
  };
  // the oldest unit has landed / is consumed
  auto wait_oldest = [&]() { wait_vm(total_ops - (int)(fifo & 63u) + pend_extra); };
  auto pop = [&]() {
    total_ops -= (int)(fifo & 63u);
    fifo >>= 6;
    cbuf = (cbuf + 1 == NBUF) ? 0 : cbuf + 1;
    --inflight;
  };
  top_up();

  short8 qf[NSL][8];
  for (int64_t pair = p0; pair < p1; ++pair) {
    // ---- this pair's masks ----------------------------------------------------------------------------------
    int len;
    uint32_t qbits, dw[8];
    if (I64) {
      typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
      const u32x4* e = (const u32x4*)(ment + (int)(pair & (kMaskEntries - 1)) * kMaskEntry);
      const u32x4 e0 = e[0], e1 = e[1], e2 = e[2];
      len = __builtin_amdgcn_readfirstlane((int)e0[0]);
      qbits = __builtin_amdgcn_readfirstlane(e0[1]);
      dw[0] = __builtin_amdgcn_readfirstlane(e0[2]); dw[1] = __builtin_amdgcn_readfirstlane(e0[3]);
      dw[2] = __builtin_amdgcn_readfirstlane(e1[0]); dw[3] = __builtin_amdgcn_readfirstlane(e1[1]);
      dw[4] = __builtin_amdgcn_readfirstlane(e1[2]); dw[5] = __builtin_amdgcn_readfirstlane(e1[3]);
      dw[6] = __builtin_amdgcn_readfirstlane(e2[0]); dw[7] = __builtin_amdgcn_readfirstlane(e2[1]);
    } else {
      len = a.dm.len ? (int)sload_u32(a.dm.len, pair) : D;
      len = len < 0 ? 0 : (len > D ? D : len);
      int qlen = a.qm.len ? (int)sload_u32(a.qm.len, pair) : Q;
      qlen = qlen < 0 ? 0 : (qlen > 32 ? 32 : qlen);
      qbits = qlen >= 32 ? 0xffffffffu : ((1u << qlen) - 1u);
      if (a.qm.bits) qbits &= sload_u32(a.qm.bits, pair);
#pragma unroll
      for (int i = 0; i < 8; ++i) dw[i] = 0xffffffffu;
    }
    const bool qvalid = r < Q && ((qbits >> r) & 1u);
    // ---- query tile out of the ring ----------------------------------------------------------------------------
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
      top_up();
      wait_oldest();
      const char* buf = smem + cbuf * kBlkBytes;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) qf[sl][kk] = *(const short8*)(buf + lo[kk]);
      pop();
    }
    const int nb = (len + 31) >> 5;
    const float fill = len < D ? -1000.0f : neg_inf();
    float m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = fill;
    for (int t = 0; t < nb; ++t) {
      f32x16 acc = f32x16{0};
#pragma unroll
      for (int sl = 0; sl < NSL; ++sl) {
        top_up();
        wait_oldest();
        const char* buf = smem + cbuf * kBlkBytes;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const short8 av = *(const short8*)(buf + lo[kk]);
          acc = Mfma32x16<DT>::run(av, qf[sl][kk], acc);
        }
        pop();
      }
      const int rem = len - 32 * t;
      const uint32_t ex = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
      uint32_t va = ex;
      if (I64) {
        uint32_t w = dw[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) w = t == i ? dw[i] : w;
        va = w & ex;
      } else if (a.dm.bits) {
        va = sload_u32(a.dm.bits, pair * nblk_tot + t) & ex;
      }
      block_max(m, acc, ex, va, fill, h);
    }
    const float s = finish_pair<DT>(m, qvalid, h, a.rnd);
    if (lane == 0) a.out[pair] = finish_sum<DT>(s, a.rnd);
  }
}

template <int DT, int NBUF, int NSL, bool I64>
__global__ void __launch_bounds__(64) maxsim_pair_kernel(const MaxsimArgs a) {
  maxsim_pair_body<DT, NBUF, NSL, I64>(a, (int)blockIdx.x);
}

// Several pair-per-row batches of ONE shape in one launch (mm_maxsim_fwd_batched): blockIdx.y picks the batch's tensors out of
// the kernel argument itself (no descriptor table in memory, no copy); everything else is maxsim_pair_kernel — the scores of a
// batch are bit-equal to its own mm_maxsim_fwd call (same pairs per wavefront order inside a pair: a pair never spans wavefronts).
// Why: eval.py's 512-pair calls (defaults.yaml:115) at dim 128 are 3.6 us of HBM time behind a ~9 us launch chain; a caller
// that holds several tokenised batches (rerank.evaluate_batches(score_group=...)) pays that chain once per group.
constexpr int kMaxBatches = MM_MAXSIM_MAX_BATCHES;
struct MaxsimBatches {
  const void* q[kMaxBatches];
  const void* d[kMaxBatches];
  const int64_t* qm64[kMaxBatches];
  const int64_t* dm64[kMaxBatches];
  float* out[kMaxBatches];
  int64_t n_pairs[kMaxBatches];
};

template <int DT, int NBUF, int NSL, bool I64>
__global__ void __launch_bounds__(64) maxsim_pair_batched_kernel(const MaxsimArgs a_in, const MaxsimBatches bt) {
  MaxsimArgs a = a_in;
  const int y = blockIdx.y;
  a.q = bt.q[y];
  a.d = bt.d[y];
  a.qm64 = bt.qm64[y];
  a.dm64 = bt.dm64[y];
  a.out = bt.out[y];
  a.n_pairs = bt.n_pairs[y];
  maxsim_pair_body<DT, NBUF, NSL, I64>(a, (int)blockIdx.x);
}

template <int DT, int NSL, bool I64>
static int launch_pair_batched(const MaxsimArgs& a0, const MaxsimBatches& bt, int nb, int64_t total, int64_t max_pairs, hipStream_t stream) {
  MaxsimArgs a = a0;
  const int nbuf = env().maxsim_nbuf >= 3 ? 3 : 2;
  const int lds = nbuf * kBlkBytes + (I64 ? kMaskRaw + kMaskEntries * kMaskEntry * 4 : 0);
  int wpc = env().maxsim_wpc > 0 ? env().maxsim_wpc : 4;
  if (wpc > 8) wpc = 8;
  int64_t waves = (int64_t)kCUs * wpc;                   // over ALL batches
  if (waves > total) waves = total;
  a.pairs_per_wave = (total + waves - 1) / waves;
  const int64_t gx = (max_pairs + a.pairs_per_wave - 1) / a.pairs_per_wave;
  const dim3 grid((unsigned)gx, (unsigned)nb);
  if (nbuf == 3)
    hipLaunchKernelGGL((maxsim_pair_batched_kernel<DT, 3, NSL, I64>), grid, dim3(64), lds, stream, a, bt);
  else
    hipLaunchKernelGGL((maxsim_pair_batched_kernel<DT, 2, NSL, I64>), grid, dim3(64), lds, stream, a, bt);
  return check_launch("maxsim_pair_batched_kernel");
}

template <int DT, bool I64>
static int launch_pair_batched_e(const MaxsimArgs& a, const MaxsimBatches& bt, int nb, int64_t total, int64_t max_pairs, hipStream_t stream) {
  switch (a.E / 128) {
    case 1: return launch_pair_batched<DT, 1, I64>(a, bt, nb, total, max_pairs, stream);
    case 2: return launch_pair_batched<DT, 2, I64>(a, bt, nb, total, max_pairs, stream);
    case 3: return launch_pair_batched<DT, 3, I64>(a, bt, nb, total, max_pairs, stream);
    case 4: return launch_pair_batched<DT, 4, I64>(a, bt, nb, total, max_pairs, stream);
    default: return launch_pair_batched<DT, 6, I64>(a, bt, nb, total, max_pairs, stream);
  }
}

template <int DT, int NSL, bool I64>
static int launch_pair(const MaxsimArgs& a0, hipStream_t stream) {
  MaxsimArgs a = a0;
  const int nbuf = env().maxsim_nbuf >= 3 ? 3 : 2;
  const int lds = nbuf * kBlkBytes + (I64 ? kMaskRaw + kMaskEntries * kMaskEntry * 4 : 0);
  int wpc = env().maxsim_wpc > 0 ? env().maxsim_wpc : 4;
  if (wpc > 8) wpc = 8;
  int64_t waves = (int64_t)kCUs * wpc;
  if (waves > a.n_pairs) waves = a.n_pairs;
  a.pairs_per_wave = (a.n_pairs + waves - 1) / waves;
  waves = (a.n_pairs + a.pairs_per_wave - 1) / a.pairs_per_wave;
  if (nbuf == 3)
    hipLaunchKernelGGL((maxsim_pair_kernel<DT, 3, NSL, I64>), dim3((unsigned)waves), dim3(64), lds, stream, a);
  else
    hipLaunchKernelGGL((maxsim_pair_kernel<DT, 2, NSL, I64>), dim3((unsigned)waves), dim3(64), lds, stream, a);
  return check_launch("maxsim_pair_kernel");
}

template <int DT, bool I64>
static int launch_pair_e(const MaxsimArgs& a, hipStream_t stream) {
  switch (a.E / 128) {
    case 1: return launch_pair<DT, 1, I64>(a, stream);
    case 2: return launch_pair<DT, 2, I64>(a, stream);
    case 3: return launch_pair<DT, 3, I64>(a, stream);
    case 4: return launch_pair<DT, 4, I64>(a, stream);
    default: return launch_pair<DT, 6, I64>(a, stream);
  }
}

bool maxsim_pair_supported(int Q, int E, int dtype) {
  return dtype != MM_F32 && Q <= 32 && (E == 128 || E == 256 || E == 384 || E == 512 || E == 768);
}

// int64 masks straight from the tokenizer: rows must be 16-byte multiples (even Q / D) for the LDS-DMA fetch
bool maxsim_pair_i64_supported(int Q, int D, const void* qm, const void* dm) {
  return D <= 256 && D >= 2 && Q >= 2 && !(D & 1) && !(Q & 1) && !(((uintptr_t)qm | (uintptr_t)dm) & 15);
}

int maxsim_pair_launch(const MaxsimArgs& a, int dtype, bool i64, hipStream_t stream) {
  if (dtype == MM_BF16) return i64 ? launch_pair_e<MM_BF16, true>(a, stream) : launch_pair_e<MM_BF16, false>(a, stream);
  return i64 ? launch_pair_e<MM_F16, true>(a, stream) : launch_pair_e<MM_F16, false>(a, stream);
}

}  // namespace mm

using namespace mm;

extern "C" int mm_maxsim_fwd_batched(const mm_maxsim_batch_t* batches, int n_batches, int q_mask_kind, int d_mask_kind, int Q, int D,
                                     int E, int dtype, int flags, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!batches || n_batches < 1) return set_error(MM_EINVAL, "maxsim_batched: no batches");
  if (n_batches > kMaxBatches) return set_error(MM_EUNSUPPORTED, "maxsim_batched: %d batches (at most %d per launch)", n_batches, kMaxBatches);
  if (Q <= 0 || D <= 0 || E <= 0) return set_error(MM_EINVAL, "maxsim_batched: bad shape");
  if (!maxsim_pair_supported(Q, E, dtype))
    return set_error(MM_EUNSUPPORTED, "maxsim_batched: Q=%d E=%d dtype=%d is not the pair-per-row kernel's (16-bit vectors, Q <= 32, "
                                      "E in {128, 256, 384, 512, 768}): call mm_maxsim_fwd per batch", Q, E, dtype);
  const bool i64 = q_mask_kind == MM_MASK_I64 && d_mask_kind == MM_MASK_I64;
  if (!i64 && !(q_mask_kind == MM_MASK_NONE && d_mask_kind == MM_MASK_NONE))
    return set_error(MM_EUNSUPPORTED, "maxsim_batched: masks are either both int64 tokenizer masks or both absent");
  MaxsimArgs a{};
  a.Q = Q; a.D = D; a.E = E; a.ppq = 1; a.rnd = flags & (MM_SIM_ROUND | MM_SUM_ROUND);
  MaxsimBatches bt{};
  int64_t total = 0, max_pairs = 0;
  for (int i = 0; i < n_batches; ++i) {
    const mm_maxsim_batch_t& b = batches[i];
    if (b.n_pairs < 0 || (b.n_pairs > 0 && (!b.q || !b.d || !b.out))) return set_error(MM_EINVAL, "maxsim_batched: batch %d: null pointer", i);
    if (((uintptr_t)b.q | (uintptr_t)b.d) & 15) return set_error(MM_EINVAL, "maxsim_batched: batch %d: q / d must be 16-byte aligned", i);
    if (i64 && (!b.q_mask || !b.d_mask || !maxsim_pair_i64_supported(Q, D, b.q_mask, b.d_mask)))
      return set_error(MM_EUNSUPPORTED, "maxsim_batched: batch %d: int64 masks need even Q, D <= 256 and 16-byte aligned rows", i);
    bt.q[i] = b.q; bt.d[i] = b.d; bt.qm64[i] = (const int64_t*)b.q_mask; bt.dm64[i] = (const int64_t*)b.d_mask;
    bt.out[i] = b.out; bt.n_pairs[i] = b.n_pairs;
    total += b.n_pairs;
    max_pairs = b.n_pairs > max_pairs ? b.n_pairs : max_pairs;
  }
  if (total == 0) return MM_OK;
  if (max_pairs > 0x7fffffffLL) return set_error(MM_EUNSUPPORTED, "maxsim_batched: too many pairs in one batch");
  if (dtype == MM_BF16)
    return i64 ? launch_pair_batched_e<MM_BF16, true>(a, bt, n_batches, total, max_pairs, stream)
               : launch_pair_batched_e<MM_BF16, false>(a, bt, n_batches, total, max_pairs, stream);
  return i64 ? launch_pair_batched_e<MM_F16, true>(a, bt, n_batches, total, max_pairs, stream)
             : launch_pair_batched_e<MM_F16, false>(a, bt, n_batches, total, max_pairs, stream);
}
