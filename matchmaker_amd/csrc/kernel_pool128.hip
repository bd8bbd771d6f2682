// Streaming kernel pooling for embedding widths that are multiples of 64 floats (E = 64*NSL <= 384):
// Conv-KNRM's n-gram vectors (conv_out_dim 128, conv_knrm.py:144-170), IDCM's passage sampler
// ("ck-small" 128, "tk" 384; sigir21_idcm.py:167-186) and any TK / KNRM run on 64n-wide embeddings.
// Same arithmetic as kernel_pool_split_kernel (split-bf16 dot products: x = hi + lo, 4 bf16 MFMAs per
// 16-wide K step; RBF epilogue of kp_device.h), different tiling of the stream:
//
//   slice  = 32 document tokens x 64 floats (16 chunks of 16 B) = 8 KiB = 8 LDS-DMA instructions;
//            a block of 32 tokens is NSL slices, NBUF = 4 slices (32 KiB) in flight per wavefront so that
//            four single-wave workgroups fit a CU's 160 KiB;
//   LDS    row stride 256 B would put the 16 rows of a ds_read_b128 service group on one bank slot, so
//            chunk c of row r is stored at slot c ^ (r & 15) — the swizzle costs nothing because every
//            LDS-DMA lane fetches from its own global address;
//   K step p (0..3) of a slice: lane (r, h) feeds floats 16p + 8h .. +7 of row r (chunks 4p+2h, 4p+2h+1);
//   query  tile as bf16 hi / lo B fragments in VGPRs: 4*NSL steps x 8 registers (192 at E = 384).
//
// E = 512 / 768 (IDCM's default "ck" sampler works on DistilBERT's 768-wide vectors): the query tile alone
// is 96 KiB of bf16 hi + lo, more than one wavefront's registers can keep beside the accumulators.  KS = 2
// wavefronts per workgroup therefore split the K axis: each streams its own half of every document row
// (its own LDS-DMA ring, its half of the query tile in VGPRs), the two partial 32x32 dot tiles and row
// norms meet in LDS once per block (4 KiB each way, two barriers), and the RBF epilogue is shared by
// kernel pairs (wave 0: kernels 0-5, wave 1: 6-10) so that both matrix pipes and both VALUs stay busy.
#include "mm_internal.h"
#include "kp_device.h"

namespace mm {

constexpr int kS128Instr = 8;
constexpr int kS128Bytes = kS128Instr * 1024;
constexpr int kS128Steps = 4;
constexpr int kS128Nbuf = 4;

__device__ __forceinline__ void issue_slice8(const char* gbase, const uint32_t (&v)[kS128Instr], uint32_t lds_dst) {
  uint32_t keep;
#define MM_GLDS(N) "s_nop 0\n\tglobal_load_lds_dwordx4 %" #N ", %9 nt\n\ts_add_u32 m0, m0, 0x400\n\t"
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %10\n\t" MM_GLDS(1) MM_GLDS(2) MM_GLDS(3) MM_GLDS(4)
                   MM_GLDS(5) MM_GLDS(6) MM_GLDS(7) MM_GLDS(8) "s_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "s"(gbase),
                 "s"(lds_dst)
               : "memory", "scc");
#undef MM_GLDS
}

// one dword per lane through the LDS-DMA path into a 256-byte scratch row: a cache-line prefetch that owns no register
__device__ __forceinline__ void touch_line(const char* gbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(gbase), "s"(lds_dst)
               : "memory");
}

// wait until at most `younger` whole slices are still in flight (extra vector loads in the queue — the
// query tile, the gate — only make this wait longer, never shorter: completion is in order)
__device__ __forceinline__ void wait_slices8(int younger) {
  switch (younger) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
  }
}

// LDS map of a workgroup of KS waves: [KS rings of NBUF slices][KS x rdbuf 128 B][exchange, KS == 2 only:
// xacc 2 x 4 KiB | xss 2 x 128 B | xq 2 x 2 x 128 B | xtot 2 x 16 B][KS gate vectors]
constexpr int kXchgBytes = 2 * 4096 + 2 * 128 + 4 * 128 + 32;
__host__ __device__ constexpr int kp128_lds_fixed(int KS, int nbuf = kS128Nbuf) {
  return KS * (nbuf * kS128Bytes + 128) + (KS == 2 ? kXchgBytes : 0);
}

// MaxSim epilogue (MX): running maximum per accumulator register with ColBERT's -1000 sentinel for masked rows
// (colbert.py:68-75), as maxsim.hip's block_max / finish_pair
__device__ __forceinline__ void mx_block(float (&m)[16], const f32x16& acc, uint32_t ex, uint32_t va, float fill, int h) {
  if (va == 0xffffffffu) {
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = fmaxf(m[i], acc[i]);
  } else {
    const uint32_t exs = ex >> (4 * h), vas = va >> (4 * h);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int bit = rowof(i);
      const float v = ((vas >> bit) & 1u) ? acc[i] : (((exs >> bit) & 1u) ? -1000.0f : fill);
      m[i] = fmaxf(m[i], v);
    }
  }
}

// MX: 0 = kernel pooling; 1 = fp32 MaxSim with the two-term split (4 MFMAs per K step); 2 = fp32 MaxSim with the
// three-term split x = hi + lo + c (6 MFMAs: + c.hi and hi.c; operand error 2^-25, i.e. fp32-class scores)
// OCC = 2 (round 5): TWO wavefronts per SIMD, each with a ring of two slices (16.5 KiB of LDS: eight single-wave workgroups per
// CU) and at most 256 registers.  For SHORT documents — IDCM's ck-small sampler: 64-token passages, at most two blocks per
// pair — a lone wavefront per SIMD spends more cycles between the blocks (scalar length / mask lookups with their waits, the
// log pooling's eleven wave reductions, the store) than in them, and nothing else is resident to fill the gaps; a second
// wavefront does.  E <= 128 only (the query fragments of wider rows need the whole register file).
// MW > 1 (round 5): the workgroup holds up to MW INDEPENDENT wavefronts — the query tensors of a multi launch (KpArgs::m_flat
// = 2) — each with its own ring, query tile and scores; they only meet at one s_barrier per block (a rate limiter, see
// KpArgs).
template <int NSL, int K, bool W, int KS, int MX = 0, int OCC = 1, int MW = 1>
__global__ void __launch_bounds__(64 * KS * MW, OCC) kernel_pool_split128_kernel(const KpArgs a_in) {
  static_assert(KS == 1 || KS == 2, "one wave, or two waves splitting the K axis");
  static_assert(OCC == 1 || (OCC == 2 && KS == 1 && NSL <= 2 && !W), "two wavefronts per SIMD: E <= 128, no gate");
  static_assert(MW == 1 || (KS == 1 && !W && !MX), "independent wavefronts per workgroup: the plain multi launch only");
  const int wq = MW == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const KpArgs a = kp_block_args(a_in, wq);
  static_assert(!MX || !W, "the fp32 MaxSim mode has no gate");
  extern __shared__ __attribute__((aligned(16))) char smem_wg[];
  constexpr int NBUF = OCC == 2 ? 2 : kS128Nbuf;
  char* const smem = smem_wg + (MW == 1 ? 0 : wq * kp128_lds_fixed(1, NBUF));      // this wavefront's own LDS region
  const int lane = threadIdx.x & 63;
  const int wv = KS == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int r = lane & 31, h = lane >> 5;
  const int64_t p0 = (int64_t)a.block_x * a.pairs_per_wave;
  const int64_t p1 = (p0 + a.pairs_per_wave < a.n_pairs) ? p0 + a.pairs_per_wave : a.n_pairs;
  if (p0 >= p1) return;
  constexpr int E = 64 * NSL * KS;
  constexpr int RB = E * 4;
  const int D = a.D, Q = a.Q;
  const int nblk_tot = (D + 31) >> 5;
  const int rows_last = D - 32 * (nblk_tot - 1);
  char* ring = smem + wv * (NBUF * kS128Bytes);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
  float* rdbuf = (float*)(smem + KS * (NBUF * kS128Bytes) + wv * 128);
  char* xchg = smem + KS * (NBUF * kS128Bytes + 128);
  float* xacc = (float*)xchg;                      // [2 waves][4 groups][64 lanes][4]
  float* xss = (float*)(xchg + 8192);              // [2 waves][32 rows]
  float* xq = (float*)(xchg + 8192 + 256);         // [2 toggles][2 waves][32 query tokens]
  float* xtot = (float*)(xchg + 8192 + 256 + 512); // [2 toggles][4]
  float* wbuf = (float*)(smem + kp128_lds_fixed(KS, NBUF)) + wv * 32 * nblk_tot;
  int q_toggle = 0, p_toggle = 0;

  // LDS-DMA source offsets: slot s = 64n + lane -> row s >> 4, stored chunk s & 15 <- global chunk (s & 15) ^ (row & 15)
  uint32_t voff[kS128Instr], voff_tail[kS128Instr];
#pragma unroll
  for (int n = 0; n < kS128Instr; ++n) {
    const int s = 64 * n + lane;
    const int row = s >> 4, c = (s & 15) ^ (row & 15);
    const int row_t = row < rows_last ? row : rows_last - 1;  // last block of a document: stay inside it
    voff[n] = (uint32_t)(row * RB + c * 16);
    voff_tail[n] = (uint32_t)(row_t * RB + c * 16);
  }
  // A-fragment read offsets of this lane inside a slice (swizzled)
  uint32_t aoff[kS128Steps][2];
#pragma unroll
  for (int p = 0; p < kS128Steps; ++p)
#pragma unroll
    for (int j = 0; j < 2; ++j) aoff[p][j] = (uint32_t)(r * 256 + (((4 * p + 2 * h + j) ^ (r & 15)) << 4));

  Rbf rbf;
  if constexpr (!MX) load_rbf<K, KS == 1>(a.mu, a.sigma, a.alpha, a.w, rbf);   // (the K-split pair shares the epilogue by kernel pairs: direct form)

  const char* dbase = (const char*)a.d;
  auto doc_len = [&](int64_t p) -> int {
    int len = a.dm.len ? (int)sload_u32(a.dm.len, p) : D;
    return len < 0 ? 0 : (len > D ? D : len);
  };

  int64_t pp = p0;
  int pt = 0, ps = 0, pn = 0;
  while (pp < p1 && (pn = (doc_len(pp) + 31) >> 5) == 0) ++pp;
  int pbuf = 0, cbuf = 0, inflight = 0;
  auto top_up = [&]() {
    while (pp < p1 && inflight < NBUF) {
      const char* g = dbase + (pp * (int64_t)D + (int64_t)pt * 32) * RB + (wv * NSL + ps) * 256;
      if (pt == nblk_tot - 1 && rows_last != 32)
        issue_slice8(g, voff_tail, lds0 + (uint32_t)pbuf * kS128Bytes);
      else
        issue_slice8(g, voff, lds0 + (uint32_t)pbuf * kS128Bytes);
      pbuf = (pbuf + 1 == NBUF) ? 0 : pbuf + 1;
      ++inflight;
      if (++ps == NSL) {
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

  bf16x8 qhi[NSL][kS128Steps], qlo[NSL][kS128Steps];
  bf16x8 qc[MX == 2 ? NSL : 1][kS128Steps];
  float rq = 0.0f;
  bool qvalid = false;
  uint32_t qbits = 0xffffffffu;
  int qn = Q, np = 2;  // effective query length; lanes sharing one query token in the epilogue (kernel_pool.hip, 3.3)
  int rrows = 0, rtk = 0, rsub = 0;
  int64_t cur_q = -1;
  int64_t qi = p0 / a.ppq;
  int64_t q_left = a.ppq - (p0 - qi * a.ppq);

  for (int64_t pair = p0; pair < p1; ++pair) {
    if (a.pair_q) {
      qi = (int64_t)(int)sload_u32(a.pair_q, pair);
    } else {
      if (q_left == 0) {
        ++qi;
        q_left = a.ppq;
      }
      --q_left;
    }
    if (qi != cur_q) {
      cur_q = qi;
      const int qr = r < Q ? r : Q - 1;
      const char* qrow = (const char*)a.q + (qi * Q + qr) * RB + wv * (NSL * 256) + h * 32;
      float ss = 0.0f;
#pragma unroll
      for (int s = 0; s < NSL; ++s) {
#pragma unroll
        for (int p = 0; p < kS128Steps; ++p) {
          const f32x4 x0 = *(const f32x4*)(qrow + s * 256 + p * 64);
          const f32x4 x1 = *(const f32x4*)(qrow + s * 256 + p * 64 + 16);
          if constexpr (MX == 2) split8x3(x0, x1, qhi[s][p], qlo[s][p], qc[s][p]);
          else split8(x0, x1, qhi[s][p], qlo[s][p]);
          ss += sumsq4(x0) + sumsq4(x1);
        }
      }
      ss += __shfl_xor(ss, 32, 64);
      if constexpr (KS == 2) {  // the other wave holds the other half of the row
        if (h == 0) xq[(q_toggle * 2 + wv) * 32 + r] = ss;
        __syncthreads();
        ss += xq[(q_toggle * 2 + (1 - wv)) * 32 + r];
        q_toggle ^= 1;
      }
      rq = 1.0f / (sqrtf(ss) + 1e-13f);
      const int qlen = a.qm.len ? (int)sload_u32(a.qm.len, qi) : Q;
      qvalid = r < Q && r < qlen;
      qbits = a.qm.bits ? sload_u32(a.qm.bits, qi) : 0xffffffffu;
      if (a.qm.bits) qvalid = qvalid && ((qbits >> r) & 1u);
      qn = qlen < Q ? (qlen < 0 ? 0 : qlen) : Q;
      constexpr bool kRedist = KS == 1 && !MX && !(W && NSL == 6);  // (gated E = 384 would spill registers)
      rrows = kRedist ? redist_rows(qn) : 0;
      np = rrows ? (32 + rrows - 1) / rrows : 2;
      rtk = lane / np;
      rsub = lane - rtk * np;
    }
    const int len = doc_len(pair);
    const int nb = (len + 31) >> 5;
    f32x2 pk2[kMaxK / 2];
#pragma unroll
    for (int k = 0; k < kMaxK / 2; ++k) pk2[k] = f32x2{0.0f, 0.0f};
    if (W) {
      const float* gw = a.dw + pair * (int64_t)D;
      for (int j = lane; j < 32 * nb; j += 64) wbuf[j] = j < D ? gate_log2(gw[j]) : -INFINITY;
    }
    // MX: rows outside the effective length are -1000 if the document has padding at all, else they do not exist
    const float fill = len < D ? -1000.0f : neg_inf();
    float mrun[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) mrun[i] = fill;

    for (int t = 0; t < nb; ++t) {
      // the query-tensor wavefronts of a workgroup (MW > 1) enter every block together: same pairs, same lengths, same trip
      // counts in all of them, so the barrier is reached by all or by none
      if constexpr (MW > 1) __builtin_amdgcn_s_barrier();
      f32x16 acc_hh = {0}, acc_lh = {0}, acc_xl = {0};
      f32x2 ss2 = {0.0f, 0.0f};
#pragma unroll
      for (int s = 0; s < NSL; ++s) {
        top_up();
        wait_slices8(inflight - 1);
        const char* buf = ring + cbuf * kS128Bytes;
        f32x4 x[2 * kS128Steps];
#pragma unroll
        for (int p = 0; p < kS128Steps; ++p) {
          x[2 * p] = *(const f32x4*)(buf + aoff[p][0]);
          x[2 * p + 1] = *(const f32x4*)(buf + aoff[p][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 ah, al, ac;
        if constexpr (MX == 2) split8x3(x[0], x[1], ah, al, ac);
        else split8(x[0], x[1], ah, al);
#pragma unroll
        for (int p = 0; p < kS128Steps; ++p) {
          bf16x8 nh = ah, nl = al, nc = ac;
          if (p + 1 < kS128Steps) {
            if constexpr (MX == 2) split8x3(x[2 * p + 2], x[2 * p + 3], nh, nl, nc);
            else split8(x[2 * p + 2], x[2 * p + 3], nh, nl);
          }
          if constexpr (MX == 2) {  // every accumulator is reused only after two other MFMAs
            acc_hh = mfma_bf16(ah, qhi[s][p], acc_hh);
            acc_lh = mfma_bf16(al, qhi[s][p], acc_lh);
            acc_xl = mfma_bf16(ah, qlo[s][p], acc_xl);
            acc_hh = mfma_bf16(ah, qc[s][p], acc_hh);
            acc_lh = mfma_bf16(ac, qhi[s][p], acc_lh);
            acc_xl = mfma_bf16(al, qlo[s][p], acc_xl);
          } else {
            acc_hh = mfma_bf16(ah, qhi[s][p], acc_hh);
            acc_lh = mfma_bf16(al, qhi[s][p], acc_lh);
            acc_xl = mfma_bf16(ah, qlo[s][p], acc_xl);
            acc_xl = mfma_bf16(al, qlo[s][p], acc_xl);
          }
          if constexpr (!MX) {
            const f32x2 a0 = {x[2 * p][0], x[2 * p][1]}, a1 = {x[2 * p][2], x[2 * p][3]};
            const f32x2 b0 = {x[2 * p + 1][0], x[2 * p + 1][1]}, b1 = {x[2 * p + 1][2], x[2 * p + 1][3]};
            ss2 += a0 * a0;
            ss2 += a1 * a1;
            ss2 += b0 * b0;
            ss2 += b1 * b1;
          }
#pragma unroll
          for (int g = 0; g < (MX == 2 ? 6 : 4); ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);  // 5 VALU in its shadow (fp32 MaxSim, round-robin on one box: 4: 1.048-1.050 ms | 5: 1.030-1.038 | 6: 1.046-1.051 | 8: 1.048-1.057 | 11: 1.056-1.059)
          }
          ah = nh;
          al = nl;
          ac = nc;
        }
        cbuf = (cbuf + 1 == NBUF) ? 0 : cbuf + 1;
        --inflight;
      }
      float ss = ss2[0] + ss2[1];
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = acc_hh[i] + (acc_lh[i] + acc_xl[i]);
      if constexpr (MX && KS == 1) {
        const int rem = len - 32 * t;
        const uint32_t ex = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
        const uint32_t va = a.dm.bits ? (sload_u32(a.dm.bits, pair * nblk_tot + t) & ex) : ex;
        mx_block(mrun, acc, ex, va, fill, h);
        continue;
      }
      ss += __shfl_xor(ss, 32, 64);
      if constexpr (KS == 2) {
        // partial dot tile + partial row norms of this wave's K half -> LDS; add the other wave's
        __syncthreads();  // the other wave has finished reading the previous block's exchange
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(f32x4*)(xacc + ((wv * 4 + g) * 64 + lane) * 4) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        if (h == 0) xss[wv * 32 + r] = ss;
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 o = *(const f32x4*)(xacc + (((1 - wv) * 4 + g) * 64 + lane) * 4);
          acc[4 * g] += o[0]; acc[4 * g + 1] += o[1]; acc[4 * g + 2] += o[2]; acc[4 * g + 3] += o[3];
        }
        ss += xss[(1 - wv) * 32 + r];
      }
      if constexpr (MX) {  // KS == 2: both waves now hold the full dot tile; wave 0 keeps the running maximum
        if (wv == 0) {
          const int rem = len - 32 * t;
          const uint32_t ex = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
          const uint32_t va = a.dm.bits ? (sload_u32(a.dm.bits, pair * nblk_tot + t) & ex) : ex;
          mx_block(mrun, acc, ex, va, fill, h);
        }
        continue;
      }
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
      if (KS == 1 && np > 2) {
        // short query: transpose the scaled tile through the ring slot just consumed (free until the next top_up())
        // and let np lanes share each query token (kp_device.h rbf_redistributed)
        float* T = (float*)(ring + (cbuf == 0 ? NBUF - 1 : cbuf - 1) * kS128Bytes);
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(f32x4*)(T + r * kTS + 8 * g + 4 * h) = f32x4{(acc[4 * g] * rq) * rdr[4 * g], (acc[4 * g + 1] * rq) * rdr[4 * g + 1],
                                                       (acc[4 * g + 2] * rq) * rdr[4 * g + 2], (acc[4 * g + 3] * rq) * rdr[4 * g + 3]};
        const float* lwrow = W ? wbuf + 32 * t : nullptr;
        rbf_redistributed_rows<K, W>(rrows, pk2, T, lwrow, rtk, rsub, va, rbf);
      } else if constexpr (W) {
        float lw[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = *(const f32x4*)(wbuf + 32 * t + 8 * g + 4 * h);
          lw[4 * g + 0] = v[0]; lw[4 * g + 1] = v[1]; lw[4 * g + 2] = v[2]; lw[4 * g + 3] = v[3];
        }
        if (KS == 1) rbf_block<K, true>(pk2, acc, rdr, rq, va, h, rbf, lw);
        else if (wv == 0) rbf_block<K, true, 0, 3>(pk2, acc, rdr, rq, va, h, rbf, lw);
        else rbf_block<K, true, 3, (K + 1) / 2>(pk2, acc, rdr, rq, va, h, rbf, lw);
      } else {
        if (KS == 1) rbf_block<K>(pk2, acc, rdr, rq, va, h, rbf);
        else if (wv == 0) rbf_block<K, false, 0, 3>(pk2, acc, rdr, rq, va, h, rbf);
        else rbf_block<K, false, 3, (K + 1) / 2>(pk2, acc, rdr, rq, va, h, rbf);
      }
    }
    if constexpr (MX) {  // max over the 32 rows of every block, then the masked sum over query tokens
      float mx = mrun[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) mx = fmaxf(mx, mrun[i]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float s = wave_sum((qvalid && h == 0) ? mx : 0.0f);
      if (lane == 0 && wv == 0) a.out[pair] = s;
      continue;
    }
    float pk[kMaxK];
    if (KS == 1 && np > 2) {  // np consecutive lanes hold the partial sums of one query token
      pk_get<K>(pk, pk2, rbf);
      redist_reduce<K>(pk, np, lane);
      finish_pool<K>(a, pair, pk, rsub == 0 && rtk < qn && ((qbits >> rtk) & 1u), lane, rbf, rsub == 0 ? rtk : -1);
      continue;
    }
    pk_get<K>(pk, pk2, rbf);
#pragma unroll
    for (int k = 0; k < K; ++k) pk[k] += __shfl_xor(pk[k], 32, 64);
    if constexpr (KS == 1) {
      finish_pool<K>(a, pair, pk, qvalid && lane < 32, lane, rbf, lane < 32 ? lane : -1);
    } else {
      static_assert(K > 6, "wave 1 pools kernels 6..K-1");
      // each wave pools the kernels it evaluated; wave 1 hands its weighted partial to wave 0
      const float part = wv == 0 ? pool_partial<K, 0, 6>(a, pair, pk, qvalid && lane < 32, lane, rbf, lane < 32 ? lane : -1)
                                 : pool_partial<K, 6, K>(a, pair, pk, qvalid && lane < 32, lane, rbf, lane < 32 ? lane : -1);
      if (wv == 1 && lane == 0) xtot[p_toggle * 4] = part;
      __syncthreads();
      if (wv == 0 && lane == 0) a.out[pair] = part + xtot[p_toggle * 4];
      p_toggle ^= 1;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Conv-KNRM's multi launch, one wavefront per (pair range, DOCUMENT tensor) looping over the query tensors (round 6).
//
// The flat XCD-grouped order above gives every (query tensor, document tensor) combination its own wavefront: each block of a
// document n-gram tensor is streamed by n_mq wavefronts that drift apart — FETCH_SIZE 46.9 GB per Conv-KNRM 3 x 3 launch for
// 13.7 GB of needed rows (profiles/r05_experiments/conv_knrm_fetch_size_per_launch_form.txt) — and split to bf16 n_mq times.
// Here a block crosses HBM and the operand split ONCE: the wavefront keeps the n_mq query tiles of its pair as B fragments
// (AGPRs: 3 x 64 registers at E = 128), splits the block's rows once (64 registers of A fragments), and runs products +
// RBF epilogue for the query tensors back to back, n_mq sets of running kernel sums in registers.  Same MFMA order, same
// epilogue as kernel_pool_split128_kernel; the partial rows are summed in (i, t) order by the same kp_sum_blocks_kernel.  The
// scores agree with the per-combination forms' to fp32 rounding (<= 6e-6 on scores of order 1; a different instruction stream,
// different multiply-add contraction), NOT bit for bit as this comment claimed until it was measured
// (tests/test_kernel_pool_gpu.py::test_multi_launch_*).  One wavefront per SIMD (the fragments need the register file).
// MEASURED (64 x 1000 pairs, Q30 / D200 / E128, same box, round-robin; tools/conv_knrm_ab.sh): FETCH_SIZE 22.5 GB per launch
// (13.7 GB of document rows + 8.8 GB of query tiles, each read by the three document-tensor wavefronts of its pair) against
// 47.3 GB.  With the direct RBF form (twelve v_exp_f32 per cosine) it was the slower form all the same, 9.68 against 8.65 ms:
// ONE wavefront per SIMD cannot overlap its epilogue with anything.  With the middle-out recurrence (kp_device.h rbf_geo_one:
// two thirds of the epilogue's issue cycles) it is 8.38 ms, the per-combination form — which sits on its 47 GB of traffic, not
// on the epilogue — stays at 8.66: this form is the default for launches of >= 2,048 pairs (kp128_launch).  Then by its phase
// clocks (-DMM_KP_MULTI_PROF below; profiles/r06_experiments/rbf_recurrence_ab.txt 6): two fifths of a pair's 105 k cycles were
// per-PAIR work — 24 dependent load -> split round trips for the pair's own query tiles, 33 serial wave sums in the pooling —
// now 7 k + 7 k of 78 k: 6.85 ms; with the wavefronts that share query tiles on one XCD 6.65 ms and 16.6 GB fetched = the needed bytes.
#if defined(MM_KP_MULTI_PROF)   // tools only: phase clocks of the loop kernel, printed by the first wavefronts (a -D variant build)
#define MMP_STAMP(slot) do { uint64_t t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); prof[slot] += t_ - tprev; tprev = t_; } while (0)
#else
#define MMP_STAMP(slot) do { } while (0)
#endif
template <int NSL, int K, int NQ>
__global__ void __launch_bounds__(64, 1) kernel_pool_multi128_kernel(const KpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NBUF = kS128Nbuf;
  const int lane = threadIdx.x & 63;
  const int r = lane & 31, h = lane >> 5;
  const int fr = (int)blockIdx.x;
#if defined(MM_KP_MULTI_PROF)
  uint64_t prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev, tstart;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev) :: "memory");
  tstart = tprev;
#endif
  // m_flat = 3 (the default): the n_md wavefronts of a pair range get the ids g, g + 8, g + 16 — same XCD (workgroup b runs on
  // XCD b % 8: observed, used for speed only) — so the pair's query tiles, which all of them read, cross HBM once per range
  // instead of once per document tensor (8.8 -> 2.95 GB of the launch's 22.5 GB fetched).  Otherwise consecutive ids.
  int bx, td;
  if (a.m_flat == 3) {
    const int period = 8 * a.n_md, blk = fr / period, rr = fr - blk * period;
    bx = __builtin_amdgcn_readfirstlane(blk * 8 + (rr & 7));
    td = __builtin_amdgcn_readfirstlane(rr >> 3);                            // this wavefront's document tensor
    if (bx >= a.m_ranges) return;
  } else {
    bx = __builtin_amdgcn_readfirstlane(fr / a.n_md);
    td = __builtin_amdgcn_readfirstlane(fr - bx * a.n_md);
  }
  const int64_t p0 = (int64_t)bx * a.pairs_per_wave;
  const int64_t p1 = (p0 + a.pairs_per_wave < a.n_pairs) ? p0 + a.pairs_per_wave : a.n_pairs;
  if (p0 >= p1) return;
  constexpr int E = 64 * NSL;
  constexpr int RB = E * 4;
  const int D = a.D, Q = a.Q;
  const int nblk_tot = (D + 31) >> 5;
  const int rows_last = D - 32 * (nblk_tot - 1);
  char* ring = smem;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
  float* rdbuf = (float*)(smem + NBUF * kS128Bytes);
  const uint32_t lds_scratch = lds0 + (uint32_t)kp128_lds_fixed(1);     // 256 B behind the fixed map (kp128_launch adds them)

  uint32_t voff[kS128Instr], voff_tail[kS128Instr];
#pragma unroll
  for (int n = 0; n < kS128Instr; ++n) {
    const int s = 64 * n + lane;
    const int row = s >> 4, c = (s & 15) ^ (row & 15);
    const int row_t = row < rows_last ? row : rows_last - 1;
    voff[n] = (uint32_t)(row * RB + c * 16);
    voff_tail[n] = (uint32_t)(row_t * RB + c * 16);
  }
  uint32_t aoff[kS128Steps][2];
#pragma unroll
  for (int p = 0; p < kS128Steps; ++p)
#pragma unroll
    for (int j = 0; j < 2; ++j) aoff[p][j] = (uint32_t)(r * 256 + (((4 * p + 2 * h + j) ^ (r & 15)) << 4));

  Rbf rbf;
  load_rbf<K>(a.mu, a.sigma, a.alpha, nullptr, rbf);
  // bin weights of this wavefront's NQ combinations (i, td): lane k holds weight k, read back with v_readlane per pair (a
  // scalar reload per pair and combination was a memory round trip each)
  float wv[NQ];
#pragma unroll
  for (int iq = 0; iq < NQ; ++iq) wv[iq] = lane < K ? a.w[(iq * a.n_md + td) * K + lane] : 0.0f;

  const char* dbase = (const char*)a.md[td];
  auto doc_len = [&](int64_t p) -> int {
    int len = a.dm.len ? (int)sload_u32(a.dm.len, p) : D;
    return len < 0 ? 0 : (len > D ? D : len);
  };
  int64_t pp = p0;
  int pt = 0, ps = 0, pn = 0;
  while (pp < p1 && (pn = (doc_len(pp) + 31) >> 5) == 0) ++pp;
  int pbuf = 0, cbuf = 0, inflight = 0;
  auto top_up = [&]() {
    while (pp < p1 && inflight < NBUF) {
      const char* g = dbase + (pp * (int64_t)D + (int64_t)pt * 32) * RB + ps * 256;
      if (pt == nblk_tot - 1 && rows_last != 32)
        issue_slice8(g, voff_tail, lds0 + (uint32_t)pbuf * kS128Bytes);
      else
        issue_slice8(g, voff, lds0 + (uint32_t)pbuf * kS128Bytes);
      pbuf = (pbuf + 1 == NBUF) ? 0 : pbuf + 1;
      ++inflight;
      if (++ps == NSL) {
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

  bf16x8 qhi[NQ][NSL][kS128Steps], qlo[NQ][NSL][kS128Steps];
  float rq[NQ];
#pragma unroll
  for (int iq = 0; iq < NQ; ++iq) rq[iq] = 0.0f;
  bool qvalid = false;
  int64_t cur_q = -1;
  int64_t qi = p0 / a.ppq;
  int64_t q_left = a.ppq - (p0 - qi * a.ppq);

  for (int64_t pair = p0; pair < p1; ++pair) {
    MMP_STAMP(7);
    if (q_left == 0) {
      ++qi;
      q_left = a.ppq;
    }
    --q_left;
    if (qi != cur_q) {
      cur_q = qi;
      const int qr = r < Q ? r : Q - 1;
      // The three tiles are fetched tensor by tensor with the NEXT tensor's sixteen loads in flight while this one is split
      // (one load pair -> wait -> split at a time was 24 dependent memory round trips per pair: 21.8 k of a pair's 105 k
      // cycles, MM_KP_MULTI_PROF).  Same values, same order of the norm's additions.
      constexpr int NX = NSL * kS128Steps * 2;
      f32x4 xq[2][NX];
      auto q_issue = [&](int iq, f32x4 (&dst)[NX]) {
        const char* qrow = (const char*)a.mq[iq] + (qi * Q + qr) * RB + h * 32;
#pragma unroll
        for (int s = 0; s < NSL; ++s)
#pragma unroll
          for (int p = 0; p < kS128Steps; ++p) {
            dst[(s * kS128Steps + p) * 2] = *(const f32x4*)(qrow + s * 256 + p * 64);
            dst[(s * kS128Steps + p) * 2 + 1] = *(const f32x4*)(qrow + s * 256 + p * 64 + 16);
          }
      };
      q_issue(0, xq[0]);
#pragma unroll
      for (int iq = 0; iq < NQ; ++iq) {
        if (iq + 1 < NQ) q_issue(iq + 1, xq[(iq + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        float ss = 0.0f;
#pragma unroll
        for (int s = 0; s < NSL; ++s) {
#pragma unroll
          for (int p = 0; p < kS128Steps; ++p) {
            const f32x4 x0 = xq[iq & 1][(s * kS128Steps + p) * 2];
            const f32x4 x1 = xq[iq & 1][(s * kS128Steps + p) * 2 + 1];
            split8(x0, x1, qhi[iq][s][p], qlo[iq][s][p]);
            qhi[iq][s][p] = to_agpr(qhi[iq][s][p]);
            qlo[iq][s][p] = to_agpr(qlo[iq][s][p]);
            ss += sumsq4(x0) + sumsq4(x1);
          }
        }
        ss += __shfl_xor(ss, 32, 64);
        rq[iq] = 1.0f / (sqrtf(ss) + 1e-13f);
      }
      const int qlen = a.qm.len ? (int)sload_u32(a.qm.len, qi) : Q;
      qvalid = r < Q && r < qlen;
      if (a.qm.bits) qvalid = qvalid && ((sload_u32(a.qm.bits, qi) >> r) & 1u);
    }
    // Prefetch of the NEXT pair's query tiles by LDS-DMA touches (one dword per 128-byte line into a scratch row): built,
    // measured, OFF.  Same box, round-robin (tools/scratch/touch_ab.sh): none 6.82 / 6.87 ms, 22.5 GB fetched; at the head of
    // the pair 7.06 / 7.07 ms, 31.2 GB (a pair later the lines have left this XCD's L2: the tiles cross HBM twice); at the pair's
    // last block 7.01 ms, 24.3 GB.  What made the query phase cheap is the double-buffered fetch above, not the prefetch.
#ifndef MM_KP_MULTI_TOUCH
#define MM_KP_MULTI_TOUCH 0     // 0: no prefetch; 1: at the head of the pair; 2: at the pair's last block (A/B builds)
#endif
    auto touch_next = [&]() {
      const uint32_t qbytes = (uint32_t)(Q * RB);
#pragma unroll
      for (int iq = 0; iq < NQ; ++iq) {
        const char* g = (const char*)a.mq[iq] + (qi + 1) * (int64_t)qbytes;
        for (uint32_t off = (uint32_t)lane * 128u; off < qbytes; off += 64u * 128u) touch_line(g, off, lds_scratch);
      }
    };
    const bool next_own = q_left == 0 && pair + 1 < p1;
    if (MM_KP_MULTI_TOUCH == 1 && next_own) {
      touch_next();
    }
    MMP_STAMP(5);      // query tiles (when the pair brought its own)
    const int len = doc_len(pair);
    const int nb = (len + 31) >> 5;
    f32x2 pk2[NQ][kMaxK / 2];
#pragma unroll
    for (int iq = 0; iq < NQ; ++iq)
#pragma unroll
      for (int k = 0; k < kMaxK / 2; ++k) pk2[iq][k] = f32x2{0.0f, 0.0f};

    if (MM_KP_MULTI_TOUCH == 2 && next_own && nb == 0) touch_next();
    for (int t = 0; t < nb; ++t) {
      if (MM_KP_MULTI_TOUCH == 2 && next_own && t == nb - 1) touch_next();
      // the block's rows: LDS -> registers -> bf16 hi / lo A fragments, ONCE for all query tensors; row norms on the way
      bf16x8 ah[NSL][kS128Steps], al[NSL][kS128Steps];
      f32x2 ss2 = {0.0f, 0.0f};
#pragma unroll
      for (int s = 0; s < NSL; ++s) {
        top_up();
        wait_slices8(inflight - 1);
        const char* buf = ring + cbuf * kS128Bytes;
        f32x4 x[2 * kS128Steps];
#pragma unroll
        for (int p = 0; p < kS128Steps; ++p) {
          x[2 * p] = *(const f32x4*)(buf + aoff[p][0]);
          x[2 * p + 1] = *(const f32x4*)(buf + aoff[p][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        MMP_STAMP(0);    // refill + wait for the slice + its LDS reads
        cbuf = (cbuf + 1 == NBUF) ? 0 : cbuf + 1;      // the slice is in registers: its slot goes back to the producer now
        --inflight;
        top_up();
#pragma unroll
        for (int p = 0; p < kS128Steps; ++p) {
          split8(x[2 * p], x[2 * p + 1], ah[s][p], al[s][p]);
          const f32x2 a0 = {x[2 * p][0], x[2 * p][1]}, a1 = {x[2 * p][2], x[2 * p][3]};
          const f32x2 b0 = {x[2 * p + 1][0], x[2 * p + 1][1]}, b1 = {x[2 * p + 1][2], x[2 * p + 1][3]};
          ss2 += a0 * a0;
          ss2 += a1 * a1;
          ss2 += b0 * b0;
          ss2 += b1 * b1;
        }
        MMP_STAMP(1);    // refill + operand split + norms
      }
      // the first query tensor's products go out BEFORE the row norms make their round trip through LDS (they do not need them)
      f32x16 acc_hh = {0}, acc_lh = {0}, acc_xl = {0};
      auto products = [&](int iq) {
#pragma unroll
        for (int s = 0; s < NSL; ++s) {
#pragma unroll
          for (int p = 0; p < kS128Steps; ++p) {
            acc_hh = mfma_bf16(ah[s][p], qhi[iq][s][p], acc_hh);
            acc_lh = mfma_bf16(al[s][p], qhi[iq][s][p], acc_lh);
            acc_xl = mfma_bf16(ah[s][p], qlo[iq][s][p], acc_xl);
            acc_xl = mfma_bf16(al[s][p], qlo[iq][s][p], acc_xl);
          }
        }
      };
      products(0);
      float ss = ss2[0] + ss2[1];
      ss += __shfl_xor(ss, 32, 64);
      if (h == 0) rdbuf[r] = 1.0f / (sqrtf(ss) + 1e-13f);
      const int rem = len - 32 * t;
      const uint32_t ex = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
      const uint32_t va = a.dm.bits ? (sload_u32(a.dm.bits, pair * nblk_tot + t) & ex) : ex;
      MMP_STAMP(2);      // first tensor's products; row norms into LDS, mask word
#if MM_RBF_GEO && defined(MM_KP_MULTI_OVERLAP)
      if (rbf.geo && va == 0xffffffffu) {
        // A/B builds only (-DMM_KP_MULTI_OVERLAP).  Full block, recurrence epilogue: one straight-line stretch in which the
        // products of query tensor iq + 1 (the same MFMAs in the same order: same bits) are spread, NSL per row, between the RBF
        // evaluations of tensor iq, so that the matrix pipe works under the VALU stream of the ONE wavefront this SIMD has
        // (products behind one another are 2.9 k of a block's 17.6 k cycles).  The ISA shows the interleave (an MFMA every
        // ~5 v_exp_f32) — and the launch takes 7.03 ms instead of 6.93: the epilogue is packed-fp32 math, which does not run
        // beside the matrix pipe for free (the guide's "anti-lever beside MFMAs"), and the stretch spills 16 registers.
#pragma unroll
        for (int iq = 0; iq < NQ; ++iq) {
          f32x16 acc;
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[i] = acc_hh[i] + (acc_lh[i] + acc_xl[i]);
          if (iq + 1 < NQ) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_hh[i] = acc_lh[i] = acc_xl[i] = 0.0f;
          }
          f32x4 rv = {0, 0, 0, 0};                 // 1 / |row| of four rows at a time (re-read from LDS per tensor: sixteen held
                                                   // across the stretch were the registers that spilled)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            if ((i & 3) == 0) rv = *(const f32x4*)(rdbuf + 8 * (i >> 2) + 4 * h);
            if (iq + 1 < NQ) {
#pragma unroll
              for (int n = i * NSL; n < (i + 1) * NSL; ++n) {          // 16 NSL products over 16 rows
                const int sn = n / 16, pn = (n % 16) / 4, jn = n % 4;
                if (jn == 0) acc_hh = mfma_bf16(ah[sn][pn], qhi[iq + 1][sn][pn], acc_hh);
                else if (jn == 1) acc_lh = mfma_bf16(al[sn][pn], qhi[iq + 1][sn][pn], acc_lh);
                else if (jn == 2) acc_xl = mfma_bf16(ah[sn][pn], qlo[iq + 1][sn][pn], acc_xl);
                else acc_xl = mfma_bf16(al[sn][pn], qlo[iq + 1][sn][pn], acc_xl);
              }
            }
            rbf_geo_one<false>(pk2[iq], (acc[i] * rq[iq]) * rv[i & 3], 0.0f, rbf);
          }
        }
        MMP_STAMP(4);
        continue;
      }
#endif
      float rdr[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = *(const f32x4*)(rdbuf + 8 * g + 4 * h);
        rdr[4 * g + 0] = v[0]; rdr[4 * g + 1] = v[1]; rdr[4 * g + 2] = v[2]; rdr[4 * g + 3] = v[3];
      }
#pragma unroll
      for (int iq = 0; iq < NQ; ++iq) {
        if (iq > 0) {
#pragma unroll
          for (int i = 0; i < 16; ++i) acc_hh[i] = acc_lh[i] = acc_xl[i] = 0.0f;
          products(iq);
        }
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = acc_hh[i] + (acc_lh[i] + acc_xl[i]);
#if defined(MM_KP_MULTI_PROF)
        asm volatile("" :: "v"(acc));
#endif
        MMP_STAMP(3);    // matrix products of one query tensor
        rbf_block<K>(pk2[iq], acc, rdr, rq[iq], va, h, rbf);
#if defined(MM_KP_MULTI_PROF)
        asm volatile("" :: "v"(pk2[iq][1]), "v"(pk2[iq][5]));
#endif
        MMP_STAMP(4);    // its RBF epilogue
      }
    }
    // log pooling of the NQ combinations TOGETHER (kp_device.h pool_partial's arithmetic, operation for operation): the
    // NQ x K wave sums share their six butterfly steps
    float lg[NQ][K];
    const bool count_lane = qvalid && lane < 32;
#pragma unroll
    for (int iq = 0; iq < NQ; ++iq) {
      float pk[kMaxK];
      pk_get<K>(pk, pk2[iq], rbf);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        pk[k] += __shfl_xor(pk[k], 32, 64);
        const float v = __logf(fmaxf(pk[k] * rbf.alpha[k], a.clamp_min));
        lg[iq][k] = count_lane ? v : 0.0f;
      }
    }
    // (the butterfly's first step would add the zeros of lanes 32 .. 63 — x + 0 = x: skipped; then one exchange over 16 lanes
    // and the DPP row sum of pool_partial: lane 0 holds the same bits)
#pragma unroll
    for (int iq = 0; iq < NQ; ++iq)
#pragma unroll
      for (int k = 0; k < K; ++k) lg[iq][k] += __shfl_xor(lg[iq][k], 16, 64);
#pragma unroll
    for (int iq = 0; iq < NQ; ++iq)
#pragma unroll
      for (int k = 0; k < K; ++k) lg[iq][k] = row_sum_to_lane0(lg[iq][k]);
#pragma unroll
    for (int iq = 0; iq < NQ; ++iq) {
      const int y = iq * a.n_md + td;                                  // combination (i, t): its bin weights, its partial row
      if (a.per_kernel && lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) a.per_kernel[pair * K + k] = lg[iq][k];
      }
      float total = 0.0f;
#pragma unroll
      for (int k = 0; k < K; ++k)
        total += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wv[iq]), k)) * lg[iq][k];
      if (lane == 0) a.out[(int64_t)y * a.n_pairs + pair] = total;
    }
    MMP_STAMP(6);        // log pooling of the pair's three combinations
  }
#if defined(MM_KP_MULTI_PROF)
  if (lane == 0 && (blockIdx.x % 97) == 0 && blockIdx.x < 970)
    printf("[MM_KP_MULTI_PROF] wave %d pairs %d cycles %llu | slice wait+read %llu | split %llu | norms+mask %llu | mfma %llu | rbf %llu | query tiles %llu | pooling %llu | loop %llu\n",
           (int)blockIdx.x, (int)(p1 - p0), (unsigned long long)(tprev - tstart), (unsigned long long)prof[0], (unsigned long long)prof[1],
           (unsigned long long)prof[2], (unsigned long long)prof[3], (unsigned long long)prof[4], (unsigned long long)prof[5],
           (unsigned long long)prof[6], (unsigned long long)prof[7]);
#endif
}

bool kp128_supported(int Q, int D, int E, bool gated) {
  if (Q > 32 || E % 64) return false;
  const int nsl = E / 64;
  const bool ok = nsl == 1 || nsl == 2 || nsl == 3 || nsl == 4 || nsl == 6 || nsl == 8 || nsl == 12;  // instantiated
  return ok && (!gated || D <= 4096);
}

template <int NSL, bool W, int KS>
static int launch128(const KpArgs& a, const dim3 grid, int lds, hipStream_t stream) {
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kernel_pool_split128_kernel<NSL, 11, W, KS>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((kernel_pool_split128_kernel<NSL, 11, W, KS>), grid, dim3(64 * KS), lds, stream, a);
  return check_launch("kernel_pool_split128_kernel");
}

// fp32 MaxSim (ColBERT with use_fp16 = False, colbert.py:68-75) on the same stream: E = 64n <= 384 (one wave) or 512 / 768 (two waves), Q <= 32
bool kp128_maxsim_supported(int Q, int E) {
  const int nsl = E / 64;
  return Q <= 32 && E % 64 == 0 && (nsl == 1 || nsl == 2 || nsl == 3 || nsl == 4 || nsl == 6 || nsl == 8 || nsl == 12);
}

template <int NSL, int KS, int MX, int OCC = 1>
static void launch_mx(const KpArgs& a, const dim3 grid, int lds, hipStream_t stream) {
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kernel_pool_split128_kernel<NSL, 11, false, KS, MX, OCC>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((kernel_pool_split128_kernel<NSL, 11, false, KS, MX, OCC>), grid, dim3(64 * KS), lds, stream, a);
}

int kp128_maxsim_f32(const float* q, const float* d, PackedMask qm, PackedMask dm, float* out, int64_t n_pairs,
                     int64_t pairs_per_query, int Q, int D, int E, hipStream_t stream) {
  KpArgs a{};
  a.q = q; a.d = d; a.qm = qm; a.dm = dm; a.out = out; a.n_pairs = n_pairs; a.ppq = pairs_per_query;
  a.Q = Q; a.D = D; a.E = E; a.d_doc_rows = D; a.clamp_min = 1e-10f;
  const int nsl = E / 64;
  const int ks = nsl > 6 ? 2 : 1;  // 512 / 768: two waves split the K axis (the query tile does not fit one wave)
  // E <= 128: two wavefronts per SIMD (OCC = 2, see the kernel).  One wavefront per SIMD left its 6 (or 4) MFMAs per K step,
  // the three-term operand split and the LDS-DMA waits in ONE dependent stream: 64 x 1000 pairs at Q32 / D180 / dim 128,
  // same box, round-robin: 1.056 / 1.046 ms with one, 0.937 / 0.928 ms with two (0.70 -> 0.79 of the HBM peak;
  // profiles/r05_experiments/maxsim_fp32_occ.txt).  MM_KP128_OCC=1 forces the old form.
  const bool occ2 = nsl <= 2 && env().kp128_occ != 1 && n_pairs >= (int64_t)kCUs * 8 * 4;
  const int lds = occ2 ? kp128_lds_fixed(1, 2) : kp128_lds_fixed(ks);
  int64_t groups = occ2 ? (int64_t)kCUs * 8 : (int64_t)kCUs * 4 / ks;
  if (groups > a.n_pairs) groups = a.n_pairs;
  a.pairs_per_wave = (a.n_pairs + groups - 1) / groups;
  groups = (a.n_pairs + a.pairs_per_wave - 1) / a.pairs_per_wave;
  const dim3 grid((unsigned)groups);
  const bool x3 = env().maxsim_f32_terms == 3;  // MM_MAXSIM_F32_TERMS=2: two-term split (operand error 2^-17) for A/B runs
#define MM_MX(NSL, KS) (x3 ? launch_mx<NSL, KS, 2>(a, grid, lds, stream) : launch_mx<NSL, KS, 1>(a, grid, lds, stream))
  if (occ2) {
    if (nsl == 1) x3 ? launch_mx<1, 1, 2, 2>(a, grid, lds, stream) : launch_mx<1, 1, 1, 2>(a, grid, lds, stream);
    else x3 ? launch_mx<2, 1, 2, 2>(a, grid, lds, stream) : launch_mx<2, 1, 1, 2>(a, grid, lds, stream);
    return check_launch("kernel_pool_split128_kernel<maxsim, two wavefronts per SIMD>");
  }
  switch (nsl) {
    case 1: MM_MX(1, 1); break;
    case 2: MM_MX(2, 1); break;
    case 3: MM_MX(3, 1); break;
    case 4: MM_MX(4, 1); break;
    case 6: MM_MX(6, 1); break;
    case 8: MM_MX(4, 2); break;
    case 12: MM_MX(6, 2); break;
    default: return set_error(MM_EUNSUPPORTED, "maxsim: E=%d has no fp32 streaming kernel", a.E);
  }
#undef MM_MX
  return check_launch("kernel_pool_split128_kernel<maxsim>");
}

int kp128_launch(const KpArgs& a0, hipStream_t stream) {
  KpArgs a = a0;
  const bool gated = a.dw != nullptr;
  const int nsl = a.E / 64;
  const int ks = nsl > 6 ? 2 : 1;
  // two wavefronts per SIMD (OCC = 2, see the kernel): short documents (<= 3 blocks: the per-pair work — length / mask lookups,
  // the log pooling's eleven wave reductions, a query tile when the pair brings its own — is as long as the blocks), E <= 128,
  // enough pairs to give every one of the 2,048 wavefront slots several.  MM_KP128_OCC = 1 / 2 forces either form (A/B runs).
  const bool occ_ok = !gated && nsl <= 2;
  // ... and Conv-KNRM's multi launch at E = 128 (n_md > 0): in the flat XCD-grouped order (KpArgs::m_flat) two of three readers
  // of a block hit the L2, the launch turns from HBM-bound into epilogue-bound, and a second wavefront's MFMAs run under the
  // first one's RBF evaluations
  const bool occ_auto = (a.D <= 96 || a.n_md > 0) && a.n_pairs >= (int64_t)kCUs * 8 * 4;
  const bool occ2 = occ_ok && (env().kp128_occ == 2 || (env().kp128_occ == 0 && occ_auto));
  const int per_cu = occ2 ? 8 : 4 / ks;
  auto flat_grid = [&](int64_t groups) {      // multi launch: n_mblk workgroups per pair range, XCD-grouped (kp_block_args)
    a.m_flat = a.n_md > 0 && !env().kp_multi_2d;
    a.m_ranges = (int)groups;
    if (!a.m_flat) return dim3((unsigned)groups, (unsigned)(a.n_md > 0 ? a.n_mblk : 1));
    const int64_t flat = (groups * a.n_md + 7) / 8 * 8;               // (range, document tensor) slots, whole groups of 8
    return dim3((unsigned)(flat * (a.n_mblk / a.n_md)), 1u);
  };
  // Conv-KNRM's multi launch at E <= 128 with three query tensors: one wavefront per (pair range, document tensor) looping over
  // the query tensors (kernel_pool_multi128_kernel): every document block crosses HBM once.  The default once every SIMD's
  // wavefront has a few pairs of its own (with the recurrence epilogue: 8.38 vs 8.66 ms and half the traffic, see there); smaller
  // launches keep the per-combination form, which has n_mq times the wavefronts.  MM_KP_MULTI_LOOP = 0 / 1 forces either (A/B runs).
  const bool loop_auto = a.n_pairs >= (int64_t)kCUs * 4 * 2;
  if (a.n_md > 0 && !gated && nsl <= 2 && a.n_mblk / a.n_md == 3 && !a.pair_q &&
      (env().kp_multi_loop == 1 || (env().kp_multi_loop < 0 && loop_auto))) {
    const int ldsm = kp128_lds_fixed(1) + 256;      // + the scratch row of the query-tile prefetch
    int64_t groups = (int64_t)kCUs * 4 / a.n_md;
    const bool xcd = !env().kp_multi_2d && groups >= 8;   // XCD-grouped ids (see the kernel); MM_KP_MULTI_2D=1: consecutive ids (A/B runs)
    if (xcd) groups = groups / 8 * 8;                     // whole groups of 8 ranges, all resident at once (one wavefront per SIMD)
    if (groups < 1) groups = 1;
    if (groups > a.n_pairs) groups = a.n_pairs;
    a.pairs_per_wave = (a.n_pairs + groups - 1) / groups;
    groups = (a.n_pairs + a.pairs_per_wave - 1) / a.pairs_per_wave;
    a.m_flat = xcd ? 3 : 0;
    a.m_ranges = (int)groups;
    const dim3 grid((unsigned)((xcd ? (groups + 7) / 8 * 8 : groups) * a.n_md));
    if (nsl == 1) hipLaunchKernelGGL((kernel_pool_multi128_kernel<1, 11, 3>), grid, dim3(64), ldsm, stream, a);
    else hipLaunchKernelGGL((kernel_pool_multi128_kernel<2, 11, 3>), grid, dim3(64), ldsm, stream, a);
    return check_launch("kernel_pool_multi128_kernel");
  }
  if (occ2 && a.n_md > 0 && !env().kp_multi_2d && env().kp_multi_wg == 1) {
    // one workgroup per (pair range, document tensor), its wavefronts = the query tensors (m_flat = 2).  Wavefront slots per
    // CU: 8 (two per SIMD) in workgroups of n_mq -> 2 x 3 for Conv-KNRM's three n-gram widths
    const int n_mq = a.n_mblk / a.n_md;
    const int lds2 = n_mq * kp128_lds_fixed(1, 2);
    int64_t groups = (int64_t)kCUs * (8 / n_mq) / a.n_md;
    if (groups < 1) groups = 1;
    if (groups > a.n_pairs) groups = a.n_pairs;
    a.pairs_per_wave = (a.n_pairs + groups - 1) / groups;
    groups = (a.n_pairs + a.pairs_per_wave - 1) / a.pairs_per_wave;
    a.m_flat = 2;
    a.m_ranges = (int)groups;
    const dim3 grid((unsigned)(groups * a.n_md));
    if (lds2 > 64 * 1024) {
      (void)hipFuncSetAttribute((const void*)kernel_pool_split128_kernel<1, 11, false, 1, 0, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
      (void)hipFuncSetAttribute((const void*)kernel_pool_split128_kernel<2, 11, false, 1, 0, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
    }
    if (nsl == 1) hipLaunchKernelGGL((kernel_pool_split128_kernel<1, 11, false, 1, 0, 2, 4>), grid, dim3(64 * n_mq), lds2, stream, a);
    else hipLaunchKernelGGL((kernel_pool_split128_kernel<2, 11, false, 1, 0, 2, 4>), grid, dim3(64 * n_mq), lds2, stream, a);
    return check_launch("kernel_pool_split128_kernel<one wavefront per query tensor>");
  }
  if (occ2) {
    const int lds2 = kp128_lds_fixed(1, 2);
    int64_t groups = (int64_t)kCUs * per_cu;
    if (groups > a.n_pairs) groups = a.n_pairs;
    a.pairs_per_wave = (a.n_pairs + groups - 1) / groups;
    groups = (a.n_pairs + a.pairs_per_wave - 1) / a.pairs_per_wave;
    const dim3 grid = flat_grid(groups);
    if (nsl == 1) hipLaunchKernelGGL((kernel_pool_split128_kernel<1, 11, false, 1, 0, 2>), grid, dim3(64), lds2, stream, a);
    else hipLaunchKernelGGL((kernel_pool_split128_kernel<2, 11, false, 1, 0, 2>), grid, dim3(64), lds2, stream, a);
    return check_launch("kernel_pool_split128_kernel<two wavefronts per SIMD>");
  }
  const int lds = kp128_lds_fixed(ks) + (gated ? ks * 128 * ((a.D + 31) >> 5) : 0);
  int64_t groups = (int64_t)kCUs * 4 / ks;  // one wave per SIMD either way
  if (groups > a.n_pairs) groups = a.n_pairs;
  a.pairs_per_wave = (a.n_pairs + groups - 1) / groups;
  groups = (a.n_pairs + a.pairs_per_wave - 1) / a.pairs_per_wave;
  const dim3 grid = flat_grid(groups);
#define MM_KP128(NSL, KS) \
  return gated ? launch128<NSL, true, KS>(a, grid, lds, stream) : launch128<NSL, false, KS>(a, grid, lds, stream)
  switch (nsl) {
    case 1: MM_KP128(1, 1);
    case 2: MM_KP128(2, 1);
    case 3: MM_KP128(3, 1);
    case 4: MM_KP128(4, 1);
    case 6: MM_KP128(6, 1);
    case 8: MM_KP128(4, 2);
    case 12: MM_KP128(6, 2);
  }
#undef MM_KP128
  return set_error(MM_EUNSUPPORTED, "kernel_pool: E=%d has no 64-float streaming kernel", a.E);
}

}  // namespace mm
