// ColBERT late-interaction MaxSim for MI355X (gfx950 / CDNA4).
//
//   out[p] = sum_{i<Q, qmask}  max_{j<D} ( dmask[p,j] ? <q_i, d_{p,j}> : -1000 )
//
// Reference semantics: matchmaker/models/colbert.py:68-75 (masked, paired), :100-112 (unmasked),
// :154-162 (all pairs).  Sentinel is -1000 (not -inf); query padding contributes 0.
//
// Two kernels:
//   * maxsim_stream_kernel — the roofline path (16-bit dtypes, E == 128, Q <= 32).  One wavefront
//     per workgroup streams its documents through a private LDS ring with LDS-DMA
//     (global_load_lds_dwordx4: 1 KiB of contiguous HBM per instruction, no VGPR round trip),
//     keeps the whole query tile as MFMA B-fragments in 32 VGPRs, and computes each 32-token
//     document block with 8 x v_mfma_f32_32x32x16_{bf16,f16}.  Document tokens sit on the MFMA M
//     axis so the max over document tokens is an element-wise running max of the 16 accumulator
//     registers; one cross-half exchange + one wave reduction per pair finish the score.
//   * maxsim_generic_kernel — any E/Q/D and fp32: fragment-shaped direct loads, same MFMA maps.
//
// HBM-bound by design: 2*Q*E flop per D-side byte pair = 32 flop/B at bf16, far below the MFMA
// ridge, so everything here is about keeping >= 16 KiB of D stream in flight per wavefront.
#include "mm_internal.h"
#include "maxsim_device.h"

namespace mm {

// NSL = E / 128: a 32-token block is streamed as NSL slices of 32 rows x 256 B (one ring slot each);
// the accumulator runs across the slices, the query tile is NSL x 32 VGPRs of B fragments.
// NQT = query tiles of 32 tokens held in registers (Q <= 32 * NQT): the reference's defaults stay under 32
// (max_query_length 30, defaults.yaml:127) but ColBERT's [MASK] query augmentation
// (query_augment_mask_number, independent_reranking_loader.py:106-112) pushes Q to 38.
// INB: all-pairs mode (colbert.py:154-162): pair p = (query p / Bd, document p % Bd), masked with the document's own
// mask row or — bug-compatible, :158 — with the row of the query index.  A wavefront's pairs are consecutive documents of
// one query, so the query tile stays in registers and the documents stream from L2 / the Infinity Cache (they are read
// once per query).
// INB = 2: all-pairs TILED over queries: the NQT register tiles hold NQT DIFFERENT queries (Q <= 32), every document
// block read from LDS feeds NQT x 8 MFMAs and the documents are read once per query GROUP.  Work map: workgroup
// (xcd = blockIdx % 8, t, jg) sweeps document slice xcd * T + t of 8T for the query groups jg, jg + Gw, ... — the
// wavefronts of one XCD stream the same documents for different queries at the same time, so a slice comes through
// that XCD's L2 once per sweep instead of once per group.
// IM: the tokenizer's int64 masks are read by the kernel itself (lane loads + ballots before the first block is put in
// flight) — for launches in which every wavefront scores ONE pair, i.e. eval.py-sized calls of 512 pairs, where the separate
// mask-packing launch was 11 of the call's 45 us at the published checkpoint's shapes.
// WPP = 2 (with IM): TWO wavefronts per pair — when a call has fewer pairs than half the wavefront slots (512 pairs on 1,024
// SIMDs) wavefront w streams blocks w, w + 2, ... of the pair through its own ring and the running maxima meet in LDS once.
template <int DT, int NBUF, bool NT, int NSL, bool RAG, int NQT, int INB, bool IM = false, int WPP = 1>
__device__ __forceinline__ void maxsim_stream_body(const MaxsimArgs& a) {
  static_assert(WPP == 1 || (IM && NQT >= 2 && !RAG && INB == 0), "two wavefronts per pair: the one-pair-per-workgroup launches only");
  constexpr int RB = NSL * 256;  // bytes per token row
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  const int lane = WPP > 1 ? (threadIdx.x & 63) : threadIdx.x;
  const int wv = WPP > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
  char* const smem = smem_all + wv * (NBUF * kBlkBytes);   // this wavefront's ring
  const int r = lane & 31, h = lane >> 5;
  int64_t p0 = (int64_t)blockIdx.x * a.pairs_per_wave;
  int64_t p1 = (p0 + a.pairs_per_wave < a.n_pairs) ? p0 + a.pairs_per_wave : a.n_pairs;
  int64_t ppq = a.ppq;
  int64_t d_first = 0, nd = 1;   // INB == 2: this wavefront's document slice
  int g0 = 0;                    // INB == 2: first query group (then every inb_gw-th)
  if (INB == 2) {
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int t = rest % a.inb_t;
    g0 = rest / a.inb_t;
    const int64_t S = 8 * (int64_t)a.inb_t, si = (int64_t)xcd * a.inb_t + t;
    d_first = a.inb_bd * si / S;
    nd = a.inb_bd * (si + 1) / S - d_first;
    const int G = (int)((a.inb_bq + NQT - 1) / NQT);
    const int ng = g0 < G ? (G - g0 + a.inb_gw - 1) / a.inb_gw : 0;
    p0 = 0;                      // virtual pair v = (v / nd)-th group of this wavefront x document d_first + v % nd
    p1 = (int64_t)ng * nd;
    ppq = nd;
  }
  if (p0 >= p1) return;
  const int D = a.D, Q = a.Q;
  const int nblk_tot = (D + 31) >> 5;
  const int rows_last = D - 32 * (nblk_tot - 1);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // per-lane source offsets of the 8 LDS-DMA instructions of a block (and of the last block of a
  // document, whose rows past D are redirected to the last real row: never read past the tensor)
  uint32_t voff[8], voff_tail[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int row = 4 * k + (lane >> 4);
    const int c = (lane & 15) ^ (row & 15);
    const int rowt = row < rows_last ? row : rows_last - 1;
    voff[k] = (uint32_t)(row * RB + c * 16);
    voff_tail[k] = (uint32_t)(rowt * RB + c * 16);
  }
  // per-lane LDS offsets of the 8 A-fragment reads: chunk 2kk+h of row r lives at slot (2kk+h)^(r&15)
  uint32_t lo[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) lo[kk] = (uint32_t)(r * 256 + ((((2 * kk) | h) ^ (r & 15)) << 4));

  const char* dbase = (const char*)a.d;
  // (INB == 2: 32-bit arithmetic, made scalar again for the s_load helpers; a slice has < 2^31 virtual pairs)
  auto doc_row = [&](int64_t p) -> int64_t {
    if (INB == 2) return d_first + __builtin_amdgcn_readfirstlane((int)p % (int)nd);
    return INB ? p % a.inb_bd : p;
  };
  auto mask_row = [&](int64_t p) -> int64_t {
    if (INB == 2) return doc_row(p);
    return INB ? (a.inb_bug ? p / a.inb_bd : p % a.inb_bd) : p;
  };
  // IM: validity bits of this wavefront's pair (D <= 256, Q <= 64) and its length = last real token + 1, the values
  // pack_mask2_kernel would have written (common.hip)
  unsigned long long ib0 = 0, ib1 = 0, ib2 = 0, ib3 = 0, iq = 0;
  int ilen = 0;
  if constexpr (IM) {
    const int64_t* dmr = a.dm64 + p0 * D;
    const int64_t* qmr = a.qm64 + (p0 / ppq) * Q;
    const int64_t v0 = lane < D ? dmr[lane] : 0;
    const int64_t v1 = 64 + lane < D ? dmr[64 + lane] : 0;
    const int64_t v2 = 128 + lane < D ? dmr[128 + lane] : 0;
    const int64_t v3 = 192 + lane < D ? dmr[192 + lane] : 0;
    const int64_t vq = lane < Q ? qmr[lane] : 0;
    ib0 = __ballot(v0 != 0);
    ib1 = __ballot(v1 != 0);
    ib2 = __ballot(v2 != 0);
    ib3 = __ballot(v3 != 0);
    iq = __ballot(vq != 0);
    ilen = ib3 ? 256 - __builtin_clzll(ib3) : ib2 ? 192 - __builtin_clzll(ib2) : ib1 ? 128 - __builtin_clzll(ib1)
                                                                               : ib0 ? 64 - __builtin_clzll(ib0) : 0;
  }
  auto doc_len = [&](int64_t p) -> int {
    if (IM) return ilen;
    if (RAG) {
      const int64_t l = sload_i64(a.rag_end, p) - sload_i64(a.rag_begin, p);
      return l < 0 ? 0 : (l > 0x7fffffe0LL ? 0x7fffffe0 : (int)l);
    }
    int len = a.dm.len ? (int)sload_u32(a.dm.len, mask_row(p)) : D;
    return len < 0 ? 0 : (len > D ? D : len);
  };

  // ---- producer cursor: next (pair, block, slice) to put in flight ---------------------------
  int64_t pp = p0;
  int pt = wv, pn = 0, psl = 0, plen = 0;
  while (pp < p1 && (pn = ((plen = doc_len(pp)) + 31) >> 5) <= wv) ++pp;   // (WPP = 1: skips empty documents)
  int64_t prow0 = (INB == 2 && pp < p1) ? doc_row(pp) * D : 0;
  int pbuf = 0, cbuf = 0, inflight = 0;

  auto top_up = [&]() {
    while (pp < p1 && inflight < NBUF) {
      // (INB == 2: doc_row is a division; done once per document, when the cursor moves)
      const int64_t row0 = RAG ? sload_i64(a.rag_begin, pp) : (INB == 2 ? prow0 : doc_row(pp) * D);
      const char* g = dbase + (row0 + (int64_t)pt * 32) * RB + psl * 256;
      const uint32_t dst = lds0 + (uint32_t)pbuf * kBlkBytes;
      if (RAG) {
        // documents have no common padded length: rows past this document's end are redirected to
        // its last row (the last document must not read past the token matrix)
        const int rl = plen - 32 * pt;  // rows of this block that exist (>= 1)
        if (rl < 32) {
          uint32_t vt[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int over = 4 * k + (lane >> 4) - (rl - 1);
            vt[k] = voff[k] - (uint32_t)((over > 0 ? over : 0) * RB);
          }
          issue_block<NT>(g, vt, dst);
        } else {
          issue_block<NT>(g, voff, dst);
        }
      } else if (pt == nblk_tot - 1 && rows_last != 32)
        issue_block<NT>(g, voff_tail, dst);
      else
        issue_block<NT>(g, voff, dst);
      pbuf = (pbuf + 1 == NBUF) ? 0 : pbuf + 1;
      ++inflight;
      if (NSL > 1 && ++psl < NSL) continue;
      psl = 0;
      pt += WPP;
      if (pt >= pn) {
        pt = wv;
        ++pp;
        while (pp < p1 && (pn = ((plen = doc_len(pp)) + 31) >> 5) <= wv) ++pp;
        if (INB == 2 && pp < p1) prow0 = doc_row(pp) * D;
      }
    }
  };
  top_up();

  // ---- query tile(s) as MFMA B fragments ------------------------------------------------------
  short8 qf[NQT][NSL][8];
  bool qvalid[NQT];
#pragma unroll
  for (int n = 0; n < NQT; ++n) qvalid[n] = false;
  int64_t cur_q = -1;
  int64_t qi = p0 / ppq;
  int64_t q_left = ppq - (p0 - qi * ppq);  // pairs left on this query
  const int qwords = (Q + 31) >> 5;

  for (int64_t pair = p0; pair < p1; ++pair) {
    if (q_left == 0) {
      ++qi;
      q_left = ppq;
    }
    --q_left;
    if (qi != cur_q) {
      cur_q = qi;
      if (INB == 2) {
        // tile n = query (g0 + qi * inb_gw) * NQT + n, all of its (<= 32) tokens
#pragma unroll
        for (int n = 0; n < NQT; ++n) {
          int64_t qq = ((int64_t)g0 + qi * a.inb_gw) * NQT + n;
          const bool exists = qq < a.inb_bq;
          qq = exists ? qq : a.inb_bq - 1;
          const int qlen = a.qm.len ? (int)sload_u32(a.qm.len, qq) : Q;
          const int qr = r < Q ? r : Q - 1;
          const char* qrow = (const char*)a.q + (qq * Q + qr) * RB;
#pragma unroll
          for (int sl = 0; sl < NSL; ++sl) load_q_frags(qrow + sl * 256 + h * 16, qf[n][sl]);
          qvalid[n] = exists && r < Q && r < qlen;
          if (a.qm.bits) qvalid[n] = qvalid[n] && ((sload_u32(a.qm.bits, qq) >> r) & 1u);   // Q <= 32: one word per query
        }
      } else {
        const int qlen = (!IM && a.qm.len) ? (int)sload_u32(a.qm.len, qi) : Q;
#pragma unroll
        for (int n = 0; n < NQT; ++n) {
          const int qt = 32 * n + r;
          const int qr = qt < Q ? qt : Q - 1;
          const char* qrow = (const char*)a.q + (qi * Q + qr) * RB;
#pragma unroll
          for (int sl = 0; sl < NSL; ++sl) {
            // two tiles at dim >= 512 exceed the 256 VGPRs: the second tile lives in AGPRs and is read from there by the MFMA
            // (284 -> 92 v_accvgpr_read per block at dim 768; 0.885 -> 0.878 ms on the published checkpoint's shapes)
            if (NQT == 2 && NSL >= 4 && n == 1) load_q_frags_agpr(qrow + sl * 256 + h * 16, qf[n][sl]);
            else load_q_frags(qrow + sl * 256 + h * 16, qf[n][sl]);
          }
          qvalid[n] = qt < Q && qt < qlen;
          if (IM) qvalid[n] = qvalid[n] && ((iq >> qt) & 1ull);
          else if (a.qm.bits && n < qwords) qvalid[n] = qvalid[n] && ((sload_u32(a.qm.bits, qi * qwords + n) >> r) & 1u);
        }
      }
    }
    const int len = doc_len(pair);
    const int nb = (len + 31) >> 5;
    // ragged documents have no padded positions; an empty one scores like a fully padded one
    const float fill = (RAG ? len == 0 : len < D) ? -1000.0f : neg_inf();
    // several register tiles (two tiles of one long query, or the queries of the tiled all-pairs mode) keep ONE running
    // maximum per lane and tile (block_max1): at E = 768 two query tiles are 384 registers of B fragments already
    constexpr bool ONE = INB == 2 || NQT >= 2;
    float m[NQT][16];
    float m1[NQT];
#pragma unroll
    for (int n = 0; n < NQT; ++n) {
      m1[n] = fill;
#pragma unroll
      for (int i = 0; i < 16; ++i) m[n][i] = fill;
    }

    for (int t = wv; t < nb; t += WPP) {
      f32x16 acc[NQT];
#pragma unroll
      for (int n = 0; n < NQT; ++n) acc[n] = f32x16{0};
#pragma unroll
      for (int sl = 0; sl < NSL; ++sl) {
        top_up();
        wait_block(inflight - 1);
        const char* buf = smem + cbuf * kBlkBytes;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const short8 av = *(const short8*)(buf + lo[kk]);
#pragma unroll
          for (int n = 0; n < NQT; ++n) acc[n] = Mfma32x16<DT>::run(av, qf[n][sl][kk], acc[n]);
        }
        cbuf = (cbuf + 1 == NBUF) ? 0 : cbuf + 1;
        --inflight;
      }
      const int rem = len - 32 * t;
      const uint32_t ex = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
      uint32_t va;
      if (IM) {
        const unsigned long long w = t < 2 ? ib0 : t < 4 ? ib1 : t < 6 ? ib2 : ib3;
        va = (uint32_t)(w >> (32 * (t & 1))) & ex;
      } else {
        va = (!RAG && a.dm.bits) ? (sload_u32(a.dm.bits, mask_row(pair) * nblk_tot + t) & ex) : ex;
      }
#pragma unroll
      for (int n = 0; n < NQT; ++n) {
        if (ONE) block_max1(m1[n], acc[n], ex, va, fill, h);
        else block_max(m[n], acc[n], ex, va, fill, h);
      }
    }
    if (INB == 2) {
      const int64_t dj = doc_row(pair);
#pragma unroll
      for (int n = 0; n < NQT; ++n) {
        const float s = finish_sum<DT>(finish_pair1<DT>(m1[n], qvalid[n], h, a.rnd), a.rnd);
        const int64_t qq = ((int64_t)g0 + qi * a.inb_gw) * NQT + n;
        if (lane == 0 && qq < a.inb_bq) a.out[qq * a.inb_bd + dj] = s;
      }
    } else {
      if constexpr (WPP > 1) {
        // the second wavefront's maxima (its ring is drained: every block it requested has been consumed)
        float* comb = (float*)(smem_all + WPP * NBUF * kBlkBytes);   // [NQT][64]
        if (wv == 1) {
#pragma unroll
          for (int n = 0; n < NQT; ++n) comb[n * 64 + lane] = m1[n];
        }
        __syncthreads();
        if (wv == 1) return;
#pragma unroll
        for (int n = 0; n < NQT; ++n) m1[n] = fmaxf(m1[n], comb[n * 64 + lane]);
      }
      float s = 0.0f;
#pragma unroll
      for (int n = 0; n < NQT; ++n)  // tiles in index order: deterministic
        s += ONE ? finish_pair1<DT>(m1[n], qvalid[n], h, a.rnd) : finish_pair<DT>(m[n], qvalid[n], h, a.rnd);
      if (lane == 0) a.out[pair] = finish_sum<DT>(s, a.rnd);
    }
  }
}

template <int DT, int NBUF, bool NT, int NSL, bool RAG, int NQT, int INB = 0, bool IM = false>
__global__ void __launch_bounds__(64) maxsim_stream_kernel(const MaxsimArgs a) {
  maxsim_stream_body<DT, NBUF, NT, NSL, RAG, NQT, INB, IM>(a);
}

template <int DT, int NBUF, bool NT, int NSL, int NQT>
__global__ void __launch_bounds__(128) maxsim_stream_wpp2_kernel(const MaxsimArgs a) {
  maxsim_stream_body<DT, NBUF, NT, NSL, false, NQT, 0, true, 2>(a);
}

// The tiled all-pairs instantiations are compiled for two wavefronts per SIMD (<= 256 registers): that makes the
// compiler keep the MFMA accumulators in VGPRs, where the max epilogue can read them — with the whole 512-register file
// it parks them in AGPRs and spends 64 v_accvgpr_read per block on getting them back.
template <int DT, int NSL, int NQT>
__global__ void __launch_bounds__(64, 2) maxsim_allpairs_tiled_kernel(const MaxsimArgs a) {
  maxsim_stream_body<DT, 2, false, NSL, false, NQT, 2>(a);
}


// ---------------------------------------------------------------------------------------------
// All pairs, workgroup-shared document ring (large teacher batches).
//
// maxsim_allpairs_tiled_kernel gives every wavefront its own 2-slab ring, so the four wavefronts of a CU pull the same
// documents through LDS four times (12.9 GB of LDS-DMA per 1024 x 1024 launch) and each pays 8 LDS-DMA instructions per
// 32 MFMAs: 29 % of the MFMA peak (profiles/r02_allpairs_pmc.json: MFMA busy 31 %).  Here the 4 wavefronts of a
// workgroup hold 4 x NQT DIFFERENT queries and share ONE ring: per 8 KiB slab a wavefront issues only its quarter (two
// LDS-DMA instructions), waits for those two with a counted vmcnt, and one s_barrier makes the other three quarters
// visible; the slot a slab leaves is refilled after the NEXT slab's barrier, when every wavefront is done reading it
// (ring depth 3: two slabs in flight while one is computed).  Work map as in the tiled kernel: (XCD = document slice) x
// (query-group lane), two workgroups per CU.
// ---------------------------------------------------------------------------------------------
template <bool TAIL>
__device__ __forceinline__ void issue_quarter(const char* gbase, uint32_t v0, uint32_t v1, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %4\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %3\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %3\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v0), "v"(v1), "s"(gbase), "s"(lds_dst)
      : "memory", "scc");
}

template <int DT, int NSL, int NQT>
__global__ void __launch_bounds__(256, 2) maxsim_allpairs_wg_kernel(const MaxsimArgs a) {
  constexpr int RB = NSL * 256;
  constexpr int NB = NSL == 1 ? 3 : 4;                  // ring depth in 8 KiB slabs
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
  const int ts = rest % a.inb_t, jg = rest / a.inb_t;
  const int64_t S = 8 * (int64_t)a.inb_t, si = (int64_t)xcd * a.inb_t + ts;
  const int64_t d_first = a.inb_bd * si / S;
  const int nd = (int)(a.inb_bd * (si + 1) / S - d_first);
  const int G = (int)((a.inb_bq + 4 * NQT - 1) / (4 * NQT));
  if (nd <= 0 || jg >= G) return;
  const int D = a.D, Q = a.Q;
  const int nblk_tot = (D + 31) >> 5;
  const int rows_last = D - 32 * (nblk_tot - 1);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // this wavefront's two LDS-DMA instructions of a slab: instruction k = 2 wv + u moves rows 4k..4k+3 (16 lanes per row),
  // source-side swizzle as in issue_block; tail variant: rows past D are redirected to the last real row
  uint32_t vq[2], vq_tail[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = 4 * (2 * wv + u) + (lane >> 4);
    const int c = (lane & 15) ^ (row & 15);
    const int rowt = row < rows_last ? row : rows_last - 1;
    vq[u] = (uint32_t)(row * RB + c * 16);
    vq_tail[u] = (uint32_t)(rowt * RB + c * 16);
  }
  uint32_t lo[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) lo[kk] = (uint32_t)(r * 256 + ((((2 * kk) | h) ^ (r & 15)) << 4));

  const char* dbase = (const char*)a.d;
  auto doc_len = [&](int dd) -> int {
    int len = a.dm.len ? (int)sload_u32(a.dm.len, d_first + dd) : D;
    return len < 0 ? 0 : (len > D ? D : len);
  };

  for (int g = jg; g < G; g += a.inb_gw) {
    // ---- this wavefront's NQT queries as MFMA B fragments (the ring is empty here: plain loads may use vmcnt(0)) ----
    short8 qf[NQT][NSL][8];
    bool qvalid[NQT];
#pragma unroll
    for (int n = 0; n < NQT; ++n) {
      int64_t qq = ((int64_t)g * 4 + wv) * NQT + n;
      const bool exists = qq < a.inb_bq;
      qq = exists ? qq : a.inb_bq - 1;
      const int qlen = a.qm.len ? (int)sload_u32(a.qm.len, qq) : Q;
      const int qr = r < Q ? r : Q - 1;
      const char* qrow = (const char*)a.q + (qq * Q + qr) * RB;
#pragma unroll
      for (int sl = 0; sl < NSL; ++sl) load_q_frags(qrow + sl * 256 + h * 16, qf[n][sl]);
      qvalid[n] = exists && r < Q && r < qlen;
      if (a.qm.bits) qvalid[n] = qvalid[n] && ((sload_u32(a.qm.bits, qq) >> r) & 1u);
    }
    // nobody may still be reading the previous group's last slabs when the first slabs of this one are requested
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- producer cursor over (document, block, slab); identical in all four wavefronts --------------------------
    int pd = 0, pt = 0, psl = 0, pn = 0;
    while (pd < nd && (pn = (doc_len(pd) + 31) >> 5) == 0) ++pd;
    int pbuf = 0, cbuf = 0, inflight = 0;
    int pre = 0, nst = 0;   // score stores share the in-order vmcnt queue: `pre` slabs are older than the last burst of `nst` stores
    auto top_up = [&]() {
      while (pd < nd && inflight < NB) {
        const char* gsrc = dbase + ((d_first + pd) * (int64_t)D + (int64_t)pt * 32) * RB + psl * 256;
        const uint32_t dst = lds0 + (uint32_t)pbuf * kBlkBytes + (uint32_t)wv * 2048;
        if (pt == nblk_tot - 1 && rows_last != 32)
          issue_quarter<true>(gsrc, vq_tail[0], vq_tail[1], dst);
        else
          issue_quarter<false>(gsrc, vq[0], vq[1], dst);
        pbuf = (pbuf + 1 == NB) ? 0 : pbuf + 1;
        ++inflight;
        if (NSL > 1 && ++psl < NSL) continue;
        psl = 0;
        if (++pt == pn) {
          pt = 0;
          ++pd;
          while (pd < nd && (pn = (doc_len(pd) + 31) >> 5) == 0) ++pd;
        }
      }
    };
    top_up();

    for (int dd = 0; dd < nd; ++dd) {
      const int len = doc_len(dd);
      const int nb = (len + 31) >> 5;
      const float fill = len < D ? -1000.0f : neg_inf();
      float m1[NQT];
#pragma unroll
      for (int n = 0; n < NQT; ++n) m1[n] = fill;
      for (int t = 0; t < nb; ++t) {
        f32x16 acc[NQT];
#pragma unroll
        for (int n = 0; n < NQT; ++n) acc[n] = f32x16{0};
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) {
          // my quarter of the oldest slab has landed (two instructions per slab in flight behind it, + the last
          // document's score stores while they are younger than it) ...
          if (pre > 0) {
            wait_vm(2 * (inflight - 1) + nst);
            --pre;
          } else {
            nst = 0;
            wait_vm(2 * (inflight - 1));
          }
          // ... and so have the others'; everyone is also done with the slab before it, whose slot is refilled now
          // (lgkmcnt(0): this wavefront's own fragment reads of that slab have RETURNED before anyone's DMA may overwrite it)
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          top_up();
          const char* buf = smem + cbuf * kBlkBytes;
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
            const short8 av = *(const short8*)(buf + lo[kk]);
#pragma unroll
            for (int n = 0; n < NQT; ++n) acc[n] = Mfma32x16<DT>::run(av, qf[n][sl][kk], acc[n]);
          }
          cbuf = (cbuf + 1 == NB) ? 0 : cbuf + 1;
          --inflight;
        }
        const int rem = len - 32 * t;
        const uint32_t ex = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
        const uint32_t va = a.dm.bits ? (sload_u32(a.dm.bits, (d_first + dd) * nblk_tot + t) & ex) : ex;
#pragma unroll
        for (int n = 0; n < NQT; ++n) block_max1(m1[n], acc[n], ex, va, fill, h);
      }
      int stores = 0;
#pragma unroll
      for (int n = 0; n < NQT; ++n) {
        const float sc = finish_sum<DT>(finish_pair1<DT>(m1[n], qvalid[n], h, a.rnd), a.rnd);
        const int64_t qq = ((int64_t)g * 4 + wv) * NQT + n;
        if (qq < a.inb_bq) {                                 // wave-uniform
          if (lane == 0) a.out[qq * a.inb_bd + d_first + dd] = sc;
          ++stores;
        }
      }
      nst = stores;       // (counting fewer than were issued would only wait longer; `stores` is exact)
      pre = inflight;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Generic path: any E (16-B aligned rows), any Q, any D; bf16 / f16 / f32.
// One wavefront per pair; query tiles of 32 tokens looped sequentially (deterministic sum order).
// ---------------------------------------------------------------------------------------------
template <int DT>
__device__ __forceinline__ f32x16 dot_block(const char* drow, const char* qrow, int E) {
  f32x16 acc = {0};
  if constexpr (DT == MM_F32) {
    // v_mfma_f32_32x32x2_f32: lane (r,h) supplies A[r][k=h].  K is walked in 16-B chunks: chunk
    // pair (2c, 2c+1) -> h=0 takes chunk 2c, h=1 chunk 2c+1; the 4 floats of a chunk are 4 steps.
    const int h = (threadIdx.x >> 5) & 1;      // lane half within the wavefront (workgroups may hold several)
    const int nch = E >> 2;
    for (int c = 0; c < nch; c += 2) {
      const int cc = c + h;
      f32x4 av = {0, 0, 0, 0}, bv = {0, 0, 0, 0};
      if (cc < nch) {
        av = *(const f32x4*)(drow + cc * 16);
        bv = *(const f32x4*)(qrow + cc * 16);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
    }
  } else {
    const int h = (threadIdx.x >> 5) & 1;      // lane half within the wavefront (workgroups may hold several)
    const int nch = E >> 3;
    // four K steps per trip: their eight 16-byte loads are issued before the first MFMA waits for any of them (the one
    // load pair -> one MFMA form ran a row block as nch / 2 dependent memory round trips); same accumulation order
    for (int c = 0; c < nch; c += 8) {
      short8 av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cc = c + 2 * u + h;
        av[u] = short8{0, 0, 0, 0, 0, 0, 0, 0};
        bv[u] = av[u];
        if (cc < nch) {
          av[u] = *(const short8*)(drow + cc * 16);
          bv[u] = *(const short8*)(qrow + cc * 16);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (c + 2 * u < nch) acc = Mfma32x16<DT == MM_F32 ? MM_BF16 : DT>::run(av[u], bv[u], acc);   // wave-uniform
    }
  }
  return acc;
}

template <int DT>
__global__ void __launch_bounds__(64) maxsim_generic_kernel(const MaxsimArgs a) {
  const int lane = threadIdx.x;
  const int r = lane & 31, h = lane >> 5;
  const int64_t pair = blockIdx.x;
  if (pair >= a.n_pairs) return;
  const int Q = a.Q, E = a.E;
  constexpr int ES = (DT == MM_F32) ? 4 : 2;
  const int64_t rowb = (int64_t)E * ES;
  int64_t qi, di, mi;
  if (a.inb_bd > 0) {
    qi = pair / a.inb_bd;
    di = pair - qi * a.inb_bd;
    mi = a.inb_bug ? qi : di;
  } else {
    qi = pair / a.ppq;
    di = pair;
    mi = pair;
  }
  const bool rag = a.rag_begin != nullptr;
  int D = a.D;
  int64_t drow0 = di * D;
  if (rag) {  // document = rows [begin, end) of the token matrix; every row is a real token
    drow0 = a.rag_begin[pair];
    const int64_t l = a.rag_end[pair] - drow0;
    D = l < 0 ? 0 : (l > 0x7fffffe0LL ? 0x7fffffe0 : (int)l);
  }
  const int nblk_tot = (D + 31) >> 5;
  const int qwords = (Q + 31) >> 5;
  int len = (!rag && a.dm.len) ? a.dm.len[mi] : D;
  len = len < 0 ? 0 : (len > D ? D : len);
  const int nb = (len + 31) >> 5;
  const float fill = (rag ? len == 0 : len < D) ? -1000.0f : neg_inf();
  const int qlen = a.qm.len ? a.qm.len[qi] : Q;
  const char* dbase = (const char*)a.d + drow0 * rowb;
  const char* qbase = (const char*)a.q + qi * Q * rowb;

  float total = 0.0f;
  for (int n = 0; n < qwords; ++n) {
    const int qtok = 32 * n + r;
    const int qr = qtok < Q ? qtok : Q - 1;
    bool qvalid = qtok < Q && qtok < qlen;
    if (a.qm.bits) qvalid = qvalid && ((a.qm.bits[qi * qwords + n] >> r) & 1u);
    float m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = fill;
    for (int t = 0; t < nb; ++t) {
      const int drow = 32 * t + r;
      const int dr = drow < D ? drow : D - 1;
      const f32x16 acc = dot_block<DT>(dbase + dr * rowb, qbase + qr * rowb, E);
      const int rem = len - 32 * t;
      const uint32_t ex = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
      const uint32_t va = (!rag && a.dm.bits) ? (a.dm.bits[mi * nblk_tot + t] & ex) : ex;
      block_max(m, acc, ex, va, fill, h);
    }
    total += finish_pair<DT>(m, qvalid, h, a.rnd);
  }
  if (lane == 0) a.out[pair] = finish_sum<DT>(total, a.rnd);
}

// ---------------------------------------------------------------------------------------------
// Backward of the paired MaxSim (training path, train.py:503-524): one wavefront per pair.
//   d out / d s[i, j] = g  for j = j*(i) = the FIRST arg-max over the document positions of query
//   token i (torch.max's tie rule), 0 elsewhere; nothing flows when the maximum is the -1000
//   sentinel of a padded position (colbert.py:69 assigns a constant there) or the query token is
//   padding (:73).  Hence
//     grad_q[p, i, :]   = g[p] * d[p, j*(i), :]
//     grad_d[p, j, :]   = g[p] * sum_{i : j*(i) = j} q[p, i, :]
// The similarities are recomputed with the same MFMA maps as the forward generic kernel (the
// forward keeps no [B,Q,D] tensor to save); gradients are written as float32.
// ---------------------------------------------------------------------------------------------
struct MaxsimBwdArgs {
  const void* q;
  const void* d;
  PackedMask qm, dm;
  const float* go;
  void* gq;
  void* gd;
  int64_t n_pairs;
  int Q, D, E;
  int row_masks;   // 1: LDS holds, per document row, the bit set of the query tokens whose first arg-max it is
};

// One 16-byte chunk of a token row as floats (4 for float32 rows, 8 for 16-bit rows), and the matching gradient store
// (16 bytes for 16-bit gradients, 16 or 2 x 16 for float32 ones).
template <int DT>
__device__ __forceinline__ void load_chunk(const char* p, float* v) {
  if constexpr (DT == MM_F32) {
    const f32x4 x = *(const f32x4*)p;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = x[k];
  } else if constexpr (DT == MM_F16) {
    typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
    const half8_t x = *(const half8_t*)p;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (float)x[k];
  } else {
    const short8 x = *(const short8*)p;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __uint_as_float((uint32_t)(uint16_t)x[k] << 16);
  }
}

template <int GT, int PER>
__device__ __forceinline__ void store_chunk(char* base, int64_t elem, const float* v) {
  if constexpr (GT == MM_F32) {
    f32x4* o = (f32x4*)(base + elem * 4);
#pragma unroll
    for (int k = 0; k < PER / 4; ++k) o[k] = f32x4{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
  } else {
    static_assert(PER == 8, "16-bit gradients come from 16-bit rows");
    if constexpr (GT == MM_F16) {
      typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
      half8_t o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (_Float16)v[k];
      *(half8_t*)(base + elem * 2) = o;
    } else {
      short8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (short)(uint16_t)(__float_as_uint(round_like<MM_BF16>(v[k])) >> 16);
      *(short8*)(base + elem * 2) = o;
    }
  }
}

// GT: element type of the gradients — float32, or the token vectors' own 16-bit type (what autograd hands back to an
// fp16 / bf16 encoder: written once, rounded once; rounds 1-3 wrote fp32 into a memset buffer and cast afterwards:
// three launches and 2.5 x the bytes).  Every row of grad_d is written by this kernel (zeros where no query token's
// maximum sits): no memset in front of it.
template <int DT, int GT>
__global__ void __launch_bounds__(256) maxsim_bwd_kernel(const MaxsimBwdArgs a) {
  // One 4-wavefront workgroup per pair (rounds 1-3: one wavefront, ~100 dependent memory round trips in a row — 150 us per
  // launch whatever the batch): wavefront w takes document blocks w, w + 4, ... of the arg-max search, the four partial
  // (maximum, first position) results per query token meet in LDS, and the workgroup then writes the gradient rows as
  // 16-byte items (one thread per chunk of a row: every byte of grad_q / grad_d is written exactly once).
  extern __shared__ int smem_i[];
  int* jstar = smem_i;                       // [Q] first arg-max document position of every query token, -1 = no gradient
  float* pbest = (float*)(smem_i + a.Q);     // [4][32]
  int* prow = smem_i + a.Q + 128;            // [4][32]
  uint32_t* rmask = (uint32_t*)(smem_i + a.Q + 256);   // [D][ceil(Q / 32)] when a.row_masks
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int64_t pair = blockIdx.x;
  if (pair >= a.n_pairs) return;
  const int D = a.D, Q = a.Q, E = a.E;
  constexpr int ES = (DT == MM_F32) ? 4 : 2;
  constexpr int GS = (GT == MM_F32) ? 4 : 2;
  const int64_t rowb = (int64_t)E * ES;
  const int nblk_tot = (D + 31) >> 5;
  const int qwords = (Q + 31) >> 5;
  int len = a.dm.len ? a.dm.len[pair] : D;
  len = len < 0 ? 0 : (len > D ? D : len);
  const int nb = (len + 31) >> 5;
  const float fill = len < D ? -1000.0f : neg_inf();
  const int qlen = a.qm.len ? a.qm.len[pair] : Q;
  const char* dbase = (const char*)a.d + pair * D * rowb;
  const char* qbase = (const char*)a.q + pair * Q * rowb;
  const float g = a.go[pair];
  char* gq = (char*)a.gq + pair * Q * (int64_t)E * GS;
  char* gd = (char*)a.gd + pair * D * (int64_t)E * GS;
  if (a.row_masks)                                       // (ordered before the atomicOr below by the loop's barriers)
    for (int k = threadIdx.x; k < D * qwords; k += 256) rmask[k] = 0u;

  for (int n = 0; n < qwords; ++n) {
    const int qtok = 32 * n + r;
    const int qr = qtok < Q ? qtok : Q - 1;
    bool qvalid = qtok < Q && qtok < qlen;
    if (a.qm.bits) qvalid = qvalid && ((a.qm.bits[pair * qwords + n] >> r) & 1u);
    float m[16];
    int bt[16];  // block of the running maximum; -1 = a padded position / nothing yet
#pragma unroll
    for (int i = 0; i < 16; ++i) { m[i] = fill; bt[i] = -1; }
    for (int t = wv; t < nb; t += 4) {
      const int drow = 32 * t + r;
      const int dr = drow < D ? drow : D - 1;
      const f32x16 acc = dot_block<DT>(dbase + dr * rowb, qbase + qr * rowb, E);
      const int rem = len - 32 * t;
      const uint32_t ex = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
      const uint32_t va = a.dm.bits ? (a.dm.bits[pair * nblk_tot + t] & ex) : ex;
      const uint32_t exs = ex >> (4 * h), vas = va >> (4 * h);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int bit = rowof(i);
        const bool real = (vas >> bit) & 1u;
        const float v = real ? acc[i] : (((exs >> bit) & 1u) ? -1000.0f : fill);
        if (v > m[i]) { m[i] = v; bt[i] = real ? t : -1; }  // strict: the first block of this wavefront's sequence wins ties
      }
    }
    // first arg-max over this lane's 16 row classes, then over the two lane halves, then over the four wavefronts
    // (equal maxima: the smaller position wins = torch.max's first arg-max; a padded position carries no index)
    float best = m[0];
    int brow = bt[0] < 0 ? 0x7fffffff : 32 * bt[0] + rowof(0) + 4 * h;
#pragma unroll
    for (int i = 1; i < 16; ++i) {
      const int row = bt[i] < 0 ? 0x7fffffff : 32 * bt[i] + rowof(i) + 4 * h;
      if (m[i] > best || (m[i] == best && row < brow)) { best = m[i]; brow = row; }
    }
    const float ob = __shfl_xor(best, 32, 64);
    const int orow = __shfl_xor(brow, 32, 64);
    if (ob > best || (ob == best && orow < brow)) { best = ob; brow = orow; }
    __syncthreads();                                     // (the previous tile's partials have been read)
    if (h == 0) { pbest[wv * 32 + r] = best; prow[wv * 32 + r] = brow; }
    __syncthreads();
    if (wv == 0 && h == 0 && qtok < Q) {
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        const float ov = pbest[k * 32 + r];
        const int orw = prow[k * 32 + r];
        if (ov > best || (ov == best && orw < brow)) { best = ov; brow = orw; }
      }
      const int js = (qvalid && brow != 0x7fffffff) ? brow : -1;
      jstar[qtok] = js;
      if (a.row_masks && js >= 0) atomicOr(&rmask[js * qwords + n], 1u << r);
    }
  }
  __syncthreads();
  // Gradient rows as 16-byte items over the whole workgroup (rounds 1-4/1 wrote them element by element — 2 bytes a lane —
  // and found a document row's contributors by scanning the arg-max table once per query token).
  constexpr int PER = 16 / ES;
  const int nch = E / PER;
  for (int it = threadIdx.x; it < Q * nch; it += 256) {   // grad_q_i = g d_j*(i)  (0 without a gradient)
    const int qt = it / nch, c = it - qt * nch;
    const int j = jstar[qt];
    float v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] = 0.0f;
    if (j >= 0) {
      load_chunk<DT>(dbase + j * rowb + c * 16, v);
#pragma unroll
      for (int k = 0; k < PER; ++k) v[k] *= g;
    }
    store_chunk<GT, PER>(gq, (int64_t)qt * E + c * PER, v);
  }
  for (int it = threadIdx.x; it < D * nch; it += 256) {   // grad_d_j = g sum_{i: j*(i) = j} q_i: ascending i, fp32, written once
    const int j = it / nch, c = it - j * nch;
    float v[PER], x[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] = 0.0f;
    if (a.row_masks) {
      for (int n = 0; n < qwords; ++n) {
        uint32_t mk = rmask[j * qwords + n];
        while (mk) {
          const int b = __builtin_ctz(mk);
          mk &= mk - 1;
          load_chunk<DT>(qbase + (32 * n + b) * rowb + c * 16, x);
#pragma unroll
          for (int k = 0; k < PER; ++k) v[k] += x[k];
        }
      }
    } else {                                              // (document x query too large for the LDS masks: scan the table)
      for (int p = 0; p < Q; ++p)
        if (jstar[p] == j) {
          load_chunk<DT>(qbase + p * rowb + c * 16, x);
#pragma unroll
          for (int k = 0; k < PER; ++k) v[k] += x[k];
        }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] *= g;
    store_chunk<GT, PER>(gd, (int64_t)j * E + c * PER, v);
  }
}

// ---------------------------------------------------------------------------------------------
// Calibration: maxsim_stream_kernel's HBM read stream with the arithmetic taken out — the same launch geometry (one
// wavefront per workgroup, 4 per CU), the same 8-instruction LDS-DMA blocks into a 2-slot ring, the same counted waits;
// no fragment reads, no MFMA, no maximum.  What this box's memory system gives THIS access pattern: bench.py prints the
// headline kernel's rate as a fraction of it (a slow box and a slow kernel look different in that ratio).
// ---------------------------------------------------------------------------------------------
template <bool NT>
__global__ void __launch_bounds__(64) hbm_stream_probe_kernel(const char* src, int64_t n_blocks, int64_t blocks_per_wave,
                                                              uint32_t* sink) {
  constexpr int NBUF = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * blocks_per_wave;
  const int64_t b1 = (b0 + blocks_per_wave < n_blocks) ? b0 + blocks_per_wave : n_blocks;
  if (b0 >= b1) return;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  uint32_t voff[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int row = 4 * k + (lane >> 4);
    voff[k] = (uint32_t)(row * 256 + (((lane & 15) ^ (row & 15)) << 4));
  }
  int64_t pb = b0;
  int pbuf = 0, cbuf = 0, inflight = 0;
  uint32_t acc = 0;
  for (int64_t b = b0; b < b1; ++b) {
    while (pb < b1 && inflight < NBUF) {
      issue_block<NT>(src + pb * kBlkBytes, voff, lds0 + (uint32_t)pbuf * kBlkBytes);
      pbuf = (pbuf + 1 == NBUF) ? 0 : pbuf + 1;
      ++inflight;
      ++pb;
    }
    wait_block(inflight - 1);
    acc ^= *(const uint32_t*)(smem + cbuf * kBlkBytes + lane * 4);   // one dword per lane: the block did land
    cbuf = (cbuf + 1 == NBUF) ? 0 : cbuf + 1;
    --inflight;
  }
  if (sink && acc == 0x9e3779b9u) sink[blockIdx.x] = acc;            // (keeps the read alive; practically never taken)
}

// ---------------------------------------------------------------------------------------------
static int validate(const void* q, const void* d, float* out, int64_t n_pairs, int Q, int D, int E, int dtype) {
  if (!q || !d || !out) return set_error(MM_EINVAL, "maxsim: null tensor pointer");
  if (n_pairs < 0 || Q <= 0 || D <= 0 || E <= 0) return set_error(MM_EINVAL, "maxsim: non-positive shape");
  if (dtype != MM_F32 && dtype != MM_F16 && dtype != MM_BF16) return set_error(MM_EINVAL, "maxsim: bad dtype %d", dtype);
  const int per16 = dtype == MM_F32 ? 4 : 8;
  if (E % per16) return set_error(MM_EUNSUPPORTED, "maxsim: E=%d rows are not 16-byte multiples (pad E to a multiple of %d)", E, per16);
  if (((uintptr_t)q | (uintptr_t)d) & 15) return set_error(MM_EINVAL, "maxsim: q/d must be 16-byte aligned");
  return MM_OK;
}

static int validate_flags(int flags) {
  if (flags & ~(MM_SIM_ROUND | MM_SUM_ROUND)) return set_error(MM_EINVAL, "maxsim: unknown flags 0x%x", flags);
  return MM_OK;
}

// Measured on MI355X (profiles/r01_sweep_nbuf_wpc.log): one wavefront per SIMD (4 / CU) with a
// 2-slot ring is the fastest point (6.9 TB/s); more wavefronts or deeper rings only add contention.
// (defaults and their environment overrides: EnvCfg in mm_internal.h)
template <int DT, int NBUF, bool NT, int NSL, bool RAG, int NQT = 1>
static int launch_stream(const MaxsimArgs& a0, hipStream_t stream) {
  MaxsimArgs a = a0;
  const int lds = NBUF * kBlkBytes;
  int wpc = env().maxsim_wpc > 0 ? env().maxsim_wpc : (160 * 1024) / lds;
  if (wpc > 16) wpc = 16;
  int64_t waves = (int64_t)kCUs * wpc;
  if (waves > a.n_pairs) waves = a.n_pairs;
  a.pairs_per_wave = (a.n_pairs + waves - 1) / waves;
  waves = (a.n_pairs + a.pairs_per_wave - 1) / a.pairs_per_wave;
  if constexpr (NQT == 2 && !RAG) {      // (a query of <= 32 tokens in this layout is the pair kernel's, maxsim_pair.hip)
    if (a.dm64) {
      if (a.pairs_per_wave != 1) return set_error(MM_EINVAL, "maxsim: in-kernel masks need one pair per wavefront");
      if (a.n_pairs * 2 <= (int64_t)kCUs * 4 && a.D > 32 && !env().maxsim_no_wpp2) {
        hipLaunchKernelGGL((maxsim_stream_wpp2_kernel<DT, NBUF, NT, NSL, NQT>), dim3((unsigned)waves), dim3(128), 2 * lds + NQT * 256, stream, a);
        return check_launch("maxsim_stream_wpp2_kernel");
      }
      hipLaunchKernelGGL((maxsim_stream_kernel<DT, NBUF, NT, NSL, RAG, NQT, 0, true>), dim3((unsigned)waves), dim3(64), lds, stream, a);
      return check_launch("maxsim_stream_kernel<in-kernel masks>");
    }
  }
  hipLaunchKernelGGL((maxsim_stream_kernel<DT, NBUF, NT, NSL, RAG, NQT>), dim3((unsigned)waves), dim3(64), lds, stream, a);
  return check_launch("maxsim_stream_kernel");
}

// every wavefront of launch_stream<.., NBUF = 2, ..> gets at most one pair
static bool stream_one_pair_per_wave(int64_t n_pairs) {
  int wpc = env().maxsim_wpc > 0 ? env().maxsim_wpc : (160 * 1024) / (2 * kBlkBytes);
  if (wpc > 16) wpc = 16;
  return n_pairs <= (int64_t)kCUs * wpc;
}

template <int DT, int NSL>
static int launch_stream_inb(const MaxsimArgs& a0, hipStream_t stream) {
  MaxsimArgs a = a0;
  const int lds = 2 * kBlkBytes;
  int64_t waves = (int64_t)kCUs * 4;
  if (waves > a.n_pairs) waves = a.n_pairs;
  a.pairs_per_wave = (a.n_pairs + waves - 1) / waves;
  waves = (a.n_pairs + a.pairs_per_wave - 1) / a.pairs_per_wave;
  if (a.Q > 32) {
    hipLaunchKernelGGL((maxsim_stream_kernel<DT, 2, false, NSL, false, 2, true>), dim3((unsigned)waves), dim3(64), lds, stream, a);
    return check_launch("maxsim_stream_kernel<all pairs>");
  }
  hipLaunchKernelGGL((maxsim_stream_kernel<DT, 2, false, NSL, false, 1, true>), dim3((unsigned)waves), dim3(64), lds, stream, a);
  return check_launch("maxsim_stream_kernel<all pairs>");
}

// all pairs tiled over queries (INB = 2): NQT queries per wavefront, XCD-aware (document slice, query group) map
constexpr int kNotLaunched = 1;   // (not an MM_* code: those are <= 0)
template <int DT, int NSL, int NQT>
static int launch_stream_inb_tiled(const MaxsimArgs& a0, hipStream_t stream) {
  MaxsimArgs a = a0;
  const int lds = 2 * kBlkBytes;
  const int64_t G = (a.inb_bq + NQT - 1) / NQT;
  const int64_t target = (int64_t)kCUs * 4 / 8;                 // wavefronts per XCD
  a.inb_gw = (int)(G < target ? G : target);                    // query-group lanes per (XCD, slice)
  int64_t T = target / a.inb_gw;                                 // document slices per XCD
  const int64_t max_t = (a.inb_bd + 7) / 8;                      // >= 1 document per slice
  if (T > max_t) T = max_t;
  if (T < 1) T = 1;
  a.inb_t = (int)T;
  // the kernel indexes a wavefront's (query group, document) items with 32 bits
  if (((G + a.inb_gw - 1) / a.inb_gw) * ((a.inb_bd + 8 * T - 1) / (8 * T) + 1) >= (1LL << 31)) return kNotLaunched;
  const int64_t waves = 8 * T * a.inb_gw;
  hipLaunchKernelGGL((maxsim_allpairs_tiled_kernel<DT, NSL, NQT>), dim3((unsigned)waves), dim3(64), lds, stream, a);
  return check_launch("maxsim_allpairs_tiled_kernel");
}

// workgroup-shared ring (maxsim_allpairs_wg_kernel): 4 x NQT queries per workgroup, two workgroups per CU
template <int DT, int NSL, int NQT>
static int launch_inb_wg(const MaxsimArgs& a0, hipStream_t stream) {
  MaxsimArgs a = a0;
  const int lds = (NSL == 1 ? 3 : 4) * kBlkBytes;
  const int64_t G = (a.inb_bq + 4 * NQT - 1) / (4 * NQT);
  const int64_t target = (int64_t)kCUs * 2 / 8;                  // workgroups per XCD
  a.inb_gw = (int)(G < target ? G : target);
  int64_t T = target / a.inb_gw;
  const int64_t max_t = (a.inb_bd + 7) / 8;
  if (T > max_t) T = max_t;
  if (T < 1) T = 1;
  a.inb_t = (int)T;
  if (a.inb_bd >= (1LL << 31)) return kNotLaunched;
  hipLaunchKernelGGL((maxsim_allpairs_wg_kernel<DT, NSL, NQT>), dim3((unsigned)(8 * T * a.inb_gw)), dim3(256), lds, stream, a);
  return check_launch("maxsim_allpairs_wg_kernel");
}

template <int DT>
static int launch_stream_inb_cfg(const MaxsimArgs& a, hipStream_t stream) {
  // tiled over queries when the query tiles fit the register file (E <= 256), the masks are the documents' own
  // (bug-compatible masking, colbert.py:158, depends on the query index) and there is more than one query
  if (a.Q <= 32 && !a.inb_bug && a.inb_bq > 1 && !env().maxsim_inb_untiled) {
    int e = kNotLaunched;
    // teacher batches large enough to give every workgroup 16 queries and a few documents: the shared-ring kernel
    if (a.inb_bq >= 64 && a.inb_bd >= 64 && !env().maxsim_inb_nowg) {
      if (a.E == 128) e = launch_inb_wg<DT, 1, 4>(a, stream);
      if (a.E == 256) e = launch_inb_wg<DT, 2, 2>(a, stream);
    }
    if (e != kNotLaunched) return e;
    if (a.E == 128) e = launch_stream_inb_tiled<DT, 1, 4>(a, stream);
    if (a.E == 256) e = launch_stream_inb_tiled<DT, 2, 2>(a, stream);
    if (e != kNotLaunched) return e;      // index range too large for the tiled map: one query per wavefront below
  }
  switch (a.E / 128) {
    case 1: return launch_stream_inb<DT, 1>(a, stream);
    case 2: return launch_stream_inb<DT, 2>(a, stream);
    case 3: return launch_stream_inb<DT, 3>(a, stream);
    case 4: return launch_stream_inb<DT, 4>(a, stream);
    default: return launch_stream_inb<DT, 6>(a, stream);
  }
}

template <int DT, int NSL, bool RAG>
static int launch_stream_nsl(const MaxsimArgs& a, hipStream_t stream) {
  const bool nt = env().maxsim_nt != 0;
  if (a.Q > 32) return launch_stream<DT, 2, true, NSL, RAG, 2>(a, stream);  // two query tiles in registers
  if (NSL == 1 && !RAG) {  // the tuning knobs are only instantiated for the headline shape
    switch (env().maxsim_nbuf) {
      case 3: return nt ? launch_stream<DT, 3, true, NSL, false>(a, stream) : launch_stream<DT, 3, false, NSL, false>(a, stream);
      case 4: return nt ? launch_stream<DT, 4, true, NSL, false>(a, stream) : launch_stream<DT, 4, false, NSL, false>(a, stream);
      default: return nt ? launch_stream<DT, 2, true, NSL, false>(a, stream) : launch_stream<DT, 2, false, NSL, false>(a, stream);
    }
  }
  return launch_stream<DT, 2, true, NSL, RAG>(a, stream);
}

template <int DT, bool RAG>
static int launch_stream_cfg(const MaxsimArgs& a, hipStream_t stream) {
  switch (a.E / 128) {
    case 1: return launch_stream_nsl<DT, 1, RAG>(a, stream);
    case 2: return launch_stream_nsl<DT, 2, RAG>(a, stream);
    case 3: return launch_stream_nsl<DT, 3, RAG>(a, stream);
    case 4: return launch_stream_nsl<DT, 4, RAG>(a, stream);
    default: return launch_stream_nsl<DT, 6, RAG>(a, stream);
  }
}

static int launch_generic(const MaxsimArgs& a, int dtype, hipStream_t stream) {
  if (a.n_pairs > 0x7fffffffLL) return set_error(MM_EUNSUPPORTED, "maxsim: more than 2^31-1 pairs in one generic launch");
  const dim3 grid((unsigned)a.n_pairs), block(64);
  if (dtype == MM_F32)
    hipLaunchKernelGGL(maxsim_generic_kernel<MM_F32>, grid, block, 0, stream, a);
  else if (dtype == MM_F16)
    hipLaunchKernelGGL(maxsim_generic_kernel<MM_F16>, grid, block, 0, stream, a);
  else
    hipLaunchKernelGGL(maxsim_generic_kernel<MM_BF16>, grid, block, 0, stream, a);
  return check_launch("maxsim_generic_kernel");
}

}  // namespace mm

using namespace mm;

extern "C" size_t mm_maxsim_workspace_bytes(int64_t n_pairs, int64_t pairs_per_query, int Q, int D,
                                             int q_mask_kind, int d_mask_kind) {
  if (pairs_per_query <= 0) pairs_per_query = 1;
  const int64_t nq = (n_pairs + pairs_per_query - 1) / pairs_per_query;
  return packed_mask_bytes(q_mask_kind, nq, Q) + packed_mask_bytes(d_mask_kind, n_pairs, D);
}

extern "C" int mm_maxsim_fwd(const void* q, const void* d, const void* q_mask, int q_mask_kind,
                             const void* d_mask, int d_mask_kind, float* out, int64_t n_pairs,
                             int64_t pairs_per_query, int Q, int D, int E, int dtype, int flags, void* workspace,
                             size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int e = validate(q, d, out, n_pairs, Q, D, E, dtype)) return e;
  if (int e = validate_flags(flags)) return e;
  if (pairs_per_query <= 0) return set_error(MM_EINVAL, "maxsim: pairs_per_query must be >= 1");
  if (n_pairs == 0) return MM_OK;
  const int64_t nq = (n_pairs + pairs_per_query - 1) / pairs_per_query;
  MaxsimArgs a{};
  a.q = q; a.d = d; a.out = out; a.n_pairs = n_pairs; a.ppq = pairs_per_query; a.inb_bd = 0; a.inb_bug = 0;
  a.Q = Q; a.D = D; a.E = E; a.rnd = flags;
  char* ws = (char*)workspace;
  size_t left = workspace ? workspace_bytes : 0;
  // the reference's batch layout (one query tile per pair, eval.py:108): the pair kernel; HF int64 masks are read by
  // the kernel itself (no pack pass) when their rows are 16-byte multiples
  const bool pair_kernel = pairs_per_query == 1 && !env().maxsim_generic && maxsim_pair_supported(Q, E, dtype);
  if (pair_kernel && q_mask_kind == MM_MASK_I64 && d_mask_kind == MM_MASK_I64 && q_mask && d_mask &&
      maxsim_pair_i64_supported(Q, D, q_mask, d_mask)) {
    a.qm64 = (const int64_t*)q_mask;
    a.dm64 = (const int64_t*)d_mask;
    return maxsim_pair_launch(a, dtype, true, stream);
  }
  const bool stream_ok = !env().maxsim_generic && dtype != MM_F32 && Q <= 64 &&
                         (E == 128 || E == 256 || E == 384 || E == 512 || E == 768);
  // a long query (ColBERT's [MASK] augmentation: 30 + 8 tokens) in a call small enough that every wavefront scores one
  // pair: the two-tile streaming kernel reads the int64 masks itself, one launch per call
  if (stream_ok && !pair_kernel && Q > 32 && D <= 256 && q_mask_kind == MM_MASK_I64 && d_mask_kind == MM_MASK_I64 && q_mask &&
      d_mask && !env().maxsim_no_inline_masks && stream_one_pair_per_wave(n_pairs)) {
    a.qm64 = (const int64_t*)q_mask;
    a.dm64 = (const int64_t*)d_mask;
    return dtype == MM_BF16 ? launch_stream_cfg<MM_BF16, false>(a, stream) : launch_stream_cfg<MM_F16, false>(a, stream);
  }
  if (int e = resolve_mask_pair(q_mask, q_mask_kind, nq, Q, &a.qm, d_mask, d_mask_kind, n_pairs, D, &a.dm, &ws, &left, stream)) return e;
  if (pair_kernel) return maxsim_pair_launch(a, dtype, false, stream);
  if (stream_ok) return dtype == MM_BF16 ? launch_stream_cfg<MM_BF16, false>(a, stream) : launch_stream_cfg<MM_F16, false>(a, stream);
  // fp32 token vectors (ColBERT run with use_fp16 = False): the split-bf16 streaming kernel of kernel_pool128.hip
  // with the MaxSim epilogue
  if (!env().maxsim_generic && dtype == MM_F32 && kp128_maxsim_supported(Q, E)) {
    return kp128_maxsim_f32((const float*)q, (const float*)d, a.qm, a.dm, out, n_pairs, pairs_per_query, Q, D, E, stream);
  }
  return launch_generic(a, dtype, stream);
}

extern "C" size_t mm_maxsim_inbatch_workspace_bytes(int64_t Bq, int64_t Bd, int Q, int D, int q_mask_kind,
                                                     int d_mask_kind) {
  return packed_mask_bytes(q_mask_kind, Bq, Q) + packed_mask_bytes(d_mask_kind, Bd, D);
}

extern "C" int mm_maxsim_inbatch_fwd(const void* q, const void* d, const void* q_mask, int q_mask_kind,
                                     const void* d_mask, int d_mask_kind, float* out, int64_t Bq, int64_t Bd,
                                     int Q, int D, int E, int dtype, int bug_compatible, int flags, void* workspace,
                                     size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (Bq < 0 || Bd < 0) return set_error(MM_EINVAL, "maxsim_inbatch: negative batch");
  if (int e = validate_flags(flags)) return e;
  if (int e = validate(q, d, out, Bq * Bd, Q, D, E, dtype)) return e;
  if (bug_compatible && Bq != Bd)
    return set_error(MM_EINVAL, "maxsim_inbatch: bug_compatible masking (colbert.py:158) requires Bq == Bd (got %lld, %lld)",
                     (long long)Bq, (long long)Bd);
  if (Bq * Bd == 0) return MM_OK;
  MaxsimArgs a{};
  a.q = q; a.d = d; a.out = out; a.n_pairs = Bq * Bd; a.ppq = 1; a.inb_bd = Bd; a.inb_bq = Bq; a.inb_bug = bug_compatible ? 1 : 0;
  a.Q = Q; a.D = D; a.E = E; a.rnd = flags;
  char* ws = (char*)workspace;
  size_t left = workspace ? workspace_bytes : 0;
  if (int e = resolve_mask(q_mask, q_mask_kind, Bq, Q, &ws, &left, stream, &a.qm)) return e;
  if (int e = resolve_mask(d_mask, d_mask_kind, Bd, D, &ws, &left, stream, &a.dm)) return e;
  // the streaming kernel in all-pairs mode (query tile resident, documents of one query consecutive: no non-temporal
  // hint, every document is read once per query); Q > 64 and fp32 stay on the one-wavefront-per-pair kernel
  const bool stream_ok = !env().maxsim_generic && dtype != MM_F32 && Q <= 64 &&
                         (E == 128 || E == 256 || E == 384 || E == 512 || E == 768);
  if (stream_ok) {
    a.ppq = Bd;
    return dtype == MM_BF16 ? launch_stream_inb_cfg<MM_BF16>(a, stream) : launch_stream_inb_cfg<MM_F16>(a, stream);
  }
  return launch_generic(a, dtype, stream);
}

extern "C" size_t mm_maxsim_ragged_workspace_bytes(int64_t n_pairs, int64_t pairs_per_query, int Q, int q_mask_kind) {
  if (pairs_per_query <= 0) pairs_per_query = 1;
  return packed_mask_bytes(q_mask_kind, (n_pairs + pairs_per_query - 1) / pairs_per_query, Q);
}

extern "C" int mm_maxsim_ragged_fwd(const void* q, const void* tokens, const int64_t* doc_begin, const int64_t* doc_end,
                                    const void* q_mask, int q_mask_kind, float* out, int64_t n_pairs,
                                    int64_t pairs_per_query, int Q, int E, int dtype, int flags, void* workspace,
                                    size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int e = validate(q, tokens, out, n_pairs, Q, 1, E, dtype)) return e;
  if (int e = validate_flags(flags)) return e;
  if (!doc_begin || !doc_end) return set_error(MM_EINVAL, "maxsim_ragged: null document range pointer");
  if (pairs_per_query <= 0) return set_error(MM_EINVAL, "maxsim_ragged: pairs_per_query must be >= 1");
  if (n_pairs == 0) return MM_OK;
  const int64_t nq = (n_pairs + pairs_per_query - 1) / pairs_per_query;
  MaxsimArgs a{};
  a.q = q; a.d = tokens; a.out = out; a.n_pairs = n_pairs; a.ppq = pairs_per_query;
  a.Q = Q; a.D = 32; a.E = E; a.rag_begin = doc_begin; a.rag_end = doc_end; a.rnd = flags;
  char* ws = (char*)workspace;
  size_t left = workspace ? workspace_bytes : 0;
  if (int e = resolve_mask(q_mask, q_mask_kind, nq, Q, &ws, &left, stream, &a.qm)) return e;
  const bool stream_ok = !env().maxsim_generic && dtype != MM_F32 && Q <= 64 &&
                         (E == 128 || E == 256 || E == 384 || E == 512 || E == 768);
  if (stream_ok) return dtype == MM_BF16 ? launch_stream_cfg<MM_BF16, true>(a, stream) : launch_stream_cfg<MM_F16, true>(a, stream);
  return launch_generic(a, dtype, stream);
}

extern "C" int mm_hbm_stream_probe(const void* src, int64_t bytes, int nt, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!src || bytes <= 0 || (bytes % kBlkBytes) || ((uintptr_t)src & 15))
    return set_error(MM_EINVAL, "hbm_stream_probe: need a 16-byte aligned buffer of a multiple of %d bytes", kBlkBytes);
  const int64_t n_blocks = bytes / kBlkBytes;
  int64_t waves = (int64_t)kCUs * 4;
  if (waves > n_blocks) waves = n_blocks;
  const int64_t per = (n_blocks + waves - 1) / waves;
  waves = (n_blocks + per - 1) / per;
  if (nt)
    hipLaunchKernelGGL(hbm_stream_probe_kernel<true>, dim3((unsigned)waves), dim3(64), 2 * kBlkBytes, stream, (const char*)src, n_blocks, per, (uint32_t*)nullptr);
  else
    hipLaunchKernelGGL(hbm_stream_probe_kernel<false>, dim3((unsigned)waves), dim3(64), 2 * kBlkBytes, stream, (const char*)src, n_blocks, per, (uint32_t*)nullptr);
  return check_launch("hbm_stream_probe_kernel");
}

extern "C" size_t mm_maxsim_bwd_workspace_bytes(int64_t n_pairs, int Q, int D, int q_mask_kind, int d_mask_kind) {
  return packed_mask_bytes(q_mask_kind, n_pairs, Q) + packed_mask_bytes(d_mask_kind, n_pairs, D);
}

extern "C" int mm_maxsim_bwd(const void* q, const void* d, const void* q_mask, int q_mask_kind, const void* d_mask,
                             int d_mask_kind, const float* grad_out, void* grad_q, void* grad_d, int grad_dtype, int64_t n_pairs,
                             int Q, int D, int E, int dtype, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!grad_out || !grad_q || !grad_d) return set_error(MM_EINVAL, "maxsim_bwd: null gradient pointer");
  if (int e = validate(q, d, (float*)grad_q, n_pairs, Q, D, E, dtype)) return e;
  if (grad_dtype != MM_F32 && grad_dtype != dtype)
    return set_error(MM_EINVAL, "maxsim_bwd: gradients are float32 or have the token vectors' own type (got %d for %d)", grad_dtype, dtype);
  if (n_pairs == 0) return MM_OK;
  if (n_pairs > 0x7fffffffLL) return set_error(MM_EUNSUPPORTED, "maxsim_bwd: more than 2^31-1 pairs in one launch");
  if ((size_t)Q * 4 > 48 * 1024) return set_error(MM_EUNSUPPORTED, "maxsim_bwd: Q = %d query tokens exceed the arg-max table", Q);
  MaxsimBwdArgs a{};
  a.q = q; a.d = d; a.go = grad_out; a.gq = grad_q; a.gd = grad_d; a.n_pairs = n_pairs; a.Q = Q; a.D = D; a.E = E;
  char* ws = (char*)workspace;
  size_t left = workspace ? workspace_bytes : 0;
  if (int e = resolve_mask_pair(q_mask, q_mask_kind, n_pairs, Q, &a.qm, d_mask, d_mask_kind, n_pairs, D, &a.dm, &ws, &left, stream)) return e;
  if (((uintptr_t)grad_q | (uintptr_t)grad_d) & 15) return set_error(MM_EINVAL, "maxsim_bwd: gradients must be 16-byte aligned");
  const dim3 grid((unsigned)n_pairs), block(256);
  size_t lds = (size_t)Q * 4 + 2 * 128 * 4;
  const size_t mask_bytes = (size_t)D * ((Q + 31) / 32) * 4;
  a.row_masks = lds + mask_bytes <= 60 * 1024;
  if (a.row_masks) lds += mask_bytes;
  if (dtype == MM_F32)
    hipLaunchKernelGGL((maxsim_bwd_kernel<MM_F32, MM_F32>), grid, block, lds, stream, a);
  else if (dtype == MM_F16 && grad_dtype == MM_F32)
    hipLaunchKernelGGL((maxsim_bwd_kernel<MM_F16, MM_F32>), grid, block, lds, stream, a);
  else if (dtype == MM_F16)
    hipLaunchKernelGGL((maxsim_bwd_kernel<MM_F16, MM_F16>), grid, block, lds, stream, a);
  else if (grad_dtype == MM_F32)
    hipLaunchKernelGGL((maxsim_bwd_kernel<MM_BF16, MM_F32>), grid, block, lds, stream, a);
  else
    hipLaunchKernelGGL((maxsim_bwd_kernel<MM_BF16, MM_BF16>), grid, block, lds, stream, a);
  return check_launch("maxsim_bwd_kernel");
}
