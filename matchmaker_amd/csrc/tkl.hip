// TKL (long-document TK) scoring for MI355X (gfx950): everything of TKL_sigir20.forward after the
// chunk contextualiser — matchmaker/models/published/sigir20_tkl.py:180-286.
//
// Stage 1 (kernel_pool.hip, TKL mode): per packed chunk, cosine match of the query against the 40
//   centre tokens, K RBF kernels, document mask (:184-194); emitted as pair sums
//   ps[p][u][q][0..K-1] = act[2u] + act[2u+1] and ps[..][K] = number of those two positions whose
//   activation is non-zero.  This replaces the reference's [P,Q,40,K] tensor, its zero-fill +
//   boolean scatter (:196-197) and the transposing reshape (:199): dropped chunks are simply absent
//   and read as zeros through the slot -> packed-index map.
// Stage 2 (tkl_window_kernel): sliding windows of 30 positions, stride 2 (:209) = 15 consecutive
//   pair sums; window lengths (:210), kernel sums (:211), saturation ("embedding" :224-234 or
//   "log" :245-246), query mask and empty-window factor (:248), sum over query tokens (:249),
//   dense layer (:251-252)  ->  win_scores [B, W].
// Stage 3 (tkl_region_kernel): 0 -> -9900 (:257), three arg-max rounds with +-15 suppression
//   (:268-273), the peaks' +-1 / +-2 neighbours (:276-282), chunk_scoring dot (:286)  ->  out [B].
#include "mm_internal.h"

namespace mm {

constexpr int kU = 20;        // position pairs per chunk
constexpr int kWinPairs = 15; // 30 positions, stride 2
constexpr int kWT = 64;       // windows per workgroup in stage 2 (pair rows staged: kWT + 14)
constexpr int kWThreads = 256;
#ifndef MM_TKL_TOK_GROUP
#define MM_TKL_TOK_GROUP 10     // -D overrides for A/B builds only (profiles/r06_experiments/tkl_window_token_groups.txt)
#endif
constexpr int kTokGroup = MM_TKL_TOK_GROUP;  // query tokens per window workgroup (see tkl_window_kernel)
#ifndef MM_TKL_EPILOGUE_ORDER
#define MM_TKL_EPILOGUE_ORDER 1   // 0: round 4 (relaxed counter + s_waitcnt), 1: acq_rel counter by thread 0, 2: fences in every thread
#endif
constexpr int kRThreads = 256;  // region kernel (1,024 threads, one window each, measured slower: 11.3 vs 9.8-10.5 us)

__device__ __forceinline__ void region_topk(float* orig, float* work, float* rv, int* ri, int Wp, const float* __restrict__ prm,
                                            float* __restrict__ out, int32_t* __restrict__ peaks, int tid);   // (defined below the window kernel)


// Preparation in ONE launch (round 1: memset + emb + two mask packs + slot map = five, ~25 us of a 0.3 ms call) —
// tkl_prep_kernel, four roles by block range, all independent of each other:
//   [0, n_fill)        slot2p[...] = -1 (dropped chunk) for the B*C slots.  The entries of the packed chunks,
//                      slot2p[slot] = (packed chunk index << 2) | number of 32-row blocks stage 1 writes for it (0..2),
//                      are published by the stage-1 kernel itself (KpArgs::slot2p), later in the stream.  Stage 1 only
//                      writes the blocks below a chunk's effective length, so pair rows past them are taken as zeros
//                      by stage 2 instead of being zero-filled in HBM first (that memset of the whole pair buffer was
//                      255 MB at B = 256 x 2048 tokens);
//   next n_emb         sat_emb_reduce1(q_ctx) (:224): one wavefront per (document, query token) -> emb[b][i];
//   next n_q           effective query lengths + validity bits of the float query masks (one wavefront per row);
//   the rest           one wavefront per packed chunk: effective length + validity bits of its 40 centre tokens
//                      (columns 5..44 of the [P, 50] mask).
__global__ void __launch_bounds__(256) tkl_prep_kernel(int32_t* __restrict__ slot2p, int64_t BC, int n_fill,
                                                       const float* __restrict__ q_ctx, const float* __restrict__ prm,
                                                       float* __restrict__ emb, int64_t BQ, int E, int n_emb,
                                                       const float* __restrict__ q_mask, int64_t B, int Q, int n_q,
                                                       int32_t* __restrict__ qlen_out, uint32_t* __restrict__ qbits_out,
                                                       const float* __restrict__ chunk_mask, int64_t P,
                                                       int32_t* __restrict__ clen_out, uint32_t* __restrict__ cbits_out,
                                                       const int32_t* __restrict__ chunk_slot, int C, int32_t* __restrict__ ntile,
                                                       int32_t* __restrict__ done_cnt) {
  const int lane = threadIdx.x & 63;
  int blk = blockIdx.x;
  if (blk < n_fill) {
    const int64_t i = (int64_t)blk * 256 + threadIdx.x;
    if (i < BC) slot2p[i] = -1;
    // Per document (the first B threads of this role): the arrival counter of the window kernel's last-workgroup epilogue; the
    // live window tiles come from the chunk role below (without packed chunks: 0 for every document, here).
    if (i < B) {
      done_cnt[i] = 0;
      if (P == 0) ntile[i] = 0;
    }
    return;
  }
  blk -= n_fill;
  if (blk < n_emb) {
    const int64_t row = (int64_t)blk * 4 + (threadIdx.x >> 6);
    if (row >= BQ) return;
    const float* qr = q_ctx + row * E;
    float s = 0.0f;
    for (int e = lane; e < E; e += 64) s += qr[e] * prm[TklParams::emb() + e];
    s = wave_sum(s);
    if (lane == 0) emb[row] = s;
    return;
  }
  blk -= n_emb;
  if (blk < n_q) {
    const int64_t row = (int64_t)blk * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const int words = (Q + 31) >> 5;
    const float* m = q_mask + row * Q;
    int last = 0;
    for (int base = 0; base < Q; base += 64) {
      const int j = base + lane;
      const unsigned long long bal = __ballot(j < Q && m[j < Q ? j : Q - 1] != 0.0f);
      if (lane == 0) {
        const int w = base >> 5;
        qbits_out[row * words + w] = (uint32_t)bal;
        if (w + 1 < words) qbits_out[row * words + w + 1] = (uint32_t)(bal >> 32);
      }
      if (bal) last = base + 64 - __builtin_clzll(bal);
    }
    if (lane == 0) qlen_out[row] = last;
    return;
  }
  blk -= n_q;
  const int64_t p = (int64_t)blk * 4 + (threadIdx.x >> 6);
  if (p >= P) return;
  const float* m = chunk_mask + p * 50 + 5;
  const unsigned long long bal = __ballot(lane < 40 && m[lane < 40 ? lane : 39] != 0.0f);
  if (lane == 0) {
    clen_out[p] = bal ? 64 - __builtin_clzll(bal) : 0;
    cbits_out[p * 2] = (uint32_t)bal;
    cbits_out[p * 2 + 1] = (uint32_t)(bal >> 32);
    // Live window tiles per document: the LAST kept chunk c of a document bounds them (windows w <= 20 c + 19 touch chunk c).
    // chunk_slot is ascending, so the wavefront of the last chunk of a document sees the next document in chunk_slot[p + 1] and
    // writes the entry — and zeros for the documents without chunks between the two (and in front of the first, behind the last
    // chunk): every entry is written on every call, by exactly one wavefront.  (A binary search per document in the fill role,
    // thirteen dependent loads, was this launch's critical path: 5.7 us.)
    const int64_t nb = B;
    int64_t sl = chunk_slot[p];
    sl = sl < 0 ? 0 : (sl >= nb * C ? nb * C - 1 : sl);
    const int64_t b = sl / C;
    int64_t bn = nb;                                           // document of the next chunk (B behind the last)
    if (p + 1 < P) {
      int64_t sn = chunk_slot[p + 1];
      sn = sn < 0 ? 0 : (sn >= nb * C ? nb * C - 1 : sn);
      bn = sn / C;
    }
    if (p == 0)
      for (int64_t d = 0; d < b; ++d) ntile[d] = 0;
    if (bn != b) {
      ntile[b] = (kU * (int)(sl - b * C) + kU - 1) / kWT + 1;
      for (int64_t d = b + 1; d < bn; ++d) ntile[d] = 0;
    }
  }
}

// The (window, query token) items of one tile: see tkl_window_kernel.  red[w][i] for i < ql is written for every
// window of the tile (0 for windows past W).
template <int SAT, int kG, bool ONCE>
__device__ __forceinline__ void window_items(const float* tile, float* red, const float* emb, const float* __restrict__ q_mask,
                                             const float* __restrict__ prm, const float* sp, int b, int Q, int ql, int rowf,
                                             int w0, int W, int tid, int wt) {
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  constexpr int kE = kG - 1;                              // edge rows on each side
  for (int item = tid; item < (wt / kG) * ql; item += kWThreads) {
    const int wp = item / ql, i = item - wp * ql;
    const int wl = kG * wp;
    if (w0 + wl >= W) {
#pragma unroll
      for (int which = 0; which < kG; ++which) red[(wl + which) * (ql | 1) + i] = 0.0f;
      continue;
    }
    auto row = [&](int j, f32x2 (&dst)[kKC / 2]) {
      const f32x4* src = (const f32x4*)(tile + (size_t)(wl + j) * rowf + i * kKC);
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        const f32x4 x = src[v];
        dst[2 * v] = f32x2{x[0], x[1]};
        dst[2 * v + 1] = f32x2{x[2], x[3]};
      }
    };
    // rows 0..kE-1 and 15..15+kE-1 are the edges, rows kE..14 the core every window of the item contains
    f32x2 core[kKC / 2], edge[2 * kE + 1][kKC / 2], tmp[kKC / 2];
#pragma unroll
    for (int e = 0; e < kE; ++e) row(e, edge[e]);
    row(kE, core);
#pragma unroll
    for (int j = kE + 1; j < kWinPairs; ++j) {
      row(j, tmp);
#pragma unroll
      for (int k = 0; k < kKC / 2; ++k) core[k] += tmp[k];
    }
#pragma unroll
    for (int e = 0; e < kE; ++e) row(kWinPairs + e, edge[kE + e]);
    const float qmv = q_mask[(int64_t)b * Q + i];
#pragma unroll
    for (int which = 0; which < kG; ++which) {
      // window wl + which = head edges which..kE-1 + core + tail edges 0..which-1
      f32x2 acc2[kKC / 2];
#pragma unroll
      for (int k = 0; k < kKC / 2; ++k) {
        f32x2 v = core[k];
#pragma unroll
        for (int e = 0; e < 2 * kE; ++e)
          if ((e < kE && e >= which) || (e >= kE && e < kE + which)) v += edge[e][k];
        acc2[k] = v;
      }
      float pk[kKC];
#pragma unroll
      for (int k = 0; k < kKC; ++k) pk[k] = acc2[k >> 1][k & 1];
      float val = 0.0f;
      const float len = pk[kK];                                        // :210 (exact small integer)
      const float factor = qmv * (len > 0.0f ? 1.0f : 0.0f);          // :248
      if (SAT == MM_TKL_SAT_EMBEDDING) {
        const float x0 = emb[i], x1 = len;                             // :224-225
        const float mean = (x0 + x1) * 0.5f;                           // LayerNorm(2) :228
        const float d0 = x0 - mean, d1 = x1 - mean;
        // v_rsq_f32 / v_rcp_f32 (1 ulp) instead of the IEEE sqrt + two IEEE divisions (~35 instructions per window in a
        // VALU-bound kernel): 1e-7 relative on s1 .. s3, two orders below the window error of the fp32 evaluation itself
        const float rstd = __builtin_amdgcn_rsqf((d0 * d0 + d1 * d1) * 0.5f + 1e-5f);
        const float n0 = d0 * rstd * sp[9] + sp[11], n1 = d1 * rstd * sp[10] + sp[12];
        const float s1 = n0 * sp[0] + n1 * sp[1] + sp[2];              // :230
        const float s2 = __builtin_amdgcn_rcpf(n0 * sp[3] + n1 * sp[4] + sp[5]);   // :231
        const float s3 = n0 * sp[6] + n1 * sp[7] + sp[8];              // :232
#pragma unroll
        for (int k = 0; k < kK; ++k) {
          // x^s2 = exp2(s2 * log2 x) on the hardware transcendentals (x >= 1e-10 > 0): 3 instructions
          // instead of ~80 for powf; |s2 * log2 x| <= ~33 |s2| keeps the error ~1e-6 relative
          const float xp = __builtin_amdgcn_exp2f(s2 * __builtin_amdgcn_logf(fmaxf(pk[k], 1e-10f)));
          const float sat = s1 * xp - s3;  // :234
          // (folding this to factor (s1 sum_k w_k x_k^s2 - s3 sum_k w_k) saves two operations per kernel and was tried:
          // with the reference's biases of 100 in s1 and s3 the two sums cancel to ~1 % of their size, and the window
          // error against fp64 grew from 9e-7 to 7e-6 — more than the reference's own fp32 evaluation makes)
          val += prm[TklParams::dense() + k] * sat;
        }
      } else {
#pragma unroll
        for (int k = 0; k < kK; ++k) {
          const float sat = logf(fmaxf(pk[k] * prm[TklParams::kmult() + k], 1e-10f));   // :246
          val += prm[TklParams::dense() + k] * sat;
        }
      }
      // :248 the query mask and the empty-window factor once per window instead of once per kernel: factor is 0 or 1 for the
      // reference's {0, 1} masks, and x * 1 is exact (eleven multiplies less per window)
      val *= factor;
      red[(wl + which) * (ql | 1) + i] = (w0 + wl + which < W) ? val : 0.0f;
    }
    if (ONCE) break;      // the caller knows the 512 threads cover the items in one pass (no loop-carried registers)
  }
}

// LDS bytes of one pass over `wt` windows of a query of effective length ql: the wt + 14 pair rows, ql x 12 floats each
// (rows are stored compactly: tokens past the effective length do not exist), sat_emb_reduce1, the per-(window, token)
// values, the chunk lookups.
__host__ __device__ inline size_t window_pass_bytes(int wt, int ql) {
  return ((size_t)(wt + kWinPairs - 1) * ql * kKC + ((ql + 3) & ~3) + (size_t)wt * (ql | 1)) * 4 + 32;
}
#ifndef MM_WIN_LDS
#define MM_WIN_LDS (40 * 1024 - 512)      // -D overrides for A/B builds only
#endif
constexpr int kWindowLds = MM_WIN_LDS;     // four workgroups per CU (a 64-window tile of 10 tokens: 40,336 B)

// One workgroup = kWT consecutive windows of one document, in passes of wt = 64 / 32 / 16 windows — the largest that
// fits the workgroup's LDS at the document's effective query length.
// COS: `ps` holds stage 1's scaled, masked cosines (KpArgs::cos_out: per document, rows of ql real tokens) and the pair
// rows of the tile are EVALUATED here while they are staged (RBF kernels of the two positions of a pair + the
// non-zero count, the arithmetic of kernel_pool.hip's tkl_block_run), instead of being read back from a 6x larger
// pair-sum buffer.
template <int SAT, bool COS>
__global__ void __launch_bounds__(kWThreads, 4) tkl_window_kernel(const float* __restrict__ ps, const int32_t* __restrict__ slot2p,
                                                         const float* __restrict__ emb_g,
                                                         const float* __restrict__ q_mask,
                                                         const int32_t* __restrict__ q_len,
                                                         const float* __restrict__ prm, float* win,
                                                         int C, int Q, int W, int lds_bytes, const int32_t* __restrict__ ntile,
                                                         int n_planes, float* win_final, float* __restrict__ out,
                                                         int32_t* __restrict__ done_cnt, int32_t* __restrict__ peaks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float rv[kRThreads / 64];
  __shared__ int ri[kRThreads / 64];
  __shared__ int last_flag;
  const int b = blockIdx.y;
  const int w0 = blockIdx.x * kWT;
  const int tid = threadIdx.x;
  // ---- last-workgroup epilogue (MM_TKL_FOLD_REGIONS=1 only; round 4's default: the region top-k of :254-286 without its own launch) ----
  // The "last block" reduction pattern, written to the HIP memory model (round 5; round 4 relied on gfx950's sc1 write-through
  // behaviour alone).  Every workgroup that scores a live tile of document b
  //   1. publishes its 64 partial window scores (agent-scope atomic stores: sc1 write-through),
  //   2. meets at the workgroup barrier (workgroup-scope release / acquire between its threads),
  //   3. thread 0 arrives on done_cnt[b] with an ACQ_REL agent-scope fetch_add: the release half (buffer_wbl2 sc1 + s_waitcnt)
  //      covers the whole workgroup's scores through the barrier (happens-before is transitive across scopes).
  // The workgroup whose arrival completes the document (old + 1 == expected) has, through thread 0's acquire half
  // (buffer_inv sc1) and the second barrier, every other workgroup's scores ordered before its own loads (which are agent-scope
  // atomic loads besides: no stale line of another XCD's L2), sums the planes and runs the region search in the LDS its tile
  // just left.  (A release fence in EVERY thread, round 5's first form, cost +0.19 ms per 256-document call — a buffer_wbl2
  // from every wavefront; MM_TKL_EPILOGUE_ORDER=2 keeps it for A/B.)  expected = live tiles x live token groups
  // of the document — dead tiles (past the last kept chunk, tkl_prep_kernel) and groups without a real token neither write
  // nor arrive.  out == nullptr (the DEFAULT since round 5, see mm_tkl_fwd_peaks): the standalone tkl_region_kernel follows and this kernel only
  // publishes.  win / win_final are NOT __restrict__: with one token group (Q <= 10) they are the same buffer (each thread
  // reads and writes only its own windows there).
  const float* const part = win;
  const int64_t plane_stride = (int64_t)gridDim.y * W;
  const int nt_live = ntile[b] < (int)gridDim.x ? ntile[b] : (int)gridDim.x;
  int np_live = n_planes;
  {
    int qa = q_len ? q_len[b] : Q;
    qa = qa < 0 ? 0 : (qa > Q ? Q : qa);
    const int npq = (qa + kTokGroup - 1) / kTokGroup;
    np_live = npq < n_planes ? npq : n_planes;
  }
  const int expected = nt_live * np_live;
  auto finalize = [&]() {
    const int Wp = W < 3 ? 3 : W;
    float* orig = (float*)smem;
    float* work = orig + Wp;
    const int live_w = nt_live * kWT;
    for (int w = tid; w < Wp; w += kWThreads) {
      float s = 0.0f;
      if (w < W) {
        if (w < live_w)
          for (int g = 0; g < np_live; ++g)
            s += __hip_atomic_load(part + g * plane_stride + (int64_t)b * W + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        win_final[(int64_t)b * W + w] = s;
      }
      if (s == 0.0f) s = -9900.0f;                                     // :257
      orig[w] = s;
      work[w] = s;
    }
    __syncthreads();
    region_topk(orig, work, rv, ri, Wp, prm, out + b, peaks ? peaks + 3 * (int64_t)b : nullptr, tid);
  };
  auto arrive = [&]() {
    if (!out) return;
#if MM_TKL_EPILOGUE_ORDER == 2
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                 // (A/B: a release fence in EVERY thread: +0.19 ms at 256 documents)
#elif MM_TKL_EPILOGUE_ORDER == 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // (A/B: round 4's hardware-level ordering, not the memory model's)
#endif
    __syncthreads();   // workgroup-scope release / acquire: every thread's published scores happen-before thread 0's RMW below
    if (tid == 0) {
#if MM_TKL_EPILOGUE_ORDER == 0
      const int old = __hip_atomic_fetch_add(done_cnt + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
      // ACQ_REL at agent scope: release = the workgroup's scores (cumulative over the barrier) are visible device-wide before
      // the count; acquire = the finalizer sees the scores of every workgroup that counted before it
      const int old = __hip_atomic_fetch_add(done_cnt + b, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#endif
      last_flag = old + 1 == expected;
    }
    __syncthreads();   // ... and thread 0's acquire happens-before every thread's loads in finalize()
    if (last_flag) {
#if MM_TKL_EPILOGUE_ORDER == 2
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
      finalize();
    }
  };
  if (out && expected == 0) {            // no live tile or no real query token: every window is empty, the score is 0 (:257, :282)
    if (blockIdx.x == 0 && blockIdx.z == 0) {
      for (int w = tid; w < W; w += kWThreads) win_final[(int64_t)b * W + w] = 0.0f;
      if (tid == 0) out[b] = 0.0f;
      // every window is -9900: the arg-max rounds pick 0, then the first index outside +-15 of it, and so on (:268-273)
      if (peaks && tid < 3) { const int Wp = W < 3 ? 3 : W; peaks[3 * (int64_t)b + tid] = 15 * tid < Wp ? 15 * tid : 0; }
    }
    return;
  }
  // Token groups: workgroup z evaluates query tokens [kTokGroup z, kTokGroup (z + 1)) of its 64 windows and writes a partial
  // window score (summed over ITS tokens) to plane z of `win`; the region kernel adds the planes in order.  A tile of
  // 10 tokens is 37 KiB of LDS instead of 75: FOUR independent 256-thread workgroups per CU instead of two of 512 —
  // the kernel is a chain of latencies (lookups, cosine rows, barrier, evaluation, barrier), and the number of tiles
  // in flight per CU is what hides them.
  const int t0 = blockIdx.z * kTokGroup;
  win += (int64_t)blockIdx.z * gridDim.y * W;
  q_mask += t0;
  int ql = q_len ? q_len[b] : Q;                      // effective query length (rows of later tokens do not exist)
  ql = ql < 0 ? 0 : (ql > Q ? Q : ql);
  const int qfull = ql;                               // row stride of this document's cosine rows (stage 1 stores real tokens only)
  ql = ql - t0 < kTokGroup ? ql - t0 : kTokGroup;     // this group's tokens (local index i = token t0 + i)
  if (ql <= 0) return;                                // no token of this group is real: the region kernel skips the plane
  // tiles past the document's last kept chunk are empty (half of all tiles at config 3's U{50..2048} lengths): out after
  // one scalar load instead of after the lookup barrier (tkl_prep_kernel's hint, see there)
  if ((int)blockIdx.x >= nt_live) return;              // (nothing to publish: the summing workgroup takes these windows as zeros)
  int wt = kWT;
  while (wt > 4 && window_pass_bytes(wt, ql) > (size_t)lds_bytes) wt >>= 1;
  const int nu = wt + kWinPairs - 1;                  // pair rows needed by one pass
  const int rowf = ql * kKC;                          // floats per staged pair row
  const int srcf = Q * kKC;                           // floats per pair row in the pair-sum buffer
  int* cinfo = (int*)smem;                            // slot2p entries of the (<= 5) chunks this tile touches
  float* tile = (float*)(smem + 32);                  // [nu][ql][12]
  float* emb = tile + (size_t)nu * rowf;              // [ql]  sat_emb_reduce1(q_ctx)  (:224)
  float* red = emb + ((ql + 3) & ~3);                 // [wt][ql | 1] per-(window, query token) dense-weighted value

  // the chunk lookups first, once per tile, so that the row loads below are independent of each other
  const int c0 = w0 / kU;
  if (tid < 8) {
    const int c = c0 + tid;
    cinfo[tid] = c < C ? slot2p[(int64_t)b * C + c] : -1;
  }
  // COS: the cosine rows are indexed by chunk slot, so their addresses do not depend on the lookups: the first batch of
  // the first pass is requested BEFORE the barrier and both latencies run in parallel (a row of a dropped chunk was
  // never written; its values are discarded below)
  constexpr int kStageC = 4;
  float ca[kStageC], cb[kStageC];
  int cl[kStageC];
  auto issue_cos = [&](int ws_, int base_) {
    const int items_ = nu * ql;
#pragma unroll
    for (int s = 0; s < kStageC; ++s) {
      const int idx = base_ + kWThreads * s;
      cl[s] = -1;
      ca[s] = cb[s] = 1.0e5f;
      if (idx < items_) {
        const int j = idx / ql, i = idx - j * ql;
        const int ug = ws_ + j;
        const int c = ug / kU, uu = ug - c * kU;
        if (c < C) {
          const float* src = ps + (int64_t)b * C * 40 * Q + ((int64_t)c * 40 + 2 * uu) * qfull + t0 + i;
          ca[s] = src[0];
          cb[s] = src[qfull];
          cl[s] = c - c0;                                                // <= (19 + kWT + 14) / 20 < 8
        }
      }
    }
  };
  if constexpr (COS) issue_cos(w0, tid);
  __syncthreads();
  // a tile none of whose chunks exists (padding past the document's end) is all zeros: every window is empty and
  // scores exactly 0 (:248) — half of all tiles with config 3's U{50..2048} document lengths; so is every window of
  // a query without a real token
  {
    bool live = false;
    const int cb = (w0 + kWT + kWinPairs - 2) / kU - c0;
#pragma unroll
    for (int c = 0; c < 8; ++c) live = live || (c <= cb && cinfo[c] >= 0);
    if (!live) {                                       // (a hole: below the last kept chunk, so it is counted as a live tile)
      if (tid < kWT && w0 + tid < W)
        __hip_atomic_store(win + (int64_t)b * W + w0 + tid, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      arrive();
      return;
    }
  }
  if (SAT == MM_TKL_SAT_EMBEDDING && tid < ql) emb[tid] = emb_g[(int64_t)b * Q + t0 + tid];
  const float* sp = prm + TklParams::sat();
  for (int ws = w0; ws < w0 + kWT && ws < W; ws += wt) {
    // the thread index is re-materialised per pass: with everything derived from it loop-invariant the compiler
    // hoisted the index arithmetic of all passes' phases and kept 175-198 registers live (85 for a single pass);
    // two workgroups per CU need <= 128
    int tz;
    asm volatile("v_mov_b32 %0, %1" : "=v"(tz) : "v"(tid) : "memory");
    // ---- stage the pair-sum rows (zeros for dropped chunks) --------------------------------------
    // All row loads of a batch of kStage elements are issued before the first LDS store, so a thread pays
    // one memory latency per batch (a tile of Q = 20 is ONE batch); the kernel was 67 % s_waitcnt when every
    // element did its own dependent slot -> chunk -> row chain.
    if constexpr (COS) {
      typedef __attribute__((ext_vector_type(2))) float f32x2;
      // RBF constants in the packed exp2 form of kp_device.h (pack_rbf): exp(-(c - mu)^2 / (2 s^2)) = exp2(-(c sq - mu sq)^2);
      // wave-uniform, kept in scalar registers
      f32x2 sq2[kKC / 2], msq2[kKC / 2];
#pragma unroll
      for (int kp = 0; kp < kKC / 2; ++kp) {
        float sq[2], msq[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int k = 2 * kp + u;
          if (k < kK) {
            const float sg = prm[TklParams::sigma() + k];
            const float c2 = -1.4426950408889634f / (2.0f * sg * sg);
            sq[u] = sqrtf(-c2);
            msq[u] = prm[TklParams::mu() + k] * sq[u];
          } else {
            sq[u] = 0.0f;
            msq[u] = 1.0e3f;      // the dummy 12th kernel: exp2(-(0 c - 1e3)^2) = 0 for every c
          }
          sq[u] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sq[u])));
          msq[u] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, msq[u])));
        }
        sq2[kp] = f32x2{sq[0], sq[1]};
        msq2[kp] = f32x2{msq[0], msq[1]};
      }
      // item = (pair row j of the tile, query token i), token fastest: the two cosine rows of a pair are read in
      // runs of ql consecutive floats and the tile is written contiguously.  All loads of a batch are issued before
      // the first evaluation (one memory latency per batch; a 64-window tile at ql = 20 is one batch).
      const int items = nu * ql;
      for (int base = tz; base < items; base += kWThreads * kStageC) {
        if (!(ws == w0 && base < kWThreads)) issue_cos(ws, base);        // (the first batch is already in flight)
#pragma unroll
        for (int s = 0; s < kStageC; ++s) {
          const int idx = base + kWThreads * s;
          if (idx < items) {
            f32x2 o2[kKC / 2];
#pragma unroll
            for (int kp = 0; kp < kKC / 2; ++kp) o2[kp] = f32x2{0.0f, 0.0f};
            float cnt = 0.0f;
            if (cl[s] >= 0 && cinfo[cl[s]] >= 0) {
              // eleven kernels = five packed pairs + one alone (the dummy twelfth of the packed form was two v_exp per item
              // for an exact 0); the first position ASSIGNS its activations (0 + e = e exactly), the second adds
#pragma unroll
              for (int half = 0; half < 2; ++half) {
                const float c = half ? cb[s] : ca[s];
                const f32x2 cc = {c, c};
                f32x2 any2 = {0.0f, 0.0f};
#pragma unroll
                for (int kp = 0; kp < kK / 2; ++kp) {
                  const f32x2 sv = cc * sq2[kp] - msq2[kp];
                  const f32x2 av = -(sv * sv);
                  const f32x2 e = {__builtin_amdgcn_exp2f(av[0]), __builtin_amdgcn_exp2f(av[1])};
                  o2[kp] = half ? o2[kp] + e : e;
                  any2 += e;
                }
                const float sv = c * sq2[kK / 2][0] - msq2[kK / 2][0];
                const float e = __builtin_amdgcn_exp2f(-(sv * sv));
                o2[kK / 2][0] = half ? o2[kK / 2][0] + e : e;
                cnt += ((any2[0] + any2[1]) + e) != 0.0f ? 1.0f : 0.0f;  // (:210)
              }
            }
            f32x4* dst = (f32x4*)(tile + (size_t)idx * kKC);             // idx = j * ql + i: row j, token i
            dst[0] = f32x4{o2[0][0], o2[0][1], o2[1][0], o2[1][1]};
            dst[1] = f32x4{o2[2][0], o2[2][1], o2[3][0], o2[3][1]};
            dst[2] = f32x4{o2[4][0], o2[4][1], o2[5][0], cnt};
          }
        }
      }
    } else {
      const int row4 = rowf / 4;
      const int total4 = nu * row4;
      constexpr int kStage = 10;
      for (int base = tz; base < total4; base += kWThreads * kStage) {
        int pidx[kStage];
        int off[kStage];
    #pragma unroll
        for (int s = 0; s < kStage; ++s) {
          const int idx = base + kWThreads * s;
          pidx[s] = -1;
          off[s] = 0;
          if (idx < total4) {
            const int j = idx / row4, v = idx - j * row4;
            const int ug = ws + j;
            const int c = ug / kU, uu = ug - c * kU;
            off[s] = uu * srcf + t0 * kKC + v * 4;
            const int info = cinfo[c - c0];                                // c - c0 <= (19 + kWT + 14) / 20 < 8
            // rows of unwritten blocks are zeros
            if (info >= 0 && uu < 16 * (info & 3)) pidx[s] = info >> 2;
          }
        }
        f32x4 val[kStage];
    #pragma unroll
        for (int s = 0; s < kStage; ++s) {
          val[s] = f32x4{0, 0, 0, 0};
          if (pidx[s] >= 0) val[s] = *(const f32x4*)(ps + (int64_t)pidx[s] * kU * srcf + off[s]);
        }
    #pragma unroll
        for (int s = 0; s < kStage; ++s) {
          const int idx = base + kWThreads * s;
          if (idx < total4) *(f32x4*)(tile + (size_t)idx * 4) = val[s];
        }
      }
    }
    __syncthreads();

    // One item = kG adjacent windows of one query token: the windows share 15 - (kG - 1) of their pair rows, which are
    // summed once (kG = 4: 18 row reads for four windows instead of 60).  Only tokens below the effective query length
    // are evaluated (the others are multiplied by 0 in :248), and kG adapts to it so that one pass of the 512 threads
    // covers the tile with as little work per thread as possible: 16 ql items of four windows for long queries,
    // 32 ql of two / 64 ql of one when that still fits one pass (MSMARCO queries average ~6 tokens).
    // Only additions of non-negative terms: exact zeros stay exact, `lengths` stays an exact integer.
    if (ql * (wt / 1) <= kWThreads) window_items<SAT, 1, true>(tile, red, emb, q_mask, prm, sp, b, Q, ql, rowf, ws, W, tz, wt);
    else if (ql * (wt / 2) <= kWThreads) window_items<SAT, 2, true>(tile, red, emb, q_mask, prm, sp, b, Q, ql, rowf, ws, W, tz, wt);
    else if (ql * (wt / 4) <= kWThreads) window_items<SAT, 4, true>(tile, red, emb, q_mask, prm, sp, b, Q, ql, rowf, ws, W, tz, wt);
    else window_items<SAT, 4, false>(tile, red, emb, q_mask, prm, sp, b, Q, ql, rowf, ws, W, tz, wt);
    __syncthreads();
    if (tz < wt && ws + tz < W) {                                    // :249 sum over query tokens, :251 dense
      float s = 0.0f;
      const float* r = red + tz * (ql | 1);                              // odd stride: the 64 lanes hit distinct banks
#pragma unroll 4
      for (int i = 0; i < ql; ++i) s += r[i];
      __hip_atomic_store(win + (int64_t)b * W + ws + tz, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the next pass overwrites `tile` (last read before the barrier above) and, after its own barrier, `red`
  }
  arrive();
}

// Region top-k over the window scores of ONE document (:254-286) by the 256 threads of a workgroup: orig / work [Wp] in LDS
// hold the scores with 0 -> -9900 (:257) on entry.  Three arg-max rounds (ties -> lowest index, like torch.argmax), +-15
// suppression (:268-273), the peaks' +-1 / +-2 neighbours (:276-282), chunk_scoring dot (:286) -> *out.
// peaks (nullable): the three arg-max indices in round order = the reference's top_non_overlapping_idx row (:266-271, :290).
__device__ __forceinline__ void region_topk(float* orig, float* work, float* rv, int* ri, int Wp, const float* __restrict__ prm,
                                            float* __restrict__ out, int32_t* __restrict__ peaks, int tid) {
  const int lane = tid & 63, wv = tid >> 6;
  int top[3];
  for (int c = 0; c < 3; ++c) {                                        // :268-273
    float bv = -__builtin_huge_valf();
    int bi = 0x7fffffff;
    for (int w = tid; w < Wp; w += kRThreads) {
      const float v = work[w];
      if (v > bv) { bv = v; bi = w; }                                  // first maximal index within the thread
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }      // ties -> lowest index (torch.argmax)
    }
    if (lane == 0) { rv[wv] = bv; ri[wv] = bi; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kRThreads / 64; ++k) {
      const float ov = rv[k];
      const int oi = ri[k];
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    top[c] = bi;
    __syncthreads();                                                   // rv / ri are read; work may change
    for (int w = tid; w < Wp; w += kRThreads) {
      const int dlt = w > bi ? w - bi : bi - w;
      if (dlt < 15) work[w] = -10001.0f - (float)c;                    // |r - best| < 30/2
    }
    __syncthreads();
  }
  if (tid < 64) {
    // the 15 terms of :286 on 15 lanes (thread 0 walking them one after the other was a chain of 15 dependent loads of
    // the chunk_scoring weights at the end of every document's workgroup), summed by a fixed shuffle tree
    float term = 0.0f;
    if (lane < 15) {
      const int g = lane / 3, c = lane - 3 * g;
      const int off = g == 0 ? 0 : (g == 1 ? -1 : (g == 2 ? 1 : (g == 3 ? -2 : 2)));   // :276 peaks, -1, +1, -2, +2
      int idx = (c == 0 ? top[0] : (c == 1 ? top[1] : top[2])) + off;
      idx = idx < 0 ? 0 : (idx >= Wp ? Wp - 1 : idx);                  // :277-278
      float v = orig[idx];
      if (v <= -9900.0f) v = 0.0f;                                     // :282
      term = v * prm[TklParams::chunk_scoring() + lane];               // :286 (weight index g * 3 + c = lane)
    }
    const float s = wave_sum(term);
    if (lane == 0) *out = s;
    if (peaks && lane < 3) peaks[lane] = lane == 0 ? top[0] : (lane == 1 ? top[1] : top[2]);
  }
}

// One 256-thread workgroup per document: the default form (round 4 let the LAST window workgroup of a document do this itself,
// see tkl_window_kernel: MM_TKL_FOLD_REGIONS=1 for A/B runs).  (One wavefront per document spent 16 us on sixteen
// dependent 4-byte loads per lane; four wavefronts load the ~1,000 scores of a 2,048-token document in four rounds.)
__global__ void __launch_bounds__(kRThreads) tkl_region_kernel(const float* part, int n_planes, int64_t plane,
                                                         const int32_t* __restrict__ q_len, float* win,
                                                         const float* __restrict__ prm, float* __restrict__ out, int W,
                                                         const int32_t* __restrict__ ntile, int32_t* __restrict__ peaks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float rv[kRThreads / 64];
  __shared__ int ri[kRThreads / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Wp = W < 3 ? 3 : W;                                        // :254-255
  float* orig = (float*)smem;                                          // [Wp]
  float* work = orig + Wp;                                             // [Wp]
  // window score = sum over the token groups' planes in order (:249); planes whose tokens are all past the query's
  // effective length were never written and count as zeros, and so do the tiles past the document's last kept chunk
  int np = n_planes;
  if (q_len) {
    const int ql = q_len[b];
    np = ql <= 0 ? 0 : (ql + kTokGroup - 1) / kTokGroup;
    np = np < n_planes ? np : n_planes;
  }
  const int live_w = ntile[b] * kWT;
  // Four windows per thread at a time, every plane's value requested before the first store: `win` may be the plane buffer
  // itself (one token group), so with a store per window the compiler kept each window's loads behind the previous window's
  // store — four dependent memory round trips per document in a launch that is nothing but latency (9.8 us for 256 workgroups).
  constexpr int kIt = 4;
  for (int w0 = tid; w0 < Wp; w0 += kIt * kRThreads) {
    float s[kIt];
#pragma unroll
    for (int it = 0; it < kIt; ++it) s[it] = 0.0f;
    for (int g = 0; g < np; ++g) {
      float v[kIt];
#pragma unroll
      for (int it = 0; it < kIt; ++it) {
        const int w = w0 + it * kRThreads;
        v[it] = (w < W && w < live_w) ? part[g * plane + (int64_t)b * W + w] : 0.0f;
      }
#pragma unroll
      for (int it = 0; it < kIt; ++it) s[it] += v[it];                   // (planes in order, as before: 0 + v0 + v1 ...)
    }
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
      const int w = w0 + it * kRThreads;
      if (w < Wp) {
        float sv = s[it];
        if (w < W) win[(int64_t)b * W + w] = sv;
        if (sv == 0.0f) sv = -9900.0f;                                   // :257
        orig[w] = sv;
        work[w] = sv;
      }
    }
  }
  __syncthreads();
  region_topk(orig, work, rv, ri, Wp, prm, out + b, peaks ? peaks + 3 * (int64_t)b : nullptr, tid);
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// stage 1 -> stage 2 hand-off region: pair sums [P, 20, Q, 12] (A/B paths) or the slot-indexed cosine rows [B * C * 40, Q]
static size_t handoff_bytes(int64_t B, int64_t P, int C, int Q) {
  const size_t pairs = (size_t)P * kU * Q * kKC * 4, cosr = (size_t)B * C * 40 * Q * 4;
  return align256(pairs > cosr ? pairs : cosr);
}

}  // namespace mm

using namespace mm;

extern "C" size_t mm_tkl_workspace_bytes(int64_t B, int64_t P, int C, int Q, int K) {
  if (B <= 0 || P < 0 || C <= 0 || Q <= 0 || K != kK) return 0;
  const int W = ((C * 40 > 30 ? C * 40 : 30) - 30) / 2 + 1;
  return align256((size_t)B * C * 4) + handoff_bytes(B, P, C, Q) +
         packed_mask_bytes(MM_MASK_F32, P, 40) + align256((size_t)B * W * 4) + align256((size_t)B * Q * 4) +
         packed_mask_bytes(MM_MASK_F32, B, Q) +  // + the packed query mask (effective lengths)
         align256((size_t)((Q + kTokGroup - 1) / kTokGroup) * B * W * 4) +  // + the token groups' partial window scores
         align256((size_t)B * 4) +                                         // + live window tiles per document
         align256((size_t)B * 4);                                          // + arrival counters of the last-workgroup epilogue
}

extern "C" int mm_tkl_fwd(const void* q_ctx, const void* chunks, const float* chunk_mask, const int32_t* chunk_slot,
                          const float* q_mask, const float* params, float* win_scores, float* out, int64_t B,
                          int64_t P, int C, int Q, int E, int K, int saturation, void* workspace,
                          size_t workspace_bytes, void* stream_) {
  return mm_tkl_fwd_peaks(q_ctx, chunks, chunk_mask, chunk_slot, q_mask, params, win_scores, out, nullptr, B, P, C, Q, E, K,
                          saturation, workspace, workspace_bytes, stream_);
}

extern "C" int mm_tkl_fwd_peaks(const void* q_ctx, const void* chunks, const float* chunk_mask, const int32_t* chunk_slot,
                                const float* q_mask, const float* params, float* win_scores, float* out, int32_t* top_idx,
                                int64_t B, int64_t P, int C, int Q, int E, int K, int saturation, void* workspace,
                                size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!q_ctx || !q_mask || !params || !out) return set_error(MM_EINVAL, "tkl: null pointer");
  if (P > 0 && (!chunks || !chunk_mask || !chunk_slot)) return set_error(MM_EINVAL, "tkl: null chunk pointer");
  if (B <= 0 || P < 0 || C <= 0 || Q <= 0 || E <= 0) return set_error(MM_EINVAL, "tkl: bad shape");
  if (K != kK) return set_error(MM_EUNSUPPORTED, "tkl: K=%d kernels (only the reference's 11 are instantiated)", K);
  if (E % 4) return set_error(MM_EUNSUPPORTED, "tkl: E=%d rows are not 16-byte multiples", E);
  if (saturation != MM_TKL_SAT_EMBEDDING && saturation != MM_TKL_SAT_LOG)
    return set_error(MM_EUNSUPPORTED, "tkl: saturation %d (idf/linear read an undefined variable in the reference)", saturation);
  if (P > B * (int64_t)C) return set_error(MM_EINVAL, "tkl: more packed chunks than slots");
  const size_t need = mm_tkl_workspace_bytes(B, P, C, Q, K);
  if (!workspace || workspace_bytes < need) return set_error(MM_EWORKSPACE, "tkl: workspace needs %zu bytes", need);
  const int W = ((C * 40 > 30 ? C * 40 : 30) - 30) / 2 + 1;
  if ((size_t)(W < 3 ? 3 : W) * 8 > 64 * 1024) return set_error(MM_EUNSUPPORTED, "tkl: %d windows per document exceed the region kernel's LDS", W);

  char* ws = (char*)workspace;
  int32_t* slot2p = (int32_t*)ws;
  ws += align256((size_t)B * C * 4);
  float* ps = (float*)ws;
  const size_t ps_bytes = handoff_bytes(B, P, C, Q);
  ws += ps_bytes;
  size_t left = workspace_bytes - (size_t)(ws - (char*)workspace);
  float* win = win_scores;
  char* tail = (char*)workspace + align256((size_t)B * C * 4) + ps_bytes + packed_mask_bytes(MM_MASK_F32, P, 40);
  if (!win) win = (float*)tail;
  float* emb = (float*)(tail + align256((size_t)B * W * 4));
  // launch 1: slot map cleared, sat_emb_reduce1(q_ctx), effective query lengths (last real token + 1: query tokens
  // past them are masked in :248, so stage 1 does not write their pair rows and stage 2 does not evaluate them),
  // chunk masks (effective length + validity bits of the 40 centre tokens)
  PackedMask qmk, dm;
  float* planes = nullptr;                              // [n_planes][B][W] partial window scores of the token groups
  int32_t* ntile = nullptr;                             // [B] live window tiles per document (tkl_prep_kernel)
  int32_t* done_cnt = nullptr;                          // [B] window workgroups of the document that have published their scores
  const int n_planes = (Q + kTokGroup - 1) / kTokGroup;
  bool fold_regions = false;
  {
    if (P >= (1LL << 29)) return set_error(MM_EUNSUPPORTED, "tkl: too many packed chunks for one launch");
    const size_t need_dm = packed_mask_bytes(MM_MASK_F32, P, 40);
    if (left < need_dm) return set_error(MM_EWORKSPACE, "tkl: workspace too small for the chunk masks");
    int32_t* clen = (int32_t*)ws;
    uint32_t* cbits = (uint32_t*)(ws + (size_t)P * 4);
    char* qws = (char*)emb + align256((size_t)B * Q * 4);
    planes = (float*)(qws + packed_mask_bytes(MM_MASK_F32, B, Q));
    ntile = (int32_t*)((char*)planes + align256((size_t)n_planes * B * W * 4));
    done_cnt = (int32_t*)((char*)ntile + align256((size_t)B * 4));
    int32_t* qlen = (int32_t*)qws;
    uint32_t* qbits = (uint32_t*)(qws + (size_t)B * 4);
    const int n_fill = (int)((B * (int64_t)C + 255) / 256);
    const int n_emb = saturation == MM_TKL_SAT_EMBEDDING ? (int)((B * (int64_t)Q + 3) / 4) : 0;
    const int n_q = (int)((B + 3) / 4);
    const int n_chunk = (int)((P + 3) / 4);
    hipLaunchKernelGGL(tkl_prep_kernel, dim3((unsigned)(n_fill + n_emb + n_q + n_chunk)), dim3(256), 0, stream, slot2p,
                       B * (int64_t)C, n_fill, (const float*)q_ctx, params, emb, B * (int64_t)Q, E, n_emb, q_mask, B, Q, n_q,
                       qlen, qbits, chunk_mask, P, clen, cbits, chunk_slot, C, ntile, done_cnt);
    if (int e = check_launch("tkl_prep_kernel")) return e;
    qmk.len = qlen;
    qmk.bits = qbits;
    dm.len = clen;
    dm.bits = cbits;
  }
  // launch 2: stage 1 (scaled cosines — or, on the A/B paths, pair sums — of every kept chunk + its slot-map entry).
  // The cosine buffer [P * 40, Q] lives in the pair-sum region of the workspace (a sixth of its size).
  const bool use_cos = tkl_cos_supported(Q, E);
  if (P > 0) {
    if (int e = tkl_stage1_stream((const float*)q_ctx, (const float*)chunks, dm, qmk.len, chunk_slot, C,
                                  params + TklParams::mu(), params + TklParams::sigma(), ps, P, Q, E, slot2p,
                                  B * (int64_t)C, stream, use_cos ? ps : nullptr))
      return e;
  }
  {
    // LDS per workgroup: the whole 64-window tile when that leaves room for a second workgroup on the CU, otherwise
    // kWindowLds and the kernel makes passes of 32 / 16 windows for the documents whose queries need it
    const int qg = Q < kTokGroup ? Q : kTokGroup;                   // tokens per workgroup
    const size_t full = window_pass_bytes(kWT, qg);
    const size_t lds2 = full < (size_t)kWindowLds ? full : (size_t)kWindowLds;
    if (window_pass_bytes(8, qg) > lds2) return set_error(MM_EUNSUPPORTED, "tkl: Q=%d too large for the window kernel's LDS tile", Q);
    if (n_planes > 65535 || B > 65535) return set_error(MM_EUNSUPPORTED, "tkl: grid limits (B=%lld, Q=%d)", (long long)B, Q);
    const dim3 grid2((unsigned)((W + kWT - 1) / kWT), (unsigned)B, (unsigned)n_planes);
    float* wdst = n_planes > 1 ? planes : win;                         // one group: its plane IS the window-score output
    // The region top-k is its own launch (tkl_region_kernel) again since round 5.  Round 4 folded it into the last window
    // workgroup of each document (MM_TKL_FOLD_REGIONS=1 still does, when the tile's LDS holds the two score rows).  Measured on
    // one box, 256 documents, round-robin (profiles/r05_experiments/tkl_epilogue_orderings.txt):
    //     standalone launch                                   0.1388 / 0.1396 ms
    //     folded, round 4's relaxed counter + s_waitcnt       0.1413 / 0.1418     (correct on gfx950's sc1 path, not by the HIP model)
    //     folded, acq_rel arrival by thread 0 (the default of the folded form: correct by the model)   0.1927 / 0.1914
    //     folded, release / acquire fences in every thread    0.3367 / 0.3359
    // A kernel boundary is the cheapest agent-scope release there is: the buffer_wbl2 an in-launch release needs costs more per
    // workgroup (~1.7 us, MI355X_MICROARCH.md) than the 9.7 us launch it was meant to save, and even the unordered form was not
    // faster than the launch.
    fold_regions = env().tkl_fold_regions && (size_t)(W < 3 ? 3 : W) * 8 <= lds2;
    auto launch = [&](auto kern) {
      if (lds2 > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
      hipLaunchKernelGGL(kern, grid2, dim3(kWThreads), lds2, stream, (const float*)ps, (const int32_t*)slot2p, (const float*)emb, q_mask,
                         (const int32_t*)qmk.len, params, wdst, C, Q, W, (int)lds2, (const int32_t*)ntile, n_planes, win,
                         fold_regions ? out : (float*)nullptr, done_cnt, top_idx);
    };
    if (saturation == MM_TKL_SAT_EMBEDDING) {
      if (use_cos) launch(tkl_window_kernel<MM_TKL_SAT_EMBEDDING, true>);
      else launch(tkl_window_kernel<MM_TKL_SAT_EMBEDDING, false>);
    } else {
      if (use_cos) launch(tkl_window_kernel<MM_TKL_SAT_LOG, true>);
      else launch(tkl_window_kernel<MM_TKL_SAT_LOG, false>);
    }
    if (int e = check_launch("tkl_window_kernel")) return e;
  }
  if (fold_regions) return MM_OK;
  const int Wp = W < 3 ? 3 : W;
  hipLaunchKernelGGL(tkl_region_kernel, dim3((unsigned)B), dim3(kRThreads), (size_t)Wp * 8, stream,
                     (const float*)(n_planes > 1 ? planes : win), n_planes, (int64_t)B * W, (const int32_t*)qmk.len, win, params, out, W,
                     (const int32_t*)ntile, top_idx);
  return check_launch("tkl_region_kernel");
}
