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
constexpr int kWThreads = 512;


// slot -> (packed chunk index << 2) | number of 32-row blocks stage 1 writes for it (0..2); -1 = dropped
// chunk.  Stage 1 only writes the blocks below a chunk's effective length, so pair rows past them are
// taken as zeros by stage 2 instead of being zero-filled in HBM first (that memset of the whole pair
// buffer was 255 MB at B = 256 x 2048 tokens).
__global__ void __launch_bounds__(256) tkl_slot_map_kernel(const int32_t* __restrict__ chunk_slot,
                                                           const int32_t* __restrict__ chunk_len, int64_t P,
                                                           int64_t BC, int all_pairs, int32_t* __restrict__ slot2p) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p < P) {
    const int32_t s = chunk_slot[p];
    int len = chunk_len[p];
    len = len < 0 ? 0 : (len > 40 ? 40 : len);
    // the grouped stage-1 kernel writes every pair of a chunk; the per-chunk kernels only the blocks below its length
    if (s >= 0 && s < BC) slot2p[s] = (int32_t)((p << 2) | (all_pairs ? 2 : ((len + 31) >> 5)));
  }
}

// sat_emb_reduce1(q_ctx) (:224): one wavefront per (document, query token) -> emb[b][i]
__global__ void __launch_bounds__(256) tkl_emb_kernel(const float* __restrict__ q_ctx, const float* __restrict__ prm,
                                                      float* __restrict__ emb, int64_t BQ, int E) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= BQ) return;
  const float* qr = q_ctx + row * E;
  float s = 0.0f;
  for (int e = lane; e < E; e += 64) s += qr[e] * prm[TklParams::emb() + e];
  s = wave_sum(s);
  if (lane == 0) emb[row] = s;
}

// One workgroup = kWT consecutive windows of one document.
template <int SAT>
__global__ void __launch_bounds__(kWThreads) tkl_window_kernel(const float* __restrict__ ps, const int32_t* __restrict__ slot2p,
                                                         const float* __restrict__ emb_g,
                                                         const float* __restrict__ q_mask,
                                                         const int32_t* __restrict__ q_len,
                                                         const float* __restrict__ prm, float* __restrict__ win,
                                                         int C, int Q, int W) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.y;
  const int w0 = blockIdx.x * kWT;
  const int tid = threadIdx.x;
  const int nu = kWT + kWinPairs - 1;                 // pair rows needed by this tile
  const int rowf = Q * kKC;                           // floats per pair row
  float* tile = (float*)smem;                         // [nu][Q][12]
  float* emb = tile + (size_t)nu * rowf;              // [Q]   sat_emb_reduce1(q_ctx)  (:224)
  float* red = emb + ((Q + 3) & ~3);                  // [kWT][Q] per-(window, query token) dense-weighted value
  __shared__ int cinfo[8];                            // slot2p entries of the (<= 4) chunks this tile touches

  // the chunk lookups first, once per tile, so that the row loads below are independent of each other
  int ql = q_len ? q_len[b] : Q;                      // effective query length (rows of later tokens do not exist)
  ql = ql < 0 ? 0 : (ql > Q ? Q : ql);
  const int c0 = w0 / kU;
  if (tid < 8) {
    const int c = c0 + tid;
    cinfo[tid] = c < C ? slot2p[(int64_t)b * C + c] : -1;
  }
  __syncthreads();

  // ---- stage the pair-sum rows (zeros for dropped chunks) --------------------------------------
  // All row loads of a batch of kStage elements are issued before the first LDS store, so a thread pays
  // one memory latency per batch (a tile of Q = 20 is ONE batch); the kernel was 67 % s_waitcnt when every
  // element did its own dependent slot -> chunk -> row chain.
  const int row4 = rowf / 4;
  const int total4 = nu * row4;
  constexpr int kStage = 10;
  for (int base = tid; base < total4; base += kWThreads * kStage) {
    int pidx[kStage];
    int off[kStage];
#pragma unroll
    for (int s = 0; s < kStage; ++s) {
      const int idx = base + kWThreads * s;
      pidx[s] = -1;
      off[s] = 0;
      if (idx < total4) {
        const int j = idx / row4, v = idx - j * row4;
        const int ug = w0 + j;
        const int c = ug / kU, uu = ug - c * kU;
        off[s] = uu * rowf + v * 4;
        const int info = cinfo[c - c0];                                // c - c0 <= (19 + nu) / 20 < 8
        // rows of unwritten blocks and of query tokens past the effective length are zeros
        if (info >= 0 && uu < 16 * (info & 3) && v < 3 * ql) pidx[s] = info >> 2;
      }
    }
    f32x4 val[kStage];
#pragma unroll
    for (int s = 0; s < kStage; ++s) {
      val[s] = f32x4{0, 0, 0, 0};
      if (pidx[s] >= 0) val[s] = *(const f32x4*)(ps + (int64_t)pidx[s] * kU * rowf + off[s]);
    }
#pragma unroll
    for (int s = 0; s < kStage; ++s) {
      const int idx = base + kWThreads * s;
      if (idx < total4) *(f32x4*)(tile + (size_t)idx * 4) = val[s];
    }
  }
  if (SAT == MM_TKL_SAT_EMBEDDING && tid < Q) emb[tid] = emb_g[(int64_t)b * Q + tid];
  __syncthreads();

  const float* sp = prm + TklParams::sat();
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  // One item = FOUR adjacent windows of one query token: windows wl .. wl + 3 share 12 of their 15 pair rows, so
  // the shared rows are summed once (18 row reads for four windows instead of 60).  Only additions of
  // non-negative terms: exact zeros stay exact, `lengths` stays an exact integer.
  constexpr int kG = 4;
  for (int item = tid; item < (kWT / kG) * Q; item += kWThreads) {
    const int wp = item / Q, i = item - wp * Q;
    const int wl = kG * wp;
    if (w0 + wl >= W) {
#pragma unroll
      for (int which = 0; which < kG; ++which) red[(wl + which) * Q + i] = 0.0f;
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
    // rows 0..2 and 15..17 are the edges, rows 3..14 the core every window of the item contains
    f32x2 core[kKC / 2], edge[6][kKC / 2], tmp[kKC / 2];
    row(0, edge[0]);
    row(1, edge[1]);
    row(2, edge[2]);
    row(3, core);
#pragma unroll
    for (int j = 4; j < kWinPairs; ++j) {
      row(j, tmp);
#pragma unroll
      for (int k = 0; k < kKC / 2; ++k) core[k] += tmp[k];
    }
    row(kWinPairs, edge[3]);
    row(kWinPairs + 1, edge[4]);
    row(kWinPairs + 2, edge[5]);
    const float qmv = q_mask[(int64_t)b * Q + i];
#pragma unroll
    for (int which = 0; which < kG; ++which) {
      // window wl + which = edges which..2 (head) + core + edges 3..2+which (tail)
      f32x2 acc2[kKC / 2];
#pragma unroll
      for (int k = 0; k < kKC / 2; ++k) {
        f32x2 v = core[k];
#pragma unroll
        for (int e = 0; e < 6; ++e)
          if ((e < 3 && e >= which) || (e >= 3 && e < 3 + which)) v += edge[e][k];
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
        const float rstd = 1.0f / sqrtf((d0 * d0 + d1 * d1) * 0.5f + 1e-5f);
        const float n0 = d0 * rstd * sp[9] + sp[11], n1 = d1 * rstd * sp[10] + sp[12];
        const float s1 = n0 * sp[0] + n1 * sp[1] + sp[2];              // :230
        const float s2 = 1.0f / (n0 * sp[3] + n1 * sp[4] + sp[5]);      // :231
        const float s3 = n0 * sp[6] + n1 * sp[7] + sp[8];              // :232
#pragma unroll
        for (int k = 0; k < kK; ++k) {
          // x^s2 = exp2(s2 * log2 x) on the hardware transcendentals (x >= 1e-10 > 0): 3 instructions
          // instead of ~80 for powf; |s2 * log2 x| <= ~33 |s2| keeps the error ~1e-6 relative
          const float xp = __builtin_amdgcn_exp2f(s2 * __builtin_amdgcn_logf(fmaxf(pk[k], 1e-10f)));
          const float sat = s1 * xp - s3;  // :234
          val += prm[TklParams::dense() + k] * (sat * factor);
        }
      } else {
#pragma unroll
        for (int k = 0; k < kK; ++k) {
          const float sat = logf(fmaxf(pk[k] * prm[TklParams::kmult() + k], 1e-10f));   // :246
          val += prm[TklParams::dense() + k] * (sat * factor);
        }
      }
      red[(wl + which) * Q + i] = (w0 + wl + which < W) ? val : 0.0f;
    }
  }
  __syncthreads();
  if (tid < kWT && w0 + tid < W) {                                     // :249 sum over query tokens, :251 dense
    float s = 0.0f;
    for (int i = 0; i < ql; ++i) s += red[tid * Q + i];
    win[(int64_t)b * W + w0 + tid] = s;
  }
}

// One 256-thread workgroup per document: region top-k over the window scores (:254-286).  (One wavefront per
// document spent 16 us on sixteen dependent 4-byte loads per lane; four wavefronts load the ~1,000 scores of a
// 2,048-token document in four rounds and share the arg-max.)
__global__ void __launch_bounds__(256) tkl_region_kernel(const float* __restrict__ win, const float* __restrict__ prm,
                                                         float* __restrict__ out, int W) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float rv[4];
  __shared__ int ri[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int Wp = W < 3 ? 3 : W;                                        // :254-255
  float* orig = (float*)smem;                                          // [Wp]
  float* work = orig + Wp;                                             // [Wp]
  for (int w = tid; w < Wp; w += 256) {
    float s = w < W ? win[(int64_t)b * W + w] : 0.0f;
    if (s == 0.0f) s = -9900.0f;                                       // :257
    orig[w] = s;
    work[w] = s;
  }
  __syncthreads();
  int top[3];
  for (int c = 0; c < 3; ++c) {                                        // :268-273
    float bv = -__builtin_huge_valf();
    int bi = 0x7fffffff;
    for (int w = tid; w < Wp; w += 256) {
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
    for (int k = 0; k < 4; ++k) {
      const float ov = rv[k];
      const int oi = ri[k];
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    top[c] = bi;
    __syncthreads();                                                   // rv / ri are read; work may change
    for (int w = tid; w < Wp; w += 256) {
      const int dlt = w > bi ? w - bi : bi - w;
      if (dlt < 15) work[w] = -10001.0f - (float)c;                    // |r - best| < 30/2
    }
    __syncthreads();
  }
  if (tid == 0) {
    const int offs[5] = {0, -1, 1, -2, 2};                             // :276 peaks, -1, +1, -2, +2
    float s = 0.0f;
    for (int g = 0; g < 5; ++g)
      for (int c = 0; c < 3; ++c) {
        int idx = top[c] + offs[g];
        idx = idx < 0 ? 0 : (idx >= Wp ? Wp - 1 : idx);                // :277-278
        float v = orig[idx];
        if (v <= -9900.0f) v = 0.0f;                                   // :282
        s += v * prm[TklParams::chunk_scoring() + g * 3 + c];          // :286
      }
    out[b] = s;
  }
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace mm

using namespace mm;

extern "C" size_t mm_tkl_workspace_bytes(int64_t B, int64_t P, int C, int Q, int K) {
  if (B <= 0 || P < 0 || C <= 0 || Q <= 0 || K != kK) return 0;
  const int W = ((C * 40 > 30 ? C * 40 : 30) - 30) / 2 + 1;
  return align256((size_t)B * C * 4) + align256((size_t)P * kU * Q * kKC * 4) +
         packed_mask_bytes(MM_MASK_F32, P, 40) + align256((size_t)B * W * 4) + align256((size_t)B * Q * 4) +
         packed_mask_bytes(MM_MASK_F32, B, Q);  // + the packed query mask (effective lengths)
}

extern "C" int mm_tkl_fwd(const void* q_ctx, const void* chunks, const float* chunk_mask, const int32_t* chunk_slot,
                          const float* q_mask, const float* params, float* win_scores, float* out, int64_t B,
                          int64_t P, int C, int Q, int E, int K, int saturation, void* workspace,
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
  const size_t ps_bytes = align256((size_t)P * kU * Q * kKC * 4);
  ws += ps_bytes;
  size_t left = workspace_bytes - (size_t)(ws - (char*)workspace);
  float* win = win_scores;
  if (hipMemsetAsync(slot2p, 0xFF, (size_t)B * C * 4, stream) != hipSuccess) return set_error(MM_ELAUNCH, "tkl: memset failed");
  char* tail = (char*)workspace + align256((size_t)B * C * 4) + ps_bytes + packed_mask_bytes(MM_MASK_F32, P, 40);
  if (!win) win = (float*)tail;
  float* emb = (float*)(tail + align256((size_t)B * W * 4));
  if (saturation == MM_TKL_SAT_EMBEDDING) {
    hipLaunchKernelGGL(tkl_emb_kernel, dim3((unsigned)((B * Q + 3) / 4)), dim3(256), 0, stream, (const float*)q_ctx, params,
                       emb, B * (int64_t)Q, E);
    if (int e = check_launch("tkl_emb_kernel")) return e;
  }
  // effective query lengths (last real token + 1): query tokens past them are masked in :248, so stage 1 does not
  // write their pair rows and stage 2 does not read or evaluate them
  PackedMask qmk;
  {
    char* qws = (char*)emb + align256((size_t)B * Q * 4);
    size_t qleft = packed_mask_bytes(MM_MASK_F32, B, Q);
    if (int e = resolve_mask(q_mask, MM_MASK_F32, B, Q, &qws, &qleft, stream, &qmk)) return e;
  }
  // Stages 1 + 2 fused per document (pair sums stay in LDS) when the shape fits; otherwise stage 1 writes the
  // pair sums to the workspace and the window kernel reads them back.
  const bool fused = tkl_fused_supported(C, Q, E);
  PackedMask dm;
  if (P > 0) {
    if (P >= (1LL << 29)) return set_error(MM_EUNSUPPORTED, "tkl: too many packed chunks for one launch");
    if (int e = resolve_mask(chunk_mask, MM_MASK_F32, P, 40, &ws, &left, stream, &dm, 50, 5)) return e;
    hipLaunchKernelGGL(tkl_slot_map_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, chunk_slot, dm.len, P,
                       B * (int64_t)C, tkl_stage1_writes_all_pairs(Q, E) ? 1 : 0, slot2p);
    if (int e = check_launch("tkl_slot_map_kernel")) return e;
    if (!fused) {
      if (int e = tkl_stage1_stream((const float*)q_ctx, (const float*)chunks, dm, qmk.len, chunk_slot, C,
                                    params + TklParams::mu(), params + TklParams::sigma(), ps, P, Q, E, stream))
        return e;
    }
  }
  if (fused) {
    if (int e = tkl_fused((const float*)q_ctx, (const float*)chunks, dm, slot2p, q_mask, params,
                          saturation == MM_TKL_SAT_EMBEDDING ? emb : nullptr, win, B, C, Q, E, W, saturation, stream))
      return e;
  } else {
    const int nu = kWT + kWinPairs - 1;
    const size_t lds2 = ((size_t)nu * Q * kKC + ((Q + 3) & ~3) + (size_t)kWT * Q) * 4;
    if (lds2 > 160 * 1024) return set_error(MM_EUNSUPPORTED, "tkl: Q=%d too large for the window kernel's LDS tile", Q);
    const dim3 grid2((unsigned)((W + kWT - 1) / kWT), (unsigned)B);
    if (saturation == MM_TKL_SAT_EMBEDDING) {
      if (lds2 > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)tkl_window_kernel<MM_TKL_SAT_EMBEDDING>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
      hipLaunchKernelGGL(tkl_window_kernel<MM_TKL_SAT_EMBEDDING>, grid2, dim3(kWThreads), lds2, stream, ps, slot2p,
                         emb, q_mask, qmk.len, params, win, C, Q, W);
    } else {
      if (lds2 > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)tkl_window_kernel<MM_TKL_SAT_LOG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
      hipLaunchKernelGGL(tkl_window_kernel<MM_TKL_SAT_LOG>, grid2, dim3(kWThreads), lds2, stream, ps, slot2p,
                         emb, q_mask, qmk.len, params, win, C, Q, W);
    }
    if (int e = check_launch("tkl_window_kernel")) return e;
  }
  const int Wp = W < 3 ? 3 : W;
  hipLaunchKernelGGL(tkl_region_kernel, dim3((unsigned)B), dim3(256), (size_t)Wp * 8, stream, win, params, out, W);
  return check_launch("tkl_region_kernel");
}
