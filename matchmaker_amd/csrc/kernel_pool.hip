// TK kernel pooling for MI355X (gfx950): cosine match matrix + K RBF kernels + log-sum pooling.
//
//   cos[i,j] = <q_i, d_j> / ((|q_i| + 1e-13)(|d_j| + 1e-13))          (allennlp CosineMatrixAttention,
//                                                                       call site ecai20_tk.py:105)
//   pkq[i,k] = sum_j dmask[j] * exp(-(cos[i,j] - mu_k)^2 / (2 sigma_k^2))     ecai20_tk.py:112-120
//   out      = sum_k w_k * sum_i qmask[i] * log(max(pkq[i,k] * alpha_k, 1e-10))   ecai20_tk.py:121-124
//
// fp32 throughout (tk.yaml: use_fp16 False; the RBF with sigma = 0.1 amplifies cosine error ~6x,
// so the dot products use the exact-f32 MFMA v_mfma_f32_32x32x2_f32).
//
// kernel_pool_stream_kernel<NS> (E == 100*NS, Q <= 32): one wavefront per workgroup.  The D stream
// moves HBM -> LDS by LDS-DMA in slices of 32 document tokens x 25 16-B chunks (12.8 KB; row stride
// 400 B = 100 dwords keeps every ds_read_b128 service group on 16 distinct bank slots, no swizzle
// needed); the query tile lives in VGPRs as MFMA B operands; document tokens sit on the MFMA M
// axis, so each lane owns one query token (column) x 16 document rows and the K RBF sums are
// lane-local accumulators.  Row norms of the document tokens are accumulated by the lanes from the
// very A operands they feed to the MFMA (free VALU work in the MFMA shadow).
//
// kernel_pool_generic_kernel: any E (multiple of 4), Q, D; direct fragment loads.
#include "mm_internal.h"
#include "kp_device.h"

// VALU instructions scheduled behind each MFMA of the split kernels' K steps (sched_group_barrier).  Measured on one box,
// round-robin: 3: 2.82-2.84 ms | 4: 2.79-2.83 | 5: 2.78-2.79 | 6: 2.79 | 7: 2.81-2.83 | 8: 2.83-2.87 | 10: 2.83-2.85 for TK
// (profiles/r03_experiments/tk_schedule_ab.txt); TKL's stage 1 does not care (0.135-0.138 ms for 3 .. 12).
constexpr int kShadowValu = 5;

// The split-bf16 dot product of the TK pooling kernel is  hi.hi + lo.hi + hi.lo  (three MFMAs per K step): the fourth
// product lo.lo is below 2^-16 of the result and was kept through round 3 on a 1e-6-scale criterion on exact duplicates;
// against the contract (scores within 1e-3, decided ranks) it changes nothing that can be measured — max |score - fp64|
// 5.2745e-6 without it, 5.2733e-6 with it on the 16 x 1000 rank lists, 15,984 vs 15,986 of 16,000 positions equal to the
// stable sort of the fp32 reference (tests/test_zz_rank_order_gpu.py, profiles/r04_rank_parity/) — and it costs 4.5 % of the
// call on a kernel that runs into the board's power limit.  -DMM_KP_LOLO=1 (tools/build_variant.sh) builds
// the four-product kernel for A/B runs.
#ifndef MM_KP_LOLO
#define MM_KP_LOLO 0
#endif
#ifndef MM_TKL_LOLO      // the same switch for TKL's stage 1 (tkl_stage1_run_kernel): parity-neutral there too (window error 2.5e-5 either
                         // way, 256 / 256 rank positions), -1 us of 138 — the kernel is not power-limited, so it buys little
#define MM_TKL_LOLO 0
#endif

namespace mm {


// TKL epilogue of one block of a chunk's centre tokens (block t = rows 32t..32t+31 of the 40):
// per position pair u the K summed activations (ecai-style RBF, masked: sigir20_tkl.py:192-194) and
// the count of positions with a non-zero activation (feeds `lengths`, :210).
template <int K>
__device__ __forceinline__ void tkl_block(float* ps_chunk, int Q, int t, int r, int h, const f32x16& acc,
                                          const float (&rdr)[16], float rq, uint32_t vbits, const Rbf& rbf) {
  constexpr int KC = K + 1;
  constexpr int KP = (K + 1) / 2;
  static_assert(KC % 4 == 0 && K % 2 == 1, "K kernels + the count channel must fill whole float4s");
  const int npairs = t == 0 ? 8 : 2;  // block 1 only holds positions 32..39 (rows 0..3 + 4h)
#pragma unroll
  for (int ip = 0; ip < 8; ++ip) {
    if (ip < npairs) {  // wave-uniform guard; keeps every register index static
      f32x2 o2[KP];
#pragma unroll
      for (int k = 0; k < KP; ++k) o2[k] = f32x2{0.0f, 0.0f};
      float cnt = 0.0f;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int i = 2 * ip + half;
        float c = (acc[i] * rq) * rdr[i];
        // masked position: cosine 1e5 underflows every kernel to exactly 0 (= the reference's
        // multiply by the 0 mask, :194); the dummy 12th kernel is built to be 0 everywhere
        c = ((vbits >> rowof(i)) & 1u) ? c : 1.0e5f;
        const f32x2 cc = {c, c};
        f32x2 any2 = {0.0f, 0.0f};
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
          const f32x2 sv = cc * rbf.sq2[kp] - rbf.msq2[kp];
          const f32x2 av = -(sv * sv);
          const f32x2 e = {__builtin_amdgcn_exp2f(av[0]), __builtin_amdgcn_exp2f(av[1])};
          o2[kp] += e;
          any2 += e;
        }
        cnt += (any2[0] + any2[1]) != 0.0f ? 1.0f : 0.0f;  // activations are >= 0: sum != 0 <=> any != 0 (:210)
      }
      const int row0 = rowof(2 * ip) + 4 * h;
      const int u = (32 * t + row0) >> 1;
      if (r < Q && u < 20) {
        f32x4* dst = (f32x4*)(ps_chunk + ((int64_t)u * Q + r) * KC);
#pragma unroll
        for (int v = 0; v < KC / 4 - 1; ++v) dst[v] = f32x4{o2[2 * v][0], o2[2 * v][1], o2[2 * v + 1][0], o2[2 * v + 1][1]};
        dst[KC / 4 - 1] = f32x4{o2[KP - 2][0], o2[KP - 2][1], o2[KP - 1][0], cnt};
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// streaming kernel
// ---------------------------------------------------------------------------------------------
constexpr int kSC = 25;                       // 16-B chunks per row per slice (odd: conflict-free)
constexpr int kSliceInstr = 13;               // ceil(32*25 / 64) LDS-DMA instructions per slice
constexpr int kSliceBytes = kSliceInstr * 1024;
constexpr int kPairSteps = 13;                // chunk pairs per slice (last one half empty)

template <bool NT>
__device__ __forceinline__ void issue_slice(const char* gbase, const uint32_t (&voff)[kSliceInstr], uint32_t vmax,
                                            bool clamp, uint32_t lds_dst) {
  uint32_t v[kSliceInstr];
#pragma unroll
  for (int n = 0; n < kSliceInstr; ++n) v[n] = clamp ? (voff[n] < vmax ? voff[n] : vmax) : voff[n];
  uint32_t keep;
#define MM_GLDS(N) "s_nop 0\n\tglobal_load_lds_dwordx4 %" #N ", %14" NTS "\n\ts_add_u32 m0, m0, 0x400\n\t"
#define NTS " nt"
  if (NT) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %15\n\t" MM_GLDS(1) MM_GLDS(2) MM_GLDS(3)
                     MM_GLDS(4) MM_GLDS(5) MM_GLDS(6) MM_GLDS(7) MM_GLDS(8) MM_GLDS(9) MM_GLDS(10) MM_GLDS(11)
                         MM_GLDS(12) MM_GLDS(13) "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]),
                   "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "s"(gbase), "s"(lds_dst)
                 : "memory", "scc");
  } else {
#undef NTS
#define NTS ""
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %15\n\t" MM_GLDS(1) MM_GLDS(2) MM_GLDS(3)
                     MM_GLDS(4) MM_GLDS(5) MM_GLDS(6) MM_GLDS(7) MM_GLDS(8) MM_GLDS(9) MM_GLDS(10) MM_GLDS(11)
                         MM_GLDS(12) MM_GLDS(13) "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]),
                   "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "s"(gbase), "s"(lds_dst)
                 : "memory", "scc");
  }
#undef NTS
#undef MM_GLDS
}

__device__ __forceinline__ void wait_slices(int younger) {
  switch (younger) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(39)" ::: "memory"); break;
  }
}

// 13 x 16 B of this lane's query row chunk pairs of one slice (+ vmcnt(0): runs once per query)
__device__ __forceinline__ void load_q_slice(const char* base, f32x4 (&qf)[kPairSteps]) {
  asm volatile(
      "global_load_dwordx4 %0, %13, off\n\t"
      "global_load_dwordx4 %1, %13, off offset:32\n\t"
      "global_load_dwordx4 %2, %13, off offset:64\n\t"
      "global_load_dwordx4 %3, %13, off offset:96\n\t"
      "global_load_dwordx4 %4, %13, off offset:128\n\t"
      "global_load_dwordx4 %5, %13, off offset:160\n\t"
      "global_load_dwordx4 %6, %13, off offset:192\n\t"
      "global_load_dwordx4 %7, %13, off offset:224\n\t"
      "global_load_dwordx4 %8, %13, off offset:256\n\t"
      "global_load_dwordx4 %9, %13, off offset:288\n\t"
      "global_load_dwordx4 %10, %13, off offset:320\n\t"
      "global_load_dwordx4 %11, %13, off offset:352\n\t"
      "global_load_dwordx4 %12, %14, off\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(qf[0]), "=&v"(qf[1]), "=&v"(qf[2]), "=&v"(qf[3]), "=&v"(qf[4]), "=&v"(qf[5]), "=&v"(qf[6]),
        "=&v"(qf[7]), "=&v"(qf[8]), "=&v"(qf[9]), "=&v"(qf[10]), "=&v"(qf[11]), "=&v"(qf[12])
      : "v"(base), "v"(base + 384 - (threadIdx.x >> 5) * 16)  // last pair: chunk 24 only (h=1 lanes get a dummy)
      : "memory");
}

template <int NS, int K, int NBUF, bool NT, bool TKL>
__global__ void __launch_bounds__(64) kernel_pool_stream_kernel(const KpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int r = lane & 31, h = lane >> 5;
  const int64_t p0 = (int64_t)blockIdx.x * a.pairs_per_wave;
  const int64_t p1 = (p0 + a.pairs_per_wave < a.n_pairs) ? p0 + a.pairs_per_wave : a.n_pairs;
  if (p0 >= p1) return;
  constexpr int E = 100 * NS;
  constexpr int RB = E * 4;  // row bytes
  const int D = a.D, Q = a.Q;
  const int nblk_tot = (D + 31) >> 5;
  const int rows_last = D - 32 * (nblk_tot - 1);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  float* rdbuf = (float*)(smem + NBUF * kSliceBytes);  // 32 floats: 1/(|d|+tiny) of the block's rows

  // LDS-DMA source offsets: slot s = 64n + lane of the slice image [32 rows][25 chunks]
  uint32_t voff[kSliceInstr];
#pragma unroll
  for (int n = 0; n < kSliceInstr; ++n) {
    int s = 64 * n + lane;
    if (s > 32 * kSC - 1) s = 32 * kSC - 1;  // the last half instruction re-reads the final chunk
    const int row = s / kSC, c = s - row * kSC;
    voff[n] = (uint32_t)(row * RB + c * 16);
  }
  const uint32_t vmax_tail = (uint32_t)((rows_last - 1) * RB + (kSC - 1) * 16);
  const uint32_t a_off = (uint32_t)(r * (kSC * 16) + h * 16);  // this lane's A-fragment base inside a slice

  Rbf rbf;
  load_rbf<K, false>(a.mu, a.sigma, a.alpha, a.w, rbf);   // (the exact-fp32 twin keeps the direct RBF form)

  const char* dbase = (const char*)a.d;
  auto doc_len = [&](int64_t p) -> int {
    int len = a.dm.len ? (int)sload_u32(a.dm.len, p) : D;
    return len < 0 ? 0 : (len > D ? D : len);
  };

  // producer cursor over (pair, block, slice)
  int64_t pp = p0;
  int pt = 0, ps = 0, pn = 0;
  while (pp < p1 && (pn = (doc_len(pp) + 31) >> 5) == 0) ++pp;
  int pbuf = 0, cbuf = 0, inflight = 0;
  auto top_up = [&]() {
    while (pp < p1 && inflight < NBUF) {
      const char* g = dbase + (pp * a.d_doc_rows + a.d_row0 + (int64_t)pt * 32) * RB + ps * (kSC * 16);
      issue_slice<NT>(g, voff, vmax_tail, pt == nblk_tot - 1 && rows_last != 32, lds0 + (uint32_t)pbuf * kSliceBytes);
      pbuf = (pbuf + 1 == NBUF) ? 0 : pbuf + 1;
      ++inflight;
      if (++ps == NS) {
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

  f32x4 qf[NS][kPairSteps];
  float rq = 0.0f;
  bool qvalid = false;
  int64_t cur_q = -1;
  int64_t qi = TKL ? 0 : p0 / a.ppq;
  int64_t q_left = TKL ? 0 : a.ppq - (p0 - qi * a.ppq);

  for (int64_t pair = p0; pair < p1; ++pair) {
    if (TKL) {
      qi = (int64_t)((int)sload_u32(a.chunk_slot, pair) / a.C);
      tkl_publish_slot(a, pair, (doc_len(pair) + 31) >> 5, lane);
    } else if (a.pair_q) {
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
      const char* qrow = (const char*)a.q + (qi * Q + qr) * RB + h * 16;
      float ss = 0.0f;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        load_q_slice(qrow + s * (kSC * 16), qf[s]);
        if (h) qf[s][kPairSteps - 1] = f32x4{0, 0, 0, 0};  // chunk 25 of the slice does not exist
#pragma unroll
        for (int p = 0; p < kPairSteps; ++p)
#pragma unroll
          for (int j = 0; j < 4; ++j) ss += qf[s][p][j] * qf[s][p][j];
      }
      ss += __shfl_xor(ss, 32, 64);
      rq = 1.0f / (sqrtf(ss) + 1e-13f);
      const int qlen = a.qm.len ? (int)sload_u32(a.qm.len, qi) : Q;
      qvalid = r < Q && r < qlen;
      if (a.qm.bits) qvalid = qvalid && ((sload_u32(a.qm.bits, qi) >> r) & 1u);
    }
    const int len = doc_len(pair);
    const int nb = (len + 31) >> 5;
    f32x2 pk2[kMaxK / 2];
#pragma unroll
    for (int k = 0; k < kMaxK / 2; ++k) pk2[k] = f32x2{0.0f, 0.0f};

    for (int t = 0; t < nb; ++t) {
      f32x16 acc = {0};
      f32x2 ss2 = {0.0f, 0.0f};
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        top_up();
        wait_slices(inflight - 1);
        const char* buf = smem + cbuf * kSliceBytes + a_off;
#pragma unroll
        for (int p = 0; p < kPairSteps; ++p) {
          f32x4 av = *(const f32x4*)(buf + p * 32);
          if (p == kPairSteps - 1 && h) av = f32x4{0, 0, 0, 0};
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], qf[s][p][j], acc, 0, 0, 0);
          const f32x2 lo = {av[0], av[1]}, hi = {av[2], av[3]};
          ss2 += lo * lo;
          ss2 += hi * hi;
        }
        cbuf = (cbuf + 1 == NBUF) ? 0 : cbuf + 1;
        --inflight;
      }
      // document-token norms: lane (r,h) summed the even/odd chunks of row r
      float ss = ss2[0] + ss2[1];
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
      if (TKL)
        tkl_block<K>(a.ps_out + pair * (20 * (int64_t)Q * (K + 1)), Q, t, r, h, acc, rdr, rq, va >> (4 * h), rbf);
      else
        rbf_block<K>(pk2, acc, rdr, rq, va, h, rbf);
    }
    if (!TKL) {
      float pk[kMaxK];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        pk[k] = pk2[k >> 1][k & 1];
        pk[k] += __shfl_xor(pk[k], 32, 64);
      }
      finish_pool<K>(a, pair, pk, qvalid && lane < 32, lane, rbf, lane < 32 ? lane : -1);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// split-bf16 streaming kernel (the default for E == 100*NS, Q <= 32)
//
// The exact-f32 MFMA runs at the fp32 VECTOR rate (64 cycles per 32x32x2 on a SIMD) and shares the
// SIMD's FMA lanes with the VALU epilogue: at E = 300 it alone is 9,600 cycles per 32-token block
// (profiles/r01_tk_pmc.json: MFMA busy 51 %, no overlap with the RBF math).  This kernel feeds the
// 16x faster bf16 matrix pipe instead: every fp32 operand x is split as x = hi + lo + eps with
// hi = bf16_rne(x), lo = bf16_rne(x - hi) (|eps| <= 2^-18 |x|), and the dot product is the sum of the
// FOUR bf16 MFMAs hi.hi + lo.hi + hi.lo + lo.lo accumulated in fp32.  Keeping the lo.lo term
// matters: for near-identical vectors (the cosine ~ 1 matches that feed the mu = 1.0 / 0.9
// kernels) the lo_i^2 terms are all positive and would bias the cosine by ~3e-6.  With it the
// cosine error is ~1e-7 on ordinary inputs (= an fp32 accumulation in a different order) and at most
// ~1.5e-6 on planted near-duplicates (bounded against the fp64 oracle on the CPU by
// tests/test_host_cpu.py::test_split_bf16_numerics, and on the GPU by the parity tests).
// Cost per 16 K-values: 4 x 32 MFMA cycles (was 8 x 64) + 20 VALU ops for the split of the 8 values
// a lane feeds.  The bf16 pipe is independent of the VALU, so the split and the RBF epilogue can
// overlap it.
//
// K order inside a slice of 25 chunks: step p (0..5) takes chunks 4p..4p+3 (lane half h: chunks
// 4p+2h, 4p+2h+1 = 8 consecutive floats); chunk 24 of each slice is parked and the NS parked chunks
// form one extra step (slice s -> lane half s>>1, position s&1) = 6*NS + 1 steps per block.
// ---------------------------------------------------------------------------------------------
constexpr int kSplitSteps = 6;  // full 16-wide K steps per 100-float slice

// this lane's 12 chunks of one query-row slice (base = row + slice*400 + h*32) + the parked chunk 24
// (left = row + slice*400 + 384); own vmcnt(0): runs once per query
template <bool WAIT = true>
__device__ __forceinline__ void load_q_slice_split(const char* base, const char* left, f32x4 (&qr)[13]) {
  if constexpr (!WAIT) {
    asm volatile(
        "global_load_dwordx4 %0, %13, off\n\t"
        "global_load_dwordx4 %1, %13, off offset:16\n\t"
        "global_load_dwordx4 %2, %13, off offset:64\n\t"
        "global_load_dwordx4 %3, %13, off offset:80\n\t"
        "global_load_dwordx4 %4, %13, off offset:128\n\t"
        "global_load_dwordx4 %5, %13, off offset:144\n\t"
        "global_load_dwordx4 %6, %13, off offset:192\n\t"
        "global_load_dwordx4 %7, %13, off offset:208\n\t"
        "global_load_dwordx4 %8, %13, off offset:256\n\t"
        "global_load_dwordx4 %9, %13, off offset:272\n\t"
        "global_load_dwordx4 %10, %13, off offset:320\n\t"
        "global_load_dwordx4 %11, %13, off offset:336\n\t"
        "global_load_dwordx4 %12, %14, off"
        : "=&v"(qr[0]), "=&v"(qr[1]), "=&v"(qr[2]), "=&v"(qr[3]), "=&v"(qr[4]), "=&v"(qr[5]), "=&v"(qr[6]),
          "=&v"(qr[7]), "=&v"(qr[8]), "=&v"(qr[9]), "=&v"(qr[10]), "=&v"(qr[11]), "=&v"(qr[12])
        : "v"(base), "v"(left)
        : "memory");
    return;
  }
  asm volatile(
      "global_load_dwordx4 %0, %13, off\n\t"
      "global_load_dwordx4 %1, %13, off offset:16\n\t"
      "global_load_dwordx4 %2, %13, off offset:64\n\t"
      "global_load_dwordx4 %3, %13, off offset:80\n\t"
      "global_load_dwordx4 %4, %13, off offset:128\n\t"
      "global_load_dwordx4 %5, %13, off offset:144\n\t"
      "global_load_dwordx4 %6, %13, off offset:192\n\t"
      "global_load_dwordx4 %7, %13, off offset:208\n\t"
      "global_load_dwordx4 %8, %13, off offset:256\n\t"
      "global_load_dwordx4 %9, %13, off offset:272\n\t"
      "global_load_dwordx4 %10, %13, off offset:320\n\t"
      "global_load_dwordx4 %11, %13, off offset:336\n\t"
      "global_load_dwordx4 %12, %14, off\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(qr[0]), "=&v"(qr[1]), "=&v"(qr[2]), "=&v"(qr[3]), "=&v"(qr[4]), "=&v"(qr[5]), "=&v"(qr[6]),
        "=&v"(qr[7]), "=&v"(qr[8]), "=&v"(qr[9]), "=&v"(qr[10]), "=&v"(qr[11]), "=&v"(qr[12])
      : "v"(base), "v"(left)
      : "memory");
}


// one vmcnt(0) for the loads of load_q_slice_split<false>: every register is an in/out operand, so no use of the data
// can be scheduled above the wait (an asm load's destination counts as written when its statement ends)
#define MM_Q13(A) "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]), "+v"(A[8]), "+v"(A[9]), "+v"(A[10]), "+v"(A[11]), "+v"(A[12])
template <int NS>
__device__ __forceinline__ void wait_q_slices(f32x4 (&raw)[NS][13]) {
  if constexpr (NS == 1) asm volatile("s_waitcnt vmcnt(0)" : MM_Q13(raw[0])::"memory");
  if constexpr (NS == 2) asm volatile("s_waitcnt vmcnt(0)" : MM_Q13(raw[0]), MM_Q13(raw[1])::"memory");
  if constexpr (NS == 3) asm volatile("s_waitcnt vmcnt(0)" : MM_Q13(raw[0]), MM_Q13(raw[1]), MM_Q13(raw[2])::"memory");
  if constexpr (NS == 4) asm volatile("s_waitcnt vmcnt(0)" : MM_Q13(raw[0]), MM_Q13(raw[1]), MM_Q13(raw[2]), MM_Q13(raw[3])::"memory");
}
#undef MM_Q13

// WPP = 2 (eval.py-sized calls: fewer pairs than wavefront slots, defaults.yaml:115 batch_size_eval 512): TWO
// wavefronts per pair — wavefront w of the 128-thread workgroup streams the pair's blocks w, w + 2, ... through its own
// ring, the kernel sums of the two meet in LDS once per pair and wavefront 0 pools.  A 200-token document is 4 + 3
// blocks instead of 7 in a row on one wavefront while half of the chip's wavefront slots idle.
template <int NS, int K, int NBUF, bool NT, bool TKL, bool W = false, int WPP = 1>
__global__ void __launch_bounds__(64 * WPP) kernel_pool_split_kernel(const KpArgs a_in) {
  static_assert(!(TKL && W), "the gate is a TK-Sparse feature");
  static_assert(WPP == 1 || (WPP == 2 && !TKL && !W), "two wavefronts per pair: plain TK pooling only");
  const KpArgs a = kp_block_args(a_in);
  static_assert(NS >= 1 && NS <= 4, "parked-chunk step holds at most 4 chunks");
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  const int lane = threadIdx.x & 63;
  const int wv = WPP > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;   // wave-uniform for the compiler too
  constexpr int kWaveLds = NBUF * kSliceBytes + 128;
  char* smem = smem_all + (WPP > 1 ? wv * kWaveLds : 0);
  // WPP = 2: wavefront 1's partial kernel sums, [64 lanes][12], in the head of ITS ring — idle by then, because a
  // two-wavefront workgroup scores exactly one pair (two rings + scratch beside them would be 83 KB: one workgroup per CU)
  float* comb = (float*)(smem_all + kWaveLds);
  const int r = lane & 31, h = lane >> 5;
  const int64_t p0 = (int64_t)blockIdx.x * (WPP > 1 ? 1 : a.pairs_per_wave);
  const int64_t p1 = WPP > 1 ? p0 + 1 : ((p0 + a.pairs_per_wave < a.n_pairs) ? p0 + a.pairs_per_wave : a.n_pairs);
  if (p0 >= p1 || p0 >= a.n_pairs) return;
  constexpr int E = 100 * NS;
  constexpr int RB = E * 4;  // row bytes
  const int D = a.D, Q = a.Q;
  const int nblk_tot = (D + 31) >> 5;
  const int rows_last = D - 32 * (nblk_tot - 1);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  float* rdbuf = (float*)(smem + NBUF * kSliceBytes);  // 32 floats: 1/(|d|+tiny) of the block's rows
  float* wbuf = rdbuf + 32;                            // W: log2 gate of every row of the current document

  uint32_t voff[kSliceInstr];
#pragma unroll
  for (int n = 0; n < kSliceInstr; ++n) {
    int s = 64 * n + lane;
    if (s > 32 * kSC - 1) s = 32 * kSC - 1;
    const int row = s / kSC, c = s - row * kSC;
    voff[n] = (uint32_t)(row * RB + c * 16);
  }
  const uint32_t vmax_tail = (uint32_t)((rows_last - 1) * RB + (kSC - 1) * 16);
  const uint32_t a_off = (uint32_t)(r * (kSC * 16) + h * 32);  // this lane's 32-B A window of step 0
  const uint32_t l_off = (uint32_t)(r * (kSC * 16) + 24 * 16); // the parked chunk of this lane's row

  Rbf rbf;
  load_rbf<K>(a.mu, a.sigma, a.alpha, a.w, rbf);

  const char* dbase = (const char*)a.d;
  // WPP = 2 with float masks handed over as they are: this workgroup's ONE pair; both wavefronts derive the same words
  uint32_t iw[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // validity bits of the document's <= 256 positions
  int ilen = 0;                                   // its effective length (last real token + 1)
  uint32_t iqbits = 0xffffffffu;
  int iqlen = 0;
  const bool inl = WPP == 2 && a.fdm != nullptr;
  if (WPP == 2 && inl) {
    const float* m = a.fdm + p0 * (int64_t)D;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = 64 * i + lane;
      const unsigned long long bal = __ballot(j < D && m[j < D ? j : D - 1] != 0.0f);
      iw[2 * i] = (uint32_t)bal;
      iw[2 * i + 1] = (uint32_t)(bal >> 32);
      if (bal) ilen = 64 * i + 64 - __builtin_clzll(bal);
    }
    const int64_t qrow = a.pair_q ? (int64_t)a.pair_q[p0] : p0 / a.ppq;
    const unsigned long long qb = __ballot(lane < Q && a.fqm[qrow * Q + (lane < Q ? lane : Q - 1)] != 0.0f);
    iqbits = (uint32_t)qb;
    iqlen = qb ? 64 - __builtin_clzll(qb) : 0;
  }
  auto doc_len = [&](int64_t p) -> int {
    if (WPP == 2 && inl) return ilen;
    int len = a.dm.len ? (int)sload_u32(a.dm.len, p) : D;
    return len < 0 ? 0 : (len > D ? D : len);
  };

  int64_t pp = p0;
  int pt = wv, ps = 0, pn = 0;                          // this wavefront's blocks of a pair: wv, wv + WPP, ...
  while (pp < p1 && (pn = (doc_len(pp) + 31) >> 5) <= wv) ++pp;
  int pbuf = 0, cbuf = 0, inflight = 0;
  auto top_up = [&]() {
    while (pp < p1 && inflight < NBUF) {
      const char* g = dbase + (pp * a.d_doc_rows + a.d_row0 + (int64_t)pt * 32) * RB + ps * (kSC * 16);
      issue_slice<NT>(g, voff, vmax_tail, pt == nblk_tot - 1 && rows_last != 32, lds0 + (uint32_t)pbuf * kSliceBytes);
      pbuf = (pbuf + 1 == NBUF) ? 0 : pbuf + 1;
      ++inflight;
      if (++ps == NS) {
        ps = 0;
        pt += WPP;
        if (pt >= pn) {
          pt = wv;
          ++pp;
          while (pp < p1 && (pn = (doc_len(pp) + 31) >> 5) <= wv) ++pp;
        }
      }
    }
  };
  top_up();

  // query tile as bf16 hi / lo B fragments: (6*NS + 1) steps x 4 VGPRs x 2
  bf16x8 qhi[NS][kSplitSteps], qlo[NS][kSplitSteps], qhiL, qloL;
  float rq = 0.0f;
  bool qvalid = false;
  uint32_t qbits = 0xffffffffu;
  int qn = Q, np = 2;  // effective query length; lanes sharing one query token in the epilogue (2 = the MFMA layout)
  int rrows = 0, rtk = 0, rsub = 0;  // redistributed epilogue: rows per lane (0 = MFMA layout), this lane's token and row group
  int64_t cur_q = -1;
  int64_t qi = TKL ? 0 : p0 / a.ppq;
  int64_t q_left = TKL ? 0 : a.ppq - (p0 - qi * a.ppq);

  for (int64_t pair = p0; pair < p1; ++pair) {
    if (TKL) {
      qi = (int64_t)((int)sload_u32(a.chunk_slot, pair) / a.C);
      tkl_publish_slot(a, pair, (doc_len(pair) + 31) >> 5, lane);
    } else if (a.pair_q) {
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
      const char* qrow = (const char*)a.q + (qi * Q + qr) * RB;
      float ss = 0.0f;
      f32x4 park[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
      // all 13 NS loads of the tile are in flight together: one memory round trip per query, not NS
      f32x4 raw[NS][13];
#pragma unroll
      for (int s = 0; s < NS; ++s)
        load_q_slice_split<false>(qrow + s * (kSC * 16) + h * 32, qrow + s * (kSC * 16) + 24 * 16, raw[s]);
      wait_q_slices<NS>(raw);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int p = 0; p < kSplitSteps; ++p) {
          split8(raw[s][2 * p], raw[s][2 * p + 1], qhi[s][p], qlo[s][p]);
          qhi[s][p] = to_agpr(qhi[s][p]);
          qlo[s][p] = to_agpr(qlo[s][p]);
          ss += sumsq4(raw[s][2 * p]) + sumsq4(raw[s][2 * p + 1]);
        }
        if (h == 0) ss += sumsq4(raw[s][12]);  // both halves loaded the parked chunk: count it once
        if (h == (s >> 1)) park[s & 1] = raw[s][12];
      }
      split8(park[0], park[1], qhiL, qloL);
      qhiL = to_agpr(qhiL);
      qloL = to_agpr(qloL);
      ss += __shfl_xor(ss, 32, 64);
      rq = 1.0f / (sqrtf(ss) + 1e-13f);
      const int qlen = (WPP == 2 && inl) ? iqlen : (a.qm.len ? (int)sload_u32(a.qm.len, qi) : Q);
      qvalid = r < Q && r < qlen;
      qbits = (WPP == 2 && inl) ? iqbits : (a.qm.bits ? sload_u32(a.qm.bits, qi) : 0xffffffffu);
      if (a.qm.bits || (WPP == 2 && inl)) qvalid = qvalid && ((qbits >> r) & 1u);
      // queries of <= 21 real tokens: share each query token's epilogue between np = ceil(32 / rows) lanes (see
      // rbf_redistributed): 3 lanes x 11 rows at qn = 20 instead of 2 x 16 with 24 of the 64 lanes idle
      qn = qlen < Q ? (qlen < 0 ? 0 : qlen) : Q;
      rrows = TKL ? 0 : redist_rows(qn);
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
      // plain loads share the vmcnt queue with the LDS-DMA slices, so this wait also lands the prefetched
      // slices: one drain per document (7 blocks of 3 slices at D = 200), refilled by the next top_up()
      const float* gw = a.dw + pair * (int64_t)D;
      for (int j = lane; j < 32 * nb; j += 64) wbuf[j] = j < D ? gate_log2(gw[j]) : -INFINITY;
    }

    for (int t = wv; t < nb; t += WPP) {
      // two accumulators: hi.hi, and the three cross terms (all ~2^-9 of it) together — a third one was 16 more
      // v_accvgpr_read + 8 v_pk_add per block in the epilogue, and this kernel's time is its instruction energy (3.3)
      f32x16 acc_hh = {0}, acc_xl = {0};
      f32x4 park[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
      f32x2 ss2 = {0.0f, 0.0f};
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        top_up();
        wait_slices(inflight - 1);
        const char* buf = smem + cbuf * kSliceBytes;
        // all 13 LDS reads of the slice go out first; the split of step p+1 and the norm update are
        // scheduled between the four MFMAs of step p (the bf16 matrix pipe runs beside the VALU)
        f32x4 x[13];
#pragma unroll
        for (int p = 0; p < kSplitSteps; ++p) {
          x[2 * p] = *(const f32x4*)(buf + a_off + p * 64);
          x[2 * p + 1] = *(const f32x4*)(buf + a_off + p * 64 + 16);
        }
        x[12] = *(const f32x4*)(buf + l_off);
        __builtin_amdgcn_sched_barrier(0);
        // The slice is in registers: its slot goes back to the producer NOW, before the split / MFMA work, so that three
        // slices instead of two are in flight while this one is computed (the kernel waits on memory most of the time
        // with one wavefront per SIMD).  Not after a block's LAST slice: the epilogue borrows that slot as scratch.
        const bool early = s + 1 < NS;
        if (early) {
          cbuf = (cbuf + 1 == NBUF) ? 0 : cbuf + 1;
          --inflight;
          top_up();
        }
        bf16x8 ah, al;
        split8(x[0], x[1], ah, al);
#pragma unroll
        for (int p = 0; p < kSplitSteps; ++p) {
          bf16x8 nh = ah, nl = al;
          if (p + 1 < kSplitSteps) split8(x[2 * p + 2], x[2 * p + 3], nh, nl);
          acc_hh = mfma_bf16(ah, qhi[s][p], acc_hh);
          acc_xl = mfma_bf16(al, qhi[s][p], acc_xl);
          acc_xl = mfma_bf16(ah, qlo[s][p], acc_xl);
          if (MM_KP_LOLO) acc_xl = mfma_bf16(al, qlo[s][p], acc_xl);
          {
            const f32x2 a0 = {x[2 * p][0], x[2 * p][1]}, a1 = {x[2 * p][2], x[2 * p][3]};
            const f32x2 b0 = {x[2 * p + 1][0], x[2 * p + 1][1]}, b1 = {x[2 * p + 1][2], x[2 * p + 1][3]};
            ss2 += a0 * a0;
            ss2 += a1 * a1;
            ss2 += b0 * b0;
            ss2 += b1 * b1;
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, kShadowValu, 0);  // VALU in its shadow
          }
          ah = nh;
          al = nl;
        }
        const f32x4 xl = x[12];
        if (h == 0) ss2 += f32x2{xl[0] * xl[0] + xl[1] * xl[1], xl[2] * xl[2] + xl[3] * xl[3]};
        if (h == (s >> 1)) park[s & 1] = xl;
        if (!early) {
          cbuf = (cbuf + 1 == NBUF) ? 0 : cbuf + 1;
          --inflight;
        }
      }
      {
        bf16x8 ah, al;
        split8(park[0], park[1], ah, al);
        acc_hh = mfma_bf16(ah, qhiL, acc_hh);
        acc_xl = mfma_bf16(al, qhiL, acc_xl);
        acc_xl = mfma_bf16(ah, qloL, acc_xl);
        if (MM_KP_LOLO) acc_xl = mfma_bf16(al, qloL, acc_xl);
      }
      float ss = ss2[0] + ss2[1];
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = acc_hh[i] + acc_xl[i];
      // document-token norms: lane (r,h) summed its K-halves of row r (from the fp32 values)
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
      uint32_t va = ex;
      if (WPP == 2 && inl) {
        uint32_t wsel = iw[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) wsel = t == i ? iw[i] : wsel;     // (t is wave-uniform; keeps the words in registers)
        va = wsel & ex;
      } else if (a.dm.bits) {
        va = sload_u32(a.dm.bits, pair * nblk_tot + t) & ex;
      }
      if constexpr (TKL) {
        tkl_block<K>(a.ps_out + pair * (20 * (int64_t)Q * (K + 1)), Q, t, r, h, acc, rdr, rq, va >> (4 * h), rbf);
      } else if (np > 2) {
        // transpose the scaled tile through the ring slot just consumed (free until the next top_up()):
        // T[query token][32 rows]; then np lanes per token evaluate 32 / np rows each
        float* T = (float*)(smem + (cbuf == 0 ? NBUF - 1 : cbuf - 1) * kSliceBytes);
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
        rbf_block<K, true>(pk2, acc, rdr, rq, va, h, rbf, lw);
      } else {
        rbf_block<K>(pk2, acc, rdr, rq, va, h, rbf);
      }
    }
    if constexpr (WPP == 2) {
      // both wavefronts hold partial sums in the same lane layout: wavefront 1 hands its twelve to wavefront 0 through LDS.
      // Barriers that wait for LDS only (a __syncthreads() would drain the LDS-DMA prefetch of the next pair with vmcnt(0)).
      if (wv == 1) {
#pragma unroll
        for (int v = 0; v < 3; ++v)
          *(f32x4*)(comb + lane * 12 + 4 * v) = f32x4{pk2[2 * v][0], pk2[2 * v][1], pk2[2 * v + 1][0], pk2[2 * v + 1][1]};
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (wv == 0) {
#pragma unroll
        for (int v = 0; v < 3; ++v) {
          const f32x4 o = *(const f32x4*)(comb + lane * 12 + 4 * v);
          pk2[2 * v] += f32x2{o[0], o[1]};
          pk2[2 * v + 1] += f32x2{o[2], o[3]};
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // comb may be rewritten for the next pair
    }
    if (!TKL && wv == 0) {
      float pk[kMaxK];
      if (np > 2) {  // np consecutive lanes hold the partial sums of one query token
        pk_get<K>(pk, pk2, rbf);
        redist_reduce<K>(pk, np, lane);
        const bool count = rsub == 0 && rtk < qn && ((qbits >> rtk) & 1u);
        finish_pool<K>(a, pair, pk, count, lane, rbf, rsub == 0 ? rtk : -1);
      } else {
        pk_get<K>(pk, pk2, rbf);
#pragma unroll
        for (int k = 0; k < K; ++k) pk[k] += __shfl_xor(pk[k], 32, 64);
        finish_pool<K>(a, pair, pk, qvalid && lane < 32, lane, rbf, lane < 32 ? lane : -1);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// TKL stage 1, grouped (the default TKL path for E == 100*NS, Q <= 32).
//
// A chunk contributes its 40 centre rows = one full 32-row MFMA block + an 8-row remainder that costs
// a full block of split / MFMA / LDS-DMA work.  Up to four consecutive packed chunks of the SAME
// document (same query) are therefore treated as one virtual document of 40n rows: n = 4 gives exactly
// five full blocks instead of eight.  LDS-DMA sources are per-lane addresses anyway, so virtual row i
// simply maps to row 5 + i % 40 of chunk p + i / 40; position pairs never straddle a chunk (40 is even).
// Every pair of every chunk of the run is written (zeros where the mask says padding), so stage 2 needs
// no per-chunk block count.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void sload4_u32(const void* base, int64_t idx, uint32_t (&v)[4]) {
  const uint32_t* p = (const uint32_t*)base + idx;
  asm volatile(
      "s_load_dword %0, %4, 0x0\n\t"
      "s_load_dword %1, %4, 0x4\n\t"
      "s_load_dword %2, %4, 0x8\n\t"
      "s_load_dword %3, %4, 0xc\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(v[0]), "=&s"(v[1]), "=&s"(v[2]), "=&s"(v[3])
      : "s"(p));
}

// Epilogue of block t of a run: pairs of virtual rows (i0, i0 + 1), i0 = 32t + rowof(2ip) + 4h, belong
// to chunk i0 / 40, pair (i0 % 40) / 2.  rows_blk = rows of this block that exist (multiple of 8).
//
// The pair sums leave through LDS.  A lane owns one query token, so writing them straight from the registers is
// 24 store instructions per block whose 40 active lanes each put 16 B at a 48-B stride (measured by removal on
// config 3: 49 of stage 1's 189 us).  The 16 pair rows of a block are ONE contiguous region of the pair buffer
// ((chunk * 20 + pair) * Q * 12 floats is linear in the run's pair index), so each half of the block (8 pair rows,
// <= 12 KiB) is staged in the ring slot the block just freed and written out as one contiguous run of qlim * 48 B per
// store instruction: 16 stores per block, every one a full-width burst.  (Measured: 189 -> 184 us — the cost is the
// write traffic in the read stream and the stores sitting in the in-order vmcnt queue, not the shape of the stores.)
template <int K>
__device__ __forceinline__ void tkl_block_run(float* ps_run, float* stage, int lane, int Q, int qlim, int t, int rows_blk, int r,
                                              int h, const f32x16& acc, const float (&rdr)[16], float rq, uint32_t vbits,
                                              const Rbf& rbf) {
  constexpr int KC = K + 1;
  constexpr int KP = (K + 1) / 2;
  static_assert(KC == 12 && K % 2 == 1, "K kernels + the count channel fill three float4s");
  f32x4* stage4 = (f32x4*)stage;                                  // [8 pair rows][Q][3]
  f32x4* out4 = (f32x4*)ps_run + (int64_t)16 * t * (3 * Q);       // pair row g of the run at g * Q * 12 floats
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    if (16 * hf >= rows_blk) break;                               // wave-uniform
#pragma unroll
    for (int iq = 0; iq < 4; ++iq) {
      const int ip = 4 * hf + iq;
      if (rowof(2 * ip) < rows_blk) {  // wave-uniform (rows_blk is a multiple of 8, rowof(2ip) + 4h + 1 < its 8-row group end)
        f32x2 o2[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) o2[k] = f32x2{0.0f, 0.0f};
        float cnt = 0.0f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int i = 2 * ip + half;
          float c = (acc[i] * rq) * rdr[i];
          c = ((vbits >> rowof(i)) & 1u) ? c : 1.0e5f;  // masked: every kernel underflows to exactly 0 (:194)
          const f32x2 cc = {c, c};
          f32x2 any2 = {0.0f, 0.0f};
#pragma unroll
          for (int kp = 0; kp < KP; ++kp) {
            const f32x2 sv = cc * rbf.sq2[kp] - rbf.msq2[kp];
            const f32x2 av = -(sv * sv);
            const f32x2 e = {__builtin_amdgcn_exp2f(av[0]), __builtin_amdgcn_exp2f(av[1])};
            o2[kp] += e;
            any2 += e;
          }
          cnt += (any2[0] + any2[1]) != 0.0f ? 1.0f : 0.0f;  // (:210)
        }
        const int j8 = ((rowof(2 * ip) + 4 * h) >> 1) - 8 * hf;   // pair row inside this half (0..7)
        if (r < qlim) {  // query tokens past the query's effective length are masked in stage 2 (:248): never written, never read
          f32x4* dst = stage4 + (j8 * Q + r) * 3;
          dst[0] = f32x4{o2[0][0], o2[0][1], o2[1][0], o2[1][1]};
          dst[1] = f32x4{o2[2][0], o2[2][1], o2[3][0], o2[3][1]};
          dst[2] = f32x4{o2[KP - 2][0], o2[KP - 2][1], o2[KP - 1][0], cnt};
        }
      }
    }
    // the half's pair rows that exist: rows_blk / 2 - 8 hf of them (a multiple of 4), one contiguous store each
    const int npr = rows_blk / 2 - 8 * hf < 8 ? rows_blk / 2 - 8 * hf : 8;
    for (int l = lane; l < 3 * qlim; l += 64) {                  // (one pass for queries of <= 21 tokens)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < npr) out4[(int64_t)(8 * hf + j) * (3 * Q) + l] = stage4[j * (3 * Q) + l];
    }
  }
}

template <int NS, int K, int NBUF, bool NT, bool COS>
__global__ void __launch_bounds__(64) tkl_stage1_run_kernel(const KpArgs a) {
  static_assert(NS >= 1 && NS <= 4, "parked-chunk step holds at most 4 chunks");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int r = lane & 31, h = lane >> 5;
  const int64_t p0 = (int64_t)blockIdx.x * a.pairs_per_wave;
  const int64_t p1 = (p0 + a.pairs_per_wave < a.n_pairs) ? p0 + a.pairs_per_wave : a.n_pairs;
  if (p0 >= p1) return;
  constexpr int E = 100 * NS;
  constexpr int RB = E * 4;  // row bytes
  const int Q = a.Q;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  float* rdbuf = (float*)(smem + NBUF * kSliceBytes);

  // slice image: slot s = 64n + lane -> (row, 16-B column) of the 32-row x 25-chunk slice
  int srow[kSliceInstr];
  uint32_t scol[kSliceInstr];
#pragma unroll
  for (int n = 0; n < kSliceInstr; ++n) {
    int s = 64 * n + lane;
    if (s > 32 * kSC - 1) s = 32 * kSC - 1;
    srow[n] = s / kSC;
    scol[n] = (uint32_t)((s - srow[n] * kSC) * 16);
  }
  const uint32_t a_off = (uint32_t)(r * (kSC * 16) + h * 32);
  const uint32_t l_off = (uint32_t)(r * (kSC * 16) + 24 * 16);

  Rbf rbf;
  if constexpr (!COS) load_rbf<K, false>(a.mu, a.sigma, nullptr, nullptr, rbf);   // (the cosine hand-off evaluates no kernels here)

  const char* dbase = (const char*)a.d;
  // run of up to 4 consecutive packed chunks of one document starting at chunk p (inside [p, p1))
  auto run_len = [&](int64_t p) -> int {
    if (p + 4 <= a.n_pairs) {
      uint32_t sl[4];
      sload4_u32(a.chunk_slot, p, sl);
      const int b0 = (int)sl[0] / a.C;
      int n = 1;
      while (n < 4 && p + n < p1 && (int)sl[n] / a.C == b0 && (!COS || (int)sl[n] == (int)sl[0] + n)) ++n;
      return n;
    }
    const int s0 = (int)sload_u32(a.chunk_slot, p), b0 = s0 / a.C;
    int n = 1;
    while (n < 4 && p + n < p1) {
      const int sn = (int)sload_u32(a.chunk_slot, p + n);
      if (sn / a.C != b0 || (COS && sn != s0 + n)) break;
      ++n;
    }
    return n;
  };

  // ---- producer cursor over (run, block, slice) ------------------------------------------------
  int64_t pp = p0;
  int prun = run_len(pp), pn = (40 * prun + 31) >> 5, pt = 0, ps = 0;
  uint32_t vrun[kSliceInstr];
  int pbuf = 0, cbuf = 0, inflight = 0;
  // The epilogue's global stores share the in-order vmcnt queue with the LDS-DMA slices.  Counting only slices made
  // the first wait of every block (queue: [slice a][slice b][16 stores][slice c], wait for a with vmcnt(2 x 13))
  // also wait for slice b — one slice of prefetch depth lost per block.  `pre` = slices in the queue older than the
  // last store burst, `nst` = stores of that burst: the count of operations younger than the oldest slice is
  // 13 (inflight - 1) + nst while pre > 0, and the burst has retired before any younger slice once pre == 0.
  // (Counting too FEW younger operations only waits longer; counting too many would read a slice before it landed.)
  int pre = 0, nst = 0;
  auto top_up = [&]() {
    while (pp < p1 && inflight < NBUF) {
      if (ps == 0) {  // source offsets of block pt of the run: virtual row i -> chunk i / 40, row 5 + i % 40
        const int last = 40 * prun - 1;
#pragma unroll
        for (int n = 0; n < kSliceInstr; ++n) {
          int i = 32 * pt + srow[n];
          i = i < last ? i : last;  // rows past the run re-read its last row (excluded by the masks)
          const int ci = (i * 205) >> 13;
          vrun[n] = (uint32_t)((ci * 50 + 5 + (i - 40 * ci)) * RB) + scol[n];
        }
      }
      const char* g = dbase + pp * 50 * (int64_t)RB + ps * (kSC * 16);
      issue_slice<NT>(g, vrun, 0u, false, lds0 + (uint32_t)pbuf * kSliceBytes);
      pbuf = (pbuf + 1 == NBUF) ? 0 : pbuf + 1;
      ++inflight;
      if (++ps == NS) {
        ps = 0;
        if (++pt == pn) {
          pt = 0;
          pp += prun;
          if (pp < p1) {
            prun = run_len(pp);
            pn = (40 * prun + 31) >> 5;
          }
        }
      }
    }
  };
  top_up();

  bf16x8 qhi[NS][kSplitSteps], qlo[NS][kSplitSteps], qhiL, qloL;
  float rq = 0.0f;
  int64_t cur_q = -1;
  int qlim = Q;  // effective length of the current query (a.qm.len, when the caller resolved the query mask)

  for (int64_t pair = p0; pair < p1;) {
    const int crun = run_len(pair);
    const int nb = (40 * crun + 31) >> 5;
    for (int k = 0; k < crun; ++k) tkl_publish_slot(a, pair + k, 2, lane);  // every pair row of every chunk is written
    const int slot0 = (int)sload_u32(a.chunk_slot, pair);
    const int64_t qi = (int64_t)(slot0 / a.C);
    if (qi != cur_q) {
      cur_q = qi;
      const int qr = r < Q ? r : Q - 1;
      const char* qrow = (const char*)a.q + (qi * Q + qr) * RB;
      float ss = 0.0f;
      f32x4 park[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
      // all 13 NS loads of the tile are in flight together: one memory round trip per query, not NS
      f32x4 raw[NS][13];
#pragma unroll
      for (int s = 0; s < NS; ++s)
        load_q_slice_split<false>(qrow + s * (kSC * 16) + h * 32, qrow + s * (kSC * 16) + 24 * 16, raw[s]);
      wait_q_slices<NS>(raw);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int p = 0; p < kSplitSteps; ++p) {
          split8(raw[s][2 * p], raw[s][2 * p + 1], qhi[s][p], qlo[s][p]);
          qhi[s][p] = to_agpr(qhi[s][p]);
          qlo[s][p] = to_agpr(qlo[s][p]);
          ss += sumsq4(raw[s][2 * p]) + sumsq4(raw[s][2 * p + 1]);
        }
        if (h == 0) ss += sumsq4(raw[s][12]);
        if (h == (s >> 1)) park[s & 1] = raw[s][12];
      }
      split8(park[0], park[1], qhiL, qloL);
      qhiL = to_agpr(qhiL);
      qloL = to_agpr(qloL);
      ss += __shfl_xor(ss, 32, 64);
      rq = 1.0f / (sqrtf(ss) + 1e-13f);
      if (a.qm.len) {
        const int ql = (int)sload_u32(a.qm.len, qi);
        qlim = ql < 0 ? 0 : (ql > Q ? Q : ql);
      }
    }
    float* ps_run = a.ps_out + pair * (20 * (int64_t)Q * (K + 1));

    for (int t = 0; t < nb; ++t) {
      f32x16 acc_hh = {0}, acc_lh = {0}, acc_xl = {0};
      f32x4 park[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
      f32x2 ss2 = {0.0f, 0.0f};
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        top_up();
        if (pre > 0) {
          wait_vm(kSliceInstr * (inflight - 1) + nst);
          --pre;
        } else {
          nst = 0;
          wait_slices(inflight - 1);
        }
        const char* buf = smem + cbuf * kSliceBytes;
        f32x4 x[13];
#pragma unroll
        for (int p = 0; p < kSplitSteps; ++p) {
          x[2 * p] = *(const f32x4*)(buf + a_off + p * 64);
          x[2 * p + 1] = *(const f32x4*)(buf + a_off + p * 64 + 16);
        }
        x[12] = *(const f32x4*)(buf + l_off);
        __builtin_amdgcn_sched_barrier(0);
        // as in kernel_pool_split_kernel: the slot goes back before the split / MFMA work — after EVERY slice with the cosine
        // hand-off, whose epilogue needs no scratch (three slices in flight throughout: 159.8 -> 149.3 us with two of three
        // slices released early, config 3's full documents)
        const bool early = COS || s + 1 < NS;
        if (early) {
          cbuf = (cbuf + 1 == NBUF) ? 0 : cbuf + 1;
          --inflight;
          top_up();
        }
        bf16x8 ah, al;
        split8(x[0], x[1], ah, al);
#pragma unroll
        for (int p = 0; p < kSplitSteps; ++p) {
          bf16x8 nh = ah, nl = al;
          if (p + 1 < kSplitSteps) split8(x[2 * p + 2], x[2 * p + 3], nh, nl);
          acc_hh = mfma_bf16(ah, qhi[s][p], acc_hh);
          acc_lh = mfma_bf16(al, qhi[s][p], acc_lh);
          acc_xl = mfma_bf16(ah, qlo[s][p], acc_xl);
          if (MM_TKL_LOLO) acc_xl = mfma_bf16(al, qlo[s][p], acc_xl);
          {
            const f32x2 a0 = {x[2 * p][0], x[2 * p][1]}, a1 = {x[2 * p][2], x[2 * p][3]};
            const f32x2 b0 = {x[2 * p + 1][0], x[2 * p + 1][1]}, b1 = {x[2 * p + 1][2], x[2 * p + 1][3]};
            ss2 += a0 * a0;
            ss2 += a1 * a1;
            ss2 += b0 * b0;
            ss2 += b1 * b1;
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, kShadowValu, 0);
          }
          ah = nh;
          al = nl;
        }
        const f32x4 xl = x[12];
        if (h == 0) ss2 += f32x2{xl[0] * xl[0] + xl[1] * xl[1], xl[2] * xl[2] + xl[3] * xl[3]};
        if (h == (s >> 1)) park[s & 1] = xl;
        if (!early) {
          cbuf = (cbuf + 1 == NBUF) ? 0 : cbuf + 1;
          --inflight;
        }
      }
      {
        bf16x8 ah, al;
        split8(park[0], park[1], ah, al);
        acc_hh = mfma_bf16(ah, qhiL, acc_hh);
        acc_lh = mfma_bf16(al, qhiL, acc_lh);
        acc_xl = mfma_bf16(ah, qloL, acc_xl);
        if (MM_TKL_LOLO) acc_xl = mfma_bf16(al, qloL, acc_xl);
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
      // validity bits of virtual rows 32t .. 32t+31: they span at most two chunks of the run
      const int i0 = 32 * t;
      const int c0 = (i0 * 205) >> 13, r0 = i0 - 40 * c0;
      uint32_t va;
      {
        const uint32_t w0 = sload_u32(a.dm.bits, (pair + c0) * 2), w1 = sload_u32(a.dm.bits, (pair + c0) * 2 + 1);
        const unsigned long long b0 = ((unsigned long long)(w1 & 0xffu) << 32) | w0;
        va = (uint32_t)(b0 >> r0);
        const int n0 = 40 - r0;  // rows of this block that come from chunk c0 (when < 32)
        if (n0 < 32 && c0 + 1 < crun) va |= sload_u32(a.dm.bits, (pair + c0 + 1) * 2) << n0;
      }
      const int rows_blk = 40 * crun - i0 < 32 ? 40 * crun - i0 : 32;
      if (rows_blk < 32) va &= (1u << rows_blk) - 1u;
      if constexpr (COS) {
        // Cosine rows are indexed by document and position: document b owns cos_out[b * C * 40 * Q ...], and inside it the
        // row of position n = c * 40 + p (chunk c, centre token p) starts at n * ql with ql = the query's effective
        // length — only the ql real tokens are stored.  The chunks of a run sit in consecutive slots, so virtual row iv of
        // the run is position (slot0 % C) * 40 + iv of its document.  Lane (token r, half h) stores its 16 rows directly:
        // 16 store instructions of <= 2 x ql x 4 B per block.  Measured by removal the stores cost ~20 us of stage 1's 160
        // on config 3's full documents WHATEVER their shape or cache policy — transposed through LDS into 3 full-width
        // 16-B stores per block: 160.6 vs 160.5 us; ordinary / nt / sc0 sc1: 257.7 / 259.0 / 260.4 us per call — it is the
        // write traffic inside the read stream, so fewer bytes is what helps (no columns past ql: 24.5 MB, not 42.6),
        // and the direct form leaves the ring slot free for an early hand-back.
        if (r < qlim) {
          float* dst = a.cos_out + qi * ((int64_t)a.C * 40 * Q) + ((int64_t)(slot0 - (int)qi * a.C) * 40 + 32 * t + 4 * h) * qlim + r;
          const uint32_t vbits = va >> (4 * h);
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            if (rowof(i) < rows_blk) {  // wave-uniform: rows_blk is a multiple of 8 and rowof(i) + 4h stays in rowof(i)'s group of 8
              const float c = (acc[i] * rq) * rdr[i];
              dst[rowof(i) * qlim] = ((vbits >> rowof(i)) & 1u) ? c : 1.0e5f;   // masked: every kernel underflows to exactly 0
            }
          }
        }
        if (qlim > 0) {
          nst = rows_blk / 2;
          pre = inflight;
        }
      } else {
        // staging area: the ring slot this block's last slice just left (free until the next top_up())
        float* stage = (float*)(smem + (cbuf == 0 ? NBUF - 1 : cbuf - 1) * kSliceBytes);
        tkl_block_run<K>(ps_run, stage, lane, Q, qlim, t, rows_blk, r, h, acc, rdr, rq, va >> (4 * h), rbf);
        if (qlim > 0) {
          const int burst = ((3 * qlim + 63) >> 6) * (rows_blk / 2);      // tkl_block_run: one store per pair row and 64-lane pass
          nst = burst;
          pre = inflight;
        }
      }
    }
    pair += crun;
  }
}

// ---------------------------------------------------------------------------------------------
// generic kernel: one wavefront per pair, direct fragment loads, any E % 4 == 0, any Q / D.
// ---------------------------------------------------------------------------------------------
template <int K, bool TKL, bool W = false>
__global__ void __launch_bounds__(64) kernel_pool_generic_kernel(const KpArgs a_in) {
  const KpArgs a = kp_block_args(a_in);
  __shared__ float rdbuf[32];
  __shared__ float lwbuf[32];
  const int lane = threadIdx.x;
  const int r = lane & 31, h = lane >> 5;
  const int64_t pair = blockIdx.x;
  if (pair >= a.n_pairs) return;
  const int D = a.D, Q = a.Q, E = a.E;
  const int64_t rowb = (int64_t)E * 4;
  const int64_t qi = TKL ? (int64_t)(a.chunk_slot[pair] / a.C) : (a.pair_q ? (int64_t)a.pair_q[pair] : pair / a.ppq);
  const int nblk_tot = (D + 31) >> 5;
  const int qwords = (Q + 31) >> 5;
  const int nch = E >> 2;
  int len = a.dm.len ? a.dm.len[pair] : D;
  len = len < 0 ? 0 : (len > D ? D : len);
  const int nb = (len + 31) >> 5;
  if (TKL) tkl_publish_slot(a, pair, nb, lane);
  const int qlen = a.qm.len ? a.qm.len[qi] : Q;
  const char* dbase = (const char*)a.d + (pair * a.d_doc_rows + a.d_row0) * rowb;
  const char* qbase = (const char*)a.q + qi * Q * rowb;
  // K is the number of kernel slots this instantiation evaluates, a.K <= K the run-time kernel count
  // (K = 11 = a.K for the reference's configs; the K = 32 instantiation serves every other count)
  const int nk = a.K < K ? a.K : K;
  Rbf rbf;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const bool real = k < nk;
    const float sg = real ? a.sigma[k] : 1.0f;
    rbf.mu[k] = real ? a.mu[k] : 0.0f;
    rbf.c2[k] = -1.4426950408889634f / (2.0f * sg * sg);
    rbf.alpha[k] = (real && a.alpha) ? a.alpha[k] : 1.0f;
    rbf.w[k] = (real && a.w) ? a.w[k] : 0.0f;
  }
  pack_rbf<K>(rbf, nk);
  float tot[kMaxK];
#pragma unroll
  for (int k = 0; k < kMaxK; ++k) tot[k] = 0.0f;

  for (int n = 0; n < qwords; ++n) {
    const int qtok = 32 * n + r;
    const int qr = qtok < Q ? qtok : Q - 1;
    bool qvalid = qtok < Q && qtok < qlen;
    if (a.qm.bits) qvalid = qvalid && ((a.qm.bits[qi * qwords + n] >> r) & 1u);
    const char* qrow = qbase + qr * rowb;
    float qss = 0.0f;
    for (int c = h; c < nch; c += 2) {
      const f32x4 v = *(const f32x4*)(qrow + c * 16);
      qss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    qss += __shfl_xor(qss, 32, 64);
    const float rq = 1.0f / (sqrtf(qss) + 1e-13f);
    f32x2 pk2[kMaxK / 2];
#pragma unroll
    for (int k = 0; k < kMaxK / 2; ++k) pk2[k] = f32x2{0.0f, 0.0f};
    for (int t = 0; t < nb; ++t) {
      const int drow = 32 * t + r;
      const char* dr = dbase + (drow < D ? drow : D - 1) * rowb;
      f32x16 acc = {0};
      float ss = 0.0f;
      for (int c = 0; c < nch; c += 2) {
        const int cc = c + h;
        f32x4 av = {0, 0, 0, 0}, bv = {0, 0, 0, 0};
        if (cc < nch) {
          av = *(const f32x4*)(dr + cc * 16);
          bv = *(const f32x4*)(qrow + cc * 16);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
          ss += av[j] * av[j];
        }
      }
      ss += __shfl_xor(ss, 32, 64);
      __syncthreads();
      if (h == 0) rdbuf[r] = 1.0f / (sqrtf(ss) + 1e-13f);
      if (W && h == 1) lwbuf[r] = drow < D ? gate_log2(a.dw[pair * (int64_t)D + drow]) : -INFINITY;
      __syncthreads();
      float rdr[16], lw[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) rdr[i] = rdbuf[rowof(i) + 4 * h];
#pragma unroll
      for (int i = 0; i < 16; ++i) lw[i] = W ? lwbuf[rowof(i) + 4 * h] : 0.0f;
      const int rem = len - 32 * t;
      const uint32_t ex = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
      const uint32_t va = a.dm.bits ? (a.dm.bits[pair * nblk_tot + t] & ex) : ex;
      if constexpr (TKL)
        tkl_block<K>(a.ps_out + pair * (20 * (int64_t)Q * (K + 1)), Q, t, qtok, h, acc, rdr, rq, va >> (4 * h), rbf);
      else
        rbf_block<K, W>(pk2, acc, rdr, rq, va, h, rbf, lw);
    }
    if (TKL) continue;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k < nk) {
        const float pkk = pk2[k >> 1][k & 1];
        const float v = pkk + __shfl_xor(pkk, 32, 64);
        if (a.pooled && h == 0 && qtok < Q) a.pooled[(pair * Q + qtok) * nk + k] = v;
        float lg = __logf(fmaxf(v * rbf.alpha[k], a.clamp_min));
        tot[k] += wave_sum((qvalid && h == 0) ? lg : 0.0f);
      }
    }
  }
  if (TKL) return;
  float total = 0.0f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (k < nk) {
      if (a.per_kernel && lane == 0) a.per_kernel[pair * (int64_t)nk + k] = tot[k];
      total += rbf.w[k] * tot[k];
    }
  }
  if (lane == 0) a.out[pair] = total;
}

template <int K, bool TKL, bool W = false>
static int launch_stream(const KpArgs& a0, hipStream_t stream) {
  KpArgs a = a0;
#ifndef MM_KP_WPC
#define MM_KP_WPC 4            // -DMM_KP_WPC=2 / 3: fewer, longer wavefront streams (A/B builds, tools/build_variant.sh)
#endif
#ifndef MM_TKL_WPC             // the same two knobs for TKL's stage 1 alone (it is bound by its HBM stream, not by its wavefronts:
#define MM_TKL_WPC MM_KP_WPC   // three per CU are as fast as four, profiles/r05_experiments/wavefronts_per_cu_tk_tkl.txt)
#endif
#ifndef MM_TKL_NBUF
#define MM_TKL_NBUF 3
#endif
  constexpr int NBUF = TKL ? MM_TKL_NBUF : 3;
  const int lds = NBUF * kSliceBytes + 128 + (W ? 128 * ((a.D + 31) >> 5) : 0);
  int64_t waves = (int64_t)kCUs * (TKL ? MM_TKL_WPC : MM_KP_WPC);  // one wavefront per SIMD: the fp32 MFMA pipe is the co-limiter
  if (waves > a.n_pairs) waves = a.n_pairs;
  a.pairs_per_wave = (a.n_pairs + waves - 1) / waves;
  waves = (a.n_pairs + a.pairs_per_wave - 1) / a.pairs_per_wave;
  const dim3 grid((unsigned)waves, (unsigned)(a.n_md > 0 ? a.n_mblk : 1)), block(64);
  // MM_KP_F32MFMA=1 selects the exact-f32 MFMA kernel (A/B runs, tools/bench_kernel_pool.py);
  // the default is the split-bf16 kernel (same numerics class, 4x less matrix-pipe time)
  if (env().kp_f32mfma && !W && !a.pair_q && a.n_md == 0) {
    if (a.E == 100)
      hipLaunchKernelGGL((kernel_pool_stream_kernel<1, K, NBUF, true, TKL>), grid, block, lds, stream, a);
    else if (a.E == 200)
      hipLaunchKernelGGL((kernel_pool_stream_kernel<2, K, NBUF, true, TKL>), grid, block, lds, stream, a);
    else
      hipLaunchKernelGGL((kernel_pool_stream_kernel<3, K, NBUF, true, TKL>), grid, block, lds, stream, a);
    return check_launch("kernel_pool_stream_kernel");
  }
  // No non-temporal hint on these K-sliced streams: the 400-B row pieces of neighbouring slices share cache
  // lines, and with `nt` the shared lines came from HBM twice (FETCH_SIZE 1.22 x the padded bytes, 1.08 x without).
  if constexpr (TKL) {  // grouped runs of chunks (tkl_stage1_run_kernel)
    // the cosine hand-off streams whole rows since round 6 (tkl_stage1_rows.hip).  A/B: MM_TKL_STAGE1_KSPLIT=1 = two K-splitting wavefronts per
    // workgroup (tkl_stage1_ksplit.hip: correct, not faster), MM_TKL_STAGE1_SLICES=1 = the K-sliced ring of rounds 2-5 below
    if (a.cos_out && env().tkl_stage1_ksplit && tkl_stage1_ksplit_supported(a.Q, a.E)) return tkl_stage1_ksplit_launch(a, stream);
    if (a.cos_out && !env().tkl_stage1_slices && tkl_stage1_rows_supported(a.Q, a.E)) return tkl_stage1_rows_launch(a, stream);
    if (a.cos_out) {
      if (a.E == 100)
        hipLaunchKernelGGL((tkl_stage1_run_kernel<1, K, NBUF, false, true>), grid, block, lds, stream, a);
      else if (a.E == 200)
        hipLaunchKernelGGL((tkl_stage1_run_kernel<2, K, NBUF, false, true>), grid, block, lds, stream, a);
      else
        hipLaunchKernelGGL((tkl_stage1_run_kernel<3, K, NBUF, false, true>), grid, block, lds, stream, a);
      return check_launch("tkl_stage1_run_kernel<cos>");
    }
    if (a.E == 100)
      hipLaunchKernelGGL((tkl_stage1_run_kernel<1, K, NBUF, false, false>), grid, block, lds, stream, a);
    else if (a.E == 200)
      hipLaunchKernelGGL((tkl_stage1_run_kernel<2, K, NBUF, false, false>), grid, block, lds, stream, a);
    else
      hipLaunchKernelGGL((tkl_stage1_run_kernel<3, K, NBUF, false, false>), grid, block, lds, stream, a);
    return check_launch("tkl_stage1_run_kernel");
  } else {
    if constexpr (!W) {
      // fewer pairs than half the wavefront slots (eval.py-sized calls): two wavefronts per pair
      if (a.n_pairs * 2 <= (int64_t)kCUs * 4 && a.n_md == 0 && a.D > 32) {
        KpArgs b = a;
        b.pairs_per_wave = 1;
        const int lds2 = 2 * (NBUF * kSliceBytes + 128);
        const dim3 grid2((unsigned)a.n_pairs), block2(128);
        if (a.E == 100) {
          (void)hipFuncSetAttribute((const void*)kernel_pool_split_kernel<1, K, NBUF, false, false, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
          hipLaunchKernelGGL((kernel_pool_split_kernel<1, K, NBUF, false, false, false, 2>), grid2, block2, lds2, stream, b);
        } else if (a.E == 200) {
          (void)hipFuncSetAttribute((const void*)kernel_pool_split_kernel<2, K, NBUF, false, false, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
          hipLaunchKernelGGL((kernel_pool_split_kernel<2, K, NBUF, false, false, false, 2>), grid2, block2, lds2, stream, b);
        } else {
          (void)hipFuncSetAttribute((const void*)kernel_pool_split_kernel<3, K, NBUF, false, false, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
          hipLaunchKernelGGL((kernel_pool_split_kernel<3, K, NBUF, false, false, false, 2>), grid2, block2, lds2, stream, b);
        }
        return check_launch("kernel_pool_split_kernel<2 wavefronts per pair>");
      }
    }
    if (a.fdm) return set_error(MM_ELAUNCH, "kernel_pool: float masks were left to a kernel that does not read them (internal)");
    if (a.E == 100)
      hipLaunchKernelGGL((kernel_pool_split_kernel<1, K, NBUF, false, false, W>), grid, block, lds, stream, a);
    else if (a.E == 200)
      hipLaunchKernelGGL((kernel_pool_split_kernel<2, K, NBUF, false, false, W>), grid, block, lds, stream, a);
    else
      hipLaunchKernelGGL((kernel_pool_split_kernel<3, K, NBUF, false, false, W>), grid, block, lds, stream, a);
    return check_launch("kernel_pool_split_kernel");
  }
}

bool kp_stream_supported(int Q, int E) { return Q <= 32 && (E == 100 || E == 200 || E == 300); }

// the cosine hand-off exists on the grouped-run kernel only (the exact-f32 A/B kernel and the generic kernel emit pair sums)
bool tkl_cos_supported(int Q, int E) { return !env().kp_generic && !env().kp_f32mfma && !env().tkl_pairsums && kp_stream_supported(Q, E); }

// TKL stage 1 entry (called from tkl.hip): chunks [P,50,E] -> ps_out [P,20,Q,12]
int tkl_stage1_stream(const float* q_ctx, const float* chunks, PackedMask dm, const int32_t* q_len,
                      const int32_t* chunk_slot, int C, const float* mu, const float* sigma, float* ps_out, int64_t P,
                      int Q, int E, int32_t* slot2p, int64_t n_slots, hipStream_t stream, float* cos_out) {
  KpArgs a{};
  a.cos_out = cos_out;
  a.q = q_ctx; a.d = chunks; a.dm = dm; a.mu = mu; a.sigma = sigma; a.alpha = nullptr; a.w = nullptr;
  a.qm.len = q_len;  // effective query lengths [B] (may be null): pair rows of later tokens are not written
  a.n_pairs = P; a.ppq = 1; a.Q = Q; a.D = 40; a.E = E; a.K = 11;
  a.d_doc_rows = 50; a.d_row0 = 5; a.chunk_slot = chunk_slot; a.C = C; a.ps_out = ps_out;
  a.slot2p = slot2p; a.n_slots = n_slots;
  if (!env().kp_generic && kp_stream_supported(Q, E)) return launch_stream<11, true>(a, stream);
  if (P > 0x7fffffffLL) return set_error(MM_EUNSUPPORTED, "tkl: too many chunks for one launch");
  hipLaunchKernelGGL((kernel_pool_generic_kernel<11, true>), dim3((unsigned)P), dim3(64), 0, stream, a);
  return check_launch("kernel_pool_generic_kernel<TKL>");
}

template <int K>
static int launch_k(const KpArgs& a0, hipStream_t stream) {
  KpArgs a = a0;
  const bool force_generic = env().kp_generic != 0;
  const bool stream_ok = !force_generic && a.Q <= 32 && (a.E == 100 || a.E == 200 || a.E == 300);
  if (!force_generic && !stream_ok && kp128_supported(a.Q, a.D, a.E, a.dw != nullptr)) return kp128_launch(a, stream);
  if (a.dw) {  // gated (TK-Sparse); the gate vector of a document sits in LDS: D <= 4096 on the streaming path (LDS stays under 64 KB)
    if (stream_ok && a.D <= 4096) return launch_stream<K, false, true>(a, stream);
    if (a.n_pairs > 0x7fffffffLL) return set_error(MM_EUNSUPPORTED, "kernel_pool: too many pairs for one launch");
    hipLaunchKernelGGL((kernel_pool_generic_kernel<K, false, true>), dim3((unsigned)a.n_pairs), dim3(64), 0, stream, a);
    return check_launch("kernel_pool_generic_kernel<gated>");
  }
  if (stream_ok) return launch_stream<K, false>(a, stream);
  if (a.n_pairs > 0x7fffffffLL) return set_error(MM_EUNSUPPORTED, "kernel_pool: too many pairs for one launch");
  hipLaunchKernelGGL((kernel_pool_generic_kernel<K, false>), dim3((unsigned)a.n_pairs, (unsigned)(a.n_md > 0 ? a.n_mblk : 1)), dim3(64), 0,
                     stream, a);
  return check_launch("kernel_pool_generic_kernel");
}

}  // namespace mm

using namespace mm;

extern "C" size_t mm_kernel_pool_workspace_bytes(int64_t n_pairs, int64_t pairs_per_query, int Q, int D,
                                                  int q_mask_kind, int d_mask_kind) {
  if (pairs_per_query <= 0) pairs_per_query = 1;
  return packed_mask_bytes(q_mask_kind, (n_pairs + pairs_per_query - 1) / pairs_per_query, Q) +
         packed_mask_bytes(d_mask_kind, n_pairs, D);
}

extern "C" int mm_kernel_pool_ex_fwd(const void* q, const void* d, const void* q_mask, int q_mask_kind,
                                     const void* d_mask, int d_mask_kind, const float* d_gate,
                                     const int32_t* pair_query, int64_t n_queries, const float* mu,
                                     const float* sigma, const float* alpha, const float* w, float clamp_min, float* out,
                                     float* per_kernel, int64_t n_pairs, int64_t pairs_per_query, int Q, int D, int E,
                                     int K, int dtype, void* workspace, size_t workspace_bytes, void* stream_) {
  return mm_kernel_pool_ex_fwd2(q, d, q_mask, q_mask_kind, d_mask, d_mask_kind, d_gate, pair_query, n_queries, mu, sigma, alpha, w,
                                clamp_min, out, per_kernel, nullptr, n_pairs, pairs_per_query, Q, D, E, K, dtype, workspace,
                                workspace_bytes, stream_);
}

extern "C" int mm_kernel_pool_ex_fwd2(const void* q, const void* d, const void* q_mask, int q_mask_kind,
                                      const void* d_mask, int d_mask_kind, const float* d_gate,
                                      const int32_t* pair_query, int64_t n_queries, const float* mu,
                                      const float* sigma, const float* alpha, const float* w, float clamp_min, float* out,
                                      float* per_kernel, float* pooled, int64_t n_pairs, int64_t pairs_per_query, int Q, int D,
                                      int E, int K, int dtype, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!q || !d || !out || !mu || !sigma || !alpha || !w) return set_error(MM_EINVAL, "kernel_pool: null pointer");
  if (dtype != MM_F32)
    return set_error(MM_EUNSUPPORTED, "kernel_pool: float32 only (the reference cosine rejects bf16; tk.yaml use_fp16: False)");
  if (n_pairs < 0 || Q <= 0 || D <= 0 || E <= 0 || pairs_per_query <= 0) return set_error(MM_EINVAL, "kernel_pool: bad shape");
  if (!(clamp_min > 0.0f)) return set_error(MM_EINVAL, "kernel_pool: clamp_min must be > 0 (it sits inside a log)");
  if (pair_query && n_queries <= 0) return set_error(MM_EINVAL, "kernel_pool: pair_query needs n_queries");
  const int64_t q_rows = pair_query ? n_queries : (n_pairs + pairs_per_query - 1) / pairs_per_query;
  if (K <= 0 || K > kMaxK) return set_error(MM_EUNSUPPORTED, "kernel_pool: K=%d kernels (1..%d supported)", K, kMaxK);
  if (E % 4) return set_error(MM_EUNSUPPORTED, "kernel_pool: E=%d rows are not 16-byte multiples", E);
  if (((uintptr_t)q | (uintptr_t)d) & 15) return set_error(MM_EINVAL, "kernel_pool: q/d must be 16-byte aligned");
  if (n_pairs == 0) return MM_OK;
  KpArgs a{};
  a.q = (const float*)q; a.d = (const float*)d; a.mu = mu; a.sigma = sigma; a.alpha = alpha; a.w = w;
  a.out = out; a.per_kernel = per_kernel; a.pooled = pooled; a.n_pairs = n_pairs; a.ppq = pairs_per_query;
  a.Q = Q; a.D = D; a.E = E; a.K = K;
  a.d_doc_rows = D; a.d_row0 = 0;
  a.dw = d_gate; a.clamp_min = clamp_min; a.pair_q = pair_query;
  char* ws = (char*)workspace;
  size_t left = workspace ? workspace_bytes : 0;
  // eval.py-sized calls of TK at E = 100n with float masks: the two-wavefronts-per-pair kernel reads the masks itself
  // (one launch instead of two: the call is bound by the host, ~4 us per launch)
  const bool inline_masks = K == 11 && !d_gate && q_mask_kind == MM_MASK_F32 && d_mask_kind == MM_MASK_F32 && q_mask && d_mask &&
                            Q <= 32 && D > 32 && D <= 256 && (E == 100 || E == 200 || E == 300) && n_pairs * 2 <= (int64_t)kCUs * 4 &&
                            !env().kp_generic && !env().kp_f32mfma;
  if (inline_masks) {
    a.fqm = (const float*)q_mask;
    a.fdm = (const float*)d_mask;
  } else if (int e = resolve_mask_pair(q_mask, q_mask_kind, q_rows, Q, &a.qm, d_mask, d_mask_kind, n_pairs, D, &a.dm, &ws, &left, stream)) {
    return e;
  }
  if (K == 11) return launch_k<11>(a, stream);
  // any other kernel count (the lists in tk_kernels_mu / knrm_kernels are configuration): the generic kernel with
  // run-time K
  if (n_pairs > 0x7fffffffLL) return set_error(MM_EUNSUPPORTED, "kernel_pool: too many pairs for one launch");
  if (a.dw)
    hipLaunchKernelGGL((kernel_pool_generic_kernel<kMaxK, false, true>), dim3((unsigned)n_pairs), dim3(64), 0, stream, a);
  else
    hipLaunchKernelGGL((kernel_pool_generic_kernel<kMaxK, false, false>), dim3((unsigned)n_pairs), dim3(64), 0, stream, a);
  return check_launch("kernel_pool_generic_kernel<run-time K>");
}

// out[p] = sum over the partial rows, in block order (deterministic)
__global__ void __launch_bounds__(256) kp_sum_blocks_kernel(const float* __restrict__ partial, int64_t n, int nblk,
                                                            float* __restrict__ out) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  float s = 0.0f;
  for (int y = 0; y < nblk; ++y) s += partial[(int64_t)y * n + p];
  out[p] = s;
}

extern "C" size_t mm_kernel_pool_multi_workspace_bytes(int64_t n_pairs, int64_t pairs_per_query, int n_q, int n_d, int Q, int D,
                                                        int q_mask_kind, int d_mask_kind) {
  return mm_kernel_pool_workspace_bytes(n_pairs, pairs_per_query, Q, D, q_mask_kind, d_mask_kind) +
         (((size_t)n_q * n_d * (size_t)n_pairs * 4 + 255) & ~(size_t)255);
}

extern "C" int mm_kernel_pool_multi_fwd(const void* const* q_list, int n_q, const void* const* d_list, int n_d,
                                        const void* q_mask, int q_mask_kind, const void* d_mask, int d_mask_kind,
                                        const float* mu, const float* sigma, const float* alpha, const float* w,
                                        float clamp_min, float* out, int64_t n_pairs, int64_t pairs_per_query, int Q, int D,
                                        int E, int K, int dtype, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!q_list || !d_list || !out || !mu || !sigma || !alpha || !w) return set_error(MM_EINVAL, "kernel_pool_multi: null pointer");
  if (n_q < 1 || n_q > 4 || n_d < 1 || n_d > 4) return set_error(MM_EUNSUPPORTED, "kernel_pool_multi: 1..4 query and document tensors");
  if (dtype != MM_F32) return set_error(MM_EUNSUPPORTED, "kernel_pool_multi: float32 only");
  if (n_pairs < 0 || Q <= 0 || D <= 0 || E <= 0 || pairs_per_query <= 0) return set_error(MM_EINVAL, "kernel_pool_multi: bad shape");
  if (!(clamp_min > 0.0f)) return set_error(MM_EINVAL, "kernel_pool_multi: clamp_min must be > 0");
  if (K != 11) return set_error(MM_EUNSUPPORTED, "kernel_pool_multi: K=%d kernels (the 11-kernel instantiation only)", K);
  if (E % 4) return set_error(MM_EUNSUPPORTED, "kernel_pool_multi: E=%d rows are not 16-byte multiples", E);
  if (n_pairs == 0) return MM_OK;
  if (n_pairs > 0x7fffffffLL) return set_error(MM_EUNSUPPORTED, "kernel_pool_multi: too many pairs for one launch");
  KpArgs a{};
  for (int i = 0; i < n_q; ++i) {
    if (!q_list[i] || ((uintptr_t)q_list[i] & 15)) return set_error(MM_EINVAL, "kernel_pool_multi: query tensor %d null / not 16-byte aligned", i);
    a.mq[i] = (const float*)q_list[i];
  }
  for (int t = 0; t < n_d; ++t) {
    if (!d_list[t] || ((uintptr_t)d_list[t] & 15)) return set_error(MM_EINVAL, "kernel_pool_multi: document tensor %d null / not 16-byte aligned", t);
    a.md[t] = (const float*)d_list[t];
  }
  a.q = a.mq[0]; a.d = a.md[0];
  a.n_md = n_d; a.n_mblk = n_q * n_d;
  a.mu = mu; a.sigma = sigma; a.alpha = alpha; a.w = w;
  a.n_pairs = n_pairs; a.ppq = pairs_per_query; a.Q = Q; a.D = D; a.E = E; a.K = K;
  a.d_doc_rows = D; a.d_row0 = 0; a.clamp_min = clamp_min;
  char* ws = (char*)workspace;
  size_t left = workspace ? workspace_bytes : 0;
  const int64_t q_rows = (n_pairs + pairs_per_query - 1) / pairs_per_query;
  if (int e = resolve_mask_pair(q_mask, q_mask_kind, q_rows, Q, &a.qm, d_mask, d_mask_kind, n_pairs, D, &a.dm, &ws, &left, stream)) return e;
  const size_t need = (size_t)a.n_mblk * (size_t)n_pairs * 4;
  if (!ws || left < need) return set_error(MM_EWORKSPACE, "kernel_pool_multi: workspace needs %zu more bytes", need);
  float* partial = (float*)ws;
  a.out = partial;
  if (int e = launch_k<11>(a, stream)) return e;
  hipLaunchKernelGGL(kp_sum_blocks_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, stream, partial, n_pairs, a.n_mblk, out);
  return check_launch("kp_sum_blocks_kernel");
}

extern "C" int mm_kernel_pool_fwd(const void* q, const void* d, const void* q_mask, int q_mask_kind,
                                  const void* d_mask, int d_mask_kind, const float* mu, const float* sigma,
                                  const float* alpha, const float* w, float* out, float* per_kernel,
                                  int64_t n_pairs, int64_t pairs_per_query, int Q, int D, int E, int K, int dtype,
                                  void* workspace, size_t workspace_bytes, void* stream_) {
  return mm_kernel_pool_ex_fwd(q, d, q_mask, q_mask_kind, d_mask, d_mask_kind, nullptr, nullptr, 0, mu, sigma, alpha, w, 1e-10f, out,
                               per_kernel, n_pairs, pairs_per_query, Q, D, E, K, dtype, workspace, workspace_bytes, stream_);
}
