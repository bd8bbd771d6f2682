// Error reporting + mask packing for libmm_native.so.
#include "mm_internal.h"
#include <stdlib.h>

namespace mm {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

static int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}

const EnvCfg& env() {
  static const EnvCfg cfg = [] {
    EnvCfg c;
    c.maxsim_nbuf = env_int("MM_MAXSIM_NBUF", c.maxsim_nbuf);
    if (c.maxsim_nbuf < 2) c.maxsim_nbuf = 2;
    if (c.maxsim_nbuf > 4) c.maxsim_nbuf = 4;
    c.maxsim_wpc = env_int("MM_MAXSIM_WPC", c.maxsim_wpc);
    c.maxsim_nt = env_int("MM_MAXSIM_NT", c.maxsim_nt);
    c.maxsim_generic = env_int("MM_MAXSIM_GENERIC", 0);
    c.maxsim_inb_untiled = env_int("MM_MAXSIM_INB_UNTILED", 0);
    c.maxsim_inb_nowg = env_int("MM_MAXSIM_INB_NOWG", 0);
    c.maxsim_no_inline_masks = env_int("MM_MAXSIM_NO_INLINE_MASKS", 0);
    c.maxsim_no_wpp2 = env_int("MM_MAXSIM_NO_WPP2", 0);
    c.maxsim_f32_terms = env_int("MM_MAXSIM_F32_TERMS", 3) == 2 ? 2 : 3;
    c.kp_generic = env_int("MM_KP_GENERIC", 0);
    c.kp_f32mfma = env_int("MM_KP_F32MFMA", 0);
    c.dot_prof = env_int("MM_DOT_PROF", 0);
    c.tkl_pairsums = env_int("MM_TKL_PAIRSUMS", 0);
    c.tkl_fold_regions = env_int("MM_TKL_FOLD_REGIONS", 0);
    c.tkl_bwd_nosplit = env_int("MM_TKL_BWD_NOSPLIT", 0);
    c.kp128_occ = env_int("MM_KP128_OCC", 0);
    c.kp_multi_2d = env_int("MM_KP_MULTI_2D", 0);
    c.kp_multi_wg = env_int("MM_KP_MULTI_WG", 0);
    c.kp_bwd_untiled = env_int("MM_KP_BWD_UNTILED", 0);
    c.kp_bwd_f32 = env_int("MM_KP_BWD_F32", 0);
    c.kp_bwd_nsplit = env_int("MM_KP_BWD_NSPLIT", 0);
    c.tkl_stage1_slices = env_int("MM_TKL_STAGE1_SLICES", 0);
    c.tkl_stage1_ksplit = env_int("MM_TKL_STAGE1_KSPLIT", 0);
    c.kp_multi_loop = env_int("MM_KP_MULTI_LOOP", -1);
    c.kp_bwd_threads = env_int("MM_KP_BWD_THREADS", 1024);
    return c;
  }();
  return cfg;
}

// One wave per mask row.  Lanes sweep the row 64 positions at a time; __ballot gives the 64
// validity bits -> two uint32 words.  len = index of the last real token + 1 (so all-padding tails
// are never loaded by the scoring kernels); holes inside [0, len) stay visible through `bits`.
template <typename T>
__global__ void __launch_bounds__(256) pack_mask_kernel(const T* __restrict__ mask, int64_t rows, int L,
                                                        int64_t row_stride, int col0, int words,
                                                        int32_t* __restrict__ len_out,
                                                        uint32_t* __restrict__ bits_out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* m = mask + row * row_stride + col0;
  int last = 0;
  for (int base = 0; base < L; base += 64) {
    const int j = base + lane;
    const bool v = (j < L) && (m[j] != T(0));
    const unsigned long long b = __ballot(v);
    if (lane == 0) {
      const int w = base >> 5;
      bits_out[row * words + w] = (uint32_t)b;
      if (w + 1 < words) bits_out[row * words + w + 1] = (uint32_t)(b >> 32);
    }
    if (b) last = base + 64 - __builtin_clzll(b);
  }
  if (lane == 0) len_out[row] = last;
}

// The same for the query AND the document masks of one call in ONE launch (eval.py-sized calls of 512 pairs are
// launch-bound on the host side: every launch saved is ~4 us of the ~20 us a call costs).
template <typename T>
__global__ void __launch_bounds__(256) pack_mask2_kernel(const T* __restrict__ m0, int64_t rows0, int L0, int words0,
                                                         int32_t* __restrict__ len0, uint32_t* __restrict__ bits0, int nblk0,
                                                         const T* __restrict__ m1, int64_t rows1, int L1, int words1,
                                                         int32_t* __restrict__ len1, uint32_t* __restrict__ bits1) {
  const int lane = threadIdx.x & 63;
  const bool second = (int)blockIdx.x >= nblk0;
  const int64_t row = (int64_t)(second ? blockIdx.x - nblk0 : blockIdx.x) * 4 + (threadIdx.x >> 6);
  const int64_t rows = second ? rows1 : rows0;
  if (row >= rows) return;
  const int L = second ? L1 : L0, words = second ? words1 : words0;
  const T* m = (second ? m1 : m0) + row * L;
  int32_t* len_out = second ? len1 : len0;
  uint32_t* bits_out = second ? bits1 : bits0;
  int last = 0;
  for (int base = 0; base < L; base += 64) {
    const int j = base + lane;
    const bool v = (j < L) && (m[j] != T(0));
    const unsigned long long b = __ballot(v);
    if (lane == 0) {
      const int w = base >> 5;
      bits_out[row * words + w] = (uint32_t)b;
      if (w + 1 < words) bits_out[row * words + w + 1] = (uint32_t)(b >> 32);
    }
    if (b) last = base + 64 - __builtin_clzll(b);
  }
  if (lane == 0) len_out[row] = last;
}

size_t packed_mask_bytes(int kind, int64_t rows, int L) {
  if (kind == MM_MASK_U8 || kind == MM_MASK_I64 || kind == MM_MASK_F32) {
    const int words = (L + 31) / 32;
    size_t b = (size_t)rows * 4 + (size_t)rows * words * 4;
    return (b + 255) & ~(size_t)255;
  }
  return 0;
}

int resolve_mask(const void* mask, int kind, int64_t rows, int L, char** ws, size_t* ws_left,
                 hipStream_t stream, PackedMask* out, int64_t row_stride, int col0) {
  if (row_stride <= 0) row_stride = L;
  out->len = nullptr;
  out->bits = nullptr;
  switch (kind) {
    case MM_MASK_NONE:
      return MM_OK;
    case MM_MASK_LEN_I32:
      if (!mask) return set_error(MM_EINVAL, "mask kind LEN_I32 given with a null pointer");
      out->len = (const int32_t*)mask;
      return MM_OK;
    case MM_MASK_U8:
    case MM_MASK_I64:
    case MM_MASK_F32: {
      if (!mask) return set_error(MM_EINVAL, "dense mask given with a null pointer");
      const size_t need = packed_mask_bytes(kind, rows, L);
      if (!*ws || *ws_left < need)
        return set_error(MM_EWORKSPACE, "workspace too small for mask packing: need %zu more bytes, have %zu",
                         need, *ws_left);
      const int words = (L + 31) / 32;
      int32_t* len = (int32_t*)*ws;
      uint32_t* bits = (uint32_t*)(*ws + (size_t)rows * 4);
      *ws += need;
      *ws_left -= need;
      const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
      if (kind == MM_MASK_U8)
        hipLaunchKernelGGL(pack_mask_kernel<uint8_t>, grid, block, 0, stream, (const uint8_t*)mask, rows, L, row_stride, col0, words, len, bits);
      else if (kind == MM_MASK_I64)
        hipLaunchKernelGGL(pack_mask_kernel<int64_t>, grid, block, 0, stream, (const int64_t*)mask, rows, L, row_stride, col0, words, len, bits);
      else
        hipLaunchKernelGGL(pack_mask_kernel<float>, grid, block, 0, stream, (const float*)mask, rows, L, row_stride, col0, words, len, bits);
      out->len = len;
      out->bits = bits;
      return check_launch("pack_mask_kernel");
    }
    default:
      return set_error(MM_EINVAL, "unknown mask kind %d", kind);
  }
}

int resolve_mask_pair(const void* m0, int kind0, int64_t rows0, int L0, PackedMask* out0, const void* m1, int kind1,
                      int64_t rows1, int L1, PackedMask* out1, char** ws, size_t* ws_left, hipStream_t stream) {
  const bool dense0 = kind0 == MM_MASK_U8 || kind0 == MM_MASK_I64 || kind0 == MM_MASK_F32;
  if (!(dense0 && kind1 == kind0 && m0 && m1)) {  // not two dense masks of one element type: one at a time
    if (int e = resolve_mask(m0, kind0, rows0, L0, ws, ws_left, stream, out0)) return e;
    return resolve_mask(m1, kind1, rows1, L1, ws, ws_left, stream, out1);
  }
  const size_t need0 = packed_mask_bytes(kind0, rows0, L0), need1 = packed_mask_bytes(kind1, rows1, L1);
  if (!*ws || *ws_left < need0 + need1)
    return set_error(MM_EWORKSPACE, "workspace too small for mask packing: need %zu more bytes, have %zu", need0 + need1, *ws_left);
  const int words0 = (L0 + 31) / 32, words1 = (L1 + 31) / 32;
  int32_t* len0 = (int32_t*)*ws;
  uint32_t* bits0 = (uint32_t*)(*ws + (size_t)rows0 * 4);
  int32_t* len1 = (int32_t*)(*ws + need0);
  uint32_t* bits1 = (uint32_t*)(*ws + need0 + (size_t)rows1 * 4);
  *ws += need0 + need1;
  *ws_left -= need0 + need1;
  const int nblk0 = (int)((rows0 + 3) / 4), nblk1 = (int)((rows1 + 3) / 4);
  const dim3 grid((unsigned)(nblk0 + nblk1)), block(256);
  if (kind0 == MM_MASK_U8)
    hipLaunchKernelGGL(pack_mask2_kernel<uint8_t>, grid, block, 0, stream, (const uint8_t*)m0, rows0, L0, words0, len0, bits0, nblk0,
                       (const uint8_t*)m1, rows1, L1, words1, len1, bits1);
  else if (kind0 == MM_MASK_I64)
    hipLaunchKernelGGL(pack_mask2_kernel<int64_t>, grid, block, 0, stream, (const int64_t*)m0, rows0, L0, words0, len0, bits0, nblk0,
                       (const int64_t*)m1, rows1, L1, words1, len1, bits1);
  else
    hipLaunchKernelGGL(pack_mask2_kernel<float>, grid, block, 0, stream, (const float*)m0, rows0, L0, words0, len0, bits0, nblk0,
                       (const float*)m1, rows1, L1, words1, len1, bits1);
  out0->len = len0; out0->bits = bits0;
  out1->len = len1; out1->bits = bits1;
  return check_launch("pack_mask2_kernel");
}

}  // namespace mm

extern "C" int mm_abi_version(void) { return MM_ABI_VERSION; }
extern "C" const char* mm_last_error(void) { return mm::g_err; }
