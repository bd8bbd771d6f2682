// placeholder until the TKL kernels land
#include "mm_internal.h"
extern "C" size_t mm_tkl_workspace_bytes(int64_t, int, int, int) { return 0; }
extern "C" int mm_tkl_fwd(const void*, const void*, const float*, const int32_t*, const float*, const float*, float*,
                          float*, int64_t, int64_t, int, int, int, int, int, void*, size_t, void*) {
  return mm::set_error(MM_EUNSUPPORTED, "mm_tkl_fwd: not built yet");
}
