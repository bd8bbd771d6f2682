// placeholder until the TK kernel lands (next commit)
#include "mm_internal.h"
extern "C" int mm_kernel_pool_fwd(const void*, const void*, const void*, int, const void*, int, const float*,
                                  const float*, const float*, const float*, float*, float*, int64_t, int64_t, int,
                                  int, int, int, int, void*) {
  return mm::set_error(MM_EUNSUPPORTED, "mm_kernel_pool_fwd: not built yet");
}
