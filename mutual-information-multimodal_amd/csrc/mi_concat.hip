// placeholder: replaced by the fused concat-MLP kernels
#include "mi_common.h"
using namespace mi;
extern "C" {
size_t mi_concat_mlp_workspace_bytes(int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int, int) { return 256; }
int mi_concat_mlp_fwd(const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                      const float*, const int64_t*, const int64_t*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                      int64_t, int, int, int, float*, mi_stats*, float*, float*, void*, size_t, void*) {
  set_error("mi_concat_mlp_fwd: not built yet");
  return MI_ESHAPE;
}
int mi_concat_mlp_bwd(const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                      const float*, const int64_t*, const int64_t*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                      int64_t, int, const mi_stats*, const float*, const float*, float*, float*, float*, float*, float*,
                      float*, float*, float*, void*, size_t, void*) {
  set_error("mi_concat_mlp_bwd: not built yet");
  return MI_ESHAPE;
}
}
