// Fused concat-MLP critic (the reference's mi_discriminator): host orchestration + C ABI.
//   reference call site: mutual_info_img_txt/main_utils.py:220-226
//   forward : U = X W1x^T, V = Y W1y^T + b1 (fp32 MFMA GEMMs) -> concat_fwd_kernel (scores + sign bits)
//             -> masked log-sum-exp over S (HBM-bound) -> stats, loss
//   backward: see mi_concat_bwd.h
#include "mi_concat_fwd.h"
#include "mi_gemm.h"

namespace mi {

// kernels defined in mi_bound.hip
__global__ void matrix_partials_kernel(const float*, const int64_t*, const int64_t*, int64_t, int64_t, int64_t, Partial*);

constexpr int kMatrixPartialBlocks = 2048;

struct ConcatPlan {
  float* u;
  float* v;
  bf16_t* w2bf;
  Partial* partials;
  unsigned long long* bitsP;
  unsigned* bitsN;
  size_t bytes;
};

static ConcatPlan plan_concat(Workspace& ws, int64_t br, int64_t b, int64_t h1, int64_t h2, int precision,
                              int need_grad) {
  ConcatPlan p{};
  p.u = ws.take<float>(br * h1);
  p.v = ws.take<float>(b * h1);
  p.w2bf = (precision == MI_PREC_BF16) ? ws.take<bf16_t>(h2 * h1) : nullptr;
  p.partials = ws.take<Partial>(kMatrixPartialBlocks);
  if (need_grad) {
    p.bitsP = ws.take<unsigned long long>(br * b * (h2 / 64));
    p.bitsN = ws.take<unsigned>(br * ((b + 31) / 32) * h2);
  }
  p.bytes = ws.off;
  return p;
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) dst[e] = (bf16_t)src[e];
}

static int check_concat_shape(const char* fn, int64_t br, int64_t b, int64_t row_offset, int64_t dx, int64_t dy,
                              int64_t h1, int64_t h2, int precision) {
  MI_CHECK_ARG(br >= 1 && b >= 1 && br <= b, "%s: need 1 <= b_rows <= b", fn);
  MI_CHECK_ARG(row_offset >= 0 && row_offset + br <= b, "%s: row block outside [0, b)", fn);
  MI_CHECK_ARG(dx >= 1 && dy >= 1, "%s: embedding widths must be >= 1", fn);
  MI_CHECK_ARG(precision == MI_PREC_F32 || precision == MI_PREC_BF16, "%s: unknown precision %d", fn, precision);
  if (h1 < 64 || h1 % 64 != 0 || h2 < 256 || h2 % 256 != 0) {
    set_error("%s: the fused kernels need h1 %% 64 == 0 and h2 %% 256 == 0 (got h1=%lld, h2=%lld)", fn, (long long)h1,
              (long long)h2);
    return MI_ESHAPE;
  }
  return MI_OK;
}

template <typename OpT>
static int launch_concat_fwd(const float* u, const float* v, const OpT* w2, const float* b2, const float* w3,
                             const float* b3, int64_t br, int64_t b, int h1, int h2, float* scores,
                             unsigned long long* bitsP, unsigned* bitsN, hipStream_t st) {
  const size_t smem = sizeof(FwdSmem<OpT>);
  hipError_t e = hipFuncSetAttribute((const void*)concat_fwd_kernel<OpT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)smem);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(concat_fwd_kernel)");
  dim3 grid((unsigned)((b + kFwdTJ - 1) / kFwdTJ), (unsigned)((br + kFwdTI - 1) / kFwdTI));
  hipLaunchKernelGGL(concat_fwd_kernel<OpT>, grid, dim3(512), smem, st, u, v, w2, b2, w3, b3, br, b, h1, h2, scores,
                     bitsP, bitsN);
  MI_LAUNCH_CHECK("concat_fwd_kernel");
  return MI_OK;
}

}  // namespace mi

using namespace mi;

extern "C" {

size_t mi_concat_mlp_workspace_bytes(int64_t b_rows, int64_t b, int64_t d_img, int64_t d_txt, int64_t h1, int64_t h2,
                                     int precision, int need_grad) {
  (void)d_img;
  (void)d_txt;
  Workspace ws(nullptr, 0);
  return plan_concat(ws, b_rows, b, h1, h2, precision, need_grad).bytes + 256;
}

int mi_concat_mlp_fwd(const float* x, const float* y, const float* w1, const float* b1, const float* w2, const float* b2,
                      const float* w3, const float* b3, const int64_t* sid_rows, const int64_t* sid_cols, int64_t b_rows,
                      int64_t b, int64_t row_offset, int64_t d_img, int64_t d_txt, int64_t h1, int64_t h2, int estimator,
                      int precision, int need_grad, float* loss_out, mi_stats* stats, float* partials_out,
                      float* scores_out, void* workspace, size_t workspace_bytes, void* stream) {
  MI_CHECK_ARG(x && y && w1 && b1 && w2 && b2 && w3 && b3 && sid_rows && sid_cols && stats && scores_out && workspace,
               "mi_concat_mlp_fwd: null pointer");
  int rc = check_concat_shape("mi_concat_mlp_fwd", b_rows, b, row_offset, d_img, d_txt, h1, h2, precision);
  if (rc) return rc;
  MI_CHECK_ARG(estimator == MI_DV || estimator == MI_INFONCE, "mi_concat_mlp_fwd: unknown estimator %d", estimator);
  Workspace ws(workspace, workspace_bytes);
  ConcatPlan p = plan_concat(ws, b_rows, b, h1, h2, precision, need_grad);
  if (!ws.ok()) {
    set_error("mi_concat_mlp_fwd: workspace too small (%zu < %zu)", workspace_bytes, ws.off);
    return MI_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int64_t d = d_img + d_txt;
  // U = X W1x^T, V = Y W1y^T + b1: exact fp32 products (these are < 1 % of the flops)
  rc = launch_gemm<float>(make_operand(x, d_img, 1), make_operand(w1, d, 1), b_rows, h1, d_img,
                          EpiStore{p.u, h1, nullptr, 1.0f, 0}, st, "concat U = X W1x^T");
  if (rc) return rc;
  rc = launch_gemm<float>(make_operand(y, d_txt, 1), make_operand(w1 + d_img, d, 1), b, h1, d_txt,
                          EpiStore{p.v, h1, b1, 1.0f, 0}, st, "concat V = Y W1y^T + b1");
  if (rc) return rc;
  unsigned long long* bitsP = need_grad ? p.bitsP : nullptr;
  unsigned* bitsN = need_grad ? p.bitsN : nullptr;
  if (precision == MI_PREC_BF16) {
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(512), dim3(256), 0, st, w2, p.w2bf, h1 * h2);
    MI_LAUNCH_CHECK("f32_to_bf16_kernel");
    rc = launch_concat_fwd<bf16_t>(p.u, p.v, p.w2bf, b2, w3, b3, b_rows, b, (int)h1, (int)h2, scores_out, bitsP, bitsN,
                                   st);
  } else {
    rc = launch_concat_fwd<float>(p.u, p.v, w2, b2, w3, b3, b_rows, b, (int)h1, (int)h2, scores_out, bitsP, bitsN, st);
  }
  if (rc) return rc;
  const int grid = (int)(b_rows < kMatrixPartialBlocks ? b_rows : kMatrixPartialBlocks);
  hipLaunchKernelGGL(matrix_partials_kernel, dim3(grid), dim3(256), 0, st, (const float*)scores_out, sid_rows, sid_cols,
                     b_rows, b, row_offset, p.partials);
  MI_LAUNCH_CHECK("matrix_partials_kernel");
  return launch_finalize(p.partials, grid, b, estimator, loss_out, stats, partials_out, st);
}

int mi_concat_mlp_bwd(const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                      const float*, const int64_t*, const int64_t*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                      int64_t, int, const mi_stats*, const float*, const float*, float*, float*, float*, float*, float*,
                      float*, float*, float*, void*, size_t, void*) {
  set_error("mi_concat_mlp_bwd: not built yet");
  return MI_ESHAPE;
}

}  // extern "C"
