// Fused concat-MLP critic (the reference's mi_discriminator): host orchestration + C ABI.
//   reference call site: mutual_info_img_txt/main_utils.py:220-226
//   forward : U = X W1x^T, V = Y W1y^T + b1 (fp32 MFMA GEMMs) -> concat_fwd_kernel (scores + sign-bit images)
//             -> masked log-sum-exp over S (HBM-bound) -> stats, loss
//   backward: prep (w3*W2 permuted copy) -> duv kernel (dU, dV slabs) -> dw2 kernel (D slabs) -> db2 kernel ->
//             slab reductions / finish -> dX = dU W1x, dY = dV W1y, dW1 = [dU^T X | dV^T Y], db1 = colsum(dV)
#include "mi_concat_bwd.h"
#include "mi_concat_fwd.h"
#include "mi_concat_fwd_dma.h"
#include "mi_concat_f16.h"
#include "mi_gemm.h"
#include "mi_gemm_bf16.h"

#include <stdlib.h>

#include <type_traits>

namespace mi {

// kernel defined in mi_bound.hip
__global__ void matrix_partials_kernel(const float*, const int64_t*, const int64_t*, int64_t, int64_t, int64_t, Partial*);

constexpr int kMatrixPartialBlocks = 2048;

struct ConcatPlan {
  // forward
  float* u;
  float* v;
  bf16_t* w2bf;
  // fp16 mode (mi_concat_f16.h)
  f16_t *uh, *vh, *w2h, *vht;
  unsigned* upk;
  F16Scales* f16sc;
  // two-part fp16 mode (MI_PREC_F16X3): scaled fp32 copies of U, V and the interleaved [hi | lo] copy of W2
  float *us, *vs;
  f16_t* w2x;
  Partial* partials;
  unsigned long long* bitsP;
  unsigned* bitsN;
  // backward
  void* w2wp;
  float* du_slab;
  float* dv_slab;
  float* du;
  float* dv;
  float* d_slab;
  float* m_slab;
  float* g_sum;
  float* w1_slab;
  float* col_part;
  int n_w1split;
  int64_t w1_kchunk_x, w1_kchunk_y;
  int n_jsplit, cols_per_split, n_iblk;
  int n_dsplit, rows_per_dsplit;
  int n_msplit, rows_per_msplit;
  size_t bytes;
};

static ConcatPlan plan_concat(Workspace& ws, int64_t br, int64_t b, int64_t h1, int64_t h2, int precision,
                              int need_grad, int64_t dmax) {
  ConcatPlan p{};
  p.u = ws.take<float>(br * h1);
  p.v = ws.take<float>(b * h1);
  p.w2bf = (precision == MI_PREC_BF16) ? ws.take<bf16_t>(h2 * h1) : nullptr;
  if (precision == MI_PREC_F16) {
    p.uh = ws.take<f16_t>(br * h1);
    p.vh = ws.take<f16_t>(b * h1);
    p.w2h = ws.take<f16_t>(h2 * h1);
    p.f16sc = ws.take<F16Scales>(1);
    p.upk = need_grad ? ws.take<unsigned>(br * h1) : nullptr;
    p.vht = need_grad ? ws.take<f16_t>(b * h1) : nullptr;
  }
  if (precision == MI_PREC_F16X3) {
    p.us = ws.take<float>(br * h1);
    p.vs = ws.take<float>(b * h1);
    p.w2x = ws.take<f16_t>(2 * h2 * h1);
    p.f16sc = ws.take<F16Scales>(1);
  }
  p.partials = ws.take<Partial>(kMatrixPartialBlocks);
  if (need_grad) {
    p.bitsP = ws.take<unsigned long long>(bitsp_words(br, b, h2));
    p.bitsN = ws.take<unsigned>(br * ((b + 31) / 32) * h2);
    const bool op16 = precision == MI_PREC_BF16 || precision == MI_PREC_F16 || precision == MI_PREC_F16X3;
    const int kc = op16 ? DuvCfg<bf16_t>::KC : DuvCfg<float>::KC;
    const int64_t n_kc = (h1 + kc - 1) / kc;
    p.n_iblk = (int)((br + kDuvTI - 1) / kDuvTI);
    int64_t js = (512 + n_kc * p.n_iblk - 1) / (n_kc * p.n_iblk);
    if (js < 1) js = 1;
    if (js > 8) js = 8;
    int64_t cps = (b + js - 1) / js;
    cps = (cps + kDuvTJ - 1) / kDuvTJ * kDuvTJ;
    p.cols_per_split = (int)cps;
    p.n_jsplit = (int)((b + cps - 1) / cps);
    int64_t rps = (br + 31) / 32;
    if (rps < 16) rps = 16;
    p.rows_per_dsplit = (int)rps;
    p.n_dsplit = (int)((br + rps - 1) / rps);
    // row slices of the db2 launch: 1,024; the fp16 kernel measured 0.543 / 0.503 / 0.478 ms with 1,024 / 2,048 / 4,096
    // (the finish kernel then 0.058 / 0.061 / 0.068).  MI_DB2_SPLITS: A/B knob.
    static const int64_t db2_env = getenv("MI_DB2_SPLITS") ? atoll(getenv("MI_DB2_SPLITS")) : 0;
    const int64_t db2_splits = db2_env > 0 ? db2_env : (precision == MI_PREC_F16 ? 2048 : 1024);
    int64_t rpm = (br + db2_splits - 1) / db2_splits;
    p.rows_per_msplit = (int)rpm;
    p.n_msplit = (int)((br + rpm - 1) / rpm);
    if (op16) p.w2wp = ws.take<bf16_t>((precision == MI_PREC_F16X3 ? 2 : 1) * h1 * h2);  // F16X3: hi and lo parts
    else p.w2wp = ws.take<float>(h1 * h2);
    p.du_slab = ws.take<float>((int64_t)p.n_jsplit * br * h1);
    p.dv_slab = ws.take<float>((int64_t)p.n_iblk * b * h1);
    p.du = ws.take<float>(br * h1);
    p.dv = ws.take<float>(b * h1);
    p.d_slab = ws.take<float>((int64_t)p.n_dsplit * h2 * h1);
    p.m_slab = ws.take<float>((int64_t)p.n_msplit * h2);
    p.g_sum = ws.take<float>(p.n_msplit);
    // dW1 = [dU^T X | dV^T Y]: K = batch; few output tiles -> split-K slabs (16 splits of >= 64 samples)
    int64_t sp = (b + 63) / 64;
    if (sp > 16) sp = 16;
    if (sp < 1) sp = 1;
    p.n_w1split = (int)sp;
    p.w1_kchunk_x = ((br + sp - 1) / sp + 15) / 16 * 16;
    p.w1_kchunk_y = ((b + sp - 1) / sp + 15) / 16 * 16;
    p.w1_slab = ws.take<float>(sp * h1 * (dmax > 0 ? dmax : 1));
    p.col_part = ws.take<float>(64 * h1);
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
  MI_CHECK_ARG(precision == MI_PREC_F32 || precision == MI_PREC_BF16 || precision == MI_PREC_F16 || precision == MI_PREC_F16X3,
               "%s: precision %d is not one of MI_PREC_F32, MI_PREC_BF16, MI_PREC_F16, MI_PREC_F16X3", fn, precision);
  if (h1 < 64 || h1 % 64 != 0 || h2 < 256 || h2 % 256 != 0 || h2 > 512) {
    set_error("%s: the fused kernels need h1 %% 64 == 0 and h2 in {256, 512} (got h1=%lld, h2=%lld)", fn, (long long)h1,
              (long long)h2);
    return MI_ESHAPE;
  }
  return MI_OK;
}

template <typename OpT>
static int launch_concat_fwd(const float* u, const float* v, const OpT* w2, const float* b2, const float* w3,
                             const float* b3, int64_t br, int64_t b, int h1, int h2, float* scores,
                             unsigned long long* bitsP, unsigned* bitsN, hipStream_t st) {
  // NWN = 1: 256-thread workgroups, two per CU (independent barriers for the two waves of a SIMD); NWN = 2: one
  // 512-thread workgroup per CU.  MI_CONCAT_FWD_NWN=2 selects the latter for A/B measurements.
  static const int nwn = (getenv("MI_CONCAT_FWD_NWN") && atoi(getenv("MI_CONCAT_FWD_NWN")) == 2) ? 2 : 1;
  dim3 grid((unsigned)((b + kFwdTJ - 1) / kFwdTJ), (unsigned)((br + kFwdTI - 1) / kFwdTI));
  // the LDS-DMA kernel decodes an XCD-aware tile order from a 1-D grid
  const dim3 grid1d((unsigned)(8 * (((b + kFwdTJ - 1) / kFwdTJ + 7) / 8) * ((br + kFwdTI - 1) / kFwdTI)));
  if constexpr (sizeof(OpT) == 2) {
    // bf16: LDS-DMA staged variant (mi_concat_fwd_dma.h); MI_CONCAT_FWD_REGSTAGE=1 selects the register-staged kernel
    static const bool regstage = getenv("MI_CONCAT_FWD_REGSTAGE") != nullptr;
    if (!regstage) {
      if (nwn == 2) {
        MI_SET_DYN_SMEM((concat_fwd_dma_kernel<2>), FwdDmaSmem<2>::TOTAL, "hipFuncSetAttribute(concat_fwd_dma_kernel)");
        ProfScope prof_("concat_fwd_kernel", st);
        hipLaunchKernelGGL((concat_fwd_dma_kernel<2>), grid1d, dim3(512), FwdDmaSmem<2>::TOTAL, st, u, v, w2, b2, w3, b3,
                           br, b, h1, h2, scores, bitsP, bitsN, xcd_natural());
      } else {
        MI_SET_DYN_SMEM((concat_fwd_dma_kernel<1>), FwdDmaSmem<1>::TOTAL, "hipFuncSetAttribute(concat_fwd_dma_kernel)");
        ProfScope prof_("concat_fwd_kernel", st);
        hipLaunchKernelGGL((concat_fwd_dma_kernel<1>), grid1d, dim3(256), FwdDmaSmem<1>::TOTAL, st, u, v, w2, b2, w3, b3,
                           br, b, h1, h2, scores, bitsP, bitsN, xcd_natural());
      }
      MI_LAUNCH_CHECK("concat_fwd_dma_kernel");
      return MI_OK;
    }
  }
  if (nwn == 2) {
    const size_t smem = sizeof(FwdSmem<OpT, 2>);
    MI_SET_DYN_SMEM((concat_fwd_kernel<OpT, 2>), smem, "hipFuncSetAttribute(concat_fwd_kernel)");
    ProfScope prof_("concat_fwd_kernel", st);
    hipLaunchKernelGGL((concat_fwd_kernel<OpT, 2>), grid, dim3(512), smem, st, u, v, w2, b2, w3, b3, br, b, h1, h2,
                       scores, bitsP, bitsN);
  } else {
    const size_t smem = sizeof(FwdSmem<OpT, 1>);
    MI_SET_DYN_SMEM((concat_fwd_kernel<OpT, 1>), smem, "hipFuncSetAttribute(concat_fwd_kernel)");
    ProfScope prof_("concat_fwd_kernel", st);
    hipLaunchKernelGGL((concat_fwd_kernel<OpT, 1>), grid, dim3(256), smem, st, u, v, w2, b2, w3, b3, br, b, h1, h2,
                       scores, bitsP, bitsN);
  }
  MI_LAUNCH_CHECK("concat_fwd_kernel");
  return MI_OK;
}

// OpT = float (exact fp32 products), bf16_t, or f16_t (mi_concat_f16.h: scaled fp16 operands, packed generation)
template <typename OpT, bool X3 = false>
static int concat_bwd_impl(const float* x, const float* y, const float* w1, const float* w2, const float* b2,
                           const float* w3, const int64_t* sid_rows, const int64_t* sid_cols, int64_t br, int64_t b,
                           int64_t row_offset, int64_t dx, int64_t dy, int h1, int h2, const mi_stats* stats,
                           const float* grad_out, const float* scores, float* grad_x, float* grad_y, float* grad_w1,
                           float* grad_b1, float* grad_w2, float* grad_b2, float* grad_w3, float* grad_b3,
                           const ConcatPlan& p, hipStream_t st) {
  using DC = DuvCfg<OpT>;
  constexpr bool kF16 = std::is_same<OpT, f16_t>::value;
  OpT* w2wp = (OpT*)p.w2wp;
  {
    ProfScope prof_("prep_w2w_kernel", st);
    if constexpr (X3) {
      hipLaunchKernelGGL(f16x3_prep_w2w_kernel, dim3(512), dim3(256), 0, st, w2, w3, h1, h2, (const F16Scales*)p.f16sc, w2wp);
    } else if constexpr (kF16) {
      hipLaunchKernelGGL(f16_prep_w2w_kernel, dim3(512), dim3(256), 0, st, w2, w3, h1, h2, (const F16Scales*)p.f16sc, w2wp);
    } else {
      hipLaunchKernelGGL(prep_w2w_kernel<OpT>, dim3(512), dim3(256), 0, st, w2, w3, h1, h2, w2wp);
    }
  }
  MI_LAUNCH_CHECK("prep_w2w_kernel");

  // ---- dU / dV --------------------------------------------------------------------------------------------------
  {
    const int ldw = h2 + DC::PADW;
    size_t smem = (((size_t)DC::KC * ldw * sizeof(OpT)) + 15) & ~(size_t)15;
    smem += 256 * sizeof(bf16x8) + kDuvTI * kDuvTJ * sizeof(float) + kDuvTJ * DC::KC * sizeof(float) +
            4 * kDuvTJ * DC::KC * sizeof(float);
    dim3 grid(xcd_grid((h1 + DC::KC - 1) / DC::KC, (int64_t)p.n_iblk * p.n_jsplit));
    // 16-bit modes: the round-4 kernel (mi_concat_f16.h, concat_bwd_duv3_kernel: 256-thread workgroups, two per CU, 64 k
    // each); MI_DUV_OLD=1 keeps the first kernel (A/B)
    static const bool duv_old = getenv("MI_DUV_OLD") != nullptr;
    if constexpr (sizeof(OpT) == 2) {
      if ((!duv_old || X3) && (h2 % 128) == 0) {
        const size_t smem3 = Duv3Smem::total(h2, X3);
        const dim3 grid3(xcd_grid((h1 + Duv3Smem::KC - 1) / Duv3Smem::KC, (int64_t)p.n_iblk * p.n_jsplit));
        const F16Scales* scp = kF16 ? (const F16Scales*)p.f16sc : (const F16Scales*)nullptr;
        if constexpr (X3) {
          MI_SET_DYN_SMEM((concat_bwd_duv3_kernel<f16_t, float, true>), smem3, "hipFuncSetAttribute(concat_bwd_duv3_kernel)");
          ProfScope prof_("concat_bwd_duv_kernel", st);
          hipLaunchKernelGGL((concat_bwd_duv3_kernel<f16_t, float, true>), grid3, dim3(256), smem3, st, (const float*)p.u,
                             (const float*)p.v, (const f16_t*)w2wp, (const unsigned long long*)p.bitsP, scores, sid_rows,
                             sid_cols, stats, grad_out, br, b, row_offset, h1, h2, p.cols_per_split, xcd_natural(),
                             p.du_slab, p.dv_slab, scp);
        } else if constexpr (kF16) {
          MI_SET_DYN_SMEM((concat_bwd_duv3_kernel<f16_t, f16_t>), smem3, "hipFuncSetAttribute(concat_bwd_duv3_kernel)");
          ProfScope prof_("concat_bwd_duv_kernel", st);
          hipLaunchKernelGGL((concat_bwd_duv3_kernel<f16_t, f16_t>), grid3, dim3(256), smem3, st, (const f16_t*)p.uh,
                             (const f16_t*)p.vh, (const f16_t*)w2wp, (const unsigned long long*)p.bitsP, scores, sid_rows,
                             sid_cols, stats, grad_out, br, b, row_offset, h1, h2, p.cols_per_split, xcd_natural(),
                             p.du_slab, p.dv_slab, scp);
        } else {
          MI_SET_DYN_SMEM((concat_bwd_duv3_kernel<OpT, float>), smem3, "hipFuncSetAttribute(concat_bwd_duv3_kernel)");
          ProfScope prof_("concat_bwd_duv_kernel", st);
          hipLaunchKernelGGL((concat_bwd_duv3_kernel<OpT, float>), grid3, dim3(256), smem3, st, (const float*)p.u,
                             (const float*)p.v, (const OpT*)w2wp, (const unsigned long long*)p.bitsP, scores, sid_rows,
                             sid_cols, stats, grad_out, br, b, row_offset, h1, h2, p.cols_per_split, xcd_natural(),
                             p.du_slab, p.dv_slab, scp);
        }
        MI_LAUNCH_CHECK("concat_bwd_duv3_kernel");
        goto duv_done;
      }
    }
    if constexpr (kF16) {
      MI_SET_DYN_SMEM((concat_bwd_duv_kernel<f16_t, f16_t>), smem, "hipFuncSetAttribute(concat_bwd_duv_kernel)");
      ProfScope prof_("concat_bwd_duv_kernel", st);
      hipLaunchKernelGGL((concat_bwd_duv_kernel<f16_t, f16_t>), grid, dim3(512), smem, st, (const f16_t*)p.uh,
                         (const f16_t*)p.vh, (const f16_t*)w2wp, (const unsigned long long*)p.bitsP, scores, sid_rows,
                         sid_cols, stats, grad_out, br, b, row_offset, h1, h2, p.cols_per_split, xcd_natural(), p.du_slab,
                         p.dv_slab, (const F16Scales*)p.f16sc);
    } else {
      MI_SET_DYN_SMEM((concat_bwd_duv_kernel<OpT>), smem, "hipFuncSetAttribute(concat_bwd_duv_kernel)");
      ProfScope prof_("concat_bwd_duv_kernel", st);
      hipLaunchKernelGGL(concat_bwd_duv_kernel<OpT>, grid, dim3(512), smem, st, (const float*)p.u, (const float*)p.v,
                         (const OpT*)w2wp, (const unsigned long long*)p.bitsP, scores, sid_rows, sid_cols, stats,
                         grad_out, br, b, row_offset, h1, h2, p.cols_per_split, xcd_natural(), p.du_slab, p.dv_slab,
                         (const F16Scales*)nullptr);
    }
    MI_LAUNCH_CHECK("concat_bwd_duv_kernel");
  duv_done:
    {
      ProfScope prof_("slab_reduce_kernel", st);
      hipLaunchKernelGGL(slab_reduce_kernel, dim3(1024), dim3(256), 0, st, (const float*)p.du_slab, p.n_jsplit, br,
                         (int64_t)h1, p.du);
      hipLaunchKernelGGL(slab_reduce_kernel, dim3(1024), dim3(256), 0, st, (const float*)p.dv_slab, p.n_iblk, b,
                         (int64_t)h1, p.dv);
    }
    MI_LAUNCH_CHECK("slab_reduce_kernel");
  }

  // ---- D = sum_p M g H1  ->  dW2, dW3, db2, db3 --------------------------------------------------------------------
  {
    dim3 grid(xcd_grid(((h1 + 255) / 256) * (h2 / 256), p.n_dsplit));
    if constexpr (X3) {
      const size_t smem = (256 * 36 + kDw2IB * 32) * sizeof(float) + 256 * sizeof(bf16x8);
      MI_SET_DYN_SMEM((concat_bwd_dw2_kernel<f16_t, true>), smem, "hipFuncSetAttribute(concat_bwd_dw2_kernel)");
      ProfScope prof_("concat_bwd_dw2_kernel", st);
      hipLaunchKernelGGL((concat_bwd_dw2_kernel<f16_t, true>), grid, dim3(512), smem, st, (const float*)p.us,
                         (const float*)p.vs, (const unsigned*)p.bitsN, scores, sid_rows, sid_cols, stats, grad_out, br, b,
                         row_offset, h1, h2, p.rows_per_dsplit, xcd_natural(), p.d_slab);
    } else if constexpr (kF16) {
      MI_SET_DYN_SMEM(concat_bwd_dw2_f16_kernel, Dw2F16Smem::TOTAL, "hipFuncSetAttribute(concat_bwd_dw2_f16_kernel)");
      ProfScope prof_("concat_bwd_dw2_kernel", st);
      hipLaunchKernelGGL(concat_bwd_dw2_f16_kernel, grid, dim3(512), Dw2F16Smem::TOTAL, st, (const unsigned*)p.upk,
                         (const f16_t*)p.vht, (const unsigned*)p.bitsN, scores, sid_rows, sid_cols, stats, br, b,
                         row_offset, h1, h2, p.rows_per_dsplit, xcd_natural(), p.d_slab);
    } else {
      const size_t smem = (256 * 36 + kDw2IB * 32) * sizeof(float) + 256 * sizeof(bf16x8);
      MI_SET_DYN_SMEM((concat_bwd_dw2_kernel<OpT>), smem, "hipFuncSetAttribute(concat_bwd_dw2_kernel)");
      ProfScope prof_("concat_bwd_dw2_kernel", st);
      hipLaunchKernelGGL(concat_bwd_dw2_kernel<OpT>, grid, dim3(512), smem, st, (const float*)p.u, (const float*)p.v,
                         (const unsigned*)p.bitsN, scores, sid_rows, sid_cols, stats, grad_out, br, b, row_offset, h1,
                         h2, p.rows_per_dsplit, xcd_natural(), p.d_slab);
    }
    MI_LAUNCH_CHECK("concat_bwd_dw2_kernel");
    static const bool db2_old = getenv("MI_DB2_OLD") != nullptr;  // A/B switch: the bit-by-bit kernel in the fp16 mode too
    if (kF16 && !X3 && !db2_old) {
      // fp16 mode: g' as an fp16 operand, two bits per v_dot2c_f32_f16 (mi_concat_f16.h)
      const size_t smem2 = 256 * 16 + (size_t)((b + 31) / 32) * 32 * sizeof(f16_t);
      MI_SET_DYN_SMEM((concat_bwd_db2_f16_kernel), smem2, "hipFuncSetAttribute(concat_bwd_db2_f16_kernel)");
      ProfScope prof_("concat_bwd_db2_kernel", st);
      hipLaunchKernelGGL(concat_bwd_db2_f16_kernel, dim3((unsigned)p.n_msplit), dim3(512), smem2, st,
                         (const unsigned*)p.bitsN, scores, sid_rows, sid_cols, stats, grad_out, br, b, row_offset, h2,
                         p.rows_per_msplit, p.m_slab, p.g_sum);
    } else {
      const size_t smem2 = (size_t)((b + 31) / 32) * 32 * sizeof(float);
      MI_SET_DYN_SMEM((concat_bwd_db2_kernel), smem2, "hipFuncSetAttribute(concat_bwd_db2_kernel)");
      ProfScope prof_("concat_bwd_db2_kernel", st);
      hipLaunchKernelGGL(concat_bwd_db2_kernel, dim3((unsigned)p.n_msplit), dim3(512), smem2, st,
                         (const unsigned*)p.bitsN, scores, sid_rows, sid_cols, stats, grad_out, br, b, row_offset, h2,
                         p.rows_per_msplit, p.m_slab, p.g_sum);
    }
    MI_LAUNCH_CHECK("concat_bwd_db2_kernel");
    {
      ProfScope prof_("concat_bwd_finish_w2_kernel", st);
      hipLaunchKernelGGL(concat_bwd_finish_w2_kernel, dim3((unsigned)h2), dim3(256), 0, st, (const float*)p.d_slab,
                         p.n_dsplit, (const float*)p.m_slab, (const float*)p.g_sum, p.n_msplit, w2, b2, w3, h1, h2,
                         grad_w2, grad_w3, grad_b2, grad_b3, kF16 ? (const F16Scales*)p.f16sc : (const F16Scales*)nullptr,
                         grad_out, stats, X3 ? 1 : 0);
    }
    MI_LAUNCH_CHECK("concat_bwd_finish_w2_kernel");
  }

  // ---- first layer: exact fp32 GEMMs (< 1 % of the flops) ------------------------------------------------------------
  const int64_t d = dx + dy;
  int rc = launch_gemm<float>(make_operand((const float*)p.du, h1, 1), make_operand(w1, 1, d), br, dx, h1,
                              EpiStore{grad_x, dx, nullptr, 1.0f, 0}, st, "concat dX = dU W1x");
  if (rc) return rc;
  rc = launch_gemm<float>(make_operand((const float*)p.dv, h1, 1), make_operand(w1 + dx, 1, d), b, dy, h1,
                          EpiStore{grad_y, dy, nullptr, 1.0f, 0}, st, "concat dY = dV W1y");
  if (rc) return rc;
  // dW1x[h, a] = sum_i dU[i, h] X[i, a] and dW1y[h, a] = sum_j dV[j, h] Y[j, a]: K = batch -> split-K slabs, ordered reduce
  {
    EpiStore e{p.w1_slab, dx, nullptr, 1.0f, 0};
    e.slab_stride = (int64_t)h1 * dx;
    const int sx = (int)((br + p.w1_kchunk_x - 1) / p.w1_kchunk_x);
    rc = launch_gemm<float>(make_operand((const float*)p.du, 1, h1), make_operand(x, 1, dx), h1, dx, br, e, st,
                            "concat dW1x = dU^T X", sx, p.w1_kchunk_x);
    if (rc) return rc;
    rc = launch_slab_reduce_ld(p.w1_slab, sx, h1, dx, grad_w1, d, st, "slab_reduce_ld_kernel");
    if (rc) return rc;
    EpiStore e2{p.w1_slab, dy, nullptr, 1.0f, 0};
    e2.slab_stride = (int64_t)h1 * dy;
    const int sy = (int)((b + p.w1_kchunk_y - 1) / p.w1_kchunk_y);
    rc = launch_gemm<float>(make_operand((const float*)p.dv, 1, h1), make_operand(y, 1, dy), h1, dy, b, e2, st,
                            "concat dW1y = dV^T Y", sy, p.w1_kchunk_y);
    if (rc) return rc;
    rc = launch_slab_reduce_ld(p.w1_slab, sy, h1, dy, grad_w1 + dx, d, st, "slab_reduce_ld_kernel");
    if (rc) return rc;
  }
  {
    ProfScope prof_("colsum_kernel", st);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)((h1 + 63) / 64), 64), dim3(256), 0, st,
                       (const float*)p.dv, b, (int64_t)h1, p.col_part);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((h1 + 255) / 256)), dim3(256), 0, st,
                       (const float*)p.col_part, 64, (int64_t)h1, grad_b1);
  }
  MI_LAUNCH_CHECK("colsum_kernel");
  return MI_OK;
}

}  // namespace mi

using namespace mi;

extern "C" {

size_t mi_concat_mlp_workspace_bytes(int64_t b_rows, int64_t b, int64_t d_img, int64_t d_txt, int64_t h1, int64_t h2,
                                     int precision, int need_grad) {
  if (b_rows <= 0 || b <= 0 || d_img <= 0 || d_txt <= 0 || h1 <= 0 || h2 <= 0) return 0;
  Workspace ws(nullptr, 0);
  return plan_concat(ws, b_rows, b, h1, h2, precision, need_grad, d_img > d_txt ? d_img : d_txt).bytes + 256;
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
  ConcatPlan p = plan_concat(ws, b_rows, b, h1, h2, precision, need_grad, d_img > d_txt ? d_img : d_txt);
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
  if (precision == MI_PREC_F16X3) {
    // the fp32-tolerance mode: scaled fp32 U, V, two-part fp16 W2, three MFMAs per product (mi_concat_f16.h)
    rc = launch_zero_words(p.f16sc, sizeof(F16Scales), st, "zero_words_kernel(fp16 scales)");
    if (rc) return rc;
    F16AbsmaxJobs aj{{p.u, p.v, w2, w3}, {b_rows * h1, b * h1, h2 * h1, h2}, p.f16sc};
    {
      ProfScope prof_("f16 absmax U V W2 w3", st);
      hipLaunchKernelGGL(f16_absmax_kernel, dim3(128, 4), dim3(256), 0, st, aj);
    }
    MI_LAUNCH_CHECK("f16_absmax_kernel");
    F16X3PrepArgs pa{p.u, p.v, w2, b_rows, b, (int)h1, (int)h2, p.f16sc, p.us, p.vs, p.w2x};
    {
      ProfScope prof_("f16x3 prep Us Vs W2x", st);
      hipLaunchKernelGGL(f16x3_prep_kernel, dim3(1024), dim3(256), 0, st, pa);
    }
    MI_LAUNCH_CHECK("f16x3_prep_kernel");
    const dim3 grid1d((unsigned)(8 * (((b + kFwdTJ - 1) / kFwdTJ + 7) / 8) * ((b_rows + kFwdTI - 1) / kFwdTI)));
    MI_SET_DYN_SMEM(concat_fwd_f16x3_kernel, FwdF16Smem::TOTAL, "hipFuncSetAttribute(concat_fwd_f16x3_kernel)");
    {
      ProfScope prof_("concat_fwd_kernel", st);
      hipLaunchKernelGGL(concat_fwd_f16x3_kernel, grid1d, dim3(256), FwdF16Smem::TOTAL, st, (const float*)p.us,
                         (const float*)p.vs, (const f16_t*)p.w2x, b2, w3, b3, (const F16Scales*)p.f16sc, b_rows, b, (int)h1,
                         (int)h2, scores_out, bitsP, bitsN, xcd_natural());
    }
    MI_LAUNCH_CHECK("concat_fwd_f16x3_kernel");
    rc = MI_OK;
  } else if (precision == MI_PREC_F16) {
    // absmax of U, V, W2, w3 -> power-of-two scales (device side) -> fp16 operand copies -> the packed-generation kernel
    rc = launch_zero_words(p.f16sc, sizeof(F16Scales), st, "zero_words_kernel(fp16 scales)");
    if (rc) return rc;
    F16AbsmaxJobs aj{{p.u, p.v, w2, w3}, {b_rows * h1, b * h1, h2 * h1, h2}, p.f16sc};
    {
      ProfScope prof_("f16 absmax U V W2 w3", st);
      hipLaunchKernelGGL(f16_absmax_kernel, dim3(128, 4), dim3(256), 0, st, aj);
    }
    MI_LAUNCH_CHECK("f16_absmax_kernel");
    F16PrepArgs pa{p.u, p.v, w2, b_rows, b, (int)h1, (int)h2, p.f16sc, p.uh, p.vh, p.w2h, need_grad ? p.upk : nullptr,
                   need_grad ? p.vht : nullptr};
    {
      ProfScope prof_("f16 prep Uh Vh W2h", st);
      hipLaunchKernelGGL(f16_prep_kernel, dim3(1024), dim3(256), 0, st, pa);
      if (need_grad)
        hipLaunchKernelGGL(f16_transpose_v_kernel, dim3((unsigned)(h1 / 32), (unsigned)((b + 31) / 32)), dim3(256), 0, st,
                           (const f16_t*)p.vh, b, (int)h1, p.vht);
    }
    MI_LAUNCH_CHECK("f16_prep_kernel");
    const dim3 grid1d((unsigned)(8 * (((b + kFwdTJ - 1) / kFwdTJ + 7) / 8) * ((b_rows + kFwdTI - 1) / kFwdTI)));
    MI_SET_DYN_SMEM(concat_fwd_f16_kernel, FwdF16Smem::TOTAL, "hipFuncSetAttribute(concat_fwd_f16_kernel)");
    {
      ProfScope prof_("concat_fwd_kernel", st);
      hipLaunchKernelGGL(concat_fwd_f16_kernel, grid1d, dim3(256), FwdF16Smem::TOTAL, st, (const f16_t*)p.uh,
                         (const f16_t*)p.vh, (const f16_t*)p.w2h, b2, w3, b3, (const F16Scales*)p.f16sc, b_rows, b, (int)h1,
                         (int)h2, scores_out, bitsP, bitsN, xcd_natural());
    }
    MI_LAUNCH_CHECK("concat_fwd_f16_kernel");
    rc = MI_OK;
  } else if (precision == MI_PREC_BF16) {
    {
      ProfScope prof_("f32_to_bf16_kernel", st);
      hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(512), dim3(256), 0, st, w2, p.w2bf, h1 * h2);
    }
    MI_LAUNCH_CHECK("f32_to_bf16_kernel");
    rc = launch_concat_fwd<bf16_t>(p.u, p.v, p.w2bf, b2, w3, b3, b_rows, b, (int)h1, (int)h2, scores_out, bitsP, bitsN,
                                   st);
  } else {
    rc = launch_concat_fwd<float>(p.u, p.v, w2, b2, w3, b3, b_rows, b, (int)h1, (int)h2, scores_out, bitsP, bitsN, st);
  }
  if (rc) return rc;
  const int grid = (int)(b_rows < kMatrixPartialBlocks ? b_rows : kMatrixPartialBlocks);
  {
    ProfScope prof_("matrix_partials_kernel", st);
    hipLaunchKernelGGL(matrix_partials_kernel, dim3(grid), dim3(256), 0, st, (const float*)scores_out, sid_rows,
                       sid_cols, b_rows, b, row_offset, p.partials);
  }
  MI_LAUNCH_CHECK("matrix_partials_kernel");
  return launch_finalize(p.partials, grid, b, estimator, loss_out, stats, partials_out, st);
}

int mi_concat_mlp_bwd(const float* x, const float* y, const float* w1, const float* b1, const float* w2, const float* b2,
                      const float* w3, const float* b3, const int64_t* sid_rows, const int64_t* sid_cols, int64_t b_rows,
                      int64_t b, int64_t row_offset, int64_t d_img, int64_t d_txt, int64_t h1, int64_t h2, int precision,
                      const mi_stats* stats, const float* grad_out, const float* scores, float* grad_x, float* grad_y,
                      float* grad_w1, float* grad_b1, float* grad_w2, float* grad_b2, float* grad_w3, float* grad_b3,
                      void* workspace, size_t workspace_bytes, void* stream) {
  (void)b1;
  (void)b3;
  MI_CHECK_ARG(x && y && w1 && w2 && b2 && w3 && sid_rows && sid_cols && stats && scores && grad_x && grad_y &&
                   grad_w1 && grad_b1 && grad_w2 && grad_b2 && grad_w3 && grad_b3 && workspace,
               "mi_concat_mlp_bwd: null pointer");
  int rc = check_concat_shape("mi_concat_mlp_bwd", b_rows, b, row_offset, d_img, d_txt, h1, h2, precision);
  if (rc) return rc;
  Workspace ws(workspace, workspace_bytes);
  ConcatPlan p = plan_concat(ws, b_rows, b, h1, h2, precision, 1, d_img > d_txt ? d_img : d_txt);
  if (!ws.ok()) {
    set_error("mi_concat_mlp_bwd: workspace too small (%zu < %zu): pass the workspace of the forward call made with "
              "need_grad = 1",
              workspace_bytes, ws.off);
    return MI_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  if (precision == MI_PREC_F16X3)
    return concat_bwd_impl<f16_t, true>(x, y, w1, w2, b2, w3, sid_rows, sid_cols, b_rows, b, row_offset, d_img, d_txt,
                                        (int)h1, (int)h2, stats, grad_out, scores, grad_x, grad_y, grad_w1, grad_b1, grad_w2,
                                        grad_b2, grad_w3, grad_b3, p, st);
  if (precision == MI_PREC_F16)
    return concat_bwd_impl<f16_t>(x, y, w1, w2, b2, w3, sid_rows, sid_cols, b_rows, b, row_offset, d_img, d_txt,
                                  (int)h1, (int)h2, stats, grad_out, scores, grad_x, grad_y, grad_w1, grad_b1, grad_w2,
                                  grad_b2, grad_w3, grad_b3, p, st);
  if (precision == MI_PREC_BF16)
    return concat_bwd_impl<bf16_t>(x, y, w1, w2, b2, w3, sid_rows, sid_cols, b_rows, b, row_offset, d_img, d_txt,
                                   (int)h1, (int)h2, stats, grad_out, scores, grad_x, grad_y, grad_w1, grad_b1, grad_w2,
                                   grad_b2, grad_w3, grad_b3, p, st);
  return concat_bwd_impl<float>(x, y, w1, w2, b2, w3, sid_rows, sid_cols, b_rows, b, row_offset, d_img, d_txt, (int)h1,
                                (int)h2, stats, grad_out, scores, grad_x, grad_y, grad_w1, grad_b1, grad_w2, grad_b2,
                                grad_w3, grad_b3, p, st);
}

}  // extern "C"
