/*
 * mi_critic.h -- C ABI of the MI355X-native mutual-information critic path (libmi_critic_hip.so).
 *
 * The reference (vnoz/Mutual-Information-MultiModal) is pure Python and has no FFI; the interface each entry
 * point replaces is therefore a Python callable.  Citations are file:line relative to the reference root.
 *
 *   mi_bound_*            <- mutual_info_img_txt/mi_critics.py:3-12  (dv_bound_loss)
 *                            mutual_info_img_txt/mi_critics.py:14-23 (infonce_bound_loss)
 *   mi_pair_index,
 *   mi_create_pairs*      <- MultiModalManager.create_mi_pairs, mutual_info_img_txt/main_utils.py:80-110
 *   mi_concat_mlp_*       <- the call site mutual_info_img_txt/main_utils.py:220-226:
 *                            create_mi_pairs -> mi_discriminator (make_mlp(1536,[1024,512]), model.py:18-32,
 *                            instantiated main_utils.py:77) -> mi_critic -> loss.backward()
 *   mi_separable_*        <- same call site with the separable critic S = (X Wg)(Y Wh)^T (BASELINE.json configs[1])
 *   mi_bilinear_*         <- same call site with the bilinear critic S = (X W) Y^T named by BASELINE.json
 *                            (an extension: the reference has no bilinear critic; the bound, the masking and the
 *                            pair semantics applied to its scores are the reference's)
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; all tensors are dense row-major
 *   - embeddings, parameters, scores and gradients are float32 (the reference is fp32 throughout);
 *     study ids are int64 codes (equal code <=> equal study id; the reference compares ids with != only,
 *     main_utils.py:105)
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work on it and never synchronise
 *     (the reference synchronises only at loss.item(), main_utils.py:233); the only exception is
 *     mi_pairs_count_host, which returns a host integer
 *   - the callee never allocates or frees: outputs, saved statistics and workspace are caller-owned;
 *     query sizes with the *_workspace_bytes functions (pure host arithmetic; 0 for a non-positive size)
 *   - return value: 0 on success, negative MI_E* code on failure; mi_last_error() describes the last failure
 *     on the calling thread.  No C++ exception crosses this boundary.
 *   - re-entrant: forward and backward may be called from different host threads (autograd does).  Process-wide state is
 *     limited to (a) per-device caches of kernel attributes (atomic, raised under a mutex) and (b) the optional
 *     profiling hook mi_profile_begin/end, which is NOT thread-safe (single-threaded benchmarking only).  The library
 *     never sets the device: the caller makes the tensors' device current (the Python binding does).
 *
 * estimator: MI_DV = 0 (loss = LSE(neg) - log N_neg - mean(pos)), MI_INFONCE = 1 (no log N_neg term).
 * precision: MI_PREC_F32 = 0 (fp32-input MFMA, exact fp32 products, parity mode),
 *            MI_PREC_BF16 = 1 (bf16 MFMA operands, fp32 accumulate),
 *            MI_PREC_BF16X3 = 2 (bilinear critic with a weight matrix only: every operand split into two bf16 parts,
 *            three bf16 MFMAs per product, fp32 accumulate -- products good to ~2^-16 at a third of the bf16 rate; the
 *            separable critic runs its fp32 path under this code, the concat-MLP entry points reject it),
 *            MI_PREC_FP8 = 3 (bilinear critic with a weight matrix only; widths multiples of 16: x, y, W and T = x W are
 *            quantised to OCP e4m3 with per-tensor scales absmax / 448 computed on the device, both forward products
 *            run on the fp8 MFMA with fp32 accumulation, the backward is straight-through on the quantised values
 *            with bf16 MFMA operands; other entry points reject it),
 *            MI_PREC_F16 = 4 (concat-MLP critic only: fp16 MFMA operands -- U, V, W2 and w3 W2 scaled by powers of two
 *            derived on the device from their absmax, the generated operand relu(U_i + V_j) formed by packed fp16
 *            arithmetic -- fp32 accumulate; the fast mode of the reference's critic, csrc/mi_concat_f16.h).
 */
#ifndef MI_CRITIC_H
#define MI_CRITIC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_OK 0
#define MI_EINVAL (-1)   /* bad argument (null pointer, negative size, unknown enum) */
#define MI_ESHAPE (-2)   /* shape not supported by the fused kernels */
#define MI_EWORKSPACE (-3) /* workspace too small */
#define MI_EHIP (-4)     /* a HIP runtime call failed */

#define MI_DV 0
#define MI_INFONCE 1

#define MI_PREC_F32 0
#define MI_PREC_BF16 1
#define MI_PREC_BF16X3 2
#define MI_PREC_FP8 3
#define MI_PREC_F16 4   /* concat-MLP critic: fp16 MFMA operands with power-of-two tensor scales, fp32 accumulate */
#define MI_PREC_F16X3 5 /* concat-MLP critic: every operand as two fp16 parts, three MFMAs per product (fp32 tolerances) */

/* Statistics block written by every forward call and read by the matching backward call (64 bytes). */
typedef struct mi_stats {
  float lse;          /* log-sum-exp over the negative scores                                  */
  float pos_mean;     /* mean of the positive scores                                           */
  float loss_dv;      /* lse - log(n_neg) - pos_mean      (mi_critics.py:10,12)                */
  float loss_infonce; /* lse - pos_mean                   (mi_critics.py:21,23)                */
  float log_n_neg;    /* logf((float)n_neg), float32 as in the reference (mi_critics.py:10)    */
  float neg_max;      /* running maximum of the negative scores                                */
  float reserved0, reserved1;
  int64_t n_neg;      /* number of negative rows N - pos_size                                  */
  int64_t n_pos;      /* pos_size                                                              */
  int64_t reserved2, reserved3;
} mi_stats;

/* Kernel paths reported by mi_bilinear_path / mi_separable_path (host-side queries; nothing is launched) */
#define MI_PATH_GENERIC 0    /* strided-operand kernels of mi_gemm.h (fp32 parity mode, odd shapes)            */
#define MI_PATH_GEMMS 1      /* 16-bit GEMM chain with G / G^T materialised (widths outside the fused kernel) */
#define MI_PATH_FUSED 2      /* fused B x B kernel + the three-launch backward tail                          */
#define MI_PATH_FUSED_TAIL 3 /* fused B x B kernel + the two-launch tail (the four-launch step)              */
#define MI_PATH_FP8_GEMMS 4  /* fp8 forward products + bf16 backward chain                                   */

int mi_abi_version(void);  /* 4: bf16 / fp16 boundary, path queries, MI_PREC_F16 / MI_PREC_F16X3 */
const char* mi_last_error(void);

/* Optional per-kernel timing (bench.py's roofline leg): between mi_profile_begin and mi_profile_end every kernel
 * launch of this library is bracketed by HIP events on its stream.  mi_profile_end synchronises the device and returns
 * up to `capacity` (name, milliseconds) records in launch order; names is a NUL-separated list.  Not thread-safe. */
int mi_profile_begin(void);
int mi_profile_end(char* names, size_t names_bytes, float* ms, int capacity, int* n_out);

/* ---- a3 / a4: bound on materialised logits (mi_critics.py:3-23) ------------------------------------ */
size_t mi_bound_workspace_bytes(int64_t n);
/* logits[n] (the reference's [N,1] tensor), first pos_size rows positive.  Writes *stats and loss_out[0]. */
int mi_bound_fwd(const float* logits, int64_t n, int64_t pos_size, int estimator, float* loss_out,
                 mi_stats* stats, void* workspace, size_t workspace_bytes, void* stream);
/* grad_logits[r] = grad_out[0] * (r < pos ? -1/pos : exp(logits[r] - lse)); identical for both estimators. */
int mi_bound_bwd(const float* logits, int64_t n, int64_t pos_size, const mi_stats* stats, const float* grad_out,
                 float* grad_logits, void* stream);

/* ---- bound on a B x B score matrix with study-id masking ------------------------------------------- */
/* positives = diagonal; negatives = (i != j and sid[i] != sid[j]); other pairs are dropped (main_utils.py:105) */
size_t mi_matrix_bound_workspace_bytes(int64_t b);
int mi_matrix_bound_fwd(const float* scores, const int64_t* sid, int64_t b, int estimator, float* loss_out,
                        mi_stats* stats, void* workspace, size_t workspace_bytes, void* stream);
/* grad_scores[i,j] = grad_out[0] * dloss/dS[i,j] */
int mi_matrix_bound_bwd(const float* scores, const int64_t* sid, int64_t b, const mi_stats* stats,
                        const float* grad_out, float* grad_scores, void* stream);

/* ---- a1: pair builder (main_utils.py:80-110) ------------------------------------------------------- */
/* number of rows N of mi_input for these ids (host result; synchronises the stream) */
int mi_pairs_count_host(const int64_t* sid, int64_t b, int64_t* n_rows_host, void* workspace,
                        size_t workspace_bytes, void* stream);
size_t mi_pair_index_workspace_bytes(int64_t b);
/* pair_i/pair_j[capacity] receive the (i,j) of every row in reference order (positives first, then gap-major /
 * i-minor negatives); n_rows_dev[0] receives N.  rowpos (optional, [b*(b-1)] int32) receives for every (gap,i)
 * the output row or -1 when the pair is dropped. */
int mi_pair_index(const int64_t* sid, int64_t b, int32_t* pair_i, int32_t* pair_j, int64_t capacity,
                  int64_t* n_rows_dev, int32_t* rowpos, void* workspace, size_t workspace_bytes, void* stream);
/* out[n_rows, d_img + d_txt] = [img[pair_i[r]] ; txt[pair_j[r]]] */
int mi_create_pairs(const float* embedding_img, const float* embedding_txt, const int32_t* pair_i,
                    const int32_t* pair_j, int64_t n_rows, int64_t d_img, int64_t d_txt, float* out, void* stream);
/* grad_img[i] = sum of grad_out rows whose image index is i (left part), same for txt (right part); fixed
 * summation order (positive row, then gaps ascending). */
int mi_create_pairs_bwd(const float* grad_out, const int32_t* rowpos, int64_t b, int64_t d_img, int64_t d_txt,
                        float* grad_img, float* grad_txt, void* stream);

/* ---- fused bilinear critic: S = (X W) Y^T, bound, all gradients ------------------------------------- */
/* X [b_rows, d_img] (the local row block), Y [b, d_txt] (all columns), W [d_img, d_txt], sid_rows [b_rows],
 * sid_cols [b]; row_offset = global index of local row 0 (diagonal of the global B x B matrix).  Single GPU:
 * b_rows = b, row_offset = 0.  scores_out (optional) [b_rows, b].  w == NULL selects the separable form
 * S = X Y^T on already-projected embeddings (d_img == d_txt; grad_w unused).
 * need_grad bit 0: the forward's fused B x B launch also accumulates the two gradient contractions (the loss has one
 * global log-sum-exp, so they only need a scale once it is known); the matching backward then never recomputes scores. */
size_t mi_bilinear_workspace_bytes(int64_t b_rows, int64_t b, int64_t d_img, int64_t d_txt, int precision);
int mi_bilinear_fwd(const float* x, const float* y, const float* w, const int64_t* sid_rows,
                    const int64_t* sid_cols, int64_t b_rows, int64_t b, int64_t row_offset, int64_t d_img,
                    int64_t d_txt, int estimator, int precision, int need_grad, float* loss_out, mi_stats* stats,
                    float* partials_out, float* scores_out, void* workspace, size_t workspace_bytes, void* stream);
/* stats must hold the GLOBAL lse / n_pos (after the cross-rank merge when sharded). grad_out[0] = dL/dloss.
 * Outputs: grad_x [b_rows, d_img], grad_y [b, d_txt] (partial over this row block), grad_w [d_img, d_txt].
 * workspace_from_forward != 0: `workspace` is the buffer the matching mi_bilinear_fwd call wrote (same inputs,
 * need_grad != 0); the backward then reuses the operand copies, T and the fused sums found there instead of
 * rebuilding them. */
int mi_bilinear_bwd(const float* x, const float* y, const float* w, const int64_t* sid_rows,
                    const int64_t* sid_cols, int64_t b_rows, int64_t b, int64_t row_offset, int64_t d_img,
                    int64_t d_txt, int precision, const mi_stats* stats, const float* grad_out, float* grad_x,
                    float* grad_y, float* grad_w, void* workspace, size_t workspace_bytes,
                    int workspace_from_forward, void* stream);

/* Sharded batches (b_rows < b): the part of the forward's preparation that depends on the rank's own rows only (bf16
 * copies of X and W, T = X W), so that it runs while the all-gather of the text embeddings is in flight (SURVEY.md 8e:
 * "overlap the gather with local work").  Then mi_bilinear_fwd(..., need_grad | 4, ...) on the SAME workspace (bit 2 of
 * need_grad: the local part is prepared).  MI_ESHAPE where the shape does not take the fused kernels: call
 * mi_bilinear_fwd alone then. */
int mi_bilinear_prep_local(const float* x, const float* w, int64_t b_rows, int64_t b, int64_t d_img, int64_t d_txt,
                           int precision, void* workspace, size_t workspace_bytes, void* stream);

/* Sharded batches without the finalize and merge launches: mi_bilinear_fwd(..., need_grad | 8, ...) leaves one 16-byte
 * record per wave of the fused kernel in the workspace and writes no loss / stats / partials_out;
 * mi_bilinear_raw_records returns their count (0: the shape does not take this path) and, through offset_bytes, where
 * they start in the workspace.  All-gather that region from every rank (rank order) and call mi_bilinear_bwd_records on
 * the forward's workspace: its first launch merges all n_records records in the given order on every workgroup
 * (bit-identical loss / stats on every rank; n_pos = the global batch), then computes the gradients as mi_bilinear_bwd. */
size_t mi_bilinear_raw_records(int64_t b_rows, int64_t b, int64_t d_img, int64_t d_txt, int precision,
                               size_t* offset_bytes);
int mi_bilinear_bwd_records(const float* x, const float* y, const float* w, const int64_t* sid_rows,
                            const int64_t* sid_cols, int64_t b_rows, int64_t b, int64_t row_offset, int64_t d_img,
                            int64_t d_txt, int precision, int estimator, const float* records, int64_t n_records,
                            int64_t n_pos, const float* grad_out, float* loss_out, mi_stats* stats_out, float* grad_x,
                            float* grad_y, float* grad_w, void* workspace, size_t workspace_bytes, void* stream);
/* mi_bilinear_bwd_records with grad_w == NULL stops after its first launch (statistics, loss, grad_x, the partial grad_y);
 * mi_bilinear_bwd_dw then launches dW = X^T dT from the same workspace.  Between the two a sharded step starts the
 * reduce-scatter of grad_y, which so overlaps the dW launch (ABI 4). */
int mi_bilinear_bwd_dw(int64_t b_rows, int64_t b, int64_t d_img, int64_t d_txt, int precision, float* grad_w,
                       void* workspace, size_t workspace_bytes, void* stream);


/* fp8 mode (MI_PREC_FP8) on a sharded batch: the per-tensor scales must be the whole batch's.  The forward's preparation
 * in three stages around the caller's two MAX all-reduces of amax_io (4 floats on the device: x, y, w, T):
 *   stage 0 -> amax_io[0..2] | all-reduce MAX | stage 1 -> amax_io[3] | all-reduce MAX | stage 2,
 * then mi_bilinear_fwd(..., need_grad | 2, ...) on the SAME workspace (bit 1 of need_grad: the fp8 operands are staged)
 * and mi_bilinear_bwd(..., workspace_from_forward = 1).  BASELINE.json configs[4] (fp8, 8 GPUs). */
int mi_bilinear_fp8_stage(const float* x, const float* y, const float* w, int64_t b_rows, int64_t b, int64_t d_img,
                          int64_t d_txt, int stage, float* amax_io, void* workspace, size_t workspace_bytes, void* stream);

/* One critic step in ONE call (single GPU; the sharded case needs the cross-rank merge between the forward and the
 * backward and uses the two calls above): forward, statistics, loss and every gradient of  grad_out[0] * loss
 * (grad_out == NULL: 1).  Same inputs, outputs and workspace as mi_bilinear_fwd + mi_bilinear_bwd with b_rows == b,
 * row_offset == 0; replaces the whole call site main_utils.py:220-226 (pairs -> critic -> bound -> loss.backward()).
 * Where the fused kernels take the shape it saves the finalize launch: nothing needs the loss between the fused B x B
 * kernel and the gradients, so the per-wave records are merged by the launch that also turns the partial sums into
 * dT / grad_y / grad_x.  loss_out, *stats and partials_out (optional) are valid when the call's work has completed. */
int mi_bilinear_step(const float* x, const float* y, const float* w, const int64_t* sid, int64_t b, int64_t d_img,
                     int64_t d_txt, int estimator, int precision, const float* grad_out, float* loss_out,
                     mi_stats* stats, float* partials_out, float* grad_x, float* grad_y, float* grad_w, void* workspace,
                     size_t workspace_bytes, void* stream);

/* The step at a bf16 boundary (ABI 4): x_bf16 [b][d_img], y_bf16 [b][d_txt] hold bfloat16 -- what the encoders emit under
 * autocast (model.py:540-555 run in bf16) --; grad_x / grad_y are bfloat16 buffers when grads_bf16 != 0, float32 buffers
 * otherwise; w, grad_w, the loss and the statistics are float32.  The kernels' first act on fp32 embeddings is to round
 * them to bf16, so this entry point computes the same bits as mi_bilinear_step(precision = MI_PREC_BF16) on the same
 * bf16-representable values; what it saves is the 25 MB of conversion traffic per step at B = 4096, d = 512.  Shapes:
 * mi_bilinear_path(b, b, d_img, d_txt, MI_PREC_BF16) == MI_PATH_FUSED_TAIL, 16-byte aligned embeddings; else MI_ESHAPE. */
int mi_bilinear_step_bf16(const void* x_bf16, const void* y_bf16, const float* w, const int64_t* sid, int64_t b,
                          int64_t d_img, int64_t d_txt, int estimator, const float* grad_out, float* loss_out,
                          mi_stats* stats, float* partials_out, void* grad_x, void* grad_y, int grads_bf16,
                          float* grad_w, void* workspace, size_t workspace_bytes, void* stream);

/* Which kernels a shape takes: one of MI_PATH_* (or a negative MI_E* code).  Host-side arithmetic on the plan only; the
 * Python binding uses it to warn once per shape when a 16-bit call leaves the fused kernels. */
int mi_bilinear_path(int64_t b_rows, int64_t b, int64_t d_img, int64_t d_txt, int precision);
int mi_separable_path(int64_t b_rows, int64_t b, int64_t d_img, int64_t d_txt, int64_t d_proj, int precision);

/* ---- fused separable critic: S = (X Wg)(Y Wh)^T, bound, all gradients ------------------------------- */
/* BASELINE.json configs[1] (an extension: the reference has no separable critic; bound, masking and pair semantics are
 * the reference's).  X [b_rows, d_img], Y [b, d_txt], Wg [d_img, d_proj], Wh [d_txt, d_proj].  The projections run on
 * this library's MFMA kernels (no library GEMM).  Sharding arguments as for mi_bilinear_*. */
size_t mi_separable_workspace_bytes(int64_t b_rows, int64_t b, int64_t d_img, int64_t d_txt, int64_t d_proj,
                                    int precision);
int mi_separable_fwd(const float* x, const float* y, const float* wg, const float* wh, const int64_t* sid_rows,
                     const int64_t* sid_cols, int64_t b_rows, int64_t b, int64_t row_offset, int64_t d_img,
                     int64_t d_txt, int64_t d_proj, int estimator, int precision, int need_grad, float* loss_out,
                     mi_stats* stats, float* partials_out, void* workspace, size_t workspace_bytes, void* stream);
/* grad_y [b, d_txt] is the partial over this row block; grad_wg / grad_wh are this row block's contributions. */
int mi_separable_bwd(const float* x, const float* y, const float* wg, const float* wh, const int64_t* sid_rows,
                     const int64_t* sid_cols, int64_t b_rows, int64_t b, int64_t row_offset, int64_t d_img,
                     int64_t d_txt, int64_t d_proj, int precision, const mi_stats* stats, const float* grad_out,
                     float* grad_x, float* grad_y, float* grad_wg, float* grad_wh, void* workspace,
                     size_t workspace_bytes, int workspace_from_forward, void* stream);

/* One separable-critic step in one call (single GPU; ABI 4): as mi_bilinear_step.  Five launches where the fused kernels
 * take the shape (mi_separable_path == MI_PATH_FUSED_TAIL): conversions, the two projections, the fused B x B kernel,
 * [statistics + dA, dC rows + dX, dY on the matrix cores], [dWg | dWh]. */
int mi_separable_step(const float* x, const float* y, const float* wg, const float* wh, const int64_t* sid, int64_t b,
                      int64_t d_img, int64_t d_txt, int64_t d_proj, int estimator, int precision, const float* grad_out,
                      float* loss_out, mi_stats* stats, float* partials_out, float* grad_x, float* grad_y,
                      float* grad_wg, float* grad_wh, void* workspace, size_t workspace_bytes, void* stream);

/* ---- fused concat-MLP critic (the reference's mi_discriminator) ------------------------------------ */
/* params in PyTorch [out,in] layout: w1 [h1, d_img+d_txt], b1 [h1], w2 [h2, h1], b2 [h2], w3 [h2], b3 [1]. */
size_t mi_concat_mlp_workspace_bytes(int64_t b_rows, int64_t b, int64_t d_img, int64_t d_txt, int64_t h1,
                                     int64_t h2, int precision, int need_grad);
int mi_concat_mlp_fwd(const float* x, const float* y, const float* w1, const float* b1, const float* w2,
                      const float* b2, const float* w3, const float* b3, const int64_t* sid_rows,
                      const int64_t* sid_cols, int64_t b_rows, int64_t b, int64_t row_offset, int64_t d_img,
                      int64_t d_txt, int64_t h1, int64_t h2, int estimator, int precision, int need_grad,
                      float* loss_out, mi_stats* stats, float* partials_out, float* scores_out /* [b_rows,b] */,
                      void* workspace, size_t workspace_bytes, void* stream);
int mi_concat_mlp_bwd(const float* x, const float* y, const float* w1, const float* b1, const float* w2,
                      const float* b2, const float* w3, const float* b3, const int64_t* sid_rows,
                      const int64_t* sid_cols, int64_t b_rows, int64_t b, int64_t row_offset, int64_t d_img,
                      int64_t d_txt, int64_t h1, int64_t h2, int precision, const mi_stats* stats,
                      const float* grad_out, const float* scores /* [b_rows,b] from fwd */, float* grad_x,
                      float* grad_y, float* grad_w1, float* grad_b1, float* grad_w2, float* grad_b2,
                      float* grad_w3, float* grad_b3, void* workspace, size_t workspace_bytes, void* stream);

/* ---- cross-rank merge of per-rank partial statistics (global-batch negatives) -------- */
/* partials [n_ranks][4] = (neg_max, sum exp(s - neg_max), sum of positives, n_neg as float pair) gathered in
 * rank order; writes the global stats and loss.  Merging in rank order makes every rank compute identical bits. */
int mi_merge_partials(const float* partials, int64_t n_ranks, int64_t n_pos_global, int estimator,
                      float* loss_out, mi_stats* stats, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI_CRITIC_H */
