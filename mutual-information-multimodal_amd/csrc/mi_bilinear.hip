// Fused bilinear critic  S = (X W) Y^T  + bound + all gradients (BASELINE.json headline configuration).
// Replaces the call site mutual_info_img_txt/main_utils.py:220-226 for a bilinear critic: the B x B score matrix
// is never written to HBM in the forward (the log-sum-exp is an epilogue of the score GEMM) and the pair rows of
// create_mi_pairs (main_utils.py:80-110) are index arithmetic plus a study-id compare in that epilogue.
// The reference has no bilinear critic: scorer parity is pinned by the oracle only; bound, masking and pair
// semantics are the reference's (mi_critics.py:3-23, main_utils.py:99-108).
//
// bf16 fast path (all widths and batch sizes multiples of 8): every operand is converted once to bf16 in both
// orientations so that every GEMM reads K-contiguous rows (mi_gemm_bf16.h):
//   forward : prep X, Y, W -> T = X W (bf16, both orientations) -> S tiles = T Y^T with masked online-LSE epilogue
//             -> fixed-order merge -> stats, loss
//   backward: G = dL/dS from recomputed S tiles (bf16, both orientations, 2 x 33 MB at B = 4096: cache resident)
//             -> {dT = G Y, dY = G^T T} as ONE two-problem launch (2 x 128 tiles fill the chip) -> dW = X^T dT (split-K
//             slabs + ordered reduce) -> dX = dT W^T.  The backward reuses the forward's workspace (X, Y, W, T copies).
// generic path (fp32 parity mode, odd shapes): mi_gemm.h kernels with strided operands, T recomputed in the backward.
#include "mi_gemm.h"
#include "mi_gemm_bf16.h"
#include "mi_bilinear_flash.h"
#include "mi_bilinear_tail.h"
#include "mi_fp8.h"

namespace mi {

struct BilinearPlan {
  // generic path
  float* t;
  float* dt;
  void* g;
  Partial* partials;
  int64_t n_partials;
  // bf16 fast path
  bf16_t *xb, *xtb, *yb, *ytb, *wb, *wtb, *tb, *ttb, *gb, *gtb, *dtb, *dttb;
  float* dw_slab;
  int dw_splits;
  int64_t dw_kchunk;
  // row blocks of a sharded batch (b_rows < b): dT = G Y has few output tiles and a long K; split K into slabs
  float* dt_slab;
  int dt_splits;
  int64_t dt_kchunk;
  // fused B x B stage (mi_bilinear_flash.h): partial sums and per-wave records of the two problems
  FlashPlan fl;
  float* fl_slab[2];
  Partial* fl_rec[2];
  unsigned char* fl_dup[2];  // equal-id flags per 32 x 32 block: [br / 32][b / 32] and its transpose
  bf16_t *tfb, *yfb;         // fragment-major copies of T and Y: the stationary operands of the fused kernel
  // the backward's tail (mi_bilinear_tail.h): fragment-major W (B operand of dX = dT W^T), X^T and dT^T (operands of dW)
  bool tail;
  bf16_t *wfb, *xtfb, *dttfb;
  // bf16x3 (MI_PREC_BF16X3): every bf16 operand copy holds hi and lo parts along a tripled K (mi_gemm_bf16.h,
  // split3_offsets); A-side role: xb, xtb, tb, gb, gtb, dtb; B-side role: yb, ytb, wb, wtb, ttb, dttb
  int x3;  // 3 in that mode, else 1
  // fp8 mode (MI_PREC_FP8, mi_fp8.h): e4m3 copies of the operands of the two forward products and the scale block; the
  // bf16 copies of the quantised values for the backward live in xtb, ytb, wb, ttb
  fp8_t *qx8, *qy8, *qwt8, *qt8;
  Fp8Scales* f8sc;
  // exact-fp32 path (MI_PREC_F32): split-K slabs of the products whose output has too few 128 x 128 tiles to fill the chip
  float* gen_slab;
  int64_t gen_slab_floats;
  size_t bytes;
};

// Split-K of a generic-path product: with fewer than 192 output tiles the launch leaves CUs idle (dW = X^T dT at d = 512 is
// 16 tiles with K = B: 567 us for 2 GFLOP at B = 4096).  Splits fill ~256 workgroups, K chunks are multiples of 64.
static void generic_splitk(int64_t m, int64_t n, int64_t k, int& splits, int64_t& kchunk) {
  const int64_t tiles = ((m + kTile - 1) / kTile) * ((n + kTile - 1) / kTile);
  int64_t sp = tiles >= 192 ? 1 : (256 + tiles - 1) / tiles;
  if (sp > 16) sp = 16;
  kchunk = (k + sp - 1) / sp;
  kchunk = (kchunk + 63) / 64 * 64;
  sp = kchunk > 0 ? (k + kchunk - 1) / kchunk : 1;
  splits = (int)(sp < 1 ? 1 : sp);
}
static int64_t generic_slab_floats(int64_t m, int64_t n, int64_t k) {
  int sp;
  int64_t kc;
  generic_splitk(m, n, k, sp, kc);
  return sp > 1 ? (int64_t)sp * m * n : 0;
}

static bool fp8_ok(int64_t br, int64_t b, int64_t dx, int64_t dy, int precision, bool has_w) {
  return precision == MI_PREC_FP8 && has_w && br % 8 == 0 && b % 8 == 0 && dx % 16 == 0 && dy % 16 == 0;
}
static bool fast_ok(int64_t br, int64_t b, int64_t dx, int64_t dy, int precision, bool has_w) {
  return (precision == MI_PREC_BF16 || precision == MI_PREC_BF16X3) && has_w && br % 8 == 0 && b % 8 == 0 &&
         dx % 8 == 0 && dy % 8 == 0;
}

static BilinearPlan plan_bilinear(Workspace& ws, int64_t br, int64_t b, int64_t dx, int64_t dy, int precision) {
  BilinearPlan p{};
  p.n_partials = ((b + kTile - 1) / kTile) * ((br + kTile - 1) / kTile);
  p.partials = ws.take<Partial>(p.n_partials);
  // forward buffers of the fast path come first so that forward-only callers can pass a smaller workspace
  const int64_t x3 = precision == MI_PREC_BF16X3 ? 3 : 1;
  p.x3 = (int)x3;
  p.xb = ws.take<bf16_t>(x3 * br * dx);
  p.xtb = ws.take<bf16_t>(x3 * br * dx);
  p.yb = ws.take<bf16_t>(x3 * b * dy);
  p.ytb = ws.take<bf16_t>(x3 * b * dy);
  p.wb = ws.take<bf16_t>(x3 * dx * dy);
  p.wtb = ws.take<bf16_t>(x3 * dx * dy);
  p.tb = ws.take<bf16_t>(x3 * br * dy);
  p.ttb = ws.take<bf16_t>(x3 * br * dy);
  p.t = ws.take<float>(br * dy);
  if (precision == MI_PREC_FP8) {
    p.qx8 = ws.take<fp8_t>(br * dx);
    p.qy8 = ws.take<fp8_t>(b * dy);
    p.qwt8 = ws.take<fp8_t>(dx * dy);
    p.qt8 = ws.take<fp8_t>(br * dy);
    p.f8sc = ws.take<Fp8Scales>(1);
  }
  p.fl = FlashPlan{};
  if (precision == MI_PREC_BF16 && br % 8 == 0 && b % 8 == 0 && dx % 8 == 0) p.fl = flash_plan(br, b, dy);
  for (int q = 0; q < 2; ++q) {
    p.fl_rec[q] = p.fl.ok ? ws.take<Partial>(p.fl.n_rec[q]) : nullptr;
    p.fl_slab[q] = p.fl.ok ? ws.take<float>((p.fl.slab_bytes[q] + 3) / 4) : nullptr;
    p.fl_dup[q] = p.fl.ok ? ws.take<unsigned char>((br / 32) * (b / 32)) : nullptr;
  }
  p.tfb = p.fl.ok ? ws.take<bf16_t>(br * dy) : nullptr;
  p.yfb = p.fl.ok ? ws.take<bf16_t>(b * dy) : nullptr;
  static const bool no_tail = getenv("MI_NO_TAIL") != nullptr;  // A/B switch: the round-2 backward tail (three launches)
  p.tail = p.fl.ok && !no_tail && flash_tail_ok(br, dx, dy);
  p.wfb = p.tail ? ws.take<bf16_t>(dx * dy) : nullptr;
  p.xtfb = p.tail ? ws.take<bf16_t>(br * dx) : nullptr;
  p.dttfb = p.tail ? ws.take<bf16_t>(br * dy) : nullptr;
  // backward
  p.dt = ws.take<float>(br * dy);
  if (precision == MI_PREC_F32) p.g = ws.take<float>(br * b);
  else p.g = ws.take<bf16_t>(x3 * br * b);
  p.gb = (bf16_t*)p.g;
  p.gtb = ws.take<bf16_t>(x3 * br * b);
  p.dtb = ws.take<bf16_t>(x3 * br * dy);
  p.dttb = ws.take<bf16_t>(x3 * br * dy);
  int64_t tiles = ((dx + kTile - 1) / kTile) * ((dy + kTile - 1) / kTile);
  // split-K so that dW brings ~128 workgroups to the launch it shares with dX (measured at B = 4096, d = 512: 8 splits
  // 16.7 us, 16 splits 19.0 us, 4 splits 19.8 us for the pair)
  // (the fp8 mode launches dW on its own: 256 workgroups there -- 44 -> 2x fewer idle CUs at d = 1024)
  const int64_t dw_target = precision == MI_PREC_FP8 ? 256 : 128;
  int64_t splits = (dw_target + tiles - 1) / tiles;
  if (const char* e = getenv("MI_DW_SPLITS")) splits = atoi(e) > 0 ? atoi(e) : splits;  // A/B switch
  int64_t kchunk = (br + splits - 1) / splits;
  kchunk = (kchunk + kG2KT - 1) / kG2KT * kG2KT;
  splits = (br + kchunk - 1) / kchunk;
  p.dw_splits = (int)splits;
  p.dw_kchunk = kchunk;
  p.dw_slab = ws.take<float>(splits * dx * dy);
  {
    const int64_t t0 = ((br + kTile - 1) / kTile) * ((dy + kTile - 1) / kTile);
    int64_t sp = br < b ? (128 + t0 - 1) / t0 : 1;
    if (sp > 16) sp = 16;
    int64_t kc = (b + sp - 1) / sp;
    kc = (kc + kG2KT - 1) / kG2KT * kG2KT;
    sp = (b + kc - 1) / kc;
    p.dt_splits = (int)sp;
    p.dt_kchunk = kc;
    p.dt_slab = sp > 1 ? ws.take<float>(sp * br * dy) : nullptr;
  }
  if (precision == MI_PREC_F32) {
    int64_t need = generic_slab_floats(br, dy, dx);                      // T = X W
    const int64_t cand[3] = {generic_slab_floats(br, dy, b),             // dT = G Y
                             generic_slab_floats(b, dy, br),             // dY = G^T T
                             generic_slab_floats(dx, dy, br)};           // dW = X^T dT
    for (int64_t c : cand) need = c > need ? c : need;
    const int64_t c4 = generic_slab_floats(br, dx, dy);                  // dX = dT W^T
    need = c4 > need ? c4 : need;
    p.gen_slab_floats = need;
    p.gen_slab = need > 0 ? ws.take<float>(need) : nullptr;
  }
  p.bytes = ws.off;
  return p;
}

static GemmBf16Args one_problem(const bf16_t* a, int64_t lda, const bf16_t* b, int64_t ldb, int64_t m, int64_t n,
                                int64_t k, int64_t k_chunk = 0) {
  GemmBf16Args g{};
  g.p[0] = GemmBf16Problem{a, lda, b, ldb, m, n, k};
  g.p[1] = g.p[0];
  g.n_problems = 1;
  g.k_chunk = k_chunk > 0 ? k_chunk : k;
  return g;
}

// ------------------------------------------------------------------------------------------------ fast path
static int fast_prep_and_t(const float* x, const float* y, const float* w, const int64_t* sid_rows,
                           const int64_t* sid_cols, int64_t br, int64_t b, int64_t row_offset, int64_t dx, int64_t dy,
                           const BilinearPlan& p, hipStream_t st, int part = 0) {
  // part 1 / part 2: a sharded run prepares what depends on its OWN rows only (the conversions of X and W, T = X W)
  // while the all-gather of the text embeddings is in flight (mi_bilinear_prep_local), and the rest -- the conversions of
  // the gathered Y, the equal-id flags -- behind it.  Fused-kernel shapes only (the callers check p.fl.ok).
  if (part == 2) {
    CvtJobs rest{};
    rest.j[1] = CvtJob{y, b, dy, p.yb, nullptr, 0, 0, p.yfb};
    rest.dup = DupFlagJob{sid_rows, sid_cols, (int)(br / 32), (int)(b / 32), p.fl_dup[0], p.fl_dup[1], row_offset};
    return launch_cvt_transpose3(rest, st, "bilinear prep Y, id flags");
  }
  // With the fused B x B kernel nobody reads Y^T or T^T any more; instead T and Y get a fragment-major copy (the
  // kernel's stationary operand, loaded straight into MFMA B fragments) and the equal-id tile flags ride along.
  if (p.fl.ok) {
    // one launch: T tiles straight from the fp32 operands, the other conversions on the CUs the tiles leave free
    CvtJobs side{};
    // (with the two-launch tail nobody reads X^T row-major any more: only its fragment-major form)
    side.j[0] = CvtJob{x, br, dx, nullptr, p.tail ? nullptr : p.xtb, 0, 0, nullptr, 0, 0, p.xtfb};
    if (part == 0) side.j[1] = CvtJob{y, b, dy, p.yb, nullptr, 0, 0, p.yfb};
    side.j[2] = CvtJob{w, dx, dy, p.wb, nullptr, 0, 0, p.wfb};
    if (part == 0) side.dup = DupFlagJob{sid_rows, sid_cols, (int)(br / 32), (int)(b / 32), p.fl_dup[0], p.fl_dup[1], row_offset};
    const int rc1 = launch_prep_t(x, w, br, dy, dx, p.tb, p.tfb, side, st, "bilinear prep + T = X W");
    if (rc1 != MI_EINVAL) return rc1;
  }
  const int x3 = p.x3;
  // G-materialising path in plain bf16 (widths outside the fused kernel, e.g. the reference's 768): the same single
  // launch -- T tiles (row-major and transposed) beside the conversions.  MI_NO_PREP_T=1: the two launches below.
  if (!p.fl.ok && x3 == 1 && part == 0) {
    CvtJobs side{};
    side.j[0] = CvtJob{x, br, dx, nullptr, p.xtb, 0, 0, nullptr, 0, 0, p.xtfb};
    side.j[1] = CvtJob{y, b, dy, p.yb, p.ytb};
    side.j[2] = CvtJob{w, dx, dy, p.wb, p.wtb, 0, 0, p.wfb};
    const int rc1 = launch_prep_t(x, w, br, dy, dx, p.tb, nullptr, side, st, "bilinear prep + T = X W", false, nullptr, p.ttb);
    if (rc1 != MI_EINVAL) return rc1;
  }
  const int ra = x3 == 3 ? 1 : 0, rb = x3 == 3 ? 2 : 0;  // bf16x3 roles of A-side and B-side operands
  CvtJobs jobs{};
  jobs.j[0] = CvtJob{x, br, dx, p.xb, p.xtb, 0, 0, nullptr, ra, ra, p.xtfb};
  if (part == 0) jobs.j[1] = CvtJob{y, b, dy, p.yb, p.fl.ok ? nullptr : p.ytb, 0, 0, p.fl.ok ? p.yfb : nullptr, rb, rb};
  jobs.j[2] = CvtJob{w, dx, dy, p.wb, p.wtb, 0, 0, p.wfb, rb, rb};
  if (p.fl.ok && part == 0) jobs.dup = DupFlagJob{sid_rows, sid_cols, (int)(br / 32), (int)(b / 32), p.fl_dup[0], p.fl_dup[1], row_offset};
  int rc = launch_cvt_transpose3(jobs, st, "bilinear prep X Y W");
  if (rc) return rc;
  // T[i, c] = sum_a X[i, a] W[a, c]: A = Xb [br][dx], B = W^T [dy][dx]
  EpiStoreMulti e{};
  e.out[0] = EpiOut{nullptr, 0, 0, p.tb, dy, p.fl.ok ? nullptr : p.ttb, br, p.fl.ok ? p.tfb : nullptr, ra, rb};
  return launch_gemm_bf16(one_problem(p.xb, x3 * dx, p.wtb, x3 * dx, br, dy, x3 * dx), 1, e, st, "bilinear T = X W");
}

// The bf16 boundary: x [br][dx] and y [b][dy] arrive as bf16 (encoders under autocast).  Same launch as above minus the
// conversions: T tiles from the bf16 rows as they lie, X^T / Y fragment-major copies, W copies, the equal-id flags; the
// row-major bf16 copy of Y IS the input.
static int fast_prep_and_t_bf16(const bf16_t* x, const bf16_t* y, const float* w, const int64_t* sid_rows,
                                const int64_t* sid_cols, int64_t br, int64_t b, int64_t row_offset, int64_t dx, int64_t dy,
                                const BilinearPlan& p, hipStream_t st) {
  CvtJobs side{};
  side.j[0] = CvtJob{reinterpret_cast<const float*>(x), br, dx, nullptr, nullptr, 0, 0, nullptr, 0, 0, p.xtfb, 1};
  side.j[1] = CvtJob{reinterpret_cast<const float*>(y), b, dy, nullptr, nullptr, 0, 0, p.yfb, 0, 0, nullptr, 1};
  side.j[2] = CvtJob{w, dx, dy, p.wb, nullptr, 0, 0, p.wfb};
  side.dup = DupFlagJob{sid_rows, sid_cols, (int)(br / 32), (int)(b / 32), p.fl_dup[0], p.fl_dup[1], row_offset};
  return launch_prep_t(reinterpret_cast<const float*>(x), w, br, dy, dx, p.tb, p.tfb, side, st,
                       "bilinear prep + T = X W (bf16 in)", true);
}

static int bilinear_bwd_small(int64_t br, int64_t dx, int64_t dy, float* grad_x, float* grad_w, const BilinearPlan& p,
                              hipStream_t st);

// the fused B x B launch: scores, masked log-sum-exp partials and (grad) the unnormalised sums U, V of both gradient
// contractions.  Problem 0 sweeps the text rows for every local image row, problem 1 the local image rows for every
// text row.
static int flash_stage(const int64_t* sid_rows, const int64_t* sid_cols, int64_t br, int64_t b, int64_t row_offset,
                       int64_t dy, bool grad, const BilinearPlan& p, hipStream_t st) {
  FlashArgs a{};
  a.p[0] = FlashProblem{p.tfb, p.yb, sid_rows, sid_cols, br, b, row_offset, p.fl.n_rb[0], p.fl.n_split[0],
                        p.fl.tiles_per_split[0], p.fl_dup[0], p.fl_slab[0], p.fl_rec[0]};
  a.p[1] = FlashProblem{p.yfb, p.tb, sid_cols, sid_rows, b, br, -row_offset, p.fl.n_rb[1], p.fl.n_split[1],
                        p.fl.tiles_per_split[1], p.fl_dup[1], p.fl_slab[1], p.fl_rec[1]};
  a.n_problems = grad ? 2 : 1;
  a.slab_f16 = p.fl.slab_f16 ? 1 : 0;
  return launch_flash(a, dy, grad, st, grad ? "bilinear fused S | P Y | P^T T" : "bilinear fused S + LSE");
}

// The backward's tail in two launches (mi_bilinear_tail.h): [partial sums -> dT rows, grad_y; dX = dT W^T; dT^T
// fragment-major] and [dW = X^T dT].  `merge` != null: the first launch also merges the fused kernel's records into the
// statistics and the loss (the one-call step: no finalize launch).
struct TailMerge {
  int estimator;
  float* loss_out;
  mi_stats* stats_out;
  float* partials_out;
  const float* records = nullptr;  // sharded: the raw records gathered from every rank (rank order), else the workspace's
  int64_t n_records = 0, n_pos = 0;
  int grads_bf16 = 0;              // grad_x / grad_y point to bf16 buffers (the bf16 boundary)
};
static int bilinear_tail(int64_t br, int64_t b, int64_t row_offset, int64_t dx, int64_t dy, const mi_stats* stats,
                         const float* grad_out, float* grad_x, float* grad_y, float* grad_w, const BilinearPlan& p,
                         const TailMerge* merge, hipStream_t st) {
  FlashTailArgs ta{};
  ta.j[0] = FlashReduceJob{p.fl_slab[0], p.fl_rec[0], p.fl.n_split[0], p.fl.n_rb[0], br, p.yb, b, row_offset,
                           nullptr, nullptr, nullptr};
  const bool gbf = merge && merge->grads_bf16;
  ta.j[1] = FlashReduceJob{p.fl_slab[1], p.fl_rec[1], p.fl.n_split[1], p.fl.n_rb[1], b, p.tb, br, -row_offset,
                           gbf ? nullptr : grad_y, gbf ? reinterpret_cast<bf16_t*>(grad_y) : nullptr, nullptr};
  ta.stats = stats;
  ta.grad_out = grad_out;
  if (merge) {
    ta.merge_rec = merge->records ? (const Partial*)merge->records : p.fl_rec[0];
    ta.n_merge = merge->records ? merge->n_records : p.fl.n_rec[0];
    ta.n_pos = merge->records ? merge->n_pos : b;
    ta.estimator = merge->estimator;
    ta.loss_out = merge->loss_out;
    ta.stats_out = merge->stats_out;
    ta.partials_out = merge->partials_out;
  }
  ta.w_frag = p.wfb;
  ta.dx = dx;
#ifdef MI_STAMPS  // diagnostic library only (make STAMPS=1): a timing experiment that leaves grad_x / grad_w wrong
  static const bool diag_no_dx = getenv("MI_TAIL_DIAG_NO_DX") != nullptr;
  ta.grad_x = diag_no_dx ? nullptr : grad_x;
#else
  ta.grad_x = grad_x;
#endif
  if (gbf) {
    ta.grad_x_bf = reinterpret_cast<bf16_t*>(grad_x);
    ta.grad_x = nullptr;
  }
  ta.dtt_frag = p.dttfb;
  int rc = launch_flash_tail(ta, dy, p.fl.slab_f16, merge != nullptr, st,
                             merge ? "bilinear sums -> loss, dT, dY | dX = dT W^T" : "bilinear sums -> dT, dY | dX = dT W^T");
  if (rc) return rc;
  if (!grad_w) return MI_OK;  // the caller launches dW itself (mi_bilinear_bwd_dw), e.g. behind the start of a reduce-scatter
  return launch_bilinear_dw(DwArgs{p.xtfb, p.dttfb, dx, dy, br, grad_w}, st, "bilinear dW = X^T dT");
}

static int bilinear_fwd_fast(const float* x, const float* y, const float* w, const int64_t* sid_rows,
                             const int64_t* sid_cols, int64_t br, int64_t b, int64_t row_offset, int64_t dx, int64_t dy,
                             int estimator, int need_grad, float* loss_out, mi_stats* stats, float* partials_out,
                             float* scores_out, const BilinearPlan& p, hipStream_t st) {
  // need_grad bit 2: mi_bilinear_prep_local already made the part of the preparation that needs neither Y nor the column ids
  int rc = fast_prep_and_t(x, y, w, sid_rows, sid_cols, br, b, row_offset, dx, dy, p, st, (need_grad & 4) && p.fl.ok ? 2 : 0);
  if (rc) return rc;
  if (p.fl.ok) {
    // bit 0: gradients wanted (the fused launch then accumulates both contractions); bits 1-3 are flags (fp8 staged,
    // local part prepared, raw records)
    rc = flash_stage(sid_rows, sid_cols, br, b, row_offset, dy, (need_grad & 1) != 0, p, st);
    if (rc) return rc;
    if (scores_out) {  // per-pair scores are a diagnostic output: the stand-alone score GEMM writes them
      rc = launch_gemm_bf16(one_problem(p.tb, dy, p.yb, dy, br, b, dy), 1,
                            EpiScoreLse2{sid_rows, sid_cols, row_offset, scores_out, p.partials}, st,
                            "bilinear score+LSE");
      if (rc) return rc;
    }
    // need_grad bit 3: the caller gathers the RAW per-wave records (mi_bilinear_raw_records) from every rank and hands
    // them to mi_bilinear_bwd_records, whose first launch merges them: no finalize launch, nothing written to loss / stats
    if ((need_grad & 8) && p.tail) return MI_OK;
    return launch_finalize(p.fl_rec[0], p.fl.n_rec[0], b, estimator, loss_out, stats, partials_out, st);
  }
  const int x3 = p.x3;
  rc = launch_gemm_bf16(one_problem(p.tb, x3 * dy, p.yb, x3 * dy, br, b, x3 * dy), 1,
                        EpiScoreLse2{sid_rows, sid_cols, row_offset, scores_out, p.partials}, st, "bilinear score+LSE");
  if (rc) return rc;
  // the launcher picks 256 x 256 tiles for large score matrices: fewer partials than the plan reserved
  return launch_finalize(p.partials, gemm_bf16_n_partials(br, b, x3 * dy), b, estimator, loss_out, stats, partials_out,
                         st);
}

static int bilinear_bwd_fast(const int64_t* sid_rows, const int64_t* sid_cols, int64_t br, int64_t b, int64_t row_offset,
                             int64_t dx, int64_t dy, const mi_stats* stats, const float* grad_out, float* grad_x,
                             float* grad_y, float* grad_w, const BilinearPlan& p, bool flash_sums, hipStream_t st) {
  int rc = MI_OK;
  if (flash_sums && p.tail) return bilinear_tail(br, b, row_offset, dx, dy, stats, grad_out, grad_x, grad_y, grad_w, p, nullptr, st);
  if (flash_sums) {
    // the forward's fused launch left U, V as slabs: scale by exp(m_ref - lse), subtract the diagonal term, add the
    // slabs in a fixed order -> dT (bf16, both orientations, operands of dW | dX) and grad_y
    FlashReduceArgs ra{};
    ra.j[0] = FlashReduceJob{p.fl_slab[0], p.fl_rec[0], p.fl.n_split[0], p.fl.n_rb[0], br, p.yb, b, row_offset,
                             nullptr, p.dtb, p.dttb};
    ra.j[1] = FlashReduceJob{p.fl_slab[1], p.fl_rec[1], p.fl.n_split[1], p.fl.n_rb[1], b, p.tb, br, -row_offset,
                             grad_y, nullptr, nullptr};
    ra.stats = stats;
    ra.grad_out = grad_out;
    rc = launch_flash_reduce(ra, 2, dy, p.fl.slab_f16, st, "bilinear dT, dY from the fused sums");
    if (rc) return rc;
    return bilinear_bwd_small(br, dx, dy, grad_x, grad_w, p, st);
  }
  // G and G^T (bf16) from recomputed score tiles
  const int x3 = p.x3;
  const int ra = x3 == 3 ? 1 : 0, rbk = x3 == 3 ? 2 : 0;
  rc = launch_gemm_bf16(one_problem(p.tb, x3 * dy, p.yb, x3 * dy, br, b, x3 * dy), 1,
                            EpiGradScore2{sid_rows, sid_cols, row_offset, stats, grad_out, p.gb, p.gtb, x3 == 3 ? 1 : 0}, st,
                            "bilinear G");
  if (rc) return rc;
  // problem 0: dT[i, c] = sum_j G[i, j] Y[j, c]   (A = G [br][b], B = Y^T [dy][b])  -> bf16 both orientations
  // problem 1: dY[j, c] = sum_i G[i, j] T[i, c]   (A = G^T [b][br], B = T^T [dy][br]) -> fp32 grad_y
  GemmBf16Args two{};
  two.p[0] = GemmBf16Problem{p.gb, x3 * b, p.ytb, x3 * b, br, dy, x3 * b};
  two.p[1] = GemmBf16Problem{p.gtb, x3 * br, p.ttb, x3 * br, b, dy, x3 * br};
  two.n_problems = 2;
  two.k_chunk = x3 * (b > br ? b : br);
  EpiStoreMulti e2{};
  e2.out[0] = EpiOut{nullptr, 0, 0, p.dtb, dy, p.dttb, br, nullptr, ra, rbk};
  e2.out[1] = EpiOut{grad_y, dy, 0, nullptr, 0, nullptr, 0};
  bool split_done = false;
  if (p.dt_splits > 1 && x3 == 1) {
    // sharded row block: dT partial sums over K chunks into fp32 slabs, beside dY; then one pass that adds the slabs and
    // writes the two bf16 orientations of dT
    EpiStoreMulti es{};
    es.out[0] = EpiOut{p.dt_slab, dy, br * dy, nullptr, 0, nullptr, 0};
    es.out[1] = e2.out[1];
    const int sp[2] = {p.dt_splits, 1};
    const int64_t ch[2] = {p.dt_kchunk, br};
    rc = launch_gemm_bf16_flat(two, sp, ch, es, st, "bilinear dT = G Y | dY = G^T T");
    if (rc == MI_OK) {
      CvtJobs jobs{};
      jobs.j[0] = CvtJob{p.dt_slab, br, dy, p.dtb, p.dttb, p.dt_splits, br * dy};
      rc = launch_cvt_transpose3(jobs, st, "bilinear dT slabs -> bf16");
      if (rc) return rc;
      split_done = true;
    } else if (rc != MI_EINVAL) {
      return rc;
    }
  }
  if (!split_done) {
    rc = launch_gemm_bf16(two, 1, e2, st, "bilinear dT = G Y | dY = G^T T");
    if (rc) return rc;
  }
  return bilinear_bwd_small(br, dx, dy, grad_x, grad_w, p, st);
}

// dW[a, c] = sum_i X[i, a] dT[i, c]: A = X^T [dx][br], B = dT^T [dy][br], split over i into slabs
// dX[i, a] = sum_c dT[i, c] W[a, c]: A = dT [br][dy], B = W [dx][dy]
// Both only wait for dT: one launch (few tiles each; on their own they leave most CUs idle).
static int bilinear_bwd_small(int64_t br, int64_t dx, int64_t dy, float* grad_x, float* grad_w, const BilinearPlan& p,
                              hipStream_t st) {
  int rc = MI_OK;
  GemmBf16Args dwx{};
  const int x3 = p.x3;
  dwx.p[0] = GemmBf16Problem{p.xtb, x3 * br, p.dttb, x3 * br, dx, dy, x3 * br};
  dwx.p[1] = GemmBf16Problem{p.dtb, x3 * dy, p.wb, x3 * dy, br, dx, x3 * dy};
  dwx.n_problems = 2;
  EpiStoreMulti e3{};
  e3.out[0] = EpiOut{p.dw_slab, dy, dx * dy, nullptr, 0, nullptr, 0};
  e3.out[1] = EpiOut{grad_x, dx, 0, nullptr, 0, nullptr, 0};
  const int splits[2] = {p.dw_splits, 1};
  const int64_t chunks[2] = {x3 * p.dw_kchunk, x3 * dy};
  rc = launch_gemm_bf16_flat(dwx, splits, chunks, e3, st, "bilinear dW = X^T dT | dX = dT W^T");
  if (rc == MI_EINVAL) {  // shapes the LDS-DMA kernel does not take: one launch each
    EpiStoreMulti e4{};
    e4.out[0] = e3.out[0];
    rc = launch_gemm_bf16(one_problem(p.xtb, x3 * br, p.dttb, x3 * br, dx, dy, x3 * br, x3 * p.dw_kchunk), p.dw_splits,
                          e4, st, "bilinear dW = X^T dT");
    if (rc) return rc;
    e4.out[0] = e3.out[1];
    rc = launch_gemm_bf16(one_problem(p.dtb, x3 * dy, p.wb, x3 * dy, br, dx, x3 * dy), 1, e4, st,
                          "bilinear dX = dT W^T");
  }
  if (rc) return rc;
  return launch_slab_reduce_ld(p.dw_slab, p.dw_splits, dx, dy, grad_w, dy, st, "slab_reduce_ld_kernel");
}

// ------------------------------------------------------------------------------------------------ fp8 path (mi_fp8.h)
// absmax -> scales -> e4m3 operands; T on the fp8 MFMA with its absmax in the epilogue; T quantised.  Three stages, so
// that a sharded run can make the scales GLOBAL between them (mi_bilinear_fp8_stage): the absmax of the local image rows
// and of the local rows of T differ between ranks; a MAX all-reduce of the four numbers makes every rank quantise with
// the scales a single GPU would have used.
static int fp8_stage0(const float* x, const float* y, const float* w, int64_t br, int64_t b, int64_t dx, int64_t dy,
                      const BilinearPlan& p, hipStream_t st) {
  const int rc0 = launch_zero_words(p.f8sc, sizeof(Fp8Scales), st, "zero_words_kernel(fp8 scales)");
  if (rc0) return rc0;
  AbsmaxJobs aj{};
  aj.in[0] = x; aj.n[0] = br * dx; aj.slot[0] = 0;
  aj.in[1] = y; aj.n[1] = b * dy; aj.slot[1] = 1;
  aj.in[2] = w; aj.n[2] = dx * dy; aj.slot[2] = 2;
  aj.sc = p.f8sc;
  {
    ProfScope prof_("fp8 absmax X Y W", st);
    hipLaunchKernelGGL(fp8_absmax_kernel, dim3(kAbsmaxBlocks, 3), dim3(256), 0, st, aj);
  }
  MI_LAUNCH_CHECK("fp8_absmax_kernel");
  return MI_OK;
}
static int fp8_stage1(const float* x, const float* y, const float* w, int64_t br, int64_t b, int64_t dx, int64_t dy,
                      const BilinearPlan& p, hipStream_t st) {
  QuantJobs qj{};
  qj.j[0] = QuantJob{x, br, dx, 0, p.qx8, nullptr, nullptr, p.xtb};   // A of T = X W; X^T (bf16) for dW
  qj.j[1] = QuantJob{y, b, dy, 1, p.qy8, nullptr, nullptr, p.ytb};    // B of S = T Y^T; Y^T (bf16) for dT
  qj.j[2] = QuantJob{w, dx, dy, 2, nullptr, p.qwt8, p.wb, nullptr};   // B of T = X W is W^T; W (bf16) for dX
  qj.sc = p.f8sc;
  int64_t rmax = br > b ? br : b, cmax = dx > dy ? dx : dy;
  if (dx > rmax) rmax = dx;
  {
    ProfScope prof_("fp8 quantize X Y W", st);
    hipLaunchKernelGGL(fp8_quantize_kernel, dim3((unsigned)((cmax + 63) / 64), (unsigned)((rmax + 63) / 64), 3), dim3(256),
                       0, st, qj);
  }
  MI_LAUNCH_CHECK("fp8_quantize_kernel");
  return launch_gemm_fp8(GemmF8Args{p.qx8, dx, p.qwt8, dx, br, dy, dx}, EpiT8{p.t, p.f8sc}, st, "fp8 T = X W");
}
static int fp8_stage2(int64_t br, int64_t dy, const BilinearPlan& p, hipStream_t st) {
  QuantJobs qt{};
  qt.j[0] = QuantJob{p.t, br, dy, 3, p.qt8, nullptr, nullptr, p.ttb};  // A of S; T^T (bf16) for dY
  qt.sc = p.f8sc;
  {
    ProfScope prof_("fp8 quantize T", st);
    hipLaunchKernelGGL(fp8_quantize_kernel, dim3((unsigned)((dy + 63) / 64), (unsigned)((br + 63) / 64), 1), dim3(256), 0, st,
                       qt);
  }
  MI_LAUNCH_CHECK("fp8_quantize_kernel");
  return MI_OK;
}
static int fp8_prep_and_t(const float* x, const float* y, const float* w, int64_t br, int64_t b, int64_t dx, int64_t dy,
                          const BilinearPlan& p, hipStream_t st) {
  int rc = fp8_stage0(x, y, w, br, b, dx, dy, p, st);
  if (rc) return rc;
  rc = fp8_stage1(x, y, w, br, b, dx, dy, p, st);
  if (rc) return rc;
  return fp8_stage2(br, dy, p, st);
}

static int bilinear_fwd_fp8(const float* x, const float* y, const float* w, const int64_t* sid_rows, const int64_t* sid_cols,
                            int64_t br, int64_t b, int64_t row_offset, int64_t dx, int64_t dy, int estimator,
                            float* loss_out, mi_stats* stats, float* partials_out, float* scores_out,
                            const BilinearPlan& p, bool staged, hipStream_t st) {
  int rc = staged ? MI_OK : fp8_prep_and_t(x, y, w, br, b, dx, dy, p, st);  // staged: mi_bilinear_fp8_stage made them
  if (rc) return rc;
  EpiScaled<EpiScoreLse2> e{EpiScoreLse2{sid_rows, sid_cols, row_offset, scores_out, p.partials},
                            {&p.f8sc->scale[3], nullptr}, {&p.f8sc->scale[1], nullptr}, p.partials};
  rc = launch_gemm_fp8(GemmF8Args{p.qt8, dy, p.qy8, dy, br, b, dy}, e, st, "fp8 score+LSE");
  if (rc) return rc;
  return launch_finalize(p.partials, gemm_fp8_n_partials(br, b, dy), b, estimator, loss_out, stats, partials_out, st);
}

static int bilinear_bwd_fp8(const int64_t* sid_rows, const int64_t* sid_cols, int64_t br, int64_t b, int64_t row_offset,
                            int64_t dx, int64_t dy, const mi_stats* stats, const float* grad_out, float* grad_x,
                            float* grad_y, float* grad_w, const BilinearPlan& p, hipStream_t st) {
  const float* sc = p.f8sc->scale;  // device addresses: x, y, w, T
  // G and G^T (bf16) from recomputed fp8 score tiles
  EpiScaled<EpiGradScore2> eg{EpiGradScore2{sid_rows, sid_cols, row_offset, stats, grad_out, p.gb, p.gtb, 0},
                              {sc + 3, nullptr}, {sc + 1, nullptr}};
  int rc = launch_gemm_fp8(GemmF8Args{p.qt8, dy, p.qy8, dy, br, b, dy}, eg, st, "fp8 G");
  if (rc) return rc;
  // dT = s_y (G y_q) -> bf16 both orientations | dY = s_t (G^T t_q) -> grad_y; operands: the quantised values as bf16
  GemmBf16Args two{};
  two.p[0] = GemmBf16Problem{p.gb, b, p.ytb, b, br, dy, b};
  two.p[1] = GemmBf16Problem{p.gtb, br, p.ttb, br, b, dy, br};
  two.n_problems = 2;
  two.k_chunk = b > br ? b : br;
  EpiScaled<EpiStoreMulti> e2{};
  e2.inner.out[0] = EpiOut{nullptr, 0, 0, p.dtb, dy, p.dttb, br};
  e2.inner.out[1] = EpiOut{grad_y, dy, 0, nullptr, 0, nullptr, 0};
  e2.sa[0] = sc + 1;
  e2.sa[1] = sc + 3;
  rc = launch_gemm_bf16(two, 1, e2, st, "fp8 mode dT = G Y | dY = G^T T");
  if (rc) return rc;
  // dW = s_x (x_q^T dT) (split-K slabs) | dX = s_w (dT W_q^T)
  EpiScaled<EpiStoreMulti> e3{};
  e3.inner.out[0] = EpiOut{p.dw_slab, dy, dx * dy, nullptr, 0, nullptr, 0};
  e3.sa[0] = sc + 0;
  rc = launch_gemm_bf16(one_problem(p.xtb, br, p.dttb, br, dx, dy, br, p.dw_kchunk), p.dw_splits, e3, st,
                        "fp8 mode dW = X^T dT");
  if (rc) return rc;
  EpiScaled<EpiStoreMulti> e4{};
  e4.inner.out[0] = EpiOut{grad_x, dx, 0, nullptr, 0, nullptr, 0};
  e4.sa[0] = sc + 2;
  rc = launch_gemm_bf16(one_problem(p.dtb, dy, p.wb, dy, br, dx, dy), 1, e4, st, "fp8 mode dX = dT W^T");
  if (rc) return rc;
  return launch_slab_reduce_ld(p.dw_slab, p.dw_splits, dx, dy, grad_w, dy, st, "slab_reduce_ld_kernel");
}

// ------------------------------------------------------------------------------------------------ generic path
// out [M][N] (ld) = A B^T with K split into slabs when the output has too few tiles (exact-fp32 mode only: the plan holds
// the slab buffer); the slabs are added in split order (fixed order: reproducible)
template <typename OpT, typename TA, typename TB>
static int generic_gemm_store(const Operand<TA>& A, const Operand<TB>& B, int64_t M, int64_t N, int64_t K, float* out,
                              int64_t ld, const BilinearPlan& p, hipStream_t st, const char* what) {
  int sp = 1;
  int64_t kc = K;
  static const bool off = getenv("MI_GENERIC_NO_SPLITK") != nullptr;  // A/B switch
  if (p.gen_slab && !off) generic_splitk(M, N, K, sp, kc);
  if (sp <= 1 || (int64_t)sp * M * N > p.gen_slab_floats)
    return launch_gemm<OpT>(A, B, M, N, K, EpiStore{out, ld, nullptr, 1.0f, 0}, st, what);
  EpiStore e{p.gen_slab, N, nullptr, 1.0f, 0};
  e.slab_stride = M * N;
  const int rc = launch_gemm<OpT>(A, B, M, N, K, e, st, what, sp, kc);
  if (rc) return rc;
  return launch_slab_reduce_ld(p.gen_slab, sp, M, N, out, ld, st, "slab_reduce_ld_kernel");
}

template <typename OpT>
static int bilinear_fwd_impl(const float* x, const float* y, const float* w, const int64_t* sid_rows,
                             const int64_t* sid_cols, int64_t br, int64_t b, int64_t row_offset, int64_t dx, int64_t dy,
                             int estimator, float* loss_out, mi_stats* stats, float* partials_out, float* scores_out,
                             const BilinearPlan& p, hipStream_t st) {
  int rc = MI_OK;
  const float* t = x;  // w == nullptr: separable form, the caller passes the projected embeddings (d_img == d_txt)
  if (w) {
    rc = generic_gemm_store<OpT>(make_operand(x, dx, 1), make_operand(w, 1, dy), br, dy, dx, p.t, dy, p, st,
                                 "bilinear T = X W (generic)");
    if (rc) return rc;
    t = p.t;
  }
  rc = launch_gemm<OpT>(make_operand(t, dy, 1), make_operand(y, dy, 1), br, b, dy,
                        EpiScoreLse{sid_rows, sid_cols, row_offset, scores_out, p.partials}, st,
                        "bilinear score+LSE (generic)");
  if (rc) return rc;
  return launch_finalize(p.partials, p.n_partials, b, estimator, loss_out, stats, partials_out, st);
}

template <typename OpT, typename TG>
static int bilinear_bwd_impl(const float* x, const float* y, const float* w, const int64_t* sid_rows,
                             const int64_t* sid_cols, int64_t br, int64_t b, int64_t row_offset, int64_t dx, int64_t dy,
                             const mi_stats* stats, const float* grad_out, float* grad_x, float* grad_y, float* grad_w,
                             const BilinearPlan& p, hipStream_t st) {
  TG* g = (TG*)p.g;
  int rc = MI_OK;
  const float* t = x;
  float* dt = grad_x;  // w == nullptr: dT is dX
  if (w) {
    rc = generic_gemm_store<OpT>(make_operand(x, dx, 1), make_operand(w, 1, dy), br, dy, dx, p.t, dy, p, st,
                                 "bilinear T = X W (generic, bwd)");
    if (rc) return rc;
    t = p.t;
    dt = p.dt;
  }
  rc = launch_gemm<OpT>(make_operand(t, dy, 1), make_operand(y, dy, 1), br, b, dy,
                        EpiGradScore<TG>{sid_rows, sid_cols, row_offset, stats, grad_out, g}, st, "bilinear G (generic)");
  if (rc) return rc;
  rc = generic_gemm_store<OpT>(make_operand((const TG*)g, b, 1), make_operand(y, 1, dy), br, dy, b, dt, dy, p, st,
                               "bilinear dT = G Y (generic)");
  if (rc) return rc;
  rc = generic_gemm_store<OpT>(make_operand((const TG*)g, 1, b), make_operand(t, 1, dy), b, dy, br, grad_y, dy, p, st,
                               "bilinear dY = G^T T (generic)");
  if (rc) return rc;
  if (!w) return MI_OK;
  rc = generic_gemm_store<OpT>(make_operand(x, 1, dx), make_operand((const float*)p.dt, 1, dy), dx, dy, br, grad_w, dy, p, st,
                               "bilinear dW = X^T dT (generic)");
  if (rc) return rc;
  return generic_gemm_store<OpT>(make_operand((const float*)p.dt, dy, 1), make_operand(w, dy, 1), br, dx, dy, grad_x, dx, p,
                                 st, "bilinear dX = dT W^T (generic)");
}

static int check_common(const char* fn, int64_t br, int64_t b, int64_t row_offset, int64_t dx, int64_t dy,
                        int precision) {
  MI_CHECK_ARG(br >= 1 && b >= 1 && br <= b, "%s: need 1 <= b_rows <= b (got %lld, %lld)", fn, (long long)br,
               (long long)b);
  MI_CHECK_ARG(row_offset >= 0 && row_offset + br <= b, "%s: row block [%lld, %lld) outside [0, %lld)", fn,
               (long long)row_offset, (long long)(row_offset + br), (long long)b);
  MI_CHECK_ARG(dx >= 1 && dy >= 1, "%s: embedding widths must be >= 1", fn);
  MI_CHECK_ARG(precision == MI_PREC_F32 || precision == MI_PREC_BF16 || precision == MI_PREC_BF16X3 ||
                   precision == MI_PREC_FP8,
               "%s: unknown precision %d", fn, precision);
  return MI_OK;
}

}  // namespace mi

using namespace mi;

extern "C" {

size_t mi_bilinear_workspace_bytes(int64_t b_rows, int64_t b, int64_t d_img, int64_t d_txt, int precision) {
  if (b_rows <= 0 || b <= 0 || d_img <= 0 || d_txt <= 0) return 0;  // (the planner divides by tile counts)
  Workspace ws(nullptr, 0);
  return plan_bilinear(ws, b_rows, b, d_img, d_txt, precision).bytes + 256;
}

int mi_bilinear_fwd(const float* x, const float* y, const float* w, const int64_t* sid_rows, const int64_t* sid_cols,
                    int64_t b_rows, int64_t b, int64_t row_offset, int64_t d_img, int64_t d_txt, int estimator,
                    int precision, int need_grad, float* loss_out, mi_stats* stats, float* partials_out,
                    float* scores_out, void* workspace, size_t workspace_bytes, void* stream) {
  MI_CHECK_ARG(x && y && sid_rows && sid_cols && stats && workspace, "mi_bilinear_fwd: null pointer");
  MI_CHECK_ARG(w || d_img == d_txt, "mi_bilinear_fwd: w == NULL (separable form) needs d_img == d_txt");
  int rc = check_common("mi_bilinear_fwd", b_rows, b, row_offset, d_img, d_txt, precision);
  if (rc) return rc;
  MI_CHECK_ARG(estimator == MI_DV || estimator == MI_INFONCE, "mi_bilinear_fwd: unknown estimator %d", estimator);
  Workspace ws(workspace, workspace_bytes);
  BilinearPlan p = plan_bilinear(ws, b_rows, b, d_img, d_txt, precision);
  if (!ws.ok()) {
    set_error("mi_bilinear_fwd: workspace too small (%zu < %zu)", workspace_bytes, ws.off);
    return MI_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  if (precision == MI_PREC_FP8) {
    if (!fp8_ok(b_rows, b, d_img, d_txt, precision, w != nullptr)) {
      set_error("mi_bilinear_fwd: the fp8 mode needs a weight matrix, batch sizes that are multiples of 8 and widths that "
                "are multiples of 16");
      return MI_ESHAPE;
    }
    return bilinear_fwd_fp8(x, y, w, sid_rows, sid_cols, b_rows, b, row_offset, d_img, d_txt, estimator, loss_out, stats,
                            partials_out, scores_out, p, (need_grad & 2) != 0, st);
  }
  if (fast_ok(b_rows, b, d_img, d_txt, precision, w != nullptr))
    return bilinear_fwd_fast(x, y, w, sid_rows, sid_cols, b_rows, b, row_offset, d_img, d_txt, estimator, need_grad,
                             loss_out, stats, partials_out, scores_out, p, st);
  if (precision == MI_PREC_BF16)
    return bilinear_fwd_impl<bf16_t>(x, y, w, sid_rows, sid_cols, b_rows, b, row_offset, d_img, d_txt, estimator,
                                     loss_out, stats, partials_out, scores_out, p, st);
  return bilinear_fwd_impl<float>(x, y, w, sid_rows, sid_cols, b_rows, b, row_offset, d_img, d_txt, estimator, loss_out,
                                  stats, partials_out, scores_out, p, st);
}

int mi_bilinear_bwd(const float* x, const float* y, const float* w, const int64_t* sid_rows, const int64_t* sid_cols,
                    int64_t b_rows, int64_t b, int64_t row_offset, int64_t d_img, int64_t d_txt, int precision,
                    const mi_stats* stats, const float* grad_out, float* grad_x, float* grad_y, float* grad_w,
                    void* workspace, size_t workspace_bytes, int workspace_from_forward, void* stream) {
  MI_CHECK_ARG(x && y && sid_rows && sid_cols && stats && grad_x && grad_y && workspace,
               "mi_bilinear_bwd: null pointer");
  MI_CHECK_ARG((w && grad_w) || (!w && d_img == d_txt),
               "mi_bilinear_bwd: w == NULL (separable form) needs d_img == d_txt; w != NULL needs grad_w");
  int rc = check_common("mi_bilinear_bwd", b_rows, b, row_offset, d_img, d_txt, precision);
  if (rc) return rc;
  Workspace ws(workspace, workspace_bytes);
  BilinearPlan p = plan_bilinear(ws, b_rows, b, d_img, d_txt, precision);
  if (!ws.ok()) {
    set_error("mi_bilinear_bwd: workspace too small (%zu < %zu)", workspace_bytes, ws.off);
    return MI_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  if (precision == MI_PREC_FP8) {
    if (!fp8_ok(b_rows, b, d_img, d_txt, precision, w != nullptr)) {
      set_error("mi_bilinear_bwd: the fp8 mode needs a weight matrix, batch sizes that are multiples of 8 and widths that "
                "are multiples of 16");
      return MI_ESHAPE;
    }
    if (!workspace_from_forward) {
      rc = fp8_prep_and_t(x, y, w, b_rows, b, d_img, d_txt, p, st);
      if (rc) return rc;
    }
    return bilinear_bwd_fp8(sid_rows, sid_cols, b_rows, b, row_offset, d_img, d_txt, stats, grad_out, grad_x, grad_y, grad_w,
                            p, st);
  }
  if (fast_ok(b_rows, b, d_img, d_txt, precision, w != nullptr)) {
    if (!workspace_from_forward) {  // rebuild the bf16 operand copies, T and the fused sums
      rc = fast_prep_and_t(x, y, w, sid_rows, sid_cols, b_rows, b, row_offset, d_img, d_txt, p, st);
      if (rc) return rc;
      if (p.fl.ok) {
        rc = flash_stage(sid_rows, sid_cols, b_rows, b, row_offset, d_txt, true, p, st);
        if (rc) return rc;
      }
    }
    return bilinear_bwd_fast(sid_rows, sid_cols, b_rows, b, row_offset, d_img, d_txt, stats, grad_out, grad_x, grad_y,
                             grad_w, p, p.fl.ok, st);
  }
  if (precision == MI_PREC_BF16)
    return bilinear_bwd_impl<bf16_t, bf16_t>(x, y, w, sid_rows, sid_cols, b_rows, b, row_offset, d_img, d_txt, stats,
                                             grad_out, grad_x, grad_y, grad_w, p, st);
  return bilinear_bwd_impl<float, float>(x, y, w, sid_rows, sid_cols, b_rows, b, row_offset, d_img, d_txt, stats,
                                         grad_out, grad_x, grad_y, grad_w, p, st);
}

/* Sharded batches without the finalize and merge launches.  The fused kernel leaves one 16-byte record per wave in the
 * workspace; mi_bilinear_raw_records says where (byte offset) and how many.  A rank runs mi_bilinear_fwd(need_grad | 8)
 * (no finalize: loss / stats / partials_out untouched), all-gathers that region from every rank (rank order) and calls
 * mi_bilinear_bwd_records, whose first launch merges ALL records in the gathered order on every workgroup (bit-identical
 * statistics and loss on every rank), writes loss / stats and goes on with the backward's tail.  0 records = the shape
 * does not take this path (use mi_bilinear_fwd / mi_merge_partials / mi_bilinear_bwd). */
size_t mi_bilinear_raw_records(int64_t b_rows, int64_t b, int64_t d_img, int64_t d_txt, int precision,
                               size_t* offset_bytes) {
  if (b_rows <= 0 || b <= 0 || d_img <= 0 || d_txt <= 0) return 0;
  char* fake = reinterpret_cast<char*>(uintptr_t(1) << 20);  // only pointer differences are used
  Workspace ws(fake, ~size_t(0) >> 1);
  BilinearPlan p = plan_bilinear(ws, b_rows, b, d_img, d_txt, precision);
  if (!(fast_ok(b_rows, b, d_img, d_txt, precision, true) && p.fl.ok && p.tail)) return 0;
  if (offset_bytes) *offset_bytes = (size_t)(reinterpret_cast<char*>(p.fl_rec[0]) - fake);
  return (size_t)p.fl.n_rec[0];
}

int mi_bilinear_bwd_records(const float* x, const float* y, const float* w, const int64_t* sid_rows,
                            const int64_t* sid_cols, int64_t b_rows, int64_t b, int64_t row_offset, int64_t d_img,
                            int64_t d_txt, int precision, int estimator, const float* records, int64_t n_records,
                            int64_t n_pos, const float* grad_out, float* loss_out, mi_stats* stats_out, float* grad_x,
                            float* grad_y, float* grad_w, void* workspace, size_t workspace_bytes, void* stream) {
  MI_CHECK_ARG(x && y && w && sid_rows && sid_cols && records && stats_out && grad_x && grad_y && workspace,
               "mi_bilinear_bwd_records: null pointer");
  MI_CHECK_ARG(n_records > 0 && n_pos > 0, "mi_bilinear_bwd_records: n_records and n_pos must be positive");
  MI_CHECK_ARG(estimator == MI_DV || estimator == MI_INFONCE, "mi_bilinear_bwd_records: unknown estimator %d", estimator);
  int rc = check_common("mi_bilinear_bwd_records", b_rows, b, row_offset, d_img, d_txt, precision);
  if (rc) return rc;
  Workspace ws(workspace, workspace_bytes);
  BilinearPlan p = plan_bilinear(ws, b_rows, b, d_img, d_txt, precision);
  if (!ws.ok()) {
    set_error("mi_bilinear_bwd_records: workspace too small (%zu < %zu)", workspace_bytes, ws.off);
    return MI_EWORKSPACE;
  }
  if (!(fast_ok(b_rows, b, d_img, d_txt, precision, true) && p.fl.ok && p.tail)) {
    set_error("mi_bilinear_bwd_records: shape / precision outside the fused kernels (mi_bilinear_raw_records returns 0)");
    return MI_ESHAPE;
  }
  TailMerge tm{estimator, loss_out, stats_out, nullptr, records, n_records, n_pos};
  return bilinear_tail(b_rows, b, row_offset, d_img, d_txt, nullptr, grad_out, grad_x, grad_y, grad_w, p, &tm,
                       (hipStream_t)stream);
}

/* The second launch of the backward's tail on its own (dW = X^T dT from the fragment-major operands the first launch and
 * the forward left in the workspace).  For callers that ran mi_bilinear_bwd_records with grad_w == NULL: the reduce-scatter
 * of grad_y only needs the first launch, so a sharded step starts it (on RCCL's stream) BEFORE this launch and the two
 * overlap. */
int mi_bilinear_bwd_dw(int64_t b_rows, int64_t b, int64_t d_img, int64_t d_txt, int precision, float* grad_w,
                       void* workspace, size_t workspace_bytes, void* stream) {
  MI_CHECK_ARG(grad_w && workspace, "mi_bilinear_bwd_dw: null pointer");
  int rc = check_common("mi_bilinear_bwd_dw", b_rows, b, 0, d_img, d_txt, precision);
  if (rc) return rc;
  Workspace ws(workspace, workspace_bytes);
  BilinearPlan p = plan_bilinear(ws, b_rows, b, d_img, d_txt, precision);
  if (!ws.ok()) {
    set_error("mi_bilinear_bwd_dw: workspace too small (%zu < %zu)", workspace_bytes, ws.off);
    return MI_EWORKSPACE;
  }
  if (!(fast_ok(b_rows, b, d_img, d_txt, precision, true) && p.fl.ok && p.tail)) {
    set_error("mi_bilinear_bwd_dw: shape / precision outside the fused kernels");
    return MI_ESHAPE;
  }
  return launch_bilinear_dw(DwArgs{p.xtfb, p.dttfb, d_img, d_txt, b_rows, grad_w}, (hipStream_t)stream, "bilinear dW = X^T dT");
}

/* Sharded batches: the part of the forward's preparation that depends on the rank's OWN rows only -- the bf16 copies of
 * X and W and T = X W -- so that it can run while the all-gather of the text embeddings is in flight.  Follow with
 * mi_bilinear_fwd(..., need_grad | 4, ...) on the same workspace (bit 2: the local part is prepared).  Returns
 * MI_ESHAPE where the shape does not take the fused kernels (the caller then simply calls mi_bilinear_fwd as usual). */
int mi_bilinear_prep_local(const float* x, const float* w, int64_t b_rows, int64_t b, int64_t d_img, int64_t d_txt,
                           int precision, void* workspace, size_t workspace_bytes, void* stream) {
  MI_CHECK_ARG(x && w && workspace, "mi_bilinear_prep_local: null pointer");
  int rc = check_common("mi_bilinear_prep_local", b_rows, b, 0, d_img, d_txt, precision);
  if (rc) return rc;
  Workspace ws(workspace, workspace_bytes);
  BilinearPlan p = plan_bilinear(ws, b_rows, b, d_img, d_txt, precision);
  if (!ws.ok()) {
    set_error("mi_bilinear_prep_local: workspace too small (%zu < %zu)", workspace_bytes, ws.off);
    return MI_EWORKSPACE;
  }
  if (!(fast_ok(b_rows, b, d_img, d_txt, precision, true) && p.fl.ok)) {
    set_error("mi_bilinear_prep_local: shape / precision outside the fused kernels (call mi_bilinear_fwd alone)");
    return MI_ESHAPE;
  }
  return fast_prep_and_t(x, nullptr, w, nullptr, nullptr, b_rows, b, 0, d_img, d_txt, p, (hipStream_t)stream, 1);
}

/* fp8 mode on a SHARDED batch: the per-tensor scales must be those of the whole batch.  The forward's preparation in
 * three stages around the caller's two MAX all-reduces of amax_io (4 floats, device):
 *   stage 0: local absmax of x (slot 0), y (1), w (2) -> amax_io[0..2]                    | all-reduce MAX amax_io
 *   stage 1: quantise x, y, w with amax_io[0..2]; T = x_q W_q; local absmax of T -> amax_io[3] | all-reduce MAX amax_io
 *   stage 2: quantise T with amax_io[3]
 * then mi_bilinear_fwd(..., need_grad | 2, ...) on the same workspace (bit 1: "the fp8 operands are staged"). */
int mi_bilinear_fp8_stage(const float* x, const float* y, const float* w, int64_t b_rows, int64_t b, int64_t d_img,
                          int64_t d_txt, int stage, float* amax_io, void* workspace, size_t workspace_bytes, void* stream) {
  MI_CHECK_ARG(x && y && w && amax_io && workspace, "mi_bilinear_fp8_stage: null pointer");
  MI_CHECK_ARG(stage >= 0 && stage <= 2, "mi_bilinear_fp8_stage: stage %d outside 0..2", stage);
  int rc = check_common("mi_bilinear_fp8_stage", b_rows, b, 0, d_img, d_txt, MI_PREC_FP8);
  if (rc) return rc;
  if (!fp8_ok(b_rows, b, d_img, d_txt, MI_PREC_FP8, true)) {
    set_error("mi_bilinear_fp8_stage: the fp8 mode needs batch sizes that are multiples of 8 and widths that are multiples of 16");
    return MI_ESHAPE;
  }
  Workspace ws(workspace, workspace_bytes);
  BilinearPlan p = plan_bilinear(ws, b_rows, b, d_img, d_txt, MI_PREC_FP8);
  if (!ws.ok()) {
    set_error("mi_bilinear_fp8_stage: workspace too small (%zu < %zu)", workspace_bytes, ws.off);
    return MI_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  // non-negative floats and their bit patterns order alike: the absmax slots hold bit patterns, amax_io floats
  auto copy = [&](void* dst, const void* src, size_t n) {
    return launch_copy_words(dst, src, n, st, "copy_words_kernel(fp8 absmax)");
  };
  if (stage == 0) {
    rc = fp8_stage0(x, y, w, b_rows, b, d_img, d_txt, p, st);
    if (rc) return rc;
    return copy(amax_io, p.f8sc->amax_bits, 3 * sizeof(float));
  }
  if (stage == 1) {
    rc = copy(p.f8sc->amax_bits, amax_io, 3 * sizeof(float));
    if (rc) return rc;
    rc = fp8_stage1(x, y, w, b_rows, b, d_img, d_txt, p, st);
    if (rc) return rc;
    return copy(amax_io + 3, &p.f8sc->amax_bits[3], sizeof(float));
  }
  rc = copy(&p.f8sc->amax_bits[3], amax_io + 3, sizeof(float));
  if (rc) return rc;
  return fp8_stage2(b_rows, d_txt, p, st);
}

/* One critic step -- forward, loss and every gradient -- in one call (single GPU: b_rows == b).  Where the fused kernels
 * take the shape this is four launches: conversions + T, the fused B x B kernel, [records -> statistics and loss; partial
 * sums -> dT, grad_y; dX], dW.  Nothing needs the loss between the fused kernel and the gradients, so the statistics are
 * merged inside the third launch (no finalize launch).  Other shapes / precisions: the forward and the backward entry
 * points, one after the other.  grad_out may be NULL (dL/dloss = 1). */
int mi_bilinear_step(const float* x, const float* y, const float* w, const int64_t* sid, int64_t b, int64_t d_img,
                     int64_t d_txt, int estimator, int precision, const float* grad_out, float* loss_out, mi_stats* stats,
                     float* partials_out, float* grad_x, float* grad_y, float* grad_w, void* workspace,
                     size_t workspace_bytes, void* stream) {
  MI_CHECK_ARG(x && y && sid && stats && grad_x && grad_y && workspace, "mi_bilinear_step: null pointer");
  MI_CHECK_ARG((w && grad_w) || (!w && d_img == d_txt),
               "mi_bilinear_step: w == NULL (separable form) needs d_img == d_txt; w != NULL needs grad_w");
  int rc = check_common("mi_bilinear_step", b, b, 0, d_img, d_txt, precision);
  if (rc) return rc;
  MI_CHECK_ARG(estimator == MI_DV || estimator == MI_INFONCE, "mi_bilinear_step: unknown estimator %d", estimator);
  Workspace ws(workspace, workspace_bytes);
  BilinearPlan p = plan_bilinear(ws, b, b, d_img, d_txt, precision);
  if (!ws.ok()) {
    set_error("mi_bilinear_step: workspace too small (%zu < %zu)", workspace_bytes, ws.off);
    return MI_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  if (fast_ok(b, b, d_img, d_txt, precision, w != nullptr) && p.fl.ok && p.tail) {
    rc = fast_prep_and_t(x, y, w, sid, sid, b, b, 0, d_img, d_txt, p, st);
    if (rc) return rc;
    rc = flash_stage(sid, sid, b, b, 0, d_txt, true, p, st);
    if (rc) return rc;
    const TailMerge m{estimator, loss_out, stats, partials_out};
    return bilinear_tail(b, b, 0, d_img, d_txt, stats, grad_out, grad_x, grad_y, grad_w, p, &m, st);
  }
  rc = mi_bilinear_fwd(x, y, w, sid, sid, b, b, 0, d_img, d_txt, estimator, precision, 1, loss_out, stats, partials_out,
                       nullptr, workspace, workspace_bytes, stream);
  if (rc) return rc;
  return mi_bilinear_bwd(x, y, w, sid, sid, b, b, 0, d_img, d_txt, precision, stats, grad_out, grad_x, grad_y, grad_w,
                         workspace, workspace_bytes, 1, stream);
}

/* The same step at a bf16 boundary: x [b][d_img] and y [b][d_txt] are bf16 (what encoders under the autocast policy of
 * encoders.py emit), the gradients for them are written in bf16 when grads_bf16 != 0 (fp32 buffers otherwise); w, grad_w,
 * loss and statistics stay fp32.  The kernels round x and y to bf16 as their first act anyway: with bf16-representable
 * inputs the results are bit-identical to mi_bilinear_step(precision = MI_PREC_BF16), minus 25 MB of conversion traffic
 * per step.  Only where the fused kernels and the two-launch tail take the shape (mi_bilinear_path == MI_PATH_FUSED_TAIL);
 * MI_ESHAPE otherwise (convert and call mi_bilinear_step). */
int mi_bilinear_step_bf16(const void* x_bf16, const void* y_bf16, const float* w, const int64_t* sid, int64_t b,
                          int64_t d_img, int64_t d_txt, int estimator, const float* grad_out, float* loss_out,
                          mi_stats* stats, float* partials_out, void* grad_x, void* grad_y, int grads_bf16, float* grad_w,
                          void* workspace, size_t workspace_bytes, void* stream) {
  MI_CHECK_ARG(x_bf16 && y_bf16 && w && sid && stats && grad_x && grad_y && grad_w && workspace,
               "mi_bilinear_step_bf16: null pointer");
  int rc = check_common("mi_bilinear_step_bf16", b, b, 0, d_img, d_txt, MI_PREC_BF16);
  if (rc) return rc;
  MI_CHECK_ARG(estimator == MI_DV || estimator == MI_INFONCE, "mi_bilinear_step_bf16: unknown estimator %d", estimator);
  Workspace ws(workspace, workspace_bytes);
  BilinearPlan p = plan_bilinear(ws, b, b, d_img, d_txt, MI_PREC_BF16);
  if (!ws.ok()) {
    set_error("mi_bilinear_step_bf16: workspace too small (%zu < %zu)", workspace_bytes, ws.off);
    return MI_EWORKSPACE;
  }
  if (!(fast_ok(b, b, d_img, d_txt, MI_PREC_BF16, true) && p.fl.ok && p.tail) || (uintptr_t)x_bf16 % 16 != 0 ||
      (uintptr_t)y_bf16 % 16 != 0) {
    set_error("mi_bilinear_step_bf16: shape outside the fused kernels (mi_bilinear_path != MI_PATH_FUSED_TAIL) or "
              "embeddings not 16-byte aligned: convert to fp32 and call mi_bilinear_step");
    return MI_ESHAPE;
  }
  hipStream_t st = (hipStream_t)stream;
  const bf16_t* xb = reinterpret_cast<const bf16_t*>(x_bf16);
  p.yb = const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(y_bf16));  // the streamed operand of problem 0: the input itself
  rc = fast_prep_and_t_bf16(xb, p.yb, w, sid, sid, b, b, 0, d_img, d_txt, p, st);
  if (rc) return rc == MI_EINVAL ? MI_ESHAPE : rc;
  rc = flash_stage(sid, sid, b, b, 0, d_txt, true, p, st);
  if (rc) return rc;
  TailMerge m{estimator, loss_out, stats, partials_out};
  m.grads_bf16 = grads_bf16 ? 1 : 0;
  return bilinear_tail(b, b, 0, d_img, d_txt, stats, grad_out, reinterpret_cast<float*>(grad_x),
                       reinterpret_cast<float*>(grad_y), grad_w, p, &m, st);
}

/* Which kernels a bilinear-critic shape takes (host-side arithmetic only, nothing is launched): MI_PATH_*.  The Python
 * binding warns once per shape when a 16-bit call leaves the fused kernels. */
int mi_bilinear_path(int64_t b_rows, int64_t b, int64_t d_img, int64_t d_txt, int precision) {
  if (b_rows <= 0 || b <= 0 || d_img <= 0 || d_txt <= 0) return MI_EINVAL;
  if (precision == MI_PREC_FP8) return fp8_ok(b_rows, b, d_img, d_txt, precision, true) ? MI_PATH_FP8_GEMMS : MI_ESHAPE;
  if (!fast_ok(b_rows, b, d_img, d_txt, precision, true)) return MI_PATH_GENERIC;
  char* fake = reinterpret_cast<char*>(uintptr_t(1) << 20);
  Workspace ws(fake, ~size_t(0) >> 1);
  BilinearPlan p = plan_bilinear(ws, b_rows, b, d_img, d_txt, precision);
  if (!p.fl.ok) return MI_PATH_GEMMS;
  return p.tail ? MI_PATH_FUSED_TAIL : MI_PATH_FUSED;
}

}  // extern "C"

// ================================================================================================ separable critic
// S = (X Wg)(Y Wh)^T  (BASELINE.json configs[1]; an extension: no reference code, oracle-pinned only).  Projections,
// the fused B x B stage and every gradient are this library's kernels:
//   forward : prep X, Y, Wg, Wh -> A = X Wg | C = Y Wh (one two-problem launch; bf16 row-major + fragment-major copies)
//             -> fused kernel (problem 0: A rows x C rows, problem 1: C rows x A rows) -> statistics
//   backward: slab reduce -> dA, dC (bf16, both orientations) -> {dWg = X^T dA | dX = dA Wg^T}, {dWh = Y^T dC | dY = dC Wh^T}
// Other shapes and the fp32 parity mode run on the generic strided-operand kernels (mi_gemm.h).
namespace mi {

struct SeparablePlan {
  bool fast;
  FlashPlan fl;
  bf16_t *xb, *xtb, *yb, *ytb, *gb, *gtb, *hb, *htb;   // X, Y, Wg, Wh as bf16, row-major and transposed
  bf16_t *ab, *afb, *cb, *cfb;                          // projections, row-major and fragment-major
  bf16_t *dab, *datb, *dcb, *dctb;                      // their gradients, both orientations
  float *slab[2];
  Partial* rec[2];
  unsigned char* dup[2];
  float *wg_slab, *wh_slab;
  int wg_splits, wh_splits;
  int64_t wg_kchunk, wh_kchunk;
  // the two-launch backward tail (mi_bilinear_tail.h, round 4): fragment-major Wg, Wh (B operands of dX = dA Wg^T and
  // dY = dC Wh^T inside the tail kernel), X^T, Y^T, dA^T, dC^T (operands of the one dWg | dWh launch)
  bool tail;
  bf16_t *gfb, *hfb, *xtfb, *ytfb, *datfb, *dctfb;
  // generic path
  float *a32, *c32, *da32, *dc32;
  void* g;
  Partial* partials;
  int64_t n_partials;
  size_t bytes;
};

static void split_plan(int64_t rows, int64_t m, int64_t n, int& splits, int64_t& kchunk) {
  const int64_t tiles = ((m + kTile - 1) / kTile) * ((n + kTile - 1) / kTile);
  int64_t sp = (128 + tiles - 1) / tiles;
  kchunk = (rows + sp - 1) / sp;
  kchunk = (kchunk + kG2KT - 1) / kG2KT * kG2KT;
  splits = (int)((rows + kchunk - 1) / kchunk);
}

static SeparablePlan plan_separable(Workspace& ws, int64_t br, int64_t b, int64_t dx, int64_t dy, int64_t k,
                                    int precision) {
  SeparablePlan p{};
  p.fl = FlashPlan{};
  if (precision == MI_PREC_BF16 && dx % 8 == 0 && dy % 8 == 0 && k % 64 == 0) p.fl = flash_plan(br, b, k);
  p.fast = p.fl.ok;
  if (p.fast) {
    p.xb = ws.take<bf16_t>(br * dx);
    p.xtb = ws.take<bf16_t>(br * dx);
    p.yb = ws.take<bf16_t>(b * dy);
    p.ytb = ws.take<bf16_t>(b * dy);
    p.gb = ws.take<bf16_t>(dx * k);
    p.gtb = ws.take<bf16_t>(dx * k);
    p.hb = ws.take<bf16_t>(dy * k);
    p.htb = ws.take<bf16_t>(dy * k);
    p.ab = ws.take<bf16_t>(br * k);
    p.afb = ws.take<bf16_t>(br * k);
    p.cb = ws.take<bf16_t>(b * k);
    p.cfb = ws.take<bf16_t>(b * k);
    p.dab = ws.take<bf16_t>(br * k);
    p.datb = ws.take<bf16_t>(br * k);
    p.dcb = ws.take<bf16_t>(b * k);
    p.dctb = ws.take<bf16_t>(b * k);
    for (int q = 0; q < 2; ++q) {
      p.rec[q] = ws.take<Partial>(p.fl.n_rec[q]);
      p.slab[q] = ws.take<float>((p.fl.slab_bytes[q] + 3) / 4);
      p.dup[q] = ws.take<unsigned char>((br / 32) * (b / 32));
    }
    split_plan(br, dx, k, p.wg_splits, p.wg_kchunk);
    split_plan(b, dy, k, p.wh_splits, p.wh_kchunk);
    p.wg_slab = ws.take<float>((int64_t)p.wg_splits * dx * k);
    p.wh_slab = ws.take<float>((int64_t)p.wh_splits * dy * k);
    static const bool no_tail = getenv("MI_NO_TAIL") != nullptr;  // A/B switch, as for the bilinear critic
    p.tail = !no_tail && flash_tail_ok(br, dx, k) && flash_tail_ok(b, dy, k) && p.fl.slab_f16;
    if (p.tail) {
      p.gfb = ws.take<bf16_t>(dx * k);
      p.hfb = ws.take<bf16_t>(dy * k);
      p.xtfb = ws.take<bf16_t>(br * dx);
      p.ytfb = ws.take<bf16_t>(b * dy);
      p.datfb = ws.take<bf16_t>(br * k);
      p.dctfb = ws.take<bf16_t>(b * k);
    }
  } else {
    p.a32 = ws.take<float>(br * k);
    p.c32 = ws.take<float>(b * k);
    p.da32 = ws.take<float>(br * k);
    p.dc32 = ws.take<float>(b * k);
    p.n_partials = ((b + kTile - 1) / kTile) * ((br + kTile - 1) / kTile);
    p.partials = ws.take<Partial>(p.n_partials);
    if (precision == MI_PREC_BF16) p.g = ws.take<bf16_t>(br * b);
    else p.g = ws.take<float>(br * b);
  }
  p.bytes = ws.off;
  return p;
}

static int separable_prep_project(const float* x, const float* y, const float* wg, const float* wh,
                                  const int64_t* sid_rows, const int64_t* sid_cols, int64_t br, int64_t b,
                                  int64_t row_offset, int64_t dx, int64_t dy, int64_t k, const SeparablePlan& p,
                                  hipStream_t st) {
  CvtJobs jobs{};
  jobs.j[0] = CvtJob{x, br, dx, p.xb, p.xtb, 0, 0, nullptr, 0, 0, p.tail ? p.xtfb : nullptr};
  jobs.j[1] = CvtJob{y, b, dy, p.yb, p.ytb, 0, 0, nullptr, 0, 0, p.tail ? p.ytfb : nullptr};
  jobs.j[2] = CvtJob{wg, dx, k, p.gb, p.gtb, 0, 0, p.tail ? p.gfb : nullptr};
  jobs.j[3] = CvtJob{wh, dy, k, p.hb, p.htb, 0, 0, p.tail ? p.hfb : nullptr};
  jobs.dup = DupFlagJob{sid_rows, sid_cols, (int)(br / 32), (int)(b / 32), p.dup[0], p.dup[1], row_offset};
  int rc = launch_cvt_transpose3(jobs, st, "separable prep X Y Wg Wh");
  if (rc) return rc;
  // A[i, c] = sum_a X[i, a] Wg[a, c] (A operand Xb [br][dx], B operand Wg^T [k][dx]);  C likewise from Y, Wh
  GemmBf16Args two{};
  two.p[0] = GemmBf16Problem{p.xb, dx, p.gtb, dx, br, k, dx};
  two.p[1] = GemmBf16Problem{p.yb, dy, p.htb, dy, b, k, dy};
  two.n_problems = 2;
  two.k_chunk = dx > dy ? dx : dy;
  EpiStoreMulti e{};
  e.out[0] = EpiOut{nullptr, 0, 0, p.ab, k, nullptr, 0, p.afb};
  e.out[1] = EpiOut{nullptr, 0, 0, p.cb, k, nullptr, 0, p.cfb};
  return launch_gemm_bf16(two, 1, e, st, "separable A = X Wg | C = Y Wh");
}

// The one-call step's form of the above (mi_separable_step on the two-launch tail): conversions and BOTH projections in ONE
// launch (bilinear_prep_t_kernel with a second product) -- nobody reads the row-major / transposed bf16 copies of X, Y, Wg,
// Wh on that path, only the fragment-major ones.  MI_EINVAL: a shape the launch does not take (the caller falls back).
static int separable_prep_project_fused(const float* x, const float* y, const float* wg, const float* wh,
                                        const int64_t* sid_rows, const int64_t* sid_cols, int64_t br, int64_t b,
                                        int64_t row_offset, int64_t dx, int64_t dy, int64_t k, const SeparablePlan& p,
                                        hipStream_t st) {
  static const bool off = getenv("MI_SEP_NO_FUSED_PREP") != nullptr;  // A/B switch
  if (off) return MI_EINVAL;
  CvtJobs side{};
  side.j[0] = CvtJob{x, br, dx, nullptr, nullptr, 0, 0, nullptr, 0, 0, p.xtfb};
  side.j[1] = CvtJob{y, b, dy, nullptr, nullptr, 0, 0, nullptr, 0, 0, p.ytfb};
  side.j[2] = CvtJob{wg, dx, k, nullptr, nullptr, 0, 0, p.gfb};
  side.j[3] = CvtJob{wh, dy, k, nullptr, nullptr, 0, 0, p.hfb};
  side.dup = DupFlagJob{sid_rows, sid_cols, (int)(br / 32), (int)(b / 32), p.dup[0], p.dup[1], row_offset};
  const PrepTSecond c{y, wh, b, k, dy, p.cb, p.cfb};
  return launch_prep_t(x, wg, br, k, dx, p.ab, p.afb, side, st, "separable prep + A = X Wg | C = Y Wh", false, &c);
}

static int separable_flash(const int64_t* sid_rows, const int64_t* sid_cols, int64_t br, int64_t b, int64_t row_offset,
                           int64_t k, bool grad, const SeparablePlan& p, hipStream_t st) {
  FlashArgs a{};
  a.p[0] = FlashProblem{p.afb, p.cb, sid_rows, sid_cols, br, b, row_offset, p.fl.n_rb[0], p.fl.n_split[0],
                        p.fl.tiles_per_split[0], p.dup[0], p.slab[0], p.rec[0]};
  a.p[1] = FlashProblem{p.cfb, p.ab, sid_cols, sid_rows, b, br, -row_offset, p.fl.n_rb[1], p.fl.n_split[1],
                        p.fl.tiles_per_split[1], p.dup[1], p.slab[1], p.rec[1]};
  a.n_problems = grad ? 2 : 1;
  a.slab_f16 = p.fl.slab_f16 ? 1 : 0;
  return launch_flash(a, k, grad, st, grad ? "separable fused S | P C | P^T A" : "separable fused S + LSE");
}

static int check_separable(const char* fn, int64_t br, int64_t b, int64_t row_offset, int64_t dx, int64_t dy, int64_t k,
                           int precision) {
  int rc = check_common(fn, br, b, row_offset, dx, dy, precision);
  if (rc) return rc;
  MI_CHECK_ARG(k >= 1, "%s: projection width must be >= 1", fn);
  return MI_OK;
}

}  // namespace mi

extern "C" {

size_t mi_separable_workspace_bytes(int64_t b_rows, int64_t b, int64_t d_img, int64_t d_txt, int64_t d_proj,
                                    int precision) {
  if (b_rows <= 0 || b <= 0 || d_img <= 0 || d_txt <= 0 || d_proj <= 0) return 0;
  Workspace ws(nullptr, 0);
  return plan_separable(ws, b_rows, b, d_img, d_txt, d_proj, precision).bytes + 256;
}

int mi_separable_fwd(const float* x, const float* y, const float* wg, const float* wh, const int64_t* sid_rows,
                     const int64_t* sid_cols, int64_t b_rows, int64_t b, int64_t row_offset, int64_t d_img,
                     int64_t d_txt, int64_t d_proj, int estimator, int precision, int need_grad, float* loss_out,
                     mi_stats* stats, float* partials_out, void* workspace, size_t workspace_bytes, void* stream) {
  MI_CHECK_ARG(x && y && wg && wh && sid_rows && sid_cols && stats && workspace, "mi_separable_fwd: null pointer");
  int rc = check_separable("mi_separable_fwd", b_rows, b, row_offset, d_img, d_txt, d_proj, precision);
  if (rc) return rc;
  MI_CHECK_ARG(estimator == MI_DV || estimator == MI_INFONCE, "mi_separable_fwd: unknown estimator %d", estimator);
  Workspace ws(workspace, workspace_bytes);
  SeparablePlan p = plan_separable(ws, b_rows, b, d_img, d_txt, d_proj, precision);
  if (!ws.ok()) {
    set_error("mi_separable_fwd: workspace too small (%zu < %zu)", workspace_bytes, ws.off);
    return MI_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  if (p.fast) {
    rc = separable_prep_project(x, y, wg, wh, sid_rows, sid_cols, b_rows, b, row_offset, d_img, d_txt, d_proj, p, st);
    if (rc) return rc;
    rc = separable_flash(sid_rows, sid_cols, b_rows, b, row_offset, d_proj, need_grad != 0, p, st);
    if (rc) return rc;
    return launch_finalize(p.rec[0], p.fl.n_rec[0], b, estimator, loss_out, stats, partials_out, st);
  }
  // generic: projections on the strided-operand kernels (exact fp32 products in the parity mode), then the S = A C^T
  // score + LSE kernel of the bilinear path's generic form
  const bool f32 = precision == MI_PREC_F32;
  auto project = [&](const float* in, const float* w, int64_t rows, int64_t d, float* out, const char* what) {
    return f32 ? launch_gemm<float>(make_operand(in, d, 1), make_operand(w, 1, d_proj), rows, d_proj, d,
                                    EpiStore{out, d_proj, nullptr, 1.0f, 0}, st, what)
               : launch_gemm<bf16_t>(make_operand(in, d, 1), make_operand(w, 1, d_proj), rows, d_proj, d,
                                     EpiStore{out, d_proj, nullptr, 1.0f, 0}, st, what);
  };
  rc = project(x, wg, b_rows, d_img, p.a32, "separable A = X Wg (generic)");
  if (rc) return rc;
  rc = project(y, wh, b, d_txt, p.c32, "separable C = Y Wh (generic)");
  if (rc) return rc;
  EpiScoreLse epi{sid_rows, sid_cols, row_offset, nullptr, p.partials};
  rc = f32 ? launch_gemm<float>(make_operand((const float*)p.a32, d_proj, 1), make_operand((const float*)p.c32, d_proj, 1),
                                b_rows, b, d_proj, epi, st, "separable score+LSE (generic)")
           : launch_gemm<bf16_t>(make_operand((const float*)p.a32, d_proj, 1), make_operand((const float*)p.c32, d_proj, 1),
                                 b_rows, b, d_proj, epi, st, "separable score+LSE (generic)");
  if (rc) return rc;
  return launch_finalize(p.partials, p.n_partials, b, estimator, loss_out, stats, partials_out, st);
}

int mi_separable_bwd(const float* x, const float* y, const float* wg, const float* wh, const int64_t* sid_rows,
                     const int64_t* sid_cols, int64_t b_rows, int64_t b, int64_t row_offset, int64_t d_img,
                     int64_t d_txt, int64_t d_proj, int precision, const mi_stats* stats, const float* grad_out,
                     float* grad_x, float* grad_y, float* grad_wg, float* grad_wh, void* workspace,
                     size_t workspace_bytes, int workspace_from_forward, void* stream) {
  MI_CHECK_ARG(x && y && wg && wh && sid_rows && sid_cols && stats && grad_x && grad_y && grad_wg && grad_wh && workspace,
               "mi_separable_bwd: null pointer");
  int rc = check_separable("mi_separable_bwd", b_rows, b, row_offset, d_img, d_txt, d_proj, precision);
  if (rc) return rc;
  Workspace ws(workspace, workspace_bytes);
  SeparablePlan p = plan_separable(ws, b_rows, b, d_img, d_txt, d_proj, precision);
  if (!ws.ok()) {
    set_error("mi_separable_bwd: workspace too small (%zu < %zu)", workspace_bytes, ws.off);
    return MI_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int64_t br = b_rows, k = d_proj;
  if (p.fast) {
    if (!workspace_from_forward) {
      rc = separable_prep_project(x, y, wg, wh, sid_rows, sid_cols, br, b, row_offset, d_img, d_txt, k, p, st);
      if (rc) return rc;
      rc = separable_flash(sid_rows, sid_cols, br, b, row_offset, k, true, p, st);
      if (rc) return rc;
    }
    FlashReduceArgs ra{};
    ra.j[0] = FlashReduceJob{p.slab[0], p.rec[0], p.fl.n_split[0], p.fl.n_rb[0], br, p.cb, b, row_offset, nullptr, p.dab,
                             p.datb};
    ra.j[1] = FlashReduceJob{p.slab[1], p.rec[1], p.fl.n_split[1], p.fl.n_rb[1], b, p.ab, br, -row_offset, nullptr, p.dcb,
                             p.dctb};
    ra.stats = stats;
    ra.grad_out = grad_out;
    rc = launch_flash_reduce(ra, 2, k, p.fl.slab_f16, st, "separable dA, dC from the fused sums");
    if (rc) return rc;
    // per modality: dW[a, c] = sum_i X[i, a] dA[i, c] (split over i into slabs) | dX[i, a] = sum_c dA[i, c] W[a, c]
    auto pair = [&](const bf16_t* xt, const bf16_t* dat, const bf16_t* da, const bf16_t* wb, int64_t rows, int64_t d,
                    float* slab, int splits, int64_t kchunk, float* gx, float* gw, const char* what) {
      GemmBf16Args g{};
      g.p[0] = GemmBf16Problem{xt, rows, dat, rows, d, k, rows};
      g.p[1] = GemmBf16Problem{da, k, wb, k, rows, d, k};
      g.n_problems = 2;
      EpiStoreMulti e{};
      e.out[0] = EpiOut{slab, k, d * k, nullptr, 0, nullptr, 0};
      e.out[1] = EpiOut{gx, d, 0, nullptr, 0, nullptr, 0};
      const int sp[2] = {splits, 1};
      const int64_t ch[2] = {kchunk, k};
      int r = launch_gemm_bf16_flat(g, sp, ch, e, st, what);
      if (r == MI_EINVAL) {
        EpiStoreMulti e1{};
        e1.out[0] = e.out[0];
        r = launch_gemm_bf16(one_problem(xt, rows, dat, rows, d, k, rows, kchunk), splits, e1, st, what);
        if (r) return r;
        e1.out[0] = e.out[1];
        r = launch_gemm_bf16(one_problem(da, k, wb, k, rows, d, k), 1, e1, st, what);
      }
      if (r) return r;
      return launch_slab_reduce_ld(slab, splits, d, k, gw, k, st, "slab_reduce_ld_kernel");
    };
    rc = pair(p.xtb, p.datb, p.dab, p.gb, br, d_img, p.wg_slab, p.wg_splits, p.wg_kchunk, grad_x, grad_wg,
              "separable dWg = X^T dA | dX = dA Wg^T");
    if (rc) return rc;
    return pair(p.ytb, p.dctb, p.dcb, p.hb, b, d_txt, p.wh_slab, p.wh_splits, p.wh_kchunk, grad_y, grad_wh,
                "separable dWh = Y^T dC | dY = dC Wh^T");
  }
  // generic path: recompute the projections (cheap), G from recomputed scores, then six strided-operand products
  const bool f32 = precision == MI_PREC_F32;
#define MI_SEP_GEMM(A, B, M, N, K, OUT, LD, WHAT)                                                                    \
  do {                                                                                                               \
    rc = f32 ? launch_gemm<float>(A, B, M, N, K, EpiStore{OUT, LD, nullptr, 1.0f, 0}, st, WHAT)                       \
             : launch_gemm<bf16_t>(A, B, M, N, K, EpiStore{OUT, LD, nullptr, 1.0f, 0}, st, WHAT);                     \
    if (rc) return rc;                                                                                               \
  } while (0)
  MI_SEP_GEMM(make_operand(x, d_img, 1), make_operand(wg, 1, k), br, k, d_img, p.a32, k, "separable A = X Wg (generic, bwd)");
  MI_SEP_GEMM(make_operand(y, d_txt, 1), make_operand(wh, 1, k), b, k, d_txt, p.c32, k, "separable C = Y Wh (generic, bwd)");
  const float* a = p.a32;
  const float* c = p.c32;
  if (f32) {
    float* g = (float*)p.g;
    rc = launch_gemm<float>(make_operand(a, k, 1), make_operand(c, k, 1), br, b, k,
                            EpiGradScore<float>{sid_rows, sid_cols, row_offset, stats, grad_out, g}, st, "separable G (generic)");
    if (rc) return rc;
    MI_SEP_GEMM(make_operand((const float*)g, b, 1), make_operand(c, 1, k), br, k, b, p.da32, k, "separable dA = G C (generic)");
    MI_SEP_GEMM(make_operand((const float*)g, 1, b), make_operand(a, 1, k), b, k, br, p.dc32, k, "separable dC = G^T A (generic)");
  } else {
    bf16_t* g = (bf16_t*)p.g;
    rc = launch_gemm<bf16_t>(make_operand(a, k, 1), make_operand(c, k, 1), br, b, k,
                             EpiGradScore<bf16_t>{sid_rows, sid_cols, row_offset, stats, grad_out, g}, st, "separable G (generic)");
    if (rc) return rc;
    MI_SEP_GEMM(make_operand((const bf16_t*)g, b, 1), make_operand(c, 1, k), br, k, b, p.da32, k, "separable dA = G C (generic)");
    MI_SEP_GEMM(make_operand((const bf16_t*)g, 1, b), make_operand(a, 1, k), b, k, br, p.dc32, k, "separable dC = G^T A (generic)");
  }
  const float* da = p.da32;
  const float* dc = p.dc32;
  MI_SEP_GEMM(make_operand(da, k, 1), make_operand(wg, k, 1), br, d_img, k, grad_x, d_img, "separable dX = dA Wg^T (generic)");
  MI_SEP_GEMM(make_operand(x, 1, d_img), make_operand(da, 1, k), d_img, k, br, grad_wg, k, "separable dWg = X^T dA (generic)");
  MI_SEP_GEMM(make_operand(dc, k, 1), make_operand(wh, k, 1), b, d_txt, k, grad_y, d_txt, "separable dY = dC Wh^T (generic)");
  MI_SEP_GEMM(make_operand(y, 1, d_txt), make_operand(dc, 1, k), d_txt, k, b, grad_wh, k, "separable dWh = Y^T dC (generic)");
#undef MI_SEP_GEMM
  return MI_OK;
}

}  // extern "C"

/* One separable-critic step in ONE call (single GPU): prep + projections (one launch), the fused B x B kernel, then the two-launch tail
 * of the bilinear critic generalised to two products -- [records -> statistics and loss; partial sums -> dA, dC rows; dX =
 * dA Wg^T, dY = dC Wh^T on the matrix cores; dA^T, dC^T fragment-major] and [dWg = X^T dA | dWh = Y^T dC] -- four launches
 * (round 3: eight, with a finalize, a slab-reduce and two split-K reduce launches).  Shapes outside the fused kernels: the
 * forward and the backward entry points, one after the other. */
extern "C" int mi_separable_step(const float* x, const float* y, const float* wg, const float* wh, const int64_t* sid,
                                 int64_t b, int64_t d_img, int64_t d_txt, int64_t d_proj, int estimator, int precision,
                                 const float* grad_out, float* loss_out, mi_stats* stats, float* partials_out, float* grad_x,
                                 float* grad_y, float* grad_wg, float* grad_wh, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  using namespace mi;
  MI_CHECK_ARG(x && y && wg && wh && sid && stats && grad_x && grad_y && grad_wg && grad_wh && workspace,
               "mi_separable_step: null pointer");
  int rc = check_separable("mi_separable_step", b, b, 0, d_img, d_txt, d_proj, precision);
  if (rc) return rc;
  MI_CHECK_ARG(estimator == MI_DV || estimator == MI_INFONCE, "mi_separable_step: unknown estimator %d", estimator);
  Workspace ws(workspace, workspace_bytes);
  SeparablePlan p = plan_separable(ws, b, b, d_img, d_txt, d_proj, precision);
  if (!ws.ok()) {
    set_error("mi_separable_step: workspace too small (%zu < %zu)", workspace_bytes, ws.off);
    return MI_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  if (p.fast && p.tail) {
    rc = separable_prep_project_fused(x, y, wg, wh, sid, sid, b, b, 0, d_img, d_txt, d_proj, p, st);
    if (rc == MI_EINVAL) rc = separable_prep_project(x, y, wg, wh, sid, sid, b, b, 0, d_img, d_txt, d_proj, p, st);
    if (rc) return rc;
    rc = separable_flash(sid, sid, b, b, 0, d_proj, true, p, st);
    if (rc) return rc;
    FlashTailArgs ta{};
    ta.j[0] = FlashReduceJob{p.slab[0], p.rec[0], p.fl.n_split[0], p.fl.n_rb[0], b, p.cb, b, 0, nullptr, nullptr, nullptr};
    ta.j[1] = FlashReduceJob{p.slab[1], p.rec[1], p.fl.n_split[1], p.fl.n_rb[1], b, p.ab, b, 0, nullptr, nullptr, nullptr};
    ta.stats = stats;
    ta.grad_out = grad_out;
    ta.merge_rec = p.rec[0];
    ta.n_merge = p.fl.n_rec[0];
    ta.n_pos = b;
    ta.estimator = estimator;
    ta.loss_out = loss_out;
    ta.stats_out = stats;
    ta.partials_out = partials_out;
    ta.w_frag = p.gfb;  ta.dx = d_img;  ta.grad_x = grad_x;  ta.dtt_frag = p.datfb;
    ta.w_frag1 = p.hfb; ta.dx1 = d_txt; ta.grad_x1 = grad_y; ta.dtt_frag1 = p.dctfb;
    rc = launch_flash_tail(ta, d_proj, p.fl.slab_f16, true, st, "separable sums -> loss, dA, dC | dX = dA Wg^T | dY = dC Wh^T");
    if (rc) return rc;
    const DwArgs g{p.xtfb, p.datfb, d_img, d_proj, b, grad_wg}, h{p.ytfb, p.dctfb, d_txt, d_proj, b, grad_wh};
    return launch_bilinear_dw(g, st, "separable dWg = X^T dA | dWh = Y^T dC", &h);
  }
  rc = mi_separable_fwd(x, y, wg, wh, sid, sid, b, b, 0, d_img, d_txt, d_proj, estimator, precision, 1, loss_out, stats,
                        partials_out, workspace, workspace_bytes, stream);
  if (rc) return rc;
  return mi_separable_bwd(x, y, wg, wh, sid, sid, b, b, 0, d_img, d_txt, d_proj, precision, stats, grad_out, grad_x, grad_y,
                          grad_wg, grad_wh, workspace, workspace_bytes, 1, stream);
}

extern "C" int mi_separable_path(int64_t b_rows, int64_t b, int64_t d_img, int64_t d_txt, int64_t d_proj, int precision) {
  if (b_rows <= 0 || b <= 0 || d_img <= 0 || d_txt <= 0 || d_proj <= 0) return MI_EINVAL;
  char* fake = reinterpret_cast<char*>(uintptr_t(1) << 20);
  mi::Workspace ws(fake, ~size_t(0) >> 1);
  const mi::SeparablePlan p = mi::plan_separable(ws, b_rows, b, d_img, d_txt, d_proj, precision);
  return p.fast ? (p.tail ? MI_PATH_FUSED_TAIL : MI_PATH_FUSED) : MI_PATH_GENERIC;
}

#ifdef MI_STAMPS
// diagnostic build only: where the stamped kernels of this translation unit write their s_memtime values
extern "C" int mi_debug_set_stamps(void* buf) {
  mi::g_stamp_buf = (unsigned long long*)buf;
  return 0;
}
#endif
