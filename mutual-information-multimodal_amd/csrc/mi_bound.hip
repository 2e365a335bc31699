// Bound reductions of the MI critic path: Donsker-Varadhan and the reference's "InfoNCE" bound.
//   reference: mutual_info_img_txt/mi_critics.py:3-12 (dv_bound_loss), :14-23 (infonce_bound_loss)
// HBM-bound kernels: the forward reads 4 bytes per logit once, the backward reads 4 and writes 4.
// Reductions are two-stage and merged in a fixed order (no float atomics): results are bit-reproducible.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "mi_common.h"

namespace mi {

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int hip_fail(hipError_t e, const char* what) {
  set_error("%s: %s", what, hipGetErrorString(e));
  return MI_EHIP;
}

// ------------------------------------------------------------------------------------------------ profiling
struct ProfEntry {
  const char* name;
  hipEvent_t a, b;
};
static bool g_prof_on = false;
static std::vector<ProfEntry> g_prof;
static std::vector<hipEvent_t> g_prof_pool;
bool profile_enabled() { return g_prof_on; }
int xcd_natural() {
  static const int v = getenv("MI_XCD_NATURAL") ? 1 : 0;
  return v;
}
static hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) {
    hipEvent_t e = g_prof_pool.back();
    g_prof_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
void profile_push(const char* name, hipStream_t st, bool begin) {
  if (begin) {
    ProfEntry e{name, prof_event(), prof_event()};
    (void)hipEventRecord(e.a, st);
    g_prof.push_back(e);
  } else {
    (void)hipEventRecord(g_prof.back().b, st);
  }
}

// ------------------------------------------------------------------------------------------------ kernels
constexpr int kBlock = 256;
constexpr int kMaxPartialBlocks = 2048;  // 256 CUs x 8 blocks (guide: cap memory-bound grids, grid-stride the rest)

__global__ __launch_bounds__(kBlock) void bound_partials_kernel(const float* __restrict__ logits, int64_t n,
                                                                int64_t pos, Partial* __restrict__ out) {
  __shared__ Partial scratch[kBlock / 64];
  Partial p{MI_NEG_INF, 0.0f, 0.0f, 0u};
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool vec_ok = (((uintptr_t)logits) & 15) == 0;
  const int64_t nvec = vec_ok ? n / 4 : 0;
  for (int64_t v = tid; v < nvec; v += stride) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(logits + 4 * v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t r = 4 * v + e;
      if (r < pos) {
        p.pos += x[e];
      } else {
        lse_push(p.m, p.s, x[e]);
        p.cnt += 1;
      }
    }
  }
  for (int64_t r = 4 * nvec + tid; r < n; r += stride) {
    const float x = logits[r];
    if (r < pos) {
      p.pos += x;
    } else {
      lse_push(p.m, p.s, x);
      p.cnt += 1;
    }
  }
  p = block_reduce_partial<kBlock / 64>(p, scratch);
  if (threadIdx.x == 0) out[blockIdx.x] = p;
}

// One workgroup merges all partial records in a fixed order.
__global__ __launch_bounds__(kBlock) void finalize_kernel(const Partial* __restrict__ partials, int64_t n_partials,
                                                          int64_t n_pos, int estimator, float* loss_out,
                                                          mi_stats* stats, float* local_record) {
  __shared__ Partial scratch[kBlock / 64];
  __shared__ unsigned long long cnt_scratch[kBlock / 64];
  Partial p{MI_NEG_INF, 0.0f, 0.0f, 0u};
  unsigned long long cnt = 0;
  for (int64_t k = threadIdx.x; k < n_partials; k += kBlock) {
    const Partial q = partials[k];
    lse_merge(p.m, p.s, q.m, q.s);
    p.pos += q.pos;
    cnt += q.cnt;
  }
  // 64-bit count reduction
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if ((threadIdx.x & 63) == 0) cnt_scratch[threadIdx.x >> 6] = cnt;
  p.cnt = 0;
  p = block_reduce_partial<kBlock / 64>(p, scratch);
  if (threadIdx.x == 0) {
    unsigned long long total = 0;
    for (int w = 0; w < kBlock / 64; ++w) total += cnt_scratch[w];
    if (local_record) {
      local_record[0] = p.m;
      local_record[1] = p.s;
      local_record[2] = p.pos;
      local_record[3] = (float)(total & 0xFFFFFFull);
      local_record[4] = (float)(total >> 24);
      local_record[5] = local_record[6] = local_record[7] = 0.0f;
    }
    if (stats) {
      const float lse = (p.s > 0.0f) ? p.m + logf(p.s) : MI_NEG_INF;  // logsumexp(empty) = -inf
      const float pos_mean = p.pos / (float)n_pos;
      const float log_n = logf((float)total);  // float32 constant as in mi_critics.py:10
      stats->lse = lse;
      stats->pos_mean = pos_mean;
      stats->log_n_neg = log_n;
      stats->neg_max = p.m;
      stats->loss_dv = (lse - log_n) - pos_mean;
      stats->loss_infonce = lse - pos_mean;
      stats->reserved0 = stats->reserved1 = 0.0f;
      stats->n_neg = (int64_t)total;
      stats->n_pos = n_pos;
      stats->reserved2 = stats->reserved3 = 0;
      if (loss_out) loss_out[0] = (estimator == MI_DV) ? stats->loss_dv : stats->loss_infonce;
    }
  }
}

// Cross-rank merge: records gathered in rank order, merged in rank order (identical bits on every rank).
__global__ void merge_records_kernel(const float* __restrict__ records, int64_t n_ranks, int64_t n_pos,
                                     int estimator, float* loss_out, mi_stats* stats) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float m = MI_NEG_INF, s = 0.0f, pos = 0.0f;
  unsigned long long total = 0;
  for (int64_t r = 0; r < n_ranks; ++r) {
    const float* q = records + 8 * r;
    lse_merge(m, s, q[0], q[1]);
    pos += q[2];
    total += (unsigned long long)q[3] + ((unsigned long long)q[4] << 24);
  }
  const float lse = (s > 0.0f) ? m + logf(s) : MI_NEG_INF;
  const float pos_mean = pos / (float)n_pos;
  const float log_n = logf((float)total);
  stats->lse = lse;
  stats->pos_mean = pos_mean;
  stats->log_n_neg = log_n;
  stats->neg_max = m;
  stats->loss_dv = (lse - log_n) - pos_mean;
  stats->loss_infonce = lse - pos_mean;
  stats->reserved0 = stats->reserved1 = 0.0f;
  stats->n_neg = (int64_t)total;
  stats->n_pos = n_pos;
  stats->reserved2 = stats->reserved3 = 0;
  if (loss_out) loss_out[0] = (estimator == MI_DV) ? stats->loss_dv : stats->loss_infonce;
}

__global__ __launch_bounds__(kBlock) void bound_bwd_kernel(const float* __restrict__ logits, int64_t n, int64_t pos,
                                                           const mi_stats* __restrict__ stats,
                                                           const float* __restrict__ grad_out,
                                                           float* __restrict__ grad_logits) {
  const float go = grad_out ? grad_out[0] : 1.0f;
  const float lse = stats->lse;
  const float gpos = -go / (float)pos;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += stride) {
    grad_logits[r] = (r < pos) ? gpos : go * expf(logits[r] - lse);
  }
}

// scores [b_rows, b]; row r is global row row_offset + r.  One workgroup walks whole rows (no integer division),
// 16-byte loads when the row pitch allows.
__global__ __launch_bounds__(256) void matrix_partials_kernel(const float* __restrict__ scores,
                                                                 const int64_t* __restrict__ sid_rows,
                                                                 const int64_t* __restrict__ sid_cols,
                                                                 int64_t b_rows, int64_t b, int64_t row_offset,
                                                                 Partial* __restrict__ out) {
  __shared__ Partial scratch[kBlock / 64];
  Partial p{MI_NEG_INF, 0.0f, 0.0f, 0u};
  const bool vec_ok = ((((uintptr_t)scores) & 15) == 0) && (b % 4 == 0);
  for (int64_t i = blockIdx.x; i < b_rows; i += gridDim.x) {
    const int64_t si = sid_rows[i];
    const int64_t gi = row_offset + i;
    const float* row = scores + i * b;
    if (vec_ok) {
      for (int64_t j0 = 4 * (int64_t)threadIdx.x; j0 < b; j0 += 4 * kBlock) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(row + j0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kind = pair_kind(gi, j0 + e, si, sid_cols[j0 + e]);
          if (kind == 1) {
            p.pos += x[e];
          } else if (kind == 2) {
            lse_push(p.m, p.s, x[e]);
            p.cnt += 1;
          }
        }
      }
    } else {
      for (int64_t j = threadIdx.x; j < b; j += kBlock) {
        const int kind = pair_kind(gi, j, si, sid_cols[j]);
        const float x = row[j];
        if (kind == 1) {
          p.pos += x;
        } else if (kind == 2) {
          lse_push(p.m, p.s, x);
          p.cnt += 1;
        }
      }
    }
  }
  p = block_reduce_partial<kBlock / 64>(p, scratch);
  if (threadIdx.x == 0) out[blockIdx.x] = p;
}

__global__ __launch_bounds__(kBlock) void matrix_bwd_kernel(const float* __restrict__ scores,
                                                            const int64_t* __restrict__ sid_rows,
                                                            const int64_t* __restrict__ sid_cols, int64_t b_rows,
                                                            int64_t b, int64_t row_offset,
                                                            const mi_stats* __restrict__ stats,
                                                            const float* __restrict__ grad_out,
                                                            float* __restrict__ grad_scores) {
  const float go = grad_out ? grad_out[0] : 1.0f;
  const float lse = stats->lse;
  const float gpos = -go / (float)stats->n_pos;
  for (int64_t i = blockIdx.x; i < b_rows; i += gridDim.x) {
    const int64_t si = sid_rows[i];
    const int64_t gi = row_offset + i;
    for (int64_t j = threadIdx.x; j < b; j += kBlock) {
      const int kind = pair_kind(gi, j, si, sid_cols[j]);
      float g = 0.0f;
      if (kind == 1) g = gpos;
      else if (kind == 2) g = go * expf(scores[i * b + j] - lse);
      grad_scores[i * b + j] = g;
    }
  }
}

static int grid_rows(int64_t rows) {
  int64_t g = rows < 1 ? 1 : rows;
  if (g > kMaxPartialBlocks) g = kMaxPartialBlocks;
  return (int)g;
}

static int grid_for(int64_t n_items) {
  int64_t g = (n_items + kBlock * 4 - 1) / (kBlock * 4);
  if (g < 1) g = 1;
  if (g > kMaxPartialBlocks) g = kMaxPartialBlocks;
  return (int)g;
}

int launch_finalize(const Partial* partials, int64_t n_partials, int64_t n_pos, int estimator, float* loss_out,
                    mi_stats* stats, float* local_record, hipStream_t stream) {
  {
    ProfScope prof_("finalize_kernel", stream);
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(kBlock), 0, stream, partials, n_partials, n_pos, estimator,
                       loss_out, stats, local_record);
  }
  MI_LAUNCH_CHECK("finalize_kernel");
  return MI_OK;
}

}  // namespace mi

using namespace mi;

extern "C" {

int mi_abi_version(void) { return 4; }

int mi_profile_begin(void) {
  for (auto& e : mi::g_prof) {
    mi::g_prof_pool.push_back(e.a);
    mi::g_prof_pool.push_back(e.b);
  }
  mi::g_prof.clear();
  mi::g_prof_on = true;
  return MI_OK;
}

int mi_profile_end(char* names, size_t names_bytes, float* ms, int capacity, int* n_out) {
  mi::g_prof_on = false;
  MI_CHECK_ARG(names && ms && n_out, "mi_profile_end: null pointer");
  hipError_t err = hipDeviceSynchronize();
  if (err != hipSuccess) return hip_fail(err, "hipDeviceSynchronize");
  // An event pair around a launch also times marker-packet handling (a few microseconds, 5-10 % of the bilinear
  // kernels).  Calibrate against the shortest of a few EMPTY pairs on an idle stream: half of it is subtracted, which is
  // what brings these durations to rocprofv3's kernel trace on this stack (r1_d: 43.5 us raw, 40.2 us rocprofv3 for the
  // dominant kernel; subtracting the whole empty pair lands 3 % under the trace, half of it 3 % over).
  float overhead = 0.0f;
  {
    hipEvent_t a = mi::prof_event(), b = mi::prof_event();
    float best = 1e30f;
    for (int rep = 0; rep < 8; ++rep) {
      (void)hipEventRecord(a, nullptr);
      (void)hipEventRecord(b, nullptr);
      if (hipEventSynchronize(b) != hipSuccess) break;
      float t = 0.0f;
      if (hipEventElapsedTime(&t, a, b) == hipSuccess && t < best) best = t;
    }
    if (best < 1e29f) overhead = 0.5f * best;
    mi::g_prof_pool.push_back(a);
    mi::g_prof_pool.push_back(b);
  }
  int n = 0;
  size_t off = 0;
  for (auto& e : mi::g_prof) {
    if (n >= capacity) break;
    const size_t len = strlen(e.name) + 1;
    if (off + len > names_bytes) break;
    float t = 0.0f;
    (void)hipEventElapsedTime(&t, e.a, e.b);
    t -= overhead;
    if (t < 0.0f) t = 0.0f;
    ms[n] = t;
    memcpy(names + off, e.name, len);  // NUL-separated list
    off += len;
    ++n;
  }
  *n_out = n;
  return MI_OK;
}
const char* mi_last_error(void) { return mi::g_err; }

size_t mi_bound_workspace_bytes(int64_t n) {
  (void)n;
  return align_up(sizeof(Partial) * kMaxPartialBlocks, 256) + 256;
}

int mi_bound_fwd(const float* logits, int64_t n, int64_t pos_size, int estimator, float* loss_out, mi_stats* stats,
                 void* workspace, size_t workspace_bytes, void* stream) {
  MI_CHECK_ARG(logits && stats && workspace, "mi_bound_fwd: null pointer");
  MI_CHECK_ARG(n >= 0 && pos_size >= 0 && pos_size <= n, "mi_bound_fwd: need 0 <= pos_size <= n (got %lld, %lld)",
               (long long)pos_size, (long long)n);
  MI_CHECK_ARG(estimator == MI_DV || estimator == MI_INFONCE, "mi_bound_fwd: unknown estimator %d", estimator);
  Workspace ws(workspace, workspace_bytes);
  Partial* partials = ws.take<Partial>(kMaxPartialBlocks);
  if (!ws.ok()) {
    set_error("mi_bound_fwd: workspace too small (%zu < %zu)", workspace_bytes, ws.off);
    return MI_EWORKSPACE;
  }
  const int grid = grid_for(n);
  hipStream_t st = (hipStream_t)stream;
  {
    ProfScope prof_("bound_partials_kernel", st);
    hipLaunchKernelGGL(bound_partials_kernel, dim3(grid), dim3(kBlock), 0, st, logits, n, pos_size, partials);
  }
  MI_LAUNCH_CHECK("bound_partials_kernel");
  return launch_finalize(partials, grid, pos_size, estimator, loss_out, stats, nullptr, st);
}

int mi_bound_bwd(const float* logits, int64_t n, int64_t pos_size, const mi_stats* stats, const float* grad_out,
                 float* grad_logits, void* stream) {
  MI_CHECK_ARG(logits && stats && grad_logits, "mi_bound_bwd: null pointer");
  MI_CHECK_ARG(n >= 0 && pos_size >= 0 && pos_size <= n, "mi_bound_bwd: bad sizes");
  if (n == 0) return MI_OK;
  {
    ProfScope prof_("bound_bwd_kernel", (hipStream_t)stream);
    hipLaunchKernelGGL(bound_bwd_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, logits, n, pos_size,
                       stats, grad_out, grad_logits);
  }
  MI_LAUNCH_CHECK("bound_bwd_kernel");
  return MI_OK;
}

size_t mi_matrix_bound_workspace_bytes(int64_t b) { return mi_bound_workspace_bytes(b * b); }

int mi_matrix_bound_fwd(const float* scores, const int64_t* sid, int64_t b, int estimator, float* loss_out,
                        mi_stats* stats, void* workspace, size_t workspace_bytes, void* stream) {
  MI_CHECK_ARG(scores && sid && stats && workspace, "mi_matrix_bound_fwd: null pointer");
  MI_CHECK_ARG(b >= 1, "mi_matrix_bound_fwd: b must be >= 1");
  MI_CHECK_ARG(estimator == MI_DV || estimator == MI_INFONCE, "mi_matrix_bound_fwd: unknown estimator %d", estimator);
  Workspace ws(workspace, workspace_bytes);
  Partial* partials = ws.take<Partial>(kMaxPartialBlocks);
  if (!ws.ok()) {
    set_error("mi_matrix_bound_fwd: workspace too small");
    return MI_EWORKSPACE;
  }
  const int grid = grid_rows(b);
  hipStream_t st = (hipStream_t)stream;
  {
    ProfScope prof_("matrix_partials_kernel", st);
    hipLaunchKernelGGL(matrix_partials_kernel, dim3(grid), dim3(kBlock), 0, st, scores, sid, sid, b, b, (int64_t)0,
                       partials);
  }
  MI_LAUNCH_CHECK("matrix_partials_kernel");
  return launch_finalize(partials, grid, b, estimator, loss_out, stats, nullptr, st);
}

int mi_matrix_bound_bwd(const float* scores, const int64_t* sid, int64_t b, const mi_stats* stats,
                        const float* grad_out, float* grad_scores, void* stream) {
  MI_CHECK_ARG(scores && sid && stats && grad_scores, "mi_matrix_bound_bwd: null pointer");
  MI_CHECK_ARG(b >= 1, "mi_matrix_bound_bwd: b must be >= 1");
  {
    ProfScope prof_("matrix_bwd_kernel", (hipStream_t)stream);
    hipLaunchKernelGGL(matrix_bwd_kernel, dim3(grid_rows(b)), dim3(kBlock), 0, (hipStream_t)stream, scores, sid, sid,
                       b, b, (int64_t)0, stats, grad_out, grad_scores);
  }
  MI_LAUNCH_CHECK("matrix_bwd_kernel");
  return MI_OK;
}

int mi_merge_partials(const float* partials, int64_t n_ranks, int64_t n_pos_global, int estimator, float* loss_out,
                      mi_stats* stats, void* stream) {
  MI_CHECK_ARG(partials && stats, "mi_merge_partials: null pointer");
  MI_CHECK_ARG(n_ranks >= 1 && n_pos_global >= 1, "mi_merge_partials: bad sizes");
  MI_CHECK_ARG(estimator == MI_DV || estimator == MI_INFONCE, "mi_merge_partials: unknown estimator %d", estimator);
  {
    ProfScope prof_("merge_records_kernel", (hipStream_t)stream);
    hipLaunchKernelGGL(merge_records_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partials, n_ranks, n_pos_global,
                       estimator, loss_out, stats);
  }
  MI_LAUNCH_CHECK("merge_records_kernel");
  return MI_OK;
}

}  // extern "C"
