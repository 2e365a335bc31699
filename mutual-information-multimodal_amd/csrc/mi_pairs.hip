// Pair builder of the MI critic path (the literal, materialising form kept for API compatibility).
//   reference: MultiModalManager.create_mi_pairs, mutual_info_img_txt/main_utils.py:80-110
//     rows 0..B-1           : [img_r ; txt_r]                                   (main_utils.py:93)
//     then gap = 0..B-2, i = 0..B-1, j = (i+gap+1) mod B, kept iff sid_i != sid_j (main_utils.py:99-108)
// The reference builds this with one torch.cat per row (O(B^4 d) bytes); here the row list is an integer
// stream compaction (count -> scan -> emit) and the rows are one coalesced gather.  The fused critic kernels
// never call this: they enumerate (i,j) by index arithmetic.  All integer work is exact; order is the reference's.
#include "mi_common.h"

namespace mi {

constexpr int kPBlock = 256;
constexpr int kPerThread = 4;
constexpr int kChunk = kPBlock * kPerThread;  // 1024 (gap,i) entries per workgroup

__device__ __forceinline__ bool pair_keep(const int64_t* __restrict__ sid, int64_t b, int64_t e, int& i, int& j) {
  const int64_t gap = e / b;
  const int64_t ii = e - gap * b;
  int64_t jj = ii + gap + 1;
  if (jj >= b) jj -= b;
  i = (int)ii;
  j = (int)jj;
  return sid[ii] != sid[jj];
}

// exclusive scan of one int per thread over the workgroup; returns the exclusive prefix, *total = block sum
__device__ __forceinline__ int block_excl_scan(int v, int* total, int* lds /* kPBlock/64 + 1 */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) lds[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < kPBlock / 64; ++w) {
    const int c = lds[w];
    if (w < wave) base += c;
    tot += c;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

__global__ __launch_bounds__(kPBlock) void pairs_count_kernel(const int64_t* __restrict__ sid, int64_t b, int64_t n_entries,
                                                              int* __restrict__ chunk_counts) {
  __shared__ int lds[kPBlock / 64 + 1];
  const int64_t start = (int64_t)blockIdx.x * kChunk + (int64_t)threadIdx.x * kPerThread;
  int c = 0;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int64_t e = start + k;
    int i, j;
    if (e < n_entries && pair_keep(sid, b, e, i, j)) ++c;
  }
  int total;
  block_excl_scan(c, &total, lds);
  if (threadIdx.x == 0) chunk_counts[blockIdx.x] = total;
}

// single workgroup: exclusive scan of the chunk counts; n_rows = b + total
__global__ __launch_bounds__(kPBlock) void pairs_scan_kernel(const int* __restrict__ chunk_counts, int64_t n_chunks,
                                                             int64_t b, int64_t* __restrict__ chunk_offsets,
                                                             int64_t* __restrict__ n_rows) {
  __shared__ int lds[kPBlock / 64 + 1];
  int64_t running = 0;
  for (int64_t base = 0; base < n_chunks; base += kPBlock) {
    const int64_t c = base + threadIdx.x;
    const int v = c < n_chunks ? chunk_counts[c] : 0;
    int total;
    const int ex = block_excl_scan(v, &total, lds);
    if (c < n_chunks) chunk_offsets[c] = running + ex;
    running += total;
  }
  if (threadIdx.x == 0) n_rows[0] = b + running;
}

__global__ __launch_bounds__(kPBlock) void pairs_emit_kernel(const int64_t* __restrict__ sid, int64_t b, int64_t n_entries,
                                                             const int64_t* __restrict__ chunk_offsets,
                                                             int32_t* __restrict__ pair_i, int32_t* __restrict__ pair_j,
                                                             int64_t capacity, int32_t* __restrict__ rowpos) {
  __shared__ int lds[kPBlock / 64 + 1];
  const int64_t start = (int64_t)blockIdx.x * kChunk + (int64_t)threadIdx.x * kPerThread;
  int ii[kPerThread], jj[kPerThread];
  bool keep[kPerThread];
  int c = 0;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int64_t e = start + k;
    keep[k] = (e < n_entries) && pair_keep(sid, b, e, ii[k], jj[k]);
    c += keep[k] ? 1 : 0;
  }
  int total;
  const int ex = block_excl_scan(c, &total, lds);
  int64_t row = b + chunk_offsets[blockIdx.x] + ex;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int64_t e = start + k;
    if (e >= n_entries) break;
    if (keep[k]) {
      if (row < capacity) {
        pair_i[row] = ii[k];
        pair_j[row] = jj[k];
      }
      if (rowpos) rowpos[e] = (int32_t)row;
      ++row;
    } else if (rowpos) {
      rowpos[e] = -1;
    }
  }
}

__global__ void pairs_positive_kernel(int64_t b, int64_t capacity, int32_t* __restrict__ pair_i,
                                      int32_t* __restrict__ pair_j) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < b && r < capacity) {
    pair_i[r] = (int32_t)r;
    pair_j[r] = (int32_t)r;
  }
}

// one workgroup per output row (grid-stride): out[r] = [img[pair_i[r]] ; txt[pair_j[r]]]
__global__ __launch_bounds__(kPBlock) void create_pairs_kernel(const float* __restrict__ img, const float* __restrict__ txt,
                                                               const int32_t* __restrict__ pair_i,
                                                               const int32_t* __restrict__ pair_j, int64_t n_rows,
                                                               int64_t d_img, int64_t d_txt, float* __restrict__ out,
                                                               int vec_ok) {
  const int64_t d = d_img + d_txt;
  for (int64_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
    const float* a = img + (int64_t)pair_i[r] * d_img;
    const float* c = txt + (int64_t)pair_j[r] * d_txt;
    float* o = out + r * d;
    if (vec_ok) {
      for (int64_t k = 4 * (int64_t)threadIdx.x; k < d; k += 4 * kPBlock) {
        const f32x4 v = (k < d_img) ? *reinterpret_cast<const f32x4*>(a + k)
                                    : *reinterpret_cast<const f32x4*>(c + (k - d_img));
        *reinterpret_cast<f32x4*>(o + k) = v;
      }
    } else {
      for (int64_t k = threadIdx.x; k < d; k += kPBlock) o[k] = (k < d_img) ? a[k] : c[k - d_img];
    }
  }
}

// grid (b, 2): blockIdx.y == 0 -> grad_img[i], 1 -> grad_txt[j].  Fixed summation order: positive row, then gaps
// ascending (deterministic).
__global__ __launch_bounds__(kPBlock) void create_pairs_bwd_kernel(const float* __restrict__ grad_out,
                                                                   const int32_t* __restrict__ rowpos, int64_t b,
                                                                   int64_t d_img, int64_t d_txt,
                                                                   float* __restrict__ grad_img,
                                                                   float* __restrict__ grad_txt) {
  const int64_t d = d_img + d_txt;
  const int64_t t = blockIdx.x;
  const bool is_txt = blockIdx.y == 1;
  const int64_t width = is_txt ? d_txt : d_img;
  const int64_t col0 = is_txt ? d_img : 0;
  float* dst = is_txt ? grad_txt + t * d_txt : grad_img + t * d_img;
  for (int64_t k0 = 0; k0 < width; k0 += kPBlock) {
    const int64_t k = k0 + threadIdx.x;
    float acc = (k < width) ? grad_out[t * d + col0 + k] : 0.0f;
    for (int64_t gap = 0; gap < b - 1; ++gap) {
      int64_t i = t;
      if (is_txt) {  // rows whose text index is t: i = (t - gap - 1) mod b
        i = t - gap - 1;
        if (i < 0) i += b;
      }
      const int32_t row = rowpos[gap * b + i];
      if (row >= 0 && k < width) acc += grad_out[(int64_t)row * d + col0 + k];
    }
    if (k < width) dst[k] = acc;
  }
}

}  // namespace mi

using namespace mi;

extern "C" {

size_t mi_pair_index_workspace_bytes(int64_t b) {
  if (b <= 0) return 0;
  const int64_t n_entries = b * (b - 1);
  const int64_t n_chunks = (n_entries + kChunk - 1) / kChunk + 1;
  return align_up(sizeof(int) * n_chunks, 256) + align_up(sizeof(int64_t) * n_chunks, 256) + 512;
}

static int run_count_scan(const int64_t* sid, int64_t b, Workspace& ws, int** counts_out, int64_t** offsets_out,
                          int64_t** nrows_out, int64_t* n_chunks_out, hipStream_t st) {
  const int64_t n_entries = b * (b - 1);
  const int64_t n_chunks = (n_entries + kChunk - 1) / kChunk;
  int* counts = ws.take<int>(n_chunks + 1);
  int64_t* offsets = ws.take<int64_t>(n_chunks + 1);
  int64_t* nrows = ws.take<int64_t>(1);
  if (!ws.ok()) {
    set_error("pair index: workspace too small (%zu needed)", ws.off);
    return MI_EWORKSPACE;
  }
  if (n_chunks > 0) {
    {
      ProfScope prof_("pairs_count_kernel", st);
      hipLaunchKernelGGL(pairs_count_kernel, dim3((unsigned)n_chunks), dim3(kPBlock), 0, st, sid, b, n_entries, counts);
    }
    MI_LAUNCH_CHECK("pairs_count_kernel");
  }
  {
    ProfScope prof_("pairs_scan_kernel", st);
    hipLaunchKernelGGL(pairs_scan_kernel, dim3(1), dim3(kPBlock), 0, st, counts, n_chunks, b, offsets, nrows);
  }
  MI_LAUNCH_CHECK("pairs_scan_kernel");
  *counts_out = counts;
  *offsets_out = offsets;
  *nrows_out = nrows;
  *n_chunks_out = n_chunks;
  return MI_OK;
}

int mi_pairs_count_host(const int64_t* sid, int64_t b, int64_t* n_rows_host, void* workspace, size_t workspace_bytes,
                        void* stream) {
  MI_CHECK_ARG(sid && n_rows_host && workspace, "mi_pairs_count_host: null pointer");
  MI_CHECK_ARG(b >= 1 && b <= 46340, "mi_pairs_count_host: b out of range (1..46340)");
  Workspace ws(workspace, workspace_bytes);
  int* counts;
  int64_t *offsets, *nrows, n_chunks;
  hipStream_t st = (hipStream_t)stream;
  int rc = run_count_scan(sid, b, ws, &counts, &offsets, &nrows, &n_chunks, st);
  if (rc != MI_OK) return rc;
  hipError_t e = hipMemcpyAsync(n_rows_host, nrows, sizeof(int64_t), hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(n_rows)");
  e = hipStreamSynchronize(st);
  if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
  return MI_OK;
}

int mi_pair_index(const int64_t* sid, int64_t b, int32_t* pair_i, int32_t* pair_j, int64_t capacity,
                  int64_t* n_rows_dev, int32_t* rowpos, void* workspace, size_t workspace_bytes, void* stream) {
  MI_CHECK_ARG(sid && pair_i && pair_j && workspace, "mi_pair_index: null pointer");
  MI_CHECK_ARG(b >= 1 && b <= 46340, "mi_pair_index: b out of range (1..46340)");
  MI_CHECK_ARG(capacity >= b, "mi_pair_index: capacity %lld < b", (long long)capacity);
  Workspace ws(workspace, workspace_bytes);
  int* counts;
  int64_t *offsets, *nrows, n_chunks;
  hipStream_t st = (hipStream_t)stream;
  int rc = run_count_scan(sid, b, ws, &counts, &offsets, &nrows, &n_chunks, st);
  if (rc != MI_OK) return rc;
  {
    ProfScope prof_("pairs_positive_kernel", st);
    hipLaunchKernelGGL(pairs_positive_kernel, dim3((unsigned)((b + 255) / 256)), dim3(256), 0, st, b, capacity, pair_i,
                       pair_j);
  }
  MI_LAUNCH_CHECK("pairs_positive_kernel");
  if (n_chunks > 0) {
    {
      ProfScope prof_("pairs_emit_kernel", st);
      hipLaunchKernelGGL(pairs_emit_kernel, dim3((unsigned)n_chunks), dim3(kPBlock), 0, st, sid, b, b * (b - 1), offsets,
                         pair_i, pair_j, capacity, rowpos);
    }
    MI_LAUNCH_CHECK("pairs_emit_kernel");
  }
  if (n_rows_dev) {
    const int rc = launch_copy_words(n_rows_dev, nrows, sizeof(int64_t), st, "copy_words_kernel(n_rows_dev)");
    if (rc) return rc;
  }
  return MI_OK;
}

int mi_create_pairs(const float* embedding_img, const float* embedding_txt, const int32_t* pair_i,
                    const int32_t* pair_j, int64_t n_rows, int64_t d_img, int64_t d_txt, float* out, void* stream) {
  MI_CHECK_ARG(embedding_img && embedding_txt && pair_i && pair_j && out, "mi_create_pairs: null pointer");
  MI_CHECK_ARG(n_rows >= 0 && d_img >= 1 && d_txt >= 1, "mi_create_pairs: bad sizes");
  if (n_rows == 0) return MI_OK;
  const int vec_ok = (d_img % 4 == 0) && (d_txt % 4 == 0) && ((((uintptr_t)embedding_img) & 15) == 0) &&
                     ((((uintptr_t)embedding_txt) & 15) == 0) && ((((uintptr_t)out) & 15) == 0);
  const int64_t grid = n_rows < 65536 ? n_rows : 65536;
  {
    ProfScope prof_("create_pairs_kernel", (hipStream_t)stream);
    hipLaunchKernelGGL(create_pairs_kernel, dim3((unsigned)grid), dim3(kPBlock), 0, (hipStream_t)stream, embedding_img,
                       embedding_txt, pair_i, pair_j, n_rows, d_img, d_txt, out, vec_ok);
  }
  MI_LAUNCH_CHECK("create_pairs_kernel");
  return MI_OK;
}

int mi_create_pairs_bwd(const float* grad_out, const int32_t* rowpos, int64_t b, int64_t d_img, int64_t d_txt,
                        float* grad_img, float* grad_txt, void* stream) {
  MI_CHECK_ARG(grad_out && grad_img && grad_txt, "mi_create_pairs_bwd: null pointer");
  MI_CHECK_ARG(b >= 1 && d_img >= 1 && d_txt >= 1, "mi_create_pairs_bwd: bad sizes");
  MI_CHECK_ARG(rowpos || b == 1, "mi_create_pairs_bwd: rowpos is required for b > 1");
  {
    ProfScope prof_("create_pairs_bwd_kernel", (hipStream_t)stream);
    hipLaunchKernelGGL(create_pairs_bwd_kernel, dim3((unsigned)b, 2), dim3(kPBlock), 0, (hipStream_t)stream, grad_out,
                       rowpos, b, d_img, d_txt, grad_img, grad_txt);
  }
  MI_LAUNCH_CHECK("create_pairs_bwd_kernel");
  return MI_OK;
}

}  // extern "C"
