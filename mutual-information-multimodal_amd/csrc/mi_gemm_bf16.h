// bf16 MFMA GEMMs for K-contiguous bf16 operands (the fast path of the bilinear critic):
//   C[m, n] = sum_k A[m][k] * B[n][k]        A: [M][lda], B: [N][ldb], bf16, 16-byte aligned rows, K % 8 == 0
// Up to two independent problems per launch and split-K (partial sums go to slabs, reduced in a fixed order afterwards);
// the epilogue functors (store in several formats, masked log-sum-exp partial, dL/dS in both orientations) are shared.
//
// Four kernels, picked per launch by launch_gemm_bf16 / launch_gemm_bf16_flat (measurements: profiles/README.md):
//   gemm_bf16_kernel       128 x 128 x 64 tile, 4 waves, global -> registers -> LDS with 144-byte padded rows.  Any K % 8
//                          == 0 shape; the fallback when K is not a multiple of 64.
//   gemm_bf16_glds_kernel  same tile, operands L2 -> LDS by global_load_lds_dwordx4 (XOR-swizzled image), two workgroups
//                          per CU.  The short products with few tiles: T = X W, and dW | dX as one flat launch.
//   gemm_bf16_big_kernel   256 x 256 x 64 tile, 8 waves, one workgroup per CU.  The K = d products over the B x B
//                          matrix (score + LSE, G): half the LDS-DMA bytes per flop of the 128 x 128 tile.
//   gemm_bf16_pipe_kernel  128 x 128 x 64 tile, 8 waves in two groups half an iteration apart (ping-pong), three LDS
//                          stages, counted s_waitcnt vmcnt.  The long-K pair dT = G Y | dY = G^T T.
#pragma once
#include "mi_common.h"
#include "mi_gemm.h"

namespace mi {

// Diagnostic build only (make STAMPS=1 -> lib_stamps/, never loaded by the package): s_memtime stamps of workgroup
// phases into a side buffer that no kernel reads (tools/diag/stamps.py prints the medians).
#ifdef MI_STAMPS
static __device__ unsigned long long* g_stamps;
static unsigned long long* g_stamp_buf = nullptr;
static inline void stamp_select(const char* what, hipStream_t st) {
  const char* f = getenv("MI_STAMP_KERNEL");
  unsigned long long* p = (g_stamp_buf && f && strstr(what, f)) ? g_stamp_buf : nullptr;
  (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_stamps), &p, sizeof(p), 0, hipMemcpyHostToDevice, st);
}
#define MI_STAMP(slot)                                                                                           \
  do {                                                                                                           \
    if (threadIdx.x == 0 && g_stamps)                                                                            \
      g_stamps[((size_t)blockIdx.x + gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z)) * 8 + (slot)] =  \
          __builtin_amdgcn_s_memtime();                                                                          \
  } while (0)
#define MI_STAMP_SELECT(what, st)                                         \
  do {                                                                    \
    stamp_select(what, st);                                               \
    args.exp_mode = getenv("MI_EXP") ? atoi(getenv("MI_EXP")) : 0;        \
  } while (0)
#else
#define MI_STAMP(slot) do {} while (0)
#define MI_STAMP_SELECT(what, st) do {} while (0)
#endif

constexpr int kG2KT = 64;   // k per tile
constexpr int kG2LD = 72;   // LDS row pitch in bf16 elements

// XCD-aware tile order (guide T1).  Workgroups are dealt round-robin over the 8 XCDs (private L2 each) in dispatch
// order (x fastest): with the natural order the tiles that share an A row-panel land on different XCDs and every XCD
// fetches the panel again (measured on dT = G Y: 286 MB fetched per launch against 75 MB of operands).  Remap the
// linear id so that, inside each group of 8 row-panels, panel p runs all its column tiles on XCD p % 8.  Bijective for
// any grid; affects speed only.
__device__ __forceinline__ void xcd_tile_lin(int L, int nx, int ny, int& bx, int& by) {
  const int full = ny / 8;
  if (L < full * 8 * nx) {
    const int g = L / (8 * nx), r = L - g * 8 * nx;
    by = 8 * g + (r & 7);
    bx = r >> 3;
  } else {
    const int r = L - full * 8 * nx;
    by = 8 * full + r / nx;
    bx = r - (r / nx) * nx;
  }
}
__device__ __forceinline__ void xcd_tile(int& bx, int& by) {
  xcd_tile_lin((int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y, (int)gridDim.x, (int)gridDim.y, bx, by);
}

// Short-K problems over a large M x N grid (score / G: K = d): the row-panel order above makes every XCD stream all of
// B (N x K, larger than its 4 MB L2), and the tiles are then fed from the Infinity Cache at about half the L2 rate.
// Cut the tile grid into gy x gx = 8 rectangular blocks, one per XCD, so that the rows of A and B one XCD touches
// ((M / gy + N / gx) * K * 2 bytes) stay L2-resident.  Needs gridDim.y % gy == 0 and gridDim.x % gx == 0.
__device__ __forceinline__ void xcd_block_tile(int gy, int gx, int& bx, int& by) {
  const int nx = (int)gridDim.x, ny = (int)gridDim.y;
  const int L = (int)blockIdx.x + nx * (int)blockIdx.y;
  const int x = L & 7, s = L >> 3;
  const int ty = ny / gy, tx = nx / gx;
  by = (x / gx) * ty + s / tx;
  bx = (x % gx) * tx + s % tx;
}

// launcher side: pick the block grid for an (ny x nx)-tile problem, or 0 x 0 for the row-panel order
static inline void xcd_pick_blocks(int64_t ny, int64_t nx, int64_t tile, int64_t k, int nz, int& gy, int& gx) {
  gy = gx = 0;
  static const bool off = getenv("MI_GEMM_NO_XCD_BLOCKS") != nullptr;  // A/B switch
  if (off || nz != 1) return;
  const int64_t l2 = 3 << 20;  // leave some of the 4 MB for the output stream
  if (nx * tile * k * 2 + 8 * tile * k * 2 <= l2) return;  // all of B plus a few A panels fit anyway
  int64_t best = -1;
  const int cand[4][2] = {{2, 4}, {4, 2}, {1, 8}, {8, 1}};
  for (auto& c : cand) {
    if (ny % c[0] || nx % c[1]) continue;
    const int64_t bytes = (ny / c[0] + nx / c[1]) * tile * k * 2;
    if (best < 0 || bytes < best) {
      best = bytes;
      gy = c[0];
      gx = c[1];
    }
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt_barrier_n() {
  // counted wait for this wave's own LDS-DMA pieces, then the workgroup barrier (a __syncthreads would drain vmcnt to 0)
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// ---- bf16x3: fp32-grade products from three bf16 MFMAs.  v = hi + lo + O(2^-17 |v|) with hi = bf16(v), lo = bf16(v - hi);
// a . b ~= hi_a hi_b + hi_a lo_b + lo_a hi_b.  Instead of three GEMMs the K dimension is tripled: the operand on the A
// side of a product is stored as [hi | hi | lo] along K (role 1), the operand on the B side as [hi | lo | hi] (role 2),
// and the unchanged bf16 kernels run over K' = 3 K.
__device__ __forceinline__ void split3_offsets(int role, int64_t K, int64_t& hi0, int64_t& hi1, int64_t& lo) {
  hi0 = 0;
  hi1 = role == 1 ? K : 2 * K;
  lo = role == 1 ? 2 * K : K;
}
__device__ __forceinline__ float bf16_residual(float v) { return v - (float)(bf16_t)v; }

struct GemmBf16Problem {
  const bf16_t* a;
  int64_t lda;
  const bf16_t* b;
  int64_t ldb;
  int64_t m, n, k;
  // flat launches only (launch_gemm_bf16_flat): own split-K chunk and the problem's range of workgroup ids
  int64_t k_chunk;
  int wg_begin, nx, ny;
};

struct GemmBf16Args {
  GemmBf16Problem p[2];
  int n_problems;   // 1 or 2
  int64_t k_chunk;  // split-K chunk (multiple of 64) when n_problems == 1 and gridDim.z > 1; else >= k
  int exp_mode;        // diagnostic (MI_STAMPS) builds only
  int flat;            // 1: 1-D grid, problem / split / tile decoded from blockIdx.x (set by launch_gemm_bf16_flat)
  int xcd_gy, xcd_gx;  // set by the launcher: 8 = xcd_gy * xcd_gx blocks of the tile grid, one per XCD (0: row panels)
};

// Generic multi-output store epilogue (one set of outputs per problem): out = alpha * acc
struct EpiOut {
  float* f32;           // [M][ld_f32] or null; split-K: slab z at f32 + z * slab_stride
  int64_t ld_f32;
  int64_t slab_stride;
  bf16_t* bf;           // [M][ld_bf] or null
  int64_t ld_bf;
  bf16_t* bf_t;         // transposed [N][ld_bf_t] or null
  int64_t ld_bf_t;
  bf16_t* bf_frag;      // fragment-major copy or null (needs M % 32 == 0, N % 16 == 0): the 8 elements
                        // [row][16 kk + 8 h .. + 7] sit at ((row / 32 * (N / 16) + kk) * 64 + 32 h + row % 32) * 8, i.e. one
                        // 1 KB block per (32-row block, 16-deep step) in the lane order of an MFMA 32x32x16 operand
  int split_bf, split_bf_t;  // bf16x3 roles of bf ([M][3 N]) and bf_t ([N][3 M]); 0: plain.  ld_bf / ld_bf_t then are
                             // the NATURAL widths N and M (staged shapes only: M, N multiples of 8)
};

__device__ __forceinline__ int64_t frag_major_offset(int64_t row, int64_t col8, int64_t n_cols) {
  // col8: first column of an aligned group of 8
  return (((row >> 5) * (n_cols >> 4) + (col8 >> 4)) * 64 + ((col8 >> 3) & 1) * 32 + (row & 31)) * 8;
}

template <class F>
__device__ __forceinline__ void foreach_acc4(f32x16 (&acc)[2][2], int64_t m_base, int64_t n_base, F&& f) {
  // groups of 4 consecutive rows: f(row0, col, v0..v3)
  const int lane = threadIdx.x & 63;
  const int col_l = lane & 31, half = lane >> 5;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t row0 = m_base + tm * 32 + 8 * g + 4 * half;
        const int64_t col = n_base + tn * 32 + col_l;
        f(row0, col, acc[tm][tn][4 * g], acc[tm][tn][4 * g + 1], acc[tm][tn][4 * g + 2], acc[tm][tn][4 * g + 3]);
      }
}

__device__ __forceinline__ void store4_transposed(bf16_t* dst_t, int64_t ld_t, int64_t row0, int64_t col, int64_t M,
                                                  float v0, float v1, float v2, float v3) {
  bf16_t* p = dst_t + col * ld_t + row0;
  if (row0 + 3 < M && ((ld_t & 3) == 0)) {
    bf16x4 v = {(bf16_t)v0, (bf16_t)v1, (bf16_t)v2, (bf16_t)v3};
    *reinterpret_cast<bf16x4*>(p) = v;
  } else {
    const float vv[4] = {v0, v1, v2, v3};
    for (int q = 0; q < 4; ++q)
      if (row0 + q < M) p[q] = (bf16_t)vv[q];
  }
}

// Wave-private LDS staging area of the epilogues (free once the K loop's last barrier has passed)
constexpr int kEpiPitch = 144;                    // bytes per staged row: 64 bf16 + 16 bytes of padding
constexpr int kEpiLdsPerWave = 64 * kEpiPitch;    // 9,216 bytes

__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wave execute in order; this only stops the compiler from moving them across
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Store a wave's 64 x 64 tile (fp32, MFMA accumulator layout) as bf16 -- row-major into rm[(mb + r) * ld_rm + nb + c]
// and / or transposed into tr[(nb + c) * ld_tr + mb + r] -- through the wave's LDS staging area, so that every global
// store is a 16-byte chunk of a 128-byte row segment (the direct accumulator-layout stores are 2-byte / strided 8-byte
// writes).  Needs ld_rm % 8 == 0 (ld_tr % 8 == 0) and 16-byte aligned bases; elements beyond M x N are dropped, which
// with N % 8 == 0 (M % 8 == 0) happens in whole chunks.
__device__ __forceinline__ void wave_tile_store_bf16(f32x16 (&acc)[2][2], char* lds, bf16_t* rm, int64_t ld_rm,
                                                     bf16_t* tr, int64_t ld_tr, int64_t mb, int64_t nb, int64_t M,
                                                     int64_t N, bf16_t* frag = nullptr) {
  const int lane = threadIdx.x & 63;
  const int col_l = lane & 31, half = lane >> 5;
  const int srow = lane >> 3, chunk = lane & 7;
  if (rm || frag) {
    wave_lds_fence();
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row_l = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          *reinterpret_cast<bf16_t*>(lds + row_l * kEpiPitch + (tn * 32 + col_l) * 2) = (bf16_t)acc[tm][tn][r];
        }
    wave_lds_fence();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row_l = it * 8 + srow;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(lds + row_l * kEpiPitch + chunk * 16);
      const int64_t grow = mb + row_l, gcol = nb + chunk * 8;
      if (grow < M && gcol < N) {
        if (rm) *reinterpret_cast<bf16x8*>(rm + grow * ld_rm + gcol) = v;
        if (frag) *reinterpret_cast<bf16x8*>(frag + frag_major_offset(grow, gcol, N)) = v;
      }
    }
  }
  if (tr) {
    wave_lds_fence();
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row0_l = tm * 32 + 8 * g + 4 * half;
          const bf16x4 v = {(bf16_t)acc[tm][tn][4 * g], (bf16_t)acc[tm][tn][4 * g + 1], (bf16_t)acc[tm][tn][4 * g + 2],
                            (bf16_t)acc[tm][tn][4 * g + 3]};
          *reinterpret_cast<bf16x4*>(lds + (tn * 32 + col_l) * kEpiPitch + row0_l * 2) = v;
        }
    wave_lds_fence();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int c_l = it * 8 + srow;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(lds + c_l * kEpiPitch + chunk * 16);
      const int64_t gcol = nb + c_l, grow = mb + chunk * 8;
      if (gcol < N && grow < M) *reinterpret_cast<bf16x8*>(tr + gcol * ld_tr + grow) = v;
    }
  }
}

// bf16x3 variant of the tile store: the natural widths are N (row-major) and M (transposed); the outputs are [M][3 N] and
// [N][3 M] with the hi part written twice and the residual once (split3_offsets).  role == 0: plain store.
__device__ __forceinline__ void wave_tile_store_split(f32x16 (&acc)[2][2], char* lds, bf16_t* rm, int role_rm, bf16_t* tr,
                                                      int role_tr, int64_t mb, int64_t nb, int64_t M, int64_t N) {
  int64_t r0 = 0, r1 = 0, rl = 0, t0 = 0, t1 = 0, tl = 0;
  split3_offsets(role_rm, N, r0, r1, rl);
  split3_offsets(role_tr, M, t0, t1, tl);
  const int64_t ld_rm = role_rm ? 3 * N : N, ld_tr = role_tr ? 3 * M : M;
  wave_tile_store_bf16(acc, lds, rm, ld_rm, tr, ld_tr, mb, nb, M, N);
  if (!role_rm && !role_tr) return;
  wave_tile_store_bf16(acc, lds, role_rm ? rm + r1 : nullptr, ld_rm, role_tr ? tr + t1 : nullptr, ld_tr, mb, nb, M, N);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = bf16_residual(acc[a][b][r]);
  wave_tile_store_bf16(acc, lds, role_rm ? rm + rl : nullptr, ld_rm, role_tr ? tr + tl : nullptr, ld_tr, mb, nb, M, N);
}

// Store a wave's 64 x 64 fp32 tile row-major through the wave's LDS staging area, 32 rows at a time, so that every global
// store is a 16-byte chunk of a 256-byte row segment.  (The direct accumulator-layout stores are 64 four-byte stores per
// lane; in the in-kernel stamps of the backward's dW | dX launch they were 13.7 k of the 29.4 k cycles of a workgroup.)
// Needs ld % 4 == 0, N % 4 == 0 and a 16-byte aligned base; elements beyond M x N are dropped.
__device__ __forceinline__ void wave_tile_store_f32(f32x16 (&acc)[2][2], char* lds, float* dst, int64_t ld, int64_t mb,
                                                    int64_t nb, int64_t M, int64_t N) {
  constexpr int P = 272;  // bytes per staged row: 64 floats + 16 of padding (32 rows: 8,704 bytes <= kEpiLdsPerWave)
  const int lane = threadIdx.x & 63;
  const int col_l = lane & 31, half = lane >> 5;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    wave_lds_fence();
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row_l = (r & 3) + 8 * (r >> 2) + 4 * half;
        *reinterpret_cast<float*>(lds + row_l * P + (tn * 32 + col_l) * 4) = acc[tm][tn][r];
      }
    wave_lds_fence();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = it * 64 + lane, row_l = idx >> 4, c4 = idx & 15;
      const f32x4 v = *reinterpret_cast<const f32x4*>(lds + row_l * P + c4 * 16);
      const int64_t grow = mb + tm * 32 + row_l, gcol = nb + c4 * 4;
      if (grow < M && gcol < N) *reinterpret_cast<f32x4*>(dst + grow * ld + gcol) = v;
    }
  }
}

struct EpiStoreMulti {
  static constexpr bool kReducesPartial = false;
  EpiOut out[2];
  __device__ __forceinline__ void operator()(f32x16 (&acc)[2][2], int64_t mb, int64_t nb, int64_t M, int64_t N, int prob,
                                             int zsplit, char* lds) const {
    const EpiOut& o = out[prob];
    const bool staged = (M % 8 == 0) && (N % 8 == 0) && (!o.bf || o.ld_bf % 8 == 0) && (!o.bf_t || o.ld_bf_t % 8 == 0);
    // (a fragment-major output is only requested for shapes that take the staged path: M % 32 == 0, N % 16 == 0)
    float* f32dst = o.f32 ? o.f32 + (int64_t)zsplit * o.slab_stride : nullptr;
    const bool f32_staged = f32dst && N % 4 == 0 && o.ld_f32 % 4 == 0 && ((uintptr_t)f32dst & 15) == 0;
    if (f32_staged) wave_tile_store_f32(acc, lds, f32dst, o.ld_f32, mb, nb, M, N);
    if ((o.f32 && !f32_staged) || (o.bf && !staged)) {
      float* f = f32_staged ? nullptr : f32dst;
      bf16_t* bfd = staged ? nullptr : o.bf;
      foreach_acc(acc, mb, nb, [&](int64_t row, int64_t col, float v) {
        if (row < M && col < N) {
          if (f) f[row * o.ld_f32 + col] = v;
          if (bfd) bfd[row * o.ld_bf + col] = (bf16_t)v;
        }
      });
    }
    if (staged && (o.split_bf || o.split_bf_t)) {
      wave_tile_store_split(acc, lds, o.bf, o.split_bf, o.bf_t, o.split_bf_t, mb, nb, M, N);  // consumes acc: last use
    } else if (staged) {
      if (o.bf || o.bf_t || o.bf_frag)
        wave_tile_store_bf16(acc, lds, o.bf, o.ld_bf, o.bf_t, o.ld_bf_t, mb, nb, M, N, o.bf_frag);
    } else if (o.bf_t) {
      foreach_acc4(acc, mb, nb, [&](int64_t row0, int64_t col, float v0, float v1, float v2, float v3) {
        if (col < N) store4_transposed(o.bf_t, o.ld_bf_t, row0, col, M, v0, v1, v2, v3);
      });
    }
  }
};

// masked log-sum-exp partial per workgroup (bilinear forward), optional fp32 score store
struct EpiScoreLse2 {
  const int64_t* sid_rows;
  const int64_t* sid_cols;
  int64_t row_offset;
  float* scores;
  Partial* partials;
  static constexpr bool kReducesPartial = true;
  // Per-lane partial of one 64 x 64 wave tile.  Two passes over the 64 accumulators of this lane: the maximum of its
  // negatives first, then one hardware exponential (v_exp_f32) per negative -- no data-dependent rescale branch (bf16
  // mode: the scores themselves carry ~1e-2 relative error, the native exponential's 1e-6 is irrelevant here).
  __device__ __forceinline__ Partial lane_partial(f32x16 (&acc)[2][2], int64_t mb, int64_t nb, int64_t M,
                                                  int64_t N) const {
    const int lane = threadIdx.x & 63;
    const int col_l = lane & 31, half = lane >> 5;
    float mx = MI_NEG_INF, pos = 0.0f;
    unsigned cnt = 0;
    if (!scores && mb + 64 <= M && nb + 64 <= N) {
      // interior tile, no score output.  A pair is a negative iff the study ids differ (the diagonal pairs a sample
      // with itself, so it never counts as one).  Per element: one 64-bit compare, one select, one max; the negative
      // count comes from the compare's lane mask on the scalar unit (s_bcnt1), not from a per-lane add.
      const int64_t dlo = row_offset + mb - nb;  // global row - column of the tile's (0, 0)
      if (__builtin_amdgcn_readfirstlane((int)(dlo > -64 && dlo < 64))) {
        // the tile's row and column ranges meet: pick the positives up first (one tile in 16 at B = 4096)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) {
            const int d = (int)(tn * 32 + col_l - tm * 32 - 4 * half - dlo);  // row offset in the 32-row group
#pragma unroll
            for (int r = 0; r < 16; ++r) pos += (d == (r & 3) + 8 * (r >> 2)) ? acc[tm][tn][r] : 0.0f;
          }
      }
      unsigned cnt_wave = 0;  // wave-uniform
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        int64_t sr[16];
        const int64_t* srp = sid_rows + mb + tm * 32 + 4 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r) sr[r] = srp[(r & 3) + 8 * (r >> 2)];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          const int64_t sc = sid_cols[nb + tn * 32 + col_l];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool neg = sr[r] != sc;
            cnt_wave += (unsigned)__popcll(__ballot(neg));
            const float vn = neg ? acc[tm][tn][r] : MI_NEG_INF;
            mx = fmaxf(mx, vn);
            acc[tm][tn][r] = vn;
          }
        }
      }
      cnt = lane == 0 ? cnt_wave : 0u;
    } else {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          const int64_t col = nb + tn * 32 + col_l;
          const int64_t sc = col < N ? sid_cols[col] : 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t row = mb + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = acc[tm][tn][r];
            int kind = 0;
            if (row < M && col < N) {
              if (scores) scores[row * N + col] = v;
              kind = pair_kind(row_offset + row, col, sid_rows[row], sc);
            }
            if (kind == 1) pos += v;
            if (kind == 2) {
              mx = fmaxf(mx, v);
              cnt += 1;
            } else {
              v = MI_NEG_INF;  // contributes exp(-inf) = 0 below
            }
            acc[tm][tn][r] = v;
          }
        }
    }
    float s = 0.0f;
    if (mx > MI_NEG_INF) {
      // exp(v - mx) = 2^(v log2e - mx log2e): one fma and one v_exp_f32 per element (v = -inf gives 0)
      constexpr float kLog2e = 1.4426950408889634f;
      const float off = -mx * kLog2e;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int r = 0; r < 16; ++r) s += __builtin_amdgcn_exp2f(__builtin_fmaf(acc[tm][tn][r], kLog2e, off));
    }
    return Partial{mx, s, pos, cnt};
  }
  __device__ __forceinline__ void operator()(f32x16 (&acc)[2][2], int64_t mb, int64_t nb, int64_t M, int64_t N, int,
                                             int, char*) const {
    __shared__ Partial scratch[4];
    Partial p = lane_partial(acc, mb, nb, M, N);
    p = block_reduce_partial<4, true>(p, scratch);
    if (threadIdx.x == 0) partials[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = p;
  }
};

// G = grad_out * dL/dS as bf16, row-major [M][N] and transposed [N][M]
struct EpiGradScore2 {
  static constexpr bool kReducesPartial = false;
  const int64_t* sid_rows;
  const int64_t* sid_cols;
  int64_t row_offset;
  const mi_stats* stats;
  const float* grad_out;
  bf16_t* g;
  bf16_t* gt;
  int split;  // bf16x3: g is [M][3 N], gt is [N][3 M], both in the A-side role (staged shapes only)
  __device__ __forceinline__ void operator()(f32x16 (&acc)[2][2], int64_t mb, int64_t nb, int64_t M, int64_t N, int,
                                             int, char* lds) const {
    const float go = grad_out ? grad_out[0] : 1.0f;
    const float lse = stats->lse;
    const float gpos = -go / (float)stats->n_pos;
    // overwrite the accumulators with G, then store both orientations
    const int lane = threadIdx.x & 63;
    const int col_l = lane & 31, half = lane >> 5;
    const bool staged = (M % 8 == 0) && (N % 8 == 0);
    if (staged && mb + 64 <= M && nb + 64 <= N) {
      // interior tile (see EpiScoreLse2::lane_partial): G = go * exp(S - lse) where the study ids differ, else 0;
      // go * exp(v - lse) = go * 2^(v log2e - lse log2e), one fma + v_exp_f32 + mul (bf16 output: native exp is ample)
      const int64_t dlo = row_offset + mb - nb;
      constexpr float kLog2e = 1.4426950408889634f;
      const float off = -lse * kLog2e;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        int64_t sr[16];
        const int64_t* srp = sid_rows + mb + tm * 32 + 4 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r) sr[r] = srp[(r & 3) + 8 * (r >> 2)];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          const int64_t sc = sid_cols[nb + tn * 32 + col_l];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float e = go * __builtin_amdgcn_exp2f(__builtin_fmaf(acc[tm][tn][r], kLog2e, off));
            acc[tm][tn][r] = sr[r] != sc ? e : 0.0f;
          }
        }
      }
      if (__builtin_amdgcn_readfirstlane((int)(dlo > -64 && dlo < 64))) {
        // the tile's row and column ranges meet: the positives get -go / n_pos (one tile in 16 at B = 4096)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) {
            const int d = (int)(tn * 32 + col_l - tm * 32 - 4 * half - dlo);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = (d == (r & 3) + 8 * (r >> 2)) ? gpos : acc[tm][tn][r];
          }
      }
    } else {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          const int64_t col = nb + tn * 32 + col_l;
          const int64_t sc = col < N ? sid_cols[col] : 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t row = mb + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float gv = 0.0f;
            if (row < M && col < N) {
              const int kind = pair_kind(row_offset + row, col, sid_rows[row], sc);
              if (kind == 1) gv = gpos;
              else if (kind == 2) gv = go * __expf(acc[tm][tn][r] - lse);
              if (!staged) g[row * N + col] = (bf16_t)gv;
            }
            acc[tm][tn][r] = gv;
          }
        }
    }
    if (staged && split) {
      wave_tile_store_split(acc, lds, g, 1, gt, 1, mb, nb, M, N);
    } else if (staged) {
      wave_tile_store_bf16(acc, lds, g, N, gt, M, mb, nb, M, N);
    } else {
      foreach_acc4(acc, mb, nb, [&](int64_t row0, int64_t col, float v0, float v1, float v2, float v3) {
        if (col < N) store4_transposed(gt, M, row0, col, M, v0, v1, v2, v3);
      });
    }
  }
};

template <class Epi>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmBf16Args args, Epi epi) {
  kernarg_prefetch<(int)(sizeof(GemmBf16Args) + sizeof(Epi))>();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem_raw);      // [2][128][72]
  bf16_t* Bs = As + 2 * kTile * kG2LD;                    // [2][128][72]

  const int prob = (int)blockIdx.z % args.n_problems;
  const int zsplit = (int)blockIdx.z / args.n_problems;
  const GemmBf16Problem& P = args.p[prob];
  int bx_, by_;
  if (args.xcd_gy) xcd_block_tile(args.xcd_gy, args.xcd_gx, bx_, by_);
  else xcd_tile(bx_, by_);
  const int64_t m0 = (int64_t)by_ * kTile, n0 = (int64_t)bx_ * kTile;
  if (m0 >= P.m || n0 >= P.n) return;  // the grid covers the larger of two problems
  const int64_t kbeg = (int64_t)zsplit * args.k_chunk;
  int64_t kend = kbeg + args.k_chunk;
  if (kend > P.k) kend = P.k;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, half = lane >> 5;
  const int srow = tid >> 3, skv = tid & 7;  // staging: rows srow + 32 q, 16-byte vector skv of the row

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  u32x4 ra[4], rb[4];
  auto load_tile = [&](int64_t k0) {
    const int64_t k = k0 + skv * 8;
    const bool kin = k + 7 < kend;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t am = m0 + srow + 32 * q, bn = n0 + srow + 32 * q;
      const u32x4 z = {0u, 0u, 0u, 0u};
      ra[q] = (kin && am < P.m) ? *reinterpret_cast<const u32x4*>(P.a + am * P.lda + k) : z;
      rb[q] = (kin && bn < P.n) ? *reinterpret_cast<const u32x4*>(P.b + bn * P.ldb + k) : z;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<u32x4*>(&As[(buf * kTile + srow + 32 * q) * kG2LD + skv * 8]) = ra[q];
      *reinterpret_cast<u32x4*>(&Bs[(buf * kTile + srow + 32 * q) * kG2LD + skv * 8]) = rb[q];
    }
  };

  const int64_t nt = (kend - kbeg + kG2KT - 1) / kG2KT;
  if (nt > 0) {
    load_tile(kbeg);
    store_tile(0);
  }
  __syncthreads();
  for (int64_t t = 0; t < nt; ++t) {
    const int buf = (int)(t & 1);
    const bool more = t + 1 < nt;
    if (more) load_tile(kbeg + (t + 1) * kG2KT);
    const bf16_t* at = As + buf * kTile * kG2LD;
    const bf16_t* bt = Bs + buf * kTile * kG2LD;
#pragma unroll
    for (int kk = 0; kk < kG2KT / 16; ++kk) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
        af[tm] = *reinterpret_cast<const bf16x8*>(&at[(wm * 64 + tm * 32 + r32) * kG2LD + kk * 16 + 8 * half]);
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
        bfr[tn] = *reinterpret_cast<const bf16x8*>(&bt[(wn * 64 + tn * 32 + r32) * kG2LD + kk * 16 + 8 * half]);
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm], bfr[tn], acc[tm][tn], 0, 0, 0);
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }
  epi(acc, m0 + wm * 64, n0 + wn * 64, P.m, P.n, prob, zsplit, smem_raw + wave * kEpiLdsPerWave);
}


// ------------------------------------------------------------------------------------------------ LDS-DMA variant
// Same tiling, but the tiles go HBM/L2 -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write
// pass).  The DMA writes a wave's 64 x 16 bytes linearly, so the LDS image is unpadded [128 rows][128 bytes] and bank
// conflicts are removed by an XOR swizzle applied on the SOURCE address and again on the fragment read (guide rule 21):
//   chunk c (16 bytes = 8 k) of row r lives at LDS chunk position c ^ ((r >> 1) & 7).
// Requires K (and the split-K chunk) to be multiples of 64; rows beyond M / N are clamped (they only feed outputs that
// the epilogue discards).  64 KB of LDS -> 2 workgroups per CU.
//
// Flat launches (args.flat): a 1-D grid over two problems of different shape and split count (dX = dT W^T, one pass over
// K = d, next to dW = X^T dT, split 16 ways over K = B); each problem owns a contiguous range of workgroup ids.
template <class Epi>
__global__ __launch_bounds__(256, 2) void gemm_bf16_glds_kernel(GemmBf16Args args, Epi epi) {
  kernarg_prefetch<(int)(sizeof(GemmBf16Args) + sizeof(Epi))>();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // [buf][A | B][128 rows][128 bytes]
  int prob, zsplit, bx_, by_;
  int64_t k_chunk = args.k_chunk;
  if (args.flat) {
    const int L = (int)blockIdx.x;
    prob = (args.n_problems == 2 && L >= args.p[1].wg_begin) ? 1 : 0;
    const GemmBf16Problem& Q = args.p[prob];
    const int local = L - Q.wg_begin, tiles = Q.nx * Q.ny;
    zsplit = local / tiles;
    xcd_tile_lin(local - zsplit * tiles, Q.nx, Q.ny, bx_, by_);
    k_chunk = Q.k_chunk;
  } else {
    prob = (int)blockIdx.z % args.n_problems;
    zsplit = (int)blockIdx.z / args.n_problems;
    if (args.xcd_gy) xcd_block_tile(args.xcd_gy, args.xcd_gx, bx_, by_);
    else xcd_tile(bx_, by_);
  }
  const GemmBf16Problem& P = args.p[prob];
  const int64_t m0 = (int64_t)by_ * kTile, n0 = (int64_t)bx_ * kTile;
  if (m0 >= P.m || n0 >= P.n) return;
  const int64_t kbeg = (int64_t)zsplit * k_chunk;
  if (kbeg >= P.k) return;  // flat launches: padding ids between the problems' ranges
  int64_t kend = kbeg + k_chunk;
  if (kend > P.k) kend = P.k;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, half = lane >> 5;

  // staging: wave w, instruction i (0..3): rows 32 w + 8 i + (lane >> 3), LDS chunk position lane & 7
  constexpr int NI = 4;
  const char* asrc[NI];
  const char* bsrc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int row = 8 * NI * wave + 8 * i + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int64_t am = m0 + row, bn = n0 + row;
    if (am >= P.m) am = P.m - 1;
    if (bn >= P.n) bn = P.n - 1;
    asrc[i] = reinterpret_cast<const char*>(P.a + am * P.lda + kbeg) + chunk * 16;
    bsrc[i] = reinterpret_cast<const char*>(P.b + bn * P.ldb + kbeg) + chunk * 16;
  }
  auto issue_tile = [&](int64_t t, int buf) {
    char* abase = smem_raw + buf * 32768 + (8 * NI * wave) * 128;
    char* bbase = abase + 16384;
    const int64_t koff = t * (kG2KT * 2);  // bytes
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + koff),
                                       (__attribute__((address_space(3))) void*)(abase + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[i] + koff),
                                       (__attribute__((address_space(3))) void*)(bbase + i * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  // fragment read offsets (bytes within a 16 KB operand tile), per kk the chunk index is 2 kk + half
  int aoff[2], boff[2], aswz[2], bswz[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ar = wm * 64 + t * 32 + r32, br_ = wn * 64 + t * 32 + r32;
    aoff[t] = ar * 128;
    boff[t] = br_ * 128;
    aswz[t] = (ar >> 1) & 7;
    bswz[t] = (br_ >> 1) & 7;
  }

  constexpr int KK = kG2KT / 16;  // 16-deep MFMA steps per K tile
  constexpr int kk0 = 0;
  const int64_t nt = (kend - kbeg) / kG2KT;
  if (nt > 0) issue_tile(0, 0);
  __syncthreads();
  for (int64_t t = 0; t < nt; ++t) {
    const int buf = (int)(t & 1);
    if (t + 1 < nt) issue_tile(t + 1, buf ^ 1);
    const char* at = smem_raw + buf * 32768;
    const char* bt = at + 16384;
    // fragments of step q + 1 are requested before the MFMAs of step q (one wave per SIMD has nothing else to cover
    // the ds_read latency); the sched_barriers keep the compiler from sinking the reads back to their first use
    bf16x8 af[2][2], bfr[2][2];
    auto read_frags = [&](int kk, int slot) {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
        af[slot][tm] = *reinterpret_cast<const bf16x8*>(at + aoff[tm] + 16 * ((2 * kk + half) ^ aswz[tm]));
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
        bfr[slot][tn] = *reinterpret_cast<const bf16x8*>(bt + boff[tn] + 16 * ((2 * kk + half) ^ bswz[tn]));
    };
    read_frags(kk0, 0);
#pragma unroll
    for (int q = 0; q < KK; ++q) {
      if (q + 1 < KK) read_frags(kk0 + q + 1, (q + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q & 1][tm], bfr[q & 1][tn], acc[tm][tn], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();  // hipcc waits vmcnt(0) before the barrier: tile t+1 has landed, buffer `buf` is free
  }
  epi(acc, m0 + wm * 64, n0 + wn * 64, P.m, P.n, prob, zsplit, smem_raw + wave * kEpiLdsPerWave);
}

// Four-stage form of the kernel above for the SHORT-K products (T = X W, dX = dT W^T, the split-K slices of dW: eight
// 64-deep steps each).  With two stages and one __syncthreads per step (which drains vmcnt to 0) every step waits out the
// L2 latency of the tile issued at its start (~1,800 cycles per step, stamps of round 1, against 512 cycles of MFMA);
// here three tiles are in flight behind counted waits (s_waitcnt vmcnt(pieces still allowed in flight) + bare
// s_barrier), 128 KB of LDS, one workgroup per CU.
template <class Epi>
__global__ __launch_bounds__(256, 1) void gemm_bf16_glds4_kernel(GemmBf16Args args, Epi epi) {
  kernarg_prefetch<(int)(sizeof(GemmBf16Args) + sizeof(Epi))>();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // [buf][A | B][128 rows][128 bytes]
  int prob, zsplit, bx_, by_;
  int64_t k_chunk = args.k_chunk;
  if (args.flat) {
    const int L = (int)blockIdx.x;
    prob = (args.n_problems == 2 && L >= args.p[1].wg_begin) ? 1 : 0;
    const GemmBf16Problem& Q = args.p[prob];
    const int local = L - Q.wg_begin, tiles = Q.nx * Q.ny;
    zsplit = local / tiles;
    xcd_tile_lin(local - zsplit * tiles, Q.nx, Q.ny, bx_, by_);
    k_chunk = Q.k_chunk;
  } else {
    prob = (int)blockIdx.z % args.n_problems;
    zsplit = (int)blockIdx.z / args.n_problems;
    if (args.xcd_gy) xcd_block_tile(args.xcd_gy, args.xcd_gx, bx_, by_);
    else xcd_tile(bx_, by_);
  }
  const GemmBf16Problem& P = args.p[prob];
  const int64_t m0 = (int64_t)by_ * kTile, n0 = (int64_t)bx_ * kTile;
  if (m0 >= P.m || n0 >= P.n) return;
  const int64_t kbeg = (int64_t)zsplit * k_chunk;
  if (kbeg >= P.k) return;  // flat launches: padding ids between the problems' ranges
  int64_t kend = kbeg + k_chunk;
  if (kend > P.k) kend = P.k;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, half = lane >> 5;

  // staging: wave w, instruction i (0..3): rows 32 w + 8 i + (lane >> 3), LDS chunk position lane & 7
  constexpr int NI = 4;
  const char* asrc[NI];
  const char* bsrc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int row = 8 * NI * wave + 8 * i + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int64_t am = m0 + row, bn = n0 + row;
    if (am >= P.m) am = P.m - 1;
    if (bn >= P.n) bn = P.n - 1;
    asrc[i] = reinterpret_cast<const char*>(P.a + am * P.lda + kbeg) + chunk * 16;
    bsrc[i] = reinterpret_cast<const char*>(P.b + bn * P.ldb + kbeg) + chunk * 16;
  }
  auto issue_tile = [&](int64_t t, int buf) {
    char* abase = smem_raw + buf * 32768 + (8 * NI * wave) * 128;
    char* bbase = abase + 16384;
    const int64_t koff = t * (kG2KT * 2);  // bytes
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + koff),
                                       (__attribute__((address_space(3))) void*)(abase + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[i] + koff),
                                       (__attribute__((address_space(3))) void*)(bbase + i * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  // fragment read offsets (bytes within a 16 KB operand tile), per kk the chunk index is 2 kk + half
  int aoff[2], boff[2], aswz[2], bswz[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ar = wm * 64 + t * 32 + r32, br_ = wn * 64 + t * 32 + r32;
    aoff[t] = ar * 128;
    boff[t] = br_ * 128;
    aswz[t] = (ar >> 1) & 7;
    bswz[t] = (br_ >> 1) & 7;
  }

  constexpr int KK = kG2KT / 16;  // 16-deep MFMA steps per K tile
  constexpr int kk0 = 0;
  constexpr int NST = 4, PCS = 2 * NI;  // stages; LDS-DMA pieces per wave and tile
  const int64_t nt = (kend - kbeg) / kG2KT;
  MI_STAMP(0);
  for (int t = 0; t < NST - 1; ++t)
    if (t < nt) issue_tile(t, t);
  for (int64_t t = 0; t < nt; ++t) {
    if (t == 1) MI_STAMP(1);
    const int buf = (int)(t & (NST - 1));
    // tile t landed (own pieces: all but the younger tiles' may stay in flight), then everybody's; every wave has left
    // tile t - 1, whose stage the issue below refills
    const int64_t younger = nt - 1 - t < NST - 2 ? nt - 1 - t : NST - 2;
    if (younger >= 2) wait_vmcnt_barrier_n<2 * PCS>();
    else if (younger == 1) wait_vmcnt_barrier_n<PCS>();
    else wait_vmcnt_barrier_n<0>();
    if (t + NST - 1 < nt) issue_tile(t + NST - 1, (int)((t + NST - 1) & (NST - 1)));
    const char* at = smem_raw + buf * 32768;
    const char* bt = at + 16384;
    bf16x8 af[2][2], bfr[2][2];
    auto read_frags = [&](int kk, int slot) {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
        af[slot][tm] = *reinterpret_cast<const bf16x8*>(at + aoff[tm] + 16 * ((2 * kk + half) ^ aswz[tm]));
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
        bfr[slot][tn] = *reinterpret_cast<const bf16x8*>(bt + boff[tn] + 16 * ((2 * kk + half) ^ bswz[tn]));
    };
    read_frags(kk0, 0);
#pragma unroll
    for (int q = 0; q < KK; ++q) {
      if (q + 1 < KK) read_frags(kk0 + q + 1, (q + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q & 1][tm], bfr[q & 1][tn], acc[tm][tn], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  MI_STAMP(2);
  __syncthreads();  // all fragment reads done: the stages are free for the epilogue's staging areas
  epi(acc, m0 + wm * 64, n0 + wn * 64, P.m, P.n, prob, zsplit, smem_raw + wave * kEpiLdsPerWave);
  MI_STAMP(3);
}

// ------------------------------------------------------------------------------------------------ pipelined kernel
// In-kernel s_memtime stamps of the double-buffered kernels above (profiles/README.md): a 128 x 128 x 64 step took
// ~1,800 cycles against 512 cycles of MFMA per SIMD.  The time goes to ISSUING the LDS-DMA pieces (a wave is held
// ~180 cycles per 1 KB piece while eight waves issue) and, behind one barrier per K tile, both waves of a SIMD sit in
// that phase together and then compete for the MFMA pipe together.  This kernel:
//   * runs 8 waves as two groups (one wave per SIMD each) half an iteration apart: the memory phase of one group
//     (fragment reads, LDS-DMA issue, wait) lies under the MFMA phase of the other, a barrier ends every phase;
//   * keeps three LDS stages, two K tiles in flight: the wait is a counted s_waitcnt vmcnt(pieces of ONE tile) followed
//     by a bare s_barrier (hipcc's __syncthreads would drain vmcnt to 0).
// Instantiated shape: 128 x 128 tile, K tile 64, waves 2 x 2 x 2 K groups (waves 4..7 take the second half of every K
// tile; the two partial accumulators are added through LDS before the epilogue).  The configuration struct also
// admits a 256 x 256 tile with 32-deep K tiles (waves 2 x 4, each 128 x 64); on the K = d products it measured slower
// than gemm_bf16_big_kernel (score 26.8 -> 29.5 us) and is not instantiated.
// A stage is 32 KB (96 KB of LDS, one workgroup per CU); every wave issues 4 LDS-DMA pieces per K tile.
// Swizzle: chunk c (16 bytes) of row r sits at chunk position c ^ f(r), f(r) = (r >> 1) & 7 for 128-byte rows and
// (r >> 2) & 3 for 64-byte rows: conflict-free for the lane groups ds_read_b128 serves per LDS cycle.
template <int BM_, int BN_, int KT_, int WM_, int WN_, int KG_>
struct PipeCfg {
  static constexpr int BM = BM_, BN = BN_, KT = KT_, WM = WM_, WN = WN_, KG = KG_;
  static constexpr int NW = WM * WN * KG;                 // waves
  static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;  // MFMA tiles per wave
  static constexpr int ROWB = KT * 2;                     // bytes per LDS row
  static constexpr int CPR = ROWB / 16;                   // 16-byte chunks per row
  static constexpr int RPP = 1024 / ROWB;                 // rows per LDS-DMA piece
  static constexpr int STAGE = (BM + BN) * ROWB;          // bytes per stage
  static constexpr int STAGES = 3;
  static constexpr int PIECES = STAGE / 1024 / NW;        // pieces per wave and K tile
  static constexpr int APIECES = BM * ROWB / 1024;        // pieces 0 .. APIECES-1 belong to A
  static constexpr int KK = KT / 16 / KG;                 // MFMA K steps per wave and K tile
  static constexpr size_t SMEM = (size_t)STAGES * STAGE;
  static_assert(NW == 8 && PIECES * NW * 1024 == STAGE && TN == 2 && (TM == 2 || TM == 4), "unsupported shape");
  __device__ static __forceinline__ int swz(int row) { return KT == 64 ? (row >> 1) & 7 : (row >> 2) & 3; }
};
using PipeCfg128 = PipeCfg<128, 128, 64, 2, 2, 2>;
// 256 x 128 tile, waves 2 x 2 x 2 K groups, each 128 x 64: for two long-K problems whose N is 768 or 1024 (dT | dY of the
// G-materialising path at the reference's embedding width): 2 x 16 x 6 = 192 / 2 x 16 x 8 = 256 workgroups at B = 4096,
// where 128 x 128 tiles make 384 / 512 (one and a half / two rounds of workgroups) and 256 x 256 tiles 96 / 128
using PipeCfg256x128 = PipeCfg<256, 128, 64, 2, 2, 2>;

template <int N>
__device__ __forceinline__ void wait_vmcnt_barrier() {
  // counted wait for this wave's own pieces, then the workgroup barrier; "memory" keeps LDS accesses on their side
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <class Cfg, class Epi>
__global__ __launch_bounds__(512, 1) void gemm_bf16_pipe_kernel(GemmBf16Args args, Epi epi) {
  kernarg_prefetch<(int)(sizeof(GemmBf16Args) + sizeof(Epi))>();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int prob, zsplit, bx_, by_;
  int64_t k_chunk = args.k_chunk;
  if (args.flat) {
    const int L = (int)blockIdx.x;
    prob = (args.n_problems == 2 && L >= args.p[1].wg_begin) ? 1 : 0;
    const GemmBf16Problem& Q = args.p[prob];
    const int local = L - Q.wg_begin, tiles = Q.nx * Q.ny;
    zsplit = local / tiles;
    xcd_tile_lin(local - zsplit * tiles, Q.nx, Q.ny, bx_, by_);
    k_chunk = Q.k_chunk;
  } else {
    prob = (int)blockIdx.z % args.n_problems;
    zsplit = (int)blockIdx.z / args.n_problems;
    if (args.xcd_gy) xcd_block_tile(args.xcd_gy, args.xcd_gx, bx_, by_);
    else xcd_tile(bx_, by_);
  }
  const GemmBf16Problem& P = args.p[prob];
  const int64_t m0 = (int64_t)by_ * Cfg::BM, n0 = (int64_t)bx_ * Cfg::BN;
  if (m0 >= P.m || n0 >= P.n) return;
  const int64_t kbeg = (int64_t)zsplit * k_chunk;
  if (kbeg >= P.k) return;  // flat launches: padding ids between the problems' ranges
  int64_t kend = kbeg + k_chunk;
  if (kend > P.k) kend = P.k;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int kgrp = wave / (Cfg::WM * Cfg::WN), wmn = wave % (Cfg::WM * Cfg::WN);
  const int wm = wmn / Cfg::WN, wn = wmn % Cfg::WN;
  const int r32 = lane & 31, half = lane >> 5;

  // LDS-DMA pieces of this wave: piece q = PIECES * wave + i; a piece is RPP rows of ROWB bytes
  const char* src[Cfg::PIECES];
  int dst[Cfg::PIECES];
#pragma unroll
  for (int i = 0; i < Cfg::PIECES; ++i) {
    const int q = Cfg::PIECES * wave + i;
    const bool is_b = q >= Cfg::APIECES;
    const int ql = is_b ? q - Cfg::APIECES : q;
    const int row = ql * Cfg::RPP + lane / Cfg::CPR;
    const int chunk = (lane % Cfg::CPR) ^ Cfg::swz(row);
    int64_t g = (is_b ? n0 : m0) + row;
    const int64_t lim = is_b ? P.n : P.m;
    if (g >= lim) g = lim - 1;  // clamped rows only feed outputs the epilogue drops
#ifdef MI_STAMPS
    if (args.exp_mode == 95) g &= 255;               // diagnostic: every tile reads the same 256 rows (L2-resident)
#endif
    const bf16_t* base = is_b ? P.b + g * P.ldb : P.a + g * P.lda;
    src[i] = reinterpret_cast<const char*>(base + kbeg) + chunk * 16;
    dst[i] = (is_b ? Cfg::BM * Cfg::ROWB : 0) + ql * 1024;
  }
  auto issue_piece = [&](int64_t t, int stage, int i) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + t * Cfg::ROWB),
                                     (__attribute__((address_space(3))) void*)(smem_raw + stage * Cfg::STAGE + dst[i]),
                                     16, 0, 0);
  };

  f32x16 acc[Cfg::TM / 2][2][2];
#pragma unroll
  for (int h = 0; h < Cfg::TM / 2; ++h)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][a][b][r] = 0.0f;

  int aoff[Cfg::TM], aswz[Cfg::TM], boff[2], bswz[2];
#pragma unroll
  for (int t = 0; t < Cfg::TM; ++t) {
    const int ar = wm * (Cfg::BM / Cfg::WM) + t * 32 + r32;
    aoff[t] = ar * Cfg::ROWB;
    aswz[t] = Cfg::swz(ar);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int br_ = wn * (Cfg::BN / Cfg::WN) + t * 32 + r32;
    boff[t] = Cfg::BM * Cfg::ROWB + br_ * Cfg::ROWB;
    bswz[t] = Cfg::swz(br_);
  }

  const int64_t nt = (kend - kbeg) / Cfg::KT;
  MI_STAMP(0);
#pragma unroll
  for (int i = 0; i < Cfg::PIECES; ++i)
    if (nt > 0) issue_piece(0, 0, i);
#pragma unroll
  for (int i = 0; i < Cfg::PIECES; ++i)
    if (nt > 1) issue_piece(1, 1, i);
  if (nt > 1) wait_vmcnt_barrier<Cfg::PIECES>();  // tile 0 has landed (every wave's pieces)
  else wait_vmcnt_barrier<0>();
  MI_STAMP(1);

  // Ping-pong schedule.  Issuing LDS-DMA pieces holds a wave for hundreds of cycles (the stamps: ~180 per piece with
  // eight waves issuing at once), and behind one barrier per K tile both waves of a SIMD sit in that phase together,
  // then compete for the MFMA pipe together.  Here the waves form two groups, one wave per SIMD each, half an
  // iteration apart: while one group is in its memory phase (fragment reads of tile t, LDS-DMA for tile t + 2, wait
  // for its pieces of tile t + 1) the other runs the MFMAs of its tile, and a barrier ends every phase.
  //   phase:    0        1        2        3      ...   2 nt
  //   group 0:  MEM(0)   MMA(0)   MEM(1)   MMA(1) ...   -
  //   group 1:  -        MEM(0)   MMA(0)   MEM(1) ...   MMA(nt - 1)
  // Tile t is read in phases 2t (group 0) and 2t + 1 (group 1); every wave has waited for its own pieces of it by the
  // end of phase 2t - 1, and its stage is refilled (tile t + 3) from phase 2t + 2 on.
  const int grp = wave >> 2;  // waves w and w + 4 share a SIMD
  constexpr int NM = Cfg::TM * 2;  // MFMAs per 16-deep step
  bf16x8 af[Cfg::KK][Cfg::TM], bfr[Cfg::KK][2];
  const int kk0 = kgrp * Cfg::KK;
  if (grp == 1) asm volatile("s_barrier" ::: "memory");  // phase 0 belongs to group 0
  int stage = 0;
  for (int64_t t = 0; t < nt; ++t) {
    if (t == nt / 4) MI_STAMP(2);
    if (t == (3 * nt) / 4) MI_STAMP(3);
    // ---- memory phase
    const char* sb = smem_raw + stage * Cfg::STAGE;
#pragma unroll
    for (int q = 0; q < Cfg::KK; ++q) {
#pragma unroll
      for (int tm = 0; tm < Cfg::TM; ++tm)
        af[q][tm] = *reinterpret_cast<const bf16x8*>(sb + aoff[tm] + 16 * ((2 * (kk0 + q) + half) ^ aswz[tm]));
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
        bfr[q][tn] = *reinterpret_cast<const bf16x8*>(sb + boff[tn] + 16 * ((2 * (kk0 + q) + half) ^ bswz[tn]));
    }
    const int nstage = stage >= 1 ? stage - 1 : 2;  // (stage + 2) % 3
    if (t + 2 < nt) {
#pragma unroll
      for (int i = 0; i < Cfg::PIECES; ++i) issue_piece(t + 2, nstage, i);
      wait_vmcnt_barrier<Cfg::PIECES>();  // own pieces of tile t + 1 done; end of the phase
    } else {
      wait_vmcnt_barrier<0>();
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- MFMA phase (the fragment reads had the whole barrier to land)
#pragma unroll
    for (int q = 0; q < Cfg::KK; ++q)
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const int tm = m >> 1, tn = m & 1;
        acc[tm >> 1][tm & 1][tn] =
            __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q][tm], bfr[q][tn], acc[tm >> 1][tm & 1][tn], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");  // end of the phase
    stage = stage == 2 ? 0 : stage + 1;
  }
  if (grp == 0) asm volatile("s_barrier" ::: "memory");  // phase 2 nt belongs to group 1
  __syncthreads();  // all fragment reads done: the stages are free for the K-group reduction and the epilogue staging
  MI_STAMP(4);
  if constexpr (Cfg::KG == 2) {
    // add the second K group's accumulators: [wmn][register 0..63][lane] fp32 = 64 KB
    // (one 64-row half of the wave tile at a time: the 256-row shape has two)
    float* red = reinterpret_cast<float*>(smem_raw) + wmn * 4096 + lane;
#pragma unroll
    for (int h = 0; h < Cfg::TM / 2; ++h) {
      if (kgrp == 1) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((a * 2 + b) * 16 + r) * 64] = acc[h][a][b][r];
      }
      __syncthreads();
      if (kgrp == 0) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[h][a][b][r] += red[((a * 2 + b) * 16 + r) * 64];
      }
      __syncthreads();  // the next half / the epilogue's staging areas overlap other waves' parts of `red`
    }
    if (kgrp == 1) return;
  }
  const int64_t mb = m0 + wm * (Cfg::BM / Cfg::WM), nb = n0 + wn * (Cfg::BN / Cfg::WN);
  char* lds = smem_raw + wmn * kEpiLdsPerWave;
  if constexpr (Epi::kReducesPartial) {
    static_assert(Cfg::KG == 1, "reducing epilogues run on all waves");
    __shared__ Partial scratch[8];
    Partial p = epi.lane_partial(acc[0], mb, nb, P.m, P.n);
    if constexpr (Cfg::TM == 4) {
      const Partial q = epi.lane_partial(acc[Cfg::TM / 2 - 1], mb + 64, nb, P.m, P.n);
      lse_merge<true>(p.m, p.s, q.m, q.s);
      p.pos += q.pos;
      p.cnt += q.cnt;
    }
    p = block_reduce_partial<8, true>(p, scratch);
    if (threadIdx.x == 0) epi.partials[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = p;
  } else {
    epi(acc[0], mb, nb, P.m, P.n, prob, zsplit, lds);
    if constexpr (Cfg::TM == 4) epi(acc[Cfg::TM / 2 - 1], mb + 64, nb, P.m, P.n, prob, zsplit, lds);
  }
  MI_STAMP(5);
}

// ------------------------------------------------------------------------------------------------ 256 x 256 tiles
// The K = d (512) GEMMs over the B x B score matrix (score+LSE, G) are bound by LDS ingest per CU, not by MFMA issue:
// a 128 x 128 tile moves 32 KB per 64-deep K step for 2 MFLOP.  A 256 x 256 tile doubles the flops per ingested byte
// and reads 6 fragments per 8 MFMAs instead of 4 per 4.  512 threads = 8 waves (2 x 4), each 128 x 64 = 4 x 2 MFMA
// tiles (128 accumulator registers); same LDS-DMA + XOR swizzle as above; 2 x 64 KB of LDS, one workgroup per CU, so a
// 4096 x 4096 problem is exactly one workgroup per CU.  Single problem, no split-K.
// FP8 = true (mi_fp8.h): the same kernel on e4m3 operands -- a 128-byte tile row is then 128 K elements instead of 64;
// the problem's pointers, pitches and K are passed in units of two bytes, so staging and swizzle are byte-identical and
// only the fragment reads (32 bytes per lane and 64-deep step: chunks 4 kk + 2 half, + 1) and the MFMA
// (v_mfma_scale_f32_32x32x64_f8f6f4, unit block scales) differ.
typedef int i32x8_t __attribute__((ext_vector_type(8)));
template <class Epi, bool FP8 = false>
__global__ __launch_bounds__(512, 1) void gemm_bf16_big_kernel(GemmBf16Args args, Epi epi) {
  kernarg_prefetch<(int)(sizeof(GemmBf16Args) + sizeof(Epi))>();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // [buf][A | B][256 rows][128 bytes]
  const int prob = (int)blockIdx.z;  // two independent problems in one grid (dT | dY of the G-materialising paths)
  const GemmBf16Problem& P = args.p[prob];
  int bx_, by_;
  if (args.xcd_gy) xcd_block_tile(args.xcd_gy, args.xcd_gx, bx_, by_);
  else xcd_tile(bx_, by_);
  const int64_t m0 = (int64_t)by_ * 256, n0 = (int64_t)bx_ * 256;
  if (m0 >= P.m || n0 >= P.n) return;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave >> 2, wn = wave & 3;
  const int r32 = lane & 31, half = lane >> 5;

  const char* asrc[4];
  const char* bsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 32 * wave + 8 * i + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int64_t am = m0 + row, bn = n0 + row;
    if (am >= P.m) am = P.m - 1;
    if (bn >= P.n) bn = P.n - 1;
    asrc[i] = reinterpret_cast<const char*>(P.a + am * P.lda) + chunk * 16;
    bsrc[i] = reinterpret_cast<const char*>(P.b + bn * P.ldb) + chunk * 16;
  }
  auto issue_tile = [&](int64_t t, int buf) {
    char* abase = smem_raw + buf * 65536 + (32 * wave) * 128;
    char* bbase = abase + 32768;
    const int64_t koff = t * (kG2KT * 2);  // bytes
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + koff),
                                       (__attribute__((address_space(3))) void*)(abase + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[i] + koff),
                                       (__attribute__((address_space(3))) void*)(bbase + i * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[2][2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][a][b][r] = 0.0f;

  int aoff[4], boff[2], aswz[4], bswz[2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ar = wm * 128 + t * 32 + r32;
    aoff[t] = ar * 128;
    aswz[t] = (ar >> 1) & 7;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int br_ = wn * 64 + t * 32 + r32;
    boff[t] = br_ * 128;
    bswz[t] = (br_ >> 1) & 7;
  }

  const int64_t nt = P.k / kG2KT;
  MI_STAMP(0);
  if (nt > 0) issue_tile(0, 0);
  __syncthreads();
  MI_STAMP(1);
  for (int64_t t = 0; t < nt; ++t) {
    const int buf = (int)(t & 1);
    if (t == nt / 4) MI_STAMP(2);
    if (t == (3 * nt) / 4) MI_STAMP(3);
    if (t + 1 < nt) issue_tile(t + 1, buf ^ 1);
    const char* at = smem_raw + buf * 65536;
    const char* bt = at + 32768;
    bf16x8 af[2][4], bfr[2][2];
    auto read_frags = [&](int kk, int slot) {
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
        af[slot][tm] = *reinterpret_cast<const bf16x8*>(at + aoff[tm] + 16 * ((2 * kk + half) ^ aswz[tm]));
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
        bfr[slot][tn] = *reinterpret_cast<const bf16x8*>(bt + boff[tn] + 16 * ((2 * kk + half) ^ bswz[tn]));
    };
    if constexpr (FP8) {
      union Frag8 {
        i32x8_t v;
        u32x4 h[2];
      } a8[2][4], b8[2][2];
      auto read8 = [&](int kk, int slot) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
          a8[slot][tm].h[0] = *reinterpret_cast<const u32x4*>(at + aoff[tm] + 16 * ((4 * kk + 2 * half) ^ aswz[tm]));
          a8[slot][tm].h[1] = *reinterpret_cast<const u32x4*>(at + aoff[tm] + 16 * ((4 * kk + 2 * half + 1) ^ aswz[tm]));
        }
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          b8[slot][tn].h[0] = *reinterpret_cast<const u32x4*>(bt + boff[tn] + 16 * ((4 * kk + 2 * half) ^ bswz[tn]));
          b8[slot][tn].h[1] = *reinterpret_cast<const u32x4*>(bt + boff[tn] + 16 * ((4 * kk + 2 * half + 1) ^ bswz[tn]));
        }
      };
      read8(0, 0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if (kk == 0) read8(1, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
            acc[tm >> 1][tm & 1][tn] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                a8[kk][tm].v, b8[kk][tn].v, acc[tm >> 1][tm & 1][tn], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      read_frags(0, 0);
#pragma unroll
      for (int kk = 0; kk < kG2KT / 16; ++kk) {
        if (kk + 1 < kG2KT / 16) read_frags(kk + 1, (kk + 1) & 1);  // ahead of this step's MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
            acc[tm >> 1][tm & 1][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][tm], bfr[kk & 1][tn],
                                                                               acc[tm >> 1][tm & 1][tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }
  const int64_t mb = m0 + wm * 128, nb = n0 + wn * 64;
  MI_STAMP(4);
  if constexpr (Epi::kReducesPartial) {
    __shared__ Partial scratch[8];
    Partial p = epi.lane_partial(acc[0], mb, nb, P.m, P.n);
    MI_STAMP(6);
    const Partial q = epi.lane_partial(acc[1], mb + 64, nb, P.m, P.n);
    MI_STAMP(7);
    lse_merge<true>(p.m, p.s, q.m, q.s);
    p.pos += q.pos;
    p.cnt += q.cnt;
    p = block_reduce_partial<8, true>(p, scratch);
    if (threadIdx.x == 0) epi.partials[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = p;
  } else {
    char* lds = smem_raw + wave * kEpiLdsPerWave;
    epi(acc[0], mb, nb, P.m, P.n, prob, 0, lds);
    MI_STAMP(6);
    epi(acc[1], mb + 64, nb, P.m, P.n, prob, 0, lds);
    MI_STAMP(7);
  }
  MI_STAMP(5);
}

constexpr size_t kG2SmemBig = 2 * 2 * 32768;  // 131,072 bytes

// number of workgroup partials a reducing epilogue writes for an M x N problem (the launcher picks the tile size)
static inline bool gemm_old_kernels() {  // A/B switch: the double-buffered kernels instead of the pipelined one
  static const bool v = getenv("MI_GEMM_OLD") != nullptr;
  return v;
}
static inline bool gemm_use_glds4() {  // A/B switch: the two-stage kernel instead of the four-stage one
  static const bool off = getenv("MI_GEMM_NO_GLDS4") != nullptr;
  return !off;
}
static inline bool gemm_bf16_use_big(int64_t m, int64_t n, int64_t k) {
  static const bool off = getenv("MI_GEMM_NO_BIG") != nullptr;
  return !off && k % kG2KT == 0 && k > 0 && ((m + 255) / 256) * ((n + 255) / 256) >= 192;
}
static inline int64_t gemm_bf16_n_partials(int64_t m, int64_t n, int64_t k) {
  const int64_t t = gemm_bf16_use_big(m, n, k) ? 256 : kTile;
  return ((m + t - 1) / t) * ((n + t - 1) / t);
}

constexpr size_t kG2SmemGlds = 2 * 2 * 16384;  // 65,536 bytes
constexpr size_t kG2Smem = 2 * 2 * kTile * kG2LD * sizeof(bf16_t);  // 73,728 bytes

template <class Epi>
static inline int launch_gemm_bf16(const GemmBf16Args& args_in, int n_splits, const Epi& epi, hipStream_t st,
                                   const char* what) {
  GemmBf16Args args = args_in;
  args.flat = 0;
  args.xcd_gy = args.xcd_gx = 0;
  int64_t mm = args.p[0].m, nn = args.p[0].n;
  if (args.n_problems == 2) {
    if (args.p[1].m > mm) mm = args.p[1].m;
    if (args.p[1].n > nn) nn = args.p[1].n;
  }
  if (mm <= 0 || nn <= 0) return MI_OK;
  // per Epi instantiation, per device, thread-safe (mi_common.h)
  MI_SET_DYN_SMEM((gemm_bf16_kernel<Epi>), kG2Smem, "hipFuncSetAttribute(gemm_bf16_kernel)");
  MI_SET_DYN_SMEM((gemm_bf16_glds_kernel<Epi>), kG2SmemGlds, "hipFuncSetAttribute(gemm_bf16_glds_kernel)");
  MI_SET_DYN_SMEM((gemm_bf16_glds4_kernel<Epi>), 2 * kG2SmemGlds, "hipFuncSetAttribute(gemm_bf16_glds4_kernel)");
  if constexpr (!Epi::kReducesPartial)
    MI_SET_DYN_SMEM((gemm_bf16_pipe_kernel<PipeCfg128, Epi>), PipeCfg128::SMEM,
                    "hipFuncSetAttribute(gemm_bf16_pipe_kernel 128)");
  bool dma_ok = args.k_chunk % kG2KT == 0 || n_splits == 1;
  for (int q = 0; q < args.n_problems; ++q) dma_ok = dma_ok && args.p[q].k % kG2KT == 0 && args.p[q].k > 0;
  // two long-K problems that fill the chip with 256 x 256 tiles together (dT | dY at B = 8192: 2 x 128 tiles): the
  // same kernel, blockIdx.z = problem.  MI_GEMM_NO_BIG2=1: A/B switch.
  static const bool no_big2 = getenv("MI_GEMM_NO_BIG2") != nullptr;
  bool big2 = false;
  if constexpr (!Epi::kReducesPartial)
    big2 = !no_big2 && dma_ok && args.n_problems == 2 && n_splits == 1 && args.p[0].k >= 2048 && args.p[1].k >= 2048 &&
           !gemm_old_kernels() && 2 * ((mm + 255) / 256) * ((nn + 255) / 256) >= 192;
  // long K, N too wide for one round of 128 x 128 tiles, too narrow to fill the chip with 256 x 256 ones: 256 x 128 tiles on
  // the ping-pong kernel when they make one round that fills at least 5/8 of the CUs.  MI_GEMM_NO_P256=1: A/B switch.
  if constexpr (!Epi::kReducesPartial) {
    static const bool no_p256 = getenv("MI_GEMM_NO_P256") != nullptr;
    const int64_t t128 = ((mm + 127) / 128) * ((nn + 127) / 128) * args.n_problems;
    const int64_t t256 = ((mm + 255) / 256) * ((nn + 127) / 128) * args.n_problems;
    static const int64_t p256_min_k = getenv("MI_GEMM_P256_MINK") ? atoll(getenv("MI_GEMM_P256_MINK")) : 2048;  // A/B
    if (!big2 && !no_p256 && dma_ok && n_splits == 1 && args.p[0].k >= p256_min_k && !gemm_old_kernels() && t128 > 256 &&
        t256 <= 256 && t256 >= 160 && !(args.n_problems == 1 && gemm_bf16_use_big(mm, nn, args.p[0].k))) {
      MI_SET_DYN_SMEM((gemm_bf16_pipe_kernel<PipeCfg256x128, Epi>), PipeCfg256x128::SMEM,
                      "hipFuncSetAttribute(gemm_bf16_pipe_kernel 256x128)");
      dim3 grid((unsigned)((nn + 127) / 128), (unsigned)((mm + 255) / 256), (unsigned)args.n_problems);
      MI_STAMP_SELECT(what, st);
      {
        ProfScope prof_(what, st);
        hipLaunchKernelGGL((gemm_bf16_pipe_kernel<PipeCfg256x128, Epi>), grid, dim3(512), PipeCfg256x128::SMEM, st, args, epi);
      }
      MI_LAUNCH_CHECK(what);
      return MI_OK;
    }
  }
  if (big2 || (dma_ok && args.n_problems == 1 && n_splits == 1 && gemm_bf16_use_big(mm, nn, args.p[0].k))) {
    MI_SET_DYN_SMEM((gemm_bf16_big_kernel<Epi>), kG2SmemBig, "hipFuncSetAttribute(gemm_bf16_big_kernel)");
    dim3 grid((unsigned)((nn + 255) / 256), (unsigned)((mm + 255) / 256), (unsigned)args.n_problems);
    xcd_pick_blocks(grid.y, grid.x, 256, args.p[0].k, 1, args.xcd_gy, args.xcd_gx);
    MI_STAMP_SELECT(what, st);
    {
      ProfScope prof_(what, st);
      hipLaunchKernelGGL((gemm_bf16_big_kernel<Epi>), grid, dim3(512), kG2SmemBig, st, args, epi);
    }
    MI_LAUNCH_CHECK(what);
    return MI_OK;
  }
  dim3 grid((unsigned)((nn + kTile - 1) / kTile), (unsigned)((mm + kTile - 1) / kTile),
            (unsigned)(args.n_problems * n_splits));
  xcd_pick_blocks(grid.y, grid.x, kTile, args.p[0].k, (int)grid.z, args.xcd_gy, args.xcd_gx);
  MI_STAMP_SELECT(what, st);
  {
    ProfScope prof_(what, st);
    bool piped = false;
    if constexpr (!Epi::kReducesPartial) {
      // measured (profiles/README.md): the ping-pong kernel wins on the long-K launch that has one workgroup per CU
      // (dT | dY: 51.5 -> 43.3 us) and loses where two 4-wave workgroups per CU fit (T, dW | dX) or K is short
      piped = dma_ok && !gemm_old_kernels() && args.p[0].k >= 2048 &&
              (int64_t)grid.x * grid.y * grid.z <= 256;
      if (piped)
        hipLaunchKernelGGL((gemm_bf16_pipe_kernel<PipeCfg128, Epi>), grid, dim3(512), PipeCfg128::SMEM, st, args, epi);
    }
    if (piped) {
    } else if (dma_ok && gemm_use_glds4() && args.p[0].k >= 4 * kG2KT && (int64_t)grid.x * grid.y * grid.z <= 512)
      hipLaunchKernelGGL((gemm_bf16_glds4_kernel<Epi>), grid, dim3(256), 2 * kG2SmemGlds, st, args, epi);
    else if (dma_ok) hipLaunchKernelGGL((gemm_bf16_glds_kernel<Epi>), grid, dim3(256), kG2SmemGlds, st, args, epi);
    else hipLaunchKernelGGL((gemm_bf16_kernel<Epi>), grid, dim3(256), kG2Smem, st, args, epi);
  }
  MI_LAUNCH_CHECK(what);
  return MI_OK;
}

// Two problems with their own split counts in one launch (problem q: splits[q] K chunks of k_chunks[q], partial sums to
// the epilogue's slabs).  Falls back to one launch per problem when a shape does not suit the LDS-DMA kernel.
template <class Epi>
static inline int launch_gemm_bf16_flat(const GemmBf16Args& args_in, const int (&splits)[2], const int64_t (&k_chunks)[2],
                                        const Epi& epi, hipStream_t st, const char* what) {
  GemmBf16Args args = args_in;
  bool ok = args.n_problems == 2;
  int total = 0;
  for (int q = 0; q < 2 && ok; ++q) {
    GemmBf16Problem& P = args.p[q];
    ok = P.m > 0 && P.n > 0 && P.k > 0 && P.k % kG2KT == 0 && (splits[q] == 1 || k_chunks[q] % kG2KT == 0);
    P.k_chunk = splits[q] == 1 ? P.k : k_chunks[q];
    P.nx = (int)((P.n + kTile - 1) / kTile);
    P.ny = (int)((P.m + kTile - 1) / kTile);
    P.wg_begin = total;
    total += P.nx * P.ny * splits[q];
    total = (total + 7) / 8 * 8;  // keep every problem's first workgroup on XCD 0 (ids in the gap return at once)
  }
  if (!ok) return MI_EINVAL;
  MI_SET_DYN_SMEM((gemm_bf16_glds_kernel<Epi>), kG2SmemGlds, "hipFuncSetAttribute(gemm_bf16_glds_kernel)");
  MI_SET_DYN_SMEM((gemm_bf16_glds4_kernel<Epi>), 2 * kG2SmemGlds, "hipFuncSetAttribute(gemm_bf16_glds4_kernel)");
  args.flat = 1;
  args.xcd_gy = args.xcd_gx = 0;
  args.k_chunk = 0;
  MI_STAMP_SELECT(what, st);
  {
    ProfScope prof_(what, st);
    if (gemm_use_glds4()) hipLaunchKernelGGL((gemm_bf16_glds4_kernel<Epi>), dim3((unsigned)total), dim3(256), 2 * kG2SmemGlds, st, args, epi);
    else hipLaunchKernelGGL((gemm_bf16_glds_kernel<Epi>), dim3((unsigned)total), dim3(256), kG2SmemGlds, st, args, epi);
  }
  MI_LAUNCH_CHECK(what);
  return MI_OK;
}

// ------------------------------------------------------------------------------------------------ prep kernels
// in [R][C] fp32 -> out_rm [R][C] bf16 (optional) and out_t [C][R] bf16 (optional); 32 x 32 tiles through LDS
static __global__ __launch_bounds__(256) void cvt_transpose_kernel(const float* __restrict__ in, int64_t R, int64_t C,
                                                            bf16_t* __restrict__ out_rm, bf16_t* __restrict__ out_t) {
  __shared__ float tile[32][33];
  const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 8 row groups
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t r = r0 + ty + 8 * q, c = c0 + tx;
    const float v = (r < R && c < C) ? in[r * C + c] : 0.0f;
    tile[ty + 8 * q][tx] = v;
    if (out_rm && r < R && c < C) out_rm[r * C + c] = (bf16_t)v;
  }
  if (!out_t) return;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t c = c0 + ty + 8 * q, r = r0 + tx;
    if (c < C && r < R) out_t[c * R + r] = (bf16_t)tile[tx][ty + 8 * q];
  }
}

static inline int launch_cvt_transpose(const float* in, int64_t R, int64_t C, bf16_t* out_rm, bf16_t* out_t,
                                       hipStream_t st, const char* what) {
  dim3 grid((unsigned)((C + 31) / 32), (unsigned)((R + 31) / 32));
  {
    ProfScope prof_(what, st);
    hipLaunchKernelGGL(cvt_transpose_kernel, grid, dim3(256), 0, st, in, R, C, out_rm, out_t);
  }
  MI_LAUNCH_CHECK(what);
  return MI_OK;
}

// up to three conversions in one launch (blockIdx.z selects the job; the grid covers the largest one)
struct CvtJob {
  const float* in;
  int64_t R, C;
  bf16_t* out_rm;
  bf16_t* out_t;
  int n_slab;           // > 1: in holds n_slab partial sums [n_slab][R][C], added in slab order before the conversion
  int64_t slab_stride;  // elements between slabs
  bf16_t* out_frag;     // fragment-major copy (see EpiOut::bf_frag; R % 32 == 0, C % 16 == 0) or null
  int split_rm;         // bf16x3 mode (split3_offsets): out_rm is [R][3 C] holding hi and lo parts; 0: plain bf16
  int split_t;          // same for out_t, [C][3 R]
  bf16_t* out_t_frag;   // fragment-major copy of the TRANSPOSE [C][R] (C % 32 == 0, R % 16 == 0) or null
  int in_bf16;          // `in` points to bf16 data (the bf16 boundary, mi_bilinear_step_bf16): copies instead of conversions
};
// equal-id flags of the fused bilinear kernel, computed by spare workgroups of the conversion launch (blockIdx.z == 3)
struct DupFlagJob {
  const int64_t* sid_rows;
  const int64_t* sid_cols;
  int na, nb;              // 32-row blocks of the two id lists; na == 0: no job
  unsigned char* flag;     // [na][nb]
  unsigned char* flag_t;   // [nb][na]
  int64_t row_offset;      // global index of local row 0: pair (i, j) is a sample with itself iff j == i + row_offset
};
struct CvtJobs {
  CvtJob j[4];     // blockIdx.z = 0..3 (unused jobs have R == 0)
  DupFlagJob dup;  // blockIdx.z = 4
};

// flag[a][b]: 0 = no image row of block a shares a study id with a text row of block b; 1 = the only such pairs are the
// 32 samples paired with themselves on the block's main diagonal (the fused kernel then masks by index, no id compares);
// 2 = anything else (exact id compares in the kernel).  One call handles the 64 column blocks starting at b0 for row
// block a (256 threads: 4 per column block, 8 rows each); exact 64-bit compares.
__device__ __forceinline__ void dup_flags_block(const DupFlagJob& J, int a, int b0, int64_t* rows_lds) {
  const int tid = threadIdx.x;
  if (tid < 32) rows_lds[tid] = J.sid_rows[a * 32 + tid];
  __syncthreads();
  const int q = tid & 3, b = b0 + (tid >> 2);
  if (b < J.nb) {
    int64_t mine[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) mine[e] = rows_lds[8 * q + e];
    const int64_t* c = J.sid_cols + (int64_t)b * 32;
    // the block's main diagonal pairs samples with themselves iff the diagonal of the pair matrix runs through it aligned
    const bool diag_block = (J.row_offset & 31) == 0 && (int64_t)b * 32 == (int64_t)a * 32 + J.row_offset;
    bool any = false, other = false;
#pragma unroll 8
    for (int j = 0; j < 32; ++j) {
      const int64_t v = c[j];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool eq = v == mine[e];
        any |= eq;
        other |= eq && !(diag_block && j == 8 * q + e);
      }
    }
    int f = other ? 2 : (any ? 1 : 0);
    f |= __shfl_xor(f, 1);
    f |= __shfl_xor(f, 2);
    f = (f & 2) ? 2 : f;  // (1 | 2 = 3)
    if (q == 0) {
      J.flag[(int64_t)a * J.nb + b] = (unsigned char)f;
      J.flag_t[(int64_t)b * J.na + a] = (unsigned char)f;
    }
  }
}
// 64 x 64 tiles, 16-byte loads and 8-byte stores in both orientations (R % 4 == 0 and C % 4 == 0; 16-byte aligned
// bases): 34 MB move per bilinear forward, ~10 us with 4-byte accesses on 32 x 32 tiles.
__device__ __forceinline__ void cvt_tile_block(const CvtJob& J, int bx, int by, float (*tile)[65]) {
  const int64_t r0 = (int64_t)by * 64, c0 = (int64_t)bx * 64;
  if (r0 >= J.R || c0 >= J.C) return;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 column quads x 16 rows per pass
  f32x4 vq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {  // all four loads in flight before the first use
    const int64_t r = r0 + ty + 16 * q, c = c0 + 4 * tx;
    vq[q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (r < J.R && c < J.C) {
      if (J.in_bf16) {
        const bf16x4 b4 = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(J.in) + r * J.C + c);
        vq[q] = f32x4{(float)b4[0], (float)b4[1], (float)b4[2], (float)b4[3]};
      } else {
        vq[q] = *reinterpret_cast<const f32x4*>(J.in + r * J.C + c);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rl = ty + 16 * q;
    const int64_t r = r0 + rl, c = c0 + 4 * tx;
    f32x4 v = vq[q];
    if (r < J.R && c < J.C)
      for (int sl = 1; sl < J.n_slab; ++sl) v += *reinterpret_cast<const f32x4*>(J.in + sl * J.slab_stride + r * J.C + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[rl][4 * tx + e] = v[e];
    if ((J.out_rm || J.out_frag) && r < J.R && c < J.C) {
      const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      if (J.out_rm && !J.split_rm) *reinterpret_cast<bf16x4*>(J.out_rm + r * J.C + c) = o;
      if (J.out_rm && J.split_rm) {
        int64_t h0, h1, lo;
        split3_offsets(J.split_rm, J.C, h0, h1, lo);
        const bf16x4 l = {(bf16_t)bf16_residual(v[0]), (bf16_t)bf16_residual(v[1]), (bf16_t)bf16_residual(v[2]),
                          (bf16_t)bf16_residual(v[3])};
        bf16_t* row = J.out_rm + r * 3 * J.C + c;
        *reinterpret_cast<bf16x4*>(row + h0) = o;
        *reinterpret_cast<bf16x4*>(row + h1) = o;
        *reinterpret_cast<bf16x4*>(row + lo) = l;
      }
      if (J.out_frag) *reinterpret_cast<bf16x4*>(J.out_frag + frag_major_offset(r, c & ~(int64_t)7, J.C) + (c & 4)) = o;
    }
  }
  if (!J.out_t && !J.out_t_frag) return;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int cl = ty + 16 * q;
    const int64_t c = c0 + cl, r = r0 + 4 * tx;
    if (c < J.C && r < J.R) {
      const bf16x4 o = {(bf16_t)tile[4 * tx][cl], (bf16_t)tile[4 * tx + 1][cl], (bf16_t)tile[4 * tx + 2][cl],
                        (bf16_t)tile[4 * tx + 3][cl]};
      if (J.out_t_frag) *reinterpret_cast<bf16x4*>(J.out_t_frag + frag_major_offset(c, r & ~(int64_t)7, J.R) + (r & 4)) = o;
      if (!J.out_t) {
      } else if (!J.split_t) {
        *reinterpret_cast<bf16x4*>(J.out_t + c * J.R + r) = o;
      } else {
        int64_t h0, h1, lo;
        split3_offsets(J.split_t, J.R, h0, h1, lo);
        const bf16x4 l = {(bf16_t)bf16_residual(tile[4 * tx][cl]), (bf16_t)bf16_residual(tile[4 * tx + 1][cl]),
                          (bf16_t)bf16_residual(tile[4 * tx + 2][cl]), (bf16_t)bf16_residual(tile[4 * tx + 3][cl])};
        bf16_t* row = J.out_t + c * 3 * J.R + r;
        *reinterpret_cast<bf16x4*>(row + h0) = o;
        *reinterpret_cast<bf16x4*>(row + h1) = o;
        *reinterpret_cast<bf16x4*>(row + lo) = l;
      }
    }
  }
}
static __global__ __launch_bounds__(256) void cvt_transpose3_kernel(CvtJobs jobs) {
  kernarg_prefetch<(int)sizeof(CvtJobs)>();
  __shared__ float tile[64][65];
  if (blockIdx.z == 4) {  // the equal-id flags ride along: (row block, 64 column blocks) pairs over this slice's blocks
    const DupFlagJob& D = jobs.dup;
    const int chunks = (D.nb + 63) / 64, total = D.na * chunks;
    for (int w = (int)(blockIdx.y * gridDim.x + blockIdx.x); w < total; w += (int)(gridDim.x * gridDim.y)) {
      dup_flags_block(D, w / chunks, (w % chunks) * 64, reinterpret_cast<int64_t*>(&tile[0][0]));
      __syncthreads();
    }
    return;
  }
  cvt_tile_block(jobs.j[blockIdx.z], (int)blockIdx.x, (int)blockIdx.y, tile);
}
// any shape: 32 x 32 tiles, element accesses
static __global__ __launch_bounds__(256) void cvt_transpose3_generic_kernel(CvtJobs jobs) {
  __shared__ float tile[32][33];
  const CvtJob& J = jobs.j[blockIdx.z];
  const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
  if (r0 >= J.R || c0 >= J.C) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t r = r0 + ty + 8 * q, c = c0 + tx;
    float v = (r < J.R && c < J.C) ? J.in[r * J.C + c] : 0.0f;
    if (r < J.R && c < J.C)
      for (int sl = 1; sl < J.n_slab; ++sl) v += J.in[sl * J.slab_stride + r * J.C + c];
    tile[ty + 8 * q][tx] = v;
    if (J.out_rm && r < J.R && c < J.C) J.out_rm[r * J.C + c] = (bf16_t)v;
  }
  if (!J.out_t) return;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t c = c0 + ty + 8 * q, r = r0 + tx;
    if (c < J.C && r < J.R) J.out_t[c * J.R + r] = (bf16_t)tile[tx][ty + 8 * q];
  }
}
static inline int launch_cvt_transpose3(const CvtJobs& jobs, hipStream_t st, const char* what) {
  int64_t rmax = 0, cmax = 0;
  bool vec = true;
  for (int q = 0; q < 4; ++q) {
    const CvtJob& J = jobs.j[q];
    if (J.R > rmax) rmax = J.R;
    if (J.C > cmax) cmax = J.C;
    vec = vec && J.R % 4 == 0 && J.C % 4 == 0 && (uintptr_t)J.in % (J.in_bf16 ? 8 : 16) == 0 && (uintptr_t)J.out_rm % 8 == 0 &&
          (uintptr_t)J.out_t % 8 == 0 && J.slab_stride % 4 == 0 && (!J.out_frag || (J.R % 32 == 0 && J.C % 16 == 0)) &&
          (!J.out_t_frag || (J.C % 32 == 0 && J.R % 16 == 0));
  }
  for (int q = 0; q < 4; ++q)
    if (!vec && jobs.j[q].R > 0 && jobs.j[q].in_bf16) {
      set_error("%s: bf16 inputs need the vectorised conversion kernel (aligned, multiples of 4)", what);
      return MI_ESHAPE;
    }
  if (!vec && (jobs.dup.na > 0 || jobs.j[0].out_frag || jobs.j[1].out_frag || jobs.j[2].out_frag || jobs.j[3].out_frag ||
               jobs.j[0].out_t_frag || jobs.j[1].out_t_frag || jobs.j[2].out_t_frag || jobs.j[3].out_t_frag)) {
    set_error("%s: fragment-major outputs / id flags need the vectorised conversion kernel (aligned, multiples of 4)", what);
    return MI_ESHAPE;
  }
  {
    ProfScope prof_(what, st);
    if (vec) {
      dim3 grid((unsigned)((cmax + 63) / 64), (unsigned)((rmax + 63) / 64), jobs.dup.na > 0 ? 5 : 4);
      hipLaunchKernelGGL(cvt_transpose3_kernel, grid, dim3(256), 0, st, jobs);
    } else {
      dim3 grid((unsigned)((cmax + 31) / 32), (unsigned)((rmax + 31) / 32), 4);
      hipLaunchKernelGGL(cvt_transpose3_generic_kernel, grid, dim3(256), 0, st, jobs);
    }
  }
  MI_LAUNCH_CHECK(what);
  return MI_OK;
}

// ------------------------------------------------------------------------------------------------ prep + T, one launch
// The bilinear forward used to open with two short launches, the second waiting for the first: the fp32 -> bf16
// conversions (9 us) and T = X W on 128 tiles (10 us with half the chip idle); a launch of this size costs ~4 us in
// ramp and drain alone.  Here both are ONE grid: blocks [0, n_t) compute a 128 x 128 tile of T straight from the fp32
// operands (converted on the way into LDS, both as they lie: X rows are K-contiguous for ds_read_b128, the [k][n] image
// of W feeds the B fragments through the transposing ds_read_b64_tr_b16), all other blocks do the conversions nobody in
// this launch depends on (X^T and W for the backward's products, Y row-major and fragment-major, the equal-id flags) on
// the CUs the tiles leave free.  T is rounded exactly as before: bf16 operands, fp32 accumulation, one rounding to bf16.
struct PrepTArgs {
  const float* x;    // [m][k]
  const float* w;    // [k][n]
  int64_t m, n, k;   // k % 64 == 0, n % 128 == 0
  bf16_t* tb;        // [m][n]
  bf16_t* tfb;       // fragment-major copy (EpiOut::bf_frag) or null
  bf16_t* ttb;       // T^T [n][m] or null (the G-materialising path reads it as the B operand of dY = G^T T; m % 8 == 0)
  int n_t;           // blocks of the T role (tiles, padded to 8 x tiles-per-panel x ceil(panels / 8))
  int job_begin[6];  // conversion blocks (counted from n_t): first block of job q; [4] = flags, [5] = end
  int job_nx[4];     // 64-column tiles per tile row of job q
  CvtJobs jobs;
  int x_bf16;        // x points to bf16 data [m][k] (the bf16 boundary): the tile goes to LDS as it lies
  // an optional SECOND product in the same launch (separable critic: A = X Wg and C = Y Wh): blocks [n_t, n_t + n_t2)
  const float* x2;
  const float* w2;
  int64_t m2, n2, k2;
  bf16_t* tb2;
  bf16_t* tfb2;
  int n_t2;
};
struct PrepTSecond {
  const float* x;
  const float* w;
  int64_t m, n, k;
  bf16_t* tb;
  bf16_t* tfb;
};
constexpr int kPrepTLdB = 160;  // W image row pitch in bf16 elements (128 columns + 32 of padding = 320 bytes)
constexpr size_t kPrepTSmem = (2 * kTile * kG2LD + 2 * kG2KT * kPrepTLdB) * sizeof(bf16_t);  // 77,824 bytes

// XB16: x arrives as bf16 (a compile-time variant: as a run-time branch the second staging path cost the fp32 kernel 10
// registers and 48 bytes of scratch)
template <bool XB16>
static __global__ __launch_bounds__(256, 2) void bilinear_prep_t_kernel(PrepTArgs a) {
  kernarg_prefetch<(int)sizeof(PrepTArgs)>();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int blk = (int)blockIdx.x;
  MI_STAMP(0);
  if (blk >= a.n_t + a.n_t2) {  // ---- conversion roles
    float(*tile)[65] = reinterpret_cast<float(*)[65]>(smem_raw);
    const int c = blk - a.n_t - a.n_t2;
    if (c >= a.job_begin[4]) {
      const DupFlagJob& D = a.jobs.dup;
      const int chunks = (D.nb + 63) / 64, w = c - a.job_begin[4];
      if (w < D.na * chunks) dup_flags_block(D, w / chunks, (w % chunks) * 64, reinterpret_cast<int64_t*>(smem_raw));
      return;
    }
    int q = 0;
    while (q < 3 && c >= a.job_begin[q + 1]) ++q;
    const int t = c - a.job_begin[q];
    cvt_tile_block(a.jobs.j[q], t % a.job_nx[q], t / a.job_nx[q], tile);
    MI_STAMP(5);
    return;
  }
  // ---- a 128 x 128 tile of T.  What bounds this role is bytes through the fabric (the fp32 operands are twice the
  // bf16 copies the plain GEMM reads, every X panel is wanted by n / 128 tiles): the tiles of one row panel run back to
  // back on ONE XCD (xcd_decode), so its L2 fetches the panel once and keeps W.  (A 128 x 64 tiling with a 128-deep
  // k-step -- twice the tiles, half the round trips -- moved 96 MB instead of 64 and took 22.9 us instead of 16.4.)
  bf16_t* As = reinterpret_cast<bf16_t*>(smem_raw);  // [2][128 rows][72]: X as it lies, K-contiguous
  bf16_t* Bs = As + 2 * kTile * kG2LD;               // [2][64 k][160]: W as it lies, n-contiguous; the B fragments come
                                                     // out K-major through ds_read_b64_tr_b16 (320-byte rows: the four
                                                     // rows of a transposed read fall on disjoint quarters of the banks)
  // which product (the second one's blocks follow the first one's; both ranges are multiples of 8 long)
  const bool second = blk >= a.n_t;
  const float* const px = second ? a.x2 : a.x;
  const float* const pw = second ? a.w2 : a.w;
  const int64_t pm = second ? a.m2 : a.m, pn = second ? a.n2 : a.n, pk = second ? a.k2 : a.k;
  bf16_t* const ptb = second ? a.tb2 : a.tb;
  bf16_t* const ptfb = second ? a.tfb2 : a.tfb;
  int nb_, mb_;
  if (!xcd_decode((int)(pn / kTile), (int)((pm + kTile - 1) / kTile), nb_, mb_, 0, second ? blk - a.n_t : blk)) return;
  const int64_t m0 = (int64_t)mb_ * kTile, n0 = (int64_t)nb_ * kTile;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, half = lane >> 5;
  const int xr = tid >> 4, xc = tid & 15;  // X staging: rows xr + 16 q, float4 number xc of the 64-wide k slice
  const int wk = tid >> 5, wc = tid & 31;  // W staging: k rows wk + 8 q of the slice, float4 number wc of the 128 columns
  // transposed read (guide T10): lane 4 q + p of a 16-lane group addresses row q, columns 4 p .. 4 p + 3 of a 4 x 16 block
  // and receives column (lane & 15) of its four rows; a lane's B fragment is rows 8 (lane >> 5) .. + 7 of column lane & 31
  const int tr_off = (8 * half + ((lane >> 2) & 3)) * (kPrepTLdB * 2) + (wn * 64 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  f32x4 rx[8], rw[8];
  const float* xp[8];
  if constexpr (!XB16) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      int64_t row = m0 + xr + 16 * q;
      if (row >= pm) row = pm - 1;  // clamped rows only feed outputs the epilogue drops
      xp[q] = px + row * pk + 4 * xc;
    }
  }
  const float* wp = pw + (int64_t)wk * pn + n0 + 4 * wc;
  // bf16 x: a 128 x 64 tile is 1024 chunks of 16 bytes, four per thread: rows (tid >> 3) + 32 q, chunk tid & 7; they
  // travel in rx[0..3] as raw bits
  const bf16_t* xbp[4];
  const int br_ = tid >> 3, bc_ = tid & 7;
  if constexpr (XB16) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int64_t row = m0 + br_ + 32 * q;
      if (row >= pm) row = pm - 1;
      xbp[q] = reinterpret_cast<const bf16_t*>(px) + row * pk + 8 * bc_;
    }
  }
  auto load_tile = [&](int64_t k0) {
    if constexpr (XB16) {
#pragma unroll
      for (int q = 0; q < 4; ++q) rx[q] = *reinterpret_cast<const f32x4*>(xbp[q] + k0);
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) rx[q] = *reinterpret_cast<const f32x4*>(xp[q] + k0);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) rw[q] = *reinterpret_cast<const f32x4*>(wp + (k0 + 8 * q) * pn);
  };
  auto store_tile = [&](int buf) {
    if constexpr (XB16) {
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(&As[(buf * kTile + br_ + 32 * q) * kG2LD + 8 * bc_]) = rx[q];
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const bf16x4 o = {(bf16_t)rx[q][0], (bf16_t)rx[q][1], (bf16_t)rx[q][2], (bf16_t)rx[q][3]};
        *reinterpret_cast<bf16x4*>(&As[(buf * kTile + xr + 16 * q) * kG2LD + 4 * xc]) = o;
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const bf16x4 o = {(bf16_t)rw[q][0], (bf16_t)rw[q][1], (bf16_t)rw[q][2], (bf16_t)rw[q][3]};
      *reinterpret_cast<bf16x4*>(&Bs[(buf * kG2KT + wk + 8 * q) * kPrepTLdB + 4 * wc]) = o;
    }
  };

  const int nt = (int)(pk / kG2KT);
  load_tile(0);
  store_tile(0);
  __syncthreads();
  MI_STAMP(1);
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    const bool more = t + 1 < nt;
    if (more) load_tile((int64_t)(t + 1) * kG2KT);
    const bf16_t* at = As + buf * kTile * kG2LD;
    const char* bt = reinterpret_cast<const char*>(Bs + buf * kG2KT * kPrepTLdB) + tr_off;
#pragma unroll
    for (int kk = 0; kk < kG2KT / 16; ++kk) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
        af[tm] = *reinterpret_cast<const bf16x8*>(&at[(wm * 64 + tm * 32 + r32) * kG2LD + kk * 16 + 8 * half]);
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        using lds_b4 = __attribute__((address_space(3))) bf16x4;
        const char* pb = bt + kk * 16 * (kPrepTLdB * 2) + tn * 64;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(pb));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(pb + 4 * (kPrepTLdB * 2)));
        bfr[tn] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm], bfr[tn], acc[tm][tn], 0, 0, 0);
    }
    if (t == 0) MI_STAMP(2);
    if (more) store_tile(buf ^ 1);
    __syncthreads();
    if (t == 0) MI_STAMP(3);
  }
  MI_STAMP(4);
  wave_tile_store_bf16(acc, smem_raw + wave * kEpiLdsPerWave, ptb, pn, second ? nullptr : a.ttb, pm, m0 + wm * 64,
                       n0 + wn * 64, pm, pn, ptfb);
  MI_STAMP(5);
}

// conversions: up to four CvtJob (64 x 64 tiles each) + the flag job; returns MI_EINVAL for shapes the fused kernel does
// not take (the caller then runs the conversion launch and the plain GEMM)
static inline int launch_prep_t(const float* x, const float* w, int64_t m, int64_t n, int64_t k, bf16_t* tb, bf16_t* tfb,
                                const CvtJobs& jobs, hipStream_t st, const char* what, bool x_bf16 = false,
                                const PrepTSecond* second = nullptr, bf16_t* ttb = nullptr) {
  static const bool off = getenv("MI_NO_PREP_T") != nullptr;  // A/B switch: separate conversion and GEMM launches
  if (off || k % 64 != 0 || n % kTile != 0 || m < 1 || (uintptr_t)x % 16 != 0 || (uintptr_t)w % 16 != 0 ||
      (uintptr_t)tb % 16 != 0 || (tfb && (m % 32 != 0 || (uintptr_t)tfb % 16 != 0)))
    return MI_EINVAL;
  PrepTArgs a{};
  a.x = x; a.w = w; a.m = m; a.n = n; a.k = k; a.tb = tb; a.tfb = tfb;
  a.x_bf16 = x_bf16 ? 1 : 0;
  if (ttb && (m % 8 != 0 || (uintptr_t)ttb % 16 != 0)) return MI_EINVAL;
  a.ttb = ttb;
  a.n_t = (int)(8 * (n / kTile) * (((m + kTile - 1) / kTile + 7) / 8));  // padded for the XCD mapping
  if (second) {
    const PrepTSecond& q = *second;
    if (x_bf16 || q.k % 64 != 0 || q.n % kTile != 0 || q.m < 1 || (uintptr_t)q.x % 16 != 0 || (uintptr_t)q.w % 16 != 0 ||
        (uintptr_t)q.tb % 16 != 0 || (q.tfb && (q.m % 32 != 0 || (uintptr_t)q.tfb % 16 != 0)))
      return MI_EINVAL;
    a.x2 = q.x; a.w2 = q.w; a.m2 = q.m; a.n2 = q.n; a.k2 = q.k; a.tb2 = q.tb; a.tfb2 = q.tfb;
    a.n_t2 = (int)(8 * (q.n / kTile) * (((q.m + kTile - 1) / kTile + 7) / 8));
  }
  a.jobs = jobs;
  int total = 0;
  for (int q = 0; q < 4; ++q) {
    const CvtJob& J = jobs.j[q];
    if (J.R > 0 && (J.R % 4 != 0 || J.C % 4 != 0 || (uintptr_t)J.in % (J.in_bf16 ? 8 : 16) != 0 || (uintptr_t)J.out_rm % 8 != 0 ||
                    (uintptr_t)J.out_t % 8 != 0 || J.n_slab > 1 || (J.out_frag && (J.R % 32 != 0 || J.C % 16 != 0)) ||
                    (J.out_t_frag && (J.C % 32 != 0 || J.R % 16 != 0))))
      return MI_EINVAL;
    a.job_begin[q] = total;
    a.job_nx[q] = J.R > 0 ? (int)((J.C + 63) / 64) : 1;
    if (J.R > 0) total += a.job_nx[q] * (int)((J.R + 63) / 64);
  }
  a.job_begin[4] = total;
  if (jobs.dup.na > 0) total += jobs.dup.na * ((jobs.dup.nb + 63) / 64);
  a.job_begin[5] = total;
#ifdef MI_STAMPS
  stamp_select(what, st);
#endif
  if (x_bf16) {
    MI_SET_DYN_SMEM(bilinear_prep_t_kernel<true>, kPrepTSmem, "hipFuncSetAttribute(bilinear_prep_t_kernel)");
    ProfScope prof_(what, st);
    hipLaunchKernelGGL(bilinear_prep_t_kernel<true>, dim3((unsigned)(a.n_t + a.n_t2 + total)), dim3(256), kPrepTSmem, st, a);
  } else {
    MI_SET_DYN_SMEM(bilinear_prep_t_kernel<false>, kPrepTSmem, "hipFuncSetAttribute(bilinear_prep_t_kernel)");
    ProfScope prof_(what, st);
    hipLaunchKernelGGL(bilinear_prep_t_kernel<false>, dim3((unsigned)(a.n_t + a.n_t2 + total)), dim3(256), kPrepTSmem, st, a);
  }
  MI_LAUNCH_CHECK(what);
  return MI_OK;
}

// out[r][c] (ld = ldo) = sum_s slab[s][r][c]   (fixed order).  One float4 per thread when the shapes allow (the
// slabs' loads are independent, only the adds are ordered), grid-stride scalar loop otherwise.
static __global__ __launch_bounds__(256) void slab_reduce_ld_kernel(const float* __restrict__ slab, int n_slab,
                                                                    int64_t rows, int64_t cols,
                                                                    float* __restrict__ out, int64_t ldo, int vec) {
  const int64_t total = rows * cols;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (vec) {
    for (int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; e < total; e += stride * 4) {
      f32x4 a = *reinterpret_cast<const f32x4*>(slab + e);
      int s = 1;
      for (; s + 3 < n_slab; s += 4) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(slab + (int64_t)s * total + e);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(slab + (int64_t)(s + 1) * total + e);
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(slab + (int64_t)(s + 2) * total + e);
        const f32x4 b3 = *reinterpret_cast<const f32x4*>(slab + (int64_t)(s + 3) * total + e);
        a += b0;
        a += b1;
        a += b2;
        a += b3;
      }
      for (; s < n_slab; ++s) a += *reinterpret_cast<const f32x4*>(slab + (int64_t)s * total + e);
      *reinterpret_cast<f32x4*>(out + (e / cols) * ldo + (e % cols)) = a;
    }
    return;
  }
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    float a = slab[e];
    for (int s = 1; s < n_slab; ++s) a += slab[(int64_t)s * total + e];
    out[(e / cols) * ldo + (e % cols)] = a;
  }
}

static inline int launch_slab_reduce_ld(const float* slab, int n_slab, int64_t rows, int64_t cols, float* out,
                                        int64_t ldo, hipStream_t st, const char* what) {
  const int64_t total = rows * cols;
  if (total <= 0) return MI_OK;
  const int vec = (cols % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)slab % 16 == 0) && ((uintptr_t)out % 16 == 0)) ? 1 : 0;
  int64_t work = vec ? total / 4 : total;
  int64_t blocks = (work + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  {
    ProfScope prof_(what, st);
    hipLaunchKernelGGL(slab_reduce_ld_kernel, dim3((unsigned)blocks), dim3(256), 0, st, slab, n_slab, rows, cols, out,
                       ldo, vec);
  }
  MI_LAUNCH_CHECK(what);
  return MI_OK;
}

}  // namespace mi
