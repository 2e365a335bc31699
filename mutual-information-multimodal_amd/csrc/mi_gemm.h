// LDS-tiled MFMA GEMM used for the dense contractions of the critic path that are plain matrix products:
//   C[m, n] = sum_k A(m, k) * B(n, k)        ("NT" canonical form; either operand may be stored transposed)
// with a pluggable epilogue (store / bias, fused log-sum-exp partials, fused d loss / d score).
//
// Two arithmetic modes (template OpT):
//   float   -> v_mfma_f32_32x32x2_f32 : exact fp32 products, fp32 accumulate (parity mode, 1/16 of the bf16 rate)
//   bf16_t  -> v_mfma_f32_32x32x16_bf16: operands rounded to bf16 (RNE) when staged into LDS, fp32 accumulate
//
// Geometry: 128 x 128 output tile per 256-thread workgroup (4 waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles),
// K tile 32 (bf16) / 16 (f32), global -> registers -> LDS staging with the next tile's loads issued before the
// current tile's MFMAs (guide T14), padded LDS rows (bf16: 80-byte rows -> ds_read_b128 conflict-free;
// f32: 17-float rows -> ds_read_b32 conflict-free).
#pragma once
#include "mi_common.h"

namespace mi {

// Operand element (o, k) lives at p[o * so + k * sk].  mode: 0 = k-contiguous 4-wide loads, 1 = outer-contiguous
// 4-wide loads (transposed while staging), 2 = scalar loads.
template <typename T>
struct Operand {
  const T* p;
  int64_t so, sk;
  int mode;
};

template <typename T>
static inline Operand<T> make_operand(const T* p, int64_t so, int64_t sk) {
  Operand<T> o{p, so, sk, 2};
  const bool aligned = (((uintptr_t)p) % (4 * sizeof(T))) == 0;
  if (sk == 1 && so % 4 == 0 && aligned) o.mode = 0;
  else if (so == 1 && sk % 4 == 0 && aligned) o.mode = 1;
  return o;
}

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }

template <typename OpT>
struct GemmCfg;
template <>
struct GemmCfg<float> {
  static constexpr int KT = 16, LD = 17, KSTEP = 2;
};
template <>
struct GemmCfg<bf16_t> {
  static constexpr int KT = 32, LD = 40, KSTEP = 16;
};

constexpr int kTile = 128;

// Registers holding one staged 128 x KT operand tile for this thread (KT/2 elements).
template <int KT>
struct StageRegs {
  float v[KT / 2];
};

template <typename TIn, int KT>
__device__ __forceinline__ void stage_load(const Operand<TIn>& op, int64_t o0, int64_t k0, int64_t n_outer, int64_t n_k,
                                           StageRegs<KT>& r) {
  const int t = threadIdx.x;
  if (op.mode == 1) {
    // 32 vectors of 4 along the outer dim per k; 8 k per pass
    constexpr int PASSES = KT / 8;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int64_t k = k0 + p * 8 + (t >> 5);
      const int64_t o = o0 + (t & 31) * 4;
      if (k < n_k && o + 3 < n_outer) {
        const TIn* src = op.p + o + k * op.sk;
        if constexpr (sizeof(TIn) == 4) {
          const f32x4 x = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
          for (int q = 0; q < 4; ++q) r.v[p * 4 + q] = x[q];
        } else {
          const bf16x4 x = *reinterpret_cast<const bf16x4*>(src);
#pragma unroll
          for (int q = 0; q < 4; ++q) r.v[p * 4 + q] = (float)x[q];
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          r.v[p * 4 + q] = (k < n_k && o + q < n_outer) ? to_f32(op.p[(o + q) + k * op.sk]) : 0.0f;
      }
    }
  } else {
    constexpr int KVEC = KT / 4;          // vectors per row
    constexpr int RPP = 256 / KVEC;       // rows per pass
    constexpr int PASSES = kTile / RPP;   // == KT / 8
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int64_t o = o0 + p * RPP + t / KVEC;
      const int64_t k = k0 + (t % KVEC) * 4;
      if (op.mode == 0 && o < n_outer && k + 3 < n_k) {
        const TIn* src = op.p + o * op.so + k;
        if constexpr (sizeof(TIn) == 4) {
          const f32x4 x = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
          for (int q = 0; q < 4; ++q) r.v[p * 4 + q] = x[q];
        } else {
          const bf16x4 x = *reinterpret_cast<const bf16x4*>(src);
#pragma unroll
          for (int q = 0; q < 4; ++q) r.v[p * 4 + q] = (float)x[q];
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          r.v[p * 4 + q] = (o < n_outer && k + q < n_k) ? to_f32(op.p[o * op.so + (k + q) * op.sk]) : 0.0f;
      }
    }
  }
}

template <typename OpT, int KT, int LD>
__device__ __forceinline__ void stage_store(int mode, const StageRegs<KT>& r, OpT* tile /* [128][LD] */) {
  const int t = threadIdx.x;
  if (mode == 1) {
    constexpr int PASSES = KT / 8;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int k = p * 8 + (t >> 5);
      const int o = (t & 31) * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) tile[(o + q) * LD + k] = (OpT)r.v[p * 4 + q];
    }
  } else {
    constexpr int KVEC = KT / 4;
    constexpr int RPP = 256 / KVEC;
    constexpr int PASSES = kTile / RPP;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int o = p * RPP + t / KVEC;
      const int k = (t % KVEC) * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) tile[o * LD + k + q] = (OpT)r.v[p * 4 + q];
    }
  }
}

// Accumulator element -> (row, col) of the 64 x 64 wave tile (MFMA 32x32 C/D layout, guide section 3):
//   col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
template <class F>
__device__ __forceinline__ void foreach_acc(f32x16 (&acc)[2][2], int64_t m_base, int64_t n_base, F&& f) {
  const int lane = threadIdx.x & 63;
  const int col_l = lane & 31, half = lane >> 5;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m_base + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int64_t col = n_base + tn * 32 + col_l;
        f(row, col, acc[tm][tn][r]);
      }
}

template <typename OpT, typename TA, typename TB, class Epi>
__global__ __launch_bounds__(256) void gemm_nt_kernel(Operand<TA> A, Operand<TB> B, int64_t M, int64_t N, int64_t K,
                                                      int64_t k_chunk, Epi epi) {
  using Cfg = GemmCfg<OpT>;
  constexpr int KT = Cfg::KT, LD = Cfg::LD, KSTEP = Cfg::KSTEP;
  __shared__ __attribute__((aligned(16))) OpT As[kTile * LD];
  __shared__ __attribute__((aligned(16))) OpT Bs[kTile * LD];

  const int64_t m0 = (int64_t)blockIdx.y * kTile, n0 = (int64_t)blockIdx.x * kTile;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, half = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  StageRegs<KT> ra, rb;
  // split-K: blockIdx.z owns k in [kbeg, kend); the epilogue writes slab blockIdx.z
  const int64_t kbeg = (int64_t)blockIdx.z * k_chunk;
  const int64_t kend = (kbeg + k_chunk < K) ? kbeg + k_chunk : K;
  const int64_t nt = (kend - kbeg + KT - 1) / KT;
  if (nt > 0) {
    stage_load<TA, KT>(A, m0, kbeg, M, kend, ra);
    stage_load<TB, KT>(B, n0, kbeg, N, kend, rb);
    stage_store<OpT, KT, LD>(A.mode, ra, As);
    stage_store<OpT, KT, LD>(B.mode, rb, Bs);
  }
  __syncthreads();
  for (int64_t t = 0; t < nt; ++t) {
    const bool more = t + 1 < nt;
    if (more) {
      stage_load<TA, KT>(A, m0, kbeg + (t + 1) * KT, M, kend, ra);
      stage_load<TB, KT>(B, n0, kbeg + (t + 1) * KT, N, kend, rb);
    }
#pragma unroll
    for (int kk = 0; kk < KT / KSTEP; ++kk) {
      if constexpr (sizeof(OpT) == 2) {
        bf16x8 af[2], bfr[2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
          af[tm] = *reinterpret_cast<const bf16x8*>(&As[(wm * 64 + tm * 32 + r32) * LD + kk * 16 + 8 * half]);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          bfr[tn] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * 64 + tn * 32 + r32) * LD + kk * 16 + 8 * half]);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm], bfr[tn], acc[tm][tn], 0, 0, 0);
      } else {
        float af[2], bfr[2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) af[tm] = As[(wm * 64 + tm * 32 + r32) * LD + kk * 2 + half];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) bfr[tn] = Bs[(wn * 64 + tn * 32 + r32) * LD + kk * 2 + half];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm], bfr[tn], acc[tm][tn], 0, 0, 0);
      }
    }
    __syncthreads();
    if (more) {
      stage_store<OpT, KT, LD>(A.mode, ra, As);
      stage_store<OpT, KT, LD>(B.mode, rb, Bs);
    }
    __syncthreads();
  }
  epi(acc, m0 + wm * 64, n0 + wn * 64, M, N);
}

// ------------------------------------------------------------------------------------------------ epilogues
// C[row, col] = alpha * acc (+ bias[col]); optional accumulate into C.
struct EpiStore {
  float* c;
  int64_t ldc;
  const float* bias;  // indexed by col, may be null
  float alpha;
  int accumulate;
  int64_t slab_stride = 0;  // split-K: slab z at c + z * slab_stride
  __device__ __forceinline__ void operator()(f32x16 (&acc)[2][2], int64_t mb, int64_t nb, int64_t M, int64_t N) const {
    float* cz = c + (int64_t)blockIdx.z * slab_stride;
    foreach_acc(acc, mb, nb, [&](int64_t row, int64_t col, float v) {
      if (row < M && col < N) {
        float o = alpha * v + (bias ? bias[col] : 0.0f);
        float* dst = cz + row * ldc + col;
        if (accumulate) o += *dst;
        *dst = o;
      }
    });
  }
};

// Scores tile -> masked log-sum-exp partial (one Partial per workgroup), optional score write.
struct EpiScoreLse {
  const int64_t* sid_rows;
  const int64_t* sid_cols;
  int64_t row_offset;
  float* scores;  // [M, N] or null
  Partial* partials;
  __device__ __forceinline__ void operator()(f32x16 (&acc)[2][2], int64_t mb, int64_t nb, int64_t M, int64_t N) const {
    __shared__ Partial scratch[4];
    Partial p{MI_NEG_INF, 0.0f, 0.0f, 0u};
    foreach_acc(acc, mb, nb, [&](int64_t row, int64_t col, float v) {
      if (row < M && col < N) {
        if (scores) scores[row * N + col] = v;
        const int kind = pair_kind(row_offset + row, col, sid_rows[row], sid_cols[col]);
        if (kind == 1) {
          p.pos += v;
        } else if (kind == 2) {
          lse_push(p.m, p.s, v);
          p.cnt += 1;
        }
      }
    });
    p = block_reduce_partial<4>(p, scratch);
    if (threadIdx.x == 0) partials[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = p;
  }
};

// Scores tile -> G = grad_out * d loss / d S, written as TG (float or bf16).
template <typename TG>
struct EpiGradScore {
  const int64_t* sid_rows;
  const int64_t* sid_cols;
  int64_t row_offset;
  const mi_stats* stats;
  const float* grad_out;
  TG* g;  // [M, N]
  __device__ __forceinline__ void operator()(f32x16 (&acc)[2][2], int64_t mb, int64_t nb, int64_t M, int64_t N) const {
    const float go = grad_out ? grad_out[0] : 1.0f;
    const float lse = stats->lse;
    const float gpos = -go / (float)stats->n_pos;
    foreach_acc(acc, mb, nb, [&](int64_t row, int64_t col, float v) {
      if (row < M && col < N) {
        const int kind = pair_kind(row_offset + row, col, sid_rows[row], sid_cols[col]);
        float gv = 0.0f;
        if (kind == 1) gv = gpos;
        else if (kind == 2) gv = go * expf(v - lse);
        g[row * N + col] = (TG)gv;
      }
    });
  }
};

template <typename OpT, typename TA, typename TB, class Epi>
static inline int launch_gemm(const Operand<TA>& A, const Operand<TB>& B, int64_t M, int64_t N, int64_t K,
                              const Epi& epi, hipStream_t st, const char* what, int n_splits = 1, int64_t k_chunk = 0) {
  if (M <= 0 || N <= 0) return MI_OK;
  dim3 grid((unsigned)((N + kTile - 1) / kTile), (unsigned)((M + kTile - 1) / kTile), (unsigned)n_splits);
  if (n_splits <= 1 || k_chunk <= 0) k_chunk = K > 0 ? K : 1;
  {
    ProfScope prof_(what, st);
    hipLaunchKernelGGL((gemm_nt_kernel<OpT, TA, TB, Epi>), grid, dim3(256), 0, st, A, B, M, N, K, k_chunk, epi);
  }
  MI_LAUNCH_CHECK(what);
  return MI_OK;
}

}  // namespace mi
