// fp8 mode of the bilinear critic (BASELINE.json configs[4]: "fp8 embeddings on CDNA4 fp8 MFMA critic"; SURVEY.md 8d
// config 5: e4m3, per-tensor scale = absmax / 448, fp32 accumulate; hazard H7: parity against an oracle fed identically
// quantised inputs -- oracle/mi_oracle.py bilinear_step_fp8).  The reference has no such path (it is fp32 throughout).
//
//   x_q = e4m3(x / s_x), s_x = absmax(x) / 448            (same for y, W; scales and divisions in fp32, on the device)
//   T   = s_x s_w (x_q W_q)        v_mfma_scale_f32_32x32x64_f8f6f4, unit block scales: exact products, fp32 accumulation
//   t_q = e4m3(T / s_t)            its own per-tensor scale (the amax rides on the T product's epilogue)
//   S   = s_t s_y (t_q y_q^T)      fp8 MFMA again; masked log-sum-exp epilogue as in the bf16 path
// The backward treats the quantisers as straight-through and runs on the bf16 kernels with the quantised VALUES as
// operands (every e4m3 value is exact in bf16) and the scales as output factors: G = dL/dS from recomputed fp8 score
// tiles, dT = s_y (G y_q), dY = s_t (G^T t_q), dW = s_x (x_q^T dT), dX = s_w (dT W_q^T).
//
// OCP e4m3fn (gfx950), not MI300's fnuz.
#pragma once
#include "mi_gemm_bf16.h"

namespace mi {

typedef unsigned char fp8_t;  // e4m3fn bits
constexpr float kE4m3Max = 448.0f;

// device-side scale block: absmax bit patterns (non-negative floats order like unsigned integers) and the scales made
// from them.  Index: 0 x, 1 y, 2 w, 3 T.
struct Fp8Scales {
  unsigned amax_bits[4];
  float scale[4];
};

__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// absmax of up to three tensors (blockIdx.y selects one): float4 grid-stride loop with four loads in flight per thread,
// one atomicMax per workgroup (256 workgroups per tensor: more of them only queue up on the one address)
struct AbsmaxJobs {
  const float* in[3];
  int64_t n[3];  // elements (multiples of 4; 16-byte aligned bases)
  Fp8Scales* sc;
  int slot[3];
};
constexpr int kAbsmaxBlocks = 256;
static __global__ __launch_bounds__(256) void fp8_absmax_kernel(AbsmaxJobs J) {
  const int q = blockIdx.y;
  const int64_t n4 = J.n[q] / 4, stride = (int64_t)gridDim.x * 256;
  const f32x4* p = reinterpret_cast<const f32x4*>(J.in[q]);
  float m = 0.0f;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += 4 * stride) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = e + u * stride < n4 ? p[e + u * stride] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < 4; ++u)
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u][0]), fabsf(v[u][1]))), fmaxf(fabsf(v[u][2]), fabsf(v[u][3])));
  }
  m = wave_max_f(m);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    atomicMax(&J.sc->amax_bits[J.slot[q]], __float_as_uint(m));
  }
}

__device__ __forceinline__ float fp8_scale_of(unsigned amax_bits) {
  const float a = __uint_as_float(amax_bits);
  return a > 0.0f ? a / kE4m3Max : 1.0f;  // an all-zero tensor quantises to zeros under any scale
}
// quantise one value: the e4m3 VALUE of v / s as a float (exact) -- hardware conversion, round to nearest even
__device__ __forceinline__ float fp8_round(float v, float s) {
  const float r = fminf(fmaxf(v / s, -kE4m3Max), kE4m3Max);
  return __builtin_amdgcn_cvt_f32_fp8(__builtin_amdgcn_cvt_pk_fp8_f32(r, r, 0, false), 0);
}
__device__ __forceinline__ unsigned fp8_pack4(float a, float b, float c, float d) {  // values already on the e4m3 grid
  unsigned w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  return __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
}

// quantise up to four fp32 tensors (64 x 64 tiles; blockIdx.z selects the job): fp8 and / or bf16 copies of the QUANTISED
// values, row-major and / or transposed; thread 0 of block (0, 0) of each job publishes the scale.
struct QuantJob {
  const float* in;  // [R][C], R % 4 == 0, C % 4 == 0
  int64_t R, C;
  int slot;         // scale slot
  fp8_t* q_rm;      // [R][C] or null
  fp8_t* q_t;       // [C][R] or null
  bf16_t* b_rm;     // [R][C] or null
  bf16_t* b_t;      // [C][R] or null
};
struct QuantJobs {
  QuantJob j[4];
  Fp8Scales* sc;
};
static __global__ __launch_bounds__(256) void fp8_quantize_kernel(QuantJobs jobs) {
  __shared__ float tile[64][65];
  const QuantJob& J = jobs.j[blockIdx.z];
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  if (r0 >= J.R || c0 >= J.C) return;
  const float s = fp8_scale_of(jobs.sc->amax_bits[J.slot]);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) jobs.sc->scale[J.slot] = s;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rl = ty + 16 * q;
    const int64_t r = r0 + rl, c = c0 + 4 * tx;
    float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (r < J.R && c < J.C) {
      const f32x4 in = *reinterpret_cast<const f32x4*>(J.in + r * J.C + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fp8_round(in[e], s);
      if (J.q_rm) *reinterpret_cast<unsigned*>(J.q_rm + r * J.C + c) = fp8_pack4(v[0], v[1], v[2], v[3]);
      if (J.b_rm) *reinterpret_cast<bf16x4*>(J.b_rm + r * J.C + c) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[rl][4 * tx + e] = v[e];
  }
  if (!J.q_t && !J.b_t) return;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int cl = ty + 16 * q;
    const int64_t c = c0 + cl, r = r0 + 4 * tx;
    if (c < J.C && r < J.R) {
      const float a0 = tile[4 * tx][cl], a1 = tile[4 * tx + 1][cl], a2 = tile[4 * tx + 2][cl], a3 = tile[4 * tx + 3][cl];
      if (J.q_t) *reinterpret_cast<unsigned*>(J.q_t + c * J.R + r) = fp8_pack4(a0, a1, a2, a3);
      if (J.b_t) *reinterpret_cast<bf16x4*>(J.b_t + c * J.R + r) = bf16x4{(bf16_t)a0, (bf16_t)a1, (bf16_t)a2, (bf16_t)a3};
    }
  }
}

// ------------------------------------------------------------------------------------------------ fp8 GEMM
// C[m][n] = sum_k A[m][k] B[n][k], both operands e4m3, K-contiguous.  128 x 128 tile, 4 waves of 64 x 64, k-step 128
// (128-byte rows), register-staged double buffer; LDS rows are 144 bytes apart.
constexpr int kF8KT = 128;  // k per tile (bytes per row)
constexpr int kF8LD = 144;  // LDS row pitch in bytes: 16-byte aligned rows, 36 dwords = 4 x odd -> the 16-lane groups of a
                            // ds_read_b128 over rows r32 fall on 16 distinct 4-bank spans
typedef int i32x8 __attribute__((ext_vector_type(8)));
// The products run on v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (E8M0 127 = 2^0): K = 64 per instruction
// at twice the rate of v_mfma_f32_32x32x16_fp8_fp8 (the dense fp8 peak of the chip, 5 PFLOP/s).  A lane holds 32
// consecutive K bytes of its row: bytes [32 half, 32 half + 32) of the 64-deep step, for the A and the B operand alike --
// whatever order the instruction gives the bytes inside a lane, the two operands agree on it, which is all a dot product
// over K needs; the accumulator layout is the 32 x 32 one of the bf16 instruction, so every epilogue still applies.
constexpr int kF8UnitScale = 0x7F7F7F7F;
constexpr size_t kF8Smem = 2 * 2 * kTile * kF8LD;  // 73,728 bytes (>= 4 epilogue staging areas)

struct GemmF8Args {
  const fp8_t* a;
  int64_t lda;
  const fp8_t* b;
  int64_t ldb;
  int64_t m, n, k;  // k % 16 == 0
};

template <class Epi>
__global__ __launch_bounds__(256, 2) void gemm_fp8_kernel(GemmF8Args P, Epi epi) {
  kernarg_prefetch<(int)(sizeof(GemmF8Args) + sizeof(Epi))>();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  char* As = smem_raw;                       // [2][128][144]
  char* Bs = smem_raw + 2 * kTile * kF8LD;   // [2][128][144]
  int bx_, by_;
  xcd_tile(bx_, by_);
  const int64_t m0 = (int64_t)by_ * kTile, n0 = (int64_t)bx_ * kTile;
  if (m0 >= P.m || n0 >= P.n) return;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, half = lane >> 5;
  const int srow = tid >> 3, sch = tid & 7;  // staging: rows srow + 32 q, 16-byte chunk sch of the 128-byte row

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  u32x4 ra[4], rb[4];
  auto load_tile = [&](int64_t k0) {
    const int64_t k = k0 + sch * 16;
    const bool kin = k + 15 < P.k;
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
      *reinterpret_cast<u32x4*>(As + (buf * kTile + srow + 32 * q) * kF8LD + sch * 16) = ra[q];
      *reinterpret_cast<u32x4*>(Bs + (buf * kTile + srow + 32 * q) * kF8LD + sch * 16) = rb[q];
    }
  };

  const int64_t nt = (P.k + kF8KT - 1) / kF8KT;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int64_t t = 0; t < nt; ++t) {
    const int buf = (int)(t & 1);
    const bool more = t + 1 < nt;
    if (more) load_tile((t + 1) * kF8KT);
    const char* at = As + buf * kTile * kF8LD;
    const char* bt = Bs + buf * kTile * kF8LD;
#pragma unroll
    for (int kk = 0; kk < kF8KT / 64; ++kk) {
      union Frag {
        i32x8 v;
        u32x4 h[2];
      } af[2], bfr[2];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const char* src = at + (wm * 64 + tm * 32 + r32) * kF8LD + kk * 64 + 32 * half;
        af[tm].h[0] = *reinterpret_cast<const u32x4*>(src);
        af[tm].h[1] = *reinterpret_cast<const u32x4*>(src + 16);
      }
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const char* src = bt + (wn * 64 + tn * 32 + r32) * kF8LD + kk * 64 + 32 * half;
        bfr[tn].h[0] = *reinterpret_cast<const u32x4*>(src);
        bfr[tn].h[1] = *reinterpret_cast<const u32x4*>(src + 16);
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[tm].v, bfr[tn].v, acc[tm][tn], 0, 0, 0,
                                                                        kF8UnitScale, 0, kF8UnitScale);
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }
  epi(acc, m0 + wm * 64, n0 + wn * 64, P.m, P.n, 0, 0, smem_raw + wave * kEpiLdsPerWave);
}

// Large single problems (the B x B score products) take the 256 x 256 LDS-DMA kernel of mi_gemm_bf16.h in its fp8 form;
// it writes one partial per 256 x 256 tile (gemm_fp8_n_partials).  MI_FP8_NO_BIG=1: A/B switch.
static inline bool gemm_fp8_use_big(int64_t m, int64_t n, int64_t k) {
  static const bool off = getenv("MI_FP8_NO_BIG") != nullptr;
  return !off && k % 128 == 0 && k > 0 && ((m + 255) / 256) * ((n + 255) / 256) >= 192;
}
static inline int64_t gemm_fp8_n_partials(int64_t m, int64_t n, int64_t k) {
  const int64_t t = gemm_fp8_use_big(m, n, k) ? 256 : kTile;
  return ((m + t - 1) / t) * ((n + t - 1) / t);
}

template <class Epi>
static inline int launch_gemm_fp8(const GemmF8Args& a, const Epi& epi, hipStream_t st, const char* what) {
  if (a.k % 16 != 0 || a.lda % 16 != 0 || a.ldb % 16 != 0 || (uintptr_t)a.a % 16 != 0 || (uintptr_t)a.b % 16 != 0) {
    set_error("%s: the fp8 GEMM needs K and the row pitches to be multiples of 16", what);
    return MI_ESHAPE;
  }
  if (gemm_fp8_use_big(a.m, a.n, a.k)) {
    GemmBf16Args args{};  // units of two bytes: see gemm_bf16_big_kernel<Epi, true>
    args.p[0] = GemmBf16Problem{reinterpret_cast<const bf16_t*>(a.a), a.lda / 2, reinterpret_cast<const bf16_t*>(a.b),
                                a.ldb / 2, a.m, a.n, a.k / 2};
    args.n_problems = 1;
    args.k_chunk = a.k / 2;
    MI_SET_DYN_SMEM((gemm_bf16_big_kernel<Epi, true>), kG2SmemBig, "hipFuncSetAttribute(gemm_bf16_big_kernel fp8)");
    dim3 grid((unsigned)((a.n + 255) / 256), (unsigned)((a.m + 255) / 256), 1);
    xcd_pick_blocks(grid.y, grid.x, 256, a.k / 2, 1, args.xcd_gy, args.xcd_gx);
    {
      ProfScope prof_(what, st);
      hipLaunchKernelGGL((gemm_bf16_big_kernel<Epi, true>), grid, dim3(512), kG2SmemBig, st, args, epi);
    }
    MI_LAUNCH_CHECK(what);
    return MI_OK;
  }
  MI_SET_DYN_SMEM((gemm_fp8_kernel<Epi>), kF8Smem, "hipFuncSetAttribute(gemm_fp8_kernel)");
  dim3 grid((unsigned)((a.n + kTile - 1) / kTile), (unsigned)((a.m + kTile - 1) / kTile), 1);
  {
    ProfScope prof_(what, st);
    hipLaunchKernelGGL((gemm_fp8_kernel<Epi>), grid, dim3(256), kF8Smem, st, a, epi);
  }
  MI_LAUNCH_CHECK(what);
  return MI_OK;
}

// any epilogue of mi_gemm_bf16.h behind a device-side output factor acc *= sa[prob][0] * sb[prob][0] (null: 1)
template <class Inner>
struct EpiScaled {
  static constexpr bool kReducesPartial = Inner::kReducesPartial;
  Inner inner;
  const float* sa[2];
  const float* sb[2];
  Partial* partials = nullptr;  // = inner.partials for a reducing epilogue (the 256 x 256 kernel writes them itself)
  __device__ __forceinline__ Partial lane_partial(f32x16 (&acc)[2][2], int64_t mb, int64_t nb, int64_t M, int64_t N) const {
    const float f = (sa[0] ? sa[0][0] : 1.0f) * (sb[0] ? sb[0][0] : 1.0f);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][tn][r] *= f;
    if constexpr (Inner::kReducesPartial) return inner.lane_partial(acc, mb, nb, M, N);
    else return Partial{};
  }
  __device__ __forceinline__ void operator()(f32x16 (&acc)[2][2], int64_t mb, int64_t nb, int64_t M, int64_t N, int prob,
                                             int zsplit, char* lds) const {
    const float f = (sa[prob] ? sa[prob][0] : 1.0f) * (sb[prob] ? sb[prob][0] : 1.0f);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][tn][r] *= f;
    inner(acc, mb, nb, M, N, prob, zsplit, lds);
  }
};

// epilogue of T = s_x s_w (x_q W_q): fp32 T (row-major) and its absmax
struct EpiT8 {
  static constexpr bool kReducesPartial = false;
  float* t;  // [M][N]
  Fp8Scales* sc;
  __device__ __forceinline__ void operator()(f32x16 (&acc)[2][2], int64_t mb, int64_t nb, int64_t M, int64_t N, int, int,
                                             char* lds) const {
    const float f = sc->scale[0] * sc->scale[2];
    const int lane = threadIdx.x & 63;
    const int col_l = lane & 31, half = lane >> 5;
    float m = 0.0f;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[tm][tn][r] *= f;
          const int64_t row = mb + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, col = nb + tn * 32 + col_l;
          if (row < M && col < N) m = fmaxf(m, fabsf(acc[tm][tn][r]));
        }
    m = wave_max_f(m);
    if (lane == 0) atomicMax(&sc->amax_bits[3], __float_as_uint(m));
    wave_tile_store_f32(acc, lds, t, N, mb, nb, M, N);
  }
};

}  // namespace mi
