// Fused concat-MLP critic, fp16-operand mode (MI_PREC_F16; round 4).
//
// Why fp16 and not bf16 here: the big products of this critic have ONE operand that never exists in memory -- it is
// generated in registers per pair: H1[p, k] = relu(U_i[k] + V_j[k]) in the forward, g_p H1[p, k] in the dW2 kernel.  In
// the bf16 kernels that generation is 20 - 28 vector instructions per 8-element MFMA fragment (fp32 adds, a max, a
// conversion per pair of elements; a multiply by g on top in the backward): 6.9 (forward) and 10.9 (dW2) vector
// instructions per MFMA, which made the vector port, not the matrix pipe, the bound of these kernels at two waves per SIMD
// (profiles/r3_concat_sq_counters.txt).  gfx950 has PACKED fp16 arithmetic (v_pk_add_f16, v_pk_mul_f16; there is no
// packed bf16 arithmetic) and its MFMA takes fp16 operands at the bf16 rate.  With U, V pre-scaled by a power of two such
// that |U_i[k] + V_j[k]| s <= 1/2, the hardware clamp modifier of v_pk_add_f16 IS the relu:
//     forward fragment  : 4 x  v_pk_add_f16 ... clamp                      (was ~20 instructions)
//     dW2 fragment      : 4 x (v_pk_add_f16 ... clamp ; v_pk_mul_f16)       (was ~28)
// and the fragments need no conversion.  fp16 also carries 11 significant bits where bf16 carries 8: against the
// unrounded reference the gradients of this mode are closer than the bf16 mode's.  What fp16 lacks is range; every
// operand tensor therefore carries a power-of-two scale derived on the device from its absmax (no host round trip):
//     s_uv : (max|U| + max|V|) s_uv in (1/4, 1/2]            -> Uh = fp16(U s_uv), Vh = fp16(V s_uv)
//     s_w  : max|W2| s_w in [2^13, 2^14)                      -> W2h = fp16(W2 s_w)
//     s_ww : max|w3| max|W2| s_ww in [2^13, 2^14) (a bound)   -> W2wP = fp16(w3[n] W2[n, k] s_ww)  (dU / dV kernel)
//     s_g  : max|g| s_g in [2^13, 2^14), max|g| from the statistics block (f16_g_scale); grad_out is applied to the
//            finished sums
// Scales are powers of two: scaling is exact, the only roundings are the conversions to fp16 and the packed operations
// (each correctly rounded).  Sums are accumulated in fp32 by the MFMA; the scale factors are undone once, in fp32, in the
// epilogues.  Values below 2^-14 after scaling become fp16 subnormals (absolute error 2^-25 of the tensor's largest
// magnitude), values below 2^-25 vanish: both far below the bf16 mode's relative 2^-9 per element.
//
// Rounding points of this mode (what oracle/mi_oracle.py concat_step_f16 restates):
//     forward   Z2 = sum_k fp16(clamp(Uh_i[k] + Vh_j[k])) W2h[n, k]        (fp32 accumulate), scores from fp32 Z2
//     dU / dV   E[p, k] = sum_n M[p, n] W2wP[n, k];  relu' of layer 1 decided on Uh_i[k] + Vh_j[k] > 0
//     dW2 ...   D[n, k] = sum_p M[p, n] fp16(fp16(g_p s_g) h_pk),  h_pk the forward's fp16 operand
#pragma once
#include "mi_concat_bwd.h"
#include "mi_concat_fwd_dma.h"

namespace mi {

__device__ __forceinline__ float wave_max_f16s(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
struct F16AbsmaxJobs {
  const float* in[4];
  int64_t n[4];
  F16Scales* sc;
};
// absmax of up to four fp32 tensors (blockIdx.y selects the tensor); the slots must be zeroed before the launch
static __global__ __launch_bounds__(256) void f16_absmax_kernel(F16AbsmaxJobs J) {
  const int q = blockIdx.y;
  const int64_t n = J.n[q], stride = (int64_t)gridDim.x * 256;
  const float* p = J.in[q];
  float m = 0.0f;
  if ((n & 3) == 0 && ((uintptr_t)p & 15) == 0) {
    const f32x4* p4 = reinterpret_cast<const f32x4*>(p);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n / 4; e += stride) {
      const f32x4 v = p4[e];
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
  } else {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) m = fmaxf(m, fabsf(p[e]));
  }
  m = wave_max_f16s(m);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    atomicMax(&J.sc->amax_bits[q], __float_as_uint(m));
  }
}

// Uh, Vh, W2h (forward operands) and, when the backward will run, Upk[i][k] = {Uh, Uh} (one dword per element: the dW2
// kernel adds it to a PAIR of columns) and VhT [H1][b] (the dW2 kernel stages [k][column] tiles).
struct F16PrepArgs {
  const float *u, *v, *w2;
  int64_t b_rows, b;
  int h1, h2;
  const F16Scales* sc;
  f16_t *uh, *vh, *w2h;
  unsigned* upk;  // may be null
  f16_t* vht;     // may be null
};
static __global__ __launch_bounds__(256) void f16_prep_kernel(F16PrepArgs A) {
  const F16ScaleSet s = f16_scales(A.sc);
  const int64_t nu = A.b_rows * A.h1, nv = A.b * A.h1, nw = (int64_t)A.h2 * A.h1;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nu; e += stride) {
    const f16_t h = (f16_t)(A.u[e] * s.s_uv);
    A.uh[e] = h;
    if (A.upk) {
      const unsigned short bits = __builtin_bit_cast(unsigned short, h);
      A.upk[e] = ((unsigned)bits << 16) | bits;
    }
  }
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nv; e += stride) A.vh[e] = (f16_t)(A.v[e] * s.s_uv);
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nw; e += stride) A.w2h[e] = (f16_t)(A.w2[e] * s.s_w);
}
// VhT[k][j] = Vh[j][k] through a 32 x 32 LDS tile (coalesced on both sides); grid (H1 / 32, ceil(b / 32))
static __global__ __launch_bounds__(256) void f16_transpose_v_kernel(const f16_t* __restrict__ vh, int64_t b, int h1,
                                                                     f16_t* __restrict__ vht) {
  __shared__ f16_t tile[32][34];
  const int k0 = blockIdx.x * 32;
  const int64_t j0 = (int64_t)blockIdx.y * 32;
  for (int e = threadIdx.x; e < 1024; e += 256) {
    const int jl = e >> 5, kl = e & 31;
    tile[jl][kl] = (j0 + jl < b && k0 + kl < h1) ? vh[(j0 + jl) * h1 + k0 + kl] : (f16_t)0.0f;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 1024; e += 256) {
    const int kl = e >> 5, jl = e & 31;
    if (j0 + jl < b && k0 + kl < h1) vht[(int64_t)(k0 + kl) * b + j0 + jl] = tile[jl][kl];
  }
}

// W2wP[k][m] = fp16(W2[n(m)][k] w3[n(m)] s_ww)   (transposed, permuted, scaled copy; [H1][H2]; see prep_w2w_kernel)
static __global__ void f16_prep_w2w_kernel(const float* __restrict__ w2, const float* __restrict__ w3, int H1, int H2,
                                           const F16Scales* __restrict__ sc, f16_t* __restrict__ out) {
  const float sww = f16_scales(sc).s_ww;
  const int64_t total = (int64_t)H1 * H2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int k = (int)(e / H2), m = (int)(e % H2);
    const int n = slot_to_n(m);
    out[e] = (f16_t)(w2[(int64_t)n * H1 + k] * w3[n] * sww);
  }
}

// two-part form for MI_PREC_F16X3: out [2][H1][H2] (hi, lo)
static __global__ void f16x3_prep_w2w_kernel(const float* __restrict__ w2, const float* __restrict__ w3, int H1, int H2,
                                             const F16Scales* __restrict__ sc, f16_t* __restrict__ out) {
  const float sww = f16_scales(sc).s_ww;
  const int64_t total = (int64_t)H1 * H2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int k = (int)(e / H2), m = (int)(e % H2);
    const int n = slot_to_n(m);
    const float v = w2[(int64_t)n * H1 + k] * w3[n] * sww;
    const f16_t hi = (f16_t)v;
    out[e] = hi;
    out[total + e] = (f16_t)(v - (float)hi);
  }
}

// relu(u + v) on four packed pairs: v_pk_add_f16 ... clamp (hipcc folds min(max(x, 0), 1) into the clamp modifier; the
// scaled sums never reach 1/2).  A NaN operand gives 0 under the kernel's DX10_CLAMP mode, as fmaxf(NaN, 0) does.
__device__ __forceinline__ f16x8 gen_h1_f16(const f16x8& u, const f16x8& v) {
  const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0}, o = {1, 1, 1, 1, 1, 1, 1, 1};
  return __builtin_elementwise_min(__builtin_elementwise_max(u + v, z), o);
}

// ================================================================================================= forward
// Same tiling, outputs and LDS-DMA staging as concat_fwd_dma_kernel<1> (mi_concat_fwd_dma.h): a 256-thread workgroup owns
// 8 image rows x 32 text columns and sweeps the H2 hidden units in passes of 128; per 64-deep k tile the W2 rows (128 x
// 128 B), the V rows (32 x 128 B) and the U rows (8 x 128 B) arrive by LDS-DMA.  All three tiles now have 128-byte rows
// and share one swizzle (16-byte chunk c of row r at chunk position c ^ ((r >> 1) & 7); U unswizzled: broadcast reads).
struct FwdF16Smem {
  static constexpr int NP = 128;
  static constexpr int W_BYTES = NP * 128;
  static constexpr int V_BYTES = kFwdTJ * 128;
  static constexpr int U_BYTES = kFwdTI * 128;
  static constexpr int BUF_BYTES = W_BYTES + V_BYTES + U_BYTES;
  static constexpr int TOTAL = 2 * BUF_BYTES + 2 * NP * 4 + kFwdTI * kFwdTJ * 4;
};

__global__ __launch_bounds__(256, 2) void concat_fwd_f16_kernel(
    const f16_t* __restrict__ Uh, const f16_t* __restrict__ Vh, const f16_t* __restrict__ W2h,
    const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3,
    const F16Scales* __restrict__ sc, int64_t b_rows, int64_t b, int H1, int H2, float* __restrict__ S,
    unsigned long long* __restrict__ bitsP, unsigned* __restrict__ bitsN, int natural_order) {
  using L = FwdF16Smem;
  constexpr int NP = L::NP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* b2s = reinterpret_cast<float*>(smem + 2 * L::BUF_BYTES);
  float* w3s = b2s + NP;
  float* sred = w3s + NP;  // [256]

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wp = wave;
  const int c = lane & 31, h = lane >> 5;
  const BitTransposeLane btl = bit_transpose_lane(c);
  // XCD-aware tile order: see concat_fwd_dma_kernel
  const int n_jt = (int)((b + kFwdTJ - 1) / kFwdTJ), n_it = (int)((b_rows + kFwdTI - 1) / kFwdTI);
  const int njx = (n_jt + 7) / 8;
  int jt = (int)(blockIdx.x & 7) + 8 * (int)((blockIdx.x >> 3) / n_it);
  int it = (int)((blockIdx.x >> 3) % n_it);
  if (natural_order) {
    jt = (int)(blockIdx.x % (8 * njx));
    it = (int)(blockIdx.x / (8 * njx));
  }
  if (jt >= n_jt || it >= n_it) return;
  const int64_t i0 = (int64_t)it * kFwdTI, j0 = (int64_t)jt * kFwdTJ;
  const int n_pass = H2 / NP;
  const int n_kt = H1 / 64;
  const int64_t JB = (b + 31) / 32;
  const F16ScaleSet scl = f16_scales(sc);
  const float s_in = scl.s_uv * scl.s_w, inv_in = 1.0f / s_in;  // powers of two: exact

  // ---- DMA source addresses: every instruction moves 8 rows x 128 B ----------------------------------------------------
  int woff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 32 * wave + 8 * q + (lane >> 3);
    woff[q] = row * H1 * 2 + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
  }
  const char* vsrc;
  {
    const int row = 8 * wave + (lane >> 3);
    int64_t gj = j0 + row;
    if (gj >= b) gj = b - 1;
    vsrc = reinterpret_cast<const char*>(Vh + gj * H1) + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
  }
  const char* usrc;
  {
    int64_t li = i0 + (lane >> 3);
    if (li >= b_rows) li = b_rows - 1;
    usrc = reinterpret_cast<const char*>(Uh + li * H1) + ((lane & 7) << 4);
  }
  auto issue_tile = [&](int pass, int kt, int buf) {
    char* base = smem + buf * L::BUF_BYTES;
    const char* wsrc = reinterpret_cast<const char*>(W2h + ((int64_t)pass * NP) * H1 + kt * 64);
#pragma unroll
    for (int q = 0; q < 4; ++q) MI_GLDS16(wsrc + woff[q], base + (32 * wave + 8 * q) * 128);
    MI_GLDS16(vsrc + kt * 128, base + L::W_BYTES + (8 * wave) * 128);
    if (wave == 0) MI_GLDS16(usrc + kt * 128, base + L::W_BYTES + L::V_BYTES);
  };

  // ---- fragment read offsets -----------------------------------------------------------------------------------------
  const int wrow_off = c * 128;  // + a * 32 * 128
  const int wswz = (c >> 1) & 7;
  const int vrow_off = L::W_BYTES + c * 128;
  const int urow_off = L::W_BYTES + L::V_BYTES + (2 * wp) * 128;  // + t * 128

  if (tid < NP) b2s[tid] = 0.0f;  // placeholder write so the first pass's loads below are ordered by the barrier
  float s_total[2] = {0.0f, 0.0f};

  for (int pass = 0; pass < n_pass; ++pass) {
    __syncthreads();  // previous pass: every wave has left its epilogue (reads w3s) and its last tile's LDS reads
    if (tid < NP) {
      b2s[tid] = b2[pass * NP + tid] * s_in;     // the accumulators hold s_uv s_w Z2
      w3s[tid] = w3[pass * NP + tid] * inv_in;
    }
    issue_tile(pass, 0, 0);
    __syncthreads();  // hipcc waits vmcnt(0) before the barrier: tile 0 landed; b2s / w3s visible

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float bias = b2s[a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
        acc[a][0][r] = bias;
        acc[a][1][r] = bias;
      }

    for (int kt = 0; kt < n_kt; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < n_kt) issue_tile(pass, kt + 1, buf ^ 1);
      const char* base = smem + buf * L::BUF_BYTES;
      struct Frag {
        f16x8 wf[4];
        f16x8 v, u[2];
      };
      auto read_frag = [&](int kk, Frag& f) {
        const int wpos = ((2 * kk + h) ^ wswz) << 4;
#pragma unroll
        for (int a = 0; a < 4; ++a) f.wf[a] = *reinterpret_cast<const f16x8*>(base + wrow_off + a * 32 * 128 + wpos);
        f.v = *reinterpret_cast<const f16x8*>(base + vrow_off + wpos);
#pragma unroll
        for (int t = 0; t < 2; ++t) f.u[t] = *reinterpret_cast<const f16x8*>(base + urow_off + t * 128 + ((2 * kk + h) << 4));
      };
      Frag fa, fb;
      f16x8 hfa[2], hfb[2];
      read_frag(0, fa);
#pragma unroll
      for (int t = 0; t < 2; ++t) hfa[t] = gen_h1_f16(fa.u[t], fa.v);
#define MI_FWD16_STEP(KK, CUR, HCUR, NXT, HNXT)                                                            \
  {                                                                                                       \
    if ((KK) < 3) read_frag((KK) + 1, NXT);                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) _Pragma("unroll") for (int t = 0; t < 2; ++t)           \
        acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(CUR.wf[a], HCUR[t], acc[a][t], 0, 0, 0);       \
    if ((KK) < 3) {                                                                                       \
      _Pragma("unroll") for (int t = 0; t < 2; ++t) HNXT[t] = gen_h1_f16(NXT.u[t], NXT.v);                \
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                  \
      _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                                  \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                                \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                \
      }                                                                                                   \
    }                                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
  }
      MI_FWD16_STEP(0, fa, hfa, fb, hfb)
      MI_FWD16_STEP(1, fb, hfb, fa, hfa)
      MI_FWD16_STEP(2, fa, hfa, fb, hfb)
      MI_FWD16_STEP(3, fb, hfb, fa, hfa)
#undef MI_FWD16_STEP
      __syncthreads();  // vmcnt(0) + barrier: tile kt+1 landed, buffer `buf` free for tile kt+2
    }

    // ---- epilogue of the pass: relu, dot with w3, sign bits (fwd_epilogue_row, mi_concat_fwd.h) ------------------------
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int64_t li = i0 + 2 * wp + t;
      const int64_t gj = j0 + c;
      const bool row_ok = li < b_rows, col_ok = gj < b;
      const int64_t lic = row_ok ? li : 0, gjc = col_ok ? gj : 0;
      s_total[t] += fwd_epilogue_row(acc[0][t], acc[1][t], acc[2][t], acc[3][t], w3s, h, c, bitsP != nullptr, row_ok,
                                     col_ok, bitsP + bitsp_index(lic, gjc, h, pass, (b + 31) / 32, (int)(H2 / 128)),
                                     bitsN + (lic * JB + jt) * H2 + pass * 128, btl);
    }
  }

#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float s = s_total[t] + __shfl_xor(s_total[t], 32);
    if (h == 0) sred[(2 * wp + t) * 32 + c] = s;
  }
  __syncthreads();
  {
    const int il = tid >> 5, jl = tid & 31;
    const int64_t li = i0 + il, gj = j0 + jl;
    if (li < b_rows && gj < b) S[li * b + gj] = sred[tid] + b3[0];
  }
}

// ================================================================================================= forward, two-part fp16
// MI_PREC_F16X3: the fp32-tolerance mode of this critic at a third of the fp16 rate (the exact fp32-input MFMA runs at a
// sixteenth).  Every operand of the big product is TWO fp16 parts, hi = fp16(v), lo = fp16(v - hi) (22 significant bits
// under the same power-of-two scales as the fp16 mode), and a product is hi*hi + hi*lo + lo*hi on three MFMAs with fp32
// accumulation (the lo*lo term is below 2^-22).  Same tiling, LDS layout and DMA as concat_fwd_f16_kernel with a 32-deep k
// tile: a W2 tile row is [hi 32 k | lo 32 k] (128 bytes, from the interleaved copy W2x), the V / U tile rows are 32 fp32
// values (128 bytes) of the pre-scaled copies Us = U s_uv, Vs = V s_uv -- relu(u + v) is formed in fp32 and split in
// registers (two conversions, one subtraction per pair of elements).
__global__ __launch_bounds__(256, 2) void concat_fwd_f16x3_kernel(
    const float* __restrict__ Us, const float* __restrict__ Vs, const f16_t* __restrict__ W2x /* [H2][H1 / 32][hi 32 | lo 32] */,
    const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3,
    const F16Scales* __restrict__ sc, int64_t b_rows, int64_t b, int H1, int H2, float* __restrict__ S,
    unsigned long long* __restrict__ bitsP, unsigned* __restrict__ bitsN, int natural_order) {
  using L = FwdF16Smem;
  constexpr int NP = L::NP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* b2s = reinterpret_cast<float*>(smem + 2 * L::BUF_BYTES);
  float* w3s = b2s + NP;
  float* sred = w3s + NP;  // [256]

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wp = wave;
  const int c = lane & 31, h = lane >> 5;
  const BitTransposeLane btl = bit_transpose_lane(c);
  const int n_jt = (int)((b + kFwdTJ - 1) / kFwdTJ), n_it = (int)((b_rows + kFwdTI - 1) / kFwdTI);
  const int njx = (n_jt + 7) / 8;
  int jt = (int)(blockIdx.x & 7) + 8 * (int)((blockIdx.x >> 3) / n_it);
  int it = (int)((blockIdx.x >> 3) % n_it);
  if (natural_order) {
    jt = (int)(blockIdx.x % (8 * njx));
    it = (int)(blockIdx.x / (8 * njx));
  }
  if (jt >= n_jt || it >= n_it) return;
  const int64_t i0 = (int64_t)it * kFwdTI, j0 = (int64_t)jt * kFwdTJ;
  const int n_pass = H2 / NP;
  const int n_kt = H1 / 32;
  const int64_t JB = (b + 31) / 32;
  const F16ScaleSet scl = f16_scales(sc);
  const float s_in = scl.s_uv3 * scl.s_w, inv_in = 1.0f / s_in;

  // ---- DMA source addresses: every instruction moves 8 rows x 128 B (rows of W2x, Vs, Us are H1 * 4 bytes long) -----------
  int woff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 32 * wave + 8 * q + (lane >> 3);
    woff[q] = row * H1 * 4 + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
  }
  const char* vsrc;
  {
    const int row = 8 * wave + (lane >> 3);
    int64_t gj = j0 + row;
    if (gj >= b) gj = b - 1;
    vsrc = reinterpret_cast<const char*>(Vs + gj * H1) + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
  }
  const char* usrc;
  {
    int64_t li = i0 + (lane >> 3);
    if (li >= b_rows) li = b_rows - 1;
    usrc = reinterpret_cast<const char*>(Us + li * H1) + ((lane & 7) << 4);
  }
  auto issue_tile = [&](int pass, int kt, int buf) {
    char* base = smem + buf * L::BUF_BYTES;
    const char* wsrc = reinterpret_cast<const char*>(W2x) + ((int64_t)pass * NP) * H1 * 4 + kt * 128;
#pragma unroll
    for (int q = 0; q < 4; ++q) MI_GLDS16(wsrc + woff[q], base + (32 * wave + 8 * q) * 128);
    MI_GLDS16(vsrc + kt * 128, base + L::W_BYTES + (8 * wave) * 128);
    if (wave == 0) MI_GLDS16(usrc + kt * 128, base + L::W_BYTES + L::V_BYTES);
  };

  const int wrow_off = c * 128;
  const int wswz = (c >> 1) & 7;
  const int vrow_off = L::W_BYTES + c * 128;
  const int urow_off = L::W_BYTES + L::V_BYTES + (2 * wp) * 128;

  if (tid < NP) b2s[tid] = 0.0f;
  float s_total[2] = {0.0f, 0.0f};

  for (int pass = 0; pass < n_pass; ++pass) {
    __syncthreads();
    if (tid < NP) {
      b2s[tid] = b2[pass * NP + tid] * s_in;
      w3s[tid] = w3[pass * NP + tid] * inv_in;
    }
    issue_tile(pass, 0, 0);
    __syncthreads();

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float bias = b2s[a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
        acc[a][0][r] = bias;
        acc[a][1][r] = bias;
      }

    for (int kt = 0; kt < n_kt; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < n_kt) issue_tile(pass, kt + 1, buf ^ 1);
      const char* base = smem + buf * L::BUF_BYTES;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        f16x8 wh[4], wl[4];
        const int hpos = ((2 * kk + h) ^ wswz) << 4, lpos = ((4 + 2 * kk + h) ^ wswz) << 4;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          wh[a] = *reinterpret_cast<const f16x8*>(base + wrow_off + a * 32 * 128 + hpos);
          wl[a] = *reinterpret_cast<const f16x8*>(base + wrow_off + a * 32 * 128 + lpos);
        }
        // this lane's 8 k values: 16 kk + 8 h .. + 7 = 16-byte chunks 4 kk + 2 h and 4 kk + 2 h + 1 of the fp32 rows
        const int q0 = 4 * kk + 2 * h;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(base + vrow_off + ((q0 ^ wswz) << 4));
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(base + vrow_off + (((q0 + 1) ^ wswz) << 4));
        f16x8 hh[2], hl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f32x4 u0 = *reinterpret_cast<const f32x4*>(base + urow_off + t * 128 + q0 * 16);
          const f32x4 u1 = *reinterpret_cast<const f32x4*>(base + urow_off + t * 128 + q0 * 16 + 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x0 = fmaxf(u0[e] + v0[e], 0.0f), x1 = fmaxf(u1[e] + v1[e], 0.0f);
            const f16_t h0 = (f16_t)x0, h1 = (f16_t)x1;
            hh[t][e] = h0;
            hh[t][4 + e] = h1;
            hl[t][e] = (f16_t)(x0 - (float)h0);
            hl[t][4 + e] = (f16_t)(x1 - (float)h1);
          }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[a], hh[t], acc[a][t], 0, 0, 0);  // small terms first
            acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[a], hl[t], acc[a][t], 0, 0, 0);
            acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[a], hh[t], acc[a][t], 0, 0, 0);
          }
      }
      __syncthreads();
    }

#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int64_t li = i0 + 2 * wp + t;
      const int64_t gj = j0 + c;
      const bool row_ok = li < b_rows, col_ok = gj < b;
      const int64_t lic = row_ok ? li : 0, gjc = col_ok ? gj : 0;
      s_total[t] += fwd_epilogue_row(acc[0][t], acc[1][t], acc[2][t], acc[3][t], w3s, h, c, bitsP != nullptr, row_ok,
                                     col_ok, bitsP + bitsp_index(lic, gjc, h, pass, (b + 31) / 32, (int)(H2 / 128)),
                                     bitsN + (lic * JB + jt) * H2 + pass * 128, btl);
    }
  }

#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float s = s_total[t] + __shfl_xor(s_total[t], 32);
    if (h == 0) sred[(2 * wp + t) * 32 + c] = s;
  }
  __syncthreads();
  {
    const int il = tid >> 5, jl = tid & 31;
    const int64_t li = i0 + il, gj = j0 + jl;
    if (li < b_rows && gj < b) S[li * b + gj] = sred[tid] + b3[0];
  }
}

// Us = U s_uv3, Vs = V s_uv3 (fp32, exact; s_uv3 puts max|U| + max|V| in [2^13, 2^14): no clamp trick in this mode), W2x[n][kt][hi 32 | lo 32] (two-part fp16 of W2 s_w); when the backward will run
// also VsT [H1][b] (fp32 transpose of Vs: the dW2 kernel stages [k][column] tiles).
struct F16X3PrepArgs {
  const float *u, *v, *w2;
  int64_t b_rows, b;
  int h1, h2;
  const F16Scales* sc;
  float *us, *vs;
  f16_t* w2x;
};
static __global__ __launch_bounds__(256) void f16x3_prep_kernel(F16X3PrepArgs A) {
  const F16ScaleSet s = f16_scales(A.sc);
  const int64_t nu = A.b_rows * A.h1, nv = A.b * A.h1, nw = (int64_t)A.h2 * A.h1;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nu; e += stride) A.us[e] = A.u[e] * s.s_uv3;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nv; e += stride) A.vs[e] = A.v[e] * s.s_uv3;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nw; e += stride) {
    const int64_t n = e / A.h1;
    const int k = (int)(e % A.h1);
    const float w = A.w2[e] * s.s_w;
    const f16_t hi = (f16_t)w;
    const f16_t lo = (f16_t)(w - (float)hi);
    f16_t* row = A.w2x + n * (2 * (int64_t)A.h1) + (k >> 5) * 64 + (k & 31);
    row[0] = hi;
    row[32] = lo;
  }
}

// ================================================================================================= dW2 kernel
// D[n, k] = sum_p M[p, n] fp16(g'_p h_pk),  g' = fp16(g s_g) (grad_out NOT included), h_pk = clamp(Uh_i[k] + Vh_j[k]).
// Same decomposition as concat_bwd_dw2_kernel: workgroup = (k block of 256, n block of 256, row split), wave (wn, wk):
// 128 n x 64 k, rows outside, the 32 columns of a column block in two 16-deep MFMA steps.  What changed:
//   * the generated operand: 4 x (v_pk_add_f16 clamp + v_pk_mul_f16) per fragment instead of 8 x (add, max, mul) + 4
//     conversions; U arrives as {u, u} dwords, the V tile as fp16 [k][column] rows (one 16-byte LDS read per fragment),
//     g as fp16 rows (one 16-byte broadcast read per row and step);
//   * the bit -> fragment table holds fp16 {0, 2};
//   * Dslab holds 2 s_uv s_g D (undone by the finishing kernel together with grad_out).
struct Dw2F16Smem {
  static constexpr int VT_PITCH = 80;                       // bytes per k row: 32 columns x 2 B + 16 (conflict-free b128)
  static constexpr int VT_BYTES = 256 * VT_PITCH;
  static constexpr int GS_BYTES = kDw2IB * 64;              // [128 rows][32 columns] fp16
  static constexpr int LUT_BYTES = 256 * 16;
  static constexpr int TOTAL = VT_BYTES + GS_BYTES + LUT_BYTES;
};

__global__ __launch_bounds__(512) void concat_bwd_dw2_f16_kernel(
    const unsigned* __restrict__ Upk, const f16_t* __restrict__ VhT, const unsigned* __restrict__ bitsN,
    const float* __restrict__ S, const int64_t* __restrict__ sid_rows, const int64_t* __restrict__ sid_cols,
    const mi_stats* __restrict__ stats, int64_t b_rows, int64_t b, int64_t row_offset, int H1, int H2,
    int rows_per_split, int natural_order, float* __restrict__ Dslab) {
  using L = Dw2F16Smem;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  char* vt = smem_raw;                                                 // [256 k][VT_PITCH]
  f16_t* gs = reinterpret_cast<f16_t*>(smem_raw + L::VT_BYTES);        // [kDw2IB][32]
  f16x8* lut = reinterpret_cast<f16x8*>(smem_raw + L::VT_BYTES + L::GS_BYTES);  // [256]: byte of sign bits -> fragment {0, 2}

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wn = wave >> 2, wk = wave & 3;
  const int c = lane & 31, h = lane >> 5;
  const int n_kb = (H1 + 255) / 256, n_nb = H2 / 256;
  const int n_split = (int)((b_rows + rows_per_split - 1) / rows_per_split);
  int tile, zsp;
  if (!xcd_decode(n_kb * n_nb, n_split, tile, zsp, natural_order)) return;
  const int kb0 = (tile % n_kb) * 256, nb0 = (tile / n_kb) * 256;
  const int64_t ilo = (int64_t)zsp * rows_per_split;
  int64_t ihi = ilo + rows_per_split;
  if (ihi > b_rows) ihi = b_rows;
  const int64_t JB = (b + 31) / 32;

  const float lse = stats->lse;
  const float gpos = -1.0f / (float)stats->n_pos;
  const float gscale = f16_g_scale(stats, kF16GLo);

  if (tid < 256) {
    f16x8 f;
#pragma unroll
    for (int q = 0; q < 8; ++q) f[q] = (f16_t)(((tid >> q) & 1) ? 2.0f : 0.0f);
    lut[tid] = f;
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.0f;

  const int kcol[2] = {kb0 + 64 * wk + c, kb0 + 64 * wk + 32 + c};
  const bool kok[2] = {kcol[0] < H1, kcol[1] < H1};
  const int kclamp[2] = {kok[0] ? kcol[0] : 0, kok[1] ? kcol[1] : 0};

  for (int64_t jb = 0; jb < JB; ++jb) {
    const int64_t j0 = jb * 32;
    __syncthreads();  // previous column block's readers of vt / gs are done
    // V tile: vt[k][column] = VhT[kb0 + k][j0 + column]; 1024 chunks of 16 bytes, two per thread
    for (int e = tid; e < 1024; e += 512) {
      const int kk = e >> 2, ch = e & 3;
      u32x4 x = {0u, 0u, 0u, 0u};
      if (kb0 + kk < H1) {
        const f16_t* src = VhT + (int64_t)(kb0 + kk) * b + j0 + 8 * ch;
        if (j0 + 8 * ch + 8 <= b && (b & 7) == 0) {
          x = *reinterpret_cast<const u32x4*>(src);
        } else {
          union { u32x4 v; f16_t hh[8]; } t;
          t.v = x;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            int64_t gj = j0 + 8 * ch + q;
            if (gj >= b) gj = b - 1;
            t.hh[q] = VhT[(int64_t)(kb0 + kk) * b + gj];
          }
          x = t.v;
        }
      }
      *reinterpret_cast<u32x4*>(vt + kk * L::VT_PITCH + 16 * ch) = x;
    }
    for (int64_t ib = ilo; ib < ihi; ib += kDw2IB) {
      __syncthreads();  // vt visible (first batch); previous batch's readers of gs are done
      for (int e = tid; e < kDw2IB * 32; e += 512) {
        const int il = e >> 5, jl = e & 31;
        const int64_t li = ib + il, gj = j0 + jl;
        float g = 0.0f;
        if (li < ihi && gj < b)
          g = gscale * pair_grad(S[li * b + gj], row_offset + li, gj, sid_rows[li], sid_cols[gj], lse, 1.0f, gpos);
        gs[e] = (f16_t)g;
      }
      __syncthreads();
      const int n_i = (int)((ihi - ib) < kDw2IB ? (ihi - ib) : kDw2IB);
#pragma unroll
      for (int s = 0; s < 2; ++s) {  // 16 columns per MFMA K step: this lane half takes columns 16 s + 8 h + [0, 8)
        f16x8 vreg[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
          vreg[t] = *reinterpret_cast<const f16x8*>(vt + (64 * wk + 32 * t + c) * L::VT_PITCH + 2 * (16 * s + 8 * h));
        // sign words and U values travel through a ring of four register sets, three rows ahead of their use (a row is
        // 8 MFMAs ~ 260 cycles: one row of lead did not cover a miss of the streamed bit image); rows beyond the batch
        // are clamped (loaded, never used) so that the loop body holds no branch around its loads
        unsigned wr[4][4], ur[4][2];
        const unsigned* wbase = bitsN + (ib * JB + jb) * H2 + nb0 + 128 * wn + c;
        const unsigned wstride = (unsigned)(JB * H2);
        const unsigned* ubase[2] = {Upk + ib * H1 + kclamp[0], Upk + ib * H1 + kclamp[1]};
        auto load_row = [&](auto SLOT, int il) __attribute__((always_inline)) {
          constexpr int SL = decltype(SLOT)::value;
          const unsigned ilc = (unsigned)(il < n_i ? il : n_i - 1);  // 32-bit offsets inside the batch (<= 128 rows)
          const unsigned* wp_ = wbase + ilc * wstride;
#pragma unroll
          for (int a = 0; a < 4; ++a) wr[SL][a] = wp_[32 * a];
#pragma unroll
          for (int t = 0; t < 2; ++t) ur[SL][t] = ubase[t][ilc * (unsigned)H1];
        };
        auto row = [&](auto SLOT, int il) __attribute__((always_inline)) {
          constexpr int SL = decltype(SLOT)::value;
          const f16x8 g8 = *reinterpret_cast<const f16x8*>(gs + il * 32 + 16 * s + 8 * h);
          f16x8 hf[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const f16x2 u2 = __builtin_bit_cast(f16x2, ur[SL][t]);
            const f16x8 u8 = {u2[0], u2[1], u2[0], u2[1], u2[0], u2[1], u2[0], u2[1]};
            hf[t] = gen_h1_f16(u8, vreg[t]) * g8;
          }
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const f16x8 mf = lut[(wr[SL][a] >> (16 * s + 8 * h)) & 0xFFu];
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(mf, hf[t], acc[a][t], 0, 0, 0);
          }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        // (scheduling fences: hipcc otherwise sinks the look-ahead loads to the row that uses them)
        load_row(I0{}, 0);
        load_row(I1{}, 1);
        load_row(I2{}, 2);
        __builtin_amdgcn_sched_barrier(0);
        for (int il = 0; il < n_i; il += 4) {
          load_row(I3{}, il + 3);
          __builtin_amdgcn_sched_barrier(0);
          row(I0{}, il);
          load_row(I0{}, il + 4);
          __builtin_amdgcn_sched_barrier(0);
          row(I1{}, il + 1);  // rows in [n_i, 4 ceil(n_i / 4)) carry g = 0 (their gs rows are zero): no branch in the body
          load_row(I1{}, il + 5);
          __builtin_amdgcn_sched_barrier(0);
          row(I2{}, il + 2);
          load_row(I2{}, il + 6);
          __builtin_amdgcn_sched_barrier(0);
          row(I3{}, il + 3);
        }
      }
    }
  }

  // ---- store the partial D tile (rows n = registers, columns k = lanes) -------------------------------------------
  float* out = Dslab + (int64_t)zsp * H2 * H1;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nb0 + 128 * wn + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (kok[t]) out[(int64_t)n * H1 + kcol[t]] = acc[a][t][r];
      }
}

}  // namespace mi

namespace mi {

// ================================================================================================= db2, fp16 mode
// concat_bwd_db2_kernel (mi_concat_bwd.h) walks the 32 bits of every sign word with ~3 vector instructions per bit
// (0.85 ms at B = 4096 for 2 MB of output).  In the fp16 mode g is an fp16 operand anyway (the dW2 kernel multiplies
// g' = g s_g into its generated operand): here a byte of the word selects eight 0 / 1 halves from a table and
// v_dot2c_f32_f16 adds two products g' * bit per instruction into an fp32 sum -- 4 table reads + 4 reads of g' + 16 dots per
// word.  Same outputs (row-slice sums of g M per hidden unit, the slice's sum of g for db3).  The sums come out scaled by
// s_g and without grad_out, both applied to the finished sum.
__global__ __launch_bounds__(512) void concat_bwd_db2_f16_kernel(const unsigned* __restrict__ bitsN, const float* __restrict__ S,
                                                                 const int64_t* __restrict__ sid_rows,
                                                                 const int64_t* __restrict__ sid_cols,
                                                                 const mi_stats* __restrict__ stats,
                                                                 const float* __restrict__ grad_out, int64_t b_rows,
                                                                 int64_t b, int64_t row_offset, int H2, int rows_per_split,
                                                                 float* __restrict__ mslab /* [n_split][H2] */,
                                                                 float* __restrict__ gsum /* [n_split] */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
  f16x8v* lut = reinterpret_cast<f16x8v*>(smem_raw);                     // [256]: bit q of the index -> half q = 1
  f16_t* grow = reinterpret_cast<f16_t*>(smem_raw + 256 * 16);          // [32 * JB]: g' of the row, zero beyond the batch
  __shared__ float red[8];
  const int tid = threadIdx.x;
  const int64_t JB = (b + 31) / 32;
  const int64_t ilo = (int64_t)blockIdx.x * rows_per_split;
  int64_t ihi = ilo + rows_per_split;
  if (ihi > b_rows) ihi = b_rows;
  const float go = grad_out ? grad_out[0] : 1.0f;
  const float lse = stats->lse;
  const float gpos = -1.0f / (float)stats->n_pos;
  const float gscale = f16_g_scale(stats, kF16GLo);
  if (tid < 256) {
    f16x8v f;
#pragma unroll
    for (int q = 0; q < 8; ++q) f[q] = (f16_t)(((tid >> q) & 1) ? 1.0f : 0.0f);
    lut[tid] = f;
  }
  float gtot = 0.0f;
  float macc[2] = {0.0f, 0.0f};  // hidden units tid and tid + 512
  for (int64_t li = ilo; li < ihi; ++li) {
    __syncthreads();
    const int64_t si = sid_rows[li];
    for (int64_t gj = tid; gj < JB * 32; gj += 512) {
      float g = 0.0f;
      if (gj < b) g = pair_grad(S[li * b + gj], row_offset + li, gj, si, sid_cols[gj], lse, 1.0f, gpos);
      grow[gj] = (f16_t)(g * gscale);
      gtot += g;
    }
    __syncthreads();
    for (int pass = 0; pass * 512 < H2; ++pass) {
      const int n = pass * 512 + tid;
      if (n < H2) {
        float a = 0.0f;
        for (int64_t jb = 0; jb < JB; ++jb) {
          const unsigned w = bitsN[(li * JB + jb) * H2 + n];
          const f16x8v* gp = reinterpret_cast<const f16x8v*>(grow + jb * 32);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f16x8v l = lut[(w >> (8 * q)) & 255u];
            const f16x8v g8 = gp[q];
#pragma unroll
            for (int e = 0; e < 4; ++e)
              a = __builtin_amdgcn_fdot2(f16x2{l[2 * e], l[2 * e + 1]}, f16x2{g8[2 * e], g8[2 * e + 1]}, a, false);
          }
        }
        macc[pass & 1] += a;
      }
    }
  }
  const float unscale = go / gscale;
  for (int pass = 0; pass * 512 < H2 && pass < 2; ++pass) {
    const int n = pass * 512 + tid;
    if (n < H2) mslab[(int64_t)blockIdx.x * H2 + n] = macc[pass] * unscale;
  }
  gtot = wave_sum(gtot) * go;
  if ((tid & 63) == 0) red[tid >> 6] = gtot;
  __syncthreads();
  if (tid == 0) {
    float t = 0.0f;
    for (int w = 0; w < 8; ++w) t += red[w];
    gsum[blockIdx.x] = t;
  }
}

// ================================================================================================= dU / dV kernel, round 4
// concat_bwd_duv_kernel (mi_concat_bwd.h) spent 58 % of its wave cycles parked at waits (profiles/r3_concat_sq_counters.txt,
// SQ_WAIT_ANY): four workgroup barriers per 8-column step (g and -V staged through LDS, a two-phase cross-wave sum of the
// dV partials), the step's sign words loaded at its start and needed at once, and BOTH waves of every SIMD in one
// 512-thread workgroup: they reach their MFMA phase, their epilogue and the barriers together, so the matrix pipe idles
// through every epilogue.  Same math, same slabs, new structure (the recipe of the forward kernel):
//   * 256-thread workgroups, TWO per CU: the two waves of a SIMD belong to different workgroups with independent
//     barriers and drift apart, one's epilogue runs under the other's MFMAs.  A workgroup owns 64 image rows x 64 k (was
//     128 k: the [64 k x H2] weight slice is 64 KB, unpadded under an XOR swizzle) and steps over 4 text columns at a time;
//   * a wave's 32-pair MFMA tile is 8 image rows x 4 text columns; the four waves are four row groups of 16 rows: dU needs
//     no cross-wave sum at all, a dV partial is shared by the four waves and all of them fit the LDS at once (4 KB):
//     write -> barrier -> read instead of two phases;
//   * nothing else goes through workgroup-shared LDS: each wave computes the g of its own 64 pairs (handed over through
//     256 bytes of LDS of its own, no barrier) and reads the -V values of its epilogue straight from L2;
//   * the sign words of the NEXT step are loaded into the registers of the current step as soon as their last MFMA has
//     issued; S and the column id of the next step one step ahead.
// 130 registers, no spills (the first attempt -- the same ideas inside one 512-thread workgroup with 128 k per wave -- sat at
// 256 registers with 528 bytes of scratch, reloaded between the MFMAs behind vmcnt(0): 31 ms against 16; profiles/README.md).
struct Duv3Smem {
  static constexpr int KC = 64;
  static constexpr int LUT_BYTES = 256 * 16;
  static constexpr int DVRED_BYTES = 4 * 4 * KC * 4;  // [wave][4 columns][KC] fp32
  static constexpr int GSW_BYTES = 4 * 64 * 4;        // [wave][64 pairs] fp32
  static size_t total(int h2, bool x3 = false) {
    return (size_t)KC * h2 * 2 * (x3 ? 2 : 1) + LUT_BYTES + DVRED_BYTES + GSW_BYTES;
  }
};
constexpr int kDuv3TJ = 4;  // text columns per step

// X3 (MI_PREC_F16X3): W2wP holds TWO parts per element, [2][H1][H2] (hi, lo); the slice of both sits in LDS (one workgroup
// per CU then) and every product is two MFMAs (the bit operand is exact).
template <typename OpT, typename UvT, bool X3 = false>
__global__ __launch_bounds__(256, 2) void concat_bwd_duv3_kernel(
    const UvT* __restrict__ U, const UvT* __restrict__ V, const OpT* __restrict__ W2wP,
    const unsigned long long* __restrict__ bitsP, const float* __restrict__ S, const int64_t* __restrict__ sid_rows,
    const int64_t* __restrict__ sid_cols, const mi_stats* __restrict__ stats, const float* __restrict__ grad_out,
    int64_t b_rows, int64_t b, int64_t row_offset, int H1, int H2, int cols_per_split, int natural_order,
    float* __restrict__ dUslab /* [n_jsplit][b_rows][H1] */, float* __restrict__ dVslab /* [n_iblk][b][H1] */,
    const F16Scales* __restrict__ sc) {
  using Vec8 = typename Op16<OpT>::Vec8;
  constexpr int KC = Duv3Smem::KC, NT = 2;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  char* wt = smem_raw;                                                       // [KC][H2] 16-bit, swizzled (X3: hi, then lo)
  const int wt_part = KC * H2 * 2;
  const int wt_bytes = X3 ? 2 * wt_part : wt_part;
  Vec8* lut = reinterpret_cast<Vec8*>(smem_raw + wt_bytes);                  // [256]
  float* dvred = reinterpret_cast<float*>(smem_raw + wt_bytes + Duv3Smem::LUT_BYTES);  // [4][4][KC]
  float* gsw = dvred + 4 * 4 * KC;                                           // [4][64]

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int c = lane & 31, h = lane >> 5;
  const int n_kc = (H1 + KC - 1) / KC, n_iblk = (int)((b_rows + kDuvTI - 1) / kDuvTI);
  const int n_js = (int)((b + cols_per_split - 1) / cols_per_split);
  int kci, oi;
  if (!xcd_decode(n_kc, n_iblk * n_js, kci, oi, natural_order)) return;
  const int iblk = oi % n_iblk, jsp = oi / n_iblk;
  const int kc0 = kci * KC;
  const int64_t i0 = (int64_t)iblk * kDuvTI;
  const int64_t jlo = (int64_t)jsp * cols_per_split;
  int64_t jhi = jlo + cols_per_split;
  if (jhi > b) jhi = b;
  const int hw = H2 / 128;  // 64-bit sign words per pair and lane half
  const int64_t jb32 = (b + 31) / 32;

  const float go = grad_out ? grad_out[0] : 1.0f;
  const float lse = stats->lse;
  const float gpos = -go / (float)stats->n_pos;
  float gscale = 0.5f;  // the fragment table holds 2.0 for a set bit
  if (sc) gscale /= f16_scales(sc).s_ww;

  // ---- one-time setup: weight slice (swizzled), table, U registers ---------------------------------------------------
  {
    const int cpr = H2 / 8;  // 16-byte chunks per row
    for (int e = tid; e < KC * cpr; e += 256) {
      const int row = e / cpr, q = e % cpr;
      u32x4 x = {0u, 0u, 0u, 0u};
      if (kc0 + row < H1) x = *reinterpret_cast<const u32x4*>(W2wP + (int64_t)(kc0 + row) * H2 + 8 * q);
      *reinterpret_cast<u32x4*>(wt + row * (H2 * 2) + ((q ^ (row & 15)) << 4)) = x;
      if constexpr (X3) {
        u32x4 xl = {0u, 0u, 0u, 0u};
        if (kc0 + row < H1) xl = *reinterpret_cast<const u32x4*>(W2wP + (int64_t)H1 * H2 + (int64_t)(kc0 + row) * H2 + 8 * q);
        *reinterpret_cast<u32x4*>(wt + wt_part + row * (H2 * 2) + ((q ^ (row & 15)) << 4)) = xl;
      }
    }
    {
      Vec8 f;
#pragma unroll
      for (int q = 0; q < 8; ++q) f[q] = (OpT)(((tid >> q) & 1) ? 2.0f : 0.0f);
      lut[tid] = f;
    }
  }
  // accumulator row rho = (r & 3) + 8 (r >> 2) + 4 h of tile m  <->  image row 16 wave + 8 m + 2 (r >> 2) + h, column r & 3
  float ureg[2][4][NT], duacc[2][4][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int is = 0; is < 4; ++is) {
      int64_t li = i0 + 16 * wave + 8 * m + 2 * is + h;
      if (li >= b_rows) li = b_rows - 1;
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        int k = kc0 + 32 * ct + c;  // columns beyond H1 are never stored: clamp the address, no select
        if (k >= H1) k = H1 - 1;
        ureg[m][is][ct] = (float)U[li * H1 + k];
        duacc[m][is][ct] = 0.0f;
      }
    }
  // the pair this lane LOADS (A-operand row c of tile m): image row 16 wave + 8 m + (c >> 2), column c & 3.  Rows and
  // columns outside the batch are clamped: their g is 0, so whatever bits they load multiply nothing.
  int64_t lrow[2];
  bool lrow_ok[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    lrow[m] = i0 + 16 * wave + 8 * m + (c >> 2);
    lrow_ok[m] = lrow[m] < b_rows;
    if (!lrow_ok[m]) lrow[m] = b_rows - 1;
  }
  const int64_t grow = lrow[h];  // the pair whose g this lane COMPUTES: tile m = h, row c of it
  const bool grow_ok = lrow_ok[h];
  const int64_t gsid = sid_rows[grow];
  auto load_words = [&](unsigned long long (&w)[2][4], int pw, int64_t jw) __attribute__((always_inline)) {
    int64_t gj = jw + (c & 3);
    if (gj >= b) gj = b - 1;
#pragma unroll
    for (int m = 0; m < 2; ++m) w[m][pw] = bitsP[bitsp_index(lrow[m], gj, h, pw, jb32, hw)];
  };
  unsigned long long words[2][4] = {};
#pragma unroll
  for (int pw = 0; pw < 4; ++pw)
    if (pw < hw) load_words(words, pw, jlo);
  float s_next = 0.0f;
  int64_t sidc_next = 0;
  {
    const int64_t gj = jlo + (c & 3);
    if (grow_ok && gj < jhi) {
      s_next = S[grow * b + gj];
      sidc_next = sid_cols[gj];
    }
  }
  __syncthreads();

  const int brow_off = c * (H2 * 2);  // + 32 ct rows
  for (int64_t j = jlo; j < jhi; j += kDuv3TJ) {
    // ---- g of this wave's 64 pairs -> its own 256 bytes of LDS (wave-level hand-over) ---------------------------------
    {
      const int64_t gj = j + (c & 3);
      float g = 0.0f;
      if (grow_ok && gj < jhi) g = gscale * pair_grad(s_next, row_offset + grow, gj, gsid, sidc_next, lse, go, gpos);
      gsw[wave * 64 + lane] = g;
      const int64_t gjn = gj + kDuv3TJ;
      if (grow_ok && gjn < jhi) {
        s_next = S[grow * b + gjn];
        sidc_next = sid_cols[gjn];
      }
    }
    // ---- E[pair, k] = sum_n M[pair, n] W2w[n, k] on MFMA ------------------------------------------------------------
    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][ct][r] = 0.0f;
    // the epilogue's V values: issued before the step's MFMAs (8 registers; issued under the last block of MFMAs their
    // latency was still exposed: the fp16 form, whose 2-byte loads return later, ran 5 % behind the bf16 form)
    UvT vraw[4][NT];
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) {
      int64_t gj = j + jq;
      if (gj >= b) gj = b - 1;
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        int k = kc0 + 32 * ct + c;
        if (k >= H1) k = H1 - 1;
        vraw[jq][ct] = V[gj * H1 + k];
      }
    }
    // (A hand-pipelined form of this loop -- the four fragments of step s + 2 fetched through rotating register sets before
    // the MFMAs of step s -- made hipcc extract table indices far ahead and spill: 224 - 400 bytes of scratch reloaded
    // between the MFMAs, 22 ms against 14; profiles/README.md.)
#pragma unroll
    for (int pw = 0; pw < 4; ++pw) {
      if (pw < hw) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          Vec8 af[2];
#pragma unroll
          for (int m = 0; m < 2; ++m) af[m] = lut[(unsigned)(words[m][pw] >> (8 * s)) & 0xFFu];
          const int q = 8 * (2 * pw + h) + s;
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) {
            // row 32 ct + c: (row & 15) == (c & 15)
            const Vec8 bfr = *reinterpret_cast<const Vec8*>(wt + brow_off + ct * 32 * (H2 * 2) + ((q ^ (c & 15)) << 4));
            if constexpr (X3) {
              const Vec8 bfl = *reinterpret_cast<const Vec8*>(wt + wt_part + brow_off + ct * 32 * (H2 * 2) + ((q ^ (c & 15)) << 4));
#pragma unroll
              for (int m = 0; m < 2; ++m) acc[m][ct] = mfma16(af[m], bfl, acc[m][ct]);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[m][ct] = mfma16(af[m], bfr, acc[m][ct]);
          }
        }
        load_words(words, pw, j + kDuv3TJ);  // next step's words into the registers this block has finished with
      }
    }

    // ---- epilogue: relu' of layer 1, g, row / column sums -------------------------------------------------------------
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float dvp[4][NT], vn[4][NT];
#pragma unroll
    for (int jq = 0; jq < 4; ++jq)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        dvp[jq][ct] = 0.0f;
        vn[jq][ct] = -(float)vraw[jq][ct];
      }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int is = 0; is < 4; ++is) {
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(&gsw[wave * 64 + 32 * m + 8 * is + 4 * h]);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
          for (int jq = 0; jq < 4; ++jq) {
            // (measured: the mask on g and the two sums as FMAs -- 4 vector instructions per element instead of 5 -- ran 3 %
            // SLOWER, 15.05 against 14.63 ms on one box: the FMAs chain through the sums, the multiply does not)
            const float e = (ureg[m][is][ct] > vn[jq][ct]) ? acc[m][ct][4 * is + jq] : 0.0f;
            const float ge = g4[jq] * e;
            duacc[m][is][ct] += ge;
            dvp[jq][ct] += ge;
          }
      }
    // ---- dV: the two lane halves hold different image rows; the four waves different row groups ------------------------
#pragma unroll
    for (int jq = 0; jq < 4; ++jq)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) dvp[jq][ct] += __shfl_xor(dvp[jq][ct], 32);
    __syncthreads();  // the previous step's readers of dvred are done
#pragma unroll
    for (int jq2 = 0; jq2 < 2; ++jq2)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        const int jq = 2 * h + jq2;  // lane half h hands over columns 2 h, 2 h + 1
        dvred[(wave * 4 + jq) * KC + 32 * ct + c] = h ? dvp[2 + jq2][ct] : dvp[jq2][ct];
      }
    __syncthreads();
    {
      const int jl = tid >> 6, kk = tid & 63;
      const float v = ((dvred[(0 * 4 + jl) * KC + kk] + dvred[(1 * 4 + jl) * KC + kk]) + dvred[(2 * 4 + jl) * KC + kk]) +
                      dvred[(3 * 4 + jl) * KC + kk];
      const int64_t gj = j + jl;
      if (gj < jhi && kc0 + kk < H1) dVslab[((int64_t)iblk * b + gj) * H1 + kc0 + kk] = v;
    }
  }

  // ---- dU of this (row block, k chunk, column split): every wave owns its rows -----------------------------------------
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int is = 0; is < 4; ++is) {
      const int64_t li = i0 + 16 * wave + 8 * m + 2 * is + h;
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
        if (li < b_rows && kc0 + 32 * ct + c < H1)
          dUslab[((int64_t)jsp * b_rows + li) * H1 + kc0 + 32 * ct + c] = duacc[m][is][ct];
    }
}

}  // namespace mi
