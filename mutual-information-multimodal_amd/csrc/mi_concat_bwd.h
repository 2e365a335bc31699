// Fused concat-MLP critic, backward, from the sign-bit images and scores saved by the forward.
//
// With g_p = d loss / d score of pair p = (i, j), M[p, n] = 1[Z2[p, n] > 0] (saved bits) and H1[p, k] = relu(U_i[k] + V_j[k]):
//   dZ2[p, n] = g_p w3[n] M[p, n]
//   dU_i[k]   = sum_j g_p 1[U_i[k] + V_j[k] > 0] E[p, k],   E[p, k] = sum_n M[p, n] (w3[n] W2[n, k])     ("duv" kernel)
//   dV_j[k]   = sum_i  (same summand)
//   D[n, k]   = sum_p M[p, n] g_p H1[p, k];   dW2 = w3[n] D[n, k];   dW3[n] = sum_k W2[n, k] D[n, k] + b2[n] m[n];
//   db2[n] = w3[n] m[n],  m[n] = sum_p g_p M[p, n];   db3 = sum_p g_p                                     ("dw2" kernel)
// Both big contractions run on MFMA with one operand expanded on the fly from the bit images (a 256-entry LDS table
// turns a byte of sign bits into an 8-element bf16 fragment of {0, 2}; the factor 2 is folded into g) and the other
// either LDS-resident (duv: a [128 k x H2] slice of w3*W2 stays in LDS for the whole column sweep) or generated in
// registers (dw2: g_p relu(U_i + V_j)).  All cross-workgroup sums go through slabs reduced in a fixed order.
#pragma once
#include "mi_common.h"
#include "mi_concat_fwd.h"

namespace mi {

// d loss / d score of pair (global row gi, column gj) times grad_out; 0 for dropped / out-of-range pairs
__device__ __forceinline__ float pair_grad(float score, int64_t gi, int64_t gj, int64_t sid_i, int64_t sid_j, float lse,
                                           float go, float gpos) {
  const int kind = pair_kind(gi, gj, sid_i, sid_j);
  if (kind == 1) return gpos;
  if (kind == 2) return go * expf(score - lse);
  return 0.0f;
}

// n index of slot m of the permuted hidden-unit order used by bitsP / W2wP:
//   m = ((pw * 2 + h) * 64 + q), pw = pass * 2 + wn, q = 16 a + r  ->  n = 128 pw + 32 a + (r & 3) + 8 (r >> 2) + 4 h
__host__ __device__ __forceinline__ int slot_to_n(int m) {
  const int q = m & 63, h = (m >> 6) & 1, pw = m >> 7;
  const int a = q >> 4, r = q & 15;
  return 128 * pw + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
}

// W2wP[k][m] = W2[n(m)][k] * w3[n(m)]   (transposed, permuted, scaled copy; [H1][H2])
template <typename OpT>
__global__ void prep_w2w_kernel(const float* __restrict__ w2, const float* __restrict__ w3, int H1, int H2,
                                OpT* __restrict__ out) {
  const int64_t total = (int64_t)H1 * H2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int k = (int)(e / H2), m = (int)(e % H2);
    const int n = slot_to_n(m);
    out[e] = (OpT)(w2[(int64_t)n * H1 + k] * w3[n]);
  }
}

// ---- fp16 mode (MI_PREC_F16, mi_concat_f16.h): power-of-two operand scales derived on the device from absmax slots ----
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
struct F16Scales {
  unsigned amax_bits[4];  // bit patterns of max|U|, max|V|, max|W2|, max|w3| (non-negative floats order as unsigned)
};
// s_g: the scale of g in the dW2 kernel's generated operand g' relu(u + v).  A power of two derived from the statistics so
// that the LARGEST |g| of the batch -- max(exp(neg_max - lse), 1 / n_pos), grad_out left out: it is applied to the finished
// sums -- lands in [2^lo_exp, 2^(lo_exp + 1)).  (A fixed 2^12 put the products of a 16.8 M-pair batch, g ~ 6e-8, into the
// fp16 subnormals.)  Every kernel derives it from the same statistics block with this function.
__device__ __forceinline__ float f16_g_scale(const mi_stats* st, int lo_exp);
// 2^e with 2^e * a in [2^lo_exp, 2^(lo_exp + 1)) for finite a > 0; 1 otherwise.  A pure function of the bits: every
// kernel derives the same scale from the same absmax slots.
__device__ __forceinline__ float f16_pow2_scale(float a, int lo_exp) {
  if (!(a > 0.0f) || !(a < __builtin_inff())) return 1.0f;
  int e;
  (void)frexpf(a, &e);  // a = m 2^e, m in [0.5, 1)
  return ldexpf(1.0f, lo_exp + 1 - e);
}
struct F16ScaleSet {
  float s_uv, s_w, s_ww;
  float s_uv3;  // two-part mode (MI_PREC_F16X3): no clamp trick there, U + V sits high in the fp16 range
};
__device__ __forceinline__ F16ScaleSet f16_scales(const F16Scales* sc) {
  const float au = __uint_as_float(sc->amax_bits[0]), av = __uint_as_float(sc->amax_bits[1]);
  const float aw = __uint_as_float(sc->amax_bits[2]), a3 = __uint_as_float(sc->amax_bits[3]);
  F16ScaleSet r;
  r.s_uv = f16_pow2_scale(au + av, -2);  // (max|U| + max|V|) s in [1/4, 1/2): the clamp at 1 is never reached
  r.s_w = f16_pow2_scale(aw, 13);
  r.s_ww = f16_pow2_scale(aw * a3, 13);
  r.s_uv3 = f16_pow2_scale(au + av, 13);
  return r;
}
__device__ __forceinline__ float f16_g_scale(const mi_stats* st, int lo_exp) {
  const float gneg = expf(st->neg_max - st->lse);  // neg_max = -inf (no negatives): 0
  const float gpos = 1.0f / (float)st->n_pos;
  return f16_pow2_scale(fmaxf(gneg, gpos), lo_exp);
}
// fp16 mode: g' h <= 2^14 / 2; two-part mode: h < 2^14, so g' < 2 keeps g' h below the fp16 maximum
constexpr int kF16GLo = 13, kF16X3GLo = 0;
// 16-bit operand traits of the bit-expanding kernels
template <typename OpT>
struct Op16 {
  using Vec8 = bf16x8;
};
template <>
struct Op16<f16_t> {
  using Vec8 = f16x8;
};
__device__ __forceinline__ f32x16 mfma16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma16(const f16x8& a, const f16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// ================================================================================================= duv kernel
template <typename OpT>
struct DuvCfg;
template <>
struct DuvCfg<bf16_t> {
  static constexpr int KC = 128;  // k columns per workgroup
  static constexpr int NT = 4;    // 32-wide column tiles per wave
  static constexpr int PADW = 8;  // row pad of the LDS weight slice in elements (1040-byte rows: conflict-free b128)
};
template <>
struct DuvCfg<f16_t> : DuvCfg<bf16_t> {};
template <>
struct DuvCfg<float> {
  static constexpr int KC = 32;
  static constexpr int NT = 1;
  static constexpr int PADW = 1;
};

constexpr int kDuvTI = 64;  // image rows per workgroup
constexpr int kDuvTJ = 8;   // text columns per step

// UvT: element type of U / V (fp32; fp16 in the fp16 mode, where relu' of layer 1 must be decided on the very values the
// forward added).  sc != null (fp16 mode): W2wP carries the scale s_ww, undone through g.
template <typename OpT, typename UvT = float>
__global__ __launch_bounds__(512) void concat_bwd_duv_kernel(
    const UvT* __restrict__ U, const UvT* __restrict__ V, const OpT* __restrict__ W2wP,
    const unsigned long long* __restrict__ bitsP, const float* __restrict__ S, const int64_t* __restrict__ sid_rows,
    const int64_t* __restrict__ sid_cols, const mi_stats* __restrict__ stats, const float* __restrict__ grad_out,
    int64_t b_rows, int64_t b, int64_t row_offset, int H1, int H2, int cols_per_split, int natural_order,
    float* __restrict__ dUslab /* [n_jsplit][b_rows][H1] */, float* __restrict__ dVslab /* [n_iblk][b][H1] */,
    const F16Scales* __restrict__ sc = nullptr) {
  using Cfg = DuvCfg<OpT>;
  using Vec8 = typename Op16<OpT>::Vec8;
  constexpr int KC = Cfg::KC, NT = Cfg::NT;
  constexpr bool kBf16 = sizeof(OpT) == 2;
  const int LDW = H2 + Cfg::PADW;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  OpT* wt = reinterpret_cast<OpT*>(smem_raw);                                   // [KC][LDW]
  size_t off = (((size_t)KC * LDW * sizeof(OpT)) + 15) & ~(size_t)15;
  Vec8* lut = reinterpret_cast<Vec8*>(smem_raw + off);                          // [256] (16-bit modes only)
  off += 256 * sizeof(Vec8);
  float* gs = reinterpret_cast<float*>(smem_raw + off);                         // [64][8]
  off += kDuvTI * kDuvTJ * sizeof(float);
  float* vneg = reinterpret_cast<float*>(smem_raw + off);                       // [8][KC]
  off += kDuvTJ * KC * sizeof(float);
  float* dvred = reinterpret_cast<float*>(smem_raw + off);                      // [4][8][KC]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c = lane & 31, h = lane >> 5;
  // XCD-aware order: the k chunks of one (row block, column split) run back to back on one XCD and share its sign words
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
  const int wpp = H2 / 64, hw = wpp / 2;  // 64-bit words per pair, per lane half

  const float go = grad_out ? grad_out[0] : 1.0f;
  const float lse = stats->lse;
  const float gpos = -go / (float)stats->n_pos;
  float gscale = kBf16 ? 0.5f : 1.0f;  // the 16-bit fragment table holds 2.0 for a set bit
  if (sc) gscale /= f16_scales(sc).s_ww;

  // ---- one-time setup: weight slice, table, U registers ----------------------------------------------------------
  {
    const int vec = 16 / (int)sizeof(OpT);          // elements per 16-byte vector
    const int vec_per_row = H2 / vec;
    for (int e = tid; e < KC * vec_per_row; e += 512) {
      const int row = e / vec_per_row, v = e % vec_per_row;
      u32x4 x = {0u, 0u, 0u, 0u};
      if (kc0 + row < H1) x = *reinterpret_cast<const u32x4*>(W2wP + (int64_t)(kc0 + row) * H2 + v * vec);
      if constexpr (kBf16) {
        *reinterpret_cast<u32x4*>(&wt[row * LDW + v * vec]) = x;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned bits = x[q];
          reinterpret_cast<float*>(wt)[row * LDW + v * 4 + q] = __builtin_bit_cast(float, bits);
        }
      }
    }
    if constexpr (kBf16) {
      if (tid < 256) {
        Vec8 f;
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = (OpT)(((tid >> q) & 1) ? 2.0f : 0.0f);
        lut[tid] = f;
      }
    }
  }
  float ureg[2][4][NT];
  float duacc[2][4][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int is = 0; is < 4; ++is) {
      int64_t li = i0 + 8 * wave + 4 * m + is;
      if (li >= b_rows) li = b_rows - 1;
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        const int k = kc0 + 32 * ct + c;
        ureg[m][is][ct] = (k < H1) ? (float)U[li * H1 + k] : 0.0f;
        duacc[m][is][ct] = 0.0f;
      }
    }
  __syncthreads();

  for (int64_t j = jlo; j < jhi; j += kDuvTJ) {
    // ---- (a) sign words of this lane's two pairs: issued first, they fly while the g / -V tiles are staged ---------
    unsigned long long words[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int64_t li = i0 + 8 * wave + 4 * m + (c >> 3), gj = j + (c & 7);
      const bool ok = li < b_rows && gj < jhi;
#pragma unroll
      for (int pw = 0; pw < 4; ++pw)
        words[m][pw] = (ok && pw < hw) ? bitsP[bitsp_index(li, gj, h, pw, (b + 31) / 32, hw)] : 0ull;
    }
    // ---- (b) g tile and -V tile -------------------------------------------------------------------------------------
    {
      const int il = tid >> 3, jl = tid & 7;
      const int64_t li = i0 + il, gj = j + jl;
      float g = 0.0f;
      if (li < b_rows && gj < jhi)
        g = gscale * pair_grad(S[li * b + gj], row_offset + li, gj, sid_rows[li], sid_cols[gj], lse, go, gpos);
      gs[il * kDuvTJ + jl] = g;
      for (int e = tid; e < kDuvTJ * KC; e += 512) {
        const int jl2 = e / KC, kk = e % KC;
        int64_t gj2 = j + jl2;
        if (gj2 >= b) gj2 = b - 1;
        vneg[e] = (kc0 + kk < H1) ? -(float)V[gj2 * H1 + kc0 + kk] : 0.0f;
      }
    }
    __syncthreads();

    // ---- (c) E[pair, k] = sum_n M[pair, n] W2w[n, k] on MFMA -------------------------------------------------------
    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][ct][r] = 0.0f;

#pragma unroll
    for (int pw = 0; pw < 4; ++pw) {
      if (pw < hw) {
        if constexpr (kBf16) {
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            Vec8 af[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) af[m] = lut[(unsigned)(words[m][pw] >> (8 * s)) & 0xFFu];
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
              const Vec8 bfr = *reinterpret_cast<const Vec8*>(&wt[(32 * ct + c) * LDW + (pw * 2 + h) * 64 + 8 * s]);
#pragma unroll
              for (int m = 0; m < 2; ++m) acc[m][ct] = mfma16(af[m], bfr, acc[m][ct]);
            }
          }
        } else {
          for (int s = 0; s < 64; ++s) {
            float af[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) af[m] = ((words[m][pw] >> s) & 1ull) ? 1.0f : 0.0f;
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
              const float bfr = reinterpret_cast<const float*>(wt)[(32 * ct + c) * LDW + (pw * 2 + h) * 64 + s];
#pragma unroll
              for (int m = 0; m < 2; ++m)
                acc[m][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m], bfr, acc[m][ct], 0, 0, 0);
            }
          }
        }
      }
    }

    // ---- (d) epilogue: relu' of layer 1, g, row / column sums -------------------------------------------------------
    float dvp[4][NT];
#pragma unroll
    for (int jq = 0; jq < 4; ++jq)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) dvp[jq][ct] = 0.0f;
    // column-tile outermost: only 4 -V values are live at a time
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
      float vn[4];
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) vn[jq] = vneg[(jq + 4 * h) * KC + 32 * ct + c];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int is = r >> 2, jq = r & 3;  // row rho = jq + 8 is + 4 h -> local row 4m + is, local column jq + 4h
          const float g = gs[(8 * wave + 4 * m + is) * kDuvTJ + jq + 4 * h];
          const float e = (ureg[m][is][ct] > vn[jq]) ? acc[m][ct][r] : 0.0f;
          duacc[m][is][ct] += g * e;
          dvp[jq][ct] += g * e;
        }
    }

    // ---- (e) dV: deterministic cross-wave reduction (waves w and w+4 share a slot, in that order) -----------------
    if (wave < 4) {
#pragma unroll
      for (int jq = 0; jq < 4; ++jq)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) dvred[(wave * kDuvTJ + jq + 4 * h) * KC + 32 * ct + c] = dvp[jq][ct];
    }
    __syncthreads();
    if (wave >= 4) {
#pragma unroll
      for (int jq = 0; jq < 4; ++jq)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) dvred[((wave - 4) * kDuvTJ + jq + 4 * h) * KC + 32 * ct + c] += dvp[jq][ct];
    }
    __syncthreads();
    for (int e = tid; e < kDuvTJ * KC; e += 512) {
      const int jl = e / KC, kk = e % KC;
      const int64_t gj = j + jl;
      if (gj < jhi && kc0 + kk < H1) {
        const float v = ((dvred[(0 * kDuvTJ + jl) * KC + kk] + dvred[(1 * kDuvTJ + jl) * KC + kk]) +
                         dvred[(2 * kDuvTJ + jl) * KC + kk]) + dvred[(3 * kDuvTJ + jl) * KC + kk];
        dVslab[((int64_t)iblk * b + gj) * H1 + kc0 + kk] = v;
      }
    }
    __syncthreads();  // gs / vneg / dvred are rewritten by the next step
  }

  // ---- dU of this (row block, k chunk, column split) -----------------------------------------------------------------
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int is = 0; is < 4; ++is) {
      const int64_t li = i0 + 8 * wave + 4 * m + is;
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        const float v = duacc[m][is][ct] + __shfl_xor(duacc[m][is][ct], 32);
        if (h == 0 && li < b_rows && kc0 + 32 * ct + c < H1)
          dUslab[((int64_t)jsp * b_rows + li) * H1 + kc0 + 32 * ct + c] = v;
      }
    }
}

// ================================================================================================= dw2 kernel
constexpr int kDw2IB = 128;  // image rows per g batch

// D slab [n_split][H2][H1]; workgroup = (k block of 256, n block of 256, pair split); wave (wn, wk): 128 n x 64 k.
// X3 (OpT = f16_t; MI_PREC_F16X3): U, V are the scaled copies Us, Vs; the generated operand g' relu(u + v) is formed in
// fp32 and split into two fp16 parts, two MFMAs per product (the bit operand {0, 2} is exact); grad_out is applied by the
// finishing kernel (the slabs hold 2 s_uv s_g D, as in the fp16 mode).
template <typename OpT, bool X3 = false>
__global__ __launch_bounds__(512) void concat_bwd_dw2_kernel(
    const float* __restrict__ U, const float* __restrict__ V, const unsigned* __restrict__ bitsN,
    const float* __restrict__ S, const int64_t* __restrict__ sid_rows, const int64_t* __restrict__ sid_cols,
    const mi_stats* __restrict__ stats, const float* __restrict__ grad_out, int64_t b_rows, int64_t b,
    int64_t row_offset, int H1, int H2, int rows_per_split, int natural_order, float* __restrict__ Dslab) {
  constexpr bool kBf16 = sizeof(OpT) == 2;
  constexpr int LDV = 36;  // floats per k row of the transposed V tile (32 columns + pad: conflict-free b128)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* vt = reinterpret_cast<float*>(smem_raw);                       // [256 k][LDV]
  float* gs = vt + 256 * LDV;                                           // [kDw2IB][32]
  using LutVec = typename Op16<OpT>::Vec8;
  LutVec* lut = reinterpret_cast<LutVec*>(gs + kDw2IB * 32);            // [256]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wn = wave >> 2, wk = wave & 3;
  const int c = lane & 31, h = lane >> 5;
  // XCD-aware order: the (k block, n block) tiles of one row split run on one XCD and share its sign words, U, V, S
  const int n_kb = (H1 + 255) / 256, n_nb = H2 / 256;
  const int n_split = (int)((b_rows + rows_per_split - 1) / rows_per_split);
  int tile, zsp;
  if (!xcd_decode(n_kb * n_nb, n_split, tile, zsp, natural_order)) return;
  const int kb0 = (tile % n_kb) * 256, nb0 = (tile / n_kb) * 256;
  const int64_t ilo = (int64_t)zsp * rows_per_split;
  int64_t ihi = ilo + rows_per_split;
  if (ihi > b_rows) ihi = b_rows;
  const int64_t JB = (b + 31) / 32;

  const float go = X3 ? 1.0f : (grad_out ? grad_out[0] : 1.0f);
  const float lse = stats->lse;
  const float gpos = -go / (float)stats->n_pos;
  const float gscale = X3 ? f16_g_scale(stats, kF16X3GLo) : (kBf16 ? 0.5f : 1.0f);

  if constexpr (kBf16) {
    if (tid < 256) {
      LutVec f;
#pragma unroll
      for (int q = 0; q < 8; ++q) f[q] = (OpT)(((tid >> q) & 1) ? 2.0f : 0.0f);
      lut[tid] = f;
    }
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

  for (int64_t jb = 0; jb < JB; ++jb) {
    const int64_t j0 = jb * 32;
    __syncthreads();  // previous column block's readers of vt / gs are done
    // V tile transposed: vt[k][j] = V[j0 + j][kb0 + k]
    for (int e = tid; e < 32 * 256; e += 512) {
      const int jl = e >> 8, kk = e & 255;
      int64_t gj = j0 + jl;
      if (gj >= b) gj = b - 1;
      vt[kk * LDV + jl] = (kb0 + kk < H1) ? V[gj * H1 + kb0 + kk] : 0.0f;
    }
    for (int64_t ib = ilo; ib < ihi; ib += kDw2IB) {
      __syncthreads();  // vt visible (first batch); previous batch's readers of gs are done
      for (int e = tid; e < kDw2IB * 32; e += 512) {
        const int il = e >> 5, jl = e & 31;
        const int64_t li = ib + il, gj = j0 + jl;
        float g = 0.0f;
        if (li < ihi && gj < b)
          g = gscale * pair_grad(S[li * b + gj], row_offset + li, gj, sid_rows[li], sid_cols[gj], lse, go, gpos);
        gs[e] = g;
      }
      __syncthreads();
      const int n_i = (int)((ihi - ib) < kDw2IB ? (ihi - ib) : kDw2IB);
#pragma unroll
      for (int s = 0; s < 2; ++s) {  // 16 columns per MFMA K step: this lane half takes columns 16 s + 8 h + [0, 8)
        float vreg[2][8];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(&vt[(64 * wk + 32 * t + c) * LDV + 16 * s + 8 * h]);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(&vt[(64 * wk + 32 * t + c) * LDV + 16 * s + 8 * h + 4]);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            vreg[t][q] = v0[q];
            vreg[t][4 + q] = v1[q];
          }
        }
        // software prefetch of the next row's sign words and U values
        unsigned wnext[4];
        float unext[2];
        {
          const int64_t li = ib;
#pragma unroll
          for (int a = 0; a < 4; ++a) wnext[a] = bitsN[(li * JB + jb) * H2 + nb0 + 128 * wn + 32 * a + c];
#pragma unroll
          for (int t = 0; t < 2; ++t) unext[t] = kok[t] ? U[li * H1 + kcol[t]] : 0.0f;
        }
        for (int il = 0; il < n_i; ++il) {
          unsigned wcur[4];
          float ucur[2];
#pragma unroll
          for (int a = 0; a < 4; ++a) wcur[a] = wnext[a];
#pragma unroll
          for (int t = 0; t < 2; ++t) ucur[t] = unext[t];
          if (il + 1 < n_i) {
            const int64_t li = ib + il + 1;
#pragma unroll
            for (int a = 0; a < 4; ++a) wnext[a] = bitsN[(li * JB + jb) * H2 + nb0 + 128 * wn + 32 * a + c];
#pragma unroll
            for (int t = 0; t < 2; ++t) unext[t] = kok[t] ? U[li * H1 + kcol[t]] : 0.0f;
          }
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(&gs[il * 32 + 16 * s + 8 * h]);
          const f32x4 g1 = *reinterpret_cast<const f32x4*>(&gs[il * 32 + 16 * s + 8 * h + 4]);
          if constexpr (X3) {
            f16x8 ph[2], pl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                const float pv = fmaxf(ucur[t] + vreg[t][q], 0.0f) * (q < 4 ? g0[q] : g1[q - 4]);
                const f16_t hi = (f16_t)pv;
                ph[t][q] = hi;
                pl[t][q] = (f16_t)(pv - (float)hi);
              }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
              const f16x8 mf = lut[(wcur[a] >> (16 * s + 8 * h)) & 0xFFu];
#pragma unroll
              for (int t = 0; t < 2; ++t) {
                acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(mf, pl[t], acc[a][t], 0, 0, 0);
                acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(mf, ph[t], acc[a][t], 0, 0, 0);
              }
            }
          } else if constexpr (kBf16) {
            bf16x8 hf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                hf[t][q] = (bf16_t)(fmaxf(ucur[t] + vreg[t][q], 0.0f) * g0[q]);
                hf[t][4 + q] = (bf16_t)(fmaxf(ucur[t] + vreg[t][4 + q], 0.0f) * g1[q]);
              }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
              const bf16x8 mf = lut[(wcur[a] >> (16 * s + 8 * h)) & 0xFFu];
#pragma unroll
              for (int t = 0; t < 2; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf, hf[t], acc[a][t], 0, 0, 0);
            }
          } else {
            // fp32 MFMA: K = 2 per instruction; lane half h takes column 16 s + 2 q + h of step q
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              // columns of this step: h = 0 -> 16 s + 8*0 + q ... keep the bf16 slot order: slot (h, q) = column 16 s + 8 h + q
              const float gq = (q < 4) ? g0[q] : g1[q - 4];
              float hv[2];
#pragma unroll
              for (int t = 0; t < 2; ++t) hv[t] = fmaxf(ucur[t] + vreg[t][q], 0.0f) * gq;
#pragma unroll
              for (int a = 0; a < 4; ++a) {
                const float mf = ((wcur[a] >> (16 * s + 8 * h + q)) & 1u) ? 1.0f : 0.0f;
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(mf, hv[t], acc[a][t], 0, 0, 0);
              }
            }
          }
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

// ================================================================================================= small kernels
// m[n] partial = sum over a slice of rows of g_p M[p, n]; also the slice's sum of g_p (db3).  One workgroup per row
// slice, one thread per hidden unit n (512 threads -> H2 <= 512 per pass over n).
__global__ __launch_bounds__(512) void concat_bwd_db2_kernel(const unsigned* __restrict__ bitsN,
                                                             const float* __restrict__ S,
                                                             const int64_t* __restrict__ sid_rows,
                                                             const int64_t* __restrict__ sid_cols,
                                                             const mi_stats* __restrict__ stats,
                                                             const float* __restrict__ grad_out, int64_t b_rows, int64_t b,
                                                             int64_t row_offset, int H2, int rows_per_split,
                                                             float* __restrict__ mslab /* [n_split][H2] */,
                                                             float* __restrict__ gsum /* [n_split] */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* grow = reinterpret_cast<float*>(smem_raw);  // [32 * JB]
  __shared__ float red[8];
  const int tid = threadIdx.x;
  const int64_t JB = (b + 31) / 32;
  const int64_t ilo = (int64_t)blockIdx.x * rows_per_split;
  int64_t ihi = ilo + rows_per_split;
  if (ihi > b_rows) ihi = b_rows;
  const float go = grad_out ? grad_out[0] : 1.0f;
  const float lse = stats->lse;
  const float gpos = -go / (float)stats->n_pos;
  float gtot = 0.0f;
  float macc[2] = {0.0f, 0.0f};  // hidden units tid and tid + 512
  for (int64_t li = ilo; li < ihi; ++li) {
    __syncthreads();
    const int64_t si = sid_rows[li];
    for (int64_t gj = tid; gj < JB * 32; gj += 512) {
      float g = 0.0f;
      if (gj < b) g = pair_grad(S[li * b + gj], row_offset + li, gj, si, sid_cols[gj], lse, go, gpos);
      grow[gj] = g;
      gtot += g;
    }
    __syncthreads();
    for (int pass = 0; pass * 512 < H2; ++pass) {
      const int n = pass * 512 + tid;
      if (n < H2) {
        float a = 0.0f;
        for (int64_t jb = 0; jb < JB; ++jb) {
          const unsigned w = bitsN[(li * JB + jb) * H2 + n];
          const float* gr = grow + jb * 32;
#pragma unroll
          for (int q = 0; q < 32; ++q) a += ((w >> q) & 1u) ? gr[q] : 0.0f;
        }
        macc[pass & 1] += a;
      }
    }
  }
  for (int pass = 0; pass * 512 < H2 && pass < 2; ++pass) {
    const int n = pass * 512 + tid;
    if (n < H2) mslab[(int64_t)blockIdx.x * H2 + n] = macc[pass];
  }
  gtot = wave_sum(gtot);
  if ((tid & 63) == 0) red[tid >> 6] = gtot;
  __syncthreads();
  if (tid == 0) {
    float t = 0.0f;
    for (int w = 0; w < 8; ++w) t += red[w];
    gsum[blockIdx.x] = t;
  }
}

// One workgroup per hidden unit n: D[n, :] = sum of slabs (fixed order); dW2[n, :] = w3[n] D; dW3[n], db2[n]; n == 0: db3.
__global__ __launch_bounds__(256) void concat_bwd_finish_w2_kernel(const float* __restrict__ Dslab, int n_dsplit,
                                                                   const float* __restrict__ mslab,
                                                                   const float* __restrict__ gsum, int n_msplit,
                                                                   const float* __restrict__ w2, const float* __restrict__ b2,
                                                                   const float* __restrict__ w3, int H1, int H2,
                                                                   float* __restrict__ dW2, float* __restrict__ dW3,
                                                                   float* __restrict__ db2, float* __restrict__ db3,
                                                                   const F16Scales* __restrict__ sc = nullptr,
                                                                   const float* __restrict__ grad_out = nullptr,
                                                                   const mi_stats* __restrict__ stats = nullptr,
                                                                   int x3 = 0) {
  __shared__ float red[4];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float w3n = w3[n];
  // fp16 mode: the slabs hold 2 s_uv s_g D computed for grad_out = 1 (mi_concat_f16.h)
  float dscale = 1.0f;
  if (sc) {
    const F16ScaleSet ss = f16_scales(sc);
    dscale = (grad_out ? grad_out[0] : 1.0f) /
             (2.0f * f16_g_scale(stats, x3 ? kF16X3GLo : kF16GLo) * (x3 ? ss.s_uv3 : ss.s_uv));
  }
  float dot = 0.0f;
  for (int k = tid; k < H1; k += 256) {
    float d = 0.0f;
    for (int s = 0; s < n_dsplit; ++s) d += Dslab[((int64_t)s * H2 + n) * H1 + k];
    d *= dscale;
    dW2[(int64_t)n * H1 + k] = w3n * d;
    dot += w2[(int64_t)n * H1 + k] * d;
  }
  // the row-slice sums of the db2 kernel (up to 1,024 of them): thread t adds slices t, t + 256, ..., then the fixed block
  // reduction below.  (One thread walking all of them -- 1,024 dependent strided loads per workgroup -- was most of this
  // kernel's 0.3 ms.)
  float mp = 0.0f, gp = 0.0f;
  for (int s = tid; s < n_msplit; s += 256) mp += mslab[(int64_t)s * H2 + n];
  if (n == 0)
    for (int s = tid; s < n_msplit; s += 256) gp += gsum[s];
  dot = wave_sum(dot);
  mp = wave_sum(mp);
  gp = wave_sum(gp);
  __shared__ float redm[4], redg[4];
  if ((tid & 63) == 0) {
    red[tid >> 6] = dot;
    redm[tid >> 6] = mp;
    redg[tid >> 6] = gp;
  }
  __syncthreads();
  if (tid == 0) {
    const float total = (red[0] + red[1]) + (red[2] + red[3]);
    const float m = (redm[0] + redm[1]) + (redm[2] + redm[3]);
    dW3[n] = total + b2[n] * m;
    db2[n] = w3n * m;
    if (n == 0) db3[0] = (redg[0] + redg[1]) + (redg[2] + redg[3]);
  }
}

// out[r, k] = sum_s slab[s][r][k]   (fixed order); optional column sums of the result are done by colsum_kernel
__global__ void slab_reduce_kernel(const float* __restrict__ slab, int n_slab, int64_t rows, int64_t cols,
                                   float* __restrict__ out) {
  const int64_t total = rows * cols / 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    f32x4 a = *reinterpret_cast<const f32x4*>(slab + 4 * e);
    for (int s = 1; s < n_slab; ++s) a += *reinterpret_cast<const f32x4*>(slab + (int64_t)s * rows * cols + 4 * e);
    *reinterpret_cast<f32x4*>(out + 4 * e) = a;
  }
}

// column sums in two deterministic stages: grid (cols/64, 64 row chunks), 256 threads = 64 columns x 4 row groups
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ m, int64_t rows, int64_t cols,
                                                             float* __restrict__ part /* [gridDim.y][cols] */) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int64_t k = (int64_t)blockIdx.x * 64 + cl;
  const int64_t chunk = (rows + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * chunk;
  int64_t r1 = r0 + chunk;
  if (r1 > rows) r1 = rows;
  float a = 0.0f;
  if (k < cols)
    for (int64_t r = r0 + rg; r < r1; r += 4) a += m[r * cols + k];
  red[rg][cl] = a;
  __syncthreads();
  if (rg == 0 && k < cols) part[(int64_t)blockIdx.y * cols + k] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
__global__ void colsum_final_kernel(const float* __restrict__ part, int n_part, int64_t cols, float* __restrict__ out) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cols) return;
  float a = 0.0f;
  for (int p = 0; p < n_part; ++p) a += part[(int64_t)p * cols + k];
  out[k] = a;
}

}  // namespace mi
