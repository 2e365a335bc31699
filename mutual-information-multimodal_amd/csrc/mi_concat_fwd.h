// Fused concat-MLP critic, forward:  S[i,j] = w3 . relu(W2 relu(U_i + V_j) + b2) + b3
//   reference: create_mi_pairs (main_utils.py:80-110) -> make_mlp(1536,[1024,512]) (model.py:18-32, main_utils.py:77)
//   with the first Linear factorised: W1 [x_i ; y_j] + b1 = U_i + V_j, U = X W1x^T, V = Y W1y^T + b1.
//
// A workgroup owns a pair tile of 8 image rows x 32 text columns = 256 pairs and runs H2 / (128 NWN) passes; in a pass it
// accumulates Z2^T[n, pair] for 128 NWN hidden units n over all H1 = K:
//     A operand (MFMA rows  = n)    : W2[n, k]  streamed HBM/L2 -> registers -> LDS (double-buffered k tiles)
//     B operand (MFMA cols  = pair) : H1[pair, k] = relu(U_i[k] + V_j[k]) generated in registers from small U/V tiles
// so the [B^2, H1] activation matrix of the reference (34 GB in bf16 at B = 4096) never exists.  The epilogue applies
// relu and the dot with w3 in registers (b2 is the accumulator's initial value), and, when gradients are needed,
// emits the sign pattern of Z2 as two bit images used by the backward kernels (1 bit per (pair, n), twice):
//     bitsP: one 64-bit word per (pair, lane half h, pw = n / 128); bit q = 16*a + r  <->  n = 128*pw + 32*a +
//            (r & 3) + 8*(r >> 2) + 4*h        (lane-local MFMA accumulator order; consumed by mi_concat_bwd dU/dV).
//            Stored as [row][32-column block][pw][h][column % 32] (bitsp_index): the 64 lanes of a wave hold 32
//            consecutive columns x 2 halves of one (row, pw), so one store instruction writes 512 contiguous bytes.
//            (The earlier per-pair layout [pair][h][pw] made every store a scatter of 8-byte pieces into lines that
//            were completed a pass later: 16.8 GB of read-for-ownership fetches and 5.6 GB of writes per forward at
//            B = 4096 for 1 GB of payload -- profiles/r1_c_pmc_traffic.json.)
//     bitsN: per (row i, 32-column block), one 32-bit word per n; bit q <-> column 32*block + q   (consumed by dW2)
// Wave (wn, wp): hidden units [128 wn, +128) of the pass, pairs of local rows {2 wp, 2 wp + 1} x 32 columns:
// 4 x 2 MFMA 32x32 tiles, 128 accumulator registers.
//
// NWN = 2: 512 threads, 256 hidden units per pass, ~100 KB of LDS -> one workgroup per CU (both waves of a SIMD belong
//          to the same workgroup and run in lock-step between its barriers).
// NWN = 1: 256 threads, 128 hidden units per pass, ~59 KB of LDS -> TWO workgroups per CU: the two waves of a SIMD
//          belong to different workgroups with independent barriers, so one computes while the other waits.
//          Same W2 bytes streamed per pair (each pass streams half the rows, twice as many passes).
#pragma once
#include <utility>

#include "mi_common.h"

namespace mi {
// word index into bitsP; jb32 = ceil(B / 32), hw = H2 / 128
__host__ __device__ __forceinline__ int64_t bitsp_index(int64_t li, int64_t gj, int h, int pw, int64_t jb32, int hw) {
  return ((((li * jb32 + (gj >> 5)) * hw + pw) * 2 + h) << 5) + (gj & 31);
}
static inline int64_t bitsp_words(int64_t b_rows, int64_t b, int64_t h2) {
  return b_rows * ((b + 31) / 32) * (h2 / 128) * 64;
}
}  // namespace mi

namespace mi {

template <typename OpT>
struct FwdCfg;
template <>
struct FwdCfg<bf16_t> {
  static constexpr int KT = 64;     // k per staged tile
  static constexpr int KSTEP = 16;  // k per MFMA
  static constexpr int LDW = 72;    // W2 tile row pitch in elements (144 B: ds_read_b128 conflict-free)
  static constexpr int LDUV = 68;   // U/V tile row pitch in floats (272 B)
};
template <>
struct FwdCfg<float> {
  static constexpr int KT = 32;
  static constexpr int KSTEP = 2;
  static constexpr int LDW = 33;
  static constexpr int LDUV = 33;
};

#define LANE_OF_REG(r) (((r) & 3) + 8 * ((r) >> 2))

// compile-time unrolled loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N-1>)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}
constexpr int kFwdTI = 8;    // image rows per pair tile
constexpr int kFwdTJ = 32;   // text columns per pair tile

template <typename OpT, int NWN>
struct FwdSmem {
  using Cfg = FwdCfg<OpT>;
  static constexpr int NP = 128 * NWN;  // hidden units per pass
  OpT w[2][NP * Cfg::LDW];
  float v[2][kFwdTJ * Cfg::LDUV];
  float u[2][kFwdTI * Cfg::LDUV];
  float b2[NP];
  float w3[NP];
  float sred[2][kFwdTI * kFwdTJ];  // [wn][pair]
};

// relu(u + v) for 8 consecutive k, packed to bf16 (RNE).  relu on the packed pair as a signed 16-bit max with 0:
// negative bf16 values have the sign bit set, i.e. are negative int16.
__device__ __forceinline__ bf16x8 gen_h1_bf16(const f32x4& u0, const f32x4& u1, const f32x4& v0, const f32x4& v1) {
  union {
    bf16x8 v;
    s16x2 s[4];
  } o;
  const f32x4 a = u0 + v0, b = u1 + v1;
  const s16x2 zero = {0, 0};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    f32x2 p = {a[2 * q], a[2 * q + 1]};
    o.s[q] = __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(p, bf16x2)), zero);
    f32x2 p2 = {b[2 * q], b[2 * q + 1]};
    o.s[2 + q] = __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(p2, bf16x2)), zero);
  }
  return o.v;
}

// Lanes `lane_lo` and `lane_lo + 4` of `word` receive the low / high half of a wave-uniform 64-bit ballot.
// v_writelane_b32 has no builtin in this toolchain; the leading s_nop covers the VALU-writes-SGPR -> lane-op hazard
// (hipcc pads nothing inside an asm statement).
template <int LANE_LO>
__device__ __forceinline__ unsigned ballot_to_lanes(unsigned word, unsigned long long bal) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)bal);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(bal >> 32));
  asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %3\n\tv_writelane_b32 %0, %2, %4"
               : "+v"(word)
               : "s"(lo), "s"(hi), "n"(LANE_LO), "n"(LANE_LO + 4));
  return word;
}

// ---- epilogue of a pass (shared by the register-staged and the LDS-DMA kernel) --------------------------------------
// Straight-line, no SGPR hand-overs.  (The first version branched per accumulator register -- `s += pos ? z * w3 : 0`
// became an exec-masked block with its own LDS read and wait -- and built the column words from 128 ballots moved into
// lanes with 256 v_writelane: ~3000 instructions and 128 exposed LDS latencies per row of the tile and pass.)
//   s      : zc = max(z, 0); s += zc * w3          (w3 of the lane's 64 hidden units: 16 ds_read_b128 issued up front)
//   bitsP  : bit = (zc != 0) = sign of (0 - bits(zc)), shifted into the word with v_alignbit (registers taken from the
//            last to the first, so that register r ends at bit 16 (a & 1) + r)
//   bitsN  : the 32 x 32 bit transpose of the bitsP words of the 32 columns of a half-wave (5 butterfly stages over
//            ds_swizzle: stage j exchanges bit-index bit j with lane-index bit j), i.e. lane c ends with the column word
//            of the hidden unit of ITS bit position q = c -- instead of a ballot per accumulator register.
struct BitTransposeLane {
  unsigned msel[5];  // bits kept from the lane's own word at stage j = 16 >> s
  unsigned amt[5];   // right-rotation of the partner's word
};
__device__ __forceinline__ BitTransposeLane bit_transpose_lane(int c) {
  BitTransposeLane k;
  constexpr unsigned low[5] = {0x0000FFFFu, 0x00FF00FFu, 0x0F0F0F0Fu, 0x33333333u, 0x55555555u};
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int j = 16 >> s;
    const bool lo = (c & j) == 0;
    k.msel[s] = lo ? low[s] : ~low[s];
    k.amt[s] = lo ? 32 - j : j;
  }
  return k;
}
template <int S>
__device__ __forceinline__ unsigned bit_transpose_stage(unsigned w, const BitTransposeLane& k) {
  constexpr int J = 16 >> S;
  const unsigned other = (unsigned)__builtin_amdgcn_ds_swizzle((int)w, (J << 10) | 0x1f);  // lane ^ J within 32 lanes
  const unsigned rot = __builtin_amdgcn_alignbit(other, other, k.amt[S]);
  return (w & k.msel[S]) | (rot & ~k.msel[S]);
}
__device__ __forceinline__ unsigned bit_transpose32(unsigned w, const BitTransposeLane& k) {
  w = bit_transpose_stage<0>(w, k);
  w = bit_transpose_stage<1>(w, k);
  w = bit_transpose_stage<2>(w, k);
  w = bit_transpose_stage<3>(w, k);
  return bit_transpose_stage<4>(w, k);
}

// acc[a][r] of ONE row t of the tile: hidden unit 32 a + (r & 3) + 8 (r >> 2) + 4 h of the wave's 128, column c.
// Returns this lane's partial score; stores the row's bitsP word and the two bitsN words of the lane when `bits`.
__device__ __forceinline__ float fwd_epilogue_row(const f32x16& a0, const f32x16& a1, const f32x16& a2, const f32x16& a3,
                                                  const float* w3w /* LDS: the wave's 128 values of w3 */, int h, int c,
                                                  bool bits, bool row_ok, bool col_ok, unsigned long long* bitsP_word,
                                                  unsigned* bitsN_row /* + hidden unit */, const BitTransposeLane& k) {
  f32x4 w3v[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int g = 0; g < 4; ++g) w3v[a][g] = *reinterpret_cast<const f32x4*>(w3w + a * 32 + 8 * g + 4 * h);
  const f32x16* acc[4] = {&a0, &a1, &a2, &a3};
  float s0 = 0.0f, s1 = 0.0f;
  unsigned pw[2] = {0u, 0u};
#pragma unroll
  for (int a = 3; a >= 0; --a)
#pragma unroll
    for (int r = 15; r >= 0; --r) {
      const float zc = fmaxf((*acc[a])[r], 0.0f);
      if (r & 1) s1 = fmaf(zc, w3v[a][r >> 2][r & 3], s1);
      else s0 = fmaf(zc, w3v[a][r >> 2][r & 3], s0);
      pw[a >> 1] = __builtin_amdgcn_alignbit(pw[a >> 1], 0u - __builtin_bit_cast(unsigned, zc), 31);
    }
  if (bits) {
    if (row_ok && col_ok) *bitsP_word = ((unsigned long long)pw[1] << 32) | pw[0];
    const int nn = ((c & 15) & 3) + 8 * ((c & 15) >> 2) + 4 * h;
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const unsigned t = bit_transpose32(pw[w], k);
      if (row_ok) bitsN_row[(2 * w + (c >> 4)) * 32 + nn] = t;
    }
  }
  return s0 + s1;
}

template <typename OpT, int NWN>
__global__ __launch_bounds__(256 * NWN, 2) /* 2 waves per SIMD: <= 256 registers */ void concat_fwd_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                                               const OpT* __restrict__ W2, const float* __restrict__ b2,
                                                               const float* __restrict__ w3, const float* __restrict__ b3,
                                                               int64_t b_rows, int64_t b, int H1, int H2,
                                                               float* __restrict__ S,
                                                               unsigned long long* __restrict__ bitsP,
                                                               unsigned* __restrict__ bitsN) {
  using Cfg = FwdCfg<OpT>;
  using Smem = FwdSmem<OpT, NWN>;
  constexpr int KT = Cfg::KT, KSTEP = Cfg::KSTEP, LDW = Cfg::LDW, LDUV = Cfg::LDUV;
  constexpr int NTHR = 256 * NWN, NP = Smem::NP;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  Smem& sm = *reinterpret_cast<Smem*>(smem_raw);

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wn = wave >> 2, wp = wave & 3;
  const int c = lane & 31, h = lane >> 5;
  const BitTransposeLane btl = bit_transpose_lane(c);
  const int64_t i0 = (int64_t)blockIdx.y * kFwdTI, j0 = (int64_t)blockIdx.x * kFwdTJ;
  const int n_pass = H2 / NP;
  const int n_kt = H1 / KT;
  const int64_t JB = (b + 31) / 32;

  // ---- staging assignment (global -> registers -> LDS) ---------------------------------------------------------
  // W2 tile: NP rows x 128 bytes, 4 16-byte vectors per thread.  V tile: 32 rows, U tile: 8 rows of KT floats.
  constexpr int VEC_UV = KT / 4;                                     // float4 per U/V row
  constexpr int NV = (kFwdTJ * VEC_UV + NTHR - 1) / NTHR;            // V vectors per thread (1 or 2)
  const bool has_u = tid < kFwdTI * VEC_UV;
  const float* vsrc[NV];
  int vdst[NV];
  bool has_v[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int e = tid + NTHR * q;
    has_v[q] = e < kFwdTJ * VEC_UV;
    const int row = (e / VEC_UV) % kFwdTJ, kv = e % VEC_UV;
    int64_t gj = j0 + row;
    if (gj >= b) gj = b - 1;  // clamp: the pair is discarded in the epilogue
    vsrc[q] = V + gj * H1 + kv * 4;
    vdst[q] = row * LDUV + kv * 4;
  }
  const int u_row = (tid / VEC_UV) % kFwdTI, u_kv = tid % VEC_UV;
  int64_t urow_g = i0 + u_row;
  if (urow_g >= b_rows) urow_g = b_rows - 1;
  const float* usrc = U + urow_g * H1 + u_kv * 4;

  u32x4 rw[4];
  f32x4 rv[NV], ru;

  auto stage_load = [&](int pass, int kt) {
    const int64_t k0 = (int64_t)kt * KT;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int vidx = tid + NTHR * q;
      const int row = vidx >> 3, kv = vidx & 7;
      const char* src = reinterpret_cast<const char*>(W2 + ((int64_t)(pass * NP + row)) * H1 + k0) + kv * 16;
      rw[q] = *reinterpret_cast<const u32x4*>(src);
    }
#pragma unroll
    for (int q = 0; q < NV; ++q)
      if (has_v[q]) rv[q] = *reinterpret_cast<const f32x4*>(vsrc[q] + k0);
    if (has_u) ru = *reinterpret_cast<const f32x4*>(usrc + k0);
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int vidx = tid + NTHR * q;
      const int row = vidx >> 3, kv = vidx & 7;
      if constexpr (sizeof(OpT) == 2) {
        *reinterpret_cast<u32x4*>(&sm.w[buf][row * LDW + kv * 8]) = rw[q];
      } else {
        float* dst = reinterpret_cast<float*>(&sm.w[buf][row * LDW + kv * 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned bits = rw[q][e];  // copy first: bit_cast of a vector-element lvalue miscompiles (ROCm 7.2)
          dst[e] = __builtin_bit_cast(float, bits);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      if (has_v[q]) {
        if constexpr (sizeof(OpT) == 2) {
          *reinterpret_cast<f32x4*>(&sm.v[buf][vdst[q]]) = rv[q];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) sm.v[buf][vdst[q] + e] = rv[q][e];
        }
      }
    }
    if (has_u) {
      if constexpr (sizeof(OpT) == 2) {
        *reinterpret_cast<f32x4*>(&sm.u[buf][u_row * LDUV + u_kv * 4]) = ru;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) sm.u[buf][u_row * LDUV + u_kv * 4 + e] = ru[e];
      }
    }
  };

  float s_total[2] = {0.0f, 0.0f};  // per pair-tile t: this lane's partial score (summed over passes)

  for (int pass = 0; pass < n_pass; ++pass) {
    __syncthreads();  // every wave has left the previous pass's epilogue (it reads sm.w3)
    // b2 / w3 of this pass's hidden units
    if (tid < NP) {
      sm.b2[tid] = b2[pass * NP + tid];
      sm.w3[tid] = w3[pass * NP + tid];
    }
    stage_load(pass, 0);
    __syncthreads();  // b2/w3 visible; previous pass's LDS reads finished
    stage_store(0);

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float bias = sm.b2[wn * 128 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
        acc[a][0][r] = bias;
        acc[a][1][r] = bias;
      }
    __syncthreads();

    for (int kt = 0; kt < n_kt; ++kt) {
      const int buf = kt & 1;
      const bool more = kt + 1 < n_kt;
      if (more) stage_load(pass, kt + 1);
      const OpT* wt = sm.w[buf];
      const float* vt = sm.v[buf];
      const float* ut = sm.u[buf];
#pragma unroll
      for (int kk = 0; kk < KT / KSTEP; ++kk) {
        if constexpr (sizeof(OpT) == 2) {
          const int ko = kk * 16 + 8 * h;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(&vt[c * LDUV + ko]);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(&vt[c * LDUV + ko + 4]);
          bf16x8 hf[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(&ut[(2 * wp + t) * LDUV + ko]);
            const f32x4 u1 = *reinterpret_cast<const f32x4*>(&ut[(2 * wp + t) * LDUV + ko + 4]);
            hf[t] = gen_h1_bf16(u0, u1, v0, v1);
          }
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(&wt[(wn * 128 + a * 32 + c) * LDW + ko]);
#pragma unroll
            for (int t = 0; t < 2; ++t)
              acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, hf[t], acc[a][t], 0, 0, 0);
          }
        } else {
          const int ko = kk * 2 + h;
          const float vv = vt[c * LDUV + ko];
          float hv[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) hv[t] = fmaxf(ut[(2 * wp + t) * LDUV + ko] + vv, 0.0f);
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const float wf = wt[(wn * 128 + a * 32 + c) * LDW + ko];
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf, hv[t], acc[a][t], 0, 0, 0);
          }
        }
      }
      if (more) stage_store(buf ^ 1);
      __syncthreads();
    }

    // ---- epilogue of the pass: relu, dot with w3, sign bits (fwd_epilogue_row) ----------------------------------------
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int64_t li = i0 + 2 * wp + t;   // local image row of this pair tile
      const int64_t gj = j0 + c;            // text column of this lane
      const int pw = pass * NWN + wn;       // 128-wide hidden-unit group
      const bool row_ok = li < b_rows, col_ok = gj < b;
      const int64_t lic = row_ok ? li : 0, gjc = col_ok ? gj : 0;
      s_total[t] += fwd_epilogue_row(acc[0][t], acc[1][t], acc[2][t], acc[3][t], &sm.w3[wn * 128], h, c, bitsP != nullptr,
                                     row_ok, col_ok,
                                     bitsP + bitsp_index(lic, gjc, h, pw, (b + 31) / 32, (int)(H2 / 128)),
                                     bitsN + (lic * JB + blockIdx.x) * H2 + pw * 128, btl);
    }
  }

  // ---- combine: halves (h), hidden-unit waves (wn) -> score -----------------------------------------------------
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float s = s_total[t] + __shfl_xor(s_total[t], 32);
    if (h == 0) sm.sred[wn][(2 * wp + t) * 32 + c] = s;
  }
  __syncthreads();
  if (tid < kFwdTI * kFwdTJ) {
    const int il = tid >> 5, jl = tid & 31;
    const int64_t li = i0 + il, gj = j0 + jl;
    if (li < b_rows && gj < b) {
      float s = sm.sred[0][tid];
      if (NWN == 2) s += sm.sred[1][tid];
      S[li * b + gj] = s + b3[0];
    }
  }
}

}  // namespace mi
