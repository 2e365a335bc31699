// Fused concat-MLP critic forward, bf16, LDS-DMA staging variant of concat_fwd_kernel (mi_concat_fwd.h): same tiling,
// same outputs (scores, bitsP, bitsN), but the W2 / U / V tiles go L2 -> LDS with global_load_lds_dwordx4.
//
// Why: in the register-staged kernel the 24 staging registers push the wave to the 256-register limit; hipcc then sinks
// every W fragment read directly in front of its two MFMAs (R, wait, MM, R, wait, MM ...), exposing the LDS latency
// four times per 16-wide K step (46 % of the wave cycles were spent in s_waitcnt).  Without staging registers all ten
// fragment reads of a step are issued together, ahead of the operand generation, behind one wait.
//
// LDS images are linear (the DMA writes a wave's 64 x 16 bytes contiguously); bank conflicts are removed by an XOR
// swizzle applied on the DMA source address and again on the read (guide rule 21):
//   W tile [NP rows][128 B]: 16-byte chunk c of row r at chunk position c ^ ((r >> 1) & 7)
//   V tile [32 rows][256 B]: chunk c of row r at position c ^ (r & 15);  U tile [8 rows][256 B]: unswizzled (its reads
//   are broadcasts).
#pragma once
#include "mi_concat_fwd.h"

namespace mi {

#define MI_GLDS16(src, dst)                                                                                   \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                      \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

template <int NWN>
struct FwdDmaSmem {
  static constexpr int NP = 128 * NWN;
  static constexpr int W_BYTES = NP * 128;
  static constexpr int V_BYTES = kFwdTJ * 256;
  static constexpr int U_BYTES = kFwdTI * 256;
  static constexpr int BUF_BYTES = W_BYTES + V_BYTES + U_BYTES;
  static constexpr int TOTAL = 2 * BUF_BYTES + 2 * NP * 4 + 2 * kFwdTI * kFwdTJ * 4;
};

template <int NWN>
__global__ __launch_bounds__(256 * NWN, 2) void concat_fwd_dma_kernel(
    const float* __restrict__ U, const float* __restrict__ V, const bf16_t* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ w3, const float* __restrict__ b3, int64_t b_rows, int64_t b, int H1, int H2,
    float* __restrict__ S, unsigned long long* __restrict__ bitsP, unsigned* __restrict__ bitsN, int natural_order) {
  using L = FwdDmaSmem<NWN>;
  constexpr int NP = L::NP, NWAVES = 4 * NWN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* b2s = reinterpret_cast<float*>(smem + 2 * L::BUF_BYTES);
  float* w3s = b2s + NP;
  float* sred = w3s + NP;  // [2][256]

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wn = wave >> 2, wp = wave & 3;
  const int c = lane & 31, h = lane >> 5;
  const BitTransposeLane btl = bit_transpose_lane(c);
  // XCD-aware tile order: XCD e owns the column tiles jt = e + 8 m and takes them one at a time, sweeping all row tiles
  // under each.  The workgroups resident on an XCD then share ONE V tile (128 KB) and differ in their U tiles (32 KB
  // each): per workgroup only its U tile is new to the L2.  (Cycling through the XCD's 16 column tiles first kept 2 MB
  // of V + 1 MB of W2 live beside the bit-image write stream and re-fetched V about twice per workgroup: 18.3 GB of
  // fetches per launch at B = 4096, profiles/r1_d_pmc_traffic.json.)
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

  // ---- DMA source addresses -------------------------------------------------------------------------------------
  // W: 32 rows per wave, 4 instructions of 8 rows: row = 32 wave + 8 q + (lane >> 3), chunk position lane & 7
  int woff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 32 * wave + 8 * q + (lane >> 3);
    woff[q] = row * H1 * 2 + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
  }
  // V: 8 instructions of 4 rows over the workgroup: instruction id = wave + NWAVES * q
  constexpr int NVI = 8 / NWAVES;  // V instructions per wave (1 or 2)
  const char* vsrc[NVI];
#pragma unroll
  for (int q = 0; q < NVI; ++q) {
    const int row = 4 * (wave + NWAVES * q) + (lane >> 4);
    int64_t gj = j0 + row;
    if (gj >= b) gj = b - 1;
    vsrc[q] = reinterpret_cast<const char*>(V + gj * H1) + (((lane & 15) ^ (row & 15)) << 4);
  }
  // U: 2 instructions of 4 rows; every wave issues the one of its parity (duplicates write the same bytes)
  const char* usrc;
  {
    int64_t li = i0 + 4 * (wave & 1) + (lane >> 4);
    if (li >= b_rows) li = b_rows - 1;
    usrc = reinterpret_cast<const char*>(U + li * H1) + ((lane & 15) << 4);
  }
  auto issue_tile = [&](int pass, int kt, int buf) {
    char* base = smem + buf * L::BUF_BYTES;
    const char* wsrc = reinterpret_cast<const char*>(W2 + ((int64_t)pass * NP) * H1 + kt * 64);
#pragma unroll
    for (int q = 0; q < 4; ++q) MI_GLDS16(wsrc + woff[q], base + (32 * wave + 8 * q) * 128);
#pragma unroll
    for (int q = 0; q < NVI; ++q) MI_GLDS16(vsrc[q] + kt * 256, base + L::W_BYTES + 4 * (wave + NWAVES * q) * 256);
    MI_GLDS16(usrc + kt * 256, base + L::W_BYTES + L::V_BYTES + 4 * (wave & 1) * 256);
  };

  // ---- fragment read offsets ---------------------------------------------------------------------------------------
  const int wrow_off = (128 * wn + c) * 128;  // + a * 32 * 128
  const int wswz = (c >> 1) & 7;
  const int vrow_off = L::W_BYTES + c * 256;
  const int vswz = c & 15;
  const int urow_off = L::W_BYTES + L::V_BYTES + (2 * wp) * 256;  // + t * 256

  if (tid < NP) {  // placeholder write so the first pass's loads below are ordered by the barrier
    b2s[tid] = 0.0f;
  }
  float s_total[2] = {0.0f, 0.0f};

  for (int pass = 0; pass < n_pass; ++pass) {
    __syncthreads();  // previous pass: every wave has left its epilogue (reads w3s) and its last tile's LDS reads
    if (tid < NP) {
      b2s[tid] = b2[pass * NP + tid];
      w3s[tid] = w3[pass * NP + tid];
    }
    issue_tile(pass, 0, 0);
    __syncthreads();  // hipcc waits vmcnt(0) before the barrier: tile 0 landed; b2s / w3s visible

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float bias = b2s[wn * 128 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
        acc[a][0][r] = bias;
        acc[a][1][r] = bias;
      }

    for (int kt = 0; kt < n_kt; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < n_kt) issue_tile(pass, kt + 1, buf ^ 1);
      const char* base = smem + buf * L::BUF_BYTES;
      // Software pipeline over the four 16-wide K steps of the tile: while the 8 MFMAs of step kk run, the fragments of
      // step kk+1 are already in flight (issued before the MFMAs, pinned there by a scheduling fence) and relu(U+V)
      // of step kk+1 is generated between the MFMAs (one MFMA : three VALU).
      struct Frag {
        bf16x8 wf[4];
        f32x4 v0, v1, u0[2], u1[2];
      };
      auto read_frag = [&](int kk, Frag& f) {
        const int wpos = ((2 * kk + h) ^ wswz) << 4;
#pragma unroll
        for (int a = 0; a < 4; ++a) f.wf[a] = *reinterpret_cast<const bf16x8*>(base + wrow_off + a * 32 * 128 + wpos);
        const int q0 = 4 * kk + 2 * h;
        f.v0 = *reinterpret_cast<const f32x4*>(base + vrow_off + ((q0 ^ vswz) << 4));
        f.v1 = *reinterpret_cast<const f32x4*>(base + vrow_off + (((q0 + 1) ^ vswz) << 4));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f.u0[t] = *reinterpret_cast<const f32x4*>(base + urow_off + t * 256 + q0 * 16);
          f.u1[t] = *reinterpret_cast<const f32x4*>(base + urow_off + t * 256 + q0 * 16 + 16);
        }
      };
      Frag fa, fb;
      bf16x8 hfa[2], hfb[2];
      read_frag(0, fa);
#pragma unroll
      for (int t = 0; t < 2; ++t) hfa[t] = gen_h1_bf16(fa.u0[t], fa.u1[t], fa.v0, fa.v1);
#define MI_FWD_STEP(KK, CUR, HCUR, NXT, HNXT)                                                              \
  {                                                                                                       \
    if ((KK) < 3) read_frag((KK) + 1, NXT);                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) _Pragma("unroll") for (int t = 0; t < 2; ++t)           \
        acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(CUR.wf[a], HCUR[t], acc[a][t], 0, 0, 0);      \
    if ((KK) < 3) {                                                                                       \
      _Pragma("unroll") for (int t = 0; t < 2; ++t) HNXT[t] = gen_h1_bf16(NXT.u0[t], NXT.u1[t], NXT.v0, NXT.v1); \
      _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                \
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                                \
      }                                                                                                   \
    }                                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
  }
      MI_FWD_STEP(0, fa, hfa, fb, hfb)
      MI_FWD_STEP(1, fb, hfb, fa, hfa)
      MI_FWD_STEP(2, fa, hfa, fb, hfb)
      MI_FWD_STEP(3, fb, hfb, fa, hfa)
#undef MI_FWD_STEP
      __syncthreads();  // vmcnt(0) + barrier: tile kt+1 landed, buffer `buf` free for tile kt+2
    }

    // ---- epilogue of the pass: relu, dot with w3, sign bits (fwd_epilogue_row, mi_concat_fwd.h) ------------------------
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int64_t li = i0 + 2 * wp + t;
      const int64_t gj = j0 + c;
      const int pw = pass * NWN + wn;
      const bool row_ok = li < b_rows, col_ok = gj < b;
      const int64_t lic = row_ok ? li : 0, gjc = col_ok ? gj : 0;
      s_total[t] += fwd_epilogue_row(acc[0][t], acc[1][t], acc[2][t], acc[3][t], w3s + wn * 128, h, c, bitsP != nullptr,
                                     row_ok, col_ok,
                                     bitsP + bitsp_index(lic, gjc, h, pw, (b + 31) / 32, (int)(H2 / 128)),
                                     bitsN + (lic * JB + jt) * H2 + pw * 128, btl);
    }
  }

#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float s = s_total[t] + __shfl_xor(s_total[t], 32);
    if (h == 0) sred[wn * 256 + (2 * wp + t) * 32 + c] = s;
  }
  __syncthreads();
  if (tid < kFwdTI * kFwdTJ) {
    const int il = tid >> 5, jl = tid & 31;
    const int64_t li = i0 + il, gj = j0 + jl;
    if (li < b_rows && gj < b) {
      float s = sred[tid];
      if (NWN == 2) s += sred[256 + tid];
      S[li * b + gj] = s + b3[0];
    }
  }
}

}  // namespace mi
