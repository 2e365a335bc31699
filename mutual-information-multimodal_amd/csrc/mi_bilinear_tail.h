// The backward's tail of the fused bilinear critic in TWO launches (round 2: slab reduce, dW | dX, dW slab reduce, and
// finalize in front: four launches, 35 us around a 70 us kernel).
//
//   reference call site: mutual_info_img_txt/main_utils.py:220-226 (loss.backward() of the critic step); the bilinear
//   scorer is an extension (BASELINE.json's headline critic), bound / masking semantics are the reference's.
//
// (1) flash_tail_kernel: one workgroup per 32-row block of an output.
//       * adds the fused kernel's partial sums (fp16 slabs under per-wave scales, or fp32) in split order with
//         exp(m_ref - lse), subtracts the diagonal term: rows of dT (job 0) / of grad_y (job 1);
//       * MERGE: merges the per-wave records of problem 0 itself (the order of finalize_kernel: thread k takes records
//         k, k + 256, ..., then the fixed block reduction) -- every workgroup gets the same bits, workgroup 0 writes
//         the statistics and the loss.  Used by the one-call step (mi_bilinear_step), where nothing needs the loss
//         between the fused kernel and here: no finalize launch;
//       * job 0 additionally: dX = dT W^T on the matrix cores (the dT tile is in LDS anyway; W arrives as fragment-major
//         1 KB blocks, coalesced), and dT^T written fragment-major for (2).  No dT round trip through HBM for dX, no
//         second orientation of dT, no separate launch.
// (2) bilinear_dw_kernel: dW = X^T dT, one 32 x 32 tile per workgroup (256 of them at d = 512), the batch dimension
//     split over the four WAVES of the workgroup and reduced through LDS in a fixed order: no split-K slabs in HBM, no
//     reduce launch, no float atomics.  Both operands are read as fragment-major 1 KB blocks.
#pragma once
#include "mi_bilinear_flash.h"

namespace mi {

struct FlashTailArgs {
  FlashReduceJob j[2];       // out_bf / out_bf_t unused here
  const mi_stats* stats;     // !MERGE: the global statistics (after the cross-rank merge when sharded)
  const float* grad_out;     // dL/dloss or null (1)
  // MERGE
  const Partial* merge_rec;
  int64_t n_merge, n_pos;
  int estimator;
  float* loss_out;
  mi_stats* stats_out;
  float* partials_out;       // optional local record (8 floats)
  // dX = dT W^T (job 0)
  const bf16_t* w_frag;      // W [dx][D] fragment-major: block (a / 32, c / 16)
  int64_t dx;
  float* grad_x;             // [m0][dx]
  bf16_t* grad_x_bf;         // [m0][dx] bf16 instead (the bf16 boundary: mi_bilinear_step_bf16) or null
  bf16_t* dtt_frag;          // dT^T [D][m0] fragment-major: block (c / 32, i / 16); operand of bilinear_dw_kernel
  int n_blocks0;             // 32-row blocks of job 0 (they come first in the grid: the longer ones)
  // the same product for job 1 (separable critic: dY = dC Wh^T and dC^T for dWh; null for the bilinear critic, whose job 1
  // writes grad_y rows directly)
  const bf16_t* w_frag1;
  int64_t dx1;
  float* grad_x1;
  bf16_t* dtt_frag1;
};

constexpr int kTailPad = 4;  // floats of padding per LDS row: 16-byte row reads conflict-free, column reads too
constexpr int kTailThreads = 512;  // eight waves: the grid is one workgroup per CU, so the loads a CU keeps in flight are
                                   // this workgroup's; every slab load of the block is issued before the first use

// GBF: the gradients for the embeddings go out as bf16 (the bf16 boundary) -- a compile-time variant: as a run-time branch the
// extra store paths pushed the fp32 kernel from 246 registers to 256 + 172 bytes of scratch (17 -> 26 us).
template <int D, bool F16, bool MERGE, bool GBF = false>
__global__ __launch_bounds__(kTailThreads, 2) void flash_tail_kernel(FlashTailArgs args) {
  kernarg_prefetch<(int)sizeof(FlashTailArgs)>();
  extern __shared__ __attribute__((aligned(16))) char smem_tail[];
  float(*tile)[D + kTailPad] = reinterpret_cast<float(*)[D + kTailPad]>(smem_tail);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: see bilinear_dw_kernel)
  const int blk = (int)blockIdx.x;
  const int job = blk >= args.n_blocks0 ? 1 : 0;
  const FlashReduceJob& J = args.j[job];
  const int64_t wb = job ? blk - args.n_blocks0 : blk;  // 32-row block of this job
  if (wb * 32 >= J.m) return;
  const int rb = (int)(wb >> 2), w = (int)(wb & 3);
  const int64_t i0 = wb * 32;
  const f16_t* slab16 = reinterpret_cast<const f16_t*>(J.slab);
  const float* unscale = reinterpret_cast<const float*>(slab16 + (int64_t)J.n_split * J.n_rb * 4 * (32 * D));
  constexpr int NC = D / 128;  // 128-column chunks: 4096 elements of a wave's slab each

  // ---- every slab load of the first four splits goes out before anything else (the record merge below runs under
  // their latency).  Element index inside a chunk: thread t holds 8 (fp16) / 2 x 4 (fp32) consecutive elements.
  int64_t wvs[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) wvs[q] = ((int64_t)(q < J.n_split ? q : J.n_split - 1) * J.n_rb + rb) * 4 + w;
  f16x8 vh[NC][4];
  f32x4 vf[NC][4][2];
#pragma unroll
  for (int cc = 0; cc < NC; ++cc)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if constexpr (F16) {
        vh[cc][q] = *reinterpret_cast<const f16x8*>(slab16 + wvs[q] * (32 * D) + (int64_t)cc * 4096 + tid * 8);
      } else {
#pragma unroll
        for (int u = 0; u < 2; ++u)
          vf[cc][q][u] = *reinterpret_cast<const f32x4*>(J.slab + wvs[q] * (32 * D) + (int64_t)cc * 4096 + (u * 512 + tid) * 4);
      }
    }

  // ... and so does everything else this workgroup reads that does not depend on the statistics: the splits' reference
  // points and scales, and the rows subtracted on the diagonal.  (A block moves ~200 KB; what it costs is the chain of
  // DEPENDENT round trips to memory, each 1 - 2 us under load.)
  float m_rec[4], u_rec[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    m_rec[q] = J.rec[wvs[q]].m;
    u_rec[q] = F16 ? unscale[wvs[q]] : 1.0f;
  }
  const int orow = tid >> 4;  // row-major pass below: 16 threads per row, each D / 128 groups of 8 columns
  const int64_t ojo = i0 + orow + J.diag;
  const bool has_other = ojo >= 0 && ojo < J.n_other;
  bf16x8 oth[D / 128];
#pragma unroll
  for (int g = 0; g < D / 128; ++g) {
    oth[g] = bf16x8{};
    if (has_other) oth[g] = *reinterpret_cast<const bf16x8*>(J.other + ojo * D + ((tid & 15) + 16 * g) * 8);
  }

  // ---- statistics
  float lse, n_pos_f;
  if constexpr (MERGE) {
    // (carved from the dynamic region behind the tile: a static __shared__ object in front of it would shift its base
    // off the 16-byte alignment the tile's row reads need)
    char* extra = smem_tail + (size_t)32 * (D + kTailPad) * sizeof(float);
    Partial* scratch = reinterpret_cast<Partial*>(extra);
    unsigned long long* cnt_scratch = reinterpret_cast<unsigned long long*>(extra + 64);
    float& lse_sh = *reinterpret_cast<float*>(extra + 96);
    // the merge order of finalize_kernel (256 threads: thread k takes records k, k + 256, ...; wave reductions; waves in
    // order), so that this path and the two-call path produce the same bits: waves 0 - 3 do it, the others pass by
    if (wave < 4) {
      Partial p{MI_NEG_INF, 0.0f, 0.0f, 0u};
      unsigned long long cnt = 0;
      for (int64_t k = tid; k < args.n_merge; k += 256) {
        const Partial q = args.merge_rec[k];
        lse_merge(p.m, p.s, q.m, q.s);
        p.pos += q.pos;
        cnt += q.cnt;
      }
      for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
      wave_lse(p.m, p.s);
      p.pos = wave_sum(p.pos);
      if (lane == 0) {
        cnt_scratch[wave] = cnt;
        p.cnt = 0;
        scratch[wave] = p;
      }
    }
    __syncthreads();
    if (tid == 0) {
      Partial p = scratch[0];
      for (int k = 1; k < 4; ++k) {
        lse_merge(p.m, p.s, scratch[k].m, scratch[k].s);
        p.pos += scratch[k].pos;
      }
      const unsigned long long total = cnt_scratch[0] + cnt_scratch[1] + cnt_scratch[2] + cnt_scratch[3];
      const float l = (p.s > 0.0f) ? p.m + logf(p.s) : MI_NEG_INF;  // logsumexp(empty) = -inf
      lse_sh = l;
      if (blk == 0) {  // one workgroup publishes (the others computed the same bits)
        if (args.partials_out) {
          float* r = args.partials_out;
          r[0] = p.m; r[1] = p.s; r[2] = p.pos;
          r[3] = (float)(total & 0xFFFFFFull); r[4] = (float)(total >> 24);
          r[5] = r[6] = r[7] = 0.0f;
        }
        mi_stats* st = args.stats_out;
        const float pos_mean = p.pos / (float)args.n_pos;
        const float log_n = logf((float)total);  // float32 constant as in mi_critics.py:10
        st->lse = l; st->pos_mean = pos_mean; st->log_n_neg = log_n; st->neg_max = p.m;
        st->loss_dv = (l - log_n) - pos_mean;
        st->loss_infonce = l - pos_mean;
        st->reserved0 = st->reserved1 = 0.0f;
        st->n_neg = (int64_t)total; st->n_pos = args.n_pos; st->reserved2 = st->reserved3 = 0;
        if (args.loss_out) args.loss_out[0] = (args.estimator == MI_DV) ? st->loss_dv : st->loss_infonce;
      }
    }
    __syncthreads();
    lse = lse_sh;
    n_pos_f = (float)args.n_pos;
  } else {
    lse = args.stats->lse;
    n_pos_f = (float)args.stats->n_pos;
  }
  const float go = args.grad_out ? args.grad_out[0] : 1.0f;
  const float gpos = go / n_pos_f;

  // ---- add the splits (split order) into the LDS tile [32][D]
  float cs[4];
  for (int s0 = 0; s0 < J.n_split; s0 += 4) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int s = s0 + q < J.n_split ? s0 + q : J.n_split - 1;
      wvs[q] = ((int64_t)s * J.n_rb + rb) * 4 + w;
      if (s0 > 0) {
        m_rec[q] = J.rec[wvs[q]].m;
        u_rec[q] = F16 ? unscale[wvs[q]] : 1.0f;
      }
      cs[q] = (s0 + q < J.n_split && m_rec[q] > MI_NEG_INF) ? go * __expf(m_rec[q] - lse) * u_rec[q] : 0.0f;
    }
    if (s0 > 0) {  // (more than four splits: the next round's loads)
#pragma unroll
      for (int cc = 0; cc < NC; ++cc)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if constexpr (F16) {
            vh[cc][q] = *reinterpret_cast<const f16x8*>(slab16 + wvs[q] * (32 * D) + (int64_t)cc * 4096 + tid * 8);
          } else {
#pragma unroll
            for (int u = 0; u < 2; ++u)
              vf[cc][q][u] = *reinterpret_cast<const f32x4*>(J.slab + wvs[q] * (32 * D) + (int64_t)cc * 4096 + (u * 512 + tid) * 4);
          }
        }
    }
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      if constexpr (F16) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (cs[q] != 0.0f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)vh[cc][q][e] * cs[q];
          }
        // 16-byte chunk h = tid = ct_l * 128 + gp * 64 + lane holds accumulator registers 8 gp .. 8 gp + 7 of the lane:
        // rows 8 g + 4 (lane >> 5) + e for g = 2 gp and 2 gp + 1, column ct_l * 32 + (lane & 31)
        const int ct_l = tid >> 7, gp = (tid >> 6) & 1, ln = tid & 63;
        const int col = cc * 128 + ct_l * 32 + (ln & 31);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int row0 = 8 * (2 * gp + hh) + 4 * (ln >> 5);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (s0 == 0) tile[row0 + e][col] = acc[4 * hh + e];
            else tile[row0 + e][col] += acc[4 * hh + e];
          }
        }
      } else {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (cs[q] != 0.0f) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[4 * u + e] += vf[cc][q][u][e] * cs[q];
          }
        // float4 index f = u * 512 + tid = ct_l * 256 + g * 64 + lane
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int f = u * 512 + tid;
          const int ct_l = f >> 8, g = (f >> 6) & 3, ln = f & 63;
          const int col = cc * 128 + ct_l * 32 + (ln & 31), row0 = 8 * g + 4 * (ln >> 5);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (s0 == 0) tile[row0 + e][col] = acc[4 * u + e];
            else tile[row0 + e][col] += acc[4 * u + e];
          }
        }
      }
    }
  }
  __syncthreads();

  // ---- row-major pass: subtract the diagonal term; grad_y (job 1) goes out here
  {
    const int row = orow;
    const int64_t i = i0 + row;
#pragma unroll
    for (int g = 0; g < D / 128; ++g) {
      const int c = ((tid & 15) + 16 * g) * 8;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[row][c + e];
      if (has_other) {
        const bf16x8 o = oth[g];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] -= gpos * (float)o[e];
        if (job == 0 || args.w_frag1 != nullptr) {  // the tile feeds the product below
#pragma unroll
          for (int e = 0; e < 8; ++e) tile[row][c + e] = v[e];
        }
      }
      if constexpr (GBF) {  // the bf16 boundary: the gradient in the dtype of the embeddings (one rounding of the fp32 sum)
        if (J.out_bf) {
          const bf16x8 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3],
                            (bf16_t)v[4], (bf16_t)v[5], (bf16_t)v[6], (bf16_t)v[7]};
          *reinterpret_cast<bf16x8*>(J.out_bf + i * D + c) = o;
        }
      } else if (J.out_f32) {
        *reinterpret_cast<f32x4*>(J.out_f32 + i * D + c) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(J.out_f32 + i * D + c + 4) = f32x4{v[4], v[5], v[6], v[7]};
      }
    }
  }
  if (job != 0) {
    if (GBF || args.w_frag1 == nullptr) return;
  } else if (GBF ? args.grad_x_bf == nullptr : args.grad_x == nullptr) {
    return;
  }
  // (scalar selects: job is workgroup-uniform)
  const bf16_t* const w_frag = job ? args.w_frag1 : args.w_frag;
  const int64_t dx_out = job ? args.dx1 : args.dx;
  float* const gx_out = job ? args.grad_x1 : args.grad_x;
  bf16_t* const dtt_out = job ? args.dtt_frag1 : args.dtt_frag;
  __syncthreads();

  // ---- dX[i0 .. +32][a] = sum_c dT[i][c] W[a][c]: A = the dT tile (rows on the lane; bf16 from LDS), B = W fragments
  // straight from the fragment-major copy (1 KB per (32 rows of W, 16 of c), coalesced).  Wave w: two 32-column tiles.
  // The W loads go out first: dT^T below is written under their latency.
  constexpr int NK = D / 16;
  const int r = lane & 31, h = lane >> 5;
  const int64_t n_at = dx_out / 32;  // 32-column tiles of dX
  constexpr int PF = 8;               // k-steps of W fragments in flight
  bf16x8 bq[PF][2];
  const bf16_t* wbase[2];
  {
    const int64_t at0 = wave * 2;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int64_t at = at0 + t < n_at ? at0 + t : n_at - 1;
      wbase[t] = w_frag + (at * NK * 64 + lane) * 8;
    }
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
      for (int t = 0; t < 2; ++t) bq[p][t] = *reinterpret_cast<const bf16x8*>(wbase[t] + (int64_t)p * 512);
  }

  // ---- dT^T, fragment-major: block (cb = c / 32, kb = i / 16) is 1 KB in MFMA operand lane order: lane (r = c % 32, h)
  // holds dT[16 kb + 8 h + j][32 cb + r], j = 0..7.  This workgroup owns i in [i0, i0 + 32): kb = i0 / 16 + {0, 1}.
  {
    const int64_t nkb = J.m / 16;
#pragma unroll
    for (int q = 0; q < (D / 32) * 2 / 8; ++q) {
      const int bi = wave + 8 * q;  // block index 0 .. 2 D / 32: (cb, kl)
      const int cb = bi >> 1, kl = bi & 1;
      bf16x8 o;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) o[jj] = (bf16_t)tile[16 * kl + 8 * h + jj][32 * cb + r];
      *reinterpret_cast<bf16x8*>(dtt_out + (((int64_t)cb * nkb + (i0 / 16 + kl)) * 64 + lane) * 8) = o;
    }
  }

  for (int64_t at0 = wave * 2; at0 < n_at; at0 += 16) {
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
    if (at0 != wave * 2) {  // (more than sixteen column tiles: this round's first fragments)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int64_t at = at0 + t < n_at ? at0 + t : n_at - 1;
        wbase[t] = w_frag + (at * NK * 64 + lane) * 8;
      }
#pragma unroll
      for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int t = 0; t < 2; ++t) bq[p][t] = *reinterpret_cast<const bf16x8*>(wbase[t] + (int64_t)p * 512);
    }
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(&tile[r][16 * kk + 8 * h]);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(&tile[r][16 * kk + 8 * h + 4]);
      const bf16x8 af = {(bf16_t)a0[0], (bf16_t)a0[1], (bf16_t)a0[2], (bf16_t)a0[3],
                         (bf16_t)a1[0], (bf16_t)a1[1], (bf16_t)a1[2], (bf16_t)a1[3]};
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bq[kk % PF][t], acc[t], 0, 0, 0);
      if (kk + PF < NK) {
#pragma unroll
        for (int t = 0; t < 2; ++t) bq[kk % PF][t] = *reinterpret_cast<const bf16x8*>(wbase[t] + (int64_t)(kk + PF) * 512);
      }
    }
    // C layout: column (lane & 31) = a within the tile, rows 8 g + 4 h + e
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (at0 + t >= n_at) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t o = (i0 + 8 * g + 4 * h + e) * dx_out + (at0 + t) * 32 + r;
          if constexpr (GBF) args.grad_x_bf[o] = (bf16_t)acc[t][4 * g + e];
          else gx_out[o] = acc[t][4 * g + e];
        }
    }
  }
}

template <int D, bool F16, bool MERGE, bool GBF = false>
static inline int launch_flash_tail_t(const FlashTailArgs& a, unsigned grid, hipStream_t st, const char* what) {
  const size_t smem = (size_t)32 * (D + kTailPad) * sizeof(float) + 128;
  MI_SET_DYN_SMEM((flash_tail_kernel<D, F16, MERGE, GBF>), smem, "hipFuncSetAttribute(flash_tail_kernel)");
  {
    ProfScope prof_(what, st);
    hipLaunchKernelGGL((flash_tail_kernel<D, F16, MERGE, GBF>), dim3(grid), dim3(kTailThreads), smem, st, a);
  }
  MI_LAUNCH_CHECK(what);
  return MI_OK;
}

// job 0 blocks first; a.n_blocks0 is set here
static inline int launch_flash_tail(FlashTailArgs a, int64_t d, bool slab_f16, bool merge, hipStream_t st, const char* what) {
  a.n_blocks0 = (int)(a.j[0].m / 32);
  const unsigned grid = (unsigned)(a.n_blocks0 + a.j[1].m / 32);
#define MI_TAIL_CASE(DD)                                                                          \
  if (d == DD) {                                                                                  \
    if (a.grad_x_bf) {  /* the bf16 boundary: fp16 slabs, merged statistics (mi_bilinear_step_bf16) */ \
      if (!(slab_f16 && merge)) {                                                                 \
        set_error("launch_flash_tail: bf16 gradients need fp16 slabs and the merged form");       \
        return MI_EINVAL;                                                                         \
      }                                                                                           \
      return launch_flash_tail_t<DD, true, true, true>(a, grid, st, what);                        \
    }                                                                                             \
    if (slab_f16) return merge ? launch_flash_tail_t<DD, true, true>(a, grid, st, what)           \
                               : launch_flash_tail_t<DD, true, false>(a, grid, st, what);         \
    return merge ? launch_flash_tail_t<DD, false, true>(a, grid, st, what)                        \
                 : launch_flash_tail_t<DD, false, false>(a, grid, st, what);                      \
  }
  MI_TAIL_CASE(512)
  MI_TAIL_CASE(256)
  MI_TAIL_CASE(128)
#undef MI_TAIL_CASE
  set_error("launch_flash_tail: unsupported width %lld", (long long)d);
  return MI_ESHAPE;
}

// ------------------------------------------------------------------------------------------------ dW = X^T dT
// out[a][c] = sum_i X[i][a] dT[i][c].  A = X^T (rows a, K = i), B = dT^T (rows c, K = i), both fragment-major:
//   xt_frag  block (a / 32, i / 16), dtt_frag block (c / 32, i / 16), 1 KB each in MFMA operand lane order.
// One workgroup per 32 x 32 tile of dW; wave w sums i in [w K / 4, (w + 1) K / 4); the four partial tiles are added
// through LDS in wave order.
struct DwArgs {
  const bf16_t* xt_frag;
  const bf16_t* dtt_frag;
  int64_t dx, dy, k;  // k = rows of the batch block (multiple of 64)
  float* out;         // [dx][dy]
};

constexpr int kDwWaves = 8;  // the batch dimension is split over the waves of a workgroup (one workgroup per CU: the loads
                             // a CU keeps in flight are this workgroup's)
struct DwArgs2 {
  DwArgs p[2];
  int n_tiles0;  // workgroups of problem 0 (they come first)
};
static __global__ __launch_bounds__(64 * kDwWaves, 1) void bilinear_dw_kernel(DwArgs2 a2) {
  __shared__ float part[kDwWaves - 1][32][33];
  const int prob = (int)blockIdx.x >= a2.n_tiles0 ? 1 : 0;
  const DwArgs& a = a2.p[prob];
  const int64_t tile_id = prob ? (int64_t)blockIdx.x - a2.n_tiles0 : (int64_t)blockIdx.x;
  // (readfirstlane: the wave index must be a SCALAR for hipcc -- an MFMA ignores EXEC, so a matrix instruction under a
  // condition hipcc takes for divergent, and lowers to an EXEC mask without a skip branch, would still execute)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t nct = a.dy / 32;
  const int64_t at = tile_id / nct, ct = tile_id % nct;
  const int64_t nkb = a.k / 16;                 // 16-deep blocks over the batch
  const int64_t kb0 = nkb * wave / kDwWaves, kb1 = nkb * (wave + 1) / kDwWaves;
  const bf16_t* ap = a.xt_frag + ((at * nkb + kb0) * 64 + lane) * 8;
  const bf16_t* bp = a.dtt_frag + ((ct * nkb + kb0) * 64 + lane) * 8;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
  constexpr int PF = 16;
  bf16x8 af[PF], bfr[PF];
  const int64_t n = kb1 - kb0;
#pragma unroll
  for (int p = 0; p < PF; ++p) {
    const int64_t q = p < n ? p : n - 1;
    af[p] = *reinterpret_cast<const bf16x8*>(ap + q * 512);
    bfr[p] = *reinterpret_cast<const bf16x8*>(bp + q * 512);
  }
  for (int64_t q0 = 0; q0 < n; q0 += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      if (q0 + p < n) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[p], bfr[p], acc, 0, 0, 0);
      const int64_t q = q0 + p + PF;
      if (q < n) {
        af[p] = *reinterpret_cast<const bf16x8*>(ap + q * 512);
        bfr[p] = *reinterpret_cast<const bf16x8*>(bp + q * 512);
      }
    }
  }
  // acc: column (lane & 31) = c, rows 8 g + 4 h + e = a
  const int r = lane & 31, h = lane >> 5;
  if (wave > 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) part[wave - 1][8 * g + 4 * h + e][r] = acc[4 * g + e];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = 8 * g + 4 * h + e;
        float v = acc[4 * g + e];
#pragma unroll
        for (int k = 0; k < kDwWaves - 1; ++k) v += part[k][row][r];  // wave order: bit-reproducible
        a.out[(at * 32 + row) * a.dy + ct * 32 + r] = v;
      }
  }
}

// one launch for up to two products (the separable critic's dWg and dWh); b == null: one
static inline int launch_bilinear_dw(const DwArgs& a, hipStream_t st, const char* what, const DwArgs* b = nullptr) {
  DwArgs2 a2{};
  a2.p[0] = a;
  a2.p[1] = b ? *b : a;
  a2.n_tiles0 = (int)((a.dx / 32) * (a.dy / 32));
  const unsigned grid = (unsigned)(a2.n_tiles0 + (b ? (b->dx / 32) * (b->dy / 32) : 0));
  {
    ProfScope prof_(what, st);
    hipLaunchKernelGGL(bilinear_dw_kernel, dim3(grid), dim3(64 * kDwWaves), 0, st, a2);
  }
  MI_LAUNCH_CHECK(what);
  return MI_OK;
}

// shapes the two kernels take
static inline bool flash_tail_ok(int64_t br, int64_t dx, int64_t dy) {
  return br % 128 == 0 && dx % 32 == 0 && dy % 32 == 0 && flash_width_ok(dy);
}

}  // namespace mi
