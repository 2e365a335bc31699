// Fused B x B stage of the bilinear critic (flash-attention shaped): scores, masked log-sum-exp AND the two B x B
// gradient contractions in ONE launch, without ever writing S, G or G^T to HBM.
//
//   reference call site: mutual_info_img_txt/main_utils.py:220-226 (pairs -> critic -> bound -> backward); bound and
//   masking semantics: mi_critics.py:3-23, main_utils.py:99-108.  The bilinear scorer itself is an extension.
//
// The loss has ONE global log-sum-exp (not one per row), so  dL/dS = go * (mask * exp(S - lse) - I / B)  factors as
//   dT = go * (exp(m - lse) * U - Yb / B),   U = sum_j mask_ij exp(S_ij - m) Y_j      (m: any reference point)
//   dY = go * (exp(m - lse) * V - Tb / B),   V = sum_i mask_ij exp(S_ij - m) T_i
// and U, V can be accumulated in the same pass that produces the log-sum-exp: no recompute pass after the forward, no
// materialised gradient matrix (the round-1 design wrote G and G^T, 2 x 33 MB, and read them back).
//
// Two "problems" share the kernel (one output-stationary sweep each; S tiles are recomputed per problem, which is
// cheaper than a cross-workgroup reduction of a [B, d] fp32 output per tile):
//   problem 0: stationary rows = T (local row block), streamed rows = Y (all columns)  -> U, LSE partials, positives
//   problem 1: stationary rows = Y (all columns),     streamed rows = T (local rows)   -> V
// A workgroup = 4 waves (one per SIMD, up to 512 registers each) owns 128 stationary rows, 32 per wave.  Per wave:
//   * the 32 stationary rows live in registers as MFMA B fragments for the whole sweep (D / 16 fragments);
//   * per streamed tile of 32 rows (LDS, filled by LDS-DMA, 4 stages): X = K Q^T (32 x 32 fp32, streamed row in the
//     registers, stationary row on the lane) by D / 16 chained v_mfma_f32_32x32x16_bf16;
//   * P = mask * exp(X - m_ref) in registers with a wave-uniform reference m_ref (raised, with a rescale of the
//     accumulators, only when a tile maximum exceeds it by kFlThr); packed to bf16, the accumulator layout of X IS the A
//     operand of the next product (guide: "an accumulator tile as the next MFMA's operand"), no LDS round trip;
//   * O[32 x D] += P^T-as-A x V-tile, the tile read a second time from the SAME LDS image with ds_read_b64_tr_b16
//     (image (b) of guide T10: chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)) serves row reads and transposed reads
//     conflict-free).
// The streamed range is split over `n_split` workgroups per row block so that ~256 workgroups fill the chip; every
// (problem, split) pair is pinned to one XCD (L % n_combo), whose 4 MB L2 then holds the streamed rows all its
// workgroups sweep.  Partial O tiles go to slabs in accumulator order (16-byte stores, 1 KB per wave store) with the
// wave's (m_ref, sum, positives, count) record; `flash_reduce_kernel` adds the slabs in a fixed order with the global
// lse -- bit-reproducible, no float atomics.  The slabs are the step's largest HBM traffic (fp32: 64 MB written and read
// back at B = 4096, d = 512), so by default a wave stores its tile as fp16 under one power-of-two scale of its own
// (largest element in [2^13, 2^14); the factor back sits behind the slabs): 11 significant bits on partial sums whose
// addends were rounded to 8 (P and the operands are bf16), half the traffic.  MI_FLASH_SLAB_F32=1 keeps fp32 slabs.
#pragma once
#include <type_traits>

#include "mi_common.h"

namespace mi {

// diagnostic build only (make STAMPS=1): 16 s_memtime slots per workgroup, written by wave 0 (tools/diag/stamps_flash.py)
#ifdef MI_STAMPS
#define MI_FL_STAMP(slot)                                                                       \
  do {                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    if (threadIdx.x == 0 && g_stamps) g_stamps[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)
#else
#define MI_FL_STAMP(slot) do {} while (0)
#endif

// Timing experiments with WRONG results (diagnostic builds only: make STAMPS=1 DIAG=<bits>):
// bit 0 no LDS-DMA in the loop, bit 1 no softmax, bit 2 no workgroup barrier in the loop, bit 3 no fragment reads,
// bit 4 no counted LDS waits; bits 5..9: no exponentials / no sum additions / no bf16 packing / no softmax head up
// to the reference-point decision / no prescaling.  Compile-time: a run-time test per MFMA gap costs more than what it
// switches off (a scalar branch beside the MFMAs is several issue slots).
#ifndef MI_FL_DIAG
#define MI_FL_DIAG 0
#endif
constexpr int kFlDiag = MI_FL_DIAG;
constexpr int kFlRows = 128;     // stationary rows per workgroup
constexpr int kFlBN = 32;        // streamed rows per tile
constexpr int kFlStages = 4;     // LDS stages (three tiles in flight)
// raise the reference point when a tile maximum exceeds it by this much.  exp(60) = 1e26: the exponentials are bf16 (fp32's
// exponent range), the sums fp32 (<= 1e26 * 4096 * |operand|), the slabs carry a per-wave power-of-two scale.  Every raise
// costs a drain of the matrix pipe and ~260 multiplications; at 24 a batch with score deviations of ~20 (the benchmark's)
// raised several times per workgroup.
constexpr float kFlThr = 60.0f;
constexpr int kFlMaxTilesPerSplit = 64;  // streamed study ids of a split sit in LDS: 64 tiles x 32 x 8 bytes = 16 KB

struct FlashProblem {
  const bf16_t* q;        // stationary operand [m][D] in FRAGMENT-MAJOR order (EpiOut::bf_frag, mi_gemm_bf16.h)
  const bf16_t* kv;       // streamed operand   [n][D]
  const int64_t* sid_q;   // [m]
  const int64_t* sid_kv;  // [n]
  int64_t m, n;
  int64_t diag;           // streamed index of the positive of stationary row i:  j == i + diag
  int n_rb;               // row blocks of 128 stationary rows
  int n_split;            // workgroups per row block
  int tiles_per_split;
  const unsigned char* dup;  // [m / 32][n / 32]: 1 iff the 32 x 32 block holds a pair with equal study ids
  float* slab;            // [n_split][n_rb][4 waves][32 * D] partial sums in accumulator order: fp32, or (slab_f16) fp16
                          // followed by one fp32 factor per wave [n_split][n_rb][4] that undoes the wave's scale
  Partial* rec;           // [n_split][n_rb][4]
};
struct FlashArgs {
  FlashProblem p[2];
  int n_problems;
  int n_combo;  // (problem, split) pairs, padded to a multiple of 8
  int xcd_rows; // 1: (problem, row block) units pinned to XCDs; 0: (problem, split) pairs pinned to XCDs
  int diag;     // unused (the timing experiments are compile-time switches: MI_FL_DIAG)
  int slab_f16; // 1: scaled fp16 slabs (see FlashProblem::slab)
  int no_index_mask;  // A/B switch MI_FLASH_NO_INDEX_MASK: diagonal-only tiles take the exact id compares as well
};

template <int D>
struct FlashCfg {
  static constexpr int NK = D / 16;                    // 16-deep MFMA steps over the embedding width
  static constexpr int NT = D / 32;                    // 32-wide output column tiles
  static constexpr int RB = D * 2;                     // bytes per LDS row
  static constexpr int CPR = RB / 16;                  // 16-byte chunks per row
  static constexpr int RPP = 1024 / RB;                // rows per LDS-DMA piece
  static constexpr int STAGE = kFlBN * RB;             // bytes per stage
  static constexpr int PIECES = STAGE / 1024 / 4;      // LDS-DMA pieces per wave and tile
  static constexpr int SID_OFF = kFlStages * STAGE;    // streamed study ids
  static constexpr size_t SMEM = (size_t)SID_OFF + (size_t)kFlMaxTilesPerSplit * kFlBN * 8;
  static constexpr int QA = NT >= 16 ? NK / 2 : 0;     // stationary fragments [0, QA) in accumulator registers
  static constexpr int OA = NT >= 16 ? 12 : NT;        // output tiles [0, OA) in accumulator registers
  static_assert(D == 128 || D == 256 || D == 512, "unsupported embedding width");
};

// swizzle of image (b), guide T10: XOR on the low four bits of the 16-byte chunk index
__host__ __device__ constexpr int fl_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// wave-wide maximum as a wave-uniform value (DPP butterfly inside rows of 16, row broadcasts, lane 63)
__device__ __forceinline__ float wave_max_uniform(float v) {
  int x;
#define MI_DPP_MAX(ctrl, rmask)                                                        \
  x = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, rmask, 0xF, false); \
  v = fmaxf(v, __int_as_float(x));
  MI_DPP_MAX(0xB1, 0xF)   // quad_perm [1,0,3,2]
  MI_DPP_MAX(0x4E, 0xF)   // quad_perm [2,3,0,1]
  MI_DPP_MAX(0x141, 0xF)  // row_half_mirror
  MI_DPP_MAX(0x140, 0xF)  // row_mirror: every lane of a row holds the row maximum
  MI_DPP_MAX(0x142, 0xA)  // row_bcast:15 into rows 1 and 3
  MI_DPP_MAX(0x143, 0xC)  // row_bcast:31 into rows 2 and 3
#undef MI_DPP_MAX
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// one LDS-DMA piece: 64 lanes x 16 bytes from sbase + voff (per lane) to the LDS byte address lds_addr + 16 * lane.
// M0 carries the LDS address and is compiler-reserved: saved and restored inside the statement (guide 5.7).
__device__ __forceinline__ void fl_dma16(unsigned voff, const void* sbase, unsigned lds_addr) {
  unsigned keep;
  // the scalar operands are wave-uniform by construction; readfirstlane makes that provable to hipcc (guide T20).  A
  // VMEM instruction needs 4 wait states after a VALU (v_readfirstlane) write of its scalar base: s_mov, s_mov, s_nop 1.
  const uintptr_t b = (uintptr_t)sbase;
  const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)b), bhi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  sbase = (const void*)(((uintptr_t)bhi << 32) | blo);
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_addr)
               : "memory");
}

// the same piece inside the pipelined loop, where every instruction costs an issue slot next to the MFMAs (in-kernel
// stamps: 475 cycles per tile for 8 pieces of 12 instructions each).  The tile's scalar bases are made once per
// iteration; a piece is then M0 = lds_base + LOFF, one wait state, the load with an immediate offset: 3 instructions.
// The instruction's immediate offset GOFF is added to the global address AND to the LDS address (M0 + GOFF + 16 lane):
// LOFF carries only the part of the piece's LDS offset that the immediate does not.
// M0 is NOT saved: hipcc emits no M0 use of its own in this kernel, which tools/diag/audit_flash_isa.py verifies.
// The lane offset of piece i is vlane_w ^ XORC (a compile-time constant per piece: the swizzle term is XOR-linear in the
// piece number), formed in the statement: eight precomputed offsets cost eight registers this kernel does not have.
template <int LOFF>
__device__ __forceinline__ void fl_dma_set_m0(unsigned lds_base) {
  asm volatile("s_add_u32 m0, %0, %1" : : "s"(lds_base), "n"(LOFF) : "scc");
}
// ... and M0 is written only where LOFF changes (twice per tile, two gaps ahead of the piece that needs it: no wait state
// to pad), so that a piece is TWO instructions beside the MFMAs.
template <int GOFF, int XORC>
__device__ __forceinline__ void fl_dma16_at(unsigned vlane_w, const void* sbase) {
  unsigned voff;
  asm volatile("v_xor_b32 %0, %3, %1\n\tglobal_load_lds_dwordx4 %0, %2 offset:%4"
               : "=&v"(voff)
               : "v"(vlane_w), "s"(sbase), "n"(XORC), "n"(GOFF)
               : "memory");
}

// max of three without the canonicalising v_max(x, x) hipcc puts in front of fmaxf on MFMA outputs (NaNs propagate to
// the loss either way: a NaN score makes the log-sum-exp NaN)
__device__ __forceinline__ float fl_max3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// ---- softmax micro-ops: ONE volatile asm statement each, so that each stays in the MFMA gap it is written in (volatile
// statements keep their order; hipcc sinks and clumps plain C++ arithmetic, see the kernel).  Hazards the assembler does
// not pad: an exponential's result needs one wait state before a VALU reads it (callers never put the consumer next to
// it); a DPP source needs two wait states behind the VALU that wrote it (NOP = true adds them).
// The exponential and the addition name what they overwrite as an INPUT only (as fl_mfma_s does, and for the same
// reason: hipcc pads a wait state in front of an asm statement that reads what the asm statement before it wrote).  The
// values stay opaque to hipcc: the prescaled scores are asm outputs, and fl_v_opaque(lsum) hands it a new value of the
// running sum once per iteration and before the sum is read.
__device__ __forceinline__ void fl_v_exp(const float& x) { asm volatile("v_exp_f32 %0, %0" : : "v"(x)); }
__device__ __forceinline__ void fl_v_add(const float& acc, const float& p) { asm volatile("v_add_f32 %0, %0, %1" : : "v"(acc), "v"(p)); }
__device__ __forceinline__ void fl_v_opaque(float& x) { asm volatile("; fl_opaque" : "+v"(x)); }  // marker for the ISA audit
__device__ __forceinline__ unsigned fl_v_cvt_pk(const float& lo, const float& hi) {
  unsigned d;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(lo), "v"(hi));
  return d;
}
// x = s * log2(e) + off
__device__ __forceinline__ void fl_v_prescale(float& x, const float& s, const float& off) {
  asm volatile("v_fmamk_f32 %0, %1, 0x3fb8aa3b, %2" : "=v"(x) : "v"(s), "v"(off));
}
__device__ __forceinline__ void fl_v_max(float& d, const float& a, const float& b) {
  asm volatile("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void fl_v_max3(float& d, const float& a, const float& b) {
  asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(d) : "v"(a), "v"(b));
}
// step STEP of the wave maximum (the six DPP controls of wave_max_uniform): afterwards lane 63 holds the maximum.  The
// register is named as an input only (see fl_v_exp); the value is written by an asm statement before the first step and
// read by one (v_readlane in the decision statement) after the last.
template <int STEP, bool NOP>
__device__ __forceinline__ void fl_dpp_max(const float& v) {
#define MI_FL_DPP(ctrl)                                                                            \
  if constexpr (NOP) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " ctrl : : "v"(v));          \
  else asm volatile("v_max_f32_dpp %0, %0, %0 " ctrl : : "v"(v))
  if constexpr (STEP == 0) { MI_FL_DPP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"); }
  else if constexpr (STEP == 1) { MI_FL_DPP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"); }
  else if constexpr (STEP == 2) { MI_FL_DPP("row_half_mirror row_mask:0xf bank_mask:0xf"); }
  else if constexpr (STEP == 3) { MI_FL_DPP("row_mirror row_mask:0xf bank_mask:0xf"); }
  else if constexpr (STEP == 4) { MI_FL_DPP("row_bcast:15 row_mask:0xa bank_mask:0xf"); }
  else { MI_FL_DPP("row_bcast:31 row_mask:0xc bank_mask:0xf"); }
#undef MI_FL_DPP
}

template <int N>
__device__ __forceinline__ void fl_wait_vmcnt_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void fl_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    fl_static_for<B + 1, E>(f);
  }
}

// MFMA wrappers with explicit register classes.  The 512 registers of a one-wave-per-SIMD kernel are 256 architectural
// VGPRs plus 256 accumulator registers; hipcc (ROCm 7.2) keeps every MFMA A / B operand in the first file and, with the
// 256 output accumulators in the second, spilled all 128 registers of stationary rows to scratch (215 spills, reloaded
// every tile).  The hardware takes A, B and C / D from either file, so the split is made by hand: output tiles
// [0, OA) and stationary fragments [0, QA) live in accumulator registers, the rest in VGPRs.
//
// The streamed-operand fragments go through a RING of 9 four-register slots in FIXED registers v[216:251], filled by
// LDS reads issued from asm eight MFMAs ahead of their use (hipcc would not keep that many reads in flight, and it
// treats the transposed-read builtin as aliasing the LDS-DMA in flight: an s_waitcnt vmcnt(0) per tile that drained the
// prefetch).  An asm load's destination is valid only after the reader's own counted lgkmcnt wait (guide 5.7 item 1),
// and a transposed fragment is filled by two loads, so the registers are named literally; every MFMA wrapper starts with
// the counted wait.  LDS reads the compiler issues in between only make a counted wait stronger (LDS returns in order;
// no scalar loads inside the loop).  Hazards the compiler cannot see inside an asm statement (guide 5.7 item 2):
//   * a VALU-written operand needs two wait states before the MFMA reads it: NOP = true puts an s_nop 1 in front;
//   * an MFMA result needs the matrix pipe's passes before any non-MFMA reader: fl_mfma_drain() after a chain.
constexpr int kFlRing = 9;   // slots: one more than the fragments in flight
constexpr int kFlAhead = 8;  // fragments in flight

template <int SLOT, int OFF>
__device__ __forceinline__ void fl_ring_read_s(bf16x8& f, int addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  if constexpr (SLOT == 0)
    asm volatile("ds_read_b128 v[216:219], %1 offset:%c2" : "={v[216:219]}"(f) : "v"(addr), "i"(OFF));
  else if constexpr (SLOT == 1)
    asm volatile("ds_read_b128 v[220:223], %1 offset:%c2" : "={v[220:223]}"(f) : "v"(addr), "i"(OFF));
  else if constexpr (SLOT == 2)
    asm volatile("ds_read_b128 v[224:227], %1 offset:%c2" : "={v[224:227]}"(f) : "v"(addr), "i"(OFF));
  else if constexpr (SLOT == 3)
    asm volatile("ds_read_b128 v[228:231], %1 offset:%c2" : "={v[228:231]}"(f) : "v"(addr), "i"(OFF));
  else if constexpr (SLOT == 4)
    asm volatile("ds_read_b128 v[232:235], %1 offset:%c2" : "={v[232:235]}"(f) : "v"(addr), "i"(OFF));
  else if constexpr (SLOT == 5)
    asm volatile("ds_read_b128 v[236:239], %1 offset:%c2" : "={v[236:239]}"(f) : "v"(addr), "i"(OFF));
  else if constexpr (SLOT == 6)
    asm volatile("ds_read_b128 v[240:243], %1 offset:%c2" : "={v[240:243]}"(f) : "v"(addr), "i"(OFF));
  else if constexpr (SLOT == 7)
    asm volatile("ds_read_b128 v[244:247], %1 offset:%c2" : "={v[244:247]}"(f) : "v"(addr), "i"(OFF));
  else if constexpr (SLOT == 8)
    asm volatile("ds_read_b128 v[248:251], %1 offset:%c2" : "={v[248:251]}"(f) : "v"(addr), "i"(OFF));
  else if constexpr (SLOT == 9)
    asm volatile("ds_read_b128 v[252:255], %1 offset:%c2" : "={v[252:255]}"(f) : "v"(addr), "i"(OFF));
}
template <int SLOT, int OFF_LO, int OFF_HI>
__device__ __forceinline__ void fl_ring_read_v(bf16x8& f, int addr_lo, int addr_hi) {
  static_assert(OFF_LO >= 0 && OFF_HI < 65536, "ds offset field");
  if constexpr (SLOT == 0)
    asm volatile("ds_read_b64_tr_b16 v[216:217], %1 offset:%c3\n\tds_read_b64_tr_b16 v[218:219], %2 offset:%c4"
                 : "={v[216:219]}"(f) : "v"(addr_lo), "v"(addr_hi), "i"(OFF_LO), "i"(OFF_HI));
  else if constexpr (SLOT == 1)
    asm volatile("ds_read_b64_tr_b16 v[220:221], %1 offset:%c3\n\tds_read_b64_tr_b16 v[222:223], %2 offset:%c4"
                 : "={v[220:223]}"(f) : "v"(addr_lo), "v"(addr_hi), "i"(OFF_LO), "i"(OFF_HI));
  else if constexpr (SLOT == 2)
    asm volatile("ds_read_b64_tr_b16 v[224:225], %1 offset:%c3\n\tds_read_b64_tr_b16 v[226:227], %2 offset:%c4"
                 : "={v[224:227]}"(f) : "v"(addr_lo), "v"(addr_hi), "i"(OFF_LO), "i"(OFF_HI));
  else if constexpr (SLOT == 3)
    asm volatile("ds_read_b64_tr_b16 v[228:229], %1 offset:%c3\n\tds_read_b64_tr_b16 v[230:231], %2 offset:%c4"
                 : "={v[228:231]}"(f) : "v"(addr_lo), "v"(addr_hi), "i"(OFF_LO), "i"(OFF_HI));
  else if constexpr (SLOT == 4)
    asm volatile("ds_read_b64_tr_b16 v[232:233], %1 offset:%c3\n\tds_read_b64_tr_b16 v[234:235], %2 offset:%c4"
                 : "={v[232:235]}"(f) : "v"(addr_lo), "v"(addr_hi), "i"(OFF_LO), "i"(OFF_HI));
  else if constexpr (SLOT == 5)
    asm volatile("ds_read_b64_tr_b16 v[236:237], %1 offset:%c3\n\tds_read_b64_tr_b16 v[238:239], %2 offset:%c4"
                 : "={v[236:239]}"(f) : "v"(addr_lo), "v"(addr_hi), "i"(OFF_LO), "i"(OFF_HI));
  else if constexpr (SLOT == 6)
    asm volatile("ds_read_b64_tr_b16 v[240:241], %1 offset:%c3\n\tds_read_b64_tr_b16 v[242:243], %2 offset:%c4"
                 : "={v[240:243]}"(f) : "v"(addr_lo), "v"(addr_hi), "i"(OFF_LO), "i"(OFF_HI));
  else if constexpr (SLOT == 7)
    asm volatile("ds_read_b64_tr_b16 v[244:245], %1 offset:%c3\n\tds_read_b64_tr_b16 v[246:247], %2 offset:%c4"
                 : "={v[244:247]}"(f) : "v"(addr_lo), "v"(addr_hi), "i"(OFF_LO), "i"(OFF_HI));
  else if constexpr (SLOT == 8)
    asm volatile("ds_read_b64_tr_b16 v[248:249], %1 offset:%c3\n\tds_read_b64_tr_b16 v[250:251], %2 offset:%c4"
                 : "={v[248:251]}"(f) : "v"(addr_lo), "v"(addr_hi), "i"(OFF_LO), "i"(OFF_HI));
  else if constexpr (SLOT == 9)
    asm volatile("ds_read_b64_tr_b16 v[252:253], %1 offset:%c3\n\tds_read_b64_tr_b16 v[254:255], %2 offset:%c4"
                 : "={v[252:255]}"(f) : "v"(addr_lo), "v"(addr_hi), "i"(OFF_LO), "i"(OFF_HI));
}
// s (+)= A x B with A = ring fragment (streamed rows), B = stationary fragment (either register file).
// NOTE on the accumulator: hipcc may copy an asm operand between two statements.  With a plain C++ "previous = next" at
// the end of the pipelined loop it renamed the score accumulator in the MIDDLE of the chain to coalesce that copy
// (v_mov of an MFMA result still in the matrix pipe: wrong scores, no fault).  There is no such copy any more (the
// softmax head writes the prescaled scores into registers of its own), and tools/diag/audit_flash_isa.py checks the
// compiled kernel for any compiler-issued instruction that touches the destination of an MFMA still in flight.
// WAIT >= 0: `s_waitcnt lgkmcnt(WAIT)` first (its own volatile statement: volatile asms keep their order); WAIT < 0: none.
// Only every second MFMA waits, for its own fragment and the next one's: a wait is an issue slot next to the MFMAs
// whether or not it has anything to wait for (71 of them per streamed tile before).
template <int WAIT>
__device__ __forceinline__ void fl_wait_lgkm() {
  if constexpr (kFlDiag & 16) return;
  if constexpr (WAIT >= 0) asm volatile("s_waitcnt lgkmcnt(%c0)" ::"i"(WAIT) : "memory");
}
// The ACCUMULATING form names the accumulator as an input only and writes it behind hipcc's back.  Declared "+v", hipcc
// pads one wait state (s_nop 0) between any two consecutive asm statements of which the second reads what the first
// wrote (asm statements in between do not count for it): an issue slot per MFMA of the chain, 4 of the ~24 cycles a gap
// can hide.  What makes the lie safe: (1) nothing but these statements reads the accumulator while a chain is open --
// the first reader of the finished chain sits behind `asm volatile("" : "+v"(s_next))`, which hands hipcc a NEW value in
// the same registers; (2) a copy hipcc might slip between two statements would change the register the next statement
// names: tools/diag/audit_flash_isa.py checks that all D / 16 MFMAs of a chain accumulate the same registers and that no
// other instruction touches them meanwhile.
template <bool B_IN_ACC, bool FIRST, int WAIT>
__device__ __forceinline__ void fl_mfma_s(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  fl_wait_lgkm<WAIT>();
  if constexpr (FIRST) {
    if constexpr (B_IN_ACC)
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "a"(b));
    else
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b));
  } else {
    if constexpr (B_IN_ACC)
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : : "v"(acc), "v"(a), "a"(b));
    else
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : : "v"(acc), "v"(a), "v"(b));
  }
}
// o += P x V with P (A operand) in VGPRs, V = ring fragment
template <bool ACC_IN_ACC, bool NOP, int WAIT, class AT>
__device__ __forceinline__ void fl_mfma_o(f32x16& acc, const AT& a, const bf16x8& b) {
  static_assert(sizeof(AT) == 16, "a 32x32x16 bf16 operand is four registers");
  fl_wait_lgkm<WAIT>();
  if constexpr (NOP) {
    if constexpr (ACC_IN_ACC)
      asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else
      asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  } else {
    if constexpr (ACC_IN_ACC)
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  }
}
// 20 wait states: covers the 8-pass and the 16-pass rule for "matrix result -> any other reader".
// DRAIN = false: ordering point only (no instruction).
template <bool DRAIN>
__device__ __forceinline__ void fl_score_fence(f32x16& acc) {
  if constexpr (DRAIN) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc));
  else asm volatile("" : "+v"(acc));
}
__device__ __forceinline__ void fl_mfma_drain_all() { asm volatile("s_nop 15\n\ts_nop 3" ::: "memory"); }
// tile *= f (the rare rescale of the reference point).  For tiles in accumulator registers one asm statement per
// register with ONE temporary: written as plain C++ the compiler reads all 192 accumulators into VGPRs first, and the
// register pressure of that block alone made it spill the stationary rows in the main loop.
template <bool IN_ACC>
__device__ __forceinline__ void fl_scale_tile(f32x16& t, float f) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if constexpr (IN_ACC) {
      float e = t[r], tmp;
      asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 %0, %1"
                   : "+a"(e), "=&v"(tmp)
                   : "v"(f));
      t[r] = e;
    } else {
      t[r] *= f;
    }
  }
}
template <bool IN_ACC>
__device__ __forceinline__ void fl_pin(bf16x8& v) {
  if constexpr (IN_ACC) asm volatile("" : "+a"(v));
  else asm volatile("" : "+v"(v));
}
template <bool IN_ACC>
__device__ __forceinline__ void fl_pin_o(f32x16& v) {
  if constexpr (IN_ACC) asm volatile("" : "+a"(v));
  else asm volatile("" : "+v"(v));
}

// Fragment stream of one loop iteration: F[0 .. NK) = row fragments of the NEXT tile (score product, one ds_read_b128
// each), F[NK .. NK + 2 NT) = transposed fragments of the CURRENT tile (output product, two ds_read_b64_tr_b16 each).
// lgkmcnt to wait for before the MFMA that consumes F[n] when reads up to F[min(n + AHEAD - 1, last)] have been issued.
// Waits come in groups of kFlWaitGroup: the MFMA of an n that is a multiple of it waits for F[n] ... F[n + group - 1]
// (only the reads behind those may be outstanding), the others do not wait at all (-1).
#ifndef MI_FL_WAIT_GROUP
#define MI_FL_WAIT_GROUP 2
#endif
constexpr int kFlWaitGroup = MI_FL_WAIT_GROUP;
template <int NS, int NV>
constexpr int fl_wait_count(int n) {
  if (n % kFlWaitGroup) return -1;
  int c = 0;
  for (int m = n + kFlWaitGroup; m <= n + kFlAhead - 1 && m < NS + NV; ++m) c += m < NS ? 1 : 2;
  return c;
}

template <int D, bool GRAD>
__global__ __launch_bounds__(256, 1) void bilinear_flash_kernel(FlashArgs args) {
  using C = FlashCfg<D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- which (problem, split, row block).  Every field of BOTH problem records is read before anything depends on
  // one of them: the kernel-argument segment is cold when a workgroup starts, and the natural "decode, pick a problem,
  // then read its fields" order cost four to five DEPENDENT scalar-load round trips (~11,000 cycles per workgroup, in-
  // kernel stamps).  Read up front, the loads overlap; the problem is then picked with scalar selects.
  const FlashProblem A = args.p[0], Bp = args.p[1];
  const int n_combo = args.n_combo, n_problems = args.n_problems;
  asm volatile("" ::"s"(A.q), "s"(A.kv), "s"(A.sid_q), "s"(A.sid_kv), "s"(A.m), "s"(A.n), "s"(A.diag), "s"(A.n_rb),
               "s"(A.n_split), "s"(A.tiles_per_split), "s"(A.dup), "s"(A.slab), "s"(A.rec));
  asm volatile("" ::"s"(Bp.q), "s"(Bp.kv), "s"(Bp.sid_q), "s"(Bp.sid_kv), "s"(Bp.m), "s"(Bp.n), "s"(Bp.diag), "s"(Bp.n_rb),
               "s"(Bp.n_split), "s"(Bp.tiles_per_split), "s"(Bp.dup), "s"(Bp.slab), "s"(Bp.rec), "s"(n_combo),
               "s"(n_problems));
  const int L = (int)blockIdx.x;
  int prob, split, rb;
  if (args.xcd_rows) {
    // a (problem, row block) unit and all its splits on ONE XCD (workgroups are dealt round-robin over the 8 XCDs:
    // L % 8 labels the XCD): the stationary rows are then fetched into that XCD's L2 once for all splits.  With the other
    // order -- a (problem, split) pair per XCD -- every XCD pulls a whole stationary operand through the fabric while its
    // workgroups start (32 MB at B = 4096, d = 512: ~5 us, the in-kernel stamps of the prologue).  Speed only.
    const int ns_max = A.n_split > Bp.n_split ? A.n_split : Bp.n_split;
    const int x = L & 7, k = L >> 3;
    const int unit = x + 8 * (k / ns_max);
    split = k % ns_max;
    const int n_units = A.n_rb + (n_problems == 2 ? Bp.n_rb : 0);
    if (unit >= n_units) return;
    prob = unit >= A.n_rb ? 1 : 0;
    rb = prob ? unit - A.n_rb : unit;
    if (split >= (prob ? Bp.n_split : A.n_split)) return;
  } else {
    const int combo = L % n_combo;
    rb = L / n_combo;
    if (combo < A.n_split) {
      prob = 0;
      split = combo;
    } else if (n_problems == 2 && combo < A.n_split + Bp.n_split) {
      prob = 1;
      split = combo - A.n_split;
    } else {
      return;
    }
  }
  FlashProblem P;
  P.q = prob ? Bp.q : A.q;
  P.kv = prob ? Bp.kv : A.kv;
  P.sid_q = prob ? Bp.sid_q : A.sid_q;
  P.sid_kv = prob ? Bp.sid_kv : A.sid_kv;
  P.m = prob ? Bp.m : A.m;
  P.n = prob ? Bp.n : A.n;
  P.diag = prob ? Bp.diag : A.diag;
  P.n_rb = prob ? Bp.n_rb : A.n_rb;
  P.n_split = prob ? Bp.n_split : A.n_split;
  P.tiles_per_split = prob ? Bp.tiles_per_split : A.tiles_per_split;
  P.dup = prob ? Bp.dup : A.dup;
  P.slab = prob ? Bp.slab : A.slab;
  P.rec = prob ? Bp.rec : A.rec;
  if (rb >= P.n_rb) return;
  const int64_t n_tiles_all = P.n / kFlBN;
  const int64_t tile0 = (int64_t)split * P.tiles_per_split;
  int nt = (int)(n_tiles_all - tile0 < P.tiles_per_split ? n_tiles_all - tile0 : P.tiles_per_split);
  if (nt < 0) nt = 0;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r32 = lane & 31, half = lane >> 5;
  const int64_t m_wave = (int64_t)rb * kFlRows + wave * 32;  // first stationary row of this wave
  const bool wave_active = m_wave < P.m;                      // P.m % 32 == 0: a wave is all in or all out
  const int64_t gi = wave_active ? m_wave + r32 : r32;  // an idle wave re-reads rows 0..31; its outputs are dropped
  MI_FL_STAMP(0);

  // ---- stationary rows as B fragments: lane (n = r32, half) holds Q[gi][16 kk + 8 half .. + 7].  Issued first: these
  // plain loads have the longest latency of the prologue; the LDS-DMA issue below runs under it.
  bf16x8 qf[C::NK];
  {
    // fragment-major source: the (32-row block, kk) fragment block is 1 KB in lane order -> fully coalesced loads
    const bf16_t* qblk = P.q + ((gi >> 5) * C::NK * 64 + lane) * 8;
#pragma unroll
    for (int kk = 0; kk < C::NK; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(qblk + kk * 512);
  }
  const int64_t sid_i = P.sid_q[gi];
  MI_FL_STAMP(11);
  // which streamed tiles hold a pair with equal study ids for this wave's rows (bit t: tile tile0 + t); the diagonal is
  // such a pair, so the positives only ever show up in flagged tiles
  unsigned long long dupmask, diagmask;
  {
    // flag 2: some pair of the tile shares a study id off the main diagonal -> exact id compares; flag 1: the tile's main
    // diagonal pairs this wave's samples with themselves and nothing else matches -> mask by index
    const unsigned char* df = P.dup + (wave_active ? m_wave / 32 : 0) * n_tiles_all + tile0;
    const int flag = lane < nt ? (int)df[lane] : 0;
    dupmask = __ballot(flag == 2 || (flag == 1 && args.no_index_mask));
    diagmask = __ballot(flag == 1 && !args.no_index_mask);
  }

  // ---- LDS-DMA pieces of this wave: piece q = PIECES * wave + i covers tile rows [q * RPP, (q + 1) * RPP).  Lane l
  // writes LDS chunk position cp = l % CPR of row q * RPP + l / CPR, which holds source chunk cp ^ swz(row); the swizzle
  // splits into a per-lane part and a wave-uniform part U(q), so one VGPR (vlane) serves every piece:
  //   source = [kv_base + t * STAGE + q * 1024] (scalar) + (vlane ^ 16 U(q)) (32-bit vector offset)
  // Issued from asm in the scalar-base form: as a builtin hipcc kept a 64-bit address pair per piece (16 VGPRs), spilled
  // some of them and waited vmcnt(0) on the scratch reloads inside the loop.
  const char* kv_base = reinterpret_cast<const char*>(P.kv + tile0 * kFlBN * D);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int lr = lane / C::CPR;
  const unsigned vlane = (unsigned)(lr * C::RB + 16 * ((lane % C::CPR) ^ (C::RPP > 1 ? (lr << 2) : 0)));
  // swizzle term of piece q: a permutation of q's low bits, hence u(PIECES * wave + i) = u(PIECES * wave) ^ u(i)
  auto piece_u = [](int q) __attribute__((always_inline)) {
    return C::RPP == 1 ? fl_swz(q) : (C::RPP == 2 ? (8 * (q & 1)) | ((q >> 1) & 3) : (q & 3));
  };
  const unsigned vlane_w = vlane ^ (unsigned)(16 * piece_u(C::PIECES * wave));
  auto piece_voff = [&](int i) __attribute__((always_inline)) { return vlane_w ^ (unsigned)(16 * piece_u(i)); };
  auto issue_piece = [&](int t, int i) __attribute__((always_inline)) {
    if ((kFlDiag & 1) && t >= 3) return;
    const int q = C::PIECES * wave + i;
    const int stage = t & (kFlStages - 1);
    fl_dma16(piece_voff(i), kv_base + (int64_t)t * C::STAGE + q * 1024, lds0 + stage * C::STAGE + q * 1024);
  };
  auto issue_tile = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < C::PIECES; ++i) issue_piece(t, i);
  };
  MI_FL_STAMP(12);
  if (nt > 0) issue_tile(0);
  if (nt > 1) issue_tile(1);
  if (nt > 2) issue_tile(2);
  MI_FL_STAMP(13);

  // streamed study ids of this split -> LDS (plain loads: done before the pipeline's counted waits start)
  {
    // all loads first, then the LDS writes: as a plain copy loop hipcc waited for each load before its store, four to
    // eight dependent L2 round trips in the prologue of every workgroup
    int64_t* sl = reinterpret_cast<int64_t*>(smem + C::SID_OFF);
    const int64_t* sg = P.sid_kv + tile0 * kFlBN;
    constexpr int PER = kFlMaxTilesPerSplit * kFlBN / 256;
    int64_t v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] = tid + 256 * k < nt * kFlBN ? sg[tid + 256 * k] : 0;
#pragma unroll
    for (int k = 0; k < PER; ++k)
      if (tid + 256 * k < nt * kFlBN) sl[tid + 256 * k] = v[k];
  }

  constexpr int NO = GRAD ? C::NT : 1;
  f32x16 o[NO];
#pragma unroll
  for (int c = 0; c < NO; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[c][r] = 0.0f;
  float lsum = 0.0f, pos = 0.0f;
  unsigned cnt = 0;                // wave-uniform
  float mref = MI_NEG_INF;         // wave-uniform reference point of the exponentials

  // per-lane constants of the fragment addresses (derivation: DESIGN.md section 4.1, "Fragment addressing"; the same
  // formulas restated in numpy and checked end to end: tests/test_flash_addressing.py)
  const int fxh = half ^ fl_swz(r32);                       // row read: chunk (2 kk + half) ^ swz(row)
  const int a0_lane = r32 * C::RB + 16 * fxh;
  const int g16 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int cl = 2 * (g16 & 1) + (p4 >> 1);
  const int a1_lane = (4 * half + q4) * C::RB + 16 * ((cl ^ half) | (q4 << 2)) + 8 * (p4 & 1);
  const int d0_wave = (int)(m_wave + P.diag - tile0 * kFlBN);  // streamed offset of the wave's first positive

  // the Q loads, the id copy and the first tiles must have landed (the asm-issued LDS-DMA is invisible to hipcc's own
  // wait insertion: drain it by hand)
  MI_FL_STAMP(14);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  MI_FL_STAMP(15);
  __syncthreads();
  MI_FL_STAMP(1);
  // first "use" of the plain loads here, so that hipcc places their vmcnt wait in the prologue and not at the first
  // compare of the general mask path inside the loop (where it would drain the LDS-DMA prefetch)
  asm volatile("" ::"v"(sid_i), "s"(dupmask), "s"(diagmask));
  // pin the register classes once: from here on only the asm MFMAs touch these values
  fl_static_for<0, C::NK>([&](auto KK) __attribute__((always_inline)) { fl_pin<(decltype(KK)::value < C::QA)>(qf[decltype(KK)::value]); });
  if constexpr (GRAD) fl_static_for<0, C::NT>([&](auto CT) __attribute__((always_inline)) { fl_pin_o<(decltype(CT)::value < C::OA)>(o[decltype(CT)::value]); });

  bf16x8 ring[kFlRing];
  u32x4 pfw[2];    // P of the tile being multiplied, packed bf16 pairs: the A operand of the output product
  f32x16 s_next;   // scores of the newest tile (the score product's accumulator)
  float xs[16];    // the pending tile: prescaled scores (x - m_ref) log2(e), then -- in place -- their exponentials
  float tA = MI_NEG_INF, tB = MI_NEG_INF, off = 0.0f;
  // wave-uniform values of the reference point, kept per lane (every lane the same number): gfx950 has no scalar float
  // compare, and selects on SGPR values turn into branches
  float f_pend = 1.0f;  // a raise of the reference point decided beside the output product is APPLIED to the sums at the
                        // next iteration's start (f_pend != 1: exp(old - new)); every lane the same value

  // ---- softmax of a tile as two lists of micro-ops, each ONE volatile asm statement placed in a chosen MFMA gap.
  // Written as plain C++ in "slices" hipcc undid the placement (ISA of round 2: the sixteen lsum additions sunk into one
  // dependent chain behind the last MFMA, eight of the exponentials and their packing in a single gap, the wave maximum as
  // thirty instructions in another; ~690 cycles per tile that no MFMA hid).  Volatile statements keep their order
  // relative to the MFMAs and reads (which are volatile statements too).
  //  * HEAD of tile tn (scores in s_next), beside the OUTPUT product of tile tn - 1: lane maxima, wave maximum (6 DPP
  //    steps), reference-point decision, x = s * log2(e) - m_ref * log2(e) into xs[].  Writing xs[] here replaces the
  //    sixteen register copies "previous = next" of round 2.  NO branch: a branch beside the MFMAs -- above all a TAKEN
  //    one, which refills the wave's instruction buffer -- cost more than everything else of the softmax together
  //    (timing experiments of round 3: the head's slots 0 - 10 with their four branches 470 cycles per tile, the sixteen
  //    exponentials nothing).  Tiles with masked pairs (the diagonal, duplicated study ids: rare) are therefore treated
  //    as unmasked here and REDONE at the next iteration's start (fix_masked_tile).
  //  * TAIL of tile tp (xs[]), beside the SCORE product of tile tp + 1: p = exp2(x) in place, lsum += p one slot later
  //    (no dependent back-to-back pair: an exponential's consumer needs a wait state), cvt_pk pairs two slots later.
  constexpr int kTailSlots = 17, kHeadSlots = 17;
  auto tail_slot = [&](auto S_) __attribute__((always_inline)) {
    constexpr int sl = decltype(S_)::value;
    if constexpr (kFlDiag & 2) return;
    if constexpr (sl < 16 && !(kFlDiag & 32)) fl_v_exp(xs[sl]);
    if constexpr (sl >= 1 && sl <= 16 && !(kFlDiag & 64)) fl_v_add(lsum, xs[sl - 1]);
    if constexpr (GRAD && sl >= 2 && sl <= 16 && sl % 2 == 0 && !(kFlDiag & 128)) {
      constexpr int k = sl / 2 - 1;  // dword k & 3 of fragment k >> 2 = (bf16(p[2k]), bf16(p[2k + 1]))
      pfw[k >> 2][k & 3] = fl_v_cvt_pk(xs[2 * k], xs[2 * k + 1]);
    }
  };
  // The reference-point decision, branch-free, in ONE statement and -- behind the v_readlane that broadcasts the wave
  // maximum (lane 63 of `tmaxv`) -- on the vector unit only: every hand-over between the vector and the scalar unit
  // (v_cmp -> s_and, SALU-written VCC -> v_cndmask) stalls the one wave on its SIMD for tens of cycles.
  //   raise  = tmax > mref + thr      (mref = -inf: -inf + thr = -inf, so any finite maximum raises; thr = +inf for a
  //            tile with masked pairs: its unmasked HEAD must not move the reference point)
  //   f_pend = raise, from a FINITE reference point ? exp2((mref - tmax) log2 e) : 1     (1: nothing to rescale)
  //   off    = -log2(e) max(mref, -1e30)  (all entries masked so far: x = -inf * c + 1.4e30 = -inf, exp2 gives 0)
  auto decide = [&](const float& tmaxv, float thr) __attribute__((always_inline)) {
    float tmax_s, tmax_v, m_old;
    asm volatile(
        "s_nop 0\n\t"  // (gfx950 wait states the assembler does not pad: VALU write -> v_readlane 1; VALU-written SGPR /
        "v_readlane_b32 %[t], %[ta], 63\n\t"  //  VCC -> VALU read 2)
        "v_mov_b32 %[mo], %[m]\n\t"
        "v_add_f32 %[f], %[th], %[m]\n\t"
        "v_mov_b32 %[tv], %[t]\n\t"
        "v_cmp_gt_f32 vcc, %[tv], %[f]\n\t"
        "v_sub_f32 %[f], %[m], %[tv]\n\t"
        "v_mul_f32 %[f], 0x3fb8aa3b, %[f]\n\t"
        "v_exp_f32 %[f], %[f]\n\t"
        "v_cndmask_b32 %[m], %[m], %[tv], vcc\n\t"
        "s_nop 0\n\t"  // (the exponential's consumer)
        "v_cndmask_b32 %[f], 1.0, %[f], vcc\n\t"
        "v_cmp_class_f32 vcc, %[mo], 4\n\t"
        "v_max_f32 %[o], 0xf149f2ca, %[m]\n\t"
        "v_mul_f32 %[o], 0xbfb8aa3b, %[o]\n\t"
        "v_cndmask_b32 %[f], %[f], 1.0, vcc"
        : [t] "=&s"(tmax_s), [tv] "=&v"(tmax_v), [mo] "=&v"(m_old), [f] "+v"(f_pend), [m] "+v"(mref),
          [o] "+v"(off)  // ("+": written in place, so that the rare second decision of a masked tile adds no copies)
        : [ta] "v"(tmaxv), [th] "s"(thr)
        : "vcc");
  };
  // NOPS: the ops run back to back (no MFMA and reads between them): a DPP step then needs two wait states behind the
  // VALU write of its source
  auto head_slot = [&](auto V_, auto NOPS_, float thr) __attribute__((always_inline)) {
    constexpr int v = decltype(V_)::value;
    constexpr bool NOPS = decltype(NOPS_)::value;
    if constexpr (kFlDiag & 2) return;
    if constexpr ((kFlDiag & 256) && v <= 10) return;
    if constexpr ((kFlDiag & 512) && v >= 11) return;
    if constexpr ((kFlDiag & 1024) && v <= 3) return;             // ... no lane maxima
    if constexpr ((kFlDiag & 2048) && v >= 4 && v <= 9) return;   // ... no wave maximum
    if constexpr ((kFlDiag & 4096) && v == 10) return;            // ... no decision
    if constexpr (v == 0) {
      asm volatile("" : "+v"(s_next));  // nothing below reads the scores earlier than this point of the stream
      cnt += 1024u;
      fl_v_max(tA, s_next[0], s_next[1]);
      fl_v_max(tB, s_next[2], s_next[3]);
    } else if constexpr (v >= 1 && v <= 3) {
      fl_v_max3(tA, s_next[4 * v], s_next[4 * v + 1]);
      fl_v_max3(tB, s_next[4 * v + 2], s_next[4 * v + 3]);
    } else if constexpr (v == 4) {
      fl_v_max(tA, tA, tB);
      fl_dpp_max<0, true>(tA);  // (behind the merge: always two wait states)
    } else if constexpr (v >= 5 && v <= 9) {
      fl_dpp_max<v - 4, NOPS>(tA);
    } else if constexpr (v == 10) {
      decide(tA, thr);
    } else if constexpr (v >= 11 && v <= 14) {
      constexpr int r = 3 * (v - 11);
      fl_v_prescale(xs[r], s_next[r], off);
      fl_v_prescale(xs[r + 1], s_next[r + 1], off);
      fl_v_prescale(xs[r + 2], s_next[r + 2], off);
    } else if constexpr (v == 15 || v == 16) {
      constexpr int r = 12 + 2 * (v - 15);
      fl_v_prescale(xs[r], s_next[r], off);
      fl_v_prescale(xs[r + 1], s_next[r + 1], off);
    }
  };
  // A tile with masked pairs for this wave's rows, behind its (unmasked) HEAD: count, positives, and the decision and the
  // prescaling AGAIN with the masked entries left out -- the wave maximum must not see them (a trained critic's positives
  // sit far above its negatives: a reference point at a positive underflows every negative's exponential).  Rare, runs
  // on its own at the iteration's start; the scores are still in s_next (the next score chain has not started).
  auto fix_masked_tile = [&](int tn) __attribute__((always_inline)) {
    unsigned mbits = 0u;  // bit r: accumulator element r (streamed row (r & 3) + 8 (r >> 2) + 4 half) is masked
    float tq;
    unsigned long long m1;
    cnt -= 1024u;
    const int d0 = d0_wave - tn * kFlBN;  // streamed row (inside this tile) of stationary row 0's positive
    const int want = r32 + d0 - 4 * half;  // ... of this lane's positive, as an accumulator row index
    const bool has_pos = want >= 0 && want < 32 && !(want & 4) && d0 > -32 && d0 < 32;
    const unsigned pbit = has_pos ? 1u << ((want & 3) + 4 * (want >> 3)) : 0u;
    if (__builtin_amdgcn_readfirstlane((int)((dupmask >> tn) & 1ull))) {
      MI_FL_STAMP(5);
      // some pair of this tile shares a study id (always so on the diagonal): exact 64-bit compares, one id at a time
      // (batched, the sixteen ids are 32 registers this kernel does not have)
      const int64_t* sl = reinterpret_cast<const int64_t*>(smem + C::SID_OFF) + tn * kFlBN + 4 * half;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool neg = sl[(r & 3) + 8 * (r >> 2)] != sid_i;
        cnt += (unsigned)__popcll(__ballot(neg));
        mbits |= neg ? 0u : (1u << r);
        asm volatile("" : "+v"(mbits) : : "memory");  // (one id at a time)
      }
      MI_FL_STAMP(6);
    } else {
      // only the tile's main diagonal is masked: this wave's 32 positives.  No id loads.
      mbits = pbit;
      cnt += 1024u - 32u;
    }
    // Everything that lives on is changed IN PLACE by asm statements ("+v"): a new value per path would be a register
    // copy at the join for hipcc, which has none to spare.
    const float ninf = MI_NEG_INF;
    // (1) the positives.  (2) Did the unmasked HEAD see a maximum that WOULD have raised the reference point?  (tA still
    // holds the wave maximum over ALL entries in lane 63.)  If not -- the usual case: an untrained critic's positives look
    // like its negatives, and a trained one's reference point already sits high -- the prescaled scores of the unmasked
    // entries stand as the HEAD wrote them and only the masked ones are set to -inf.
    const float tmax_all = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tA), 63));
    const bool redo = __builtin_amdgcn_readfirstlane((int)(tmax_all > mref + kFlThr || mref == MI_NEG_INF)) != 0;
    if (!redo) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        unsigned t, u;
        const float s0 = s_next[r], s1 = s_next[r + 1];
        // pos += positive ? s : 0;  x = masked ? -inf : x   (two elements per statement: each compare's two wait states
        // are the other element's instructions)
        asm volatile("v_and_b32 %[t], %[b0], %[pb]\n\t"
                     "v_and_b32 %[u], %[b1], %[pb]\n\t"
                     "v_cmp_ne_u32 vcc, 0, %[t]\n\t"
                     "v_cmp_ne_u32 %[m1], 0, %[u]\n\t"
                     "v_and_b32 %[t], %[b0], %[mb]\n\t"
                     "v_and_b32 %[u], %[b1], %[mb]\n\t"
                     "v_cndmask_b32 %[q], 0, %[s0], vcc\n\t"
                     "v_add_f32 %[p], %[p], %[q]\n\t"
                     "v_cndmask_b32 %[q], 0, %[s1], %[m1]\n\t"
                     "v_add_f32 %[p], %[p], %[q]\n\t"
                     "v_cmp_ne_u32 vcc, 0, %[t]\n\t"
                     "v_cmp_ne_u32 %[m1], 0, %[u]\n\t"
                     "s_nop 1\n\t"
                     "v_cndmask_b32 %[x0], %[x0], %[ni], vcc\n\t"
                     "v_cndmask_b32 %[x1], %[x1], %[ni], %[m1]"
                     : [t] "=&v"(t), [u] "=&v"(u), [q] "=&v"(tq), [m1] "=&s"(m1), [p] "+v"(pos), [x0] "+v"(xs[r]), [x1] "+v"(xs[r + 1])
                     : [b0] "n"(1 << r), [b1] "n"(2 << r), [mb] "v"(mbits), [pb] "v"(pbit), [s0] "v"(s0), [s1] "v"(s1), [ni] "v"(ninf)
                     : "vcc");
      }
      return;
    }
    // the full redo: lane maximum over the unmasked entries, wave maximum, decision, prescaling
    float tM = MI_NEG_INF;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      unsigned t;
      const float sr = s_next[r];
      // tM = max(tM, masked ? -inf : s);  pos += positive ? s : 0
      asm volatile("v_and_b32 %[t], %[b], %[mb]\n\t"
                   "v_cmp_eq_u32 vcc, 0, %[t]\n\t"
                   "s_nop 1\n\t"  // (VALU-written VCC -> VALU read: two wait states on gfx950)
                   "v_cndmask_b32 %[t], %[ni], %[s], vcc\n\t"
                   "v_max_f32 %[tm], %[tm], %[t]\n\t"
                   "v_and_b32 %[t], %[b], %[pb]\n\t"
                   "v_cmp_ne_u32 vcc, 0, %[t]\n\t"
                   "s_nop 1\n\t"
                   "v_cndmask_b32 %[t], 0, %[s], vcc\n\t"
                   "v_add_f32 %[p], %[p], %[t]"
                   : [t] "=&v"(t), [tm] "+v"(tM), [p] "+v"(pos)
                   : [b] "n"(1 << r), [mb] "v"(mbits), [pb] "v"(pbit), [s] "v"(sr), [ni] "v"(ninf)
                   : "vcc");
    }
    float tmaxv = wave_max_uniform(tM);
    asm volatile("s_nop 1" : "+v"(tmaxv));
    decide(tmaxv, kFlThr);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      unsigned t;
      const float sr = s_next[r];
      // x = masked ? -inf : s * log2(e) + off   (exp2 then gives 0)
      asm volatile("v_fmamk_f32 %[x], %[s], 0x3fb8aa3b, %[o]\n\t"
                   "v_and_b32 %[t], %[b], %[mb]\n\t"
                   "v_cmp_eq_u32 vcc, 0, %[t]\n\t"
                   "s_nop 1\n\t"
                   "v_cndmask_b32 %[x], %[ni], %[x], vcc"
                   : [x] "+v"(xs[r]), [t] "=&v"(t)
                   : [b] "n"(1 << r), [mb] "v"(mbits), [s] "v"(sr), [o] "v"(off), [ni] "v"(ninf)
                   : "vcc");
    }
  };

  // does tile t hold masked pairs for this wave's rows?  (wave-uniform; the waves of a workgroup may differ)
  const unsigned long long spmask = dupmask | diagmask;

  // ---- one loop iteration: [score product of tile tn = tp + 1 (HAS_S)] with the TAIL of tile tp in its MFMA gaps, then
  // [output product of tile tp (HAS_V)] with, in its gaps, the HEAD of tile tn and the LDS-DMA pieces of tile tp + 3.
  // Where a product is missing (first / last tile, forward-only kernel) the list it would have carried runs on its own.
  auto iteration = [&](auto HAS_S_, auto HAS_V_, int tp) __attribute__((always_inline)) {
    constexpr bool HAS_S = decltype(HAS_S_)::value, HAS_V = decltype(HAS_V_)::value && GRAD;
    constexpr bool DO_TAIL = decltype(HAS_V_)::value;  // a finished tile is waiting for its exponentials
    constexpr int NS = HAS_S ? C::NK : 0, NV = HAS_V ? 2 * C::NT : 0, NF = NS + NV;
    // the tile whose HEAD this iteration carries (tp + 1) holds masked pairs for this wave's rows: its unmasked HEAD then
    // leaves the reference point alone (threshold +inf) and fix_masked_tile decides at the iteration's end
    const bool masked_next = __builtin_amdgcn_readfirstlane((int)((spmask >> ((tp + 1) & 63)) & 1ull)) != 0;
    const float thr_next = masked_next ? __builtin_inff() : kFlThr;
    // A raise of a finite reference point by the previous HEAD: every sum still stands at the old point (this tile's exponentials, taken
    //    against the new one, have not been added yet) -- applied here, when every product of the old point is issued.
    // (first thing of the iteration: before the fragment addresses and the first reads are live)
    fl_v_opaque(lsum);
    if constexpr (DO_TAIL && !(kFlDiag & 2)) {
      if (__builtin_expect(__builtin_amdgcn_readfirstlane((int)__float_as_uint(f_pend)) != 0x3f800000, 0)) {
        lsum *= f_pend;
        fl_v_opaque(lsum);
        if constexpr (GRAD) {
          fl_mfma_drain_all();
          fl_static_for<0, C::NT>([&](auto CT) __attribute__((always_inline)) { fl_scale_tile<(decltype(CT)::value < C::OA)>(o[decltype(CT)::value], f_pend); });
        }
        f_pend = 1.0f;
        fl_v_opaque(f_pend);
      }
    }
    const int so = ((tp + 1) & (kFlStages - 1)) * C::STAGE + a0_lane;  // row reads of tile tp + 1
    const int vo = (tp & (kFlStages - 1)) * C::STAGE + a1_lane;       // transposed reads of tile tp
    int abase[8], tbase[8];
    if constexpr (HAS_S) {
#pragma unroll
      for (int v = 0; v < 8; ++v) abase[v] = so ^ (32 * v);
    }
    if constexpr (HAS_V) {
#pragma unroll
      for (int v = 0; v < 8; ++v) tbase[v] = vo ^ (16 * ((4 * (v >> 1)) ^ (2 * (v & 1))));
    }
    auto issue_read = [&](auto NI) __attribute__((always_inline)) {
      constexpr int n = decltype(NI)::value;
      if constexpr (kFlDiag & 8) return;
      if constexpr (n < NS) {
        fl_ring_read_s<n % kFlRing, 256 * (n >> 3)>(ring[n % kFlRing], abase[n & 7]);
      } else if constexpr (n < NF) {
        constexpr int u = n - NS, ks = u / C::NT, ct = u % C::NT;
        fl_ring_read_v<n % kFlRing, 256 * (ct >> 2) + 16 * ks * C::RB, 256 * (ct >> 2) + (16 * ks + 8) * C::RB>(
            ring[n % kFlRing], tbase[2 * (ct & 3)], tbase[2 * (ct & 3) + 1]);
      }
    };
    fl_static_for<0, kFlAhead>([&](auto NI) __attribute__((always_inline)) { issue_read(NI); });
    __builtin_amdgcn_sched_barrier(0);
    // score product; the TAIL slots spread over its gaps in order
    fl_static_for<0, NS>([&](auto NI) __attribute__((always_inline)) {
      constexpr int n = decltype(NI)::value;
      fl_mfma_s<(n < C::QA), n == 0, fl_wait_count<NS, NV>(n)>(s_next, ring[n % kFlRing], qf[n]);
      issue_read(std::integral_constant<int, n + kFlAhead>{});
      if constexpr (DO_TAIL) {
        constexpr int NSD = NS > 0 ? NS : 1;
        constexpr int per = (kTailSlots + NSD - 1) / NSD;  // 1 at D = 512, 2 at 256, 3 at 128
        constexpr int lo = n * per, hi = (lo + per < kTailSlots ? lo + per : kTailSlots);
        fl_static_for<(lo < kTailSlots ? lo : kTailSlots), hi>([&](auto SI) __attribute__((always_inline)) { tail_slot(SI); });
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (DO_TAIL && !HAS_S) {
      fl_static_for<0, kTailSlots>([&](auto SI) __attribute__((always_inline)) { tail_slot(SI); });
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (HAS_S && !HAS_V) {
      // no output product to carry the new tile's HEAD (the pipeline's first tile; the forward-only kernel)
      fl_score_fence<true>(s_next);
      fl_static_for<0, kHeadSlots>([&](auto VI) __attribute__((always_inline)) { head_slot(VI, std::true_type{}, thr_next); });
      __builtin_amdgcn_sched_barrier(0);
    }
    // output product + HEAD of the next tile + LDS-DMA issue for tile tp + 3
    if constexpr (HAS_V) {
      constexpr int EVERY = NV / C::PIECES;
      // The pieces of tile tp + 3 are issued UNCONDITIONALLY: behind the last tile the source index is clamped (the stage
      // they land in belongs to tile tp - 1, which nobody reads again), so that no gap carries a branch and the counted
      // waits of the loop are the same in every iteration.  At most two tiles per workgroup are fetched for nothing.
      const int tq = tp + 3 < nt ? tp + 3 : nt - 1;
      // scalar bases of that tile for this wave's pieces (uniform by construction; readfirstlane proves it to hipcc)
      const uintptr_t gb = (uintptr_t)(kv_base + (int64_t)tq * C::STAGE + wave * (C::PIECES * 1024));
      const unsigned gb_lo = __builtin_amdgcn_readfirstlane((unsigned)gb);  // (the builtin returns int: keep the halves
      const unsigned gb_hi = __builtin_amdgcn_readfirstlane((unsigned)(gb >> 32));  // unsigned, or the low one sign-extends)
      const void* sb0 = (const void*)(((uintptr_t)gb_hi << 32) | gb_lo);
      const void* sb1 = (const char*)sb0 + 4096;
      const unsigned lb = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(((tp + 3) & (kFlStages - 1)) * C::STAGE) +
                                                         (unsigned)(wave * (C::PIECES * 1024)));
      // gaps that may carry HEAD slots: from the sixth on (the score chain ended >= 5 MFMAs and their reads earlier: the
      // matrix pipe's result is readable), and not the ones that carry an LDS-DMA piece
      constexpr int U0 = 5;
      constexpr int n_avail = (NV - U0) - (NV / EVERY - U0 / EVERY);  // gaps u >= U0 without a piece (pieces sit at u % EVERY == EVERY - 1)
      static_assert(n_avail >= 1, "no gap for the softmax head");
      fl_static_for<0, NV>([&](auto UI) __attribute__((always_inline)) {
        constexpr int u = decltype(UI)::value, n = NS + u, ks = u / C::NT, ct = u % C::NT;
        fl_mfma_o<(ct < C::OA), (u % C::NT == 0), fl_wait_count<NS, NV>(n)>(o[ct], pfw[ks], ring[n % kFlRing]);
        issue_read(std::integral_constant<int, n + kFlAhead>{});
        constexpr bool dma_gap = u % EVERY == EVERY - 1;
        if constexpr (HAS_S && u >= U0 && !dma_gap) {
          constexpr int k = (u - U0) - (u / EVERY - U0 / EVERY);  // index among the available gaps
          constexpr int lo = k * kHeadSlots / n_avail, hi = (k + 1) * kHeadSlots / n_avail;
          constexpr bool nops = kHeadSlots > n_avail;  // several slots share a gap
          fl_static_for<lo, hi>([&](auto VI) __attribute__((always_inline)) { head_slot(VI, std::integral_constant<bool, nops>{}, thr_next); });
        }
        // M0 (the pieces' LDS base) changes with i & ~3: set it two gaps ahead of pieces 0 and 4
        if constexpr (HAS_S && !(kFlDiag & 1) && u % EVERY == EVERY - 3 && ((u / EVERY) & 3) == 0 && u / EVERY < C::PIECES)
          fl_dma_set_m0<((u / EVERY) & ~3) * 1024>(lb);
        if constexpr (dma_gap && HAS_S && !(kFlDiag & 1)) {  // (not behind the last tile: no barrier protects its stage)
          constexpr int i = u / EVERY;
          fl_dma16_at<(i & 3) * 1024,
                      16 * (C::RPP == 1 ? fl_swz(i) : (C::RPP == 2 ? (8 * (i & 1)) | ((i >> 1) & 3) : (i & 3)))>(
              vlane_w, i < 4 ? sb0 : sb1);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    } else if constexpr (DO_TAIL) {
      if (tp + 3 < nt) issue_tile(tp + 3);
    }
    // the tile whose HEAD just ran holds masked pairs for this wave's rows (rare): redo its decision and prescaling
    if constexpr (HAS_S && !(kFlDiag & 2)) {
      if (__builtin_expect(masked_next, 0)) fix_masked_tile(tp + 1);
    }
  };
  using T_ = std::true_type;
  using F_ = std::false_type;

  if (nt > 0) {
    // tile 0's scores and HEAD, nothing to overlap with yet (tp = -1: the "next" tile is tile 0)
    iteration(T_{}, F_{}, -1);
    for (int t = 0; t + 1 < nt; ++t) {
      if (t == nt / 2) MI_FL_STAMP(2);
      // tile t + 1 must have landed (own pieces, then everybody's); every wave is done with tile t - 1, whose stage the
      // pieces issued in this iteration refill.  Outstanding here: tiles t + 1 and t + 2.
      if constexpr (kFlDiag & 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if constexpr (GRAD) fl_wait_vmcnt_barrier<C::PIECES>();  // (pieces are issued in every iteration: see there)
      else if (t + 2 < nt) fl_wait_vmcnt_barrier<C::PIECES>();
      else fl_wait_vmcnt_barrier<0>();
      if (t == nt / 2) MI_FL_STAMP(3);
      iteration(T_{}, T_{}, t);
      if (t == nt / 2) MI_FL_STAMP(4);
    }
    MI_FL_STAMP(8);
    iteration(F_{}, T_{}, nt - 1);
  }
  MI_FL_STAMP(9);

  // ---- records and partial sums
  // (the clamped pieces of the last iterations may still be in flight: nobody reads them, and s_endpgm waits for every
  // counter before the workgroup's LDS is released)
  fl_v_opaque(lsum);
  lsum = wave_sum(lsum);
  pos = wave_sum(pos);
  if (lane == 0) {
    Partial rec{mref, lsum, pos, cnt};
    if (!wave_active) rec = Partial{MI_NEG_INF, 0.0f, 0.0f, 0u};
    P.rec[((int64_t)split * P.n_rb + rb) * 4 + wave] = rec;
  }
  if constexpr (GRAD) {
    fl_mfma_drain_all();
    // tie every accumulator to a statement BEHIND the drain: hipcc is free to hoist a plain read of o[] above an asm that
    // does not name it (the audit caught a v_accvgpr_read 16 wait states after the last MFMA at D = 256)
    fl_static_for<0, C::NT>([&](auto CT) __attribute__((always_inline)) { fl_pin_o<(decltype(CT)::value < C::OA)>(o[decltype(CT)::value]); });
    const int64_t wv = ((int64_t)split * P.n_rb + rb) * 4 + wave;
    if (wave_active && args.slab_f16) {
      // one power-of-two scale per wave: the largest |element| lands in [2^13, 2^14), fp16's range is never left and
      // every element within 2^-27 of the largest keeps its 11 bits (smaller ones cannot matter to the sum)
      float amax = 0.0f;
#pragma unroll
      for (int c = 0; c < C::NT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) amax = fmaxf(amax, __builtin_fabsf(o[c][r]));
      amax = wave_max_uniform(amax);
      int k = 13 - ((int)((__float_as_uint(amax) >> 23) & 255u) - 127);
      k = amax > 0.0f ? (k < -100 ? -100 : (k > 100 ? 100 : k)) : 0;
      const float up = __uint_as_float((unsigned)(127 + k) << 23), down = __uint_as_float((unsigned)(127 - k) << 23);
      f16_t* slab16 = reinterpret_cast<f16_t*>(P.slab);
      float* unscale = reinterpret_cast<float*>(slab16 + (int64_t)P.n_split * P.n_rb * 4 * (32 * D));
      if (lane == 0) unscale[wv] = down;
      f16_t* dst = slab16 + wv * (32 * D) + lane * 8;
#pragma unroll
      for (int c = 0; c < C::NT; ++c)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          f16x8 h;
#pragma unroll
          for (int e = 0; e < 8; ++e) h[e] = (f16_t)(o[c][8 * gp + e] * up);
          *reinterpret_cast<f16x8*>(dst + c * 1024 + gp * 512) = h;
        }
    } else if (wave_active) {
      float* dst = P.slab + wv * (32 * D) + lane * 4;
#pragma unroll
      for (int c = 0; c < C::NT; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = {o[c][4 * g], o[c][4 * g + 1], o[c][4 * g + 2], o[c][4 * g + 3]};
          *reinterpret_cast<f32x4*>(dst + c * 1024 + g * 256) = v;
        }
    }
  }
  MI_FL_STAMP(10);
}

// ------------------------------------------------------------------------------------------------ equal-id tile flags
// flag[a][b] = 1 iff some image row i of block a (32 rows) and some text row j of block b share a study id (the diagonal
// pairs a sample with itself, so diagonal blocks are always flagged); flag_t is the transposed copy (problem 1 walks the
// image blocks for a fixed text block).  Exact 64-bit compares, done once per step for all B^2 / 1024 tiles: the fused
// kernel then takes its general mask path only in flagged tiles (none but the diagonal for a batch of distinct studies).
static __global__ __launch_bounds__(256) void flash_dup_flags_kernel(const int64_t* __restrict__ sid_rows,
                                                                     const int64_t* __restrict__ sid_cols, int na, int nb,
                                                                     unsigned char* __restrict__ flag,
                                                                     unsigned char* __restrict__ flag_t) {
  __shared__ int64_t rows[32];
  const int a = blockIdx.x, tid = threadIdx.x;
  if (tid < 32) rows[tid] = sid_rows[a * 32 + tid];
  __syncthreads();
  const int q = tid & 3;             // rows 8 q .. 8 q + 7
  int64_t mine[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) mine[e] = rows[8 * q + e];
  for (int b = blockIdx.y * 64 + (tid >> 2); b < nb; b += gridDim.y * 64) {
    const int64_t* c = sid_cols + (int64_t)b * 32;
    bool any = false;
    for (int j = 0; j < 32; ++j) {
      const int64_t v = c[j];
#pragma unroll
      for (int e = 0; e < 8; ++e) any |= v == mine[e];
    }
    int f = any ? 1 : 0;
    f |= __shfl_xor(f, 1);
    f |= __shfl_xor(f, 2);
    if (q == 0) {
      flag[(int64_t)a * nb + b] = (unsigned char)f;
      flag_t[(int64_t)b * na + a] = (unsigned char)f;
    }
  }
}

static inline int launch_flash_dup_flags(const int64_t* sid_rows, const int64_t* sid_cols, int64_t br, int64_t b,
                                         unsigned char* flag, unsigned char* flag_t, hipStream_t st) {
  const int na = (int)(br / 32), nb = (int)(b / 32);
  int gy = (nb + 63) / 64;
  if (gy > 8) gy = 8;
  {
    ProfScope prof_("bilinear equal-id tile flags", st);
    hipLaunchKernelGGL(flash_dup_flags_kernel, dim3((unsigned)na, (unsigned)gy), dim3(256), 0, st, sid_rows, sid_cols, na, nb,
                       flag, flag_t);
  }
  MI_LAUNCH_CHECK("flash_dup_flags_kernel");
  return MI_OK;
}

// ------------------------------------------------------------------------------------------------ slab reduce
// out[i][c] = go * ( sum_s exp(m_ref[s][i / 32] - lse) * slab_s[i][c]  -  other[i + diag][c] / n_pos )
// One workgroup per (32-row wave block, 128 columns).  Outputs: fp32 row-major (grad_y) and / or bf16 row-major and
// transposed (dT for the dW | dX products).
struct FlashReduceJob {
  const float* slab;
  const Partial* rec;
  int n_split, n_rb;
  int64_t m;              // rows
  const bf16_t* other;    // [n_other][D]: the rows subtracted on the diagonal (Yb for dT, Tb for dY)
  int64_t n_other, diag;
  float* out_f32;         // [m][D] or null
  bf16_t* out_bf;         // [m][D] or null
  bf16_t* out_bf_t;       // [D][m] or null
};
struct FlashReduceArgs {
  FlashReduceJob j[2];
  const mi_stats* stats;
  const float* grad_out;
};

template <int D, bool F16>
__global__ __launch_bounds__(256) void flash_reduce_kernel(FlashReduceArgs args) {
  kernarg_prefetch<(int)sizeof(FlashReduceArgs)>();
  __shared__ float tile[32][129];
  const FlashReduceJob& J = args.j[blockIdx.z];
  const int64_t wb = blockIdx.y;  // 32-row block
  const int cc = blockIdx.x;      // 128-column chunk
  if (wb * 32 >= J.m) return;
  const int rb = (int)(wb >> 2), w = (int)(wb & 3);
  const float go = args.grad_out ? args.grad_out[0] : 1.0f;
  const float lse = args.stats->lse;
  const float gpos = go / (float)args.stats->n_pos;
  const int tid = threadIdx.x;
  float acc[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
  // the rows subtracted on the diagonal are known now: their loads go out with the slabs', not after the first barrier
  const int64_t i0 = wb * 32;
  const int64_t c0 = (int64_t)cc * 128;
  const int orow = tid >> 3, ocs = (tid & 7) * 16;
  const int64_t ojo = i0 + orow + J.diag;
  const bool has_other = ojo >= 0 && ojo < J.n_other;
  bf16x8 oth_a = {}, oth_b = {};
  if (has_other) {
    oth_a = *reinterpret_cast<const bf16x8*>(J.other + ojo * D + c0 + ocs);
    oth_b = *reinterpret_cast<const bf16x8*>(J.other + ojo * D + c0 + ocs + 8);
  }
  const f16_t* slab16 = reinterpret_cast<const f16_t*>(J.slab);
  const float* unscale = reinterpret_cast<const float*>(slab16 + (int64_t)J.n_split * J.n_rb * 4 * (32 * D));
  // four splits per round: their loads are independent and issued together, the additions keep the split order.  A wave
  // that accumulated nothing (m = -inf) left its slab unwritten: its factor is 0 and its values are not touched.
  for (int s0 = 0; s0 < J.n_split; s0 += 4) {
    float cs[4];
    int64_t wvs[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int s = s0 + q < J.n_split ? s0 + q : J.n_split - 1;
      wvs[q] = ((int64_t)s * J.n_rb + rb) * 4 + w;
      const float m = J.rec[wvs[q]].m;
      cs[q] = (s0 + q < J.n_split && m > MI_NEG_INF) ? go * __expf(m - lse) * (F16 ? unscale[wvs[q]] : 1.0f) : 0.0f;
    }
    if constexpr (F16) {
      f16x8 v[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          v[q][u] = *reinterpret_cast<const f16x8*>(slab16 + wvs[q] * (32 * D) + (int64_t)cc * 4096 + (u * 256 + tid) * 8);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (cs[q] != 0.0f) {
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[8 * u + e] += (float)v[q][u][e] * cs[q];
        }
    } else {
      f32x4 v[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          v[q][u] = *reinterpret_cast<const f32x4*>(J.slab + wvs[q] * (32 * D) + (int64_t)cc * 4096 + (u * 256 + tid) * 4);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (cs[q] != 0.0f) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * u + e] += v[q][u][e] * cs[q];
        }
    }
  }
  if constexpr (F16) {
    // 16-byte chunk h = u * 256 + tid = ct_l * 128 + gp * 64 + lane holds accumulator registers 8 gp .. 8 gp + 7 of the
    // lane: rows 8 g + 4 (lane >> 5) + e for g = 2 gp and 2 gp + 1, column ct_l * 32 + (lane & 31)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int h = u * 256 + tid;
      const int ct_l = h >> 7, gp = (h >> 6) & 1, lane = h & 63;
      const int col = ct_l * 32 + (lane & 31);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int row0 = 8 * (2 * gp + hh) + 4 * (lane >> 5);
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[row0 + e][col] = acc[8 * u + 4 * hh + e];
      }
    }
  } else {
    // float4 index f = u * 256 + tid = ct_l * 256 + g * 64 + lane
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f = u * 256 + tid;
      const int ct_l = f >> 8, g = (f >> 6) & 3, lane = f & 63;
      const int col = ct_l * 32 + (lane & 31), row0 = 8 * g + 4 * (lane >> 5);
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[row0 + e][col] = acc[4 * u + e];
    }
  }
  __syncthreads();
  {
    // row-major outputs: thread -> row tid >> 3, 16 columns
    const int row = orow, cs = ocs;
    const int64_t i = i0 + row;
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = tile[row][cs + e];
    if (has_other) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] -= gpos * (float)oth_a[e];
        v[8 + e] -= gpos * (float)oth_b[e];
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) tile[row][cs + e] = v[e];  // the transposed output reads the tile again
    }
    if (J.out_f32) {
#pragma unroll
      for (int e = 0; e < 16; e += 4)
        *reinterpret_cast<f32x4*>(J.out_f32 + i * D + c0 + cs + e) = f32x4{v[e], v[e + 1], v[e + 2], v[e + 3]};
    }
    if (J.out_bf) {
      bf16x8 a, b;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        a[e] = (bf16_t)v[e];
        b[e] = (bf16_t)v[8 + e];
      }
      *reinterpret_cast<bf16x8*>(J.out_bf + i * D + c0 + cs) = a;
      *reinterpret_cast<bf16x8*>(J.out_bf + i * D + c0 + cs + 8) = b;
    }
  }
  if (J.out_bf_t) {
    __syncthreads();
    // transposed output [D][m]: thread -> column tid >> 1, 16 rows
    const int col = tid >> 1, rs = (tid & 1) * 16;
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[e] = (bf16_t)tile[rs + e][col];
      b[e] = (bf16_t)tile[rs + 8 + e][col];
    }
    bf16_t* dst = J.out_bf_t + (c0 + col) * J.m + i0 + rs;
    *reinterpret_cast<bf16x8*>(dst) = a;
    *reinterpret_cast<bf16x8*>(dst + 8) = b;
  }
}

// ------------------------------------------------------------------------------------------------ host side
struct FlashPlan {
  bool ok;
  int n_split[2], n_rb[2], tiles_per_split[2];
  int64_t slab_bytes[2], n_rec[2];
  bool slab_f16;
};

static inline bool flash_width_ok(int64_t d) { return d == 128 || d == 256 || d == 512; }

// problem 0: stationary [br] x streamed [b]; problem 1: stationary [b] x streamed [br]
static inline FlashPlan flash_plan(int64_t br, int64_t b, int64_t d) {
  FlashPlan fp{};
  static const bool off = getenv("MI_NO_FLASH") != nullptr;  // A/B switch: the round-1 G-materialising path
  fp.ok = !off && flash_width_ok(d) && br % 32 == 0 && b % 32 == 0 && br >= 32;
  if (!fp.ok) return fp;
  static const bool slab_f32 = getenv("MI_FLASH_SLAB_F32") != nullptr;  // A/B switch: fp32 partial sums
  fp.slab_f16 = !slab_f32;
  const int64_t m[2] = {br, b}, n[2] = {b, br};
  int64_t total_tiles = 0;
  for (int q = 0; q < 2; ++q) {
    fp.n_rb[q] = (int)((m[q] + kFlRows - 1) / kFlRows);
    total_tiles += (int64_t)fp.n_rb[q] * (n[q] / kFlBN);
  }
  int64_t tau = (total_tiles + 255) / 256;  // streamed tiles per workgroup for ~256 workgroups
  if (tau < 4) tau = 4;
  if (tau > kFlMaxTilesPerSplit) tau = kFlMaxTilesPerSplit;
  if (const char* e = getenv("MI_FLASH_TILES")) tau = atoi(e) > 0 ? atoi(e) : tau;  // A/B switch
  for (int q = 0; q < 2; ++q) {
    const int64_t tiles = n[q] / kFlBN;
    int64_t tps = tau < tiles ? tau : tiles;
    fp.n_split[q] = (int)((tiles + tps - 1) / tps);
    fp.tiles_per_split[q] = (int)((tiles + fp.n_split[q] - 1) / fp.n_split[q]);  // balanced
    fp.n_split[q] = (int)((tiles + fp.tiles_per_split[q] - 1) / fp.tiles_per_split[q]);
    fp.n_rec[q] = (int64_t)fp.n_split[q] * fp.n_rb[q] * 4;
    fp.slab_bytes[q] = fp.slab_f16 ? fp.n_rec[q] * (32 * d * 2 + 4) : fp.n_rec[q] * 32 * d * 4;
  }
  return fp;
}

template <int D, bool GRAD>
static inline int launch_flash_t(const FlashArgs& a, unsigned grid, hipStream_t st, const char* what) {
  MI_SET_DYN_SMEM((bilinear_flash_kernel<D, GRAD>), FlashCfg<D>::SMEM, "hipFuncSetAttribute(bilinear_flash_kernel)");
#ifdef MI_STAMPS
  stamp_select(what, st);
#endif
  {
    ProfScope prof_(what, st);
    hipLaunchKernelGGL((bilinear_flash_kernel<D, GRAD>), dim3(grid), dim3(256), FlashCfg<D>::SMEM, st, a);
  }
  MI_LAUNCH_CHECK(what);
  return MI_OK;
}

static inline int launch_flash(FlashArgs a, int64_t d, bool grad, hipStream_t st, const char* what) {
  int combos = a.p[0].n_split + (a.n_problems == 2 ? a.p[1].n_split : 0);
  a.n_combo = (combos + 7) / 8 * 8;
  a.diag = 0;
  static const int no_index_mask = getenv("MI_FLASH_NO_INDEX_MASK") ? 1 : 0;  // A/B switch
  a.no_index_mask = no_index_mask;
  static const int xcd_rows = getenv("MI_FLASH_XCD_COMBO") ? 0 : 1;  // A/B switch
  a.xcd_rows = xcd_rows;
  int max_rb = a.p[0].n_rb;
  if (a.n_problems == 2 && a.p[1].n_rb > max_rb) max_rb = a.p[1].n_rb;
  unsigned grid = (unsigned)(a.n_combo * max_rb);
  if (a.xcd_rows) {
    const int ns_max = a.n_problems == 2 && a.p[1].n_split > a.p[0].n_split ? a.p[1].n_split : a.p[0].n_split;
    const int n_units = a.p[0].n_rb + (a.n_problems == 2 ? a.p[1].n_rb : 0);
    grid = (unsigned)(8 * ((n_units + 7) / 8) * ns_max);
  }
  if (d == 512) return grad ? launch_flash_t<512, true>(a, grid, st, what) : launch_flash_t<512, false>(a, grid, st, what);
  if (d == 256) return grad ? launch_flash_t<256, true>(a, grid, st, what) : launch_flash_t<256, false>(a, grid, st, what);
  if (d == 128) return grad ? launch_flash_t<128, true>(a, grid, st, what) : launch_flash_t<128, false>(a, grid, st, what);
  set_error("launch_flash: unsupported width %lld", (long long)d);
  return MI_ESHAPE;
}

static inline int launch_flash_reduce(const FlashReduceArgs& a, int n_jobs, int64_t d, bool slab_f16, hipStream_t st,
                                      const char* what) {
  int64_t mmax = a.j[0].m;
  if (n_jobs == 2 && a.j[1].m > mmax) mmax = a.j[1].m;
  dim3 grid((unsigned)(d / 128), (unsigned)(mmax / 32), (unsigned)n_jobs);
  {
    ProfScope prof_(what, st);
#define MI_FL_REDUCE(DD)                                                                                  \
  if (slab_f16) hipLaunchKernelGGL((flash_reduce_kernel<DD, true>), grid, dim3(256), 0, st, a);           \
  else hipLaunchKernelGGL((flash_reduce_kernel<DD, false>), grid, dim3(256), 0, st, a)
    if (d == 512) { MI_FL_REDUCE(512); }
    else if (d == 256) { MI_FL_REDUCE(256); }
    else { MI_FL_REDUCE(128); }
#undef MI_FL_REDUCE
  }
  MI_LAUNCH_CHECK(what);
  return MI_OK;
}

}  // namespace mi
