// Shared device/host helpers for libmi_critic_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "../../include/mi_critic.h"

namespace mi {

// ------------------------------------------------------------------------------------------------ errors
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define MI_CHECK_ARG(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      ::mi::set_error(__VA_ARGS__);    \
      return MI_EINVAL;                \
    }                                  \
  } while (0)

#define MI_LAUNCH_CHECK(what)                                   \
  do {                                                          \
    hipError_t e__ = hipGetLastError();                         \
    if (e__ != hipSuccess) return ::mi::hip_fail(e__, what);    \
  } while (0)

// Optional per-kernel timing with HIP events on the launch stream (mi_profile_begin / mi_profile_end); off by default,
// one branch per launch when off.
bool profile_enabled();
void profile_push(const char* name, hipStream_t st, bool begin);
struct ProfScope {
  const char* name;
  hipStream_t st;
  bool on;
  ProfScope(const char* n, hipStream_t s) : name(n), st(s), on(profile_enabled()) {
    if (on) profile_push(name, st, true);
  }
  ~ProfScope() {
    if (on) profile_push(name, st, false);
  }
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Zeroes a few dwords of workspace (absmax slots) AS A KERNEL.  hipMemsetAsync did the job on an eager stream, but
// captured into a hipGraph the memset node did not stay ordered between the kernels around it on the second and later
// replays (round 4: the fp16 scales of a replayed concat step differed from the eager call's, gradients off by 1e-3);
// a kernel node is ordered like every other launch of the capture.
static __global__ void zero_words_kernel(unsigned* p, int n) {
  for (int e = threadIdx.x; e < n; e += 64) p[e] = 0u;
}
// ... and the same for small device-to-device copies (statistics words, counters): kernels, not memcpy nodes
static __global__ void copy_words_kernel(unsigned* dst, const unsigned* src, int n) {
  for (int e = threadIdx.x; e < n; e += 64) dst[e] = src[e];
}
static inline int launch_copy_words(void* dst, const void* src, size_t bytes, hipStream_t st, const char* what) {
  hipLaunchKernelGGL(copy_words_kernel, dim3(1), dim3(64), 0, st, (unsigned*)dst, (const unsigned*)src, (int)(bytes / 4));
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, what);
}
static inline int launch_zero_words(void* p, size_t bytes, hipStream_t st, const char* what) {
  hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, st, (unsigned*)p, (int)(bytes / 4));
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, what);
}

// Bump allocator over the caller's workspace (never allocates).
struct Workspace {
  char* base;
  size_t size;
  size_t off;
  Workspace(void* p, size_t n) : base((char*)p), size(n), off(0) {}
  template <typename T>
  T* take(size_t count) {
    off = align_up(off, 256);
    T* p = (T*)(base ? base + off : nullptr);
    off += count * sizeof(T);
    return p;
  }
  bool ok() const { return off <= size && (base != nullptr || off == 0); }
};

// ------------------------------------------------------------------------------------------------ types
typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Partial record of the bound reduction: merged in a fixed order, so results are bit-reproducible.
struct Partial {
  float m;    // max of the negative scores seen (or -inf)
  float s;    // sum exp(score - m)
  float pos;  // sum of positive scores
  unsigned cnt;  // number of negatives
};

#define MI_NEG_INF (-__builtin_inff())

// (m, s) <- (m, s) (+) x
__device__ __forceinline__ void lse_push(float& m, float& s, float x) {
  if (x > m) {
    s = s * expf(m - x) + 1.0f;  // m == -inf: s == 0 and exp(-inf) == 0
    m = x;
  } else {
    s += expf(x - m);
  }
}
// (m, s) <- (m, s) (+) (m2, s2); safe for empty sets (m == -inf, s == 0).  FAST: hardware exponential (v_exp_f32) for
// the bf16-operand paths, whose scores already carry ~1e-2 relative error.
template <bool FAST = false>
__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
  float mm = fmaxf(m, m2);
  if (mm == MI_NEG_INF) {
    m = mm;
    s = 0.0f;
    return;
  }
  if (FAST) s = s * __expf(m - mm) + s2 * __expf(m2 - mm);
  else s = s * expf(m - mm) + s2 * expf(m2 - mm);
  m = mm;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ unsigned wave_sum_u(unsigned v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
template <bool FAST = false>
__device__ __forceinline__ void wave_lse(float& m, float& s) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float m2 = __shfl_xor(m, o);
    float s2 = __shfl_xor(s, o);
    lse_merge<FAST>(m, s, m2, s2);
  }
}

// Reduce a Partial over a workgroup of NWAVES*64 threads; result valid in thread 0.  `scratch` holds
// NWAVES Partials.  All threads must call.
template <int NWAVES, bool FAST = false>
__device__ __forceinline__ Partial block_reduce_partial(Partial p, Partial* scratch) {
  wave_lse<FAST>(p.m, p.s);
  p.pos = wave_sum(p.pos);
  p.cnt = wave_sum_u(p.cnt);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) scratch[wave] = p;
  __syncthreads();
  if (threadIdx.x == 0) {
    Partial r = scratch[0];
    for (int w = 1; w < NWAVES; ++w) {
      lse_merge<FAST>(r.m, r.s, scratch[w].m, scratch[w].s);
      r.pos += scratch[w].pos;
      r.cnt += scratch[w].cnt;
    }
    p = r;
  }
  __syncthreads();
  return p;
}

// Touch every 64-byte line of the kernel-argument segment in ONE batch of independent scalar loads.  The segment is
// cold when a kernel starts and hipcc reads struct arguments lazily, field by field behind branches: the small-GEMM
// kernel went through 11 DEPENDENT scalar-load waits before its first memory operation, the fused bilinear kernel
// through 5 (~11,000 cycles per workgroup in the in-kernel stamps).  After this call those loads hit the scalar cache.
template <int BYTES>
__device__ __forceinline__ void kernarg_prefetch() {
  const __attribute__((address_space(4))) unsigned* p =
      (const __attribute__((address_space(4))) unsigned*)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned acc = 0;
#pragma unroll
  for (int o = 0; o < BYTES; o += 64) acc |= p[o / 4];
  asm volatile("" ::"s"(acc));
}

// d loss / d score of one pair given the global statistics.
// kind: 0 dropped, 1 positive (diagonal), 2 negative.
__device__ __forceinline__ int pair_kind(int64_t gi, int64_t gj, int64_t sid_i, int64_t sid_j) {
  if (gi == gj) return 1;
  return sid_i != sid_j ? 2 : 0;
}

// Raise a kernel's dynamic-LDS limit, once per (device, size): hipFuncSetAttribute is host-side state, and calling it on
// every launch also put such calls inside stream captures (the concat-MLP step could not be captured into a hipGraph).
// The cache is per launch site and per device and is safe to use from several host threads (autograd runs the backward
// on another thread than the forward): an atomic high-water mark per device, raised under a mutex.
struct DynSmemCache {
  static constexpr int kMaxDevices = 16;
  std::atomic<size_t> cur[kMaxDevices];
  std::mutex mu;
  DynSmemCache() {
    for (auto& c : cur) c.store(0, std::memory_order_relaxed);
  }
  // returns hipSuccess when the limit of `fn` on the current device is >= bytes afterwards
  hipError_t ensure(const void* fn, size_t bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (cur[dev].load(std::memory_order_acquire) >= bytes) return hipSuccess;
    std::lock_guard<std::mutex> lock(mu);
    if (cur[dev].load(std::memory_order_relaxed) >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) cur[dev].store(bytes, std::memory_order_release);
    return e;
  }
};
#define MI_SET_DYN_SMEM(fn, bytes, what)                                                  \
  do {                                                                                    \
    static ::mi::DynSmemCache mi_smem_cache_;                                             \
    hipError_t mi_e_ = mi_smem_cache_.ensure((const void*)(fn), (size_t)(bytes));         \
    if (mi_e_ != hipSuccess) return hip_fail(mi_e_, what);                                \
  } while (0)

// XCD affinity for 1-D grids (speed only, never correctness: guide T1 / section 6 G16).  Workgroups are dealt
// round-robin over the 8 XCDs in dispatch order, each XCD has a private L2: workgroup L runs on XCD L % 8 and is the
// (L / 8)-th workgroup of that XCD.  Work items that share operand panels get the same `outer` index and run back to
// back on one XCD:  inner = (L / 8) % n_inner,  outer = (L % 8) + 8 * ((L / 8) / n_inner).
// Launch 8 * n_inner * ceil(n_outer / 8) workgroups; returns false for the padding ones.
__device__ __forceinline__ bool xcd_decode(int n_inner, int n_outer, int& inner, int& outer, int natural = 0,
                                           int lin = -1) {
  // (lin >= 0: the linear id to decode instead of blockIdx.x -- a grid that holds several problems one after the other,
  // each starting at a multiple of 8)
  const int L = lin >= 0 ? lin : (int)blockIdx.x, e = L & 7, q = L >> 3;
  if (natural) {  // A/B switch: plain row-major order
    inner = L % n_inner;
    outer = L / n_inner;
    return outer < n_outer;
  }
  inner = q % n_inner;
  outer = e + 8 * (q / n_inner);
  return outer < n_outer;
}
// MI_XCD_NATURAL=1 in the environment selects the natural order in every kernel that uses xcd_decode (A/B runs)
int xcd_natural();
static inline unsigned xcd_grid(int64_t n_inner, int64_t n_outer) { return (unsigned)(8 * n_inner * ((n_outer + 7) / 8)); }

// launchers shared between translation units --------------------------------------------------------
// partials[n_partials] -> stats, loss_out (one workgroup).  local_record (optional, 8 floats) receives the
// merged (m, s, pos, cnt_lo, cnt_hi) for the cross-rank merge.
int launch_finalize(const Partial* partials, int64_t n_partials, int64_t n_pos, int estimator, float* loss_out,
                    mi_stats* stats, float* local_record, hipStream_t stream);

}  // namespace mi
