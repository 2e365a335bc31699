// microbenchmark of the GEMM K-loop skeleton (diagnostic): which part costs what
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MASK bits: 1 = LDS-DMA, 2 = ds_reads, 4 = MFMAs, 8 = barrier
template <int MASK, int NWAVES, int SHARED>
__global__ __launch_bounds__(NWAVES * 64, 1) void k(const char* a, unsigned long long* out, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int PIECES = 32 / NWAVES;
  const char* src[PIECES];
  for (int i = 0; i < PIECES; ++i) {
    const int q = PIECES * wave + i;
    const int row = q * 8 + (lane >> 3);
    src[i] = a + ((size_t)(SHARED ? (blockIdx.x & 7) : blockIdx.x) * 256 + row) * 8192 + (size_t)((lane & 7) ^ ((row >> 1) & 7)) * 16;
  }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 fr[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 8; ++r) fr[i][r] = (__bf16)0.0f;
  const int rrow = (wave & 3) * 32 + (lane & 31);
  const int roff = rrow * 128;
  const int rswz = (rrow >> 1) & 7;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  int stage = 0;
  for (int t = 0; t < iters; ++t) {
    if (MASK & 8) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PIECES) : "memory");
    if (MASK & 1) {
#pragma unroll
      for (int i = 0; i < PIECES; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)(t & 63) * 128),
                                         (__attribute__((address_space(3))) void*)(smem + stage * 32768 + (PIECES * wave + i) * 1024), 16, 0, 0);
    }
    if (MASK & 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        fr[i] = *reinterpret_cast<const bf16x8*>(smem + stage * 32768 + (i >> 2) * 16384 + roff + (((2 * (i & 3) + (lane >> 5)) ^ rswz) * 16));
    }
    if (MASK & 4) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i], fr[(i + 1) & 7], acc[i & 3], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(fr[i]));
    }
    stage = stage == 2 ? 0 : stage + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; out[256 + blockIdx.x] = r1 - r0; }
  float s = 0;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.f) sink[0] = s;
}

template <int MASK, int NWAVES, int SHARED>
void run(const char* a, unsigned long long* d, float* sink, const char* name) {
  hipFuncSetAttribute((const void*)k<MASK, NWAVES, SHARED>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  const int iters = 4096;
  unsigned long long h[512];
  for (int rep = 0; rep < 6; ++rep) {
    hipLaunchKernelGGL((k<MASK, NWAVES, SHARED>), dim3(256), dim3(NWAVES * 64), 98304, 0, a, d, iters, sink);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0, r = 0;
  for (int i = 0; i < 256; ++i) { s += h[i]; r += h[256 + i]; }
  printf("%-44s shared %d waves %d: %7.0f memtime ticks / iteration, %.3f us / iteration, memtime rate %.0f MHz\n", name, SHARED, NWAVES, s / 256 / iters, r / 256 / iters / 100.0, 100.0 * s / r);
}

int main() {
  char* a; hipMalloc(&a, (size_t)256 * 256 * 8192);
  hipMemset(a, 0, (size_t)256 * 256 * 8192);
  if (getenv("RANDOM_DATA")) {
    size_t n = (size_t)256 * 256 * 8192 / 2;
    unsigned short* hbuf = (unsigned short*)malloc(n * 2);
    unsigned x = 12345;
    for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; hbuf[i] = (unsigned short)(0x3c00 + ((x >> 16) & 0x3ff) + ((x >> 31) << 15)); }
    hipMemcpy(a, hbuf, n * 2, hipMemcpyHostToDevice);
    free(hbuf);
    printf("random bf16 data\n");
  }
  unsigned long long* d; hipMalloc(&d, 512 * 8);
  float* sink; hipMalloc(&sink, 4);
  run<8 | 4, 8, 1>(a, d, sink, "barrier + MFMA");
  run<8 | 2 | 4, 8, 1>(a, d, sink, "barrier + reads + MFMA");
  run<8 | 1, 8, 1>(a, d, sink, "barrier + 4 DMA pieces/wave (32 KB/iter)");
  run<15, 8, 1>(a, d, sink, "all");
  return 0;
}
