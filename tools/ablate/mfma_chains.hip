// Ablation (not part of the product): fp32 16x16x4 MFMA issue rate as a function of the number of independent
// accumulator chains per wave, one wave per SIMD (the fused decoder's situation).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int CH>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
  __shared__ float pad[25000];   // 100 KB: one workgroup (= one wave per SIMD) per CU
  if (iters < 0) pad[threadIdx.x] = a0;
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f32x4{0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
      if (CH == 1) asm volatile("" : "+a"(acc[0]));
      if (CH == 2) asm volatile("" : "+a"(acc[0]), "+a"(acc[1]));
      if (CH == 3) asm volatile("" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]));
      if (CH == 4) asm volatile("" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));
    }
  }
  float s = iters < 0 ? pad[(threadIdx.x * 7) % 25000] : 0.f;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CH> void run() {
  const int blocks = 256 * 4, iters = 4000 / CH;
  float *out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)blocks * 4 * iters * 16 * CH;
  printf("chains=%d  %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz, 1 wave/SIMD)\n", CH, ms, mfma * 2048 / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (mfma / 1024.0));
  hipFree(out);
}
int main() { run<1>(); run<2>(); run<3>(); run<4>(); return 0; }
