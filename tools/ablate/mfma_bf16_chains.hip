// Ablation (not part of the product): v_mfma_f32_16x16x32_bf16 issue rate vs. number of independent accumulator
// chains, one wave per SIMD; and the cost of the fp32 -> 3 x bf16 truncation split on the VALU next to it.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int CH, int DEP>   // DEP dependent MFMAs per chain back to back before switching chain
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0) {
  __shared__ float pad[25000];
  if (iters < 0) pad[threadIdx.x] = a0;
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f32x4{0, 0, 0, 0};
  u32x4 au = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  bf16x8 a = __builtin_bit_cast(bf16x8, au), b = a;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
#pragma unroll
        for (int d = 0; d < DEP; ++d) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
      }
      if (CH == 1) asm volatile("" : "+a"(acc[0]));
      if (CH == 2) asm volatile("" : "+a"(acc[0]), "+a"(acc[1]));
      if (CH == 4) asm volatile("" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));
    }
  }
  float s = iters < 0 ? pad[(threadIdx.x * 7) % 25000] : 0.f;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CH, int DEP> void run() {
  const int blocks = 256 * 4, iters = 16000 / (CH * DEP);
  float *out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CH, DEP>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<CH, DEP>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)blocks * 4 * iters * 8 * CH * DEP;
  printf("chains=%d dep=%d  %.3f ms  %.0f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", CH, DEP, ms,
         mfma * 16384 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mfma / 1024.0));
  hipFree(out);
}
int main() { run<1, 1>(); run<2, 1>(); run<4, 1>(); run<2, 3>(); run<2, 6>(); run<4, 6>(); return 0; }
