// Ablation (not part of the product): sustained rate of v_mfma_f32_16x16x32_bf16 against v_mfma_f32_32x32x16_bf16 on
// operands with random mantissas (switching power), whole chip, WAVES waves per SIMD, ~50 ms per measurement so the
// clock settles.  Accumulator chains: 8 independent for 16x16 (32 AGPRs), 2 for 32x32 (32 AGPRs).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ inline unsigned rnd(unsigned &s) { s = s * 1664525u + 1013904223u; return s; }
__device__ inline u32x4 operand(unsigned &s, int zero) {
  u32x4 v;
  for (int i = 0; i < 4; ++i) {
    unsigned r = rnd(s);
    // two bf16 of magnitude ~1 with random sign / mantissa: exponent 0x3f8 >> ..., keep it simple: 0x3f80 | 7 random bits
    unsigned lo = 0x3f80u | (r & 0x7f) | ((r >> 7 & 1) << 15), hi = 0x3f80u | (r >> 8 & 0x7f) | ((r >> 15 & 1) << 15);
    v[i] = zero ? 0u : (lo | hi << 16);
  }
  return v;
}

template <int SHAPE>   // 0: 16x16x32, 1: 32x32x16
__global__ __launch_bounds__(512) void k(float *out, int iters, int zero) {
  unsigned s = blockIdx.x * 512 + threadIdx.x + 1;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = __builtin_bit_cast(bf16x8, operand(s, zero)); b[i] = __builtin_bit_cast(bf16x8, operand(s, zero)); }
  float sum = 0.f;
  if (SHAPE == 0) {
    f32x4 acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(c + r) & 3], b[c & 3], acc[c], 0, 0, 0);
    }
    for (int c = 0; c < 8; ++c) sum += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  } else {
    f32x16 acc[2];
    for (int c = 0; c < 2; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(c + r) & 3], b[(r >> 1) & 3], acc[c], 0, 0, 0);
    }
    for (int c = 0; c < 2; ++c) for (int i = 0; i < 16; ++i) sum += acc[c][i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = sum;
}

template <int SHAPE> void run(int waves, int zero) {
  const int threads = 256 * waves, blocks = 256;   // one workgroup per CU, `waves` waves per SIMD
  const int per_iter = SHAPE == 0 ? 32 : 16;       // MFMAs per iteration per wave
  const double flop_per = SHAPE == 0 ? 16384.0 : 32768.0;
  int iters = 400000 / waves;
  float *out; hipMalloc(&out, (size_t)blocks * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(threads), 0, 0, out, iters, zero);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(threads), 0, 0, out, iters, zero);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)blocks * threads / 64 * iters * per_iter;
  const double cyc = SHAPE == 0 ? 16.0 : 32.0;     // pipe cycles per instruction at the 2.5 PF peak
  printf("%s  %d wave/SIMD  %s operands  %.1f ms  %.0f TFLOP/s  -> %.2f GHz effective\n", SHAPE == 0 ? "16x16x32" : "32x32x16", waves,
         zero ? "zero  " : "random", ms, mfma * flop_per / ms / 1e9, mfma / 1024.0 * cyc / (ms * 1e-3) / 1e9);
  hipFree(out);
}
int main() {
  for (int zero = 0; zero < 2; ++zero)
    for (int waves = 1; waves <= 2; ++waves) { run<0>(waves, zero); run<1>(waves, zero); }
  return 0;
}
