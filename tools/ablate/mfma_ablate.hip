// Ablation of the fp32 MFMA inner loop (not part of the product): what limits linear_nt_kernel<2,13>?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int VARIANT>
__global__ __launch_bounds__(256) void k(const float* __restrict__ X, const float* __restrict__ W, float* out, int iters, int K) {
  constexpr int MT = 2, NT = 13;
  const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
  f32x4 acc[MT][NT];
  for (int a = 0; a < MT; ++a) for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
  float4 av[MT], bv[NT];
  for (int a = 0; a < MT; ++a) av[a] = make_float4(lane, 1, 2, 3);
  for (int b = 0; b < NT; ++b) bv[b] = make_float4(b, lane, 2, 3);
  const float* xp = X + (size_t)((blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + li) * K + 4 * g;
  const float* wp = W + (size_t)li * K + 4 * g;
  for (int it = 0; it < iters; ++it) {
    if (VARIANT >= 2) {
      const int ko = (it % (K / 16)) * 16;
      for (int a = 0; a < MT; ++a) av[a] = *reinterpret_cast<const float4*>(xp + (size_t)a * 16 * K + ko);
      for (int b = 0; b < NT; ++b) bv[b] = *reinterpret_cast<const float4*>(wp + (size_t)b * 16 * K + ko);
    }
    if (VARIANT == 1 || VARIANT == 3) {
      const bool kv = (it + g) < iters + 8;
      for (int b = 0; b < NT; ++b) { bv[b].x = kv ? bv[b].x : 0.f; bv[b].y = kv ? bv[b].y : 0.f; bv[b].z = kv ? bv[b].z : 0.f; bv[b].w = kv ? bv[b].w : 0.f; }
    }
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a].x, bv[b].x, acc[a][b], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a].y, bv[b].y, acc[a][b], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a].z, bv[b].z, acc[a][b], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a].w, bv[b].w, acc[a][b], 0, 0, 0);
  }
  float s = 0;
  for (int a = 0; a < MT; ++a) for (int b = 0; b < NT; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V> void run(const char* name, int blocks) {
  const int K = 208, iters = 13 * 40;
  float *X, *W, *out;
  hipMalloc(&X, (size_t)blocks * 128 * K * 4); hipMalloc(&W, 208 * K * 4); hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipMemset(X, 0, (size_t)blocks * 128 * K * 4); hipMemset(W, 0, 208 * K * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, X, W, out, iters, K);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, X, W, out, iters, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double flops = (double)blocks * 4 * iters * 104.0 * 2048.0;
  printf("%-28s blocks %5d: %.3f ms  %.1f TF\n", name, blocks, ms, flops / ms / 1e9);
  hipFree(X); hipFree(W); hipFree(out);
}
int main() {
  for (int blocks : {256, 512, 2048}) {
    run<0>("mfma only", blocks);
    run<1>("mfma + cndmask", blocks);
    run<2>("mfma + loads", blocks);
    run<3>("mfma + loads + cndmask", blocks);
  }
  return 0;
}
