// Ablation (not part of the product): scatter-add of rows into a small per-shape table held in REGISTERS
// (lane = channel, dynamic register indexing with s_set_gpr_idx) instead of LDS float atomics.
//   out[b][idx[r]][c] -= du[r][c]   for the rows r of shape b;  N <= 128 anchors, d channels
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x32 __attribute__((ext_vector_type(32)));

template <int UNROLL>
__global__ __launch_bounds__(256) void k(const float *__restrict__ du, const int *__restrict__ idx, float *__restrict__ out,
                                         int rows_per_shape, int rows_per_wg, int N, int d) {
  f32x32 t0 = {}, t1 = {}, t2 = {}, t3 = {};
  const int c = threadIdx.x;
  const int b = blockIdx.y;
  const long long r0 = (long long)b * rows_per_shape + (long long)blockIdx.x * rows_per_wg;
  const bool cv = c < d;
  const float *p = du + r0 * d + (cv ? c : 0);
  const int *ip = idx + r0;
  for (int r = 0; r < rows_per_wg; r += UNROLL) {
    float x[UNROLL];
    int a[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      x[u] = p[(long long)(r + u) * d];
      a[u] = __builtin_amdgcn_readfirstlane(ip[r + u]);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int i = a[u] & 31, g = a[u] >> 5;
      const float v = cv ? x[u] : 0.f;
      t0[i] -= g == 0 ? v : 0.f;
      t1[i] -= g == 1 ? v : 0.f;
      t2[i] -= g == 2 ? v : 0.f;
      t3[i] -= g == 3 ? v : 0.f;
    }
  }
  if (cv) {
    float *o = out + (long long)b * N * d + c;
    for (int i = 0; i < 32; ++i) {
      if (i < N) atomicAdd(o + (long long)i * d, t0[i]);
      if (32 + i < N) atomicAdd(o + (long long)(32 + i) * d, t1[i]);
      if (64 + i < N) atomicAdd(o + (long long)(64 + i) * d, t2[i]);
      if (96 + i < N) atomicAdd(o + (long long)(96 + i) * d, t3[i]);
    }
  }
}

int main() {
  const int B = 32, rows_per_shape = 8192 * 7, N = 100, d = 200;
  const long long R = (long long)B * rows_per_shape;
  float *du, *out; int *idx;
  hipMalloc(&du, R * d * 4); hipMalloc(&out, (size_t)B * N * d * 4); hipMalloc(&idx, R * 4);
  std::vector<int> h(R);
  for (long long i = 0; i < R; ++i) h[i] = (int)((i * 2654435761u) % N);
  hipMemcpy(idx, h.data(), R * 4, hipMemcpyHostToDevice);
  hipMemset(du, 0x3f, R * d * 4);   // non-zero values: the flush atomics are real
  for (int wgs_per_shape : {8, 16, 32, 64}) {
    const int rows_per_wg = rows_per_shape / wgs_per_shape;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<8>), dim3(wgs_per_shape, B), dim3(256), 0, 0, du, idx, out, rows_per_shape, rows_per_wg, N, d);
    hipEventRecord(e0);
    for (int it = 0; it < 5; ++it)
      hipLaunchKernelGGL((k<8>), dim3(wgs_per_shape, B), dim3(256), 0, 0, du, idx, out, rows_per_shape, rows_per_wg, N, d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("wgs/shape %d: %.3f ms  (%.2f TB/s of du)\n", wgs_per_shape, ms, R * d * 4.0 / ms / 1e9);
  }
  return 0;
}
