// Per-thread bodies of the two weight-pack kernels (nsdp_pack_weight_f32, nsdp_pack_weight_bf16x3), shared with the
// batched launcher (pack_batched.hip): thread q of the launch writes float4 / (3 x uint4) number q of each output.
#pragma once
#include <hip/hip_runtime.h>

namespace nsdp {
namespace pack {

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;

// threads a launch needs (a multiple of 64)
__host__ __device__ inline long long fp32_threads(int N, int K) {
  return static_cast<long long>((N + 15) >> 4) * ((K + 15) >> 4) * 64;
}
__host__ __device__ inline long long x3_threads(int N, int K, bool fwd, bool transposed) {
  const long long b0 = fwd ? static_cast<long long>((K + 31) >> 5) * ((N + 15) >> 4) : 0;
  const long long b1 = transposed ? static_cast<long long>((N + 31) >> 5) * ((K + 15) >> 4) : 0;
  return (b0 > b1 ? b0 : b1) * 64;
}

// fragment-major fp32 packs (layout: include/nsdp_hip.h, nsdp_pack_weight_f32)
__device__ __forceinline__ void fp32_body(const float *__restrict__ W, int N, int K, float *__restrict__ Wp,
                                          float *__restrict__ WpT, long long q) {
  const int NB = (N + 15) >> 4, KB = (K + 15) >> 4;
  if (q >= static_cast<long long>(NB) * KB * 64) return;
  const int lane = static_cast<int>(q & 63), li = lane & 15, g = lane >> 4;
  const long long blk = q >> 6;
  if (Wp) {
    const int tn = static_cast<int>(blk / KB), kb = static_cast<int>(blk % KB);
    const int n = tn * 16 + li, k0 = kb * 16 + 4 * g;
    float4 v;
    v.x = n < N && k0 + 0 < K ? W[static_cast<long long>(n) * K + k0 + 0] : 0.f;
    v.y = n < N && k0 + 1 < K ? W[static_cast<long long>(n) * K + k0 + 1] : 0.f;
    v.z = n < N && k0 + 2 < K ? W[static_cast<long long>(n) * K + k0 + 2] : 0.f;
    v.w = n < N && k0 + 3 < K ? W[static_cast<long long>(n) * K + k0 + 3] : 0.f;
    reinterpret_cast<float4 *>(Wp)[q] = v;
  }
  if (WpT) {
    const int tk = static_cast<int>(blk / NB), nb = static_cast<int>(blk % NB);
    const int k = tk * 16 + li, n0 = nb * 16 + 4 * g;
    float4 v;
    v.x = k < K && n0 + 0 < N ? W[static_cast<long long>(n0 + 0) * K + k] : 0.f;
    v.y = k < K && n0 + 1 < N ? W[static_cast<long long>(n0 + 1) * K + k] : 0.f;
    v.z = k < K && n0 + 2 < N ? W[static_cast<long long>(n0 + 2) * K + k] : 0.f;
    v.w = k < K && n0 + 3 < N ? W[static_cast<long long>(n0 + 3) * K + k] : 0.f;
    reinterpret_cast<float4 *>(WpT)[q] = v;
  }
}

// two fp32 values -> the packed (lo, hi) bf16 pairs of their three split planes (round to nearest, exact residuals)
__device__ __forceinline__ void split3(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
  const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
  const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
}

// bf16x3 plane packs (layout: include/nsdp_hip.h, nsdp_pack_weight_bf16x3; kperm(g, j) = 16 (j / 4) + 4 g + j % 4)
__device__ __forceinline__ void x3_body(const float *__restrict__ W, int N, int K, u32x4 *__restrict__ Wp,
                                        u32x4 *__restrict__ WpT, long long q) {
  const int lane = static_cast<int>(q & 63), li = lane & 15, g = lane >> 4;
  const long long blk = q >> 6;
  if (Wp) {
    const int NT = (N + 15) >> 4, KB = (K + 31) >> 5;
    if (blk < static_cast<long long>(NT) * KB) {
      const int kb = static_cast<int>(blk / NT), tn = static_cast<int>(blk % NT);
      const int n = tn * 16 + li, k0 = kb * 32 + 4 * g;
      u32x4 h, m, l;
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        const int k = k0 + 16 * (pr >> 1) + 2 * (pr & 1);
        const float x0 = n < N && k < K ? W[static_cast<long long>(n) * K + k] : 0.f;
        const float x1 = n < N && k + 1 < K ? W[static_cast<long long>(n) * K + k + 1] : 0.f;
        unsigned a, b, c;
        split3(x0, x1, a, b, c);
        h[pr] = a; m[pr] = b; l[pr] = c;
      }
      u32x4 *dst = Wp + (blk * 3) * 64 + lane;
      dst[0] = h; dst[64] = m; dst[128] = l;
    }
  }
  if (WpT) {
    const int KT = (K + 15) >> 4, NB = (N + 31) >> 5;
    if (blk < static_cast<long long>(KT) * NB) {
      const int nb = static_cast<int>(blk / KT), tk = static_cast<int>(blk % KT);
      const int k = tk * 16 + li, n0 = nb * 32 + 4 * g;
      u32x4 h, m, l;
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        const int n = n0 + 16 * (pr >> 1) + 2 * (pr & 1);
        const float x0 = k < K && n < N ? W[static_cast<long long>(n) * K + k] : 0.f;
        const float x1 = k < K && n + 1 < N ? W[static_cast<long long>(n + 1) * K + k] : 0.f;
        unsigned a, b, c;
        split3(x0, x1, a, b, c);
        h[pr] = a; m[pr] = b; l[pr] = c;
      }
      u32x4 *dst = WpT + (blk * 3) * 64 + lane;
      dst[0] = h; dst[64] = m; dst[128] = l;
    }
  }
}

}  // namespace pack
}  // namespace nsdp
