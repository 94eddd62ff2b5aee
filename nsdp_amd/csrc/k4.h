// Pre-activation of a K = 4 dense layer (the first layer of every position-encoding MLP: 3-d relative coordinates,
// zero-padded), shared by every kernel that has to reproduce the layer's output or its ReLU mask bit for bit: the forward stream and
// the weight-gradient kernel that recomputes the mask from the 16-byte input rows (gemm.hip), the dX GEMM whose epilogue does the same
// and reduces the layer's weight gradient on the spot (gemm_bf16x3.hip, TAIL forms), and the operand producers that recompute the
// layer's OUTPUT instead of reading it (H0 forms of gemm_bf16x3.hip and wgrad_bf16x3.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace nsdp {
// b + (x.x w.x + x.y w.y + x.z w.z + x.w w.w) with its rounding pinned: explicit fused multiply-adds, nothing left to the
// compiler's contraction.  Until round 5 it was left to it (-ffp-contract=fast), and the packed-math forward kernel evaluated
// x.x w.x + x.y w.y as fma(x.x, w.x, fl(x.y w.y)) for even output channels and as fma(x.y, w.y, fl(x.x w.x)) for odd ones (the two
// halves of one v_pk_fma_f32 with crossed operand selects) -- equally good roundings; the per-parity choice is kept so that every
// value is the one the golden fixtures were recorded against.  n: the output channel.
__device__ __forceinline__ float k4_preact_n(float4 xv, float4 w, float b, int n) {
  const float t = (n & 1) ? __builtin_fmaf(xv.y, w.y, xv.x * w.x) : __builtin_fmaf(xv.x, w.x, xv.y * w.y);
  return b + __builtin_fmaf(xv.w, w.w, __builtin_fmaf(xv.z, w.z, t));
}
// Two adjacent output channels (n even, n + 1) of the same pre-activation on the packed fp32 pipe (v_pk_fma_f32: two lanes of
// arithmetic per instruction), bit for bit k4_preact_n(.., n) and k4_preact_n(.., n + 1): the crossed products of the two parities
// become straight packed operands when the layer's table is stored pair-wise,
//   p0 = (w_n.x, w_n1.y), p1 = (w_n.y, w_n1.x), p2 = (w_n.z, w_n1.z), p3 = (w_n.w, w_n1.w), pb = (b_n, b_n1)      (k4_pair_table)
typedef __attribute__((ext_vector_type(2))) float k4_f2;
typedef __attribute__((ext_vector_type(4))) float k4_f4;
__device__ __forceinline__ k4_f2 k4_preact_pair(float4 xv, k4_f2 p0, k4_f2 p1, k4_f2 p2, k4_f2 p3, k4_f2 pb) {
  const k4_f2 xy = {xv.x, xv.y}, yx = {xv.y, xv.x}, zz = {xv.z, xv.z}, ww = {xv.w, xv.w};
  k4_f2 t = __builtin_elementwise_fma(xy, p0, yx * p1);
  t = __builtin_elementwise_fma(zz, p2, t);
  t = __builtin_elementwise_fma(ww, p3, t);
  return pb + t;
}
// rows n = 2 j, 2 j + 1 of a zero-padded [K, 4] weight (rows >= K: zeros) as the two float4 of pair j: (p0, p1), (p2, p3)
__device__ __forceinline__ void k4_pair_table(const float *__restrict__ w4, int K, int j, k4_f4 &lo, k4_f4 &hi) {
  const k4_f4 z = {0.f, 0.f, 0.f, 0.f};
  const k4_f4 e = 2 * j < K ? *reinterpret_cast<const k4_f4 *>(w4 + static_cast<long long>(2 * j) * 4) : z;
  const k4_f4 o = 2 * j + 1 < K ? *reinterpret_cast<const k4_f4 *>(w4 + static_cast<long long>(2 * j + 1) * 4) : z;
  lo = k4_f4{e[0], o[1], e[1], o[0]};
  hi = k4_f4{e[2], o[2], e[3], o[3]};
}
}  // namespace nsdp
