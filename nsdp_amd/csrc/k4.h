// Pre-activation of a K = 4 dense layer (the first layer of every position-encoding MLP: 3-d relative coordinates,
// zero-padded).  ONE expression (one inline function, contracted by the compiler the same way in every translation unit built
// with the same flags: nsdp_amd/build.py) for every kernel that needs the layer's ReLU mask to be bit for bit the forward kernel's: the forward stream (gemm.hip), the weight-gradient kernel that recomputes the mask from
// the 16-byte input rows (gemm.hip), and the dX GEMM whose epilogue does the same and reduces the layer's weight gradient
// on the spot (gemm_bf16x3.hip, TAIL forms).
#pragma once
#include <hip/hip_runtime.h>

namespace nsdp {
__device__ __forceinline__ float k4_preact(float4 xv, float4 w, float b) {
  return b + (xv.x * w.x + xv.y * w.y + xv.z * w.z + xv.w * w.w);
}
}  // namespace nsdp
