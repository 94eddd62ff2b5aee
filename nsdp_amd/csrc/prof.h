// Optional per-kernel timing with HIP events recorded on the launch stream (bench.py roofline).
// Disabled by default: a Scope then costs one relaxed atomic load.
#pragma once
#include <hip/hip_runtime.h>

namespace nsdp {
namespace prof {

enum Kind {
  kLinear = 0,   // linear_nt_kernel   (fp32 MFMA, Y = X W^T)
  kWgrad,        // linear_wgrad_kernel (fp32 MFMA, dW = dY^T X)
  kFps,          // fps_*_kernel
  kKnn,          // knn_kernel
  kGatherRows,   // gather_rows_kernel
  kScatterRows,  // scatter_add_rows_kernel
  kAttnFwd,      // fused attention elementwise forward kernels
  kAttnBwd,      // fused attention elementwise backward kernels
  kBatchNorm,    // batch-norm kernels
  kDecoderFwd,   // fused decoder forward
  kLinearX3,     // linear_bf16x3_kernel (bf16 MFMA, 3-way split, 6 products per fp32 product)
  kWgradX3,      // wgrad_bf16x3_kernel + its reduce
  kNumKinds
};

const char *kind_name(int kind);

struct Scope {
  Scope(Kind kind, hipStream_t stream, double flops, double bytes);
  ~Scope();
  int slot_;
  hipStream_t stream_;
};

}  // namespace prof
}  // namespace nsdp
