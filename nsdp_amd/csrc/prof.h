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
  kLinearB16,    // linear_bf16_kernel (bf16 storage, one bf16 MFMA product)
  kWgradB16,     // wgrad_bf16_kernel + its reduce
  kNumKinds
};

const char *kind_name(int kind);

struct Scope {
  Scope(Kind kind, hipStream_t stream, double flops, double bytes);
  ~Scope();
  int slot_;
  hipStream_t stream_;
};

// Kernel-variant trace (tests): when enabled, every launcher notes which template instance it picked, so a parity test
// can assert that the code path it means to check (8-wave bf16x3 GEMM, LDS-table attention backward, ...) really ran.
// Disabled by default: one relaxed atomic load per launch.
bool trace_on();
void trace_note(const char *fmt, ...);
#define NSDP_TRACE(...)                                         \
  do {                                                          \
    if (nsdp::prof::trace_on()) nsdp::prof::trace_note(__VA_ARGS__); \
  } while (0)

}  // namespace prof
}  // namespace nsdp
