// Shared host/device helpers for libnsdp_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/nsdp_hip.h"

namespace nsdp {

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Returns a C-ABI status for the most recent launch on this thread.
inline int launch_status(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return static_cast<int>(e);
  }
  return 0;
}

#define NSDP_REQUIRE(cond, ...)     \
  do {                              \
    if (!(cond)) {                  \
      nsdp::set_error(__VA_ARGS__); \
      return NSDP_EINVAL;           \
    }                               \
  } while (0)

#define NSDP_HIP_TRY(expr)                                                \
  do {                                                                    \
    hipError_t e__ = (expr);                                              \
    if (e__ != hipSuccess) {                                              \
      nsdp::set_error("%s failed: %s", #expr, hipGetErrorString(e__));    \
      return static_cast<int>(e__);                                       \
    }                                                                     \
  } while (0)

// Exactly-rounded fp32 primitives: the geometry kernels must reproduce
// ((dx*dx + dy*dy) + dz*dz) with one rounding per operation (no FMA contraction).
__device__ __forceinline__ float sq_dist3(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// compute units of the current device (cached per device)
inline int num_cus() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

inline int ceil_div(long long a, long long b) { return static_cast<int>((a + b - 1) / b); }

}  // namespace nsdp
