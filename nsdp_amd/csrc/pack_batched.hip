// nsdp_pack_weights_batched: every weight pack of a model in one or two launches.
//
// The optimizer rewrites all weights once per train step, so every layer's pack (fragment-major fp32 or bf16x3 planes,
// forward and / or transposed) is rebuilt once per step: ~116 launches of 4 us each, and -- what matters for small
// batches -- ~30 us of host time apiece.  Here the descriptors travel by value in the kernel arguments (64 per launch,
// 2.5 KiB), grid.y selects the descriptor and grid.x covers the largest pack of the launch.
#include "common.h"
#include "pack_bodies.h"
#include "../../include/nsdp_hip.h"

namespace {

constexpr int kBatch = 64;

struct Batch {
  NsdpPackDesc d[kBatch];
};

__global__ __launch_bounds__(256) void pack_batched_kernel(Batch b) {
  const NsdpPackDesc &e = b.d[blockIdx.y];
  const long long q = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (e.kind == 0) {
    nsdp::pack::fp32_body(e.W, e.N, e.K, static_cast<float *>(e.Wp), static_cast<float *>(e.WpT), q);
  } else if (e.kind == 3) {      // row-major [N, 4] copy of a K <= 4 weight, zero-padded (the K = 4 kernels' 16-byte weight rows)
    if (q < 4LL * e.N) static_cast<float *>(e.Wp)[q] = static_cast<int>(q & 3) < e.K ? e.W[(q >> 2) * e.K + (q & 3)] : 0.f;
  } else {
    if (q < nsdp::pack::x3_threads(e.N, e.K, e.Wp != nullptr, e.WpT != nullptr))
      nsdp::pack::x3_body(e.W, e.N, e.K, static_cast<nsdp::pack::u32x4 *>(e.Wp), static_cast<nsdp::pack::u32x4 *>(e.WpT), q);
  }
}

}  // namespace

extern "C" int nsdp_pack_weights_batched(const NsdpPackDesc *descs, int count, void *stream) {
  if (count <= 0) return 0;
  NSDP_REQUIRE(descs, "pack_weights_batched: null descriptor array");
  hipStream_t st = nsdp::as_stream(stream);
  for (int base = 0; base < count; base += kBatch) {
    Batch b;
    const int n = count - base < kBatch ? count - base : kBatch;
    long long threads = 64;
    for (int i = 0; i < n; ++i) {
      const NsdpPackDesc &e = descs[base + i];
      NSDP_REQUIRE(e.W && (e.Wp || e.WpT) && e.N > 0 && e.K > 0 && (e.kind == 0 || e.kind == 1 || (e.kind == 3 && e.Wp && e.K <= 4)),
                   "pack_weights_batched: bad descriptor %d", base + i);
      b.d[i] = e;
      const long long t = e.kind == 0   ? nsdp::pack::fp32_threads(e.N, e.K)
                          : e.kind == 3 ? 4LL * e.N
                                        : nsdp::pack::x3_threads(e.N, e.K, e.Wp != nullptr, e.WpT != nullptr);
      threads = t > threads ? t : threads;
    }
    for (int i = n; i < kBatch; ++i) b.d[i] = b.d[0];     // never indexed (grid.y = n)
    hipLaunchKernelGGL(pack_batched_kernel, dim3(static_cast<unsigned>((threads + 255) / 256), n), dim3(256), 0, st, b);
    const int rc = nsdp::launch_status("pack_batched_kernel");
    if (rc) return rc;
  }
  return 0;
}
