// nsdp_adam_multi_f32: the Adam update of EVERY parameter tensor of a model in one launch.
//
// The reference steps `torch.optim.Adam` once per train step (/root/reference/model/__init__.py:10-41 builds it,
// model/deformation_networks.py:63-77 calls optimizer.step() after loss.backward()).  For the ~300 parameter tensors of a
// TDNet PyTorch's own multi-tensor kernels need 8 launches of 40 us each (36 tensors per launch travel in the kernel
// arguments; 126 MB of traffic at 0.4 TB/s) at the very tail of the step, where nothing else is left to run beside them.
// Here the tensor table lives in device memory (it is static across the replays of a captured step): one workgroup per
// 4096-element chunk of one tensor, all chunks of all tensors in one grid; 4 reads + 3 writes of 4 bytes per element,
// HBM-bound.
//
// Arithmetic = torch's single-tensor Adam on fp32 tensors, operation by operation with one rounding each (this file is
// compiled with contraction off; fp32 division and sqrt are correctly rounded):
//   g' = g + wd * p                      (weight_decay != 0;  -g first when maximize)
//   m  = m + (1 - beta1) * (g' - m)      (Tensor.lerp_, weight < 0.5)
//   v  = v * beta2 + ((1 - beta2) * g') * g'
//   p  = p + ((-lr / (1 - beta1^t)) * m) / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// with the scalars formed in double and rounded to fp32 where torch hands them to an fp32 tensor op.
// WHICH torch: the CPU, non-capturable, single-tensor path -- torch.optim.Adam(foreach=False) on CPU tensors -- is the sequence
// restated here and what tests/test_adam_gpu.py compares with (moments to a few ulp of the largest term, parameters to one).
// torch's CUDA / ROCm kernels (foreach, fused, capturable: bias corrections in fp32, addcdiv as a + alpha * (m / denom)) may
// differ from it -- and hence from this kernel -- in the last place: nobody should rely on bit-equality with a GPU torch run.
// The step counter t is one fp32 scalar per tensor in device memory (torch's `capturable` layout: state["step"]); every
// chunk reads it, the LAST chunk of a tensor to finish stores t + 1 (a per-tensor arrival counter that the same workgroup
// resets) -- no second launch, and the value never depends on the order in which the chunks ran.
#include <math.h>

#include "common.h"

namespace {

constexpr int kChunk = 4096;      // elements per workgroup: 256 lanes x 4 float4

struct Scalars {
  float neg_step_size, bc2_sqrt, w1, beta2, w2, eps, wd;
  int maximize;
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, const Scalars &s) {
  if (s.maximize) g = -g;
  if (s.wd != 0.f) g = g + s.wd * p;
  m = m + s.w1 * (g - m);
  v = v * s.beta2 + (s.w2 * g) * g;
  const float denom = sqrtf(v) / s.bc2_sqrt + s.eps;
  p = p + (s.neg_step_size * m) / denom;
}

__global__ __launch_bounds__(256) void adam_multi_kernel(const NsdpAdamDesc *__restrict__ descs,
                                                         const int2 *__restrict__ chunks, int *__restrict__ done,
                                                         const float *__restrict__ lr_dev, double lr_host, double beta1,
                                                         double beta2, double eps, double weight_decay, int maximize) {
  const int2 c = chunks[blockIdx.x];      // (tensor, chunk of that tensor)
  const NsdpAdamDesc d = descs[c.x];
  const float step_old = *d.step;
  const double t = static_cast<double>(step_old) + 1.0;
  const double lr = lr_dev ? static_cast<double>(*lr_dev) : lr_host;
  const double bc1 = 1.0 - pow(beta1, t), bc2 = 1.0 - pow(beta2, t);
  Scalars s;
  s.neg_step_size = static_cast<float>(-(lr / bc1));
  s.bc2_sqrt = static_cast<float>(sqrt(bc2));
  s.w1 = static_cast<float>(1.0 - beta1);
  s.beta2 = static_cast<float>(beta2);
  s.w2 = static_cast<float>(1.0 - beta2);
  s.eps = static_cast<float>(eps);
  s.wd = static_cast<float>(weight_decay);
  s.maximize = maximize;

  const long long base = static_cast<long long>(c.y) * kChunk;
  const long long left = d.numel - base;
  const int cnt = left < kChunk ? static_cast<int>(left) : kChunk;
  float *__restrict__ p = d.param + base;
  const float *__restrict__ g = d.grad + base;
  float *__restrict__ m = d.exp_avg + base;
  float *__restrict__ v = d.exp_avg_sq + base;
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                     reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  if (vec) {
    const int n4 = cnt >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
      float4 pp = reinterpret_cast<const float4 *>(p)[i];
      const float4 gg = reinterpret_cast<const float4 *>(g)[i];
      float4 mm = reinterpret_cast<const float4 *>(m)[i];
      float4 vv = reinterpret_cast<const float4 *>(v)[i];
      adam_one(pp.x, gg.x, mm.x, vv.x, s);
      adam_one(pp.y, gg.y, mm.y, vv.y, s);
      adam_one(pp.z, gg.z, mm.z, vv.z, s);
      adam_one(pp.w, gg.w, mm.w, vv.w, s);
      reinterpret_cast<float4 *>(p)[i] = pp;
      reinterpret_cast<float4 *>(m)[i] = mm;
      reinterpret_cast<float4 *>(v)[i] = vv;
    }
    for (int i = 4 * n4 + threadIdx.x; i < cnt; i += 256) {
      float pp = p[i], mm = m[i], vv = v[i];
      adam_one(pp, g[i], mm, vv, s);
      p[i] = pp; m[i] = mm; v[i] = vv;
    }
  } else {
    for (int i = threadIdx.x; i < cnt; i += 256) {
      float pp = p[i], mm = m[i], vv = v[i];
      adam_one(pp, g[i], mm, vv, s);
      p[i] = pp; m[i] = mm; v[i] = vv;
    }
  }
  // Every chunk has READ the counter (its value went into the scalars above) by the time it arrives here; the last one
  // to arrive advances it.  No fence: nothing another workgroup WROTE has to be visible to the one that stores t + 1
  // (an agent-scope release per workgroup is what made the last-workgroup finalize of the BatchNorm reductions 2.5x
  // slower).
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nchunks = static_cast<int>((d.numel + kChunk - 1) / kChunk);
    if (atomicAdd(&done[c.x], 1) == nchunks - 1) {
      done[c.x] = 0;
      *d.step = step_old + 1.0f;
    }
  }
}

}  // namespace

extern "C" int nsdp_adam_chunk_elems(void) { return kChunk; }

extern "C" int nsdp_adam_multi_f32(const NsdpAdamDesc *descs_dev, const int32_t *chunks_dev, int n_chunks,
                                   int32_t *done_dev, const float *lr_dev, double lr, double beta1, double beta2,
                                   double eps, double weight_decay, int maximize, void *stream) {
  if (n_chunks <= 0) return 0;
  NSDP_REQUIRE(descs_dev && chunks_dev && done_dev, "adam_multi_f32: null table pointer");
  NSDP_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0,
               "adam_multi_f32: betas must be in [0, 1), eps >= 0 (beta1=%g beta2=%g eps=%g)", beta1, beta2, eps);
  NSDP_REQUIRE(lr_dev || lr >= 0.0, "adam_multi_f32: negative learning rate %g", lr);
  hipStream_t st = nsdp::as_stream(stream);
  hipLaunchKernelGGL(adam_multi_kernel, dim3(static_cast<unsigned>(n_chunks)), dim3(256), 0, st, descs_dev,
                     reinterpret_cast<const int2 *>(chunks_dev), done_dev, lr_dev, lr, beta1, beta2, eps, weight_decay,
                     maximize);
  return nsdp::launch_status("adam_multi_kernel");
}
