// BatchNorm1d over channels-last rows for gfx950 -- replaces the 35 nn.BatchNorm1d calls of the encoder
// (reference model/encoder/blocks.py:132, :158, :300-312: each one permutes to [B,C,n], runs 3-4 ATen
// kernels forward and 3 backward, plus separate ReLU / residual-add kernels around it).
//
// x is [R, C] (R = B*n rows, C = 120 or 256 channels, C % 4 == 0).  Per direction: one column-reduction
// kernel (float4 per lane, register partial sums, one LDS hop, per-workgroup partials) + a tiny finalize
// + one elementwise apply kernel.  The residual add in FRONT of the norm (bn(x + addend)) and the ReLU
// BEHIND it are fused into both passes, so the sum tensor and the pre-ReLU tensor never exist.
// Everything here is HBM-bound byte movement: 1 read for the statistics, 1 read + 1 write for the apply.
#include "common.h"
#include "prof.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxParts = 256;  // row chunks (one workgroup each) of the column reductions

struct Geo {
  int lpr;    // lanes per row = C / 4
  int rpi;    // rows handled per workgroup iteration = kThreads / lpr
};
__device__ __forceinline__ Geo geo(int C) {
  Geo g;
  g.lpr = C >> 2;
  g.rpi = kThreads / g.lpr;
  return g;
}

// Storage type of x / addend / y / dy / dx: float, or bf16 (BASELINE config 3).  Statistics, affine parameters and all
// arithmetic are fp32 in either mode.
struct bf16_t {
  unsigned short v;
};
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float4 ld4(const bf16_t *p) {
  const uint2 r = *reinterpret_cast<const uint2 *>(p);
  return make_float4(__builtin_bit_cast(float, r.x << 16), __builtin_bit_cast(float, r.x & 0xffff0000u),
                     __builtin_bit_cast(float, r.y << 16), __builtin_bit_cast(float, r.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(bf16_t *p, float4 v) {
  using f32x2 = __attribute__((ext_vector_type(2))) float;
  using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
  *reinterpret_cast<uint2 *>(p) = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v.x, v.y}, bf16x2)),
                                             __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v.z, v.w}, bf16x2)));
}
// per-channel pivot of the shifted sums (row 0 of the batch): sum (v - p) and sum (v - p)^2 do not cancel when
// |mean| >> std, unlike E[v^2] - mean^2 on raw fp32 sums
template <typename T>
__device__ __forceinline__ float4 pivot4(const T *x, const T *addend, int c4) {
  float4 p = ld4(x + 4 * c4);
  if (addend) {
    const float4 a = ld4(addend + 4 * c4);
    p.x += a.x; p.y += a.y; p.z += a.z; p.w += a.w;
  }
  return p;
}

// column sums of (v - pivot) and (v - pivot)^2 (v = x [+ addend]) over this workgroup's row chunk -> part[blk][2][C]
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_stats_kernel(const T *__restrict__ x,
                                                            const T *__restrict__ addend, long long R,
                                                            int C, long long rows_per_blk,
                                                            float *__restrict__ part) {
  __shared__ float4 red[2][kThreads];
  const Geo g = geo(C);
  const int sub = threadIdx.x / g.lpr, cq = threadIdx.x - sub * g.lpr;
  const bool active = sub < g.rpi;
  const long long r0 = blockIdx.x * rows_per_blk;
  const long long r1 = min(R, r0 + rows_per_blk);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
  if (active) {
    const float4 pv = pivot4(x, addend, cq);
    for (long long r = r0 + sub; r < r1; r += g.rpi) {
      float4 v = ld4(x + r * C + 4 * cq);
      if (addend) {
        const float4 a = ld4(addend + r * C + 4 * cq);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
      v.x -= pv.x; v.y -= pv.y; v.z -= pv.z; v.w -= pv.w;
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
    }
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = q;
  __syncthreads();
  if (threadIdx.x < g.lpr) {
    float4 ts = red[0][threadIdx.x], tq = red[1][threadIdx.x];
    for (int u = 1; u < g.rpi; ++u) {
      const float4 a = red[0][u * g.lpr + threadIdx.x], b = red[1][u * g.lpr + threadIdx.x];
      ts.x += a.x; ts.y += a.y; ts.z += a.z; ts.w += a.w;
      tq.x += b.x; tq.y += b.y; tq.z += b.z; tq.w += b.w;
    }
    float *o = part + static_cast<long long>(blockIdx.x) * 2 * C;
    *reinterpret_cast<float4 *>(o + 4 * threadIdx.x) = ts;
    *reinterpret_cast<float4 *>(o + C + 4 * threadIdx.x) = tq;
  }
}

// Column-wise combination of the per-workgroup partials: 16 lanes share one channel (each sums every
// 16th partial in double, then a 4-step xor shuffle), so the serial chain is nparts/16 long instead of
// nparts -- these finalize kernels sit on the critical path of 70 tiny launches per step.
__device__ __forceinline__ void combine_partials(const float *__restrict__ part, int nparts, int C, int c,
                                                 int sub16, double &s, double &q) {
  s = 0.0;
  q = 0.0;
  if (c < C) {
    for (int p = sub16; p < nparts; p += 16) {
      s += part[static_cast<long long>(p) * 2 * C + c];
      q += part[static_cast<long long>(p) * 2 * C + C + c];
    }
  }
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) {
    s += __shfl_xor(s, off);
    q += __shfl_xor(q, off);
  }
}

// mean / invstd from the partials + running-statistics update
// (momentum form of nn.BatchNorm1d: running = (1-m) running + m batch, unbiased variance for running_var)
template <typename T>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const T *__restrict__ x, const T *__restrict__ addend,
                                                          const float *__restrict__ part, int nparts, long long R,
                                                          int C, float eps, float momentum,
                                                          float *__restrict__ running_mean,
                                                          float *__restrict__ running_var, float *__restrict__ mean,
                                                          float *__restrict__ invstd,
                                                          long long *__restrict__ num_batches_tracked, int updates) {
  if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += updates;
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub16 = threadIdx.x & 15;
  double s, q;
  combine_partials(part, nparts, C, c, sub16, s, q);
  if (c >= C || sub16 != 0) return;
  const float4 pv4 = pivot4(x, addend, c >> 2);
  const float pv = (c & 3) == 0 ? pv4.x : (c & 3) == 1 ? pv4.y : (c & 3) == 2 ? pv4.z : pv4.w;
  const double ms = s / static_cast<double>(R);            // mean of the shifted values
  double var = q / static_cast<double>(R) - ms * ms;
  if (var < 0.0) var = 0.0;
  const double m = static_cast<double>(pv) + ms;
  mean[c] = static_cast<float>(m);
  invstd[c] = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
  if (running_mean) {
    const double unbiased = R > 1 ? var * static_cast<double>(R) / static_cast<double>(R - 1) : var;
    float rm = running_mean[c], rv = running_var[c];
    for (int u = 0; u < updates; ++u) {
      rm = static_cast<float>((1.0 - momentum) * rm + momentum * m);
      rv = static_cast<float>((1.0 - momentum) * rv + momentum * unbiased);
    }
    running_mean[c] = rm;
    running_var[c] = rv;
  }
}

// y = ((x [+ addend]) - mean) * invstd * gamma + beta, optional ReLU
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_apply_kernel(const T *__restrict__ x,
                                                            const T *__restrict__ addend,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ invstd,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, long long total4,
                                                            int C4, int relu, T *__restrict__ y) {
  for (long long e = blockIdx.x * static_cast<long long>(kThreads) + threadIdx.x; e < total4;
       e += static_cast<long long>(gridDim.x) * kThreads) {
    const int cq = static_cast<int>(e % C4);
    float4 v = ld4(x + 4 * e);
    if (addend) {
      const float4 a = ld4(addend + 4 * e);
      v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    const float4 m = ld4(mean + 4 * cq), is = ld4(invstd + 4 * cq), ga = ld4(gamma + 4 * cq), be = ld4(beta + 4 * cq);
    float4 o = make_float4((v.x - m.x) * is.x * ga.x + be.x, (v.y - m.y) * is.y * ga.y + be.y,
                           (v.z - m.z) * is.z * ga.z + be.z, (v.w - m.w) * is.w * ga.w + be.w);
    if (relu) {
      o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
    }
    st4(y + 4 * e, o);
  }
}

// backward column sums: dbeta = sum dy', dgamma = sum dy' * xhat, with dy' = dy * (y > 0) when relu
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_kernel(
    const T *__restrict__ dy, const T *__restrict__ y, const T *__restrict__ x,
    const T *__restrict__ addend, const float *__restrict__ mean, const float *__restrict__ invstd,
    long long R, int C, long long rows_per_blk, float *__restrict__ part) {
  __shared__ float4 red[2][kThreads];
  const Geo g = geo(C);
  const int sub = threadIdx.x / g.lpr, cq = threadIdx.x - sub * g.lpr;
  const bool active = sub < g.rpi;
  const long long r0 = blockIdx.x * rows_per_blk;
  const long long r1 = min(R, r0 + rows_per_blk);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
  if (active) {
    const float4 m = ld4(mean + 4 * cq), is = ld4(invstd + 4 * cq);
    for (long long r = r0 + sub; r < r1; r += g.rpi) {
      float4 d = ld4(dy + r * C + 4 * cq);
      if (y) {
        const float4 yv = ld4(y + r * C + 4 * cq);
        d.x = yv.x > 0.f ? d.x : 0.f; d.y = yv.y > 0.f ? d.y : 0.f;
        d.z = yv.z > 0.f ? d.z : 0.f; d.w = yv.w > 0.f ? d.w : 0.f;
      }
      float4 v = ld4(x + r * C + 4 * cq);
      if (addend) {
        const float4 a = ld4(addend + r * C + 4 * cq);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
      s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
      q.x += d.x * (v.x - m.x) * is.x; q.y += d.y * (v.y - m.y) * is.y;
      q.z += d.z * (v.z - m.z) * is.z; q.w += d.w * (v.w - m.w) * is.w;
    }
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = q;
  __syncthreads();
  if (threadIdx.x < g.lpr) {
    float4 ts = red[0][threadIdx.x], tq = red[1][threadIdx.x];
    for (int u = 1; u < g.rpi; ++u) {
      const float4 a = red[0][u * g.lpr + threadIdx.x], b = red[1][u * g.lpr + threadIdx.x];
      ts.x += a.x; ts.y += a.y; ts.z += a.z; ts.w += a.w;
      tq.x += b.x; tq.y += b.y; tq.z += b.z; tq.w += b.w;
    }
    float *o = part + static_cast<long long>(blockIdx.x) * 2 * C;
    *reinterpret_cast<float4 *>(o + 4 * threadIdx.x) = ts;
    *reinterpret_cast<float4 *>(o + C + 4 * threadIdx.x) = tq;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float *__restrict__ part, int nparts, int C,
                                                              float *__restrict__ dgamma,
                                                              float *__restrict__ dbeta) {
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub16 = threadIdx.x & 15;
  double s, q;
  combine_partials(part, nparts, C, c, sub16, s, q);
  if (c >= C || sub16 != 0) return;
  dbeta[c] = static_cast<float>(s);
  dgamma[c] = static_cast<float>(q);
}

// dx = gamma * invstd * (dy' - dbeta/R - xhat * dgamma/R)   (training)
// dx = gamma * invstd * dy'                                   (eval: statistics are constants)
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(
    const T *__restrict__ dy, const T *__restrict__ y, const T *__restrict__ x,
    const T *__restrict__ addend, const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ gamma, const float *__restrict__ dgamma, const float *__restrict__ dbeta,
    long long total4, int C4, float inv_r, int training, T *__restrict__ dx) {
  for (long long e = blockIdx.x * static_cast<long long>(kThreads) + threadIdx.x; e < total4;
       e += static_cast<long long>(gridDim.x) * kThreads) {
    const int cq = static_cast<int>(e % C4);
    float4 d = ld4(dy + 4 * e);
    if (y) {
      const float4 yv = ld4(y + 4 * e);
      d.x = yv.x > 0.f ? d.x : 0.f; d.y = yv.y > 0.f ? d.y : 0.f;
      d.z = yv.z > 0.f ? d.z : 0.f; d.w = yv.w > 0.f ? d.w : 0.f;
    }
    const float4 is = ld4(invstd + 4 * cq), ga = ld4(gamma + 4 * cq);
    float4 o;
    if (training) {
      float4 v = ld4(x + 4 * e);
      if (addend) {
        const float4 a = ld4(addend + 4 * e);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
      const float4 m = ld4(mean + 4 * cq), dg = ld4(dgamma + 4 * cq), db = ld4(dbeta + 4 * cq);
      o = make_float4(ga.x * is.x * (d.x - db.x * inv_r - (v.x - m.x) * is.x * dg.x * inv_r),
                      ga.y * is.y * (d.y - db.y * inv_r - (v.y - m.y) * is.y * dg.y * inv_r),
                      ga.z * is.z * (d.z - db.z * inv_r - (v.z - m.z) * is.z * dg.z * inv_r),
                      ga.w * is.w * (d.w - db.w * inv_r - (v.w - m.w) * is.w * dg.w * inv_r));
    } else {
      o = make_float4(ga.x * is.x * d.x, ga.y * is.y * d.y, ga.z * is.z * d.z, ga.w * is.w * d.w);
    }
    st4(dx + 4 * e, o);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// One-launch forms for R <= 16384 rows (34 of the encoder's 35 BatchNorms at 32 shapes per GPU: 3 200 and 16 000 rows).
//
// A workgroup owns a SLAB of VEC channels and ALL rows, held in registers (row t + i*THREADS in slot i of thread t), so the
// statistics, the running-average update and the normalisation need no pass across workgroups: one launch instead of
// stats + finalize + apply (a dependent kernel boundary costs 1.5-1.9 us, more than these tensors take to stream), and x is
// read ONCE.  Every lane issues all of its loads before the first use (up to 16 x 16 B in flight per lane).  Slabs that
// share 128-byte lines run on the same XCD (block b runs on XCD b % 8), so the line is fetched into one L2.  The
// variance is the two-pass form (the values are in registers); sums are combined in double in a fixed order (bit-reproducible).
// ---------------------------------------------------------------------------------------------------------------------------
template <int VEC>
struct VecOps;
template <>
struct VecOps<4> {
  template <typename T>
  static __device__ __forceinline__ void ld(const T *p, float (&v)[4]) {
    const float4 t = ld4(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  template <typename T>
  static __device__ __forceinline__ void st(T *p, const float (&v)[4]) { st4(p, make_float4(v[0], v[1], v[2], v[3])); }
};
template <>
struct VecOps<2> {
  static __device__ __forceinline__ void ld(const float *p, float (&v)[2]) {
    const float2 t = *reinterpret_cast<const float2 *>(p);
    v[0] = t.x; v[1] = t.y;
  }
  static __device__ __forceinline__ void ld(const bf16_t *p, float (&v)[2]) {
    const unsigned r = *reinterpret_cast<const unsigned *>(p);
    v[0] = __builtin_bit_cast(float, r << 16);
    v[1] = __builtin_bit_cast(float, r & 0xffff0000u);
  }
  static __device__ __forceinline__ void st(float *p, const float (&v)[2]) {
    *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]);
  }
  static __device__ __forceinline__ void st(bf16_t *p, const float (&v)[2]) {
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
    *reinterpret_cast<unsigned *>(p) = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
  }
};

// sum of N per-thread values over the workgroup, every thread gets the totals (fixed combination order)
template <int N, int THREADS>
__device__ __forceinline__ void block_sum(double (&a)[N], double *red /* [N][THREADS/64] */) {
  constexpr int W = THREADS / 64;
#pragma unroll
  for (int j = 0; j < N; ++j) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) a[j] += __shfl_xor(a[j], off);
  }
  __syncthreads();          // (red may still be read by the previous call)
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int j = 0; j < N; ++j) red[j * W + (threadIdx.x >> 6)] = a[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double t = red[j * W];
#pragma unroll
    for (int w = 1; w < W; ++w) t += red[j * W + w];
    a[j] = t;
  }
}

// slab of block b: blocks b, b + 8, b + 16, ... (one XCD) take adjacent slabs
__device__ __forceinline__ int slab_of_block(int nslabs) {
  const int per = (nslabs + 7) >> 3;
  const int j = blockIdx.x >> 3;
  const int s = (blockIdx.x & 7) * per + j;
  return (j < per && s < nslabs) ? s : -1;
}

template <typename T, int VEC, int NPT, int THREADS>
__global__ __launch_bounds__(THREADS) void bn_slab_fwd_kernel(
    const T *__restrict__ x, const T *__restrict__ addend, int R, int C, float eps, float momentum, int updates,
    float *__restrict__ running_mean, float *__restrict__ running_var, long long *__restrict__ num_batches_tracked,
    const float *__restrict__ gamma, const float *__restrict__ beta, int relu, T *__restrict__ y,
    float *__restrict__ mean_out, float *__restrict__ invstd_out) {
  __shared__ double red[VEC * (THREADS / 64)];
  if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += updates;
  const int slab = slab_of_block(C / VEC);
  if (slab < 0) return;
  const int c0 = slab * VEC;
  // (everything the tail of the kernel needs is requested now, next to the row loads: a second dependent memory round trip
  // costs 1-2 us in a kernel that takes 7)
  float ga[VEC], be[VEC], rm0[VEC], rv0[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    ga[j] = gamma[c0 + j];
    be[j] = beta[c0 + j];
    rm0[j] = running_mean ? running_mean[c0 + j] : 0.f;
    rv0[j] = running_mean ? running_var[c0 + j] : 0.f;
  }
  float v[NPT][VEC];
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int r = i * THREADS + threadIdx.x;
    if (r < R) {
      VecOps<VEC>::ld(x + static_cast<long long>(r) * C + c0, v[i]);
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[i][j] = 0.f;
    }
  }
  if (addend) {
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int r = i * THREADS + threadIdx.x;
      if (r < R) {
        float a[VEC];
        VecOps<VEC>::ld(addend + static_cast<long long>(r) * C + c0, a);
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[i][j] += a[j];
      }
    }
  }
  double acc[VEC];
  {
    float s[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = 0.f;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) s[j] += v[i][j];      // (rows past R hold zeros)
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = s[j];
  }
  block_sum<VEC, THREADS>(acc, red);
  float mean[VEC];
  double mean_d[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    mean_d[j] = acc[j] / static_cast<double>(R);
    mean[j] = static_cast<float>(mean_d[j]);
  }
  {
    float q[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) q[j] = 0.f;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const bool ok = i * THREADS + static_cast<int>(threadIdx.x) < R;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float d = v[i][j] - mean[j];
        q[j] += ok ? d * d : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = q[j];
  }
  block_sum<VEC, THREADS>(acc, red);
  float invstd[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    // (sum (v - mean_f32)^2 = sum (v - mean)^2 + R (mean - mean_f32)^2: the rounding of the mean is removed exactly)
    const double dm = mean_d[j] - static_cast<double>(mean[j]);
    double var = acc[j] / static_cast<double>(R) - dm * dm;
    if (var < 0.0) var = 0.0;
    invstd[j] = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
    if (threadIdx.x == 0) {
      mean_out[c0 + j] = mean[j];
      invstd_out[c0 + j] = invstd[j];
      if (running_mean) {
        const double unbiased = R > 1 ? var * static_cast<double>(R) / static_cast<double>(R - 1) : var;
        float rm = rm0[j], rv = rv0[j];
        for (int u = 0; u < updates; ++u) {       // (the same batch statistics folded in `updates` times, see nsdp_bn_train_fwd)
          rm = static_cast<float>((1.0 - momentum) * rm + momentum * mean_d[j]);
          rv = static_cast<float>((1.0 - momentum) * rv + momentum * unbiased);
        }
        running_mean[c0 + j] = rm;
        running_var[c0 + j] = rv;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int r = i * THREADS + threadIdx.x;
    if (r < R) {
      float o[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        o[j] = (v[i][j] - mean[j]) * invstd[j] * ga[j] + be[j];
        if (relu) o[j] = fmaxf(o[j], 0.f);
      }
      VecOps<VEC>::st(y + static_cast<long long>(r) * C + c0, o);
    }
  }
}

template <typename T, int VEC, int NPT, int THREADS>
__global__ __launch_bounds__(THREADS) void bn_slab_bwd_kernel(
    const T *__restrict__ dy, const T *__restrict__ y, const T *__restrict__ x, const T *__restrict__ addend,
    const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ gamma, int R, int C,
    int training, T *__restrict__ dx, float *__restrict__ dgamma, float *__restrict__ dbeta) {
  __shared__ double red[2 * VEC * (THREADS / 64)];
  const int slab = slab_of_block(C / VEC);
  if (slab < 0) return;
  const int c0 = slab * VEC;
  float d[NPT][VEC], xh[NPT][VEC];
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int r = i * THREADS + threadIdx.x;
    if (r < R) {
      VecOps<VEC>::ld(dy + static_cast<long long>(r) * C + c0, d[i]);
      VecOps<VEC>::ld(x + static_cast<long long>(r) * C + c0, xh[i]);
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) d[i][j] = xh[i][j] = 0.f;
    }
  }
  if (y) {
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int r = i * THREADS + threadIdx.x;
      if (r < R) {
        float yv[VEC];
        VecOps<VEC>::ld(y + static_cast<long long>(r) * C + c0, yv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) d[i][j] = yv[j] > 0.f ? d[i][j] : 0.f;
      }
    }
  }
  if (addend) {
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int r = i * THREADS + threadIdx.x;
      if (r < R) {
        float a[VEC];
        VecOps<VEC>::ld(addend + static_cast<long long>(r) * C + c0, a);
#pragma unroll
        for (int j = 0; j < VEC; ++j) xh[i][j] += a[j];
      }
    }
  }
  float m[VEC], is[VEC], ga[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    m[j] = mean[c0 + j];
    is[j] = invstd[c0 + j];
    ga[j] = gamma[c0 + j];
  }
  double acc[2 * VEC];
  {
    float s[VEC], q[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = q[j] = 0.f;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        xh[i][j] = (xh[i][j] - m[j]) * is[j];
        s[j] += d[i][j];                       // (rows past R: d = 0)
        q[j] += d[i][j] * xh[i][j];
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      acc[j] = s[j];
      acc[VEC + j] = q[j];
    }
  }
  block_sum<2 * VEC, THREADS>(acc, red);
  float db[VEC], dg[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    db[j] = static_cast<float>(acc[j]);
    dg[j] = static_cast<float>(acc[VEC + j]);
    if (threadIdx.x == 0) {
      dbeta[c0 + j] = db[j];
      dgamma[c0 + j] = dg[j];
    }
  }
  const float inv_r = 1.0f / static_cast<float>(R);
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int r = i * THREADS + threadIdx.x;
    if (r < R) {
      float o[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        o[j] = training ? ga[j] * is[j] * (d[i][j] - db[j] * inv_r - xh[i][j] * dg[j] * inv_r) : ga[j] * is[j] * d[i][j];
      }
      VecOps<VEC>::st(dx + static_cast<long long>(r) * C + c0, o);
    }
  }
}

int g_bn_slab = 1;   // nsdp_debug_set(11, v), NSDP_BN_SLAB: 0 = always the three-launch forms, 2 = slabs up to 16384 rows (A/B)
// longer slabs lose to the three-launch forms (every 16-byte row piece pulls a whole 128-byte line into L1: 16 000 x 256 takes
// 29 us forward / 64 us backward against 26 / 30): they stay compiled for the A/B, nsdp_debug_set(11, 2)
inline long long slab_max_rows() { return g_bn_slab >= 2 ? 16384 : 4096; }
inline int slab_grid(int nslabs) { return 8 * ((nslabs + 7) / 8); }

struct Plan {
  int parts;
  long long rows_per_blk;
};
inline Plan plan(long long R, int C) {
  const int rpi = kThreads / (C >> 2);
  long long rows = (R + kMaxParts - 1) / kMaxParts;
  const long long min_rows = 8LL * rpi;
  if (rows < min_rows) rows = min_rows;
  rows = (rows + rpi - 1) / rpi * rpi;
  Plan p;
  p.rows_per_blk = rows;
  p.parts = static_cast<int>((R + rows - 1) / rows);
  return p;
}
inline int ew_grid(long long total4) {
  long long g = (total4 + kThreads - 1) / kThreads;
  if (g > 2048) g = 2048;
  return static_cast<int>(g < 1 ? 1 : g);
}
inline bool c_ok(int C) { return C >= 4 && C % 4 == 0 && C <= 1024; }

template <typename T>
int bn_stats_t(const T *x, const T *addend, long long R, int C, float eps, float momentum, float *running_mean,
               float *running_var, float *mean, float *invstd, float *workspace, long long *num_batches, void *stream,
               int updates = 1) {
  NSDP_REQUIRE(R > 0 && c_ok(C), "bn_stats: need R > 0 and C %% 4 == 0, C <= 1024 (R=%lld C=%d)", R, C);
  NSDP_REQUIRE(x && mean && invstd && workspace, "bn_stats: null pointer");
  NSDP_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_stats: running stats go together");
  hipStream_t st = nsdp::as_stream(stream);
  const Plan p = plan(R, C);
  nsdp::prof::Scope scope(nsdp::prof::kBatchNorm, st, 0.0, sizeof(T) * 1.0 * R * C * (addend ? 2 : 1));
  hipLaunchKernelGGL(bn_stats_kernel<T>, dim3(p.parts), dim3(kThreads), 0, st, x, addend, R, C, p.rows_per_blk, workspace);
  hipLaunchKernelGGL(bn_finalize_kernel<T>, dim3((C + 15) / 16), dim3(256), 0, st, x, addend, workspace, p.parts, R, C, eps,
                     momentum, running_mean, running_var, mean, invstd, num_batches, updates);
  return nsdp::launch_status("bn_stats_kernel");
}

template <typename T>
int bn_apply_t(const T *x, const T *addend, const float *mean, const float *invstd, const float *gamma, const float *beta,
               long long R, int C, int relu, T *y, void *stream) {
  if (R <= 0) return 0;
  NSDP_REQUIRE(c_ok(C), "bn_apply: C=%d must be a multiple of 4, <= 1024", C);
  NSDP_REQUIRE(x && mean && invstd && gamma && beta && y, "bn_apply: null pointer");
  hipStream_t st = nsdp::as_stream(stream);
  const long long total4 = R * (C >> 2);
  nsdp::prof::Scope scope(nsdp::prof::kBatchNorm, st, 0.0, sizeof(T) * 1.0 * R * C * (addend ? 3 : 2));
  hipLaunchKernelGGL(bn_apply_kernel<T>, dim3(ew_grid(total4)), dim3(kThreads), 0, st, x, addend, mean, invstd, gamma, beta,
                     total4, C >> 2, relu, y);
  return nsdp::launch_status("bn_apply_kernel");
}

template <typename T>
int bn_train_fwd_t(const T *x, const T *addend, long long R, int C, float eps, float momentum, int updates,
                   float *running_mean, float *running_var, long long *num_batches, const float *gamma, const float *beta,
                   int relu, T *y, float *mean, float *invstd, float *workspace, void *stream) {
  NSDP_REQUIRE(R > 0 && c_ok(C), "bn_train_fwd: need R > 0 and C %% 4 == 0, C <= 1024 (R=%lld C=%d)", R, C);
  NSDP_REQUIRE(x && gamma && beta && y && mean && invstd && workspace, "bn_train_fwd: null pointer");
  NSDP_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_fwd: running stats go together");
  NSDP_REQUIRE(updates >= 1 && updates <= 64, "bn_train_fwd: updates=%d", updates);
  if (!g_bn_slab || R > slab_max_rows()) {
    const int rc = bn_stats_t<T>(x, addend, R, C, eps, momentum, running_mean, running_var, mean, invstd, workspace,
                                 num_batches, stream, updates);
    if (rc) return rc;
    return bn_apply_t<T>(x, addend, mean, invstd, gamma, beta, R, C, relu, y, stream);
  }
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kBatchNorm, st, 0.0, sizeof(T) * 1.0 * R * C * (addend ? 3 : 2));
  const int r = static_cast<int>(R);
  if (R <= 4096) {
    NSDP_TRACE("bn_slab_fwd<4,256>");
    hipLaunchKernelGGL((bn_slab_fwd_kernel<T, 4, 16, 256>), dim3(slab_grid(C / 4)), dim3(256), 0, st, x, addend, r, C, eps,
                       momentum, updates, running_mean, running_var, num_batches, gamma, beta, relu, y, mean, invstd);
  } else {
    NSDP_TRACE("bn_slab_fwd<4,512>");
    hipLaunchKernelGGL((bn_slab_fwd_kernel<T, 4, 32, 512>), dim3(slab_grid(C / 4)), dim3(512), 0, st, x, addend, r, C, eps,
                       momentum, updates, running_mean, running_var, num_batches, gamma, beta, relu, y, mean, invstd);
  }
  return nsdp::launch_status("bn_slab_fwd_kernel");
}

template <typename T>
int bn_backward_t(const T *dy, const T *y_relu, const T *x, const T *addend, const float *mean, const float *invstd,
                  const float *gamma, long long R, int C, int training, T *dx, float *dgamma, float *dbeta,
                  float *workspace, void *stream) {
  NSDP_REQUIRE(R > 0 && c_ok(C), "bn_backward: need R > 0 and C %% 4 == 0, C <= 1024");
  NSDP_REQUIRE(dy && x && mean && invstd && gamma && dx && dgamma && dbeta && workspace, "bn_backward: null pointer");
  hipStream_t st = nsdp::as_stream(stream);
  if (g_bn_slab && R <= slab_max_rows()) {
    nsdp::prof::Scope scope(nsdp::prof::kBatchNorm, st, 0.0,
                            sizeof(T) * 1.0 * R * C * (3.0 + (y_relu ? 1 : 0) + (addend ? 1 : 0)));
    const int r = static_cast<int>(R);
    if (R <= 4096) {
      NSDP_TRACE("bn_slab_bwd<4,256>");
      hipLaunchKernelGGL((bn_slab_bwd_kernel<T, 4, 16, 256>), dim3(slab_grid(C / 4)), dim3(256), 0, st, dy, y_relu, x, addend,
                         mean, invstd, gamma, r, C, training, dx, dgamma, dbeta);
    } else if (R <= 8192) {
      NSDP_TRACE("bn_slab_bwd<4,512>");
      hipLaunchKernelGGL((bn_slab_bwd_kernel<T, 4, 16, 512>), dim3(slab_grid(C / 4)), dim3(512), 0, st, dy, y_relu, x, addend,
                         mean, invstd, gamma, r, C, training, dx, dgamma, dbeta);
    } else {
      NSDP_TRACE("bn_slab_bwd<2,512>");
      hipLaunchKernelGGL((bn_slab_bwd_kernel<T, 2, 32, 512>), dim3(slab_grid(C / 2)), dim3(512), 0, st, dy, y_relu, x,
                         addend, mean, invstd, gamma, r, C, training, dx, dgamma, dbeta);
    }
    return nsdp::launch_status("bn_slab_bwd_kernel");
  }
  const Plan p = plan(R, C);
  const long long total4 = R * (C >> 2);
  nsdp::prof::Scope scope(nsdp::prof::kBatchNorm, st, 0.0,
                          sizeof(T) * 1.0 * R * C * (5.0 + (y_relu ? 2 : 0) + (addend ? 2 : 0)));
  hipLaunchKernelGGL(bn_bwd_reduce_kernel<T>, dim3(p.parts), dim3(kThreads), 0, st, dy, y_relu, x, addend, mean, invstd, R,
                     C, p.rows_per_blk, workspace);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, st, workspace, p.parts, C, dgamma, dbeta);
  hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(ew_grid(total4)), dim3(kThreads), 0, st, dy, y_relu, x, addend, mean,
                     invstd, gamma, dgamma, dbeta, total4, C >> 2, 1.0f / static_cast<float>(R), training, dx);
  return nsdp::launch_status("bn_backward");
}

}  // namespace

namespace nsdp {
void debug_set_bn(int value) { g_bn_slab = value; }
}  // namespace nsdp

extern "C" {

size_t nsdp_bn_workspace_bytes(int C) { return static_cast<size_t>(kMaxParts) * 2 * C * sizeof(float); }

int nsdp_bn_stats(const float *x, const float *addend, long long R, int C, float eps, float momentum,
                  float *running_mean, float *running_var, float *mean, float *invstd, float *workspace,
                  long long *num_batches_tracked, void *stream) {
  return bn_stats_t<float>(x, addend, R, C, eps, momentum, running_mean, running_var, mean, invstd, workspace,
                           num_batches_tracked, stream);
}
int nsdp_bn_train_fwd(const float *x, const float *addend, long long R, int C, float eps, float momentum, int updates,
                      float *running_mean, float *running_var, long long *num_batches_tracked, const float *gamma,
                      const float *beta, int relu, float *y, float *mean, float *invstd, float *workspace, void *stream) {
  return bn_train_fwd_t<float>(x, addend, R, C, eps, momentum, updates, running_mean, running_var, num_batches_tracked,
                               gamma, beta, relu, y, mean, invstd, workspace, stream);
}
int nsdp_bn_apply(const float *x, const float *addend, const float *mean, const float *invstd,
                  const float *gamma, const float *beta, long long R, int C, int relu, float *y, void *stream) {
  return bn_apply_t<float>(x, addend, mean, invstd, gamma, beta, R, C, relu, y, stream);
}
int nsdp_bn_backward(const float *dy, const float *y_relu, const float *x, const float *addend,
                     const float *mean, const float *invstd, const float *gamma, long long R, int C,
                     int training, float *dx, float *dgamma, float *dbeta, float *workspace, void *stream) {
  return bn_backward_t<float>(dy, y_relu, x, addend, mean, invstd, gamma, R, C, training, dx, dgamma, dbeta, workspace, stream);
}

// bf16-storage variants: x, addend, y, dy, dx are bf16 tensors; statistics / affine parameters / their gradients fp32
#define B16(p) reinterpret_cast<const bf16_t *>(p)
int nsdp_bn_stats_bf16(const void *x, const void *addend, long long R, int C, float eps, float momentum,
                       float *running_mean, float *running_var, float *mean, float *invstd, float *workspace,
                       long long *num_batches_tracked, void *stream) {
  return bn_stats_t<bf16_t>(B16(x), B16(addend), R, C, eps, momentum, running_mean, running_var, mean, invstd, workspace,
                            num_batches_tracked, stream);
}
int nsdp_bn_train_fwd_bf16(const void *x, const void *addend, long long R, int C, float eps, float momentum, int updates,
                           float *running_mean, float *running_var, long long *num_batches_tracked, const float *gamma,
                           const float *beta, int relu, void *y, float *mean, float *invstd, float *workspace,
                           void *stream) {
  return bn_train_fwd_t<bf16_t>(B16(x), B16(addend), R, C, eps, momentum, updates, running_mean, running_var,
                                num_batches_tracked, gamma, beta, relu, reinterpret_cast<bf16_t *>(y), mean, invstd,
                                workspace, stream);
}
int nsdp_bn_apply_bf16(const void *x, const void *addend, const float *mean, const float *invstd, const float *gamma,
                       const float *beta, long long R, int C, int relu, void *y, void *stream) {
  return bn_apply_t<bf16_t>(B16(x), B16(addend), mean, invstd, gamma, beta, R, C, relu, reinterpret_cast<bf16_t *>(y), stream);
}
int nsdp_bn_backward_bf16(const void *dy, const void *y_relu, const void *x, const void *addend, const float *mean,
                          const float *invstd, const float *gamma, long long R, int C, int training, void *dx,
                          float *dgamma, float *dbeta, float *workspace, void *stream) {
  return bn_backward_t<bf16_t>(B16(dy), B16(y_relu), B16(x), B16(addend), mean, invstd, gamma, R, C, training,
                               reinterpret_cast<bf16_t *>(dx), dgamma, dbeta, workspace, stream);
}
#undef B16

}  // extern "C"
