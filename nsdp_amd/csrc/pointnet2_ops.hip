// Gather / group / ball-query / 3-NN / 3-interpolate operators for gfx950 -- the rest of the
// `pointnet2_ops._ext` surface (reference: _ext-src/src/{sampling,group_points,ball_query,interpolate}_gpu.cu)
// plus the row-major gather/scatter that the model's `index_points` (model/utils.py:58-70) needs.
//
// The reference launches ONE block per batch element for most of these (grid = B), which cannot fill
// 256 CUs; here every kernel is a flat grid-stride launch over output elements with the fastest-moving
// output index on consecutive lanes (coalesced stores; gathers hit L2).  All of them are HBM-bound byte
// movers -- no LDS reuse exists except for the ball-query / 3-NN source cloud, which is LDS-tiled.
#include "common.h"
#include "prof.h"

#pragma clang fp contract(off)

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 256 * 8;  // 8 workgroups per CU, grid-stride beyond that

inline int grid_for(long long work) {
  long long g = (work + kThreads - 1) / kThreads;
  if (g > kMaxBlocks) g = kMaxBlocks;
  if (g < 1) g = 1;
  return static_cast<int>(g);
}

// out[b,c,j] = points[b,c,idx[b,j]]                         (sampling_gpu.cu:8-20)
__global__ void gather_points_kernel(const float *__restrict__ points, const int32_t *__restrict__ idx,
                                     long long total, int C, int N, int M, float *__restrict__ out) {
  for (long long e = blockIdx.x * static_cast<long long>(kThreads) + threadIdx.x; e < total;
       e += static_cast<long long>(gridDim.x) * kThreads) {
    const int j = static_cast<int>(e % M);
    const long long bc = e / M;
    const int b = static_cast<int>(bc / C);
    out[e] = points[bc * N + idx[static_cast<long long>(b) * M + j]];
  }
}

// grad_points[b,c,idx[b,j]] += grad_out[b,c,j]              (sampling_gpu.cu:34-47)
__global__ void gather_points_grad_kernel(const float *__restrict__ grad_out,
                                          const int32_t *__restrict__ idx, long long total, int C,
                                          int N, int M, float *__restrict__ grad_points) {
  for (long long e = blockIdx.x * static_cast<long long>(kThreads) + threadIdx.x; e < total;
       e += static_cast<long long>(gridDim.x) * kThreads) {
    const int j = static_cast<int>(e % M);
    const long long bc = e / M;
    const int b = static_cast<int>(bc / C);
    atomicAdd(grad_points + bc * N + idx[static_cast<long long>(b) * M + j], grad_out[e]);
  }
}

// out[b,c,j,k] = points[b,c,idx[b,j,k]]                     (group_points_gpu.cu:8-28)
__global__ void group_points_kernel(const float *__restrict__ points, const int32_t *__restrict__ idx,
                                    long long total, int C, int N, int NPNS,
                                    float *__restrict__ out) {
  for (long long e = blockIdx.x * static_cast<long long>(kThreads) + threadIdx.x; e < total;
       e += static_cast<long long>(gridDim.x) * kThreads) {
    const int jk = static_cast<int>(e % NPNS);
    const long long bc = e / NPNS;
    const int b = static_cast<int>(bc / C);
    out[e] = points[bc * N + idx[static_cast<long long>(b) * NPNS + jk]];
  }
}

// grad_points[b,c,idx[b,j,k]] += grad_out[b,c,j,k]          (group_points_gpu.cu:43-64)
__global__ void group_points_grad_kernel(const float *__restrict__ grad_out,
                                         const int32_t *__restrict__ idx, long long total, int C,
                                         int N, int NPNS, float *__restrict__ grad_points) {
  for (long long e = blockIdx.x * static_cast<long long>(kThreads) + threadIdx.x; e < total;
       e += static_cast<long long>(gridDim.x) * kThreads) {
    const int jk = static_cast<int>(e % NPNS);
    const long long bc = e / NPNS;
    const int b = static_cast<int>(bc / C);
    atomicAdd(grad_points + bc * N + idx[static_cast<long long>(b) * NPNS + jk], grad_out[e]);
  }
}

constexpr int kSrcTile = 1024;

// ball_query_gpu.cu:9-44: first `nsample` indices (in index order) with d2 < r^2; the first hit also
// pre-fills every slot; no hit leaves zeros (output is zero-filled before the launch).
__global__ __launch_bounds__(kThreads) void ball_query_kernel(
    const float *__restrict__ new_xyz_all, const float *__restrict__ xyz_all, int N, int M,
    float radius2, int nsample, int32_t *__restrict__ idx_all) {
  __shared__ float4 tile[kSrcTile];
  const int b = blockIdx.y;
  const float *xyz = xyz_all + static_cast<size_t>(b) * N * 3;
  const int j = blockIdx.x * kThreads + threadIdx.x;
  const bool active = j < M;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (active) {
    const float *q = new_xyz_all + (static_cast<size_t>(b) * M + j) * 3;
    qx = q[0]; qy = q[1]; qz = q[2];
  }
  int32_t *out = idx_all + (static_cast<size_t>(b) * M + (active ? j : 0)) * nsample;
  int cnt = active ? 0 : nsample;
  for (int base = 0; base < N; base += kSrcTile) {
    const int n_tile = min(kSrcTile, N - base);
    __syncthreads();
    for (int t = threadIdx.x; t < n_tile; t += kThreads) {
      const float *p = xyz + static_cast<size_t>(base + t) * 3;
      tile[t] = make_float4(p[0], p[1], p[2], 0.f);
    }
    __syncthreads();
    for (int t = 0; t < n_tile && cnt < nsample; ++t) {
      const float4 s = tile[t];
      const float d2 = nsdp::sq_dist3(qx, qy, qz, s.x, s.y, s.z);
      if (d2 < radius2) {
        const int k = base + t;
        if (cnt == 0)
          for (int l = 0; l < nsample; ++l) out[l] = k;
        out[cnt] = k;
        ++cnt;
      }
    }
  }
}

// interpolate_gpu.cu:9-59: three nearest `known` points per `unknown` point, bests kept in double.
__global__ __launch_bounds__(kThreads) void three_nn_kernel(const float *__restrict__ unknown_all,
                                                            const float *__restrict__ known_all, int n,
                                                            int m, float *__restrict__ dist2_all,
                                                            int32_t *__restrict__ idx_all) {
  __shared__ float4 tile[kSrcTile];
  const int b = blockIdx.y;
  const float *known = known_all + static_cast<size_t>(b) * m * 3;
  const int j = blockIdx.x * kThreads + threadIdx.x;
  const bool active = j < n;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (active) {
    const float *u = unknown_all + (static_cast<size_t>(b) * n + j) * 3;
    ux = u[0]; uy = u[1]; uz = u[2];
  }
  double best1 = 1e40, best2 = 1e40, best3 = 1e40;
  int besti1 = 0, besti2 = 0, besti3 = 0;
  for (int base = 0; base < m; base += kSrcTile) {
    const int n_tile = min(kSrcTile, m - base);
    __syncthreads();
    for (int t = threadIdx.x; t < n_tile; t += kThreads) {
      const float *p = known + static_cast<size_t>(base + t) * 3;
      tile[t] = make_float4(p[0], p[1], p[2], 0.f);
    }
    __syncthreads();
    for (int t = 0; t < n_tile; ++t) {
      const float4 s = tile[t];
      const double d = static_cast<double>(nsdp::sq_dist3(ux, uy, uz, s.x, s.y, s.z));
      const int k = base + t;
      if (d < best1) {
        best3 = best2; besti3 = besti2;
        best2 = best1; besti2 = besti1;
        best1 = d; besti1 = k;
      } else if (d < best2) {
        best3 = best2; besti3 = besti2;
        best2 = d; besti2 = k;
      } else if (d < best3) {
        best3 = d; besti3 = k;
      }
    }
  }
  if (active) {
    float *d2 = dist2_all + (static_cast<size_t>(b) * n + j) * 3;
    int32_t *io = idx_all + (static_cast<size_t>(b) * n + j) * 3;
    d2[0] = static_cast<float>(best1); d2[1] = static_cast<float>(best2); d2[2] = static_cast<float>(best3);
    io[0] = besti1; io[1] = besti2; io[2] = besti3;
  }
}

// out[b,l,j] = sum_t points[b,l,idx[b,j,t]] * weight[b,j,t]  (interpolate_gpu.cu:72-101)
__global__ void three_interpolate_kernel(const float *__restrict__ points,
                                         const int32_t *__restrict__ idx,
                                         const float *__restrict__ weight, long long total, int c,
                                         int m, int n, float *__restrict__ out) {
  for (long long e = blockIdx.x * static_cast<long long>(kThreads) + threadIdx.x; e < total;
       e += static_cast<long long>(gridDim.x) * kThreads) {
    const int j = static_cast<int>(e % n);
    const long long bl = e / n;
    const long long bj = (bl / c) * n + j;
    const float *p = points + bl * m;
    const float w1 = weight[bj * 3 + 0], w2 = weight[bj * 3 + 1], w3 = weight[bj * 3 + 2];
    const int i1 = idx[bj * 3 + 0], i2 = idx[bj * 3 + 1], i3 = idx[bj * 3 + 2];
    out[e] = p[i1] * w1 + p[i2] * w2 + p[i3] * w3;
  }
}

// interpolate_gpu.cu:116-143
__global__ void three_interpolate_grad_kernel(const float *__restrict__ grad_out,
                                              const int32_t *__restrict__ idx,
                                              const float *__restrict__ weight, long long total, int c,
                                              int n, int m, float *__restrict__ grad_points) {
  for (long long e = blockIdx.x * static_cast<long long>(kThreads) + threadIdx.x; e < total;
       e += static_cast<long long>(gridDim.x) * kThreads) {
    const int j = static_cast<int>(e % n);
    const long long bl = e / n;
    const long long bj = (bl / c) * n + j;
    float *g = grad_points + bl * m;
    const float go = grad_out[e];
    atomicAdd(g + idx[bj * 3 + 0], go * weight[bj * 3 + 0]);
    atomicAdd(g + idx[bj * 3 + 1], go * weight[bj * 3 + 1]);
    atomicAdd(g + idx[bj * 3 + 2], go * weight[bj * 3 + 2]);
  }
}

// index_points on row-major features: out[b,s,:] = points[b,idx[b,s],:]; VEC floats per lane.
template <int VEC>
__global__ void gather_rows_kernel(const float *__restrict__ points, const int32_t *__restrict__ idx,
                                   long long total_vec, int N, int CV, int S,
                                   float *__restrict__ out) {
  for (long long e = blockIdx.x * static_cast<long long>(kThreads) + threadIdx.x; e < total_vec;
       e += static_cast<long long>(gridDim.x) * kThreads) {
    const int cv = static_cast<int>(e % CV);
    const long long bs = e / CV;
    const long long b = bs / S;
    const long long src = (b * N + idx[bs]) * CV + cv;
    if (VEC == 4)
      reinterpret_cast<float4 *>(out)[e] = reinterpret_cast<const float4 *>(points)[src];
    else
      out[e] = points[src];
  }
}

__global__ void scatter_add_rows_kernel(const float *__restrict__ grad_out,
                                        const int32_t *__restrict__ idx, long long total, int N, int C,
                                        int S, float *__restrict__ grad_points) {
  for (long long e = blockIdx.x * static_cast<long long>(kThreads) + threadIdx.x; e < total;
       e += static_cast<long long>(gridDim.x) * kThreads) {
    const int c = static_cast<int>(e % C);
    const long long bs = e / C;
    const long long b = bs / S;
    atomicAdd(grad_points + (b * N + idx[bs]) * C + c, grad_out[e]);
  }
}

}  // namespace

extern "C" {

int nsdp_gather_points(const float *points, const int32_t *idx, int B, int C, int N, int M, float *out,
                       void *stream) {
  const long long total = static_cast<long long>(B) * C * M;
  if (total <= 0) return 0;
  NSDP_REQUIRE(points && idx && out && N > 0, "gather_points: bad argument");
  hipLaunchKernelGGL(gather_points_kernel, dim3(grid_for(total)), dim3(kThreads), 0,
                     nsdp::as_stream(stream), points, idx, total, C, N, M, out);
  return nsdp::launch_status("gather_points_kernel");
}

int nsdp_gather_points_grad(const float *grad_out, const int32_t *idx, int B, int C, int N, int M,
                            float *grad_points, void *stream) {
  const long long total = static_cast<long long>(B) * C * M;
  NSDP_REQUIRE(grad_points || static_cast<long long>(B) * C * N == 0, "gather_points_grad: null output");
  hipStream_t st = nsdp::as_stream(stream);
  if (static_cast<long long>(B) * C * N > 0)
    NSDP_HIP_TRY(hipMemsetAsync(grad_points, 0, sizeof(float) * static_cast<size_t>(B) * C * N, st));
  if (total <= 0) return 0;
  NSDP_REQUIRE(grad_out && idx, "gather_points_grad: null pointer");
  hipLaunchKernelGGL(gather_points_grad_kernel, dim3(grid_for(total)), dim3(kThreads), 0, st, grad_out,
                     idx, total, C, N, M, grad_points);
  return nsdp::launch_status("gather_points_grad_kernel");
}

int nsdp_group_points(const float *points, const int32_t *idx, int B, int C, int N, int NP, int NS,
                      float *out, void *stream) {
  const long long total = static_cast<long long>(B) * C * NP * NS;
  if (total <= 0) return 0;
  NSDP_REQUIRE(points && idx && out && N > 0, "group_points: bad argument");
  hipLaunchKernelGGL(group_points_kernel, dim3(grid_for(total)), dim3(kThreads), 0,
                     nsdp::as_stream(stream), points, idx, total, C, N, NP * NS, out);
  return nsdp::launch_status("group_points_kernel");
}

int nsdp_group_points_grad(const float *grad_out, const int32_t *idx, int B, int C, int N, int NP,
                           int NS, float *grad_points, void *stream) {
  const long long total = static_cast<long long>(B) * C * NP * NS;
  NSDP_REQUIRE(grad_points || static_cast<long long>(B) * C * N == 0, "group_points_grad: null output");
  hipStream_t st = nsdp::as_stream(stream);
  if (static_cast<long long>(B) * C * N > 0)
    NSDP_HIP_TRY(hipMemsetAsync(grad_points, 0, sizeof(float) * static_cast<size_t>(B) * C * N, st));
  if (total <= 0) return 0;
  NSDP_REQUIRE(grad_out && idx, "group_points_grad: null pointer");
  hipLaunchKernelGGL(group_points_grad_kernel, dim3(grid_for(total)), dim3(kThreads), 0, st, grad_out,
                     idx, total, C, N, NP * NS, grad_points);
  return nsdp::launch_status("group_points_grad_kernel");
}

int nsdp_ball_query(const float *new_xyz, const float *xyz, int B, int N, int M, float radius,
                    int nsample, int32_t *idx_out, void *stream) {
  const long long total = static_cast<long long>(B) * M * nsample;
  if (total <= 0) return 0;
  NSDP_REQUIRE(new_xyz && xyz && idx_out, "ball_query: null pointer");
  NSDP_REQUIRE(B <= 65535, "ball_query: batch %d too large", B);
  hipStream_t st = nsdp::as_stream(stream);
  NSDP_HIP_TRY(hipMemsetAsync(idx_out, 0, sizeof(int32_t) * static_cast<size_t>(total), st));
  const float radius2 = radius * radius;
  hipLaunchKernelGGL(ball_query_kernel, dim3(nsdp::ceil_div(M, kThreads), B), dim3(kThreads), 0, st,
                     new_xyz, xyz, N, M, radius2, nsample, idx_out);
  return nsdp::launch_status("ball_query_kernel");
}

int nsdp_three_nn(const float *unknown, const float *known, int B, int n, int m, float *dist2,
                  int32_t *idx, void *stream) {
  if (static_cast<long long>(B) * n <= 0) return 0;
  NSDP_REQUIRE(unknown && dist2 && idx && (known || m == 0), "three_nn: null pointer");
  NSDP_REQUIRE(B <= 65535, "three_nn: batch %d too large", B);
  hipLaunchKernelGGL(three_nn_kernel, dim3(nsdp::ceil_div(n, kThreads), B), dim3(kThreads), 0,
                     nsdp::as_stream(stream), unknown, known, n, m, dist2, idx);
  return nsdp::launch_status("three_nn_kernel");
}

int nsdp_three_interpolate(const float *points, const int32_t *idx, const float *weight, int B, int c,
                           int m, int n, float *out, void *stream) {
  const long long total = static_cast<long long>(B) * c * n;
  if (total <= 0) return 0;
  NSDP_REQUIRE(points && idx && weight && out && m > 0, "three_interpolate: bad argument");
  hipLaunchKernelGGL(three_interpolate_kernel, dim3(grid_for(total)), dim3(kThreads), 0,
                     nsdp::as_stream(stream), points, idx, weight, total, c, m, n, out);
  return nsdp::launch_status("three_interpolate_kernel");
}

int nsdp_three_interpolate_grad(const float *grad_out, const int32_t *idx, const float *weight, int B,
                                int c, int n, int m, float *grad_points, void *stream) {
  const long long total = static_cast<long long>(B) * c * n;
  NSDP_REQUIRE(grad_points || static_cast<long long>(B) * c * m == 0, "three_interpolate_grad: null output");
  hipStream_t st = nsdp::as_stream(stream);
  if (static_cast<long long>(B) * c * m > 0)
    NSDP_HIP_TRY(hipMemsetAsync(grad_points, 0, sizeof(float) * static_cast<size_t>(B) * c * m, st));
  if (total <= 0) return 0;
  NSDP_REQUIRE(grad_out && idx && weight, "three_interpolate_grad: null pointer");
  hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(grid_for(total)), dim3(kThreads), 0, st,
                     grad_out, idx, weight, total, c, n, m, grad_points);
  return nsdp::launch_status("three_interpolate_grad_kernel");
}

int nsdp_gather_rows(const float *points, const int32_t *idx, int B, int N, int C, int S, float *out,
                     void *stream) {
  const long long total = static_cast<long long>(B) * S * C;
  if (total <= 0) return 0;
  NSDP_REQUIRE(points && idx && out && N > 0, "gather_rows: bad argument");
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kGatherRows, st, 0.0,
                          4.0 * (static_cast<double>(B) * S * (1 + C) + static_cast<double>(B) * N * C));
  const bool vec4 = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(points) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
  if (vec4) {
    const long long tv = total / 4;
    hipLaunchKernelGGL((gather_rows_kernel<4>), dim3(grid_for(tv)), dim3(kThreads), 0, st, points, idx,
                       tv, N, C / 4, S, out);
  } else {
    hipLaunchKernelGGL((gather_rows_kernel<1>), dim3(grid_for(total)), dim3(kThreads), 0, st, points,
                       idx, total, N, C, S, out);
  }
  return nsdp::launch_status("gather_rows_kernel");
}

int nsdp_scatter_add_rows(const float *grad_out, const int32_t *idx, int B, int N, int C, int S,
                          float *grad_points, void *stream) {
  const long long total = static_cast<long long>(B) * S * C;
  NSDP_REQUIRE(grad_points || static_cast<long long>(B) * N * C == 0, "scatter_add_rows: null output");
  hipStream_t st = nsdp::as_stream(stream);
  if (static_cast<long long>(B) * N * C > 0)
    NSDP_HIP_TRY(hipMemsetAsync(grad_points, 0, sizeof(float) * static_cast<size_t>(B) * N * C, st));
  if (total <= 0) return 0;
  NSDP_REQUIRE(grad_out && idx, "scatter_add_rows: null pointer");
  nsdp::prof::Scope scope(nsdp::prof::kScatterRows, st, 0.0,
                          4.0 * (static_cast<double>(B) * S * (1 + C) + 2.0 * static_cast<double>(B) * N * C));
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid_for(total)), dim3(kThreads), 0, st, grad_out,
                     idx, total, N, C, S, grad_points);
  return nsdp::launch_status("scatter_add_rows_kernel");
}

}  // extern "C"
