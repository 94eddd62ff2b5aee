// K = 4 layers with bf16 storage: the first layer of every position-encoding MLP (3-d relative coordinates, zero-padded
// to 4; reference model/encoder/blocks.py:104-105, :290-291, model/decoder/blocks.py:72-76).  The input stays fp32
// (coordinates are never rounded to bf16), the output / the incoming gradient are bf16 tensors.  8 flops per output
// value: pure streams, nothing for the matrix pipe.
//   forward : Y[M,N] (bf16) = act( X[M,4] W[N,4]^T + b ): N/8 lanes per row, each with the 8 x 4 weights and the bias of
//             its eight channels in registers; one 16-byte load of the row's coordinates, one 16-byte store.
//   wgrad   : dW[N,4] = dY^T X, db = colsum(dY): N/8 lanes per row read dY as 16 bytes (8 bf16), keep 8 x 4 products and
//             8 column sums in registers; row slots of a workgroup combined through LDS in fixed order, per-workgroup
//             partials, fixed-order final reduce (deterministic).
// (dX = dY W is an ordinary nsdp_linear_bf16 call with 4 outputs.)
#include "common.h"
#include "prof.h"

namespace {

using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ __forceinline__ unsigned pack2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float hi_f(unsigned v) { return __builtin_bit_cast(float, v & 0xffff0000u); }

struct K4Fwd {
  const float *X, *W, *bias;     // [M,4], [N,4] row-major, [N] or null
  unsigned short *Y;             // [M,N] bf16
  long long M;
  int N, relu_out;
};

__global__ __launch_bounds__(256) void k4_fwd_bf16_kernel(K4Fwd p) {
  const int N = p.N;
  const int lpr = N >> 3, slots = 256 / lpr;                 // lanes per row (8 channels each), rows per iteration
  const int sub = threadIdx.x / lpr, cg = threadIdx.x - sub * lpr;
  if (sub >= slots) return;
  float4 w[8];
  float b[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    w[c] = *reinterpret_cast<const float4 *>(p.W + (8 * cg + c) * 4);
    b[c] = p.bias ? p.bias[8 * cg + c] : 0.f;
  }
  const long long stride = static_cast<long long>(gridDim.x) * slots;
  constexpr int U = 4;
  for (long long r = static_cast<long long>(blockIdx.x) * slots + sub; r < p.M; r += U * stride) {
    float4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long rr = r + u * stride;
      x[u] = *reinterpret_cast<const float4 *>(p.X + (rr < p.M ? rr : p.M - 1) * 4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long rr = r + u * stride;
      float y[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        y[c] = b[c] + (x[u].x * w[c].x + x[u].y * w[c].y + x[u].z * w[c].z + x[u].w * w[c].w);
        if (p.relu_out) y[c] = fmaxf(y[c], 0.f);
      }
      if (rr < p.M)
        *reinterpret_cast<u32x4 *>(p.Y + rr * N + 8 * cg) =
            u32x4{pack2(y[0], y[1]), pack2(y[2], y[3]), pack2(y[4], y[5]), pack2(y[6], y[7])};
    }
  }
}

struct K4Wg {
  const unsigned short *dY, *mask;   // [M,N] bf16 (mask: dY * (mask > 0), may be null)
  const float *X;                    // [M,4]
  float *ws;                         // per-workgroup partials [grid][N*4 + N]
  long long M, rows_per_wg;
  int N, want_db;
};

template <bool MASK>
__global__ __launch_bounds__(256) void k4_wgrad_bf16_kernel(K4Wg p) {
  __shared__ float red[256][41];      // 32 products + 8 column sums per thread (odd stride: conflict-free columns)
  const int N = p.N;
  const int lpr = N >> 3, slots = 256 / lpr;
  const int sub = threadIdx.x / lpr, cg = threadIdx.x - sub * lpr;
  const bool active = sub < slots;
  const long long r0 = static_cast<long long>(blockIdx.x) * p.rows_per_wg;
  long long r1 = r0 + p.rows_per_wg;
  r1 = r1 < p.M ? r1 : p.M;
  float a[8][4], bs[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    bs[c] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) a[c][k] = 0.f;
  }
  if (active) {
    constexpr int U = 4;
    for (long long r = r0 + sub; r < r1; r += static_cast<long long>(U) * slots) {
      u32x4 dy[U], mk[U];
      float4 x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        long long rr = r + static_cast<long long>(u) * slots;
        rr = rr < r1 ? rr : (r1 - 1);
        dy[u] = *reinterpret_cast<const u32x4 *>(p.dY + rr * N + 8 * cg);
        if (MASK) mk[u] = *reinterpret_cast<const u32x4 *>(p.mask + rr * N + 8 * cg);
        x[u] = *reinterpret_cast<const float4 *>(p.X + rr * 4);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool rv = r + static_cast<long long>(u) * slots < r1;
        float d[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          d[2 * q] = lo_f(dy[u][q]);
          d[2 * q + 1] = hi_f(dy[u][q]);
          if (MASK) {
            if (!(static_cast<int>(mk[u][q] << 16) > 0)) d[2 * q] = 0.f;
            if (!(static_cast<int>(mk[u][q] & 0xffff0000u) > 0)) d[2 * q + 1] = 0.f;
          }
        }
        const float xv[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float dc = rv ? d[c] : 0.f;
          bs[c] += dc;
#pragma unroll
          for (int k = 0; k < 4; ++k) a[c][k] += dc * xv[k];
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
#pragma unroll
    for (int k = 0; k < 4; ++k) red[threadIdx.x][4 * c + k] = a[c][k];
    red[threadIdx.x][32 + c] = bs[c];
  }
  __syncthreads();
  // thread t < lpr * 40: value v of channel group t / 40, summed over the row slots in fixed order
  float *out = p.ws + static_cast<long long>(blockIdx.x) * (static_cast<long long>(N) * 4 + N);
  for (int t = threadIdx.x; t < lpr * 40; t += 256) {
    const int q = t / 40, v = t - q * 40;
    float s = 0.f;
    for (int u = 0; u < slots; ++u) s += red[u * lpr + q][v];
    if (v < 32) out[(8 * q + (v >> 2)) * 4 + (v & 3)] = s;            // dW[n][k], n = 8 q + v / 4
    else if (p.want_db) out[static_cast<long long>(N) * 4 + 8 * q + (v - 32)] = s;
  }
}

__global__ __launch_bounds__(256) void k4_reduce_kernel(const float *__restrict__ ws, int S, long long stride, long long nw,
                                                        float *__restrict__ dW, long long nb, float *__restrict__ db) {
  const long long e = blockIdx.x * 256LL + threadIdx.x;
  if (e >= nw + nb) return;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int c = 0;
  for (; c + 8 <= S; c += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] += ws[(c + u) * stride + e];
  }
  for (; c < S; ++c) acc[0] += ws[c * stride + e];
  const float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  if (e < nw) dW[e] = s;
  else db[e - nw] = s;
}

inline int k4_grid(long long M, int N) {
  const int slots = 256 / (N >> 3);
  long long g = 4LL * nsdp::num_cus();
  const long long iters = (M + slots - 1) / slots;
  return static_cast<int>(g < iters ? g : (iters > 0 ? iters : 1));
}

}  // namespace

extern "C" {

int nsdp_linear_k4_bf16(const float *X, const float *W, const float *bias, void *Y, long long M, int N, int relu_out,
                        void *stream) {
  if (M <= 0 || N <= 0) return 0;
  NSDP_REQUIRE(X && W && Y, "linear_k4_bf16: null pointer");
  NSDP_REQUIRE(N % 8 == 0 && N >= 8 && N <= 256, "linear_k4_bf16: N=%d must be a multiple of 8 in [8, 256]", N);
  NSDP_REQUIRE(((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(Y)) & 15) == 0,
               "linear_k4_bf16: operands must be 16-byte aligned");
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kLinearB16, st, 8.0 * M * N, static_cast<double>(M) * (16.0 + 2.0 * N));
  K4Fwd p{X, W, bias, static_cast<unsigned short *>(Y), M, N, relu_out};
  hipLaunchKernelGGL(k4_fwd_bf16_kernel, dim3(k4_grid(M, N)), dim3(256), 0, st, p);
  return nsdp::launch_status("k4_fwd_bf16_kernel");
}

size_t nsdp_linear_wgrad_k4_bf16_workspace_bytes(long long M, int N) {
  if (M <= 0 || N <= 0) return 0;
  return static_cast<size_t>(k4_grid(M, N)) * (static_cast<size_t>(N) * 4 + N) * sizeof(float);
}

int nsdp_linear_wgrad_k4_bf16(const void *dY, const float *X, const void *mask, float *dW, float *db, long long M, int N,
                              float *workspace, size_t workspace_bytes, void *stream) {
  if (M <= 0 || N <= 0) return 0;
  NSDP_REQUIRE(dY && X && dW && workspace, "linear_wgrad_k4_bf16: null pointer");
  NSDP_REQUIRE(N % 8 == 0 && N >= 8 && N <= 256, "linear_wgrad_k4_bf16: N=%d must be a multiple of 8 in [8, 256]", N);
  NSDP_REQUIRE(workspace_bytes >= nsdp_linear_wgrad_k4_bf16_workspace_bytes(M, N), "linear_wgrad_k4_bf16: workspace too small");
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kWgradB16, st, 8.0 * M * N, static_cast<double>(M) * (16.0 + (mask ? 4.0 : 2.0) * N));
  int grid = k4_grid(M, N);
  if (grid > nsdp::num_cus()) grid = nsdp::num_cus();      // one partial per workgroup: keep the final reduce short
  const int slots = 256 / (N >> 3);
  long long per = (M + grid - 1) / grid;
  per = (per + slots - 1) / slots * slots;
  const int g2 = static_cast<int>((M + per - 1) / per);
  K4Wg p{static_cast<const unsigned short *>(dY), static_cast<const unsigned short *>(mask), X, workspace, M, per, N,
         db != nullptr};
  if (mask) hipLaunchKernelGGL(k4_wgrad_bf16_kernel<true>, dim3(g2), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(k4_wgrad_bf16_kernel<false>, dim3(g2), dim3(256), 0, st, p);
  int rc = nsdp::launch_status("k4_wgrad_bf16_kernel");
  if (rc) return rc;
  const long long nw = static_cast<long long>(N) * 4, nb = db ? N : 0;
  hipLaunchKernelGGL(k4_reduce_kernel, dim3(static_cast<unsigned>((nw + N + 255) / 256)), dim3(256), 0, st, workspace, g2,
                     nw + N, nw, dW, nb, db);
  return nsdp::launch_status("k4_reduce_kernel");
}

}  // extern "C"
