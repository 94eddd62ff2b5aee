// Farthest-point sampling for gfx950 -- replaces furthest_point_sampling_kernel
// (/root/reference/pointnet2_ops_lib/pointnet2_ops/_ext-src/src/sampling_gpu.cu:69-229, host sampling.cpp:66-87).
//
// MI355X design (not the reference's): FPS is a chain of `nsamples` dependent arg-max steps, i.e.
// latency-bound (algorithmic HBM traffic is only N*12 + nsamples*4 bytes per cloud).  So
//   * the whole cloud and the running min-distance live in VGPRs (P points per lane), never re-read
//     from global/L2 as the reference does (it re-reads `dataset` and `temp` every iteration);
//   * one workgroup per cloud, 8 points per lane: T = 256 threads (one wave per SIMD) for N <= 2048,
//     a single wave for N <= 512 (no barrier at all), 512/1024 threads up to N = 8192; clouds of a
//     batch run on different CUs;
//   * the arg-max is a 64-bit key max {bits(min-dist) : tie-priority} done with DPP quad/row mirrors +
//     v_permlane16/32_swap (pure VALU, no LDS round trip), then ONE LDS hop across the 4 waves with
//     parity-double-buffered slots, i.e. one s_barrier per iteration;
//   * the winner's coordinates come from an LDS copy of the cloud (one ds_read_b128).
// Exactness: distances use one fp32 rounding per operation (file built with -ffp-contract=off), and the
// low key word encodes the reference kernel's tie rule -- thread t = k mod BS scans k ascending with a
// strict '>', and the shared-memory tree keeps the entry with the smaller bit-reversed thread id -- so
// the indices are the reference's even on exact ties (BS = opt_n_threads(N), cuda_utils.h:15-19).
#include <climits>
#include <cmath>

#include "common.h"
#include "prof.h"

#pragma clang fp contract(off)

namespace {

constexpr int kRankShift = 22;  // low bits: k div BS, high bits: bit-reversed (k mod BS)

template <int CTRL>
__device__ __forceinline__ long long dpp_max_step(long long v) {
  const int lo = __builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, static_cast<int>(v >> 32), CTRL, 0xf, 0xf, false);
  const long long o = (static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo);
  return o > v ? o : v;
}

// max over the 64 lanes of a wave, result in every lane.
__device__ __forceinline__ long long wave_max_i64(long long v) {
  v = dpp_max_step<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_max_step<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_max_step<0x141>(v);  // row_half_mirror
  v = dpp_max_step<0x140>(v);  // row_mirror  -> every lane of a 16-lane row holds the row max
  {
    const unsigned lo = static_cast<unsigned>(v), hi = static_cast<unsigned>(v >> 32);
    const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const long long a = (static_cast<long long>(h[0]) << 32) | l[0];
    const long long b = (static_cast<long long>(h[1]) << 32) | l[1];
    v = a > b ? a : b;
  }
  {
    const unsigned lo = static_cast<unsigned>(v), hi = static_cast<unsigned>(v >> 32);
    const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const long long a = (static_cast<long long>(h[0]) << 32) | l[0];
    const long long b = (static_cast<long long>(h[1]) << 32) | l[1];
    v = a > b ? a : b;
  }
  return v;
}

__device__ __forceinline__ unsigned tie_priority(int k, int BS, int log2BS) {
  const unsigned kmod = static_cast<unsigned>(k) & static_cast<unsigned>(BS - 1);
  const unsigned br = log2BS ? (__brev(kmod) >> (32 - log2BS)) : 0u;
  const unsigned rank = (br << kRankShift) | (static_cast<unsigned>(k) >> log2BS);
  return ~rank;  // larger = preferred on a tie
}

__device__ __forceinline__ int decode_winner(long long key, int log2BS) {
  if (key < 0) return 0;  // no valid point at all: reference keeps besti = 0
  const unsigned rank = ~static_cast<unsigned>(key);
  const unsigned br = rank >> kRankShift;
  const unsigned q = rank & ((1u << kRankShift) - 1u);
  const unsigned kmod = log2BS ? (__brev(br) >> (32 - log2BS)) : 0u;
  return static_cast<int>((q << log2BS) | kmod);
}

__device__ __forceinline__ bool point_valid(float x, float y, float z) {
  const float mag = (x * x) + (y * y) + (z * z);  // contraction is off in this file
  return !(static_cast<double>(mag) <= 1e-3);     // sampling_gpu.cu:100-101 (float vs double literal)
}

// Register-resident FPS: T threads, P points per thread (N <= T*P).
template <int T, int P, bool LDS_XYZ>
__global__ __launch_bounds__(T) void fps_reg_kernel(const float *__restrict__ xyz_all, int N, int M,
                                                    int BS, int log2BS,
                                                    int32_t *__restrict__ idx_all) {
  constexpr int W = T / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  long long *slots = reinterpret_cast<long long *>(smem);          // [2][W] (padded to 16*W bytes)
  float4 *sxyz = reinterpret_cast<float4 *>(smem + 16 * (W > 1 ? W : 1));

  const float *xyz = xyz_all + static_cast<size_t>(blockIdx.x) * N * 3;
  int32_t *out = idx_all + static_cast<size_t>(blockIdx.x) * M;
  const int tid = threadIdx.x;

  float px[P], py[P], pz[P], pt[P];
  unsigned prio[P];
#pragma unroll
  for (int s = 0; s < P; ++s) {
    const int k = tid + s * T;
    if (k < N) {
      const float x = xyz[k * 3 + 0], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
      px[s] = x; py[s] = y; pz[s] = z;
      pt[s] = point_valid(x, y, z) ? 1e10f : -1.0f;  // invalid points never win and never update
      prio[s] = tie_priority(k, BS, log2BS);
      if (LDS_XYZ) sxyz[k] = make_float4(x, y, z, 0.f);
    } else {
      px[s] = py[s] = pz[s] = 0.f;
      pt[s] = -1.0f;
      prio[s] = 0u;
    }
  }
  if (tid == 0) out[0] = 0;
  if (LDS_XYZ) __syncthreads();

  float cx, cy, cz;
  if (LDS_XYZ) {
    const float4 c = sxyz[0];
    cx = c.x; cy = c.y; cz = c.z;
  } else {
    cx = xyz[0]; cy = xyz[1]; cz = xyz[2];
  }

  for (int j = 1; j < M; ++j) {
    long long best = LLONG_MIN;
#pragma unroll
    for (int s = 0; s < P; ++s) {
      const float d = nsdp::sq_dist3(px[s], py[s], pz[s], cx, cy, cz);
      const float t = fminf(d, pt[s]);
      pt[s] = t;
      const long long key = (static_cast<long long>(__float_as_int(t)) << 32) | prio[s];
      best = key > best ? key : best;
    }
    best = wave_max_i64(best);
    if (W > 1) {
      long long *slot = slots + (j & 1) * W;
      if ((tid & 63) == 0) slot[tid >> 6] = best;
      __syncthreads();
      long long g = slot[0];
#pragma unroll
      for (int w = 1; w < W; ++w) {
        const long long o = slot[w];
        g = o > g ? o : g;
      }
      best = g;
    }
    const int old = decode_winner(best, log2BS);
    if (LDS_XYZ) {
      const float4 c = sxyz[old];
      cx = c.x; cy = c.y; cz = c.z;
    } else {
      cx = xyz[old * 3 + 0]; cy = xyz[old * 3 + 1]; cz = xyz[old * 3 + 2];
    }
    if (tid == 0) out[j] = old;
  }
}

// Generic fallback for very large clouds (N > 8192): running min-distance in global scratch.
template <int T>
__global__ __launch_bounds__(T) void fps_big_kernel(const float *__restrict__ xyz_all,
                                                    float *__restrict__ tmp_all, int N, int M, int BS,
                                                    int log2BS, int32_t *__restrict__ idx_all) {
  constexpr int W = T / 64;
  __shared__ long long slots[2 * W];
  const float *xyz = xyz_all + static_cast<size_t>(blockIdx.x) * N * 3;
  float *tmp = tmp_all + static_cast<size_t>(blockIdx.x) * N;
  int32_t *out = idx_all + static_cast<size_t>(blockIdx.x) * M;
  const int tid = threadIdx.x;
  for (int k = tid; k < N; k += T)
    tmp[k] = point_valid(xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2]) ? 1e10f : -1.0f;
  if (tid == 0) out[0] = 0;
  float cx = xyz[0], cy = xyz[1], cz = xyz[2];
  for (int j = 1; j < M; ++j) {
    long long best = LLONG_MIN;
    for (int k = tid; k < N; k += T) {
      const float d = nsdp::sq_dist3(xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2], cx, cy, cz);
      const float t = fminf(d, tmp[k]);
      tmp[k] = t;
      const long long key =
          (static_cast<long long>(__float_as_int(t)) << 32) | tie_priority(k, BS, log2BS);
      best = key > best ? key : best;
    }
    best = wave_max_i64(best);
    long long *slot = slots + (j & 1) * W;
    if ((tid & 63) == 0) slot[tid >> 6] = best;
    __syncthreads();
    long long g = slot[0];
#pragma unroll
    for (int w = 1; w < W; ++w) {
      const long long o = slot[w];
      g = o > g ? o : g;
    }
    const int old = decode_winner(g, log2BS);
    cx = xyz[old * 3 + 0]; cy = xyz[old * 3 + 1]; cz = xyz[old * 3 + 2];
    if (tid == 0) out[j] = old;
  }
}

// cuda_utils.h:15-19 -- same double arithmetic as the reference host code.
int opt_n_threads(int work_size) {
  const int pow_2 = static_cast<int>(std::log(static_cast<double>(work_size)) / std::log(2.0));
  int t = 1 << pow_2;
  if (t > 512) t = 512;
  if (t < 1) t = 1;
  return t;
}

template <int T, int P, bool LDS_XYZ>
int launch_reg(const float *xyz, int B, int N, int M, int BS, int log2BS, int32_t *idx, hipStream_t st) {
  constexpr int W = T / 64;
  const size_t smem = 16 * (W > 1 ? W : 1) + (LDS_XYZ ? static_cast<size_t>(N) * 16 : 0);
  if (smem > 64 * 1024) {
    NSDP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&fps_reg_kernel<T, P, LDS_XYZ>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem)));
  }
  hipLaunchKernelGGL((fps_reg_kernel<T, P, LDS_XYZ>), dim3(B), dim3(T), smem, st, xyz, N, M, BS, log2BS,
                     idx);
  return nsdp::launch_status("fps_reg_kernel");
}

}  // namespace

extern "C" int nsdp_furthest_point_sampling(const float *xyz, int B, int N, int nsamples, float *tmp,
                                            int32_t *idx_out, void *stream) {
  if (B <= 0 || nsamples <= 0) return 0;
  NSDP_REQUIRE(xyz && idx_out, "fps: null pointer");
  NSDP_REQUIRE(N > 0, "fps: N must be positive (got %d)", N);
  NSDP_REQUIRE((static_cast<long long>(N) >> kRankShift) == 0, "fps: N too large (%d)", N);
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kFps, st, 0.0, static_cast<double>(B) * (12.0 * N + 4.0 * nsamples));
  const int BS = opt_n_threads(N);
  int log2BS = 0;
  while ((1 << log2BS) < BS) ++log2BS;
  if (N <= 512) return launch_reg<64, 8, true>(xyz, B, N, nsamples, BS, log2BS, idx_out, st);
  if (N <= 2048) return launch_reg<256, 8, true>(xyz, B, N, nsamples, BS, log2BS, idx_out, st);
  if (N <= 4096) return launch_reg<512, 8, true>(xyz, B, N, nsamples, BS, log2BS, idx_out, st);
  if (N <= 8192) return launch_reg<1024, 8, false>(xyz, B, N, nsamples, BS, log2BS, idx_out, st);
  NSDP_REQUIRE(tmp, "fps: N=%d > 8192 needs the (B,N) scratch buffer", N);
  hipLaunchKernelGGL((fps_big_kernel<1024>), dim3(B), dim3(1024), 0, st, xyz, tmp, N, nsamples, BS,
                     log2BS, idx_out);
  return nsdp::launch_status("fps_big_kernel");
}
