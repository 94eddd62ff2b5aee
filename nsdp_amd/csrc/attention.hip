// Point-Transformer vector attention glue for gfx950: everything between the dense layers of one
// attention block, fused into two memory passes per direction.
//
// Reference (ATen, every step a materialised [B,n,k,d] tensor): model/encoder/blocks.py:104-124 and
// :290-308, model/decoder/blocks.py:72-91:
//     k_nb = index_points(w_ks(x), idx); v_nb = index_points(w_vs(x), idx)          (2 gathers)
//     u    = q[:, :, None] - k_nb + pos                                               (2 elementwise)
//     attn = softmax(fc_gamma(u), dim=-2)                                             (softmax over k, per channel)
//     res  = einsum('bmnf,bmnf->bmf', attn, v_nb + pos) (+ x)                         (add, mul, reduce, add)
// Here:
//     attn_pre_fwd :  u = q_i - kf[idx] + pos                       (gather fused, one write)
//     attn_post_fwd:  y_i = sum_j softmax_j(a)_j * (vf[idx] + pos) (+ global token) (+ residual), lse_i
//     attn_post_bwd:  da_j = w_j dy (s_j - y);  ds_j = w_j dy;  dvf[idx] += ds_j   (w recomputed from lse)
//     attn_pre_bwd :  dq_i = sum_j du_j;  dkf[idx] -= du_j
// Layout: channels-last; a workgroup walks a chunk of points with one lane per channel, so every global
// access is a contiguous d*4-byte row (fully coalesced) and the softmax over the k neighbours is a
// per-lane online reduction (no cross-lane traffic at all).  All of it is HBM-bound byte movement:
// algorithmic bytes are 2-3 [R,d] tensors per kernel instead of the reference's ~10.
#include "common.h"
#include "prof.h"

namespace {

struct AttnShape {
  int B, n, N, k, d;  // n centres per shape, N source points per shape, k neighbours, d channels
  int qb;             // 1: q is one vector per shape, (B,1,d), shared by all centres (decoder)
};

constexpr int kPointsPerBlock = 16;

// u[b,i,j,c] = q[b,i,c] - kf[b,idx[b,i,j],c] + pos[b,i,j,c]
template <int T>
__global__ __launch_bounds__(T) void attn_pre_fwd_kernel(AttnShape s, const float *__restrict__ q,
                                                         const float *__restrict__ kf,
                                                         const float *__restrict__ pos,
                                                         const int32_t *__restrict__ idx,
                                                         float *__restrict__ u) {
  const int c = threadIdx.x;
  if (c >= s.d) return;
  const int b = blockIdx.y;
  const int i0 = blockIdx.x * kPointsPerBlock;
  const int i1 = min(s.n, i0 + kPointsPerBlock);
  const float *kfb = kf + static_cast<size_t>(b) * s.N * s.d;
  for (int i = i0; i < i1; ++i) {
    const size_t pt = static_cast<size_t>(b) * s.n + i;
    const float qv = s.qb ? q[static_cast<size_t>(b) * s.d + c] : q[pt * s.d + c];
    const int32_t *ip = idx + pt * s.k;
    const size_t row0 = pt * s.k;
    for (int j = 0; j < s.k; ++j) {
      const size_t e = (row0 + j) * s.d + c;
      u[e] = qv - kfb[static_cast<size_t>(ip[j]) * s.d + c] + pos[e];
    }
  }
}

// dq[b,i,c] = sum_j du[b,i,j,c];  dkf[b,idx,c] -= du   (dkf zero-filled by the host wrapper)
template <int T>
__global__ __launch_bounds__(T) void attn_pre_bwd_kernel(AttnShape s, const float *__restrict__ du,
                                                         const int32_t *__restrict__ idx,
                                                         float *__restrict__ dq, float *__restrict__ dkf) {
  const int c = threadIdx.x;
  if (c >= s.d) return;
  const int b = blockIdx.y;
  const int i0 = blockIdx.x * kPointsPerBlock;
  const int i1 = min(s.n, i0 + kPointsPerBlock);
  float *dkfb = dkf + static_cast<size_t>(b) * s.N * s.d;
  float qb_acc = 0.f;
  for (int i = i0; i < i1; ++i) {
    const size_t pt = static_cast<size_t>(b) * s.n + i;
    const int32_t *ip = idx + pt * s.k;
    const size_t row0 = pt * s.k;
    float acc = 0.f;
    for (int j = 0; j < s.k; ++j) {
      const float g = du[(row0 + j) * s.d + c];
      acc += g;
      atomicAdd(dkfb + static_cast<size_t>(ip[j]) * s.d + c, -g);
    }
    if (s.qb) qb_acc += acc;
    else dq[pt * s.d + c] = acc;
  }
  if (s.qb) atomicAdd(dq + static_cast<size_t>(b) * s.d + c, qb_acc);  // dq (B,1,d) zero-filled by the host
}

// y = sum_j softmax_j(a) (vf[idx] + pos) [+ w_g v_g] [+ residual];  lse = log-sum-exp of the logits.
// HAS_V = false: pos_only block (values = pos).  a_g / v_g: per-shape global token (decoder) or NULL.
template <int T, bool HAS_V>
__global__ __launch_bounds__(T) void attn_post_fwd_kernel(AttnShape s, const float *__restrict__ a,
                                                          const float *__restrict__ vf,
                                                          const float *__restrict__ pos,
                                                          const int32_t *__restrict__ idx,
                                                          const float *__restrict__ a_g,
                                                          const float *__restrict__ v_g,
                                                          const float *__restrict__ residual,
                                                          float *__restrict__ y, float *__restrict__ lse) {
  const int c = threadIdx.x;
  if (c >= s.d) return;
  const int b = blockIdx.y;
  const int i0 = blockIdx.x * kPointsPerBlock;
  const int i1 = min(s.n, i0 + kPointsPerBlock);
  const float *vfb = HAS_V ? vf + static_cast<size_t>(b) * s.N * s.d : nullptr;
  const bool has_g = a_g != nullptr;
  const float ag = has_g ? a_g[static_cast<size_t>(b) * s.d + c] : 0.f;
  const float vg = has_g ? v_g[static_cast<size_t>(b) * s.d + c] : 0.f;
  for (int i = i0; i < i1; ++i) {
    const size_t pt = static_cast<size_t>(b) * s.n + i;
    const int32_t *ip = idx + pt * s.k;
    const size_t row0 = pt * s.k;
    float m = has_g ? ag : -INFINITY;
    float l = has_g ? 1.f : 0.f;
    float acc = has_g ? vg : 0.f;
    for (int j = 0; j < s.k; ++j) {
      const size_t e = (row0 + j) * s.d + c;
      const float av = a[e];
      float sv = pos[e];
      if (HAS_V) sv += vfb[static_cast<size_t>(ip[j]) * s.d + c];
      const float mn = fmaxf(m, av);
      const float sc = __expf(m - mn);   // exp(-inf) = 0 on the first neighbour
      const float w = __expf(av - mn);
      l = l * sc + w;
      acc = acc * sc + w * sv;
      m = mn;
    }
    float out = acc / l;
    lse[pt * s.d + c] = m + __logf(l);
    if (residual) out += residual[pt * s.d + c];
    y[pt * s.d + c] = out;
  }
}

// Backward of attn_post_fwd for one upstream gradient dy (the residual branch is handled by autograd):
//   w_j = exp(a_j - lse);  s_j = vf[idx_j] + pos_j;  yb = y - residual (the attention output itself,
//   recovered from the saved forward output so that no second [B,n,d] tensor has to be kept)
//   da_j = w_j dy (s_j - yb);  ds_j = w_j dy  (written to dpos, scattered into dvf)
//   global token: da_g += sum_i w_g dy (v_g - yb), dv_g += sum_i w_g dy   (one atomic per block+channel)
template <int T, bool HAS_V>
__global__ __launch_bounds__(T) void attn_post_bwd_kernel(
    AttnShape s, const float *__restrict__ dy, const float *__restrict__ a, const float *__restrict__ vf,
    const float *__restrict__ pos, const int32_t *__restrict__ idx, const float *__restrict__ a_g,
    const float *__restrict__ v_g, const float *__restrict__ y, const float *__restrict__ residual,
    const float *__restrict__ lse, float *__restrict__ da, float *__restrict__ dpos, float *__restrict__ dvf, float *__restrict__ da_g,
    float *__restrict__ dv_g) {
  const int c = threadIdx.x;
  if (c >= s.d) return;
  const int b = blockIdx.y;
  const int i0 = blockIdx.x * kPointsPerBlock;
  const int i1 = min(s.n, i0 + kPointsPerBlock);
  const float *vfb = HAS_V ? vf + static_cast<size_t>(b) * s.N * s.d : nullptr;
  float *dvfb = HAS_V ? dvf + static_cast<size_t>(b) * s.N * s.d : nullptr;
  const bool has_g = a_g != nullptr;
  const float ag = has_g ? a_g[static_cast<size_t>(b) * s.d + c] : 0.f;
  const float vg = has_g ? v_g[static_cast<size_t>(b) * s.d + c] : 0.f;
  float dag_acc = 0.f, dvg_acc = 0.f;
  for (int i = i0; i < i1; ++i) {
    const size_t pt = static_cast<size_t>(b) * s.n + i;
    const int32_t *ip = idx + pt * s.k;
    const size_t row0 = pt * s.k;
    const float g = dy[pt * s.d + c];
    const float yb = y[pt * s.d + c] - (residual ? residual[pt * s.d + c] : 0.f);
    const float L = lse[pt * s.d + c];
    for (int j = 0; j < s.k; ++j) {
      const size_t e = (row0 + j) * s.d + c;
      const float w = __expf(a[e] - L);
      float sv = pos[e];
      if (HAS_V) sv += vfb[static_cast<size_t>(ip[j]) * s.d + c];
      const float ds = w * g;
      da[e] = ds * (sv - yb);
      dpos[e] = ds;
      if (HAS_V) atomicAdd(dvfb + static_cast<size_t>(ip[j]) * s.d + c, ds);
    }
    if (has_g) {
      const float ds = __expf(ag - L) * g;
      dag_acc += ds * (vg - yb);
      dvg_acc += ds;
    }
  }
  if (has_g) {
    atomicAdd(da_g + static_cast<size_t>(b) * s.d + c, dag_acc);
    atomicAdd(dv_g + static_cast<size_t>(b) * s.d + c, dvg_acc);
  }
}

inline bool shape_ok(const AttnShape &s) {
  return s.B > 0 && s.n > 0 && s.N > 0 && s.k > 0 && s.d > 0 && s.d <= 256 && s.B <= 65535;
}

inline double rows(const AttnShape &s) { return static_cast<double>(s.B) * s.n * s.k; }

}  // namespace

#define NSDP_ATTN_LAUNCH(KERNEL, ...)                                                      \
  do {                                                                                     \
    const dim3 grid(nsdp::ceil_div(s.n, kPointsPerBlock), s.B);                            \
    if (s.d <= 128) hipLaunchKernelGGL((KERNEL<128>), grid, dim3(128), 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<256>), grid, dim3(256), 0, st, __VA_ARGS__);            \
  } while (0)

#define NSDP_ATTN_LAUNCH_V(KERNEL, HASV, ...)                                                     \
  do {                                                                                            \
    const dim3 grid(nsdp::ceil_div(s.n, kPointsPerBlock), s.B);                                   \
    if (s.d <= 128) {                                                                             \
      if (HASV) hipLaunchKernelGGL((KERNEL<128, true>), grid, dim3(128), 0, st, __VA_ARGS__);     \
      else hipLaunchKernelGGL((KERNEL<128, false>), grid, dim3(128), 0, st, __VA_ARGS__);         \
    } else {                                                                                      \
      if (HASV) hipLaunchKernelGGL((KERNEL<256, true>), grid, dim3(256), 0, st, __VA_ARGS__);     \
      else hipLaunchKernelGGL((KERNEL<256, false>), grid, dim3(256), 0, st, __VA_ARGS__);         \
    }                                                                                             \
  } while (0)

extern "C" {

int nsdp_attn_pre_fwd(const float *q, const float *kf, const float *pos, const int32_t *idx, int B, int n,
                      int N, int k, int d, int q_per_shape, float *u, void *stream) {
  const AttnShape s{B, n, N, k, d, q_per_shape};
  if (static_cast<long long>(B) * n * k * d <= 0) return 0;
  NSDP_REQUIRE(shape_ok(s), "attn_pre_fwd: unsupported shape (d=%d must be <= 256)", d);
  NSDP_REQUIRE(q && kf && pos && idx && u, "attn_pre_fwd: null pointer");
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kAttnFwd, st, 0.0,
                          4.0 * (rows(s) * (2.0 * d + 1) + static_cast<double>(B) * (n + N) * d));
  NSDP_ATTN_LAUNCH(attn_pre_fwd_kernel, s, q, kf, pos, idx, u);
  return nsdp::launch_status("attn_pre_fwd_kernel");
}

int nsdp_attn_pre_bwd(const float *du, const int32_t *idx, int B, int n, int N, int k, int d,
                      int q_per_shape, float *dq, float *dkf, void *stream) {
  const AttnShape s{B, n, N, k, d, q_per_shape};
  hipStream_t st = nsdp::as_stream(stream);
  if (q_per_shape && dq && static_cast<long long>(B) * d > 0)
    NSDP_HIP_TRY(hipMemsetAsync(dq, 0, sizeof(float) * static_cast<size_t>(B) * d, st));
  if (dkf && static_cast<long long>(B) * N * d > 0)
    NSDP_HIP_TRY(hipMemsetAsync(dkf, 0, sizeof(float) * static_cast<size_t>(B) * N * d, st));
  if (static_cast<long long>(B) * n * k * d <= 0) return 0;
  NSDP_REQUIRE(shape_ok(s), "attn_pre_bwd: unsupported shape (d=%d must be <= 256)", d);
  NSDP_REQUIRE(du && idx && dq && dkf, "attn_pre_bwd: null pointer");
  nsdp::prof::Scope scope(nsdp::prof::kAttnBwd, st, 0.0,
                          4.0 * (rows(s) * (d + 1.0) + static_cast<double>(B) * (n + 2.0 * N) * d));
  NSDP_ATTN_LAUNCH(attn_pre_bwd_kernel, s, du, idx, dq, dkf);
  return nsdp::launch_status("attn_pre_bwd_kernel");
}

int nsdp_attn_post_fwd(const float *a, const float *vf, const float *pos, const int32_t *idx,
                       const float *a_g, const float *v_g, const float *residual, int B, int n, int N,
                       int k, int d, float *y, float *lse, void *stream) {
  const AttnShape s{B, n, N, k, d, 0};
  if (static_cast<long long>(B) * n * d <= 0) return 0;
  NSDP_REQUIRE(shape_ok(s), "attn_post_fwd: unsupported shape (d=%d must be <= 256)", d);
  NSDP_REQUIRE(a && pos && idx && y && lse, "attn_post_fwd: null pointer");
  NSDP_REQUIRE((a_g == nullptr) == (v_g == nullptr), "attn_post_fwd: a_g and v_g go together");
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kAttnFwd, st, 0.0,
                          4.0 * (rows(s) * (2.0 * d + 1) + static_cast<double>(B) * (2.0 * n + (vf ? N : 0)) * d));
  const bool has_v = vf != nullptr;
  NSDP_ATTN_LAUNCH_V(attn_post_fwd_kernel, has_v, s, a, vf, pos, idx, a_g, v_g, residual, y, lse);
  return nsdp::launch_status("attn_post_fwd_kernel");
}

int nsdp_attn_post_bwd(const float *dy, const float *a, const float *vf, const float *pos,
                       const int32_t *idx, const float *a_g, const float *v_g, const float *y,
                       const float *residual, const float *lse, int B, int n, int N, int k, int d, float *da, float *dpos,
                       float *dvf, float *da_g, float *dv_g, void *stream) {
  const AttnShape s{B, n, N, k, d, 0};
  hipStream_t st = nsdp::as_stream(stream);
  if (dvf && static_cast<long long>(B) * N * d > 0)
    NSDP_HIP_TRY(hipMemsetAsync(dvf, 0, sizeof(float) * static_cast<size_t>(B) * N * d, st));
  if (da_g && static_cast<long long>(B) * d > 0) {
    NSDP_HIP_TRY(hipMemsetAsync(da_g, 0, sizeof(float) * static_cast<size_t>(B) * d, st));
    NSDP_HIP_TRY(hipMemsetAsync(dv_g, 0, sizeof(float) * static_cast<size_t>(B) * d, st));
  }
  if (static_cast<long long>(B) * n * k * d <= 0) return 0;
  NSDP_REQUIRE(shape_ok(s), "attn_post_bwd: unsupported shape (d=%d must be <= 256)", d);
  NSDP_REQUIRE(dy && a && pos && idx && y && lse && da && dpos, "attn_post_bwd: null pointer");
  NSDP_REQUIRE((vf == nullptr) == (dvf == nullptr), "attn_post_bwd: vf and dvf go together");
  NSDP_REQUIRE((a_g == nullptr) == (da_g == nullptr) && (a_g == nullptr) == (v_g == nullptr) &&
                   (a_g == nullptr) == (dv_g == nullptr),
               "attn_post_bwd: global-token pointers go together");
  nsdp::prof::Scope scope(nsdp::prof::kAttnBwd, st, 0.0,
                          4.0 * (rows(s) * (4.0 * d + 1) + static_cast<double>(B) * (3.0 * n + (vf ? 2.0 * N : 0)) * d));
  const bool has_v = vf != nullptr;
  NSDP_ATTN_LAUNCH_V(attn_post_bwd_kernel, has_v, s, dy, a, vf, pos, idx, a_g, v_g, y, residual, lse, da, dpos,
                     dvf, da_g, dv_g);
  return nsdp::launch_status("attn_post_bwd_kernel");
}

}  // extern "C"
