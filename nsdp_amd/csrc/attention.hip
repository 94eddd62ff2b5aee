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
// Layout: channels-last; a lane owns 4 consecutive channels of one point (float4 accesses, whole rows
// coalesced) and walks the k neighbours, so the softmax over the neighbours is a per-lane online
// reduction (no cross-lane traffic at all).  All of it is HBM-bound byte movement:
// algorithmic bytes are 2-3 [R,d] tensors per kernel instead of the reference's ~10.
#include "common.h"
#include "prof.h"

namespace {

struct AttnShape {
  int B, n, N, k, d;  // n centres per shape, N source points per shape, k neighbours, d channels
  int qb;             // 1: q is one vector per shape, (B,1,d), shared by all centres (decoder)
  int iters;          // point groups walked per wave (fewer for large k so that the grid still fills the chip)
};

// Work decomposition shared by the four kernels: a lane owns 4 consecutive channels (one float4, so a
// wave-level access is up to 1 KiB -- 4x the bytes in flight of a dword-per-lane mapping), d/4 lanes form a
// point, floor(64 / (d/4)) points share a wave, and every wave walks kIters such point groups.
inline int iters_for(int k) { return k >= 64 ? 1 : (k >= 32 ? 2 : (k >= 16 ? 4 : 8)); }

struct Lane {
  bool active;
  int cq;        // channel quad: channels 4*cq .. 4*cq+3
  int sub;       // which of the wave's concurrent points
  int ppw;       // points per wave per iteration
  long long p0;  // first flattened point (b*n + i) of this wave
};

__device__ __forceinline__ Lane lane_setup(const AttnShape &s) {
  Lane L;
  const int lane = threadIdx.x & 63;
  const int lpp = s.d >> 2;
  L.ppw = 64 / lpp;
  L.sub = lane / lpp;
  L.cq = lane - L.sub * lpp;
  L.active = L.sub < L.ppw;
  const long long wave = static_cast<long long>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  L.p0 = wave * (static_cast<long long>(L.ppw) * s.iters);
  return L;
}

// The two backward kernels scatter with fp32 atomics.  With the float4 mapping one atomic instruction
// would touch 64 words spread over 1 KiB (7-8 cache lines, 16 B stride); there a lane instead owns the 4
// channels {cq, cq+lpp, cq+2lpp, cq+3lpp} ("strided quad"): every load/store/atomic instruction then
// covers one contiguous lpp*4-byte run (2 lines), with the same number of bytes in flight per lane.
struct Quad {
  float x, y, z, w;
};

// Storage type of the activation tensors: float, or bf16 (BASELINE config 3: bf16 storage, all arithmetic in fp32).
// Every accessor below exists for both; scatter targets (dkf, dvf, dq, global-token gradients) and the log-sum-exp
// are fp32 in either mode.
struct bf16_t {
  unsigned short v;
};
__device__ __forceinline__ float b2f(unsigned short h) { return __builtin_bit_cast(float, static_cast<unsigned>(h) << 16); }
__device__ __forceinline__ unsigned short f2b(float f) {          // round to nearest even
  using f32x2 = __attribute__((ext_vector_type(2))) float;
  using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
  return static_cast<unsigned short>(__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{f, 0.f}, bf16x2)) & 0xffffu);
}
__device__ __forceinline__ unsigned f2b2(float a, float b) {
  using f32x2 = __attribute__((ext_vector_type(2))) float;
  using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float ldf(const float *p) { return *p; }
__device__ __forceinline__ float ldf(const bf16_t *p) { return b2f(p->v); }
__device__ __forceinline__ void stf(float *p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16_t *p, float v) { p->v = f2b(v); }
__device__ __forceinline__ Quad ldq(const float *row, int cq, int lpp) {
  return Quad{row[cq], row[cq + lpp], row[cq + 2 * lpp], row[cq + 3 * lpp]};
}
__device__ __forceinline__ void stq(float *row, int cq, int lpp, Quad v) {
  row[cq] = v.x; row[cq + lpp] = v.y; row[cq + 2 * lpp] = v.z; row[cq + 3 * lpp] = v.w;
}
__device__ __forceinline__ Quad ldq(const bf16_t *row, int cq, int lpp) {
  return Quad{b2f(row[cq].v), b2f(row[cq + lpp].v), b2f(row[cq + 2 * lpp].v), b2f(row[cq + 3 * lpp].v)};
}
__device__ __forceinline__ void stq(bf16_t *row, int cq, int lpp, Quad v) {
  row[cq].v = f2b(v.x); row[cq + lpp].v = f2b(v.y); row[cq + 2 * lpp].v = f2b(v.z); row[cq + 3 * lpp].v = f2b(v.w);
}
__device__ __forceinline__ void atomic_addq(float *row, int cq, int lpp, Quad v) {
  atomicAdd(row + cq, v.x); atomicAdd(row + cq + lpp, v.y);
  atomicAdd(row + cq + 2 * lpp, v.z); atomicAdd(row + cq + 3 * lpp, v.w);
}
// Lane -> channel mapping of the two atomic (global scatter) backward kernels: the strided quad above for both storage
// types.  (Measured for bf16: contiguous quads -- one 8-byte access per tensor and lane instead of four 2-byte ones --
// make the loads cheaper but turn every fp32 atomic instruction into 64 words at a 16-byte stride: 1.15 ms instead of
// 0.40 ms on the 2048-point block.  The atomics, not the loads, bound these kernels.)
template <typename T> struct QuadMap { static constexpr bool kStrided = true; };
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float4 ld4(const bf16_t *p) {           // 4 channels = one 8-byte load
  const uint2 r = *reinterpret_cast<const uint2 *>(p);
  return make_float4(__builtin_bit_cast(float, r.x << 16), __builtin_bit_cast(float, r.x & 0xffff0000u),
                     __builtin_bit_cast(float, r.y << 16), __builtin_bit_cast(float, r.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(bf16_t *p, float4 v) {
  *reinterpret_cast<uint2 *>(p) = make_uint2(f2b2(v.x, v.y), f2b2(v.z, v.w));
}
__device__ __forceinline__ void atomic_add4(float *p, float4 v) {
  atomicAdd(p + 0, v.x); atomicAdd(p + 1, v.y); atomicAdd(p + 2, v.z); atomicAdd(p + 3, v.w);
}
// contiguous-quad accessors (lane owns channels 4cq..4cq+3) for the LDS-table kernels: global traffic is
// float4, the LDS adds tolerate the 4-way bank conflict (they are two orders of magnitude off the critical path)
__device__ __forceinline__ Quad ldc(const float *row, int cq) {
  const float4 v = *reinterpret_cast<const float4 *>(row + 4 * cq);
  return Quad{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void stc(float *row, int cq, Quad v) {
  *reinterpret_cast<float4 *>(row + 4 * cq) = make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ Quad ldc(const bf16_t *row, int cq) {
  const float4 v = ld4(row + 4 * cq);
  return Quad{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void stc(bf16_t *row, int cq, Quad v) { st4(row + 4 * cq, make_float4(v.x, v.y, v.z, v.w)); }
__device__ __forceinline__ void atomic_addc(float *row, int cq, Quad v) {
  atomicAdd(row + 4 * cq, v.x); atomicAdd(row + 4 * cq + 1, v.y);
  atomicAdd(row + 4 * cq + 2, v.z); atomicAdd(row + 4 * cq + 3, v.w);
}

template <bool ST, typename P>
__device__ __forceinline__ Quad ldQ(const P *row, int cq, int lpp) {
  if constexpr (ST) return ldq(row, cq, lpp);
  else return ldc(row, cq);
}
template <bool ST, typename P>
__device__ __forceinline__ void stQ(P *row, int cq, int lpp, Quad v) {
  if constexpr (ST) stq(row, cq, lpp, v);
  else stc(row, cq, v);
}
template <bool ST>
__device__ __forceinline__ void atQ(float *row, int cq, int lpp, Quad v) {
  if constexpr (ST) atomic_addq(row, cq, lpp, v);
  else atomic_addc(row, cq, v);
}

// u[b,i,j,c] = q[b,i,c] - kf[b,idx[b,i,j],c] + pos[b,i,j,c]
template <typename T>
__global__ __launch_bounds__(256) void attn_pre_fwd_kernel(AttnShape s, const T *__restrict__ q,
                                                           const T *__restrict__ kf,
                                                           const T *__restrict__ pos,
                                                           const int32_t *__restrict__ idx,
                                                           T *__restrict__ u) {
  const Lane L = lane_setup(s);
  if (!L.active) return;
  const long long total = static_cast<long long>(s.B) * s.n;
  for (int it = 0; it < s.iters; ++it) {
    const long long pt = L.p0 + static_cast<long long>(it) * L.ppw + L.sub;
    if (pt >= total) return;
    const int b = static_cast<int>(pt / s.n);
    const float4 qv = ld4(q + (s.qb ? static_cast<long long>(b) : pt) * s.d + 4 * L.cq);
    const T *kfb = kf + static_cast<long long>(b) * s.N * s.d + 4 * L.cq;
    const int32_t *ip = idx + pt * s.k;
    const long long e0 = pt * s.k * s.d + 4 * L.cq;
#pragma unroll 4
    for (int j = 0; j < s.k; ++j) {
      const float4 kv = ld4(kfb + static_cast<long long>(ip[j]) * s.d);
      const float4 pv = ld4(pos + e0 + static_cast<long long>(j) * s.d);
      st4(u + e0 + static_cast<long long>(j) * s.d,
          make_float4(qv.x - kv.x + pv.x, qv.y - kv.y + pv.y, qv.z - kv.z + pv.z, qv.w - kv.w + pv.w));
    }
  }
}

// dq[b,i,c] = sum_j du[b,i,j,c];  dkf[b,idx,c] -= du   (dkf / per-shape dq zero-filled by the host wrapper)
template <typename T>
__global__ __launch_bounds__(256) void attn_pre_bwd_kernel(AttnShape s, const T *__restrict__ du,
                                                           const int32_t *__restrict__ idx,
                                                           float *__restrict__ dq, float *__restrict__ dkf,
                                                           T *__restrict__ dpos_acc, const T *__restrict__ dq_sub = nullptr) {
  constexpr bool ST = QuadMap<T>::kStrided;
  const Lane L = lane_setup(s);
  if (!L.active) return;
  const int lpp = s.d >> 2, cq = L.cq;
  const long long total = static_cast<long long>(s.B) * s.n;
  Quad qb_acc{0.f, 0.f, 0.f, 0.f};
  int qb_b = -1;
  for (int it = 0; it < s.iters; ++it) {
    const long long pt = L.p0 + static_cast<long long>(it) * L.ppw + L.sub;
    if (pt >= total) break;
    const int b = static_cast<int>(pt / s.n);
    float *dkfb = dkf + static_cast<long long>(b) * s.N * s.d;
    const int32_t *ip = idx + pt * s.k;
    const T *dur = du + pt * s.k * s.d;
    T *par = dpos_acc ? dpos_acc + pt * s.k * s.d : nullptr;
    Quad acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int j = 0; j < s.k; ++j) {
      const Quad g = ldQ<ST>(dur + static_cast<long long>(j) * s.d, cq, lpp);
      acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
      if (dkf) atQ<ST>(dkfb + static_cast<long long>(ip[j]) * s.d, cq, lpp, Quad{-g.x, -g.y, -g.z, -g.w});
      if (par) {   // d(pos) += du while the row is in registers (saves the separate add kernel's re-read of du)
        T *pr = par + static_cast<long long>(j) * s.d;
        const Quad o = ldQ<ST>(pr, cq, lpp);
        stQ<ST>(pr, cq, lpp, Quad{o.x + g.x, o.y + g.y, o.z + g.z, o.w + g.w});
      }
    }
    if (s.qb) {
      if (b != qb_b) {
        if (qb_b >= 0) atQ<ST>(dq + static_cast<long long>(qb_b) * s.d, cq, lpp, qb_acc);
        qb_acc = Quad{0.f, 0.f, 0.f, 0.f};
        qb_b = b;
      }
      qb_acc.x += acc.x; qb_acc.y += acc.y; qb_acc.z += acc.z; qb_acc.w += acc.w;
    } else {
      if (dq_sub) {      // (the caller's next statement was `dq -= dq_sub`)
        const Quad o = ldQ<ST>(dq_sub + pt * s.d, cq, lpp);
        acc.x -= o.x; acc.y -= o.y; acc.z -= o.z; acc.w -= o.w;
      }
      stQ<ST>(dq + pt * s.d, cq, lpp, acc);
    }
  }
  if (s.qb && qb_b >= 0) atQ<ST>(dq + static_cast<long long>(qb_b) * s.d, cq, lpp, qb_acc);
}

#define NSDP_ONLINE_STEP(C)                      \
  {                                              \
    const float mn = fmaxf(m.C, av.C);           \
    const float sc = __expf(m.C - mn);           \
    const float w = __expf(av.C - mn);           \
    l.C = l.C * sc + w;                          \
    acc.C = acc.C * sc + w * sv.C;               \
    m.C = mn;                                    \
  }

// y = sum_j softmax_j(a) (vf[idx] + pos) [+ w_g v_g] [+ residual];  lse = log-sum-exp of the logits.
// HAS_V = false: pos_only block (values = pos).  a_g / v_g: per-shape global token (decoder) or NULL.
template <typename T, bool HAS_V>
__global__ __launch_bounds__(256) void attn_post_fwd_kernel(AttnShape s, const T *__restrict__ a,
                                                            const T *__restrict__ vf,
                                                            const T *__restrict__ pos,
                                                            const int32_t *__restrict__ idx,
                                                            const T *__restrict__ a_g,
                                                            const T *__restrict__ v_g,
                                                            const T *__restrict__ residual,
                                                            T *__restrict__ y, float *__restrict__ lse,
                                                            const T *__restrict__ qsub = nullptr) {
  const Lane L = lane_setup(s);
  if (!L.active) return;
  const long long total = static_cast<long long>(s.B) * s.n;
  const bool has_g = a_g != nullptr;
  for (int it = 0; it < s.iters; ++it) {
    const long long pt = L.p0 + static_cast<long long>(it) * L.ppw + L.sub;
    if (pt >= total) return;
    const int b = static_cast<int>(pt / s.n);
    const T *vfb = HAS_V ? vf + static_cast<long long>(b) * s.N * s.d + 4 * L.cq : nullptr;
    const int32_t *ip = idx + pt * s.k;
    const long long e0 = pt * s.k * s.d + 4 * L.cq;
    // (qsub: `pos` holds u = q_i - k_j + pos and `vf` the table v + k -- the values are then u + (v + k)[idx] - q_i; see
    // nsdp_linear_bf16x3_gather_f32)
    const float4 qs = (HAS_V && qsub) ? ld4(qsub + pt * s.d + 4 * L.cq) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 m, l, acc;
    if (has_g) {
      m = ld4(a_g + static_cast<long long>(b) * s.d + 4 * L.cq);
      acc = ld4(v_g + static_cast<long long>(b) * s.d + 4 * L.cq);
      l = make_float4(1.f, 1.f, 1.f, 1.f);
    } else {
      m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      l = make_float4(0.f, 0.f, 0.f, 0.f);
      acc = l;
    }
#pragma unroll 4
    for (int j = 0; j < s.k; ++j) {
      const float4 av = ld4(a + e0 + static_cast<long long>(j) * s.d);
      float4 sv = ld4(pos + e0 + static_cast<long long>(j) * s.d);
      if (HAS_V) {
        const float4 vv = ld4(vfb + static_cast<long long>(ip[j]) * s.d);
        sv.x += vv.x - qs.x; sv.y += vv.y - qs.y; sv.z += vv.z - qs.z; sv.w += vv.w - qs.w;
      }
      NSDP_ONLINE_STEP(x) NSDP_ONLINE_STEP(y) NSDP_ONLINE_STEP(z) NSDP_ONLINE_STEP(w)
    }
    float4 out = make_float4(acc.x / l.x, acc.y / l.y, acc.z / l.z, acc.w / l.w);
    st4(lse + pt * s.d + 4 * L.cq,
        make_float4(m.x + __logf(l.x), m.y + __logf(l.y), m.z + __logf(l.z), m.w + __logf(l.w)));
    if (residual) {
      const float4 r = ld4(residual + pt * s.d + 4 * L.cq);
      out.x += r.x; out.y += r.y; out.z += r.z; out.w += r.w;
    }
    st4(y + pt * s.d + 4 * L.cq, out);
  }
}

// Backward of attn_post_fwd for one upstream gradient dy (the residual branch is handled by autograd):
//   w_j = exp(a_j - lse);  s_j = vf[idx_j] + pos_j;  yb = y - residual (the attention output itself,
//   recovered from the saved forward output so that no second [B,n,d] tensor has to be kept)
//   da_j = w_j dy (s_j - yb);  ds_j = w_j dy  (written to dpos, scattered into dvf)
//   global token: da_g += sum_i w_g dy (v_g - yb), dv_g += sum_i w_g dy   (register partial sums over the
//   wave's points, one atomic per lane and shape)
// ST: strided-quad lane -> channel mapping (needed when the kernel scatters with atomics); false = contiguous quads (one
// 8 / 16-byte access per tensor and lane): the pure-stream form used when the caller scatters d(pos) itself
template <typename T, bool HAS_V, bool ST = true>
__global__ __launch_bounds__(256) void attn_post_bwd_kernel(
    AttnShape s, const T *__restrict__ dy, const T *__restrict__ a, const T *__restrict__ vf,
    const T *__restrict__ pos, const int32_t *__restrict__ idx, const T *__restrict__ a_g,
    const T *__restrict__ v_g, const T *__restrict__ y, const T *__restrict__ residual,
    const float *__restrict__ lse, T *__restrict__ da, T *__restrict__ dpos,
    float *__restrict__ dvf, float *__restrict__ da_g, float *__restrict__ dv_g, const T *__restrict__ qsub = nullptr) {
  const Lane L = lane_setup(s);
  if (!L.active) return;
  const int lpp = s.d >> 2, cq = L.cq;
  const long long total = static_cast<long long>(s.B) * s.n;
  const bool has_g = a_g != nullptr;
  Quad dag{0.f, 0.f, 0.f, 0.f}, dvg{0.f, 0.f, 0.f, 0.f};
  int gb = -1;
  for (int it = 0; it < s.iters; ++it) {
    const long long pt = L.p0 + static_cast<long long>(it) * L.ppw + L.sub;
    if (pt >= total) break;
    const int b = static_cast<int>(pt / s.n);
    const T *vfb = HAS_V ? vf + static_cast<long long>(b) * s.N * s.d : nullptr;
    float *dvfb = (HAS_V && dvf) ? dvf + static_cast<long long>(b) * s.N * s.d : nullptr;   // null: scatter done elsewhere
    const int32_t *ip = idx + pt * s.k;
    const long long r0 = pt * s.k * s.d;
    const Quad g = ldQ<ST>(dy + pt * s.d, cq, lpp);
    Quad yb = ldQ<ST>(y + pt * s.d, cq, lpp);
    if (residual) {
      const Quad r = ldQ<ST>(residual + pt * s.d, cq, lpp);
      yb.x -= r.x; yb.y -= r.y; yb.z -= r.z; yb.w -= r.w;
    }
    const Quad Lse = ldQ<ST>(lse + pt * s.d, cq, lpp);
    if (HAS_V && qsub) {      // the attention output without the "- q_i" of the values (see attn_post_fwd_kernel): yb + q_i
      const Quad qs = ldQ<ST>(qsub + pt * s.d, cq, lpp);
      yb.x += qs.x; yb.y += qs.y; yb.z += qs.z; yb.w += qs.w;
    }
#pragma unroll 4
    for (int j = 0; j < s.k; ++j) {
      const long long rj = r0 + static_cast<long long>(j) * s.d;
      const Quad av = ldQ<ST>(a + rj, cq, lpp);
      Quad sv = ldQ<ST>(pos + rj, cq, lpp);
      if (HAS_V) {
        const Quad vv = ldQ<ST>(vfb + static_cast<long long>(ip[j]) * s.d, cq, lpp);
        sv.x += vv.x; sv.y += vv.y; sv.z += vv.z; sv.w += vv.w;
      }
      const Quad ds{__expf(av.x - Lse.x) * g.x, __expf(av.y - Lse.y) * g.y, __expf(av.z - Lse.z) * g.z,
                    __expf(av.w - Lse.w) * g.w};
      stQ<ST>(da + rj, cq, lpp,
          Quad{ds.x * (sv.x - yb.x), ds.y * (sv.y - yb.y), ds.z * (sv.z - yb.z), ds.w * (sv.w - yb.w)});
      stQ<ST>(dpos + rj, cq, lpp, ds);
      if (HAS_V && dvfb) atQ<ST>(dvfb + static_cast<long long>(ip[j]) * s.d, cq, lpp, ds);
    }
    if (has_g) {
      if (b != gb) {
        if (gb >= 0) {
          atQ<ST>(da_g + static_cast<long long>(gb) * s.d, cq, lpp, dag);
          atQ<ST>(dv_g + static_cast<long long>(gb) * s.d, cq, lpp, dvg);
        }
        dag = Quad{0.f, 0.f, 0.f, 0.f};
        dvg = dag;
        gb = b;
      }
      const Quad ag = ldQ<ST>(a_g + static_cast<long long>(b) * s.d, cq, lpp);
      const Quad vg = ldQ<ST>(v_g + static_cast<long long>(b) * s.d, cq, lpp);
      const Quad ds{__expf(ag.x - Lse.x) * g.x, __expf(ag.y - Lse.y) * g.y, __expf(ag.z - Lse.z) * g.z,
                    __expf(ag.w - Lse.w) * g.w};
      dag.x += ds.x * (vg.x - yb.x); dag.y += ds.y * (vg.y - yb.y);
      dag.z += ds.z * (vg.z - yb.z); dag.w += ds.w * (vg.w - yb.w);
      dvg.x += ds.x; dvg.y += ds.y; dvg.z += ds.z; dvg.w += ds.w;
    }
  }
  if (has_g && gb >= 0) {
    atQ<ST>(da_g + static_cast<long long>(gb) * s.d, cq, lpp, dag);
    atQ<ST>(dv_g + static_cast<long long>(gb) * s.d, cq, lpp, dvg);
  }
}

// ------------------------------------------------------------------------------------------------
// LDS-privatised scatter variants of the two backward kernels, used when the scatter target of one shape
// (N x d floats) fits in LDS -- the decoder (100 anchors x 200) and the 100-anchor encoder blocks.  A
// workgroup of 8 waves owns a chunk of centres of ONE shape, accumulates the scattered gradient into an
// LDS table with ds_add_f32 (conflict-free: consecutive lanes = consecutive channels) and flushes the
// table with one global atomic per entry: ~n/chunk times fewer global atomics on the few hot rows that
// every query of a shape hits (the decoder has 57 k (query, neighbour) pairs per anchor row).
// ------------------------------------------------------------------------------------------------
constexpr int kLdsThreads = 512;

struct BlockLane {
  bool active;
  int cq, sub, ppw;
  long long p0, pend;  // first centre of this wave and end of this workgroup's range inside shape blockIdx.y
  int stride;          // centres advanced per iteration (all 8 waves move together)
};

__device__ __forceinline__ BlockLane block_lane_setup(const AttnShape &s) {
  BlockLane L;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lpp = s.d >> 2;
  L.ppw = 64 / lpp;
  L.sub = lane / lpp;
  L.cq = lane - L.sub * lpp;
  L.active = L.sub < L.ppw;
  const int per_iter = L.ppw * (kLdsThreads / 64);
  const long long base = static_cast<long long>(blockIdx.y) * s.n;
  const long long i0 = static_cast<long long>(blockIdx.x) * per_iter * s.iters + static_cast<long long>(wave) * L.ppw;
  L.p0 = base + i0;                                  // iteration `it` adds it * per_iter (see stride below)
  const long long blk_end = static_cast<long long>(blockIdx.x + 1) * per_iter * s.iters;
  L.pend = base + (blk_end < s.n ? blk_end : s.n);
  L.stride = per_iter;
  return L;
}

template <typename T>
__global__ __launch_bounds__(kLdsThreads) void attn_pre_bwd_lds_kernel(AttnShape s, const T *__restrict__ du,
                                                                       const int32_t *__restrict__ idx,
                                                                       float *__restrict__ dq,
                                                                       float *__restrict__ dkf,
                                                                       T *__restrict__ dpos_acc) {
  extern __shared__ __attribute__((aligned(16))) float table[];  // [N][d] partial of -sum du
  const int b = blockIdx.y;
  const int tsz = s.N * s.d;
  for (int e = threadIdx.x; e < tsz; e += kLdsThreads) table[e] = 0.f;
  __syncthreads();
  const BlockLane L = block_lane_setup(s);
  const int cq = L.cq;
  Quad qb_acc{0.f, 0.f, 0.f, 0.f};
  if (L.active) {
    for (int it = 0; it < s.iters; ++it) {
      const long long pt = L.p0 + static_cast<long long>(it) * L.stride + L.sub;
      if (pt >= L.pend) break;
      const int32_t *ip = idx + pt * s.k;
      const T *dur = du + pt * s.k * s.d;
      T *par = dpos_acc ? dpos_acc + pt * s.k * s.d : nullptr;
      Quad acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int j = 0; j < s.k; ++j) {
        const Quad g = ldc(dur + static_cast<long long>(j) * s.d, cq);
        acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
        atomic_addc(table + ip[j] * s.d, cq, Quad{-g.x, -g.y, -g.z, -g.w});
        if (par) {
          T *pr = par + static_cast<long long>(j) * s.d;
          const Quad o = ldc(pr, cq);
          stc(pr, cq, Quad{o.x + g.x, o.y + g.y, o.z + g.z, o.w + g.w});
        }
      }
      if (s.qb) {
        qb_acc.x += acc.x; qb_acc.y += acc.y; qb_acc.z += acc.z; qb_acc.w += acc.w;
      } else {
        stc(dq + pt * s.d, cq, acc);
      }
    }
    if (s.qb) atomic_addc(dq + static_cast<long long>(b) * s.d, cq, qb_acc);
  }
  __syncthreads();
  float *out = dkf + static_cast<long long>(b) * tsz;
  for (int e = threadIdx.x; e < tsz; e += kLdsThreads) {
    const float v = table[e];
    if (v != 0.f) atomicAdd(out + e, v);
  }
}

template <typename T, bool HAS_V>
__global__ __launch_bounds__(kLdsThreads) void attn_post_bwd_lds_kernel(
    AttnShape s, const T *__restrict__ dy, const T *__restrict__ a, const T *__restrict__ vf,
    const T *__restrict__ pos, const int32_t *__restrict__ idx, const T *__restrict__ a_g,
    const T *__restrict__ v_g, const T *__restrict__ y, const T *__restrict__ residual,
    const float *__restrict__ lse, T *__restrict__ da, T *__restrict__ dpos,
    float *__restrict__ dvf, float *__restrict__ da_g, float *__restrict__ dv_g) {
  extern __shared__ __attribute__((aligned(16))) float table[];  // [N][d] partial of dvf
  const int b = blockIdx.y;
  const int tsz = s.N * s.d;
  if (HAS_V) {
    for (int e = threadIdx.x; e < tsz; e += kLdsThreads) table[e] = 0.f;
    __syncthreads();
  }
  const BlockLane L = block_lane_setup(s);
  const int cq = L.cq;
  const bool has_g = a_g != nullptr;
  if (L.active) {
    const T *vfb = HAS_V ? vf + static_cast<long long>(b) * tsz : nullptr;
    Quad dag{0.f, 0.f, 0.f, 0.f}, dvg{0.f, 0.f, 0.f, 0.f};
    Quad ag{0.f, 0.f, 0.f, 0.f}, vg{0.f, 0.f, 0.f, 0.f};
    if (has_g) {
      ag = ldc(a_g + static_cast<long long>(b) * s.d, cq);
      vg = ldc(v_g + static_cast<long long>(b) * s.d, cq);
    }
    for (int it = 0; it < s.iters; ++it) {
      const long long pt = L.p0 + static_cast<long long>(it) * L.stride + L.sub;
      if (pt >= L.pend) break;
      const int32_t *ip = idx + pt * s.k;
      const long long r0 = pt * s.k * s.d;
      const Quad g = ldc(dy + pt * s.d, cq);
      Quad yb = ldc(y + pt * s.d, cq);
      if (residual) {
        const Quad r = ldc(residual + pt * s.d, cq);
        yb.x -= r.x; yb.y -= r.y; yb.z -= r.z; yb.w -= r.w;
      }
      const Quad Lse = ldc(lse + pt * s.d, cq);
#pragma unroll 2
      for (int j = 0; j < s.k; ++j) {
        const long long rj = r0 + static_cast<long long>(j) * s.d;
        const Quad av = ldc(a + rj, cq);
        Quad sv = ldc(pos + rj, cq);
        if (HAS_V) {
          const Quad vv = ldc(vfb + static_cast<long long>(ip[j]) * s.d, cq);
          sv.x += vv.x; sv.y += vv.y; sv.z += vv.z; sv.w += vv.w;
        }
        const Quad ds{__expf(av.x - Lse.x) * g.x, __expf(av.y - Lse.y) * g.y, __expf(av.z - Lse.z) * g.z,
                      __expf(av.w - Lse.w) * g.w};
        stc(da + rj, cq,
            Quad{ds.x * (sv.x - yb.x), ds.y * (sv.y - yb.y), ds.z * (sv.z - yb.z), ds.w * (sv.w - yb.w)});
        stc(dpos + rj, cq, ds);
        if (HAS_V) atomic_addc(table + ip[j] * s.d, cq, ds);
      }
      if (has_g) {
        const Quad ds{__expf(ag.x - Lse.x) * g.x, __expf(ag.y - Lse.y) * g.y, __expf(ag.z - Lse.z) * g.z,
                      __expf(ag.w - Lse.w) * g.w};
        dag.x += ds.x * (vg.x - yb.x); dag.y += ds.y * (vg.y - yb.y);
        dag.z += ds.z * (vg.z - yb.z); dag.w += ds.w * (vg.w - yb.w);
        dvg.x += ds.x; dvg.y += ds.y; dvg.z += ds.z; dvg.w += ds.w;
      }
    }
    if (has_g) {
      atomic_addc(da_g + static_cast<long long>(b) * s.d, cq, dag);
      atomic_addc(dv_g + static_cast<long long>(b) * s.d, cq, dvg);
    }
  }
  if (HAS_V) {
    __syncthreads();
    float *out = dvf + static_cast<long long>(b) * tsz;
    for (int e = threadIdx.x; e < tsz; e += kLdsThreads) {
      const float v = table[e];
      if (v != 0.f) atomicAdd(out + e, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Register-table scatter: table[b][idx[r]][c] += sign * src[r][c] over the rows r of shape b (and optionally the
// column sum of src), for a table of at most 128 rows x 256 channels per shape (the decoder's 100 anchors).
// One lane owns one channel and keeps its column of the table -- 128 floats -- in four 32-register vectors that
// are indexed dynamically (s_set_gpr_idx): no LDS, no atomics in the loop (LDS float atomics retire ~1 lane per
// clock and made the LDS-table kernels 5x slower than their traffic).  All four vectors are updated
// unconditionally with a masked addend: a switch over the vector would make the tables loop-carried phis and
// hipcc then copies whole vectors around.  Rows are fetched UNROLL at a time (one 4-byte scalar index and one
// coalesced d*4-byte row each); the table is flushed with global atomics once per workgroup.
// ------------------------------------------------------------------------------------------------
typedef float f32x32_t __attribute__((ext_vector_type(32)));

template <typename T, int UNROLL>
__global__ __launch_bounds__(256) void scatter_rows_regtab_kernel(const T *__restrict__ src,
                                                                  const int32_t *__restrict__ idx,
                                                                  float *__restrict__ table, float *__restrict__ colsum,
                                                                  T *__restrict__ acc_rows,
                                                                  long long rows_per_shape, long long rows_per_wg, int N,
                                                                  int d, float sign, float colsum_sign) {
  f32x32_t t0 = {}, t1 = {}, t2 = {}, t3 = {};
  const int c = threadIdx.x, b = blockIdx.y;
  const bool cv = c < d;
  const long long begin = static_cast<long long>(blockIdx.x) * rows_per_wg;
  long long end = begin + rows_per_wg;
  end = end < rows_per_shape ? end : rows_per_shape;
  const long long base = static_cast<long long>(b) * rows_per_shape;
  const T *p = src + base * d + (cv ? c : 0);
  T *pa = acc_rows ? acc_rows + base * d + (cv ? c : 0) : nullptr;      // acc_rows[row] += src[row] (optional)
  const int32_t *ip = idx + base;
  float total = 0.f;
  for (long long r = begin; r < end; r += UNROLL) {
    float x[UNROLL], y[UNROLL];
    int a[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long long rr = r + u < end ? r + u : end - 1;          // clamped: the tail re-reads the last row, masked below
      x[u] = ldf(p + rr * d);
      y[u] = pa ? ldf(pa + rr * d) : 0.f;
      a[u] = __builtin_amdgcn_readfirstlane(ip[rr]);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int i = a[u] & 31, g = a[u] >> 5;
      const float v = (cv && r + u < end) ? x[u] : 0.f;
      if (pa && cv && r + u < end) stf(pa + (r + u) * d, y[u] + x[u]);
      total += v;
      t0[i] += g == 0 ? v : 0.f;
      t1[i] += g == 1 ? v : 0.f;
      t2[i] += g == 2 ? v : 0.f;
      t3[i] += g == 3 ? v : 0.f;
    }
  }
  if (cv) {
    float *o = table + static_cast<long long>(b) * N * d + c;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if (i < N && t0[i] != 0.f) atomicAdd(o + static_cast<long long>(i) * d, sign * t0[i]);
      if (32 + i < N && t1[i] != 0.f) atomicAdd(o + static_cast<long long>(32 + i) * d, sign * t1[i]);
      if (64 + i < N && t2[i] != 0.f) atomicAdd(o + static_cast<long long>(64 + i) * d, sign * t2[i]);
      if (96 + i < N && t3[i] != 0.f) atomicAdd(o + static_cast<long long>(96 + i) * d, sign * t3[i]);
    }
    if (colsum) atomicAdd(colsum + static_cast<long long>(b) * d + c, colsum_sign * total);
  }
}

inline bool regtab_fits(int N, int d, long long rows_per_shape) { return N <= 128 && d <= 256 && rows_per_shape >= 4096; }

// table (zero-filled by the caller) += sign * scatter(src); colsum (zero-filled, may be null) += colsum_sign * sum_r src[r]
template <typename T>
inline int launch_regtab_scatter(const T *src, const int32_t *idx, float *table, float *colsum, T *acc_rows,
                                 int B, long long rows_per_shape, int N, int d, float sign, float colsum_sign,
                                 hipStream_t st) {
  long long wgs = (4LL * nsdp::num_cus() + B - 1) / B;                 // ~4 workgroups per CU in total
  if (wgs * 512 > rows_per_shape) wgs = rows_per_shape / 512 > 0 ? rows_per_shape / 512 : 1;
  long long per = (rows_per_shape + wgs - 1) / wgs;
  per = (per + 15) / 16 * 16;
  wgs = (rows_per_shape + per - 1) / per;
  NSDP_TRACE("scatter_rows_regtab<8>");
  // (a bf16 row is half the bytes per load instruction: twice the rows in flight per lane)
  constexpr int kUnroll = sizeof(T) == 2 ? 16 : 8;
  hipLaunchKernelGGL((scatter_rows_regtab_kernel<T, kUnroll>), dim3(static_cast<unsigned>(wgs), B), dim3(256), 0, st, src, idx,
                     table, colsum, acc_rows, rows_per_shape, per, N, d, sign, colsum_sign);
  return nsdp::launch_status("scatter_rows_regtab_kernel");
}


// ------------------------------------------------------------------------------------------------
// Deterministic pure-stream form of attn_post_bwd for the blocks whose scatter is done elsewhere (the decoder: dvf comes out
// of nsdp_scatter_rows_onehot_*): da and dpos as above, and the global-token gradients WITHOUT atomics -- a workgroup owns a
// range of centres of ONE shape (grid.y), its waves' partial sums are combined through LDS in a fixed order and written to
// gpart[b][blockIdx.x][{da_g, dv_g}][d]; global_token_reduce_kernel adds a shape's partials in order.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_post_bwd_shape_kernel(
    AttnShape s, const T *__restrict__ dy, const T *__restrict__ a, const T *__restrict__ vf,
    const T *__restrict__ pos, const int32_t *__restrict__ idx, const T *__restrict__ a_g,
    const T *__restrict__ v_g, const T *__restrict__ y, const T *__restrict__ residual,
    const float *__restrict__ lse, T *__restrict__ da, T *__restrict__ dpos, float *__restrict__ gpart) {
  __shared__ __attribute__((aligned(16))) float red[4 * 2 * 256];      // [wave][sub][{dag, dvg}][lpp quads]: <= 4 x 2 x 256 floats
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lpp = s.d >> 2, ppw = 64 / lpp;
  const int sub = lane / lpp, cq = lane - sub * lpp;
  const bool active = sub < ppw;
  const int per_iter = ppw * 4;
  const long long base = static_cast<long long>(b) * s.n;
  const long long i0 = static_cast<long long>(blockIdx.x) * per_iter * s.iters + wave * ppw;
  const long long blk_end = static_cast<long long>(blockIdx.x + 1) * per_iter * s.iters;
  const long long pend = base + (blk_end < s.n ? blk_end : s.n);
  const bool has_g = a_g != nullptr;
  const long long tsz = static_cast<long long>(s.N) * s.d;
  Quad dag{0.f, 0.f, 0.f, 0.f}, dvg{0.f, 0.f, 0.f, 0.f};
  if (active) {
    const T *vfb = vf + static_cast<long long>(b) * tsz;
    Quad ag{0.f, 0.f, 0.f, 0.f}, vg{0.f, 0.f, 0.f, 0.f};
    if (has_g) {
      ag = ldc(a_g + static_cast<long long>(b) * s.d, cq);
      vg = ldc(v_g + static_cast<long long>(b) * s.d, cq);
    }
    for (int it = 0; it < s.iters; ++it) {
      const long long pt = base + i0 + static_cast<long long>(it) * per_iter + sub;
      if (pt >= pend) break;
      const int32_t *ip = idx + pt * s.k;
      const long long r0 = pt * s.k * s.d;
      const Quad g = ldc(dy + pt * s.d, cq);
      Quad yb = ldc(y + pt * s.d, cq);
      if (residual) {
        const Quad r = ldc(residual + pt * s.d, cq);
        yb.x -= r.x; yb.y -= r.y; yb.z -= r.z; yb.w -= r.w;
      }
      const Quad Lse = ldc(lse + pt * s.d, cq);
#pragma unroll 2
      for (int j = 0; j < s.k; ++j) {
        const long long rj = r0 + static_cast<long long>(j) * s.d;
        const Quad av = ldc(a + rj, cq);
        Quad sv = ldc(pos + rj, cq);
        const Quad vv = ldc(vfb + static_cast<long long>(ip[j]) * s.d, cq);
        sv.x += vv.x; sv.y += vv.y; sv.z += vv.z; sv.w += vv.w;
        const Quad ds{__expf(av.x - Lse.x) * g.x, __expf(av.y - Lse.y) * g.y, __expf(av.z - Lse.z) * g.z,
                      __expf(av.w - Lse.w) * g.w};
        stc(da + rj, cq,
            Quad{ds.x * (sv.x - yb.x), ds.y * (sv.y - yb.y), ds.z * (sv.z - yb.z), ds.w * (sv.w - yb.w)});
        stc(dpos + rj, cq, ds);
      }
      if (has_g) {
        const Quad ds{__expf(ag.x - Lse.x) * g.x, __expf(ag.y - Lse.y) * g.y, __expf(ag.z - Lse.z) * g.z,
                      __expf(ag.w - Lse.w) * g.w};
        dag.x += ds.x * (vg.x - yb.x); dag.y += ds.y * (vg.y - yb.y);
        dag.z += ds.z * (vg.z - yb.z); dag.w += ds.w * (vg.w - yb.w);
        dvg.x += ds.x; dvg.y += ds.y; dvg.z += ds.z; dvg.w += ds.w;
      }
    }
  }
  if (!has_g) return;      // (uniform)
  // slot (wave, sub) -> red[(wave * ppw + sub) * 2 + {0, 1}][4 lpp floats]; every slot is written (zeros where no centre)
  if (active) {
    float *r0 = red + static_cast<size_t>((wave * ppw + sub) * 2) * (4 * lpp);
    stc(r0, cq, dag);
    stc(r0 + 4 * lpp, cq, dvg);
  }
  __syncthreads();
  const int slots = 4 * ppw;
  for (int e = threadIdx.x; e < 2 * s.d; e += 256) {
    const int which = e / s.d, c = e - which * s.d;
    float t = 0.f;
    for (int sl = 0; sl < slots; ++sl) t += red[static_cast<size_t>(sl * 2 + which) * (4 * lpp) + c];
    gpart[(static_cast<long long>(b) * gridDim.x + blockIdx.x) * (2 * s.d) + e] = t;
  }
}

// da_g[b][c] = sum over the S block partials of shape b (fixed order, 4 chains); dv_g likewise
__global__ __launch_bounds__(256) void global_token_reduce_kernel(const float *__restrict__ gpart, int S, int d,
                                                                  float *__restrict__ da_g, float *__restrict__ dv_g) {
  const int b = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 2 * d) return;
  const float *w = gpart + static_cast<long long>(b) * S * (2 * d) + e;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int sidx = 0;
  for (; sidx + 4 <= S; sidx += 4) {
    a0 += w[static_cast<long long>(sidx) * 2 * d]; a1 += w[static_cast<long long>(sidx + 1) * 2 * d];
    a2 += w[static_cast<long long>(sidx + 2) * 2 * d]; a3 += w[static_cast<long long>(sidx + 3) * 2 * d];
  }
  for (; sidx < S; ++sidx) a0 += w[static_cast<long long>(sidx) * 2 * d];
  const float t = (a0 + a1) + (a2 + a3);
  if (e < d) da_g[static_cast<long long>(b) * d + e] = t;
  else dv_g[static_cast<long long>(b) * d + (e - d)] = t;
}

// centres per workgroup / workgroups per shape of the deterministic stream form: ~4096 workgroups in total
inline void shape_plan(AttnShape &s, dim3 &grid) {
  const int ppw = 64 / (s.d >> 2), per_iter = ppw * 4;
  long long iters = (static_cast<long long>(s.B) * s.n + 4096LL * per_iter - 1) / (4096LL * per_iter);
  if (iters < iters_for(s.k)) iters = iters_for(s.k);
  const long long max_iters = (s.n + per_iter - 1) / per_iter;
  if (iters > max_iters) iters = max_iters;
  s.iters = static_cast<int>(iters);
  grid = dim3(static_cast<unsigned>((s.n + iters * per_iter - 1) / (iters * per_iter)), s.B);
}

inline bool lds_table_fits(const AttnShape &s) {
  return static_cast<long long>(s.N) * s.d * 4 <= 110 * 1024 && s.B <= 65535 && s.n >= 4 * s.N;
}

// The table flush costs N*d global atomics per workgroup, so the LDS variants use FEW, LONG workgroups:
// about 512 in total (two per CU would not fit in LDS anyway), each walking ~n*B/512 centres.
inline void lds_plan(AttnShape &s, dim3 &grid) {
  const int ppw = 64 / (s.d >> 2);
  const int per_iter = ppw * (kLdsThreads / 64);             // centres per workgroup iteration
  int blocks_per_shape = 512 / s.B;
  if (blocks_per_shape < 1) blocks_per_shape = 1;
  const int max_blocks = (s.n + per_iter - 1) / per_iter;
  if (blocks_per_shape > max_blocks) blocks_per_shape = max_blocks;
  const int pts_per_block = (s.n + blocks_per_shape - 1) / blocks_per_shape;
  s.iters = (pts_per_block + per_iter - 1) / per_iter;
  grid = dim3((s.n + s.iters * per_iter - 1) / (s.iters * per_iter), s.B);
}

template <typename Kern>
inline int allow_big_lds(Kern kern, size_t bytes, const char *what) {
  if (bytes <= 64 * 1024) return 0;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
  if (e != hipSuccess) {
    nsdp::set_error("%s: opting in to %zu bytes of LDS failed: %s", what, bytes, hipGetErrorString(e));
    return static_cast<int>(e);
  }
  return 0;
}

inline bool shape_ok(const AttnShape &s) {
  return s.B > 0 && s.n > 0 && s.N > 0 && s.k > 0 && s.d >= 4 && s.d <= 256 && s.d % 4 == 0;
}

inline double rows(const AttnShape &s) { return static_cast<double>(s.B) * s.n * s.k; }

}  // namespace

inline dim3 attn_grid(const AttnShape &s) {
  const long long ppw = 64 / (s.d >> 2);
  const long long per_block = 4 * ppw * s.iters;  // 4 waves per workgroup
  return dim3(static_cast<unsigned>((static_cast<long long>(s.B) * s.n + per_block - 1) / per_block));
}

#define NSDP_ATTN_LAUNCH(KERNEL, ...) hipLaunchKernelGGL(KERNEL, attn_grid(s), dim3(256), 0, st, __VA_ARGS__)

#define NSDP_ATTN_LAUNCH_V(KERNEL, HASV, ...)                                                      \
  do {                                                                                             \
    if (HASV) hipLaunchKernelGGL((KERNEL<T, true>), attn_grid(s), dim3(256), 0, st, __VA_ARGS__);  \
    else hipLaunchKernelGGL((KERNEL<T, false>), attn_grid(s), dim3(256), 0, st, __VA_ARGS__);      \
  } while (0)

namespace {

template <typename T>
int attn_pre_fwd_t(const T *q, const T *kf, const T *pos, const int32_t *idx, int B, int n, int N, int k, int d,
                   int q_per_shape, T *u, void *stream) {
  constexpr double kEl = sizeof(T);
  const AttnShape s{B, n, N, k, d, q_per_shape, iters_for(k)};
  if (static_cast<long long>(B) * n * k * d <= 0) return 0;
  NSDP_REQUIRE(shape_ok(s), "attn_pre_fwd: unsupported shape (d=%d must be a multiple of 4 in [4, 256])", d);
  NSDP_REQUIRE(q && kf && pos && idx && u, "attn_pre_fwd: null pointer");
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kAttnFwd, st, 0.0,
                          kEl * (rows(s) * (2.0 * d + 1) + static_cast<double>(B) * (n + N) * d));
  NSDP_ATTN_LAUNCH(attn_pre_fwd_kernel<T>, s, q, kf, pos, idx, u);
  return nsdp::launch_status("attn_pre_fwd_kernel");
}

template <typename T>
int attn_pre_bwd_t(const T *du, const int32_t *idx, int B, int n, int N, int k, int d, int q_per_shape, float *dq,
                   float *dkf, T *dpos_acc, void *stream, const T *dq_sub = nullptr) {
  constexpr double kEl = sizeof(T);
  const AttnShape s{B, n, N, k, d, q_per_shape, iters_for(k)};
  hipStream_t st = nsdp::as_stream(stream);
  if (q_per_shape && dq && static_cast<long long>(B) * d > 0)
    NSDP_HIP_TRY(hipMemsetAsync(dq, 0, sizeof(float) * static_cast<size_t>(B) * d, st));
  if (dkf && static_cast<long long>(B) * N * d > 0)
    NSDP_HIP_TRY(hipMemsetAsync(dkf, 0, sizeof(float) * static_cast<size_t>(B) * N * d, st));
  if (static_cast<long long>(B) * n * k * d <= 0) return 0;
  NSDP_REQUIRE(shape_ok(s), "attn_pre_bwd: unsupported shape (d=%d must be a multiple of 4 in [4, 256])", d);
  NSDP_REQUIRE(du && idx && dq, "attn_pre_bwd: null pointer");
  NSDP_REQUIRE(!dq_sub || (!dkf && !q_per_shape), "attn_pre_bwd: dq_sub goes with the dq-only form (dkf NULL, per-point queries)");
  if (!dkf) {      // the caller scatters itself (inverse neighbour lists): only dq (+ the optional d(pos) accumulation) here
    NSDP_TRACE("attn_pre_bwd_stream");
    NSDP_ATTN_LAUNCH(attn_pre_bwd_kernel<T>, s, du, idx, dq, dkf, dpos_acc, dq_sub);
    return nsdp::launch_status("attn_pre_bwd_kernel");
  }
  nsdp::prof::Scope scope(nsdp::prof::kAttnBwd, st, 0.0,
                          kEl * (rows(s) * ((dpos_acc ? 3.0 : 1.0) * d + 1.0) + static_cast<double>(B) * (n + 2.0 * N) * d));
  if (q_per_shape && regtab_fits(N, d, static_cast<long long>(n) * k)) {
    // decoder: one query vector per shape.  dkf = -scatter(du), dq = +column sum of du: the register-table scatter
    return launch_regtab_scatter(du, idx, dkf, dq, dpos_acc, B, static_cast<long long>(n) * k, N, d, -1.f, 1.f, st);
  }
  if (lds_table_fits(s)) {
    const size_t lds = static_cast<size_t>(N) * d * 4;
    if (const int rc = allow_big_lds(attn_pre_bwd_lds_kernel<T>, lds, "attn_pre_bwd_lds_kernel")) return rc;
    AttnShape sl = s;
    dim3 grid;
    lds_plan(sl, grid);
    NSDP_TRACE("attn_pre_bwd_lds");
    hipLaunchKernelGGL(attn_pre_bwd_lds_kernel<T>, grid, dim3(kLdsThreads), lds, st, sl, du, idx, dq, dkf, dpos_acc);
    return nsdp::launch_status("attn_pre_bwd_lds_kernel");
  }
  NSDP_TRACE("attn_pre_bwd_atomic");
  NSDP_ATTN_LAUNCH(attn_pre_bwd_kernel<T>, s, du, idx, dq, dkf, dpos_acc);
  return nsdp::launch_status("attn_pre_bwd_kernel");
}

template <typename T>
int attn_post_fwd_t(const T *a, const T *vf, const T *pos, const int32_t *idx, const T *a_g, const T *v_g,
                    const T *residual, int B, int n, int N, int k, int d, T *y, float *lse, void *stream,
                    const T *qsub = nullptr) {
  constexpr double kEl = sizeof(T);
  const AttnShape s{B, n, N, k, d, 0, iters_for(k)};
  if (static_cast<long long>(B) * n * d <= 0) return 0;
  NSDP_REQUIRE(shape_ok(s), "attn_post_fwd: unsupported shape (d=%d must be a multiple of 4 in [4, 256])", d);
  NSDP_REQUIRE(a && pos && idx && y && lse, "attn_post_fwd: null pointer");
  NSDP_REQUIRE((a_g == nullptr) == (v_g == nullptr), "attn_post_fwd: a_g and v_g go together");
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kAttnFwd, st, 0.0,
                          kEl * (rows(s) * (2.0 * d + 1) + static_cast<double>(B) * (2.0 * n + (vf ? N : 0)) * d));
  const bool has_v = vf != nullptr;
  NSDP_REQUIRE(!qsub || (has_v && !a_g), "attn_post_fwd: qsub goes with a value table and no global token");
  NSDP_ATTN_LAUNCH_V(attn_post_fwd_kernel, has_v, s, a, vf, pos, idx, a_g, v_g, residual, y, lse, qsub);
  return nsdp::launch_status("attn_post_fwd_kernel");
}

template <typename T>
int attn_post_bwd_t(const T *dy, const T *a, const T *vf, const T *pos, const int32_t *idx, const T *a_g, const T *v_g,
                    const T *y, const T *residual, const float *lse, int B, int n, int N, int k, int d, T *da, T *dpos,
                    float *dvf, float *da_g, float *dv_g, void *stream, const T *qsub = nullptr) {
  constexpr double kEl = sizeof(T);
  const AttnShape s{B, n, N, k, d, 0, iters_for(k)};
  hipStream_t st = nsdp::as_stream(stream);
  if (dvf && static_cast<long long>(B) * N * d > 0)
    NSDP_HIP_TRY(hipMemsetAsync(dvf, 0, sizeof(float) * static_cast<size_t>(B) * N * d, st));
  if (da_g && static_cast<long long>(B) * d > 0) {
    NSDP_HIP_TRY(hipMemsetAsync(da_g, 0, sizeof(float) * static_cast<size_t>(B) * d, st));
    NSDP_HIP_TRY(hipMemsetAsync(dv_g, 0, sizeof(float) * static_cast<size_t>(B) * d, st));
  }
  if (static_cast<long long>(B) * n * k * d <= 0) return 0;
  NSDP_REQUIRE(shape_ok(s), "attn_post_bwd: unsupported shape (d=%d must be a multiple of 4 in [4, 256])", d);
  NSDP_REQUIRE(dy && a && pos && idx && y && lse && da && dpos, "attn_post_bwd: null pointer");
  NSDP_REQUIRE(vf != nullptr || dvf == nullptr, "attn_post_bwd: dvf without vf");
  // (vf without dvf: the caller scatters d(pos) itself -- nsdp_scatter_rows_onehot_bf16 -- and this call is a pure stream)
  NSDP_REQUIRE((a_g == nullptr) == (da_g == nullptr) && (a_g == nullptr) == (v_g == nullptr) &&
                   (a_g == nullptr) == (dv_g == nullptr),
               "attn_post_bwd: global-token pointers go together");
  nsdp::prof::Scope scope(nsdp::prof::kAttnBwd, st, 0.0,
                          kEl * (rows(s) * (4.0 * d + 1) + static_cast<double>(B) * (3.0 * n + (vf ? 2.0 * N : 0)) * d));
  const bool has_v = vf != nullptr;
  NSDP_REQUIRE(!qsub || (has_v && !a_g), "attn_post_bwd: qsub goes with a value table and no global token");
  if (lds_table_fits(s) && has_v && dvf && !qsub) {
    const size_t lds = static_cast<size_t>(N) * d * 4;
    if (const int rc = allow_big_lds(attn_post_bwd_lds_kernel<T, true>, lds, "attn_post_bwd_lds_kernel")) return rc;
    AttnShape sl = s;
    dim3 grid;
    lds_plan(sl, grid);
    NSDP_TRACE("attn_post_bwd_lds");
    hipLaunchKernelGGL((attn_post_bwd_lds_kernel<T, true>), grid, dim3(kLdsThreads), lds, st, sl, dy, a, vf, pos,
                       idx, a_g, v_g, y, residual, lse, da, dpos, dvf, da_g, dv_g);
    return nsdp::launch_status("attn_post_bwd_lds_kernel");
  }
  if (has_v && !dvf) {
    NSDP_TRACE("attn_post_bwd_stream");
    hipLaunchKernelGGL((attn_post_bwd_kernel<T, true, false>), attn_grid(s), dim3(256), 0, st, s, dy, a, vf, pos, idx, a_g, v_g,
                       y, residual, lse, da, dpos, dvf, da_g, dv_g, qsub);
    return nsdp::launch_status("attn_post_bwd_kernel");
  }
  NSDP_TRACE("attn_post_bwd_atomic");
  NSDP_ATTN_LAUNCH_V(attn_post_bwd_kernel, has_v, s, dy, a, vf, pos, idx, a_g, v_g, y, residual, lse, da, dpos,
                     dvf, da_g, dv_g, qsub);
  return nsdp::launch_status("attn_post_bwd_kernel");
}


template <typename T>
int attn_post_bwd_det_t(const T *dy, const T *a, const T *vf, const T *pos, const int32_t *idx, const T *a_g, const T *v_g,
                        const T *y, const T *residual, const float *lse, int B, int n, int N, int k, int d, T *da, T *dpos,
                        float *da_g, float *dv_g, float *workspace, size_t workspace_bytes, void *stream) {
  constexpr double kEl = sizeof(T);
  AttnShape s{B, n, N, k, d, 0, iters_for(k)};
  hipStream_t st = nsdp::as_stream(stream);
  if (static_cast<long long>(B) * n * k * d <= 0) {
    if (da_g && static_cast<long long>(B) * d > 0) {
      NSDP_HIP_TRY(hipMemsetAsync(da_g, 0, sizeof(float) * static_cast<size_t>(B) * d, st));
      NSDP_HIP_TRY(hipMemsetAsync(dv_g, 0, sizeof(float) * static_cast<size_t>(B) * d, st));
    }
    return 0;
  }
  NSDP_REQUIRE(shape_ok(s) && B <= 65535, "attn_post_bwd_det: unsupported shape (d=%d must be a multiple of 4 in [4, 256])", d);
  NSDP_REQUIRE(dy && a && vf && pos && idx && y && lse && da && dpos, "attn_post_bwd_det: null pointer");
  NSDP_REQUIRE((a_g == nullptr) == (da_g == nullptr) && (a_g == nullptr) == (v_g == nullptr) &&
                   (a_g == nullptr) == (dv_g == nullptr),
               "attn_post_bwd_det: global-token pointers go together");
  dim3 grid;
  shape_plan(s, grid);
  const size_t need = a_g ? static_cast<size_t>(B) * grid.x * 2 * d * sizeof(float) : 0;
  NSDP_REQUIRE(workspace_bytes >= need && (!a_g || workspace), "attn_post_bwd_det: workspace too small");
  nsdp::prof::Scope scope(nsdp::prof::kAttnBwd, st, 0.0,
                          kEl * (rows(s) * (4.0 * d + 1) + static_cast<double>(B) * (3.0 * n + N) * d));
  NSDP_TRACE("attn_post_bwd_det");
  hipLaunchKernelGGL((attn_post_bwd_shape_kernel<T>), grid, dim3(256), 0, st, s, dy, a, vf, pos, idx, a_g, v_g, y, residual,
                     lse, da, dpos, workspace);
  int rc = nsdp::launch_status("attn_post_bwd_shape_kernel");
  if (rc || !a_g) return rc;
  hipLaunchKernelGGL(global_token_reduce_kernel, dim3((2 * d + 255) / 256, B), dim3(256), 0, st, workspace,
                     static_cast<int>(grid.x), d, da_g, dv_g);
  return nsdp::launch_status("global_token_reduce_kernel");
}

}  // namespace

extern "C" {

size_t nsdp_attn_post_bwd_det_workspace_bytes(int B, int n, int k, int d) {
  if (B <= 0 || n <= 0 || d < 4) return 0;
  AttnShape s{B, n, 1, k, d, 0, iters_for(k)};
  dim3 grid;
  shape_plan(s, grid);
  return static_cast<size_t>(B) * grid.x * 2 * d * sizeof(float);
}
int nsdp_attn_post_bwd_det(const float *dy, const float *a, const float *vf, const float *pos, const int32_t *idx,
                           const float *a_g, const float *v_g, const float *y, const float *residual, const float *lse,
                           int B, int n, int N, int k, int d, float *da, float *dpos, float *da_g, float *dv_g,
                           float *workspace, size_t workspace_bytes, void *stream) {
  return attn_post_bwd_det_t<float>(dy, a, vf, pos, idx, a_g, v_g, y, residual, lse, B, n, N, k, d, da, dpos, da_g, dv_g,
                                    workspace, workspace_bytes, stream);
}
int nsdp_attn_post_bwd_det_bf16(const void *dy, const void *a, const void *vf, const void *pos, const int32_t *idx,
                                const void *a_g, const void *v_g, const void *y, const void *residual, const float *lse,
                                int B, int n, int N, int k, int d, void *da, void *dpos, float *da_g, float *dv_g,
                                float *workspace, size_t workspace_bytes, void *stream) {
  return attn_post_bwd_det_t<bf16_t>(reinterpret_cast<const bf16_t *>(dy), reinterpret_cast<const bf16_t *>(a),
                                     reinterpret_cast<const bf16_t *>(vf), reinterpret_cast<const bf16_t *>(pos), idx,
                                     reinterpret_cast<const bf16_t *>(a_g), reinterpret_cast<const bf16_t *>(v_g),
                                     reinterpret_cast<const bf16_t *>(y), reinterpret_cast<const bf16_t *>(residual), lse, B, n,
                                     N, k, d, reinterpret_cast<bf16_t *>(da), reinterpret_cast<bf16_t *>(dpos), da_g, dv_g,
                                     workspace, workspace_bytes, stream);
}

int nsdp_attn_pre_fwd(const float *q, const float *kf, const float *pos, const int32_t *idx, int B, int n,
                      int N, int k, int d, int q_per_shape, float *u, void *stream) {
  return attn_pre_fwd_t<float>(q, kf, pos, idx, B, n, N, k, d, q_per_shape, u, stream);
}
int nsdp_attn_pre_bwd(const float *du, const int32_t *idx, int B, int n, int N, int k, int d,
                      int q_per_shape, float *dq, float *dkf, float *dpos_acc, void *stream) {
  return attn_pre_bwd_t<float>(du, idx, B, n, N, k, d, q_per_shape, dq, dkf, dpos_acc, stream);
}
int nsdp_attn_pre_bwd_sub(const float *du, const int32_t *idx, int B, int n, int N, int k, int d, const float *dq_sub, float *dq,
                          void *stream) {
  return attn_pre_bwd_t<float>(du, idx, B, n, N, k, d, 0, dq, nullptr, nullptr, stream, dq_sub);
}
int nsdp_attn_pre_bwd_sub_bf16(const void *du, const int32_t *idx, int B, int n, int N, int k, int d, const void *dq_sub, float *dq,
                               void *stream) {
  return attn_pre_bwd_t<bf16_t>(reinterpret_cast<const bf16_t *>(du), idx, B, n, N, k, d, 0, dq, nullptr, nullptr, stream,
                                reinterpret_cast<const bf16_t *>(dq_sub));
}
int nsdp_attn_post_fwd(const float *a, const float *vf, const float *pos, const int32_t *idx,
                       const float *a_g, const float *v_g, const float *residual, int B, int n, int N,
                       int k, int d, float *y, float *lse, void *stream) {
  return attn_post_fwd_t<float>(a, vf, pos, idx, a_g, v_g, residual, B, n, N, k, d, y, lse, stream);
}
int nsdp_attn_post_bwd(const float *dy, const float *a, const float *vf, const float *pos,
                       const int32_t *idx, const float *a_g, const float *v_g, const float *y,
                       const float *residual, const float *lse, int B, int n, int N, int k, int d, float *da, float *dpos,
                       float *dvf, float *da_g, float *dv_g, void *stream) {
  return attn_post_bwd_t<float>(dy, a, vf, pos, idx, a_g, v_g, y, residual, lse, B, n, N, k, d, da, dpos, dvf, da_g, dv_g,
                                stream);
}

// The same two with `pos` holding u = q_i - k_j + pos (the output of nsdp_linear_bf16x3_gather_f32: pos itself is never
// materialised) and `vf` holding the table v + k: the values are u + (v + k)[idx] - qsub_i.  qsub (B, n, d).  No global token
// (the decoder's one-query-per-shape form folds q into the table on the host instead).
int nsdp_attn_post_fwd_q(const float *a, const float *vk, const float *u, const int32_t *idx, const float *qsub,
                         const float *residual, int B, int n, int N, int k, int d, float *y, float *lse, void *stream) {
  return attn_post_fwd_t<float>(a, vk, u, idx, nullptr, nullptr, residual, B, n, N, k, d, y, lse, stream, qsub);
}
int nsdp_attn_post_bwd_q(const float *dy, const float *a, const float *vk, const float *u, const int32_t *idx, const float *qsub,
                         const float *y, const float *residual, const float *lse, int B, int n, int N, int k, int d,
                         float *da, float *dpos, float *dvf, void *stream) {
  return attn_post_bwd_t<float>(dy, a, vk, u, idx, nullptr, nullptr, y, residual, lse, B, n, N, k, d, da, dpos, dvf, nullptr,
                                nullptr, stream, qsub);
}

// bf16-storage variants: every activation tensor (q, kf, vf, pos, u, a, y, residual, a_g, v_g and the gradients du, dy,
// da, dpos, dpos_acc) is bf16; lse and the scatter / reduction outputs (dq, dkf, dvf, da_g, dv_g) stay fp32.
#define B16(p) reinterpret_cast<const bf16_t *>(p)
#define B16W(p) reinterpret_cast<bf16_t *>(p)
int nsdp_attn_pre_fwd_bf16(const void *q, const void *kf, const void *pos, const int32_t *idx, int B, int n, int N, int k,
                           int d, int q_per_shape, void *u, void *stream) {
  return attn_pre_fwd_t<bf16_t>(B16(q), B16(kf), B16(pos), idx, B, n, N, k, d, q_per_shape, B16W(u), stream);
}
int nsdp_attn_pre_bwd_bf16(const void *du, const int32_t *idx, int B, int n, int N, int k, int d, int q_per_shape,
                           float *dq, float *dkf, void *dpos_acc, void *stream) {
  return attn_pre_bwd_t<bf16_t>(B16(du), idx, B, n, N, k, d, q_per_shape, dq, dkf, B16W(dpos_acc), stream);
}
int nsdp_attn_post_fwd_bf16(const void *a, const void *vf, const void *pos, const int32_t *idx, const void *a_g,
                            const void *v_g, const void *residual, int B, int n, int N, int k, int d, void *y, float *lse,
                            void *stream) {
  return attn_post_fwd_t<bf16_t>(B16(a), B16(vf), B16(pos), idx, B16(a_g), B16(v_g), B16(residual), B, n, N, k, d, B16W(y),
                                 lse, stream);
}
int nsdp_attn_post_bwd_bf16(const void *dy, const void *a, const void *vf, const void *pos, const int32_t *idx,
                            const void *a_g, const void *v_g, const void *y, const void *residual, const float *lse, int B,
                            int n, int N, int k, int d, void *da, void *dpos, float *dvf, float *da_g, float *dv_g,
                            void *stream) {
  return attn_post_bwd_t<bf16_t>(B16(dy), B16(a), B16(vf), B16(pos), idx, B16(a_g), B16(v_g), B16(y), B16(residual), lse, B, n,
                                 N, k, d, B16W(da), B16W(dpos), dvf, da_g, dv_g, stream);
}
#undef B16
#undef B16W

}  // extern "C"
